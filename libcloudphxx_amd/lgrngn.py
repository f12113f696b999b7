"""Python host mirror of the reference's lgrngn call surface, on top of the C ABI (include/lcx.h).

Same names, argument meaning and error behaviour as the reference's Python bindings
(bindings/python/lgrngn.hpp:40-140, bindings/python/lib.cpp:366-434) so that tests written
for the reference read the same here:

    opts_init = lgrngn.opts_init_t(); opts_init.dry_distros = {(kappa, rd_insol): fun}
    prtcls = lgrngn.factory(lgrngn.backend_t.HIP, opts_init)
    prtcls.init(th, rv, rhod, Cx=..., Cz=...)
    prtcls.step_sync(opts, th, rv, rhod); prtcls.step_async(opts)
    prtcls.diag_all(); prtcls.diag_sd_conc(); numpy.frombuffer(prtcls.outbuf())

Errors reported by the library are raised as RuntimeError with the library's message
(the reference throws std::runtime_error, translated by Boost.Python to RuntimeError).
"""
import ctypes as C
import enum
import numpy as np

from . import _lib

# ----------------------------------------------------------------------------- enums
class _bp_enum(enum.IntEnum):
    """str() of a Boost.Python enum value is its bare name (the reference's scripts write e.g. str(RH_formula) into CSV keys)"""
    def __str__(self):
        return self.name
    __format__ = enum.Enum.__format__


class backend_t(_bp_enum):          # lgrngn/backend.hpp:8 + the two new slots
    undefined = 0
    serial = 1
    OpenMP = 2
    CUDA = 3
    multi_CUDA = 4
    HIP = 5
    multi_HIP = 6


class kernel_t(_bp_enum):           # lgrngn/kernel.hpp:8
    undefined = 0
    geometric = 1
    golovin = 2
    hall = 3
    hall_davis_no_waals = 4
    long = 5                            # the Python module's spelling (ref: bindings/python/lib.cpp, kernel_t::Long in C++)
    Long = 5
    onishi_hall = 6
    onishi_hall_davis_no_waals = 7
    hall_pinsky_1000mb_grav = 8
    hall_pinsky_cumulonimbus = 9
    hall_pinsky_stratocumulus = 10
    vohl_davis_no_waals = 11


class vt_t(_bp_enum):               # lgrngn/terminal_velocity.hpp:8
    undefined = 0
    beard76 = 1
    beard77 = 2
    beard77fast = 3
    khvorostyanov_spherical = 4
    khvorostyanov_nonspherical = 5


class as_t(_bp_enum):               # lgrngn/advection_scheme.hpp:8
    undefined = 0
    implicit = 1
    euler = 2
    pred_corr = 3


class RH_formula_t(_bp_enum):       # lgrngn/RH_formula.hpp:8
    pv_cc = 0
    rv_cc = 1
    pv_tet = 2
    rv_tet = 3


class chem_species_t(_bp_enum):        # common/chem.hpp via bindings/python/lib.cpp:257-265 (the chemistry itself is outside this library)
    H = 0
    SO2 = 1
    O3 = 2
    H2O2 = 3
    CO2 = 4
    NH3 = 5
    HNO3 = 6
    S_VI = 7


class dbg(enum.IntFlag):
    """opts_init.dbg_flags: test / measurement switches (include/lcx.h, enum lcx_dbg; no reference counterpart)"""
    NO_COND_PRE = 1 << 0
    NO_WAVE_FLAGS = 1 << 1
    EAGER_COMPACT = 1 << 2
    SHUFFLE_PHILOX = 1 << 3
    NO_DEFERRED_SORT = 1 << 4
    COND_NO_FOLD = 1 << 5
    COND_SORTED_ORDER = 1 << 6
    NO_OVERLAP = 1 << 7
    MULTI_NO_PEER = 1 << 8
    MULTI_SERIALIZE = 1 << 9
    TAG = 1 << 10
    KPA_ARRAY = 1 << 11
    HOST_SYNC_LOOP = 1 << 12
    COND_LEAN_R3 = 1 << 13
    EXCH_SORT_NOW = 1 << 14
    COND_TOMS_TWO_PASS = 1 << 15
    FINISH_STAGED = 1 << 16
    NO_RANK_OVERLAP = 1 << 17
    RANK_BY_COUNTING = 1 << 18
    COND_FOLD = 1 << 19
    COND_NO_LIST = 1 << 20
    COND_WQ = 1 << 21
    COND_WQ_CAP128 = 1 << 22
    COND_WQ_PF = 1 << 23
    COND_WQ_PF2 = 1 << 24
    COND_BUDGET = 1 << 25
    COND_PROBE = 1 << 26
    COND_NO_FUSED_SUBSTEPS = 1 << 27
    VTERM_INVALID_OWN_PASS = 1 << 28


class cond_kernel(enum.IntEnum):
    """enum lcx_cond_kernel (include/lcx.h): the kernel of an object's last condensation launch, particles_t.mode()"""
    none = 0
    strict = 1
    fast_per_droplet_setup = 2
    lean = 3
    lean_sorted = 4
    fold_toms748 = 5
    lean_toms748 = 6
    lean_toms748_sorted = 7
    toms748_two_pass = 8
    lean_r3 = 9
    fold_lean = 10
    lean_wq = 11
    per_particle = 12
    substeps = 13


class src_t(_bp_enum):              # lgrngn/ccn_source.hpp:8
    off = 0
    simple = 1
    matching = 2


# common::output_names (common/output.hpp:26-42)
output_names = ["HNO3", "NH3", "CO2", "SO2", "H2O2", "O3", "S_VI", "H", "liquid_volume", "dry_volume",
                "particle_number", "ice_mass", "liquid_number", "ice_number"]

# ----------------------------------------------------------------------------- C structs (include/lcx.h)
DISTRO_FN = C.CFUNCTYPE(C.c_double, C.c_double, C.c_void_p)


class _distro_c(C.Structure):
    _fields_ = [("kappa", C.c_double), ("rd_insol", C.c_double), ("fn", DISTRO_FN), ("user", C.c_void_p),
                ("n_modes", C.c_int), ("mean_rd", C.c_double * 4), ("sdev", C.c_double * 4), ("n_stp", C.c_double * 4)]


class _dry_size_c(C.Structure):
    _fields_ = [("kappa", C.c_double), ("rd_insol", C.c_double), ("radius", C.c_double), ("conc", C.c_double),
                ("sd_count", C.c_int)]


class _opts_init_c(C.Structure):
    _fields_ = [
        ("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int),
        ("dx", C.c_double), ("dy", C.c_double), ("dz", C.c_double), ("dt", C.c_double),
        ("sstp_cond", C.c_int), ("sstp_coal", C.c_int), ("sstp_cond_act", C.c_int), ("sstp_chem", C.c_int),
        ("x0", C.c_double), ("y0", C.c_double), ("z0", C.c_double), ("x1", C.c_double), ("y1", C.c_double), ("z1", C.c_double),
        ("sd_conc", C.c_ulonglong),
        ("sd_conc_large_tail", C.c_int), ("aerosol_independent_of_rhod", C.c_int), ("variable_dt_switch", C.c_int),
        ("sd_const_multi", C.c_ulonglong), ("n_sd_max", C.c_ulonglong),
        ("kernel", C.c_int), ("terminal_velocity", C.c_int), ("adve_scheme", C.c_int), ("RH_formula", C.c_int),
        ("kernel_parameters", C.POINTER(C.c_double)), ("n_kernel_parameters", C.c_int),
        ("chem_switch", C.c_int), ("coal_switch", C.c_int), ("sedi_switch", C.c_int), ("subs_switch", C.c_int),
        ("rlx_switch", C.c_int), ("turb_adve_switch", C.c_int), ("turb_cond_switch", C.c_int), ("turb_coal_switch", C.c_int),
        ("ice_switch", C.c_int), ("exact_sstp_cond", C.c_int), ("sstp_cond_mix", C.c_int), ("adaptive_sstp_cond", C.c_int),
        ("time_dep_ice_nucl", C.c_int),
        ("RH_max", C.c_double),
        ("sstp_cond_adapt_drw2_eps", C.c_double), ("sstp_cond_adapt_drw2_max", C.c_double), ("rc2_T", C.c_double),
        ("rng_seed", C.c_int), ("rng_seed_init", C.c_int), ("rng_seed_init_switch", C.c_int),
        ("dev_count", C.c_int), ("dev_id", C.c_int),
        ("w_LS", C.POINTER(C.c_double)), ("n_w_LS", C.c_int),
        ("SGS_mix_len", C.POINTER(C.c_double)), ("n_SGS_mix_len", C.c_int),
        ("aerosol_conc_factor", C.POINTER(C.c_double)), ("n_aerosol_conc_factor", C.c_int),
        ("rd_min", C.c_double), ("rd_max", C.c_double),
        ("no_ccn_at_init", C.c_int), ("open_side_walls", C.c_int), ("periodic_topbot_walls", C.c_int),
        ("src_type", C.c_int),
        ("th_dry", C.c_int), ("const_p", C.c_int),
        ("diag_incloud_time", C.c_int),
        ("dry_distros", C.POINTER(_distro_c)), ("n_dry_distros", C.c_int),
        ("dry_sizes", C.POINTER(_dry_size_c)), ("n_dry_sizes", C.c_int),
        ("n_x_tot", C.c_int), ("n_x_bfr", C.c_int), ("bcond_lft", C.c_int), ("bcond_rgt", C.c_int),
        ("strict_fp", C.c_int), ("cond_solver", C.c_int), ("reorder_every", C.c_int), ("stream_ordered", C.c_int),
        ("dbg_flags", C.c_uint), ("dbg_cond_budget", C.c_int), ("dbg_pack_delay_us", C.c_int),
    ]


class _opts_c(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("adve", "sedi", "subs", "cond", "coal", "src", "rlx", "rcyc", "turb_adve",
                                        "turb_cond", "turb_coal", "ice_nucl", "chem_dsl", "chem_dsc", "chem_rct")] + \
               [("RH_max", C.c_double), ("dt", C.c_double)]


class _arrinfo_c(C.Structure):
    _fields_ = [("data", C.c_void_p), ("strides", C.POINTER(C.c_ssize_t)), ("on_device", C.c_int)]


# ----------------------------------------------------------------------------- user-facing option structs
class lognormal:
    """Built-in sum of lognormal modes n(ln rd) (common/lognormal.hpp:25-37); evaluated natively,
    no Python callback per super-droplet."""

    def __init__(self, mean_rd, sdev, n_stp):
        self.mean_rd = np.atleast_1d(np.asarray(mean_rd, dtype=float))
        self.sdev = np.atleast_1d(np.asarray(sdev, dtype=float))
        self.n_stp = np.atleast_1d(np.asarray(n_stp, dtype=float))
        assert len(self.mean_rd) == len(self.sdev) == len(self.n_stp) <= 4

    def __call__(self, lnrd):
        return float(np.sum(self.n_stp / np.sqrt(2 * np.pi) / np.log(self.sdev) *
                            np.exp(-(lnrd - np.log(self.mean_rd)) ** 2 / 2. / np.log(self.sdev) ** 2)))


class expvolume:
    """Built-in exponential-in-volume spectrum n(ln r) = 3 n_zero (r / r_zero)^3 exp(-(r / r_zero)^3) (Shima et al. 2009; the function
    the reference's Golovin test passes, tests/python/physics/coalescence_golovin.py:41-44); evaluated natively (lcx_distro_t.n_modes = -1)."""

    def __init__(self, r_zero, n_zero):
        self.r_zero, self.n_zero = float(r_zero), float(n_zero)

    def __call__(self, lnrd):
        q = np.exp(lnrd) ** 3 / self.r_zero ** 3
        return float(self.n_zero * 3. * q * np.exp(-q))


class opts_init_t:
    """opts_init_t<real_t> with the reference's field names and defaults (opts_init.hpp:29-253)."""

    def __init__(self):
        self.dry_distros = {}
        self.dry_sizes = {}
        self.nx = self.ny = self.nz = 0
        self.dx = self.dy = self.dz = 1.
        self.dt = 0.
        self.sstp_cond = self.sstp_coal = self.sstp_cond_act = self.sstp_chem = 1
        self.x0 = self.y0 = self.z0 = 0.
        self.x1 = self.y1 = self.z1 = 1.
        self.sd_conc = 0
        self.sd_conc_large_tail = False
        self.aerosol_independent_of_rhod = False
        self.variable_dt_switch = False
        self.sd_const_multi = 0
        self.n_sd_max = 0
        self.kernel = kernel_t.undefined
        self.terminal_velocity = vt_t.undefined
        self.adve_scheme = as_t.implicit
        self.RH_formula = RH_formula_t.pv_cc
        self.kernel_parameters = np.zeros(0)
        self.chem_switch = False
        self.coal_switch = True
        self.sedi_switch = True
        self.subs_switch = False
        self.rlx_switch = False
        self.turb_adve_switch = self.turb_cond_switch = self.turb_coal_switch = False
        self.ice_switch = False
        self.exact_sstp_cond = False
        self.sstp_cond_mix = True
        self.adaptive_sstp_cond = False
        self.time_dep_ice_nucl = False
        self.RH_max = .95
        self.sstp_cond_adapt_drw2_eps, self.sstp_cond_adapt_drw2_max, self.rc2_T = 1e-4, 4., 10.
        self.rng_seed = 44
        self.rng_seed_init = 44
        self.rng_seed_init_switch = False
        self.dev_count = 0
        self.dev_id = -1
        self.w_LS = np.zeros(0)
        self.SGS_mix_len = np.zeros(0)
        self.aerosol_conc_factor = np.zeros(0)
        self.rd_min = self.rd_max = -1.
        self.no_ccn_at_init = False
        self.open_side_walls = False
        self.periodic_topbot_walls = False
        self.src_type = src_t.off
        self.th_dry = True
        self.const_p = False
        self.diag_incloud_time = False
        # fields of the parts that are outside this library (chemistry, aerosol sources / relaxation): kept so that scripts written
        # for the reference can set and print them; the matching switches make the constructor throw
        self.chem_rho = 0.
        self.src_x0 = self.src_y0 = self.src_z0 = self.src_x1 = self.src_y1 = self.src_z1 = 0.
        self.rlx_dry_distros = {}
        self.rlx_bins, self.rlx_timescale, self.rlx_sd_per_bin, self.supstp_rlx = 0, 1., 0., 1
        # extensions (include/lcx.h)
        self.n_x_tot = 0
        self.n_x_bfr = 0
        self.bcond_lft = self.bcond_rgt = 0
        # the API default since round 5: fast arithmetic with the reference's TOMS748 iterates (held to the strict bars in every test);
        # strict_fp = True is the opt-in IEEE-order mode, cond_solver = 0 the lean bracketed secant of bench.py's headline
        self.strict_fp = False
        self.cond_solver = 1          # fast arithmetic only: 1 the reference's TOMS748 iterates, 0 lean bracketed secant (include/lcx.h)
        self.reorder_every = 0
        self.stream_ordered = False   # device arrays: step_sync returns once its work is queued on lcx_stream (include/lcx.h)
        # test / measurement switches (include/lcx.h, enum lcx_dbg): all off in production
        self.dbg_flags = 0
        self.dbg_cond_budget = 0
        self.dbg_pack_delay_us = 0

    def _to_c(self, keep):
        c = _opts_init_c()
        special = {"kernel_parameters", "n_kernel_parameters", "w_LS", "n_w_LS", "SGS_mix_len", "n_SGS_mix_len", "aerosol_conc_factor",
                   "n_aerosol_conc_factor", "dry_distros", "n_dry_distros", "dry_sizes", "n_dry_sizes"}
        for name, ctype in _opts_init_c._fields_:
            if name in special:
                continue
            v = getattr(self, name)
            setattr(c, name, float(v) if ctype is C.c_double else int(v))
        for name in ("kernel_parameters", "w_LS", "SGS_mix_len", "aerosol_conc_factor"):
            arr = np.ascontiguousarray(np.asarray(getattr(self, name), dtype=np.float64).ravel())
            keep.append(arr)
            setattr(c, name, arr.ctypes.data_as(C.POINTER(C.c_double)))
            setattr(c, "n_" + name, arr.size)
        # std::map iteration order = sorted by (kappa, rd_insol) (distro_t.hpp:17-21)
        keys = sorted(self.dry_distros.keys())
        darr = (_distro_c * max(1, len(keys)))()
        for i, k in enumerate(keys):
            fun = self.dry_distros[k]
            darr[i].kappa, darr[i].rd_insol = float(k[0]), float(k[1])
            if isinstance(fun, lognormal):
                darr[i].n_modes = len(fun.mean_rd)
                for m in range(len(fun.mean_rd)):
                    darr[i].mean_rd[m], darr[i].sdev[m], darr[i].n_stp[m] = fun.mean_rd[m], fun.sdev[m], fun.n_stp[m]
            elif isinstance(fun, expvolume):
                darr[i].n_modes = -1
                darr[i].mean_rd[0], darr[i].n_stp[0] = fun.r_zero, fun.n_zero
            else:
                cb = DISTRO_FN(lambda lnrd, user, _f=fun: float(_f(lnrd)))
                keep.append(cb)
                darr[i].fn = cb
        keep.append(darr)
        c.dry_distros = C.cast(darr, C.POINTER(_distro_c))
        c.n_dry_distros = len(keys)
        # dry_sizes: {(kappa, rd_insol): {radius: [conc, count]}}
        flat = []
        for k in sorted(self.dry_sizes.keys()):
            for r in sorted(self.dry_sizes[k].keys()):
                conc, cnt = self.dry_sizes[k][r]
                flat.append((float(k[0]), float(k[1]), float(r), float(conc), int(cnt)))
        sarr = (_dry_size_c * max(1, len(flat)))()
        for i, t in enumerate(flat):
            sarr[i].kappa, sarr[i].rd_insol, sarr[i].radius, sarr[i].conc, sarr[i].sd_count = t
        keep.append(sarr)
        c.dry_sizes = C.cast(sarr, C.POINTER(_dry_size_c))
        c.n_dry_sizes = len(flat)
        return c


class opts_t:
    """opts_t<real_t> (opts.hpp:20-50)."""

    def __init__(self):
        self.adve = self.sedi = self.cond = self.coal = True
        self.subs = self.src = self.rlx = self.rcyc = False
        self.turb_adve = self.turb_cond = self.turb_coal = self.ice_nucl = False
        self.chem_dsl = self.chem_dsc = self.chem_rct = False
        self.RH_max = 44.
        self.dt = -1.
        self.chem = False    # accepted for source compatibility with the reference's tests (no-op)
        self.chem_gas = {}   # ambient trace gases, aerosol sources: holders only (chemistry / sources are outside this library)
        self.src_dry_distros = {}
        self.src_dry_sizes = {}

    def _to_c(self):
        c = _opts_c()
        for name, ctype in _opts_c._fields_:
            v = getattr(self, name)
            setattr(c, name, float(v) if ctype is C.c_double else int(bool(v)))
        return c


# ----------------------------------------------------------------------------- particles
class DeviceArray:
    """A device-resident n-d array handed to init/step_sync instead of a numpy array
    (extension: lcx_arrinfo_t.on_device).  `ptr` is a raw device address, strides in elements."""

    def __init__(self, ptr, shape, strides=None):
        self.ptr = int(ptr)
        self.shape = tuple(shape)
        if strides is None:
            strides, acc = [], 1
            for s in reversed(self.shape):
                strides.insert(0, acc)
                acc *= s
        self.strides = tuple(strides)


class DeviceArrays:
    """The field of a multi-device object kept as one device array per slab (extension: lcx_arrinfo_t.on_device == 2):
    ptrs[i] = raw address of slab i's planes on slab i's device, shape = the shape of ONE slab's array (all alike)."""

    def __init__(self, ptrs, shape, strides=None):
        self.ptrs = [int(p) for p in ptrs]
        one = DeviceArray(0, shape, strides)
        self.shape, self.strides = one.shape, one.strides


class particles_t:
    """particles_t<real_t, HIP>; with multi=True particles_t<real_t, multi_HIP>: one object over opts_init.dev_count devices of
    this process (lcx_create_multi, include/lcx.h) -- same calls, global arrays."""

    def __init__(self, opts_init, real_t=np.float64, lib=None, prefix="lcx_", multi=False, _borrowed=None):
        self._lib = lib if lib is not None else _lib.load()
        self._px = prefix
        self._keep = []
        self.real_t = np.dtype(real_t)
        self.opts_init = opts_init
        self._arr_keep = []
        self._owner = None
        if _borrowed is not None:                  # a slab of a multi-device object: the parent owns the handle
            self._h, self._owner = _borrowed
            return
        c = opts_init._to_c(self._keep)
        self._h = C.c_void_p()
        self._chk(self._f("create_multi" if multi else "create")(C.byref(c), C.c_int(self.real_t.itemsize), C.byref(self._h)))

    # -- multi-device objects
    @property
    def dev_count(self):
        n = C.c_int()
        self._chk(self._f("multi_dev_count")(self._h, C.byref(n)))
        return n.value

    def slab(self, i):
        """particles_t view of slab i of a multi-device object (state getters, set_particles, random replay), owned by the parent"""
        h = C.c_void_p()
        self._chk(self._f("multi_slab")(self._h, C.c_int(i), C.byref(h)))
        return particles_t(self.opts_init, self.real_t, lib=self._lib, prefix=self._px, _borrowed=(h, self))

    # -- plumbing
    def _f(self, name):
        f = getattr(self._lib, self._px + name)
        return f

    def _chk(self, rc):
        if rc != 0:
            f = self._f("last_error")
            f.restype = C.c_char_p
            raise RuntimeError(f().decode())

    def __del__(self):
        try:
            if getattr(self, "_owner", None) is not None:
                return
            if getattr(self, "_h", None) and self._h.value:
                self._f("destroy")(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def _arr(self, a):
        if a is None:
            return None
        ai = _arrinfo_c()
        if isinstance(a, DeviceArrays):
            st = (C.c_ssize_t * max(1, len(a.strides)))(*a.strides) if a.strides else (C.c_ssize_t * 1)(1)
            tab = (C.c_void_p * len(a.ptrs))(*a.ptrs)
            ai.data, ai.strides, ai.on_device = C.cast(tab, C.c_void_p).value, C.cast(st, C.POINTER(C.c_ssize_t)), 2
            self._arr_keep.append((st, tab))
            return ai
        if isinstance(a, DeviceArray):
            st = (C.c_ssize_t * max(1, len(a.strides)))(*a.strides) if a.strides else (C.c_ssize_t * 1)(1)
            ai.data, ai.strides, ai.on_device = a.ptr, C.cast(st, C.POINTER(C.c_ssize_t)), 1
            self._arr_keep.append(st)
            return ai
        if a.dtype != self.real_t:
            raise RuntimeError("libcloudph++: array dtype does not match real_t")
        if a.ndim == 0 or a.size == 0:
            raise RuntimeError("libcloudph++: empty array")
        if any(s % a.itemsize for s in a.strides):
            raise RuntimeError("libcloudph++: unsupported strides")
        # the reference passes numpy strides in elements (bindings/python/lgrngn.hpp np2ai)
        st = (C.c_ssize_t * a.ndim)(*[s // a.itemsize for s in a.strides])
        ai.data, ai.strides, ai.on_device = a.ctypes.data, C.cast(st, C.POINTER(C.c_ssize_t)), 0
        self._arr_keep.append((a, st))
        return ai

    @staticmethod
    def _p(ai):
        return C.byref(ai) if ai is not None else None

    # -- API (particles.hpp:17-134)
    def init(self, th, rv, rhod, p=None, Cx=None, Cy=None, Cz=None):
        self._arr_keep = []
        a = [self._arr(x) for x in (th, rv, rhod, p, Cx, Cy, Cz)]
        self._chk(self._f("init")(self._h, *[self._p(x) for x in a]))

    def sync_in(self, th, rv, rhod=None, Cx=None, Cy=None, Cz=None, diss_rate=None):
        self._arr_keep = []
        a = [self._arr(x) for x in (th, rv, rhod, Cx, Cy, Cz, diss_rate)]
        self._chk(self._f("sync_in")(self._h, *[self._p(x) for x in a]))

    def step_cond(self, opts, th, rv):
        self._arr_keep = []
        a = [self._arr(x) for x in (th, rv)]
        oc = opts._to_c()
        self._chk(self._f("step_cond")(self._h, C.byref(oc), *[self._p(x) for x in a]))

    def step_sync(self, opts, th, rv, rhod=None, Cx=None, Cy=None, Cz=None, diss_rate=None):
        self._arr_keep = []
        a = [self._arr(x) for x in (th, rv, rhod, Cx, Cy, Cz, diss_rate)]
        oc = opts._to_c()
        self._chk(self._f("step_sync")(self._h, C.byref(oc), *[self._p(x) for x in a]))

    def step_async(self, opts):
        oc = opts._to_c()
        self._chk(self._f("step_async")(self._h, C.byref(oc)))

    def _diag0(name):
        def f(self):
            self._chk(self._f(name)(self._h))
        f.__name__ = name
        return f

    def _diag2(name):
        def f(self, a, b):
            self._chk(self._f(name)(self._h, C.c_double(a), C.c_double(b)))
        f.__name__ = name
        return f

    def _diagk(name):
        def f(self, k):
            self._chk(self._f(name)(self._h, C.c_int(int(k))))
        f.__name__ = name
        return f

    diag_sd_conc = _diag0("diag_sd_conc")
    diag_pressure = _diag0("diag_pressure")
    diag_temperature = _diag0("diag_temperature")
    diag_RH = _diag0("diag_RH")
    diag_all = _diag0("diag_all")
    diag_vel_div = _diag0("diag_vel_div")
    diag_water = _diag0("diag_water")
    diag_precip_rate = _diag0("diag_precip_rate")
    diag_RH_ge_Sc = _diag0("diag_RH_ge_Sc")
    diag_rw_ge_rc = _diag0("diag_rw_ge_rc")

    def diag_wet_mass_dens(self, rad, sig0):
        self._chk(self._f("diag_wet_mass_dens")(self._h, C.c_double(rad), C.c_double(sig0)))
    diag_max_rw = _diag0("diag_max_rw")
    diag_dry_rng = _diag2("diag_dry_rng")
    diag_wet_rng = _diag2("diag_wet_rng")
    diag_kappa_rng = _diag2("diag_kappa_rng")
    diag_dry_rng_cons = _diag2("diag_dry_rng_cons")
    diag_wet_rng_cons = _diag2("diag_wet_rng_cons")
    diag_kappa_rng_cons = _diag2("diag_kappa_rng_cons")
    diag_dry_mom = _diagk("diag_dry_mom")
    diag_wet_mom = _diagk("diag_wet_mom")
    diag_kappa_mom = _diagk("diag_kappa_mom")
    diag_incloud_time_mom = _diagk("diag_incloud_time_mom")
    diag_up_mom = _diagk("diag_up_mom")
    diag_vp_mom = _diagk("diag_vp_mom")
    diag_wp_mom = _diagk("diag_wp_mom")
    diag_water_cons = _diag0("diag_water_cons")

    def outbuf(self):
        """bytes-like view of n_cell reals (use numpy.frombuffer(..., dtype=real_t), default float64)."""
        ptr, n = C.c_void_p(), C.c_size_t()
        self._chk(self._f("outbuf")(self._h, C.byref(ptr), C.byref(n)))
        buf = (C.c_char * (n.value * self.real_t.itemsize)).from_address(ptr.value)
        return memoryview(buf)

    def outbuf_array(self):
        return np.frombuffer(self.outbuf(), dtype=self.real_t).copy()

    def get_attr(self, name):
        n = C.c_size_t()
        self._chk(self._f("get_attr")(self._h, name.encode(), None, C.c_size_t(0), C.byref(n)))
        out = np.empty(n.value, dtype=self.real_t)
        self._chk(self._f("get_attr")(self._h, name.encode(), out.ctypes.data_as(C.c_void_p), C.c_size_t(out.size), C.byref(n)))
        return out

    def diag_puddle(self):
        out = (C.c_double * len(output_names))()
        self._chk(self._f("diag_puddle")(self._h, out))
        return {nm: out[i] for i, nm in enumerate(output_names)}

    # -- introspection hooks (no reference counterpart)
    @property
    def n_part(self):
        n = C.c_size_t()
        self._chk(self._f("n_part")(self._h, C.byref(n)))
        return n.value

    @property
    def n_cell(self):
        n = C.c_size_t()
        self._chk(self._f("n_cell")(self._h, C.byref(n)))
        return n.value

    def state_u64(self, name):
        n = C.c_size_t()
        self._chk(self._f("get_state_u64")(self._h, name.encode(), None, C.c_size_t(0), C.byref(n)))
        out = np.empty(n.value, dtype=np.uint64)
        self._chk(self._f("get_state_u64")(self._h, name.encode(), out.ctypes.data_as(C.POINTER(C.c_ulonglong)),
                                           C.c_size_t(out.size), C.byref(n)))
        return out

    def state_real(self, name):
        n = C.c_size_t()
        self._chk(self._f("get_state_real")(self._h, name.encode(), None, C.c_size_t(0), C.byref(n)))
        out = np.empty(n.value, dtype=np.float64)
        self._chk(self._f("get_state_real")(self._h, name.encode(), out.ctypes.data_as(C.POINTER(C.c_double)),
                                            C.c_size_t(out.size), C.byref(n)))
        return out

    def set_particles(self, n, rd3, rw2, kpa, vt, x=None, y=None, z=None):
        def d(a):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dtype=np.float64)
            self._arr_keep.append(a)
            return a.ctypes.data_as(C.POINTER(C.c_double))
        nn = np.ascontiguousarray(n, dtype=np.uint64)
        self._chk(self._f("set_particles")(self._h, C.c_size_t(nn.size), nn.ctypes.data_as(C.POINTER(C.c_ulonglong)),
                                           d(rd3), d(rw2), d(kpa), d(vt), d(x), d(y), d(z)))

    def rng_replay_push(self, kind, data):
        a = np.ascontiguousarray(data, dtype=np.float64)
        self._chk(self._f("rng_replay_push")(self._h, C.c_int(kind), a.ctypes.data_as(C.POINTER(C.c_double)), C.c_size_t(a.size)))

    def set_state_real(self, name, data):
        """test hook of tests/test_hip_reverse_replay.py (the product sets "tag" only; the oracle also rw2, th, rv)"""
        a = np.ascontiguousarray(data, dtype=np.float64)
        self._chk(self._f("set_state_real")(self._h, name.encode(), a.ctypes.data_as(C.POINTER(C.c_double)), C.c_size_t(a.size)))

    def rng_replay_pending(self):
        n = C.c_size_t()
        self._chk(self._f("rng_replay_pending")(self._h, C.byref(n)))
        return n.value

    def rng_dump(self, call, which):
        """what coalescence call `call` of the last step_async consumed of the generator (lcx_rng_dump; needs dbg.TAG)"""
        n = C.c_size_t()
        self._chk(self._f("rng_dump")(self._h, C.c_int(call), C.c_int(which), None, C.c_size_t(0), C.byref(n)))
        out = np.empty(n.value, dtype=np.float64)
        self._chk(self._f("rng_dump")(self._h, C.c_int(call), C.c_int(which), out.ctypes.data_as(C.POINTER(C.c_double)),
                                      C.c_size_t(out.size), C.byref(n)))
        return out

    def stage(self, name, opts=None):
        oc = opts._to_c() if opts is not None else None
        self._chk(self._f("stage")(self._h, name.encode(), C.byref(oc) if oc is not None else None))

    def timings(self):
        cap = 64
        names = (C.c_char_p * cap)()
        ms = (C.c_double * cap)()
        n = C.c_size_t()
        self._chk(self._f("timings")(self._h, names, ms, C.c_size_t(cap), C.byref(n)))
        return {names[i].decode(): ms[i] for i in range(n.value)}

    def mode(self):
        """what the OBJECT runs (lcx_get_state_u64 "raw_mode"): (strict_fp, cond_solver, cond_kernel of the last condensation launch,
        dbg_flags) -- a test that claims the benchmarked or the default path asserts these instead of trusting what it meant to set"""
        v = self.state_u64("raw_mode")
        return bool(v[0]), int(v[1]), cond_kernel(int(v[2])), dbg(int(v[3]))

    def set_profiling(self, on):
        self._chk(self._f("set_profiling")(self._h, C.c_int(int(on))))

    # -- 1-D decomposition primitives
    def migrate_counts(self):
        l, r = C.c_size_t(), C.c_size_t()
        self._chk(self._f("migrate_counts")(self._h, C.byref(l), C.byref(r)))
        return l.value, r.value

    def courant_halo_count(self, which):
        f = self._f("courant_halo_count")
        f.restype = C.c_size_t
        return int(f(self._h, C.c_int(which)))

    def courant_halo_pack(self, which, side, ptr):
        self._chk(self._f("courant_halo_pack")(self._h, C.c_int(which), C.c_int(side), C.c_void_p(ptr)))

    def courant_halo_unpack(self, which, side, ptr):
        self._chk(self._f("courant_halo_unpack")(self._h, C.c_int(which), C.c_int(side), C.c_void_p(ptr)))

    def migrate_record_bytes(self):
        f = self._f("migrate_record_bytes")
        f.restype = C.c_size_t
        return f(self._h)

    def migrate_pack(self, side, x_rmt, buf_ptr, cap_bytes):
        self._chk(self._f("migrate_pack")(self._h, C.c_int(side), C.c_double(x_rmt), C.c_void_p(buf_ptr), C.c_size_t(cap_bytes)))

    def migrate_unpack(self, buf_ptr, count):
        self._chk(self._f("migrate_unpack")(self._h, C.c_void_p(buf_ptr), C.c_size_t(count)))

    def migrate_finish(self, opts):
        oc = opts._to_c()
        self._chk(self._f("migrate_finish")(self._h, C.byref(oc)))

    # -- the device-driven exchange for one process per GPU (include/lcx.h lcx_exch_*; driven by libcloudphxx_amd.multi)
    def exch_enable(self, nx_min):
        cap = C.c_size_t()
        self._chk(self._f("exch_enable")(self._h, C.c_int(nx_min), C.byref(cap)))
        return cap.value

    def exch_buffers(self):
        """addresses of (outbox to the left, outbox to the right, inbox from the left, inbox from the right)"""
        p4 = (C.c_void_p * 4)()
        self._chk(self._f("exch_buffers")(self._h, p4))
        return [int(v or 0) for v in p4]

    def exch_message_bytes(self, n_rec):
        f = self._f("exch_message_bytes")
        f.restype = C.c_size_t
        return f(self._h, C.c_size_t(n_rec))

    def exch_pack(self, has_lft, lft_x1, has_rgt, rgt_x0, next_cap_lft=0, next_cap_rgt=0):
        self._chk(self._f("exch_pack")(self._h, C.c_int(bool(has_lft)), C.c_double(lft_x1), C.c_int(bool(has_rgt)), C.c_double(rgt_x0),
                                       C.c_uint(next_cap_lft), C.c_uint(next_cap_rgt)))

    def exch_sort_interior(self):
        self._chk(self._f("exch_sort_interior")(self._h))

    def exch_unpack(self, from_lft, from_rgt, have_lft=0xFFFFFFFF, have_rgt=0xFFFFFFFF):
        self._chk(self._f("exch_unpack")(self._h, C.c_int(bool(from_lft)), C.c_int(bool(from_rgt)), C.c_uint(have_lft), C.c_uint(have_rgt)))

    def exch_finish(self, opts):
        """the step's one host synchronisation: (complete, record) -- record = [dead, out_lft, out_rgt, in_lft, in_rgt, flags, crowded
        cells, largest cell, shift, next_cap_lft, next_cap_rgt, 0]"""
        oc = opts._to_c()
        rec = (C.c_uint * 12)()
        done = C.c_int()
        self._chk(self._f("exch_finish")(self._h, C.byref(oc), rec, C.byref(done)))
        return bool(done.value), list(rec)

    def stream(self):
        """the hipStream_t (as an integer) every launch of this object is queued on; 0 for an engine without one"""
        p = C.c_void_p()
        self._chk(self._f("stream")(self._h, C.byref(p)))
        return int(p.value or 0)


def factory(backend, opts_init, real_t=np.float64):
    """factory<real_t>(backend, opts_init)  (factory.hpp:12-15, src/lib.cpp:13-40).

    Only the HIP backends exist in this library; asking for a backend that was not compiled in
    raises, as the reference does (src/lib.cpp:21,27,33,38)."""
    backend = backend_t(backend)
    if backend == backend_t.HIP or backend == backend_t.CUDA:
        return particles_t(opts_init, real_t)
    if backend in (backend_t.multi_HIP, backend_t.multi_CUDA):
        # one object, opts_init.dev_count devices of this process (0: all visible) -- the reference's multi_CUDA.  The SPMD flavour
        # (one process per GPU, torch.distributed) is libcloudphxx_amd.multi.particles_multi_t.
        return particles_t(opts_init, real_t, multi=True)
    raise RuntimeError("libcloudph++: backend %s not compiled in this library (available: HIP, multi_HIP)" % backend.name)
