"""Instruction mix of one kernel: the STATIC histogram of its gfx950 ISA (hipcc -save-temps of csrc/lcx_core.hip) next to the DYNAMIC
per-launch counts of a tools/profile_round.sh run (traffic.json), so that the instructions a wave executes are accounted for.

    python3 tools/isa_histogram.py <kernel-name-substring> [traffic.json] > profiles/rNN_<kernel>_instruction_mix.txt
"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = sys.argv[1]
tj = sys.argv[2] if len(sys.argv) > 2 else None
with tempfile.TemporaryDirectory() as d:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                           "-Wno-unused-result", "-save-temps", "-o", os.path.join(d, "x.so"),
                           os.path.join(ROOT, "libcloudphxx_amd", "csrc", "lcx_core.hip")], cwd=d, stderr=subprocess.DEVNULL)
    s = open(os.path.join(d, "lcx_core-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
names = sorted(set(m.group(1) for m in re.finditer(r"^(_Z\S+):\s", s, re.M)))
demangled = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
pick = [(n, dm) for n, dm in zip(names, demangled) if pat in dm and ("double" in dm or "<double" not in pat and "float" not in dm)]
pick.sort(key=lambda nd: 0 if "k_cond_fast_fold<double, 3>" in nd[1] else 1 if "<double, 3, false>" in nd[1] else 1 if "k_cond_lean<double, 7, true, 0>" in nd[1] else 2)   # the production kernel first
if not pick:
    sys.exit("no kernel matches " + pat)
name, dm = pick[0]
body = re.search(r"^" + re.escape(name) + r":.*?\n(.*?)\.Lfunc_end\d+:", s, re.S | re.M).group(1)
desc = re.search(r"\.amdhsa_kernel " + re.escape(name) + r"(.*?)\.end_amdhsa_kernel", s, re.S).group(1)
g = lambda k: re.search(k + r" (\d+)", desc).group(1)
cnt = collections.Counter()
for line in body.split("\n"):
    line = line.strip()
    if not line or line[0] in ";." or line.endswith(":"):
        continue
    cnt[line.split()[0]] += 1


def group(op):
    if re.match(r"v_(fma|fmac|mul|add)_f64", op): return "VALU fp64 fma / mul / add"
    if re.match(r"v_(rcp|rsq|sqrt)_f64", op): return "VALU fp64 transcendental (rcp, rsq, sqrt)"
    if re.match(r"v_(ldexp|frexp|rndne|fract|trunc|floor|ceil|div_scale|div_fmas|div_fixup|min|max|cvt_i32)_f64|v_cmp_class_f64", op) or (op.endswith("_f64") and op.startswith("v_") and not op.startswith("v_cmp")): return "VALU fp64 other (ldexp, frexp, rounding, min / max)"
    if re.match(r"v_cmp\w*_f64", op): return "VALU compare fp64"
    if re.match(r"v_(exp|log|rcp|rsq|sqrt)_f32", op): return "VALU fp32 transcendental (seeds of cbrt / exp)"
    if re.match(r"v_cvt", op): return "VALU convert"
    if op.startswith("v_cmp"): return "VALU compare (integer / fp32)"
    if op.startswith("v_cndmask"): return "VALU select (v_cndmask)"
    if op.startswith("v_mov") or op.startswith("v_accvgpr") or op.startswith("v_readlane") or op.startswith("v_writelane") or op.startswith("v_readfirstlane"): return "VALU move / lane access"
    if op.startswith("v_"): return "VALU integer / fp32 / bit operations"
    if op.startswith("scratch_"): return "scratch (spill) access"
    if op.startswith(("global_", "flat_", "buffer_")): return "global memory access"
    if op.startswith("ds_"): return "LDS access"
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "scalar memory"
    if op.startswith("s_"): return "SALU (constants, exec masks, loop control)"
    return "other"


groups = collections.Counter()
for op, n in cnt.items():
    groups[group(op)] += n
tot = sum(cnt.values())
print("== " + dm)
print("   registers: %s VGPR (+AGPR), %s SGPR, scratch %s B per lane, LDS %s B per workgroup" % (
    g("amdhsa_next_free_vgpr"), g("amdhsa_next_free_sgpr"), g("amdhsa_private_segment_fixed_size"), g("amdhsa_group_segment_fixed_size")))
print()
print("-- static ISA histogram (%d instructions, %.0f KB of code)" % (tot, tot * 6.5 / 1024))
for k, n in groups.most_common():
    print("   %-58s %6d  %5.1f %%" % (k, n, 100. * n / tot))
print("   most frequent opcodes: " + ", ".join("%s %d" % kv for kv in cnt.most_common(14)))
if tj:
    t = json.load(open(tj))
    keys = [k for k in t if pat in k and "double" in k]
    key = next((k for k in keys if k.split("(")[0] in dm), keys[0] if keys else None)
    if key:
        v = t[key]
        w = v.get("waves", 1)
        print()
        print("-- dynamic counts of one launch on C3 (rocprofv3 PMC, %s): %d waves" % (os.path.basename(tj), w))
        rows = [("VALU total", "valu_insts"), ("  fp64 FMA", "valu_fma_f64"), ("  fp64 MUL", "valu_mul_f64"), ("  fp64 ADD", "valu_add_f64"),
                ("  fp64 transcendental", "valu_trans_f64"), ("  fp32 FMA / MUL / ADD", None), ("  fp32 transcendental", "valu_trans_f32"),
                ("  int32", "valu_int32"), ("  int64", "valu_int64"), ("  convert", "valu_cvt"), ("SALU", "salu_insts"), ("branch", "branch_insts"),
                ("global loads", "vmem_rd_insts"), ("global stores", "vmem_wr_insts"), ("LDS", "lds_insts")]
        acc = 0
        for label, k in rows:
            val = v.get(k, 0) if k else v.get("valu_fma_f32", 0) + v.get("valu_mul_f32", 0) + v.get("valu_add_f32", 0)
            if label.startswith("  "):
                acc += val
            print("   %-28s %14.4g per launch  %9.1f per wave" % (label, val, val / w))
        rest = v["valu_insts"] - acc
        print("   %-28s %14.4g per launch  %9.1f per wave   (v_mov, v_cndmask, v_cmp, bit operations: not broken out by the counters)" % ("  other VALU", rest, rest / w))
        lane = v["thread_cycles_valu"] / (v["valu_insts"] * 64.)
        f64 = v["valu_f64_insts"]
        print("   lane utilisation (SQ_THREAD_CYCLES_VALU / (SQ_INSTS_VALU x 64)): %.3f" % lane)
        print("   wave-cycles: active %.3g, waiting on instruction issue %.3g, waiting on memory / barriers %.3g  (of %.3g)" % (
            v.get("active_inst_any", 0), v.get("wait_inst_any", 0), v.get("wait_any", 0), v.get("wave_cycles", 0)))
        # (the counter, not a guess at cycles per instruction -- SQ_ACTIVE_INST_VALU counts quad-cycles in which a SIMD's
        # vector ALU is executing; every wave64 instruction takes four cycles, the fp64 transcendentals sixteen)
        if v.get("active_inst_valu") and v.get("grbm_gui_active"):
            print("   vector ALU busy: SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs) = %.3f  (%.2f cycles per instruction; %.2f ms of issue at 2.4 GHz)" % (
                v["active_inst_valu"] * 4 / (1024 * v["grbm_gui_active"] / 8), v["active_inst_valu"] * 4 / v["valu_insts"],
                v["active_inst_valu"] * 4 / 1024 / 2.4e9 * 1e3))
        print("   HBM per launch: %.2f GB read (FETCH_SIZE x 2) + %.2f GB written" % (v["read_bytes"] / 1e9, v["write_bytes"] / 1e9))
