"""Condenses a tools/profile_round.sh output directory into the text summary committed under profiles/."""
import collections
import csv
import glob
import json
import sys

out = sys.argv[1]
print("== rocprofv3 --kernel-trace --stats : python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline")
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    print("%-60s %8s %12s %12s %7s" % ("kernel", "calls", "total_ms", "avg_us", "%"))
    for r in rows[:30]:
        print("%-60s %8s %12.3f %12.2f %7.2f" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                              float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
print()
print("== same trace, TIMED REGION only (the last 10 launches of each per-step kernel; the first warm-up launch of k_cond")
print("   starts from the un-equilibrated initial state and takes ~3x longer, which skews the all-calls average above)")
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        per[r["Kernel_Name"].replace("void ", "").split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, d in sorted(per.items(), key=lambda kv: -sum(kv[1][-10:])):
        if len(d) >= 20 and k.startswith("lcx::") and sum(d[-10:]) > 1000:
            print("%-60s avg_us(last 10) %10.2f   min %10.2f  max %10.2f" % (k[:60], sum(d[-10:]) / 10, min(d[-10:]), max(d[-10:])))
try:
    print("bench line under trace:", open(out + "/bench_under_trace.json").read().strip()[:400])
except OSError:
    pass
print()
print("== PMC passes (python3 bench.py --steps 2 --warmup 18): per-launch averages of the last two launches = steps 19-20 of the box, its steady state")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].replace("void ", "").split("(")[0][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
traffic = {}
for k, v in sorted(agg.items()):
    if not k.startswith("lcx::") or k.startswith("lcx::k_init") or k.startswith("lcx::k_scan") or "strided" in k:
        continue
    n = len(next(iter(v.values())))
    line = {c: "%.4g" % (sum(x[-2:]) / len(x[-2:])) for c, x in sorted(v.items())}      # last two launches = timed steps
    print("%-48s n=%-4d %s" % (k, n, line))
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        # MI355X_MICROARCH.md: counters are in KiB.  FETCH_SIZE tallies every read request of the L2 at 64 B.  Calibration on known-byte
        # patterns (tools/traffic_calib.hip, profiles/r03_traffic_calibration.json): coalesced reads of 4, 8 and 16 B per lane go out as
        # 128-B requests (the counter shows exactly half the bytes); a sparse 8-byte gather goes out as one request per element whose
        # size the counters do not tell (64-B sectors by the time it takes: 128 B each would be 6.9 TB/s of random reads).  A kernel
        # that mixes both -- k_cond's index stream + attribute gathers -- lies between: read_bytes_low = every request a 64-B one,
        # read_bytes = every request a 128-B one (what rounds 1-2 reported).  WRITE_SIZE needs no correction.
        raw = sum(v["FETCH_SIZE"][-2:]) / len(v["FETCH_SIZE"][-2:]) * 1024
        rd = raw * 2
        wr = sum(v["WRITE_SIZE"][-2:]) / len(v["WRITE_SIZE"][-2:]) * 1024
        traffic[k] = {"read_bytes": rd, "read_bytes_low": raw, "write_bytes": wr, "hbm_bytes": rd + wr, "hbm_bytes_low": raw + wr}
        if "SQ_INSTS_VALU" in v:      # wave-level vector instructions per launch (fp64 vector utilisation of the cond kernel)
            traffic[k]["valu_insts"] = sum(v["SQ_INSTS_VALU"][-2:]) / len(v["SQ_INSTS_VALU"][-2:])
            traffic[k]["valu_f64_insts"] = sum(sum(v[c][-2:]) / len(v[c][-2:]) for c in
                                               ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64") if c in v)
            for c, nm in (("SQ_INSTS_VALU_FMA_F64", "valu_fma_f64"), ("SQ_INSTS_VALU_MUL_F64", "valu_mul_f64"), ("SQ_INSTS_VALU_ADD_F64", "valu_add_f64"),
                          ("SQ_INSTS_VALU_TRANS_F64", "valu_trans_f64"), ("SQ_THREAD_CYCLES_VALU", "thread_cycles_valu"), ("SQ_INSTS_SALU", "salu_insts"),
                          ("SQ_INSTS_VMEM_RD", "vmem_rd_insts"), ("SQ_INSTS_VMEM_WR", "vmem_wr_insts"), ("SQ_INSTS_LDS", "lds_insts"),
                          ("SQ_INSTS_BRANCH", "branch_insts"), ("SQ_INSTS_VALU_INT32", "valu_int32"), ("SQ_INSTS_VALU_INT64", "valu_int64"),
                          ("SQ_INSTS_VALU_CVT", "valu_cvt"), ("SQ_INSTS_VALU_TRANS_F32", "valu_trans_f32"), ("SQ_INSTS_VALU_FMA_F32", "valu_fma_f32"),
                          ("SQ_INSTS_VALU_MUL_F32", "valu_mul_f32"), ("SQ_INSTS_VALU_ADD_F32", "valu_add_f32"), ("SQ_WAVES", "waves"),
                          ("SQ_WAVE_CYCLES", "wave_cycles"), ("SQ_WAIT_ANY", "wait_any"), ("SQ_WAIT_INST_ANY", "wait_inst_any"),
                          ("SQ_ACTIVE_INST_ANY", "active_inst_any"), ("SQ_ACTIVE_INST_VALU", "active_inst_valu"), ("GRBM_GUI_ACTIVE", "grbm_gui_active")):
                if c in v:
                    traffic[k][nm] = sum(v[c][-2:]) / len(v[c][-2:])
print()
print("== HBM traffic per launch (FETCH_SIZE*2 KiB + WRITE_SIZE KiB)")
for k, t in traffic.items():
    print("%-48s read %.3f GB  write %.3f GB  total %.3f GB" % (k, t["read_bytes"] / 1e9, t["write_bytes"] / 1e9, t["hbm_bytes"] / 1e9))
traffic["_meta"] = {"launches": "steady: the last two launches of every PMC pass, behind 18 warm-up steps (tools/profile_round.sh)"}
json.dump(traffic, open(out + "/traffic.json", "w"), indent=1)
