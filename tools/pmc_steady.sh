#!/bin/bash
# PMC counters of the STEADY-STATE launches of the default bench box (the round profile's PMC passes take steps 2-3, when half of the
# droplets are still adjusting to the box's humidity):  gpurun -- 'bash tools/pmc_steady.sh <tag> [bench args]'
tag=${1:-x}; shift; extra="$@"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmcs_$tag
mkdir -p $out
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $out/pmc_$i -- python3 bench.py --steps 2 --warmup 18 --no-cpu-baseline --no-strict-leg --no-toms-leg --no-host-leg --no-extra-legs --no-stage-timers $extra > $out/pmc_$i.log 2>&1
done
python3 - $out <<'PY'
import collections, csv, glob, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].replace("void ", "").split("(")[0][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    if "k_cond" not in k and "k_coal" not in k and "k_move" not in k and "k_cellrank" not in k:
        continue
    g = {c: sum(x[-2:]) / len(x[-2:]) for c, x in v.items()}
    w = g.get("SQ_WAVES", 1)
    print("%-44s waves %.4g  VALU/wave %.1f (fma %.1f mul %.1f add %.1f trans %.1f f64)  SALU/wave %.1f  lane_util %.3f  VALU-busy %.3f  cycles(GRBM/8) %.4g  wait_any %.2f active_any %.2f" % (
        k, w, g["SQ_INSTS_VALU"] / w, g.get("SQ_INSTS_VALU_FMA_F64", 0) / w, g.get("SQ_INSTS_VALU_MUL_F64", 0) / w, g.get("SQ_INSTS_VALU_ADD_F64", 0) / w,
        g.get("SQ_INSTS_VALU_TRANS_F64", 0) / w, g.get("SQ_INSTS_SALU", 0) / w, g["SQ_THREAD_CYCLES_VALU"] / (g["SQ_INSTS_VALU"] * 64),
        g["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * g["GRBM_GUI_ACTIVE"] / 8), g["GRBM_GUI_ACTIVE"] / 8, g["SQ_WAIT_ANY"] / g["SQ_WAVE_CYCLES"], g["SQ_ACTIVE_INST_ANY"] / g["SQ_WAVE_CYCLES"]))
PY
