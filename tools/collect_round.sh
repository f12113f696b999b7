#!/bin/bash
# Copies what a round's GPU runs left under gpurun_out/ into profiles/ under the round's tag (run in the build container, after
#   gpurun -- 'bash tools/profile_round.sh <tag>; bash tools/power_probe.sh <tag>; bash tools/pmc_steady.sh <tag>_plain; ...'):
#   bash tools/collect_round.sh <tag>
tag=$1
P=profiles
src=gpurun_out/prof_$tag
cp $src/summary.txt $P/${tag}_profile_summary.txt
cp $src/traffic.json $P/${tag}_traffic.json
cp $src/bench_under_trace.json $P/${tag}_bench_under_trace.json
cp $(ls $src/trace/*/*kernel_stats.csv | head -1) $P/${tag}_kernel_stats.csv
[ -f gpurun_out/power_$tag.txt ] && cp gpurun_out/power_$tag.txt $P/${tag}_power.txt
for v in plain fold toms toms_nofold coal_stress probe wq budget; do
  [ -f gpurun_out/pmcs_${tag}_$v.txt ] && grep -v "^$" gpurun_out/pmcs_${tag}_$v.txt > $P/${tag}_pmc_$v.txt
done
[ -f gpurun_out/bench_$tag.json ] && cp gpurun_out/bench_$tag.json $P/${tag}_bench_default.json
[ -f gpurun_out/soak_$tag.json ] && cp gpurun_out/soak_$tag.json $P/${tag}_soak.json
python3 - $tag <<'PY'
import glob, json, os, sys
tag = sys.argv[1]
out = {}
for f in sorted(glob.glob("gpurun_out/*_%s*.json" % tag)):
    name = os.path.basename(f)[:-5].replace("_" + tag, "")
    if name in ("bench", "soak"):
        continue
    try:
        out[name] = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception:
        pass
json.dump(out, open("profiles/%s_measurements.json" % tag, "w"), indent=1)
print("measurements:", sorted(out))
PY
python3 tools/isa_histogram.py "k_cond_lean<double, 15, true, 0, -1>" $P/${tag}_traffic.json > $P/${tag}_k_cond_lean_instruction_mix.txt 2>/dev/null
python3 tools/isa_histogram.py "k_cond_lean_fold<double, true, 2>" $P/${tag}_traffic.json > $P/${tag}_k_cond_toms748_fold_instruction_mix.txt 2>/dev/null
ls -la $P/${tag}_*
