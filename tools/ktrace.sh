#!/bin/bash
# kernel-trace statistics of a short bench run:  gpurun -- 'bash tools/ktrace.sh <tag> [bench args]'  -> the cond kernels' rows
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/kt_$tag; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 20 --warmup 20 --no-cpu-baseline --no-strict-leg --no-toms-leg --no-host-leg --no-extra-legs --stage-steps 0 "$@" > $out/bench.json 2> $out/trace.log
f=$(ls $out/trace/*/*kernel_stats.csv | head -1)
python3 - $f $out/bench.json <<'PY'
import csv, sys, json
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    print("%-70s calls %5s avg %9.1f us  %5.1f %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
try:
    b = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); print("ms_per_step", b["ms_per_step"])
except Exception as e:
    print("bench line unreadable", e)
PY
