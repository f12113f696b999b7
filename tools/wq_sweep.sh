#!/bin/bash
# variants of the lean condensation kernel on ONE box (box-to-box spread is ~6 %):
#   gpurun -- 'bash tools/wq_sweep.sh <tag> [variants...]'  ->  gpurun_out/wq_<tag>/*.json + a table
# a variant = comma-separated opts_init.dbg_flags names / budget=N (bench.py --dbg), or `default`
tag=$1; shift
out=gpurun_out/wq_$tag; mkdir -p $out
B="python3 bench.py --no-cpu-baseline --no-strict-leg --no-toms-leg --no-host-leg --no-extra-legs --steps ${STEPS:-60} --warmup ${WARMUP:-20} $BENCH_ARGS"
vars="$@"
[ -z "$vars" ] && vars="default COND_BUDGET COND_BUDGET,budget=1 COND_BUDGET,budget=3 COND_WQ default"
i=0
for v in $vars; do
  i=$((i+1))
  if [ "$v" = default ]; then $B > $out/${i}_$v.json 2> $out/${i}_$v.err; else $B --dbg $v > $out/${i}_$v.json 2> $out/${i}_$v.err; fi
done
python3 - $out <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json"), key=lambda p: int(os.path.basename(p).split("_")[0])):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    st = r.get("stage_ms_per_step") or {}
    rl = r.get("roofline") or {}
    print("%-36s ms_per_step %.3f  cond_kernel_ms %.3f  frac %.4f  cond stage %.3f  listed %s" % (os.path.basename(f), r.get("ms_per_step"), rl.get("avg_launch_ms", 0), rl.get("frac", 0), st.get("cond", 0), rl.get("listed_share")))
PY
