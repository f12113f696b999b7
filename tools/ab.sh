#!/bin/bash
# A/B of two builds of the library on ONE box (box-to-box spread is ~6 %): the in-tree liblcx_hip.so against tools/_ab/<name>.so
#   gpurun -- 'bash tools/ab.sh <tag> <other.so> [bench args]'  ->  gpurun_out/ab_<tag>/{new,old}.json
tag=$1; other=$2; shift 2
out=gpurun_out/ab_$tag; mkdir -p $out
B="python3 bench.py --no-cpu-baseline --no-strict-leg --no-toms-leg --no-host-leg --steps 100 $*"
so=libcloudphxx_amd/csrc/liblcx_hip.so
$B > $out/new.json 2> $out/new.err
cp $so $out/../_new.so; cp tools/_ab/$other $so
$B > $out/old.json 2> $out/old.err
cp $out/../_new.so $so; rm -f $out/../_new.so
$B > $out/new2.json 2> $out/new2.err
python3 - $out <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    st = r.get("stage_ms_per_step") or {}
    print(os.path.basename(f), "ms_per_step", r.get("ms_per_step"), "cond_kernel_ms", (r.get("roofline") or {}).get("avg_launch_ms"), "frac", (r.get("roofline") or {}).get("frac"),
          {k: round(v, 3) for k, v in st.items()} if isinstance(st, dict) else "")
PY
