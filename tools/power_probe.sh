#!/bin/bash
# Package power and shader clock while bench.py's default box steps (rocm-smi sampled every 0.15 s beside a 1500-step run):
#   gpurun -- 'bash tools/power_probe.sh <tag> [bench args]'   ->  gpurun_out/power_<tag>.txt  (copy to profiles/<tag>_power.txt)
# What it shows for the headline: the package sits at its power cap through the whole step and the shader clock below its 2.4 GHz --
# the chip prices a launch in lanes that compute, not in instructions that issue (the condensation kernel).
tag=${1:-x}; shift
out=gpurun_out/power_$tag; mkdir -p $out
B="python3 bench.py --no-cpu-baseline --no-strict-leg --no-toms-leg --no-host-leg --no-extra-legs --steps 1500 --stage-steps 0 $*"
( for i in $(seq 1 400); do /opt/rocm/bin/rocm-smi --showpower --showclocks --showtemp --json 2>/dev/null | head -c 4000; echo; sleep 0.15; done ) > $out/smi.log 2>&1 &
SMI=$!
$B > $out/bench.json 2> $out/bench.err
kill $SMI 2>/dev/null
/opt/rocm/bin/rocm-smi --showmaxpower > $out/cap.txt 2>&1
python3 - $out "$B" > gpurun_out/power_$tag.txt <<'PY'
import json, re, sys
out, cmd = sys.argv[1], sys.argv[2]
rows = []
for l in open(out + "/smi.log"):
    l = l.strip()
    if not l.startswith("{"):
        continue
    try:
        c = json.loads(l).get("card0", {})
    except Exception:
        continue
    g = lambda pat: next((v for k, v in c.items() if re.search(pat, k)), None)
    num = lambda v: float(re.sub(r"[^0-9.]", "", v)) if v else float("nan")
    rows.append((num(g("Package Power")), num(g("sclk clock speed")), num(g("mclk clock speed")), num(g("junction"))))
cap = re.search(r"Max Graphics Package Power \(W\): ([0-9.]+)", open(out + "/cap.txt").read())
print("# rocm-smi beside:", cmd)
print("# power cap of the package: %s W" % (cap.group(1) if cap else "?"))
busy = [r for r in rows if r[0] > 600]
print("# samples %d, of which under load (> 600 W) %d" % (len(rows), len(busy)))
if busy:
    import statistics as st
    print("# under load: package power mean %.0f W (min %.0f, max %.0f); sclk mean %.0f MHz (min %.0f, max %.0f); mclk %.0f MHz; junction up to %.0f C"
          % (st.mean(r[0] for r in busy), min(r[0] for r in busy), max(r[0] for r in busy), st.mean(r[1] for r in busy), min(r[1] for r in busy),
             max(r[1] for r in busy), busy[0][2], max(r[3] for r in busy)))
try:
    r = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1])
    print("# the run: %.3f ms per step, condensation launch %.3f ms" % (r["ms_per_step"], r["roofline"]["avg_launch_ms"]))
except Exception as e:
    print("# bench line unreadable:", e)
print("# t[s]  package_W  sclk_MHz  mclk_MHz  junction_C")
for i, r in enumerate(rows):
    print("%5.2f  %7.0f  %7.0f  %7.0f  %5.0f" % (i * 0.15, *r))
PY
cat gpurun_out/power_$tag.txt | head -12
