"""Per-kernel steady-state averages and the idle share of a rocprofv3 kernel trace (kernel_trace.csv):
   python3 tools/trace_gaps.py <dir> [skip_fraction]
Looks at the launches behind the first `skip_fraction` of the trace (default 0.5: warm-up and init excluded)."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").split("(")[0]))
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
cut = t0 + (t1 - t0) * skip
sel = [r for r in rows if r[0] >= cut]
span = sel[-1][1] - sel[0][0]
busy, end = 0, sel[0][0]
gaps = []
for s, e, _ in sel:
    if s > end:
        gaps.append(s - end)
        busy += e - s
    else:
        busy += max(0, e - max(s, end))
    end = max(end, e)
per = collections.defaultdict(list)
for s, e, k in sel:
    per[k].append(e - s)
print("window %.1f ms, %d launches, GPU busy %.1f %%, idle %.1f %% in %d gaps (median gap %.1f us, gaps > 20 us: %d, their sum %.2f ms)" % (
    span / 1e6, len(sel), 100. * busy / span, 100. * (span - busy) / span, len(gaps), sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0,
    sum(1 for g in gaps if g > 20000), sum(g for g in gaps if g > 20000) / 1e6))
print("%-58s %7s %10s %10s" % ("kernel", "calls", "avg_us", "total_ms"))
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:40]:
    print("%-58s %7d %10.2f %10.3f" % (k[:58], len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6))
