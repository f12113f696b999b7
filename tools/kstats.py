"""per-kernel launch statistics of a rocprofv3 --kernel-trace csv: calls, average / min / max duration of the launches behind the warm-up
    python3 tools/kstats.py <dir> [skip-first-N-launches]"""
import collections, csv, glob, sys
d = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
per = collections.defaultdict(list)
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        per[r["Kernel_Name"].replace("void ", "").split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v[skip:]) for v in per.values())
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1][skip:])):
    w = v[skip:] if len(v) > skip else v
    print("%-70s calls %6d  avg_us %9.2f  min %9.2f  max %9.2f  share %5.1f%%" % (k[:70], len(w), sum(w) / len(w), min(w), max(w), 100 * sum(w) / tot))
