"""one step's kernels on the device's clock, from a kernel trace:
  rocprofv3 --kernel-trace --output-format csv -d out -- python3 bench.py ... ;  python3 tools/step_timeline.py out [kernel that ends a step] [which step from the end]
prints start, end, duration (us; 0 = the end of the previous step's last kernel), the queue, the kernel; gaps are the host's or the dependencies'"""
import csv
import glob
import sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
last = sys.argv[2] if len(sys.argv) > 2 else "k_move"
back = int(sys.argv[3]) if len(sys.argv) > 3 else 2
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if last in r["Kernel_Name"]]
a, b = idx[-back - 1], idx[-back]
t0 = int(rows[a]["End_Timestamp"])
busy = 0
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    print("%9.1f %9.1f %8.1f  q%-2s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:90]))
print("step %.1f us, kernels %.1f us (sum over queues), %d launches" % ((int(rows[b]["End_Timestamp"]) - t0) / 1e3, busy / 1e3, b - a))
