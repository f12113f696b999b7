"""reads the csv files of tools/calibrate_traffic.sh: per access pattern the counter's bytes against the bytes the kernel asked for"""
import collections, csv, glob, json, sys
out = sys.argv[1]
TABLE = 2 << 30
want = {"stream16": TABLE, "stream8": TABLE, "stream4": TABLE, "gather8_far": TABLE, "gather8_near": TABLE, "gather8_mix": TABLE}

def counters(d):
    res = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(out + "/" + d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            res[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return res

dur = collections.defaultdict(list)
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
fetch, req = counters("pmc_fetch"), counters("pmc_req")
res = {}
for k, asked in want.items():
    e = {"bytes_requested_by_lanes": asked}
    if k in fetch and "FETCH_SIZE" in fetch[k]:
        v = fetch[k]["FETCH_SIZE"][-1] * 1024
        e["FETCH_SIZE_bytes"] = v
        e["requested_over_FETCH_SIZE"] = asked / v
    if k in req:
        rd, r32, bub = (req[k].get(n, [0])[-1] for n in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_BUBBLE_sum"))
        e.update({"TCC_EA0_RDREQ": rd, "TCC_EA0_RDREQ_32B": r32, "TCC_BUBBLE": bub, "requested_bytes_per_RDREQ": asked / rd if rd else None})
    if k in dur:
        e["us"] = dur[k][-1]
        e["requested_GBs"] = asked / dur[k][-1] / 1e3
    res[k] = e
json.dump(res, sys.stdout, indent=1)
