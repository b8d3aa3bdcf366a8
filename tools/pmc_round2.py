"""Join the PMC passes of tools/round2_profile.sh: per kernel of the metric step
(hop-1 K1, hop-2 K1 over the distinct roots, numbering, resolve, expansion) the mean
counter values per launch, and the calibration of FETCH_SIZE / WRITE_SIZE on the
known request shapes of tools/ubench_fetch.  Prints one JSON document."""
import collections, csv, glob, json, os, re, sys

out_dir, tag = sys.argv[1], sys.argv[2]


def counters(prefix):
    """{kernel short name: {counter: [values per dispatch]}} and durations"""
    res = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in glob.glob(os.path.join(out_dir, prefix + "*", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"]
            short = re.sub(r"^void ", "", name).split("(")[0].replace("euler_gpu::", "")
            res[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
            if "Start_Timestamp" in row and row.get("End_Timestamp"):
                dur[short].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    return res, dur


def mean(v):
    return sum(v) / len(v) if v else None


bench, _ = counters(tag + "_pmc_")
cal, _ = counters(tag + "_cal_")
asked = {}
p = os.path.join(out_dir, tag + "_cal_asked.txt")
if os.path.exists(p):
    for line in open(p):
        m = re.match(r"(\w+)\s+([\d.]+) ms\s+accesses (\d+)\s+asked_bytes (\d+)\s+sectors64 (\d+)\s+lines128 (\d+)", line)
        if m:
            asked[m.group(1)] = {"ms": float(m.group(2)), "accesses": int(m.group(3)),
                                 "asked_bytes": int(m.group(4)), "sectors64": int(m.group(5)),
                                 "lines128": int(m.group(6))}
doc = {"calibration": {}, "kernels": {}}
for k, a in asked.items():
    c = {n: mean(v) for n, v in cal.get(k, {}).items()}
    e = dict(a)
    e["counters"] = c
    # FETCH_SIZE / WRITE_SIZE are reported in KiB
    if c.get("FETCH_SIZE") is not None:
        e["FETCH_SIZE_bytes"] = c["FETCH_SIZE"] * 1024
        e["fetch_bytes_per_access"] = c["FETCH_SIZE"] * 1024 / a["accesses"]
        e["fetch_over_sectors64_bytes"] = c["FETCH_SIZE"] * 1024 / (a["sectors64"] * 64)
        e["fetch_over_lines128_bytes"] = c["FETCH_SIZE"] * 1024 / (a["lines128"] * 128)
    if c.get("WRITE_SIZE") is not None and k.startswith("wr"):
        e["WRITE_SIZE_bytes"] = c["WRITE_SIZE"] * 1024
        e["write_bytes_per_access"] = c["WRITE_SIZE"] * 1024 / a["accesses"]
    doc["calibration"][k] = e
want = ("SampleNeighborPivotKernel", "SampleNeighborPivotDualKernel", "SampleNeighborRowKernel",
        "SampleNeighborSlowKernel", "DedupNumberKernel", "DedupResolveNumberedKernel",
        "DedupExpandLeanKernel", "DedupExpandKernel", "DedupMarkKernel")
for short, cs in sorted(bench.items()):
    if not any(w in short for w in want):
        continue
    doc["kernels"][short] = {n: {"mean": mean(v), "n": len(v), "min": min(v), "max": max(v)}
                             for n, v in cs.items()}
print(json.dumps(doc, indent=1))
