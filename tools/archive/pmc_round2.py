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
# true HBM-side bytes per launch: every read request moves one 128-byte line and is
# tallied at 64 B (calibration above: a 16-byte read that needs BOTH sectors of a
# line costs one request / 64 counted bytes, like a 4-byte read), streaming writes
# are counted exactly
for short, e in doc["kernels"].items():
    f = e.get("FETCH_SIZE", {}).get("mean")
    w = e.get("WRITE_SIZE", {}).get("mean")
    if f is not None and w is not None:
        e["hbm_bytes_per_launch"] = 2.0 * f * 1024 + w * 1024
        e["read_bytes_per_launch"] = 2.0 * f * 1024
        e["write_bytes_per_launch"] = w * 1024
hop1 = next((e for k, e in doc["kernels"].items() if k.startswith("SampleNeighborPivotKernel") and "hbm_bytes_per_launch" in e), None)
hop2 = next((e for k, e in doc["kernels"].items() if k.startswith("SampleNeighborPivotDualKernel") and "hbm_bytes_per_launch" in e), None)
if hop1 and hop2:
    doc["pmc_latest"] = {
        "source": "profiles/%s_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes; "
                  "tools/round2_profile.sh)" % tag,
        "batch": 131072, "nodes": 100000000,
        "hbm_bytes_hop1_launch": hop1["hbm_bytes_per_launch"],
        "hbm_bytes_hop2_launch": hop2["hbm_bytes_per_launch"],
        "hbm_bytes_per_launch": (hop1["hbm_bytes_per_launch"] + hop2["hbm_bytes_per_launch"]) / 2,
        "note": "(2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 per launch: the x2 of MI355X_MICROARCH.md applies to "
                "these kernels too - calibration in the same file: one L2 read request = one 128-byte line, "
                "tallied at 64 B, whatever the width of the access that missed"}
print(json.dumps(doc, indent=1))
