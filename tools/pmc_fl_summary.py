"""Per-dispatch means of the rocprofv3 counter passes of tools/pmc_fl.sh, per kernel."""
import collections, csv, glob, json, os, re, sys
out_dir, prefix = sys.argv[1], sys.argv[2]
res = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(out_dir, prefix + "*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        short = re.sub(r"^void ", "", row["Kernel_Name"]).split("(")[0].replace("euler_gpu::", "")
        res[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for f in glob.glob(os.path.join(out_dir, prefix + "*", "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        short = re.sub(r"^void ", "", row["Kernel_Name"]).split("(")[0].replace("euler_gpu::", "")
        dur[short].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
doc = {}
for k, cs in sorted(res.items()):
    if not any(w in k for w in ("Sample", "Dedup", "Expand", "Walk", "N2v", "Node2Vec", "Flow", "Front", "Segment", "Gather", "Cw")):
        continue
    # skip the first dispatch of every kernel (cold caches, lazy allocations)
    doc[k] = {n: round(sum(v[1:]) / max(1, len(v) - 1), 1) if len(v) > 1 else v[0] for n, v in cs.items()}
    doc[k]["dispatches"] = max(len(v) for v in cs.values())
    if dur.get(k):
        d = dur[k][1:] if len(dur[k]) > 1 else dur[k]
        doc[k]["mean_us_under_pmc"] = round(sum(d) / len(d) / 1e3, 2)
print(json.dumps(doc, indent=1))
