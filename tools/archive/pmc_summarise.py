"""Summarise rocprofv3 --pmc csv output: per kernel (substring filter) and
counter, mean/min/max per dispatch, optionally split by grid size."""
import csv, glob, json, sys, collections
def main():
    pat = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else "SampleNeighbor"
    out = collections.defaultdict(list)
    for f in glob.glob(pat, recursive=True):
        for row in csv.DictReader(open(f)):
            if filt not in row["Kernel_Name"]:
                continue
            d = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
            key = (row["Kernel_Name"].split("(")[0][:60],
                   "long" if d > 300000 else "short", row["Counter_Name"])
            out[key].append((float(row["Counter_Value"]), d))
    res = {}
    for (k, g, c), v in sorted(out.items()):
        vals = [x[0] for x in v]; dur = [x[1] for x in v]
        res["%s|%s|%s" % (k, g, c)] = {"n": len(v), "mean": sum(vals) / len(vals),
                                           "min": min(vals), "max": max(vals),
                                           "mean_ns": sum(dur) / len(dur)}
    print(json.dumps(res, indent=1))
main()
