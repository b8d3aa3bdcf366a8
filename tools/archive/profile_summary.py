"""Turn gpurun_out/<tag>_* (tools/round_profile.sh) into the committed
profiles/: <tag>_bench.json, <tag>_kernel_stats.csv, <tag>_pmc_summary.json and
profiles/pmc_latest.json (read by bench.py for roofline.traffic)."""
import csv, glob, json, os, shutil, sys, collections

tag = sys.argv[1]
out = "profiles"
os.makedirs(out, exist_ok=True)
g = "gpurun_out/%s" % tag
line = None
for l in open(g + "_bench.json"):
    if l.startswith("{"):
        line = json.loads(l)
json.dump(line, open("%s/%s_bench.json" % (out, tag), "w"), indent=1)
for f in glob.glob(g + "_stats/**/*kernel_stats.csv", recursive=True):
    shutil.copy(f, "%s/%s_kernel_stats.csv" % (out, tag))

def passes(name, kernel="SampleNeighbor"):
    rows = collections.defaultdict(list)
    for f in glob.glob(g + "_pmc_%s/**/*counter_collection.csv" % name, recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                rows[r["Counter_Name"]].append(
                    (int(r["Dispatch_Id"]), float(r["Counter_Value"]),
                     int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return rows

summary = {"note": "rocprofv3 --pmc passes (one counter set per pass, --kernel-trace only) over "
                   "`bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-check`; per-launch "
                   "means of the K1 kernel: hop1 = SampleNeighborPivotKernel<true,1,true> over the batch, "
                   "hop2 = SampleNeighborPivotDualKernel over the distinct hop-2 roots; "
                   "expand = DedupExpandKernel. FETCH_SIZE / WRITE_SIZE are reported in KiB."}
# K1 launches of a step that do work: <true, 1> = hop 1 (odd count, one sample
# per lane), <true, 2> = hop 2 over the distinct roots; the <true, 2> launches
# that last a few microseconds are the gated no-op pass (see DedupGate).
def split(v):
    # hop 1 and the distinct-root pass of hop 2 are the same kernel (one sample
    # per lane); the launcher sizes their grids from 131072 * 25 and 3276800 * 10
    # samples, which tells them apart; launches of a few microseconds are the
    # gated no-op pass (DedupGate)
    work = [x for x in v if x[2] > 30000]
    h1 = [x for x in work if x[4] < 4_000_000]
    h2 = [x for x in work if x[4] >= 4_000_000]
    return h1, h2

def passes_named(name):
    rows = collections.defaultdict(list)
    for f in glob.glob(g + "_pmc_%s/**/*counter_collection.csv" % name, recursive=True):
        for r in csv.DictReader(open(f)):
            if "SampleNeighborPivot" in r["Kernel_Name"] or "DedupExpand" in r["Kernel_Name"]:
                rows[r["Counter_Name"]].append(
                    (int(r["Dispatch_Id"]), float(r["Counter_Value"]),
                     int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"],
                     int(r["Grid_Size"])))
    return rows

for name in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum"):
    for ctr, v in passes_named(name).items():
        h1, h2 = split([x for x in v if "SampleNeighborPivot" in x[3]])
        ex = [x for x in v if "DedupExpand" in x[3] and x[2] > 30000]
        mean = lambda xs, i: sum(x[i] for x in xs) / max(len(xs), 1)
        summary[ctr] = {"hop1_mean": mean(h1, 1), "hop2_mean": mean(h2, 1),
                        "hop1_n": len(h1), "hop2_n": len(h2),
                        "hop1_mean_ns": mean(h1, 2), "hop2_mean_ns": mean(h2, 2),
                        "expand_mean": mean(ex, 1), "expand_n": len(ex),
                        "expand_mean_ns": mean(ex, 2)}
# calibration: tools/ubench_tcp "divergent dword" on the 4 GiB working set reads
# 256*8*256 lanes * 32 iters * 8 loads = 134 217 728 random 4-byte words, one per
# 64-byte sector (collisions negligible) -> at least 8.59e9 bytes by sector count
calib = passes("calib", kernel="Gather<0>")
if calib.get("FETCH_SIZE"):
    big = max(x[1] for x in calib["FETCH_SIZE"])
    lanes = 256 * 8 * 256 * 32 * 8
    summary["calibration"] = {
        "pattern": "ubench_tcp Gather<0> (lane-divergent dword), 4 GiB working set",
        "random_word_reads": lanes, "FETCH_SIZE_KiB_max_dispatch": big,
        "bytes_counted_per_random_read": big * 1024 / lanes}
json.dump(summary, open("%s/%s_pmc_summary.json" % (out, tag), "w"), indent=1)
if "FETCH_SIZE" in summary and "WRITE_SIZE" in summary:
    f, w = summary["FETCH_SIZE"], summary["WRITE_SIZE"]
    h1 = (f["hop1_mean"] + w["hop1_mean"]) * 1024
    h2 = (f["hop2_mean"] + w["hop2_mean"]) * 1024
    json.dump({"source": "profiles/%s_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, "
                         "separate passes)" % tag,
               "batch": line["config"]["roots_per_step_per_gpu"], "nodes": 100000000,
               "hbm_bytes_hop1_launch": h1, "hbm_bytes_hop2_launch": h2,
               "hbm_bytes_per_launch": (h1 + h2) / 2,
               "note": "(FETCH_SIZE + WRITE_SIZE) KiB x 1024, mean of the hop-1 and hop-2 launches of a "
                       "step (same averaging as roofline.achieved). The reads of this kernel are narrow "
                       "random reads, not the wide coalesced streams the x2 correction of "
                       "MI355X_MICROARCH.md is calibrated on; see `calibration` in the summary for what "
                       "FETCH_SIZE counts per random 4-byte read on this box."},
              open("%s/pmc_latest.json" % out, "w"), indent=1)
print(json.dumps(summary, indent=1)[:3000])
