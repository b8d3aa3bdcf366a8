"""gpurun_out/<tag>_* (tools/r5_profile.sh) -> profiles/r5_<name>_{kernel_stats.csv,pmc.json} and
profiles/pmc_sharded_latest.json (what bench.py quotes as the sharded line's roofline.traffic).
  python tools/r5_collect.py <tag>"""
import glob, json, os, re, sys
tag = sys.argv[1]
os.makedirs("profiles", exist_ok=True)
skip = ("Synth", "BuildBlocks", "BuildPivot", "WbFill", "WbRec", "WbCount", "VerifyTotals", "FatFill",
        "HashI", "rocprim", "distribution_", "elementwise", "fillBuffer", "copyBuffer")
for d in glob.glob("gpurun_out/%s_*_trace" % tag):
    name = os.path.basename(d)[len(tag) + 1:-len("_trace")]
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        rows = open(f).read().splitlines()
        keep = [rows[0]] + [l for l in rows[1:] if not any(s in l for s in skip)]
        open("profiles/r5_%s_kernel_stats.csv" % name, "w").write("\n".join(keep) + "\n")
        print("profiles/r5_%s_kernel_stats.csv" % name, len(keep) - 1, "kernels")
for f in glob.glob("gpurun_out/%s_*_pmc.json" % tag):
    name = os.path.basename(f)[len(tag) + 1:-len("_pmc.json")]
    try:
        doc = json.load(open(f))
    except Exception as e:
        print("skip", f, e)
        continue
    for k, c in doc.items():
        if "TCC_EA0_RDREQ_sum" in c:
            c["read_bytes_128B_lines"] = c["TCC_EA0_RDREQ_sum"] * 128.0
        if "WRITE_SIZE" in c:
            c["write_bytes"] = c["WRITE_SIZE"] * 1024.0
        if "SQ_WAIT_ANY" in c and c.get("SQ_WAVE_CYCLES"):
            c["waiting_share_of_wave_cycles"] = round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3)
    json.dump(doc, open("profiles/r5_%s_pmc.json" % name, "w"), indent=1)
    print("profiles/r5_%s_pmc.json" % name, list(doc)[:4])
    if name == "sharded_step":
        for k, c in doc.items():
            if "SampleNeighborPivotKernel" in k and "TCC_EA0_RDREQ_sum" in c:
                latest = {"kernel": k, "batch": 131072, "nodes": 100000000,
                          "read_requests": c["TCC_EA0_RDREQ_sum"], "read_bytes": c["read_bytes_128B_lines"],
                          "write_bytes": c.get("write_bytes"),
                          "hbm_bytes_per_launch": c["read_bytes_128B_lines"] + (c.get("write_bytes") or 0.0),
                          "mean_us_under_pmc": c.get("mean_us_under_pmc"), "dispatches": c.get("dispatches"),
                          "source": "profiles/r5_sharded_step_pmc.json: mean over the two hops' launches of one "
                                    "rank's sharded step (tools/r5_one.py sharded_step): TCC_EA0_RDREQ x 128 B + "
                                    "WRITE_SIZE KiB x 1024, separate rocprofv3 --pmc passes"}
                json.dump(latest, open("profiles/pmc_sharded_latest.json", "w"), indent=1)
                print(json.dumps(latest))
