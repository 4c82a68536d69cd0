#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (tools/profile_round.sh) into profiles/<tag>_summary.md and
profiles/traffic_latest.json (read back by bench.py for roofline.traffic)."""
import csv, glob, json, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
lines = [f"# rocprofv3 summary {tag}", "", "Command profiled: `python3 bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-extras` (default workload:",
         "2^20 patches, shader angle mode, f16x3 pooling) on one MI355X via `tools/profile_round.sh`.", ""]
bench = None
for l in open(os.path.join(src, "stats.log")):
    if l.startswith("{"):
        bench = json.loads(l)
stats = sorted(glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv")), key=os.path.getmtime, reverse=True)
kern_ms = None
# the timed kernel variant: mkd_pool<ANGLE, POOL>; bench.py also runs the exact-angle variant once as a
# secondary figure, which must not be mixed into the headline kernel's numbers
variant = "mkd_pool<%d, %d, 8>" % (0 if (not bench or bench["config"]["angle_mode"] == "shader") else 1,
                                   1 if (not bench or bench["config"]["pool_mode"] == "f16x3") else 2)
if stats:
    lines += ["## `--kernel-trace --stats` (kernel_stats.csv)", "", "| kernel | calls | avg ms | min ms | max ms | % |", "|---|---|---|---|---|---|"]
    for r in csv.DictReader(open(stats[0])):
        lines.append(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['AverageNs'])/1e6:.4f} | {float(r['MinNs'])/1e6:.4f} | {float(r['MaxNs'])/1e6:.4f} | {r['Percentage']} |")
        if variant in r["Name"]:
            kern_ms = float(r["AverageNs"]) / 1e6
    lines.append("")
def pmc(sub):
    f = sorted(glob.glob(os.path.join(src, sub, "*", "*counter_collection.csv")), key=os.path.getmtime, reverse=True)
    agg = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f[0])):
            if variant in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}
fetch, write = pmc("fetch"), pmc("write")
n = bench["config"]["patches_per_gpu"] if bench else 1 << 20
traffic = None
if fetch and write:
    # MI355X_MICROARCH.md HBM section: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half
    # of the bytes of a wide (16 B/lane) coalesced streaming read -> doubled; WRITE_SIZE is exact for
    # 16 B/lane streaming stores.
    fb = fetch["FETCH_SIZE"] * 1024 * 2
    wb = write["WRITE_SIZE"] * 1024
    traffic = fb + wb
    lines += [f"## HBM traffic of `{variant}` per launch (separate `--pmc` passes)", "",
              f"- FETCH_SIZE = {fetch['FETCH_SIZE']:.0f} KiB -> x1024 x2 (gfx950 wide-read correction) = {fb/1e9:.3f} GB",
              f"- WRITE_SIZE = {write['WRITE_SIZE']:.0f} KiB -> x1024 = {wb/1e9:.3f} GB",
              f"- total {traffic/1e9:.3f} GB per launch; algorithmic {4608*n/1e9:.3f} GB ({n} descriptors x 4608 B) "
              f"-> ratio {traffic/(4608*n):.2f}", ""]
    json.dump({"tag": tag, "patches": n, "pool": bench["config"]["pool_mode"] if bench else "f16x3",
               "hbm_bytes_per_launch": traffic, "fetch_kib": fetch["FETCH_SIZE"], "write_kib": write["WRITE_SIZE"],
               "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 wide streaming reads count half)"},
              open(os.path.join(ROOT, "profiles", "traffic_latest.json"), "w"), indent=1)
sq = {**pmc("sq"), **pmc("sq2")}
if sq:
    lines += [f"## SQ counters of `{variant}` per launch", "", "| counter | value |", "|---|---|"]
    lines += [f"| {k} | {v:.4g} |" for k, v in sorted(sq.items())]
    if "SQ_WAVE_CYCLES" in sq:
        w = sq["SQ_WAVE_CYCLES"]
        lines += ["", f"Shares of wave time: active {sq.get('SQ_ACTIVE_INST_ANY',0)/w:.0%}, waiting (s_waitcnt/barrier) {sq.get('SQ_WAIT_ANY',0)/w:.0%}, "
                  f"issue stalls {sq.get('SQ_WAIT_INST_ANY',0)/w:.0%}; VALU instructions per descriptor {sq.get('SQ_INSTS_VALU',0)/n:.0f}, "
                  f"MFMA per descriptor {sq.get('SQ_INSTS_MFMA',0)/n:.1f}."]
    lines.append("")
if bench:
    lines += ["## bench.py line of the `--stats` run", "", "```json", json.dumps(bench), "```", ""]
    if kern_ms:
        lines.append(f"Kernel time agreement: rocprofv3 average {kern_ms:.4f} ms vs bench.py HIP events {bench['roofline']['kernel_ms']:.4f} ms.")
        # the --stats average includes the warm-up launches (the first one runs cold); the trace has every dispatch
        import csv, glob
        traces = sorted(glob.glob(os.path.join(os.path.dirname(stats[-1]), "*kernel_trace.csv")), key=os.path.getmtime) if stats else []
        if traces:
            durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(traces[-1]))
                    if variant.split("(")[0] in r["Kernel_Name"]]
            k = bench["steps"]
            if len(durs) >= k:
                lines.append(f"Over the {k} timed launches alone (kernel trace, warm-up launches left out): rocprofv3 average "
                             f"{sum(durs[-k:]) / k:.4f} ms.")
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
open(os.path.join(ROOT, "profiles", f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
for f in stats:
    import shutil
    shutil.copy(f, os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
print("\n".join(lines))
