#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (tools/profile_round.sh) into profiles/<tag>_summary.md, <tag>_kernel_stats.csv,
<tag>_keypoint_mode.md and the two traffic files bench.py quotes (profiles/traffic_latest.json,
profiles/traffic_keypoint_mode.json) -- everything from ONE run of the script, stamped with the sha256 of the kernel
sources it measured (bench.py prints traffic: null when the sources have changed since)."""
import csv, glob, hashlib, json, os, shutil, subprocess, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"


def stamp(files):
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(ROOT, "local-features_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


try:
    head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"], text=True).strip()
except Exception:
    head = None
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
lines = [f"# rocprofv3 summary {tag}", "", "Command profiled: `python3 bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-extras --no-match` (default workload:",
         "2^20 patches, shader angle mode, f16x3 pooling) on one MI355X via `tools/profile_round.sh`.", ""]
bench = None
for l in open(os.path.join(src, "stats.log")):
    if l.startswith("{"):
        bench = json.loads(l)
stats = sorted(glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv")), key=os.path.getmtime, reverse=True)
kern_ms = None
# the timed kernel variant: mkd_pool<ANGLE, POOL>; bench.py also runs the exact-angle variant once as a
# secondary figure, which must not be mixed into the headline kernel's numbers
variant = "mkd_pool<%d, %d, 8, 0>" % (0 if (not bench or bench["config"]["angle_mode"] == "shader") else 1,
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
               "angle": bench["config"]["angle_mode"] if bench else "shader",
               "source_sha256": stamp(("mkd_describe.hip", "mkd_sample.h", "mkd_device.h")), "git_head_at_summary": head,
               "hbm_bytes_per_launch": traffic, "fetch_kib": fetch["FETCH_SIZE"], "write_kib": write["WRITE_SIZE"],
               "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 wide streaming reads count half)"},
              open(os.path.join(ROOT, "profiles", "traffic_latest.json"), "w"), indent=1)
sq = {**pmc("sq"), **pmc("sq2")}
# the clock the chip held during the profiled launches (GRBM_GUI_ACTIVE sums the 8 XCDs: MI355X_MICROARCH.md, DVFS give-back)
# and the share of SIMD-cycles in which the matrix pipe was busy (SQ_VALU_MFMA_BUSY_CYCLES over 1024 SIMDs x kernel cycles)
clock_mhz = mfma_busy = None
if sq.get("GRBM_GUI_ACTIVE") and kern_ms:
    cycles = sq["GRBM_GUI_ACTIVE"] / 8.0
    clock_mhz = cycles / (kern_ms * 1e3)
    if sq.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        mfma_busy = sq["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cycles)
projection = None
pj = os.path.join(ROOT, "gpurun_out", f"{tag}_projection.json")
if os.path.exists(pj):
    projection = json.load(open(pj))
tl = os.path.join(ROOT, "profiles", "traffic_latest.json")
if traffic is not None and os.path.exists(tl):
    tj = json.load(open(tl))
    tj.update({"shader_clock_mhz_profiled": clock_mhz, "mfma_busy_frac": mfma_busy, "projection": projection})
    json.dump(tj, open(tl, "w"), indent=1)
if sq:
    lines += [f"## SQ counters of `{variant}` per launch", "", "| counter | value |", "|---|---|"]
    lines += [f"| {k} | {v:.4g} |" for k, v in sorted(sq.items())]
    if "SQ_WAVE_CYCLES" in sq:
        w = sq["SQ_WAVE_CYCLES"]
        lines += ["", f"Shares of wave time: active {sq.get('SQ_ACTIVE_INST_ANY',0)/w:.0%}, waiting (s_waitcnt/barrier) {sq.get('SQ_WAIT_ANY',0)/w:.0%}, "
                  f"issue stalls {sq.get('SQ_WAIT_INST_ANY',0)/w:.0%}; VALU instructions per descriptor {sq.get('SQ_INSTS_VALU',0)/n:.0f}, "
                  f"MFMA per descriptor {sq.get('SQ_INSTS_MFMA',0)/n:.1f}."]
    if clock_mhz:
        lines += ["", f"Shader clock during the profiled launches: GRBM_GUI_ACTIVE / 8 / kernel time = {clock_mhz:.0f} MHz"
                  + (f"; matrix pipe busy {mfma_busy:.1%} of all SIMD-cycles (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles))."
                     if mfma_busy else ".")]
    if projection:
        lines += ["", "### The whitening projection on its own (`tools/phase_timing.py`, a `-DLF_PHASE_TIMING` build)", "",
                  f"- {projection['epilogue_cycles_per_wave_batch']:.0f} shader cycles per wave and batch of 16 descriptors "
                  f"({projection['stage_share_of_kernel']:.1%} of the kernel), 264 `v_mfma_f32_16x16x32_f16` in it",
                  f"- matrix pipe busy {projection['matrix_pipe_busy_frac_in_stage']:.1%} of the stage (two waves per SIMD x 264 x 16 cycles), "
                  f"{projection['f16_pflops_in_stage']:.3f} PFLOP/s of f16 MFMA chip-wide while in it = "
                  f"{projection['f16_pflops_in_stage'] / 2.5:.1%} of the 2.5 PFLOP/s peak"]
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
if stats:
    shutil.copy(stats[0], os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))     # the newest run, and only that
print("\n".join(lines))

# ---- keypoint mode: tools/prof_keypoints.py (configs[2] as one batch: 256 frames 640x480 x 2000 keypoints) ----------
def newest(sub, pattern):
    f = sorted(glob.glob(os.path.join(src, sub, "**", pattern), recursive=True), key=os.path.getmtime, reverse=True)
    return f[0] if f else None


kp_stats = newest("kp_stats", "*kernel_stats.csv")
if kp_stats:
    n_kp, frames, w, h = 512000, 256, 640, 480
    calls = 3                                                     # prof_keypoints.py runs the batch three times
    rows = [r for r in csv.DictReader(open(kp_stats)) if "lfmkd" in r["Name"]]
    out = [f"# rocprofv3 summary {tag}: keypoint mode", "",
           "Command profiled: `python3 tools/prof_keypoints.py` -- BASELINE configs[2] as one batch (256 frames 640x480, 2000 given",
           "keypoints each = 512 000 descriptors; set_images + describe_keypoints_frames, three times) via `tools/profile_round.sh`.", "",
           "| kernel | calls | avg us | total ms |", "|---|---|---|---|"]
    total = 0.0
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
        name = r["Name"].split("(")[0].replace("void ", "").replace("lfmkd::", "")
        out.append(f"| `{name}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['TotalDurationNs'])/1e6:.2f} |")
        total += float(r["TotalDurationNs"]) / 1e6
    out += ["", f"Sum of kernel time per batch: {total / calls:.2f} ms (one stream; `mkd_pool<0, 1, 4, 1>` is the keypoint-mode form of the describe",
            "kernel: 4 describe + 4 producer waves per workgroup, patches sampled into its LDS ring).", ""]

    def per_kernel(sub, counter):
        f = newest(sub, "*counter_collection.csv")
        agg = collections.defaultdict(float)
        if f:
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == counter:
                    agg[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("lfmkd::", "")] += float(r["Counter_Value"])
        return agg
    fetch, write = per_kernel("kp_fetch", "FETCH_SIZE"), per_kernel("kp_write", "WRITE_SIZE")
    tcp = per_kernel("kp_tcp", "TCP_TOTAL_CACHE_ACCESSES")
    ours = lambda k: k.startswith(("mkd_pool", "pyr_", "sample_patches"))      # the library's kernels (torch's own synthesise the inputs)
    per_call = {}
    if fetch and write:
        out += ["## HBM traffic per batch (separate `--pmc` passes), the library's kernels.  FETCH_SIZE / WRITE_SIZE in KiB as counted.",
                "MI355X_MICROARCH.md: on gfx950 FETCH_SIZE counts half the bytes of a 16-byte-per-lane streaming read, WRITE_SIZE is exact,",
                "other widths are to be calibrated on a known byte count.  Calibration (`tools/calibrate_fetch.sh`,",
                f"`profiles/{tag}_fetch_calibration.txt`: `tools/micro/tile_copy` moves a known 314.6 MB each way in the pyramid kernels' access",
                "shapes): FETCH_SIZE counts exactly 0.500 x the bytes of EVERY row-contiguous read, 4 bytes per lane and 16 bytes per lane",
                "alike, WRITE_SIZE 1.000 x.  So the pyramid kernels' reads are doubled in the `read GB` column.  The describe kernel's",
                "reads are the producers' 8-byte gathers (its 16-byte LDS-DMA requests fetch the LUT, which stays in L2): not calibrated",
                "(no known byte count for a gather); given as counted and, in the second total, doubled like the rest.", "",
                "| kernel | FETCH_SIZE KiB | read GB (corrected) | WRITE_SIZE KiB | written GB | L1 accesses per keypoint |", "|---|---|---|---|---|---|"]
        tot = tot_counted = pool_read = 0.0
        for k in sorted((k for k in set(fetch) | set(write) if ours(k)), key=lambda k: -(fetch.get(k, 0) + write.get(k, 0))):
            rb, wb = fetch.get(k, 0) / calls * 1024, write.get(k, 0) / calls * 1024
            tot_counted += rb + wb
            if k.startswith("mkd_pool"):
                pool_read += rb
                shown = f"{rb / 1e9:.3f} (as counted)"
            else:
                rb *= 2.0
                shown = f"{rb / 1e9:.3f}"
            tot += rb + wb
            acc = f"{tcp[k] / calls / n_kp:.0f}" if k in tcp else ""
            out.append(f"| `{k}` | {fetch.get(k, 0) / calls:.0f} | {shown} | {write.get(k, 0) / calls:.0f} | {wb / 1e9:.3f} | {acc} |")
        alg = n_kp * (16 + 512) + frames * w * h * 4
        out += ["", f"Total {tot / 1e9:.2f} GB per batch = {tot / n_kp:.0f} B per descriptor with the calibrated correction ({(tot + pool_read) / n_kp:.0f} B with the describe "
                f"kernel's reads doubled too); algorithmic {alg / 1e9:.3f} GB = {alg / n_kp:.0f} B per descriptor",
                f"(16 B keypoint + 512 B descriptor + frame bytes / keypoints, SURVEY 8d) -> ratio {tot / alg:.1f}.",
                f"In the units of the earlier rounds (every counter as counted, no correction): {tot_counted / n_kp:.0f} B per descriptor; round 2, with the",
                "sampled patches written by `sample_patches` and read back by `mkd_pool`: 14121 B in those units.  What is left above the",
                "algorithmic bytes is the pyramid: the frame read with the halo rows of its row tiles (each tile's halo is fetched again: the",
                "tiles of a frame run on different XCDs), level 0 written once and read again for level 1, every level written with its",
                "mirrored apron, each pyramid read once by the producers.", ""]
        per_call["configs2_256x640x480_2k_keypoints"] = tot
    # configs[1]: one 1080p frame, 10 000 keypoints
    f1, w1 = per_kernel("kp1_fetch", "FETCH_SIZE"), per_kernel("kp1_write", "WRITE_SIZE")
    if f1 and w1:
        rd1 = lambda k, v: v * (1.0 if k.startswith("mkd_pool") else 2.0)      # the same correction
        tot1 = sum(rd1(k, v) for k, v in f1.items() if ours(k)) / calls * 1024 + sum(v for k, v in w1.items() if ours(k)) / calls * 1024
        alg1 = 10000 * (16 + 512) + 1920 * 1080 * 4
        out += [f"configs[1] (one 1920 x 1080 frame, 10 000 keypoints; `prof_keypoints.py configs1`): {tot1 / 1e6:.1f} MB per call = "
                f"{tot1 / 10000:.0f} B per descriptor (same correction); algorithmic {alg1 / 1e6:.1f} MB = {alg1 / 10000:.0f} B per descriptor "
                f"-> ratio {tot1 / alg1:.1f}.", ""]
        per_call["configs1_1080p_10k_keypoints"] = tot1
    # the reference's own settings: one 1080p frame, 3000 keypoints
    fr, wr = per_kernel("kpr_fetch", "FETCH_SIZE"), per_kernel("kpr_write", "WRITE_SIZE")
    if fr and wr:
        rdr = lambda k, v: v * (1.0 if k.startswith("mkd_pool") else 2.0)      # the same correction
        totr = sum(rdr(k, v) for k, v in fr.items() if ours(k)) / calls * 1024 + sum(v for k, v in wr.items() if ours(k)) / calls * 1024
        algr = 3000 * (16 + 512) + 1920 * 1080 * 4
        out += [f"The reference's own settings (one 1920 x 1080 frame, 3000 keypoints; `prof_keypoints.py refdefaults`): {totr / 1e6:.1f} MB per call = "
                f"{totr / 3000:.0f} B per descriptor (same correction); algorithmic {algr / 1e6:.1f} MB = {algr / 3000:.0f} B per descriptor "
                f"-> ratio {totr / algr:.1f}.", ""]
        per_call["reference_defaults_1080p_3000_keypoints"] = totr
    # configs[3] in its own form, one GPU's share: 128 frames 1920 x 1080, 8192 keypoints each
    f3, w3 = per_kernel("kp3_fetch", "FETCH_SIZE"), per_kernel("kp3_write", "WRITE_SIZE")
    kp3_stats = newest("kp3_stats", "*kernel_stats.csv")
    if f3 and w3:
        rd3 = lambda k, v: v * (1.0 if k.startswith("mkd_pool") else 2.0)      # the same correction
        n3 = 128 * 8192
        tot3 = sum(rd3(k, v) for k, v in f3.items() if ours(k)) / calls * 1024 + sum(v for k, v in w3.items() if ours(k)) / calls * 1024
        alg3 = n3 * (16 + 512) + 128 * 1920 * 1080 * 4
        out += [f"configs[3] in its own form, one GPU's share (128 frames 1920 x 1080, 8192 keypoints each = 2^20 descriptors; "
                f"`prof_keypoints.py configs3`): {tot3 / 1e9:.2f} GB per call = {tot3 / n3:.0f} B per descriptor (same correction); "
                f"algorithmic {alg3 / 1e9:.3f} GB = {alg3 / n3:.0f} B per descriptor -> ratio {tot3 / alg3:.1f}.", ""]
        per_call["configs3_128x1080p_8192_keypoints"] = tot3
        if kp3_stats:
            # per-kernel bytes (counter passes, same correction) over the kernel trace's time: GB/s per kernel, and per CALL
            # (a kernel launched three times per call -- the decimating one -- is summed)
            out += ["| kernel (configs[3] own form) | launches per call | us per call | read GB | written GB | GB/s |", "|---|---|---|---|---|---|"]
            pyr_us = pyr_bytes = 0.0
            for r in sorted((r for r in csv.DictReader(open(kp3_stats)) if "lfmkd" in r["Name"]), key=lambda r: -float(r["TotalDurationNs"])):
                name = r["Name"].split("(")[0].replace("void ", "").replace("lfmkd::", "")
                us = float(r["TotalDurationNs"]) / 1e3 / calls
                rb = rd3(name, f3.get(name, 0.0)) / calls * 1024
                wb = w3.get(name, 0.0) / calls * 1024
                out.append(f"| `{name}` | {int(r['Calls']) // calls} | {us:.1f} | {rb / 1e9:.3f} | {wb / 1e9:.3f} | {(rb + wb) / us / 1e3:.0f} |")
                if name.startswith("pyr_"):
                    pyr_us += us
                    pyr_bytes += rb + wb
            minimal = 128 * 1920 * 1080 * 4 * (1 + 1 + 1 / 3)
            out += ["", f"Pyramid build (`lf_mkd_set_images_device`, all `pyr_*` launches of a call): {pyr_us:.0f} us for {pyr_bytes / 1e9:.2f} GB counted = "
                    f"{pyr_bytes / pyr_us / 1e3:.0f} GB/s; on its minimal bytes (frames read once, every level written once: {minimal / 1e9:.2f} GB) "
                    f"{minimal / pyr_us / 1e3:.0f} GB/s.", ""]
    if per_call:
        json.dump({"tag": tag, "source_sha256": stamp(("mkd_describe.hip", "mkd_pyramid.hip", "mkd_sample.h", "mkd_device.h", "lf_mkd.cpp")),
                   "git_head_at_summary": head, "hbm_bytes_per_call": per_call,
                   "note": "WRITE_SIZE as counted + FETCH_SIZE doubled for the pyramid kernels (row-contiguous reads count 0.500 x their "
                           "bytes on gfx950: profiles/r03_fetch_calibration.txt), as counted for the describe kernel's 8-byte gathers"},
                  open(os.path.join(ROOT, "profiles", "traffic_keypoint_mode.json"), "w"), indent=1)
    shutil.copy(kp_stats, os.path.join(ROOT, "profiles", f"{tag}_keypoint_mode_kernel_stats.csv"))
    open(os.path.join(ROOT, "profiles", f"{tag}_keypoint_mode.md"), "w").write("\n".join(out) + "\n")
    print("\n".join(out))
