#!/usr/bin/env python3
"""gpurun_out/prof_<tag>_{detect,match}/ (tools/profile_next_rows.sh) -> profiles/<tag>_next_rows.md + the kernel_stats CSVs."""
import csv, glob, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out = [f"# rocprofv3 summary {tag}: the widened rows (detector, orientation, pipelines, matcher)\n",
       "Commands profiled on one MI355X via `tools/profile_next_rows.sh` (`rocprofv3 --kernel-trace --stats`):",
       "`python3 tools/bench_detect.py` and `python3 tools/bench_match.py`.  Only this library's kernels are listed; the",
       "frames are synthesised with torch (its kernels are left out).\n"]
for what in ("detect", "match"):
    fs = glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{tag}_{what}", "**", "*kernel_stats.csv"), recursive=True)
    if not fs:
        continue
    fs.sort(key=os.path.getmtime, reverse=True)      # the newest run
    shutil.copy(fs[0], os.path.join(ROOT, "profiles", f"{tag}_{what}_kernel_stats.csv"))
    rows = [r for r in csv.DictReader(open(fs[0])) if "lfmkd" in r["Name"]]
    out.append(f"## `tools/bench_{what}.py`\n")
    out.append("| kernel | calls | total ms | avg us | min us | max us |")
    out.append("|---|---|---|---|---|---|")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
        name = r["Name"].split("(")[0].replace("void ", "").replace("lfmkd::", "")
        out.append(f"| `{name}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e3:.1f} | "
                   f"{float(r['MinNs'])/1e3:.1f} | {float(r['MaxNs'])/1e3:.1f} |")
    log = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_{what}.log")
    if os.path.exists(log):
        lines = [l.rstrip() for l in open(log) if ("configs[" in l or " x " in l) and "rocprofv3" not in l]
        out.append("\nOutput of the profiled run (timings include the profiler's overhead):\n\n```")
        out += lines
        out.append("```\n")
open(os.path.join(ROOT, "profiles", f"{tag}_next_rows.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out)[:3000])
