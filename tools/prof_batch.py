#!/usr/bin/env python3
"""rocprofv3 target: only the batched configs[2] pipeline (tools/bench_detect.run_batch)."""
import os, sys
sys.argv = [sys.argv[0], "batch-only"]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import importlib.util
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_detect.py")).read()
src = src[:src.index("run_batch(640, 480, 1400, 256")]      # definitions only
exec(compile(src, "bench_detect_defs", "exec"))
run_batch(640, 480, 1400, 256, "configs[2] batched")
