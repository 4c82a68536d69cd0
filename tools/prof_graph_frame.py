#!/usr/bin/env python3
"""One frame size through the recorded pipeline (lf_mkd_stream_*), a few frames one at a time: run under
`rocprofv3 --kernel-trace` to see the frame's kernel timeline (tools/graph_timeline.sh).  Usage: prof_graph_frame.py W H TOPN"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import bench_detect as bd

if __name__ == "__main__":
    w, h, top_n = (int(x) for x in sys.argv[1:4])
    bd.run_graph(w, h, top_n, 8, "frame")
