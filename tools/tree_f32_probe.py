#!/usr/bin/env python3
"""bench.py's tree_1M leg by itself: the N = 10^6, L = 4 build with the stop rule's pdfs in float64 and in float32
(hgmm_tree_set_precision), kernel times and evaluated pairs of both.  -> profiles/r05/tree_f32_probe.log"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench        # noqa: E402
import hgmm_amd     # noqa: E402

ctx = hgmm_amd.Context(0)
out = bench.tree_1m_leg(ctx)
out.pop("roofline", None)
print(json.dumps(out, indent=1))
ctx.close()
