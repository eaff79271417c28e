#!/usr/bin/env python3
"""One warm KMeans(k = 800, max_iter = 50) fit on the C3 frame alone, for a kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d <dir> -o kt -- python tools/kmeans_trace.py;  python tools/trace_summary.py <dir>"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hgmm_amd, bench
from hgmm_amd.kmeans import KMeans
ctx = hgmm_amd.Context(0)
X = bench.synth_frame(0).astype(np.float64)
KMeans(n_clusters=800, random_state=1, max_iter=50, ctx=ctx).fit(X)
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    km = KMeans(n_clusters=800, random_state=1, max_iter=50, ctx=ctx).fit(X)
    ts.append((time.perf_counter() - t0) * 1e3)
print("KMeans k=800 on 1M points, warm fits ms:", np.round(ts, 2), "seeding", round(km.seeding_ms_, 2), "iterations", km.n_iter_)
