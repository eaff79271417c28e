#!/usr/bin/env python3
"""BASELINE configs[0] / [1] (bun000, J = 100 / 800, 20 iterations) alone, for a kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d <dir> -o kt -- python tools/bunny_trace.py [J] [reps]
then  python tools/trace_summary.py <dir> --seq 12"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hgmm_amd
J = int(sys.argv[1]) if len(sys.argv) > 1 else 100
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = hgmm_amd.Context(0)
X = np.load(os.path.join(ROOT, "tests", "golden", "bun000_xyz.npy")).astype(np.float32)
idx = np.random.RandomState(100).choice(len(X), J, replace=False)
mu0 = X[idx].copy(); w0 = (np.ones(J) / J).astype(np.float32); cov0 = (0.1 * np.ones((J, 3))).astype(np.float32)
ctx.set_points(X)
ctx.flat_train(20, 0.0, mu0, cov0, w0, "diag", "W")
ts = []
for _ in range(reps):
    t0 = time.perf_counter()
    ctx.flat_train(20, 0.0, mu0, cov0, w0, "diag", "W")
    ts.append(time.perf_counter() - t0)
print("bun000 J=%d: 20 iterations in %s ms -> %.0f it/s" % (J, np.round(np.array(ts) * 1e3, 3), 20 / np.median(ts)))
