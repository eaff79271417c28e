#!/usr/bin/env python3
"""m_step's kernel with plain / non-temporal loads x row length: rows that are not a whole number of 128-byte lines share
lines with their neighbours -- does a load that does not allocate fetch those lines twice?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hgmm_amd
ctx = hgmm_amd.Context(0)
N = 1_000_000
X = np.random.RandomState(0).rand(N, 3).astype(np.float32)
ctx.set_points(X)
for J in (800, 100, 72, 200, 37, 513, 1000):
    resp = ctx.empty((N, J), np.float32)
    ctx.util_fill(resp, 1.0 / J, False, 0, 1)
    mu = X[:J].copy(); dmu = ctx.to_device(mu)
    out = []
    for nt in ("0", "1"):
        os.environ["HGMM_MSTEP_NT"] = nt
        r = []
        for rnd in range(3):
            for _ in range(3): ctx.flat_mstep(resp, "diag", "W", centre_hint=dmu, device_out=True)
            ctx.profile_reset(); ctx.profile_enable(True)
            for _ in range(10): ctx.flat_mstep(resp, "diag", "W", centre_hint=dmu, device_out=True)
            ctx.profile_enable(False)
            ms, n = ctx.profile_get("flat_mstep"); r.append(ms / n)
        out.append(float(np.median(r)))
    gb = (4.0 * N * J + 12.0 * N) / 1e6
    print("J=%4d (%5d-byte rows): plain %.4f ms = %5.0f GB/s | non-temporal %.4f ms = %5.0f GB/s" % (J, 4 * J, out[0], gb / out[0], out[1], gb / out[1]))
    del resp
