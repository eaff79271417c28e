#!/usr/bin/env python3
"""How much of the level log-likelihood's time at N = 1e6 (bench.py's tree_1M) is tile building -- node loads, reach
tests, compaction, barriers -- and how much is pdf evaluation: the same build with HGMM_TREE_LL_NOEVAL=1 (the kernels
skip the evaluations; q is wrong, levels stop early, so the comparison is per LAUNCH and per level)."""
import os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    import hgmm_amd
    ctx = hgmm_amd.Context(0)
    P = np.random.RandomState(0).rand(1_000_000, 3).astype(np.float32).astype(np.float64)
    idx = np.random.RandomState(72).randint(len(P), size=4680)
    ctx.set_points(P)
    for L in (1, 2, 3, 4):                                  # builds of growing depth: the deepest level's launches by difference
        ctx.tree_build(L, 80.0, 1e-4, P[idx[:8 * (8 ** L - 1) // 7]], 0.01, 4, want_leaf=False)
        ctx.profile_reset(); ctx.profile_enable(True)
        out = ctx.tree_build(L, 80.0, 1e-4, P[idx[:8 * (8 ** L - 1) // 7]], 0.01, 4, want_leaf=False)
        ctx.profile_enable(False)
        ms, n = ctx.profile_get("tree_loglik")
        print("L=%d iterations %s  log-likelihood kernels: %d launches, %.3f ms total" % (L, list(out[4]), n, ms), flush=True)
else:
    for tag, env in (("full kernels", {}), ("HGMM_TREE_LL_NOEVAL=1 (tiles built, no pdf evaluated)", {"HGMM_TREE_LL_NOEVAL": "1"})):
        print("==", tag, flush=True)
        e = dict(os.environ); e.update(env)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=e)
