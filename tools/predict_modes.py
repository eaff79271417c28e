#!/usr/bin/env python3
"""predict() on the C3 frame (N = 10^6, J = 800, diag, flavour W) with the index search in the per-lane VALU stream
(HGMM_PREDICT_MODE=0, rounds 3-4) and on the scalar unit behind a ballot (HGMM_PREDICT_MODE=1): hipEvent time per launch,
labels compared bit for bit between the modes and with flat_estep's arg-max.  Also J = 100 / 400 / 1024 and a cloud with
exact ties (duplicated components) that must take the slow path.

    python tools/predict_modes.py  ->  profiles/r05/predict_modes.log (via gpurun_out)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench            # noqa: E402
import hgmm_amd         # noqa: E402


def timed(ctx, p, reps=30):
    for _ in range(5):
        lab = ctx.flat_predict(*p)
    ctx.synchronize()
    ctx.profile_reset()
    ctx.profile_enable(True)
    for _ in range(reps):
        lab = ctx.flat_predict(*p)
    ctx.synchronize()
    ctx.profile_enable(False)
    ms, n = ctx.profile_get("flat_estep")
    return ms / max(n, 1), lab.get()


def main():
    ctx = hgmm_amd.Context(0)
    X = bench.synth_frame(0)
    ctx.set_points(X)
    for J in (800, 100, 400, 1024, 64):
        mu0, w0, cov0 = bench.init_params(X, J)
        inv, mu, w, cov, lls, _ = ctx.flat_train(5, 0.0, mu0, cov0, w0, "diag", "W")
        p = (ctx.to_device(inv), ctx.to_device(mu), ctx.to_device(w))
        out = {}
        for mode in ("0", "1", "0", "1"):
            os.environ["HGMM_PREDICT_MODE"] = mode
            ms, lab = timed(ctx, p)
            out.setdefault(mode, []).append(ms)
            out["lab" + mode] = lab
        same = bool(np.array_equal(out["lab0"], out["lab1"]))
        _, _, _, am = ctx.flat_estep(inv, mu, w, "diag", "W", want_log_resp=False, want_argmax=True)
        eq_e = bool(np.array_equal(am.get(), out["lab1"]))
        print("J %4d: mode 0 %s ms   mode 1 %s ms   labels equal %s   equal to e_step's arg-max %s"
              % (J, ["%.4f" % v for v in out["0"]], ["%.4f" % v for v in out["1"]], same, eq_e), flush=True)
    # exact ties: every component twice (and a zero-weight block): the smallest index must win in both modes
    J = 400
    mu0, w0, cov0 = bench.init_params(X, J)
    mu2 = np.concatenate([mu0, mu0]); w2 = np.concatenate([w0, w0]) / 2; inv2 = np.full((2 * J, 3), 3.0, np.float32)
    p = (ctx.to_device(inv2), ctx.to_device(mu2), ctx.to_device(w2.astype(np.float32)))
    labs = {}
    for mode in ("0", "1"):
        os.environ["HGMM_PREDICT_MODE"] = mode
        ms, labs[mode] = timed(ctx, p, reps=10)
        print("ties J = 800 (400 doubled): mode %s %.4f ms, max label %d" % (mode, ms, labs[mode].max()), flush=True)
    print("ties: labels equal", bool(np.array_equal(labs["0"], labs["1"])), "all below 400:", bool(labs["1"].max() < 400))
    ctx.close()


if __name__ == "__main__":
    main()
