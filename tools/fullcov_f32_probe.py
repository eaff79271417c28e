#!/usr/bin/env python3
"""Flat full-covariance EM at C3 size (uniform cloud N = 10^6, J = 800) with the float64 tile and with the float32 tile
(Context.tree_set_precision(np.float32)): kernel ms per launch (hipEvent profiler), and how far the float32 fit's
(pi, mu, Sigma), q and labels are from the float64 fit's after the same iterations.
    python tools/fullcov_f32_probe.py [iterations]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import hgmm_amd  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    ctx = hgmm_amd.Context(0)
    P = bench.synth_frame(0).astype(np.float64)
    J = bench.J_COMP
    idx = np.random.RandomState(100).choice(len(P), J, replace=False)
    ctx.set_points(P)
    res = {}
    for name, dt in (("float64", np.float64), ("float32", np.float32), ("float64", np.float64), ("float32", np.float32)):
        ctx.tree_set_precision(dt)
        ctx.fullcov_fit(J, 1e-30, 1e-4, P[idx], 0.01, 2)
        ctx.profile_reset()
        ctx.profile_enable(True)
        t0 = time.perf_counter()
        out = ctx.fullcov_fit(J, 1e-30, 1e-4, P[idx], 0.01, iters)
        wall = time.perf_counter() - t0
        ctx.profile_enable(False)
        ms, n = ctx.profile_get("full_fused")
        print("%s tile: one-pass kernel %.4f ms per launch (%d launches), fit %.2f ms per iteration" % (name, ms / n, n, 1e3 * wall / iters))
        res[name] = out
    # phase clocks of one E-step launch at the fitted parameters (cycles per wave of workgroup 7: A, B, C, barrier wait)
    pi_f, mu_f, cov_f = res["float64"][0], res["float64"][1], res["float64"][2]
    for name, dt in (("float64", np.float64), ("float32", np.float32)):
        ctx.tree_set_precision(dt)
        ctx.fullcov_estep(pi_f, mu_f, cov_f)
        ctx.fullcov_phase_clocks(True)
        ctx.fullcov_estep(pi_f, mu_f, cov_f)
        ck = ctx.fullcov_phase_clocks(False)
        tot = ck.sum(1)
        print("%s tile, phase clocks of workgroup 7 (k cycles; waves 0..7): A %s | B %s | C %s | wait %s | sum of the slowest wave %.0f k"
              % (name, (ck[:, 0] // 1000).tolist(), (ck[:, 1] // 1000).tolist(), (ck[:, 2] // 1000).tolist(), (ck[:, 3] // 1000).tolist(),
                 tot.max() / 1e3))
    a, b = res["float64"], res["float32"]
    pi_a, mu_a, cov_a, lab_a, q_a = a
    pi_b, mu_b, cov_b, lab_b, q_b = b
    sig2 = np.abs(np.einsum("jii->j", cov_a)) / 3.0
    print("after %d iterations: labels differing %d of %d; max |dq| / |q| %.3g; max |d pi| / pi %.3g; max |d mu| %.3g "
          "(extent 1); max |d cov| / mean variance %.3g"
          % (iters, int((lab_a != lab_b).sum()), len(lab_a), float(np.max(np.abs(q_a - q_b) / np.abs(q_a))),
             float(np.max(np.abs(pi_a - pi_b) / np.maximum(pi_a, 1e-300))), float(np.abs(mu_a - mu_b).max()),
             float(np.max(np.abs(cov_a - cov_b).reshape(J, -1).max(1) / sig2))))
    ctx.tree_set_precision(np.float64)
    ctx.close()


if __name__ == "__main__":
    main()
