#!/usr/bin/env python3
"""Kernel micro-benchmarks on the GPU box (hipEvent timing through hgmm_profile_*).

    python tools/kbench.py [--n 1000000] [--j 800] [--reps 20]
Environment switches read by the library at launch time are swept here."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--j", type=int, default=800)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    import hgmm_amd
    ctx = hgmm_amd.Context(0)
    N, J = args.n, args.j
    X = np.random.RandomState(0).rand(N, 3).astype(np.float32)
    idx = np.random.RandomState(100).choice(N, J, replace=False)
    mu0 = X[idx].copy()
    w0 = (np.ones(J) / J).astype(np.float32)
    cov0 = (0.1 * np.ones((J, 3))).astype(np.float32)
    ctx.set_points(X)
    inv, mu, w, cov, lls, _ = ctx.flat_train(3, 0.0, mu0, cov0, w0, "diag", "W")
    lr = ctx.empty((N, J), np.float32)
    alg = 12 * N + 4 * N * J + 4 * N + 28 * J
    res = {}

    def time_estep(label):
        ctx.flat_estep(inv, mu, w, "diag", "W", out=lr)
        ctx.profile_reset(); ctx.profile_enable(True)
        for _ in range(args.reps):
            ctx.flat_estep(inv, mu, w, "diag", "W", out=lr)
        ctx.profile_enable(False)
        ms, n = ctx.profile_get("flat_estep")
        res[label] = {"ms": ms / n, "GBs": alg / (ms / n * 1e-3) / 1e9}
        print("%-28s %.4f ms  %.0f GB/s (%.1f%% of 8 TB/s)" % (label, ms / n, res[label]["GBs"], res[label]["GBs"] / 80))

    for label, nt, mode, gm in (("fill 8 WG/CU grid-stride nt", True, 0, 8), ("fill 8 WG/CU grid-stride", False, 0, 8),
                               ("fill 1 WG/CU grid-stride", False, 0, 1), ("fill 1 WG/CU grid-stride nt", True, 0, 1),
                               ("fill 1 WG/CU private streams", False, 1, 1)):
        ctx.util_fill(lr, 1.0, nt, mode, gm)
        ctx.profile_reset(); ctx.profile_enable(True)
        for _ in range(args.reps):
            ctx.util_fill(lr, 1.0, nt, mode, gm)
        ctx.profile_enable(False)
        ms, n = ctx.profile_get("util_fill")
        print("%-32s %.4f ms  %.0f GB/s  (pure 16-byte store stream, %d MB)" % (label, ms / n, 4 * N * J / (ms / n * 1e-3) / 1e9, 4 * N * J >> 20))
        res[label] = {"ms": ms / n, "GBs": 4 * N * J / (ms / n * 1e-3) / 1e9}
    os.environ["HGMM_ESTEP_NT"] = "1"
    for rnd in range(2):
        for rows, bpc in (("1", "1"), ("4", "1"), ("4", "2")):
            os.environ["HGMM_ESTEP_BPC"] = bpc
            os.environ["HGMM_ESTEP_ROWS"] = rows
            time_estep("estep rows=%s workgroups/CU=%s" % (rows, bpc if rows != "1" else "2"))
    for k in ("HGMM_ESTEP_ROWS", "HGMM_ESTEP_NT", "HGMM_ESTEP_BPC"):
        os.environ.pop(k)

    for pk_ in ("0", "1", "0", "1"):
        os.environ["HGMM_FUSED_PK"] = pk_
        ctx.flat_train(3, 0.0, mu0, cov0, w0, "diag", "W")
        ctx.profile_reset(); ctx.profile_enable(True)
        o = ctx.flat_train(args.reps, 0.0, mu0, cov0, w0, "diag", "W")
        ctx.profile_enable(False)
        ms, n = ctx.profile_get("flat_fused")
        print("%-28s %.4f ms  lls[-1]=%.7f" % ("fused paired=%s" % pk_, ms / n, o[4][-1]))
    os.environ.pop("HGMM_FUSED_PK")
    for bpc in ("1", "2", "3"):
        os.environ["HGMM_FUSED_BPC"] = bpc
        ctx.profile_reset(); ctx.profile_enable(True)
        ctx.flat_train(args.reps, 0.0, mu0, cov0, w0, "diag", "W")
        ctx.profile_enable(False)
        ms, n = ctx.profile_get("flat_fused")
        print("%-28s %.4f ms  %.3g pairs/s" % ("fused blocks/CU=%s" % bpc, ms / n, N * J / (ms / n * 1e-3)))
    os.environ.pop("HGMM_FUSED_BPC")
    for rr, bpc in (("0", "2"), ("1", "1"), ("1", "2"), ("1", "3"), ("0", "2"), ("1", "2")):
        os.environ["HGMM_MSTEP_BPC"] = bpc
        os.environ["HGMM_MSTEP_RR"] = rr
        ctx.flat_mstep(lr.exp(), "diag", "W", centre_hint=mu)
        ctx.profile_reset(); ctx.profile_enable(True)
        for _ in range(5):
            ctx.flat_mstep(lr.exp(), "diag", "W", centre_hint=mu)
        ctx.profile_enable(False)
        ms, n = ctx.profile_get("flat_mstep")
        print("%-28s %.4f ms  %.0f GB/s" % ("mstep rr=%s blocks/CU=%s" % (rr, bpc), ms / n, (4 * N * J + 12 * N) / (ms / n * 1e-3) / 1e9))
    os.environ.pop("HGMM_MSTEP_BPC"); os.environ.pop("HGMM_MSTEP_RR")

    ctx.profile_reset(); ctx.profile_enable(True)
    ctx.flat_train(args.reps, 0.0, mu0, cov0, w0, "diag", "W")
    ctx.profile_enable(False)
    ms, n = ctx.profile_get("flat_fused")
    print("%-28s %.4f ms  %.3g pairs/s" % ("fused", ms / n, N * J / (ms / n * 1e-3)))
    res["fused"] = {"ms": ms / n}

    ctx.flat_mstep(lr.exp(), "diag", "W", centre_hint=mu)
    ctx.profile_reset(); ctx.profile_enable(True)
    for _ in range(5):
        ctx.flat_mstep(lr.exp(), "diag", "W", centre_hint=mu)
    ctx.profile_enable(False)
    ms, n = ctx.profile_get("flat_mstep")
    print("%-28s %.4f ms  %.0f GB/s" % ("mstep(log_resp.exp())", ms / n, (4 * N * J + 12 * N) / (ms / n * 1e-3) / 1e9))
    res["mstep"] = {"ms": ms / n}
    # flat full-covariance EM (fp64, MFMA statistics) at the same size
    P64 = X.astype(np.float64)
    ctx.set_points(P64)
    ctx.fullcov_fit(J, 1e-30, 1e-4, P64[idx], 0.01, 2)
    ctx.profile_reset(); ctx.profile_enable(True)
    import time
    t0 = time.perf_counter()
    ctx.fullcov_fit(J, 1e-30, 1e-4, P64[idx], 0.01, 5)
    dt = time.perf_counter() - t0
    ctx.profile_enable(False)
    pm, pn = ctx.profile_get("full_pass")
    mm, mn = ctx.profile_get("full_moments")
    print("fullcov J=%d N=%d: %.2f ms/iteration (pass %.3f ms, moments %.3f ms)" % (J, N, dt / 5 * 1e3, pm / pn, mm / mn))
    res["fullcov"] = {"ms_per_iter": dt / 5 * 1e3, "pass_ms": pm / pn, "moments_ms": mm / mn}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
