#!/usr/bin/env python3
"""Why is the materialising E-step slower inside the API-granular loop (e_step -> m_step -> host) than back to back?
Times the E-step kernel (hipEvents) at C3 size in several call patterns."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hgmm_amd
import bench

ctx = hgmm_amd.Context(0)
X = bench.synth_frame(0)
mu0, w0, cov0 = bench.init_params(X)
ctx.set_points(X)
inv0 = (1.0 / np.sqrt(cov0)).astype(np.float32)
lr = ctx.empty((bench.N_POINTS, bench.J_COMP), np.float32)
lr2 = ctx.empty((bench.N_POINTS, bench.J_COMP), np.float32)


def run(label, body, reps=8):
    for _ in range(3):
        body()
    ctx.synchronize()
    ctx.profile_reset(); ctx.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        body()
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / reps
    ctx.profile_enable(False)
    e_ms, e_n = ctx.profile_get("flat_estep")
    m_ms, m_n = ctx.profile_get("flat_mstep")
    print("%-58s wall %.3f ms/iter  E kernel %.3f ms (%d)  M kernel %.3f ms (%d)" %
          (label, dt * 1e3, e_ms / max(e_n, 1), e_n, m_ms / max(m_n, 1), m_n), flush=True)


def e_only():
    ctx.flat_estep(inv0, mu0, w0, "diag", "W", out=lr, lazy_mean=True)


def e_sync():
    ctx.flat_estep(inv0, mu0, w0, "diag", "W", out=lr, lazy_mean=True)
    ctx.synchronize()


def e_then_m():
    ctx.flat_estep(inv0, mu0, w0, "diag", "W", out=lr, lazy_mean=True)
    ctx.flat_mstep(lr.exp(), "diag", "W", centre_hint=mu0)


def e_other_then_m():
    # E writes lr2, M reads lr (never rewritten): does M's read of the SAME buffer matter?
    ctx.flat_estep(inv0, mu0, w0, "diag", "W", out=lr2, lazy_mean=True)
    ctx.flat_mstep(lr.exp(), "diag", "W", centre_hint=mu0)


def m_only():
    ctx.flat_mstep(lr.exp(), "diag", "W", centre_hint=mu0)


run("E back to back, no sync", e_only)
run("E, sync, E, sync", e_sync)
run("E -> M (same buffer) -> host", e_then_m)
run("E (other buffer) -> M -> host", e_other_then_m)
run("M only", m_only)
os.environ["HGMM_ESTEP_NT"] = "0"
run("E -> M, E with temporal stores (HGMM_ESTEP_NT=0)", e_then_m)
run("E back to back, temporal stores", e_only)
os.environ.pop("HGMM_ESTEP_NT", None)
for bpc, grid in ((1, 160), (1, 192), (1, 224), (1, 256), (2, 320), (2, 384), (2, 512)):
    os.environ["HGMM_ESTEP_BPC"] = str(bpc)
    os.environ["HGMM_ESTEP_GRID"] = str(grid)
    run("E -> M -> host, E grid %d" % grid, e_then_m)
    run("E, sync, E, sync;   E grid %d" % grid, e_sync)
