#!/usr/bin/env python3
"""The materialising E-step inside the device-array loop (e_step -> m_step -> elementwise, nothing waits): what the
kernel that ran before it does to its duration, and which grid / store flavour suits it there."""
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
e6, e8 = np.float32(1e-6), np.float32(1e-8)


def report(label, body, reps=8):
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
    f_ms, f_n = ctx.profile_get("util_fill")
    print("%-66s wall %.3f ms/iter  E %.3f ms (%d)  M %.3f ms (%d)  fill %.3f (%d)" %
          (label, dt * 1e3, e_ms / max(e_n, 1), e_n, m_ms / max(m_n, 1), m_n, f_ms / max(f_n, 1), f_n), flush=True)


state = {}


def reset_state():
    state["p"] = (ctx.to_device(inv0), ctx.to_device(mu0), ctx.to_device(w0))


def dev_loop():
    inv, mu, w = state["p"]
    ctx.flat_estep(inv, mu, w, "diag", "W", out=lr, lazy_mean=True)
    w, mu, cov = ctx.flat_mstep(lr.exp(), "diag", "W", centre_hint=mu, device_out=True)
    state["p"] = (1.0 / (np.sqrt(cov + e6) + e8), mu, w)


def e_only():
    inv, mu, w = state["p"]
    ctx.flat_estep(inv, mu, w, "diag", "W", out=lr, lazy_mean=True)


def m_sleep_e():
    inv, mu, w = state["p"]
    ctx.flat_mstep(lr.exp(), "diag", "W", centre_hint=mu, device_out=True)
    ctx.synchronize()
    time.sleep(0.02)
    ctx.flat_estep(inv, mu, w, "diag", "W", out=lr, lazy_mean=True)
    ctx.synchronize()


def e_sleep_e():
    inv, mu, w = state["p"]
    ctx.synchronize()
    time.sleep(0.02)
    ctx.flat_estep(inv, mu, w, "diag", "W", out=lr, lazy_mean=True)
    ctx.synchronize()


def fill_e():
    inv, mu, w = state["p"]
    ctx.util_fill(lr2, 0.0, False, 0, 1)
    ctx.flat_estep(inv, mu, w, "diag", "W", out=lr, lazy_mean=True)


def m_other_e():
    # M reads lr2 (never rewritten), E writes lr
    inv, mu, w = state["p"]
    ctx.flat_mstep(lr2.exp(), "diag", "W", centre_hint=mu, device_out=True)
    ctx.flat_estep(inv, mu, w, "diag", "W", out=lr, lazy_mean=True)


reset_state()
report("E back to back (device parameters)", e_only)
report("E, 20 ms idle, E", e_sleep_e)
report("M, 20 ms idle, E", m_sleep_e)
report("fill (write stream, other buffer) -> E", fill_e)
ctx.flat_estep(inv0, mu0, w0, "diag", "W", out=lr2)
report("M (other buffer) -> E", m_other_e)
reset_state(); report("device loop, default", dev_loop)
for nt in ("0",):
    os.environ["HGMM_ESTEP_NT"] = nt
    reset_state(); report("device loop, E temporal stores", dev_loop)
    os.environ.pop("HGMM_ESTEP_NT")
for bpc, grid in ((1, 192), (1, 256), (2, 320), (2, 384), (2, 448), (2, 512), (3, 640), (3, 768), (4, 1024)):
    os.environ["HGMM_ESTEP_BPC"] = str(bpc)
    os.environ["HGMM_ESTEP_GRID"] = str(grid)
    reset_state(); report("device loop, E grid %d" % grid, dev_loop)
os.environ.pop("HGMM_ESTEP_BPC"); os.environ.pop("HGMM_ESTEP_GRID")
for mb in (1, 3, 4):
    os.environ["HGMM_MSTEP_BPC"] = str(mb)
    reset_state(); report("device loop, M %d workgroups per CU" % mb, dev_loop)
os.environ.pop("HGMM_MSTEP_BPC")
