#!/usr/bin/env python3
"""Where the time of a BATCH of scan pairs goes (registration_gmmtree_batch: bench.py --mode pairs --batch B):
per batch size B the wall time of the four C calls -- upload of the sources, forest build, upload of the targets,
batched registration -- and the pairs/s they add up to, next to the serial call on one pair.

    python tools/pair_batch_probe.py [--f32] [--device-solve] [B ...]
        --f32: float32 scans -> the stop rule's pdfs in float32;  --device-solve: the registration loop on the device alone
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import hgmm_amd  # noqa: E402
from hgmm_amd.hgmm.hgmm_gpu import n_total_nodes  # noqa: E402


def main():
    f32 = "--f32" in sys.argv
    sizes = [int(v) for v in sys.argv[1:] if not v.startswith("--")] or [1, 2, 4, 8, 16, 32, 64]
    ctx = hgmm_amd.Context(0)
    if "--device-solve" in sys.argv:
        ctx.config_set("reg_device_solve", 1)
    source, pairs = bench.scan_pairs(0)
    if f32:
        source = source.astype(np.float32)
        ctx.tree_set_precision(np.float32)                 # (the batched calls below; the serial mirror follows the source's type)
    kw = bench.PAIR_KW
    L = kw["tree_level"]
    T = n_total_nodes(L)
    idx = np.random.RandomState(72).randint(T, size=T)
    for _ in range(3):
        bench.register_pair(ctx, source, pairs[0][0])
    t0 = time.perf_counter()
    for k in range(8):
        bench.register_pair(ctx, source, pairs[k % len(pairs)][0])
    ctx.synchronize()
    serial = (time.perf_counter() - t0) / 8
    print("serial registration_gmmtree: %.3f ms per pair = %.0f pairs/s" % (serial * 1e3, 1 / serial))
    for B in sizes:
        srcs = [source] * B
        tgts = [pairs[k % len(pairs)][0].astype(np.float32) if f32 else pairs[k % len(pairs)][0] for k in range(B)]
        rows = []
        for rep in range(6):
            t = [time.perf_counter()]
            arrs = ctx.set_points_batch(srcs)
            t.append(time.perf_counter())
            init = np.stack([a[idx] for a in arrs])
            t.append(time.perf_counter())
            _, iters, _ = ctx.tree_build_batch([len(a) for a in arrs], L, kw["ls"], 1e-4, init, kw["sig2"], want_tables=False)
            t.append(time.perf_counter())
            ctx.tree_set_targets_batch(tgts)
            t.append(time.perf_counter())
            rot, tt, it_reg, q, status, _ = ctx.tree_register_batch(np.tile(np.eye(3), (B, 1, 1)), np.zeros((B, 3)), 1.0,
                                                                    kw["lambda_c"], bench.PAIR_MAXITER, bench.PAIR_TOL)
            t.append(time.perf_counter())
            rows.append(np.diff(t))
        r = np.median(np.array(rows[1:]), axis=0) * 1e3
        tot = r.sum()
        print("B = %3d: sources up %.2f | init draw %.2f | build %.2f | targets up %.2f | register %.2f | total %.2f ms = %.3f ms "
              "per pair = %.0f pairs/s   (build iterations %s, registration iterations %s)"
              % (B, r[0], r[1], r[2], r[3], r[4], tot, tot / B, 1e3 * B / tot, iters[0].tolist(), sorted(set(int(v) for v in it_reg))))
    ctx.close()


if __name__ == "__main__":
    main()
