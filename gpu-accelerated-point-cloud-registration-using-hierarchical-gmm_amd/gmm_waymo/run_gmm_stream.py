"""Headless version of the reference's streaming demo drivers
(``src/python/gmm_waymo/src/run_gmm_waymo_gpu.py:32-61`` / ``run_gmm_static.py:35-49``): refit the
mixture every ``fit_every`` frames, label every frame with ``predict``; no viewer.  Returns the
per-frame labels and the frames/second the reference plots in its FPS chart (README.md:246).

    python -m hgmm_amd.gmm_waymo.run_gmm_stream frame1.pcd frame2.pcd ... [--components 50]
"""
import argparse
import collections
import time

import numpy as np

from .gmm import GMM_GPU
from .gmm_impl import asarray
from ..pointcloud_io import read_point_cloud, voxel_down_sample


def run_stream(frames, n_components=50, max_iter=50, cov_type='spherical', fit_every=10, tol=1e-4,
               voxel_size=None, seed=0, verbose=False):
    """frames: iterable of [N,3] arrays.  -> dict(labels=[...], fps=float, fit_s=[...], models=[...]):
    ``models[k]`` = (means, weights, covariances, inv_covs) of the k-th refit, i.e. the parameters frame i was
    labelled with are ``models[i // fit_every]``."""
    gmm = GMM_GPU(n_gmm_components=n_components, max_iter=max_iter, tol=tol, cov_type=cov_type)
    gmm.init()
    gmm._clf._verbose = verbose
    labels, fit_s, models = [], [], []
    # every frame is uploaded ONCE and stays resident (like the reference's CuPy frames, waymoutils.py: the H2D copy per
    # frame) for the refit and the predict that use it; the last `fit_every` frames are kept in HBM
    resident = collections.deque(maxlen=max(int(fit_every), 1))
    np.random.seed(seed)
    t0 = time.perf_counter()
    n = 0
    for i, pts in enumerate(frames):
        pts = np.asarray(pts)
        if voxel_size:
            pts = voxel_down_sample(pts, voxel_size)
        dev = asarray(pts.astype(np.float32))
        resident.append(dev)
        if i % fit_every == 0:
            t1 = time.perf_counter()
            clf = gmm._clf
            clf.fit(dev, init=clf._init_params(pts))               # (the initialiser samples from the host array)
            models.append(tuple(np.array(a) for a in (clf.means_, clf.weights_, clf.covariances_, clf.inv_covs)))
            fit_s.append(time.perf_counter() - t1)
        labels.append(np.asarray(gmm.predict(dev)).astype(np.int64))    # = cupy.asnumpy(gmm_idxs), run_gmm_waymo_gpu.py:55
        n += 1
    dt = time.perf_counter() - t0
    return {"labels": labels, "fps": n / dt if dt > 0 else float("inf"), "fit_s": fit_s, "frames": n,
            "models": models}


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("files", nargs="+")
    ap.add_argument("--components", type=int, default=50)
    ap.add_argument("--max-iter", type=int, default=50)
    ap.add_argument("--cov-type", default="spherical")
    ap.add_argument("--fit-every", type=int, default=10)
    ap.add_argument("--voxel", type=float, default=None)
    args = ap.parse_args(argv)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        res = run_stream((read_point_cloud(f) for f in args.files), args.components, args.max_iter,
                         args.cov_type, args.fit_every, voxel_size=args.voxel)
    print("frames %d  fps %.2f  mean fit %.4f s" % (res["frames"], res["fps"], float(np.mean(res["fit_s"]))))


if __name__ == "__main__":
    main()
