#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference, which never travels to the
GPU box).  The reference's Python is imported / exec'd unmodified with small stand-in
modules for packages this image lacks and that the EM path does not actually use
(cupy -> NumPy array module, open3d, transformations); nothing from the reference
is written into this repository -- the fixtures hold only inputs and outputs.

    python tools/gen_golden.py [--only flat|bunny|hgmm|hgmm3|reg|fullcov|gmmreg]

Outputs (all small .npz, float arrays):
    bun000_xyz.npy                vertex block of data/bun000.ply (reference data file)
    flat_small_{W,G}_{diag,spherical}.npz   N=512, J=16 synthetic; 1 and 5 iterations
    flat_bunny_J{100,800}.npz     bun000, seeded init, 20 iterations tol=0 (variant W diag)
    hgmm_build_L2.npz             CPU twin buildGMMTree on bun000[::20], L=2
    hgmm_build_L3.npz             same on bun000[::40], L=3
    hgmm_reg_L2.npz               gmmTreeRegESTep moments + GMMTree.registration trace
    fullcov_flat.npz              CPU twin with n_node=J, one level (flat full-cov EM)
    gmmreg_l2.npz                 L2 GMMReg: d_rot, Gauss transform, cost/gradient, one registration
"""
import argparse
import contextlib
import importlib.util
import io
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


# ---------------------------------------------------------------------------
# stand-in modules
# ---------------------------------------------------------------------------
class _CupyRandom:
    def __init__(self):
        self._rs = np.random.RandomState(0)
        self.last_randint = None

    def seed(self, s):
        self._rs = np.random.RandomState(int(s))

    def randint(self, hi, size=None):
        self.last_randint = self._rs.randint(int(hi), size=size)
        return self.last_randint

    def choice(self, *a, **k):
        return self._rs.choice(*a, **k)


def install_stubs():
    if not hasattr(np, "infty"):
        np.infty = np.inf                      # removed in NumPy 2; used by gmm_impl.py
    cp = types.ModuleType("cupy")
    cp.get_array_module = lambda *a: np
    cp.clip = lambda a, a_min=None, a_max=None: np.clip(a, a_min, a_max)   # cupy allows a_max omitted
    cp.dot = np.dot
    cp.float32 = np.float32
    cp.asarray = np.asarray
    cp.asnumpy = np.asarray
    cp.power = np.power
    cp.copied = []

    def _copy(a):
        cp.copied.append(np.array(a, copy=True))
        return np.array(a, copy=True)
    cp.copy = _copy
    cp.random = _CupyRandom()
    cuda = types.SimpleNamespace(Stream=types.SimpleNamespace(
        null=types.SimpleNamespace(synchronize=lambda: None)))
    cp.cuda = cuda
    sys.modules["cupy"] = cp

    o3 = types.ModuleType("open3d")
    o3.__version__ = "0.0.0"
    o3.utility = types.SimpleNamespace(Vector3dVector=type("Vector3dVector", (), {}))
    o3.geometry = types.SimpleNamespace(PointCloud=type("PointCloud", (), {}))
    sys.modules["open3d"] = o3
    sys.modules["transformations"] = types.ModuleType("transformations")
    return cp


def load_module(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def read_ply_vertices(path):
    with open(path, "r") as f:
        n = None
        while True:
            line = f.readline().strip()
            if line.startswith("element vertex"):
                n = int(line.split()[-1])
            if line == "end_header":
                break
        return np.loadtxt(f, max_rows=n, dtype=np.float64)[:, :3]


@contextlib.contextmanager
def quiet():
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        yield


# ---------------------------------------------------------------------------
# flat EM goldens
# ---------------------------------------------------------------------------
def gen_flat_small(W, G):
    rs = np.random.RandomState(7)
    N, J = 512, 16
    centres = rs.rand(6, 3)
    X = (centres[rs.randint(6, size=N)] + 0.05 * rs.randn(N, 3)).astype(np.float32)
    mu0 = X[rs.choice(N, J, replace=False)].copy()
    w0 = (np.ones(J) / J).astype(np.float32)
    for variant, mod in (("W", W), ("G", G)):
        for cov_type in ("diag", "spherical"):
            if variant == "G" and cov_type == "spherical":
                continue
            cov0 = (0.1 * np.ones((J, 3) if cov_type == "diag" else (J,))).astype(np.float32)
            kw = {"cov_type": cov_type} if variant == "W" else {}
            out = {"X": X, "mu0": mu0, "w0": w0, "cov0": cov0}
            inv0 = 1 / np.sqrt(cov0)
            ll, lr = mod.e_step(X, inv0, mu0, w0, **kw)
            out["e0_ll"], out["e0_log_resp"] = np.float32(ll), lr
            with quiet():
                wts, mus, covs = mod.m_step(X, np.exp(lr), **kw)
            out["m0_w"], out["m0_mu"], out["m0_cov"] = wts, mus, covs
            for iters in (1, 5):
                with quiet():
                    inv, mu, w, cov, lls = mod.train_gmm(X, iters, 0.0, mu0.copy(), cov0.copy(),
                                                         w0.copy(), **kw)
                ll2, lr2 = mod.e_step(X, inv, mu, w, **kw)
                pre = "it%d_" % iters
                out[pre + "inv"], out[pre + "mu"], out[pre + "w"], out[pre + "cov"] = inv, mu, w, cov
                out[pre + "lls"] = np.array(lls, dtype=np.float32)
                out[pre + "log_resp"] = lr2
                out[pre + "predict"] = mod.predict(X, inv, mu, w, **kw).astype(np.int32)
            np.savez_compressed(os.path.join(OUT, "flat_small_%s_%s.npz" % (variant, cov_type)), **out)
            print("flat_small", variant, cov_type, "lls", out["it5_lls"])


def gen_flat_bunny(W, G, pts, flavours=(("W", "diag"),)):
    """BASELINE configs 1/2 (bun000.ply, J = 100 / 800, 20 iterations, tol = 0).  ("W", "diag") is the original
    fixture pair flat_bunny_J<J>.npz; the other flavours -- gmm_waymo spherical, gmmreg_gpu diag -- go to
    flat_bunny_J<J>_<variant>_<cov>.npz with 64 instead of 256 sampled responsibility rows."""
    X = pts.astype(np.float32)
    N = len(X)
    for variant, cov_type in flavours:
        mod = W if variant == "W" else G
        kw = {"cov_type": cov_type} if variant == "W" else {}
        base = (variant, cov_type) == ("W", "diag")
        for J in (100, 800):
            rs = np.random.RandomState(0)
            idx = rs.choice(N, J, replace=False)
            mu0 = X[idx].copy()
            w0 = (np.ones(J) / J).astype(np.float32)
            cov0 = (0.1 * np.ones((J, 3) if cov_type == "diag" else (J,))).astype(np.float32)
            with quiet():
                inv, mu, w, cov, lls = mod.train_gmm(X, 20, 0.0, mu0.copy(), cov0.copy(), w0.copy(), **kw)
            out = {"init_idx": idx.astype(np.int32), "lls": np.array(lls, dtype=np.float32),
                   "inv": inv, "mu": mu, "w": w, "cov": cov}
            rows = np.random.RandomState(1).choice(N, 256 if base else 64, replace=False)
            rows.sort()
            out["rows"] = rows.astype(np.int32)
            inv0 = 1 / np.sqrt(cov0)
            for tag, (a, b, c) in {"init": (inv0, mu0, w0), "final": (inv, mu, w)}.items():
                ll32, lr32 = mod.e_step(X, a, b, c, **kw)
                ll64, lr64 = mod.e_step(X.astype(np.float64), a.astype(np.float64), b.astype(np.float64),
                                        c.astype(np.float64), **kw)
                out[tag + "_ll32"], out[tag + "_ll64"] = np.float32(ll32), np.float64(ll64)
                out[tag + "_resp32_rows"] = np.exp(lr32[rows])
                out[tag + "_resp64_rows"] = np.exp(lr64[rows])
                out[tag + "_argmax32"] = lr32.argmax(1).astype(np.uint16)
                out[tag + "_argmax64"] = lr64.argmax(1).astype(np.uint16)
                r64 = np.exp(lr64)
                part = np.partition(r64, -2, axis=1)
                out[tag + "_top2gap64"] = (part[:, -1] - part[:, -2]).astype(np.float32)
                out[tag + "_noise_max_abs_dresp"] = np.float64(np.abs(np.exp(lr32.astype(np.float64)) - r64).max())
            out["predict"] = mod.predict(X, inv, mu, w, **kw).astype(np.uint16)
            name = "flat_bunny_J%d.npz" % J if base else "flat_bunny_J%d_%s_%s.npz" % (J, variant, cov_type)
            np.savez_compressed(os.path.join(OUT, name), **out)
            print(name, "lls[-1]", lls[-1], "fp32-vs-fp64 noise", out["final_noise_max_abs_dresp"])


# ---------------------------------------------------------------------------
# HGMM goldens (CPU twin)
# ---------------------------------------------------------------------------
def load_cpu_twin(cp):
    path = os.path.join(REF, "src/python/hgmm/hgmm_cupy_cpu_working.py")
    with open(path) as f:
        src = "".join(f.readlines()[:431])      # lines 432+ are the viewer demo
    ns = {"__name__": "hgmm_cpu_twin"}
    exec(compile(src, path, "exec"), ns)
    return ns


def nodes_to_arrays(nodes):
    pi = np.array([float(n.mixingCoeff) for n in nodes])
    mu = np.array([np.asarray(n.mean, dtype=np.float64).reshape(3) for n in nodes])
    cov = np.array([np.asarray(n.covar, dtype=np.float64).reshape(3, 3) for n in nodes])
    return pi, mu, cov


def moments_to_arrays(moments):
    m0 = np.array([float(m.zero) for m in moments])
    m1 = np.array([np.asarray(m.one, dtype=np.float64).reshape(3) for m in moments])
    m2 = np.array([np.asarray(m.two, dtype=np.float64).reshape(3, 3) for m in moments])
    return m0, m1, m2


def run_build(ns, cp, P, L, ls, ld):
    qs = []
    orig_ll = ns["logLikelihoodValue"]

    def rec_ll(*a):
        q = orig_ll(*a)
        qs.append(float(q))
        return q
    ns["logLikelihoodValue"] = rec_ll
    cp.copied.clear()
    try:
        with quiet():
            nodes = ns["buildGMMTree"](P, L, ls, ld)
    finally:
        ns["logLikelihoodValue"] = orig_ll
    idxs = np.array(cp.random.last_randint, copy=True)
    cur_levels = [np.array(c, dtype=np.int32) for c in cp.copied[:L]]
    # iterations per level: prevQ resets to 0 per level; reconstruct from the stop rule
    iters, prev, cnt = [], 0.0, 0
    for q in qs:
        cnt += 1
        if abs(q - prev) < ls:
            iters.append(cnt)
            cnt, prev = 0, 0.0
        else:
            prev = q
    return nodes, idxs, np.array(qs), np.array(iters, dtype=np.int32), cur_levels


def gen_hgmm_build(ns, cp, pts, L, stride, name):
    import warnings
    warnings.simplefilter("ignore")
    P = np.ascontiguousarray(pts[::stride])
    ls, ld = 80.0, 1.0e-4
    nodes, idxs, qs, iters, cur_levels = run_build(ns, cp, P, L, ls, ld)
    pi, mu, cov = nodes_to_arrays(nodes)
    out = {"points": P, "L": np.int32(L), "ls": np.float64(ls), "ld": np.float64(ld),
           "sig2": np.float64(0.00034), "init_idx": idxs.astype(np.int32), "q_trace": qs,
           "iters_per_level": iters, "pi": pi, "mu": mu, "cov": cov}
    for l, c in enumerate(cur_levels):
        out["current_idx_L%d" % l] = c
    # gamma for 64 sampled points at the final tree, level 0 (children of the root)
    moms = None
    np.savez_compressed(os.path.join(OUT, name), **out)
    print(name, "N", len(P), "iters", iters, "q_last", qs[-1], "dead nodes", int((pi == 0).sum()))
    return nodes, P


def gen_hgmm_reg(ns, cp, nodes, P, L):
    import warnings
    warnings.simplefilter("ignore")
    lc = 0.01
    pi, mu, cov = nodes_to_arrays(nodes)
    out = {"points": P, "L": np.int32(L), "lambda_c": np.float64(lc), "pi": pi, "mu": mu, "cov": cov}
    for deg in (10, 30):
        th = np.deg2rad(float(deg))
        Rz = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
        target = P @ Rz.T
        with quiet():
            moms = ns["gmmTreeRegESTep"](target, nodes, L, lc)
        m0, m1, m2 = moments_to_arrays(moms)
        tag = "rot%d_" % deg
        out[tag + "target"] = target
        out[tag + "m0"], out[tag + "m1"], out[tag + "m2"] = m0, m1, m2
        # full registration loop, recording tf per iteration through the callback hook
        gt = ns["GMMTree"](None, tree_level=L, lambda_c=lc)
        gt._source = P
        gt._nodes = nodes
        trace = []
        gt.set_callbacks([lambda tf: trace.append((np.array(tf.rot), np.array(tf.t)))])
        with quiet():
            res = gt.registration(target, 5, 1.0e-4)
        out[tag + "iter_rot"] = np.array([r for r, _ in trace])
        out[tag + "iter_t"] = np.array([t for _, t in trace])
        out[tag + "final_rot"] = np.array(res.transformation.rot)
        out[tag + "final_t"] = np.array(res.transformation.t)
        out[tag + "final_q"] = np.asarray(res.q, dtype=np.float64)
        print("reg", deg, "iters", len(trace), "q", res.q)
    np.savez_compressed(os.path.join(OUT, "hgmm_reg_L2.npz"), **out)


def gen_fullcov(ns, cp, pts):
    import warnings
    warnings.simplefilter("ignore")
    rs = np.random.RandomState(3)
    P = np.ascontiguousarray(pts[rs.choice(len(pts), 1000, replace=False)])
    out = {"points": P}
    for J in (8, 32):
        ns["n_node"] = J
        try:
            # one level with n_node = J  ==  flat full-covariance EM over J components
            nodes, idxs, qs, iters, cur = run_build(ns, cp, P, 1, 80.0, 1.0e-4)
        finally:
            ns["n_node"] = 8
        pi, mu, cov = nodes_to_arrays(nodes)
        tag = "J%d_" % J
        out[tag + "init_idx"] = idxs.astype(np.int32)
        out[tag + "q_trace"] = qs
        out[tag + "pi"], out[tag + "mu"], out[tag + "cov"] = pi, mu, cov
        out[tag + "current_idx"] = cur[0]
        print("fullcov J", J, "iters", len(qs), "q", qs[-1])
    np.savez_compressed(os.path.join(OUT, "fullcov_flat.npz"), **out)


# ---------------------------------------------------------------------------
# L2 GMMReg (gmmreg_gpu) goldens: cost function, gradient, Gauss transform, one registration
# ---------------------------------------------------------------------------
def _quaternion_matrix(q):
    """Stand-in for the absent third-party `transformations.quaternion_matrix` (w, x, y, z order,
    normalising; 4x4 homogeneous) -- standard formula, written for this tool."""
    q = np.array(q, dtype=np.float64, copy=True)
    n = np.dot(q, q)
    if n < np.finfo(float).eps * 4.0:
        return np.identity(4)
    w, x, y, z = q * np.sqrt(2.0 / n)
    m = np.identity(4)
    m[0, 0] = 1.0 - y * y - z * z; m[0, 1] = x * y - z * w;       m[0, 2] = x * z + y * w
    m[1, 0] = x * y + z * w;       m[1, 1] = 1.0 - x * x - z * z; m[1, 2] = y * z - x * w
    m[2, 0] = x * z - y * w;       m[2, 1] = y * z + x * w;       m[2, 2] = 1.0 - x * x - y * y
    return m


def gen_gmmreg(pts):
    import warnings
    warnings.simplefilter("ignore")
    sys.modules["transformations"].quaternion_matrix = _quaternion_matrix
    sys.modules["thundersvm"] = types.ModuleType("thundersvm")
    refdir = os.path.join(REF, "src/python/gmmreg_gpu")
    sys.path.insert(0, refdir)
    for m in ("gmm", "gmm_impl", "transforms", "so", "cost_functions", "gmmreg"):
        sys.modules.pop(m, None)
    try:
        import so, transforms, cost_functions
        rs = np.random.RandomState(11)
        out = {}
        qs = rs.randn(5, 4)
        qs[0] = [1, 0, 0, 0]
        out["q"] = qs
        out["d_rot"] = np.array([so.diff_rot_from_quaternion(q) for q in qs])
        out["rot"] = np.array([_quaternion_matrix(q)[:3, :3] for q in qs])
        mu_s, mu_t = rs.rand(20, 3), rs.rand(25, 3)
        phi_s, phi_t = rs.rand(20) + 0.1, rs.rand(25) + 0.1
        sigma = 0.37
        f, g = cost_functions.compute_l2_dist(mu_s, phi_s, mu_t, phi_t, sigma)
        out.update(mu_s=mu_s, mu_t=mu_t, phi_s=phi_s, phi_t=phi_t, sigma=np.float64(sigma), l2_f=np.float64(f), l2_g=g)
        gt = transforms.GaussTransform(mu_t, 0.5)
        out["gt_1d"] = gt.compute(mu_s, phi_t)
        out["gt_2d"] = gt.compute(mu_s, phi_t * mu_t.T)
        cfn = cost_functions.RigidCostFunction()
        thetas = np.c_[qs, rs.randn(5, 3) * 0.1]
        out["theta"] = thetas
        fs, gs = zip(*[cfn(th, mu_s, phi_s, mu_t, phi_t, sigma) for th in thetas])
        out["cost_f"], out["cost_g"] = np.array(fs), np.array(gs)
        # one full registration with the reference's own (NumPy-backed) GMM features
        with quiet():
            import gmmreg
        src = np.ascontiguousarray(pts[::20])
        th = np.deg2rad(10.0)
        Rz = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
        tgt = src @ Rz.T + np.array([0.01, 0.0, -0.005])
        with quiet():
            reg = gmmreg.RigidGMMReg(src, n_gmm_components=30)
        feats = []
        orig_compute = reg._feature_gen.compute

        def rec_compute(data):
            r = orig_compute(data)
            feats.append((np.array(r[0]), np.array(r[1])))
            return r
        reg._feature_gen.compute = rec_compute
        with quiet():
            tfm = reg.registration(tgt)
        out.update(reg_source=src, reg_target=tgt, reg_sigma0=np.float64(
            np.power(np.linalg.det(np.dot((src - src.mean(0)).T, src - src.mean(0)) / (len(src) - 1)), 1.0 / 6.0)),
            reg_mu_target=feats[0][0], reg_phi_target=feats[0][1], reg_mu_source=feats[1][0], reg_phi_source=feats[1][1],
            reg_rot=np.array(tfm.rot), reg_t=np.array(tfm.t))
        np.savez_compressed(os.path.join(OUT, "gmmreg_l2.npz"), **out)
        print("gmmreg: f", fs[:2], "reg rot[0]", tfm.rot[0], "t", tfm.t)
    finally:
        sys.path.remove(refdir)


def gen_waymo_frames():
    """The five Waymo LIDAR frames the reference ships (src/python/gmmreg_gpu/waymo{1,2,5,10,50}.pcd, binary PCD,
    xyz float32) as plain arrays -- parsed HERE with a few lines of NumPy, independently of the package's
    pointcloud_io reader, which tests/test_pointcloud_io_cpu.py then checks against them; they are also the frames
    of the streaming-harness parity test (run_gmm_waymo_gpu.py:32-61)."""
    out = {}
    for k in (1, 2, 5, 10, 50):
        raw = open(os.path.join(REF, "src/python/gmmreg_gpu/waymo%d.pcd" % k), "rb").read()
        head, _, body = raw.partition(b"DATA binary\n")
        fields = dict(l.split(None, 1) for l in head.decode("ascii").splitlines() if l and not l.startswith("#"))
        assert fields["FIELDS"].split() == ["x", "y", "z"] and fields["SIZE"].split() == ["4", "4", "4"]
        n = int(fields["POINTS"])
        out["waymo%d" % k] = np.frombuffer(body, dtype="<f4", count=3 * n).reshape(n, 3).copy()
        print("waymo%d" % k, out["waymo%d" % k].shape)
    np.savez_compressed(os.path.join(OUT, "waymo_frames.npz"), **out)


def gen_scan_pair():
    """Second Stanford scan + the ground-truth scan poses of data/bun.conf (data fixtures for the
    end-to-end registration accuracy test): bun045 vertices (float32) and, per scan, translation
    (3) + quaternion (x, y, z, w) exactly as listed in the file."""
    pts = read_ply_vertices(os.path.join(REF, "data/bun045.ply"))
    np.save(os.path.join(OUT, "bun045_xyz.npy"), pts.astype(np.float32))
    names, poses = [], []
    with open(os.path.join(REF, "data/bun.conf")) as f:
        for line in f:
            tok = line.split()
            if tok and tok[0] == "bmesh":
                names.append(tok[1])
                poses.append([float(v) for v in tok[2:9]])
    np.savez(os.path.join(OUT, "bun_conf.npz"), names=np.array(names), poses=np.array(poses))
    print("scan pair: bun045", pts.shape, "poses", len(names))


def gen_kmeans(G, pts):
    """KMeans initialiser: outputs of the reference's own init_gmm_params (gmmreg_gpu/gmm_impl.py:18-24,
    i.e. scikit-learn's KMeans(k, random_state=1, max_iter=50, n_init=1)) plus the estimator's
    labels / iteration count / seeds for the same call, and one Lloyd run from a hand-made init
    that leaves two clusters empty (exercises the relocation rule)."""
    import warnings
    import sklearn
    from sklearn.cluster import KMeans, kmeans_plusplus
    warnings.simplefilter("ignore")
    rs = np.random.RandomState(11)
    bun = pts.astype(np.float32).astype(np.float64)
    blobs = rs.rand(12, 3)[rs.randint(12, size=3000)] + 0.02 * rs.randn(3000, 3)
    cases = {"uniform": (rs.rand(2000, 3), 16), "blobs": (blobs, 20), "bunny": (bun[::10], 50)}
    out = {"sklearn_version": np.array(sklearn.__version__)}
    for name, (X, k) in cases.items():
        means, weights = G.init_gmm_params(X, k)
        km = KMeans(n_clusters=k, random_state=1, max_iter=50, n_init=1).fit(X)
        assert np.allclose(km.cluster_centers_, means, rtol=0, atol=1e-12)    # thread-order noise only
        _, idx = kmeans_plusplus(X - X.mean(axis=0), k, random_state=1)
        if name != "bunny":
            out[name + "_X"] = X
        out[name + "_k"] = np.array(k)
        out[name + "_centres"] = means
        out[name + "_weights"] = weights
        out[name + "_labels"] = km.labels_.astype(np.int32)
        out[name + "_n_iter"] = np.array(km.n_iter_)
        out[name + "_inertia"] = np.array(km.inertia_)
        out[name + "_init_indices"] = idx.astype(np.int64)
        print("kmeans", name, "k", k, "n_iter", km.n_iter_, "inertia", km.inertia_)
    X = rs.rand(500, 3)
    init = X[rs.choice(500, 6, replace=False)].copy()
    init[2] = [50, 50, 50]
    init[4] = [-40, 3, 3]
    km = KMeans(n_clusters=6, init=init, n_init=1, max_iter=50).fit(X)
    out.update(reloc_X=X, reloc_init=init, reloc_centres=km.cluster_centers_, reloc_labels=km.labels_.astype(np.int32),
               reloc_n_iter=np.array(km.n_iter_), reloc_inertia=np.array(km.inertia_))
    np.savez_compressed(os.path.join(OUT, "kmeans_init.npz"), **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    cp = install_stubs()
    pts = read_ply_vertices(os.path.join(REF, "data/bun000.ply"))
    assert pts.shape == (40256, 3)
    np.save(os.path.join(OUT, "bun000_xyz.npy"), pts.astype(np.float32))
    want = lambda k: args.only in (None, k)
    if want("flat") or want("bunny") or want("bunny_flavours"):
        W = load_module("ref_gmm_impl_W", os.path.join(REF, "src/python/gmm_waymo/src/gmm_impl.py"))
        G = load_module("ref_gmm_impl_G", os.path.join(REF, "src/python/gmmreg_gpu/gmm_impl.py"))
        if want("flat"):
            gen_flat_small(W, G)
        if want("bunny"):
            gen_flat_bunny(W, G, pts)
        if want("bunny_flavours"):
            gen_flat_bunny(W, G, pts, flavours=(("W", "spherical"), ("G", "diag")))
    if want("scans"):
        gen_scan_pair()
    if want("frames"):
        gen_waymo_frames()
    if want("kmeans"):
        G = load_module("ref_gmm_impl_G", os.path.join(REF, "src/python/gmmreg_gpu/gmm_impl.py"))
        gen_kmeans(G, pts)
    if want("gmmreg"):
        gen_gmmreg(pts)
    if want("hgmm") or want("reg") or want("hgmm3") or want("fullcov"):
        ns = load_cpu_twin(cp)
        if want("hgmm") or want("reg"):
            nodes, P = gen_hgmm_build(ns, cp, pts, 2, 20, "hgmm_build_L2.npz")
            if want("reg"):
                gen_hgmm_reg(ns, cp, nodes, P, 2)
        if want("fullcov"):
            gen_fullcov(ns, cp, pts)
        if want("hgmm3"):
            gen_hgmm_build(ns, cp, pts, 3, 40, "hgmm_build_L3.npz")


if __name__ == "__main__":
    main()
