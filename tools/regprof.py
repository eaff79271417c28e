import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import hgmm_amd
from hgmm_amd.hgmm.hgmm_gpu import GMMTree
ctx = hgmm_amd.Context(0)
P = np.load("/root/repo/tests/golden/bun000_xyz.npy").astype(np.float64)
th = np.deg2rad(10.0)
rz = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
target = P @ rz.T + np.array([0.005, -0.003, 0.002])
gt = GMMTree(P, tree_level=3, lambda_c=0.01, ls=80, sig2=0.00034, ctx=ctx)
gt.registration(target, maxiter=20, tol=1e-4)
ctx.tree_set_nodes(3, gt._mixingCoeff, gt._mean, gt._covar)
ctx.tree_set_target(target)
R, t = np.eye(3), np.zeros(3)
for name, fn in (("tree_reg_normal", lambda: ctx.tree_reg_normal(R, t, 1.0, 0.01)),):
    fn()
    ctx.profile_reset(); ctx.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(200): fn()
    dt = (time.perf_counter() - t0) / 200
    ctx.profile_enable(False)
    ms, n = ctx.profile_get("tree_reg")
    print(name, "wall %.1f us per call, estep kernel %.1f us" % (dt * 1e6, ms / n * 1e3))
ata, atb, btb = ctx.tree_reg_normal(R, t, 1.0, 0.01)
import timeit
print("eigvalsh %.1f us" % (timeit.timeit(lambda: np.linalg.eigvalsh(ata), number=2000) / 2000 * 1e6))
print("solve %.1f us" % (timeit.timeit(lambda: np.linalg.solve(ata, atb), number=2000) / 2000 * 1e6))
from hgmm_amd.hgmm.hgmm_gpu import twist_mul, RigidTransformation
x = np.linalg.solve(ata, atb)
print("twist_mul %.1f us" % (timeit.timeit(lambda: twist_mul(x, R, t), number=2000) / 2000 * 1e6))
tf = RigidTransformation(R, t)
print("inverse %.1f us" % (timeit.timeit(lambda: tf.inverse(), number=2000) / 2000 * 1e6))
t0 = time.perf_counter(); gt2 = GMMTree(P, tree_level=3, lambda_c=0.01, ls=80, sig2=0.00034, ctx=ctx); print("build %.2f ms" % ((time.perf_counter()-t0)*1e3))
t0 = time.perf_counter(); ctx.tree_set_target(target); print("set_target %.2f ms" % ((time.perf_counter()-t0)*1e3))
t0 = time.perf_counter(); ctx.tree_set_nodes(3, gt._mixingCoeff, gt._mean, gt._covar); print("set_nodes %.2f ms" % ((time.perf_counter()-t0)*1e3))
