set -u
python - <<'PY'
import os, sys
import numpy as np
sys.path.insert(0, '.')
import hgmm_amd
ctx = hgmm_amd.Context(0)
for n, k, trials in ((60000, 1200, None), (4000, 50, 16), (4000, 50, 1), (70000, 3, 3), (1100000, 64, 5)):
    X = np.random.RandomState(n).rand(n, 3); ctx.set_points(X - X.mean(0))
    t = trials or (2 + int(np.log(k))); rand = np.random.RandomState(1).uniform(size=(k - 1, t))
    os.environ.pop("HGMM_KMPP_TWO_LAUNCHES", None)
    a = ctx.kmeans_plusplus(k, 7 % n, rand)
    os.environ["HGMM_KMPP_TWO_LAUNCHES"] = "1"
    b = ctx.kmeans_plusplus(k, 7 % n, rand)
    os.environ["HGMM_KMPP_UNFUSED"] = "1"
    c = ctx.kmeans_plusplus(k, 7 % n, rand)
    os.environ.pop("HGMM_KMPP_UNFUSED")
    print(n, k, t, "fused==two-launch", np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), "==four-kernel form", np.array_equal(a[0], c[0]))
PY
