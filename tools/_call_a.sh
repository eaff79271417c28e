set -u
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kmeans_gpu.py tests/test_dropin_gpu.py tests/test_gmmreg_gpu.py -m gpu -q -x --timeout 600 2>&1 | tail -5
echo "--- fused"; timeout 120 python tools/kmpp_prof.py 2>&1 | tail -4
echo "--- two launches"; HGMM_KMPP_TWO_LAUNCHES=1 timeout 120 python tools/kmpp_prof.py 2>&1 | tail -4
echo "--- fused N=1e5"; KMPP_N=100000 timeout 120 python tools/kmpp_prof.py 2>&1 | tail -2
echo "--- two N=1e5"; KMPP_N=100000 HGMM_KMPP_TWO_LAUNCHES=1 timeout 120 python tools/kmpp_prof.py 2>&1 | tail -2
python - <<'PY'
import os, sys
import numpy as np
sys.path.insert(0, '.')
import hgmm_amd
ctx = hgmm_amd.Context(0)
for n, k in ((1000000, 800), (123457, 300), (5000, 17), (300, 2)):
    X = np.random.RandomState(n).rand(n, 3); ctx.set_points(X - X.mean(0))
    trials = 2 + int(np.log(k)); rand = np.random.RandomState(1).uniform(size=(k - 1, trials))
    os.environ.pop("HGMM_KMPP_TWO_LAUNCHES", None)
    a = ctx.kmeans_plusplus(k, 7 % n, rand)
    os.environ["HGMM_KMPP_TWO_LAUNCHES"] = "1"
    b = ctx.kmeans_plusplus(k, 7 % n, rand)
    print(n, k, "ids equal", np.array_equal(a[0], b[0]), "centres equal", np.array_equal(a[1], b[1]))
PY
