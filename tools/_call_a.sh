set -u
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fullcov_gpu.py tests/test_tree_gpu.py -m gpu -q -x --timeout 600 2>&1 | tail -4
echo "look-ahead"; timeout 100 python tools/fullcov_prof.py 20 2>&1 | tail -1; timeout 100 python tools/fullcov_prof.py 20 2>&1 | tail -1
echo "sync"; HGMM_FULLCOV_SYNC=1 timeout 100 python tools/fullcov_prof.py 20 2>&1 | tail -1
python - <<'PY'
import os, sys
import numpy as np
sys.path.insert(0, '.')
import hgmm_amd
ctx = hgmm_amd.Context(0)
for N, J, it, ls in ((200000, 800, 7, 1e-30), (5032, 100, 40, 1.0), (777, 17, 40, 1.0), (3000, 40, 1, 1.0), (3000, 40, 2, 1e-30)):
    P = np.random.RandomState(N).rand(N, 3); idx = np.random.RandomState(J).choice(N, J, replace=False)
    ctx.set_points(P)
    os.environ.pop("HGMM_FULLCOV_SYNC", None)
    a = ctx.fullcov_fit(J, ls, 1e-4, P[idx], 0.003, it)
    os.environ["HGMM_FULLCOV_SYNC"] = "1"
    b = ctx.fullcov_fit(J, ls, 1e-4, P[idx], 0.003, it)
    print(N, J, "iterations", len(a[4]), len(b[4]), "bitwise", all(np.array_equal(x, y) for x, y in zip(a, b)))
PY
