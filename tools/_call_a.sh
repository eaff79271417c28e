set -u
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_flat_gpu.py -m gpu -q -x --timeout 600 -k "device_array or async_estep or function_level or dropin or module" 2>&1 | tail -30
timeout 300 python bench.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print(json.dumps(d['materialised_iteration'], indent=1)[:3000]); print(d['value'], d['roofline']['frac'])
"
