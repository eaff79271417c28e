set -u
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_tree_gpu.py tests/test_dropin_gpu.py tests/test_integration_snippet_gpu.py -m gpu -q -x --timeout 600 2>&1 | tail -3
timeout 300 python bench.py 2>/dev/null > gpurun_out/bench_try.json; python -c "
import json
d=json.loads(open('gpurun_out/bench_try.json').readline())
print('reg',json.dumps(d['registration']))
print('hgmm',d['hgmm']['build_ms'],'t1M',d['tree_1M']['build_ms'], 'fullcov', d['fullcov']['ms_per_iteration'], d['fullcov'].get('marginal_ms_per_iteration'))
k=d['kmeans_init'] if 'kmeans_init' in d else d['kmeans']; print('kmeans', k['fit_ms_warm'], k['seeding_ms_warm'], k['bun000_k100_fit_ms'])
print('value', d['value'], d['roofline']['frac'], d['materialised_iteration']['it_per_s'])
"
