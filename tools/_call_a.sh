set -u
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 2>&1 | tail -6
timeout 300 python bench.py 2>/dev/null > gpurun_out/bench_try.json; python -c "
import json
d=json.loads(open('gpurun_out/bench_try.json').readline())
m=d['materialised_iteration']; r=d['roofline']
print('value',d['value'],'frac',r['frac'],'cold',r['cold_frac'],'stream',r['unsynchronised_stream_avg_ms'],r['unsynchronised_stream_frac'])
print('api loop',m['it_per_s'],m['ms_per_iteration'],m['kernel_ms_per_iteration'],'host loop',m['host_array_loop']['it_per_s'],m['device_and_host_array_loops_bitwise_equal'])
print('hgmm',d['hgmm']['build_ms'],'tree1M',d['tree_1M']['build_ms'],d['tree_1M']['roofline']['executed_fraction'],d['tree_1M']['roofline']['frac'])
"
