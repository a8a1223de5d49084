set -o pipefail
timeout -k 10 600 python -u -m pytest tests/test_hip_kernels.py tests/test_fullsize_gpu.py tests/test_shared_eigh_gpu.py tests/test_e2e_gpu.py tests/test_edge_cases_gpu.py -m gpu -q -s --timeout 300 --timeout-method thread > gpurun_out/r2c_tests.log 2>&1
rc=$?
grep -a "excess\|passed\|failed\|Error\|fp32-loss perplexity" gpurun_out/r2c_tests.log | cut -c1-330 | tail -40
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2c_bench.json 2> gpurun_out/r2c_bench.err || exit 1
python -c "
import json
d=json.load(open('gpurun_out/r2c_bench.json'))
print('ms/step', d['ms_per_step'], 'GB/s', d['value'])
"
grep -a "single stream" -A 12 gpurun_out/r2c_bench.err | cut -c1-150
