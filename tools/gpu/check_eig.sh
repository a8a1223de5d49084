set -o pipefail
timeout -k 10 600 python -u -m pytest tests/test_f32gemm_gpu.py tests/test_eigh_gpu.py tests/test_shared_eigh_gpu.py -m gpu -q -s --timeout 300 --timeout-method thread > gpurun_out/r2d_tests.log 2>&1
rc=$?
grep -a "passed\|failed\|Error\|assert" gpurun_out/r2d_tests.log | cut -c1-300 | tail -40
exit $rc
