set -o pipefail
timeout -k 10 900 python -u -m pytest tests -m gpu -q -x --timeout 400 --timeout-method thread > gpurun_out/r2_all_tests.log 2>&1
rc=$?
tail -5 gpurun_out/r2_all_tests.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2_bench.json 2> gpurun_out/r2_bench.err || exit 1
python -c "
import json
d=json.load(open('gpurun_out/r2_bench.json'))
print('ms/step', d['ms_per_step'], 'GB/s', d['value'])
"
grep -a "single stream" -A 14 gpurun_out/r2_bench.err | cut -c1-150
