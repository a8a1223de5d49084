for e in 1 2 3; do
  echo "== SLIM_FIT_COND_E=$e"
  SLIM_FIT_COND_E=$e timeout -k 10 300 python -u -m pytest tests/test_fullsize_gpu.py -m gpu -q -s -k "fit_within" --timeout 300 --timeout-method thread 2>&1 | grep -a "excess\|passed\|failed" | cut -c1-200
  SLIM_FIT_COND_E=$e timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms/step', d['ms_per_step'])"
done
