#!/bin/bash
# 2-rank rehearsal of bench.py on the one GPU of a test box (gloo, host-staged collectives: the numbers mean nothing,
# the code path is the one N > 1 runs), then the plain N = 1 line.
set -e
mkdir -p gpurun_out
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29631 \
    bench.py --gpus 2 --steps 2 --warmup 1 --rehearse-backend gloo > gpurun_out/rehearse2.json 2> gpurun_out/rehearse2.err
tail -c 1500 gpurun_out/rehearse2.json
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-e2e --no-cpu-baseline > gpurun_out/bench1.json 2> gpurun_out/bench1.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench1.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "n_gpus")}, d["config"]["partition"])
d = json.loads(open("gpurun_out/rehearse2.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "n_gpus")}, d["config"]["partition"], d["config"]["partition_trial_ms"])
PY
