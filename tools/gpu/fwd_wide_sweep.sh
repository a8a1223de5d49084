#!/bin/bash
# forced-tiled forward at mid / large batches, once per wide-tile choice (SLIM_FWD_WIDE)
set -o pipefail
mkdir -p gpurun_out
for w in ${WIDE_LIST:-0 1 2 3}; do
  echo "== SLIM_FWD_WIDE=$w" >> gpurun_out/fwd_wide.log
  SLIM_FWD_WIDE=$w SLIM_PACKED_LARGE_BATCH=1000000 timeout -k 10 200 python tools/forward_bench.py --batches ${BATCHES:-256,512,1024,4096} >> gpurun_out/fwd_wide.log 2>&1 || exit 1
done
