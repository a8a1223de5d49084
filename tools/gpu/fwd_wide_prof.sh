#!/bin/bash
# kernel stats of the forced-tiled forward at one batch size
set -o pipefail
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
export SLIM_FWD_WIDE=${WIDE:-1} SLIM_PACKED_LARGE_BATCH=1000000
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fwdprof -o fwd -- python3 $GRAFT_REPO_ROOT/tools/forward_bench.py --batches ${BATCH:-4096} > $OUT/fwd_prof.log 2>&1 || exit 1
cp $(find /tmp/fwdprof -name '*kernel_stats.csv' | head -1) $OUT/fwd_prof_kernel_stats.csv
