#!/bin/bash
set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/gram_prof
timeout -k 10 200 python tools/gram_bench.py > gpurun_out/gram_bench.jsonl 2> gpurun_out/gram_bench.err
cat gpurun_out/gram_bench.jsonl
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gram_prof -o gram -- python tools/gram_bench.py --iters 5 > gpurun_out/gram_prof.log 2>&1
python - <<'PY'
import csv
for r in csv.DictReader(open("gpurun_out/gram_prof/gram_kernel_stats.csv")):
    if float(r["Percentage"]) > 1.0:
        print("%-70s calls %4s avg %10.1f us  %5s %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
