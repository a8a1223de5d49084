#!/bin/bash
# Forward microbenchmark (28672 x 8192, rank 1024) plain and under rocprofv3 kernel-trace; writes gpurun_out/fwd_*.
set -e
mkdir -p gpurun_out/fwd_prof
export TMPDIR=/tmp
timeout -k 10 300 python tools/forward_bench.py --batches ${BATCHES:-1,16,64,256,512,1024,4096} > gpurun_out/fwd_bench.jsonl 2> gpurun_out/fwd_bench.err
timeout -k 10 300 python tools/forward_bench.py --lora-tile 0 --batches ${BATCHES:-1,16,64,256,512,1024,4096} > gpurun_out/fwd_bench_bf16.jsonl 2>> gpurun_out/fwd_bench.err
for b in ${PBATCHES:-1 16 256}; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fwd_prof -o fwd_b$b -- python tools/forward_bench.py --batches $b --iters 20 > gpurun_out/fwd_prof_b$b.log 2>&1
done
python - <<'PY'
import csv, glob, json
for f in sorted(glob.glob("gpurun_out/fwd_prof/*kernel_stats.csv")):
    print(f)
    for r in csv.DictReader(open(f)):
        if "linear24" in r["Name"] or "unpack24" in r["Name"]:
            print("   %-60s calls %4s avg %10.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
for f in ("gpurun_out/fwd_bench.jsonl", "gpurun_out/fwd_bench_bf16.jsonl"):
    for l in open(f):
        d = json.loads(l)
        print(d["adapters"], d["batch"], "emulated", d["emulated_ms"], "fused", d["fused_ms"], "x", d["speedup"], "GB/s", d["fused_GBps_packed"], "TF", d["fused_TFLOPs"])
PY
