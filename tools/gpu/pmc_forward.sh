#!/bin/bash
# counters (default: SQ; COUNTERS overrides) of the tiled forward kernels at one batch size; FIRST names the first counter
set -e
export TMPDIR=/tmp
rm -rf gpurun_out/fwd_pmc
mkdir -p gpurun_out/fwd_pmc
SLIM_PACKED_LARGE_BATCH=1000000 timeout -s KILL 120 rocprofv3 --pmc ${COUNTERS:-SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS} --output-format csv -d gpurun_out/fwd_pmc -o pmc -- python tools/forward_bench.py --batches ${B:-4096} --iters 2 > gpurun_out/fwd_pmc.log 2>&1
python - <<'PY'
import csv, glob, collections, os
FIRST = os.environ.get('COUNTERS', 'SQ_WAVE_CYCLES').split()[0]
f = glob.glob("gpurun_out/fwd_pmc/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:40] + " grid " + r["Grid_Size"]
    if "linear24" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    if r["Counter_Name"] == FIRST: n[k] += 1
for k, v in acc.items():
    print(k, "launches", n[k])
    for c, x in v.items(): print("   %-28s %.4g per launch" % (c, x / max(n[k], 1)))
PY
