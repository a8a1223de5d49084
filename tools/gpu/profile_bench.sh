# Round-3 measurement run on the GPU box: the default bench line, then rocprofv3 kernel statistics and two counter passes
# of the same workload on a single stream fed by one thread (profilers do not tolerate dispatches from several threads).
set -o pipefail
mkdir -p gpurun_out/prof
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 400 python bench.py > gpurun_out/prof/r3_bench.json 2> gpurun_out/prof/r3_bench_family_tables.txt || { tail -5 gpurun_out/prof/r3_bench_family_tables.txt; exit 1; }
tail -c 1500 gpurun_out/prof/r3_bench.json; echo
export SLIM_STREAM_THREADS=0
ARGS="bench.py --streams 1 --steps 2 --warmup 1 --no-e2e --no-cpu-baseline"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/trace -o r3 -- python3 $ARGS > gpurun_out/prof/r3_bench_under_rocprof_single_stream.json 2> gpurun_out/prof/trace.err || { tail -5 gpurun_out/prof/trace.err; exit 1; }
ARGS1="bench.py --streams 1 --steps 1 --warmup 1 --no-e2e --no-cpu-baseline"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof/fetch -o r3 -- python3 $ARGS1 > /dev/null 2> gpurun_out/prof/fetch.err || { tail -5 gpurun_out/prof/fetch.err; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof/write -o r3 -- python3 $ARGS1 > /dev/null 2> gpurun_out/prof/write.err || { tail -5 gpurun_out/prof/write.err; exit 1; }
find gpurun_out/prof -name "*.csv" | head -20
F=$(find gpurun_out/prof/fetch -name "*counter_collection.csv" | head -1)
W=$(find gpurun_out/prof/write -name "*counter_collection.csv" | head -1)
# the statistics kernel reads 128 * 2048 * 2 bytes per input channel: 3 launches over 4096 channels and one over 11008 per step
rm -f gpurun_out/prof/r3_pmc_traffic.json
python tools/pmc_summarize.py "$F" "$W" gpurun_out/prof/r3_pmc_traffic.json "3053453312/launch" "${COMMIT:-unrecorded}" > gpurun_out/prof/pmc_summary.txt 2>&1 || { tail -5 gpurun_out/prof/pmc_summary.txt; exit 1; }
tail -40 gpurun_out/prof/pmc_summary.txt
S=$(find gpurun_out/prof/trace -name "*kernel_stats.csv" | head -1)
cp "$S" gpurun_out/prof/r3_bench_kernel_stats_single_stream.csv
head -25 "$S" | cut -c1-200
# keep the merged output small: the raw traces stay on the box
rm -rf gpurun_out/prof/trace gpurun_out/prof/fetch gpurun_out/prof/write
