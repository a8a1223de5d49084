# Kernel-by-kernel trace (rocprofv3 --kernel-trace) of the bench step restricted to the four attention linears of the block
# (one lockstep chain), on a single stream fed by one thread: start / end of every launch of the last step.
set -o pipefail
mkdir -p gpurun_out/trace
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp SLIM_STREAM_THREADS=0
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace/raw -o t -- python3 bench.py --streams 1 --steps 1 --warmup 0 --no-e2e --no-cpu-baseline --only ${ONLY:-0,1,2,3} > gpurun_out/trace/bench.json 2> gpurun_out/trace/err.txt || { tail -5 gpurun_out/trace/err.txt; exit 1; }
F=$(find gpurun_out/trace/raw -name "*kernel_trace.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
out = open("gpurun_out/trace/all.txt", "w")
prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][:48]
    out.write(f"{(s - t0) / 1e3:10.1f} us  +gap {(s - prev_end) / 1e3:7.1f}  dur {(e - s) / 1e3:8.1f}  grid {r.get('Grid_Size_X','?'):>8}x{r.get('Grid_Size_Y','?')}x{r.get('Grid_Size_Z','?')}  {name}\n")
    prev_end = max(prev_end, e)
out.close()
print("launches:", len(rows))
PY
rm -rf gpurun_out/trace/raw
