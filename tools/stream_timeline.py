"""Timeline of one bench step's compression phase: per linear, when its host thread enqueued it and when its stream ran it.
    python tools/stream_timeline.py [--streams 7]"""
import os
import sys

os.environ["SLIM_STREAM_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from slim_amd import prune as P

orig = P.run_on_streams
calls = []


def traced(jobs, dev, n_streams=None, stream_ids=None, pool=None):
    orig(jobs, dev, n_streams, stream_ids, pool)
    if P.LAST_STREAM_TRACE:
        calls.append(list(P.LAST_STREAM_TRACE))


P.run_on_streams = traced
if "--only" in sys.argv:                      # e.g. --only 0,1,2,3: the timeline of a subset of the block's linears
    k = sys.argv.index("--only")
    keep = [int(x) for x in sys.argv[k + 1].split(",")]
    B.LLAMA2_7B_BLOCK[:] = [B.LLAMA2_7B_BLOCK[i] for i in keep]
    del sys.argv[k:k + 2]
sys.argv = [sys.argv[0], "--no-cpu-baseline", "--no-e2e", "--steps", "1", "--warmup", "1"] + sys.argv[1:]
B.main()
# every run_on_streams call of the run, in order (jobs are linears, or groups of linears in the grouped phases)
for k, c in enumerate(calls):
    print(f"-- call {k}: {len(c)} jobs; gpu spans (ms): " +
          "  ".join(f"{r['gpu_start_ms']:.2f}-{r['gpu_end_ms']:.2f}" for r in c) + "; host enqueue spans (ms): " +
          "  ".join(f"{r['host_start_ms']:.2f}-{r['host_end_ms']:.2f}" for r in c), file=sys.stderr)
