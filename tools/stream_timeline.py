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


def traced(jobs, dev, n_streams=None):
    orig(jobs, dev, n_streams)
    if P.LAST_STREAM_TRACE:
        calls.append(list(P.LAST_STREAM_TRACE))


P.run_on_streams = traced
sys.argv = [sys.argv[0], "--no-cpu-baseline", "--steps", "1", "--warmup", "1"] + sys.argv[1:]
B.main()
last = [c for c in calls if len(c) == 7][-2]          # the last multi-stream step (the final call is the single-stream one)
names = [x[0] for x in B.LLAMA2_7B_BLOCK]
for r in last:
    print(f"{names[r['job']]:<10} stream {r['stream']}  host enqueue {r['host_start_ms']:6.2f} -> {r['host_end_ms']:6.2f} ms   "
          f"gpu {r['gpu_start_ms']:6.2f} -> {r['gpu_end_ms']:6.2f} ms", file=sys.stderr)
