"""Per-phase durations of the chains of one bench step (SLIM_PHASE_TRACE: every chain waits for the GPU after each of its
phases, so the phases of a chain are serialised against the host; the other chains still run beside it).
    python tools/phase_trace.py [bench flags]"""
import os
import sys

os.environ["SLIM_PHASE_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from slim_amd import prune as P

sys.argv = [sys.argv[0], "--no-cpu-baseline", "--no-e2e", "--steps", "1", "--warmup", "1"] + sys.argv[1:]
orig = P.compress_linears_unchecked
steps = []


def traced(*a, **k):
    del P.LAST_PHASE_TRACE[:]
    orig(*a, **k)
    steps.append(list(P.LAST_PHASE_TRACE))


P.compress_linears_unchecked = traced
B.main()
# the timed step is the fourth call (two set-up steps, one warm-up, then the timed one)
for rec in steps[3:4]:
    last = {}
    for chain, name, ms in rec:
        print(f"chain {chain}  {name:<16} at {ms:7.2f} ms  (+{ms - last.get(chain, 0.0):6.2f})", file=sys.stderr)
        last[chain] = ms
