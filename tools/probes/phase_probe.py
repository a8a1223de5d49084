"""Phase times of the chains of one compression step of the benchmark block (SLIM_PHASE_TRACE): where a chain spends its time
when it waits for the GPU after every phase.  Run on the GPU box: python tools/probes/phase_probe.py"""
import os, sys, subprocess
os.environ["SLIM_PHASE_TRACE"] = "1"
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
sys.argv = ["bench.py", "--steps", "3", "--warmup", "1", "--no-e2e", "--no-cpu-baseline"]
import runpy
import slim_amd.prune as P
try:
    runpy.run_path(os.path.join(R, "bench.py"), run_name="__main__")
except SystemExit:
    pass
tr = P.LAST_PHASE_TRACE
chains = sorted({c for c, _, _ in tr})
n_last = {c: [x for x in tr if x[0] == c][-6:] for c in chains}
for c in chains:
    prev = 0.0
    print("chain", c, " ".join(f"{name} {t - p:.2f}" for (_, name, t), p in zip(n_last[c], [0.0] + [x[2] for x in n_last[c]][:-1])),
          "| total %.2f" % n_last[c][-1][2], file=sys.stderr)
