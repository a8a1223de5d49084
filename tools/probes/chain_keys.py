"""Prints the chain keys of the benchmark block's linears (which linears share a lockstep chain, and why not)."""
import os, sys, runpy
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
sys.argv = ["bench.py", "--steps", "2", "--warmup", "1", "--no-e2e", "--no-cpu-baseline"]
import slim_amd.prune as P
real = P._run_fit_chains
def show(phases, plans, n_streams):
    for (name, lin, w), plan in zip(phases.items, plans):
        print("[keys]", name, plan, "hint", hex(w.fit_level_hint), "spread %.2f" % w.scale_spread, "outliers", None if w.outliers is None else w.outliers[1], file=sys.stderr)
    return real(phases, plans, n_streams)
P._run_fit_chains = show
try:
    runpy.run_path(os.path.join(R, "bench.py"), run_name="__main__")
except SystemExit:
    pass
