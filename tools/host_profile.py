"""Host-side cost of one bench step: cProfile of bench.py on a single stream (everything enqueued by one thread), top
functions by own time.    python tools/host_profile.py"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B

sys.argv = [sys.argv[0], "--no-cpu-baseline", "--steps", "4", "--warmup", "1", "--streams", "1"]
pr = cProfile.Profile()
pr.enable()
B.main()
pr.disable()
st = pstats.Stats(pr, stream=sys.stderr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats("slim_amd", 30)
