#!/bin/bash
# A/B aid: export a commit's library, host code and bench into ab_prev/ and build it there, so that one GPU call can time
# two versions on the same box:  tools/ab_setup.sh <commit>;  then on the box: python ab_prev/bench.py ... ; python bench.py ...
set -e
cd "$(dirname "$0")/.."
rm -rf ab_prev && mkdir ab_prev
git archive "${1:-HEAD}" slim_amd slim include bench.py __graft_entry__.py profiles/r2_pmc_traffic.json oracle | tar -x -C ab_prev
(cd ab_prev && python -m slim_amd.build > build.log 2>&1 && tail -1 build.log)
