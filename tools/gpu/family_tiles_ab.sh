# A/B of the GEMM tile per shape family on the bench step (SLIM_GEMM_FAMILY_TILES="family:tile,...").
# families: 0 gram, 1 small, 2 apply, 3 sketch, 4 big, 5 residual; tiles: see the enum in csrc/gemm.hip
for cfg in "${@}"; do
    SLIM_GEMM_FAMILY_TILES="$cfg" timeout -k 10 200 python bench.py --steps ${STEPS:-6} --warmup 2 --no-e2e --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_single_stream']
print('$cfg', round(d['ms_per_step'],2), {n[8:]:round(v['total_ms'],2) for n,v in k.items() if n.startswith('gemm_nt')}, 'splitk', round(k['splitk_reduce']['total_ms'],2))" || exit 1
done
