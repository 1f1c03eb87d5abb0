cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python bench.py --cache /tmp --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > /dev/null 2>&1
for rep in 1 2; do
for L in "$@"; do
STRL_LIB=$L timeout 600 python bench.py --cache /tmp --steps 20 --warmup 3 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']; g=d['roofline']['group_ms']
print('$L', 'ms_per_step', d['ms_per_step'], {a:k[a] for a in k if 'segment' in a or 'soft_compact' in a}, 'soft group', g['score_kernel<soft>'], d['config'].get('last_step_equals_synchronous_pass'))"
done
done
