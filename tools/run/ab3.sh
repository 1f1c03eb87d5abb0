cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python bench.py --cache /tmp --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > /dev/null 2>&1
for rep in 1 2; do
for spec in "$@"; do
L=${spec%%:*}; G=${spec##*:}; [ "$G" = "$spec" ] && G=""
STRL_GRID_C=$G STRL_LIB=$L timeout 600 python bench.py --cache /tmp --steps 20 --warmup 3 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']
print('$spec', 'ms_per_step', d['ms_per_step'], 'classify', k['classify_kernel'])"
done
done
