cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_index.py tests/test_product_kats.py tests/test_pair_total.py tests/test_front_device.py -x -q -m gpu 2>&1 | grep -v "^$" | grep -i "passed\|failed\|error" | tail -3
python bench.py --cache /tmp --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > /dev/null 2>&1
for rep in 1 2; do
for L in tools/ab/lib_cur.so strling_amd/lib/libstrling_amd.so; do
STRL_LIB=$L timeout 600 python bench.py --cache /tmp --steps 20 --warmup 3 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']
print('$L', 'ms_per_step', d['ms_per_step'], {a:k[a] for a in k if 'compact' in a or 'soft_items' in a}, d['config'].get('last_step_equals_synchronous_pass'))"
done
done
