cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_index.py tests/test_product_kats.py -x -q -m gpu 2>&1 | grep -v "^$" | tail -4
timeout 400 python tests/fuzz/fuzz_parity.py 120 2>&1 | tail -3
timeout 600 python bench.py --cache /tmp --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'value', d['value']); print(json.dumps(d['roofline'])[:1800])"
