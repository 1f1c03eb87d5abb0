cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_pair_total.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
python bench.py --cache /tmp --no-e2e --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print(d['roofline']['kernel_ms']); print(d['roofline']['group_ms'])"
