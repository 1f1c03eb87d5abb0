cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_cli.py tests/test_front_device.py tests/test_bgzf_device.py -x -q -m gpu 2>&1 | grep -v "^$" | grep -i "passed\|failed\|error" | tail -4
timeout 600 python tools/e2e_bench.py 8388608 0 0 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for r in d['runs']:
    print('wall', r['wall_s'], 'reads/s wall', r['reads_per_s_wall'], 'loop', r['loop_s'], 'reads/s loop', r['reads_per_s_loop']); print({k:v for k,v in r['device_front_end'].items() if k!='note'})"
