cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python tools/e2e_bench.py 8388608 0 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for r in d['runs']:
    print('wall', r['wall_s'], 'reads/s wall', r['reads_per_s_wall'], 'loop', r['loop_s'], 'reads/s loop', r['reads_per_s_loop']); print(r['phases']); print(r['device_front_end'])"
