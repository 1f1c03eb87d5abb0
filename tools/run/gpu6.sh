cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/envprobe -o x -- env 2>/dev/null | grep -i "rocp\|preload" | cut -c1-150
cd $GRAFT_REPO_ROOT
timeout 600 python tools/e2e_bench.py 8388608 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['runs'][0]
print('wall', r['wall_s'], 'reads/s wall', r['reads_per_s_wall'], 'loop', r['loop_s'], 'reads/s loop', r['reads_per_s_loop']); print(r['phases'])"
