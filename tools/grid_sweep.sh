#!/bin/bash
# Launch-grid sweep of the two streaming kernels (classify, probe) on the bench batch; prints kernel_ms per setting.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python bench.py --cache /tmp --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > /dev/null 2>&1   # generate the batch once
one() { env "$@" STEPS=10 python tools/prof_run.py 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); k = d['roofline']['kernel_ms']
print('$*', 'ms/step %.3f' % d['ms_per_step'], ' '.join('%s=%.3f' % (a, k[a]) for a in k if 'classify' in a or 'probe' in a))"; }
for c in 1024 1280 1536 2048 2560 3840 5120 10240; do one STRL_GRID_C=$c; done
for p in 256 512 768 1024 2048 4096; do one STRL_GRID_P=$p; done
