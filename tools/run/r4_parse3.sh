#!/bin/bash
# round 4: two inflate streams + carry buffer: tests, e2e, chunk-size sweep, kernel timeline
mkdir -p gpurun_out/r4
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${TAG:-two}
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_front_device.py tests/test_cli.py tests/test_bgzf_device.py -m gpu -x -q > gpurun_out/r4/${T}_tests.log 2>&1
tail -3 gpurun_out/r4/${T}_tests.log
N=${PAIRS:-33554432}
timeout 1500 python tools/e2e_bench.py $N --dir /tmp --repeats 3 --check-slabs 4 --keep --out gpurun_out/r4/${T}_e2e.json > gpurun_out/r4/${T}_e2e.log 2>&1
python - <<P
import json
j=json.load(open('gpurun_out/r4/${T}_e2e.json'))
for r in j['runs']: print(r['wall_s'], r['loop_s'], r['device_front_end'], r['phases'][:330], '|', r['outside_the_loop'][:200])
print(j['check']['ok'], j['call_s'], j['merge_s'])
P
CLI=$R/strling_amd/lib/strling
cd /tmp
for B in ${SWEEP:-}; do
  for k in 1 2; do
    sleep 2
    STRL_CHUNK_BLOCKS=$B $CLI extract -v -g /tmp/e2e_${N}_6.str /tmp/e2e_${N}_6.bam /tmp/e2e_sweep.bin 2>&1 | grep "seconds: total\|open context" | sed "s/^/blocks $B: /" | cut -c1-420
  done
done > $R/gpurun_out/r4/${T}_sweep.txt 2>&1
cat $R/gpurun_out/r4/${T}_sweep.txt | cut -c1-260
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4/${T}_kt -o run -- $CLI extract -v -g /tmp/e2e_${N}_6.str /tmp/e2e_${N}_6.bam /tmp/e2e_prof.bin > $R/gpurun_out/r4/${T}_kt.log 2>&1
f=$(find $R/gpurun_out/r4/${T}_kt -name 'run_kernel_trace.csv' | head -1)
python $R/tools/trace_timeline.py $f 60 > $R/gpurun_out/r4/${T}_timeline.txt 2>&1
grep -v "^   +" $R/gpurun_out/r4/${T}_timeline.txt
find $R/gpurun_out/r4/${T}_kt -name 'run_kernel_trace.csv' -delete; find $R/gpurun_out/r4/${T}_kt -name '*agent_info*' -delete
