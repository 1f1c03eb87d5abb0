#!/bin/bash
# where the 0.2 s behind `strling extract`'s last line go: the process' clock at its last line against the caller's, with the
# feeds, with the teardown done by hand (timed), with the host feed alone, on the 1.3e8-read file
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && export TMPDIR=/tmp
O=gpurun_out/r6o; mkdir -p $O
CLI=$R/strling_amd/lib/strling
python - > $O/make_2p27.log 2>&1 <<'PY'
import sys; sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
import e2e_bench
inp = e2e_bench.make_input(67108864, d='/dev/shm')
PY
B=$(ls /dev/shm/e2e_67108864_*.bam | head -1); S=${B%.bam}.str
run() {  # label, env...
  local label="$1"; shift
  sleep 3
  local s=$(date +%s.%N)
  env "$@" timeout 300 $CLI extract -v -g $S $B /dev/shm/x.bin > /tmp/run.err 2>&1
  local e=$(date +%s.%N)
  python3 - "$label" "$s" "$e" <<'PY'
import re, sys
t = open('/tmp/run.err').read()
m = re.findall(r'now ([0-9.]+) s after exec', t)
td = re.findall(r'teardown by hand: (.*?); now', t)
print("%-34s whole process %.3f s, its own clock at the last line %s s%s" % (sys.argv[1], float(sys.argv[3]) - float(sys.argv[2]), m[-1] if m else "?", (" | " + td[-1]) if td else ""))
PY
}
{
for rep in 1 2 3; do
  run "default (mapped feed)" A=1
  run "STRL_FEED=pread" STRL_FEED=pread
  run "STRL_TEARDOWN=1" STRL_TEARDOWN=1
  run "STRL_TEARDOWN=1 STRL_FEED=pread" STRL_TEARDOWN=1 STRL_FEED=pread
  run "STRL_THREADS=4" STRL_THREADS=4
done
} > $O/exit_where.log 2>&1
cat $O/exit_where.log
