cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
BENCH_DATA=1 READS=4194304 STRL_LIB=tools/ab/libstrl_phase.so timeout 900 python tools/phase_timing.py 2>&1 | tail -16
