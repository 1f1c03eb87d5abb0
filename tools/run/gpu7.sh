cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
STRL_LIB=tools/ab/libstrl_phase.so timeout 600 python tools/phase_timing.py 2>&1 | tail -16
