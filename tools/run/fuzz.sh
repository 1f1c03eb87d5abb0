cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python tests/fuzz/fuzz_parity.py 480 2>&1 | tail -1
timeout 600 python tests/fuzz/fuzz_call.py 240 2>&1 | tail -1
