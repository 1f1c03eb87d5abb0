cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
{ timeout 900 python tests/fuzz/fuzz_parity.py ${1:-300} 2>&1 | tail -1
  timeout 900 python tests/fuzz/fuzz_call.py ${1:-300} 2>&1 | tail -3; } | tee gpurun_out/r4/fuzz.log
