cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not_acgt or ragged" 2>&1 | grep -v "^$" | tail -8
