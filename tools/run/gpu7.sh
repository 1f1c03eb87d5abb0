cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_cli.py -x -q -m gpu -k "several or byte_identical" 2>&1 | tail -25
