cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_front_device.py tests/test_cli.py tests/test_bgzf_device.py -x -q -m gpu 2>&1 | grep -v "^$" | tail -12
