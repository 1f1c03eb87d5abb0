cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_front_device.py tests/test_bgzf_device.py -x -q 2>&1 | tail -25
timeout 600 python -m pytest tests/test_cli.py -x -q -m gpu 2>&1 | tail -15
