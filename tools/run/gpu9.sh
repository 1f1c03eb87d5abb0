cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_pair_total.py tests/test_front_device.py tests/test_cli.py -x -q -m gpu 2>&1 | grep -v "^$" | tail -6
