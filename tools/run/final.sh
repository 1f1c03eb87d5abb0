cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^$" | grep -i "passed\|failed\|error" | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
mkdir -p gpurun_out/final
( time python bench.py ) > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err; tail -4 gpurun_out/final/bench.err
cut -c1-300 gpurun_out/final/bench.json
