cd $GRAFT_REPO_ROOT
bash tools/prof_inflate.sh 2>&1 | tail -22
grep "GB/s" /tmp/ibp.log | tail -2
