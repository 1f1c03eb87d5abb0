cd $GRAFT_REPO_ROOT
bash tools/prof_inflate.sh 2>&1 | tail -30
tail -3 /tmp/ibp.log
