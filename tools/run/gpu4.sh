cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python - <<'PY'
import sys
sys.path.insert(0, "tools")
import e2e_bench
print(e2e_bench.make_input(8388608))
PY
for cb in 6144 8192 12288 12288 18432 24576; do echo "== chunk blocks $cb"; python - $cb <<'PY'
import subprocess, sys, time, os
env = dict(os.environ, STRL_CHUNK_BLOCKS=sys.argv[1], STRL_FRONT_TIMING="1")
t = time.time()
r = subprocess.run(["strling_amd/lib/strling", "extract", "-v", "-g", "/tmp/e2e_8388608.str", "/tmp/e2e_8388608.bam", "/tmp/e2e_8388608.bin"], capture_output=True, text=True, env=env)
w = time.time() - t
print("\n".join(l[:330] for l in r.stderr.splitlines() if "seconds: total" in l or "device front" in l))
print("wall %.3f s" % w)
PY
done
