cd $GRAFT_REPO_ROOT
lscpu | grep -i numa
python - <<'PY'
import sys, os, json, subprocess, time
sys.path.insert(0, "tools")
import e2e_bench
from strling_amd import build
inp = e2e_bench.make_input(2**23)
def run(prefix, threads):
    env = dict(os.environ, STRL_DECODE_TIMING="1", STRL_THREADS=str(threads))
    t=time.time()
    r = subprocess.run(prefix + [build.CLI, "extract", "-v", "-g", inp["bed"], inp["bam"], inp["out"]], capture_output=True, text=True, env=env)
    wall=time.time()-t
    line = [l for l in r.stderr.splitlines() if "seconds: total" in l]
    dec = [l.split("decode seconds:")[1].strip() for l in r.stderr.splitlines() if "decode seconds:" in l]
    print(" ".join(prefix), threads, "wall %.2f" % wall, line[-1].split("seconds:")[1] if line else r.stderr[-200:], "|", dec[-1] if dec else "")
for pre, th in (([], 32), (["taskset", "-c", "0-63"], 32), (["taskset", "-c", "0-63"], 64), (["taskset", "-c", "0-63,128-191"], 64), (["taskset", "-c", "0-63,128-191"], 128), (["numactl", "--interleave=all"], 64), ([], 48), ([], 24)):
    try:
        run(pre, th)
    except Exception as e:
        print(pre, th, "failed", e)
PY
