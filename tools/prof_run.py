"""Small driver for rocprofv3 runs: steps of the device hot path on the bench batch (generated once into /tmp by a
previous un-profiled `python bench.py --cache /tmp ...` run, so that nothing forks under the profiler)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0], "--steps", os.environ.get("STEPS", "50"), "--warmup", "1", "--no-cpu-baseline", "--no-e2e", "--cache", "/tmp",
            "--reads-per-gpu", os.environ.get("READS", str(2 ** 25))]
import bench
bench.main()
