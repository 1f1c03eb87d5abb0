"""Small driver for rocprofv3 runs: a few steps of the device hot path on a reduced batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0], "--steps", os.environ.get("STEPS", "100"), "--warmup", "1", "--no-cpu-baseline",
            "--reads-per-gpu", os.environ.get("READS", str(2 ** 25))]
import bench
bench.main()
