#!/bin/bash
# round 4 evidence in one call: the bench step's rocprofv3 kernel stats + PMC passes + calibration, then the front end's
bash tools/collect_profiles.sh r4a > gpurun_out/r4_collect_a.log 2>&1
PAIRS=33554432 bash tools/collect_e2e.sh r4e > gpurun_out/r4_collect_e.log 2>&1
tail -5 gpurun_out/r4_collect_a.log | cut -c1-800; tail -8 gpurun_out/r4_collect_e.log | cut -c1-600
