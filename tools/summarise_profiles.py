"""Reduce the rocprofv3 CSVs of tools/collect_profiles.sh to per-kernel averages.
usage: python tools/summarise_profiles.py gpurun_out/prof_<tag>   -> writes pmc_counters.json, traffic.json there."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
per = defaultdict(lambda: defaultdict(list))
for d in ("fetch", "write", "pmc1", "pmc2"):
    for path in glob.glob(os.path.join(root, d, "**", "run_counter_collection.csv"), recursive=True):
        disp = defaultdict(dict)
        for row in csv.DictReader(open(path)):
            name = row["Kernel_Name"]
            if "strl::" not in name:
                continue
            key = (name, row["Dispatch_Id"])
            disp[key][row["Counter_Name"]] = disp[key].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            per[name]["_vgpr_sgpr_lds_wg_grid"] = [row.get("VGPR_Count"), row.get("SGPR_Count"), row.get("LDS_Block_Size"),
                                                    row.get("Workgroup_Size"), row.get("Grid_Size")]
        for (name, _), cs in disp.items():
            for c, v in cs.items():
                per[name][c].append(v)
out = {}
for name, cs in per.items():
    out[name] = {c: (round(sum(v) / len(v), 1) if c[0] != "_" else v) for c, v in cs.items()}
    # the first launch of every kernel is the warm-up batch of prof_run.py and has the same size: plain mean is fine
json.dump(out, open(os.path.join(root, "pmc_counters.json"), "w"), indent=1)
traffic = {"note": "per launch, tools/prof_run.py (2^25 reads); hbm_bytes_corrected = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: gfx950 "
                   "FETCH_SIZE reports half of wide coalesced reads (MI355X_MICROARCH.md), WRITE_SIZE taken as is", "kernels": {}}
for name, cs in out.items():
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        traffic["kernels"][name] = {"fetch_KiB_raw": cs["FETCH_SIZE"], "write_KiB_raw": cs["WRITE_SIZE"],
                                    "hbm_bytes_corrected": int((2 * cs["FETCH_SIZE"] + cs["WRITE_SIZE"]) * 1024)}
json.dump(traffic, open(os.path.join(root, "traffic.json"), "w"), indent=1)
print("kernels summarised:", len(out))
