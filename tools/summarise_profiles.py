"""Reduce the rocprofv3 CSVs of tools/collect_profiles.sh to per-kernel averages.
usage: python tools/summarise_profiles.py gpurun_out/prof_<tag>   -> writes pmc_counters.json, traffic.json there."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]

# ---- calibration run (tools/calib_traffic.hip): counter bytes / known bytes per access shape --------------------
calib = {}
try:
    known = json.load(open(os.path.join(root, "calib_bytes.json")))["bytes"]
    raw = defaultdict(lambda: defaultdict(list))
    for d, ctr in (("calib_fetch", "FETCH_SIZE"), ("calib_write", "WRITE_SIZE")):
        for path in glob.glob(os.path.join(root, d, "**", "run_counter_collection.csv"), recursive=True):
            disp = defaultdict(float)
            for row in csv.DictReader(open(path)):
                if row["Counter_Name"] == ctr:
                    disp[(row["Kernel_Name"], row["Dispatch_Id"])] += float(row["Counter_Value"])
            for (name, _), v in disp.items():
                raw[name][ctr].append(v)
    for name, cs in raw.items():
        key = next((k for k in known if name.startswith(k) or k in name), None)
        if key is None:
            continue
        kb = known[key]
        ent = {"known_bytes": kb}
        for ctr, vs in cs.items():
            ent[ctr + "_KiB"] = round(sum(vs) / len(vs), 1)
        f = ent.get("FETCH_SIZE_KiB", 0.0) * 1024
        w = ent.get("WRITE_SIZE_KiB", 0.0) * 1024
        if isinstance(kb, dict):
            ent["fetch_over_element_bytes"] = round(f / (kb["elements"] + kb["index"]), 3)
            ent["fetch_over_line_bytes"] = round(f / (kb["lines64"] + kb["index"]), 3)
        elif "store" in key:
            ent["write_over_known"] = round(w / kb, 3)
        else:
            ent["fetch_over_known"] = round(f / kb, 3)
        calib[key] = ent
    json.dump({"note": "rocprofv3 FETCH_SIZE / WRITE_SIZE (KiB, raw) of tools/calib_traffic.hip against the bytes each kernel must move; "
                       "fetch_over_known ~0.5 confirms the guide's x2 correction for wide streaming reads on gfx950",
               "kernels": calib}, open(os.path.join(root, "calibration.json"), "w"), indent=1)
except Exception as e:
    print("no calibration:", e)

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
