#!/usr/bin/env python3
"""Digest of tools/profile_bench.sh output: kernel durations, PMC per launch, HBM traffic."""
import csv
import glob
import json
import os
import sys

out = sys.argv[1]


def rows(pattern):
    r = []
    for f in glob.glob(os.path.join(out, pattern)):
        r += list(csv.DictReader(open(f)))
    return r


def last_json(path):
    try:
        for line in reversed(open(path).read().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
    except OSError:
        pass
    return None


digest = {}
stats = [r for r in rows("trace/*/*kernel_stats.csv")]
digest["kernel_stats"] = [{"name": r["Name"][:60], "calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
                           "pct": float(r["Percentage"])} for r in stats[:4]]
digest["bench_full"] = last_json(os.path.join(out, "trace.log"))
pmc = {}
for d in ("pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write"):
    per = {}
    for r in rows(d + "/*/*counter_collection.csv"):
        if "render_kernel" in r["Kernel_Name"]:
            per.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in per.items():
        pmc[k] = v[-1]  # last launch of the short run (a timed step)
    tr = [r for r in rows(d + "/*/*kernel_trace.csv") if "render_kernel" in r["Kernel_Name"]]
    if tr:
        pmc.setdefault("_kernel_ms_" + d, (int(tr[-1]["End_Timestamp"]) - int(tr[-1]["Start_Timestamp"])) / 1e6)
        pmc["_vgpr"], pmc["_sgpr"], pmc["_scratch"], pmc["_lds"] = (tr[-1]["VGPR_Count"], tr[-1]["SGPR_Count"],
                                                                  tr[-1]["Scratch_Size"], tr[-1]["LDS_Block_Size"])
digest["pmc_short_run"] = pmc
short = last_json(os.path.join(out, "pmc_sq.log"))
if short:
    rays = short["config"]["rays_per_step"]
    digest["short_run"] = {"workload": short["config"]["workload"], "rays_per_launch": rays}
    if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
        # FETCH_SIZE / WRITE_SIZE are in KiB (rocprofv3 -L).  The gfx950 x2 correction of
        # MI355X_MICROARCH.md applies to wide coalesced streaming reads only; this kernel's
        # memory traffic is narrow (dword) so the raw figure is kept and labelled as such.
        b = (pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
        digest["short_run"]["hbm_bytes_per_launch"] = b
        digest["short_run"]["hbm_bytes_per_ray"] = b / rays
        digest["short_run"]["algorithmic_bytes_per_ray"] = short["config"]["bytes_per_ray"]
    if "SQ_INSTS_VALU" in pmc:
        digest["short_run"]["valu_wave_insts_per_64_rays"] = pmc["SQ_INSTS_VALU"] / (rays / 64.0)
        digest["short_run"]["salu_wave_insts_per_64_rays"] = pmc["SQ_INSTS_SALU"] / (rays / 64.0)
print(json.dumps(digest, indent=1))
