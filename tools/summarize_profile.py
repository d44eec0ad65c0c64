#!/usr/bin/env python3
"""Digest of tools/profile_bench.sh output: kernel durations, PMC per launch of the trace kernel,
HBM traffic, and the instruction-mix inputs bench.py's roofline reads.

usage: summarize_profile.py <out-dir> <tag> <workload> [<workload> ...]
Prints the digest (JSON) and writes <out-dir>/roofline_inputs.json."""
import csv
import glob
import json
import os
import sys

out, tag, wls = sys.argv[1], sys.argv[2], sys.argv[3:]
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402  (WORKLOADS, roofline_key, kernel_source_hash: one definition of the digest's keys)


def shape_of(wl):
    """workload tag of tools/profile_bench.sh -> (scene, digest key); "c4s0of8" = shard 0 of 8 of c4."""
    name, _, sh = wl.partition("s") if wl[:2] in ("c4", "c5") and "s" in wl[2:] else (wl, "", "")
    w = bench.WORKLOADS[name]
    shard = tuple(int(x) for x in sh.split("of")) if sh else None
    return w["scene"], bench.roofline_key(w, shard)


def rows(pattern):
    r = []
    for f in glob.glob(os.path.join(out, pattern), recursive=True):
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


digest = {"tag": tag}
stats = rows("trace/**/*kernel_stats.csv")
digest["kernel_stats"] = [{"name": r["Name"][:70], "calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
                           "pct": float(r["Percentage"])} for r in stats[:8]]
digest["bench_full"] = last_json(os.path.join(out, "trace.log"))
inputs = {"note": "instruction mix and HBM bytes of ONE launch of render_kernel per workload, from separate rocprofv3 "
                  "--pmc passes (tools/profile_bench.sh); bench.py's roofline multiplies the per-64-rays counts by the "
                  "rays of its own run and divides by its own HIP-event kernel time",
          "tag": tag, "kernel_source_hash": bench.kernel_source_hash(), "kernels": {}}
for wl in wls:
    scene, key = shape_of(wl)
    pmc = {}
    for d in ("pmc_sq", "pmc_sq2", "pmc_sq3", "pmc_fetch", "pmc_write"):
        # One step = the trace kernel's launches of ONE rtmi_render call: the first pass (probe_kernel: the frame's own
        # first samples since round 4, resumed by the second launch) + render_kernel.  Counters and durations of the two
        # are added; the profiled commands run exactly one step, so the last launch of each name is that step's.
        per, per_probe = {}, {}
        for r in rows("%s_%s/**/*counter_collection.csv" % (wl, d)):
            if "render_kernel" in r["Kernel_Name"]:
                per.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            elif "probe_kernel" in r["Kernel_Name"]:
                per_probe.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in per.items():
            pmc[k] = v[-1] + (per_probe[k][-1] if k in per_probe else 0.0)
        trace = rows("%s_%s/**/*kernel_trace.csv" % (wl, d))
        tr = [r for r in trace if "render_kernel" in r["Kernel_Name"]]
        tp = [r for r in trace if "probe_kernel" in r["Kernel_Name"]]
        if tr:
            pmc["_kernel_ms_" + d] = (int(tr[-1]["End_Timestamp"]) - int(tr[-1]["Start_Timestamp"])) / 1e6
            if tp:
                pmc["_first_pass_ms_" + d] = (int(tp[-1]["End_Timestamp"]) - int(tp[-1]["Start_Timestamp"])) / 1e6
                pmc["_kernel_ms_" + d] += pmc["_first_pass_ms_" + d]
            pmc["_vgpr"], pmc["_sgpr"], pmc["_scratch"], pmc["_lds"] = (tr[-1]["VGPR_Count"], tr[-1]["SGPR_Count"],
                                                                      tr[-1]["Scratch_Size"], tr[-1]["LDS_Block_Size"])
    run = last_json(os.path.join(out, "%s_pmc_sq.log" % wl))
    d = {"pmc": pmc}
    if run and "SQ_INSTS_VALU" in pmc:
        rays = run["config"]["rays_per_step"]
        w64 = rays / 64.0
        d["workload"] = run["config"]["workload"]
        d["rays_per_launch"] = rays
        d["valu_per_64_rays"] = pmc["SQ_INSTS_VALU"] / w64
        d["trans_per_64_rays"] = pmc.get("SQ_INSTS_VALU_TRANS_F32", 0.0) / w64
        d["salu_per_64_rays"] = pmc["SQ_INSTS_SALU"] / w64
        # binary64 arithmetic issues at half the rate (4 cycles per wave-instruction against 2), transcendentals at a
        # quarter (8): bench.CYC_*; SQ_ACTIVE_INST_VALU is the counters' own view of the same thing (cycles the VALU
        # was busy, in units of 4 per SIMD-quad: x 2 / SIMDs = issue cycles per SIMD, as the round-3 review took it)
        f64 = sum(pmc.get(k, 0.0) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64"))
        tr64 = pmc.get("SQ_INSTS_VALU_TRANS_F64", 0.0)
        d["f64_per_64_rays"] = f64 / w64
        d["trans_f64_per_64_rays"] = tr64 / w64
        d["active_valu_per_64_rays"] = pmc.get("SQ_ACTIVE_INST_VALU", 0.0) / w64
        n_simd = 1024
        ms = pmc["_kernel_ms_pmc_sq"]
        tr = pmc.get("SQ_INSTS_VALU_TRANS_F32", 0.0) + tr64
        cyc = (bench.CYC_VALU * (pmc["SQ_INSTS_VALU"] - f64 - tr) + bench.CYC_F64 * f64 + bench.CYC_TRANS * tr) / n_simd
        d["valu_issue_frac_of_profiled_launch"] = cyc / (ms * 1e-3 * 2.4e9)
        d["counter_frac_of_profiled_launch"] = pmc.get("SQ_ACTIVE_INST_VALU", 0.0) * 2.0 / n_simd / (ms * 1e-3 * 2.4e9)
        if "SQ_WAVE_CYCLES" in pmc and "SQ_BUSY_CYCLES" in pmc:
            d["waves"] = pmc["SQ_WAVES"]
        if "SQ_WAIT_ANY" in pmc and "SQ_ACTIVE_INST_ANY" in pmc:
            tot = pmc["SQ_WAIT_ANY"] + pmc["SQ_WAIT_INST_ANY"] + pmc["SQ_ACTIVE_INST_ANY"]
            d["wave_time_split"] = {"waiting": pmc["SQ_WAIT_ANY"] / tot, "issue_stalled": pmc["SQ_WAIT_INST_ANY"] / tot,
                                    "issuing": pmc["SQ_ACTIVE_INST_ANY"] / tot}
        entry = {"valu_per_64_rays": d["valu_per_64_rays"], "trans_per_64_rays": d["trans_per_64_rays"] + d["trans_f64_per_64_rays"],
                 "f64_per_64_rays": d["f64_per_64_rays"], "active_valu_per_64_rays": d["active_valu_per_64_rays"],
                 "salu_per_64_rays": d["salu_per_64_rays"], "vgpr": pmc.get("_vgpr"), "sgpr": pmc.get("_sgpr"),
                 "lds_bytes": pmc.get("_lds"), "profiled_kernel_ms": ms,
                 "source": "profiles/%s_summary.json (%s)" % (tag, wl), "hbm_bytes_per_launch": None}
        if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
            # FETCH_SIZE / WRITE_SIZE are in KiB.  The gfx950 x2 correction of MI355X_MICROARCH.md applies to wide
            # coalesced streaming reads; this kernel's HBM traffic is narrow (dword RNG-state / radiance accesses,
            # 16-byte gathers for meshes), so the raw figure is kept and labelled as such.
            b = (pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
            d["hbm_bytes_per_launch"] = b
            d["hbm_fetch_bytes"], d["hbm_write_bytes"] = pmc["FETCH_SIZE"] * 1024.0, pmc["WRITE_SIZE"] * 1024.0
            entry["hbm_bytes_per_launch"] = b
        inputs["kernels"].setdefault(scene, {"shapes": {}})["shapes"][key] = entry
    digest[wl] = d
json.dump(inputs, open(os.path.join(out, "roofline_inputs.json"), "w"), indent=1)
digest["copy"] = ["cp gpurun_out/%s/%s_summary.json profiles/" % (tag, tag),
                  "cp gpurun_out/%s/roofline_inputs.json profiles/" % tag]
print(json.dumps(digest, indent=1))
