#!/usr/bin/env python3
"""bench.py — Mrays/s of the path-tracing hot path on MI355X.

One *step* = one frame of the workload through the C ABI: the trace kernel for this rank's
pixel tiles, plus (N > 1) the RCCL gather of the tile buffers to rank 0 and the untile
kernel.  Scene and RNG states are resident in HBM before the timed region.

Workloads (BASELINE.json ``configs``; scene constants are the reference's InitWorld literals):
  c2  scenes/cornell_box 1024x1024 x1024 spp depth 50      — the N=1 line (configs[1])
  c3  scenes/bunny (stand-in mesh) 1024x1024 x512 spp d10   — reported under config.extra at N=1
  c4  scenes/cornell_box 2048x2048 x4096 spp depth 50       — the N>1 line: FIXED frame, strong scaling
  c5  scenes/birthday 4096x4096 x8192 spp depth 10
  c1  scenes/spheres 256x256 x16 spp depth 8 (configs[0]) and c1big, the same scene at 1024x1024 x64 spp
``--workload auto`` (default) picks c2 for N=1 and c4 for N>1.  ``--shard r/G`` renders shard r of a G-rank frame
on this one GPU (what rank r of G would render); ``--shard-sweep G`` renders the full frame and then each of the G
shards of the workload in turn and reports ``predicted_speedup = T(full frame) / max_r T(shard r)`` -- the
strong-scaling figure a G-GPU node can reach at best, measured on one GPU.

Launch: ``python bench.py --gpus N`` starts its own N ranks (one process per GPU, before any
GPU call in the parent) unless it already runs under ``python -m torch.distributed.run``
(RANK/WORLD_SIZE set).  It never runs fewer ranks than asked for.

Prints ONE JSON line on rank 0, including
  roofline     — the bound the counters support: VALU issue.  ``achieved`` = issue cycles per SIMD
                 per launch = (2 x plain + 4 x binary64 + 8 x transcendental wave-instructions per 64
                 rays, from the committed rocprofv3 PMC digest profiles/roofline_inputs.json) x this
                 run's rays / 64 / SIMDs; ``peak`` = the kernel's duration measured here with HIP events
                 x 2.4 GHz.  ``counter_frac`` is the same fraction from a different counter
                 (SQ_ACTIVE_INST_VALU x 2 / SIMDs).  ``logical_bytes`` keeps the SURVEY 8(d) bytes-per-ray
                 figure -- scene bytes a query CONSULTS, served from SGPRs / LDS, not a memory roofline --
                 and ``traffic`` is the FETCH_SIZE + WRITE_SIZE bytes per launch of the same workload.
  cpu_baseline — the CPU oracle ("port") on this box's cores, on SURVEY 8(d)'s two samples.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing-cuda_amd"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
CLOCK_GHZ = 2.4        # ... max clock 2400 MHz
# issue cycles per wave64 instruction on a SIMD (same guide): plain VALU 2, binary64 arithmetic 4 (half rate on this
# part), transcendental 8
CYC_VALU, CYC_F64, CYC_TRANS = 2, 4, 8

WORKLOADS = {
    "c2": dict(scene="cornell_box", size=1024, spp=1024, depth=50),
    "c3": dict(scene="bunny", size=1024, spp=512, depth=10),
    "c4": dict(scene="cornell_box", size=2048, spp=4096, depth=50),
    "c5": dict(scene="birthday", size=4096, spp=8192, depth=10),
    "c1": dict(scene="spheres", size=256, spp=16, depth=8),
    "c1big": dict(scene="spheres", size=1024, spp=64, depth=8),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="auto", choices=["auto"] + sorted(WORKLOADS))
    ap.add_argument("--scene", default=None, choices=["cornell_box", "spheres", "bunny", "birthday"])
    ap.add_argument("--size", type=int, default=None, help="frame side in pixels (the WHOLE frame, for any N)")
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--depth", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the config.extra lines (C3, C1, C4 / C5 shard 0 of 8)")
    ap.add_argument("--shard", default=None, help="r/G: render shard r of a G-rank frame on this one GPU")
    ap.add_argument("--shard-sweep", type=int, default=0, metavar="G",
                    help="full frame, then each of the G shards, on this one GPU: predicted G-GPU speedup")
    ap.add_argument("--probe-spp", type=int, default=0, help="rtmi_render_opts.probe_spp (0: library default)")
    ap.add_argument("--blocks-per-cu", type=int, default=0)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--rank-timeout", type=float, default=None,
                    help="seconds after which the launcher (--gpus N started without torch.distributed.run) gives up on its ranks")
    ap.add_argument("--selftest-die-rank", type=int, default=-1,
                    help="with --selftest-exchange: this rank exits 3 before the exchange (launcher test)")
    ap.add_argument("--selftest-exchange", action="store_true",
                    help="CPU-only check of the launcher + the N-rank exchange (gloo, no rendering)")
    return ap.parse_args(argv)


def pick_workload(args, world):
    name = args.workload
    if name == "auto":
        name = "c2" if world == 1 else "c4"
    w = dict(WORKLOADS[name])
    custom = False
    for k in ("scene", "size", "spp", "depth"):
        v = getattr(args, k)
        if v is not None and v != w[k]:
            w[k] = v
            custom = True
    w["name"] = "custom" if custom else name
    return w


def build_scene(builder, name, aspect):
    from rtmi import scenes
    if name == "cornell_box":
        scenes.cornell_box(builder, aspect)
    elif name == "spheres":
        scenes.spheres(builder, aspect)
    elif name == "bunny":
        scenes.bunny(builder, aspect, scenes.procedural_bunny_mesh())
    elif name == "birthday":
        scenes.birthday(builder, aspect, scenes.procedural_earthmap(1024, 2048))
    return builder


def host_cores():
    """CPU threads this process may really use: affinity, capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = max(1, min(n, int(q / p + 0.5)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(target_s=6.0):
    """The oracle (CPU restatement, kind "port") on SURVEY 8(d)'s two scenes: C1 = spheres 256x256 depth 8
    (configs[0]) and a reduced C2 = cornell_box 256x256 depth 50.  Each is first timed at SURVEY's sample count
    (16 / 64 spp: a fraction of a second, thread start-up included), then at as many samples per pixel as make the
    run last about ``target_s`` seconds -- that second run is what is reported.  RNG seeding is outside the timed
    region, as the reference's kernel-only timing is (utils.cu:155-170).  ``value`` is the reduced-C2 rate (the
    scene of the N=1 line)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oraclelib
    from rtmi import scenes
    cores = host_cores()
    res = {}
    for tag, name, side, spp0, depth in (("c1_spheres_256x256_d8", "spheres", 256, 16, 8),
                                         ("c2_reduced_cornell_256x256_d50", "cornell_box", 256, 64, 50)):
        seed = scenes.SCENE_SEEDS[name]
        b = build_scene(oraclelib.OracleBuilder(seed), name, 1.0)

        def run(spp):
            states = oraclelib.rng_init(seed, side * side)
            states[0] = b.state0
            t0 = time.time()
            _, _, _, rays = b.render(side, side, spp, depth, threads=cores, states=states)
            return rays, time.time() - t0

        rays0, dt0 = run(spp0)
        spp = int(min(64 * spp0, max(spp0, spp0 * round(target_s / max(dt0, 1e-3)))))
        rays, dt = run(spp) if spp > spp0 else (rays0, dt0)
        res[tag] = {"spp": spp, "mrays_per_s": rays / dt / 1e6, "rays": rays, "seconds": dt,
                    "mrays_per_s_per_thread": rays / dt / 1e6 / cores,
                    "survey_sample": {"spp": spp0, "rays": rays0, "seconds": dt0, "mrays_per_s": rays0 / dt0 / 1e6}}
    main, c1 = res["c2_reduced_cornell_256x256_d50"], res["c1_spheres_256x256_d8"]
    return {"value": main["mrays_per_s"], "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "cornell_box 256x256 x%dspp depth50: %d rays in %.1fs; spheres 256x256 x%dspp depth8: %d rays in "
                      "%.1fs (oracle, g++ -O2 -ffp-contract=off, %d threads, render loop only)" %
                      (main["spp"], main["rays"], main["seconds"], c1["spp"], c1["rays"], c1["seconds"], cores),
            "per_thread": main["mrays_per_s_per_thread"], "samples": res}


def roofline_inputs():
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "roofline_inputs.json")))
    except (OSError, ValueError):
        return {}


def kernel_source_hash():
    """sha256 over the kernel sources (ray-tracing-cuda_amd/csrc): tools/summarize_profile.py records it with
    every PMC digest, and a digest taken from other sources than the ones built here is not used."""
    import hashlib
    d = os.path.join(ROOT, "ray-tracing-cuda_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")) or name == "Makefile":
            h.update(name.encode() + b"\0" + open(os.path.join(d, name), "rb").read() + b"\0")
    return h.hexdigest()[:16]


def roofline_key(w, shard=None):
    key = "%dx%dx%d_d%d" % (w["size"], w["size"], w["spp"], w["depth"])
    return key + ("_s%dof%d" % shard if shard else "")


def roofline(w, rays_rank, kern_ms, bytes_per_ray, n_simd, shard=None):
    """VALU-issue roofline of the trace kernel on this rank (see the module docstring).  The instruction counts per
    64 rays come from the committed PMC digest of THIS scene and frame shape; a digest taken from other kernel
    sources (or none for this shape) leaves ``frac`` null and says why -- the line is printed either way."""
    digest = roofline_inputs()
    scene = digest.get("kernels", {}).get(w["scene"], {})
    key = roofline_key(w, shard)
    inp = scene.get("shapes", {}).get(key)
    logical = rays_rank * bytes_per_ray / (kern_ms * 1e-3) / 1e9
    out = {"bound": "valu_issue", "achieved": None, "peak": kern_ms * 1e-3 * CLOCK_GHZ * 1e9, "unit": "cycles/SIMD",
           "frac": None, "counter_frac": None, "traffic": None,
           "logical_bytes": {"bytes_per_ray": bytes_per_ray, "gbs": logical, "over_hbm_peak": logical / HBM_PEAK_GBS,
                             "note": "SURVEY 8(d): scene bytes a closest-hit query consults under the reference's "
                                     "algorithm x rays / kernel time.  NOT a roofline: these bytes are served from SGPRs "
                                     "/ the scalar cache / LDS (a ratio above 1 says exactly that); HBM sees `traffic`"},
           "hbm": {"peak_gbs": HBM_PEAK_GBS}}
    if not inp:
        out["stale_inputs"] = "no PMC digest for %s %s in profiles/roofline_inputs.json" % (w["scene"], key)
        return out
    built, profiled = kernel_source_hash(), digest.get("kernel_source_hash")
    if profiled != built:
        out["stale_inputs"] = ("profiles/roofline_inputs.json was taken from kernel sources %s, these are %s: "
                               "re-run tools/profile_bench.sh" % (profiled, built))
        sys.stderr.write("bench.py: warning: %s\n" % out["stale_inputs"])
        return out
    valu, trans, f64 = inp["valu_per_64_rays"], inp["trans_per_64_rays"], inp.get("f64_per_64_rays", 0.0)
    cyc = (CYC_VALU * (valu - trans - f64) + CYC_F64 * f64 + CYC_TRANS * trans) * (rays_rank / 64.0) / n_simd
    out.update(achieved=cyc, frac=cyc / out["peak"], valu_per_64_rays=valu, trans_per_64_rays=trans, f64_per_64_rays=f64,
               cycles_per_instruction={"valu": CYC_VALU, "f64": CYC_F64, "trans": CYC_TRANS},
               clock_ghz=CLOCK_GHZ, n_simd=n_simd, source=inp.get("source"), kernel_source_hash=built)
    if inp.get("active_valu_per_64_rays"):
        out["counter_frac"] = inp["active_valu_per_64_rays"] * (rays_rank / 64.0) * 2.0 / n_simd / out["peak"]
    tr = inp.get("hbm_bytes_per_launch")
    if tr is not None:
        out["traffic"] = tr
        out["hbm"]["measured_gbs"] = tr / (kern_ms * 1e-3) / 1e9
        out["hbm"]["measured_frac"] = tr / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    if out["frac"] > 1.0:  # cannot be: the counts are not this kernel's after all
        out["stale_inputs"] = "frac %.3f > 1: the digest does not describe the kernel that ran" % out["frac"]
        sys.stderr.write("bench.py: warning: %s\n" % out["stale_inputs"])
        out["achieved"] = out["frac"] = None
    return out


def run_workload(rtmi, torch, dist, w, args, rank, world, use_dist, steps, warmup, shard=None, scene=None):
    """Time `steps` frames of workload `w`.  shard = (r, G): this one GPU renders what rank r of G would (no
    exchange); otherwise rank / world are the process group's."""
    from rtmi import scenes
    from rtmi.dist import gather_to_root
    H = W = w["size"]
    seed = scenes.SCENE_SEEDS[w["scene"]]
    if scene is None:
        scene = build_scene(rtmi.SceneBuilder(seed), w["scene"], W / H).commit()
    bytes_per_ray = scene.bytes_per_ray()
    r_rank, r_world = shard if shard else (rank, world)
    R = rtmi.Renderer(scene, H, W, w["spp"], w["depth"], True, rank=r_rank, world_size=r_world)
    R.init_rng()
    pristine = R.states.clone()
    opts = rtmi.render_opts(probe_spp=args.probe_spp) if args.probe_spp > 0 else None
    shape = R.launch_shape(opts)
    mode = R.mode(opts)
    torch.cuda.synchronize()
    # (the exchange's tensors stay on the device with RCCL; the gloo test hook of main() takes them through the host)
    on_device = not (use_dist and dist.get_backend() == "gloo")
    xdev = "cuda" if on_device else "cpu"

    def step(ev=None):
        R.states.copy_(pristine)
        if ev:
            ev[0].record()
        R.render(opts=opts)
        if ev:
            ev[1].record()
        if shard:
            R.check()  # one shard of G: there is no frame to assemble on this GPU
        elif use_dist:
            R.check()  # an incomplete frame raises here, before it is handed on
            allt = gather_to_root(R.tiles if on_device else R.tiles.cpu(), 0)
            if rank == 0:
                R.untile(allt if on_device else allt.cuda())
        else:
            R.untile()

    for _ in range(warmup):
        step()
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(events[i])
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0

    rays_rank = R.total_rays()  # rays of one step on this rank (identical every step)
    kerns = [a.elapsed_time(b) for a, b in events]
    kern_ms = sum(kerns) / max(1, steps)
    per_rank_ms = [kern_ms]
    rays_all = float(rays_rank)
    n_seen = 1
    if use_dist and not shard:
        tt = torch.tensor([dt, float(rays_rank), kern_ms], dtype=torch.float64, device=xdev)
        tmax = tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tt.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        allk = [torch.zeros(1, dtype=torch.float64, device=xdev) for _ in range(world)]
        dist.all_gather(allk, tt[2:3].clone())
        dt, rays_all = float(tmax[0]), float(tsum[1])
        per_rank_ms = [float(x[0]) for x in allk]
        n_seen = dist.get_world_size()
    n_simd = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count * 4
    pixels_rank = int((rtmi.pixel_map(R.frame) >= 0).sum())
    return dict(H=H, W=W, seed=seed, dt=dt, rays_rank=rays_rank, rays_all=rays_all, kern_ms=kern_ms, kern_ms_all=kerns,
                per_rank_ms=per_rank_ms, n_seen=n_seen, bytes_per_ray=bytes_per_ray, n_simd=n_simd, shape=shape, mode=mode,
                pixels_rank=pixels_rank, items=R.items, scene=scene)


def extra_line(tag, w, r, steps, shard=None):
    """One config.extra entry: a further BASELINE config (or one shard of it) on this GPU, with its own roofline."""
    lanes = r["shape"]["blocks"] * r["shape"]["threads"]
    return {"workload": "%s: scenes/%s %dx%d x%dspp depth%d seed%d%s" %
                        (tag, w["scene"], r["H"], r["W"], w["spp"], w["depth"], r["seed"],
                         " -- shard %d of %d (what one GPU of %d renders)" % (shard[0], shard[1], shard[1]) if shard else ""),
            "value": r["rays_all"] * steps / r["dt"] / 1e6, "unit": "Mrays/s", "ms_per_step": r["dt"] / steps * 1e3,
            "kernel_ms": r["kern_ms"], "kernel_ms_min_max": [min(r["kern_ms_all"]), max(r["kern_ms_all"])],
            "rays_per_step": r["rays_all"], "pixels": r["pixels_rank"], "resident_lanes": lanes, "schedule": r["mode"],
            "pixels_per_lane": r["pixels_rank"] / float(lanes),
            "roofline": roofline(w, r["rays_rank"], r["kern_ms"], r["bytes_per_ray"], r["n_simd"], shard)}


def shard_sweep(rtmi, torch, w, G, args, rounds=3, full_rounds=3):
    """T(full frame on one GPU) against T(each of the G shards on one GPU): rank r of G renders exactly shard r, so
    max_r T(shard r) (+ the gather, bytes stated) is the G-GPU frame time a node can reach, measured here.  Single
    renders scatter by a few per cent (which wave draws which pixel is left to atomics), so every shard is rendered
    ``rounds`` times, the shards taking turns (0..G-1, 0..G-1, ...: no back-to-back repeats), and a shard's time is the
    MEDIAN of its renders; a frame ends with its slowest rank, so the figure is T_full (median) / max_r median_r."""
    from rtmi import scenes
    H = W = w["size"]
    seed = scenes.SCENE_SEEDS[w["scene"]]
    scene = build_scene(rtmi.SceneBuilder(seed), w["scene"], W / H).commit()
    opts = rtmi.render_opts(probe_spp=args.probe_spp) if args.probe_spp > 0 else None

    def prepare(rank, world):
        R = rtmi.Renderer(scene, H, W, w["spp"], w["depth"], True, rank=rank, world_size=world)
        R.init_rng()
        return R, R.states.clone()

    def once(R, pristine):
        R.states.copy_(pristine)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        R.render(opts=opts)
        e1.record()
        torch.cuda.synchronize()
        R.check()
        R.last_rays = R.total_rays()  # (the scene's counters are those of its most recent render: read them now)
        return e0.elapsed_time(e1)

    Rf, pf = prepare(0, 1)
    once(Rf, pf)  # warm-up loads the code object
    full_ms = sorted(once(Rf, pf) for _ in range(full_rounds))
    full_rays = Rf.last_rays
    shape = Rf.launch_shape(opts)
    full = {"kernel_ms": full_ms[len(full_ms) // 2], "kernel_ms_min_max": [full_ms[0], full_ms[-1]], "renders": full_rounds,
            "rays": float(full_rays), "pixels": H * W, "resident_lanes": shape["blocks"] * shape["threads"]}
    del Rf, pf
    rs = [prepare(r, G) for r in range(G)]
    times = [[] for _ in range(G)]
    for _ in range(rounds):
        for r in range(G):
            times[r].append(once(*rs[r]))
    shards = []
    for r in range(G):
        R = rs[r][0]
        sh = R.launch_shape(opts)
        lanes = sh["blocks"] * sh["threads"]
        t = sorted(times[r])
        px = int((rtmi.pixel_map(R.frame) >= 0).sum())
        rays = float(R.last_rays)
        shards.append({"shard": r, "kernel_ms": t[len(t) // 2], "kernel_ms_min_max": [t[0], t[-1]], "renders": rounds,
                       "rays": rays, "pixels": px, "resident_lanes": lanes, "pixels_per_lane": px / float(lanes),
                       "mrays_per_s": rays / t[len(t) // 2] / 1e3, "gather_bytes": R.items * 12})
    worst = max(s_["kernel_ms"] for s_ in shards)
    worst_single = max(s_["kernel_ms_min_max"][1] for s_ in shards)
    return {"workload": "%s: scenes/%s %dx%d x%dspp depth%d" % (w["name"], w["scene"], w["size"], w["size"], w["spp"], w["depth"]),
            "G": G, "full_frame": full, "shards": shards, "max_shard_ms": worst, "max_single_render_ms": worst_single,
            "predicted_speedup": full["kernel_ms"] / worst, "predicted_efficiency": full["kernel_ms"] / worst / G,
            "predicted_speedup_worst_single_render": full["kernel_ms_min_max"][0] / worst_single,
            "ray_total_matches": abs(sum(s_["rays"] for s_ in shards) - full["rays"]) < 0.5,
            "note": "kernel-only times on ONE GPU, medians of %d renders per shard taken in turns (full frame: median of "
                    "%d); predicted_speedup = T_full / max over shards of the shard's median; the G-GPU step adds one "
                    "gather of gather_bytes per rank to rank 0 (direct xGMI links, ~153 GB/s each) and the untile kernel"
                    % (rounds, full_rounds)}


def sweep_digest(sw):
    return {"workload": sw["workload"], "G": sw["G"], "full_frame_kernel_ms": sw["full_frame"]["kernel_ms"],
            "full_frame_kernel_ms_min_max": sw["full_frame"]["kernel_ms_min_max"],
            "shard_kernel_ms": [round(x["kernel_ms"], 2) for x in sw["shards"]],
            "shard_kernel_ms_min": [round(x["kernel_ms_min_max"][0], 2) for x in sw["shards"]],
            "shard_kernel_ms_max": [round(x["kernel_ms_min_max"][1], 2) for x in sw["shards"]],
            "renders_per_shard": sw["shards"][0]["renders"],
            "predicted_speedup": sw["predicted_speedup"],
            "predicted_speedup_worst_single_render": sw["predicted_speedup_worst_single_render"],
            "pixels_per_lane_per_shard": sw["shards"][0]["pixels_per_lane"],
            "gather_bytes_per_rank": sw["shards"][0]["gather_bytes"], "ray_total_matches": sw["ray_total_matches"],
            "note": sw["note"]}


def selftest_exchange(args):
    """The N-rank exchange of one step on CPU tensors over gloo: every rank fills its tile-major
    buffer with the global pixel indices it owns, rank 0 gathers and untiles; no rendering."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from rtmi.dist import gather_to_root, shard_pixel_map, untile_host
    from rtmi.launch import env_rank_world
    rank, _, world = env_rank_world()
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    if rank == args.selftest_die_rank:
        raise SystemExit(3)  # a rank that dies before the rendezvous: the launcher must not wait for the others
    dist.init_process_group("gloo", rank=rank, world_size=world)
    h, w = 40, 56
    pm = shard_pixel_map(h, w, rank, world)
    tiles = np.full((pm.size, 3), -1.0, dtype=np.float32)
    tiles[pm >= 0] = pm[pm >= 0, None].astype(np.float32) + np.array([0.0, 0.25, 0.5], dtype=np.float32)
    allt = gather_to_root(torch.from_numpy(tiles), 0)
    tot = torch.tensor([float((pm >= 0).sum())], dtype=torch.float64)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    ok = int(tot.item()) == h * w
    if rank == 0:
        img = untile_host(allt.numpy(), h, w, world)
        want = np.arange(h * w, dtype=np.float32).reshape(h, w, 1) + np.array([0.0, 0.25, 0.5], dtype=np.float32)
        ok = ok and bool(np.array_equal(img, want))
        print(json.dumps({"selftest": "ok" if ok else "FAILED", "n_gpus": world, "n_ranks_seen": dist.get_world_size(),
                          "scaling": "strong"}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    if not ok:
        raise SystemExit(1)


def main():
    args = parse()
    from rtmi import launch
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and not launch.under_launcher():
        # become the launcher: N children, one per GPU, before this process touches a GPU
        rc = launch.spawn_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:],
                                need_gpus=not args.selftest_exchange, timeout=args.rank_timeout)
        raise SystemExit(rc)
    if args.selftest_exchange:
        if not launch.under_launcher():
            os.environ.update({"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1",
                               "MASTER_PORT": str(launch.free_port())})
        return selftest_exchange(args)

    import torch
    import torch.distributed as dist
    import rtmi

    rank, local_rank, world = launch.env_rank_world()
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d: refusing to run a different number of ranks" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    # TEST HOOK (tests/test_gpu_round3.py): RTMI_BENCH_TEST_ONE_GPU=1 runs every rank on cuda:0 and the exchange over
    # gloo, so that the N-rank code of this file runs with real rendering on a one-GPU box (RCCL refuses two ranks on
    # one device).  Its line is marked "test_one_gpu": it is not a measurement.
    one_gpu_test = os.environ.get("RTMI_BENCH_TEST_ONE_GPU") == "1"
    if one_gpu_test:
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit("rank %d needs GPU %d but only %d visible" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    use_dist = launch.under_launcher()  # under torch.distributed.run even with one rank
    if use_dist and one_gpu_test:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    elif use_dist:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if args.blocks_per_cu or args.threads:
        rtmi.lib().rtmi_set_launch(args.blocks_per_cu, args.threads)

    w = pick_workload(args, world)
    if args.shard_sweep:
        if world != 1:
            raise SystemExit("--shard-sweep runs on one GPU")
        print(json.dumps({"shard_sweep": shard_sweep(rtmi, torch, w, args.shard_sweep, args)}), flush=True)
        return
    shard = None
    if args.shard:
        if world != 1:
            raise SystemExit("--shard r/G renders one shard on ONE GPU")
        shard = tuple(int(x) for x in args.shard.split("/"))
        if len(shard) != 2 or not 0 <= shard[0] < shard[1]:
            raise SystemExit("--shard wants r/G with 0 <= r < G")
    r = run_workload(rtmi, torch, dist, w, args, rank, world, use_dist, args.steps, args.warmup, shard=shard)

    if rank == 0:
        value = r["rays_all"] * args.steps / r["dt"] / 1e6
        lanes = r["shape"]["blocks"] * r["shape"]["threads"]
        out = {
            "metric": "Mrays/s", "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": r["dt"] / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            **({"test_one_gpu": "all ranks on cuda:0, exchange over gloo: a functional test, not a measurement"}
               if os.environ.get("RTMI_BENCH_TEST_ONE_GPU") == "1" else {}),
            "config": {
                "workload": "%s: scenes/%s %dx%d x%dspp depth%d seed%d (fixed frame; 8x8 tiles interleaved over %d "
                            "rank(s)%s; step = trace kernel%s)" %
                            (w["name"], w["scene"], r["H"], r["W"], w["spp"], w["depth"], r["seed"],
                             shard[1] if shard else world, ", shard %d only" % shard[0] if shard else "",
                             " + RCCL gather + untile" if world > 1 else " + untile"),
                "rays_per_step": r["rays_all"], "msamples_per_s": (r["pixels_rank"] if shard else r["H"] * r["W"]) * w["spp"] * args.steps / r["dt"] / 1e6,
                "bytes_per_ray": r["bytes_per_ray"], "kernel_ms": max(r["per_rank_ms"]),
                "kernel_ms_per_rank": r["per_rank_ms"], "n_ranks_seen": r["n_seen"],
                "resident_lanes_per_rank": lanes, "pixels_per_lane": r["pixels_rank"] / float(lanes),
                "schedule": r["mode"],  # rtmi_render_mode: first pass, planned chains or queue, priorities, lane stride
            },
            # (N > 1: rank 0's kernel is shard 0 of N of the frame -- the digest of exactly that shard, where there is one)
            "roofline": roofline(w, r["rays_rank"], r["kern_ms"], r["bytes_per_ray"], r["n_simd"],
                                 shard or ((rank, world) if world > 1 else None)),
        }
    if world == 1 and not args.no_extra and w["name"] == "c2" and not shard:
        # the other BASELINE configs one GPU carries, beside the headline: same contract, shorter runs.
        # configs[2] (mesh + BVH); configs[0] (spheres) and its 1024x1024 shape; configs[3] / configs[4] as the
        # shard one GPU of eight renders, at their full sample counts
        extras = []
        for tag, wl, sh, n in (("c3", "c3", None, 5), ("c1", "c1", None, 5), ("c1big", "c1big", None, 3),
                               ("c4", "c4", (0, 8), 2), ("c5", "c5", (0, 8), 1)):
            wx = dict(WORKLOADS[wl], name=tag)
            nx = max(1, min(args.steps, n))
            rx = run_workload(rtmi, torch, dist, wx, args, rank, world, use_dist, nx, 1, shard=sh)
            if rank == 0:
                extras.append(extra_line(tag, wx, rx, nx, sh))
        if rank == 0:
            out["config"]["extra"] = extras
        # what an 8-GPU node can reach on the N > 1 workload (C4), measured on this one GPU: full frame against each
        # of the eight shards (kernel only; the N-rank step adds one gather of ~6 MB per rank and the untile kernel)
        sw = shard_sweep(rtmi, torch, dict(WORKLOADS["c4"], name="c4"), 8, args)
        # ... and on configs[4] (C5, birthday 4096^2 x 8192 spp: eight pixels per lane per shard; 25 s per full frame, so
        # one render of it and of each shard)
        sw5 = shard_sweep(rtmi, torch, dict(WORKLOADS["c5"], name="c5"), 8, args, rounds=1, full_rounds=1)
        if rank == 0:
            out["config"]["shard_sweep"] = sweep_digest(sw)
            out["config"]["shard_sweep_c5"] = sweep_digest(sw5)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
