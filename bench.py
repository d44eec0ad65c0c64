#!/usr/bin/env python3
"""bench.py — Mrays/s of the path-tracing hot path on MI355X.

One *step* = one frame of the workload through the C ABI: trace kernel for this rank's
pixel tiles (+ RCCL gather of the tile buffers to rank 0 and the untile kernel when
N > 1).  Inputs (scene, RNG states) are resident in HBM before the timed region.

Workload at N=1 (BASELINE.json configs[1]): scenes/cornell_box, 1024x1024, 1024 spp,
depth limit 50, seed 1024.  For N > 1 the job is weak-scaled: the same scene and aspect
at side 1024*sqrt(N) (rounded to a multiple of 8), so every GPU keeps 1024^2 pixels of
1024 spp; tiles are interleaved over ranks and nothing but the final gather is exchanged.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     — algorithmic bytes per launch (rays x bytes/ray) over the
                 trace kernel's mean duration measured with HIP events on its stream;
  cpu_baseline — the CPU oracle ("port") on this box's cores on a bounded sample.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing-cuda_amd"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="cornell_box", choices=["cornell_box", "spheres", "bunny", "birthday"])
    ap.add_argument("--size", type=int, default=1024, help="per-GPU frame side (pixels)")
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", default="256x256x32",
                    help="HxWxSPP of the CPU-baseline sample (default: 256x256, spp sized for ~15 s)")
    ap.add_argument("--blocks-per-cu", type=int, default=0)
    ap.add_argument("--threads", type=int, default=0)
    return ap.parse_args()


def build_scene(rtmi, name, aspect, seed):
    from rtmi import scenes
    b = rtmi.SceneBuilder(seed)
    if name == "cornell_box":
        scenes.cornell_box(b, aspect)
    elif name == "spheres":
        scenes.spheres(b, aspect)
    elif name == "bunny":
        scenes.bunny(b, aspect, scenes.procedural_bunny_mesh())
    elif name == "birthday":
        scenes.birthday(b, aspect, scenes.procedural_earthmap(1024, 2048))
    return b


def cpu_baseline(args, name, seed):
    """Oracle (CPU restatement, kind "port") on the host cores, bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oraclelib
    from rtmi import scenes
    h, w, spp = (int(x) for x in args.cpu_sample.split("x"))
    b = oraclelib.OracleBuilder(seed)
    if name == "cornell_box":
        scenes.cornell_box(b, w / h)
    elif name == "spheres":
        scenes.spheres(b, w / h)
    elif name == "bunny":
        scenes.bunny(b, w / h, scenes.procedural_bunny_mesh())
    else:
        scenes.birthday(b, w / h, scenes.procedural_earthmap(1024, 2048))
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    # pilot run to size the sample for roughly 15 s of CPU work (bounded 10-30 s)
    t0 = time.time()
    _, _, _, rays0 = b.render(128, 128, 8, args.depth, threads=cores)
    rate = rays0 / max(time.time() - t0, 1e-3)
    rays_per_sample = rays0 / (128 * 128 * 8)
    spp = int(max(4, min(8192, 15.0 * rate / (h * w * rays_per_sample))))
    if args.cpu_sample != "256x256x32":
        spp = int(args.cpu_sample.split("x")[2])
    t0 = time.time()
    _, _, _, rays = b.render(h, w, spp, args.depth, threads=cores)
    dt = time.time() - t0
    if dt < 8.0 and args.cpu_sample == "256x256x32":  # the pilot under-estimated the rate: one re-run, sized from dt
        spp = int(min(8192, spp * 14.0 / max(dt, 0.1)))
        t0 = time.time()
        _, _, _, rays = b.render(h, w, spp, args.depth, threads=cores)
        dt = time.time() - t0
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "%s %dx%d x%dspp depth%d, %d rays in %.1fs (oracle, g++ -O2 -ffp-contract=off, %d threads)" %
                      (name, h, w, spp, args.depth, rays, dt, cores)}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import rtmi
    from rtmi import scenes
    from rtmi.dist import env_rank_world, gather_to_root

    rank, local_rank, world = env_rank_world()
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or "RANK" in os.environ  # under torch.distributed.run even with one rank
    if use_dist:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if args.blocks_per_cu or args.threads:
        rtmi.lib().rtmi_set_launch(args.blocks_per_cu, args.threads)

    side = int(round(args.size * math.sqrt(world) / 8.0)) * 8
    H = W = side
    seed = scenes.SCENE_SEEDS[args.scene]
    scene = build_scene(rtmi, args.scene, W / H, seed).commit()
    bytes_per_ray = scene.bytes_per_ray()
    R = rtmi.Renderer(scene, H, W, args.spp, args.depth, True, rank=rank, world_size=world)
    R.init_rng()
    pristine = R.states.clone()
    torch.cuda.synchronize()

    def step(ev=None):
        R.states.copy_(pristine)
        if ev:
            ev[0].record()
        R.render()
        if ev:
            ev[1].record()
        if use_dist:
            allt = gather_to_root(R.tiles, 0)
            if rank == 0:
                R.untile(allt)
        else:
            R.untile()

    for _ in range(args.warmup):
        step()
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(events[i])
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0

    rays_rank = R.total_rays()  # rays of one step on this rank (identical every step)
    kern_ms = sum(a.elapsed_time(b) for a, b in events) / max(1, args.steps)
    tt = torch.tensor([dt, float(rays_rank), kern_ms], dtype=torch.float64, device="cuda")
    if use_dist:
        tmax = tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tt.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt = float(tmax[0])
        rays_all = float(tsum[1])
        kern_ms = float(tmax[2])
    else:
        rays_all = float(rays_rank)

    if rank == 0:
        # HBM bytes per launch from the PMC passes committed under profiles/ (rocprofv3 cannot
        # run inside this process); scaled by this run's ray count, scene must match.
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
            if tj.get("workload_scene") == args.scene:
                traffic = tj["hbm_bytes_per_ray"] * rays_rank
        except (OSError, ValueError, KeyError):
            pass
        value = rays_all * args.steps / dt / 1e6
        achieved = rays_rank * bytes_per_ray / (kern_ms * 1e-3) / 1e9  # GB/s, dominant kernel on this rank
        out = {
            "metric": "Mrays/s", "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "scenes/%s %dx%d x%dspp depth%d seed%d (%dx%d px per GPU, 8x8 tiles interleaved over %d "
                            "rank(s))" % (args.scene, H, W, args.spp, args.depth, seed, args.size, args.size, world),
                "rays_per_step": rays_all, "msamples_per_s": H * W * args.spp * args.steps / dt / 1e6,
                "bytes_per_ray": bytes_per_ray, "kernel_ms": kern_ms,
            },
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, args.scene, seed)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
