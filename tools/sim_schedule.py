"""Offline list-scheduling simulation on measured per-pixel ray counts (tools/gpu_dump_counts.py):
makespan of the work queue in units of the ideal (total rays / lanes) for image order, tile-level
longest-first (what the kernel does), pixel-level longest-first with exact and with noisy costs.
Model: a lane advances one ray per iteration, all iterations cost the same (the real machine speeds
up as waves drain, so measured tails are shorter than these ratios)."""
import numpy as np, heapq
c64 = np.load('/root/repo/gpurun_out/c2_counts_64spp.npy').astype(np.float64)   # per work item (tile-major)
c2 = np.load('/root/repo/gpurun_out/c2_counts_2spp.npy').astype(np.float64)
cost = c64 * 16.0   # stand-in for 1024 spp (same distribution, chains 16x longer)
n = cost.size
print("pixels", n, "mean %.0f max %.0f min %.0f p99 %.0f" % (cost.mean(), cost.max(), cost.min(), np.percentile(cost, 99)))
def simulate(order, lanes):
    # list scheduling: lanes pull next item in 'order' when free; time = rays processed (1 ray per iteration)
    heap = [0.0] * lanes
    heapq.heapify(heap)
    for q in order:
        t = heapq.heappop(heap)
        heapq.heappush(heap, t + cost[q])
    return max(heap)
for lanes in (7 * 4 * 256 * 64, 6 * 4 * 256 * 64, 4 * 4 * 256 * 64):
    ideal = cost.sum() / lanes
    img = simulate(np.arange(n), lanes)
    tiles2 = c2.reshape(-1, 64).sum(1)
    order_t = np.argsort(-tiles2, kind='stable')
    tile_lpt = simulate((order_t[:, None] * 64 + np.arange(64)[None, :]).reshape(-1), lanes)
    tiles_true = cost.reshape(-1, 64).sum(1)
    order_tt = np.argsort(-tiles_true, kind='stable')
    tile_lpt_true = simulate((order_tt[:, None] * 64 + np.arange(64)[None, :]).reshape(-1), lanes)
    pix_lpt = simulate(np.argsort(-cost, kind='stable'), lanes)
    print("lanes %d: ideal %.0f | image order %.3f | tile LPT (2spp est) %.3f | tile LPT (true) %.3f | pixel LPT (true) %.3f | max pixel/ideal %.3f" % (
        lanes, ideal, img / ideal, tile_lpt / ideal, tile_lpt_true / ideal, pix_lpt / ideal, cost.max() / ideal))

# what noisy per-pixel estimates achieve: estimate from k spp = poisson-ish noise around true
rng = np.random.default_rng(0)
lanes = 458752
ideal = cost.sum() / lanes
for spp_est in (2, 8, 32):
    # emulate: estimate = true/1024*spp_est with relative noise ~ 1.2/sqrt(spp_est)
    est = cost * (1 + rng.normal(0, 1.2 / np.sqrt(spp_est), n))
    tile_mean = np.repeat(cost.reshape(-1, 64).mean(1), 64)
    for blend in (0.0, 0.5):
        key = (1 - blend) * est + blend * tile_mean
        print("est %2d spp blend %.1f: pixel-LPT makespan %.3f" % (spp_est, blend, simulate(np.argsort(-key, kind='stable'), lanes) / ideal))
