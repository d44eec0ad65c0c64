// Internal launch interface between the C ABI (capi.hip) and the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rtmi.h"  // RTMI_COUNTER_WORDS
#include "scene_dev.h"

#define RTMI_KERNEL_MAX_DEPTH 64  // == RTMI_MAX_DEPTH of include/rtmi.h

namespace rtmi {

int64_t frame_pixel_of(const FrameDev &fr, int rank, int64_t q);

hipError_t launch_rng_init(uint64_t seed, const FrameDev &fr, const uint32_t *d_jump, uint32_t *d_states,
                           hipStream_t stream);

// Kernel specialisation covering a feature set, its occupancy, and its launch.
uint32_t pick_variant(uint32_t features);
int render_occupancy(uint32_t variant, const SceneDev &sc, const FrameDev &fr, int threads);
// d_tile_order (nullable): the work queue hands out local tile d_tile_order[k] as its k-th tile.
// d_sparse_items (nullable; needs d_tile_order): device word, how many leading work items are outlier
// tiles that mesh kernels spread one pixel per tune.sparse_stride lanes (launch_tile_order writes it to d_max[1]).
// probe: launch under the probe_kernel name (the scheduler's cost-estimation pass).
// Per-call scheduling parameters (capi.hip fills them from rtmi_render_opts over the process defaults).
struct RenderTuning {
  int schedule;       // 0 image order, 1 longest-first when it can pay, 2 always longest-first
  int blocks_per_cu;  // 0: as many as fit
  int threads;        // 0: chosen per kernel variant
  int sparse_stride;  // outlier tiles of mesh frames: one pixel per this many lanes (power of two, 1..64)
  int exclusive;      // 1: a wave holding an outlier pixel takes no other new pixels (its lanes work for it)
  int outlier_x10;    // a tile is an outlier from this many tenths of the mean tile cost
  int head_pct[3];    // mesh frames: per cent of the frame's largest probe count from which a pixel gets a wave to itself,
                      // shares one with another, gets one lane in 16 (80 / 55 / 30)
  int probe_spp;      // samples per pixel of the scheduler's cost probe; 0: chosen per frame (capi.hip)
  int promote;        // samples after which a mesh frame's pixel may be promoted to a head class by its own ray count; 0: never
  int first_pass;     // 0: the scheduler's probe is rendered on scratch copies and discarded; 1: it is the frame's own first spp / 16
                      // samples (the second launch resumes); N > 1: spp / N
  int cost_probe;     // mesh frames: 1 = the probe books its searches' lane-steps per pixel and the queue follows that cost
  int lane_stride;    // list frames smaller than the grid: one pixel per this many lanes (power of two; 0: chosen per frame)
  int plan;           // 1: list frames with a probe behind them are rendered as planned chains (launch_chain_plan), not from the queue
  int prio_every;     // > 0: the waves of a SIMD are served longest-remaining-chain-first, each looking at its priority every
                      // this many iterations (a power of two); 0: the hardware's oldest-first order
};
// What the scheduler's probe pass leaves for the real pass (device pointers, all optional).
struct SchedPlan {
  const uint32_t *tile_order = nullptr;    // the queue's order per quarter tile (launch_quarter_order)
  uint32_t *visit_counts = nullptr;        // probe pass of a mesh frame: receives per work item the lane-steps of its searches
  const uint32_t *sparse_items = nullptr;  // one word: leading work items handed to every sparse_stride-th lane only
                                           // (with head_list: + [1], [2] = ends of its 64- and 32-lane classes)
  const uint32_t *head_list = nullptr;     // optional: the head's work items, heaviest pixels first (kHeadCap words)
  const uint32_t *probe_marks = nullptr;   // with head_list: per work item, bit 31 set = listed in the head
  int probe_spp = 2;                       // samples per pixel of the probe behind these (thresholds: sparse_items[23..25])
  uint32_t *prio_tab = nullptr;            // optional: the wave-priority table (render_body.h: kPrioRows x 16 words, zeroed)
  const uint32_t *tile_cost = nullptr;     // optional: the probe's ray count per tile (launch_tile_order's d_cost)
  // planned chains (launch_chain_plan): all or none; with them tile_order is per TILE and prio_tab is required
  const int32_t *chain_next = nullptr;
  const uint32_t *chain_fut = nullptr;
  const int32_t *chain_first = nullptr;
  uint32_t *claims = nullptr;              // one word per tile, zeroed
  int plan_simds = 0, plan_rounds = 0;
};
// d_params: render_params_bytes() of device memory that stays untouched until the launch has finished (the kernel's
// argument block, written in stream order just before it).
size_t render_params_bytes();
hipError_t launch_render(uint32_t variant, const SceneDev &sc, const FrameDev &fr, uint32_t *d_states, float *d_out,
                         uint32_t *d_ray_counts, unsigned long long *d_counters, const SchedPlan &plan, bool probe,
                         int blocks, int threads, const RenderTuning &tune, void *d_params, hipStream_t stream);
// Tiles sorted by descending cost (sum of 64 ray counts each); d_cost/d_order hold n_tiles words,
// d_meta 16: [0] the largest tile cost, [1] the sparse item count.
// sparse_cap: work items the grid holds at one pixel per tune.sparse_stride lanes (a multiple of 64).
// d_head (nullable, kHeadCap words): the head's work items sorted into three classes by their own probe count --
// pixels that get a wave each, pixels that share one between two, the rest (one per 16 lanes); d_meta[2], [3] =
// where the first two classes end.  grid_waves: waves of the render launch (the classes may use a quarter of them).
constexpr int kHeadCap = 16384;
hipError_t launch_tile_order(uint32_t *d_ray_counts, int n_tiles, uint32_t *d_cost, uint32_t *d_meta,
                             uint32_t *d_order, uint32_t *d_head, uint32_t sparse_cap, int grid_waves, int outlier_x10,
                             const int head_pct[3], hipStream_t stream);

// The per-quarter-tile order the trace kernel's queue follows (4 * n_tiles words): d_order's tiles with their quarters in
// sequence, or -- d_work != nullptr: mesh frames with a cost probe -- the quarters sorted by probed cost and dealt to the
// 64-item blocks in a snake (kernels.hip).  d_qcost / d_qsorted: 4 * n_tiles words of scratch each, d_qmax one word.
hipError_t launch_quarter_order(const uint32_t *d_order, const uint32_t *d_work, const uint32_t *d_rays, int n_tiles,
                                uint32_t *d_qcost, uint32_t *d_qsorted, uint32_t *d_qmax, uint32_t *d_qmap, hipStream_t stream);

// Planned chains for list frames.  The tiles in longest-first order (d_order) are dealt to the SIMDs in a snake -- SIMD s of
// S gets ranks s, 2S - 1 - s, 2S + s, ... -- so that every SIMD's share costs about the same, and a SIMD's share is dealt
// to its R waves in a snake again (its k-th tile goes to wave k, 2R - 1 - k, 2R + k, ...: the lightest first tiles are
// paired with the tiles of the second round).  Chain c = wave * S + SIMD.  d_first[c] = its first tile or -1,
// d_next[tile] = the tile after it in its chain (-1: none), d_fut[tile] = estimated queries per lane of the tiles after
// it: d_cost x spp / (64 probe_spp).  Which wave of which SIMD a wave IS it finds out when it starts (render_body.h).
hipError_t launch_chain_plan(const uint32_t *d_order, const uint32_t *d_cost, int n_tiles, int simds, int rounds, int spp,
                             int probe_spp, int32_t *d_first, int32_t *d_next, uint32_t *d_fut, hipStream_t stream);

hipError_t launch_untile(const FrameDev &fr, const float *d_tiles, float *d_image, hipStream_t stream);
hipError_t launch_untile_u32(const FrameDev &fr, const uint32_t *d_tiles, uint32_t *d_image, hipStream_t stream);
hipError_t launch_post(float *d_img, int64_t n, int spp, hipStream_t stream);
#ifdef RTMI_STATS
hipError_t copy_wave_stats(unsigned long long *host, size_t bytes);  // diagnostic builds only
#endif
// d_bad[4]: see arithmetic_selftest in kernels.hip.
hipError_t launch_arithmetic_selftest(unsigned long long *d_bad, hipStream_t stream);

}  // namespace rtmi
