// The persistent per-pixel trace loop.  Included by kernels.hip inside namespace rtmi, after closest_hit.h
// (not a stand-alone header).
#pragma once

// ================================================================== trace kernel
// Dynamic LDS: [ material records: lds_mats * 32 B ][ id stack: max_depth * blockDim entries ]
// The id stack is laid out [depth][thread] so the lanes of a wave touch consecutive
// bytes; entries are 4 bits when there are at most 16 materials (two levels per byte, lc.wide_ids == 2:
// half the LDS, which is what lets a sixth wave per SIMD of the list kernel in at depth 50), uint8 when
// every material id fits a byte, else uint16 (lc.wide_ids == 1).
struct LaunchCfg {
  int32_t lds_mats;    // materials staged in LDS (0: read them from global memory)
  int32_t wide_ids;    // 0: uint8 stack entries, 1: uint16, 2: 4-bit (two levels per byte)
  int32_t stack_off;   // byte offset of the id stack inside dynamic LDS
  int32_t nodes_off;   // byte offset of the staged reference-tree nodes
  int32_t lds_nodes;   // reference-tree nodes staged in LDS (the first lds_nodes of SceneDev::nodes)
  int32_t mesh_off;    // byte offset of the per-wave mesh-search regions (kMeshWaveWords words each; BVH variants)
  int32_t exclusive;   // 1: while a wave holds an outlier pixel, its other lanes take no new pixels (they work for it)
  int32_t pairs_off;   // byte offset of the staged PairPts records, -1: not staged (the culled scan, if on, gathers them
                       // from global memory)
  int32_t list_off;    // byte offset of the per-wave regions of the shared candidate tests, -1: each lane tests its own
  int32_t paths_off;   // byte offset of the staged leaf-path words
  int32_t lds_paths;   // leaf-path words staged in LDS (the first lds_paths of SceneDev::leaf_paths)
  int32_t threads;     // lanes per workgroup (== blockDim.x)
  int32_t nrm_off;     // byte offset of the staged TriNrm records (with pairs_off), -1: not staged
  int32_t probe_spp;   // samples per pixel of the probe whose counts the head classes' thresholds refer to
  int32_t promote;     // > 0: mesh frames promote pixels to a head class at run time, from this many samples on (render_body: thr16)
  int32_t cand_off;    // byte offset of the per-lane candidate slots of the grouped sphere scan, -1: none
  const uint32_t *tile_order;  // optional: the queue's order, per QUARTER tile (16 work items): its k-th 16 items are
                               // quarter tile_order[k] (items tile_order[k] * 16 ...)
  uint32_t *visit_counts;      // optional (cost probe, mesh variants): per work item, lane-steps of its rays' searches
  const uint32_t *head_list;     // optional (with sparse_items): the head's items by weight class (SchedPlan::head_list)
  const uint32_t *probe_marks;   // with head_list: bit 31 of an item's word = it is in the head
  const uint32_t *sparse_items;  // optional (with tile_order): leading work items handed to every sparse_stride-th lane only
  int32_t sparse_stride;         // power of two (RenderTuning::sparse_stride)
  int32_t lane_stride;           // list variants, frames smaller than the grid: only every lane_stride-th lane takes pixels
                                 // (a power of two; the others are workers of the shared candidate tests)
  int32_t prio_every;            // with prio_tab: a wave looks at its priority every this many iterations (power of two)
  uint32_t *prio_tab;            // optional: kPrioRows x 16 words, zeroed per launch -- per SIMD (row: XCC | SE | SH | CU | SIMD
                                 // of HW_ID) and wave slot (column: HW_ID.WAVE_ID), the queries the wave still has to do
  // Planned chains (list variants with prio_tab; kernels.hip: chain_link_kernel): instead of the queue every wave walks
  // a chain of tiles fixed before the launch, lane l rendering pixel l of each.  The chains are laid out per SIMD: a
  // wave learns at its start which SIMD it runs on and how many waves came there before it (prio_tab columns 14, 15),
  // SIMDs are numbered in order of arrival (counters[35]), and chain `round x plan_simds + simd` is the wave's.
  const int32_t *chain_next;     // optional: per local tile, the tile that follows it in its chain, -1: the last
  const uint32_t *chain_fut;     // per tile, estimated queries per lane of the tiles after it in its chain
  const int32_t *chain_first;    // per chain, its first tile (-1: an empty chain)
  int32_t plan_simds, plan_rounds;  // the plan's shape: SIMDs x waves per SIMD
  uint32_t *claims;              // one word per local tile, zeroed per launch: the wave (index + 1) that renders it.  A
                                 // wave that has nothing left takes the lightest tile nobody has started (cursor:
                                 // counters[36], from the far end of tile_order, which in this mode is per TILE), so
                                 // every tile is rendered whatever the hardware's placement of the waves was
  int32_t fetch_batch;           // list variants, queue mode: items a wave takes from the queue per atomic, at most (1..64)
  const uint32_t *tile_cost;     // optional: the probe's ray count per tile (64 pixels x probe_spp samples): a pixel's
  float rate_scale;              // rays per sample are first taken as tile_cost x rate_scale = 1 / (64 probe_spp)
};

// Longest-remaining-chain-first between the waves of a SIMD (DESIGN "Wave priorities").  A SIMD's issue arbiter serves
// the highest s_setprio level first and the OLDEST wave within a level: left alone, the first-dispatched wave of a
// SIMD runs at the speed of a wave that has the SIMD to itself and the last at a third of it, whatever they hold.  A
// frame ends with its last wave, so the wave with the longest chain of queries still ahead of it should be the one
// that is served first.  Every wave publishes that figure -- the largest, over its lanes, of (rays per sample so far) x
// (samples left) -- in its SIMD's row of a table in global memory, reads the rows' other entries and takes the level
// its rank gives it.  Only which wave issues first changes: no pixel's arithmetic does.
__device__ __forceinline__ void wave_priority_update(uint32_t *tab, uint32_t left) {
  for (int off = 32; off > 0; off >>= 1) left = max(left, (uint32_t)__shfl_xor((int)left, off));
  const uint32_t mine = (uint32_t)__builtin_amdgcn_readfirstlane((int)left) + 1u;  // (a live wave is never 0)
  const uint32_t hw = __builtin_amdgcn_s_getreg(0xF804), xcc = __builtin_amdgcn_s_getreg(0xF814);  // HW_ID, XCC_ID
  const uint32_t row = ((xcc & 15u) << 10) | (((hw >> 8) & 0xffu) << 2) | ((hw >> 4) & 3u);
  const uint32_t col = hw & 15u;  // WAVE_ID: the wave's slot on its SIMD
  if (col >= 12u) return;         // (columns 12..15 are not wave slots: such a wave keeps the priority it has)
  uint32_t *rowp = tab + row * 16u;
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t other = 0u;
  // (columns 0..11: WAVE_ID is below 10 on this part; 14 and 15 belong to the planned chains' registration)
  // Workgroup scope: a row's readers and writers are the waves of ONE SIMD, so they share their compute unit's vector
  // L1 and an access need not go further (agent scope is a round trip to the far side of the L2 every time: C4 shard
  // 486.6 -> 481.3 ms, C1 1.85 -> 1.82, C2 234.3 -> 233.6).  The waves belong to different workgroups, which the
  // memory model does not promise this scope for -- a stale entry costs a wave its rank until it looks again, never a pixel.
  if (lane < 12u) other = __hip_atomic_load(rowp + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (lane == col) __hip_atomic_store(rowp + lane, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  const bool ahead = lane < 12u && lane != col && (other > mine || (other == mine && lane < col));
  const int rank = __builtin_amdgcn_readfirstlane(__popcll(__builtin_amdgcn_ballot_w64(ahead)));
  // level by rank: 3, 2, 1, 1, 0, 0, ... (within a level the arbiter serves the older wave first)
#ifndef RTMI_PRIO_LEVELS
#define RTMI_PRIO_LEVELS {3, 2, 1, 1, 0, 0, 0, 0}
#endif
  constexpr int levels[8] = RTMI_PRIO_LEVELS;
  const int level = rank == 0 ? levels[0] : rank == 1 ? levels[1] : rank == 2 ? levels[2] : rank == 3 ? levels[3]
                  : rank == 4 ? levels[4] : rank == 5 ? levels[5] : rank == 6 ? levels[6] : levels[7];
  // (s_setprio takes an immediate and ignores EXEC: one scalar branch per level)
  if (level == 3) __builtin_amdgcn_s_setprio(3);
  else if (level == 2) __builtin_amdgcn_s_setprio(2);
  else if (level == 1) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(0);
}
__device__ __forceinline__ int wave_chain_id(uint32_t *tab, unsigned long long *simd_counter, int plan_simds, int plan_rounds) {
  const uint32_t hw = __builtin_amdgcn_s_getreg(0xF804), xcc = __builtin_amdgcn_s_getreg(0xF814);
  const uint32_t row = ((xcc & 15u) << 10) | (((hw >> 8) & 0xffu) << 2) | ((hw >> 4) & 3u);
  uint32_t *rowp = tab + row * 16u;
  uint32_t round = 0u, simd = 0u;
  if ((threadIdx.x & 63u) == 0u) {
    round = atomicAdd(rowp + 15, 1u);
    if (round == 0u) {
      simd = (uint32_t)atomicAdd(simd_counter, 1ull);
      __hip_atomic_store(rowp + 14, simd + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while ((simd = __hip_atomic_load(rowp + 14, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u) __builtin_amdgcn_s_sleep(2);
      simd -= 1u;
    }
  }
  round = (uint32_t)__builtin_amdgcn_readfirstlane((int)round), simd = (uint32_t)__builtin_amdgcn_readfirstlane((int)simd);
  return (round < (uint32_t)plan_rounds && simd < (uint32_t)plan_simds) ? (int)(round * (uint32_t)plan_simds + simd) : -1;
}
__device__ __forceinline__ void wave_priority_leave(uint32_t *tab) {
  const uint32_t hw = __builtin_amdgcn_s_getreg(0xF804), xcc = __builtin_amdgcn_s_getreg(0xF814);
  const uint32_t row = ((xcc & 15u) << 10) | (((hw >> 8) & 0xffu) << 2) | ((hw >> 4) & 3u);
  if ((threadIdx.x & 63u) == 0u && (hw & 15u) < 12u)
    __hip_atomic_store(tab + row * 16u + (hw & 15u), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <uint32_t F>
__device__ __forceinline__ void render_body(const SceneDev &sc, const FrameDev &fr, const LaunchCfg &lc,
                                            uint32_t *__restrict__ states, float *__restrict__ out,
                                            uint32_t *__restrict__ ray_counts,
                                            unsigned long long *__restrict__ counters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  MatRec *s_mats = reinterpret_cast<MatRec *>(smem);
  // id stack: byte offset of entry [level][thread] in LDS, kept as 32-bit arithmetic (pointer
  // arithmetic on the generic pointers costs a register pair per live address)
  // lanes per workgroup, from the kernel arguments: blockDim.x is a 16-bit field of the dispatch packet, which the
  // compiler reads with a VECTOR load (global_load_ushort + s_waitcnt vmcnt(0)) at every use inside the loop --
  // three round trips to memory per iteration before round 3
  const uint32_t n_threads = (uint32_t)lc.threads;
  const uint32_t ids_shift = lc.wide_ids == 1 ? 1u : 0u;
  const bool nibble_ids = lc.wide_ids == 2;
  auto ids_offset = [&](int level) -> uint32_t {  // (nibble_ids: the byte of levels 2k and 2k + 1 is row k)
    return (uint32_t)lc.stack_off + (((uint32_t)level * n_threads + threadIdx.x) << ids_shift);
  };
  const BvhNode *s_nodes = reinterpret_cast<const BvhNode *>(smem + lc.nodes_off);
  int *wl = nullptr;  // this wave's mesh-search region
  if (F & F_BVH)
    wl = reinterpret_cast<int *>(smem + lc.mesh_off) +
         __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * kMeshWaveWords;
  const float4 *s_pairs = nullptr;  // corners of the world-list pairs (culled scan) or nullptr (plain scan)
  if ((F & F_TRIS) && lc.pairs_off >= 0) {
    s_pairs = reinterpret_cast<const float4 *>(smem + lc.pairs_off);
    const uint32_t *src = reinterpret_cast<const uint32_t *>(sc.pair_pts);
    uint32_t *dst = reinterpret_cast<uint32_t *>(smem + lc.pairs_off);
    for (int w = threadIdx.x; w < sc.n_pairs * 16; w += blockDim.x) dst[w] = src[w];
    const uint32_t *nsrc = reinterpret_cast<const uint32_t *>(sc.tri_nrm);
    uint32_t *ndst = reinterpret_cast<uint32_t *>(smem + lc.nrm_off);
    for (int w = threadIdx.x; w < sc.n_pairs * 8; w += blockDim.x) ndst[w] = nsrc[w];
  }
  const float4 *s_nrm = reinterpret_cast<const float4 *>(smem + (lc.nrm_off >= 0 ? lc.nrm_off : 0));
  int *ll = nullptr;  // this wave's region for the shared candidate tests of the culled list scan
  if ((F & (F_TRIS | F_SGROUP)) && lc.list_off >= 0)
    ll = reinterpret_cast<int *>(smem + lc.list_off) +
         __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * kListWaveWords(F);
  uint16_t *cands = nullptr;  // this wave's candidate slots of the grouped sphere scan
  if ((F & F_SGROUP) && lc.cand_off >= 0)
    cands = reinterpret_cast<uint16_t *>(smem + lc.cand_off) +
            __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * (64 * kSphCand + 128);  // slots + 64 counters
  const bool mats_in_lds = lc.lds_mats > 0;
  auto lds_rgb = [&](int m) -> V3 {  // a staged material's colour: one 16-byte read (MatRec: r, g, b, kind)
    const float4 c = load_lds<float4>(s_mats + m);
    return mk(c.x, c.y, c.z);
  };
  // a byte of dynamic LDS by its byte offset.  The kernels declare no static LDS, so the dynamic array starts at LDS
  // address 0 (group_segment_fixed_size == 0 in the code object: tests/test_host_logic.py) and the offset IS the address.
  auto lds_byte = [&](uint32_t byte_offset) -> uint32_t {
    return *(const RT_LDS uint8_t *)(uintptr_t)byte_offset;
  };
  const bool fast_fold = mats_in_lds && lc.wide_ids != 1 && sc.unsigned_colours;  // see the radiance fold
  if (mats_in_lds) {
    const uint32_t *src = reinterpret_cast<const uint32_t *>(sc.mats);
    uint32_t *dst = reinterpret_cast<uint32_t *>(s_mats);
    for (int w = threadIdx.x; w < lc.lds_mats * 8; w += blockDim.x) dst[w] = src[w];
  }
  if ((F & F_BVH) && lc.lds_nodes > 0) {
    const uint32_t *src = reinterpret_cast<const uint32_t *>(sc.nodes);
    uint32_t *dst = reinterpret_cast<uint32_t *>(smem + lc.nodes_off);
    for (int w = threadIdx.x; w < lc.lds_nodes * 8; w += blockDim.x) dst[w] = src[w];
  }
  const int *s_paths = reinterpret_cast<const int *>(smem + lc.paths_off);
  if ((F & F_BVH) && lc.lds_paths > 0) {
    int *dst = reinterpret_cast<int *>(smem + lc.paths_off);
    for (int w = threadIdx.x; w < lc.lds_paths; w += blockDim.x) dst[w] = sc.leaf_paths[w];
  }
  __syncthreads();

  const int64_t n_items = fr.items;
  const bool w_pow2 = (fr.width & (fr.width - 1)) == 0, h_pow2 = (fr.height & (fr.height - 1)) == 0;
  const double inv_w = 1.0 / (double)fr.width, inv_h = 1.0 / (double)fr.height;
  const bool f32_jitter = w_pow2 && h_pow2 && fr.width <= (1 << 20) && fr.height <= (1 << 20);  // (wave-uniform)
  // per-lane pixel state
  int32_t q32 = -1;  // the lane's work item (items < 2^31: make_frame); widened where it addresses memory; -1: none yet
  uint32_t work_px = 0;  // (cost probe) lane-steps of the mesh searches of this pixel's rays
  uint32_t ray_acc = 0;  // closest-hit queries of this lane's finished pixels, modulo what it has flushed (below)
  uint32_t pij = 0;  // the pixel's row << 16 | column (frames are below 65536 x 65536: make_frame)
  int k = 0;
  bool has_px = false, done = false, active = false;
  bool heavy = false;  // a pixel of the queue's sparse head (see below)
  V3 color = splat(0.f);
  uint32_t rays = 0;
  Rng rng = {0, 0, 0, 0, 0, 0};
  // per-lane path state
  V3 o = splat(0.f), d = splat(0.f);
  int depth = 0;
  // Layer stack of ray_tracing.cuh:9-15.  Layer::emitted is 0 for every material that
  // scatters (only DiffuseLight and Sky emit, and neither scatters), so a layer is its
  // attenuation.  Without image textures the attenuation is the material's constant
  // colour and the layer is stored as a material id in LDS; with image textures the
  // sampled colour itself is kept (private memory).
  // (round 2 kept 64 x 3 floats of private memory per lane for this: 8.2 TB of scratch traffic per launch of a C5
  // shard, 43 % of the waves' time spent waiting.)  Image-textured scenes: a layer is one 32-bit word in LDS,
  // [level][thread] -- the material id, or bit 31 + the sampled texel's three bytes (trace_helpers.h: tex_fetch).
  auto tex_layer_offset = [&](int level) -> uint32_t {
    return (uint32_t)lc.stack_off + (((uint32_t)level * n_threads + threadIdx.x) << 2);
  };

  // Mesh variants, frames dominated by a few outlier tiles (their pixels bounce to the depth limit
  // inside the mesh, tens of times the median cost): the frame time is the serial chain of the
  // slowest pixel, and what shortens a chain is the wave-cooperative search, which needs few rays
  // per wave.  The first sparse_limit work items (the outlier tiles, longest-first order) are
  // therefore spread thin -- one pixel per lc.sparse_stride lanes -- while the rest of the frame runs
  // with full waves.
  unsigned long long sparse_limit = 0ull;
  if ((F & F_BVH) && lc.sparse_items) sparse_limit = *lc.sparse_items;
  // With a head list (scheduler: head_scan / count / plan / scatter kernels) the head's pixels come in three weight classes: one per
  // wave (taken by lane 0), one per 32 lanes (lanes 0, 32), one per 16 lanes.  `cls` = the class of the pixel
  // a lane holds (64 / 32 / 16, 1 for an ordinary pixel); a wave that holds a pixel of class S lets only lanes
  // that are multiples of S take new pixels.  The head has its own queue (counters[3]); the ordinary queue
  // (counters[0]) starts behind it.
  const bool classes = (F & F_BVH) && lc.head_list != nullptr && sparse_limit != 0ull;
  uint32_t end64 = 0u, end32 = 0u;
  if (classes) end64 = lc.sparse_items[1], end32 = lc.sparse_items[2];
  // Promotion at run time (round 3).  The 2-spp probe misses about a third of the outlier pixels (a pixel whose
  // samples go deep with p = 0.4 looks shallow twice with p = 0.36); such a pixel then sits in a full wave -- 50-100 k
  // cycles per query instead of 26 k -- whose other lanes keep taking new pixels, and the frame waits for it.  After 16
  // samples a pixel's own ray count says what the probe could not: from then on a pixel whose rays per sample reach a
  // head class's threshold (the probe's thresholds, per sample) IS of that class, and its wave thins out around it
  // like around any head pixel (`exclusive`).  Nothing moves between lanes; only which lanes may fetch changes.
  uint32_t thr64 = 0xffffffffu, thr32 = 0xffffffffu, thr16 = 0xffffffffu;  // rays per lc.probe_spp samples
  if (classes && lc.exclusive && lc.promote) thr64 = lc.sparse_items[23], thr32 = lc.sparse_items[24], thr16 = lc.sparse_items[25];
  const bool promote = thr16 != 0xffffffffu;
  int cls = 1;
  bool head_open = classes;  // (wave-uniform) the head queue may still hold items
  auto take_item = [&](int64_t item) -> bool {  // false: ragged-tile padding (or nothing to sample), written as black
    q32 = (int32_t)item;
    const int64_t q = item;
    int64_t idx = frame_pixel_of_rank(fr, fr.rank, q);
    if (idx < 0 || fr.spp <= 0) {
      out[q * 3 + 0] = 0.f, out[q * 3 + 1] = 0.f, out[q * 3 + 2] = 0.f;
      if (ray_counts) ray_counts[q] = 0;
      return false;
    }
    pij = ((uint32_t)(idx / fr.width) << 16) | (uint32_t)(idx % fr.width);
    rng.d = states[0 * n_items + q];
    rng.v0 = states[1 * n_items + q];
    rng.v1 = states[2 * n_items + q];
    rng.v2 = states[3 * n_items + q];
    rng.v3 = states[4 * n_items + q];
    rng.v4 = states[5 * n_items + q];
    k = fr.k_begin;
    rays = 0;
    work_px = 0;
    color = splat(0.f);
    if (fr.k_begin > 0) {  // (wave-uniform) resume: the first pass left the pixel's raw sum and ray count in the buffers
      color = mk(out[q * 3 + 0], out[q * 3 + 1], out[q * 3 + 2]);
      rays = ray_counts[q] & 0x7fffffffu;  // (bit 31: the scheduler's head mark on a mesh frame)
    }
    has_px = true;
    return true;
  };

#ifdef RTMI_CHECK_MARGINS
#ifndef RTMI_CHECK_EVERY
#define RTMI_CHECK_EVERY 256u  // (a build with 1 re-does every query: small worlds, tools/gpu_check_margins.py meshes)
#endif
  unsigned check_tick = (blockIdx.x * 7u + (threadIdx.x >> 6)) % RTMI_CHECK_EVERY;  // (wave-uniform) stagger the waves' samples
#endif
  RTMI_STAT(MeshStats st = {}; unsigned wave_queries = 0; const unsigned long long t_begin = stat_real();
            const unsigned long long t_begin_rt = __builtin_amdgcn_s_memrealtime();)
  uint32_t prio_tick = 0u;  // (wave-uniform)
  uint32_t pool_at = 0u, pool_end = 0u;  // (wave-uniform) list variants: the wave's share of the queue, [pool_at, pool_end)
  bool queue_dry = false;                // (wave-uniform) ... and the queue has nothing more to give
  // A frame with fewer pixels than the grid has lanes (C1: 65,536 on 262,144) is spread THIN: one pixel per
  // lane_stride lanes, so that every SIMD gets a wave and a wave's shared candidate tests serve 16 rays with 64 lanes
  // instead of 64 rays on a quarter of the SIMDs.  The idle lanes never fetch; they work in closest_hit.
  if (!(F & F_BVH) && lc.lane_stride > 1 && (threadIdx.x & (uint32_t)(lc.lane_stride - 1)) != 0u) done = true;
  if (!(F & F_BVH) && lc.chain_next != nullptr) {  // planned chains: this wave's chain and its first tile
    const int chain = wave_chain_id(lc.prio_tab, counters + 35, lc.plan_simds, lc.plan_rounds);
    const int32_t t = chain >= 0 ? lc.chain_first[chain] : -1;
    heavy = t < 0;
    if (t >= 0) {
      q32 = t * 64 + (int32_t)(threadIdx.x & 63u);
      uint32_t was = 1u;
      if ((threadIdx.x & 63u) == 0u) was = atomicCAS(&lc.claims[t], 0u, ((blockIdx.x * n_threads + threadIdx.x) >> 6) + 1u);
      if (__builtin_amdgcn_readfirstlane((int)was) == 0) (void)take_item((int64_t)q32);
    }
  }
  for (;;) {
    RTMI_STAT(const unsigned long long tq0 = stat_now();)
    if (lc.prio_tab != nullptr && (prio_tick++ & (uint32_t)(lc.prio_every - 1)) == 0u) {
      // queries this lane's pixel still has to do, from its own rays per sample so far (+ 8 rays over one more sample:
      // a pixel that has not started counts as an average one)
      // -- with the scheduler's probe behind the launch, its tile's rays per sample stand in as 64 samples' worth of
      // prior -- and, on a planned chain, what the probe said of the tiles still to come
      float left = 0.f;
      if (has_px && (active || k < fr.k_end)) {
        float prior_rays = 8.f, prior_n = 1.f;
        if (lc.tile_cost != nullptr) prior_rays = 64.f * ((float)lc.tile_cost[q32 >> 6] * lc.rate_scale), prior_n = 64.f;
        left = ((float)rays + prior_rays) * (float)(fr.k_end - k + 1) * __builtin_amdgcn_rcpf((float)k + prior_n);
        if (!(F & F_BVH) && lc.chain_next != nullptr && !heavy) left += (float)lc.chain_fut[q32 >> 6];
      }
      wave_priority_update(lc.prio_tab, (uint32_t)fminf(left, 4.0e9f));
    }
    // -------------------------------------------------------- sample / pixel bookkeeping
    if (!active && !done && has_px && k >= fr.k_end) {
      const int64_t q = (int64_t)q32;
      V3 c = color;
      if (fr.post && fr.k_end >= fr.spp) {  // ray_tracing.cu:78-83 (a first pass leaves the raw sum)
        c = c / (float)fr.spp;
        c = mk(clamp1(c.x, 0.f, 1.f), clamp1(c.y, 0.f, 1.f), clamp1(c.z, 0.f, 1.f));
        c = mk(sqrtf(c.x), sqrtf(c.y), sqrtf(c.z));
      }
      out[q * 3 + 0] = c.x;
      out[q * 3 + 1] = c.y;
      out[q * 3 + 2] = c.z;
      if (ray_counts) ray_counts[q] = rays;
      if ((F & F_BVH) && lc.visit_counts != nullptr) lc.visit_counts[q] = work_px;
      // the lane's ray total in ONE register: 2^31 at a time goes to the global counter (a constant addend: the
      // compiler's wave-level combining of atomics needs no scan for it), the rest at the end of the kernel
      ray_acc += rays;
      if (ray_acc >= 0x80000000u) {
        atomicAdd(&counters[1], 0x80000000ull);
        ray_acc -= 0x80000000u;
      }
      states[0 * n_items + q] = rng.d;
      states[1 * n_items + q] = rng.v0;
      states[2 * n_items + q] = rng.v1;
      states[3 * n_items + q] = rng.v2;
      states[4 * n_items + q] = rng.v3;
      states[5 * n_items + q] = rng.v4;
      has_px = false;
    }
    const bool wave_heavy = (F & F_BVH) && lc.exclusive &&
                            __builtin_amdgcn_ballot_w64(has_px && heavy && (active || k < fr.k_end)) != 0ull;
    if ((F & F_BVH) && classes) {
      const bool wants = !active && !done && !has_px;
      if (__builtin_amdgcn_ballot_w64(wants) != 0ull) {
        if (head_open)
          head_open = __hip_atomic_load(&counters[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < sparse_limit;
        const uint32_t lane = threadIdx.x & 63u;
        // four rounds, the lanes that can hold the heavier classes first; the wave's class is looked at again
        // before each round
        for (int round = 0; round < 4; round++) {
          const int lvl = round == 0 ? 64 : round == 1 ? 32 : round == 2 ? 16 : 1;
          const bool live_px = has_px && (active || k < fr.k_end);
          int wave_cls = 1;
          if (lc.exclusive) {
            if (__builtin_amdgcn_ballot_w64(live_px && cls == 16) != 0ull) wave_cls = 16;
            if (__builtin_amdgcn_ballot_w64(live_px && cls == 32) != 0ull) wave_cls = 32;
            if (__builtin_amdgcn_ballot_w64(live_px && cls == 64) != 0ull) wave_cls = 64;
          }
          if (wave_cls > lvl) break;  // no lane of this or a later round is a multiple of the wave's class
          const bool my_round = round == 0 ? lane == 0u : round == 1 ? lane == 32u : round == 2 ? (lane & 31u) == 16u
                                                                                                 : (lane & 15u) != 0u;
          if (!(wants && my_round)) continue;
          while (!has_px && !done) {
            // the head first (its front is the heaviest class left; a lane of level lvl may take classes <= lvl)
            if (head_open) {
              const unsigned long long front = __hip_atomic_load(&counters[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              const int front_cls = front < end64 ? 64 : front < end32 ? 32 : 16;
              if (front < sparse_limit && front_cls <= lvl) {
                const unsigned long long hq = atomicAdd(&counters[3], 1ull);
                if (hq < sparse_limit) {
                  if (take_item((int64_t)lc.head_list[hq])) cls = hq < end64 ? 64 : hq < end32 ? 32 : 16;
                  continue;
                }
              } else if (front < sparse_limit) {
                break;  // the head's front is for better-aligned lanes: look again next iteration
              }
            }
            const unsigned long long mq = atomicAdd(&counters[0], 1ull);
            if ((int64_t)mq >= n_items) {
              // nothing ordinary left: finished once the head is empty too
              if (__hip_atomic_load(&counters[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= sparse_limit) done = true;
              break;
            }
            const int64_t item = (int64_t)lc.tile_order[mq >> 4] * 16 + (int64_t)(mq & 15);
            if (lc.probe_marks[item] >> 31) continue;  // a head item: not from this queue
            if (take_item(item)) cls = 1;
          }
        }
      }
    } else if (!(F & F_BVH) && lc.chain_next != nullptr) {  // (wave-uniform) planned chains
      const uint32_t me = ((blockIdx.x * n_threads + threadIdx.x) >> 6) + 1u;  // this wave's mark in `claims`
      if (!active && !has_px && !heavy) {  // (`heavy` in this mode: the lane has walked its chain to the end)
        for (;;) {
          const int32_t t = lc.chain_next[q32 >> 6];
          if (t < 0) {
            heavy = true;
            break;
          }
          q32 = t * 64 + (int32_t)(threadIdx.x & 63u);  // (the walk goes on from here also when another wave has the tile)
          const uint32_t was = atomicCAS(&lc.claims[t], 0u, me);
          if ((was == 0u || was == me) && take_item((int64_t)q32)) break;
        }
      }
      // every lane idle, every chain walked: the wave takes over tiles nobody has started, lightest first
      if (!done && __builtin_amdgcn_ballot_w64(active || has_px || !heavy) == 0ull) {
        for (;;) {
          unsigned long long c = 0ull;
          if ((threadIdx.x & 63u) == 0u) c = atomicAdd(&counters[36], 1ull);
          const int ci = __builtin_amdgcn_readfirstlane((int)(c < (unsigned long long)fr.local_tiles ? c : (unsigned long long)fr.local_tiles));
          if (ci >= fr.local_tiles) {
            done = true;
            break;
          }
          const int32_t t = (int32_t)lc.tile_order[fr.local_tiles - 1 - ci];
          uint32_t was = 1u;
          if ((threadIdx.x & 63u) == 0u) was = atomicCAS(&lc.claims[t], 0u, me);
          if (__builtin_amdgcn_readfirstlane((int)was) != 0) continue;
          (void)take_item((int64_t)t * 64 + (int64_t)(threadIdx.x & 63u));
          if (__builtin_amdgcn_ballot_w64(has_px) != 0ull) break;  // (a tile of padding only: look further)
        }
      }
    } else if (!(F & F_BVH)) {
      // The queue of a list frame, drawn by the WAVE: one atomic takes the next `batch` items for all its lanes, which
      // help themselves from that pool as they finish their pixels.  Every atomic on the queue's cursor is a round
      // trip to the one L2 channel that owns its line, and they are served there one after the other (5.5 ns each,
      // measured): with one per pixel a first pass of two samples over a million pixels was 3.8 ms of atomics around
      // 0.55 ms of rendering.  `batch` shrinks with what is left of the queue (at most half a wave's fair share of it),
      // so that no wave sits on items while others have run dry; lc.fetch_batch = 1 is one atomic per fetch for the
      // lanes that wait at that moment -- the queue of rounds 1-3, kept for image-order frames of 64 samples and more,
      // whose last tiles weigh as much as any (kernels.hip: launch_render_t).
      for (;;) {
        const bool wants = !active && !done && !has_px;
        const unsigned long long wm = __builtin_amdgcn_ballot_w64(wants);
        if (wm == 0ull) break;
        if (pool_at == pool_end) {
          unsigned long long nq = 0ull;
          if (!queue_dry) {
            const uint32_t left = (uint32_t)n_items - pool_end;  // (what the cursor had left after this wave's last draw)
            uint32_t batch = left / (2u * (gridDim.x * (n_threads >> 6)));
            batch = batch < 1u ? 1u : batch > (uint32_t)lc.fetch_batch ? (uint32_t)lc.fetch_batch : batch;
            const uint32_t asked = (uint32_t)__popcll(wm);  // (never less than the lanes that are waiting right now)
            batch = batch < asked ? asked : batch;
            if ((threadIdx.x & 63u) == 0u) nq = atomicAdd(&counters[0], (unsigned long long)batch);
            nq = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(nq >> 32)) << 32) |
                 (uint32_t)__builtin_amdgcn_readfirstlane((int)nq);
            if (nq < (unsigned long long)n_items) {
              pool_at = (uint32_t)nq;
              pool_end = nq + batch < (unsigned long long)n_items ? (uint32_t)nq + batch : (uint32_t)n_items;
            } else {
              queue_dry = true;
            }
          }
          if (queue_dry) {
            done = done || wants;
            break;
          }
        }
        const uint32_t avail = pool_end - pool_at, want_n = (uint32_t)__popcll(wm);
        const uint32_t rank = (uint32_t)lane_rank(wm);
        if (wants && rank < avail) {
          const uint32_t nq = pool_at + rank;
          int64_t item = (int64_t)nq;
          if (lc.tile_order) item = (int64_t)lc.tile_order[nq >> 4] * 16 + (int64_t)(nq & 15u);
          (void)take_item(item);  // (padding: the lane still wants, and looks again)
        }
        pool_at += want_n < avail ? want_n : avail;
      }
    } else if (!active && !done) {
      while (!has_px && !done) {
        if ((F & F_BVH) && sparse_limit != 0ull && (threadIdx.x & (uint32_t)(lc.sparse_stride - 1)) != 0) {
          if (wave_heavy) break;  // this wave is busy with an outlier pixel: stay a helper
          // the head of the queue holds the outlier tiles: only every sparse_stride-th lane takes
          // pixels there (the others look again next round), so that a wave carries few rays
          // and the mesh search runs in its cooperative mode
          if (__hip_atomic_load(&counters[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < sparse_limit) break;
        }
        const unsigned long long nq = atomicAdd(&counters[0], 1ull);
        if ((int64_t)nq >= n_items) {
          done = true;
          break;
        }
        int64_t item = (int64_t)nq;
        if (lc.tile_order) item = (int64_t)lc.tile_order[nq >> 4] * 16 + (int64_t)(nq & 15);
        if (!take_item(item)) continue;
        heavy = nq < sparse_limit;
      }
    }
    if (!active && !done) {
      if (has_px) {
        // ray_tracing.cu:68-74 + camera.cu:57-70
        float r1 = rng_01(rng);
        float r2 = rng_01(rng);
        // ray_tracing.cu:68-73 in binary64: x = (u + j) / W, then x = 2x - 1 and, inside RayAt, x = (x + 1) / 2.
        float xf, yf;
        if (f32_jitter) {
          // Power-of-two frame: every binary64 step is EXACT -- u + j needs at most 24 + 13 bits, the division
          // is a scaling, 2x - 1 and (x + 1) / 2 undo each other without rounding -- so the only rounding is the
          // final (float)x of the exact (u + j) / W; and rounding commutes with a power-of-two scaling, so that
          // is RN(u + j) * (1 / W): one binary32 addition (correctly rounded sum of two binary32 numbers) and an
          // exact multiplication.  (tests/test_host_logic.py::test_jitter_in_binary32_for_power_of_two_frames)
          xf = (r1 + (float)(pij & 0xffffu)) * (float)inv_w;
          yf = (r2 + (float)(fr.height - (int)(pij >> 16))) * (float)inv_h;
        } else {
          double x = (double)r1 + (double)(int)(pij & 0xffffu);
          double y = (double)r2 + (double)(fr.height - (int)(pij >> 16));
          // division by a power of two is an exact scaling: multiply by the (exact) reciprocal
          x = w_pow2 ? x * inv_w : x / (double)fr.width;
          y = h_pow2 ? y * inv_h : y / (double)fr.height;
          x = 2 * x - 1;
          y = 2 * y - 1;
          x = (x + 1) / 2;
          y = (y + 1) / 2;
          xf = (float)x, yf = (float)y;
        }
        V3 target = sc.cam.llc + xf * sc.cam.horizontal + yf * sc.cam.vertical;
        V3 origin = sc.cam.position;
        if (F & F_DEFOCUS) {
          if (sc.cam.defocus) {  // camera.cu:63-65,74-77 (a square, drawn left to right)
            float ox = rng_range(0.f, sc.cam.lens_radius, rng);
            float oy = rng_range(0.f, sc.cam.lens_radius, rng);
            origin = sc.cam.position + sc.cam.u * ox + sc.cam.v * oy;
          }
        }
        o = origin;
        d = unit3_rn_twice(target - origin);  // RayAt normalises, Ray's constructor normalises again
        k++;
        if ((F & F_BVH) && promote && cls == 1 && k >= lc.promote) {
          const unsigned long long have = (unsigned long long)rays * (unsigned)lc.probe_spp, per = (unsigned long long)k;
          if (have >= thr16 * per) cls = have >= thr64 * per ? 64 : have >= thr32 * per ? 32 : 16;
        }
        depth = 0;
        active = true;
      }
    }
    if (!__any(active)) {
      if (!(F & F_BVH) || __all(done)) break;
      continue;  // lanes held back from the sparse head of the queue: it has just moved on
    }

#if defined(RTMI_STATS) && RTMI_STATS == 9
    // (lite build: when this wave has done 600, 1200, ... 5400 queries -> columns 1..9 of its record)
    if ((wave_queries + 1u) % 600u == 0u && (wave_queries + 1u) / 600u <= 9u && (threadIdx.x & 63u) == 0u)
      g_wave_stats[((blockIdx.x * blockDim.x + threadIdx.x) >> 6) & 16383u][(wave_queries + 1u) / 600u] =
          ((stat_real() - t_begin) & 0xffffffffffull) | ((unsigned long long)(st.node_steps + st.face_steps) << 40);
#endif
    RTMI_STAT(wave_queries++; const unsigned long long tq1 = stat_now(); st.cyc[0] += tq1 - tq0;
              const unsigned long long in0 = st.cyc[2] + st.cyc[3];)
    Hit h = {};
    const bool all_lanes_in = (F & F_BVH) || ((F & F_TRIS) && ll != nullptr && sc.n_pairs >= kCullMinPairs) ||
                              ((F & F_SGROUP) && cands != nullptr);  // wave-uniform
    if (all_lanes_in)  // every lane goes in, with or without a ray of its own: see closest_hit
      h = closest_hit<F>(sc, s_nodes, lc.lds_nodes, s_paths, lc.lds_paths, s_pairs, ll, cands, wl, counters + 2, o, d, active,
                         (F & F_BVH) && lc.visit_counts != nullptr
#ifdef RTMI_STATS
                         , st
#endif
      );
    RTMI_STAT(const unsigned long long tq2 = stat_now(); st.cyc[1] += (tq2 - tq1) - (st.cyc[2] + st.cyc[3] - in0);)
#ifdef RTMI_CHECK_MARGINS
    // Diagnostic build (tools/check_margins.sh): every 256th query of a wave is answered a second time WITHOUT the
    // per-lane culls -- the plain wave-uniform walk of the world list, the plain sphere loop -- and compared: a
    // padded bound or distance slack that let an acceptable primitive slip (closest_hit.h, scene.hip) shows as a
    // disagreement.  counters[33] += rays re-done, counters[34] += disagreements (rtmi_debug_counters).
    // Mesh variants: the second answer walks the reference's own tree (closest_hit.h: bvh_reference_walk), thousands
    // of triangle tests per ray -- for small frames (tools/gpu_check_margins.py).
    if (all_lanes_in && (check_tick++ % RTMI_CHECK_EVERY) == 0u) {
      const Hit h2 = closest_hit<F>(sc, s_nodes, 0, s_paths, 0, nullptr, nullptr, nullptr, nullptr, nullptr, o, d, active, false
#ifdef RTMI_STATS
                                    , st
#endif
      );
      const bool differs = active && (h2.ok != h.ok || (h.ok && (__float_as_uint(h2.t) != __float_as_uint(h.t) || h2.win != h.win ||
                                                                 ((F & F_BVH) && h2.aux != h.aux))));
      const unsigned long long na = __builtin_amdgcn_ballot_w64(active), nd = __builtin_amdgcn_ballot_w64(differs);
      if ((threadIdx.x & 63u) == 0u) {
        atomicAdd(&counters[33], (unsigned long long)__popcll(na));
        if (nd) atomicAdd(&counters[34], (unsigned long long)__popcll(nd));
      }
    }
#endif
    if (active) {
      if (!all_lanes_in)
        h = closest_hit<F>(sc, s_nodes, 0, s_paths, 0, s_pairs, nullptr, nullptr, nullptr, nullptr, o, d, true, false
#ifdef RTMI_STATS
                           , st
#endif
        );
      rays++;
      if ((F & F_BVH) && lc.visit_counts != nullptr) work_px += (uint32_t)h.work;

      V3 result = splat(0.f);
      bool ended = true;
      if (h.ok && depth < fr.max_depth) {  // ray_tracing.cu:23
        const uint32_t kind = h.win >> 29;
        const uint32_t index = h.win & ID_INDEX_MASK;
        V3 p = o + h.t * d;  // ray_tracing.cu:32 and the materials' own `p`
        if (kind == RUN_SKY) {
          // sky.cu:9-14: Scatter false; Emit(p) = gradient on normalize(p)
          V3 dir = unit3_rn(p);
          float tg = (float)(0.5 * ((double)dir.y + 1.0));
          float w0 = 1.0f - tg;
          result = mk(w0 * 1.0f + tg * 0.5f, w0 * 1.0f + tg * 0.7f, w0 * 1.0f + tg * 1.0f);
        } else {
          V3 nrm = splat(0.f);
          int mat = 0;
          float tu = 0.f, tv = 0.f;  // record.u, record.v (only read by image textures)
          if ((F & F_TRIS) && kind == RUN_TRIS) {
            V3 n;
            int flags;
            if (s_pairs != nullptr) {  // (wave-uniform) the winner's record is in LDS
              const float4 tn = s_nrm[index];
              n = mk(tn.x, tn.y, tn.z);
              mat = __float_as_int(tn.w) & 0xffffff, flags = __float_as_int(tn.w) >> 24;
            } else {
              const HotTri &tr = sc.tris[index];  // per-lane gather of the winner (L1/L2 resident)
              n = mk(tr.n[0], tr.n[1], tr.n[2]);
              mat = tr.mat, flags = tr.flags;
            }
            nrm = dot3(d, n) < 0.f ? n : -n;  // utils.cu:80
            if (F & F_TEX) {
              if (flags & TRI_PGRAM) {  // parallelogram.cu:26-29,35-38
                float w = (float)((1.0 - (double)h.u) - (double)h.v);
                if (!(flags & TRI_SECOND)) {
                  tu = (0.f * w + 1.f * h.u) + 0.f * h.v;
                  tv = (1.f * w + 1.f * h.u) + 0.f * h.v;
                } else {
                  tu = (1.f * w + 0.f * h.u) + 1.f * h.v;
                  tv = (1.f * w + 0.f * h.u) + 0.f * h.v;
                }
              } else {
                tu = h.u, tv = h.v;  // triangle.cu:13
              }
            }
          }
          if ((F & F_SPHERE) && kind == RUN_SPHERE) {
            const SphereRec &sr = sc.spheres[index];
            nrm = unit3_rn(p - mk(sr.cx, sr.cy, sr.cz));  // sphere.cu:25-26
            mat = sr.mat;
            if (F & F_TEX) {  // sphere.cu:60-63
              const float pi_f = 3.14159265358979323846264338327950288f;
              float theta = acosf(-nrm.y);
              float phi = atan2f(-nrm.z, nrm.x) + pi_f;
              tu = phi / (2 * pi_f);
              tv = theta / pi_f;
            }
          }
          if ((F & F_BVH) && kind == RUN_BVH) {
            const FaceRec &fc = sc.faces[index];
            // utils.cu:79: normalize(cross(v0v1, v0v2)), recomputed for the winning face only
            V3 n = unit3_rn(cross3(mk(fc.e1[0], fc.e1[1], fc.e1[2]), mk(fc.e2[0], fc.e2[1], fc.e2[2])));
            nrm = dot3(d, n) < 0.f ? n : -n;
            const BvhRec br = sc.bvhs[h.aux];
            mat = br.mat;
            if ((F & F_TEX) && br.has_uv) {  // bvh.cuh:41-45
              const float *tc = sc.face_uv + (size_t)(br.face_base + fc.orig) * 6;
              float w = (float)((1.0 - (double)h.u) - (double)h.v);
              tu = (tc[0] * w + tc[2] * h.u) + tc[4] * h.v;
              tv = (tc[1] * w + tc[3] * h.u) + tc[5] * h.v;
            }
          }
          MatRec m;  // r, g, b, kind | param, tex (the padding is not read)
          {
            float4 m0;
            float2 m1;
            if (mats_in_lds) {  // (wave-uniform)
              m0 = load_lds<float4>(s_mats + mat), m1 = load_lds<float2>(reinterpret_cast<const char *>(s_mats + mat) + 16);
            } else {
              m0 = load_global<float4>(sc.mats + mat), m1 = load_global<float2>(reinterpret_cast<const char *>(sc.mats + mat) + 16);
            }
            m.r = m0.x, m.g = m0.y, m.b = m0.z, m.kind = __float_as_int(m0.w), m.param = m1.x, m.tex = __float_as_int(m1.y);
          }
          V3 rgb = mk(m.r, m.g, m.b);
          RTMI_STAT2(if (!(F & F_BVH)) { const unsigned long long tsa = stat_now();  // (divergent code: first active lane reports)
            if ((int)(threadIdx.x & 63u) == __builtin_ctzll(__ballot(1))) g_wave_stats[((blockIdx.x * blockDim.x + threadIdx.x) >> 6) & 16383u][9] += tsa - tq2; })
          uint32_t layer = (uint32_t)mat;  // (F_TEX) what the fold will need of this bounce
          if (F & F_TEX) {
            if (m.tex >= 0 && (m.kind == MAT_LAMBERTIAN || m.kind == MAT_LIGHT)) {
              // image_texture.cu:11-13: v = 1.0 - v in double, then float coordinates
              const uint32_t px = tex_fetch(sc.texs[m.tex], tu, (float)(1.0 - (double)tv));
              layer = 0x80000000u | px;
              if (m.kind == MAT_LIGHT) rgb = texel_rgb(px);
            }
          }
          if (m.kind == MAT_LIGHT) {
            result = rgb;  // diffuse_light.cu:5-13
          } else {
            const float dn = dot3(d, nrm);
            V3 nd = splat(0.f);
            bool scattered = false;
            if (m.kind == MAT_LAMBERTIAN) {  // lambertian.cu:33-43
              if (!(dn >= 0.f)) {
                float sum;
                V3 s = ball_sample(rng, sum);
                s = sampler_on_sphere(s, sum);  // l = (float)pow((double)sum, 0.5); vec /= l (lambertian.cu:25-29)
                nd = unit3_rn_twice(s + nrm);   // normalize(S + n), then Ray's constructor (lambertian.cu:41-42)
                scattered = true;
              }
            } else if (m.kind == MAT_METAL) {  // metal.cu:12-25
              if (!(dn >= 0.f)) {
                V3 refl = reflect3(d, nrm);
                if (m.param > 0.f) {
                  float sum;
                  V3 s = ball_sample(rng, sum);
                  nd = refl + m.param * s;
                } else {
                  nd = refl;
                }
                nd = unit3_rn(nd);  // Ray's constructor
                scattered = true;
              }
            } else {  // MAT_DIELECTRIC, dielectric.cu:16-44
              if (dn >= 0.f)
                nd = refract3(d, -nrm, m.param / 1.0f);
              else
                nd = refract3(d, nrm, 1.0f / m.param);
              bool zero = (nd.x == 0.f && nd.y == 0.f && nd.z == 0.f);
              bool nan = (nd.x != nd.x) || (nd.y != nd.y) || (nd.z != nd.z);
              scattered = !(zero || nan);
              if (scattered) nd = unit3_rn(nd);  // Ray's constructor
            }
            if (scattered) {
              if (F & F_TEX) {
                *reinterpret_cast<uint32_t *>(smem + tex_layer_offset(depth)) = layer;
              } else if (nibble_ids) {
                const uint32_t at = ids_offset(depth >> 1);  // this lane's own byte: no other lane writes it
                const uint32_t old = smem[at];
                smem[at] = (uint8_t)((depth & 1) ? ((old & 0x0fu) | ((uint32_t)mat << 4)) : ((old & 0xf0u) | (uint32_t)mat));
              } else if (lc.wide_ids) {
                *reinterpret_cast<uint16_t *>(smem + ids_offset(depth)) = (uint16_t)mat;
              } else {
                smem[ids_offset(depth)] = (uint8_t)mat;
              }
              depth++;
              o = p;
              d = nd;  // (normalised by Ray's constructor in its material's branch above)
              ended = false;
            }
          }
        }
      }
      RTMI_STAT2(if (!(F & F_BVH)) { const unsigned long long tsb = stat_now();
        if ((int)(threadIdx.x & 63u) == __builtin_ctzll(__ballot(1))) g_wave_stats[((blockIdx.x * blockDim.x + threadIdx.x) >> 6) & 16383u][10] += tsb - tq2; })
      if (ended) {
        // ray_tracing.cu:50-52 with emitted == 0 on every stored layer: result = emitted +
        // attenuation * result, deepest layer first.  The addition only matters for a product of -0,
        // which needs a colour with its sign bit set (sc.unsigned_colours).
        int i = depth - 1;
        if (!(F & F_TEX) && fast_fold) {
          // common case (byte ids, material table in LDS, no signed colours) without the per-layer
          // uniform branches: four layers at a time, ids first, then colours, then the products.
          // The id bytes are read at a RUNNING byte offset that steps down a row at a time, as an LDS address proper
          // (lds_byte): `smem[ids_offset(level)]` cost a 32-bit multiply per group of levels and an addition of the
          // array's link-time base -- zero -- per byte.  The wave folds as long as its deepest finished path is
          // (17 levels on average in a Cornell box: the deepest of the ten paths that end per iteration).
          const uint32_t step = n_threads;
          if (nibble_ids) {
            uint32_t at = ids_offset(i >= 0 ? i >> 1 : 0);
            if (i >= 0 && !(i & 1)) {  // an even top level sits alone in the low half of its byte
              const int m0 = lds_byte(at) & 15;
              { const V3 a_ = lds_rgb(m0); result = mk(a_.x * result.x, a_.y * result.y, a_.z * result.z); }
              i--, at -= step;
            }
            for (; i >= 3; i -= 4, at -= 2u * step) {  // i odd: bytes (i >> 1) and (i >> 1) - 1 hold levels i, i - 1 and i - 2, i - 3
              const uint32_t b0 = lds_byte(at), b1 = lds_byte(at - step);
              const int m0 = b0 >> 4, m1 = b0 & 15, m2 = b1 >> 4, m3 = b1 & 15;
              const V3 a0 = lds_rgb(m0), a1 = lds_rgb(m1);
              const V3 a2 = lds_rgb(m2), a3 = lds_rgb(m3);
              result = mk(a0.x * result.x, a0.y * result.y, a0.z * result.z);
              result = mk(a1.x * result.x, a1.y * result.y, a1.z * result.z);
              result = mk(a2.x * result.x, a2.y * result.y, a2.z * result.z);
              result = mk(a3.x * result.x, a3.y * result.y, a3.z * result.z);
            }
            for (; i >= 1; i -= 2, at -= step) {
              const uint32_t b0 = lds_byte(at);
              const int m0 = b0 >> 4, m1 = b0 & 15;
              { const V3 a_ = lds_rgb(m0); result = mk(a_.x * result.x, a_.y * result.y, a_.z * result.z); }
              { const V3 a_ = lds_rgb(m1); result = mk(a_.x * result.x, a_.y * result.y, a_.z * result.z); }
            }
          } else {
            uint32_t at = ids_offset(i >= 0 ? i : 0);
            for (; i >= 3; i -= 4, at -= 4u * step) {
              const int m0 = lds_byte(at), m1 = lds_byte(at - step), m2 = lds_byte(at - 2u * step), m3 = lds_byte(at - 3u * step);
              const V3 a0 = lds_rgb(m0), a1 = lds_rgb(m1);
              const V3 a2 = lds_rgb(m2), a3 = lds_rgb(m3);
              result = mk(a0.x * result.x, a0.y * result.y, a0.z * result.z);
              result = mk(a1.x * result.x, a1.y * result.y, a1.z * result.z);
              result = mk(a2.x * result.x, a2.y * result.y, a2.z * result.z);
              result = mk(a3.x * result.x, a3.y * result.y, a3.z * result.z);
            }
            for (; i >= 0; i--, at -= step) {
              const int m0 = lds_byte(at);
              { const V3 a_ = lds_rgb(m0); result = mk(a_.x * result.x, a_.y * result.y, a_.z * result.z); }
            }
          }
        }
        for (; i >= 0; i--) {
          V3 a;
          if (F & F_TEX) {
            const uint32_t lw = *reinterpret_cast<const uint32_t *>(smem + tex_layer_offset(i));
            if (lw >> 31) {
              a = texel_rgb(lw);  // the same byte / 255 the sample would have returned
            } else if (mats_in_lds) {
              a = lds_rgb((int)lw);
            } else {
              a = mk(sc.mats[lw].r, sc.mats[lw].g, sc.mats[lw].b);
            }
          } else {
            const int mi = nibble_ids     ? (int)((smem[ids_offset(i >> 1)] >> ((i & 1) * 4)) & 15u)
                           : lc.wide_ids ? (int)*reinterpret_cast<const uint16_t *>(smem + ids_offset(i))
                                         : (int)smem[ids_offset(i)];
            if (mats_in_lds) {
              a = lds_rgb(mi);
            } else {
              a = mk(sc.mats[mi].r, sc.mats[mi].g, sc.mats[mi].b);
            }
          }
          if (sc.unsigned_colours) {
            result = mk(a.x * result.x, a.y * result.y, a.z * result.z);
          } else {
            result = mk(0.f + a.x * result.x, 0.f + a.y * result.y, 0.f + a.z * result.z);
          }
        }
        color = color + result;
        active = false;
      }
    }
    RTMI_STAT(st.cyc[4] += stat_now() - tq2;)
  }

  if (lc.prio_tab != nullptr) wave_priority_leave(lc.prio_tab);
  {  // total closest-hit queries: wave reduce, one atomic per wave
    unsigned long long ray_total = ray_acc;
    for (int off = 32; off > 0; off >>= 1) ray_total += __shfl_down(ray_total, off);
    if ((threadIdx.x & 63) == 0 && ray_total) atomicAdd(&counters[1], ray_total);
  }
#ifdef RTMI_STATS
  if ((threadIdx.x & 63) == 0) {
    const unsigned v[13] = {wave_queries, st.searches, st.node_steps, st.face_steps, st.nodes_popped, st.blocks_popped,
                            st.insert_rounds, st.steps_hist[0], st.steps_hist[1], st.steps_hist[2], st.steps_hist[3],
                            st.steps_hist[4], st.steps_hist[5]};
    for (int i = 0; i < 13; i++) atomicAdd(&counters[4 + i], (unsigned long long)v[i]);
    for (int i = 0; i < 9; i++) atomicAdd(&counters[17 + i], st.cyc[i]);
    const unsigned long long life = stat_real() - t_begin;
    atomicAdd(&counters[26], life);
    atomicMax(&counters[27], life);
    atomicAdd(&counters[28], 1ull);
    atomicAdd(&counters[29], st.calib);
    atomicAdd(&counters[30], st.cull_bits);
    atomicAdd(&counters[31], st.cull_iters);
    atomicAdd(&counters[32], st.cull_rays);
    const unsigned wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (!(F & F_BVH) && wid < 16384u) {
      atomicAdd(&counters[25], g_wave_stats[wid][9]);
      atomicAdd(&counters[6], g_wave_stats[wid][10]);
      g_wave_stats[wid][9] = 0, g_wave_stats[wid][10] = 0;
      // per wave (tools/gpu_shard_waves.py): lifetime in shader cycles, queries, start and end on the 100 MHz clock
      g_wave_stats[wid][0] = life, g_wave_stats[wid][11] = wave_queries;
      g_wave_stats[wid][12] = t_begin_rt, g_wave_stats[wid][13] = __builtin_amdgcn_s_memrealtime();
      g_wave_stats[wid][14] = (unsigned long long)__builtin_amdgcn_s_getreg(0xF804) | ((unsigned long long)__builtin_amdgcn_s_getreg(0xF814) << 32);
    } else if (wid < 16384u) {
      g_wave_stats[wid][0] = life;
#if RTMI_STATS != 9
      for (int i = 0; i < 9; i++) g_wave_stats[wid][1 + i] = st.cyc[i];
#endif
      g_wave_stats[wid][10] = wave_queries, g_wave_stats[wid][11] = st.node_steps, g_wave_stats[wid][12] = st.face_steps;
      g_wave_stats[wid][13] = st.nodes_popped, g_wave_stats[wid][14] = st.insert_rounds;
      g_wave_stats[wid][15] = (st.cyc[9] << 32) | (st.cyc[10] >> 8);  // setup cycles | loop-control cycles / 256
    }
  }
#endif
}
