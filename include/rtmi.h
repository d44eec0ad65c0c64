/*
 * rtmi.h — C ABI of the MI355X-native path-tracing hot path (librtmi.so).
 *
 * This is the drop-in boundary for the per-pixel trace loop of
 * tigert1998/ray-tracing-cuda.  Every entry point names the reference
 * interface it replaces (paths relative to /root/reference/ray-tracing-cuda/).
 * The reference's host driver (`Main` / `DistributedMain`, utils.cu:132-242)
 * performs, in order: allocate states+image -> CudaRandomInit kernel ->
 * user `init_world` callback -> PathTracing kernel -> D2H -> (MPI reduce) ->
 * JPEG.  The calls below are those steps with plain pointers and sizes.
 *
 * Conventions
 *   - All functions return 0 on success or a negative rtmi_status; the message
 *     is available from rtmi_last_error() (thread-local).  The reference aborts
 *     through glog CHECK (utils.cu:143-144); the C++ wrappers in
 *     ray-tracing-cuda_amd/api/utils.cuh turn a non-zero status into the same CHECK failure.
 *   - `d_*` pointers are device (HBM) pointers owned by the caller.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  All
 *     device work is enqueued on it; functions documented "synchronous" wait
 *     for it before returning.
 *   - There is NO CPU fallback: every compute entry point fails with
 *     RTMI_ERR_NO_DEVICE when no gfx950-class device is usable.
 *
 * Pixel ownership (multi-GPU): the frame is cut into 8x8-pixel tiles numbered
 * row-major; rank r of world_size G owns tiles t with t % G == r.  A rank's
 * pixels are addressed by *work item* q in [0, rtmi_frame_work_items()):
 * local tile q/64, pixel-in-tile q%64 (row-major 8x8).  RNG states and the
 * radiance buffer of a rank are indexed by q ("tile-major").  Every pixel keeps
 * cuRAND subsequence == its GLOBAL index i*width+j, so the image is
 * bit-identical for every world_size.  Work items that fall outside the image
 * (ragged right/bottom tiles) are inert padding.
 */
#ifndef RTMI_H_
#define RTMI_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTMI_VERSION 3
#define RTMI_TILE 8            /* tile edge in pixels; 64 work items per tile = one wavefront */
#define RTMI_STATE_WORDS 6     /* live words of curandState: d, v[0..4] */

typedef enum rtmi_status {
  RTMI_OK = 0,
  RTMI_ERR_INVALID = -1,    /* bad argument / scene not committed / handle out of range */
  RTMI_ERR_NO_DEVICE = -2,  /* HIP runtime or device unavailable */
  RTMI_ERR_HIP = -3,        /* a HIP call failed; see rtmi_last_error() */
  RTMI_ERR_CAPACITY = -4,   /* HitableList::kMaxHitables (1024) exceeded, hitable_list.cuh:10 */
  RTMI_ERR_DEPTH = -5,      /* max_depth outside [0, RTMI_MAX_DEPTH] */
  RTMI_ERR_INTERNAL = -6    /* an internal invariant of the kernels did not hold; the output is not to be used */
} rtmi_status;

#define RTMI_MAX_DEPTH 64      /* TRACE_DEPTH_LIMIT is 10 in ray_tracing.cu:10; BASELINE configs use 8/10/50 */
#define RTMI_MAX_HITABLES 1024 /* hitable_list.cuh:10 */

typedef struct rtmi_scene rtmi_scene; /* opaque; replaces the device-resident HitableList + Camera pair */

/* Frame + shard description; replaces the (height, width, spp, post_processing)
 * arguments of PathTracing (ray_tracing.cuh:19-21) and the rank/world_size of
 * DistributedMain (utils.cu:186-189). */
typedef struct rtmi_frame {
  int32_t height;       /* 1 .. RTMI_MAX_EXTENT (the reference has no such limit: a pixel's row and column share one */
  int32_t width;        /* 32-bit register of the trace kernel); a larger frame is refused with RTMI_ERR_INVALID */
  int32_t spp;          /* samples per pixel rendered by THIS call; spp * (max_depth + 1) <= RTMI_MAX_PIXEL_QUERIES (a */
                        /* pixel's closest-hit queries are counted in 31 bits), else rtmi_render refuses the frame */
  int32_t max_depth;    /* TRACE_DEPTH_LIMIT, ray_tracing.cu:10,23 */
  int32_t post_process; /* 1: out = sqrt(clamp(sum/spp,0,1)) (ray_tracing.cu:78-83); 0: raw sum */
  int32_t rank;         /* tile shard owner, 0 <= rank < world_size */
  int32_t world_size;   /* number of shards (GPUs) */
} rtmi_frame;

#define RTMI_MAX_EXTENT 65535
#define RTMI_MAX_PIXEL_QUERIES 2147483647
const char *rtmi_last_error(void);
int rtmi_version(void);
/* Number of usable GPUs (0 when there is none); never fails. */
int rtmi_device_count(void);

/* ------------------------------------------------------------------ scene --
 * Host-side recording of the scene graph, one call per reference constructor.
 * Texture / material calls return a handle >= 0 (or a negative rtmi_status).
 * Hitables are appended to the world in call order == HitableList::Append
 * order (hitable_list.cu:27-29); list order decides ties (hitable_list.cu:18). */
rtmi_scene *rtmi_scene_create(void);
void rtmi_scene_destroy(rtmi_scene *s);

int rtmi_constant_texture(rtmi_scene *s, const float rgb[3]);                 /* textures/constant_texture.cu:7-9 */
int rtmi_image_texture(rtmi_scene *s, const uint8_t *rgba, int height, int width,
                       size_t pitch_bytes);                                   /* textures/image_texture.cu:17-38 (host RGBA8, point/wrap) */
int rtmi_lambertian(rtmi_scene *s, const float rgb[3]);                        /* lambertian.cu:14-17 */
int rtmi_lambertian_tex(rtmi_scene *s, int texture);                           /* lambertian.cu:9-12 */
int rtmi_metal(rtmi_scene *s, const float rgb[3], float fuzz);                 /* metal.cu:7-10 */
int rtmi_dielectric(rtmi_scene *s, const float rgb[3], double refractive_index); /* dielectric.cu:10-14 */
int rtmi_diffuse_light(rtmi_scene *s, int texture);                            /* diffuse_light.cu:15-17 */

int rtmi_add_sphere(rtmi_scene *s, const float center[3], double radius, int material);       /* sphere.cu:7-9 */
int rtmi_add_triangle(rtmi_scene *s, const float p[9], int material);                           /* triangle.cu:6-9 */
int rtmi_add_parallelogram(rtmi_scene *s, const float p[9], int material);                      /* parallelogram.cu:10-15 */
int rtmi_add_parallelepiped(rtmi_scene *s, const float p[12], int material);                    /* parallelepiped.cu:8-18 */
typedef void (*rtmi_transform_fn)(const float in[3], float out[3], void *user);
int rtmi_add_parallelepiped_lengths(rtmi_scene *s, const float lengths[3], int material,
                                    rtmi_transform_fn transform, void *user);                   /* parallelepiped.cu:34-55 */
/* A Parallelepiped given as the six parallelograms AddCorner appended (3 points each, 54
 * floats); used when the corners were derived elsewhere (device-side constructors). */
int rtmi_add_parallelepiped_faces(rtmi_scene *s, const float faces[54], int material);          /* parallelepiped.cu:25-32 */
int rtmi_add_sky(rtmi_scene *s);                                                                /* sky.cu:16 */
/* A HitableList appended to the list under construction (HitableList is itself a Hitable,
 * hitable_list.cuh:8): `l = new HitableList(); l->Append(...); parent->Append(l)`.  The hitables added
 * between begin and end are its entries; lists nest to any depth.  It counts as one entry of its
 * parent and holds up to RTMI_MAX_HITABLES entries of its own.  The library inlines it at its position:
 * the closest hit -- ties included -- is the one the nested call returns (DESIGN.md "List flattening"). */
int rtmi_list_begin(rtmi_scene *s);
int rtmi_list_end(rtmi_scene *s);
/* BVH<Face<HasTexCoord>,AABB>(faces, n, material) (bvh.cuh:170-173).  faces:
 * n*9 floats; uvs: n*6 floats or NULL (Face<false>); material < 0 keeps
 * "material_ptr_ == nullptr" (bvh.cuh:178).  leaf_max is BVHNode::kMin (2048,
 * bvh.cuh:105); pass 0 for the reference value. */
int rtmi_add_bvh(rtmi_scene *s, const float *faces, const float *uvs, int n, int material, int leaf_max);

int rtmi_camera_pinhole(rtmi_scene *s, const float pos[3], const float look_at[3], const float up[3],
                        double fov, double aspect);                                              /* camera.cu:24-38 */
int rtmi_camera_defocus(rtmi_scene *s, const float pos[3], const float look_at[3], const float up[3],
                        double fov, double aspect, double aperture, double focus_distance);     /* camera.cu:6-22 */
int rtmi_camera_raw(rtmi_scene *s, const float pos[3], const float lower_left[3], const float horizontal[3],
                    const float vertical[3]);                                                    /* camera.cu:40-47 */
/* position, lower_left_corner, horizontal, vertical, u, v, w (21 floats) */
int rtmi_camera_get(const rtmi_scene *s, float out[21]);
/* Install a camera whose frame was computed elsewhere (a Camera constructed on the device):
 * the same 21 floats, is_defocus_camera_ and lens_radius_ (camera.cuh:12-14). */
int rtmi_camera_set(rtmi_scene *s, const float frame[21], int is_defocus, double lens_radius);

/* Flatten the recorded graph into the device layout and upload it to the
 * current HIP device.  Synchronous.  Replaces the point in Main where
 * init_world has run and cudaDeviceSynchronize returns (utils.cu:148-152). */
int rtmi_scene_commit(rtmi_scene *s);
/* Counts of the flattened scene: {entries of the world list (a nested list counts once), spheres,
 * parallelograms (incl. box faces), triangles, bvh faces, bvh nodes, materials, textures}. */
int rtmi_scene_stats(const rtmi_scene *s, int64_t out[8]);
/* Mesh faces whose smallest interior angle is below 1.8 degrees (sine below 1/32).  The reference's binary32 triangle
 * test (utils.cu:49-85) accepts rays that pass such a face at a distance of about eps x (distance to the ray's origin)
 * / sin(angle) -- further than the 2^-16 distance slack every search box gets.  Since round 3 the nodes of the search
 * tree above a thin face widen their children's boxes by what it asks for (8 eps / sin(angle), as a power of two), so
 * the count is informational: the search cost of rays that come near those nodes grows with it, exactness does not
 * depend on it (tests: test_far_views_and_thin_faces, test_needles_and_grazing_views_match_the_oracle;
 * tools/gpu_check_margins.py meshes re-answers every query by the reference's own tree walk). */
int64_t rtmi_scene_sliver_faces(const rtmi_scene *s);
/* Algorithmic bytes one closest-hit query consults; BVH scenes
 * need the measured per-ray node/face visits and report only the fixed part. */
int64_t rtmi_scene_bytes_per_ray(const rtmi_scene *s);

/* ------------------------------------------------------------------ frame -- */
/* Work items (pixels incl. ragged-tile padding) owned by frame->rank. */
int64_t rtmi_frame_work_items(const rtmi_frame *f);
/* Global pixel index (i*width+j) of work item q of this shard, or -1 for padding. */
int64_t rtmi_frame_pixel_of(const rtmi_frame *f, int64_t q);
/* Bulk form: out[q] for every work item q of this shard (out has rtmi_frame_work_items entries). */
int rtmi_frame_pixel_map(const rtmi_frame *f, int64_t *out);
/* Bytes the caller must allocate for d_states / d_tiles of this shard. */
size_t rtmi_states_bytes(const rtmi_frame *f);   /* 6 planes of uint32[work_items] (struct-of-arrays) */
size_t rtmi_tiles_bytes(const rtmi_frame *f);    /* float[work_items][3] */

/* -------------------------------------------------------------------- RNG --
 * Replaces CudaRandomInit<<<>>>(seed, states, n) (utils.cu:43-47,146,202):
 * state(q) = curand_init(seed, subsequence = global pixel index, offset 0).
 * Asynchronous on `stream`. */
int rtmi_rng_init(uint64_t seed, const rtmi_frame *f, void *d_states, void *stream);
/* Host copy of curand_init(seed, subsequence, 0): {d, v0..v4}. */
int rtmi_rng_host_state(uint64_t seed, uint64_t subsequence, uint32_t state[RTMI_STATE_WORDS]);
/* CudaRandomFloat(min, max, state) on a host state (utils.cuh:22-27); scene
 * programs that draw their layout from pixel 0's stream (scenes/spheres.cu:105)
 * use this and then store the advanced state with rtmi_rng_set_state. */
float rtmi_rng_host_random_float(float min, float max, uint32_t state[RTMI_STATE_WORDS]);
/* Overwrite / read back the state of work item q.  Synchronous. */
int rtmi_rng_set_state(const rtmi_frame *f, void *d_states, int64_t q, const uint32_t state[RTMI_STATE_WORDS],
                       void *stream);
int rtmi_rng_get_state(const rtmi_frame *f, const void *d_states, int64_t q, uint32_t state[RTMI_STATE_WORDS],
                       void *stream);

/* ----------------------------------------------------------------- render --
 * Replaces PathTracing<<<grid,block>>>(world, camera, H, W, spp, post, states,
 * out) (ray_tracing.cu:56-85; launches utils.cu:158-163, 216-221) for the
 * pixels of frame->rank.  d_tiles receives float[work_items][3] (tile-major);
 * d_ray_counts (nullable) receives the per-pixel number of closest-hit queries
 * issued by Trace (ray_tracing.cu:22).  RNG states are advanced in place.
 * Asynchronous on `stream`. */
int rtmi_render(const rtmi_scene *s, const rtmi_frame *f, void *d_states, float *d_tiles,
                uint32_t *d_ray_counts, void *stream);
/* Did the render complete?  Waits for `stream`, then returns RTMI_OK, or RTMI_ERR_INTERNAL when the render
 * abandoned a mesh search (the frame is then incomplete and must not be used).  `d_scratch`: the
 * rtmi_render_opts.d_scratch that call was given, or NULL for a call without one (the most recent such call on
 * this scene).  out_rays (nullable) receives the call's total of closest-hit queries.  The reference has no
 * counterpart: its CHECKs abort (utils.cu:164-166). */
int rtmi_render_status(const rtmi_scene *s, const void *d_scratch, uint64_t *out_rays, void *stream);
/* rtmi_render_status(s, NULL, out_rays, stream). */
int rtmi_last_ray_total(const rtmi_scene *s, uint64_t *out_rays, void *stream);
/* Diagnostic: the raw device counter words of a render ([0] work-queue head, [1] closest-hit queries,
 * [2] abandoned mesh searches, [3] head-queue cursor; a -DRTMI_STATS build of the kernels adds wave-level
 * step counts of the mesh search from word 4 on; a -DRTMI_CHECK_MARGINS build counts, in [33] / [34], the
 * sampled queries it re-did without the cull and the disagreements it found). */
#define RTMI_COUNTER_WORDS 40
int rtmi_debug_counters(const rtmi_scene *s, unsigned long long out[RTMI_COUNTER_WORDS], void *stream);
int rtmi_debug_counters_ex(const rtmi_scene *s, const void *d_scratch, unsigned long long out[RTMI_COUNTER_WORDS],
                           void *stream);

/* d_all_tiles holds the tile-major buffers of ranks 0..world_size-1 back to
 * back (what an RCCL gather to the root produces; world_size==1: the buffer
 * rtmi_render wrote).  Writes the row-major float[H*W][3] image the reference
 * keeps in d_image (ray_tracing.cu:84).  Asynchronous on `stream`. */
int rtmi_untile(const rtmi_frame *f, const float *d_all_tiles, float *d_image, void *stream);
/* Same for per-pixel ray counts. */
int rtmi_untile_u32(const rtmi_frame *f, const uint32_t *d_all_counts, uint32_t *d_image_counts, void *stream);

/* The N > 1 exchange, on caller-owned buffers and a caller-owned RCCL communicator (an ncclComm_t passed as
 * void*, rank == frame->rank, size == frame->world_size).  Replaces GatherImageData's MPI_Reduce of full frames on
 * host memory (utils.cu:115-130, 232-238): every rank sends its float[work_items][3] tile buffer to `root` over its
 * direct xGMI link (grouped ncclSend / ncclRecv), the root's d_all_tiles (world_size buffers back to back, what
 * rtmi_untile takes) also receives its own.  world_size 1: a device copy, comm may be NULL.  Asynchronous on
 * `stream`.  RCCL is looked up in the process at run time; librtmi.so does not link against it. */
int rtmi_gather(void *nccl_comm, const rtmi_frame *f, const float *d_tiles, float *d_all_tiles, int root, void *stream);
/* The reference's own decomposition (every rank renders the whole frame with GetWorkload's share of the samples,
 * post_process = 0): ncclReduce(sum) of the tile buffers into `root`, in place; follow with rtmi_untile and
 * rtmi_post_process.  In that split every rank's frame is the whole frame, so the number of ranks is the
 * communicator's, not the frame's: comm NULL means this rank is the only one (root must then be 0 and the call is a
 * no-op); with a communicator, root must be one of its ranks (checked with ncclCommCount). */
int rtmi_reduce_sum(void *nccl_comm, const rtmi_frame *f, float *d_tiles, int root, void *stream);

/* GatherImageData's root-side step on a summed image (utils.cu:126-129):
 * rgb = sqrt(clamp(rgb / spp, 0, 1)), in place over n_pixels*3 floats. */
int rtmi_post_process(float *d_image, int64_t n_pixels, int spp, void *stream);
/* GetWorkload (utils.cu:111-113). */
int rtmi_get_workload(int rank, int world_size, int spp);

/* Device self-test of the arithmetic shortcuts that are justified by exhaustion rather than by
 * argument alone; each is run against the expression it replaces on all 2^32 inputs, on the
 * current device (about a second):
 *  [0] the triangle test's `1.0f / det` (utils.cu:59) computed as hardware reciprocal + one FMA
 *      Newton step when |det| < 2^126: inputs with 2^-126 <= |x| < 2^126 where it differs from the
 *      IEEE quotient -- must be 0, the trace kernels rely on it;
 *  [1] differing inputs outside that range, where the kernels divide (informative, > 0);
 *  [2] CudaRandomFloat(-1, 1) (utils.cuh:22-27) as fma(x, 2^-31, 2^-32) - 1: draws x that differ
 *      from curand_uniform(x) * (1 - -1) + -1 -- must be 0;
 *  [3] CudaRandomFloat(0, 1) as fma(x, 2^-32, 2^-33) -- must be 0;
 *  [4] sqrtf(x) (glm::normalize's sqrt, ray.cu:10; lambertian.cu:25) as reciprocal square root + one FMA correction,
 *      on every binary32 in [2^-100, 2^100) -- must be 0;
 *  [5] the Lambertian sampler's `vec /= l` (lambertian.cu:29) as one correctly rounded reciprocal + two FMA
 *      corrections per coordinate, on 2^32 triples of sampler draws -- must be 0;
 *  [6] the same with one correction only (informative);  [7] unused. */
int rtmi_selftest_arithmetic(unsigned long long mismatches[8]);

/* Per-call scheduling options of rtmi_render_ex.  The reference fixes its launch shape at compile
 * time (dim3(8,8) blocks, utils.cu:158); here the shape and the work-queue order are run-time
 * parameters of ONE call -- no process state is involved.  A zero / negative field keeps the
 * library default.  None of them changes any pixel's value. */
typedef struct rtmi_render_opts {
  int32_t size;              /* sizeof(rtmi_render_opts) of the caller: must match the library's */
  int32_t schedule;          /* -1 default; 0 = tiles in image order; 1 = longest-first when it can pay (a 2-spp
                              * probe on scratch RNG states estimates each tile's cost; pixels are indivisible
                              * serial chains, so starting the expensive ones first shortens the end-of-frame
                              * tail); 2 = always longest-first */
  int32_t blocks_per_cu;     /* 0 default (as many workgroups per CU as fit) */
  int32_t threads_per_block; /* 0 default; a multiple of 64, at most 512 (256 for scenes without meshes) */
  int32_t sparse_stride;     /* 0 default: mesh frames put their outlier PIXELS (found by the probe, anywhere in the
                              * frame) at the head of the queue in three weight classes -- a wave each, two per
                              * wave, one per 16 lanes; > 0 (power of two, 1..64): the outlier TILES instead, taken
                              * by every sparse_stride-th lane only (the scheme before the classes, kept for
                              * comparison) */
  int32_t exclusive;         /* -1 default (1); 1: a wave holding an outlier pixel lets only the lanes its class allows
                              * take new pixels, the remaining lanes only help with that pixel's mesh searches; 0:
                              * they render too */
  int32_t outlier_x10;       /* 0 default (20); with sparse_stride > 0: a tile is an outlier from this many tenths of
                              * the mean tile cost */
  int32_t probe_spp;         /* 0 default: samples per pixel of the scheduler's first look at the frame (see first_pass), 1..64 */
  int32_t head_pct[3];       /* 0 default (80, 55, 30): mesh frames, per cent of the frame's largest probe count from
                              * which a pixel gets a wave to itself / shares one with another / gets one lane in 16 */
  int32_t plan;              /* -1 default (1).  List scenes (no mesh) with a probe behind the launch: 0 = every lane takes its
                              * pixels from the work queue; 1 = when the grid's waves have at most three tiles each, every
                              * wave walks a CHAIN of tiles planned before the launch (tiles dealt to the SIMDs, then to a
                              * SIMD's waves, in snakes over the longest-first order, so that all chains of a SIMD and all
                              * SIMDs cost about the same; needs wave_priority); 2 = chains for any number of tiles */
  int32_t wave_priority;     /* -1 default (16; 4 on frames below 64 spp; none below 8 spp); 0 = the hardware's oldest-wave-first issue order; N (a power of two) = every
                              * N iterations a wave publishes how many queries it still has to do and takes the s_setprio
                              * level its rank among the waves of its SIMD gives it (longest remaining chain first) */
  int32_t lane_stride;       /* 0 default: list scenes, a frame with fewer pixels than the grid has lanes is spread thin, one
                              * pixel per 2 / 4 / 8 / 16 lanes as far as the grid has room; else a power of two, 1..64 */
  int32_t promote_after;     /* -1 default (16); mesh frames: samples after which a pixel's own ray count may promote it to
                              * a head class (its wave then thins out around it); 0 = never */
  int32_t cost_probe;        /* -1 default (1); mesh frames: 1 = the probe books the lane-steps of its mesh searches on the
                              * pixels they serve and the queue's order follows that cost, 0 = it follows the ray counts */
  int32_t first_pass;        /* -1 default (1).  1 = the scheduler's probe is the frame's OWN first samples: samples [0, s1) of
                              * every pixel go into the caller's buffers, their ray counts order / plan the rest, a second
                              * launch resumes every pixel at sample s1 (s1 = probe_spp, or by default spp / 16, at most 64,
                              * for a frame that will be planned, else 2); N > 1 = the same with spp / N; 0 = probe_spp
                              * samples on a scratch copy of the RNG states, discarded */
  int32_t reserved;
  void *d_scratch;           /* optional device memory for ALL per-call state (work-queue cursors, ray total,
                              * completion flag, the scheduler's buffers), owned by the caller, at least */
  size_t scratch_bytes;      /* rtmi_render_scratch_bytes(frame) bytes: with it, concurrent renders of one scene
                              * (several streams, or N shards on one device) share nothing but the read-only scene,
                              * and rtmi_render_status(s, d_scratch, ..) reports on exactly that call.  NULL: the
                              * scene's own, which ties renders of that scene to one at a time. */
} rtmi_render_opts;
/* Bytes of d_scratch a render of this frame / shard needs. */
size_t rtmi_render_scratch_bytes(const rtmi_frame *f);
/* The launch a render of this frame would use on the current device: {workgroups, lanes per workgroup,
 * workgroups per compute unit, compute units}.  workgroups x lanes = the lanes resident at once, each of which
 * holds one pixel at a time (utils.cu:158 fixes dim3(8,8) blocks over the whole frame instead). */
int rtmi_render_launch_shape(const rtmi_scene *s, const rtmi_frame *f, const rtmi_render_opts *opts, int32_t out[4]);
/* How a render of this frame would be scheduled on the current device (what rtmi_render_ex decides before it launches
 * anything): out = {scheduled (1: a first pass orders / plans the rest), samples of the first pass (0: none), 1 if that
 * pass is the frame's own first samples (resumed) and 0 if it is a discarded probe, 1 if the frame is rendered as
 * planned chains (0: from the work queue), wave-priority interval in iterations (0: off), lane stride (1: every lane
 * takes pixels), waves of the grid, tiles of this shard}.  Informational: none of it changes a pixel. */
int rtmi_render_mode(const rtmi_scene *s, const rtmi_frame *f, const rtmi_render_opts *opts, int32_t out[8]);
/* rtmi_render with per-call options (opts == NULL: the defaults). */
int rtmi_render_ex(const rtmi_scene *s, const rtmi_frame *f, const rtmi_render_opts *opts, void *d_states,
                   float *d_tiles, uint32_t *d_ray_counts, void *stream);

/* Process-wide DEFAULTS for the same fields (what rtmi_render and a zero field of rtmi_render_opts use).
 * Kept for callers of the first ABI version; prefer rtmi_render_opts.  The RTMI_SPARSE_STRIDE /
 * RTMI_EXCLUSIVE / RTMI_OUTLIER_X10 / RTMI_HEAD_CLASSES (0: tiles) / RTMI_PROBE_SPP / RTMI_PLAN / RTMI_PRIO (wave_priority) /
 * RTMI_LANE_STRIDE / RTMI_PROMOTE (promote_after) / RTMI_COST_PROBE / RTMI_FIRST_PASS environment variables override the built-in defaults
 * of those fields and are read once, when the library is first used.  (RTMI_FETCH_BATCH / RTMI_FETCH_BATCH_FIRST, 1..64: the
 * largest batch a wave of a list frame draws from the work queue per atomic in longest-first order / in a first pass of a
 * few samples -- measurement knobs without an option field; defaults 16 / 64.) */
int rtmi_set_launch(int blocks_per_cu, int threads_per_block);
int rtmi_set_schedule(int mode);

#ifdef __cplusplus
}
#endif
#endif /* RTMI_H_ */
