// The wave-wide search of a mesh's 4-wide tree (DESIGN.md "Mesh queries").  Included by kernels.hip inside
// namespace rtmi, after trace_helpers.h (not a stand-alone header).
#pragma once

// ------------------------------------------------------------------ mesh search (wave-wide)
// The search of a mesh's 4-wide tree is done by the WAVE, not by the lane that owns the ray
// (DESIGN.md "Mesh queries").  What has to be found is every face the triangle test accepts with
// t_from <= t <= t_to -- no pruning by nearer hits, because the reference's box semantics (quirk g8)
// make farther hits matter -- so the order in which (ray, node) pairs are looked at is free.  All
// pending pairs of the wave's 64 rays sit on one stack in LDS; a step pops up to 64 of them, one per
// lane, whichever ray they belong to.  A wave whose rays need 3, 40 and 0 steps therefore takes
// ceil(43 / 64) steps per level instead of 40, and a single expensive ray is searched by all 64
// lanes: its latency is the depth of the tree, not the number of nodes it touches.
//
// Stack words: [ray lane : 6][payload : 26]; node entries (payload = node index) grow up from
// word 0, face-block entries (payload = first face * 8 + count) grow down from the top, so that a
// step pops entries of one kind.  A node step pops k <= 64 entries and pushes at most 4k; k is
// chosen so that `reserve` = 3 * depth + 3 + kMeshFaceSlack words stay free afterwards, or 1 when
// they would not.  Popping one node at a time is a depth-first search: above the level it started
// from the node end never holds more than 3 * depth entries, and the face end at most 67 (it is
// drained as soon as it holds 64, and a node adds at most 4), so from a state with `reserve` free
// words at least 4 stay free and the stack cannot overflow whatever the mesh.  (Should it ever, the
// search is abandoned and counters[2] reports it: no out-of-range access either way.)
__device__ __forceinline__ int lane_rank(unsigned long long mask) {  // set bits below my lane
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}
// Exclusive prefix sum of a small non-negative per-lane count over the wave's 64 lanes, in registers: four
// row_shr steps inside each row of 16 lanes, then the two row broadcasts of the gfx9 DPP unit (row_bcast:15 into
// rows 1 and 3, row_bcast:31 into rows 2 and 3).  Six add-with-DPP instructions in place of one ballot + mbcnt
// pair per bit of the count.
__device__ __forceinline__ int wave_prefix_excl(int v) {
  int s = v;
  s += __builtin_amdgcn_update_dpp(0, s, 0x111, 0xf, 0xf, true);   // row_shr:1
  s += __builtin_amdgcn_update_dpp(0, s, 0x112, 0xf, 0xf, true);   // row_shr:2
  s += __builtin_amdgcn_update_dpp(0, s, 0x114, 0xf, 0xf, true);   // row_shr:4
  s += __builtin_amdgcn_update_dpp(0, s, 0x118, 0xf, 0xf, true);   // row_shr:8
  s += __builtin_amdgcn_update_dpp(0, s, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
  s += __builtin_amdgcn_update_dpp(0, s, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
  return s - v;
}
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ float ubyte_f32(uint32_t x, int byte) { return (float)((x >> (8 * byte)) & 0xffu); }

// Distance slack of the search boxes.  The binary32 Moller-Trumbore test can accept a ray that
// misses the exact triangle by about 4e-7 |o - p0| / sin(smallest angle) (the error of
// dot(tvec, pvec) / det); the padded, outward-quantised boxes cover a fixed margin, and every box is
// widened by this fraction of (|o|_inf + largest mesh coordinate) >= |o - p0|_inf on top of it.
#ifndef MESH_DIST_SLACK  // (a diagnostic build sets it to 0 to show what the far-face tests catch)
#define MESH_DIST_SLACK kDistSlack
#endif

// The ray in a node's grid: per axis the time per grid step (idq) and the constants of
// t_lo = qlo * idq + ka, t_hi = qhi * idq + kb for the child planes qlo - rho and qhi + rho.
struct NodeFrame {
  float ka[3], kb[3], idq[3];
};
__device__ __forceinline__ void node_frame(uint4 w0, float4 r0, float4 r2, float mag, NodeFrame &f) {
  const float delta = MESH_DIST_SLACK * (fmaxf(fmaxf(fabsf(r0.x), fabsf(r0.y)), fabsf(r0.z)) + mag);
  const float oo[3] = {r0.x, r0.y, r0.z}, ii[3] = {r2.x, r2.y, r2.z};
  const float org[3] = {__uint_as_float(w0.x), __uint_as_float(w0.y), __uint_as_float(w0.z)};
#ifdef RTMI_NO_SLACK_EXP  // (diagnostic: what the exponent is for -- tools/gpu_check_margins.py finds the needles then)
  const int sk = 0;
#else
  const int sk = (int)w0.w >> 24;  // QNode4::slack_exp: thin faces below this node ask for 2^sk times the slack
#endif
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const int ex = (int)(int8_t)((w0.w >> (8 * a)) & 0xffu);
    const float oq = ldexpf(oo[a] - org[a], -ex);
    const float rho = ldexpf(delta, sk - ex);
    // finite even for a clamped reciprocal on a coarse grid: |q - oq| >= rho > 0 keeps the product
    // away from 0 * inf, and med3 keeps it below infinity
    f.idq[a] = __builtin_amdgcn_fmed3f(ldexpf(ii[a], ex), -1e35f, 1e35f);
    f.ka[a] = -(oq + rho) * f.idq[a];
    f.kb[a] = -(oq - rho) * f.idq[a];
  }
}
__device__ __forceinline__ bool child_box_hit(const NodeFrame &f, float qlx, float qly, float qlz, float qhx, float qhy,
                                              float qhz, float lo0, float hi0) {
  const float ql[3] = {qlx, qly, qlz}, qh[3] = {qhx, qhy, qhz};
  float en = -INFINITY, le = INFINITY;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const float tl = __builtin_fmaf(ql[a], f.idq[a], f.ka[a]);
    const float th = __builtin_fmaf(qh[a], f.idq[a], f.kb[a]);
    en = fmaxf(en, fminf(tl, th));
    le = fminf(le, fmaxf(tl, th));
  }
  const float lo = fmaxf(lo0, __builtin_fmaf(-fabsf(en), kSlabTimeRel, en));
  const float hi = fminf(hi0, __builtin_fmaf(fabsf(le), kSlabTimeRel, le));
  return lo <= hi;
}

// Record a face that passed the triangle test with parameter t in its ray's candidate list (`rr` =
// the ray record).  Same leaf: the smaller t wins, the higher reference index among equal t (what an
// in-order scan with `t <= t_to` keeps, bvh.cuh:127-134).  A full list keeps the leaves that come
// first in visiting order and moves `cut` down to the first leaf it had to leave to the next pass.
__device__ __forceinline__ void hit_list_insert(const SceneDev &sc, int *rr, uint32_t code, int face, int orig,
                                                float t) {
  const int cnt = rr[12];
  const uint32_t cut = (uint32_t)rr[13], lo_code = (uint32_t)rr[14];
  if (!(code >= lo_code && code < cut)) return;
  int found = -1, jmax = 0;
  uint32_t cmax = 0u;
#pragma unroll
  for (int j = 0; j < kHitSlots; j++) {
    if (j < cnt) {
      const uint32_t cj = (uint32_t)rr[16 + j * kHitWords];
      if (cj == code) found = j;
      if (cj >= cmax) cmax = cj, jmax = j;
    }
  }
  int slot = -1;
  if (found >= 0) {
    const float tj = __int_as_float(rr[16 + found * kHitWords + 2]);
    bool better = t < tj;
    if (t == tj) better = orig > sc.faces[rr[16 + found * kHitWords + 1]].orig;
    if (better) slot = found;
  } else if (cnt < kHitSlots) {
    slot = cnt;
    rr[12] = cnt + 1;
  } else if (code > cmax) {
    rr[13] = (int)code;  // this leaf and everything after it: next pass
  } else {
    rr[13] = (int)cmax;  // drop the last listed leaf instead
    slot = jmax;
  }
  if (slot >= 0) {
    rr[16 + slot * kHitWords] = (int)code;
    rr[16 + slot * kHitWords + 1] = face;
    rr[16 + slot * kHitWords + 2] = __float_as_int(t);
  }
}

// Diagnostic build only (-DRTMI_STATS, tools/mesh_stats.sh): wave-level step counts of the search.
#ifdef RTMI_STATS
struct MeshStats {
  unsigned searches, node_steps, face_steps, nodes_popped, blocks_popped, insert_rounds, steps_hist[6];
  // shader cycles (s_memtime) of this wave: [0] sample bookkeeping + camera ray, [1] world list before the mesh,
  // [2] mesh search, [3] replay, [4] shading; of the node steps: [5] pop + node/ray fetch, [6] box tests, [7] pushes;
  // [8] face steps incl. inserts
  unsigned long long cull_bits, cull_rays, cull_iters;  // culled list scan: candidate pairs, rays, wave iterations
  unsigned long long calib;  // two stamps back to back, once per search: what a stamp costs
  unsigned long long cyc[11];  // [9] search setup before the first step, [10] between steps (loop control)
};
__device__ unsigned long long g_wave_stats[16384][16];  // per wave: life, cyc[0..8], wave_queries, node_steps, face_steps
__device__ __forceinline__ unsigned long long stat_real() {
  __builtin_amdgcn_sched_barrier(0);
  unsigned long long t;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#if RTMI_STATS == 9  // counts and wave lifetimes only: no stamps inside the loop, timing as in the product build
__device__ __forceinline__ unsigned long long stat_now() { return 0ull; }
#else
__device__ __forceinline__ unsigned long long stat_now() { return stat_real(); }
#endif
#define RTMI_STAT(x) x
#if RTMI_STATS == 2
#define RTMI_STAT2(x) x  // per-step stamps (a stamp costs several hundred cycles: they distort what they measure)
#else
#define RTMI_STAT2(x)
#endif
#if RTMI_STATS == 3
#define RTMI_STAT3(x) x  // replay sections instead of the per-step stamps: [5] order + offsets, [6] (a), [7] (b), [8] (c)
#else
#define RTMI_STAT3(x)
#endif
#else
#define RTMI_STAT(x)
#define RTMI_STAT2(x)
#define RTMI_STAT3(x)
#endif

// One search pass for the lanes with `need`: afterwards every such lane's record holds, per
// reference leaf with lo_code <= code < cut, the best face with t_from <= t <= bt_to.
template <typename T, bool DT>
__device__ __forceinline__ void mesh_search(const SceneDev &sc, int sub_root, const BvhNode *top, float mag, int *wl,
                                            bool need, V3 o, V3 d, V3 inv_d, T bt_to, uint32_t lo_code,
                                            unsigned long long *overflow, bool count_work
#ifdef RTMI_STATS
                                            , MeshStats &st
#endif
) {
  const int lane = (int)(threadIdx.x & 63u);
  RTMI_STAT(st.searches++; unsigned my_steps = 0; const unsigned long long tset0 = stat_now();)
  int *stack = wl + 64 * kMeshRayWords;
  const float lo0 = T_FROM_F * kTimeLo;
  if (need) {
    int *rr = wl + lane * kMeshRayWords;
    int w3 = 0, w7 = 0;
    if (DT) {
      const double td = (double)bt_to;
      w3 = __double2loint(td), w7 = __double2hiint(td);
    } else {
      w3 = __float_as_int((float)bt_to);
    }
    *reinterpret_cast<int4 *>(rr + 0) = make_int4(__float_as_int(o.x), __float_as_int(o.y), __float_as_int(o.z), w3);
    *reinterpret_cast<int4 *>(rr + 4) = make_int4(__float_as_int(d.x), __float_as_int(d.y), __float_as_int(d.z), w7);
    *reinterpret_cast<int4 *>(rr + 8) = make_int4(__float_as_int(inv_d.x), __float_as_int(inv_d.y), __float_as_int(inv_d.z),
                                                  __float_as_int((float)bt_to * kTimeHi + kTimeAbs));
    *reinterpret_cast<int4 *>(rr + 12) = make_int4(0, (int)kCodeNone, (int)lo_code, 0);
  }
  const unsigned long long nm = __ballot(need);
  const int reserve = sc.sub_reserve;
  int sn = 0, sf = 0;
  if (__popcll(nm) <= kTopRays) {
    // Few rays (the tail of a frame, or a wave that holds outlier pixels): the levels below the root,
    // where a step has next to nothing to do, are skipped.  Lane l holds sub-tree l of the mesh's top
    // table; one ray at a time is tested against all of them at once and the sub-trees it touches go
    // onto the stack.  (Boxes as conservative as the node boxes they stand for: scene.hip.)
    const BvhNode te = top[lane];
    const float far = (float)bt_to * kTimeHi + kTimeAbs;
    for (unsigned long long m = nm; m != 0ull; m &= m - 1ull) {
      const int rl = __builtin_ctzll(m);  // wave-uniform
      if (kMeshStackWords - sn - sf - reserve < kTopEntries) {  // no room for a whole table: start this ray at the root
        if (lane == 0) stack[sn] = (rl << 26) | sub_root;
        sn++;
        continue;
      }
      const V3 ro = mk(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(o.x), rl)),
                       __int_as_float(__builtin_amdgcn_readlane(__float_as_int(o.y), rl)),
                       __int_as_float(__builtin_amdgcn_readlane(__float_as_int(o.z), rl)));
      const V3 ri = mk(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(inv_d.x), rl)),
                       __int_as_float(__builtin_amdgcn_readlane(__float_as_int(inv_d.y), rl)),
                       __int_as_float(__builtin_amdgcn_readlane(__float_as_int(inv_d.z), rl)));
      const float rfar = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(far), rl));
      const float delta = ldexpf(MESH_DIST_SLACK * (fmaxf(fmaxf(fabsf(ro.x), fabsf(ro.y)), fabsf(ro.z)) + mag), te.right);
      const bool hit = te.left != -1 && slab_touch(te, delta, ro, ri, lo0, rfar);
      const bool pn = hit && te.left >= 0, pf = hit && te.left < 0;
      const unsigned long long mn_ = __builtin_amdgcn_ballot_w64(pn), mf_ = __builtin_amdgcn_ballot_w64(pf);
      const uint32_t owner_bits = (uint32_t)rl << 26;
      if (pn) stack[sn + lane_rank(mn_)] = (int)(owner_bits | (uint32_t)te.left);
      if (pf) stack[kMeshStackWords - 1 - sf - lane_rank(mf_)] = (int)(owner_bits | (uint32_t)(-(te.left + 1)));
      sn += __popcll(mn_);
      sf += __popcll(mf_);
    }
  } else {
    if (need) stack[lane_rank(nm)] = (lane << 26) | sub_root;
    sn = __popcll(nm);
  }
  wave_lds_fence();
  RTMI_STAT(unsigned long long tprev = stat_now(); st.cyc[9] += tprev - tset0;
            { const unsigned long long cb = stat_now(); st.calib += cb - tprev; tprev = cb; } (void)tprev;)
  while ((sn | sf) != 0) {
    if (sf >= 64 || sn == 0) {
      // ---------------------------------------------------------------- face step
      // Few blocks pending (the tail of a search, or a wave with one deep ray among 64): four lanes
      // per block, one face each -- one memory round trip and one triangle test deep.  Otherwise one
      // lane per block of up to four faces.
      const bool wide = sf <= 16, pair = !wide && sf <= 32;  // four / two lanes per block, or one
      const int kf = sf < 64 ? sf : 64;
      RTMI_STAT(st.face_steps++; st.blocks_popped += kf; my_steps++;)
      RTMI_STAT2(const unsigned long long tf0 = stat_now(); st.cyc[10] += tf0 - tprev;)
      const int slot = wide ? (lane >> 2) : pair ? (lane >> 1) : lane;
      const bool mine = slot < kf;
      const int fstride = pair ? 2 : 1;  // result bit j stands for face first + j * fstride
      int e = 0;
      if (mine) e = stack[kMeshStackWords - sf + slot];
      sf -= kf;
      wave_lds_fence();
      unsigned pend = 0u;  // bit j: face first + j passed the test ...
      float pt0 = 0.f, pt1 = 0.f, pt2 = 0.f, pt3 = 0.f;  // ... with this t ...
      float po0 = 0.f, po1 = 0.f, po2 = 0.f, po3 = 0.f, pc0 = 0.f, pc1 = 0.f, pc2 = 0.f, pc3 = 0.f;  // ... orig, leaf
      const int owner = (int)((unsigned)e >> 26), fcnt = e & 7;
      int first = (e >> 3) & (kMeshMaxFaces - 1);
      int *rr = wl + owner * kMeshRayWords;
      // (cost probe) one unit of work per lane and step, booked on the ray it is done for: word 7 of the record, the
      // high half of a binary64 t_to, is free when t is binary32
      if (!DT && count_work && mine) atomicAdd(rr + 7, 1);
      if (wide) {
        const int j = lane & 3;
        first += j;
        if (mine && j < fcnt) {
          const float4 r0 = *reinterpret_cast<const float4 *>(rr + 0), r1 = *reinterpret_cast<const float4 *>(rr + 4);
          T t_to;
          if (DT) {
            t_to = (T)__hiloint2double(__float_as_int(r1.w), __float_as_int(r0.w));
          } else {
            t_to = (T)r0.w;
          }
          const float4 *fp4 = reinterpret_cast<const float4 *>(sc.faces + first);
          const float4 a = fp4[0], b = fp4[1], c = fp4[2];
          float t = 0.f, u = 0.f, v = 0.f;
          bool th = tri_test<T>(mk(a.x, a.y, a.z), mk(a.w, b.x, b.y), mk(b.z, b.w, c.x), mk(r0.x, r0.y, r0.z),
                                mk(r1.x, r1.y, r1.z), t_to, t, u, v);
#if defined(RTMI_ABLATE) && RTMI_ABLATE == 2
          th = th && t < -1.f;  // never
#endif
          if (th) pend = 1u, pt0 = t, po0 = c.y, pc0 = c.w;
        }
      } else if (pair) {
        first += lane & 1;  // this lane's faces: first, first + 2
        if (mine) {
          const float4 r0 = *reinterpret_cast<const float4 *>(rr + 0), r1 = *reinterpret_cast<const float4 *>(rr + 4);
          T t_to;
          if (DT) {
            t_to = (T)__hiloint2double(__float_as_int(r1.w), __float_as_int(r0.w));
          } else {
            t_to = (T)r0.w;
          }
          const V3 ro = mk(r0.x, r0.y, r0.z), rd = mk(r1.x, r1.y, r1.z);
          const float4 *fp4 = reinterpret_cast<const float4 *>(sc.faces + first);
          const int left = fcnt - (lane & 1);  // faces first + 2 * fi exist for 2 * fi < left
          float4 q[6];
#pragma unroll
          for (int w = 0; w < 3; w++) q[w] = fp4[w], q[3 + w] = fp4[6 + w];  // `faces` carries 4 records of padding
#pragma unroll
          for (int fi = 0; fi < 2; fi++) {
            if (2 * fi < left) {
              const float4 a = q[fi * 3], b = q[fi * 3 + 1], c = q[fi * 3 + 2];
              float t = 0.f, u = 0.f, v = 0.f;
              bool th = tri_test<T>(mk(a.x, a.y, a.z), mk(a.w, b.x, b.y), mk(b.z, b.w, c.x), ro, rd, t_to, t, u, v);
#if defined(RTMI_ABLATE) && RTMI_ABLATE == 2
              th = th && t < -1.f;  // never
#endif
              if (th) {
                pend |= 1u << fi;
                if (fi == 0) pt0 = t, po0 = c.y, pc0 = c.w;
                if (fi == 1) pt1 = t, po1 = c.y, pc1 = c.w;
              }
            }
          }
        }
      } else if (mine) {
        const float4 r0 = *reinterpret_cast<const float4 *>(rr + 0), r1 = *reinterpret_cast<const float4 *>(rr + 4);
        T t_to;
        if (DT) {
          t_to = (T)__hiloint2double(__float_as_int(r1.w), __float_as_int(r0.w));
        } else {
          t_to = (T)r0.w;
        }
        const V3 ro = mk(r0.x, r0.y, r0.z), rd = mk(r1.x, r1.y, r1.z);
        const float4 *fp4 = reinterpret_cast<const float4 *>(sc.faces + first);
#pragma unroll
        for (int half = 0; half < 2; half++) {
          if (half * 2 < fcnt) {
            float4 q[6];
#pragma unroll
            for (int w = 0; w < 6; w++) q[w] = fp4[half * 6 + w];  // `faces` carries 4 records of padding
#pragma unroll
            for (int fi = 0; fi < 2; fi++) {
              if (half * 2 + fi < fcnt) {
                const float4 a = q[fi * 3], b = q[fi * 3 + 1], c = q[fi * 3 + 2];
                float t = 0.f, u = 0.f, v = 0.f;
                bool th = tri_test<T>(mk(a.x, a.y, a.z), mk(a.w, b.x, b.y), mk(b.z, b.w, c.x), ro, rd, t_to, t, u, v);
#if defined(RTMI_ABLATE) && RTMI_ABLATE == 2
                th = th && t < -1.f;  // never
#endif
                if (th) {
                  pend |= 1u << (half * 2 + fi);
                  if (half * 2 + fi == 0) pt0 = t, po0 = c.y, pc0 = c.w;
                  if (half * 2 + fi == 1) pt1 = t, po1 = c.y, pc1 = c.w;
                  if (half * 2 + fi == 2) pt2 = t, po2 = c.y, pc2 = c.w;
                  if (half * 2 + fi == 3) pt3 = t, po3 = c.y, pc3 = c.w;
                }
              }
            }
          }
        }
      }
      // hits go to their ray's list; two lanes with hits for the same ray take turns
      while (__ballot(pend != 0u) != 0ull) {
        RTMI_STAT(st.insert_rounds++;)
        const bool has = pend != 0u;
        if (has) rr[15] = lane;
        wave_lds_fence();
        if (has && rr[15] == lane) {
          const int j = __builtin_ctz(pend);
          const float t = j == 0 ? pt0 : j == 1 ? pt1 : j == 2 ? pt2 : pt3;
          const float fo = j == 0 ? po0 : j == 1 ? po1 : j == 2 ? po2 : po3;
          const float fc = j == 0 ? pc0 : j == 1 ? pc1 : j == 2 ? pc2 : pc3;
          hit_list_insert(sc, rr, (uint32_t)__float_as_int(fc), first + j * fstride, __float_as_int(fo), t);
          pend &= pend - 1u;
        }
        wave_lds_fence();
      }
      RTMI_STAT2(tprev = stat_now(); st.cyc[8] += tprev - tf0;)
    } else {
      // ---------------------------------------------------------------- node step
      // four lanes per entry with one child box each, two with two, or one with all four (see the face step)
      const bool wide = sn <= 16, pair = !wide && sn <= 32;
      const int kmax = wide ? 16 : pair ? 32 : 64;
      int k = (kMeshStackWords - sn - sf - reserve) / 3;
      k = k < 1 ? 1 : k;
      k = k > kmax ? kmax : k;
      k = k > sn ? sn : k;
      if (3 * k > kMeshStackWords - sn - sf) {  // cannot happen (see above); never write out of range
        if (lane == 0) atomicAdd(overflow, 1ull);
        break;
      }
      RTMI_STAT(st.node_steps++; st.nodes_popped += k; my_steps++;)
      RTMI_STAT2(const unsigned long long tn0 = stat_now(); unsigned long long tn1 = tn0; st.cyc[10] += tn0 - tprev;)
      const int slot = wide ? (lane >> 2) : pair ? (lane >> 1) : lane;
      const bool mine = slot < k;
      int e = 0;
      if (mine) e = stack[sn - 1 - slot];
      sn -= k;
      wave_lds_fence();
      const uint32_t owner_bits = (uint32_t)e & 0xfc000000u;
      const int owner = (int)((unsigned)e >> 26), idx = e & (kMeshMaxNodes - 1);
      const uint4 *np = reinterpret_cast<const uint4 *>(sc.qnodes + idx);
      const int *rr = wl + owner * kMeshRayWords;
      if (!DT && count_work && mine) atomicAdd(wl + owner * kMeshRayWords + 7, 1);
      // children that were touched: nodes onto the node end, face blocks onto the face end
#define RTMI_PUSH_CHILD(H, C)                                                           \
  {                                                                                     \
    const bool pn = (H) && (C) >= 0, pf = (H) && (C) < 0;                               \
    const unsigned long long mn_ = __builtin_amdgcn_ballot_w64(pn), mf_ = __builtin_amdgcn_ballot_w64(pf); \
    if (pn) stack[sn + lane_rank(mn_)] = (int)(owner_bits | (uint32_t)(C));             \
    if (pf) stack[kMeshStackWords - 1 - sf - lane_rank(mf_)] = (int)(owner_bits | (uint32_t)(-((C) + 1))); \
    sn += __popcll(mn_);                                                                \
    sf += __popcll(mf_);                                                                \
  }
      if (pair) {
        bool ha = false, hb = false;
        int ca = -1, cb = -1;
        if (mine) {
          const int c = lane & 1;  // this lane's children: c, c + 2
          const uint4 w0 = np[0], w1 = np[1];
          const uint2 w2 = *reinterpret_cast<const uint2 *>(np + 2);
          ca = reinterpret_cast<const int *>(np + 3)[c], cb = reinterpret_cast<const int *>(np + 3)[c + 2];
          const float4 r0 = *reinterpret_cast<const float4 *>(rr + 0), r2 = *reinterpret_cast<const float4 *>(rr + 8);
          NodeFrame nf;
          node_frame(w0, r0, r2, mag, nf);
          const int sa = 8 * c, sb = 8 * c + 16;
          const bool ba = child_box_hit(nf, (float)((w1.x >> sa) & 0xffu), (float)((w1.y >> sa) & 0xffu),
                                        (float)((w1.z >> sa) & 0xffu), (float)((w1.w >> sa) & 0xffu),
                                        (float)((w2.x >> sa) & 0xffu), (float)((w2.y >> sa) & 0xffu), lo0, r2.w);
          const bool bb = child_box_hit(nf, (float)((w1.x >> sb) & 0xffu), (float)((w1.y >> sb) & 0xffu),
                                        (float)((w1.z >> sb) & 0xffu), (float)((w1.w >> sb) & 0xffu),
                                        (float)((w2.x >> sb) & 0xffu), (float)((w2.y >> sb) & 0xffu), lo0, r2.w);
          ha = (ca != -1) & ba, hb = (cb != -1) & bb;
        }
        RTMI_PUSH_CHILD(ha, ca)
        RTMI_PUSH_CHILD(hb, cb)
        wave_lds_fence();
      } else if (wide) {
        bool hit = false;
        int child = -1;
        if (mine) {
          const int c = lane & 3;
          const uint4 w0 = np[0], w1 = np[1];
          const uint2 w2 = *reinterpret_cast<const uint2 *>(np + 2);
          child = reinterpret_cast<const int *>(np + 3)[c];
          const float4 r0 = *reinterpret_cast<const float4 *>(rr + 0), r2 = *reinterpret_cast<const float4 *>(rr + 8);
          RTMI_STAT2(tn1 = stat_now();)
          NodeFrame nf;
          node_frame(w0, r0, r2, mag, nf);
          const int sh = 8 * c;
          // (tested whether or not the slot is used: the child word must not gate the other loads)
          const bool bh = child_box_hit(nf, (float)((w1.x >> sh) & 0xffu), (float)((w1.y >> sh) & 0xffu),
                                        (float)((w1.z >> sh) & 0xffu), (float)((w1.w >> sh) & 0xffu),
                                        (float)((w2.x >> sh) & 0xffu), (float)((w2.y >> sh) & 0xffu), lo0, r2.w);
          hit = (child != -1) & bh;
        }
        RTMI_STAT2(const unsigned long long tn2 = stat_now(); st.cyc[5] += tn1 - tn0; st.cyc[6] += tn2 - tn1;)
        const bool pn = hit && child >= 0, pf = hit && child < 0;
        const unsigned long long mn_ = __builtin_amdgcn_ballot_w64(pn), mf_ = __builtin_amdgcn_ballot_w64(pf);
        if (pn) stack[sn + lane_rank(mn_)] = (int)(owner_bits | (uint32_t)child);
        if (pf) stack[kMeshStackWords - 1 - sf - lane_rank(mf_)] = (int)(owner_bits | (uint32_t)(-(child + 1)));
        sn += __popcll(mn_);
        sf += __popcll(mf_);
        wave_lds_fence();
        RTMI_STAT2(tprev = stat_now(); st.cyc[7] += tprev - tn2;)
      } else {
        bool h0 = false, h1 = false, h2 = false, h3 = false;
        int c0 = -1, c1 = -1, c2 = -1, c3 = -1;
        if (mine) {
          const uint4 w0 = np[0], w1 = np[1], w2 = np[2], w3 = np[3];
          const float4 r0 = *reinterpret_cast<const float4 *>(rr + 0), r2 = *reinterpret_cast<const float4 *>(rr + 8);
          RTMI_STAT2(tn1 = stat_now();)
          NodeFrame nf;
          node_frame(w0, r0, r2, mag, nf);
          const uint32_t qlo[3] = {w1.x, w1.y, w1.z}, qhi[3] = {w1.w, w2.x, w2.y};
          const int cch[4] = {(int)w3.x, (int)w3.y, (int)w3.z, (int)w3.w};
          bool hh[4];
#pragma unroll
          for (int c = 0; c < 4; c++)
            hh[c] = cch[c] != -1 && child_box_hit(nf, ubyte_f32(qlo[0], c), ubyte_f32(qlo[1], c), ubyte_f32(qlo[2], c),
                                                  ubyte_f32(qhi[0], c), ubyte_f32(qhi[1], c), ubyte_f32(qhi[2], c), lo0, r2.w);
          h0 = hh[0], h1 = hh[1], h2 = hh[2], h3 = hh[3];
          c0 = cch[0], c1 = cch[1], c2 = cch[2], c3 = cch[3];
        }
        RTMI_STAT2(const unsigned long long tn2 = stat_now(); st.cyc[5] += tn1 - tn0; st.cyc[6] += tn2 - tn1;)
        RTMI_PUSH_CHILD(h0, c0)
        RTMI_PUSH_CHILD(h1, c1)
        RTMI_PUSH_CHILD(h2, c2)
        RTMI_PUSH_CHILD(h3, c3)
#undef RTMI_PUSH_CHILD
        wave_lds_fence();
        RTMI_STAT2(tprev = stat_now(); st.cyc[7] += tprev - tn2;)
      }
    }
  }
  RTMI_STAT(st.steps_hist[my_steps <= 1 ? 0 : my_steps <= 4 ? 1 : my_steps <= 8 ? 2 : my_steps <= 12 ? 3 : my_steps <= 20 ? 4 : 5]++;)
}

struct Hit {
  bool ok;
  float t;        // float(record.t)
  uint32_t win;   // winner id
  int32_t aux;    // BVH record index of the winner
  float u, v;     // raw barycentrics of the winning triangle
  int32_t work;   // (cost probe, mesh variants) lane-steps the wave spent on this ray's searches
};
