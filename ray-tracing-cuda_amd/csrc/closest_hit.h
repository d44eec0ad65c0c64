// HitableList::Hit over the flattened world: list scan (plain and culled), spheres, mesh search + replay.
// Included by kernels.hip inside namespace rtmi, after mesh_search.h (not a stand-alone header).
#pragma once

#ifdef RTMI_CHECK_MARGINS
// ================================================================== diagnostic build: the reference's own walk
// -DRTMI_CHECK_MARGINS (tools/gpu_check_margins.py): BVH::Hit as the reference performs it (bvh.cuh:123-183, bvh.cu:6-30)
// -- a depth-first walk of ITS tree by the lane that owns the ray, left subtree first, one running t_to; a child's box
// is tested when the walk reaches it, with AABB::Hit written out plane by plane (not through aabb_crossing_time, which
// the product's replay uses); a leaf scans its faces in the reference's order.  Nothing of the search structure, its
// padded boxes or distance slacks takes part, so a face the search lost shows up as a disagreement.
__device__ inline bool aabb_hit_reference(const BvhNode &nd, V3 o, V3 d, double t_from, double t_to) {
  const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
  for (int i = 0; i < 3; i++) {
    if (dd[i] == 0.f) continue;  // bvh.cu:24
    for (int s = 0; s < 2; s++) {
      const float tf = ((s == 0 ? nd.mn[i] : nd.mx[i]) - oo[i]) / dd[i];  // bvh.cu:25 (binary32, widened)
      const double t = (double)tf;
      if (t != t || fabs(t) == (double)INFINITY) continue;  // bvh.cu:8
      if (!(t_from <= t && t <= t_to)) continue;            // bvh.cu:9
      const float pt[3] = {oo[0] + (float)t * dd[0], oo[1] + (float)t * dd[1], oo[2] + (float)t * dd[2]};  // bvh.cu:10
      bool inside = true;
      for (int a = 0; a < 3; a++)
        if (a != i && !(nd.mn[a] <= pt[a] && pt[a] <= nd.mx[a])) inside = false;
      if (inside) return true;
    }
  }
  return false;
}
template <typename T>
__device__ inline bool bvh_reference_walk(const SceneDev &sc, const BvhRec &br, V3 o, V3 d, T &t_to, int &face, float &fu,
                                          float &fv) {
  int stk_node[kRefDepthMax + 2], stk_first[kRefDepthMax + 2], stk_n[kRefDepthMax + 2];
  int sp = 1;
  stk_node[0] = br.root, stk_first[0] = 0, stk_n[0] = br.n_faces;
  bool hit = false;
  while (sp > 0) {
    sp--;
    const int node = stk_node[sp], first = stk_first[sp], n = stk_n[sp];
    const BvhNode nd = sc.nodes[node];
    // bvh.cuh:138-150: every node but the root is entered iff its box is crossed inside [t_from, t_to] NOW
    if (node != br.root && !aabb_hit_reference(nd, o, d, 1e-3, (double)t_to)) continue;
    if (nd.right < 0) {  // leaf (bvh.cuh:125-136): faces in the reference's order, each acceptance lowers t_to
      for (int i = 0; i < n; i++) {
        const int fi = sc.face_of_orig[br.face_base + first + i];
        const FaceRec &f = sc.faces[fi];
        float t = 0.f, u = 0.f, v = 0.f;
        if (tri_test<T>(mk(f.p0[0], f.p0[1], f.p0[2]), mk(f.e1[0], f.e1[1], f.e1[2]), mk(f.e2[0], f.e2[1], f.e2[2]), o, d,
                        t_to, t, u, v))
          t_to = (T)t, face = fi, fu = u, fv = v, hit = true;
      }
    } else if (sp + 2 <= kRefDepthMax + 2) {
      const int mid = (n - 1) / 2;  // bvh.cuh:118
      stk_node[sp] = nd.right, stk_first[sp] = first + mid + 1, stk_n[sp] = n - mid - 1;
      stk_node[sp + 1] = nd.left, stk_first[sp + 1] = first, stk_n[sp + 1] = mid + 1;
      sp += 2;
    }
  }
  return hit;
}
#endif

// ================================================================== closest hit
// HitableList::Hit (hitable_list.cu:7-25) over the flattened world.  A nested
// Parallelepiped list is equivalent to its six parallelograms inlined at its
// position (DESIGN.md "List flattening").
// `live`: mesh variants are entered by ALL lanes of the wave (the mesh search borrows idle
// lanes); a lane that is not tracing passes live = false and gets an unused result.  The other
// variants are only entered by tracing lanes and pass true.
template <uint32_t F>
__device__ __forceinline__ Hit closest_hit(const SceneDev &sc, const BvhNode *s_nodes, int lds_nodes, const int *s_paths,
                                           int lds_paths, const float4 *s_pairs, int *ll, uint16_t *cands, int *wl,
                                           unsigned long long *overflow, V3 o, V3 d, bool live, bool count_work
#ifdef RTMI_STATS
                                           , MeshStats &st
#endif
) {
  constexpr bool DT = (F & F_SPHERE) != 0;
  typedef typename TSel<DT>::type T;
  bool ok = false;
  T t_to = (T)INFINITY;
  uint32_t win = ID_NONE;
  int32_t aux = 0;
  int32_t work = 0;
  float bu = 0.f, bv = 0.f;

  double sa = 0.0, sa2 = 0.0;
  float saf = 0.f;
  if (F & F_SPHERE) {
    float la = len3(d);       // sphere.cu:13: pow(length(dir), 2) in float, then widened
    saf = la * la;
    sa = (double)saf;
    sa2 = 2 * sa;
  }

  const int lane = (int)(threadIdx.x & 63u);
  const bool det_safe = sc.det_safe != 0;  // (a kernel argument: the branch on it waits for no vector result)
  // the ray as the culled list scan wants it: 1/d (the hardware reciprocal will do: the test is conservative by
  // a margin of 1e-5, not 1e-7), and -(o +- delta)/d per axis, delta = the distance slack of the mesh search
  // the culled list scan is on when the wave has its task region (ll) and the scene has pair records; the pairs'
  // corners come from LDS when the list was short enough to be staged (s_pairs), else from global memory
  const bool cull_list = (F & F_TRIS) && ll != nullptr && sc.n_pairs >= kCullMinPairs;  // (wave-uniform)
  V3 cull_inv = splat(0.f), cull_klo = splat(0.f), cull_khi = splat(0.f);
  if ((F & F_TRIS) && cull_list) {
    const float ix = __builtin_amdgcn_rcpf(d.x), iy = __builtin_amdgcn_rcpf(d.y), iz = __builtin_amdgcn_rcpf(d.z);
    cull_inv = mk(fabsf(d.x) < 1e-30f ? copysignf(1e30f, d.x) : ix, fabsf(d.y) < 1e-30f ? copysignf(1e30f, d.y) : iy,
                  fabsf(d.z) < 1e-30f ? copysignf(1e30f, d.z) : iz);
    const float delta = MESH_DIST_SLACK * (fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z)) + sc.list_mag);
    cull_klo = mk(-(o.x + delta) * cull_inv.x, -(o.y + delta) * cull_inv.y, -(o.z + delta) * cull_inv.z);  // lower planes, moved out
    cull_khi = mk(-(o.x - delta) * cull_inv.x, -(o.y - delta) * cull_inv.y, -(o.z - delta) * cull_inv.z);  // upper planes
  }

  for (int ri = 0; ri < sc.n_runs; ri++) {
    const i32x4 rv = load_run(sc.runs, ri);
    Run run;
    run.kind = rv[0], run.first = rv[1], run.count = rv[2], run.pad = rv[3];
    if (live && run.kind == RUN_SKY) {
      // sky.cu:18-27: t = 1e9; t_from <= 1e9 always holds
      const T ts = (T)1e9f;
      bool hit = ts <= t_to;
      bool acc = hit && (!ok || ts < t_to);
      ok = ok || acc;
      t_to = acc ? ts : t_to;
      win = acc ? make_id(RUN_SKY, 0) : win;
    }
    if ((F & F_TRIS) && run.kind == RUN_TRIS && cull_list) {
      // Culled scan (DESIGN.md "World-list scan").  The reference tests every entry of the list against
      // every ray (hitable_list.cu:11-22); what it RETURNS only depends on the entries whose test can
      // succeed, visited in list order.  Lanes of a wave carry unrelated rays, so no entry can be
      // skipped for the whole wave -- but each lane can skip its own: (1) every pair's padded bounds
      // (one s_load_dwordx8, wave-uniform) against the lane's ray: a slab test, 27 instructions
      // instead of the 140 of two triangle tests, builds a bit mask of the pairs this ray comes near;
      // (2) while any lane has bits left, each lane takes ITS next pair -- a different one per lane,
      // corners gathered from LDS -- and runs the reference's two triangle tests on it.  A lane visits
      // its pairs in list order with its own running t_to, so acceptance and ties are as in the full
      // scan; a pair outside the mask cannot pass the triangle test (the bounds carry the same padding
      // and distance slack as the mesh search boxes).
      const int pair0 = run.first >> 1;
      const float lo0 = T_FROM_F * kTimeLo;
      for (int c0 = 0; c0 < run.count; c0 += 32) {
        const int nc = run.count - c0 < 32 ? run.count - c0 : 32;
        const float hi0 = (float)t_to * kTimeHi + kTimeAbs;
        uint32_t mask = 0u;
        RTMI_STAT2(const unsigned long long tc0 = stat_now();)
        // t = (plane -+ delta - o) / d as one FMA per plane: plane * (1/d) - (o +- delta) * (1/d).  The rounding of
        // the two products is an error of ~6e-8 of the plane's coordinate in space, far inside delta.
        // Two SGPR buffers ping-pong (A, B: the loop is unrolled by hand so that no buffer is ever copied); the
        // verdict of pair i is shifted into `mask` from below (cull_step), bit nc - 1 - i, and the mask is turned
        // round once at the end.
#define RTMI_CULL_PAIR(bx)                                                                                                      \
  {                                                                                                                             \
    const float t0x = __builtin_fmaf(bx[0], cull_inv.x, cull_klo.x), t1x = __builtin_fmaf(bx[3], cull_inv.x, cull_khi.x); \
    const float t0y = __builtin_fmaf(bx[1], cull_inv.y, cull_klo.y), t1y = __builtin_fmaf(bx[4], cull_inv.y, cull_khi.y); \
    const float t0z = __builtin_fmaf(bx[2], cull_inv.z, cull_klo.z), t1z = __builtin_fmaf(bx[5], cull_inv.z, cull_khi.z); \
    const float en = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fminf(t0z, t1z));                                            \
    const float le = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z));                                            \
    cull_step(mask, fmaxf(lo0, en), fminf(hi0, le));                                                                            \
  }
        f32x8 A = load_pair_box(sc.pair_boxes, pair0 + c0), B;
        int i = 0;
        for (; i + 1 < nc; i += 2) {
          __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): see the plain scan below
          __builtin_amdgcn_sched_barrier(0);
          B = load_pair_box(sc.pair_boxes, pair0 + c0 + i + 1);
          __builtin_amdgcn_sched_barrier(0);
          RTMI_CULL_PAIR(A)
          __builtin_amdgcn_s_waitcnt(0xc07f);
          __builtin_amdgcn_sched_barrier(0);
          A = load_pair_box(sc.pair_boxes, pair0 + c0 + i + 2);  // (one inert record of padding at the end)
          __builtin_amdgcn_sched_barrier(0);
          RTMI_CULL_PAIR(B)
        }
        if (i < nc) {
          __builtin_amdgcn_s_waitcnt(0xc07f);
          __builtin_amdgcn_sched_barrier(0);
          RTMI_CULL_PAIR(A)
        }
#undef RTMI_CULL_PAIR
        mask = __brev(mask) >> (32 - nc);  // bit i = pair i of the chunk (1 <= nc <= 32)
        if (!live) mask = 0u;  // a lane without a ray of its own only helps
        RTMI_STAT2(const unsigned long long tc1 = stat_now(); st.cyc[5] += tc1 - tc0;)
        RTMI_STAT(st.cull_bits += (unsigned)__popcll(__builtin_amdgcn_ballot_w64(true)) * 0u; { unsigned pc = __builtin_popcount(mask); for (int off = 32; off > 0; off >>= 1) pc += __shfl_down(pc, off); st.cull_bits += __builtin_amdgcn_readfirstlane(pc); } st.cull_rays += (unsigned)__popcll(__builtin_amdgcn_ballot_w64(true));)
#ifndef RTMI_OWN_FIRST
#define RTMI_OWN_FIRST 1
#endif
#if RTMI_OWN_FIRST
        // ---- a lane's FIRST candidate of the chunk needs no hand-over: nearly every ray has one (the surface it
        // starts on, or the one it hits), so this round of tests is as full as a round of shared tasks, and the ray
        // is in the lane's own registers -- no task word, no ray record, no result words for it.  Its test sees the
        // lane's running t_to directly, as the reference's does (utils.cu:74).
        if (__builtin_amdgcn_ballot_w64(mask != 0u) != 0ull) {
          if (mask != 0u) {
            const int bit = __builtin_ctz(mask);
            mask &= mask - 1u;
            const size_t pidx = (size_t)(pair0 + c0 + bit) * 4;
            float4 qa, qb, qc, qd;
            if (s_pairs != nullptr) {  // (wave-uniform) staged in LDS
              const float4 *pp = s_pairs + pidx;
              qa = load_lds<float4>(pp), qb = load_lds<float4>(pp + 1), qc = load_lds<float4>(pp + 2), qd = load_lds<float4>(pp + 3);
            } else {
              const float4 *pp = reinterpret_cast<const float4 *>(sc.pair_pts) + pidx;
              qa = load_global<float4>(pp), qb = load_global<float4>(pp + 1), qc = load_global<float4>(pp + 2), qd = load_global<float4>(pp + 3);
            }
            const V3 p0 = mk(qa.x, qa.y, qa.z), p1 = mk(qa.w, qb.x, qb.y), p2 = mk(qb.z, qb.w, qc.x), p3 = mk(qc.y, qc.z, qc.w);
            float ta = 0.f, ua = 0.f, va = 0.f, tb = 0.f, ub = 0.f, vb = 0.f;
            const V3 e1 = p1 - p0, e2 = p2 - p0;
            const bool hit_a = tri_test_flat<T>(p0, e1, e2, cross3(d, e2), o, d, t_to, ta, ua, va, det_safe);
            bool hit_b = false;
            if (__float_as_int(qd.x) & PAIR_SECOND) {
              const V3 e1b = p2 - p1, e2b = p3 - p1;
              hit_b = tri_test_flat<T>(p1, e1b, e2b, cross3(d, e2b), o, d, t_to, tb, ub, vb, det_safe) && !hit_a;  // parallelogram.cu:33
            }
            const float t = hit_a ? ta : tb;
            const bool acc = (hit_a || hit_b) && (!ok || (T)t < t_to);
            ok = ok || acc;
            t_to = acc ? (T)t : t_to;
            win = acc ? make_id(RUN_TRIS, run.first + 2 * (c0 + bit) + (hit_a ? 0 : 1)) : win;
            if (F & F_TEX) {
              bu = acc ? (hit_a ? ua : ub) : bu;
              bv = acc ? (hit_a ? va : vb) : bv;
            }
          }
        }
#endif
        {
          // ---- the (remaining) candidates of all 64 rays are worked off by all 64 lanes.  A ray comes near 2.2 pairs on
          // average but the unluckiest of 64 near 6, and a lane-by-lane loop runs as long as that one.  So
          // (a) every lane writes its ray and one task per candidate pair to LDS (offsets: prefix sum of the
          // candidate counts by bit planes), (b) lane l takes task l, l + 64, ...: reads that ray and that
          // pair's corners and runs BOTH triangle tests against the ray's t_to at the start of the chunk,
          // (c) every lane folds the results of its own candidates in list order with its running t_to:
          // a test that passed against the older, larger t_to passes now iff its t <= the current one,
          // which is the only place t_to enters the test (utils.cu:74).
          constexpr int RW = (F & F_TEX) ? 6 : 2;  // result words per task: t of the two triangles (+ their u, v)
          int *tasks = ll + 64 * 8, *results = ll + 64 * 8 + kListTasks(F);
          const int cnt = __builtin_popcount(mask);
          const int base = wave_prefix_excl(cnt);
          if (cnt != 0) {
            int *rr = ll + lane * 8;
            int w3 = 0, w7 = 0;
            if (DT) {
              const double td = (double)t_to;
              w3 = __double2loint(td), w7 = __double2hiint(td);
            } else {
              w3 = __float_as_int((float)t_to);
            }
            *reinterpret_cast<int4 *>(rr) = make_int4(__float_as_int(o.x), __float_as_int(o.y), __float_as_int(o.z), w3);
            *reinterpret_cast<int4 *>(rr + 4) = make_int4(__float_as_int(d.x), __float_as_int(d.y), __float_as_int(d.z), w7);
          }
          bool todo = cnt != 0;
          while (__builtin_amdgcn_ballot_w64(todo) != 0ull) {
            RTMI_STAT(st.cull_iters++;)
            const int lo_t = __builtin_amdgcn_readlane(base, __builtin_ctzll(__builtin_amdgcn_ballot_w64(todo)));
            const bool now = todo && base + cnt - lo_t <= kListTasks(F);
            const int n_now = __builtin_amdgcn_readlane(base + cnt, 63 - __builtin_clzll(__builtin_amdgcn_ballot_w64(now))) - lo_t;
            if (now) {  // (a)
              int k = base - lo_t;
              for (uint32_t m = mask; m != 0u; m &= m - 1u) tasks[k++] = (lane << 5) | __builtin_ctz(m);
            }
            wave_lds_fence();
            for (int t0 = 0; t0 < n_now; t0 += 64) {  // (b)
              const int ti = t0 + lane;
              if (ti < n_now) {
                const int w = tasks[ti];
                const int *orr = ll + (w >> 5) * 8;
                const float4 r0 = *reinterpret_cast<const float4 *>(orr), r1 = *reinterpret_cast<const float4 *>(orr + 4);
                T t0_to;
                if (DT) {
                  t0_to = (T)__hiloint2double(__float_as_int(r1.w), __float_as_int(r0.w));
                } else {
                  t0_to = (T)r0.w;
                }
                const V3 ro = mk(r0.x, r0.y, r0.z), rd = mk(r1.x, r1.y, r1.z);
                const size_t pidx = (size_t)(pair0 + c0 + (w & 31)) * 4;
                float4 qa, qb, qc, qd;
                if (s_pairs != nullptr) {  // (wave-uniform) staged in LDS
                  const float4 *pp = s_pairs + pidx;
                  qa = load_lds<float4>(pp), qb = load_lds<float4>(pp + 1), qc = load_lds<float4>(pp + 2), qd = load_lds<float4>(pp + 3);
                } else {  // a list too long for the staging: per-lane gather of 64 bytes (L1 / L2 resident)
                  const float4 *pp = reinterpret_cast<const float4 *>(sc.pair_pts) + pidx;
                  qa = load_global<float4>(pp), qb = load_global<float4>(pp + 1), qc = load_global<float4>(pp + 2), qd = load_global<float4>(pp + 3);
                }
                const V3 p0 = mk(qa.x, qa.y, qa.z), p1 = mk(qa.w, qb.x, qb.y), p2 = mk(qb.z, qb.w, qc.x), p3 = mk(qc.y, qc.z, qc.w);
                float ta = 0.f, ua = 0.f, va = 0.f, tb = 0.f, ub = 0.f, vb = 0.f;
                const V3 e1 = p1 - p0, e2 = p2 - p0;  // utils.cu:54-55, the subtractions scene.hip: make_tri does
                const bool hit_a = tri_test_flat<T>(p0, e1, e2, cross3(rd, e2), ro, rd, t0_to, ta, ua, va, det_safe);
                bool hit_b = false;
                if (__float_as_int(qd.x) & PAIR_SECOND) {
                  const V3 e1b = p2 - p1, e2b = p3 - p1;
                  hit_b = tri_test_flat<T>(p1, e1b, e2b, cross3(rd, e2b), ro, rd, t0_to, tb, ub, vb, det_safe);
                }
                int *res = results + ti * RW;
                res[0] = hit_a ? __float_as_int(ta) : (int)0xffffffff;  // (a NaN pattern no t can have)
                res[1] = hit_b ? __float_as_int(tb) : (int)0xffffffff;
                if (F & F_TEX) {
                  res[2] = __float_as_int(ua), res[3] = __float_as_int(va), res[4] = __float_as_int(ub), res[5] = __float_as_int(vb);
                }
              }
            }
            wave_lds_fence();
            RTMI_STAT2(const unsigned long long tc2 = stat_now();)
            if (now) {  // (c)
              int k = base - lo_t;
              for (uint32_t m = mask; m != 0u; m &= m - 1u, k++) {
                const int tri = run.first + 2 * (c0 + __builtin_ctz(m));
                const int *res = results + k * RW;
                const int ia = res[0], ib = res[1];
                const float ta = __int_as_float(ia), tb = __int_as_float(ib);
                // parallelogram.cu:25-33 with the t_to of THIS moment: the first triangle, else the second
                const bool hit_a = ia != (int)0xffffffff && (T)ta <= t_to;
                const bool hit_b = !hit_a && ib != (int)0xffffffff && (T)tb <= t_to;
                const float t = hit_a ? ta : tb;
                const bool acc = (hit_a || hit_b) && (!ok || (T)t < t_to);
                ok = ok || acc;
                t_to = acc ? (T)t : t_to;
                win = acc ? make_id(RUN_TRIS, tri + (hit_a ? 0 : 1)) : win;
                if (F & F_TEX) {
                  bu = acc ? __int_as_float(hit_a ? res[2] : res[4]) : bu;
                  bv = acc ? __int_as_float(hit_a ? res[3] : res[5]) : bv;
                }
              }
              todo = false;
            }
            wave_lds_fence();
            RTMI_STAT2(st.cyc[7] += stat_now() - tc2;)
          }
        }
        RTMI_STAT2(st.cyc[6] += stat_now() - tc1;)  // (a) + (b) + (c); [7] is (c) alone
      }
    } else if ((F & F_TRIS) && live && run.kind == RUN_TRIS) {
      // Plain scan (lists too long for the LDS staging of the culled one).
      // Records come in (first, second) pairs: a Parallelogram's two triangles, or a lone
      // Triangle followed by an inert record.  Two SGPR buffers ping-pong: while record A
      // is tested the fetch of B is in flight, and vice versa.  Scalar-memory waits are
      // all-or-nothing (lgkmcnt counts SMEM out of order), so the order is pinned: wait
      // for the buffer about to be used, only then issue the next fetch, then test.
      const HotTri *base = sc.tris + run.first;
      f32x16 A = load_hot_tri(base, 0);
      for (int i = 0; i < run.count; i++) {
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): A has landed
        __builtin_amdgcn_sched_barrier(0);
        const f32x16 B = load_hot_tri(base, 2 * i + 1);
        __builtin_amdgcn_sched_barrier(0);
        float t = 0.f, u = 0.f, v = 0.f;
        const V3 pv_a = cross3(d, mk(A[6], A[7], A[8]));
        const bool hit_a = tri_test_flat<T>(mk(A[0], A[1], A[2]), mk(A[3], A[4], A[5]), mk(A[6], A[7], A[8]), pv_a, o, d,
                                            t_to, t, u, v, det_safe);
        {
          bool acc = hit_a && (!ok || (T)t < t_to);
          ok = ok || acc;
          t_to = acc ? (T)t : t_to;
          win = acc ? make_id(RUN_TRIS, run.first + 2 * i) : win;
          if (F & F_TEX) {
            bu = acc ? u : bu;
            bv = acc ? v : bv;
          }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);  // B has landed
        __builtin_amdgcn_sched_barrier(0);
        A = load_hot_tri(base, 2 * i + 2);  // next pair (or the inert padding pair)
        __builtin_amdgcn_sched_barrier(0);
        if (__float_as_int(B[13]) & TRI_SECOND) {  // wave-uniform: a lone Triangle has no second record
          // parallelogram.cu:33: the second triangle is tried only when the first missed
          V3 pv_b = pv_a;
          if (!(__float_as_int(B[13]) & TRI_SAME_E2)) pv_b = cross3(d, mk(B[6], B[7], B[8]));  // wave-uniform
          bool hit_b = tri_test_flat<T>(mk(B[0], B[1], B[2]), mk(B[3], B[4], B[5]), mk(B[6], B[7], B[8]), pv_b, o, d,
                                        t_to, t, u, v, det_safe);
          hit_b = hit_b && !hit_a;
          bool acc = hit_b && (!ok || (T)t < t_to);
          ok = ok || acc;
          t_to = acc ? (T)t : t_to;
          win = acc ? make_id(RUN_TRIS, run.first + 2 * i + 1) : win;
          if (F & F_TEX) {
            bu = acc ? u : bu;
            bv = acc ? v : bv;
          }
        }
      }
    }
    bool grouped_done = false;  // (wave-uniform) the grouped scan has answered this sphere run
    if ((F & F_SGROUP) && run.kind == RUN_SPHERE && run.pad >= 0 && cands != nullptr) {
      // Long sphere runs (scenes/spheres.cu: 484 small spheres): the reference tests every sphere against every ray
      // (hitable_list.cu:11-22, sphere.cu:11-44).  What that RETURNS for a stretch of spheres is order-free: a
      // sphere's candidate value v -- its near root if that is >= t_from, else its far root if that is (a root
      // beyond the running t_to fails either way) -- does not depend on t_to, a sphere is accepted iff v <= t_to
      // (v < t_to once anything was hit), so the stretch yields its smallest v, the first in list order among equal
      // ones, accepted against the incoming t_to by that same rule.  Hence:
      //  (1) the run's spheres sit in spatial groups of <= 16 (scene.hip); every lane slab-tests every group's padded
      //      bounds, widened by 2^-9 of (|o| + the group's own size): the binary32 operands of the discriminant
      //      (sphere.cu:13-17) let a ray that passes a sphere at distance m count as a hit up to
      //      m^2 <= r^2 + 6e-7 |o - c|^2, i.e. within 8e-4 |o - c| of its surface;
      //  (2) the touched (ray, group) pairs are tasks for all 64 lanes -- lane l takes task l, l + 64, ... -- which
      //      put the group's members through the binary32 pre-test of the plain loop below (exact by its margin) for
      //      that ray; the survivors go to the ray owner's candidate slots in LDS;
      //  (3) the (ray, sphere) candidates are tasks again: sphere.cu's binary64 arithmetic for that ray and sphere;
      //  (4) every lane folds its own results: smallest v, lowest list index among equal.
      // A ray with more candidates than slots (kSphCand: it passes through that many spheres' neighbourhoods) sends
      // its wave through the plain loop for this run instead.
      const int n_groups = (run.count + kSphGroupSize - 1) / kSphGroupSize;
      uint16_t *my_cands = cands + lane * kSphCand;
      int *tasks = ll + 64 * 8, *results = ll + 64 * 8 + kListTasks(F);
      int ccnt = 0;  // candidates of this lane's ray
      bool have = false;
      double best_v = 0.0;
      int best_idx = 0;
      {
        int *rr = ll + lane * 8;  // the ray as the tasks want it: origin, |d|^2 (binary32, sphere.cu:13), direction
        *reinterpret_cast<int4 *>(rr) = make_int4(__float_as_int(o.x), __float_as_int(o.y), __float_as_int(o.z), __float_as_int(saf));
        *reinterpret_cast<int4 *>(rr + 4) = make_int4(__float_as_int(d.x), __float_as_int(d.y), __float_as_int(d.z), 0);
      }
      const float sdelta = kSphDistSlack * fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z));  // (the groups' bounds carry 2^-9 of their own size)
      const float six = __builtin_amdgcn_rcpf(d.x), siy = __builtin_amdgcn_rcpf(d.y), siz = __builtin_amdgcn_rcpf(d.z);
      const V3 s_inv = mk(fabsf(d.x) < 1e-30f ? copysignf(1e30f, d.x) : six, fabsf(d.y) < 1e-30f ? copysignf(1e30f, d.y) : siy,
                          fabsf(d.z) < 1e-30f ? copysignf(1e30f, d.z) : siz);
      const V3 s_klo = mk(-(o.x + sdelta) * s_inv.x, -(o.y + sdelta) * s_inv.y, -(o.z + sdelta) * s_inv.z);
      const V3 s_khi = mk(-(o.x - sdelta) * s_inv.x, -(o.y - sdelta) * s_inv.y, -(o.z - sdelta) * s_inv.z);
      const float s_lo0 = T_FROM_F * kTimeLo, s_hi0 = (float)t_to * kTimeHi + kTimeAbs;
      const int member0 = __float_as_int(load_sph_group(sc.sph_groups, run.pad)[6]);  // first member of the run
      // (3) + (4): all 64 lanes, whoever's the candidates are
      auto flush = [&]() {
        const int cnt = ccnt;
        const int base = wave_prefix_excl(cnt);
        bool todo = cnt != 0;
        while (__builtin_amdgcn_ballot_w64(todo) != 0ull) {
          const int lo_t = __builtin_amdgcn_readlane(base, __builtin_ctzll(__builtin_amdgcn_ballot_w64(todo)));
          const bool now = todo && base + cnt - lo_t <= kListTasks(F);
          const int n_now = __builtin_amdgcn_readlane(base + cnt, 63 - __builtin_clzll(__builtin_amdgcn_ballot_w64(now))) - lo_t;
          if (now)
            for (int j = 0; j < cnt; j++) tasks[base - lo_t + j] = (lane << 16) | (int)my_cands[j];
          wave_lds_fence();
          for (int t0 = 0; t0 < n_now; t0 += 64) {
            const int ti = t0 + lane;
            if (ti < n_now) {
              const int w = tasks[ti];
              const int *orr = ll + (w >> 16) * 8;
              const float4 r0 = *reinterpret_cast<const float4 *>(orr), r1 = *reinterpret_cast<const float4 *>(orr + 4);
              const SphMember *mp = sc.sph_members + member0 + (w & 0xffff);  // per-lane gather, 32 bytes (L1 / L2 resident)
              const float4 m0 = *reinterpret_cast<const float4 *>(mp);
              const int4 m1 = *reinterpret_cast<const int4 *>(reinterpret_cast<const char *>(mp) + 16);
              const double r2 = __hiloint2double(m1.w, m1.z);
              const V3 ro = mk(r0.x, r0.y, r0.z), rd = mk(r1.x, r1.y, r1.z);
              // sphere.cu:13-23, operation by operation (as the plain loop below)
              const double tsa = (double)r0.w, tsa2 = 2 * tsa;
              const V3 oc = ro - mk(m0.x, m0.y, m0.z);
              const double b = (double)(2.0f * dot3(rd, oc));
              const float lc = len3(oc);
              const double c = (double)(lc * lc) - r2;
              const double disc = b * b - 4 * tsa * c;
              double v = __hiloint2double(0x7ff80000, 0);  // NaN: no candidate value
              if (!(disc < 0)) {
                const double sq = sqrt(disc);
                const double t1 = (-b - sq) / tsa2;
                if (1e-3 <= t1) {
                  v = t1;
                } else {
                  const double t2 = (-b + sq) / tsa2;
                  if (1e-3 <= t2) v = t2;
                }
              }
              int *res = results + ti * 3;
              res[0] = __double2loint(v), res[1] = __double2hiint(v), res[2] = m1.x;  // m1.x: SphMember::orig
            }
          }
          wave_lds_fence();
          if (now) {
            for (int j = 0; j < cnt; j++) {
              const int *res = results + (base - lo_t + j) * 3;
              const double v = __hiloint2double(res[1], res[0]);
              const int idx = res[2];
              if (v == v && (!have || v < best_v || (v == best_v && idx < best_idx))) have = true, best_v = v, best_idx = idx;
            }
            todo = false;
          }
          wave_lds_fence();
        }
        ccnt = 0;
      };
      // (1) + (2): group bounds per lane (one bit per group, 32 groups at a time), then the touched (ray, group) pairs
      // as tasks for all 64 lanes: a task pre-tests the group's members against that ray and appends the survivors
      // to the RAY OWNER's candidate slots (an LDS counter per lane).
      int *counts = reinterpret_cast<int *>(cands + 64 * kSphCand);
      counts[lane] = 0;
      bool over = false;  // this lane found no room for a candidate of some ray: the wave falls back to the plain loop
      for (int g0 = 0; g0 < n_groups; g0 += 32) {
        const int ngc = n_groups - g0 < 32 ? n_groups - g0 : 32;
        uint32_t gmask = 0u;
        f32x8 G = load_sph_group(sc.sph_groups, run.pad + g0);
        for (int g = 0; g < ngc; g++) {
          __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): G has landed (see the triangle scan)
          __builtin_amdgcn_sched_barrier(0);
          const f32x8 gb = G;
          G = load_sph_group(sc.sph_groups, run.pad + g0 + g + 1);  // (one inert record follows the last group)
          __builtin_amdgcn_sched_barrier(0);
          const float t0x = __builtin_fmaf(gb[0], s_inv.x, s_klo.x), t1x = __builtin_fmaf(gb[3], s_inv.x, s_khi.x);
          const float t0y = __builtin_fmaf(gb[1], s_inv.y, s_klo.y), t1y = __builtin_fmaf(gb[4], s_inv.y, s_khi.y);
          const float t0z = __builtin_fmaf(gb[2], s_inv.z, s_klo.z), t1z = __builtin_fmaf(gb[5], s_inv.z, s_khi.z);
          const float en = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fminf(t0z, t1z));
          const float le = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z));
          cull_step(gmask, fmaxf(s_lo0, en), fminf(s_hi0, le));
        }
        gmask = live ? __brev(gmask) >> (32 - ngc) : 0u;  // bit g = group g0 + g
        const int cnt = __builtin_popcount(gmask);
        const int base = wave_prefix_excl(cnt);
        bool todo = cnt != 0;
        while (__builtin_amdgcn_ballot_w64(todo) != 0ull) {
          const int lo_t = __builtin_amdgcn_readlane(base, __builtin_ctzll(__builtin_amdgcn_ballot_w64(todo)));
          const bool now = todo && base + cnt - lo_t <= kListTasks(F);
          const int n_now = __builtin_amdgcn_readlane(base + cnt, 63 - __builtin_clzll(__builtin_amdgcn_ballot_w64(now))) - lo_t;
          if (now) {
            int k = base - lo_t;
            for (uint32_t m = gmask; m != 0u; m &= m - 1u) tasks[k++] = (lane << 16) | (g0 + __builtin_ctz(m));
          }
          wave_lds_fence();
          for (int t0 = 0; t0 < n_now; t0 += 64) {
            const int ti = t0 + lane;
            if (ti < n_now) {
              const int w = tasks[ti];
              const int owner = w >> 16;
              const int *orr = ll + owner * 8;
              const float4 r0 = *reinterpret_cast<const float4 *>(orr), r1 = *reinterpret_cast<const float4 *>(orr + 4);
              const V3 ro = mk(r0.x, r0.y, r0.z), rd = mk(r1.x, r1.y, r1.z);
              const float ta4 = 4.0f * r0.w;
              const int2 gr = *reinterpret_cast<const int2 *>(reinterpret_cast<const char *>(sc.sph_groups + run.pad + (w & 0xffff)) + 24);
              const int rel = gr.x - member0;
              const float4 *mp = reinterpret_cast<const float4 *>(sc.sph_members + gr.x);  // per-lane gathers (L1 / L2 resident)
#pragma unroll 4
              for (int i = 0; i < kSphGroupSize; i++) {
                if (i < gr.y) {
                  // the wave-level cull of the plain loop, per ray: when the binary32 discriminant is below -1e-5 of
                  // its terms' magnitude the exact one is negative; anything else (NaN included) stays a candidate
                  const float4 mb = mp[2 * i];
                  const V3 oc = ro - mk(mb.x, mb.y, mb.z);
                  const float bf = 2.0f * dot3(rd, oc);
                  const float oc2 = dot3(oc, oc);
                  const float bb = bf * bf;
                  const float disc_f = bb - ta4 * (oc2 - mb.w);
                  const float mag = bb + ta4 * (oc2 + mb.w);
                  if (!(disc_f < -kSphDiscRel * mag)) {
                    const int slot = atomicAdd(&counts[owner], 1);
                    if (slot < kSphCand) {
                      cands[owner * kSphCand + slot] = (uint16_t)(rel + i);
                    } else {
                      over = true;
                    }
                  }
                }
              }
            }
          }
          wave_lds_fence();
          if (now) todo = false;
        }
      }
      const bool overflowed = __any(over);  // (wave-uniform)
      if (!overflowed) {
        ccnt = counts[lane];
        flush();
        if (have) {  // hitable_list.cu:13-19 against the state the run was entered with
          const bool hit = best_v <= (double)t_to;
          const bool acc = hit && (!ok || best_v < (double)t_to);
          ok = ok || acc;
          t_to = acc ? (T)best_v : t_to;
          win = acc ? make_id(RUN_SPHERE, best_idx) : win;
        }
        grouped_done = true;
      }
    }
    if ((F & F_SPHERE) && live && run.kind == RUN_SPHERE && !grouped_done) {
      f32x8 nxt = load_sphere(sc.spheres, run.first);
      for (int i = 0; i < run.count; i++) {
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): see the triangle loop
        __builtin_amdgcn_sched_barrier(0);
        const f32x8 cur = nxt;
        nxt = load_sphere(sc.spheres, run.first + i + 1);
        __builtin_amdgcn_sched_barrier(0);
        const double r2 = __hiloint2double(__float_as_int(cur[7]), __float_as_int(cur[6]));
        V3 oc = o - mk(cur[0], cur[1], cur[2]);
        const float bf = 2.0f * dot3(d, oc);
        {
          // Wave-level cull.  The reference's discriminant is b*b - 4*a*c in double from the
          // binary32 values b, a = |d|^2, |oc|^2 (sphere.cu:13-17).  The same expression in
          // binary32 (with dot(oc,oc) for |oc|^2, i.e. without the square root) is off by at
          // most a few ulp of its largest term; when it is below -1e-5 of the terms' magnitude
          // on EVERY lane the exact discriminant is negative on every lane, no lane can hit, and
          // the binary64 part is skipped.  Lanes of a wave carry unrelated rays, but a small
          // sphere is in the way of few of them.  (NaN/inf compare false: not skipped.)
          const float oc2 = dot3(oc, oc), r2f = (float)r2;
          const float disc_f = bf * bf - 4.0f * saf * (oc2 - r2f);
          const float mag = bf * bf + 4.0f * saf * (oc2 + r2f);
          if (!__any(!(disc_f < -kSphDiscRel * mag))) continue;
        }
        double b = (double)bf;
        float lc = len3(oc);
        double c = (double)(lc * lc) - r2;
        double disc = b * b - 4 * sa * c;
        bool hit = false;
        double t = 0.0;
        if (!(disc < 0)) {
          double sq = sqrt(disc);
          t = (-b - sq) / sa2;
          hit = (1e-3 <= t && t <= (double)t_to);
          if (!hit) {
            t = (-b + sq) / sa2;
            hit = (1e-3 <= t && t <= (double)t_to);
          }
        }
        bool acc = hit && (!ok || t < (double)t_to);
        ok = ok || acc;
        t_to = acc ? (T)t : t_to;
        win = acc ? make_id(RUN_SPHERE, run.first + i) : win;
      }
    }
#if defined(RTMI_ABLATE) && RTMI_ABLATE == 3
    if (false) {
#else
    if ((F & F_BVH) && run.kind == RUN_BVH) {
#endif
      // BVH::Hit (bvh.cuh:123-183) answered without walking the reference's tree.
      //
      // What the reference computes: a depth-first walk, left subtree first, with a running
      // t_to; a child is entered iff AABB::Hit(child box, [t_from, t_to]) holds at that moment
      // (the root's box is never tested); an entered leaf scans its faces in order, accepts
      // every t_from <= t <= t_to and lowers t_to to it; the last acceptance is the answer.
      // Hence (DESIGN.md "Mesh queries"):
      //  * a leaf's contribution at running bound T is its best face -- smallest t, highest
      //    reference index among equal t -- provided that t <= T; it does not depend on T
      //    otherwise, nor on the order the faces are looked at;
      //  * leaves without a hit change nothing; whether their boxes were entered is irrelevant;
      //  * a box test is only ever needed on the root-to-leaf path of a leaf that holds a hit,
      //    and it sees the t_to left by the hit leaves before it in visiting order.
      // So: (1) one search of the mesh-wide 4-wide tree (mesh_search: by the whole wave, for all
      // its rays at once) collects, per reference leaf, the best face in the ray's small list
      // keyed by the leaf's path code; (2) each lane replays its listed leaves in visiting order,
      // evaluating the reference's exact box test on the path nodes not shared with the previously
      // replayed leaf.  If more leaves hold hits than the list has slots, the leaves beyond `cut`
      // are left to a further search pass.
      const V3 inv_d = mk(safe_inverse(d.x), safe_inverse(d.y), safe_inverse(d.z));
      int *rr = wl + lane * kMeshRayWords;
      for (int i = 0; i < run.count; i++) {
        BvhRec br;
        {
          const i32x16 bw = load_bvh_rec(sc.bvhs, run.first + i);  // wave-uniform: one scalar load, root bounds included
          br.root = bw[0], br.mat = bw[1], br.has_uv = bw[2], br.face_base = bw[3], br.sub_root = bw[4];
          br.mag = __int_as_float(bw[5]);
          br.ref_depth = bw[6], br.path_base = bw[7];
#pragma unroll
          for (int c = 0; c < 3; c++) br.root_mn[c] = __int_as_float(bw[8 + c]), br.root_mx[c] = __int_as_float(bw[11 + c]);
          br.n_faces = bw[14], br.slack_exp = bw[15];
        }
        T bt_to = t_to;
        bool bhit = false;
        int bface = 0;
        float fu = 0.f, fv = 0.f;
#ifdef RTMI_CHECK_MARGINS
        if (wl == nullptr) {  // (wave-uniform) the checker's second answer: no search, no replay
          if (live) bhit = bvh_reference_walk<T>(sc, br, o, d, bt_to, bface, fu, fv);
          const bool acc = bhit && (!ok || bt_to < t_to);
          ok = ok || acc;
          t_to = acc ? bt_to : t_to;
          win = acc ? make_id(RUN_BVH, bface) : win;
          aux = acc ? run.first + i : aux;
          bu = acc ? fu : bu;
          bv = acc ? fv : bv;
          continue;
        }
#endif
        // replay state: path code of the last replayed leaf and, left-aligned like the code,
        // one bit per level "that node of its path was entered"
        bool have_prev = false;
        uint32_t prev_code = 0u, entered = 0u;
        uint32_t lo_code = 0u;
        bool need = live;
#ifndef RTMI_MESH_PRETEST
#define RTMI_MESH_PRETEST 1
#endif
#if RTMI_MESH_PRETEST
        // A ray that stays clear of the mesh's bounds needs no search (and a wave of such rays -- most of a frame is
        // background -- skips search and replay altogether): when the reference tree has inner nodes, a face only
        // counts if the ray crosses the exact box of a child of the root (bvh.cuh:138-150), which lies inside the
        // root's.  The root's bounds are padded far beyond what that binary32 test can get wrong.  (A mesh whose root is
        // a leaf has no box test at all: every face the triangle test accepts counts, from however far off.)
        if (br.ref_depth > 0) {
          BvhNode rn;
          rn.mn[0] = br.root_mn[0], rn.mn[1] = br.root_mn[1], rn.mn[2] = br.root_mn[2];
          rn.mx[0] = br.root_mx[0], rn.mx[1] = br.root_mx[1], rn.mx[2] = br.root_mx[2];
          const float diag = fmaxf(fmaxf(rn.mx[0] - rn.mn[0], rn.mx[1] - rn.mn[1]), rn.mx[2] - rn.mn[2]);
          const float pad = 1e-3f * diag + 1e-4f * br.mag +
                            ldexpf(0x1p-15f * (fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z)) + br.mag), br.slack_exp);
          need = need && slab_touch(rn, pad, o, inv_d, T_FROM_F * kTimeLo, (float)bt_to * kTimeHi + kTimeAbs);
        }
#endif
        // All 64 lanes walk this loop together (lanes without a ray with need == false): the search
        // is the wave's.
        while (__ballot(need) != 0ull) {
          // ---- (1) search: best face per leaf with lo_code <= code < cut, t <= bt_to
          RTMI_STAT(const unsigned long long ts0 = stat_now();)
          mesh_search<T, DT>(sc, br.sub_root, sc.tops + (size_t)(run.first + i) * kTopEntries, br.mag, wl, need, o, d, inv_d,
                             bt_to, lo_code, overflow, count_work
#ifdef RTMI_STATS
                             , st
#endif
          );
          if (!DT && count_work && need) work += rr[7];
          RTMI_STAT(const unsigned long long ts1 = stat_now(); st.cyc[2] += ts1 - ts0;)
          // ---- (2) replay the listed leaves in the reference's visiting order.  The box tests are
          // spread over the wave: AABB::Hit(box, [t_from, T]) is `crossing time <= T` with a crossing time
          // that does not depend on T (aabb_crossing_time), so (a) every lane with listed leaves writes one
          // word per leaf (its lane, the leaf's row in the mesh's path table) into the search's (now empty)
          // stack, (b) all 64 lanes work off the (leaf, level) pairs, whoever's they are: node from the path
          // table, its box, that ray, the crossing time, (c) each lane walks its leaves with the running
          // t_to, looking the crossing times up.  A lane whose leaves do not fit next to the others' waits
          // for the next round.
          RTMI_STAT(const unsigned long long tr0 = stat_now(); (void)tr0;)
          uint32_t cut = kCodeNone;
          int cnt = 0;
          int4 hs[kHitSlots];  // my entries: leaf, face, t
#pragma unroll
          for (int j = 0; j < kHitSlots; j++) hs[j] = make_int4((int)kCodeNone, 0, 0, 0);
          if (need) {
            const int4 head = *reinterpret_cast<const int4 *>(rr + 12);
            cnt = head.x, cut = (uint32_t)head.y;
            const int4 wa = *reinterpret_cast<const int4 *>(rr + 16), wb = *reinterpret_cast<const int4 *>(rr + 20),
                       wc = *reinterpret_cast<const int4 *>(rr + 24);
            const int4 e4[kHitSlots] = {make_int4(wa.x, wa.y, wa.z, 0), make_int4(wa.w, wb.x, wb.y, 0),
                                        make_int4(wb.z, wb.w, wc.x, 0), make_int4(wc.y, wc.z, wc.w, 0)};
#pragma unroll
            for (int j = 0; j < kHitSlots; j++)
              if (j < cnt && (uint32_t)e4[j].x < cut) hs[j] = e4[j];  // (the rest was pushed beyond `cut`: next pass)
          }
          // ... in visiting order = by ascending leaf ordinal (never kCodeNone): a five-exchange network
#define RTMI_ORDER(A, B)                                        \
  {                                                             \
    const bool sw = (uint32_t)hs[B].x < (uint32_t)hs[A].x;      \
    const int4 lo_ = sw ? hs[B] : hs[A], hi_ = sw ? hs[A] : hs[B]; \
    hs[A] = lo_, hs[B] = hi_;                                   \
  }
          RTMI_ORDER(0, 1) RTMI_ORDER(2, 3) RTMI_ORDER(0, 2) RTMI_ORDER(1, 3) RTMI_ORDER(1, 2)
#undef RTMI_ORDER
          int nleaf = 0;
#pragma unroll
          for (int j = 0; j < kHitSlots; j++) nleaf += (uint32_t)hs[j].x != kCodeNone ? 1 : 0;
          const int depth_r = br.ref_depth;
          const int log_d = depth_r < 8 ? 3 : depth_r < 16 ? 4 : 5;  // a leaf's row: its path code + the crossing times, 8, 16 or 32 words
          // (rows of this round: `leaves` holds 64 of them, `times` the rest of the stack's words)
          const int rows_max = (kMeshStackWords - 64) >> log_d < 64 ? (kMeshStackWords - 64) >> log_d : 64;
          // exclusive prefix sum of nleaf (0..4) over the wave in registers: no LDS round trips
          const int base = wave_prefix_excl(nleaf);
          int *leaves = wl + 64 * kMeshRayWords;  // [64] one word per listed leaf of this round
          int *times = leaves + 64;                // rows of crossing times
#if defined(RTMI_ABLATE) && RTMI_ABLATE == 4
          bool todo = false;  // (diagnostic: no replay, nearest listed hit)
#else
          bool todo = nleaf > 0 && depth_r > 0;
#endif
          RTMI_STAT3(unsigned long long tr1 = stat_now(); st.cyc[5] += tr1 - ts1;)
          while (__builtin_amdgcn_ballot_w64(todo) != 0ull) {
            const int lo = __builtin_amdgcn_readlane(base, __builtin_ctzll(__builtin_amdgcn_ballot_w64(todo)));
            const bool now = todo && base + nleaf - lo <= rows_max;
            const int n_rows = __builtin_amdgcn_readlane(base + nleaf, 63 - __builtin_clzll(__builtin_amdgcn_ballot_w64(now))) - lo;
            if (now) {  // (a)
#pragma unroll
              for (int k = 0; k < kHitSlots; k++)
                if (k < nleaf) leaves[base - lo + k] = (lane << 26) | hs[k].x;
            }
            wave_lds_fence();
            RTMI_STAT3(const unsigned long long tr2 = stat_now(); st.cyc[6] += tr2 - tr1;)
            // (b) crossing times, one (leaf, level) per lane and round
            const int n_now = n_rows << log_d;
            for (int t0 = 0; t0 < n_now; t0 += 64) {
              const int t = t0 + lane, lvl = t & ((1 << log_d) - 1);  // word 0 of a row: the leaf's path code
              if (t < n_now && lvl <= depth_r) {
                const int w = leaves[t >> log_d];
                const int pi = br.path_base + (w & (kMeshMaxNodes - 1)) * (depth_r + 1) + lvl;
                const int ni = pi < lds_paths ? s_paths[pi] : sc.leaf_paths[pi];
                float m = __int_as_float(lvl == 0 ? ni : (int)0xffffffff);  // marker: past the leaf
                if (lvl != 0 && ni >= 0) {
                  const int *orr = wl + (int)((unsigned)w >> 26) * kMeshRayWords;
                  const float4 r0 = *reinterpret_cast<const float4 *>(orr + 0), r1 = *reinterpret_cast<const float4 *>(orr + 4);
                  BvhNode nd;
                  if (ni < lds_nodes) {
                    nd = s_nodes[ni];
                  } else {
                    nd = sc.nodes[ni];
                  }
#if defined(RTMI_ABLATE) && RTMI_ABLATE == 5
                  m = nd.mn[0] * r0.x + r1.x;  // (diagnostic: the loads without the arithmetic)
#else
                  m = aabb_crossing_time(nd, mk(r0.x, r0.y, r0.z), mk(r1.x, r1.y, r1.z));
#endif
                }
                times[t] = __float_as_int(m);
              }
            }
            wave_lds_fence();
            RTMI_STAT3(const unsigned long long tr3 = stat_now(); st.cyc[7] += tr3 - tr2;)
            // (c) the walk of bvh.cuh:123-158 over my leaves
            // (c) the walk of bvh.cuh:123-158 over my leaves, without branches: per leaf one bit per level,
            // left-aligned like the path code -- `valid` the levels of its path (a prefix), `pass` the levels whose
            // box is entered: as the previous leaf found it where the two paths coincide, by `crossing time <=
            // t_to` below.  The walk stops at the first level that is not entered; the leaf's faces count iff
            // there is none.
            if (__builtin_amdgcn_ballot_w64(now) != 0ull) {
#pragma unroll
              for (int k = 0; k < kHitSlots; k++) {
                if (__builtin_amdgcn_ballot_w64(now && k < nleaf) == 0ull) break;  // wave-uniform
                const bool mine = now && k < nleaf;
                const int *row = times + (mine ? (base - lo + k) << log_d : 0);
                uint32_t code = 0u, valid = 0u, below = 0u;
                for (int c = 0; c <= depth_r; c += 8) {  // wave-uniform trip count
                  const int4 ma = *reinterpret_cast<const int4 *>(row + c), mb = *reinterpret_cast<const int4 *>(row + c + 4);
                  const int mv[8] = {ma.x, ma.y, ma.z, ma.w, mb.x, mb.y, mb.z, mb.w};
                  if (c == 0) code = (uint32_t)ma.x;
#pragma unroll
                  for (int j = 0; j < 8; j++) {
                    const uint32_t bit = 0x80000000u >> ((c + j - 1) & 31);  // level c + j (>= 1)
                    const bool in_row = c + j >= 1 && c + j <= depth_r;
                    valid |= in_row && mv[j] != (int)0xffffffff ? bit : 0u;
#if defined(RTMI_ABLATE) && RTMI_ABLATE == 1
                    below |= in_row ? bit : 0u;
#else
                    below |= in_row && (T)__int_as_float(mv[j]) <= bt_to ? bit : 0u;  // (the marker is a NaN: false)
#endif
                  }
                }
                const int shared = have_prev ? __clz((int)(prev_code ^ code)) : 0;  // decisions in common
                const uint32_t common = shared ? 0xffffffffu << (32 - shared) : 0u;
                const uint32_t pass = (entered & common) | (below & ~common);
                const uint32_t fail = valid & ~pass;
                const int upto = fail ? __clz((int)fail) : __popc(valid);  // levels entered before the walk stopped
                const uint32_t walked = upto ? 0xffffffffu << (32 - upto) : 0u;
                const uint32_t bits = (entered & common) | (walked & ~common);
                if (mine) {
                  have_prev = true;
                  prev_code = code;
                  entered = bits;
                  const T tj = (T)__int_as_float(hs[k].z);
                  if (fail == 0u && tj <= bt_to) {
                    bt_to = tj;
                    bhit = true;
                    bface = hs[k].y;
                  }
                }
              }
              if (now) todo = false;
            }
            wave_lds_fence();
            RTMI_STAT3(tr1 = stat_now(); st.cyc[8] += tr1 - tr3;)
          }
#if defined(RTMI_ABLATE) && RTMI_ABLATE == 4
          if (nleaf > 0) {
#else
          if (nleaf > 0 && depth_r == 0) {  // the root is the only leaf: nothing to test (its box never is)
#endif
#pragma unroll
            for (int k = 0; k < kHitSlots; k++) {
              if (k < nleaf) {
                const T tj = (T)__int_as_float(hs[k].z);
                if (tj <= bt_to) bt_to = tj, bhit = true, bface = hs[k].y;
              }
            }
          }
          need = need && cut != kCodeNone;  // leaves were deferred: search again from `cut` on
          lo_code = cut;
          RTMI_STAT(st.cyc[3] += stat_now() - ts1;)
        }
        if (bhit && (F & F_TEX)) {
          // barycentrics of the winner (the same binary32 operations as in the search)
          const FaceRec f = sc.faces[bface];
          float t = 0.f;
          const V3 fe2 = mk(f.e2[0], f.e2[1], f.e2[2]);
          (void)tri_test_flat<T>(mk(f.p0[0], f.p0[1], f.p0[2]), mk(f.e1[0], f.e1[1], f.e1[2]), fe2, cross3(d, fe2), o, d,
                                 bt_to, t, fu, fv);
        }
        bool acc = bhit && (!ok || bt_to < t_to);
        ok = ok || acc;
        t_to = acc ? bt_to : t_to;
        win = acc ? make_id(RUN_BVH, bface) : win;
        aux = acc ? run.first + i : aux;
        bu = acc ? fu : bu;
        bv = acc ? fv : bv;
      }
    }
  }
  Hit h;
  h.ok = ok;
  h.t = (float)t_to;
  h.win = win;
  h.aux = aux;
  h.u = bu;
  h.v = bv;
  h.work = work;
  return h;
}
