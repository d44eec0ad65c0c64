// Host-side scene recorder + flatten (see scene.h).  No device code here; the
// file is a .hip unit only so it shares vec.h's host/device helpers and the
// -ffp-contract=off build flags with the kernels.
#include "scene.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>

namespace rtmi {

// utils.cu:54-55,79: v0v1, v0v2 and normalize(cross(v0v1, v0v2)) do not depend on the ray.
TriRec make_tri(V3 p0, V3 p1, V3 p2) {
  V3 e1 = p1 - p0, e2 = p2 - p0;
  V3 n = unit3(cross3(e1, e2));
  TriRec r;
  r.p0[0] = p0.x, r.p0[1] = p0.y, r.p0[2] = p0.z;
  r.e1[0] = e1.x, r.e1[1] = e1.y, r.e1[2] = e1.z;
  r.e2[0] = e2.x, r.e2[1] = e2.y, r.e2[2] = e2.z;
  r.n[0] = n.x, r.n[1] = n.y, r.n[2] = n.z;
  return r;
}

static HotTri hot_tri(V3 p0, V3 p1, V3 p2, int mat, int flags) {
  TriRec t = make_tri(p0, p1, p2);
  HotTri h{};
  for (int c = 0; c < 3; c++) h.p0[c] = t.p0[c], h.e1[c] = t.e1[c], h.e2[c] = t.e2[c], h.n[c] = t.n[c];
  h.mat = mat;
  h.flags = flags;
  return h;
}

static int face_slack_exponent(const V3 p[3]);

// The per-pair records of the culled list scan: corners and padded bounds (n = 3 or 4 points).
static void push_pair(Scene &s, const V3 *p, int n, int flags) {
  PairPts pp{};
  const V3 q[4] = {p[0], p[1], p[2], n == 4 ? p[3] : p[2]};
  for (int c = 0; c < 3; c++) {
    const float x[4] = {c == 0 ? q[0].x : c == 1 ? q[0].y : q[0].z, c == 0 ? q[1].x : c == 1 ? q[1].y : q[1].z,
                        c == 0 ? q[2].x : c == 1 ? q[2].y : q[2].z, c == 0 ? q[3].x : c == 1 ? q[3].y : q[3].z};
    pp.p0[c] = x[0], pp.p1[c] = x[1], pp.p2[c] = x[2], pp.p3[c] = x[3];
  }
  pp.flags = flags;
  PairBox bx{};
  float diag = 0.f, mag = 0.f;
  for (int c = 0; c < 3; c++) {
    const float x[4] = {pp.p0[c], pp.p1[c], pp.p2[c], pp.p3[c]};
    bx.mn[c] = fminf(fminf(x[0], x[1]), fminf(x[2], x[3]));
    bx.mx[c] = fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3]));
    diag = fmaxf(diag, bx.mx[c] - bx.mn[c]);
    mag = fmaxf(mag, fmaxf(fabsf(bx.mn[c]), fabsf(bx.mx[c])));
  }
  // the padding of the mesh search boxes (padded_node_bounds): what the binary32 triangle test can accept
  // beyond the exact triangle at short range; the distance-proportional part is added at query time
  const float pad = fixed_pad(diag, mag);
  for (int c = 0; c < 3; c++) bx.mn[c] -= pad, bx.mx[c] += pad;
  // A thin triangle (face_slack_exponent: below 1.8 degrees) is accepted by the binary32 test from further off than the
  // scan's distance slack covers, by a factor that has no bound in the list's terms: its pair is not culled at all
  // -- unbounded bounds make it a candidate of every ray (the slab test's products stay +-infinity: the ray's
  // reciprocals are never zero), and its tests decide as in the full scan.
  const V3 second[3] = {q[1], q[2], q[3]};
  if (face_slack_exponent(q) > 0 || (n == 4 && face_slack_exponent(second) > 0))
    for (int c = 0; c < 3; c++) bx.mn[c] = -INFINITY, bx.mx[c] = INFINITY;
  s.list_mag = fmaxf(s.list_mag, mag);
  s.pair_pts.push_back(pp);
  s.pair_boxes.push_back(bx);
}

// parallelogram.cu:10-15 + the two triangles of parallelogram.cu:25,33
static void push_pgram(Scene &s, V3 p0, V3 p1, V3 p2, int mat) {
  std::vector<HotTri> &out = s.tris;
  V3 p3 = p1 + p2 - p0;
  const HotTri first = hot_tri(p0, p1, p2, mat, TRI_PGRAM);
  HotTri second = hot_tri(p1, p2, p3, mat, TRI_PGRAM | TRI_SECOND);
  if (memcmp(first.e2, second.e2, sizeof(first.e2)) == 0) second.flags |= TRI_SAME_E2;  // p3 - p1 == p2 - p0 exactly
  out.push_back(first);
  out.push_back(second);
  const V3 q[4] = {p0, p1, p2, p3};
  push_pair(s, q, 4, PAIR_SECOND | ((second.flags & TRI_SAME_E2) ? PAIR_SAME_E2 : 0));
}

// parallelepiped.cu:8-18: derive the four opposite corners.
void box_from_points(const V3 p[4], V3 out[8]) {
  auto fourth = [](V3 a, V3 b, V3 c) { return c + b - a; };
  V3 q[4];
  q[3] = fourth(p[0], p[1], p[2]);
  q[2] = fourth(p[0], p[1], p[3]);
  q[1] = fourth(p[0], p[2], p[3]);
  q[0] = fourth(p[1], q[2], q[3]);
  for (int i = 0; i < 4; i++) out[i] = p[i], out[4 + i] = q[i];
}

// parallelepiped.cu:25-32: AddCorner(p) then AddCorner(q), faces (c0,c1,c2), (c0,c2,c3), (c0,c3,c1)
void box_faces(const V3 corners[8], V3 faces[18]) {
  int k = 0;
  for (int set = 0; set < 2; set++) {
    const V3 *c = corners + set * 4;
    for (int i = 1; i <= 3; i++) {
      int x = i, y = (i + 1 == 4) ? 1 : x + 1;
      faces[k++] = c[0];
      faces[k++] = c[x];
      faces[k++] = c[y];
    }
  }
}

// camera.cu:24-38
void camera_pinhole(Scene &s, V3 pos, V3 look_at, V3 up, double fov, double aspect) {
  CameraDev &c = s.cam;
  c.defocus = 0;
  c.lens_radius = -1.f;
  c.position = pos;
  V3 w = unit3(pos - look_at);
  c.u = unit3(cross3(up, w));
  c.v = unit3(cross3(w, c.u));
  double half_height = std::tan(fov / 2);
  double half_width = aspect * half_height;
  c.horizontal = c.u * static_cast<float>(2 * half_width);
  c.vertical = c.v * static_cast<float>(2 * half_height);
  c.llc = pos - w - c.u * static_cast<float>(half_width) - c.v * static_cast<float>(half_height);
  s.cam_w = w;
  s.has_camera = true;
}

// camera.cu:6-22
void camera_defocus(Scene &s, V3 pos, V3 look_at, V3 up, double fov, double aspect, double aperture, double focus) {
  CameraDev &c = s.cam;
  c.defocus = 1;
  c.position = pos;
  V3 w = unit3(pos - look_at);
  c.u = unit3(cross3(up, w));
  c.v = unit3(cross3(w, c.u));
  double half_height = focus * std::tan(fov / 2);
  double half_width = aspect * half_height;
  c.horizontal = c.u * static_cast<float>(2 * half_width);
  c.vertical = c.v * static_cast<float>(2 * half_height);
  c.llc = pos - w - c.u * static_cast<float>(half_width) - c.v * static_cast<float>(half_height);
  c.lens_radius = (float)(aperture / 2);  // DiskRand(float radius) narrows lens_radius_ (camera.cu:64,74)
  s.cam_w = w;
  s.has_camera = true;
}

// camera.cu:40-47
void camera_raw(Scene &s, V3 pos, V3 llc, V3 horiz, V3 vert) {
  CameraDev &c = s.cam;
  c.defocus = 0;
  c.lens_radius = -1.f;
  c.position = pos;
  c.llc = llc;
  c.horizontal = horiz;
  c.vertical = vert;
  c.u = splat(0.f);
  c.v = splat(0.f);
  s.cam_w = splat(0.f);
  s.has_camera = true;
}

struct FacePts {
  V3 p[3];
  float uv[6];
  int orig;
  uint32_t code;
  int leaf;  // ordinal of the reference leaf in visiting order
  int kexp;  // slack exponent of the face (face_slack_exponent): the search boxes above it are widened by 2^kexp
};

// How far from a face a ray can pass and still be accepted by the binary32 Moller-Trumbore test (utils.cu:49-85):
// the error of dot(tvec, pvec) / det moves the computed (u, v) by about eps |o - p0| / (|e| sin(theta) cos(phi)), i.e.
// the accepted region reaches about eps |o - p0| / sin(theta) beyond the face (theta: its smallest angle; measured:
// at most 2.1 eps D / sin(theta), tools/exp/false_accept_reach.py).  The search boxes are widened at query time by
// 2^-16 of (|o| + the mesh's largest coordinate) -- enough down to sin(theta) = 1/32 (1.8 degrees); a thinner face asks
// for 8 eps / sin(theta), rounded up to the next power of two: that exponent.  Every node of the search tree carries
// the largest exponent of the faces below it (QNode4::slack_exp), so a sliver widens the boxes on ITS root-to-leaf
// path only.  A face whose edges span less than 1e-7 can never pass the test's |det| >= 1e-7 and asks for nothing.
static int face_slack_exponent(const V3 p[3]) {
  const double P[3][3] = {{p[0].x, p[0].y, p[0].z}, {p[1].x, p[1].y, p[1].z}, {p[2].x, p[2].y, p[2].z}};
  double min_sin = 1.0, cross_len = 0.0;
  for (int a = 0; a < 3; a++) {
    const double *A = P[a], *B = P[(a + 1) % 3], *Cc = P[(a + 2) % 3];
    const double u[3] = {B[0] - A[0], B[1] - A[1], B[2] - A[2]}, v[3] = {Cc[0] - A[0], Cc[1] - A[1], Cc[2] - A[2]};
    const double cx = u[1] * v[2] - u[2] * v[1], cy = u[2] * v[0] - u[0] * v[2], cz = u[0] * v[1] - u[1] * v[0];
    const double lu = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]), lv = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    cross_len = std::sqrt(cx * cx + cy * cy + cz * cz);  // (twice the area: the same for every corner)
    // (an obtuse corner has a small sine too, but then one of the other two is acute and smaller)
    min_sin = std::min(min_sin, lu > 0 && lv > 0 ? cross_len / (lu * lv) : 0.0);
  }
  if (!(cross_len >= 0.9e-7)) return 0;   // |det| <= |e1 x e2| for a unit direction: never accepted (NaNs: neither)
  if (!(min_sin < (double)kThinSine)) return 0;
  const int k = (int)std::ceil(std::log2((double)kThinSine / std::max(min_sin, 1e-12)));
  return k < 0 ? 0 : k > kSlackExpMax ? kSlackExpMax : k;
}

// ---- the search tree (DESIGN.md "Mesh queries"): binned-SAH binary tree over the faces, leaves
// of <= 4 faces, collapsed to four children per node and quantised into QNode4 records.  Which
// leaf of the REFERENCE's tree a face sits in is carried by the face (FaceRec::code), not by this
// tree, so the splits are free to follow the geometry.
struct BinNode {
  float mn[3], mx[3];  // exact bounds of the faces below
  int left = -1, right = -1;
  int first = 0, n = 0;
  int kexp = 0;  // largest FacePts::kexp below
};

static inline float face_centroid(const FacePts &f, int a) {
  const float c[3] = {f.p[0].x + f.p[1].x + f.p[2].x, f.p[0].y + f.p[1].y + f.p[2].y, f.p[0].z + f.p[1].z + f.p[2].z};
  return c[a];
}
static inline double half_area(const float mn[3], const float mx[3]) {
  const double dx = (double)mx[0] - mn[0], dy = (double)mx[1] - mn[1], dz = (double)mx[2] - mn[2];
  return dx * dy + dy * dz + dz * dx;
}

// Faces [first, first+n) are reordered in place.  Binned surface-area heuristic (16 bins per axis
// on the centroid bounds); a median split along the longest centroid axis when no bin boundary
// separates the faces, and below `balanced_below` levels (which bounds the depth on adversarial
// inputs: coincident or geometrically nested faces).
static int build_bin(std::vector<BinNode> &bn, std::vector<FacePts> &fp, int first, int n, int level) {
  BinNode nd;
  float cmn[3], cmx[3];
  for (int k = 0; k < 3; k++) nd.mn[k] = cmn[k] = INFINITY, nd.mx[k] = cmx[k] = -INFINITY;
  for (int i = 0; i < n; i++) {
    const FacePts &f = fp[first + i];
    for (int j = 0; j < 3; j++) {
      const float q[3] = {f.p[j].x, f.p[j].y, f.p[j].z};
      for (int k = 0; k < 3; k++) nd.mn[k] = fminf(nd.mn[k], q[k]), nd.mx[k] = fmaxf(nd.mx[k], q[k]);
    }
    for (int k = 0; k < 3; k++) {
      const float c = face_centroid(f, k);
      cmn[k] = fminf(cmn[k], c), cmx[k] = fmaxf(cmx[k], c);
    }
    nd.kexp = std::max(nd.kexp, f.kexp);
  }
  nd.first = first, nd.n = n;
  const int me = (int)bn.size();
  bn.push_back(nd);
  if (n <= 4) return me;
  constexpr int NB = 16;
  constexpr int balanced_below = 40;
  int mid = -1;
  if (level < balanced_below) {
    double best = INFINITY;
    int best_axis = -1, best_bin = -1;
    for (int a = 0; a < 3; a++) {
      const float lo = cmn[a], hi = cmx[a];
      if (!(hi > lo)) continue;
      const float scale = (float)NB / (hi - lo);
      float bmn[NB][3], bmx[NB][3];
      int cnt[NB];
      for (int b = 0; b < NB; b++) {
        cnt[b] = 0;
        for (int k = 0; k < 3; k++) bmn[b][k] = INFINITY, bmx[b][k] = -INFINITY;
      }
      for (int i = 0; i < n; i++) {
        const FacePts &f = fp[first + i];
        int b = (int)((face_centroid(f, a) - lo) * scale);
        b = b < 0 ? 0 : b > NB - 1 ? NB - 1 : b;
        cnt[b]++;
        for (int j = 0; j < 3; j++) {
          const float q[3] = {f.p[j].x, f.p[j].y, f.p[j].z};
          for (int k = 0; k < 3; k++) bmn[b][k] = fminf(bmn[b][k], q[k]), bmx[b][k] = fmaxf(bmx[b][k], q[k]);
        }
      }
      double la[NB], ra[NB];
      int lc[NB], rc[NB];
      float amn[3] = {INFINITY, INFINITY, INFINITY}, amx[3] = {-INFINITY, -INFINITY, -INFINITY};
      int c = 0;
      for (int b = 0; b < NB; b++) {
        for (int k = 0; k < 3; k++) amn[k] = fminf(amn[k], bmn[b][k]), amx[k] = fmaxf(amx[k], bmx[b][k]);
        c += cnt[b];
        lc[b] = c, la[b] = c ? half_area(amn, amx) : 0.0;
      }
      for (int k = 0; k < 3; k++) amn[k] = INFINITY, amx[k] = -INFINITY;
      c = 0;
      for (int b = NB - 1; b >= 0; b--) {
        for (int k = 0; k < 3; k++) amn[k] = fminf(amn[k], bmn[b][k]), amx[k] = fmaxf(amx[k], bmx[b][k]);
        c += cnt[b];
        rc[b] = c, ra[b] = c ? half_area(amn, amx) : 0.0;
      }
      for (int b = 0; b + 1 < NB; b++) {
        if (!lc[b] || !rc[b + 1]) continue;
        const double cost = la[b] * lc[b] + ra[b + 1] * rc[b + 1];
        if (cost < best) best = cost, best_axis = a, best_bin = b;
      }
    }
    if (best_axis >= 0) {
      const float lo = cmn[best_axis], scale = (float)NB / (cmx[best_axis] - cmn[best_axis]);
      auto it = std::stable_partition(fp.begin() + first, fp.begin() + first + n, [&](const FacePts &f) {
        int b = (int)((face_centroid(f, best_axis) - lo) * scale);
        b = b < 0 ? 0 : b > NB - 1 ? NB - 1 : b;
        return b <= best_bin;
      });
      mid = (int)(it - (fp.begin() + first));
      if (mid <= 0 || mid >= n) mid = -1;
    }
  }
  if (mid < 0) {
    int axis = 0;
    for (int k = 1; k < 3; k++)
      if (cmx[k] - cmn[k] > cmx[axis] - cmn[axis]) axis = k;
    mid = n / 2;
    std::nth_element(fp.begin() + first, fp.begin() + first + mid, fp.begin() + first + n,
                     [&](const FacePts &a, const FacePts &b) { return face_centroid(a, axis) < face_centroid(b, axis); });
  }
  const int l = build_bin(bn, fp, first, mid, level + 1);
  const int r = build_bin(bn, fp, first + mid, n - mid, level + 1);
  bn[me].left = l, bn[me].right = r;
  return me;
}

// Bounds of a binary node, padded: no face the binary32 Moller-Trumbore test can accept is ever
// culled by the (also slack) slab test of the kernel.
static void padded_node_bounds(const BinNode &b, float mn[3], float mx[3]) {
  float diag = 0.f, mag = 0.f;
  for (int k = 0; k < 3; k++) {
    diag = fmaxf(diag, b.mx[k] - b.mn[k]);
    mag = fmaxf(mag, fmaxf(fabsf(b.mn[k]), fabsf(b.mx[k])));
  }
  const float pad = fixed_pad(diag, mag);
  for (int k = 0; k < 3; k++) mn[k] = b.mn[k] - pad, mx[k] = b.mx[k] + pad;
}

// One QNode4 from up to four binary nodes: expand the child with the largest surface area until
// four children (or only leaves) remain; quantise their padded boxes outward onto the node's grid.
static int collapse4(std::vector<QNode4> &qn, const std::vector<BinNode> &bn, int root, int *depth) {
  const int me = (int)qn.size();
  qn.emplace_back();
  int kids[4], nk = 0;
  if (bn[root].left < 0) {
    kids[nk++] = root;
  } else {
    kids[nk++] = bn[root].left, kids[nk++] = bn[root].right;
    while (nk < 4) {
      int best = -1;
      double ba = -1.0;
      for (int i = 0; i < nk; i++)
        if (bn[kids[i]].left >= 0) {
          const double ar = half_area(bn[kids[i]].mn, bn[kids[i]].mx);
          if (ar > ba) ba = ar, best = i;
        }
      if (best < 0) break;
      const int k = kids[best];
      kids[best] = bn[k].left;
      kids[nk++] = bn[k].right;
    }
  }
  float cmn[4][3], cmx[4][3];
  for (int c = 0; c < nk; c++) padded_node_bounds(bn[kids[c]], cmn[c], cmx[c]);
  QNode4 nd;
  memset(&nd, 0, sizeof(nd));
  for (int a = 0; a < 3; a++) {
    float lo = INFINITY, hi = -INFINITY;
    for (int c = 0; c < nk; c++) lo = fminf(lo, cmn[c][a]), hi = fmaxf(hi, cmx[c][a]);
    nd.origin[a] = lo;
    const double extent = (double)hi - (double)lo;
    int e = -60;
    if (extent > 0) e = (int)std::ceil(std::log2(extent / 255.0));
    e = e < -60 ? -60 : e > 100 ? 100 : e;
    for (; e < 127; e++) {  // the far corner must land on the grid (finite bounds: rtmi_add_bvh refuses others,
                            // and the loop is bounded in any case: an int8 exponent is all a node can hold)
      bool fits = true;
      for (int c = 0; c < nk; c++)
        if (!(std::ceil(std::ldexp((double)cmx[c][a] - (double)lo, -e)) <= 255.0)) fits = false;
      if (fits) break;
    }
    nd.exp[a] = (int8_t)e;
    for (int c = 0; c < 4; c++) {
      if (c < nk) {
        double ql = std::floor(std::ldexp((double)cmn[c][a] - (double)lo, -e));
        double qh = std::ceil(std::ldexp((double)cmx[c][a] - (double)lo, -e));
        ql = ql < 0 ? 0 : ql > 255 ? 255 : ql;
        qh = qh < 0 ? 0 : qh > 255 ? 255 : qh;
        nd.qlo[a][c] = (uint8_t)ql, nd.qhi[a][c] = (uint8_t)qh;
      } else {
        nd.qlo[a][c] = 255, nd.qhi[a][c] = 0;
      }
    }
  }
  int deepest = 0;
  for (int c = 0; c < 4; c++) {
    int child = -1;  // count 0: never visited
    if (c < nk) {
      const BinNode &k = bn[kids[c]];
      if (k.left < 0) {
        child = -(k.first * 8 + k.n) - 1;
      } else {
        int d = 0;
        child = collapse4(qn, bn, kids[c], &d);
        deepest = std::max(deepest, d);
      }
    }
    nd.child[c] = child;
  }
  nd.slack_exp = (int8_t)bn[root].kexp;  // (its children's boxes are widened by this: the largest need below any of them)
  qn[me] = nd;
  *depth = deepest + 1;
  return me;
}

// 4-wide search tree over the faces [first, first+n) of a mesh; faces are reordered inside the
// range (each keeps `orig` and `code`).  Returns the root's index in `qn`; *depth = levels.
static int build_subtree(std::vector<QNode4> &qn, std::vector<FacePts> &fp, int first, int n, int *depth) {
  std::vector<BinNode> bn;
  bn.reserve((size_t)n);
  const int root = build_bin(bn, fp, first, n, 0);
  return collapse4(qn, bn, root, depth);
}

// The top of a mesh's search tree as a flat table of kTopEntries sub-trees (mesh_search.h:
// a wave with few rays tests all of them at once instead of descending level by level).  Starting
// from the root's children, the sub-tree with the largest surface is replaced by its children while
// the table has room.  Boxes: the children's quantised boxes, rounded outward to binary32.
static void build_top_entries(const std::vector<QNode4> &qn, int sub_root, std::vector<BvhNode> &tops) {
  struct Entry {
    float mn[3], mx[3];
    int ref;
    int kexp;  // slack exponent of the node the box was a child of
  };
  std::vector<Entry> fr;
  auto expand = [&](int node) {
    const QNode4 &nd = qn[(size_t)node];
    for (int c = 0; c < 4; c++) {
      if (nd.child[c] == -1) continue;
      Entry e;
      for (int a = 0; a < 3; a++) {
        const double lo = (double)nd.origin[a] + std::ldexp((double)nd.qlo[a][c], nd.exp[a]);
        const double hi = (double)nd.origin[a] + std::ldexp((double)nd.qhi[a][c], nd.exp[a]);
        e.mn[a] = nextafterf(nextafterf((float)lo, -INFINITY), -INFINITY);
        e.mx[a] = nextafterf(nextafterf((float)hi, INFINITY), INFINITY);
      }
      e.ref = nd.child[c];
      e.kexp = nd.slack_exp;
      fr.push_back(e);
    }
  };
  if (sub_root >= 0) expand(sub_root);
  for (;;) {
    int best = -1;
    double ba = -1.0;
    for (size_t i = 0; i < fr.size(); i++)
      if (fr[i].ref >= 0) {
        const double ar = half_area(fr[i].mn, fr[i].mx);
        if (ar > ba) ba = ar, best = (int)i;
      }
    if (best < 0 || fr.size() + 3 > (size_t)kTopEntries) break;
    const int node = fr[(size_t)best].ref;
    fr.erase(fr.begin() + best);
    expand(node);
  }
  for (int i = 0; i < kTopEntries; i++) {
    BvhNode t;
    if (i < (int)fr.size()) {
      for (int a = 0; a < 3; a++) t.mn[a] = fr[(size_t)i].mn[a], t.mx[a] = fr[(size_t)i].mx[a];
      t.left = fr[(size_t)i].ref;
      t.right = fr[(size_t)i].kexp;
    } else {
      for (int a = 0; a < 3; a++) t.mn[a] = 1.f, t.mx[a] = -1.f;
      t.left = -1;  // unused slot
      t.right = 0;
    }
    tops.push_back(t);
  }
}

// ---- the error budget, record by record (margins.h).  Every structure a cull decides from is checked at commit
// against what it stands for; a violation fails the commit (RTMI_ERR_INVALID "margin budget") instead of rendering a
// frame that may have lost rays.
// A search tree (local indices: children >= 0 are nodes of `qn`, < -1 face blocks of `fp`): the box a node holds for a
// child, decoded from its 8-bit grid coordinates, must contain the fixed-padded exact bounds of every face below that
// child, and the node's slack exponent must be at least that of every such face.
static std::string check_search_tree(const std::vector<QNode4> &qn, const std::vector<FacePts> &fp, int node, float mn[3],
                                     float mx[3], int *kexp) {
  const QNode4 &nd = qn[(size_t)node];
  for (int a = 0; a < 3; a++) mn[a] = INFINITY, mx[a] = -INFINITY;
  *kexp = 0;
  for (int c = 0; c < 4; c++) {
    if (nd.child[c] == -1) continue;
    float cmn[3] = {INFINITY, INFINITY, INFINITY}, cmx[3] = {-INFINITY, -INFINITY, -INFINITY};
    int ck = 0;
    if (nd.child[c] >= 0) {
      const std::string e = check_search_tree(qn, fp, nd.child[c], cmn, cmx, &ck);
      if (!e.empty()) return e;
    } else {
      const int enc = -(nd.child[c] + 1), cnt = enc & 7, first = enc >> 3;
      for (int i = first; i < first + cnt; i++) {
        ck = std::max(ck, fp[(size_t)i].kexp);
        for (int j = 0; j < 3; j++) {
          const float q[3] = {fp[(size_t)i].p[j].x, fp[(size_t)i].p[j].y, fp[(size_t)i].p[j].z};
          for (int a = 0; a < 3; a++) cmn[a] = fminf(cmn[a], q[a]), cmx[a] = fmaxf(cmx[a], q[a]);
        }
      }
    }
    float diag = 0.f, mag = 0.f;
    for (int a = 0; a < 3; a++) diag = fmaxf(diag, cmx[a] - cmn[a]), mag = fmaxf(mag, fmaxf(fabsf(cmn[a]), fabsf(cmx[a])));
    const float pad = fixed_pad(diag, mag);
    for (int a = 0; a < 3; a++) {
      const double lo = (double)nd.origin[a] + std::ldexp((double)nd.qlo[a][c], nd.exp[a]);
      const double hi = (double)nd.origin[a] + std::ldexp((double)nd.qhi[a][c], nd.exp[a]);
      // (the pad is applied in binary32 where the bounds were built: its rounding is up to 2^-24 of the coordinate, 0.6 % of the smallest pad)
      if (!(lo <= (double)cmn[a] - 0.98 * (double)pad && hi >= (double)cmx[a] + 0.98 * (double)pad))
        return "margin budget: a search node's child box does not contain its faces' padded bounds";
    }
    if (nd.slack_exp < ck) return "margin budget: a search node's slack exponent is below a face's under it";
    for (int a = 0; a < 3; a++) mn[a] = fminf(mn[a], cmn[a]), mx[a] = fmaxf(mx[a], cmx[a]);
    *kexp = std::max(*kexp, ck);
  }
  return "";
}

// bvh.cuh:113-121: bounds; leaf if n <= kMin; else sort by positions_[0].x and
// split at mid = (n-1)/2.  The sort key never changes down the tree, so the
// reference's per-node re-sort of an already sorted sub-range is the identity for
// a stable sort; one stable sort of the whole range reproduces it.  (thrust::sort
// does not promise an order for equal keys; this build fixes it as stable.)
// Every face of a leaf is stamped with the leaf's path code (FaceRec::code).
// `path`: the nodes below the root on the way to this one; every leaf appends its path to `leaf_paths`
// (one vector per leaf, in visiting order) and stamps its faces with its ordinal there.
static int build_bvh_nodes(std::vector<BvhNode> &nodes, std::vector<FacePts> &fp, int first, int n, int leaf_max,
                           int level, uint32_t code, int *ref_depth, std::vector<int> &path,
                           std::vector<std::vector<int>> &leaf_paths) {
  if (level > *ref_depth) *ref_depth = level;
  BvhNode nd;
  for (int k = 0; k < 3; k++) nd.mn[k] = INFINITY, nd.mx[k] = -INFINITY;
  for (int i = 0; i < n; i++)  // bvh.cuh:71-82
    for (int j = 0; j < 3; j++) {
      const V3 &p = fp[first + i].p[j];
      const float c[3] = {p.x, p.y, p.z};
      for (int k = 0; k < 3; k++) {
        nd.mn[k] = fminf(c[k], nd.mn[k]);
        nd.mx[k] = fmaxf(c[k], nd.mx[k]);
      }
    }
  int me = (int)nodes.size();
  nodes.push_back(nd);
  if (level > 0) path.push_back(me);
  if (n <= leaf_max) {
    nodes[me].left = -1;
    nodes[me].right = -n;
    for (int i = 0; i < n; i++) fp[first + i].code = code, fp[first + i].leaf = (int)leaf_paths.size();
    leaf_paths.push_back(path);
  } else {
    int mid = (n - 1) / 2;
    const uint32_t bit = level < 32 ? 0x80000000u >> level : 0u;
    int l = build_bvh_nodes(nodes, fp, first, mid + 1, leaf_max, level + 1, code, ref_depth, path, leaf_paths);
    int r = build_bvh_nodes(nodes, fp, first + mid + 1, n - mid - 1, leaf_max, level + 1, code | bit, ref_depth, path,
                            leaf_paths);
    nodes[me].left = l;
    nodes[me].right = r;
  }
  if (level > 0) path.pop_back();
  return me;
}

std::string Scene::flatten() {
  pair_boxes.clear(), pair_pts.clear(), tri_nrm.clear(), list_mag = 0.f;
  sph_groups.clear(), sph_members.clear(), sph_mag = 0.f;
  sliver_faces = 0;
  runs.clear(), spheres.clear(), tris.clear(), bvh_recs.clear(), face_of_orig.clear(), nodes.clear(), qnodes.clear(), faces.clear(), leaf_paths.clear(), tops.clear(),
      face_uv.clear(), mat_recs.clear(), tex_recs.clear();
  features = 0;
  n_pgrams = n_triangles = n_spheres = 0;
  sub_depth = 0;
  if (!has_camera) return "scene has no camera";
  if (list_counts.empty() || list_counts[0] > 1024) return "world exceeds HitableList::kMaxHitables (1024)";
  if (cam.defocus) features |= F_DEFOCUS;

  // ---- materials: fold constant textures into the record; image textures by index
  std::vector<int> tex_to_image(texs.size(), -1);
  int n_images = 0;
  for (size_t i = 0; i < texs.size(); i++)
    if (texs[i].image) tex_to_image[i] = n_images++;
  for (const HostMat &m : mats) {
    MatRec r{};
    r.kind = m.kind;
    r.r = m.rgb.x, r.g = m.rgb.y, r.b = m.rgb.z;
    r.param = m.param;
    r.tex = -1;
    if (m.tex >= 0) {
      if (m.tex >= (int)texs.size()) return "material refers to an unknown texture";
      const HostTex &t = texs[m.tex];
      if (t.image) {
        r.tex = tex_to_image[m.tex];
        features |= F_TEX;
      } else {
        r.r = t.rgb.x, r.g = t.rgb.y, r.b = t.rgb.z;
      }
    }
    mat_recs.push_back(r);
  }

  // Run::count is in records for spheres / BVHs and in record PAIRS for RUN_TRIS.
  auto push_run = [&](int kind, int first) {
    const int stride = kind == RUN_TRIS ? 2 : 1;
    if (!runs.empty() && runs.back().kind == kind && runs.back().first + stride * runs.back().count == first) {
      runs.back().count++;
    } else {
      runs.push_back(Run{kind, first, 1, -1});  // pad: first SphGroup of a grouped sphere run, else -1
    }
  };
  auto check_mat = [&](int m) { return m >= 0 && m < (int)mats.size(); };

  bytes_per_ray = 32;  // material / hit-record share per query
  for (const HostObj &ob : world) {
    switch (ob.kind) {
      case OBJ_SKY:
        runs.push_back(Run{RUN_SKY, 0, 1, 0});
        break;
      case OBJ_SPHERE: {
        if (!check_mat(ob.mat)) return "sphere without a valid material";
        SphereRec r;
        r.cx = ob.p[0].x, r.cy = ob.p[0].y, r.cz = ob.p[0].z;
        r.mat = ob.mat;
        r.radius = ob.radius;
        r.r2 = ob.radius * ob.radius;
        spheres.push_back(r);
        push_run(RUN_SPHERE, (int)spheres.size() - 1);
        features |= F_SPHERE;
        bytes_per_ray += 28;
        break;
      }
      case OBJ_TRI: {
        if (!check_mat(ob.mat)) return "triangle without a valid material";
        tris.push_back(hot_tri(ob.p[0], ob.p[1], ob.p[2], ob.mat, 0));
        tris.push_back(HotTri{});  // inert second record: the world-list loop walks pairs
        push_pair(*this, ob.p, 3, 0);
        push_run(RUN_TRIS, (int)tris.size() - 2);
        features |= F_TRIS;
        n_triangles++;
        bytes_per_ray += 40;
        break;
      }
      case OBJ_PGRAM: {
        if (!check_mat(ob.mat)) return "parallelogram without a valid material";
        push_pgram(*this, ob.p[0], ob.p[1], ob.p[2], ob.mat);
        push_run(RUN_TRIS, (int)tris.size() - 2);
        features |= F_TRIS;
        n_pgrams++;
        bytes_per_ray += 40;
        break;
      }
      case OBJ_BOX: {
        if (!check_mat(ob.mat)) return "parallelepiped without a valid material";
        for (int fidx = 0; fidx < 6; fidx++) {  // the six faces in AddCorner order
          push_pgram(*this, ob.p[fidx * 3], ob.p[fidx * 3 + 1], ob.p[fidx * 3 + 2], ob.mat);
          push_run(RUN_TRIS, (int)tris.size() - 2);
          n_pgrams++;
          bytes_per_ray += 40;
        }
        features |= F_TRIS;
        break;
      }
      case OBJ_BVH: {
        const HostBvh &hb = bvhs[ob.bvh];
        if (hb.mat >= (int)mats.size()) return "bvh refers to an unknown material";
        if (hb.mat < 0) return "bvh without a material (material_ptr_ == nullptr) is not renderable";
        const bool has_uv = !hb.uvs.empty();
        std::vector<FacePts> fp((size_t)hb.n);
        for (int i = 0; i < hb.n; i++) {
          for (int j = 0; j < 3; j++)
            fp[i].p[j] = mk(hb.faces[(size_t)i * 9 + j * 3], hb.faces[(size_t)i * 9 + j * 3 + 1],
                            hb.faces[(size_t)i * 9 + j * 3 + 2]);
          for (int j = 0; j < 6; j++) fp[i].uv[j] = has_uv ? hb.uvs[(size_t)i * 6 + j] : 0.f;
          fp[i].kexp = face_slack_exponent(fp[i].p);
          if (fp[i].kexp > 0) sliver_faces++;  // thinner than 1.8 degrees: its nodes' boxes are widened for it
        }
        int leaf_max = hb.leaf_max > 0 ? hb.leaf_max : 2048;
        if (hb.n > leaf_max)
          std::stable_sort(fp.begin(), fp.end(),
                           [](const FacePts &a, const FacePts &b) { return a.p[0].x < b.p[0].x; });
        for (int i = 0; i < hb.n; i++) fp[i].orig = i;  // index in the reference's face order
        const int face_base = (int)faces.size();
        const int sub_base = (int)qnodes.size();
        BvhRec br{};
        std::vector<BvhNode> local_nodes;
        std::vector<QNode4> local_sub;
        int depth = 0;
        // an empty mesh is a leaf that can never report a hit: it contributes nothing
        int ref_depth = 0;
        std::vector<int> path_now;
        std::vector<std::vector<int>> local_paths;
        br.root = hb.n > 0 ? build_bvh_nodes(local_nodes, fp, 0, hb.n, leaf_max, 0, 0u, &ref_depth, path_now, local_paths) : -1;
        if (ref_depth > kRefDepthMax) return "mesh tree too deep for the kernel's leaf path codes";
        br.sub_root = hb.n > 0 ? build_subtree(local_sub, fp, 0, hb.n, &depth) + sub_base : -1;
        sub_depth = std::max(sub_depth, depth);
        if (depth > kSubDepthMax) return "mesh too deep for the kernel's search stack";
        if (hb.n > 0) {  // the error budget of this mesh's search structure (margins.h)
          float tmn[3], tmx[3];
          int tk = 0;
          const std::string e = check_search_tree(local_sub, fp, 0, tmn, tmx, &tk);
          if (!e.empty()) return e;
        }
        if ((int64_t)face_base + hb.n + 4 >= kMeshMaxFaces || (int64_t)sub_base + (int64_t)local_sub.size() >= kMeshMaxNodes)
          return "mesh too large for the kernel's 26-bit search-stack entries";
        const int node_base = (int)nodes.size();
        for (BvhNode nd : local_nodes) {  // rebase indices into the scene-wide arrays
          if (nd.right >= 0) nd.left += node_base, nd.right += node_base;
          nodes.push_back(nd);
        }
        for (QNode4 nd : local_sub) {
          for (int c = 0; c < 4; c++) {
            if (nd.child[c] >= 0) {
              nd.child[c] += sub_base;
            } else {
              const int enc = -(nd.child[c] + 1), cnt = enc & 7, first = enc >> 3;
              nd.child[c] = cnt ? -((first + face_base) * 8 + cnt) - 1 : -1;
            }
          }
          qnodes.push_back(nd);
        }
        if (br.root >= 0) br.root += node_base;
        // the leaves' root-to-leaf paths as a table: row = leaf ordinal; the leaf's path code, then ref_depth
        // columns (the node at level 1, 2, ...), -1 below the leaf (closest_hit.h, replay)
        br.path_base = (int)leaf_paths.size();
        if ((int64_t)leaf_paths.size() + (int64_t)local_paths.size() * (ref_depth + 1) > (int64_t)kLeafPathMax)
          return "mesh leaf-path table too large";
        {
          std::vector<uint32_t> leaf_code(local_paths.size(), 0u);
          for (int i = 0; i < hb.n; i++) leaf_code[(size_t)fp[i].leaf] = fp[i].code;
          for (size_t li = 0; li < local_paths.size(); li++) {
            const std::vector<int> &lp = local_paths[li];
            leaf_paths.push_back((int32_t)leaf_code[li]);
            for (int lvl = 0; lvl < ref_depth; lvl++)
              leaf_paths.push_back(lvl < (int)lp.size() ? lp[(size_t)lvl] + node_base : -1);
          }
        }
        if (!face_uv.empty() || has_uv) face_uv.resize((size_t)face_base * 6, 0.f);
        if (has_uv) face_uv.resize((size_t)(face_base + hb.n) * 6, 0.f);
        face_of_orig.resize((size_t)face_base + hb.n, 0);
        for (int i = 0; i < hb.n; i++) face_of_orig[(size_t)face_base + fp[i].orig] = face_base + i;
        br.n_faces = hb.n;
        br.slack_exp = br.sub_root >= 0 ? local_sub[0].slack_exp : 0;  // (the search tree's root is its first node)
        for (int i = 0; i < hb.n; i++) {  // physical order = search-tree order
          TriRec t = make_tri(fp[i].p[0], fp[i].p[1], fp[i].p[2]);
          FaceRec f{};
          for (int c = 0; c < 3; c++) f.p0[c] = t.p0[c], f.e1[c] = t.e1[c], f.e2[c] = t.e2[c];
          f.orig = fp[i].orig;
          f.code = fp[i].code;
          f.leaf = fp[i].leaf;
          faces.push_back(f);
          if (has_uv)  // texture coordinates stay in the reference's order, addressed by `orig`
            for (int j = 0; j < 6; j++) face_uv[(size_t)(face_base + fp[i].orig) * 6 + j] = fp[i].uv[j];
        }
        br.mat = hb.mat;
        br.ref_depth = ref_depth;
        br.mag = 0.f;
        if (!local_nodes.empty())
          for (int k = 0; k < 3; k++)
            br.mag = fmaxf(br.mag, fmaxf(fabsf(local_nodes[0].mn[k]), fabsf(local_nodes[0].mx[k])));
        br.has_uv = has_uv ? 1 : 0;
        br.face_base = face_base;
        if (!local_nodes.empty())
          for (int k = 0; k < 3; k++) br.root_mn[k] = local_nodes[0].mn[k], br.root_mx[k] = local_nodes[0].mx[k];
        if (br.root >= 0) {
          build_top_entries(qnodes, br.sub_root, tops);  // row = index of the record
          bvh_recs.push_back(br);
          push_run(RUN_BVH, (int)bvh_recs.size() - 1);
          features |= F_BVH;
        }
        break;
      }
    }
  }
  for (int i = 0; i < 4; i++) runs.push_back(Run{-1, 0, 0, 0});  // the kernel fetches runs four at a time
  if (!face_uv.empty()) face_uv.resize(faces.size() * 6, 0.f);
  if (!faces.empty())  // the kernel fetches sub-leaf faces four at a time
    for (int i = 0; i < 4; i++) faces.push_back(FaceRec{});
  // spatial groups for the long sphere runs (scene_dev.h: SphGroup)
  for (Run &run : runs) {
    if (run.kind != RUN_SPHERE || run.count < kSphGroupMin || run.count > 65535) continue;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int i = 0; i < run.count; i++) {
      const SphereRec &sp = spheres[run.first + i];
      const double c[3] = {sp.cx, sp.cy, sp.cz};
      for (int a = 0; a < 3; a++) lo[a] = std::min(lo[a], c[a]), hi[a] = std::max(hi[a], c[a]);
    }
    // spheres much larger than the typical one (a ground sphere among marbles) are kept together, behind the rest:
    // in a group of small ones they would blow its bounds up to their own
    std::vector<double> radii(run.count);
    for (int i = 0; i < run.count; i++) radii[i] = std::fabs(spheres[run.first + i].radius);
    std::nth_element(radii.begin(), radii.begin() + run.count / 2, radii.end());
    const double big = 8.0 * radii[run.count / 2];
    std::vector<std::pair<uint32_t, int>> order(run.count);
    for (int i = 0; i < run.count; i++) {
      const SphereRec &sp = spheres[run.first + i];
      const double c[3] = {sp.cx, sp.cy, sp.cz};
      uint32_t code = 0;
      for (int a = 0; a < 3; a++) {
        const double ext = hi[a] - lo[a];
        uint32_t qv = ext > 0 ? (uint32_t)std::min(1023.0, std::floor((c[a] - lo[a]) / ext * 1024.0)) : 0u;
        for (int bit = 0; bit < 10; bit++) code |= ((qv >> bit) & 1u) << (3 * bit + a);  // Morton interleave
      }
      if (std::fabs(sp.radius) > big) code |= 0x80000000u;
      order[i] = {code, i};
    }
    std::stable_sort(order.begin(), order.end());
    run.pad = (int)sph_groups.size();
    for (int g0 = 0; g0 < run.count; g0 += kSphGroupSize) {
      SphGroup g{};
      g.first = (int)sph_members.size();
      g.count = std::min(kSphGroupSize, run.count - g0);
      float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
      for (int k = 0; k < g.count; k++) {
        const int idx = run.first + order[g0 + k].second;
        const SphereRec &sp = spheres[idx];
        SphMember m{};
        m.cx = sp.cx, m.cy = sp.cy, m.cz = sp.cz, m.r2f = (float)sp.r2, m.orig = idx, m.r2 = sp.r2;
        sph_members.push_back(m);
        // the member's own box, generously: |radius| (1 + 2^-10) on every side
        const float r = (float)(std::fabs(sp.radius) * (1.0 + (double)kSphRadiusPad)) + kPadFloor;
        const float c[3] = {sp.cx, sp.cy, sp.cz};
        for (int a = 0; a < 3; a++) mn[a] = fminf(mn[a], c[a] - r), mx[a] = fmaxf(mx[a], c[a] + r);
      }
      float diag = 0.f, mag = 0.f;
      for (int a = 0; a < 3; a++) diag = fmaxf(diag, mx[a] - mn[a]), mag = fmaxf(mag, fmaxf(fabsf(mn[a]), fabsf(mx[a])));
      // + the scene-dependent part of the distance slack, 2^-9 of the group's own largest coordinate (the kernel adds
      // the ray's: 2^-9 |o|): together at least 8e-4 |o - c| for every member, see closest_hit.h
      const float pad = fixed_pad(diag, mag) + kSphDistSlack * mag;
      for (int a = 0; a < 3; a++) g.mn[a] = mn[a] - pad, g.mx[a] = mx[a] + pad;
      sph_mag = fmaxf(sph_mag, mag);
      sph_groups.push_back(g);
    }
  }
  if (!sph_groups.empty()) features |= F_SGROUP;
  if (!sph_groups.empty()) {  // look-ahead targets of the scalar ping-pong loads
    sph_groups.push_back(SphGroup{});
    sph_members.push_back(SphMember{});
  }
  for (size_t t = 0; t < 2 * pair_pts.size() && t < tris.size(); t++) {
    TriNrm r{};
    for (int c = 0; c < 3; c++) r.n[c] = tris[t].n[c];
    r.mat_flags = (tris[t].mat & 0xffffff) | (tris[t].flags << 24);
    tri_nrm.push_back(r);
  }
  if (!tris.empty()) {  // prefetch target past the last pair (fetched, never tested)
    tris.push_back(HotTri{});
    tris.push_back(HotTri{});
    pair_boxes.push_back(PairBox{});
  }
  n_spheres = (int)spheres.size();
  if (!spheres.empty()) spheres.push_back(SphereRec{});  // same for the sphere look-ahead
  for (const MatRec &m : mat_recs)
    if (m.tex >= 0) bytes_per_ray += 4;  // one RGBA8 texel per textured hit
  return check_margins();
}

// The error budget of the world list's structures (margins.h), record by record: a pair's bounds contain its corners
// with the fixed pad, or are unbounded when one of its triangles is thinner than the distance slack provides for; a
// sphere group's bounds contain every member's own box with the fixed pad and the scene's share of the distance slack;
// a top-table entry contains the child box it stands for and carries its node's slack exponent.
std::string Scene::check_margins() const {
  for (size_t i = 0; i < pair_pts.size(); i++) {
    const PairPts &pp = pair_pts[i];
    const PairBox &bx = pair_boxes[i];
    const bool lone = !(tris[2 * i].flags & TRI_PGRAM);
    const V3 q[4] = {mk(pp.p0[0], pp.p0[1], pp.p0[2]), mk(pp.p1[0], pp.p1[1], pp.p1[2]), mk(pp.p2[0], pp.p2[1], pp.p2[2]),
                     lone ? mk(pp.p2[0], pp.p2[1], pp.p2[2]) : mk(pp.p3[0], pp.p3[1], pp.p3[2])};
    const V3 second[3] = {q[1], q[2], q[3]};
    const bool thin = face_slack_exponent(q) > 0 || (!lone && face_slack_exponent(second) > 0);
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int j = 0; j < 4; j++) {
      const float c[3] = {q[j].x, q[j].y, q[j].z};
      for (int a = 0; a < 3; a++) mn[a] = fminf(mn[a], c[a]), mx[a] = fmaxf(mx[a], c[a]);
    }
    float diag = 0.f, mag = 0.f;
    for (int a = 0; a < 3; a++) diag = fmaxf(diag, mx[a] - mn[a]), mag = fmaxf(mag, fmaxf(fabsf(mn[a]), fabsf(mx[a])));
    const double pad = 0.98 * (double)fixed_pad(diag, mag);
    for (int a = 0; a < 3; a++) {
      if (thin ? !(bx.mn[a] == -INFINITY && bx.mx[a] == INFINITY)
               : !((double)bx.mn[a] <= (double)mn[a] - pad && (double)bx.mx[a] >= (double)mx[a] + pad))
        return "margin budget: a world-list pair's bounds do not cover what its triangle tests can accept";
    }
    if (!(mag <= list_mag)) return "margin budget: list_mag is below a pair's coordinates";
  }
  for (size_t g = 0; g + 1 < sph_groups.size(); g++) {  // (the last record is the look-ahead's padding)
    const SphGroup &gr = sph_groups[g];
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int k = 0; k < gr.count; k++) {
      const SphMember &m = sph_members[(size_t)gr.first + k];
      const double r = std::fabs(spheres[(size_t)m.orig].radius) * (1.0 + (double)kSphRadiusPad) * 0.999999;
      const float c[3] = {m.cx, m.cy, m.cz};
      for (int a = 0; a < 3; a++) mn[a] = fminf(mn[a], (float)((double)c[a] - r)), mx[a] = fmaxf(mx[a], (float)((double)c[a] + r));
    }
    float diag = 0.f, mag = 0.f;
    for (int a = 0; a < 3; a++) diag = fmaxf(diag, mx[a] - mn[a]), mag = fmaxf(mag, fmaxf(fabsf(mn[a]), fabsf(mx[a])));
    const double pad = 0.98 * ((double)fixed_pad(diag, mag) + (double)kSphDistSlack * mag);
    for (int a = 0; a < 3; a++)
      if (!((double)gr.mn[a] <= (double)mn[a] - pad && (double)gr.mx[a] >= (double)mx[a] + pad))
        return "margin budget: a sphere group's bounds do not cover what its members' pre-tests can accept";
  }
  for (size_t b = 0; b < bvh_recs.size(); b++) {
    for (int i = 0; i < kTopEntries; i++) {
      const BvhNode &t = tops[b * kTopEntries + (size_t)i];
      if (t.left == -1) continue;
      // the entry stands for child `t.left` of some node of this mesh's search tree: find the parent's box for it
      bool found = false;
      for (size_t n = 0; n < qnodes.size() && !found; n++)
        for (int c = 0; c < 4 && !found; c++)
          if (qnodes[n].child[c] == t.left) {
            found = true;
            for (int a = 0; a < 3; a++) {
              const double lo = (double)qnodes[n].origin[a] + std::ldexp((double)qnodes[n].qlo[a][c], qnodes[n].exp[a]);
              const double hi = (double)qnodes[n].origin[a] + std::ldexp((double)qnodes[n].qhi[a][c], qnodes[n].exp[a]);
              if (!((double)t.mn[a] <= lo && (double)t.mx[a] >= hi)) return "margin budget: a top-table entry is smaller than the child box it stands for";
            }
            if (t.right < qnodes[n].slack_exp) return "margin budget: a top-table entry's slack exponent is below its node's";
          }
      if (!found) return "margin budget: a top-table entry refers to no child of the search tree";
    }
  }
  return "";
}


}  // namespace rtmi
