// CPU experiment behind the mesh-search redesign (DESIGN.md "Mesh queries"): counts what a ray of the
// C3 workload costs under different search structures, WITHOUT a GPU.  Not product code and not
// the oracle: rays are generated with an ordinary RNG and true-closest-hit bounces; only the
// statistics matter (node visits, face tests, 16-byte requests per ray, how often a pruned
// closest-hit search can be proven exact against the reference's tree).
//
//   g++ -O2 -std=c++17 tools/mesh_sim/mesh_sim.cc -o /tmp/mesh_sim
//   python3 tools/mesh_sim/dump_mesh.py /tmp/bunny.f32 && /tmp/mesh_sim /tmp/bunny.f32
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <random>
#include <vector>

struct V3 {
  float x, y, z;
};
static V3 mk(float x, float y, float z) { return V3{x, y, z}; }
static V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static V3 cross(V3 a, V3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
static V3 unit(V3 v) { return v * (1.0f / std::sqrt(dot(v, v))); }
static float comp(V3 v, int a) { return a == 0 ? v.x : a == 1 ? v.y : v.z; }

struct Face {
  V3 p[3];
  int orig;       // index in reference order
  uint32_t code;  // reference leaf path code
};
struct Box {
  float mn[3], mx[3];
  void clear() {
    for (int k = 0; k < 3; k++) mn[k] = INFINITY, mx[k] = -INFINITY;
  }
  void add(V3 p) {
    for (int k = 0; k < 3; k++) mn[k] = std::min(mn[k], comp(p, k)), mx[k] = std::max(mx[k], comp(p, k));
  }
  void add(const Box &b) {
    for (int k = 0; k < 3; k++) mn[k] = std::min(mn[k], b.mn[k]), mx[k] = std::max(mx[k], b.mx[k]);
  }
  float area() const {
    float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    return 2 * (dx * dy + dy * dz + dz * dx);
  }
};

static bool tri_hit(const Face &f, V3 o, V3 d, float t_to, float &t) {
  V3 e1 = f.p[1] - f.p[0], e2 = f.p[2] - f.p[0];
  V3 pv = cross(d, e2);
  float det = dot(e1, pv);
  if (std::fabs(det) < 1e-7f) return false;
  float inv = 1.0f / det;
  V3 tv = o - f.p[0];
  float u = dot(tv, pv) * inv;
  if (u < 0 || u > 1) return false;
  V3 qv = cross(tv, e1);
  float v = dot(d, qv) * inv;
  if (v < 0 || u + v > 1) return false;
  t = dot(e2, qv) * inv;
  return t >= 1e-3f && t <= t_to;
}

// slab test; returns entry distance through *tn
static bool slab(const Box &b, V3 o, V3 inv, float lo, float hi, float *tn) {
  float t0x = (b.mn[0] - o.x) * inv.x, t1x = (b.mx[0] - o.x) * inv.x;
  float t0y = (b.mn[1] - o.y) * inv.y, t1y = (b.mx[1] - o.y) * inv.y;
  float t0z = (b.mn[2] - o.z) * inv.z, t1z = (b.mx[2] - o.z) * inv.z;
  float en = std::max(std::max(std::min(t0x, t1x), std::min(t0y, t1y)), std::min(t0z, t1z));
  float le = std::min(std::min(std::max(t0x, t1x), std::max(t0y, t1y)), std::max(t0z, t1z));
  en = std::max(en, lo);
  le = std::min(le, hi);
  *tn = en;
  return en <= le * 1.00001f + 1e-6f;
}

// reference AABB::Hit: the segment crosses the box surface
static bool aabb_ref(const Box &b, V3 o, V3 d, float t_to) {
  const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
  for (int i = 0; i < 3; i++) {
    if (dd[i] == 0.f) continue;
    for (int s = 0; s < 2; s++) {
      float plane = s == 0 ? b.mn[i] : b.mx[i];
      float tf = (plane - oo[i]) / dd[i];
      if (!(std::fabs(tf) < INFINITY)) continue;
      if (!(1e-3f <= tf && tf <= t_to)) continue;
      bool inside = true;
      for (int a = 0; a < 3; a++) {
        if (a == i) continue;
        float pa = oo[a] + tf * dd[a];
        if (!(b.mn[a] <= pa && pa <= b.mx[a])) inside = false;
      }
      if (inside) return true;
    }
  }
  return false;
}

// ------------------------------------------------------------------ reference tree (x-sorted median)
struct RefNode {
  Box box;
  int left, right;  // leaf: left=-1
  int first, n;
};
static std::vector<RefNode> g_ref;
static int build_ref(std::vector<Face> &f, int first, int n, int leaf_max, int level, uint32_t code) {
  RefNode nd;
  nd.box.clear();
  for (int i = 0; i < n; i++)
    for (int j = 0; j < 3; j++) nd.box.add(f[first + i].p[j]);
  nd.first = first, nd.n = n, nd.left = nd.right = -1;
  int me = (int)g_ref.size();
  g_ref.push_back(nd);
  if (n <= leaf_max) {
    for (int i = 0; i < n; i++) f[first + i].code = code;
    return me;
  }
  int mid = (n - 1) / 2;
  uint32_t bit = 0x80000000u >> level;
  int l = build_ref(f, first, mid + 1, leaf_max, level + 1, code);
  int r = build_ref(f, first + mid + 1, n - mid - 1, leaf_max, level + 1, code | bit);
  g_ref[me].left = l, g_ref[me].right = r;
  return me;
}

// ------------------------------------------------------------------ binary BVH (median or binned SAH)
struct BNode {
  Box box;
  int left = -1, right = -1;  // children (binary) or -1
  int first = 0, n = 0;
};
struct Builder {
  std::vector<Face> *faces;
  std::vector<BNode> nodes;
  int leaf_max;
  bool sah;
  int build(int first, int n) {
    BNode nd;
    nd.box.clear();
    Box cb;
    cb.clear();
    auto &F = *faces;
    for (int i = 0; i < n; i++) {
      V3 c = mk(0, 0, 0);
      for (int j = 0; j < 3; j++) nd.box.add(F[first + i].p[j]), c = c + F[first + i].p[j];
      cb.add(c);
    }
    nd.first = first, nd.n = n;
    int me = (int)nodes.size();
    nodes.push_back(nd);
    if (n <= leaf_max) return me;
    int axis = 0;
    for (int k = 1; k < 3; k++)
      if (cb.mx[k] - cb.mn[k] > cb.mx[axis] - cb.mn[axis]) axis = k;
    int mid = n / 2;
    auto cen = [&](const Face &f, int a) { return comp(f.p[0], a) + comp(f.p[1], a) + comp(f.p[2], a); };
    if (sah) {
      // binned SAH over all three axes
      const int NB = 16;
      float best = INFINITY;
      int best_axis = -1, best_bin = -1;
      for (int a = 0; a < 3; a++) {
        float lo = cb.mn[a], hi = cb.mx[a];
        if (!(hi > lo)) continue;
        Box bb[NB];
        int cnt[NB] = {0};
        for (int b = 0; b < NB; b++) bb[b].clear();
        for (int i = 0; i < n; i++) {
          int b = std::min(NB - 1, (int)((cen(F[first + i], a) - lo) / (hi - lo) * NB));
          cnt[b]++;
          for (int j = 0; j < 3; j++) bb[b].add(F[first + i].p[j]);
        }
        Box acc;
        float la[NB], ra[NB];
        int lc[NB], rc[NB];
        acc.clear();
        int c = 0;
        for (int b = 0; b < NB; b++) {
          acc.add(bb[b]), c += cnt[b];
          la[b] = c ? acc.area() : 0, lc[b] = c;
        }
        acc.clear();
        c = 0;
        for (int b = NB - 1; b >= 0; b--) {
          acc.add(bb[b]), c += cnt[b];
          ra[b] = c ? acc.area() : 0, rc[b] = c;
        }
        for (int b = 0; b + 1 < NB; b++) {
          if (!lc[b] || !rc[b + 1]) continue;
          float cost = la[b] * lc[b] + ra[b + 1] * rc[b + 1];
          if (cost < best) best = cost, best_axis = a, best_bin = b;
        }
      }
      if (best_axis >= 0) {
        float lo = cb.mn[best_axis], hi = cb.mx[best_axis];
        auto it = std::partition(F.begin() + first, F.begin() + first + n, [&](const Face &f) {
          int b = std::min(NB - 1, (int)((cen(f, best_axis) - lo) / (hi - lo) * NB));
          return b <= best_bin;
        });
        mid = (int)(it - (F.begin() + first));
        if (mid == 0 || mid == n) mid = n / 2, best_axis = -1;
      }
      if (best_axis < 0)
        std::nth_element(F.begin() + first, F.begin() + first + mid, F.begin() + first + n,
                         [&](const Face &a, const Face &b) { return cen(a, axis) < cen(b, axis); });
    } else {
      std::nth_element(F.begin() + first, F.begin() + first + mid, F.begin() + first + n,
                       [&](const Face &a, const Face &b) { return cen(a, axis) < cen(b, axis); });
    }
    int l = build(first, mid);
    int r = build(first + mid, n - mid);
    nodes[me].left = l, nodes[me].right = r;
    return me;
  }
};

// ------------------------------------------------------------------ wide tree by collapsing the binary tree
struct WNode {
  int nch = 0;
  Box box[8];
  int child[8];  // >=0: wide node; <0: leaf -(first*16+n)-1
};
struct Wide {
  int W;
  std::vector<WNode> nodes;
  int depth = 0;
};
static int collapse(const std::vector<BNode> &b, int root, Wide &w, int level) {
  w.depth = std::max(w.depth, level + 1);
  int me = (int)w.nodes.size();
  w.nodes.emplace_back();
  std::vector<int> kids = {root};
  // expand the child with the largest area until W children (or all leaves)
  if (b[root].left >= 0) {
    kids = {b[root].left, b[root].right};
    while ((int)kids.size() < w.W) {
      int best = -1;
      float ba = -1;
      for (int i = 0; i < (int)kids.size(); i++)
        if (b[kids[i]].left >= 0 && b[kids[i]].box.area() > ba) ba = b[kids[i]].box.area(), best = i;
      if (best < 0) break;
      int k = kids[best];
      kids[best] = b[k].left;
      kids.push_back(b[k].right);
    }
  }
  WNode nd;
  nd.nch = (int)kids.size();
  for (int i = 0; i < nd.nch; i++) {
    nd.box[i] = b[kids[i]].box;
    if (b[kids[i]].left < 0)
      nd.child[i] = -(b[kids[i]].first * 16 + b[kids[i]].n) - 1;
    else
      nd.child[i] = collapse(b, kids[i], w, level + 1);
  }
  w.nodes[me] = nd;
  return me;
}

// ------------------------------------------------------------------ searches
struct Counts {
  double rays = 0, nodes = 0, leaves = 0, faces = 0, boxtests = 0, maxstack = 0, hitleaves = 0;
  void add(const Counts &o) {
    rays += o.rays, nodes += o.nodes, leaves += o.leaves, faces += o.faces, boxtests += o.boxtests;
    maxstack = std::max(maxstack, o.maxstack), hitleaves += o.hitleaves;
  }
};

struct HitRes {
  bool hit = false;
  float t = 0;
  int face = -1;
};

// all = true: collect every hit (no pruning by nearer hits); ordered: push far children first
static HitRes search(const Wide &w, const std::vector<Face> &F, V3 o, V3 d, float t_in, bool prune, bool ordered,
                     Counts &c, std::vector<std::pair<uint32_t, float>> *hits_by_leaf = nullptr) {
  V3 inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  HitRes best;
  float hi = t_in;
  std::vector<int> st = {0};
  std::vector<float> stt = {0.f};
  c.rays += 1;
  while (!st.empty()) {
    c.maxstack = std::max(c.maxstack, (double)st.size());
    int e = st.back();
    float et = stt.back();
    st.pop_back(), stt.pop_back();
    if (prune && et > hi) continue;
    if (e >= 0) {
      c.nodes += 1;
      const WNode &nd = w.nodes[e];
      int idx[8];
      float tn[8];
      int k = 0;
      for (int i = 0; i < nd.nch; i++) {
        c.boxtests += 1;
        float t;
        if (slab(nd.box[i], o, inv, 1e-3f, prune ? hi : t_in, &t)) idx[k] = i, tn[k] = t, k++;
      }
      if (ordered)  // far first so the nearest is popped first
        for (int a = 0; a < k; a++)
          for (int b2 = a + 1; b2 < k; b2++)
            if (tn[b2] > tn[a]) std::swap(tn[a], tn[b2]), std::swap(idx[a], idx[b2]);
      for (int a = 0; a < k; a++) st.push_back(nd.child[idx[a]]), stt.push_back(tn[a]);
    } else {
      c.leaves += 1;
      int enc = -(e + 1), n = enc & 15, first = enc >> 4;
      for (int i = 0; i < n; i++) {
        c.faces += 1;
        float t;
        if (tri_hit(F[first + i], o, d, prune ? hi : t_in, t)) {
          if (hits_by_leaf) {
            bool found = false;
            for (auto &h : *hits_by_leaf)
              if (h.first == F[first + i].code) {
                found = true;
                h.second = std::min(h.second, t);
              }
            if (!found) hits_by_leaf->push_back({F[first + i].code, t});
          }
          if (!best.hit || t < best.t || (t == best.t && F[first + i].orig > F[best.face].orig))
            best.hit = true, best.t = t, best.face = first + i;
          if (prune) hi = best.t;
        }
      }
    }
  }
  return best;
}

// is the pruned result provably the reference's answer?  All boxes on the root-to-leaf path of the
// winner's reference leaf must be entered already with T = t*
static bool verify_at_tstar(int ref_root, uint32_t code, V3 o, V3 d, float tstar) {
  int ni = ref_root;
  for (int lvl = 0;; lvl++) {
    const RefNode &nd = g_ref[ni];
    if (nd.left < 0) return true;
    ni = (code & (0x80000000u >> lvl)) ? nd.right : nd.left;
    if (!aabb_ref(g_ref[ni].box, o, d, tstar)) return false;
  }
}

// full reference semantics from the list of hit leaves
static bool replay(int ref_root, std::vector<std::pair<uint32_t, float>> hits, V3 o, V3 d, float t_in, float *tout) {
  std::sort(hits.begin(), hits.end());
  float T = t_in;
  bool any = false;
  // naive: evaluate path tests per leaf with memo of node decisions
  std::vector<int> memo(g_ref.size(), -1);
  for (auto &h : hits) {
    int ni = ref_root;
    bool ent = true;
    for (int lvl = 0;; lvl++) {
      const RefNode &nd = g_ref[ni];
      if (nd.left < 0) break;
      ni = (h.first & (0x80000000u >> lvl)) ? nd.right : nd.left;
      if (memo[ni] < 0) memo[ni] = aabb_ref(g_ref[ni].box, o, d, T) ? 1 : 0;
      if (!memo[ni]) {
        ent = false;
        break;
      }
    }
    if (ent && h.second <= T) T = h.second, any = true;
  }
  *tout = T;
  return any;
}

int main(int argc, char **argv) {
  if (argc < 2) return 1;
  FILE *fp = fopen(argv[1], "rb");
  if (!fp) return 1;
  fseek(fp, 0, SEEK_END);
  long sz = ftell(fp);
  fseek(fp, 0, SEEK_SET);
  int nf = (int)(sz / 36);
  std::vector<float> raw((size_t)nf * 9);
  if (fread(raw.data(), 36, nf, fp) != (size_t)nf) return 1;
  fclose(fp);
  std::vector<Face> faces(nf);
  for (int i = 0; i < nf; i++)
    for (int j = 0; j < 3; j++) faces[i].p[j] = mk(raw[i * 9 + j * 3], raw[i * 9 + j * 3 + 1], raw[i * 9 + j * 3 + 2]);
  std::stable_sort(faces.begin(), faces.end(), [](const Face &a, const Face &b) { return a.p[0].x < b.p[0].x; });
  for (int i = 0; i < nf; i++) faces[i].orig = i;
  int ref_root = build_ref(faces, 0, nf, 2048, 0, 0u);
  printf("%d faces, %zu reference nodes\n", nf, g_ref.size());

  const int side = argc > 2 ? atoi(argv[2]) : 192, spp = argc > 3 ? atoi(argv[3]) : 2, max_depth = 10;
  // scene: bunny.cu camera, wall parallelogram at z = 2, sky
  V3 cam = mk(-0.025f, 0.1f, -0.5f);
  float hh = std::tan(3.14159265f * 2 / 9 / 2);
  V3 wall0 = mk(-0.525f, -0.4f, 2), wall1 = mk(0.475f, -0.4f, 2), wall2 = mk(-0.525f, 0.6f, 2);
  Face wa, wb;
  wa.p[0] = wall0, wa.p[1] = wall1, wa.p[2] = wall2;
  V3 wall3 = wall1 + wall2 - wall0;
  wb.p[0] = wall1, wb.p[1] = wall2, wb.p[2] = wall3;

  struct Variant {
    const char *name;
    int W, leaf;
    bool sah, prune, ordered;
  };
  std::vector<Variant> vars = {
      {"median W4 leaf4 unpruned (current)", 4, 4, false, false, false},
      {"median W4 leaf4 pruned ordered", 4, 4, false, true, true},
      {"SAH    W4 leaf4 unpruned", 4, 4, true, false, false},
      {"SAH    W4 leaf4 pruned ordered", 4, 4, true, true, true},
      {"SAH    W4 leaf4 pruned unordered", 4, 4, true, true, false},
      {"SAH    W4 leaf2 pruned ordered", 4, 2, true, true, true},
      {"SAH    W4 leaf1 pruned ordered", 4, 1, true, true, true},
      {"SAH    W8 leaf4 pruned ordered", 8, 4, true, true, true},
      {"SAH    W8 leaf4 unpruned", 8, 4, true, false, false},
      {"SAH    W8 leaf1 pruned ordered", 8, 1, true, true, true},
      {"SAH    W2 leaf4 pruned ordered", 2, 4, true, true, true},
  };
  // generate the ray set once with the first structure
  struct Ray {
    V3 o, d;
    float t_in;
    int depth;
  };
  std::vector<Ray> rays;
  {
    std::vector<Face> F = faces;
    Builder b;
    b.faces = &F, b.leaf_max = 4, b.sah = true;
    b.build(0, nf);
    Wide w;
    w.W = 4;
    collapse(b.nodes, 0, w, 0);
    std::mt19937 rng(12345);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    Counts dummy;
    for (int i = 0; i < side; i++)
      for (int j = 0; j < side; j++)
        for (int s = 0; s < spp; s++) {
          float x = (j + U(rng)) / side * 2 - 1, y = 1 - (i + U(rng)) / side * 2;
          V3 o = cam, d = unit(mk(x * hh, y * hh, 1.f));
          for (int depth = 0; depth <= max_depth; depth++) {
            float t_in = 1e9f, tw;
            bool wall = false;
            if (tri_hit(wa, o, d, t_in, tw) || tri_hit(wb, o, d, t_in, tw)) t_in = tw, wall = true;
            rays.push_back({o, d, t_in, depth});
            HitRes h = search(w, F, o, d, t_in, true, true, dummy);
            V3 n;
            float t;
            if (h.hit) {
              const Face &f = F[h.face];
              n = unit(cross(f.p[1] - f.p[0], f.p[2] - f.p[0]));
              t = h.t;
            } else if (wall) {
              n = mk(0, 0, -1), t = t_in;
            } else
              break;  // sky
            if (dot(n, d) > 0) n = n * -1.f;
            V3 p = o + d * t, sdir;
            do {
              sdir = mk(U(rng) * 2 - 1, U(rng) * 2 - 1, U(rng) * 2 - 1);
            } while (dot(sdir, sdir) > 1);
            o = p, d = unit(unit(sdir) + n);
          }
        }
  }
  size_t n0 = 0;
  for (auto &r : rays) n0 += r.depth == 0;
  printf("%zu rays (%zu camera, %zu bounce), %.3f rays/sample\n", rays.size(), n0, rays.size() - n0,
         (double)rays.size() / n0);

  for (const Variant &v : vars) {
    std::vector<Face> F = faces;
    Builder b;
    b.faces = &F, b.leaf_max = v.leaf, b.sah = v.sah;
    b.build(0, nf);
    Wide w;
    w.W = v.W;
    collapse(b.nodes, 0, w, 0);
    Counts c0, c1;
    double ver_pass = 0, ver_n = 0, mismatch = 0, hit0 = 0, hit1 = 0;
    for (const Ray &r : rays) {
      Counts c;
      std::vector<std::pair<uint32_t, float>> hl;
      HitRes h = search(w, F, r.o, r.d, r.t_in, v.prune, v.ordered, c, v.prune ? nullptr : &hl);
      if (!v.prune) c.hitleaves = (double)hl.size();
      (r.depth == 0 ? c0 : c1).add(c);
      if (h.hit) (r.depth == 0 ? hit0 : hit1) += 1;
      if (v.prune && h.hit && r.depth > 0) {
        ver_n += 1;
        if (verify_at_tstar(ref_root, F[h.face].code, r.o, r.d, h.t)) ver_pass += 1;
      }
      if (!v.prune && r.depth > 0) {
        float T;
        bool any = replay(ref_root, hl, r.o, r.d, r.t_in, &T);
        if (any != h.hit || (any && T != h.t)) mismatch += 1;
      }
    }
    printf("%-38s nodes %zu depth %d\n", v.name, w.nodes.size(), w.depth);
    auto pr = [&](const char *tag, const Counts &c, double hits) {
      printf("   %-7s rays %9.0f hit %5.1f%%  nodes/ray %6.2f  leaves/ray %5.2f  faces/ray %6.2f  boxtests/ray %6.1f  "
             "maxstack %2.0f  hitleaves/ray %.2f\n",
             tag, c.rays, 100.0 * hits / c.rays, c.nodes / c.rays, c.leaves / c.rays, c.faces / c.rays,
             c.boxtests / c.rays, c.maxstack, c.hitleaves / c.rays);
    };
    pr("camera", c0, hit0);
    pr("bounce", c1, hit1);
    if (v.prune) printf("   bounce rays with a hit: %.0f, provably exact at t*: %.1f%%\n", ver_n, 100.0 * ver_pass / std::max(1.0, ver_n));
    if (!v.prune) printf("   bounce rays where reference replay != true closest: %.0f (%.2f%% of bounce rays)\n", mismatch, 100.0 * mismatch / c1.rays);
  }
  return 0;
}
