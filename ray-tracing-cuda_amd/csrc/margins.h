// The error budget of the three culls (DESIGN.md "Error budget of the culls"): ONE place for the constants, the
// inequalities between them (static_assert), and the host-side check every emitted record goes through at commit
// (scene.hip: Scene::check_margins, check_search_tree).
//
// What is being budgeted.  The culled world-list scan, the grouped sphere scan and the mesh search all decide "this
// ray cannot be accepted by that primitive's test" from bounds, while the reference tests everything
// (hitable_list.cu:7-25, bvh.cuh:123-158).  The primitives' own tests are binary32 and accept rays that miss the exact
// primitive: the cull is exact iff its bounds contain every point where the TEST can still say yes.
//
//   term                         value                          covers
//   ---------------------------  -----------------------------  -------------------------------------------------------------
//   reach of the triangle test   <= 2.1 eps D / sin(theta)      utils.cu:49-85 in binary32: the error of dot(tvec, pvec) / det
//   (false accepts, measured)    eps = 2^-24, D = |o - p0|,     moves (u, v); measured over theta, phi, D, face size by
//                                theta = smallest angle         tools/exp/false_accept_reach.py; |det| >= 1e-7 (utils.cu:60)
//                                                               removes the edge-on views before 1 / cos(phi) grows
//   distance slack (triangles)   kDistSlack 2^k (|o|inf + mag)  provides 8 eps D' / sin(theta) with D' = |o|inf + mag >=
//                                per axis, at query time        D / sqrt(3): 8 >= 2.1 sqrt(3) = 3.64.  k = 0 down to
//                                                               sin(theta) = kThinSine = 1/32; thinner faces carry
//                                                               k = ceil(log2(1 / (32 sin theta))) in their nodes
//                                                               (QNode4::slack_exp), thinner world-list pairs are not
//                                                               culled at all (unbounded PairBox)
//   fixed pad of every bound     kPadOfExtent diag +            near range (D ~ diag: 2.1 eps 32 diag = 4e-6 diag) and the
//                                kPadOfMagnitude mag            slab arithmetic's own rounding (one v_rcp, one FMA per plane:
//                                                               < 4 eps of the plane's coordinate = 2.4e-7 mag)
//   8-bit child boxes            rounded OUTWARD on the grid    quantisation can only enlarge (checked per node at commit)
//   time interval of a slab      [kTimeLo t_from,               t of the triangle test against the slab's entry / exit times,
//   test                          kTimeHi t_to + kTimeAbs],     each a few eps relative in a different operation order
//                                 entry (1 - kSlabTimeRel),
//                                 exit (1 + kSlabTimeRel)
//   reach of the sphere pre-test m - r <= sqrt(6e-7) D          sphere.cu:13-17 operands in binary32: m^2 <= r^2 + 6e-7 D^2
//   sphere pre-test cut-off      disc < -kSphDiscRel x terms    the binary32 discriminant's rounding (a handful of eps of its terms)
//   distance slack (spheres)     kSphDistSlack (|o|inf + mag)   2^-9 >= 7.75e-4 sqrt(3) = 1.34e-3
//   sphere member box            |r| (1 + kSphRadiusPad)        the binary64 radius against binary32 bounds
//
// The first and the last-but-two rows are measurements, not theorems; what holds them to account is the every-query
// check build (tests/test_gpu_margins.py): every query answered a second time without any of this, 0 disagreements.
#pragma once

namespace rtmi {

constexpr float kEps32 = 0x1p-24f;              // unit round-off of binary32
constexpr float kReachMeasured = 2.1f;          // false-accept reach <= this x eps x D / sin(theta)
constexpr float kReachProvided = 8.0f;          // the distance slack provides this x eps / sin(theta) of (|o|inf + mag)
constexpr float kThinSine = 1.0f / 32.0f;       // smallest angle a face may have without a slack exponent (1.8 degrees)
constexpr float kDistSlack = 0x1p-16f;          // triangles: per-axis widening as a fraction of (|o|inf + mag)
constexpr float kPadOfExtent = 1e-4f;           // fixed pad: this x the bounds' largest extent ...
constexpr float kPadOfMagnitude = 1e-5f;        // ... + this x their largest |coordinate|
constexpr float kPadFloor = 1e-30f;             // ... + this (bounds of a single point)
constexpr float kTimeLo = 0.999f, kTimeHi = 1.0001f, kTimeAbs = 1e-6f;  // a slab test's [t_from, t_to], widened
constexpr float kSlabTimeRel = 1e-5f;           // a slab's entry / exit times, moved apart by this fraction
constexpr float kSphReach = 7.75e-4f;           // sqrt(6e-7): binary32 sphere pre-test, fraction of D
constexpr float kSphDiscRel = 1e-5f;            // binary32 discriminant below -this x its terms' magnitude: the exact one is negative
constexpr float kSphDistSlack = 0x1p-9f;        // spheres: per-axis widening as a fraction of (|o|inf + mag)
constexpr float kSphRadiusPad = 0x1p-10f;       // a member's own box: |radius| (1 + this)
constexpr float kSqrt3 = 1.7320508f;            // |x|_2 <= sqrt(3) |x|_inf

static_assert(kDistSlack >= kReachProvided * kEps32 / kThinSine, "the distance slack must provide 8 eps / sin(theta) down to the thin-face limit");
static_assert(kReachProvided >= kReachMeasured * kSqrt3, "the provision must cover the measured reach with D taken in the max norm");
static_assert(kPadOfExtent >= kReachMeasured * kEps32 / kThinSine * 4.0f, "the fixed pad covers the reach at distances of the bounds' own size");
static_assert(kPadOfMagnitude >= 8.0f * kEps32, "the fixed pad covers the rounding of the slab arithmetic (a reciprocal and an FMA per plane)");
static_assert(kSphDistSlack >= kSphReach * kSqrt3, "the sphere groups' slack must cover the binary32 pre-test's reach");
static_assert(kTimeLo < 1.0f - 64.0f * kEps32 && kTimeHi > 1.0f + 64.0f * kEps32 && kSlabTimeRel > 64.0f * kEps32,
              "the time fudges must be far above the few-eps disagreement of two operation orders");

// The fixed pad of a set of bounds (the same expression wherever bounds are padded).
inline float fixed_pad(float diag, float mag) { return kPadOfExtent * diag + kPadOfMagnitude * mag + kPadFloor; }

}  // namespace rtmi
