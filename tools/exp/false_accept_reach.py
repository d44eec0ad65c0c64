"""How far from a face can a ray be that the binary32 Moller-Trumbore test (utils.cu:49-85) still accepts?
Numpy emulation of the reference's operation order in binary32 against the exact hit point in binary64.
For a face with edges e1, e2 (angle theta at p0) seen from distance D under the angle phi to its normal, prints the
largest in-plane distance of an accepted ray's exact plane point from the triangle, next to two bounds:
  a = 7 eps D / (sin(theta) cos(phi))          (the error of dot(tvec, pvec) / det)
  b = 7 eps D |e1| |e2| / 1e-7                 (a with the test's own cut-off |det| >= 1e-7)
usage: tools/exp/false_accept_reach.py"""
import numpy as np
f32 = np.float32
EPS = 2.0 ** -24


def cross(a, b):
    return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1], a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                     a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], -1)


def dot(a, b):
    return (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]


def accept32(p, o, d):
    p = p.astype(f32); o = o.astype(f32); d = d.astype(f32)
    e1, e2 = p[1] - p[0], p[2] - p[0]
    pv = cross(d, e2[None])
    det = dot(e1[None], pv)
    ok = np.abs(det.astype(np.float64)) >= 1e-7
    inv = f32(1) / det
    tv = o - p[0][None]
    u = dot(tv, pv) * inv
    ok &= (u >= 0) & (u <= 1)
    qv = cross(tv, e1[None])
    v = dot(d, qv) * inv
    ok &= (v >= 0) & (u + v <= 1)
    t = dot(e2[None], qv) * inv
    ok &= (t >= 1e-3)
    return ok


def plane_point_dist(p, o, d):
    """exact: where the ray meets the face's plane, and that point's distance from the triangle (0 inside)"""
    p = p.astype(np.float64); o = o.astype(np.float64); d = d.astype(np.float64)
    n = np.cross(p[1] - p[0], p[2] - p[0])
    t = ((p[0] - o) @ n) / (d @ n)
    P = o + t[:, None] * d
    best = np.full(len(P), np.inf)
    inside = np.ones(len(P), bool)
    for a in range(3):
        A, B, C = p[a], p[(a + 1) % 3], p[(a + 2) % 3]
        ab = B - A
        s = np.clip(((P - A) @ ab) / (ab @ ab), 0, 1)
        best = np.minimum(best, np.linalg.norm(P - (A + s[:, None] * ab), axis=1))
        side = np.cross(ab, P - A) @ n
        inside &= side * (np.cross(ab, C - A) @ n) >= 0
    return np.where(inside, 0.0, best)


rng = np.random.default_rng(1)
print("  size   theta   phi       D   accepted   reach      a=7epsD/(sin cos)   b=7epsD|e1||e2|/1e-7   slack 2^-16 D")
for size in (2e-3, 2e-2, 0.3):
    for theta_deg in (60, 5, 0.6, 0.1):
        for phi_deg in (0, 60, 85, 89.5):
            for D in (1e2, 1e3, 1e4):
                th = np.radians(theta_deg); ph = np.radians(phi_deg)
                p = np.array([[0, 0, 0], [size, 0, 0], [size * np.cos(th), size * np.sin(th), 0]]) + np.array([0.013, 0.021, 0.017])
                centre = p.mean(0)
                az = rng.uniform(0, 2 * np.pi, 400000)
                # view direction phi off the normal (z), any azimuth; aim points scattered around the face
                view = np.stack([np.sin(ph) * np.cos(az), np.sin(ph) * np.sin(az), np.full_like(az, np.cos(ph))], -1)
                reach_guess = min(7 * EPS * D / max(np.sin(th) * np.cos(ph), 1e-9), 7 * EPS * D * size * size / 1e-7, 50 * size + 1e-3 * D)
                spread = size + 3 * reach_guess
                aim = centre + np.stack([rng.uniform(-spread, spread, len(az)), rng.uniform(-spread, spread, len(az)), np.zeros(len(az))], -1)
                o = (aim + D * view).astype(f32)
                d = (aim - o.astype(np.float64))
                d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(f32)
                d = (d / np.sqrt(dot(d, d))[:, None]).astype(f32)
                ok = accept32(p, o, d)
                dist = plane_point_dist(p.astype(f32), o, d)
                reach = dist[ok].max() if ok.any() else 0.0
                a = 7 * EPS * D / max(np.sin(th) * np.cos(ph), 1e-12)
                bb = 7 * EPS * D * size * size / 1e-7
                print("%6.0e %6.1f %5.1f %7.0e %9d %9.2e %12.2e %18.2e %16.2e %s" % (size, theta_deg, phi_deg, D, ok.sum(), reach, a, bb, D * 2.0 ** -16,
                      "BEYOND SLACK" if reach * np.cos(ph) > D * 2.0 ** -16 else ""))
