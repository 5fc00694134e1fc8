"""How much of "bit-identical ViewVisibility" rides on the glam operation orders nobody could check here.

The oracle (and the kernels) restate glam 0.33.2's SSE2 op order from memory: glam's sources are not under /root/reference and
there is no Rust toolchain in this image.  The orders that could be remembered wrongly are few:
    Vec4::dot       pairwise (x x' + z z') + (y y' + w w')   [used]   | left to right | fused multiply-adds (two chain orders)
    Vec3A::dot      (x x' + y y') + z z'                      [used]   | x x' + (y y' + z z')  | fused
    Mat3A * Vec3A   ((X v.x) + Y v.y) + Z v.z, mul then add   [used]   | three fused multiply-adds
This module re-evaluates every plane test of a scene under each alternative -- one substitution at a time, everything else as
used -- in numpy float32 (each numpy op is one IEEE rounding, like an SSE lane; an FMA is emulated as the float64 product and sum
rounded once to float32: exact but for double rounding, ~2^-29 per op) and counts the rows whose per-view visibility flips.  It
also histograms how close every DECIDING value `n . c + d + r` sits to zero, in ulps of its largest term: the rows a last-ulp
difference could move at all.

tests/test_boundary_census.py checks the emulation against the oracle (the `used` orders must reproduce the oracle's flags
bit for bit), recomputes the census of the BASELINE configs and compares it with the committed tests/golden/boundary_census.json,
which bench.py quotes as `parity_census`."""
import numpy as np

F = np.float32
ONE = F(1.0)


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F)


# ---- the three operations, each with its alternatives --------------------------------------------------------------------------
def dot4(mode, px, py, pz, pw, cx, cy, cz):
    """plane (px,py,pz,pw) . (cx,cy,cz,1)"""
    if mode == "pairwise":
        return (px * cx + pz * cz) + (py * cy + pw * ONE)
    if mode == "ltr":
        return ((px * cx + py * cy) + pz * cz) + pw * ONE
    if mode == "fma_xyzw":  # ((x x' fused-into y y') ...) accumulate from x
        return fma(np.broadcast_to(pw, cx.shape).astype(F), np.full_like(cx, ONE), fma(np.broadcast_to(pz, cx.shape).astype(F), cz, fma(np.broadcast_to(py, cx.shape).astype(F), cy, px * cx)))
    if mode == "fma_wzyx":
        return fma(np.broadcast_to(px, cx.shape).astype(F), cx, fma(np.broadcast_to(py, cx.shape).astype(F), cy, fma(np.broadcast_to(pz, cx.shape).astype(F), cz, np.broadcast_to(pw * ONE, cx.shape).astype(F))))
    raise ValueError(mode)


def dot3(mode, ax, ay, az, bx, by, bz):
    if mode == "xy_z":
        return (ax * bx + ay * by) + az * bz
    if mode == "x_yz":
        return ax * bx + (ay * by + az * bz)
    if mode == "fma":
        return fma(az, bz, fma(ay, by, ax * bx))
    raise ValueError(mode)


def mul_vec3(mode, m, vx, vy, vz):
    """m = 9 columns (x_axis xyz, y_axis xyz, z_axis xyz) -> 3 components"""
    out = []
    for k in range(3):
        X, Y, Z = m[k], m[3 + k], m[6 + k]
        if mode == "seq":
            out.append((X * vx + Y * vy) + Z * vz)
        elif mode == "fma":
            out.append(fma(Z, vz, fma(Y, vy, X * vx)))
        else:
            raise ValueError(mode)
    return out


USED = dict(dot4="pairwise", dot3="xy_z", mul_vec3="seq")
ALTERNATIVES = [("dot4", "ltr"), ("dot4", "fma_xyzw"), ("dot4", "fma_wzyx"), ("dot3", "x_yz"), ("dot3", "fma"), ("mul_vec3", "fma")]


def global_transform(t, r, s):
    """Affine3A::from_scale_rotation_translation (no dot products, no matrix-vector products: nothing to substitute).  -> 12 arrays."""
    t, r, s = np.asarray(t, F).reshape(-1, 3), np.asarray(r, F).reshape(-1, 4), np.asarray(s, F).reshape(-1, 3)
    x, y, z, w = r[:, 0], r[:, 1], r[:, 2], r[:, 3]
    x2, y2, z2 = x + x, y + y, z + z
    xx, xy, xz, yy, yz, zz, wx, wy, wz = x * x2, x * y2, x * z2, y * y2, y * z2, z * z2, w * x2, w * y2, w * z2
    sx, sy, sz = s[:, 0], s[:, 1], s[:, 2]
    m = [(ONE - (yy + zz)) * sx, (xy + wz) * sx, (xz - wy) * sx, (xy - wz) * sy, (ONE - (xx + zz)) * sy, (yz + wx) * sy,
         (xz + wy) * sz, (yz - wx) * sz, (ONE - (xx + yy)) * sz]
    return m + [t[:, 0], t[:, 1], t[:, 2]]


def visibility(g, center, half, flags, layers, frusta, orders, view_kinds=None, margins=None):
    """check_visibility_cpu_culling's per-entity closure (visibility/mod.rs:800-846; shadow views: bevy_light/src/lib.rs:425-475)
    for every view.  view_kinds[v]: 0 camera (sphere pre-test + OBB over planes 0..4), 1 cascade (OBB only, planes 0..3 and 5).
    Returns uint8 [views][rows].  margins: a list that receives (|value| / ulp of its largest term) of every deciding plane test."""
    c = np.asarray(center, F).reshape(-1, 3)
    h = np.asarray(half, F).reshape(-1, 3)
    fl = np.asarray(flags)
    n = len(fl)
    has_aabb, has_sphere = (fl & 0x04) != 0, (fl & 0x08) != 0
    m, tx, ty, tz = g[:9], g[9], g[10], g[11]
    mc = mul_vec3(orders["mul_vec3"], m, c[:, 0], c[:, 1], c[:, 2])
    cw = [np.where(has_aabb, mc[0] + tx, c[:, 0]), np.where(has_aabb, mc[1] + ty, c[:, 1]), np.where(has_aabb, mc[2] + tz, c[:, 2])]
    mh = mul_vec3(orders["mul_vec3"], m, h[:, 0], h[:, 1], h[:, 2])
    sr = np.where(has_aabb, np.sqrt(dot3(orders["dot3"], mh[0], mh[1], mh[2], mh[0], mh[1], mh[2])), h[:, 0])
    fr = np.asarray(frusta, F).reshape(-1, 6, 4)
    out = np.zeros((len(fr), n), np.uint8)
    base = ((fl & 0x01) != 0) & ((fl & 0x10) == 0)
    bounded = (has_aabb | has_sphere) & ((fl & 0x02) == 0)
    lay = np.asarray(layers)
    for v in range(len(fr)):
        kind = 0 if view_kinds is None else view_kinds[v]
        vis = base & ((lay & 1) != 0)
        if kind == 1:
            vis = vis & ((fl & 0x80) != 0)
        inside = np.ones(n, bool)
        planes = (0, 1, 2, 3, 4) if kind == 0 else (0, 1, 2, 3, 5)
        for p in planes:
            px, py, pz, pw = (F(q) for q in fr[v, p])
            d = dot4(orders["dot4"], px, py, pz, pw, cw[0], cw[1], cw[2])
            if kind == 0:  # intersects_sphere
                val = d + sr
                inside &= ~(val <= F(0.0))
                if margins is not None:
                    big = np.maximum.reduce([np.abs(px * cw[0]), np.abs(py * cw[1]), np.abs(pz * cw[2]), np.full(n, abs(pw), F), np.abs(sr)])
                    margins.append((np.abs(val) / np.spacing(big))[bounded & vis])
            # intersects_obb: relative_radius = |(n.X, n.Y, n.Z)| . half
            ax = np.abs(dot3(orders["dot3"], px, py, pz, m[0], m[1], m[2]))
            ay = np.abs(dot3(orders["dot3"], px, py, pz, m[3], m[4], m[5]))
            az = np.abs(dot3(orders["dot3"], px, py, pz, m[6], m[7], m[8]))
            rr = dot3(orders["dot3"], ax, ay, az, h[:, 0], h[:, 1], h[:, 2])
            val = d + rr
            inside &= ~(has_aabb & (val <= F(0.0)))
            if margins is not None:
                big = np.maximum.reduce([np.abs(px * cw[0]), np.abs(py * cw[1]), np.abs(pz * cw[2]), np.full(n, abs(pw), F), np.abs(rr)])
                margins.append((np.abs(val) / np.spacing(big))[bounded & vis & has_aabb])
        cull = bounded if kind == 0 else (has_aabb & ((fl & 0x02) == 0))
        out[v] = vis & (inside | ~cull)
    return out


def census(scene, frusta, view_kinds=None):
    """-> dict: rows, views, visible (per view, used orders), flips per alternative (rows whose flag differs in some view), and the
    histogram of deciding values by distance to zero in ulps of their largest term."""
    g = global_transform(scene["translation"], scene["rotation"], scene["scale"])
    a = (scene["aabb_center"], scene["aabb_half"], scene["flags"], scene["layers"], frusta)
    margins = []
    used = visibility(g, *a, USED, view_kinds, margins)
    out = {"rows": int(len(scene["flags"])), "views": int(len(used)), "visible": [int(x) for x in used.sum(axis=1)], "flips": {}}
    for op, mode in ALTERNATIVES:
        alt = visibility(g, *a, dict(USED, **{op: mode}), view_kinds)
        out["flips"][f"{op}={mode}"] = int(np.count_nonzero((alt != used).any(axis=0)))
    allm = np.concatenate(margins) if margins else np.zeros(0)
    out["deciding_values"] = int(allm.size)
    out["within_ulps"] = {str(k): int(np.count_nonzero(allm < k)) for k in (1, 4, 16, 64, 1024)}
    return out, used
