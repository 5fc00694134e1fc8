// glam_math.h -- glam 0.33.2 (SSE2 backend) arithmetic in the exact operation order Bevy's CPU
// systems execute it, usable from HIP device code and from the host helpers.
//
// Bit-exactness rules (DESIGN.md "Numerics"):
//   * every TU including this header is compiled with -ffp-contract=off: no FMA contraction, each
//     mul/add is one IEEE-754 binary32 rounding exactly like an SSE2 lane;
//   * sqrt and division are the correctly rounded forms (HIP's default
//     -fhip-fp32-correctly-rounded-divide-sqrt; spelled explicitly below on the device);
//   * f32 denormals are preserved (gfx950 default for HIP; x86 SSE default).
// Op orders restated (reference call sites in parentheses, paths under crates/):
//   Vec3A::dot  = (x*x' + y*y') + z*z'                         (bevy_camera/src/primitives.rs:112-118)
//   Vec4::dot   = (x*x' + z*z') + (y*y' + w*w')  [pairwise]    (bevy_camera/src/primitives.rs:263,289)
//   Mat3A*Vec3A = ((X*v.x) + Y*v.y) + Z*v.z                    (bevy_transform/.../global_transform.rs:253)
//   Affine3A*Affine3A: m3 = A.m3*B.m3 ; t = A.m3*B.t + A.t     (global_transform.rs:316)
//   Mat3A::from_quat / Affine3A::from_scale_rotation_translation (transform.rs:274)
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MI_HD __host__ __device__ __forceinline__
#else
#define MI_HD inline
#endif

namespace mi {

struct V3 {
    float x, y, z;
};
struct V4 {
    float x, y, z, w;
};
struct M3 {
    V3 x_axis, y_axis, z_axis;
};
struct Affine {
    M3 m;
    V3 t;
};
struct M4 {
    V4 c[4];
};

MI_HD float f_sqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __fsqrt_rn(x);
#else
    return sqrtf(x);
#endif
}
MI_HD float f_div(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __fdiv_rn(a, b);
#else
    return a / b;
#endif
}
MI_HD float f_abs(float x) { return fabsf(x); }
// _mm_min_ps / _mm_max_ps lane semantics (second operand wins on NaN / equality)
MI_HD float lane_min(float a, float b) { return a < b ? a : b; }
MI_HD float lane_max(float a, float b) { return a > b ? a : b; }
// Rust f32::min / f32::max (NaN operand ignored)
MI_HD float rust_min(float a, float b) { return fminf(a, b); }
MI_HD float rust_max(float a, float b) { return fmaxf(a, b); }

MI_HD V3 v3(float x, float y, float z) { return V3{x, y, z}; }
MI_HD V4 v4(float x, float y, float z, float w) { return V4{x, y, z, w}; }
MI_HD V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
MI_HD V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
MI_HD V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
MI_HD V3 neg(V3 a) { return V3{-a.x, -a.y, -a.z}; }
MI_HD V3 abs3(V3 a) { return V3{f_abs(a.x), f_abs(a.y), f_abs(a.z)}; }
MI_HD V3 min3(V3 a, V3 b) { return V3{lane_min(a.x, b.x), lane_min(a.y, b.y), lane_min(a.z, b.z)}; }
MI_HD V3 max3(V3 a, V3 b) { return V3{lane_max(a.x, b.x), lane_max(a.y, b.y), lane_max(a.z, b.z)}; }
MI_HD float dot3(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
MI_HD float length3(V3 a) { return f_sqrt(dot3(a, a)); }
MI_HD V3 cross3(V3 a, V3 b) { return V3{a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }

MI_HD V4 operator+(V4 a, V4 b) { return V4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
MI_HD V4 operator-(V4 a, V4 b) { return V4{a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
MI_HD V4 operator*(V4 a, float s) { return V4{a.x * s, a.y * s, a.z * s, a.w * s}; }
MI_HD V4 mul4(V4 a, V4 b) { return V4{a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
MI_HD float dot4(V4 a, V4 b) { return (a.x * b.x + a.z * b.z) + (a.y * b.y + a.w * b.w); }
MI_HD V4 extend(V3 a, float w) { return V4{a.x, a.y, a.z, w}; }
MI_HD V3 xyz(V4 a) { return V3{a.x, a.y, a.z}; }

MI_HD V3 mul(const M3& m, V3 v) {
    V3 r = m.x_axis * v.x;
    r = r + m.y_axis * v.y;
    r = r + m.z_axis * v.z;
    return r;
}
MI_HD M3 mul(const M3& a, const M3& b) { return M3{mul(a, b.x_axis), mul(a, b.y_axis), mul(a, b.z_axis)}; }

// Mat3A::from_quat (rotation as x,y,z,w)
MI_HD M3 m3_from_quat(float x, float y, float z, float w) {
    float x2 = x + x, y2 = y + y, z2 = z + z;
    float xx = x * x2, xy = x * y2, xz = x * z2;
    float yy = y * y2, yz = y * z2, zz = z * z2;
    float wx = w * x2, wy = w * y2, wz = w * z2;
    M3 r;
    r.x_axis = V3{1.0f - (yy + zz), xy + wz, xz - wy};
    r.y_axis = V3{xy - wz, 1.0f - (xx + zz), yz + wx};
    r.z_axis = V3{xz + wy, yz - wx, 1.0f - (xx + yy)};
    return r;
}
// Transform::compute_affine = Affine3A::from_scale_rotation_translation(scale, rotation, translation)
MI_HD Affine affine_from_srt(V3 s, V4 q, V3 t) {
    M3 rot = m3_from_quat(q.x, q.y, q.z, q.w);
    Affine a;
    a.m.x_axis = rot.x_axis * s.x;
    a.m.y_axis = rot.y_axis * s.y;
    a.m.z_axis = rot.z_axis * s.z;
    a.t = t;
    return a;
}
// GlobalTransform::mul_transform: self.0 * transform.compute_affine()
MI_HD Affine mul(const Affine& a, const Affine& b) {
    Affine r;
    r.m = mul(a.m, b.m);
    r.t = mul(a.m, b.t) + a.t;
    return r;
}
MI_HD V3 transform_point(const Affine& a, V3 p) { return mul(a.m, p) + a.t; }
MI_HD float determinant(const M3& m) { return dot3(m.z_axis, cross3(m.x_axis, m.y_axis)); }
MI_HD M3 inverse(const M3& m) {
    V3 tmp0 = cross3(m.y_axis, m.z_axis);
    V3 tmp1 = cross3(m.z_axis, m.x_axis);
    V3 tmp2 = cross3(m.x_axis, m.y_axis);
    float det = dot3(m.z_axis, tmp2);
    float inv = f_div(1.0f, det);
    V3 c0 = tmp0 * inv, c1 = tmp1 * inv, c2 = tmp2 * inv;
    return M3{V3{c0.x, c1.x, c2.x}, V3{c0.y, c1.y, c2.y}, V3{c0.z, c1.z, c2.z}};
}
MI_HD Affine inverse(const Affine& a) {
    Affine r;
    r.m = inverse(a.m);
    r.t = neg(mul(r.m, a.t));
    return r;
}

MI_HD Affine load_affine(const float* g) {
    Affine a;
    a.m.x_axis = V3{g[0], g[1], g[2]};
    a.m.y_axis = V3{g[3], g[4], g[5]};
    a.m.z_axis = V3{g[6], g[7], g[8]};
    a.t = V3{g[9], g[10], g[11]};
    return a;
}
MI_HD void store_affine(const Affine& a, float* g) {
    g[0] = a.m.x_axis.x; g[1] = a.m.x_axis.y; g[2] = a.m.x_axis.z;
    g[3] = a.m.y_axis.x; g[4] = a.m.y_axis.y; g[5] = a.m.y_axis.z;
    g[6] = a.m.z_axis.x; g[7] = a.m.z_axis.y; g[8] = a.m.z_axis.z;
    g[9] = a.t.x; g[10] = a.t.y; g[11] = a.t.z;
}
// PartialEq for GlobalTransform: 12 float == comparisons, NaN != NaN (Vec3A w lanes excluded)
MI_HD bool affine_eq(const Affine& a, const Affine& b) {
    return a.m.x_axis.x == b.m.x_axis.x && a.m.x_axis.y == b.m.x_axis.y && a.m.x_axis.z == b.m.x_axis.z &&
           a.m.y_axis.x == b.m.y_axis.x && a.m.y_axis.y == b.m.y_axis.y && a.m.y_axis.z == b.m.y_axis.z &&
           a.m.z_axis.x == b.m.z_axis.x && a.m.z_axis.y == b.m.z_axis.y && a.m.z_axis.z == b.m.z_axis.z &&
           a.t.x == b.t.x && a.t.y == b.t.y && a.t.z == b.t.z;
}

MI_HD V4 mul(const M4& m, V4 v) {
    V4 r = m.c[0] * v.x;
    r = r + m.c[1] * v.y;
    r = r + m.c[2] * v.z;
    r = r + m.c[3] * v.w;
    return r;
}
MI_HD M4 mul(const M4& a, const M4& b) { return M4{{mul(a, b.c[0]), mul(a, b.c[1]), mul(a, b.c[2]), mul(a, b.c[3])}}; }
MI_HD V4 row(const M4& m, int i) {
    const float* c0 = &m.c[0].x; const float* c1 = &m.c[1].x;
    const float* c2 = &m.c[2].x; const float* c3 = &m.c[3].x;
    return V4{c0[i], c1[i], c2[i], c3[i]};
}
MI_HD M4 m4_from_affine(const Affine& a) {
    return M4{{extend(a.m.x_axis, 0.0f), extend(a.m.y_axis, 0.0f), extend(a.m.z_axis, 0.0f), extend(a.t, 1.0f)}};
}
MI_HD M4 load_m4(const float* f) {
    M4 m;
    for (int i = 0; i < 4; ++i) m.c[i] = V4{f[4 * i], f[4 * i + 1], f[4 * i + 2], f[4 * i + 3]};
    return m;
}
MI_HD void store_m4(const M4& m, float* f) {
    for (int i = 0; i < 4; ++i) { f[4 * i] = m.c[i].x; f[4 * i + 1] = m.c[i].y; f[4 * i + 2] = m.c[i].z; f[4 * i + 3] = m.c[i].w; }
}

// HalfSpace::new (bevy_math/src/primitives/half_space.rs:53-57)
MI_HD V4 half_space_new(V4 nd) {
    float len = f_sqrt((nd.x * nd.x + nd.y * nd.y) + nd.z * nd.z);
    return nd * f_div(1.0f, len);
}

// Frustum::intersects_sphere (bevy_camera/src/primitives.rs:255-268); planes = 6 x V4
MI_HD bool frustum_intersects_sphere(const V4* planes, V3 center, float radius, bool intersect_far) {
    V4 c = extend(center, 1.0f);
    const int max = intersect_far ? 5 : 4;
    bool inside = true;
    for (int i = 0; i <= max; ++i) inside = inside && !(dot4(planes[i], c) + radius <= 0.0f);
    return inside;
}
// Aabb::relative_radius (primitives.rs:109-119)
MI_HD float aabb_relative_radius(V3 half_extents, V3 p_normal, const M3& world_from_local) {
    V3 v = V3{dot3(p_normal, world_from_local.x_axis), dot3(p_normal, world_from_local.y_axis),
              dot3(p_normal, world_from_local.z_axis)};
    return dot3(abs3(v), half_extents);
}
// Frustum::intersects_obb (primitives.rs:272-294)
MI_HD bool frustum_intersects_obb(const V4* planes, V3 center, V3 half_extents, const Affine& world_from_local,
                                  bool intersect_near, bool intersect_far) {
    V4 c = extend(transform_point(world_from_local, center), 1.0f);
    bool inside = true;
    for (int idx = 0; idx < 6; ++idx) {
        if ((idx == 4 && !intersect_near) || (idx == 5 && !intersect_far)) continue;
        float rr = aabb_relative_radius(half_extents, xyz(planes[idx]), world_from_local.m);
        inside = inside && !(dot4(planes[idx], c) + rr <= 0.0f);
    }
    return inside;
}

// Sphere::intersects_obb (bevy_camera/src/primitives.rs:219-226)
MI_HD bool sphere_intersects_obb(V3 sphere_center, float sphere_radius, V3 aabb_center_world, V3 half_extents,
                                 const M3& world_from_local) {
    const V3 v = aabb_center_world - sphere_center;
    const float d_sq = dot3(v, v);
    const float d = f_sqrt(d_sq);
    const float rr = aabb_relative_radius(half_extents, v, world_from_local);
    return d_sq <= sphere_radius * d + rr;
}

// Rust `f32 as u32` (saturating, NaN -> 0)
MI_HD uint32_t f32_as_u32(float f) {
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}

}  // namespace mi
