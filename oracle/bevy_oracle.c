/*
 * bevy_oracle.c -- CPU ORACLE (test infrastructure; see bevy_oracle.h for the rules and the
 * parity-pinning status).  Plain C99, scalar f32, built with
 *     gcc -O2 -std=c99 -ffp-contract=off -fno-fast-math -msse2 -mfpmath=sse
 * so that every operation is an individually rounded IEEE-754 binary32 op, exactly like the
 * SSE2 lanes glam 0.33.2 uses on x86-64 (bevy does not enable glam's `scalar-math` or
 * `fast-math` features: crates/bevy_math/Cargo.toml:13).
 *
 * Each function cites the reference lines it restates (paths relative to /root/reference).
 */
#define _GNU_SOURCE
#include "bevy_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ======================================================================================= */
/* glam 0.33.2 (SSE2 backend) arithmetic, restated                                          */
/* ======================================================================================= */

typedef struct { float x, y, z; } v3;
typedef struct { float x, y, z, w; } v4;
typedef struct { v3 x_axis, y_axis, z_axis; } m3;
typedef struct { m3 m; v3 t; } aff;
typedef struct { v4 c[4]; } m4;

static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v4 V4(float x, float y, float z, float w) { v4 r = {x, y, z, w}; return r; }
static inline v3 v3_add(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_mul(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v3_scale(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 v3_abs(v3 a) { return V3(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
static inline v3 v3_neg(v3 a) { return V3(-a.x, -a.y, -a.z); }
/* glam min/max lanes: _mm_min_ps(a,b) = a < b ? a : b ; _mm_max_ps(a,b) = a > b ? a : b */
static inline float lane_min(float a, float b) { return a < b ? a : b; }
static inline float lane_max(float a, float b) { return a > b ? a : b; }
static inline v3 v3_min(v3 a, v3 b) { return V3(lane_min(a.x, b.x), lane_min(a.y, b.y), lane_min(a.z, b.z)); }
static inline v3 v3_max(v3 a, v3 b) { return V3(lane_max(a.x, b.x), lane_max(a.y, b.y), lane_max(a.z, b.z)); }
/* Rust f32::min / f32::max (IEEE minNum/maxNum: a NaN operand is ignored) */
static inline float rust_min(float a, float b) { return fminf(a, b); }
static inline float rust_max(float a, float b) { return fmaxf(a, b); }

/* PARITY-MARGIN SWITCHES (tests/test_parity_margin.py; never defined in the oracle proper).  glam 0.33.2 is not under
 * /root/reference and cannot be built here, so the lane orders below are restated from memory; each ORC_* macro swaps ONE of them
 * for the order a wrong memory would give, and the test counts what that changes on the BASELINE configs:
 *   ORC_DOT4_LEFT_TO_RIGHT  Vec4::dot as ((x x' + y y') + z z') + w w'
 *   ORC_DOT3_PAIRWISE       Vec3A::dot as (x x' + z z') + y y'   (dot4_in_x's shuffle pattern with a zero w lane)
 *   ORC_DOT3_X_YZ           Vec3A::dot as x x' + (y y' + z z')
 *   ORC_LENGTH_RSQRT        Vec3A::length as 1 / (1 / sqrt(dot))   (length through length_recip)
 *   ORC_MAT3_COLUMNS_ZYX    Mat3A * Vec3A accumulated from the z column: (Z v.z + Y v.y) + X v.x
 * and the build flags -ffp-contract=fast -mfma fuse every a*b+c the compiler can (variant "fma"). */
/* Vec3A::dot, glam sse2 dot3_in_x: (x*x' + y*y') + z*z'.  The scalar Vec3::dot has the same
 * left-to-right order. */
#if defined(ORC_DOT3_PAIRWISE)
static inline float v3_dot(v3 a, v3 b) { return (a.x * b.x + a.z * b.z) + a.y * b.y; }
#elif defined(ORC_DOT3_X_YZ)
static inline float v3_dot(v3 a, v3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
#else
static inline float v3_dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
#endif
/* Vec4::dot, glam sse2 dot4_in_x: (x*x' + z*z') + (y*y' + w*w') -- pairwise, NOT left-to-right. */
#if defined(ORC_DOT4_LEFT_TO_RIGHT)
static inline float v4_dot(v4 a, v4 b) { return ((a.x * b.x + a.y * b.y) + a.z * b.z) + a.w * b.w; }
#else
static inline float v4_dot(v4 a, v4 b) { return (a.x * b.x + a.z * b.z) + (a.y * b.y + a.w * b.w); }
#endif
#if defined(ORC_LENGTH_RSQRT)
static inline float v3_length(v3 a) { float d = v3_dot(a, a); return d > 0.0f ? 1.0f / (1.0f / sqrtf(d)) : sqrtf(d); }
#else
static inline float v3_length(v3 a) { return sqrtf(v3_dot(a, a)); }
#endif
/* Vec3A::cross (sse2): (a.zxy*b - a*b.zxy).zxy */
static inline v3 v3_cross(v3 a, v3 b) {
    return V3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
static inline v4 v4_scale(v4 a, float s) { return V4(a.x * s, a.y * s, a.z * s, a.w * s); }
static inline v4 v4_add(v4 a, v4 b) { return V4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
static inline v4 v4_sub(v4 a, v4 b) { return V4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
static inline v4 v4_mul(v4 a, v4 b) { return V4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
static inline v4 v3_extend(v3 a, float w) { return V4(a.x, a.y, a.z, w); }
static inline v3 v4_xyz(v4 a) { return V3(a.x, a.y, a.z); }

/* Mat3A::mul_vec3a (sse2): r = X*v.x; r += Y*v.y; r += Z*v.z */
static inline v3 m3_mul_v3(const m3* m, v3 v) {
#if defined(ORC_MAT3_COLUMNS_ZYX)
    v3 r = v3_scale(m->z_axis, v.z);
    r = v3_add(r, v3_scale(m->y_axis, v.y));
    r = v3_add(r, v3_scale(m->x_axis, v.x));
    return r;
#else
    v3 r = v3_scale(m->x_axis, v.x);
    r = v3_add(r, v3_scale(m->y_axis, v.y));
    r = v3_add(r, v3_scale(m->z_axis, v.z));
    return r;
#endif
}
/* Mat3A * Mat3A: columns (A*B.x, A*B.y, A*B.z) */
static inline m3 m3_mul(const m3* a, const m3* b) {
    m3 r;
    r.x_axis = m3_mul_v3(a, b->x_axis);
    r.y_axis = m3_mul_v3(a, b->y_axis);
    r.z_axis = m3_mul_v3(a, b->z_axis);
    return r;
}
/* Mat3A::from_quat */
static inline m3 m3_from_quat(const float q[4]) {
    float x = q[0], y = q[1], z = q[2], w = q[3];
    float x2 = x + x, y2 = y + y, z2 = z + z;
    float xx = x * x2, xy = x * y2, xz = x * z2;
    float yy = y * y2, yz = y * z2, zz = z * z2;
    float wx = w * x2, wy = w * y2, wz = w * z2;
    m3 r;
    r.x_axis = V3(1.0f - (yy + zz), xy + wz, xz - wy);
    r.y_axis = V3(xy - wz, 1.0f - (xx + zz), yz + wx);
    r.z_axis = V3(xz + wy, yz - wx, 1.0f - (xx + yy));
    return r;
}
/* Affine3A::from_scale_rotation_translation */
static inline aff aff_from_srt(const float s[3], const float q[4], const float t[3]) {
    m3 rot = m3_from_quat(q);
    aff a;
    a.m.x_axis = v3_scale(rot.x_axis, s[0]);
    a.m.y_axis = v3_scale(rot.y_axis, s[1]);
    a.m.z_axis = v3_scale(rot.z_axis, s[2]);
    a.t = V3(t[0], t[1], t[2]);
    return a;
}
/* Affine3A * Affine3A: m3 = A.m3*B.m3 ; t = A.m3*B.t + A.t */
static inline aff aff_mul(const aff* a, const aff* b) {
    aff r;
    r.m = m3_mul(&a->m, &b->m);
    r.t = v3_add(m3_mul_v3(&a->m, b->t), a->t);
    return r;
}
/* Affine3A::transform_point3a */
static inline v3 aff_point(const aff* a, v3 p) { return v3_add(m3_mul_v3(&a->m, p), a->t); }
static inline float m3_determinant(const m3* m) { return v3_dot(m->z_axis, v3_cross(m->x_axis, m->y_axis)); }
/* Mat3A::inverse (sse2) */
static inline m3 m3_inverse(const m3* m) {
    v3 tmp0 = v3_cross(m->y_axis, m->z_axis);
    v3 tmp1 = v3_cross(m->z_axis, m->x_axis);
    v3 tmp2 = v3_cross(m->x_axis, m->y_axis);
    float det = v3_dot(m->z_axis, tmp2);
    float inv = 1.0f / det;
    v3 c0 = v3_scale(tmp0, inv), c1 = v3_scale(tmp1, inv), c2 = v3_scale(tmp2, inv);
    m3 r; /* transpose */
    r.x_axis = V3(c0.x, c1.x, c2.x);
    r.y_axis = V3(c0.y, c1.y, c2.y);
    r.z_axis = V3(c0.z, c1.z, c2.z);
    return r;
}
/* Affine3A::inverse */
static inline aff aff_inverse(const aff* a) {
    aff r;
    r.m = m3_inverse(&a->m);
    r.t = v3_neg(m3_mul_v3(&r.m, a->t));
    return r;
}
static inline aff aff_load(const float g[12]) {
    aff a;
    a.m.x_axis = V3(g[0], g[1], g[2]);
    a.m.y_axis = V3(g[3], g[4], g[5]);
    a.m.z_axis = V3(g[6], g[7], g[8]);
    a.t = V3(g[9], g[10], g[11]);
    return a;
}
static inline void aff_store(const aff* a, float g[12]) {
    g[0] = a->m.x_axis.x; g[1] = a->m.x_axis.y; g[2] = a->m.x_axis.z;
    g[3] = a->m.y_axis.x; g[4] = a->m.y_axis.y; g[5] = a->m.y_axis.z;
    g[6] = a->m.z_axis.x; g[7] = a->m.z_axis.y; g[8] = a->m.z_axis.z;
    g[9] = a->t.x; g[10] = a->t.y; g[11] = a->t.z;
}
/* PartialEq for Affine3A: 12 lanes of float == (NaN != NaN); Vec3A w lanes are not compared */
static inline int aff_eq(const float a[12], const float b[12]) {
    for (int i = 0; i < 12; ++i) if (!(a[i] == b[i])) return 0;
    return 1;
}

/* Mat4::mul_vec4 (sse2): sequential column accumulate */
static inline v4 m4_mul_v4(const m4* m, v4 v) {
    v4 r = v4_scale(m->c[0], v.x);
    r = v4_add(r, v4_scale(m->c[1], v.y));
    r = v4_add(r, v4_scale(m->c[2], v.z));
    r = v4_add(r, v4_scale(m->c[3], v.w));
    return r;
}
static inline m4 m4_mul(const m4* a, const m4* b) {
    m4 r;
    for (int i = 0; i < 4; ++i) r.c[i] = m4_mul_v4(a, b->c[i]);
    return r;
}
static inline v4 m4_row(const m4* m, int i) {
    const float* c0 = &m->c[0].x; const float* c1 = &m->c[1].x;
    const float* c2 = &m->c[2].x; const float* c3 = &m->c[3].x;
    return V4(c0[i], c1[i], c2[i], c3[i]);
}
static inline m4 m4_from_affine(const aff* a) {
    m4 r;
    r.c[0] = v3_extend(a->m.x_axis, 0.0f);
    r.c[1] = v3_extend(a->m.y_axis, 0.0f);
    r.c[2] = v3_extend(a->m.z_axis, 0.0f);
    r.c[3] = v3_extend(a->t, 1.0f);
    return r;
}
static inline m4 m4_load(const float f[16]) { m4 r; memcpy(&r, f, sizeof r); return r; }
static inline void m4_store(const m4* m, float f[16]) { memcpy(f, m, sizeof *m); }

/* Mat4::inverse -- glam's cofactor scheme (GLM-derived); same structure in the sse2 backend. */
static m4 m4_inverse(const m4* s) {
    float m00 = s->c[0].x, m01 = s->c[0].y, m02 = s->c[0].z, m03 = s->c[0].w;
    float m10 = s->c[1].x, m11 = s->c[1].y, m12 = s->c[1].z, m13 = s->c[1].w;
    float m20 = s->c[2].x, m21 = s->c[2].y, m22 = s->c[2].z, m23 = s->c[2].w;
    float m30 = s->c[3].x, m31 = s->c[3].y, m32 = s->c[3].z, m33 = s->c[3].w;

    float coef00 = m22 * m33 - m32 * m23, coef02 = m12 * m33 - m32 * m13, coef03 = m12 * m23 - m22 * m13;
    float coef04 = m21 * m33 - m31 * m23, coef06 = m11 * m33 - m31 * m13, coef07 = m11 * m23 - m21 * m13;
    float coef08 = m21 * m32 - m31 * m22, coef10 = m11 * m32 - m31 * m12, coef11 = m11 * m22 - m21 * m12;
    float coef12 = m20 * m33 - m30 * m23, coef14 = m10 * m33 - m30 * m13, coef15 = m10 * m23 - m20 * m13;
    float coef16 = m20 * m32 - m30 * m22, coef18 = m10 * m32 - m30 * m12, coef19 = m10 * m22 - m20 * m12;
    float coef20 = m20 * m31 - m30 * m21, coef22 = m10 * m31 - m30 * m11, coef23 = m10 * m21 - m20 * m11;

    v4 fac0 = V4(coef00, coef00, coef02, coef03), fac1 = V4(coef04, coef04, coef06, coef07);
    v4 fac2 = V4(coef08, coef08, coef10, coef11), fac3 = V4(coef12, coef12, coef14, coef15);
    v4 fac4 = V4(coef16, coef16, coef18, coef19), fac5 = V4(coef20, coef20, coef22, coef23);

    v4 vec0 = V4(m10, m00, m00, m00), vec1 = V4(m11, m01, m01, m01);
    v4 vec2 = V4(m12, m02, m02, m02), vec3 = V4(m13, m03, m03, m03);

    v4 inv0 = v4_add(v4_sub(v4_mul(vec1, fac0), v4_mul(vec2, fac1)), v4_mul(vec3, fac2));
    v4 inv1 = v4_add(v4_sub(v4_mul(vec0, fac0), v4_mul(vec2, fac3)), v4_mul(vec3, fac4));
    v4 inv2 = v4_add(v4_sub(v4_mul(vec0, fac1), v4_mul(vec1, fac3)), v4_mul(vec3, fac5));
    v4 inv3 = v4_add(v4_sub(v4_mul(vec0, fac2), v4_mul(vec1, fac4)), v4_mul(vec2, fac5));

    v4 sign_a = V4(1.0f, -1.0f, 1.0f, -1.0f), sign_b = V4(-1.0f, 1.0f, -1.0f, 1.0f);
    m4 inv;
    inv.c[0] = v4_mul(inv0, sign_a);
    inv.c[1] = v4_mul(inv1, sign_b);
    inv.c[2] = v4_mul(inv2, sign_a);
    inv.c[3] = v4_mul(inv3, sign_b);

    v4 col0 = V4(inv.c[0].x, inv.c[1].x, inv.c[2].x, inv.c[3].x);
    v4 dot0 = v4_mul(s->c[0], col0);
    float dot1 = (dot0.x + dot0.y) + (dot0.z + dot0.w);
    float rcp_det = 1.0f / dot1;
    for (int i = 0; i < 4; ++i) inv.c[i] = v4_scale(inv.c[i], rcp_det);
    return inv;
}

/* ---- exported primitive wrappers -------------------------------------------------------- */

void orc_transform_to_affine(const float t[3], const float r[4], const float s[3], float out[12]) {
    aff a = aff_from_srt(s, r, t);
    aff_store(&a, out);
}
void orc_affine_mul(const float a[12], const float b[12], float out[12]) {
    aff A = aff_load(a), B = aff_load(b), R = aff_mul(&A, &B);
    aff_store(&R, out);
}
void orc_affine_inverse(const float a[12], float out[12]) {
    aff A = aff_load(a), R = aff_inverse(&A);
    aff_store(&R, out);
}
void orc_affine_transform_point(const float a[12], const float p[3], float out[3]) {
    aff A = aff_load(a);
    v3 r = aff_point(&A, V3(p[0], p[1], p[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
/* (tests/test_parity_margin.py: the raw lane orders, so that the test can see that a variant build really swapped one) */
void orc_probe_lane_orders(const float a[4], const float b[4], float out[3]) {
    out[0] = v4_dot(V4(a[0], a[1], a[2], a[3]), V4(b[0], b[1], b[2], b[3]));
    out[1] = v3_dot(V3(a[0], a[1], a[2]), V3(b[0], b[1], b[2]));
    out[2] = v3_length(V3(a[0], a[1], a[2]));
}
float orc_radius_vec3a(const float a[12], const float e[3]) {
    aff A = aff_load(a);
    return v3_length(m3_mul_v3(&A.m, V3(e[0], e[1], e[2])));
}
/* HalfSpace::new: normal_d * normal_d.xyz().length_recip();  Vec3::length_recip = length().recip() */
static inline v4 half_space_new(v4 nd) {
    float len = sqrtf((nd.x * nd.x + nd.y * nd.y) + nd.z * nd.z);
    float recip = 1.0f / len;
    return v4_scale(nd, recip);
}
void orc_half_space_new(const float nd[4], float out[4]) {
    v4 r = half_space_new(V4(nd[0], nd[1], nd[2], nd[3]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
void orc_mat4_inverse(const float m[16], float out[16]) { m4 M = m4_load(m), R = m4_inverse(&M); m4_store(&R, out); }
void orc_mat4_mul(const float a[16], const float b[16], float out[16]) {
    m4 A = m4_load(a), B = m4_load(b), R = m4_mul(&A, &B);
    m4_store(&R, out);
}
void orc_mat4_mul_vec4(const float m[16], const float v[4], float out[4]) {
    m4 M = m4_load(m);
    v4 r = m4_mul_v4(&M, V4(v[0], v[1], v[2], v[3]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
/* perspective_infinite_reverse (RH, depth 0..1 reversed): f = 1/tan(fov/2) via sin_cos */
void orc_perspective_infinite_reverse(float fov, float aspect, float near, float out[16]) {
    float s = sinf(0.5f * fov), c = cosf(0.5f * fov);
    float h = c / s;
    float w = h / aspect;
    memset(out, 0, 16 * sizeof(float));
    out[0] = w; out[5] = h; out[11] = -1.0f; out[14] = near;
}

/* ViewFrustum::from_clip_from_world_no_far, view_frustum.rs:91-108 */
static void frustum_no_far(const m4* cfw, float out[24]) {
    v4 row0 = m4_row(cfw, 0), row1 = m4_row(cfw, 1), row2 = m4_row(cfw, 2), row3 = m4_row(cfw, 3);
    v4 hs[6];
    hs[0] = half_space_new(v4_add(row3, row0));
    hs[1] = half_space_new(v4_sub(row3, row0));
    hs[2] = half_space_new(v4_add(row3, row1));
    hs[3] = half_space_new(v4_sub(row3, row1));
    hs[4] = half_space_new(v4_add(row3, row2));
    /* INACTIVE_HALF_SPACE = (0,0,0,inf) through HalfSpace::new: 0*(1/0)=NaN normal, inf*inf=inf */
    hs[5] = half_space_new(V4(0.0f, 0.0f, 0.0f, INFINITY));
    memcpy(out, hs, sizeof hs);
}
void orc_frustum_from_clip_from_world(const float clip_from_world[16], float out[24]) {
    m4 cfw = m4_load(clip_from_world);
    frustum_no_far(&cfw, out);
    v4 far = half_space_new(m4_row(&cfw, 2));
    memcpy(out + 20, &far, sizeof far);
}
/* CameraProjection::compute_frustum (projection.rs:72-80) with
 * ViewFrustum::from_clip_from_world_custom_far (view_frustum.rs:52-64) */
void orc_compute_frustum_perspective(float fov, float aspect, float near, float far,
                                     const float camera_affine[12], float out[24]) {
    float proj[16];
    orc_perspective_infinite_reverse(fov, aspect, near, proj);
    m4 P = m4_load(proj);
    aff cam = aff_load(camera_affine);
    aff inv = aff_inverse(&cam);
    m4 V = m4_from_affine(&inv);
    m4 cfw = m4_mul(&P, &V);
    frustum_no_far(&cfw, out);
    /* camera_transform.translation(), .back() = (matrix3 * Vec3::Z).normalize() */
    v3 view_translation = cam.t;
    v3 back = m3_mul_v3(&cam.m, V3(0.0f, 0.0f, 1.0f));
    float recip = 1.0f / sqrtf(v3_dot(back, back));
    back = v3_scale(back, recip);
    v3 far_center = v3_sub(view_translation, v3_scale(back, far)); /* far * view_backward */
    v4 hs = half_space_new(v3_extend(back, -v3_dot(back, far_center)));
    memcpy(out + 20, &hs, sizeof hs);
}

static inline v4 plane_at(const float* frustum, int i) {
    return V4(frustum[4 * i], frustum[4 * i + 1], frustum[4 * i + 2], frustum[4 * i + 3]);
}

/* Frustum::intersects_sphere, primitives.rs:255-268 */
static inline int frustum_intersects_sphere(const float* frustum, v3 center, float radius, int intersect_far) {
    v4 c = v3_extend(center, 1.0f);
    int max = intersect_far ? 5 : 4;
    for (int i = 0; i <= max; ++i) {
        if (v4_dot(plane_at(frustum, i), c) + radius <= 0.0f) return 0;
    }
    return 1;
}
/* Aabb::relative_radius, primitives.rs:109-119 */
static inline float aabb_relative_radius(v3 half_extents, v3 p_normal, const m3* world_from_local) {
    v3 v = V3(v3_dot(p_normal, world_from_local->x_axis), v3_dot(p_normal, world_from_local->y_axis),
              v3_dot(p_normal, world_from_local->z_axis));
    return v3_dot(v3_abs(v), half_extents);
}
/* Frustum::intersects_obb, primitives.rs:272-294 */
static inline int frustum_intersects_obb(const float* frustum, v3 center, v3 half, const aff* wfl,
                                         int intersect_near, int intersect_far) {
    v4 c = v3_extend(aff_point(wfl, center), 1.0f);
    for (int idx = 0; idx < 6; ++idx) {
        if ((idx == 4 && !intersect_near) || (idx == 5 && !intersect_far)) continue;
        v4 hs = plane_at(frustum, idx);
        float rr = aabb_relative_radius(half, v4_xyz(hs), &wfl->m);
        if (v4_dot(hs, c) + rr <= 0.0f) return 0;
    }
    return 1;
}
int orc_frustum_intersects_sphere(const float frustum[24], const float c[3], float radius, int intersect_far) {
    return frustum_intersects_sphere(frustum, V3(c[0], c[1], c[2]), radius, intersect_far);
}
int orc_frustum_intersects_obb(const float frustum[24], const float c[3], const float h[3],
                               const float wfl[12], int intersect_near, int intersect_far) {
    aff A = aff_load(wfl);
    return frustum_intersects_obb(frustum, V3(c[0], c[1], c[2]), V3(h[0], h[1], h[2]), &A, intersect_near,
                                  intersect_far);
}
int orc_frustum_intersects_obb_identity(const float frustum[24], const float c[3], const float h[3]) {
    v4 cw = V4(c[0], c[1], c[2], 1.0f);
    v3 he = v3_abs(V3(h[0], h[1], h[2]));
    for (int i = 0; i < 6; ++i) {
        v4 hs = plane_at(frustum, i);
        float rr = v3_dot(he, v3_abs(v4_xyz(hs)));
        if (v4_dot(hs, cw) + rr <= 0.0f) return 0;
    }
    return 1;
}
/* Aabb::is_in_half_space, primitives.rs:134-143.  Mat3A::abs() is lane-wise. */
int orc_aabb_is_in_half_space(const float c[3], const float h[3], const float hs[4], const float wfl[12]) {
    aff A = aff_load(wfl);
    m3 ma;
    ma.x_axis = v3_abs(A.m.x_axis); ma.y_axis = v3_abs(A.m.y_axis); ma.z_axis = v3_abs(A.m.z_axis);
    v3 hew = m3_mul_v3(&ma, v3_abs(V3(h[0], h[1], h[2])));
    v3 n = V3(hs[0], hs[1], hs[2]);
    float r = v3_dot(hew, v3_abs(n));
    v3 cw = aff_point(&A, V3(c[0], c[1], c[2]));
    float sd = v3_dot(n, cw) + hs[3];
    return sd > r;
}
int orc_aabb_is_in_half_space_identity(const float c[3], const float h[3], const float hs[4]) {
    v3 n = V3(hs[0], hs[1], hs[2]);
    float r = v3_dot(v3_abs(V3(h[0], h[1], h[2])), v3_abs(n));
    float sd = v3_dot(n, V3(c[0], c[1], c[2])) + hs[3];
    return sd > r;
}
int orc_frustum_contains_aabb(const float frustum[24], const float c[3], const float h[3], const float wfl[12]) {
    for (int i = 0; i < 6; ++i)
        if (!orc_aabb_is_in_half_space(c, h, frustum + 4 * i, wfl)) return 0;
    return 1;
}
/* Sphere::intersects_obb, primitives.rs:219-226 */
int orc_sphere_intersects_obb(const float sc[3], float sr, const float c[3], const float h[3], const float wfl[12]) {
    aff A = aff_load(wfl);
    v3 cw = aff_point(&A, V3(c[0], c[1], c[2]));
    v3 v = v3_sub(cw, V3(sc[0], sc[1], sc[2]));
    float d_sq = v3_dot(v, v);
    float d = sqrtf(d_sq);
    float rr = aabb_relative_radius(V3(h[0], h[1], h[2]), v, &A.m);
    return d_sq <= sr * d + rr;
}

/* ======================================================================================= */
/* transform propagation                                                                     */
/* ======================================================================================= */

void orc_sync_simple_transforms(uint32_t n, const float* t, const float* r, const float* s,
                                const uint8_t* dirty, float* global, uint8_t* changed_out) {
    for (uint32_t i = 0; i < n; ++i) {
        if (dirty && !dirty[i]) continue;
        aff a = aff_from_srt(s + 3 * i, r + 4 * i, t + 3 * i);
        aff_store(&a, global + 12 * i);
        if (changed_out) changed_out[i] = 1;
    }
}

void orc_mark_dirty_trees(uint32_t n, const uint32_t* parent, const uint8_t* changed, uint8_t* tree_changed) {
    /* serial flavour, systems.rs:136-152: climb until an already-marked node */
    for (uint32_t i = 0; i < n; ++i) {
        if (!changed[i]) continue;
        uint32_t next = i;
        for (;;) {
            if (tree_changed[next]) break;
            tree_changed[next] = 1;
            uint32_t p = parent[next];
            if (p == ORC_NO_PARENT || p >= n) break;
            next = p;
        }
    }
}

int orc_propagate_transforms(uint32_t n, const uint32_t* parent, const float* t, const float* r,
                             const float* s, int static_opt, const uint8_t* tree_changed,
                             const uint8_t* transform_changed, float* global, uint8_t* changed_out) {
    if (n == 0) return 0;
    /* children CSR in row order (Children is a Vec<Entity> in insertion order; row order here) */
    uint32_t* child_count = (uint32_t*)calloc((size_t)n + 1, sizeof(uint32_t));
    uint32_t* child_start = (uint32_t*)malloc(((size_t)n + 1) * sizeof(uint32_t));
    uint32_t* child_list = (uint32_t*)malloc((size_t)n * sizeof(uint32_t));
    uint32_t* stack = (uint32_t*)malloc((size_t)n * sizeof(uint32_t));
    uint8_t* g_changed = (uint8_t*)calloc(n, 1);
    uint8_t* visited = (uint8_t*)calloc(n, 1);
    int rc = 0;
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t p = parent[i];
        if (p == ORC_NO_PARENT) continue;
        if (p >= n || p == i) { rc = -1; goto done; }
        child_count[p]++;
    }
    child_start[0] = 0;
    for (uint32_t i = 0; i < n; ++i) child_start[i + 1] = child_start[i] + child_count[i];
    memset(child_count, 0, (size_t)n * sizeof(uint32_t));
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t p = parent[i];
        if (p == ORC_NO_PARENT) continue;
        child_list[child_start[p] + child_count[p]++] = i;
    }
    /* cycle check: every node must be reachable from a root */
    {
        uint32_t reached = 0, sp = 0;
        for (uint32_t i = 0; i < n; ++i) if (parent[i] == ORC_NO_PARENT) { stack[sp++] = i; }
        while (sp) {
            uint32_t v = stack[--sp];
            reached++;
            for (uint32_t k = child_start[v]; k < child_start[v + 1]; ++k) stack[sp++] = child_list[k];
        }
        if (reached != n) { rc = -1; goto done; }
    }

    for (uint32_t root = 0; root < n; ++root) {
        if (parent[root] != ORC_NO_PARENT) continue;
        int has_children = child_start[root + 1] > child_start[root];
        if (!has_children) {
            /* sync_simple_transforms, systems.rs:42-63 */
            if (transform_changed && !transform_changed[root]) continue;
            aff a = aff_from_srt(s + 3 * root, r + 4 * root, t + 3 * root);
            aff_store(&a, global + 12 * root);
            g_changed[root] = 1;
            continue;
        }
        /* propagate_parent_transforms roots, systems.rs:522-555 */
        if (static_opt && tree_changed && !tree_changed[root]) continue;
        {
            aff a = aff_from_srt(s + 3 * root, r + 4 * root, t + 3 * root);
            aff_store(&a, global + 12 * root); /* plain assignment: always marks changed */
            g_changed[root] = 1;
        }
        uint32_t sp = 0;
        stack[sp++] = root;
        while (sp) {
            uint32_t p = stack[--sp];
            aff gp = aff_load(global + 12 * p);
            for (uint32_t k = child_start[p]; k < child_start[p + 1]; ++k) {
                uint32_t c = child_list[k];
                /* static scene optimisation, systems.rs:708-714 */
                if (static_opt && !(tree_changed == NULL || tree_changed[c]) && !g_changed[p]) continue;
                if (visited[c]) { rc = -1; goto done; }
                visited[c] = 1;
                aff local = aff_from_srt(s + 3 * c, r + 4 * c, t + 3 * c);
                aff gc = aff_mul(&gp, &local);
                float tmp[12];
                aff_store(&gc, tmp);
                /* set_if_neq, systems.rs:719 */
                if (!aff_eq(tmp, global + 12 * c)) {
                    memcpy(global + 12 * c, tmp, sizeof tmp);
                    g_changed[c] = 1;
                }
                if (child_start[c + 1] > child_start[c]) stack[sp++] = c;
            }
        }
    }
    if (changed_out) memcpy(changed_out, g_changed, n);
done:
    free(child_count); free(child_start); free(child_list); free(stack); free(g_changed); free(visited);
    return rc;
}

int orc_compute_global_transform(uint32_t n, const uint32_t* parent, const float* t, const float* r,
                                 const float* s, uint32_t row, float out[12]) {
    if (row >= n) return -1;
    aff g = aff_from_srt(s + 3 * row, r + 4 * row, t + 3 * row);
    uint32_t cur = parent[row], steps = 0;
    while (cur != ORC_NO_PARENT) {
        if (cur >= n || ++steps > n) return -1;
        /* Transform * GlobalTransform = GlobalTransform::from(T) * G, transform.rs:673-680 */
        aff a = aff_from_srt(s + 3 * cur, r + 4 * cur, t + 3 * cur);
        g = aff_mul(&a, &g);
        cur = parent[cur];
    }
    aff_store(&g, out);
    return 0;
}

/* ======================================================================================= */
/* visibility                                                                                */
/* ======================================================================================= */

void orc_reset_view_visibility(uint32_t n, const uint8_t* flags, uint8_t* vv) {
    for (uint32_t i = 0; i < n; ++i) {
        if (flags[i] & ORC_FLAG_NO_CPU_CULLING) continue; /* Without<NoCpuCulling> */
        vv[i] = (uint8_t)((vv[i] & 1u) << 1);
    }
}

/* the per-entity closure body of check_visibility_cpu_culling, visibility/mod.rs:788-858 */
static inline int entity_visible_in_view(const float* g, const float* c, const float* h, uint8_t fl,
                                         uint32_t entity_mask, int in_range, const float* frustum,
                                         uint32_t view_mask, int no_cpu_culling_camera) {
    if (!(fl & ORC_FLAG_INHERITED_VISIBLE)) return 0;
    if (!(view_mask & entity_mask)) return 0;
    if ((fl & ORC_FLAG_HAS_VISIBILITY_RANGE) && !in_range) return 0;
    if (!(fl & ORC_FLAG_NO_FRUSTUM_CULLING) && !no_cpu_culling_camera) {
        if (fl & ORC_FLAG_HAS_AABB) {
            aff wfl = aff_load(g);
            v3 center = V3(c[0], c[1], c[2]), half = V3(h[0], h[1], h[2]);
            v3 sc = aff_point(&wfl, center);
            float sr = v3_length(m3_mul_v3(&wfl.m, half));
            if (!frustum_intersects_sphere(frustum, sc, sr, 0)) return 0;
            if (!frustum_intersects_obb(frustum, center, half, &wfl, 1, 0)) return 0;
        } else if (fl & ORC_FLAG_HAS_SPHERE) {
            if (!frustum_intersects_sphere(frustum, V3(c[0], c[1], c[2]), h[0], 0)) return 0;
        }
    }
    return 1;
}

/* SetViewVisibility::set_visible, visibility/mod.rs:290-306 */
static inline void set_visible(uint8_t* vv, uint8_t* changed) {
    if ((*vv & 1u) == 0) {
        if (*vv & 2u) {
            *vv |= 1u; /* bypass_change_detection */
        } else {
            *vv |= 1u;
            if (changed) *changed = 1;
        }
    }
}

void orc_check_visibility(uint32_t n, const float* global, const float* aabb_center, const float* aabb_half,
                          const uint8_t* flags, const uint32_t* layer_mask, const uint8_t* in_range,
                          uint8_t* vv, const float* frusta, const uint32_t* view_layer_masks,
                          const uint8_t* view_flags, uint32_t n_views, uint8_t* visible_out,
                          uint8_t* vv_changed_out) {
    for (uint32_t v = 0; v < n_views; ++v) {
        const float* frustum = frusta + 24 * v;
        uint32_t view_mask = view_layer_masks ? view_layer_masks[v] : 1u;
        int ncc = view_flags ? (view_flags[v] & ORC_VIEW_FLAG_NO_CPU_CULLING) != 0 : 0;
        for (uint32_t i = 0; i < n; ++i) {
            uint8_t fl = flags[i];
            uint8_t vis = 0;
            if (!(fl & ORC_FLAG_NO_CPU_CULLING)) {
                uint32_t em = layer_mask ? layer_mask[i] : 1u;
                int ir = in_range ? in_range[(size_t)v * n + i] : 1;
                vis = (uint8_t)entity_visible_in_view(global + 12 * (size_t)i, aabb_center + 3 * (size_t)i,
                                                      aabb_half + 3 * (size_t)i, fl, em, ir, frustum, view_mask, ncc);
                if (vis) set_visible(&vv[i], vv_changed_out ? &vv_changed_out[i] : NULL);
            }
            if (visible_out) visible_out[(size_t)v * n + i] = vis;
        }
    }
}

/* PARITY-MARGIN CENSUS (tests/test_parity_margin.py): how close every DECIDING value of check_visibility's plane tests sits to
 * its `<= 0.0` (primitives.rs:263, 289), in ulps of the largest term of the sum that forms it -- the only rows a last-ulp
 * difference in glam's lane order could move.  Camera views; for every row that reaches the frustum tests (InheritedVisibility,
 * RenderLayers, not NoFrustumCulling) all five planes of intersects_sphere and, for rows with an Aabb, of intersects_obb are
 * binned (no early exit: a later plane's value decides under another order).  hist[0..4] = values within 1 / 4 / 16 / 64 / 1024
 * ulps (cumulative), hist[5] = values binned. */
static inline void margin_bin(float val, float big, uint64_t* hist) {
    float ulp = nextafterf(big, INFINITY) - big;
    float m = fabsf(val) / ulp;
    static const float lim[5] = {1.0f, 4.0f, 16.0f, 64.0f, 1024.0f};
    for (int k = 0; k < 5; ++k) if (m < lim[k]) hist[k]++;
    hist[5]++;
}
void orc_visibility_margin_census(uint32_t n, const float* global, const float* aabb_center, const float* aabb_half,
                                  const uint8_t* flags, const uint32_t* layer_mask, const float* frusta,
                                  const uint32_t* view_layer_masks, uint32_t n_views, uint64_t hist[6]) {
    for (uint32_t v = 0; v < n_views; ++v) {
        const float* frustum = frusta + 24 * v;
        uint32_t view_mask = view_layer_masks ? view_layer_masks[v] : 1u;
        for (uint32_t i = 0; i < n; ++i) {
            uint8_t fl = flags[i];
            if ((fl & ORC_FLAG_NO_CPU_CULLING) || !(fl & ORC_FLAG_INHERITED_VISIBLE) || (fl & ORC_FLAG_NO_FRUSTUM_CULLING)) continue;
            if (!(view_mask & (layer_mask ? layer_mask[i] : 1u))) continue;
            if (!(fl & (ORC_FLAG_HAS_AABB | ORC_FLAG_HAS_SPHERE))) continue;
            const float* c = aabb_center + 3 * (size_t)i;
            const float* h = aabb_half + 3 * (size_t)i;
            aff wfl = aff_load(global + 12 * (size_t)i);
            v3 center = V3(c[0], c[1], c[2]), half = V3(h[0], h[1], h[2]);
            int has_aabb = (fl & ORC_FLAG_HAS_AABB) != 0;
            v3 sc = has_aabb ? aff_point(&wfl, center) : center;
            float sr = has_aabb ? v3_length(m3_mul_v3(&wfl.m, half)) : h[0];
            v4 c4 = v3_extend(sc, 1.0f);
            for (int p = 0; p < 5; ++p) {
                v4 hs = plane_at(frustum, p);
                float d = v4_dot(hs, c4);
                float big = fmaxf(fmaxf(fabsf(hs.x * sc.x), fabsf(hs.y * sc.y)), fmaxf(fabsf(hs.z * sc.z), fabsf(hs.w)));
                margin_bin(d + sr, fmaxf(big, fabsf(sr)), hist);
                if (has_aabb) {
                    float rr = aabb_relative_radius(half, v4_xyz(hs), &wfl.m);
                    margin_bin(d + rr, fmaxf(big, fabsf(rr)), hist);
                }
            }
        }
    }
}

/* The same with RenderLayers of up to 64 layers -- the first u64 word of the reference's bitset; RenderLayers::intersects
 * (render_layers.rs:121-135) is "some word of the two masks has a common bit".  layer_mask_hi / view_layer_masks_hi: layers 32..63
 * (NULL = none). */
void orc_check_visibility_layers64(uint32_t n, const float* global, const float* aabb_center, const float* aabb_half,
                                   const uint8_t* flags, const uint32_t* layer_mask, const uint32_t* layer_mask_hi, const uint8_t* in_range,
                                   uint8_t* vv, const float* frusta, const uint32_t* view_layer_masks, const uint32_t* view_layer_masks_hi,
                                   const uint8_t* view_flags, uint32_t n_views, uint8_t* visible_out, uint8_t* vv_changed_out) {
    for (uint32_t v = 0; v < n_views; ++v) {
        const float* frustum = frusta + 24 * v;
        uint64_t view_mask = (view_layer_masks ? view_layer_masks[v] : 1u) | ((uint64_t)(view_layer_masks_hi ? view_layer_masks_hi[v] : 0u) << 32);
        int ncc = view_flags ? (view_flags[v] & ORC_VIEW_FLAG_NO_CPU_CULLING) != 0 : 0;
        for (uint32_t i = 0; i < n; ++i) {
            uint8_t fl = flags[i];
            uint8_t vis = 0;
            if (!(fl & ORC_FLAG_NO_CPU_CULLING)) {
                uint64_t em = (layer_mask ? layer_mask[i] : 1u) | ((uint64_t)(layer_mask_hi ? layer_mask_hi[i] : 0u) << 32);
                int ir = in_range ? in_range[(size_t)v * n + i] : 1;
                vis = (uint8_t)entity_visible_in_view(global + 12 * (size_t)i, aabb_center + 3 * (size_t)i, aabb_half + 3 * (size_t)i, fl,
                                                      (view_mask & em) ? 1u : 0u, ir, frustum, 1u, ncc);
                if (vis) set_visible(&vv[i], vv_changed_out ? &vv_changed_out[i] : NULL);
            }
            if (visible_out) visible_out[(size_t)v * n + i] = vis;
        }
    }
}

/* VisibilityRange::is_visible_at_all(distance), range.rs:159-161, with the model position rule of :255-263 */
static inline int entity_in_range_of(const float* g, const float* c, uint8_t fl, const float* start_end,
                                     const float* view_pos) {
    aff wfl = aff_load(g);
    v3 model = ((fl & ORC_FLAG_RANGE_USE_AABB) && (fl & ORC_FLAG_HAS_AABB)) ? aff_point(&wfl, V3(c[0], c[1], c[2]))
                                                                           : wfl.t;
    float d = v3_length(v3_sub(V3(view_pos[0], view_pos[1], view_pos[2]), model));
    return d >= start_end[0] && d < start_end[1];
}

void orc_check_visibility_ranges(uint32_t n, const float* global, const float* aabb_center,
                                 const uint8_t* flags, const float* range_start_end,
                                 const float* view_positions, uint32_t n_views, uint8_t* in_range_out) {
    if (n_views > 32) n_views = 32; /* .take(32), range.rs:240 */
    for (uint32_t v = 0; v < n_views; ++v)
        for (uint32_t i = 0; i < n; ++i) {
            uint8_t fl = flags[i];
            int in = 0;
            if ((fl & ORC_FLAG_HAS_VISIBILITY_RANGE) && !(fl & ORC_FLAG_NO_CPU_CULLING))
                in = entity_in_range_of(global + 12 * (size_t)i, aabb_center + 3 * (size_t)i, fl,
                                        range_start_end + 2 * (size_t)i, view_positions + 3 * (size_t)v);
            in_range_out[(size_t)v * n + i] = (uint8_t)in;
        }
}

/* shadow-view closures: bevy_light/src/lib.rs:425-475 (cascades), :592-650 (cube faces), :694-738 (spot) */
static inline int entity_visible_in_shadow_view(const float* g, const float* c, const float* h, uint8_t fl,
                                                uint32_t entity_mask, int in_range, const orc_view* view) {
    if (!(fl & ORC_FLAG_SHADOW_CASTER)) return 0; /* not matched by visible_entity_query */
    if (!(fl & ORC_FLAG_INHERITED_VISIBLE)) return 0;
    if (!(view->layer_mask & entity_mask)) return 0;
    if ((fl & ORC_FLAG_HAS_VISIBILITY_RANGE) && !in_range) return 0;
    if (fl & ORC_FLAG_HAS_AABB) { /* (Some(aabb), Some(transform)) */
        aff wfl = aff_load(g);
        v3 center = V3(c[0], c[1], c[2]), half = V3(h[0], h[1], h[2]);
        if (!(fl & ORC_FLAG_NO_FRUSTUM_CULLING)) {
            if ((view->flags & ORC_VIEW_FLAG_LIGHT_SPHERE) &&
                !orc_sphere_intersects_obb(view->light_sphere, view->light_sphere[3], c, h, g))
                return 0;
            if (!frustum_intersects_obb(view->frustum, center, half, &wfl, !(view->flags & ORC_VIEW_FLAG_SKIP_NEAR),
                                        (view->flags & ORC_VIEW_FLAG_TEST_FAR) != 0))
                return 0;
        }
    }
    return 1;
}

void orc_check_visibility_views(uint32_t n, const float* global, const float* aabb_center,
                                const float* aabb_half, const uint8_t* flags, const uint32_t* layer_mask,
                                const float* range_start_end, uint8_t* vv, const orc_view* views,
                                uint32_t n_views, uint8_t* visible_out, uint8_t* vv_changed_out) {
    for (uint32_t v = 0; v < n_views; ++v) {
        const orc_view* view = &views[v];
        for (uint32_t i = 0; i < n; ++i) {
            uint8_t fl = flags[i];
            uint8_t vis = 0;
            if (!(fl & ORC_FLAG_NO_CPU_CULLING)) {
                uint32_t em = layer_mask ? layer_mask[i] : 1u;
                const float* g = global + 12 * (size_t)i;
                const float* c = aabb_center + 3 * (size_t)i;
                /* has_visibility_range && visible_entity_ranges.is_some_and(|r| !r.entity_is_in_range_of_view(..)) */
                int in_range = 1;
                if ((fl & ORC_FLAG_HAS_VISIBILITY_RANGE) && range_start_end) {
                    if (view->flags & ORC_VIEW_FLAG_RANGES_NO_ORIGIN) in_range = 0;
                    else if (view->flags & ORC_VIEW_FLAG_RANGES)
                        in_range = entity_in_range_of(g, c, fl, range_start_end + 2 * (size_t)i, view->position);
                    else in_range = 0; /* the view has no index in VisibleEntityRanges::views */
                }
                if (view->flags & ORC_VIEW_FLAG_SHADOW)
                    vis = (uint8_t)entity_visible_in_shadow_view(g, c, aabb_half + 3 * (size_t)i, fl, em, in_range, view);
                else
                    vis = (uint8_t)entity_visible_in_view(g, c, aabb_half + 3 * (size_t)i, fl, em, in_range, view->frustum,
                                                          view->layer_mask,
                                                          (view->flags & ORC_VIEW_FLAG_NO_CPU_CULLING) != 0);
                if (vis) set_visible(&vv[i], vv_changed_out ? &vv_changed_out[i] : NULL);
            }
            if (visible_out) visible_out[(size_t)v * n + i] = vis;
        }
    }
}

int orc_visibility_propagate(uint32_t n, const uint32_t* parent, const uint8_t* visibility,
                             uint8_t* inherited, uint8_t* changed_out) {
    /* resolve in root-to-leaf order: depth of every row first (also detects cycles) */
    uint32_t* depth = (uint32_t*)malloc((size_t)(n ? n : 1) * sizeof(uint32_t));
    uint32_t* order = (uint32_t*)malloc((size_t)(n ? n : 1) * sizeof(uint32_t));
    uint32_t maxd = 0;
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t d = 0, cur = i;
        while (parent && parent[cur] != ORC_NO_PARENT) {
            cur = parent[cur];
            if (cur >= n || ++d > n) { free(depth); free(order); return -1; }
        }
        depth[i] = d;
        if (d > maxd) maxd = d;
    }
    /* counting sort by depth */
    uint32_t* start = (uint32_t*)calloc((size_t)maxd + 2, sizeof(uint32_t));
    for (uint32_t i = 0; i < n; ++i) start[depth[i] + 1]++;
    for (uint32_t d = 0; d <= maxd; ++d) start[d + 1] += start[d];
    for (uint32_t i = 0; i < n; ++i) order[start[depth[i]]++] = i;
    for (uint32_t k = 0; k < n; ++k) {
        uint32_t i = order[k];
        if (changed_out) changed_out[i] = 0;
        uint8_t vis = visibility[i];
        if (vis & 0x80u) continue; /* no Visibility / InheritedVisibility: never touched */
        uint8_t is_visible;
        if (vis == 2u) is_visible = 1;
        else if (vis == 1u) is_visible = 0;
        else {
            uint32_t p = parent ? parent[i] : ORC_NO_PARENT;
            /* "fall back to true if no parent is found or parent lacks components" (mod.rs:656-659) */
            is_visible = (p == ORC_NO_PARENT || (visibility[p] & 0x80u)) ? 1 : (inherited[p] & 1u);
        }
        if ((inherited[i] & 1u) != is_visible) {
            inherited[i] = is_visible;
            if (changed_out) changed_out[i] = 1;
        }
    }
    free(depth); free(order); free(start);
    return 0;
}

void orc_check_visibility_gpu_culling(uint32_t n, const uint8_t* flags, uint8_t* vv, uint8_t* vv_changed_out) {
    for (uint32_t i = 0; i < n; ++i) {
        if (!(flags[i] & ORC_FLAG_NO_CPU_CULLING)) continue;
        uint8_t nv = (flags[i] & ORC_FLAG_INHERITED_VISIBLE) ? 3u : 0u; /* VISIBLE / HIDDEN */
        if (vv[i] != nv) { /* set_if_neq */
            vv[i] = nv;
            if (vv_changed_out) vv_changed_out[i] = 1;
        }
    }
}

void orc_mark_newly_hidden(uint32_t n, const uint8_t* flags, uint8_t* vv, uint8_t* vv_changed_out) {
    for (uint32_t i = 0; i < n; ++i) {
        if (flags[i] & ORC_FLAG_NO_CPU_CULLING) continue;
        if ((vv[i] & 3u) == 2u) {
            vv[i] = 0;
            if (vv_changed_out) vv_changed_out[i] = 1;
        }
    }
}

typedef struct { uint64_t key; uint32_t row; } key_row;
static int key_row_cmp(const void* a, const void* b) {
    const key_row* x = (const key_row*)a; const key_row* y = (const key_row*)b;
    if (x->key < y->key) return -1;
    if (x->key > y->key) return 1;
    return (x->row > y->row) - (x->row < y->row);
}
uint32_t orc_visible_entities_sorted(uint32_t n, const uint8_t* visible, const uint32_t* class_mask,
                                     uint32_t class_bit, const uint64_t* entity_keys, uint64_t* out_keys,
                                     uint32_t* out_rows) {
    key_row* tmp = (key_row*)malloc((size_t)(n ? n : 1) * sizeof(key_row));
    uint32_t m = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (!visible[i]) continue;
        if (!((class_mask ? class_mask[i] : 1u) & (1u << class_bit))) continue;
        tmp[m].key = entity_keys ? entity_keys[i] : (uint64_t)i;
        tmp[m].row = i;
        ++m;
    }
    qsort(tmp, m, sizeof(key_row), key_row_cmp); /* sort_unstable on Entity = to_bits order */
    for (uint32_t i = 0; i < m; ++i) {
        if (out_keys) out_keys[i] = tmp[i].key;
        if (out_rows) out_rows[i] = tmp[i].row;
    }
    free(tmp);
    return m;
}

/* ======================================================================================= */
/* light clustering (crates/bevy_light/src/cluster/{assign,mod}.rs)                          */
/* ======================================================================================= */

/* Rust `f32 as u32`: saturating, NaN -> 0 */
static inline uint32_t f32_as_u32(float f) {
    if (!(f > 0.0f)) return 0u; /* negative, zero, NaN */
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}

void orc_cluster_dimensions_fixed_z(uint32_t total, uint32_t z_slices, uint32_t sw, uint32_t sh, uint32_t out[3]) {
    float aspect_ratio = (float)sw / (float)sh; /* AspectRatio::try_from_pixels(..).ratio() */
    if (total < z_slices) z_slices = total;
    float per_layer = (float)total / (float)z_slices;
    float y = sqrtf(per_layer / aspect_ratio);
    uint32_t x = f32_as_u32(y * aspect_ratio);
    uint32_t yi = f32_as_u32(y);
    if (x == 0) { x = 1; yi = f32_as_u32(per_layer); }
    if (yi == 0) { x = f32_as_u32(per_layer); yi = 1; }
    out[0] = x; out[1] = yi; out[2] = z_slices;
}

void orc_clusters_update(uint32_t sw, uint32_t sh, const uint32_t req[3], uint32_t tile[2], uint32_t dims[3]) {
    uint32_t tx = f32_as_u32(ceilf((float)sw / (float)req[0]));
    uint32_t ty = f32_as_u32(ceilf((float)sh / (float)req[1]));
    if (tx < 1) tx = 1;
    if (ty < 1) ty = 1;
    tile[0] = tx; tile[1] = ty;
    uint32_t dx = f32_as_u32(ceilf((float)sw / (float)tx));
    uint32_t dy = f32_as_u32(ceilf((float)sh / (float)ty));
    dims[0] = dx < 1 ? 1 : dx;
    dims[1] = dy < 1 ? 1 : dy;
    dims[2] = req[2] < 1 ? 1 : req[2];
}

/* ClusterConfig::default(), cluster/mod.rs:297-308 (ClusterZConfig::default :288-295) */
void orc_cluster_config_default(orc_cluster_config* out) {
    memset(out, 0, sizeof *out);
    out->kind = 3; /* FixedZ */
    out->total = 4096;
    out->z_slices = 24;
    out->first_slice_depth = 5.0f;
    out->far_z_mode = 0; /* MaxClusterableObjectRange */
    out->dynamic_resizing = 1;
}

/* assign.rs:324-404 up to (not including) clusters.update() */
int orc_cluster_config_resolve(const orc_cluster_config* config, const float* last_farthest_z, const uint64_t* last_total,
                               uint32_t sw, uint32_t sh, uint64_t max_indices, uint32_t req[3],
                               float* out_first_slice_depth, float* out_far_z) {
    /* :328-331 */
    if (config->kind == 0) return 0;
    /* :333-339: physical_viewport_size() must be Some and non-zero */
    if (sw == 0 || sh == 0) return 0;
    /* :342 config.dimensions_for_screen_size(screen_size), mod.rs:311-347 */
    switch (config->kind) {
    case 1: req[0] = req[1] = req[2] = 1; break;                       /* Single => UVec3::ONE */
    case 2: memcpy(req, config->dimensions, 3 * sizeof(uint32_t)); break; /* XYZ => *dimensions */
    default: orc_cluster_dimensions_fixed_z(config->total, config->z_slices, sw, sh, req); break;
    }
    /* mod.rs:349-382 */
    float first_slice_depth = (config->kind == 2 || config->kind == 3) ? config->first_slice_depth : 0.0f;
    int far_constant = (config->kind == 2 || config->kind == 3) && config->far_z_mode == 1;
    int dynamic_resizing = (config->kind == 2 || config->kind == 3) && config->dynamic_resizing != 0;
    /* :350-355 */
    float far_z;
    if (far_constant) far_z = config->far_z_constant;
    else far_z = last_farthest_z ? *last_farthest_z : 1000.0f; /* unwrap_or(DEFAULT_FAR_DEPTH) */
    /* :384-404 */
    if (dynamic_resizing && last_total && *last_total > max_indices) {
        float index_ratio = (float)max_indices / (float)*last_total;
        float xy_ratio = sqrtf(index_ratio);
        uint32_t x = f32_as_u32(floorf((float)req[0] * xy_ratio));
        uint32_t y = f32_as_u32(floorf((float)req[1] * xy_ratio));
        req[0] = x > 1 ? x : 1;
        req[1] = y > 1 ? y : 1;
    }
    *out_first_slice_depth = first_slice_depth;
    *out_far_z = far_z;
    return 1;
}

/* ClusterableObjectType::ordering(), assign.rs:108-128, followed by the Entity: the sort key of :301-306 */
typedef struct { uint8_t type, not_shadow, not_volumetric; uint64_t entity; uint32_t index; } sort_key_t;
static int sort_key_cmp(const void* pa, const void* pb) {
    const sort_key_t* a = (const sort_key_t*)pa;
    const sort_key_t* b = (const sort_key_t*)pb;
    if (a->type != b->type) return a->type < b->type ? -1 : 1;
    if (a->not_shadow != b->not_shadow) return a->not_shadow < b->not_shadow ? -1 : 1;
    if (a->not_volumetric != b->not_volumetric) return a->not_volumetric < b->not_volumetric ? -1 : 1;
    if (a->entity != b->entity) return a->entity < b->entity ? -1 : 1;
    return a->index < b->index ? -1 : (a->index > b->index ? 1 : 0); /* stable (sort_by_cached_key is) */
}
uint32_t orc_cluster_sort_truncate(uint32_t n, const uint8_t* obj_type, const uint8_t* shadow_maps_enabled,
                                   const uint8_t* volumetric, const uint64_t* entity_bits, uint32_t max_objects,
                                   int supports_storage_buffers, uint32_t* order) {
    for (uint32_t i = 0; i < n; ++i) order[i] = i;
    if (!(n > max_objects && !supports_storage_buffers)) return n; /* :297-300 */
    sort_key_t* keys = (sort_key_t*)malloc((size_t)n * sizeof *keys);
    for (uint32_t i = 0; i < n; ++i) {
        uint8_t t = obj_type ? obj_type[i] : ORC_OBJ_POINT_LIGHT;
        int light = t == ORC_OBJ_POINT_LIGHT || t == ORC_OBJ_SPOT_LIGHT;
        keys[i].type = t;
        keys[i].not_shadow = light ? (uint8_t)!(shadow_maps_enabled && shadow_maps_enabled[i]) : 0;
        keys[i].not_volumetric = light ? (uint8_t)!(volumetric && volumetric[i]) : 0;
        keys[i].entity = entity_bits[i];
        keys[i].index = i;
    }
    qsort(keys, n, sizeof *keys, sort_key_cmp);
    for (uint32_t i = 0; i < n; ++i) order[i] = keys[i].index;
    free(keys);
    return max_objects; /* truncate, :319-320 */
}

/* calculate_cluster_factors, assign.rs:817-832 */
static void calculate_cluster_factors(float near, float far, float z_slices, int ortho, float out[2]) {
    if (ortho) {
        out[0] = -near;
        out[1] = z_slices / (-far - -near);
    } else {
        float k = (z_slices - 1.0f) / logf(far / near);
        out[0] = k;
        out[1] = logf(near) * k;
    }
}
/* z_slice_to_view_z, assign.rs:903-920 */
static float z_slice_to_view_z(float near, float far, uint32_t z_slices, uint32_t z_slice, int ortho) {
    if (ortho) return -near - (far - near) * (float)z_slice / (float)z_slices;
    if (z_slice == 0) return 0.0f;
    return -near * powf(far / near, (float)(z_slice - 1) / (float)(z_slices - 1));
}
/* clip_to_view, assign.rs:1064-1067 */
static v4 clip_to_view(const m4* view_from_clip, v4 clip) {
    v4 view = m4_mul_v4(view_from_clip, clip);
    return V4(view.x / view.w, view.y / view.w, view.z / view.w, view.w / view.w);
}
/* scalar Vec3::cross */
static inline v3 vec3_cross(v3 a, v3 b) {
    return V3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
static inline float signum(float x) { return isnan(x) ? x : (signbit(x) ? -1.0f : 1.0f); }

void orc_cluster_view_setup(const float camera_affine[12], const float clip_from_view[16], const float frustum[24],
                            uint32_t sw, uint32_t sh, const uint32_t requested_dims[3], float first_slice_depth_cfg,
                            float far_z, uint32_t view_layer_mask, orc_cluster_view* out) {
    memset(out, 0, sizeof *out);
    aff world_from_view = aff_load(camera_affine);
    /* camera_transform.compute_transform().scale.recip(), assign.rs:345 (to_scale_rotation_translation) */
    float det = m3_determinant(&world_from_view.m);
    v3 scale = V3(v3_length(world_from_view.m.x_axis) * signum(det), v3_length(world_from_view.m.y_axis),
                  v3_length(world_from_view.m.z_axis));
    v3 vfw_scale = V3(1.0f / scale.x, 1.0f / scale.y, 1.0f / scale.z);
    float scale_max = rust_max(rust_max(fabsf(vfw_scale.x), fabsf(vfw_scale.y)), fabsf(vfw_scale.z));
    aff inv = aff_inverse(&world_from_view);
    m4 view_from_world = m4_from_affine(&inv);
    m4 cfv = m4_load(clip_from_view);
    int ortho = cfv.c[3].w == 1.0f;

    float first_slice_depth;
    if (ortho) first_slice_depth = (cfv.c[3].z - 1.0f) / cfv.c[2].z;
    else if (requested_dims[2] == 1) first_slice_depth = rust_max(first_slice_depth_cfg, far_z);
    else first_slice_depth = first_slice_depth_cfg;
    first_slice_depth = first_slice_depth * vfw_scale.z;
    far_z = rust_max(far_z, first_slice_depth);
    calculate_cluster_factors(first_slice_depth, far_z, (float)requested_dims[2], ortho, out->cluster_factors);

    orc_clusters_update(sw, sh, requested_dims, out->tile_size, out->dims);
    out->screen_size[0] = sw; out->screen_size[1] = sh;
    out->is_orthographic = (uint32_t)ortho;
    out->view_layer_mask = view_layer_mask;
    out->view_layer_mask_hi = 0u; /* (layers 32..63: the caller sets it) */
    out->near_ = first_slice_depth;
    out->far_ = far_z;
    m4_store(&view_from_world, out->view_from_world);
    m4_store(&cfv, out->clip_from_view);
    m4 view_from_clip = m4_inverse(&cfv);
    m4_store(&view_from_clip, out->view_from_clip);
    out->view_from_world_scale[0] = vfw_scale.x; out->view_from_world_scale[1] = vfw_scale.y;
    out->view_from_world_scale[2] = vfw_scale.z;
    out->view_from_world_scale_max = scale_max;
    memcpy(out->frustum, frustum, 24 * sizeof(float));

    uint32_t dx = out->dims[0], dy = out->dims[1], dz = out->dims[2];
    out->n_x_planes = dx + 1; out->n_y_planes = dy + 1; out->n_z_planes = dz + 1;
    /* assign.rs:434-476 */
    if (ortho) {
        float x_slices = (float)dx;
        for (uint32_t x = 0; x <= dx; ++x) {
            float x_proportion = (float)x / x_slices;
            float x_pos = x_proportion * 2.0f - 1.0f;
            float view_x = clip_to_view(&view_from_clip, V4(x_pos, 0.0f, 1.0f, 1.0f)).x;
            v3 normal = V3(1.0f, 0.0f, 0.0f);
            float d = view_x * normal.x;
            v4 hs = half_space_new(v3_extend(normal, d));
            memcpy(out->x_planes + 4 * x, &hs, sizeof hs);
        }
        float y_slices = (float)dy;
        for (uint32_t y = 0; y <= dy; ++y) {
            float y_proportion = 1.0f - (float)y / y_slices;
            float y_pos = y_proportion * 2.0f - 1.0f;
            float view_y = clip_to_view(&view_from_clip, V4(0.0f, y_pos, 1.0f, 1.0f)).y;
            v3 normal = V3(0.0f, 1.0f, 0.0f);
            float d = view_y * normal.y;
            v4 hs = half_space_new(v3_extend(normal, d));
            memcpy(out->y_planes + 4 * y, &hs, sizeof hs);
        }
    } else {
        float x_slices = (float)dx;
        for (uint32_t x = 0; x <= dx; ++x) {
            float x_proportion = (float)x / x_slices;
            float x_pos = x_proportion * 2.0f - 1.0f;
            v3 nb = v4_xyz(clip_to_view(&view_from_clip, V4(x_pos, -1.0f, 1.0f, 1.0f)));
            v3 nt = v4_xyz(clip_to_view(&view_from_clip, V4(x_pos, 1.0f, 1.0f, 1.0f)));
            v3 normal = vec3_cross(nb, nt);
            float d = v3_dot(nb, normal);
            v4 hs = half_space_new(v3_extend(normal, d));
            memcpy(out->x_planes + 4 * x, &hs, sizeof hs);
        }
        float y_slices = (float)dy;
        for (uint32_t y = 0; y <= dy; ++y) {
            float y_proportion = 1.0f - (float)y / y_slices;
            float y_pos = y_proportion * 2.0f - 1.0f;
            v3 nl = v4_xyz(clip_to_view(&view_from_clip, V4(-1.0f, y_pos, 1.0f, 1.0f)));
            v3 nr = v4_xyz(clip_to_view(&view_from_clip, V4(1.0f, y_pos, 1.0f, 1.0f)));
            v3 normal = vec3_cross(nr, nl);
            float d = v3_dot(nr, normal);
            v4 hs = half_space_new(v3_extend(normal, d));
            memcpy(out->y_planes + 4 * y, &hs, sizeof hs);
        }
    }
    /* assign.rs:478-485 */
    for (uint32_t z = 0; z <= dz; ++z) {
        float view_z = z_slice_to_view_z(first_slice_depth, far_z, dz, z, ortho);
        v3 normal = V3(-0.0f, -0.0f, -1.0f); /* -Vec3::Z */
        float d = view_z * normal.z;
        v4 hs = half_space_new(v3_extend(normal, d));
        memcpy(out->z_planes + 4 * z, &hs, sizeof hs);
    }
}

/* screen_to_view, assign.rs:1069-1078 */
static v4 screen_to_view(float sw, float sh, const m4* view_from_clip, float sx, float sy, float ndc_z) {
    float tx = sx / sw, ty = sy / sh;
    v4 clip = V4(tx * 2.0f - 1.0f, (1.0f - ty) * 2.0f - 1.0f, ndc_z, 1.0f);
    return clip_to_view(view_from_clip, clip);
}
/* line_intersection_to_z_plane with origin = Vec3::ZERO, assign.rs:1039-1043 */
static v3 line_intersection_to_z_plane(v3 p, float z) {
    v3 origin = V3(0.0f, 0.0f, 0.0f);
    v3 v = v3_sub(p, origin);
    float zo = (0.0f * origin.x + 0.0f * origin.y) + 1.0f * origin.z;
    float zv = (0.0f * v.x + 0.0f * v.y) + 1.0f * v.z;
    float t = (z - zo) / zv;
    return v3_add(origin, v3_scale(v, t));
}

void orc_cluster_aabb_sphere(const orc_cluster_view* view, uint32_t x, uint32_t y, uint32_t z, float out[4]) {
    /* compute_aabb_for_cluster, assign.rs:834-900 */
    m4 view_from_clip = m4_load(view->view_from_clip);
    float z_near = view->near_, z_far = view->far_;
    float tsx = (float)view->tile_size[0], tsy = (float)view->tile_size[1];
    float sw = (float)view->screen_size[0], sh = (float)view->screen_size[1];
    float ix = (float)x, iy = (float)y, iz = (float)z;
    float pminx = ix * tsx, pminy = iy * tsy;
    float pmaxx = pminx + tsx, pmaxy = pminy + tsy;
    v3 cmin, cmax;
    if (view->is_orthographic) {
        v3 p_min = v4_xyz(screen_to_view(sw, sh, &view_from_clip, pminx, pminy, 0.0f));
        v3 p_max = v4_xyz(screen_to_view(sw, sh, &view_from_clip, pmaxx, pmaxy, 0.0f));
        p_min.z = -z_near + (z_near - z_far) * iz / (float)view->dims[2];
        p_max.z = -z_near + (z_near - z_far) * (iz + 1.0f) / (float)view->dims[2];
        cmin = v3_min(p_min, p_max);
        cmax = v3_max(p_min, p_max);
    } else {
        v3 p_min = v4_xyz(screen_to_view(sw, sh, &view_from_clip, pminx, pminy, 1.0f));
        v3 p_max = v4_xyz(screen_to_view(sw, sh, &view_from_clip, pmaxx, pmaxy, 1.0f));
        float ratio = -z_far / -z_near;
        float cluster_near = (iz == 0.0f) ? 0.0f : -z_near * powf(ratio, (iz - 1.0f) / (float)(view->dims[2] - 1));
        float cluster_far = (view->dims[2] == 1) ? -z_far : -z_near * powf(ratio, iz / (float)(view->dims[2] - 1));
        v3 a = line_intersection_to_z_plane(p_min, cluster_near);
        v3 b = line_intersection_to_z_plane(p_min, cluster_far);
        v3 c = line_intersection_to_z_plane(p_max, cluster_near);
        v3 d = line_intersection_to_z_plane(p_max, cluster_far);
        cmin = v3_min(v3_min(a, b), v3_min(c, d));
        cmax = v3_max(v3_max(a, b), v3_max(c, d));
    }
    /* Aabb::from_min_max, primitives.rs:72-81 */
    v3 center = v3_scale(v3_add(cmax, cmin), 0.5f);
    v3 half = v3_scale(v3_sub(cmax, cmin), 0.5f);
    out[0] = center.x; out[1] = center.y; out[2] = center.z;
    out[3] = v3_length(half);
}

/* view_z_to_z_slice, assign.rs:1046-1062 */
static uint32_t view_z_to_z_slice(const float f[2], uint32_t z_slices, float view_z, int ortho) {
    uint32_t z_slice;
    if (ortho) z_slice = f32_as_u32(floorf((view_z - f[0]) * f[1]));
    else z_slice = f32_as_u32(logf(-view_z) * f[0] - f[1] + 1.0f);
    uint32_t lim = z_slices - 1;
    return z_slice < lim ? z_slice : lim;
}
static inline float clampf(float v, float lo, float hi) { return lane_min(lane_max(v, lo), hi); }
/* ndc_position_to_cluster, assign.rs:922-941 */
static void ndc_position_to_cluster(const uint32_t dims[3], const float factors[2], int ortho, float ndc_x,
                                    float ndc_y, float view_z, uint32_t out[3]) {
    float fx = clampf(ndc_x * 0.5f + 0.5f, 0.0f, 1.0f);
    float fy = clampf(ndc_y * -0.5f + 0.5f, 0.0f, 1.0f);
    float xf = floorf(fx * (float)dims[0]);
    float yf = floorf(fy * (float)dims[1]);
    uint32_t zs = view_z_to_z_slice(factors, dims[2], view_z, ortho);
    uint32_t xi = f32_as_u32(xf), yi = f32_as_u32(yf);
    out[0] = xi > dims[0] - 1 ? dims[0] - 1 : xi;
    out[1] = yi > dims[1] - 1 ? dims[1] - 1 : yi;
    out[2] = zs > dims[2] - 1 ? dims[2] - 1 : zs;
}

/* cluster_space_clusterable_object_aabb, assign.rs:948-1036 */
static void cluster_space_object_aabb(const m4* view_from_world, v3 vfw_scale, const m4* clip_from_view,
                                      v3 center, float radius, v3* out_min, v3* out_max) {
    v3 c = v4_xyz(m4_mul_v4(view_from_world, v3_extend(center, 1.0f)));
    v3 he = v3_scale(v3_abs(vfw_scale), radius); /* radius * scale.abs() */
    v3 vmin = v3_sub(c, he), vmax = v3_add(c, he);
    const float NEG_MIN_POS = -1.17549435e-38f; /* -f32::MIN_POSITIVE */
    vmin.z = rust_min(vmin.z, NEG_MIN_POS);
    vmax.z = rust_min(vmax.z, NEG_MIN_POS);
    v3 p[4] = {vmin, V3(vmin.x, vmin.y, vmax.z), V3(vmax.x, vmax.y, vmin.z), vmax};
    v3 ndc[4];
    for (int i = 0; i < 4; ++i) {
        v4 clip = m4_mul_v4(clip_from_view, v3_extend(p[i], 1.0f));
        ndc[i] = V3(clip.x / clip.w, clip.y / clip.w, clip.z / clip.w);
    }
    v3 nmin = v3_min(v3_min(v3_min(ndc[0], ndc[1]), ndc[2]), ndc[3]);
    v3 nmax = v3_max(v3_max(v3_max(ndc[0], ndc[1]), ndc[2]), ndc[3]);
    *out_min = V3(clampf(nmin.x, -1.0f, 1.0f), clampf(nmin.y, -1.0f, 1.0f), vmin.z);
    *out_max = V3(clampf(nmax.x, -1.0f, 1.0f), clampf(nmax.y, -1.0f, 1.0f), vmax.z);
}

typedef struct { v3 center; float radius; } sphere_t;

/* project_to_plane_z, assign.rs:1094-1113 */
static int project_to_plane_z(sphere_t* s, const float* plane) {
    float z = plane[3] / plane[2];
    float dist = z - s->center.z;
    if (fabsf(dist) > s->radius) return 0;
    s->center.z = z;
    s->radius = sqrtf(s->radius * s->radius - dist * dist);
    return 1;
}
/* project_to_plane_y, assign.rs:1116-1134 */
static int project_to_plane_y(sphere_t* s, const float* plane, int ortho) {
    float dist;
    if (ortho) dist = plane[3] - s->center.y;
    else dist = -(s->center.y * plane[1] + s->center.z * plane[2]);
    if (fabsf(dist) > s->radius) return 0;
    v3 n = V3(plane[0], plane[1], plane[2]);
    s->center = v3_add(s->center, v3_scale(n, dist));
    s->radius = sqrtf(s->radius * s->radius - dist * dist);
    return 1;
}
/* get_distance_x, assign.rs:1081-1091 */
static inline float get_distance_x(const float* plane, v3 point, int ortho) {
    if (ortho) return point.x - plane[3];
    return plane[0] * point.x + plane[2] * point.z;
}

typedef void (*emit_fn)(void* ctx, uint32_t cluster_index, uint32_t object, uint32_t type);

/* Per-object body of the loop at assign.rs:487-804; calls emit in exactly the push order. */
static void assign_one_object(const orc_cluster_view* view, const m4* view_from_world, const m4* clip_from_view,
                              v4 view_from_world_row_2, float* cluster_spheres, uint8_t* cluster_sphere_valid,
                              uint32_t obj, v3 center, float range, uint32_t type, uint64_t layer_mask,
                              const float* spot_dir, const float* spot_sin_cos, float* farthest_z,
                              emit_fn emit, void* ectx) {
    const uint32_t* dims = view->dims;
    int ortho = (int)view->is_orthographic;
    v3 vfw_scale = V3(view->view_from_world_scale[0], view->view_from_world_scale[1], view->view_from_world_scale[2]);
    float scale_max = view->view_from_world_scale_max;

    /* :489 RenderLayers::intersects over the first u64 word of the bitset (render_layers.rs:121-135) */
    if (!((((uint64_t)view->view_layer_mask_hi << 32) | view->view_layer_mask) & layer_mask)) return;
    if (!frustum_intersects_sphere(view->frustum, center, range, 1)) return;   /* :496 */

    v3 amin, amax;
    cluster_space_object_aabb(view_from_world, vfw_scale, clip_from_view, center, range, &amin, &amax);
    uint32_t c0[3], c1[3], minc[3], maxc[3];
    ndc_position_to_cluster(dims, view->cluster_factors, ortho, amin.x, amin.y, amin.z, c0);
    ndc_position_to_cluster(dims, view->cluster_factors, ortho, amax.x, amax.y, amax.z, c1);
    for (int k = 0; k < 3; ++k) { minc[k] = c0[k] < c1[k] ? c0[k] : c1[k]; maxc[k] = c0[k] > c1[k] ? c0[k] : c1[k]; }

    sphere_t vs;
    vs.center = v4_xyz(m4_mul_v4(view_from_world, v3_extend(center, 1.0f)));
    vs.radius = range * scale_max;

    float this_far_z = -v4_dot(view_from_world_row_2, v3_extend(center, 1.0f)) + range * vfw_scale.z; /* :558-560 */
    *farthest_z = rust_max(*farthest_z, this_far_z);

    v3 light_dir = V3(0, 0, 0);
    float angle_sin = 0.0f, angle_cos = 0.0f;
    if (type == ORC_OBJ_SPOT_LIGHT) {
        v4 d = m4_mul_v4(view_from_world, V4(spot_dir[0], spot_dir[1], spot_dir[2], 0.0f));
        v3 dv = v4_xyz(d);
        float recip = 1.0f / sqrtf(v3_dot(dv, dv));
        light_dir = v3_scale(dv, recip);
        angle_sin = spot_sin_cos[0]; angle_cos = spot_sin_cos[1];
    }
    v4 center_clip = m4_mul_v4(clip_from_view, v3_extend(vs.center, 1.0f));
    v3 ndc = V3(center_clip.x / center_clip.w, center_clip.y / center_clip.w, center_clip.z / center_clip.w);
    uint32_t cc[3];
    ndc_position_to_cluster(dims, view->cluster_factors, ortho, ndc.x, ndc.y, vs.center.z, cc);
    int z_center_some = ndc.z <= 1.0f; uint32_t z_center = cc[2];
    int y_center_some; uint32_t y_center = 0;
    if (ndc.y > 1.0f) y_center_some = 0;
    else if (ndc.y < -1.0f) { y_center_some = 1; y_center = dims[1] + 1; }
    else { y_center_some = 1; y_center = cc[1]; }

    for (uint32_t z = minc[2]; z <= maxc[2]; ++z) {
        sphere_t z_object = vs;
        if (!z_center_some || z != z_center) {
            const float* z_plane = (z_center_some && z < z_center) ? view->z_planes + 4 * (z + 1) : view->z_planes + 4 * z;
            if (!project_to_plane_z(&z_object, z_plane)) continue;
        }
        for (uint32_t y = minc[1]; y <= maxc[1]; ++y) {
            sphere_t y_object = z_object;
            if (!y_center_some || y != y_center) {
                const float* y_plane = (y_center_some && y < y_center) ? view->y_planes + 4 * (y + 1) : view->y_planes + 4 * y;
                if (!project_to_plane_y(&y_object, y_plane, ortho)) continue;
            }
            uint32_t min_x = minc[0];
            for (;;) {
                if (min_x >= maxc[0] ||
                    -get_distance_x(view->x_planes + 4 * (min_x + 1), y_object.center, ortho) + y_object.radius > 0.0f)
                    break;
                min_x += 1;
            }
            uint32_t max_x = maxc[0];
            for (;;) {
                if (max_x <= min_x ||
                    get_distance_x(view->x_planes + 4 * max_x, y_object.center, ortho) + y_object.radius > 0.0f)
                    break;
                max_x -= 1;
            }
            uint32_t cluster_index = (y * dims[0] + min_x) * dims[2] + z;
            if (type == ORC_OBJ_SPOT_LIGHT) {
                for (uint32_t x = min_x; x <= max_x; ++x) {
                    float* cs = cluster_spheres + 4 * (size_t)cluster_index;
                    if (!cluster_sphere_valid[cluster_index]) {
                        orc_cluster_aabb_sphere(view, x, y, z, cs);
                        cluster_sphere_valid[cluster_index] = 1;
                    }
                    v3 off = v3_sub(vs.center, V3(cs[0], cs[1], cs[2]));
                    float dist_sq = v3_dot(off, off);
                    float v1_len = v3_dot(off, light_dir);
                    float dcp = (angle_cos * sqrtf(dist_sq - v1_len * v1_len)) - v1_len * angle_sin;
                    int angle_cull = dcp > cs[3];
                    int front_cull = v1_len > cs[3] + range * scale_max;
                    int back_cull = v1_len < -cs[3];
                    if (!angle_cull && !front_cull && !back_cull) emit(ectx, cluster_index, obj, type);
                    cluster_index += dims[2];
                }
            } else {
                for (uint32_t x = min_x; x <= max_x; ++x) {
                    emit(ectx, cluster_index, obj, type);
                    cluster_index += dims[2];
                }
            }
        }
    }
}

typedef struct { uint32_t* per_cluster; uint32_t* counts; uint64_t total; } count_ctx;
static void emit_count(void* c, uint32_t cluster, uint32_t obj, uint32_t type) {
    (void)obj;
    count_ctx* k = (count_ctx*)c;
    k->per_cluster[cluster]++;
    k->counts[6 * (size_t)cluster + type]++;
    k->total++;
}
typedef struct { uint32_t* cursor; uint32_t* indices; } fill_ctx;
static void emit_fill(void* c, uint32_t cluster, uint32_t obj, uint32_t type) {
    (void)type;
    fill_ctx* k = (fill_ctx*)c;
    k->indices[k->cursor[cluster]++] = obj;
}

uint64_t orc_assign_objects_to_clusters(const orc_cluster_view* view, uint32_t n_objects, const float* pos_range,
                                        const uint8_t* obj_type, const uint32_t* obj_layer_mask,
                                        const float* spot_dir, const float* spot_sin_cos, uint32_t* offsets,
                                        uint32_t* indices, uint64_t capacity, uint32_t* counts,
                                        float* farthest_z_out) {
    return orc_assign_objects_to_clusters_layers64(view, n_objects, pos_range, obj_type, obj_layer_mask, NULL, spot_dir, spot_sin_cos, offsets,
                                                   indices, capacity, counts, farthest_z_out);
}

#define ORC_OBJ_LAYERS(i) ((uint64_t)(obj_layer_mask ? obj_layer_mask[i] : 1u) | ((uint64_t)(obj_layer_mask_hi ? obj_layer_mask_hi[i] : 0u) << 32))
uint64_t orc_assign_objects_to_clusters_layers64(const orc_cluster_view* view, uint32_t n_objects, const float* pos_range,
                                                 const uint8_t* obj_type, const uint32_t* obj_layer_mask, const uint32_t* obj_layer_mask_hi,
                                                 const float* spot_dir, const float* spot_sin_cos, uint32_t* offsets,
                                                 uint32_t* indices, uint64_t capacity, uint32_t* counts,
                                                 float* farthest_z_out) {
    uint32_t C = view->dims[0] * view->dims[1] * view->dims[2];
    m4 view_from_world = m4_load(view->view_from_world);
    m4 clip_from_view = m4_load(view->clip_from_view);
    v4 row2 = m4_row(&view_from_world, 2);
    float* spheres = (float*)calloc((size_t)C * 4 + 4, sizeof(float));
    uint8_t* valid = (uint8_t*)calloc((size_t)C + 1, 1);
    uint32_t* per_cluster = (uint32_t*)calloc((size_t)C + 1, sizeof(uint32_t));
    memset(counts, 0, (size_t)C * 6 * sizeof(uint32_t));
    float farthest = 0.0f;
    count_ctx cc = {per_cluster, counts, 0};
    for (uint32_t i = 0; i < n_objects; ++i) {
        v3 center = V3(pos_range[4 * (size_t)i], pos_range[4 * (size_t)i + 1], pos_range[4 * (size_t)i + 2]);
        assign_one_object(view, &view_from_world, &clip_from_view, row2, spheres, valid, i, center,
                          pos_range[4 * (size_t)i + 3], obj_type ? obj_type[i] : ORC_OBJ_POINT_LIGHT,
                          ORC_OBJ_LAYERS(i), spot_dir ? spot_dir + 3 * (size_t)i : NULL,
                          spot_sin_cos ? spot_sin_cos + 2 * (size_t)i : NULL, &farthest, emit_count, &cc);
    }
    offsets[0] = 0;
    for (uint32_t c = 0; c < C; ++c) offsets[c + 1] = offsets[c] + per_cluster[c];
    if (cc.total <= capacity && indices) {
        uint32_t* cursor = (uint32_t*)malloc(((size_t)C + 1) * sizeof(uint32_t));
        memcpy(cursor, offsets, ((size_t)C + 1) * sizeof(uint32_t));
        fill_ctx fc = {cursor, indices};
        float dummy = 0.0f;
        for (uint32_t i = 0; i < n_objects; ++i) {
            v3 center = V3(pos_range[4 * (size_t)i], pos_range[4 * (size_t)i + 1], pos_range[4 * (size_t)i + 2]);
            assign_one_object(view, &view_from_world, &clip_from_view, row2, spheres, valid, i, center,
                              pos_range[4 * (size_t)i + 3], obj_type ? obj_type[i] : ORC_OBJ_POINT_LIGHT,
                              ORC_OBJ_LAYERS(i), spot_dir ? spot_dir + 3 * (size_t)i : NULL,
                              spot_sin_cos ? spot_sin_cos + 2 * (size_t)i : NULL, &dummy, emit_fill, &fc);
        }
        free(cursor);
    }
    if (farthest_z_out) *farthest_z_out = farthest;
    free(spheres); free(valid); free(per_cluster);
    return cc.total;
}

/* ---- CPU baseline of the cluster stage as the reference runs it (bench.py only) ---------------------------------
 * assign_objects_to_clusters walks the objects ONCE and pushes each into the Vec<Entity> of every cluster it touches
 * (clusters.clusterable_objects[cluster_index].add_*(entity), assign.rs:740-800); the vectors are cleared at the start of
 * the frame and keep their capacity (Clusters::clear / the `clusterable_objects.clear()` + resize_with of assign.rs:412-420).
 * orc_assign_objects_to_clusters above walks twice (count, then fill a CSR) because the tests want the flat arrays; timing THAT
 * as the baseline flatters the device by the second walk.  This is the one-walk form, `iters` frames over the same objects. */
typedef struct { uint32_t* data; uint32_t len, cap; uint32_t counts[6]; } push_vec;
typedef struct { push_vec* clusters; uint64_t total; } push_ctx;
static void emit_push(void* c, uint32_t cluster, uint32_t obj, uint32_t type) {
    push_ctx* k = (push_ctx*)c;
    push_vec* v = &k->clusters[cluster];
    if (v->len == v->cap) {
        v->cap = v->cap ? 2 * v->cap : 4;
        v->data = (uint32_t*)realloc(v->data, (size_t)v->cap * sizeof(uint32_t));
    }
    v->data[v->len++] = obj;
    v->counts[type]++;
    k->total++;
}
double orc_bench_assign_objects_to_clusters(const orc_cluster_view* view, uint32_t n_objects, const float* pos_range,
                                            const uint8_t* obj_type, const float* spot_dir, const float* spot_sin_cos, int iters,
                                            uint64_t* total_out, float* farthest_z_out) {
    uint32_t C = view->dims[0] * view->dims[1] * view->dims[2];
    m4 view_from_world = m4_load(view->view_from_world);
    m4 clip_from_view = m4_load(view->clip_from_view);
    v4 row2 = m4_row(&view_from_world, 2);
    float* spheres = (float*)calloc((size_t)C * 4 + 4, sizeof(float));
    uint8_t* valid = (uint8_t*)calloc((size_t)C + 1, 1);
    push_vec* clusters = (push_vec*)calloc((size_t)C + 1, sizeof(push_vec));
    push_ctx pc = {clusters, 0};
    float farthest = 0.0f;
    struct timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    for (int it = 0; it < iters; ++it) {
        for (uint32_t c = 0; c < C; ++c) { clusters[c].len = 0; memset(clusters[c].counts, 0, sizeof clusters[c].counts); }
        pc.total = 0;
        farthest = 0.0f;
        for (uint32_t i = 0; i < n_objects; ++i) {
            v3 center = V3(pos_range[4 * (size_t)i], pos_range[4 * (size_t)i + 1], pos_range[4 * (size_t)i + 2]);
            assign_one_object(view, &view_from_world, &clip_from_view, row2, spheres, valid, i, center,
                              pos_range[4 * (size_t)i + 3], obj_type ? obj_type[i] : ORC_OBJ_POINT_LIGHT, 1u,
                              spot_dir ? spot_dir + 3 * (size_t)i : NULL, spot_sin_cos ? spot_sin_cos + 2 * (size_t)i : NULL,
                              &farthest, emit_push, &pc);
        }
    }
    clock_gettime(CLOCK_MONOTONIC, &b);
    if (total_out) *total_out = pc.total;
    if (farthest_z_out) *farthest_z_out = farthest;
    for (uint32_t c = 0; c < C; ++c) free(clusters[c].data);
    free(clusters); free(spheres); free(valid);
    return (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
}

void orc_mesh_inputs(const float g[12], const float c[3], const float h[3], int has_aabb, float wfl[12], float cull[8]) {
    /* transpose_3x3.{x,y,z}_axis.extend(translation.{x,y,z}) */
    for (int r = 0; r < 3; ++r) {
        wfl[4 * r + 0] = g[0 + r];
        wfl[4 * r + 1] = g[3 + r];
        wfl[4 * r + 2] = g[6 + r];
        wfl[4 * r + 3] = g[9 + r];
    }
    if (has_aabb) {
        cull[0] = c[0]; cull[1] = c[1]; cull[2] = c[2]; cull[3] = 0.0f;
        cull[4] = h[0]; cull[5] = h[1]; cull[6] = h[2]; cull[7] = 0.0f;
    } else {
        cull[0] = cull[1] = cull[2] = cull[3] = 0.0f;
        cull[4] = cull[5] = cull[6] = INFINITY; cull[7] = 0.0f;
    }
}

void orc_cluster_bindings_storage(uint32_t n_clusters, const uint32_t* offsets, const uint32_t* counts,
                                  const uint32_t* indices, const uint32_t* remap,
                                  uint32_t* out_oc, uint32_t* out_idx) {
    uint32_t n_indices = 0; /* ViewClusterBindings::n_indices */
    for (uint32_t c = 0; c < n_clusters; ++c) {
        /* ClusterHeader -> push_offset_and_counts(n_indices, counts) */
        out_oc[8 * c + 0] = n_indices;
        out_oc[8 * c + 1] = counts[6 * c + 0]; /* point_lights */
        out_oc[8 * c + 2] = counts[6 * c + 1]; /* spot_lights */
        out_oc[8 * c + 3] = counts[6 * c + 2]; /* rect_lights */
        out_oc[8 * c + 4] = counts[6 * c + 3]; /* reflection_probes */
        out_oc[8 * c + 5] = counts[6 * c + 4]; /* irradiance_volumes */
        out_oc[8 * c + 6] = counts[6 * c + 5]; /* decals */
        out_oc[8 * c + 7] = 0;
        for (uint32_t i = offsets[c]; i < offsets[c + 1]; ++i) {
            uint32_t obj = indices[i];
            out_idx[n_indices++] = remap ? remap[obj] : obj; /* push_index / push_dummy_index (!0) */
        }
    }
}

/* ======================================================================================= */
/* CPU baseline driver: Bevy-shaped par_iter (one batch of ceil(n/threads) rows per thread,  */
/* crates/bevy_ecs/src/batching.rs:95-106), systems run back-to-back with a join in between  */
/* ======================================================================================= */

typedef struct {
    int phase; /* 0 sync, 1 reset, 2 check (view), 3 mark */
    uint32_t lo, hi, n, view;
    const float *t, *r, *s, *c, *h;
    const uint8_t* flags;
    const uint32_t* layers;
    float* g;
    uint8_t *vv, *vis;
    const float* frusta;
    const uint32_t* vmasks;
    const uint8_t* vflags;
} job_t;

static void* job_run(void* p) {
    job_t* j = (job_t*)p;
    uint32_t lo = j->lo, cnt = j->hi - j->lo;
    switch (j->phase) {
    case 0:
        orc_sync_simple_transforms(cnt, j->t + 3 * (size_t)lo, j->r + 4 * (size_t)lo, j->s + 3 * (size_t)lo, NULL,
                                   j->g + 12 * (size_t)lo, NULL);
        break;
    case 1:
        orc_reset_view_visibility(cnt, j->flags + lo, j->vv + lo);
        break;
    case 2: {
        const float* frustum = j->frusta + 24 * (size_t)j->view;
        uint32_t vm = j->vmasks ? j->vmasks[j->view] : 1u;
        int ncc = j->vflags ? (j->vflags[j->view] & 1) : 0;
        for (uint32_t i = lo; i < j->hi; ++i) {
            uint8_t fl = j->flags[i], vis = 0;
            if (!(fl & ORC_FLAG_NO_CPU_CULLING)) {
                vis = (uint8_t)entity_visible_in_view(j->g + 12 * (size_t)i, j->c + 3 * (size_t)i, j->h + 3 * (size_t)i, fl,
                                                      j->layers ? j->layers[i] : 1u, 1, frustum, vm, ncc);
                if (vis) set_visible(&j->vv[i], NULL);
            }
            j->vis[(size_t)j->view * j->n + i] = vis;
        }
        break;
    }
    case 3:
        orc_mark_newly_hidden(cnt, j->flags + lo, j->vv + lo, NULL);
        break;
    case 4: {
        /* reset_view_visibility + check_visibility (every view) + mark_newly_hidden_entities_invisible in ONE pass over the
         * batch: row-independent systems run back to back give the same bytes, and the batch stays in cache between them --
         * the kindest reading of what a task pool does with three par_iter systems over the same tables (bench.py's
         * "fused_visibility" CPU baseline; the reference runs them as separate systems with a join in between). */
        for (uint32_t i = lo; i < j->hi; ++i) {
            uint8_t fl = j->flags[i];
            if (!(fl & ORC_FLAG_NO_CPU_CULLING)) j->vv[i] = (uint8_t)((j->vv[i] & 1u) << 1);
            for (uint32_t v = 0; v < j->view; ++v) { /* j->view = number of views in this phase */
                uint8_t vis = 0;
                if (!(fl & ORC_FLAG_NO_CPU_CULLING)) {
                    vis = (uint8_t)entity_visible_in_view(j->g + 12 * (size_t)i, j->c + 3 * (size_t)i, j->h + 3 * (size_t)i, fl,
                                                          j->layers ? j->layers[i] : 1u, 1, j->frusta + 24 * (size_t)v,
                                                          j->vmasks ? j->vmasks[v] : 1u, j->vflags ? (j->vflags[v] & 1) : 0);
                    if (vis) set_visible(&j->vv[i], NULL);
                }
                j->vis[(size_t)v * j->n + i] = vis;
            }
        }
        orc_mark_newly_hidden(cnt, j->flags + lo, j->vv + lo, NULL);
        break;
    }
    }
    return NULL;
}

/* A persistent pool, like Bevy's ComputeTaskPool (threads are not created per system run): the workers park on a
 * barrier, run their batch of the current phase, and meet the caller on a second barrier. */
typedef struct pool_t {
    job_t* jobs;
    int threads;
    volatile int stop;
    pthread_barrier_t start, done;
} pool_t;
typedef struct worker_arg_t { pool_t* pool; int k; } worker_arg_t;

static void* pool_worker(void* p) {
    worker_arg_t* wa = (worker_arg_t*)p;
    for (;;) {
        pthread_barrier_wait(&wa->pool->start);
        if (wa->pool->stop) return NULL;
        job_run(&wa->pool->jobs[wa->k]);
        pthread_barrier_wait(&wa->pool->done);
    }
}

static void run_phase(pool_t* pool) {
    pthread_barrier_wait(&pool->start);
    job_run(&pool->jobs[0]);
    pthread_barrier_wait(&pool->done);
}

double orc_bench_flat_frame2(uint32_t n, const float* t, const float* r, const float* s, const float* c,
                             const float* h, const uint8_t* flags, const uint32_t* layers, float* g, uint8_t* vv,
                             uint8_t* vis, const float* frusta, const uint32_t* vmasks, const uint8_t* vflags,
                             uint32_t n_views, int threads, int iters, int fused_visibility);
double orc_bench_flat_frame(uint32_t n, const float* t, const float* r, const float* s, const float* c,
                            const float* h, const uint8_t* flags, const uint32_t* layers, float* g, uint8_t* vv,
                            uint8_t* vis, const float* frusta, const uint32_t* vmasks, const uint8_t* vflags,
                            uint32_t n_views, int threads, int iters) {
    return orc_bench_flat_frame2(n, t, r, s, c, h, flags, layers, g, vv, vis, frusta, vmasks, vflags, n_views, threads, iters, 0);
}
double orc_bench_flat_frame2(uint32_t n, const float* t, const float* r, const float* s, const float* c,
                             const float* h, const uint8_t* flags, const uint32_t* layers, float* g, uint8_t* vv,
                             uint8_t* vis, const float* frusta, const uint32_t* vmasks, const uint8_t* vflags,
                             uint32_t n_views, int threads, int iters, int fused_visibility) {
    if (threads < 1) threads = 1;
    job_t* jobs = (job_t*)calloc((size_t)threads, sizeof(job_t));
    pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
    worker_arg_t* wargs = (worker_arg_t*)calloc((size_t)threads, sizeof(worker_arg_t));
    pool_t pool;
    pool.jobs = jobs;
    pool.threads = threads;
    pool.stop = 0;
    pthread_barrier_init(&pool.start, NULL, (unsigned)threads);
    pthread_barrier_init(&pool.done, NULL, (unsigned)threads);
    uint32_t batch = (n + (uint32_t)threads - 1) / (uint32_t)threads;
    for (int k = 0; k < threads; ++k) {
        job_t* j = &jobs[k];
        uint64_t lo = (uint64_t)batch * (uint64_t)k, hi = lo + batch;
        j->lo = (uint32_t)(lo > n ? n : lo); j->hi = (uint32_t)(hi > n ? n : hi); j->n = n;
        j->t = t; j->r = r; j->s = s; j->c = c; j->h = h; j->flags = flags; j->layers = layers;
        j->g = g; j->vv = vv; j->vis = vis; j->frusta = frusta; j->vmasks = vmasks; j->vflags = vflags;
    }
    for (int k = 1; k < threads; ++k) {
        wargs[k].pool = &pool;
        wargs[k].k = k;
        pthread_create(&th[k], NULL, pool_worker, &wargs[k]);
    }
    struct timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    for (int it = 0; it < iters; ++it) {
        for (int k = 0; k < threads; ++k) jobs[k].phase = 0;
        run_phase(&pool);
        if (fused_visibility) {
            for (int k = 0; k < threads; ++k) { jobs[k].phase = 4; jobs[k].view = n_views; }
            run_phase(&pool);
            continue;
        }
        for (int k = 0; k < threads; ++k) jobs[k].phase = 1;
        run_phase(&pool);
        for (uint32_t v = 0; v < n_views; ++v) {
            for (int k = 0; k < threads; ++k) { jobs[k].phase = 2; jobs[k].view = v; }
            run_phase(&pool);
        }
        for (int k = 0; k < threads; ++k) jobs[k].phase = 3;
        run_phase(&pool);
    }
    clock_gettime(CLOCK_MONOTONIC, &b);
    pool.stop = 1;
    pthread_barrier_wait(&pool.start);
    for (int k = 1; k < threads; ++k) pthread_join(th[k], NULL);
    pthread_barrier_destroy(&pool.start);
    pthread_barrier_destroy(&pool.done);
    free(jobs); free(th); free(wargs);
    return (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
}

/* ---- parallel CPU baseline for the hierarchy (bench.py only) ---------------------------------------------------
 * propagate_parent_transforms fans out over the task pool (par_iter over roots, then a work queue of subtrees,
 * systems.rs:506-640).  Rows here are in level (BFS) order, so the same parallelism is available as "all rows of a
 * level in parallel, levels in order" on the persistent pool above; every node does exactly what
 * orc_propagate_transforms does for it with every Transform changed (aff_from_srt, parent * local, set_if_neq). */
typedef struct tree_job_t {
    uint32_t lo, hi, root_level;
    const uint32_t* parent;
    const float *t, *r, *s;
    float* g;
} tree_job_t;
typedef struct tree_pool_t {
    tree_job_t* jobs;
    volatile int stop;
    pthread_barrier_t start, done;
} tree_pool_t;
typedef struct tree_arg_t { tree_pool_t* pool; int k; } tree_arg_t;

static void tree_job_run(const tree_job_t* j) {
    for (uint32_t i = j->lo; i < j->hi; ++i) {
        aff local = aff_from_srt(j->s + 3 * (size_t)i, j->r + 4 * (size_t)i, j->t + 3 * (size_t)i);
        if (j->root_level || j->parent[i] == ORC_NO_PARENT) {
            aff_store(&local, j->g + 12 * (size_t)i);
            continue;
        }
        aff gp = aff_load(j->g + 12 * (size_t)j->parent[i]);
        aff gc = aff_mul(&gp, &local);
        float tmp[12];
        aff_store(&gc, tmp);
        if (!aff_eq(tmp, j->g + 12 * (size_t)i)) memcpy(j->g + 12 * (size_t)i, tmp, sizeof tmp);
    }
}
static void* tree_worker(void* p) {
    tree_arg_t* a = (tree_arg_t*)p;
    for (;;) {
        pthread_barrier_wait(&a->pool->start);
        if (a->pool->stop) return NULL;
        tree_job_run(&a->pool->jobs[a->k]);
        pthread_barrier_wait(&a->pool->done);
    }
}

double orc_bench_tree_frame(uint32_t n, const uint32_t* parent, const uint32_t* level_offsets, uint32_t n_levels, const float* t,
                            const float* r, const float* s, float* g, int threads, int iters) {
    if (threads < 1) threads = 1;
    if (threads == 1) { /* one core: rows in level order ARE a valid sequential order -- one sweep, no pool, no barriers (what a
                           deep narrow hierarchy wants: a chain is 2 500 levels of one row) */
        tree_job_t j;
        j.parent = parent; j.t = t; j.r = r; j.s = s; j.g = g;
        struct timespec a1, b1;
        clock_gettime(CLOCK_MONOTONIC, &a1);
        for (int it = 0; it < iters; ++it) {
            j.lo = level_offsets[0]; j.hi = level_offsets[1]; j.root_level = 1;
            tree_job_run(&j);
            j.lo = level_offsets[1]; j.hi = level_offsets[n_levels]; j.root_level = 0;
            if (n_levels > 1) tree_job_run(&j);
        }
        clock_gettime(CLOCK_MONOTONIC, &b1);
        return (double)(b1.tv_sec - a1.tv_sec) + 1e-9 * (double)(b1.tv_nsec - a1.tv_nsec);
    }
    tree_job_t* jobs = (tree_job_t*)calloc((size_t)threads, sizeof(tree_job_t));
    pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
    tree_arg_t* args = (tree_arg_t*)calloc((size_t)threads, sizeof(tree_arg_t));
    tree_pool_t pool;
    pool.jobs = jobs;
    pool.stop = 0;
    pthread_barrier_init(&pool.start, NULL, (unsigned)threads);
    pthread_barrier_init(&pool.done, NULL, (unsigned)threads);
    for (int k = 0; k < threads; ++k) { jobs[k].parent = parent; jobs[k].t = t; jobs[k].r = r; jobs[k].s = s; jobs[k].g = g; }
    for (int k = 1; k < threads; ++k) {
        args[k].pool = &pool;
        args[k].k = k;
        pthread_create(&th[k], NULL, tree_worker, &args[k]);
    }
    (void)n;
    struct timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    for (int it = 0; it < iters; ++it)
        for (uint32_t l = 0; l < n_levels; ++l) {
            const uint32_t lo = level_offsets[l], cnt = level_offsets[l + 1] - lo;
            const uint32_t batch = (cnt + (uint32_t)threads - 1) / (uint32_t)threads;
            for (int k = 0; k < threads; ++k) {
                uint64_t a0 = (uint64_t)batch * (uint64_t)k, a1 = a0 + batch;
                jobs[k].lo = lo + (uint32_t)(a0 > cnt ? cnt : a0);
                jobs[k].hi = lo + (uint32_t)(a1 > cnt ? cnt : a1);
                jobs[k].root_level = l == 0;
            }
            pthread_barrier_wait(&pool.start);
            tree_job_run(&jobs[0]);
            pthread_barrier_wait(&pool.done);
        }
    clock_gettime(CLOCK_MONOTONIC, &b);
    pool.stop = 1;
    pthread_barrier_wait(&pool.start);
    for (int k = 1; k < threads; ++k) pthread_join(th[k], NULL);
    pthread_barrier_destroy(&pool.start);
    pthread_barrier_destroy(&pool.done);
    free(jobs); free(th); free(args);
    return (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
}
