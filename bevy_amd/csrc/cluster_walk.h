// cluster_walk.h -- the first half of the light-cluster assignment as device code: everything one workgroup does for its 256
// objects (see kernels_cluster.hip for the algorithm).  A header because two kernels run it: k_cluster_walk on its own, and the
// frame kernel (kernels_flat.hip), where the walk of a MI_CULL_WITH_CLUSTERS frame rides in extra workgroups of the frame's
// own launch -- it re-derives the lights' ViewVisibility with the cull's rule (visibility_rule.h), so it depends on nothing the
// rest of that launch produces.
#pragma once
#include "glam_math.h"
#include "kernels.h"
#include "visibility_rule.h"

namespace mi {

#ifdef MI_EXP_TIMELINE
// (experiment build) phase timestamps of the walking workgroups: [8b + k], see tools/exp_timeline.py
__device__ unsigned long long mi_walk_marks[16 * 4096];
#define MI_WALK_MARK(b, k)                                                                  \
    do {                                                                                    \
        if (threadIdx.x == 0 && (b) < 4096u) mi_walk_marks[16u * (b) + (k)] = wall_clock64(); \
    } while (0)
#else
#define MI_WALK_MARK(b, k)
#endif


// glibc >= 2.28 logf (ARM optimized-routines algorithm, table size 16, degree-3 polynomial in
// double).  Rust's f32::ln is the platform libm's logf (bevy_math/src/ops.rs:22-60), so this is the
// function view_z_to_z_slice (assign.rs:1057) evaluates on the reference's CPU path.  Verified
// bit-identical to libm logf for every non-negative binary32 (tests/test_gpu_parity.py::test_device_logf_matches_libm keeps a sample; glibc >= 2.28 on x86-64 is the supported host libm).
static __device__ __constant__ double LOGF_TAB[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
    {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},  {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
    {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
    {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1p+0, 0x0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
    {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3},
    {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},  {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};

// tab: the 16 x 2 table -- LOGF_TAB itself, or a copy (the walk keeps one in LDS: read from memory the table is a round trip per
// call, microseconds while the frame's rows keep HBM busy)
__device__ __forceinline__ float libm_logf(float x, const double* tab = &LOGF_TAB[0][0]) {
    uint32_t ix = __float_as_uint(x);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
        if (ix * 2u == 0u) return -INFINITY;
        if (ix == 0x7f800000u) return x;
        if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return NAN;
        ix = __float_as_uint(x * 0x1p23f);
        ix -= 23u << 23;
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) & 15u);
    const int k = (int32_t)tmp >> 23;
    const uint32_t iz = ix - (tmp & 0xff800000u);
    const double invc = tab[2 * i], logc = tab[2 * i + 1];
    const double z = (double)__uint_as_float(iz);
    const double r = z * invc - 1.0;
    const double y0 = logc + (double)k * 0x1.62e42fefa39efp-1;
    const double r2 = r * r;
    double y = 0x1.5575b0be00b6ap-2 * r + -0x1.ffffef20a4123p-2;
    y = -0x1.00ea348b88334p-2 * r2 + y;
    y = y * r2 + (y0 + r);
    return (float)y;
}
struct Sphere {
    V3 center;
    float radius;
};

// view_z_to_z_slice, assign.rs:1046-1062
__device__ __forceinline__ uint32_t view_z_to_z_slice(const ClusterViewDev& v, float view_z, const double* logf_tab) {
    uint32_t z_slice;
    if (v.is_orthographic) z_slice = f32_as_u32(floorf((view_z - v.cluster_factors[0]) * v.cluster_factors[1]));
    else z_slice = f32_as_u32(libm_logf(-view_z, logf_tab) * v.cluster_factors[0] - v.cluster_factors[1] + 1.0f);
    const uint32_t lim = v.dims[2] - 1u;
    return z_slice < lim ? z_slice : lim;
}
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return lane_min(lane_max(x, lo), hi); }
// ndc_position_to_cluster, assign.rs:922-941
__device__ __forceinline__ void ndc_position_to_cluster(const ClusterViewDev& v, float ndc_x, float ndc_y, float view_z,
                                                        uint32_t out[3], const double* logf_tab) {
    const float fx = clampf(ndc_x * 0.5f + 0.5f, 0.0f, 1.0f);
    const float fy = clampf(ndc_y * -0.5f + 0.5f, 0.0f, 1.0f);
    const uint32_t xi = f32_as_u32(floorf(fx * (float)v.dims[0]));
    const uint32_t yi = f32_as_u32(floorf(fy * (float)v.dims[1]));
    const uint32_t zs = view_z_to_z_slice(v, view_z, logf_tab);
    out[0] = xi > v.dims[0] - 1u ? v.dims[0] - 1u : xi;
    out[1] = yi > v.dims[1] - 1u ? v.dims[1] - 1u : yi;
    out[2] = zs > v.dims[2] - 1u ? v.dims[2] - 1u : zs;
}
__device__ __forceinline__ V4 ldp(const float* planes, uint32_t i) {
    const float4 p = reinterpret_cast<const float4*>(planes)[i];
    return V4{p.x, p.y, p.z, p.w};
}
// project_to_plane_z, assign.rs:1094-1113
__device__ __forceinline__ bool project_to_plane_z(Sphere& s, V4 plane) {
    const float z = f_div(plane.w, plane.z);
    const float dist = z - s.center.z;
    if (f_abs(dist) > s.radius) return false;
    s.center.z = z;
    s.radius = f_sqrt(s.radius * s.radius - dist * dist);
    return true;
}
// project_to_plane_y, assign.rs:1116-1134
__device__ __forceinline__ bool project_to_plane_y(Sphere& s, V4 plane, bool ortho) {
    float dist;
    if (ortho) dist = plane.w - s.center.y;
    else dist = -(s.center.y * plane.y + s.center.z * plane.z);
    if (f_abs(dist) > s.radius) return false;
    s.center = s.center + xyz(plane) * dist;
    s.radius = f_sqrt(s.radius * s.radius - dist * dist);
    return true;
}
// get_distance_x, assign.rs:1081-1091
__device__ __forceinline__ float get_distance_x(V4 plane, V3 p, bool ortho) {
    if (ortho) return p.x - plane.w;
    return plane.x * p.x + plane.z * p.z;
}

// The body of `for clusterable_object in &clusterable_objects` (assign.rs:487-804) for one object.
// emit(cluster_index) is called for every cluster the reference would push this object into.
// The two early-outs at the top of the per-object loop (assign.rs:489 RenderLayers, :496 frustum vs light sphere).
struct F3c {
    float x, y, z;
};
__device__ __forceinline__ V3 ld3c(const float* base, uint32_t row) {
    const F3c v = reinterpret_cast<const F3c*>(base)[row];
    return V3{v.x, v.y, v.z};
}
// The row's GlobalTransform: the column (behind this frame's propagate) or, in derive mode and for a row this frame's propagate
// writes, From(Transform) itself (sync_simple_transforms: the rows are flat).
// derive mode: does this frame's propagate write the row's GlobalTransform (then it is From(Transform), the column being written
// by the same launch) or does the row keep the resident one?  The fused all-rows frame writes every row; the changed-rows frame
// the rows whose change byte is set; a cull-only frame none.
// the row object `obj` is bound to: first_row + obj, or -- mi_cluster_bind_objects_to_row_list -- row_list[obj]
__device__ __forceinline__ uint32_t object_row(const ClusterObjects& o, uint32_t obj) {
    return o.row_list ? o.row_list[obj] : o.first_row + obj;
}
__device__ __forceinline__ bool object_row_is_propagated(const ClusterObjects& o, uint32_t row) {
    return !o.derive_resident && (!o.row_changed || row_changed(o.row_changed[row], o.changed_gen));
}
__device__ __forceinline__ Affine object_row_affine(const ClusterObjects& o, uint32_t row) {
    if (o.derive && object_row_is_propagated(o, row)) {
        const float4 q = reinterpret_cast<const float4*>(o.row_rotation)[row];
        return affine_from_srt(ld3c(o.row_scale, row), V4{q.x, q.y, q.z, q.w}, ld3c(o.row_translation, row));
    }
    return load_affine(o.row_global + 12ull * row);
}
// ViewVisibility::get() of a light row after this frame's visibility systems, without waiting for them: reset, then
// set_visible() by any view of the frame (check_visibility_cpu_culling, visibility/mod.rs:788-858), or -- NoCpuCulling rows --
// check_visibility_gpu_culling (:884-903).
// Every load of the row is issued in ONE batch, whether the row turns out to need it or not (a NoCpuCulling row never reads its
// bounds): riding in the frame kernel a round trip to memory costs microseconds -- the rows of the launch keep HBM saturated --
// and flags -> Transform -> sphere as three dependent trips were most of what the walk added to the frame.  *g_out = the
// GlobalTransform the frame gives the row.
__device__ __forceinline__ bool derive_row_visible(const ClusterObjects& o, const ViewSet& views, uint32_t row, Affine* g_out) {
    // The rows' RowSummary (kernels.h) first: lights are entities of one kind, spawned together -- their Aabb / Sphere, flags and
    // RenderLayers agree over whole waves of rows, and a row without an Aabb needs of its GlobalTransform only the translation
    // (visibility_rule.h: the Sphere branch).  Where every lane of the wave finds its rows summarised, the walk reads 12 + 16 bytes
    // per light (translation, pos_range) instead of 85.
    uint32_t fl = 0, emask = 0, emask_hi = 0;
    V3 center = {}, half = {};
    bool summarised = false;
    if (o.row_summary && !o.derive_resident && !o.row_changed) {  // (uniform)
        const uint4* p = reinterpret_cast<const uint4*>(o.row_summary) + 2ull * (row >> 6);
        const uint4 a = p[0], b = p[1];
        summarised = (b.w & (ROWSUM_UNIFORM_AABB | ROWSUM_UNIFORM_FLAGS)) == (ROWSUM_UNIFORM_AABB | ROWSUM_UNIFORM_FLAGS) && !(b.w & 0x04u);
        fl = b.w & 0xFFu;
        emask = b.z;
        center = V3{__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z)};
        half = V3{__uint_as_float(a.w), __uint_as_float(b.x), __uint_as_float(b.y)};
    }
    Affine g;
    if (__ballot(summarised) == __ballot(true)) {  // every active lane: Transform -> translation only, nothing else to fetch
        g = Affine{M3{V3{1.0f, 0.0f, 0.0f}, V3{0.0f, 1.0f, 0.0f}, V3{0.0f, 0.0f, 1.0f}}, ld3c(o.row_translation, row)};  // (the matrix is never read: no Aabb)
    } else {
        fl = o.row_flags[row];
        emask = o.row_layers[row];
        if (o.row_layers_hi) emask_hi = o.row_layers_hi[row];
        center = ld3c(o.row_aabb_center, row);
        half = ld3c(o.row_aabb_half, row);
        if (o.derive_resident) {  // (uniform) a cull-only frame: every row keeps its resident GlobalTransform
            g = load_affine(o.row_global + 12ull * row);
        } else if (!o.row_changed) {  // (uniform) the all-rows frame: From(Transform)
            const float4 q = reinterpret_cast<const float4*>(o.row_rotation)[row];
            g = affine_from_srt(ld3c(o.row_scale, row), V4{q.x, q.y, q.z, q.w}, ld3c(o.row_translation, row));
        } else {  // the changed-rows frame: one more trip, by the row's change byte
            g = object_row_affine(o, row);
        }
    }
    float range_lo = 0.0f, range_hi = 0.0f;
    if (o.row_range && (fl & 0x20u)) {
        const float2 r2 = reinterpret_cast<const float2*>(o.row_range)[row];
        range_lo = r2.x;
        range_hi = r2.y;
    }
    *g_out = g;
    if (fl & 0x10u) return (fl & 0x01u) != 0;
    bool any = false;
    for (uint32_t v = 0; v < o.n_views; ++v)
        any = any || row_visible_in_view(g, center, half, fl, emask, emask_hi, o.row_range != nullptr, range_lo, range_hi, views.v[v]);
    return any;
}
// The two early-outs at the top of the per-object loop (assign.rs:489 RenderLayers, :496 frustum vs light sphere) behind the
// gather's `if view_visibility.get()` (:194).  *sphere_out = ClusterableObjectAssignmentData::sphere of the object, which the
// caller keeps: the walk needs it again, and fetching it again is one more trip.
__device__ __forceinline__ bool object_in_view(const ClusterViewDev& v, const ClusterObjects& o, const ViewSet& views, uint32_t obj, float4* sphere_out) {
    float4 pr = reinterpret_cast<const float4*>(o.pos_range)[obj];  // (issued with the row's loads below)
    const uint32_t layers = o.layer_mask ? o.layer_mask[obj] : 1u, layers_hi = o.layer_mask_hi ? o.layer_mask_hi[obj] : 0u;
    bool visible = true;
    if (o.derive) {
        Affine g;
        visible = derive_row_visible(o, views, object_row(o, obj), &g);
        pr.x = g.t.x;  // point lights: GlobalTransform::from_translation(transform.translation()), assign.rs:198
        pr.y = g.t.y;
        pr.z = g.t.z;
    } else {
        if (o.row_vv) visible = (o.row_vv[object_row(o, obj)] & 1u) != 0;
        if (o.row_global) {
            const float* g = o.row_global + 12ull * object_row(o, obj);
            pr.x = g[9];
            pr.y = g[10];
            pr.z = g[11];
        }
    }
    *sphere_out = pr;
    if (!visible) return false;
    if (!((v.view_layer_mask & layers) | (v.view_layer_mask_hi & layers_hi))) return false;  // :489, RenderLayers::intersects over the first u64 word
    V4 fr[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) fr[i] = V4{v.frustum[4 * i], v.frustum[4 * i + 1], v.frustum[4 * i + 2], v.frustum[4 * i + 3]};
    return frustum_intersects_sphere(fr, V3{pr.x, pr.y, pr.z}, pr.w, true);  // :496
}

// The rest of the body for an object that passed object_in_view, in two parts: what the reference computes once per object
// in front of its z loop (assign.rs:501-605) and the z -> y -> x-range refinement itself (:606-800), which can be run for
// any sub-range of z slices -- the walk kernel sweeps the grid in z chunks that fit a small LDS.
struct ObjectWalk {
    uint32_t minc[3], maxc[3];
    Sphere vs;
    float range;
    uint32_t type;
    V3 light_dir;
    float angle_sin, angle_cos;
    uint32_t z_center, y_center;
    bool z_center_some, y_center_some;
};

// The z part of object_setup alone -- the z slices the object's cluster-space AABB spans (the same expressions, :948-1036) and its
// far_z (:558-560): what a chunked walk needs of every object before its z loop, at a fraction of the arithmetic of the whole setup.
__device__ __forceinline__ void object_z_extent(const ClusterViewDev& v, float4 pr, uint32_t* z_lo, uint32_t* z_hi, float* far_z_out,
                                                const double* logf_tab) {
    const V3 center = V3{pr.x, pr.y, pr.z};
    const float range = pr.w;
    const M4 view_from_world = load_m4(v.view_from_world);
    const V3 scale = V3{v.view_from_world_scale[0], v.view_from_world_scale[1], v.view_from_world_scale[2]};
    const V3 cv = xyz(mul(view_from_world, extend(center, 1.0f)));
    const V3 he = abs3(scale) * range;
    const float NEG_MIN_POS = -1.17549435e-38f;
    const float zmin = rust_min((cv - he).z, NEG_MIN_POS), zmax = rust_min((cv + he).z, NEG_MIN_POS);
    const uint32_t lim = v.dims[2] - 1u;
    uint32_t a = view_z_to_z_slice(v, zmin, logf_tab), b = view_z_to_z_slice(v, zmax, logf_tab);
    a = a > lim ? lim : a;
    b = b > lim ? lim : b;
    *z_lo = a < b ? a : b;
    *z_hi = a > b ? a : b;
    *far_z_out = -dot4(row(view_from_world, 2), extend(center, 1.0f)) + range * scale.z;
}

template <bool SPOTS = true>
__device__ __forceinline__ ObjectWalk object_setup(const ClusterViewDev& v, const ClusterObjects& o, uint32_t obj, float4 pr, float* far_z_out,
                                                   const double* logf_tab) {
    ObjectWalk ow;
    const V3 center = V3{pr.x, pr.y, pr.z};
    const float range = pr.w;
    ow.range = range;
    ow.type = o.obj_type ? o.obj_type[obj] : 0u;

    const M4 view_from_world = load_m4(v.view_from_world);
    const M4 clip_from_view = load_m4(v.clip_from_view);
    const V3 scale = V3{v.view_from_world_scale[0], v.view_from_world_scale[1], v.view_from_world_scale[2]};

    // cluster_space_clusterable_object_aabb, :948-1036
    const V3 cv = xyz(mul(view_from_world, extend(center, 1.0f)));
    const V3 he = abs3(scale) * range;
    V3 vmin = cv - he, vmax = cv + he;
    const float NEG_MIN_POS = -1.17549435e-38f;
    vmin.z = rust_min(vmin.z, NEG_MIN_POS);
    vmax.z = rust_min(vmax.z, NEG_MIN_POS);
    const V3 corner[4] = {vmin, V3{vmin.x, vmin.y, vmax.z}, V3{vmax.x, vmax.y, vmin.z}, vmax};
    V3 nmin, nmax;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const V4 clip = mul(clip_from_view, extend(corner[i], 1.0f));
        const V3 ndc = V3{f_div(clip.x, clip.w), f_div(clip.y, clip.w), f_div(clip.z, clip.w)};
        if (i == 0) { nmin = ndc; nmax = ndc; }
        else { nmin = min3(nmin, ndc); nmax = max3(nmax, ndc); }
    }
    uint32_t c0[3], c1[3];
    ndc_position_to_cluster(v, clampf(nmin.x, -1.0f, 1.0f), clampf(nmin.y, -1.0f, 1.0f), vmin.z, c0, logf_tab);
    ndc_position_to_cluster(v, clampf(nmax.x, -1.0f, 1.0f), clampf(nmax.y, -1.0f, 1.0f), vmax.z, c1, logf_tab);
#pragma unroll
    for (int k = 0; k < 3; ++k) { ow.minc[k] = c0[k] < c1[k] ? c0[k] : c1[k]; ow.maxc[k] = c0[k] > c1[k] ? c0[k] : c1[k]; }

    ow.vs.center = cv;  // same expression as :552-554
    ow.vs.radius = range * v.view_from_world_scale_max;

    *far_z_out = -dot4(row(view_from_world, 2), extend(center, 1.0f)) + range * scale.z;  // :558-560

    ow.light_dir = V3{0.0f, 0.0f, 0.0f};
    ow.angle_sin = 0.0f;
    ow.angle_cos = 0.0f;
    if (SPOTS && ow.type == 1u) {  // spot light, :563-573
        V3 d;
        if (o.row_global || o.derive) {  // GlobalTransform::back() = (matrix3 * Vec3::Z).normalize(), global_transform.rs:62-68,206
            const Affine ga = object_row_affine(o, object_row(o, obj));
            const V3 z = mul(ga.m, V3{0.0f, 0.0f, 1.0f});
            d = z * f_div(1.0f, f_sqrt((z.x * z.x + z.y * z.y) + z.z * z.z));
        } else {
            d = V3{o.spot_dir[3 * obj], o.spot_dir[3 * obj + 1], o.spot_dir[3 * obj + 2]};
        }
        const V3 dv = xyz(mul(view_from_world, extend(d, 0.0f)));
        ow.light_dir = dv * f_div(1.0f, f_sqrt(dot3(dv, dv)));
        ow.angle_sin = o.spot_sin_cos[2 * obj];
        ow.angle_cos = o.spot_sin_cos[2 * obj + 1];
    }
    const V4 center_clip = mul(clip_from_view, extend(ow.vs.center, 1.0f));
    const V3 ndc = V3{f_div(center_clip.x, center_clip.w), f_div(center_clip.y, center_clip.w),
                      f_div(center_clip.z, center_clip.w)};
    uint32_t cc[3];
    ndc_position_to_cluster(v, ndc.x, ndc.y, ow.vs.center.z, cc, logf_tab);
    ow.z_center_some = ndc.z <= 1.0f;
    ow.z_center = cc[2];
    ow.y_center = 0;
    if (ndc.y > 1.0f) ow.y_center_some = false;
    else if (ndc.y < -1.0f) { ow.y_center_some = true; ow.y_center = v.dims[1] + 1u; }
    else { ow.y_center_some = true; ow.y_center = cc[1]; }
    return ow;
}

// emit(xy, z) is called for every cluster (y * dims.x + x, z) with z in [z_lo, z_hi] the reference would push this object into.
// xp / yp / zp: the view's x, y, z cluster planes (LDS copies in the kernel below).
template <bool SPOTS = true, typename Emit>
__device__ __forceinline__ void object_walk(const ClusterViewDev& v, const ObjectWalk& ow, uint32_t z_lo, uint32_t z_hi, const float* xp,
                                            const float* yp, const float* zp, Emit emit) {
    const bool ortho = v.is_orthographic != 0;
    const uint32_t za = ow.minc[2] > z_lo ? ow.minc[2] : z_lo, zb = ow.maxc[2] < z_hi ? ow.maxc[2] : z_hi;
    for (uint32_t z = za; z <= zb && z >= za; ++z) {
        Sphere z_object = ow.vs;
        if (!ow.z_center_some || z != ow.z_center) {
            const V4 z_plane = (ow.z_center_some && z < ow.z_center) ? ldp(zp, z + 1u) : ldp(zp, z);
            if (!project_to_plane_z(z_object, z_plane)) continue;
        }
        for (uint32_t y = ow.minc[1]; y <= ow.maxc[1]; ++y) {
            Sphere y_object = z_object;
            if (!ow.y_center_some || y != ow.y_center) {
                const V4 y_plane = (ow.y_center_some && y < ow.y_center) ? ldp(yp, y + 1u) : ldp(yp, y);
                if (!project_to_plane_y(y_object, y_plane, ortho)) continue;
            }
            uint32_t min_x = ow.minc[0];
            for (;;) {
                if (min_x >= ow.maxc[0] ||
                    -get_distance_x(ldp(xp, min_x + 1u), y_object.center, ortho) + y_object.radius > 0.0f)
                    break;
                min_x += 1u;
            }
            uint32_t max_x = ow.maxc[0];
            for (;;) {
                if (max_x <= min_x ||
                    get_distance_x(ldp(xp, max_x), y_object.center, ortho) + y_object.radius > 0.0f)
                    break;
                max_x -= 1u;
            }
            const uint32_t xy0 = y * v.dims[0];
            if (SPOTS && ow.type == 1u) {
                for (uint32_t x = min_x; x <= max_x; ++x) {
                    const float4 cs = reinterpret_cast<const float4*>(v.cluster_spheres)[(size_t)(xy0 + x) * v.dims[2] + z];
                    const V3 off = ow.vs.center - V3{cs.x, cs.y, cs.z};
                    const float dist_sq = dot3(off, off);
                    const float v1_len = dot3(off, ow.light_dir);
                    const float dcp = (ow.angle_cos * f_sqrt(dist_sq - v1_len * v1_len)) - v1_len * ow.angle_sin;
                    const bool angle_cull = dcp > cs.w;
                    const bool front_cull = v1_len > cs.w + ow.range * v.view_from_world_scale_max;
                    const bool back_cull = v1_len < -cs.w;
                    if (!angle_cull && !front_cull && !back_cull) emit(xy0 + x, z);
                }
            } else {
                for (uint32_t x = min_x; x <= max_x; ++x) emit(xy0 + x, z);
            }
        }
    }
}


// LDS arena of one walking workgroup.  The grid is swept in chunks of `zc` z slices: one 256-bit row per cluster OF THE CHUNK
// (dims.x * dims.y * zc rows), 6 x 8 words of type masks, the view's planes, and -- per chunk -- one "touched" bit per row, the
// list of touched rows (u16) and its length.  k_cluster_walk normally takes zc = dims.z: the whole grid in one sweep (16x9x24:
// 108 KB of rows); a grid whose rows do not fit the LDS is swept in as many slices at a time as do.  Riding in the frame
// kernel the arena is that kernel's 16 KB, i.e. two or three slices of a 16x9 grid at a time.
// (cluster_walk_lds_bytes in kernels.h sizes this arena.)

// PLANES_IN_LDS = false only for degenerate grids whose plane tables do not fit next to the bit rows.
// CHUNKED = false: zc == dims.z, the whole grid in one sweep (no z-range bookkeeping, row index == cluster index).
// In the chunked form the per-object setup is recomputed for every chunk instead of being kept in registers across the chunk
// loop: 67 instead of 103 VGPRs, which is what lets the walk share a kernel with the frame rows at their occupancy.
// SPOTS = false: the caller guarantees there is no spot light among the objects (the walk riding in the frame kernel: ctx_cluster.cpp
// sends scenes with spot lights to the walk kernel of its own) -- the cone test and its five registers per object drop out.
// What a walking workgroup fetches in front of everything else (with its objects' own loads): its share of the view's plane tables
// (floats threadIdx.x + k * 256; 16x9x24 has 208) and of the logf table, both bound for LDS.
struct WalkPrefetch {
    float pl0, pl1, pl2, pl3;
    double logf_reg;
};
// planes: where the table is read from -- the view's own pointer (the staged copy), or the frame kernel's argument segment (WalkPlanes,
// kernels.h) when the view carries none
template <bool PLANES_IN_LDS>
__device__ __forceinline__ WalkPrefetch walk_prefetch(const ClusterViewDev& v, const float* planes_arg = nullptr) {
    const uint32_t n_plane_floats = PLANES_IN_LDS ? 4u * (v.dims[0] + v.dims[1] + v.dims[2] + 3u) : 0u;
    WalkPrefetch p = {0.f, 0.f, 0.f, 0.f, 0.0};
    const float* const src = v.x_planes ? v.x_planes : planes_arg;
    if (threadIdx.x < n_plane_floats) p.pl0 = src[threadIdx.x];
    if (threadIdx.x + CLUSTER_BLOCK < n_plane_floats) p.pl1 = src[threadIdx.x + CLUSTER_BLOCK];
    if (threadIdx.x + 2u * CLUSTER_BLOCK < n_plane_floats) p.pl2 = src[threadIdx.x + 2u * CLUSTER_BLOCK];
    if (threadIdx.x + 3u * CLUSTER_BLOCK < n_plane_floats) p.pl3 = src[threadIdx.x + 3u * CLUSTER_BLOCK];
    p.logf_reg = (&LOGF_TAB[0][0])[threadIdx.x & 31u];
    return p;
}

template <bool PLANES_IN_LDS, bool CHUNKED, bool SPOTS>
__device__ __forceinline__ void cluster_walk_tail(const ClusterViewDev& v, const ClusterObjects& o, const ClusterWork& w, uint32_t zc_arg, uint32_t bx,
                                                  uint32_t* arena, uint32_t obj, bool in_view, float4 sphere, WalkPrefetch pf);

// One workgroup of the walk over 256 objects: the objects' own tests (object_in_view), then cluster_walk_tail.
template <bool PLANES_IN_LDS, bool CHUNKED, bool SPOTS = true>
__device__ __forceinline__ void cluster_walk_block(const ClusterViewDev& v, const ClusterObjects& o, const ClusterWork& w, const ViewSet& views,
                                                   uint32_t zc_arg, uint32_t bx, uint32_t* arena, const float* planes_arg = nullptr) {
    const uint32_t obj = bx * CLUSTER_BLOCK + threadIdx.x;
    float4 sphere = make_float4(0.f, 0.f, 0.f, 0.f);
    MI_WALK_MARK(bx, 0);
    // the plane tables are requested together with the objects' loads (they are contiguous in device memory: x | y | z)
    const WalkPrefetch pf = walk_prefetch<PLANES_IN_LDS>(v, planes_arg);
    const bool in_view = obj < o.n && object_in_view(v, o, views, obj, &sphere);
    cluster_walk_tail<PLANES_IN_LDS, CHUNKED, SPOTS>(v, o, w, zc_arg, bx, arena, obj, in_view, sphere, pf);
}

// Everything a walking workgroup does once it knows which of its 256 objects are in view (bit k of the block = thread k): called
// by cluster_walk_block, and by the row workgroups of the frame kernel for the objects that ARE their rows (ClusterWalkJob::inrow).
// Every thread of the workgroup must call it.
template <bool PLANES_IN_LDS, bool CHUNKED, bool SPOTS>
__device__ __forceinline__ void cluster_walk_tail(const ClusterViewDev& v, const ClusterObjects& o, const ClusterWork& w, uint32_t zc_arg, uint32_t bx,
                                                  uint32_t* arena, uint32_t obj, bool in_view, float4 sphere, WalkPrefetch pf) {
    const uint32_t dxy = v.dims[0] * v.dims[1], dz = v.dims[2];
    const uint32_t zc = CHUNKED ? zc_arg : dz;
    const uint32_t RC = dxy * zc;  // rows of one chunk
    uint32_t* rows = arena;
    uint32_t* type_rows = rows + RC * 8u;
    double* logf_tab = reinterpret_cast<double*>(type_rows + 48u);  // 32 doubles: the table of libm_logf
    float* planes = reinterpret_cast<float*>(logf_tab + 32);
    const uint32_t nx = v.dims[0] + 1u, ny = v.dims[1] + 1u, nz = dz + 1u;
    uint32_t* touched_bits = reinterpret_cast<uint32_t*>(planes + (PLANES_IN_LDS ? 4u * (nx + ny + nz) : 0u));
    uint32_t* n_touched = touched_bits + ((RC + 31u) >> 5);
    uint32_t* z_range = n_touched + 1;  // [0] min, [1] max z slice any object of the block may touch, [2] the block's first pair slot,
                                        // [3] the block's farthest_z (bits of a positive float, 0 = none)
    uint16_t* touched_list = reinterpret_cast<uint16_t*>(z_range + 4);
    const float* xp = PLANES_IN_LDS ? planes : v.x_planes;
    const float* yp = PLANES_IN_LDS ? planes + 4u * nx : v.y_planes;
    const float* zp = PLANES_IN_LDS ? planes + 4u * (nx + ny) : v.z_planes;

    // Most blocks of a big light set see nothing of it in this view: leave before touching LDS.
    // (the barriers below order LDS only: a __syncthreads() also waits for every global store and atomic in flight -- the far_z
    // atomic, the pair stores of the previous chunk --, a round trip of microseconds each while the frame's rows load HBM)
    const uint32_t n_plane_floats = PLANES_IN_LDS ? 4u * (nx + ny + nz) : 0u;
    const float pl0 = pf.pl0, pl1 = pf.pl1, pl2 = pf.pl2, pl3 = pf.pl3;
    const double logf_reg = pf.logf_reg;
    if (!__syncthreads_or(in_view ? 1 : 0)) return;
    MI_WALK_MARK(bx, 1);

    auto clear_chunk = [&]() {
        uint4* z4 = reinterpret_cast<uint4*>(rows);
        for (uint32_t i = threadIdx.x; i < RC * 2u; i += CLUSTER_BLOCK) z4[i] = make_uint4(0u, 0u, 0u, 0u);
        for (uint32_t i = threadIdx.x; i <= ((RC + 31u) >> 5); i += CLUSTER_BLOCK) touched_bits[i] = 0u;  // bits + counter
    };
    if (threadIdx.x < 48u) type_rows[threadIdx.x] = 0u;
    if (threadIdx.x == 0) { z_range[0] = 0xFFFFFFFFu; z_range[1] = 0u; z_range[3] = 0u; }
    if (!CHUNKED) clear_chunk();
    if (threadIdx.x < 32u) logf_tab[threadIdx.x] = logf_reg;
    if (PLANES_IN_LDS) {
        if (threadIdx.x < n_plane_floats) planes[threadIdx.x] = pl0;
        if (threadIdx.x + CLUSTER_BLOCK < n_plane_floats) planes[threadIdx.x + CLUSTER_BLOCK] = pl1;
        if (threadIdx.x + 2u * CLUSTER_BLOCK < n_plane_floats) planes[threadIdx.x + 2u * CLUSTER_BLOCK] = pl2;
        if (threadIdx.x + 3u * CLUSTER_BLOCK < n_plane_floats) planes[threadIdx.x + 3u * CLUSTER_BLOCK] = pl3;
        for (uint32_t i = threadIdx.x + 4u * CLUSTER_BLOCK; i < n_plane_floats; i += CLUSTER_BLOCK) planes[i] = v.x_planes[i];  // (grids beyond 255 planes)
    }
    MI_WG_LDS_BARRIER();
    MI_WALK_MARK(bx, 8);

    const uint32_t word = threadIdx.x >> 5, bit = 1u << (threadIdx.x & 31u);
    ObjectWalk ow = {};
    uint32_t my_lo = 0xFFFFFFFFu, my_hi = 0u, my_far = 0u;
    if (in_view) {
        float far_z = 0.0f;
        uint32_t type = 0u;
        if (CHUNKED) {  // the whole setup is computed inside the chunk loop: here only what the loop's bounds need
            object_z_extent(v, sphere, &my_lo, &my_hi, &far_z, logf_tab);
            type = o.obj_type ? o.obj_type[obj] : 0u;
        } else {
            ow = object_setup<SPOTS>(v, o, obj, sphere, &far_z, logf_tab);
            my_lo = ow.minc[2];
            my_hi = ow.maxc[2];
            type = ow.type;
        }
        atomicOr(&type_rows[(type < 6u ? type : 5u) * 8u + word], bit);
        // farthest_z = farthest_z.max(this_object_far_z), starting from 0.0 (assign.rs:421,561): only positive values can raise it,
        // and positive floats order like their bit patterns.  Reduced over the block first (wave shuffle, one LDS atomic per wave)
        // and sent as ONE global atomic when the block leaves: an atomic per visible light was 8 000 of them on one address, issued
        // into a memory pipeline the frame's rows keep full.
        if (far_z > 0.0f) my_far = __float_as_uint(far_z);
    }
    uint32_t bz0 = 0, bz1 = dz - 1u;
    {
        uint32_t fz = my_far;
#pragma unroll
        for (uint32_t off = 32u; off; off >>= 1) {
            const uint32_t f2 = __shfl_xor(fz, off, 64);
            fz = f2 > fz ? f2 : fz;
        }
        if ((threadIdx.x & 63u) == 0 && fz) atomicMax(&z_range[3], fz);
    }
    if (CHUNKED) {  // the z slices the block touches: wave reduction, then one LDS atomic per wave
        uint32_t lo = my_lo, hi = my_hi;
#pragma unroll
        for (uint32_t off = 32u; off; off >>= 1) {
            const uint32_t lo2 = __shfl_xor(lo, off, 64), hi2 = __shfl_xor(hi, off, 64);
            lo = lo2 < lo ? lo2 : lo;
            hi = hi2 > hi ? hi2 : hi;
        }
        if ((threadIdx.x & 63u) == 0) { atomicMin(&z_range[0], lo); atomicMax(&z_range[1], hi); }
        MI_WG_LDS_BARRIER();
        MI_WALK_MARK(bx, 9);
        bz0 = z_range[0];
        bz1 = z_range[1];
    }
    for (uint32_t z0 = bz0; z0 <= bz1 && z0 < dz; z0 += zc) {  // chunks start at the block's own first slice
        if (CHUNKED) {
            clear_chunk();
            MI_WG_LDS_BARRIER();
            MI_WALK_MARK(bx, 10);
        }
        if (in_view && my_lo <= z0 + zc - 1u && my_hi >= z0) {
            if (CHUNKED) {  // recomputed, not carried across the loop (see above); the asm keeps the compiler from hoisting it back out
                float unused;
                float4 sphere_again = sphere;  // (the sphere itself stays in registers: fetching it again would be a trip per chunk)
                asm volatile("" : "+v"(sphere_again.x), "+v"(sphere_again.y), "+v"(sphere_again.z), "+v"(sphere_again.w));
                ow = object_setup<SPOTS>(v, o, obj, sphere_again, &unused, logf_tab);
            }
            object_walk<SPOTS>(v, ow, z0, z0 + zc - 1u, xp, yp, zp, [&](uint32_t xy, uint32_t z) {
                const uint32_t r = xy * zc + (z - z0);
                atomicOr(&rows[r * 8u + word], bit);
                const uint32_t tb = 1u << (r & 31u);
                if (!(touched_bits[r >> 5] & tb) && !(atomicOr(&touched_bits[r >> 5], tb) & tb))
                    touched_list[atomicAdd(n_touched, 1u)] = (uint16_t)r;  // first toucher records the row
            });
        }
        MI_WG_LDS_BARRIER();
        MI_WALK_MARK(bx, 2);

        // Epilogue over the rows this workgroup touched in the chunk (a list kept next to the bit rows, so nothing is swept):
        // every touched row becomes a (cluster, block, 256-bit mask) pair in ONE global list -- the group reserves its
        // slots with a single atomic -- so the fill can spread pairs evenly over the chip no matter how unevenly the objects
        // are distributed.  Only non-empty entries of the (cluster, block) count matrix are written.
        const uint32_t nt = *n_touched;
        // the kinds of object this block holds among its objects in view: one (the rule) or several
        uint32_t single_type = 6u;
        {
            uint32_t kinds = 0u;
#pragma unroll
            for (uint32_t t = 0; t < 6; ++t) {
                uint32_t any_t = 0u;
#pragma unroll
                for (uint32_t k = 0; k < 8; ++k) any_t |= type_rows[t * 8u + k];
                kinds |= (any_t ? 1u : 0u) << t;
            }
            if (__popc(kinds) == 1u) single_type = (uint32_t)__ffs((int)kinds) - 1u;
        }
        if (threadIdx.x == 0) {
            if (z_range[3]) {  // (all waves' LDS atomics lie behind the barrier above; sent once, with the first chunk's reservation)
                atomicMax(reinterpret_cast<unsigned int*>(w.farthest_z), z_range[3]);
                z_range[3] = 0u;
            }
            z_range[2] = nt ? atomicAdd(w.pair_total, nt) : 0u;
        }
        MI_WG_LDS_BARRIER();
        MI_WALK_MARK(bx, 3);
        const uint32_t pair_base = z_range[2];
        for (uint32_t i = threadIdx.x; i < nt; i += CLUSTER_BLOCK) {
            const uint32_t r = touched_list[i];
            const uint32_t c = CHUNKED ? (r / zc) * dz + z0 + (r % zc) : r;
            // the row's 256-bit mask goes out half by half: this loop is the register peak of every kernel that carries the walk
            const uint32_t slot = pair_base + i;
            uint32_t cnt = 0;
            {
                const uint4 lo = reinterpret_cast<const uint4*>(rows)[r * 2u];
                cnt += __popc(lo.x) + __popc(lo.y) + __popc(lo.z) + __popc(lo.w);
                reinterpret_cast<uint4*>(w.pair_mask)[(size_t)slot * 2u] = lo;
            }
            {
                const uint4 hi = reinterpret_cast<const uint4*>(rows)[r * 2u + 1u];
                cnt += __popc(hi.x) + __popc(hi.y) + __popc(hi.z) + __popc(hi.w);
                reinterpret_cast<uint4*>(w.pair_mask)[(size_t)slot * 2u + 1u] = hi;
            }
            w.pair_cb[slot] = (bx << 12) | c;
            w.block_counts[(size_t)c * w.row_stride + bx] = (uint16_t)cnt;  // cluster-major
            atomicAdd(&w.totals[c], cnt);
            // The per-type counts (ClusterableObjectCounts, assign.rs:741-800).  Until round 5 the six 256-bit type masks were and-ed
            // with the row here; the compiler hoisted their 48 words out of this loop into registers, which made the walk 88 VGPRs
            // and -- with its 31 KB arena -- held every launch that carries it to 5 waves per SIMD: the ROWS of the metric frame paid
            // for that (the same rows through the lean kernel 16.9 us, through the walk-carrying one with nothing to walk 19.4;
            // profiles/r06_experiments.md).  Almost every block holds objects of ONE kind (lights are spawned together): then the
            // row's population is that kind's count and the masks are never read; mixed blocks read row and masks from LDS word by
            // word.  72 VGPRs and a 22 KB arena: 7 waves per SIMD.
#ifdef MI_EXP_TYPE_ROWS_HOISTED  // (A/B build: as until round 5)
            {
                const uint4 lo = reinterpret_cast<const uint4*>(rows)[r * 2u], hi = reinterpret_cast<const uint4*>(rows)[r * 2u + 1u];
                const uint32_t m[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
                for (uint32_t t = 0; t < 6; ++t) {
                    uint32_t tc = 0;
#pragma unroll
                    for (uint32_t k = 0; k < 8; ++k) tc += __popc(m[k] & type_rows[t * 8u + k]);
                    if (tc) atomicAdd(&w.counts[6u * c + t], tc);
                }
            }
#else
            if (single_type < 6u) {  // (workgroup-uniform)
                atomicAdd(&w.counts[6u * c + single_type], cnt);
            } else {
                for (uint32_t t = 0; t < 6; ++t) {
                    uint32_t tc = 0;
#pragma nounroll
                    for (uint32_t k = 0; k < 8; ++k) tc += __popc(rows[r * 8u + k] & type_rows[t * 8u + k]);
                    if (tc) atomicAdd(&w.counts[6u * c + t], tc);
                }
            }
#endif
        }
        MI_WALK_MARK(bx, 4);
#ifdef MI_EXP_TIMELINE
        if (threadIdx.x == 0 && bx < 4096u) { mi_walk_marks[16u * bx + 5u] = nt; mi_walk_marks[16u * bx + 6u] += 1; }
#endif
        if (!CHUNKED) break;
        MI_WG_LDS_BARRIER();  // the next chunk zeroes the rows
    }
    MI_WALK_MARK(bx, 7);
}

}  // namespace mi
