// host_helpers.cpp -- pure host entry points of the C ABI: what the Rust shim would otherwise do
// with glam before calling into the device path (camera frusta, per-view cluster constants,
// hierarchy flattening).  No device code, no HIP calls.  Compiled with -ffp-contract=off so the
// arithmetic follows glam_math.h's operation order.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/bevy_mi355x.h"
#include "glam_math.h"
#include "kernels.h"  // TILE_MAX_LEVELS: mi_hierarchy_advice_for asks the planner's question
#include "strip_plan.h"
#include "../../include/bevy_mi355x_debug.h"

using namespace mi;

namespace {

float signum(float x) { return isnan(x) ? x : (signbit(x) ? -1.0f : 1.0f); }

// Mat4::inverse (glam: cofactor expansion, GLM-derived)
M4 inverse4(const M4& s) {
    const float m00 = s.c[0].x, m01 = s.c[0].y, m02 = s.c[0].z, m03 = s.c[0].w;
    const float m10 = s.c[1].x, m11 = s.c[1].y, m12 = s.c[1].z, m13 = s.c[1].w;
    const float m20 = s.c[2].x, m21 = s.c[2].y, m22 = s.c[2].z, m23 = s.c[2].w;
    const float m30 = s.c[3].x, m31 = s.c[3].y, m32 = s.c[3].z, m33 = s.c[3].w;
    const float coef00 = m22 * m33 - m32 * m23, coef02 = m12 * m33 - m32 * m13, coef03 = m12 * m23 - m22 * m13;
    const float coef04 = m21 * m33 - m31 * m23, coef06 = m11 * m33 - m31 * m13, coef07 = m11 * m23 - m21 * m13;
    const float coef08 = m21 * m32 - m31 * m22, coef10 = m11 * m32 - m31 * m12, coef11 = m11 * m22 - m21 * m12;
    const float coef12 = m20 * m33 - m30 * m23, coef14 = m10 * m33 - m30 * m13, coef15 = m10 * m23 - m20 * m13;
    const float coef16 = m20 * m32 - m30 * m22, coef18 = m10 * m32 - m30 * m12, coef19 = m10 * m22 - m20 * m12;
    const float coef20 = m20 * m31 - m30 * m21, coef22 = m10 * m31 - m30 * m11, coef23 = m10 * m21 - m20 * m11;
    const V4 fac0 = v4(coef00, coef00, coef02, coef03), fac1 = v4(coef04, coef04, coef06, coef07);
    const V4 fac2 = v4(coef08, coef08, coef10, coef11), fac3 = v4(coef12, coef12, coef14, coef15);
    const V4 fac4 = v4(coef16, coef16, coef18, coef19), fac5 = v4(coef20, coef20, coef22, coef23);
    const V4 vec0 = v4(m10, m00, m00, m00), vec1 = v4(m11, m01, m01, m01);
    const V4 vec2 = v4(m12, m02, m02, m02), vec3 = v4(m13, m03, m03, m03);
    const V4 inv0 = (mul4(vec1, fac0) - mul4(vec2, fac1)) + mul4(vec3, fac2);
    const V4 inv1 = (mul4(vec0, fac0) - mul4(vec2, fac3)) + mul4(vec3, fac4);
    const V4 inv2 = (mul4(vec0, fac1) - mul4(vec1, fac3)) + mul4(vec3, fac5);
    const V4 inv3 = (mul4(vec0, fac2) - mul4(vec1, fac4)) + mul4(vec2, fac5);
    const V4 sign_a = v4(1.0f, -1.0f, 1.0f, -1.0f), sign_b = v4(-1.0f, 1.0f, -1.0f, 1.0f);
    M4 inv;
    inv.c[0] = mul4(inv0, sign_a);
    inv.c[1] = mul4(inv1, sign_b);
    inv.c[2] = mul4(inv2, sign_a);
    inv.c[3] = mul4(inv3, sign_b);
    const V4 col0 = v4(inv.c[0].x, inv.c[1].x, inv.c[2].x, inv.c[3].x);
    const V4 dot0 = mul4(s.c[0], col0);
    const float dot1 = (dot0.x + dot0.y) + (dot0.z + dot0.w);
    const float rcp = 1.0f / dot1;
    for (int i = 0; i < 4; ++i) inv.c[i] = inv.c[i] * rcp;
    return inv;
}

// clip_to_view, crates/bevy_light/src/cluster/assign.rs:1064-1067
V4 clip_to_view(const M4& view_from_clip, V4 clip) {
    const V4 view = mul(view_from_clip, clip);
    return v4(view.x / view.w, view.y / view.w, view.z / view.w, view.w / view.w);
}
// z_slice_to_view_z, assign.rs:903-920
float z_slice_to_view_z(float near, float far, uint32_t z_slices, uint32_t z_slice, bool ortho) {
    if (ortho) return -near - (far - near) * (float)z_slice / (float)z_slices;
    if (z_slice == 0) return 0.0f;
    return -near * powf(far / near, (float)(z_slice - 1) / (float)(z_slices - 1));
}
V4 screen_to_view(float sw, float sh, const M4& view_from_clip, float sx, float sy, float ndc_z) {
    const float tx = sx / sw, ty = sy / sh;
    return clip_to_view(view_from_clip, v4(tx * 2.0f - 1.0f, (1.0f - ty) * 2.0f - 1.0f, ndc_z, 1.0f));
}
V3 line_intersection_to_z_plane(V3 p, float z) {
    const V3 origin = v3(0.0f, 0.0f, 0.0f);
    const V3 v = p - origin;
    const float zo = (0.0f * origin.x + 0.0f * origin.y) + 1.0f * origin.z;
    const float zv = (0.0f * v.x + 0.0f * v.y) + 1.0f * v.z;
    const float t = (z - zo) / zv;
    return origin + v * t;
}

}  // namespace

extern "C" {

int32_t mi_perspective_clip_from_view(float fov, float aspect_ratio, float near, float out[16]) {
    if (!out) return MI_ERR_INVALID_ARG;
    // glam perspective_infinite_reverse (RH, reversed depth 0..1): f = 1/tan(fov/2) from sin_cos
    const float s = sinf(0.5f * fov), c = cosf(0.5f * fov);
    const float h = c / s;
    const float w = h / aspect_ratio;
    memset(out, 0, 16 * sizeof(float));
    out[0] = w;
    out[5] = h;
    out[11] = -1.0f;
    out[14] = near;
    return MI_OK;
}

int32_t mi_compute_frustum(const float clip_from_view[16], const float camera_affine[12], float far, float out[24]) {
    if (!clip_from_view || !camera_affine || !out) return MI_ERR_INVALID_ARG;
    // CameraProjection::compute_frustum, crates/bevy_camera/src/projection.rs:72-80
    const M4 proj = load_m4(clip_from_view);
    const Affine cam = load_affine(camera_affine);
    const M4 clip_from_world = mul(proj, m4_from_affine(inverse(cam)));
    // ViewFrustum::from_clip_from_world_no_far, crates/bevy_math/src/primitives/view_frustum.rs:91-108
    const V4 row0 = row(clip_from_world, 0), row1 = row(clip_from_world, 1), row2 = row(clip_from_world, 2),
             row3 = row(clip_from_world, 3);
    V4 hs[6];
    hs[0] = half_space_new(row3 + row0);
    hs[1] = half_space_new(row3 - row0);
    hs[2] = half_space_new(row3 + row1);
    hs[3] = half_space_new(row3 - row1);
    hs[4] = half_space_new(row3 + row2);
    // from_clip_from_world_custom_far, view_frustum.rs:52-64
    V3 back = mul(cam.m, v3(0.0f, 0.0f, 1.0f));
    back = back * (1.0f / sqrtf(dot3(back, back)));
    const V3 far_center = cam.t - back * far;
    hs[5] = half_space_new(extend(back, -dot3(back, far_center)));
    for (int i = 0; i < 6; ++i) { out[4 * i] = hs[i].x; out[4 * i + 1] = hs[i].y; out[4 * i + 2] = hs[i].z; out[4 * i + 3] = hs[i].w; }
    return MI_OK;
}

int32_t mi_cluster_dimensions_fixed_z(uint32_t total, uint32_t z_slices, uint32_t sw, uint32_t sh, uint32_t out[3]) {
    if (!out || sw == 0 || sh == 0 || total == 0 || z_slices == 0) return MI_ERR_INVALID_ARG;
    // ClusterConfig::dimensions_for_screen_size, crates/bevy_light/src/cluster/mod.rs:311-347
    const float aspect_ratio = (float)sw / (float)sh;
    if (total < z_slices) z_slices = total;
    const float per_layer = (float)total / (float)z_slices;
    const float y = sqrtf(per_layer / aspect_ratio);
    uint32_t x = f32_as_u32(y * aspect_ratio);
    uint32_t yi = f32_as_u32(y);
    if (x == 0) { x = 1; yi = f32_as_u32(per_layer); }
    if (yi == 0) { x = f32_as_u32(per_layer); yi = 1; }
    out[0] = x; out[1] = yi; out[2] = z_slices;
    return MI_OK;
}

int32_t mi_cluster_config_default(mi_cluster_config* out) {
    if (!out) return MI_ERR_INVALID_ARG;
    // ClusterConfig::default() + ClusterZConfig::default(), crates/bevy_light/src/cluster/mod.rs:288-308
    memset(out, 0, sizeof *out);
    out->kind = MI_CLUSTER_CONFIG_FIXED_Z;
    out->total = 4096;
    out->z_slices = 24;
    out->first_slice_depth = 5.0f;
    out->far_z_mode = MI_CLUSTER_FAR_Z_MAX_CLUSTERABLE_OBJECT_RANGE;
    out->dynamic_resizing = 1;
    return MI_OK;
}

int32_t mi_cluster_config_resolve(const mi_cluster_config* config, const mi_cluster_history* last, uint32_t sw, uint32_t sh,
                                  uint64_t max_indices, mi_cluster_resolved* out) {
    if (!config || !out || config->kind > MI_CLUSTER_CONFIG_FIXED_Z) return MI_ERR_INVALID_ARG;
    memset(out, 0, sizeof *out);
    // assign.rs:328-339: ClusterConfig::None, or a viewport without pixels -> Clusters::clear()
    if (config->kind == MI_CLUSTER_CONFIG_NONE || sw == 0 || sh == 0) return MI_OK;
    // ClusterConfig::dimensions_for_screen_size, cluster/mod.rs:311-347
    uint32_t dims[3] = {1, 1, 1};
    if (config->kind == MI_CLUSTER_CONFIG_XYZ) {
        memcpy(dims, config->dimensions, sizeof dims);
    } else if (config->kind == MI_CLUSTER_CONFIG_FIXED_Z) {
        if (config->total == 0 || config->z_slices == 0) return MI_ERR_INVALID_ARG;
        int32_t rc = mi_cluster_dimensions_fixed_z(config->total, config->z_slices, sw, sh, dims);
        if (rc) return rc;
    }
    if (dims[0] == 0 || dims[1] == 0 || dims[2] == 0) return MI_ERR_INVALID_ARG;  // Clusters::update debug_assert, mod.rs:399-401
    const bool has_z_config = config->kind == MI_CLUSTER_CONFIG_XYZ || config->kind == MI_CLUSTER_CONFIG_FIXED_Z;
    // first_slice_depth() / far_z_mode() / dynamic_resizing(), mod.rs:349-382: Single = (0.0, MaxClusterableObjectRange, false)
    out->first_slice_depth = has_z_config ? config->first_slice_depth : 0.0f;
    const bool constant_far = has_z_config && config->far_z_mode == MI_CLUSTER_FAR_Z_CONSTANT;
    // assign.rs:350-355 (DEFAULT_FAR_DEPTH :37)
    out->far_z = constant_far ? config->far_z_constant : ((last && last->has_farthest_z) ? last->farthest_z : 1000.0f);
    // assign.rs:384-404
    if (has_z_config && config->dynamic_resizing && last && last->has_total_cluster_index_count &&
        last->total_cluster_index_count > max_indices) {
        const float index_ratio = (float)max_indices / (float)last->total_cluster_index_count;  // usize as f32
        const float xy_ratio = sqrtf(index_ratio);
        dims[0] = std::max(f32_as_u32(floorf((float)dims[0] * xy_ratio)), 1u);
        dims[1] = std::max(f32_as_u32(floorf((float)dims[1] * xy_ratio)), 1u);
    }
    memcpy(out->requested_dims, dims, sizeof dims);
    out->active = 1;
    return MI_OK;
}

int32_t mi_cluster_sort_truncate(uint32_t n, const uint8_t* obj_type, const uint8_t* shadow_maps_enabled, const uint8_t* volumetric,
                                 const uint64_t* entity_bits, uint32_t max_objects, uint32_t supports_storage_buffers,
                                 uint32_t* out_order, uint32_t* out_n) {
    if ((n && !out_order) || !out_n) return MI_ERR_INVALID_ARG;
    for (uint32_t i = 0; i < n; ++i) out_order[i] = i;
    *out_n = n;
    if (n <= max_objects || supports_storage_buffers) return MI_OK;  // assign.rs:297-300
    if (!entity_bits) return MI_ERR_INVALID_ARG;
    // sort_by_cached_key(|o| (o.object_type.ordering(), o.entity)), assign.rs:301-306; ordering() :108-128.
    // Tuple order: type, then the two negated bools (false < true), then Entity (= to_bits, entity/mod.rs:566-568).
    auto key = [&](uint32_t i, uint32_t k[3]) {
        const uint32_t t = obj_type ? obj_type[i] : (uint32_t)MI_OBJ_POINT_LIGHT;
        const bool light = t == MI_OBJ_POINT_LIGHT || t == MI_OBJ_SPOT_LIGHT;
        k[0] = t;
        k[1] = light ? !(shadow_maps_enabled && shadow_maps_enabled[i]) : 0u;
        k[2] = light ? !(volumetric && volumetric[i]) : 0u;
    };
    std::stable_sort(out_order, out_order + n, [&](uint32_t a, uint32_t b) {
        uint32_t ka[3], kb[3];
        key(a, ka);
        key(b, kb);
        for (int j = 0; j < 3; ++j)
            if (ka[j] != kb[j]) return ka[j] < kb[j];
        return entity_bits[a] < entity_bits[b];
    });
    *out_n = max_objects;  // truncate, :319-320
    return MI_OK;
}

int32_t mi_cluster_view_dims(uint32_t sw, uint32_t sh, const uint32_t req[3], uint32_t tile[2], uint32_t dims[3]) {
    if (!req || !tile || !dims || sw == 0 || sh == 0 || req[0] == 0 || req[1] == 0 || req[2] == 0) return MI_ERR_INVALID_ARG;
    // Clusters::update, cluster/mod.rs:398-416
    uint32_t tx = f32_as_u32(ceilf((float)sw / (float)req[0]));
    uint32_t ty = f32_as_u32(ceilf((float)sh / (float)req[1]));
    tile[0] = std::max(tx, 1u);
    tile[1] = std::max(ty, 1u);
    dims[0] = std::max(f32_as_u32(ceilf((float)sw / (float)tile[0])), 1u);
    dims[1] = std::max(f32_as_u32(ceilf((float)sh / (float)tile[1])), 1u);
    dims[2] = std::max(req[2], 1u);
    return MI_OK;
}

int32_t mi_cluster_view_build(const float camera_affine[12], const float clip_from_view[16], const float frustum[24],
                              uint32_t sw, uint32_t sh, const uint32_t requested_dims[3], float first_slice_depth_cfg,
                              float far_z, uint32_t view_layer_mask, float* plane_storage, float* sphere_storage,
                              mi_cluster_view* out) {
    if (!camera_affine || !clip_from_view || !frustum || !requested_dims || !plane_storage || !out) return MI_ERR_INVALID_ARG;
    int32_t rc = mi_cluster_view_dims(sw, sh, requested_dims, out->tile_size, out->dims);
    if (rc) return rc;
    // assign.rs:344-380
    const Affine world_from_view = load_affine(camera_affine);
    const float det = determinant(world_from_view.m);
    const V3 scale = v3(length3(world_from_view.m.x_axis) * signum(det), length3(world_from_view.m.y_axis),
                        length3(world_from_view.m.z_axis));
    const V3 vfw_scale = v3(1.0f / scale.x, 1.0f / scale.y, 1.0f / scale.z);
    const float scale_max = rust_max(rust_max(f_abs(vfw_scale.x), f_abs(vfw_scale.y)), f_abs(vfw_scale.z));
    const M4 view_from_world = m4_from_affine(inverse(world_from_view));
    const M4 cfv = load_m4(clip_from_view);
    const bool ortho = cfv.c[3].w == 1.0f;
    float first_slice_depth;
    if (ortho) first_slice_depth = (cfv.c[3].z - 1.0f) / cfv.c[2].z;
    else if (requested_dims[2] == 1) first_slice_depth = rust_max(first_slice_depth_cfg, far_z);
    else first_slice_depth = first_slice_depth_cfg;
    first_slice_depth = first_slice_depth * vfw_scale.z;
    far_z = rust_max(far_z, first_slice_depth);
    // calculate_cluster_factors, assign.rs:817-832
    const float z_slices = (float)requested_dims[2];
    if (ortho) {
        out->cluster_factors[0] = -first_slice_depth;
        out->cluster_factors[1] = z_slices / (-far_z - -first_slice_depth);
    } else {
        const float k = (z_slices - 1.0f) / logf(far_z / first_slice_depth);
        out->cluster_factors[0] = k;
        out->cluster_factors[1] = logf(first_slice_depth) * k;
    }
    out->screen_size[0] = sw; out->screen_size[1] = sh;
    out->is_orthographic = ortho ? 1u : 0u;
    out->view_layer_mask = view_layer_mask;
    out->view_layer_mask_hi = 0u;  // (layers 32..63: the caller's to set)
    out->near_ = first_slice_depth;
    out->far_ = far_z;
    store_m4(view_from_world, out->view_from_world);
    store_m4(cfv, out->clip_from_view);
    const M4 view_from_clip = inverse4(cfv);
    store_m4(view_from_clip, out->view_from_clip);
    out->view_from_world_scale[0] = vfw_scale.x; out->view_from_world_scale[1] = vfw_scale.y; out->view_from_world_scale[2] = vfw_scale.z;
    out->view_from_world_scale_max = scale_max;
    memcpy(out->frustum, frustum, 24 * sizeof(float));

    const uint32_t dx = out->dims[0], dy = out->dims[1], dz = out->dims[2];
    float* xp = plane_storage;
    float* yp = xp + 4 * (size_t)(dx + 1);
    float* zp = yp + 4 * (size_t)(dy + 1);
    auto put = [](float* dst, V4 hs) { dst[0] = hs.x; dst[1] = hs.y; dst[2] = hs.z; dst[3] = hs.w; };
    // assign.rs:434-476
    const float x_slices = (float)dx, y_slices = (float)dy;
    for (uint32_t x = 0; x <= dx; ++x) {
        const float x_pos = ((float)x / x_slices) * 2.0f - 1.0f;
        if (ortho) {
            const float view_x = clip_to_view(view_from_clip, v4(x_pos, 0.0f, 1.0f, 1.0f)).x;
            put(xp + 4 * x, half_space_new(v4(1.0f, 0.0f, 0.0f, view_x * 1.0f)));
        } else {
            const V3 nb = xyz(clip_to_view(view_from_clip, v4(x_pos, -1.0f, 1.0f, 1.0f)));
            const V3 nt = xyz(clip_to_view(view_from_clip, v4(x_pos, 1.0f, 1.0f, 1.0f)));
            const V3 normal = cross3(nb, nt);
            put(xp + 4 * x, half_space_new(extend(normal, dot3(nb, normal))));
        }
    }
    for (uint32_t y = 0; y <= dy; ++y) {
        const float y_pos = (1.0f - (float)y / y_slices) * 2.0f - 1.0f;
        if (ortho) {
            const float view_y = clip_to_view(view_from_clip, v4(0.0f, y_pos, 1.0f, 1.0f)).y;
            put(yp + 4 * y, half_space_new(v4(0.0f, 1.0f, 0.0f, view_y * 1.0f)));
        } else {
            const V3 nl = xyz(clip_to_view(view_from_clip, v4(-1.0f, y_pos, 1.0f, 1.0f)));
            const V3 nr = xyz(clip_to_view(view_from_clip, v4(1.0f, y_pos, 1.0f, 1.0f)));
            const V3 normal = cross3(nr, nl);
            put(yp + 4 * y, half_space_new(extend(normal, dot3(nr, normal))));
        }
    }
    // assign.rs:478-485
    for (uint32_t z = 0; z <= dz; ++z) {
        const float view_z = z_slice_to_view_z(first_slice_depth, far_z, dz, z, ortho);
        put(zp + 4 * z, half_space_new(v4(-0.0f, -0.0f, -1.0f, view_z * -1.0f)));
    }
    out->x_planes = xp;
    out->y_planes = yp;
    out->z_planes = zp;
    out->cluster_spheres = nullptr;
    if (sphere_storage) {
        // compute_aabb_for_cluster -> bounding sphere for every cluster (lazily built at assign.rs:693-707)
        const float tsx = (float)out->tile_size[0], tsy = (float)out->tile_size[1];
        const float fw = (float)sw, fh = (float)sh;
        for (uint32_t y = 0; y < dy; ++y)
            for (uint32_t x = 0; x < dx; ++x)
                for (uint32_t z = 0; z < dz; ++z) {
                    const float ix = (float)x, iy = (float)y, iz = (float)z;
                    const float pminx = ix * tsx, pminy = iy * tsy;
                    const float pmaxx = pminx + tsx, pmaxy = pminy + tsy;
                    V3 cmin, cmax;
                    if (ortho) {
                        V3 p_min = xyz(screen_to_view(fw, fh, view_from_clip, pminx, pminy, 0.0f));
                        V3 p_max = xyz(screen_to_view(fw, fh, view_from_clip, pmaxx, pmaxy, 0.0f));
                        p_min.z = -first_slice_depth + (first_slice_depth - far_z) * iz / (float)dz;
                        p_max.z = -first_slice_depth + (first_slice_depth - far_z) * (iz + 1.0f) / (float)dz;
                        cmin = min3(p_min, p_max);
                        cmax = max3(p_min, p_max);
                    } else {
                        const V3 p_min = xyz(screen_to_view(fw, fh, view_from_clip, pminx, pminy, 1.0f));
                        const V3 p_max = xyz(screen_to_view(fw, fh, view_from_clip, pmaxx, pmaxy, 1.0f));
                        const float ratio = -far_z / -first_slice_depth;
                        const float cluster_near = (iz == 0.0f) ? 0.0f : -first_slice_depth * powf(ratio, (iz - 1.0f) / (float)(dz - 1));
                        const float cluster_far = (dz == 1) ? -far_z : -first_slice_depth * powf(ratio, iz / (float)(dz - 1));
                        const V3 a = line_intersection_to_z_plane(p_min, cluster_near);
                        const V3 b = line_intersection_to_z_plane(p_min, cluster_far);
                        const V3 c = line_intersection_to_z_plane(p_max, cluster_near);
                        const V3 d = line_intersection_to_z_plane(p_max, cluster_far);
                        cmin = min3(min3(a, b), min3(c, d));
                        cmax = max3(max3(a, b), max3(c, d));
                    }
                    const V3 center = (cmax + cmin) * 0.5f;
                    const V3 half = (cmax - cmin) * 0.5f;
                    float* dst = sphere_storage + 4 * (size_t)((y * dx + x) * dz + z);
                    dst[0] = center.x; dst[1] = center.y; dst[2] = center.z; dst[3] = length3(half);
                }
        out->cluster_spheres = sphere_storage;
    }
    return MI_OK;
}

// Level (BFS) order of an arbitrary ChildOf array -- replaces the Children Vec<Entity> pointer chase
// (crates/bevy_ecs/src/hierarchy.rs:107,152) with contiguous per-level ranges.
// test hook (bevy_mi355x_debug.h): the strips plan of a hierarchy as mi_upload_hierarchy would make it at `width` rows to a level -- host
// code only, no device, no context (tests/test_strip_plan.py walks the plan on the CPU).  out_counts = {strips, table entries, bands,
// snapshot rows}; all zero when the hierarchy cannot be planned.  out_strips: 3 words per strip (first entry, the StripDesc word, first
// own level); out_rounds: 4 words per entry (StripRound).
int32_t mi_debug_plan_strips(uint32_t n_levels, const uint32_t* level_offsets, const uint32_t* parent_idx, uint32_t width, uint32_t* out_strips,
                             uint32_t cap_strips, uint32_t* out_rounds, uint32_t cap_rounds, uint32_t* out_counts) {
    if (!out_counts || !n_levels || !level_offsets || !parent_idx) return MI_ERR_INVALID_ARG;
    const uint32_t n = level_offsets[n_levels];
    out_counts[0] = out_counts[1] = out_counts[2] = out_counts[3] = 0;
    for (uint32_t l = 0; l < n_levels; ++l)
        if (level_offsets[l + 1] < level_offsets[l]) return MI_ERR_MALFORMED_HIERARCHY;
    // rows of a level ordered by parent, parents in the level above (what mi_upload_hierarchy checks)
    for (uint32_t l = 1; l < n_levels; ++l) {
        uint32_t prev = level_offsets[l - 1];
        for (uint32_t i = level_offsets[l]; i < level_offsets[l + 1]; ++i) {
            const uint32_t p = parent_idx[i];
            if (p < level_offsets[l - 1] || p >= level_offsets[l] || p < prev) return MI_ERR_MALFORMED_HIERARCHY;
            prev = p;
        }
    }
    std::vector<uint32_t> first_child((size_t)n + 1, 0);
    for (uint32_t l = 0; l + 1 < n_levels; ++l) {
        uint32_t ch = level_offsets[l + 1];
        const uint32_t chi = level_offsets[l + 2];
        for (uint32_t p = level_offsets[l]; p < level_offsets[l + 1]; ++p) {
            first_child[p] = ch;
            while (ch < chi && parent_idx[ch] == p) ++ch;
        }
    }
    for (uint32_t p = level_offsets[n_levels - 1]; p <= n; ++p) first_child[p] = n;
    StripPlan sp;
    if (!plan_strips(n, n_levels, level_offsets, parent_idx, first_child.data(), width, true, 1000u, sp)) return MI_OK;
    out_counts[0] = (uint32_t)sp.strips.size();
    out_counts[1] = (uint32_t)sp.rounds.size();
    out_counts[2] = sp.n_bands;
    out_counts[3] = sp.snap_rows;
    for (size_t i = 0; out_strips && i < sp.strips.size() && i < cap_strips; ++i) {
        out_strips[3 * i] = sp.strips[i].first_round;
        out_strips[3 * i + 1] = sp.strips[i].n_rounds;
        out_strips[3 * i + 2] = sp.strip_top[i];
    }
    for (size_t i = 0; out_rounds && i < sp.rounds.size() && i < cap_rounds; ++i) {
        out_rounds[4 * i] = sp.rounds[i].row0;
        out_rounds[4 * i + 1] = sp.rounds[i].pstart;
        out_rounds[4 * i + 2] = sp.rounds[i].info;
        out_rounds[4 * i + 3] = sp.rounds[i].level;
    }
    return MI_OK;
}

// (the rule mi_upload_hierarchy applies -- ctx_hierarchy.cpp: ctx->narrow -- as a question the host can ask before it uploads anything)
int32_t mi_hierarchy_advice_for(uint32_t n_levels, const uint32_t* level_offsets, mi_hierarchy_advice* out) {
    if (!out || (n_levels && !level_offsets)) return MI_ERR_INVALID_ARG;
    mi_hierarchy_advice a{};
    a.n_levels = n_levels;
    bool every_level_fits_a_wave = n_levels > 0;
    for (uint32_t l = 0; l < n_levels; ++l) {
        if (level_offsets[l + 1] < level_offsets[l]) return MI_ERR_MALFORMED_HIERARCHY;
        const uint32_t w = level_offsets[l + 1] - level_offsets[l];
        a.widest_level = w > a.widest_level ? w : a.widest_level;
        every_level_fits_a_wave = every_level_fits_a_wave && w >= 1u && w <= 64u;
    }
    const uint32_t n = n_levels ? level_offsets[n_levels] - level_offsets[0] : 0u;
    a.est_host_us = 0.02f * (float)n;
    a.plan = n_levels <= 1u ? MI_HIERARCHY_PLAN_FLAT : (every_level_fits_a_wave && n_levels > mi::TILE_MAX_LEVELS) ? MI_HIERARCHY_PLAN_ONE_WAVE : MI_HIERARCHY_PLAN_TILES;
    if (a.plan == MI_HIERARCHY_PLAN_ONE_WAVE) {
        // a launch and its round trips + the dependent level steps: 0.32 us a level with a node per quad of lanes (no level above 16
        // rows: chain 774 us / 2 500 levels), 0.85 us with a lane per row (ropes: 260 us / 300 levels) -- profiles/r06y/shapes_table.md
        a.est_device_us = 8.0f + (a.widest_level <= 16u ? 0.32f : 0.85f) * (float)n_levels;
        a.keep_on_host = a.est_host_us < a.est_device_us ? 1u : 0u;
    }
    *out = a;
    return MI_OK;
}

int32_t mi_hierarchy_sort(uint32_t n, const uint32_t* parent, uint32_t* new_to_old, uint32_t* out_parent_idx,
                          uint32_t* out_level_offsets, uint32_t level_capacity, uint32_t* out_n_levels) {
    if (!parent || !new_to_old || !out_parent_idx || !out_level_offsets || !out_n_levels || level_capacity < 2) return MI_ERR_INVALID_ARG;
    // children CSR in caller order (stable)
    std::vector<uint32_t> start((size_t)n + 1, 0), cursor;
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t p = parent[i];
        if (p == MI_NO_PARENT) continue;
        if (p >= n || p == i) return MI_ERR_MALFORMED_HIERARCHY;
        start[p + 1]++;
    }
    for (uint32_t i = 0; i < n; ++i) start[i + 1] += start[i];
    cursor.assign(start.begin(), start.end() - 1);
    std::vector<uint32_t> kids(n ? n : 1);
    for (uint32_t i = 0; i < n; ++i)
        if (parent[i] != MI_NO_PARENT) kids[cursor[parent[i]]++] = i;
    std::vector<uint32_t> old_to_new(n ? n : 1, MI_NO_PARENT);
    uint32_t filled = 0, levels = 0;
    out_level_offsets[0] = 0;
    for (uint32_t i = 0; i < n; ++i)
        if (parent[i] == MI_NO_PARENT) { old_to_new[i] = filled; new_to_old[filled++] = i; }
    uint32_t lo = 0;
    while (true) {
        ++levels;
        if (levels >= level_capacity) return MI_ERR_CAPACITY;
        out_level_offsets[levels] = filled;
        if (filled == lo) { --levels; break; }  // empty level: done
        const uint32_t hi = filled;
        for (uint32_t q = lo; q < hi; ++q) {
            const uint32_t o = new_to_old[q];
            for (uint32_t k = start[o]; k < start[o + 1]; ++k) {
                const uint32_t ch = kids[k];
                old_to_new[ch] = filled;
                new_to_old[filled++] = ch;
            }
        }
        lo = hi;
    }
    if (filled != n) return MI_ERR_MALFORMED_HIERARCHY;  // unreachable rows: a cycle
    for (uint32_t q = 0; q < n; ++q) {
        const uint32_t p = parent[new_to_old[q]];
        out_parent_idx[q] = p == MI_NO_PARENT ? MI_NO_PARENT : old_to_new[p];
    }
    out_level_offsets[levels] = n;
    *out_n_levels = n ? levels : 0;
    if (n == 0) { *out_n_levels = 1; out_level_offsets[1] = 0; }
    return MI_OK;
}

}  // extern "C"
