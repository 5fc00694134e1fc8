// visibility_rule.h -- the per-entity visibility closure of the main-world visibility systems, shared by the frame kernel
// (kernels_flat.hip) and by the light-cluster walk when it re-derives a light row's ViewVisibility itself instead of waiting
// for the frame kernel of the same frame (kernels_cluster.hip): same function, same bits.
#pragma once
#include "glam_math.h"
#include "kernels.h"

namespace mi {

// One row against one view.  The view's flags select which of the reference's per-entity closures applies:
//   camera view        check_visibility_cpu_culling, crates/bevy_camera/src/visibility/mod.rs:788-858
//   VIEW_SHADOW        check_dir_light_mesh_visibility (cascades, crates/bevy_light/src/lib.rs:425-475: OBB only,
//                      near plane skipped, far plane tested) and check_point_light_mesh_visibility (cube faces
//                      :592-650, spot :694-738: light-sphere pre-test, then all six planes)
// Camera views: the sphere pre-test and the OBB test share the world-space centre and the plane dot products
// (bit-identical values in the reference: both call transform_point3a on the same inputs, mod.rs:827 and
// primitives.rs:279), so they are computed once; the far plane is never tested (mod.rs:831,835).
// Visibility ranges (range.rs:159-161,255-263) are evaluated on the fly from the row's (start, end) pair and the
// view's position instead of a per-(view, entity) table.
// Frustum::intersects_sphere over planes 0..4 (the far plane is never tested on this path, visibility/mod.rs:829-832): true = no
// plane has the sphere wholly behind it.  (Two planes per packed-FP32 instruction -- v_pk_mul_f32 / v_pk_add_f32 on float2 values,
// bit-identical -- was measured again in round 3 on the kernel that is bound by instruction issue, k_frame_sph at 10 M rows x 4
// views: 150.0 against 151.8 us, and 220.6 against 214.4 on k_frame; not kept.)
__device__ __forceinline__ bool sphere_inside_five_planes(const float* planes, V4 c4, float r) {
    bool inside = true;
#ifdef MI_EXP_PLANES_UPFRONT  // all five planes in one scalar load, no early exit between planes: k_frame 19.5 -> 19.25 us, 10 M x 1 view 151.7 -> 149.8,
                              // 10 M x 4 views 177.9 -> 179.1; k_frame_cells<true> drops to 7 waves per SIMD -- not the product (profiles/r05a/planes_upfront_ab.txt)
    float P[20];
#pragma unroll
    for (int i = 0; i < 20; ++i) P[i] = planes[i];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const V4 pl = V4{P[4 * i], P[4 * i + 1], P[4 * i + 2], P[4 * i + 3]};
        inside = inside & !(dot4(pl, c4) + r <= 0.0f);
    }
    return inside;
#endif
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const V4 pl = V4{planes[4 * i], planes[4 * i + 1], planes[4 * i + 2], planes[4 * i + 3]};
        inside = inside && !(dot4(pl, c4) + r <= 0.0f);
    }
    return inside;
}

__device__ __forceinline__ bool row_visible_in_view(const Affine& g, V3 center, V3 half, uint32_t fl,
                                                    uint32_t entity_mask, uint32_t entity_mask_hi, bool have_ranges, float range_lo,
                                                    float range_hi, const ViewParams& vp) {
    const bool shadow = (vp.flags & VIEW_SHADOW) != 0;
    bool vis = (fl & 0x01u) != 0;                          // InheritedVisibility
    vis = vis && (!shadow || (fl & 0x80u));                // shadow views only see shadow casters
    vis = vis && ((vp.layer_mask & entity_mask) | (vp.layer_mask_hi & entity_mask_hi)) != 0;  // RenderLayers::intersects (render_layers.rs:121-135), layers 0..63
    const bool has_aabb = (fl & 0x04u) != 0;
    if ((fl & 0x20u) && have_ranges) {                     // Has<VisibilityRange> && VisibleEntityRanges exists
        bool in_range = false;
        if ((vp.flags & (VIEW_RANGES | VIEW_RANGES_NO_ORIGIN)) == VIEW_RANGES) {
            const V3 model = ((fl & 0x40u) && has_aabb) ? transform_point(g, center) : g.t;
            const float d = length3(V3{vp.position[0], vp.position[1], vp.position[2]} - model);
            in_range = d >= range_lo && d < range_hi;
        }
        vis = vis && in_range;
    }
    if (shadow) {
        if (has_aabb && !(fl & 0x02u)) {
            const V3 cw = transform_point(g, center);
            bool inside = true;
            if (vp.flags & VIEW_LIGHT_SPHERE)
                inside = sphere_intersects_obb(V3{vp.light_sphere[0], vp.light_sphere[1], vp.light_sphere[2]},
                                               vp.light_sphere[3], cw, half, g.m);
            const V4 c4 = extend(cw, 1.0f);
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                if ((i == 4 && (vp.flags & VIEW_SKIP_NEAR)) || (i == 5 && !(vp.flags & VIEW_TEST_FAR))) continue;
                const V4 pl = V4{vp.planes[4 * i], vp.planes[4 * i + 1], vp.planes[4 * i + 2], vp.planes[4 * i + 3]};
                const float rr = aabb_relative_radius(half, xyz(pl), g.m);
                inside = inside && !(dot4(pl, c4) + rr <= 0.0f);
            }
            vis = vis && inside;
        }
        return vis;
    }
    const bool cull = !(fl & 0x02u) && !(vp.flags & VIEW_NO_CPU_CULLING);  // !NoFrustumCulling && !camera NoCpuCulling
    if (cull && (fl & (0x04u | 0x08u))) {
        // world-space sphere: Aabb -> (affine*center, |M3*half|) ; Sphere component used as is -- or, a light's sphere that follows
        // its entity (MI_SPHERE_AT_TRANSLATION), centred at the row's GlobalTransform translation
        const bool at_translation = !has_aabb && __float_as_uint(half.y) == SPHERE_AT_TRANSLATION;
        const V3 cw = has_aabb ? transform_point(g, center) : V3{at_translation ? g.t.x : center.x, at_translation ? g.t.y : center.y, at_translation ? g.t.z : center.z};
        const float sr = has_aabb ? length3(mul(g.m, half)) : half.x;
        const V4 c4 = extend(cw, 1.0f);
        // intersects_sphere over the five planes first (mod.rs:829-832) ...
        bool inside = sphere_inside_five_planes(vp.planes, c4, sr);
        // ... then intersects_obb (:833-836) -- 20 flops per plane and row -- only in waves where some row is still a candidate:
        // rows are usually numbered with some spatial coherence and a view sees a few percent of them, so most waves skip it.
        // (The reference returns early per entity; a conjunction of the same tests gives the same answer in any order.)
        if (__builtin_amdgcn_ballot_w64(vis && inside && has_aabb) != 0ull) {
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const V4 pl = V4{vp.planes[4 * i], vp.planes[4 * i + 1], vp.planes[4 * i + 2], vp.planes[4 * i + 3]};
                if (has_aabb) inside = inside && !(dot4(pl, c4) + aabb_relative_radius(half, xyz(pl), g.m) <= 0.0f);
            }
        }
        vis = vis && inside;
    }
    return vis;
}


}  // namespace mi
