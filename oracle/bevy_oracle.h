/*
 * bevy_oracle.h -- CPU ORACLE for the propagate -> cull -> cluster render-prep path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it, and only as the checker / reported
 * baseline.  The product (bevy_amd/, include/) never links, imports or calls it.
 *
 * It is a plain-C restatement of the reference's CPU systems (bevyengine/bevy @ v0.20.0-dev,
 * paths relative to /root/reference) and of the glam 0.33.2 SSE2 arithmetic they call
 * (glam is a crates.io dependency, crates/bevy_math/Cargo.toml:13, NOT present under
 * /root/reference; its published algorithm is restated, see the per-function notes).
 *
 * PARITY PINNING STATUS
 *   - Frustum/sphere/OBB/half-space tests: pinned against every known-answer vector in
 *     crates/bevy_camera/src/primitives.rs:462-857 and benches/benches/bevy_camera/primitives.rs:41-52
 *     (tests/test_oracle_golden.py).
 *   - Transform propagation: pinned against the exact-equality tests in
 *     crates/bevy_transform/src/systems.rs:827-1221 and helper.rs:97-146.
 *   - ViewVisibility bit protocol: pinned against visibility/mod.rs:1313-1448.
 *   - Cluster tiling: pinned against crates/bevy_light/src/cluster/test.rs.
 *   - assign_objects_to_clusters: PARITY UNPINNED by the reference (it has no test that
 *     exercises it); the restatement below is the spec, cross-checked by invariants.
 *   - Last-ulp order of glam's SSE2 lanes: PARITY UNPINNED here (no Rust toolchain, glam
 *     source absent); the operation order is restated from glam's published source.
 *
 * Data conventions (shared with include/bevy_mi355x.h):
 *   translation f32[3n], rotation f32[4n] (x,y,z,w), scale f32[3n]
 *   global transform f32[12n] = Affine3A::to_cols_array(): x_axis, y_axis, z_axis, translation
 *   aabb center f32[3n], half extents f32[3n]; sphere = (center, half.x as radius)
 *   frusta f32[V*24]: six half-spaces (nx,ny,nz,d), order left,right,top,bottom,near,far
 */
#ifndef BEVY_ORACLE_H
#define BEVY_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* entity flag bits (same values as MI_FLAG_* in include/bevy_mi355x.h) */
#define ORC_FLAG_INHERITED_VISIBLE   0x01u
#define ORC_FLAG_NO_FRUSTUM_CULLING  0x02u
#define ORC_FLAG_HAS_AABB            0x04u
#define ORC_FLAG_HAS_SPHERE          0x08u
#define ORC_FLAG_NO_CPU_CULLING      0x10u
#define ORC_FLAG_HAS_VISIBILITY_RANGE 0x20u
#define ORC_FLAG_RANGE_USE_AABB      0x40u /* VisibilityRange::use_aabb */
#define ORC_FLAG_SHADOW_CASTER       0x80u /* With<Mesh3d>, Without<NotShadowCaster>, Without<DirectionalLight> */

/* per-view flags (same values as MI_VIEW_FLAG_*) */
#define ORC_VIEW_FLAG_NO_CPU_CULLING 0x01u /* camera Has<NoCpuCulling> */
#define ORC_VIEW_FLAG_SHADOW         0x02u /* a shadow view (cascade / cube face / spot), bevy_light/src/lib.rs:342-757 */
#define ORC_VIEW_FLAG_SKIP_NEAR      0x04u /* intersects_obb(.., intersect_near = false, ..) */
#define ORC_VIEW_FLAG_TEST_FAR       0x08u /* intersects_obb(.., .., intersect_far = true) */
#define ORC_VIEW_FLAG_LIGHT_SPHERE   0x10u /* point/spot: light_sphere.intersects_obb pre-test */
#define ORC_VIEW_FLAG_RANGES         0x20u /* VisibleEntityRanges exists and this view has an index in it */
#define ORC_VIEW_FLAG_RANGES_NO_ORIGIN 0x40u /* point/spot with no shadow LOD origin: ranged entities are culled */

#define ORC_NO_PARENT 0xFFFFFFFFu

/* ---- glam / bevy_math primitives ------------------------------------------------------- */

/* Transform::compute_affine, crates/bevy_transform/src/components/transform.rs:273-275 */
void orc_transform_to_affine(const float t[3], const float r[4], const float s[3], float out[12]);
/* Affine3A * Affine3A (GlobalTransform::mul_transform is a * affine(T)), global_transform.rs:315-317 */
void orc_affine_mul(const float a[12], const float b[12], float out[12]);
/* Affine3A::inverse */
void orc_affine_inverse(const float a[12], float out[12]);
/* Affine3A::transform_point3a */
void orc_affine_transform_point(const float a[12], const float p[3], float out[3]);
/* GlobalTransform::radius_vec3a, global_transform.rs:252-254 */
void orc_probe_lane_orders(const float a[4], const float b[4], float out[3]);
float orc_radius_vec3a(const float a[12], const float extents[3]);
/* HalfSpace::new, crates/bevy_math/src/primitives/half_space.rs:53-57 */
void orc_half_space_new(const float normal_d[4], float out[4]);
/* Mat4::inverse (col-major 16) */
void orc_mat4_inverse(const float m[16], float out[16]);
void orc_mat4_mul(const float a[16], const float b[16], float out[16]);
void orc_mat4_mul_vec4(const float m[16], const float v[4], float out[4]);
/* glam::camera::rh::proj::directx::perspective_infinite_reverse (bevy_math/src/lib.rs:57) */
void orc_perspective_infinite_reverse(float fov, float aspect, float near, float out[16]);
/* CameraProjection::compute_frustum for PerspectiveProjection,
 * crates/bevy_camera/src/projection.rs:72-80,339-343; view_frustum.rs:43-108 */
void orc_compute_frustum_perspective(float fov, float aspect, float near, float far,
                                     const float camera_affine[12], float out_frustum[24]);
/* ViewFrustum::from_clip_from_world (far = row 2), view_frustum.rs:43-47 */
void orc_frustum_from_clip_from_world(const float clip_from_world[16], float out_frustum[24]);

/* Frustum::intersects_sphere, crates/bevy_camera/src/primitives.rs:255-268 */
int orc_frustum_intersects_sphere(const float frustum[24], const float center[3], float radius,
                                  int intersect_far);
/* Frustum::intersects_obb, primitives.rs:272-294 */
int orc_frustum_intersects_obb(const float frustum[24], const float aabb_center[3],
                               const float aabb_half[3], const float world_from_local[12],
                               int intersect_near, int intersect_far);
/* Frustum::intersects_obb_identity, primitives.rs:299-309 */
int orc_frustum_intersects_obb_identity(const float frustum[24], const float aabb_center[3],
                                        const float aabb_half[3]);
/* Frustum::contains_aabb / Aabb::is_in_half_space, primitives.rs:134-143,314-321 */
int orc_frustum_contains_aabb(const float frustum[24], const float aabb_center[3],
                              const float aabb_half[3], const float world_from_local[12]);
int orc_aabb_is_in_half_space(const float aabb_center[3], const float aabb_half[3],
                              const float half_space[4], const float world_from_local[12]);
int orc_aabb_is_in_half_space_identity(const float aabb_center[3], const float aabb_half[3],
                                       const float half_space[4]);
/* Sphere::intersects_obb, primitives.rs:219-226 */
int orc_sphere_intersects_obb(const float sphere_center[3], float sphere_radius,
                              const float aabb_center[3], const float aabb_half[3],
                              const float world_from_local[12]);

/* ---- transform propagation -------------------------------------------------------------- */

/* sync_simple_transforms, crates/bevy_transform/src/systems.rs:42-79.
 * Rows with dirty==NULL or dirty[i]!=0 get G = From(T).  Rows are assumed flat (no ChildOf,
 * no Children).  changed_out (optional, u8 per row) is set for every written row. */
void orc_sync_simple_transforms(uint32_t n, const float* translation, const float* rotation,
                                const float* scale, const uint8_t* dirty, float* global,
                                uint8_t* changed_out);

/* mark_dirty_trees, systems.rs:111-306: tree_changed |= ancestors-or-self of every changed row. */
void orc_mark_dirty_trees(uint32_t n, const uint32_t* parent, const uint8_t* changed,
                          uint8_t* tree_changed);

/* mark_dirty_trees + sync_simple_transforms + propagate_parent_transforms (parallel flavour,
 * systems.rs:506-748) over rows in ARBITRARY order.  parent[i] = row of the ChildOf target or
 * ORC_NO_PARENT.  static_opt != 0 -> StaticTransformOptimizations::Enabled.
 *   tree_changed (u8[n] or NULL = all changed): TransformTreeChanged.is_changed()
 *   transform_changed (u8[n] or NULL = all): Changed<Transform> || Added<GlobalTransform>
 *       (only consulted for flat rows -- those with neither parent nor children)
 *   global: in/out (old values are needed by set_if_neq, systems.rs:719)
 *   changed_out (u8[n], optional): 1 where GlobalTransform's change tick would be bumped.
 * Returns 0, or -1 if the parent array contains a cycle / out-of-range parent
 * (the reference panics, systems.rs:715). */
int orc_propagate_transforms(uint32_t n, const uint32_t* parent, const float* translation,
                             const float* rotation, const float* scale, int static_opt,
                             const uint8_t* tree_changed, const uint8_t* transform_changed,
                             float* global, uint8_t* changed_out);

/* TransformHelper::compute_global_transform, crates/bevy_transform/src/helper.rs:28-50:
 * independent definition (leaf affine, left-multiplied by each ancestor's Transform). */
int orc_compute_global_transform(uint32_t n, const uint32_t* parent, const float* translation,
                                 const float* rotation, const float* scale, uint32_t row,
                                 float out[12]);

/* ---- visibility ------------------------------------------------------------------------- */

/* reset_view_visibility, crates/bevy_camera/src/visibility/mod.rs:733-737 (+ :270-274) */
void orc_reset_view_visibility(uint32_t n, const uint8_t* flags, uint8_t* view_visibility);
/* check_visibility_cpu_culling, visibility/mod.rs:748-876.
 *   in_range: optional u8[n_views*n] (VisibleEntityRanges::entity_is_in_range_of_view), NULL = all in range
 *   visible_out: u8[n_views*n], 1 where the entity reached set_visible() for that view
 *   vv_changed_out: optional u8[n], 1 where set_visible() bumped the change tick (:290-306) */
void orc_check_visibility(uint32_t n, const float* global, const float* aabb_center,
                          const float* aabb_half, const uint8_t* flags, const uint32_t* layer_mask,
                          const uint8_t* in_range, uint8_t* view_visibility, const float* frusta,
                          const uint32_t* view_layer_masks, const uint8_t* view_flags,
                          uint32_t n_views, uint8_t* visible_out, uint8_t* vv_changed_out);
/* ... with RenderLayers of up to 64 layers (render_layers.rs:121-135: the first u64 word of the bitset); *_hi = layers 32..63, NULL = none */
void orc_check_visibility_layers64(uint32_t n, const float* global, const float* aabb_center, const float* aabb_half,
                                   const uint8_t* flags, const uint32_t* layer_mask, const uint32_t* layer_mask_hi, const uint8_t* in_range,
                                   uint8_t* view_visibility, const float* frusta, const uint32_t* view_layer_masks,
                                   const uint32_t* view_layer_masks_hi, const uint8_t* view_flags, uint32_t n_views, uint8_t* visible_out,
                                   uint8_t* vv_changed_out);
/* One view of any kind the main-world visibility systems test entities against. */
typedef struct orc_view {
    float frustum[24];
    uint32_t layer_mask;
    uint32_t flags;        /* ORC_VIEW_FLAG_* */
    float position[3];     /* GlobalTransform::translation of the view used for VisibilityRange distances */
    float light_sphere[4]; /* point/spot light: (translation, range) */
} orc_view;

/* check_visibility_ranges, crates/bevy_camera/src/visibility/range.rs:225-284.
 *   range_start_end f32[2n]: (start_margin.start, end_margin.end) -- all is_visible_at_all (:159-161) reads
 *   in_range_out u8[n_views*n]: VisibleEntityRanges::entity_is_in_range_of_view (0 for rows without a range) */
void orc_check_visibility_ranges(uint32_t n, const float* global, const float* aabb_center,
                                 const uint8_t* flags, const float* range_start_end,
                                 const float* view_positions, uint32_t n_views, uint8_t* in_range_out);

/* parity-margin census (tests/test_parity_margin.py): hist[0..4] = deciding plane-test values within 1/4/16/64/1024 ulps of zero, hist[5] = all */
void orc_visibility_margin_census(uint32_t n, const float* global, const float* aabb_center, const float* aabb_half,
                                  const uint8_t* flags, const uint32_t* layer_mask, const float* frusta,
                                  const uint32_t* view_layer_masks, uint32_t n_views, uint64_t hist[6]);

/* The per-entity closures of check_visibility_cpu_culling (visibility/mod.rs:788-858),
 * check_dir_light_mesh_visibility (bevy_light/src/lib.rs:425-475) and check_point_light_mesh_visibility
 * (lib.rs:592-650 cube faces, :694-738 spot), selected per view by orc_view.flags; every survivor calls
 * set_visible().  Visibility ranges are evaluated from range_start_end and each view's position
 * (range_start_end == NULL: no VisibleEntityRanges resource). */
void orc_check_visibility_views(uint32_t n, const float* global, const float* aabb_center,
                                const float* aabb_half, const uint8_t* flags, const uint32_t* layer_mask,
                                const float* range_start_end, uint8_t* view_visibility,
                                const orc_view* views, uint32_t n_views, uint8_t* visible_out,
                                uint8_t* vv_changed_out);

/* visibility_propagate_system + propagate_recursive, visibility/mod.rs:638-729, as the fixpoint they
 * maintain: Visible -> true, Hidden -> false, Inherited -> the parent's InheritedVisibility (true without a
 * parent or when the parent lacks the visibility components).
 *   visibility u8[n]: 0 Inherited, 1 Hidden, 2 Visible, 0x80 = entity has no Visibility/InheritedVisibility
 *   inherited  u8[n] in/out; changed_out u8[n] optional (1 where InheritedVisibility was assigned).
 * Returns -1 on a cycle / out-of-range parent. */
int orc_visibility_propagate(uint32_t n, const uint32_t* parent, const uint8_t* visibility,
                             uint8_t* inherited, uint8_t* changed_out);

/* check_visibility_gpu_culling, visibility/mod.rs:884-903 (applied to every NoCpuCulling row) */
void orc_check_visibility_gpu_culling(uint32_t n, const uint8_t* flags, uint8_t* view_visibility,
                                      uint8_t* vv_changed_out);
/* mark_newly_hidden_entities_invisible, visibility/mod.rs:908-918 */
void orc_mark_newly_hidden(uint32_t n, const uint8_t* flags, uint8_t* view_visibility,
                           uint8_t* vv_changed_out);
/* VisibleEntities for one (view, class): sorted Entity::to_bits keys (mod.rs:852-874).
 * visible: u8[n] for this view; class_mask: u32[n]; returns count written to out_keys/out_rows. */
uint32_t orc_visible_entities_sorted(uint32_t n, const uint8_t* visible, const uint32_t* class_mask,
                                     uint32_t class_bit, const uint64_t* entity_keys,
                                     uint64_t* out_keys, uint32_t* out_rows);

/* ---- light clustering ------------------------------------------------------------------- */

#define ORC_OBJ_POINT_LIGHT 0
#define ORC_OBJ_SPOT_LIGHT 1
#define ORC_OBJ_RECT_LIGHT 2
#define ORC_OBJ_REFLECTION_PROBE 3
#define ORC_OBJ_IRRADIANCE_VOLUME 4
#define ORC_OBJ_DECAL 5

#define ORC_MAX_CLUSTER_DIM 4096

typedef struct orc_cluster_view {
    uint32_t dims[3];
    uint32_t tile_size[2];
    uint32_t screen_size[2];
    uint32_t is_orthographic;
    uint32_t view_layer_mask;
    float near_; /* clusters.near = first_slice_depth (scaled) */
    float far_;  /* clusters.far */
    float cluster_factors[2];
    float view_from_world[16]; /* col-major Mat4 */
    float clip_from_view[16];
    float view_from_clip[16];
    float view_from_world_scale[3];
    float view_from_world_scale_max;
    float frustum[24];
    uint32_t n_x_planes, n_y_planes, n_z_planes;
    float x_planes[(ORC_MAX_CLUSTER_DIM + 1) * 4];
    float y_planes[(ORC_MAX_CLUSTER_DIM + 1) * 4];
    float z_planes[(ORC_MAX_CLUSTER_DIM + 1) * 4];
    uint32_t view_layer_mask_hi; /* layers 32..63 of the view's RenderLayers (orc_cluster_view_setup leaves 0) */
} orc_cluster_view;

/* ClusterConfig::dimensions_for_screen_size (FixedZ), crates/bevy_light/src/cluster/mod.rs:311-347 */
void orc_cluster_dimensions_fixed_z(uint32_t total, uint32_t z_slices, uint32_t screen_w,
                                    uint32_t screen_h, uint32_t out_dims[3]);
/* Clusters::update, cluster/mod.rs:398-416 */
void orc_clusters_update(uint32_t screen_w, uint32_t screen_h, const uint32_t requested[3],
                         uint32_t out_tile_size[2], uint32_t out_dims[3]);
/* ClusterConfig (cluster/mod.rs:107-139) and the part of assign_objects_to_clusters that turns it, the viewport and
 * last frame's statistics into what Clusters::update and the per-view setup consume (assign.rs:324-404):
 *   kind 0 None, 1 Single, 2 XYZ, 3 FixedZ;  far_z_mode 0 MaxClusterableObjectRange, 1 Constant.
 *   last_farthest_z / last_total: Clusters::last_frame_* (NULL = None).
 * Returns 0 when the view is cleared (ClusterConfig::None or an empty viewport, :328-339), else 1 with
 *   out_requested_dims (dimensions_for_screen_size + dynamic_resizing :384-404), out_first_slice_depth
 *   (config.first_slice_depth()), out_far_z (:350-355). */
typedef struct orc_cluster_config {
    uint32_t kind;
    uint32_t dimensions[3];
    uint32_t total, z_slices;
    float first_slice_depth;
    uint32_t far_z_mode;
    float far_z_constant;
    uint32_t dynamic_resizing;
} orc_cluster_config;
void orc_cluster_config_default(orc_cluster_config* out);
int orc_cluster_config_resolve(const orc_cluster_config* config, const float* last_farthest_z, const uint64_t* last_total,
                               uint32_t screen_w, uint32_t screen_h, uint64_t view_cluster_bindings_max_indices,
                               uint32_t out_requested_dims[3], float* out_first_slice_depth, float* out_far_z);
/* assign.rs:297-321: sort_by_cached_key((object_type.ordering(), entity)) + truncate when the gathered objects exceed
 * max_uniform_buffer_clusterable_objects on a platform without storage buffers.  order[n] receives the surviving
 * objects (indices into the gathered arrays); returns how many survive. */
uint32_t orc_cluster_sort_truncate(uint32_t n, const uint8_t* obj_type, const uint8_t* shadow_maps_enabled,
                                   const uint8_t* volumetric, const uint64_t* entity_bits, uint32_t max_objects,
                                   int supports_storage_buffers, uint32_t* order);

/* Per-view setup of assign_objects_to_clusters, cluster/assign.rs:342-485.
 * far_z: the value selected by ClusterFarZMode (:350-355); first_slice_depth_cfg: config value. */
void orc_cluster_view_setup(const float camera_affine[12], const float clip_from_view[16],
                            const float frustum[24], uint32_t screen_w, uint32_t screen_h,
                            const uint32_t requested_dims[3], float first_slice_depth_cfg,
                            float far_z, uint32_t view_layer_mask, orc_cluster_view* out);
/* compute_aabb_for_cluster -> bounding sphere, assign.rs:693-707,834-900.  out[4] = center, radius */
void orc_cluster_aabb_sphere(const orc_cluster_view* view, uint32_t x, uint32_t y, uint32_t z,
                             float out[4]);

/* Per-object loop of assign_objects_to_clusters, assign.rs:487-811.
 * Objects are given in gather order (:190-296).  For spot lights spot_dir = transform.back()
 * (world space, unnormalised input is fine: it is re-normalised in view space :567-569) and
 * spot_sin_cos = sin_cos(outer_angle).
 * Output is the per-cluster Vec<Entity> in push order, flattened CSR:
 *   offsets u32[C+1], indices (object index) u32[capacity], counts u32[6*C] by object type.
 * Returns total_cluster_index_count (may exceed capacity: then only counts/offsets are valid). */
uint64_t orc_assign_objects_to_clusters(const orc_cluster_view* view, uint32_t n_objects,
                                        const float* pos_range /*4n*/, const uint8_t* obj_type,
                                        const uint32_t* obj_layer_mask /*NULL = all default*/,
                                        const float* spot_dir /*3n or NULL*/,
                                        const float* spot_sin_cos /*2n or NULL*/,
                                        uint32_t* offsets, uint32_t* indices, uint64_t capacity,
                                        uint32_t* counts, float* farthest_z_out);
/* The same with RenderLayers 32..63 of the objects (obj_layer_mask_hi, NULL = none) against view->view_layer_mask_hi:
 * RenderLayers::intersects over the first u64 word of the bitset (crates/bevy_camera/src/visibility/render_layers.rs:121-135). */
uint64_t orc_assign_objects_to_clusters_layers64(const orc_cluster_view* view, uint32_t n_objects, const float* pos_range,
                                                 const uint8_t* obj_type, const uint32_t* obj_layer_mask,
                                                 const uint32_t* obj_layer_mask_hi, const float* spot_dir,
                                                 const float* spot_sin_cos, uint32_t* offsets, uint32_t* indices,
                                                 uint64_t capacity, uint32_t* counts, float* farthest_z_out);

/* bench.py's CPU baseline of the cluster stage: the per-object loop ONCE per frame, every touched cluster's list grown by push
 * (assign.rs:740-800; the lists keep their capacity from frame to frame) -- `iters` frames, returns the seconds they took.
 * (orc_assign_objects_to_clusters walks the objects twice to hand out a CSR; as a baseline that would flatter the device.) */
double orc_bench_assign_objects_to_clusters(const orc_cluster_view* view, uint32_t n_objects, const float* pos_range,
                                            const uint8_t* obj_type, const float* spot_dir, const float* spot_sin_cos, int iters,
                                            uint64_t* total_out, float* farthest_z_out);

/* The GPU wire format of one view's clusters, storage-buffer flavour: extract_clusters_for_cpu_clustering +
 * prepare_clusters_for_cpu_clustering, crates/bevy_pbr/src/cluster/mod.rs:394-476,478-582, push_offset_and_counts
 * :634-650, push_index / push_dummy_index :690-700.  Input = the per-cluster lists (CSR) and counts the assignment
 * produced; remap[object] = GlobalClusterableObjectMeta::entity_to_index / render light-probe index (0xFFFFFFFF =
 * not loaded yet -> push_dummy_index); NULL = identity.
 *   out_offsets_and_counts u32[8*C]: uvec4(offset, point, spot, rect), uvec4(refl probes, irradiance vols, decals, 0)
 *   out_index_list u32[total] */
void orc_cluster_bindings_storage(uint32_t n_clusters, const uint32_t* offsets, const uint32_t* counts,
                                  const uint32_t* indices, const uint32_t* remap,
                                  uint32_t* out_offsets_and_counts, uint32_t* out_index_list);

/* The transform / bounds part of the mesh-instance wire format: MeshInputUniform::world_from_local =
 * Affine3::to_transpose() (crates/bevy_math/src/affine3.rs:27-34; crates/bevy_pbr/src/render/mesh.rs:568-571) and
 * MeshCullingData::new (mesh.rs:1646-1657).  out_world_from_local f32[12] = three Vec4 rows, out_culling f32[8]. */
void orc_mesh_inputs(const float global[12], const float aabb_center[3], const float aabb_half[3], int has_aabb,
                     float out_world_from_local[12], float out_culling[8]);

/* ---- CPU baseline drivers (Bevy-shaped: ceil(n/threads) batches, batching.rs:95-106) ---- */

/* One frame of the flat path on `threads` pthreads: sync_simple_transforms (all dirty) +
 * reset + check_visibility (n_views) + mark_newly_hidden.  Returns seconds for `iters` frames. */
double orc_bench_flat_frame(uint32_t n, const float* translation, const float* rotation,
                            const float* scale, const float* aabb_center, const float* aabb_half,
                            const uint8_t* flags, const uint32_t* layer_mask, float* global,
                            uint8_t* view_visibility, uint8_t* visible_out, const float* frusta,
                            const uint32_t* view_layer_masks, const uint8_t* view_flags,
                            uint32_t n_views, int threads, int iters);
/* the same with reset + check (every view) + mark_newly_hidden fused into one pass per batch when fused_visibility != 0 */
double orc_bench_flat_frame2(uint32_t n, const float* translation, const float* rotation,
                            const float* scale, const float* aabb_center, const float* aabb_half,
                            const uint8_t* flags, const uint32_t* layer_mask, float* global,
                            uint8_t* view_visibility, uint8_t* visible_out, const float* frusta,
                            const uint32_t* view_layer_masks, const uint8_t* view_flags,
                            uint32_t n_views, int threads, int iters, int fused_visibility);

double orc_bench_tree_frame(uint32_t n, const uint32_t* parent, const uint32_t* level_offsets, uint32_t n_levels, const float* t,
                            const float* r, const float* s, float* g, int threads, int iters);

/* ---- batching work-item build (batching_oracle.c) ---------------------------------------- */
#define ORC_NO_BATCH_SET 0xFFFFFFFFu
typedef struct orc_binned_mesh_instance { uint32_t input_uniform_index, bin_index; } orc_binned_mesh_instance; /* render_phase/mod.rs:777 */
typedef struct orc_bin_metadata { uint32_t indirect_parameters_offset, bin_index, instance_count; } orc_bin_metadata; /* mesh_preprocess_types.wesl:130-149 */
typedef struct orc_preprocess_work_item { uint32_t input_index, output_or_indirect_parameters_index; } orc_preprocess_work_item; /* gpu_preprocessing.rs:783-799 */
typedef struct orc_indirect_parameters_metadata {  /* gpu_preprocessing.rs:898-934 */
    uint32_t base_output_index, batch_set_index, mesh_index, early_instance_count, late_instance_count;
} orc_indirect_parameters_metadata;
typedef struct orc_indirect_batch_set { uint32_t indirect_parameters_count, indirect_parameters_base; } orc_indirect_batch_set; /* :946-965 */
typedef struct orc_batch_set_record {  /* what BinnedRenderPhaseBatchSet keeps, gpu_preprocessing.rs:2560-2577 */
    uint32_t set, indexed, index, first_work_item_index, instance_count, first_indirect_parameters_index, batch_count,
        first_output_mesh_uniform_index;
} orc_batch_set_record;
typedef struct orc_batch_initial {  /* lengths of the phase's buffers before the multidrawable pass, [0] non-indexed [1] indexed */
    uint32_t work_item_index[2], indirect_parameters_index[2], batch_set_index[2], output_mesh_uniform_index;
} orc_batch_initial;
typedef struct orc_batch_totals {
    uint32_t work_item_len[2], indirect_parameters_len[2], batch_set_len[2], data_buffer_len, n_records;
} orc_batch_totals;
void orc_unpack_bins(uint32_t base_output_work_item_index, uint32_t base_indirect_parameters_index,
                     uint32_t binned_mesh_instance_count, const orc_binned_mesh_instance* binned_mesh_instances,
                     const orc_bin_metadata* bin_metadata, const uint32_t* bin_index_to_bin_metadata_index,
                     orc_preprocess_work_item* preprocess_work_items);
void orc_allocate_uniforms(uint32_t batch_set_index, uint32_t bin_count, uint32_t first_indirect_parameters_index,
                           uint32_t first_output_mesh_uniform_index, const orc_bin_metadata* bin_metadata,
                           orc_indirect_parameters_metadata* indirect_parameters_metadata, uint32_t* fan_buffer);
uint32_t orc_batch_build(uint32_t n_list, const uint32_t* rows, const uint32_t* row_batch_set, const uint32_t* row_bin_index,
                         const uint32_t* row_input_uniform_index, uint32_t n_sets, const uint8_t* set_indexed,
                         const uint32_t* bin_table_offset, const uint32_t* bin_table, const uint32_t* meta_offset,
                         orc_bin_metadata* bin_metadata, const orc_batch_initial* initial,
                         orc_preprocess_work_item* work_items[2], orc_indirect_parameters_metadata* metadata[2],
                         orc_indirect_batch_set* batch_sets[2], orc_batch_set_record* records, orc_batch_totals* totals);

/* ---- the CPU-built part of a binned phase and the sorted phases (batching_oracle.c) ------------------------ */
#define ORC_ROW_MULTIDRAWABLE 0u /* row_kind: the row's place is row_batch_set (or nowhere) */
#define ORC_ROW_BATCHABLE 1u     /* in batchable bin row_bin */
#define ORC_ROW_UNBATCHABLE 2u   /* in unbatchable bin row_bin */
#define ORC_NO_INPUT_INDEX 0xFFFFFFFFu /* GetFullBatchData::get_binned_index / get_index_and_compare_data returned None */
#define ORC_NO_INDEX 0xFFFFFFFFu       /* PhaseItemExtraIndex::None */
#define ORC_RECORD_BATCHABLE_BIN 0x80000000u /* orc_batch_set_record.set of a batchable bin's batch */
typedef struct orc_unbatchable_index { uint32_t bin, instance_index; } orc_unbatchable_index; /* UnbatchableBinnedEntityIndices */
uint32_t orc_batch_cpu_bins(uint32_t n_list, const uint32_t* rows, const uint8_t* row_kind, const uint32_t* row_bin,
                            const uint32_t* row_input_uniform_index, uint32_t n_unbatchable_bins, const uint8_t* unbatchable_indexed,
                            uint32_t n_batchable_bins, const uint8_t* batchable_indexed, int no_indirect_drawing,
                            const orc_batch_initial* initial, orc_preprocess_work_item* work_items[2],
                            orc_indirect_parameters_metadata* metadata[2], orc_indirect_batch_set* batch_sets[2],
                            orc_unbatchable_index* unbatchable_indices, uint32_t* n_unbatchable_indices,
                            orc_batch_set_record* records, orc_batch_totals* totals);
#define ORC_ITEM_INDEXED 1u          /* SortedPhaseItem::indexed() */
#define ORC_ITEM_HAS_COMPARE_DATA 2u /* the Option<(BatchSetCompareData, BatchCompareData)> is Some */
typedef struct orc_sorted_item { uint32_t input_index, batch_set_key, bin_key, flags; } orc_sorted_item;
typedef struct orc_sorted_batch {  /* what flush() leaves on the batch set's first phase item, gpu_preprocessing.rs:1767-1794 */
    uint32_t first_item, instance_start, instance_end, indirect_parameters_start, indirect_parameters_end, indexed;
} orc_sorted_batch;
uint32_t orc_batch_sorted(uint32_t n_items, const orc_sorted_item* items, int automatic_batching, int no_indirect_drawing,
                          const orc_batch_initial* initial, orc_preprocess_work_item* work_items[2],
                          orc_indirect_parameters_metadata* metadata[2], orc_indirect_batch_set* batch_sets[2],
                          orc_sorted_batch* batches, orc_batch_totals* totals);
uint32_t orc_batch_sorted_merge(uint32_t n_items, const orc_sorted_item* items, int automatic_batching, uint32_t first_index,
                                orc_sorted_batch* batches, uint32_t* buffer_len);

#ifdef __cplusplus
}
#endif
#endif
