/*
 * bevy_mi355x.h -- C ABI of the MI355X-native render-prep path for Bevy
 *                  (transform propagate -> frustum cull -> light-cluster assign).
 *
 * This is the drop-in boundary: exactly the entry points a Rust `bevy_mi355x` plugin crate
 * binds with `#[link(name = "bevy_mi355x")] unsafe extern "C" { ... }` to replace the stock
 * CPU systems (INTEGRATION.md shows the binding and the Plugin that registers it).  Plain
 * pointers and sizes only -- no C++/torch types.  All reference citations are relative to
 * the Bevy checkout (v0.20.0-dev).
 *
 * Conventions
 *   - Every function returns an int32 status: MI_OK (0) or a negative MI_ERR_*.  Nothing
 *     throws or aborts.  MI_ERR_MALFORMED_HIERARCHY is what the shim turns into the
 *     reference's panic (crates/bevy_transform/src/systems.rs:715); any MI_ERR_DEVICE means
 *     "run the stock CPU system this frame".
 *   - Host slices are owned by the caller (the ECS) and only borrowed for the duration of the
 *     call: the library copies in to pinned staging it owns and copies out into caller
 *     buffers.  Device memory is library-owned and persists across frames.
 *   - A context may be used from any thread, one call at a time (Bevy systems have no thread
 *     affinity: crates/bevy_ecs/src/schedule/executor/multi_threaded.rs:241); every entry
 *     point selects the context's device itself.
 *   - Rows are dense indices 0..n chosen by the shim.  When a hierarchy is uploaded rows must
 *     be in level (BFS) order -- mi_hierarchy_sort() computes that order from an arbitrary
 *     ChildOf array.
 *   - Layouts (f32 unless noted): translation[3n], rotation[4n] (x,y,z,w), scale[3n];
 *     GlobalTransform[12n] = Affine3A::to_cols_array() (x_axis,y_axis,z_axis,translation);
 *     Aabb center[3n], half_extents[3n]; a Sphere component is (center, half_extents.x = radius);
 *     Frustum = 6 x (nx,ny,nz,d) in ViewFrustum order left,right,top,bottom,near,far
 *     (crates/bevy_math/src/primitives/view_frustum.rs:25-34).
 */
#ifndef BEVY_MI355X_H
#define BEVY_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3: mi_cluster_view.view_layer_mask_hi, component-granular upload windows (round 4); mi_check_light_mesh_visibility
 *    (round 5).  A caller built against an older header must not pass the version check. */
#define MI_ABI_VERSION 3

/* ---- status codes ---------------------------------------------------------------------- */
#define MI_OK 0
#define MI_ERR_INVALID_ARG (-1)
#define MI_ERR_DEVICE (-2)
#define MI_ERR_OUT_OF_MEMORY (-3)
#define MI_ERR_MALFORMED_HIERARCHY (-4) /* systems.rs:715 assert_eq!(child_of.parent(), parent) */
#define MI_ERR_NOT_READY (-5)           /* a required column / hierarchy / cull result is missing */
#define MI_ERR_CAPACITY (-6)            /* caller buffer too small; required size is reported */

/* ---- per-entity flag byte (mi_upload_bounds) ------------------------------------------- */
#define MI_FLAG_INHERITED_VISIBLE 0x01u  /* InheritedVisibility::get(), visibility/mod.rs:804 */
#define MI_FLAG_NO_FRUSTUM_CULLING 0x02u /* Has<NoFrustumCulling>, visibility/mod.rs:768,823 */
#define MI_FLAG_HAS_AABB 0x04u           /* Option<&Aabb> is Some, visibility/mod.rs:824 */
#define MI_FLAG_HAS_SPHERE 0x08u         /* Option<&Sphere> is Some (Aabb takes precedence), :838 */
#define MI_FLAG_NO_CPU_CULLING 0x10u     /* With<NoCpuCulling>: excluded from a11/a12, handled by :884-903 */
#define MI_FLAG_HAS_VISIBILITY_RANGE 0x20u /* Has<VisibilityRange>, :814-820 (see mi_upload_visibility_ranges) */
#define MI_FLAG_RANGE_USE_AABB 0x40u     /* VisibilityRange::use_aabb, visibility/range.rs:255-263 */
#define MI_FLAG_SHADOW_CASTER 0x80u      /* With<Mesh3d>, Without<NotShadowCaster>, Without<DirectionalLight>:
                                            matched by the shadow-view queries, crates/bevy_light/src/lib.rs:355-372 */

/* A bounding Sphere that FOLLOWS its entity: update_point_light_bounding_spheres (crates/bevy_light/src/point_light.rs:195-208) keeps
 * Sphere { center: GlobalTransform::translation, radius: PointLight::range } on every point light.  A row with MI_FLAG_HAS_SPHERE (and
 * no Aabb) whose aabb_half_extents is (radius, this bit pattern, anything) gets its centre from the row's OWN GlobalTransform
 * translation as the frame's propagate leaves it -- that system folded into the frame: a moving light costs the host nothing (no
 * bounds upload), and the Sphere is never a frame late (in the reference it is inserted through Commands).  aabb_center is ignored.
 * The value is a quiet NaN: memcpy the bits into the float array. */
#define MI_SPHERE_AT_TRANSLATION 0x7FC0A11Du

/* ---- per-view flags (mi_view.flags; mi_cull's view_flags byte carries the low bits) ------ */
#define MI_VIEW_FLAG_NO_CPU_CULLING 0x01u /* camera Has<NoCpuCulling>: skip frustum tests, :756,823 */
#define MI_VIEW_FLAG_SHADOW 0x02u         /* shadow view (cascade / cube face / spot): only MI_FLAG_SHADOW_CASTER rows,
                                             OBB test only; crates/bevy_light/src/lib.rs:342-757 */
#define MI_VIEW_FLAG_SKIP_NEAR 0x04u      /* intersects_obb(.., intersect_near = false, ..): cascades, lib.rs:455-458 */
#define MI_VIEW_FLAG_TEST_FAR 0x08u       /* intersects_obb(.., .., intersect_far = true): every shadow view */
#define MI_VIEW_FLAG_LIGHT_SPHERE 0x10u   /* point / spot: light_sphere.intersects_obb pre-test, lib.rs:617-623 */
#define MI_VIEW_FLAG_RANGES 0x20u         /* this view has an index in VisibleEntityRanges (range.rs:238-243):
                                             ranged rows are tested against mi_view.position */
#define MI_VIEW_FLAG_RANGES_NO_ORIGIN 0x40u /* point / spot shadow view without a shadow LOD origin
                                             (lib.rs:601-611): ranged rows are culled */
/* typical combinations */
#define MI_VIEW_KIND_CASCADE (MI_VIEW_FLAG_SHADOW | MI_VIEW_FLAG_SKIP_NEAR | MI_VIEW_FLAG_TEST_FAR)
#define MI_VIEW_KIND_CUBE_FACE_OR_SPOT (MI_VIEW_FLAG_SHADOW | MI_VIEW_FLAG_TEST_FAR | MI_VIEW_FLAG_LIGHT_SPHERE)

/* ---- Visibility component values (mi_upload_visibility) ----------------------------------- */
#define MI_VISIBILITY_INHERITED 0u
#define MI_VISIBILITY_HIDDEN 1u
#define MI_VISIBILITY_VISIBLE 2u
#define MI_VISIBILITY_NONE 0x80u /* the entity has no Visibility / InheritedVisibility components */

/* ---- mi_cull / mi_propagate_and_cull flags ------------------------------------------------- */
#define MI_CULL_BEGIN_FRAME 0x1u /* also apply reset_view_visibility (mod.rs:733-737) in the same pass */
#define MI_CULL_END_FRAME 0x2u   /* also apply check_visibility_gpu_culling + mark_newly_hidden_entities_invisible
                                    (mod.rs:884-918) in the same pass: valid when nothing else (e.g. shadow-view
                                    culling, bevy_light/src/lib.rs:499-510) ORs into ViewVisibility this frame */
#define MI_CULL_MORE_FRAMES 0x4u /* another cull frame follows at once: this frame's VisibleEntities compaction is deferred into
                                   * the tail workgroups of that frame's kernel -- one launch per frame instead of two (any
                                   * dispatch costs >= 4.3 us on this part).  If something else comes first, the entry points
                                   * that expose the lists (mi_download_visible_entities, mi_batch_build,
                                   * mi_device_buffer(MI_BUF_VISIBLE_ROWS), mi_synchronize, mi_columns_resize) enqueue the
                                   * compaction themselves, so results never depend on the flag; masks and ViewVisibility
                                   * are unaffected.  With the multi-GPU exchange on, the frame's all-gather is issued together
                                   * with its compaction, i.e. one call later (mi_exchange_last joins first). */

#define MI_CULL_WITH_CLUSTERS 0x8u /* the frame also assigns the row-bound lights (mi_cluster_bind_objects_to_rows) to the clusters of the
                                   * view uploaded with mi_cluster_upload_view -- the whole metric frame, propagate + cull + cluster, in ONE
                                   * call: the assignment is enqueued behind the cull and reads the ViewVisibility column as this call
                                   * leaves it (NoCpuCulling rows get theirs when the frame is closed: pass MI_CULL_END_FRAME or call
                                   * mi_visibility_end_frame first if there are any).  Same results as mi_cluster_assign_resident. */
#define MI_CULL_CLUSTERS_CONCURRENT 0x10u /* with MI_CULL_WITH_CLUSTERS, on a call that decides the frame's ViewVisibility on its own
                                   * (MI_CULL_BEGIN_FRAME and MI_CULL_END_FRAME both apply, <= 8 views, no hierarchy): the assignment does
                                   * not wait for the frame kernel -- it re-derives each light's ViewVisibility::get() with the cull's own
                                   * rule and runs on a second stream next to it.  Identical results.  Measured on MI355X it does NOT pay at
                                   * 100 k lights (the two kernels of the assignment are latency-bound and stretch when they share the
                                   * chip: 46.3 us per frame against 44.1 us one behind the other), so it is opt-in. */

#define MI_CULL_STATIC_OPT 0x40u   /* mi_propagate_and_cull(_views) with a hierarchy uploaded: StaticTransformOptimizations::Enabled for the
                                   * propagate part (= MI_PROPAGATE_STATIC_OPT, systems.rs:87-103) */
#define MI_CULL_CHANGED_ROWS 0x20u /* mi_propagate_and_cull only: propagate the rows whose Transform change byte is set (mi_upload_changed,
                                   * mi_upload_transforms_indexed; new rows carry it as Added<GlobalTransform>) -- the filter of
                                   * sync_simple_transforms, systems.rs:45-50 -- instead of every row; the others keep their GlobalTransform and
                                   * their change tick.  Same results as mi_propagate(0) followed by mi_cull(MI_CULL_BEGIN_FRAME | ...), in one
                                   * launch: the steady-state frame of a scene in which few entities move.  The change bytes are consumed. */

/* ---- mi_propagate flags ----------------------------------------------------------------- */
#define MI_PROPAGATE_ALL_DIRTY 0x1u  /* every Transform counts as changed (worst case / first frame) */
#define MI_PROPAGATE_STATIC_OPT 0x2u /* StaticTransformOptimizations::Enabled, systems.rs:87-103 */

#define MI_NO_PARENT 0xFFFFFFFFu

/* ---- clusterable object types (gather order of assign.rs:190-296) ---------------------- */
#define MI_OBJ_POINT_LIGHT 0
#define MI_OBJ_SPOT_LIGHT 1
#define MI_OBJ_RECT_LIGHT 2
#define MI_OBJ_REFLECTION_PROBE 3
#define MI_OBJ_IRRADIANCE_VOLUME 4
#define MI_OBJ_DECAL 5

typedef struct mi_ctx mi_ctx;

/* ======================================================================================= */
/* lifecycle                                                                                 */
/* ======================================================================================= */

/* Creates a context on HIP device `device`.  `hip_stream` is an existing hipStream_t to enqueue
 * on (e.g. the host framework's current stream) or NULL to let the library create its own.
 * Replaces: nothing in Bevy -- this is the Plugin::build-time setup (crates/bevy_app/src/plugin.rs:57-92). */
int32_t mi_ctx_create(int32_t device, void* hip_stream, mi_ctx** out_ctx);
int32_t mi_ctx_destroy(mi_ctx* ctx);
/* Human-readable description of the last error on this context (or of the last failed
 * mi_ctx_create when ctx == NULL).  Never NULL. */
const char* mi_last_error_string(mi_ctx* ctx);
/* Blocks until everything enqueued on the context's stream has finished. */
int32_t mi_synchronize(mi_ctx* ctx);
int32_t mi_abi_version(void);

/* ======================================================================================= */
/* component columns (ECS table columns -> device-resident SoA)                              */
/* Bulk source on the Bevy side: Query::contiguous_iter() per-table slices                   */
/* (crates/bevy_ecs/src/system/query.rs:1509-1560)                                           */
/* ======================================================================================= */

/* Sets the number of live rows (grows device columns geometrically; contents of existing rows are kept). */
int32_t mi_columns_resize(mi_ctx* ctx, uint32_t n_rows);

/* Transform { translation: Vec3, rotation: Quat, scale: Vec3 },
 * crates/bevy_transform/src/components/transform.rs:86-106.  Rows [first_row, first_row+n). */
int32_t mi_upload_transforms(mi_ctx* ctx, uint32_t first_row, uint32_t n, const float* translation,
                             const float* rotation, const float* scale);

/* Sparse form for the steady state: exactly the rows a `Changed<Transform>` query yields this frame
 * (crates/bevy_transform/src/systems.rs:45-50,111-116), in any order.  One staging block, one scatter kernel;
 * also raises the rows' "changed" byte, so a following mi_propagate(0) treats precisely these rows as dirty. */
int32_t mi_upload_transforms_indexed(mi_ctx* ctx, uint32_t n, const uint32_t* rows, const float* translation,
                                     const float* rotation, const float* scale);

/* The same two uploads WITHOUT the staging copy: the library hands out a window of its pinned (page-locked, device-mapped) host
 * memory and the caller -- whose gather loop over the ECS tables has to write the values somewhere anyway -- fills it in place:
 *   mi_map_upload_window(capacity, flags)   rows[capacity] (NULL with MI_UPLOAD_DENSE), translation[3 capacity], rotation[4 capacity],
 *                                           scale[3 capacity]; valid until its commit.  Several windows may be mapped at once (a
 *                                           parallel gather fills one per thread; map and commit themselves are calls on the
 *                                           context: one at a time) and other calls may come in between.  Every mapped window is
 *                                           committed exactly once (n = 0 just gives it back)
 *   mi_commit_upload_window(w, n, first_row) the first n entries go to the device: MI_UPLOAD_DENSE = rows [first_row, first_row + n)
 *                                           by DMA straight from the window (mi_upload_transforms); otherwise rows[i] in any order,
 *                                           scattered by one kernel that reads the window over PCIe and raises the rows' change
 *                                           bytes (mi_upload_transforms_indexed; first_row ignored).
 * Two shapes of commit let the results of the frame travel ahead of it (mi_download_frame_results, below): dense windows that
 * carry the whole flat table one after the other, and one indexed window whose rows strictly ascend or descend. */
#define MI_UPLOAD_DENSE 0x1u
/* Component-granular windows: the window carries ONLY the named components of the rows (any combination; none of the three bits =
 * all three, as before) -- the pointers of the others are NULL and their columns keep what they hold.  A scene in which every cube
 * rotates (examples/stress_tests/many_cubes.rs:641-648, --rotate-cubes) sends 16 bytes per row instead of 40; one in which things
 * only move, 12.  Everything else about a window is unchanged (dense or indexed, sequences in pieces, results travelling ahead). */
#define MI_UPLOAD_TRANSLATION 0x2u
#define MI_UPLOAD_ROTATION 0x4u
#define MI_UPLOAD_SCALE 0x8u
typedef struct mi_upload_window {
    uint32_t* rows;
    float* translation;
    float* rotation;
    float* scale;
    uint32_t capacity, flags;
    uint64_t token; /* library bookkeeping */
} mi_upload_window;
int32_t mi_map_upload_window(mi_ctx* ctx, uint32_t capacity, uint32_t flags, mi_upload_window* out);
int32_t mi_commit_upload_window(mi_ctx* ctx, const mi_upload_window* w, uint32_t n, uint32_t first_row);

/* Existing GlobalTransform values (global_transform.rs:60).  Only needed when the previous values
 * matter: set_if_neq change detection (systems.rs:719) and MI_PROPAGATE_STATIC_OPT skipping. */
int32_t mi_upload_global_transforms(mi_ctx* ctx, uint32_t first_row, uint32_t n, const float* global12);

/* Aabb (crates/bevy_camera/src/primitives.rs:63-68) or Sphere (:197-211), the MI_FLAG_* byte and the
 * first 32 RenderLayers bits (crates/bevy_camera/src/visibility/render_layers.rs:20; default layer 0 = mask 1).
 * flags / layer_mask may be NULL (= MI_FLAG_INHERITED_VISIBLE|MI_FLAG_HAS_AABB, mask 1). */
int32_t mi_upload_bounds(mi_ctx* ctx, uint32_t first_row, uint32_t n, const float* aabb_center,
                         const float* aabb_half_extents, const uint8_t* flags, const uint32_t* layer_mask);

/* RenderLayers 32..63 of the rows (optional column; rows never uploaded have none).  RenderLayers::intersects
 * (crates/bevy_camera/src/visibility/render_layers.rs:121-135) compares the masks word by word: with this column and
 * mi_view.layer_mask_hi the first 64-bit word is covered; entities on layers >= 64 stay with the stock system. */
int32_t mi_upload_render_layers_hi(mi_ctx* ctx, uint32_t first_row, uint32_t n, const uint32_t* layer_mask_hi);

/* ViewVisibility's packed byte (bit0 current, bit1 previous; visibility/mod.rs:226-275). */
int32_t mi_upload_view_visibility(mi_ctx* ctx, uint32_t first_row, uint32_t n, const uint8_t* view_visibility);

/* VisibilityClass (visibility/mod.rs:208-211) mapped by the shim to a bit set of small class ids,
 * and Entity::to_bits() (crates/bevy_ecs/src/entity/mod.rs:566-568), the VisibleEntities sort key. */
int32_t mi_upload_visibility_classes(mi_ctx* ctx, uint32_t first_row, uint32_t n, const uint32_t* class_mask);
int32_t mi_upload_entity_keys(mi_ctx* ctx, uint32_t first_row, uint32_t n, const uint64_t* entity_bits);

/* Per-row "Changed<Transform> || Changed<ChildOf> || Added<GlobalTransform> || orphaned" byte:
 * the `changed` query + RemovedComponents of mark_dirty_trees / sync_simple_transforms
 * (systems.rs:42-55,111-116).  Consumed (cleared) by the next mi_propagate. */
int32_t mi_upload_changed(mi_ctx* ctx, uint32_t first_row, uint32_t n, const uint8_t* changed);

/* VisibilityRange (crates/bevy_camera/src/visibility/range.rs:78-103): start_end[2n] =
 * (start_margin.start, end_margin.end), the two bounds is_visible_at_all reads (:159-161); only consulted for
 * rows with MI_FLAG_HAS_VISIBILITY_RANGE.  check_visibility_ranges (:225-284) is then evaluated on the device
 * inside mi_cull, per view, from mi_view.position.  start_end == NULL = no VisibleEntityRanges resource (ranged
 * rows are not range-culled, visibility/mod.rs:814-816 `is_some_and`). */
int32_t mi_upload_visibility_ranges(mi_ctx* ctx, uint32_t first_row, uint32_t n, const float* start_end);

/* Visibility component per row (MI_VISIBILITY_*), input of mi_visibility_propagate. */
int32_t mi_upload_visibility(mi_ctx* ctx, uint32_t first_row, uint32_t n, const uint8_t* visibility);

/* ChildOf as a row index (MI_NO_PARENT for roots), rows in level order: level l occupies rows
 * [level_offsets[l], level_offsets[l+1]); level 0 holds every root and every flat entity;
 * parent_idx[row] must lie in the previous level.  Validates and returns
 * MI_ERR_MALFORMED_HIERARCHY otherwise.  n_levels == 1 / parent_idx == NULL means "all rows flat".
 * Replaces the Children/ChildOf walk of propagate_descendants_unchecked (systems.rs:679-748). */
int32_t mi_upload_hierarchy(mi_ctx* ctx, uint32_t n, const uint32_t* parent_idx, const uint32_t* level_offsets,
                            uint32_t n_levels);

/* Host helper: computes the level order for an arbitrary ChildOf array.
 *   parent[n]            : row of each row's parent in the CALLER's order (MI_NO_PARENT = root)
 *   out_new_to_old[n]    : permutation, new row -> old row (stable: siblings keep caller order)
 *   out_parent_idx[n]    : parent in NEW row numbering, ready for mi_upload_hierarchy
 *   out_level_offsets    : capacity level_capacity entries; *out_n_levels + 1 are written
 * Returns MI_ERR_MALFORMED_HIERARCHY for cycles / out-of-range parents, MI_ERR_CAPACITY if
 * level_capacity is too small.  Pure host code (no ctx). */
int32_t mi_hierarchy_sort(uint32_t n, const uint32_t* parent, uint32_t* out_new_to_old, uint32_t* out_parent_idx,
                          uint32_t* out_level_offsets, uint32_t level_capacity, uint32_t* out_n_levels);

/* Host helper: how mi_propagate would walk a hierarchy of these level sizes, and whether the STOCK systems should keep it
 * (round 6).  propagate_parent_transforms (crates/bevy_transform/src/systems.rs:506-657) walks a tree depth first on a CPU core: a
 * node is ~20 ns of arithmetic whatever the shape.  The device wins by running thousands of nodes side by side; a hierarchy in which
 * NO level holds more than a wave of rows (transform_hierarchy.rs's `chain`: 2 500 levels of one node; a rope; one rig) has nothing
 * to run side by side -- it is levels x the latency of one dependent level step (~0.32 us on MI355X with at most 16 rows to a level,
 * ~0.85 us with up to 64: one wave's instruction stream, k_propagate_narrow), i.e. 26 x a CPU core's time on `chain`.  For such a World the plugin keeps mark_dirty_trees /
 * propagate_parent_transforms / sync_simple_transforms registered (the host layers: bevy_amd/host/bevy_mi355x_host.hpp,
 * rust/bevy_mi355x/src/lib.rs) and hands the GlobalTransforms to the visibility stage.  Pure host code (no ctx).
 *   level_offsets[n_levels + 1]   as for mi_upload_hierarchy (mi_hierarchy_sort's output) */
#define MI_HIERARCHY_PLAN_FLAT 0u      /* one level: every row a root (k_frame / k_level0_propagate) */
#define MI_HIERARCHY_PLAN_TILES 1u     /* subtree tiles (or a wave per tree, or level by level): the device has rows to run side by side */
#define MI_HIERARCHY_PLAN_ONE_WAVE 2u  /* every level at most 64 rows and more than 16 levels: one wave walks the whole hierarchy */
typedef struct mi_hierarchy_advice {
    uint32_t plan;           /* MI_HIERARCHY_PLAN_* */
    uint32_t keep_on_host;   /* 1 = the stock CPU systems are expected to be faster for this hierarchy */
    uint32_t n_levels;
    uint32_t widest_level;   /* rows of the widest level */
    float est_device_us;     /* the model behind keep_on_host: a launch + levels x the dependent level step (plan ONE_WAVE; 0 otherwise) */
    float est_host_us;       /* rows x 20 ns on one core */
} mi_hierarchy_advice;
int32_t mi_hierarchy_advice_for(uint32_t n_levels, const uint32_t* level_offsets, mi_hierarchy_advice* out);

/* ======================================================================================= */
/* systems                                                                                   */
/* ======================================================================================= */

/* mark_dirty_trees + sync_simple_transforms + propagate_parent_transforms
 * (crates/bevy_transform/src/systems.rs:111-306, 42-79, 506-748) in one call.
 * Writes GlobalTransform and the per-row "change tick bumped" bit. */
int32_t mi_propagate(mi_ctx* ctx, uint32_t flags);

/* visibility_propagate_system (visibility/mod.rs:638-729): recomputes InheritedVisibility (bit
 * MI_FLAG_INHERITED_VISIBLE of the flag byte) from the Visibility column over the uploaded hierarchy
 * (flat rows: Hidden -> false, otherwise true); assigns only where the value differs. */
int32_t mi_visibility_propagate(mi_ctx* ctx);
/* InheritedVisibility::get() per row (0/1) and the bitmask of rows the system assigned (change ticks). */
int32_t mi_download_inherited_visibility(mi_ctx* ctx, uint32_t first_row, uint32_t n, uint8_t* out_inherited,
                                         uint32_t* changed_bitmask);

/* reset_view_visibility (visibility/mod.rs:733-737) -- call once per frame before mi_cull
 * (or pass MI_CULL_BEGIN_FRAME to mi_cull and skip this call). */
int32_t mi_visibility_begin_frame(mi_ctx* ctx);

/* check_visibility_cpu_culling (visibility/mod.rs:748-876) for n_views ACTIVE cameras in one pass over
 * the columns: sets ViewVisibility bit0 (set_visible, :290-306), writes one packed visibility bitmask
 * per view, and builds the per-view, per-class VisibleEntities lists sorted by Entity bits (:861-874).
 *   frusta[24*n_views]; view_layer_masks[n_views] (NULL = layer 0); view_flags[n_views] (NULL = 0).
 * (2 .. 4 camera views run intersects_obb over (row, view) pairs instead of view by view -- same results; MI_MULTI_VIEW=1 in the
 * environment of mi_ctx_create keeps the per-view form, an A/B switch.) */
int32_t mi_cull(mi_ctx* ctx, const float* frusta, const uint32_t* view_layer_masks, const uint8_t* view_flags,
                uint32_t n_views, uint32_t flags /* MI_CULL_* */);

/* One view of any kind the main-world visibility systems test entities against: a camera
 * (check_visibility_cpu_culling), a directional-light cascade (check_dir_light_mesh_visibility,
 * crates/bevy_light/src/lib.rs:342-515) or a point-light cube face / spot-light frustum
 * (check_point_light_mesh_visibility, lib.rs:517-757).  Every kind ORs into ViewVisibility (set_visible) and gets
 * its own bitmask / VisibleEntities (VisibleMeshEntities) lists, indexed by its position in the array. */
typedef struct mi_view {
    float frustum[24];     /* Frustum::half_spaces */
    uint32_t layer_mask;   /* RenderLayers of the camera / light: layers 0..31 (layers 32..63: layer_mask_hi below) */
    uint32_t flags;        /* MI_VIEW_FLAG_* */
    float position[3];     /* GlobalTransform::translation of the view for VisibilityRange distances: the camera,
                              the cascade's camera (lib.rs:437-443) or the shadow LOD origin (lib.rs:601-611) */
    float light_sphere[4]; /* point / spot light: (translation, range) */
    uint32_t layer_mask_hi; /* layers 32..63 of the view's RenderLayers (the first u64 word of the reference's bitset,
                               render_layers.rs:121-135); matched against mi_upload_render_layers_hi's column */
    uint32_t reserved[2];  /* 0 */
} mi_view;
int32_t mi_cull_views(mi_ctx* ctx, const mi_view* views, uint32_t n_views, uint32_t flags /* MI_CULL_* */);
int32_t mi_propagate_and_cull_views(mi_ctx* ctx, const mi_view* views, uint32_t n_views, uint32_t flags);

/* The frame in one call: propagate + reset_view_visibility + check_visibility_cpu_culling.  Flat rows (no hierarchy uploaded):
 * sync_simple_transforms with every Transform dirty and the cull in ONE pass (Transform is read once, GlobalTransform is written
 * once and never re-read).  With a hierarchy: the tile launches of mi_propagate, the cull enqueued behind them.  Results are
 * identical to mi_propagate(MI_PROPAGATE_ALL_DIRTY); mi_visibility_begin_frame(); mi_cull(...) [; mi_visibility_end_frame()
 * when flags has MI_CULL_END_FRAME].  MI_CULL_BEGIN_FRAME is implied.  With MI_CULL_CHANGED_ROWS only the rows marked
 * changed are propagated (= mi_propagate(0) in front of the cull); MI_CULL_STATIC_OPT = MI_PROPAGATE_STATIC_OPT. */
int32_t mi_propagate_and_cull(mi_ctx* ctx, const float* frusta, const uint32_t* view_layer_masks,
                              const uint8_t* view_flags, uint32_t n_views, uint32_t flags /* MI_CULL_* */);

/* check_dir_light_mesh_visibility + check_point_light_mesh_visibility (crates/bevy_light/src/lib.rs:342-515, 517-757;
 * SimulationLightSystems::CheckLightVisibility, ordered after VisibilitySystems::CheckVisibility and before
 * MarkNewlyHiddenEntitiesInvisible, lib.rs:217-230): the frame's SHADOW views in one pass over the resident columns, behind the
 * camera pass of the same frame (mi_cull / mi_propagate_and_cull* WITHOUT MI_CULL_END_FRAME).  Every view must carry
 * MI_VIEW_FLAG_SHADOW: one mi_view per cascade of every (directional light, camera) pair (MI_VIEW_KIND_CASCADE; layer mask = the
 * light's RenderLayers; MI_VIEW_FLAG_RANGES + position = that camera's translation when the camera has an index in
 * VisibleEntityRanges), six per shadow-mapped point light and one per spot light among the lights some camera sees
 * (MI_VIEW_KIND_CUBE_FACE_OR_SPOT + light_sphere = (translation, range); MI_VIEW_FLAG_RANGES + position = the shadow LOD origin's
 * translation, or MI_VIEW_FLAG_RANGES_NO_ORIGIN without one, lib.rs:601-611).  Only rows with MI_FLAG_SHADOW_CASTER are seen.
 * Survivors are ORed into ViewVisibility (set_visible: lib.rs:499-510, 629, 723); flags = MI_CULL_END_FRAME also closes the frame
 * (check_visibility_gpu_culling + mark_newly_hidden_entities_invisible) in the same pass.  n_views = 0 is allowed (only the
 * END_FRAME part runs).  A host that keeps ViewVisibility in its ECS and lets the stock mark_newly_hidden_entities_invisible run
 * there (the Rust plugin) may also close the cameras' frame at once (MI_CULL_END_FRAME on that call) and pass flags = 0 here: bit 0
 * of the device's column -- ViewVisibility::get(), all the next frame's reset reads -- comes out the same; only the device's own
 * previous-frame bit and change mask of THIS frame then differ from the reference's.  Out, in ONE device wait:
 *   out_bitmasks[n_views * ceil(n_rows/32)]  view v's packed VisibleMeshEntities: bit r = row r was pushed to that cascade's /
 *                                            face's / spot light's list (the caller maps rows to Entity and sorts, lib.rs:489, 664, 745)
 *   out_any[ceil(n_rows/32)]                 OR over the views = the rows set_visible() was called on (NULL = not wanted)
 * Afterwards view indices of mi_download_visibility / mi_download_visible_entities / mi_batch_build refer to THESE views (the
 * shadow phases are batched from them); fetch the cameras' lists before this call. */
int32_t mi_check_light_mesh_visibility(mi_ctx* ctx, const mi_view* shadow_views, uint32_t n_views, uint32_t flags /* MI_CULL_END_FRAME or 0 */,
                                       uint32_t* out_bitmasks, uint32_t* out_any);

/* check_visibility_gpu_culling for NoCpuCulling rows (visibility/mod.rs:884-903) followed by
 * mark_newly_hidden_entities_invisible (:908-918). */
int32_t mi_visibility_end_frame(mi_ctx* ctx);

/* ======================================================================================= */
/* results                                                                                   */
/* ======================================================================================= */

/* GlobalTransform rows [first_row, first_row+n) and (optional) bitmask, bit (row-first_row) set where
 * the reference would have bumped GlobalTransform's change tick.  first_row must be a multiple of 32
 * when changed_bitmask != NULL.  changed_bitmask has ceil(n/32) words. */
int32_t mi_download_global_transforms(mi_ctx* ctx, uint32_t first_row, uint32_t n, float* out_global12,
                                      uint32_t* changed_bitmask);

/* Sparse form for the steady state: only the rows whose GlobalTransform change tick the reference would have
 * bumped in the last propagate (ascending rows + their matrices, compacted and gathered on the device), so the
 * write-back through `Mut<GlobalTransform>` touches -- and the PCIe copy carries -- only what changed.
 * *out_count receives the number of changed rows; MI_ERR_CAPACITY if it exceeds `capacity`. */
int32_t mi_download_changed_global_transforms(mi_ctx* ctx, uint32_t* out_rows, float* out_global12, uint32_t capacity,
                                              uint32_t* out_count);

/* The same rows in the render world's mesh-instance wire format (next row 8f-3 i): MeshInputUniform::world_from_local
 * = the affine transposed to three Vec4 rows (crates/bevy_math/src/affine3.rs:27-34, crates/bevy_pbr/src/render/
 * mesh.rs:568-571) and MeshCullingData (aabb center / half extents as Vec4; infinite half extents without an Aabb,
 * mesh.rs:1646-1657), i.e. the transform / bounds part of what extract_meshes_for_gpu_building (mesh.rs:1933-2262)
 * rebuilds per changed mesh on the CPU.  The remaining MeshInputUniform fields are renderer bookkeeping. */
int32_t mi_download_changed_mesh_inputs(mi_ctx* ctx, uint32_t* out_rows, float* out_world_from_local12, float* out_culling8,
                                        uint32_t capacity, uint32_t* out_count);

/* Packed per-view visibility of the last mi_cull: bit r of word r/32 = row r reached set_visible()
 * for that view.  bitmask has ceil(n_rows/32) words. */
int32_t mi_download_visibility(mi_ctx* ctx, uint32_t view, uint32_t* bitmask);

/* ViewVisibility bytes and (optional) bitmask of rows whose change tick the reference would bump
 * (set_visible hidden->visible, set_if_neq, mark_newly_hidden) since mi_visibility_begin_frame. */
int32_t mi_download_view_visibility(mi_ctx* ctx, uint32_t first_row, uint32_t n, uint8_t* out_view_visibility,
                                    uint32_t* changed_bitmask);

/* VisibleEntities::get(class) of one view (visibility/mod.rs:342-402): entity keys ascending
 * (= sort_unstable on Entity) and the matching rows.  Either output may be NULL.
 * *out_count receives the list length; MI_ERR_CAPACITY if it exceeds `capacity`. */
int32_t mi_download_visible_entities(mi_ctx* ctx, uint32_t view, uint32_t class_bit, uint64_t* out_entity_keys,
                                     uint32_t* out_rows, uint32_t capacity, uint32_t* out_count);

/* ======================================================================================= */
/* light clustering -- assign_objects_to_clusters, crates/bevy_light/src/cluster/assign.rs:137-813 */
/* ======================================================================================= */

/* Per-view constants of assign.rs:342-485, computed by the shim with glam (or mi_cluster_view_build). */
typedef struct mi_cluster_view {
    uint32_t dims[3];        /* Clusters::dimensions */
    uint32_t tile_size[2];   /* Clusters::tile_size */
    uint32_t screen_size[2]; /* Camera::physical_viewport_size */
    uint32_t is_orthographic;
    uint32_t view_layer_mask;
    float near_; /* Clusters::near (first_slice_depth * view_from_world_scale.z) */
    float far_;  /* Clusters::far */
    float cluster_factors[2];
    float view_from_world[16]; /* column-major Mat4 */
    float clip_from_view[16];
    float view_from_clip[16];
    float view_from_world_scale[3];
    float view_from_world_scale_max;
    float frustum[24];
    const float* x_planes; /* (dims[0]+1) x 4, view space, assign.rs:434-476 */
    const float* y_planes; /* (dims[1]+1) x 4 */
    const float* z_planes; /* (dims[2]+1) x 4, assign.rs:478-485 */
    /* Bounding spheres of every cluster AABB (compute_aabb_for_cluster, assign.rs:693-707,834-900),
     * dims[0]*dims[1]*dims[2] x (cx,cy,cz,r), index (y*dims[0]+x)*dims[2]+z.  Only read for spot lights;
     * may be NULL when there are none. */
    const float* cluster_spheres;
    /* RenderLayers 32..63 of the view (with view_layer_mask the first u64 word of the reference's bitset, render_layers.rs:121-135);
     * matched against mi_cluster_upload_object_layers_hi's column.  mi_cluster_view_build leaves 0. */
    uint32_t view_layer_mask_hi;
} mi_cluster_view;

/* Host helper (pure host code): fills *out from the camera's GlobalTransform, clip_from_view, Frustum,
 * viewport and the already-resolved ClusterConfig values (requested dims, first_slice_depth, far_z of
 * ClusterFarZMode).  plane_storage must hold (dx+dy+dz+3)*4 floats and sphere_storage dx*dy*dz*4 floats
 * (NULL = skip spheres) for the dims this call computes; query them first with mi_cluster_view_dims. */
int32_t mi_cluster_view_dims(uint32_t screen_w, uint32_t screen_h, const uint32_t requested_dims[3],
                             uint32_t out_tile_size[2], uint32_t out_dims[3]);
int32_t mi_cluster_view_build(const float camera_affine[12], const float clip_from_view[16], const float frustum[24],
                              uint32_t screen_w, uint32_t screen_h, const uint32_t requested_dims[3],
                              float first_slice_depth, float far_z, uint32_t view_layer_mask, float* plane_storage,
                              float* sphere_storage, mi_cluster_view* out);
/* ClusterConfig::dimensions_for_screen_size for FixedZ (cluster/mod.rs:311-347). */
int32_t mi_cluster_dimensions_fixed_z(uint32_t total, uint32_t z_slices, uint32_t screen_w, uint32_t screen_h,
                                      uint32_t out_dims[3]);

/* ---- ClusterConfig and the per-frame feedback of assign_objects_to_clusters ------------------------------------
 * ClusterConfig (crates/bevy_light/src/cluster/mod.rs:107-139; Default = FixedZ{4096, 24, first_slice_depth 5.0,
 * MaxClusterableObjectRange, dynamic_resizing} :288-308), the two statistics a frame leaves for the next one
 * (Clusters::last_frame_farthest_z / last_frame_total_cluster_index_count, cluster/mod.rs:152-164, written at
 * assign.rs:810-811) and what assign.rs:324-404 derives from them before Clusters::update: the requested grid
 * (dimensions_for_screen_size :311-347, then dynamic_resizing against view_cluster_bindings_max_indices =
 * ViewClusterBindings::MAX_INDICES, crates/bevy_pbr/src/cluster/mod.rs:587, assign.rs:384-404), the configured first
 * slice depth (:349-365 finishes it inside mi_cluster_view_build) and the far_z chosen by ClusterFarZMode (:350-355,
 * DEFAULT_FAR_DEPTH = 1000 on the first frame, :37). */
#define MI_CLUSTER_CONFIG_NONE 0u
#define MI_CLUSTER_CONFIG_SINGLE 1u
#define MI_CLUSTER_CONFIG_XYZ 2u
#define MI_CLUSTER_CONFIG_FIXED_Z 3u
#define MI_CLUSTER_FAR_Z_MAX_CLUSTERABLE_OBJECT_RANGE 0u
#define MI_CLUSTER_FAR_Z_CONSTANT 1u
#define MI_VIEW_CLUSTER_BINDINGS_MAX_INDICES 16384u /* ViewClusterBindings::MAX_INDICES */
#define MI_MAX_UNIFORM_BUFFER_CLUSTERABLE_OBJECTS 204u /* crates/bevy_pbr/src/cluster/mod.rs:31 */
typedef struct mi_cluster_config {
    uint32_t kind;            /* MI_CLUSTER_CONFIG_* */
    uint32_t dimensions[3];   /* XYZ */
    uint32_t total, z_slices; /* FixedZ */
    float first_slice_depth;  /* ClusterZConfig::first_slice_depth (XYZ, FixedZ) */
    uint32_t far_z_mode;      /* MI_CLUSTER_FAR_Z_* (XYZ, FixedZ) */
    float far_z_constant;     /* ClusterFarZMode::Constant */
    uint32_t dynamic_resizing;
} mi_cluster_config;
typedef struct mi_cluster_history {
    uint32_t has_farthest_z;  /* Option::is_some */
    float farthest_z;
    uint32_t has_total_cluster_index_count;
    uint32_t reserved;
    uint64_t total_cluster_index_count;
} mi_cluster_history;
typedef struct mi_cluster_resolved {
    uint32_t active;            /* 0 = Clusters::clear(): ClusterConfig::None or an empty viewport (assign.rs:328-339) */
    uint32_t requested_dims[3]; /* after dynamic resizing: what Clusters::update receives */
    float first_slice_depth;    /* ClusterConfig::first_slice_depth() */
    float far_z;                /* the value ClusterFarZMode selects */
} mi_cluster_resolved;
/* ClusterConfig::default() */
int32_t mi_cluster_config_default(mi_cluster_config* out);
/* Pure host code.  last == NULL = first frame (both statistics None). */
int32_t mi_cluster_config_resolve(const mi_cluster_config* config, const mi_cluster_history* last, uint32_t screen_w,
                                  uint32_t screen_h, uint64_t view_cluster_bindings_max_indices, mi_cluster_resolved* out);
/* The UBO fallback of the gather (assign.rs:297-321): when storage buffers are unsupported and more than max_objects
 * objects were gathered, they are sorted (stable, sort_by_cached_key) by (ClusterableObjectType::ordering(), Entity) --
 * ordering() = (type, !shadow_maps_enabled, !volumetric) for point / spot lights, (type, false, false) otherwise
 * (:108-128) -- and truncated.  out_order[0 .. *out_n) = the surviving objects as indices into the gathered arrays, in
 * the order the per-object loop then visits them; identity when nothing has to be dropped.  shadow_maps_enabled /
 * volumetric may be NULL (= false).  Pure host code. */
int32_t mi_cluster_sort_truncate(uint32_t n, const uint8_t* obj_type, const uint8_t* shadow_maps_enabled, const uint8_t* volumetric,
                                 const uint64_t* entity_bits, uint32_t max_objects, uint32_t supports_storage_buffers,
                                 uint32_t* out_order, uint32_t* out_n);

/* The per-object loop (assign.rs:487-811) over n_objects clusterable objects given in gather order
 * (point, spot, rect, reflection probes, irradiance volumes, decals; :190-296):
 *   pos_range[4n]   world translation + range (ClusterableObjectAssignmentData::sphere, :52-59)
 *   obj_type[n]     MI_OBJ_* (NULL = all point lights)
 *   layer_mask[n]   RenderLayers bits (NULL = default layer)
 *   spot_dir[3n]    GlobalTransform::back() of spot lights, spot_sin_cos[2n] = sin_cos(outer_angle) (:563-573)
 * Output = every cluster's Vec<Entity> in push order, flattened:
 *   out_offsets[C+1], out_indices[capacity] (object index), out_counts[6*C] (ClusterableObjectCounts order),
 *   out_total = last_frame_total_cluster_index_count, out_farthest_z = last_frame_farthest_z (:810-811).
 * Returns MI_ERR_CAPACITY (with out_total/out_offsets/out_counts valid) when capacity < total.
 * Supported host libm: the z slice of a depth is floor(ln(z) * scale - bias) (view_z_to_z_slice, assign.rs:1003-1024), and the
 * device evaluates ln with glibc's logf (>= 2.28, x86-64: correctly rounded in all but a handful of inputs, reproduced bit for bit,
 * mi_debug_logf, tests/test_gpu_parity.py::test_device_logf_matches_libm).  A Bevy built against another libm (musl, macOS, Windows' UCRT) may round ln(z)
 * differently for depths within an ulp of a slice boundary: such an object could then differ by one z slice from that host's own
 * assign_objects_to_clusters.  The library does not assume: the first perspective view uploaded on a context compares the device's
 * logf with the host's over a fixed table of 3 297 probes, and on ANY difference mi_cluster_upload_view (hence every assignment)
 * returns MI_ERR_DEVICE from then on -- "run the stock CPU system": clusters fall back, propagate and cull are unaffected.  Everything else on the device is +, -, *, /, sqrt, floor, min, max: IEEE, no library; the host
 * helpers (mi_cluster_view_build's powf, mi_perspective_clip_from_view's sin / cos) call the host's own libm, as Bevy would. */
int32_t mi_cluster_assign(mi_ctx* ctx, const mi_cluster_view* view, uint32_t n_objects, const float* pos_range,
                          const uint8_t* obj_type, const uint32_t* layer_mask, const float* spot_dir,
                          const float* spot_sin_cos, uint32_t* out_offsets, uint32_t* out_indices, uint64_t capacity,
                          uint32_t* out_counts, uint64_t* out_total, float* out_farthest_z);

/* Device-resident variant for steady-state frames: objects are uploaded once and re-assigned per frame. */
int32_t mi_cluster_upload_objects(mi_ctx* ctx, uint32_t n_objects, const float* pos_range, const uint8_t* obj_type,
                                  const uint32_t* layer_mask, const float* spot_dir, const float* spot_sin_cos);
/* RenderLayers 32..63 of the uploaded objects (n_objects as in the last mi_cluster_upload_objects, which clears this column again;
 * NULL clears it too).  Objects on layers >= 64 stay with the stock system. */
int32_t mi_cluster_upload_object_layers_hi(mi_ctx* ctx, uint32_t n_objects, const uint32_t* layer_mask_hi);
int32_t mi_cluster_upload_view(mi_ctx* ctx, const mi_cluster_view* view);
/* Lights that are rows of this context (they have a Transform, a bounding Sphere and a ViewVisibility like every other
 * entity: update_point_light_bounding_spheres, crates/bevy_light/src/point_light.rs:195-208): object i of the uploaded
 * object arrays IS row first_row + i.  The assignment then applies the gather of assign.rs:190-296 on the device --
 * an object takes part only if its row's ViewVisibility::get() is true (the cull of the same frame decided that), its
 * sphere centre is the row's GlobalTransform translation (:198, :52-59) and a spot light's direction the row's
 * GlobalTransform::back() (:567) -- so nothing about the lights crosses PCIe per frame; pos_range[4i+3] (range),
 * obj_type, layer_mask and spot_sin_cos stay as uploaded.  Object indices in the output are positions in the
 * uploaded arrays (= row - first_row); objects that are not visible simply appear in no cluster, exactly like
 * entities the reference never gathered.  n_objects must equal the uploaded object count; n_objects = 0 unbinds. */
int32_t mi_cluster_bind_objects_to_rows(mi_ctx* ctx, uint32_t first_row, uint32_t n_objects);
/* The same for lights that are NOT a contiguous block of rows (with a hierarchy rows are in level order, and a light parented to
 * something sits wherever its level puts it): object i IS row rows[i].  Everything else as above; a later mi_columns_resize that
 * shrinks the context unbinds. */
int32_t mi_cluster_bind_objects_to_row_list(mi_ctx* ctx, uint32_t n_objects, const uint32_t* rows);
int32_t mi_cluster_assign_resident(mi_ctx* ctx, uint64_t* out_total);
/* Several clustered views.  Every camera with a ClusterConfig has Clusters of its own (assign.rs:324-486 runs per view; split screen,
 * a picture-in-picture camera): the context keeps MI_CLUSTER_MAX_VIEWS view slots -- each with its view constants, the outputs of its
 * assignment and its history of buffers -- over ONE set of resident objects (mi_cluster_upload_objects and the row bindings are shared).
 * mi_cluster_select_view makes `slot` the one that mi_cluster_upload_view, mi_cluster_assign_resident / _frame, mi_cluster_download(_bindings),
 * MI_CULL_WITH_CLUSTERS and the cluster parts of mi_download_frame_results refer to (default: slot 0).  A frame of two clustered cameras:
 * select 0, upload its view, the frame call with MI_CULL_WITH_CLUSTERS (the walk rides in the frame kernel); select 1, upload its view,
 * mi_cluster_assign_resident (two launches behind the frame, same ViewVisibility); results per slot.  Selecting enqueues whatever the
 * slot that is left still owes (a deferred fill). */
#define MI_CLUSTER_MAX_VIEWS 8u
int32_t mi_cluster_select_view(mi_ctx* ctx, uint32_t slot);
/* One view of one frame exactly as the system runs it (assign.rs:324-811): resolve the config against `history`,
 * build and upload the view constants, assign the resident objects, and store this frame's
 * total_cluster_index_count / farthest_z back into `history` (:810-811).  out_view (optional) receives the view
 * constants that were used (plane / sphere pointers are library-owned and valid until the next call).
 * *out_active = 0 when the config resolved to Clusters::clear() (nothing assigned, history untouched). */
int32_t mi_cluster_assign_frame(mi_ctx* ctx, const mi_cluster_config* config, mi_cluster_history* history,
                                const float camera_affine[12], const float clip_from_view[16], const float frustum[24],
                                uint32_t screen_w, uint32_t screen_h, uint32_t view_layer_mask,
                                uint64_t view_cluster_bindings_max_indices, mi_cluster_view* out_view, uint32_t* out_active);
int32_t mi_cluster_download(mi_ctx* ctx, uint32_t* out_offsets, uint32_t* out_indices, uint64_t capacity,
                            uint32_t* out_counts, uint64_t* out_total, float* out_farthest_z);

/* Everything a frame hands back to the ECS in ONE call, one launch and one device wait: the device packs counts and lists into a
 * window of the library's pinned (page-locked, device-mapped) host memory, the host waits once and reads them there (results too
 * big for the window, or a cluster list that outgrew its device buffer, take two waits: counts, then lists; the separate downloads
 * wait ten times between them, and on an idle stream a wait is ~30 us).  The parts, each switched on by its flag:
 *   MI_RESULTS_CHANGED_ROWS / _GLOBALS   the rows whose GlobalTransform changed (ascending) / their matrices
 *                                        (mi_download_changed_global_transforms)
 *   lists[n_lists]                       VisibleEntities row lists of any (view, class) pairs (mi_download_visible_entities; rows
 *                                        only -- the caller maps rows to Entity itself -- and only when rows are numbered in key order,
 *                                        which is the case unless mi_upload_entity_keys said otherwise: MI_ERR_NOT_READY then).
 *                                        The union of a frame's lists is exactly the set of rows SetViewVisibility::set_visible
 *                                        was called on (visibility/mod.rs:846-857) -- except rows without a VisibilityClass.
 *   MI_RESULTS_CLUSTERS / _CLUSTER_INDICES  offsets + counts / the index list of the resident assignment (mi_cluster_download)
 * Without MI_RESULTS_IN_PLACE the caller passes buffers and capacities and the library copies out of the window.  With it the
 * library SETS the pointers to the data where it lies in the pinned window -- no copy; the caller reads it in place (e.g. while
 * writing the ECS components), must not write to it, and must be done before its next call on this context.  Capacities still bound
 * what is fetched.
 * Every GlobalTransform of a flat table, every frame (the all-dirty case): commit the Transforms as a SEQUENCE of dense windows that
 * carries the whole table in ascending order -- one window for every row, or several committed one after the other, each starting
 * where the one before ended (mi_commit_upload_window; the caller fills window k + 1 while window k crosses PCIe) -- then run the
 * all-rows frame and ask for the changed GlobalTransforms here.  From the second such frame on the library computes each window's
 * GlobalTransforms as soon as it has arrived and sends them back under the rest of the upload (PCIe is full duplex); this call then
 * finds them on the host.  They are handed out only when the frame in between rewrote every row from exactly those Transforms
 * (any other upload, resize or hierarchy in between: fetched the usual way); the changed-row list of such a frame is 0 .. n-1 and is
 * never fetched.  Nothing to switch on, same results either way.
 * Some GlobalTransforms of a flat table (the changed-rows frame): commit the moved rows through ONE indexed upload window whose
 * rows strictly ascend or strictly descend (a Changed<Transform> query in table order over rows numbered by Entity key is one or
 * the other) and run the MI_CULL_CHANGED_ROWS frame.  From the second such frame on the scatter launch that reads the window over
 * PCIe writes each row's GlobalTransform straight back into pinned memory, and this call hands out the window's rows (ascending)
 * and those GlobalTransforms without compacting the change mask, gathering or copying anything -- when the frame's change mask is
 * exactly that window: the change column was clean before it (the frame before consumed it) and nothing raised a mark, wrote a
 * Transform or propagated between the window and the frame.  Otherwise the usual way; same results.
 * Counts are always filled in for the parts that ran; MI_ERR_CAPACITY if a list exceeds its capacity (counts are valid, that list
 * was not delivered, the others were). */
#define MI_RESULTS_CHANGED_ROWS 0x1u
#define MI_RESULTS_CHANGED_GLOBALS 0x2u
#define MI_RESULTS_CLUSTERS 0x4u
#define MI_RESULTS_CLUSTER_INDICES 0x8u /* implies MI_RESULTS_CLUSTERS */
#define MI_RESULTS_IN_PLACE 0x10u
#define MI_RESULTS_MAX_LISTS 16u
typedef struct mi_visible_list {
    uint32_t view, class_bit; /* in: which VisibleEntities list */
    uint32_t capacity;        /* in: entries of rows (in place: the most the caller is prepared to read) */
    uint32_t count;           /* out */
    uint32_t* rows;           /* in: [capacity]; MI_RESULTS_IN_PLACE: out */
} mi_visible_list;
typedef struct mi_frame_results {
    /* ---- in ---- */
    uint32_t flags;            /* MI_RESULTS_* */
    uint32_t n_lists;          /* <= MI_RESULTS_MAX_LISTS */
    mi_visible_list* lists;    /* [n_lists] or NULL */
    uint32_t changed_capacity; /* rows of changed_rows / changed_global12 */
    uint32_t reserved;         /* 0 */
    uint64_t cluster_capacity; /* entries of cluster_indices */
    /* ---- in, or out with MI_RESULTS_IN_PLACE ---- */
    uint32_t* changed_rows;    /* [changed_capacity] */
    float* changed_global12;   /* [12 * changed_capacity] */
    uint32_t* cluster_offsets; /* [n_clusters + 1] */
    uint32_t* cluster_counts;  /* [6 * n_clusters] */
    uint32_t* cluster_indices; /* [cluster_capacity] */
    /* ---- out ---- */
    uint32_t changed_count;
    float farthest_z;
    uint64_t cluster_total;
} mi_frame_results;
int32_t mi_download_frame_results(mi_ctx* ctx, mi_frame_results* io);

/* The GPU wire format of the view's clusters, storage-buffer flavour, built on the device from the last
 * assignment: what extract_clusters_for_cpu_clustering + prepare_clusters_for_cpu_clustering assemble element by
 * element on the CPU (crates/bevy_pbr/src/cluster/mod.rs:394-476,478-582; push_offset_and_counts :634-650):
 *   out_offsets_and_counts[8*C]  per cluster uvec4(offset, point, spot, rect), uvec4(reflection probes, irradiance
 *                                volumes, decals, 0)
 *   out_index_list[total]        remap[object] for every entry in cluster order; remap = the shim's table
 *                                object -> GlobalClusterableObjectMeta::entity_to_index / render light-probe index
 *                                (0xFFFFFFFF = push_dummy_index, :698-700); NULL = the object index itself.
 * The same arrays stay on the device as MI_BUF_CLUSTER_OFFSETS_AND_COUNTS / MI_BUF_CLUSTER_INDEX_LIST. */
int32_t mi_cluster_download_bindings(mi_ctx* ctx, const uint32_t* remap, uint32_t n_remap, uint32_t* out_offsets_and_counts,
                                     uint32_t* out_index_list, uint64_t capacity, uint64_t* out_total);

/* ======================================================================================= */
/* batching work-item build                                                 */
/* ======================================================================================= */
/* The multidrawable part of batch_and_prepare_binned_render_phase
 * (crates/bevy_render/src/batching/gpu_preprocessing.rs:2360-2447, MultidrawableBatchSetPreparer::
 * prepare_multidrawable_binned_batch_set :2497-2580) together with the two compute passes it feeds --
 * allocate_uniforms.wesl and unpack_bins.wesl (crates/bevy_pbr/src/render/) -- built on the device straight from a
 * view's VisibleEntities list, which the cull pass left in HBM: no D2H of the list, no CPU bin hash maps
 * (render_phase/mod.rs:268-400), no H2D of GpuRenderBinnedMeshInstance arrays.
 *
 * The render world uploads, per row, which batch set the mesh instance belongs to (MI_NO_BATCH_SET = none; rows of the phase's
 * unbatchable / batchable-only bins are placed with mi_batch_upload_row_bins), its RenderBinIndex inside that set and its InputUniformIndex;
 * and per phase the batch sets in the order phase.multidrawable_meshes iterates them: mesh class (indexed or not),
 * the RenderBinIndex -> bin-metadata-index table (holes allowed) and the GpuBinMetadata array of every set, both
 * concatenated with offset arrays of n_sets + 1 entries.  instance_count of the metadata is an output.
 *
 * mi_batch_build(view, class_bit) then produces, for that view's list of that visibility class:
 *   work items      PreprocessWorkItem {input_index, output_or_indirect_parameters_index} per mesh class; a set's run
 *                   starts at its first_work_item_index and lists its visible instances in VisibleEntities order
 *   metadata        IndirectParametersMetadata {base_output_index, batch_set_index, 0, 0, 0} per bin, at
 *                   first_indirect_parameters_index + GpuBinMetadata.indirect_parameters_offset
 *   batch sets      IndirectBatchSet {indirect_parameters_count = 0, indirect_parameters_base}
 *   records         one mi_batch_set_record per non-empty set = what BinnedRenderPhaseBatchSet records (:2560-2577)
 *   totals          the lengths of the phase's buffers after the pass
 * A set none of whose instances is visible is skipped, like a batch set without bins (:2520-2524).  All indices start
 * at `initial` (the buffer lengths the unbatchable / batchable passes left, :2431-2447); entries below them are zero.
 * Mesh class index everywhere: 0 = non-indexed, 1 = indexed. */
#define MI_NO_BATCH_SET 0xFFFFFFFFu
typedef struct mi_bin_metadata {  /* GpuBinMetadata, mesh_preprocess_types.wesl:130-149 */
    uint32_t indirect_parameters_offset, bin_index, instance_count;
} mi_bin_metadata;
typedef struct mi_preprocess_work_item {  /* gpu_preprocessing.rs:783-799 */
    uint32_t input_index, output_or_indirect_parameters_index;
} mi_preprocess_work_item;
typedef struct mi_indirect_parameters_metadata {  /* gpu_preprocessing.rs:898-934 */
    uint32_t base_output_index, batch_set_index, mesh_index, early_instance_count, late_instance_count;
} mi_indirect_parameters_metadata;
typedef struct mi_indirect_batch_set {  /* gpu_preprocessing.rs:946-965 */
    uint32_t indirect_parameters_count, indirect_parameters_base;
} mi_indirect_batch_set;
/* One per non-empty multidrawable batch set (`set` = its index) and one per non-empty batchable bin (`set` =
 * MI_BATCH_RECORD_BATCHABLE_BIN | bin; first_work_item_index 0 -- "Unused", :2345-2350 --, batch_count 1, instance_count and
 * first_output_mesh_uniform_index = first_batch.instance_range, first_indirect_parameters_index = its extra index or MI_NO_INDEX). */
#define MI_BATCH_RECORD_BATCHABLE_BIN 0x80000000u
typedef struct mi_batch_set_record {
    uint32_t set, indexed, index, first_work_item_index, instance_count, first_indirect_parameters_index, batch_count,
        first_output_mesh_uniform_index;
} mi_batch_set_record;
typedef struct mi_unbatchable_index {  /* UnbatchableBinnedEntityIndices (:2195-2222): extra index = instance..instance+1, or None */
    uint32_t bin, instance_index;
} mi_unbatchable_index;
typedef struct mi_batch_initial {
    uint32_t work_item_index[2], indirect_parameters_index[2], batch_set_index[2], output_mesh_uniform_index;
} mi_batch_initial;
typedef struct mi_batch_totals {
    uint32_t work_item_len[2], indirect_parameters_len[2], batch_set_len[2], data_buffer_len, n_records, n_unbatchable;
} mi_batch_totals;

/* input_uniform_index = MI_NO_INPUT_INDEX: GetFullBatchData::get_binned_index / get_index_and_compare_data returned None. */
#define MI_NO_INPUT_INDEX 0xFFFFFFFFu
#define MI_NO_INDEX 0xFFFFFFFFu /* PhaseItemExtraIndex::None in an output */
int32_t mi_batch_upload_rows(mi_ctx* ctx, uint32_t first_row, uint32_t n, const uint32_t* batch_set, const uint32_t* bin_index,
                             const uint32_t* input_uniform_index);
/* n_sets <= 65536.  Replaces the previous tables. */
int32_t mi_batch_upload_sets(mi_ctx* ctx, uint32_t n_sets, const uint8_t* set_indexed, const uint32_t* bin_table_offset,
                             const uint32_t* bin_index_to_bin_metadata_index, const uint32_t* meta_offset,
                             const mi_bin_metadata* bin_metadata);
/* The part of the same binned phase the reference builds on the CPU -- unbatchable and batchable-but-not-multidrawable bins
 * (gpu_preprocessing.rs:2135-2357).  Per row its kind and, for the two CPU kinds, its bin; bins are numbered in the order
 * phase.unbatchable_meshes / phase.batchable_meshes iterate after sort_binned_render_phase (batching/mod.rs:199-209) and carry
 * their mesh class (key.0.indexed()).  Rows never uploaded are MI_BATCH_ROW_MULTIDRAWABLE (placed by mi_batch_upload_rows).
 * n_sets + n_batchable_bins + 2 * n_unbatchable_bins <= 65536. */
#define MI_BATCH_ROW_MULTIDRAWABLE 0u
#define MI_BATCH_ROW_BATCHABLE 1u
#define MI_BATCH_ROW_UNBATCHABLE 2u
#define MI_BATCH_ROW_NONE 3u
int32_t mi_batch_upload_row_bins(mi_ctx* ctx, uint32_t first_row, uint32_t n, const uint8_t* kind, const uint32_t* bin);
int32_t mi_batch_upload_bins(mi_ctx* ctx, uint32_t n_unbatchable_bins, const uint8_t* unbatchable_indexed, uint32_t n_batchable_bins,
                             const uint8_t* batchable_indexed);
/* After mi_cull / mi_propagate_and_cull of the same frame.  initial = NULL: all zero.  Enqueues only: three kernel launches.
 * One view's whole phase in the reference's order: unbatchables, batchables, multidrawable batch sets.
 * mi_batch_build = mi_batch_build_phase with flags 0.  MI_BATCH_NO_INDIRECT_DRAWING: the view has NoIndirectDrawing -- work items
 * carry output indices, no indirect parameters or batch sets are allocated, multidrawable rows are ignored (the reference puts
 * nothing into multidrawable_meshes without multidraw). */
#define MI_BATCH_NO_INDIRECT_DRAWING 0x1u
int32_t mi_batch_build(mi_ctx* ctx, uint32_t view, uint32_t class_bit, const mi_batch_initial* initial);
int32_t mi_batch_build_phase(mi_ctx* ctx, uint32_t view, uint32_t class_bit, const mi_batch_initial* initial, uint32_t flags);
int32_t mi_batch_download_totals(mi_ctx* ctx, mi_batch_totals* out);
/* Element [0, len) of one output array, indices absolute (the region below `initial` reads as zeros). */
#define MI_BATCH_WORK_ITEMS 0                   /* mi_preprocess_work_item */
#define MI_BATCH_INDIRECT_PARAMETERS_METADATA 1 /* mi_indirect_parameters_metadata */
#define MI_BATCH_SETS 2                         /* mi_indirect_batch_set */
#define MI_BATCH_RECORDS 3                      /* mi_batch_set_record; mesh_class ignored */
#define MI_BATCH_BIN_METADATA 4                 /* mi_bin_metadata with instance_count filled in; mesh_class ignored */
#define MI_BATCH_UNBATCHABLE_INDICES 5          /* mi_unbatchable_index in (bin, list) order; mesh_class ignored */
#define MI_BATCH_SORTED_BATCHES 6               /* mi_sorted_batch after mi_batch_sorted_build; mesh_class ignored */
int32_t mi_batch_download(mi_ctx* ctx, uint32_t what, uint32_t mesh_class, void* out, uint32_t capacity_elems, uint32_t* out_count);

/* Sorted phases: gpu_preprocessing::batch_and_prepare_sorted_render_phase (gpu_preprocessing.rs:1850-2061) and, with
 * MI_SORTED_NO_GPU_PREPROCESSING, the range merge of batching::batch_and_prepare_sorted_render_phase (batching/mod.rs:219-244) as
 * no_gpu_preprocessing.rs:76-103 drives it.  items[] (host) is phase.items in its sorted order.  Per item what
 * get_index_and_compare_data returned: input_index (MI_NO_INPUT_INDEX = None), and if MI_SORTED_ITEM_HAS_COMPARE_DATA the batch-set
 * key -- the whole BatchSetMeta (pipeline, draw function, dynamic offset, compare data; batching/mod.rs:42-72) interned to one id --
 * and the bin key (BatchCompareData).  Outputs: work items / metadata / batch sets per mesh class as above (appended at `initial`),
 * one mi_sorted_batch per batch set = what flush() writes on the set's first item (:1767-1794), totals (n_records = batch sets).
 * MI_SORTED_NO_GPU_PREPROCESSING: only the batches (instance ranges starting at initial->output_mesh_uniform_index). */
#define MI_SORTED_ITEM_INDEXED 1u
#define MI_SORTED_ITEM_HAS_COMPARE_DATA 2u
typedef struct mi_sorted_item { uint32_t input_index, batch_set_key, bin_key, flags; } mi_sorted_item;
typedef struct mi_sorted_batch {
    uint32_t first_item, instance_start, instance_end, indirect_parameters_start, indirect_parameters_end, indexed;
} mi_sorted_batch;
#define MI_SORTED_AUTOMATIC_BATCHING 0x1u   /* I::AUTOMATIC_BATCHING */
#define MI_SORTED_NO_INDIRECT_DRAWING 0x2u
#define MI_SORTED_NO_GPU_PREPROCESSING 0x4u
int32_t mi_batch_sorted_build(mi_ctx* ctx, uint32_t n_items, const mi_sorted_item* items, const mi_batch_initial* initial, uint32_t flags);

/* ======================================================================================= */
/* camera helpers (pure host code; what update_frusta computes, visibility/mod.rs:627-636)   */
/* ======================================================================================= */

/* PerspectiveProjection::get_clip_from_view without a custom near plane (projection.rs:339-343). */
int32_t mi_perspective_clip_from_view(float fov, float aspect_ratio, float near, float out_clip_from_view[16]);
/* CameraProjection::compute_frustum (projection.rs:72-80) for any clip_from_view. */
int32_t mi_compute_frustum(const float clip_from_view[16], const float camera_affine[12], float far,
                           float out_frustum[24]);

/* ======================================================================================= */
/* device interop, multi-GPU plumbing, timing                                                */
/* ======================================================================================= */

/* Row-range sharding over GPUs (one context per GPU/process): this context owns global rows
 * [first_global_row, first_global_row + n_rows).  Visibility bitmasks can be written straight into a
 * caller-provided device buffer shared by an RCCL all-gather: view v's words for this shard land at
 * device_ptr + (v * words_per_view + word_offset) * 8 bytes (any layout expressible that way, e.g.
 * [gpu][view][words] for an in-place all-gather); the shard's first global row must be a multiple of 64
 * and the caller guarantees the buffer covers every view's words.
 * Pass device_ptr = NULL to go back to the internal buffer. */
int32_t mi_bind_visibility_output(mi_ctx* ctx, void* device_ptr, uint64_t words_per_view, uint64_t word_offset);

/* The exchange itself, issued by the library: after every mi_cull / mi_propagate_and_cull the packed masks are
 * all-gathered IN PLACE across the ranks of an RCCL communicator.  The kernels write frame f's masks straight into
 * gathered buffer f % n_bufs; a library-owned host thread enqueues ncclAllGather on a library-owned communication
 * stream behind them (RCCL's enqueue costs tens of microseconds of CPU, kept off the caller's thread), so frame f's
 * collective overlaps the kernels of the following frames.  A buffer is reused n_bufs frames later; that dependency
 * is enforced by pacing the CALLER (it blocks until the all-gather of frame f - n_bufs has completed, normally
 * never), not by a cross-stream wait in the compute stream.  One FFI call per frame does everything.
 *   nccl_comm            ncclComm_t of this rank (created by the host: ncclCommInitRank); NULL switches it off
 *   fn_nccl_all_gather   address of ncclAllGather in the RCCL library the communicator belongs to
 *   device_bufs[n_bufs]  2..8 [world][n_views][words_per_view] uint64 buffers on this device (3+ recommended)
 *   word_offset          rank * n_views * words_per_view;  block_bytes = n_views * words_per_view * 8
 * The reference has no counterpart (single process); this replaces nothing and adds the only collective. */
/* How the library drives the exchange (set before mi_exchange_configure; default MI_EXCHANGE_SIMPLE):
 *   MI_EXCHANGE_SIMPLE     one communicator, one library-owned communication stream, plain event ordering: an event behind the
 *                          frame's kernels, wait + ncclAllGather + event on the communication stream, and the compute stream
 *                          waits for a buffer's previous all-gather (hipStreamWaitEvent) before its kernels overwrite it.  Since
 *                          round 5 the three communication-stream calls are made by a library-owned thread the frame call hands
 *                          the buffer's slot to (RCCL's enqueue path is 15 - 20 us of CPU per call: more than the rest of the
 *                          frame call, and at an 8-GPU shard size more than the frame's kernel); every rank's thread issues its
 *                          all-gathers in frame order.  MI_XCH_SYNC_ENQUEUE=1 in the environment keeps them on the calling thread.
 *                          What that thread changes for a caller: (1) ncclAllGather is ENQUEUED FROM ANOTHER THREAD, one per
 *                          context -- a process that drives several contexts in this mode issues its collectives from that many
 *                          unsynchronised threads (use MI_EXCHANGE_GROUPED, or MI_XCH_SYNC_ENQUEUE=1, when the order across
 *                          contexts must be the caller's); (2) a failing all-gather is reported LATE -- not by the frame call that
 *                          queued it but by the next frame call's exchange step or by mi_exchange_last / mi_exchange_download;
 *                          (3) the thread polls for up to 500 us after its last job before it sleeps.  mi_exchange_set_mode is
 *                          refused while the exchange is on, so the thread never outlives its mode.
 *   MI_EXCHANGE_PIPELINED  the latency-hiding variant built in round 1 against a 1-rank communicator: a library-owned host
 *                          thread enqueues the collectives, several communicators alternate by frame, the "masks complete"
 *                          signal is stored by the compaction kernel itself and awaited with hipStreamWaitValue32, buffer reuse
 *                          is paced on the host, the communication stream is picked by probing the hardware-queue mapping.
 *                          Its motivating numbers are 1-GPU measurements (DESIGN.md section 6 lists them as hypotheses); use it
 *                          once an N > 1 measurement says it pays. */
#define MI_EXCHANGE_SIMPLE 0u
#define MI_EXCHANGE_PIPELINED 1u
/*   MI_EXCHANGE_GROUPED    ONE process, ONE thread, several GPUs -- what a Bevy App is (one World, systems on one schedule): a context
 *                          per device, communicators from ncclCommInitAll.  One thread cannot issue the ranks' collectives one by one
 *                          (the first would wait for peers the same thread has not reached yet), so the frame calls leave the frame's
 *                          all-gather pending and mi_exchange_group_flush issues the pending all-gathers of all contexts between
 *                          ncclGroupStart and ncclGroupEnd.  Otherwise as MI_EXCHANGE_SIMPLE (event ordering, rotating buffers).
 *                          tests/cpp/multi_gpu_single_process.cpp drives a sharded frame this way over every GPU of the node. */
#define MI_EXCHANGE_GROUPED 2u
int32_t mi_exchange_set_mode(mi_ctx* ctx, uint32_t mode);
int32_t mi_exchange_configure(mi_ctx* ctx, void* nccl_comm, void* fn_nccl_all_gather, void* const* device_bufs,
                              uint32_t n_bufs, uint64_t words_per_view, uint64_t word_offset, uint64_t block_bytes,
                              uint32_t rank);
/* The same with up to 4 communicators over the same ranks, used round-robin by frame (frame f travels on communicator
 * f % n_comms, each on its own library stream): one communicator runs its collectives strictly one after the other,
 * and a ~125 KB all-gather over 8 GPUs is pure latency -- with two, the all-gathers of consecutive frames are in
 * flight together.  Every rank must pass its communicators in the same order.  n_comms = 0 switches the exchange off. */
int32_t mi_exchange_configure_multi(mi_ctx* ctx, void* const* nccl_comms, uint32_t n_comms, void* fn_nccl_all_gather,
                                    void* const* device_bufs, uint32_t n_bufs, uint64_t words_per_view, uint64_t word_offset,
                                    uint64_t block_bytes, uint32_t rank);
/* The same with the gathered buffers owned by the library (a host without a device allocator of its own: the Rust plugin, the C++ host
 * layer): n_bufs buffers of world x block_bytes bytes each are allocated on the context's device, zeroed, and freed when the exchange is
 * reconfigured (with or without buffers of the caller's, or switched off) or the context destroyed. */
int32_t mi_exchange_configure_owned(mi_ctx* ctx, void* const* nccl_comms, uint32_t n_comms, void* fn_nccl_all_gather, uint32_t n_bufs,
                                    uint32_t world, uint64_t words_per_view, uint64_t word_offset, uint64_t block_bytes, uint32_t rank);
/* MI_EXCHANGE_GROUPED: issues the pending all-gather of every listed context inside one ncclGroupStart / ncclGroupEnd pair
 * (fn_* = the addresses of those two functions in the RCCL library the communicators belong to).  Call it after the frame calls of
 * all contexts, from the thread that made them. */
int32_t mi_exchange_group_flush(mi_ctx* const* contexts, uint32_t n, void* fn_nccl_group_start, void* fn_nccl_group_end);
/* The gathered buffer of the most recent frame (optionally after waiting for its collective). */
int32_t mi_exchange_last(mi_ctx* ctx, void** out_device_buf, int32_t wait);

/* The same buffer on the HOST: waits for the most recent frame's all-gather and copies the first `bytes` bytes of its gathered buffer
 * ([rank][view][word], 64-bit words; bytes = world x block_bytes for all of it) into out_host -- ONE device-to-host copy gives a
 * single-process host (a Bevy plugin driving several GPUs) every shard's ViewVisibility masks, whichever context it asks.
 * Buffers of mi_exchange_configure_owned: bytes beyond world x block_bytes is MI_ERR_INVALID_ARG.  Buffers the caller bound
 * (mi_exchange_configure / _multi): the library does not know their size -- bytes must not exceed what the caller allocated. */
int32_t mi_exchange_download(mi_ctx* ctx, void* out_host, uint64_t bytes);

/* Raw device pointers of library-owned columns, for zero-copy READERS on the same device (write through the upload entry points only:
 * the library keeps results that travel ahead of a frame and does not see a write through a raw pointer)
 * (e.g. the render world's mesh-uniform builder).  Valid until the next mi_columns_resize. */
#define MI_BUF_GLOBAL_TRANSFORM 0
#define MI_BUF_VISIBILITY_BITMASK 1
#define MI_BUF_VIEW_VISIBILITY 2
#define MI_BUF_VISIBLE_ROWS 3
#define MI_BUF_CLUSTER_OFFSETS_AND_COUNTS 4
#define MI_BUF_CLUSTER_INDEX_LIST 5
#define MI_BUF_BATCH_WORK_ITEMS_NON_INDEXED 6
#define MI_BUF_BATCH_WORK_ITEMS_INDEXED 7
#define MI_BUF_BATCH_METADATA_NON_INDEXED 8
#define MI_BUF_BATCH_METADATA_INDEXED 9
#define MI_BUF_BATCH_SETS_NON_INDEXED 10
#define MI_BUF_BATCH_SETS_INDEXED 11
int32_t mi_device_buffer(mi_ctx* ctx, uint32_t which, void** out_device_ptr, uint64_t* out_bytes);

/* Bench / test instrumentation (HIP-event timers, the per-kernel profile, planner and trace hooks) is declared in
 * bevy_mi355x_debug.h: exported by the same library, not part of the drop-in boundary. */

#ifdef __cplusplus
}
#endif
#endif /* BEVY_MI355X_H */
