//! `bevy_mi355x` -- the Bevy side of the drop-in boundary of `libbevy_mi355x.so` (include/bevy_mi355x.h).
//!
//! [`Mi355xRenderPrepPlugin`] takes the three per-frame render-prep systems out of the `PostUpdate` schedule and puts one system
//! with the same inputs, outputs, system set and ordering in the place of each:
//!
//! | stock system (reference file:line)                                                        | replacement                         |
//! |--------------------------------------------------------------------------------------------|-------------------------------------|
//! | `mark_dirty_trees`, `propagate_parent_transforms`, `sync_simple_transforms`                 | [`mi_propagate_transforms`]         |
//! |   (crates/bevy_transform/src/systems.rs:42-79, 111-306, 506-748)                            |                                     |
//! | `check_visibility_cpu_culling` (crates/bevy_camera/src/visibility/mod.rs:748-880)           | [`mi_check_visibility`]             |
//! | `assign_objects_to_clusters` (crates/bevy_light/src/cluster/assign.rs:137-813)              | [`mi_assign_objects_to_clusters`]   |
//! | `check_dir_light_mesh_visibility`, `check_point_light_mesh_visibility`                      | [`mi_check_light_mesh_visibility`]  |
//! |   (crates/bevy_light/src/lib.rs:342-515, 517-757)                                           |                                     |
//!
//! With `fused: true` (the default) the three collapse into ONE device round trip per frame: [`mi_fused_frame`], placed in
//! `TransformSystems::Propagate`, uploads the changed `Transform`s, issues `mi_propagate_and_cull_views(MI_CULL_CHANGED_ROWS |
//! MI_CULL_WITH_CLUSTERS | MI_CULL_END_FRAME)` -- propagate, reset, cull of every active camera, gather of the visible lights and
//! cluster assignment: one kernel launch for a flat scene -- and reads everything back with ONE `mi_download_frame_results`
//! (delivered in place: one packing launch, one device wait, no copy).  It writes `GlobalTransform` at once and parks the
//! rest in [`Mi355xFrame`]; [`mi_apply_visibility`] (in `VisibilitySystems::CheckVisibility`, between the stock
//! `reset_view_visibility` and `mark_newly_hidden_entities_invisible`, which stay registered) and [`mi_apply_clusters`] (in front
//! of `SimulationLightSystems::AssignLightsToClusters`) make the remaining ECS writes where the stock systems make them, without
//! touching the device.  The cameras' frusta are computed by the fused system itself, from the `GlobalTransform` each camera is
//! ABOUT to get (`update_frusta` runs behind `TransformSystems::Propagate`; same `compute_frustum`, same inputs, same bits);
//! should an input of the cull turn out to have changed after the submit (an `Aabb` inserted by `calculate_bounds`, an
//! `InheritedVisibility` flipped by `visibility_propagate_system`, a `Frustum` that differs from the predicted one), the parked
//! results are dropped and the three-system form -- registered behind the apply systems, gated on [`Mi355xFrame::valid`] -- runs
//! for that frame.  `bevy_amd/host/bevy_mi355x_host.hpp` (`Mi355xPlugin::frame`) is the same design in C++, compiled and run
//! against the reference's system tests in both forms (`tests/cpp/host_systems_test.cpp`).
//!
//! The ECS stays the owner of every component.  The library owns device-resident COLUMNS, one row per entity, and the systems
//! below move only what changed: rows whose `Transform` change tick is newer than the system's last run go up
//! (`mi_upload_transforms_indexed`), rows whose `GlobalTransform` the device changed come down
//! (`mi_download_changed_global_transforms`) and are written through `Mut`, so change ticks move exactly where the stock systems
//! move them.  Bulk reads go through `Query::contiguous_iter` (crates/bevy_ecs/src/system/query.rs:1509-1560): one slice per
//! table, no per-entity fetch.
//!
//! # The CPU-fallback rule
//!
//! Every `mi_*` call returns an `int32` status.  A system below
//!   1. makes all of its `mi_*` calls BEFORE it writes anything to the ECS;
//!   2. on `MI_ERR_MALFORMED_HIERARCHY` panics with the library's message -- the reference panics on the same input
//!      (crates/bevy_transform/src/systems.rs:715);
//!   3. on ANY other negative status logs `mi_last_error_string` once, sets its bit in [`CpuFallback`] and returns without having
//!      touched the ECS.  The stock system it replaced is still in the schedule, ordered right behind it and gated on that bit
//!      (`run_if`), so the SAME frame is computed by the reference's own code; the bit stays set, i.e. the GPU path is off for
//!      the rest of the run (a device that failed once is not retried every frame).
//! `assign_objects_to_clusters` is `pub(crate)`, so it cannot be re-added; instead its system set
//! (`SimulationLightSystems::AssignLightsToClusters`, which holds only that system, crates/bevy_light/src/lib.rs:187-191) is
//! gated with the same condition and the replacement runs just before the set.
//!
//! There is no Rust toolchain in the image this repository is built in: this crate is written against the reference checkout
//! (Bevy 0.20.0-dev) but has not been compiled.  `tests/test_rust_ffi.py` checks `ffi.rs` against the C header item by item and
//! checks that every `ffi::mi_*` call below passes the declared number of arguments; `tools/check_rust_names.py` resolves every
//! import, type identifier, method name and trait scope of this file against the reference checkout (and counts system parameters);
//! `bevy_amd/host/bevy_mi355x_host.hpp` is the same host layer in C++, which IS compiled and run against the reference's system
//! tests on the GPU.

pub mod ffi;
pub mod sharded;

use core::ffi::CStr;
use core::ptr;

use bevy_app::{App, Plugin, PostStartup, PostUpdate};
use bevy_camera::{
    primitives::{Aabb, CascadesFrusta, CubemapFrusta, Frustum, Sphere},
    visibility::{
        check_visibility_cpu_culling, CascadesVisibleEntities, CubemapVisibleEntities, InheritedVisibility, NoCpuCulling,
        NoFrustumCulling, RenderLayers, SetViewVisibility, ViewVisibility, VisibilityClass, VisibilityRange, VisibilitySystems,
        VisibleEntities, VisibleEntityRanges, VisibleMeshEntities,
    },
    Camera, CameraUpdateSystems, RenderTarget, ShadowLodOrigin,
};
use bevy_ecs::{
    change_detection::Tick,
    entity::{Entity, EntityHashMap, EntityHashSet},
    prelude::*,
    schedule::{IntoScheduleConfigs, ScheduleCleanupPolicy::RemoveSystemsOnly, ScheduleLabel},
    system::SystemChangeTick,
};
use bevy_light::{
    check_dir_light_mesh_visibility, check_point_light_mesh_visibility,
    cluster::{
        ClusterConfig, ClusterFarZMode, ClusterVisibilityClass, ClusterableObjects, Clusters, GlobalClusterSettings, ObjectsInClusterCpu,
    },
    get_shadow_lod_origin, ClusteredDecal, DirectionalLight, EnvironmentMapLight, LightProbe, NotShadowCaster, PointLight, RectLight,
    SimulationLightSystems, SpotLight, VolumetricLight,
};
use bevy_log::error;
#[cfg(feature = "trace")]
use bevy_log::info_span;
use bevy_math::{Affine3A, UVec2, UVec3};
use bevy_mesh::Mesh3d;
use bevy_platform::collections::HashMap;
use bevy_transform::{
    components::{GlobalTransform, Transform},
    systems::{mark_dirty_trees, propagate_parent_transforms, sync_simple_transforms},
    TransformSystems,
};
use core::any::TypeId;

/// Which of the three replaced systems have handed their work back to the stock CPU systems (see the module docs).
#[derive(Resource, Default, Clone, Copy, PartialEq, Eq, Debug)]
pub struct CpuFallback {
    pub transforms: bool,
    pub visibility: bool,
    pub clusters: bool,
    pub light_visibility: bool,
}

fn light_visibility_fell_back(f: Res<CpuFallback>) -> bool {
    f.light_visibility
}
fn transforms_fell_back(f: Res<CpuFallback>) -> bool {
    f.transforms
}
fn visibility_fell_back(f: Res<CpuFallback>) -> bool {
    f.visibility
}
fn clusters_fell_back(f: Res<CpuFallback>) -> bool {
    f.clusters
}

/// The library context and the Entity <-> row table.
#[derive(Resource)]
pub struct Mi355x {
    ctx: *mut ffi::MiCtx,
    /// row -> entity; rows `[0, n_tree)` are the hierarchy in the level order `mi_hierarchy_sort` produced, rows above are free
    /// of any hierarchy (roots without children) in table order.
    row_entity: Vec<Entity>,
    entity_row: EntityHashMap<u32>,
    /// `TypeId` of a visibility class -> bit of the `class_mask` column (at most 32 classes, like the column).
    class_bits: HashMap<TypeId, u32>,
    /// Per-view cluster feedback (`Clusters::last_frame_*`, crates/bevy_light/src/cluster/mod.rs:153-163).
    cluster_history: EntityHashMap<ffi::MiClusterHistory>,
    /// Storage of the x / y / z cluster planes `mi_cluster_view_build` fills for the fused frame's view.
    plane_storage: Vec<f32>,
    /// ... and of the clusters' bounding spheres (only filled when there are spot lights: their cone test reads them).
    sphere_storage: Vec<f32>,
    /// Rows follow the tables (a flat world: the level order of a forest of single nodes is the order they were listed in): a walk
    /// over the tables visits rows 0, 1, 2 ... -- what lets an all-dirty frame fill dense upload windows in one pass.
    rows_in_table_order: bool,
    /// The last steady-state upload of the fused frame carried every row through dense windows: its frame call is the all-rows one.
    every_row_moved: bool,
    /// Which components of a changed `Transform` the steady-state upload windows carry: 0 = all three (default), else a combination of
    /// `ffi::MI_UPLOAD_TRANSLATION | MI_UPLOAD_ROTATION | MI_UPLOAD_SCALE`.  `Changed<Transform>` does not say WHAT changed, the app
    /// does: a scene whose systems only ever turn things (`many_cubes --rotate-cubes`, examples/stress_tests/many_cubes.rs:641-648) sets
    /// `MI_UPLOAD_ROTATION` and sends 16 bytes per moved row instead of 40.  A component that is not carried keeps the value of the
    /// last full upload (every structural rebuild sends all three) -- setting this while a system writes the others is the app's bug.
    pub upload_components: u32,
    /// The last structural rebuild found a hierarchy the stock systems should keep (`mi_hierarchy_advice_for`: no level wider than a
    /// wave): the frames fall back (see [`CpuFallback`]) until the structure changes, when the question is asked again.
    hierarchy_kept_on_host: bool,
    /// Whether the device's VisibilityRange column was last staged with a `VisibleEntityRanges` resource present (`None`: never staged).
    /// `Option<Res<VisibleEntityRanges>>` being `None` means no range hides anything (visibility/mod.rs:814-816): no column then.
    ranges_resource: Option<bool>,
    scratch: Scratch,
}

/// Host staging reused from frame to frame (the reference keeps `Local<Vec<..>>`s for the same reason, assign.rs:179-180).
#[derive(Default)]
struct Scratch {
    rows: Vec<u32>,
    translation: Vec<f32>,
    rotation: Vec<f32>,
    scale: Vec<f32>,
    global12: Vec<f32>,
    parent: Vec<u32>,
    new_to_old: Vec<u32>,
    parent_idx: Vec<u32>,
    level_offsets: Vec<u32>,
    keys: Vec<u64>,
    aabb_center: Vec<f32>,
    aabb_half: Vec<f32>,
    flags: Vec<u8>,
    layers: Vec<u32>,
    layers_hi: Vec<u32>, // RenderLayers 32..63 (uploaded only once some entity uses one: mi_upload_render_layers_hi)
    any_layers_hi: bool,
    classes: Vec<u32>,
    view_visibility: Vec<u8>,
    vv_changed: Vec<u32>,
    views: Vec<ffi::MiView>,
    view_entities: Vec<Entity>,
    visible_keys: Vec<u64>,
    visible_rows: Vec<u32>,
    obj_pos_range: Vec<f32>,
    obj_type: Vec<u8>,
    obj_layers: Vec<u32>,
    obj_layers_hi: Vec<u32>, // RenderLayers 32..63 of the fused frame's cluster objects (mi_cluster_upload_object_layers_hi)
    obj_spot_dir: Vec<f32>,
    obj_spot_sin_cos: Vec<f32>,
    obj_shadows: Vec<u8>,
    obj_volumetric: Vec<u8>,
    obj_entity: Vec<Entity>,
    obj_order: Vec<u32>,
    cl_offsets: Vec<u32>,
    cl_counts: Vec<u32>,
    cl_indices: Vec<u32>,
    ranges: Vec<f32>, // VisibilityRange (start_margin.start, end_margin.end) per row (mi_upload_visibility_ranges)
    shadow_views: Vec<ffi::MiView>,
    light_masks: Vec<u32>, // mi_check_light_mesh_visibility: [view][ceil(n / 32)]
    light_any: Vec<u32>,
}

// SAFETY: the context is only used through `ResMut<Mi355x>`, i.e. by one system at a time, which is the library's contract
// ("one call at a time per context", include/bevy_mi355x.h).
unsafe impl Send for Mi355x {}
unsafe impl Sync for Mi355x {}

impl Mi355x {
    pub fn new(device: i32) -> Result<Self, String> {
        // SAFETY: plain FFI; `out_ctx` points at a live local.
        unsafe {
            if ffi::mi_abi_version() != ffi::MI_ABI_VERSION {
                return Err("libbevy_mi355x: ABI version mismatch".into());
            }
            let mut ctx = ptr::null_mut();
            let status = ffi::mi_ctx_create(device, ptr::null_mut(), &mut ctx);
            if status != ffi::MI_OK {
                return Err(format!("mi_ctx_create({device}) failed with status {status}"));
            }
            Ok(Self {
                ctx,
                row_entity: Vec::new(),
                entity_row: EntityHashMap::default(),
                class_bits: HashMap::default(),
                cluster_history: EntityHashMap::default(),
                plane_storage: Vec::new(),
                rows_in_table_order: false,
                every_row_moved: false,
                upload_components: 0,
                hierarchy_kept_on_host: false,
            ranges_resource: None,
                sphere_storage: Vec::new(),
                scratch: Scratch::default(),
            })
        }
    }

}

fn message(ctx: *mut ffi::MiCtx) -> String {
    // SAFETY: the library returns a NUL-terminated string that lives until the next call on this context.
    unsafe { CStr::from_ptr(ffi::mi_last_error_string(ctx)).to_string_lossy().into_owned() }
}

/// The status rule of the module docs: `Ok` for `MI_OK`, panic for a malformed hierarchy, `Err` for everything else.
/// (A free function over the raw handle, so that callers can hold `&mut` borrows of the resource's staging vectors.)
pub(crate) fn check(ctx: *mut ffi::MiCtx, what: &str, status: i32) -> Result<(), ()> {
    if status == ffi::MI_OK {
        return Ok(());
    }
    if status == ffi::MI_ERR_MALFORMED_HIERARCHY {
        panic!("{}", message(ctx));
    }
    error!("bevy_mi355x: {what} failed with status {status}: {}; falling back to the CPU system", message(ctx));
    Err(())
}

impl Drop for Mi355x {
    fn drop(&mut self) {
        // SAFETY: `ctx` came from `mi_ctx_create` and is destroyed exactly once.
        unsafe {
            ffi::mi_synchronize(self.ctx);
            ffi::mi_ctx_destroy(self.ctx);
        }
    }
}

/// Replaces transform propagation, visibility checking and light-cluster assignment with the MI355X library.
pub struct Mi355xRenderPrepPlugin {
    /// HIP device ordinal.
    pub device: i32,
    /// One device round trip per frame ([`mi_fused_frame`]); `false` = the three systems, a round trip each.
    pub fused: bool,
    /// More than one entry: the multi-GPU form ([`sharded::mi_sharded_frame`]) -- a context per listed device, rows sharded by range,
    /// the packed ViewVisibility masks all-gathered by RCCL (north_star's partition; flat Worlds).  Empty or one entry: `device`.
    pub devices: Vec<i32>,
}

impl Default for Mi355xRenderPrepPlugin {
    fn default() -> Self {
        Self { device: 0, fused: true, devices: Vec::new() }
    }
}

impl Mi355xRenderPrepPlugin {
    /// The multi-GPU form: `mi_sharded_frame` in `TransformSystems::Propagate` (propagate + cull of every shard, the masks gathered),
    /// `mi_apply_visibility` in `VisibilitySystems::CheckVisibility`; the stock systems behind both for the frames they hand back.  Light
    /// clusters and shadow views stay with the stock systems in this form.
    fn build_sharded(&self, app: &mut App) {
        let shards = match sharded::Mi355xShards::new(&self.devices) {
            Ok(shards) => shards,
            Err(message) => {
                error!("bevy_mi355x: {message}; the stock CPU systems stay in place");
                return;
            }
        };
        app.insert_resource(shards).init_resource::<CpuFallback>().init_resource::<Mi355xFrame>();
        for schedule in [PostStartup.intern(), PostUpdate.intern()] {
            take_out(app, schedule, mark_dirty_trees);
            take_out(app, schedule, propagate_parent_transforms);
            take_out(app, schedule, sync_simple_transforms);
            app.add_systems(
                schedule,
                (
                    sharded::mi_sharded_frame,
                    (mark_dirty_trees, propagate_parent_transforms, sync_simple_transforms).chain().run_if(transforms_fell_back),
                )
                    .chain()
                    .in_set(TransformSystems::Propagate),
            );
        }
        take_out(app, PostUpdate, check_visibility_cpu_culling);
        app.add_systems(
            PostUpdate,
            (
                mi_apply_visibility,
                // (no single-context cull to fall back on: a frame whose parked lists are stale, or that fell back, is the stock system's)
                check_visibility_cpu_culling.run_if(frame_results_missing),
            )
                .chain()
                .in_set(VisibilitySystems::CheckVisibility),
        );
    }
}

impl Plugin for Mi355xRenderPrepPlugin {
    fn build(&self, app: &mut App) {
        if self.devices.len() > 1 {
            self.build_sharded(app);
            return;
        }
        let mi = match Mi355x::new(self.devices.first().copied().unwrap_or(self.device)) {
            Ok(mi) => mi,
            Err(message) => {
                // No device, no library: leave the stock systems where they are.
                error!("bevy_mi355x: {message}; the stock CPU systems stay in place");
                return;
            }
        };
        app.insert_resource(mi).init_resource::<CpuFallback>().init_resource::<Mi355xFrame>();
        let fused = self.fused;

        // --- transforms: TransformPlugin registers the same three systems in PostStartup and PostUpdate
        //     (crates/bevy_transform/src/plugins.rs:22-48).  Every system fn is its own implicit set
        //     (crates/bevy_app/src/app.rs:330-360), which is how they are taken out again.
        for schedule in [PostStartup.intern(), PostUpdate.intern()] {
            take_out(app, schedule, mark_dirty_trees);
            take_out(app, schedule, propagate_parent_transforms);
            take_out(app, schedule, sync_simple_transforms);
            // PostStartup has no cameras to cull for yet: the propagate-only system either way
            let fused_here = fused && schedule == PostUpdate.intern();
            app.add_systems(
                schedule,
                (
                    mi_fused_frame.run_if(move || fused_here),
                    mi_propagate_transforms.run_if(move || !fused_here),
                    // the stock trio in its stock order, only when the replacement gave up (this frame or earlier)
                    (mark_dirty_trees, propagate_parent_transforms, sync_simple_transforms)
                        .chain()
                        .run_if(transforms_fell_back),
                )
                    .chain()
                    .in_set(TransformSystems::Propagate),
            );
        }

        // --- visibility: `reset_view_visibility` before and `mark_newly_hidden_entities_invisible` after stay registered (they
        //     are private, visibility/mod.rs:529-533); the middle step is replaced.
        take_out(app, PostUpdate, check_visibility_cpu_culling);
        app.add_systems(
            PostUpdate,
            (
                // fused: the parked lists, no device call; if they turn out stale, the round trip of its own right behind
                mi_apply_visibility.run_if(move || fused),
                mi_check_visibility.run_if(frame_results_missing),
                check_visibility_cpu_culling.run_if(visibility_fell_back),
            )
                .chain()
                .in_set(VisibilitySystems::CheckVisibility),
        );

        // --- shadow views: check_dir_light_mesh_visibility and check_point_light_mesh_visibility are both `pub` (crates/bevy_light/src/
        //     lib.rs:342, 517); one system over the device columns takes their place in SimulationLightSystems::CheckLightVisibility with
        //     the stock ordering (lib.rs:217-230), the stock pair right behind it for the frames it gives up on.
        take_out(app, PostUpdate, check_dir_light_mesh_visibility);
        take_out(app, PostUpdate, check_point_light_mesh_visibility);
        app.add_systems(
            PostUpdate,
            (
                mi_check_light_mesh_visibility,
                (check_dir_light_mesh_visibility, check_point_light_mesh_visibility).run_if(light_visibility_fell_back),
            )
                .chain()
                .in_set(SimulationLightSystems::CheckLightVisibility)
                .after(VisibilitySystems::CalculateBounds)
                .after(TransformSystems::Propagate)
                .after(SimulationLightSystems::UpdateLightFrusta)
                // lights mark the shadow casters they see visible before the newly hidden ones are marked (lib.rs:224-229)
                .after(VisibilitySystems::CheckVisibility)
                .before(VisibilitySystems::MarkNewlyHiddenEntitiesInvisible),
        );

        // --- clusters: gate the stock set, run the replacement right in front of it with the stock ordering constraints
        //     (crates/bevy_light/src/lib.rs:187-191).
        app.configure_sets(PostUpdate, SimulationLightSystems::AssignLightsToClusters.run_if(clusters_fell_back));
        app.add_systems(
            PostUpdate,
            (mi_apply_clusters.run_if(move || fused), mi_assign_objects_to_clusters.run_if(frame_clusters_missing))
                .chain()
                .before(SimulationLightSystems::AssignLightsToClusters)
                .after(TransformSystems::Propagate)
                .after(VisibilitySystems::CheckVisibility)
                .after(CameraUpdateSystems),
        );
    }
}

/// Takes a stock system out of its schedule, keeping the ordering edges that ran through it (`RemoveSystemsOnly`,
/// crates/bevy_ecs/src/schedule/schedule.rs:1715-1735).  A system that is not there -- the stock plugin was not added, or was added
/// after this one -- is said so once: the replacement would otherwise run beside it.
fn take_out<M>(app: &mut App, schedule: impl ScheduleLabel, system: impl IntoSystemSet<M>) {
    if let Err(e) = app.remove_systems_in_set(schedule, system, RemoveSystemsOnly) {
        error!("bevy_mi355x: a stock system could not be taken out of its schedule ({e}); add Mi355xRenderPrepPlugin after DefaultPlugins");
    }
}

/// The rows of the visibility columns: `visible_aabb_query` of `check_visibility_cpu_culling` (visibility/mod.rs:757-772) plus what the
/// shadow-view systems' query filters on (`With<Mesh3d>, Without<NotShadowCaster>, Without<DirectionalLight>`,
/// crates/bevy_light/src/lib.rs:355-372 -> `MI_FLAG_SHADOW_CASTER`).
pub(crate) type RowsQuery<'w, 's> = Query<
    'w,
    's,
    (
        Entity,
        &'static InheritedVisibility,
        Option<&'static VisibilityClass>,
        Option<&'static RenderLayers>,
        Option<&'static Aabb>,
        Option<&'static Sphere>,
        Option<&'static PointLight>,
        Option<&'static SpotLight>,
        Has<NoFrustumCulling>,
        Option<&'static VisibilityRange>,
        Has<Mesh3d>,
        Has<NotShadowCaster>,
        Has<DirectionalLight>,
    ),
    Without<NoCpuCulling>,
>;
/// "Some input of the rarely-changing columns was written" (what `stage_bounds` re-stages for).
pub(crate) type BoundsChanged<'w, 's> = Query<
    'w,
    's,
    (),
    Or<(
        Changed<Aabb>,
        Changed<Sphere>,
        Changed<InheritedVisibility>,
        Changed<RenderLayers>,
        Changed<VisibilityClass>,
        Added<NoFrustumCulling>,
        Changed<VisibilityRange>,
        Added<Mesh3d>,
        Added<NotShadowCaster>,
    )>,
>;
/// The view query of `check_visibility_ranges` (visibility/range.rs:228): its first 32 entities get an index in `VisibleEntityRanges`.
pub(crate) type RangeViews<'w, 's> = Query<'w, 's, (Entity, &'static GlobalTransform), Or<(With<Camera>, With<ShadowLodOrigin>)>>;

/// The entities `check_visibility_ranges` gives an index (range.rs:238-243: `.take(32)` of its view query, cameras and
/// `ShadowLodOrigin`s, active or not) with the translation it measures distances from (`view_transform.translation_vec3a()`).  A view
/// that is not in here has no index: `entity_is_in_range_of_view` is false for every ranged entity (range.rs:209-217) -- the device does
/// the same for a view without `MI_VIEW_FLAG_RANGES`.
pub(crate) fn range_view_table(range_views: &RangeViews) -> EntityHashMap<[f32; 3]> {
    let mut table = EntityHashMap::default();
    for (entity, global) in range_views.iter().take(32) {
        table.insert(entity, global.translation().to_array());
    }
    table
}

#[inline]
fn changed_since(tick: Tick, ticks: &SystemChangeTick) -> bool {
    tick.is_newer_than(ticks.last_run(), ticks.this_run())
}

/// `mark_dirty_trees` + `propagate_parent_transforms` + `sync_simple_transforms` in one device pass.
///
/// Same observable behaviour: `GlobalTransform` of every entity equals `parent.GlobalTransform * Transform` (or `Transform` for
/// roots), and a `GlobalTransform` is written through `Mut` -- i.e. its change tick moves -- only where the stock systems would
/// have written it: rows whose own `Transform`, or that of an ancestor, changed since the last run
/// (`TransformTreeChanged` in the reference, systems.rs:42-79; the device keeps the same dirty bit per row).
pub fn mi_propagate_transforms(
    mut mi: ResMut<Mi355x>,
    mut fallback: ResMut<CpuFallback>,
    ticks: SystemChangeTick,
    structure_changed: Query<(), Or<(Added<Transform>, Changed<ChildOf>)>>,
    mut orphaned: RemovedComponents<ChildOf>,
    mut despawned: RemovedComponents<Transform>,
    transforms: Query<(Entity, Ref<Transform>, Option<&ChildOf>)>,
    mut globals: Query<&mut GlobalTransform>,
) {
    // (as the reference does inside its own systems, crates/bevy_transform/src/systems.rs:169-283, :592: a span per stage under `trace`)
    #[cfg(feature = "trace")]
    let _span = info_span!("mi_propagate_transforms").entered();
    let mi = &mut *mi;
    let rebuild = !structure_changed.is_empty() || orphaned.read().count() != 0 || despawned.read().count() != 0;
    if fallback.transforms {
        if !(mi.hierarchy_kept_on_host && rebuild) {
            return;
        }
        // not a device error: the stock systems kept a hierarchy as narrow as a chain, and its structure has just changed -- ask again
        *fallback = CpuFallback::default();
    }
    let result = upload_and_propagate(mi, rebuild, &ticks, &transforms, &globals.as_readonly(), None);
    let Ok(count) = result else {
        fallback.transforms = true; // nothing was written to the ECS; the stock trio runs next, in this same frame
        return;
    };
    write_back_global_transforms(mi, count, &mut globals);
}

/// Rows in, propagate, changed rows out -- shared by [`mi_propagate_transforms`] and the rebuild frames of [`mi_fused_frame`].
/// `frame`: instead of `mi_propagate` + the sparse download, the caller's own frame call follows (steady-state fused frame):
/// only the upload happens here and 0 is returned.
fn upload_and_propagate(
    mi: &mut Mi355x,
    rebuild: bool,
    ticks: &SystemChangeTick,
    transforms: &Query<(Entity, Ref<Transform>, Option<&ChildOf>)>,
    globals: &Query<&GlobalTransform>,
    frame: Option<()>,
) -> Result<u32, ()> {
    #[cfg(feature = "trace")]
    let _span = info_span!("mi355x upload + propagate").entered();
    let ctx = mi.ctx;
    mi.every_row_moved = false;
    let tables = transforms.contiguous_iter().expect("Transform and ChildOf are table components");
    {
        let s = &mut mi.scratch;
        if rebuild {
            // (1) provisional rows in table order, (2) parent as a provisional row, (3) level order from the library.
            mi.row_entity.clear();
            mi.entity_row.clear();
            s.translation.clear();
            s.rotation.clear();
            s.scale.clear();
            let mut parents_of: Vec<Option<Entity>> = Vec::new();
            for (entities, table_transforms, child_of) in tables {
                for (i, (entity, t)) in entities.iter().zip(table_transforms.iter()).enumerate() {
                    mi.entity_row.insert(*entity, mi.row_entity.len() as u32);
                    mi.row_entity.push(*entity);
                    parents_of.push(child_of.map(|c| c[i].parent()));
                    s.translation.extend_from_slice(&t.translation.to_array());
                    s.rotation.extend_from_slice(&t.rotation.to_array());
                    s.scale.extend_from_slice(&t.scale.to_array());
                }
            }
            let n = mi.row_entity.len() as u32;
            s.parent.clear();
            // A `ChildOf` that points at an entity without `Transform` makes the child a root, as in the reference
            // (systems.rs:528-531 only descends through entities that match the transform query).
            s.parent.extend(parents_of.iter().map(|p| p.and_then(|p| mi.entity_row.get(&p).copied()).unwrap_or(ffi::MI_NO_PARENT)));
            s.new_to_old.resize(n as usize, 0);
            s.parent_idx.resize(n as usize, 0);
            s.level_offsets.resize(n as usize + 2, 0);
            let mut n_levels = 0u32;
            // SAFETY: every pointer is a live Vec of at least the length the header asks for.
            check(ctx, "mi_hierarchy_sort", unsafe {
                ffi::mi_hierarchy_sort(
                    n,
                    s.parent.as_ptr(),
                    s.new_to_old.as_mut_ptr(),
                    s.parent_idx.as_mut_ptr(),
                    s.level_offsets.as_mut_ptr(),
                    s.level_offsets.len() as u32,
                    &mut n_levels,
                )
            })?;
            // A hierarchy no wider than a wave per level (transform_hierarchy.rs's `chain`, a rope, one rig) is one wave's chain of
            // dependent level steps on the device and 20 ns a node on a CPU core: the library says so (mi_hierarchy_advice_for), and the
            // stock systems keep such a World -- this frame and every frame until its structure changes again.  Not an error: no log.
            {
                // SAFETY: level_offsets holds n_levels + 1 entries; the out struct is plain data.
                let mut advice: ffi::MiHierarchyAdvice = unsafe { core::mem::zeroed() };
                check(ctx, "mi_hierarchy_advice_for", unsafe { ffi::mi_hierarchy_advice_for(n_levels, s.level_offsets.as_ptr(), &mut advice) })?;
                mi.hierarchy_kept_on_host = advice.keep_on_host != 0;
                if mi.hierarchy_kept_on_host {
                    return Err(());
                }
            }
            // rows := level order
            let old_entities = core::mem::take(&mut mi.row_entity);
            let gather3 = |src: &Vec<f32>, w: usize| -> Vec<f32> {
                s.new_to_old.iter().flat_map(|&o| src[o as usize * w..(o as usize + 1) * w].iter().copied()).collect()
            };
            let (t, r, sc) = (gather3(&s.translation, 3), gather3(&s.rotation, 4), gather3(&s.scale, 3));
            mi.row_entity = s.new_to_old.iter().map(|&o| old_entities[o as usize]).collect();
            mi.rows_in_table_order = n_levels <= 1 && s.new_to_old.iter().enumerate().all(|(row, &o)| o as usize == row);
            for (row, e) in mi.row_entity.iter().enumerate() {
                mi.entity_row.insert(*e, row as u32);
            }
            s.keys.clear();
            s.keys.extend(mi.row_entity.iter().map(|e| e.to_bits()));
            // Every row has a new occupant: the device's GlobalTransform column is re-seeded with the ECS values in the new row
            // order.  set_if_neq (systems.rs:719) compares against it -- against whatever entity used to sit in a row, a freshly
            // spawned child whose GlobalTransform happens to equal the previous occupant's would never be listed as changed and
            // would keep GlobalTransform::IDENTITY; other rows would be listed (and ticked) spuriously.
            s.global12.clear();
            for e in &mi.row_entity {
                let g = globals.get(*e).map(|g| g.affine()).unwrap_or(Affine3A::IDENTITY);
                s.global12.extend_from_slice(&g.to_cols_array());
            }
            // SAFETY: as above; the columns hold `n` rows after `mi_columns_resize`.
            unsafe {
                check(ctx, "mi_columns_resize", ffi::mi_columns_resize(ctx, n))?;
                check(ctx, "mi_upload_transforms", ffi::mi_upload_transforms(ctx, 0, n, t.as_ptr(), r.as_ptr(), sc.as_ptr()))?;
                check(ctx, "mi_upload_global_transforms", ffi::mi_upload_global_transforms(ctx, 0, n, s.global12.as_ptr()))?;
                check(ctx, "mi_upload_entity_keys", ffi::mi_upload_entity_keys(ctx, 0, n, s.keys.as_ptr()))?;
                check(ctx, 
                    "mi_upload_hierarchy",
                    ffi::mi_upload_hierarchy(ctx, n, s.parent_idx.as_ptr(), s.level_offsets.as_ptr(), n_levels),
                )?;
                check(ctx, "mi_propagate", ffi::mi_propagate(ctx, ffi::MI_PROPAGATE_ALL_DIRTY))?;
            }
        } else {
            // steady state: only the rows whose Transform changed since this system last ran
            if frame.is_some() {
                // the fused frame writes them straight into the library's pinned upload window (no Vec, no staging copy): count,
                // map, fill, commit
                let changed_in = |ticks_of: &[Tick]| ticks_of.iter().filter(|t| changed_since(**t, ticks)).count();
                let transforms_again = transforms.contiguous_iter().expect("Transform and ChildOf are table components");
                let capacity: usize = transforms_again.map(|(_, t, _)| changed_in(t.changed_ticks_slice())).sum();
                if capacity == mi.row_entity.len() && mi.rows_in_table_order && capacity >= 65_536 {
                    // Every Transform moved and rows follow the tables: dense windows, one after the other, each starting where the
                    // one before ended.  Window k crosses PCIe while this loop fills window k + 1, and the library -- which sees a
                    // sequence of dense windows that carries the whole table -- computes each window's GlobalTransforms as it
                    // arrives and sends them back under the rest of the upload: `mi_download_frame_results` finds them on the host
                    // (bevy_mi355x.h, mi_download_frame_results; the C++ host layer does the same).  The caller's frame call is then
                    // the all-rows one: no MI_CULL_CHANGED_ROWS (`every_row_moved`).
                    const WINDOWS: usize = 8;
                    let components = mi.upload_components;
                    let n = capacity;
                    let mut row = 0usize; // rows written so far == the table walk's position
                    let mut window: ffi::MiUploadWindow = unsafe { core::mem::zeroed() };
                    let (mut lo, mut hi) = (0usize, 0usize); // the open window carries rows [lo, hi)
                    let mut next_window = 0usize;            // windows mapped so far: window k carries [n k / 8, n (k + 1) / 8)
                    let mut open = false;
                    // A failing map / commit must not leave a window mapped: it would never be recycled (its pinned chunk stays out of
                    // the pool for the life of the context).  Every exit below gives the open window back with n = 0.
                    let r = (|| -> Result<(), ()> {
                        for (_, table_transforms, _) in tables {
                            for t in table_transforms.iter() {
                                if row == hi {
                                    if open {
                                        open = false;
                                        check(ctx, "mi_commit_upload_window", unsafe { ffi::mi_commit_upload_window(ctx, &window, (hi - lo) as u32, lo as u32) })?;
                                    }
                                    lo = hi;
                                    // exactly WINDOWS pieces (fewer when n < WINDOWS): boundaries n (k + 1) / WINDOWS, empty ones skipped
                                    while hi <= lo {
                                        next_window += 1;
                                        hi = n * next_window / WINDOWS;
                                    }
                                    check(ctx, "mi_map_upload_window", unsafe {
                                        ffi::mi_map_upload_window(ctx, (hi - lo) as u32, ffi::MI_UPLOAD_DENSE | components, &mut window)
                                    })?;
                                    open = true;
                                }
                                let k = row - lo;
                                // SAFETY: k < hi - lo, the capacity the window was mapped with; a component the window does not
                                // carry has a null pointer.
                                unsafe {
                                    if !window.translation.is_null() {
                                        core::ptr::copy_nonoverlapping(t.translation.to_array().as_ptr(), window.translation.add(3 * k), 3);
                                    }
                                    if !window.rotation.is_null() {
                                        core::ptr::copy_nonoverlapping(t.rotation.to_array().as_ptr(), window.rotation.add(4 * k), 4);
                                    }
                                    if !window.scale.is_null() {
                                        core::ptr::copy_nonoverlapping(t.scale.to_array().as_ptr(), window.scale.add(3 * k), 3);
                                    }
                                }
                                row += 1;
                            }
                        }
                        if open {
                            open = false;
                            check(ctx, "mi_commit_upload_window", unsafe { ffi::mi_commit_upload_window(ctx, &window, (hi - lo) as u32, lo as u32) })?;
                        }
                        Ok(())
                    })();
                    if r.is_err() {
                        if open {
                            // SAFETY: `window` is the mapped window; n = 0 only returns it to the pool.
                            let _ = unsafe { ffi::mi_commit_upload_window(ctx, &window, 0, 0) };
                        }
                        return Err(());
                    }
                    mi.every_row_moved = true;
                    return Ok(0);
                }
                // SAFETY: a zeroed window is the valid "nothing mapped" value; the library fills it in.
                let mut window: ffi::MiUploadWindow = unsafe { core::mem::zeroed() };
                check(ctx, "mi_map_upload_window", unsafe { ffi::mi_map_upload_window(ctx, capacity as u32, mi.upload_components, &mut window) })?;
                let mut k = 0usize;
                for (entities, table_transforms, _) in tables {
                    let changed = table_transforms.changed_ticks_slice();
                    for (i, t) in table_transforms.iter().enumerate() {
                        if changed_since(changed[i], ticks) {
                            // SAFETY: k < capacity: the same filter counted the rows above.
                            unsafe {
                                *window.rows.add(k) = mi.entity_row[&entities[i]];
                                if !window.translation.is_null() {
                                    core::ptr::copy_nonoverlapping(t.translation.to_array().as_ptr(), window.translation.add(3 * k), 3);
                                }
                                if !window.rotation.is_null() {
                                    core::ptr::copy_nonoverlapping(t.rotation.to_array().as_ptr(), window.rotation.add(4 * k), 4);
                                }
                                if !window.scale.is_null() {
                                    core::ptr::copy_nonoverlapping(t.scale.to_array().as_ptr(), window.scale.add(3 * k), 3);
                                }
                            }
                            k += 1;
                        }
                    }
                }
                check(ctx, "mi_commit_upload_window", unsafe { ffi::mi_commit_upload_window(ctx, &window, k as u32, 0) })?;
                return Ok(0); // the caller's frame call propagates (MI_CULL_CHANGED_ROWS) and its results carry the rows
            }
            s.rows.clear();
            s.translation.clear();
            s.rotation.clear();
            s.scale.clear();
            for (entities, table_transforms, _) in tables {
                let changed = table_transforms.changed_ticks_slice();
                for (i, t) in table_transforms.iter().enumerate() {
                    if changed_since(changed[i], &ticks) {
                        s.rows.push(mi.entity_row[&entities[i]]);
                        s.translation.extend_from_slice(&t.translation.to_array());
                        s.rotation.extend_from_slice(&t.rotation.to_array());
                        s.scale.extend_from_slice(&t.scale.to_array());
                    }
                }
            }
            // SAFETY: `rows` and the three columns hold `rows.len()` entries each.
            unsafe {
                check(ctx, 
                    "mi_upload_transforms_indexed",
                    ffi::mi_upload_transforms_indexed(
                        ctx,
                        s.rows.len() as u32,
                        s.rows.as_ptr(),
                        s.translation.as_ptr(),
                        s.rotation.as_ptr(),
                        s.scale.as_ptr(),
                    ),
                )?;
                check(ctx, "mi_propagate", ffi::mi_propagate(ctx, 0))?;
            }
        }
        // what the device changed, compacted on the device: ascending rows + 3x4 column-major matrices
        let capacity = mi.row_entity.len() as u32;
        s.rows.resize(capacity as usize, 0);
        s.global12.resize(capacity as usize * 12, 0.0);
        let mut count = 0u32;
        // SAFETY: both outputs hold `capacity` entries.
        check(ctx, "mi_download_changed_global_transforms", unsafe {
            ffi::mi_download_changed_global_transforms(ctx, s.rows.as_mut_ptr(), s.global12.as_mut_ptr(), capacity, &mut count)
        })?;
        Ok(count)
    }
}

/// `scratch.rows[..count]` / `scratch.global12` -> `Mut<GlobalTransform>`.
fn write_back_global_transforms(mi: &Mi355x, count: u32, globals: &mut Query<&mut GlobalTransform>) {
    #[cfg(feature = "trace")]
    let _span = info_span!("mi355x write back GlobalTransform").entered();
    let s = &mi.scratch;
    for (k, row) in s.rows[..count as usize].iter().enumerate() {
        let cols: &[f32; 12] = s.global12[k * 12..k * 12 + 12].try_into().unwrap();
        if let Ok(mut global) = globals.get_mut(mi.row_entity[*row as usize]) {
            // plain assignment: the device already compared against the old value (the row is listed only if it differs or its
            // tree was dirty), which is `set_if_neq`'s contract at systems.rs:719
            *global = GlobalTransform::from(Affine3A::from_cols_array(cols));
        }
    }
}

/// `check_visibility_cpu_culling` on the device: per active camera the frustum / render-layer / visibility-range tests of
/// visibility/mod.rs:800-846 over the resident columns, `ViewVisibility::set_visible` on the rows that passed for any view, and
/// the per-class `VisibleEntities` lists, sorted.
pub fn mi_check_visibility(
    mut mi: ResMut<Mi355x>,
    mut fallback: ResMut<CpuFallback>,
    ticks: SystemChangeTick,
    mut view_query: Query<(Entity, &mut VisibleEntities, &Frustum, Option<&RenderLayers>, &Camera, Has<NoCpuCulling>)>,
    bounds_changed: BoundsChanged,
    rows_query: RowsQuery,
    range_views: RangeViews,
    visible_entity_ranges: Option<Res<VisibleEntityRanges>>,
    mut view_visibilities: Query<&mut ViewVisibility, Without<NoCpuCulling>>,
) {
    #[cfg(feature = "trace")]
    let _span = info_span!("mi_check_visibility").entered();
    let _ = ticks;
    if fallback.visibility || fallback.transforms {
        // without the device-resident GlobalTransform column there is nothing to cull against
        fallback.visibility = true;
        return;
    }
    let mi = &mut *mi;
    let ctx = mi.ctx;
    let n = mi.row_entity.len() as u32;

    let result: Result<(), ()> = (|| {
        // --- columns that change rarely: re-staged only when one of them changed (or the VisibleEntityRanges resource came / went)
        let ranges_on = visible_entity_ranges.is_some();
        if !bounds_changed.is_empty() || mi.ranges_resource != Some(ranges_on) {
            stage_bounds(mi, &rows_query, ranges_on)?;
        }
        // (this system runs behind check_visibility_ranges and TransformSystems::Propagate: the views' GlobalTransforms are this frame's)
        let range_table = if ranges_on { range_view_table(&range_views) } else { EntityHashMap::default() };

        // --- the frame's views: active cameras in query order (visibility/mod.rs:778-784)
        let s = &mut mi.scratch;
        s.views.clear();
        s.view_entities.clear();
        for (entity, _, frustum, layers, camera, no_cpu_culling) in view_query.iter() {
            if !camera.is_active {
                continue;
            }
            let (layer_mask, layer_mask_hi) = match layers {
                None => (1, 0),
                Some(l) => layer_words_or_log(l)?,
            };
            // visibility/mod.rs:814-820: with the resource, a ranged entity is tested against THIS view's index and position
            let range_origin = range_table.get(&entity);
            let mut view = ffi::MiView {
                frustum: [0.0; 24],
                layer_mask,
                flags: (if no_cpu_culling { ffi::MI_VIEW_FLAG_NO_CPU_CULLING } else { 0 }) | (if range_origin.is_some() { ffi::MI_VIEW_FLAG_RANGES } else { 0 }),
                position: range_origin.copied().unwrap_or([0.0; 3]),
                light_sphere: [0.0; 4],
                layer_mask_hi,
                reserved: [0; 2],
            };
            for (p, half_space) in frustum.half_spaces.iter().enumerate() {
                view.frustum[p * 4..p * 4 + 4].copy_from_slice(&half_space.normal_d().to_array());
            }
            s.views.push(view);
            s.view_entities.push(entity);
        }
        // one whole frame on the device: reset, every view, newly-hidden -> one kernel launch
        // SAFETY: `views` holds `views.len()` entries.
        check(ctx, "mi_cull_views", unsafe {
            ffi::mi_cull_views(ctx, s.views.as_ptr(), s.views.len() as u32, ffi::MI_CULL_BEGIN_FRAME | ffi::MI_CULL_END_FRAME)
        })?;
        s.view_visibility.resize(n as usize, 0);
        s.vv_changed.resize((n as usize + 31) / 32, 0);
        // SAFETY: `n` bytes and ceil(n / 32) words.
        check(ctx, "mi_download_view_visibility", unsafe {
            ffi::mi_download_view_visibility(ctx, 0, n, s.view_visibility.as_mut_ptr(), s.vv_changed.as_mut_ptr())
        })?;
        Ok(())
    })();
    if result.is_err() {
        fallback.visibility = true;
        return;
    }

    // --- ECS writes.  `reset_view_visibility` has already shifted current -> previous on every component; `set_visible` on the
    //     rows the device found visible reproduces the stock system's writes (and its change ticks, mod.rs:290-306) one to one.
    for (row, vv) in mi.scratch.view_visibility.iter().enumerate() {
        if vv & 1 != 0 {
            if let Ok(mut view_visibility) = view_visibilities.get_mut(mi.row_entity[row]) {
                view_visibility.set_visible();
            }
        }
    }
    let class_bits: Vec<(TypeId, u32)> = mi.class_bits.iter().map(|(k, v)| (*k, *v)).collect();
    for (slot, view_entity) in mi.scratch.view_entities.clone().into_iter().enumerate() {
        let Ok((_, mut visible_entities, ..)) = view_query.get_mut(view_entity) else { continue };
        visible_entities.clear_all();
        for (class, bit) in &class_bits {
            let s = &mut mi.scratch;
            s.visible_keys.resize(n as usize, 0);
            s.visible_rows.resize(n as usize, 0);
            let mut count = 0u32;
            // SAFETY: both outputs hold `n` entries, the most one list can have.
            let status = unsafe {
                ffi::mi_download_visible_entities(
                    ctx,
                    slot as u32,
                    *bit,
                    s.visible_keys.as_mut_ptr(),
                    s.visible_rows.as_mut_ptr(),
                    n,
                    &mut count,
                )
            };
            if check(ctx, "mi_download_visible_entities", status).is_err() {
                // ViewVisibility is already final and equal to what the CPU system would write; only this frame's lists are
                // rebuilt by it (it clears and refills them, mod.rs:861-875).
                fallback.visibility = true;
                return;
            }
            let list = visible_entities.get_mut(*class);
            list.extend(mi.scratch.visible_keys[..count as usize].iter().map(|bits| Entity::from_bits(*bits)));
            list.sort_unstable(); // mod.rs:872-875: the render world's diffing needs sorted lists
        }
    }
}

/// `RenderLayers` as the two 32-bit words the library matches (`layer_mask`, `layer_mask_hi`: the first u64 word of the bitset,
/// render_layers.rs:121-135), `None` if a layer above 63 is set.
fn layer_words(layers: &RenderLayers) -> Option<(u32, u32)> {
    let mut bits = 0u64;
    for layer in layers.iter() {
        if layer >= 64 {
            return None;
        }
        bits |= 1 << layer;
    }
    Some((bits as u32, (bits >> 32) as u32))
}

/// `RenderLayers` as the one 32-bit word of a cluster object's `layer_mask`, `None` if a layer above 31 is set.
fn layer_word(layers: &RenderLayers) -> Option<u32> {
    let mut word = 0u32;
    for layer in layers.iter() {
        if layer >= 32 {
            return None;
        }
        word |= 1 << layer;
    }
    Some(word)
}

/// The status rule for a capability limit: said once in the log, then `Err` like any failed call (the stock systems take over).
pub(crate) fn layer_words_or_log(layers: &RenderLayers) -> Result<(u32, u32), ()> {
    layer_words(layers).ok_or_else(|| error!("bevy_mi355x: a RenderLayers beyond layer 63; the device columns hold layers 0..63 -- falling back to the CPU systems"))
}
fn layer_word_or_log(layers: &RenderLayers) -> Result<u32, ()> {
    layer_word(layers).ok_or_else(|| {
        error!("bevy_mi355x: a clusterable object or clustered view beyond layer 31 outside the fused frame; mi_cluster_assign_frame takes one 32-bit word -- falling back to the CPU system")
    })
}

pub(crate) fn mi_class_bit(table: &mut HashMap<TypeId, u32>, class: TypeId) -> Option<u32> {
    let next = table.len() as u32;
    if let Some(bit) = table.get(&class) {
        return Some(*bit);
    }
    (next < 32).then(|| {
        table.insert(class, next);
        next
    })
}

/// `ClusterConfig` (crates/bevy_light/src/cluster/mod.rs:104-139) as the plain struct of the C ABI.
fn cluster_config_to_ffi(config: &ClusterConfig) -> ffi::MiClusterConfig {
    let mut out = ffi::MiClusterConfig {
        kind: ffi::MI_CLUSTER_CONFIG_NONE,
        dimensions: [0; 3],
        total: 0,
        z_slices: 0,
        first_slice_depth: 0.0,
        far_z_mode: ffi::MI_CLUSTER_FAR_Z_CONSTANT,
        far_z_constant: 0.0,
        dynamic_resizing: 0,
    };
    let mut z = |z_config: &bevy_light::cluster::ClusterZConfig, out: &mut ffi::MiClusterConfig| {
        out.first_slice_depth = z_config.first_slice_depth;
        match z_config.far_z_mode {
            ClusterFarZMode::MaxClusterableObjectRange => out.far_z_mode = ffi::MI_CLUSTER_FAR_Z_MAX_CLUSTERABLE_OBJECT_RANGE,
            ClusterFarZMode::Constant(far) => {
                out.far_z_mode = ffi::MI_CLUSTER_FAR_Z_CONSTANT;
                out.far_z_constant = far;
            }
        }
    };
    match config {
        ClusterConfig::None => {}
        ClusterConfig::Single => out.kind = ffi::MI_CLUSTER_CONFIG_SINGLE,
        ClusterConfig::XYZ { dimensions, z_config, dynamic_resizing } => {
            out.kind = ffi::MI_CLUSTER_CONFIG_XYZ;
            out.dimensions = dimensions.to_array();
            out.dynamic_resizing = *dynamic_resizing as u32;
            z(z_config, &mut out);
        }
        ClusterConfig::FixedZ { total, z_slices, z_config, dynamic_resizing } => {
            out.kind = ffi::MI_CLUSTER_CONFIG_FIXED_Z;
            out.total = *total;
            out.z_slices = *z_slices;
            out.dynamic_resizing = *dynamic_resizing as u32;
            z(z_config, &mut out);
        }
    }
    out
}

/// `assign_objects_to_clusters` on the device.  The object list is gathered exactly as the reference gathers it (query order:
/// point lights, spot lights, rect lights, light probes, decals; visible ones only; assign.rs:190-296), limited and sorted as it
/// limits and sorts it (`mi_cluster_sort_truncate`, assign.rs:298-356), and each view is resolved, walked and filled by
/// `mi_cluster_assign_frame` (config -> dimensions incl. the `MaxClusterableObjectRange` / `dynamic_resizing` feedback of the
/// previous frame, assign.rs:358-470; the walk itself, assign.rs:472-805).
pub fn mi_assign_objects_to_clusters(
    mut mi: ResMut<Mi355x>,
    mut fallback: ResMut<CpuFallback>,
    mut views: Query<(Entity, &GlobalTransform, &Camera, &Frustum, Option<&ClusterConfig>, &mut Clusters, Option<&RenderLayers>)>,
    point_lights: Query<(Entity, &GlobalTransform, &ViewVisibility, &PointLight, Option<&RenderLayers>, Option<&VolumetricLight>)>,
    spot_lights: Query<(Entity, &GlobalTransform, &ViewVisibility, &SpotLight, Option<&RenderLayers>, Option<&VolumetricLight>)>,
    rect_lights: Query<(Entity, &GlobalTransform, &ViewVisibility, &RectLight, Option<&RenderLayers>)>,
    light_probes: Query<(Entity, &GlobalTransform, &ViewVisibility, Has<EnvironmentMapLight>), With<LightProbe>>,
    decals: Query<(Entity, &GlobalTransform, &ViewVisibility), With<ClusteredDecal>>,
    settings: Option<Res<GlobalClusterSettings>>,
) {
    #[cfg(feature = "trace")]
    let _span = info_span!("mi_assign_objects_to_clusters").entered();
    let Some(settings) = settings else { return };
    if fallback.clusters || settings.gpu_clustering.is_some() {
        // wgpu-side clustering is a different path (assign.rs:187); leave it to the stock system
        fallback.clusters = true;
        return;
    }
    let mi = &mut *mi;
    let ctx = mi.ctx;

    struct ViewResult {
        entity: Entity,
        view: ffi::MiClusterView,
        active: bool,
        offsets: Vec<u32>,
        counts: Vec<u32>,
        indices: Vec<u32>,
        farthest_z: f32,
        total: u64,
    }

    let result: Result<Vec<ViewResult>, ()> = (|| {
        let s = &mut mi.scratch;
        for v in [&mut s.obj_pos_range, &mut s.obj_spot_dir, &mut s.obj_spot_sin_cos] {
            v.clear();
        }
        for v in [&mut s.obj_type, &mut s.obj_shadows, &mut s.obj_volumetric] {
            v.clear();
        }
        s.obj_layers.clear();
        s.obj_entity.clear();
        let mut push = |s: &mut Scratch,
                        entity: Entity,
                        transform: &GlobalTransform,
                        range: f32,
                        kind: i32,
                        layers: Option<&RenderLayers>,
                        shadows: bool,
                        volumetric: bool,
                        spot: Option<(f32, [f32; 3])>|
         -> Result<(), ()> {
            let p = transform.translation();
            s.obj_pos_range.extend_from_slice(&[p.x, p.y, p.z, range]);
            s.obj_type.push(kind as u8);
            s.obj_layers.push(match layers {
                None => 1,
                Some(l) => layer_word_or_log(l)?,
            });
            s.obj_shadows.push(shadows as u8);
            s.obj_volumetric.push(volumetric as u8);
            let (angle, dir) = spot.unwrap_or((0.0, [0.0; 3]));
            let (sin, cos) = ops_sin_cos(angle);
            s.obj_spot_dir.extend_from_slice(&dir);
            s.obj_spot_sin_cos.extend_from_slice(&[sin, cos]);
            s.obj_entity.push(entity);
            Ok(())
        };
        for (e, t, vv, light, layers, vol) in point_lights.iter().filter(|q| q.2.get()) {
            push(s, e, t, light.range, ffi::MI_OBJ_POINT_LIGHT, layers, light.shadow_maps_enabled, vol.is_some(), None)?;
            let _ = vv;
        }
        for (e, t, _, light, layers, vol) in spot_lights.iter().filter(|q| q.2.get()) {
            // the column takes `GlobalTransform::back()` as the reference computes it (assign.rs:563-573)
            let dir = t.back().to_array();
            push(s, e, t, light.range, ffi::MI_OBJ_SPOT_LIGHT, layers, light.shadow_maps_enabled, vol.is_some(), Some((light.outer_angle, dir)))?;
        }
        if settings.supports_storage_buffers {
            // rect lights are gathered only where they are clustered at all (assign.rs:231-248) ...
            for (e, t, _, light, layers) in rect_lights.iter().filter(|q| q.2.get()) {
                push(s, e, t, light.range, ffi::MI_OBJ_RECT_LIGHT, layers, false, false, None)?;
            }
            // ... and so are light probes (UBOs cannot hold their indices, assign.rs:250-277); range = `transform.radius_vec3a(Vec3A::ONE)`
            for (e, t, _, is_reflection_probe) in light_probes.iter().filter(|q| q.2.get()) {
                let kind = if is_reflection_probe { ffi::MI_OBJ_REFLECTION_PROBE } else { ffi::MI_OBJ_IRRADIANCE_VOLUME };
                push(s, e, t, t.radius_vec3a(bevy_math::Vec3A::ONE), kind, None, false, false, None)?;
            }
        }
        if settings.clustered_decals_are_usable {
            // decals have a gate of their own (assign.rs:279-296); range = `transform.scale().length()`
            for (e, t, _) in decals.iter().filter(|q| q.2.get()) {
                push(s, e, t, t.scale().length(), ffi::MI_OBJ_DECAL, None, false, false, None)?;
            }
        }

        // --- the uniform-buffer limit: stable sort by (type, shadows, volumetric, entity) and truncate (assign.rs:298-356)
        let n_all = s.obj_entity.len() as u32;
        s.keys.clear();
        s.keys.extend(s.obj_entity.iter().map(|e| e.to_bits()));
        s.obj_order.resize(n_all as usize, 0);
        let mut n_kept = 0u32;
        // SAFETY: every input holds `n_all` entries, `out_order` too.
        check(ctx, "mi_cluster_sort_truncate", unsafe {
            ffi::mi_cluster_sort_truncate(
                n_all,
                s.obj_type.as_ptr(),
                s.obj_shadows.as_ptr(),
                s.obj_volumetric.as_ptr(),
                s.keys.as_ptr(),
                settings.max_uniform_buffer_clusterable_objects as u32,
                settings.supports_storage_buffers as u32,
                s.obj_order.as_mut_ptr(),
                &mut n_kept,
            )
        })?;
        let order = &s.obj_order[..n_kept as usize];
        let pick = |src: &Vec<f32>, w: usize| -> Vec<f32> {
            order.iter().flat_map(|&o| src[o as usize * w..(o as usize + 1) * w].iter().copied()).collect()
        };
        let pos_range = pick(&s.obj_pos_range, 4);
        let spot_dir = pick(&s.obj_spot_dir, 3);
        let spot_sin_cos = pick(&s.obj_spot_sin_cos, 2);
        let obj_type: Vec<u8> = order.iter().map(|&o| s.obj_type[o as usize]).collect();
        let obj_layers: Vec<u32> = order.iter().map(|&o| s.obj_layers[o as usize]).collect();
        let entities: Vec<Entity> = order.iter().map(|&o| s.obj_entity[o as usize]).collect();
        s.obj_entity = entities;
        s.obj_type.clone_from(&obj_type);
        // SAFETY: `n_kept` entries per column.
        check(ctx, "mi_cluster_upload_objects", unsafe {
            ffi::mi_cluster_upload_objects(
                ctx,
                n_kept,
                pos_range.as_ptr(),
                obj_type.as_ptr(),
                obj_layers.as_ptr(),
                spot_dir.as_ptr(),
                spot_sin_cos.as_ptr(),
            )
        })?;

        // --- per view: resolve the config, walk, fill, fetch
        let mut results = Vec::new();
        for (entity, camera_transform, camera, frustum, config, _, layers) in views.iter() {
            let Some(screen) = camera.physical_viewport_size() else {
                continue; // assign.rs:372-375: `clusters.clear()` is the caller's default below
            };
            let config = cluster_config_to_ffi(&config.copied().unwrap_or_default());
            let history = mi.cluster_history.entry(entity).or_insert(ffi::MiClusterHistory {
                has_farthest_z: 0,
                farthest_z: 0.0,
                has_total_cluster_index_count: 0,
                reserved: 0,
                total_cluster_index_count: 0,
            });
            let mut frustum12 = [0f32; 24];
            for (p, half_space) in frustum.half_spaces.iter().enumerate() {
                frustum12[p * 4..p * 4 + 4].copy_from_slice(&half_space.normal_d().to_array());
            }
            let camera_affine = camera_transform.affine().to_cols_array();
            let clip_from_view = camera.clip_from_view().to_cols_array();
            // SAFETY: a zeroed mi_cluster_view is a valid "empty" value (null plane pointers, zero dims).
            let mut view: ffi::MiClusterView = unsafe { core::mem::zeroed() };
            let mut active = 0u32;
            // SAFETY: fixed-size arrays of the sizes the header states; `history` and `view` are live.
            let status = unsafe {
                ffi::mi_cluster_assign_frame(
                    ctx,
                    &config,
                    history,
                    camera_affine.as_ptr(),
                    clip_from_view.as_ptr(),
                    frustum12.as_ptr(),
                    screen.x,
                    screen.y,
                    match layers {
                        None => 1,
                        Some(l) => layer_word_or_log(l)?,
                    },
                    settings.view_cluster_bindings_max_indices as u64,
                    &mut view,
                    &mut active,
                )
            };
            check(ctx, "mi_cluster_assign_frame", status)?;
            let n_clusters = (view.dims[0] * view.dims[1] * view.dims[2]) as usize;
            let mut r = ViewResult {
                entity,
                view,
                active: active != 0,
                offsets: vec![0; n_clusters + 1],
                counts: vec![0; n_clusters * 6],
                indices: Vec::new(),
                farthest_z: 0.0,
                total: 0,
            };
            if r.active {
                // the history already holds this frame's total (mi_cluster_assign_frame wrote it back, assign.rs:810-811)
                r.indices.resize(history.total_cluster_index_count as usize, 0);
                // SAFETY: offsets n+1, counts 6n, indices `capacity` entries.
                let status = unsafe {
                    ffi::mi_cluster_download(
                        ctx,
                        r.offsets.as_mut_ptr(),
                        r.indices.as_mut_ptr(),
                        r.indices.len() as u64,
                        r.counts.as_mut_ptr(),
                        &mut r.total,
                        &mut r.farthest_z,
                    )
                };
                check(ctx, "mi_cluster_download", status)?;
            }
            results.push(r);
        }
        Ok(results)
    })();

    let Ok(results) = result else {
        fallback.clusters = true; // `Clusters` untouched; the gated stock set runs next
        return;
    };
    for r in results {
        let Ok((.., mut clusters, _)) = views.get_mut(r.entity) else { continue };
        let history = mi.cluster_history[&r.entity];
        clusters.tile_size = UVec2::from_array(r.view.tile_size);
        clusters.dimensions = UVec3::from_array(r.view.dims);
        clusters.near = r.view.near_;
        clusters.far = r.view.far_;
        clusters.last_frame_farthest_z = (history.has_farthest_z != 0).then_some(history.farthest_z);
        clusters.last_frame_total_cluster_index_count =
            (history.has_total_cluster_index_count != 0).then_some(history.total_cluster_index_count as usize);
        let n_clusters = (r.view.dims[0] * r.view.dims[1] * r.view.dims[2]) as usize;
        let mut per_cluster: Vec<ObjectsInClusterCpu> = Vec::with_capacity(n_clusters);
        for c in 0..n_clusters {
            let mut objects = ObjectsInClusterCpu::default();
            if r.active {
                // a cluster's list is in object order (ascending index), the order the reference's per-object loop makes its
                // `add_*` calls in (assign.rs:560-800, cluster/mod.rs:479-512)
                for &object in &r.indices[r.offsets[c] as usize..r.offsets[c + 1] as usize] {
                    let entity = mi.scratch.obj_entity[object as usize];
                    match mi.scratch.obj_type[object as usize] as i32 {
                        ffi::MI_OBJ_POINT_LIGHT => objects.add_point_light(entity),
                        ffi::MI_OBJ_SPOT_LIGHT => objects.add_spot_light(entity),
                        ffi::MI_OBJ_RECT_LIGHT => objects.add_rect_light(entity),
                        ffi::MI_OBJ_REFLECTION_PROBE => objects.add_reflection_probe(entity),
                        ffi::MI_OBJ_IRRADIANCE_VOLUME => objects.add_irradiance_volume(entity),
                        _ => objects.add_decal(entity),
                    }
                }
                debug_assert_eq!(objects.counts.point_lights, r.counts[c * 6]);
            }
            per_cluster.push(objects);
        }
        clusters.clusterable_objects = ClusterableObjects::Cpu(per_cluster);
    }
}

// =====================================================================================================================
// Shadow views
// =====================================================================================================================

/// Where the list of one shadow view goes: `kind` 0 = cascade `index` of (`light`, camera `view`), 1 = cube face `index` of point light
/// `light`, 2 = spot light `light`.
struct ShadowList {
    kind: u8,
    light: Entity,
    view: Entity,
    index: usize,
}

/// `check_dir_light_mesh_visibility` + `check_point_light_mesh_visibility` (crates/bevy_light/src/lib.rs:342-515, 517-757) on the
/// device: every cascade of every (directional light, camera) pair, the six cube faces of every shadow-mapped point light and the
/// frustum of every shadow-mapped spot light some camera sees, as ONE pass over the resident columns (`mi_check_light_mesh_visibility`:
/// the per-entity closures of lib.rs:425-475, 592-650, 694-738, selected per view by its flags), one device wait.  The survivors
/// are ORed into `ViewVisibility` (`set_visible`, lib.rs:499-510, 629, 723) -- here on the ECS components, between the cameras' pass
/// and the stock `mark_newly_hidden_entities_invisible`; on the device into the column the next frame's `reset` reads -- and every
/// view's `VisibleMeshEntities` is rebuilt, sorted (lib.rs:489, 664, 745).  The light frusta are inputs, as for the stock systems:
/// `build_directional_light_cascades`, `update_directional_light_frusta`, `update_point_light_frusta`, `update_spot_light_frusta` stay.
#[allow(clippy::too_many_arguments, clippy::type_complexity)]
pub fn mi_check_light_mesh_visibility(
    mut mi: ResMut<Mi355x>,
    mut fallback: ResMut<CpuFallback>,
    visible_point_lights: Query<&VisibleEntities>,
    mut directional_lights: Query<
        (Entity, &DirectionalLight, &CascadesFrusta, &mut CascadesVisibleEntities, Option<&RenderLayers>, &ViewVisibility),
        Without<SpotLight>,
    >,
    mut point_lights: Query<(&PointLight, &GlobalTransform, &CubemapFrusta, &mut CubemapVisibleEntities, Option<&RenderLayers>)>,
    mut spot_lights: Query<(&SpotLight, &GlobalTransform, &Frustum, &mut VisibleMeshEntities, Option<&RenderLayers>)>,
    mut camera_query: Query<(Entity, &RenderTarget), With<Camera>>,
    mut shadow_lod_origin_query: Query<Entity, With<ShadowLodOrigin>>,
    mut point_and_spot_light_query: Query<Entity, Or<(With<PointLight>, With<SpotLight>)>>,
    range_views: RangeViews,
    visible_entity_ranges: Option<Res<VisibleEntityRanges>>,
    // (Without<DirectionalLight>: disjoint from `directional_lights`, which reads the lights' own ViewVisibility; the shadow views'
    // query excludes directional lights anyway, lib.rs:368)
    mut view_visibilities: Query<&mut ViewVisibility, (Without<NoCpuCulling>, Without<DirectionalLight>)>,
    mut checked_lights: Local<EntityHashSet>,
) {
    #[cfg(feature = "trace")]
    let _span = info_span!("mi_check_light_mesh_visibility").entered();
    if fallback.light_visibility || fallback.visibility || fallback.transforms {
        // the columns the shadow views are tested against (GlobalTransform, flags, ranges) are no longer kept current
        fallback.light_visibility = true;
        return;
    }
    let mi = &mut *mi;
    let ctx = mi.ctx;
    let n = mi.row_entity.len();
    let ranges_on = visible_entity_ranges.is_some();
    let range_table = if ranges_on { range_view_table(&range_views) } else { EntityHashMap::default() };
    let frustum_planes = |frustum: &Frustum| {
        let mut planes = [0f32; 24];
        for (p, half_space) in frustum.half_spaces.iter().enumerate() {
            planes[p * 4..p * 4 + 4].copy_from_slice(&half_space.normal_d().to_array());
        }
        planes
    };
    let mut lists: Vec<ShadowList> = Vec::new();
    let mut views = core::mem::take(&mut mi.scratch.shadow_views);
    views.clear();

    // ---- directional lights: the bookkeeping of lib.rs:380-404 as it is, one view per cascade (lib.rs:408-475)
    let gather: Result<(), ()> = (|| {
        for (light, directional_light, frusta, mut visible_entities, maybe_view_mask, light_view_visibility) in &mut directional_lights {
            let mut views_to_remove = Vec::new();
            for (view, cascade_view_entities) in &mut visible_entities.entities {
                match frusta.frusta.get(view) {
                    Some(view_frusta) => cascade_view_entities.resize(view_frusta.len(), Default::default()),
                    None => views_to_remove.push(*view),
                };
            }
            for (view, view_frusta) in &frusta.frusta {
                visible_entities.entities.entry(*view).or_insert_with(|| vec![VisibleMeshEntities::default(); view_frusta.len()]);
            }
            for v in views_to_remove {
                visible_entities.entities.remove(&v);
            }
            // NOTE: If shadow mapping is disabled for the light then it must have no visible entities (lib.rs:400-404)
            if !directional_light.shadow_maps_enabled || !light_view_visibility.get() {
                visible_entities.entities.clear();
                continue;
            }
            let (layer_mask, layer_mask_hi) = match maybe_view_mask {
                None => (1, 0),
                Some(l) => layer_words_or_log(l)?,
            };
            for (view, view_frusta) in &frusta.frusta {
                // entity_is_in_range_of_view(entity, *view): the cascade's CAMERA is the range view (lib.rs:437-443)
                let range_origin = range_table.get(view);
                for (cascade, frustum) in view_frusta.iter().enumerate() {
                    views.push(ffi::MiView {
                        frustum: frustum_planes(frustum),
                        layer_mask,
                        flags: ffi::MI_VIEW_KIND_CASCADE | (if range_origin.is_some() { ffi::MI_VIEW_FLAG_RANGES } else { 0 }),
                        position: range_origin.copied().unwrap_or([0.0; 3]),
                        light_sphere: [0.0; 4],
                        layer_mask_hi,
                        reserved: [0; 2],
                    });
                    lists.push(ShadowList { kind: 0, light, view: *view, index: cascade });
                }
            }
        }

        // ---- point and spot lights: the ones in some camera's VisibleEntities, each once (lib.rs:559-569)
        checked_lights.clear();
        let shadow_lod_origin = get_shadow_lod_origin(
            camera_query.transmute_lens_filtered(),
            shadow_lod_origin_query.transmute_lens_filtered(),
            point_and_spot_light_query.transmute_lens_filtered(),
        );
        // lib.rs:601-611: `visible_entity_ranges.is_some_and(|r| shadow_lod_origin.is_none_or(|o| !r.entity_is_in_range_of_view(entity, o)))`
        // -- no origin, or an origin without an index: every ranged entity is culled
        let (range_flags, range_position) = if !ranges_on {
            (0, [0.0; 3])
        } else {
            match shadow_lod_origin.and_then(|origin| range_table.get(&origin)) {
                Some(position) => (ffi::MI_VIEW_FLAG_RANGES, *position),
                None => (ffi::MI_VIEW_FLAG_RANGES_NO_ORIGIN, [0.0; 3]),
            }
        };
        for visible_lights in &visible_point_lights {
            for &light_entity in visible_lights.get(TypeId::of::<ClusterVisibilityClass>()) {
                if !checked_lights.insert(light_entity) {
                    continue;
                }
                if let Ok((point_light, transform, cubemap_frusta, _, maybe_view_mask)) = point_lights.get(light_entity) {
                    if !point_light.shadow_maps_enabled {
                        continue; // (lib.rs:579-581; `continue` also skips the spot branch, as there)
                    }
                    let (layer_mask, layer_mask_hi) = match maybe_view_mask {
                        None => (1, 0),
                        Some(l) => layer_words_or_log(l)?,
                    };
                    let center = transform.translation().to_array();
                    for (face, frustum) in cubemap_frusta.iter().enumerate() {
                        views.push(ffi::MiView {
                            frustum: frustum_planes(frustum),
                            layer_mask,
                            flags: ffi::MI_VIEW_KIND_CUBE_FACE_OR_SPOT | range_flags,
                            position: range_position,
                            light_sphere: [center[0], center[1], center[2], point_light.range], // lib.rs:584-587
                            layer_mask_hi,
                            reserved: [0; 2],
                        });
                        lists.push(ShadowList { kind: 1, light: light_entity, view: light_entity, index: face });
                    }
                }
                if let Ok((spot_light, transform, frustum, _, maybe_view_mask)) = spot_lights.get(light_entity) {
                    if !spot_light.shadow_maps_enabled {
                        continue;
                    }
                    let (layer_mask, layer_mask_hi) = match maybe_view_mask {
                        None => (1, 0),
                        Some(l) => layer_words_or_log(l)?,
                    };
                    let center = transform.translation().to_array();
                    views.push(ffi::MiView {
                        frustum: frustum_planes(frustum),
                        layer_mask,
                        flags: ffi::MI_VIEW_KIND_CUBE_FACE_OR_SPOT | range_flags,
                        position: range_position,
                        light_sphere: [center[0], center[1], center[2], spot_light.range], // lib.rs:679-682
                        layer_mask_hi,
                        reserved: [0; 2],
                    });
                    lists.push(ShadowList { kind: 2, light: light_entity, view: light_entity, index: 0 });
                }
            }
        }
        Ok(())
    })();

    // ---- the device pass.  flags = 0: both camera paths of this plugin have closed the frame on the device already (their
    //      MI_CULL_END_FRAME); the survivors' bit 0 is ORed into the column the next frame's reset reads, the ECS components get their
    //      set_visible() below, in front of the stock mark_newly_hidden_entities_invisible.
    let words = (n + 31) / 32;
    let run = gather.and_then(|()| {
        let s = &mut mi.scratch;
        s.light_masks.clear();
        s.light_masks.resize((views.len() * words).max(1), 0);
        s.light_any.clear();
        s.light_any.resize(words.max(1), 0);
        // SAFETY: `views` holds `views.len()` entries; the outputs hold views.len() * ceil(n / 32) and ceil(n / 32) words.
        check(ctx, "mi_check_light_mesh_visibility", unsafe {
            ffi::mi_check_light_mesh_visibility(ctx, views.as_ptr(), views.len() as u32, 0, s.light_masks.as_mut_ptr(), s.light_any.as_mut_ptr())
        })
    });
    if run.is_err() {
        mi.scratch.shadow_views = views;
        fallback.light_visibility = true; // the stock pair, chained right behind, computes this frame (it redoes the bookkeeping above)
        return;
    }

    // ---- ECS writes: set_visible() on the union, then every view's list
    for (w, word) in mi.scratch.light_any[..words].iter().enumerate() {
        let mut bits = *word;
        while bits != 0 {
            let row = w * 32 + bits.trailing_zeros() as usize;
            bits &= bits - 1;
            if let Ok(mut view_visibility) = view_visibilities.get_mut(mi.row_entity[row]) {
                view_visibility.set_visible();
            }
        }
    }
    for (k, list) in lists.iter().enumerate() {
        let mask = &mi.scratch.light_masks[k * words..(k + 1) * words];
        let mut entities: Vec<Entity> = Vec::new();
        for (w, word) in mask.iter().enumerate() {
            let mut bits = *word;
            while bits != 0 {
                entities.push(mi.row_entity[w * 32 + bits.trailing_zeros() as usize]);
                bits &= bits - 1;
            }
        }
        entities.sort_unstable(); // lib.rs:489, 664, 745
        if list.kind == 0 {
            if let Ok((_, _, _, mut visible_entities, _, _)) = directional_lights.get_mut(list.light) {
                if let Some(view_dest) = visible_entities.entities.get_mut(&list.view).and_then(|cascades| cascades.get_mut(list.index)) {
                    view_dest.entities = entities;
                    view_dest.shrink();
                }
            }
        } else if list.kind == 1 {
            if let Ok((_, _, _, mut cubemap_visible_entities, _)) = point_lights.get_mut(list.light) {
                let view_dest = cubemap_visible_entities.get_mut(list.index);
                view_dest.entities = entities;
                view_dest.shrink();
            }
        } else if let Ok((_, _, _, mut visible_entities, _)) = spot_lights.get_mut(list.light) {
            visible_entities.entities = entities;
            visible_entities.shrink();
        }
    }
    mi.scratch.shadow_views = views;
}

// =====================================================================================================================
// The fused frame
// =====================================================================================================================

/// What [`mi_fused_frame`] parks for the systems that write it into the ECS where the stock systems would.
#[derive(Resource, Default)]
pub struct Mi355xFrame {
    /// The visibility results below belong to this frame and nothing they depend on has changed since the submit.
    pub valid: bool,
    /// The same for the cluster lists.
    pub clusters_valid: bool,
    /// `this_run` of the fused system: inputs whose change tick is newer than this were written after the submit.
    pub(crate) submit_tick: Tick,
    /// Per active camera, in query order: the frustum the cull used and its `VisibleEntities` lists per class.
    pub(crate) views: Vec<FrameView>,
    /// One per clustered camera.
    clusters: Vec<FrameClusters>,
}
// SAFETY: the only raw pointers in here are the plane / sphere tables of a parked `ffi::MiClusterView`.  They are never read again
// after `mi_cluster_upload_view` (the library copied the tables); the systems that consume the parked view use its plain fields.
unsafe impl Send for Mi355xFrame {}
unsafe impl Sync for Mi355xFrame {}
pub(crate) struct FrameView {
    pub(crate) entity: Entity,
    pub(crate) frustum: [f32; 24],
    pub(crate) lists: Vec<(TypeId, Vec<Entity>)>,
}
struct FrameClusters {
    view_entity: Entity,
    view: ffi::MiClusterView,
    offsets: Vec<u32>,
    counts: Vec<u32>,
    indices: Vec<u32>,
    objects: Vec<(Entity, u8)>,
}

fn frame_results_missing(frame: Res<Mi355xFrame>) -> bool {
    !frame.valid
}
fn frame_clusters_missing(frame: Res<Mi355xFrame>) -> bool {
    !frame.clusters_valid
}

/// `GlobalTransform` an entity is about to get: its `Transform` chained up the `ChildOf` links with the reference's own operators
/// (`GlobalTransform::from`, `mul_transform`: global_transform.rs:315-330) -- what `propagate_parent_transforms` computes for it.
pub(crate) fn expected_global(entity: Entity, transforms: &Query<(Entity, Ref<Transform>, Option<&ChildOf>)>) -> Option<GlobalTransform> {
    let mut chain = Vec::new();
    let mut cur = Some(entity);
    while let Some(e) = cur {
        let (_, t, child_of) = transforms.get(e).ok()?;
        chain.push(*t);
        cur = child_of.map(ChildOf::parent).filter(|p| transforms.contains(*p));
    }
    let mut g = GlobalTransform::from(*chain.last()?);
    for t in chain.iter().rev().skip(1) {
        g = g.mul_transform(*t);
    }
    Some(g)
}

/// The whole render-prep frame in one device round trip (module docs).  Runs in `TransformSystems::Propagate`.
#[allow(clippy::too_many_arguments)]
pub fn mi_fused_frame(
    mut mi: ResMut<Mi355x>,
    mut fallback: ResMut<CpuFallback>,
    mut frame: ResMut<Mi355xFrame>,
    ticks: SystemChangeTick,
    static_opt: Option<Res<bevy_transform::systems::StaticTransformOptimizations>>,
    structure_changed: Query<(), Or<(Added<Transform>, Changed<ChildOf>)>>,
    mut orphaned: RemovedComponents<ChildOf>,
    mut despawned: RemovedComponents<Transform>,
    transforms: Query<(Entity, Ref<Transform>, Option<&ChildOf>)>,
    mut globals: Query<&mut GlobalTransform>,
    cameras: Query<(Entity, &Camera, &bevy_camera::Projection, Option<&RenderLayers>, Has<NoCpuCulling>, Option<&ClusterConfig>, Has<Clusters>)>,
    bounds_changed: BoundsChanged,
    rows_query: RowsQuery,
    // VisibilityRange: whether the resource exists and which views check_visibility_ranges will index this frame (it runs later, in
    // VisibilitySystems::CheckVisibility; both are functions of the entity set, not of this frame's transforms)
    (range_views, visible_entity_ranges): (RangeViews, Option<Res<VisibleEntityRanges>>),
    // (one tuple parameter: a system function takes at most 16 parameters, function_system.rs:950)
    (point_lights, spot_lights, rect_lights, light_probes, decals): (
        // (With<ViewVisibility>, With<GlobalTransform>: the reference's queries fetch both, assign.rs:146-178 -- an entity without
        // one of them is not gathered there, so it is not gathered here)
        Query<(Entity, &PointLight, Option<&RenderLayers>), (With<ViewVisibility>, With<GlobalTransform>)>,
        Query<(Entity, &SpotLight, Option<&RenderLayers>), (With<ViewVisibility>, With<GlobalTransform>)>,
        Query<(Entity, &RectLight, Option<&RenderLayers>), (With<ViewVisibility>, With<GlobalTransform>)>,
        // light probes and decals take their range from the GlobalTransform this very frame computes (assign.rs:262, 287): formed here
        // by `expected_global` -- From(Transform) chained down the ChildOf links with the reference's own operators, the products
        // `propagate_parent_transforms` forms for the entity, in its order -- so they ride like the lights, with or without a parent
        Query<(Entity, Has<EnvironmentMapLight>), (With<LightProbe>, With<ViewVisibility>, With<GlobalTransform>)>,
        Query<Entity, (With<ClusteredDecal>, With<ViewVisibility>, With<GlobalTransform>)>,
    ),
    settings: Option<Res<GlobalClusterSettings>>,
) {
    #[cfg(feature = "trace")]
    let _span = info_span!("mi_fused_frame").entered();
    frame.valid = false;
    frame.clusters_valid = false;
    let mi = &mut *mi;
    let ctx = mi.ctx;
    let rebuild = !structure_changed.is_empty() || orphaned.read().count() != 0 || despawned.read().count() != 0;
    if fallback.transforms {
        if !(mi.hierarchy_kept_on_host && rebuild) {
            return;
        }
        *fallback = CpuFallback::default(); // (as in mi_propagate_transforms: a narrow hierarchy whose structure changed is looked at again)
    }
    if rebuild {
        // Rows are renumbered and every column goes up again: the slow frame.  Propagate alone here; the visibility and cluster
        // systems of the three-system form (gated on `frame.valid` / `frame.clusters_valid`) take the rest of this frame.
        match upload_and_propagate(mi, true, &ticks, &transforms, &globals.as_readonly(), None) {
            Ok(count) => write_back_global_transforms(mi, count, &mut globals),
            Err(()) => fallback.transforms = true,
        }
        return;
    }
    let n = mi.row_entity.len() as u32;

    // ---- the frame's views: active cameras in query order, each with the frustum update_frusta WILL give it
    let mut views: Vec<ffi::MiView> = Vec::new();
    let mut frame_views: Vec<FrameView> = Vec::new();
    // every camera with Clusters (assign.rs:324-486 runs per view): (entity, GlobalTransform, frustum, viewport, config, layers, clip_from_view)
    let mut cluster_cameras: Vec<(Entity, GlobalTransform, [f32; 24], UVec2, ClusterConfig, (u32, u32), [f32; 16])> = Vec::new();
    let mut clustered_cameras = 0;
    let ranges_on = visible_entity_ranges.is_some();
    let range_table = if ranges_on { range_view_table(&range_views) } else { EntityHashMap::default() };
    let result: Result<(), ()> = (|| {
        for (entity, camera, projection, layers, no_cpu_culling, config, has_clusters) in cameras.iter() {
            if !camera.is_active {
                continue;
            }
            // (a camera without Transform has no row and no GlobalTransform to predict: the stock systems take this frame)
            let global = expected_global(entity, &transforms).ok_or_else(|| error!("bevy_mi355x: an active camera without a Transform -- falling back to the CPU systems"))?;
            // (Projection derefs to `dyn CameraProjection`: method syntax, as update_frusta itself calls it, visibility/mod.rs:627-636)
            let frustum = projection.compute_frustum(&global);
            let mut planes = [0f32; 24];
            for (p, half_space) in frustum.half_spaces.iter().enumerate() {
                planes[p * 4..p * 4 + 4].copy_from_slice(&half_space.normal_d().to_array());
            }
            let (layer_mask, layer_mask_hi) = match layers {
                None => (1, 0),
                Some(l) => layer_words_or_log(l)?,
            };
            // a view with an index in VisibleEntityRanges: distances from the translation the camera is ABOUT to have (check_visibility_ranges
            // reads its GlobalTransform behind TransformSystems::Propagate, range.rs:245)
            let indexed = range_table.contains_key(&entity);
            views.push(ffi::MiView {
                frustum: planes,
                layer_mask,
                flags: (if no_cpu_culling { ffi::MI_VIEW_FLAG_NO_CPU_CULLING } else { 0 }) | (if indexed { ffi::MI_VIEW_FLAG_RANGES } else { 0 }),
                position: if indexed { global.translation().to_array() } else { [0.0; 3] },
                light_sphere: [0.0; 4],
                layer_mask_hi,
                reserved: [0; 2],
            });
            frame_views.push(FrameView { entity, frustum: planes, lists: Vec::new() });
            if has_clusters {
                clustered_cameras += 1;
                if let Some(size) = camera.physical_viewport_size() {
                    let clip = projection.get_clip_from_view().to_cols_array();
                    cluster_cameras.push((entity, global, planes, size, config.copied().unwrap_or_default(), (layer_mask, layer_mask_hi), clip));
                }
            }
        }
        // ---- columns that change rarely (exactly as mi_check_visibility stages them)
        if !bounds_changed.is_empty() || mi.ranges_resource != Some(ranges_on) {
            stage_bounds(mi, &rows_query, ranges_on)?;
        }
        // ---- rows in
        upload_and_propagate(mi, false, &ticks, &transforms, &globals.as_readonly(), Some(()))?;
        Ok(())
    })();
    if result.is_err() {
        fallback.transforms = true;
        return;
    }
    if views.is_empty() || views.len() * mi.class_bits.len().max(1) > ffi::MI_RESULTS_MAX_LISTS as usize {
        // nothing to cull (or more lists than one results call takes): the propagate-only path
        let r = (|| -> Result<u32, ()> {
            // Dense windows raise no change marks (they carry every row): when upload_and_propagate took that route the propagate
            // must count every Transform as changed itself -- mi_propagate(0) would find no mark and return without a launch, with the
            // ECS change ticks already consumed.  StaticTransformOptimizations applies here as in the frame call below.
            let propagate_flags = (if mi.every_row_moved { ffi::MI_PROPAGATE_ALL_DIRTY } else { 0 })
                | (if static_opt.as_ref().is_some_and(|s| s.is_enabled()) { ffi::MI_PROPAGATE_STATIC_OPT } else { 0 });
            // SAFETY: plain calls on a live context.
            check(ctx, "mi_propagate", unsafe { ffi::mi_propagate(ctx, propagate_flags) })?;
            let s = &mut mi.scratch;
            s.rows.resize(n as usize, 0);
            s.global12.resize(n as usize * 12, 0.0);
            let mut count = 0u32;
            // SAFETY: both outputs hold `n` entries.
            check(ctx, "mi_download_changed_global_transforms", unsafe {
                ffi::mi_download_changed_global_transforms(ctx, s.rows.as_mut_ptr(), s.global12.as_mut_ptr(), n, &mut count)
            })?;
            Ok(count)
        })();
        match r {
            Ok(count) => write_back_global_transforms(mi, count, &mut globals),
            Err(()) => fallback.transforms = true,
        }
        return;
    }

    // ---- the lights ride along: point, spot and (with storage buffers) rect lights, the gather order of assign.rs:190-248, as rows
    //      of the frame -- the device takes a light's centre (and a spot light's direction) from its row's GlobalTransform and leaves
    //      out the ones whose ViewVisibility::get() is false.  The first clustered camera's walk rides in the frame kernel; every
    //      further one (split screen) is assigned behind the frame through a view slot of its own (mi_cluster_select_view).  Not with
    //      the UBO limit (sort / truncate, assign.rs:297-321) or GPU clustering.
    let mut with_clusters = false;
    let mut cluster_objects: Vec<(Entity, u8)> = Vec::new();
    let mut cluster_views: Vec<ffi::MiClusterView> = Vec::new();
    if let Some(settings) = settings.as_ref() {
        let every_camera_has_a_viewport = cluster_cameras.len() == clustered_cameras;
        if !cluster_cameras.is_empty()
            && every_camera_has_a_viewport
            && cluster_cameras.len() <= ffi::MI_CLUSTER_MAX_VIEWS as usize
            && settings.supports_storage_buffers
            && settings.gpu_clustering.is_none()
            && !fallback.clusters
        {
            let r = (|| -> Result<(), ()> {
                let s = &mut mi.scratch;
                s.obj_pos_range.clear();
                s.obj_layers.clear();
                s.obj_layers_hi.clear();
                s.obj_type.clear();
                s.obj_spot_sin_cos.clear();
                s.rows.clear();
                let mut any_spot = false;
                let mut push = |s: &mut Scratch, e: Entity, range: f32, kind: i32, layers: Option<&RenderLayers>, outer_angle: Option<f32>| -> Result<(), ()> {
                    // (the reference's queries gather this entity, assign.rs:146-178; without a Transform it has no row here: the stock
                    // system takes the clusters of this frame rather than this one leaving the object out)
                    let Some(&row) = mi.entity_row.get(&e) else {
                        error!("bevy_mi355x: a clusterable object without a Transform row -- the stock cluster system takes over");
                        return Err(());
                    };
                    s.obj_pos_range.extend_from_slice(&[0.0, 0.0, 0.0, range]); // the centre is the row's GlobalTransform
                    s.obj_type.push(kind as u8);
                    // the first u64 word of the bitset, as for rows and views (render_layers.rs:121-135); a light above layer 63: stock systems
                    let (lo, hi) = match layers {
                        None => (1, 0),
                        Some(l) => layer_words_or_log(l)?,
                    };
                    s.obj_layers.push(lo);
                    s.obj_layers_hi.push(hi);
                    let (sin, cos) = outer_angle.map_or((0.0, 0.0), ops_sin_cos);
                    s.obj_spot_sin_cos.extend_from_slice(&[sin, cos]);
                    s.rows.push(row);
                    cluster_objects.push((e, kind as u8));
                    Ok(())
                };
                for (e, light, layers) in point_lights.iter() {
                    push(s, e, light.range, ffi::MI_OBJ_POINT_LIGHT, layers, None)?;
                }
                for (e, light, layers) in spot_lights.iter() {
                    any_spot = true;
                    push(s, e, light.range, ffi::MI_OBJ_SPOT_LIGHT, layers, Some(light.outer_angle))?;
                }
                // (rect lights are gathered only where they are clustered at all, assign.rs:231-248: storage buffers, checked above)
                for (e, light, layers) in rect_lights.iter() {
                    push(s, e, light.range, ffi::MI_OBJ_RECT_LIGHT, layers, None)?;
                }
                // light probes (same gate, assign.rs:250-277) and decals (their own, :279-296), RenderLayers::default() both: the range is
                // `radius_vec3a(Vec3A::ONE)` / `scale().length()` of the GlobalTransform this frame gives them (`expected_global`: the
                // same glam calls the reference makes on the same operands); an entity outside the transform table is not a row
                for (e, is_reflection_probe) in light_probes.iter() {
                    let Some(g) = expected_global(e, &transforms) else { return Err(()) }; // (no Transform: as above)
                    let kind = if is_reflection_probe { ffi::MI_OBJ_REFLECTION_PROBE } else { ffi::MI_OBJ_IRRADIANCE_VOLUME };
                    push(s, e, g.radius_vec3a(bevy_math::Vec3A::ONE), kind, None, None)?;
                }
                if settings.clustered_decals_are_usable {
                    for e in decals.iter() {
                        let Some(g) = expected_global(e, &transforms) else { return Err(()) };
                        push(s, e, g.scale().length(), ffi::MI_OBJ_DECAL, None, None)?;
                    }
                }
                if cluster_objects.is_empty() {
                    return Err(());
                }
                let n_obj = cluster_objects.len() as u32;
                // SAFETY: every column holds `n_obj` entries (a spot light's direction comes from its row: no spot_dir column).
                unsafe {
                    check(
                        ctx,
                        "mi_cluster_upload_objects",
                        ffi::mi_cluster_upload_objects(
                            ctx,
                            n_obj,
                            s.obj_pos_range.as_ptr(),
                            s.obj_type.as_ptr(),
                            s.obj_layers.as_ptr(),
                            ptr::null(),
                            s.obj_spot_sin_cos.as_ptr(),
                        ),
                    )?;
                    if s.obj_layers_hi.iter().any(|&w| w != 0) {
                        check(ctx, "mi_cluster_upload_object_layers_hi", ffi::mi_cluster_upload_object_layers_hi(ctx, n_obj, s.obj_layers_hi.as_ptr()))?;
                    }
                    check(ctx, "mi_cluster_bind_objects_to_row_list", ffi::mi_cluster_bind_objects_to_row_list(ctx, n_obj, s.rows.as_ptr()))?;
                }
                // the views: slot k for the k-th clustered camera; every one is resolved against its own history and uploaded now
                // (slot 0 last, so that it is the selected one when the frame call runs)
                for (k, (view_entity, cam_global, planes, size, config, layer_mask, clip)) in cluster_cameras.iter().enumerate().rev() {
                    let config = cluster_config_to_ffi(config);
                    let history = mi.cluster_history.entry(*view_entity).or_insert(ffi::MiClusterHistory {
                        has_farthest_z: 0,
                        farthest_z: 0.0,
                        has_total_cluster_index_count: 0,
                        reserved: 0,
                        total_cluster_index_count: 0,
                    });
                    // SAFETY: zeroed plain structs are valid "empty" values; every pointer below is a live Vec of the stated length.
                    let mut resolved: ffi::MiClusterResolved = unsafe { core::mem::zeroed() };
                    check(ctx, "mi_cluster_config_resolve", unsafe {
                        ffi::mi_cluster_config_resolve(&config, history, size.x, size.y, settings.view_cluster_bindings_max_indices as u64, &mut resolved)
                    })?;
                    if resolved.active == 0 {
                        return Err(()); // Clusters::clear(): the cluster system of its own handles it
                    }
                    let (mut tile, mut dims) = ([0u32; 2], [0u32; 3]);
                    check(ctx, "mi_cluster_view_dims", unsafe {
                        ffi::mi_cluster_view_dims(size.x, size.y, resolved.requested_dims.as_ptr(), tile.as_mut_ptr(), dims.as_mut_ptr())
                    })?;
                    mi.plane_storage.resize(((dims[0] + dims[1] + dims[2] + 3) * 4) as usize, 0.0);
                    // the clusters' bounding spheres are only read by the cone test of spot lights (assign.rs:693-707)
                    mi.sphere_storage.resize(if any_spot { (dims[0] * dims[1] * dims[2] * 4) as usize } else { 0 }, 0.0);
                    let mut view: ffi::MiClusterView = unsafe { core::mem::zeroed() };
                    let camera_affine = cam_global.affine().to_cols_array();
                    unsafe {
                        check(
                            ctx,
                            "mi_cluster_view_build",
                            ffi::mi_cluster_view_build(
                                camera_affine.as_ptr(),
                                clip.as_ptr(),
                                planes.as_ptr(),
                                size.x,
                                size.y,
                                resolved.requested_dims.as_ptr(),
                                resolved.first_slice_depth,
                                resolved.far_z,
                                layer_mask.0,
                                mi.plane_storage.as_mut_ptr(),
                                if any_spot { mi.sphere_storage.as_mut_ptr() } else { ptr::null_mut() },
                                &mut view,
                            ),
                        )?;
                        view.view_layer_mask_hi = layer_mask.1; // (the helper leaves the second word 0)
                        check(ctx, "mi_cluster_select_view", ffi::mi_cluster_select_view(ctx, k as u32))?;
                        check(ctx, "mi_cluster_upload_view", ffi::mi_cluster_upload_view(ctx, &view))?; // (the library copies the tables)
                    }
                    cluster_views.push(view);
                }
                cluster_views.reverse(); // cluster_views[k] belongs to cluster_cameras[k]
                Ok(())
            })();
            with_clusters = r.is_ok();
            if !with_clusters {
                // SAFETY: plain call; slot 0 is what every other system of this plugin expects to find selected.
                let _ = unsafe { ffi::mi_cluster_select_view(ctx, 0) };
            }
        }
    }

    // ---- run + read back: one call each
    let class_bits: Vec<(TypeId, u32)> = mi.class_bits.iter().map(|(k, v)| (*k, *v)).collect();
    let mut lists: Vec<ffi::MiVisibleList> = Vec::new();
    for v in 0..views.len() as u32 {
        for (_, bit) in &class_bits {
            lists.push(ffi::MiVisibleList { view: v, class_bit: *bit, capacity: n, count: 0, rows: ptr::null_mut() });
        }
    }
    let n_clusters = cluster_views.first().map_or(0, |v| v.dims[0] * v.dims[1] * v.dims[2]);
    let mut results = ffi::MiFrameResults {
        flags: ffi::MI_RESULTS_IN_PLACE
            | ffi::MI_RESULTS_CHANGED_ROWS
            | ffi::MI_RESULTS_CHANGED_GLOBALS
            | if with_clusters { ffi::MI_RESULTS_CLUSTERS | ffi::MI_RESULTS_CLUSTER_INDICES } else { 0 },
        n_lists: lists.len() as u32,
        lists: lists.as_mut_ptr(),
        changed_capacity: n,
        reserved: 0,
        cluster_capacity: cluster_objects.len() as u64 * n_clusters as u64, // an object is in a cluster at most once
        changed_rows: ptr::null_mut(),
        changed_global12: ptr::null_mut(),
        cluster_offsets: ptr::null_mut(),
        cluster_counts: ptr::null_mut(),
        cluster_indices: ptr::null_mut(),
        changed_count: 0,
        farthest_z: 0.0,
        cluster_total: 0,
    };
    let static_flag = if static_opt.is_some_and(|s| s.is_enabled()) { ffi::MI_CULL_STATIC_OPT } else { 0 };
    let run: Result<(), ()> = (|| {
        // SAFETY: `views` holds `views.len()` entries; `results` and `lists` are live for the call.
        check(ctx, "mi_propagate_and_cull_views", unsafe {
            ffi::mi_propagate_and_cull_views(
                ctx,
                views.as_ptr(),
                views.len() as u32,
                // (dense windows carried every row: the all-rows frame -- they raise no change marks, and it is the one the library
                // has fetched every GlobalTransform ahead for)
                (if mi.every_row_moved { 0 } else { ffi::MI_CULL_CHANGED_ROWS })
                    | ffi::MI_CULL_END_FRAME
                    | static_flag
                    | if with_clusters { ffi::MI_CULL_WITH_CLUSTERS } else { 0 },
            )
        })?;
        check(ctx, "mi_download_frame_results", unsafe { ffi::mi_download_frame_results(ctx, &mut results) })
    })();
    if run.is_err() {
        fallback.transforms = true; // nothing was written to the ECS; the stock systems compute this frame
        return;
    }

    // ---- everything below reads the library's pinned window in place (valid until the next call on the context)
    // SAFETY: the library filled in pointers to `changed_count` rows / 12 x `changed_count` floats.
    let (rows, g12) = unsafe {
        (
            core::slice::from_raw_parts(results.changed_rows, results.changed_count as usize),
            core::slice::from_raw_parts(results.changed_global12, results.changed_count as usize * 12),
        )
    };
    for (k, row) in rows.iter().enumerate() {
        let cols: &[f32; 12] = g12[k * 12..k * 12 + 12].try_into().unwrap();
        if let Ok(mut global) = globals.get_mut(mi.row_entity[*row as usize]) {
            *global = GlobalTransform::from(Affine3A::from_cols_array(cols)); // listed = the reference would have written it (systems.rs:719)
        }
    }
    for (k, list) in lists.iter().enumerate() {
        let (view, class) = (k / class_bits.len(), class_bits[k % class_bits.len()].0);
        // SAFETY: `count` rows at `rows`.
        let rows = unsafe { core::slice::from_raw_parts(list.rows, list.count as usize) };
        let mut entities: Vec<Entity> = rows.iter().map(|r| mi.row_entity[*r as usize]).collect();
        entities.sort_unstable(); // a no-op for a flat scene (rows are in key order); with a hierarchy the device sorted by key already
        frame_views[view].lists.push((class, entities));
    }
    frame.clusters.clear();
    if with_clusters {
        let c = n_clusters as usize;
        // SAFETY: offsets c + 1, counts 6 c, indices `cluster_total` entries.
        let (offsets, counts, indices) = unsafe {
            (
                core::slice::from_raw_parts(results.cluster_offsets, c + 1).to_vec(),
                core::slice::from_raw_parts(results.cluster_counts, 6 * c).to_vec(),
                core::slice::from_raw_parts(results.cluster_indices, results.cluster_total as usize).to_vec(),
            )
        };
        let view_entity = cluster_cameras[0].0;
        let history = mi.cluster_history.get_mut(&view_entity).unwrap();
        history.has_total_cluster_index_count = 1; // assign.rs:810-811
        history.total_cluster_index_count = results.cluster_total;
        history.has_farthest_z = 1;
        history.farthest_z = results.farthest_z;
        frame.clusters.push(FrameClusters { view_entity, view: cluster_views[0], offsets, counts, indices, objects: cluster_objects.clone() });
        // the other clustered cameras: the same resident objects and ViewVisibility, their own view slot, assigned behind the frame
        let others: Result<(), ()> = (|| {
            for k in 1..cluster_cameras.len() {
                let view = cluster_views[k];
                let c = (view.dims[0] * view.dims[1] * view.dims[2]) as usize;
                let (mut offsets, mut counts) = (vec![0u32; c + 1], vec![0u32; 6 * c]);
                let mut indices = vec![0u32; cluster_objects.len() * c];
                let (mut total, mut farthest_z) = (0u64, 0f32);
                // SAFETY: plain calls on a live context; the outputs hold what the header asks for.
                unsafe {
                    check(ctx, "mi_cluster_select_view", ffi::mi_cluster_select_view(ctx, k as u32))?;
                    check(ctx, "mi_cluster_assign_resident", ffi::mi_cluster_assign_resident(ctx, ptr::null_mut()))?;
                    check(
                        ctx,
                        "mi_cluster_download",
                        ffi::mi_cluster_download(ctx, offsets.as_mut_ptr(), indices.as_mut_ptr(), indices.len() as u64, counts.as_mut_ptr(), &mut total, &mut farthest_z),
                    )?;
                }
                indices.truncate(total as usize);
                let view_entity = cluster_cameras[k].0;
                let history = mi.cluster_history.get_mut(&view_entity).unwrap();
                history.has_total_cluster_index_count = 1;
                history.total_cluster_index_count = total;
                history.has_farthest_z = 1;
                history.farthest_z = farthest_z;
                frame.clusters.push(FrameClusters { view_entity, view, offsets, counts, indices, objects: cluster_objects.clone() });
            }
            Ok(())
        })();
        // SAFETY: plain call.
        let _ = unsafe { ffi::mi_cluster_select_view(ctx, 0) };
        frame.clusters_valid = others.is_ok(); // (else: `mi_assign_objects_to_clusters` runs for every view this frame)
    }
    frame.views = frame_views;
    frame.submit_tick = ticks.this_run();
    frame.valid = true;
}

/// Stages the bounds / flags / layers / class columns from the ECS (shared by [`mi_check_visibility`] and [`mi_fused_frame`]).
#[allow(clippy::type_complexity)]
fn stage_bounds(mi: &mut Mi355x, rows_query: &RowsQuery, ranges_resource: bool) -> Result<(), ()> {
    let ctx = mi.ctx;
    let n = mi.row_entity.len();
    let s = &mut mi.scratch;
    for (v, w) in [(&mut s.aabb_center, 3), (&mut s.aabb_half, 3)] {
        v.clear();
        v.resize(n * w, 0.0);
    }
    s.flags.clear();
    s.flags.resize(n, 0);
    s.layers.clear();
    s.layers.resize(n, 0);
    s.layers_hi.clear();
    s.layers_hi.resize(n, 0);
    s.classes.clear();
    s.classes.resize(n, 0);
    s.ranges.clear();
    s.ranges.resize(n * 2, 0.0);
    for (entity, inherited, classes, layers, aabb, sphere, point_light, spot_light, no_frustum_culling, range, mesh3d, not_shadow_caster, directional_light) in
        rows_query.iter()
    {
        let Some(&row) = mi.entity_row.get(&entity) else { continue };
        let row = row as usize;
        let mut flags = 0u32;
        if inherited.get() {
            flags |= ffi::MI_FLAG_INHERITED_VISIBLE;
        }
        if no_frustum_culling {
            flags |= ffi::MI_FLAG_NO_FRUSTUM_CULLING;
        }
        if let Some(range) = range {
            // Has<VisibilityRange> (visibility/mod.rs:814); the two bounds is_visible_at_all reads (range.rs:159-161) and use_aabb
            // (range.rs:255-263: the model position is the Aabb centre in world space when the entity has one)
            flags |= ffi::MI_FLAG_HAS_VISIBILITY_RANGE;
            if range.use_aabb {
                flags |= ffi::MI_FLAG_RANGE_USE_AABB;
            }
            s.ranges[row * 2] = range.start_margin.start;
            s.ranges[row * 2 + 1] = range.end_margin.end;
        }
        if mesh3d && !not_shadow_caster && !directional_light {
            flags |= ffi::MI_FLAG_SHADOW_CASTER; // the shadow views' query filter, crates/bevy_light/src/lib.rs:355-372
        }
        if let Some(aabb) = aabb {
            flags |= ffi::MI_FLAG_HAS_AABB;
            s.aabb_center[row * 3..row * 3 + 3].copy_from_slice(&aabb.center.to_array());
            s.aabb_half[row * 3..row * 3 + 3].copy_from_slice(&aabb.half_extents.to_array());
        } else if let Some(range) = point_light.map(|l| l.range).or(spot_light.map(|l| l.range)) {
            // update_point_light_bounding_spheres / update_spot_light_bounding_spheres keep Sphere { GlobalTransform::translation,
            // range } on every point and spot light (point_light.rs:195-208, spot_light.rs:221-234, inserted through Commands): on
            // the device the sphere follows the row's own GlobalTransform, so a moving light needs no bounds upload and is never
            // culled against last frame's position
            flags |= ffi::MI_FLAG_HAS_SPHERE;
            s.aabb_half[row * 3] = range;
            s.aabb_half[row * 3 + 1] = f32::from_bits(ffi::MI_SPHERE_AT_TRANSLATION);
        } else if let Some(sphere) = sphere {
            flags |= ffi::MI_FLAG_HAS_SPHERE; // any other world-space Sphere, as it is (visibility/mod.rs:838-843)
            s.aabb_center[row * 3..row * 3 + 3].copy_from_slice(&sphere.center.to_array());
            s.aabb_half[row * 3] = sphere.radius;
        }
        s.flags[row] = flags as u8;
        (s.layers[row], s.layers_hi[row]) = match layers {
            None => (1, 0), // RenderLayers::default() == layer 0
            Some(l) => layer_words_or_log(l)?,
        };
        s.any_layers_hi |= s.layers_hi[row] != 0;
        if let Some(classes) = classes {
            for class in classes.iter() {
                s.classes[row] |= 1 << mi_class_bit(&mut mi.class_bits, *class)
                    .ok_or_else(|| error!("bevy_mi355x: more than 32 visibility classes; the class column holds 32 -- falling back to the CPU systems"))?;
            }
        }
    }
    // SAFETY: every column holds `n` rows.
    unsafe {
        check(
            ctx,
            "mi_upload_bounds",
            ffi::mi_upload_bounds(ctx, 0, n as u32, s.aabb_center.as_ptr(), s.aabb_half.as_ptr(), s.flags.as_ptr(), s.layers.as_ptr()),
        )?;
        if s.any_layers_hi {
            // (once a layer above 31 was seen the column stays in use: a row that leaves it must be written back to 0)
            check(ctx, "mi_upload_render_layers_hi", ffi::mi_upload_render_layers_hi(ctx, 0, n as u32, s.layers_hi.as_ptr()))?;
        }
        check(ctx, "mi_upload_visibility_classes", ffi::mi_upload_visibility_classes(ctx, 0, n as u32, s.classes.as_ptr()))?;
        // Option<Res<VisibleEntityRanges>>: no resource = no range column, and no VisibilityRange hides anything (visibility/mod.rs:814-816)
        check(
            ctx,
            "mi_upload_visibility_ranges",
            ffi::mi_upload_visibility_ranges(ctx, 0, n as u32, if ranges_resource { s.ranges.as_ptr() } else { ptr::null() }),
        )?;
    }
    mi.ranges_resource = Some(ranges_resource);
    Ok(())
}

/// `VisibilitySystems::CheckVisibility`, fused form: the parked lists become `set_visible()` calls and `VisibleEntities`.  No
/// device call.  Drops the parked results (-> [`mi_check_visibility`] runs next) when an input of the cull was written after the submit.
pub fn mi_apply_visibility(
    mut frame: ResMut<Mi355xFrame>,
    ticks: SystemChangeTick,
    mut view_query: Query<(&mut VisibleEntities, &Frustum)>,
    inputs: Query<(Ref<InheritedVisibility>, Option<Ref<Aabb>>, Option<Ref<Sphere>>, Option<Ref<RenderLayers>>, Option<Ref<VisibilityRange>>), Without<NoCpuCulling>>,
    mut view_visibilities: Query<&mut ViewVisibility, Without<NoCpuCulling>>,
) {
    #[cfg(feature = "trace")]
    let _span = info_span!("mi_apply_visibility").entered();
    if !frame.valid {
        return;
    }
    let written_since = |t: Tick| t.is_newer_than(frame.submit_tick, ticks.this_run());
    let stale_inputs = inputs.iter().any(|(inherited, aabb, sphere, layers, range)| {
        written_since(inherited.last_changed())
            || aabb.is_some_and(|a| written_since(a.last_changed()))
            || sphere.is_some_and(|s| written_since(s.last_changed()))
            || layers.is_some_and(|l| written_since(l.last_changed()))
            || range.is_some_and(|r| written_since(r.last_changed()))
    });
    let stale_frusta = frame.views.iter().any(|v| {
        view_query.get(v.entity).map_or(true, |(_, frustum)| {
            frustum.half_spaces.iter().enumerate().any(|(p, h)| h.normal_d().to_array().map(f32::to_bits) != [0, 1, 2, 3].map(|k| v.frustum[p * 4 + k].to_bits()))
        })
    });
    if stale_inputs || stale_frusta {
        frame.valid = false; // mi_check_visibility, chained right behind, culls this frame with the inputs as they are now
        return;
    }
    for view in &frame.views {
        let Ok((mut visible_entities, _)) = view_query.get_mut(view.entity) else { continue };
        visible_entities.clear_all();
        for (class, entities) in &view.lists {
            for entity in entities {
                if let Ok(mut view_visibility) = view_visibilities.get_mut(*entity) {
                    view_visibility.set_visible(); // mod.rs:846: the tick moves only on hidden -> visible (:290-306)
                }
            }
            visible_entities.get_mut(*class).extend_from_slice(entities); // sorted already (mod.rs:872-875)
        }
    }
}

/// In front of `SimulationLightSystems::AssignLightsToClusters`, fused form: the parked cluster lists become `Clusters`.
pub fn mi_apply_clusters(mi: Res<Mi355x>, mut frame: ResMut<Mi355xFrame>, mut views: Query<&mut Clusters>) {
    #[cfg(feature = "trace")]
    let _span = info_span!("mi_apply_clusters").entered();
    if !frame.valid {
        frame.clusters_valid = false; // the lights' ViewVisibility was decided by the cull that has just been dropped
    }
    if !frame.clusters_valid {
        return;
    }
    for r in frame.clusters.iter() {
        let Ok(mut clusters) = views.get_mut(r.view_entity) else { continue };
        let history = mi.cluster_history[&r.view_entity];
        clusters.tile_size = UVec2::from_array(r.view.tile_size);
        clusters.dimensions = UVec3::from_array(r.view.dims);
        clusters.near = r.view.near_;
        clusters.far = r.view.far_;
        clusters.last_frame_farthest_z = (history.has_farthest_z != 0).then_some(history.farthest_z);
        clusters.last_frame_total_cluster_index_count =
            (history.has_total_cluster_index_count != 0).then_some(history.total_cluster_index_count as usize);
        let n_clusters = (r.view.dims[0] * r.view.dims[1] * r.view.dims[2]) as usize;
        let mut per_cluster: Vec<ObjectsInClusterCpu> = Vec::with_capacity(n_clusters);
        for c in 0..n_clusters {
            let mut objects = ObjectsInClusterCpu::default();
            // (a cluster's list is in gather order -- points, spots, rects, probes, decals -- like the reference's pushes: assign.rs:740-800)
            for &object in &r.indices[r.offsets[c] as usize..r.offsets[c + 1] as usize] {
                let (entity, kind) = r.objects[object as usize];
                match kind as i32 {
                    ffi::MI_OBJ_POINT_LIGHT => objects.add_point_light(entity),
                    ffi::MI_OBJ_SPOT_LIGHT => objects.add_spot_light(entity),
                    ffi::MI_OBJ_RECT_LIGHT => objects.add_rect_light(entity),
                    ffi::MI_OBJ_REFLECTION_PROBE => objects.add_reflection_probe(entity),
                    ffi::MI_OBJ_IRRADIANCE_VOLUME => objects.add_irradiance_volume(entity),
                    _ => objects.add_decal(entity),
                }
            }
            debug_assert_eq!(objects.counts.point_lights, r.counts[c * 6]);
            debug_assert_eq!(objects.counts.spot_lights, r.counts[c * 6 + 1]);
            per_cluster.push(objects);
        }
        clusters.clusterable_objects = ClusterableObjects::Cpu(per_cluster);
    }
}

/// `bevy_math::ops::sin_cos` -- the libm the reference is built with decides the last bit of a spot light's cone; the column
/// takes the host's values so that the device never evaluates sin/cos itself (include/bevy_mi355x.h, `spot_sin_cos`).
fn ops_sin_cos(angle: f32) -> (f32, f32) {
    bevy_math::ops::sin_cos(angle)
}
