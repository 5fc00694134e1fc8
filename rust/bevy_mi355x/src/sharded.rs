//! The plugin over SEVERAL GPUs of one node: `Mi355xRenderPrepPlugin { devices: vec![0, 1, .., 7], .. }`.
//!
//! A Bevy `App` is one `World` in one process, so the shard of north_star -- "entity ranges shard across the 8 GPUs of one node with
//! an RCCL all-gather of the packed ViewVisibility bitmask" -- is driven from the ONE thread the system runs on: a library context per
//! device, rows (in `Entity` order) cut into contiguous 256-aligned ranges, and per frame
//!
//! ```text
//! mi_sharded_frame  (in TransformSystems::Propagate)
//!     per shard: the rows a Changed<Transform> query yields      -> mi_upload_transforms_indexed
//!     per shard: ONE frame call (enqueue only: the devices run side by side)
//!                                                                -> mi_propagate_and_cull_views(CHANGED_ROWS | END_FRAME)
//!     the N in-place all-gathers, together                       -> mi_exchange_group_flush(ncclGroupStart, ncclGroupEnd)
//!     per shard: its changed GlobalTransforms (they stay sharded) -> mi_download_changed_global_transforms -> Mut<GlobalTransform>
//!     ONE copy of ONE context's gathered buffer: every shard's masks of every view -> mi_exchange_download
//!     -> per view and class the VisibleEntities lists, parked in Mi355xFrame
//! mi_apply_visibility  (in VisibilitySystems::CheckVisibility)     set_visible() + VisibleEntities, as in the fused single-GPU form
//! ```
//!
//! `bevy_amd/host/bevy_mi355x_sharded.hpp` (`Mi355xShardedPlugin`) is the same design in C++, compiled and run against the
//! single-device plugin on twin Worlds (`tests/cpp/host_systems_test.cpp: sharded_plugin_leaves_the_same_world`, device lists `{0}`,
//! `{0, 0, 0}` and every GPU of the node).
//!
//! Scope: flat Worlds (no `ChildOf`: configs[1] / configs[3]).  A World with a hierarchy shards by root subtree
//! (`bevy_amd/sharding.py: shard_hierarchy`); this system hands such a frame -- and every frame after a device error -- to the stock
//! systems through [`CpuFallback`], like every other system of the crate.  Light clusters and shadow views stay with the stock systems
//! in this form (clusters shard over lights, optional at 100 k; batching is replicas only).  RCCL is loaded with
//! `dlopen` -- no link-time dependency -- and the communicators come from `ncclCommInitAll` over the device list.

use core::ffi::{c_char, c_int, c_void};
use core::ptr;

use bevy_camera::{
    visibility::{NoCpuCulling, RenderLayers, VisibleEntityRanges},
    Camera,
};
use bevy_ecs::{
    entity::{Entity, EntityHashMap},
    prelude::*,
    system::SystemChangeTick,
};
use bevy_log::error;
use bevy_math::Affine3A;
use bevy_transform::components::{GlobalTransform, Transform};
use core::any::TypeId;

use crate::{ffi, layer_words_or_log, mi_class_bit, range_view_table, CpuFallback, FrameView, Mi355xFrame, RangeViews, RowsQuery};

unsafe extern "C" {
    fn dlopen(filename: *const c_char, flags: c_int) -> *mut c_void;
    fn dlsym(handle: *mut c_void, symbol: *const c_char) -> *mut c_void;
}
const RTLD_NOW: c_int = 2;
const RTLD_GLOBAL: c_int = 0x100;

/// One context per device, the communicators, and the Entity <-> (shard, row) tables.
#[derive(Resource)]
pub struct Mi355xShards {
    ctxs: Vec<*mut ffi::MiCtx>,
    comms: Vec<*mut c_void>,
    all_gather: *mut c_void,
    group_start: *mut c_void,
    group_end: *mut c_void,
    /// Global row -> entity, rows in `Entity` order; shard `d` holds rows `[d * rows_per, d * rows_per + cnt[d])`.
    row_entity: Vec<Entity>,
    entity_row: EntityHashMap<u32>,
    rows_per: u32,
    cnt: Vec<u32>,
    /// Number of views the gathered buffers are laid out for (`u32::MAX`: not yet).
    exchange_views: u32,
    class_bits: bevy_platform::collections::HashMap<TypeId, u32>,
    classes: Vec<u32>,
    ranges_resource: Option<bool>,
    masks: Vec<u64>,
}
// SAFETY: used through `ResMut`, i.e. by one system at a time -- the library's contract per context, and RCCL's for a group call.
unsafe impl Send for Mi355xShards {}
unsafe impl Sync for Mi355xShards {}

impl Mi355xShards {
    pub fn new(devices: &[i32]) -> Result<Self, String> {
        let mut ctxs = Vec::new();
        // SAFETY: plain FFI; out pointers are live locals; strings are NUL-terminated literals.
        unsafe {
            if ffi::mi_abi_version() != ffi::MI_ABI_VERSION {
                return Err("libbevy_mi355x: ABI version mismatch".into());
            }
            for &device in devices {
                let mut ctx = ptr::null_mut();
                let status = ffi::mi_ctx_create(device, ptr::null_mut(), &mut ctx);
                if status != ffi::MI_OK {
                    for c in ctxs {
                        ffi::mi_ctx_destroy(c);
                    }
                    return Err(format!("mi_ctx_create({device}) failed with status {status}"));
                }
                ctxs.push(ctx);
            }
            let mut lib = ptr::null_mut();
            for path in [c"librccl.so".as_ptr(), c"/opt/rocm/lib/librccl.so".as_ptr()] {
                if lib.is_null() {
                    lib = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
                }
            }
            if lib.is_null() {
                return Err("librccl.so could not be loaded".into());
            }
            let init_all = dlsym(lib, c"ncclCommInitAll".as_ptr());
            let all_gather = dlsym(lib, c"ncclAllGather".as_ptr());
            let group_start = dlsym(lib, c"ncclGroupStart".as_ptr());
            let group_end = dlsym(lib, c"ncclGroupEnd".as_ptr());
            if init_all.is_null() || all_gather.is_null() || group_start.is_null() || group_end.is_null() {
                return Err("RCCL symbols missing".into());
            }
            let init_all: unsafe extern "C" fn(*mut *mut c_void, c_int, *const c_int) -> c_int = core::mem::transmute(init_all);
            let mut comms = vec![ptr::null_mut(); devices.len()];
            if init_all(comms.as_mut_ptr(), devices.len() as c_int, devices.as_ptr()) != 0 {
                return Err("ncclCommInitAll failed".into());
            }
            Ok(Self {
                ctxs,
                comms,
                all_gather,
                group_start,
                group_end,
                row_entity: Vec::new(),
                entity_row: EntityHashMap::default(),
                rows_per: 0,
                cnt: Vec::new(),
                exchange_views: u32::MAX,
                class_bits: Default::default(),
                classes: Vec::new(),
                ranges_resource: None,
                masks: Vec::new(),
            })
        }
    }
}

impl Drop for Mi355xShards {
    fn drop(&mut self) {
        // SAFETY: every context came from mi_ctx_create and is destroyed once; a NULL communicator switches the exchange off first.
        unsafe {
            for &ctx in &self.ctxs {
                ffi::mi_exchange_configure(ctx, ptr::null_mut(), ptr::null_mut(), ptr::null(), 0, 0, 0, 0, 0);
                ffi::mi_ctx_destroy(ctx);
            }
        }
    }
}

fn check(ctx: *mut ffi::MiCtx, what: &str, status: i32) -> Result<(), ()> {
    crate::check(ctx, what, status)
}

/// The whole sharded frame (module docs).  Runs in `TransformSystems::Propagate`, in place of [`crate::mi_fused_frame`].
#[allow(clippy::too_many_arguments, clippy::type_complexity)]
pub fn mi_sharded_frame(
    mut shards: ResMut<Mi355xShards>,
    mut fallback: ResMut<CpuFallback>,
    mut frame: ResMut<Mi355xFrame>,
    ticks: SystemChangeTick,
    structure_changed: Query<(), Or<(Added<Transform>, Changed<ChildOf>)>>,
    mut despawned: RemovedComponents<Transform>,
    hierarchy: Query<(), With<ChildOf>>,
    transforms: Query<(Entity, Ref<Transform>)>,
    mut globals: Query<&mut GlobalTransform>,
    cameras: Query<(Entity, &Camera, &bevy_camera::Projection, Option<&RenderLayers>, Has<NoCpuCulling>)>,
    bounds_changed: crate::BoundsChanged,
    rows_query: RowsQuery,
    (range_views, visible_entity_ranges): (RangeViews, Option<Res<VisibleEntityRanges>>),
) {
    #[cfg(feature = "trace")]
    let _span = bevy_log::info_span!("mi_sharded_frame").entered();
    frame.valid = false;
    frame.clusters_valid = false;
    if fallback.transforms {
        return;
    }
    if !hierarchy.is_empty() {
        error!("bevy_mi355x: the multi-GPU form shards flat Worlds by row range; this World has ChildOf -- the stock systems take over");
        fallback.transforms = true;
        return;
    }
    let sh = &mut *shards;
    let n_shards = sh.ctxs.len();
    let rebuild = !structure_changed.is_empty() || despawned.read().count() != 0 || sh.cnt.is_empty();
    let ranges_on = visible_entity_ranges.is_some();

    let result: Result<(), ()> = (|| {
        // ---- structure: rows in Entity order, cut into contiguous ranges of a multiple of 256 rows (a mask word never straddles shards)
        if rebuild {
            let mut entities: Vec<Entity> = transforms.iter().map(|(e, _)| e).collect();
            entities.sort_unstable();
            let n = entities.len() as u32;
            sh.rows_per = ((n + n_shards as u32 - 1) / n_shards as u32 + 255) / 256 * 256;
            if sh.rows_per == 0 {
                sh.rows_per = 256;
            }
            sh.entity_row.clear();
            for (row, e) in entities.iter().enumerate() {
                sh.entity_row.insert(*e, row as u32);
            }
            sh.cnt = (0..n_shards as u32).map(|d| n.saturating_sub(d * sh.rows_per).min(sh.rows_per)).collect();
            for d in 0..n_shards {
                let (ctx, lo, m) = (sh.ctxs[d], d as u32 * sh.rows_per, sh.cnt[d]);
                // SAFETY: plain call on a live context.
                check(ctx, "mi_columns_resize", unsafe { ffi::mi_columns_resize(ctx, m) })?;
                if m == 0 {
                    continue;
                }
                let (mut t, mut r, mut s, mut g, mut keys) = (Vec::new(), Vec::new(), Vec::new(), Vec::new(), Vec::new());
                for e in &entities[lo as usize..(lo + m) as usize] {
                    let (_, tr) = transforms.get(*e).map_err(|_| ())?;
                    t.extend_from_slice(&tr.translation.to_array());
                    r.extend_from_slice(&tr.rotation.to_array());
                    s.extend_from_slice(&tr.scale.to_array());
                    let old = globals.get(*e).map(|x| x.affine().to_cols_array()).unwrap_or(Affine3A::IDENTITY.to_cols_array());
                    g.extend_from_slice(&old);
                    keys.push(e.to_bits());
                }
                // SAFETY: every column holds `m` rows.
                unsafe {
                    check(ctx, "mi_upload_transforms", ffi::mi_upload_transforms(ctx, 0, m, t.as_ptr(), r.as_ptr(), s.as_ptr()))?;
                    check(ctx, "mi_upload_global_transforms", ffi::mi_upload_global_transforms(ctx, 0, m, g.as_ptr()))?;
                    check(ctx, "mi_upload_entity_keys", ffi::mi_upload_entity_keys(ctx, 0, m, keys.as_ptr()))?;
                    // (new rows count as changed -- Added<GlobalTransform> -- until their first propagate: no change column yet)
                }
            }
            sh.row_entity = entities;
            sh.exchange_views = u32::MAX;
            sh.ranges_resource = None;
        }
        let n = sh.row_entity.len() as u32;

        // ---- the columns that change rarely, per shard (flags, bounds, layers, classes, ranges: what stage_bounds stages for one context)
        if rebuild || !bounds_changed.is_empty() || sh.ranges_resource != Some(ranges_on) {
            let nn = n as usize;
            let (mut center, mut half, mut flags, mut layers, mut layers_hi, mut ranges) =
                (vec![0f32; nn * 3], vec![0f32; nn * 3], vec![0u8; nn], vec![0u32; nn], vec![0u32; nn], vec![0f32; nn * 2]);
            sh.classes.clear();
            sh.classes.resize(nn, 0);
            let mut any_hi = false;
            for (entity, inherited, classes, row_layers, aabb, sphere, _point, _spot, no_frustum_culling, range, _mesh, _nsc, _dl) in rows_query.iter() {
                let Some(&row) = sh.entity_row.get(&entity) else { continue };
                let row = row as usize;
                let mut fl = 0u32;
                if inherited.get() {
                    fl |= ffi::MI_FLAG_INHERITED_VISIBLE;
                }
                if no_frustum_culling {
                    fl |= ffi::MI_FLAG_NO_FRUSTUM_CULLING;
                }
                if let Some(range) = range {
                    fl |= ffi::MI_FLAG_HAS_VISIBILITY_RANGE | if range.use_aabb { ffi::MI_FLAG_RANGE_USE_AABB } else { 0 };
                    ranges[row * 2] = range.start_margin.start;
                    ranges[row * 2 + 1] = range.end_margin.end;
                }
                if let Some(aabb) = aabb {
                    fl |= ffi::MI_FLAG_HAS_AABB;
                    center[row * 3..row * 3 + 3].copy_from_slice(&aabb.center.to_array());
                    half[row * 3..row * 3 + 3].copy_from_slice(&aabb.half_extents.to_array());
                } else if let Some(sphere) = sphere {
                    fl |= ffi::MI_FLAG_HAS_SPHERE;
                    center[row * 3..row * 3 + 3].copy_from_slice(&sphere.center.to_array());
                    half[row * 3] = sphere.radius;
                }
                flags[row] = fl as u8;
                (layers[row], layers_hi[row]) = match row_layers {
                    None => (1, 0),
                    Some(l) => layer_words_or_log(l)?,
                };
                any_hi |= layers_hi[row] != 0;
                if let Some(classes) = classes {
                    for class in classes.iter() {
                        sh.classes[row] |= 1 << mi_class_bit(&mut sh.class_bits, *class).ok_or(())?;
                    }
                }
            }
            for d in 0..n_shards {
                let (ctx, lo, m) = (sh.ctxs[d], (d as u32 * sh.rows_per) as usize, sh.cnt[d]);
                if m == 0 {
                    continue;
                }
                // SAFETY: every slice starts at the shard's first row and holds at least `m` rows.
                unsafe {
                    check(
                        ctx,
                        "mi_upload_bounds",
                        ffi::mi_upload_bounds(ctx, 0, m, center[lo * 3..].as_ptr(), half[lo * 3..].as_ptr(), flags[lo..].as_ptr(), layers[lo..].as_ptr()),
                    )?;
                    if any_hi {
                        check(ctx, "mi_upload_render_layers_hi", ffi::mi_upload_render_layers_hi(ctx, 0, m, layers_hi[lo..].as_ptr()))?;
                    }
                    check(ctx, "mi_upload_visibility_classes", ffi::mi_upload_visibility_classes(ctx, 0, m, sh.classes[lo..].as_ptr()))?;
                    check(
                        ctx,
                        "mi_upload_visibility_ranges",
                        ffi::mi_upload_visibility_ranges(ctx, 0, m, if ranges_on { ranges[lo * 2..].as_ptr() } else { ptr::null() }),
                    )?;
                }
            }
            sh.ranges_resource = Some(ranges_on);
        }

        // ---- rows in: Changed<Transform>, by shard
        if !rebuild {
            let mut rows: Vec<Vec<u32>> = vec![Vec::new(); n_shards];
            let mut trs: Vec<(Vec<f32>, Vec<f32>, Vec<f32>)> = vec![Default::default(); n_shards];
            for (entity, tr) in transforms.iter() {
                if !tr.last_changed().is_newer_than(ticks.last_run(), ticks.this_run()) {
                    continue;
                }
                let Some(&row) = sh.entity_row.get(&entity) else { continue };
                let d = (row / sh.rows_per) as usize;
                rows[d].push(row - d as u32 * sh.rows_per);
                trs[d].0.extend_from_slice(&tr.translation.to_array());
                trs[d].1.extend_from_slice(&tr.rotation.to_array());
                trs[d].2.extend_from_slice(&tr.scale.to_array());
            }
            for d in 0..n_shards {
                if sh.cnt[d] == 0 {
                    continue;
                }
                let ctx = sh.ctxs[d];
                // SAFETY: the three value arrays hold rows[d].len() entries each.
                unsafe {
                    check(
                        ctx,
                        "mi_upload_transforms_indexed",
                        ffi::mi_upload_transforms_indexed(ctx, rows[d].len() as u32, rows[d].as_ptr(), trs[d].0.as_ptr(), trs[d].1.as_ptr(), trs[d].2.as_ptr()),
                    )?;
                    if rows[d].is_empty() {
                        // "nothing changed" must stay distinct from "no change information" (= every row counts as changed)
                        let zero = 0u8;
                        check(ctx, "mi_upload_changed", ffi::mi_upload_changed(ctx, 0, 1, &zero))?;
                    }
                }
            }
        }

        // ---- the frame's views: active cameras in query order, frusta as update_frusta WILL compute them (flat World: GlobalTransform =
        //      From(Transform) of the camera, the same value propagate is about to write)
        let range_table = if ranges_on { range_view_table(&range_views) } else { EntityHashMap::default() };
        let mut views: Vec<ffi::MiView> = Vec::new();
        let mut frame_views: Vec<FrameView> = Vec::new();
        for (entity, camera, projection, layers, no_cpu_culling) in cameras.iter() {
            if !camera.is_active {
                continue;
            }
            let (_, tr) = transforms.get(entity).map_err(|_| error!("bevy_mi355x: an active camera without a Transform"))?;
            let global = GlobalTransform::from(*tr);
            let frustum = projection.compute_frustum(&global);
            let mut planes = [0f32; 24];
            for (p, half_space) in frustum.half_spaces.iter().enumerate() {
                planes[p * 4..p * 4 + 4].copy_from_slice(&half_space.normal_d().to_array());
            }
            let (layer_mask, layer_mask_hi) = match layers {
                None => (1, 0),
                Some(l) => layer_words_or_log(l)?,
            };
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
        }
        let n_views = views.len() as u32;
        let words_per_view = (sh.rows_per / 64) as u64;

        // ---- the gathered buffers: [rank][view][word], laid out again when the shard size or the number of views changes
        if n_views != 0 && sh.exchange_views != n_views {
            let block = n_views as u64 * words_per_view * 8;
            for d in 0..n_shards {
                let ctx = sh.ctxs[d];
                // SAFETY: one communicator per context; the library allocates and owns the buffers.
                unsafe {
                    check(ctx, "mi_exchange_configure", ffi::mi_exchange_configure(ctx, ptr::null_mut(), ptr::null_mut(), ptr::null(), 0, 0, 0, 0, 0))?;
                    check(ctx, "mi_exchange_set_mode", ffi::mi_exchange_set_mode(ctx, ffi::MI_EXCHANGE_GROUPED))?;
                    check(
                        ctx,
                        "mi_exchange_configure_owned",
                        ffi::mi_exchange_configure_owned(
                            ctx,
                            &sh.comms[d],
                            1,
                            sh.all_gather,
                            3,
                            n_shards as u32,
                            words_per_view,
                            d as u64 * n_views as u64 * words_per_view,
                            block,
                            d as u32,
                        ),
                    )?;
                }
            }
            sh.exchange_views = n_views;
        }

        // ---- run: every context's frame call (enqueue only), then the all-gathers of all of them in one group
        for d in 0..n_shards {
            let ctx = sh.ctxs[d];
            // SAFETY: `views` holds n_views entries.
            unsafe {
                if n_views != 0 {
                    check(
                        ctx,
                        "mi_propagate_and_cull_views",
                        ffi::mi_propagate_and_cull_views(ctx, views.as_ptr(), n_views, ffi::MI_CULL_CHANGED_ROWS | ffi::MI_CULL_END_FRAME),
                    )?;
                } else if sh.cnt[d] != 0 {
                    check(ctx, "mi_propagate", ffi::mi_propagate(ctx, 0))?;
                }
            }
        }
        if n_views != 0 {
            // SAFETY: the contexts are live and all in MI_EXCHANGE_GROUPED mode; the two addresses come from the RCCL the communicators belong to.
            check(sh.ctxs[0], "mi_exchange_group_flush", unsafe {
                ffi::mi_exchange_group_flush(sh.ctxs.as_ptr(), n_shards as u32, sh.group_start, sh.group_end)
            })?;
        }

        // ---- out: each shard's changed GlobalTransforms ...
        let (mut crow, mut cg) = (Vec::new(), Vec::new());
        let mut written: Vec<(Entity, [f32; 12])> = Vec::new();
        for d in 0..n_shards {
            let (ctx, m) = (sh.ctxs[d], sh.cnt[d]);
            if m == 0 {
                continue;
            }
            crow.resize(m as usize, 0u32);
            cg.resize(m as usize * 12, 0f32);
            let mut count = 0u32;
            // SAFETY: both outputs hold `m` entries.
            check(ctx, "mi_download_changed_global_transforms", unsafe {
                ffi::mi_download_changed_global_transforms(ctx, crow.as_mut_ptr(), cg.as_mut_ptr(), m, &mut count)
            })?;
            for k in 0..count as usize {
                let cols: [f32; 12] = cg[k * 12..k * 12 + 12].try_into().unwrap();
                written.push((sh.row_entity[(d as u32 * sh.rows_per + crow[k]) as usize], cols));
            }
        }
        // ... and every shard's masks of every view, from ONE context's gathered buffer
        if n_views != 0 {
            sh.masks.clear();
            sh.masks.resize(n_shards * n_views as usize * words_per_view as usize, 0);
            // SAFETY: the buffer holds world x block_bytes bytes.
            check(sh.ctxs[0], "mi_exchange_download", unsafe {
                ffi::mi_exchange_download(sh.ctxs[0], sh.masks.as_mut_ptr() as *mut c_void, sh.masks.len() as u64 * 8)
            })?;
        }

        // ---- every library call has succeeded: the ECS writes.  GlobalTransform at once (listed = the reference would have written it,
        //      systems.rs:62), the lists parked for mi_apply_visibility.
        for (entity, cols) in written {
            if let Ok(mut global) = globals.get_mut(entity) {
                *global = GlobalTransform::from(Affine3A::from_cols_array(&cols));
            }
        }
        let class_bits: Vec<(TypeId, u32)> = sh.class_bits.iter().map(|(k, v)| (*k, *v)).collect();
        for (v, fv) in frame_views.iter_mut().enumerate() {
            let mut per_class: Vec<Vec<Entity>> = vec![Vec::new(); class_bits.len()];
            for d in 0..n_shards {
                let base = (d * n_views as usize + v) * words_per_view as usize;
                for k in 0..((sh.cnt[d] + 63) / 64) as usize {
                    let mut bits = sh.masks[base + k];
                    while bits != 0 {
                        let local = k as u32 * 64 + bits.trailing_zeros();
                        bits &= bits - 1;
                        if local >= sh.cnt[d] {
                            break;
                        }
                        let row = (d as u32 * sh.rows_per + local) as usize;
                        for (c, (_, bit)) in class_bits.iter().enumerate() {
                            if sh.classes[row] & (1 << bit) != 0 {
                                per_class[c].push(sh.row_entity[row]); // rows ascend = Entity order: the lists come out sorted (mod.rs:872-875)
                            }
                        }
                    }
                }
            }
            // (as in the fused single-GPU frame: set_visible() is applied to the union of the lists -- an entity without a VisibilityClass,
            // which the reference marks visible without listing it, mod.rs:846-857, is the one case the parked lists do not carry)
            fv.lists = class_bits.iter().map(|(class, _)| *class).zip(per_class).collect();
        }
        frame.views = frame_views;
        Ok(())
    })();
    if result.is_err() {
        // (nothing was written to the ECS before the first failing call returned -- the writes are the last step above)
        fallback.transforms = true;
        return;
    }
    frame.submit_tick = ticks.this_run();
    frame.valid = true;
}
