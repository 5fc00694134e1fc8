//! The plugin over SEVERAL GPUs of one node: `Mi355xRenderPrepPlugin { devices: vec![0, 1, .., 7], .. }`.
//!
//! A Bevy `App` is one `World` in one process, so the shard of north_star -- "entity ranges shard across the 8 GPUs of one node with
//! an RCCL all-gather of the packed ViewVisibility bitmask" -- is driven from the ONE thread the system runs on: a library context per
//! device, the World's rows partitioned over them (below), and per frame
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
//! single-device plugin on twin Worlds (`tests/cpp/host_systems_test.cpp: sharded_plugin_leaves_the_same_world`,
//! `sharded_plugin_shards_trees_and_lights`; device lists `{0}`, `{0, 0, 0}` and every GPU of the node).
//!
//! Two partitions, chosen by the World: a FLAT World (no `ChildOf`: configs[1] / configs[3]) is cut into
//! contiguous 256-aligned ranges of its rows in `Entity` order; a World WITH `ChildOf` is sharded by TREE -- forest roots are independent
//! (`propagate_parent_transforms`, crates/bevy_transform/src/systems.rs:522), so whole trees go to contexts, biggest first onto the
//! least loaded, and a tree bigger than a context's fair share (x 1.10) is opened: its root becomes a replicated row that every context
//! holding something below it recomputes (same products, same bits; the lowest such context owns it) and its child subtrees are placed
//! instead ([`place_trees`], the same greedy placement as `bevy_amd/sharding.py: shard_hierarchy` and the C++ host layer's).  Every
//! context then holds its rows in level order with a hierarchy of its own (`mi_upload_hierarchy`); no collective is needed for the
//! transforms, and the lists a view gets are sorted by `Entity` here (rows are in level order, not in `Entity` order).
//! Light clusters (every context assigns the objects whose rows it owns, the per-cluster lists are merged in gather
//! order -- [`merge_cluster_lists`]) are driven through the compiled C++ host layer today (`Mi355xShardedPlugin::frame(.., cam)`,
//! tested on twin Worlds); in this crate the sharded form leaves `assign_objects_to_clusters` to the stock system, as it leaves the
//! shadow views.  Every frame after a device error goes to the stock systems through [`CpuFallback`], like every other system of the
//! crate.  RCCL is loaded with `dlopen` -- no link-time dependency -- and the communicators come from `ncclCommInitAll` over the device
//! list.

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
    /// Per shard: local row -> entity.  Flat Worlds: contiguous ranges of the `Entity` order; Worlds with `ChildOf`: the shard's trees in
    /// level order.
    rows_of: Vec<Vec<Entity>>,
    /// Per shard and local row: this context is responsible for the row (false: it replicates the root of a tree that was opened).
    owned: Vec<Vec<bool>>,
    /// Entity -> the (shard, row) that owns it, and -- replicated roots only -- the other holders.
    primary: EntityHashMap<(u32, u32)>,
    replicas: EntityHashMap<Vec<(u32, u32)>>,
    /// The World has `ChildOf`: sharded by tree (module docs).
    by_tree: bool,
    /// Rows of a block of the gathered buffers: the widest shard, rounded up to whole 256-row workgroups.
    rows_per: u32,
    cnt: Vec<u32>,
    /// Number of views the gathered buffers are laid out for (`u32::MAX`: not yet).
    exchange_views: u32,
    class_bits: bevy_platform::collections::HashMap<TypeId, u32>,
    /// Per shard and local row: the row's visibility-class mask.
    classes: Vec<Vec<u32>>,
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
                rows_of: Vec::new(),
                owned: Vec::new(),
                primary: EntityHashMap::default(),
                replicas: EntityHashMap::default(),
                by_tree: false,
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

/// Where the trees of a forest go.  `parent` / `offs`: the World's rows in level order as `mi_hierarchy_sort` leaves
/// them (`u32::MAX` for roots; rows of a level ordered by parent).
pub(crate) struct TreePlacement {
    /// Per row: the shard that owns it, or -1 for the replicated root of a tree that was opened.
    pub node_rank: Vec<i32>,
    /// Per shard and row: the shard holds the row (its own rows and the replicated ancestors above them).
    pub need: Vec<Vec<bool>>,
    /// Per replicated row: the shard that owns it (the lowest one that holds it); -1 elsewhere.
    pub rep_owner: Vec<i32>,
}

/// Whole trees to shards, biggest first onto the least loaded; while the fullest shard exceeds 1.10 x its fair share, the biggest tree
/// that has children is opened -- its root replicated, its child subtrees placed instead.  The placement of `bevy_amd/sharding.py:
/// shard_hierarchy` and of `Mi355xShardedPlugin::place_trees` (bevy_amd/host/bevy_mi355x_sharded.hpp), which is the compiled and tested one.
pub(crate) fn place_trees(parent: &[u32], offs: &[u32], n_shards: usize) -> TreePlacement {
    let n = parent.len();
    let n_levels = offs.len() - 1;
    let mut size = vec![1u64; n];
    for l in (1..n_levels).rev() {
        for i in offs[l] as usize..offs[l + 1] as usize {
            size[parent[i] as usize] += size[i];
        }
    }
    // children of a row: contiguous in the next level
    let (mut first_child, mut n_children) = (vec![0usize; n], vec![0usize; n]);
    for l in 1..n_levels {
        for i in offs[l] as usize..offs[l + 1] as usize {
            let p = parent[i] as usize;
            if n_children[p] == 0 {
                first_child[p] = i;
            }
            n_children[p] += 1;
        }
    }
    let mut units: Vec<(u64, usize)> = (offs[0] as usize..offs[1] as usize).map(|i| (size[i], i)).collect();
    let mut replicated = vec![false; n];
    let mut unit_rank = vec![-2i32; n];
    let mut load = vec![0u64; n_shards];
    let mut pack = |units: &mut Vec<(u64, usize)>, unit_rank: &mut Vec<i32>, load: &mut Vec<u64>| {
        units.sort_unstable_by_key(|&(sz, root)| (core::cmp::Reverse(sz), root));
        load.fill(0);
        for &(sz, root) in units.iter() {
            let best = (0..n_shards).min_by_key(|&k| (load[k], k)).unwrap();
            load[best] += sz;
            unit_rank[root] = best as i32;
        }
    };
    pack(&mut units, &mut unit_rank, &mut load);
    let ideal = n as f64 / n_shards as f64;
    for _ in 0..64 * n_shards {
        if n_shards == 1 || *load.iter().max().unwrap() as f64 <= 1.10 * ideal {
            break;
        }
        // (units are sorted: the first one that has children is the biggest such)
        let Some(big) = units.iter().position(|&(_, root)| n_children[root] != 0) else { break };
        let (_, root) = units.remove(big);
        replicated[root] = true;
        unit_rank[root] = -2;
        for c in first_child[root]..first_child[root] + n_children[root] {
            units.push((size[c], c));
        }
        pack(&mut units, &mut unit_rank, &mut load);
    }
    let mut node_rank = vec![-2i32; n];
    for l in 0..n_levels {
        for i in offs[l] as usize..offs[l + 1] as usize {
            node_rank[i] = if replicated[i] { -1 } else if unit_rank[i] >= 0 { unit_rank[i] } else { node_rank[parent[i] as usize] };
        }
    }
    let mut need = vec![vec![false; n]; n_shards];
    for i in 0..n {
        if node_rank[i] >= 0 {
            need[node_rank[i] as usize][i] = true;
        }
    }
    for l in (1..n_levels).rev() {
        for i in offs[l] as usize..offs[l + 1] as usize {
            for shard in need.iter_mut() {
                if shard[i] {
                    shard[parent[i] as usize] = true;
                }
            }
        }
    }
    let mut rep_owner = vec![-1i32; n];
    for i in 0..n {
        if replicated[i] {
            rep_owner[i] = (0..n_shards).find(|&k| need[k][i]).map_or(0, |k| k as i32);
            need[rep_owner[i] as usize][i] = true;
        }
    }
    TreePlacement { node_rank, need, rep_owner }
}

/// every shard has assigned the clusterable objects whose rows it owns -- `lists[d] = (offsets[C + 1], indices)`, the
/// indices being positions in the shard's own sublist `sublist[d]` of the gathered list (assign.rs:190-296) -- and this is the view's
/// assignment: per cluster the positions in the GATHERED list, ascending, i.e. the reference's push order (assign.rs:740-800).  A
/// shard's sublist keeps the list's order, so each shard's run ascends and the merge is a merge of sorted runs.  (Per-type counts add
/// up, farthest_z is the maximum: not done here.)
pub(crate) fn merge_cluster_lists(n_clusters: usize, lists: &[(Vec<u32>, Vec<u32>)], sublist: &[Vec<u32>]) -> (Vec<u32>, Vec<u32>) {
    let mut offsets = Vec::with_capacity(n_clusters + 1);
    let mut indices = Vec::new();
    offsets.push(0u32);
    for c in 0..n_clusters {
        let start = indices.len();
        for (d, (off, idx)) in lists.iter().enumerate() {
            if off.is_empty() {
                continue;
            }
            indices.extend(idx[off[c] as usize..off[c + 1] as usize].iter().map(|&local| sublist[d][local as usize]));
        }
        indices[start..].sort_unstable();
        offsets.push(indices.len() as u32);
    }
    (offsets, indices)
}

/// The whole sharded frame (module docs).  Runs in `TransformSystems::Propagate`, in place of [`crate::mi_fused_frame`].
#[allow(clippy::too_many_arguments, clippy::type_complexity)]
pub fn mi_sharded_frame(
    mut shards: ResMut<Mi355xShards>,
    mut fallback: ResMut<CpuFallback>,
    mut frame: ResMut<Mi355xFrame>,
    ticks: SystemChangeTick,
    structure_changed: Query<(), Or<(Added<Transform>, Changed<ChildOf>)>>,
    mut orphaned: RemovedComponents<ChildOf>,
    mut despawned: RemovedComponents<Transform>,
    transforms: Query<(Entity, Ref<Transform>, Option<&ChildOf>)>,
    mut globals: Query<&mut GlobalTransform>,
    cameras: Query<(Entity, &Camera, &bevy_camera::Projection, Option<&RenderLayers>, Has<NoCpuCulling>)>,
    bounds_changed: crate::BoundsChanged,
    rows_query: RowsQuery,
    (range_views, visible_entity_ranges, static_optimizations): (RangeViews, Option<Res<VisibleEntityRanges>>, Option<Res<bevy_transform::systems::StaticTransformOptimizations>>),
) {
    #[cfg(feature = "trace")]
    let _span = bevy_log::info_span!("mi_sharded_frame").entered();
    frame.valid = false;
    frame.clusters_valid = false;
    if fallback.transforms {
        return;
    }
    let sh = &mut *shards;
    let n_shards = sh.ctxs.len();
    let rebuild = !structure_changed.is_empty() || orphaned.read().count() != 0 || despawned.read().count() != 0 || sh.cnt.is_empty();
    let ranges_on = visible_entity_ranges.is_some();
    // mark_dirty_trees returns early unless the optimisation is enabled (systems.rs:131-133); the device takes the same switch
    let static_opt = static_optimizations.is_some_and(|s| s.is_enabled());

    let result: Result<(), ()> = (|| {
        // ---- structure: who holds which row (module docs)
        if rebuild {
            let mut entities: Vec<Entity> = transforms.iter().map(|(e, _, _)| e).collect();
            entities.sort_unstable();
            let n = entities.len();
            sh.by_tree = transforms.iter().any(|(_, _, child_of)| child_of.is_some());
            sh.rows_of = vec![Vec::new(); n_shards];
            sh.owned = vec![Vec::new(); n_shards];
            sh.primary.clear();
            sh.replicas.clear();
            let mut local_parent: Vec<Vec<u32>> = vec![Vec::new(); n_shards];
            let mut local_offs: Vec<Vec<u32>> = vec![Vec::new(); n_shards];
            if !sh.by_tree {
                // row 1: contiguous ranges of a multiple of 256 rows of the Entity order (a mask word never straddles shards)
                let per = (((n + n_shards - 1) / n_shards + 255) / 256 * 256).max(256);
                for d in 0..n_shards {
                    let lo = (d * per).min(n);
                    let m = (n - lo).min(per);
                    sh.rows_of[d] = entities[lo..lo + m].to_vec();
                    sh.owned[d] = vec![true; m];
                    for (k, e) in sh.rows_of[d].iter().enumerate() {
                        sh.primary.insert(*e, (d as u32, k as u32));
                    }
                }
            } else {
                // row 2: the World's rows in level order, then whole trees to shards
                let slot: EntityHashMap<u32> = entities.iter().enumerate().map(|(i, e)| (*e, i as u32)).collect();
                let parent: Vec<u32> = entities
                    .iter()
                    .map(|e| transforms.get(*e).ok().and_then(|(_, _, c)| c.map(ChildOf::parent)).and_then(|p| slot.get(&p).copied()).unwrap_or(ffi::MI_NO_PARENT))
                    .collect();
                let (mut new_to_old, mut pidx, mut offs) = (vec![0u32; n.max(1)], vec![0u32; n.max(1)], vec![0u32; n + 2]);
                let mut n_levels = 0u32;
                // SAFETY: the three outputs hold n, n and n + 2 entries.
                let status = unsafe {
                    ffi::mi_hierarchy_sort(n as u32, parent.as_ptr(), new_to_old.as_mut_ptr(), pidx.as_mut_ptr(), offs.as_mut_ptr(), n as u32 + 2, &mut n_levels)
                };
                if status != ffi::MI_OK {
                    // (MI_ERR_MALFORMED_HIERARCHY: a cycle in ChildOf -- the reference panics there, systems.rs:715; the stock systems decide)
                    error!("bevy_mi355x: mi_hierarchy_sort failed with status {status}");
                    return Err(());
                }
                offs.truncate(n_levels as usize + 1);
                pidx.truncate(n);
                let placed = place_trees(&pidx, &offs, n_shards);
                let mut local_of = vec![0u32; n];
                for d in 0..n_shards {
                    let mut m = 0u32;
                    local_offs[d].push(0);
                    for l in 0..n_levels as usize {
                        for i in offs[l] as usize..offs[l + 1] as usize {
                            if !placed.need[d][i] {
                                continue;
                            }
                            local_of[i] = m;
                            let e = entities[new_to_old[i] as usize];
                            let own = placed.node_rank[i] == d as i32 || placed.rep_owner[i] == d as i32;
                            sh.rows_of[d].push(e);
                            sh.owned[d].push(own);
                            // (a held row's parent is held: `need` is closed upwards)
                            local_parent[d].push(if pidx[i] == ffi::MI_NO_PARENT { ffi::MI_NO_PARENT } else { local_of[pidx[i] as usize] });
                            if own {
                                sh.primary.insert(e, (d as u32, m));
                            } else {
                                sh.replicas.entry(e).or_default().push((d as u32, m));
                            }
                            m += 1;
                        }
                        if m != *local_offs[d].last().unwrap() {
                            local_offs[d].push(m); // (levels the shard holds no row of are dropped)
                        }
                    }
                }
            }
            sh.cnt = sh.rows_of.iter().map(|r| r.len() as u32).collect();
            let widest = sh.cnt.iter().copied().max().unwrap_or(0);
            sh.rows_per = ((widest + 255) / 256 * 256).max(256);
            for d in 0..n_shards {
                let (ctx, m) = (sh.ctxs[d], sh.cnt[d]);
                // SAFETY: plain call on a live context.
                check(ctx, "mi_columns_resize", unsafe { ffi::mi_columns_resize(ctx, m) })?;
                if m == 0 {
                    continue;
                }
                let (mut t, mut r, mut s, mut g, mut keys) = (Vec::new(), Vec::new(), Vec::new(), Vec::new(), Vec::new());
                for e in &sh.rows_of[d] {
                    let (_, tr, _) = transforms.get(*e).map_err(|_| ())?;
                    t.extend_from_slice(&tr.translation.to_array());
                    r.extend_from_slice(&tr.rotation.to_array());
                    s.extend_from_slice(&tr.scale.to_array());
                    let old = globals.get(*e).map(|x| x.affine().to_cols_array()).unwrap_or(Affine3A::IDENTITY.to_cols_array());
                    g.extend_from_slice(&old);
                    keys.push(e.to_bits());
                }
                let levels = if sh.by_tree { local_offs[d].len() as u32 - 1 } else { 1 };
                // SAFETY: every column holds `m` rows; the hierarchy arrays hold m and levels + 1 entries.
                unsafe {
                    check(ctx, "mi_upload_transforms", ffi::mi_upload_transforms(ctx, 0, m, t.as_ptr(), r.as_ptr(), s.as_ptr()))?;
                    check(ctx, "mi_upload_global_transforms", ffi::mi_upload_global_transforms(ctx, 0, m, g.as_ptr()))?;
                    check(ctx, "mi_upload_entity_keys", ffi::mi_upload_entity_keys(ctx, 0, m, keys.as_ptr()))?;
                    let (p, o) = if levels > 1 { (local_parent[d].as_ptr(), local_offs[d].as_ptr()) } else { (ptr::null(), ptr::null()) };
                    check(ctx, "mi_upload_hierarchy", ffi::mi_upload_hierarchy(ctx, m, p, o, levels))?;
                    // (new rows count as changed -- Added<GlobalTransform> -- until their first propagate: no change column yet)
                }
            }
            sh.exchange_views = u32::MAX;
            sh.ranges_resource = None;
        }
        // every holder of an entity: the owner first, then the shards that replicate it
        let holders = |sh: &Mi355xShards, e: Entity, out: &mut Vec<(u32, u32)>| {
            out.clear();
            if let Some(&h) = sh.primary.get(&e) {
                out.push(h);
            }
            if let Some(more) = sh.replicas.get(&e) {
                out.extend_from_slice(more);
            }
        };
        let mut held: Vec<(u32, u32)> = Vec::new();

        // ---- the columns that change rarely, per shard (flags, bounds, layers, classes, ranges: what stage_bounds stages for one context)
        if rebuild || !bounds_changed.is_empty() || sh.ranges_resource != Some(ranges_on) {
            struct Cols {
                center: Vec<f32>,
                half: Vec<f32>,
                flags: Vec<u8>,
                layers: Vec<u32>,
                layers_hi: Vec<u32>,
                ranges: Vec<f32>,
            }
            let mut cols: Vec<Cols> = sh
                .cnt
                .iter()
                .map(|&m| {
                    let m = m as usize;
                    Cols { center: vec![0.0; m * 3], half: vec![0.0; m * 3], flags: vec![0; m], layers: vec![0; m], layers_hi: vec![0; m], ranges: vec![0.0; m * 2] }
                })
                .collect();
            sh.classes = sh.cnt.iter().map(|&m| vec![0u32; m as usize]).collect();
            let mut any_hi = false;
            for (entity, inherited, classes, row_layers, aabb, sphere, _point, _spot, no_frustum_culling, range, _mesh, _nsc, _dl) in rows_query.iter() {
                holders(sh, entity, &mut held);
                if held.is_empty() {
                    continue;
                }
                let mut fl = 0u32;
                // (InheritedVisibility comes from the ECS: the stock visibility_propagate_system stays registered and has run)
                if inherited.get() {
                    fl |= ffi::MI_FLAG_INHERITED_VISIBLE;
                }
                if no_frustum_culling {
                    fl |= ffi::MI_FLAG_NO_FRUSTUM_CULLING;
                }
                let (mut center, mut half, mut row_ranges) = ([0f32; 3], [0f32; 3], [0f32; 2]);
                if let Some(range) = range {
                    fl |= ffi::MI_FLAG_HAS_VISIBILITY_RANGE | if range.use_aabb { ffi::MI_FLAG_RANGE_USE_AABB } else { 0 };
                    row_ranges = [range.start_margin.start, range.end_margin.end];
                }
                if let Some(aabb) = aabb {
                    fl |= ffi::MI_FLAG_HAS_AABB;
                    center = aabb.center.to_array();
                    half = aabb.half_extents.to_array();
                } else if let Some(sphere) = sphere {
                    fl |= ffi::MI_FLAG_HAS_SPHERE;
                    center = sphere.center.to_array();
                    half[0] = sphere.radius;
                }
                let (lo, hi) = match row_layers {
                    None => (1, 0),
                    Some(l) => layer_words_or_log(l)?,
                };
                any_hi |= hi != 0;
                let mut class_mask = 0u32;
                if let Some(classes) = classes {
                    for class in classes.iter() {
                        class_mask |= 1 << mi_class_bit(&mut sh.class_bits, *class).ok_or(())?;
                    }
                }
                for &(d, row) in &held {
                    let (c, row) = (&mut cols[d as usize], row as usize);
                    c.center[row * 3..row * 3 + 3].copy_from_slice(&center);
                    c.half[row * 3..row * 3 + 3].copy_from_slice(&half);
                    c.ranges[row * 2..row * 2 + 2].copy_from_slice(&row_ranges);
                    c.flags[row] = fl as u8;
                    c.layers[row] = lo;
                    c.layers_hi[row] = hi;
                    sh.classes[d as usize][row] = class_mask;
                }
            }
            for d in 0..n_shards {
                let (ctx, m, c) = (sh.ctxs[d], sh.cnt[d], &cols[d]);
                if m == 0 {
                    continue;
                }
                // SAFETY: every column holds `m` rows.
                unsafe {
                    check(ctx, "mi_upload_bounds", ffi::mi_upload_bounds(ctx, 0, m, c.center.as_ptr(), c.half.as_ptr(), c.flags.as_ptr(), c.layers.as_ptr()))?;
                    if any_hi {
                        check(ctx, "mi_upload_render_layers_hi", ffi::mi_upload_render_layers_hi(ctx, 0, m, c.layers_hi.as_ptr()))?;
                    }
                    check(ctx, "mi_upload_visibility_classes", ffi::mi_upload_visibility_classes(ctx, 0, m, sh.classes[d].as_ptr()))?;
                    check(
                        ctx,
                        "mi_upload_visibility_ranges",
                        ffi::mi_upload_visibility_ranges(ctx, 0, m, if ranges_on { c.ranges.as_ptr() } else { ptr::null() }),
                    )?;
                }
            }
            sh.ranges_resource = Some(ranges_on);
        }

        // ---- rows in: Changed<Transform>, to every shard that holds the row
        if !rebuild {
            let mut rows: Vec<Vec<u32>> = vec![Vec::new(); n_shards];
            let mut trs: Vec<(Vec<f32>, Vec<f32>, Vec<f32>)> = vec![Default::default(); n_shards];
            for (entity, tr, _) in transforms.iter() {
                if !tr.last_changed().is_newer_than(ticks.last_run(), ticks.this_run()) {
                    continue;
                }
                holders(sh, entity, &mut held);
                for &(d, row) in &held {
                    let d = d as usize;
                    rows[d].push(row);
                    trs[d].0.extend_from_slice(&tr.translation.to_array());
                    trs[d].1.extend_from_slice(&tr.rotation.to_array());
                    trs[d].2.extend_from_slice(&tr.scale.to_array());
                }
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

        // ---- the frame's views: active cameras in query order, frusta as update_frusta WILL compute them -- from the GlobalTransform
        //      propagate is about to give the camera (its Transform chained up the ChildOf links: crate::expected_global)
        let range_table = if ranges_on { range_view_table(&range_views) } else { EntityHashMap::default() };
        let mut views: Vec<ffi::MiView> = Vec::new();
        let mut frame_views: Vec<FrameView> = Vec::new();
        for (entity, camera, projection, layers, no_cpu_culling) in cameras.iter() {
            if !camera.is_active {
                continue;
            }
            let global = crate::expected_global(entity, &transforms).ok_or_else(|| error!("bevy_mi355x: an active camera without a Transform"))?;
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
                    let flags = ffi::MI_CULL_CHANGED_ROWS | ffi::MI_CULL_END_FRAME | if static_opt { ffi::MI_CULL_STATIC_OPT } else { 0 };
                    check(ctx, "mi_propagate_and_cull_views", ffi::mi_propagate_and_cull_views(ctx, views.as_ptr(), n_views, flags))?;
                } else if sh.cnt[d] != 0 {
                    check(ctx, "mi_propagate", ffi::mi_propagate(ctx, if static_opt { ffi::MI_PROPAGATE_STATIC_OPT } else { 0 }))?;
                }
            }
        }
        if n_views != 0 {
            // SAFETY: the contexts are live and all in MI_EXCHANGE_GROUPED mode; the two addresses come from the RCCL the communicators belong to.
            check(sh.ctxs[0], "mi_exchange_group_flush", unsafe {
                ffi::mi_exchange_group_flush(sh.ctxs.as_ptr(), n_shards as u32, sh.group_start, sh.group_end)
            })?;
        }

        // ---- out: each shard's changed GlobalTransforms of the rows it OWNS (a replicated root comes from its owner: same bits) ...
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
                if !sh.owned[d][crow[k] as usize] {
                    continue;
                }
                let cols: [f32; 12] = cg[k * 12..k * 12 + 12].try_into().unwrap();
                written.push((sh.rows_of[d][crow[k] as usize], cols));
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
        //      systems.rs:62, :719), the lists parked for mi_apply_visibility.
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
                        let row = local as usize;
                        if !sh.owned[d][row] {
                            continue; // (a replicated root: listed by its owner)
                        }
                        for (c, (_, bit)) in class_bits.iter().enumerate() {
                            if sh.classes[d][row] & (1 << bit) != 0 {
                                per_class[c].push(sh.rows_of[d][row]);
                            }
                        }
                    }
                }
            }
            // flat Worlds: rows ascend = Entity order, shard after shard: the lists come out sorted (mod.rs:872-875); sharded by tree the
            // rows of a shard are in level order: sorted here
            if sh.by_tree {
                for list in per_class.iter_mut() {
                    list.sort_unstable();
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
