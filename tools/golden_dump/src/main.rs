//! golden_dump -- the reference's OWN systems run on this repository's fixture inputs.
//!
//!     cargo run --release -- <inputs.migd> <reference_dump.migd>
//!
//! Reads the arrays `tools/golden_dump/export_inputs.py` wrote, builds a Bevy `App` with the stock `TransformPlugin`,
//! `VisibilityPlugin`, `CameraProjectionPlugin` and `LightPlugin` systems, runs one frame per case and writes what those systems
//! produced.  `tests/test_reference_dump.py` compares the oracle (and, on a GPU box, the HIP path) with the dump bit for bit;
//! with the dump present the oracle's pin is the reference itself rather than its literal test vectors.
//!
//! Nothing of this repository is linked in: the tool depends on the Bevy crates only.  It cannot be built in the repository's
//! own image (no Rust toolchain), which is why it is a separate, tiny program and why the test skips when its output is absent.
use std::{collections::BTreeMap, env, fs};

use bevy_app::{App, PostUpdate, TaskPoolPlugin};
use bevy_asset::{AssetApp, AssetPlugin};
use bevy_camera::{
    primitives::{Aabb, Frustum, Sphere},
    visibility::{
        InheritedVisibility, NoCpuCulling, NoFrustumCulling, RenderLayers, ViewVisibility, Visibility, VisibilityClass,
        VisibilityPlugin, VisibleEntities,
    },
    Camera, CameraProjection, CameraProjectionPlugin, PerspectiveProjection, Projection,
};
use bevy_ecs::prelude::*;
use bevy_light::{
    cluster::{ClusterConfig, ClusterFarZMode, ClusterZConfig, ClusterableObjects, Clusters, GlobalClusterSettings},
    LightPlugin, PointLight, SpotLight,
};
use bevy_math::{Affine3A, Quat, UVec2, UVec3, Vec3, Vec3A};
use bevy_transform::{
    components::{GlobalTransform, Transform},
    TransformPlugin,
};

// ---------------------------------------------------------------- MIGD container (tools/golden_dump/migd.py)
enum Array {
    U8(Vec<u8>),
    U32(Vec<u32>),
    F32(Vec<f32>),
    U64(Vec<u64>),
}

fn read_migd(path: &str) -> BTreeMap<String, Array> {
    let buf = fs::read(path).expect("inputs file");
    assert_eq!(&buf[..4], b"MIGD");
    let u32_at = |o: usize| u32::from_le_bytes(buf[o..o + 4].try_into().unwrap());
    let u64_at = |o: usize| u64::from_le_bytes(buf[o..o + 8].try_into().unwrap());
    assert_eq!(u32_at(4), 1);
    let (mut off, mut out) = (12usize, BTreeMap::new());
    for _ in 0..u32_at(8) {
        let len = u32_at(off) as usize;
        let name = String::from_utf8(buf[off + 4..off + 4 + len].to_vec()).unwrap();
        let (code, count) = (u32_at(off + 4 + len), u64_at(off + 8 + len) as usize);
        off += len + 16;
        let array = match code {
            0 => Array::U8(buf[off..off + count].to_vec()),
            1 => Array::U32((0..count).map(|i| u32_at(off + 4 * i)).collect()),
            2 => Array::F32((0..count).map(|i| f32::from_bits(u32_at(off + 4 * i))).collect()),
            3 => Array::U64((0..count).map(|i| u64_at(off + 8 * i)).collect()),
            _ => panic!("dtype {code}"),
        };
        off += count * [1, 4, 4, 8][code as usize];
        out.insert(name, array);
    }
    out
}

fn write_migd(path: &str, arrays: &BTreeMap<String, Array>) {
    let mut buf = b"MIGD".to_vec();
    buf.extend(1u32.to_le_bytes());
    buf.extend((arrays.len() as u32).to_le_bytes());
    for (name, array) in arrays {
        buf.extend((name.len() as u32).to_le_bytes());
        buf.extend(name.as_bytes());
        let (code, count): (u32, usize) = match array {
            Array::U8(v) => (0, v.len()),
            Array::U32(v) => (1, v.len()),
            Array::F32(v) => (2, v.len()),
            Array::U64(v) => (3, v.len()),
        };
        buf.extend(code.to_le_bytes());
        buf.extend((count as u64).to_le_bytes());
        match array {
            Array::U8(v) => buf.extend(v),
            Array::U32(v) => v.iter().for_each(|x| buf.extend(x.to_le_bytes())),
            Array::F32(v) => v.iter().for_each(|x| buf.extend(x.to_bits().to_le_bytes())),
            Array::U64(v) => v.iter().for_each(|x| buf.extend(x.to_le_bytes())),
        }
    }
    fs::write(path, buf).expect("output file");
}

fn f32s<'a>(m: &'a BTreeMap<String, Array>, k: &str) -> &'a [f32] {
    match &m[k] {
        Array::F32(v) => v,
        _ => panic!("{k}: f32 expected"),
    }
}
fn u32s<'a>(m: &'a BTreeMap<String, Array>, k: &str) -> &'a [u32] {
    match &m[k] {
        Array::U32(v) => v,
        _ => panic!("{k}: u32 expected"),
    }
}
fn u8s<'a>(m: &'a BTreeMap<String, Array>, k: &str) -> &'a [u8] {
    match &m[k] {
        Array::U8(v) => v,
        _ => panic!("{k}: u8 expected"),
    }
}

// ---------------------------------------------------------------- helpers
fn transform_at(t: &[f32], r: &[f32], s: &[f32], i: usize) -> Transform {
    Transform {
        translation: Vec3::from_slice(&t[3 * i..]),
        rotation: Quat::from_slice(&r[4 * i..]), // stored x, y, z, w like glam
        scale: Vec3::from_slice(&s[3 * i..]),
    }
}

/// 3x4 column-major, the library's GlobalTransform row layout.
fn global_bits(g: &GlobalTransform) -> [f32; 12] {
    g.affine().to_cols_array()
}

fn camera_transform(cols: &[f32]) -> Transform {
    Transform::from_matrix(Affine3A::from_cols_slice(cols).into())
}

struct Mesh; // the one visibility class of the dump

/// The stock systems of case 2, set up as the reference's own test sets them up (visibility/mod.rs:1314-1322): `VisibilityPlugin`'s
/// bounds systems read `Assets<Mesh>`, so the asset server and the mesh assets have to be there.
fn visibility_app() -> App {
    let mut app = App::new();
    app.add_plugins((TaskPoolPlugin::default(), AssetPlugin::default(), bevy_mesh::MeshPlugin));
    app.add_plugins((TransformPlugin, VisibilityPlugin, CameraProjectionPlugin));
    app
}

/// ... and of cases 3 and 4: `LightPlugin::build` also registers an asset type and a system that reads `Assets<Image>`
/// (crates/bevy_light/src/lib.rs:166-170, atmosphere.rs:549-553).
fn light_app() -> App {
    let mut app = visibility_app();
    app.init_asset::<bevy_image::Image>();
    app.add_plugins(LightPlugin);
    app
}

// flags of include/bevy_mi355x.h
const INHERITED_VISIBLE: u8 = 0x01;
const NO_FRUSTUM_CULLING: u8 = 0x02;
const HAS_AABB: u8 = 0x04;
const HAS_SPHERE: u8 = 0x08;
const NO_CPU_CULLING: u8 = 0x10;

fn main() {
    let args: Vec<String> = env::args().collect();
    let inputs = read_migd(&args[1]);
    let mut out = BTreeMap::new();
    let cam = f32s(&inputs, "camera.fov_aspect_near_far");
    let projection = || {
        Projection::Perspective(PerspectiveProjection { fov: cam[0], aspect_ratio: cam[1], near: cam[2], far: cam[3], ..Default::default() })
    };

    // ------------------------------------------------------------ 1. hierarchy: TransformPlugin over tree.*
    {
        let (parent, t, r, s) = (u32s(&inputs, "tree.parent"), f32s(&inputs, "tree.translation"), f32s(&inputs, "tree.rotation"), f32s(&inputs, "tree.scale"));
        let mut app = App::new();
        // (propagate_parent_transforms takes the ComputeTaskPool, systems.rs:167-172: it has to exist)
        app.add_plugins((TaskPoolPlugin::default(), TransformPlugin));
        let entities: Vec<Entity> = (0..parent.len()).map(|i| app.world_mut().spawn(transform_at(t, r, s, i)).id()).collect();
        for (i, p) in parent.iter().enumerate() {
            if *p != u32::MAX {
                app.world_mut().entity_mut(entities[i]).insert(ChildOf(entities[*p as usize]));
            }
        }
        app.update();
        let g: Vec<f32> = entities.iter().flat_map(|e| global_bits(app.world().get::<GlobalTransform>(*e).unwrap())).collect();
        out.insert("tree.global".to_string(), Array::F32(g));
    }

    // ------------------------------------------------------------ 2. flat frame: propagate + check_visibility over flat.*
    {
        let (t, r, s) = (f32s(&inputs, "flat.translation"), f32s(&inputs, "flat.rotation"), f32s(&inputs, "flat.scale"));
        let (center, half) = (f32s(&inputs, "flat.aabb_center"), f32s(&inputs, "flat.aabb_half"));
        let (flags, layers, view_masks) = (u8s(&inputs, "flat.flags"), u32s(&inputs, "flat.layers"), u32s(&inputs, "flat.view_masks"));
        let cameras = f32s(&inputs, "flat.cameras");
        let layers_of = |word: u32| RenderLayers::from_layers(&(0..32).filter(|b| word >> b & 1 != 0).collect::<Vec<usize>>());
        let mut app = visibility_app();
        let n = flags.len();
        let entities: Vec<Entity> = (0..n)
            .map(|i| {
                let visibility = if flags[i] & INHERITED_VISIBLE != 0 { Visibility::Visible } else { Visibility::Hidden };
                let mut e = app.world_mut().spawn((
                    transform_at(t, r, s, i),
                    visibility,
                    InheritedVisibility::default(),
                    ViewVisibility::default(),
                    VisibilityClass(smallvec::smallvec![core::any::TypeId::of::<Mesh>()]),
                    layers_of(layers[i]),
                ));
                if flags[i] & HAS_AABB != 0 {
                    e.insert(Aabb { center: Vec3A::from_slice(&center[3 * i..]), half_extents: Vec3A::from_slice(&half[3 * i..]) });
                }
                if flags[i] & HAS_SPHERE != 0 {
                    // a world-space bounding sphere: centre in aabb_center, radius in aabb_half[0] (include/bevy_mi355x.h)
                    e.insert(Sphere { center: Vec3A::from_slice(&center[3 * i..]), radius: half[3 * i] });
                }
                if flags[i] & NO_FRUSTUM_CULLING != 0 {
                    e.insert(NoFrustumCulling);
                }
                if flags[i] & NO_CPU_CULLING != 0 {
                    e.insert(NoCpuCulling);
                }
                e.id()
            })
            .collect();
        let views: Vec<Entity> = (0..view_masks.len())
            .map(|v| {
                app.world_mut()
                    .spawn((
                        Camera { is_active: true, ..Default::default() },
                        projection(),
                        camera_transform(&cameras[12 * v..12 * v + 12]),
                        Frustum::default(),
                        VisibleEntities::default(),
                        layers_of(view_masks[v]),
                    ))
                    .id()
            })
            .collect();
        app.update();
        let world = app.world();
        out.insert("flat.global".into(), Array::F32(entities.iter().flat_map(|e| global_bits(world.get::<GlobalTransform>(*e).unwrap())).collect()));
        // the cameras' GlobalTransforms as the systems saw them: `Transform::from_matrix` decomposes the fixture's affine and the
        // propagation recomposes it, which need not give back the same bits -- the test feeds the oracle THESE
        out.insert("flat.camera_global".into(), Array::F32(views.iter().flat_map(|e| global_bits(world.get::<GlobalTransform>(*e).unwrap())).collect()));
        out.insert("flat.view_visible".into(), Array::U8(entities.iter().map(|e| world.get::<ViewVisibility>(*e).unwrap().get() as u8).collect()));
        for (v, view) in views.iter().enumerate() {
            let frustum = world.get::<Frustum>(*view).unwrap();
            out.insert(format!("flat.frustum.{v}"), Array::F32(frustum.half_spaces.iter().flat_map(|h| h.normal_d().to_array()).collect()));
            let list = world.get::<VisibleEntities>(*view).unwrap().get(core::any::TypeId::of::<Mesh>());
            // as rows of the input arrays, ascending (the list is sorted by entity, entities were spawned in row order)
            let mut rows: Vec<u32> = list.iter().map(|e| entities.iter().position(|x| x == e).unwrap() as u32).collect();
            rows.sort_unstable();
            out.insert(format!("flat.visible_rows.{v}"), Array::U32(rows));
        }
    }

    // ------------------------------------------------------------ 3. clusters: assign_objects_to_clusters over cluster.*
    {
        let (camera, lights) = (f32s(&inputs, "cluster.camera"), f32s(&inputs, "cluster.lights_pos_range"));
        let dims = u32s(&inputs, "cluster.screen_dims_z");
        let z = f32s(&inputs, "cluster.first_slice_depth_far_z");
        let mut app = light_app();
        app.insert_resource(GlobalClusterSettings {
            supports_storage_buffers: true,
            clustered_decals_are_usable: true,
            gpu_clustering: None,
            max_uniform_buffer_clusterable_objects: 204,
            view_cluster_bindings_max_indices: 16384,
        });
        let light_entities: Vec<Entity> = (0..lights.len() / 4)
            .map(|i| {
                app.world_mut()
                    .spawn((
                        PointLight { range: lights[4 * i + 3], ..Default::default() },
                        Transform::from_translation(Vec3::from_slice(&lights[4 * i..])),
                        Visibility::Visible,
                        // every light visible: the dump pins the cluster walk, the visibility of lights is case 2's business
                        NoFrustumCulling,
                    ))
                    .id()
            })
            .collect();
        let mut cam_component = Camera { is_active: true, ..Default::default() };
        cam_component.computed.target_info = Some(bevy_camera::RenderTargetInfo { physical_size: UVec2::new(dims[0], dims[1]), scale_factor: 1.0 });
        cam_component.computed.clip_from_view = PerspectiveProjection { fov: cam[0], aspect_ratio: cam[1], near: cam[2], far: cam[3], ..Default::default() }
            .get_clip_from_view();
        let view = app
            .world_mut()
            .spawn((
                cam_component,
                projection(),
                camera_transform(camera),
                Frustum::default(),
                VisibleEntities::default(),
                Clusters::default(),
                ClusterConfig::XYZ {
                    dimensions: UVec3::new(dims[2], dims[3], dims[4]),
                    z_config: ClusterZConfig { first_slice_depth: z[0], far_z_mode: ClusterFarZMode::Constant(z[1]) },
                    dynamic_resizing: false,
                },
            ))
            .id();
        app.update();
        out.insert("cluster.camera_global".into(), Array::F32(global_bits(app.world().get::<GlobalTransform>(view).unwrap()).to_vec()));
        let clusters = app.world().get::<Clusters>(view).unwrap();
        out.insert("cluster.dims".into(), Array::U32(clusters.dimensions.to_array().to_vec()));
        out.insert("cluster.near_far".into(), Array::F32(vec![clusters.near, clusters.far]));
        let ClusterableObjects::Cpu(per_cluster) = &clusters.clusterable_objects else { panic!("CPU clustering expected") };
        let (mut offsets, mut indices) = (vec![0u32], Vec::new());
        for objects in per_cluster {
            // as indices into cluster.lights_pos_range, in list order
            indices.extend(objects.iter().map(|e| light_entities.iter().position(|x| x == e).unwrap() as u32));
            offsets.push(indices.len() as u32);
        }
        out.insert("cluster.offsets".into(), Array::U32(offsets));
        out.insert("cluster.indices".into(), Array::U32(indices));
        out.insert("cluster.farthest_z".into(), Array::F32(vec![clusters.last_frame_farthest_z.unwrap_or(f32::NAN)]));
        out.insert("cluster.total".into(), Array::U64(vec![clusters.last_frame_total_cluster_index_count.unwrap_or(0) as u64]));
    }

    // ------------------------------------------------------------ 4. clusters again: point AND spot lights, TWO clustered cameras
    // (the shapes round 4 added to the fused frame: the cone test of assign.rs:681-738 and one Clusters component per camera)
    {
        let lights = f32s(&inputs, "cluster2.lights_pos_range");
        let (kind, rotation, outer) = (u8s(&inputs, "cluster2.type"), f32s(&inputs, "cluster2.rotation"), f32s(&inputs, "cluster2.outer_angle"));
        let cameras = f32s(&inputs, "cluster2.cameras");
        let dims = u32s(&inputs, "cluster.screen_dims_z");
        let z = f32s(&inputs, "cluster.first_slice_depth_far_z");
        let mut app = light_app();
        app.insert_resource(GlobalClusterSettings {
            supports_storage_buffers: true,
            clustered_decals_are_usable: true,
            gpu_clustering: None,
            max_uniform_buffer_clusterable_objects: 204,
            view_cluster_bindings_max_indices: 16384,
        });
        // the inputs list every point light before the first spot light: the gather order of assign.rs:189-230
        let light_entities: Vec<Entity> = (0..kind.len())
            .map(|i| {
                let transform = Transform {
                    translation: Vec3::from_slice(&lights[4 * i..]),
                    rotation: Quat::from_slice(&rotation[4 * i..]),
                    scale: Vec3::ONE,
                };
                let mut e = app.world_mut().spawn((transform, Visibility::Visible, NoFrustumCulling));
                if kind[i] == 1 {
                    e.insert(SpotLight { range: lights[4 * i + 3], outer_angle: outer[i], inner_angle: 0.0, ..Default::default() });
                } else {
                    e.insert(PointLight { range: lights[4 * i + 3], ..Default::default() });
                }
                e.id()
            })
            .collect();
        let views: Vec<Entity> = (0..cameras.len() / 12)
            .map(|v| {
                let mut cam_component = Camera { is_active: true, order: v as isize, ..Default::default() };
                cam_component.computed.target_info =
                    Some(bevy_camera::RenderTargetInfo { physical_size: UVec2::new(dims[0], dims[1]), scale_factor: 1.0 });
                cam_component.computed.clip_from_view =
                    PerspectiveProjection { fov: cam[0], aspect_ratio: cam[1], near: cam[2], far: cam[3], ..Default::default() }.get_clip_from_view();
                app.world_mut()
                    .spawn((
                        cam_component,
                        projection(),
                        camera_transform(&cameras[12 * v..12 * v + 12]),
                        Frustum::default(),
                        VisibleEntities::default(),
                        Clusters::default(),
                        ClusterConfig::XYZ {
                            dimensions: UVec3::new(dims[2], dims[3], dims[4]),
                            z_config: ClusterZConfig { first_slice_depth: z[0], far_z_mode: ClusterFarZMode::Constant(z[1]) },
                            dynamic_resizing: false,
                        },
                    ))
                    .id()
            })
            .collect();
        app.update();
        let world = app.world();
        // what the assignment reads of a spot light: GlobalTransform::back() and ops::sin_cos(outer_angle) (assign.rs:563-573)
        out.insert(
            "cluster2.spot_back".into(),
            Array::F32(light_entities.iter().flat_map(|e| world.get::<GlobalTransform>(*e).unwrap().back().to_array()).collect()),
        );
        out.insert("cluster2.camera_global".into(), Array::F32(views.iter().flat_map(|e| global_bits(world.get::<GlobalTransform>(*e).unwrap())).collect()));
        out.insert("cluster2.sin_cos".into(), Array::F32(outer.iter().flat_map(|a| { let (s, c) = bevy_math::ops::sin_cos(*a); [s, c] }).collect()));
        for (v, view) in views.iter().enumerate() {
            let clusters = world.get::<Clusters>(*view).unwrap();
            let ClusterableObjects::Cpu(per_cluster) = &clusters.clusterable_objects else { panic!("CPU clustering expected") };
            let (mut offsets, mut indices) = (vec![0u32], Vec::new());
            for objects in per_cluster {
                indices.extend(objects.iter().map(|e| light_entities.iter().position(|x| x == e).unwrap() as u32));
                offsets.push(indices.len() as u32);
            }
            out.insert(format!("cluster2.dims.{v}"), Array::U32(clusters.dimensions.to_array().to_vec()));
            out.insert(format!("cluster2.offsets.{v}"), Array::U32(offsets));
            out.insert(format!("cluster2.indices.{v}"), Array::U32(indices));
            out.insert(format!("cluster2.farthest_z.{v}"), Array::F32(vec![clusters.last_frame_farthest_z.unwrap_or(f32::NAN)]));
            out.insert(format!("cluster2.total.{v}"), Array::U64(vec![clusters.last_frame_total_cluster_index_count.unwrap_or(0) as u64]));
        }
    }

    write_migd(&args[2], &out);
    println!("wrote {} ({} arrays)", args[2], out.len());
    let _ = PostUpdate; // schedules are the stock ones; named here only to keep the import honest
}
