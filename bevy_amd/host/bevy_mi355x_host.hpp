// bevy_mi355x_host.hpp -- C++ host side of the drop-in, above the C ABI (include/bevy_mi355x.h).
//
// The reference's host language is Rust and this image has no Rust toolchain, so the host layer a Bevy maintainer
// would write as a `bevy_mi355x` plugin crate is written here in C++ with the same shape: a World
// holding the components of the path (Transform, GlobalTransform, ChildOf / Children, Visibility, InheritedVisibility,
// ViewVisibility, Aabb), change detection flags the systems consume, and a plugin whose systems have the names, the
// ordering and the observable behaviour of the stock ones they replace:
//     propagate_transforms()    = (mark_dirty_trees, sync_simple_transforms, propagate_parent_transforms).chain()
//                                 crates/bevy_transform/src/systems.rs:42-79,111-306,506-748, plugins.rs:36-47
//     visibility_propagate()    = visibility_propagate_system, crates/bevy_camera/src/visibility/mod.rs:638-729
//     check_visibility()        = reset_view_visibility + check_visibility_cpu_culling + mark_newly_hidden_entities_invisible
//                                 crates/bevy_camera/src/visibility/mod.rs:733-737,748-876,908-918
// No arithmetic of the path happens here: every system copies the changed rows in, calls the library, and writes the
// changed rows back with the reference's change-tick behaviour.  A malformed hierarchy throws (the reference panics,
// systems.rs:715).  tests/cpp/host_systems_test.cpp restates the reference's own system tests against this layer.
#pragma once

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <optional>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/bevy_mi355x.h"
#include "../csrc/glam_math.h"  // host build of the glam op order: used only to construct values (from_xyz, operator*)

namespace bevy_mi355x {

struct Vec3 {
    float x = 0, y = 0, z = 0;
};
struct Quat {
    float x = 0, y = 0, z = 0, w = 1;
};

// Transform, crates/bevy_transform/src/components/transform.rs:86-106
struct Transform {
    Vec3 translation;
    Quat rotation;
    Vec3 scale{1, 1, 1};
    static Transform from_xyz(float x, float y, float z) { Transform t; t.translation = {x, y, z}; return t; }
    static Transform from_translation(Vec3 v) { Transform t; t.translation = v; return t; }
    static Transform identity() { return Transform{}; }
};

// GlobalTransform(Affine3A), components/global_transform.rs:60 -- cols = Affine3A::to_cols_array()
struct GlobalTransform {
    float cols[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
    // radius_vec3a(Vec3A::ONE) = (matrix3 * extents).length() (global_transform.rs:252-254): a light probe's range (assign.rs:262)
    float radius_of_unit_extents() const {
        const mi::Affine a = mi::load_affine(cols);
        return mi::length3(mi::mul(a.m, mi::V3{1.0f, 1.0f, 1.0f}));
    }
    // scale().length() (global_transform.rs:240-248; the sign of the determinant only flips x's sign, which the square drops):
    // a clustered decal's range (assign.rs:287)
    float scale_length() const {
        const mi::Affine a = mi::load_affine(cols);
        const float x = mi::length3(a.m.x_axis), y = mi::length3(a.m.y_axis), z = mi::length3(a.m.z_axis);
        return mi::f_sqrt((x * x + y * y) + z * z);
    }
    static GlobalTransform from(const Transform& t) {  // global_transform.rs:326-330
        const mi::Affine a = mi::affine_from_srt(mi::V3{t.scale.x, t.scale.y, t.scale.z},
                                                 mi::V4{t.rotation.x, t.rotation.y, t.rotation.z, t.rotation.w},
                                                 mi::V3{t.translation.x, t.translation.y, t.translation.z});
        GlobalTransform g;
        mi::store_affine(a, g.cols);
        return g;
    }
    static GlobalTransform from_xyz(float x, float y, float z) { return from(Transform::from_xyz(x, y, z)); }
    static GlobalTransform from_translation(Vec3 v) { return from(Transform::from_translation(v)); }
    // GlobalTransform * Transform = mul_transform, global_transform.rs:315-317
    GlobalTransform operator*(const Transform& t) const {
        const mi::Affine r = mi::mul(mi::load_affine(cols), mi::load_affine(from(t).cols));
        GlobalTransform g;
        mi::store_affine(r, g.cols);
        return g;
    }
    Vec3 translation() const { return {cols[9], cols[10], cols[11]}; }
    bool operator==(const GlobalTransform& o) const {  // PartialEq: 12 float compares
        for (int i = 0; i < 12; ++i)
            if (!(cols[i] == o.cols[i])) return false;
        return true;
    }
    bool operator!=(const GlobalTransform& o) const { return !(*this == o); }
};

struct Aabb {  // crates/bevy_camera/src/primitives.rs:63-68
    Vec3 center, half_extents;
};
enum class Visibility : uint8_t { Inherited = MI_VISIBILITY_INHERITED, Hidden = MI_VISIBILITY_HIDDEN, Visible = MI_VISIBILITY_VISIBLE };

// Entity, crates/bevy_ecs/src/entity/mod.rs:424; to_bits orders by generation then by the NonMax-encoded (inverted) index
struct Entity {
    uint32_t index = 0xFFFFFFFFu, generation = 0;
    uint64_t to_bits() const { return ((uint64_t)generation << 32) | (uint64_t)(index ^ 0xFFFFFFFFu); }
    bool operator==(const Entity& o) const { return index == o.index && generation == o.generation; }
    bool operator!=(const Entity& o) const { return !(*this == o); }
};

// VisibilityRange, crates/bevy_camera/src/visibility/range.rs:78-112
struct VisibilityRange {
    float start_margin_start = 0, start_margin_end = 0, end_margin_start = 0, end_margin_end = 0;
    bool use_aabb = false;
    static VisibilityRange abrupt(float start, float end) { return VisibilityRange{start, start, end, end, false}; }  // range.rs:135-141
    // is_visible_at_all (range.rs:159-161): what check_visibility_ranges asks per (entity, view)
    bool is_visible_at_all(float camera_distance) const { return camera_distance >= start_margin_start && camera_distance < end_margin_end; }
};
// What the shadow-view systems read of a light (crates/bevy_light/src/lib.rs:342-757).  The frusta are INPUTS here, as they are for the
// reference's systems: build_directional_light_cascades / update_directional_light_frusta, update_point_light_frusta and
// update_spot_light_frusta stay the stock systems (per light, not per entity) and run in front of SimulationLightSystems::CheckLightVisibility.
struct Frustum6 {
    float half_spaces[24];
};
struct DirectionalLight {                          // DirectionalLight + CascadesFrusta
    bool shadow_maps_enabled = true;
    std::vector<std::vector<Frustum6>> cascades;   // CascadesFrusta::frusta: per view (index into the frame's views), one frustum per cascade
};
struct PointLightShadows {                         // PointLight::shadow_maps_enabled + CubemapFrusta
    bool shadow_maps_enabled = true;
    Frustum6 cubemap_frusta[6];
};
struct SpotLightShadows {                          // SpotLight::shadow_maps_enabled + Frustum
    bool shadow_maps_enabled = true;
    Frustum6 frustum;
};

// The slice of the ECS the path touches: dense per-entity columns plus the change flags the systems read.
struct MeshBinning {
    uint64_t batch_set_key = 0;  // BinnedPhaseItem::BatchSetKey, ordered (phase.multidrawable_meshes is an IndexMap sorted by key)
    bool indexed = true;         // PhaseItemBatchSetKey::indexed()
    uint64_t bin_key = 0;        // BinnedPhaseItem::BinKey
    uint32_t input_uniform_index = 0;
};
struct BatchSetRecord {  // BinnedRenderPhaseBatchSet, gpu_preprocessing.rs:2560-2577
    uint64_t batch_set_key;
    bool indexed;
    uint32_t index, first_work_item_index, instance_count, first_indirect_parameters_index, batch_count, first_output_mesh_uniform_index;
};
struct BatchBin {  // one bin of a batch set, with the metadata the device filled in
    uint64_t batch_set_key, bin_key;
    mi_bin_metadata metadata;
};
struct PhaseBatches {  // what the multidrawable pass of batch_and_prepare_binned_render_phase leaves behind for one view
    std::vector<mi_preprocess_work_item> work_items[2];               // [0] non-indexed, [1] indexed
    std::vector<mi_indirect_parameters_metadata> metadata[2];
    std::vector<mi_indirect_batch_set> batch_sets[2];
    std::vector<BatchSetRecord> records;
    std::vector<BatchBin> bins;
    mi_batch_totals totals{};
};

class World {
  public:
    Entity spawn(const Transform& t = Transform{}) {
        uint32_t i;
        if (!free_.empty()) { i = free_.back(); free_.pop_back(); }
        else {
            i = (uint32_t)rec_.size();
            rec_.emplace_back(); vv_.push_back(0); vv_changed_.push_back(0);
            transform_.emplace_back(); global_.emplace_back(); moved_.push_back(0); global_changed_.push_back(0); touched_flag_.push_back(0);
        }
        Rec& r = rec_[i];
        const uint32_t gen = r.generation;
        r = Rec{};
        r.generation = gen;
        r.alive = true;
        transform_[i] = t;
        global_[i] = GlobalTransform{};
        global_changed_[i] = 0;
        r.transform_changed = r.added = true;
        moved_[i] = 1;
        vv_[i] = vv_changed_[i] = 0;
        touch(i);
        ++structure_version_;
        return Entity{i, gen};
    }
    Entity spawn_child(Entity parent, const Transform& t = Transform{}) { Entity c = spawn(t); add_child(parent, c); return c; }
    // despawn: ChildOf is a linked_spawn relationship -- descendants go too (crates/bevy_ecs/src/hierarchy.rs:107)
    bool despawn(Entity e) {
        if (!contains(e)) return false;
        const std::vector<Entity> kids = rec(e).children;
        for (Entity c : kids) despawn(c);
        remove_parent(e);
        Rec& r = rec(e);
        if (r.point_light_range || r.spot_light || r.rect_light_range || r.light_probe || r.decal) ++lights_version_;
        n_shadow_lights_ -= (r.directional_light ? 1u : 0u) + (r.point_shadows ? 1u : 0u) + (r.spot_shadows ? 1u : 0u);
        r.alive = false;
        moved_[e.index] = 0;
        ++r.generation;
        free_.push_back(e.index);
        ++structure_version_;
        return true;
    }
    bool contains(Entity e) const { return e.index < rec_.size() && rec_[e.index].alive && rec_[e.index].generation == e.generation; }

    // entity.insert(ChildOf(parent)) / parent.add_child(child)
    void add_child(Entity parent, Entity child) {
        remove_parent(child);
        rec(child).parent = parent;
        rec(child).parent_changed = true;
        moved_[child.index] = 1;
        touch(child.index);
        rec(parent).children.push_back(child);
        ++structure_version_;
    }
    void add_children(Entity parent, const std::vector<Entity>& kids) { for (Entity c : kids) add_child(parent, c); }
    // entity.remove::<ChildOf>()
    void remove_parent(Entity child) {
        Rec& r = rec(child);
        if (!r.parent) return;
        auto& sib = rec(*r.parent).children;
        sib.erase(std::remove(sib.begin(), sib.end(), child), sib.end());
        r.parent.reset();
        r.orphaned = true;
        moved_[child.index] = 1;
        touch(child.index);
        ++structure_version_;
    }
    // test-only: corrupt ChildOf without touching Children (systems.rs:1127-1147 does the same with unsafe code)
    void set_child_of_unchecked(Entity child, Entity parent) { rec(child).parent = parent; ++structure_version_; }

    const Transform& transform(Entity e) const { rec(e); return transform_[e.index]; }
    Transform& transform_mut(Entity e) {  // DerefMut bumps the tick
        Rec& r = rec(e);
        if (r.light_probe || r.decal) ++lights_version_;  // (their cluster range is a function of the Transform: the fused frame's object list goes up again)
        r.transform_changed = true;
        moved_[e.index] = 1;
        touch(e.index);
        return transform_[e.index];
    }
    const GlobalTransform& global_transform(Entity e) const { rec(e); return global_[e.index]; }
    bool global_transform_changed(Entity e) const { rec(e); return global_changed_[e.index] != 0; }
    std::optional<Entity> parent(Entity e) const { return rec(e).parent; }
    const std::vector<Entity>& children(Entity e) const { return rec(e).children; }

    void insert_visibility(Entity e, Visibility v) {
        Rec& r = rec(e);
        r.visibility = v;
        r.has_visibility = true;
        r.visibility_changed = true;
        touch(e.index);
        ++visibility_version_;
        ++bounds_version_;  // (entities without Visibility default to visible: the flag byte depends on its presence)
    }
    bool inherited_visibility(Entity e) const { return rec(e).inherited; }
    bool has_aabb(Entity e) const { return rec(e).aabb.has_value(); }
    bool has_point_light(Entity e) const { return rec(e).point_light_range.has_value(); }
    bool inherited_visibility_changed(Entity e) const { return rec(e).inherited_changed; }
    void insert_aabb(Entity e, Aabb a) { rec(e).aabb = a; rec(e).bounds_changed = true; touch(e.index); ++bounds_version_; }
    // RenderLayers (first word), NoFrustumCulling, VisibilityRange: inputs of every visibility closure (visibility/mod.rs:800-846)
    void insert_render_layers(Entity e, uint32_t mask) { rec(e).render_layers = mask; ++bounds_version_; }
    void insert_no_frustum_culling(Entity e) { rec(e).no_frustum_culling = true; ++bounds_version_; }
    void insert_visibility_range(Entity e, VisibilityRange r) { rec(e).visibility_range = r; ++bounds_version_; }
    // init_resource::<VisibleEntityRanges>() (VisibilityRangePlugin, range.rs:33-40): without it `visible_entity_ranges` is None in every
    // visibility system and no VisibilityRange hides anything (visibility/mod.rs:814-816 is_some_and)
    void set_visible_entity_ranges(bool on) { if (visible_entity_ranges_ != on) { visible_entity_ranges_ = on; ++bounds_version_; } }
    bool visible_entity_ranges() const { return visible_entity_ranges_; }
    // Mesh3d / NotShadowCaster: the shadow views' query is With<Mesh3d>, Without<NotShadowCaster>, Without<DirectionalLight> (bevy_light/src/lib.rs:355-372)
    void insert_mesh3d(Entity e) { rec(e).mesh3d = true; ++bounds_version_; }
    void insert_not_shadow_caster(Entity e) { rec(e).not_shadow_caster = true; ++bounds_version_; }
    void insert_directional_light(Entity e, DirectionalLight l) { n_shadow_lights_ += rec(e).directional_light ? 0 : 1; rec(e).directional_light = std::move(l); ++bounds_version_; }
    DirectionalLight& directional_light_mut(Entity e) { return *rec(e).directional_light; }
    void insert_point_light_shadows(Entity e, PointLightShadows s) { n_shadow_lights_ += rec(e).point_shadows ? 0 : 1; rec(e).point_shadows = s; }
    void insert_spot_light_shadows(Entity e, SpotLightShadows s) { n_shadow_lights_ += rec(e).spot_shadows ? 0 : 1; rec(e).spot_shadows = s; }
    PointLightShadows& point_light_shadows_mut(Entity e) { return *rec(e).point_shadows; }
    SpotLightShadows& spot_light_shadows_mut(Entity e) { return *rec(e).spot_shadows; }
    void insert_point_light(Entity e, float range) { rec(e).point_light_range = range; ++lights_version_; ++bounds_version_; }  // PointLight { range, .. }
    void insert_spot_light(Entity e, float range, float outer_angle) {  // SpotLight { range, outer_angle, .. }
        rec(e).spot_light = std::make_pair(range, outer_angle);
        ++lights_version_;
        ++bounds_version_;
    }
    void insert_rect_light(Entity e, float range) { rec(e).rect_light_range = range; ++lights_version_; ++bounds_version_; }  // RectLight { range, .. }
    // LightProbe (+ EnvironmentMapLight = a reflection probe, else an irradiance volume) and ClusteredDecal: clustered with a range
    // taken from the entity's GlobalTransform of this very frame (assign.rs:250-296)
    void insert_light_probe(Entity e, bool is_reflection_probe) { rec(e).light_probe = is_reflection_probe; ++lights_version_; }
    void insert_clustered_decal(Entity e) { rec(e).decal = true; ++lights_version_; }
    void set_clustered_decals_are_usable(bool on) { clustered_decals_are_usable = on; ++lights_version_; }
    bool clustered_decals_are_usable = true;
    // GlobalClusterSettings::supports_storage_buffers: without it rect lights are not gathered at all (assign.rs:231-248)
    void set_supports_storage_buffers(bool on) { supports_storage_buffers = on; ++lights_version_; }
    bool supports_storage_buffers = true;
    // What queue_material_meshes decides for a multidrawable mesh instance: the batch set key (pipeline + bind groups +
    // slabs), the bin key (mesh asset) and its slot in the MeshInputUniform buffer (render_phase/mod.rs:1086-1180)
    void insert_mesh_binning(Entity e, MeshBinning b) { rec(e).binning = b; ++binning_version_; }
    void remove_mesh_binning(Entity e) { rec(e).binning.reset(); ++binning_version_; }
    bool view_visibility(Entity e) const { rec(e); return (vv_[e.index] & 1u) != 0; }  // ViewVisibility::get
    bool view_visibility_changed(Entity e) const { rec(e); return vv_changed_[e.index] != 0; }
    uint8_t view_visibility_bits(Entity e) const { rec(e); return vv_[e.index]; }  // ViewVisibility's packed byte (bit 0 current, bit 1 previous)

    // ---- the stock systems that STAY registered next to the fused frame (they are private in the reference, so a plugin cannot
    // take them out; in the three-system form the device runs them as part of mi_cull and the whole column comes back).
    // ViewVisibility is a dense byte column here, as it is a table column in the ECS. ----
    // reset_view_visibility, visibility/mod.rs:733-737: ViewVisibility::update() under bypass_change_detection
    void reset_view_visibility() {  // (eight bytes of the column at a time)
        uint8_t* vv = vv_.data();
        const size_t n = vv_.size();
        size_t i = 0;
        for (; i + 8 <= n; i += 8) {
            uint64_t x;
            std::memcpy(&x, vv + i, 8);
            x = (x & 0x0101010101010101ull) << 1;
            std::memcpy(vv + i, &x, 8);
        }
        for (; i < n; ++i) vv[i] = (uint8_t)((vv[i] & 1u) << 1);
    }
    // SetViewVisibility::set_visible, visibility/mod.rs:290-306: the tick moves only on a hidden -> visible transition
    void set_visible(Entity e) {
        uint8_t& v = vv_[e.index];
        if (v & 1u) return;
        if (!(v & 2u)) vv_changed_[e.index] = 1;
        v |= 1u;
    }
    // mark_newly_hidden_entities_invisible, visibility/mod.rs:908-918
    void mark_newly_hidden_entities_invisible() {
        const size_t n = vv_.size();
        uint8_t* vv = vv_.data();
        uint8_t* chg = vv_changed_.data();
        size_t i = 0;
        for (; i + 8 <= n; i += 8) {  // (eight bytes at a time: hide = previous-frame bit set, this-frame bit clear)
            uint64_t x, c;
            std::memcpy(&x, vv + i, 8);
            const uint64_t hide = (x >> 1) & ~x & 0x0101010101010101ull;
            if (!hide) continue;
            std::memcpy(&c, chg + i, 8);
            x &= ~(hide * 0xFFull);
            c |= hide;
            std::memcpy(vv + i, &x, 8);
            std::memcpy(chg + i, &c, 8);
        }
        for (; i < n; ++i) {
            const uint8_t v = vv[i];
            const uint8_t hide = (uint8_t)((v & 3u) == 2u);
            vv[i] = hide ? (uint8_t)0 : v;
            chg[i] = (uint8_t)(chg[i] | hide);
        }
    }

    // World::clear_trackers(): change flags older than this frame are no longer "changed"
    void clear_trackers() {
        for (uint32_t i : touched_) {
            Rec& r = rec_[i];
            r.transform_changed = r.added = r.parent_changed = r.orphaned = false;
            r.inherited_changed = false;
            r.visibility_changed = r.bounds_changed = false;
            moved_[i] = global_changed_[i] = touched_flag_[i] = 0;
        }
        touched_.clear();
        std::fill(vv_changed_.begin(), vv_changed_.end(), (uint8_t)0);
    }
    std::vector<Entity> entities() const {
        std::vector<Entity> out;
        for (uint32_t i = 0; i < rec_.size(); ++i)
            if (rec_[i].alive) out.push_back(Entity{i, rec_[i].generation});
        return out;
    }
    bool static_transform_optimizations = false;  // Res<StaticTransformOptimizations>, systems.rs:87-103

  private:
    friend class Mi355xPlugin;
    friend class Mi355xShardedPlugin;  // bevy_mi355x_sharded.hpp: the same World driven over several GPUs
    struct Rec {
        bool alive = false;
        uint32_t generation = 0;
        std::optional<Entity> parent;
        std::vector<Entity> children;
        Visibility visibility = Visibility::Inherited;
        bool has_visibility = false;
        bool inherited = false;  // InheritedVisibility::default() == HIDDEN
        std::optional<Aabb> aabb;
        std::optional<float> point_light_range;
        std::optional<std::pair<float, float>> spot_light;  // (range, outer_angle)
        std::optional<float> rect_light_range;
        std::optional<bool> light_probe;  // is_reflection_probe
        bool decal = false;
        std::optional<MeshBinning> binning;
        uint32_t render_layers = 1;  // RenderLayers::default() = layer 0
        bool no_frustum_culling = false, mesh3d = false, not_shadow_caster = false;
        std::optional<VisibilityRange> visibility_range;
        std::optional<DirectionalLight> directional_light;
        std::optional<PointLightShadows> point_shadows;
        std::optional<SpotLightShadows> spot_shadows;
        bool transform_changed = false, added = false, parent_changed = false, orphaned = false;
        bool inherited_changed = false;
        bool visibility_changed = false, bounds_changed = false;
    };
    Rec& rec(Entity e) {
        if (!contains(e)) throw std::out_of_range("no such entity");
        return rec_[e.index];
    }
    const Rec& rec(Entity e) const {
        if (!contains(e)) throw std::out_of_range("no such entity");
        return rec_[e.index];
    }
    void touch(uint32_t i) {
        if (!touched_flag_[i]) {
            touched_flag_[i] = 1;
            touched_.push_back(i);
        }
    }
    std::vector<Rec> rec_;
    // The components the per-frame loops stream through live in columns of their own, by entity index -- as they do in Bevy's tables
    // (a Rec is ~200 bytes: walking a million of them for one field each was the whole cost of the gather and write-back loops):
    std::vector<Transform> transform_;
    std::vector<GlobalTransform> global_;
    std::vector<uint8_t> moved_;           // alive && (Changed<Transform> || Added || Changed<ChildOf> || orphaned): what the propagate's filter asks
    std::vector<uint8_t> global_changed_;  // Changed<GlobalTransform>
    std::vector<uint8_t> touched_flag_;    // listed in touched_: some change flag is set (what a change-tick scan would find)
    std::vector<uint8_t> vv_, vv_changed_;  // ViewVisibility's packed byte and its change flag, by entity index
    std::vector<uint32_t> touched_;         // entity indices with a change flag set since clear_trackers
    std::vector<uint32_t> free_;
    uint64_t structure_version_ = 1;
    uint64_t binning_version_ = 1;
    uint64_t lights_version_ = 1;
    uint64_t visibility_version_ = 1;  // Visibility components written
    uint64_t bounds_version_ = 1;      // Aabb / light bounds / layers / ranges / shadow-caster markers written
    bool visible_entity_ranges_ = false;
    uint32_t n_shadow_lights_ = 0;     // entities with a DirectionalLight / PointLightShadows / SpotLightShadows
};

// Clusters + ObjectsInClusterCpu, crates/bevy_light/src/cluster/mod.rs:143-213
struct ObjectsInCluster {
    std::vector<Entity> entities;  // push order of assign_objects_to_clusters
    uint32_t counts[6] = {0, 0, 0, 0, 0, 0};  // ClusterableObjectCounts: point, spot, rect, reflection probes, irradiance volumes, decals
};
struct Clusters {
    uint32_t dimensions[3] = {0, 0, 0};
    std::vector<ObjectsInCluster> clusterable_objects;  // index (y * dims.x + x) * dims.z + z
    float farthest_z = 0.0f;
    uint64_t total_index_count = 0;
};
struct ClusterCamera {  // what the per-view setup of assign.rs:342-485 reads
    float camera_affine[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
    float clip_from_view[16];
    float frustum[24];
    uint32_t screen_width = 1920, screen_height = 1080;
    uint32_t requested_dimensions[3] = {16, 9, 24};  // ClusterConfig::XYZ
    float first_slice_depth = 5.0f, far_z = 1000.0f; // ClusterFarZMode::Constant
    uint32_t layer_mask = 1;
};

struct View {  // an active camera: Frustum + RenderLayers (crates/bevy_camera/src/visibility/mod.rs:756-781)
    float frustum[24];
    uint32_t layer_mask = 1;
    // check_visibility_ranges (range.rs:225-284) gives the first 32 entities of its view query (cameras and ShadowLodOrigins, active
    // or not) an index in VisibleEntityRanges and measures distances from their GlobalTransform translation; a view past the 32nd
    // has none, and every ranged entity is out of its range (entity_is_in_range_of_view, range.rs:209-217)
    Vec3 position{};
    bool has_range_index = true;
};
// get_shadow_lod_origin (bevy_light/src/lib.rs:752-799): the entity point and spot light shadow maps take their LOD distances from
struct ShadowLodOrigin {
    Vec3 position{};
    bool has_range_index = true;
};
// CascadesVisibleEntities / CubemapVisibleEntities / VisibleMeshEntities as the shadow-view systems leave them (sorted by Entity,
// bevy_light/src/lib.rs:489, 664, 745).  A light whose shadow maps are off or that is not visible has empty lists (directional,
// lib.rs:400-404) or is not listed (point / spot: the systems `continue`, lib.rs:579-581, 674-676).
struct LightVisibility {
    struct Directional { Entity light; std::vector<std::vector<std::vector<Entity>>> entities; };  // [view][cascade]
    struct Point { Entity light; std::vector<Entity> faces[6]; };
    struct Spot { Entity light; std::vector<Entity> entities; };
    std::vector<Directional> directional;
    std::vector<Point> point;
    std::vector<Spot> spot;
    uint32_t n_shadow_views = 0;  // mi_views of the device pass
};

// A handful of threads for the per-frame loops over the tables: the gather of the moved Transforms into the upload windows and the
// write-back of the changed GlobalTransforms (Bevy runs such loops as par_iter on its ComputeTaskPool; single-threaded they cost
// more than the PCIe transfers they feed).  for_each_chunk(n, fn): fn(0) .. fn(n-1), each once, on the workers and the caller.
class TaskPool {
  public:
    explicit TaskPool(unsigned n_workers) {
        for (unsigned i = 0; i < n_workers; ++i) workers_.emplace_back([this] { run(); });
    }
    ~TaskPool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
            ++gen_;
        }
        cv_.notify_all();
        for (std::thread& t : workers_) t.join();
    }
    TaskPool(const TaskPool&) = delete;
    TaskPool& operator=(const TaskPool&) = delete;
    unsigned threads() const { return (unsigned)workers_.size() + 1u; }
    void for_each_chunk(uint32_t n_chunks, const std::function<void(uint32_t)>& fn) {
        if (n_chunks <= 1 || workers_.empty()) {
            for (uint32_t c = 0; c < n_chunks; ++c) fn(c);
            return;
        }
        {
            // Nobody may be inside work() while the job's fields change: a worker that woke late for the job BEFORE (after its caller
            // had already returned: active_ was 0 then, the worker still on its way out of cv_.wait) would draw a stale index from
            // next_, compare it with the new total_ and run a chunk of this job a second time -- done_ overshoots total_ and the wait
            // below never ends.  Such a worker finds next_ >= total_ of the finished job and leaves at once: wait for it here.
            std::unique_lock<std::mutex> lk(m_);
            cv_done_.wait(lk, [&] { return active_ == 0; });
            job_ = &fn;
            total_ = n_chunks;
            done_.store(0, std::memory_order_relaxed);
            next_.store(0, std::memory_order_release);
            ++gen_;
        }
        cv_.notify_all();
        work();
        // done when every chunk is -- and when no worker is still inside work(): one that drew its last (out-of-range) index late
        // must not draw again from the counters of the NEXT job while they are being set up
        std::unique_lock<std::mutex> lk(m_);
        cv_done_.wait(lk, [&] { return done_.load(std::memory_order_acquire) == total_ && active_ == 0; });
    }

  private:
    void work() {
        for (;;) {
            const uint32_t c = next_.fetch_add(1, std::memory_order_acq_rel);
            if (c >= total_) return;
            (*job_)(c);
            if (done_.fetch_add(1, std::memory_order_acq_rel) + 1 == total_) {
                std::lock_guard<std::mutex> lk(m_);
                cv_done_.notify_all();
            }
        }
    }
    void run() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                ++active_;  // (under the mutex: the job's fields are set and stay put while anybody is active)
            }
            work();
            {
                std::lock_guard<std::mutex> lk(m_);
                if (--active_ == 0) cv_done_.notify_all();
            }
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, cv_done_;
    const std::function<void(uint32_t)>* job_ = nullptr;
    uint32_t total_ = 0;  // (written under m_ before next_ is reset: a worker that draws a chunk sees the job it belongs to)
    std::atomic<uint32_t> next_{0}, done_{0};
    uint64_t gen_ = 0;
    uint32_t active_ = 0;  // workers inside work() (under m_)
    bool stop_ = false;
};

class Mi355xPlugin {
  public:
    explicit Mi355xPlugin(int device = 0) {
        if (mi_ctx_create(device, nullptr, &ctx_) != MI_OK) throw std::runtime_error(std::string("mi_ctx_create: ") + mi_last_error_string(nullptr));
    }
    ~Mi355xPlugin() { mi_ctx_destroy(ctx_); }
    Mi355xPlugin(const Mi355xPlugin&) = delete;
    Mi355xPlugin& operator=(const Mi355xPlugin&) = delete;

    // A hierarchy the stock systems should keep (mi_hierarchy_advice_for: no level wider than a wave -- a chain, a rope, one rig -- where
    // the device has nothing to run side by side and a CPU core is faster; transform_hierarchy.rs's `chain`: 0.8 ms against 0.05):
    // the plugin then leaves mark_dirty_trees / propagate_parent_transforms / sync_simple_transforms to the host (below) and hands
    // the GlobalTransforms to the visibility stage.  set_keep_narrow_hierarchies_on_host(false): everything on the device (tests).
    bool transforms_on_host() const { return transforms_on_host_; }
    void set_keep_narrow_hierarchies_on_host(bool on) { allow_host_transforms_ = on; seen_version_ = 0; }

    // TransformSystems::Propagate
    void propagate_transforms(World& w) {
        sync_structure(w);
        const uint32_t n = (uint32_t)entity_of_row_.size();
        if (n == 0) return;
        if (transforms_on_host_) {
            stock_propagate_transforms(w);
            return;
        }
        // Changed<Transform> rows: sparse upload (also raises their "changed" byte)
        std::vector<uint32_t> rows;
        std::vector<float> t, r, s;
        for (uint32_t row = 0; row < n; ++row) {
            const uint32_t i = entity_of_row_[row].index;
            if (!w.moved_[i]) continue;
            const Transform& tr = w.transform_[i];
            rows.push_back(row);
            t.insert(t.end(), {tr.translation.x, tr.translation.y, tr.translation.z});
            r.insert(r.end(), {tr.rotation.x, tr.rotation.y, tr.rotation.z, tr.rotation.w});
            s.insert(s.end(), {tr.scale.x, tr.scale.y, tr.scale.z});
        }
        check(mi_upload_transforms_indexed(ctx_, (uint32_t)rows.size(), rows.data(), t.data(), r.data(), s.data()));
        if (rows.empty()) {  // keep "nothing changed" distinct from "no change information" (= all dirty)
            const uint8_t zero = 0;
            check(mi_upload_changed(ctx_, 0, 1, &zero));
        }
        check(mi_propagate(ctx_, w.static_transform_optimizations ? MI_PROPAGATE_STATIC_OPT : 0u));
        // write back exactly the rows whose tick the reference would bump
        uint32_t count = 0;
        std::vector<uint32_t> crow(n);
        std::vector<float> cg(12 * (size_t)n);
        check(mi_download_changed_global_transforms(ctx_, crow.data(), cg.data(), n, &count));
        for (uint32_t k = 0; k < count; ++k) {
            const uint32_t i = entity_of_row_[crow[k]].index;
            std::memcpy(w.global_[i].cols, &cg[12 * (size_t)k], 48);
            w.global_changed_[i] = 1;
            w.touch(i);
        }
    }

    // ==================================================================================================================
    // The FUSED form -- what Mi355xRenderPrepPlugin { fused: true } registers: the whole render-prep frame is ONE device
    // round trip.  frame() is the system placed in TransformSystems::Propagate (rust/bevy_mi355x/src/lib.rs: mi_fused_frame):
    //   in    the rows a Changed<Transform> query yields                          mi_upload_transforms_indexed
    //   run   propagate + reset + check_visibility + mark_newly_hidden + the       mi_propagate_and_cull_views(MI_CULL_CHANGED_ROWS |
    //         gather of the visible lights + assign_objects_to_clusters            MI_CULL_WITH_CLUSTERS | MI_CULL_END_FRAME): one launch for
    //                                                                              flat scenes, the tile launches + the cull with a hierarchy
    //   out   changed GlobalTransforms, every view's VisibleEntities list,         ONE mi_download_frame_results, in place: one packing
    //         the cluster lists                                                    launch, one device wait, no copy
    // and the ECS writes follow where the stock systems make them: GlobalTransform at once; ViewVisibility (set_visible on the
    // listed rows, between the stock reset_view_visibility and mark_newly_hidden_entities_invisible, which stay registered) and
    // VisibleEntities in VisibilitySystems::CheckVisibility; Clusters in front of SimulationLightSystems::AssignLightsToClusters.
    // Same World afterwards as the three-system form below (tests/cpp/host_systems_test.cpp runs the reference's system tests
    // against both).  Frames in which the structure changed (spawn / despawn / ChildOf) or a Visibility component was written pay
    // the slow path once (full column upload, InheritedVisibility round trip).
    // ==================================================================================================================
    struct FrameOutput {
        double gather_s = 0, device_s = 0, apply_s = 0;    // ECS -> staging; the library calls (upload .. results); ECS writes
        std::vector<std::vector<Entity>> visible_entities;  // per view: VisibleEntities::get(class 0), ascending by Entity
        bool has_clusters = false;
        Clusters clusters;
        bool has_light_visibility = false;
        LightVisibility light_visibility;
        uint32_t changed_global_transforms = 0;  // rows written back this frame
        uint32_t device_waits = 0;               // host waits for the device this frame (1 in the steady state)
    };
    // `shadows` != nullptr: bevy_light's shadow-view systems are part of the schedule (SimulationLightSystems::CheckLightVisibility):
    // a World with shadow-mapped lights then takes a second device call per frame -- the shadow views depend on what the cameras' pass
    // found (which lights some camera sees, lib.rs:563-566; a directional light's own ViewVisibility, :400) and on frusta the stock
    // light systems derive from this frame's GlobalTransforms.
    struct ShadowSetup {
        std::optional<ShadowLodOrigin> lod_origin;
    };
    FrameOutput frame(World& w, const std::vector<View>& views, const ClusterCamera* cam = nullptr, const ShadowSetup* shadows = nullptr) {
        FrameOutput out;
        const bool rebuilt = sync_structure(w);
        if (rebuilt) out.device_waits += 1;  // (the rebuild path synchronises in mi_columns_resize / its uploads)
        const uint32_t n = (uint32_t)entity_of_row_.size();
        if (n == 0) return out;
        if (transforms_on_host_) {
            // the stock transform systems keep this World (propagate_transforms above); visibility, light visibility and clusters follow
            // as the systems they are, on the device, over the GlobalTransforms the host computed
            propagate_transforms(w);
            if (rebuilt || seen_visibility_ != w.visibility_version_) {
                visibility_propagate(w);
                seen_visibility_ = w.visibility_version_;
            }
            const bool light_pass_h = shadows != nullptr && !views.empty() && w.n_shadow_lights_ != 0;
            if (!views.empty()) {
                check_visibility(w, views, /*close_frame=*/!light_pass_h);
                for (uint32_t v = 0; v < views.size(); ++v) out.visible_entities.push_back(visible_entities(v));
                if (light_pass_h) {
                    out.light_visibility = check_light_mesh_visibility(w, views, out.visible_entities, shadows->lod_origin);
                    out.has_light_visibility = true;
                } else if (shadows) out.has_light_visibility = true;
            }
            if (cam != nullptr && !views.empty()) {
                out.clusters = assign_objects_to_clusters(w, *cam);
                out.has_clusters = true;
            }
            for (uint32_t row = 0; row < n; ++row) out.changed_global_transforms += w.global_changed_[entity_of_row_[row].index] ? 1u : 0u;
            return out;
        }
        // InheritedVisibility is an input of the cull: recomputed (on the device) only in frames that wrote a Visibility
        if (rebuilt || seen_visibility_ != w.visibility_version_) {
            visibility_propagate(w);
            seen_visibility_ = w.visibility_version_;
            out.device_waits += 2;
        }
        upload_bounds(w);
        const auto t_start = std::chrono::steady_clock::now();
        // ---- in: Changed<Transform> rows (World::touched_ is what the query's change-tick scan yields), written straight into the
        //      library's pinned upload window: no Vec of our own, no staging copy
        // (in chunks of the touched list: each chunk counts its moved entities, then writes them behind the chunks in front of it)
        const uint32_t n_touched = (uint32_t)w.touched_.size();
        const uint32_t g_parts = n_touched >= 32768u ? pool().threads() : 1u;
        uint32_t g_count[64] = {0}, g_first[65] = {0};
        pool().for_each_chunk(g_parts, [&](uint32_t c) {
            const uint32_t a = (uint32_t)((uint64_t)n_touched * c / g_parts), b = (uint32_t)((uint64_t)n_touched * (c + 1) / g_parts);
            uint32_t m = 0;
            for (uint32_t j = a; j < b; ++j) m += w.moved_[w.touched_[j]];
            g_count[c] = m;
        });
        for (uint32_t c = 0; c < g_parts; ++c) g_first[c + 1] = g_first[c] + g_count[c];
        uint32_t n_in = g_first[g_parts];
        // every row moved: the dense window (filled by row, DMA straight from it); otherwise rows + values for the scatter kernel
        const bool dense = n_in == n;
        mi_upload_window win{};
        double commit_in_gather_s = 0;
        if (dense && n >= 65536u) {
            // The whole table, by row, a window at a time: window k crosses PCIe while this loop fills window k + 1, and the
            // library -- which sees a sequence of dense windows that carries every row -- computes each window's
            // GlobalTransforms at once and sends them back under the rest of the upload; the results call below finds them here.
            const uint32_t pieces = 8;
            for (uint32_t p = 0; p < pieces; ++p) {
                const uint32_t lo = (uint32_t)((uint64_t)n * p / pieces), hi = (uint32_t)((uint64_t)n * (p + 1) / pieces);
                check(mi_map_upload_window(ctx_, hi - lo, MI_UPLOAD_DENSE, &win));
                const uint32_t parts = pool().threads();
                pool().for_each_chunk(parts, [&](uint32_t c) {
                    const uint32_t a = lo + (uint32_t)((uint64_t)(hi - lo) * c / parts), b = lo + (uint32_t)((uint64_t)(hi - lo) * (c + 1) / parts);
                    for (uint32_t row = a; row < b; ++row) {
                        const Transform& tr = w.transform_[entity_of_row_[row].index];
                        std::memcpy(win.translation + 3 * (size_t)(row - lo), &tr.translation, 12);
                        std::memcpy(win.rotation + 4 * (size_t)(row - lo), &tr.rotation, 16);
                        std::memcpy(win.scale + 3 * (size_t)(row - lo), &tr.scale, 12);
                    }
                });
                const auto tc = std::chrono::steady_clock::now();
                check(mi_commit_upload_window(ctx_, &win, hi - lo, lo));
                commit_in_gather_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - tc).count();
            }
            n_in = 0;  // (committed)
            win = mi_upload_window{};
        } else
        check(mi_map_upload_window(ctx_, n_in, dense ? MI_UPLOAD_DENSE : 0u, &win));
        if (win.capacity) pool().for_each_chunk(g_parts, [&](uint32_t c) {
            const uint32_t a = (uint32_t)((uint64_t)n_touched * c / g_parts), b = (uint32_t)((uint64_t)n_touched * (c + 1) / g_parts);
            uint32_t k = g_first[c];
            for (uint32_t j = a; j < b; ++j) {
                const uint32_t i = w.touched_[j];
                if (!w.moved_[i]) continue;
                const Transform& tr = w.transform_[i];
                const uint32_t row = row_of_index_[i];
                const size_t at = dense ? row : k;
                if (!dense) win.rows[k] = row;
                std::memcpy(win.translation + 3 * at, &tr.translation, 12);
                std::memcpy(win.rotation + 4 * at, &tr.rotation, 16);
                std::memcpy(win.scale + 3 * at, &tr.scale, 12);
                ++k;
            }
        });
        const auto t_gathered = std::chrono::steady_clock::now();
        if (win.capacity) check(mi_commit_upload_window(ctx_, &win, n_in, 0));
        // (every row moved: the frame below is the all-rows frame -- no change bytes to raise, and nothing between the upload and the
        // frame that would have to wait for it: the library then runs upload, frame and result download in overlapping pieces)
        if (n_in == 0 && !dense) {  // keep "nothing changed" distinct from "no change information" (= all dirty)
            const uint8_t zero = 0;
            check(mi_upload_changed(ctx_, 0, 1, &zero));
        }
        // ---- the lights: rows like everything else, bound to the cluster stage by row (query order = Entity order here)
        uint32_t n_clusters = 0;
        mi_cluster_view cview{};
        const bool with_clusters = cam != nullptr && !views.empty() && sync_lights(w);
        if (with_clusters) {
            uint32_t tile[2], dims[3];
            if (mi_cluster_view_dims(cam->screen_width, cam->screen_height, cam->requested_dimensions, tile, dims) != MI_OK)
                throw std::runtime_error("mi_cluster_view_dims failed");
            n_clusters = dims[0] * dims[1] * dims[2];
            plane_storage_.assign((size_t)(dims[0] + dims[1] + dims[2] + 3) * 4, 0.0f);
            sphere_storage_.assign(lights_any_spot_ ? (size_t)n_clusters * 4 : 0, 0.0f);  // (read by the spot lights' cone test only)
            if (mi_cluster_view_build(cam->camera_affine, cam->clip_from_view, cam->frustum, cam->screen_width, cam->screen_height,
                                      cam->requested_dimensions, cam->first_slice_depth, cam->far_z, cam->layer_mask, plane_storage_.data(),
                                      lights_any_spot_ ? sphere_storage_.data() : nullptr, &cview) != MI_OK)
                throw std::runtime_error("mi_cluster_view_build failed");
            check(mi_cluster_upload_view(ctx_, &cview));
        }
        // ---- run: one call
        const uint32_t static_opt = w.static_transform_optimizations ? 1u : 0u;
        const bool light_pass = shadows != nullptr && !views.empty() && w.n_shadow_lights_ != 0;
        if (views.empty()) {
            check(mi_propagate(ctx_, (dense && n ? MI_PROPAGATE_ALL_DIRTY : 0u) | (static_opt ? MI_PROPAGATE_STATIC_OPT : 0u)));
        } else {
            camera_views(w, views);
            // (the light pass closes the frame when there is one: it ORs into ViewVisibility before MarkNewlyHiddenEntitiesInvisible)
            check(mi_propagate_and_cull_views(ctx_, mviews_.data(), (uint32_t)mviews_.size(),
                                              (dense && n ? 0u : MI_CULL_CHANGED_ROWS) | (light_pass ? 0u : MI_CULL_END_FRAME) | (static_opt ? MI_CULL_STATIC_OPT : 0u) |
                                                  (with_clusters ? MI_CULL_WITH_CLUSTERS : 0u)));
        }
        // ---- out: one call, one wait, read in place
        if (views.size() > MI_RESULTS_MAX_LISTS) throw std::runtime_error("more views than mi_download_frame_results takes lists");
        mi_visible_list lists[MI_RESULTS_MAX_LISTS] = {};
        for (size_t v = 0; v < views.size(); ++v) {
            lists[v].view = (uint32_t)v;
            lists[v].class_bit = 0;
            lists[v].capacity = n;
        }
        mi_frame_results fr{};
        fr.flags = MI_RESULTS_IN_PLACE | MI_RESULTS_CHANGED_ROWS | MI_RESULTS_CHANGED_GLOBALS | (with_clusters ? (MI_RESULTS_CLUSTERS | MI_RESULTS_CLUSTER_INDICES) : 0u);
        fr.n_lists = (uint32_t)views.size();
        fr.lists = lists;
        fr.changed_capacity = n;
        fr.cluster_capacity = with_clusters ? (uint64_t)light_entities_.size() * n_clusters : 0;  // an object is in a cluster at most once
        check(mi_download_frame_results(ctx_, &fr));
        out.device_waits += 1;
        const auto t_results = std::chrono::steady_clock::now();
        // ---- ECS writes (the pointers lead into the library's pinned window: read here, nothing kept)
        {  // TransformSystems::Propagate: Mut<GlobalTransform> of the returned rows (distinct entities: chunks write disjoint records)
            const uint32_t n_chg = fr.changed_count, parts = n_chg >= 32768u ? pool().threads() : 1u;
            std::vector<uint32_t> fresh[64];  // per chunk: entities this write-back is the first to touch since clear_trackers
            pool().for_each_chunk(parts, [&](uint32_t c) {
                const uint32_t a = (uint32_t)((uint64_t)n_chg * c / parts), b = (uint32_t)((uint64_t)n_chg * (c + 1) / parts);
                for (uint32_t k = a; k < b; ++k) {
                    const uint32_t i = entity_of_row_[fr.changed_rows[k]].index;
                    std::memcpy(w.global_[i].cols, fr.changed_global12 + 12 * (size_t)k, 48);
                    w.global_changed_[i] = 1;
                    if (!w.touched_flag_[i]) {
                        w.touched_flag_[i] = 1;
                        fresh[c].push_back(i);
                    }
                }
            });
            for (uint32_t c = 0; c < parts; ++c) w.touched_.insert(w.touched_.end(), fresh[c].begin(), fresh[c].end());
        }
        out.changed_global_transforms = fr.changed_count;
        if (!views.empty()) {  // VisibilitySystems::CheckVisibility, between the two stock systems
            w.reset_view_visibility();
            out.visible_entities.resize(views.size());
            for (size_t v = 0; v < views.size(); ++v) {
                std::vector<Entity>& list = out.visible_entities[v];
                list.reserve(lists[v].count);
                for (uint32_t k = 0; k < lists[v].count; ++k) {
                    const Entity e = entity_of_row_[lists[v].rows[k]];
                    w.set_visible(e);
                    list.push_back(e);
                }
            }
            if (light_pass) {  // SimulationLightSystems::CheckLightVisibility: after CheckVisibility, before MarkNewlyHiddenEntitiesInvisible
                shadow_views(w, views, out.visible_entities, shadows->lod_origin, /*end_frame=*/true, out.light_visibility);
                out.has_light_visibility = true;
                out.device_waits += 1;
                for (uint32_t row = 0; row < n; ++row)
                    if ((light_any_[row >> 5] >> (row & 31)) & 1u) w.set_visible(entity_of_row_[row]);  // lib.rs:499-510, 629, 723
            } else if (shadows) out.has_light_visibility = true;  // (no shadow-mapped light: nothing to list)
            w.mark_newly_hidden_entities_invisible();
        }
        if (with_clusters) {  // in front of SimulationLightSystems::AssignLightsToClusters
            out.has_clusters = true;
            Clusters& cl = out.clusters;
            std::memcpy(cl.dimensions, cview.dims, sizeof cl.dimensions);
            cl.farthest_z = fr.farthest_z;
            cl.total_index_count = fr.cluster_total;
            cl.clusterable_objects.resize(n_clusters);
            for (uint32_t c = 0; c < n_clusters; ++c) {
                ObjectsInCluster& o = cl.clusterable_objects[c];
                for (uint32_t i = fr.cluster_offsets[c]; i < fr.cluster_offsets[c + 1]; ++i) o.entities.push_back(light_entities_[fr.cluster_indices[i]]);
                std::memcpy(o.counts, fr.cluster_counts + 6 * (size_t)c, sizeof o.counts);
            }
        }
        const auto t_end = std::chrono::steady_clock::now();
        out.gather_s = std::chrono::duration<double>(t_gathered - t_start).count() - commit_in_gather_s;
        out.device_s = std::chrono::duration<double>(t_results - t_gathered).count() + commit_in_gather_s;
        out.apply_s = std::chrono::duration<double>(t_end - t_results).count();
        return out;
    }

    // VisibilitySystems::VisibilityPropagate
    void visibility_propagate(World& w) {
        sync_structure(w);
        const uint32_t n = (uint32_t)entity_of_row_.size();
        if (n == 0) return;
        std::vector<uint8_t> vis(n);
        for (uint32_t row = 0; row < n; ++row) {
            const World::Rec& e = w.rec_[entity_of_row_[row].index];
            vis[row] = e.has_visibility ? (uint8_t)e.visibility : (uint8_t)MI_VISIBILITY_NONE;
        }
        check(mi_upload_visibility(ctx_, 0, n, vis.data()));
        check(mi_visibility_propagate(ctx_));
        std::vector<uint8_t> inh(n);
        std::vector<uint32_t> chg((n + 31) / 32);
        check(mi_download_inherited_visibility(ctx_, 0, n, inh.data(), chg.data()));
        for (uint32_t row = 0; row < n; ++row) {
            World::Rec& e = w.rec_[entity_of_row_[row].index];
            if ((chg[row >> 5] >> (row & 31)) & 1u) { e.inherited = inh[row] != 0; e.inherited_changed = true; w.touch(entity_of_row_[row].index); ++w.bounds_version_; }
        }
    }

    // VisibilitySystems::CheckVisibility .. MarkNewlyHiddenEntitiesInvisible.  close_frame = false: the shadow-view systems follow
    // (check_light_mesh_visibility below), which OR into ViewVisibility before the newly hidden entities are marked.
    void check_visibility(World& w, const std::vector<View>& views, bool close_frame = true) {
        sync_structure(w);
        const uint32_t n = (uint32_t)entity_of_row_.size();
        if (n == 0 || views.empty()) return;
        upload_bounds(w);
        camera_views(w, views);
        check(mi_cull_views(ctx_, mviews_.data(), (uint32_t)mviews_.size(), MI_CULL_BEGIN_FRAME | (close_frame ? MI_CULL_END_FRAME : 0u)));
        fetch_view_visibility(w);
    }
    // SimulationLightSystems::CheckLightVisibility = check_dir_light_mesh_visibility + check_point_light_mesh_visibility
    // (crates/bevy_light/src/lib.rs:342-515, 517-757) followed by MarkNewlyHiddenEntitiesInvisible, after check_visibility(.., false).
    // camera_visible_entities = the cameras' VisibleEntities of this frame (which lights some camera sees, lib.rs:563-566).
    LightVisibility check_light_mesh_visibility(World& w, const std::vector<View>& views, const std::vector<std::vector<Entity>>& camera_visible_entities,
                                                const std::optional<ShadowLodOrigin>& lod_origin) {
        LightVisibility out;
        sync_structure(w);
        if (entity_of_row_.empty()) return out;
        upload_bounds(w);
        shadow_views(w, views, camera_visible_entities, lod_origin, /*end_frame=*/true, out);
        fetch_view_visibility(w);
        return out;
    }
    // VisibleEntities::get(class) of one view: entities in ascending Entity order (visibility/mod.rs:861-874)
    std::vector<Entity> visible_entities(uint32_t view) {
        const uint32_t n = (uint32_t)entity_of_row_.size();
        std::vector<uint32_t> rows(n);
        uint32_t count = 0;
        check(mi_download_visible_entities(ctx_, view, 0, nullptr, rows.data(), n, &count));
        std::vector<Entity> out;
        for (uint32_t k = 0; k < count; ++k) out.push_back(entity_of_row_[rows[k]]);
        return out;
    }

    // RenderSystems::PrepareResources, multidrawable part of batch_and_prepare_binned_render_phase
    // (gpu_preprocessing.rs:2360-2447): the view's visible mesh instances -> PreprocessWorkItems, per-bin
    // IndirectParametersMetadata and batch-set records, built on the device from the VisibleEntities list.
    // Batch sets and bins are visited in key order.  Call after check_visibility.
    PhaseBatches batch_multidrawables(World& w, uint32_t view, const mi_batch_initial* initial = nullptr) {
        sync_structure(w);
        sync_binning(w);
        PhaseBatches out;
        check(mi_batch_build(ctx_, view, 0, initial));
        check(mi_batch_download_totals(ctx_, &out.totals));
        for (uint32_t c = 0; c < 2; ++c) {
            uint32_t k = 0;
            out.work_items[c].resize(out.totals.work_item_len[c]);
            check(mi_batch_download(ctx_, MI_BATCH_WORK_ITEMS, c, out.work_items[c].data(), (uint32_t)out.work_items[c].size(), &k));
            out.metadata[c].resize(out.totals.indirect_parameters_len[c]);
            check(mi_batch_download(ctx_, MI_BATCH_INDIRECT_PARAMETERS_METADATA, c, out.metadata[c].data(), (uint32_t)out.metadata[c].size(), &k));
            out.batch_sets[c].resize(out.totals.batch_set_len[c]);
            check(mi_batch_download(ctx_, MI_BATCH_SETS, c, out.batch_sets[c].data(), (uint32_t)out.batch_sets[c].size(), &k));
        }
        std::vector<mi_batch_set_record> recs(out.totals.n_records);
        uint32_t k = 0;
        check(mi_batch_download(ctx_, MI_BATCH_RECORDS, 0, recs.data(), (uint32_t)recs.size(), &k));
        for (const auto& r : recs)
            out.records.push_back(BatchSetRecord{set_keys_[r.set].first, r.indexed != 0, r.index, r.first_work_item_index, r.instance_count,
                                                 r.first_indirect_parameters_index, r.batch_count, r.first_output_mesh_uniform_index});
        std::vector<mi_bin_metadata> meta(bin_keys_.size());
        check(mi_batch_download(ctx_, MI_BATCH_BIN_METADATA, 0, meta.data(), (uint32_t)meta.size(), &k));
        for (size_t i = 0; i < meta.size(); ++i) out.bins.push_back(BatchBin{bin_keys_[i].first, bin_keys_[i].second, meta[i]});
        return out;
    }

    // SimulationLightSystems::AssignLightsToClusters: gathers the clusterable objects in query order (point lights;
    // assign.rs:190-296, clustered with GlobalTransform::from_translation(translation), :198) and fills Clusters.
    Clusters assign_objects_to_clusters(World& w, const ClusterCamera& cam) {
        // the gather of assign.rs:190-248: the visible point lights, then the visible spot lights, then -- only where they are
        // clustered at all: with storage buffers -- the visible rect lights, each kind in query (Entity) order
        std::vector<Entity> lights;
        std::vector<float> pos_range, spot_dir, spot_sin_cos;
        std::vector<uint8_t> types;
        bool any_spot = false;
        auto gather = [&](Entity e, float range, uint8_t type, float outer_angle) {
            const float* g = w.global_[e.index].cols;
            lights.push_back(e);
            pos_range.insert(pos_range.end(), {g[9], g[10], g[11], range});
            types.push_back(type);
            float d[3] = {0.f, 0.f, 0.f}, sc[2] = {0.f, 0.f};
            if (type == MI_OBJ_SPOT_LIGHT) {
                // GlobalTransform::back() = (matrix3 * Vec3::Z).normalize() (global_transform.rs:62-68, 206), glam's order of operations
                const float len = std::sqrt((g[6] * g[6] + g[7] * g[7]) + g[8] * g[8]), inv = 1.0f / len;
                d[0] = g[6] * inv, d[1] = g[7] * inv, d[2] = g[8] * inv;
                sc[0] = std::sin(outer_angle), sc[1] = std::cos(outer_angle);  // ops::sin_cos: the host's libm, like Bevy's
                any_spot = true;
            }
            spot_dir.insert(spot_dir.end(), d, d + 3);
            spot_sin_cos.insert(spot_sin_cos.end(), sc, sc + 2);
        };
        const std::vector<Entity> ents = w.entities();
        auto visible = [&](Entity e) { return (w.vv_[e.index] & 1u) != 0; };  // `.filter(|(.., visibility)| visibility.get())`, assign.rs:194
        for (Entity e : ents)
            if (w.rec_[e.index].point_light_range && visible(e)) gather(e, *w.rec_[e.index].point_light_range, MI_OBJ_POINT_LIGHT, 0.f);
        for (Entity e : ents)
            if (w.rec_[e.index].spot_light && visible(e)) gather(e, w.rec_[e.index].spot_light->first, MI_OBJ_SPOT_LIGHT, w.rec_[e.index].spot_light->second);
        if (w.supports_storage_buffers) {
            for (Entity e : ents)
                if (w.rec_[e.index].rect_light_range && visible(e)) gather(e, *w.rec_[e.index].rect_light_range, MI_OBJ_RECT_LIGHT, 0.f);
            // light probes behind the same gate (assign.rs:250-277), range = transform.radius_vec3a(Vec3A::ONE)
            for (Entity e : ents)
                if (w.rec_[e.index].light_probe && visible(e))
                    gather(e, w.global_[e.index].radius_of_unit_extents(), *w.rec_[e.index].light_probe ? MI_OBJ_REFLECTION_PROBE : MI_OBJ_IRRADIANCE_VOLUME, 0.f);
        }
        if (w.clustered_decals_are_usable)  // decals behind their own (assign.rs:279-296), range = transform.scale().length()
            for (Entity e : ents)
                if (w.rec_[e.index].decal && visible(e)) gather(e, w.global_[e.index].scale_length(), MI_OBJ_DECAL, 0.f);
        uint32_t tile[2], dims[3];
        if (mi_cluster_view_dims(cam.screen_width, cam.screen_height, cam.requested_dimensions, tile, dims) != MI_OK)
            throw std::runtime_error("mi_cluster_view_dims failed");
        const size_t C = (size_t)dims[0] * dims[1] * dims[2];
        std::vector<float> planes((size_t)(dims[0] + dims[1] + dims[2] + 3) * 4), spheres(any_spot ? C * 4 : 0);
        mi_cluster_view view;
        if (mi_cluster_view_build(cam.camera_affine, cam.clip_from_view, cam.frustum, cam.screen_width, cam.screen_height,
                                  cam.requested_dimensions, cam.first_slice_depth, cam.far_z, cam.layer_mask, planes.data(),
                                  any_spot ? spheres.data() : nullptr, &view) != MI_OK)  // (the cluster spheres: read by the spot lights' cone test)
            throw std::runtime_error("mi_cluster_view_build failed");
        std::vector<uint32_t> offsets(C + 1), counts(6 * C), indices(1);
        uint64_t total = 0;
        float farthest = 0.0f;
        int32_t rc = mi_cluster_assign(ctx_, &view, (uint32_t)lights.size(), pos_range.data(), types.data(), nullptr, spot_dir.data(), spot_sin_cos.data(),
                                       offsets.data(), nullptr, 0, counts.data(), &total, &farthest);
        check(rc);
        indices.resize(std::max<uint64_t>(total, 1));
        check(mi_cluster_download(ctx_, nullptr, indices.data(), indices.size(), nullptr, &total, nullptr));
        Clusters out;
        std::memcpy(out.dimensions, dims, sizeof dims);
        out.farthest_z = farthest;
        out.total_index_count = total;
        out.clusterable_objects.resize(C);
        for (size_t c = 0; c < C; ++c) {
            ObjectsInCluster& o = out.clusterable_objects[c];
            for (uint32_t i = offsets[c]; i < offsets[c + 1]; ++i) o.entities.push_back(lights[indices[i]]);
            std::memcpy(o.counts, &counts[6 * c], sizeof o.counts);
        }
        return out;
    }

  private:
    void check(int32_t rc) {
        if (rc == MI_OK) return;
        const std::string msg = mi_last_error_string(ctx_);
        if (rc == MI_ERR_MALFORMED_HIERARCHY) throw std::logic_error("malformed hierarchy (the reference panics here): " + msg);
        throw std::runtime_error("bevy_mi355x error " + std::to_string(rc) + ": " + msg);
    }
    // mi_view of every camera: frustum, RenderLayers, and -- with a VisibleEntityRanges resource -- the view's index / position
    void camera_views(const World& w, const std::vector<View>& views) {
        mviews_.resize(views.size());
        for (size_t v = 0; v < views.size(); ++v) {
            mi_view& m = mviews_[v];
            std::memset(&m, 0, sizeof(mi_view));
            std::memcpy(m.frustum, views[v].frustum, sizeof m.frustum);
            m.layer_mask = views[v].layer_mask;
            if (w.visible_entity_ranges_ && views[v].has_range_index) {
                m.flags |= MI_VIEW_FLAG_RANGES;
                std::memcpy(m.position, &views[v].position, 12);
            }
        }
    }
    void fetch_view_visibility(World& w) {  // the three-system form: the whole column and its change mask come back
        const uint32_t n = (uint32_t)entity_of_row_.size();
        std::vector<uint8_t> vv(n);
        std::vector<uint32_t> chg((n + 31) / 32);
        check(mi_download_view_visibility(ctx_, 0, n, vv.data(), chg.data()));
        for (uint32_t row = 0; row < n; ++row) {
            const uint32_t i = entity_of_row_[row].index;
            w.vv_[i] = vv[row];
            if ((chg[row >> 5] >> (row & 31)) & 1u) w.vv_changed_[i] = 1;
        }
    }
    // The frame's shadow views in the order the two systems visit them, one mi_check_light_mesh_visibility, the masks turned into the
    // lights' lists.  light_any_ = the rows set_visible() is called on.
    void shadow_views(World& w, const std::vector<View>& views, const std::vector<std::vector<Entity>>& camera_visible_entities,
                      const std::optional<ShadowLodOrigin>& lod_origin, bool end_frame, LightVisibility& out) {
        const uint32_t n = (uint32_t)entity_of_row_.size();
        const bool ranges = w.visible_entity_ranges_;
        std::vector<mi_view> sv;
        struct Slot { int kind; size_t light, a, b; };  // where view k's list goes: 0 directional [a = view][b = cascade], 1 point face a, 2 spot
        std::vector<Slot> slots;
        auto shadow_view = [&](const Frustum6& f, uint32_t kind, uint32_t layer_mask) -> mi_view& {
            sv.emplace_back();
            mi_view& m = sv.back();
            std::memset(&m, 0, sizeof m);
            std::memcpy(m.frustum, f.half_spaces, sizeof m.frustum);
            m.layer_mask = layer_mask;
            m.flags = kind;
            return m;
        };
        // check_dir_light_mesh_visibility: directional lights in query order; cascades keyed by camera view (lib.rs:377-424)
        for (Entity e : w.entities()) {
            const World::Rec& r = w.rec_[e.index];
            if (!r.directional_light) continue;
            out.directional.push_back({e, {}});
            if (!r.directional_light->shadow_maps_enabled || !(w.vv_[e.index] & 1u)) continue;  // lib.rs:400-404: entities.clear()
            auto& lists = out.directional.back().entities;
            lists.resize(r.directional_light->cascades.size());
            for (size_t v = 0; v < r.directional_light->cascades.size() && v < views.size(); ++v) {
                lists[v].resize(r.directional_light->cascades[v].size());
                for (size_t c = 0; c < r.directional_light->cascades[v].size(); ++c) {
                    mi_view& m = shadow_view(r.directional_light->cascades[v][c], MI_VIEW_KIND_CASCADE, r.render_layers);
                    if (ranges && views[v].has_range_index) {  // entity_is_in_range_of_view(entity, *view): the cascade's camera (lib.rs:437-443)
                        m.flags |= MI_VIEW_FLAG_RANGES;
                        std::memcpy(m.position, &views[v].position, 12);
                    }
                    slots.push_back({0, out.directional.size() - 1, v, c});
                }
            }
        }
        // check_point_light_mesh_visibility: the lights some camera sees, each once, in the order the views' lists yield them (lib.rs:563-569)
        std::vector<uint8_t> checked(w.rec_.size(), 0);
        for (const std::vector<Entity>& visible : camera_visible_entities)
            for (Entity e : visible) {
                if (checked[e.index]) continue;
                checked[e.index] = 1;
                const World::Rec& r = w.rec_[e.index];
                const bool point = r.point_light_range && r.point_shadows, spot = r.spot_light && r.spot_shadows;
                if (!point && !spot) continue;
                if (!(point ? r.point_shadows->shadow_maps_enabled : r.spot_shadows->shadow_maps_enabled)) continue;  // lib.rs:579-581, 674-676
                const float* g = w.global_[e.index].cols;
                const float range = point ? *r.point_light_range : r.spot_light->first;
                if (point) out.point.push_back({e, {}});
                else out.spot.push_back({e, {}});
                for (size_t f = 0; f < (point ? 6u : 1u); ++f) {
                    mi_view& m = shadow_view(point ? r.point_shadows->cubemap_frusta[f] : r.spot_shadows->frustum, MI_VIEW_KIND_CUBE_FACE_OR_SPOT, r.render_layers);
                    const float sphere[4] = {g[9], g[10], g[11], range};  // Sphere { center: transform.translation(), radius: range } (lib.rs:584-587)
                    std::memcpy(m.light_sphere, sphere, sizeof sphere);
                    if (ranges) {  // lib.rs:601-611: no origin, or an origin without an index -> ranged entities are culled
                        if (lod_origin && lod_origin->has_range_index) {
                            m.flags |= MI_VIEW_FLAG_RANGES;
                            std::memcpy(m.position, &lod_origin->position, 12);
                        } else m.flags |= MI_VIEW_FLAG_RANGES_NO_ORIGIN;
                    }
                    slots.push_back({point ? 1 : 2, point ? out.point.size() - 1 : out.spot.size() - 1, f, 0});
                }
            }
        out.n_shadow_views = (uint32_t)sv.size();
        const size_t words = ((size_t)n + 31) / 32;
        light_masks_.assign(std::max<size_t>(sv.size() * words, 1), 0u);
        light_any_.assign(std::max<size_t>(words, 1), 0u);
        check(mi_check_light_mesh_visibility(ctx_, sv.data(), (uint32_t)sv.size(), end_frame ? MI_CULL_END_FRAME : 0u, light_masks_.data(), light_any_.data()));
        for (size_t k = 0; k < slots.size(); ++k) {
            const Slot& sl = slots[k];
            std::vector<Entity>& list = sl.kind == 0 ? out.directional[sl.light].entities[sl.a][sl.b] : sl.kind == 1 ? out.point[sl.light].faces[sl.a] : out.spot[sl.light].entities;
            const uint32_t* mask = light_masks_.data() + k * words;
            for (size_t wd = 0; wd < words; ++wd)
                for (uint32_t bits = mask[wd]; bits; bits &= bits - 1) list.push_back(entity_of_row_[wd * 32 + (uint32_t)__builtin_ctz(bits)]);
            std::sort(list.begin(), list.end(), [](Entity a, Entity b) { return a.to_bits() < b.to_bits(); });  // sort_unstable (rows are in level order)
        }
    }
    // Entity -> row table.  Rebuilt (level order via mi_hierarchy_sort, full column upload) whenever entities were
    // spawned / despawned or a ChildOf changed; otherwise only dirty rows travel.
    bool sync_structure(World& w) {
        if (seen_version_ == w.structure_version_) return false;
        std::vector<Entity> ents = w.entities();
        // rows in Entity::to_bits order first, so sibling order and VisibleEntities order follow the key
        std::sort(ents.begin(), ents.end(), [](Entity a, Entity b) { return a.to_bits() < b.to_bits(); });
        const uint32_t n = (uint32_t)ents.size();
        std::vector<uint32_t> slot_of_index(w.rec_.size(), MI_NO_PARENT);
        for (uint32_t i = 0; i < n; ++i) slot_of_index[ents[i].index] = i;
        std::vector<uint32_t> parent(n, MI_NO_PARENT);
        for (uint32_t i = 0; i < n; ++i) {
            const auto& p = w.rec_[ents[i].index].parent;
            if (p) {
                if (!w.contains(*p)) throw std::logic_error("ChildOf points at a despawned entity");
                parent[i] = slot_of_index[p->index];
            }
        }
        std::vector<uint32_t> new_to_old(std::max(n, 1u)), pidx(std::max(n, 1u)), offs((size_t)n + 2);
        uint32_t n_levels = 0;
        int32_t rc = mi_hierarchy_sort(n, parent.data(), new_to_old.data(), pidx.data(), offs.data(), n + 2, &n_levels);
        if (rc == MI_ERR_MALFORMED_HIERARCHY) throw std::logic_error("malformed hierarchy (the reference panics here): cycle in ChildOf");
        if (rc != MI_OK) throw std::runtime_error("mi_hierarchy_sort failed");
        entity_of_row_.resize(n);
        for (uint32_t row = 0; row < n; ++row) entity_of_row_[row] = ents[new_to_old[row]];
        parent_row_.assign(pidx.begin(), pidx.begin() + n);
        {
            mi_hierarchy_advice advice{};
            if (mi_hierarchy_advice_for(n_levels, offs.data(), &advice) != MI_OK) throw std::runtime_error("mi_hierarchy_advice_for failed");
            transforms_on_host_ = allow_host_transforms_ && advice.keep_on_host != 0;
        }
        row_of_index_.assign(w.rec_.size(), MI_NO_PARENT);
        for (uint32_t row = 0; row < n; ++row) row_of_index_[entity_of_row_[row].index] = row;
        check(mi_columns_resize(ctx_, n));
        if (n) {
            std::vector<float> t(3 * (size_t)n), r(4 * (size_t)n), s(3 * (size_t)n), g(12 * (size_t)n);
            std::vector<uint8_t> changed(n), vv(n);
            std::vector<uint64_t> keys(n);
            for (uint32_t row = 0; row < n; ++row) {
                const World::Rec& e = w.rec_[entity_of_row_[row].index];
                const uint32_t ei = entity_of_row_[row].index;
                std::memcpy(&t[3 * (size_t)row], &w.transform_[ei].translation, 12);
                std::memcpy(&r[4 * (size_t)row], &w.transform_[ei].rotation, 16);
                std::memcpy(&s[3 * (size_t)row], &w.transform_[ei].scale, 12);
                std::memcpy(&g[12 * (size_t)row], w.global_[ei].cols, 48);
                changed[row] = (e.transform_changed || e.added || e.parent_changed || e.orphaned) ? 1 : 0;
                vv[row] = w.vv_[entity_of_row_[row].index];
                keys[row] = entity_of_row_[row].to_bits();
            }
            check(mi_upload_transforms(ctx_, 0, n, t.data(), r.data(), s.data()));
            check(mi_upload_global_transforms(ctx_, 0, n, g.data()));
            check(mi_upload_view_visibility(ctx_, 0, n, vv.data()));
            check(mi_upload_entity_keys(ctx_, 0, n, keys.data()));
            check(mi_upload_hierarchy(ctx_, n, n_levels > 1 ? pidx.data() : nullptr, offs.data(), n_levels));
            check(mi_upload_changed(ctx_, 0, n, changed.data()));
            bounds_dirty_ = true;
        }
        seen_version_ = w.structure_version_;
        upload_bounds(w);  // the flag byte carries InheritedVisibility: the device must start from the World's values
        lights_version_ = 0;  // rows were renumbered: the lights are bound again
        return true;
    }
    // mark_dirty_trees + propagate_parent_transforms + sync_simple_transforms (crates/bevy_transform/src/systems.rs:42-79, 111-306,
    // 506-748) as the stock systems run them, on the host: rows are in level order, so one sweep meets every parent before its
    // children.  Same rule as the kernels (kernels_tree.hip: node_apply) and the same glam operations (csrc/glam_math.h through
    // GlobalTransform::from / operator*), hence the same bits and the same change ticks -- tests/cpp/host_systems_test.cpp runs
    // `chain` both ways.  Then the GlobalTransform column goes to the device for the visibility stage (a narrow hierarchy is small).
    void stock_propagate_transforms(World& w) {
        const uint32_t n = (uint32_t)entity_of_row_.size();
        const bool static_opt = w.static_transform_optimizations;
        std::vector<uint8_t> tree(n, 0), bumped(n, 0);
        if (static_opt)  // mark_dirty_trees returns early unless the optimisation is enabled (systems.rs:131-133)
            for (uint32_t row = 0; row < n; ++row)
                if (w.moved_[entity_of_row_[row].index])
                    for (uint32_t r = row; r != MI_NO_PARENT && !tree[r]; r = parent_row_[r]) tree[r] = 1;
        for (uint32_t row = 0; row < n; ++row) {
            const uint32_t i = entity_of_row_[row].index, p = parent_row_[row];
            if (p == MI_NO_PARENT) {
                // a root with children: assigned unless its tree is static (systems.rs:522-530); a flat row: iff its own Transform
                // changed (sync_simple_transforms, :58-63) -- plain assignments, the tick moves
                const bool write = !w.rec_[i].children.empty() ? (!static_opt || tree[row]) : w.moved_[i] != 0;
                if (write) {
                    w.global_[i] = GlobalTransform::from(w.transform_[i]);
                    bumped[row] = 1;
                }
                continue;
            }
            if (static_opt && !tree[row] && !bumped[p]) continue;  // the static-scene rule (systems.rs:708-714); its subtree is not visited either
            const GlobalTransform nw = w.global_[entity_of_row_[p].index] * w.transform_[i];
            if (!(nw == w.global_[i])) {  // set_if_neq (systems.rs:719)
                w.global_[i] = nw;
                bumped[row] = 1;
            }
        }
        std::vector<float> g(12 * (size_t)n);
        for (uint32_t row = 0; row < n; ++row) {
            const uint32_t i = entity_of_row_[row].index;
            if (bumped[row]) {
                w.global_changed_[i] = 1;
                w.touch(i);
            }
            std::memcpy(&g[12 * (size_t)row], w.global_[i].cols, 48);
        }
        check(mi_upload_global_transforms(ctx_, 0, n, g.data()));
    }
    // The clusterable objects of the fused frame: point lights in query (Entity) order, each bound to its ROW -- the device takes
    // the centre from the row's GlobalTransform and gathers only the lights whose ViewVisibility::get() is true (assign.rs:190-296).
    // Re-uploaded only when a light was added / removed / changed its range or rows were renumbered.  Returns false without lights.
    bool sync_lights(World& w) {
        // (a probe or decal WITH a parent: its range follows every ancestor's Transform, which no version counts -- such a World's list
        // goes up every frame)
        if (lights_version_ == w.lights_version_ && lights_version_ != 0 && !probes_or_decals_with_parents_) return !light_entities_.empty();
        light_entities_.clear();
        std::vector<uint32_t> rows;
        // every light that could be gathered, in gather order (points, spots, rects where they are clustered at all: assign.rs:190-248);
        // the device leaves out the ones whose ViewVisibility::get() is false this frame
        std::vector<float> pos_range, spot_sin_cos;
        std::vector<uint8_t> types;
        lights_any_spot_ = false;
        std::vector<Entity> ents = w.entities();
        auto add = [&](Entity e, float range, uint8_t type, float outer_angle) {
            light_entities_.push_back(e);
            rows.push_back(row_of_index_[e.index]);
            pos_range.insert(pos_range.end(), {0.0f, 0.0f, 0.0f, range});  // the position (and a spot light's direction) comes from the row
            types.push_back(type);
            const bool spot = type == MI_OBJ_SPOT_LIGHT;
            spot_sin_cos.push_back(spot ? std::sin(outer_angle) : 0.0f);
            spot_sin_cos.push_back(spot ? std::cos(outer_angle) : 0.0f);
            lights_any_spot_ = lights_any_spot_ || spot;
        };
        for (Entity e : ents)
            if (w.rec_[e.index].point_light_range) add(e, *w.rec_[e.index].point_light_range, MI_OBJ_POINT_LIGHT, 0.f);
        for (Entity e : ents)
            if (w.rec_[e.index].spot_light) add(e, w.rec_[e.index].spot_light->first, MI_OBJ_SPOT_LIGHT, w.rec_[e.index].spot_light->second);
        // light probes and decals take their range from the GlobalTransform this frame computes (assign.rs:262, 287): From(Transform)
        // chained down the ChildOf links with the reference's own operators -- the products propagate_parent_transforms forms for the
        // entity, in its order, hence its bits -- known here, before the frame runs
        probes_or_decals_with_parents_ = false;
        auto range_rider = [&](Entity e, bool probe) {
            std::vector<const Transform*> chain;
            for (std::optional<Entity> cur = e; cur; cur = w.rec_[cur->index].parent) chain.push_back(&w.transform_[cur->index]);
            probes_or_decals_with_parents_ = probes_or_decals_with_parents_ || chain.size() > 1;
            GlobalTransform g = GlobalTransform::from(*chain.back());
            for (size_t i = chain.size() - 1; i-- > 0;) g = g * *chain[i];
            return probe ? g.radius_of_unit_extents() : g.scale_length();
        };
        if (w.supports_storage_buffers) {
            for (Entity e : ents)
                if (w.rec_[e.index].rect_light_range) add(e, *w.rec_[e.index].rect_light_range, MI_OBJ_RECT_LIGHT, 0.f);
            for (Entity e : ents)
                if (w.rec_[e.index].light_probe) add(e, range_rider(e, true), *w.rec_[e.index].light_probe ? MI_OBJ_REFLECTION_PROBE : MI_OBJ_IRRADIANCE_VOLUME, 0.f);
        }
        if (w.clustered_decals_are_usable)
            for (Entity e : ents)
                if (w.rec_[e.index].decal) add(e, range_rider(e, false), MI_OBJ_DECAL, 0.f);
        lights_version_ = w.lights_version_;
        if (light_entities_.empty()) {
            check(mi_cluster_bind_objects_to_row_list(ctx_, 0, nullptr));
            return false;
        }
        check(mi_cluster_upload_objects(ctx_, (uint32_t)light_entities_.size(), pos_range.data(), types.data(), nullptr, nullptr, spot_sin_cos.data()));
        check(mi_cluster_bind_objects_to_row_list(ctx_, (uint32_t)rows.size(), rows.data()));
        return true;
    }
    void upload_bounds(World& w) {
        const uint32_t n = (uint32_t)entity_of_row_.size();
        if (!bounds_dirty_ && seen_bounds_ == w.bounds_version_) return;
        seen_bounds_ = w.bounds_version_;
        std::vector<float> c(3 * (size_t)n, 0.f), h(3 * (size_t)n, 0.f), ranges(2 * (size_t)n, 0.f);
        std::vector<uint8_t> flags(n);
        std::vector<uint32_t> layers(n);
        for (uint32_t row = 0; row < n; ++row) {
            const World::Rec& e = w.rec_[entity_of_row_[row].index];
            // entities without the visibility components never enter the query; without Visibility they default visible
            flags[row] = (uint8_t)(((!e.has_visibility || e.inherited) ? MI_FLAG_INHERITED_VISIBLE : 0u) | (e.aabb ? MI_FLAG_HAS_AABB : 0u) |
                                   (e.no_frustum_culling ? MI_FLAG_NO_FRUSTUM_CULLING : 0u) |
                                   // the shadow views' query: With<Mesh3d>, Without<NotShadowCaster>, Without<DirectionalLight> (bevy_light/src/lib.rs:355-372)
                                   ((e.mesh3d && !e.not_shadow_caster && !e.directional_light) ? MI_FLAG_SHADOW_CASTER : 0u));
            layers[row] = e.render_layers;
            if (e.visibility_range) {  // Has<VisibilityRange>; the two bounds is_visible_at_all reads (range.rs:159-161); use_aabb (:255-263)
                flags[row] |= (uint8_t)(MI_FLAG_HAS_VISIBILITY_RANGE | (e.visibility_range->use_aabb ? MI_FLAG_RANGE_USE_AABB : 0u));
                ranges[2 * (size_t)row] = e.visibility_range->start_margin_start;
                ranges[2 * (size_t)row + 1] = e.visibility_range->end_margin_end;
            }
            if (e.aabb) { std::memcpy(&c[3 * (size_t)row], &e.aabb->center, 12); std::memcpy(&h[3 * (size_t)row], &e.aabb->half_extents, 12); }
            else if (e.point_light_range || e.spot_light) {
                // a point light is culled by its bounding Sphere { GlobalTransform::translation, range } (update_point_light_bounding_spheres,
                // point_light.rs:195-208): a sphere that follows the row's own GlobalTransform on the device -- a moving light costs nothing here
                flags[row] |= MI_FLAG_HAS_SPHERE;
                const uint32_t at_translation = MI_SPHERE_AT_TRANSLATION;
                h[3 * (size_t)row] = e.point_light_range ? *e.point_light_range : e.spot_light->first;  // (spot_light.rs:221-234: the same Sphere)
                std::memcpy(&h[3 * (size_t)row + 1], &at_translation, 4);
            }
        }
        check(mi_upload_bounds(ctx_, 0, n, c.data(), h.data(), flags.data(), layers.data()));
        // Option<Res<VisibleEntityRanges>>: no resource = no range column (visibility/mod.rs:814-816)
        check(mi_upload_visibility_ranges(ctx_, 0, n, w.visible_entity_ranges_ ? ranges.data() : nullptr));
        bounds_dirty_ = false;
    }

    // Flattens the World's MeshBinning components into the row columns and bin tables mi_batch_* take: batch sets in key
    // order, a set's bins in bin-key order (RenderBinIndex = rank of the bin key, metadata in the same order).
    void sync_binning(World& w) {
        if (seen_binning_ == w.binning_version_ && seen_binning_structure_ == w.structure_version_) return;
        const uint32_t n = (uint32_t)entity_of_row_.size();
        std::map<std::pair<uint64_t, bool>, std::map<uint64_t, uint32_t>> sets;  // (set key, indexed) -> bin key -> RenderBinIndex
        for (uint32_t row = 0; row < n; ++row) {
            const auto& b = w.rec_[entity_of_row_[row].index].binning;
            if (b) sets[{b->batch_set_key, b->indexed}][b->bin_key] = 0;
        }
        set_keys_.clear();
        bin_keys_.clear();
        std::vector<uint8_t> indexed;
        std::vector<uint32_t> table_off{0}, meta_off{0}, table;
        std::vector<mi_bin_metadata> meta;
        std::map<std::pair<uint64_t, bool>, uint32_t> set_id;
        for (auto& [key, bins] : sets) {
            set_id[key] = (uint32_t)set_keys_.size();
            set_keys_.push_back(key);
            indexed.push_back(key.second ? 1 : 0);
            uint32_t k = 0;
            for (auto& [bin_key, idx] : bins) {
                idx = k;
                table.push_back(k);
                meta.push_back(mi_bin_metadata{k, k, 0});
                bin_keys_.push_back({key.first, bin_key});
                ++k;
            }
            table_off.push_back((uint32_t)table.size());
            meta_off.push_back((uint32_t)meta.size());
        }
        std::vector<uint32_t> rs(n, MI_NO_BATCH_SET), rb(n, 0), ri(n, 0);
        for (uint32_t row = 0; row < n; ++row) {
            const auto& b = w.rec_[entity_of_row_[row].index].binning;
            if (!b) continue;
            rs[row] = set_id[{b->batch_set_key, b->indexed}];
            rb[row] = sets[{b->batch_set_key, b->indexed}][b->bin_key];
            ri[row] = b->input_uniform_index;
        }
        if (n) check(mi_batch_upload_rows(ctx_, 0, n, rs.data(), rb.data(), ri.data()));
        else check(mi_batch_upload_rows(ctx_, 0, 0, nullptr, nullptr, nullptr));
        check(mi_batch_upload_sets(ctx_, (uint32_t)set_keys_.size(), indexed.data(), table_off.data(), table.data(), meta_off.data(), meta.data()));
        seen_binning_ = w.binning_version_;
        seen_binning_structure_ = w.structure_version_;
    }

    mi_ctx* ctx_ = nullptr;
    std::vector<float> plane_storage_;
    std::vector<uint8_t> scratch_ones_;
    std::unique_ptr<TaskPool> pool_;
    TaskPool& pool() {  // a few threads, started at the first big frame (half the hardware threads, eight at most)
        if (!pool_) pool_.reset(new TaskPool(std::min(8u, std::max(2u, std::thread::hardware_concurrency() / 2u)) - 1u));
        return *pool_;
    }
    std::vector<mi_view> mviews_;
    std::vector<uint32_t> light_masks_, light_any_;  // mi_check_light_mesh_visibility's outputs
    std::vector<Entity> light_entities_;
    bool lights_any_spot_ = false;
    bool probes_or_decals_with_parents_ = false;  // (sync_lights: such a World's object list is gathered anew every frame)
    std::vector<float> sphere_storage_;
    std::vector<uint32_t> row_of_index_;
    uint64_t lights_version_ = 0, seen_visibility_ = 0, seen_bounds_ = 0;
    std::vector<std::pair<uint64_t, bool>> set_keys_;
    std::vector<std::pair<uint64_t, uint64_t>> bin_keys_;  // per metadata entry: (batch set key, bin key)
    uint64_t seen_binning_ = 0, seen_binning_structure_ = 0;
    std::vector<Entity> entity_of_row_;
    std::vector<uint32_t> parent_row_;  // per row: its parent's row (MI_NO_PARENT for roots); rows are in level order
    uint64_t seen_version_ = 0;
    bool bounds_dirty_ = true;
    bool transforms_on_host_ = false, allow_host_transforms_ = true;
};

}  // namespace bevy_mi355x
