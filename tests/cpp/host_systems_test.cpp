// host_systems_test.cpp -- the reference's own system tests, restated against the C++ host layer
// (bevy_amd/host/bevy_mi355x_host.hpp) that sits on the C ABI.  Each test names the reference test it follows:
//   crates/bevy_transform/src/systems.rs:826-1221          propagate semantics (exact assert_eq! on GlobalTransform)
//   crates/bevy_camera/src/visibility/mod.rs:950-1279       InheritedVisibility propagation + change detection
//   crates/bevy_camera/src/visibility/mod.rs:1313-1448      ViewVisibility 2-bit lifecycle over frames
//   benches/benches/bevy_camera/primitives.rs:41-52         an OBB inside a perspective frustum is visible
// Every test runs against BOTH forms of the plugin: the three systems of round 2 (propagate_transforms / visibility_propagate /
// check_visibility / assign_objects_to_clusters, a device wait each) and the fused frame (Mi355xPlugin::frame: one upload, one
// mi_propagate_and_cull_views, one mi_download_frame_results in place -- the call sequence bench.py times).
// Needs an MI355X (the systems run on the device).  Exit code 0 = all passed.  `host_systems_test --bench` times both call
// sequences on the BASELINE metric scene instead (bench.py's end_to_end block).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>

#include "../../bevy_amd/host/bevy_mi355x_host.hpp"
#include "../../bevy_amd/host/bevy_mi355x_sharded.hpp"

using namespace bevy_mi355x;

static int g_failed = 0, g_checks = 0;
#define CHECK(cond, msg)                                                                     \
    do {                                                                                     \
        ++g_checks;                                                                          \
        if (!(cond)) { ++g_failed; std::printf("  FAILED %s:%d: %s -- %s\n", __FILE__, __LINE__, #cond, msg); } \
    } while (0)

static bool g_fused = false;  // which form of the plugin the tests drive

// the chained transform systems of one schedule.run()
static void propagate(Mi355xPlugin& p, World& w) {
    if (g_fused) p.frame(w, {});
    else p.propagate_transforms(w);
}
static void run(Mi355xPlugin& p, World& w) {
    propagate(p, w);
    w.clear_trackers();
}
static void visibility_propagate(Mi355xPlugin& p, World& w) {
    if (g_fused) p.frame(w, {});  // (a frame without views: propagate + InheritedVisibility, nothing to cull)
    else p.visibility_propagate(w);
}
// PostUpdate of one frame with cameras: returns the cameras' VisibleEntities
static std::vector<std::vector<Entity>> post_update(Mi355xPlugin& p, World& w, const std::vector<View>& views) {
    if (g_fused) return p.frame(w, views).visible_entities;
    p.propagate_transforms(w);
    p.visibility_propagate(w);
    p.check_visibility(w, views);
    std::vector<std::vector<Entity>> out;
    for (uint32_t v = 0; v < views.size(); ++v) out.push_back(p.visible_entities(v));
    return out;
}

// systems.rs:826-886
static void correct_parent_removed() {
    World world;
    Mi355xPlugin plugin;
    auto offset_global_transform = [](float o) { return GlobalTransform::from(Transform::from_xyz(o, o, o)); };
    auto offset_transform = [](float o) { return Transform::from_xyz(o, o, o); };
    Entity root = world.spawn(offset_transform(3.3f));
    Entity parent = world.spawn(offset_transform(4.4f));
    Entity child = world.spawn(offset_transform(5.5f));
    world.add_child(root, parent);
    world.add_child(parent, child);
    run(plugin, world);
    CHECK(world.global_transform(parent) == offset_global_transform(4.4f + 3.3f), "GlobalTransform wasn't updated");
    world.remove_parent(parent);
    run(plugin, world);
    CHECK(world.global_transform(parent) == offset_global_transform(4.4f), "orphaned entity wasn't updated properly");
    world.remove_parent(child);
    run(plugin, world);
    CHECK(world.global_transform(child) == offset_global_transform(5.5f), "orphaned entity wasn't updated properly");
}

// systems.rs:888-925 (and :928-965, the command-buffer variant: same world after apply)
static void did_propagate() {
    World world;
    Mi355xPlugin plugin;
    world.spawn(Transform::from_xyz(1.0f, 0.0f, 0.0f));  // root entity without children
    Entity parent = world.spawn(Transform::from_xyz(1.0f, 0.0f, 0.0f));
    Entity c0 = world.spawn_child(parent, Transform::from_xyz(0.0f, 2.0f, 0.0f));
    Entity c1 = world.spawn_child(parent, Transform::from_xyz(0.0f, 0.0f, 3.0f));
    run(plugin, world);
    CHECK(world.global_transform(c0) == GlobalTransform::from_xyz(1.0f, 0.0f, 0.0f) * Transform::from_xyz(0.0f, 2.0f, 0.0f), "child 0");
    CHECK(world.global_transform(c1) == GlobalTransform::from_xyz(1.0f, 0.0f, 0.0f) * Transform::from_xyz(0.0f, 0.0f, 3.0f), "child 1");
}

// systems.rs:967-1046
static void correct_children() {
    World world;
    Mi355xPlugin plugin;
    Entity parent = world.spawn(Transform::from_xyz(1.0f, 0.0f, 0.0f));
    std::vector<Entity> children = {world.spawn_child(parent, Transform::from_xyz(0.0f, 2.0f, 0.0f)),
                                    world.spawn_child(parent, Transform::from_xyz(0.0f, 3.0f, 0.0f))};
    run(plugin, world);
    CHECK(world.children(parent) == children, "Children of parent");
    world.add_child(children[1], children[0]);
    run(plugin, world);
    CHECK(world.children(parent) == std::vector<Entity>{children[1]}, "Children of parent after re-parenting");
    CHECK(world.children(children[1]) == std::vector<Entity>{children[0]}, "Children of child 1");
    // and the values follow the new hierarchy: child0 = parent * child1 * child0
    CHECK(world.global_transform(children[0]) ==
              (GlobalTransform::from_xyz(1.0f, 0.0f, 0.0f) * Transform::from_xyz(0.0f, 3.0f, 0.0f)) * Transform::from_xyz(0.0f, 2.0f, 0.0f),
          "re-parented child follows its new parent");
    CHECK(world.despawn(children[0]), "despawn");
    run(plugin, world);
    CHECK(world.children(parent) == std::vector<Entity>{children[1]}, "Children of parent after despawn");
}

// systems.rs:1048-1097
static void correct_transforms_when_no_children() {
    World world;
    Mi355xPlugin plugin;
    const Vec3 translation{1.0f, 0.0f, 0.0f};
    Entity parent = world.spawn(Transform::from_translation(translation));
    Entity child = world.spawn_child(parent, Transform::identity());
    Entity grandchild = world.spawn_child(child, Transform::identity());
    run(plugin, world);
    CHECK(world.children(parent) == std::vector<Entity>{child}, "children of parent");
    CHECK(world.children(child) == std::vector<Entity>{grandchild}, "children of child");
    run(plugin, world);
    for (Entity e : world.entities()) CHECK(world.global_transform(e) == GlobalTransform::from_translation(translation), "every GlobalTransform");
}

// systems.rs:1099-1165 (#[should_panic])
static void panic_when_hierarchy_cycle() {
    World world;
    Mi355xPlugin plugin;
    Entity child = world.spawn(Transform::identity());
    Entity grandchild = world.spawn_child(child, Transform::identity());
    Entity top = world.spawn(Transform::identity());
    world.add_child(top, child);
    world.set_child_of_unchecked(child, grandchild);  // ChildOf of child and grandchild now point at each other
    bool panicked = false;
    try {
        run(plugin, world);
    } catch (const std::logic_error&) {
        panicked = true;
    }
    CHECK(panicked, "a hierarchy cycle must be reported (the reference panics)");
}

// systems.rs:1167-1221
static void global_transform_should_not_be_overwritten_after_reparenting() {
    World world;
    Mi355xPlugin plugin;
    const Vec3 translation{1.0f, 1.0f, 1.0f};
    Entity parent = world.spawn(Transform::from_translation(translation));
    Entity child = world.spawn(Transform::from_translation(translation));
    world.add_child(parent, child);
    run(plugin, world);
    const GlobalTransform pg = world.global_transform(parent), cg = world.global_transform(child);
    CHECK(std::fabs(pg.translation().x - 1.0f) < 0.1f && std::fabs(pg.translation().y - 1.0f) < 0.1f, "parent translation");
    CHECK(std::fabs(cg.translation().x - 2.0f) < 0.1f && std::fabs(cg.translation().z - 2.0f) < 0.1f, "child translation");
    world.remove_parent(child);
    world.add_child(parent, child);
    run(plugin, world);
    CHECK(pg == world.global_transform(parent), "parent GlobalTransform unchanged");
    CHECK(cg == world.global_transform(child), "child GlobalTransform unchanged");
}

// change ticks: set_if_neq leaves unchanged descendants untouched (systems.rs:719) but roots with children are
// re-assigned every run unless StaticTransformOptimizations is enabled (systems.rs:522-530)
static void change_ticks_follow_set_if_neq() {
    World world;
    Mi355xPlugin plugin;
    Entity root = world.spawn(Transform::from_xyz(1, 0, 0));
    Entity a = world.spawn_child(root, Transform::from_xyz(0, 1, 0));
    Entity b = world.spawn_child(a, Transform::from_xyz(0, 0, 1));
    Entity flat = world.spawn(Transform::from_xyz(5, 5, 5));
    run(plugin, world);
    propagate(plugin, world);  // nothing changed
    CHECK(world.global_transform_changed(root), "root is re-assigned each run");
    CHECK(!world.global_transform_changed(a) && !world.global_transform_changed(b), "set_if_neq: equal value, no tick");
    CHECK(!world.global_transform_changed(flat), "flat entity without Changed<Transform> is not touched");
    world.clear_trackers();
    world.transform_mut(a).translation.y = 2.0f;
    propagate(plugin, world);
    CHECK(world.global_transform_changed(a) && world.global_transform_changed(b), "moved subtree");
    CHECK(world.global_transform(b) == (GlobalTransform::from_xyz(1, 0, 0) * Transform::from_xyz(0, 2, 0)) * Transform::from_xyz(0, 0, 1), "value");
    world.clear_trackers();
    world.static_transform_optimizations = true;
    propagate(plugin, world);
    CHECK(!world.global_transform_changed(root), "static scene optimisation: clean tree is skipped");
}

// visibility/mod.rs:950-1037
static void visibility_propagation() {
    World w;
    Mi355xPlugin plugin;
    auto spawn = [&](Visibility v) { Entity e = w.spawn(); w.insert_visibility(e, v); return e; };
    Entity root1 = spawn(Visibility::Hidden), root1_child1 = spawn(Visibility::Inherited), root1_child2 = spawn(Visibility::Hidden);
    Entity r1c1g = spawn(Visibility::Inherited), r1c2g = spawn(Visibility::Inherited);
    w.add_children(root1, {root1_child1, root1_child2});
    w.add_child(root1_child1, r1c1g);
    w.add_child(root1_child2, r1c2g);
    Entity root2 = spawn(Visibility::Inherited), root2_child1 = spawn(Visibility::Inherited), root2_child2 = spawn(Visibility::Hidden);
    Entity r2c1g = spawn(Visibility::Inherited), r2c2g = spawn(Visibility::Inherited);
    w.add_children(root2, {root2_child1, root2_child2});
    w.add_child(root2_child1, r2c1g);
    w.add_child(root2_child2, r2c2g);
    visibility_propagate(plugin, w);
    for (Entity e : {root1, root1_child1, root1_child2, r1c1g, r1c2g}) CHECK(!w.inherited_visibility(e), "invisibility propagates down tree from root");
    CHECK(w.inherited_visibility(root2) && w.inherited_visibility(root2_child1) && w.inherited_visibility(r2c1g), "visibility propagates down tree from root");
    CHECK(!w.inherited_visibility(root2_child2), "local invisibility is preserved");
    CHECK(!w.inherited_visibility(r2c2g), "child's invisibility propagates down to grandchild");
}

// visibility/mod.rs:1189-1262
static void visibility_propagation_change_detection() {
    World w;
    Mi355xPlugin plugin;
    auto spawn = [&](Visibility v) { Entity e = w.spawn(); w.insert_visibility(e, v); return e; };
    Entity id1 = spawn(Visibility::Inherited), id2 = spawn(Visibility::Inherited), id3 = spawn(Visibility::Hidden), id4 = spawn(Visibility::Inherited);
    w.add_child(id1, id2);
    w.add_child(id2, id3);
    w.add_child(id3, id4);
    auto step = [&](std::function<void()> edit, bool c1, bool c2, bool c3, bool c4, const char* what) {
        w.clear_trackers();
        if (edit) edit();
        visibility_propagate(plugin, w);
        CHECK(w.inherited_visibility_changed(id1) == c1 && w.inherited_visibility_changed(id2) == c2 &&
                  w.inherited_visibility_changed(id3) == c3 && w.inherited_visibility_changed(id4) == c4, what);
    };
    visibility_propagate(plugin, w);
    step(nullptr, false, false, false, false, "nothing changed");
    step([&] { w.insert_visibility(id1, Visibility::Hidden); }, true, true, false, false, "id1 hidden");
    step(nullptr, false, false, false, false, "stable");
    step([&] { w.insert_visibility(id3, Visibility::Inherited); }, false, false, false, false, "id3 inherits a hidden parent");
    step([&] { w.insert_visibility(id2, Visibility::Visible); }, false, true, true, true, "id2 visible");
    step(nullptr, false, false, false, false, "stable again");
}

static View camera_looking_down_neg_z() {
    View v;
    float cfv[16], cam[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
    mi_perspective_clip_from_view(1.0f, 1.0f, 0.1f, cfv);
    mi_compute_frustum(cfv, cam, 1000.0f, v.frustum);
    return v;
}

// visibility/mod.rs:1313-1448 (the "manual mark" is a camera the entity is inside / outside of)
static void view_visibility_lifecycle() {
    World w;
    Mi355xPlugin plugin;
    Entity e = w.spawn(Transform::from_xyz(0, 0, 50));  // behind the camera: hidden
    w.insert_aabb(e, Aabb{{0, 0, 0}, {0.5f, 0.5f, 0.5f}});
    const std::vector<View> views = {camera_looking_down_neg_z()};
    bool first = true;
    auto update = [&](std::function<void()> edit) {  // App::update(): trackers are cleared at the END of a frame
        if (!first) w.clear_trackers();
        first = false;
        if (edit) edit();
        post_update(plugin, w, views);
    };
    update(nullptr);
    update(nullptr);
    CHECK(!w.view_visibility(e), "Frame 1: should be hidden");
    CHECK(!w.view_visibility_changed(e), "Frame 1: should not be changed");
    update([&] { w.transform_mut(e).translation.z = -50.0f; });
    CHECK(w.view_visibility(e), "Frame 2: should be visible");
    CHECK(w.view_visibility_changed(e), "Frame 2: should be changed");
    update(nullptr);
    CHECK(w.view_visibility(e), "Frame 3: should be visible");
    CHECK(!w.view_visibility_changed(e), "Frame 3: should NOT be changed");
    update([&] { w.transform_mut(e).translation.z = 50.0f; });
    CHECK(!w.view_visibility(e), "Frame 4: should be hidden");
    CHECK(w.view_visibility_changed(e), "Frame 4: should be changed");
    update(nullptr);
    CHECK(!w.view_visibility(e), "Frame 5: should be hidden");
    CHECK(!w.view_visibility_changed(e), "Frame 5: should NOT be changed");
}

// benches/benches/bevy_camera/primitives.rs:41-52 + VisibleEntities ordering (visibility/mod.rs:861-874)
static void visible_entities_are_sorted_by_entity() {
    World w;
    Mi355xPlugin plugin;
    std::vector<Entity> inside;
    for (int i = 0; i < 40; ++i) {
        const bool in = (i % 3) != 0;
        Entity e = w.spawn(Transform::from_xyz((float)(i % 5) - 2.0f, 0.0f, in ? -20.0f : 20.0f));
        w.insert_aabb(e, Aabb{{0, 0, 0}, {0.5f, 0.5f, 0.5f}});
        if (in) inside.push_back(e);
    }
    std::vector<Entity> got = post_update(plugin, w, {camera_looking_down_neg_z()})[0];
    std::sort(inside.begin(), inside.end(), [](Entity a, Entity b) { return a.to_bits() < b.to_bits(); });
    CHECK(got == inside, "VisibleEntities = the entities in view, ascending by Entity::to_bits");
    for (Entity e : w.entities()) CHECK(w.view_visibility(e) == (std::find(inside.begin(), inside.end(), e) != inside.end()), "ViewVisibility");
}

// crates/bevy_light/src/cluster/test.rs (tiling) + the structural invariants of assign_objects_to_clusters
// (assign.rs:487-804): the reference has no known-answer test for the assignment itself.
static void lights_are_assigned_to_clusters() {
    World w;
    Mi355xPlugin plugin;
    Entity in_front = w.spawn(Transform::from_xyz(0.0f, 0.0f, -20.0f));
    Entity behind = w.spawn(Transform::from_xyz(0.0f, 0.0f, 30.0f));
    Entity off_right = w.spawn(Transform::from_xyz(6.0f, 0.0f, -20.0f));
    Entity huge = w.spawn(Transform::from_xyz(0.0f, 0.0f, -50.0f));
    w.insert_point_light(in_front, 1.0f);
    w.insert_point_light(behind, 1.0f);
    w.insert_point_light(off_right, 1.0f);
    w.insert_point_light(huge, 1.0e6f);
    w.spawn(Transform::from_xyz(1, 2, 3));  // not a light
    ClusterCamera cam;
    mi_perspective_clip_from_view(3.14159265f / 4.0f, 16.0f / 9.0f, 0.1f, cam.clip_from_view);
    mi_compute_frustum(cam.clip_from_view, cam.camera_affine, 1000.0f, cam.frustum);
    Clusters cl;
    if (g_fused) {  // the camera's frame: propagate + cull + gather of the visible lights + assignment, one round trip
        View view;
        std::memcpy(view.frustum, cam.frustum, sizeof view.frustum);
        Mi355xPlugin::FrameOutput out = plugin.frame(w, {view}, &cam);
        CHECK(out.has_clusters, "the fused frame carries the cluster stage");
        cl = out.clusters;
        // the lights' own ViewVisibility comes from their bounding Sphere (point_light.rs:195-208): the one behind the camera is hidden
        CHECK(w.view_visibility(in_front) && !w.view_visibility(behind) && w.view_visibility(off_right) && w.view_visibility(huge), "ViewVisibility of the lights");
    } else {
        View view;
        std::memcpy(view.frustum, cam.frustum, sizeof view.frustum);
        plugin.propagate_transforms(w);
        plugin.check_visibility(w, {view});  // the gather takes the lights whose ViewVisibility::get() is true (assign.rs:194)
        cl = plugin.assign_objects_to_clusters(w, cam);
    }
    CHECK(cl.dimensions[0] == 16 && cl.dimensions[1] == 9 && cl.dimensions[2] == 24, "1920x1080 with 16x9x24 requested");
    CHECK(cl.dimensions[0] * cl.dimensions[1] * cl.dimensions[2] <= 4096, "at most 4096 clusters (test.rs)");
    size_t n_front = 0, n_behind = 0, n_right = 0, n_huge = 0, total = 0;
    bool ordered = true, counts_ok = true;
    for (const ObjectsInCluster& c : cl.clusterable_objects) {
        total += c.entities.size();
        counts_ok = counts_ok && c.counts[0] == c.entities.size();
        for (size_t i = 0; i < c.entities.size(); ++i) {
            const Entity e = c.entities[i];
            n_front += e == in_front; n_behind += e == behind; n_right += e == off_right; n_huge += e == huge;
            if (i && !(c.entities[i - 1].index < e.index)) ordered = false;  // push order = gather (query) order
        }
    }
    CHECK(n_front >= 1 && n_front <= 24, "a small light straight ahead touches a handful of clusters");
    CHECK(n_behind == 0, "a light behind the camera is culled by the frustum test (assign.rs:496)");
    CHECK(n_right >= 1, "a light off to the right but in view is assigned");
    CHECK(n_huge == cl.clusterable_objects.size(), "a light that swallows the frustum is in every cluster");
    CHECK(ordered, "per-cluster lists keep the gather order");
    CHECK(counts_ok && total == cl.total_index_count, "counts and total agree with the lists");
    // the small light's clusters are central in x/y: cluster index (y * dims.x + x) * dims.z + z
    for (size_t c = 0; c < cl.clusterable_objects.size(); ++c)
        for (Entity e : cl.clusterable_objects[c].entities)
            if (e == in_front) {
                const uint32_t xy = (uint32_t)(c / cl.dimensions[2]), x = xy % cl.dimensions[0], y = xy / cl.dimensions[0];
                CHECK(x >= 6 && x <= 9 && y >= 3 && y <= 5, "cluster of the centred light is central");
            }
}

// assign.rs:190-248, 563-573, 681-738: the gather takes point lights, then spot lights, then -- only with storage buffers -- rect
// lights; a spot light is assigned to the clusters its cone reaches; every cluster's list is in gather order and its per-type counts
// add up.  Runs in both forms of the boundary (driven by main); `compare` = the other form's result of the same World.
static Clusters all_kinds_frame(bool fused, bool storage_buffers, std::vector<Entity>* spots_out, std::vector<Entity>* rects_out) {
    World w;
    Mi355xPlugin plugin;
    w.set_supports_storage_buffers(storage_buffers);
    std::vector<Entity> points, spots, rects;
    uint64_t rng = 0x1234567ull;
    auto next = [&rng]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (float)(rng % 20001) / 10000.0f - 1.0f; };
    for (int i = 0; i < 90; ++i) {  // interleaved spawn order: the gather order is by kind, then by Entity
        Transform t = Transform::from_xyz(14.0f * next(), 8.0f * next(), -6.0f - 40.0f * (0.5f + 0.5f * next()));
        const float a = 1.5f * next(), b = 1.0f * next();
        t.rotation = {std::sin(a) * std::cos(b), std::sin(b), 0.1f, std::cos(a)};
        const float ln = std::sqrt(t.rotation.x * t.rotation.x + t.rotation.y * t.rotation.y + t.rotation.z * t.rotation.z + t.rotation.w * t.rotation.w);
        t.rotation = {t.rotation.x / ln, t.rotation.y / ln, t.rotation.z / ln, t.rotation.w / ln};
        Entity e = w.spawn(t);
        if (i % 3 == 0) { w.insert_point_light(e, 2.0f + 2.0f * (0.5f + 0.5f * next())); points.push_back(e); }
        else if (i % 3 == 1) { w.insert_spot_light(e, 6.0f + 6.0f * (0.5f + 0.5f * next()), 0.2f + 0.5f * (0.5f + 0.5f * next())); spots.push_back(e); }
        else { w.insert_rect_light(e, 3.0f + 2.0f * (0.5f + 0.5f * next())); rects.push_back(e); }
    }
    Entity behind = w.spawn(Transform::from_xyz(0.0f, 0.0f, 40.0f));
    w.insert_spot_light(behind, 3.0f, 0.4f);  // behind the camera: hidden by its bounding Sphere (spot_light.rs:221-234), never gathered
    ClusterCamera cam;
    mi_perspective_clip_from_view(3.14159265f / 4.0f, 16.0f / 9.0f, 0.1f, cam.clip_from_view);
    mi_compute_frustum(cam.clip_from_view, cam.camera_affine, 1000.0f, cam.frustum);
    View view;
    std::memcpy(view.frustum, cam.frustum, sizeof view.frustum);
    Clusters cl;
    if (fused) {
        Mi355xPlugin::FrameOutput out = plugin.frame(w, {view}, &cam);
        CHECK(out.has_clusters, "the fused frame carries the cluster stage with every kind of light");
        cl = out.clusters;
    } else {
        plugin.propagate_transforms(w);
        plugin.check_visibility(w, {view});
        cl = plugin.assign_objects_to_clusters(w, cam);
    }
    CHECK(!w.view_visibility(behind), "a spot light behind the camera is not visible");
    if (spots_out) *spots_out = spots;
    if (rects_out) *rects_out = rects;
    return cl;
}
static void every_kind_of_light_is_clustered() {
    for (int storage = 1; storage >= 0; --storage) {
        std::vector<Entity> spots, rects;
        const Clusters cl = all_kinds_frame(g_fused, storage != 0, &spots, &rects);
        const Clusters other = all_kinds_frame(!g_fused, storage != 0, nullptr, nullptr);  // the same World through the other form
        auto is_in = [](const std::vector<Entity>& v, Entity e) { return std::find(v.begin(), v.end(), e) != v.end(); };
        uint64_t n_point = 0, n_spot = 0, n_rect = 0, total = 0;
        bool grouped = true, counts_ok = true, same = cl.clusterable_objects.size() == other.clusterable_objects.size();
        for (size_t c = 0; c < cl.clusterable_objects.size(); ++c) {
            const ObjectsInCluster& o = cl.clusterable_objects[c];
            total += o.entities.size();
            uint32_t k[3] = {0, 0, 0};
            int last_kind = 0;
            for (Entity e : o.entities) {
                const int kind = is_in(spots, e) ? 1 : is_in(rects, e) ? 2 : 0;
                grouped = grouped && kind >= last_kind;  // points, then spots, then rects: the gather order
                last_kind = kind;
                ++k[kind];
            }
            counts_ok = counts_ok && o.counts[0] == k[0] && o.counts[1] == k[1] && o.counts[2] == k[2] && o.counts[3] + o.counts[4] + o.counts[5] == 0;
            n_point += k[0], n_spot += k[1], n_rect += k[2];
            if (same) same = o.entities.size() == other.clusterable_objects[c].entities.size() &&
                             std::equal(o.entities.begin(), o.entities.end(), other.clusterable_objects[c].entities.begin()) &&
                             std::memcmp(o.counts, other.clusterable_objects[c].counts, sizeof o.counts) == 0;
        }
        CHECK(grouped, "every cluster lists point lights, then spot lights, then rect lights");
        CHECK(counts_ok, "ClusterableObjectCounts per type agree with the lists");
        CHECK(n_point > 0 && n_spot > 0, "point and spot lights reach clusters");
        CHECK(storage ? n_rect > 0 : n_rect == 0, "rect lights are gathered only with storage buffers (assign.rs:231-248)");
        CHECK(total == cl.total_index_count, "total_cluster_index_count");
        CHECK(same && cl.farthest_z == other.farthest_z && cl.total_index_count == other.total_index_count, "both forms of the boundary leave the same Clusters");
    }
}

// Light probes and clustered decals (assign.rs:250-296): range = transform.radius_vec3a(Vec3A::ONE) / transform.scale().length()
// of the GlobalTransform of THIS frame, RenderLayers::default(), probes behind supports_storage_buffers and decals behind
// clustered_decals_are_usable.  In the fused frame an unparented one rides like a light (range from From(Transform), formed before the
// frame runs); a parented one leaves the frame's clusters to assign_objects_to_clusters.  Both forms must leave the same Clusters --
// also after a probe was scaled -- and the per-type counts must land in their own slots.
static Clusters probes_and_decals_frame(bool fused, bool decals_usable, bool parented, bool* rode) {
    World w;
    Mi355xPlugin plugin;
    w.set_clustered_decals_are_usable(decals_usable);
    uint64_t rng = 0xABCDEFull;
    auto next = [&rng]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (float)(rng % 20001) / 10000.0f - 1.0f; };
    // (scaled and turned: a child's range is not the range of its own Transform)
    Transform parent_t = Transform::from_xyz(0.5f, -0.25f, -2.0f);
    parent_t.scale = {1.5f, 0.75f, 2.0f};
    parent_t.rotation = {0.0f, 0.14943813f, 0.0f, 0.98877108f};  // 0.3 rad about y
    Entity parent = w.spawn(parent_t);
    std::vector<Entity> probes;
    for (int i = 0; i < 40; ++i) {
        Transform t = Transform::from_xyz(12.0f * next(), 7.0f * next(), -8.0f - 30.0f * (0.5f + 0.5f * next()));
        t.scale = {1.0f + 2.0f * (0.5f + 0.5f * next()), 0.5f + (0.5f + 0.5f * next()), 1.0f + (0.5f + 0.5f * next())};
        Entity e = w.spawn(t);
        if (i % 4 == 0) w.insert_point_light(e, 3.0f);
        else if (i % 4 == 1) { w.insert_light_probe(e, true); probes.push_back(e); }
        else if (i % 4 == 2) w.insert_light_probe(e, false);
        else w.insert_clustered_decal(e);
        if (parented && (i == 5 || i == 7)) w.add_child(parent, e);  // a reflection probe and a decal
    }
    ClusterCamera cam;
    mi_perspective_clip_from_view(3.14159265f / 4.0f, 16.0f / 9.0f, 0.1f, cam.clip_from_view);
    mi_compute_frustum(cam.clip_from_view, cam.camera_affine, 1000.0f, cam.frustum);
    View view;
    std::memcpy(view.frustum, cam.frustum, sizeof view.frustum);
    Clusters cl;
    for (int frame = 0; frame < 2; ++frame) {
        if (frame == 1) w.transform_mut(probes[1]).scale = {4.0f, 3.0f, 5.0f};  // a bigger probe: its range follows in the same frame
        if (frame == 1 && parented) w.transform_mut(parent).scale = {2.5f, 1.25f, 0.5f};  // ... and so does a child's when an ancestor changes
        if (fused) {
            Mi355xPlugin::FrameOutput out = plugin.frame(w, {view}, &cam);
            if (rode) *rode = out.has_clusters;
            cl = out.has_clusters ? out.clusters : plugin.assign_objects_to_clusters(w, cam);
        } else {
            plugin.propagate_transforms(w);
            plugin.check_visibility(w, {view});
            cl = plugin.assign_objects_to_clusters(w, cam);
        }
    }
    return cl;
}
static void light_probes_and_decals_are_clustered() {
    for (int parented = 0; parented < 2; ++parented)
        for (int usable = 1; usable >= 0; --usable) {
            bool rode = false;
            const Clusters cl = probes_and_decals_frame(g_fused, usable != 0, parented != 0, &rode);
            const Clusters other = probes_and_decals_frame(!g_fused, usable != 0, parented != 0, nullptr);
            if (g_fused) CHECK(rode, "probes and decals ride in the fused frame, with or without a parent");
            uint64_t k[6] = {0, 0, 0, 0, 0, 0}, total = 0;
            bool same = cl.clusterable_objects.size() == other.clusterable_objects.size();
            for (size_t c = 0; c < cl.clusterable_objects.size(); ++c) {
                const ObjectsInCluster& o = cl.clusterable_objects[c];
                total += o.entities.size();
                uint64_t in_counts = 0;
                for (int t = 0; t < 6; ++t) { k[t] += o.counts[t]; in_counts += o.counts[t]; }
                CHECK(in_counts == o.entities.size(), "the per-type counts of a cluster add up to its list");
                if (same) same = o.entities.size() == other.clusterable_objects[c].entities.size() &&
                                 std::equal(o.entities.begin(), o.entities.end(), other.clusterable_objects[c].entities.begin()) &&
                                 std::memcmp(o.counts, other.clusterable_objects[c].counts, sizeof o.counts) == 0;
            }
            CHECK(k[0] > 0 && k[3] > 0 && k[4] > 0, "point lights, reflection probes and irradiance volumes reach clusters");
            CHECK(usable ? k[5] > 0 : k[5] == 0, "decals are gathered only where clustered decals are usable (assign.rs:279-296)");
            CHECK(k[1] == 0 && k[2] == 0, "no spot or rect lights in this World");
            CHECK(total == cl.total_index_count, "total_cluster_index_count");
            CHECK(same && cl.farthest_z == other.farthest_z, "both forms of the boundary leave the same Clusters");
        }
}

// crates/bevy_render/src/render_phase/mod.rs:2356-2700 (proptest render_multidrawable_batch_set): random Add / Remove
// of mock mesh instances (entity 0..32, bin 0..8, distinct input uniform indices), then the invariants -- a bin's
// instance_count is the number of entities in it, every binned instance appears exactly once with its input uniform
// index -- here for the instances the camera sees, plus what prepare_multidrawable_binned_batch_set promises
// (gpu_preprocessing.rs:2511-2579): contiguous work-item, MeshUniform and indirect-parameter ranges per batch set.
static void render_multidrawable_batch_set() {
    World w;
    Mi355xPlugin plugin;
    std::vector<Entity> ents;
    for (int i = 0; i < 32; ++i) {
        // two out of three in front of the camera, the rest behind it
        Entity e = w.spawn(Transform::from_xyz((float)(i % 7) - 3.0f, (float)(i % 3) - 1.0f, (i % 3) ? -25.0f : 25.0f));
        w.insert_aabb(e, Aabb{{0, 0, 0}, {0.5f, 0.5f, 0.5f}});
        ents.push_back(e);
    }
    propagate(plugin, w);
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    auto next = [&rng]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
    std::map<uint32_t, MeshBinning> binned;  // entity index -> control copy
    uint32_t next_input = 0;
    const mi_batch_initial initial = {{2, 5}, {1, 3}, {1, 2}, 7};
    for (int round = 0; round < 6; ++round) {
        for (int op = 0; op < 40; ++op) {
            const uint32_t id = (uint32_t)(next() % 32);
            if (next() % 3) {  // Add (skipped when already binned, like the proptest)
                if (binned.count(id)) continue;
                MeshBinning b;
                b.batch_set_key = 100 + next() % 3;
                b.indexed = (b.batch_set_key & 1) != 0;
                b.bin_key = next() % 8;
                b.input_uniform_index = next_input++;
                binned[id] = b;
                w.insert_mesh_binning(ents[id], b);
            } else if (binned.count(id)) {  // Remove
                binned.erase(id);
                w.remove_mesh_binning(ents[id]);
            }
        }
        const std::vector<Entity> visible = post_update(plugin, w, {camera_looking_down_neg_z()})[0];
        const PhaseBatches pb = plugin.batch_multidrawables(w, 0, &initial);
        // control: visible binned instances per (set, bin)
        std::map<std::pair<uint64_t, uint64_t>, std::vector<uint32_t>> expect;  // -> input uniform indices, VisibleEntities order
        std::map<uint64_t, uint32_t> per_set;
        for (Entity e : visible) {
            auto it = binned.find(e.index);
            if (it == binned.end()) continue;
            expect[{it->second.batch_set_key, it->second.bin_key}].push_back(it->second.input_uniform_index);
            per_set[it->second.batch_set_key] += 1;
        }
        for (const BatchBin& b : pb.bins) {
            auto it = expect.find({b.batch_set_key, b.bin_key});
            CHECK(b.metadata.instance_count == (it == expect.end() ? 0u : (uint32_t)it->second.size()), "instance_count == visible entities in the bin");
        }
        CHECK(pb.records.size() == per_set.size(), "one record per batch set with a visible instance");
        uint32_t wi_cursor[2] = {initial.work_item_index[0], initial.work_item_index[1]};
        uint32_t ip_cursor[2] = {initial.indirect_parameters_index[0], initial.indirect_parameters_index[1]};
        uint32_t bs_cursor[2] = {initial.batch_set_index[0], initial.batch_set_index[1]};
        uint32_t out_cursor = initial.output_mesh_uniform_index;
        uint64_t prev_key = 0;
        for (const BatchSetRecord& r : pb.records) {
            const uint32_t c = r.indexed ? 1u : 0u;
            CHECK(r.batch_set_key > prev_key, "batch sets in key order");
            prev_key = r.batch_set_key;
            CHECK(r.instance_count == per_set[r.batch_set_key], "record.instance_count");
            CHECK(r.first_work_item_index == wi_cursor[c] && r.first_indirect_parameters_index == ip_cursor[c] && r.index == bs_cursor[c] &&
                      r.first_output_mesh_uniform_index == out_cursor,
                  "ranges are allocated back to back (prepare_multidrawable_binned_batch_set)");
            CHECK(pb.batch_sets[c][r.index].indirect_parameters_base == r.first_indirect_parameters_index &&
                      pb.batch_sets[c][r.index].indirect_parameters_count == 0,
                  "IndirectBatchSet");
            // work items of the set: every visible instance once, pointing at its bin's indirect parameters
            std::map<uint32_t, std::vector<uint32_t>> by_slot;
            for (uint32_t k = 0; k < r.instance_count; ++k) {
                const mi_preprocess_work_item& wi = pb.work_items[c][r.first_work_item_index + k];
                by_slot[wi.output_or_indirect_parameters_index].push_back(wi.input_index);
            }
            uint32_t expect_base = r.first_output_mesh_uniform_index, bins_of_set = 0;
            for (const BatchBin& b : pb.bins) {
                if (b.batch_set_key != r.batch_set_key) continue;
                ++bins_of_set;
                const uint32_t slot = r.first_indirect_parameters_index + b.metadata.indirect_parameters_offset;
                const auto it = expect.find({b.batch_set_key, b.bin_key});
                const std::vector<uint32_t> none;
                CHECK(by_slot[slot] == (it == expect.end() ? none : it->second), "a bin's work items = its visible instances, in VisibleEntities order");
                const mi_indirect_parameters_metadata& md = pb.metadata[c][slot];
                CHECK(md.base_output_index == expect_base && md.batch_set_index == r.index && md.mesh_index == 0 && md.early_instance_count == 0 &&
                          md.late_instance_count == 0,
                      "allocate_uniforms: MeshUniform ranges of the bins tile the set's range");
                expect_base += b.metadata.instance_count;
            }
            CHECK(bins_of_set == r.batch_count, "batch_count = bins of the set");
            wi_cursor[c] += r.instance_count;
            ip_cursor[c] += r.batch_count;
            bs_cursor[c] += 1;
            out_cursor += r.instance_count;
        }
        CHECK(pb.totals.data_buffer_len == out_cursor && pb.totals.work_item_len[0] == wi_cursor[0] && pb.totals.work_item_len[1] == wi_cursor[1], "totals");
        w.clear_trackers();
    }
    CHECK(!binned.empty(), "the random walk left instances binned");
}

// Twin worlds, one edit script; one runs the three systems, the other the fused frame: after every frame the two Worlds are the
// same component for component -- GlobalTransform bits and change ticks, InheritedVisibility, ViewVisibility and its ticks,
// VisibleEntities, Clusters.  (The fused frame makes its ECS writes where the stock systems make theirs; this is the check.)
static void both_forms_leave_the_same_world() {
    World wa, wb;
    Mi355xPlugin pa, pb;
    uint64_t rng = 0x2545F4914F6CDD1Dull;
    auto next = [&rng]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
    auto frand = [&](float lo, float hi) { return lo + (hi - lo) * (float)(next() % 10000) / 10000.0f; };
    std::vector<Entity> ents;  // the same handles are valid in both worlds (same spawn / despawn sequence)
    auto spawn = [&](std::optional<Entity> parent, bool light) {
        const Transform t = Transform::from_xyz(frand(-12, 12), frand(-6, 6), frand(-60, 10));
        Entity ea = wa.spawn(t), eb = wb.spawn(t);
        CHECK(ea == eb, "twin worlds hand out the same entity");
        if (parent) { wa.add_child(*parent, ea); wb.add_child(*parent, eb); }
        if (light) { const float range = frand(0.5f, 6.0f); wa.insert_point_light(ea, range); wb.insert_point_light(eb, range); }
        else { const Aabb a{{0, 0, 0}, {frand(0.2f, 1.5f), frand(0.2f, 1.5f), frand(0.2f, 1.5f)}}; wa.insert_aabb(ea, a); wb.insert_aabb(eb, a); }
        if (next() % 3 == 0) {
            const Visibility v = (Visibility)(next() % 3);
            wa.insert_visibility(ea, v); wb.insert_visibility(eb, v);
        }
        ents.push_back(ea);
        return ea;
    };
    for (int i = 0; i < 150; ++i) {
        std::optional<Entity> parent;
        if (i > 10 && next() % 2) parent = ents[next() % ents.size()];
        spawn(parent, i % 9 == 0);
    }
    ClusterCamera cam;
    mi_perspective_clip_from_view(3.14159265f / 4.0f, 16.0f / 9.0f, 0.1f, cam.clip_from_view);
    mi_compute_frustum(cam.clip_from_view, cam.camera_affine, 1000.0f, cam.frustum);
    View view;
    std::memcpy(view.frustum, cam.frustum, sizeof view.frustum);
    View side = camera_looking_down_neg_z();
    const std::vector<View> views = {view, side};
    for (int frame = 0; frame < 14; ++frame) {
        if (frame) { wa.clear_trackers(); wb.clear_trackers(); }
        const bool steady = frame > 0 && frame % 3 != 1 && frame != 4 && frame != 9 && frame != 6 && frame != 7;  // transforms only
        // edits
        const int n_moves = frame % 4 == 3 ? 0 : 1 + (int)(next() % 25);
        for (int k = 0; k < n_moves; ++k) {
            const Entity e = ents[next() % ents.size()];
            if (!wa.contains(e)) continue;
            const float dz = frand(-3, 3), dx = frand(-1, 1);
            wa.transform_mut(e).translation.z += dz; wb.transform_mut(e).translation.z += dz;
            wa.transform_mut(e).translation.x += dx; wb.transform_mut(e).translation.x += dx;
        }
        if (frame % 3 == 1) {
            const Entity e = ents[next() % ents.size()];
            if (wa.contains(e)) { const Visibility v = (Visibility)(next() % 3); wa.insert_visibility(e, v); wb.insert_visibility(e, v); }
        }
        if (frame == 4 || frame == 9) spawn(ents[next() % 20], frame == 9);
        if (frame == 6) {
            const Entity e = ents[40 + next() % 50];
            if (wa.contains(e) && wa.children(e).empty()) { wa.despawn(e); wb.despawn(e); }
        }
        if (frame == 7) {
            const Entity e = ents[30 + next() % 30];
            if (wa.contains(e) && wa.parent(e)) { wa.remove_parent(e); wb.remove_parent(e); }
        }
        wa.static_transform_optimizations = wb.static_transform_optimizations = frame >= 10;
        // A: the three systems
        pa.propagate_transforms(wa);
        pa.visibility_propagate(wa);
        pa.check_visibility(wa, views);
        const Clusters ca = pa.assign_objects_to_clusters(wa, cam);
        // B: the fused frame
        const Mi355xPlugin::FrameOutput fb = pb.frame(wb, views, &cam);
        if (steady) CHECK(fb.device_waits == 1, "a frame without structural edits or Visibility writes is one device wait");
        bool same = true, same_ticks = true;
        for (Entity e : wa.entities()) {
            same = same && wb.contains(e) && wa.global_transform(e) == wb.global_transform(e) && wa.inherited_visibility(e) == wb.inherited_visibility(e) &&
                   wa.view_visibility(e) == wb.view_visibility(e);
            same_ticks = same_ticks && wa.global_transform_changed(e) == wb.global_transform_changed(e) &&
                         wa.view_visibility_changed(e) == wb.view_visibility_changed(e) &&
                         wa.inherited_visibility_changed(e) == wb.inherited_visibility_changed(e);
        }
        CHECK(same, "component values");
        CHECK(same_ticks, "change ticks");
        for (uint32_t v = 0; v < views.size(); ++v) CHECK(pa.visible_entities(v) == fb.visible_entities[v], "VisibleEntities");
        CHECK(fb.has_clusters && ca.clusterable_objects.size() == fb.clusters.clusterable_objects.size(), "Clusters: grid");
        bool lists = fb.has_clusters && ca.total_index_count == fb.clusters.total_index_count && ca.farthest_z == fb.clusters.farthest_z;
        for (size_t c = 0; lists && c < ca.clusterable_objects.size(); ++c)
            lists = ca.clusterable_objects[c].entities == fb.clusters.clusterable_objects[c].entities &&
                    std::memcmp(ca.clusterable_objects[c].counts, fb.clusters.clusterable_objects[c].counts, sizeof ca.clusterable_objects[c].counts) == 0;
        CHECK(lists, "Clusters: lists, counts, farthest_z");
        if (frame == 2) CHECK(ca.total_index_count > 0, "some light is in view");
    }
}

// The plugin over several GPUs (bevy_mi355x_sharded.hpp: a context per device, rows in contiguous ranges, the masks all-gathered by RCCL
// between ncclGroupStart / End and read back with ONE copy) against the single-device plugin on a twin World: after every frame the
// same GlobalTransforms and change ticks, ViewVisibility and its ticks, VisibleEntities.  Device lists: {0} -- a 1-rank communicator,
// every code path of the exchange but the wire --, {0, 0, 0} -- three shards on one GPU, no communicator (the masks are read per
// context): the sharding itself, shard boundaries inside and at the end of the row space --, and, where the node has them (MI_TEST_DEVICES
// = their number), every GPU.
static void sharded_plugin_leaves_the_same_world() {
    std::vector<std::vector<int>> lists = {{0}, {0, 0, 0}};
    if (const char* nd = std::getenv("MI_TEST_DEVICES")) {
        std::vector<int> all;
        for (int d = 0; d < std::atoi(nd); ++d) all.push_back(d);
        if (all.size() > 1) lists.push_back(all);
    }
    for (const std::vector<int>& devices : lists) {
        World wa, wb;
        Mi355xPlugin pa;
        Mi355xShardedPlugin pb(devices);
        if (devices.size() == 1 || (devices.size() > 1 && devices[0] != devices[1])) CHECK(pb.exchanged(), pb.exchange_note().c_str());
        else CHECK(!pb.exchanged(), "a device named twice cannot form a communicator");
        uint64_t rng = 0x853C49E6748FEA9Bull + devices.size();
        auto next = [&rng]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
        auto frand = [&](float lo, float hi) { return lo + (hi - lo) * (float)(next() % 10000) / 10000.0f; };
        std::vector<Entity> ents;
        const int n = devices.size() == 3 ? 1100 : 3000;  // (1100 rows over three shards of 512: the last one is short)
        for (int i = 0; i < n; ++i) {
            Transform t = Transform::from_xyz(frand(-40, 40), frand(-25, 25), frand(-90, 30));
            const float a = frand(-1.5f, 1.5f);
            t.rotation = {std::sin(a), 0.0f, 0.0f, std::cos(a)};
            t.scale = {frand(0.5f, 2.0f), frand(0.5f, 2.0f), frand(0.5f, 2.0f)};
            Entity ea = wa.spawn(t), eb = wb.spawn(t);
            CHECK(ea == eb, "twin worlds hand out the same entity");
            if (next() % 8) { const Aabb bb{{frand(-1, 1), 0, 0}, {frand(0.2f, 2), frand(0.2f, 2), frand(0.2f, 2)}}; wa.insert_aabb(ea, bb); wb.insert_aabb(eb, bb); }
            if (next() % 9 == 0) { wa.insert_visibility(ea, Visibility::Hidden); wb.insert_visibility(eb, Visibility::Hidden); }
            if (next() % 5 == 0) { const VisibilityRange vr = VisibilityRange::abrupt(frand(0, 40), frand(40, 120)); wa.insert_visibility_range(ea, vr); wb.insert_visibility_range(eb, vr); }
            if (next() % 6 == 0) { wa.insert_render_layers(ea, 2u); wb.insert_render_layers(eb, 2u); }
            ents.push_back(ea);
        }
        wa.set_visible_entity_ranges(true);
        wb.set_visible_entity_ranges(true);
        float cam1[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 5, 2, 20};
        View v0 = camera_looking_down_neg_z(), v1;
        float cfv[16];
        mi_perspective_clip_from_view(1.1f, 1.5f, 0.1f, cfv);
        mi_compute_frustum(cfv, cam1, 400.0f, v1.frustum);
        v1.layer_mask = 3u;
        v1.position = {5, 2, 20};
        std::vector<View> views = {v0, v1};
        for (int frame = 0; frame < 7; ++frame) {
            if (frame) { wa.clear_trackers(); wb.clear_trackers(); }
            const int n_moves = frame == 3 ? 0 : frame == 5 ? n : 1 + (int)(next() % 200);  // a quiet frame, an all-dirty one, sparse ones
            for (int k = 0; k < n_moves; ++k) {
                const Entity e = frame == 5 ? ents[(size_t)k] : ents[next() % ents.size()];
                const float dz = frand(-6, 6), dx = frand(-2, 2);
                wa.transform_mut(e).translation.z += dz; wb.transform_mut(e).translation.z += dz;
                wa.transform_mut(e).translation.x += dx; wb.transform_mut(e).translation.x += dx;
            }
            if (frame == 4) {  // one view fewer: the gathered buffers are laid out again
                views.pop_back();
            }
            if (frame == 6) {  // structure: rows are renumbered and the shards cut again
                const Transform t = Transform::from_xyz(0, 0, -30);
                Entity ea = wa.spawn(t), eb = wb.spawn(t);
                CHECK(ea == eb, "twin spawn");
                ents.push_back(ea);
                wa.despawn(ents[17]); wb.despawn(ents[17]);
            }
            pa.propagate_transforms(wa);
            pa.visibility_propagate(wa);
            pa.check_visibility(wa, views);
            const Mi355xShardedPlugin::FrameOutput fb = pb.frame(wb, views);
            bool same = true, same_ticks = true;
            for (Entity e : wa.entities()) {
                same = same && wb.contains(e) && wa.global_transform(e) == wb.global_transform(e) && wa.view_visibility_bits(e) == wb.view_visibility_bits(e);
                same_ticks = same_ticks && wa.global_transform_changed(e) == wb.global_transform_changed(e) && wa.view_visibility_changed(e) == wb.view_visibility_changed(e);
            }
            CHECK(same, "GlobalTransform and ViewVisibility of every entity");
            CHECK(same_ticks, "their change ticks");
            for (uint32_t v = 0; v < views.size(); ++v) CHECK(pa.visible_entities(v) == fb.visible_entities[v], "VisibleEntities");
            if (frame == 1) CHECK(!fb.visible_entities[0].empty() && fb.visible_entities[0].size() < (size_t)n, "the camera sees some of the scene");
        }
    }
}

// A hierarchy no wider than a wave per level -- transform_hierarchy.rs's `chain` (:29-160), a rope of a few strands -- is one wave's
// chain of dependent level steps on the device (0.32 us a level) and 20 ns a node on a CPU core: mi_hierarchy_advice_for says
// "keep it on the host", and the plugin then leaves the three transform systems to the host (stock_propagate_transforms) and runs
// visibility over the GlobalTransforms it uploads.  Twin Worlds: the default plugin (host transforms) against one that is told to
// keep everything on the device -- the same World after every frame, both forms.
static void narrow_hierarchy_stays_with_the_stock_systems() {
    World wa, wb;
    Mi355xPlugin pa, pb;
    pb.set_keep_narrow_hierarchies_on_host(false);
    uint64_t rng = 0xD1B54A32D192ED03ull;
    auto next = [&rng]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
    auto frand = [&](float lo, float hi) { return lo + (hi - lo) * (float)(next() % 10000) / 10000.0f; };
    std::vector<Entity> ents;
    auto spawn_pair = [&](const Transform& t) {
        Entity ea = wa.spawn(t), eb = wb.spawn(t);
        CHECK(ea == eb, "twin worlds hand out the same entity");
        ents.push_back(ea);
        return ea;
    };
    // a chain of 60 nodes in front of the camera and a rope of three strands hanging off its root, 40 levels deep
    std::vector<Entity> tips = {spawn_pair(Transform::from_xyz(0, 0, -30))};
    const Entity root = tips[0];
    for (int k = 0; k < 3; ++k) {
        Entity c = spawn_pair(Transform::from_xyz(frand(-1, 1), frand(-1, 1), frand(-1, 0)));
        wa.add_child(root, c); wb.add_child(root, c);
        tips.push_back(c);
    }
    for (int level = 0; level < 59; ++level)
        for (size_t k = 0; k < tips.size(); ++k) {
            if (k > 0 && level >= 39) continue;
            Transform t = Transform::from_xyz(frand(-0.3f, 0.3f), frand(-0.3f, 0.3f), frand(-0.4f, 0.1f));
            const float a = frand(-0.2f, 0.2f);
            t.rotation = {0.0f, std::sin(a), 0.0f, std::cos(a)};
            t.scale = {frand(0.95f, 1.05f), frand(0.95f, 1.05f), frand(0.95f, 1.05f)};
            const Entity c = spawn_pair(t);
            wa.add_child(tips[k], c); wb.add_child(tips[k], c);
            tips[k] = c;
        }
    for (Entity e : ents)
        if (next() % 3) { const Aabb bb{{0, 0, 0}, {0.3f, 0.3f, 0.3f}}; wa.insert_aabb(e, bb); wb.insert_aabb(e, bb); }
    const std::vector<View> views = {camera_looking_down_neg_z()};
    for (int frame = 0; frame < 5; ++frame) {
        if (frame) { wa.clear_trackers(); wb.clear_trackers(); }
        wa.static_transform_optimizations = wb.static_transform_optimizations = frame >= 2;
        const int n_moves = frame == 3 ? 0 : 1 + (int)(next() % 12);
        for (int k = 0; k < n_moves; ++k) {
            const Entity e = ents[next() % ents.size()];
            const float dz = frand(-0.5f, 0.5f);
            wa.transform_mut(e).translation.z += dz; wb.transform_mut(e).translation.z += dz;
        }
        const std::vector<std::vector<Entity>> la = post_update(pa, wa, views), lb = post_update(pb, wb, views);
        if (frame == 0) {
            CHECK(pa.transforms_on_host(), "a chain and a rope: the stock transform systems keep the World");
            CHECK(!pb.transforms_on_host(), "told otherwise, the plugin propagates on the device");
        }
        bool same = true, same_ticks = true;
        for (Entity e : wa.entities()) {
            same = same && wa.global_transform(e) == wb.global_transform(e) && wa.view_visibility_bits(e) == wb.view_visibility_bits(e);
            same_ticks = same_ticks && wa.global_transform_changed(e) == wb.global_transform_changed(e) && wa.view_visibility_changed(e) == wb.view_visibility_changed(e);
        }
        CHECK(same, "GlobalTransform and ViewVisibility of every entity: host transforms against the device's");
        CHECK(same_ticks, "their change ticks");
        CHECK(la == lb, "VisibleEntities");
        if (frame == 0) CHECK(!la[0].empty(), "the camera sees the chain");
    }
    // a bushy tree is the device's: the advice is per World
    World wc;
    Mi355xPlugin pc;
    const Entity r2 = wc.spawn(Transform::from_xyz(0, 0, -10));
    for (int k = 0; k < 200; ++k) wc.add_child(r2, wc.spawn(Transform::from_xyz((float)k, 0, 0)));
    post_update(pc, wc, views);
    CHECK(!pc.transforms_on_host(), "a wide tree stays on the device");
}

// SURVEY 8(e) rows 2 and 3 behind the plugin (round 6): a World WITH ChildOf -- a forest of rigs, plus one tree too big for a context's
// fair share, which the placement opens at its root (the root becomes a replicated row) -- and clustered lights, through
// Mi355xShardedPlugin over {0}, {0, 0, 0} (and every GPU where there are several) against the single-device plugin's fused frame on a
// twin World: GlobalTransform, ViewVisibility, InheritedVisibility and their ticks of every entity, VisibleEntities (sorted by
// Entity), and the view's Clusters list for list (entities in gather order, per-type counts, farthest_z, total).
static void sharded_plugin_shards_trees_and_lights() {
    std::vector<std::vector<int>> lists = {{0}, {0, 0, 0}};
    if (const char* nd = std::getenv("MI_TEST_DEVICES")) {
        std::vector<int> all;
        for (int d = 0; d < std::atoi(nd); ++d) all.push_back(d);
        if (all.size() > 1) lists.push_back(all);
    }
    for (const std::vector<int>& devices : lists) {
        World wa, wb;
        Mi355xPlugin pa;
        Mi355xShardedPlugin pb(devices);
        uint64_t rng = 0x9E3779B97F4A7C15ull + devices.size();
        auto next = [&rng]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
        auto frand = [&](float lo, float hi) { return lo + (hi - lo) * (float)(next() % 10000) / 10000.0f; };
        std::vector<Entity> ents, lights;
        auto spawn_pair = [&](const Transform& t) {
            Entity ea = wa.spawn(t), eb = wb.spawn(t);
            CHECK(ea == eb, "twin worlds hand out the same entity");
            ents.push_back(ea);
            return ea;
        };
        auto small_transform = [&]() {
            Transform t = Transform::from_xyz(frand(-2, 2), frand(-2, 2), frand(-2, 2));
            const float a = frand(-0.6f, 0.6f);
            t.rotation = {0.0f, std::sin(a), 0.0f, std::cos(a)};
            t.scale = {frand(0.8f, 1.2f), frand(0.8f, 1.2f), frand(0.8f, 1.2f)};
            return t;
        };
        // 90 rigs: a root somewhere in front of the camera, a spine, limbs (depth up to 7)
        for (int rig = 0; rig < 90; ++rig) {
            Transform rt = Transform::from_xyz(frand(-40, 40), frand(-25, 25), frand(-90, -5));
            const Entity root = spawn_pair(rt);
            std::vector<Entity> frontier = {root};
            const int depth = 2 + (int)(next() % 6);
            for (int dlev = 0; dlev < depth; ++dlev) {
                std::vector<Entity> nxt;
                for (Entity p : frontier) {
                    const int kids = dlev == 0 ? 2 + (int)(next() % 2) : (int)(next() % 3);
                    for (int k = 0; k < kids && nxt.size() < 12; ++k) {
                        const Entity c = spawn_pair(small_transform());
                        wa.add_child(p, c); wb.add_child(p, c);
                        nxt.push_back(c);
                    }
                }
                if (nxt.empty()) break;
                frontier = nxt;
            }
        }
        // one big tree (a 5-ary tree of depth 5: 781 nodes -- more than a third of the World): opened at its root over three shards
        {
            const Entity root = spawn_pair(Transform::from_xyz(0, 0, -40));
            std::vector<Entity> frontier = {root};
            for (int dlev = 0; dlev < 4; ++dlev) {
                std::vector<Entity> nxt;
                for (Entity p : frontier)
                    for (int k = 0; k < 5; ++k) {
                        const Entity c = spawn_pair(small_transform());
                        wa.add_child(p, c); wb.add_child(p, c);
                        nxt.push_back(c);
                    }
                frontier = nxt;
            }
        }
        // bounds, visibility components, and lights riding on some of the nodes (and some free-standing ones)
        for (Entity e : ents) {
            if (next() % 4) { const Aabb bb{{0, 0, 0}, {frand(0.2f, 1.5f), frand(0.2f, 1.5f), frand(0.2f, 1.5f)}}; wa.insert_aabb(e, bb); wb.insert_aabb(e, bb); }
            const uint64_t v = next() % 12;
            if (v == 0) { wa.insert_visibility(e, Visibility::Hidden); wb.insert_visibility(e, Visibility::Hidden); }
            else if (v == 1) { wa.insert_visibility(e, Visibility::Visible); wb.insert_visibility(e, Visibility::Visible); }
            else { wa.insert_visibility(e, Visibility::Inherited); wb.insert_visibility(e, Visibility::Inherited); }
        }
        for (int i = 0; i < 400; ++i) {
            Entity e;
            if (i % 3 == 0) e = spawn_pair(Transform::from_xyz(frand(-30, 30), frand(-15, 15), frand(-80, -3)));
            else e = ents[next() % ents.size()];
            if (wa.has_point_light(e) || wa.has_aabb(e)) continue;
            const float range = frand(0.5f, 6.0f);
            wa.insert_point_light(e, range); wb.insert_point_light(e, range);
            wa.insert_visibility(e, Visibility::Inherited); wb.insert_visibility(e, Visibility::Inherited);
            lights.push_back(e);
        }
        ClusterCamera cam;
        mi_perspective_clip_from_view(3.14159265f / 4.0f, 16.0f / 9.0f, 0.1f, cam.clip_from_view);
        mi_compute_frustum(cam.clip_from_view, cam.camera_affine, 1000.0f, cam.frustum);
        View view;
        std::memcpy(view.frustum, cam.frustum, sizeof view.frustum);
        const std::vector<View> views = {view};
        for (int frame = 0; frame < 6; ++frame) {
            if (frame) { wa.clear_trackers(); wb.clear_trackers(); }
            wa.static_transform_optimizations = wb.static_transform_optimizations = frame >= 3;
            const int n_moves = frame == 2 ? 0 : 1 + (int)(next() % 300);
            for (int k = 0; k < n_moves; ++k) {
                const Entity e = ents[next() % ents.size()];
                const float dz = frand(-3, 3), dx = frand(-1, 1);
                wa.transform_mut(e).translation.z += dz; wb.transform_mut(e).translation.z += dz;
                wa.transform_mut(e).translation.x += dx; wb.transform_mut(e).translation.x += dx;
            }
            if (frame == 4) {  // a Visibility component is written: InheritedVisibility is swept again, on every shard
                const Entity e = ents[7];
                wa.insert_visibility(e, Visibility::Hidden); wb.insert_visibility(e, Visibility::Hidden);
            }
            if (frame == 5) {  // structure: a subtree changes its parent, the partition is rebuilt
                const Entity child = ents[3], new_parent = ents[ents.size() - 5];
                wa.remove_parent(child); wb.remove_parent(child);
                wa.add_child(new_parent, child); wb.add_child(new_parent, child);
            }
            const Mi355xPlugin::FrameOutput fa = pa.frame(wa, views, &cam);
            const Mi355xShardedPlugin::FrameOutput fb = pb.frame(wb, views, &cam);
            if (frame == 0) {
                CHECK(pb.sharded_by_tree(), "a World with ChildOf shards by tree");
                if (devices.size() == 3) CHECK(pb.replicated_rows() >= 1, "the big tree was opened: its root is a replicated row");
                uint32_t held = 0;
                for (uint32_t c : pb.shard_rows()) held += c;
                CHECK(held >= wb.entities().size() && held <= wb.entities().size() + pb.replicated_rows() * (uint32_t)(devices.size() - 1), "every entity is held by a shard, replicated roots by several");
            }
            bool same = true, same_ticks = true, same_inh = true;
            for (Entity e : wa.entities()) {
                same = same && wb.contains(e) && wa.global_transform(e) == wb.global_transform(e) && wa.view_visibility_bits(e) == wb.view_visibility_bits(e);
                same_ticks = same_ticks && wa.global_transform_changed(e) == wb.global_transform_changed(e) && wa.view_visibility_changed(e) == wb.view_visibility_changed(e);
                same_inh = same_inh && wa.inherited_visibility(e) == wb.inherited_visibility(e);
            }
            CHECK(same, "GlobalTransform and ViewVisibility of every entity");
            CHECK(same_ticks, "their change ticks");
            CHECK(same_inh, "InheritedVisibility of every entity");
            CHECK(fa.visible_entities.size() == 1 && fb.visible_entities.size() == 1 && fa.visible_entities[0] == fb.visible_entities[0], "VisibleEntities, sorted by Entity");
            if (frame == 0) CHECK(!fb.visible_entities[0].empty() && fb.visible_entities[0].size() < wb.entities().size(), "the camera sees some of the scene");
            CHECK(fa.has_clusters && fb.has_clusters, "both frames carry the cluster stage");
            bool lists_same = fa.clusters.clusterable_objects.size() == fb.clusters.clusterable_objects.size();
            size_t total = 0;
            for (size_t c = 0; lists_same && c < fa.clusters.clusterable_objects.size(); ++c) {
                const ObjectsInCluster &x = fa.clusters.clusterable_objects[c], &y = fb.clusters.clusterable_objects[c];
                lists_same = x.entities == y.entities && std::memcmp(x.counts, y.counts, sizeof x.counts) == 0;
                total += x.entities.size();
            }
            if (!lists_same) {
                size_t nd = 0, first = (size_t)-1;
                for (size_t c = 0; c < fa.clusters.clusterable_objects.size() && c < fb.clusters.clusterable_objects.size(); ++c)
                    if (!(fa.clusters.clusterable_objects[c].entities == fb.clusters.clusterable_objects[c].entities)) { ++nd; if (first == (size_t)-1) first = c; }
                std::printf("    devices %zu frame %d: totals %llu / %llu, %zu clusters differ, first %zu:", devices.size(), frame, (unsigned long long)fa.clusters.total_index_count,
                            (unsigned long long)fb.clusters.total_index_count, nd, first);
                if (first != (size_t)-1) {
                    for (Entity e : fa.clusters.clusterable_objects[first].entities) std::printf(" %u", e.index);
                    std::printf(" |");
                    for (Entity e : fb.clusters.clusterable_objects[first].entities) std::printf(" %u", e.index);
                    std::printf(" | counts %u %u / %u %u", fa.clusters.clusterable_objects[first].counts[0], fa.clusters.clusterable_objects[first].counts[1],
                                fb.clusters.clusterable_objects[first].counts[0], fb.clusters.clusterable_objects[first].counts[1]);
                }
                std::printf("\n");
            }
            CHECK(lists_same, "every cluster's list (gather order) and per-type counts");
            CHECK(fa.clusters.total_index_count == fb.clusters.total_index_count && total == fb.clusters.total_index_count, "the total index count");
            CHECK(fa.clusters.farthest_z == fb.clusters.farthest_z, "farthest_z");
            if (frame == 0) CHECK(total > 0, "some light is assigned");
        }
    }
}

static Transform on_sphere(uint64_t i, uint64_t n, double radius, uint64_t& seed, bool rotate);  // (below, with the bench)
// The same twin-world check at a size where the fused frame takes its big-table routes: the gather and the write-back in chunks on the
// plugin's threads, an all-dirty table committed as eight dense windows (which the library sends in pieces, fetching the
// GlobalTransforms ahead of the frame from the second such frame on), indexed windows whose rows descend (GlobalTransforms written
// ahead by the scatter launch).  Flat scene, every tenth entity a point light.
static void big_flat_worlds_agree() {
    World wa, wb;
    Mi355xPlugin pa, pb;
    const uint32_t n = 300000;
    uint64_t seed = 7;
    std::vector<Entity> ents;
    ents.reserve(n);
    for (uint32_t i = 0; i < n; ++i) {
        const Transform t = on_sphere(i, n, i % 10 == 9 ? 30.0 : 120.0, seed, i % 10 != 9);
        Entity ea = wa.spawn(t), eb = wb.spawn(t);
        if (i % 10 == 9) { wa.insert_point_light(ea, 0.5f); wb.insert_point_light(eb, 0.5f); }
        else { const Aabb a{{0, 0, 0}, {0.5f, 0.5f, 0.5f}}; wa.insert_aabb(ea, a); wb.insert_aabb(eb, a); }
        ents.push_back(ea);
    }
    ClusterCamera cam;
    mi_perspective_clip_from_view(3.14159265f / 4.0f, 16.0f / 9.0f, 0.1f, cam.clip_from_view);
    mi_compute_frustum(cam.clip_from_view, cam.camera_affine, 1000.0f, cam.frustum);
    View view;
    std::memcpy(view.frustum, cam.frustum, sizeof view.frustum);
    const std::vector<View> views = {view};
    const int moves[8] = {0 /* everything was just spawned */, 1, 1, 10, 10, 0, 1, 10};  // every k-th entity moves (0: none)
    for (int frame = 0; frame < 8; ++frame) {
        if (frame) { wa.clear_trackers(); wb.clear_trackers(); }
        if (moves[frame])
            for (uint32_t i = (uint32_t)frame % (uint32_t)moves[frame]; i < n; i += (uint32_t)moves[frame]) {
                const float d = 0.01f * (float)(frame + 1);
                wa.transform_mut(ents[i]).translation.y += d; wb.transform_mut(ents[i]).translation.y += d;
            }
        pa.propagate_transforms(wa);
        pa.visibility_propagate(wa);
        pa.check_visibility(wa, views);
        const Clusters ca = pa.assign_objects_to_clusters(wa, cam);
        const Mi355xPlugin::FrameOutput fb = pb.frame(wb, views, &cam);
        bool same = true, same_ticks = true;
        for (Entity e : ents) {
            same = same && wa.global_transform(e) == wb.global_transform(e) && wa.view_visibility(e) == wb.view_visibility(e);
            same_ticks = same_ticks && wa.global_transform_changed(e) == wb.global_transform_changed(e) &&
                         wa.view_visibility_changed(e) == wb.view_visibility_changed(e);
        }
        CHECK(same, "big worlds: component values");
        CHECK(same_ticks, "big worlds: change ticks");
        CHECK(pa.visible_entities(0) == fb.visible_entities[0], "big worlds: VisibleEntities");
        bool lists = fb.has_clusters && ca.total_index_count == fb.clusters.total_index_count && ca.farthest_z == fb.clusters.farthest_z &&
                     ca.clusterable_objects.size() == fb.clusters.clusterable_objects.size();
        for (size_t c = 0; lists && c < ca.clusterable_objects.size(); ++c) lists = ca.clusterable_objects[c].entities == fb.clusters.clusterable_objects[c].entities;
        CHECK(lists, "big worlds: Clusters");
        if (frame == 1) CHECK(fb.changed_global_transforms == n, "every GlobalTransform came back");
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// --bench: BASELINE's metric scene (1 M many_cubes entities + 10 k meshes + 100 k point lights, one camera) through the host
// layer, both call sequences, at 1 % / 10 % / 100 % of the Transforms moved per frame.  Prints one JSON object.
// ---------------------------------------------------------------------------------------------------------------------
static uint64_t splitmix64(uint64_t& x) {
    uint64_t z = (x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static Transform on_sphere(uint64_t i, uint64_t n, double radius, uint64_t& seed, bool rotate) {
    const double golden = 3.883222077450933;  // pi * (3 - sqrt(5))
    const double phi = std::acos(1.0 - 2.0 * ((double)i + 0.36) / ((double)n - 1.0 + 0.72)), theta = golden * (double)i;
    Transform t = Transform::from_xyz((float)(radius * std::cos(theta) * std::sin(phi)), (float)(radius * std::sin(theta) * std::sin(phi)),
                                      (float)(radius * std::cos(phi)));
    if (rotate) {
        double q[4], len = 0;
        for (double& c : q) { c = (double)(splitmix64(seed) >> 11) / 9007199254740992.0 * 2.0 - 1.0; len += c * c; }
        len = std::sqrt(len);
        t.rotation = {(float)(q[0] / len), (float)(q[1] / len), (float)(q[2] / len), (float)(q[3] / len)};
    }
    return t;
}
static int bench(uint32_t n_cubes, uint32_t n_meshes, uint32_t n_lights, int frames) {
    std::printf("{\"scene\": {\"cubes\": %u, \"meshes\": %u, \"lights\": %u}", n_cubes, n_meshes, n_lights);
    for (int fused = 0; fused < 2; ++fused) {
        World w;
        Mi355xPlugin plugin;
        uint64_t seed = 42;
        std::vector<Entity> ents;
        ents.reserve((size_t)n_cubes + n_meshes + n_lights);
        for (uint32_t i = 0; i < n_cubes; ++i) { Entity e = w.spawn(on_sphere(i, n_cubes, 500.0, seed, true)); w.insert_aabb(e, Aabb{{0, 0, 0}, {0.5f, 0.5f, 0.5f}}); ents.push_back(e); }
        for (uint32_t i = 0; i < n_meshes; ++i) { Entity e = w.spawn(on_sphere(i, n_meshes, 40.0, seed, true)); w.insert_aabb(e, Aabb{{0, 0, 0}, {0.5f, 0.5f, 0.5f}}); ents.push_back(e); }
        for (uint32_t i = 0; i < n_lights; ++i) { Entity e = w.spawn(on_sphere(i, n_lights, 50.0, seed, false)); w.insert_point_light(e, 0.3f); ents.push_back(e); }
        ClusterCamera cam;
        mi_perspective_clip_from_view(3.14159265f / 4.0f, 16.0f / 9.0f, 0.1f, cam.clip_from_view);
        auto set_camera = [&](int frame) {  // many_cubes.rs:590-603: rotate_z(d); rotate_x(d), d = 0.15 / 60 per frame
            const float a = 0.5f * 0.0025f * (float)frame, sz = std::sin(a), cz = std::cos(a);
            Transform t;  // q = qz * qx
            t.rotation = {cz * sz, sz * sz, sz * cz, cz * cz};
            const GlobalTransform g = GlobalTransform::from(t);
            std::memcpy(cam.camera_affine, g.cols, sizeof cam.camera_affine);
            mi_compute_frustum(cam.clip_from_view, cam.camera_affine, 1000.0f, cam.frustum);
        };
        std::printf(", \"%s\": {", fused ? "fused_frame" : "three_systems");
        const int pcts[3] = {1, 10, 100};
        for (int pi = 0; pi < 3; ++pi) {
            const size_t k = ents.size() * (size_t)pcts[pi] / 100;
            std::vector<double> total, gather, device, apply;
            uint32_t visible = 0, waits = 0;
            uint64_t cluster_entries = 0;
            for (int f = 0; f < frames + 3; ++f) {
                // the game's own update: k entities move (spread over the whole scene)
                const size_t stride = ents.size() / std::max<size_t>(k, 1), off = (size_t)f % std::max<size_t>(stride, 1);
                for (size_t j = 0; j < k; ++j) w.transform_mut(ents[std::min(j * stride + off, ents.size() - 1)]).translation.y += 0.001f;
                set_camera(f);
                View view;
                std::memcpy(view.frustum, cam.frustum, sizeof view.frustum);
                const auto t0 = std::chrono::steady_clock::now();
                if (fused) {
                    const Mi355xPlugin::FrameOutput o = plugin.frame(w, {view}, &cam);
                    visible = (uint32_t)o.visible_entities[0].size();
                    cluster_entries = o.clusters.total_index_count;
                    waits = o.device_waits;
                    if (f >= 3) { gather.push_back(o.gather_s); device.push_back(o.device_s); apply.push_back(o.apply_s); }
                } else {
                    plugin.propagate_transforms(w);
                    plugin.check_visibility(w, {view});
                    visible = (uint32_t)plugin.visible_entities(0).size();
                    cluster_entries = plugin.assign_objects_to_clusters(w, cam).total_index_count;
                }
                const auto t1 = std::chrono::steady_clock::now();
                if (f >= 3) total.push_back(std::chrono::duration<double>(t1 - t0).count());
                w.clear_trackers();
            }
            auto med = [](std::vector<double> v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
            std::printf("%s\"%dpct_dirty\": {\"dirty_rows\": %zu, \"us_per_frame\": %.1f, \"visible\": %u, \"cluster_entries\": %llu", pi ? ", " : "", pcts[pi], k,
                        1e6 * med(total), visible, (unsigned long long)cluster_entries);
            if (fused) std::printf(", \"gather_us\": %.1f, \"library_calls_us\": %.1f, \"ecs_writes_us\": %.1f, \"device_waits\": %u", 1e6 * med(gather), 1e6 * med(device), 1e6 * med(apply), waits);
            std::printf("}");
        }
        std::printf("}");
    }
    std::printf("}\n");
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && std::string(argv[1]) == "--bench") {
        const uint32_t n = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 1000000u;
        return bench(n, n / 100, n / 10, argc > 3 ? std::atoi(argv[3]) : 10);
    }
    struct T { const char* name; void (*fn)(); };
    const T tests[] = {{"correct_parent_removed", correct_parent_removed},
                       {"did_propagate", did_propagate},
                       {"correct_children", correct_children},
                       {"correct_transforms_when_no_children", correct_transforms_when_no_children},
                       {"panic_when_hierarchy_cycle", panic_when_hierarchy_cycle},
                       {"global_transform_should_not_be_overwritten_after_reparenting", global_transform_should_not_be_overwritten_after_reparenting},
                       {"change_ticks_follow_set_if_neq", change_ticks_follow_set_if_neq},
                       {"visibility_propagation", visibility_propagation},
                       {"visibility_propagation_change_detection", visibility_propagation_change_detection},
                       {"view_visibility_lifecycle", view_visibility_lifecycle},
                       {"visible_entities_are_sorted_by_entity", visible_entities_are_sorted_by_entity},
                       {"lights_are_assigned_to_clusters", lights_are_assigned_to_clusters},
                       {"every_kind_of_light_is_clustered", every_kind_of_light_is_clustered},
                       {"light_probes_and_decals_are_clustered", light_probes_and_decals_are_clustered},
                       {"render_multidrawable_batch_set", render_multidrawable_batch_set},
                       {"both_forms_leave_the_same_world", both_forms_leave_the_same_world},
                       {"narrow_hierarchy_stays_with_the_stock_systems", narrow_hierarchy_stays_with_the_stock_systems},
                       {"sharded_plugin_leaves_the_same_world", sharded_plugin_leaves_the_same_world},
                       {"sharded_plugin_shards_trees_and_lights", sharded_plugin_shards_trees_and_lights},
                       {"big_flat_worlds_agree", big_flat_worlds_agree}};
    int n_failed_tests = 0, n_tests = 0;
    for (int form = 0; form < 2; ++form) {
        g_fused = form == 1;
        for (const T& t : tests) {
            if (form == 1 && (t.fn == both_forms_leave_the_same_world || t.fn == big_flat_worlds_agree || t.fn == sharded_plugin_leaves_the_same_world || t.fn == sharded_plugin_shards_trees_and_lights)) continue;  // (drive both forms themselves)
            const int before = g_failed;
            ++n_tests;
            try {
                t.fn();
            } catch (const std::exception& e) {
                ++g_failed;
                std::printf("  EXCEPTION in %s: %s\n", t.name, e.what());
            }
            std::printf("%s %s [%s]\n", g_failed == before ? "ok    " : "FAILED", t.name, g_fused ? "fused frame" : "three systems");
            if (g_failed != before) ++n_failed_tests;
        }
    }
    std::printf("%d tests, %d failed (%d checks)\n", n_tests, n_failed_tests, g_checks);
    return n_failed_tests ? 1 : 0;
}
