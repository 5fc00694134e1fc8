// bevy_mi355x_sharded.hpp -- the plugin over SEVERAL GPUs of one node, from one process and one thread: what north_star asks of the
// host side ("entity ranges shard across the 8 GPUs of one node with an RCCL all-gather of the packed ViewVisibility bitmask") behind
// the same boundary as bevy_mi355x_host.hpp's Mi355xPlugin, not behind torchrun.  A Bevy App is one World in one process; a plugin
// that wants every GPU creates one context per device and drives them all from the thread its system runs on:
//
//   Mi355xShardedPlugin plugin({0, 1, 2, 3, 4, 5, 6, 7});           // Mi355xRenderPrepPlugin { devices: vec![0, .., 7] } in the Rust crate
//   plugin.frame(world, views)                                        // TransformSystems::Propagate .. MarkNewlyHiddenEntitiesInvisible
//       rows (Entity order) in contiguous 256-aligned ranges, one per context; per frame and context the rows a
//       Changed<Transform> query yields go up (mi_upload_transforms_indexed), ONE frame call each (mi_propagate_and_cull_views: enqueue
//       only, the devices run side by side), mi_exchange_group_flush = the N in-place all-gathers between ncclGroupStart / ncclGroupEnd,
//       each context's changed GlobalTransforms back (its shard only: GlobalTransform stays sharded), and ONE device-to-host copy of
//       ONE context's gathered buffer -- every shard's masks of every view -- from which ViewVisibility and VisibleEntities are written.
//
// Scope: flat Worlds (every entity a root without children -- configs[1] / configs[3]); a World with ChildOf is refused (its sharding is
// by root subtree, bevy_amd/sharding.py: shard_hierarchy, and stays with the single-device plugin here).  Lights, clusters and batching
// are the single-device plugin's: SURVEY 8e shards clusters over LIGHTS (optional at 100 k) and leaves batching as replicas.
// A device named twice in the list ({0, 0, 0}: three shards on one GPU) cannot form a communicator: such a plugin runs without the
// exchange and reads each shard's masks from its own context -- which is how tests/cpp/host_systems_test.cpp exercises several shards on
// a one-GPU box; with distinct devices ({0} there) it goes through RCCL.  RCCL is loaded with dlopen: no link-time dependency.
#pragma once

#include <dlfcn.h>

#include "bevy_mi355x_host.hpp"

namespace bevy_mi355x {

class Mi355xShardedPlugin {
  public:
    explicit Mi355xShardedPlugin(std::vector<int> devices) : devices_(std::move(devices)) {
        if (devices_.empty()) throw std::runtime_error("Mi355xShardedPlugin: no device");
        bool distinct = true;
        for (size_t i = 0; i < devices_.size(); ++i)
            for (size_t j = 0; j < i; ++j) distinct = distinct && devices_[i] != devices_[j];
        ctxs_.assign(devices_.size(), nullptr);
        for (size_t d = 0; d < devices_.size(); ++d)
            if (mi_ctx_create(devices_[d], nullptr, &ctxs_[d]) != MI_OK) {
                const std::string msg = mi_last_error_string(nullptr);
                for (mi_ctx* c : ctxs_)
                    if (c) mi_ctx_destroy(c);
                throw std::runtime_error("mi_ctx_create: " + msg);
            }
        if (distinct) load_rccl();
    }
    ~Mi355xShardedPlugin() {
        for (mi_ctx* c : ctxs_)
            if (c) {
                mi_exchange_configure(c, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0);
                mi_ctx_destroy(c);
            }
        if (comm_destroy_)
            for (void* c : comms_)
                if (c) comm_destroy_(c);
    }
    Mi355xShardedPlugin(const Mi355xShardedPlugin&) = delete;
    Mi355xShardedPlugin& operator=(const Mi355xShardedPlugin&) = delete;

    bool exchanged() const { return !comms_.empty(); }  // the masks travel through the RCCL all-gather (else: read per context)
    const std::string& exchange_note() const { return note_; }
    size_t shards() const { return ctxs_.size(); }

    struct FrameOutput {
        std::vector<std::vector<Entity>> visible_entities;  // per view: VisibleEntities::get(class 0), ascending by Entity
        uint32_t changed_global_transforms = 0;
        bool exchanged = false;
    };

    FrameOutput frame(World& w, const std::vector<View>& views) {
        FrameOutput out;
        out.exchanged = exchanged();
        sync_structure(w);
        const uint32_t n = (uint32_t)entity_of_row_.size(), N = (uint32_t)ctxs_.size();
        if (n == 0) return out;
        upload_bounds(w);
        // ---- in: the rows a Changed<Transform> query yields, by shard
        std::vector<std::vector<uint32_t>> rows(N);
        std::vector<std::vector<float>> t(N), r(N), s(N);
        for (uint32_t i : w.touched_) {
            if (!w.moved_[i]) continue;
            const uint32_t row = row_of_index_[i], d = row / rows_per_;
            const Transform& tr = w.transform_[i];
            rows[d].push_back(row - d * rows_per_);
            t[d].insert(t[d].end(), {tr.translation.x, tr.translation.y, tr.translation.z});
            r[d].insert(r[d].end(), {tr.rotation.x, tr.rotation.y, tr.rotation.z, tr.rotation.w});
            s[d].insert(s[d].end(), {tr.scale.x, tr.scale.y, tr.scale.z});
        }
        for (uint32_t d = 0; d < N; ++d) {
            if (!cnt_[d]) continue;
            check(d, mi_upload_transforms_indexed(ctxs_[d], (uint32_t)rows[d].size(), rows[d].data(), t[d].data(), r[d].data(), s[d].data()));
            if (rows[d].empty()) {  // keep "nothing changed" distinct from "no change information" (= all dirty)
                const uint8_t zero = 0;
                check(d, mi_upload_changed(ctxs_[d], 0, 1, &zero));
            }
        }
        // ---- run: one frame call per context (enqueue only: the devices run side by side), then the all-gathers together
        std::vector<mi_view> mv(views.size());
        for (size_t v = 0; v < views.size(); ++v) {
            std::memset(&mv[v], 0, sizeof(mi_view));
            std::memcpy(mv[v].frustum, views[v].frustum, sizeof mv[v].frustum);
            mv[v].layer_mask = views[v].layer_mask;
            if (w.visible_entity_ranges() && views[v].has_range_index) {
                mv[v].flags |= MI_VIEW_FLAG_RANGES;
                std::memcpy(mv[v].position, &views[v].position, 12);
            }
        }
        const uint32_t n_views = (uint32_t)views.size();
        if (n_views != exchange_views_) configure_exchange(n_views);
        for (uint32_t d = 0; d < N; ++d) {
            if (n_views) check(d, mi_propagate_and_cull_views(ctxs_[d], mv.data(), n_views, MI_CULL_CHANGED_ROWS | MI_CULL_END_FRAME));
            else if (cnt_[d]) check(d, mi_propagate(ctxs_[d], 0));
        }
        if (n_views && exchanged())
            check(0, mi_exchange_group_flush(ctxs_.data(), N, group_start_, group_end_));
        // ---- out: every shard's changed GlobalTransforms (GlobalTransform stays sharded: each context returns its own rows)
        std::vector<uint32_t> crow;
        std::vector<float> cg;
        for (uint32_t d = 0; d < N; ++d) {
            if (!cnt_[d]) continue;
            crow.resize(cnt_[d]);
            cg.resize(12 * (size_t)cnt_[d]);
            uint32_t count = 0;
            check(d, mi_download_changed_global_transforms(ctxs_[d], crow.data(), cg.data(), cnt_[d], &count));
            for (uint32_t k = 0; k < count; ++k) {
                const uint32_t i = entity_of_row_[d * rows_per_ + crow[k]].index;
                std::memcpy(w.global_[i].cols, &cg[12 * (size_t)k], 48);
                w.global_changed_[i] = 1;
                w.touch(i);
            }
            out.changed_global_transforms += count;
        }
        if (!n_views) return out;
        // ---- the masks: ONE copy of one context's gathered buffer ([rank][view][word]) -- or, without the exchange, each context's own
        const uint64_t wpv = rows_per_ / 64u;
        masks_.assign((size_t)N * n_views * wpv, 0ull);
        if (exchanged()) {
            check(0, mi_exchange_download(ctxs_[0], masks_.data(), masks_.size() * 8));
        } else {
            std::vector<uint32_t> bits;
            for (uint32_t d = 0; d < N; ++d) {
                if (!cnt_[d]) continue;
                bits.assign(((size_t)cnt_[d] + 31) / 32 + 1, 0u);
                for (uint32_t v = 0; v < n_views; ++v) {
                    check(d, mi_download_visibility(ctxs_[d], v, bits.data()));
                    uint64_t* dst = &masks_[((size_t)d * n_views + v) * wpv];
                    for (size_t k = 0; k < ((size_t)cnt_[d] + 31) / 32; ++k) dst[k >> 1] |= (uint64_t)bits[k] << (32u * (k & 1u));
                }
            }
        }
        // ---- ECS writes: VisibilitySystems::CheckVisibility between the stock reset and mark-newly-hidden systems
        w.reset_view_visibility();
        out.visible_entities.resize(n_views);
        for (uint32_t v = 0; v < n_views; ++v)
            for (uint32_t d = 0; d < N; ++d) {
                const uint64_t* src = &masks_[((size_t)d * n_views + v) * wpv];
                for (uint64_t k = 0; k < (cnt_[d] + 63u) / 64u; ++k)
                    for (uint64_t m = src[k]; m; m &= m - 1) {
                        const uint32_t local = (uint32_t)(k * 64u) + (uint32_t)__builtin_ctzll(m);
                        if (local >= cnt_[d]) break;
                        const Entity e = entity_of_row_[d * rows_per_ + local];
                        w.set_visible(e);
                        out.visible_entities[v].push_back(e);
                    }
            }
        w.mark_newly_hidden_entities_invisible();
        return out;
    }

  private:
    void check(uint32_t d, int32_t rc) {
        if (rc == MI_OK) return;
        throw std::runtime_error("bevy_mi355x (shard " + std::to_string(d) + ") error " + std::to_string(rc) + ": " + mi_last_error_string(ctxs_[d]));
    }
    void load_rccl() {
        void* lib = nullptr;
        for (const char* path : {std::getenv("MI_RCCL_LIB") ? std::getenv("MI_RCCL_LIB") : "librccl.so", "/opt/rocm/lib/librccl.so",
                                 "/usr/local/lib/python3.10/dist-packages/torch/lib/librccl.so"})
            if (!lib) lib = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
        if (!lib) { note_ = "librccl.so could not be loaded: the masks are read per context"; return; }
        typedef int (*init_all_fn)(void**, int, const int*);
        auto init_all = (init_all_fn)dlsym(lib, "ncclCommInitAll");
        comm_destroy_ = (int (*)(void*))dlsym(lib, "ncclCommDestroy");
        all_gather_ = dlsym(lib, "ncclAllGather");
        group_start_ = dlsym(lib, "ncclGroupStart");
        group_end_ = dlsym(lib, "ncclGroupEnd");
        if (!init_all || !comm_destroy_ || !all_gather_ || !group_start_ || !group_end_) { note_ = "RCCL symbols missing: the masks are read per context"; return; }
        comms_.assign(devices_.size(), nullptr);
        if (init_all(comms_.data(), (int)devices_.size(), devices_.data()) != 0) {
            comms_.clear();
            note_ = "ncclCommInitAll failed: the masks are read per context";
            return;
        }
        note_ = "ncclCommInitAll over " + std::to_string(devices_.size()) + " device(s), MI_EXCHANGE_GROUPED";
    }
    void configure_exchange(uint32_t n_views) {
        exchange_views_ = n_views;
        if (!exchanged() || n_views == 0 || rows_per_ == 0) return;
        const uint64_t wpv = rows_per_ / 64u, block = (uint64_t)n_views * wpv * 8u;
        for (uint32_t d = 0; d < ctxs_.size(); ++d) {
            check(d, mi_exchange_configure(ctxs_[d], nullptr, nullptr, nullptr, 0, 0, 0, 0, 0));
            check(d, mi_exchange_set_mode(ctxs_[d], MI_EXCHANGE_GROUPED));
            check(d, mi_exchange_configure_owned(ctxs_[d], &comms_[d], 1, all_gather_, 3, (uint32_t)ctxs_.size(), wpv, (uint64_t)d * n_views * wpv, block, d));
        }
    }
    // Entity -> (shard, row): rows in Entity::to_bits order, cut into contiguous 256-aligned ranges (a mask word never straddles shards)
    void sync_structure(World& w) {
        if (seen_version_ == w.structure_version_) return;
        std::vector<Entity> ents = w.entities();
        for (Entity e : ents)
            if (w.rec_[e.index].parent || !w.rec_[e.index].children.empty())
                throw std::runtime_error("Mi355xShardedPlugin: a World with ChildOf shards by root subtree (bevy_amd/sharding.py); use Mi355xPlugin");
        std::sort(ents.begin(), ents.end(), [](Entity a, Entity b) { return a.to_bits() < b.to_bits(); });
        entity_of_row_ = ents;
        const uint32_t n = (uint32_t)ents.size(), N = (uint32_t)ctxs_.size();
        row_of_index_.assign(w.rec_.size(), MI_NO_PARENT);
        for (uint32_t row = 0; row < n; ++row) row_of_index_[ents[row].index] = row;
        rows_per_ = ((n + N - 1) / N + 255u) / 256u * 256u;
        if (rows_per_ == 0) rows_per_ = 256u;
        cnt_.assign(N, 0u);
        for (uint32_t d = 0; d < N; ++d) {
            const uint32_t lo = std::min<uint64_t>(n, (uint64_t)d * rows_per_);
            cnt_[d] = std::min(n - lo, rows_per_);
            check(d, mi_columns_resize(ctxs_[d], cnt_[d]));
            if (!cnt_[d]) continue;
            const uint32_t m = cnt_[d];
            std::vector<float> t(3 * (size_t)m), r(4 * (size_t)m), s(3 * (size_t)m), g(12 * (size_t)m);
            std::vector<uint8_t> changed(m), vv(m);
            std::vector<uint64_t> keys(m);
            for (uint32_t k = 0; k < m; ++k) {
                const Entity e = ents[lo + k];
                const World::Rec& rec = w.rec_[e.index];
                std::memcpy(&t[3 * (size_t)k], &w.transform_[e.index].translation, 12);
                std::memcpy(&r[4 * (size_t)k], &w.transform_[e.index].rotation, 16);
                std::memcpy(&s[3 * (size_t)k], &w.transform_[e.index].scale, 12);
                std::memcpy(&g[12 * (size_t)k], w.global_[e.index].cols, 48);
                changed[k] = (rec.transform_changed || rec.added || rec.parent_changed || rec.orphaned) ? 1 : 0;
                vv[k] = w.vv_[e.index];
                keys[k] = e.to_bits();
            }
            check(d, mi_upload_transforms(ctxs_[d], 0, m, t.data(), r.data(), s.data()));
            check(d, mi_upload_global_transforms(ctxs_[d], 0, m, g.data()));
            check(d, mi_upload_view_visibility(ctxs_[d], 0, m, vv.data()));
            check(d, mi_upload_entity_keys(ctxs_[d], 0, m, keys.data()));
            check(d, mi_upload_changed(ctxs_[d], 0, m, changed.data()));
        }
        seen_version_ = w.structure_version_;
        seen_bounds_ = 0;
        exchange_views_ = 0xFFFFFFFFu;  // the shard size may have changed: the gathered buffers are laid out again
    }
    void upload_bounds(World& w) {
        if (seen_bounds_ == w.bounds_version_) return;
        seen_bounds_ = w.bounds_version_;
        for (uint32_t d = 0; d < ctxs_.size(); ++d) {
            const uint32_t m = cnt_[d], lo = d * rows_per_;
            if (!m) continue;
            std::vector<float> c(3 * (size_t)m, 0.f), h(3 * (size_t)m, 0.f), ranges(2 * (size_t)m, 0.f);
            std::vector<uint8_t> flags(m);
            std::vector<uint32_t> layers(m);
            for (uint32_t k = 0; k < m; ++k) {
                const World::Rec& e = w.rec_[entity_of_row_[lo + k].index];
                flags[k] = (uint8_t)(((!e.has_visibility || e.visibility != Visibility::Hidden) ? MI_FLAG_INHERITED_VISIBLE : 0u) | (e.aabb ? MI_FLAG_HAS_AABB : 0u) |
                                     (e.no_frustum_culling ? MI_FLAG_NO_FRUSTUM_CULLING : 0u));  // (flat rows: InheritedVisibility = Visibility != Hidden, visibility/mod.rs:650-660)
                layers[k] = e.render_layers;
                if (e.visibility_range) {
                    flags[k] |= (uint8_t)(MI_FLAG_HAS_VISIBILITY_RANGE | (e.visibility_range->use_aabb ? MI_FLAG_RANGE_USE_AABB : 0u));
                    ranges[2 * (size_t)k] = e.visibility_range->start_margin_start;
                    ranges[2 * (size_t)k + 1] = e.visibility_range->end_margin_end;
                }
                if (e.aabb) { std::memcpy(&c[3 * (size_t)k], &e.aabb->center, 12); std::memcpy(&h[3 * (size_t)k], &e.aabb->half_extents, 12); }
            }
            check(d, mi_upload_bounds(ctxs_[d], 0, m, c.data(), h.data(), flags.data(), layers.data()));
            check(d, mi_upload_visibility_ranges(ctxs_[d], 0, m, w.visible_entity_ranges() ? ranges.data() : nullptr));
        }
    }

    std::vector<int> devices_;
    std::vector<mi_ctx*> ctxs_;
    std::vector<void*> comms_;
    int (*comm_destroy_)(void*) = nullptr;
    void *all_gather_ = nullptr, *group_start_ = nullptr, *group_end_ = nullptr;
    std::string note_ = "a device is named more than once: no communicator, the masks are read per context";
    std::vector<Entity> entity_of_row_;
    std::vector<uint32_t> row_of_index_, cnt_;
    std::vector<uint64_t> masks_;
    uint32_t rows_per_ = 0, exchange_views_ = 0xFFFFFFFFu;
    uint64_t seen_version_ = 0, seen_bounds_ = 0;
};

}  // namespace bevy_mi355x
