// ctx_cluster.cpp -- clustered-forward light assignment entry points.
#include "ctx.h"

#include <cmath>

using namespace mi;
using namespace mi_detail;

extern "C" {

// =============================================================================================
// clustering
// =============================================================================================
int32_t mi_cluster_upload_object_layers_hi(mi_ctx* ctx, uint32_t n, const uint32_t* layer_mask_hi) {
    ENTER(ctx);
    if (n != ctx->cl_n) return fail(ctx, MI_ERR_INVALID_ARG, "mi_cluster_upload_object_layers_hi: %u objects, %u uploaded", n, ctx->cl_n);
    int32_t rc;
    if ((rc = cluster_join(ctx))) return rc;
    ctx->cl_have_layers_hi = false;
    ctx->cl_assigned = false;
    for (auto& parked : ctx->cl_parked) parked.assigned = false;  // (every view's assignment was over the old layers)
    if (!layer_mask_hi || n == 0) return MI_OK;
    if ((rc = ensure(ctx, ctx->cl_layers_hi, (size_t)n * 4))) return rc;
    if ((rc = upload(ctx, ctx->cl_layers_hi.p, layer_mask_hi, (size_t)n * 4))) return rc;
    ctx->cl_have_layers_hi = true;
    return MI_OK;
}

int32_t mi_cluster_upload_objects(mi_ctx* ctx, uint32_t n, const float* pos_range, const uint8_t* obj_type,
                                  const uint32_t* layer_mask, const float* spot_dir, const float* spot_sin_cos) {
    ENTER(ctx);
    if (n && !pos_range) return fail(ctx, MI_ERR_INVALID_ARG, "mi_cluster_upload_objects: pos_range NULL");
    bool any_spot = false;
    if (obj_type)
        for (uint32_t i = 0; i < n; ++i) {
            if (obj_type[i] > MI_OBJ_DECAL) return fail(ctx, MI_ERR_INVALID_ARG, "object %u: unknown type %u", i, obj_type[i]);
            any_spot |= obj_type[i] == MI_OBJ_SPOT_LIGHT;
        }
    if (any_spot && !spot_sin_cos) return fail(ctx, MI_ERR_INVALID_ARG, "spot lights need spot_sin_cos");
    int32_t rc;
    if ((rc = cluster_join(ctx))) return rc;
    // From here on the old set is gone: an upload that fails half way leaves an EMPTY set (nothing can be assigned over columns of
    // two different uploads), not the old one with some new columns.
    const uint32_t n_before = ctx->cl_n;
    ctx->cl_n = 0;
    ctx->cl_assigned = false;
    for (auto& parked : ctx->cl_parked) parked.assigned = false;  // (every view's assignment was over the old objects)
    const bool bound_before = ctx->cl_rows_bound;
    ctx->cl_rows_bound = false;
    if ((rc = ensure(ctx, ctx->cl_pos, (size_t)n * 16))) return rc;
    if ((rc = upload(ctx, ctx->cl_pos.p, pos_range, (size_t)n * 16))) return rc;
    ctx->cl_have_type = obj_type != nullptr;
    if (obj_type) {
        if ((rc = ensure(ctx, ctx->cl_type, n))) return rc;
        if ((rc = upload(ctx, ctx->cl_type.p, obj_type, n))) return rc;
    }
    ctx->cl_have_layers_hi = false;  // (mi_cluster_upload_object_layers_hi follows, for objects on layers 32..63)
    ctx->cl_have_layers = layer_mask != nullptr;
    if (layer_mask) {
        if ((rc = ensure(ctx, ctx->cl_layers, (size_t)n * 4))) return rc;
        if ((rc = upload(ctx, ctx->cl_layers.p, layer_mask, (size_t)n * 4))) return rc;
    }
    ctx->cl_have_spot = spot_sin_cos != nullptr;
    ctx->cl_have_spot_dir = spot_dir != nullptr;
    if (spot_dir) {
        if ((rc = ensure(ctx, ctx->cl_dir, (size_t)n * 12))) return rc;
        if ((rc = upload(ctx, ctx->cl_dir.p, spot_dir, (size_t)n * 12))) return rc;
    }
    if (spot_sin_cos) {
        if ((rc = ensure(ctx, ctx->cl_sincos, (size_t)n * 8))) return rc;
        if ((rc = upload(ctx, ctx->cl_sincos.p, spot_sin_cos, (size_t)n * 8))) return rc;
    }
    ctx->cl_any_spot = any_spot;
    ctx->cl_n = n;
    ctx->cl_rows_bound = bound_before && n == n_before;  // a different object set: the row binding has to be renewed
    return MI_OK;
}

int32_t mi_cluster_bind_objects_to_rows(mi_ctx* ctx, uint32_t first_row, uint32_t n_objects) {
    ENTER(ctx);
    if (n_objects == 0) {
        ctx->cl_rows_bound = false;
        ctx->cl_assigned = false;
        for (auto& parked : ctx->cl_parked) parked.assigned = false;
        return MI_OK;
    }
    if (n_objects != ctx->cl_n)
        return fail(ctx, MI_ERR_INVALID_ARG, "mi_cluster_bind_objects_to_rows: %u objects, %u uploaded", n_objects, ctx->cl_n);
    int32_t rc = check_rows(ctx, first_row, n_objects, "mi_cluster_bind_objects_to_rows");
    if (rc) return rc;
    ctx->cl_rows_bound = true;
    ctx->cl_rows_listed = false;
    ctx->cl_first_row = first_row;
    ctx->cl_assigned = false;
    for (auto& parked : ctx->cl_parked) parked.assigned = false;  // (the binding is shared: every view's assignment was over the old one)
    return MI_OK;
}

int32_t mi_cluster_bind_objects_to_row_list(mi_ctx* ctx, uint32_t n_objects, const uint32_t* rows) {
    ENTER(ctx);
    if (n_objects == 0) {
        ctx->cl_rows_bound = false;
        return MI_OK;
    }
    if (!rows) return fail(ctx, MI_ERR_INVALID_ARG, "mi_cluster_bind_objects_to_row_list: rows NULL");
    if (n_objects != ctx->cl_n)
        return fail(ctx, MI_ERR_INVALID_ARG, "mi_cluster_bind_objects_to_row_list: %u objects, %u uploaded", n_objects, ctx->cl_n);
    for (uint32_t i = 0; i < n_objects; ++i)
        if (rows[i] >= ctx->n) return fail(ctx, MI_ERR_INVALID_ARG, "mi_cluster_bind_objects_to_row_list: object %u names row %u of %u", i, rows[i], ctx->n);
    int32_t rc;
    if ((rc = cluster_join(ctx))) return rc;
    if ((rc = ensure(ctx, ctx->cl_row_list, (size_t)n_objects * 4))) return rc;
    if ((rc = upload(ctx, ctx->cl_row_list.p, rows, (size_t)n_objects * 4))) return rc;
    ctx->cl_rows_bound = true;
    ctx->cl_rows_listed = true;
    ctx->cl_first_row = 0;
    ctx->cl_assigned = false;
    for (auto& parked : ctx->cl_parked) parked.assigned = false;  // (as above)
    return MI_OK;
}

// The selected slot's state lives in the context's own fields; the others are parked (ctx.h, ClusterSlot).
static void cluster_slot_exchange(mi_ctx* ctx, mi_ctx::ClusterSlot& s) {
    std::swap(ctx->cl_planes, s.planes);
    std::swap(ctx->cl_spheres, s.spheres);
    std::swap(ctx->cl_remap, s.remap);
    std::swap(ctx->cl_bind_oc, s.bind_oc);
    std::swap(ctx->cl_bind_idx, s.bind_idx);
    std::swap(ctx->cl_block_counts, s.block_counts);
    std::swap(ctx->cl_pair_cb, s.pair_cb);
    std::swap(ctx->cl_pair_mask, s.pair_mask);
    std::swap(ctx->cl_acc, s.acc);
    std::swap(ctx->cl_offsets, s.offsets);
    std::swap(ctx->cl_indices, s.indices);
    std::swap(ctx->cl_scalars, s.scalars);
    std::swap(ctx->cl_parity, s.parity);
    std::swap(ctx->cl_acc_clusters, s.acc_clusters);
    std::swap(ctx->cl_acc_blocks, s.acc_blocks);
    std::swap(ctx->cl_host_planes, s.host_planes);
    std::swap(ctx->cl_host_spheres, s.host_spheres);
    std::swap(ctx->cl_planes_host, s.planes_host);
    std::swap(ctx->cl_spheres_sent, s.spheres_sent);
    std::swap(ctx->cl_planes_epoch, s.planes_epoch);
    for (int k = 0; k < 3; ++k) std::swap(ctx->cl_plane_counts[k], s.plane_counts[k]);
    std::swap(ctx->cl_fill_pending, s.fill_pending);
    std::swap(ctx->cl_fill_job, s.fill_job);
    std::swap(ctx->cl_view, s.view);
    std::swap(ctx->cl_have_view, s.have_view);
    std::swap(ctx->cl_assigned, s.assigned);
}

int32_t mi_cluster_select_view(mi_ctx* ctx, uint32_t slot) {
    ENTER(ctx);
    if (slot >= MI_CLUSTER_MAX_VIEWS) return fail(ctx, MI_ERR_INVALID_ARG, "mi_cluster_select_view: slot %u of %u", slot, MI_CLUSTER_MAX_VIEWS);
    if (slot == ctx->cl_slot) return MI_OK;
    int32_t rc = cluster_join(ctx);  // (a deferred fill of the slot that is left goes out now; a side-stream assignment is joined)
    if (rc) return rc;
    cluster_slot_exchange(ctx, ctx->cl_parked[ctx->cl_slot]);  // park the selected slot ...
    cluster_slot_exchange(ctx, ctx->cl_parked[slot]);          // ... and take the other one out
    ctx->cl_slot = slot;
    return MI_OK;
}

namespace {
// The one library call on the device side of the path: ln() in view_z_to_z_slice (assign.rs:1057).  The device carries glibc's logf;
// whether THIS host's libm is that function is checked here, once per context, over a fixed table (every binade from 2^-12 to 2^20,
// 96 mantissas each, plus the values around 1 where the algorithm switches branches) instead of being assumed (include/bevy_mi355x.h
// used to say "glibc >= 2.28 on x86-64 is the supported host libm" and leave it at that).
int32_t libm_self_test(mi_ctx* ctx) {
    if (ctx->libm_state == 1) return MI_OK;
    if (ctx->libm_state == 0) {
        std::vector<float> in;
        uint64_t rng = 0x9E3779B97F4A7C15ull;
        for (int e = -12; e <= 20; ++e)
            for (int k = 0; k < 96; ++k) {
                rng ^= rng << 13, rng ^= rng >> 7, rng ^= rng << 17;
                const uint32_t bits = ((uint32_t)(e + 127) << 23) | (uint32_t)(rng & 0x7FFFFFu);
                float x;
                memcpy(&x, &bits, 4);
                in.push_back(x);
            }
        for (int k = -64; k <= 64; ++k) {
            const uint32_t bits = 0x3f800000u + (uint32_t)k;
            float x;
            memcpy(&x, &bits, 4);
            in.push_back(x);
        }
        const uint32_t n = (uint32_t)in.size();
        std::vector<float> out(n);
        int32_t rc = mi_debug_logf(ctx, in.data(), out.data(), n);
        if (rc) return rc;
        uint32_t bad = 0, first = 0;
        for (uint32_t i = 0; i < n; ++i) {
            const float h = logf(in[i]);
            if (memcmp(&h, &out[i], 4) != 0 && !bad++) first = i;
        }
        if (getenv("MI_DEBUG_FORCE_LIBM_MISMATCH")) bad = bad ? bad : 1;  // (test hook: tests/test_gpu_cluster.py)
        ctx->libm_state = bad ? 2 : 1;
        if (bad) {
            uint32_t xb, hb, db;
            const float h = logf(in[first]);
            memcpy(&xb, &in[first], 4), memcpy(&hb, &h, 4), memcpy(&db, &out[first], 4);
            return fail(ctx, MI_ERR_DEVICE, "the host's logf differs from the device's (glibc's) at %u of %u probes, first at 0x%08x: host 0x%08x, device 0x%08x -- "
                        "cluster z slices could differ from this host's own assign_objects_to_clusters: the stock system must assign", bad, n, xb, hb, db);
        }
        return MI_OK;
    }
    return fail(ctx, MI_ERR_DEVICE, "the host's logf differs from the device's (checked when the first perspective view was uploaded): the stock system must assign");
}
}  // namespace

int32_t mi_cluster_upload_view(mi_ctx* ctx, const mi_cluster_view* view) {
    ENTER(ctx);
    if (!view || !view->x_planes || !view->y_planes || !view->z_planes) return fail(ctx, MI_ERR_INVALID_ARG, "mi_cluster_upload_view: NULL");
    if (!view->is_orthographic) {  // (an orthographic view's z slices are linear in view_z: no ln, assign.rs:1051-1054)
        const int32_t rcl = libm_self_test(ctx);
        if (rcl) return rcl;
    }
    const uint64_t C = (uint64_t)view->dims[0] * view->dims[1] * view->dims[2];
    if (C == 0 || C > 4096) return fail(ctx, MI_ERR_INVALID_ARG, "cluster count %llu outside 1..4096 (assign.rs:410-413)", (unsigned long long)C);
    const uint32_t nx = view->dims[0] + 1, ny = view->dims[1] + 1, nz = view->dims[2] + 1;
    int32_t rc;
    // The cluster planes live in VIEW space: they depend on the projection, the grid and near / far -- but near follows the
    // camera's scale (assign.rs:345,366), which moves by an ulp as the camera turns, so the table is new nearly every frame.
    // It travels as a kernel argument, or -- big tables -- is staged in pinned memory and read from there (see ctx.h);
    // stage_cluster_planes re-stages it when the arena wraps.
    ctx->cl_planes_host.resize((size_t)(nx + ny + nz) * 4);
    memcpy(ctx->cl_planes_host.data(), view->x_planes, (size_t)nx * 16);
    memcpy(ctx->cl_planes_host.data() + 4 * (size_t)nx, view->y_planes, (size_t)ny * 16);
    memcpy(ctx->cl_planes_host.data() + 4 * (size_t)(nx + ny), view->z_planes, (size_t)nz * 16);
    ctx->cl_plane_counts[0] = nx;
    ctx->cl_plane_counts[1] = ny;
    ctx->cl_plane_counts[2] = nz;
    ctx->cl_planes_epoch = ~0ull;
    ClusterViewDev& d = ctx->cl_view;
    memcpy(d.dims, view->dims, sizeof d.dims);
    d.is_orthographic = view->is_orthographic;
    d.view_layer_mask = view->view_layer_mask;
    d.view_layer_mask_hi = view->view_layer_mask_hi;
    d.n_clusters = (uint32_t)C;
    memcpy(d.cluster_factors, view->cluster_factors, sizeof d.cluster_factors);
    memcpy(d.view_from_world, view->view_from_world, sizeof d.view_from_world);
    memcpy(d.clip_from_view, view->clip_from_view, sizeof d.clip_from_view);
    memcpy(d.view_from_world_scale, view->view_from_world_scale, sizeof d.view_from_world_scale);
    d.view_from_world_scale_max = view->view_from_world_scale_max;
    memcpy(d.frustum, view->frustum, sizeof d.frustum);
    d.x_planes = d.y_planes = d.z_planes = nullptr;  // staged at launch time
    d.cluster_spheres = nullptr;
    if (view->cluster_spheres) {  // view space as well (compute_aabb_for_cluster, assign.rs:834-900)
        const void* spheres_before = ctx->cl_spheres.p;
        if ((rc = ensure(ctx, ctx->cl_spheres, (size_t)C * 16))) return rc;
        if (ctx->cl_spheres.p != spheres_before) ctx->cl_spheres_sent.clear();
        if (ctx->cl_spheres_sent.size() != (size_t)C * 4 || memcmp(view->cluster_spheres, ctx->cl_spheres_sent.data(), (size_t)C * 16) != 0) {
            if ((rc = upload(ctx, ctx->cl_spheres.p, view->cluster_spheres, (size_t)C * 16))) return rc;
            ctx->cl_spheres_sent.assign(view->cluster_spheres, view->cluster_spheres + (size_t)C * 4);
        }
        d.cluster_spheres = (const float*)ctx->cl_spheres.p;
    }
    ctx->cl_have_view = true;
    ctx->cl_assigned = false;
    return MI_OK;
}

}  // extern "C"

namespace mi_detail {

// Everything on the main stream that touches what the cluster kernels read or write goes through here first.
int32_t cluster_fill_join(mi_ctx* ctx) {
    if (!ctx->cl_fill_pending) return MI_OK;
    ctx->cl_fill_pending = false;
    ProfScope ps(ctx, K_CLUSTER_FILL);
    HIP_TRY(ctx, launch_cluster_fill(ctx->cl_fill_job.w, ctx->cl_fill_job.n_clusters, ctx->cl_fill_job.n_objects, ctx->stream));
    return MI_OK;
}

int32_t cluster_join(mi_ctx* ctx) {
    {
        int32_t rcf = cluster_fill_join(ctx);
        if (rcf) return rcf;
    }
    if (!ctx->cl_on_side) return MI_OK;
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_cl_done, 0));
    ctx->cl_on_side = false;
    return MI_OK;
}

// ---- one assignment = a walk and a fill over one SET of working buffers ---------------------------------------------------
// Three sets rotate (accumulators, count matrix, pair list): assignment k walks into set k % 3 and its fill zeroes set
// (k + 2) % 3 for assignment k + 2.  Two would do while walk and fill run one behind the other; the third lets the fill of
// assignment k share a launch with the walk of assignment k + 1 (both ride in the same frame kernel).
namespace {
constexpr uint32_t CL_SETS = 3;

struct ClusterPrep {
    ClusterObjects o{};
    ClusterWork w{};
    uint32_t C = 0;
    size_t acc_words = 0, off_totals = 0, off_misc = 0, mat_bytes = 0, pair_slots = 0;
};

// the view's plane table in the staging arena (again, if the arena has wrapped since)
int32_t stage_cluster_planes(mi_ctx* ctx) {
    if (ctx->cl_planes_epoch == ctx->stage_epoch && ctx->cl_view.x_planes) return MI_OK;
    void* st = nullptr;
    int32_t rc = stage_alloc(ctx, ctx->cl_planes_host.size() * 4, &st);
    if (rc) return rc;
    memcpy(st, ctx->cl_planes_host.data(), ctx->cl_planes_host.size() * 4);
    void* dev = nullptr;
    HIP_TRY(ctx, hipHostGetDevicePointer(&dev, st, 0));
    ClusterViewDev& d = ctx->cl_view;
    d.x_planes = (const float*)dev;
    d.y_planes = d.x_planes + 4 * (size_t)ctx->cl_plane_counts[0];
    d.z_planes = d.y_planes + 4 * (size_t)ctx->cl_plane_counts[1];
    ctx->cl_planes_epoch = ctx->stage_epoch;
    return MI_OK;
}

int32_t cluster_objects(mi_ctx* ctx, bool derive, ClusterObjects* po) {
    if (!ctx->cl_have_view) return fail(ctx, MI_ERR_NOT_READY, "mi_cluster_assign_resident: no view uploaded");
    {
        int32_t rcs = stage_cluster_planes(ctx);
        if (rcs) return rcs;
    }
    if (ctx->cl_any_spot && !ctx->cl_view.cluster_spheres)
        return fail(ctx, MI_ERR_INVALID_ARG, "spot lights present but mi_cluster_view.cluster_spheres is NULL");
    ClusterObjects& o = *po;
    o = ClusterObjects{};
    o.n = ctx->cl_n;
    o.pos_range = (const float*)ctx->cl_pos.p;
    o.obj_type = ctx->cl_have_type ? (const uint8_t*)ctx->cl_type.p : nullptr;
    o.layer_mask = ctx->cl_have_layers ? (const uint32_t*)ctx->cl_layers.p : nullptr;
    o.layer_mask_hi = ctx->cl_have_layers_hi ? (const uint32_t*)ctx->cl_layers_hi.p : nullptr;
    o.spot_dir = ctx->cl_have_spot_dir ? (const float*)ctx->cl_dir.p : nullptr;
    if (ctx->cl_any_spot && !o.spot_dir && !ctx->cl_rows_bound)
        return fail(ctx, MI_ERR_INVALID_ARG, "spot lights need spot_dir unless the objects are bound to rows (mi_cluster_bind_objects_to_rows)");
    o.spot_sin_cos = ctx->cl_have_spot ? (const float*)ctx->cl_sincos.p : nullptr;
    if (ctx->cl_rows_bound) {
        if (!ctx->cl_rows_listed && (uint64_t)ctx->cl_first_row + o.n > ctx->n)
            return fail(ctx, MI_ERR_NOT_READY, "cluster objects are bound to rows [%u,%u) but the context has %u rows", ctx->cl_first_row,
                        ctx->cl_first_row + o.n, ctx->n);
        o.first_row = ctx->cl_first_row;
        o.row_list = ctx->cl_rows_listed ? (const uint32_t*)ctx->cl_row_list.p : nullptr;
        if (derive) {
            o.derive = 1;
            o.row_global = ctx->g;
            o.row_changed = ctx->cl_derive_changed;
            o.changed_gen = ctx->changed_gen;
            o.derive_resident = ctx->cl_derive_resident ? 1u : 0u;
            o.n_views = ctx->n_views;
            o.row_translation = ctx->t;
            o.row_rotation = ctx->r;
            o.row_scale = ctx->s;
            o.row_aabb_center = ctx->c;
            o.row_aabb_half = ctx->h;
            o.row_range = ctx->have_ranges ? ctx->range : nullptr;
            o.row_flags = ctx->flags;
            o.row_layers = ctx->layers;
            o.row_layers_hi = ctx->layers_hi;
            const Columns cc = columns_of(ctx);
            o.row_summary = cc.row_summary_on ? cc.row_summary : nullptr;
        } else {
            o.row_global = ctx->g;
            o.row_vv = ctx->vv;
        }
    }
    return MI_OK;
}

// (re)allocates the working buffers for the current view / object count; `*fresh` = something was (re)allocated or zeroed on
// the main stream (a side stream has to be ordered behind that)
int32_t cluster_buffers(mi_ctx* ctx, ClusterPrep* p, bool* fresh) {
    const uint32_t C = ctx->cl_view.n_clusters;
    ClusterWork& w = p->w;
    p->C = C;
    // (one block more than the objects fill: blocks that are row tiles of the frame kernel -- ClusterWalkJob::inrow -- start and end
    // anywhere inside a tile, and the two forms share the buffers)
    w.n_blocks = std::max(1u, (p->o.n + CLUSTER_BLOCK - 1) / CLUSTER_BLOCK) + 1u;
    w.obj_delta = 0;
    p->off_totals = (6 * (size_t)C + 3) & ~(size_t)3;
    p->off_misc = p->off_totals + (((size_t)C + 3) & ~(size_t)3);
    p->acc_words = p->off_misc + 4;  // per set; 16-byte aligned sections: counts | totals | misc
    w.row_stride = (w.n_blocks + 7u) & ~7u;
    p->mat_bytes = (size_t)C * w.row_stride * 2;  // per set
    p->pair_slots = (size_t)w.n_blocks * C;       // per set
    const bool reshape = !ctx->cl_acc.p || ctx->cl_acc_clusters != C || ctx->cl_acc_blocks != w.n_blocks;
    const bool grow = ctx->cl_pair_cb.bytes < CL_SETS * p->pair_slots * 4 || ctx->cl_pair_mask.bytes < CL_SETS * p->pair_slots * 32 ||
                      ctx->cl_offsets.bytes < ((size_t)C + 1) * 4 || !ctx->cl_scalars.p || !ctx->cl_indices.p;
    *fresh = reshape || grow;
    int32_t rc;
    if (*fresh && (rc = cluster_join(ctx))) return rc;  // nothing may still be using what is about to move
    if ((rc = ensure(ctx, ctx->cl_pair_cb, CL_SETS * p->pair_slots * 4))) return rc;
    if ((rc = ensure(ctx, ctx->cl_pair_mask, CL_SETS * p->pair_slots * 32))) return rc;
    if ((rc = ensure(ctx, ctx->cl_offsets, ((size_t)C + 1) * 4))) return rc;
    if ((rc = ensure(ctx, ctx->cl_scalars, 16))) return rc;
    if (reshape) {
        // (re)shaped accumulators / count matrices start zeroed in every set; afterwards each fill keeps the set two ahead zeroed
        if ((rc = ensure(ctx, ctx->cl_acc, CL_SETS * p->acc_words * 4))) return rc;
        if ((rc = ensure(ctx, ctx->cl_block_counts, CL_SETS * p->mat_bytes))) return rc;
        HIP_TRY(ctx, hipMemsetAsync(ctx->cl_acc.p, 0, CL_SETS * p->acc_words * 4, ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(ctx->cl_block_counts.p, 0, CL_SETS * p->mat_bytes, ctx->stream));
        ctx->cl_acc_clusters = C;
        ctx->cl_acc_blocks = w.n_blocks;
    }
    if (!ctx->cl_indices.p && (rc = ensure(ctx, ctx->cl_indices, (size_t)1 << 20))) return rc;
    return MI_OK;
}

// the next set becomes current
void cluster_next_set(mi_ctx* ctx, ClusterPrep* p) {
    ctx->cl_parity = (ctx->cl_parity + 1u) % CL_SETS;
    const uint32_t cur = ctx->cl_parity, zero = (cur + 2u) % CL_SETS;
    ClusterWork& w = p->w;
    uint32_t* acc = (uint32_t*)ctx->cl_acc.p + cur * p->acc_words;
    w.block_counts = (uint16_t*)((char*)ctx->cl_block_counts.p + cur * p->mat_bytes);
    w.block_counts_next = (uint16_t*)((char*)ctx->cl_block_counts.p + zero * p->mat_bytes);
    w.counts = acc;
    w.totals = acc + p->off_totals;
    w.farthest_z = (float*)(acc + p->off_misc);
    w.pair_total = acc + p->off_misc + 1;
    w.acc_words = (uint32_t)p->acc_words;
    w.acc_next = (uint32_t*)ctx->cl_acc.p + zero * p->acc_words;
    w.pair_cb = (uint32_t*)ctx->cl_pair_cb.p + cur * p->pair_slots;
    w.pair_mask = (uint32_t*)ctx->cl_pair_mask.p + cur * p->pair_slots * 8;
    w.offsets = (uint32_t*)ctx->cl_offsets.p;
    w.indices = (uint32_t*)ctx->cl_indices.p;
    w.capacity = ctx->cl_indices.bytes / 4;
    w.total = (uint64_t*)ctx->cl_scalars.p;
}
}  // namespace

// One assignment of the resident objects to the uploaded view.  concurrent: launched on the cluster stream next to the frame
// kernel of the same frame; row-bound objects then re-derive their ViewVisibility from the frame's views (ctx->view_set)
// instead of reading the column that kernel is writing.  defer_fill: the fill is left for the next frame's kernel to carry.
int32_t cluster_assign_launch(mi_ctx* ctx, bool concurrent, uint64_t* out_total, bool defer_fill) {
    ClusterPrep p;
    int32_t rc = cluster_objects(ctx, concurrent, &p.o);
    if (rc) return rc;
    hipStream_t stream = ctx->stream;
    if (concurrent) {
        if (!ctx->cl_stream) {
            // HIP maps streams onto a small pool of hardware queues; a cluster stream that lands on the main stream's queue
            // would simply run behind the frame kernel.  pick_side_streams (ctx_exchange.cpp) probes candidates and returns
            // one that does not share it.
            bool shares = false;
            int32_t rcp = pick_side_streams(ctx, &ctx->cl_stream, 1, &shares);
            if (rcp) return rcp;
            HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_cl_done, hipEventDisableTiming));
            HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_cl_inputs, hipEventDisableTiming));
        }
        stream = ctx->cl_stream;
        if (ctx->cl_fill_pending) {  // a fill still waiting for a frame to ride in goes out on the main stream first
            if ((rc = cluster_fill_join(ctx))) return rc;
            ctx->cl_inputs_dirty = true;
        }
    } else {
        if ((rc = cluster_join(ctx))) return rc;  // behind whatever the cluster stream still runs (shared outputs)
        ctx->cl_inputs_dirty = true;               // ... and the cluster stream behind this, next time
    }
    bool fresh = false;
    if ((rc = cluster_buffers(ctx, &p, &fresh))) return rc;
    if (concurrent && (fresh || ctx->cl_inputs_dirty)) {  // uploads / allocations / earlier main-stream cluster work
        HIP_TRY(ctx, hipEventRecord(ctx->ev_cl_inputs, ctx->stream));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->cl_stream, ctx->ev_cl_inputs, 0));
        ctx->cl_inputs_dirty = false;
    }
    for (int attempt = 0; attempt < 2; ++attempt) {
        // the plane table lives in the staging arena, and the download between the attempts may have wrapped (or moved) it
        if ((rc = stage_cluster_planes(ctx))) return rc;
        cluster_next_set(ctx, &p);
        const ClusterWork& w = p.w;
        const bool defer = defer_fill && !concurrent && !out_total;
        HIP_TRY(ctx, launch_cluster_assign(ctx->cl_view, p.o, w, concurrent ? &ctx->view_set : nullptr, concurrent, !defer, stream, prof_mark, ctx,
                                           WalkPlanesHost{ctx->cl_planes_host.data(), (uint32_t)ctx->cl_planes_host.size()}));  // (the table as a kernel argument, kernels.h)
        if (defer) {  // the fill rides in the next frame's kernel (or cluster_fill_join launches it)
            ctx->cl_fill_job.w = w;
            ctx->cl_fill_job.n_clusters = p.C;
            ctx->cl_fill_job.n_objects = p.o.n;
            ctx->cl_fill_pending = true;
            break;
        }
        if (concurrent) {  // fire and forget: capacity is re-checked at download
            HIP_TRY(ctx, hipEventRecord(ctx->ev_cl_done, ctx->cl_stream));
            ctx->cl_on_side = true;
            break;
        }
        if (!out_total && attempt == 0) break;  // fire and forget: capacity is re-checked at download
        uint64_t total = 0;
        if ((rc = download(ctx, &total, w.total, 8))) return rc;
        if (out_total) *out_total = total;
        if (total <= w.capacity) break;
        // index list overflowed the device buffer: grow and redo (the reference's Vecs grow the same way)
        if ((rc = ensure(ctx, ctx->cl_indices, (size_t)total * 4 * 5 / 4))) return rc;
    }
    ctx->cl_assigned = true;
    return MI_OK;
}

// The walk of this frame's assignment as a job for the frame kernel's own launch (MI_CULL_WITH_CLUSTERS on a call that decides
// the frame's ViewVisibility alone): *job gets the view, the objects in derive mode, the set to walk into and the z chunk
// that fits the frame kernel's LDS; the fill is left pending.  *can_ride = false (nothing changed) when the grid is too wide
// for that LDS -- the caller then runs the assignment behind the frame kernel.  Call AFTER taking a pending fill for the
// same launch: this one's replaces it.
int32_t cluster_ride_prepare(mi_ctx* ctx, ClusterWalkJob* job, bool* can_ride) {
    *can_ride = false;
    if (!ctx->cl_have_view || !ctx->cl_rows_bound) return MI_OK;
    const ClusterViewDev& v = ctx->cl_view;
    const uint32_t dxy = v.dims[0] * v.dims[1], n_planes = v.dims[0] + v.dims[1] + v.dims[2] + 3u;
    uint32_t zc = 0;
    while (zc < v.dims[2] && cluster_walk_lds_bytes(dxy, zc + 1u, n_planes, true) <= FRAME_KERNEL_LDS_BYTES) ++zc;
    if (zc == 0) return MI_OK;
    // (spot lights: the riding walk runs the cone test against the clusters' bounding spheres too -- the frame kernel's WALK = 2
    // variant; without the sphere table the assignment of its own reports the missing table)
    if (ctx->cl_any_spot && !v.cluster_spheres) return MI_OK;
    ClusterPrep p;
    int32_t rc = cluster_objects(ctx, true, &p.o);
    if (rc) return rc;
    if ((rc = cluster_join(ctx))) return rc;
    bool fresh = false;
    if ((rc = cluster_buffers(ctx, &p, &fresh))) return rc;
    cluster_next_set(ctx, &p);
    // objects bound to a contiguous row range: the frame kernel's own row workgroups walk them (a block = a row tile); a row list
    // keeps the extra workgroups that re-derive the rows' visibility
    job->inrow = (!ctx->cl_rows_listed && ctx->walk_inrow_mode == 0 && p.o.n) ? 1u : 0u;
    job->tile0 = 0;
    job->n_blocks = p.w.n_blocks;
    if (job->inrow) {
        job->tile0 = ctx->cl_first_row / 256u;
        job->n_blocks = (uint32_t)(((uint64_t)ctx->cl_first_row + p.o.n + 255u) / 256u) - job->tile0;
        p.w.obj_delta = (int32_t)(job->tile0 * 256u) - (int32_t)ctx->cl_first_row;
    }
    job->spots = ctx->cl_any_spot ? 1u : 0u;
    job->view = v;
    job->objs = p.o;
    job->w = p.w;
    job->zc = zc;
    ctx->cl_fill_job.w = p.w;
    ctx->cl_fill_job.n_clusters = p.C;
    ctx->cl_fill_job.n_objects = p.o.n;
    ctx->cl_fill_pending = true;
    ctx->cl_inputs_dirty = true;
    ctx->cl_assigned = true;
    *can_ride = true;
    return MI_OK;
}

}  // namespace mi_detail

extern "C" {

int32_t mi_cluster_assign_resident(mi_ctx* ctx, uint64_t* out_total) {
    ENTER(ctx);
    return cluster_assign_launch(ctx, false, out_total);
}

int32_t mi_cluster_download(mi_ctx* ctx, uint32_t* out_offsets, uint32_t* out_indices, uint64_t capacity, uint32_t* out_counts,
                            uint64_t* out_total, float* out_farthest_z) {
    ENTER(ctx);
    if (!ctx->cl_assigned) return fail(ctx, MI_ERR_NOT_READY, "mi_cluster_download before mi_cluster_assign_resident");
    {
        int32_t rcj = cluster_join(ctx);
        if (rcj) return rcj;
    }
    const uint32_t C = ctx->cl_view.n_clusters;
    uint64_t total = 0;
    int32_t rc;
    if ((rc = download(ctx, &total, ctx->cl_scalars.p, 8))) return rc;
    if (total > ctx->cl_indices.bytes / 4) {
        // fire-and-forget assign overflowed: redo with a big enough buffer
        uint64_t t2 = 0;
        if ((rc = mi_cluster_assign_resident(ctx, &t2))) return rc;
        total = t2;
    }
    if (out_total) *out_total = total;
    const size_t off_totals = (6 * (size_t)C + 3) & ~(size_t)3, off_misc = off_totals + (((size_t)C + 3) & ~(size_t)3);
    const uint32_t* acc = (const uint32_t*)ctx->cl_acc.p + ctx->cl_parity * (off_misc + 4);  // the current set
    if (out_farthest_z && (rc = download(ctx, out_farthest_z, acc + off_misc, 4))) return rc;
    if (out_offsets && (rc = download(ctx, out_offsets, ctx->cl_offsets.p, ((size_t)C + 1) * 4))) return rc;
    if (out_counts && (rc = download(ctx, out_counts, acc, (size_t)C * 6 * 4))) return rc;
    if (out_indices) {
        if (total > capacity) return fail(ctx, MI_ERR_CAPACITY, "cluster index list has %llu entries, capacity %llu", (unsigned long long)total, (unsigned long long)capacity);
        if ((rc = download(ctx, out_indices, ctx->cl_indices.p, (size_t)total * 4))) return rc;
    }
    return MI_OK;
}

// test hook (include/bevy_mi355x_debug.h): the lists of the last fill that ran -- a pending one is NOT launched
int32_t mi_debug_cluster_download_unjoined(mi_ctx* ctx, uint32_t* out_offsets, uint32_t* out_indices, uint64_t capacity, uint64_t* out_total) {
    ENTER(ctx);
    if (!ctx->cl_assigned || !out_offsets) return fail(ctx, MI_ERR_NOT_READY, "mi_debug_cluster_download_unjoined: nothing assigned / out_offsets NULL");
    const uint32_t C = ctx->cl_view.n_clusters;
    int32_t rc;
    if ((rc = download(ctx, out_offsets, ctx->cl_offsets.p, ((size_t)C + 1) * 4))) return rc;
    const uint64_t total = out_offsets[C];
    if (out_total) *out_total = total;
    if (out_indices) {
        if (total > capacity || total > ctx->cl_indices.bytes / 4) return fail(ctx, MI_ERR_CAPACITY, "cluster index list has %llu entries", (unsigned long long)total);
        if ((rc = download(ctx, out_indices, ctx->cl_indices.p, (size_t)total * 4))) return rc;
    }
    return MI_OK;
}

int32_t mi_cluster_download_bindings(mi_ctx* ctx, const uint32_t* remap, uint32_t n_remap, uint32_t* out_offsets_and_counts,
                                     uint32_t* out_index_list, uint64_t capacity, uint64_t* out_total) {
    ENTER(ctx);
    if (!ctx->cl_assigned) return fail(ctx, MI_ERR_NOT_READY, "mi_cluster_download_bindings before mi_cluster_assign_resident");
    {
        int32_t rcj = cluster_join(ctx);
        if (rcj) return rcj;
    }
    const uint32_t C = ctx->cl_view.n_clusters;
    uint64_t total = 0;
    int32_t rc;
    if ((rc = download(ctx, &total, ctx->cl_scalars.p, 8))) return rc;
    if (total > ctx->cl_indices.bytes / 4) {  // fire-and-forget assign overflowed: redo with a big enough buffer
        uint64_t t2 = 0;
        if ((rc = mi_cluster_assign_resident(ctx, &t2))) return rc;
        total = t2;
    }
    if (out_total) *out_total = total;
    const size_t off_totals = (6 * (size_t)C + 3) & ~(size_t)3, off_misc = off_totals + (((size_t)C + 3) & ~(size_t)3);
    const uint32_t* acc = (const uint32_t*)ctx->cl_acc.p + ctx->cl_parity * (off_misc + 4);  // the current set
    if ((rc = ensure(ctx, ctx->cl_bind_oc, (size_t)C * 32))) return rc;
    if ((rc = ensure(ctx, ctx->cl_bind_idx, std::max<size_t>(total, 1) * 4))) return rc;
    const uint32_t* d_remap = nullptr;
    if (remap) {
        if ((rc = ensure(ctx, ctx->cl_remap, std::max<size_t>(n_remap, 1) * 4))) return rc;
        if ((rc = upload(ctx, ctx->cl_remap.p, remap, (size_t)n_remap * 4))) return rc;
        d_remap = (const uint32_t*)ctx->cl_remap.p;
    }
    HIP_TRY(ctx, launch_cluster_bindings(C, (const uint32_t*)ctx->cl_offsets.p, acc, (const uint32_t*)ctx->cl_indices.p, d_remap, n_remap,
                                         total, (uint32_t*)ctx->cl_bind_oc.p, (uint32_t*)ctx->cl_bind_idx.p, ctx->stream));
    if (out_offsets_and_counts && (rc = download(ctx, out_offsets_and_counts, ctx->cl_bind_oc.p, (size_t)C * 32))) return rc;
    if (out_index_list) {
        if (total > capacity) return fail(ctx, MI_ERR_CAPACITY, "cluster index list has %llu entries, capacity %llu", (unsigned long long)total, (unsigned long long)capacity);
        if ((rc = download(ctx, out_index_list, ctx->cl_bind_idx.p, (size_t)total * 4))) return rc;
    }
    return MI_OK;
}

int32_t mi_cluster_assign_frame(mi_ctx* ctx, const mi_cluster_config* config, mi_cluster_history* history,
                                const float camera_affine[12], const float clip_from_view[16], const float frustum[24],
                                uint32_t screen_w, uint32_t screen_h, uint32_t view_layer_mask, uint64_t max_indices,
                                mi_cluster_view* out_view, uint32_t* out_active) {
    ENTER(ctx);
    if (!config || !history || !camera_affine || !clip_from_view || !frustum) return fail(ctx, MI_ERR_INVALID_ARG, "mi_cluster_assign_frame: NULL");
    mi_cluster_resolved res{};
    int32_t rc = mi_cluster_config_resolve(config, history, screen_w, screen_h, max_indices, &res);
    if (rc) return fail(ctx, rc, "mi_cluster_assign_frame: invalid ClusterConfig");
    if (out_active) *out_active = res.active;
    if (!res.active) {  // Clusters::clear(): dimensions = 0, nothing assigned; the statistics are left as they are
        ctx->cl_have_view = false;
        ctx->cl_assigned = false;
        return MI_OK;
    }
    uint32_t tile[2], dims[3];
    if ((rc = mi_cluster_view_dims(screen_w, screen_h, res.requested_dims, tile, dims))) return fail(ctx, rc, "mi_cluster_assign_frame: bad grid");
    ctx->cl_host_planes.assign((size_t)(dims[0] + dims[1] + dims[2] + 3) * 4, 0.0f);
    ctx->cl_host_spheres.assign(ctx->cl_any_spot ? (size_t)dims[0] * dims[1] * dims[2] * 4 : 0, 0.0f);
    mi_cluster_view view{};
    if ((rc = mi_cluster_view_build(camera_affine, clip_from_view, frustum, screen_w, screen_h, res.requested_dims, res.first_slice_depth,
                                    res.far_z, view_layer_mask, ctx->cl_host_planes.data(),
                                    ctx->cl_any_spot ? ctx->cl_host_spheres.data() : nullptr, &view)))
        return fail(ctx, rc, "mi_cluster_assign_frame: mi_cluster_view_build failed");
    if ((rc = mi_cluster_upload_view(ctx, &view))) return rc;
    uint64_t total = 0;
    if ((rc = mi_cluster_assign_resident(ctx, &total))) return rc;
    float farthest = 0.0f;
    if ((rc = mi_cluster_download(ctx, nullptr, nullptr, 0, nullptr, nullptr, &farthest))) return rc;
    // assign.rs:810-811
    history->has_total_cluster_index_count = 1;
    history->total_cluster_index_count = total;
    history->has_farthest_z = 1;
    history->farthest_z = farthest;
    if (out_view) *out_view = view;
    return MI_OK;
}

int32_t mi_cluster_assign(mi_ctx* ctx, const mi_cluster_view* view, uint32_t n_objects, const float* pos_range,
                          const uint8_t* obj_type, const uint32_t* layer_mask, const float* spot_dir, const float* spot_sin_cos,
                          uint32_t* out_offsets, uint32_t* out_indices, uint64_t capacity, uint32_t* out_counts,
                          uint64_t* out_total, float* out_farthest_z) {
    int32_t rc;
    if ((rc = mi_cluster_upload_objects(ctx, n_objects, pos_range, obj_type, layer_mask, spot_dir, spot_sin_cos))) return rc;
    if ((rc = mi_cluster_upload_view(ctx, view))) return rc;
    uint64_t total = 0;
    if ((rc = mi_cluster_assign_resident(ctx, &total))) return rc;
    return mi_cluster_download(ctx, out_offsets, out_indices, capacity, out_counts, out_total, out_farthest_z);
}

}  // extern "C"
