// ctx_cluster.cpp -- clustered-forward light assignment entry points.
#include "ctx.h"

using namespace mi;
using namespace mi_detail;

extern "C" {

// =============================================================================================
// clustering
// =============================================================================================
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
    if ((rc = ensure(ctx, ctx->cl_pos, (size_t)n * 16))) return rc;
    if ((rc = upload(ctx, ctx->cl_pos.p, pos_range, (size_t)n * 16))) return rc;
    ctx->cl_have_type = obj_type != nullptr;
    if (obj_type) {
        if ((rc = ensure(ctx, ctx->cl_type, n))) return rc;
        if ((rc = upload(ctx, ctx->cl_type.p, obj_type, n))) return rc;
    }
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
    if (n != ctx->cl_n) ctx->cl_rows_bound = false;  // a different object set: the row binding has to be renewed
    ctx->cl_n = n;
    ctx->cl_assigned = false;
    return MI_OK;
}

int32_t mi_cluster_bind_objects_to_rows(mi_ctx* ctx, uint32_t first_row, uint32_t n_objects) {
    ENTER(ctx);
    if (n_objects == 0) {
        ctx->cl_rows_bound = false;
        return MI_OK;
    }
    if (n_objects != ctx->cl_n)
        return fail(ctx, MI_ERR_INVALID_ARG, "mi_cluster_bind_objects_to_rows: %u objects, %u uploaded", n_objects, ctx->cl_n);
    int32_t rc = check_rows(ctx, first_row, n_objects, "mi_cluster_bind_objects_to_rows");
    if (rc) return rc;
    ctx->cl_rows_bound = true;
    ctx->cl_first_row = first_row;
    ctx->cl_assigned = false;
    return MI_OK;
}

int32_t mi_cluster_upload_view(mi_ctx* ctx, const mi_cluster_view* view) {
    ENTER(ctx);
    if (!view || !view->x_planes || !view->y_planes || !view->z_planes) return fail(ctx, MI_ERR_INVALID_ARG, "mi_cluster_upload_view: NULL");
    const uint64_t C = (uint64_t)view->dims[0] * view->dims[1] * view->dims[2];
    if (C == 0 || C > 4096) return fail(ctx, MI_ERR_INVALID_ARG, "cluster count %llu outside 1..4096 (assign.rs:410-413)", (unsigned long long)C);
    const uint32_t nx = view->dims[0] + 1, ny = view->dims[1] + 1, nz = view->dims[2] + 1;
    int32_t rc;
    const void* planes_before = ctx->cl_planes.p;
    if ((rc = ensure(ctx, ctx->cl_planes, (size_t)(nx + ny + nz) * 16))) return rc;
    if (ctx->cl_planes.p != planes_before) ctx->cl_planes_sent.clear();  // a new buffer holds nothing yet
    float* base = (float*)ctx->cl_planes.p;
    // The cluster planes live in VIEW space: they depend on the projection, the grid and near / far, not on where the
    // camera is -- a moving camera re-sends only the matrices and the frustum, which travel in the kernarg segment.
    {
        std::vector<float> planes((size_t)(nx + ny + nz) * 4);
        memcpy(planes.data(), view->x_planes, (size_t)nx * 16);
        memcpy(planes.data() + 4 * (size_t)nx, view->y_planes, (size_t)ny * 16);
        memcpy(planes.data() + 4 * (size_t)(nx + ny), view->z_planes, (size_t)nz * 16);
        if (planes.size() != ctx->cl_planes_sent.size() || memcmp(planes.data(), ctx->cl_planes_sent.data(), planes.size() * 4) != 0) {
            if ((rc = upload(ctx, base, planes.data(), planes.size() * 4))) return rc;
            ctx->cl_planes_sent.swap(planes);
        }
    }
    ClusterViewDev& d = ctx->cl_view;
    memcpy(d.dims, view->dims, sizeof d.dims);
    d.is_orthographic = view->is_orthographic;
    d.view_layer_mask = view->view_layer_mask;
    d.n_clusters = (uint32_t)C;
    memcpy(d.cluster_factors, view->cluster_factors, sizeof d.cluster_factors);
    memcpy(d.view_from_world, view->view_from_world, sizeof d.view_from_world);
    memcpy(d.clip_from_view, view->clip_from_view, sizeof d.clip_from_view);
    memcpy(d.view_from_world_scale, view->view_from_world_scale, sizeof d.view_from_world_scale);
    d.view_from_world_scale_max = view->view_from_world_scale_max;
    memcpy(d.frustum, view->frustum, sizeof d.frustum);
    d.x_planes = base;
    d.y_planes = base + 4 * (size_t)nx;
    d.z_planes = base + 4 * (size_t)(nx + ny);
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

int32_t mi_cluster_assign_resident(mi_ctx* ctx, uint64_t* out_total) {
    ENTER(ctx);
    if (!ctx->cl_have_view) return fail(ctx, MI_ERR_NOT_READY, "mi_cluster_assign_resident: no view uploaded");
    if (ctx->cl_any_spot && !ctx->cl_view.cluster_spheres)
        return fail(ctx, MI_ERR_INVALID_ARG, "spot lights present but mi_cluster_view.cluster_spheres is NULL");
    const uint32_t C = ctx->cl_view.n_clusters;
    ClusterObjects o{};
    o.n = ctx->cl_n;
    o.pos_range = (const float*)ctx->cl_pos.p;
    o.obj_type = ctx->cl_have_type ? (const uint8_t*)ctx->cl_type.p : nullptr;
    o.layer_mask = ctx->cl_have_layers ? (const uint32_t*)ctx->cl_layers.p : nullptr;
    o.spot_dir = ctx->cl_have_spot_dir ? (const float*)ctx->cl_dir.p : nullptr;
    if (ctx->cl_any_spot && !o.spot_dir && !ctx->cl_rows_bound)
        return fail(ctx, MI_ERR_INVALID_ARG, "spot lights need spot_dir unless the objects are bound to rows (mi_cluster_bind_objects_to_rows)");
    o.spot_sin_cos = ctx->cl_have_spot ? (const float*)ctx->cl_sincos.p : nullptr;
    if (ctx->cl_rows_bound) {
        if ((uint64_t)ctx->cl_first_row + o.n > ctx->n)
            return fail(ctx, MI_ERR_NOT_READY, "cluster objects are bound to rows [%u,%u) but the context has %u rows", ctx->cl_first_row,
                        ctx->cl_first_row + o.n, ctx->n);
        o.row_global = ctx->g;
        o.row_vv = ctx->vv;
        o.first_row = ctx->cl_first_row;
    }
    ClusterWork w{};
    w.n_blocks = std::max(1u, (o.n + CLUSTER_BLOCK - 1) / CLUSTER_BLOCK);
    int32_t rc;
    const size_t off_totals = (6 * (size_t)C + 3) & ~(size_t)3, off_misc = off_totals + (((size_t)C + 3) & ~(size_t)3);
    const size_t acc_words = off_misc + 4;  // per parity; 16-byte aligned sections: counts | totals | misc
    w.row_stride = (w.n_blocks + 7u) & ~7u;
    const size_t mat_bytes = (size_t)C * w.row_stride * 2;  // per parity
    if ((rc = ensure(ctx, ctx->cl_pair_cb, (size_t)w.n_blocks * C * 4))) return rc;
    if ((rc = ensure(ctx, ctx->cl_pair_mask, (size_t)w.n_blocks * C * 32))) return rc;
    if ((rc = ensure(ctx, ctx->cl_offsets, ((size_t)C + 1) * 4))) return rc;
    if ((rc = ensure(ctx, ctx->cl_scalars, 16))) return rc;
    if (!ctx->cl_acc.p || ctx->cl_acc_clusters != C || ctx->cl_acc_blocks != w.n_blocks) {
        // (re)shaped accumulators / count matrix start zeroed in both parities; afterwards the fill kernel keeps
        // the idle parity zeroed
        if ((rc = ensure(ctx, ctx->cl_acc, 2 * acc_words * 4))) return rc;
        if ((rc = ensure(ctx, ctx->cl_block_counts, 2 * mat_bytes))) return rc;
        HIP_TRY(ctx, hipMemsetAsync(ctx->cl_acc.p, 0, 2 * acc_words * 4, ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(ctx->cl_block_counts.p, 0, 2 * mat_bytes, ctx->stream));
        ctx->cl_acc_clusters = C;
        ctx->cl_acc_blocks = w.n_blocks;
    }
    if (!ctx->cl_indices.p && (rc = ensure(ctx, ctx->cl_indices, (size_t)1 << 20))) return rc;
    for (int attempt = 0; attempt < 2; ++attempt) {
        ctx->cl_parity ^= 1u;
        const uint32_t par = ctx->cl_parity;
        uint32_t* acc = (uint32_t*)ctx->cl_acc.p + par * acc_words;
        w.block_counts = (uint16_t*)((char*)ctx->cl_block_counts.p + par * mat_bytes);
        w.block_counts_next = (uint16_t*)((char*)ctx->cl_block_counts.p + (par ^ 1u) * mat_bytes);
        w.counts = acc;
        w.totals = acc + off_totals;
        w.farthest_z = (float*)(acc + off_misc);
        w.pair_total = acc + off_misc + 1;
        w.acc_words = (uint32_t)acc_words;
        w.acc_next = (uint32_t*)ctx->cl_acc.p + (par ^ 1u) * acc_words;
        w.pair_cb = (uint32_t*)ctx->cl_pair_cb.p;
        w.pair_mask = (uint32_t*)ctx->cl_pair_mask.p;
        w.offsets = (uint32_t*)ctx->cl_offsets.p;
        w.indices = (uint32_t*)ctx->cl_indices.p;
        w.capacity = ctx->cl_indices.bytes / 4;
        w.total = (uint64_t*)ctx->cl_scalars.p;
        HIP_TRY(ctx, launch_cluster_assign(ctx->cl_view, o, w, ctx->stream, prof_mark, ctx));
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

int32_t mi_cluster_download(mi_ctx* ctx, uint32_t* out_offsets, uint32_t* out_indices, uint64_t capacity, uint32_t* out_counts,
                            uint64_t* out_total, float* out_farthest_z) {
    ENTER(ctx);
    if (!ctx->cl_assigned) return fail(ctx, MI_ERR_NOT_READY, "mi_cluster_download before mi_cluster_assign_resident");
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
    const uint32_t* acc = (const uint32_t*)ctx->cl_acc.p + ctx->cl_parity * (off_misc + 4);
    if (out_farthest_z && (rc = download(ctx, out_farthest_z, acc + off_misc, 4))) return rc;
    if (out_offsets && (rc = download(ctx, out_offsets, ctx->cl_offsets.p, ((size_t)C + 1) * 4))) return rc;
    if (out_counts && (rc = download(ctx, out_counts, acc, (size_t)C * 6 * 4))) return rc;
    if (out_indices) {
        if (total > capacity) return fail(ctx, MI_ERR_CAPACITY, "cluster index list has %llu entries, capacity %llu", (unsigned long long)total, (unsigned long long)capacity);
        if ((rc = download(ctx, out_indices, ctx->cl_indices.p, (size_t)total * 4))) return rc;
    }
    return MI_OK;
}

int32_t mi_cluster_download_bindings(mi_ctx* ctx, const uint32_t* remap, uint32_t n_remap, uint32_t* out_offsets_and_counts,
                                     uint32_t* out_index_list, uint64_t capacity, uint64_t* out_total) {
    ENTER(ctx);
    if (!ctx->cl_assigned) return fail(ctx, MI_ERR_NOT_READY, "mi_cluster_download_bindings before mi_cluster_assign_resident");
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
    const uint32_t* acc = (const uint32_t*)ctx->cl_acc.p + ctx->cl_parity * (off_misc + 4);
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
