// ctx_batch.cpp -- the batching work-item build entry points.
#include "ctx.h"

using namespace mi;
using namespace mi_detail;

extern "C" {

// =============================================================================================
// batching work-item build
// =============================================================================================
int32_t mi_batch_upload_rows(mi_ctx* ctx, uint32_t first_row, uint32_t n, const uint32_t* batch_set, const uint32_t* bin_index,
                             const uint32_t* input_uniform_index) {
    ENTER(ctx);
    if (n && (!batch_set || !bin_index || !input_uniform_index)) return fail(ctx, MI_ERR_INVALID_ARG, "mi_batch_upload_rows: NULL column");
    int32_t rc = check_rows(ctx, first_row, n, "mi_batch_upload_rows");
    if (rc) return rc;
    if (n == 0) return MI_OK;
    if ((rc = upload(ctx, ctx->bt_set + first_row, batch_set, (size_t)n * 4))) return rc;
    if ((rc = upload(ctx, ctx->bt_bin + first_row, bin_index, (size_t)n * 4))) return rc;
    if ((rc = upload(ctx, ctx->bt_input + first_row, input_uniform_index, (size_t)n * 4))) return rc;
    ctx->bt_have_rows = true;
    ctx->bt_resolve = true;
    return MI_OK;
}

int32_t mi_batch_upload_sets(mi_ctx* ctx, uint32_t n_sets, const uint8_t* set_indexed, const uint32_t* bin_table_offset,
                             const uint32_t* bin_index_to_bin_metadata_index, const uint32_t* meta_offset,
                             const mi_bin_metadata* bin_metadata) {
    ENTER(ctx);
    if (n_sets > 65536u) return fail(ctx, MI_ERR_INVALID_ARG, "mi_batch_upload_sets: %u batch sets, at most 65536", n_sets);
    if (n_sets && (!set_indexed || !bin_table_offset || !bin_index_to_bin_metadata_index || !meta_offset || !bin_metadata))
        return fail(ctx, MI_ERR_INVALID_ARG, "mi_batch_upload_sets: NULL table");
    static const uint32_t zero_offsets[1] = {0};
    if (n_sets == 0) bin_table_offset = meta_offset = zero_offsets;
    for (uint32_t s = 0; s < n_sets; ++s) {
        if (bin_table_offset[s + 1] < bin_table_offset[s] || meta_offset[s + 1] < meta_offset[s])
            return fail(ctx, MI_ERR_INVALID_ARG, "mi_batch_upload_sets: offsets of set %u decrease", s);
        const uint32_t bins = meta_offset[s + 1] - meta_offset[s];
        for (uint32_t k = 0; k < bins; ++k)
            if (bin_metadata[meta_offset[s] + k].indirect_parameters_offset >= bins)
                return fail(ctx, MI_ERR_INVALID_ARG, "mi_batch_upload_sets: set %u bin %u: indirect_parameters_offset %u >= %u bins", s, k,
                            bin_metadata[meta_offset[s] + k].indirect_parameters_offset, bins);
    }
    if (bin_table_offset[0] != 0 || meta_offset[0] != 0) return fail(ctx, MI_ERR_INVALID_ARG, "mi_batch_upload_sets: offsets must start at 0");
    const uint32_t n_table = bin_table_offset[n_sets], n_meta = meta_offset[n_sets];
    int32_t rc;
    if ((rc = ensure(ctx, ctx->bt_set_indexed, std::max<size_t>(n_sets, 1)))) return rc;
    if ((rc = ensure(ctx, ctx->bt_table_off, ((size_t)n_sets + 1) * 4))) return rc;
    if ((rc = ensure(ctx, ctx->bt_meta_off, ((size_t)n_sets + 1) * 4))) return rc;
    if ((rc = ensure(ctx, ctx->bt_table, std::max<size_t>(n_table, 1) * 4))) return rc;
    if ((rc = ensure(ctx, ctx->bt_meta, std::max<size_t>(n_meta, 1) * 12))) return rc;
    if (n_sets && (rc = upload(ctx, ctx->bt_set_indexed.p, set_indexed, n_sets))) return rc;
    if ((rc = upload(ctx, ctx->bt_table_off.p, bin_table_offset, ((size_t)n_sets + 1) * 4))) return rc;
    if ((rc = upload(ctx, ctx->bt_meta_off.p, meta_offset, ((size_t)n_sets + 1) * 4))) return rc;
    if (n_table && (rc = upload(ctx, ctx->bt_table.p, bin_index_to_bin_metadata_index, (size_t)n_table * 4))) return rc;
    if (n_meta && (rc = upload(ctx, ctx->bt_meta.p, bin_metadata, (size_t)n_meta * 12))) return rc;
    ctx->bt_n_sets = n_sets;
    ctx->bt_n_meta = n_meta;
    ctx->bt_have_sets = true;
    ctx->bt_resolve = true;
    ctx->bt_built = false;
    return MI_OK;
}

int32_t mi_batch_build(mi_ctx* ctx, uint32_t view, uint32_t class_bit, const mi_batch_initial* initial) {
    ENTER(ctx);
    if (!ctx->culled) return fail(ctx, MI_ERR_NOT_READY, "mi_batch_build before mi_cull");
    {
        int32_t rcj = compaction_join(ctx);  // asynchronous compaction: the list this build reads must be complete
        if (rcj) return rcj;
    }
    if (!ctx->bt_have_sets || !ctx->bt_have_rows) return fail(ctx, MI_ERR_NOT_READY, "mi_batch_build before mi_batch_upload_rows / mi_batch_upload_sets");
    if (view >= ctx->compact_views) return fail(ctx, MI_ERR_INVALID_ARG, "mi_batch_build: view %u of %u", view, ctx->compact_views);
    static_assert(sizeof(mi_batch_initial) == sizeof(BatchInitial), "mi_batch_initial layout");
    BatchArgs a{};
    if (initial) memcpy(&a.initial, initial, sizeof a.initial);
    uint32_t slot = 0xFFFFFFFFu;
    for (uint32_t k = 0; k < ctx->compact_classes; ++k)
        if (ctx->class_bits[k] == class_bit) slot = k;
    int32_t rc;
    const uint32_t n_sets = ctx->bt_n_sets;
    // capacities: every row of the list could be a work item of either class; every bin gets a metadata entry
    const uint32_t n_tiles = std::max<uint32_t>(1u, (ctx->n + BATCH_TILE - 1u) / BATCH_TILE);
    const size_t cap_rows = (size_t)n_tiles * BATCH_TILE;
    if ((rc = ensure(ctx, ctx->bt_rows_a, cap_rows * 4))) return rc;
    if (n_sets > 256u && (rc = ensure(ctx, ctx->bt_rows_b, cap_rows * 4))) return rc;
    if ((rc = ensure(ctx, ctx->bt_hist, (size_t)256 * n_tiles * 4))) return rc;
    if ((rc = ensure(ctx, ctx->bt_set_count, std::max<size_t>(n_sets, 1) * 2 * 4))) return rc;
    if ((rc = ensure(ctx, ctx->bt_set_scan, std::max<size_t>(n_sets, 1) * 5 * 4))) return rc;
    if ((rc = ensure(ctx, ctx->bt_counters, 64))) return rc;
    if ((rc = ensure(ctx, ctx->bt_records, std::max<size_t>(n_sets, 1) * 32))) return rc;
    if ((rc = ensure(ctx, ctx->bt_totals, 32))) return rc;
    for (int c = 0; c < 2; ++c) {
        const size_t wi = ((size_t)a.initial.work_item_index[c] + ctx->n + 1) * 8;
        const size_t md = ((size_t)a.initial.indirect_parameters_index[c] + ctx->bt_n_meta + 1) * 20;
        const size_t bs = ((size_t)a.initial.batch_set_index[c] + n_sets + 1) * 8;
        if ((rc = ensure(ctx, ctx->bt_wi[c], wi))) return rc;
        if ((rc = ensure(ctx, ctx->bt_md[c], md))) return rc;
        if ((rc = ensure(ctx, ctx->bt_bs[c], bs))) return rc;
        // entries below `initial` belong to the CPU-built part of the phase: they read as zeros here
        if (a.initial.work_item_index[c]) HIP_TRY(ctx, hipMemsetAsync(ctx->bt_wi[c].p, 0, (size_t)a.initial.work_item_index[c] * 8, ctx->stream));
        if (a.initial.indirect_parameters_index[c])
            HIP_TRY(ctx, hipMemsetAsync(ctx->bt_md[c].p, 0, (size_t)a.initial.indirect_parameters_index[c] * 20, ctx->stream));
        if (a.initial.batch_set_index[c]) HIP_TRY(ctx, hipMemsetAsync(ctx->bt_bs[c].p, 0, (size_t)a.initial.batch_set_index[c] * 8, ctx->stream));
        a.work_items[c] = (uint32_t*)ctx->bt_wi[c].p;
        a.metadata[c] = (uint32_t*)ctx->bt_md[c].p;
        a.batch_sets[c] = (uint32_t*)ctx->bt_bs[c].p;
    }
    if (slot == 0xFFFFFFFFu || ctx->n == 0) {
        // no row carries this class: VisibleEntities::get() is empty -> nothing is appended
        mi_batch_totals t{};
        for (int c = 0; c < 2; ++c) {
            t.work_item_len[c] = a.initial.work_item_index[c];
            t.indirect_parameters_len[c] = a.initial.indirect_parameters_index[c];
            t.batch_set_len[c] = a.initial.batch_set_index[c];
        }
        t.data_buffer_len = a.initial.output_mesh_uniform_index;
        if ((rc = upload(ctx, ctx->bt_totals.p, &t, sizeof t))) return rc;
        if (ctx->bt_n_meta) {  // instance counts of an empty build are all zero
            std::vector<mi_bin_metadata> m(ctx->bt_n_meta);
            if ((rc = download(ctx, m.data(), ctx->bt_meta.p, m.size() * 12))) return rc;
            for (auto& e : m) e.instance_count = 0;
            if ((rc = upload(ctx, ctx->bt_meta.p, m.data(), m.size() * 12))) return rc;
        }
        ctx->bt_built = true;
        return MI_OK;
    }
    const uint32_t seg = view * ctx->compact_classes + slot;
    a.list_count = (const uint32_t*)ctx->fb[ctx->cur].seg_totals.p + seg;
    if (ctx->compact_fast) {
        a.list = (const uint32_t*)ctx->fb[ctx->cur].out_rows.p + (size_t)seg * ctx->seg_stride;
        a.list_base = nullptr;
    } else {
        a.list = (const uint32_t*)ctx->fb[ctx->cur].out_rows.p;
        a.list_base = (const uint64_t*)ctx->seg_bases.p + seg;
    }
    a.row_set = ctx->bt_set;
    a.row_bin = ctx->bt_bin;
    a.row_input = ctx->bt_input;
    a.row_meta = ctx->bt_row_meta;
    if (ctx->bt_resolve) {
        HIP_TRY(ctx, launch_batch_resolve_rows(ctx->n, n_sets, ctx->bt_set, ctx->bt_bin, (const uint32_t*)ctx->bt_table_off.p,
                                               (const uint32_t*)ctx->bt_table.p, (const uint32_t*)ctx->bt_meta_off.p, ctx->bt_row_meta,
                                               ctx->stream));
        ctx->bt_resolve = false;
    }
    a.n_sets = n_sets;
    a.n_meta = ctx->bt_n_meta;
    a.set_indexed = (const uint8_t*)ctx->bt_set_indexed.p;
    a.bin_table_offset = (const uint32_t*)ctx->bt_table_off.p;
    a.bin_table = (const uint32_t*)ctx->bt_table.p;
    a.meta_offset = (const uint32_t*)ctx->bt_meta_off.p;
    a.bin_metadata = (uint32_t*)ctx->bt_meta.p;
    a.rows_a = (uint32_t*)ctx->bt_rows_a.p;
    a.rows_b = (uint32_t*)ctx->bt_rows_b.p;
    a.tile_hist = (uint32_t*)ctx->bt_hist.p;
    a.n_tiles = n_tiles;
    a.set_count = (uint32_t*)ctx->bt_set_count.p;
    a.set_scan = (uint32_t*)ctx->bt_set_scan.p;
    a.counters = (uint32_t*)ctx->bt_counters.p;
    a.records = (uint32_t*)ctx->bt_records.p;
    a.totals = (uint32_t*)ctx->bt_totals.p;
    HIP_TRY(ctx, launch_batch_build(a, ctx->stream, prof_mark, ctx));
    ctx->bt_built = true;
    return MI_OK;
}

int32_t mi_batch_download_totals(mi_ctx* ctx, mi_batch_totals* out) {
    ENTER(ctx);
    if (!out) return fail(ctx, MI_ERR_INVALID_ARG, "mi_batch_download_totals: NULL");
    if (!ctx->bt_built) return fail(ctx, MI_ERR_NOT_READY, "mi_batch_download_totals before mi_batch_build");
    return download(ctx, out, ctx->bt_totals.p, sizeof *out);
}

int32_t mi_batch_download(mi_ctx* ctx, uint32_t what, uint32_t mesh_class, void* out, uint32_t capacity_elems, uint32_t* out_count) {
    ENTER(ctx);
    if (!out_count) return fail(ctx, MI_ERR_INVALID_ARG, "mi_batch_download: NULL out_count");
    if (!ctx->bt_built) return fail(ctx, MI_ERR_NOT_READY, "mi_batch_download before mi_batch_build");
    if (mesh_class > 1u && what <= MI_BATCH_SETS) return fail(ctx, MI_ERR_INVALID_ARG, "mi_batch_download: mesh class %u", mesh_class);
    mi_batch_totals t{};
    int32_t rc = download(ctx, &t, ctx->bt_totals.p, sizeof t);
    if (rc) return rc;
    const void* src = nullptr;
    uint32_t count = 0, elem = 0;
    switch (what) {
    case MI_BATCH_WORK_ITEMS: src = ctx->bt_wi[mesh_class].p; count = t.work_item_len[mesh_class]; elem = 8; break;
    case MI_BATCH_INDIRECT_PARAMETERS_METADATA: src = ctx->bt_md[mesh_class].p; count = t.indirect_parameters_len[mesh_class]; elem = 20; break;
    case MI_BATCH_SETS: src = ctx->bt_bs[mesh_class].p; count = t.batch_set_len[mesh_class]; elem = 8; break;
    case MI_BATCH_RECORDS: src = ctx->bt_records.p; count = t.n_records; elem = 32; break;
    case MI_BATCH_BIN_METADATA: src = ctx->bt_meta.p; count = ctx->bt_n_meta; elem = 12; break;
    default: return fail(ctx, MI_ERR_INVALID_ARG, "mi_batch_download: unknown array %u", what);
    }
    *out_count = count;
    if (count > capacity_elems) return fail(ctx, MI_ERR_CAPACITY, "mi_batch_download: %u elements, capacity %u", count, capacity_elems);
    if (count && !out) return fail(ctx, MI_ERR_INVALID_ARG, "mi_batch_download: NULL out");
    if (count) return download(ctx, out, src, (size_t)count * elem);
    return MI_OK;
}

}  // extern "C"
