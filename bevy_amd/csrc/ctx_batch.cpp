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
    if ((uint64_t)n_sets + ctx->bt_bat_indexed.size() + 2u * ctx->bt_unb_indexed.size() > 65536u)
        return fail(ctx, MI_ERR_INVALID_ARG, "mi_batch_upload_sets: %u sets + %zu batchable + 2 x %zu unbatchable bins exceed 65536 buckets", n_sets,
                    ctx->bt_bat_indexed.size(), ctx->bt_unb_indexed.size());
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
    if ((rc = ensure(ctx, ctx->bt_meta_out, std::max<size_t>(n_meta, 1) * 12))) return rc;
    for (int k = 0; k < 2; ++k) {
        if ((rc = ensure(ctx, ctx->bt_inst[k], std::max<size_t>(n_meta, 1) * 4 * BATCH_INST_STRIDE))) return rc;
        HIP_TRY(ctx, hipMemsetAsync(ctx->bt_inst[k].p, 0, std::max<size_t>(n_meta, 1) * 4 * BATCH_INST_STRIDE, ctx->stream));  // builds keep them zero from here on
    }
    if (n_sets && (rc = upload(ctx, ctx->bt_set_indexed.p, set_indexed, n_sets))) return rc;
    if ((rc = upload(ctx, ctx->bt_table_off.p, bin_table_offset, ((size_t)n_sets + 1) * 4))) return rc;
    if ((rc = upload(ctx, ctx->bt_meta_off.p, meta_offset, ((size_t)n_sets + 1) * 4))) return rc;
    if (n_table && (rc = upload(ctx, ctx->bt_table.p, bin_index_to_bin_metadata_index, (size_t)n_table * 4))) return rc;
    if (n_meta && (rc = upload(ctx, ctx->bt_meta.p, bin_metadata, (size_t)n_meta * 12))) return rc;
    // what MI_BATCH_BIN_METADATA reads when a build does not visit the sets: the table with zero counts
    ctx->bt_meta_zero.assign((const uint32_t*)bin_metadata, (const uint32_t*)bin_metadata + 3u * (size_t)n_meta);
    for (uint32_t m = 0; m < n_meta; ++m) ctx->bt_meta_zero[3u * m + 2u] = 0u;
    if (n_meta && (rc = upload(ctx, ctx->bt_meta_out.p, ctx->bt_meta_zero.data(), (size_t)n_meta * 12))) return rc;
    ctx->bt_set_indexed_host.assign(set_indexed, set_indexed + n_sets);
    ctx->bt_n_sets = n_sets;
    ctx->bt_n_meta = n_meta;
    ctx->bt_have_sets = true;
    ctx->bt_resolve = true;
    ctx->bt_desc_dirty = true;
    ctx->bt_built = false;
    return MI_OK;
}

int32_t mi_batch_upload_row_bins(mi_ctx* ctx, uint32_t first_row, uint32_t n, const uint8_t* kind, const uint32_t* bin) {
    ENTER(ctx);
    if (n && (!kind || !bin)) return fail(ctx, MI_ERR_INVALID_ARG, "mi_batch_upload_row_bins: NULL column");
    int32_t rc = check_rows(ctx, first_row, n, "mi_batch_upload_row_bins");
    if (rc) return rc;
    if (n == 0) return MI_OK;
    for (uint32_t i = 0; i < n; ++i)
        if (kind[i] > MI_BATCH_ROW_NONE) return fail(ctx, MI_ERR_INVALID_ARG, "mi_batch_upload_row_bins: row %u has kind %u", first_row + i, kind[i]);
    if ((rc = upload(ctx, ctx->bt_kind + first_row, kind, n))) return rc;
    if ((rc = upload(ctx, ctx->bt_cpu_bin + first_row, bin, (size_t)n * 4))) return rc;
    ctx->bt_resolve = true;
    return MI_OK;
}

int32_t mi_batch_upload_bins(mi_ctx* ctx, uint32_t n_unbatchable_bins, const uint8_t* unbatchable_indexed, uint32_t n_batchable_bins,
                             const uint8_t* batchable_indexed) {
    ENTER(ctx);
    if ((n_unbatchable_bins && !unbatchable_indexed) || (n_batchable_bins && !batchable_indexed))
        return fail(ctx, MI_ERR_INVALID_ARG, "mi_batch_upload_bins: NULL table");
    if ((uint64_t)ctx->bt_n_sets + n_batchable_bins + 2ull * n_unbatchable_bins > 65536u)
        return fail(ctx, MI_ERR_INVALID_ARG, "mi_batch_upload_bins: %u sets + %u batchable + 2 x %u unbatchable bins exceed 65536 buckets", ctx->bt_n_sets,
                    n_batchable_bins, n_unbatchable_bins);
    ctx->bt_unb_indexed.assign(unbatchable_indexed, unbatchable_indexed + n_unbatchable_bins);
    ctx->bt_bat_indexed.assign(batchable_indexed, batchable_indexed + n_batchable_bins);
    ctx->bt_resolve = true;
    ctx->bt_desc_dirty = true;
    ctx->bt_built = false;
    return MI_OK;
}

namespace {

// output arrays big enough for `extra` more entries of each kind above `initial`; the region below `initial` reads as zeros
int32_t batch_outputs(mi_ctx* ctx, const BatchInitial& ini, size_t extra_items, size_t extra_ip, size_t extra_sets, uint32_t* wi[2], uint32_t* md[2],
                      uint32_t* bs[2]) {
    int32_t rc;
    for (int c = 0; c < 2; ++c) {
        if ((rc = ensure(ctx, ctx->bt_wi[c], ((size_t)ini.work_item_index[c] + extra_items + 1) * 8))) return rc;
        if ((rc = ensure(ctx, ctx->bt_md[c], ((size_t)ini.indirect_parameters_index[c] + extra_ip + 1) * 20))) return rc;
        if ((rc = ensure(ctx, ctx->bt_bs[c], ((size_t)ini.batch_set_index[c] + extra_sets + 1) * 8))) return rc;
        if (ini.work_item_index[c]) HIP_TRY(ctx, hipMemsetAsync(ctx->bt_wi[c].p, 0, (size_t)ini.work_item_index[c] * 8, ctx->stream));
        if (ini.indirect_parameters_index[c]) HIP_TRY(ctx, hipMemsetAsync(ctx->bt_md[c].p, 0, (size_t)ini.indirect_parameters_index[c] * 20, ctx->stream));
        if (ini.batch_set_index[c]) HIP_TRY(ctx, hipMemsetAsync(ctx->bt_bs[c].p, 0, (size_t)ini.batch_set_index[c] * 8, ctx->stream));
        wi[c] = (uint32_t*)ctx->bt_wi[c].p;
        md[c] = (uint32_t*)ctx->bt_md[c].p;
        bs[c] = (uint32_t*)ctx->bt_bs[c].p;
    }
    return MI_OK;
}

int32_t batch_empty_totals(mi_ctx* ctx, const BatchInitial& ini) {
    mi_batch_totals t{};
    for (int c = 0; c < 2; ++c) {
        t.work_item_len[c] = ini.work_item_index[c];
        t.indirect_parameters_len[c] = ini.indirect_parameters_index[c];
        t.batch_set_len[c] = ini.batch_set_index[c];
    }
    t.data_buffer_len = ini.output_mesh_uniform_index;
    return upload(ctx, ctx->bt_totals.p, &t, sizeof t);
}

}  // namespace

int32_t mi_batch_build(mi_ctx* ctx, uint32_t view, uint32_t class_bit, const mi_batch_initial* initial) {
    return mi_batch_build_phase(ctx, view, class_bit, initial, 0u);
}

int32_t mi_batch_build_phase(mi_ctx* ctx, uint32_t view, uint32_t class_bit, const mi_batch_initial* initial, uint32_t flags) {
    ENTER(ctx);
    if (!ctx->culled) return fail(ctx, MI_ERR_NOT_READY, "mi_batch_build before mi_cull");
    if (flags & ~MI_BATCH_NO_INDIRECT_DRAWING) return fail(ctx, MI_ERR_INVALID_ARG, "mi_batch_build_phase: unknown flags 0x%x", flags);
    {
        int32_t rcj = compaction_join(ctx);  // asynchronous compaction: the list this build reads must be complete
        if (rcj) return rcj;
    }
    const uint32_t n_unb = (uint32_t)ctx->bt_unb_indexed.size(), n_bat = (uint32_t)ctx->bt_bat_indexed.size();
    if ((!ctx->bt_have_sets && n_unb + n_bat == 0) || !ctx->bt_have_rows)
        return fail(ctx, MI_ERR_NOT_READY, "mi_batch_build before mi_batch_upload_rows / mi_batch_upload_sets");
    if (view >= ctx->compact_views) return fail(ctx, MI_ERR_INVALID_ARG, "mi_batch_build: view %u of %u", view, ctx->compact_views);
    static_assert(sizeof(mi_batch_initial) == sizeof(BatchInitial), "mi_batch_initial layout");
    static_assert(sizeof(mi_batch_totals) == 36, "mi_batch_totals layout");
    const bool no_indirect = (flags & MI_BATCH_NO_INDIRECT_DRAWING) != 0;
    BatchArgs a{};
    if (initial) memcpy(&a.initial, initial, sizeof a.initial);
    uint32_t slot = 0xFFFFFFFFu;
    for (uint32_t k = 0; k < ctx->compact_classes; ++k)
        if (ctx->class_bits[k] == class_bit) slot = k;
    int32_t rc;
    // without indirect drawing the multidrawable sets are not part of the phase
    const uint32_t n_sets = no_indirect ? 0u : ctx->bt_n_sets;
    const uint32_t n_buckets = 2u * n_unb + n_bat + n_sets;
    const uint32_t n_tiles = std::max<uint32_t>(1u, (ctx->n + BATCH_TILE - 1u) / BATCH_TILE);
    const size_t cap_rows = (size_t)n_tiles * BATCH_TILE;
    if (n_buckets > 256u) {
        if ((rc = ensure(ctx, ctx->bt_rows_a, cap_rows * 4))) return rc;
        if ((rc = ensure(ctx, ctx->bt_rows_b, cap_rows * 4))) return rc;
        if ((rc = ensure(ctx, ctx->bt_set_count, (size_t)n_buckets * 2 * 4))) return rc;
    }
    if ((rc = ensure(ctx, ctx->bt_hist, (size_t)256 * n_tiles * 4))) return rc;
    if ((rc = ensure(ctx, ctx->bt_plan, std::max<size_t>(n_buckets, 1) * 7 * 4))) return rc;
    if ((rc = ensure(ctx, ctx->bt_bucket_desc, std::max<size_t>(n_buckets, 1) * 4))) return rc;
    if ((rc = ensure(ctx, ctx->bt_counters, 64))) return rc;
    if ((rc = ensure(ctx, ctx->bt_records, std::max<size_t>(n_bat + n_sets, 1) * 32))) return rc;
    if ((rc = ensure(ctx, ctx->bt_unb, ((size_t)ctx->n + 1) * 8))) return rc;
    if ((rc = ensure(ctx, ctx->bt_totals, 64))) return rc;
    if ((rc = ensure(ctx, ctx->bt_meta_out, std::max<size_t>(ctx->bt_n_meta, 1) * 12))) return rc;
    for (int k = 0; k < 2; ++k)
        if (!ctx->bt_inst[k].p) {
            if ((rc = ensure(ctx, ctx->bt_inst[k], std::max<size_t>(ctx->bt_n_meta, 1) * 4 * BATCH_INST_STRIDE))) return rc;
            HIP_TRY(ctx, hipMemsetAsync(ctx->bt_inst[k].p, 0, std::max<size_t>(ctx->bt_n_meta, 1) * 4 * BATCH_INST_STRIDE, ctx->stream));
        }
    // capacities: every row of the list could be a work item / an unbatchable entity with its own slot and batch set; every bin
    // of every set gets a metadata entry
    if ((rc = batch_outputs(ctx, a.initial, ctx->n, (size_t)ctx->n + ctx->bt_n_meta, (size_t)ctx->n + n_sets, a.work_items, a.metadata, a.batch_sets)))
        return rc;
    ctx->bt_last_sorted = false;
    if ((slot == 0xFFFFFFFFu || ctx->n == 0 || n_sets == 0) && ctx->bt_n_meta)  // no batch set is visited: every bin is empty
        if ((rc = upload(ctx, ctx->bt_meta_out.p, ctx->bt_meta_zero.data(), (size_t)ctx->bt_n_meta * 12))) return rc;
    if (slot == 0xFFFFFFFFu || ctx->n == 0 || n_buckets == 0) {
        // no row carries this class: VisibleEntities::get() is empty -> nothing is appended
        if ((rc = batch_empty_totals(ctx, a.initial))) return rc;
        ctx->bt_built = true;
        return MI_OK;
    }
    if (ctx->bt_desc_dirty || ctx->bt_desc_no_indirect != no_indirect) {
        std::vector<uint32_t> desc(n_buckets);
        uint32_t k = 0;
        for (uint32_t b = 0; b < n_unb; ++b) {
            const uint32_t cls = ctx->bt_unb_indexed[b] ? 4u : 0u;
            desc[k++] = 0u | cls | (b << 3);  // rows with an input index
            desc[k++] = 1u | cls | (b << 3);  // rows without
        }
        for (uint32_t b = 0; b < n_bat; ++b) desc[k++] = 2u | (ctx->bt_bat_indexed[b] ? 4u : 0u) | (b << 3);
        for (uint32_t s2 = 0; s2 < n_sets; ++s2) desc[k++] = 3u | (ctx->bt_set_indexed_host[s2] ? 4u : 0u) | (s2 << 3);
        if ((rc = upload(ctx, ctx->bt_bucket_desc.p, desc.data(), desc.size() * 4))) return rc;
        ctx->bt_desc_dirty = false;
        ctx->bt_desc_no_indirect = no_indirect;
        ctx->bt_resolve = true;  // the bucket numbering depends on the same tables
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
    if (ctx->bt_resolve) {
        static const uint32_t* none = nullptr;
        (void)none;
        HIP_TRY(ctx, launch_batch_resolve_rows(ctx->n, n_sets, n_unb, n_bat, ctx->bt_kind, ctx->bt_cpu_bin, ctx->bt_set, ctx->bt_bin, ctx->bt_input,
                                               (const uint32_t*)ctx->bt_table_off.p, (const uint32_t*)ctx->bt_table.p,
                                               (const uint32_t*)ctx->bt_meta_off.p, ctx->bt_row_meta, ctx->bt_bucket, ctx->stream));
        ctx->bt_resolve = false;
    }
    a.row_bucket = ctx->bt_bucket;
    a.row_input = ctx->bt_input;
    a.row_meta = ctx->bt_row_meta;
    a.n_buckets = n_buckets;
    a.first_set_bucket = 2u * n_unb + n_bat;
    a.n_sets = n_sets;
    a.n_meta = ctx->bt_n_meta;
    a.no_indirect = no_indirect ? 1u : 0u;
    a.bucket_desc = (const uint32_t*)ctx->bt_bucket_desc.p;
    a.set_indexed = (const uint8_t*)ctx->bt_set_indexed.p;
    a.meta_offset = (const uint32_t*)ctx->bt_meta_off.p;
    a.bin_meta_in = (const uint32_t*)ctx->bt_meta.p;
    a.bin_metadata_out = (uint32_t*)ctx->bt_meta_out.p;
    a.inst_count = (uint32_t*)ctx->bt_inst[ctx->bt_inst_cur].p;
    a.inst_count_next = (uint32_t*)ctx->bt_inst[ctx->bt_inst_cur ^ 1u].p;
    if (n_sets) ctx->bt_inst_cur ^= 1u;  // the allocate workgroups of this build zero the other buffer
    a.rows_a = (uint32_t*)ctx->bt_rows_a.p;
    a.rows_b = (uint32_t*)ctx->bt_rows_b.p;
    a.tile_hist = (uint32_t*)ctx->bt_hist.p;
    a.n_tiles = n_tiles;
    a.set_count = (uint32_t*)ctx->bt_set_count.p;
    a.plan = (uint32_t*)ctx->bt_plan.p;
    a.counters = (uint32_t*)ctx->bt_counters.p;
    a.unbatchable = (uint32_t*)ctx->bt_unb.p;
    a.records = (uint32_t*)ctx->bt_records.p;
    a.totals = (uint32_t*)ctx->bt_totals.p;
    HIP_TRY(ctx, launch_batch_build(a, ctx->stream, prof_mark, ctx));
    ctx->bt_built = true;
    return MI_OK;
}

int32_t mi_batch_sorted_build(mi_ctx* ctx, uint32_t n_items, const mi_sorted_item* items, const mi_batch_initial* initial, uint32_t flags) {
    ENTER(ctx);
    if (n_items && !items) return fail(ctx, MI_ERR_INVALID_ARG, "mi_batch_sorted_build: NULL items");
    if (flags & ~(MI_SORTED_AUTOMATIC_BATCHING | MI_SORTED_NO_INDIRECT_DRAWING | MI_SORTED_NO_GPU_PREPROCESSING))
        return fail(ctx, MI_ERR_INVALID_ARG, "mi_batch_sorted_build: unknown flags 0x%x", flags);
    static_assert(sizeof(mi_sorted_item) == 16 && sizeof(mi_sorted_batch) == 24, "sorted item layouts");
    SortedArgs a{};
    if (initial) memcpy(&a.initial, initial, sizeof a.initial);
    int32_t rc;
    if ((rc = ensure(ctx, ctx->bt_batches, ((size_t)n_items + 1) * 24))) return rc;
    if ((rc = ensure(ctx, ctx->bt_totals, 64))) return rc;
    if ((rc = batch_outputs(ctx, a.initial, n_items, n_items, n_items, a.work_items, a.metadata, a.batch_sets))) return rc;
    const bool one_wg = n_items <= SORTED_ONE_WG_ITEMS && n_items <= ctx->bt_sorted_one_wg_limit;
    if (n_items && one_wg) {
        // a short phase is walked by one workgroup that reads the items where they were staged -- pinned host memory, once, in
        // lane-contiguous 16-byte loads: no copy launch in front of the kernel (a DMA submission is ~5 us, the items' trip ~2)
        void* st = nullptr;
        if ((rc = stage_alloc(ctx, (size_t)n_items * 16, &st))) return rc;
        memcpy(st, items, (size_t)n_items * 16);
        void* dev = nullptr;
        HIP_TRY(ctx, hipHostGetDevicePointer(&dev, st, 0));
        a.items = (const uint32_t*)dev;
    } else {
        if ((rc = ensure(ctx, ctx->bt_items, std::max<size_t>(n_items, 1) * 16))) return rc;
        if (n_items && (rc = upload(ctx, ctx->bt_items.p, items, (size_t)n_items * 16))) return rc;
        a.items = (const uint32_t*)ctx->bt_items.p;
    }
    a.n_items = n_items;
    a.automatic_batching = (flags & MI_SORTED_AUTOMATIC_BATCHING) ? 1u : 0u;
    a.no_indirect = (flags & MI_SORTED_NO_INDIRECT_DRAWING) ? 1u : 0u;
    a.merge_only = (flags & MI_SORTED_NO_GPU_PREPROCESSING) ? 1u : 0u;
    a.batches = (uint32_t*)ctx->bt_batches.p;
    a.totals = (uint32_t*)ctx->bt_totals.p;
    if ((rc = ensure(ctx, ctx->bt_sorted_partials, (size_t)batch_sorted_partial_words(std::max(n_items, 1u)) * 4))) return rc;
    HIP_TRY(ctx, launch_batch_sorted(a, ctx->stream, prof_mark, ctx, (uint32_t*)ctx->bt_sorted_partials.p, ctx->bt_sorted_one_wg_limit));
    ctx->bt_built = true;
    ctx->bt_last_sorted = true;
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
    case MI_BATCH_RECORDS: src = ctx->bt_records.p; count = ctx->bt_last_sorted ? 0u : t.n_records; elem = 32; break;
    case MI_BATCH_BIN_METADATA: src = ctx->bt_meta_out.p; count = ctx->bt_n_meta; elem = 12; break;
    case MI_BATCH_UNBATCHABLE_INDICES: src = ctx->bt_unb.p; count = ctx->bt_last_sorted ? 0u : t.n_unbatchable; elem = 8; break;
    case MI_BATCH_SORTED_BATCHES: src = ctx->bt_batches.p; count = ctx->bt_last_sorted ? t.n_records : 0u; elem = 24; break;
    default: return fail(ctx, MI_ERR_INVALID_ARG, "mi_batch_download: unknown array %u", what);
    }
    *out_count = count;
    if (count > capacity_elems) return fail(ctx, MI_ERR_CAPACITY, "mi_batch_download: %u elements, capacity %u", count, capacity_elems);
    if (count && !out) return fail(ctx, MI_ERR_INVALID_ARG, "mi_batch_download: NULL out");
    if (count) return download(ctx, out, src, (size_t)count * elem);
    return MI_OK;
}

// test / bench hook: phases up to this many items take the single-workgroup kernel (default SORTED_ONE_WG_ITEMS); 0 = every phase
// goes through the tiled two-launch form, 0xFFFFFFFF = none does
int32_t mi_debug_set_sorted_one_wg_limit(mi_ctx* ctx, uint32_t items) {
    ENTER(ctx);
    ctx->bt_sorted_one_wg_limit = items;
    return MI_OK;
}

}  // extern "C"
