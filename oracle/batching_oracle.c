/*
 * CPU oracle, part 2: the batching work-item build.  TEST INFRASTRUCTURE ONLY -- same rules as
 * bevy_oracle.c: nothing under bevy_amd/ may call, link or import this.
 *
 * What it restates (all u32 arithmetic):
 *   orc_unpack_bins         crates/bevy_pbr/src/render/unpack_bins.wesl:64-93  (one invocation per binned mesh instance)
 *   orc_allocate_uniforms   crates/bevy_pbr/src/render/allocate_uniforms.wesl:84-236 (local scan / global scan / fan,
 *                           256-wide chunks, Hillis-Steele inside a chunk -- restated step by step, not as "a prefix sum")
 *   orc_batch_build         the CPU bookkeeping around them, MultidrawableBatchSetPreparer::prepare_multidrawable_binned_batch_set
 *                           crates/bevy_render/src/batching/gpu_preprocessing.rs:2497-2580 and its caller :2360-2447, fed from a
 *                           view's VisibleEntities list instead of the retained bins: a bin's instances are the visible rows
 *                           that name it (render_phase/mod.rs:268-318 keeps GpuBinMetadata.instance_count == entities in
 *                           the bin), a batch set without instances is skipped (:2520-2524 returns when it has no bin).
 * Order of a batch set's GpuRenderBinnedMeshInstance array: the reference leaves it unspecified ("this array isn't
 * sorted", unpack_bins.wesl:31-33; swap_remove in render_phase/mod.rs:330-400).  This oracle and the device path both
 * use the VisibleEntities order (ascending Entity), which is one of the orders the reference can produce.
 *
 *   orc_batch_cpu_bins      the unbatchable and batchable loops of gpu_preprocessing::batch_and_prepare_binned_render_phase
 *                           (gpu_preprocessing.rs:2135-2357), the part of a binned phase the reference builds on the CPU
 *   orc_batch_sorted        gpu_preprocessing::batch_and_prepare_sorted_render_phase (:1850-2061) with SortedRenderBatchSet::flush
 *                           (:1767-1794)
 *   orc_batch_sorted_merge  batching::batch_and_prepare_sorted_render_phase (batching/mod.rs:219-244) as
 *                           no_gpu_preprocessing::batch_and_prepare_sorted_render_phase drives it (no_gpu_preprocessing.rs:76-103)
 *
 * PARITY UNPINNED for this file: the reference's only test of these structures is a proptest of invariants
 * (render_phase/mod.rs:2356-2700, no golden values) and the shaders have no test; tests/test_oracle_batching.py checks
 * the same invariants (instance counts per bin, work items <-> instances, contiguous output ranges) on this oracle.
 */
#include "bevy_oracle.h"

#include <stdlib.h>
#include <string.h>

/* unpack_bins.wesl:64-93 */
void orc_unpack_bins(uint32_t base_output_work_item_index, uint32_t base_indirect_parameters_index,
                     uint32_t binned_mesh_instance_count, const orc_binned_mesh_instance* binned_mesh_instances,
                     const orc_bin_metadata* bin_metadata, const uint32_t* bin_index_to_bin_metadata_index,
                     orc_preprocess_work_item* preprocess_work_items) {
    for (uint32_t global_id = 0; global_id < binned_mesh_instance_count; ++global_id) {
        const uint32_t input_uniform_index = binned_mesh_instances[global_id].input_uniform_index;
        const uint32_t bin_index = binned_mesh_instances[global_id].bin_index;
        const uint32_t bin_metadata_index = bin_index_to_bin_metadata_index[bin_index];
        const uint32_t indirect_parameters_offset = bin_metadata[bin_metadata_index].indirect_parameters_offset;
        const uint32_t output_index = base_output_work_item_index + global_id;
        preprocess_work_items[output_index].input_index = input_uniform_index;
        preprocess_work_items[output_index].output_or_indirect_parameters_index =
            base_indirect_parameters_index + indirect_parameters_offset;
    }
}

#define WG 256u

/* allocate_uniforms.wesl:222-232: in-place Hillis-Steele over the 256 workgroup slots */
static void hillis_steele_scan(uint32_t* output_offsets) {
    uint32_t term[WG];
    for (uint32_t offset = 1; offset < WG; offset *= 2) {
        for (uint32_t l = 0; l < WG; ++l) term[l] = l >= offset ? output_offsets[l - offset] : 0u;
        for (uint32_t l = 0; l < WG; ++l) output_offsets[l] += term[l];
    }
}

/* allocate_uniforms.wesl: allocate_local_scan :84-150, allocate_global_scan :160-192, allocate_fan :202-218.
 * fan_buffer needs ceil(bin_count / 256) words. */
void orc_allocate_uniforms(uint32_t batch_set_index, uint32_t bin_count, uint32_t first_indirect_parameters_index,
                           uint32_t first_output_mesh_uniform_index, const orc_bin_metadata* bin_metadata,
                           orc_indirect_parameters_metadata* indirect_parameters_metadata, uint32_t* fan_buffer) {
    const uint32_t chunk_count = (bin_count + WG - 1u) / WG;
    uint32_t output_offsets[WG];
    /* step 1, one workgroup per chunk */
    for (uint32_t group = 0; group < chunk_count; ++group) {
        const uint32_t block_start = group * WG;
        const uint32_t block_end = block_start + WG < bin_count ? block_start + WG : bin_count;
        for (uint32_t l = 0; l < WG; ++l) output_offsets[l] = group == 0 ? first_output_mesh_uniform_index : 0u;
        for (uint32_t l = 0; l < WG - 1u; ++l)
            if (block_start + l < block_end) output_offsets[l + 1] = bin_metadata[block_start + l].instance_count;
        /* NB: slot 0 keeps first_output (group 0) and slots l+1 are *assigned*, as the shader does */
        hillis_steele_scan(output_offsets);
        for (uint32_t l = 0; l < WG; ++l) {
            const uint32_t gid = block_start + l;
            if (gid < block_end) {
                const uint32_t off = first_indirect_parameters_index + bin_metadata[gid].indirect_parameters_offset;
                indirect_parameters_metadata[off].base_output_index = output_offsets[l];
                indirect_parameters_metadata[off].batch_set_index = batch_set_index;
                indirect_parameters_metadata[off].mesh_index = 0;
                indirect_parameters_metadata[off].early_instance_count = 0;
                indirect_parameters_metadata[off].late_instance_count = 0;
            }
        }
        uint32_t chunk_total = output_offsets[WG - 1u];
        if (block_start + WG - 1u < block_end) chunk_total += bin_metadata[block_start + WG - 1u].instance_count;
        fan_buffer[group] = chunk_total;
    }
    if (bin_count <= WG) return; /* steps 2 and 3 are not dispatched (gpu_preprocess.rs allocate_uniforms, "fewer than 256") */
    /* step 2, one workgroup: exclusive offsets going INTO each chunk, blocks of 256 chunks with a running sum */
    uint32_t sum = 0;
    memset(output_offsets, 0, sizeof output_offsets); /* a new dispatch: workgroup memory starts zeroed (WGSL) */
    for (uint32_t block_start = 0; block_start < chunk_count; block_start += WG) {
        const uint32_t block_end = block_start + WG < chunk_count ? block_start + WG : chunk_count;
        /* slots past block_end keep whatever the previous block left there; they only ever feed slots to their
         * right, which are past block_end too, except through slot 255 which the shader reads as the block's sum.
         * The shader has the same property; restate it literally. */
        for (uint32_t l = 0; l < WG; ++l)
            if (block_start + l < block_end) output_offsets[l] = fan_buffer[block_start + l];
        hillis_steele_scan(output_offsets);
        for (uint32_t l = 0; l < WG; ++l)
            if (block_start + l < block_end) fan_buffer[block_start + l] = sum + output_offsets[l];
        sum += output_offsets[WG - 1u];
    }
    /* step 3: chunk g >= 1 adds fan_buffer[g - 1] (the inclusive total of everything before it) */
    for (uint32_t id = WG; id < bin_count; ++id) {
        const uint32_t group = (id - WG) / WG;
        const uint32_t off = first_indirect_parameters_index + bin_metadata[id].indirect_parameters_offset;
        indirect_parameters_metadata[off].base_output_index += fan_buffer[group];
    }
}

/* One phase of one view: every batch set of the phase against the view's VisibleEntities list.
 * rows[]: ascending rows of the list.  Per row: batch set (ORC_NO_BATCH_SET = not multidrawable, skipped), RenderBinIndex
 * inside that set, InputUniformIndex.  Sets are visited in index order (the iteration order of
 * phase.multidrawable_meshes); per set s its bins are bin_table[bin_table_offset[s] .. bin_table_offset[s+1]) (RenderBinIndex
 * -> metadata index, relative to the set) and its metadata bin_metadata[meta_offset[s] .. meta_offset[s+1]).
 * Outputs are appended per mesh class (0 = non-indexed, 1 = indexed), exactly as the two preparers do; MeshUniform
 * slots (data_buffer) are shared by both classes and run in set order.  instance_count of every bin_metadata entry is
 * (re)written.  Returns the number of non-empty batch sets. */
uint32_t orc_batch_build(uint32_t n_list, const uint32_t* rows, const uint32_t* row_batch_set, const uint32_t* row_bin_index,
                         const uint32_t* row_input_uniform_index, uint32_t n_sets, const uint8_t* set_indexed,
                         const uint32_t* bin_table_offset, const uint32_t* bin_table, const uint32_t* meta_offset,
                         orc_bin_metadata* bin_metadata, const orc_batch_initial* initial,
                         orc_preprocess_work_item* work_items[2], orc_indirect_parameters_metadata* metadata[2],
                         orc_indirect_batch_set* batch_sets[2], orc_batch_set_record* records, orc_batch_totals* totals) {
    uint32_t work_item_len[2] = {initial->work_item_index[0], initial->work_item_index[1]};
    uint32_t indirect_index[2] = {initial->indirect_parameters_index[0], initial->indirect_parameters_index[1]};
    uint32_t batch_set_index[2] = {initial->batch_set_index[0], initial->batch_set_index[1]};
    uint32_t data_buffer_len = initial->output_mesh_uniform_index;
    uint32_t n_records = 0;
    orc_binned_mesh_instance* inst = (orc_binned_mesh_instance*)malloc(sizeof(orc_binned_mesh_instance) * (n_list ? n_list : 1));
    for (uint32_t m = 0; m < meta_offset[n_sets]; ++m) bin_metadata[m].instance_count = 0;
    for (uint32_t s = 0; s < n_sets; ++s) {
        const uint32_t cls = set_indexed[s] ? 1u : 0u;
        orc_bin_metadata* meta = bin_metadata + meta_offset[s];
        const uint32_t* table = bin_table + bin_table_offset[s];
        const uint32_t bin_count = meta_offset[s + 1] - meta_offset[s];
        /* the set's binned instances = the visible rows that name it, in list order; insert() bumps instance_count */
        uint32_t instance_count = 0;
        for (uint32_t i = 0; i < n_list; ++i) {
            const uint32_t row = rows[i];
            if (row_batch_set[row] != s) continue;
            inst[instance_count].input_uniform_index = row_input_uniform_index[row];
            inst[instance_count].bin_index = row_bin_index[row];
            meta[table[row_bin_index[row]]].instance_count += 1;
            ++instance_count;
        }
        if (instance_count == 0) continue; /* no bin in the set */
        /* prepare_multidrawable_binned_batch_set, gpu_preprocessing.rs:2511-2579 */
        const uint32_t current_output_index = data_buffer_len;
        const uint32_t first_work_item_index = work_item_len[cls];
        const uint32_t indirect_parameters_base = indirect_index[cls];
        const uint32_t first_indirect_parameters_index = indirect_parameters_base; /* metadata and draw params are parallel arrays (:2545) */
        data_buffer_len += instance_count;
        work_item_len[cls] += instance_count;
        orc_indirect_batch_set* bs = batch_sets[cls] + batch_set_index[cls];
        bs->indirect_parameters_count = 0;
        bs->indirect_parameters_base = indirect_parameters_base;
        orc_batch_set_record* rec = records + n_records++;
        rec->set = s;
        rec->indexed = cls;
        rec->index = batch_set_index[cls];
        rec->first_work_item_index = first_work_item_index;
        rec->instance_count = instance_count;
        rec->first_indirect_parameters_index = first_indirect_parameters_index;
        rec->batch_count = bin_count;
        rec->first_output_mesh_uniform_index = current_output_index;
        indirect_index[cls] += bin_count;
        /* the two compute passes of this batch set */
        uint32_t* fan = (uint32_t*)calloc((bin_count + WG - 1u) / WG + 1u, 4);
        orc_allocate_uniforms(batch_set_index[cls], bin_count, first_indirect_parameters_index, current_output_index, meta,
                              metadata[cls], fan);
        free(fan);
        orc_unpack_bins(first_work_item_index, indirect_parameters_base, instance_count, inst, meta, table, work_items[cls]);
        batch_set_index[cls] += 1;
    }
    free(inst);
    for (int c = 0; c < 2; ++c) {
        totals->work_item_len[c] = work_item_len[c];
        totals->indirect_parameters_len[c] = indirect_index[c];
        totals->batch_set_len[c] = batch_set_index[c];
    }
    totals->data_buffer_len = data_buffer_len;
    totals->n_records = n_records;
    return n_records;
}


/* ================================================================================================================
 * The CPU-built part of a binned phase: unbatchables, then batchables (gpu_preprocessing.rs:2135-2357), for ONE view.
 * The phase's bins are derived from the view's VisibleEntities list exactly like the multidrawable ones above: bin b of a
 * kind holds the listed rows that name it, in list order; a bin nobody names does not exist in the phase.  Bins are visited
 * in index order = the order phase.unbatchable_meshes / phase.batchable_meshes iterate after sort_binned_render_phase
 * (batching/mod.rs:199-209).  The buffers are modelled as the reference's growable vectors: *_len is the vector length,
 * allocate(count) = push_multiple_init(count) zero-fills (:1515-1522).  Returns the number of batch records written.
 * ================================================================================================================ */
uint32_t orc_batch_cpu_bins(uint32_t n_list, const uint32_t* rows, const uint8_t* row_kind, const uint32_t* row_bin,
                            const uint32_t* row_input_uniform_index, uint32_t n_unbatchable_bins, const uint8_t* unbatchable_indexed,
                            uint32_t n_batchable_bins, const uint8_t* batchable_indexed, int no_indirect_drawing,
                            const orc_batch_initial* initial, orc_preprocess_work_item* work_items[2],
                            orc_indirect_parameters_metadata* metadata[2], orc_indirect_batch_set* batch_sets[2],
                            orc_unbatchable_index* unbatchable_indices, uint32_t* n_unbatchable_indices,
                            orc_batch_set_record* records, orc_batch_totals* totals) {
    uint32_t work_item_len[2] = {initial->work_item_index[0], initial->work_item_index[1]};
    uint32_t indirect_len[2] = {initial->indirect_parameters_index[0], initial->indirect_parameters_index[1]};
    uint32_t batch_set_len[2] = {initial->batch_set_index[0], initial->batch_set_index[1]};
    uint32_t data_buffer_len = initial->output_mesh_uniform_index;
    uint32_t n_records = 0, n_unb = 0;
    uint32_t* entities = (uint32_t*)malloc(4u * (n_list ? n_list : 1u));

    /* Prepare unbatchables (:2148-2224) */
    for (uint32_t bin = 0; bin < n_unbatchable_bins; ++bin) {
        uint32_t len = 0;
        for (uint32_t i = 0; i < n_list; ++i)
            if (row_kind[rows[i]] == ORC_ROW_UNBATCHABLE && row_bin[rows[i]] == bin) entities[len++] = rows[i];
        if (len == 0) continue;
        const uint32_t cls = unbatchable_indexed[bin] ? 1u : 0u; /* key.0.indexed() */
        /* allocate(unbatchables.entities.len()) -- for every entity of the bin, whether or not it has an input index */
        int have_offset = !no_indirect_drawing;
        uint32_t indirect_parameters_index = 0;
        if (have_offset) {
            indirect_parameters_index = indirect_len[cls];
            memset(metadata[cls] + indirect_len[cls], 0, sizeof(orc_indirect_parameters_metadata) * len);
            indirect_len[cls] += len;
        }
        for (uint32_t e = 0; e < len; ++e) {
            const uint32_t input_index = row_input_uniform_index[entities[e]];
            if (input_index == ORC_NO_INPUT_INDEX) continue; /* get_binned_index() == None */
            const uint32_t output_index = data_buffer_len++;   /* data_buffer.add() */
            if (have_offset) {
                orc_indirect_parameters_metadata* md = metadata[cls] + indirect_parameters_index;
                md->base_output_index = output_index; /* write_batch_indirect_parameters_metadata, mesh.rs:3003-3031 */
                md->batch_set_index = 0xFFFFFFFFu;    /* None => !0 */
                md->mesh_index = md->early_instance_count = md->late_instance_count = 0;
                work_items[cls][work_item_len[cls]].input_index = input_index;
                work_items[cls][work_item_len[cls]].output_or_indirect_parameters_index = indirect_parameters_index;
                ++work_item_len[cls];
                unbatchable_indices[n_unb].bin = bin;
                unbatchable_indices[n_unb].instance_index = indirect_parameters_index; /* extra: range idx..idx+1, set None */
                ++n_unb;
                batch_sets[cls][batch_set_len[cls]].indirect_parameters_base = indirect_parameters_index; /* add_batch_set :1293-1320 */
                batch_sets[cls][batch_set_len[cls]].indirect_parameters_count = 0;
                ++batch_set_len[cls];
                ++indirect_parameters_index;
            } else {
                work_items[cls][work_item_len[cls]].input_index = input_index;
                work_items[cls][work_item_len[cls]].output_or_indirect_parameters_index = output_index;
                ++work_item_len[cls];
                unbatchable_indices[n_unb].bin = bin;
                unbatchable_indices[n_unb].instance_index = output_index; /* extra: None */
                ++n_unb;
            }
        }
    }

    /* Prepare batchables (:2228-2353) */
    for (uint32_t bin = 0; bin < n_batchable_bins; ++bin) {
        const uint32_t cls = batchable_indexed[bin] ? 1u : 0u;
        int have_batch = 0;
        uint32_t range_start = 0, range_end = 0, extra_start = ORC_NO_INDEX, batch_set_index = 0;
        for (uint32_t i = 0; i < n_list; ++i) {
            const uint32_t row = rows[i];
            if (row_kind[row] != ORC_ROW_BATCHABLE || row_bin[row] != bin) continue;
            const uint32_t input_index = row_input_uniform_index[row];
            const uint32_t output_index = data_buffer_len++;
            orc_preprocess_work_item* wi = work_items[cls] + work_item_len[cls]++;
            wi->input_index = input_index;
            if (have_batch) {
                range_end = output_index + 1u;
                /* indirect: the first output index of the batch's indirect parameters; direct: the output index itself */
                wi->output_or_indirect_parameters_index = no_indirect_drawing ? output_index : extra_start;
            } else if (!no_indirect_drawing) {
                const uint32_t indirect_parameters_index = indirect_len[cls]; /* allocate(indexed, 1) */
                memset(metadata[cls] + indirect_len[cls], 0, sizeof(orc_indirect_parameters_metadata));
                indirect_len[cls] += 1;
                batch_set_index = batch_set_len[cls]; /* get_next_batch_set_index */
                orc_indirect_parameters_metadata* md = metadata[cls] + indirect_parameters_index;
                md->base_output_index = output_index;
                md->batch_set_index = batch_set_index;
                md->mesh_index = md->early_instance_count = md->late_instance_count = 0;
                batch_sets[cls][batch_set_len[cls]].indirect_parameters_base = indirect_parameters_index;
                batch_sets[cls][batch_set_len[cls]].indirect_parameters_count = 0;
                ++batch_set_len[cls];
                wi->output_or_indirect_parameters_index = indirect_parameters_index;
                have_batch = 1;
                range_start = output_index;
                range_end = output_index + 1u;
                extra_start = indirect_parameters_index;
            } else {
                wi->output_or_indirect_parameters_index = output_index;
                have_batch = 1;
                range_start = output_index;
                range_end = output_index + 1u;
                extra_start = ORC_NO_INDEX;
            }
        }
        if (have_batch) { /* :2313-2352: Direct(vec).push(batch) or MultidrawIndirect(vec).push(batch set of one batch) */
            orc_batch_set_record* rec = records + n_records++;
            rec->set = ORC_RECORD_BATCHABLE_BIN | bin;
            rec->indexed = cls;
            rec->index = no_indirect_drawing ? 0u : batch_set_index;
            rec->first_work_item_index = 0; /* "Unused." */
            rec->instance_count = range_end - range_start;
            rec->first_indirect_parameters_index = extra_start; /* extra_index range start, or None */
            rec->batch_count = 1;
            rec->first_output_mesh_uniform_index = range_start; /* first_batch.instance_range.start */
        }
    }
    free(entities);
    for (int c = 0; c < 2; ++c) {
        totals->work_item_len[c] = work_item_len[c];
        totals->indirect_parameters_len[c] = indirect_len[c];
        totals->batch_set_len[c] = batch_set_len[c];
    }
    totals->data_buffer_len = data_buffer_len;
    totals->n_records = n_records;
    *n_unbatchable_indices = n_unb;
    return n_records;
}

/* ================================================================================================================
 * Sorted phases with GPU preprocessing (gpu_preprocessing.rs:1850-2061).  items[] is phase.items in its sorted order; per item
 * what GFBD::get_index_and_compare_data returned: input_index (ORC_NO_INPUT_INDEX = None: "not part of this pipeline"), and, if
 * ORC_ITEM_HAS_COMPARE_DATA, the batch-set key -- BatchSetMeta::new(item, batch_set_compare_data), i.e. pipeline id, draw
 * function, dynamic offset and the compare data interned into one id (batching/mod.rs:42-72) -- and the bin key (BatchCompareData).
 * Returns the number of batch sets; batches[k] is what flush() writes on the k-th set's first item.
 * ================================================================================================================ */
typedef struct sorted_batch_set {
    int some;
    uint32_t phase_item_start_index, instance_start_index, indexed;
    int have_range;
    uint32_t range_start, range_end;
    int meta_some;
    uint32_t meta_set_key, meta_bin_key;
} sorted_batch_set;

static void sorted_flush(sorted_batch_set* bs, uint32_t instance_end_index, orc_indirect_batch_set* batch_sets[2],
                         uint32_t batch_set_len[2], orc_sorted_batch* batches, uint32_t* n_batches) {
    orc_sorted_batch* b = batches + (*n_batches)++;
    b->first_item = bs->phase_item_start_index;
    b->instance_start = bs->instance_start_index;
    b->instance_end = instance_end_index;
    b->indirect_parameters_start = bs->have_range ? bs->range_start : ORC_NO_INDEX;
    b->indirect_parameters_end = bs->have_range ? bs->range_end : ORC_NO_INDEX;
    b->indexed = bs->indexed;
    if (bs->have_range) {
        batch_sets[bs->indexed][batch_set_len[bs->indexed]].indirect_parameters_base = bs->range_start;
        batch_sets[bs->indexed][batch_set_len[bs->indexed]].indirect_parameters_count = 0;
        ++batch_set_len[bs->indexed];
    }
    bs->some = 0;
}

uint32_t orc_batch_sorted(uint32_t n_items, const orc_sorted_item* items, int automatic_batching, int no_indirect_drawing,
                          const orc_batch_initial* initial, orc_preprocess_work_item* work_items[2],
                          orc_indirect_parameters_metadata* metadata[2], orc_indirect_batch_set* batch_sets[2],
                          orc_sorted_batch* batches, orc_batch_totals* totals) {
    enum { BATCH_OK, BREAK_BATCH, BREAK_BATCH_SET };
    uint32_t work_item_len[2] = {initial->work_item_index[0], initial->work_item_index[1]};
    uint32_t indirect_len[2] = {initial->indirect_parameters_index[0], initial->indirect_parameters_index[1]};
    uint32_t batch_set_len[2] = {initial->batch_set_index[0], initial->batch_set_index[1]};
    uint32_t data_buffer_len = initial->output_mesh_uniform_index;
    uint32_t n_batches = 0;
    sorted_batch_set batch_set;
    memset(&batch_set, 0, sizeof batch_set);
    for (uint32_t current_index = 0; current_index < n_items; ++current_index) {
        const orc_sorted_item* item = items + current_index;
        const uint32_t item_is_indexed = (item->flags & ORC_ITEM_INDEXED) ? 1u : 0u;
        if (item->input_index == ORC_NO_INPUT_INDEX) { /* :1905-1917: break the batch, skip the item */
            if (batch_set.some) sorted_flush(&batch_set, data_buffer_len, batch_sets, batch_set_len, batches, &n_batches);
            continue;
        }
        const int current_meta_some = automatic_batching && (item->flags & ORC_ITEM_HAS_COMPARE_DATA); /* :1918-1929 */
        int can_batch = BREAK_BATCH_SET; /* :1933-1957 */
        if (batch_set.some && current_meta_some && batch_set.meta_some) {
            if (item->batch_set_key == batch_set.meta_set_key) {
                if (item->bin_key == batch_set.meta_bin_key) can_batch = BATCH_OK;
                else can_batch = no_indirect_drawing ? BREAK_BATCH_SET : BREAK_BATCH;
            }
        }
        const uint32_t output_index = data_buffer_len++; /* :1960 */
        if (can_batch == BREAK_BATCH_SET) { /* :1965-1997 */
            if (batch_set.some) sorted_flush(&batch_set, output_index, batch_sets, batch_set_len, batches, &n_batches);
            const int have_index = !no_indirect_drawing;
            uint32_t indirect_parameters_index = 0;
            if (have_index) {
                indirect_parameters_index = indirect_len[item_is_indexed]++;
                orc_indirect_parameters_metadata* md = metadata[item_is_indexed] + indirect_parameters_index;
                md->base_output_index = output_index;
                md->batch_set_index = 0xFFFFFFFFu;
                md->mesh_index = md->early_instance_count = md->late_instance_count = 0;
            }
            batch_set.some = 1;
            batch_set.phase_item_start_index = current_index;
            batch_set.instance_start_index = output_index;
            batch_set.indexed = item_is_indexed;
            batch_set.have_range = have_index;
            batch_set.range_start = indirect_parameters_index;
            batch_set.range_end = indirect_parameters_index + 1u;
            batch_set.meta_some = current_meta_some;
            batch_set.meta_set_key = item->batch_set_key;
            batch_set.meta_bin_key = item->bin_key;
        } else if (can_batch == BREAK_BATCH) { /* :1999-2027; only reached with indirect drawing and a live batch set */
            const uint32_t indirect_parameters_index = indirect_len[item_is_indexed]++;
            orc_indirect_parameters_metadata* md = metadata[item_is_indexed] + indirect_parameters_index;
            md->base_output_index = output_index;
            md->batch_set_index = 0xFFFFFFFFu;
            md->mesh_index = md->early_instance_count = md->late_instance_count = 0;
            batch_set.meta_some = current_meta_some;
            batch_set.meta_set_key = item->batch_set_key;
            batch_set.meta_bin_key = item->bin_key;
            batch_set.range_end += 1; /* debug_assert_eq!(indirect_parameters_index, range.end) holds when the set is one mesh class */
        }
        /* :2034-2052 */
        orc_preprocess_work_item* wi = work_items[item_is_indexed] + work_item_len[item_is_indexed]++;
        wi->input_index = item->input_index;
        wi->output_or_indirect_parameters_index = no_indirect_drawing ? output_index : (batch_set.have_range ? batch_set.range_end - 1u : 0u);
    }
    if (batch_set.some) sorted_flush(&batch_set, data_buffer_len, batch_sets, batch_set_len, batches, &n_batches); /* :2055-2061 */
    for (int c = 0; c < 2; ++c) {
        totals->work_item_len[c] = work_item_len[c];
        totals->indirect_parameters_len[c] = indirect_len[c];
        totals->batch_set_len[c] = batch_set_len[c];
    }
    totals->data_buffer_len = data_buffer_len;
    totals->n_records = n_batches;
    return n_batches;
}

/* ================================================================================================================
 * Sorted phases without GPU preprocessing: the range merge of batching/mod.rs:219-244, fed by the closure of
 * no_gpu_preprocessing.rs:87-101 -- every item with batch data gets the next slot of the instance buffer
 * (batch_range = index..index + 1), then `reduce` folds an item into its predecessor's range when both carry Some(meta) and
 * the metas are equal.  batches[k] = the k-th surviving range (indirect fields unused = ORC_NO_INDEX); an item without batch
 * data keeps whatever batch_range it had and is not listed.  *buffer_len = instance buffer length afterwards.
 * ================================================================================================================ */
uint32_t orc_batch_sorted_merge(uint32_t n_items, const orc_sorted_item* items, int automatic_batching, uint32_t first_index,
                                orc_sorted_batch* batches, uint32_t* buffer_len) {
    uint32_t n_batches = 0, next = first_index;
    int have_prev = 0, prev_meta_some = 0, prev_listed = 0;
    uint32_t prev_set_key = 0, prev_bin_key = 0;
    for (uint32_t i = 0; i < n_items; ++i) {
        const orc_sorted_item* item = items + i;
        const int has_data = item->input_index != ORC_NO_INPUT_INDEX;           /* get_batch_data()? */
        const int meta_some = has_data && automatic_batching && (item->flags & ORC_ITEM_HAS_COMPARE_DATA);
        uint32_t index = 0;
        if (has_data) index = next++;                                           /* batched_instance_buffer.push */
        if (have_prev && meta_some && prev_meta_some && item->batch_set_key == prev_set_key && item->bin_key == prev_bin_key) {
            batches[n_batches - 1u].instance_end = index + 1u;                  /* start_range.end = range.end */
        } else {
            prev_listed = has_data;
            if (has_data) {
                orc_sorted_batch* b = batches + n_batches++;
                b->first_item = i;
                b->instance_start = index;
                b->instance_end = index + 1u;
                b->indirect_parameters_start = b->indirect_parameters_end = ORC_NO_INDEX;
                b->indexed = (item->flags & ORC_ITEM_INDEXED) ? 1u : 0u;
            }
            prev_meta_some = meta_some;
            prev_set_key = item->batch_set_key;
            prev_bin_key = item->bin_key;
        }
        have_prev = 1;
        (void)prev_listed;
    }
    *buffer_len = next;
    return n_batches;
}
