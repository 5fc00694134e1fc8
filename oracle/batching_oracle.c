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
