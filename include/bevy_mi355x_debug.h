/*
 * bevy_mi355x_debug.h -- instrumentation and test hooks of libbevy_mi355x.so.
 *
 * NOT part of the drop-in boundary (include/bevy_mi355x.h): nothing here replaces anything in Bevy, and a plugin never calls it.
 * bench.py times kernels through the mi_profile_* / mi_timer_* functions, the tests force planner decisions and read
 * device-side probes through mi_debug_*.  Same conventions as the main header (int32 status, any thread, one call at a time).
 */
#ifndef BEVY_MI355X_DEBUG_H
#define BEVY_MI355X_DEBUG_H

#include "bevy_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

/* HIP-event timing on the context's stream (torch.cuda.Event only sees torch's stream). */
int32_t mi_timer_begin(mi_ctx* ctx);
int32_t mi_timer_end(mi_ctx* ctx, float* out_ms); /* synchronises */

/* Per-kernel HIP-event profile: while enabled every launch is bracketed by events.
 * mi_profile_read synchronises and returns, for kernel id k < *inout_n: launches[k], total_ms[k].
 * mi_profile_kernel_name(k) names the ids (NULL past the end). */
int32_t mi_profile_enable(mi_ctx* ctx, int32_t enabled);
/* Restricts the profile to the kernels whose id bit is set in kernel_mask (default: all).  Bracketing only
 * the dominant kernel keeps the event overhead out of a timed region. */
int32_t mi_profile_filter(mi_ctx* ctx, uint64_t kernel_mask);
/* Times only every n-th launch of each selected kernel (a timed launch costs several microseconds of host time). */
int32_t mi_profile_sample(mi_ctx* ctx, uint32_t every_n);
/* Times at most the first first_n launches of each selected kernel after mi_profile_enable(1) (0 = no limit): a
 * timed dispatch is fenced off from its neighbours by its timestamp packets (the same isolation rocprofv3's kernel
 * trace imposes), which costs ~5 us of GPU time per launch -- a burst keeps that out of the rest of a timed region. */
int32_t mi_profile_burst(mi_ctx* ctx, uint32_t first_n);
int32_t mi_profile_read(mi_ctx* ctx, uint32_t* inout_n, uint64_t* launches, double* total_ms);
const char* mi_profile_kernel_name(uint32_t k);

/* How the NEXT mi_upload_hierarchy plans mi_propagate: 0 = subtree tiles (a subtree too big for one is cut; a forest of small trees: a wave per tree; a deep or lopsided tree up to 2^20 rows: strips; default), 1 = always level by level, 2 = as 0, 3 = as 0 with the streamed-level thresholds at their test values (2^20 / 2^21 rows), 4 = as 0 without the wave tiles of a forest of small trees and without strips, 5 = strips (one launch of independent workgroups, kernels.h) wherever they can be planned.  (Environment, read at mi_upload_hierarchy: MI_STRIP_W = rows to a level of a strip, 1 .. 128, instead of the planner's own 64 / 128 -- the tests' knob.) */
int32_t mi_debug_set_tile_mode(mi_ctx* ctx, int32_t mode);
/* Per-tile phase timestamps of the light tile kernel (8 x s_memrealtime, 100 MHz, per tile of the first launch).  enable != 0
 * allocates the buffer (mi_propagate then fills it every frame); out != NULL copies n_tiles x 8 stamps out; enable == 0 with
 * out == NULL switches the stamps off again. */
int32_t mi_debug_tree_trace(mi_ctx* ctx, int32_t enable, unsigned long long* out, uint32_t n_tiles);
/* What the multi-GPU exchange cost the calling thread: out5 = frames, ns in the begin step, ns of those spent WAITING for a gathered
 * buffer's previous all-gather (device back-pressure in MI_EXCHANGE_PIPELINED, not work), ns in the end step, ns of the exchange thread. */
int32_t mi_debug_exchange_times(mi_ctx* ctx, double* out5, int32_t reset);
/* The shape of the current tile plan: launches per mi_propagate, tiles, chain tiles (self-evaluated ancestor chains), bands. */
int32_t mi_debug_tile_plan(mi_ctx* ctx, uint32_t* out_launches, uint32_t* out_tiles, uint32_t* out_chain_tiles, uint32_t* out_bands);
/* The strips plan of a hierarchy as mi_upload_hierarchy would make it at `width` (1 .. 128) rows to a level: host code only -- no device,
 * no context.  out_counts = {strips, table entries, bands, snapshot rows} (all zero: cannot be planned); out_strips: 3 words per strip
 * (first table entry; entries | batches << 16 | owns-snapshot-rows << 31; first own level); out_rounds: 4 words per entry (first row;
 * first row of the strip's range in the level above; rows | first LDS slot << 8 | flags from bit 16: level parity, own, forest root,
 * -, directly above the own rows, -, entries of the batch << 22; level).  A strip lists the cone of its rows' ancestors, level 0
 * downwards, then the rows it owns. */
int32_t mi_debug_plan_strips(uint32_t n_levels, const uint32_t* level_offsets, const uint32_t* parent_idx, uint32_t width, uint32_t* out_strips,
                             uint32_t cap_strips, uint32_t* out_rounds, uint32_t cap_rounds, uint32_t* out_counts);
/* The strips of the current plan (kernels.h: one launch of independent waves): rounds and cone rounds per strip, the strip count and the
 * rounds of the whole table (0 when the plan has no strips). */
int32_t mi_debug_strip_plan(mi_ctx* ctx, uint32_t* out_rounds, uint32_t* out_cone_rounds, uint32_t cap, uint32_t* out_n, uint32_t* out_total_rounds);
/* The launches of the current tile plan: out_groups[4g ..] = (first tile, tiles, chain tiles, deep instantiation) per launch; out_tiles[3t ..] =
 * (levels, rows, chain length | 0x100 for a tile of forest roots) per tile (tools/shape_trace.py). */
int32_t mi_debug_tile_groups(mi_ctx* ctx, uint32_t* out_groups, uint32_t cap_groups, uint32_t* out_n_groups, uint32_t* out_tiles, uint32_t cap_tiles);
/* Sorted phases (mi_batch_sorted_build): phases up to `items` long take the single-workgroup kernel (one launch), longer ones the
 * tiled two-launch form over the whole chip.  Default 1024; 0 = always tiled, 0xFFFFFFFF = never. */
int32_t mi_debug_set_sorted_one_wg_limit(mi_ctx* ctx, uint32_t items);
/* The flags-first test of the light tile kernel under the static-scene rule (kernels_tree.hip): 0 = when few rows changed since the
 * last propagate (default), 1 = never, 2 = always. */
int32_t mi_debug_set_tile_pretest(mi_ctx* ctx, int32_t mode);
/* Dense uploads in pieces with the GlobalTransforms fetched ahead (bevy_mi355x.h, mi_download_frame_results): a sequence of dense
 * windows that carries the whole flat table goes out piece by piece on a stream of its own, and -- once the caller has fetched every
 * GlobalTransform of such a frame -- each piece's GlobalTransforms are computed at once and sent back under the rest of the upload.
 * 0 = tables of 262144 rows and more (default), 1 = never, 2 = any row count and fetching ahead from the first sequence on.  Same
 * results. */
int32_t mi_debug_set_chunked_frames(mi_ctx* ctx, int32_t mode);
/* How many dense windows went out as pieces of such a sequence / how many mi_download_frame_results calls handed out GlobalTransforms
 * fetched ahead of an all-rows frame / of a changed-rows frame (written back by the indexed upload window's scatter launch) (tests). */
int32_t mi_debug_chunked_counts(mi_ctx* ctx, uint32_t* out_windows, uint32_t* out_downloads, uint32_t* out_sparse_downloads);
/* The cluster walk of a MI_CULL_WITH_CLUSTERS frame whose objects are bound to a row RANGE: 0 = the frame kernel's row workgroups
 * of those rows go on into the walk (default), 1 = extra workgroups re-derive the rows' visibility (as for row lists).  Same results. */
int32_t mi_debug_set_walk_inrow(mi_ctx* ctx, int32_t mode);
/* The all-dirty hierarchy frame of mi_propagate_and_cull[_views]: 0 = every tile also culls its own rows when the frame has one
 * view and nothing else needs the frame kernels (default: measured faster there), 1 = always tile launch + cull launch,
 * 2 = fused whenever it applies.  Results are identical. */
int32_t mi_debug_set_tree_cull(mi_ctx* ctx, int32_t mode);
/* The per-wave summary of Aabb / flags / RenderLayers (64 aligned rows that agree read 32 bytes instead of 64 x 29): 0 = in use
 * (default), 1 = off -- every row reads its own columns.  Results are identical; A/B timing and tests. */
int32_t mi_debug_set_row_summary(mi_ctx* ctx, int32_t mode);

/* The world-sphere path of the cull-only and changed-rows frames (kernels_flat.hip, k_frame_sph): 0 = used from the second frame in
 * a row that rewrites no or few GlobalTransforms (default), 1 = never, 2 = at once (the first such frame rebuilds the column). */
int32_t mi_debug_set_sphere_path(mi_ctx* ctx, int32_t mode);
/* The static cull order (kernels_cells.hip): cull-only frames of a scene that has gone static -- world spheres current, camera views,
 * MI_CULL_BEGIN_FRAME | MI_CULL_END_FRAME, no classes / ranges / exchange -- run over a cell-ordered copy whose waves are first tested
 * as a whole against each view (reject only).  0 = built by the second such frame in a row on contexts of 3 000 000 rows and more
 * (default), 1 = never, 2 = at once and at any row count, 3 = as 2 with the list kernels' runs as long as those of a table beyond
 * 16.7 M rows.  Results are identical.
 * mi_debug_static_cull_counts: orders built / frames that ran over one (tests). */
int32_t mi_debug_set_static_cull_order(mi_ctx* ctx, int32_t mode);
int32_t mi_debug_static_cull_counts(mi_ctx* ctx, uint32_t* out_builds, uint32_t* out_frames);
/* The cluster lists as the LAST FILL THAT RAN left them, without launching a pending one (mi_cluster_download would: with
 * MI_CULL_MORE_FRAMES frame f's fill rides in frame f + 1's launch and frame f + 1's own is still pending after it).  For the
 * test that checks what a riding fill wrote: out_offsets[C + 1], out_indices[capacity]; *out_total = out_offsets[C]. */
int32_t mi_debug_cluster_download_unjoined(mi_ctx* ctx, uint32_t* out_offsets, uint32_t* out_indices, uint64_t capacity, uint64_t* out_total);
/* The device's restatement of glibc logf (view_z_to_z_slice, crates/bevy_light/src/cluster/assign.rs:1057) over n inputs. */
int32_t mi_debug_logf(mi_ctx* ctx, const float* in, float* out, uint32_t n);

#ifdef __cplusplus
}
#endif
#endif /* BEVY_MI355X_DEBUG_H */
