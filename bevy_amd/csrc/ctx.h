// ctx.h -- internal to the library: the context object behind the opaque mi_ctx handle and the helpers the
// entry-point files (context.cpp, ctx_hierarchy.cpp, ctx_exchange.cpp, ctx_batch.cpp, ctx_cluster.cpp) share.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <numeric>
#include <chrono>
#include <thread>
#include <string>
#include <vector>

#include "../../include/bevy_mi355x.h"
#include "../../include/bevy_mi355x_debug.h"
#include "kernels.h"

namespace mi {
hipError_t set_cluster_lds_limit();
hipError_t launch_logf_probe(const float* in, float* out, uint32_t n, hipStream_t stream);
}  // namespace mi

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

struct ProfSpan {
    uint32_t kernel;
    hipEvent_t a, b;
};

struct mi_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err = "no error";

    // ---- columns ----
    uint32_t n = 0, cap = 0;
    float *t = nullptr, *r = nullptr, *s = nullptr, *g = nullptr, *c = nullptr, *h = nullptr;
    uint8_t *flags = nullptr, *vv = nullptr, *changed = nullptr, *g_changed_bytes = nullptr;
    uint32_t *layers = nullptr, *class_mask = nullptr;
    uint32_t* layers_hi = nullptr;  // RenderLayers 32..63: allocated by the first mi_upload_render_layers_hi (nullptr = no row has any)
    uint64_t *keys = nullptr, *g_chg_bits = nullptr, *vv_chg_bits = nullptr;
    uint8_t* tree_bytes = nullptr;  // TransformTreeChanged, a byte per row; two halves of tree_half_words 32-bit words (double-buffered by frame: tree_parity)
    uint32_t tree_half_words = 0, tree_parity = 0;
    bool tree_clean[2] = {true, true};  // the half is known to be all zero
    // The indexed uploads mark for the coming frame themselves (k_upload_trs_indexed) when the last propagate ran under the
    // static-scene rule: marks_live = it did; marks_in_cur = the current half holds marks of this frame's uploads (a subset of what
    // k_mark_dirty would set: valid, never to be cleared before use); marks_complete = every change since the last propagate was
    // marked that way, so mi_propagate needs no mark launch; marks_other_cleared = an upload of this frame zeroed the other half
    bool marks_live = false, marks_in_cur = false, marks_complete = false, marks_other_cleared = false;
    bool have_class_mask = false, have_keys = false, have_changed = false;
    uint32_t propagated_rows = 0;  // rows [0, propagated_rows) have been through a propagate (their Added<GlobalTransform> is consumed)
    uint32_t classes_present = 1u;
    std::vector<uint64_t> h_keys;
    bool order_dirty = false, order_identity = true;
    DevBuf order;
    float* range = nullptr;        // VisibilityRange (start_margin.start, end_margin.end) per row
    bool have_ranges = false;      // a VisibleEntityRanges resource exists (mi_upload_visibility_ranges was called)
    uint8_t* visibility = nullptr; // Visibility component: 0 Inherited, 1 Hidden, 2 Visible, 0x80 none
    uint8_t* inh_changed = nullptr;  // InheritedVisibility assigned by the last mi_visibility_propagate (bytes)
    DevBuf inh_bits, sparse_cnt, sparse_rows, sparse_total, sparse_g;

    // ---- staging ----
    void* stage = nullptr;
    size_t stage_bytes = 0, stage_used = 0;
    uint64_t stage_epoch = 0;  // bumped whenever the arena wraps or moves: pointers into it from before are stale

    // ---- upload windows (mi_map_upload_window): pinned chunks of their own -- several windows may be open at once, and a window
    // must not move or be recycled while it is (the staging arena above wraps).  Recycled when a window is mapped while none is open.
    struct WinChunk { void* p; size_t bytes, used; };
    std::vector<WinChunk> win_chunks;
    uint32_t win_open = 0;
    uint64_t win_gen = 1;  // mi_upload_window::token of the windows handed out since the last recycling

    // ---- hierarchy ----
    uint32_t n_levels = 1;
    std::vector<uint32_t> level_offsets;  // n_levels + 1
    DevBuf parent_idx, node_flags, tiles;
    std::vector<std::pair<uint32_t, uint32_t>> passes;  // (first tile, n tiles); pass 0 starts at level 0 (roots)
    struct TileGroup { uint32_t first, count, n_chain, owner_rows; bool deep = false; /* some tile spans more than TILE_FAST_LEVELS levels */ };
    bool narrow = false;            // every level is at most a wave wide and there are more levels than a tile spans: one wave walks the hierarchy (k_propagate_narrow)
    bool narrow_quad = false;       // ... and no level holds more than 16 rows
    DevBuf level_offs_dev;          // level_offsets on the device (that kernel reads them)
    bool wave_forest = false;       // a forest every tree of which fits a wave tile: a wave per tile walks it (k_propagate_wave_tiles)
    bool wave_quad = false;         // ... and no level of any tile holds more than 16 rows
    uint32_t n_wtiles = 0;
    DevBuf wtiles;                  // the wave tiles (TileDesc, kind TILE_ROOTS)
    bool strip_plan = false;        // the hierarchy is walked by strips: one launch of independent waves (k_propagate_strips)
    uint32_t n_strips = 0, n_strip_rounds = 0, strip_bands = 0;
    DevBuf strips, strip_rounds;    // StripDesc per strip, the flat table of StripRound
    bool by_levels = false;         // the row count overflows the tile kernel's 32-bit offsets (or mi_debug_set_tile_mode(1)): mi_propagate sweeps level by level
    DevBuf anc;                     // the ancestor table (kernels.h, ANC_DEPTH): built by mi_upload_hierarchy; in use while anc_valid
    bool anc_valid = false;
    const mi::TreeCull* tcull = nullptr;  // set by the fused hierarchy frame around its mi_propagate: the tile launches also cull
    int32_t tree_cull_mode = 0;           // mi_debug_set_tree_cull: 0 = fused where it applies and there is one view (default), 1 = never, 2 = whenever it applies
    int32_t tile_pretest_mode = 0;  // mi_debug_set_tile_pretest
    int32_t tile_mode = 0;          // 0 = tiles where they fit, 1 = always level by level, 2 = as 0, 3 = as 0 with the streamed-level thresholds at their test values (mi_debug_set_tile_mode)
    std::vector<TileGroup> groups;  // tile launches of mi_propagate: roots + chain bands in one, then one per dependent band
    std::vector<std::pair<uint32_t, uint32_t>> stream_levels;  // (start, count), top-down: the wide deepest levels, one streaming launch each
    DevBuf tree_trace;              // mi_debug_tree_trace: 8 timestamps per tile of the light tile kernel
    DevBuf chains, snap;            // snap: 2 x snap_rows x 48 B, pre-frame GlobalTransforms of the owner rows (see kernels_tree.hip)
    uint32_t snap_rows = 0, snap_parity = 0;
    bool snap_valid = false;
    bool have_hierarchy = false;
    bool g_chg_in_bytes = false;  // the GlobalTransform change mask currently lives in g_changed_bytes (tree path)
    // host-side knowledge that lets a frame with no dirty Transform skip its launches: some byte of `changed` may be
    // non-zero (set by the uploads that mark rows, cleared when mi_propagate consumes the column); the change masks of
    // the last propagate may hold set bits
    bool changed_maybe = true, g_chg_maybe = true;
    // rows marked changed since the last propagate consumed the column, when every mark came through mi_upload_transforms_indexed /
    // mi_commit_upload_window (UINT64_MAX: a bulk mi_upload_changed or new rows -- unknown)
    uint64_t changed_rows_hint = UINT64_MAX;
    // The change column holds stamps (row_changed(), kernels.h): what the indexed uploads write is the current generation, consuming
    // the column is `++changed_gen`.  changed_bulk: some byte may hold the plain 1 of a bulk upload / a fresh row -- those need the memset.
    uint32_t changed_gen = 2;
    bool changed_bulk = true;

    // ---- world-sphere column (k_frame_sph): (affine * aabb.center, |M3 * half_extents|) per row, the bounding sphere
    // check_visibility tests first.  Valid relative to the GlobalTransform / bounds columns as `sph_state` says.
    enum SphState : uint32_t {
        SPH_INVALID = 0,        // nothing is known to be current
        SPH_EXCEPT_CHANGED = 1, // current but for the rows the GlobalTransform change mask of the last propagate flags
        SPH_VALID = 2
    };
    DevBuf sph;
    uint32_t sph_state = SPH_INVALID;
    uint32_t sph_quiet = 0;  // cull frames since the last wholesale GlobalTransform rewrite (the column is rebuilt on the second)
    int32_t sph_mode = 0;    // mi_debug_set_sphere_path: 0 = as described, 1 = never, 2 = rebuild at once

    // ---- static cull order (kernels_cells.hip; k_frame_cells): a cell-ordered copy of a scene that has gone static.  `valid` = the
    // copies (world spheres, GlobalTransforms, ViewVisibility mirror, summaries) agree with the columns: cleared by whatever writes
    // ViewVisibility other than k_frame_cells, by whatever writes flags / RenderLayers / bounds (row_summary_touch), by a resize, and by
    // every frame that takes another kernel (any GlobalTransform change leaves sph_state != SPH_VALID, which sends the next frame
    // there).  Built by the second eligible frame in a row that finds none (cells_frame_eligible, context.cpp).
    struct Cells {
        DevBuf perm, sph_s, g_s, vv_s, sum_a, sum_b, sum_h, state, keys_a, keys_b, vals_a, vals_b, sort_tmp, minmax;
        DevBuf work, work_n;         // the frame's work list (k_cells_test -> k_frame_cells) and its two alternating counters
        uint32_t work_parity = 0;
        DevBuf pass_s, fin_scratch;  // per slot: its row's bits in the masks of the last frame over the order; k_cells_blocks' prefix / totals
        // the frame that continues the one before: k_cells_blocks copied that frame's masks into the set this frame writes
        bool chain_ok = false;
        const void* chain_mask = nullptr;
        uint64_t chain_words = 0;
        uint32_t chain_views = 0;
        bool valid = false;
        uint32_t n_waves = 0;
        uint32_t quiet = 0;          // eligible frames in a row that found no valid order
        int32_t mode = 0;            // mi_debug_set_static_cull_order: 0 = as described, 1 = never, 2 = at once and at any row count
        uint32_t min_rows = 3000000; // the frame over the order is four short launches (~22 us at 1 M rows against k_frame_sph's 10; even at 4 M x 1 view, far ahead at 10 M)
        uint32_t builds = 0, frames = 0;  // mi_debug_static_cull_counts
        // MI_CULL_MORE_FRAMES: a frame's lists (k_cells_lists' work) are not launched behind it but ride in the next such frame's
        // k_frame_cells launch; cells_lists_join (every frame of another kind, everything that exposes the lists) launches them on
        // their own.  They read that frame's masks, its block prefixes (fin_scratch) and write its own list buffers.
        bool lists_pending = false;
        mi::CellsFinishArgs lists_args{};
        uint32_t lists_views = 0;
    } cells;

    // ---- dense uploads in pieces, GlobalTransforms ahead of the frame (context.cpp: mi_commit_upload_window,
    //      mi_download_frame_results; profiles/r03_experiments.md 12) ----
    // A SEQUENCE is a run of dense windows that carries the whole flat table in ascending order: one window for every row (split
    // into SPLIT pieces here) or several the caller commits one after the other (first_row = where the one before ended).  Each
    // piece goes out on a stream of its own (up_stream), the context's stream waits for it, and -- once the caller has shown that
    // it fetches every GlobalTransform of such frames -- computes the piece's GlobalTransforms at once (k_globals_ahead, into g_pre:
    // From(Transform), what the all-rows frame will write) and sends them home on a third stream (dn_stream) while the later
    // pieces of the upload are still arriving: PCIe is full duplex.  mi_download_frame_results hands out what arrived when the
    // frame in between was an all-rows one of the same Transforms (trs_version), and fetches as usual otherwise.
    static constexpr uint32_t UP_PIECES = 16, SPLIT = 8;
    hipStream_t up_stream = nullptr, dn_stream = nullptr;
    hipEvent_t ev_up[UP_PIECES] = {}, ev_pre[UP_PIECES] = {};
    bool piece_streams = false;
    uint32_t seq_cov = 0;            // the sequence so far carries rows [0, seq_cov); 0 = none in progress (every ENTER ends one)
    uint32_t seq_pieces = 0;         // events of this sequence in use
    bool seq_ahead = false;          // this sequence computes and fetches GlobalTransforms ahead
    uint64_t trs_version = 1;        // bumped by whatever writes the Transform columns, renumbers rows or changes their count
    uint64_t pre_version = 0;        // trs_version for which g_host holds (or is receiving) every row's GlobalTransform; 0 = none
    uint64_t frame_all_version = 0;  // trs_version of the last frame that rewrote every GlobalTransform of a flat table
    uint32_t pre_n = 0;
    bool pre_used = false;           // the last fetch ahead was handed out
    bool ahead_wanted = false;       // the caller fetched every GlobalTransform after its last all-rows frame
    DevBuf g_pre;                    // device: [cap] GlobalTransforms ahead of the frame
    void* g_host = nullptr;          // pinned: where they land (memory of its own: valid until the next call, like the arena)
    size_t g_host_bytes = 0;
    uint32_t* iota_host = nullptr;   // pinned 0, 1, 2 ...: the changed-row list of a frame in which every row changed (nothing to fetch)
    size_t iota_rows = 0;
    uint32_t n_piece_uploads = 0, n_ahead_downloads = 0;  // mi_debug_chunked_counts
    // The same for the changed rows of an indexed upload window (the changed-rows frame of a flat table): the scatter kernel that reads
    // the window over PCIe writes each entry's GlobalTransform straight back into pinned memory (k_upload_trs_indexed, g_ahead);
    // mi_download_frame_results hands rows and GlobalTransforms out without compacting, gathering or fetching anything when the
    // change mask the last frame left is exactly that upload's rows: the change column was clean before it, nothing raised a mark
    // or wrote a Transform between it and the frame, the rows were strictly ascending, and no propagate ran since.
    uint64_t marks_serial = 1;          // bumped by whatever raises change marks
    void* gs_host = nullptr;            // pinned, device-mapped: [gs_k] GlobalTransforms in upload order
    size_t gs_host_bytes = 0;
    const uint32_t* gs_rows = nullptr;  // the window's rows (pinned; stays put until a later mi_map_upload_window recycles the windows) ...
    std::vector<uint32_t> gs_rows_rev;  // ... or, for a window whose rows descend, their reversed copy
    uint32_t gs_k = 0;                  // 0 = the last indexed upload wrote nothing ahead
    uint64_t gs_marks_serial = 0, gs_trs_version = 0;  // marks_serial / trs_version right after that upload (the former until a propagate consumes the marks)
    uint64_t gs_frame_serial = 0;       // gs_marks_serial as the current propagate call found it
    bool gs_frame_ok = false;           // the last propagate of any kind was the changed-rows frame of a flat table over exactly those marks
    bool gs_used = false, sparse_ahead_wanted = false;
    uint32_t n_sparse_ahead_downloads = 0;
    int32_t chunk_mode = 0;          // mi_debug_set_chunked_frames: 0 = tables of >= 262144 rows (default), 1 = never, 2 = any size

    // ---- row summary (RowSummary, kernels.h): Aabb / flags / RenderLayers per 64 rows where they are uniform.  Derived from the
    // columns by k_row_summary; rs_lo / rs_hi = the waves [lo, hi) whose summary is out of date, per part (0 = Aabb, 1 = flags +
    // layers).  The frame entry points bring it up to date first (row_summary_ensure); columns_of hands it to kernels only then.
    DevBuf row_sum;
    uint32_t rs_lo[2] = {0, 0}, rs_hi[2] = {0, 0};
    int32_t walk_inrow_mode = 0;  // mi_debug_set_walk_inrow: 0 = row-range bindings are walked by the frame kernel's own row workgroups, 1 = never
    int32_t row_sum_mode = 0;  // mi_debug_set_row_summary: 0 = in use, 1 = off (every row reads its own Aabb / flags / layers)

    // ---- views / visibility ----
    DevBuf views;
    uint32_t n_views = 0;
    uint64_t words_per_view = 0;
    // multi-GPU exchange (mi_exchange_configure): in-place all-gather of the masks after every cull
    struct Exchange {
        bool on = false;
        bool simple = true;      // MI_EXCHANGE_SIMPLE (default) or MI_EXCHANGE_GROUPED / MI_EXCHANGE_PIPELINED
        bool grouped = false;    // MI_EXCHANGE_GROUPED: the frame calls leave the all-gather pending for mi_exchange_group_flush
        // MI_EXCHANGE_SIMPLE (not grouped): ncclAllGather is enqueued by the library's exchange thread instead of the caller's -- RCCL's
        // enqueue path costs 15 - 20 us of CPU per call, more than the rest of the frame call; the ordering stays plain events (the
        // caller records "kernels done" and hands the slot over; the thread does wait / all-gather / record on the communication
        // stream).  MI_XCH_SYNC_ENQUEUE=1 (read at mi_ctx_create) keeps the call on the caller's thread.
        bool async_enqueue = true;
        bool group_pending = false;
        uint32_t group_slot = 0;
        bool debug = false;      // MI_XCH_DEBUG, read once at mi_ctx_create
        hipEvent_t ev_kernels[8] = {nullptr};  // simple mode: recorded behind each frame's kernels on the compute stream
        int (*all_gather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;  // ncclAllGather
        // Up to MAX_COMMS communicators, used round-robin by frame, each on its own stream: the all-gathers of
        // consecutive frames are then in flight together (a ~125 KB all-gather over 8 GPUs is pure latency, and one
        // communicator runs its collectives strictly one after the other).  Every rank issues them in frame order.
        static constexpr uint32_t MAX_COMMS = 4;
        uint32_t n_comms = 0;
        void* comm[MAX_COMMS] = {nullptr};
        static constexpr uint32_t MAX_BUFS = 8;
        uint32_t n_bufs = 0;
        void* buf[MAX_BUFS] = {nullptr};
        void* owned[MAX_BUFS] = {nullptr};  // mi_exchange_configure_owned: buffers the library allocated (freed on reconfigure / destroy)
        size_t owned_bytes = 0;             // ... and the size of each (world * block_bytes): what mi_exchange_download may read; 0 = the caller's buffers
        uint64_t words_per_view = 0, word_offset = 0, block_bytes = 0;
        uint32_t rank = 0;
        uint64_t frame = 0;
        hipStream_t comm_stream[MAX_COMMS] = {nullptr};
        bool comm_shares_queue[MAX_COMMS] = {false};  // shares the compute stream's hardware queue (pick_side_streams)
        hipEvent_t ev_gathered[MAX_BUFS] = {nullptr};  // recorded behind each buffer's all-gather (mi_exchange_last waits on it)
        // The collective is enqueued by a library-owned host thread: RCCL's enqueue path costs tens of
        // microseconds of CPU per call, which would otherwise sit in the frame's critical path on the caller's
        // thread.  The caller's thread never runs more than two frames ahead of it.
        std::thread worker;
        std::mutex m;
        std::condition_variable cv;
        struct Job {
            uint32_t slot;     // gathered buffer of the frame
            uint32_t* flag;    // device word to wait on ...
            uint32_t value;    // ... until it is >= value: "this frame's masks are complete"
        };
        std::deque<Job> queue;
        uint32_t* wait_flag = nullptr;  // how the current frame announces its masks (set while the frame is enqueued)
        uint32_t wait_value = 0;
        bool job_deferred = false;      // ... in the next frame's launch: exchange_end must not queue the all-gather yet
        std::atomic<uint64_t> submitted_fast{0};  // == submitted, readable without the lock (the thread polls it)
        std::atomic<bool> sleeping{false}, stop_fast{false};
        bool stop = false;
        uint64_t submitted = 0, issued = 0;  // guarded by m
        uint64_t worker_frames = 0;          // exchange thread only
        int worker_error = 0;
        volatile uint32_t* done_flag = nullptr;  // pinned host words [MAX_COMMS]: all-gathers completed on each communicator
        // device word: number of frames whose masks are complete.  Written by the compaction kernel itself (see
        // CompactFastArgs::signal) or, when that kernel is not the one running, by a write-value packet behind the
        // frame's kernels; the communication stream waits on it with hipStreamWaitValue32.
        uint32_t* kernels_flag = nullptr;
        bool kernel_signal = true;      // MI_XCH_NO_KERNEL_SIGNAL forces the packet
        bool signalled = false;         // this frame's compaction launch carries the signal
        double dbg_wait_ns = 0, dbg_begin_ns = 0, dbg_end_ns = 0, dbg_worker_ns = 0;  // MI_XCH_DEBUG
    } xch;
    void* ext_bitmask = nullptr;
    uint64_t ext_words_per_view = 0, ext_word_offset = 0;
    bool culled = false;
    mi::ViewSet view_set{};      // views passed by value when n_views <= mi::MAX_INLINE_VIEWS
    bool views_inline = false;
    // compaction
    DevBuf block_counts, seg_bases, out_keys;
    // What one cull frame leaves behind: the view masks (unless bound elsewhere), the by-products the compaction
    // consumes and the lists it writes.  Two sets, alternating by frame: a frame culled with MI_CULL_MORE_FRAMES defers
    // its compaction into the tail workgroups of the NEXT frame's kernel (or into a launch of its own at the first
    // entry point that exposes the lists), which reads set f while that kernel writes set f + 1.
    // Three sets, taken in turn (frame_begin): the frame being written, the previous frame's (whose deferred compaction may still
    // read it), and -- for the hierarchy frames that OR their results in with atomics (tree_frame_fused) -- the next frame's, which
    // this frame's launch zeroes.  fb_zeroed[i]: masks and wave counts of set i are known to be zero for the shape in fb_zero_shape[i].
    static constexpr uint32_t N_FB = 3;
    struct FbZero {
        void *bitmask = nullptr, *wave_cnt = nullptr;  // what was zeroed (a reallocated buffer is not it any more)
        uint64_t bitmask_words = 0, wave_cnt_bytes = 0;
        bool ok = false;
    } fb_zero[3];
    FbZero fb_zero_taken;  // state of the set frame_begin has just made current (every frame writes into its set: the state is consumed)
    uint64_t* vv_chg_alt = nullptr;  // second ViewVisibility change-tick buffer of those frames (they alternate); zeroed iff vv_alt_zeroed
    bool vv_alt_zeroed = false;
    struct FrameBufs {
        DevBuf bitmask, wave_cnt, seg_mask, out_rows, seg_totals;
        uint64_t seg_totals_layout = 0;  // (segments, chunks) of the last compaction that used seg_totals: where its stamp table lay
    } fb[N_FB];
    uint32_t cur = 0;  // set of the current / last frame
    struct DeferredCompaction {
        bool pending = false;
        mi::CompactFastArgs args{};  // everything the compaction of the last frame needs
        bool has_job = false;        // with the exchange on: the frame's all-gather, queued once the compaction is submitted
        Exchange::Job job{};
    } defer;
    uint32_t compact_tag = 0;  // stamp of the chunk totals of the hierarchical compaction (CompactFastArgs::tag): a new one per compaction
    uint32_t compact_views = 0, compact_classes = 0;
    uint32_t class_bits[32] = {0};
    bool compact_fast = false;   // last compaction used the single-launch path (out_rows strided per segment)
    uint64_t seg_stride = 0;

    // ---- clustering ----
    // view_z_to_z_slice evaluates ln() with a restatement of glibc's logf (cluster_walk.h); the host's own logf is what a Bevy built
    // here would call (bevy_math::ops::ln -> std -> libm).  Compared once, on the first perspective view: 0 = not yet, 1 = agree,
    // 2 = they differ (musl, another glibc, a vector libm): the cluster entry points then refuse, and the shim's stock system takes over
    uint32_t libm_state = 0;
    DevBuf cl_pos, cl_type, cl_layers, cl_layers_hi, cl_dir, cl_sincos, cl_planes, cl_spheres;
    // batching work-item build (kernels_batch.hip)
    uint32_t *bt_set = nullptr, *bt_bin = nullptr, *bt_input = nullptr, *bt_row_meta = nullptr;  // per-row columns
    uint8_t* bt_kind = nullptr;                              // MI_BATCH_ROW_*
    uint32_t *bt_cpu_bin = nullptr, *bt_bucket = nullptr;    // unbatchable / batchable bin; resolved bucket
    std::vector<uint8_t> bt_unb_indexed, bt_bat_indexed, bt_set_indexed_host;
    std::vector<uint32_t> bt_meta_zero;                      // the uploaded GpuBinMetadata with instance_count = 0
    DevBuf bt_bucket_desc, bt_meta_out, bt_inst[2], bt_plan, bt_unb, bt_items, bt_batches, bt_sorted_partials;
    uint32_t bt_sorted_one_wg_limit = mi::SORTED_ONE_WG_ITEMS;  // mi_debug_set_sorted_one_wg_limit
    uint32_t bt_inst_cur = 0;
    bool bt_desc_dirty = true, bt_desc_no_indirect = false, bt_last_sorted = false;
    bool bt_resolve = true;  // rows or tables changed: bt_row_meta must be recomputed
    DevBuf bt_set_indexed, bt_table_off, bt_table, bt_meta_off, bt_meta, bt_rows_a, bt_rows_b, bt_hist, bt_set_count, bt_set_scan,
        bt_counters, bt_wi[2], bt_md[2], bt_bs[2], bt_records, bt_totals;
    uint32_t bt_n_sets = 0, bt_n_meta = 0;
    bool bt_have_rows = false, bt_have_sets = false, bt_built = false;
    DevBuf cl_remap, cl_bind_oc, cl_bind_idx, cl_block_counts, cl_pair_cb, cl_pair_mask, cl_acc, cl_offsets, cl_indices, cl_scalars;
    // derive mode of an assignment that runs in or next to the frame kernel: which rows that frame's propagate writes (cull_frame sets them)
    const uint8_t* cl_derive_changed = nullptr;
    bool cl_derive_resident = false;
    uint32_t cl_parity = 0, cl_acc_clusters = 0, cl_acc_blocks = 0;  // cl_parity: the current working set; cl_acc = 3 x [counts 6C | totals C | farthest_z + pad]
    uint32_t cl_n = 0;
    bool cl_have_type = false, cl_have_layers = false, cl_have_layers_hi = false, cl_have_spot = false, cl_have_spot_dir = false, cl_any_spot = false;
    bool cl_rows_bound = false;      // mi_cluster_bind_objects_to_rows: object i is row cl_first_row + i
    uint32_t cl_first_row = 0;
    DevBuf cl_row_list;              // mi_cluster_bind_objects_to_row_list: the rows, in object order
    bool cl_rows_listed = false;
    std::vector<float> cl_host_planes, cl_host_spheres;  // storage of the view mi_cluster_assign_frame builds
    // The view's cluster planes: the z planes follow the camera's scale by an ulp from frame to frame, and a device copy would put
    // an H2D blit (~6 us) in front of every frame.  Tables of up to WALK_PLANES_MAX floats (16 x 9 x 24: 208) travel as a trailing
    // kernel argument of the launch that walks (WalkPlanes, kernels.h: the argument segment is device memory); bigger ones are read
    // from the pinned staging arena (mapped host memory, a trip over PCIe per walking workgroup -- which is what every table did
    // until round 5, 1.1 us of the metric frame).  cl_planes_host is the current table, cl_planes_epoch the arena epoch its
    // staged copy belongs to.
    std::vector<float> cl_planes_host, cl_spheres_sent;
    uint64_t cl_planes_epoch = ~0ull;
    uint32_t cl_plane_counts[3] = {0, 0, 0};
    // The cluster kernels of a MI_CULL_WITH_CLUSTERS frame run on a stream of their own, concurrently with the frame kernel.
    hipStream_t cl_stream = nullptr;
    hipEvent_t ev_cl_done = nullptr, ev_cl_inputs = nullptr;
    bool cl_on_side = false;         // an assignment is (possibly) still running on cl_stream: join before touching its inputs / outputs
    bool cl_inputs_dirty = true;     // the main stream has written something the cluster kernels read since they last waited for it
    // The fill of a MI_CULL_WITH_CLUSTERS | MI_CULL_MORE_FRAMES frame is deferred like the VisibleEntities compaction: it rides in
    // extra workgroups of the next frame's kernel, or cluster_fill_join launches it on its own at the first entry point that
    // needs the lists.
    bool cl_fill_pending = false;
    mi::ClusterFillJob cl_fill_job{};
    mi::ClusterViewDev cl_view{};
    bool cl_have_view = false, cl_assigned = false;

    // ---- several clustered views (mi_cluster_select_view): everything above that belongs to ONE view -- its constants and plane
    // tables, the working sets and outputs of its assignment, its pending fill, its binding arrays -- lives in the context's fields for
    // the SELECTED slot and in cl_parked[k] for the others; selecting swaps.  The objects and their row binding are shared.
    struct ClusterSlot {
        DevBuf planes, spheres, remap, bind_oc, bind_idx, block_counts, pair_cb, pair_mask, acc, offsets, indices, scalars;
        uint32_t parity = 0, acc_clusters = 0, acc_blocks = 0;
        std::vector<float> host_planes, host_spheres, planes_host, spheres_sent;
        uint64_t planes_epoch = ~0ull;
        uint32_t plane_counts[3] = {0, 0, 0};
        bool fill_pending = false;
        mi::ClusterFillJob fill_job{};
        mi::ClusterViewDev view{};
        bool have_view = false, assigned = false;
    };
    ClusterSlot cl_parked[MI_CLUSTER_MAX_VIEWS];
    uint32_t cl_slot = 0;  // the selected one (its ClusterSlot entry is unused while it is selected)

    // ---- timing ----
    hipEvent_t timer_a = nullptr, timer_b = nullptr;
    bool profiling = false;
    uint64_t prof_mask = ~0ull;
    uint32_t prof_every = 1, prof_tick[mi::K_NUM_KERNELS] = {0};  // time every n-th launch of a kernel
    uint32_t prof_burst = 0, prof_timed[mi::K_NUM_KERNELS] = {0};  // ... and at most the first prof_burst of them (0 = no limit)
    std::vector<ProfSpan> spans;
    bool span_open = false;
    uint64_t prof_launches[mi::K_NUM_KERNELS] = {0};
    double prof_ms[mi::K_NUM_KERNELS] = {0};
};


namespace mi_detail {
using namespace mi;

int32_t fail(mi_ctx* ctx, int32_t code, const char* fmt, ...);

#define HIP_TRY(ctx, expr)                                                                                   \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            return fail(ctx, e_ == hipErrorOutOfMemory ? MI_ERR_OUT_OF_MEMORY : MI_ERR_DEVICE, "%s: %s (%s:%d)", #expr, \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                          \
    } while (0)

// The same as a value, for a caller that has something to give back before it returns (cells_frame): MI_OK, or the recorded failure.
inline int32_t hip_rc(mi_ctx* ctx, hipError_t e, const char* what) {
    if (e == hipSuccess) return MI_OK;
    return fail(ctx, e == hipErrorOutOfMemory ? MI_ERR_OUT_OF_MEMORY : MI_ERR_DEVICE, "%s: %s", what, hipGetErrorString(e));
}

// ENTER_RAW: mi_map_upload_window / mi_commit_upload_window, which continue a sequence of dense windows (seq_cov above); ENTER:
// everybody else -- whatever they launch may read the Transform columns, so a window committed after them starts over on the
// context's stream (or starts a new sequence at row 0, which waits for that stream first)
// (statements of the entry point's own scope, not a block: the range object lives until the entry point returns)
#define ENTER_RAW(ctx)                                                                 \
    if (!(ctx)) return mi_detail::fail(nullptr, MI_ERR_INVALID_ARG, "ctx is NULL");    \
    if (mi_detail::trace_on()) fprintf(stderr, "[mi] %s\n", __func__);                 \
    mi_detail::TraceRange mi_trace_range_(__func__);                                   \
    HIP_TRY(ctx, hipSetDevice((ctx)->device))
#define ENTER(ctx)  \
    ENTER_RAW(ctx); \
    (ctx)->seq_cov = 0

// Tracing hooks (the reference wraps these systems in info_span!, crates/bevy_transform/src/systems.rs:169-283): every
// entry point of the library is a roctx range named after itself, so a rocprofv3 --marker-trace (or any roctx consumer) of a Bevy
// frame shows mi_propagate_and_cull_views, mi_download_frame_results, ... around the kernels they launch.  No link-time
// dependency: roctxRangePushA / roctxRangePop are looked up at first use -- among the symbols already loaded (rocprofv3 preloads
// librocprofiler-sdk-roctx.so), then in librocprofiler-sdk-roctx.so / libroctx64.so.  On when MI_ROCTX=1, or when rocprofv3 asked for
// marker tracing (ROCPROF_MARKER_API_TRACE) and MI_ROCTX is not 0; otherwise an entry point pays one predictable branch.
struct RoctxApi {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
};
const RoctxApi& roctx_api();
struct TraceRange {
    const RoctxApi& api;
    explicit TraceRange(const char* name) : api(roctx_api()) {
        if (api.push) api.push(name);
    }
    ~TraceRange() {
        if (api.push) api.pop();
    }
    TraceRange(const TraceRange&) = delete;
    TraceRange& operator=(const TraceRange&) = delete;
};

inline bool trace_on() {
    static const bool on = getenv("MI_TRACE") != nullptr;  // every entry point announces itself on stderr
    return on;
}

inline uint64_t words64(uint32_t n) { return ((uint64_t)n + 63u) / 64u; }
// bitmask words written by a launch of ceil(n/256) workgroups x 4 waves
inline uint64_t padded_words(uint32_t n) { return (((uint64_t)n + 255u) / 256u) * 4u; }

int32_t ensure(mi_ctx* ctx, DevBuf& b, size_t bytes);
template <typename T>
int32_t grow_column(mi_ctx* ctx, T*& col, size_t elems_per_row, uint32_t old_rows, uint32_t new_cap, int fill_byte) {
    T* np = nullptr;
    const size_t bytes = (size_t)new_cap * elems_per_row * sizeof(T) + 256;
    HIP_TRY(ctx, hipMalloc((void**)&np, bytes));
    HIP_TRY(ctx, hipMemsetAsync(np, fill_byte, bytes, ctx->stream));
    if (col && old_rows)
        HIP_TRY(ctx, hipMemcpyAsync(np, col, (size_t)old_rows * elems_per_row * sizeof(T), hipMemcpyDeviceToDevice, ctx->stream));
    if (col) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipFree(col));
    }
    col = np;
    return MI_OK;
}

// Pinned staging arena: host slices are copied here (the ECS owns them only for the call) and the
// H2D copy runs asynchronously on the context's stream.
int32_t stage_alloc(mi_ctx* ctx, size_t bytes, void** out);
int32_t upload(mi_ctx* ctx, void* dst, const void* src, size_t bytes);
int32_t download(mi_ctx* ctx, void* dst, const void* src, size_t bytes);
int32_t check_rows(mi_ctx* ctx, uint32_t first, uint32_t n, const char* what);
void row_summary_touch(mi_ctx* ctx, uint32_t parts, uint32_t first_row, uint32_t n_rows);  // the columns of these rows were written
int32_t row_summary_ensure(mi_ctx* ctx);
void trs_written(mi_ctx* ctx);  // the Transform columns (or the numbering / count of rows) changed: nothing fetched ahead applies any more
int32_t consume_changed(mi_ctx* ctx);  // the propagate has read the change column: every row is unchanged from here on
void prof_close(mi_ctx* ctx);
void prof_mark(void* vctx, uint32_t kernel);
void prof_collect(mi_ctx* ctx);
// Times exactly one launch (the next MI_LAUNCH on this thread) with its dispatch timestamps.
struct ProfScope {
    mi_ctx* ctx;
    LaunchTimer lt{};
    bool armed = false;
    ProfScope(mi_ctx* c, uint32_t k) : ctx(c) {
        if (!c->profiling || k >= K_NUM_KERNELS || !((c->prof_mask >> k) & 1ull)) return;
        if (c->prof_every > 1 && (c->prof_tick[k]++ % c->prof_every) != 0) return;
        if (c->prof_burst && c->prof_timed[k] >= c->prof_burst) return;
        ++c->prof_timed[k];
        prof_close(c);
        ProfSpan sp;
        sp.kernel = k;
        hipEventCreate(&sp.a);
        hipEventCreate(&sp.b);
        c->spans.push_back(sp);
        lt.start = sp.a;
        lt.stop = sp.b;
        g_launch_timer = &lt;
        armed = true;
    }
    ~ProfScope() {
        if (armed && g_launch_timer == &lt) {  // nothing was launched inside the scope
            g_launch_timer = nullptr;
            hipEventDestroy(ctx->spans.back().a);
            hipEventDestroy(ctx->spans.back().b);
            ctx->spans.pop_back();
        }
    }
};
Columns columns_of(mi_ctx* ctx);
int32_t prepare_views(mi_ctx* ctx, const mi_view* views, uint32_t n_views, VisibilityOut* out);
int32_t prepare_segments(mi_ctx* ctx, uint32_t n_views, SegOut* seg);
int32_t run_compaction(mi_ctx* ctx, const VisibilityOut& vo, const SegOut& seg, uint32_t flags = 0);
// start of a cull frame: switches to the other buffer set and hands out the previous frame's deferred compaction (to
// ride in this frame's launch); returns whether there is one
bool frame_begin(mi_ctx* ctx, mi::CompactFastArgs* prev, bool* prev_has_job, mi_ctx::Exchange::Job* prev_job);
int32_t compaction_join(mi_ctx* ctx);   // enqueues a deferred compaction now: the lists are complete in stream order afterwards
// ctx_cluster.cpp
int32_t cluster_join(mi_ctx* ctx);      // orders the main stream behind an assignment running on the cluster stream
int32_t cluster_assign_launch(mi_ctx* ctx, bool concurrent_with_frame, uint64_t* out_total, bool defer_fill = false);
int32_t cluster_fill_join(mi_ctx* ctx);  // enqueues a deferred fill now
int32_t cluster_ride_prepare(mi_ctx* ctx, mi::ClusterWalkJob* job, bool* can_ride);
// ctx_exchange.cpp
int32_t pick_side_streams(mi_ctx* ctx, hipStream_t* out, uint32_t n_keep, bool* out_shares_queue);
int32_t exchange_begin(mi_ctx* ctx);
int32_t exchange_end(mi_ctx* ctx);
void exchange_push(mi_ctx* ctx, const mi_ctx::Exchange::Job& job);  // hands one frame's all-gather to the exchange thread
int32_t exchange_wait_issued(mi_ctx* ctx, uint64_t upto);
void exchange_stop(mi_ctx* ctx);

}  // namespace mi_detail
