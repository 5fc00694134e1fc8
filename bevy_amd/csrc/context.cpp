// context.cpp -- implementation of the C ABI in include/bevy_mi355x.h on top of the gfx950 kernels.
//
// Owns: the device-resident component columns (SoA across components, packed within a component),
// a pinned staging arena for host->device copies, the hierarchy tile plan, the per-view constants,
// the visibility bitmasks / VisibleEntities lists, the clustering scratch, and HIP-event timing.
// Everything is enqueued on one HIP stream per context; the only host synchronisations are in the
// download / timer-read entry points.
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
#include "kernels.h"

namespace mi {
hipError_t set_cluster_lds_limit();
hipError_t launch_logf_probe(const float* in, float* out, uint32_t n, hipStream_t stream);
}  // namespace mi

using namespace mi;

thread_local const mi::LaunchTimer* mi::g_launch_timer = nullptr;

namespace {

std::mutex g_err_mutex;
std::string g_create_error = "no error";

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

struct ProfSpan {
    uint32_t kernel;
    hipEvent_t a, b;
};

}  // namespace

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
    uint64_t *keys = nullptr, *g_chg_bits = nullptr, *vv_chg_bits = nullptr;
    uint32_t* tree_bits = nullptr;
    bool have_class_mask = false, have_keys = false, have_changed = false;
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

    // ---- hierarchy ----
    uint32_t n_levels = 1;
    std::vector<uint32_t> level_offsets;  // n_levels + 1
    DevBuf parent_idx, node_flags, tiles;
    std::vector<std::pair<uint32_t, uint32_t>> passes;  // (first tile, n tiles); pass 0 starts at level 0 (roots)
    struct TileGroup { uint32_t first, count, n_chain, owner_rows; };
    std::vector<TileGroup> groups;  // launches of mi_propagate: an owner pass + at most one pass of chain tiles
    DevBuf chains, snap;            // snap: 2 x snap_rows x 48 B, pre-frame GlobalTransforms of the owner rows (see kernels_tree.hip)
    uint32_t snap_rows = 0, snap_parity = 0;
    bool snap_valid = false;
    bool have_hierarchy = false;
    bool g_chg_in_bytes = false;  // the GlobalTransform change mask currently lives in g_changed_bytes (tree path)
    // host-side knowledge that lets a frame with no dirty Transform skip its launches: some byte of `changed` may be
    // non-zero (set by the uploads that mark rows, cleared when mi_propagate consumes the column); the change masks of
    // the last propagate may hold set bits
    bool changed_maybe = true, g_chg_maybe = true;

    // ---- views / visibility ----
    DevBuf views;
    uint32_t n_views = 0;
    DevBuf bitmask;
    uint64_t words_per_view = 0;
    // multi-GPU exchange (mi_exchange_configure): in-place all-gather of the masks after every cull
    struct Exchange {
        bool on = false;
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
        uint64_t words_per_view = 0, word_offset = 0, block_bytes = 0;
        uint32_t rank = 0;
        uint64_t frame = 0;
        hipStream_t comm_stream[MAX_COMMS] = {nullptr};
        hipEvent_t ev_kernels[MAX_BUFS] = {nullptr}, ev_gathered[MAX_BUFS] = {nullptr};
        // The collective is enqueued by a library-owned host thread: RCCL's enqueue path costs tens of
        // microseconds of CPU per call, which would otherwise sit in the frame's critical path on the caller's
        // thread.  The caller's thread never runs more than two frames ahead of it.
        std::thread worker;
        std::mutex m;
        std::condition_variable cv;
        std::deque<uint32_t> queue;
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
    ViewSet view_set{};      // views passed by value when n_views <= MAX_INLINE_VIEWS
    bool views_inline = false;
    // compaction
    DevBuf block_counts, seg_totals, seg_bases, out_rows, out_keys, wave_cnt, seg_mask;
    uint32_t compact_views = 0, compact_classes = 0;
    uint32_t class_bits[32] = {0};
    bool compact_fast = false;   // last compaction used the single-launch path (out_rows strided per segment)
    uint64_t seg_stride = 0;

    // ---- clustering ----
    DevBuf cl_pos, cl_type, cl_layers, cl_dir, cl_sincos, cl_planes, cl_spheres;
    // batching work-item build (kernels_batch.hip)
    uint32_t *bt_set = nullptr, *bt_bin = nullptr, *bt_input = nullptr, *bt_row_meta = nullptr;  // per-row columns
    bool bt_resolve = true;  // rows or tables changed: bt_row_meta must be recomputed
    DevBuf bt_set_indexed, bt_table_off, bt_table, bt_meta_off, bt_meta, bt_rows_a, bt_rows_b, bt_hist, bt_set_count, bt_set_scan,
        bt_counters, bt_wi[2], bt_md[2], bt_bs[2], bt_records, bt_totals;
    uint32_t bt_n_sets = 0, bt_n_meta = 0;
    bool bt_have_rows = false, bt_have_sets = false, bt_built = false;
    DevBuf cl_remap, cl_bind_oc, cl_bind_idx, cl_block_counts, cl_pair_cb, cl_pair_mask, cl_acc, cl_offsets, cl_indices, cl_scalars;
    uint32_t cl_parity = 0, cl_acc_clusters = 0, cl_acc_blocks = 0;  // cl_acc = 2 x [counts 6C | totals C | farthest_z + pad]
    uint32_t cl_n = 0;
    bool cl_have_type = false, cl_have_layers = false, cl_have_spot = false, cl_any_spot = false;
    ClusterViewDev cl_view{};
    bool cl_have_view = false, cl_assigned = false;

    // ---- timing ----
    hipEvent_t timer_a = nullptr, timer_b = nullptr;
    bool profiling = false;
    uint64_t prof_mask = ~0ull;
    uint32_t prof_every = 1, prof_tick[K_NUM_KERNELS] = {0};  // time every n-th launch of a kernel
    uint32_t prof_burst = 0, prof_timed[K_NUM_KERNELS] = {0};  // ... and at most the first prof_burst of them (0 = no limit)
    std::vector<ProfSpan> spans;
    bool span_open = false;
    uint64_t prof_launches[K_NUM_KERNELS] = {0};
    double prof_ms[K_NUM_KERNELS] = {0};
};

namespace {

int32_t fail(mi_ctx* ctx, int32_t code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    else {
        std::lock_guard<std::mutex> lk(g_err_mutex);
        g_create_error = buf;
    }
    return code;
}

#define HIP_TRY(ctx, expr)                                                                                   \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            return fail(ctx, e_ == hipErrorOutOfMemory ? MI_ERR_OUT_OF_MEMORY : MI_ERR_DEVICE, "%s: %s (%s:%d)", #expr, \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                          \
    } while (0)

#define ENTER(ctx)                                                       \
    do {                                                                 \
        if (!(ctx)) return fail(nullptr, MI_ERR_INVALID_ARG, "ctx is NULL"); \
        HIP_TRY(ctx, hipSetDevice((ctx)->device));                       \
    } while (0)

inline uint64_t words64(uint32_t n) { return ((uint64_t)n + 63u) / 64u; }
// bitmask words written by a launch of ceil(n/256) workgroups x 4 waves
inline uint64_t padded_words(uint32_t n) { return (((uint64_t)n + 255u) / 256u) * 4u; }

int32_t ensure(mi_ctx* ctx, DevBuf& b, size_t bytes) {
    if (b.bytes >= bytes && b.p) return MI_OK;
    if (b.p) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipFree(b.p));
        b.p = nullptr;
        b.bytes = 0;
    }
    size_t want = std::max<size_t>(bytes, 256);
    HIP_TRY(ctx, hipMalloc(&b.p, want));
    b.bytes = want;
    return MI_OK;
}

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
int32_t stage_alloc(mi_ctx* ctx, size_t bytes, void** out) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (ctx->stage_used + bytes > ctx->stage_bytes) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // everything staged so far has been consumed
        ctx->stage_used = 0;
        if (bytes > ctx->stage_bytes) {
            if (ctx->stage) HIP_TRY(ctx, hipHostFree(ctx->stage));
            ctx->stage = nullptr;
            size_t want = std::max<size_t>(bytes, (size_t)64 << 20);
            HIP_TRY(ctx, hipHostMalloc(&ctx->stage, want, hipHostMallocMapped));
            ctx->stage_bytes = want;
        }
    }
    *out = (char*)ctx->stage + ctx->stage_used;
    ctx->stage_used += bytes;
    return MI_OK;
}

int32_t upload(mi_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (!bytes) return MI_OK;
    void* st = nullptr;
    int32_t rc = stage_alloc(ctx, bytes, &st);
    if (rc) return rc;
    memcpy(st, src, bytes);
    HIP_TRY(ctx, hipMemcpyAsync(dst, st, bytes, hipMemcpyHostToDevice, ctx->stream));
    return MI_OK;
}

int32_t download(mi_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (!bytes) return MI_OK;
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MI_OK;
}

int32_t check_rows(mi_ctx* ctx, uint32_t first, uint32_t n, const char* what) {
    if ((uint64_t)first + n > ctx->n) return fail(ctx, MI_ERR_INVALID_ARG, "%s: rows [%u,%u) exceed %u live rows", what, first, first + n, ctx->n);
    return MI_OK;
}

// ---- profiling ------------------------------------------------------------------------------
void prof_close(mi_ctx* ctx) {
    if (ctx->span_open) {
        hipEventRecord(ctx->spans.back().b, ctx->stream);
        ctx->span_open = false;
    }
}
// Callback form used by the multi-kernel launch wrappers: arms dispatch-timestamp timing for the NEXT launch
// (kernel id K_NUM_KERNELS = disarm).
thread_local LaunchTimer g_cb_timer;
void prof_mark(void* vctx, uint32_t kernel) {
    mi_ctx* ctx = (mi_ctx*)vctx;
    if (!ctx->profiling) return;
    prof_close(ctx);
    if (g_launch_timer == &g_cb_timer) {  // previous arm was never consumed
        g_launch_timer = nullptr;
        hipEventDestroy(ctx->spans.back().a);
        hipEventDestroy(ctx->spans.back().b);
        ctx->spans.pop_back();
    }
    if (kernel >= K_NUM_KERNELS || !((ctx->prof_mask >> kernel) & 1ull)) return;
    ProfSpan sp;
    sp.kernel = kernel;
    hipEventCreate(&sp.a);
    hipEventCreate(&sp.b);
    ctx->spans.push_back(sp);
    g_cb_timer.start = sp.a;
    g_cb_timer.stop = sp.b;
    g_launch_timer = &g_cb_timer;
}
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
void prof_collect(mi_ctx* ctx) {
    prof_close(ctx);
    hipStreamSynchronize(ctx->stream);
    for (auto& sp : ctx->spans) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) {
            ctx->prof_ms[sp.kernel] += ms;
            ctx->prof_launches[sp.kernel] += 1;
        }
        hipEventDestroy(sp.a);
        hipEventDestroy(sp.b);
    }
    ctx->spans.clear();
}

Columns columns_of(mi_ctx* ctx) {
    Columns c;
    c.n = ctx->n;
    c.translation = ctx->t;
    c.rotation = ctx->r;
    c.scale = ctx->s;
    c.global = ctx->g;
    c.aabb_center = ctx->c;
    c.aabb_half = ctx->h;
    c.flags = ctx->flags;
    c.layer_mask = ctx->layers;
    c.view_visibility = ctx->vv;
    c.range_start_end = ctx->have_ranges ? ctx->range : nullptr;
    c.g_changed_bits = ctx->g_chg_bits;
    c.vv_changed_bits = ctx->vv_chg_bits;
    return c;
}

static_assert(sizeof(mi_view) == sizeof(ViewParams), "mi_view and ViewParams share one layout");

int32_t prepare_views(mi_ctx* ctx, const mi_view* views, uint32_t n_views, VisibilityOut* out) {
    if (!views || n_views == 0) return fail(ctx, MI_ERR_INVALID_ARG, "mi_cull: views NULL or n_views == 0");
    std::vector<ViewParams> vp(n_views);
    memcpy(vp.data(), views, sizeof(ViewParams) * n_views);
    for (auto& v : vp) v.pad[0] = v.pad[1] = v.pad[2] = 0;
    int32_t rc = MI_OK;
    ctx->views_inline = n_views <= MAX_INLINE_VIEWS;
    if (ctx->views_inline) {
        memcpy(ctx->view_set.v, vp.data(), sizeof(ViewParams) * n_views);  // travels in the kernarg segment
    } else {
        rc = ensure(ctx, ctx->views, sizeof(ViewParams) * n_views);
        if (rc) return rc;
        rc = upload(ctx, ctx->views.p, vp.data(), sizeof(ViewParams) * n_views);
        if (rc) return rc;
    }
    ctx->n_views = n_views;
    if (ctx->ext_bitmask) {
        out->bitmask = (uint64_t*)ctx->ext_bitmask;
        out->words_per_view = ctx->ext_words_per_view;
        out->word_offset = ctx->ext_word_offset;
    } else {
        ctx->words_per_view = padded_words(ctx->cap);
        rc = ensure(ctx, ctx->bitmask, ctx->words_per_view * 8 * n_views);
        if (rc) return rc;
        out->bitmask = (uint64_t*)ctx->bitmask.p;
        out->words_per_view = ctx->words_per_view;
        out->word_offset = 0;
    }
    return MI_OK;
}

int32_t rebuild_order(mi_ctx* ctx) {
    if (!ctx->order_dirty) return MI_OK;
    ctx->order_dirty = false;
    const uint32_t n = ctx->n;
    ctx->order_identity = true;
    if (!ctx->have_keys || n == 0) return MI_OK;
    const uint64_t* k = ctx->h_keys.data();
    bool sorted = true;
    for (uint32_t i = 1; i < n; ++i)
        if (k[i] < k[i - 1]) { sorted = false; break; }
    if (sorted) return MI_OK;
    std::vector<uint32_t> ord(n);
    std::iota(ord.begin(), ord.end(), 0u);
    std::stable_sort(ord.begin(), ord.end(), [k](uint32_t a, uint32_t b) { return k[a] < k[b]; });
    int32_t rc = ensure(ctx, ctx->order, (size_t)n * 4);
    if (rc) return rc;
    rc = upload(ctx, ctx->order.p, ord.data(), (size_t)n * 4);
    if (rc) return rc;
    ctx->order_identity = false;
    return MI_OK;
}

// Class slots present + the by-product buffers of the cull pass.  Decides between the single-launch
// compaction (rows already in Entity-key order) and the general count/scan/scatter path.
int32_t prepare_segments(mi_ctx* ctx, uint32_t n_views, SegOut* seg) {
    int32_t rc = rebuild_order(ctx);
    if (rc) return rc;
    uint32_t k = 0;
    const uint32_t present = ctx->have_class_mask ? ctx->classes_present : 1u;
    for (uint32_t b = 0; b < 32; ++b)
        if (present & (1u << b)) { ctx->class_bits[k] = b; ++k; }
    if (k == 0) { ctx->class_bits[0] = 0; k = 1; }
    ctx->compact_views = n_views;
    ctx->compact_classes = k;
    ctx->compact_fast = ctx->order_identity;
    memset(seg, 0, sizeof *seg);
    seg->n_classes = k;
    for (uint32_t i = 0; i < k; ++i) seg->class_bits[i] = (uint8_t)ctx->class_bits[i];
    seg->class_mask = nullptr;
    if (!ctx->compact_fast) return MI_OK;  // general path reads class_mask itself
    const size_t segs = (size_t)n_views * k;
    seg->n_waves = (uint32_t)((padded_words(ctx->cap) + 63u) / 64u * 64u);
    if ((rc = ensure(ctx, ctx->wave_cnt, segs * seg->n_waves))) return rc;
    seg->wave_cnt = (uint8_t*)ctx->wave_cnt.p;
    if (ctx->have_class_mask) {
        seg->class_mask = ctx->class_mask;
        seg->seg_words = padded_words(ctx->cap);
        if ((rc = ensure(ctx, ctx->seg_mask, segs * seg->seg_words * 8))) return rc;
        seg->seg_mask = (uint64_t*)ctx->seg_mask.p;
    }
    return MI_OK;
}

int32_t run_compaction(mi_ctx* ctx, const VisibilityOut& vo, const SegOut& seg) {
    int32_t rc;
    const uint32_t n_classes = ctx->compact_classes;
    const size_t segs = (size_t)ctx->n_views * n_classes;
    if ((rc = ensure(ctx, ctx->seg_totals, segs * 4))) return rc;
    if (ctx->n == 0) {
        HIP_TRY(ctx, hipMemsetAsync(ctx->seg_totals.p, 0, segs * 4, ctx->stream));
        return MI_OK;
    }
    if (ctx->compact_fast) {
        CompactFastArgs f{};
        f.n = ctx->n;
        f.n_segments = (uint32_t)segs;
        f.n_classes = n_classes;
        f.n_waves = seg.n_waves;
        f.wave_cnt = seg.wave_cnt;
        f.seg_mask = seg.seg_mask;
        f.seg_words = seg.seg_words;
        f.bitmask = vo.bitmask;
        f.words_per_view = vo.words_per_view;
        f.word_offset = vo.word_offset;
        ctx->seg_stride = ctx->cap;
        if ((rc = ensure(ctx, ctx->out_rows, segs * ctx->seg_stride * 4))) return rc;
        f.out_rows = (uint32_t*)ctx->out_rows.p;
        f.seg_stride = ctx->seg_stride;
        f.seg_totals = (uint32_t*)ctx->seg_totals.p;
        if (ctx->xch.on && ctx->xch.kernel_signal) {
            f.signal = ctx->xch.kernels_flag;
            f.signal_value = (uint32_t)(ctx->xch.frame + 1);
            ctx->xch.signalled = true;
        }
        ProfScope ps(ctx, K_COMPACT_FAST);
        HIP_TRY(ctx, launch_compact_fast(f, ctx->stream));
        return MI_OK;
    }
    CompactArgs a{};
    a.n = ctx->n;
    a.n_views = ctx->n_views;
    for (uint32_t i = 0; i < n_classes; ++i) a.class_bits[i] = ctx->class_bits[i];
    a.n_classes = n_classes;
    a.order = ctx->order_identity ? nullptr : (const uint32_t*)ctx->order.p;
    a.class_mask = ctx->have_class_mask ? ctx->class_mask : nullptr;
    a.entity_keys = ctx->have_keys ? ctx->keys : nullptr;
    a.bitmask = vo.bitmask;
    a.words_per_view = vo.words_per_view;
    a.word_offset = vo.word_offset;
    a.n_blocks = (ctx->n + COMPACT_BLOCK_ROWS - 1) / COMPACT_BLOCK_ROWS;
    if ((rc = ensure(ctx, ctx->block_counts, segs * a.n_blocks * 4))) return rc;
    if ((rc = ensure(ctx, ctx->seg_bases, segs * 8))) return rc;
    // worst case: every row of every view in every class it belongs to
    const size_t max_entries = (size_t)a.n_views * ctx->cap * (ctx->have_class_mask ? a.n_classes : 1);
    if ((rc = ensure(ctx, ctx->out_rows, max_entries * 4))) return rc;
    if ((rc = ensure(ctx, ctx->out_keys, max_entries * 8))) return rc;
    a.block_counts = (uint32_t*)ctx->block_counts.p;
    a.seg_totals = (uint32_t*)ctx->seg_totals.p;
    a.seg_bases = (uint64_t*)ctx->seg_bases.p;
    a.out_rows = (uint32_t*)ctx->out_rows.p;
    a.out_keys = (uint64_t*)ctx->out_keys.p;
    HIP_TRY(ctx, launch_compact(a, ctx->stream, prof_mark, ctx));
    return MI_OK;
}

// Multi-GPU exchange around a cull: bind this frame's gathered buffer (after its previous all-gather drained),
// and afterwards hand the in-place all-gather to the exchange thread, which enqueues it on the communication
// stream behind the kernels.
void exchange_worker(mi_ctx* ctx) {
    auto& x = ctx->xch;
    hipSetDevice(ctx->device);
    for (;;) {
        uint32_t slot;
        // While frames are flowing the thread must not go to sleep between them: waking a thread through a futex
        // takes tens of microseconds, more than a frame.  Poll the submission counter for a while first.
        if (x.submitted_fast.load(std::memory_order_acquire) == x.worker_frames) {
            const auto spin0 = std::chrono::steady_clock::now();
            uint32_t spins = 0;
            while (x.submitted_fast.load(std::memory_order_acquire) == x.worker_frames && !x.stop_fast.load(std::memory_order_relaxed)) {
                __builtin_ia32_pause();
                if ((++spins & 255u) == 0 && std::chrono::steady_clock::now() - spin0 > std::chrono::microseconds(500)) break;
            }
        }
        {
            std::unique_lock<std::mutex> lk(x.m);
            if (x.queue.empty() && !x.stop) {
                x.sleeping.store(true, std::memory_order_seq_cst);
                x.cv.wait(lk, [&] { return x.stop || !x.queue.empty(); });
                x.sleeping.store(false, std::memory_order_relaxed);
            }
            if (x.queue.empty()) return;  // stop requested and drained
            slot = x.queue.front();
            x.queue.pop_front();
        }
        int err = 0;
        const auto tw0 = std::chrono::steady_clock::now();
        const uint32_t k = (uint32_t)(x.worker_frames % x.n_comms);  // frame f travels on communicator f % n_comms
        hipStream_t cs = x.comm_stream[k];
        if (hipStreamWaitValue32(cs, x.kernels_flag, (uint32_t)(x.worker_frames + 1), hipStreamWaitValueGte, 0xFFFFFFFFu) != hipSuccess) err = -1;
        char* base = (char*)x.buf[slot];
        if (!err) err = x.all_gather(base + (size_t)x.rank * x.block_bytes, base, (size_t)x.block_bytes, 1 /* ncclUint8 */, x.comm[k], cs);
        if (hipEventRecord(x.ev_gathered[slot], cs) != hipSuccess && !err) err = -2;
        // completion counter the caller's thread can read without a driver call
        if (hipStreamWriteValue32(cs, (void*)(x.done_flag + k), (uint32_t)(x.worker_frames / x.n_comms + 1), 0) != hipSuccess && !err) err = -3;
        ++x.worker_frames;
        x.dbg_worker_ns += std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - tw0).count();
        {
            std::lock_guard<std::mutex> lk(x.m);
            if (err && !x.worker_error) x.worker_error = err;
            ++x.issued;
        }
        x.cv.notify_all();
    }
}
// blocks until the exchange thread has enqueued the collectives of frames < upto
int32_t exchange_wait_issued(mi_ctx* ctx, uint64_t upto) {
    auto& x = ctx->xch;
    std::unique_lock<std::mutex> lk(x.m);
    x.cv.wait(lk, [&] { return x.issued >= upto || x.worker_error; });
    if (x.worker_error) return fail(ctx, MI_ERR_DEVICE, "exchange thread: ncclAllGather / HIP call failed (%d)", x.worker_error);
    return MI_OK;
}
void exchange_stop(mi_ctx* ctx) {
    auto& x = ctx->xch;
    if (x.worker.joinable()) {
        {
            std::lock_guard<std::mutex> lk(x.m);
            x.stop = true;
        }
        x.stop_fast.store(true);
        x.cv.notify_all();
        x.worker.join();
    }
    x.stop = false;
    x.stop_fast.store(false);
}
int32_t exchange_begin(mi_ctx* ctx) {
    auto& x = ctx->xch;
    if (!x.on) return MI_OK;
    const uint32_t slot = (uint32_t)(x.frame % x.n_bufs);
    const auto tb0 = std::chrono::steady_clock::now();
    struct Acc { double& d; std::chrono::steady_clock::time_point t; ~Acc() { d += std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t).count(); } } acc{x.dbg_begin_ns, tb0};
    if (x.frame >= x.n_bufs) {
        // This buffer was last used n_bufs frames ago and its all-gather must have completed before the kernels
        // overwrite it.  The dependency is enforced on the HOST (the communication stream bumps a pinned counter
        // behind every all-gather), not with a cross-stream wait: a barrier packet on the compute queue costs
        // ~6 us of GPU time per frame, while pacing the caller n_bufs - 1 frames ahead of the exchange costs
        // nothing as long as the next frame is already queued.
        const uint64_t need = x.frame - x.n_bufs + 1;
        uint32_t spins = 0;

        // frames 0 .. need-1 complete <=> every communicator k has finished its ceil((need - k) / n_comms) of them
        auto drained = [&]() {
            for (uint32_t k = 0; k < x.n_comms; ++k)
                if ((uint64_t)x.done_flag[k] < (need + x.n_comms - 1 - k) / x.n_comms) return false;
            return true;
        };
        while (!drained()) {
            if ((++spins & 1023u) == 0) {
                {
                    std::lock_guard<std::mutex> lk(x.m);
                    if (x.worker_error) return fail(ctx, MI_ERR_DEVICE, "exchange thread: ncclAllGather / HIP call failed (%d)", x.worker_error);
                }
                if (std::chrono::steady_clock::now() - tb0 > std::chrono::seconds(30))
                    return fail(ctx, MI_ERR_DEVICE, "exchange: the all-gather of frame %llu did not complete within 30 s",
                                (unsigned long long)(need - 1));
                std::this_thread::yield();
            }
        }
        x.dbg_wait_ns += std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - tb0).count();
    }
    ctx->ext_bitmask = x.buf[slot];
    ctx->ext_words_per_view = x.words_per_view;
    ctx->ext_word_offset = x.word_offset;
    return MI_OK;
}
int32_t exchange_end(mi_ctx* ctx) {
    auto& x = ctx->xch;
    if (!x.on) return MI_OK;
    const uint32_t slot = (uint32_t)(x.frame % x.n_bufs);
    const auto te0 = std::chrono::steady_clock::now();
    if (!x.signalled) HIP_TRY(ctx, hipStreamWriteValue32(ctx->stream, x.kernels_flag, (uint32_t)(x.frame + 1), 0));
    x.signalled = false;
    {
        std::lock_guard<std::mutex> lk(x.m);
        x.queue.push_back(slot);
        ++x.submitted;
    }
    x.submitted_fast.fetch_add(1, std::memory_order_seq_cst);
    if (x.sleeping.load(std::memory_order_seq_cst)) x.cv.notify_all();  // no futex call while the thread is polling
    ++x.frame;
    x.dbg_end_ns += std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - te0).count();
    return MI_OK;
}

}  // namespace

// =============================================================================================
// lifecycle
// =============================================================================================
extern "C" {

int32_t mi_abi_version(void) { return MI_ABI_VERSION; }

int32_t mi_ctx_create(int32_t device, void* hip_stream, mi_ctx** out_ctx) {
    if (!out_ctx) return fail(nullptr, MI_ERR_INVALID_ARG, "out_ctx is NULL");
    *out_ctx = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(nullptr, MI_ERR_DEVICE, "no HIP device available (%s)", hipGetErrorString(e));
    if (device < 0 || device >= count) return fail(nullptr, MI_ERR_INVALID_ARG, "device %d out of range [0,%d)", device, count);
    mi_ctx* ctx = new (std::nothrow) mi_ctx();
    if (!ctx) return fail(nullptr, MI_ERR_OUT_OF_MEMORY, "host allocation failed");
    ctx->device = device;
    if ((e = hipSetDevice(device)) != hipSuccess) {
        delete ctx;
        return fail(nullptr, MI_ERR_DEVICE, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            std::string arch = prop.gcnArchName;
            delete ctx;
            return fail(nullptr, MI_ERR_DEVICE, "device %d is %s; this library is built for gfx950 (MI355X) only", device,
                        arch.c_str());
        }
    }
    if (hip_stream) {
        ctx->stream = (hipStream_t)hip_stream;
    } else {
        if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) {
            delete ctx;
            return fail(nullptr, MI_ERR_DEVICE, "hipStreamCreate: %s", hipGetErrorString(e));
        }
        ctx->own_stream = true;
    }
    hipEventCreate(&ctx->timer_a);
    hipEventCreate(&ctx->timer_b);
    if ((e = set_cluster_lds_limit()) != hipSuccess) {
        std::string msg = hipGetErrorString(e);
        mi_ctx_destroy(ctx);
        return fail(nullptr, MI_ERR_DEVICE, "raising the LDS limit of the clustering kernel failed: %s", msg.c_str());
    }
    ctx->level_offsets = {0, 0};
    *out_ctx = ctx;
    return MI_OK;
}

int32_t mi_ctx_destroy(mi_ctx* ctx) {
    if (!ctx) return MI_OK;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    prof_collect(ctx);
    void* cols[] = {ctx->t, ctx->r, ctx->s, ctx->g, ctx->c, ctx->h, ctx->flags, ctx->vv, ctx->changed, ctx->g_changed_bytes,
                    ctx->layers, ctx->class_mask, ctx->keys, ctx->g_chg_bits, ctx->vv_chg_bits, ctx->tree_bits,
                    ctx->range, ctx->visibility, ctx->inh_changed, ctx->bt_set, ctx->bt_bin, ctx->bt_input, ctx->bt_row_meta};
    for (void* p : cols)
        if (p) hipFree(p);
    DevBuf* bufs[] = {&ctx->order, &ctx->chains, &ctx->snap, &ctx->inh_bits, &ctx->sparse_cnt, &ctx->sparse_rows, &ctx->sparse_total, &ctx->sparse_g, &ctx->parent_idx, &ctx->node_flags, &ctx->tiles, &ctx->views, &ctx->bitmask,
                      &ctx->block_counts, &ctx->seg_totals, &ctx->seg_bases, &ctx->out_rows, &ctx->out_keys, &ctx->wave_cnt, &ctx->seg_mask, &ctx->cl_pos,
                      &ctx->cl_type, &ctx->cl_layers, &ctx->cl_dir, &ctx->cl_sincos, &ctx->cl_planes, &ctx->cl_spheres,
                      &ctx->bt_set_indexed, &ctx->bt_table_off, &ctx->bt_table, &ctx->bt_meta_off, &ctx->bt_meta, &ctx->bt_rows_a, &ctx->bt_rows_b,
                      &ctx->bt_hist, &ctx->bt_set_count, &ctx->bt_set_scan, &ctx->bt_counters, &ctx->bt_wi[0], &ctx->bt_wi[1], &ctx->bt_md[0],
                      &ctx->bt_md[1], &ctx->bt_bs[0], &ctx->bt_bs[1], &ctx->bt_records, &ctx->bt_totals,
                      &ctx->cl_remap, &ctx->cl_bind_oc, &ctx->cl_bind_idx, &ctx->cl_block_counts, &ctx->cl_pair_cb, &ctx->cl_pair_mask, &ctx->cl_acc,
                      &ctx->cl_offsets, &ctx->cl_indices, &ctx->cl_scalars};
    for (DevBuf* b : bufs)
        if (b->p) hipFree(b->p);
    if (ctx->stage) hipHostFree(ctx->stage);
    if (ctx->timer_a) hipEventDestroy(ctx->timer_a);
    if (ctx->timer_b) hipEventDestroy(ctx->timer_b);
    exchange_stop(ctx);
    if (getenv("MI_XCH_DEBUG") && ctx->xch.frame)
        fprintf(stderr, "[mi exchange] frames %llu: begin %.2f us (wait-issued %.2f us), end %.2f us, worker %.2f us per frame\n",
                (unsigned long long)ctx->xch.frame, ctx->xch.dbg_begin_ns / ctx->xch.frame / 1e3, ctx->xch.dbg_wait_ns / ctx->xch.frame / 1e3,
                ctx->xch.dbg_end_ns / ctx->xch.frame / 1e3, ctx->xch.dbg_worker_ns / ctx->xch.frame / 1e3);
    if (ctx->xch.comm_stream[0]) {
        for (hipStream_t cs : ctx->xch.comm_stream)
            if (cs) hipStreamSynchronize(cs);
        for (uint32_t i = 0; i < mi_ctx::Exchange::MAX_BUFS; ++i) {
            if (ctx->xch.ev_kernels[i]) hipEventDestroy(ctx->xch.ev_kernels[i]);
            if (ctx->xch.ev_gathered[i]) hipEventDestroy(ctx->xch.ev_gathered[i]);
        }
        for (hipStream_t cs : ctx->xch.comm_stream)
            if (cs) hipStreamDestroy(cs);
        if (ctx->xch.done_flag) hipHostFree((void*)ctx->xch.done_flag);
    }
    if (ctx->own_stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    return MI_OK;
}

const char* mi_last_error_string(mi_ctx* ctx) {
    if (ctx) return ctx->err.c_str();
    std::lock_guard<std::mutex> lk(g_err_mutex);
    static thread_local std::string copy;
    copy = g_create_error;
    return copy.c_str();
}

int32_t mi_synchronize(mi_ctx* ctx) {
    ENTER(ctx);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->xch.on) {
        int32_t rc = exchange_wait_issued(ctx, ctx->xch.frame);
        if (rc) return rc;
    }
    for (hipStream_t cs : ctx->xch.comm_stream)
        if (cs) HIP_TRY(ctx, hipStreamSynchronize(cs));
    return MI_OK;
}

// =============================================================================================
// columns
// =============================================================================================
int32_t mi_columns_resize(mi_ctx* ctx, uint32_t n_rows) {
    ENTER(ctx);
    ctx->bt_resolve = true;
    ctx->changed_maybe = true;
    if (n_rows > ctx->cap) {
        uint32_t new_cap = std::max<uint64_t>(n_rows, std::min<uint64_t>((uint64_t)ctx->cap * 3 / 2, 0xFFFFFF00ull));
        new_cap = (uint32_t)(((uint64_t)new_cap + 255u) / 256u * 256u);  // whole workgroups
        const uint32_t old = ctx->n;
        int32_t rc;
        if ((rc = grow_column(ctx, ctx->t, 3, old, new_cap, 0))) return rc;
        if ((rc = grow_column(ctx, ctx->r, 4, old, new_cap, 0))) return rc;
        if ((rc = grow_column(ctx, ctx->s, 3, old, new_cap, 0))) return rc;
        if ((rc = grow_column(ctx, ctx->g, 12, old, new_cap, 0))) return rc;
        if ((rc = grow_column(ctx, ctx->c, 3, old, new_cap, 0))) return rc;
        if ((rc = grow_column(ctx, ctx->h, 3, old, new_cap, 0))) return rc;
        if ((rc = grow_column(ctx, ctx->flags, 1, old, new_cap, MI_FLAG_INHERITED_VISIBLE))) return rc;
        if ((rc = grow_column(ctx, ctx->vv, 1, old, new_cap, 0))) return rc;
        if ((rc = grow_column(ctx, ctx->changed, 1, old, new_cap, 1))) return rc;
        ctx->changed_maybe = true;
        if ((rc = grow_column(ctx, ctx->g_changed_bytes, 1, old, new_cap, 0))) return rc;
        if ((rc = grow_column(ctx, ctx->layers, 1, old, new_cap, 0))) return rc;
        if ((rc = grow_column(ctx, ctx->class_mask, 1, old, new_cap, 0))) return rc;
        if ((rc = grow_column(ctx, ctx->keys, 1, old, new_cap, 0))) return rc;
        if ((rc = grow_column(ctx, ctx->range, 2, old, new_cap, 0))) return rc;
        if ((rc = grow_column(ctx, ctx->visibility, 1, old, new_cap, 0))) return rc;
        if ((rc = grow_column(ctx, ctx->inh_changed, 1, old, new_cap, 0))) return rc;
        if ((rc = grow_column(ctx, ctx->bt_set, 1, old, new_cap, 0xFF))) return rc;  // MI_NO_BATCH_SET until uploaded
        if ((rc = grow_column(ctx, ctx->bt_bin, 1, old, new_cap, 0))) return rc;
        if ((rc = grow_column(ctx, ctx->bt_input, 1, old, new_cap, 0))) return rc;
        if ((rc = grow_column(ctx, ctx->bt_row_meta, 1, old, new_cap, 0xFF))) return rc;
        // default RenderLayers = layer 0 (mask 1) for rows never uploaded
        {
            std::vector<uint32_t> ones(new_cap - old, 1u);
            if ((rc = upload(ctx, ctx->layers + old, ones.data(), ones.size() * 4))) return rc;
        }
        uint64_t* nb = nullptr;
        const size_t wbytes = padded_words(new_cap) * 8 + 256;
        for (uint64_t** bits : {&ctx->g_chg_bits, &ctx->vv_chg_bits}) {
            HIP_TRY(ctx, hipMalloc((void**)&nb, wbytes));
            HIP_TRY(ctx, hipMemsetAsync(nb, 0, wbytes, ctx->stream));
            if (*bits) {
                HIP_TRY(ctx, hipMemcpyAsync(nb, *bits, words64(old) * 8, hipMemcpyDeviceToDevice, ctx->stream));
                HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
                HIP_TRY(ctx, hipFree(*bits));
            }
            *bits = nb;
        }
        if (ctx->tree_bits) {
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            HIP_TRY(ctx, hipFree(ctx->tree_bits));
        }
        HIP_TRY(ctx, hipMalloc((void**)&ctx->tree_bits, padded_words(new_cap) * 8 + 256));
        HIP_TRY(ctx, hipMemsetAsync(ctx->tree_bits, 0, padded_words(new_cap) * 8 + 256, ctx->stream));
        ctx->cap = new_cap;
    }
    if (n_rows != ctx->n) {
        // a different row count invalidates the hierarchy and any cull result
        ctx->have_hierarchy = false;
        ctx->n_levels = 1;
        ctx->culled = false;
        ctx->h_keys.resize(n_rows, 0);
        ctx->order_dirty = true;
        ctx->level_offsets = {0, n_rows};
        ctx->passes.clear();
        ctx->groups.clear();
    }
    ctx->n = n_rows;
    return MI_OK;
}

int32_t mi_upload_transforms(mi_ctx* ctx, uint32_t first_row, uint32_t n, const float* translation, const float* rotation,
                             const float* scale) {
    ENTER(ctx);
    if (!translation || !rotation || !scale) return fail(ctx, MI_ERR_INVALID_ARG, "mi_upload_transforms: NULL column");
    int32_t rc = check_rows(ctx, first_row, n, "mi_upload_transforms");
    if (rc) return rc;
    if (n && n <= SMALL_UPLOAD_ROWS) {  // dirty-row sized: one staging block, one scatter kernel
        void* st = nullptr;
        if ((rc = stage_alloc(ctx, (size_t)n * 40, &st))) return rc;
        float* f = (float*)st;
        memcpy(f, translation, (size_t)n * 12);
        memcpy(f + 3 * (size_t)n, rotation, (size_t)n * 16);
        memcpy(f + 7 * (size_t)n, scale, (size_t)n * 12);
        void* dev = nullptr;
        HIP_TRY(ctx, hipHostGetDevicePointer(&dev, st, 0));
        HIP_TRY(ctx, launch_upload_trs((const float*)dev, ctx->t, ctx->r, ctx->s, first_row, n, ctx->stream));
        return MI_OK;
    }
    if ((rc = upload(ctx, ctx->t + 3 * (size_t)first_row, translation, (size_t)n * 12))) return rc;
    if ((rc = upload(ctx, ctx->r + 4 * (size_t)first_row, rotation, (size_t)n * 16))) return rc;
    if ((rc = upload(ctx, ctx->s + 3 * (size_t)first_row, scale, (size_t)n * 12))) return rc;
    return MI_OK;
}

int32_t mi_upload_transforms_indexed(mi_ctx* ctx, uint32_t n, const uint32_t* rows, const float* translation,
                                     const float* rotation, const float* scale) {
    ENTER(ctx);
    if (n == 0) return MI_OK;
    if (!rows || !translation || !rotation || !scale) return fail(ctx, MI_ERR_INVALID_ARG, "mi_upload_transforms_indexed: NULL");
    for (uint32_t i = 0; i < n; ++i)
        if (rows[i] >= ctx->n) return fail(ctx, MI_ERR_INVALID_ARG, "mi_upload_transforms_indexed: row %u >= %u live rows", rows[i], ctx->n);
    void* st = nullptr;
    int32_t rc = stage_alloc(ctx, (size_t)n * 44, &st);
    if (rc) return rc;
    uint32_t* u = (uint32_t*)st;
    float* f = (float*)(u + n);
    memcpy(u, rows, (size_t)n * 4);
    memcpy(f, translation, (size_t)n * 12);
    memcpy(f + 3 * (size_t)n, rotation, (size_t)n * 16);
    memcpy(f + 7 * (size_t)n, scale, (size_t)n * 12);
    void* dev = nullptr;
    HIP_TRY(ctx, hipHostGetDevicePointer(&dev, st, 0));
    if (!ctx->have_changed) {
        // first use of the change column: rows never marked count as unchanged from here on
        HIP_TRY(ctx, hipMemsetAsync(ctx->changed, 0, ctx->n, ctx->stream));
        ctx->have_changed = true;
    }
    HIP_TRY(ctx, launch_upload_trs_indexed((const uint32_t*)dev, n, ctx->t, ctx->r, ctx->s, ctx->changed, ctx->stream));
    ctx->changed_maybe = true;
    return MI_OK;
}

int32_t mi_upload_global_transforms(mi_ctx* ctx, uint32_t first_row, uint32_t n, const float* global12) {
    ENTER(ctx);
    if (!global12) return fail(ctx, MI_ERR_INVALID_ARG, "mi_upload_global_transforms: NULL");
    int32_t rc = check_rows(ctx, first_row, n, "mi_upload_global_transforms");
    if (rc) return rc;
    ctx->snap_valid = false;  // externally supplied values: the tree path re-snapshots its owner rows
    return upload(ctx, ctx->g + 12 * (size_t)first_row, global12, (size_t)n * 48);
}

int32_t mi_upload_bounds(mi_ctx* ctx, uint32_t first_row, uint32_t n, const float* aabb_center, const float* aabb_half,
                         const uint8_t* flags, const uint32_t* layer_mask) {
    ENTER(ctx);
    if (!aabb_center || !aabb_half) return fail(ctx, MI_ERR_INVALID_ARG, "mi_upload_bounds: NULL column");
    int32_t rc = check_rows(ctx, first_row, n, "mi_upload_bounds");
    if (rc) return rc;
    if ((rc = upload(ctx, ctx->c + 3 * (size_t)first_row, aabb_center, (size_t)n * 12))) return rc;
    if ((rc = upload(ctx, ctx->h + 3 * (size_t)first_row, aabb_half, (size_t)n * 12))) return rc;
    if (flags) {
        if ((rc = upload(ctx, ctx->flags + first_row, flags, n))) return rc;
    } else {
        std::vector<uint8_t> def(n, (uint8_t)(MI_FLAG_INHERITED_VISIBLE | MI_FLAG_HAS_AABB));
        if ((rc = upload(ctx, ctx->flags + first_row, def.data(), n))) return rc;
    }
    if (layer_mask) {
        if ((rc = upload(ctx, ctx->layers + first_row, layer_mask, (size_t)n * 4))) return rc;
    } else {
        std::vector<uint32_t> def(n, 1u);
        if ((rc = upload(ctx, ctx->layers + first_row, def.data(), (size_t)n * 4))) return rc;
    }
    return MI_OK;
}

int32_t mi_upload_view_visibility(mi_ctx* ctx, uint32_t first_row, uint32_t n, const uint8_t* vv) {
    ENTER(ctx);
    if (!vv) return fail(ctx, MI_ERR_INVALID_ARG, "mi_upload_view_visibility: NULL");
    int32_t rc = check_rows(ctx, first_row, n, "mi_upload_view_visibility");
    if (rc) return rc;
    return upload(ctx, ctx->vv + first_row, vv, n);
}

int32_t mi_upload_visibility_classes(mi_ctx* ctx, uint32_t first_row, uint32_t n, const uint32_t* class_mask) {
    ENTER(ctx);
    if (!class_mask) return fail(ctx, MI_ERR_INVALID_ARG, "mi_upload_visibility_classes: NULL");
    int32_t rc = check_rows(ctx, first_row, n, "mi_upload_visibility_classes");
    if (rc) return rc;
    uint32_t present = 0;
    for (uint32_t i = 0; i < n; ++i) present |= class_mask[i];
    if (!ctx->have_class_mask) ctx->classes_present = 0;
    ctx->classes_present |= present;
    ctx->have_class_mask = true;
    return upload(ctx, ctx->class_mask + first_row, class_mask, (size_t)n * 4);
}

int32_t mi_upload_entity_keys(mi_ctx* ctx, uint32_t first_row, uint32_t n, const uint64_t* entity_bits) {
    ENTER(ctx);
    if (!entity_bits) return fail(ctx, MI_ERR_INVALID_ARG, "mi_upload_entity_keys: NULL");
    int32_t rc = check_rows(ctx, first_row, n, "mi_upload_entity_keys");
    if (rc) return rc;
    ctx->h_keys.resize(ctx->n, 0);
    memcpy(ctx->h_keys.data() + first_row, entity_bits, (size_t)n * 8);
    ctx->have_keys = true;
    ctx->order_dirty = true;
    return upload(ctx, ctx->keys + first_row, entity_bits, (size_t)n * 8);
}

int32_t mi_upload_changed(mi_ctx* ctx, uint32_t first_row, uint32_t n, const uint8_t* changed) {
    ENTER(ctx);
    if (!changed) return fail(ctx, MI_ERR_INVALID_ARG, "mi_upload_changed: NULL");
    int32_t rc = check_rows(ctx, first_row, n, "mi_upload_changed");
    if (rc) return rc;
    ctx->have_changed = true;
    ctx->changed_maybe = true;
    return upload(ctx, ctx->changed + first_row, changed, n);
}

int32_t mi_upload_visibility_ranges(mi_ctx* ctx, uint32_t first_row, uint32_t n, const float* start_end) {
    ENTER(ctx);
    if (!start_end) {  // no VisibleEntityRanges resource: ranged rows are not range-culled
        ctx->have_ranges = false;
        return MI_OK;
    }
    int32_t rc = check_rows(ctx, first_row, n, "mi_upload_visibility_ranges");
    if (rc) return rc;
    ctx->have_ranges = true;
    return upload(ctx, ctx->range + 2 * (size_t)first_row, start_end, (size_t)n * 8);
}

int32_t mi_upload_visibility(mi_ctx* ctx, uint32_t first_row, uint32_t n, const uint8_t* visibility) {
    ENTER(ctx);
    if (!visibility) return fail(ctx, MI_ERR_INVALID_ARG, "mi_upload_visibility: NULL");
    int32_t rc = check_rows(ctx, first_row, n, "mi_upload_visibility");
    if (rc) return rc;
    return upload(ctx, ctx->visibility + first_row, visibility, n);
}

// ---------------------------------------------------------------------------------------------
// hierarchy: validation + subtree-tile planning (host), see kernels_tree.hip for the consumer.
// ---------------------------------------------------------------------------------------------
int32_t mi_upload_hierarchy(mi_ctx* ctx, uint32_t n, const uint32_t* parent_idx, const uint32_t* level_offsets,
                            uint32_t n_levels) {
    ENTER(ctx);
    if (n != ctx->n) return fail(ctx, MI_ERR_INVALID_ARG, "mi_upload_hierarchy: n (%u) != live rows (%u)", n, ctx->n);
    ctx->changed_maybe = true;  // conservative: the next propagate looks at the rows again
    if (!parent_idx || n_levels <= 1) {
        ctx->have_hierarchy = false;
        ctx->n_levels = 1;
        ctx->level_offsets = {0, n};
        ctx->passes.clear();
        ctx->groups.clear();
        return MI_OK;
    }
    if (!level_offsets) return fail(ctx, MI_ERR_INVALID_ARG, "mi_upload_hierarchy: level_offsets NULL");
    if (level_offsets[0] != 0 || level_offsets[n_levels] != n)
        return fail(ctx, MI_ERR_MALFORMED_HIERARCHY, "level_offsets must start at 0 and end at n");
    for (uint32_t l = 0; l < n_levels; ++l)
        if (level_offsets[l + 1] < level_offsets[l]) return fail(ctx, MI_ERR_MALFORMED_HIERARCHY, "level_offsets not monotone");
    // level 0: roots; level l>0: parent in level l-1, parents non-decreasing (BFS order)
    for (uint32_t i = level_offsets[0]; i < level_offsets[1]; ++i)
        if (parent_idx[i] != MI_NO_PARENT)
            return fail(ctx, MI_ERR_MALFORMED_HIERARCHY, "row %u in level 0 has a parent", i);
    for (uint32_t l = 1; l < n_levels; ++l) {
        const uint32_t plo = level_offsets[l - 1], phi = level_offsets[l];
        uint32_t prev = plo;
        for (uint32_t i = level_offsets[l]; i < level_offsets[l + 1]; ++i) {
            const uint32_t p = parent_idx[i];
            if (p < plo || p >= phi)
                return fail(ctx, MI_ERR_MALFORMED_HIERARCHY, "row %u (level %u): parent %u is not in level %u", i, l, p, l - 1);
            if (p < prev)
                return fail(ctx, MI_ERR_MALFORMED_HIERARCHY, "row %u: rows of a level must be ordered by parent (use mi_hierarchy_sort)", i);
            prev = p;
        }
    }
    // node flags + first-child table
    std::vector<uint8_t> nflags(n, 0);
    std::vector<uint32_t> first_child((size_t)n + 1, 0);
    for (uint32_t l = 0; l + 1 < n_levels; ++l) {
        uint32_t ch = level_offsets[l + 1];
        const uint32_t chi = level_offsets[l + 2];
        for (uint32_t p = level_offsets[l]; p < level_offsets[l + 1]; ++p) {
            first_child[p] = ch;
            while (ch < chi && parent_idx[ch] == p) { ++ch; nflags[p] |= 1; }
        }
    }
    for (uint32_t p = level_offsets[n_levels - 1]; p <= n; ++p) first_child[p] = n;
    // first_child[level end] of level l must read as "end of level l+1": patch boundaries
    auto child_begin = [&](uint32_t l, uint32_t row) -> uint32_t {
        // first row of level l+1 whose parent >= row (row in level l, or == end of level l)
        if (row >= level_offsets[l + 1]) return level_offsets[l + 2];
        return first_child[row];
    };

    // ---- tile plan ----
    // A pass = one launch covering `d` consecutive levels; its tiles partition the rows of the level the pass is
    // rooted in.  Pass 0 is rooted in level 0 itself (tile level 0 = a range of roots / flat rows); later passes
    // are rooted in the last level of the previous pass (tile level 0 = the children of a range of its rows).
    // A tile "fits" when all its levels but the last together hold <= TILE_UCAP rows (they live in LDS).
    const char* env_levels = getenv("MI_TILE_LEVELS");
    const uint32_t max_d = env_levels ? std::max(1, std::min((int)TILE_MAX_LEVELS, atoi(env_levels))) : TILE_MAX_LEVELS;
    std::vector<TileDesc> tiles;
    std::vector<uint32_t> chains;
    ctx->passes.clear();
    ctx->groups.clear();
    auto level_size = [&](uint32_t lv) -> uint64_t { return lv < n_levels ? level_offsets[lv + 1] - level_offsets[lv] : 0; };
    uint32_t l = 0;  // first level this pass computes
    while (l < n_levels) {
        const bool roots = l == 0;
        // band depth: as deep as possible while an average root range of one row still fits in LDS
        const uint64_t n_roots = std::max<uint64_t>(1, roots ? level_size(0) : level_size(l - 1));
        uint32_t d = 1;
        uint64_t upper = level_size(l);  // rows of the levels that would be non-last if we add one more level
        while (d < max_d && l + d < n_levels && upper <= (uint64_t)TILE_UCAP * n_roots) {
            ++d;
            upper += level_size(l + d - 1);
        }
        // [lo,hi) is a row range of the rooting level; returns the tile and whether it fits
        auto build = [&](uint32_t lo, uint32_t hi, TileDesc& td) -> bool {
            uint32_t clo = lo, chi2 = hi;
            td = TileDesc{};
            for (uint32_t k = 0; k < d; ++k) {
                uint32_t nlo, nhi;
                if (roots && k == 0) { nlo = lo; nhi = hi; }
                else {
                    const uint32_t plevel = roots ? k - 1 : l - 1 + k;
                    nlo = child_begin(plevel, clo);
                    nhi = child_begin(plevel, chi2);
                }
                td.start[k] = nlo;
                td.count[k] = nhi - nlo;
                clo = nlo; chi2 = nhi;
                if (nhi > nlo) td.n_levels = k + 1;
            }
            uint64_t up = 0;
            for (uint32_t k = 0; k + 1 < td.n_levels; ++k) up += td.count[k];
            return up <= TILE_UCAP;
        };
        // Chain candidate (see "Tile kinds" below): then every tile hangs below exactly one node, as long as that
        // still gives tiles of a decent size.
        uint64_t pass_rows = 0;
        for (uint32_t k = 0; k < d; ++k) pass_rows += level_size(l + k);
        const bool chain_candidate = !roots && l <= TILE_MAX_CHAIN && !ctx->groups.empty() && ctx->groups.back().n_chain == 0 &&
                                     ctx->groups.back().count <= 64 && getenv("MI_TILE_NO_CHAIN") == nullptr &&
                                     n_roots <= 16384 && pass_rows >= 128 * n_roots;
        const uint32_t first_tile = (uint32_t)tiles.size();
        const uint32_t rl = roots ? 0 : l - 1;
        const uint32_t rlo = level_offsets[rl], rhi = level_offsets[rl + 1];
        uint32_t a = rlo;
        while (a < rhi) {
            TileDesc best{};
            uint32_t b = a + 1;
            build(a, b, best);
            uint32_t step = 1;  // galloping extension of the root range
            while (b < rhi && !chain_candidate) {
                const uint32_t nb = (uint32_t)std::min<uint64_t>((uint64_t)b + step, rhi);
                TileDesc cand{};
                // keep tiles small enough to spread over the chip: at most 4 x TILE_UCAP rows in the streamed last level
                if (build(a, nb, cand) && (cand.n_levels == 0 || cand.count[cand.n_levels - 1] <= 4 * TILE_UCAP || nb == a + 1)) { best = cand; b = nb; step *= 2; }
                else if (step > 1) step = 1;
                else break;
            }
            if (best.n_levels) tiles.push_back(best);
            a = b;
        }
        const uint32_t n_pass_tiles = (uint32_t)tiles.size() - first_tile;
        ctx->passes.emplace_back(first_tile, n_pass_tiles);
        // Tile kinds.  Pass 0: roots.  A later pass whose every tile hangs below ONE node of a short enough ancestor
        // chain lets each tile re-evaluate that chain itself (kernels_tree.hip), which makes the pass independent of
        // the one above it: it joins the previous launch.  Otherwise its tiles read their parents from global memory
        // and the pass needs its own launch behind the previous one.
        // (the owners wait for the chain tiles to start, so there must be few of them and only one chained pass per launch)
        bool chainable = chain_candidate;
        for (uint32_t ti = first_tile; chainable && ti < tiles.size(); ++ti) {
            const TileDesc& td = tiles[ti];
            if (td.n_levels == 0) continue;
            const uint32_t p0 = parent_idx[td.start[0]];
            if (parent_idx[td.start[0] + td.count[0] - 1] != p0) chainable = false;  // level 0 of the tile spans several parents
        }
        chains.resize(tiles.size() * (size_t)TILE_MAX_CHAIN, 0u);
        for (uint32_t ti = first_tile; ti < tiles.size(); ++ti) {
            TileDesc& td = tiles[ti];
            td.kind = roots ? TILE_ROOTS : 0u;
            if (chainable && td.n_levels) {
                uint32_t row = parent_idx[td.start[0]], len = 0;
                while (row != MI_NO_PARENT && len < TILE_MAX_CHAIN) {
                    chains[(size_t)ti * TILE_MAX_CHAIN + len++] = row;
                    row = parent_idx[row];
                }
                td.kind = len;
            }
        }
        if (chainable) {
            ctx->groups.back().count += n_pass_tiles;
            ctx->groups.back().n_chain = n_pass_tiles;
            ctx->groups.back().owner_rows = level_offsets[l];  // the owners' rows are the prefix [0, first row of level l)
        } else {
            ctx->groups.push_back({first_tile, n_pass_tiles, 0u, 0u});
        }
        l += d;
    }
    int32_t rc;
    if ((rc = ensure(ctx, ctx->parent_idx, (size_t)n * 4))) return rc;
    if ((rc = ensure(ctx, ctx->node_flags, n))) return rc;
    if ((rc = ensure(ctx, ctx->tiles, std::max<size_t>(tiles.size(), 1) * sizeof(TileDesc)))) return rc;
    if ((rc = upload(ctx, ctx->parent_idx.p, parent_idx, (size_t)n * 4))) return rc;
    if ((rc = upload(ctx, ctx->node_flags.p, nflags.data(), n))) return rc;
    if ((rc = upload(ctx, ctx->tiles.p, tiles.data(), tiles.size() * sizeof(TileDesc)))) return rc;
    ctx->snap_rows = 0;
    for (auto& gr : ctx->groups) ctx->snap_rows = std::max(ctx->snap_rows, gr.owner_rows);
    ctx->snap_valid = false;
    if (ctx->snap_rows && (rc = ensure(ctx, ctx->snap, 2 * (size_t)ctx->snap_rows * 48))) return rc;
    if ((rc = ensure(ctx, ctx->chains, std::max<size_t>(chains.size(), 1) * 4))) return rc;
    if ((rc = upload(ctx, ctx->chains.p, chains.data(), chains.size() * 4))) return rc;
    ctx->level_offsets.assign(level_offsets, level_offsets + n_levels + 1);
    ctx->n_levels = n_levels;
    ctx->have_hierarchy = true;
    return MI_OK;
}

// =============================================================================================
// systems
// =============================================================================================
int32_t mi_propagate(mi_ctx* ctx, uint32_t flags) {
    ENTER(ctx);
    if (ctx->n == 0) return MI_OK;
    const bool all_dirty = (flags & MI_PROPAGATE_ALL_DIRTY) != 0 || !ctx->have_changed;
    const bool static_opt = (flags & MI_PROPAGATE_STATIC_OPT) != 0;
    if (!all_dirty && !ctx->changed_maybe && (!ctx->have_hierarchy || static_opt)) {
        // No Transform was marked since the last propagate consumed the change column: every row keeps its
        // GlobalTransform (set_if_neq would compare equal, systems.rs:719) and no change tick moves.  Only the change
        // masks of the previous frame have to read as empty.  (Not with a hierarchy and the static optimisation off:
        // there the reference re-assigns every root each frame, which bumps the roots' ticks, systems.rs:522-530.)
        if (ctx->g_chg_maybe) {
            HIP_TRY(ctx, hipMemsetAsync(ctx->g_chg_bits, 0, padded_words(ctx->n) * 8, ctx->stream));
            if (ctx->g_changed_bytes) HIP_TRY(ctx, hipMemsetAsync(ctx->g_changed_bytes, 0, ctx->n, ctx->stream));
            ctx->g_chg_in_bytes = false;
            ctx->g_chg_maybe = false;
        }
        return MI_OK;
    }
    ctx->g_chg_maybe = true;
    Columns c = columns_of(ctx);
    const uint32_t n0 = ctx->have_hierarchy ? ctx->level_offsets[1] : ctx->n;
    const uint32_t* tree_bits = nullptr;
    // mark_dirty_trees returns early unless the static optimisation is enabled (systems.rs:131-133)
    if (ctx->have_hierarchy && static_opt && !all_dirty) {
        {
            ProfScope ps(ctx, K_CLEAR);
            HIP_TRY(ctx, launch_clear_u32(ctx->tree_bits, padded_words(ctx->n) * 2, ctx->stream));
        }
        ProfScope ps(ctx, K_MARK_DIRTY);
        HIP_TRY(ctx, launch_mark_dirty(ctx->n, ctx->changed, (const uint32_t*)ctx->parent_idx.p, ctx->tree_bits, ctx->stream));
        tree_bits = ctx->tree_bits;
    }
    if (!ctx->have_hierarchy) {
        ProfScope ps(ctx, K_LEVEL0_PROPAGATE);
        HIP_TRY(ctx, launch_level0_propagate(c, n0, nullptr, ctx->changed, tree_bits, all_dirty, static_opt, ctx->stream));
        ctx->g_chg_in_bytes = false;
    } else {
        float* snap_r = nullptr;
        float* snap_w = nullptr;
        if (ctx->snap_rows) {
            snap_r = (float*)ctx->snap.p + (size_t)ctx->snap_parity * ctx->snap_rows * 12;
            snap_w = (float*)ctx->snap.p + (size_t)(ctx->snap_parity ^ 1u) * ctx->snap_rows * 12;
            if (!ctx->snap_valid) {  // first frame after a (re)plan or an external GlobalTransform upload
                HIP_TRY(ctx, hipMemcpyAsync(snap_r, ctx->g, (size_t)ctx->snap_rows * 48, hipMemcpyDeviceToDevice, ctx->stream));
                ctx->snap_valid = true;
            }
            ctx->snap_parity ^= 1u;
        }
        for (auto& gr : ctx->groups) {
            ProfScope sc(ctx, K_PROPAGATE_TILES);
            HIP_TRY(ctx, launch_propagate_tiles(c, (const uint32_t*)ctx->parent_idx.p, (const TileDesc*)ctx->tiles.p + gr.first,
                                                (const uint32_t*)ctx->chains.p + (size_t)gr.first * TILE_MAX_CHAIN, gr.count,
                                                (const uint8_t*)ctx->node_flags.p, ctx->changed, tree_bits, ctx->g_changed_bytes,
                                                gr.n_chain ? snap_r : nullptr, gr.n_chain ? snap_w : nullptr, all_dirty, static_opt,
                                                ctx->stream));
        }
        ctx->g_chg_in_bytes = true;
    }
    if (ctx->have_changed && ctx->changed_maybe) {
        HIP_TRY(ctx, hipMemsetAsync(ctx->changed, 0, ctx->n, ctx->stream));  // change flags are consumed
        ctx->changed_maybe = false;
    }
    return MI_OK;
}

int32_t mi_visibility_begin_frame(mi_ctx* ctx) {
    ENTER(ctx);
    ProfScope ps(ctx, K_VIS_BEGIN);
    HIP_TRY(ctx, launch_vis_begin(columns_of(ctx), ctx->stream));
    return MI_OK;
}

int32_t mi_visibility_end_frame(mi_ctx* ctx) {
    ENTER(ctx);
    ProfScope ps(ctx, K_VIS_END);
    HIP_TRY(ctx, launch_vis_end(columns_of(ctx), ctx->stream));
    return MI_OK;
}

namespace {
void simple_views(std::vector<mi_view>& v, const float* frusta, const uint32_t* masks, const uint8_t* vflags, uint32_t n_views) {
    v.resize(frusta ? n_views : 0);
    for (uint32_t i = 0; i < v.size(); ++i) {
        memset(&v[i], 0, sizeof(mi_view));
        memcpy(v[i].frustum, frusta + 24 * (size_t)i, sizeof v[i].frustum);
        v[i].layer_mask = masks ? masks[i] : 1u;
        v[i].flags = vflags ? vflags[i] : 0u;
    }
}
}  // namespace

int32_t mi_cull_views(mi_ctx* ctx, const mi_view* views, uint32_t n_views, uint32_t flags) {
    ENTER(ctx);
    VisibilityOut vo{};
    int32_t rc = exchange_begin(ctx);
    if (rc) return rc;
    if ((rc = prepare_views(ctx, views, n_views, &vo))) return rc;
    SegOut seg;
    if ((rc = prepare_segments(ctx, n_views, &seg))) return rc;
    Columns c = columns_of(ctx);
    {
        ProfScope ps(ctx, K_CULL);
        HIP_TRY(ctx, launch_cull(c, ctx->views_inline ? &ctx->view_set : nullptr, (const ViewParams*)ctx->views.p, n_views, vo,
                                 seg, flags & (MI_CULL_BEGIN_FRAME | MI_CULL_END_FRAME), ctx->stream));
    }
    if ((rc = run_compaction(ctx, vo, seg))) return rc;
    ctx->culled = true;
    return exchange_end(ctx);
}

int32_t mi_cull(mi_ctx* ctx, const float* frusta, const uint32_t* view_layer_masks, const uint8_t* view_flags,
                uint32_t n_views, uint32_t flags) {
    std::vector<mi_view> v;
    simple_views(v, frusta, view_layer_masks, view_flags, n_views);
    return mi_cull_views(ctx, v.empty() ? nullptr : v.data(), n_views, flags);
}

int32_t mi_propagate_and_cull_views(mi_ctx* ctx, const mi_view* views, uint32_t n_views, uint32_t flags) {
    ENTER(ctx);
    if (ctx->have_hierarchy)
        return fail(ctx, MI_ERR_NOT_READY, "mi_propagate_and_cull is the flat fast path; a hierarchy is uploaded -- use mi_propagate + mi_cull");
    VisibilityOut vo{};
    int32_t rc = exchange_begin(ctx);
    if (rc) return rc;
    if ((rc = prepare_views(ctx, views, n_views, &vo))) return rc;
    SegOut seg;
    if ((rc = prepare_segments(ctx, n_views, &seg))) return rc;
    Columns c = columns_of(ctx);
    {
        ProfScope ps(ctx, K_FLAT_PROPAGATE_CULL);
        HIP_TRY(ctx, launch_flat_propagate_cull(c, ctx->views_inline ? &ctx->view_set : nullptr, (const ViewParams*)ctx->views.p,
                                                n_views, vo, seg, flags & MI_CULL_END_FRAME, ctx->stream));
    }
    if ((rc = run_compaction(ctx, vo, seg))) return rc;
    if (ctx->have_changed && ctx->changed_maybe) {
        HIP_TRY(ctx, hipMemsetAsync(ctx->changed, 0, ctx->n, ctx->stream));
        ctx->changed_maybe = false;
    }
    ctx->g_chg_maybe = true;
    ctx->g_chg_in_bytes = false;
    ctx->culled = true;
    return exchange_end(ctx);
}

int32_t mi_propagate_and_cull(mi_ctx* ctx, const float* frusta, const uint32_t* view_layer_masks, const uint8_t* view_flags,
                              uint32_t n_views, uint32_t flags) {
    std::vector<mi_view> v;
    simple_views(v, frusta, view_layer_masks, view_flags, n_views);
    return mi_propagate_and_cull_views(ctx, v.empty() ? nullptr : v.data(), n_views, flags);
}

// visibility_propagate_system (crates/bevy_camera/src/visibility/mod.rs:638-729) over the uploaded hierarchy
int32_t mi_visibility_propagate(mi_ctx* ctx) {
    ENTER(ctx);
    if (ctx->n == 0) return MI_OK;
    if (!ctx->have_hierarchy) {
        ProfScope ps(ctx, K_INHERIT);
        HIP_TRY(ctx, launch_inherit_flat(ctx->n, ctx->visibility, ctx->flags, ctx->inh_changed, ctx->stream));
        return MI_OK;
    }
    bool first = true;
    for (auto& ps : ctx->passes) {
        ProfScope sc(ctx, K_INHERIT);
        HIP_TRY(ctx, launch_inherit_tiles((const uint32_t*)ctx->parent_idx.p, (const TileDesc*)ctx->tiles.p + ps.first, ps.second,
                                          first, ctx->visibility, ctx->flags, ctx->inh_changed, ctx->stream));
        first = false;
    }
    return MI_OK;
}

int32_t mi_download_inherited_visibility(mi_ctx* ctx, uint32_t first_row, uint32_t n, uint8_t* out_inherited,
                                         uint32_t* changed_bitmask) {
    ENTER(ctx);
    int32_t rc = check_rows(ctx, first_row, n, "mi_download_inherited_visibility");
    if (rc) return rc;
    if (changed_bitmask && (first_row & 31u)) return fail(ctx, MI_ERR_INVALID_ARG, "first_row must be a multiple of 32 for the change bitmask");
    if (out_inherited) {
        if ((rc = download(ctx, out_inherited, ctx->flags + first_row, n))) return rc;
        for (uint32_t i = 0; i < n; ++i) out_inherited[i] &= 1u;
    }
    if (changed_bitmask) {
        if ((rc = ensure(ctx, ctx->inh_bits, padded_words(ctx->cap) * 8 + 256))) return rc;
        HIP_TRY(ctx, launch_bytes_to_bits(ctx->inh_changed, ctx->n, (uint64_t*)ctx->inh_bits.p, ctx->stream));
        const size_t words32 = ((size_t)n + 31) / 32;
        if ((rc = download(ctx, changed_bitmask, (const uint32_t*)ctx->inh_bits.p + first_row / 32, words32 * 4))) return rc;
        if (n & 31u) changed_bitmask[words32 - 1] &= (1u << (n & 31u)) - 1u;
    }
    return MI_OK;
}

// =============================================================================================
// results
// =============================================================================================
int32_t mi_download_global_transforms(mi_ctx* ctx, uint32_t first_row, uint32_t n, float* out, uint32_t* changed_bitmask) {
    ENTER(ctx);
    int32_t rc = check_rows(ctx, first_row, n, "mi_download_global_transforms");
    if (rc) return rc;
    if (changed_bitmask && (first_row & 31u)) return fail(ctx, MI_ERR_INVALID_ARG, "first_row must be a multiple of 32 for the change bitmask");
    if (out && (rc = download(ctx, out, ctx->g + 12 * (size_t)first_row, (size_t)n * 48))) return rc;
    if (changed_bitmask) {
        if (ctx->g_chg_in_bytes) {  // the tree path records one byte per row; pack on demand
            HIP_TRY(ctx, launch_bytes_to_bits(ctx->g_changed_bytes, ctx->n, ctx->g_chg_bits, ctx->stream));
            ctx->g_chg_in_bytes = false;
        }
        const size_t words32 = ((size_t)n + 31) / 32;
        if ((rc = download(ctx, changed_bitmask, (const uint32_t*)ctx->g_chg_bits + first_row / 32, words32 * 4))) return rc;
        if (n & 31u) changed_bitmask[words32 - 1] &= (1u << (n & 31u)) - 1u;
    }
    return MI_OK;
}

// Compacts the GlobalTransform change mask into an ascending row list on the device; *total = its length.
static int32_t changed_rows_on_device(mi_ctx* ctx, uint32_t* total) {
    int32_t rc;
    if (ctx->g_chg_in_bytes) {
        HIP_TRY(ctx, launch_bytes_to_bits(ctx->g_changed_bytes, ctx->n, ctx->g_chg_bits, ctx->stream));
        ctx->g_chg_in_bytes = false;
    }
    const uint32_t n_waves = (uint32_t)((padded_words(ctx->cap) + 63u) / 64u * 64u);
    if ((rc = ensure(ctx, ctx->sparse_cnt, n_waves))) return rc;
    if ((rc = ensure(ctx, ctx->sparse_rows, (size_t)ctx->cap * 4))) return rc;
    if ((rc = ensure(ctx, ctx->sparse_total, 16))) return rc;
    HIP_TRY(ctx, launch_popcount_words(ctx->g_chg_bits, ctx->n, (uint8_t*)ctx->sparse_cnt.p, ctx->stream));
    CompactFastArgs f{};
    f.n = ctx->n;
    f.n_segments = 1;
    f.n_classes = 1;
    f.n_waves = n_waves;
    f.wave_cnt = (const uint8_t*)ctx->sparse_cnt.p;
    f.seg_mask = ctx->g_chg_bits;
    f.seg_words = padded_words(ctx->cap);
    f.out_rows = (uint32_t*)ctx->sparse_rows.p;
    f.seg_stride = ctx->cap;
    f.seg_totals = (uint32_t*)ctx->sparse_total.p;
    HIP_TRY(ctx, launch_compact_fast(f, ctx->stream));
    return download(ctx, total, ctx->sparse_total.p, 4);
}

int32_t mi_download_changed_global_transforms(mi_ctx* ctx, uint32_t* out_rows, float* out_global12, uint32_t capacity,
                                              uint32_t* out_count) {
    ENTER(ctx);
    if (!out_count) return fail(ctx, MI_ERR_INVALID_ARG, "mi_download_changed_global_transforms: out_count NULL");
    *out_count = 0;
    if (ctx->n == 0) return MI_OK;
    int32_t rc;
    uint32_t total = 0;
    if ((rc = changed_rows_on_device(ctx, &total))) return rc;
    *out_count = total;
    if (total > capacity) return fail(ctx, MI_ERR_CAPACITY, "%u GlobalTransforms changed, capacity %u", total, capacity);
    if (total == 0) return MI_OK;
    if (out_rows && (rc = download(ctx, out_rows, ctx->sparse_rows.p, (size_t)total * 4))) return rc;
    if (out_global12) {
        if ((rc = ensure(ctx, ctx->sparse_g, (size_t)total * 48))) return rc;
        HIP_TRY(ctx, launch_gather_global((const uint32_t*)ctx->sparse_rows.p, (const uint32_t*)ctx->sparse_total.p, total, ctx->g,
                                          (float*)ctx->sparse_g.p, ctx->stream));
        if ((rc = download(ctx, out_global12, ctx->sparse_g.p, (size_t)total * 48))) return rc;
    }
    return MI_OK;
}

int32_t mi_download_changed_mesh_inputs(mi_ctx* ctx, uint32_t* out_rows, float* out_world_from_local12, float* out_culling8,
                                        uint32_t capacity, uint32_t* out_count) {
    ENTER(ctx);
    if (!out_count) return fail(ctx, MI_ERR_INVALID_ARG, "mi_download_changed_mesh_inputs: out_count NULL");
    *out_count = 0;
    if (ctx->n == 0) return MI_OK;
    int32_t rc;
    uint32_t total = 0;
    if ((rc = changed_rows_on_device(ctx, &total))) return rc;
    *out_count = total;
    if (total > capacity) return fail(ctx, MI_ERR_CAPACITY, "%u GlobalTransforms changed, capacity %u", total, capacity);
    if (total == 0) return MI_OK;
    if (out_rows && (rc = download(ctx, out_rows, ctx->sparse_rows.p, (size_t)total * 4))) return rc;
    if (out_world_from_local12 || out_culling8) {
        if ((rc = ensure(ctx, ctx->sparse_g, (size_t)total * 80))) return rc;
        float* wfl = (float*)ctx->sparse_g.p;
        float* cull = wfl + 12 * (size_t)total;
        HIP_TRY(ctx, launch_gather_mesh_inputs((const uint32_t*)ctx->sparse_rows.p, (const uint32_t*)ctx->sparse_total.p, total,
                                               columns_of(ctx), wfl, cull, ctx->stream));
        if (out_world_from_local12 && (rc = download(ctx, out_world_from_local12, wfl, (size_t)total * 48))) return rc;
        if (out_culling8 && (rc = download(ctx, out_culling8, cull, (size_t)total * 32))) return rc;
    }
    return MI_OK;
}

int32_t mi_download_visibility(mi_ctx* ctx, uint32_t view, uint32_t* bitmask) {
    ENTER(ctx);
    if (!ctx->culled) return fail(ctx, MI_ERR_NOT_READY, "mi_download_visibility before mi_cull");
    if (view >= ctx->n_views || !bitmask) return fail(ctx, MI_ERR_INVALID_ARG, "mi_download_visibility: bad view or NULL");
    const uint64_t* base;
    if (ctx->ext_bitmask) base = (const uint64_t*)ctx->ext_bitmask + view * ctx->ext_words_per_view + ctx->ext_word_offset;
    else base = (const uint64_t*)ctx->bitmask.p + view * ctx->words_per_view;
    const size_t words32 = ((size_t)ctx->n + 31) / 32;
    int32_t rc = download(ctx, bitmask, base, words32 * 4);
    if (rc) return rc;
    if (ctx->n & 31u) bitmask[words32 - 1] &= (1u << (ctx->n & 31u)) - 1u;
    return MI_OK;
}

int32_t mi_download_view_visibility(mi_ctx* ctx, uint32_t first_row, uint32_t n, uint8_t* out_vv, uint32_t* changed_bitmask) {
    ENTER(ctx);
    int32_t rc = check_rows(ctx, first_row, n, "mi_download_view_visibility");
    if (rc) return rc;
    if (changed_bitmask && (first_row & 31u)) return fail(ctx, MI_ERR_INVALID_ARG, "first_row must be a multiple of 32 for the change bitmask");
    if (out_vv && (rc = download(ctx, out_vv, ctx->vv + first_row, n))) return rc;
    if (changed_bitmask) {
        const size_t words32 = ((size_t)n + 31) / 32;
        if ((rc = download(ctx, changed_bitmask, (const uint32_t*)ctx->vv_chg_bits + first_row / 32, words32 * 4))) return rc;
        if (n & 31u) changed_bitmask[words32 - 1] &= (1u << (n & 31u)) - 1u;
    }
    return MI_OK;
}

int32_t mi_download_visible_entities(mi_ctx* ctx, uint32_t view, uint32_t class_bit, uint64_t* out_keys, uint32_t* out_rows,
                                     uint32_t capacity, uint32_t* out_count) {
    ENTER(ctx);
    if (!ctx->culled) return fail(ctx, MI_ERR_NOT_READY, "mi_download_visible_entities before mi_cull");
    if (view >= ctx->compact_views || !out_count) return fail(ctx, MI_ERR_INVALID_ARG, "mi_download_visible_entities: bad view or NULL out_count");
    uint32_t slot = 0xFFFFFFFFu;
    for (uint32_t k = 0; k < ctx->compact_classes; ++k)
        if (ctx->class_bits[k] == class_bit) slot = k;
    if (slot == 0xFFFFFFFFu) {  // no row carries this class: VisibleEntities::get() returns &[]
        *out_count = 0;
        return MI_OK;
    }
    const uint32_t seg = view * ctx->compact_classes + slot;
    uint32_t total = 0;
    uint64_t base = 0;
    int32_t rc;
    if ((rc = download(ctx, &total, (const uint32_t*)ctx->seg_totals.p + seg, 4))) return rc;
    *out_count = total;
    if (total > capacity) return fail(ctx, MI_ERR_CAPACITY, "visible list has %u entries, capacity %u", total, capacity);
    if (ctx->compact_fast) {
        // rows are in key order already; keys are looked up in the host copy of the key column
        base = (uint64_t)seg * ctx->seg_stride;
        std::vector<uint32_t> tmp;
        uint32_t* rows = out_rows;
        if (!rows && out_keys) { tmp.resize(total); rows = tmp.data(); }
        if (rows && (rc = download(ctx, rows, (const uint32_t*)ctx->out_rows.p + base, (size_t)total * 4))) return rc;
        if (out_keys)
            for (uint32_t i = 0; i < total; ++i) out_keys[i] = ctx->have_keys ? ctx->h_keys[rows[i]] : (uint64_t)rows[i];
        return MI_OK;
    }
    if ((rc = download(ctx, &base, (const uint64_t*)ctx->seg_bases.p + seg, 8))) return rc;
    if (out_keys && (rc = download(ctx, out_keys, (const uint64_t*)ctx->out_keys.p + base, (size_t)total * 8))) return rc;
    if (out_rows && (rc = download(ctx, out_rows, (const uint32_t*)ctx->out_rows.p + base, (size_t)total * 4))) return rc;
    return MI_OK;
}

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
    a.list_count = (const uint32_t*)ctx->seg_totals.p + seg;
    if (ctx->compact_fast) {
        a.list = (const uint32_t*)ctx->out_rows.p + (size_t)seg * ctx->seg_stride;
        a.list_base = nullptr;
    } else {
        a.list = (const uint32_t*)ctx->out_rows.p;
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
    if (any_spot && (!spot_dir || !spot_sin_cos)) return fail(ctx, MI_ERR_INVALID_ARG, "spot lights need spot_dir and spot_sin_cos");
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
    ctx->cl_have_spot = spot_dir && spot_sin_cos;
    if (ctx->cl_have_spot) {
        if ((rc = ensure(ctx, ctx->cl_dir, (size_t)n * 12))) return rc;
        if ((rc = upload(ctx, ctx->cl_dir.p, spot_dir, (size_t)n * 12))) return rc;
        if ((rc = ensure(ctx, ctx->cl_sincos, (size_t)n * 8))) return rc;
        if ((rc = upload(ctx, ctx->cl_sincos.p, spot_sin_cos, (size_t)n * 8))) return rc;
    }
    ctx->cl_any_spot = any_spot;
    ctx->cl_n = n;
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
    if ((rc = ensure(ctx, ctx->cl_planes, (size_t)(nx + ny + nz) * 16))) return rc;
    float* base = (float*)ctx->cl_planes.p;
    if ((rc = upload(ctx, base, view->x_planes, (size_t)nx * 16))) return rc;
    if ((rc = upload(ctx, base + 4 * (size_t)nx, view->y_planes, (size_t)ny * 16))) return rc;
    if ((rc = upload(ctx, base + 4 * (size_t)(nx + ny), view->z_planes, (size_t)nz * 16))) return rc;
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
    if (view->cluster_spheres) {
        if ((rc = ensure(ctx, ctx->cl_spheres, (size_t)C * 16))) return rc;
        if ((rc = upload(ctx, ctx->cl_spheres.p, view->cluster_spheres, (size_t)C * 16))) return rc;
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
    o.spot_dir = ctx->cl_have_spot ? (const float*)ctx->cl_dir.p : nullptr;
    o.spot_sin_cos = ctx->cl_have_spot ? (const float*)ctx->cl_sincos.p : nullptr;
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

// =============================================================================================
// interop, timing
// =============================================================================================
int32_t mi_bind_visibility_output(mi_ctx* ctx, void* device_ptr, uint64_t words_per_view, uint64_t word_offset) {
    ENTER(ctx);
    ctx->ext_bitmask = device_ptr;
    ctx->ext_words_per_view = words_per_view;
    ctx->ext_word_offset = word_offset;
    ctx->culled = false;
    return MI_OK;
}

int32_t mi_exchange_configure(mi_ctx* ctx, void* nccl_comm, void* fn_nccl_all_gather, void* const* device_bufs, uint32_t n_bufs,
                              uint64_t words_per_view, uint64_t word_offset, uint64_t block_bytes, uint32_t rank) {
    void* comms[1] = {nccl_comm};
    return mi_exchange_configure_multi(ctx, nccl_comm ? comms : nullptr, nccl_comm ? 1u : 0u, fn_nccl_all_gather, device_bufs, n_bufs,
                                       words_per_view, word_offset, block_bytes, rank);
}

int32_t mi_exchange_configure_multi(mi_ctx* ctx, void* const* nccl_comms, uint32_t n_comms, void* fn_nccl_all_gather,
                                    void* const* device_bufs, uint32_t n_bufs, uint64_t words_per_view, uint64_t word_offset,
                                    uint64_t block_bytes, uint32_t rank) {
    ENTER(ctx);
    auto& x = ctx->xch;
    void* const nccl_comm = (nccl_comms && n_comms) ? nccl_comms[0] : nullptr;
    if (n_comms > mi_ctx::Exchange::MAX_COMMS) return fail(ctx, MI_ERR_INVALID_ARG, "mi_exchange_configure: at most %u communicators", mi_ctx::Exchange::MAX_COMMS);
    for (uint32_t k = 0; k < n_comms; ++k)
        if (!nccl_comms[k]) return fail(ctx, MI_ERR_INVALID_ARG, "mi_exchange_configure: communicator %u is NULL", k);
    if (x.on) {  // drain whatever is in flight before changing anything
        int32_t rc0 = exchange_wait_issued(ctx, x.frame);
        exchange_stop(ctx);
        for (hipStream_t cs : x.comm_stream)
            if (cs) HIP_TRY(ctx, hipStreamSynchronize(cs));
        x.on = false;
        if (rc0) return rc0;
    }
    if (!nccl_comm) {  // off: back to the internal mask buffer
        x.on = false;
        ctx->ext_bitmask = nullptr;
        ctx->culled = false;
        return MI_OK;
    }
    if (!fn_nccl_all_gather || !device_bufs || n_bufs < 2 || n_bufs > mi_ctx::Exchange::MAX_BUFS || block_bytes == 0)
        return fail(ctx, MI_ERR_INVALID_ARG, "mi_exchange_configure: NULL function / buffers, n_bufs outside 2..8, or empty block");
    for (uint32_t i = 0; i < n_bufs; ++i)
        if (!device_bufs[i]) return fail(ctx, MI_ERR_INVALID_ARG, "mi_exchange_configure: buffer %u is NULL", i);
    if (!x.done_flag) HIP_TRY(ctx, hipHostMalloc((void**)&x.done_flag, 64, hipHostMallocMapped));
    if (!x.kernels_flag) HIP_TRY(ctx, hipMalloc((void**)&x.kernels_flag, 64));
    HIP_TRY(ctx, hipMemsetAsync(x.kernels_flag, 0, 64, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    x.kernel_signal = getenv("MI_XCH_NO_KERNEL_SIGNAL") == nullptr;
    x.signalled = false;
    if (!x.comm_stream[0]) {
        for (uint32_t i = 0; i < mi_ctx::Exchange::MAX_BUFS; ++i) {
            HIP_TRY(ctx, hipEventCreateWithFlags(&x.ev_kernels[i], hipEventDisableTiming));
            HIP_TRY(ctx, hipEventCreateWithFlags(&x.ev_gathered[i], hipEventDisableTiming));
        }
        // Pick the communication stream empirically.  HIP maps streams onto a small pool of hardware queues; if the
        // communication stream lands on the compute stream's queue, its wait / record / write packets serialise with
        // the frame kernels (measured: 32 us -> 48 us per frame), and which stream collides depends on how many
        // streams the process created before.  So: make a few candidates (normal and high priority), drive each with
        // the per-frame pattern over a stand-in kernel, keep the fastest.
        int prio_lo = 0, prio_hi = 0;
        HIP_TRY(ctx, hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
        constexpr int N_CAND = 6;
        hipStream_t cand[N_CAND] = {nullptr};
        for (int i = 0; i < N_CAND; ++i) {
            if (i == N_CAND - 1) HIP_TRY(ctx, hipStreamCreateWithPriority(&cand[i], hipStreamNonBlocking, prio_hi));
            else HIP_TRY(ctx, hipStreamCreateWithFlags(&cand[i], hipStreamNonBlocking));
        }
        const size_t probe_words = (size_t)16 << 20;  // 64 MB clear: a stand-in for one frame of kernels
        uint32_t* probe = nullptr;
        HIP_TRY(ctx, hipMalloc((void**)&probe, probe_words * 4));
        double cand_t[N_CAND];
        for (int rep = 0; rep < 2; ++rep)
            for (int i = 0; i < N_CAND; ++i) {
                HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
                HIP_TRY(ctx, hipStreamSynchronize(cand[i]));
                const auto t0 = std::chrono::steady_clock::now();
                for (uint32_t it = 0; it < 24; ++it) {
                    HIP_TRY(ctx, launch_clear_u32(probe, probe_words, ctx->stream));
                    HIP_TRY(ctx, hipEventRecord(x.ev_kernels[it & 1u], ctx->stream));
                    HIP_TRY(ctx, hipStreamWaitEvent(cand[i], x.ev_kernels[it & 1u], 0));
                    HIP_TRY(ctx, hipEventRecord(x.ev_gathered[it & 1u], cand[i]));
                    HIP_TRY(ctx, hipStreamWriteValue32(cand[i], (void*)x.done_flag, it + 1, 0));
                }
                HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
                HIP_TRY(ctx, hipStreamSynchronize(cand[i]));
                const double t = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                if (rep == 1) cand_t[i] = t;
                if (getenv("MI_XCH_DEBUG")) fprintf(stderr, "[mi exchange] comm stream candidate %d%s: %.1f us / frame\n", i,
                                                    i == N_CAND - 1 ? " (high priority)" : "", t / 24.0);
            }
        HIP_TRY(ctx, hipFree(probe));
        int order[N_CAND];
        std::iota(order, order + N_CAND, 0);
        std::sort(order, order + N_CAND, [&](int a, int b) { return cand_t[a] < cand_t[b]; });
        for (int i = 0; i < N_CAND; ++i) {  // keep the MAX_COMMS fastest, fastest first
            if (i < (int)mi_ctx::Exchange::MAX_COMMS) x.comm_stream[i] = cand[order[i]];
            else HIP_TRY(ctx, hipStreamDestroy(cand[order[i]]));
        }
    }
    x.all_gather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))fn_nccl_all_gather;
    x.n_comms = n_comms;
    for (uint32_t k = 0; k < n_comms; ++k) x.comm[k] = nccl_comms[k];
    x.n_bufs = n_bufs;
    for (uint32_t i = 0; i < n_bufs; ++i) x.buf[i] = device_bufs[i];
    x.words_per_view = words_per_view;
    x.word_offset = word_offset;
    x.block_bytes = block_bytes;
    x.rank = rank;
    for (uint32_t k = 0; k < mi_ctx::Exchange::MAX_COMMS; ++k) x.done_flag[k] = 0;
    x.worker_frames = 0;
    x.frame = 0;
    x.submitted = x.issued = 0;
    x.submitted_fast.store(0);
    x.worker_error = 0;
    x.queue.clear();
    x.worker = std::thread(exchange_worker, ctx);
    x.on = true;
    ctx->culled = false;
    return MI_OK;
}

int32_t mi_exchange_last(mi_ctx* ctx, void** out_device_buf, int32_t wait) {
    ENTER(ctx);
    auto& x = ctx->xch;
    if (!x.on || x.frame == 0) return fail(ctx, MI_ERR_NOT_READY, "mi_exchange_last: no exchanged frame yet");
    const uint32_t slot = (uint32_t)((x.frame - 1) % x.n_bufs);
    int32_t rc = exchange_wait_issued(ctx, x.frame);
    if (rc) return rc;
    if (wait) HIP_TRY(ctx, hipEventSynchronize(x.ev_gathered[slot]));
    if (out_device_buf) *out_device_buf = x.buf[slot];
    return MI_OK;
}

int32_t mi_device_buffer(mi_ctx* ctx, uint32_t which, void** out_ptr, uint64_t* out_bytes) {
    ENTER(ctx);
    if (!out_ptr) return fail(ctx, MI_ERR_INVALID_ARG, "mi_device_buffer: NULL");
    void* p = nullptr;
    uint64_t bytes = 0;
    switch (which) {
    case MI_BUF_GLOBAL_TRANSFORM: p = ctx->g; bytes = (uint64_t)ctx->n * 48; break;
    case MI_BUF_VISIBILITY_BITMASK:
        if (ctx->ext_bitmask) { p = ctx->ext_bitmask; bytes = ctx->ext_words_per_view * 8 * ctx->n_views; }
        else { p = ctx->bitmask.p; bytes = ctx->words_per_view * 8 * ctx->n_views; }
        break;
    case MI_BUF_VIEW_VISIBILITY: p = ctx->vv; bytes = ctx->n; break;
    case MI_BUF_VISIBLE_ROWS: p = ctx->out_rows.p; bytes = ctx->out_rows.bytes; break;
    case MI_BUF_CLUSTER_OFFSETS_AND_COUNTS: p = ctx->cl_bind_oc.p; bytes = ctx->cl_bind_oc.bytes; break;
    case MI_BUF_CLUSTER_INDEX_LIST: p = ctx->cl_bind_idx.p; bytes = ctx->cl_bind_idx.bytes; break;
    case MI_BUF_BATCH_WORK_ITEMS_NON_INDEXED: p = ctx->bt_wi[0].p; bytes = ctx->bt_wi[0].bytes; break;
    case MI_BUF_BATCH_WORK_ITEMS_INDEXED: p = ctx->bt_wi[1].p; bytes = ctx->bt_wi[1].bytes; break;
    case MI_BUF_BATCH_METADATA_NON_INDEXED: p = ctx->bt_md[0].p; bytes = ctx->bt_md[0].bytes; break;
    case MI_BUF_BATCH_METADATA_INDEXED: p = ctx->bt_md[1].p; bytes = ctx->bt_md[1].bytes; break;
    case MI_BUF_BATCH_SETS_NON_INDEXED: p = ctx->bt_bs[0].p; bytes = ctx->bt_bs[0].bytes; break;
    case MI_BUF_BATCH_SETS_INDEXED: p = ctx->bt_bs[1].p; bytes = ctx->bt_bs[1].bytes; break;
    default: return fail(ctx, MI_ERR_INVALID_ARG, "mi_device_buffer: unknown buffer %u", which);
    }
    *out_ptr = p;
    if (out_bytes) *out_bytes = bytes;
    return MI_OK;
}

int32_t mi_timer_begin(mi_ctx* ctx) {
    ENTER(ctx);
    HIP_TRY(ctx, hipEventRecord(ctx->timer_a, ctx->stream));
    return MI_OK;
}
int32_t mi_timer_end(mi_ctx* ctx, float* out_ms) {
    ENTER(ctx);
    HIP_TRY(ctx, hipEventRecord(ctx->timer_b, ctx->stream));
    HIP_TRY(ctx, hipEventSynchronize(ctx->timer_b));
    float ms = 0.f;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->timer_a, ctx->timer_b));
    if (out_ms) *out_ms = ms;
    return MI_OK;
}

int32_t mi_profile_enable(mi_ctx* ctx, int32_t enabled) {
    ENTER(ctx);
    if (!enabled) prof_collect(ctx);
    else {
        prof_collect(ctx);
        memset(ctx->prof_launches, 0, sizeof ctx->prof_launches);
        memset(ctx->prof_ms, 0, sizeof ctx->prof_ms);
        memset(ctx->prof_timed, 0, sizeof ctx->prof_timed);
    }
    ctx->profiling = enabled != 0;
    return MI_OK;
}
int32_t mi_profile_filter(mi_ctx* ctx, uint64_t kernel_mask) {
    ENTER(ctx);
    ctx->prof_mask = kernel_mask ? kernel_mask : ~0ull;
    return MI_OK;
}
int32_t mi_profile_sample(mi_ctx* ctx, uint32_t every_n) {
    ENTER(ctx);
    ctx->prof_every = every_n ? every_n : 1;
    memset(ctx->prof_tick, 0, sizeof ctx->prof_tick);
    return MI_OK;
}
int32_t mi_profile_burst(mi_ctx* ctx, uint32_t first_n) {
    ENTER(ctx);
    ctx->prof_burst = first_n;
    memset(ctx->prof_timed, 0, sizeof ctx->prof_timed);
    return MI_OK;
}
int32_t mi_profile_read(mi_ctx* ctx, uint32_t* inout_n, uint64_t* launches, double* total_ms) {
    ENTER(ctx);
    if (!inout_n) return fail(ctx, MI_ERR_INVALID_ARG, "mi_profile_read: NULL");
    prof_collect(ctx);
    const uint32_t n = std::min<uint32_t>(*inout_n, K_NUM_KERNELS);
    for (uint32_t k = 0; k < n; ++k) {
        if (launches) launches[k] = ctx->prof_launches[k];
        if (total_ms) total_ms[k] = ctx->prof_ms[k];
    }
    *inout_n = K_NUM_KERNELS;
    return MI_OK;
}
const char* mi_profile_kernel_name(uint32_t k) {
    static const char* names[K_NUM_KERNELS] = {"k_flat_propagate_cull", "k_level0_propagate", "k_cull", "k_vis_begin",
                                               "k_vis_end", "k_compact_count", "k_compact_scan", "k_compact_scatter",
                                               "k_compact_fast", "k_mark_dirty", "k_propagate_tiles", "k_cluster_walk", "k_cluster_fill",
                                               "k_clear_u32", "k_inherit", "k_batch_clear", "k_batch_hist", "k_batch_scan",
                                               "k_batch_scatter", "k_batch_bounds", "k_batch_sets", "k_batch_allocate", "k_batch_unpack"};
    return k < K_NUM_KERNELS ? names[k] : nullptr;
}

// test hook: device logf probe (not part of the public header; used by tests/test_logf.py)
int32_t mi_debug_logf(mi_ctx* ctx, const float* in, float* out, uint32_t n) {
    ENTER(ctx);
    DevBuf a, b;
    int32_t rc;
    if ((rc = ensure(ctx, a, (size_t)n * 4))) return rc;
    if ((rc = ensure(ctx, b, (size_t)n * 4))) return rc;
    if ((rc = upload(ctx, a.p, in, (size_t)n * 4))) return rc;
    HIP_TRY(ctx, launch_logf_probe((const float*)a.p, (float*)b.p, n, ctx->stream));
    rc = download(ctx, out, b.p, (size_t)n * 4);
    hipFree(a.p);
    hipFree(b.p);
    return rc;
}

}  // extern "C"
