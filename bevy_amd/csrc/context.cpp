// context.cpp -- implementation of the C ABI in include/bevy_mi355x.h on top of the gfx950 kernels.
//
// Owns: the device-resident component columns (SoA across components, packed within a component),
// a pinned staging arena for host->device copies, the hierarchy tile plan, the per-view constants,
// the visibility bitmasks / VisibleEntities lists, the clustering scratch, and HIP-event timing.
// Everything is enqueued on one HIP stream per context; the only host synchronisations are in the
// download / timer-read entry points.
#include <dlfcn.h>

#include "ctx.h"

using namespace mi;
using namespace mi_detail;

thread_local const mi::LaunchTimer* mi::g_launch_timer = nullptr;

namespace {
std::mutex g_err_mutex;
std::string g_create_error = "no error";
}  // namespace

namespace mi_detail {

// roctx, looked up once (ctx.h: TraceRange).  dlopen only: the library has no link-time dependency on a tracing runtime.
const RoctxApi& roctx_api() {
    static const RoctxApi api = [] {
        RoctxApi a;
        const char* want = getenv("MI_ROCTX");
        const char* prof = getenv("ROCPROF_MARKER_API_TRACE");
        const bool on = want ? (want[0] != '\0' && want[0] != '0') : (prof && prof[0] != '\0' && prof[0] != '0');
        if (!on) return a;
        void* push = dlsym(RTLD_DEFAULT, "roctxRangePushA");
        void* pop = dlsym(RTLD_DEFAULT, "roctxRangePop");
        for (const char* lib : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
            if (push && pop) break;
            if (void* h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL)) {
                push = dlsym(h, "roctxRangePushA");
                pop = dlsym(h, "roctxRangePop");
            }
        }
        if (push && pop) {
            a.push = (int (*)(const char*))push;
            a.pop = (int (*)())pop;
        } else if (want) {
            fprintf(stderr, "[mi] MI_ROCTX is set but no roctx library could be loaded: no ranges\n");
        }
        return a;
    }();
    return api;
}

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

int32_t stage_alloc(mi_ctx* ctx, size_t bytes, void** out) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (ctx->stage_used + bytes > ctx->stage_bytes) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // everything staged so far has been consumed
        if (ctx->cl_stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->cl_stream));
        ctx->stage_used = 0;
        ++ctx->stage_epoch;
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
    // the destination may be something an assignment on the cluster stream is still reading; and that stream has to see
    // this write before its next launch
    int32_t rc = cluster_join(ctx);
    if (rc) return rc;
    ctx->cl_inputs_dirty = true;
    if ((rc = stage_alloc(ctx, bytes, &st))) return rc;
    memcpy(st, src, bytes);
    HIP_TRY(ctx, hipMemcpyAsync(dst, st, bytes, hipMemcpyHostToDevice, ctx->stream));
    return MI_OK;
}

// The caller's slice is pageable memory: a device-to-host copy straight into it goes through the runtime's own bounce
// buffers, synchronously and in chunks (measured: 40 - 50 us per call before the first byte moves).  Through the pinned
// arena it is one DMA, one wait and one memcpy.
int32_t download(mi_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (!bytes) return MI_OK;
    if (bytes <= ((size_t)32 << 20)) {
        void* st = nullptr;
        int32_t rc = stage_alloc(ctx, bytes, &st);
        if (rc) return rc;
        HIP_TRY(ctx, hipMemcpyAsync(st, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        memcpy(dst, st, bytes);
        return MI_OK;
    }
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MI_OK;
}

int32_t consume_changed(mi_ctx* ctx) {
    if (!ctx->have_changed || !ctx->changed_maybe) return MI_OK;
    if (ctx->changed_bulk || ctx->changed_gen >= 255u) {  // plain 1s somewhere, or the stamps are about to wrap: really clear
        HIP_TRY(ctx, hipMemsetAsync(ctx->changed, 0, ctx->n, ctx->stream));
        ctx->changed_bulk = false;
        ctx->changed_gen = ctx->changed_gen >= 255u ? 2u : ctx->changed_gen + 1u;
    } else {
        ++ctx->changed_gen;  // every stamp in the column is now a past generation: nothing to launch
    }
    ctx->changed_maybe = false;
    ctx->changed_rows_hint = 0;
    return MI_OK;
}

void trs_written(mi_ctx* ctx) {
    ++ctx->trs_version;
    ctx->seq_cov = 0;
}
static int32_t piece_streams(mi_ctx* ctx) {
    if (ctx->piece_streams) return MI_OK;
    // HIP streams share a few hardware queues per priority class, and a queue runs its barrier packets (event waits and records)
    // in order: when the upload stream landed on the queue of the download stream, the latter's first wait stood behind every
    // event record of the upload (measured: the first GlobalTransform piece left when the last Transform piece had arrived,
    // profiles/r03_experiments.md 12).  A priority class of its own gives each of the two a queue no other stream of the context
    // shares.
    int prio_lo = 0, prio_hi = 0;
    HIP_TRY(ctx, hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
    HIP_TRY(ctx, hipStreamCreateWithPriority(&ctx->up_stream, hipStreamNonBlocking, prio_hi));
    HIP_TRY(ctx, hipStreamCreateWithPriority(&ctx->dn_stream, hipStreamNonBlocking, prio_lo));
    for (uint32_t k = 0; k < mi_ctx::UP_PIECES; ++k) {
        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_up[k], hipEventDisableTiming));
        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_pre[k], hipEventDisableTiming));
    }
    ctx->piece_streams = true;
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
    c.layer_mask_hi = ctx->layers_hi;
    c.view_visibility = ctx->vv;
    c.range_start_end = ctx->have_ranges ? ctx->range : nullptr;
    c.g_changed_bits = ctx->g_chg_bits;
    c.vv_changed_bits = ctx->vv_chg_bits;
    c.changed_gen = ctx->changed_gen;
    const bool rs_current = ctx->rs_lo[0] >= ctx->rs_hi[0] && ctx->rs_lo[1] >= ctx->rs_hi[1];
    c.row_summary = (const uint32_t*)ctx->row_sum.p;
    c.row_summary_on = (ctx->row_sum_mode == 0 && ctx->row_sum.p && rs_current) ? 1u : 0u;
    return c;
}

void row_summary_touch(mi_ctx* ctx, uint32_t parts, uint32_t first_row, uint32_t n_rows) {
    if (n_rows == 0) return;
    ctx->cells.valid = false;  // (the static cull order copies flags / RenderLayers / half extents per cell)
    ctx->cells.quiet = 0;
    const uint32_t lo = first_row >> 6, hi = (uint32_t)(((uint64_t)first_row + n_rows + 63u) >> 6);
    for (uint32_t p = 0; p < 2u; ++p) {
        if (!(parts & (1u << p))) continue;
        if (ctx->rs_lo[p] >= ctx->rs_hi[p]) {
            ctx->rs_lo[p] = lo;
            ctx->rs_hi[p] = hi;
        } else {
            ctx->rs_lo[p] = std::min(ctx->rs_lo[p], lo);
            ctx->rs_hi[p] = std::max(ctx->rs_hi[p], hi);
        }
    }
}

int32_t row_summary_ensure(mi_ctx* ctx) {
    if (ctx->n == 0) return MI_OK;
    const uint32_t n_waves = (uint32_t)words64(ctx->n);
    const size_t bytes = (size_t)words64(ctx->cap) * ROWSUM_WORDS * 4;
    if (!ctx->row_sum.p || ctx->row_sum.bytes < bytes) {
        int32_t rc = ensure(ctx, ctx->row_sum, bytes);
        if (rc) return rc;
        HIP_TRY(ctx, hipMemsetAsync(ctx->row_sum.p, 0, ctx->row_sum.bytes, ctx->stream));
        ctx->rs_lo[0] = ctx->rs_lo[1] = 0;
        ctx->rs_hi[0] = ctx->rs_hi[1] = n_waves;
    }
    if (ctx->row_sum_mode != 0) return MI_OK;  // (allocated all the same: the frame kernels' load of it is unconditional)
    uint32_t lo[2], hi[2];
    for (uint32_t p = 0; p < 2u; ++p) {
        lo[p] = std::min(ctx->rs_lo[p], n_waves);
        hi[p] = std::min(ctx->rs_hi[p], n_waves);
    }
    Columns c = columns_of(ctx);
    uint32_t* sum = (uint32_t*)ctx->row_sum.p;
    if (lo[0] < hi[0] && lo[1] < hi[1] && lo[0] == lo[1] && hi[0] == hi[1]) {
        HIP_TRY(ctx, launch_row_summary(c, lo[0], hi[0] - lo[0], ROWSUM_PART_AABB | ROWSUM_PART_FLAGS, sum, ctx->stream));
    } else {
        if (lo[0] < hi[0]) HIP_TRY(ctx, launch_row_summary(c, lo[0], hi[0] - lo[0], ROWSUM_PART_AABB, sum, ctx->stream));
        if (lo[1] < hi[1]) HIP_TRY(ctx, launch_row_summary(c, lo[1], hi[1] - lo[1], ROWSUM_PART_FLAGS, sum, ctx->stream));
    }
    ctx->rs_lo[0] = ctx->rs_lo[1] = ctx->rs_hi[0] = ctx->rs_hi[1] = 0;
    return MI_OK;
}

static_assert(sizeof(mi_view) == sizeof(ViewParams), "mi_view and ViewParams share one layout");

int32_t prepare_views(mi_ctx* ctx, const mi_view* views, uint32_t n_views, VisibilityOut* out) {
    if (!views || n_views == 0) return fail(ctx, MI_ERR_INVALID_ARG, "mi_cull: views NULL or n_views == 0");
    std::vector<ViewParams> vp(n_views);
    memcpy(vp.data(), views, sizeof(ViewParams) * n_views);
    for (auto& v : vp) v.pad[0] = v.pad[1] = 0;  // (mi_view::reserved; layer_mask_hi in front of it is the caller's)
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
        rc = ensure(ctx, ctx->fb[ctx->cur].bitmask, ctx->words_per_view * 8 * n_views);
        if (rc) return rc;
        out->bitmask = (uint64_t*)ctx->fb[ctx->cur].bitmask.p;
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
    if ((rc = ensure(ctx, ctx->fb[ctx->cur].wave_cnt, segs * seg->n_waves))) return rc;
    seg->wave_cnt = (uint8_t*)ctx->fb[ctx->cur].wave_cnt.p;
    if (ctx->have_class_mask) {
        seg->class_mask = ctx->class_mask;
        seg->seg_words = padded_words(ctx->cap);
        if ((rc = ensure(ctx, ctx->fb[ctx->cur].seg_mask, segs * seg->seg_words * 8))) return rc;
        seg->seg_mask = (uint64_t*)ctx->fb[ctx->cur].seg_mask.p;
    }
    return MI_OK;
}

int32_t run_compaction(mi_ctx* ctx, const VisibilityOut& vo, const SegOut& seg, uint32_t flags) {
    int32_t rc;
    const uint32_t n_classes = ctx->compact_classes;
    const size_t segs = (size_t)ctx->n_views * n_classes;
    {
        DevBuf& st = ctx->fb[ctx->cur].seg_totals;  // (+ the chunk totals of the hierarchical mode at this capacity)
        const void* before = st.p;
        if ((rc = ensure(ctx, st, compact_fast_totals_bytes(segs, ctx->cap)))) return rc;
        // Fresh memory must not hold a word that reads as a stamp -- and neither must the words of an earlier LAYOUT: the stamp table
        // starts behind the segment totals, so with fewer segments (or another chunk count) old totals and old table entries lie where
        // stamps are now read, and a total that happens to equal this frame's tag would pass for a published chunk sum.
        const uint64_t layout = ((uint64_t)segs << 32) | compact_fast_chunks(ctx->cap, true);
        uint64_t& last = ctx->fb[ctx->cur].seg_totals_layout;
        if (st.p != before || last != layout) HIP_TRY(ctx, hipMemsetAsync(st.p, 0, st.bytes, ctx->stream));
        last = layout;
    }
    if (ctx->n == 0) {
        HIP_TRY(ctx, hipMemsetAsync(ctx->fb[ctx->cur].seg_totals.p, 0, segs * 4, ctx->stream));
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
        if ((rc = ensure(ctx, ctx->fb[ctx->cur].out_rows, segs * ctx->seg_stride * 4))) return rc;
        f.out_rows = (uint32_t*)ctx->fb[ctx->cur].out_rows.p;
        f.seg_stride = ctx->seg_stride;
        f.seg_totals = (uint32_t*)ctx->fb[ctx->cur].seg_totals.p;
        if (!++ctx->compact_tag) ++ctx->compact_tag;  // (never 0: freshly allocated memory is not a valid stamp often enough to matter, 0 never)
        f.tag = ctx->compact_tag;
        // A deferred compaction reads this frame's masks while the next frame's kernel writes its own: that needs the
        // masks in alternating buffers.  The internal sets alternate, the exchange's gathered buffers rotate and the
        // per-class segment masks live in the frame set; a single caller-bound buffer (mi_bind_visibility_output
        // without the exchange) read as the segment mask does not -- compact inline then.
        const bool masks_alternate = !ctx->ext_bitmask || ctx->xch.on || seg.seg_mask != nullptr;
        if ((flags & MI_CULL_MORE_FRAMES) && masks_alternate && (!ctx->xch.on || ctx->xch.kernel_signal || ctx->xch.simple)) {
            // Another frame follows at once: this frame's compaction rides in extra workgroups of that frame's kernel (one
            // launch per frame instead of two); compaction_join launches it on its own if something else comes first.
            // With the exchange on it still publishes "this frame's masks are complete", and the frame's all-gather is
            // queued when that launch is submitted.
            ctx->defer.has_job = false;
            if (ctx->xch.on && !ctx->xch.simple) {
                auto& x = ctx->xch;
                f.signal = x.kernels_flag;
                f.signal_value = (uint32_t)(x.frame + 1);
                ctx->defer.has_job = true;
                ctx->defer.job = mi_ctx::Exchange::Job{(uint32_t)(x.frame % x.n_bufs), f.signal, f.signal_value};
                x.signalled = true;
                x.job_deferred = true;
            }
            ctx->defer.args = f;
            ctx->defer.pending = true;
            return MI_OK;
        }
        if (ctx->xch.on && !ctx->xch.simple && ctx->xch.kernel_signal) {
            f.signal = ctx->xch.kernels_flag;
            f.signal_value = (uint32_t)(ctx->xch.frame + 1);
            ctx->xch.wait_flag = f.signal;
            ctx->xch.wait_value = f.signal_value;
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
    if ((rc = ensure(ctx, ctx->fb[ctx->cur].out_rows, max_entries * 4))) return rc;
    if ((rc = ensure(ctx, ctx->out_keys, max_entries * 8))) return rc;
    a.block_counts = (uint32_t*)ctx->block_counts.p;
    a.seg_totals = (uint32_t*)ctx->fb[ctx->cur].seg_totals.p;
    a.seg_bases = (uint64_t*)ctx->seg_bases.p;
    a.out_rows = (uint32_t*)ctx->fb[ctx->cur].out_rows.p;
    a.out_keys = (uint64_t*)ctx->out_keys.p;
    HIP_TRY(ctx, launch_compact(a, ctx->stream, prof_mark, ctx));
    return MI_OK;
}

// The deferred lists of a frame over the static cull order (ctx.h, Cells::lists_pending): launched on their own.
int32_t cells_lists_join(mi_ctx* ctx) {
    auto& ce = ctx->cells;
    if (!ce.lists_pending) return MI_OK;
    ce.lists_pending = false;
    ProfScope ps(ctx, K_COMPACT_FAST);
    HIP_TRY(ctx, launch_cells_lists(ce.lists_args, ce.lists_views, ctx->stream));
    return MI_OK;
}

bool frame_begin(mi_ctx* ctx, CompactFastArgs* prev, bool* prev_has_job, mi_ctx::Exchange::Job* prev_job) {
    cells_lists_join(ctx);  // (cells_frame takes a pending job out before it gets here: its launch carries it)
    const bool have = ctx->defer.pending;
    *prev_has_job = false;
    if (have) {
        *prev = ctx->defer.args;
        *prev_has_job = ctx->defer.has_job;
        *prev_job = ctx->defer.job;
    }
    ctx->defer.pending = false;
    ctx->defer.has_job = false;
    ctx->cur = (ctx->cur + 1u) % mi_ctx::N_FB;
    ctx->fb_zero_taken = ctx->fb_zero[ctx->cur];
    ctx->fb_zero[ctx->cur].ok = false;
    return have;
}

// A cull frame failed after frame_begin took the previous frame's deferred compaction out of the context: launch it on
// its own and hand its all-gather to the exchange thread (what compaction_join would have done), so that nothing waits
// for a launch that will never be submitted.  Returns `rc` for `return frame_abort(...)`.
int32_t frame_abort(mi_ctx* ctx, int32_t rc, const CompactFastArgs* prev, bool prev_has_job, const mi_ctx::Exchange::Job& prev_job) {
    if (prev) launch_compact_fast(*prev, ctx->stream);
    if (prev_has_job) exchange_push(ctx, prev_job);
    ctx->cur = (ctx->cur + mi_ctx::N_FB - 1u) % mi_ctx::N_FB;  // the failed frame wrote nothing: the previous frame's set stays the current one
    return rc;
}

// Everything that exposes VisibleEntities (downloads, the batching build, MI_BUF_VISIBLE_ROWS, mi_synchronize) joins
// first; the join only enqueues, so device-side consumers on the context's stream are ordered behind it.
int32_t compaction_join(mi_ctx* ctx) {
    {
        const int32_t rc = cells_lists_join(ctx);
        if (rc) return rc;
    }
    if (!ctx->defer.pending) return MI_OK;
    ctx->defer.pending = false;
    {
        ProfScope ps(ctx, K_COMPACT_FAST);
        HIP_TRY(ctx, launch_compact_fast(ctx->defer.args, ctx->stream));
    }
    if (ctx->defer.has_job) exchange_push(ctx, ctx->defer.job);  // the kernel that publishes its signal is submitted now
    ctx->defer.has_job = false;
    return MI_OK;
}

}  // namespace mi_detail

// ---------------------------------------------------------------------------------------------------------------------------------
// Frames that OR their results in with atomics (the fused hierarchy frame, the static cull order): this frame's set of masks / wave
// counts and the ViewVisibility change words must be zero on entry.  They are when the previous such frame's launch zeroed them
// (fb_zero_taken / vv_alt_zeroed); else three memsets go in front.  zero[] / zero_words[] name the NEXT frame's set (same shape) for
// this frame's launch to clear; *zn is what to store in fb_zero[next] once that launch is enqueued (atomic_frame_sets_done).
// ---------------------------------------------------------------------------------------------------------------------------------
static int32_t atomic_frame_sets(mi_ctx* ctx, const VisibilityOut& vo, const SegOut& seg, uint32_t n_views, uint64_t* zero[3], uint32_t zero_words[3],
                                 mi_ctx::FbZero* zn) {
    const uint64_t bm_words = (uint64_t)n_views * vo.words_per_view, wc_bytes = seg.wave_cnt ? (uint64_t)n_views * seg.n_waves : 0;
    const size_t vv_bytes = padded_words(ctx->cap) * 8 + 256;
    const mi_ctx::FbZero& z = ctx->fb_zero_taken;
    if (!(z.ok && z.bitmask == (void*)(vo.bitmask + vo.word_offset) && z.wave_cnt == (void*)seg.wave_cnt && z.bitmask_words == bm_words && z.wave_cnt_bytes == wc_bytes)) {
        HIP_TRY(ctx, hipMemsetAsync(vo.bitmask + vo.word_offset, 0, bm_words * 8, ctx->stream));
        if (seg.wave_cnt) HIP_TRY(ctx, hipMemsetAsync(seg.wave_cnt, 0, wc_bytes, ctx->stream));
    }
    // the ViewVisibility change ticks alternate between two buffers the same way
    if (!ctx->vv_chg_alt) {
        HIP_TRY(ctx, hipMalloc((void**)&ctx->vv_chg_alt, vv_bytes));
        ctx->vv_alt_zeroed = false;
    }
    if (ctx->vv_alt_zeroed) std::swap(ctx->vv_chg_bits, ctx->vv_chg_alt);
    else HIP_TRY(ctx, hipMemsetAsync(ctx->vv_chg_bits, 0, padded_words(ctx->cap) * 8, ctx->stream));  // (the whole capacity: the row count may grow inside it)
    ctx->vv_alt_zeroed = false;
    // the next frame's set (same shape), zeroed by this frame's launch
    const uint32_t nx = (ctx->cur + 1u) % mi_ctx::N_FB;
    *zn = mi_ctx::FbZero{};
    for (int k = 0; k < 3; ++k) zero[k] = nullptr, zero_words[k] = 0;
    int32_t rc = MI_OK;
    if (!ctx->ext_bitmask && !(rc = ensure(ctx, ctx->fb[nx].bitmask, bm_words * 8)) && (!wc_bytes || !(rc = ensure(ctx, ctx->fb[nx].wave_cnt, wc_bytes)))) {
        zn->bitmask = ctx->fb[nx].bitmask.p;
        zn->wave_cnt = wc_bytes ? ctx->fb[nx].wave_cnt.p : nullptr;
        zn->bitmask_words = bm_words;
        zn->wave_cnt_bytes = wc_bytes;
        zero[0] = (uint64_t*)zn->bitmask;
        zero_words[0] = (uint32_t)bm_words;
        zero[1] = (uint64_t*)zn->wave_cnt;
        zero_words[1] = (uint32_t)(wc_bytes / 8);  // (n_waves is a multiple of 64)
        zero[2] = ctx->vv_chg_alt;
        zero_words[2] = (uint32_t)padded_words(ctx->cap);
    }
    return rc;
}
static void atomic_frame_sets_done(mi_ctx* ctx, bool zeroed_next, mi_ctx::FbZero zn) {
    if (!zeroed_next) return;
    zn.ok = true;
    ctx->fb_zero[(ctx->cur + 1u) % mi_ctx::N_FB] = zn;
    ctx->vv_alt_zeroed = true;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The static cull order (kernels_cells.hip): which frames take it, its build, the frame.
// Eligible: a cull-only frame that is the whole visibility frame (MI_CULL_BEGIN_FRAME | MI_CULL_END_FRAME) of camera views, on a context
// whose world-sphere column is current for every row (no GlobalTransform or bound changed since a k_frame_sph frame refreshed it),
// with nothing that wants the frame kernels' own by-products: no VisibilityClass segments, no visibility ranges, library-owned masks,
// no exchange, no cluster assignment in the call.  The SECOND such frame in a row builds the order (a scene that stops for one frame
// does not pay for it); every frame that takes another kernel, and every write to ViewVisibility / flags / bounds, drops it.
// ---------------------------------------------------------------------------------------------------------------------------------
static bool cells_frame_eligible(mi_ctx* ctx, const mi_view* views, uint32_t n_views, uint32_t flags, bool* build) {
    auto& ce = ctx->cells;
    *build = false;
    bool ok = views && n_views && n_views <= SPH_MAX_VIEWS && ce.mode != 1 && ctx->sph_mode != 1 && ctx->n &&
              ctx->n >= (ce.mode >= 2 ? 1u : ce.min_rows) && (flags & (MI_CULL_BEGIN_FRAME | MI_CULL_END_FRAME)) == (MI_CULL_BEGIN_FRAME | MI_CULL_END_FRAME) &&
              !(flags & (MI_CULL_WITH_CLUSTERS | MI_CULL_CHANGED_ROWS)) && !ctx->have_class_mask && !ctx->have_ranges && !ctx->ext_bitmask && !ctx->xch.on &&
              ctx->sph_state == mi_ctx::SPH_VALID && ctx->sph.p;
    for (uint32_t v = 0; ok && v < n_views; ++v) ok = !(views[v].flags & MI_VIEW_FLAG_SHADOW);
    if (!ok) {
        ce.quiet = 0;
        return false;
    }
    if (ce.valid) return true;
    if (ce.mode >= 2 || ++ce.quiet >= 2u) {
        *build = true;
        return true;
    }
    return false;
}

static int32_t cells_build(mi_ctx* ctx) {
    auto& ce = ctx->cells;
    const uint32_t n = ctx->n, n_waves = (uint32_t)words64(n);
    const size_t slots = (size_t)n_waves * 64;
    int32_t rc;
    if ((rc = ensure(ctx, ce.perm, slots * 4)) || (rc = ensure(ctx, ce.sph_s, slots * 16)) || (rc = ensure(ctx, ce.g_s, slots * 48)) ||
        (rc = ensure(ctx, ce.vv_s, slots)) || (rc = ensure(ctx, ce.sum_a, (size_t)n_waves * 16)) || (rc = ensure(ctx, ce.sum_b, (size_t)n_waves * 16)) ||
        (rc = ensure(ctx, ce.sum_h, (size_t)n_waves * 16)) || (rc = ensure(ctx, ce.state, (size_t)n_waves * 4)) || (rc = ensure(ctx, ce.keys_a, (size_t)n * 4)) ||
        (rc = ensure(ctx, ce.keys_b, (size_t)n * 4)) || (rc = ensure(ctx, ce.vals_a, (size_t)n * 4)) || (rc = ensure(ctx, ce.vals_b, (size_t)n * 4)) ||
        (rc = ensure(ctx, ce.minmax, 64)) || (rc = ensure(ctx, ce.work, (size_t)n_waves * 16)) || (rc = ensure(ctx, ce.work_n, 64)) ||
        (rc = ensure(ctx, ce.pass_s, slots * 4)))
        return rc;
    HIP_TRY(ctx, hipMemsetAsync(ce.work_n.p, 0, 64, ctx->stream));
    ce.work_parity = 0;
    ce.chain_ok = false;  // (the order's pass_s is all zero: its first frame starts from zeroed masks)
    const size_t tmp = cells_sort_temp_bytes(n);
    if ((rc = ensure(ctx, ce.sort_tmp, tmp ? tmp : 16))) return rc;
    CellsOrder o{n_waves, (uint32_t*)ce.perm.p, (float4*)ce.sph_s.p, (float*)ce.g_s.p, (uint8_t*)ce.vv_s.p, (uint32_t*)ce.pass_s.p, (float4*)ce.sum_a.p,
                 (uint4*)ce.sum_b.p, (float4*)ce.sum_h.p, (uint32_t*)ce.state.p};
    const Columns c = columns_of(ctx);
    HIP_TRY(ctx, launch_cells_build(c, (const float*)ctx->sph.p, o, (uint32_t*)ce.minmax.p, (uint32_t*)ce.keys_a.p, (uint32_t*)ce.keys_b.p,
                                    (uint32_t*)ce.vals_a.p, (uint32_t*)ce.vals_b.p, ce.sort_tmp.p, tmp, ctx->stream));
    ce.n_waves = n_waves;
    ce.valid = true;
    ce.quiet = 0;
    ++ce.builds;
    return MI_OK;
}

static int32_t cells_frame(mi_ctx* ctx, const mi_view* views, uint32_t n_views, uint32_t flags, bool build) {
    auto& ce = ctx->cells;
    // the lists the frame before deferred ride in this frame's launch -- unless this frame is of another shape (its scratch may be
    // reallocated) or builds a new order: then they go first, on their own
    if (ce.lists_pending && (build || ce.lists_views != n_views)) {
        const int32_t jrc = cells_lists_join(ctx);
        if (jrc) return jrc;
    }
    const bool ride_lists = ce.lists_pending;
    const CellsFinishArgs lists_job = ce.lists_args;
    const uint32_t lists_views = ce.lists_views;
    ce.lists_pending = false;  // (every way out below either carries the job in the frame's launch or launches it: lists_out)
    auto lists_out = [&]() {
        if (ride_lists && launch_cells_lists(lists_job, lists_views, ctx->stream) != hipSuccess)
            fail(ctx, MI_ERR_DEVICE, "the deferred lists of the frame before could not be launched");  // (recorded; the caller returns its own failure)
    };
    VisibilityOut vo{};
    CompactFastArgs prev_args{};
    bool prev_has_job = false;
    mi_ctx::Exchange::Job prev_job{};
    const CompactFastArgs* prev = frame_begin(ctx, &prev_args, &prev_has_job, &prev_job) ? &prev_args : nullptr;
    // A failure BEFORE the frame's launch gives back what the launch was to carry (the deferred lists, the previous frame's compaction
    // and its all-gather job); one BEHIND it still hands the all-gather job on: the launch that publishes its signal is submitted.
    auto give_back = [&](int32_t why) {
        lists_out();
        return frame_abort(ctx, why, prev, prev_has_job, prev_job);
    };
    auto behind_the_launch = [&](int32_t why) {
        ce.valid = false;
        if (prev_has_job) exchange_push(ctx, prev_job);
        return why;
    };
    int32_t rc = exchange_begin(ctx);
    if (rc || (rc = prepare_views(ctx, views, n_views, &vo))) return give_back(rc);
    SegOut seg;
    if ((rc = prepare_segments(ctx, n_views, &seg))) return give_back(rc);
    if (build && (rc = cells_build(ctx))) {
        ce.valid = false;
        return give_back(rc);
    }
    // ---- this frame's masks: the frame before's (its k_cells_blocks copied them into this set) -- or zero ----
    const uint64_t bm_words = (uint64_t)n_views * vo.words_per_view;
    const bool chained = ce.chain_ok && ce.chain_mask == (const void*)(vo.bitmask + vo.word_offset) && ce.chain_words == bm_words && ce.chain_views == n_views;
    ce.chain_ok = false;
    if (!chained) {
        if (!build && ce.frames) {  // pass_s describes masks this frame does not continue (another number of views, a frame in between whose
            // counts never ran): the frame starts from zeroed masks and zeroed contributions (CellsWork::fresh: pass_s is not read, every
            // processed slot rewrites it; the cells k_cells_test skips -- nothing visible, nothing was -- hold zeros already, and the
            // memset makes that independent of how they got there).  Round 4 rebuilt the whole order here: bounds, keys, a radix sort
            // of n rows and the gather, just to reach this state -- every frame, for a caller that alternates between view counts.
            if ((rc = hip_rc(ctx, hipMemsetAsync(ce.pass_s.p, 0, (size_t)ce.n_waves * 64 * 4, ctx->stream), "zeroing the cull order's contributions")))
                return give_back(rc);
        }
        if ((rc = hip_rc(ctx, hipMemsetAsync(vo.bitmask + vo.word_offset, 0, bm_words * 8, ctx->stream), "zeroing the frame's masks"))) return give_back(rc);
    }
    // the ViewVisibility change ticks are ORed into a zeroed buffer: two alternate, each launch zeroes the other one (as in the fused
    // hierarchy frame, atomic_frame_sets)
    CellsZero cz{};
    {
        const size_t vv_bytes = padded_words(ctx->cap) * 8 + 256;
        if (!ctx->vv_chg_alt) {
            if ((rc = hip_rc(ctx, hipMalloc((void**)&ctx->vv_chg_alt, vv_bytes), "the second ViewVisibility change mask"))) return give_back(rc);
            ctx->vv_alt_zeroed = false;
        }
        if (ctx->vv_alt_zeroed) std::swap(ctx->vv_chg_bits, ctx->vv_chg_alt);
        else if ((rc = hip_rc(ctx, hipMemsetAsync(ctx->vv_chg_bits, 0, padded_words(ctx->cap) * 8, ctx->stream), "zeroing the ViewVisibility change mask")))
            return give_back(rc);
        ctx->vv_alt_zeroed = false;
        cz.zero[2] = ctx->vv_chg_alt;
        cz.zero_words[2] = (uint32_t)padded_words(ctx->cap);
    }
    ClusterFillJob fill_job{};
    const bool have_fill = ctx->cl_fill_pending && ctx->n;  // a deferred cluster fill of the previous frame rides along
    if (have_fill) {
        fill_job = ctx->cl_fill_job;
        ctx->cl_fill_pending = false;
    }
    CellsOrder o{ce.n_waves, (uint32_t*)ce.perm.p, (float4*)ce.sph_s.p, (float*)ce.g_s.p, (uint8_t*)ce.vv_s.p, (uint32_t*)ce.pass_s.p, (float4*)ce.sum_a.p,
                 (uint4*)ce.sum_b.p, (float4*)ce.sum_h.p, (uint32_t*)ce.state.p};
    const Columns c = columns_of(ctx);
    const CellsWork work{(uint4*)ce.work.p, (uint32_t*)ce.work_n.p + ce.work_parity, (uint32_t*)ce.work_n.p + (ce.work_parity ^ 1u), chained ? 0u : 1u};
    {
        hipError_t e;
        {
            ProfScope ps(ctx, K_VIS_BEGIN);  // (timer slot of the cell test: the frame's first launch)
            e = launch_cells_test(o, ctx->views_inline ? &ctx->view_set : nullptr, (const ViewParams*)ctx->views.p, n_views, cz, work, ctx->stream);
        }
        if (e == hipSuccess) {
            ProfScope ps(ctx, K_CULL);
            e = launch_frame_cells(c, o, ctx->views_inline ? &ctx->view_set : nullptr, (const ViewParams*)ctx->views.p, n_views, vo, cz, work, prev,
                                   have_fill ? &fill_job : nullptr, ctx->stream, ride_lists ? &lists_job : nullptr, lists_views);
        }
        if (e != hipSuccess) {
            lists_out();
            if (have_fill) launch_cluster_fill(fill_job.w, fill_job.n_clusters, fill_job.n_objects, ctx->stream);
            ce.valid = false;
            fail(ctx, MI_ERR_DEVICE, "frame kernel launch: %s", hipGetErrorString(e));
            return frame_abort(ctx, MI_ERR_DEVICE, prev, prev_has_job, prev_job);
        }
    }
    ce.work_parity ^= 1u;
    ctx->vv_alt_zeroed = true;
    ++ce.frames;
    ++ctx->sph_quiet;
    // ---- behind the frame: the lists and the next frame's starting masks, one launch over the finished masks ----
    {
        const uint32_t nx = (ctx->cur + 1u) % mi_ctx::N_FB;
        const size_t segs = (size_t)n_views * ctx->compact_classes;  // (one class segment per view: no class masks on this path)
        if ((rc = ensure(ctx, ctx->fb[nx].bitmask, bm_words * 8)) || (rc = ensure(ctx, ctx->fb[ctx->cur].seg_totals, segs * 4))) return behind_the_launch(rc);
        CellsFinishArgs fin{};
        fin.out = vo;
        fin.n_words = (uint32_t)words64(ctx->n);
        fin.n_blks = (fin.n_words + 63u) / 64u;
        if ((rc = ensure(ctx, ce.fin_scratch, ((size_t)n_views * fin.n_blks + (size_t)n_views * CELLS_FIN_GROUPS) * 4))) return behind_the_launch(rc);
        fin.blk_pre = (uint32_t*)ce.fin_scratch.p;
        fin.grp_tot = fin.blk_pre + (size_t)n_views * fin.n_blks;
        fin.copy_to = (uint64_t*)ctx->fb[nx].bitmask.p;
        fin.copy_words_per_view = vo.words_per_view;
        if (ctx->compact_fast) {  // rows are numbered in Entity-key order: the lists come out of this launch (nothing is deferred)
            ctx->seg_stride = ctx->cap;
            if ((rc = ensure(ctx, ctx->fb[ctx->cur].out_rows, segs * ctx->seg_stride * 4))) return behind_the_launch(rc);
            fin.out_rows = (uint32_t*)ctx->fb[ctx->cur].out_rows.p;
            fin.seg_stride = ctx->seg_stride;
            fin.seg_totals = (uint32_t*)ctx->fb[ctx->cur].seg_totals.p;
        }
        fin.max_groups = ce.mode == 3 ? 3u : 0u;
        // another frame follows at once: the lists ride in ITS launch (or cells_lists_join's); the block prefixes and the next frame's
        // starting masks are wanted either way
        const bool defer_lists = (flags & MI_CULL_MORE_FRAMES) && fin.out_rows != nullptr;
        if ((rc = hip_rc(ctx, launch_cells_finish(fin, n_views, !defer_lists, ctx->stream, prof_mark, ctx), "the launch behind the frame over the cull order")))
            return behind_the_launch(rc);
        if (defer_lists) {
            ce.lists_args = fin;
            ce.lists_views = n_views;
            ce.lists_pending = true;
        }
        ce.chain_ok = true;
        ce.chain_mask = ctx->fb[nx].bitmask.p;
        ce.chain_words = bm_words;
        ce.chain_views = n_views;
    }
    if (prev_has_job) exchange_push(ctx, prev_job);
    if (!ctx->compact_fast && (rc = run_compaction(ctx, vo, seg, flags))) return rc;  // (general compaction: reads the masks itself)
    ctx->culled = true;
    return exchange_end(ctx);
}

// The frame entry points share one body: PROPAGATE selects the fused flat kernel (mi_propagate_and_cull).
template <bool PROPAGATE>
static int32_t cull_frame(mi_ctx* ctx, const mi_view* views, uint32_t n_views, uint32_t flags) {
    // validate before anything changes: a failed call must leave the previous frame's deferred compaction in place
    if (!views || n_views == 0) return fail(ctx, MI_ERR_INVALID_ARG, "mi_cull: views NULL or n_views == 0");
    if (!PROPAGATE && (flags & MI_CULL_CHANGED_ROWS))
        return fail(ctx, MI_ERR_INVALID_ARG, "MI_CULL_CHANGED_ROWS belongs to mi_propagate_and_cull (mi_cull does not propagate)");
    if ((flags & MI_CULL_WITH_CLUSTERS) && (!ctx->cl_rows_bound || !ctx->cl_have_view))
        return fail(ctx, MI_ERR_NOT_READY, "MI_CULL_WITH_CLUSTERS needs mi_cluster_bind_objects_to_rows and mi_cluster_upload_view first");
    if (!PROPAGATE) {
        // the cull-only frame of a scene that has gone static: over the cell order (cells_frame below) once there is one
        bool build = false;
        if (cells_frame_eligible(ctx, views, n_views, flags, &build)) return cells_frame(ctx, views, n_views, flags, build);
    } else {
        ctx->cells.quiet = 0;
    }
    ctx->cells.valid = false;  // (this frame's kernel writes ViewVisibility, and refreshes world spheres, behind the order's back)
    VisibilityOut vo{};
    CompactFastArgs prev_args{};
    bool prev_has_job = false;
    mi_ctx::Exchange::Job prev_job{};
    const CompactFastArgs* prev = frame_begin(ctx, &prev_args, &prev_has_job, &prev_job) ? &prev_args : nullptr;
    int32_t rc = exchange_begin(ctx);
    if (rc) return frame_abort(ctx, rc, prev, prev_has_job, prev_job);
    if ((rc = prepare_views(ctx, views, n_views, &vo))) return frame_abort(ctx, rc, prev, prev_has_job, prev_job);
    SegOut seg;
    if ((rc = prepare_segments(ctx, n_views, &seg))) return frame_abort(ctx, rc, prev, prev_has_job, prev_job);
    if ((rc = row_summary_ensure(ctx))) return frame_abort(ctx, rc, prev, prev_has_job, prev_job);
    Columns c = columns_of(ctx);
    // MI_CULL_WITH_CLUSTERS: the assignment runs behind the frame kernel and reads the ViewVisibility column -- or, with
    // MI_CULL_CLUSTERS_CONCURRENT on a call that decides the frame's ViewVisibility alone, re-derives it and runs on the cluster
    // stream next to the frame kernel
    const bool with_clusters = (flags & MI_CULL_WITH_CLUSTERS) != 0;
    const bool whole_frame = (flags & MI_CULL_END_FRAME) && (PROPAGATE || (flags & MI_CULL_BEGIN_FRAME));
    const bool derivable = with_clusters && whole_frame && ctx->views_inline && !ctx->have_hierarchy && ctx->n;
    const bool clusters_concurrent = derivable && (flags & MI_CULL_CLUSTERS_CONCURRENT);
    // MI_CULL_CHANGED_ROWS: sync_simple_transforms' own filter -- before any change column was uploaded every row still counts
    // as changed (Added<GlobalTransform>), as in mi_propagate
    const uint8_t* const changed_col = (PROPAGATE && (flags & MI_CULL_CHANGED_ROWS) && ctx->have_changed) ? ctx->changed : nullptr;
    // a derived assignment takes a light's GlobalTransform as this frame's propagate leaves it: From(Transform) for the rows it writes
    ctx->cl_derive_changed = changed_col;
    ctx->cl_derive_resident = !PROPAGATE;
    if (clusters_concurrent && (rc = cluster_assign_launch(ctx, true, nullptr))) return frame_abort(ctx, rc, prev, prev_has_job, prev_job);
    bool clusters_ride = false;
    bool use_sph = false, sph_all_stale = false;
    {
        // a deferred cluster fill of the previous frame rides along (it must be enqueued before this frame's walk anyway) ...
        ClusterFillJob fill_job{};
        const bool have_fill = ctx->cl_fill_pending && ctx->n;
        if (have_fill) {
            fill_job = ctx->cl_fill_job;
            ctx->cl_fill_pending = false;
        }
        // ... and so does THIS frame's walk when the call decides the frame's ViewVisibility alone: the walk re-derives the
        // lights' visibility with the cull's rule, so it needs nothing the rest of the launch produces
        ClusterWalkJob walk_job{};
        // what cluster_ride_prepare changes, in case the launch below fails and the walk never runs
        const uint32_t cl_parity_before = ctx->cl_parity;
        const bool cl_assigned_before = ctx->cl_assigned;
        if (derivable && !clusters_concurrent && (rc = cluster_ride_prepare(ctx, &walk_job, &clusters_ride))) {
            if (have_fill) launch_cluster_fill(fill_job.w, fill_job.n_clusters, fill_job.n_objects, ctx->stream);
            return frame_abort(ctx, rc, prev, prev_has_job, prev_job);
        }
        // The world-sphere path (k_frame_sph) for frames that rewrite no or few GlobalTransforms: cull only, or the
        // changed-rows frame.  Camera views only.  A column that is not current is rebuilt by the SECOND such frame in a row: a
        // caller that rewrites every GlobalTransform every frame (mi_propagate(ALL_DIRTY) + mi_cull) never pays for it.
        if ((!PROPAGATE || changed_col) && ctx->sph_mode != 1 && n_views <= SPH_MAX_VIEWS && ctx->n) {
            bool camera_views_only = true;
            for (uint32_t v = 0; v < n_views; ++v) camera_views_only = camera_views_only && !(views[v].flags & MI_VIEW_FLAG_SHADOW);
            if (camera_views_only && (ctx->sph_state != mi_ctx::SPH_INVALID || ctx->sph_mode == 2 || ctx->sph_quiet >= 1)) {
                if (ctx->sph.bytes < (size_t)ctx->cap * 16) ctx->sph_state = mi_ctx::SPH_INVALID;  // (re)allocated below: nothing in it
                if ((rc = ensure(ctx, ctx->sph, (size_t)ctx->cap * 16))) {
                    if (have_fill) launch_cluster_fill(fill_job.w, fill_job.n_clusters, fill_job.n_objects, ctx->stream);
                    return frame_abort(ctx, rc, prev, prev_has_job, prev_job);
                }
                use_sph = true;
                sph_all_stale = ctx->sph_state == mi_ctx::SPH_INVALID;
            }
        }
        ProfScope ps(ctx, PROPAGATE ? K_FLAT_PROPAGATE_CULL : K_CULL);
        // the riding walk's plane table travels as a kernel argument (WalkPlanes, kernels.h)
        const WalkPlanesHost walk_planes = clusters_ride ? WalkPlanesHost{ctx->cl_planes_host.data(), (uint32_t)ctx->cl_planes_host.size()} : WalkPlanesHost{nullptr, 0};
        const bool stale_from_mask = use_sph && ctx->sph_state == mi_ctx::SPH_EXCEPT_CHANGED;
        const hipError_t e = use_sph ? launch_frame_sph(c, ctx->views_inline ? &ctx->view_set : nullptr, (const ViewParams*)ctx->views.p, n_views, vo, seg,
                                                        (flags & (MI_CULL_BEGIN_FRAME | MI_CULL_END_FRAME)) | (PROPAGATE ? CULL_BEGIN_FRAME : 0u), prev,
                                                        have_fill ? &fill_job : nullptr, clusters_ride ? &walk_job : nullptr, ctx->stream, changed_col,
                                                        (float*)ctx->sph.p, stale_from_mask && !ctx->g_chg_in_bytes ? ctx->g_chg_bits : nullptr,
                                                        stale_from_mask && ctx->g_chg_in_bytes ? ctx->g_changed_bytes : nullptr, sph_all_stale, walk_planes)
                             : PROPAGATE ? launch_flat_propagate_cull(c, ctx->views_inline ? &ctx->view_set : nullptr, (const ViewParams*)ctx->views.p,
                                                                    n_views, vo, seg, flags & MI_CULL_END_FRAME, prev, have_fill ? &fill_job : nullptr,
                                                                    clusters_ride ? &walk_job : nullptr, ctx->stream, changed_col, walk_planes)
                                       : launch_cull(c, ctx->views_inline ? &ctx->view_set : nullptr, (const ViewParams*)ctx->views.p, n_views, vo,
                                                     seg, flags & (MI_CULL_BEGIN_FRAME | MI_CULL_END_FRAME), prev, have_fill ? &fill_job : nullptr,
                                                     clusters_ride ? &walk_job : nullptr, ctx->stream, walk_planes);
        if (e != hipSuccess) {
            // The launch that was to carry the previous frame's fill and this frame's walk never happened: the fill goes out
            // on its own (as in the ride_prepare failure above), and the walk's bookkeeping is taken back -- no fill may later
            // expand, and no download may return, a set nothing walked into.
            if (have_fill) launch_cluster_fill(fill_job.w, fill_job.n_clusters, fill_job.n_objects, ctx->stream);
            if (clusters_ride) {
                ctx->cl_fill_pending = false;
                ctx->cl_assigned = cl_assigned_before;
                ctx->cl_parity = cl_parity_before;
            }
            fail(ctx, MI_ERR_DEVICE, "frame kernel launch: %s", hipGetErrorString(e));
            return frame_abort(ctx, MI_ERR_DEVICE, prev, prev_has_job, prev_job);
        }
    }
    // the world-sphere column after this frame
    if (use_sph) ctx->sph_state = mi_ctx::SPH_VALID;  // stale rows (and the rows this frame propagated) were refreshed
    else if (PROPAGATE && !changed_col) {             // every GlobalTransform rewritten
        ctx->sph_state = mi_ctx::SPH_INVALID;
        ctx->sph_quiet = 0;
        if (!ctx->have_hierarchy) ctx->frame_all_version = ctx->trs_version;  // (= From(Transform) of every row: what a fetch ahead holds)
    } else if (PROPAGATE)                              // k_frame<2>: the rewritten rows are this frame's change mask
        ctx->sph_state = ctx->sph_state == mi_ctx::SPH_VALID ? mi_ctx::SPH_EXCEPT_CHANGED : mi_ctx::SPH_INVALID;
    // a changed-rows frame of a flat table: its change mask is exactly the rows of the one indexed upload since the column was clean, if
    // nothing else raised a mark or wrote a Transform in between (either frame kernel)
    if (PROPAGATE && changed_col)
        ctx->gs_frame_ok = ctx->gs_k && !ctx->have_hierarchy && ctx->gs_frame_serial && ctx->gs_frame_serial == ctx->marks_serial && ctx->gs_trs_version == ctx->trs_version;
    if (!PROPAGATE || changed_col) ++ctx->sph_quiet;
    if (prev && ctx->n == 0) HIP_TRY(ctx, launch_compact_fast(*prev, ctx->stream));  // no frame kernel to ride in
    if (prev_has_job) exchange_push(ctx, prev_job);  // the launch that publishes the previous frame's signal is submitted
    if ((rc = run_compaction(ctx, vo, seg, flags))) return rc;
    if (with_clusters && !clusters_concurrent && !clusters_ride && (rc = cluster_assign_launch(ctx, false, nullptr, (flags & MI_CULL_MORE_FRAMES) != 0))) return rc;
    if (clusters_ride && !(flags & MI_CULL_MORE_FRAMES) && (rc = cluster_fill_join(ctx))) return rc;  // nobody promised a frame to carry the fill
    if (PROPAGATE) {
        if (ctx->have_changed && ctx->changed_maybe) {
            // a concurrent walk on the cluster stream reads the change column (and the GlobalTransforms of the rows it does not
            // mark): a memset (bulk marks, stamp wrap) waits for it
            if (clusters_concurrent && changed_col && ctx->cl_on_side && (ctx->changed_bulk || ctx->changed_gen >= 255u))
                HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_cl_done, 0));
            if ((rc = consume_changed(ctx))) return rc;
        }
        ctx->g_chg_maybe = true;
        ctx->g_chg_in_bytes = false;
        ctx->propagated_rows = ctx->n;
    }
    ctx->culled = true;
    return exchange_end(ctx);
}

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
    ctx->xch.debug = getenv("MI_XCH_DEBUG") != nullptr;  // environment knobs of the exchange: read once, here
    if (const char* mv = getenv("MI_MULTI_VIEW")) g_multi_view_mode = atoi(mv);  // (1: several views never take k_frame's pair pass; an A/B switch)
    ctx->xch.async_enqueue = getenv("MI_XCH_SYNC_ENQUEUE") == nullptr;  // (set: ncclAllGather is enqueued on the caller's thread again)
    *out_ctx = ctx;
    return MI_OK;
}

int32_t mi_ctx_destroy(mi_ctx* ctx) {
    if (!ctx) return MI_OK;
    hipSetDevice(ctx->device);
    compaction_join(ctx);
    hipStreamSynchronize(ctx->stream);
    if (ctx->cl_stream) {
        hipStreamSynchronize(ctx->cl_stream);
        hipEventDestroy(ctx->ev_cl_done);
        hipEventDestroy(ctx->ev_cl_inputs);
        hipStreamDestroy(ctx->cl_stream);
    }
    if (ctx->g_host) hipHostFree(ctx->g_host);
    if (ctx->iota_host) hipHostFree(ctx->iota_host);
    if (ctx->gs_host) hipHostFree(ctx->gs_host);
    if (ctx->piece_streams) {
        hipStreamSynchronize(ctx->up_stream);
        hipStreamSynchronize(ctx->dn_stream);
        for (uint32_t k = 0; k < mi_ctx::UP_PIECES; ++k) {
            hipEventDestroy(ctx->ev_up[k]);
            hipEventDestroy(ctx->ev_pre[k]);
        }
        hipStreamDestroy(ctx->up_stream);
        hipStreamDestroy(ctx->dn_stream);
    }
    prof_collect(ctx);
    void* cols[] = {ctx->t, ctx->r, ctx->s, ctx->g, ctx->c, ctx->h, ctx->flags, ctx->vv, ctx->changed, ctx->g_changed_bytes,
                    ctx->layers, ctx->layers_hi, ctx->class_mask, ctx->keys, ctx->g_chg_bits, ctx->vv_chg_bits, ctx->vv_chg_alt, ctx->tree_bytes,
                    ctx->range, ctx->visibility, ctx->inh_changed, ctx->bt_set, ctx->bt_bin, ctx->bt_input, ctx->bt_row_meta,
                    ctx->bt_kind, ctx->bt_cpu_bin, ctx->bt_bucket};
    for (void* p : cols)
        if (p) hipFree(p);
    DevBuf* bufs[] = {&ctx->sph, &ctx->row_sum, &ctx->anc, &ctx->order, &ctx->chains, &ctx->snap, &ctx->tree_trace, &ctx->inh_bits, &ctx->sparse_cnt, &ctx->sparse_rows, &ctx->sparse_total, &ctx->sparse_g, &ctx->g_pre, &ctx->parent_idx, &ctx->node_flags, &ctx->tiles, &ctx->wtiles, &ctx->strips, &ctx->strip_rounds, &ctx->level_offs_dev, &ctx->views,
                      &ctx->block_counts, &ctx->seg_bases, &ctx->out_keys, &ctx->cl_pos,
                      &ctx->cl_type, &ctx->cl_layers, &ctx->cl_layers_hi, &ctx->cl_dir, &ctx->cl_sincos, &ctx->cl_planes, &ctx->cl_spheres,
                      &ctx->bt_set_indexed, &ctx->bt_table_off, &ctx->bt_table, &ctx->bt_meta_off, &ctx->bt_meta, &ctx->bt_rows_a, &ctx->bt_rows_b,
                      &ctx->bt_hist, &ctx->bt_set_count, &ctx->bt_set_scan, &ctx->bt_counters, &ctx->bt_wi[0], &ctx->bt_wi[1], &ctx->bt_md[0],
                      &ctx->bt_md[1], &ctx->bt_bs[0], &ctx->bt_bs[1], &ctx->bt_records, &ctx->bt_totals, &ctx->bt_bucket_desc, &ctx->bt_meta_out,
                      &ctx->bt_inst[0], &ctx->bt_inst[1], &ctx->bt_plan, &ctx->bt_unb, &ctx->bt_items, &ctx->bt_batches, &ctx->bt_sorted_partials,
                      &ctx->cl_row_list, &ctx->cl_remap, &ctx->cl_bind_oc, &ctx->cl_bind_idx, &ctx->cl_block_counts, &ctx->cl_pair_cb, &ctx->cl_pair_mask, &ctx->cl_acc,
                      &ctx->cl_offsets, &ctx->cl_indices, &ctx->cl_scalars};
    for (DevBuf* b : bufs)
        if (b->p) hipFree(b->p);
    for (auto& sl : ctx->cl_parked)  // (the parked cluster views' buffers; the selected one's are in the list above)
        for (DevBuf* b : {&sl.planes, &sl.spheres, &sl.remap, &sl.bind_oc, &sl.bind_idx, &sl.block_counts, &sl.pair_cb, &sl.pair_mask, &sl.acc, &sl.offsets,
                          &sl.indices, &sl.scalars})
            if (b->p) hipFree(b->p);
    for (auto& f : ctx->fb)
        for (DevBuf* b : {&f.bitmask, &f.wave_cnt, &f.seg_mask, &f.out_rows, &f.seg_totals})
            if (b->p) hipFree(b->p);
    for (void* p : ctx->xch.owned)  // (mi_exchange_configure_owned's gathered buffers)
        if (p) hipFree(p);
    {
        auto& ce = ctx->cells;  // the static cull order
        for (DevBuf* b : {&ce.perm, &ce.sph_s, &ce.g_s, &ce.vv_s, &ce.sum_a, &ce.sum_b, &ce.sum_h, &ce.state, &ce.keys_a, &ce.keys_b, &ce.vals_a, &ce.vals_b,
                          &ce.sort_tmp, &ce.minmax, &ce.work, &ce.work_n, &ce.pass_s, &ce.fin_scratch})
            if (b->p) hipFree(b->p);
    }
    if (ctx->stage) hipHostFree(ctx->stage);
    for (auto& c : ctx->win_chunks) hipHostFree(c.p);
    if (ctx->timer_a) hipEventDestroy(ctx->timer_a);
    if (ctx->timer_b) hipEventDestroy(ctx->timer_b);
    exchange_stop(ctx);
    if (ctx->xch.debug && !ctx->xch.simple && ctx->xch.frame)
        fprintf(stderr, "[mi exchange] frames %llu: begin %.2f us (wait-issued %.2f us), end %.2f us, worker %.2f us per frame\n",
                (unsigned long long)ctx->xch.frame, ctx->xch.dbg_begin_ns / ctx->xch.frame / 1e3, ctx->xch.dbg_wait_ns / ctx->xch.frame / 1e3,
                ctx->xch.dbg_end_ns / ctx->xch.frame / 1e3, ctx->xch.dbg_worker_ns / ctx->xch.frame / 1e3);
    if (ctx->xch.comm_stream[0]) {
        for (hipStream_t cs : ctx->xch.comm_stream)
            if (cs) hipStreamSynchronize(cs);
        for (uint32_t i = 0; i < mi_ctx::Exchange::MAX_BUFS; ++i) {
            if (ctx->xch.ev_gathered[i]) hipEventDestroy(ctx->xch.ev_gathered[i]);
            if (ctx->xch.ev_kernels[i]) hipEventDestroy(ctx->xch.ev_kernels[i]);
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
    {
        int32_t rcj = compaction_join(ctx);
        if (rcj) return rcj;
        if ((rcj = cluster_fill_join(ctx))) return rcj;
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->cl_stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->cl_stream));
    ctx->cl_on_side = false;
    if (ctx->xch.on && !ctx->xch.simple) {
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
    ctx->cells.valid = false, ctx->cells.quiet = 0;  // (the static cull order mirrors ViewVisibility and the row count)
    trs_written(ctx);
    ctx->gs_frame_ok = false;  // (rows come and go: what was written ahead for the last frame's rows is not handed out after this)
    {
        int32_t rcj = compaction_join(ctx);  // buffers may move
        if (rcj) return rcj;
    }
    {
        int32_t rcj = cluster_join(ctx);  // columns may move under an assignment that reads them
        if (rcj) return rcj;
        ctx->cl_inputs_dirty = true;
    }
    ctx->bt_resolve = true;
    ctx->changed_maybe = true, ++ctx->marks_serial;
    if (n_rows != ctx->n) {
        ctx->sph_state = mi_ctx::SPH_INVALID;
        const uint32_t a = std::min(n_rows, ctx->n) & ~63u, b = std::max(n_rows, ctx->n);  // the wave the old end lies in and everything behind
        row_summary_touch(ctx, ROWSUM_PART_AABB | ROWSUM_PART_FLAGS, a, b - a);
    }
    if (n_rows > ctx->n) {  // new rows are Added<GlobalTransform>: marked (a plain 1), uncounted
        ctx->changed_rows_hint = UINT64_MAX;
        ctx->changed_bulk = true;
    }
    if (n_rows < ctx->n && ctx->cl_rows_listed) ctx->cl_rows_bound = false;  // a listed row may be gone: the caller binds again
    const uint32_t old_cap_rows = ctx->cap;
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
        ctx->changed_maybe = true, ++ctx->marks_serial;
        if ((rc = grow_column(ctx, ctx->g_changed_bytes, 1, old, new_cap, 0))) return rc;
        if ((rc = grow_column(ctx, ctx->layers, 1, old, new_cap, 0))) return rc;
        if (ctx->layers_hi && (rc = grow_column(ctx, ctx->layers_hi, 1, old, new_cap, 0))) return rc;
        if ((rc = grow_column(ctx, ctx->class_mask, 1, old, new_cap, 0))) return rc;
        if ((rc = grow_column(ctx, ctx->keys, 1, old, new_cap, 0))) return rc;
        if ((rc = grow_column(ctx, ctx->range, 2, old, new_cap, 0))) return rc;
        if ((rc = grow_column(ctx, ctx->visibility, 1, old, new_cap, 0))) return rc;
        if ((rc = grow_column(ctx, ctx->inh_changed, 1, old, new_cap, 0))) return rc;
        if ((rc = grow_column(ctx, ctx->bt_set, 1, old, new_cap, 0xFF))) return rc;  // MI_NO_BATCH_SET until uploaded
        if ((rc = grow_column(ctx, ctx->bt_bin, 1, old, new_cap, 0))) return rc;
        if ((rc = grow_column(ctx, ctx->bt_input, 1, old, new_cap, 0))) return rc;
        if ((rc = grow_column(ctx, ctx->bt_row_meta, 1, old, new_cap, 0xFF))) return rc;
        if ((rc = grow_column(ctx, ctx->bt_kind, 1, old, new_cap, 0))) return rc;  // MI_BATCH_ROW_MULTIDRAWABLE
        if ((rc = grow_column(ctx, ctx->bt_cpu_bin, 1, old, new_cap, 0))) return rc;
        if ((rc = grow_column(ctx, ctx->bt_bucket, 1, old, new_cap, 0xFF))) return rc;
        // default RenderLayers = layer 0 (mask 1) for rows never uploaded
        {
            std::vector<uint32_t> ones(new_cap - old, 1u);
            if ((rc = upload(ctx, ctx->layers + old, ones.data(), ones.size() * 4))) return rc;
        }
        uint64_t* nb = nullptr;
        const size_t wbytes = padded_words(new_cap) * 8 + 256;
        if (ctx->vv_chg_alt) {  // (sized like vv_chg_bits: allocated again by the next frame that wants it)
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            HIP_TRY(ctx, hipFree(ctx->vv_chg_alt));
            ctx->vv_chg_alt = nullptr;
            ctx->vv_alt_zeroed = false;
        }
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
        if (ctx->tree_bytes) {
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            HIP_TRY(ctx, hipFree(ctx->tree_bytes));
        }
        // two halves: the frame's marks and the ones the frame's mark launch zeroes for the next frame
        const size_t half_bytes = (((size_t)new_cap + 255) & ~(size_t)255) + 256;
        HIP_TRY(ctx, hipMalloc((void**)&ctx->tree_bytes, 2 * half_bytes));
        HIP_TRY(ctx, hipMemsetAsync(ctx->tree_bytes, 0, 2 * half_bytes, ctx->stream));
        ctx->tree_half_words = (uint32_t)(half_bytes / 4);
        ctx->tree_clean[0] = ctx->tree_clean[1] = true;
        ctx->tree_parity = 0;
        ctx->marks_live = ctx->marks_in_cur = ctx->marks_complete = ctx->marks_other_cleared = false;
        ctx->cap = new_cap;
    }
    if (n_rows > ctx->n && ctx->n < old_cap_rows) {
        // Rows [n, min(n_rows, old capacity)) come (back) to life inside the existing allocation: they may hold a previous
        // occupant's values (shrink, then regrow).  Give them what a freshly allocated row has -- in particular
        // changed = 1: they are Added<GlobalTransform> rows and the next propagate must compute them.
        const uint32_t lo = ctx->n, cnt = std::min(n_rows, old_cap_rows) - lo;
        struct { void* p; size_t elem; int fill; } cols[] = {
            {ctx->t, 12, 0}, {ctx->r, 16, 0}, {ctx->s, 12, 0}, {ctx->g, 48, 0}, {ctx->c, 12, 0}, {ctx->h, 12, 0},
            {ctx->flags, 1, MI_FLAG_INHERITED_VISIBLE}, {ctx->vv, 1, 0}, {ctx->changed, 1, 1}, {ctx->g_changed_bytes, 1, 0},
            {ctx->layers_hi, 4, 0}, {ctx->class_mask, 4, 0}, {ctx->keys, 8, 0}, {ctx->range, 8, 0}, {ctx->visibility, 1, 0}, {ctx->inh_changed, 1, 0},
            {ctx->bt_set, 4, 0xFF}, {ctx->bt_bin, 4, 0}, {ctx->bt_input, 4, 0}, {ctx->bt_row_meta, 4, 0xFF},
            {ctx->bt_kind, 1, 0}, {ctx->bt_cpu_bin, 4, 0}, {ctx->bt_bucket, 4, 0xFF}};
        for (auto& cdesc : cols)
            if (cdesc.p) HIP_TRY(ctx, hipMemsetAsync((char*)cdesc.p + (size_t)lo * cdesc.elem, cdesc.fill, (size_t)cnt * cdesc.elem, ctx->stream));
        HIP_TRY(ctx, hipMemsetD32Async((hipDeviceptr_t)(ctx->layers + lo), 1, cnt, ctx->stream));  // default RenderLayers = layer 0
        ctx->propagated_rows = std::min(ctx->propagated_rows, lo);
    }
    if (n_rows < ctx->propagated_rows) ctx->propagated_rows = n_rows;
    if (n_rows != ctx->n) {
        // a different row count invalidates the hierarchy and any cull result
        if (ctx->marks_in_cur) ctx->tree_clean[ctx->tree_parity] = false;
        ctx->marks_live = ctx->marks_in_cur = ctx->marks_complete = ctx->marks_other_cleared = false;
        ctx->have_hierarchy = false;
        ctx->n_levels = 1;
        ctx->culled = false;
        ctx->h_keys.resize(n_rows, 0);
        ctx->order_dirty = true;
        ctx->level_offsets = {0, n_rows};
        ctx->passes.clear();
        ctx->groups.clear();
        ctx->stream_levels.clear();
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
    trs_written(ctx);
    if (n && n <= SMALL_UPLOAD_ROWS) {  // dirty-row sized: one staging block, one scatter kernel
        void* st = nullptr;
        if ((rc = stage_alloc(ctx, (size_t)n * 40, &st))) return rc;
        float* f = (float*)st;
        memcpy(f, translation, (size_t)n * 12);
        memcpy(f + 3 * (size_t)n, rotation, (size_t)n * 16);
        memcpy(f + 7 * (size_t)n, scale, (size_t)n * 12);
        void* dev = nullptr;
        HIP_TRY(ctx, hipHostGetDevicePointer(&dev, st, 0));
        if ((rc = cluster_join(ctx))) return rc;
        ctx->cl_inputs_dirty = true;
        HIP_TRY(ctx, launch_upload_trs((const float*)dev, ctx->t, ctx->r, ctx->s, first_row, n, ctx->stream));
        return MI_OK;
    }
    if ((rc = upload(ctx, ctx->t + 3 * (size_t)first_row, translation, (size_t)n * 12))) return rc;
    if ((rc = upload(ctx, ctx->r + 4 * (size_t)first_row, rotation, (size_t)n * 16))) return rc;
    if ((rc = upload(ctx, ctx->s + 3 * (size_t)first_row, scale, (size_t)n * 12))) return rc;
    return MI_OK;
}

// rows / t / r / s lie in the pinned arena: one scatter kernel reads them over PCIe and raises the rows' change bytes
// window_order: +1 / -1 = the rows lie in an upload window (they stay put until the windows are recycled) and strictly ascend / descend
static int32_t scatter_indexed(mi_ctx* ctx, const uint32_t* rows, const float* t, const float* r, const float* s, uint32_t n,
                               int window_order = 0) {
    trs_written(ctx);
    // ---- GlobalTransforms ahead of the changed-rows frame (ctx.h): only when this upload's rows will be exactly the frame's ----
    const bool column_clean = !ctx->changed_maybe && (ctx->have_changed || ctx->propagated_rows >= ctx->n);
    if (ctx->gs_k && !ctx->gs_used && ctx->chunk_mode != 2) ctx->sparse_ahead_wanted = false;  // (the last one was written for nothing)
    ctx->gs_k = 0;
    ctx->gs_frame_ok = false;
    ctx->gs_used = false;
    float* g_ahead = nullptr;
    if (window_order != 0 && column_clean && ctx->chunk_mode != 1 && (ctx->sparse_ahead_wanted || ctx->chunk_mode == 2) && !ctx->have_hierarchy &&
        !ctx->xch.on) {
        if (ctx->gs_host_bytes < (size_t)n * 48) {
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // (a scatter launch may still be writing the old one)
            if (ctx->gs_host) HIP_TRY(ctx, hipHostFree(ctx->gs_host));
            ctx->gs_host = nullptr;
            ctx->gs_host_bytes = 0;
            const size_t want = std::max<size_t>((size_t)n * 48 * 3 / 2, (size_t)1 << 20);
            HIP_TRY(ctx, hipHostMalloc(&ctx->gs_host, want, hipHostMallocMapped));
            ctx->gs_host_bytes = want;
        }
        void* d = nullptr;
        HIP_TRY(ctx, hipHostGetDevicePointer(&d, ctx->gs_host, 0));
        g_ahead = (float*)d;
    }
    void *d_rows = nullptr, *d_t = nullptr, *d_r = nullptr, *d_s = nullptr;  // (a component the window does not carry: nullptr)
    HIP_TRY(ctx, hipHostGetDevicePointer(&d_rows, (void*)rows, 0));
    if (t) HIP_TRY(ctx, hipHostGetDevicePointer(&d_t, (void*)t, 0));
    if (r) HIP_TRY(ctx, hipHostGetDevicePointer(&d_r, (void*)r, 0));
    if (s) HIP_TRY(ctx, hipHostGetDevicePointer(&d_s, (void*)s, 0));
    if (!ctx->have_changed) {
        // First use of the change column: rows a propagate has already consumed count as unchanged from here on.  Rows
        // that have not been through one yet are still Added<GlobalTransform> (systems.rs:45-50) and keep their mark.
        if (ctx->propagated_rows)
            HIP_TRY(ctx, hipMemsetAsync(ctx->changed, 0, std::min(ctx->propagated_rows, ctx->n), ctx->stream));
        ctx->have_changed = true;
    }
    int32_t rc = cluster_join(ctx);
    if (rc) return rc;
    ctx->cl_inputs_dirty = true;
    // mark_dirty_trees for these rows in the same launch, when the frames run under the static-scene rule and the half the next
    // propagate will read holds nothing but such marks
    const bool mark_here = ctx->have_hierarchy && ctx->marks_live && ctx->tree_bytes && ctx->parent_idx.p &&
                           (ctx->marks_in_cur || ctx->tree_clean[ctx->tree_parity]);
    uint8_t* const cur = mark_here ? ctx->tree_bytes + (size_t)ctx->tree_parity * ctx->tree_half_words * 4 : nullptr;
    uint32_t* other = nullptr;
    if (mark_here && !ctx->marks_other_cleared && !ctx->tree_clean[ctx->tree_parity ^ 1u])
        other = (uint32_t*)(ctx->tree_bytes + (size_t)(ctx->tree_parity ^ 1u) * ctx->tree_half_words * 4);
    HIP_TRY(ctx, launch_upload_trs_indexed((const uint32_t*)d_rows, (const float*)d_t, (const float*)d_r, (const float*)d_s, n, ctx->t, ctx->r,
                                           ctx->s, ctx->changed, ctx->changed_gen, ctx->stream, mark_here ? (const uint32_t*)ctx->parent_idx.p : nullptr,
                                           cur, other, ctx->tree_half_words, ctx->anc_valid ? (const uint32_t*)ctx->anc.p : nullptr, g_ahead, window_order < 0));
    if (mark_here) {
        if (!ctx->marks_in_cur) ctx->marks_complete = !ctx->changed_maybe;  // complete so far iff nothing was marked changed before this upload
        ctx->marks_in_cur = true;
        if (other) {
            ctx->marks_other_cleared = true;
            ctx->tree_clean[ctx->tree_parity ^ 1u] = true;
        }
    } else {
        ctx->marks_complete = false;
    }
    ctx->changed_maybe = true, ++ctx->marks_serial;
    if (ctx->changed_rows_hint != UINT64_MAX) ctx->changed_rows_hint += n;
    if (g_ahead) {
        ctx->gs_k = n;
        ctx->gs_rows = rows;
        if (window_order < 0) {  // (the results list rows in ascending order: the GlobalTransforms were written back to front, the rows follow)
            ctx->gs_rows_rev.resize(n);
            for (uint32_t i = 0; i < n; ++i) ctx->gs_rows_rev[i] = rows[n - 1u - i];
            ctx->gs_rows = ctx->gs_rows_rev.data();
        }
        ctx->gs_marks_serial = ctx->marks_serial;
        ctx->gs_trs_version = ctx->trs_version;
    }
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
    return scatter_indexed(ctx, u, f, f + 3 * (size_t)n, f + 7 * (size_t)n, n);
}

// ---- upload windows: the caller fills pinned memory in place ------------------------------------------------------------------
int32_t mi_map_upload_window(mi_ctx* ctx, uint32_t capacity, uint32_t flags, mi_upload_window* out) {
    ENTER_RAW(ctx);  // (host memory only)
    if (!out) return fail(ctx, MI_ERR_INVALID_ARG, "mi_map_upload_window: NULL");
    memset(out, 0, sizeof *out);
    out->flags = flags;
    out->capacity = capacity;
    if (capacity == 0) return MI_OK;
    if (flags & ~(MI_UPLOAD_DENSE | MI_UPLOAD_TRANSLATION | MI_UPLOAD_ROTATION | MI_UPLOAD_SCALE))
        return fail(ctx, MI_ERR_INVALID_ARG, "mi_map_upload_window: unknown flags 0x%x", flags);
    const bool dense = (flags & MI_UPLOAD_DENSE) != 0;
    const uint32_t comp = flags & (MI_UPLOAD_TRANSLATION | MI_UPLOAD_ROTATION | MI_UPLOAD_SCALE);
    const bool has_t = !comp || (comp & MI_UPLOAD_TRANSLATION), has_r = !comp || (comp & MI_UPLOAD_ROTATION), has_s = !comp || (comp & MI_UPLOAD_SCALE);
    const size_t row_bytes = (dense ? 0u : 4u) + (has_t ? 12u : 0u) + (has_r ? 16u : 0u) + (has_s ? 12u : 0u);
    const size_t need = (((size_t)capacity * row_bytes + 128) + 255) & ~(size_t)255;
    auto& chunks = ctx->win_chunks;
    if (ctx->win_open == 0 && !chunks.empty() && (chunks.size() > 1 || chunks.back().used + need > chunks.back().bytes)) {
        // nothing is mapped: recycle.  What the device still reads of earlier windows (DMA, the scatter kernel) has to be done
        // first -- by now normally long since (the frame's results were waited for)
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->gs_k && ctx->gs_rows != ctx->gs_rows_rev.data()) {
            // the rows of the last indexed window may still be owed to a results call (written-ahead GlobalTransforms, ctx.h): they
            // move out of the memory that is about to be handed out again
            ctx->gs_rows_rev.assign(ctx->gs_rows, ctx->gs_rows + ctx->gs_k);
            ctx->gs_rows = ctx->gs_rows_rev.data();
        }
        while (chunks.size() > 1) {  // keep the biggest chunk
            auto small = chunks.begin();
            for (auto it = chunks.begin(); it != chunks.end(); ++it)
                if (it->bytes < small->bytes) small = it;
            HIP_TRY(ctx, hipHostFree(small->p));
            chunks.erase(small);
        }
        chunks.back().used = 0;
        ++ctx->win_gen;
    }
    if (chunks.empty() || chunks.back().used + need > chunks.back().bytes) {
        if (ctx->win_open == 0 && !chunks.empty()) {  // (recycled above and still too small)
            HIP_TRY(ctx, hipHostFree(chunks.back().p));
            chunks.pop_back();
        }
        mi_ctx::WinChunk c{nullptr, std::max<size_t>(need, (size_t)64 << 20), 0};
        HIP_TRY(ctx, hipHostMalloc(&c.p, c.bytes, hipHostMallocMapped));
        chunks.push_back(c);
    }
    char* st = (char*)chunks.back().p + chunks.back().used;
    chunks.back().used += need;
    float* f = (float*)st;
    if (!dense) {
        out->rows = (uint32_t*)st;
        f = (float*)(out->rows + (((size_t)capacity + 3) & ~(size_t)3));  // 16-byte aligned columns
    }
    // the components the window carries, one after the other, each on a 16-byte boundary (the scatter kernel's wide loads)
    auto take = [&](size_t floats) {
        float* p = f;
        f += (floats + 3) & ~(size_t)3;
        return p;
    };
    out->translation = has_t ? take(3 * (size_t)capacity) : nullptr;
    out->rotation = has_r ? take(4 * (size_t)capacity) : nullptr;
    out->scale = has_s ? take(3 * (size_t)capacity) : nullptr;
    out->token = ctx->win_gen;
    ++ctx->win_open;
    return MI_OK;
}

int32_t mi_commit_upload_window(mi_ctx* ctx, const mi_upload_window* w, uint32_t n, uint32_t first_row) {
    ENTER_RAW(ctx);  // (may continue a sequence of dense windows, below; every other way out of here ends it)
    if (!w) return fail(ctx, MI_ERR_INVALID_ARG, "mi_commit_upload_window: NULL");
    if (w->capacity == 0) return MI_OK;
    if (n > w->capacity) return fail(ctx, MI_ERR_INVALID_ARG, "mi_commit_upload_window: %u rows, the window holds %u", n, w->capacity);
    const char* base = w->rows ? (const char*)w->rows : w->translation ? (const char*)w->translation : w->rotation ? (const char*)w->rotation : (const char*)w->scale;
    bool inside = false;
    for (auto& c : ctx->win_chunks) inside = inside || (base >= (const char*)c.p && base < (const char*)c.p + c.bytes);
    if (w->token != ctx->win_gen || !inside || ctx->win_open == 0)
        return fail(ctx, MI_ERR_NOT_READY, "mi_commit_upload_window: not a window that is mapped at present");
    --ctx->win_open;  // (committing with n == 0 just gives the window back)
    if (n == 0) return MI_OK;
    if (w->flags & MI_UPLOAD_DENSE) {
        int32_t rc = check_rows(ctx, first_row, n, "mi_commit_upload_window");
        if (rc) return rc;
        if ((rc = cluster_join(ctx))) return rc;
        ctx->cl_inputs_dirty = true;
        // ---- a sequence of dense windows that carries the whole flat table (ctx.h, "dense uploads in pieces") ----
        const uint32_t cov = ctx->seq_cov;  // (ENTER_RAW above: a sequence in progress survives this call)
        const bool starts = first_row == 0, continues = cov != 0 && first_row == cov && ctx->seq_pieces < mi_ctx::UP_PIECES;
        if (ctx->chunk_mode != 1 && (starts || continues) && ctx->n >= (ctx->chunk_mode == 2 ? 1u : 262144u) && !ctx->have_hierarchy && !ctx->xch.on) {
            if ((rc = piece_streams(ctx))) return rc;
            if (starts) {
                // Whatever still reads the columns on the context's stream comes first -- waited for by the host, not by the upload
                // stream: with an event wait in front of them the runtime sent the upload stream's copies through the DMA engine the
                // download stream uses, and the two directions took turns (2.15 ms against 1.35 for the same pattern with this wait
                // on the host, profiles/r03_experiments.md 12).  The stream is idle here in a loop of frames: the results of the
                // frame before have been delivered.
                HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
                if (ctx->pre_version && !ctx->pre_used) ctx->ahead_wanted = false;  // (the last fetch ahead was for nothing)
                ctx->seq_pieces = 0;
                ctx->pre_version = 0;
                ctx->pre_used = false;
                ctx->seq_ahead = ctx->ahead_wanted || ctx->chunk_mode == 2;
                if (ctx->seq_ahead) {
                    if ((rc = ensure(ctx, ctx->g_pre, (size_t)ctx->cap * 48))) return rc;
                    if (ctx->g_host_bytes < (size_t)ctx->n * 48) {
                        HIP_TRY(ctx, hipStreamSynchronize(ctx->dn_stream));  // (a fetch nobody asked for may still be landing)
                        if (ctx->g_host) HIP_TRY(ctx, hipHostFree(ctx->g_host));
                        ctx->g_host = nullptr;
                        ctx->g_host_bytes = 0;
                        HIP_TRY(ctx, hipHostMalloc(&ctx->g_host, (size_t)ctx->cap * 48, hipHostMallocDefault));
                        ctx->g_host_bytes = (size_t)ctx->cap * 48;
                    }
                }
            }
            // a window that carries a large part of the table goes out in pieces of about an eighth of it
            const uint32_t room = mi_ctx::UP_PIECES - ctx->seq_pieces;
            const uint32_t parts = std::max(1u, std::min<uint32_t>(room, (uint32_t)(((uint64_t)n * mi_ctx::SPLIT + ctx->n / 2) / ctx->n)));
            for (uint32_t k = 0; k < parts; ++k) {
                const size_t lo = (size_t)n * k / parts, cnt = (size_t)n * (k + 1) / parts - lo, at = first_row + lo;
                const uint32_t ev = ctx->seq_pieces++;
                if (cnt) {  // (a component the window does not carry keeps its column)
                    if (w->translation) HIP_TRY(ctx, hipMemcpyAsync(ctx->t + 3 * at, w->translation + 3 * lo, cnt * 12, hipMemcpyHostToDevice, ctx->up_stream));
                    if (w->rotation) HIP_TRY(ctx, hipMemcpyAsync(ctx->r + 4 * at, w->rotation + 4 * lo, cnt * 16, hipMemcpyHostToDevice, ctx->up_stream));
                    if (w->scale) HIP_TRY(ctx, hipMemcpyAsync(ctx->s + 3 * at, w->scale + 3 * lo, cnt * 12, hipMemcpyHostToDevice, ctx->up_stream));
                }
                HIP_TRY(ctx, hipEventRecord(ctx->ev_up[ev], ctx->up_stream));
                // the context's stream follows piece by piece: everything launched on it from here on sees the upload so far
                HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_up[ev], 0));
                if (ctx->seq_ahead && cnt) {
                    HIP_TRY(ctx, launch_globals_ahead(ctx->t, ctx->r, ctx->s, (uint32_t)at, (uint32_t)(at + cnt), (float*)ctx->g_pre.p, ctx->stream));
                    HIP_TRY(ctx, hipEventRecord(ctx->ev_pre[ev], ctx->stream));
                    HIP_TRY(ctx, hipStreamWaitEvent(ctx->dn_stream, ctx->ev_pre[ev], 0));
                    HIP_TRY(ctx, hipMemcpyAsync((char*)ctx->g_host + at * 48, (const char*)ctx->g_pre.p + at * 48, cnt * 48, hipMemcpyDeviceToHost, ctx->dn_stream));
                }
            }
            ++ctx->n_piece_uploads;
            ++ctx->trs_version;
            ctx->seq_cov = first_row + n;
            if (ctx->seq_cov == ctx->n) {  // complete: what is landing in g_host is every row's From(Transform) as of this version
                ctx->seq_cov = 0;
                if (ctx->seq_ahead) {
                    ctx->pre_version = ctx->trs_version;
                    ctx->pre_n = ctx->n;
                }
            }
            return MI_OK;
        }
        trs_written(ctx);
        // pinned -> device: three DMA copies straight from the window (no staging copy)
        if (w->translation) HIP_TRY(ctx, hipMemcpyAsync(ctx->t + 3 * (size_t)first_row, w->translation, (size_t)n * 12, hipMemcpyHostToDevice, ctx->stream));
        if (w->rotation) HIP_TRY(ctx, hipMemcpyAsync(ctx->r + 4 * (size_t)first_row, w->rotation, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
        if (w->scale) HIP_TRY(ctx, hipMemcpyAsync(ctx->s + 3 * (size_t)first_row, w->scale, (size_t)n * 12, hipMemcpyHostToDevice, ctx->stream));
        return MI_OK;
    }
    // (strictly monotonic rows -- a Changed<Transform> query over tables whose rows follow the Entity key, one way or the other -- are
    // what the results of the frame will list: scatter_indexed)
    uint32_t top = 0, ascending = 1, descending = 1;
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t row = w->rows[i];
        ascending &= (uint32_t)(i == 0 || row > w->rows[i - 1]);
        descending &= (uint32_t)(i == 0 || row < w->rows[i - 1]);
        top = std::max(top, row);
    }
    if (top >= ctx->n) return fail(ctx, MI_ERR_INVALID_ARG, "mi_commit_upload_window: row %u >= %u live rows", top, ctx->n);
    return scatter_indexed(ctx, w->rows, w->translation, w->rotation, w->scale, n, ascending ? 1 : descending ? -1 : 0);
}

int32_t mi_upload_global_transforms(mi_ctx* ctx, uint32_t first_row, uint32_t n, const float* global12) {
    ENTER(ctx);
    if (!global12) return fail(ctx, MI_ERR_INVALID_ARG, "mi_upload_global_transforms: NULL");
    int32_t rc = check_rows(ctx, first_row, n, "mi_upload_global_transforms");
    if (rc) return rc;
    ctx->snap_valid = false;  // externally supplied values: the tree path re-snapshots its owner rows
    ctx->sph_state = mi_ctx::SPH_INVALID;
    // The column no longer holds what the last frame wrote: results that travelled ahead of it (every row's From(Transform) in g_host,
    // the indexed window's rows in gs_host) are not what a download of the column would return any more -- hand out neither.
    ctx->gs_frame_ok = false;
    ctx->frame_all_version = 0;
    return upload(ctx, ctx->g + 12 * (size_t)first_row, global12, (size_t)n * 48);
}

int32_t mi_upload_bounds(mi_ctx* ctx, uint32_t first_row, uint32_t n, const float* aabb_center, const float* aabb_half,
                         const uint8_t* flags, const uint32_t* layer_mask) {
    ENTER(ctx);
    if (!aabb_center || !aabb_half) return fail(ctx, MI_ERR_INVALID_ARG, "mi_upload_bounds: NULL column");
    int32_t rc = check_rows(ctx, first_row, n, "mi_upload_bounds");
    if (rc) return rc;
    ctx->sph_state = mi_ctx::SPH_INVALID;  // the spheres are functions of the bounds
    row_summary_touch(ctx, ROWSUM_PART_AABB | ROWSUM_PART_FLAGS, first_row, n);
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

int32_t mi_upload_render_layers_hi(mi_ctx* ctx, uint32_t first_row, uint32_t n, const uint32_t* layer_mask_hi) {
    ENTER(ctx);
    if (!layer_mask_hi) return fail(ctx, MI_ERR_INVALID_ARG, "mi_upload_render_layers_hi: NULL");
    int32_t rc = check_rows(ctx, first_row, n, "mi_upload_render_layers_hi");
    if (rc) return rc;
    if (!ctx->layers_hi) {  // the column appears with its first use: until then no kernel reads a fifth byte-quad per row
        bool any = false;
        for (uint32_t i = 0; i < n && !any; ++i) any = layer_mask_hi[i] != 0;
        if (!any) return MI_OK;  // nothing above layer 31: still no column
        if ((rc = grow_column(ctx, ctx->layers_hi, 1, 0, ctx->cap, 0))) return rc;
    }
    row_summary_touch(ctx, ROWSUM_PART_FLAGS, first_row, n);  // (rows above layer 31 are not summarised)
    return upload(ctx, ctx->layers_hi + first_row, layer_mask_hi, (size_t)n * 4);
}

int32_t mi_upload_view_visibility(mi_ctx* ctx, uint32_t first_row, uint32_t n, const uint8_t* vv) {
    ENTER(ctx);
    ctx->cells.valid = false, ctx->cells.quiet = 0;  // (the static cull order mirrors ViewVisibility and the row count)
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
    ctx->changed_maybe = true, ++ctx->marks_serial;
    ctx->changed_rows_hint = UINT64_MAX;  // how many of these bytes are set is not known here
    ctx->marks_complete = false;          // rows marked changed without a climb
    ctx->changed_bulk = true;             // plain 0 / 1 bytes, not stamps (a caller's 1 must not look like a past generation either:
                                          // only 0 and 1 are meaningful in an uploaded byte)
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
// =============================================================================================
// systems
// =============================================================================================
int32_t mi_propagate(mi_ctx* ctx, uint32_t flags) {
    ENTER(ctx);
    // (the change mask is this call's from here on -- whatever was written ahead of another frame does not describe it -- and the marks
    // of the upload that wrote ahead are consumed by this call, whichever way it goes)
    ctx->gs_frame_ok = false;
    ctx->gs_frame_serial = ctx->gs_marks_serial;
    ctx->gs_marks_serial = 0;
    ctx->frame_all_version = 0;
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
            // a change mask no cull frame has consumed yet is gone: the sphere column no longer knows which of its rows are stale
            if (ctx->sph_state == mi_ctx::SPH_EXCEPT_CHANGED) ctx->sph_state = mi_ctx::SPH_INVALID;
        }
        return MI_OK;
    }
    ctx->g_chg_maybe = true;
    // the world-sphere column: the rows this propagate rewrites are the ones its change mask will flag
    if (all_dirty) {
        ctx->sph_state = mi_ctx::SPH_INVALID;
        ctx->sph_quiet = 0;
        if (!ctx->have_hierarchy) ctx->frame_all_version = ctx->trs_version;  // (every row: From(Transform), as a fetch ahead holds it)
    } else {
        ctx->sph_state = ctx->sph_state == mi_ctx::SPH_VALID ? mi_ctx::SPH_EXCEPT_CHANGED : mi_ctx::SPH_INVALID;
    }
    Columns c = columns_of(ctx);
    const uint32_t n0 = ctx->have_hierarchy ? ctx->level_offsets[1] : ctx->n;
    const uint8_t* tree_bits = nullptr;
    // mark_dirty_trees returns early unless the static optimisation is enabled (systems.rs:131-133)
    if (ctx->have_hierarchy && static_opt && !all_dirty) {
        uint8_t* const cur = ctx->tree_bytes + (size_t)ctx->tree_parity * ctx->tree_half_words * 4;
        uint8_t* const other = ctx->tree_bytes + (size_t)(ctx->tree_parity ^ 1u) * ctx->tree_half_words * 4;
        if (ctx->marks_in_cur && ctx->marks_complete && ctx->tree_clean[ctx->tree_parity ^ 1u]) {
            // every change of this frame came through the indexed uploads, which climbed and marked as they went
            // (k_upload_trs_indexed) and zeroed the other half: nothing to launch
        } else {
            if (!ctx->tree_clean[ctx->tree_parity] && !ctx->marks_in_cur) {  // (not on the usual path: the previous frame's mark launch zeroed it)
                ProfScope ps(ctx, K_CLEAR);
                HIP_TRY(ctx, launch_clear_u32((uint32_t*)cur, ctx->tree_half_words, ctx->stream));
            }
            ProfScope ps(ctx, K_MARK_DIRTY);
            HIP_TRY(ctx, launch_mark_dirty(ctx->n, ctx->changed, ctx->changed_gen, (const uint32_t*)ctx->parent_idx.p, cur,
                                           ctx->tree_clean[ctx->tree_parity ^ 1u] ? nullptr : (uint32_t*)other, ctx->tree_half_words, ctx->stream,
                                           ctx->anc_valid ? (const uint32_t*)ctx->anc.p : nullptr));
        }
        ctx->tree_clean[ctx->tree_parity] = false;
        ctx->tree_clean[ctx->tree_parity ^ 1u] = true;
        ctx->tree_parity ^= 1u;
        tree_bits = cur;
        ctx->marks_live = true;
    } else {
        if (ctx->marks_in_cur) ctx->tree_clean[ctx->tree_parity] = false;  // marks of uploads nobody read: stale for a later frame
        ctx->marks_live = false;
    }
    ctx->marks_in_cur = ctx->marks_complete = ctx->marks_other_cleared = false;
    if (!ctx->have_hierarchy) {
        ProfScope ps(ctx, K_LEVEL0_PROPAGATE);
        HIP_TRY(ctx, launch_level0_propagate(c, n0, nullptr, ctx->changed, tree_bits, all_dirty, static_opt, ctx->stream));
        ctx->g_chg_in_bytes = false;
    } else if (ctx->narrow && !ctx->by_levels) {
        ProfScope sc(ctx, K_PROPAGATE_STREAM);
        HIP_TRY(ctx, launch_propagate_narrow(c, (const uint32_t*)ctx->parent_idx.p, (const uint32_t*)ctx->level_offs_dev.p, ctx->n_levels,
                                             (const uint8_t*)ctx->node_flags.p, ctx->changed, tree_bits, ctx->g_changed_bytes, all_dirty, static_opt, ctx->narrow_quad, ctx->stream));
        ctx->g_chg_in_bytes = true;
    } else if (ctx->strip_plan && !ctx->by_levels) {
        // strips: one launch of independent waves; the cones read the pre-frame snapshot of the rows above the deepest band, the strips
        // that own those rows write next frame's
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
        ProfScope sc(ctx, K_PROPAGATE_TILES);
        const bool pretest = ctx->tile_pretest_mode == 2 || (ctx->tile_pretest_mode == 0 && static_opt && !all_dirty && ctx->changed_rows_hint != UINT64_MAX &&
                                                            ctx->changed_rows_hint * 16 <= ctx->n_strips);
        HIP_TRY(ctx, launch_propagate_strips(c, (const uint32_t*)ctx->parent_idx.p, (const StripDesc*)ctx->strips.p, (const StripRound*)ctx->strip_rounds.p, ctx->n_strips,
                                             (const uint8_t*)ctx->node_flags.p, ctx->changed, tree_bits, ctx->g_changed_bytes, snap_r, snap_w, ctx->snap_rows, all_dirty,
                                             static_opt, pretest, ctx->stream, (unsigned long long*)ctx->tree_trace.p));
        ctx->g_chg_in_bytes = true;
    } else if (ctx->wave_forest && !ctx->by_levels) {
        // a forest of small trees: a wave per tile, one launch (ctx_hierarchy.cpp).  No chain tiles, so nothing reads or keeps the
        // chain snapshot: whatever it held is stale for a later plan.
        ProfScope sc(ctx, K_PROPAGATE_TILES);
        const bool pretest = ctx->tile_pretest_mode == 2 || (ctx->tile_pretest_mode == 0 && static_opt && !all_dirty && ctx->changed_rows_hint != UINT64_MAX &&
                                                            ctx->changed_rows_hint * 16 <= ctx->n_wtiles);
        HIP_TRY(ctx, launch_propagate_wave_tiles(c, (const uint32_t*)ctx->parent_idx.p, (const TileDesc*)ctx->wtiles.p, ctx->n_wtiles, (const uint8_t*)ctx->node_flags.p,
                                                 ctx->changed, tree_bits, ctx->g_changed_bytes, all_dirty, static_opt, ctx->wave_quad, pretest, ctx->stream));
        ctx->snap_valid = false;
        ctx->g_chg_in_bytes = true;
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
        // EVERY launch mirrors the rows it owns below snap_rows into next frame's snapshot -- not only the launch the
        // chain tiles ride in: a chain runs up through rows owned by earlier launches too, and under the static-scene
        // rule the chain tiles compare against (and fall back to) the snapshot's value.
        // few rows changed (every mark came through the indexed uploads, so their number is known): the light tiles test their flags
        // before asking for anything else -- a clean tile leaves after one small round trip; with many changed rows most tiles fail
        // the test and it would only cost them that round trip
        uint32_t n_tiles_all = 0;
        for (auto& gr : ctx->groups) n_tiles_all += gr.count;
        const bool tiles_pretest = ctx->tile_pretest_mode == 2 ||
                                   (ctx->tile_pretest_mode == 0 && static_opt && !all_dirty && ctx->changed_rows_hint != UINT64_MAX &&
                                    ctx->changed_rows_hint * 16 <= n_tiles_all);
        TreeCull tcull_rest;  // (the compaction riders of a fused hierarchy frame go with the first launch only)
        const TreeCull* tcull = ctx->tcull;
        if (ctx->by_levels) {
            // a hierarchy the tiles cannot hold (ctx_hierarchy.cpp): level 0 with the roots' rule, then each level behind its parents'
            for (uint32_t l = 0; l < ctx->n_levels; ++l) {
                ProfScope sc(ctx, K_PROPAGATE_STREAM);
                const uint32_t s = ctx->level_offsets[l], cnt = ctx->level_offsets[l + 1] - s;
                HIP_TRY(ctx, launch_propagate_level(c, (const uint32_t*)ctx->parent_idx.p, s, cnt, ctx->changed, tree_bits, ctx->g_changed_bytes,
                                                    all_dirty, static_opt, ctx->stream, l == 0 ? (const uint8_t*)ctx->node_flags.p : nullptr));
            }
        }
        for (auto& gr : ctx->groups) {
            if (ctx->by_levels) break;
            ProfScope sc(ctx, K_PROPAGATE_TILES);
            if (tcull && &gr != &ctx->groups.front() && tcull == ctx->tcull) {
                tcull_rest = *tcull;
                tcull_rest.n_compact = 0;
                tcull = &tcull_rest;
            }
            HIP_TRY(ctx, launch_propagate_tiles(c, (const uint32_t*)ctx->parent_idx.p, (const TileDesc*)ctx->tiles.p + gr.first,
                                                (const uint32_t*)ctx->chains.p + (size_t)gr.first * TILE_MAX_CHAIN, gr.count,
                                                (const uint8_t*)ctx->node_flags.p, ctx->changed, tree_bits, ctx->g_changed_bytes,
                                                gr.n_chain ? snap_r : nullptr, snap_w, ctx->snap_rows, all_dirty, static_opt,
                                                ctx->stream, ctx->tree_trace.p ? (unsigned long long*)ctx->tree_trace.p + (size_t)gr.first * 8 : nullptr, tiles_pretest, tcull, gr.deep));
        }
        for (auto& lv : ctx->stream_levels) {  // the wide deepest levels, each behind the level above it
            if (ctx->by_levels) break;
            ProfScope sc(ctx, K_PROPAGATE_STREAM);
            HIP_TRY(ctx, launch_propagate_level(c, (const uint32_t*)ctx->parent_idx.p, lv.first, lv.second, ctx->changed, tree_bits,
                                                ctx->g_changed_bytes, all_dirty, static_opt, ctx->stream));
        }
        ctx->g_chg_in_bytes = true;
    }
    {
        const int32_t rcc = consume_changed(ctx);  // change flags are consumed
        if (rcc) return rcc;
    }
    ctx->propagated_rows = ctx->n;
    return MI_OK;
}

int32_t mi_visibility_begin_frame(mi_ctx* ctx) {
    ENTER(ctx);
    ctx->cells.valid = false, ctx->cells.quiet = 0;  // (the static cull order mirrors ViewVisibility and the row count)
    ProfScope ps(ctx, K_VIS_BEGIN);
    HIP_TRY(ctx, launch_vis_begin(columns_of(ctx), ctx->stream));
    return MI_OK;
}

int32_t mi_visibility_end_frame(mi_ctx* ctx) {
    ENTER(ctx);
    ctx->cells.valid = false, ctx->cells.quiet = 0;  // (the static cull order mirrors ViewVisibility and the row count)
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
    return cull_frame<false>(ctx, views, n_views, flags);
}

// The shadow views of a frame (crates/bevy_light/src/lib.rs:342-757): the same frame kernel as the cameras' pass, each view's kind
// selected by its flags (visibility_rule.h), ORing into ViewVisibility; every view's packed mask and their union come back in one
// device wait (the masks of a frame's views lie one behind the other in the frame's buffer set).
int32_t mi_check_light_mesh_visibility(mi_ctx* ctx, const mi_view* shadow_views, uint32_t n_views, uint32_t flags, uint32_t* out_bitmasks,
                                       uint32_t* out_any) {
    ENTER(ctx);
    if (flags & ~MI_CULL_END_FRAME) return fail(ctx, MI_ERR_INVALID_ARG, "mi_check_light_mesh_visibility: flags 0x%x (MI_CULL_END_FRAME or 0)", flags);
    if (n_views && (!shadow_views || !out_bitmasks)) return fail(ctx, MI_ERR_INVALID_ARG, "mi_check_light_mesh_visibility: NULL views or out_bitmasks");
    for (uint32_t v = 0; v < n_views; ++v)
        if (!(shadow_views[v].flags & MI_VIEW_FLAG_SHADOW))
            return fail(ctx, MI_ERR_INVALID_ARG, "mi_check_light_mesh_visibility: view %u is not a shadow view (MI_VIEW_FLAG_SHADOW)", v);
    const size_t words32 = ((size_t)ctx->n + 31) / 32;
    if (out_any && words32) memset(out_any, 0, words32 * 4);
    if (n_views == 0 || ctx->n == 0) {
        if (n_views && words32) memset(out_bitmasks, 0, (size_t)n_views * words32 * 4);
        if (!(flags & MI_CULL_END_FRAME) || ctx->n == 0) return MI_OK;
        ctx->cells.valid = false, ctx->cells.quiet = 0;
        ProfScope ps(ctx, K_VIS_END);
        HIP_TRY(ctx, launch_vis_end(columns_of(ctx), ctx->stream));
        return MI_OK;
    }
    int32_t rc = cull_frame<false>(ctx, shadow_views, n_views, flags & MI_CULL_END_FRAME);
    if (rc) return rc;
    const uint64_t* base;
    uint64_t stride;
    if (ctx->ext_bitmask) base = (const uint64_t*)ctx->ext_bitmask + ctx->ext_word_offset, stride = ctx->ext_words_per_view;
    else base = (const uint64_t*)ctx->fb[ctx->cur].bitmask.p, stride = ctx->words_per_view;
    // one copy of [n_views][stride] words where that is no more than twice the payload (always, for the library's own buffers),
    // a copy per view otherwise (a caller-bound buffer with the other ranks' words in between); one wait either way
    const size_t row_bytes = words32 * 4;
    const bool one_copy = stride * 8 <= 2 * row_bytes + 4096;
    const size_t st_stride = one_copy ? (size_t)stride * 8 : ((row_bytes + 255) & ~(size_t)255);
    void* st = nullptr;
    if ((rc = stage_alloc(ctx, st_stride * n_views, &st))) return rc;
    if (one_copy) HIP_TRY(ctx, hipMemcpyAsync(st, base, st_stride * n_views, hipMemcpyDeviceToHost, ctx->stream));
    else
        for (uint32_t v = 0; v < n_views; ++v)
            HIP_TRY(ctx, hipMemcpyAsync((char*)st + v * st_stride, base + (size_t)v * stride, row_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const uint32_t tail = (ctx->n & 31u) ? (1u << (ctx->n & 31u)) - 1u : 0xFFFFFFFFu;
    for (uint32_t v = 0; v < n_views; ++v) {
        uint32_t* dst = out_bitmasks + (size_t)v * words32;
        memcpy(dst, (const char*)st + v * st_stride, row_bytes);
        dst[words32 - 1] &= tail;
        if (out_any)
            for (size_t w = 0; w < words32; ++w) out_any[w] |= dst[w];
    }
    return MI_OK;
}

int32_t mi_cull(mi_ctx* ctx, const float* frusta, const uint32_t* view_layer_masks, const uint8_t* view_flags,
                uint32_t n_views, uint32_t flags) {
    std::vector<mi_view> v;
    simple_views(v, frusta, view_layer_masks, view_flags, n_views);
    return mi_cull_views(ctx, v.empty() ? nullptr : v.data(), n_views, flags);
}

// The hierarchy frame in the tile launches themselves: every Transform counts as changed, so every tile runs, and each one also
// runs the visibility systems over its own rows with their GlobalTransforms still in registers / LDS (k_propagate_fans<true, true>)
// -- no second pass over the GlobalTransform column, one launch less.  A tile's rows are not aligned to the mask words, so the
// words are ORed and counted with atomics: masks, wave counts and the ViewVisibility change words are zeroed first (three
// memsets of a few hundred KB, enqueued in front of the tiles).  Applies when nothing else wants the frame kernels' own
// machinery: a plan of light tiles only, views in the kernel arguments, one class segment per view, library-owned masks, no
// visibility ranges, no cluster assignment in the same call.  Everything else takes the tile launch + cull launch.
// What the frame kernels carry for free rides here too: the previous frame's deferred compaction in the first workgroups of the
// (first) tile launch, and the zeroing -- every tile clears a slice of the NEXT frame's set (the frame sets rotate by three; the
// ViewVisibility change ticks alternate between two buffers), so that a run of such frames has no memset in it.
// MEASURED (profiles/r03_experiments.md, 1 M-node tree): the tile kernel grows from 24.2 to 33.0 us (68 registers: 7 instead of 8
// workgroups per CU; ~1 900 more vector instructions per tile in a kernel organised around latency), the cull launch (10.6 us) and
// one launch gap go: 34.7 against 37.1 us per frame at one view, 51.4 against 44.7 at four.  Round 4 (7 workgroups per CU, the rule's
// kernel arguments read late; profiles/r04_experiments.md 3): 32.7 us per frame at one view.  mi_debug_set_tree_cull: 1 = never, 2 =
// whenever it applies.
static bool tree_frame_fusable(mi_ctx* ctx, uint32_t n_views, uint32_t flags) {
    // (default: with one view, where it measured faster -- 34.7 against 37.1 us per frame of the 1 M-node tree; with four views the
    // rule's arithmetic inside the tiles costs more than the second pass over GlobalTransform: 51.4 against 44.7)
    return (ctx->tree_cull_mode == 2 || (ctx->tree_cull_mode == 0 && n_views == 1)) && ctx->n && !ctx->by_levels && !ctx->narrow && !ctx->wave_forest && !ctx->strip_plan && ctx->stream_levels.empty() && !ctx->groups.empty() && n_views <= MAX_INLINE_VIEWS &&
           !(flags & (MI_CULL_CHANGED_ROWS | MI_CULL_WITH_CLUSTERS)) && (flags & MI_CULL_END_FRAME) && !ctx->have_class_mask && !ctx->have_ranges &&
           !ctx->ext_bitmask && !ctx->xch.on;
}
static int32_t tree_frame_fused(mi_ctx* ctx, const mi_view* views, uint32_t n_views, uint32_t flags) {
    ctx->cells.valid = false;  // (the tiles write ViewVisibility)
    ctx->cells.quiet = 0;
    VisibilityOut vo{};
    CompactFastArgs prev_args{};
    bool prev_has_job = false;
    mi_ctx::Exchange::Job prev_job{};
    const CompactFastArgs* prev = frame_begin(ctx, &prev_args, &prev_has_job, &prev_job) ? &prev_args : nullptr;
    int32_t rc = exchange_begin(ctx);
    if (rc) return frame_abort(ctx, rc, prev, prev_has_job, prev_job);
    if ((rc = prepare_views(ctx, views, n_views, &vo))) return frame_abort(ctx, rc, prev, prev_has_job, prev_job);
    SegOut seg;
    if ((rc = prepare_segments(ctx, n_views, &seg))) return frame_abort(ctx, rc, prev, prev_has_job, prev_job);
    if ((rc = row_summary_ensure(ctx))) return frame_abort(ctx, rc, prev, prev_has_job, prev_job);
    if (prev_has_job) exchange_push(ctx, prev_job);
    if ((rc = cluster_fill_join(ctx))) return frame_abort(ctx, rc, prev, false, prev_job);  // (a deferred cluster fill has no launch to ride in here)
    TreeCull cu{};
    memcpy(cu.views.v, ctx->view_set.v, sizeof(ViewParams) * n_views);
    cu.n_views = n_views;
    cu.out = vo;
    cu.wave_cnt = seg.wave_cnt;
    cu.n_waves = seg.n_waves;
    mi_ctx::FbZero zn;
    rc = atomic_frame_sets(ctx, vo, seg, n_views, cu.zero, cu.zero_words, &zn);
    if (rc) return frame_abort(ctx, rc, prev, false, prev_job);
    // the previous frame's deferred compaction rides in the (first) tile launch
    if (prev && prev->n && prev->n_segments) {
        cu.prev = *prev;
        cu.prev_gx = compact_fast_gx(prev->n, false);
        cu.n_compact = cu.prev_gx * prev->n_segments;
    } else {
        cu.prev_gx = 1;
    }
    ctx->tcull = &cu;
    rc = mi_propagate(ctx, MI_PROPAGATE_ALL_DIRTY | ((flags & MI_CULL_STATIC_OPT) ? MI_PROPAGATE_STATIC_OPT : 0u));
    ctx->tcull = nullptr;
    if (rc) {  // the launch that was to carry the previous frame's compaction may not have happened: it goes out on its own (idempotent)
        if (prev) launch_compact_fast(*prev, ctx->stream);
        return rc;
    }
    atomic_frame_sets_done(ctx, cu.zero[0] != nullptr, zn);
    if ((rc = run_compaction(ctx, vo, seg, flags))) return rc;
    ctx->culled = true;
    return exchange_end(ctx);
}

int32_t mi_propagate_and_cull_views(mi_ctx* ctx, const mi_view* views, uint32_t n_views, uint32_t flags) {
    ENTER(ctx);
    ctx->gs_frame_ok = false;  // (as in mi_propagate)
    ctx->gs_frame_serial = ctx->gs_marks_serial;
    ctx->gs_marks_serial = 0;
    ctx->frame_all_version = 0;
    if (ctx->have_hierarchy && views && n_views && tree_frame_fusable(ctx, n_views, flags)) return tree_frame_fused(ctx, views, n_views, flags);
    if (ctx->have_hierarchy) {
        // With a hierarchy the frame is the tile launches of mi_propagate with the cull behind them: the same call for the
        // caller (and no host wait in between), G written once and read once.  (Flat rows have it in one kernel.)
        if (!views || n_views == 0) return fail(ctx, MI_ERR_INVALID_ARG, "mi_cull: views NULL or n_views == 0");
        if ((flags & MI_CULL_WITH_CLUSTERS) && (!ctx->cl_rows_bound || !ctx->cl_have_view))
            return fail(ctx, MI_ERR_NOT_READY, "MI_CULL_WITH_CLUSTERS needs mi_cluster_bind_objects_to_rows and mi_cluster_upload_view first");
        const uint32_t pf = ((flags & MI_CULL_CHANGED_ROWS) ? 0u : MI_PROPAGATE_ALL_DIRTY) | ((flags & MI_CULL_STATIC_OPT) ? MI_PROPAGATE_STATIC_OPT : 0u);
        const int32_t rc = mi_propagate(ctx, pf);
        if (rc) return rc;
        return cull_frame<false>(ctx, views, n_views, (flags & ~(MI_CULL_CHANGED_ROWS | MI_CULL_STATIC_OPT)) | MI_CULL_BEGIN_FRAME);
    }
    return cull_frame<true>(ctx, views, n_views, flags & ~MI_CULL_STATIC_OPT);
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
    row_summary_touch(ctx, ROWSUM_PART_FLAGS, 0, ctx->n);  // InheritedVisibility is bit 0 of the flags column
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
    for (auto& lv : ctx->stream_levels) {
        ProfScope sc(ctx, K_INHERIT);
        HIP_TRY(ctx, launch_inherit_level((const uint32_t*)ctx->parent_idx.p, lv.first, lv.second, ctx->visibility, ctx->flags, ctx->inh_changed,
                                          ctx->stream));
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
    {
        const void* before = ctx->sparse_total.p;
        if ((rc = ensure(ctx, ctx->sparse_total, compact_fast_totals_bytes(1, ctx->cap)))) return rc;
        if (ctx->sparse_total.p != before) HIP_TRY(ctx, hipMemsetAsync(ctx->sparse_total.p, 0, ctx->sparse_total.bytes, ctx->stream));
    }
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
    if (!++ctx->compact_tag) ++ctx->compact_tag;
    f.tag = ctx->compact_tag;
    HIP_TRY(ctx, launch_compact_fast(f, ctx->stream));
    return total ? download(ctx, total, ctx->sparse_total.p, 4) : MI_OK;
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

namespace {
// Room for `bytes` more in the staging arena WITHOUT a wrap in the middle of what follows: pointers handed out after this stay
// valid together (results delivered in place live there until the caller's next call).
int32_t stage_reserve(mi_ctx* ctx, size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (ctx->stage_used + bytes <= ctx->stage_bytes) return MI_OK;
    void* st = nullptr;
    int32_t rc = stage_alloc(ctx, bytes, &st);  // wraps (and grows) now
    if (rc) return rc;
    ctx->stage_used -= bytes;                   // ... and gives the room back
    return MI_OK;
}
// several device -> host copies, one wait: the pieces land in the pinned arena and are copied out after the wait -- or, in-place
// mode, stay there and the caller gets their addresses
struct BatchedDownload {
    struct Piece { void* dst; void* stage; size_t bytes; };
    std::vector<Piece> pieces;
    bool in_place = false;
    // dst: where the caller wants the bytes; in place: *out_ptr receives the address in the arena instead
    int32_t add(mi_ctx* ctx, void* dst, const void* src, size_t bytes, void** out_ptr = nullptr) {
        if (!bytes || (!dst && !out_ptr)) return MI_OK;
        void* st = nullptr;
        int32_t rc;
        // the arena wraps (and is reused from its start) when it is full: hand out what is parked in it first
        // (in-place results reserved their room up front -- stage_reserve -- so this never fires for them)
        if (ctx->stage_used + ((bytes + 255) & ~(size_t)255) > ctx->stage_bytes && (rc = finish(ctx))) return rc;
        rc = stage_alloc(ctx, bytes, &st);
        if (rc) return rc;
        HIP_TRY(ctx, hipMemcpyAsync(st, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        if (out_ptr) *out_ptr = st;
        pieces.push_back({out_ptr ? nullptr : dst, st, bytes});
        return MI_OK;
    }
    int32_t finish(mi_ctx* ctx) {
        if (pieces.empty()) return MI_OK;
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        for (auto& p : pieces)
            if (p.dst) memcpy(p.dst, p.stage, p.bytes);
        pieces.clear();
        return MI_OK;
    }
};
}  // namespace

// 0, 1, 2 ... in pinned memory: the changed-row list of a frame in which every row changed
static int32_t identity_rows(mi_ctx* ctx, uint32_t n) {
    if (ctx->iota_rows >= n) return MI_OK;
    if (ctx->iota_host) HIP_TRY(ctx, hipHostFree(ctx->iota_host));
    ctx->iota_host = nullptr;
    ctx->iota_rows = 0;
    HIP_TRY(ctx, hipHostMalloc((void**)&ctx->iota_host, (size_t)ctx->cap * 4, hipHostMallocDefault));
    ctx->iota_rows = ctx->cap;
    for (uint32_t i = 0; i < ctx->cap; ++i) ctx->iota_host[i] = i;
    return MI_OK;
}

int32_t mi_download_frame_results(mi_ctx* ctx, mi_frame_results* io) {
    ENTER(ctx);
    if (!io) return fail(ctx, MI_ERR_INVALID_ARG, "mi_download_frame_results: NULL");
    io->changed_count = 0;
    io->cluster_total = 0;
    io->farthest_z = 0.0f;
    const bool in_place = (io->flags & MI_RESULTS_IN_PLACE) != 0;
    const bool want_rows = (io->flags & MI_RESULTS_CHANGED_ROWS) != 0, want_g = (io->flags & MI_RESULTS_CHANGED_GLOBALS) != 0;
    const bool want_changed = (want_rows || want_g) && ctx->n;
    const bool want_indices = (io->flags & MI_RESULTS_CLUSTER_INDICES) != 0;
    const bool want_clusters = want_indices || (io->flags & MI_RESULTS_CLUSTERS) != 0;
    const uint32_t n_lists = io->n_lists;
    if (n_lists > MI_RESULTS_MAX_LISTS) return fail(ctx, MI_ERR_INVALID_ARG, "mi_download_frame_results: %u lists (at most %u)", n_lists, MI_RESULTS_MAX_LISTS);
    if (n_lists && !io->lists) return fail(ctx, MI_ERR_INVALID_ARG, "mi_download_frame_results: lists NULL");
    for (uint32_t l = 0; l < n_lists; ++l) io->lists[l].count = 0;
    if (!in_place) {
        if ((want_rows && !io->changed_rows) || (want_g && !io->changed_global12) || (want_clusters && (!io->cluster_offsets || !io->cluster_counts)) ||
            (want_indices && !io->cluster_indices))
            return fail(ctx, MI_ERR_INVALID_ARG, "mi_download_frame_results: a requested part has no buffer (or pass MI_RESULTS_IN_PLACE)");
        for (uint32_t l = 0; l < n_lists; ++l)
            if (!io->lists[l].rows && io->lists[l].capacity) return fail(ctx, MI_ERR_INVALID_ARG, "mi_download_frame_results: list %u has no buffer", l);
    } else {
        io->changed_rows = nullptr;
        io->changed_global12 = nullptr;
        io->cluster_offsets = io->cluster_counts = io->cluster_indices = nullptr;
        for (uint32_t l = 0; l < n_lists; ++l) io->lists[l].rows = nullptr;
    }
    int32_t rc;
    // ---- every GlobalTransform fetched ahead (mi_commit_upload_window): good for this call when the frame in between rewrote every
    // row from the very Transforms the fetch was computed from -- and every row then counts as changed (checked below) ----
    const bool g_ahead = want_g && ctx->n && ctx->pre_version && ctx->pre_version == ctx->trs_version && ctx->frame_all_version == ctx->trs_version &&
                         ctx->pre_n == ctx->n && io->changed_capacity >= ctx->n && !ctx->have_hierarchy;
    // ---- everything that has to run before the counts are final ----
    const uint32_t* list_total[PACK_MAX_LISTS] = {nullptr};
    const uint32_t* list_rows[PACK_MAX_LISTS] = {nullptr};
    const uint64_t* list_base[PACK_MAX_LISTS] = {nullptr};
    if (n_lists) {
        if (!ctx->culled) return fail(ctx, MI_ERR_NOT_READY, "mi_download_frame_results: visible list before mi_cull");
        if ((rc = compaction_join(ctx))) return rc;
        for (uint32_t l = 0; l < n_lists; ++l) {
            const mi_visible_list& ls = io->lists[l];
            if (ls.view >= ctx->compact_views) return fail(ctx, MI_ERR_INVALID_ARG, "mi_download_frame_results: list %u names view %u", l, ls.view);
            uint32_t slot = 0xFFFFFFFFu;
            for (uint32_t k = 0; k < ctx->compact_classes; ++k)
                if (ctx->class_bits[k] == ls.class_bit) slot = k;
            if (slot == 0xFFFFFFFFu) continue;  // no row carries this class: VisibleEntities::get() returns &[]
            const uint32_t seg = ls.view * ctx->compact_classes + slot;
            list_total[l] = (const uint32_t*)ctx->fb[ctx->cur].seg_totals.p + seg;
            if (ctx->compact_fast) {  // rows in key order: one strided region per segment
                list_rows[l] = (const uint32_t*)ctx->fb[ctx->cur].out_rows.p + (uint64_t)seg * ctx->seg_stride;
            } else {                  // sorted by key through the permutation: packed back to back, the bases are on the device
                list_rows[l] = (const uint32_t*)ctx->fb[ctx->cur].out_rows.p;
                list_base[l] = (const uint64_t*)ctx->seg_bases.p + seg;
            }
        }
    }
    if (want_clusters) {
        if (!ctx->cl_assigned) return fail(ctx, MI_ERR_NOT_READY, "mi_download_frame_results: clusters before an assignment");
        if ((rc = cluster_join(ctx))) return rc;
    }
    // the changed rows are those of one indexed upload window and their GlobalTransforms were written ahead (ctx.h): nothing to compact,
    // gather or fetch -- unless the window below turns out too small for the rest (then the usual way, further down)
    bool sparse_ahead = want_changed && ctx->gs_frame_ok && ctx->gs_k && ctx->gs_k <= io->changed_capacity && !ctx->have_hierarchy;
    // ... and so is the list of an all-rows frame of a flat table whose GlobalTransforms were fetched ahead: every row, by construction
    // (sync_simple_transforms writes -- and ticks -- every row it visits, systems.rs:45-50)
    bool all_ahead = want_changed && g_ahead && !sparse_ahead;
    if (want_changed && !sparse_ahead && !all_ahead && (rc = changed_rows_on_device(ctx, nullptr))) return rc;
    uint32_t changed = 0;
    uint64_t cl_total = 0;
    const uint32_t C = ctx->cl_view.n_clusters;
    const size_t off_totals = (6 * (size_t)C + 3) & ~(size_t)3, off_misc = off_totals + (((size_t)C + 3) & ~(size_t)3);
    const uint32_t* acc = want_clusters ? (const uint32_t*)ctx->cl_acc.p + ctx->cl_parity * (off_misc + 4) : nullptr;
    // worst case of every section: what the window (or, on the fallback path, the arena) has to hold
    uint64_t need = 0;
    if (want_changed && want_rows && !sparse_ahead && !all_ahead) need += pack_align((uint64_t)io->changed_capacity * 4u);
    if (want_changed && want_g && !sparse_ahead && !all_ahead) need += pack_align((uint64_t)io->changed_capacity * 48u);
    for (uint32_t l = 0; l < n_lists; ++l)
        if (list_total[l]) need += pack_align((uint64_t)io->lists[l].capacity * 4u);
    if (want_clusters) {
        need += pack_align(((uint64_t)C + 1u) * 4u) + pack_align((uint64_t)C * 24u);
        if (want_indices) need += pack_align(io->cluster_capacity * 4u);
    }
    // (twice: should the packed launch report an overflow, the copies of the fallback path land behind its window)
    if (in_place && (rc = stage_reserve(ctx, 2 * (PACK_HEADER_BYTES + std::min<uint64_t>(need, PACK_WINDOW_BYTES_IN_PLACE)) + 65536))) return rc;
    // ---- one launch, one wait: counts and lists packed by the device into a window of the pinned arena ----
    {
        PackResultsJob j{};
        j.payload_bytes = std::min<uint64_t>(need, in_place ? PACK_WINDOW_BYTES_IN_PLACE : PACK_WINDOW_BYTES);
        void* st = nullptr;
        if ((rc = stage_alloc(ctx, PACK_HEADER_BYTES + j.payload_bytes, &st))) return rc;
        j.header = (uint32_t*)st;
        j.payload = (uint8_t*)st + PACK_HEADER_BYTES;
        if (want_changed && !sparse_ahead && !all_ahead) {
            j.changed_total = (const uint32_t*)ctx->sparse_total.p;
            j.changed_rows = (const uint32_t*)ctx->sparse_rows.p;
            j.g = want_g ? ctx->g : nullptr;
            j.want_changed_rows = want_rows ? 1u : 0u;
            j.changed_capacity = io->changed_capacity;
        }
        j.big_rows = PACK_BIG_ROWS;
        j.n_lists = n_lists;
        for (uint32_t l = 0; l < n_lists; ++l) {
            j.list_total[l] = list_total[l];
            j.list_rows[l] = list_rows[l];
            j.list_base[l] = list_base[l];
            j.list_capacity[l] = io->lists[l].capacity;
        }
        if (want_clusters) {
            j.cluster_total = (const uint64_t*)ctx->cl_scalars.p;
            j.cluster_offsets = (const uint32_t*)ctx->cl_offsets.p;
            j.cluster_counts = acc;
            j.cluster_indices = want_indices ? (const uint32_t*)ctx->cl_indices.p : nullptr;
            j.farthest_z = (const float*)(acc + off_misc);
            j.n_clusters = C;
            j.cluster_capacity = io->cluster_capacity;
            j.cluster_indices_alloc = ctx->cl_indices.bytes / 4;
        }
        HIP_TRY(ctx, launch_pack_results(j, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        const uint32_t* h = j.header;
        if (h[5]) {
            changed = sparse_ahead ? ctx->gs_k : all_ahead ? ctx->n : h[0];
            cl_total = (uint64_t)h[2] | ((uint64_t)h[3] << 32);
            io->changed_count = changed;
            io->cluster_total = cl_total;
            if (want_clusters) memcpy(&io->farthest_z, &h[4], 4);
            int32_t cap_rc = MI_OK;
            uint8_t* src = j.payload;
            // a section: copied out, or its address handed to the caller
            auto deliver = [&](auto*& dst, size_t bytes) {
                typedef typename std::remove_reference<decltype(dst)>::type P;
                if (in_place) dst = reinterpret_cast<P>(src);
                else if (bytes) memcpy(dst, src, bytes);
                src += pack_align(bytes);
            };
            const bool fits_changed = want_changed && changed <= io->changed_capacity;
            const bool by_dma = fits_changed && !sparse_ahead && !all_ahead && h[6] != 0;  // many rows: left out of the window, fetched by the copy engine below
            if (all_ahead) {  // every row: 0 .. n-1 and what has been arriving since the upload
                if (want_rows) {
                    if ((rc = identity_rows(ctx, changed))) return rc;
                    if (in_place) io->changed_rows = ctx->iota_host;
                    else memcpy(io->changed_rows, ctx->iota_host, (size_t)changed * 4);
                }
                HIP_TRY(ctx, hipStreamSynchronize(ctx->dn_stream));
                if (in_place) io->changed_global12 = (float*)ctx->g_host;
                else memcpy(io->changed_global12, ctx->g_host, (size_t)changed * 48);
                ++ctx->n_ahead_downloads;
                ctx->pre_used = true;
            }
            if (sparse_ahead) {  // (the scatter launch that wrote them is long done: the wait above was for a launch far behind it)
                if (want_rows) {
                    if (in_place) io->changed_rows = const_cast<uint32_t*>(ctx->gs_rows);
                    else memcpy(io->changed_rows, ctx->gs_rows, (size_t)changed * 4);
                }
                if (want_g) {
                    if (in_place) io->changed_global12 = (float*)ctx->gs_host;
                    else memcpy(io->changed_global12, ctx->gs_host, (size_t)changed * 48);
                }
                ctx->gs_used = true;
                ++ctx->n_sparse_ahead_downloads;
            } else if (want_changed && want_g && changed && changed < ctx->n && ctx->frame_all_version != ctx->trs_version)
                ctx->sparse_ahead_wanted = true;  // (changed GlobalTransforms fetched the usual way: the next indexed window writes them ahead)
            if (want_changed && !fits_changed) cap_rc = fail(ctx, MI_ERR_CAPACITY, "%u GlobalTransforms changed, capacity %u", changed, io->changed_capacity);
            if (fits_changed && !by_dma && !sparse_ahead && !all_ahead && want_rows) deliver(io->changed_rows, (size_t)changed * 4);
            if (fits_changed && !by_dma && !sparse_ahead && !all_ahead && want_g) deliver(io->changed_global12, (size_t)changed * 48);
            for (uint32_t l = 0; l < n_lists; ++l) {
                mi_visible_list& ls = io->lists[l];
                ls.count = h[8u + l];
                if (!list_total[l]) continue;
                if (ls.count > ls.capacity) cap_rc = fail(ctx, MI_ERR_CAPACITY, "visible list %u has %u entries, capacity %u", l, ls.count, ls.capacity);
                else deliver(ls.rows, (size_t)ls.count * 4);
            }
            if (want_clusters) {
                deliver(io->cluster_offsets, ((size_t)C + 1) * 4);
                deliver(io->cluster_counts, (size_t)C * 24);
                if (want_indices) {
                    if (cl_total > io->cluster_capacity) cap_rc = fail(ctx, MI_ERR_CAPACITY, "cluster index list has %llu entries, capacity %llu", (unsigned long long)cl_total, (unsigned long long)io->cluster_capacity);
                    else deliver(io->cluster_indices, (size_t)cl_total * 4);
                }
            }
            if (by_dma) {  // a second wait, negligible next to megabytes over PCIe
                BatchedDownload b;
                b.in_place = in_place;
                if (want_rows && changed == ctx->n) {
                    // every row changed: the list (ascending, as compacted from the mask) is 0 .. n-1 -- kept on the host, nothing to fetch
                    if ((rc = identity_rows(ctx, changed))) return rc;
                    if (in_place) io->changed_rows = ctx->iota_host;
                    else memcpy(io->changed_rows, ctx->iota_host, (size_t)changed * 4);
                } else if (want_rows && (rc = b.add(ctx, io->changed_rows, ctx->sparse_rows.p, (size_t)changed * 4, in_place ? (void**)&io->changed_rows : nullptr))) return rc;
                if (want_g) {
                    // (every GlobalTransform of an all-rows frame fetched the usual way: the next sequence of dense windows fetches ahead)
                    if (changed == ctx->n && ctx->frame_all_version == ctx->trs_version) ctx->ahead_wanted = true;
                    const bool all_rows = changed == ctx->n;  // every row changed: the list is 0 .. n-1 and the column itself is the answer
                    if (!all_rows) {
                        if ((rc = ensure(ctx, ctx->sparse_g, (size_t)changed * 48))) return rc;
                        HIP_TRY(ctx, launch_gather_global((const uint32_t*)ctx->sparse_rows.p, (const uint32_t*)ctx->sparse_total.p, changed, ctx->g,
                                                          (float*)ctx->sparse_g.p, ctx->stream));
                    }
                    if ((rc = b.add(ctx, io->changed_global12, all_rows ? (const void*)ctx->g : ctx->sparse_g.p, (size_t)changed * 48,
                                    in_place ? (void**)&io->changed_global12 : nullptr)))
                        return rc;
                }
                if ((rc = b.finish(ctx))) return rc;
            }
            return cap_rc;
        }
        changed = 0;
        cl_total = 0;
        if (sparse_ahead || all_ahead) {  // (the fallback fetches everything from the device)
            sparse_ahead = all_ahead = false;
            if ((rc = changed_rows_on_device(ctx, nullptr))) return rc;
        }
    }
    // ---- the packed window was too small (or the cluster list outgrew its device buffer): wait 1, the counts and every fixed-size array ----
    BatchedDownload b;
    b.in_place = in_place;
    uint32_t list_count[PACK_MAX_LISTS] = {0};
    if (want_changed && (rc = b.add(ctx, &changed, ctx->sparse_total.p, 4))) return rc;
    uint64_t list_base_host[PACK_MAX_LISTS] = {0};
    for (uint32_t l = 0; l < n_lists; ++l) {
        if (list_total[l] && (rc = b.add(ctx, &list_count[l], list_total[l], 4))) return rc;
        if (list_base[l] && (rc = b.add(ctx, &list_base_host[l], list_base[l], 8))) return rc;
    }
    auto add_cluster_arrays = [&]() -> int32_t {
        int32_t r;
        if ((r = b.add(ctx, &io->farthest_z, acc + off_misc, 4))) return r;
        if ((r = b.add(ctx, io->cluster_offsets, ctx->cl_offsets.p, ((size_t)C + 1) * 4, in_place ? (void**)&io->cluster_offsets : nullptr))) return r;
        return b.add(ctx, io->cluster_counts, acc, (size_t)C * 6 * 4, in_place ? (void**)&io->cluster_counts : nullptr);
    };
    if (want_clusters) {
        if ((rc = b.add(ctx, &cl_total, ctx->cl_scalars.p, 8))) return rc;
        if ((rc = add_cluster_arrays())) return rc;
    }
    if ((rc = b.finish(ctx))) return rc;
    if (want_clusters && cl_total > ctx->cl_indices.bytes / 4) {  // fire-and-forget assign overflowed: redo with a big enough buffer
        uint64_t t2 = 0;
        if ((rc = mi_cluster_assign_resident(ctx, &t2))) return rc;
        if ((rc = cluster_join(ctx))) return rc;
        cl_total = t2;
        acc = (const uint32_t*)ctx->cl_acc.p + ctx->cl_parity * (off_misc + 4);
        if ((rc = add_cluster_arrays())) return rc;
        if ((rc = b.finish(ctx))) return rc;
    }
    io->changed_count = changed;
    io->cluster_total = cl_total;
    // ---- wait 2: the lists ----
    int32_t cap_rc = MI_OK;
    if (want_changed && changed) {
        if (changed > io->changed_capacity) cap_rc = fail(ctx, MI_ERR_CAPACITY, "%u GlobalTransforms changed, capacity %u", changed, io->changed_capacity);
        else {
            if (want_rows && (rc = b.add(ctx, io->changed_rows, ctx->sparse_rows.p, (size_t)changed * 4, in_place ? (void**)&io->changed_rows : nullptr))) return rc;
            if (want_g) {
                if ((rc = ensure(ctx, ctx->sparse_g, (size_t)changed * 48))) return rc;
                HIP_TRY(ctx, launch_gather_global((const uint32_t*)ctx->sparse_rows.p, (const uint32_t*)ctx->sparse_total.p, changed, ctx->g,
                                                  (float*)ctx->sparse_g.p, ctx->stream));
                if ((rc = b.add(ctx, io->changed_global12, ctx->sparse_g.p, (size_t)changed * 48, in_place ? (void**)&io->changed_global12 : nullptr))) return rc;
            }
        }
    }
    for (uint32_t l = 0; l < n_lists; ++l) {
        mi_visible_list& ls = io->lists[l];
        ls.count = list_count[l];
        if (!list_total[l] || !ls.count) continue;
        if (ls.count > ls.capacity) cap_rc = fail(ctx, MI_ERR_CAPACITY, "visible list %u has %u entries, capacity %u", l, ls.count, ls.capacity);
        else if ((rc = b.add(ctx, ls.rows, list_rows[l] + list_base_host[l], (size_t)ls.count * 4, in_place ? (void**)&ls.rows : nullptr))) return rc;
    }
    if (want_indices && cl_total) {
        if (cl_total > io->cluster_capacity) cap_rc = fail(ctx, MI_ERR_CAPACITY, "cluster index list has %llu entries, capacity %llu", (unsigned long long)cl_total, (unsigned long long)io->cluster_capacity);
        else if ((rc = b.add(ctx, io->cluster_indices, ctx->cl_indices.p, (size_t)cl_total * 4, in_place ? (void**)&io->cluster_indices : nullptr))) return rc;
    }
    if ((rc = b.finish(ctx))) return rc;
    return cap_rc;
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
    else base = (const uint64_t*)ctx->fb[ctx->cur].bitmask.p + view * ctx->words_per_view;
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
    {
        int32_t rcj = compaction_join(ctx);
        if (rcj) return rcj;
    }
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
    if ((rc = download(ctx, &total, (const uint32_t*)ctx->fb[ctx->cur].seg_totals.p + seg, 4))) return rc;
    *out_count = total;
    if (total > capacity) return fail(ctx, MI_ERR_CAPACITY, "visible list has %u entries, capacity %u", total, capacity);
    if (ctx->compact_fast) {
        // rows are in key order already; keys are looked up in the host copy of the key column
        base = (uint64_t)seg * ctx->seg_stride;
        std::vector<uint32_t> tmp;
        uint32_t* rows = out_rows;
        if (!rows && out_keys) { tmp.resize(total); rows = tmp.data(); }
        if (rows && (rc = download(ctx, rows, (const uint32_t*)ctx->fb[ctx->cur].out_rows.p + base, (size_t)total * 4))) return rc;
        if (out_keys)
            for (uint32_t i = 0; i < total; ++i) out_keys[i] = ctx->have_keys ? ctx->h_keys[rows[i]] : (uint64_t)rows[i];
        return MI_OK;
    }
    if ((rc = download(ctx, &base, (const uint64_t*)ctx->seg_bases.p + seg, 8))) return rc;
    if (out_keys && (rc = download(ctx, out_keys, (const uint64_t*)ctx->out_keys.p + base, (size_t)total * 8))) return rc;
    if (out_rows && (rc = download(ctx, out_rows, (const uint32_t*)ctx->fb[ctx->cur].out_rows.p + base, (size_t)total * 4))) return rc;
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
        else { p = ctx->fb[ctx->cur].bitmask.p; bytes = ctx->words_per_view * 8 * ctx->n_views; }
        break;
    case MI_BUF_VIEW_VISIBILITY: p = ctx->vv; bytes = ctx->n; break;
    case MI_BUF_VISIBLE_ROWS: {
        int32_t rcj = compaction_join(ctx);  // a deferred compaction is enqueued now: complete in stream order
        if (rcj) return rcj;
        p = ctx->fb[ctx->cur].out_rows.p;
        bytes = ctx->fb[ctx->cur].out_rows.bytes;
        break;
    }
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
                                               "k_compact_fast", "k_mark_dirty", "k_propagate_fans", "k_cluster_walk", "k_cluster_fill",
                                               "k_clear_u32", "k_inherit", "k_batch_hist", "k_batch_plan", "k_batch_emit",
                                               "k_batch_scan", "k_batch_scatter", "k_batch_bounds", "k_batch_sorted", "k_propagate_stream"};
    return k < K_NUM_KERNELS ? names[k] : nullptr;
}

// test / bench hook: the flags-first test of the light tiles under the static-scene rule: 0 = when few rows changed (default),
// 1 = never, 2 = always
int32_t mi_debug_set_tile_pretest(mi_ctx* ctx, int32_t mode) {
    ENTER(ctx);
    if (mode < 0 || mode > 2) return fail(ctx, MI_ERR_INVALID_ARG, "mi_debug_set_tile_pretest: mode %d", mode);
    ctx->tile_pretest_mode = mode;
    return MI_OK;
}

// test / bench hook: dense uploads in pieces with the GlobalTransforms fetched ahead (ctx.h): 0 = tables of >= 262144 rows (default),
// 1 = never, 2 = any row count, fetching ahead from the first sequence on (tests)
int32_t mi_debug_set_chunked_frames(mi_ctx* ctx, int32_t mode) {
    ENTER(ctx);
    if (mode < 0 || mode > 2) return fail(ctx, MI_ERR_INVALID_ARG, "mi_debug_set_chunked_frames: mode %d", mode);
    ctx->chunk_mode = mode;
    return MI_OK;
}

// test hook: how many dense windows went out as pieces of a sequence, and how many result downloads handed out GlobalTransforms fetched ahead
int32_t mi_debug_chunked_counts(mi_ctx* ctx, uint32_t* out_windows, uint32_t* out_downloads, uint32_t* out_sparse_downloads) {
    ENTER_RAW(ctx);
    if (out_windows) *out_windows = ctx->n_piece_uploads;
    if (out_downloads) *out_downloads = ctx->n_ahead_downloads;
    if (out_sparse_downloads) *out_sparse_downloads = ctx->n_sparse_ahead_downloads;
    return MI_OK;
}

// test / bench hook: the cluster walk of a MI_CULL_WITH_CLUSTERS frame whose objects are a row range: 0 = by the frame kernel's own row
// workgroups (default), 1 = by extra workgroups that re-derive the rows' visibility (what row-list bindings always take)
int32_t mi_debug_set_walk_inrow(mi_ctx* ctx, int32_t mode) {
    ENTER(ctx);
    if (mode < 0 || mode > 1) return fail(ctx, MI_ERR_INVALID_ARG, "mi_debug_set_walk_inrow: mode %d", mode);
    ctx->walk_inrow_mode = mode;
    return MI_OK;
}

// test / bench hook: the all-dirty hierarchy frame: 0 = the tiles cull their own rows when there is one view (default), 1 = never, 2 = whenever it applies
int32_t mi_debug_set_tree_cull(mi_ctx* ctx, int32_t mode) {
    ENTER(ctx);
    if (mode < 0 || mode > 2) return fail(ctx, MI_ERR_INVALID_ARG, "mi_debug_set_tree_cull: mode %d", mode);
    ctx->tree_cull_mode = mode;
    return MI_OK;
}

// test / bench hook: the per-wave summary of Aabb / flags / RenderLayers (RowSummary, kernels.h): 0 = in use (default), 1 = off
int32_t mi_debug_set_row_summary(mi_ctx* ctx, int32_t mode) {
    ENTER(ctx);
    if (mode < 0 || mode > 1) return fail(ctx, MI_ERR_INVALID_ARG, "mi_debug_set_row_summary: mode %d", mode);
    ctx->row_sum_mode = mode;
    return MI_OK;
}

// test / bench hook: the world-sphere path of the cull-only and changed-rows frames (k_frame_sph): 0 = used from the second
// frame in a row that rewrites no or few GlobalTransforms (default), 1 = never, 2 = at once
int32_t mi_debug_set_sphere_path(mi_ctx* ctx, int32_t mode) {
    ENTER(ctx);
    if (mode < 0 || mode > 2) return fail(ctx, MI_ERR_INVALID_ARG, "mi_debug_set_sphere_path: mode %d", mode);
    ctx->sph_mode = mode;
    return MI_OK;
}

// test / bench hook: the static cull order of cull-only frames (kernels_cells.hip): 0 = built by the second eligible frame in a row of a
// context with Cells::min_rows rows or more (default), 1 = never, 2 = at once, at any row count
int32_t mi_debug_set_static_cull_order(mi_ctx* ctx, int32_t mode) {
    ENTER(ctx);
    if (mode < 0 || mode > 3) return fail(ctx, MI_ERR_INVALID_ARG, "mi_debug_set_static_cull_order: mode %d", mode);
    ctx->cells.mode = mode;
    if (mode == 1) ctx->cells.valid = false;
    return MI_OK;
}
int32_t mi_debug_static_cull_counts(mi_ctx* ctx, uint32_t* out_builds, uint32_t* out_frames) {
    ENTER_RAW(ctx);
    if (out_builds) *out_builds = ctx->cells.builds;
    if (out_frames) *out_frames = ctx->cells.frames;
    return MI_OK;
}

// test hook: device logf probe (include/bevy_mi355x_debug.h; tests/test_gpu_cluster.py::test_device_logf_matches_libm)
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
