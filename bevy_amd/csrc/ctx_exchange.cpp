// ctx_exchange.cpp -- the multi-GPU exchange: in-place all-gather of the visibility masks behind every cull.
#include "ctx.h"

using namespace mi;
using namespace mi_detail;

namespace mi_detail {

// Side streams are picked empirically.  HIP maps streams onto a small pool of hardware queues; if a side stream lands
// on the compute stream's queue, its wait / record / write packets serialise with the frame kernels (measured: 32 us
// -> 48 us per frame), and which stream collides depends on how many streams the process created before.  So: make a
// few candidates (normal and high priority), drive each with the per-frame pattern over a stand-in kernel, and keep
// the n_keep fastest, fastest first.
// A side stream that shares the compute stream's hardware queue is worse than slow when it waits (hipStreamWaitValue32)
// for something the compute stream has yet to submit -- the released-by-the-next-frame-kernel scheme of the
// asynchronous compaction: the wait packet would sit in front of the very kernel that satisfies it.  That is probed
// explicitly: candidate waits on a pinned word, the compute stream is asked to write it; if the candidate does not get
// through within 20 ms the host writes the word itself (so the probe cannot hang) and the candidate is marked as
// sharing the queue.  Candidates that do not share it are preferred; out_shares_queue[i] reports the rest.
int32_t pick_side_streams(mi_ctx* ctx, hipStream_t* out, uint32_t n_keep, bool* out_shares_queue) {
    int prio_lo = 0, prio_hi = 0;
    HIP_TRY(ctx, hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
    constexpr int N_CAND = 6;
    hipStream_t cand[N_CAND] = {nullptr};
    for (int i = 0; i < N_CAND; ++i) {
        if (i == N_CAND - 1) HIP_TRY(ctx, hipStreamCreateWithPriority(&cand[i], hipStreamNonBlocking, prio_hi));
        else HIP_TRY(ctx, hipStreamCreateWithFlags(&cand[i], hipStreamNonBlocking));
    }
    hipEvent_t ev_a[2], ev_b[2];
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(ctx, hipEventCreateWithFlags(&ev_a[i], hipEventDisableTiming));
        HIP_TRY(ctx, hipEventCreateWithFlags(&ev_b[i], hipEventDisableTiming));
    }
    uint32_t* flag = nullptr;
    HIP_TRY(ctx, hipHostMalloc((void**)&flag, 64, hipHostMallocMapped));
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
                HIP_TRY(ctx, hipEventRecord(ev_a[it & 1u], ctx->stream));
                HIP_TRY(ctx, hipStreamWaitEvent(cand[i], ev_a[it & 1u], 0));
                HIP_TRY(ctx, hipEventRecord(ev_b[it & 1u], cand[i]));
                HIP_TRY(ctx, hipStreamWriteValue32(cand[i], (void*)flag, it + 1, 0));
            }
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(cand[i]));
            const double t = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (rep == 1) cand_t[i] = t;
            if (ctx->xch.debug) fprintf(stderr, "[mi side stream] candidate %d%s: %.1f us / frame\n", i,
                                                i == N_CAND - 1 ? " (high priority)" : "", t / 24.0);
        }
    HIP_TRY(ctx, hipFree(probe));
    bool shares[N_CAND];
    volatile uint32_t* w = flag;  // [0] the word the candidate waits on, [8] what it writes once through
    for (int i = 0; i < N_CAND; ++i) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(cand[i]));
        w[0] = 0;
        w[8] = 0;
        HIP_TRY(ctx, hipStreamWaitValue32(cand[i], (void*)&w[0], 1u, hipStreamWaitValueGte, 0xFFFFFFFFu));
        HIP_TRY(ctx, hipStreamWriteValue32(cand[i], (void*)&w[8], 1u, 0));
        HIP_TRY(ctx, hipStreamWriteValue32(ctx->stream, (void*)&w[0], 1u, 0));
        const auto t0 = std::chrono::steady_clock::now();
        while (w[8] == 0 && std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(20)) std::this_thread::yield();
        shares[i] = w[8] == 0;
        if (shares[i]) w[0] = 1;  // release the wait from the host: the compute stream's write is stuck behind it
        HIP_TRY(ctx, hipStreamSynchronize(cand[i]));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->xch.debug) fprintf(stderr, "[mi side stream] candidate %d %s the compute stream's hardware queue\n", i,
                                            shares[i] ? "SHARES" : "does not share");
    }
    HIP_TRY(ctx, hipHostFree(flag));
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(ctx, hipEventDestroy(ev_a[i]));
        HIP_TRY(ctx, hipEventDestroy(ev_b[i]));
    }
    int order[N_CAND];
    std::iota(order, order + N_CAND, 0);
    std::sort(order, order + N_CAND, [&](int a, int b) { return shares[a] != shares[b] ? !shares[a] : cand_t[a] < cand_t[b]; });
    for (int i = 0; i < N_CAND; ++i) {
        if (i < (int)n_keep) {
            out[i] = cand[order[i]];
            if (out_shares_queue) out_shares_queue[i] = shares[order[i]];
        } else {
            HIP_TRY(ctx, hipStreamDestroy(cand[order[i]]));
        }
    }
    return MI_OK;
}

// Multi-GPU exchange around a cull: bind this frame's gathered buffer (after its previous all-gather drained),
// and afterwards hand the in-place all-gather to the exchange thread, which enqueues it on the communication
// stream behind the kernels.
void exchange_worker(mi_ctx* ctx) {
    auto& x = ctx->xch;
    hipSetDevice(ctx->device);
    for (;;) {
        mi_ctx::Exchange::Job job{};
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
            job = x.queue.front();
            x.queue.pop_front();
        }
        const uint32_t slot = job.slot;
        int err = 0;
        const auto tw0 = std::chrono::steady_clock::now();
        const uint32_t k = (uint32_t)(x.worker_frames % x.n_comms);  // frame f travels on communicator f % n_comms
        hipStream_t cs = x.comm_stream[k];
        char* base = (char*)x.buf[slot];
        if (x.simple) {
            // plain event ordering, only on this thread: behind the frame's kernels (the caller recorded ev_kernels[slot] before it
            // handed the slot over), the all-gather, the event mi_exchange_last and the buffer's next user wait on
            if (hipStreamWaitEvent(cs, x.ev_kernels[slot], 0) != hipSuccess) err = -1;
            if (!err) err = x.all_gather(base + (size_t)x.rank * x.block_bytes, base, (size_t)x.block_bytes, 1 /* ncclUint8 */, x.comm[k], cs);
            if (hipEventRecord(x.ev_gathered[slot], cs) != hipSuccess && !err) err = -2;
        } else {
            if (hipStreamWaitValue32(cs, job.flag, job.value, hipStreamWaitValueGte, 0xFFFFFFFFu) != hipSuccess) err = -1;
            if (!err) err = x.all_gather(base + (size_t)x.rank * x.block_bytes, base, (size_t)x.block_bytes, 1 /* ncclUint8 */, x.comm[k], cs);
            if (hipEventRecord(x.ev_gathered[slot], cs) != hipSuccess && !err) err = -2;
            // completion counter the caller's thread can read without a driver call
            if (hipStreamWriteValue32(cs, (void*)(x.done_flag + k), (uint32_t)(x.worker_frames / x.n_comms + 1), 0) != hipSuccess && !err) err = -3;
        }
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
    if (x.simple) {
        // (checked here, before any of the frame's work is enqueued: a frame that cannot be exchanged is refused whole)
        if (x.grouped && x.group_pending)
            return fail(ctx, MI_ERR_NOT_READY, "MI_EXCHANGE_GROUPED: the previous frame's all-gather was never flushed (mi_exchange_group_flush)");
        // the buffer was last used n_bufs frames ago: its all-gather must have drained before the kernels overwrite it
        // (asked on the host first: with the buffers rotating that all-gather is normally long done, and a wait packet in the compute
        // queue costs ~6 us of every frame whether it has anything to wait for or not)
        if (x.frame >= x.n_bufs) {
            // (asynchronous enqueue: the event is only meaningful once the exchange thread has recorded it for the buffer's last use,
            // n_bufs frames ago -- normally long since)
            if (x.async_enqueue && !x.grouped) {
                const int32_t rcw = exchange_wait_issued(ctx, x.frame - x.n_bufs + 1);
                if (rcw) return rcw;
            }
            const hipError_t q = hipEventQuery(x.ev_gathered[slot]);
            if (q != hipSuccess) {
                (void)hipGetLastError();  // hipErrorNotReady is an answer, not a failure: it must not surface as the next launch's status
                HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, x.ev_gathered[slot], 0));
            }
        }
        ctx->ext_bitmask = x.buf[slot];
        ctx->ext_words_per_view = x.words_per_view;
        ctx->ext_word_offset = x.word_offset;
        return MI_OK;
    }
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
void exchange_push(mi_ctx* ctx, const mi_ctx::Exchange::Job& job) {
    auto& x = ctx->xch;
    {
        std::lock_guard<std::mutex> lk(x.m);
        x.queue.push_back(job);
        ++x.submitted;
    }
    x.submitted_fast.fetch_add(1, std::memory_order_seq_cst);
    if (x.sleeping.load(std::memory_order_seq_cst)) x.cv.notify_all();  // no futex call while the thread is polling
}

int32_t exchange_end(mi_ctx* ctx) {
    auto& x = ctx->xch;
    if (!x.on) return MI_OK;
    const uint32_t slot = (uint32_t)(x.frame % x.n_bufs);
    if (x.simple) {
        // masks complete = everything enqueued so far on the compute stream (a deferred compaction is not: it only reads them)
        HIP_TRY(ctx, hipEventRecord(x.ev_kernels[slot], ctx->stream));
        if (x.grouped) {  // one thread drives several contexts: their all-gathers go out together (mi_exchange_group_flush)
            if (x.group_pending) return fail(ctx, MI_ERR_NOT_READY, "MI_EXCHANGE_GROUPED: the previous frame's all-gather was never flushed (mi_exchange_group_flush)");
            x.group_pending = true;
            x.group_slot = slot;
            ++x.frame;
            return MI_OK;
        }
        if (x.async_enqueue) {  // the exchange thread enqueues wait + ncclAllGather + record (exchange_worker)
            exchange_push(ctx, mi_ctx::Exchange::Job{slot, nullptr, 0u});
            ++x.frame;
            return MI_OK;
        }
        HIP_TRY(ctx, hipStreamWaitEvent(x.comm_stream[0], x.ev_kernels[slot], 0));
        char* base = (char*)x.buf[slot];
        const int err = x.all_gather(base + (size_t)x.rank * x.block_bytes, base, (size_t)x.block_bytes, 1 /* ncclUint8 */, x.comm[0], x.comm_stream[0]);
        if (err) return fail(ctx, MI_ERR_DEVICE, "ncclAllGather failed (%d)", err);
        HIP_TRY(ctx, hipEventRecord(x.ev_gathered[slot], x.comm_stream[0]));
        ++x.frame;
        return MI_OK;
    }
    const auto te0 = std::chrono::steady_clock::now();
    if (!x.signalled) {  // nobody announces this frame's masks in-kernel: a write-value packet behind its kernels does
        x.wait_flag = x.kernels_flag;
        x.wait_value = (uint32_t)(x.frame + 1);
        HIP_TRY(ctx, hipStreamWriteValue32(ctx->stream, x.kernels_flag, x.wait_value, 0));
    }
    x.signalled = false;
    if (x.job_deferred) {
        // this frame's compaction -- and with it the "masks complete" signal -- rides in the next frame's launch: its
        // all-gather is handed to the exchange thread there (or by compaction_join), once that launch is submitted, so
        // the communication stream never waits for work that has not been submitted
        x.job_deferred = false;
    } else {
        exchange_push(ctx, mi_ctx::Exchange::Job{slot, x.wait_flag, x.wait_value});
    }
    ++x.frame;
    x.dbg_end_ns += std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - te0).count();
    return MI_OK;
}

}  // namespace mi_detail

extern "C" {

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

int32_t mi_exchange_set_mode(mi_ctx* ctx, uint32_t mode) {
    ENTER(ctx);
    if (mode > MI_EXCHANGE_GROUPED) return fail(ctx, MI_ERR_INVALID_ARG, "mi_exchange_set_mode: unknown mode %u", mode);
    if (ctx->xch.on) return fail(ctx, MI_ERR_NOT_READY, "mi_exchange_set_mode: switch the exchange off first (mi_exchange_configure with a NULL communicator)");
    ctx->xch.simple = mode != MI_EXCHANGE_PIPELINED;
    ctx->xch.grouped = mode == MI_EXCHANGE_GROUPED;
    return MI_OK;
}

int32_t mi_exchange_group_flush(mi_ctx* const* contexts, uint32_t n, void* fn_nccl_group_start, void* fn_nccl_group_end) {
    if (!contexts || n == 0 || !fn_nccl_group_start || !fn_nccl_group_end) return fail(nullptr, MI_ERR_INVALID_ARG, "mi_exchange_group_flush: NULL");
    typedef int (*group_fn)(void);
    for (uint32_t i = 0; i < n; ++i)
        if (!contexts[i] || !contexts[i]->xch.on || !contexts[i]->xch.grouped)
            return fail(contexts[i], MI_ERR_NOT_READY, "mi_exchange_group_flush: context %u is not in MI_EXCHANGE_GROUPED mode", i);
    int err = ((group_fn)fn_nccl_group_start)();
    if (err) return fail(contexts[0], MI_ERR_DEVICE, "ncclGroupStart failed (%d)", err);
    int32_t rc = MI_OK;
    for (uint32_t i = 0; i < n && rc == MI_OK; ++i) {
        mi_ctx* ctx = contexts[i];
        auto& x = ctx->xch;
        if (!x.group_pending) continue;
        if (hipSetDevice(ctx->device) != hipSuccess || hipStreamWaitEvent(x.comm_stream[0], x.ev_kernels[x.group_slot], 0) != hipSuccess) {
            rc = fail(ctx, MI_ERR_DEVICE, "mi_exchange_group_flush: HIP call failed on context %u", i);
            break;
        }
        char* base = (char*)x.buf[x.group_slot];
        err = x.all_gather(base + (size_t)x.rank * x.block_bytes, base, (size_t)x.block_bytes, 1 /* ncclUint8 */, x.comm[0], x.comm_stream[0]);
        if (err) rc = fail(ctx, MI_ERR_DEVICE, "ncclAllGather failed (%d)", err);
    }
    // The group is closed whatever happened inside it (an open group would swallow every later collective of this thread).  After a
    // failure the frame's exchange is lost on EVERY listed context -- RCCL gives no defined state for a group that was closed around a
    // failed call -- so none of them stays pending: the error is reported once, here, and the next frame starts clean (its masks go to
    // the next buffer; mi_exchange_last of the lost frame returns whatever the buffer holds, which the error told the caller not to use).
    err = ((group_fn)fn_nccl_group_end)();  // the collectives of every context are enqueued here, together
    if (rc || err) {
        for (uint32_t i = 0; i < n; ++i) contexts[i]->xch.group_pending = false;
        if (rc) return rc;
        return fail(contexts[0], MI_ERR_DEVICE, "ncclGroupEnd failed (%d)", err);
    }
    for (uint32_t i = 0; i < n; ++i) {
        mi_ctx* ctx = contexts[i];
        auto& x = ctx->xch;
        if (!x.group_pending) continue;
        x.group_pending = false;
        if (hipSetDevice(ctx->device) != hipSuccess || hipEventRecord(x.ev_gathered[x.group_slot], x.comm_stream[0]) != hipSuccess) {
            for (uint32_t k = i; k < n; ++k) contexts[k]->xch.group_pending = false;
            return fail(ctx, MI_ERR_DEVICE, "mi_exchange_group_flush: event record failed on context %u", i);
        }
    }
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
    {
        int32_t rcj = compaction_join(ctx);  // releases a pending last frame (asynchronous compaction) before draining
        if (rcj) return rcj;
    }
    void* const nccl_comm = (nccl_comms && n_comms) ? nccl_comms[0] : nullptr;
    if (n_comms > mi_ctx::Exchange::MAX_COMMS) return fail(ctx, MI_ERR_INVALID_ARG, "mi_exchange_configure: at most %u communicators", mi_ctx::Exchange::MAX_COMMS);
    for (uint32_t k = 0; k < n_comms; ++k)
        if (!nccl_comms[k]) return fail(ctx, MI_ERR_INVALID_ARG, "mi_exchange_configure: communicator %u is NULL", k);
    if (x.on) {  // drain whatever is in flight before changing anything
        int32_t rc0 = x.worker.joinable() ? exchange_wait_issued(ctx, x.frame) : MI_OK;
        exchange_stop(ctx);
        for (hipStream_t cs : x.comm_stream)
            if (cs) HIP_TRY(ctx, hipStreamSynchronize(cs));
        x.on = false;
        if (rc0) return rc0;
    }
    // buffers of an earlier mi_exchange_configure_owned: nothing is in flight any more (drained above), and this configuration either
    // brings its own buffers or -- called from mi_exchange_configure_owned, after its "off" call -- fresh ones
    if (!device_bufs || device_bufs != x.owned) {
        for (void*& p : x.owned)
            if (p) { hipFree(p); p = nullptr; }
        x.owned_bytes = 0;
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
    if (x.simple) {
        if (!x.comm_stream[0]) {
            // HIP maps streams onto a few hardware queues: a communication stream that lands on the compute stream's queue runs its
            // wait / all-gather / record IN LINE with the frame kernels (1-rank communicator, 1.25 M rows x 4 views: 45 us per frame
            // against 28 with a stream of its own).  Probed like the pipelined mode's streams; a property of this device alone.
            int32_t rcp = pick_side_streams(ctx, x.comm_stream, 1, x.comm_shares_queue);
            if (rcp) return rcp;
        }
        for (uint32_t i = 0; i < mi_ctx::Exchange::MAX_BUFS; ++i) {
            if (!x.ev_gathered[i]) HIP_TRY(ctx, hipEventCreateWithFlags(&x.ev_gathered[i], hipEventDisableTiming));
            if (!x.ev_kernels[i]) HIP_TRY(ctx, hipEventCreateWithFlags(&x.ev_kernels[i], hipEventDisableTiming));
        }
        x.all_gather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))fn_nccl_all_gather;
        x.n_comms = 1;  // one communicator: the first one given
        x.comm[0] = nccl_comms[0];
        x.n_bufs = n_bufs;
        for (uint32_t i = 0; i < n_bufs; ++i) x.buf[i] = device_bufs[i];
        x.words_per_view = words_per_view;
        x.word_offset = word_offset;
        x.block_bytes = block_bytes;
        x.rank = rank;
        x.frame = 0;
        x.kernel_signal = false;
        x.signalled = false;
        if (x.async_enqueue && !x.grouped) {
            x.worker_frames = 0;
            x.submitted = x.issued = 0;
            x.submitted_fast.store(0);
            x.worker_error = 0;
            x.queue.clear();
            x.worker = std::thread(exchange_worker, ctx);
        }
        x.on = true;
        ctx->culled = false;
        return MI_OK;
    }
    if (!x.done_flag) HIP_TRY(ctx, hipHostMalloc((void**)&x.done_flag, 64, hipHostMallocMapped));
    if (!x.kernels_flag) HIP_TRY(ctx, hipMalloc((void**)&x.kernels_flag, 64));
    HIP_TRY(ctx, hipMemsetAsync(x.kernels_flag, 0, 64, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    x.kernel_signal = true;
    x.signalled = false;
    if (!x.comm_stream[0]) {
        for (uint32_t i = 0; i < mi_ctx::Exchange::MAX_BUFS; ++i) {
            HIP_TRY(ctx, hipEventCreateWithFlags(&x.ev_gathered[i], hipEventDisableTiming));
        }
        int32_t rcp = pick_side_streams(ctx, x.comm_stream, mi_ctx::Exchange::MAX_COMMS, x.comm_shares_queue);
        if (rcp) return rcp;
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

int32_t mi_exchange_configure_owned(mi_ctx* ctx, void* const* nccl_comms, uint32_t n_comms, void* fn_nccl_all_gather, uint32_t n_bufs,
                                    uint32_t world, uint64_t words_per_view, uint64_t word_offset, uint64_t block_bytes, uint32_t rank) {
    ENTER(ctx);
    auto& x = ctx->xch;
    if (n_bufs < 2 || n_bufs > mi_ctx::Exchange::MAX_BUFS || world == 0 || block_bytes == 0)
        return fail(ctx, MI_ERR_INVALID_ARG, "mi_exchange_configure_owned: n_bufs outside 2..8, world 0 or empty block");
    // off first: whatever is in flight drains before the buffers it uses are freed
    int32_t rc = mi_exchange_configure_multi(ctx, nullptr, 0, nullptr, nullptr, 0, 0, 0, 0, 0);
    if (rc) return rc;
    const size_t bytes = (size_t)world * block_bytes;
    for (uint32_t i = 0; i < n_bufs; ++i) {
        HIP_TRY(ctx, hipMalloc(&x.owned[i], bytes));
        HIP_TRY(ctx, hipMemsetAsync(x.owned[i], 0, bytes, ctx->stream));
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    rc = mi_exchange_configure_multi(ctx, nccl_comms, n_comms, fn_nccl_all_gather, x.owned, n_bufs, words_per_view, word_offset, block_bytes, rank);
    if (rc == MI_OK) x.owned_bytes = bytes;
    return rc;
}

// bench hook (include/bevy_mi355x_debug.h): what the exchange cost the CALLING thread so far -- out[0] frames, out[1] ns inside the
// begin step, out[2] of which spent waiting for a gathered buffer's previous all-gather to complete (back-pressure of the device: the
// pipelined mode paces the caller n_bufs - 1 frames ahead, so a device-bound frame shows up here, not as work), out[3] ns inside the end
// step, out[4] ns the exchange thread spent enqueueing.  reset != 0 zeroes the counters.
int32_t mi_debug_exchange_times(mi_ctx* ctx, double* out5, int32_t reset) {
    ENTER_RAW(ctx);
    auto& x = ctx->xch;
    if (out5) {
        out5[0] = (double)x.frame;
        out5[1] = x.dbg_begin_ns;
        out5[2] = x.dbg_wait_ns;
        out5[3] = x.dbg_end_ns;
        out5[4] = x.dbg_worker_ns;
    }
    if (reset) x.dbg_begin_ns = x.dbg_wait_ns = x.dbg_end_ns = x.dbg_worker_ns = 0.0;
    return MI_OK;
}

int32_t mi_exchange_last(mi_ctx* ctx, void** out_device_buf, int32_t wait) {
    ENTER(ctx);
    auto& x = ctx->xch;
    if (!x.on || x.frame == 0) return fail(ctx, MI_ERR_NOT_READY, "mi_exchange_last: no exchanged frame yet");
    const uint32_t slot = (uint32_t)((x.frame - 1) % x.n_bufs);
    int32_t rc = compaction_join(ctx);  // asynchronous compaction: also what releases the last frame's all-gather
    if (rc) return rc;
    if ((!x.simple || (x.async_enqueue && !x.grouped)) && (rc = exchange_wait_issued(ctx, x.frame))) return rc;
    if (x.grouped && x.group_pending) return fail(ctx, MI_ERR_NOT_READY, "mi_exchange_last: the frame's all-gather is still pending (mi_exchange_group_flush)");
    if (wait) HIP_TRY(ctx, hipEventSynchronize(x.ev_gathered[slot]));
    if (out_device_buf) *out_device_buf = x.buf[slot];
    return MI_OK;
}

int32_t mi_exchange_download(mi_ctx* ctx, void* out_host, uint64_t bytes) {
    void* buf = nullptr;
    int32_t rc = mi_exchange_last(ctx, &buf, 0);  // (ENTER, joins a deferred compaction, waits until the all-gather is ISSUED)
    if (rc) return rc;
    if (!out_host || bytes == 0) return fail(ctx, MI_ERR_INVALID_ARG, "mi_exchange_download: NULL or empty");
    auto& x = ctx->xch;
    if (x.owned_bytes && bytes > x.owned_bytes)  // (buffers the caller bound: their size is the caller's to know, include/bevy_mi355x.h)
        return fail(ctx, MI_ERR_INVALID_ARG, "mi_exchange_download: %llu bytes asked of a gathered buffer of %llu (world * block_bytes)",
                    (unsigned long long)bytes, (unsigned long long)x.owned_bytes);
    const uint32_t slot = (uint32_t)((x.frame - 1) % x.n_bufs);
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, x.ev_gathered[slot], 0));  // the copy runs on the context's stream, behind the collective
    return download(ctx, out_host, buf, (size_t)bytes);
}

}  // extern "C"
