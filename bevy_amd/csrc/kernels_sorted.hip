// kernels_sorted.hip -- sorted render phases (Transparent3d, the 2D phases):
//   gpu_preprocessing::batch_and_prepare_sorted_render_phase   crates/bevy_render/src/batching/gpu_preprocessing.rs:1850-2061
//   the range merge of batching::batch_and_prepare_sorted_render_phase (batching/mod.rs:219-244) as no_gpu_preprocessing.rs:76-103
//   drives it (merge_only)
//
// The reference walks the phase's items in order with a running batch set / batch: whether an item continues its predecessor's batch,
// breaks the batch or starts a new batch set depends on the two items alone, every index the walk hands out is a count of items
// before it, and "the set I belong to" is the latest set head at or before me.  So the walk is a scan.  Round 2 ran it as rounds of
// eight-plane block scans over 256 items with the per-item results parked in eight scratch planes for a second phase to read back:
// a 4 096-item phase took 16 rounds (60 us on one workgroup) or four tiles in two launches (~33 us of kernels behind a 5 us copy) --
// HALF the speed of one CPU core.  Here a thread walks ITEMS_PER_THREAD consecutive items itself, the way the CPU does,
// twice:
//   pass 0  (coalesced) the tile's items -> one byte of code per item (OK / BREAK_BATCH / HEAD / SKIP, mesh class) and its input
//           index, in LDS; an item's predecessor is its neighbour lane's item
//   pass 1  every thread adds up its items' eight counters and notes its last set head with the counts in front of it
//           ONE block scan gives every thread its starting counts; the threads' last heads, made absolute, go to LDS and a
//           max-scan tells every thread which thread holds the head that governs its first items
//   pass 2  every thread walks its items again with running counts and the current head's record in registers and writes the work
//           items, the indirect-parameters metadata and -- at a set's last item -- the batch and the batch set: no scratch planes, no
//           read-back
// Up to 8 192 items are ONE launch of one workgroup that reads the items straight from the pinned staging block (no copy launch in
// front: mi_batch_sorted_build); longer phases take tiles of 4 096 items in two launches (per-tile sums and last head, then the walk
// with the tiles in front added up), as before but with 4 x fewer, 4 x cheaper tiles.
// Integer work, bit-exact against oracle/batching_oracle.c (tests/test_gpu_batching.py runs every form on the same phases).
#include "kernels.h"

namespace mi {
namespace {

constexpr uint32_t S_OK = 0, S_BREAK = 1, S_HEAD = 2, S_SKIP = 3;
constexpr uint32_t NONE = 0xFFFFFFFFu;

// pass 0: what an item is, from the item and its predecessor (gpu_preprocessing.rs:1921-1990: the comparison of the current batch
// set's / batch's key with the item's)
__device__ __forceinline__ uint32_t sorted_code(const uint4 it, const uint4 pv, bool has_prev, uint32_t automatic_batching, bool indirect) {
    const bool has_in = it.x != NONE;
    const bool meta = has_in && automatic_batching && (it.w & 2u);
    const bool prev_meta = has_prev && pv.x != NONE && automatic_batching && (pv.w & 2u);
    uint32_t flag = S_SKIP;
    if (has_in) {
        flag = S_HEAD;
        if (meta && prev_meta && it.y == pv.y) {
            if (it.z == pv.z) flag = S_OK;
            else if (indirect) flag = S_BREAK;  // without indirect drawing a different mesh is a new batch set; the merge-only rule has no second level either
        }
    }
    return flag | ((it.w & 1u) << 2);
}
// the eight counters of the walk: with input | ... of class 0 | of class 1 | allocations of class 0 | of class 1 | batch breaks |
// set heads | set heads of class 1
__device__ __forceinline__ void sorted_counts(uint32_t code, bool indirect, uint32_t (&v)[8]) {
    const uint32_t flag = code & 3u, cls = (code >> 2) & 1u;
    const bool has_in = flag != S_SKIP, alloc = indirect && (flag == S_HEAD || flag == S_BREAK);
    v[0] = has_in ? 1u : 0u;
    v[1] = (has_in && !cls) ? 1u : 0u;
    v[2] = (has_in && cls) ? 1u : 0u;
    v[3] = (alloc && !cls) ? 1u : 0u;
    v[4] = (alloc && cls) ? 1u : 0u;
    v[5] = flag == S_BREAK ? 1u : 0u;
    v[6] = flag == S_HEAD ? 1u : 0u;
    v[7] = (flag == S_HEAD && cls) ? 1u : 0u;
}
// what the walk remembers of the current batch set's head
struct HeadRec {
    uint32_t h, out, ip, brk, k, sets1, cls;
};
__device__ __forceinline__ HeadRec head_record(const SortedArgs& a, uint32_t pos, uint32_t cls, const uint32_t (&e)[8], bool indirect) {
    HeadRec r;
    r.h = pos;
    r.out = a.initial.output_mesh_uniform_index + e[0];
    r.ip = indirect ? (cls ? a.initial.indirect_parameters_index[1] + e[4] : a.initial.indirect_parameters_index[0] + e[3]) : NONE;
    r.brk = e[5];
    r.k = e[6];
    r.sets1 = e[7];
    r.cls = cls;
    return r;
}

// K sums / exclusive scans over the THREADS threads (two barriers)
template <uint32_t K, uint32_t THREADS>
__device__ __forceinline__ void block_scan(uint32_t (&v)[K], uint32_t (*lds)[K], uint32_t (&total)[K]) {
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    uint32_t incl[K];
#pragma unroll
    for (uint32_t q = 0; q < K; ++q) {
        incl[q] = v[q];
#pragma unroll
        for (uint32_t off = 1; off < 64u; off <<= 1) {
            const uint32_t up = __shfl_up(incl[q], off, 64);
            if (lane >= off) incl[q] += up;
        }
    }
    __syncthreads();  // (lds may still be read by the call before)
    if (lane == 63u)
#pragma unroll
        for (uint32_t q = 0; q < K; ++q) lds[wv][q] = incl[q];
    __syncthreads();
#pragma unroll
    for (uint32_t q = 0; q < K; ++q) {
        uint32_t before = 0, all = 0;
#pragma unroll
        for (uint32_t k = 0; k < THREADS / 64u; ++k) {
            const uint32_t w = lds[k][q];
            before += k < wv ? w : 0u;
            all += w;
        }
        total[q] = all;
        v[q] = before + incl[q] - v[q];
    }
}

constexpr uint32_t SORTED_PARTIAL_WORDS = 20;  // sums[8] | last head position + 1 (0 = none) | counts in front of it inside the tile [8] | its class | pad

// IPT consecutive items per thread, THREADS threads: a tile is THREADS * IPT items.  PARTIALS: only the tile's record (first launch
// of the tiled form); else the walk (tile > 0 adds up the records in front first).  512 threads x 8 items: the serial part of a
// thread is short and eight waves hide each other's waits (256 threads x 16 items took 33 us for 4 096 items, most of it waiting;
// 1 024 threads x 4 are held to 128 registers and spill 285).
template <uint32_t THREADS, uint32_t IPT, bool PARTIALS>
__global__ void __launch_bounds__(THREADS) k_sorted_walk(SortedArgs a, uint32_t* __restrict__ partials, uint32_t n_tiles) {
    constexpr uint32_t T = THREADS * IPT, WAVES = THREADS / 64u;
    static_assert(IPT % 4u == 0u, "a thread's codes are whole words");
    __shared__ __attribute__((aligned(16))) uint8_t s_code[T + 16];
    __shared__ __attribute__((aligned(16))) uint32_t s_input[PARTIALS ? 4 : T];
    __shared__ uint32_t s_scan[WAVES][8];
    __shared__ uint32_t s_pay[PARTIALS ? 1 : THREADS][8];
    __shared__ uint32_t s_owner[WAVES];
    const bool indirect = a.no_indirect == 0u && a.merge_only == 0u;
    const uint32_t tile = blockIdx.x, lo = tile * T, hi = lo + T < a.n_items ? lo + T : a.n_items;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const uint4* items = reinterpret_cast<const uint4*>(a.items);

    // ---- pass 0: codes and input indices into LDS (lane-contiguous 16-byte loads, all of a thread's requests in flight together;
    //      the predecessor is the neighbour lane's item) ----
    {
        uint4 it[IPT], p0[IPT];
#pragma unroll
        for (uint32_t k = 0; k < IPT; ++k) {
            const uint32_t i = lo + k * THREADS + tid;
            it[k] = i < hi ? items[i] : make_uint4(NONE, 0u, 0u, 0u);
            p0[k] = (lane == 0u && i < hi && i) ? items[i - 1u] : make_uint4(NONE, 0u, 0u, 0u);
        }
#pragma unroll
        for (uint32_t k = 0; k < IPT; ++k) {
            const uint32_t i = lo + k * THREADS + tid;
            uint4 pv;
            pv.x = __shfl_up(it[k].x, 1, 64);
            pv.y = __shfl_up(it[k].y, 1, 64);
            pv.z = __shfl_up(it[k].z, 1, 64);
            pv.w = __shfl_up(it[k].w, 1, 64);
            if (lane == 0u) pv = p0[k];
            if (i < hi) {
                s_code[i - lo] = (uint8_t)sorted_code(it[k], pv, i != 0u, a.automatic_batching, indirect);
                if constexpr (!PARTIALS) s_input[i - lo] = it[k].x;
            } else {
                s_code[i - lo] = (uint8_t)S_SKIP;
            }
        }
    }
    if (tid == 0u) {  // the item behind the tile: the tile's last item needs to know whether it ends a set
        uint32_t c = S_SKIP;  // (nothing behind: the last item ends its set)
        if (hi < a.n_items && hi == lo + T) c = sorted_code(items[hi], items[hi - 1u], true, a.automatic_batching, indirect);
        s_code[T] = (uint8_t)c;
    }
    __syncthreads();

    // ---- pass 1: this thread's items [j0, j0 + IPT) of the tile: sums, last head ----
    const uint32_t j0 = tid * IPT;
    uint32_t sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t head_at[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t head_pos = NONE, head_cls = 0;
    {
        uint32_t codes[IPT / 4];
#pragma unroll
        for (uint32_t q = 0; q < IPT / 4u; ++q) codes[q] = reinterpret_cast<const uint32_t*>(s_code)[tid * (IPT / 4u) + q];
#pragma unroll
        for (uint32_t k = 0; k < IPT; ++k) {
            const uint32_t code = (codes[k >> 2] >> (8u * (k & 3u))) & 0xFFu;
            uint32_t v[8];
            sorted_counts(code, indirect, v);
            if ((code & 3u) == S_HEAD) {
                head_pos = j0 + k;
                head_cls = (code >> 2) & 1u;
#pragma unroll
                for (uint32_t q = 0; q < 8u; ++q) head_at[q] = sum[q];
            }
#pragma unroll
            for (uint32_t q = 0; q < 8u; ++q) sum[q] += v[q];
        }
    }
    uint32_t start[8], tot[8];
#pragma unroll
    for (uint32_t q = 0; q < 8u; ++q) start[q] = sum[q];
    block_scan<8, THREADS>(start, s_scan, tot);  // start = the counts in front of this thread's items, inside the tile

    // which thread holds the last head at or before this thread's first item (exclusive), and the tile's last head
    uint32_t own = head_pos != NONE ? tid + 1u : 0u;
#pragma unroll
    for (uint32_t off = 1; off < 64u; off <<= 1) {
        const uint32_t up = __shfl_up(own, off, 64);
        if (lane >= off && up > own) own = up;
    }
    if (lane == 63u) s_owner[wv] = own;
    uint32_t own_excl = __shfl_up(own, 1, 64);
    if (lane == 0u) own_excl = 0u;
    __syncthreads();
    uint32_t waves_before = 0u, tile_owner = 0u;
#pragma unroll
    for (uint32_t k = 0; k < WAVES; ++k) {
        const uint32_t o = s_owner[k];
        waves_before = (k < wv && o > waves_before) ? o : waves_before;
        tile_owner = o > tile_owner ? o : tile_owner;
    }
    own_excl = own_excl > waves_before ? own_excl : waves_before;

    if constexpr (PARTIALS) {
        uint32_t* p = partials + (size_t)tile * SORTED_PARTIAL_WORDS;
        if (tid < 8u) p[tid] = tot[tid];  // (tot is uniform across the workgroup)
        if (tile_owner == 0u) {
            if (tid == 0u) p[8] = 0u;
        } else if (tid + 1u == tile_owner) {
            p[8] = lo + head_pos + 1u;
#pragma unroll
            for (uint32_t q = 0; q < 8u; ++q) p[9u + q] = start[q] + head_at[q];
            p[17] = head_cls;
        }
        return;
    } else {
        // ---- what the tiles in front leave: their sums, and the last head among them as an absolute record ----
        uint32_t base[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        HeadRec in_head{NONE, 0, 0, 0, 0, 0, 0};
        if (tile) {  // (workgroup-uniform)
            uint32_t v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // [8] = 1 + the last tile in front that holds a head
            for (uint32_t t = tid; t < tile; t += THREADS) {
                const uint32_t* p = partials + (size_t)t * SORTED_PARTIAL_WORDS;
#pragma unroll
                for (uint32_t q = 0; q < 8u; ++q) v[q] += p[q];
                if (p[8]) v[8] = t + 1u;  // (t ascends per thread)
            }
#pragma unroll
            for (uint32_t off = 32u; off; off >>= 1) {
#pragma unroll
                for (uint32_t q = 0; q < 8u; ++q) v[q] += __shfl_xor(v[q], off, 64);
                const uint32_t o = __shfl_xor(v[8], off, 64);
                v[8] = o > v[8] ? o : v[8];
            }
            __syncthreads();
            if (lane == 0u)
#pragma unroll
                for (uint32_t q = 0; q < 8u; ++q) s_scan[wv][q] = v[q];
            if (lane == 0u) s_owner[wv] = v[8];
            __syncthreads();
            uint32_t th1 = 0u;
#pragma unroll
            for (uint32_t k = 0; k < WAVES; ++k) {
#pragma unroll
                for (uint32_t q = 0; q < 8u; ++q) base[q] += s_scan[k][q];
                th1 = s_owner[k] > th1 ? s_owner[k] : th1;
            }
            if (th1) {  // the sums in front of THAT tile, then its record
                uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (uint32_t t = tid; t + 1u < th1; t += THREADS) {
                    const uint32_t* p = partials + (size_t)t * SORTED_PARTIAL_WORDS;
#pragma unroll
                    for (uint32_t q = 0; q < 8u; ++q) w[q] += p[q];
                }
#pragma unroll
                for (uint32_t off = 32u; off; off >>= 1)
#pragma unroll
                    for (uint32_t q = 0; q < 8u; ++q) w[q] += __shfl_xor(w[q], off, 64);
                __syncthreads();
                if (lane == 0u)
#pragma unroll
                    for (uint32_t q = 0; q < 8u; ++q) s_scan[wv][q] = w[q];
                __syncthreads();
                const uint32_t* p = partials + (size_t)(th1 - 1u) * SORTED_PARTIAL_WORDS;
                uint32_t e[8];
#pragma unroll
                for (uint32_t q = 0; q < 8u; ++q) {
                    e[q] = p[9u + q];
#pragma unroll
                    for (uint32_t k = 0; k < WAVES; ++k) e[q] += s_scan[k][q];
                }
                in_head = head_record(a, p[8] - 1u, p[17], e, indirect);
            }
            __syncthreads();  // (s_scan / s_owner are free again)
        }
        // ---- the threads' last heads, absolute, for the threads behind them ----
        if (head_pos != NONE) {
            uint32_t e[8];
#pragma unroll
            for (uint32_t q = 0; q < 8u; ++q) e[q] = base[q] + start[q] + head_at[q];
            const HeadRec r = head_record(a, lo + head_pos, head_cls, e, indirect);
            s_pay[tid][0] = r.h, s_pay[tid][1] = r.out, s_pay[tid][2] = r.ip, s_pay[tid][3] = r.brk, s_pay[tid][4] = r.k, s_pay[tid][5] = r.sets1,
            s_pay[tid][6] = r.cls;
        }
        __syncthreads();
        HeadRec cur = in_head;
        if (own_excl) {
            const uint32_t* r = s_pay[own_excl - 1u];
            cur = HeadRec{r[0], r[1], r[2], r[3], r[4], r[5], r[6]};
        }
        // ---- pass 2: the walk ----
        uint32_t e[8];
#pragma unroll
        for (uint32_t q = 0; q < 8u; ++q) e[q] = base[q] + start[q];
        uint32_t codes[IPT / 4 + 1];
#pragma unroll
        for (uint32_t q = 0; q < IPT / 4u; ++q) codes[q] = reinterpret_cast<const uint32_t*>(s_code)[tid * (IPT / 4u) + q];
        codes[IPT / 4] = s_code[j0 + IPT];  // the item behind this thread's last one (s_code[T] for the last thread)
#pragma unroll
        for (uint32_t k = 0; k < IPT; ++k) {
            const uint32_t i = lo + j0 + k;
            const uint32_t code = (codes[k >> 2] >> (8u * (k & 3u))) & 0xFFu;
            const uint32_t next = (codes[(k + 1u) >> 2] >> (8u * ((k + 1u) & 3u))) & 3u;
            const uint32_t flag = code & 3u, cls = (code >> 2) & 1u;
            uint32_t v[8];
            sorted_counts(code, indirect, v);
            if (i < hi && flag != S_SKIP) {
                const uint32_t out = a.initial.output_mesh_uniform_index + e[0];
                const uint32_t ip = (indirect && flag != S_OK) ? (cls ? a.initial.indirect_parameters_index[1] + e[4] : a.initial.indirect_parameters_index[0] + e[3]) : NONE;
                if (flag == S_HEAD) cur = head_record(a, i, cls, e, indirect);
                const uint32_t brk_incl = e[5] + v[5];                 // batch breaks up to and including this item
                const uint32_t ip_cur = cur.ip + (brk_incl - cur.brk);  // indirect_parameters_index_range.end - 1
                if (!a.merge_only) {
                    if (indirect && flag != S_OK) {  // a new batch: its IndirectParametersMetadata (:1995-2010)
                        uint32_t* md = (cls ? a.metadata[1] : a.metadata[0]) + 5u * ip;
                        md[0] = out;
                        md[1] = NONE;
                        md[2] = md[3] = md[4] = 0u;
                    }
                    uint32_t* wi = (cls ? a.work_items[1] : a.work_items[0]) + 2u * (cls ? a.initial.work_item_index[1] + e[2] : a.initial.work_item_index[0] + e[1]);
                    wi[0] = s_input[j0 + k];
                    wi[1] = indirect ? ip_cur : out;
                }
                // the set's last item: the next item is missing, has no input index, or heads a new set -> what flush() leaves (:1767-1794)
                if (i + 1u == a.n_items || next == S_SKIP || next == S_HEAD) {
                    uint32_t* b = a.batches + 6u * cur.k;
                    b[0] = cur.h;
                    b[1] = cur.out;
                    b[2] = out + 1u;
                    b[3] = indirect ? cur.ip : NONE;
                    b[4] = indirect ? ip_cur + 1u : NONE;
                    b[5] = cur.cls;
                    if (indirect) {  // add_batch_set at flush, in flush order per class (:1787-1793)
                        const uint32_t slot = cur.cls ? a.initial.batch_set_index[1] + cur.sets1 : a.initial.batch_set_index[0] + (cur.k - cur.sets1);
                        uint32_t* bset = (cur.cls ? a.batch_sets[1] : a.batch_sets[0]) + 2u * slot;
                        bset[0] = 0u;
                        bset[1] = cur.ip;
                    }
                }
            }
#pragma unroll
            for (uint32_t q = 0; q < 8u; ++q) e[q] += v[q];
        }
        if (tile + 1u == n_tiles && tid == 0u) {
            uint32_t c[8];
#pragma unroll
            for (uint32_t q = 0; q < 8u; ++q) c[q] = base[q] + tot[q];
            a.totals[0] = a.initial.work_item_index[0] + (a.merge_only ? 0u : c[1]);
            a.totals[1] = a.initial.work_item_index[1] + (a.merge_only ? 0u : c[2]);
            a.totals[2] = a.initial.indirect_parameters_index[0] + c[3];
            a.totals[3] = a.initial.indirect_parameters_index[1] + c[4];
            a.totals[4] = a.initial.batch_set_index[0] + (indirect ? c[6] - c[7] : 0u);
            a.totals[5] = a.initial.batch_set_index[1] + (indirect ? c[7] : 0u);
            a.totals[6] = a.initial.output_mesh_uniform_index + c[0];
            a.totals[7] = c[6];
            a.totals[8] = 0u;
        }
    }
}

}  // namespace

uint32_t batch_sorted_partial_words(uint32_t n_items) { return ((n_items + SORTED_TILE_ITEMS - 1u) / SORTED_TILE_ITEMS + 1u) * SORTED_PARTIAL_WORDS; }

hipError_t launch_batch_sorted(const SortedArgs& a, hipStream_t stream, void (*mark)(void*, uint32_t), void* mctx, uint32_t* partials, uint32_t one_wg_limit) {
    if (mark) mark(mctx, K_BATCH_SORTED);
    if (a.n_items <= 4096u && a.n_items <= one_wg_limit) {  // (an empty phase too: the totals are written)
        MI_LAUNCH((k_sorted_walk<512, 8, false>), dim3(1), dim3(512), 0, stream, a, partials, 1u);
    } else if (a.n_items <= SORTED_ONE_WG_ITEMS && a.n_items <= one_wg_limit) {
        MI_LAUNCH((k_sorted_walk<512, 16, false>), dim3(1), dim3(512), 0, stream, a, partials, 1u);
    } else if (!partials) {
        return hipErrorInvalidValue;
    } else {
        const uint32_t n_tiles = a.n_items ? (a.n_items + SORTED_TILE_ITEMS - 1u) / SORTED_TILE_ITEMS : 1u;
        if (mark) mark(mctx, K_BATCH_SCAN);  // (timer slots: the tiles' records under k_batch_scan, the walk under k_batch_sorted)
        MI_LAUNCH((k_sorted_walk<512, 8, true>), dim3(n_tiles), dim3(512), 0, stream, a, partials, n_tiles);
        if (mark) mark(mctx, K_BATCH_SORTED);
        MI_LAUNCH((k_sorted_walk<512, 8, false>), dim3(n_tiles), dim3(512), 0, stream, a, partials, n_tiles);
    }
    if (mark) mark(mctx, K_NUM_KERNELS);
    return hipGetLastError();
}

}  // namespace mi
