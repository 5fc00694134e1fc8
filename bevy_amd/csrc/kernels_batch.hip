// kernels_batch.hip -- batching work-item build on gfx950.
//
// What the reference does for one view's binned phase (crates/bevy_render/src/batching/gpu_preprocessing.rs:2079-2455):
//   CPU loops  unbatchable bins (:2148-2224), batchable bins (:2228-2353): per entity a MeshUniform slot and a PreprocessWorkItem,
//              per unbatchable entity / per batchable bin an indirect-parameters slot and an IndirectBatchSet
//   then       for every multidrawable batch set (:2360-2447, :2497-2580) a run of work items, MeshUniform slots and one
//              indirect-parameters slot per bin, followed by two compute passes per set -- allocate_uniforms.wesl (exclusive prefix of
//              the bins' instance counts -> IndirectParametersMetadata.base_output_index) and unpack_bins.wesl (one work item per
//              binned mesh instance)
// and for a sorted phase (:1850-2061) one walk over the items that cuts them into batch sets and batches.
// The bins themselves are CPU hash maps kept in step with VisibleEntities (render_phase/mod.rs:268-400).
//
// Here the bins are derived on the device from the VisibleEntities list the cull pass left in HBM.  Every place a listed row can
// go is a BUCKET, numbered in the order the reference visits them:
//     [2b, 2b+1]   unbatchable bin b: rows with / without an input uniform index (the latter only count towards allocate(len))
//     [2U + b]     batchable bin b
//     [2U + B + s] multidrawable batch set s
// and every index the reference hands out while it walks is an exclusive prefix sum over the buckets.  A frame's build is
//   1. k_batch_hist        per 1024-row tile: rows per bucket digit (LDS histogram, one contiguous KB per tile), instances per
//                          multidrawable bin (integer adds: order-independent, exact; pre-aggregated per tile in a 4096-slot LDS
//                          hash table and added to counters that sit one per 64-byte line, because agent-scope atomics on one
//                          cache line serialise at ~25 ns each on this part and many_cubes has ONE bin)
//   2. k_batch_plan_emit   every workgroup first derives the PLAN in LDS -- the partition offsets (scan of the tile histograms) and
//                          nine exclusive scans over the buckets = all of the reference's per-bin / per-set bookkeeping: where each
//                          bucket's work items, MeshUniform slots, indirect-parameters slots and batch sets start -- then performs
//                          the stable scatter of its own tile, except that a row's final position is not stored: position - bucket
//                          start is the row's ordinal in its bin, which is all its work item (and, for an unbatchable entity, its
//                          metadata and batch set) needs.  allocate_uniforms for every non-empty batch set rides along as extra
//                          workgroups; workgroup 0 writes the plan's own outputs (batchable bins' metadata, batch sets, records,
//                          buffer lengths).
// Two launches when there are at most 256 buckets.  With more (up to 65 536) the partition takes two LSD passes and the last pass
// stores the partitioned list (hist, scan, scatter, hist, scan, scatter, bounds, k_batch_plan, k_batch_emit from the list): rare,
// not tuned.
// The partition is stable, so a bucket's rows keep the list order (ascending Entity) -- the order the oracle uses; the reference
// leaves the order inside a multidrawable set unspecified (unpack_bins.wesl:31-33).
// Everything is u32; HBM traffic is a few words per visible row, the kernels are latency-, not bandwidth-bound.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace mi {
namespace {

constexpr uint32_t KIND_UNB_INPUT = 0, KIND_UNB_NO_INPUT = 1, KIND_BATCHABLE = 2, KIND_SET = 3;
__device__ __forceinline__ uint32_t desc_kind(uint32_t d) { return d & 3u; }
__device__ __forceinline__ uint32_t desc_cls(uint32_t d) { return (d >> 2) & 1u; }
__device__ __forceinline__ uint32_t desc_id(uint32_t d) { return d >> 3; }

__device__ __forceinline__ uint32_t list_len(const BatchArgs& a, bool pass0) { return pass0 ? *a.list_count : a.counters[0]; }
__device__ __forceinline__ const uint32_t* list_src(const BatchArgs& a, uint32_t pass) {
    if (pass == 0) return a.list + (a.list_base ? *a.list_base : 0ull);
    return a.rows_a;
}

// row -> bucket and, for multidrawable rows, index of its bin's GpuBinMetadata in the concatenated array (once per upload)
__global__ void __launch_bounds__(256) k_batch_resolve_rows(uint32_t n, uint32_t n_sets, uint32_t n_unbatchable, uint32_t n_batchable,
                                                            const uint8_t* row_kind, const uint32_t* row_cpu_bin, const uint32_t* row_set,
                                                            const uint32_t* row_bin, const uint32_t* row_input,
                                                            const uint32_t* bin_table_offset, const uint32_t* bin_table,
                                                            const uint32_t* meta_offset, uint32_t* row_meta, uint32_t* row_bucket) {
    const uint32_t row = blockIdx.x * 256u + threadIdx.x;
    if (row >= n) return;
    const uint32_t kind = row_kind[row];
    uint32_t m = 0xFFFFFFFFu, bucket = BATCH_NO_SET;
    if (kind == 2u) {  // MI_BATCH_ROW_UNBATCHABLE
        const uint32_t b = row_cpu_bin[row];
        if (b < n_unbatchable) bucket = 2u * b + (row_input[row] == 0xFFFFFFFFu ? 1u : 0u);
    } else if (kind == 1u) {  // MI_BATCH_ROW_BATCHABLE
        const uint32_t b = row_cpu_bin[row];
        if (b < n_batchable) bucket = 2u * n_unbatchable + b;
    } else if (kind == 0u) {
        const uint32_t s = row_set[row];
        if (s < n_sets) {
            const uint32_t slots = bin_table_offset[s + 1] - bin_table_offset[s], bins = meta_offset[s + 1] - meta_offset[s];
            const uint32_t b = row_bin[row];
            const uint32_t k = b < slots ? bin_table[bin_table_offset[s] + b] : 0xFFFFFFFFu;
            m = k < bins ? meta_offset[s] + k : 0xFFFFFFFFu;  // a hole or an index out of range: the row is treated as unbatched
            if (m != 0xFFFFFFFFu) bucket = 2u * n_unbatchable + n_batchable + s;
        }
    }
    row_meta[row] = m;
    row_bucket[row] = bucket;
}

// (a tile of 1024 rows can name 1024 bins: at 512 slots a many-bin scene filled the table, every later row probed eight
// times in vain -- an LDS round trip each -- and went to memory anyway)
constexpr uint32_t BIN_HASH = 4096, BIN_HASH_SHIFT = 20, BIN_HASH_EMPTY = 0xFFFFFFFFu;

template <uint32_t PASS>
__global__ void __launch_bounds__(256) k_batch_hist(BatchArgs a) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t hkey[PASS == 0 ? BIN_HASH : 1], hval[PASS == 0 ? BIN_HASH : 1];
    if (blockIdx.x * BATCH_TILE >= list_len(a, PASS == 0)) return;  // tiles past the list are never read by the scan
    hist[threadIdx.x] = 0u;
    if (PASS == 0)
        for (uint32_t k = threadIdx.x; k < BIN_HASH; k += 256u) {
            hkey[k] = BIN_HASH_EMPTY;
            hval[k] = 0u;
        }
    __syncthreads();
    const uint32_t len = list_len(a, PASS == 0);
    const uint32_t* src = list_src(a, PASS);
    const uint32_t i0 = blockIdx.x * BATCH_TILE;
    constexpr uint32_t PER = BATCH_TILE / 256u;
    uint32_t rows[PER], buckets[PER], metas[PER];
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) {  // independent loads first: this kernel is pure latency
        const uint32_t i = i0 + k * 256u + threadIdx.x;
        rows[k] = i < len ? src[i] : 0u;
    }
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) {
        const uint32_t i = i0 + k * 256u + threadIdx.x;
        buckets[k] = i < len ? a.row_bucket[rows[k]] : BATCH_NO_SET;
        if (buckets[k] >= a.n_buckets) buckets[k] = BATCH_NO_SET;  // stale id after the tables shrank
    }
    if (PASS == 0) {
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            const uint32_t i = i0 + k * 256u + threadIdx.x;
            metas[k] = i < len ? a.row_meta[rows[k]] : 0xFFFFFFFFu;
        }
    }
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) {
        const uint32_t b = buckets[k];
        if (b == BATCH_NO_SET) continue;
        if (PASS == 0) {
            atomicAdd(&hist[b & 255u], 1u);
            // instance_count of a multidrawable row's bin (render_phase/mod.rs:307-311), through the tile's LDS table
            const uint32_t m = metas[k];
            if (b < a.first_set_bucket || m == 0xFFFFFFFFu) continue;
            uint32_t slot = (m * 2654435761u) >> BIN_HASH_SHIFT;
            bool done = false;
            for (uint32_t probe = 0; probe < 8u && !done; ++probe, slot = (slot + 1u) & (BIN_HASH - 1u)) {
                const uint32_t old = atomicCAS(&hkey[slot], BIN_HASH_EMPTY, m);
                if (old == BIN_HASH_EMPTY || old == m) {
                    atomicAdd(&hval[slot], 1u);
                    done = true;
                }
            }
            if (!done) atomicAdd(&a.inst_count[(size_t)m * BATCH_INST_STRIDE], 1u);  // table crowded: straight to memory
        } else {
            atomicAdd(&hist[b >> 8], 1u);
        }
    }
    __syncthreads();
    a.tile_hist[blockIdx.x * 256u + threadIdx.x] = hist[threadIdx.x];  // [tile][digit]: one contiguous KB per tile
    if (PASS == 0)
        for (uint32_t k = threadIdx.x; k < BIN_HASH; k += 256u)
            if (hkey[k] != BIN_HASH_EMPTY) atomicAdd(&a.inst_count[(size_t)hkey[k] * BATCH_INST_STRIDE], hval[k]);
}

// K exclusive scans across the THREADS threads of the workgroup in one go (two barriers): v[] becomes the exclusive prefix,
// total[] the sums.  lds: [THREADS / 64][K].
template <uint32_t K, uint32_t THREADS>
__device__ __forceinline__ void block_scan_multi(uint32_t (&v)[K], uint32_t (*lds)[K], uint32_t (&total)[K]) {
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
    __syncthreads();  // lds may still be read by the previous call
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
template <uint32_t K>
__device__ __forceinline__ void block_scan_1024_multi(uint32_t (&v)[K], uint32_t (*lds)[K], uint32_t (&total)[K]) {
    block_scan_multi<K, 1024>(v, lds, total);
}

// exclusive scan of tile_hist in digit-major order (= the partition offsets); counters[0] = entries kept.  Eight consecutive
// entries per thread, so up to 64 k visible rows are one trip through the loop.  digit_start (LDS, 257 words, may be null)
// receives the offset at which every digit's run starts, [256] = the total.
template <uint32_t THREADS>
__device__ __forceinline__ uint32_t scan_tile_hist(const BatchArgs& a, uint32_t len, uint32_t (*lds)[1], uint32_t* digit_start) {
    constexpr uint32_t PER = 8192u / THREADS;
    const uint32_t used = (len + BATCH_TILE - 1u) / BATCH_TILE;  // tiles beyond hold nothing
    if (THREADS == 256u && used <= PER) {
        // one thread per digit: tile k's 256 counts are one contiguous KB, so every load and store below is a coalesced wave
        // access (a single workgroup keeps only so many cache misses in flight: a digit-major table, 64 lines per wave
        // instruction, made this scan 15 us), and there is no index arithmetic (the general walk divides by `used`)
        uint32_t v[PER], sum[1] = {0};
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) v[k] = k < used ? a.tile_hist[k * 256u + threadIdx.x] : 0u;
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            const uint32_t t = v[k];
            v[k] = sum[0];
            sum[0] += t;
        }
        uint32_t tot[1];
        block_scan_multi<1, THREADS>(sum, lds, tot);
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k)
            if (k < used) a.tile_hist[k * 256u + threadIdx.x] = sum[0] + v[k];
        if (digit_start) {
            digit_start[threadIdx.x] = sum[0];
            if (threadIdx.x == 0) digit_start[256] = tot[0];
        }
        return tot[0];
    }
    uint32_t carry = 0;
    const uint32_t total_entries = 256u * used;  // walk (digit, tile < used) in order
    if (digit_start && used == 0u && threadIdx.x < 257u) digit_start[threadIdx.x] = 0u;
    if (digit_start && used == 0u && THREADS < 257u && threadIdx.x == 0) digit_start[256] = 0u;
    for (uint32_t e0 = 0; e0 < total_entries; e0 += 8192u) {
        uint32_t v[PER], idx[PER], sum[1] = {0};
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            const uint32_t e = e0 + threadIdx.x * PER + k;
            idx[k] = e < total_entries ? (e % used) * 256u + (e / used) : 0xFFFFFFFFu;
            v[k] = e < total_entries ? a.tile_hist[idx[k]] : 0u;
        }
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            const uint32_t t = v[k];
            v[k] = sum[0];
            sum[0] += t;
        }
        uint32_t tot[1];
        block_scan_multi<1, THREADS>(sum, lds, tot);
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k)
            if (idx[k] != 0xFFFFFFFFu) {
                const uint32_t off = carry + sum[0] + v[k];
                a.tile_hist[idx[k]] = off;
                const uint32_t e = e0 + threadIdx.x * PER + k;
                if (digit_start && e % used == 0u) digit_start[e / used] = off;
            }
        carry += tot[0];
    }
    if (digit_start && threadIdx.x == 0) digit_start[256] = carry;
    return carry;
}

template <uint32_t PASS>
__global__ void __launch_bounds__(1024) k_batch_scan(BatchArgs a) {
    __shared__ uint32_t lds_waves[16][1];
    const uint32_t carry = scan_tile_hist<1024>(a, list_len(a, PASS == 0), lds_waves, nullptr);
    if (threadIdx.x == 0 && PASS == 0) a.counters[0] = carry;
}

// The reference's bookkeeping for every bin and batch set of the phase at once.  Per bucket, in bucket order, exclusive scans of
//   q0      MeshUniform slots (data_buffer.add(), shared by both mesh classes)
//   q1, q2  work items per mesh class
//   q3, q4  indirect-parameters slots per class: allocate(entities.len()) for an unbatchable bin, 1 per non-empty batchable bin,
//           the bin count of a non-empty batch set
//   q5, q6  IndirectBatchSets per class: one per unbatchable entity with an input index, one per non-empty batchable bin / batch set
//   q7      records (BinnedRenderPhaseBatchSet / batches)   q8  UnbatchableBinnedEntityIndices
// count[] of a bucket comes from the partition offsets (ONE_PASS: digit starts in LDS) or from the run bounds in the list.
// THREADS: 512 in the many-bucket path (159 registers, no scratch; a 1 024-thread build is held to 128 and spilled 126 of them in the
// middle of the scans -- removed in round 4).
template <bool ONE_PASS, uint32_t THREADS>
__global__ void __launch_bounds__(THREADS) k_batch_plan(BatchArgs a) {
    __shared__ uint32_t lds_scan1[THREADS / 64][1];
    __shared__ uint32_t lds_scan[THREADS / 64][9];
    __shared__ uint32_t digit_start[258];
    if (ONE_PASS) {
        const uint32_t kept = scan_tile_hist<THREADS>(a, *a.list_count, lds_scan1, digit_start);
        if (threadIdx.x == 0) a.counters[0] = kept;
        __syncthreads();
    }
    const bool indirect = a.no_indirect == 0u;
    uint32_t carry[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t b0 = 0; b0 < a.n_buckets; b0 += THREADS) {
        const uint32_t b = b0 + threadIdx.x;
        const bool in = b < a.n_buckets;
        auto count_of = [&](uint32_t k) -> uint32_t {
            if (k >= a.n_buckets) return 0u;
            return ONE_PASS ? digit_start[k + 1u] - digit_start[k] : a.set_count[a.n_buckets + k] - a.set_count[k];
        };
        const uint32_t d = in ? a.bucket_desc[b] : 0u;
        const uint32_t kind = desc_kind(d), cls = desc_cls(d), id = desc_id(d);
        const uint32_t cnt = in ? count_of(b) : 0u;
        const uint32_t sibling = (in && kind == KIND_UNB_INPUT) ? count_of(b + 1u) : 0u;  // the bin's rows without an input index
        const uint32_t bins = (in && kind == KIND_SET && cnt) ? a.meta_offset[id + 1] - a.meta_offset[id] : 0u;
        const uint32_t items = kind == KIND_UNB_NO_INPUT ? 0u : cnt;
        uint32_t ip = 0, bs = 0, rec = 0, unb = 0;
        if (kind == KIND_UNB_INPUT) {
            ip = indirect ? cnt + sibling : 0u;
            bs = indirect ? cnt : 0u;
            unb = cnt;
        } else if (kind == KIND_BATCHABLE) {
            ip = bs = (indirect && cnt) ? 1u : 0u;
            rec = cnt ? 1u : 0u;
        } else if (kind == KIND_SET) {
            ip = bins;
            bs = rec = cnt ? 1u : 0u;
        }
        uint32_t v[9] = {items, cls ? 0u : items, cls ? items : 0u, cls ? 0u : ip, cls ? ip : 0u, cls ? 0u : bs, cls ? bs : 0u, rec, unb};
        uint32_t tot[9];
        block_scan_multi<9, THREADS>(v, lds_scan, tot);
        if (in) {
            // selects, not [cls]: dynamic indexing into the kernarg struct would spill it to scratch
            const uint32_t data0 = a.initial.output_mesh_uniform_index + carry[0] + v[0];
            const uint32_t wi0 = cls ? a.initial.work_item_index[1] + carry[2] + v[2] : a.initial.work_item_index[0] + carry[1] + v[1];
            const uint32_t ip0 = cls ? a.initial.indirect_parameters_index[1] + carry[4] + v[4]
                                     : a.initial.indirect_parameters_index[0] + carry[3] + v[3];
            const uint32_t bs0 = cls ? a.initial.batch_set_index[1] + carry[6] + v[6] : a.initial.batch_set_index[0] + carry[5] + v[5];
            const uint32_t start = ONE_PASS ? digit_start[b] : a.set_count[b];
            a.plan[0u * a.n_buckets + b] = start;
            a.plan[1u * a.n_buckets + b] = data0;
            a.plan[2u * a.n_buckets + b] = wi0;
            // a bin's rows without an input index own the zero tail of the bin's allocate(len): their ip0 is the first tail slot
            a.plan[3u * a.n_buckets + b] = kind == KIND_UNB_NO_INPUT ? ip0 - cnt : ip0;
            a.plan[4u * a.n_buckets + b] = bs0;
            a.plan[5u * a.n_buckets + b] = carry[8] + v[8];
            a.plan[6u * a.n_buckets + b] = cnt;
            if (cnt && (kind == KIND_BATCHABLE || kind == KIND_SET)) {
                if (indirect) {
                    uint32_t* bset = cls ? a.batch_sets[1] : a.batch_sets[0];
                    bset[2u * bs0 + 0u] = 0u;   // indirect_parameters_count
                    bset[2u * bs0 + 1u] = ip0;  // indirect_parameters_base
                }
                if (kind == KIND_BATCHABLE && indirect) {  // write_batch_indirect_parameters_metadata of the bin's one batch
                    uint32_t* md = (cls ? a.metadata[1] : a.metadata[0]) + 5u * ip0;
                    md[0] = data0;
                    md[1] = bs0;
                    md[2] = md[3] = md[4] = 0u;
                }
                uint32_t* rec_out = a.records + 8u * (carry[7] + v[7]);
                const bool set = kind == KIND_SET;
                rec_out[0] = set ? id : (0x80000000u | id);
                rec_out[1] = cls;
                rec_out[2] = indirect ? bs0 : 0u;
                rec_out[3] = set ? wi0 : 0u;  // "Unused" for a batchable bin (gpu_preprocessing.rs:2345-2350)
                rec_out[4] = cnt;
                rec_out[5] = indirect ? ip0 : 0xFFFFFFFFu;
                rec_out[6] = set ? bins : 1u;
                rec_out[7] = data0;
            }
        }
#pragma unroll
        for (uint32_t q = 0; q < 9u; ++q) carry[q] += tot[q];
    }
    if (threadIdx.x == 0) {
        for (uint32_t c = 0; c < 2u; ++c) {
            a.totals[0 + c] = a.initial.work_item_index[c] + carry[1 + c];
            a.totals[2 + c] = a.initial.indirect_parameters_index[c] + carry[3 + c];
            a.totals[4 + c] = a.initial.batch_set_index[c] + carry[5 + c];
        }
        a.totals[6] = a.initial.output_mesh_uniform_index + carry[0];
        a.totals[7] = carry[7];
        a.totals[8] = carry[8];
    }
}

// What one listed row writes once its ordinal j inside its bucket is known.
__device__ __forceinline__ void emit_row(const BatchArgs& a, uint32_t row, uint32_t bucket, uint32_t j) {
    const uint32_t d = a.bucket_desc[bucket];
    const uint32_t kind = desc_kind(d), cls = desc_cls(d);
    const bool indirect = a.no_indirect == 0u;
    const uint32_t ip0 = a.plan[3u * a.n_buckets + bucket];
    if (kind == KIND_UNB_NO_INPUT) {  // get_binned_index() == None: nothing but its zeroed slot of allocate(len) (:2153-2171)
        if (indirect) {
            uint32_t* md = (cls ? a.metadata[1] : a.metadata[0]) + 5u * (ip0 + j);
            md[0] = md[1] = md[2] = md[3] = md[4] = 0u;
        }
        return;
    }
    const uint32_t out = a.plan[1u * a.n_buckets + bucket] + j;
    uint32_t* wi = (cls ? a.work_items[1] : a.work_items[0]) + 2u * (a.plan[2u * a.n_buckets + bucket] + j);
    wi[0] = a.row_input[row];
    if (kind == KIND_UNB_INPUT) {  // :2173-2222
        const uint32_t ipi = ip0 + j;
        wi[1] = indirect ? ipi : out;
        uint32_t* ub = a.unbatchable + 2u * (a.plan[5u * a.n_buckets + bucket] + j);
        ub[0] = desc_id(d);
        ub[1] = indirect ? ipi : out;
        if (indirect) {
            uint32_t* md = (cls ? a.metadata[1] : a.metadata[0]) + 5u * ipi;
            md[0] = out;
            md[1] = 0xFFFFFFFFu;
            md[2] = md[3] = md[4] = 0u;
            uint32_t* bset = (cls ? a.batch_sets[1] : a.batch_sets[0]) + 2u * (a.plan[4u * a.n_buckets + bucket] + j);
            bset[0] = 0u;
            bset[1] = ipi;
        }
    } else if (kind == KIND_BATCHABLE) {  // :2232-2310: the batch's first indirect-parameters slot, or the output index
        wi[1] = indirect ? ip0 : out;
    } else {  // unpack_bins.wesl:64-93
        wi[1] = ip0 + a.bin_meta_in[3u * a.row_meta[row]];
    }
}

// allocate_uniforms.wesl for one batch set per workgroup: base_output_index of bin k (metadata order) = the set's first
// MeshUniform slot + sum of instance_count over bins < k; written at the bin's indirect parameters slot together with
// batch_set_index and zeroed mesh_index / early / late counts (:120-134).  Also publishes the set's instance counts and
// zeroes the counters the NEXT build will add into.
__device__ __forceinline__ void allocate_set(const BatchArgs& a, uint32_t s, uint32_t* lds_waves) {
    const uint32_t bucket = a.first_set_bucket + s;
    const uint32_t cls = a.set_indexed[s] ? 1u : 0u;
    const uint32_t m0 = a.meta_offset[s], bins = a.meta_offset[s + 1] - m0;
    const uint32_t first_ip = a.plan[3u * a.n_buckets + bucket], bsi = a.plan[4u * a.n_buckets + bucket];
    uint32_t carry = a.plan[1u * a.n_buckets + bucket];
    const bool nonempty = a.plan[6u * a.n_buckets + bucket] != 0u;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    for (uint32_t k0 = 0; k0 < bins; k0 += 256u) {
        const uint32_t k = k0 + threadIdx.x;
        const uint32_t v = k < bins ? a.inst_count[(size_t)(m0 + k) * BATCH_INST_STRIDE] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (uint32_t off = 1; off < 64u; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off, 64);
            if (lane >= off) incl += up;
        }
        __syncthreads();
        if (lane == 63u) lds_waves[wv] = incl;
        __syncthreads();
        uint32_t before = 0, all = 0;
#pragma unroll
        for (uint32_t w = 0; w < 4u; ++w) {
            before += w < wv ? lds_waves[w] : 0u;
            all += lds_waves[w];
        }
        if (k < bins) {
            a.bin_metadata_out[3u * (m0 + k) + 0u] = a.bin_meta_in[3u * (m0 + k)];
            a.bin_metadata_out[3u * (m0 + k) + 1u] = a.bin_meta_in[3u * (m0 + k) + 1u];
            a.bin_metadata_out[3u * (m0 + k) + 2u] = v;
            a.inst_count_next[(size_t)(m0 + k) * BATCH_INST_STRIDE] = 0u;
            if (nonempty) {  // a batch set without instances was skipped: it owns no indirect-parameters slots (:2520-2524)
                uint32_t* md = (cls ? a.metadata[1] : a.metadata[0]) + 5u * (first_ip + a.bin_meta_in[3u * (m0 + k)]);
                md[0] = carry + before + incl - v;
                md[1] = bsi;
                md[2] = 0u;
                md[3] = 0u;
                md[4] = 0u;
            }
        }
        carry += all;
    }
}

// stable scatter: items keep their list order inside a digit.  8 rounds of 256 items; in a round, wave w's items precede
// wave w+1's and lane order is item order.  FINAL: the position is not stored -- the row emits its outputs from it.
template <uint32_t PASS, bool FINAL>
__device__ __forceinline__ void scatter_tile(const BatchArgs& a, uint32_t tile, uint32_t* running, uint32_t (*wave_hist)[256]) {
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t len = list_len(a, PASS == 0);
    const uint32_t* src = list_src(a, PASS);
    uint32_t* dst = PASS == 0 ? a.rows_a : a.rows_b;
    running[threadIdx.x] = a.tile_hist[tile * 256u + threadIdx.x];
    const uint32_t i0 = tile * BATCH_TILE;
    if (i0 >= len) return;
    for (uint32_t r = 0; r < BATCH_TILE / 256u; ++r) {
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) wave_hist[k][threadIdx.x] = 0u;
        __syncthreads();
        const uint32_t i = i0 + r * 256u + threadIdx.x;
        uint32_t row = 0, digit = 0, bucket = BATCH_NO_SET;
        bool valid = i < len;
        if (valid) {
            row = src[i];
            bucket = a.row_bucket[row];
            valid = bucket < a.n_buckets;
            digit = PASS == 0 ? (bucket & 255u) : (bucket >> 8);
        }
        // lanes of this wave holding the same digit: eight ballots, one per digit bit (constant time, however many
        // distinct digits the wave holds)
        unsigned long long m = __ballot(valid);
#pragma unroll
        for (uint32_t b = 0; b < 8u; ++b) {
            const bool bit = (digit >> b) & 1u;
            const unsigned long long bb = __ballot(bit);
            m &= bit ? bb : ~bb;
        }
        const unsigned long long lt = (1ull << lane) - 1ull;
        const uint32_t rank = __popcll(m & lt), same = __popcll(m);
        if (valid && rank == 0) wave_hist[wv][digit] = same;
        __syncthreads();
        if (valid) {
            uint32_t pos = running[digit] + rank;
#pragma unroll
            for (uint32_t k = 0; k < 3u; ++k) pos += k < wv ? wave_hist[k][digit] : 0u;
            if (FINAL) emit_row(a, row, bucket, pos - a.plan[bucket]);
            else dst[pos] = row;
        }
        __syncthreads();
        running[threadIdx.x] += wave_hist[0][threadIdx.x] + wave_hist[1][threadIdx.x] + wave_hist[2][threadIdx.x] + wave_hist[3][threadIdx.x];
        __syncthreads();
    }
}

template <uint32_t PASS>
__global__ void __launch_bounds__(256) k_batch_scatter(BatchArgs a) {
    __shared__ uint32_t running[256];
    __shared__ uint32_t wave_hist[4][256];
    scatter_tile<PASS, false>(a, blockIdx.x, running, wave_hist);
}

// ONE_PASS: workgroups [0, n_tiles) scatter-and-emit, the next n_sets run allocate_uniforms.  Otherwise the partitioned list is in
// rows_b: workgroups [0, cap / 256) emit from it.
template <bool ONE_PASS>
__global__ void __launch_bounds__(256) k_batch_emit(BatchArgs a, uint32_t n_emit_blocks) {
    __shared__ uint32_t running[256];
    __shared__ uint32_t wave_hist[4][256];
    if (blockIdx.x >= n_emit_blocks) {
        allocate_set(a, blockIdx.x - n_emit_blocks, running);
        return;
    }
    if (ONE_PASS) {
        scatter_tile<0, true>(a, blockIdx.x, running, wave_hist);
    } else {
        const uint32_t p = blockIdx.x * 256u + threadIdx.x;
        if (p >= a.counters[0]) return;
        const uint32_t row = a.rows_b[p];
        const uint32_t bucket = a.row_bucket[row];
        emit_row(a, row, bucket, p - a.plan[bucket]);
    }
}

// ---------------------------------------------------------------------------------------------
// At most 256 buckets: plan and emit in ONE launch.  Every workgroup derives the whole plan itself, in LDS -- a bucket per
// thread; the tile histograms are [tile][256], so a thread's walk over the tiles is a sequence of coalesced wave loads; the
// nine scans are the ones of k_batch_plan -- and then emits its own tile's rows (or runs allocate_uniforms for one batch
// set).  The redundant part is a ~23 KB read of L2-resident counts and a few hundred instructions per workgroup; what it buys
// is a whole dependent launch (>= 4.3 us of dispatch + the kernel boundary's cache maintenance + the plan kernel's own chain of
// round trips, which a single workgroup cannot overlap).  Workgroup 0 also writes the plan's global outputs.
// A tile's rows and their columns are fetched up front, all four rounds at once: two dependent round trips per tile, not two
// per round.
// ---------------------------------------------------------------------------------------------
struct LdsPlan {
    uint32_t desc[256];
    uint32_t v[7][256];  // as BatchArgs::plan
};
__device__ __forceinline__ void emit_row_lds(const BatchArgs& a, const LdsPlan& p, uint32_t bucket, uint32_t j, uint32_t input,
                                             uint32_t bin_ip_offset) {
    const uint32_t d = p.desc[bucket];
    const uint32_t kind = desc_kind(d), cls = desc_cls(d);
    const bool indirect = a.no_indirect == 0u;
    const uint32_t ip0 = p.v[3][bucket];
    if (kind == KIND_UNB_NO_INPUT) {
        if (indirect) {
            uint32_t* md = (cls ? a.metadata[1] : a.metadata[0]) + 5u * (ip0 + j);
            md[0] = md[1] = md[2] = md[3] = md[4] = 0u;
        }
        return;
    }
    const uint32_t out = p.v[1][bucket] + j;
    uint32_t* wi = (cls ? a.work_items[1] : a.work_items[0]) + 2u * (p.v[2][bucket] + j);
    wi[0] = input;
    if (kind == KIND_UNB_INPUT) {
        const uint32_t ipi = ip0 + j;
        wi[1] = indirect ? ipi : out;
        uint32_t* ub = a.unbatchable + 2u * (p.v[5][bucket] + j);
        ub[0] = desc_id(d);
        ub[1] = indirect ? ipi : out;
        if (indirect) {
            uint32_t* md = (cls ? a.metadata[1] : a.metadata[0]) + 5u * ipi;
            md[0] = out;
            md[1] = 0xFFFFFFFFu;
            md[2] = md[3] = md[4] = 0u;
            uint32_t* bset = (cls ? a.batch_sets[1] : a.batch_sets[0]) + 2u * (p.v[4][bucket] + j);
            bset[0] = 0u;
            bset[1] = ipi;
        }
    } else if (kind == KIND_BATCHABLE) {
        wi[1] = indirect ? ip0 : out;
    } else {
        wi[1] = ip0 + bin_ip_offset;
    }
}

__global__ void __launch_bounds__(256) k_batch_plan_emit(BatchArgs a, uint32_t n_emit_blocks) {
    constexpr uint32_t PER = BATCH_TILE / 256u;  // rows per thread
    constexpr uint32_t TCH = 32;                 // tile histograms fetched per trip
    __shared__ LdsPlan plan;
    __shared__ uint32_t running[256];
    __shared__ uint32_t wave_hist[4][256];
    __shared__ uint32_t lds_scan1[4][1];
    __shared__ uint32_t lds_scan[4][9];
    __shared__ uint32_t bucket_total[257];
    const uint32_t t = threadIdx.x, lane = t & 63u, wv = t >> 6;
    const bool emitter = blockIdx.x < n_emit_blocks;
    const uint32_t tile = emitter ? blockIdx.x : 0xFFFFFFFFu;  // (an allocate workgroup sums no tile prefix)
    const uint32_t len = *a.list_count;
    const uint32_t used = (len + BATCH_TILE - 1u) / BATCH_TILE;
    const uint32_t* src = a.list + (a.list_base ? *a.list_base : 0ull);
    const uint32_t i0 = emitter ? tile * BATCH_TILE : 0u;
    const bool has_rows = emitter && i0 < len;
    // the grid covers the list's capacity (every row of the scene): a tile past the list's end has nothing to emit and nobody
    // needs its copy of the plan (workgroup 0 stays: it writes the plan's outputs even for an empty list)
    if (emitter && !has_rows && blockIdx.x != 0u) return;

    // ---- every independent load first: this tile's rows, the bucket table, the tiles' counts of this thread's bucket ----
    uint32_t rows[PER];
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) {
        const uint32_t i = i0 + k * 256u + t;
        rows[k] = (has_rows && i < len) ? src[i] : 0u;
    }
    const bool in = t < a.n_buckets;
    const uint32_t d = in ? a.bucket_desc[t] : 0u;
    uint32_t total = 0, before = 0;
    for (uint32_t k0 = 0; k0 < used; k0 += TCH) {
        uint32_t h[TCH];
#pragma unroll
        for (uint32_t k = 0; k < TCH; ++k) h[k] = k0 + k < used ? a.tile_hist[(k0 + k) * 256u + t] : 0u;
#pragma unroll
        for (uint32_t k = 0; k < TCH; ++k) {
            total += h[k];
            before += k0 + k < tile ? h[k] : 0u;
        }
    }
    // ---- second trip: what depends on the rows / the descriptor ----
    uint32_t buckets[PER], inputs[PER], metas[PER];
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) {
        const uint32_t i = i0 + k * 256u + t;
        const bool live = has_rows && i < len;
        buckets[k] = live ? a.row_bucket[rows[k]] : BATCH_NO_SET;
        inputs[k] = live ? a.row_input[rows[k]] : 0u;
        metas[k] = live ? a.row_meta[rows[k]] : 0xFFFFFFFFu;
    }
    const uint32_t kind = desc_kind(d), cls = desc_cls(d), id = desc_id(d);
    const uint32_t nbins = (in && kind == KIND_SET) ? a.meta_offset[id + 1] - a.meta_offset[id] : 0u;

    // ---- the plan (see k_batch_plan), a bucket per thread ----
    uint32_t s1[1] = {total}, tot1[1];
    block_scan_multi<1, 256>(s1, lds_scan1, tot1);  // s1[0] = where the bucket's run starts in the partition
    bucket_total[t] = in ? total : 0u;
    if (t == 0) bucket_total[256] = 0u;
    running[t] = s1[0] + before;
    __syncthreads();
    const bool indirect = a.no_indirect == 0u;
    const uint32_t cnt = in ? total : 0u;
    const uint32_t sibling = (in && kind == KIND_UNB_INPUT) ? bucket_total[t + 1u] : 0u;
    const uint32_t bins = cnt ? nbins : 0u;
    const uint32_t items = kind == KIND_UNB_NO_INPUT ? 0u : cnt;
    uint32_t ip = 0, bs = 0, rec = 0, unb = 0;
    if (kind == KIND_UNB_INPUT) {
        ip = indirect ? cnt + sibling : 0u;
        bs = indirect ? cnt : 0u;
        unb = cnt;
    } else if (kind == KIND_BATCHABLE) {
        ip = bs = (indirect && cnt) ? 1u : 0u;
        rec = cnt ? 1u : 0u;
    } else if (kind == KIND_SET) {
        ip = bins;
        bs = rec = cnt ? 1u : 0u;
    }
    uint32_t v[9] = {items, cls ? 0u : items, cls ? items : 0u, cls ? 0u : ip, cls ? ip : 0u, cls ? 0u : bs, cls ? bs : 0u, rec, unb};
    uint32_t tot[9];
    block_scan_multi<9, 256>(v, lds_scan, tot);
    const uint32_t data0 = a.initial.output_mesh_uniform_index + v[0];
    const uint32_t wi0 = cls ? a.initial.work_item_index[1] + v[2] : a.initial.work_item_index[0] + v[1];
    const uint32_t ip0 = cls ? a.initial.indirect_parameters_index[1] + v[4] : a.initial.indirect_parameters_index[0] + v[3];
    const uint32_t bs0 = cls ? a.initial.batch_set_index[1] + v[6] : a.initial.batch_set_index[0] + v[5];
    const uint32_t ip_plan = kind == KIND_UNB_NO_INPUT ? ip0 - cnt : ip0;  // the zero tail of the bin's allocate(len)
    plan.desc[t] = d;
    plan.v[0][t] = s1[0];
    plan.v[1][t] = data0;
    plan.v[2][t] = wi0;
    plan.v[3][t] = ip_plan;
    plan.v[4][t] = bs0;
    plan.v[5][t] = v[8];
    plan.v[6][t] = cnt;
    if (blockIdx.x == 0) {  // the plan's own outputs, once
        if (in) {
            a.plan[0u * a.n_buckets + t] = s1[0];
            a.plan[1u * a.n_buckets + t] = data0;
            a.plan[2u * a.n_buckets + t] = wi0;
            a.plan[3u * a.n_buckets + t] = ip_plan;
            a.plan[4u * a.n_buckets + t] = bs0;
            a.plan[5u * a.n_buckets + t] = v[8];
            a.plan[6u * a.n_buckets + t] = cnt;
            if (cnt && (kind == KIND_BATCHABLE || kind == KIND_SET)) {
                if (indirect) {
                    uint32_t* bset = cls ? a.batch_sets[1] : a.batch_sets[0];
                    bset[2u * bs0 + 0u] = 0u;
                    bset[2u * bs0 + 1u] = ip0;
                }
                if (kind == KIND_BATCHABLE && indirect) {
                    uint32_t* md = (cls ? a.metadata[1] : a.metadata[0]) + 5u * ip0;
                    md[0] = data0;
                    md[1] = bs0;
                    md[2] = md[3] = md[4] = 0u;
                }
                uint32_t* rec_out = a.records + 8u * v[7];
                const bool set = kind == KIND_SET;
                rec_out[0] = set ? id : (0x80000000u | id);
                rec_out[1] = cls;
                rec_out[2] = indirect ? bs0 : 0u;
                rec_out[3] = set ? wi0 : 0u;
                rec_out[4] = cnt;
                rec_out[5] = indirect ? ip0 : 0xFFFFFFFFu;
                rec_out[6] = set ? bins : 1u;
                rec_out[7] = data0;
            }
        }
        if (t == 0) {
            a.counters[0] = tot1[0];
            for (uint32_t c = 0; c < 2u; ++c) {
                a.totals[0 + c] = a.initial.work_item_index[c] + tot[1 + c];
                a.totals[2 + c] = a.initial.indirect_parameters_index[c] + tot[3 + c];
                a.totals[4 + c] = a.initial.batch_set_index[c] + tot[5 + c];
            }
            a.totals[6] = a.initial.output_mesh_uniform_index + tot[0];
            a.totals[7] = tot[7];
            a.totals[8] = tot[8];
        }
    }
    __syncthreads();

    if (!emitter) {  // allocate_uniforms of one batch set, from the plan in LDS
        const uint32_t s = blockIdx.x - n_emit_blocks, bucket = a.first_set_bucket + s;
        const uint32_t scls = a.set_indexed[s] ? 1u : 0u;
        const uint32_t m0 = a.meta_offset[s], sbins = a.meta_offset[s + 1] - m0;
        const uint32_t first_ip = plan.v[3][bucket], bsi = plan.v[4][bucket];
        uint32_t carry = plan.v[1][bucket];
        const bool nonempty = plan.v[6][bucket] != 0u;
        uint32_t* lds_waves = running;  // (free: an allocate workgroup scatters nothing)
        for (uint32_t k0 = 0; k0 < sbins; k0 += 256u) {
            const uint32_t k = k0 + t;
            const uint32_t c = k < sbins ? a.inst_count[(size_t)(m0 + k) * BATCH_INST_STRIDE] : 0u;
            uint32_t incl = c;
#pragma unroll
            for (uint32_t off = 1; off < 64u; off <<= 1) {
                const uint32_t up = __shfl_up(incl, off, 64);
                if (lane >= off) incl += up;
            }
            __syncthreads();
            if (lane == 63u) lds_waves[wv] = incl;
            __syncthreads();
            uint32_t bef = 0, all = 0;
#pragma unroll
            for (uint32_t w = 0; w < 4u; ++w) {
                bef += w < wv ? lds_waves[w] : 0u;
                all += lds_waves[w];
            }
            if (k < sbins) {
                a.bin_metadata_out[3u * (m0 + k) + 0u] = a.bin_meta_in[3u * (m0 + k)];
                a.bin_metadata_out[3u * (m0 + k) + 1u] = a.bin_meta_in[3u * (m0 + k) + 1u];
                a.bin_metadata_out[3u * (m0 + k) + 2u] = c;
                a.inst_count_next[(size_t)(m0 + k) * BATCH_INST_STRIDE] = 0u;
                if (nonempty) {
                    uint32_t* md = (scls ? a.metadata[1] : a.metadata[0]) + 5u * (first_ip + a.bin_meta_in[3u * (m0 + k)]);
                    md[0] = carry + bef + incl - c;
                    md[1] = bsi;
                    md[2] = 0u;
                    md[3] = 0u;
                    md[4] = 0u;
                }
            }
            carry += all;
        }
        return;
    }
    if (!has_rows) return;

    // ---- third trip: a multidrawable row's slot offset inside its set (unpack_bins.wesl:64-93) ----
    uint32_t binoff[PER];
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) {
        if (buckets[k] >= a.n_buckets) buckets[k] = BATCH_NO_SET;  // stale id after the tables shrank
        const bool set_row = buckets[k] != BATCH_NO_SET && buckets[k] >= a.first_set_bucket && metas[k] != 0xFFFFFFFFu;
        binoff[k] = set_row ? a.bin_meta_in[3u * metas[k]] : 0u;
    }
    // ---- the stable scatter of scatter_tile<0, true>, from registers ----
#pragma unroll
    for (uint32_t r = 0; r < PER; ++r) {
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) wave_hist[k][t] = 0u;
        __syncthreads();
        const uint32_t bucket = buckets[r];
        const bool valid = bucket != BATCH_NO_SET;
        const uint32_t digit = bucket & 255u;
        unsigned long long m = __ballot(valid);
#pragma unroll
        for (uint32_t b = 0; b < 8u; ++b) {
            const bool bit = (digit >> b) & 1u;
            const unsigned long long bb = __ballot(bit);
            m &= bit ? bb : ~bb;
        }
        const unsigned long long lt = (1ull << lane) - 1ull;
        const uint32_t rank = __popcll(m & lt), same = __popcll(m);
        if (valid && rank == 0) wave_hist[wv][digit] = same;
        __syncthreads();
        if (valid) {
            uint32_t pos = running[digit] + rank;
#pragma unroll
            for (uint32_t k = 0; k < 3u; ++k) pos += k < wv ? wave_hist[k][digit] : 0u;
            emit_row_lds(a, plan, bucket, pos - plan.v[0][bucket], inputs[r], binoff[r]);
        }
        __syncthreads();
        running[t] += wave_hist[0][t] + wave_hist[1][t] + wave_hist[2][t] + wave_hist[3][t];
    }
}

// first and one-past-last position of every bucket's run in the partitioned list (both stay 0 for a bucket without rows)
__global__ void __launch_bounds__(256) k_batch_bounds(BatchArgs a) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    const uint32_t len = a.counters[0];
    if (p >= len) return;
    const uint32_t b = a.row_bucket[a.rows_b[p]];
    if (p == 0u || a.row_bucket[a.rows_b[p - 1u]] != b) a.set_count[b] = p;
    if (p + 1u == len || a.row_bucket[a.rows_b[p + 1u]] != b) a.set_count[a.n_buckets + b] = p + 1u;
}

__global__ void __launch_bounds__(256) k_batch_clear_bounds(BatchArgs a) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < 2u * a.n_buckets) a.set_count[i] = 0u;
}

// (sorted phases: kernels_sorted.hip)

}  // namespace

hipError_t launch_batch_resolve_rows(uint32_t n, uint32_t n_sets, uint32_t n_unbatchable, uint32_t n_batchable, const uint8_t* row_kind,
                                     const uint32_t* row_cpu_bin, const uint32_t* row_set, const uint32_t* row_bin, const uint32_t* row_input,
                                     const uint32_t* bin_table_offset, const uint32_t* bin_table, const uint32_t* meta_offset,
                                     uint32_t* row_meta, uint32_t* row_bucket, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    MI_LAUNCH(k_batch_resolve_rows, dim3((n + 255u) / 256u), dim3(256), 0, stream, n, n_sets, n_unbatchable, n_batchable, row_kind,
              row_cpu_bin, row_set, row_bin, row_input, bin_table_offset, bin_table, meta_offset, row_meta, row_bucket);
    return hipGetLastError();
}

hipError_t launch_batch_build(const BatchArgs& a_in, hipStream_t stream, void (*mark)(void*, uint32_t), void* mctx) {
#define MARK(id) do { if (mark) mark(mctx, id); } while (0)
    const BatchArgs& a = a_in;
    if (a.n_buckets <= 256u) {
        MARK(K_BATCH_HIST);
        MI_LAUNCH(k_batch_hist<0>, dim3(a.n_tiles), dim3(256), 0, stream, a);
        MARK(K_BATCH_EMIT);
        MI_LAUNCH(k_batch_plan_emit, dim3(a.n_tiles + a.n_sets), dim3(256), 0, stream, a, a.n_tiles);
    } else {
        const uint32_t cap = a.n_tiles * BATCH_TILE;
        MARK(K_BATCH_HIST);
        MI_LAUNCH(k_batch_hist<0>, dim3(a.n_tiles), dim3(256), 0, stream, a);
        MARK(K_BATCH_SCAN);
        MI_LAUNCH(k_batch_scan<0>, dim3(1), dim3(1024), 0, stream, a);
        MARK(K_BATCH_SCATTER);
        MI_LAUNCH(k_batch_scatter<0>, dim3(a.n_tiles), dim3(256), 0, stream, a);
        MARK(K_BATCH_HIST);
        MI_LAUNCH(k_batch_hist<1>, dim3(a.n_tiles), dim3(256), 0, stream, a);
        MARK(K_BATCH_SCAN);
        MI_LAUNCH(k_batch_scan<1>, dim3(1), dim3(1024), 0, stream, a);
        MARK(K_BATCH_SCATTER);
        MI_LAUNCH(k_batch_scatter<1>, dim3(a.n_tiles), dim3(256), 0, stream, a);
        MARK(K_BATCH_BOUNDS);
        MI_LAUNCH(k_batch_clear_bounds, dim3((2u * a.n_buckets + 255u) / 256u), dim3(256), 0, stream, a);
        MI_LAUNCH(k_batch_bounds, dim3(cap / 256u), dim3(256), 0, stream, a);
        MARK(K_BATCH_PLAN);
        MI_LAUNCH((k_batch_plan<false, 512>), dim3(1), dim3(512), 0, stream, a);  // (512 threads: 256 registers each -- the 1 024-thread build spilled 126 of its 128 into scratch in the middle of the scans)
        MARK(K_BATCH_EMIT);
        MI_LAUNCH(k_batch_emit<false>, dim3(cap / 256u + a.n_sets), dim3(256), 0, stream, a, cap / 256u);
    }
    MARK(K_NUM_KERNELS);
#undef MARK
    return hipGetLastError();
}

}  // namespace mi
