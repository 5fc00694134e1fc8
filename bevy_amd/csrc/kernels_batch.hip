// kernels_batch.hip -- batching work-item build on gfx950.
//
// What the reference does for a view's phase (crates/bevy_render/src/batching/gpu_preprocessing.rs:2360-2447, :2497-2580):
// for every multidrawable batch set, reserve a run of PreprocessWorkItems, MeshUniform slots and indirect-parameter slots
// (CPU, O(1) per set), then two compute passes per set: allocate_uniforms.wesl (exclusive prefix of the bins' instance counts
// -> IndirectParametersMetadata.base_output_index) and unpack_bins.wesl (one PreprocessWorkItem per binned mesh instance).
// The bins themselves are CPU hash maps kept in step with VisibleEntities (render_phase/mod.rs:268-400).
//
// Here the bins are derived on the device from the VisibleEntities list the cull pass left in HBM: per row the render
// world uploads (batch set, RenderBinIndex, InputUniformIndex) once; a frame's build is
//   1. k_batch_clear      zero the instance counts and the per-set bounds
//   2. stable partition of the list by batch set (LSD radix on the set id, 8 bits per pass, 1 pass for <= 256 sets):
//      k_batch_hist -> k_batch_scan -> k_batch_scatter; pass 0 also drops the rows that are not multidrawable and
//      counts instances per bin (integer adds: order-independent, exact; pre-aggregated per 2048-row tile in an LDS
//      hash table, because agent-scope atomics on one cache line serialise at ~25 ns each on this part and
//      many_cubes has ONE bin)
//      k_batch_bounds     where each set's run starts and ends in the partitioned list (no atomics)
//   3. k_batch_sets       one workgroup: the O(1)-per-set CPU bookkeeping as exclusive scans over the sets
//   4. k_batch_allocate   allocate_uniforms for every non-empty set (one workgroup per set, 256-bin chunks with carry)
//   5. k_batch_unpack     unpack_bins over the partitioned list
// The partition is stable, so a set's instances keep the list order (ascending Entity) -- the order the oracle uses;
// the reference leaves that order unspecified (unpack_bins.wesl:31-33).
// Everything is u32; HBM traffic is a few words per visible row, the kernels are latency-, not bandwidth-bound.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace mi {
namespace {

__device__ __forceinline__ uint32_t list_len(const BatchArgs& a, bool pass0) { return pass0 ? *a.list_count : a.counters[0]; }
__device__ __forceinline__ const uint32_t* list_src(const BatchArgs& a, uint32_t pass, uint32_t n_pass) {
    if (pass == 0) return a.list + (a.list_base ? *a.list_base : 0ull);
    (void)n_pass;
    return a.rows_a;
}
// final home of the partitioned list: rows_a after one pass, rows_b after two
__device__ __forceinline__ uint32_t* list_dst(const BatchArgs& a, uint32_t pass) { return pass == 0 ? a.rows_a : a.rows_b; }

// row -> index of its bin's GpuBinMetadata in the concatenated array (once per upload, not per frame)
__global__ void __launch_bounds__(256) k_batch_resolve_rows(uint32_t n, uint32_t n_sets, const uint32_t* row_set, const uint32_t* row_bin,
                                                            const uint32_t* bin_table_offset, const uint32_t* bin_table,
                                                            const uint32_t* meta_offset, uint32_t* row_meta) {
    const uint32_t row = blockIdx.x * 256u + threadIdx.x;
    if (row >= n) return;
    const uint32_t s = row_set[row];
    uint32_t m = 0u;
    if (s < n_sets) {
        const uint32_t slots = bin_table_offset[s + 1] - bin_table_offset[s], bins = meta_offset[s + 1] - meta_offset[s];
        const uint32_t b = row_bin[row];
        const uint32_t k = b < slots ? bin_table[bin_table_offset[s] + b] : 0xFFFFFFFFu;
        m = k < bins ? meta_offset[s] + k : 0xFFFFFFFFu;  // a hole or an index out of range: the row is treated as unbatched
    }
    row_meta[row] = m;
}

__global__ void __launch_bounds__(256) k_batch_clear(BatchArgs a) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < a.n_meta) a.bin_metadata[3u * i + 2u] = 0u;
    if (i < 2u * a.n_sets) a.set_count[i] = 0u;  // [0][s] start, [1][s] end of the set's run
    if (i < 4u) a.counters[i] = 0u;
}

constexpr uint32_t BIN_HASH = 512, BIN_HASH_EMPTY = 0xFFFFFFFFu;

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
    const uint32_t* src = list_src(a, PASS, 0);
    const uint32_t i0 = blockIdx.x * BATCH_TILE;
    constexpr uint32_t PER = BATCH_TILE / 256u;
    uint32_t rows[PER], sets[PER], metas[PER];
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) {  // independent loads first: this kernel is pure latency
        const uint32_t i = i0 + k * 256u + threadIdx.x;
        rows[k] = i < len ? src[i] : 0u;
    }
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) {
        const uint32_t i = i0 + k * 256u + threadIdx.x;
        sets[k] = i < len ? a.row_set[rows[k]] : BATCH_NO_SET;
        // NO_BATCH_SET, a stale set id or a RenderBinIndex that names no bin: not multidrawable
        if (PASS == 0 && (sets[k] >= a.n_sets || a.row_meta[rows[k]] == 0xFFFFFFFFu)) sets[k] = BATCH_NO_SET;
    }
    if (PASS == 0) {
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            const uint32_t i = i0 + k * 256u + threadIdx.x;
            metas[k] = i < len ? a.row_meta[rows[k]] : 0u;
        }
    }
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) {
        const uint32_t s = sets[k];
        if (s == BATCH_NO_SET) continue;
        if (PASS == 0) {
            atomicAdd(&hist[s & 255u], 1u);
            // instance_count of the row's bin (render_phase/mod.rs:307-311), through the tile's LDS table
            const uint32_t m = metas[k];
            uint32_t slot = (m * 2654435761u) >> 23;
            bool done = false;
            for (uint32_t probe = 0; probe < 8u && !done; ++probe, slot = (slot + 1u) & (BIN_HASH - 1u)) {
                const uint32_t old = atomicCAS(&hkey[slot], BIN_HASH_EMPTY, m);
                if (old == BIN_HASH_EMPTY || old == m) {
                    atomicAdd(&hval[slot], 1u);
                    done = true;
                }
            }
            if (!done) atomicAdd(&a.bin_metadata[3u * m + 2u], 1u);  // table crowded: straight to memory
        } else {
            atomicAdd(&hist[s >> 8], 1u);
        }
    }
    __syncthreads();
    a.tile_hist[threadIdx.x * a.n_tiles + blockIdx.x] = hist[threadIdx.x];
    if (PASS == 0)
        for (uint32_t k = threadIdx.x; k < BIN_HASH; k += 256u)
            if (hkey[k] != BIN_HASH_EMPTY) atomicAdd(&a.bin_metadata[3u * hkey[k] + 2u], hval[k]);
}

// workgroup-wide exclusive scan of one value per thread (1024 threads); returns the exclusive prefix, *total = sum
__device__ __forceinline__ uint32_t block_scan_1024(uint32_t v, uint32_t* lds_waves, uint32_t* total) {
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (uint32_t off = 1; off < 64u; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off, 64);
        if (lane >= off) incl += up;
    }
    __syncthreads();  // lds_waves may still be read by the previous call
    if (lane == 63u) lds_waves[wv] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (uint32_t k = 0; k < 16u; ++k) {
        const uint32_t w = lds_waves[k];
        before += k < wv ? w : 0u;
        all += w;
    }
    *total = all;
    return before + incl - v;
}

// exclusive scan of tile_hist in digit-major order (= the partition offsets); counters[0] = entries kept.
// Eight consecutive entries per thread, so up to 64 k visible rows are one trip through the loop.
template <uint32_t PASS>
__global__ void __launch_bounds__(1024) k_batch_scan(BatchArgs a) {
    __shared__ uint32_t lds_waves[16];
    const uint32_t len = list_len(a, PASS == 0);
    const uint32_t used = (len + BATCH_TILE - 1u) / BATCH_TILE;  // tiles beyond hold zeros
    uint32_t carry = 0;
    const uint32_t total_entries = 256u * used;  // walk (digit, tile < used) in order
    for (uint32_t e0 = 0; e0 < total_entries; e0 += 8192u) {
        uint32_t v[8], idx[8], sum = 0;
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) {
            const uint32_t e = e0 + threadIdx.x * 8u + k;
            idx[k] = e < total_entries ? (e / used) * a.n_tiles + (e % used) : 0xFFFFFFFFu;
            v[k] = e < total_entries ? a.tile_hist[idx[k]] : 0u;
        }
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) {
            const uint32_t t = v[k];
            v[k] = sum;
            sum += t;
        }
        uint32_t tot;
        const uint32_t ex = block_scan_1024(sum, lds_waves, &tot);
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k)
            if (idx[k] != 0xFFFFFFFFu) a.tile_hist[idx[k]] = carry + ex + v[k];
        carry += tot;
    }
    if (threadIdx.x == 0 && PASS == 0) a.counters[0] = carry;
}

// stable scatter: items keep their list order inside a digit.  8 rounds of 256 items; in a round, wave w's items precede
// wave w+1's and lane order is item order.
template <uint32_t PASS>
__global__ void __launch_bounds__(256) k_batch_scatter(BatchArgs a) {
    __shared__ uint32_t running[256];
    __shared__ uint32_t wave_hist[4][256];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t len = list_len(a, PASS == 0);
    const uint32_t* src = list_src(a, PASS, 0);
    uint32_t* dst = list_dst(a, PASS);
    running[threadIdx.x] = a.tile_hist[threadIdx.x * a.n_tiles + blockIdx.x];
    const uint32_t i0 = blockIdx.x * BATCH_TILE;
    if (i0 >= len) return;
    for (uint32_t r = 0; r < BATCH_TILE / 256u; ++r) {
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) wave_hist[k][threadIdx.x] = 0u;
        __syncthreads();
        const uint32_t i = i0 + r * 256u + threadIdx.x;
        uint32_t row = 0, digit = 0;
        bool valid = i < len;
        if (valid) {
            row = src[i];
            const uint32_t s = a.row_set[row];
            if (PASS == 0) {
                valid = s < a.n_sets && a.row_meta[row] != 0xFFFFFFFFu;
                digit = s & 255u;
            } else {
                digit = s >> 8;
            }
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
            dst[pos] = row;
        }
        __syncthreads();
        running[threadIdx.x] += wave_hist[0][threadIdx.x] + wave_hist[1][threadIdx.x] + wave_hist[2][threadIdx.x] + wave_hist[3][threadIdx.x];
        __syncthreads();
    }
}

// first and one-past-last position of every set's run in the partitioned list (both stay 0 for a set without instances)
template <bool TWO_PASS>
__global__ void __launch_bounds__(256) k_batch_bounds(BatchArgs a) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    const uint32_t len = a.counters[0];
    if (p >= len) return;
    const uint32_t* rows = TWO_PASS ? a.rows_b : a.rows_a;
    const uint32_t s = a.row_set[rows[p]];
    if (p == 0u || a.row_set[rows[p - 1u]] != s) a.set_count[s] = p;
    if (p + 1u == len || a.row_set[rows[p + 1u]] != s) a.set_count[a.n_sets + s] = p + 1u;
}

// The per-set bookkeeping of prepare_multidrawable_binned_batch_set (gpu_preprocessing.rs:2511-2579) for all sets at
// once: exclusive scans, in set order, of (instances), (instances | class), (bins of non-empty sets | class),
// (non-empty | class); IndirectBatchSet entries, the BinnedRenderPhaseBatchSet records and the buffer lengths.
// K exclusive scans over the 256 threads in one go (two barriers): v[] becomes the exclusive prefix, total[] the sums
template <uint32_t K>
__device__ __forceinline__ void block_scan_256_multi(uint32_t (&v)[K], uint32_t (*lds)[K], uint32_t (&total)[K]) {
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
    __syncthreads();
    if (lane == 63u)
#pragma unroll
        for (uint32_t q = 0; q < K; ++q) lds[wv][q] = incl[q];
    __syncthreads();
#pragma unroll
    for (uint32_t q = 0; q < K; ++q) {
        uint32_t before = 0, all = 0;
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
            const uint32_t w = lds[k][q];
            before += k < wv ? w : 0u;
            all += w;
        }
        total[q] = all;
        v[q] = before + incl[q] - v[q];
    }
}

__global__ void __launch_bounds__(256) k_batch_sets(BatchArgs a) {
    __shared__ uint32_t lds_waves[4][8];
    uint32_t c_all = 0, c_items[2] = {0, 0}, c_bins[2] = {0, 0}, c_sets[2] = {0, 0}, c_rec = 0;
    for (uint32_t s0 = 0; s0 < a.n_sets; s0 += 256u) {
        const uint32_t s = s0 + threadIdx.x;
        const bool in = s < a.n_sets;
        const uint32_t cnt = in ? a.set_count[a.n_sets + s] - a.set_count[s] : 0u;
        const uint32_t cls = in && a.set_indexed[s] ? 1u : 0u;
        const uint32_t bins = in && cnt ? a.meta_offset[s + 1] - a.meta_offset[s] : 0u;
        uint32_t v[8] = {cnt, cls == 0 ? cnt : 0u, cls == 1 ? cnt : 0u, cls == 0 ? bins : 0u, cls == 1 ? bins : 0u,
                         (cls == 0 && cnt) ? 1u : 0u, (cls == 1 && cnt) ? 1u : 0u, cnt ? 1u : 0u};
        uint32_t tot[8];
        block_scan_256_multi<8>(v, lds_waves, tot);
        const uint32_t e_all = v[0], e_i0 = v[1], e_i1 = v[2], e_b0 = v[3], e_b1 = v[4], e_s0 = v[5], e_s1 = v[6], e_rec = v[7];
        const uint32_t t_all = tot[0], t_rec = tot[7];
        const uint32_t t_i[2] = {tot[1], tot[2]}, t_b[2] = {tot[3], tot[4]}, t_s[2] = {tot[5], tot[6]};
        if (in) {
            const uint32_t start = c_all + e_all;
            // selects, not [cls]: dynamic indexing into the kernarg struct would spill it to scratch
            const uint32_t first_wi = cls ? a.initial.work_item_index[1] + c_items[1] + e_i1 : a.initial.work_item_index[0] + c_items[0] + e_i0;
            const uint32_t first_ip = cls ? a.initial.indirect_parameters_index[1] + c_bins[1] + e_b1
                                          : a.initial.indirect_parameters_index[0] + c_bins[0] + e_b0;
            const uint32_t bsi = cls ? a.initial.batch_set_index[1] + c_sets[1] + e_s1 : a.initial.batch_set_index[0] + c_sets[0] + e_s0;
            const uint32_t first_out = a.initial.output_mesh_uniform_index + start;
            a.set_scan[0u * a.n_sets + s] = start;
            a.set_scan[1u * a.n_sets + s] = first_wi;
            a.set_scan[2u * a.n_sets + s] = first_ip;
            a.set_scan[3u * a.n_sets + s] = bsi;
            a.set_scan[4u * a.n_sets + s] = first_out;
            if (cnt) {
                uint32_t* bset = cls ? a.batch_sets[1] : a.batch_sets[0];
                bset[2u * bsi + 0u] = 0u;        // indirect_parameters_count
                bset[2u * bsi + 1u] = first_ip;  // indirect_parameters_base
                uint32_t* rec = a.records + 8u * (c_rec + e_rec);
                rec[0] = s;
                rec[1] = cls;
                rec[2] = bsi;
                rec[3] = first_wi;
                rec[4] = cnt;
                rec[5] = first_ip;
                rec[6] = bins;
                rec[7] = first_out;
            }
        }
        c_all += t_all;
        c_rec += t_rec;
#pragma unroll
        for (uint32_t c = 0; c < 2u; ++c) {
            c_items[c] += t_i[c];
            c_bins[c] += t_b[c];
            c_sets[c] += t_s[c];
        }
    }
    if (threadIdx.x == 0) {
        for (uint32_t c = 0; c < 2u; ++c) {
            a.totals[0 + c] = a.initial.work_item_index[c] + c_items[c];
            a.totals[2 + c] = a.initial.indirect_parameters_index[c] + c_bins[c];
            a.totals[4 + c] = a.initial.batch_set_index[c] + c_sets[c];
        }
        a.totals[6] = a.initial.output_mesh_uniform_index + c_all;
        a.totals[7] = c_rec;
    }
}

// allocate_uniforms.wesl for one batch set per workgroup: base_output_index of bin k (metadata order) =
// first_output_mesh_uniform_index + sum of instance_count over bins < k; written at the bin's indirect parameters slot
// together with batch_set_index and zeroed mesh_index / early / late counts (:120-134).
__global__ void __launch_bounds__(256) k_batch_allocate(BatchArgs a) {
    __shared__ uint32_t lds_waves[4];
    const uint32_t s = blockIdx.x;
    if (a.set_count[a.n_sets + s] == 0u) return;  // no run in the partitioned list
    const uint32_t cls = a.set_indexed[s] ? 1u : 0u;
    const uint32_t m0 = a.meta_offset[s], bins = a.meta_offset[s + 1] - m0;
    const uint32_t first_ip = a.set_scan[2u * a.n_sets + s], bsi = a.set_scan[3u * a.n_sets + s];
    uint32_t carry = a.set_scan[4u * a.n_sets + s];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    for (uint32_t k0 = 0; k0 < bins; k0 += 256u) {
        const uint32_t k = k0 + threadIdx.x;
        const uint32_t v = k < bins ? a.bin_metadata[3u * (m0 + k) + 2u] : 0u;
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
            uint32_t* md = (cls ? a.metadata[1] : a.metadata[0]) + 5u * (first_ip + a.bin_metadata[3u * (m0 + k)]);
            md[0] = carry + before + incl - v;
            md[1] = bsi;
            md[2] = 0u;
            md[3] = 0u;
            md[4] = 0u;
        }
        carry += all;
    }
}

// unpack_bins.wesl:64-93 over the partitioned list: entry p of set s is the set's binned mesh instance p - start(s)
template <bool TWO_PASS>
__global__ void __launch_bounds__(256) k_batch_unpack(BatchArgs a) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= a.counters[0]) return;
    const uint32_t row = (TWO_PASS ? a.rows_b : a.rows_a)[p];
    const uint32_t s = a.row_set[row];
    const uint32_t cls = a.set_indexed[s] ? 1u : 0u;
    const uint32_t global_id = p - a.set_scan[s];
    const uint32_t m = a.row_meta[row];
    uint32_t* wi = (cls ? a.work_items[1] : a.work_items[0]) + 2u * (a.set_scan[1u * a.n_sets + s] + global_id);
    wi[0] = a.row_input[row];
    wi[1] = a.set_scan[2u * a.n_sets + s] + a.bin_metadata[3u * m];
}

}  // namespace

hipError_t launch_batch_resolve_rows(uint32_t n, uint32_t n_sets, const uint32_t* row_set, const uint32_t* row_bin,
                                     const uint32_t* bin_table_offset, const uint32_t* bin_table, const uint32_t* meta_offset,
                                     uint32_t* row_meta, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    MI_LAUNCH(k_batch_resolve_rows, dim3((n + 255u) / 256u), dim3(256), 0, stream, n, n_sets, row_set, row_bin, bin_table_offset,
              bin_table, meta_offset, row_meta);
    return hipGetLastError();
}

hipError_t launch_batch_build(const BatchArgs& a, hipStream_t stream, void (*mark)(void*, uint32_t), void* mctx) {
#define MARK(id) do { if (mark) mark(mctx, id); } while (0)
    const uint32_t clear_n = a.n_meta > 2u * a.n_sets ? a.n_meta : 2u * a.n_sets;
    MARK(K_BATCH_CLEAR);
    MI_LAUNCH(k_batch_clear, dim3((clear_n + 255u) / 256u + 1u), dim3(256), 0, stream, a);
    const bool two = a.n_sets > 256u;
    const uint32_t cap = a.n_tiles * BATCH_TILE;
    MARK(K_BATCH_HIST);
    MI_LAUNCH(k_batch_hist<0>, dim3(a.n_tiles), dim3(256), 0, stream, a);
    MARK(K_BATCH_SCAN);
    MI_LAUNCH(k_batch_scan<0>, dim3(1), dim3(1024), 0, stream, a);
    MARK(K_BATCH_SCATTER);
    MI_LAUNCH(k_batch_scatter<0>, dim3(a.n_tiles), dim3(256), 0, stream, a);
    if (two) {
        MARK(K_BATCH_HIST);
        MI_LAUNCH(k_batch_hist<1>, dim3(a.n_tiles), dim3(256), 0, stream, a);
        MARK(K_BATCH_SCAN);
        MI_LAUNCH(k_batch_scan<1>, dim3(1), dim3(1024), 0, stream, a);
        MARK(K_BATCH_SCATTER);
        MI_LAUNCH(k_batch_scatter<1>, dim3(a.n_tiles), dim3(256), 0, stream, a);
    }
    MARK(K_BATCH_BOUNDS);
    if (two) MI_LAUNCH(k_batch_bounds<true>, dim3(cap / 256u), dim3(256), 0, stream, a);
    else MI_LAUNCH(k_batch_bounds<false>, dim3(cap / 256u), dim3(256), 0, stream, a);
    MARK(K_BATCH_SETS);
    MI_LAUNCH(k_batch_sets, dim3(1), dim3(256), 0, stream, a);
    if (a.n_sets) {
        MARK(K_BATCH_ALLOCATE);
        MI_LAUNCH(k_batch_allocate, dim3(a.n_sets), dim3(256), 0, stream, a);
    }
    MARK(K_BATCH_UNPACK);
    if (two) MI_LAUNCH(k_batch_unpack<true>, dim3(cap / 256u), dim3(256), 0, stream, a);
    else MI_LAUNCH(k_batch_unpack<false>, dim3(cap / 256u), dim3(256), 0, stream, a);
    MARK(K_NUM_KERNELS);
#undef MARK
    return hipGetLastError();
}

}  // namespace mi
