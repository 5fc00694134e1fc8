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
//   1. k_batch_clear      zero the instance counts and the per-set counters
//   2. stable partition of the list by batch set (LSD radix on the set id, 8 bits per pass, 1 pass for <= 256 sets):
//      k_batch_hist -> k_batch_scan -> k_batch_scatter; pass 0 also drops the rows that are not multidrawable and
//      counts instances per bin and per set (integer atomics: order-independent, exact)
//   3. k_batch_sets       one workgroup: the O(1)-per-set CPU bookkeeping as five exclusive scans over the sets
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

__global__ void __launch_bounds__(256) k_batch_clear(BatchArgs a) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < a.n_meta) a.bin_metadata[3u * i + 2u] = 0u;
    if (i < a.n_sets) a.set_count[i] = 0u;
    if (i < 4u) a.counters[i] = 0u;
}

template <uint32_t PASS>
__global__ void __launch_bounds__(256) k_batch_hist(BatchArgs a) {
    __shared__ uint32_t hist[256];
    hist[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t len = list_len(a, PASS == 0);
    const uint32_t* src = list_src(a, PASS, 0);
    const uint32_t i0 = blockIdx.x * BATCH_TILE;
    for (uint32_t j = threadIdx.x; j < BATCH_TILE; j += 256u) {
        const uint32_t i = i0 + j;
        if (i >= len) break;
        const uint32_t row = src[i];
        const uint32_t s = a.row_set[row];
        if (PASS == 0) {
            if (s >= a.n_sets) continue;  // NO_BATCH_SET (or a stale id): not multidrawable
            atomicAdd(&a.set_count[s], 1u);
            const uint32_t m = a.meta_offset[s] + a.bin_table[a.bin_table_offset[s] + a.row_bin[row]];
            atomicAdd(&a.bin_metadata[3u * m + 2u], 1u);
            atomicAdd(&hist[s & 255u], 1u);
        } else {
            atomicAdd(&hist[s >> 8], 1u);
        }
    }
    __syncthreads();
    a.tile_hist[threadIdx.x * a.n_tiles + blockIdx.x] = hist[threadIdx.x];
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

// exclusive scan of tile_hist in digit-major order (= the partition offsets); counters[0] = entries kept
template <uint32_t PASS>
__global__ void __launch_bounds__(1024) k_batch_scan(BatchArgs a) {
    __shared__ uint32_t lds_waves[16];
    const uint32_t len = list_len(a, PASS == 0);
    const uint32_t used = (len + BATCH_TILE - 1u) / BATCH_TILE;  // tiles beyond hold zeros
    uint32_t carry = 0;
    // only the used tiles of every digit row matter; walk (digit, tile < used) in order
    const uint32_t total_entries = 256u * used;
    for (uint32_t e0 = 0; e0 < total_entries; e0 += 1024u) {
        const uint32_t e = e0 + threadIdx.x;
        uint32_t v = 0, idx = 0;
        if (e < total_entries) {
            idx = (e / used) * a.n_tiles + (e % used);
            v = a.tile_hist[idx];
        }
        uint32_t tot;
        const uint32_t ex = block_scan_1024(v, lds_waves, &tot);
        if (e < total_entries) a.tile_hist[idx] = carry + ex;
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
                valid = s < a.n_sets;
                digit = s & 255u;
            } else {
                digit = s >> 8;
            }
        }
        // rank among the wave's earlier lanes with the same digit
        uint32_t rank = 0, same = 0;
        unsigned long long remaining = __ballot(valid);
        const unsigned long long lt = (1ull << lane) - 1ull;
        while (remaining) {
            const int leader = __ffsll((long long)remaining) - 1;
            const uint32_t d = __shfl(digit, leader, 64);
            const unsigned long long m = __ballot(valid && digit == d);
            if (valid && digit == d) {
                rank = __popcll(m & lt);
                same = __popcll(m);
            }
            remaining &= ~m;
        }
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

// The per-set bookkeeping of prepare_multidrawable_binned_batch_set (gpu_preprocessing.rs:2511-2579) for all sets at
// once: exclusive scans, in set order, of (instances), (instances | class), (bins of non-empty sets | class),
// (non-empty | class); IndirectBatchSet entries, the BinnedRenderPhaseBatchSet records and the buffer lengths.
__global__ void __launch_bounds__(1024) k_batch_sets(BatchArgs a) {
    __shared__ uint32_t lds_waves[16];
    uint32_t c_all = 0, c_items[2] = {0, 0}, c_bins[2] = {0, 0}, c_sets[2] = {0, 0}, c_rec = 0;
    for (uint32_t s0 = 0; s0 < a.n_sets; s0 += 1024u) {
        const uint32_t s = s0 + threadIdx.x;
        const bool in = s < a.n_sets;
        const uint32_t cnt = in ? a.set_count[s] : 0u;
        const uint32_t cls = in && a.set_indexed[s] ? 1u : 0u;
        const uint32_t bins = in && cnt ? a.meta_offset[s + 1] - a.meta_offset[s] : 0u;
        uint32_t t_all, t_i[2], t_b[2], t_s[2], t_rec;
        const uint32_t e_all = block_scan_1024(cnt, lds_waves, &t_all);
        const uint32_t e_i0 = block_scan_1024(cls == 0 ? cnt : 0u, lds_waves, &t_i[0]);
        const uint32_t e_i1 = block_scan_1024(cls == 1 ? cnt : 0u, lds_waves, &t_i[1]);
        const uint32_t e_b0 = block_scan_1024(cls == 0 ? bins : 0u, lds_waves, &t_b[0]);
        const uint32_t e_b1 = block_scan_1024(cls == 1 ? bins : 0u, lds_waves, &t_b[1]);
        const uint32_t e_s0 = block_scan_1024(cls == 0 && cnt ? 1u : 0u, lds_waves, &t_s[0]);
        const uint32_t e_s1 = block_scan_1024(cls == 1 && cnt ? 1u : 0u, lds_waves, &t_s[1]);
        const uint32_t e_rec = block_scan_1024(cnt ? 1u : 0u, lds_waves, &t_rec);
        if (in) {
            const uint32_t start = c_all + e_all;
            const uint32_t first_wi = a.initial.work_item_index[cls] + c_items[cls] + (cls ? e_i1 : e_i0);
            const uint32_t first_ip = a.initial.indirect_parameters_index[cls] + c_bins[cls] + (cls ? e_b1 : e_b0);
            const uint32_t bsi = a.initial.batch_set_index[cls] + c_sets[cls] + (cls ? e_s1 : e_s0);
            const uint32_t first_out = a.initial.output_mesh_uniform_index + start;
            a.set_scan[0u * a.n_sets + s] = start;
            a.set_scan[1u * a.n_sets + s] = first_wi;
            a.set_scan[2u * a.n_sets + s] = first_ip;
            a.set_scan[3u * a.n_sets + s] = bsi;
            a.set_scan[4u * a.n_sets + s] = first_out;
            if (cnt) {
                a.batch_sets[cls][2u * bsi + 0u] = 0u;        // indirect_parameters_count
                a.batch_sets[cls][2u * bsi + 1u] = first_ip;  // indirect_parameters_base
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
    if (a.set_count[s] == 0u) return;
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
            uint32_t* md = a.metadata[cls] + 5u * (first_ip + a.bin_metadata[3u * (m0 + k)]);
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
    const uint32_t m = a.meta_offset[s] + a.bin_table[a.bin_table_offset[s] + a.row_bin[row]];
    uint32_t* wi = a.work_items[cls] + 2u * (a.set_scan[1u * a.n_sets + s] + global_id);
    wi[0] = a.row_input[row];
    wi[1] = a.set_scan[2u * a.n_sets + s] + a.bin_metadata[3u * m];
}

}  // namespace

hipError_t launch_batch_build(const BatchArgs& a, hipStream_t stream) {
    const uint32_t clear_n = a.n_meta > a.n_sets ? a.n_meta : a.n_sets;
    MI_LAUNCH(k_batch_clear, dim3((clear_n + 255u) / 256u + 1u), dim3(256), 0, stream, a);
    const bool two = a.n_sets > 256u;
    MI_LAUNCH(k_batch_hist<0>, dim3(a.n_tiles), dim3(256), 0, stream, a);
    MI_LAUNCH(k_batch_scan<0>, dim3(1), dim3(1024), 0, stream, a);
    MI_LAUNCH(k_batch_scatter<0>, dim3(a.n_tiles), dim3(256), 0, stream, a);
    if (two) {
        MI_LAUNCH(k_batch_hist<1>, dim3(a.n_tiles), dim3(256), 0, stream, a);
        MI_LAUNCH(k_batch_scan<1>, dim3(1), dim3(1024), 0, stream, a);
        MI_LAUNCH(k_batch_scatter<1>, dim3(a.n_tiles), dim3(256), 0, stream, a);
    }
    MI_LAUNCH(k_batch_sets, dim3(1), dim3(1024), 0, stream, a);
    if (a.n_sets) MI_LAUNCH(k_batch_allocate, dim3(a.n_sets), dim3(256), 0, stream, a);
    const uint32_t cap = a.n_tiles * BATCH_TILE;
    if (two) MI_LAUNCH(k_batch_unpack<true>, dim3(cap / 256u), dim3(256), 0, stream, a);
    else MI_LAUNCH(k_batch_unpack<false>, dim3(cap / 256u), dim3(256), 0, stream, a);
    return hipGetLastError();
}

}  // namespace mi
