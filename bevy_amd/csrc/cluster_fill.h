// cluster_fill.h -- the second half of the light-cluster assignment (see kernels_cluster.hip): CSR offsets from the cluster totals,
// then every (cluster, block, 256-bit mask) pair expanded into its cluster's list in ascending object order.  A device function,
// because two kernels run it: k_cluster_fill on its own, and the frame kernel (kernels_flat.hip), where the fill of frame f
// rides in extra workgroups at the head of frame f + 1's launch when the caller promised that frame (MI_CULL_MORE_FRAMES) --
// one launch less per frame, exactly like the deferred VisibleEntities compaction.
#pragma once
#include "kernels.h"

namespace mi {

// One workgroup (256 threads) of a fill spread over gx workgroups.
__device__ __forceinline__ void cluster_fill_block(const ClusterWork& w, uint32_t C, uint32_t n_objects, uint32_t bx, uint32_t gx,
                                                   uint32_t* offs /* [4096] LDS */, uint32_t* part /* [4] LDS */) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const uint32_t n_pairs = n_objects ? *w.pair_total : 0u;

    // zero the other parity's buffers for the next frame (this frame's buffers are only read from here on)
    {
        for (size_t i = bx * 256u + tid; i < w.acc_words; i += (size_t)gx * 256u) w.acc_next[i] = 0u;
        uint4* m4 = reinterpret_cast<uint4*>(w.block_counts_next);
        const size_t n4 = (size_t)C * w.row_stride / 8u;
        for (size_t i = bx * 256u + tid; i < n4; i += (size_t)gx * 256u) m4[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    const uint32_t first_wave = bx * 4u;
    if (first_wave >= n_pairs && bx != 0) return;

    // CSR offsets = exclusive prefix of the cluster totals (C <= 4096: 16 per thread, wave scan, 4 wave totals)
    uint32_t loc[16];
    uint32_t sum = 0;
    {
        const uint4* t4 = reinterpret_cast<const uint4*>(w.totals) + tid * 4u;
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            const uint32_t c0 = tid * 16u + q * 4u;
            uint4 v4 = c0 + 3u < C ? t4[q] : make_uint4(c0 < C ? w.totals[c0] : 0u, c0 + 1u < C ? w.totals[c0 + 1u] : 0u,
                                                       c0 + 2u < C ? w.totals[c0 + 2u] : 0u, 0u);
            loc[q * 4u] = sum; sum += v4.x;
            loc[q * 4u + 1u] = sum; sum += v4.y;
            loc[q * 4u + 2u] = sum; sum += v4.z;
            loc[q * 4u + 3u] = sum; sum += v4.w;
        }
    }
    uint32_t incl = sum;
#pragma unroll
    for (uint32_t off = 1; off < 64u; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off, 64);
        if (lane >= off) incl += up;
    }
    if (lane == 63u) part[wv] = incl;
    MI_WG_LDS_BARRIER();
    uint32_t before = incl - sum;
#pragma unroll
    for (uint32_t k = 0; k < 4u; ++k) before += k < wv ? part[k] : 0u;
    const uint32_t grand = part[0] + part[1] + part[2] + part[3];
#pragma unroll
    for (uint32_t k = 0; k < 16; ++k) offs[tid * 16u + k] = before + loc[k];
    MI_WG_LDS_BARRIER();
    if (bx == 0) {
        for (uint32_t c = tid; c < C; c += 256u) w.offsets[c] = offs[c];
        if (tid == 0) {
            w.offsets[C] = grand;
            *w.total = (uint64_t)grand;
        }
    }

    const unsigned long long lt = (1ull << lane) - 1ull;
    for (uint32_t p = first_wave + wv; p < n_pairs; p += gx * 4u) {
        const uint32_t cb = w.pair_cb[p];
        const uint32_t c = cb & 4095u, b = cb >> 12;
        // entries of lower-numbered blocks in this cluster: a contiguous u16 row prefix
        // (rows are padded to 8 entries: one 16-byte load covers 8 blocks, 512 blocks per wave pass)
        const uint4* row4 = reinterpret_cast<const uint4*>(w.block_counts + (size_t)c * w.row_stride);
        uint32_t lower = 0;
        for (uint32_t i = lane; i * 8u < b; i += 64u) {
            uint4 q = row4[i];
            const uint32_t keep = b - i * 8u;  // entries of this vector that belong to blocks < b (>= 1)
            if (keep < 8u) {
                uint32_t wds[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (uint32_t k = 0; k < 4u; ++k) {
                    if (2u * k >= keep) wds[k] = 0u;
                    else if (2u * k + 1u >= keep) wds[k] &= 0xFFFFu;
                }
                q = make_uint4(wds[0], wds[1], wds[2], wds[3]);
            }
            lower += (q.x & 0xFFFFu) + (q.x >> 16) + (q.y & 0xFFFFu) + (q.y >> 16) + (q.z & 0xFFFFu) + (q.z >> 16) +
                     (q.w & 0xFFFFu) + (q.w >> 16);
        }
#pragma unroll
        for (uint32_t off = 32u; off; off >>= 1) lower += __shfl_xor(lower, off, 64);
        const uint32_t mw = lane < 8u ? w.pair_mask[(size_t)p * 8u + lane] : 0u;
        uint64_t dst = (uint64_t)offs[c] + lower;
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) {
            const unsigned long long m64 = (unsigned long long)__shfl(mw, (int)(2u * j), 64) |
                                           ((unsigned long long)__shfl(mw, (int)(2u * j + 1u), 64) << 32);
            if ((m64 >> lane) & 1ull) {
                const uint64_t d = dst + __popcll(m64 & lt);
                if (d < w.capacity) w.indices[d] = (uint32_t)((int32_t)(b * CLUSTER_BLOCK + j * 64u + lane) + w.obj_delta);
            }
            dst += __popcll(m64);
        }
    }
}


}  // namespace mi
