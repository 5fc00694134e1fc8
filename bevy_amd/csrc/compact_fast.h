// compact_fast.h -- one workgroup of the single-launch VisibleEntities compaction, as device code: k_compact_fast runs it on its own,
// the frame kernels (kernels_flat.hip) and the fused hierarchy frame (kernels_tree.hip) carry the previous frame's in extra
// workgroups of their own launch (MI_CULL_MORE_FRAMES).
#pragma once
#include "kernels.h"

namespace mi {

__device__ __forceinline__ uint32_t sum_bytes(uint32_t x, uint32_t acc) { return __builtin_amdgcn_sad_u8(x, 0u, acc); }

// One workgroup of the single-launch compaction: block (bx of gx, segment by).  Called from k_compact_fast and from the
// tail workgroups of the frame kernels (deferred compaction of the previous frame).
// (the shape -- steps, chunks, hierarchical mode -- is described in kernels.h next to compact_fast_gx)
// HIER_OK = false: a rider of a frame kernel (the stepped form at any size: kernels.h) -- the hierarchical path is not even compiled
// into kernels whose register allocation follows every instruction of their riders; true: k_compact_fast, a launch of its own.
template <bool HIER_OK>
__device__ __forceinline__ void compact_fast_block(const CompactFastArgs& a, uint32_t bx, uint32_t by, uint32_t gx) {
    const uint32_t seg = by;
    const uint32_t view = seg / a.n_classes;
    const uint8_t* cnt = a.wave_cnt + (size_t)seg * a.n_waves;
    const uint64_t* mask = a.seg_mask ? a.seg_mask + (size_t)seg * a.seg_words
                                      : a.bitmask + view * a.words_per_view + a.word_offset;
    const uint32_t n_words = (a.n + 63u) >> 6;
    const uint32_t steps = compact_fast_steps(a.n, HIER_OK);
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    __shared__ uint32_t red[4], wtot[4];
    const bool hier = compact_fast_hier(a.n, HIER_OK);  // (uniform over the launch)
    const uint32_t n_chunks = compact_fast_chunks(a.n, HIER_OK);
    unsigned long long* const tab = reinterpret_cast<unsigned long long*>(a.seg_totals + ((a.n_segments + 1u) & ~1u)) + (size_t)seg * n_chunks;
    if (hier && bx < n_chunks) {
        // a summer: the counts of chunk bx (a byte per wave, sixteen per thread), published with this frame's stamp.  Relaxed agent-scope
        // accesses on the table: the word carries its own validity (stamp and total in one 64-bit store), so nothing has to be ordered
        // around it -- acquire / release here are cache invalidations and write-backs per access, and 2 500 workgroups polling 153 words
        // with them took 64 us per 10 M-row segment.
        const uint32_t w = bx * COMPACT_CHUNK_WORDS + threadIdx.x * 16u;
        uint32_t p = 0u;
        if (w < n_words) {
            const uint4 q = *reinterpret_cast<const uint4*>(cnt + w);
            const uint32_t v[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k) {
                const uint32_t wk = w + 4u * k;
                uint32_t x = wk < n_words ? v[k] : 0u;
                if (wk < n_words && n_words - wk < 4u) x &= (1u << (8u * (n_words - wk))) - 1u;  // (bytes past the last wave are not counts)
                p = sum_bytes(x, p);
            }
        }
#pragma unroll
        for (uint32_t off = 32u; off; off >>= 1) p += __shfl_xor(p, off, 64);
        if (lane == 0) red[wv] = p;
        __syncthreads();
        if (threadIdx.x == 0)
            __hip_atomic_store(tab + bx, ((unsigned long long)a.tag << 32) | (unsigned long long)(red[0] + red[1] + red[2] + red[3]), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const uint32_t w00 = (bx - n_chunks) * 64u * steps;

    // phase 1: base = sum of cnt[0 .. w00)
    uint32_t partial = 0;
    if (hier) {
        const uint32_t c0 = w00 / COMPACT_CHUNK_WORDS, wc = c0 * COMPACT_CHUNK_WORDS;
        const uint32_t off16 = threadIdx.x * 16u;  // (w00 is a multiple of 64: whole uint4s)
        if (off16 < w00 - wc) {
            const uint4 q = *reinterpret_cast<const uint4*>(cnt + wc + off16);
            partial = sum_bytes(q.x, sum_bytes(q.y, sum_bytes(q.z, sum_bytes(q.w, 0u))));
        }
        for (uint32_t c = threadIdx.x; c < c0; c += 256u) {
            unsigned long long e = __hip_atomic_load(tab + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while ((uint32_t)(e >> 32) != a.tag) {  // its summer has a lower workgroup id: dispatched before this workgroup, about to publish
                __builtin_amdgcn_s_sleep(2);
                e = __hip_atomic_load(tab + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            partial += (uint32_t)e;
        }
    } else {
        const uint4* c4 = reinterpret_cast<const uint4*>(cnt);
        for (uint32_t i = threadIdx.x; i < (w00 >> 4); i += 256u) {
            const uint4 q = c4[i];
            partial = sum_bytes(q.x, partial);
            partial = sum_bytes(q.y, partial);
            partial = sum_bytes(q.z, partial);
            partial = sum_bytes(q.w, partial);
        }
    }
#pragma unroll
    for (uint32_t off = 32u; off; off >>= 1) partial += __shfl_xor(partial, off, 64);
    if (lane == 0) red[wv] = partial;

    uint32_t running = 0;  // entries of this workgroup's earlier steps
    uint32_t* out = a.out_rows + (size_t)seg * a.seg_stride;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (uint32_t s = 0; s < steps; ++s) {
        // phase 2: this wave's 16 words of the step
        const uint32_t w0 = w00 + s * 64u;
        const uint32_t my_word = w0 + wv * 16u + lane;
        const unsigned long long m = (lane < 16u && my_word < n_words) ? mask[my_word] : 0ull;
        const uint32_t pc = __popcll(m);
        uint32_t incl = pc;
#pragma unroll
        for (uint32_t off = 1; off < 16u; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off, 64);
            if (lane >= off) incl += up;
        }
        const uint32_t excl = incl - pc;
        if (s) __syncthreads();  // the step before has read wtot
        if (lane == 15u) wtot[wv] = incl;
        __syncthreads();
        uint32_t base = red[0] + red[1] + red[2] + red[3] + running;
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) base += k < wv ? wtot[k] : 0u;
#pragma unroll 4
        for (uint32_t j = 0; j < 16u; ++j) {
            const unsigned long long mj = __shfl(m, (int)j, 64);
            const uint32_t oj = __shfl(excl, (int)j, 64);
            if ((mj >> lane) & 1ull) out[base + oj + __popcll(mj & lt)] = (w0 + wv * 16u + j) * 64u + lane;
        }
        running += wtot[0] + wtot[1] + wtot[2] + wtot[3];
    }
    if (bx == gx - 1u && threadIdx.x == 0) a.seg_totals[seg] = red[0] + red[1] + red[2] + red[3] + running;
}

}  // namespace mi
