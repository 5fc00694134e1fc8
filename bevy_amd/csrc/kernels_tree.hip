// kernels_tree.hip -- hierarchy propagation on gfx950:
//   mark_dirty_trees              crates/bevy_transform/src/systems.rs:111-306
//   propagate_parent_transforms   crates/bevy_transform/src/systems.rs:506-748 (levels >= 1)
//
// Rows are in level (BFS) order, so the descendants of a contiguous range of nodes form one
// contiguous range per level.  A workgroup owns a *subtree tile*: a contiguous range of nodes
// at the band's first level plus all their descendants for the next few levels.  It walks the
// tile level by level with a workgroup barrier in between, keeping the GlobalTransforms of the
// level it just produced in LDS (2 x 512 rows x 48 B ping-pong), so a child reads its parent's
// matrix with three ds_read_b128 instead of going back to L2/HBM, and one launch covers several
// levels (the reference's mpsc work queue, systems.rs:767-813, becomes the tile list).
// A level wider than the LDS budget inside a tile falls back to reading parents from global
// memory, so any plan is correct; the host planner (context.cpp) only chooses the fast one.
//
// Algorithmic bytes per node: read T 40 + parent_idx 4 + old G 48 (set_if_neq, systems.rs:719),
// write G 48 + changed 1; parent G comes from LDS (first level of a tile: from L2).
#include "glam_math.h"
#include "kernels.h"

namespace mi {

struct F3 {
    float x, y, z;
};
__device__ __forceinline__ V3 ld3(const float* base, uint32_t row) {
    const F3 v = reinterpret_cast<const F3*>(base)[row];
    return V3{v.x, v.y, v.z};
}
__device__ __forceinline__ V4 ld4(const float* base, uint32_t row) {
    const float4 v = reinterpret_cast<const float4*>(base)[row];
    return V4{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ Affine unpack(float4 a, float4 b, float4 c) {
    Affine r;
    r.m.x_axis = V3{a.x, a.y, a.z};
    r.m.y_axis = V3{a.w, b.x, b.y};
    r.m.z_axis = V3{b.z, b.w, c.x};
    r.t = V3{c.y, c.z, c.w};
    return r;
}
__device__ __forceinline__ Affine ld_affine(const float* g, uint32_t row) {
    const float4* p = reinterpret_cast<const float4*>(g) + 3ull * row;
    return unpack(p[0], p[1], p[2]);
}
__device__ __forceinline__ void pack(const Affine& a, float4& o0, float4& o1, float4& o2) {
    o0 = make_float4(a.m.x_axis.x, a.m.x_axis.y, a.m.x_axis.z, a.m.y_axis.x);
    o1 = make_float4(a.m.y_axis.y, a.m.y_axis.z, a.m.z_axis.x, a.m.z_axis.y);
    o2 = make_float4(a.m.z_axis.z, a.t.x, a.t.y, a.t.z);
}

// mark_dirty_trees: climb from every changed row to its root, OR-ing the TransformTreeChanged bit;
// a climber stops at the first node somebody already marked (the shared atomic bitset of
// systems.rs:208-223).
__global__ void __launch_bounds__(256) k_mark_dirty(uint32_t n, const uint8_t* __restrict__ changed,
                                                     const uint32_t* __restrict__ parent_idx, uint32_t* tree_bits) {
    uint32_t row = blockIdx.x * 256u + threadIdx.x;
    if (row >= n || !changed[row]) return;
    for (uint32_t guard = 0; guard < n; ++guard) {
        const uint32_t bit = 1u << (row & 31u);
        const uint32_t old = atomicOr(&tree_bits[row >> 5], bit);
        if (old & bit) break;
        const uint32_t p = parent_idx ? parent_idx[row] : 0xFFFFFFFFu;
        if (p == 0xFFFFFFFFu) break;
        row = p;
    }
}

__global__ void __launch_bounds__(256) k_propagate_tiles(Columns c, const uint32_t* __restrict__ parent_idx,
                                                          const TileDesc* __restrict__ tiles,
                                                          const uint32_t* __restrict__ tree_bits,
                                                          uint8_t* g_changed_bytes, bool all_dirty, bool static_opt) {
    __shared__ float4 lds_g[2][TILE_LDS_ROWS * 3];
    __shared__ uint8_t lds_chg[2][TILE_LDS_ROWS];
    const TileDesc& td = tiles[blockIdx.x];
    const uint32_t n_levels = td.n_levels;
    for (uint32_t l = 0; l < n_levels; ++l) {
        const uint32_t start = td.start[l], count = td.count[l];
        const uint32_t prev_start = l ? td.start[l - 1] : 0u;
        const uint32_t prev_count = l ? td.count[l - 1] : 0u;
        const bool prev_in_lds = l && prev_count <= TILE_LDS_ROWS;
        const bool cur_to_lds = (l + 1 < n_levels) && count <= TILE_LDS_ROWS;
        const uint32_t rb = (l + 1u) & 1u, wb = l & 1u;
        for (uint32_t i = threadIdx.x; i < count; i += 256u) {
            const uint32_t row = start + i;
            const uint32_t p = parent_idx[row];
            Affine gp;
            bool p_changed;
            if (prev_in_lds) {
                const uint32_t slot = p - prev_start;
                gp = unpack(lds_g[rb][slot * 3], lds_g[rb][slot * 3 + 1], lds_g[rb][slot * 3 + 2]);
                p_changed = lds_chg[rb][slot] != 0;
            } else {
                gp = ld_affine(c.global, p);
                p_changed = g_changed_bytes[p] != 0;
            }
            const bool tree_changed = all_dirty || !tree_bits || ((tree_bits[row >> 5] >> (row & 31u)) & 1u);
            // static scene optimisation, systems.rs:708-714
            const bool skip = static_opt && !tree_changed && !p_changed;
            const Affine old = ld_affine(c.global, row);
            Affine cur = old;
            bool changed = false;
            if (!skip) {
                const Affine local = affine_from_srt(ld3(c.scale, row), ld4(c.rotation, row), ld3(c.translation, row));
                const Affine nw = mul(gp, local);  // p_global_transform.mul_transform(*transform)
                if (!affine_eq(nw, old)) {         // set_if_neq, systems.rs:719
                    float4 o0, o1, o2;
                    pack(nw, o0, o1, o2);
                    float4* dst = reinterpret_cast<float4*>(c.global) + 3ull * row;
                    dst[0] = o0; dst[1] = o1; dst[2] = o2;
                    cur = nw;
                    changed = true;
                }
            }
            g_changed_bytes[row] = changed ? 1 : 0;
            if (cur_to_lds) {
                float4 o0, o1, o2;
                pack(cur, o0, o1, o2);
                lds_g[wb][i * 3] = o0; lds_g[wb][i * 3 + 1] = o1; lds_g[wb][i * 3 + 2] = o2;
                lds_chg[wb][i] = changed ? 1 : 0;
            }
        }
        __syncthreads();  // also orders this level's global stores before the next level's fallback loads
    }
}

// Level-0 variant that also records the per-row changed byte (children look it up).
__global__ void __launch_bounds__(256) k_level0_bytes(const uint64_t* __restrict__ g_changed_bits, uint32_t n_level0,
                                                       uint8_t* g_changed_bytes) {
    const uint32_t row = blockIdx.x * 256u + threadIdx.x;
    if (row < n_level0) g_changed_bytes[row] = (uint8_t)((g_changed_bits[row >> 6] >> (row & 63u)) & 1ull);
}

hipError_t launch_mark_dirty(uint32_t n, const uint8_t* changed, const uint32_t* parent_idx, uint32_t* tree_bits,
                             hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_mark_dirty, dim3((n + 255u) / 256u), dim3(256), 0, stream, n, changed, parent_idx, tree_bits);
    return hipGetLastError();
}

hipError_t launch_propagate_tiles(const Columns& c, const uint32_t* parent_idx, const TileDesc* d_tiles,
                                  uint32_t n_tiles, const uint32_t* tree_bits, uint8_t* g_changed_bytes,
                                  bool all_dirty, bool static_opt, hipStream_t stream) {
    if (n_tiles == 0) return hipSuccess;
    hipLaunchKernelGGL(k_propagate_tiles, dim3(n_tiles), dim3(256), 0, stream, c, parent_idx, d_tiles, tree_bits,
                       g_changed_bytes, all_dirty, static_opt);
    return hipGetLastError();
}

hipError_t launch_level0_bytes(const uint64_t* g_changed_bits, uint32_t n_level0, uint8_t* g_changed_bytes,
                               hipStream_t stream) {
    if (n_level0 == 0) return hipSuccess;
    hipLaunchKernelGGL(k_level0_bytes, dim3((n_level0 + 255u) / 256u), dim3(256), 0, stream, g_changed_bits, n_level0,
                       g_changed_bytes);
    return hipGetLastError();
}

}  // namespace mi
