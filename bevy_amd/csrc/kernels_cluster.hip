// kernels_cluster.hip -- assign_objects_to_clusters on gfx950
//   crates/bevy_light/src/cluster/assign.rs:487-811 (per-object loop) and its helpers :903-1134.
//
// The reference is one serial triple loop pushing Entities into per-cluster Vecs; the order of
// every cluster's list is the object iteration order.  Here (two launches, no memsets, no scan kernel):
//   walk : one object per lane walks the same z -> y -> x-range refinement ONCE -- the view's cluster planes
//          are staged in LDS, so the data-dependent plane reads of the refinement loops cost an LDS
//          round trip, not an L2 one -- and sets ITS bit in a per-cluster 256-bit row held in LDS
//          (clusters x 32 B, 108 KB for 16x9x24).  After a barrier the workgroup popcounts every row:
//          per-(cluster, block) counts (cluster-major u16 matrix), cluster totals and per-type counts
//          (atomics into the frame-parity buffer), and the non-empty rows themselves as
//          (cluster, 256-bit mask) pairs in the block's scratch list.
//   fill : every workgroup prefix-sums the 4096-max cluster totals in LDS (-> CSR offsets), then one wave
//          per pair adds the counts of the lower-numbered blocks of that cluster (a contiguous u16 row)
//          and expands the mask in bit order: the rank of an object inside its block is the popcount of
//          the lower bits, so each cluster's segment is written in ascending object order == the
//          reference's push order, with no sort and no order-dependent atomics.  It also zeroes the
//          other parity's atomic buffers for the next frame.
// HBM traffic is tiny (17 B/object + 4 B/entry); the stage is latency-bound, see DESIGN.md.
#include "glam_math.h"
#include "kernels.h"
#include "visibility_rule.h"
#include "cluster_fill.h"
#include "cluster_walk.h"

namespace mi {

__global__ void k_logf_probe(const float* __restrict__ in, float* out, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) out[i] = libm_logf(in[i]);
}
hipError_t launch_logf_probe(const float* in, float* out, uint32_t n, hipStream_t stream) {
    if (!n) return hipSuccess;
    MI_LAUNCH(k_logf_probe, dim3((n + 255u) / 256u), dim3(256), 0, stream, in, out, n);
    return hipGetLastError();
}

extern __shared__ __attribute__((aligned(16))) uint32_t cluster_lds[];
constexpr size_t CLUSTER_MAX_DYN_LDS = 160u * 1024u - 2048u;  // 160 KB per workgroup minus the kernel's static LDS
constexpr size_t CLUSTER_SMALL_ROWS_BYTES = 24u * 1024u;

// wp: the view's plane table as a kernel argument (WalkPlanes, kernels.h) -- read when the view carries no pointer of its own
struct ClusterWalkKernargs {
    ClusterViewDev v; ClusterObjects o; ClusterWork w; ViewSet views; uint32_t zc; WalkPlanes wp;
};
template <bool PLANES_IN_LDS, bool CHUNKED>
__global__ void __launch_bounds__(CLUSTER_BLOCK) k_cluster_walk(ClusterViewDev v, ClusterObjects o, ClusterWork w, ViewSet views, uint32_t zc, WalkPlanes wp) {
    cluster_walk_block<PLANES_IN_LDS, CHUNKED>(v, o, w, views, zc, blockIdx.x, cluster_lds, &kernarg_late<float>((uint32_t)offsetof(ClusterWalkKernargs, wp)));
}

constexpr uint32_t CLUSTER_FILL_BLOCKS = 2048;


__global__ void __launch_bounds__(256) k_cluster_fill(ClusterWork w, uint32_t C, uint32_t n_objects) {
    __shared__ uint32_t offs[4096];
    __shared__ uint32_t part[4];
    cluster_fill_block(w, C, n_objects, blockIdx.x, gridDim.x, offs, part);
}

// The storage-buffer wire format of a view's clusters (crates/bevy_pbr/src/cluster/mod.rs:478-582,634-650): per
// cluster [uvec4(offset, point, spot, rect), uvec4(reflection probes, irradiance volumes, decals, 0)] and the flat index
// list with object indices replaced by their render-world indices.
__global__ void __launch_bounds__(256) k_cluster_bindings(uint32_t n_clusters, const uint32_t* __restrict__ offsets,
                                                           const uint32_t* __restrict__ counts,
                                                           const uint32_t* __restrict__ indices,
                                                           const uint32_t* __restrict__ remap, uint32_t n_remap,
                                                           uint64_t capacity, uint4* out_oc, uint32_t* out_idx) {
    const uint32_t total = offsets[n_clusters];
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n_clusters; i += gridDim.x * 256u) {
        out_oc[2u * i] = make_uint4(offsets[i], counts[6u * i], counts[6u * i + 1u], counts[6u * i + 2u]);
        out_oc[2u * i + 1u] = make_uint4(counts[6u * i + 3u], counts[6u * i + 4u], counts[6u * i + 5u], 0u);
    }
    const uint64_t lim = total < capacity ? total : capacity;
    for (uint64_t i = blockIdx.x * 256u + threadIdx.x; i < lim; i += (uint64_t)gridDim.x * 256u) {
        const uint32_t obj = indices[i];
        out_idx[i] = remap ? (obj < n_remap ? remap[obj] : 0xFFFFFFFFu) : obj;
    }
}
hipError_t launch_cluster_bindings(uint32_t n_clusters, const uint32_t* offsets, const uint32_t* counts, const uint32_t* indices,
                                   const uint32_t* remap, uint32_t n_remap, uint64_t capacity, uint32_t* out_oc,
                                   uint32_t* out_idx, hipStream_t stream) {
    MI_LAUNCH(k_cluster_bindings, dim3(1024), dim3(256), 0, stream, n_clusters, offsets, counts, indices, remap, n_remap, capacity,
              reinterpret_cast<uint4*>(out_oc), out_idx);
    return hipGetLastError();
}

hipError_t launch_cluster_assign(const ClusterViewDev& view, const ClusterObjects& objs, const ClusterWork& w, const ViewSet* frame_views,
                                 bool small_lds, bool fill, hipStream_t stream, void (*mark)(void*, uint32_t), void* mctx, WalkPlanesHost walk_planes) {
    static const ViewSet no_views = {};
    const ViewSet& fv = frame_views ? *frame_views : no_views;
    const uint32_t n_planes = view.dims[0] + view.dims[1] + view.dims[2] + 3u;
    const uint32_t C = view.n_clusters, dxy = view.dims[0] * view.dims[1], dz = view.dims[2];
    // z chunk: the whole grid at once when its bit rows fit the LDS (one sweep, the fastest); otherwise as many slices as do.
    // `small` (the concurrent mode, where the workgroups must fit on a CU next to the frame kernel's): chunks of ~24 KB.
    auto lds_for = [&](uint32_t zc_, bool with_planes) -> size_t { return cluster_walk_lds_bytes(dxy, zc_, n_planes, with_planes); };
    uint32_t zc = dz;
    if (small_lds) zc = std::max<uint32_t>(1u, std::min<uint32_t>(dz, (uint32_t)(CLUSTER_SMALL_ROWS_BYTES / ((size_t)dxy * 32u))));
    while (zc > 1u && lds_for(zc, false) > CLUSTER_MAX_DYN_LDS) --zc;
    const bool planes_in_lds = lds_for(zc, true) <= CLUSTER_MAX_DYN_LDS;
    const size_t lds = lds_for(zc, planes_in_lds);
    // this frame's parity of the accumulators and of the count matrix was zeroed by the previous frame's fill kernel
    // the plane table as a kernel argument where it fits and the walkers copy it to LDS anyway (else the staged copy, through the view's pointers)
    const WalkPlanesHost ph = walk_planes;
    WalkPlanes wp;  // (only the bytes filled below are read)
    ClusterViewDev vd = view;
#ifndef MI_EXP_STAGED_PLANES
    if (planes_in_lds && ph.f && ph.n && ph.n <= WALK_PLANES_MAX) {
        for (uint32_t i = 0; i < ph.n; ++i) wp.f[i] = ph.f[i];
        vd.x_planes = vd.y_planes = vd.z_planes = nullptr;
    }
#endif
    if (objs.n) {
        if (mark) mark(mctx, K_CLUSTER_WALK);
        const dim3 grid(w.n_blocks), block(CLUSTER_BLOCK);
        if (zc == dz) {
            if (planes_in_lds) MI_LAUNCH((k_cluster_walk<true, false>), grid, block, lds, stream, vd, objs, w, fv, zc, wp);
            else MI_LAUNCH((k_cluster_walk<false, false>), grid, block, lds, stream, vd, objs, w, fv, zc, wp);
        } else {
            if (planes_in_lds) MI_LAUNCH((k_cluster_walk<true, true>), grid, block, lds, stream, vd, objs, w, fv, zc, wp);
            else MI_LAUNCH((k_cluster_walk<false, true>), grid, block, lds, stream, vd, objs, w, fv, zc, wp);
        }
    }
    if (!fill) return hipGetLastError();
    if (mark) mark(mctx, K_CLUSTER_FILL);
    MI_LAUNCH(k_cluster_fill, dim3(CLUSTER_FILL_BLOCKS), dim3(256), 0, stream, w, C, objs.n);
    if (mark) mark(mctx, K_NUM_KERNELS);
    return hipGetLastError();
}
hipError_t launch_cluster_fill(const ClusterWork& w, uint32_t n_clusters, uint32_t n_objects, hipStream_t stream) {
    MI_LAUNCH(k_cluster_fill, dim3(CLUSTER_FILL_BLOCKS), dim3(256), 0, stream, w, n_clusters, n_objects);
    return hipGetLastError();
}

hipError_t set_cluster_lds_limit() {
    const void* fns[] = {reinterpret_cast<const void*>(k_cluster_walk<true, false>), reinterpret_cast<const void*>(k_cluster_walk<false, false>),
                         reinterpret_cast<const void*>(k_cluster_walk<true, true>), reinterpret_cast<const void*>(k_cluster_walk<false, true>)};
    for (const void* f : fns) {
        hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CLUSTER_MAX_DYN_LDS);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace mi
