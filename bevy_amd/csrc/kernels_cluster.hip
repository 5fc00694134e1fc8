// kernels_cluster.hip -- assign_objects_to_clusters on gfx950
//   crates/bevy_light/src/cluster/assign.rs:487-811 (per-object loop) and its helpers :903-1134.
//
// The reference is one serial triple loop pushing Entities into per-cluster Vecs; the order of
// every cluster's list is the object iteration order.  Here:
//   count : one object per lane walks the same z -> y -> x-range refinement and sets ITS bit in a
//           per-cluster 256-bit row held in LDS (clusters x 32 B, 108 KB for 16x9x24); after a
//           barrier the workgroup popcounts every row -> per-(cluster, block) counts, per-type counts.
//   scan  : one wave per cluster prefix-sums its row of block counts (cluster-major matrix), then one
//           workgroup prefix-sums the cluster totals -> CSR offsets.
//   fill  : the walk again, bits again, then every cluster row is expanded in bit order: the rank of
//           an object inside its block is the popcount of the lower bits, so each cluster's segment
//           is written in ascending object order == the reference's push order, with no sort and no
//           order-dependent atomics.
// HBM traffic is tiny (17 B/object + 4 B/entry); the path is latency/launch bound, see DESIGN.md.
#include "glam_math.h"
#include "kernels.h"

namespace mi {

// glibc >= 2.28 logf (ARM optimized-routines algorithm, table size 16, degree-3 polynomial in
// double).  Rust's f32::ln is the platform libm's logf (bevy_math/src/ops.rs:22-60), so this is the
// function view_z_to_z_slice (assign.rs:1057) evaluates on the reference's CPU path.  Verified
// bit-identical to libm logf for every non-negative binary32 (tests/test_logf.py keeps a sample).
__device__ __constant__ double LOGF_TAB[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
    {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},  {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
    {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
    {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1p+0, 0x0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
    {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3},
    {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},  {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};

__device__ __forceinline__ float libm_logf(float x) {
    uint32_t ix = __float_as_uint(x);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
        if (ix * 2u == 0u) return -INFINITY;
        if (ix == 0x7f800000u) return x;
        if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return NAN;
        ix = __float_as_uint(x * 0x1p23f);
        ix -= 23u << 23;
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) & 15u);
    const int k = (int32_t)tmp >> 23;
    const uint32_t iz = ix - (tmp & 0xff800000u);
    const double invc = LOGF_TAB[i][0], logc = LOGF_TAB[i][1];
    const double z = (double)__uint_as_float(iz);
    const double r = z * invc - 1.0;
    const double y0 = logc + (double)k * 0x1.62e42fefa39efp-1;
    const double r2 = r * r;
    double y = 0x1.5575b0be00b6ap-2 * r + -0x1.ffffef20a4123p-2;
    y = -0x1.00ea348b88334p-2 * r2 + y;
    y = y * r2 + (y0 + r);
    return (float)y;
}
__global__ void k_logf_probe(const float* __restrict__ in, float* out, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) out[i] = libm_logf(in[i]);
}
hipError_t launch_logf_probe(const float* in, float* out, uint32_t n, hipStream_t stream) {
    if (!n) return hipSuccess;
    MI_LAUNCH(k_logf_probe, dim3((n + 255u) / 256u), dim3(256), 0, stream, in, out, n);
    return hipGetLastError();
}

struct Sphere {
    V3 center;
    float radius;
};

// view_z_to_z_slice, assign.rs:1046-1062
__device__ __forceinline__ uint32_t view_z_to_z_slice(const ClusterViewDev& v, float view_z) {
    uint32_t z_slice;
    if (v.is_orthographic) z_slice = f32_as_u32(floorf((view_z - v.cluster_factors[0]) * v.cluster_factors[1]));
    else z_slice = f32_as_u32(libm_logf(-view_z) * v.cluster_factors[0] - v.cluster_factors[1] + 1.0f);
    const uint32_t lim = v.dims[2] - 1u;
    return z_slice < lim ? z_slice : lim;
}
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return lane_min(lane_max(x, lo), hi); }
// ndc_position_to_cluster, assign.rs:922-941
__device__ __forceinline__ void ndc_position_to_cluster(const ClusterViewDev& v, float ndc_x, float ndc_y, float view_z,
                                                        uint32_t out[3]) {
    const float fx = clampf(ndc_x * 0.5f + 0.5f, 0.0f, 1.0f);
    const float fy = clampf(ndc_y * -0.5f + 0.5f, 0.0f, 1.0f);
    const uint32_t xi = f32_as_u32(floorf(fx * (float)v.dims[0]));
    const uint32_t yi = f32_as_u32(floorf(fy * (float)v.dims[1]));
    const uint32_t zs = view_z_to_z_slice(v, view_z);
    out[0] = xi > v.dims[0] - 1u ? v.dims[0] - 1u : xi;
    out[1] = yi > v.dims[1] - 1u ? v.dims[1] - 1u : yi;
    out[2] = zs > v.dims[2] - 1u ? v.dims[2] - 1u : zs;
}
__device__ __forceinline__ V4 ldp(const float* planes, uint32_t i) {
    const float4 p = reinterpret_cast<const float4*>(planes)[i];
    return V4{p.x, p.y, p.z, p.w};
}
// project_to_plane_z, assign.rs:1094-1113
__device__ __forceinline__ bool project_to_plane_z(Sphere& s, V4 plane) {
    const float z = f_div(plane.w, plane.z);
    const float dist = z - s.center.z;
    if (f_abs(dist) > s.radius) return false;
    s.center.z = z;
    s.radius = f_sqrt(s.radius * s.radius - dist * dist);
    return true;
}
// project_to_plane_y, assign.rs:1116-1134
__device__ __forceinline__ bool project_to_plane_y(Sphere& s, V4 plane, bool ortho) {
    float dist;
    if (ortho) dist = plane.w - s.center.y;
    else dist = -(s.center.y * plane.y + s.center.z * plane.z);
    if (f_abs(dist) > s.radius) return false;
    s.center = s.center + xyz(plane) * dist;
    s.radius = f_sqrt(s.radius * s.radius - dist * dist);
    return true;
}
// get_distance_x, assign.rs:1081-1091
__device__ __forceinline__ float get_distance_x(V4 plane, V3 p, bool ortho) {
    if (ortho) return p.x - plane.w;
    return plane.x * p.x + plane.z * p.z;
}

// The body of `for clusterable_object in &clusterable_objects` (assign.rs:487-804) for one object.
// emit(cluster_index) is called for every cluster the reference would push this object into.
template <typename Emit>
__device__ __forceinline__ void assign_one_object(const ClusterViewDev& v, const ClusterObjects& o, uint32_t obj,
                                                  float* far_z_out, bool* counted, Emit emit) {
    *counted = false;
    const float4 pr = reinterpret_cast<const float4*>(o.pos_range)[obj];
    const V3 center = V3{pr.x, pr.y, pr.z};
    const float range = pr.w;
    const uint32_t type = o.obj_type ? o.obj_type[obj] : 0u;
    const uint32_t layers = o.layer_mask ? o.layer_mask[obj] : 1u;
    const bool ortho = v.is_orthographic != 0;
    if (!(v.view_layer_mask & layers)) return;  // :489
    V4 fr[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) fr[i] = V4{v.frustum[4 * i], v.frustum[4 * i + 1], v.frustum[4 * i + 2], v.frustum[4 * i + 3]};
    if (!frustum_intersects_sphere(fr, center, range, true)) return;  // :496

    const M4 view_from_world = load_m4(v.view_from_world);
    const M4 clip_from_view = load_m4(v.clip_from_view);
    const V3 scale = V3{v.view_from_world_scale[0], v.view_from_world_scale[1], v.view_from_world_scale[2]};

    // cluster_space_clusterable_object_aabb, :948-1036
    const V3 cv = xyz(mul(view_from_world, extend(center, 1.0f)));
    const V3 he = abs3(scale) * range;
    V3 vmin = cv - he, vmax = cv + he;
    const float NEG_MIN_POS = -1.17549435e-38f;
    vmin.z = rust_min(vmin.z, NEG_MIN_POS);
    vmax.z = rust_min(vmax.z, NEG_MIN_POS);
    const V3 corner[4] = {vmin, V3{vmin.x, vmin.y, vmax.z}, V3{vmax.x, vmax.y, vmin.z}, vmax};
    V3 nmin, nmax;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const V4 clip = mul(clip_from_view, extend(corner[i], 1.0f));
        const V3 ndc = V3{f_div(clip.x, clip.w), f_div(clip.y, clip.w), f_div(clip.z, clip.w)};
        if (i == 0) { nmin = ndc; nmax = ndc; }
        else { nmin = min3(nmin, ndc); nmax = max3(nmax, ndc); }
    }
    uint32_t c0[3], c1[3], minc[3], maxc[3];
    ndc_position_to_cluster(v, clampf(nmin.x, -1.0f, 1.0f), clampf(nmin.y, -1.0f, 1.0f), vmin.z, c0);
    ndc_position_to_cluster(v, clampf(nmax.x, -1.0f, 1.0f), clampf(nmax.y, -1.0f, 1.0f), vmax.z, c1);
#pragma unroll
    for (int k = 0; k < 3; ++k) { minc[k] = c0[k] < c1[k] ? c0[k] : c1[k]; maxc[k] = c0[k] > c1[k] ? c0[k] : c1[k]; }

    Sphere vs;
    vs.center = cv;  // same expression as :552-554
    vs.radius = range * v.view_from_world_scale_max;

    *far_z_out = -dot4(row(view_from_world, 2), extend(center, 1.0f)) + range * scale.z;  // :558-560
    *counted = true;

    V3 light_dir = V3{0.0f, 0.0f, 0.0f};
    float angle_sin = 0.0f, angle_cos = 0.0f;
    if (type == 1u) {  // spot light, :563-573
        const V3 d = V3{o.spot_dir[3 * obj], o.spot_dir[3 * obj + 1], o.spot_dir[3 * obj + 2]};
        const V3 dv = xyz(mul(view_from_world, extend(d, 0.0f)));
        light_dir = dv * f_div(1.0f, f_sqrt(dot3(dv, dv)));
        angle_sin = o.spot_sin_cos[2 * obj];
        angle_cos = o.spot_sin_cos[2 * obj + 1];
    }
    const V4 center_clip = mul(clip_from_view, extend(vs.center, 1.0f));
    const V3 ndc = V3{f_div(center_clip.x, center_clip.w), f_div(center_clip.y, center_clip.w),
                      f_div(center_clip.z, center_clip.w)};
    uint32_t cc[3];
    ndc_position_to_cluster(v, ndc.x, ndc.y, vs.center.z, cc);
    const bool z_center_some = ndc.z <= 1.0f;
    const uint32_t z_center = cc[2];
    bool y_center_some;
    uint32_t y_center = 0;
    if (ndc.y > 1.0f) y_center_some = false;
    else if (ndc.y < -1.0f) { y_center_some = true; y_center = v.dims[1] + 1u; }
    else { y_center_some = true; y_center = cc[1]; }

    for (uint32_t z = minc[2]; z <= maxc[2]; ++z) {
        Sphere z_object = vs;
        if (!z_center_some || z != z_center) {
            const V4 z_plane = (z_center_some && z < z_center) ? ldp(v.z_planes, z + 1u) : ldp(v.z_planes, z);
            if (!project_to_plane_z(z_object, z_plane)) continue;
        }
        for (uint32_t y = minc[1]; y <= maxc[1]; ++y) {
            Sphere y_object = z_object;
            if (!y_center_some || y != y_center) {
                const V4 y_plane = (y_center_some && y < y_center) ? ldp(v.y_planes, y + 1u) : ldp(v.y_planes, y);
                if (!project_to_plane_y(y_object, y_plane, ortho)) continue;
            }
            uint32_t min_x = minc[0];
            for (;;) {
                if (min_x >= maxc[0] ||
                    -get_distance_x(ldp(v.x_planes, min_x + 1u), y_object.center, ortho) + y_object.radius > 0.0f)
                    break;
                min_x += 1u;
            }
            uint32_t max_x = maxc[0];
            for (;;) {
                if (max_x <= min_x ||
                    get_distance_x(ldp(v.x_planes, max_x), y_object.center, ortho) + y_object.radius > 0.0f)
                    break;
                max_x -= 1u;
            }
            uint32_t cluster_index = (y * v.dims[0] + min_x) * v.dims[2] + z;
            if (type == 1u) {
                for (uint32_t x = min_x; x <= max_x; ++x) {
                    const float4 cs = reinterpret_cast<const float4*>(v.cluster_spheres)[cluster_index];
                    const V3 off = vs.center - V3{cs.x, cs.y, cs.z};
                    const float dist_sq = dot3(off, off);
                    const float v1_len = dot3(off, light_dir);
                    const float dcp = (angle_cos * f_sqrt(dist_sq - v1_len * v1_len)) - v1_len * angle_sin;
                    const bool angle_cull = dcp > cs.w;
                    const bool front_cull = v1_len > cs.w + range * v.view_from_world_scale_max;
                    const bool back_cull = v1_len < -cs.w;
                    if (!angle_cull && !front_cull && !back_cull) emit(cluster_index);
                    cluster_index += v.dims[2];
                }
            } else {
                for (uint32_t x = min_x; x <= max_x; ++x) {
                    emit(cluster_index);
                    cluster_index += v.dims[2];
                }
            }
        }
    }
}

// Dynamic LDS: n_clusters x 8 words (one 256-bit row per cluster) + 6 x 8 words of type masks.
extern __shared__ __attribute__((aligned(16))) uint32_t cluster_lds[];

__device__ __forceinline__ void walk_block_into_lds(const ClusterViewDev& v, const ClusterObjects& o, uint32_t* rows,
                                                    uint32_t* type_rows, float* farthest_z_bits_out) {
    const uint32_t C = v.n_clusters;
    for (uint32_t i = threadIdx.x; i < C * 8u + 48u; i += CLUSTER_BLOCK) cluster_lds[i] = 0u;
    __syncthreads();
    const uint32_t obj = blockIdx.x * CLUSTER_BLOCK + threadIdx.x;
    if (obj < o.n) {
        const uint32_t word = threadIdx.x >> 5, bit = 1u << (threadIdx.x & 31u);
        float far_z = 0.0f;
        bool counted = false;
        assign_one_object(v, o, obj, &far_z, &counted, [&](uint32_t cluster) { atomicOr(&rows[cluster * 8u + word], bit); });
        const uint32_t type = o.obj_type ? o.obj_type[obj] : 0u;
        atomicOr(&type_rows[(type < 6u ? type : 5u) * 8u + word], bit);
        // farthest_z = farthest_z.max(this_object_far_z), starting from 0.0 (assign.rs:421,561):
        // only positive values can raise it, and positive floats order like their bit patterns.
        if (farthest_z_bits_out && counted && far_z > 0.0f)
            atomicMax(reinterpret_cast<unsigned int*>(farthest_z_bits_out), __float_as_uint(far_z));
    }
    __syncthreads();
}

__global__ void __launch_bounds__(CLUSTER_BLOCK) k_cluster_count(ClusterViewDev v, ClusterObjects o, ClusterWork w) {
    uint32_t* rows = cluster_lds;
    uint32_t* type_rows = cluster_lds + v.n_clusters * 8u;
    walk_block_into_lds(v, o, rows, type_rows, w.farthest_z);
    const uint32_t C = v.n_clusters;
    for (uint32_t c = threadIdx.x; c < C; c += CLUSTER_BLOCK) {
        uint32_t cnt = 0;
        uint32_t tc[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) {
            const uint32_t m = rows[c * 8u + k];
            cnt += __popc(m);
            if (m) {
#pragma unroll
                for (uint32_t t = 0; t < 6; ++t) tc[t] += __popc(m & type_rows[t * 8u + k]);
            }
        }
        w.block_counts[(size_t)c * w.n_blocks + blockIdx.x] = (uint16_t)cnt;  // cluster-major
        if (cnt) {
#pragma unroll
            for (uint32_t t = 0; t < 6; ++t)
                if (tc[t]) atomicAdd(&w.counts[6u * c + t], tc[t]);
        }
    }
}

// One wave per cluster: exclusive prefix sum over its row of per-block counts.
__global__ void __launch_bounds__(256) k_cluster_scan_rows(ClusterWork w, uint32_t n_clusters, uint32_t* cluster_totals) {
    const uint32_t c = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    if (c >= n_clusters) return;
    const uint16_t* src = w.block_counts + (size_t)c * w.n_blocks;
    uint32_t* dst = w.block_bases + (size_t)c * w.n_blocks;
    uint32_t running = 0;
    for (uint32_t b0 = 0; b0 < w.n_blocks; b0 += 64u) {
        const uint32_t b = b0 + lane;
        const uint32_t val = b < w.n_blocks ? (uint32_t)src[b] : 0u;
        uint32_t incl = val;
#pragma unroll
        for (uint32_t off = 1; off < 64u; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off, 64);
            if (lane >= off) incl += up;
        }
        if (b < w.n_blocks) dst[b] = running + incl - val;
        running += __shfl(incl, 63, 64);
    }
    if (lane == 0) cluster_totals[c] = running;
}

// One workgroup: CSR offsets over clusters (n_clusters <= 4096) and the grand total.
__global__ void __launch_bounds__(1024) k_cluster_scan_offsets(ClusterWork w, uint32_t n_clusters,
                                                                const uint32_t* __restrict__ cluster_totals) {
    __shared__ uint32_t part[1024];
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t c0 = 0; c0 < n_clusters; c0 += 1024u) {
        const uint32_t c = c0 + threadIdx.x;
        const uint32_t val = c < n_clusters ? cluster_totals[c] : 0u;
        part[threadIdx.x] = val;
        __syncthreads();
        for (uint32_t off = 1; off < 1024u; off <<= 1) {
            const uint32_t add = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
            __syncthreads();
            part[threadIdx.x] += add;
            __syncthreads();
        }
        const uint32_t carry = carry_s;
        if (c < n_clusters) w.offsets[c] = carry + part[threadIdx.x] - val;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + part[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        w.offsets[n_clusters] = carry_s;
        *w.total = (uint64_t)carry_s;
    }
}

__global__ void __launch_bounds__(CLUSTER_BLOCK) k_cluster_fill(ClusterViewDev v, ClusterObjects o, ClusterWork w) {
    uint32_t* rows = cluster_lds;
    uint32_t* type_rows = cluster_lds + v.n_clusters * 8u;
    walk_block_into_lds(v, o, rows, type_rows, nullptr);
    const uint32_t C = v.n_clusters;
    const uint32_t obj_base = blockIdx.x * CLUSTER_BLOCK;
    for (uint32_t c = threadIdx.x; c < C; c += CLUSTER_BLOCK) {
        uint64_t dst = (uint64_t)w.offsets[c] + w.block_bases[(size_t)c * w.n_blocks + blockIdx.x];
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) {
            uint32_t m = rows[c * 8u + k];
            while (m) {
                const uint32_t b = __ffs(m) - 1u;
                m &= m - 1u;
                if (dst < w.capacity) w.indices[dst] = obj_base + k * 32u + b;
                ++dst;
            }
        }
    }
}

hipError_t launch_cluster_assign(const ClusterViewDev& view, const ClusterObjects& objs, const ClusterWork& w,
                                 hipStream_t stream, void (*mark)(void*, uint32_t), void* mctx) {
    const uint32_t C = view.n_clusters;
    const size_t lds = ((size_t)C * 8u + 48u) * sizeof(uint32_t);
    // w.counts, w.total, w.farthest_z and the scratch were cleared by the caller on this stream.
    if (objs.n) {
        if (mark) mark(mctx, K_CLUSTER_COUNT);
        MI_LAUNCH(k_cluster_count, dim3(w.n_blocks), dim3(CLUSTER_BLOCK), lds, stream, view, objs, w);
    }
    if (mark) mark(mctx, K_CLUSTER_SCAN);
    uint32_t* cluster_totals = w.offsets;  // reuse: totals are consumed into offsets in place
    if (objs.n) {
        MI_LAUNCH(k_cluster_scan_rows, dim3((C + 3u) / 4u), dim3(256), 0, stream, w, C, cluster_totals);
    }
    MI_LAUNCH(k_cluster_scan_offsets, dim3(1), dim3(1024), 0, stream, w, C, cluster_totals);
    if (objs.n) {
        if (mark) mark(mctx, K_CLUSTER_FILL);
        MI_LAUNCH(k_cluster_fill, dim3(w.n_blocks), dim3(CLUSTER_BLOCK), lds, stream, view, objs, w);
    }
    if (mark) mark(mctx, K_NUM_KERNELS);
    return hipGetLastError();
}

hipError_t set_cluster_lds_limit() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_cluster_count),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(k_cluster_fill),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

}  // namespace mi
