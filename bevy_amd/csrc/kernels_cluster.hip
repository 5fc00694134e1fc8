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
// The two early-outs at the top of the per-object loop (assign.rs:489 RenderLayers, :496 frustum vs light sphere).
// ClusterableObjectAssignmentData::sphere (assign.rs:52-59): (x, y, z, range).  Row-bound objects take the centre from
// their row's GlobalTransform (point lights: GlobalTransform::from_translation(transform.translation()), :198).
__device__ __forceinline__ float4 object_sphere(const ClusterObjects& o, uint32_t obj) {
    float4 pr = reinterpret_cast<const float4*>(o.pos_range)[obj];
    if (o.row_global) {
        const float* g = o.row_global + 12ull * (o.first_row + obj);
        pr.x = g[9];
        pr.y = g[10];
        pr.z = g[11];
    }
    return pr;
}
__device__ __forceinline__ bool object_in_view(const ClusterViewDev& v, const ClusterObjects& o, uint32_t obj) {
    if (o.row_vv && !(o.row_vv[o.first_row + obj] & 1u)) return false;  // the gather's `if view_visibility.get()`, :194
    const float4 pr = object_sphere(o, obj);
    const uint32_t layers = o.layer_mask ? o.layer_mask[obj] : 1u;
    if (!(v.view_layer_mask & layers)) return false;  // :489
    V4 fr[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) fr[i] = V4{v.frustum[4 * i], v.frustum[4 * i + 1], v.frustum[4 * i + 2], v.frustum[4 * i + 3]};
    return frustum_intersects_sphere(fr, V3{pr.x, pr.y, pr.z}, pr.w, true);  // :496
}

// The rest of the body for an object that passed object_in_view.
// xp / yp / zp: the view's x, y, z cluster planes (LDS copies in the kernels below).
template <typename Emit>
__device__ __forceinline__ void assign_one_object(const ClusterViewDev& v, const ClusterObjects& o, uint32_t obj,
                                                  const float* xp, const float* yp, const float* zp,
                                                  float* far_z_out, bool* counted, Emit emit) {
    *counted = false;
    const float4 pr = object_sphere(o, obj);
    const V3 center = V3{pr.x, pr.y, pr.z};
    const float range = pr.w;
    const uint32_t type = o.obj_type ? o.obj_type[obj] : 0u;
    const bool ortho = v.is_orthographic != 0;

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
        V3 d;
        if (o.row_global) {  // GlobalTransform::back() = (matrix3 * Vec3::Z).normalize(), global_transform.rs:62-68,206
            const float* g = o.row_global + 12ull * (o.first_row + obj);
            M3 m3;
            m3.x_axis = V3{g[0], g[1], g[2]};
            m3.y_axis = V3{g[3], g[4], g[5]};
            m3.z_axis = V3{g[6], g[7], g[8]};
            const V3 z = mul(m3, V3{0.0f, 0.0f, 1.0f});
            d = z * f_div(1.0f, f_sqrt((z.x * z.x + z.y * z.y) + z.z * z.z));
        } else {
            d = V3{o.spot_dir[3 * obj], o.spot_dir[3 * obj + 1], o.spot_dir[3 * obj + 2]};
        }
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
            const V4 z_plane = (z_center_some && z < z_center) ? ldp(zp, z + 1u) : ldp(zp, z);
            if (!project_to_plane_z(z_object, z_plane)) continue;
        }
        for (uint32_t y = minc[1]; y <= maxc[1]; ++y) {
            Sphere y_object = z_object;
            if (!y_center_some || y != y_center) {
                const V4 y_plane = (y_center_some && y < y_center) ? ldp(yp, y + 1u) : ldp(yp, y);
                if (!project_to_plane_y(y_object, y_plane, ortho)) continue;
            }
            uint32_t min_x = minc[0];
            for (;;) {
                if (min_x >= maxc[0] ||
                    -get_distance_x(ldp(xp, min_x + 1u), y_object.center, ortho) + y_object.radius > 0.0f)
                    break;
                min_x += 1u;
            }
            uint32_t max_x = maxc[0];
            for (;;) {
                if (max_x <= min_x ||
                    get_distance_x(ldp(xp, max_x), y_object.center, ortho) + y_object.radius > 0.0f)
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

// Dynamic LDS of k_cluster_walk: n_clusters x 8 words (one 256-bit row per cluster) + 6 x 8 words of type
// masks + the view's planes + the pair counter.
extern __shared__ __attribute__((aligned(16))) uint32_t cluster_lds[];
constexpr size_t CLUSTER_MAX_DYN_LDS = 160u * 1024u - 2048u;  // 160 KB per workgroup minus the kernel's static LDS

// PLANES_IN_LDS = false only for degenerate grids whose plane tables do not fit next to the bit rows.
template <bool PLANES_IN_LDS>
__global__ void __launch_bounds__(CLUSTER_BLOCK) k_cluster_walk(ClusterViewDev v, ClusterObjects o, ClusterWork w) {
    const uint32_t C = v.n_clusters;
    uint32_t* rows = cluster_lds;
    uint32_t* type_rows = rows + C * 8u;
    float* planes = reinterpret_cast<float*>(type_rows + 48u);
    const uint32_t nx = v.dims[0] + 1u, ny = v.dims[1] + 1u, nz = v.dims[2] + 1u;
    // after the planes: one "touched" bit per cluster, the list of touched clusters (u16) and its length
    uint32_t* touched_bits = reinterpret_cast<uint32_t*>(planes + (PLANES_IN_LDS ? 4u * (nx + ny + nz) : 0u));
    uint32_t* n_touched = touched_bits + ((C + 31u) >> 5);
    uint16_t* touched_list = reinterpret_cast<uint16_t*>(n_touched + 1);
    const float* xp = PLANES_IN_LDS ? planes : v.x_planes;
    const float* yp = PLANES_IN_LDS ? planes + 4u * nx : v.y_planes;
    const float* zp = PLANES_IN_LDS ? planes + 4u * (nx + ny) : v.z_planes;

    // Most blocks of a big light set see nothing of it in this view: test first, and leave before touching LDS.
    const uint32_t obj = blockIdx.x * CLUSTER_BLOCK + threadIdx.x;
    const bool in_view = obj < o.n && object_in_view(v, o, obj);
    if (!__syncthreads_or(in_view ? 1 : 0)) return;

    {
        uint4* z4 = reinterpret_cast<uint4*>(cluster_lds);
        for (uint32_t i = threadIdx.x; i < C * 2u + 12u; i += CLUSTER_BLOCK) z4[i] = make_uint4(0u, 0u, 0u, 0u);
        for (uint32_t i = threadIdx.x; i <= ((C + 31u) >> 5); i += CLUSTER_BLOCK) touched_bits[i] = 0u;  // bits + counter
    }
    // the three plane tables are contiguous in device memory (x | y | z)
    if (PLANES_IN_LDS)
        for (uint32_t i = threadIdx.x; i < 4u * (nx + ny + nz); i += CLUSTER_BLOCK) planes[i] = v.x_planes[i];
    __syncthreads();

    if (in_view) {
        const uint32_t word = threadIdx.x >> 5, bit = 1u << (threadIdx.x & 31u);
        float far_z = 0.0f;
        bool counted = false;
        assign_one_object(v, o, obj, xp, yp, zp, &far_z, &counted, [&](uint32_t cluster) {
            atomicOr(&rows[cluster * 8u + word], bit);
            const uint32_t tb = 1u << (cluster & 31u);
            if (!(touched_bits[cluster >> 5] & tb) && !(atomicOr(&touched_bits[cluster >> 5], tb) & tb))
                touched_list[atomicAdd(n_touched, 1u)] = (uint16_t)cluster;  // first toucher records the cluster
        });
        const uint32_t type = o.obj_type ? o.obj_type[obj] : 0u;
        atomicOr(&type_rows[(type < 6u ? type : 5u) * 8u + word], bit);
        // farthest_z = farthest_z.max(this_object_far_z), starting from 0.0 (assign.rs:421,561):
        // only positive values can raise it, and positive floats order like their bit patterns.
        if (counted && far_z > 0.0f) atomicMax(reinterpret_cast<unsigned int*>(w.farthest_z), __float_as_uint(far_z));
    }
    __syncthreads();

    // Epilogue over the clusters this workgroup touched (a list kept next to the bit rows, so nothing is swept):
    // every touched row becomes a (cluster, block, 256-bit mask) pair in ONE global list -- the group reserves its
    // slots with a single atomic -- so the fill kernel can spread pairs evenly over the chip no matter how unevenly
    // the objects are distributed.  Only non-empty entries of the (cluster, block) count matrix are written.
    __shared__ uint32_t pair_base;
    const uint32_t nt = *n_touched;
    if (threadIdx.x == 0) pair_base = nt ? atomicAdd(w.pair_total, nt) : 0u;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nt; i += CLUSTER_BLOCK) {
        const uint32_t c = touched_list[i];
        const uint4 lo = reinterpret_cast<const uint4*>(rows)[c * 2u], hi = reinterpret_cast<const uint4*>(rows)[c * 2u + 1u];
        const uint32_t m[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        uint32_t cnt = 0;
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) cnt += __popc(m[k]);
        w.block_counts[(size_t)c * w.row_stride + blockIdx.x] = (uint16_t)cnt;  // cluster-major
        atomicAdd(&w.totals[c], cnt);
#pragma unroll
        for (uint32_t t = 0; t < 6; ++t) {
            uint32_t tc = 0;
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k) tc += __popc(m[k] & type_rows[t * 8u + k]);
            if (tc) atomicAdd(&w.counts[6u * c + t], tc);
        }
        const uint32_t slot = pair_base + i;
        w.pair_cb[slot] = (blockIdx.x << 12) | c;
        reinterpret_cast<uint4*>(w.pair_mask)[(size_t)slot * 2u] = lo;
        reinterpret_cast<uint4*>(w.pair_mask)[(size_t)slot * 2u + 1u] = hi;
    }
}

constexpr uint32_t CLUSTER_FILL_BLOCKS = 2048;


__global__ void __launch_bounds__(256) k_cluster_fill(ClusterWork w, uint32_t C, uint32_t n_objects) {
    __shared__ uint32_t offs[4096];
    __shared__ uint32_t part[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const uint32_t n_pairs = n_objects ? *w.pair_total : 0u;

    // zero the other parity's buffers for the next frame (this frame's buffers are only read from here on)
    {
        for (size_t i = blockIdx.x * 256u + tid; i < w.acc_words; i += (size_t)gridDim.x * 256u) w.acc_next[i] = 0u;
        uint4* m4 = reinterpret_cast<uint4*>(w.block_counts_next);
        const size_t n4 = (size_t)C * w.row_stride / 8u;
        for (size_t i = blockIdx.x * 256u + tid; i < n4; i += (size_t)gridDim.x * 256u) m4[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    const uint32_t first_wave = blockIdx.x * 4u;
    if (first_wave >= n_pairs && blockIdx.x != 0) return;

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
    __syncthreads();
    uint32_t before = incl - sum;
#pragma unroll
    for (uint32_t k = 0; k < 4u; ++k) before += k < wv ? part[k] : 0u;
    const uint32_t grand = part[0] + part[1] + part[2] + part[3];
#pragma unroll
    for (uint32_t k = 0; k < 16; ++k) offs[tid * 16u + k] = before + loc[k];
    __syncthreads();
    if (blockIdx.x == 0) {
        for (uint32_t c = tid; c < C; c += 256u) w.offsets[c] = offs[c];
        if (tid == 0) {
            w.offsets[C] = grand;
            *w.total = (uint64_t)grand;
        }
    }

    const unsigned long long lt = (1ull << lane) - 1ull;
    for (uint32_t p = first_wave + wv; p < n_pairs; p += gridDim.x * 4u) {
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
                if (d < w.capacity) w.indices[d] = b * CLUSTER_BLOCK + j * 64u + lane;
            }
            dst += __popcll(m64);
        }
    }
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

hipError_t launch_cluster_assign(const ClusterViewDev& view, const ClusterObjects& objs, const ClusterWork& w,
                                 hipStream_t stream, void (*mark)(void*, uint32_t), void* mctx) {
    const uint32_t C = view.n_clusters;
    const uint32_t n_planes = view.dims[0] + view.dims[1] + view.dims[2] + 3u;
    const size_t lds_rows = ((size_t)C * 8u + 48u) * sizeof(uint32_t);
    const size_t lds_touched = (((size_t)C + 31u) / 32u + 1u) * 4u + (size_t)C * 2u + 16u;
    const bool planes_in_lds = lds_rows + lds_touched + (size_t)n_planes * 16u <= CLUSTER_MAX_DYN_LDS;
    const size_t lds = lds_rows + lds_touched + (planes_in_lds ? (size_t)n_planes * 16u : 0u);
    // this frame's parity of the accumulators and of the count matrix was zeroed by the previous frame's fill kernel
    if (objs.n) {
        if (mark) mark(mctx, K_CLUSTER_WALK);
        if (planes_in_lds) MI_LAUNCH(k_cluster_walk<true>, dim3(w.n_blocks), dim3(CLUSTER_BLOCK), lds, stream, view, objs, w);
        else MI_LAUNCH(k_cluster_walk<false>, dim3(w.n_blocks), dim3(CLUSTER_BLOCK), lds, stream, view, objs, w);
    }
    if (mark) mark(mctx, K_CLUSTER_FILL);
    MI_LAUNCH(k_cluster_fill, dim3(CLUSTER_FILL_BLOCKS), dim3(256), 0, stream, w, C, objs.n);
    if (mark) mark(mctx, K_NUM_KERNELS);
    return hipGetLastError();
}

hipError_t set_cluster_lds_limit() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_cluster_walk<true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)CLUSTER_MAX_DYN_LDS);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(k_cluster_walk<false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)CLUSTER_MAX_DYN_LDS);
}

}  // namespace mi
