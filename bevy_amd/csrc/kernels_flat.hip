// kernels_flat.hip -- gfx950 kernels for the flat (non-hierarchical) render-prep path:
//   sync_simple_transforms          crates/bevy_transform/src/systems.rs:42-79
//   reset_view_visibility           crates/bevy_camera/src/visibility/mod.rs:733-737
//   check_visibility_cpu_culling    crates/bevy_camera/src/visibility/mod.rs:748-876
//   check_visibility_gpu_culling    crates/bevy_camera/src/visibility/mod.rs:884-903
//   mark_newly_hidden_entities_...  crates/bevy_camera/src/visibility/mod.rs:908-918
//
// All of it is HBM-bound streaming work (roofline: DESIGN.md "Kernels").  One row per lane,
// 256-thread workgroups (4 wave64 per group, grid >> 256 CUs), component columns read as
// 12/16-byte lane-contiguous chunks, per-view visibility packed with a wave64 ballot: lane 0
// of every wave stores one 64-bit mask word per view, so the bitmask is written coalesced and
// needs no atomics.  Compiled with -ffp-contract=off (see glam_math.h).
#include <stdlib.h>

#include <algorithm>

#include "glam_math.h"
#include "kernels.h"
#include "visibility_rule.h"
#include "cluster_fill.h"
#include "compact_fast.h"
#include "cluster_walk.h"

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
__device__ __forceinline__ Affine ld_affine(const float* g, uint32_t row) {
    const float4* p = reinterpret_cast<const float4*>(g) + 3ull * row;
    const float4 a = p[0], b = p[1], c = p[2];
    Affine r;
    r.m.x_axis = V3{a.x, a.y, a.z};
    r.m.y_axis = V3{a.w, b.x, b.y};
    r.m.z_axis = V3{b.z, b.w, c.x};
    r.t = V3{c.y, c.z, c.w};
    return r;
}
__device__ __forceinline__ void st_affine(float* g, uint32_t row, const Affine& a) {
    float4* p = reinterpret_cast<float4*>(g) + 3ull * row;
    p[0] = make_float4(a.m.x_axis.x, a.m.x_axis.y, a.m.x_axis.z, a.m.y_axis.x);
    p[1] = make_float4(a.m.y_axis.y, a.m.y_axis.z, a.m.z_axis.x, a.m.z_axis.y);
    p[2] = make_float4(a.m.z_axis.z, a.t.x, a.t.y, a.t.z);
}

// The wave's RowSummary (kernels.h): 32 bytes at a wave-uniform address, fetched with scalar loads.  bits = word 7, or 0 when the
// summary is not in use / the wave has no live row.
struct RowSum {
    V3 center, half;
    uint32_t layers, bits;
};
__device__ __forceinline__ RowSum load_row_summary(const Columns& c, uint32_t wave_row0) {
    // no branch around the load (its results would have to be merged at the join, i.e. waited for there): the address is always
    // valid -- row_summary is allocated whenever a frame kernel runs -- and what is not to be used is masked out of the bits
    const bool use = c.row_summary_on != 0u && wave_row0 < c.n;
    const uint32_t w = __builtin_amdgcn_readfirstlane(use ? wave_row0 >> 6 : 0u);
    const uint4* p = reinterpret_cast<const uint4*>(c.row_summary) + 2ull * w;
    const uint4 a = p[0], b = p[1];
    RowSum r;
    r.center = V3{__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z)};
    r.half = V3{__uint_as_float(a.w), __uint_as_float(b.x), __uint_as_float(b.y)};
    r.layers = b.z;
    r.bits = use ? b.w : 0u;
    return r;
}

// ---------------------------------------------------------------------------------------------
// Wave-local transposes through LDS.  A row's GlobalTransform is 48 bytes; one lane per row means
// a lane-major float4 access touches a 3 KB span per wave-instruction and uses a third of every
// cache line it opens.  Staging the wave's 64 x 48 B = 3 KB in LDS turns the global side into three
// fully contiguous 1 KB wave-instructions (lane i <-> float4 i).  LDS side: ds_write_b128 at a 48-byte
// lane stride and ds_read_b128 at a 16-byte lane stride are both bank-conflict free (8-lane groups cover
// 32 distinct banks).  Each wave only touches its own 3 KB, so a wave-level LDS fence orders it
// (a workgroup barrier here made every wave wait for the slowest of four sets of loads).
// ---------------------------------------------------------------------------------------------
typedef float v4f __attribute__((ext_vector_type(4)));
// A row's affine in a wave's transpose buffer (three float4 per row, the layout of the GlobalTransform column).
__device__ __forceinline__ Affine lds_affine(const float4* slots, uint32_t slot) {
    const float4 a = slots[slot * 3u], b = slots[slot * 3u + 1u], cc = slots[slot * 3u + 2u];
    Affine r;
    r.m.x_axis = V3{a.x, a.y, a.z};
    r.m.y_axis = V3{a.w, b.x, b.y};
    r.m.z_axis = V3{b.z, b.w, cc.x};
    r.t = V3{cc.y, cc.z, cc.w};
    return r;
}
__device__ __forceinline__ void lds_put(float4* slots, uint32_t slot, const Affine& a) {
    slots[slot * 3u] = make_float4(a.m.x_axis.x, a.m.x_axis.y, a.m.x_axis.z, a.m.y_axis.x);
    slots[slot * 3u + 1u] = make_float4(a.m.y_axis.y, a.m.y_axis.z, a.m.z_axis.x, a.m.z_axis.y);
    slots[slot * 3u + 2u] = make_float4(a.m.z_axis.z, a.t.x, a.t.y, a.t.z);
}
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float shfl_f(float v, uint32_t src_lane) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), __float_as_int(v)));
}
__device__ __forceinline__ V3 shfl3(V3 v, uint32_t src_lane) { return V3{shfl_f(v.x, src_lane), shfl_f(v.y, src_lane), shfl_f(v.z, src_lane)}; }
__device__ __forceinline__ void store_affine_coalesced(float4* lds_wave, float* g, uint32_t wave_row0, uint32_t n,
                                                       uint32_t lane, const Affine& a, bool nt = false) {
    lds_wave[lane * 3u + 0u] = make_float4(a.m.x_axis.x, a.m.x_axis.y, a.m.x_axis.z, a.m.y_axis.x);
    lds_wave[lane * 3u + 1u] = make_float4(a.m.y_axis.y, a.m.y_axis.z, a.m.z_axis.x, a.m.z_axis.y);
    lds_wave[lane * 3u + 2u] = make_float4(a.m.z_axis.z, a.t.x, a.t.y, a.t.z);
    MI_WAVE_LDS_SYNC();  // the buffer is this wave's own: no workgroup barrier
    float4* dst = reinterpret_cast<float4*>(g) + 3ull * wave_row0;
    // float4s of live rows in this wave (none for the dead waves of the last workgroup)
    const uint32_t lim = wave_row0 < n ? (n - wave_row0 < 64u ? n - wave_row0 : 64u) * 3u : 0u;
#pragma unroll
    for (uint32_t k = 0; k < 3u; ++k) {
        const uint32_t i = k * 64u + lane;
        if (i < lim) {
            const float4 v = lds_wave[i];
            if (nt) __builtin_nontemporal_store(v4f{v.x, v.y, v.z, v.w}, reinterpret_cast<v4f*>(dst) + i);
            else dst[i] = v;
        }
    }
}
// ---------------------------------------------------------------------------------------------
// What a frame kernel leaves behind per view and per row, shared by k_frame and k_frame_sph.
// emit_view: one view's packed mask word (one wave64 ballot = one 64-bit word, lane 0 stores it) and the per-(view, class)
// wave counts / segment masks the VisibleEntities compaction consumes.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void emit_view(uint32_t v, bool vis, bool any_live, uint32_t lane, uint32_t wave, uint32_t cmask,
                                          const VisibilityOut& out, const SegOut& seg) {
    const unsigned long long m = __ballot(vis);
    if (lane == 0 && any_live) out.bitmask[v * out.words_per_view + out.word_offset + wave] = m;
    if (seg.wave_cnt) {
        if (!seg.class_mask) {
            if (lane == 0 && any_live) seg.wave_cnt[(size_t)v * seg.n_waves + wave] = (uint8_t)__popcll(m);
        } else {
            for (uint32_t k = 0; k < seg.n_classes; ++k) {
                const unsigned long long mk = __ballot(vis && ((cmask >> seg.class_bits[k]) & 1u));
                if (lane == 0 && any_live) {
                    const size_t s = (size_t)v * seg.n_classes + k;
                    seg.seg_mask[s * seg.seg_words + wave] = mk;
                    seg.wave_cnt[s * seg.n_waves + wave] = (uint8_t)__popcll(mk);
                }
            }
        }
    }
}
// The ViewVisibility byte of a row: reset (mod.rs:270-274), set_visible (:290-306), gpu-culling rows (:884-903),
// mark_newly_hidden (:908-918), and the wave's change-tick word.
// Returns ViewVisibility::get() of the row as the frame leaves it.
__device__ __forceinline__ bool view_visibility_tail(const Columns& c, uint32_t row, bool live, bool any_live, uint32_t lane, uint32_t wave,
                                                     uint32_t fl, uint32_t vv0, bool any, uint32_t fl_frame) {
    const bool ncc = (fl & 0x10u) != 0;
    uint32_t cur = vv0;
    bool vv_changed = false;
    if (live) {
        if ((fl_frame & CULL_BEGIN_FRAME) && !ncc) cur = (cur & 1u) << 1;
        if (any && !(cur & 1u)) {
            vv_changed = !(cur & 2u);
            cur |= 1u;
        }
        if (fl_frame & CULL_END_FRAME) {
            if (ncc) {
                const uint32_t nv = (fl & 0x01u) ? 3u : 0u;
                if (nv != cur) { cur = nv; vv_changed = true; }
            } else if ((cur & 3u) == 2u) {
                cur = 0u;
                vv_changed = true;
            }
        }
        if (cur != vv0) c.view_visibility[row] = (uint8_t)cur;
    }
    const unsigned long long chg = __ballot(vv_changed);
    if (lane == 0 && any_live) {
        if (fl_frame & CULL_BEGIN_FRAME) c.vv_changed_bits[wave] = chg;
        else if (chg) atomicOr(reinterpret_cast<unsigned long long*>(&c.vv_changed_bits[wave]), chg);
    }
    return live && (cur & 1u) != 0;
}

// ClusterWalkJob::inrow: the row workgroup of a tile that holds cluster objects goes on into the walk.  The object of a row is
// row - first_row; it takes part if its ViewVisibility::get() is true (the gather's filter, assign.rs:194: `visible`, just computed),
// its centre is the row's GlobalTransform translation (assign.rs:198: `translation`, in hand), and it passes the two early-outs of
// the per-object loop (assign.rs:489 RenderLayers, :496 frustum against the light's sphere).  Every thread of the workgroup calls it.
template <bool SPOTS>
__device__ __forceinline__ void inrow_cluster_walk(const ClusterWalkJob& walk, uint32_t tile, uint32_t row, bool visible, V3 translation, uint32_t* arena,
                                                   const float* planes_arg) {
    const uint32_t bx = tile - walk.tile0;
    const ClusterObjects& o = walk.objs;
    const uint32_t obj = row - o.first_row;  // (wraps for the rows of the first tile that lie in front of the objects)
    const bool is_obj = row >= o.first_row && obj < o.n;
    // (a tile none of whose lights is visible leaving here, in front of the table's and the objects' loads: no different, 20.30 against
    // 20.24 us per frame -- with the table in the argument segment the prologue no longer waits for anything far away)
    const WalkPrefetch pf = walk_prefetch<true>(walk.view, planes_arg);
    float4 sphere = make_float4(translation.x, translation.y, translation.z, 0.f);
    bool in_view = false;
    if (is_obj) {
        sphere.w = o.pos_range[4u * obj + 3u];
        const uint32_t layers = o.layer_mask ? o.layer_mask[obj] : 1u, layers_hi = o.layer_mask_hi ? o.layer_mask_hi[obj] : 0u;
        if (visible && ((walk.view.view_layer_mask & layers) | (walk.view.view_layer_mask_hi & layers_hi))) {
            V4 fr[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) fr[i] = V4{walk.view.frustum[4 * i], walk.view.frustum[4 * i + 1], walk.view.frustum[4 * i + 2], walk.view.frustum[4 * i + 3]};
            in_view = frustum_intersects_sphere(fr, translation, sphere.w, true);
        }
    }
    cluster_walk_tail<true, true, SPOTS>(walk.view, o, walk.w, walk.zc, bx, arena, obj, in_view, sphere, pf);
}
// The extra workgroups at the head of a frame kernel's grid (they overlap the ramp-up instead of lengthening the tail: 0.5 us per
// frame at 1 M rows): the deferred VisibleEntities compaction of the previous frame, the deferred fill of the previous frame's
// light-cluster assignment, and the walk of THIS frame's.  Returns true in a workgroup that was one of them.
#ifdef MI_EXP_TIMELINE
// (experiment build) per workgroup of a frame launch: [3b] start, [3b+1] end (100 MHz wall clock), [3b+2] kind + 1 -- 0 compaction,
// 1 fill, 2 walk, 3 rows.  Plain stores to the workgroup's own slots; mi_exp_timeline() below reads and clears them.
constexpr uint32_t TIMELINE_WGS = 16384;
__device__ unsigned long long mi_timeline[3 * TIMELINE_WGS];
struct TimelineScope {
    uint32_t kind;
    unsigned long long t0;
    __device__ TimelineScope(uint32_t k) : kind(k), t0(wall_clock64()) {}
    __device__ ~TimelineScope() {
        __syncthreads();
        if (threadIdx.x == 0 && blockIdx.x < TIMELINE_WGS) {
            mi_timeline[3 * blockIdx.x] = t0;
            mi_timeline[3 * blockIdx.x + 1] = wall_clock64();
            mi_timeline[3 * blockIdx.x + 2] = kind + 1;
        }
    }
};
#define MI_TIMELINE(kind) TimelineScope tl_scope_(kind)
#else
#define MI_TIMELINE(kind)
#endif
template <int WALK>
__device__ __forceinline__ bool frame_riders(uint32_t n_tiles, const CompactFastArgs& prev, uint32_t prev_gx, uint32_t n_compact, uint32_t n_fill,
                                             const ClusterFillJob& fill, const ClusterWalkJob& walk, const ViewSet& vs, uint32_t* lds_raw,
                                             const float* planes_arg = nullptr) {
    const uint32_t n_extra = gridDim.x - n_tiles;
    if (blockIdx.x >= n_extra) return false;
    const uint32_t id = blockIdx.x;
    if (id < n_compact) {
        if (prev.signal && id == 0 && threadIdx.x == 0)  // multi-GPU exchange: the previous frame's masks are complete
            __hip_atomic_store(prev.signal, prev.signal_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        MI_TIMELINE(0);
        compact_fast_block<false>(prev, id % prev_gx, id / prev_gx, prev_gx);
    } else if (id < n_compact + n_fill) {
        MI_TIMELINE(1);
        cluster_fill_block(fill.w, fill.n_clusters, fill.n_objects, id - n_compact, n_fill, lds_raw, lds_raw + 4096);
    } else if constexpr (WALK != 0) {
        // this frame's light-cluster walk: independent of the rows below (it re-derives the lights' ViewVisibility itself)
        MI_TIMELINE(2);
        cluster_walk_block<true, true, WALK == 2>(walk.view, walk.objs, walk.w, vs, walk.zc, id - n_compact - n_fill, lds_raw, planes_arg);
    }
    return true;
}

// ---------------------------------------------------------------------------------------------
// The frame kernel.  PROPAGATE = true : G = From(T) for every row (sync_simple_transforms, all dirty),
//                                       written once and never re-read (fused flat path);
//                    PROPAGATE = false: G is read from the resident column (after mi_propagate).
// Both: optional reset_view_visibility (CULL_BEGIN_FRAME), check_visibility_cpu_culling against all views,
// optional check_visibility_gpu_culling + mark_newly_hidden_entities_invisible (CULL_END_FRAME), the packed
// per-view bitmasks (one wave64 ballot = one 64-bit word, no atomics) and the per-(view,class) wave counts /
// segment masks the VisibleEntities compaction consumes.
// Algorithmic bytes per row, fused, V views: read 40 (T) + 24 (Aabb) + 1 (flags) + 4 (layers) + 1 (vv),
// write 48 (G) + 1 (vv) + (V + 2 change masks) / 8 + V / 64 (wave counts).
// ---------------------------------------------------------------------------------------------
// WALK: 0 = none; 1 = the launch may carry this frame's light-cluster walk; 2 = ... and there are spot lights among the objects (the
// walk then also runs the cone test against the clusters' bounding spheres, assign.rs:681-738: five more registers per object, so
// scenes without spot lights keep the variant without it).  A variant of its own because the walk needs more registers than
// a row tile (87 against 66 VGPRs: 5 instead of 7 waves per SIMD for the whole launch; capping it at 80 or 72 registers with
// __launch_bounds__ spills and measured 26.9 / 29.0 us per metric frame against 25.2); frames without a walk keep the lean one.
// PROP: 0 = GlobalTransform is resident (mi_cull), 1 = every row is propagated (the fused frame: Transform read once,
// GlobalTransform written once and never re-read), 2 = only rows whose Transform change byte is set are propagated
// (sync_simple_transforms' own filter, systems.rs:45-50; `changed` is the byte column), the others keep the resident value.
// MULTI (2..MULTI_MAX_VIEWS camera views, no riding walk): the intersects_obb half of the rule -- 29 of a
// view's ~45 vector instructions per plane -- runs ONCE over the wave's (row, view) pairs that survived their sphere test instead of
// once per view over all 64 lanes.  A view sees a fraction of a wave's rows (a 74-degree frustum: a fifth of the rows of a wave of
// many_cubes), so four views leave about as many pairs as the wave has lanes: one pass instead of four.  With four views the kernel
// is bound by instruction issue as much as by HBM (10 M rows: 178 us against 152 with one view).  Pairs are queued in LDS view by
// view (ballot + mbcnt), a pair lane takes its row's GlobalTransform from the wave's transpose buffer (still in LDS), the row's centre
// and half extents from the row's lane (ds_bpermute) and its view's planes from a table the workgroup wrote at its start; it ORs
// its verdict into the row's word.  Same operations on the same values per pair as row_visible_in_view: same bits.
constexpr uint32_t MULTI_MAX_VIEWS = 4, MULTI_LDS_TABLE = 3072u, MULTI_LDS_WAVE = 3200u;  // (words of lds_raw: behind the transposes)
static_assert(MULTI_LDS_WAVE + 4u * 128u <= FRAME_LDS_WORDS && MULTI_LDS_TABLE + MULTI_MAX_VIEWS * 20u <= MULTI_LDS_WAVE, "multi-view scratch fits behind the transposes");
template <int PROP, bool INLINE_VIEWS, int WALK, bool MULTI>
__device__ __forceinline__ void frame_workgroup(const Columns& c, const ViewSet& vs, const ViewParams* __restrict__ dviews, uint32_t n_views,
                                                const VisibilityOut& out, const SegOut& seg, uint32_t fl_frame, uint32_t n_tiles,
                                                const CompactFastArgs& prev, uint32_t prev_gx, uint32_t n_compact, uint32_t n_fill,
                                                const ClusterFillJob& fill, const ClusterWalkJob& walk, const uint8_t* __restrict__ changed,
                                                uint32_t wp_offset = 0) {
    constexpr bool PROPAGATE = PROP == 1, PARTIAL = PROP == 2;
    // the riding walk's plane table: the kernel's trailing argument (WalkPlanes, kernels.h), read where the walkers want it
    const float* planes_arg = nullptr;
    if constexpr (WALK != 0) planes_arg = &kernarg_late<float>(wp_offset);
    // 16 KB + 16 B: the four waves' GlobalTransform transpose buffers (12 KB) -- or, in a riding compaction / cluster-fill workgroup,
    // that rider's arena (the CSR offsets of up to 4096 clusters and the four wave totals of their scan).  A launch that carries the
    // cluster walk gives its walking workgroups 22 KB (FRAME_WALK_LDS_WORDS, kernels.h): the walk sweeps the cluster grid in chunks of
    // as many z slices as fit, and a chunk is a reservation round trip -- but every row workgroup of the launch pays for the arena
    // and the walk's registers in occupancy, and there are a hundred of those to a walker (round 6: 31 KB / 89 VGPRs / 5 waves per
    // SIMD -> 22 KB / 72 / 7: metric frame 21.8 -> 19.5 us per step, profiles/r06_experiments.md)
    __shared__ __attribute__((aligned(16))) uint32_t lds_raw[(WALK ? FRAME_WALK_LDS_WORDS : FRAME_LDS_WORDS) + 4];
    float4 (*lds_g)[192] = reinterpret_cast<float4 (*)[192]>(lds_raw);
    if (frame_riders<WALK>(n_tiles, prev, prev_gx, n_compact, n_fill, fill, walk, vs, lds_raw, planes_arg)) return;
    MI_TIMELINE(3);
    const uint32_t n_extra = gridDim.x - n_tiles;
    uint32_t tile = blockIdx.x - n_extra;
    if constexpr (WALK != 0) {  // the tiles that go on into the cluster walk are handed out first (they run longest)
        if (walk.inrow) {
            tile += walk.tile0;
            if (tile >= n_tiles) tile -= n_tiles;
        }
    }
    const uint32_t row = tile * 256u + threadIdx.x;
    const bool live = row < c.n;
    const uint32_t wave = row >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = threadIdx.x >> 6;
    const uint32_t wave_row0 = row & ~63u;

    Affine g = {};
    V3 center = {}, half = {};
    uint32_t fl = 0, emask = 0, emask_hi = 0, vv0 = 0, cmask = 1u;
    float range_lo = 0.0f, range_hi = 0.0f;
    // G resident: its three contiguous wave rows are requested first, so that the transpose only waits for them while the
    // bounds / flags / layers loads issued behind them are still in flight
    float4 gl[3];
    if (!PROPAGATE) {
        const float4* src = reinterpret_cast<const float4*>(c.global) + 3ull * wave_row0;
        const uint32_t lim = wave_row0 < c.n ? (c.n - wave_row0 < 64u ? c.n - wave_row0 : 64u) * 3u : 0u;
#pragma unroll
        for (uint32_t k = 0; k < 3u; ++k) {
            const uint32_t i = k * 64u + lane;
            gl[k] = i < lim ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    bool dirty = false;
    if (PARTIAL && live) dirty = row_changed(changed[row], c.changed_gen);
    // the wave's summary is requested first and waited for last: the loads every live row issues anyway go out in between
    const RowSum rs = load_row_summary(c, wave_row0);
    V3 t_in = {}, s_in = {};
    V4 q_in = {};
    if (PROPAGATE && live) {
        if (fl_frame & CULL_NT_LOADS) {  // (launch-uniform: a big context, kernels.h)
            const float* tp = c.translation + 3ull * row;
            const float* sp3 = c.scale + 3ull * row;
            const v4f qv = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(c.rotation) + row);
            t_in = V3{__builtin_nontemporal_load(tp), __builtin_nontemporal_load(tp + 1), __builtin_nontemporal_load(tp + 2)};
            q_in = V4{qv.x, qv.y, qv.z, qv.w};
            s_in = V3{__builtin_nontemporal_load(sp3), __builtin_nontemporal_load(sp3 + 1), __builtin_nontemporal_load(sp3 + 2)};
        } else {
            t_in = ld3(c.translation, row);
            q_in = ld4(c.rotation, row);
            s_in = ld3(c.scale, row);
        }
    }
    if (live) {
        vv0 = c.view_visibility[row];
        if (seg.class_mask) cmask = seg.class_mask[row];
    }
    const bool uni_aabb = (rs.bits & ROWSUM_UNIFORM_AABB) != 0u, uni_fl = (rs.bits & ROWSUM_UNIFORM_FLAGS) != 0u;  // (wave-uniform)
    if (uni_aabb) {
        center = rs.center;
        half = rs.half;
    } else if (live) {
        center = ld3(c.aabb_center, row);
        half = ld3(c.aabb_half, row);
    }
    if (uni_fl) {
        fl = live ? (rs.bits & 0xFFu) : 0u;
        emask = rs.layers;
#ifdef MI_EXP_MUTANT  // (a deliberately wrong build: tests/test_gpu_differential.py must notice, tools/gpu_mutant.sh)
        if ((row & 0xFFFu) == 0x123u) emask = 0u;
#endif
    } else if (live) {
        fl = c.flags[row];
        emask = c.layer_mask[row];
        if (c.layer_mask_hi) emask_hi = c.layer_mask_hi[row];
    }
    if (live) {
        if (c.range_start_end && (fl & 0x20u)) {
            const float2 r2 = reinterpret_cast<const float2*>(c.range_start_end)[row];
            range_lo = r2.x;
            range_hi = r2.y;
        }
    }
    if constexpr (MULTI) {
        // the views' planes where a lane can index them: wave w writes view w's five (under the row loads just issued)
        const uint32_t u = __builtin_amdgcn_readfirstlane(wv);
        if (u < n_views) {
            const ViewParams& vp = vs.v[u];
            float4* tbl = reinterpret_cast<float4*>(lds_raw + MULTI_LDS_TABLE) + u * 5u;
            // planes 0/1 and 2/3 interleaved component by component -- (n0.x, n1.x, n0.y, n1.y), (n0.z, n1.z, d0, d1) -- so that a pair
            // lane reads them straight into the register pairs of the packed FP32 instructions; plane 4 as it is
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float* a = vp.planes + 8 * i;
                const float4 lo = make_float4(a[0], a[4], a[1], a[5]), hi = make_float4(a[2], a[6], a[3], a[7]);
                if (lane == 0) {
                    tbl[2 * i] = lo;
                    tbl[2 * i + 1] = hi;
                }
            }
            const float4 last = make_float4(vp.planes[16], vp.planes[17], vp.planes[18], vp.planes[19]);
            if (lane == 0) tbl[4] = last;
        }
        MI_WG_LDS_BARRIER();
    }
    if (PROPAGATE) {
        if (live) g = affine_from_srt(s_in, q_in, t_in);
        // nontemporal: the fused path never reads G back (measured +3..16 % at 4 M - 10 M rows, neutral at 1 M)
#ifdef MI_EXP_STRIP_GSTORE
        if (g.t.x == 1.2345e30f)
#endif
        store_affine_coalesced(lds_g[wv], c.global, wave_row0, c.n, lane, g, true);
    } else {
        float4* lds_wave = lds_g[wv];
#pragma unroll
        for (uint32_t k = 0; k < 3u; ++k) lds_wave[k * 64u + lane] = gl[k];
        MI_WAVE_LDS_SYNC();
        const float4 a = lds_wave[lane * 3u], b = lds_wave[lane * 3u + 1u], cc = lds_wave[lane * 3u + 2u];
        g.m.x_axis = V3{a.x, a.y, a.z};
        g.m.y_axis = V3{a.w, b.x, b.y};
        g.m.z_axis = V3{b.z, b.w, cc.x};
        g.t = V3{cc.y, cc.z, cc.w};
        if (PARTIAL && dirty) {  // few rows of a wave as a rule: each stores its own 48 bytes
            const V3 t = ld3(c.translation, row);
            const V4 q = ld4(c.rotation, row);
            const V3 sc = ld3(c.scale, row);
            g = affine_from_srt(sc, q, t);
            st_affine(c.global, row, g);
            if constexpr (MULTI) lds_put(lds_wave, lane, g);  // (the pair lanes below read the rows' GlobalTransforms from here)
        }
    }
    const bool ncc = (fl & 0x10u) != 0;  // NoCpuCulling rows are not in the cull query (mod.rs:771)
    const bool any_live = wave_row0 < c.n;  // waves past the last row must not touch the masks
    bool any = false;
    if constexpr (MULTI) {
        // ---- every view's cheap half per row: InheritedVisibility, RenderLayers, the five-plane sphere test (mod.rs:797-832);
        // rows with an Aabb that pass it queue up for intersects_obb (:833-836)
        const bool has_aabb = (fl & 0x04u) != 0, bounded = (fl & (0x04u | 0x08u)) != 0;
        const bool at_translation = !has_aabb && __float_as_uint(half.y) == SPHERE_AT_TRANSLATION;
        const V3 cw = has_aabb ? transform_point(g, center) : V3{at_translation ? g.t.x : center.x, at_translation ? g.t.y : center.y, at_translation ? g.t.z : center.z};
        const float sr = has_aabb ? length3(mul(g.m, half)) : half.x;
        const V4 c4 = extend(cw, 1.0f);
        uint8_t* const queue = reinterpret_cast<uint8_t*>(lds_raw + MULTI_LDS_WAVE + wv * 128u);
        uint32_t* const verdicts = lds_raw + MULTI_LDS_WAVE + wv * 128u + 64u;
        verdicts[lane] = 0u;
        uint32_t visbits = 0u, n_pairs = 0u;
        const bool ranged = (fl & 0x20u) != 0 && c.range_start_end != nullptr;
        for (uint32_t v = 0; v < n_views; ++v) {
            const ViewParams& vp = vs.v[v];
            bool vis = live && !ncc && (fl & 0x01u) != 0 && ((vp.layer_mask & emask) | (vp.layer_mask_hi & emask_hi)) != 0;
            if (ranged) {  // Has<VisibilityRange> && VisibleEntityRanges exists (row_visible_in_view's rule, visibility_rule.h)
                bool in_range = false;
                if ((vp.flags & (VIEW_RANGES | VIEW_RANGES_NO_ORIGIN)) == VIEW_RANGES) {
                    const V3 model = ((fl & 0x40u) && has_aabb) ? cw : g.t;
                    const float d = length3(V3{vp.position[0], vp.position[1], vp.position[2]} - model);
                    in_range = d >= range_lo && d < range_hi;
                }
                vis = vis && in_range;
            }
            const bool cull = !(fl & 0x02u) && !(vp.flags & VIEW_NO_CPU_CULLING) && bounded;
            bool inside = true;
#ifndef MI_EXP_MV_NOTEST
            if (cull) inside = sphere_inside_five_planes(vp.planes, c4, sr);
#endif
#ifdef MI_EXP_MV_NOOBB
            const bool cand = false;
#else
            const bool cand = vis && cull && inside && has_aabb;
#endif
            vis = vis && (!cull || (inside && !has_aabb));
            const unsigned long long m = __ballot(cand);
            if (m) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (cand) queue[n_pairs + rank] = (uint8_t)(lane | (v << 6));
                n_pairs += (uint32_t)__popcll(m);
            }
            visbits |= (vis ? 1u : 0u) << v;
        }
        if (n_pairs) {  // (wave-uniform)
            MI_WAVE_LDS_SYNC();
            const float4* const lds_wave = lds_g[wv];
            const float4* const tbl = reinterpret_cast<const float4*>(lds_raw + MULTI_LDS_TABLE);
            for (uint32_t p0 = 0; p0 < n_pairs; p0 += 64u) {
                const bool on = p0 + lane < n_pairs;
                const uint32_t e = queue[on ? p0 + lane : 0u], src = e & 63u, pv = e >> 6;
                const Affine gs = lds_affine(lds_wave, src);
                const V3 cws = shfl3(cw, src), halfs = shfl3(half, src);
                const V4 c4s = extend(cws, 1.0f);
                bool in_obb = true;
                // intersects_obb.  -DMI_EXP_MV_PK: two planes to an instruction (v_pk_mul_f32 / v_pk_add_f32; the operations and their order per
                // plane are dot4's and aabb_relative_radius's, so are the bits): 106 instead of 165 vector instructions per pass -- and
                // no faster on any size (1.25 M x 4 views 22.7 against 22.5 us, 10 M 173.7 against 172.4): with the pairs out of the way
                // the kernel is no longer bound by instruction issue (profiles/r05a/multi_view_ab.txt)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const float4 lo = tbl[pv * 5u + 2u * i], hi = tbl[pv * 5u + 2u * i + 1u];
#ifndef MI_EXP_MV_PK  // (the product: plain FP32 instructions; the packed form below is the A/B build)
                    const V4 pa = V4{lo.x, lo.z, hi.x, hi.z}, pb = V4{lo.y, lo.w, hi.y, hi.w};
                    in_obb = in_obb & !(dot4(pa, c4s) + aabb_relative_radius(halfs, xyz(pa), gs.m) <= 0.0f);
                    in_obb = in_obb & !(dot4(pb, c4s) + aabb_relative_radius(halfs, xyz(pb), gs.m) <= 0.0f);
                    continue;
#endif
                    const f2 nx = f2{lo.x, lo.y}, ny = f2{lo.z, lo.w}, nz = f2{hi.x, hi.y}, d = f2{hi.z, hi.w};
                    const f2 dist = (nx * c4s.x + nz * c4s.z) + (ny * c4s.y + d * c4s.w);
                    const f2 vx = (nx * gs.m.x_axis.x + ny * gs.m.x_axis.y) + nz * gs.m.x_axis.z;
                    const f2 vy = (nx * gs.m.y_axis.x + ny * gs.m.y_axis.y) + nz * gs.m.y_axis.z;
                    const f2 vz = (nx * gs.m.z_axis.x + ny * gs.m.z_axis.y) + nz * gs.m.z_axis.z;
                    const f2 rr = (__builtin_elementwise_abs(vx) * halfs.x + __builtin_elementwise_abs(vy) * halfs.y) + __builtin_elementwise_abs(vz) * halfs.z;
                    const f2 sum = dist + rr;
                    in_obb = in_obb & !(sum.x <= 0.0f) & !(sum.y <= 0.0f);
                }
                {
                    const float4 p4 = tbl[pv * 5u + 4u];
                    const V4 pl = V4{p4.x, p4.y, p4.z, p4.w};
                    in_obb = in_obb & !(dot4(pl, c4s) + aabb_relative_radius(halfs, xyz(pl), gs.m) <= 0.0f);
                }
                if (on && in_obb) __hip_atomic_fetch_or(&verdicts[src], 1u << pv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
            MI_WAVE_LDS_SYNC();
            visbits |= verdicts[lane];
        }
        for (uint32_t v = 0; v < n_views; ++v) {
            const bool vis = ((visbits >> v) & 1u) != 0;
            any = any || vis;
#ifdef MI_EXP_MV_NOEMIT
            if (v == 0)
#endif
            emit_view(v, vis, any_live, lane, wave, cmask, out, seg);
        }
    } else {
        for (uint32_t v = 0; v < n_views; ++v) {
            const ViewParams& vp = INLINE_VIEWS ? vs.v[v] : dviews[v];
#ifdef MI_EXP_STRIP_VIEWS  // (timing-only builds, wrong results: what each part of the row path costs, profiles/r06_experiments.md)
            const bool vis = live && !ncc && (fl & 1u) && g.t.z < vp.planes[3];
#else
            const bool vis = live && !ncc &&
                             row_visible_in_view(g, center, half, fl, emask, emask_hi, c.range_start_end != nullptr, range_lo, range_hi, vp);
#endif
            any = any || vis;
#ifndef MI_EXP_STRIP_EMIT
            emit_view(v, vis, any_live, lane, wave, cmask, out, seg);
#endif
        }
    }
#ifdef MI_EXP_STRIP_TAIL
    const bool vv_now = any;
    if (live && any && vv0 == 77u) c.view_visibility[row] = 1;
#else
    const bool vv_now = view_visibility_tail(c, row, live, any_live, lane, wave, fl, vv0, any, fl_frame);
    if (PROPAGATE) {  // plain assignment bumps every written row's tick (systems.rs:62)
        const unsigned long long lv = __ballot(live);
        if (lane == 0 && any_live) c.g_changed_bits[wave] = lv;
    }
#endif
    if (PARTIAL) {
        const unsigned long long dm = __ballot(dirty);
        if (lane == 0 && any_live) c.g_changed_bits[wave] = dm;
    }
    if constexpr (WALK != 0) {
        // (reading the walk job here instead of at kernel entry -- kernarg_late, kernels.h: 14 -> 11 spilled SGPRs, no scratch -- made the
        // metric frame SLOWER, 21.1 -> 22.2 us: the walk's tail is a chain of round trips, and the job's scalar loads join it)
        if (walk.inrow && tile - walk.tile0 < walk.n_blocks) inrow_cluster_walk<WALK == 2>(walk, tile, row, vv_now, g.t, lds_raw, planes_arg);  // (workgroup-uniform)
    }
}

// (the kernels' argument lists as structs: where the trailing WalkPlanes sits in the argument segment)
struct FrameKernargs {
    Columns c; ViewSet vs; const ViewParams* dviews; uint32_t n_views; VisibilityOut out; SegOut seg; uint32_t fl_frame, n_tiles;
    CompactFastArgs prev; uint32_t prev_gx, n_compact, n_fill; ClusterFillJob fill; ClusterWalkJob walk; const uint8_t* changed; WalkPlanes wp;
};
template <int PROP, bool INLINE_VIEWS, int WALK>
__global__ void __launch_bounds__(256, WALK == 1 ? FRAME_WALK_WAVES : 1) k_frame(Columns c, ViewSet vs, const ViewParams* __restrict__ dviews,
                                                uint32_t n_views, VisibilityOut out, SegOut seg, uint32_t fl_frame, uint32_t n_tiles,
                                                CompactFastArgs prev, uint32_t prev_gx, uint32_t n_compact, uint32_t n_fill, ClusterFillJob fill,
                                                ClusterWalkJob walk, const uint8_t* __restrict__ changed, typename WalkPlanesArg<WALK>::type wp) {
    frame_workgroup<PROP, INLINE_VIEWS, WALK, false>(c, vs, dviews, n_views, out, seg, fl_frame, n_tiles, prev, prev_gx, n_compact, n_fill, fill, walk, changed,
                                                     (uint32_t)offsetof(FrameKernargs, wp));
}
// ... with 2 .. MULTI_MAX_VIEWS camera views: the pair pass (MULTI above).  A kernel name of its own, so that k_frame<...> keeps the
// symbols the committed profiles and the bench's kernel filters know.
template <int PROP>
__global__ void __launch_bounds__(256, 8) k_frame_pairs(Columns c, ViewSet vs, const ViewParams* __restrict__ dviews,
                                                      uint32_t n_views, VisibilityOut out, SegOut seg, uint32_t fl_frame, uint32_t n_tiles,
                                                      CompactFastArgs prev, uint32_t prev_gx, uint32_t n_compact, uint32_t n_fill, ClusterFillJob fill,
                                                      ClusterWalkJob walk, const uint8_t* __restrict__ changed) {
    frame_workgroup<PROP, true, 0, true>(c, vs, dviews, n_views, out, seg, fl_frame, n_tiles, prev, prev_gx, n_compact, n_fill, fill, walk, changed);
}

// ---------------------------------------------------------------------------------------------
// The frame kernel over the WORLD-SPHERE column -- the frame a game mostly runs: few or no Transforms moved.
//
// check_visibility_cpu_culling tests a row's bounding sphere first (visibility/mod.rs:824-832): centre = affine * aabb.center,
// radius = |M3 * half_extents| -- values that change only when the row's GlobalTransform (or Aabb) does.  The context keeps them as
// a 16-byte column (cw.xyz, sr), the same bits the reference computes, and this kernel runs the five-plane sphere test of every
// view on that column alone: 16 (sphere) + 1 (flags) + 4 (layers) + 1 (vv) = 22 B per row instead of 48 (G) + 24 (Aabb) + 6.
// GlobalTransform and half extents are fetched only by the lanes that survive a sphere test and have an Aabb (intersects_obb,
// :833-836: the OBB test shares the centre and the plane dot products, so only the relative radius is new).
// Rows whose sphere is STALE -- their GlobalTransform changed since the column was written: the change mask of the last
// propagate, as ballot words (flat path) or bytes (hierarchy path), or every row after something wholesale -- are refreshed
// first, from the resident GlobalTransform (a wave that is stale as a whole takes the coalesced transpose).
// PARTIAL (MI_CULL_CHANGED_ROWS): the rows whose Transform change byte is set are propagated here as well, exactly like
// k_frame<2> (sync_simple_transforms' filter, systems.rs:45-50): G = From(T), stored, sphere refreshed, change word written.
// Camera views only (the shadow-view kinds have no sphere pre-test): the host sends frames with shadow views to k_frame.
// ---------------------------------------------------------------------------------------------
struct SphereArgs {
    float4* sph;                  // [n] (cw.x, cw.y, cw.z, sr)
    const uint64_t* stale_bits;   // ballot words of the rows whose GlobalTransform changed since the column was current, or ...
    const uint8_t* stale_bytes;   // ... a byte per row (hierarchy path); both nullptr = none
    uint32_t all_stale;           // every row (first use, or after an all-dirty propagate / a bounds upload)
};
template <bool PARTIAL, bool INLINE_VIEWS, int WALK, bool MULTI>
__device__ __forceinline__ void frame_sph_workgroup(const Columns& c, const ViewSet& vs, const ViewParams* __restrict__ dviews, uint32_t n_views,
                                                    const VisibilityOut& out, const SegOut& seg, uint32_t fl_frame, uint32_t n_tiles,
                                                    const CompactFastArgs& prev, uint32_t prev_gx, uint32_t n_compact, uint32_t n_fill,
                                                    const ClusterFillJob& fill, const ClusterWalkJob& walk, const uint8_t* __restrict__ changed,
                                                    const SphereArgs& sa, uint32_t wp_offset = 0) {
    __shared__ __attribute__((aligned(16))) uint32_t lds_raw[(WALK ? FRAME_WALK_LDS_WORDS : FRAME_LDS_WORDS) + 4];  // (as in k_frame)
    float4 (*lds_g)[192] = reinterpret_cast<float4 (*)[192]>(lds_raw);
    const float* planes_arg = nullptr;  // (as in k_frame: the riding walk's plane table in the argument segment)
    if constexpr (WALK != 0) planes_arg = &kernarg_late<float>(wp_offset);
    if (frame_riders<WALK>(n_tiles, prev, prev_gx, n_compact, n_fill, fill, walk, vs, lds_raw, planes_arg)) return;
    if constexpr (MULTI) {  // the views' planes where a lane can index them (as in k_frame's MULTI): wave w writes view w's five
        const uint32_t u = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        if (u < n_views) {
            const ViewParams& vp = vs.v[u];
            float4* tbl = reinterpret_cast<float4*>(lds_raw + MULTI_LDS_TABLE) + u * 5u;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float* a = vp.planes + 8 * i;
                const float4 lo = make_float4(a[0], a[4], a[1], a[5]), hi = make_float4(a[2], a[6], a[3], a[7]);
                if ((threadIdx.x & 63u) == 0) {
                    tbl[2 * i] = lo;
                    tbl[2 * i + 1] = hi;
                }
            }
            const float4 last = make_float4(vp.planes[16], vp.planes[17], vp.planes[18], vp.planes[19]);
            if ((threadIdx.x & 63u) == 0) tbl[4] = last;
        }
        MI_WG_LDS_BARRIER();
    }
    const uint32_t n_extra = gridDim.x - n_tiles;
    uint32_t tile = blockIdx.x - n_extra;
    if constexpr (WALK != 0) {  // (as in k_frame: the tiles that go on into the cluster walk first)
        if (walk.inrow) {
            tile += walk.tile0;
            if (tile >= n_tiles) tile -= n_tiles;
        }
    }
    const uint32_t row = tile * 256u + threadIdx.x;
    const bool live = row < c.n;
    const uint32_t wave = row >> 6, lane = threadIdx.x & 63u, wv = threadIdx.x >> 6, wave_row0 = row & ~63u;
    const bool any_live = wave_row0 < c.n;
    const uint32_t rrow = live ? row : 0u;  // clamped: every load below is unconditional (one batch, no load behind a branch)

    // ---- burst 1: what every row needs ----
    const RowSum rs = load_row_summary(c, wave_row0);
    const bool uni_aabb = (rs.bits & ROWSUM_UNIFORM_AABB) != 0u, uni_fl = (rs.bits & ROWSUM_UNIFORM_FLAGS) != 0u;  // (wave-uniform)
    float4 sp = sa.sph[rrow];
    const uint32_t vv0 = c.view_visibility[rrow];
    uint32_t cmask = 1u;
    if (seg.class_mask) cmask = seg.class_mask[rrow];
    uint32_t fl = rs.bits & 0xFFu, emask = rs.layers, emask_hi = 0u;  // (the summary is waited for here, behind the loads every row issues)
    if (!uni_fl) {
        fl = c.flags[rrow];
        emask = c.layer_mask[rrow];
        if (c.layer_mask_hi) emask_hi = c.layer_mask_hi[rrow];
    }
    bool dirty = false;
    if (PARTIAL) dirty = live && row_changed(changed[rrow], c.changed_gen);
    bool stale = sa.all_stale != 0u;
    if (sa.stale_bytes) stale = stale || sa.stale_bytes[rrow] != 0;
    if (sa.stale_bits) stale = stale || ((sa.stale_bits[any_live ? wave : 0u] >> lane) & 1ull) != 0ull;
    stale = (stale || dirty) && live;
    if (!live) fl = 0u;
    const bool has_aabb = (fl & 0x04u) != 0;

    // ---- stale spheres: from the row's GlobalTransform (PARTIAL: from its Transform, which also becomes the GlobalTransform) ----
    const unsigned long long stale_m = __ballot(stale), live_m = __ballot(live);
    if (stale_m) {  // (wave-uniform)
        Affine g = {};
        const bool whole_wave = !PARTIAL && stale_m == live_m;
        if (whole_wave) {  // three contiguous 1 KB wave rows through the wave-private transpose, as k_frame<0> reads them
            const float4* src = reinterpret_cast<const float4*>(c.global) + 3ull * wave_row0;
            const uint32_t lim = (c.n - wave_row0 < 64u ? c.n - wave_row0 : 64u) * 3u;
            float4* lds_wave = lds_g[wv];
#pragma unroll
            for (uint32_t k = 0; k < 3u; ++k) {
                const uint32_t i = k * 64u + lane;
                lds_wave[i] = src[i < lim ? i : lim - 1u];
            }
            MI_WAVE_LDS_SYNC();
            const float4 a = lds_wave[lane * 3u], b = lds_wave[lane * 3u + 1u], cc = lds_wave[lane * 3u + 2u];
            g.m.x_axis = V3{a.x, a.y, a.z};
            g.m.y_axis = V3{a.w, b.x, b.y};
            g.m.z_axis = V3{b.z, b.w, cc.x};
            g.t = V3{cc.y, cc.z, cc.w};
        }
        if (stale) {
            if (PARTIAL && dirty) {
                const V3 t = ld3(c.translation, row);
                const V4 q = ld4(c.rotation, row);
                const V3 sc = ld3(c.scale, row);
                g = affine_from_srt(sc, q, t);
                st_affine(c.global, row, g);
            } else if (!whole_wave) {
                g = ld_affine(c.global, row);
            }
            const V3 center = uni_aabb ? rs.center : ld3(c.aabb_center, row), half = uni_aabb ? rs.half : ld3(c.aabb_half, row);
            // exactly the values row_visible_in_view computes (visibility_rule.h): Aabb -> (affine * center, |M3 * half|), a Sphere
            // component as it is
            const bool at_translation = !has_aabb && __float_as_uint(half.y) == SPHERE_AT_TRANSLATION;  // a light's sphere follows its entity
            const V3 cw = has_aabb ? transform_point(g, center) : V3{at_translation ? g.t.x : center.x, at_translation ? g.t.y : center.y, at_translation ? g.t.z : center.z};
            const float sr = has_aabb ? length3(mul(g.m, half)) : half.x;
            sp = make_float4(cw.x, cw.y, cw.z, sr);
            sa.sph[row] = sp;
        }
    }
    if (PARTIAL) {
        const unsigned long long dm = __ballot(dirty);
        if (lane == 0 && any_live) c.g_changed_bits[wave] = dm;
    }

    // ---- stage 1, every view: InheritedVisibility, RenderLayers, VisibilityRange, the five-plane sphere test ----
    const bool ncc = (fl & 0x10u) != 0;                     // NoCpuCulling rows are not in the cull query (mod.rs:771)
    const bool base_ok = live && !ncc && (fl & 0x01u) != 0;  // InheritedVisibility
    const bool bounded = (fl & (0x04u | 0x08u)) != 0 && !(fl & 0x02u);  // has an Aabb or a Sphere, and no NoFrustumCulling
    const bool ranged = (fl & 0x20u) != 0 && c.range_start_end != nullptr;
    float range_lo = 0.0f, range_hi = 0.0f;
    V3 model = {};
    if (__ballot(ranged)) {  // (rare)
        if (ranged) {
            const float2 r2 = reinterpret_cast<const float2*>(c.range_start_end)[row];
            range_lo = r2.x;
            range_hi = r2.y;
            if ((fl & 0x40u) && has_aabb) model = V3{sp.x, sp.y, sp.z};  // transform_point(g, center): the sphere's centre
            else model = ld3(c.global, row * 4u + 3u);                   // g.t: floats 9..11 of the row's 12 = the F3 at index 4 * row + 3
        }
    }
    const V4 c4 = V4{sp.x, sp.y, sp.z, 1.0f};
    const float sr = sp.w;
    uint32_t pass = 0u, need = 0u;  // bit v: the row is (still) visible in view v / it still owes view v the OBB test
    for (uint32_t v = 0; v < n_views; ++v) {
        const ViewParams& vp = INLINE_VIEWS ? vs.v[v] : dviews[v];
        bool vis = base_ok && ((vp.layer_mask & emask) | (vp.layer_mask_hi & emask_hi)) != 0;
        if (ranged) {
            bool in_range = false;
            if ((vp.flags & (VIEW_RANGES | VIEW_RANGES_NO_ORIGIN)) == VIEW_RANGES) {
                const float d = length3(V3{vp.position[0], vp.position[1], vp.position[2]} - model);
                in_range = d >= range_lo && d < range_hi;
            }
            vis = vis && in_range;
        }
        const bool cull = bounded && !(vp.flags & VIEW_NO_CPU_CULLING);
        const bool inside = sphere_inside_five_planes(vp.planes, c4, sr);
        vis = vis && (inside || !cull);
        if (vis) pass |= 1u << v;
        if (vis && cull && has_aabb) need |= 1u << v;
    }
    // ---- stage 2, survivors with an Aabb: intersects_obb -- only here are GlobalTransform and half extents touched ----
    const unsigned long long need_m = __ballot(need != 0u);
    if (need_m) {
        // a lane of its own touches one or two 128-byte lines for its 48 bytes: from a quarter of the wave on, the wave's three
        // contiguous KB through the transpose are fewer bytes (10 M rows x 4 views, a third of the rows owing some view the OBB
        // test: 689 MB per launch with per-lane loads)
        Affine g = {};
        const bool dense = __popcll(need_m) >= 16;
        if (dense) {
            const float4* src = reinterpret_cast<const float4*>(c.global) + 3ull * wave_row0;
            const uint32_t lim = (c.n - wave_row0 < 64u ? c.n - wave_row0 : 64u) * 3u;
            float4* lds_wave = lds_g[wv];
            MI_WAVE_LDS_SYNC();  // (the stale-sphere refresh above may have used the buffer)
#pragma unroll
            for (uint32_t k = 0; k < 3u; ++k) {
                const uint32_t i = k * 64u + lane;
                const float4* pp = src + (i < lim ? i : lim - 1u);
                if (fl_frame & CULL_NT_LOADS) {  // (a big static context: the survivors' GlobalTransforms are read once per frame, past the caches)
                    const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(pp));
                    lds_wave[i] = make_float4(v.x, v.y, v.z, v.w);
                } else {
                    lds_wave[i] = *pp;
                }
            }
            MI_WAVE_LDS_SYNC();
            const float4 a = lds_wave[lane * 3u], b = lds_wave[lane * 3u + 1u], cc = lds_wave[lane * 3u + 2u];
            g.m.x_axis = V3{a.x, a.y, a.z};
            g.m.y_axis = V3{a.w, b.x, b.y};
            g.m.z_axis = V3{b.z, b.w, cc.x};
            g.t = V3{cc.y, cc.z, cc.w};
        }
        if constexpr (MULTI) {
            // intersects_obb once over the wave's (row, view) pairs that owe it, 64 to a pass (k_frame's MULTI): a pair lane takes its
            // row's GlobalTransform from the transpose buffer (dense) or from the column, the row's sphere centre from the row's lane
            uint8_t* const queue = reinterpret_cast<uint8_t*>(lds_raw + MULTI_LDS_WAVE + wv * 128u);
            uint32_t* const failed = lds_raw + MULTI_LDS_WAVE + wv * 128u + 64u;
            failed[lane] = 0u;
            uint32_t n_pairs = 0u;
            for (uint32_t v = 0; v < n_views; ++v) {
                const bool cand = ((need >> v) & 1u) != 0;
                const unsigned long long m = __ballot(cand);
                if (m) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    if (cand) queue[n_pairs + rank] = (uint8_t)(lane | (v << 6));
                    n_pairs += (uint32_t)__popcll(m);
                }
            }
            MI_WAVE_LDS_SYNC();
            const float4* const lds_wave = lds_g[wv];
            const float4* const tbl = reinterpret_cast<const float4*>(lds_raw + MULTI_LDS_TABLE);
            for (uint32_t p0 = 0; p0 < n_pairs; p0 += 64u) {
                const bool on = p0 + lane < n_pairs;
                const uint32_t e = queue[on ? p0 + lane : 0u], src = e & 63u, pv = e >> 6;
                const uint32_t srow = wave_row0 + src;  // (a live row: it queued itself)
                const Affine gs = dense ? lds_affine(lds_wave, src) : ld_affine(c.global, srow);
                const V3 halfs = uni_aabb ? rs.half : ld3(c.aabb_half, srow);
                const V4 c4s = V4{shfl_f(sp.x, src), shfl_f(sp.y, src), shfl_f(sp.z, src), 1.0f};
                bool in_obb = true;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const float4 lo = tbl[pv * 5u + 2u * i], hi = tbl[pv * 5u + 2u * i + 1u];
#ifndef MI_EXP_MV_PK  // (the product: plain FP32; packed measured no faster here either -- 10 M x 4 views 155.3 against 152.9 us per frame)
                    const V4 pa = V4{lo.x, lo.z, hi.x, hi.z}, pb = V4{lo.y, lo.w, hi.y, hi.w};
                    in_obb = in_obb & !(dot4(pa, c4s) + aabb_relative_radius(halfs, xyz(pa), gs.m) <= 0.0f);
                    in_obb = in_obb & !(dot4(pb, c4s) + aabb_relative_radius(halfs, xyz(pb), gs.m) <= 0.0f);
                    continue;
#endif
                    // (-DMI_EXP_MV_PK) two planes to an instruction, v_pk_mul_f32 / v_pk_add_f32: same operations in the same order, same bits
                    const f2 nx = f2{lo.x, lo.y}, ny = f2{lo.z, lo.w}, nz = f2{hi.x, hi.y}, d = f2{hi.z, hi.w};
                    const f2 dist = (nx * c4s.x + nz * c4s.z) + (ny * c4s.y + d * c4s.w);
                    const f2 vx = (nx * gs.m.x_axis.x + ny * gs.m.x_axis.y) + nz * gs.m.x_axis.z;
                    const f2 vy = (nx * gs.m.y_axis.x + ny * gs.m.y_axis.y) + nz * gs.m.y_axis.z;
                    const f2 vz = (nx * gs.m.z_axis.x + ny * gs.m.z_axis.y) + nz * gs.m.z_axis.z;
                    const f2 rr = (__builtin_elementwise_abs(vx) * halfs.x + __builtin_elementwise_abs(vy) * halfs.y) + __builtin_elementwise_abs(vz) * halfs.z;
                    const f2 sum = dist + rr;
                    in_obb = in_obb & !(sum.x <= 0.0f) & !(sum.y <= 0.0f);
                }
                {
                    const float4 p4 = tbl[pv * 5u + 4u];
                    const V4 pl = V4{p4.x, p4.y, p4.z, p4.w};
                    in_obb = in_obb & !(dot4(pl, c4s) + aabb_relative_radius(halfs, xyz(pl), gs.m) <= 0.0f);
                }
                if (on && !in_obb) __hip_atomic_fetch_or(&failed[src], 1u << pv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
            MI_WAVE_LDS_SYNC();
            pass &= ~failed[lane];
        } else if (need) {
            if (!dense) g = ld_affine(c.global, row);
            const V3 half = uni_aabb ? rs.half : ld3(c.aabb_half, row);
            for (uint32_t v = 0; v < n_views; ++v) {
                if (!((need >> v) & 1u)) continue;
                const ViewParams& vp = INLINE_VIEWS ? vs.v[v] : dviews[v];
                bool inside = true;
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    const V4 pl = V4{vp.planes[4 * i], vp.planes[4 * i + 1], vp.planes[4 * i + 2], vp.planes[4 * i + 3]};
                    inside = inside && !(dot4(pl, c4) + aabb_relative_radius(half, xyz(pl), g.m) <= 0.0f);
                }
                if (!inside) pass &= ~(1u << v);
            }
        }
    }
    for (uint32_t v = 0; v < n_views; ++v) emit_view(v, ((pass >> v) & 1u) != 0, any_live, lane, wave, cmask, out, seg);
    const bool vv_now = view_visibility_tail(c, row, live, any_live, lane, wave, fl, vv0, pass != 0u, fl_frame);
    if constexpr (WALK != 0) {
        if (walk.inrow && tile - walk.tile0 < walk.n_blocks) {  // (workgroup-uniform)
            // the row's GlobalTransform translation: the column as this launch leaves it (a PARTIAL frame's own stores included)
            const V3 t = live ? ld3(c.global, row * 4u + 3u) : V3{0.f, 0.f, 0.f};
            inrow_cluster_walk<WALK == 2>(walk, tile, row, vv_now, t, lds_raw, planes_arg);
        }
    }
}

struct FrameSphKernargs {
    Columns c; ViewSet vs; const ViewParams* dviews; uint32_t n_views; VisibilityOut out; SegOut seg; uint32_t fl_frame, n_tiles;
    CompactFastArgs prev; uint32_t prev_gx, n_compact, n_fill; ClusterFillJob fill; ClusterWalkJob walk; const uint8_t* changed; SphereArgs sa;
    WalkPlanes wp;
};
template <bool PARTIAL, bool INLINE_VIEWS, int WALK>
__global__ void __launch_bounds__(256, WALK == 1 ? FRAME_WALK_WAVES : 1) k_frame_sph(Columns c, ViewSet vs, const ViewParams* __restrict__ dviews, uint32_t n_views,
                                                    VisibilityOut out, SegOut seg, uint32_t fl_frame, uint32_t n_tiles, CompactFastArgs prev,
                                                    uint32_t prev_gx, uint32_t n_compact, uint32_t n_fill, ClusterFillJob fill,
                                                    ClusterWalkJob walk, const uint8_t* __restrict__ changed, SphereArgs sa,
                                                    typename WalkPlanesArg<WALK>::type wp) {
    frame_sph_workgroup<PARTIAL, INLINE_VIEWS, WALK, false>(c, vs, dviews, n_views, out, seg, fl_frame, n_tiles, prev, prev_gx, n_compact, n_fill, fill, walk, changed, sa,
                                                            (uint32_t)offsetof(FrameSphKernargs, wp));
}
// ... with 2 .. MULTI_MAX_VIEWS camera views: intersects_obb over the (row, view) pairs (MULTI above)
template <bool PARTIAL>
__global__ void __launch_bounds__(256, 8) k_frame_sph_pairs(Columns c, ViewSet vs, const ViewParams* __restrict__ dviews, uint32_t n_views,
                                                             VisibilityOut out, SegOut seg, uint32_t fl_frame, uint32_t n_tiles, CompactFastArgs prev,
                                                             uint32_t prev_gx, uint32_t n_compact, uint32_t n_fill, ClusterFillJob fill,
                                                             ClusterWalkJob walk, const uint8_t* __restrict__ changed, SphereArgs sa) {
    frame_sph_workgroup<PARTIAL, true, 0, true>(c, vs, dviews, n_views, out, seg, fl_frame, n_tiles, prev, prev_gx, n_compact, n_fill, fill, walk, changed, sa);
}

// ---------------------------------------------------------------------------------------------
// The cull-only frame of a STATIC scene over the cell order (kernels_cells.hip holds the build and the rationale).
// A wave owns 64 slots of the order.  (1) Its bounding sphere against each view's five planes: a view is dropped for the whole wave
// when the sphere lies behind one of them by more than a margin that bounds every rounding error of the per-row test -- so the rows'
// own tests (Frustum::intersects_sphere, visibility/mod.rs:829-832) would all say "outside" -- and only if every slot is subject to
// that test (CELLS_REJECTABLE) and the view culls at all.  The wave test never ACCEPTS anything.  (2) A wave with no view left and
// no ViewVisibility byte set has nothing to read, nothing to write: it ends after 36 bytes.  (3) Everybody else runs k_frame_sph's
// rule per slot -- sphere test on the slot-ordered spheres, then intersects_obb (:833-836) for the survivors on the slot-ordered
// GlobalTransforms, both fully coalesced because a wave's slots are neighbours in space AND in memory -- for the views that are left.
// Results go out BY ROW: mask bits and wave counts with atomics into memory the launch before zeroed (as the fused hierarchy frame
// does, kernels_tree.hip), the ViewVisibility byte to the column and to its slot-ordered mirror.
// Same bits as k_frame / k_frame_sph by construction; tests/test_gpu_cells.py and the differential sequences compare them.
// ---------------------------------------------------------------------------------------------
struct CellsFrameArgs {
    CellsOrder o;
    CellsZero z;
    uint4* work;             // [n_waves] (cell, views left, summary bits | flags, RenderLayers) of the cells k_cells_test leaves for k_frame_cells
    uint32_t* work_n;        // how many (this frame's counter)
    uint32_t* work_n_next;   // the next frame's counter: zeroed by k_frame_cells
    uint32_t fresh;          // the masks start from zero: no slot has contributed to them yet (pass_s is not read)
};
// margin of the wave test: the row test evaluates dot4(plane, (c_i, 1)) + r_i in f32, at most a few ulp (2^-24 relative) of the
// largest term away from the real value; the bounding sphere bounds the real value from above.  1e-5 of the terms' magnitudes is
// some twenty times that.
__device__ __forceinline__ bool cells_sphere_behind_plane(const float* pl, float cx, float cy, float cz, float R) {
    const float tx = pl[0] * cx, ty = pl[1] * cy, tz = pl[2] * cz;
    const float s = tx + ty + tz + pl[3] + R;
    const float margin = 1e-5f * (fabsf(tx) + fabsf(ty) + fabsf(tz) + fabsf(pl[3]) + R) + 1e-30f;
    return s < -margin;  // (NaN anywhere: false -- not rejected)
}
// One cell (64 slots of the order) with the views `view_in` leaves it: k_frame_sph's rule per slot, results by row.  Every lane of the
// wave calls it with the same w / view_in / st / sb.
template <bool INLINE_VIEWS>
__device__ __forceinline__ void cells_process(const Columns& c, const ViewSet& vs, const ViewParams* __restrict__ dviews, uint32_t n_views,
                                              const VisibilityOut& out, const CellsFrameArgs& a, uint32_t w, uint32_t view_in, uint32_t st, uint4 sb,
                                              float4* lds_wave, uint32_t lane) {
    // ---- (3) per slot ----
    const uint32_t slot = w * 64u + lane;
    const uint32_t row = a.o.perm[slot];
    const bool live = row != 0xFFFFFFFFu;
    const uint32_t rrow = live ? row : 0u;
    const uint32_t vv0 = a.o.vv_s[slot];
    uint32_t pass = 0u;
    uint32_t fl = sb.x & 0xFFu, emask = sb.y, emask_hi = 0u;
    if (!(sb.x & CELLS_UNIFORM_FLAGS)) {  // (wave-uniform)
        fl = c.flags[rrow];
        emask = c.layer_mask[rrow];
        if (c.layer_mask_hi) emask_hi = c.layer_mask_hi[rrow];
    }
    if (!live) fl = 0u;
    const bool ncc = (fl & 0x10u) != 0;
    if (view_in) {  // (wave-uniform)
        const float4 sp = a.o.sph_s[slot];
        const bool has_aabb = (fl & 0x04u) != 0;
        const bool base_ok = live && !ncc && (fl & 0x01u) != 0;                // InheritedVisibility
        const bool bounded = (fl & (0x04u | 0x08u)) != 0 && !(fl & 0x02u);     // has an Aabb or a Sphere, and no NoFrustumCulling
        const V4 c4 = V4{sp.x, sp.y, sp.z, 1.0f};
        const float sr = sp.w;
        uint32_t need = 0u;
        for (uint32_t v = 0; v < n_views; ++v) {
            if (!((view_in >> v) & 1u)) continue;  // (wave-uniform: dropped for the whole wave)
            const ViewParams& vp = INLINE_VIEWS ? vs.v[v] : dviews[v];
            bool vis = base_ok && ((vp.layer_mask & emask) | (vp.layer_mask_hi & emask_hi)) != 0;
            const bool cull = bounded && !(vp.flags & VIEW_NO_CPU_CULLING);
            const bool inside = sphere_inside_five_planes(vp.planes, c4, sr);
            vis = vis && (inside || !cull);
            if (vis) pass |= 1u << v;
            if (vis && cull && has_aabb) need |= 1u << v;
        }
        const unsigned long long need_m = __ballot(need != 0u);
        if (need_m) {
            Affine g = {};
            const float4* src = reinterpret_cast<const float4*>(a.o.g_s) + 192ull * w;  // the wave's 64 GlobalTransforms: 3 KB, contiguous
            if (__popcll(need_m) >= 16) {
#pragma unroll
                for (uint32_t k = 0; k < 3u; ++k) lds_wave[k * 64u + lane] = src[k * 64u + lane];
                MI_WAVE_LDS_SYNC();
                const float4 qa = lds_wave[lane * 3u], qb = lds_wave[lane * 3u + 1u], qc = lds_wave[lane * 3u + 2u];
                g.m.x_axis = V3{qa.x, qa.y, qa.z};
                g.m.y_axis = V3{qa.w, qb.x, qb.y};
                g.m.z_axis = V3{qb.z, qb.w, qc.x};
                g.t = V3{qc.y, qc.z, qc.w};
            } else if (need) {
                const float4 qa = src[lane * 3u], qb = src[lane * 3u + 1u], qc = src[lane * 3u + 2u];
                g.m.x_axis = V3{qa.x, qa.y, qa.z};
                g.m.y_axis = V3{qa.w, qb.x, qb.y};
                g.m.z_axis = V3{qb.z, qb.w, qc.x};
                g.t = V3{qc.y, qc.z, qc.w};
            }
            if (need) {
                V3 half;
                if (sb.x & CELLS_UNIFORM_HALF) {
                    const float4 h4 = a.o.sum_h[w];
                    half = V3{h4.x, h4.y, h4.z};
                } else {
                    half = ld3(c.aabb_half, row);
                }
                for (uint32_t v = 0; v < n_views; ++v) {
                    if (!((need >> v) & 1u)) continue;
                    const ViewParams& vp = INLINE_VIEWS ? vs.v[v] : dviews[v];
                    bool inside = true;
#pragma unroll
                    for (int i = 0; i < 5; ++i) {
                        const V4 pl = V4{vp.planes[4 * i], vp.planes[4 * i + 1], vp.planes[4 * i + 2], vp.planes[4 * i + 3]};
                        inside = inside && !(dot4(pl, c4) + aabb_relative_radius(half, xyz(pl), g.m) <= 0.0f);
                    }
                    if (!inside) pass &= ~(1u << v);
                }
            }
        }
    }
    // ---- the ViewVisibility byte (reset, set_visible, gpu-culling rows, mark_newly_hidden: view_visibility_tail with both frame
    //      flags) -- to the column and to the mirror ----
    uint32_t cur = vv0;
    bool vv_changed = false;
    if (live) {
        if (!ncc) cur = (cur & 1u) << 1;
        if (pass && !(cur & 1u)) {
            vv_changed = !(cur & 2u);
            cur |= 1u;
        }
        if (ncc) {
            const uint32_t nv = (fl & 0x01u) ? 3u : 0u;
            if (nv != cur) { cur = nv; vv_changed = true; }
        } else if ((cur & 3u) == 2u) {
            cur = 0u;
            vv_changed = true;
        }
        if (cur != vv0) {
            c.view_visibility[row] = (uint8_t)cur;
            a.o.vv_s[slot] = (uint8_t)cur;
        }
    }
    const unsigned long long nz = __ballot(live && cur != 0u);
    if (lane == 0u) {
        const uint32_t ns = nz ? 0u : 1u;
        if (ns != st) a.o.state[w] = ns;
    }
    // ---- results by row ----
    // The masks of this frame start out as the masks of the frame before (k_cells_blocks copied them over) and pass_s holds what each
    // slot contributed to them: only the bits that CHANGE are touched -- a handful of rows per frame under a camera that turns slowly,
    // none under one that stands still -- instead of one read-modify-write per visible row and view (2 M of them, 42 us, at 10 M rows x 4
    // views; agent-scope atomics run at some 50 G/s however they are spread).  A frame that cannot continue the one before (a.fresh:
    // the first over a new order, another shape) starts from zeroed masks and counts every slot's old contribution as nothing.
    // (The wave counts are taken from the finished masks by k_cells_blocks / k_cells_lists: added up here with atomics -- a byte per word, 64 words to
    // a cache line -- the rows a view sees, neighbours in row order as well in many scenes, queued up on a dozen lines: 29 of 34 us at
    // 1 M rows x 1 view.)
    const uint32_t pass_prev = a.fresh ? 0u : a.o.pass_s[slot];
    if (pass != pass_prev || a.fresh) a.o.pass_s[slot] = pass;
    if (live) {
        const uint32_t word = row >> 6;
        const unsigned long long bit = 1ull << (row & 63u);
        uint32_t delta = pass ^ pass_prev;
        while (delta) {
            const uint32_t v = (uint32_t)__ffs((int)delta) - 1u;
            delta &= delta - 1u;
            atomicXor(reinterpret_cast<unsigned long long*>(out.bitmask + (size_t)v * out.words_per_view + out.word_offset + word), bit);
        }
        if (vv_changed) atomicOr(reinterpret_cast<unsigned long long*>(c.vv_changed_bits + word), bit);
    }
}

// The launches.  A wave per cell that tests a sphere and ends is the workgroup dispatcher's rate, not work (39 000 such workgroups
// at 10 M rows), and every lane of it computes the same thing; a wave that takes many cells strided across the order keeps the
// dispatcher idle but reads from all over memory (every load a TLB miss) and works its cells off one after the other.  So two launches:
//   k_cells_test    a THREAD per cell: (1) and (2) of the description above -- 36 bytes, coalesced -- and the cells that are left
//                   go, with the views that are left for them, onto a work list (one atomic per wave of 64 cells).  It also
//                   zeroes the next frame's masks / wave counts / change words.
//   k_frame_cells   a WAVE per listed cell (grid-stride over the list, the list's length read from the device): (3).  Neighbours on
//                   the list are neighbours in the order and in memory.  The riders (the previous frame's compaction, a cluster
//                   fill) sit in front of its grid.
// The list's counter alternates between two words by frame; k_frame_cells zeroes the one the next frame's test will count in.
template <bool INLINE_VIEWS>
__global__ void __launch_bounds__(1024) k_cells_test(ViewSet vs, const ViewParams* __restrict__ dviews, uint32_t n_views, CellsFrameArgs a) {
    const uint32_t gid = blockIdx.x * 1024u + threadIdx.x, gsz = gridDim.x * 1024u;
#pragma unroll
    for (uint32_t k = 0; k < 3u; ++k)
        for (uint32_t i = gid; i < a.z.zero_words[k]; i += gsz) a.z.zero[k][i] = 0ull;
    const uint32_t cell = gid;
    const bool have = cell < a.o.n_waves;
    const uint32_t lc = have ? cell : 0u;
    const float4 sa = a.o.sum_a[lc];
    const uint4 sb = a.o.sum_b[lc];
    const uint32_t st = a.o.state[lc];
    const bool rejectable = (sb.x & CELLS_REJECTABLE) != 0u;
    uint32_t view_in = 0u;  // bit v: some slot of the cell may be visible in view v
    for (uint32_t v = 0; v < n_views; ++v) {
        const ViewParams& vp = INLINE_VIEWS ? vs.v[v] : dviews[v];
        bool outside = false;
        if (rejectable && !(vp.flags & VIEW_NO_CPU_CULLING)) {
#pragma unroll
            for (int i = 0; i < 5; ++i) outside = outside || cells_sphere_behind_plane(vp.planes + 4 * i, sa.x, sa.y, sa.z, sa.w);
        }
        if (!outside) view_in |= 1u << v;
    }
    // nothing can be visible and nothing was: every output of the cell's rows is the zero (or, in a frame that continues the one
    // before, the unchanged bit) it already is
    const bool todo = have && !(view_in == 0u && (st & 1u));
    // one atomic per workgroup on the list's counter (agent-scope atomics on one word serialise at ~25 ns each: a wave's worth of cells
    // per atomic was 12 us of a 17 us launch)
    __shared__ uint32_t s_cnt[16], s_base;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const unsigned long long m = __ballot(todo);
    if (lane == 0u) s_cnt[wv] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0u) {
        uint32_t tot = 0;
        for (uint32_t k = 0; k < 16u; ++k) {
            const uint32_t cnt = s_cnt[k];
            s_cnt[k] = tot;
            tot += cnt;
        }
        s_base = tot ? atomicAdd(a.work_n, tot) : 0u;
    }
    __syncthreads();
    if (todo) {
        const uint32_t at = s_base + s_cnt[wv] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        a.work[at] = make_uint4(cell, view_in, sb.x | ((st & 1u) << 31), sb.y);
    }
}
__device__ __forceinline__ void cells_lists_block(const CellsFinishArgs& f, uint32_t bx, uint32_t v);
template <bool INLINE_VIEWS>
__global__ void __launch_bounds__(256) k_frame_cells(Columns c, ViewSet vs, const ViewParams* __restrict__ dviews, uint32_t n_views,
                                                      VisibilityOut out, CellsFrameArgs a, uint32_t n_tiles, CompactFastArgs prev, uint32_t prev_gx,
                                                      uint32_t n_compact, uint32_t n_fill, ClusterFillJob fill, uint32_t n_lists, uint32_t lists_views,
                                                      CellsFinishArgs lists) {
    __shared__ __attribute__((aligned(16))) uint32_t lds_raw[FRAME_LDS_WORDS + 4];
    float4 (*lds_g)[192] = reinterpret_cast<float4 (*)[192]>(lds_raw);
    if (n_lists) {
        // the previous frame's VisibleEntities lists (k_cells_lists' work: MI_CULL_MORE_FRAMES deferred them) ride behind the other riders
        // at the head of the grid: they read that frame's finished masks and block prefixes, which this launch does not touch.
        // (10 M rows x 4 views, us per frame: lists as a launch of their own 71.1; riding at the head 64.2, spread between the cell
        // workgroups 68.3, at the end of the grid 71.6 -- there they only start when the cells are done.)
        const uint32_t first = n_compact + n_fill;
        if (blockIdx.x >= first && blockIdx.x < first + n_lists) {
            const uint32_t id = blockIdx.x - first, per_view = n_lists / lists_views;
            cells_lists_block(lists, id % per_view, id / per_view);
            return;
        }
    }
    {
        ClusterWalkJob no_walk{};
        if (frame_riders<0>(n_tiles, prev, prev_gx, n_compact, n_fill, fill, no_walk, vs, lds_raw)) return;
    }
    const uint32_t tile = blockIdx.x - (gridDim.x - n_tiles);
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t n_work = *a.work_n;
    if (tile == 0u && threadIdx.x == 0u) *a.work_n_next = 0u;  // (the next frame's test counts in the other word)
    for (uint32_t i = tile * 4u + wv; i < n_work; i += n_tiles * 4u) {  // (wave-uniform)
        const uint4 job = a.work[i];
        const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)job.x), view_in = (uint32_t)__builtin_amdgcn_readfirstlane((int)job.y);
        uint4 sb;
        sb.x = (uint32_t)__builtin_amdgcn_readfirstlane((int)job.z) & 0x7FFFFFFFu;
        sb.y = (uint32_t)__builtin_amdgcn_readfirstlane((int)job.w);
        sb.z = sb.w = 0u;
        const uint32_t st = (uint32_t)__builtin_amdgcn_readfirstlane((int)job.z) >> 31;
        cells_process<INLINE_VIEWS>(c, vs, dviews, n_views, out, a, w, view_in, st, sb, lds_g[wv], lane);
        MI_WAVE_LDS_SYNC();  // (the next cell's transpose reuses the wave's buffer)
    }
}

// ---------------------------------------------------------------------------------------------
// Level 0 of propagation: flat rows (sync_simple_transforms) and tree roots
// (propagate_parent_transforms, systems.rs:522-530).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_level0_propagate(Columns c, uint32_t n_level0,
                                                           const uint8_t* __restrict__ node_flags,
                                                           const uint8_t* __restrict__ changed,
                                                           const uint8_t* __restrict__ tree_bytes, bool all_dirty,
                                                           bool static_opt) {
    __shared__ float4 lds_g[4][192];
    const uint32_t row = blockIdx.x * 256u + threadIdx.x;
    const bool live = row < n_level0;
    bool write = false;
    Affine g = {};
    if (live) {
        const bool has_children = node_flags ? (node_flags[row] & 1u) != 0 : false;
        if (has_children) {
            // roots: skipped only when the static optimisation is on and the tree is clean
            const bool tree_changed = all_dirty || !tree_bytes || tree_bytes[row] != 0;
            write = !static_opt || tree_changed;
        } else {
            // flat rows: Changed<Transform> || Added<GlobalTransform> (systems.rs:45-50)
            write = all_dirty || !changed || row_changed(changed[row], c.changed_gen);
        }
        if (write) {
            const V3 t = ld3(c.translation, row);
            const V4 q = ld4(c.rotation, row);
            const V3 s = ld3(c.scale, row);
            g = affine_from_srt(s, q, t);
        }
    }
    const unsigned long long w = __ballot(write);
    const unsigned long long lv = __ballot(live);
    // a wave that rewrites every one of its rows (the dirty frame) stores three contiguous 1 KB wave rows through the LDS
    // transpose; otherwise only the lanes that write store their own 48 bytes
    if (w == lv && lv) store_affine_coalesced(lds_g[threadIdx.x >> 6], c.global, row & ~63u, n_level0, threadIdx.x & 63u, g);
    else if (write) st_affine(c.global, row, g);
    if ((threadIdx.x & 63u) == 0 && lv) c.g_changed_bits[row >> 6] = w;
}

// GlobalTransforms ahead of the frame (context.cpp, mi_commit_upload_window): rows lo .. hi-1 of a flat table, From(Transform) as
// sync_simple_transforms writes it (systems.rs:45-50) -- the same affine_from_srt the frame kernels call -- into a buffer of the
// library's own (not the GlobalTransform column: that one is the frame's to write), from where the copy engine takes them to the host
// while later pieces of the upload are still arriving.
__global__ void __launch_bounds__(256) k_globals_ahead(const float* __restrict__ t, const float* __restrict__ r, const float* __restrict__ s,
                                                        uint32_t lo, uint32_t hi, float* __restrict__ out) {
    __shared__ float4 lds_g[4][192];
    const uint32_t row = (lo & ~63u) + blockIdx.x * 256u + threadIdx.x;
    const bool live = row >= lo && row < hi;
    Affine g = {};
    if (live) g = affine_from_srt(ld3(s, row), ld4(r, row), ld3(t, row));
#ifdef MI_EXP_MUTANT  // (a deliberately wrong build: tests/test_gpu_chunked_frames.py must notice, tools/gpu_mutant.sh)
    if ((row & 0xFFFu) == 0x321u) g.t.x += 1.0f;
#endif
    const uint32_t wave_row0 = row & ~63u;
    // (wave-uniform) a wave wholly inside the piece stores its three contiguous 1 KB rows through the LDS transpose
    if (wave_row0 >= lo && wave_row0 + 64u <= hi) store_affine_coalesced(lds_g[threadIdx.x >> 6], out, wave_row0, hi, threadIdx.x & 63u, g);
    else if (live) st_affine(out, row, g);
}
hipError_t launch_globals_ahead(const float* t, const float* r, const float* s, uint32_t lo, uint32_t hi, float* out, hipStream_t stream) {
    if (hi <= lo) return hipSuccess;
    const uint32_t first = lo & ~63u;
    hipLaunchKernelGGL(k_globals_ahead, dim3((hi - first + 255u) / 256u), dim3(256), 0, stream, t, r, s, lo, hi, out);
    return hipGetLastError();
}

// reset_view_visibility: bits = (bits & 1) << 1 for rows without NoCpuCulling; clears the change mask.
// RowSummary (kernels.h) of the waves first_wave .. first_wave + n_waves - 1, from the columns: a wave per 64 rows compares every
// live row's bits with its first row's.  parts: ROWSUM_PART_AABB rewrites words 0-5 and the Aabb bit, ROWSUM_PART_FLAGS words 6-7's
// flags / layers and their bit (the other part of word 7 is kept).
__global__ void __launch_bounds__(256) k_row_summary(Columns c, uint32_t first_wave, uint32_t n_waves, uint32_t parts, uint32_t* __restrict__ summary) {
    const uint32_t w = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (w >= n_waves) return;
    const uint32_t wave = first_wave + w, row = wave * 64u + lane;
    const bool live = row < c.n;
    if (wave * 64u >= c.n) return;
    uint32_t* o = summary + (size_t)wave * ROWSUM_WORDS;
    uint32_t bits = o[7];
    const uint32_t rrow = live ? row : wave * 64u;  // (dead lanes repeat the first row: they agree with it)
    if (parts & ROWSUM_PART_AABB) {
        const F3 ce = reinterpret_cast<const F3*>(c.aabb_center)[rrow], he = reinterpret_cast<const F3*>(c.aabb_half)[rrow];
        const uint32_t v[6] = {__float_as_uint(ce.x), __float_as_uint(ce.y), __float_as_uint(ce.z), __float_as_uint(he.x), __float_as_uint(he.y), __float_as_uint(he.z)};
        bool same = true;
#pragma unroll
        for (int k = 0; k < 6; ++k) same = same && v[k] == (uint32_t)__builtin_amdgcn_readfirstlane((int)v[k]);
        const bool uniform = __ballot(!same) == 0ull;
        bits = (bits & ~ROWSUM_UNIFORM_AABB) | (uniform ? ROWSUM_UNIFORM_AABB : 0u);
        if (lane == 0u) {  // (lane 0 is the wave's first row, live: wave * 64 < n)
#pragma unroll
            for (int k = 0; k < 6; ++k) o[k] = v[k];
        }
    }
    if (parts & ROWSUM_PART_FLAGS) {
        const uint32_t fl = c.flags[rrow], lm = c.layer_mask[rrow];
        const uint32_t hi = c.layer_mask_hi ? c.layer_mask_hi[rrow] : 0u;  // (the summary holds one word of layers: rows above 31 are not summarised)
        const uint32_t fl0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)fl), lm0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)lm);
        const bool uniform = __ballot(fl != fl0 || lm != lm0 || hi != 0u) == 0ull;
        bits = (bits & ~(ROWSUM_UNIFORM_FLAGS | 0xFFu)) | (uniform ? ROWSUM_UNIFORM_FLAGS : 0u) | (fl0 & 0xFFu);
        if (lane == 0u) o[6] = lm0;
    }
    if (lane == 0u) o[7] = bits & (ROWSUM_UNIFORM_AABB | ROWSUM_UNIFORM_FLAGS | 0xFFu);
}

__global__ void __launch_bounds__(256) k_vis_begin(Columns c) {
    const uint32_t row = blockIdx.x * 256u + threadIdx.x;
    const bool live = row < c.n;
    if (live) {
        const uint32_t fl = c.flags[row];
        if (!(fl & 0x10u)) c.view_visibility[row] = (uint8_t)((c.view_visibility[row] & 1u) << 1);
    }
    if ((threadIdx.x & 63u) == 0 && live) c.vv_changed_bits[row >> 6] = 0ull;
}

// check_visibility_gpu_culling for NoCpuCulling rows, then mark_newly_hidden_entities_invisible.
__global__ void __launch_bounds__(256) k_vis_end(Columns c) {
    const uint32_t row = blockIdx.x * 256u + threadIdx.x;
    const bool live = row < c.n;
    bool changed = false;
    if (live) {
        const uint32_t fl = c.flags[row];
        const uint32_t cur = c.view_visibility[row];
        if (fl & 0x10u) {
            const uint32_t nv = (fl & 0x01u) ? 3u : 0u;  // ViewVisibility::VISIBLE / HIDDEN, set_if_neq
            if (nv != cur) { c.view_visibility[row] = (uint8_t)nv; changed = true; }
        } else if ((cur & 3u) == 2u) {  // was_visible_now_hidden, mod.rs:264-267
            c.view_visibility[row] = 0;
            changed = true;
        }
    }
    const unsigned long long chg = __ballot(changed);
    if ((threadIdx.x & 63u) == 0 && chg)
        atomicOr(reinterpret_cast<unsigned long long*>(&c.vv_changed_bits[row >> 6]), chg);
}

static inline uint32_t blocks_for(uint32_t n) { return (n + 255u) / 256u; }

// Small dirty-row uploads: the three Transform columns of a few rows are staged back to back in pinned host
// memory and scattered by ONE kernel that reads the staging buffer over PCIe, instead of three DMA copies
// (each a separate ~4 us engine submission).  src = translation[3n] | rotation[4n] | scale[3n].
__global__ void __launch_bounds__(256) k_upload_trs(const float* __restrict__ src, float* t, float* r, float* s,
                                                     uint32_t first_row, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < 3u * n) {
        t[3ull * first_row + i] = src[i];
        s[3ull * first_row + i] = src[7ull * n + i];
    }
    if (i < 4u * n) r[4ull * first_row + i] = src[3ull * n + i];
}
// Sparse dirty-row upload (the rows a Changed<Transform> query yields): rows[n], t[3n], r[4n], s[3n] in pinned host memory;
// a thread per row scatters its Transform and raises the row's changed byte.
// With a hierarchy under the static-scene rule the kernel is also that frame's mark_dirty_trees for the rows it uploads (mark_*
// != nullptr): every uploaded row climbs to its root setting TransformTreeChanged, exactly like k_mark_dirty (kernels_tree.hip), and
// the launch zeroes the other half of the double-buffered marks -- the mark launch of its own (>= 4.3 us) drops out of a frame whose
// changes all arrive this way.
// WIDE (the three source arrays are 16-byte aligned: upload windows are): a wave fetches its 64 entries' translations and scales as
// 48 float4 each and the rotations as one float4 per lane, and redistributes through LDS -- three wide loads per wave and every line
// of the pinned memory crossing PCIe once, instead of ten dword loads per lane at 12- and 16-byte strides.
template <bool WIDE>
__global__ void __launch_bounds__(256) k_upload_trs_indexed(const uint32_t* __restrict__ rows, const float* __restrict__ ft,
                                                             const float* __restrict__ fr, const float* __restrict__ fs, uint32_t n, float* t,
                                                             float* r, float* s, uint8_t* changed, uint32_t changed_gen,
                                                             const uint32_t* __restrict__ parent_idx, uint8_t* mark_bytes,
                                                             uint32_t* __restrict__ clear_words, uint32_t n_clear_words,
                                                             const uint32_t* __restrict__ anc, float* __restrict__ g_ahead, uint32_t g_reversed) {
    // g_ahead (pinned host memory, or nullptr): entry i's GlobalTransform as the changed-rows frame of a flat table will write it --
    // From(Transform), sync_simple_transforms (systems.rs:45-50), the frame kernels' affine_from_srt -- written back over PCIe by
    // the launch that reads the Transforms over it (the link is full duplex), three contiguous 1 KB rows per wave; in upload order,
    // or (g_reversed: the window's rows descend -- a query in spawn order over rows numbered by Entity key, whose index is stored
    // inverted) back to front, so that they stand in ascending row order either way
    __shared__ float4 lds_g[4][192];
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    if (clear_words)
        for (uint32_t w = gid; w < n_clear_words; w += gridDim.x * 256u) clear_words[w] = 0u;
    // ft / fr / fs == nullptr: the window does not carry that component (MI_UPLOAD_TRANSLATION / _ROTATION / _SCALE): its column keeps
    // what it holds, and a GlobalTransform written ahead takes the resident value
    for (uint32_t i0 = gid & ~63u; i0 < n; i0 += gridDim.x * 256u) {  // (wave-uniform trip count: the transpose below is a wave's)
        const uint32_t i = i0 + (threadIdx.x & 63u);
        const bool live = i < n;
        V3 tt = {}, ss = {1.0f, 1.0f, 1.0f};
        V4 qq = {0.0f, 0.0f, 0.0f, 1.0f};
        if constexpr (WIDE) {
            const uint32_t lane = threadIdx.x & 63u;
            float4* lds_wave = lds_g[threadIdx.x >> 6];
            const uint64_t f4 = 48ull * (i0 >> 6);  // (i0 is a multiple of 64: float 3 * i0 is float4 48 * (i0 / 64))
            if (lane < 48u && 4ull * (f4 + lane) < 3ull * n) {  // (a float4 that starts inside the array; its tail lies in the window's padding)
                if (ft) lds_wave[lane] = reinterpret_cast<const float4*>(ft)[f4 + lane];
                if (fs) lds_wave[48u + lane] = reinterpret_cast<const float4*>(fs)[f4 + lane];
            }
            if (live && fr) {
                const float4 q4 = reinterpret_cast<const float4*>(fr)[i];
                qq = V4{q4.x, q4.y, q4.z, q4.w};
            }
            MI_WAVE_LDS_SYNC();
            if (live) {
                const float* lf = reinterpret_cast<const float*>(lds_wave);
                if (ft) tt = V3{lf[3u * lane], lf[3u * lane + 1u], lf[3u * lane + 2u]};
                if (fs) ss = V3{lf[192u + 3u * lane], lf[192u + 3u * lane + 1u], lf[192u + 3u * lane + 2u]};
            }
            MI_WAVE_LDS_SYNC();  // (the transpose of the GlobalTransforms below reuses the buffer)
        }
        if (live) {
            const uint32_t row = rows[i];
            if constexpr (!WIDE) {
                if (ft) tt = V3{ft[3ull * i], ft[3ull * i + 1u], ft[3ull * i + 2u]};  // (dword loads: a caller's arrays promise no alignment)
                if (fs) ss = V3{fs[3ull * i], fs[3ull * i + 1u], fs[3ull * i + 2u]};
                if (fr) qq = V4{fr[4ull * i], fr[4ull * i + 1u], fr[4ull * i + 2u], fr[4ull * i + 3u]};
            }
            if (ft) t[3ull * row] = tt.x, t[3ull * row + 1u] = tt.y, t[3ull * row + 2u] = tt.z;
            else if (g_ahead) tt = V3{t[3ull * row], t[3ull * row + 1u], t[3ull * row + 2u]};
            if (fs) s[3ull * row] = ss.x, s[3ull * row + 1u] = ss.y, s[3ull * row + 2u] = ss.z;
            else if (g_ahead) ss = V3{s[3ull * row], s[3ull * row + 1u], s[3ull * row + 2u]};
            if (fr) r[4ull * row] = qq.x, r[4ull * row + 1u] = qq.y, r[4ull * row + 2u] = qq.z, r[4ull * row + 3u] = qq.w;
            else if (g_ahead) qq = V4{r[4ull * row], r[4ull * row + 1u], r[4ull * row + 2u], r[4ull * row + 3u]};
            changed[row] = (uint8_t)changed_gen;  // a stamp, not a flag: see row_changed() in kernels.h
            if (mark_bytes) mark_row_and_ancestors(row, parent_idx, mark_bytes, anc, 0xFFFFu);  // (a hierarchy is at most 65 535 levels deep here)
        }
        if (g_ahead) {
            Affine a = affine_from_srt(ss, qq, tt);
#ifdef MI_EXP_MUTANT
            if ((i & 0xFFFu) == 0x321u) a.t.x += 1.0f;
#endif
            const uint32_t lane = threadIdx.x & 63u;
            if (!g_reversed) store_affine_coalesced(lds_g[threadIdx.x >> 6], g_ahead, i0, n, lane, a);
            else {
                // entry i goes to slot n-1-i: the wave's 64 entries fill slots base .. base+63, base = n-64-i0 (below 0 in the wave
                // that holds the last entries: those slots do not exist), lane l's entry in slot base + 63 - l
                float4* lds_wave = lds_g[threadIdx.x >> 6];
                const uint32_t sl = 63u - lane;
                lds_wave[sl * 3u + 0u] = make_float4(a.m.x_axis.x, a.m.x_axis.y, a.m.x_axis.z, a.m.y_axis.x);
                lds_wave[sl * 3u + 1u] = make_float4(a.m.y_axis.y, a.m.y_axis.z, a.m.z_axis.x, a.m.z_axis.y);
                lds_wave[sl * 3u + 2u] = make_float4(a.m.z_axis.z, a.t.x, a.t.y, a.t.z);
                MI_WAVE_LDS_SYNC();
                const long long first = 3ll * ((long long)n - 64ll - (long long)i0);  // float4 index of slot `base`
#pragma unroll
                for (uint32_t k = 0; k < 3u; ++k) {
                    const long long at = first + (long long)(k * 64u + lane);
                    if (at >= 0) reinterpret_cast<float4*>(g_ahead)[at] = lds_wave[k * 64u + lane];
                }
                MI_WAVE_LDS_SYNC();  // (the next trip of this wave writes the same buffer)
            }
        }
    }
}
hipError_t launch_upload_trs_indexed(const uint32_t* rows, const float* t_src, const float* r_src, const float* s_src, uint32_t n, float* t,
                                     float* r, float* s, uint8_t* changed, uint32_t changed_gen, hipStream_t stream, const uint32_t* parent_idx,
                                     uint8_t* mark_bytes, uint32_t* clear_words, uint32_t n_clear_words, const uint32_t* anc, float* g_ahead,
                                     bool g_reversed) {
    if (n == 0) return hipSuccess;
    // the sources are pinned host memory read over PCIe: enough lanes to keep the link busy, not one workgroup per 256 rows of a
    // million-row upload; and enough workgroups for the words to clear
    uint32_t blocks = blocks_for(n) < 2048u ? blocks_for(n) : 2048u;
    if (clear_words) {
        const uint32_t cb = (n_clear_words + 1023u) / 1024u < 1024u ? (n_clear_words + 1023u) / 1024u : 1024u;
        blocks = blocks > cb ? blocks : cb;
    }
    const bool wide = (((uintptr_t)t_src | (uintptr_t)r_src | (uintptr_t)s_src) & 15u) == 0;  // (a missing component: nullptr, aligned)
    if (wide)
        MI_LAUNCH(k_upload_trs_indexed<true>, dim3(blocks), dim3(256), 0, stream, rows, t_src, r_src, s_src, n, t, r, s, changed, changed_gen, parent_idx,
                  mark_bytes, clear_words, n_clear_words, mark_bytes ? anc : nullptr, g_ahead, g_reversed ? 1u : 0u);
    else
        MI_LAUNCH(k_upload_trs_indexed<false>, dim3(blocks), dim3(256), 0, stream, rows, t_src, r_src, s_src, n, t, r, s, changed, changed_gen, parent_idx,
                  mark_bytes, clear_words, n_clear_words, mark_bytes ? anc : nullptr, g_ahead, g_reversed ? 1u : 0u);
    return hipGetLastError();
}

// Sparse read-back helpers: u8 popcount of every 64-bit mask word (the wave counts k_compact_fast consumes), and
// a gather of the listed rows' GlobalTransforms into a dense array.
__global__ void __launch_bounds__(256) k_popcount_words(const uint64_t* __restrict__ bits, uint32_t n_words, uint32_t n_rows,
                                                         uint8_t* cnt) {
    const uint32_t w = blockIdx.x * 256u + threadIdx.x;
    if (w >= n_words) return;
    uint64_t m = bits[w];
    const uint32_t tail = n_rows - w * 64u;
    if (tail < 64u) m &= (1ull << tail) - 1ull;
    cnt[w] = (uint8_t)__popcll(m);
}
__global__ void __launch_bounds__(256) k_gather_global(const uint32_t* __restrict__ rows, const uint32_t* __restrict__ total,
                                                        uint32_t capacity, const float* __restrict__ g, float* out) {
    const uint32_t m = *total < capacity ? *total : capacity;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < m * 3u; i += gridDim.x * 256u) {
        const uint32_t row = rows[i / 3u];
        reinterpret_cast<float4*>(out)[i] = reinterpret_cast<const float4*>(g)[3ull * row + (i % 3u)];
    }
}
// Mesh-instance wire format for the listed rows: MeshInputUniform::world_from_local (the affine transposed to three
// Vec4 rows, bevy_math/src/affine3.rs:27-34) and MeshCullingData (center / half extents as Vec4, infinite half
// extents without an Aabb, bevy_pbr/src/render/mesh.rs:1646-1657).
__global__ void __launch_bounds__(256) k_gather_mesh_inputs(const uint32_t* __restrict__ rows, const uint32_t* __restrict__ total,
                                                             uint32_t capacity, Columns c, float4* out_wfl, float4* out_cull) {
    const uint32_t m = *total < capacity ? *total : capacity;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < m; i += gridDim.x * 256u) {
        const uint32_t row = rows[i];
        const Affine g = ld_affine(c.global, row);
        out_wfl[3u * i] = make_float4(g.m.x_axis.x, g.m.y_axis.x, g.m.z_axis.x, g.t.x);
        out_wfl[3u * i + 1u] = make_float4(g.m.x_axis.y, g.m.y_axis.y, g.m.z_axis.y, g.t.y);
        out_wfl[3u * i + 2u] = make_float4(g.m.x_axis.z, g.m.y_axis.z, g.m.z_axis.z, g.t.z);
        if (c.flags[row] & 0x04u) {
            const V3 ce = ld3(c.aabb_center, row), he = ld3(c.aabb_half, row);
            out_cull[2u * i] = make_float4(ce.x, ce.y, ce.z, 0.0f);
            out_cull[2u * i + 1u] = make_float4(he.x, he.y, he.z, 0.0f);
        } else {
            out_cull[2u * i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            out_cull[2u * i + 1u] = make_float4(INFINITY, INFINITY, INFINITY, 0.0f);
        }
    }
}
hipError_t launch_gather_mesh_inputs(const uint32_t* rows, const uint32_t* total, uint32_t capacity, const Columns& c, float* out_wfl,
                                     float* out_cull, hipStream_t stream) {
    if (capacity == 0) return hipSuccess;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(2048, ((uint64_t)capacity + 255) / 256);
    MI_LAUNCH(k_gather_mesh_inputs, dim3(blocks), dim3(256), 0, stream, rows, total, capacity, c, reinterpret_cast<float4*>(out_wfl),
              reinterpret_cast<float4*>(out_cull));
    return hipGetLastError();
}

hipError_t launch_popcount_words(const uint64_t* bits, uint32_t n_rows, uint8_t* cnt, hipStream_t stream) {
    const uint32_t n_words = (n_rows + 63u) / 64u;
    if (n_words == 0) return hipSuccess;
    MI_LAUNCH(k_popcount_words, dim3(blocks_for(n_words)), dim3(256), 0, stream, bits, n_words, n_rows, cnt);
    return hipGetLastError();
}
hipError_t launch_gather_global(const uint32_t* rows, const uint32_t* total, uint32_t capacity, const float* g, float* out,
                                hipStream_t stream) {
    if (capacity == 0) return hipSuccess;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(2048, ((uint64_t)capacity * 3 + 255) / 256);
    MI_LAUNCH(k_gather_global, dim3(blocks), dim3(256), 0, stream, rows, total, capacity, g, out);
    return hipGetLastError();
}

// Frame results -> pinned window (kernels.h PackResultsJob).  Every workgroup derives the same section offsets from the counts;
// the stores go straight over PCIe, lane-contiguous.
__global__ void __launch_bounds__(256) k_pack_results(PackResultsJob j) {
    const uint32_t changed = j.changed_total ? *j.changed_total : 0u;
    const uint64_t cl_total = j.cluster_total ? *j.cluster_total : 0ull;
    const bool big = changed > j.big_rows;  // the copy engine's job
    const bool s_rows = j.changed_total && j.want_changed_rows && changed <= j.changed_capacity && !big;
    const bool s_g = j.changed_total && j.g && changed <= j.changed_capacity && !big;
    const bool s_cl = j.cluster_total != nullptr;
    const bool s_idx = s_cl && j.cluster_indices && cl_total <= j.cluster_capacity;
    const bool overflow = s_cl && cl_total > j.cluster_indices_alloc;
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x, stride = gridDim.x * 256u;
    uint64_t off = 0;
    const uint64_t o_rows = off;  off += s_rows ? pack_align((uint64_t)changed * 4u) : 0u;
    const uint64_t o_g = off;     off += s_g ? pack_align((uint64_t)changed * 48u) : 0u;
    // the lists' sections: every workgroup walks the same few counts (L2 hits) to place them
    const uint64_t o_lists = off;
    for (uint32_t l = 0; l < j.n_lists; ++l) {
        const uint32_t cnt = j.list_total[l] ? *j.list_total[l] : 0u;
        if (cnt <= j.list_capacity[l]) off += pack_align((uint64_t)cnt * 4u);
    }
    const uint64_t o_off = off;   off += s_cl ? pack_align(((uint64_t)j.n_clusters + 1u) * 4u) : 0u;
    const uint64_t o_cnt = off;   off += s_cl ? pack_align((uint64_t)j.n_clusters * 24u) : 0u;
    const uint64_t o_idx = off;   off += s_idx ? pack_align(cl_total * 4u) : 0u;
    const bool fits = off <= j.payload_bytes && !overflow;
    if (tid == 0) {
        j.header[0] = changed;
        j.header[2] = (uint32_t)cl_total;
        j.header[3] = (uint32_t)(cl_total >> 32);
        j.header[4] = s_cl ? __float_as_uint(*j.farthest_z) : 0u;
        j.header[5] = fits ? 1u : 0u;
        j.header[6] = big && j.changed_total ? 1u : 0u;
    }
    if (tid < j.n_lists) j.header[8u + tid] = j.list_total[tid] ? *j.list_total[tid] : 0u;
    if (!fits) return;
    if (s_rows) {
        uint32_t* d = reinterpret_cast<uint32_t*>(j.payload + o_rows);
        for (uint32_t i = tid; i < changed; i += stride) d[i] = j.changed_rows[i];
    }
    if (s_g) {
        float4* d = reinterpret_cast<float4*>(j.payload + o_g);
        for (uint32_t i = tid; i < changed * 3u; i += stride)
            d[i] = reinterpret_cast<const float4*>(j.g)[3ull * j.changed_rows[i / 3u] + (i % 3u)];
    }
    {
        uint64_t o = o_lists;
        for (uint32_t l = 0; l < j.n_lists; ++l) {
            const uint32_t cnt = j.list_total[l] ? *j.list_total[l] : 0u;
            if (cnt > j.list_capacity[l]) continue;
            uint32_t* d = reinterpret_cast<uint32_t*>(j.payload + o);
            const uint32_t* src = j.list_rows[l] + (j.list_base[l] ? *j.list_base[l] : 0ull);
            for (uint32_t i = tid; i < cnt; i += stride) d[i] = src[i];
            o += pack_align((uint64_t)cnt * 4u);
        }
    }
    if (s_cl) {
        uint32_t* d0 = reinterpret_cast<uint32_t*>(j.payload + o_off);
        for (uint32_t i = tid; i <= j.n_clusters; i += stride) d0[i] = j.cluster_offsets[i];
        uint32_t* d1 = reinterpret_cast<uint32_t*>(j.payload + o_cnt);
        for (uint32_t i = tid; i < j.n_clusters * 6u; i += stride) d1[i] = j.cluster_counts[i];
    }
    if (s_idx) {
        uint32_t* d = reinterpret_cast<uint32_t*>(j.payload + o_idx);
        for (uint64_t i = tid; i < cl_total; i += stride) d[i] = j.cluster_indices[i];
    }
}
hipError_t launch_pack_results(const PackResultsJob& job, hipStream_t stream) {
    // enough lanes to keep the PCIe writes flowing; the window is a few MB at most
    const uint64_t units = job.payload_bytes / 4u;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(256, std::max<uint64_t>(1, (units + 1023u) / 1024u));
    MI_LAUNCH(k_pack_results, dim3(blocks), dim3(256), 0, stream, job);
    return hipGetLastError();
}

#ifdef MI_EXP_TIMELINE
extern "C" int mi_exp_timeline(unsigned long long* out /* 3 * 16384 */, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(mi_timeline), sizeof(unsigned long long) * 3 * TIMELINE_WGS) != hipSuccess) return 1;
    if (reset) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(mi_timeline)) != hipSuccess || hipMemset(p, 0, sizeof(unsigned long long) * 3 * TIMELINE_WGS) != hipSuccess) return 1;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(mi_walk_marks)) != hipSuccess || hipMemset(p, 0, sizeof(unsigned long long) * 16 * 4096) != hipSuccess) return 1;
    }
    return 0;
}
extern "C" int mi_exp_walk_marks(unsigned long long* out /* 16 * 4096 */) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(mi_walk_marks), sizeof(unsigned long long) * 16 * 4096) != hipSuccess;
}
#endif

hipError_t launch_row_summary(const Columns& c, uint32_t first_wave, uint32_t n_waves, uint32_t parts, uint32_t* summary, hipStream_t stream) {
    if (n_waves == 0) return hipSuccess;
    MI_LAUNCH(k_row_summary, dim3((n_waves + 3u) / 4u), dim3(256), 0, stream, c, first_wave, n_waves, parts, summary);
    return hipGetLastError();
}

hipError_t launch_upload_trs(const float* pinned_src, float* t, float* r, float* s, uint32_t first_row, uint32_t n,
                             hipStream_t stream) {
    if (n == 0) return hipSuccess;
    MI_LAUNCH(k_upload_trs, dim3(blocks_for(4u * n)), dim3(256), 0, stream, pinned_src, t, r, s, first_row, n);
    return hipGetLastError();
}

// prev != nullptr: the previous frame's deferred compaction rides in extra workgroups of this launch; fill != nullptr: so does the
// fill of the previous frame's light-cluster assignment
// The riding walk's plane table as a kernel argument: from the host copy the context hands the launch (h); the job's view then carries
// no table of its own.  Returns false (the staged copy stays) when there is none or it does not fit.
// (nulling view.x_planes makes the walkers read the argument segment: only the PLANES_IN_LDS instantiations do that -- the riders of
// the frame kernels are cluster_walk_tail<true, ...> -- and a table of <= WALK_PLANES_MAX floats always fits their arena.)
static bool take_walk_planes(ClusterWalkJob* wj, WalkPlanes* wp, const WalkPlanesHost& h) {
#ifdef MI_EXP_STAGED_PLANES  // (A/B build: the walkers read the table from the pinned staging arena, as until round 5)
    return false;
#endif
    if (!h.f || h.n == 0 || h.n > WALK_PLANES_MAX) return false;
    for (uint32_t i = 0; i < h.n; ++i) wp->f[i] = h.f[i];
    wj->view.x_planes = wj->view.y_planes = wj->view.z_planes = nullptr;
    return true;
}
int g_multi_view_mode = 0;  // MI_MULTI_VIEW (read by mi_ctx_create): 0 = as described at k_frame's MULTI, 1 = never
template <int PROP>
static hipError_t launch_frame(const Columns& c, const ViewSet* views_inline, const ViewParams* d_views, uint32_t n_views,
                               const VisibilityOut& out, const SegOut& seg, uint32_t flags, const CompactFastArgs* prev,
                               const ClusterFillJob* fill, const ClusterWalkJob* walk, hipStream_t stream, const uint8_t* changed = nullptr,
                               const WalkPlanesHost& walk_planes = WalkPlanesHost{nullptr, 0}) {
    if (c.n == 0) return hipSuccess;
    if (!c.row_summary) return hipErrorInvalidValue;  // (the kernel loads it unconditionally: row_summary_ensure comes first)
    const uint32_t n_tiles = blocks_for(c.n);
    CompactFastArgs pa{};
    uint32_t prev_gx = 1, prev_blocks = 0;
    if (prev && prev->n && prev->n_segments) {
        pa = *prev;
        prev_gx = compact_fast_gx(pa.n, false);
        prev_blocks = prev_gx * pa.n_segments;
    }
    ClusterFillJob fj{};
    uint32_t fill_blocks = 0;
    if (fill) {
        fj = *fill;
        fill_blocks = CLUSTER_FILL_RIDE_BLOCKS;
    }
    ClusterWalkJob wj{};
    uint32_t walk_blocks = 0;
    bool with_walk = false;
    if (walk && walk->n_blocks && n_views <= MAX_INLINE_VIEWS && views_inline) {  // the walk reads the views from the kernarg copy
        wj = *walk;
        with_walk = true;
        walk_blocks = wj.inrow ? 0u : wj.n_blocks;  // in-row: the row workgroups of the objects' tiles do it
    }
    const dim3 grid(n_tiles + prev_blocks + fill_blocks + walk_blocks);
    WalkPlanes wp;  // (only the bytes take_walk_planes fills are read)
    NoWalkPlanes nwp;
    if (with_walk) take_walk_planes(&wj, &wp, walk_planes);
#ifndef MI_EXP_NO_NT_LOADS  // (A/B build without)
    if (PROP == 1 && c.n >= NT_LOADS_MIN_ROWS) flags |= CULL_NT_LOADS;
#endif
    // several camera views: intersects_obb over the (row, view) pairs that passed their sphere test (k_frame_pairs)
    bool multi = !with_walk && views_inline && n_views >= 2u && n_views <= MULTI_MAX_VIEWS && g_multi_view_mode != 1;
    for (uint32_t v = 0; v < n_views && multi; ++v) multi = !(views_inline->v[v].flags & VIEW_SHADOW);
#ifdef MI_EXP_FORCE_WALK_VARIANT  // (experiment build: a launch without a walk takes the walk-carrying instantiation -- what that variant costs by itself)
    if (!with_walk && views_inline && n_views <= MAX_INLINE_VIEWS) {
        MI_LAUNCH((k_frame<PROP, true, 1>), grid, dim3(256), 0, stream, c, *views_inline, (const ViewParams*)nullptr, n_views, out, seg, flags,
                  n_tiles, pa, prev_gx, prev_blocks, fill_blocks, fj, wj, changed, wp);
        return hipGetLastError();
    }
#endif
    if (multi) {
        MI_LAUNCH((k_frame_pairs<PROP>), grid, dim3(256), 0, stream, c, *views_inline, (const ViewParams*)nullptr, n_views, out, seg, flags,
                  n_tiles, pa, prev_gx, prev_blocks, fill_blocks, fj, wj, changed);
    } else if (with_walk && wj.spots) {
        MI_LAUNCH((k_frame<PROP, true, 2>), grid, dim3(256), 0, stream, c, *views_inline, (const ViewParams*)nullptr, n_views, out, seg, flags,
                  n_tiles, pa, prev_gx, prev_blocks, fill_blocks, fj, wj, changed, wp);
    } else if (with_walk) {
        MI_LAUNCH((k_frame<PROP, true, 1>), grid, dim3(256), 0, stream, c, *views_inline, (const ViewParams*)nullptr, n_views, out, seg, flags,
                  n_tiles, pa, prev_gx, prev_blocks, fill_blocks, fj, wj, changed, wp);
    } else if (n_views <= MAX_INLINE_VIEWS && views_inline) {
        MI_LAUNCH((k_frame<PROP, true, 0>), grid, dim3(256), 0, stream, c, *views_inline, (const ViewParams*)nullptr, n_views, out, seg, flags,
                  n_tiles, pa, prev_gx, prev_blocks, fill_blocks, fj, wj, changed, nwp);
    } else {
        ViewSet dummy = {};
        MI_LAUNCH((k_frame<PROP, false, 0>), grid, dim3(256), 0, stream, c, dummy, d_views, n_views, out, seg, flags, n_tiles, pa, prev_gx,
                  prev_blocks, fill_blocks, fj, wj, changed, nwp);
    }
    return hipGetLastError();
}
// The frame over the world-sphere column (k_frame_sph): cull only (changed == nullptr) or the changed-rows frame.
hipError_t launch_frame_sph(const Columns& c, const ViewSet* views_inline, const ViewParams* d_views, uint32_t n_views, const VisibilityOut& out,
                            const SegOut& seg, uint32_t flags, const CompactFastArgs* prev, const ClusterFillJob* fill, const ClusterWalkJob* walk,
                            hipStream_t stream, const uint8_t* changed, float* sph, const uint64_t* stale_bits, const uint8_t* stale_bytes,
                            bool all_stale, WalkPlanesHost walk_planes) {
    if (c.n == 0) return hipSuccess;
    if (!c.row_summary) return hipErrorInvalidValue;  // (the kernel loads it unconditionally: row_summary_ensure comes first)
    const uint32_t n_tiles = blocks_for(c.n);
    CompactFastArgs pa{};
    uint32_t prev_gx = 1, prev_blocks = 0;
    if (prev && prev->n && prev->n_segments) {
        pa = *prev;
        prev_gx = compact_fast_gx(pa.n, false);
        prev_blocks = prev_gx * pa.n_segments;
    }
    ClusterFillJob fj{};
    uint32_t fill_blocks = 0;
    if (fill) {
        fj = *fill;
        fill_blocks = CLUSTER_FILL_RIDE_BLOCKS;
    }
    ClusterWalkJob wj{};
    uint32_t walk_blocks = 0;
    bool with_walk = false;
    if (walk && walk->n_blocks && n_views <= MAX_INLINE_VIEWS && views_inline) {
        wj = *walk;
        with_walk = true;
        walk_blocks = wj.inrow ? 0u : wj.n_blocks;  // in-row: the row workgroups of the objects' tiles do it
    }
    SphereArgs sa{reinterpret_cast<float4*>(sph), stale_bits, stale_bytes, all_stale ? 1u : 0u};
    const dim3 grid(n_tiles + prev_blocks + fill_blocks + walk_blocks);
    const bool inl = n_views <= MAX_INLINE_VIEWS && views_inline;
    ViewSet dummy = {};
    const ViewSet& vsr = inl ? *views_inline : dummy;
    const ViewParams* dv = inl ? nullptr : d_views;
    WalkPlanes wp;  // (as in launch_frame)
    NoWalkPlanes nwp;
    if (with_walk) take_walk_planes(&wj, &wp, walk_planes);
#ifndef MI_EXP_NO_SPH_NT
    if (c.n >= NT_LOADS_MIN_ROWS) flags |= CULL_NT_LOADS;  // (the dense fetch of the survivors' GlobalTransforms)
#endif
#define MI_SPH_LAUNCH(P, I, W, PL) \
    MI_LAUNCH((k_frame_sph<P, I, W>), grid, dim3(256), 0, stream, c, vsr, dv, n_views, out, seg, flags, n_tiles, pa, prev_gx, prev_blocks, \
              fill_blocks, fj, wj, changed, sa, PL)
    const bool pairs = !with_walk && inl && n_views >= 2u && n_views <= MULTI_MAX_VIEWS && g_multi_view_mode != 1;
    if (pairs) {
        if (changed) MI_LAUNCH((k_frame_sph_pairs<true>), grid, dim3(256), 0, stream, c, vsr, dv, n_views, out, seg, flags, n_tiles, pa, prev_gx, prev_blocks, fill_blocks, fj, wj, changed, sa);
        else MI_LAUNCH((k_frame_sph_pairs<false>), grid, dim3(256), 0, stream, c, vsr, dv, n_views, out, seg, flags, n_tiles, pa, prev_gx, prev_blocks, fill_blocks, fj, wj, changed, sa);
    } else if (changed) {
        if (with_walk && wj.spots) MI_SPH_LAUNCH(true, true, 2, wp);
        else if (with_walk) MI_SPH_LAUNCH(true, true, 1, wp);
        else if (inl) MI_SPH_LAUNCH(true, true, 0, nwp);
        else MI_SPH_LAUNCH(true, false, 0, nwp);
    } else {
        if (with_walk && wj.spots) MI_SPH_LAUNCH(false, true, 2, wp);
        else if (with_walk) MI_SPH_LAUNCH(false, true, 1, wp);
        else if (inl) MI_SPH_LAUNCH(false, true, 0, nwp);
        else MI_SPH_LAUNCH(false, false, 0, nwp);
    }
#undef MI_SPH_LAUNCH
    return hipGetLastError();
}
// Behind k_frame_cells: the frame's finished masks -> the VisibleEntities lists and the next frame's starting masks.
//   lists   per view, the rows whose bit is set, ascending (= sorted by Entity, visibility/mod.rs:861-874): what k_compact_fast writes
//           from the frame kernels' wave counts -- here straight from the masks
//   copy    the masks themselves, into the set the next frame will write (it changes only the bits that change)
// Two short launches, no wait inside either:
//   k_cells_blocks (groups <= 256, views)  a workgroup walks a contiguous run of whole 64-word blocks (a wave per block, a lane per
//       word): the copy, every block's population as its exclusive prefix INSIDE the run (blk_pre), the run's total (grp_tot)
//   k_cells_lists  (blocks / 4, views)     a wave per block: its base = the totals of the runs in front (<= 255 numbers: four per lane,
//       one round trip) + blk_pre, then the expansion.  A wave per block because the rows a view sees are often neighbours in row
//       order too: with a run per workgroup one workgroup expanded 50 000 rows while fifteen had none (22 of 26 us at 1 M rows).
// Words without a set bit cost a ballot; the expansion visits only the others.
constexpr uint32_t CELLS_FIN_MAX_BLKS = 4096;  // blocks per run the LDS holds (16 KB): 256 runs x 4096 blocks x 4096 rows > 2^32
__global__ void __launch_bounds__(256) k_cells_blocks(CellsFinishArgs f) {
    const uint32_t v = blockIdx.y, grp = blockIdx.x, lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t b0 = grp * f.blks_per < f.n_blks ? grp * f.blks_per : f.n_blks, b1 = b0 + f.blks_per < f.n_blks ? b0 + f.blks_per : f.n_blks;
    const uint64_t* mask = f.out.bitmask + (size_t)v * f.out.words_per_view + f.out.word_offset;
    __shared__ uint32_t s_blk[CELLS_FIN_MAX_BLKS];
    for (uint32_t b = b0 + wv; b < b1; b += 4u) {  // (wave-uniform)
        const uint32_t word = b * 64u + lane;
        const unsigned long long m = word < f.n_words ? mask[word] : 0ull;
        if (f.copy_to && word < f.n_words) f.copy_to[(size_t)v * f.copy_words_per_view + word] = m;
        uint32_t pc = (uint32_t)__popcll(m);
#pragma unroll
        for (uint32_t off = 32u; off; off >>= 1) pc += __shfl_xor(pc, off, 64);
        if (lane == 0u) s_blk[b - b0] = pc;
    }
    if (!f.out_rows) return;  // (the copy only: the lists are the general compaction's)
    __syncthreads();
    if (wv != 0u) return;
    const uint32_t nb = b1 - b0;
    uint32_t carry = 0;
    for (uint32_t i0 = 0; i0 < nb; i0 += 64u) {  // (wave-uniform)
        const uint32_t i = i0 + lane;
        const uint32_t cnt = i < nb ? s_blk[i] : 0u;
        uint32_t incl = cnt;
#pragma unroll
        for (uint32_t off = 1; off < 64u; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off, 64);
            if (lane >= off) incl += up;
        }
        if (i < nb) f.blk_pre[(size_t)v * f.n_blks + b0 + i] = carry + incl - cnt;
        carry += __shfl(incl, 63, 64);
    }
    if (lane == 0u) f.grp_tot[(size_t)v * CELLS_FIN_GROUPS + grp] = carry;
}
__device__ __forceinline__ void cells_lists_block(const CellsFinishArgs& f, uint32_t bx, uint32_t v) {
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t b = bx * 4u + wv;
    if (b >= f.n_blks) return;  // (wave-uniform)
    const uint64_t* mask = f.out.bitmask + (size_t)v * f.out.words_per_view + f.out.word_offset;
    const uint32_t word = b * 64u + lane;
    const unsigned long long m = word < f.n_words ? mask[word] : 0ull;
    const uint32_t grp = b / f.blks_per;
    const bool last = b == f.n_blks - 1u;  // this wave also leaves the list's length
    const uint32_t upto = last ? f.n_groups : grp;
    uint32_t mine = 0, all = 0;
#pragma unroll
    for (uint32_t k = 0; k < CELLS_FIN_GROUPS / 64u; ++k) {
        const uint32_t g = lane + 64u * k;
        const uint32_t t = g < upto ? f.grp_tot[(size_t)v * CELLS_FIN_GROUPS + g] : 0u;
        mine += g < grp ? t : 0u;
        all += t;
    }
#pragma unroll
    for (uint32_t off = 32u; off; off >>= 1) {
        mine += __shfl_xor(mine, off, 64);
        all += __shfl_xor(all, off, 64);
    }
    if (last && lane == 0u) f.seg_totals[v] = all;
    const uint32_t pre = f.blk_pre[(size_t)v * f.n_blks + b];  // (requested with the loads above: one round trip for all of them)
    unsigned long long nz = __ballot(m != 0ull);
    if (nz == 0ull) return;
    const uint32_t base = mine + pre;
    const uint32_t pc = (uint32_t)__popcll(m);
    uint32_t incl = pc;
#pragma unroll
    for (uint32_t off = 1; off < 64u; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off, 64);
        if (lane >= off) incl += up;
    }
    const uint32_t excl = incl - pc;
    uint32_t* out = f.out_rows + (size_t)v * f.seg_stride;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const uint32_t m_lo = (uint32_t)m, m_hi = (uint32_t)(m >> 32);
    while (nz) {  // (nz is wave-uniform: j lives in a scalar register and the three values come with v_readlane, not through LDS)
        const int j = __builtin_amdgcn_readfirstlane(__ffsll((long long)nz) - 1);
        nz &= nz - 1ull;
        const unsigned long long mj = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)m_hi, j) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)m_lo, j);
        const uint32_t oj = (uint32_t)__builtin_amdgcn_readlane((int)excl, j);
        if ((mj >> lane) & 1ull) out[base + oj + (uint32_t)__popcll(mj & lt)] = (b * 64u + (uint32_t)j) * 64u + lane;
    }
}
__global__ void __launch_bounds__(256) k_cells_lists(CellsFinishArgs f) { cells_lists_block(f, blockIdx.x, blockIdx.y); }
hipError_t launch_cells_lists(const CellsFinishArgs& f, uint32_t n_views, hipStream_t stream) {
    if (f.n_words == 0 || n_views == 0 || !f.out_rows) return hipSuccess;
    MI_LAUNCH(k_cells_lists, dim3((f.n_blks + 3u) / 4u, n_views), dim3(256), 0, stream, f);
    return hipGetLastError();
}
hipError_t launch_cells_finish(CellsFinishArgs& f, uint32_t n_views, bool lists_now, hipStream_t stream, void (*mark)(void*, uint32_t), void* mark_ctx) {
    if (f.n_words == 0 || n_views == 0) return hipSuccess;
    uint32_t groups = (f.n_blks + 15u) / 16u;  // 16 blocks (65 536 rows) a run while that keeps it to CELLS_FIN_GROUPS runs
    if (groups > CELLS_FIN_GROUPS) groups = CELLS_FIN_GROUPS;
    if (groups == 0) groups = 1;
    if (f.max_groups && groups > f.max_groups) groups = f.max_groups;  // (test hook: long runs -- the shape of tables beyond 16.7 M rows -- on any table)
    f.blks_per = (f.n_blks + groups - 1u) / groups;
    f.n_groups = (f.n_blks + f.blks_per - 1u) / f.blks_per;
    if (f.blks_per > CELLS_FIN_MAX_BLKS) return hipErrorInvalidValue;  // (more than 2^32 rows)
    if (mark) mark(mark_ctx, K_COMPACT_COUNT);
    MI_LAUNCH(k_cells_blocks, dim3(f.n_groups, n_views), dim3(256), 0, stream, f);
    if (f.out_rows && lists_now) {
        if (mark) mark(mark_ctx, K_COMPACT_FAST);
        MI_LAUNCH(k_cells_lists, dim3((f.n_blks + 3u) / 4u, n_views), dim3(256), 0, stream, f);
    }
    if (mark) mark(mark_ctx, K_NUM_KERNELS);
    return hipGetLastError();
}
// the test: a thread per cell
hipError_t launch_cells_test(const CellsOrder& o, const ViewSet* views_inline, const ViewParams* d_views, uint32_t n_views, const CellsZero& z,
                             const CellsWork& work, hipStream_t stream) {
    if (o.n_waves == 0) return hipSuccess;
    CellsFrameArgs a{o, z, work.list, work.n, work.n_next, work.fresh};
    const bool inl = n_views <= MAX_INLINE_VIEWS && views_inline;
    ViewSet dummy = {};
    const dim3 grid((o.n_waves + 1023u) / 1024u);
    if (inl) MI_LAUNCH((k_cells_test<true>), grid, dim3(1024), 0, stream, *views_inline, (const ViewParams*)nullptr, n_views, a);
    else MI_LAUNCH((k_cells_test<false>), grid, dim3(1024), 0, stream, dummy, d_views, n_views, a);
    return hipGetLastError();
}
// the cells it listed: a wave each, grid-stride (at most CELLS_MAX_TILES workgroups: twice what the chip holds at once)
hipError_t launch_frame_cells(const Columns& c, const CellsOrder& o, const ViewSet* views_inline, const ViewParams* d_views, uint32_t n_views,
                              const VisibilityOut& out, const CellsZero& z, const CellsWork& work, const CompactFastArgs* prev, const ClusterFillJob* fill,
                              hipStream_t stream, const CellsFinishArgs* lists, uint32_t lists_views) {
    if (c.n == 0) return lists ? launch_cells_lists(*lists, lists_views, stream) : hipSuccess;
    CellsFrameArgs a{o, z, work.list, work.n, work.n_next, work.fresh};
    const bool inl = n_views <= MAX_INLINE_VIEWS && views_inline;
    ViewSet dummy = {};
    const ViewSet& vsr = inl ? *views_inline : dummy;
    const ViewParams* dv = inl ? nullptr : d_views;
    const uint32_t want = (o.n_waves + 3u) / 4u;
    const uint32_t n_tiles = want < CELLS_MAX_TILES ? want : CELLS_MAX_TILES;
    CompactFastArgs pa{};
    uint32_t prev_gx = 1, prev_blocks = 0;
    if (prev && prev->n && prev->n_segments) {
        pa = *prev;
        prev_gx = compact_fast_gx(pa.n, false);
        prev_blocks = prev_gx * pa.n_segments;
    }
    ClusterFillJob fj{};
    uint32_t fill_blocks = 0;
    if (fill) {
        fj = *fill;
        fill_blocks = CLUSTER_FILL_RIDE_BLOCKS;
    }
    CellsFinishArgs lf{};
    uint32_t n_lists = 0;
    if (lists && lists->out_rows && lists->n_words && lists_views) {
        lf = *lists;
        n_lists = ((lf.n_blks + 3u) / 4u) * lists_views;
    }
    const dim3 grid(n_tiles + prev_blocks + fill_blocks + n_lists);
    if (inl) MI_LAUNCH((k_frame_cells<true>), grid, dim3(256), 0, stream, c, vsr, dv, n_views, out, a, n_tiles, pa, prev_gx, prev_blocks, fill_blocks, fj, n_lists,
                       lists_views, lf);
    else MI_LAUNCH((k_frame_cells<false>), grid, dim3(256), 0, stream, c, vsr, dv, n_views, out, a, n_tiles, pa, prev_gx, prev_blocks, fill_blocks, fj, n_lists,
                   lists_views, lf);
    return hipGetLastError();
}
hipError_t launch_flat_propagate_cull(const Columns& c, const ViewSet* views_inline, const ViewParams* d_views,
                                      uint32_t n_views, const VisibilityOut& out, const SegOut& seg, uint32_t flags,
                                      const CompactFastArgs* prev, const ClusterFillJob* fill, const ClusterWalkJob* walk, hipStream_t stream,
                                      const uint8_t* changed, WalkPlanesHost walk_planes) {
    if (changed) return launch_frame<2>(c, views_inline, d_views, n_views, out, seg, flags | CULL_BEGIN_FRAME, prev, fill, walk, stream, changed, walk_planes);
    return launch_frame<1>(c, views_inline, d_views, n_views, out, seg, flags | CULL_BEGIN_FRAME, prev, fill, walk, stream, nullptr, walk_planes);
}
hipError_t launch_cull(const Columns& c, const ViewSet* views_inline, const ViewParams* d_views, uint32_t n_views,
                       const VisibilityOut& out, const SegOut& seg, uint32_t flags, const CompactFastArgs* prev,
                       const ClusterFillJob* fill, const ClusterWalkJob* walk, hipStream_t stream, WalkPlanesHost walk_planes) {
    return launch_frame<0>(c, views_inline, d_views, n_views, out, seg, flags, prev, fill, walk, stream, nullptr, walk_planes);
}
hipError_t launch_level0_propagate(const Columns& c, uint32_t n_level0, const uint8_t* node_flags,
                                   const uint8_t* changed, const uint8_t* tree_bytes, bool all_dirty,
                                   bool static_opt, hipStream_t stream) {
    if (n_level0 == 0) return hipSuccess;
    MI_LAUNCH(k_level0_propagate, dim3(blocks_for(n_level0)), dim3(256), 0, stream, c, n_level0, node_flags,
                       changed, tree_bytes, all_dirty, static_opt);
    return hipGetLastError();
}
hipError_t launch_vis_begin(const Columns& c, hipStream_t stream) {
    if (c.n == 0) return hipSuccess;
    MI_LAUNCH(k_vis_begin, dim3(blocks_for(c.n)), dim3(256), 0, stream, c);
    return hipGetLastError();
}
hipError_t launch_vis_end(const Columns& c, hipStream_t stream) {
    if (c.n == 0) return hipSuccess;
    MI_LAUNCH(k_vis_end, dim3(blocks_for(c.n)), dim3(256), 0, stream, c);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// VisibleEntities compaction (visibility/mod.rs:852-874): stable stream compaction of rows in
// ascending Entity-key order, so the emitted list is already sorted like sort_unstable() leaves it.
// count -> scan -> scatter; grid.y enumerates (view, class) segments.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool compact_pred(const CompactArgs& a, uint32_t pos, uint32_t view, uint32_t class_bit,
                                             uint32_t* row_out) {
    if (pos >= a.n) return false;
    const uint32_t row = a.order ? a.order[pos] : pos;
    *row_out = row;
    const uint64_t word = a.bitmask[view * a.words_per_view + a.word_offset + (row >> 6)];
    if (!((word >> (row & 63u)) & 1ull)) return false;
    const uint32_t cm = a.class_mask ? a.class_mask[row] : 1u;
    return ((cm >> class_bit) & 1u) != 0;
}

__global__ void __launch_bounds__(256) k_compact_count(CompactArgs a) {
    const uint32_t seg = blockIdx.y, view = seg / a.n_classes, class_bit = a.class_bits[seg % a.n_classes];
    const uint32_t base = blockIdx.x * COMPACT_BLOCK_ROWS;
    uint32_t cnt = 0;
    for (uint32_t j = 0; j < COMPACT_BLOCK_ROWS; j += 256u) {
        uint32_t row;
        const bool p = compact_pred(a, base + j + threadIdx.x, view, class_bit, &row);
        cnt += __popcll(__ballot(p));
    }
    __shared__ uint32_t wsum[4];
    if ((threadIdx.x & 63u) == 0) wsum[threadIdx.x >> 6] = cnt;  // cnt is wave-uniform
    __syncthreads();
    if (threadIdx.x == 0) a.block_counts[(size_t)seg * a.n_blocks + blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// One workgroup: exclusive scan of block_counts in segment-major order; per-segment totals and bases.
__global__ void __launch_bounds__(1024) k_compact_scan(CompactArgs a) {
    __shared__ uint32_t part[1024];
    __shared__ uint64_t carry;
    const uint32_t segments = a.n_views * a.n_classes;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t seg = 0; seg < segments; ++seg) {
        uint32_t* bc = a.block_counts + (size_t)seg * a.n_blocks;
        uint32_t seg_running = 0;
        for (uint32_t b0 = 0; b0 < a.n_blocks; b0 += 1024u) {
            const uint32_t b = b0 + threadIdx.x;
            const uint32_t v = b < a.n_blocks ? bc[b] : 0u;
            part[threadIdx.x] = v;
            __syncthreads();
            for (uint32_t off = 1; off < 1024u; off <<= 1) {  // Hillis-Steele inclusive scan
                uint32_t add = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
                __syncthreads();
                part[threadIdx.x] += add;
                __syncthreads();
            }
            if (b < a.n_blocks) bc[b] = seg_running + part[threadIdx.x] - v;  // exclusive, within the segment
            seg_running += part[1023];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            a.seg_totals[seg] = seg_running;
            a.seg_bases[seg] = carry;
            carry += seg_running;
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) k_compact_scatter(CompactArgs a) {
    const uint32_t seg = blockIdx.y, view = seg / a.n_classes, class_bit = a.class_bits[seg % a.n_classes];
    const uint32_t base = blockIdx.x * COMPACT_BLOCK_ROWS;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    __shared__ uint32_t wcnt[4];
    uint64_t out_base = a.seg_bases[seg] + a.block_counts[(size_t)seg * a.n_blocks + blockIdx.x];
    for (uint32_t j = 0; j < COMPACT_BLOCK_ROWS; j += 256u) {
        uint32_t row = 0;
        const bool p = compact_pred(a, base + j + threadIdx.x, view, class_bit, &row);
        const unsigned long long m = __ballot(p);
        if (lane == 0) wcnt[wv] = __popcll(m);
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) { const uint32_t cw = wcnt[k]; before += k < wv ? cw : 0u; total += cw; }
        if (p) {
            const uint64_t dst = out_base + before + __popcll(m & ((1ull << lane) - 1ull));
            a.out_rows[dst] = row;
            a.out_keys[dst] = a.entity_keys ? a.entity_keys[row] : (uint64_t)row;
        }
        out_base += total;
        __syncthreads();
    }
}

hipError_t launch_compact(const CompactArgs& a, hipStream_t stream, void (*mark)(void*, uint32_t), void* mctx) {
    if (a.n == 0) return hipSuccess;
    const dim3 grid(a.n_blocks, a.n_views * a.n_classes);
    if (mark) mark(mctx, K_COMPACT_COUNT);
    MI_LAUNCH(k_compact_count, grid, dim3(256), 0, stream, a);
    if (mark) mark(mctx, K_COMPACT_SCAN);
    MI_LAUNCH(k_compact_scan, dim3(1), dim3(1024), 0, stream, a);
    if (mark) mark(mctx, K_COMPACT_SCATTER);
    MI_LAUNCH(k_compact_scatter, grid, dim3(256), 0, stream, a);
    if (mark) mark(mctx, K_NUM_KERNELS);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Single-launch compaction (rows already in ascending Entity-key order).  grid = (ceil(words/64), segments).
// A workgroup owns 64 mask words (4096 rows).  Its base is the sum of the u8 wave counts of all preceding
// words of the segment (<= n/64 bytes, L2 resident, read as 16-byte vectors and summed with v_sad_u8), so
// there is no scan kernel, no look-back chain and no atomics; the list order is the row order.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_compact_fast(CompactFastArgs a) {
    if (a.signal && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
        __hip_atomic_store(a.signal, a.signal_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    compact_fast_block<true>(a, blockIdx.x, blockIdx.y, gridDim.x);
}

hipError_t launch_compact_fast(const CompactFastArgs& a, hipStream_t stream) {
    if (a.n == 0 || a.n_segments == 0) return hipSuccess;
    MI_LAUNCH(k_compact_fast, dim3(compact_fast_gx(a.n, true), a.n_segments), dim3(256), 0, stream, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_clear_u32(uint32_t* p, uint64_t n_words) {
    for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * 256ull) p[i] = 0u;
}
hipError_t launch_clear_u32(uint32_t* p, uint64_t n_words, hipStream_t stream) {
    if (n_words == 0) return hipSuccess;
    const uint32_t blocks = (uint32_t)((n_words + 255ull) / 256ull > 2048ull ? 2048ull : (n_words + 255ull) / 256ull);
    MI_LAUNCH(k_clear_u32, dim3(blocks), dim3(256), 0, stream, p, n_words);
    return hipGetLastError();
}
__global__ void __launch_bounds__(256) k_bytes_to_bits(const uint8_t* __restrict__ bytes, uint32_t n, uint64_t* bits) {
    const uint32_t row = blockIdx.x * 256u + threadIdx.x;
    const bool b = row < n && bytes[row] != 0;
    const unsigned long long m = __ballot(b);
    if ((threadIdx.x & 63u) == 0 && (row & ~63u) < n) bits[row >> 6] = m;
}
hipError_t launch_bytes_to_bits(const uint8_t* bytes, uint32_t n, uint64_t* bits, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    MI_LAUNCH(k_bytes_to_bits, dim3(blocks_for(n)), dim3(256), 0, stream, bytes, n, bits);
    return hipGetLastError();
}

}  // namespace mi
