// kernels.h -- launch wrappers of the gfx950 kernels (implemented in kernels_*.hip).
// Host-callable; every wrapper only enqueues on `stream` and returns the hipError_t of the launch.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

namespace mi {

// Kernel timing with the dispatch packet's own timestamps (what rocprofv3 --kernel-trace reports): when the
// context arms g_launch_timer, the next MI_LAUNCH goes through hipExtLaunchKernelGGL with (start, stop) events
// bound to that one dispatch; hipEventElapsedTime(start, stop) is then the kernel's duration without the
// marker-packet and inter-dispatch gaps a hipEventRecord bracket would add.
struct LaunchTimer {
    hipEvent_t start, stop;
};
extern thread_local const LaunchTimer* g_launch_timer;
// Orders one wave's own LDS traffic (cross-lane exchange through a wave-private LDS region): waits for the
// wave's outstanding LDS operations and stops the compiler from moving LDS accesses across it.  The fences name the
// LDS address space: a plain workgroup fence also drains vmcnt -- every global load AND STORE in flight -- which put a full
// memory round trip in front of each transpose that follows a store.
#define MI_WAVE_LDS_SYNC()                                             \
    do {                                                               \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); \
        __builtin_amdgcn_wave_barrier();                               \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); \
    } while (0)
// Workgroup barrier that orders LDS traffic only: global loads issued before it stay in flight across it (__syncthreads()
// drains the load counter).
#define MI_WG_LDS_BARRIER()                                            \
    do {                                                               \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); \
        __builtin_amdgcn_s_barrier();                                  \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); \
    } while (0)
#if defined(__HIPCC__)
// A kernel argument read where it is used instead of where the compiler likes to read it (the kernel's first block: every argument
// word then sits in an SGPR for the whole kernel, and the fused hierarchy frame's -- 1.3 KB of views, masks, zeroing and
// compaction arguments on top of the tile kernel's own -- were 68 spilled SGPRs, v_readlane / v_writelane in every tile).  The pointer into
// the kernarg segment goes through an empty asm statement, so nothing behind it can be hoisted above that point; the loads stay
// scalar (the pointer is uniform and the address space is recovered after inlining).
template <class T>
__device__ __forceinline__ const T& kernarg_late(uint32_t offset) {
    auto p = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *(const T*)(p + offset);
}
constexpr uint32_t kernarg_up(size_t x) { return (uint32_t)((x + 7u) & ~(size_t)7u); }

#endif
#define MI_LAUNCH(kernel, grid, block, lds, stream, ...)                                                        \
    do {                                                                                                        \
        const ::mi::LaunchTimer* lt_ = ::mi::g_launch_timer;                                                    \
        ::mi::g_launch_timer = nullptr;                                                                         \
        if (lt_) hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, lt_->start, lt_->stop, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                 \
    } while (0)

// One view's constants, read through scalar loads (uniform across the wave).  Same layout as mi_view.
struct ViewParams {
    float planes[24];       // 6 x (nx,ny,nz,d)
    uint32_t layer_mask;    // RenderLayers bits of the view
    uint32_t flags;         // MI_VIEW_FLAG_*
    float position[3];      // origin for VisibilityRange distances
    float light_sphere[4];  // point / spot light (translation, range)
    uint32_t layer_mask_hi;  // RenderLayers 32..63 of the view
    uint32_t pad[2];
};
static_assert(sizeof(ViewParams) == 144, "ViewParams layout");
constexpr uint32_t SPHERE_AT_TRANSLATION = 0x7FC0A11Du;  // MI_SPHERE_AT_TRANSLATION (public header): half.y of a Sphere row whose centre is the row's GlobalTransform translation
// MI_VIEW_FLAG_* (public header)
constexpr uint32_t VIEW_NO_CPU_CULLING = 0x01u, VIEW_SHADOW = 0x02u, VIEW_SKIP_NEAR = 0x04u, VIEW_TEST_FAR = 0x08u,
                   VIEW_LIGHT_SPHERE = 0x10u, VIEW_RANGES = 0x20u, VIEW_RANGES_NO_ORIGIN = 0x40u;

// Device-resident component columns of one context (all pointers device memory).
struct Columns {
    uint32_t n;                 // live rows
    const float* translation;   // 3n
    const float* rotation;      // 4n
    const float* scale;         // 3n
    float* global;              // 12n
    const float* aabb_center;   // 3n
    const float* aabb_half;     // 3n
    const uint8_t* flags;       // n
    const uint32_t* layer_mask; // n
    const uint32_t* layer_mask_hi;  // n: RenderLayers 32..63, or nullptr = no row has any (mi_upload_render_layers_hi)
    uint8_t* view_visibility;   // n
    const float* range_start_end;  // 2n (VisibilityRange start_margin.start, end_margin.end) or nullptr = no
                                   // VisibleEntityRanges resource
    uint64_t* g_changed_bits;   // ceil(n/64) words: GlobalTransform change tick bumped
    uint64_t* vv_changed_bits;  // ceil(n/64) words: ViewVisibility change tick bumped
    uint32_t changed_gen;       // the Transform change column holds STAMPS, see row_changed()
    const uint32_t* row_summary;  // 8 words per 64 rows (RowSummary below); allocated whenever a frame kernel runs
    uint32_t row_summary_on;      // 0 = out of date / switched off: every row reads its own columns
};

// Per 64 aligned rows, what the frame kernels would otherwise read per row although it rarely differs from row to row: Aabb (the
// entities of a mesh are spawned together: one Aabb for all of them), the flags byte and the RenderLayers mask.  A wave fetches the
// 32 bytes with scalar loads; where a bit says "uniform" its lanes take the values from there and issue no loads on the column:
// 29 of the ~119 B a row of the all-dirty frame costs (24 Aabb + 1 flags + 4 layers).  Waves whose rows differ read the columns as
// before.  (A wave counts as uniform in its RenderLayers only if none of its rows has a layer above 31: the summary holds one word.)
// Written by k_row_summary from the columns themselves (after mi_upload_bounds / resize / visibility propagation, for the
// waves they touched), so the columns stay the only source of truth.
//   words 0-2 Aabb centre, 3-5 half extents (valid iff ROWSUM_UNIFORM_AABB), 6 RenderLayers mask, 7 = flags byte | ROWSUM_* bits
constexpr uint32_t ROWSUM_WORDS = 8u, ROWSUM_UNIFORM_AABB = 0x80000000u, ROWSUM_UNIFORM_FLAGS = 0x40000000u;
constexpr uint32_t ROWSUM_PART_AABB = 1u, ROWSUM_PART_FLAGS = 2u;

// The per-row Transform change byte: 0 = unchanged, 1 = changed (bulk uploads, rows never propagated), g in 2..255 = changed iff g is
// the context's current generation -- what the indexed uploads write.  Consuming the column is then `generation += 1` on the host
// instead of a memset dispatch behind every propagate (>= 4.3 us on this part: a third of a change-driven frame); a real memset
// remains for bulk marks and for the wrap at 255.
__host__ __device__ inline bool row_changed(uint32_t byte, uint32_t gen) { return byte == 1u || byte == gen; }
#ifdef __HIPCC__
// mark_dirty_trees for one changed row (systems.rs:111-306): TransformTreeChanged on the row and on every ancestor.  With the
// ancestor table (anc != nullptr, see ANC_DEPTH below) the ancestors come with one 64-byte load and the marks are independent
// stores; without it, or from the table's last entry on in a very deep tree, a climb along parent_idx that stops at the first row
// already marked (a stale test only lets two climbers repeat each other's stores).
__device__ __forceinline__ void mark_row_and_ancestors(uint32_t row, const uint32_t* __restrict__ parent_idx, uint8_t* marks,
                                                       const uint32_t* __restrict__ anc, uint32_t guard_n) {
    if (anc) {
        const uint4* src = reinterpret_cast<const uint4*>(anc + (size_t)row * 16u);
        const uint4 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3];
        const uint32_t a[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
        marks[row] = 1;
#pragma unroll
        for (uint32_t k = 0; k < 16u; ++k)
            if (a[k] != 0xFFFFFFFFu) marks[a[k]] = 1;
        if (a[15] == 0xFFFFFFFFu) return;
        row = a[15];  // deeper than the table: go on from its last entry (already marked: start at its parent)
        const uint32_t p = parent_idx[row];
        if (p == 0xFFFFFFFFu) return;
        row = p;
    }
    for (uint32_t guard = 0; guard < guard_n; ++guard) {
        const uint32_t p = parent_idx ? parent_idx[row] : 0xFFFFFFFFu;
        const uint8_t seen = __builtin_nontemporal_load(&marks[row]);  // (not a cached copy of a line another CU is writing)
        if (seen) break;
        marks[row] = 1;
        if (p == 0xFFFFFFFFu) break;
        row = p;
    }
}
#endif

struct VisibilityOut {
    uint64_t* bitmask;        // base of per-view bitmasks
    uint64_t words_per_view;  // stride between views, in 64-bit words
    uint64_t word_offset;     // this shard's first word inside a view's mask
};

// Views are passed BY VALUE in the kernarg segment (scalar loads, no per-frame H2D copy) when there are
// at most MAX_INLINE_VIEWS of them; more views go through a device array.
constexpr uint32_t MAX_INLINE_VIEWS = 8;
struct ViewSet {
    ViewParams v[MAX_INLINE_VIEWS];
};

// Per-(view, class) by-products of the cull pass that feed the single-launch VisibleEntities compaction:
// segment s = view * n_classes + class_slot.
//   wave_cnt[s * n_waves + wave]    number of rows of that wave (64 rows) visible in view AND in the class (u8, <= 64)
//   seg_mask[s * seg_words + wave]  their bitmask -- only written when class_mask != nullptr; without classes the
//                                   segment mask IS the view mask.
struct SegOut {
    uint32_t n_classes;          // >= 1
    uint32_t n_waves;            // padded wave count (stride of wave_cnt)
    const uint32_t* class_mask;  // nullptr = every row is in class slot 0
    uint8_t* wave_cnt;           // nullptr = compaction by-products not wanted
    uint64_t* seg_mask;
    uint64_t seg_words;
    uint8_t class_bits[32];      // class bit of each class slot
};

// mi_cull / mi_propagate_and_cull flags (MI_CULL_* in the public header)
constexpr uint32_t CULL_BEGIN_FRAME = 1u;  // fuse reset_view_visibility
constexpr uint32_t CULL_END_FRAME = 2u;    // fuse check_visibility_gpu_culling + mark_newly_hidden_entities_invisible
// (launch-internal) the Transform columns are fetched with nontemporal loads: set by the launcher for contexts whose frame does not
// fit the 256 MiB Infinity Cache by a wide margin (10 M rows x 1 view: 150 -> 132 us, x 4 views 179 -> 173; 8 M: 120 -> 107; 7 M:
// 98.9 -> 94.8).  Below that the columns of frame f are still (partly) cached when frame f + 1 reads them, and fetching them past
// the caches costs: 5 M rows x 4 views 83 -> 87.5 us, the 1.11 M-row metric frame 19.3 -> 22.6 (profiles/r05a/nt_loads_ab.txt)
constexpr uint32_t CULL_NT_LOADS = 0x100u;
constexpr uint32_t NT_LOADS_MIN_ROWS = 6u << 20;

enum KernelId : uint32_t {
    K_FLAT_PROPAGATE_CULL = 0,
    K_LEVEL0_PROPAGATE,
    K_CULL,
    K_VIS_BEGIN,
    K_VIS_END,
    K_COMPACT_COUNT,
    K_COMPACT_SCAN,
    K_COMPACT_SCATTER,
    K_COMPACT_FAST,
    K_MARK_DIRTY,
    K_PROPAGATE_TILES,
    K_CLUSTER_WALK,
    K_CLUSTER_FILL,
    K_CLEAR,
    K_INHERIT,
    K_BATCH_HIST,
    K_BATCH_PLAN,
    K_BATCH_EMIT,
    K_BATCH_SCAN,
    K_BATCH_SCATTER,
    K_BATCH_BOUNDS,
    K_BATCH_SORTED,
    K_PROPAGATE_STREAM,
    K_NUM_KERNELS
};

// ---- flat path ------------------------------------------------------------------------------
// views_inline is used when n_views <= MAX_INLINE_VIEWS (d_views may then be nullptr).
struct CompactFastArgs {
    uint32_t n;               // rows
    uint32_t n_segments;
    uint32_t n_classes;
    uint32_t n_waves;         // stride of wave_cnt per segment
    const uint8_t* wave_cnt;
    const uint64_t* seg_mask; // segment masks (class-filtered), or nullptr -> use the view masks below
    uint64_t seg_words;
    const uint64_t* bitmask;  // per-view masks
    uint64_t words_per_view, word_offset;
    uint32_t* out_rows;
    uint64_t seg_stride;      // entries reserved per segment in out_rows
    uint32_t* seg_totals;
    // Multi-GPU exchange: workgroup (0, 0) publishes `signal_value` at its start.  Any workgroup of this kernel running
    // means the frame kernel before it in the stream has completed and its masks are visible, which is all the
    // all-gather on the communication stream (hipStreamWaitValue32 on this word) needs -- and no signalling packet
    // has to sit between the frames in the compute queue (an event record or write-value packet there costs ~6 us
    // per frame: it keeps the next frame kernel from starting behind the compaction).
    uint32_t* signal;
    uint32_t signal_value;
    uint32_t tag;  // hierarchical mode: this frame's stamp on the chunk totals (never 0).  (In what was padding: the struct is an argument
                   // of every frame kernel, and those are sensitive to the size of their argument block -- profiles/r04_experiments.md 4.)
};
// The single-launch compaction's shape.  Every expanding workgroup needs the number of entries in front of its range.  It sums the wave
// counts in front of it itself (a byte per wave: 2 MB of L2 reads in all at 1 M rows) -- reads that grow with the square of the table
// (760 MB per frame of 4 views at 10 M rows with 64-word workgroups), which round 2 answered with up to ten 64-word steps per
// workgroup: 245 workgroups of ten serial steps for the whole chip, 29 - 39 us per 10 M-row compaction.  That is still the form the
// RIDERS of the frame kernels run (hier = false): they spend their time waiting in slots the frame's rows do not need, and most of it
// disappears under the rows (10 M rows: 153 against 151 us with and without the rider at one view, 176 against 163 at four).  A
// compaction that is a LAUNCH OF ITS OWN (everything that joins a pending one: downloads, batching, mi_synchronize; the sparse
// GlobalTransform list; --inline-compaction) of a table of COMPACT_HIER_MIN_ROWS rows and more is HIERARCHICAL (hier = true): the
// first workgroups of every segment ("summers", one per chunk of COMPACT_CHUNK_WORDS waves) add up their chunk's counts and publish
// (tag << 32 | total) with one 64-bit store; an expanding workgroup -- four 64-word steps -- adds the totals of the chunks in front of
// its own (polling a word whose stamp is not this frame's yet: summers have the lower workgroup ids of their segment, so they are
// dispatched first and are long done in practice) and the counts of its own chunk in front of its range (< 4 KB): 29 -> 11.7 us at
// 10 M rows x 1 view.  The table of totals lies behind seg_totals; stamps make zeroing it unnecessary.
// (Tried and not kept: the hierarchical form inside the riders -- k_frame<1, true, 1> went from 14 to 105 spilled SGPRs and the lean
// variants from 8 to 7 waves per SIMD -- and hierarchical launches INSTEAD of riders: 155.9 -> 164.5 us per frame at 10 M rows x 1
// view, 185.8 -> 202.9 at four: what rides is nearly free, what is launched is not.)
constexpr uint32_t COMPACT_CHUNK_WORDS = 4096u;  // (a summer's 4 KB of counts: one uint4 per thread; 39 chunks at 10 M rows)
constexpr uint32_t COMPACT_HIER_MIN_ROWS = 1u << 21;
__host__ __device__ inline bool compact_fast_hier(uint32_t n, bool own_launch) { return own_launch && n >= COMPACT_HIER_MIN_ROWS; }
__host__ __device__ inline uint32_t compact_fast_steps(uint32_t n, bool own_launch) { return compact_fast_hier(n, own_launch) ? 4u : 1u + (n >> 20); }
__host__ __device__ inline uint32_t compact_fast_chunks(uint32_t n, bool own_launch) {
    return compact_fast_hier(n, own_launch) ? (((n + 63u) >> 6) + COMPACT_CHUNK_WORDS - 1u) / COMPACT_CHUNK_WORDS : 0u;
}
// workgroups per segment: the summers, then the expanders
__host__ __device__ inline uint32_t compact_fast_gx(uint32_t n, bool own_launch) {
    const uint32_t per = 64u * compact_fast_steps(n, own_launch);
    return compact_fast_chunks(n, own_launch) + (((n + 63u) >> 6) + per - 1u) / per;
}
// bytes of the seg_totals buffer: the totals (padded to 8 bytes) and, in the hierarchical mode, n_chunks stamped totals per segment
inline size_t compact_fast_totals_bytes(size_t segs, uint32_t n) { return ((segs + 1u) & ~(size_t)1u) * 4u + segs * compact_fast_chunks(n, true) * 8u; }
// The clustered view's plane table on the host (x | y | z planes, four floats each; WalkPlanes below): what the context hands the launch_*
// wrappers of the kernels that carry a cluster walk.  {nullptr, 0}: none -- the walkers read the staged copy through the view's pointers.
struct WalkPlanesHost {
    const float* f;
    uint32_t n;
};
extern int g_multi_view_mode;  // (process-wide: set from MI_MULTI_VIEW when a context is created; kernels_flat.hip)
hipError_t launch_flat_propagate_cull(const Columns& c, const ViewSet* views_inline, const ViewParams* d_views,
                                      uint32_t n_views, const VisibilityOut& out, const SegOut& seg, uint32_t flags,
                                      const CompactFastArgs* prev, const struct ClusterFillJob* fill, const struct ClusterWalkJob* walk,
                                      hipStream_t stream, const uint8_t* changed = nullptr /* set: only rows with a nonzero byte are propagated */,
                                      WalkPlanesHost walk_planes = WalkPlanesHost{nullptr, 0});
// Level 0 of the hierarchy (roots + flat rows).  node_flags: bit0 = has children (nullptr = none do).
// changed: per-row Changed<Transform>|... byte (nullptr or all_dirty => every row recomputed).
// tree_bytes: TransformTreeChanged, a byte per row (only read when static_opt).
hipError_t launch_level0_propagate(const Columns& c, uint32_t n_level0, const uint8_t* node_flags,
                                   const uint8_t* changed, const uint8_t* tree_bytes, bool all_dirty,
                                   bool static_opt, hipStream_t stream);
// rows lo .. hi-1: out[row] = From(Transform) (sync_simple_transforms), for the result download that runs ahead of the frame
hipError_t launch_globals_ahead(const float* t, const float* r, const float* s, uint32_t lo, uint32_t hi, float* out, hipStream_t stream);
hipError_t launch_cull(const Columns& c, const ViewSet* views_inline, const ViewParams* d_views, uint32_t n_views,
                       const VisibilityOut& out, const SegOut& seg, uint32_t flags, const CompactFastArgs* prev,
                       const struct ClusterFillJob* fill, const struct ClusterWalkJob* walk, hipStream_t stream,
                       WalkPlanesHost walk_planes = WalkPlanesHost{nullptr, 0});
// The frame over the world-sphere column (kernels_flat.hip, k_frame_sph): camera views only, n_views <= 32.  changed == nullptr:
// cull only; else the changed-rows frame (flags carry CULL_BEGIN_FRAME).  sph: [n] (cw, sr); stale rows as ballot words or bytes.
hipError_t launch_frame_sph(const Columns& c, const ViewSet* views_inline, const ViewParams* d_views, uint32_t n_views, const VisibilityOut& out,
                            const SegOut& seg, uint32_t flags, const CompactFastArgs* prev, const struct ClusterFillJob* fill,
                            const struct ClusterWalkJob* walk, hipStream_t stream, const uint8_t* changed, float* sph, const uint64_t* stale_bits,
                            const uint8_t* stale_bytes, bool all_stale, WalkPlanesHost walk_planes = WalkPlanesHost{nullptr, 0});
constexpr uint32_t SPH_MAX_VIEWS = 32;

// ---- the static cull order (kernels_cells.hip builds it, k_frame_cells in kernels_flat.hip uses it) ----
// sum_b[w].x: low 8 bits = the flags byte where CELLS_UNIFORM_FLAGS, .y = the RenderLayers mask (layers 0..31) then
constexpr uint32_t CELLS_UNIFORM_FLAGS = 0x100u;  // the wave's slots agree in flags and RenderLayers (and none has a layer above 31)
constexpr uint32_t CELLS_UNIFORM_HALF = 0x200u;   // ... in their Aabb half extents (sum_h)
constexpr uint32_t CELLS_REJECTABLE = 0x400u;     // every slot: in the cull query, bounded, no NoFrustumCulling
struct CellsOrder {
    uint32_t n_waves;      // ceil(n / 64)
    uint32_t* perm;        // [n_waves * 64] row of each slot, 0xFFFFFFFF past the last
    float4* sph_s;         // [n_waves * 64] world spheres in slot order
    float* g_s;            // [n_waves * 64 * 12] GlobalTransforms in slot order
    uint8_t* vv_s;         // [n_waves * 64] ViewVisibility bytes in slot order (mirror of the column while the order is valid)
    uint32_t* pass_s;      // [n_waves * 64] bit v: the slot's row is set in view v's mask of the last frame over the order
    float4* sum_a;         // [n_waves] bounding sphere of the wave's world spheres (centre, radius rounded up)
    uint4* sum_b;          // [n_waves] bits | flags, RenderLayers
    float4* sum_h;         // [n_waves] half extents where uniform
    uint32_t* state;       // [n_waves] bit 0: every ViewVisibility byte of the wave is zero
};
size_t cells_sort_temp_bytes(uint32_t n);
// sph: the world-sphere column [n], current for every row.  keys / vals: [n] each; minmax: 6 words.
hipError_t launch_cells_build(const Columns& c, const float* sph, const CellsOrder& o, uint32_t* minmax, uint32_t* keys_a, uint32_t* keys_b,
                              uint32_t* vals_a, uint32_t* vals_b, void* sort_temp, size_t sort_temp_bytes, hipStream_t stream);
// The cull-only frame of a static scene over the cell order: camera views, MI_CULL_BEGIN_FRAME | MI_CULL_END_FRAME, one class segment
// per view, no visibility ranges.  Masks, wave counts and the ViewVisibility change words are ORed / added with atomics into memory
// that is zero on entry; zero[k] / zero_words[k] = what the launch zeroes for the NEXT such frame (as TreeCull::zero).
struct CellsZero {
    uint64_t* zero[3];
    uint32_t zero_words[3];
};
// behind the frame (k_cells_blocks + k_cells_lists): from the finished masks, the VisibleEntities row lists (one class segment per
// view) and the masks copied into the set the next frame writes
constexpr uint32_t CELLS_FIN_GROUPS = 256;
struct CellsFinishArgs {
    VisibilityOut out;
    uint32_t n_words, n_blks;          // mask words per view in use; 64-word blocks ((n_words + 63) / 64)
    uint32_t blks_per, n_groups;       // blocks per run, runs (set by the launch)
    uint32_t max_groups;               // 0, or a cap on the runs (mi_debug_set_static_cull_order(3): the long runs of tables beyond 16.7 M rows)
    uint32_t* blk_pre;                 // [views][n_blks] scratch: a block's exclusive prefix inside its run
    uint32_t* grp_tot;                 // [views][CELLS_FIN_GROUPS] scratch: the runs' totals
    uint64_t* copy_to;                 // the next frame's mask set, or nullptr
    uint64_t copy_words_per_view;
    uint32_t* out_rows;                // [views][seg_stride] the lists, or nullptr = copy only
    uint64_t seg_stride;
    uint32_t* seg_totals;              // [views]
};
// f.blks_per / f.n_groups are set here.  lists_now = false: only k_cells_blocks (the masks' copy and the block prefixes); the lists are then
// launch_cells_lists' (same f) or ride in the next k_frame_cells launch.
hipError_t launch_cells_finish(CellsFinishArgs& f, uint32_t n_views, bool lists_now, hipStream_t stream, void (*mark)(void*, uint32_t), void* mark_ctx);
hipError_t launch_cells_lists(const CellsFinishArgs& f, uint32_t n_views, hipStream_t stream);
// the work list between the two launches of the frame: list[n_cells] (cell, views left, summary bits | flags, RenderLayers), its counter
// and the next frame's counter
struct CellsWork {
    uint4* list;
    uint32_t *n, *n_next;
    uint32_t fresh;  // this frame's masks start from zero (else: from the frame before's, and only changed bits are touched)
};
constexpr uint32_t CELLS_MAX_TILES = 8192;
hipError_t launch_cells_test(const CellsOrder& o, const ViewSet* views_inline, const ViewParams* d_views, uint32_t n_views, const CellsZero& z,
                             const CellsWork& work, hipStream_t stream);
hipError_t launch_frame_cells(const Columns& c, const CellsOrder& o, const ViewSet* views_inline, const ViewParams* d_views, uint32_t n_views,
                              const VisibilityOut& out, const CellsZero& z, const CellsWork& work, const CompactFastArgs* prev,
                              const struct ClusterFillJob* fill, hipStream_t stream, const CellsFinishArgs* lists = nullptr /* the previous such
                              frame's deferred lists ride in the launch */, uint32_t lists_views = 0);
hipError_t launch_row_summary(const Columns& c, uint32_t first_wave, uint32_t n_waves, uint32_t parts, uint32_t* summary, hipStream_t stream);
hipError_t launch_upload_trs(const float* pinned_src, float* t, float* r, float* s, uint32_t first_row, uint32_t n,
                             hipStream_t stream);
// mark_bytes != nullptr: the rows also climb to their roots setting TransformTreeChanged there (= k_mark_dirty for these rows), and
// clear_words (the other half of the marks) is zeroed
hipError_t launch_upload_trs_indexed(const uint32_t* rows, const float* t_src, const float* r_src, const float* s_src, uint32_t n, float* t,
                                     float* r, float* s, uint8_t* changed, uint32_t changed_gen, hipStream_t stream,
                                     const uint32_t* parent_idx = nullptr, uint8_t* mark_bytes = nullptr, uint32_t* clear_words = nullptr,
                                     uint32_t n_clear_words = 0, const uint32_t* anc = nullptr,
                                     float* g_ahead = nullptr /* pinned: entry i's From(Transform), 48 B each, written over PCIe ... */,
                                     bool g_reversed = false /* ... at slot n-1-i instead of i */);
hipError_t launch_popcount_words(const uint64_t* bits, uint32_t n_rows, uint8_t* cnt, hipStream_t stream);
hipError_t launch_gather_global(const uint32_t* rows, const uint32_t* total, uint32_t capacity, const float* g, float* out,
                                hipStream_t stream);
hipError_t launch_gather_mesh_inputs(const uint32_t* rows, const uint32_t* total, uint32_t capacity, const Columns& c, float* out_wfl,
                                     float* out_cull, hipStream_t stream);
// One frame's results packed by ONE launch into a window of the pinned staging arena (mapped host memory), so that the host
// needs one wait instead of one for the counts and one for the lists.  Sections follow each other at 256-byte boundaries in
// this order, a section being present iff its pointer is set and its count fits the caller's capacity: changed rows,
// changed GlobalTransforms (48 B each), the visible-row lists in the caller's order, cluster offsets (C + 1), cluster counts (6 C),
// cluster indices.
// header: [0] changed, [2..3] cluster total (u64), [4] farthest_z (f32 bits), [5] 1 = payload written, 0 = it did not fit
// `payload_bytes` (or the cluster list overflowed its device buffer): the host falls back to separate copies; [6] 1 = the changed
// sections were left out (more than big_rows rows: the host fetches them by DMA); [8 + i] entries of list i.
constexpr uint32_t PACK_MAX_LISTS = 16;  // MI_RESULTS_MAX_LISTS
struct PackResultsJob {
    const uint32_t* changed_total;   // nullptr: no changed section
    const uint32_t* changed_rows;
    const float* g;                  // nullptr: rows only
    const uint64_t* cluster_total;   // nullptr: no cluster sections
    const uint32_t* cluster_offsets;
    const uint32_t* cluster_counts;
    const uint32_t* cluster_indices; // nullptr: offsets and counts only
    const float* farthest_z;
    uint32_t n_clusters;
    uint32_t want_changed_rows;
    uint32_t changed_capacity, n_lists;
    uint32_t big_rows;  // more changed rows than this: the rows / GlobalTransform sections are left to the DMA engine (header[6] = 1)
    uint64_t cluster_capacity, cluster_indices_alloc;
    uint32_t* header;
    uint8_t* payload;
    uint64_t payload_bytes;
    // VisibleEntities lists: a device count (nullptr = an empty list: no row carries the class) and the rows
    const uint32_t* list_total[PACK_MAX_LISTS];
    const uint32_t* list_rows[PACK_MAX_LISTS];
    const uint64_t* list_base[PACK_MAX_LISTS];  // device word: first entry of the list inside list_rows (nullptr = 0)
    uint32_t list_capacity[PACK_MAX_LISTS];
};
constexpr uint32_t PACK_HEADER_BYTES = 256;
// Above this many changed rows (3.4 MB of rows + matrices) the copy engine moves them faster than the packing kernel's stores over
// PCIe do (55 against 29 GB/s on this box), even with the extra wait: the kernel leaves the two sections out and says so.
constexpr uint32_t PACK_BIG_ROWS = 65536;
constexpr uint64_t PACK_WINDOW_BYTES = (uint64_t)8 << 20;  // copy-out mode: bigger frames are byte-bound anyway and take the DMA path
constexpr uint64_t PACK_WINDOW_BYTES_IN_PLACE = (uint64_t)512 << 20;  // in-place mode: the window IS the result, whatever its size
__host__ __device__ inline uint64_t pack_align(uint64_t b) { return (b + 255u) & ~(uint64_t)255u; }
hipError_t launch_pack_results(const PackResultsJob& job, hipStream_t stream);
constexpr uint32_t SMALL_UPLOAD_ROWS = 4096;  // at or below this, Transform uploads take the one-kernel path
hipError_t launch_vis_begin(const Columns& c, hipStream_t stream);
hipError_t launch_vis_end(const Columns& c, hipStream_t stream);

// ---- VisibleEntities compaction -----------------------------------------------------------
struct CompactArgs {
    uint32_t n;                  // rows
    uint32_t n_views, n_classes; // segments = n_views * n_classes (segment = view * n_classes + class_slot)
    uint32_t class_bits[32];     // class bit of each class slot
    const uint32_t* order;       // rows in ascending Entity-key order, nullptr = identity
    const uint32_t* class_mask;  // nullptr = every row in class bit 0
    const uint64_t* entity_keys; // nullptr = key == row
    const uint64_t* bitmask;     // per-view masks (local words only)
    uint64_t words_per_view, word_offset;
    uint32_t* block_counts;      // [segments * n_blocks] -> turned into exclusive bases by the scan
    uint32_t* seg_totals;        // [segments]
    uint64_t* seg_bases;         // [segments] start of each segment in out_rows/out_keys
    uint32_t* out_rows;          // packed lists
    uint64_t* out_keys;
    uint32_t n_blocks;
};
constexpr uint32_t COMPACT_BLOCK_ROWS = 4096;
hipError_t launch_compact(const CompactArgs& a, hipStream_t stream, void (*mark)(void*, uint32_t), void* mark_ctx);

// Single-launch compaction for rows already in ascending Entity-key order (the common case: the shim numbers
// rows by key).  Consumes the per-wave counts + segment masks the cull pass left behind; every workgroup
// derives its base from the preceding wave counts (L2-resident bytes), so no scan kernel and no atomics.
// Segment s writes out_rows[s * seg_stride ...] and seg_totals[s].
hipError_t launch_compact_fast(const CompactFastArgs& a, hipStream_t stream);

// ---- hierarchy ---------------------------------------------------------------------------------
constexpr uint32_t TILE_MAX_LEVELS = 16;  // levels a tile may span (TileDesc); the tile kernel has a second instantiation for tiles deeper than ...
constexpr uint32_t TILE_FAST_LEVELS = 8;  // ... this, which is what every tile of a bushy tree fits in (the LDS rows bind first): 8 keeps the
                                          // descriptor's per-level scalars in registers (16: 28 - 61 spilled SGPRs in every instantiation)
// Light tiles (the only tiles since round 4): few LDS rows and a last level of one row per thread, nothing software-pipelined
// -- eight workgroups per CU, a tile is two dependent round trips.  A hierarchy that cannot be cut into them is swept level by
// level (k_propagate_level), see ctx_hierarchy.cpp.
constexpr uint32_t TILE_LIGHT_UCAP = 112;      // LDS slots of one tile: rows of all its levels but the last
constexpr uint32_t TILE_LIGHT_LAST_CAP = 256;  // the planner keeps a tile's streamed last level at or below this
constexpr uint32_t TILE_INHERIT_UCAP = 512;    // k_inherit_tiles: upper rows whose InheritedVisibility state it keeps in LDS (one byte each)
// A level this wide is not given to tiles at all: it is swept by a streaming launch of its own (k_propagate_level) behind
// the level above it.  The deepest level qualifies earlier than the ones above it (nothing else has to wait for it).
// With the light tile kernel at 8 workgroups per CU the tiles win on everything measured -- 1.4 M nodes 36.0 against 39.6 us per
// frame, 5.6 M nodes (4.2 M leaves) 153 - 159 against 162, and the wide shapes at 1.2 M nodes: 1 + 1100 + 1.21 M rows 30.7 against
// 33.8, a root with 1.2 M children 29.9 against 32.5, fan-out 16 29.1 against 33.5 -- so the thresholds sit above that range; the
// _TEST pair (mi_debug_set_tile_mode(3)) keeps the streamed path under test at sizes the oracle handles in seconds.
constexpr uint32_t STREAM_LEVEL_MIN_ROWS_LAST = 1u << 23;
constexpr uint32_t STREAM_LEVEL_MIN_ROWS = 1u << 24;
constexpr uint32_t STREAM_LEVEL_MIN_ROWS_LAST_TEST = 1u << 20;
constexpr uint32_t STREAM_LEVEL_MIN_ROWS_TEST = 1u << 21;
constexpr uint32_t TILE_MAX_CHAIN = 24;       // ancestors a chain tile re-evaluates (levels above its first level)
constexpr uint32_t TILE_ROOTS = 0x80000000u;   // TileDesc::kind: first level = level 0 of the forest
constexpr uint32_t TILE_CHAIN_MASK = 0xFFu;    // TileDesc::kind: chain length (0 = parents come from global memory)
struct TileDesc {
    uint32_t n_levels;
    uint32_t start[TILE_MAX_LEVELS];
    uint32_t count[TILE_MAX_LEVELS];
    uint32_t kind;
};
// The ancestor table: per row its first ANC_DEPTH ancestors, parent first, 0xFFFFFFFF-padded (64 bytes per row).  mark_dirty_trees is
// a climb from every changed row to its root -- one dependent round trip per level when it follows parent_idx (eleven for a leaf of
// the 1 M-node tree); with the table a row's ancestors arrive in ONE load and their marks go out as independent stores.  Built on
// the device, a launch per level, when a hierarchy is uploaded; trees deeper than ANC_DEPTH + 1 levels go on climbing from the last entry.
constexpr uint32_t ANC_DEPTH = 16;
hipError_t launch_build_ancestors(const uint32_t* parent_idx, uint32_t start, uint32_t count, uint32_t* anc, hipStream_t stream);
hipError_t launch_mark_dirty(uint32_t n, const uint8_t* changed, uint32_t changed_gen, const uint32_t* parent_idx, uint8_t* tree_bytes,
                             uint32_t* clear_words /* the other half, zeroed for the next frame; nullptr = none */, uint32_t n_clear_words,
                             hipStream_t stream, const uint32_t* anc = nullptr /* the ancestor table, or nullptr: climb along parent_idx */);
// One launch over a group of mutually independent tiles (TileDesc::kind tells roots / chain / dependent apart).
// The hierarchy FRAME in the tile launch itself (k_propagate_fans<true, true>): every tile also runs the visibility systems over
// its own rows, GlobalTransforms still in registers / LDS.  A tile's rows are not aligned to the 64-row words of the per-view masks,
// so the words are ORed and the wave counts added with atomics into memory the host zeroed before the launch.
struct TreeCull {
    ViewSet views;
    uint32_t n_views;
    VisibilityOut out;      // zeroed: n_views * words_per_view words
    uint8_t* wave_cnt;      // zeroed: [n_views][n_waves] (one class segment per view), or nullptr
    uint32_t n_waves;
    // what the NEXT such frame will OR into: zeroed by this launch (every tile a slice), so that no memset sits between the frames
    // (the frame sets rotate by three: this frame's, the previous frame's -- being compacted --, the next frame's)
    uint64_t* zero[3];
    uint32_t zero_words[3];
    // the previous frame's deferred VisibleEntities compaction rides in the first n_compact workgroups (MI_CULL_MORE_FRAMES)
    CompactFastArgs prev;
    uint32_t prev_gx, n_compact;
};
// A hierarchy whose every level is at most a wave wide: ONE wave walks all of it (kernels_tree.hip, k_propagate_narrow); quad = no
// level holds more than 16 rows (a node per quad of lanes, a column each).
hipError_t launch_propagate_narrow(const Columns& c, const uint32_t* parent_idx, const uint32_t* level_offsets, uint32_t n_levels,
                                   const uint8_t* node_flags, const uint8_t* changed, const uint8_t* tree_bytes, uint8_t* g_changed_bytes, bool all_dirty,
                                   bool static_opt, bool quad, hipStream_t stream);
// A forest of small trees, a wave per tile (kernels_tree.hip, k_propagate_wave_tiles): every tile is a forest-root tile whose levels
// hold <= 64 rows each (quad: <= 16) and <= WAVE_TILE_ROWS rows together.
constexpr uint32_t WAVE_TILE_ROWS = 80;
hipError_t launch_propagate_wave_tiles(const Columns& c, const uint32_t* parent_idx, const TileDesc* d_wtiles, uint32_t n_tiles, const uint8_t* node_flags,
                                       const uint8_t* changed, const uint8_t* tree_bytes, uint8_t* g_changed_bytes, bool all_dirty, bool static_opt,
                                       bool quad, bool pretest, hipStream_t stream);
// STRIPS (round 6, kernels_tree.hip: k_propagate_strips): a lopsided or deep tree in ONE launch of mutually independent workgroups.  The
// levels are cut into bands; a strip OWNS a run of consecutive rows of its band's first level and all their descendants inside the band
// (no level of it wider than the planner's width), and RE-EVALUATES -- never writes -- the cone of their ancestors above the band, level
// by level, a contiguous row range each (rows are in level order and ordered by parent): the same products in the same order as the
// strips that own those rows form, hence the same bits and the same change ticks, and nothing another workgroup produces is waited for.
// A strip is a list of ROUNDS -- up to 64 rows of one level -- grouped into BATCHES: a round of more than 16 rows alone, or up to four
// consecutive narrow levels; a producer wave stages a batch in LDS, four consumer waves walk it (kernels_tree.hip).  A level's
// GlobalTransforms stay in LDS for the level below (two levels of STRIP_W_CAP rows).
constexpr uint32_t STRIP_W_CAP = 128;  // rows of one level of one strip
constexpr uint32_t STRIP_MAX_ROWS = 1u << 20;  // hierarchies above this are bandwidth: the workgroup tiles (measured: profiles/r06_experiments.md)
constexpr uint32_t STRIP_TAB_CAP = 256;  // table entries of a strip held in LDS: a strip has at most STRIP_TAB_CAP - 8 (the producer reads ahead)
constexpr uint32_t STRIP_CONSUMERS = 4;  // consumer waves of a strip's workgroup: sixteen rows of a wide round each; one more wave produces
constexpr uint32_t STRIP_THREADS = 64u * (STRIP_CONSUMERS + 1u);
struct StripRound {  // 16 bytes
    uint32_t row0;    // first row of the round
    uint32_t pstart;  // first row of the strip's range in the level above: a row's parent sits in LDS slot parent - pstart
    uint32_t info;    // bits 0-6 rows (0 = padding), 8-15 first LDS slot of the round inside its level, then the STRIP_* bits
    uint32_t level;
};
constexpr uint32_t STRIP_PARITY = 1u << 16;     // which of the two LDS level buffers the round's level writes
constexpr uint32_t STRIP_OWNED = 1u << 17;      // the strip owns the rows (writes GlobalTransform, change byte, snapshot); otherwise the cone
constexpr uint32_t STRIP_ROOT = 1u << 18;       // level 0 of the forest
constexpr uint32_t STRIP_ABOVE_TOP = 1u << 20;  // cone rounds of the level directly above the strip's first own level
constexpr uint32_t STRIP_BATCH_SHIFT = 22;      // bits 22-24 of a batch's FIRST entry: its entries (1 = one round of up to 64 rows; 2 - 4 = consecutive narrow levels)
struct StripDesc {
    uint32_t first_round;
    uint32_t n_rounds;  // bits 0-15 the strip's table entries, 16-23 its batches (even); bit 31: the strip owns rows of the snapshot prefix (no early exit)
};
hipError_t launch_propagate_strips(const Columns& c, const uint32_t* parent_idx, const StripDesc* d_strips, const StripRound* d_rounds, uint32_t n_strips,
                                   const uint8_t* node_flags, const uint8_t* changed, const uint8_t* tree_bytes, uint8_t* g_changed_bytes, const float* snap_read,
                                   float* snap_write, uint32_t snap_rows, bool all_dirty, bool static_opt, bool pretest, hipStream_t stream,
                                   unsigned long long* trace = nullptr);
hipError_t launch_propagate_tiles(const Columns& c, const uint32_t* parent_idx, const TileDesc* d_tiles, const uint32_t* d_chains,
                                  uint32_t n_tiles, const uint8_t* node_flags, const uint8_t* changed, const uint8_t* tree_bytes,
                                  uint8_t* g_changed_bytes, const float* snap_read, float* snap_write, uint32_t snap_rows, bool all_dirty,
                                  bool static_opt, hipStream_t stream,
                                  unsigned long long* trace = nullptr, bool pretest = false /* static-scene rule: flags first */,
                                  const TreeCull* cull = nullptr /* all-dirty frames: the visibility systems ride in the launch */,
                                  bool deep = false /* some tile spans more than TILE_FAST_LEVELS levels */);
// One whole level [start, start + count) as a stream: every row's parent lies in the level above, complete in global
// memory (an earlier launch).  Same per-node rule as the tiles.  root_node_flags != NULL: the level is level 0 (no parents;
// the roots' rule reads the has-children bit).
hipError_t launch_propagate_level(const Columns& c, const uint32_t* parent_idx, uint32_t start, uint32_t count, const uint8_t* changed,
                                  const uint8_t* tree_bytes, uint8_t* g_changed_bytes, bool all_dirty, bool static_opt, hipStream_t stream,
                                  const uint8_t* root_node_flags = nullptr);
hipError_t launch_inherit_level(const uint32_t* parent_idx, uint32_t start, uint32_t count, const uint8_t* visibility, uint8_t* flags,
                                uint8_t* inh_changed, hipStream_t stream);
// InheritedVisibility propagation (visibility_propagate_system): writes bit0 of flags[] and changed bytes.
hipError_t launch_inherit_flat(uint32_t n, const uint8_t* visibility, uint8_t* flags, uint8_t* inh_changed, hipStream_t stream);
hipError_t launch_inherit_tiles(const uint32_t* parent_idx, const TileDesc* d_tiles, uint32_t n_tiles, bool roots,
                                const uint8_t* visibility, uint8_t* flags, uint8_t* inh_changed, hipStream_t stream);
hipError_t launch_clear_u32(uint32_t* p, uint64_t n_words, hipStream_t stream);
hipError_t launch_bytes_to_bits(const uint8_t* bytes, uint32_t n, uint64_t* bits, hipStream_t stream);

// ---- clustering ------------------------------------------------------------------------------
struct ClusterViewDev {
    uint32_t dims[3];
    uint32_t is_orthographic;
    uint32_t view_layer_mask;
    uint32_t n_clusters;
    float cluster_factors[2];
    float view_from_world[16];
    float clip_from_view[16];
    float view_from_world_scale[3];
    float view_from_world_scale_max;
    float frustum[24];
    const float* x_planes;
    const float* y_planes;
    const float* z_planes;
    const float* cluster_spheres;
    uint32_t view_layer_mask_hi;  // RenderLayers 32..63 of the view
};
struct ClusterObjects {
    uint32_t n;
    const float* pos_range;      // 4n
    const uint8_t* obj_type;     // n or nullptr
    const uint32_t* layer_mask;  // n or nullptr
    const uint32_t* layer_mask_hi;  // n or nullptr: RenderLayers 32..63
    const float* spot_dir;       // 3n or nullptr
    const float* spot_sin_cos;   // 2n or nullptr
    // mi_cluster_bind_objects_to_rows: object i is row first_row + i of the context's columns.  It takes part only if its
    // ViewVisibility::get() is true -- the gather of assign.rs:190-296 done on the device -- its centre is the row's
    // GlobalTransform translation and a spot light's direction the row's GlobalTransform::back().  Two ways to know:
    //   row_vv != nullptr   read the byte the cull of this frame left (the assignment is ordered behind that cull);
    //   derive != 0         re-derive it with the cull's own rule (visibility_rule.h) from the row's Transform, bounds, flags
    //                       and the frame's views, so the assignment can run CONCURRENTLY with the frame kernel of the same
    //                       frame on another stream.  Flat rows only (GlobalTransform == From(Transform)).
    const float* row_global;     // 12 floats per row, or nullptr = objects are not rows
    const uint8_t* row_vv;
    // derive mode, which rows this frame's propagate writes (those are From(Transform); the others keep the resident column):
    // row_changed == nullptr && !derive_resident: every row; row_changed: the rows whose byte is set; derive_resident: none
    const uint8_t* row_changed;
    uint32_t changed_gen;        // generation of row_changed's stamps (row_changed())
    uint32_t derive_resident;
    uint32_t first_row;
    const uint32_t* row_list;    // mi_cluster_bind_objects_to_row_list: object i is row row_list[i] (nullptr: first_row + i)
    uint32_t derive, n_views;
    const float *row_translation, *row_rotation, *row_scale, *row_aabb_center, *row_aabb_half, *row_range;
    const uint8_t* row_flags;
    const uint32_t* row_layers;
    const uint32_t* row_layers_hi;  // RenderLayers 32..63 of the rows, or nullptr
    const uint32_t* row_summary;  // RowSummary of the context's rows when it is current, else nullptr (derive mode)
};
constexpr uint32_t CLUSTER_BLOCK = 256;  // objects per workgroup (= bits per cluster row in LDS)
struct ClusterWork {
    uint32_t n_blocks;
    uint32_t row_stride;          // entries per cluster row of block_counts: n_blocks rounded up to 8
    // Everything the walk kernel accumulates into is double-buffered by frame parity: the fill kernel of frame f
    // zeroes the buffers frame f+1 will use, so no memset sits in the stream.
    uint16_t* block_counts;       // [n_clusters * n_blocks] objects of block b in cluster c (cluster-major; only
                                  //  non-empty entries are written)
    uint16_t* block_counts_next;
    uint32_t* counts;             // accumulator block, 16-byte aligned sections: [6 * n_clusters] ClusterableObjectCounts
    uint32_t* totals;             //   [n_clusters]
    float* farthest_z;            //   (uint bits for atomicMax of non-negative floats)
    uint32_t* pair_total;         //   number of (cluster, block) pairs
    uint32_t* acc_next;           // the other parity's accumulator block
    uint32_t acc_words;           // its size in 32-bit words
    uint32_t* pair_cb;            // [n_blocks * n_clusters] (block << 12) | cluster of every non-empty row
    uint32_t* pair_mask;          // [n_blocks * n_clusters * 8] its 256-bit object mask
    uint32_t* offsets;            // [n_clusters + 1] out
    uint32_t* indices;            // [capacity] out
    uint64_t capacity;
    uint64_t* total;              // [1] out
    int32_t obj_delta;            // object of (block b, bit k) = b * 256 + k + obj_delta: 0 for blocks of 256 objects; blocks that are ROW
                                  // tiles of the frame kernel (ClusterWalkJob::inrow) start obj_delta objects in front of object 0
};
hipError_t launch_cluster_bindings(uint32_t n_clusters, const uint32_t* offsets, const uint32_t* counts, const uint32_t* indices,
                                   const uint32_t* remap, uint32_t n_remap, uint64_t capacity, uint32_t* out_oc,
                                   uint32_t* out_idx, hipStream_t stream);
// frame_views: the cull's views, read only when objs.derive (may be nullptr otherwise)
hipError_t launch_cluster_assign(const ClusterViewDev& view, const ClusterObjects& objs, const ClusterWork& w, const ViewSet* frame_views,
                                 bool small_lds, bool fill, hipStream_t stream, void (*mark)(void*, uint32_t), void* mark_ctx,
                                 WalkPlanesHost walk_planes = WalkPlanesHost{nullptr, 0});
// fill = false leaves the second kernel to the caller: launch_cluster_fill, or the frame kernel's extra workgroups (ClusterFillJob)
hipError_t launch_cluster_fill(const ClusterWork& w, uint32_t n_clusters, uint32_t n_objects, hipStream_t stream);
struct ClusterFillJob {
    ClusterWork w;
    uint32_t n_clusters, n_objects;
};
// The walk of this frame's assignment, riding in the frame kernel's own launch (n_blocks extra workgroups).
struct ClusterWalkJob {
    ClusterViewDev view;
    ClusterObjects objs;   // derive mode: the lights' ViewVisibility is re-derived from the frame's views
    ClusterWork w;
    uint32_t zc;           // z slices per chunk: what fits the frame kernel's LDS
    uint32_t n_blocks;     // 0 = no walk rides in this launch
    // inrow: objects bound to a CONTIGUOUS row range (mi_cluster_bind_objects_to_rows) are walked by the frame kernel's own row
    // workgroups -- the tiles tile0 .. tile0 + n_blocks - 1 hold them, a block = a row tile, and those workgroups go on into the walk
    // with the ViewVisibility and the GlobalTransform translation they have just computed: no extra workgroups, nothing re-derived.
    // The launch hands those tiles out first.
    uint32_t inrow, tile0;
    uint32_t spots;        // there are spot lights among the objects: the walk runs the cone test too (view.cluster_spheres is set)
};
// The view's cluster planes (x | y | z, four floats each) as a trailing kernel argument of the frame kernels that carry the walk: the
// table is new nearly every frame (its near plane follows the camera's scale by an ulp), a device copy would be an H2D blit in front
// of every frame, and read from the pinned staging arena it was a trip over PCIe for each of the light tiles' workgroups -- 1.1 us of
// the metric frame (profiles/r05a/rider_tax_decomposition.txt).  The kernel-argument segment is device memory; the walkers read the
// table from it with per-lane loads (kernarg_late).  Tables of more than WALK_PLANES_MAX floats keep the staged copy.
constexpr uint32_t WALK_PLANES_MAX = 256;
struct WalkPlanes {
    float f[WALK_PLANES_MAX];
};
struct NoWalkPlanes {};
template <int WALK>
struct WalkPlanesArg {
    typedef WalkPlanes type;
};
template <>
struct WalkPlanesArg<0> {
    typedef NoWalkPlanes type;
};
// (the table on the host: what the context hands the launch_* wrappers of the kernels that carry a walk -- an explicit argument since
// round 6; until then a thread_local the launchers consumed, which a launch path that did not consume it could leave dangling)
// bytes of the LDS arena a walking workgroup needs for chunks of zc z slices (layout: cluster_walk.h)
inline size_t cluster_walk_lds_bytes(uint32_t dxy, uint32_t zc, uint32_t n_planes, bool planes_in_lds) {
    const size_t RC = (size_t)dxy * zc;
    return (RC * 8u + 48u) * sizeof(uint32_t) + 256u /* logf table */ + (planes_in_lds ? (size_t)n_planes * 16u : 0u) + ((RC + 31u) / 32u + 5u) * 4u + RC * 2u + 16u;
}
// k_frame's static LDS (words): what its riders get as their arena -- the compaction and the fill 16 KB, the walk 22 KB (seven
// workgroups per CU; until round 5: 31 KB and five)
// (round 6: 22 KB and 7 waves per SIMD -- FRAME_WALK_WAVES, the launch bound of the kernels that carry the walk without spot lights --
// instead of 31 KB and 5: the arena holds 3 - 4 z slices of a 16 x 9 grid per chunk instead of 5 - 6, and the rows of the launch, which
// are 99 % of its workgroups, run at 7 waves per SIMD: metric frame 21.8 -> 19.5 us per step on one box, profiles/r06_experiments.md)
#ifndef MI_WALK_LDS_WORDS
#define MI_WALK_LDS_WORDS 5600u
#endif
#ifndef MI_FRAME_WALK_WAVES
#define MI_FRAME_WALK_WAVES 7
#endif
constexpr int FRAME_WALK_WAVES = MI_FRAME_WALK_WAVES;
constexpr uint32_t FRAME_LDS_WORDS = 4096u, FRAME_WALK_LDS_WORDS = MI_WALK_LDS_WORDS;
constexpr size_t FRAME_KERNEL_LDS_BYTES = (FRAME_WALK_LDS_WORDS + 4) * 4;  // the riding walk's arena
#ifdef MI_EXP_FILL_RIDE_BLOCKS
constexpr uint32_t CLUSTER_FILL_RIDE_BLOCKS = MI_EXP_FILL_RIDE_BLOCKS;  // (A/B builds)
#else
constexpr uint32_t CLUSTER_FILL_RIDE_BLOCKS = 128;  // workgroups a riding fill adds to the frame kernel's grid
#endif

// ---------------------------------------------------------------------------------------------
// Batching work-item build (kernels_batch.hip).
// ---------------------------------------------------------------------------------------------
// instance counters are one per 64-byte line: agent-scope atomics on one line serialise (~25 ns each), and a frame adds to a
// bin from every tile that saw it
constexpr uint32_t BATCH_INST_STRIDE = 16;
constexpr uint32_t BATCH_TILE = 1024;       // list items per workgroup in the partition passes (2048: 8.4 + 15.2 us, 1024: 6.5 + 12.4, 512: 6.8 + 13.0 at 73 k rows)
constexpr uint32_t BATCH_NO_SET = 0xFFFFFFFFu;
struct BatchInitial {
    uint32_t work_item_index[2], indirect_parameters_index[2], batch_set_index[2], output_mesh_uniform_index;
};
struct BatchArgs {
    // the view's VisibleEntities list of one class (device), its length and -- general compaction path -- its base
    const uint32_t* list;
    const uint32_t* list_count;
    const uint64_t* list_base;  // nullptr: `list` already points at the first entry
    // per-row render-world columns, resolved by k_batch_resolve_rows
    const uint32_t* row_bucket;  // where the row goes (see kernels_batch.hip), BATCH_NO_SET = nowhere in this phase
    const uint32_t* row_input;
    const uint32_t* row_meta;    // multidrawable row -> index of its bin's metadata, 0xFFFFFFFF = names no bin
    // the phase's buckets: kind (2 bits) | mesh class (1 bit) | bin or set id << 3
    uint32_t n_buckets, first_set_bucket, n_sets, n_meta, no_indirect;
    const uint32_t* bucket_desc;
    const uint8_t* set_indexed;
    const uint32_t* meta_offset;
    const uint32_t* bin_meta_in;   // as uploaded, 3 words per bin: indirect_parameters_offset, bin_index, (ignored)
    uint32_t* bin_metadata_out;    // the same with instance_count filled in
    uint32_t* inst_count;          // [n_meta * BATCH_INST_STRIDE] zero on entry; the build's instance counts
    uint32_t* inst_count_next;     // [n_meta * BATCH_INST_STRIDE] zeroed by this build for the next one
    // scratch
    uint32_t* rows_a;
    uint32_t* rows_b;
    uint32_t* tile_hist;     // [n_tiles][256]
    uint32_t n_tiles;
    uint32_t* set_count;     // two-pass only, [2][n_buckets]: start and end of the bucket's run in the partitioned list
    uint32_t* plan;          // [7][n_buckets]: start in the partition, first MeshUniform slot, first work item, first indirect
                             //                 parameters, first batch set, first unbatchable index, rows
    uint32_t* counters;      // [0] entries of the list that are in some bucket
    // outputs, [0] non-indexed [1] indexed
    uint32_t* work_items[2];   // 2 words each
    uint32_t* metadata[2];     // 5 words each
    uint32_t* batch_sets[2];   // 2 words each
    uint32_t* unbatchable;     // 2 words per unbatchable entity with an input index: bin, instance index
    uint32_t* records;         // 8 words per non-empty batchable bin / batch set
    uint32_t* totals;          // 9 words (mi_batch_totals)
    BatchInitial initial;
};
struct SortedArgs {
    const uint32_t* items;     // 4 words per phase item: input index, batch-set key, bin key, flags (device memory, or pinned host memory mapped into the device)
    uint32_t n_items, automatic_batching, no_indirect, merge_only;
    uint32_t* work_items[2];
    uint32_t* metadata[2];
    uint32_t* batch_sets[2];
    uint32_t* batches;         // 6 words per batch set (mi_sorted_batch)
    uint32_t* totals;          // 9 words
    BatchInitial initial;
};
hipError_t launch_batch_resolve_rows(uint32_t n, uint32_t n_sets, uint32_t n_unbatchable, uint32_t n_batchable, const uint8_t* row_kind,
                                     const uint32_t* row_cpu_bin, const uint32_t* row_set, const uint32_t* row_bin, const uint32_t* row_input,
                                     const uint32_t* bin_table_offset, const uint32_t* bin_table, const uint32_t* meta_offset,
                                     uint32_t* row_meta, uint32_t* row_bucket, hipStream_t stream);
// enqueues the whole build (3 launches up to 256 buckets); `mark` is called before each kernel for profiling
hipError_t launch_batch_build(const BatchArgs& a, hipStream_t stream, void (*mark)(void*, uint32_t), void* mark_ctx);
// Sorted phases (kernels_sorted.hip).  Up to SORTED_ONE_WG_ITEMS items: one workgroup, one launch, and `items` may be pinned host
// memory (read once, coalesced).  Longer: tiles of SORTED_TILE_ITEMS items in two launches; partials = batch_sorted_partial_words(n_items)
// words of scratch.  one_wg_limit (test hook): phases longer than this take the tiled form whatever their length.
constexpr uint32_t SORTED_ONE_WG_ITEMS = 8192;
constexpr uint32_t SORTED_TILE_ITEMS = 4096;
uint32_t batch_sorted_partial_words(uint32_t n_items);
hipError_t launch_batch_sorted(const SortedArgs& a, hipStream_t stream, void (*mark)(void*, uint32_t), void* mark_ctx, uint32_t* partials = nullptr,
                               uint32_t one_wg_limit = SORTED_ONE_WG_ITEMS);

}  // namespace mi
