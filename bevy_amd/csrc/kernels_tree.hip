// kernels_tree.hip -- hierarchy propagation on gfx950:
//   mark_dirty_trees              crates/bevy_transform/src/systems.rs:111-306
//   propagate_parent_transforms   crates/bevy_transform/src/systems.rs:506-748 (roots :522-530, descendants :679-748)
//   sync_simple_transforms        crates/bevy_transform/src/systems.rs:42-79 (flat rows sharing level 0 with the roots)
//
// Rows are in level (BFS) order, so the descendants of a contiguous range of nodes form one contiguous range per level.
// A workgroup owns a *subtree tile*: a contiguous range of nodes at the tile's first level plus all their descendants for
// the next few levels (the reference's mpsc work queue of 512-entity chunks, systems.rs:767-813, becomes the tile list).
// k_propagate_fans walks a tile in three steps:
//   step 0  every row of the tile's upper levels (all but the last; <= TILE_LIGHT_UCAP rows together) gets its local affine
//           computed from T/R/S into an LDS slot together with its old GlobalTransform and its parent's slot -- ALL upper
//           levels at once, so the HBM latency is paid once per tile, not once per level;
//   step 1  level by level (workgroup barrier in between) G = G_parent * local, both operands read from LDS; the result
//           overwrites the local affine in place;
//   step 2  the tile's last level -- usually ~3/4 of its rows -- is streamed: coalesced T/R/S and old-G loads (wave-local
//           LDS transpose), parent G from LDS, coalesced G stores.
// One launch therefore covers up to TILE_MAX_LEVELS levels with one HBM round trip of latency plus the streaming time.
// A subtree too big for a tile (one node with hundreds of children that have children of their own: its upper rows overflow the
// LDS slots) is cut by the planner: the levels that fit make a tile, the rows below become tiles of a later launch that read their
// first-level parents from global memory (ctx_hierarchy.cpp).  A hierarchy past this kernel's 32-bit byte offsets is swept level
// by level instead, one k_propagate_level launch per level -- the kernel the widest levels of very big trees take anyway.  (Rounds
// 1-3 kept a second, big-tile kernel for oversized subtrees; it had not followed the light kernel's changes and is gone.)
//
// Algorithmic bytes per node: read T 40 + parent_idx 4 + old G 48 (set_if_neq, systems.rs:719),
// write G 48 + changed 1; parent G comes from LDS (first level of a non-root tile: from L2).
#include "glam_math.h"
#include "kernels.h"
#include "visibility_rule.h"
#include "compact_fast.h"

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
__device__ __forceinline__ void st_affine(float* g, uint32_t row, const Affine& a) {
    float4 o0, o1, o2;
    pack(a, o0, o1, o2);
    float4* dst = reinterpret_cast<float4*>(g) + 3ull * row;
    dst[0] = o0;
    dst[1] = o1;
    dst[2] = o2;
}
__device__ __forceinline__ Affine lds_affine(const float4* slots, uint32_t slot) {
    return unpack(slots[slot * 3u], slots[slot * 3u + 1u], slots[slot * 3u + 2u]);
}
__device__ __forceinline__ void lds_put(float4* slots, uint32_t slot, const Affine& a) {
    float4 o0, o1, o2;
    pack(a, o0, o1, o2);
    slots[slot * 3u] = o0;
    slots[slot * 3u + 1u] = o1;
    slots[slot * 3u + 2u] = o2;
}

// Addressing with a uniform base and a 32-bit byte offset per lane (global_load ... v_off, s[base:base+1]): no 64-bit
// address arithmetic in the vector ALU.  Good for rows below 2^32 / 48 (the planner keeps bigger contexts on the other kernel).
template <class T>
__device__ __forceinline__ const T& at32(const void* base, uint32_t byte_off) {
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}
template <class T>
__device__ __forceinline__ T& at32w(void* base, uint32_t byte_off) {
    return *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off);
}
__device__ __forceinline__ V3 ld3_32(const float* base, uint32_t row) {
    const F3 v = at32<F3>(base, row * 12u);
    return V3{v.x, v.y, v.z};
}
__device__ __forceinline__ V4 ld4_32(const float* base, uint32_t row) {
    const float4 v = at32<float4>(base, row * 16u);
    return V4{v.x, v.y, v.z, v.w};
}
// Element-wise selects.  (`cond ? a : b` on whole structs makes the compiler park both in a scratch array and load one back
// through a computed offset -- a trip through memory in the middle of the dependent chain.)
__device__ __forceinline__ V3 sel(bool k, V3 a, V3 b) { return V3{k ? a.x : b.x, k ? a.y : b.y, k ? a.z : b.z}; }
__device__ __forceinline__ Affine sel(bool k, const Affine& a, const Affine& b) {
    Affine r;
    r.m.x_axis = sel(k, a.m.x_axis, b.m.x_axis);
    r.m.y_axis = sel(k, a.m.y_axis, b.m.y_axis);
    r.m.z_axis = sel(k, a.m.z_axis, b.m.z_axis);
    r.t = sel(k, a.t, b.t);
    return r;
}

// mark_dirty_trees: climb from every changed row to its root, setting TransformTreeChanged; a climber stops at the first node
// somebody already marked (systems.rs:208-223: there a shared atomic bitset).  Here the marks are BYTES, set with plain stores
// and tested with plain loads: the reference's test-and-set, as atomics on a bitset, put up to 128 read-modify-writes in a row on
// the same word where the climbs converge (32 nodes of an upper level share a word): 31 us for 10 000 changed leaves of an
// 11-level tree, against 5 us for a changed root.  A plain test can let two climbers that arrive together both go on -- they
// only repeat each other's stores; whoever sees a mark stops, and the one who set it goes on, so every ancestor gets marked.
// A step is ONE round trip (the parent's index and the mark are requested together).  The launch also zeroes the OTHER half of
// the double-buffered marks for the next frame (clear_words), which saves a launch per frame.
// anc[row] = (parent, anc[parent][0 .. ANC_DEPTH - 2]) for the rows of one level; the level above is complete (an earlier launch)
__global__ void __launch_bounds__(256) k_build_ancestors(const uint32_t* __restrict__ parent_idx, uint32_t start, uint32_t count, uint32_t* anc) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    const uint32_t row = start + i, p = parent_idx[row];
    uint4 d0 = make_uint4(p, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu), d1 = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu), d2 = d1, d3 = d1;
    if (p != 0xFFFFFFFFu) {
        const uint4* src = reinterpret_cast<const uint4*>(anc + (size_t)p * ANC_DEPTH);
        const uint4 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3];
        d0 = make_uint4(p, q0.x, q0.y, q0.z);
        d1 = make_uint4(q0.w, q1.x, q1.y, q1.z);
        d2 = make_uint4(q1.w, q2.x, q2.y, q2.z);
        d3 = make_uint4(q2.w, q3.x, q3.y, q3.z);
    }
    uint4* dst = reinterpret_cast<uint4*>(anc + (size_t)row * ANC_DEPTH);
    dst[0] = d0;
    dst[1] = d1;
    dst[2] = d2;
    dst[3] = d3;
}
hipError_t launch_build_ancestors(const uint32_t* parent_idx, uint32_t start, uint32_t count, uint32_t* anc, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    MI_LAUNCH(k_build_ancestors, dim3((count + 255u) / 256u), dim3(256), 0, stream, parent_idx, start, count, anc);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) k_mark_dirty(uint32_t n, const uint8_t* __restrict__ changed, uint32_t changed_gen,
                                                     const uint32_t* __restrict__ parent_idx, uint8_t* tree_bytes,
                                                     uint32_t* __restrict__ clear_words, uint32_t n_clear_words, const uint32_t* __restrict__ anc) {
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    if (clear_words)
        for (uint32_t w = gid; w < n_clear_words; w += gridDim.x * 256u) clear_words[w] = 0u;
    const uint32_t row = gid;
    if (row >= n || !row_changed(changed[row], changed_gen)) return;
    mark_row_and_ancestors(row, parent_idx, tree_bytes, parent_idx ? anc : nullptr, n);
}

// A tile's descriptor, fetched whole with scalar loads at the top of the kernel (two s_load for the 72 bytes).  Left to
// itself the compiler reads the fields where they are used: one dependent s_load after the other -- or, inside divergent
// code, per-lane vector loads -- each a round trip of its own in front of the tile's first row load.
template <uint32_t MAXL = TILE_MAX_LEVELS>
__device__ __forceinline__ TileDesc load_tile_desc(const TileDesc* p) {
    typedef const uint32_t __attribute__((address_space(4))) * const_u32;
    const_u32 src = (const_u32)(uintptr_t)p;
    static_assert(sizeof(TileDesc) == (2u + 2u * TILE_MAX_LEVELS) * 4u, "n_levels, start[], count[], kind");
    TileDesc d;  // (entries past MAXL stay unread: a kernel instantiated for MAXL levels is only given tiles of at most that many)
    d.n_levels = src[0];
#pragma unroll
    for (uint32_t k = 0; k < MAXL; ++k) {
        d.start[k] = src[1u + k];
        d.count[k] = src[1u + TILE_MAX_LEVELS + k];
    }
    d.kind = src[1u + 2u * TILE_MAX_LEVELS];
    return d;
}

// A hierarchy whose frame (141 B per node) does not fit the 256 MiB Infinity Cache by a wide margin fetches the streamed level's
// Transforms and old GlobalTransforms past the caches: 5.6 M nodes 155.8 -> 135.7 us per launch; the 1 M-node tree, whose columns the
// next frame still finds cached, 24.6 -> 28.1 the other way (profiles/r05a/nt_loads_ab.txt).
constexpr uint32_t TREE_NT_MIN_ROWS = 4u << 20;
struct TreeArgs {
    const uint32_t* parent_idx;
    const TileDesc* tiles;
    const uint8_t* node_flags;  // bit0 = has children (only level-0 rows consult it); bit1 = some strip's cone holds the row (its owner mirrors it into the snapshot)
    const uint8_t* changed;     // per-row Changed<Transform>|Added<GlobalTransform> byte, nullptr = all
    const uint8_t* tree_bytes;  // TransformTreeChanged, a byte per row, nullptr = all changed
    uint8_t* g_changed_bytes;   // out: GlobalTransform change tick bumped
    const uint32_t* chains;     // [n_tiles * TILE_MAX_CHAIN] ancestor rows of chain tiles (tile root first, forest root last)
    // Chain tiles need the PRE-frame GlobalTransforms of their ancestors while the tiles that own those ancestors
    // rewrite them in the same launch.  The owners therefore keep a snapshot of their rows (a prefix of the row
    // space), double-buffered by frame: this frame's chain tiles read snap_read (written by the previous frame's
    // launch), this frame's owners write snap_write for the next one.  No tile ever waits for another.
    const float* snap_read;
    float* snap_write;
    uint32_t snap_rows;  // the snapshot covers rows [0, snap_rows): whoever computes one of them also writes its snapshot
    uint32_t all_dirty;
    uint32_t static_opt;
    uint32_t changed_gen;  // generation of the change column's stamps (row_changed(), kernels.h)
    uint32_t pretest;  // bit 0: light tiles under the static-scene rule test the tile's flags before asking for anything else (few rows changed);
                       // bit 1: the streamed level's inputs are fetched with nontemporal loads (a hierarchy of >= TREE_NT_MIN_ROWS rows)
    unsigned long long* trace;  // debug: 8 x s_memrealtime per tile (mi_debug_tree_trace), nullptr = off
};

// The per-node rule.  Level-0 rows: roots (systems.rs:522-530) and flat rows (systems.rs:58-63) are plain
// assignments; every other node is set_if_neq(parent * local) unless the static-scene rule skips it
// (systems.rs:708-719).  node_inputs gathers what the rule reads from the per-row side tables, node_apply is the
// pure rule: *cur = the row's GlobalTransform after the system; returns "tick bumped".
struct NodeIn {
    bool tree_changed;  // TransformTreeChanged.is_changed()
    bool root_write;    // level-0 rows only: the assignment happens
};
__device__ __forceinline__ NodeIn node_inputs(const TreeArgs& a, uint32_t row, bool is_root_level) {
    NodeIn in;
    in.tree_changed = a.all_dirty || !a.tree_bytes || a.tree_bytes[row] != 0;
    in.root_write = false;
    if (is_root_level) {
        const bool has_children = a.node_flags && (a.node_flags[row] & 1u);
        in.root_write = has_children ? (!a.static_opt || in.tree_changed) : (a.all_dirty || !a.changed || row_changed(a.changed[row], a.changed_gen));
    }
    return in;
}
__device__ __forceinline__ bool node_apply(bool is_root_level, bool static_opt, NodeIn in, const Affine& gp, bool p_changed,
                                           const Affine& local, const Affine& old, Affine* cur) {
    if (is_root_level) {
        *cur = sel(in.root_write, local, old);
        return in.root_write;
    }
    const bool skip = static_opt && !in.tree_changed && !p_changed;
    const Affine nw = mul(gp, local);               // p_global_transform.mul_transform(*transform)
    const bool set = !skip && !affine_eq(nw, old);  // set_if_neq
    *cur = sel(set, nw, old);
    return set;
}
// node_apply with the old value fetched (from LDS) only after the product is formed: 36 live registers at the peak instead
// of 60 -- the light tile kernel's occupancy hangs on it.  Same rule, same results.
template <class LoadOld>
__device__ __forceinline__ bool node_apply_lazy(bool is_root_level, bool static_opt, NodeIn in, const Affine& gp, bool p_changed,
                                                const Affine& local, LoadOld load_old, Affine* cur) {
    if (is_root_level) {
        if (in.root_write) *cur = local;
        else *cur = load_old();
        return in.root_write;
    }
    const bool skip = static_opt && !in.tree_changed && !p_changed;
    if (!skip) {
        const Affine nw = mul(gp, local);
        __builtin_amdgcn_sched_barrier(0);
        const Affine old = load_old();
        const bool neq = !affine_eq(nw, old);
        *cur = sel(neq, nw, old);
        return neq;
    }
    *cur = load_old();
    return false;
}
__device__ __forceinline__ bool node_update(const TreeArgs& a, bool is_root_level, uint32_t row, const Affine& gp,
                                            bool p_changed, const Affine& local, const Affine& old, Affine* cur) {
    return node_apply(is_root_level, a.static_opt != 0, node_inputs(a, row, is_root_level), gp, p_changed, local, old, cur);
}

// node_inputs in two halves for the light tiles: the side-table bytes are fetched with the row's other loads (one batch,
// no dependent round trip inside the level steps), the rule is evaluated where it is needed.
struct NodeRaw {
    uint32_t tree_word;  // the row's TransformTreeChanged byte (nonzero when everything counts as changed)
    uint8_t nflag;       // node_flags[row] (bit0 = has children)
    uint8_t changed;     // Changed<Transform> | Added<GlobalTransform>
};
// Branch-free on purpose: a load inside an `if` makes the compiler wait for it (and, the counter being in order, for every
// load issued before it) at the end of the branch.  Absent tables are read through a stand-in pointer (parent_idx: 4 bytes per
// row, so any row offset is in bounds) and the value is replaced afterwards.
template <bool ALL_DIRTY>
__device__ __forceinline__ NodeRaw node_raw(const TreeArgs& a, uint32_t row, bool want_nflag) {
    NodeRaw r;
    const uint8_t* const standin = reinterpret_cast<const uint8_t*>(a.parent_idx);
    if (ALL_DIRTY) {
        r.tree_word = 0xFFFFFFFFu;
        r.changed = 1;
    } else {
        const uint32_t w = at32<uint8_t>(a.tree_bytes ? a.tree_bytes : reinterpret_cast<const uint8_t*>(a.parent_idx), row);
        r.tree_word = a.tree_bytes ? w : 0xFFFFFFFFu;
        const uint8_t ch = at32<uint8_t>(a.changed ? a.changed : standin, row);
        r.changed = a.changed ? (uint8_t)(row_changed(ch, a.changed_gen) ? 1 : 0) : (uint8_t)1;
    }
    const uint8_t nf = at32<uint8_t>(a.node_flags && want_nflag ? a.node_flags : standin, row);
    r.nflag = a.node_flags && want_nflag ? nf : (uint8_t)0;
    return r;
}
__device__ __forceinline__ NodeIn node_inputs_raw(const TreeArgs& a, uint32_t row, bool is_root_level, NodeRaw r) {
    NodeIn in;
    in.tree_changed = r.tree_word != 0;
    in.root_write = false;
    if (is_root_level) in.root_write = (r.nflag & 1u) ? (!a.static_opt || in.tree_changed) : r.changed != 0;
    return in;
}

// One streamed row's inputs, fetched one loop iteration ahead of their use (software pipelining: the loads of
// iteration i+1 are in flight while iteration i is multiplied and stored).
struct RowFetch {
    V3 t, s;
    V4 q;
    uint32_t p;
    float4 g0, g1, g2;  // this lane's share of the wave's coalesced old-G rows
};

__device__ __forceinline__ RowFetch fetch_row(const Columns& c, const uint32_t* __restrict__ parent_idx, uint32_t start,
                                              uint32_t count, uint32_t base, uint32_t tid, uint32_t lane, uint32_t wv,
                                              bool root_level) {
    RowFetch f;
    const uint32_t i = base + tid;
    const uint32_t wbase = base + wv * 64u;
    const uint32_t lim = wbase < count ? (count - wbase < 64u ? count - wbase : 64u) * 3u : 0u;
    // lanes (and whole waves) past the end re-read a float4 inside the level: a clamped index, never a `cond ? load : zero`
    // (that turns into a select between a global and a stack address -- flat loads and a scratch slot)
    const float4* src = reinterpret_cast<const float4*>(c.global) + 3ull * (start + (lim ? wbase : 0u));
    const uint32_t last4 = lim ? lim - 1u : 0u;
    f.g0 = src[lane < last4 ? lane : last4];
    f.g1 = src[64u + lane < last4 ? 64u + lane : last4];
    f.g2 = src[128u + lane < last4 ? 128u + lane : last4];
    f.t = V3{0.f, 0.f, 0.f};
    f.s = V3{0.f, 0.f, 0.f};
    f.q = V4{0.f, 0.f, 0.f, 0.f};
    f.p = 0;
    if (i < count) {
        const uint32_t row = start + i;
        f.t = ld3(c.translation, row);
        f.q = ld4(c.rotation, row);
        f.s = ld3(c.scale, row);
        if (!root_level) f.p = parent_idx[row];
    }
    return f;
}

// Tile kinds (TileDesc::kind):
//   TILE_ROOTS   the tile's first level is level 0 of the forest (no parents);
//   chain tile   (kind & TILE_CHAIN_MASK) = n > 0: the tile hangs below ONE node whose n-node ancestor chain
//                (a.chains) it re-evaluates itself -- same operations, same order, hence the same bits as the tile
//                that owns those ancestors writes -- so it depends on nothing another workgroup produces and can
//                share a launch with the tiles above it (a deep narrow tree becomes ONE launch);
//   otherwise    the parents of its first level are read from global memory (written by an earlier launch).
// (Letting a chain tile's workgroup also process an owner tile was measured slower: 42 us against 34.5 us for the
// 1 M-node tree, the owner's latency chain simply adds to that workgroup's time.)
// 256 threads per tile.  Measured alternatives on the 1 M-node depth-11 tree (profiles/r02_tree_experiments.md): one wave
// per tile (64 threads, 85- to 341-row tiles, 16 tiles resident per CU) 35.0 us -- every tile pays its own chain (46 % of
// the VALU instructions of an 85-row tile run on one lane) and the kernel turns issue-bound; 128 threads 36.6 us; 512 / 1024
// threads 39.0 / 50.8 us (fewer tiles resident); 256 threads 33.5 us.
// blockIdx -> tile for the tile kernels.  Workgroups go round the eight XCDs (blockIdx % 8), each with an L2 of its own: XCD x
// takes a CONTIGUOUS eighth of the tiles, so that neighbouring tiles -- whose short upper-level segments share cache lines
// (one row of a level is 12 - 48 bytes of a 128-byte line) and whose chains share ancestors -- meet in one L2 instead of
// fetching the same lines eight times (1 M-node tree: 31.0 -> 30.0 us per launch).  Tiles of a launch do not depend on each
// other, so any bijection is correct.  (The same inside chunks of 256 or 1 536 tiles, which keeps the launch's progression
// through the row space: no different at 1 M, 1.4 M and 5.6 M nodes.)
__device__ __forceinline__ uint32_t xcd_contiguous_tile(uint32_t bid, uint32_t nt) {
    const uint32_t x = bid & 7u, q = nt >> 3, r = nt & 7u;
    return x * q + (x < r ? x : r) + (bid >> 3);
}
__device__ __forceinline__ uint32_t xcd_contiguous_tile() { return xcd_contiguous_tile(blockIdx.x, gridDim.x); }

// ---------------------------------------------------------------------------------------------
// Light tiles (the planner guarantees: every level but the last holds <= TILE_LIGHT_UCAP rows together, chain <=
// TILE_MAX_CHAIN, last level normally <= 256 rows).  Organised for residency instead of reach: 8 workgroups per CU (20.4 KB
// of LDS, 64 registers in the all-dirty instantiation), a
// few rounds of short tiles per CU, so that one tile's head runs under its neighbours' loads.  Nothing is software-pipelined and
// nothing is kept in registers across the serial part that LDS can hold or that can be fetched later.  A tile's loads go out in
// three bursts of straight-line code: behind the descriptor the upper rows (lanes 0..127) and the chain's nodes (top lanes of
// wave 3, whose row numbers do not wait for the descriptor); after the chain the last level's own inputs, one row per lane, which
// travel under the level steps; at the write-back the last level's old GlobalTransforms, into the LDS the upper rows' old values
// occupied.  (Everything in one batch at the top: 30.0 us per launch at 1 M nodes; three bursts: 24.9.)
// ---------------------------------------------------------------------------------------------
// One column of an affine (x_axis, y_axis, z_axis or translation) in an LDS slot of 12 floats.
__device__ __forceinline__ V3 lds_col(const float4* slots, uint32_t slot, uint32_t c) {
    const float* p = reinterpret_cast<const float*>(slots) + slot * 12u + c * 3u;
    return V3{p[0], p[1], p[2]};
}
__device__ __forceinline__ void lds_put_col(float4* slots, uint32_t slot, uint32_t c, V3 v) {
    float* p = reinterpret_cast<float*>(slots) + slot * 12u + c * 3u;
    p[0] = v.x;
    p[1] = v.y;
    p[2] = v.z;
}
// lane k of every quad -> all four lanes of the quad (v_mov_b32 with a DPP quad_perm: a register move, no LDS)
template <int K>
__device__ __forceinline__ float quad_bcast_f(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), K * 0x55, 0xF, 0xF, true));
}
__device__ __forceinline__ V3 quad_bcast(V3 v, int k) {
    switch (k) {
        case 0: return V3{quad_bcast_f<0>(v.x), quad_bcast_f<0>(v.y), quad_bcast_f<0>(v.z)};
        case 1: return V3{quad_bcast_f<1>(v.x), quad_bcast_f<1>(v.y), quad_bcast_f<1>(v.z)};
        case 2: return V3{quad_bcast_f<2>(v.x), quad_bcast_f<2>(v.y), quad_bcast_f<2>(v.z)};
        default: return V3{quad_bcast_f<3>(v.x), quad_bcast_f<3>(v.y), quad_bcast_f<3>(v.z)};
    }
}
// node_apply spread over the FOUR lanes of a quad, one output column each (c = lane & 3: the three axes and the
// translation): the serial part of a tile -- its ancestor chain and its upper levels, one dependent node after the other --
// is bound by instruction latency, not throughput, and a column is a quarter of the dependent instructions of a node.
// Same operations in the same order per element (mul(M3, V3), then + translation), so the same bits.  Every lane of the
// wave must call it (the row-wide "differs from the old value" is a ballot); `on` marks the lanes that carry a node.
// Returns the tick decision of the lane's node; *cur_c = the lane's column of the node's GlobalTransform after the system.
__device__ __forceinline__ bool quad_node_apply(bool on, bool is_root_level, bool static_opt, uint32_t in_bits, const Affine& gp,
                                                bool p_changed, V3 local_c, V3 old_c, uint32_t c, uint32_t lane, V3* cur_c) {
    const bool tree_changed = (in_bits & 1u) != 0, root_write = (in_bits & 2u) != 0;
    const bool skip = static_opt && !tree_changed && !p_changed;
    V3 nw = mul(gp.m, local_c);
    if (c == 3u) nw = nw + gp.t;
    const bool differs = !(nw.x == old_c.x && nw.y == old_c.y && nw.z == old_c.z);
    const unsigned long long m = __ballot(on && differs);
    const bool neq = ((m >> (lane & ~3u)) & 0xFull) != 0;
    if (is_root_level) {
        *cur_c = sel(root_write, local_c, old_c);
        return root_write;
    }
    const bool set = !skip && neq;  // set_if_neq
    *cur_c = sel(set, nw, old_c);
    return set;
}

// ---------------------------------------------------------------------------------------------
// The visibility systems over a tile's own rows (k_propagate_fans<true, true>, the hierarchy frame in one launch):
// reset_view_visibility + check_visibility_cpu_culling + check_visibility_gpu_culling + mark_newly_hidden_entities_invisible
// (visibility/mod.rs:733-737, 788-858, 884-918) for the rows a wave holds, GlobalTransform in hand.  Same rule as the frame
// kernels (visibility_rule.h), same ViewVisibility byte and change tick (view_visibility_tail, kernels_flat.hip).  What differs
// is where the packed results go: a tile's rows are not aligned to the 64-row words of the masks, so a wave cuts its lanes into
// runs of consecutive rows inside one word, and the first lane of each run ORs the run's bits into the word and adds its
// population to the word's count -- atomics into memory the host zeroed before the launch (TreeCull, kernels.h).
// Every lane of the wave must call it; lanes carry rows in ascending order.
// (The rows' ViewVisibility bytes and summaries are fetched here, at the end of the tile.  Fetching them with the tile's first
// bursts and keeping them -- reduced to scalars where the wave's rows agree -- cost 12 registers and a workgroup per CU and measured
// SLOWER, 35.8 against 33.0 us per launch: what the rule adds to a tile is mostly its arithmetic, ~1 900 vector instructions per
// tile, which a kernel organised around latency has nowhere to hide.  profiles/r03_experiments.md.)
// ---------------------------------------------------------------------------------------------
struct NoCull {};
template <bool CULL>
struct CullArg {
    typedef NoCull type;
};
template <>
struct CullArg<true> {
    typedef TreeCull type;
};
// (The row's ViewVisibility byte and summary are fetched here, a round trip of their own per call: requesting them with the tile's
// last burst of loads -- 78 registers instead of 69 -- measured slower, 33.6 against 33.0 us per launch; profiles/r04_experiments.md.)
__device__ __forceinline__ void tile_cull_rows(const Columns& c, const TreeCull& cu, uint32_t lane, bool live, uint32_t row, const Affine& g) {
    const uint32_t rrow = live ? row : 0u;  // (every load below is unconditional: one batch)
    const uint32_t vv0 = c.view_visibility[rrow];
    // Aabb / flags / RenderLayers: the 64-row summary where it says the rows agree (per lane: a wave's rows span two summaries or,
    // for the upper rows, several), else the columns
    const uint4* sp = reinterpret_cast<const uint4*>(c.row_summary) + 2ull * (c.row_summary_on ? rrow >> 6 : 0u);
    const uint4 sa = sp[0], sb = sp[1];
    const uint32_t bits = c.row_summary_on ? sb.w : 0u;
    uint32_t fl = bits & 0xFFu, emask = sb.z, emask_hi = 0u;
    V3 center = V3{__uint_as_float(sa.x), __uint_as_float(sa.y), __uint_as_float(sa.z)};
    V3 half = V3{__uint_as_float(sa.w), __uint_as_float(sb.x), __uint_as_float(sb.y)};
    if (!(bits & ROWSUM_UNIFORM_FLAGS)) {
        fl = c.flags[rrow];
        emask = c.layer_mask[rrow];
        if (c.layer_mask_hi) emask_hi = c.layer_mask_hi[rrow];
    }
    if (!(bits & ROWSUM_UNIFORM_AABB)) {
        center = ld3(c.aabb_center, rrow);
        half = ld3(c.aabb_half, rrow);
    }
    if (!live) fl = 0u;
    const bool ncc = (fl & 0x10u) != 0;  // NoCpuCulling rows are not in the cull query (mod.rs:771)
    uint32_t pass = 0u;
    for (uint32_t v = 0; v < cu.n_views; ++v)
        if (live && !ncc && row_visible_in_view(g, center, half, fl, emask, emask_hi, false, 0.0f, 0.0f, cu.views.v[v])) pass |= 1u << v;
    // the ViewVisibility byte: reset (mod.rs:270-274), set_visible (:290-306), gpu-culling rows (:884-903), mark_newly_hidden (:908-918)
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
        if (cur != vv0) c.view_visibility[row] = (uint8_t)cur;
    }
    // runs of consecutive rows inside one mask word
    const uint32_t word = row >> 6, bit = row & 63u;
    const uint32_t prev_row = __shfl_up(row, 1, 64);
    const bool prev_live = __shfl_up(live ? 1 : 0, 1, 64) != 0;
    const bool leader = live && (lane == 0u || !prev_live || row != prev_row + 1u || bit == 0u);
    const unsigned long long leaders = __ballot(leader);
    const unsigned long long after = lane == 63u ? 0ull : leaders & ~((2ull << lane) - 1ull);
    const uint32_t end = after ? (uint32_t)__ffsll((long long)after) - 1u : 64u;
    const uint32_t len = end - lane;
    const unsigned long long run = len >= 64u ? ~0ull : (1ull << len) - 1ull;
    for (uint32_t v = 0; v < cu.n_views; ++v) {
        const unsigned long long m = __ballot((pass >> v) & 1u);
        if (leader) {
            const unsigned long long sbits = (m >> lane) & run;
            if (sbits) {
                atomicOr(reinterpret_cast<unsigned long long*>(cu.out.bitmask + (size_t)v * cu.out.words_per_view + cu.out.word_offset + word), sbits << bit);
                if (cu.wave_cnt)  // a byte per word, four to an atomic (a word holds <= 64 rows: no carry between the bytes)
                    atomicAdd(reinterpret_cast<uint32_t*>(cu.wave_cnt + (size_t)v * cu.n_waves) + (word >> 2), (uint32_t)__popcll(sbits) << (8u * (word & 3u)));
            }
        }
    }
    const unsigned long long cm = __ballot(vv_changed);
    if (leader) {
        const unsigned long long sbits = (cm >> lane) & run;
        if (sbits) atomicOr(reinterpret_cast<unsigned long long*>(c.vv_changed_bits + word), sbits << bit);
    }
}

// where k_propagate_fans<*, true>'s third argument lies in the kernarg segment (Columns, TreeArgs, TreeCull: each at its natural alignment)
static_assert(alignof(Columns) <= 8 && alignof(TreeArgs) <= 8 && alignof(TreeCull) <= 8, "kernarg offsets below assume 8-byte alignment at most");
constexpr uint32_t FANS_CULL_KERNARG = kernarg_up(kernarg_up(sizeof(Columns)) + sizeof(TreeArgs));
constexpr uint32_t FAN_SLOTS = TILE_LIGHT_UCAP + TILE_MAX_CHAIN;  // LDS slots: upper rows, then the chain's nodes
constexpr uint32_t FAN_CHAIN_LANE0 = 256u - TILE_MAX_CHAIN;       // chain node k is fetched by thread FAN_CHAIN_LANE0 + k

template <bool ALL_DIRTY, bool CULL = false, uint32_t MAXL = TILE_FAST_LEVELS>
// (MAXL: the levels a tile of this launch may span -- TILE_FAST_LEVELS for bushy trees, TILE_MAX_LEVELS for the launches of narrow,
// deep hierarchies: rigs, chains, lopsided trees -- whose tiles would otherwise be cut into several dependent or chain tiles)
// (workgroups per CU: the fused instantiation is bound by vector-instruction issue -- 12 - 13 M wave instructions per launch over 1 024
// SIMDs, 70 - 75 % of its span -- and by how many tiles are resident; 7 against 6 per CU: 31.5 against 33.1 us per launch; 8 -- 64
// registers, one spill -- 31.9)
__global__ void __launch_bounds__(256, CULL ? 7 : ALL_DIRTY ? 8 : 7) k_propagate_fans(Columns c, TreeArgs a, typename CullArg<CULL>::type cu) {
    __shared__ float4 lds_g[FAN_SLOTS * 3];    // local affine, then (upper rows) the GlobalTransform in place
    // GlobalTransforms before this frame of the upper rows and the chain (dead once the level steps have fetched their columns of
    // them); afterwards the same memory is the four waves' transpose buffers of the last level (3 x 1 KB rows each)
    static_assert(FAN_SLOTS * 3 <= 4 * 192, "the old values of the upper rows fit under the transpose buffers");
    __shared__ float4 lds_old_stage[4 * 192];
    float4* const lds_old = lds_old_stage;
    __shared__ uint8_t lds_chg[TILE_LIGHT_UCAP];
    __shared__ uint8_t lds_in[FAN_SLOTS];            // per slot: bit0 TransformTreeChanged, bit1 the level-0 assignment happens
    __shared__ uint32_t lds_pslot[TILE_LIGHT_UCAP];  // per upper row: its parent's LDS slot (or global row, see my_pslot)
    __shared__ uint32_t lds_row[TILE_LIGHT_UCAP];    // per upper row: its row number (the write-back runs over slots)
    __shared__ uint8_t lds_level[TILE_LIGHT_UCAP];   // per upper row: its level inside the tile
    __shared__ float4 lds_chain_g[3];
    __shared__ uint32_t lds_chain_chg;
    uint32_t tile_bid = blockIdx.x, tile_grid = gridDim.x;
    if constexpr (CULL) {
        // the previous frame's VisibleEntities compaction rides in the first workgroups of the launch ...
        if (blockIdx.x < cu.n_compact) {
            const TreeCull& cr = kernarg_late<TreeCull>(FANS_CULL_KERNARG);  // (read inside the branch: see tile_cull_rows' call)
            compact_fast_block<false>(cr.prev, blockIdx.x % cr.prev_gx, blockIdx.x / cr.prev_gx, cr.prev_gx);
            return;
        }
        tile_bid -= cu.n_compact;
        tile_grid -= cu.n_compact;
        // ... and every tile zeroes a slice of what the next such frame ORs its results into
#pragma unroll
        for (uint32_t k = 0; k < 3u; ++k)
            for (uint32_t i = tile_bid * 256u + threadIdx.x; i < cu.zero_words[k]; i += tile_grid * 256u) cu.zero[k][i] = 0ull;
    }
    const uint32_t tile = xcd_contiguous_tile(tile_bid, tile_grid);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    // development trace: stamps are kept in registers and written once at the very end (a store in front of a barrier would
    // make the barrier's vmcnt(0) wait for it and distort the phase it is meant to time)
    unsigned long long ts[7] = {0, 0, 0, 0, 0, 0, 0};
#define FAN_STAMP(i)                              \
    do {                                          \
        if (a.trace) ts[i] = wall_clock64();      \
    } while (0)
    FAN_STAMP(0);
    const bool chain_lane = tid >= FAN_CHAIN_LANE0;
    // the chain's row numbers sit at a fixed place per tile: fetched next to the descriptor, not behind it
    const uint32_t chain_row = chain_lane ? a.chains[(size_t)tile * TILE_MAX_CHAIN + (tid - FAN_CHAIN_LANE0)] : 0u;
    const TileDesc td = load_tile_desc<MAXL>(a.tiles + tile);
    const uint32_t L = td.n_levels;
    const bool ROOTS = (td.kind & TILE_ROOTS) != 0;
    const uint32_t chain_len = td.kind & TILE_CHAIN_MASK;
    float* const snap_out = a.snap_write;
    const uint32_t n_lds = L ? L - 1u : 0u;  // every level but the last is LDS-resident
    uint32_t ubase[MAXL + 1];
    ubase[0] = 0;
#pragma unroll
    for (uint32_t l = 0; l < MAXL; ++l) ubase[l + 1] = ubase[l] + (l < n_lds ? td.count[l] : 0u);
    const uint32_t U = ubase[MAXL];
    uint32_t s_start = td.start[0], s_count = L ? td.count[0] : 0u, s_pbase = 0, s_pstart = 0;  // the last level
#pragma unroll
    for (uint32_t j = 1; j < MAXL; ++j)
        if (j == n_lds) {
            s_start = td.start[j];
            s_count = td.count[j];
            s_pbase = ubase[j - 1];
            s_pstart = td.start[j - 1];
        }
    const bool s_root_level = ROOTS && n_lds == 0;

    // ---- A frame in which little moved (the frame a game mostly runs): most tiles are clean, and a clean tile should cost one small
    // round trip, not its first burst (~11 KB) and its chain.  Flags only: the tile is skipped as a whole (the rule below) iff its
    // parent's GlobalTransform does not change -- which takes a chain node whose own Transform changed: a node that is merely
    // re-evaluated reproduces its value, set_if_neq leaves it alone (systems.rs:719) and its children see an unchanged parent -- and
    // none of its top rows is marked (TransformTreeChanged) or assigned as a root / flat row.  The change bytes of the chain's nodes
    // and the marks of the top rows answer that: <= 24 + 112 bytes.  A tile that fails the test goes on as before (the test is a
    // superset of the exact rule evaluated after the chain), one round trip later: the host asks for it only when few rows changed.
    if constexpr (!ALL_DIRTY) {
        if ((a.pretest & 1u) && a.static_opt && (chain_len || ROOTS) && n_lds && (!snap_out || td.start[0] >= a.snap_rows)) {
            bool hot = false;
            if (chain_lane && tid - FAN_CHAIN_LANE0 < chain_len) hot = row_changed(at32<uint8_t>(a.changed, chain_row), a.changed_gen);
            if (tid < td.count[0]) {
                const uint32_t row = td.start[0] + tid;
                hot = hot || at32<uint8_t>(a.tree_bytes, row) != 0 || (ROOTS && row_changed(at32<uint8_t>(a.changed, row), a.changed_gen));
            }
            if (__syncthreads_or(hot ? 1 : 0) == 0) {
#pragma unroll
                for (uint32_t l = 0; l < MAXL; ++l)
                    if (l < L)
                        for (uint32_t i = tid; i < td.count[l]; i += 256u) at32w<uint8_t>(a.g_changed_bytes, td.start[l] + i) = 0;
                return;
            }
        }
    }

    // ---- burst 1: straight-line code, no load behind a divergent branch (the compiler waits where a result is first used, and
    // the load counter is in order: a use in the middle would split the batch in two) ----
    // (a) this lane's row of the last level (lanes past the end re-read the level's first row)
    const uint32_t wbase0 = wv * 64u < s_count ? wv * 64u : 0u;
    const uint32_t lim0 = (s_count - wbase0 < 64u ? s_count - wbase0 : 64u) * 3u;  // float4s of this wave's rows
    const uint32_t last0 = lim0 ? lim0 - 1u : 0u;  // (an empty last level -- the planner makes none -- must not index at -1)
    const uint32_t s_row = s_start + (tid < s_count ? tid : 0u);
    // (b) an upper row (threads below U) or a chain node (top threads of wave 3)
    const bool is_chain = chain_lane && tid - FAN_CHAIN_LANE0 < chain_len;
    const bool is_upper = tid < U;
    uint32_t my_level = 0xFFFFFFFFu, u_pbase = 0, u_pstart = 0;
    uint32_t u_row = is_chain ? chain_row : s_start;
    {
        uint32_t l = 0;
#pragma unroll
        for (uint32_t j = 1; j < MAXL; ++j)
            if (j < n_lds && tid >= ubase[j]) l = j;
        uint32_t lstart = td.start[0], lbase = 0;
#pragma unroll
        for (uint32_t j = 1; j < MAXL; ++j)
            if (j == l) {
                lstart = td.start[j];
                lbase = ubase[j];
                u_pstart = td.start[j - 1];
                u_pbase = ubase[j - 1];
            }
        if (is_upper) {
            my_level = l;
            u_row = lstart + (tid - lbase);
        }
    }
    // (the old value's base differs per lane: upper rows read the live column, chain nodes their PRE-frame snapshot, see TreeArgs)
    const float* const u_old_src = is_chain ? a.snap_read : c.global;
    const uint32_t g_off = (s_start + wbase0) * 48u;
    const uint32_t u_p = at32<uint32_t>(a.parent_idx, u_row * 4u);
    const V3 u_s = ld3_32(c.scale, u_row), u_t = ld3_32(c.translation, u_row);
    const V4 u_q = ld4_32(c.rotation, u_row);
    const Affine u_old = ld_affine(u_old_src, u_row);
    const NodeRaw u_raw = node_raw<ALL_DIRTY>(a, u_row, true);
    FAN_STAMP(1);
    __builtin_amdgcn_sched_barrier(0);  // nothing below may move above: the scheduler otherwise consumes the first loads early
    // consume, in issue order
    float4* const stage = lds_old_stage + wv * 192u;
    // per row, for the level steps: the parent's LDS slot (levels >= 1 of the tile) or its global row (level 0 of a tile
    // below another launch), and the rule's inputs
    if (is_upper || is_chain) {
        const uint32_t slot = is_upper ? tid : TILE_LIGHT_UCAP + (tid - FAN_CHAIN_LANE0);
        lds_put(lds_g, slot, affine_from_srt(u_s, u_q, u_t));
        lds_put(lds_old, slot, u_old);
        const bool root_level = is_upper ? (ROOTS && my_level == 0) : (tid - FAN_CHAIN_LANE0 + 1u == chain_len);
        const NodeIn in = node_inputs_raw(a, u_row, root_level, u_raw);
        lds_in[slot] = (uint8_t)((in.tree_changed ? 1u : 0u) | (in.root_write ? 2u : 0u));
        if (is_upper) {
            lds_pslot[tid] = my_level ? u_pbase + (u_p - u_pstart) : u_p;
            lds_row[tid] = u_row;
            lds_level[tid] = (uint8_t)my_level;
        }
    }
    __syncthreads();
    FAN_STAMP(2);

    // ---- the ancestor chain: forest root first, down to the tile's parent -- the products the owning tiles compute.
    // One quad of wave 0 (a column per lane); the node just computed is handed on through LDS. ----
    if (chain_len) {
        if (wv == 0) {
            const uint32_t cc = lane & 3u;
            const bool on = lane < 4u;
            Affine g = {};
            bool chg = false;
            uint32_t k = chain_len;
            // the next node's inputs are fetched while this one is multiplied (nothing in the loop waits for LDS but them)
            uint32_t in_n = lds_in[TILE_LIGHT_UCAP + k - 1u];
            V3 loc_n = lds_col(lds_g, TILE_LIGHT_UCAP + k - 1u, cc), old_n = lds_col(lds_old, TILE_LIGHT_UCAP + k - 1u, cc);
            while (k-- > 0) {
                const uint32_t in_k = in_n;
                const V3 loc_k = loc_n, old_k = old_n;
                if (k) {
                    in_n = lds_in[TILE_LIGHT_UCAP + k - 1u];
                    loc_n = lds_col(lds_g, TILE_LIGHT_UCAP + k - 1u, cc);
                    old_n = lds_col(lds_old, TILE_LIGHT_UCAP + k - 1u, cc);
                }
                V3 cur_c;
                chg = quad_node_apply(on, k + 1u == chain_len, a.static_opt != 0, in_k, g, chg, loc_k, old_k, cc, lane, &cur_c);
                // every lane of the quad gets all four columns: quad broadcasts (DPP), no trip through LDS
                g.m.x_axis = quad_bcast(cur_c, 0);
                g.m.y_axis = quad_bcast(cur_c, 1);
                g.m.z_axis = quad_bcast(cur_c, 2);
                g.t = quad_bcast(cur_c, 3);
            }
            if (lane == 0) {
                lds_put(lds_chain_g, 0, g);
                lds_chain_chg = chg ? 1u : 0u;
            }
        }
        __syncthreads();
    }

    // The static-scene rule for a whole tile at once (systems.rs:708-714 skips a node when neither its subtree nor its parent
    // changed): the tile's parent did not change (the chain says so; the top rows of a tile of forest roots have none) and none
    // of the tile's top rows carries TransformTreeChanged or is assigned as a root or flat row this frame -- then, by induction
    // down the levels, every row of the tile is skipped: it keeps its GlobalTransform and its change tick stays.  Such a tile
    // asks for nothing more (two thirds of its bytes) and leaves: the trees of a forest that did not move cost a first burst.
    // Tiles that mirror rows into the chain snapshot take the long way.
    if constexpr (!ALL_DIRTY) {
        if (a.static_opt && (chain_len || ROOTS) && n_lds && (!snap_out || td.start[0] >= a.snap_rows)) {
            const bool top_marked = tid < td.count[0] && (lds_in[tid] & 3u) != 0;
            const bool parent_changed = chain_len != 0u && lds_chain_chg != 0u;
            if (__syncthreads_or((top_marked || parent_changed) ? 1 : 0) == 0) {
                if (tid < U) at32w<uint8_t>(a.g_changed_bytes, lds_row[tid]) = 0;
                for (uint32_t i = tid; i < s_count; i += 256u) at32w<uint8_t>(a.g_changed_bytes, s_start + i) = 0;
                return;
            }
        }
    }

    FAN_STAMP(3);
    // the last level's own inputs: requested here, they travel under the level steps (whose barriers order LDS only)
    const uint32_t f_p = at32<uint32_t>(a.parent_idx, s_row * 4u);
    // (a hierarchy too big for the Infinity Cache: the streamed level's inputs past the caches -- launch-uniform, TreeArgs::pretest bit 1)
    typedef float v4f_t __attribute__((ext_vector_type(4)));
    const bool nt = (a.pretest & 2u) != 0;
    V3 f_s, f_t;
    V4 f_q;
    if (nt) {
        const float* sp_ = &at32<float>(c.scale, s_row * 12u);
        const float* tp_ = &at32<float>(c.translation, s_row * 12u);
        f_s = V3{__builtin_nontemporal_load(sp_), __builtin_nontemporal_load(sp_ + 1), __builtin_nontemporal_load(sp_ + 2)};
        f_t = V3{__builtin_nontemporal_load(tp_), __builtin_nontemporal_load(tp_ + 1), __builtin_nontemporal_load(tp_ + 2)};
        const v4f_t fq_ = __builtin_nontemporal_load(&at32<v4f_t>(c.rotation, s_row * 16u));
        f_q = V4{fq_.x, fq_.y, fq_.z, fq_.w};
    } else {
        f_s = ld3_32(c.scale, s_row);
        f_t = ld3_32(c.translation, s_row);
        f_q = ld4_32(c.rotation, s_row);
    }
    const NodeRaw f_raw = node_raw<ALL_DIRTY>(a, s_row, s_root_level);
    // ---- LDS-resident levels: four lanes per row.  A thread's rows are fixed before the loop -- slot tid / 4 and, for tiles with
    // more than 64 upper rows, slot 64 + tid / 4 (TILE_LIGHT_UCAP <= 128) -- and everything about them that no other row's
    // result changes (level, parent slot, the rule's inputs, this lane's column of the local affine and of the old value) is
    // fetched here, once: a level step is then the parent's read, the product, the compare and the write, nothing else. ----
    static_assert(TILE_LIGHT_UCAP <= 128u, "two rows per quad cover the upper rows");
    bool any_chg = false;
    if (n_lds) {
        const uint32_t cc = tid & 3u;
        uint32_t q_slot[2], q_level[2], q_ps[2], q_in[2];
#pragma unroll
        for (uint32_t h = 0; h < 2u; ++h) {
            const uint32_t slot = h * 64u + (tid >> 2);
            const bool on = slot < U;
            const uint32_t sl = on ? slot : 0u;
            q_slot[h] = sl;
            q_level[h] = on ? (uint32_t)lds_level[sl] : 0xFFFFFFFFu;
            q_ps[h] = lds_pslot[sl];
            q_in[h] = lds_in[sl];
        }
        for (uint32_t l = 0; l < n_lds; ++l) {
#pragma unroll
            for (uint32_t h = 0; h < 2u; ++h) {
                const bool on = q_level[h] == l;
                if (__builtin_amdgcn_ballot_w64(on) == 0ull) continue;  // (wave-uniform) no row of this level in the wave
                Affine gp = {};
                bool p_changed = false;
                if (l) {
                    gp = lds_affine(lds_g, q_ps[h]);
                    p_changed = lds_chg[q_ps[h]] != 0;
                } else if (!ROOTS) {
                    if (chain_len) {
                        gp = lds_affine(lds_chain_g, 0);
                        p_changed = lds_chain_chg != 0;
                    } else if (on) {  // a tile below another launch: the parent is in global memory
                        gp = ld_affine(c.global, q_ps[h]);
                        p_changed = a.g_changed_bytes[q_ps[h]] != 0;
                    }
                }
                V3 cur_c;
                const bool chg = quad_node_apply(on, ROOTS && l == 0, a.static_opt != 0, q_in[h], gp, p_changed, lds_col(lds_g, q_slot[h], cc),
                                                 lds_col(lds_old, q_slot[h], cc), cc, lane, &cur_c);
                if (on) {
                    lds_put_col(lds_g, q_slot[h], cc, cur_c);  // in place: the quad's lanes read and write their own column only
                    any_chg = any_chg || chg;
                    if (cc == 0) lds_chg[q_slot[h]] = chg ? 1 : 0;  // (its global copy goes out with the write-back)
                }
            }
            MI_WG_LDS_BARRIER();
        }
    }
    // write-back of the upper rows: slots and rows are contiguous per level -> straight float4 copies; a tile in which
    // nothing changed writes nothing (unchanged rows hold their old bytes), the snapshot always tracks the current value
    FAN_STAMP(4);
    // the last level's old GlobalTransforms (three contiguous 1 KB rows per wave): requested here, they travel under the
    // write-back and the last level's own products, and land in the transpose buffers -- the memory the upper rows' old values
    // occupied until the level steps had read them
    bool flush_live = false;
    if (n_lds) flush_live = __syncthreads_or(any_chg ? 1 : 0) != 0;  // (in front of the loads: the barrier drains the load counter)
    float4 f_g0, f_g1, f_g2;
    if (nt) {
        const v4f_t g0_ = __builtin_nontemporal_load(&at32<v4f_t>(c.global, g_off + (lane < last0 ? lane : last0) * 16u));
        const v4f_t g1_ = __builtin_nontemporal_load(&at32<v4f_t>(c.global, g_off + (64u + lane < last0 ? 64u + lane : last0) * 16u));
        const v4f_t g2_ = __builtin_nontemporal_load(&at32<v4f_t>(c.global, g_off + (128u + lane < last0 ? 128u + lane : last0) * 16u));
        f_g0 = make_float4(g0_.x, g0_.y, g0_.z, g0_.w);
        f_g1 = make_float4(g1_.x, g1_.y, g1_.z, g1_.w);
        f_g2 = make_float4(g2_.x, g2_.y, g2_.z, g2_.w);
    } else {
        f_g0 = at32<float4>(c.global, g_off + (lane < last0 ? lane : last0) * 16u);
        f_g1 = at32<float4>(c.global, g_off + (64u + lane < last0 ? 64u + lane : last0) * 16u);
        f_g2 = at32<float4>(c.global, g_off + (128u + lane < last0 ? 128u + lane : last0) * 16u);
    }
    if (n_lds) {
        if (tid < U) at32w<uint8_t>(a.g_changed_bytes, lds_row[tid]) = lds_chg[tid];
        if (flush_live || snap_out) {
            for (uint32_t f = tid; f < 3u * U; f += 256u) {  // float4 f of the slots: lanes walk the rows' 48 bytes contiguously
                const uint32_t slot = f / 3u, row = lds_row[slot];
                const uint32_t off = row * 48u + (f - slot * 3u) * 16u;
                const float4 v = lds_g[f];
                if (flush_live) at32w<float4>(c.global, off) = v;
                if (snap_out && row < a.snap_rows) at32w<float4>(snap_out, off) = v;
            }
        }
    }

    // ---- the last level: one row per lane; a wider level (a single node with more than 256 children) takes more batches ----
    auto batch = [&](uint32_t base, uint32_t p, V3 sc, V4 q, V3 t, NodeRaw raw, bool first) {
        const uint32_t i = base + tid;
        const bool live = i < s_count;
        const uint32_t row = s_start + i;
        const uint32_t wbase = base + wv * 64u;
        const uint32_t wave_lim = wbase < s_count ? (s_count - wbase < 64u ? s_count - wbase : 64u) * 3u : 0u;
        Affine local = {}, gp = {};
        bool p_changed = false;
        if (live) {
            local = affine_from_srt(sc, q, t);
            if (!s_root_level) {
                if (n_lds) {
                    const uint32_t slot = s_pbase + (p - s_pstart);
                    gp = lds_affine(lds_g, slot);
                    p_changed = lds_chg[slot] != 0;
                } else if (chain_len) {
                    gp = lds_affine(lds_chain_g, 0);
                    p_changed = lds_chain_chg != 0;
                } else {
                    gp = ld_affine(c.global, p);
                    p_changed = a.g_changed_bytes[p] != 0;
                }
            }
        }
        if (first) {  // as late as possible: everything above runs under the loads
            stage[lane] = f_g0;
            stage[64u + lane] = f_g1;
            stage[128u + lane] = f_g2;
        }
        MI_WAVE_LDS_SYNC();
        Affine cur = {};
        bool chg = false;
        if (live) {
            chg = node_apply_lazy(s_root_level, a.static_opt != 0, node_inputs_raw(a, row, s_root_level, raw), gp, p_changed, local,
                                  [&] { return lds_affine(stage, lane); }, &cur);
            at32w<uint8_t>(a.g_changed_bytes, row) = chg ? 1 : 0;
            if (snap_out && row < a.snap_rows) st_affine(snap_out, row, cur);
        }
        const unsigned long long cm = __ballot(chg), lm = __ballot(live);
        if (cm == lm) {  // the whole wave changed (the dirty-tree case): transpose back, three contiguous 1 KB rows out
            MI_WAVE_LDS_SYNC();
            lds_put(stage, lane, cur);
            MI_WAVE_LDS_SYNC();
            const uint32_t doff = (s_start + wbase) * 48u;
#pragma unroll
            for (uint32_t k = 0; k < 3u; ++k) {
                const uint32_t j = k * 64u + lane;
                if (j < wave_lim) {
                    const float4 v = stage[j];
                    if (nt) __builtin_nontemporal_store(v4f_t{v.x, v.y, v.z, v.w}, &at32w<v4f_t>(c.global, doff + j * 16u));  // (big hierarchies: nobody finds it cached)
                    else at32w<float4>(c.global, doff + j * 16u) = v;
                }
            }
        } else if (chg) {
            st_affine(c.global, row, cur);
        }
        // (the rule's arguments are read here, not at kernel entry: 80 -> 31 spilled SGPRs, 30.8 -> 30.1 us per launch)
        if constexpr (CULL) tile_cull_rows(kernarg_late<Columns>(0), kernarg_late<TreeCull>(FANS_CULL_KERNARG), lane, live, row, cur);
    };
    FAN_STAMP(5);
    batch(0u, f_p, f_s, f_q, f_t, f_raw, true);
    FAN_STAMP(6);
    for (uint32_t base = 256u; base < s_count; base += 256u) {  // rare
        const RowFetch fx = fetch_row(c, a.parent_idx, s_start, s_count, base, tid, lane, wv, s_root_level);
        const uint32_t xrow = s_start + (base + tid < s_count ? base + tid : 0u);
        const NodeRaw xraw = node_raw<ALL_DIRTY>(a, xrow, s_root_level);
        MI_WAVE_LDS_SYNC();
        stage[lane] = fx.g0;
        stage[64u + lane] = fx.g1;
        stage[128u + lane] = fx.g2;
        batch(base, fx.p, fx.s, fx.q, fx.t, xraw, false);
    }
    if constexpr (CULL) {  // the upper rows: GlobalTransforms in LDS, slot = thread (waves 0 and 1)
        if (wv * 64u < U) {
            const bool on = tid < U;
            tile_cull_rows(kernarg_late<Columns>(0), kernarg_late<TreeCull>(FANS_CULL_KERNARG), lane, on, on ? lds_row[tid] : 0u, lds_affine(lds_g, on ? tid : 0u));
        }
    }
    if (a.trace && tid == 0) {
        __builtin_amdgcn_s_waitcnt(0);  // stores drained
        const unsigned long long t7 = wall_clock64();
#pragma unroll
        for (uint32_t i = 0; i < 7u; ++i) a.trace[(size_t)tile * 8u + i] = ts[i];
        a.trace[(size_t)tile * 8u + 7u] = t7;
    }
#undef FAN_STAMP
}

// ---------------------------------------------------------------------------------------------
// A wide level as a stream (the deepest level of a big tree holds most of its rows).  One row per lane, no loops, so
// the register budget allows twice the waves of the tile kernel and the level moves at the flat kernel's pace: T / R / S,
// parent index and the old GlobalTransform in (the latter as three contiguous 1 KB wave rows through a wave-private LDS
// transpose), the parent's GlobalTransform and change flag gathered from the level above (complete: an earlier launch;
// the four children of a node sit in neighbouring lanes, so the gather touches 16 x 48 contiguous bytes per wave in a
// 4-ary tree), the new GlobalTransform out through the same transpose.
// Algorithmic bytes per row: 40 + 4 + 48 + 48 + 1 = 141 (+ 48 / fan-out for the parents, L2 hits).
// ---------------------------------------------------------------------------------------------
template <bool ROOT>
__global__ void __launch_bounds__(256) k_propagate_level(Columns c, TreeArgs a, uint32_t start, uint32_t count) {
    __shared__ float4 lds_stage[4][192];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const uint32_t i = blockIdx.x * 256u + tid;
    const bool live = i < count;
    const uint32_t row = start + i;
    const uint32_t wbase = blockIdx.x * 256u + wv * 64u;
    const uint32_t wave_lim = wbase < count ? (count - wbase < 64u ? count - wbase : 64u) * 3u : 0u;
    float4* stage = lds_stage[wv];
    float4* const gw = reinterpret_cast<float4*>(c.global) + 3ull * (start + wbase);
    // the wave's old GlobalTransforms first: everything issued after them can stay in flight while they are transposed.
    // Lanes past the end re-read the last float4 of the level (a clamped index, not a `cond ? load : zero`: the latter becomes
    // a select between a global and a stack ADDRESS -- a flat load and a scratch slot)
    const float4* const gr = reinterpret_cast<const float4*>(c.global) + 3ull * (start + (wave_lim ? wbase : 0u));
    const uint32_t last4 = wave_lim ? wave_lim - 1u : 0u;
    const float4 o0 = gr[lane < last4 ? lane : last4];
    const float4 o1 = gr[64u + lane < last4 ? 64u + lane : last4];
    const float4 o2 = gr[128u + lane < last4 ? 128u + lane : last4];
    V3 t = {}, sc = {};
    V4 q = {};
    Affine gp = {};
    bool p_changed = false;
    if (live) {
        t = ld3(c.translation, row);
        q = ld4(c.rotation, row);
        sc = ld3(c.scale, row);
        if (!ROOT) {  // (level 0 of a hierarchy swept by levels: forest roots and flat rows, no parents)
            const uint32_t p = a.parent_idx[row];
            gp = ld_affine(c.global, p);
            p_changed = a.g_changed_bytes[p] != 0;
        }
    }
    stage[lane] = o0;
    stage[64u + lane] = o1;
    stage[128u + lane] = o2;
    MI_WAVE_LDS_SYNC();
    const Affine old = lds_affine(stage, lane);
    Affine cur = old;
    bool chg = false;
    if (live) {
        chg = node_update(a, ROOT, row, gp, p_changed, affine_from_srt(sc, q, t), old, &cur);
        a.g_changed_bytes[row] = chg ? 1 : 0;
    }
    const unsigned long long cm = __ballot(chg), lm = __ballot(live);
    if (cm == lm) {  // the whole wave changed (the dirty-tree case): transpose back, three contiguous 1 KB rows out
        MI_WAVE_LDS_SYNC();
        lds_put(stage, lane, cur);
        MI_WAVE_LDS_SYNC();
#pragma unroll
        for (uint32_t k = 0; k < 3u; ++k) {
            const uint32_t j = k * 64u + lane;
            if (j < wave_lim) gw[j] = stage[j];
        }
    } else if (chg) {
        st_affine(c.global, row, cur);
    }
}

// ---------------------------------------------------------------------------------------------
// A hierarchy as narrow as a chain -- EVERY level at most a wave wide: transform_hierarchy.rs's `chain` (2 500 levels of one node), a
// rope, a single rig -- is a chain of dependent products whatever runs it: its cost is levels x the latency of one level step.  Through
// tiles that was a launch per TILE_MAX_LEVELS levels (155 dependent launches, 1.3 ms per frame for the 2 500-node chain; 310 and 1.8 ms
// with 8-level tiles).  Here ONE wave walks the whole hierarchy and nothing but arithmetic sits between two levels: rows are in level
// order, so consecutive levels are consecutive rows -- a chunk of whole levels (<= NARROW_CHUNK rows) is fetched with coalesced loads,
// turned into local affines and parked in LDS with the old values and the rule's inputs; the levels of the chunk then run one after
// the other, each level's inputs read from LDS while the level before is multiplied, the results handed from level to level in
// REGISTERS: through quad broadcasts / nothing at all where every node's parent sits in the lane group its child will occupy (a chain,
// ropes side by side), through ds_bpermute otherwise.  QUAD (levels of <= 16 rows): a node is a quad of lanes, a column each, as in the
// tiles' chains -- a quarter of the dependent instructions per level; otherwise a lane per row.  Same rule (node_apply), same products
// in the same order: same bits.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t NARROW_CHUNK = 256;
__device__ __forceinline__ float shfl_f(float v, uint32_t src_lane) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), __float_as_int(v)));
}
__device__ __forceinline__ V3 shfl3(V3 v, uint32_t src_lane) { return V3{shfl_f(v.x, src_lane), shfl_f(v.y, src_lane), shfl_f(v.z, src_lane)}; }
template <bool ALL_DIRTY, bool QUAD>
__global__ void __launch_bounds__(64) k_propagate_narrow(Columns c, TreeArgs a, const uint32_t* __restrict__ level_offsets, uint32_t n_levels) {
    __shared__ float4 lds_local[NARROW_CHUNK * 3];  // per staged row: its local affine
    __shared__ float4 lds_oldg[NARROW_CHUNK * 3];   // its GlobalTransform before this frame
    __shared__ uint32_t lds_par[NARROW_CHUNK];      // its parent's row
    __shared__ uint8_t lds_in[NARROW_CHUNK];        // bit0 TransformTreeChanged, bit1 the level-0 assignment happens
    __shared__ uint32_t lds_off[NARROW_CHUNK + 2];  // level_offsets[l0 ..]
    const uint32_t lane = threadIdx.x;
    const uint32_t unit = QUAD ? lane >> 2 : lane;  // the node of its level this lane works on
    const uint32_t cc = lane & 3u;                  // QUAD: the column
    uint32_t l0 = 0, prev_start = 0;
    // the level before, in registers: QUAD the lane's column of its node, otherwise the lane's whole node
    V3 pq = {};
    Affine pr = {};
    bool p_chg = false;
    while (l0 < n_levels) {
        // the chunk's levels: as many whole levels from l0 on as hold <= NARROW_CHUNK rows together (a level holds <= 64: at least four)
        for (uint32_t k = lane; k < NARROW_CHUNK + 2u; k += 64u) lds_off[k] = level_offsets[l0 + k <= n_levels ? l0 + k : n_levels];
        MI_WAVE_LDS_SYNC();
        const uint32_t row0 = lds_off[0];
        uint32_t n_lv = 0;
        {
            // first k (1-based) whose end exceeds the budget or the hierarchy; levels are non-empty, so ends are increasing
            uint32_t best = 0xFFFFFFFFu;
            for (uint32_t k = lane + 1u; k <= NARROW_CHUNK + 1u; k += 64u)
                if ((lds_off[k] - row0 > NARROW_CHUNK || l0 + k > n_levels) && k < best) best = k;
#pragma unroll
            for (uint32_t off = 32u; off; off >>= 1) {
                const uint32_t o = (uint32_t)__shfl_xor((int)best, (int)off, 64);
                best = o < best ? o : best;
            }
            n_lv = __builtin_amdgcn_readfirstlane(best - 1u);  // (>= 1: the planner takes this kernel only when every level holds 1 .. 64 rows)
            const uint32_t left = n_levels - l0;  // (and whatever the planner let through: never past the staged offsets or the hierarchy)
            n_lv = n_lv > NARROW_CHUNK ? NARROW_CHUNK : n_lv;
            n_lv = n_lv > left ? left : n_lv;
            n_lv = n_lv ? n_lv : 1u;
        }
        const uint32_t rows = lds_off[n_lv] - row0;
        // ---- stage the chunk: coalesced loads, a lane per row, four rounds
#pragma unroll
        for (uint32_t j = 0; j < NARROW_CHUNK / 64u; ++j) {
            const uint32_t i = j * 64u + lane;
            if (i < rows) {
                const uint32_t row = row0 + i;
                const bool root_level = l0 == 0 && i < lds_off[1] - row0;
                const V3 sc = ld3_32(c.scale, row), t = ld3_32(c.translation, row);
                const V4 q = ld4_32(c.rotation, row);
                const NodeRaw raw = node_raw<ALL_DIRTY>(a, row, root_level);
                lds_put(lds_local, i, affine_from_srt(sc, q, t));
                lds_put(lds_oldg, i, ld_affine(c.global, row));
                lds_par[i] = at32<uint32_t>(a.parent_idx, row * 4u);
                const NodeIn in = node_inputs_raw(a, row, root_level, raw);
                lds_in[i] = (uint8_t)((in.tree_changed ? 1u : 0u) | (in.root_write ? 2u : 0u));
            }
        }
        MI_WAVE_LDS_SYNC();
        // ---- the chunk's levels, one after the other.  One wave executes in order, so an LDS read only travels under arithmetic that is
        // issued BEHIND it: level k + 1's inputs are requested (from offsets read an iteration earlier still) before level k is
        // multiplied, and looked at in the next iteration -- the loop never waits for an address it has just asked for.
        struct LevelIn {
            uint32_t row, par, in;
            bool on;
            Affine local, old;
            V3 local_c, old_c;
        };
        auto fetch = [&](uint32_t start, uint32_t end) {  // the level's rows are [start, end)
            LevelIn f;
            f.on = unit < end - start;
            f.row = start + (f.on ? unit : 0u);
            const uint32_t i = f.row - row0 < NARROW_CHUNK ? f.row - row0 : NARROW_CHUNK - 1u;  // (asked for unconditionally, also past the chunk's last level)
            f.par = lds_par[i];
            f.in = lds_in[i];
            if constexpr (QUAD) {
                f.local_c = lds_col(lds_local, i, cc);
                f.old_c = lds_col(lds_oldg, i, cc);
            } else {
                f.local = lds_affine(lds_local, i);
                f.old = lds_affine(lds_oldg, i);
            }
            return f;
        };
        uint32_t o_k = row0, o_k1 = lds_off[1], o_k2 = lds_off[2];  // level_offsets[l0 + k], [.. + 1], [.. + 2] (clamped to the last)
        LevelIn nx = fetch(o_k, o_k1);
        uint32_t before = prev_start;  // where the level above level k starts
        for (uint32_t k = 0; k < n_lv; ++k) {
            const LevelIn cu = nx;
            const bool root_level = l0 + k == 0;
            const uint32_t slot = root_level ? unit : (cu.par - before) & (QUAD ? 15u : 63u);
            const bool same = __all(!cu.on || slot == unit) != 0;  // every parent sits where its child does (wave-uniform)
            // the parents come over FIRST: LDS answers in order, so whatever is asked behind them (the next level's inputs) can stay in
            // flight while they are used
            bool pc = p_chg;
            V3 pq_here = pq;   // QUAD: column cc of the parent, in the child's quad
            Affine gp = pr;    // otherwise: the parent
            if (!same) {
                if constexpr (QUAD) {
                    pq_here = shfl3(pq, slot * 4u + cc);
                    pc = __builtin_amdgcn_ds_bpermute((int)(slot << 4), p_chg ? 1 : 0) != 0;
                } else {
                    gp.m.x_axis = shfl3(pr.m.x_axis, slot);
                    gp.m.y_axis = shfl3(pr.m.y_axis, slot);
                    gp.m.z_axis = shfl3(pr.m.z_axis, slot);
                    gp.t = shfl3(pr.t, slot);
                    pc = __builtin_amdgcn_ds_bpermute((int)(slot << 2), p_chg ? 1 : 0) != 0;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t o_k3 = lds_off[k + 3u <= NARROW_CHUNK + 1u ? k + 3u : NARROW_CHUNK + 1u];
            nx = fetch(o_k1, o_k2);  // (no branch around it: the waits below are counted along the shortest path)
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (QUAD) {
                gp.m.x_axis = quad_bcast(pq_here, 0);
                gp.m.y_axis = quad_bcast(pq_here, 1);
                gp.m.z_axis = quad_bcast(pq_here, 2);
                gp.t = quad_bcast(pq_here, 3);
                V3 cur_c;
                const bool chg = quad_node_apply(cu.on, root_level, a.static_opt != 0, cu.in, gp, pc, cu.local_c, cu.old_c, cc, lane, &cur_c);
                if (cu.on) {
                    if (cc == 0u) at32w<uint8_t>(a.g_changed_bytes, cu.row) = chg ? 1 : 0;
                    if (chg) at32w<F3>(c.global, cu.row * 48u + cc * 12u) = F3{cur_c.x, cur_c.y, cur_c.z};
                }
                pq = cur_c;
                p_chg = chg;
            } else {
                NodeIn in;
                in.tree_changed = (cu.in & 1u) != 0;
                in.root_write = (cu.in & 2u) != 0;
                Affine cur;
                const bool chg = node_apply(root_level, a.static_opt != 0, in, gp, pc, cu.local, cu.old, &cur);
                if (cu.on) {
                    at32w<uint8_t>(a.g_changed_bytes, cu.row) = chg ? 1 : 0;
                    if (chg) st_affine(c.global, cu.row, cur);
                }
                pr = cur;
                p_chg = chg;
            }
            before = o_k;
            o_k = o_k1;
            o_k1 = o_k2;
            o_k2 = o_k3;
        }
        prev_start = lds_off[n_lv - 1u];
        l0 += n_lv;
        MI_WAVE_LDS_SYNC();  // (the next chunk overwrites the offsets and the staging)
    }
}

// ---------------------------------------------------------------------------------------------
// A FOREST OF SMALL TREES, a wave per tree (round 6): transform_hierarchy.rs's humanoids_* -- 4 000 rigs of 68 nodes in 13 levels -- and
// what a game's scene mostly is.  Through the workgroup tiles a rig held a 256-thread workgroup, most of whose lanes had no row, for a
// chain of 13 barriered level steps (23.5 us per all-dirty frame: two rounds of 4 000 tiles at 8 per CU).  Here the tile is the
// one-wave walk of k_propagate_narrow above, per TREE instead of per hierarchy: a wave stages its tile's rows (TileDesc: per level a
// row range -- the rows of one tree are contiguous inside each level, not across levels) into its own LDS with coalesced loads and
// runs the levels one after the other, results handed down in registers (quad broadcasts / ds_bpermute), nothing but arithmetic
// between two levels and no workgroup barrier anywhere.  8 KB of LDS per wave: 19 tiles per CU, 4 800 trees in flight -- the 4 000
// rigs are ONE round.  QUAD (no level of a tile holds more than 16 rows): a node per quad of lanes, a column each.  Tiles are
// forest-root tiles (kind TILE_ROOTS): the planner takes this path only when EVERY root's tree fits a wave tile (ctx_hierarchy.cpp).
// Same rule (node_apply / quad_node_apply), same products in the same order: same bits.
// ---------------------------------------------------------------------------------------------
template <bool ALL_DIRTY, bool QUAD>
__global__ void __launch_bounds__(64) k_propagate_wave_tiles(Columns c, TreeArgs a, const TileDesc* __restrict__ wtiles) {
    __shared__ float4 lds_local[WAVE_TILE_ROWS * 3];  // per staged row: its local affine
    __shared__ float4 lds_oldg[WAVE_TILE_ROWS * 3];   // its GlobalTransform before this frame
    __shared__ uint32_t lds_par[WAVE_TILE_ROWS];      // its parent's row
    __shared__ uint8_t lds_in[WAVE_TILE_ROWS];        // bit0 TransformTreeChanged, bit1 the level-0 assignment happens
    __shared__ uint32_t lds_lv[(TILE_MAX_LEVELS + 3u) * 4u];  // per level of the tile: LDS base, first row, rows; three empty levels behind the last
    const uint32_t lane = threadIdx.x;
    const uint32_t unit = QUAD ? lane >> 2 : lane;  // the node of its level this lane works on
    const uint32_t cc = lane & 3u;                  // QUAD: the column
    const TileDesc td = load_tile_desc<TILE_MAX_LEVELS>(wtiles + xcd_contiguous_tile());
    const uint32_t L = td.n_levels;
    // the level table: lane k writes level k's entry (LDS base = rows of the levels above it)
    uint32_t rows = 0;
    {
        uint32_t base = 0, st = 0, cn = 0;
#pragma unroll
        for (uint32_t k = 0; k < TILE_MAX_LEVELS; ++k) {
            const uint32_t cnt_k = k < L ? td.count[k] : 0u;
            if (lane == k) {
                base = rows;
                st = td.start[k];
                cn = cnt_k;
            }
            rows += cnt_k;
        }
        if (lane >= TILE_MAX_LEVELS) base = rows;
        if (lane < TILE_MAX_LEVELS + 3u) {
            lds_lv[lane * 4u] = base;
            lds_lv[lane * 4u + 1u] = st;
            lds_lv[lane * 4u + 2u] = cn;
        }
    }
    // where staged row i lives: its level's first row and LDS base (scalars of the descriptor, compared lane by lane)
    auto row_of = [&](uint32_t i, bool* root_level) {
        uint32_t lstart = td.start[0], lbase = 0, acc = 0;
        bool root = true;
#pragma unroll
        for (uint32_t k = 0; k < TILE_MAX_LEVELS; ++k) {
            if (k < L && i >= acc) {
                lstart = td.start[k];
                lbase = acc;
                root = k == 0u;
            }
            acc += k < L ? td.count[k] : 0u;
        }
        *root_level = root;
        return lstart + (i - lbase);
    };
    constexpr uint32_t ROUNDS = (WAVE_TILE_ROWS + 63u) / 64u;
    // ---- a frame in which little moved: the tile's flags first (as the workgroup tiles' pretest, k_propagate_fans): under the
    // static-scene rule a tree none of whose rows is marked (TransformTreeChanged: a moved descendant marks every ancestor up to the
    // root) or assigned as a root / flat row keeps every GlobalTransform and every tick -- one small round trip instead of the tile
    if constexpr (!ALL_DIRTY) {
        if ((a.pretest & 1u) && a.static_opt) {
            bool hot = false;
#pragma unroll
            for (uint32_t j = 0; j < ROUNDS; ++j) {
                const uint32_t i = j * 64u + lane;
                bool root_level;
                const uint32_t row = row_of(i < rows ? i : 0u, &root_level);
                const NodeIn in = node_inputs_raw(a, row, root_level, node_raw<false>(a, row, root_level));
                hot = hot || (i < rows && (in.tree_changed || in.root_write));
            }
            if (__ballot(hot) == 0ull) {
#pragma unroll
                for (uint32_t j = 0; j < ROUNDS; ++j) {
                    const uint32_t i = j * 64u + lane;
                    bool root_level;
                    const uint32_t row = row_of(i < rows ? i : 0u, &root_level);
                    if (i < rows) at32w<uint8_t>(a.g_changed_bytes, row) = 0;
                }
                return;
            }
        }
    }
    // ---- stage the tile: a lane per row, coalesced inside each level
    bool hot = false;
#pragma unroll
    for (uint32_t j = 0; j < ROUNDS; ++j) {
        const uint32_t i = j * 64u + lane;
        if (i < rows) {
            bool root_level;
            const uint32_t row = row_of(i, &root_level);
            const V3 sc = ld3_32(c.scale, row), t = ld3_32(c.translation, row);
            const V4 q = ld4_32(c.rotation, row);
            const NodeRaw raw = node_raw<ALL_DIRTY>(a, row, root_level);
            lds_put(lds_local, i, affine_from_srt(sc, q, t));
            lds_put(lds_oldg, i, ld_affine(c.global, row));
            lds_par[i] = at32<uint32_t>(a.parent_idx, row * 4u);
            const NodeIn in = node_inputs_raw(a, row, root_level, raw);
            lds_in[i] = (uint8_t)((in.tree_changed ? 1u : 0u) | (in.root_write ? 2u : 0u));
            hot = hot || in.tree_changed || in.root_write;
        }
    }
    MI_WAVE_LDS_SYNC();
    if constexpr (!ALL_DIRTY) {  // (the same rule without the pretest's extra round trip: everything is here, nothing is computed)
        if (a.static_opt && __ballot(hot) == 0ull) {
#pragma unroll
            for (uint32_t j = 0; j < ROUNDS; ++j) {
                const uint32_t i = j * 64u + lane;
                bool root_level;
                const uint32_t row = row_of(i < rows ? i : 0u, &root_level);
                if (i < rows) at32w<uint8_t>(a.g_changed_bytes, row) = 0;
            }
            return;
        }
    }
    // ---- the tile's levels, one after the other (the loop of k_propagate_narrow: level k + 1's inputs are requested before level k
    // is multiplied and looked at an iteration later -- the wave never waits for an address it has just asked for)
    struct LevelIn {
        uint32_t row, par, in;
        bool on;
        Affine local, old;
        V3 local_c, old_c;
    };
    auto fetch = [&](uint32_t base, uint32_t start, uint32_t cnt) {
        LevelIn f;
        f.on = unit < cnt;
        const uint32_t u = f.on ? unit : 0u;
        f.row = start + u;
        const uint32_t i = base + u < WAVE_TILE_ROWS ? base + u : WAVE_TILE_ROWS - 1u;  // (asked for unconditionally, also behind the last level)
        f.par = lds_par[i];
        f.in = lds_in[i];
        if constexpr (QUAD) {
            f.local_c = lds_col(lds_local, i, cc);
            f.old_c = lds_col(lds_oldg, i, cc);
        } else {
            f.local = lds_affine(lds_local, i);
            f.old = lds_affine(lds_oldg, i);
        }
        return f;
    };
    uint32_t b0 = lds_lv[0], s0 = lds_lv[1], c0 = lds_lv[2];  // level k
    uint32_t b1 = lds_lv[4], s1 = lds_lv[5], c1 = lds_lv[6];  // level k + 1
    LevelIn nx = fetch(b0, s0, c0);
    uint32_t before = 0;  // first row of the level above level k
    V3 pq = {};           // the level before, in registers: QUAD the lane's column of its node, otherwise the lane's whole node
    Affine pr = {};
    bool p_chg = false;
    for (uint32_t k = 0; k < L; ++k) {
        const LevelIn cu = nx;
        const bool root_level = k == 0u;
        const uint32_t slot = root_level ? unit : (cu.par - before) & (QUAD ? 15u : 63u);
        const bool same = __all(!cu.on || slot == unit) != 0;  // every parent sits where its child does (wave-uniform)
        bool pc = p_chg;
        V3 pq_here = pq;
        Affine gp = pr;
        if (!same) {
            if constexpr (QUAD) {
                pq_here = shfl3(pq, slot * 4u + cc);
                pc = __builtin_amdgcn_ds_bpermute((int)(slot << 4), p_chg ? 1 : 0) != 0;
            } else {
                gp.m.x_axis = shfl3(pr.m.x_axis, slot);
                gp.m.y_axis = shfl3(pr.m.y_axis, slot);
                gp.m.z_axis = shfl3(pr.m.z_axis, slot);
                gp.t = shfl3(pr.t, slot);
                pc = __builtin_amdgcn_ds_bpermute((int)(slot << 2), p_chg ? 1 : 0) != 0;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t b2 = lds_lv[(k + 2u) * 4u], s2 = lds_lv[(k + 2u) * 4u + 1u], c2 = lds_lv[(k + 2u) * 4u + 2u];
        nx = fetch(b1, s1, c1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (QUAD) {
            gp.m.x_axis = quad_bcast(pq_here, 0);
            gp.m.y_axis = quad_bcast(pq_here, 1);
            gp.m.z_axis = quad_bcast(pq_here, 2);
            gp.t = quad_bcast(pq_here, 3);
            V3 cur_c;
            const bool chg = quad_node_apply(cu.on, root_level, a.static_opt != 0, cu.in, gp, pc, cu.local_c, cu.old_c, cc, lane, &cur_c);
            if (cu.on) {
                if (cc == 0u) at32w<uint8_t>(a.g_changed_bytes, cu.row) = chg ? 1 : 0;
                if (chg) at32w<F3>(c.global, cu.row * 48u + cc * 12u) = F3{cur_c.x, cur_c.y, cur_c.z};
            }
            pq = cur_c;
            p_chg = chg;
        } else {
            NodeIn in;
            in.tree_changed = (cu.in & 1u) != 0;
            in.root_write = (cu.in & 2u) != 0;
            Affine cur;
            const bool chg = node_apply(root_level, a.static_opt != 0, in, gp, pc, cu.local, cu.old, &cur);
            if (cu.on) {
                at32w<uint8_t>(a.g_changed_bytes, cu.row) = chg ? 1 : 0;
                if (chg) st_affine(c.global, cu.row, cur);
            }
            pr = cur;
            p_chg = chg;
        }
        before = s0;
        b0 = b1; s0 = s1; c0 = c1;
        b1 = b2; s1 = s2; c1 = c2;
    }
}

// ---------------------------------------------------------------------------------------------
// STRIPS (round 6): a lopsided or deep tree -- transform_hierarchy.rs's large_tree / deep_tree, 250 000 - 320 000 nodes in 19 / 26
// levels whose widths swell and shrink -- in ONE launch.  Through the workgroup tiles such a tree was three or four launches, each
// behind the one that wrote its parents (a subtree of 15 levels does not fit a tile's LDS rows, and below the first cut the tiles'
// top rows have hundreds of different parents: no single chain to re-evaluate): 27 - 31 us of 5 - 9 us tiles and the gaps between
// launches.  A strip (kernels.h) is a workgroup that walks a list of rounds -- up to 64 rows of one level --: first the CONE of its
// rows' ancestors, level by level from the forest's roots down (a contiguous row range per level; evaluated, never written -- the
// strips that own those rows form the same products in the same order), then the rows it owns.  A level's results stay in LDS for the
// level below; the table of rounds is known up front, so nothing is asked for behind a result; no wait for another workgroup,
// whatever the depth.  The cone's rows compare against (and, skipped under the static-scene rule, keep) their PRE-frame
// GlobalTransforms, which their owners may be rewriting in this very launch: they read the chain snapshot (TreeArgs::snap_read), as
// the chain tiles do.  Same rule (quad_node_apply), same products in the same order: same bits, same ticks.
// ---------------------------------------------------------------------------------------------
struct StripIn {
    V3 t, s;
    float4 q;
    uint32_t par;
    float4 o0, o1, o2;  // the old GlobalTransform as it was loaded (whole register quads: what the loop carries is what the loads wrote)
    NodeRaw raw;
};
// A strip's workgroup: four CONSUMER waves and a PRODUCER wave.  What a level costs its strip is the instructions a wave has to issue for
// it, 2 - 3.5 ns apiece (tools/probes/issue_rate_probe.hip: a dependent v_fma 3.5 ns, an independent one 2.1, a dependent LDS read 25, an LDS
// hand-over through a barrier 70): with everything in ONE wave's stream -- a round's 14 loads and their addresses, From(Transform), the
// parent's read, the product, the compare, the stores -- a round was 0.72 us whatever its rows.  So the producer does everything that
// does not hang on the level above -- the loads (a batch ahead of the one being staged, in registers), From(Transform), the rule's inputs -- and leaves a
// BATCH ready in one of two LDS slots of 64 rows: one round of up to 64 rows, or up to four consecutive NARROW levels (<= 16 rows
// each) of the strip -- the cone of a deep tree is a dozen levels of one to three rows.  The consumers run the dependent chain only:
// parent from LDS, product, set_if_neq, the level's results into LDS for the level below and out to memory, a row per quad of lanes, a
// column of the affine each; a wide round sixteen rows per wave, the levels of a narrow batch one behind the other in wave 0 with
// nothing but LDS between them.  One workgroup barrier per batch hands a slot over each way.  What is left is consumer wave 0's
// instruction stream: ~0.4 us a level, narrow or wide (profiles/r06_experiments.md section 5).
struct StripStage {
    float4 local[2][64 * 3];  // From(Transform)
    float4 old[2][64 * 3];    // the GlobalTransform before this frame (the cone's rows: from the snapshot)
    uint32_t pin[2][64];      // bits 0-7 the parent's slot in the level above; bit 8 TransformTreeChanged, bit 9 the level-0 assignment happens, bit 10 mirrored into the snapshot
};
#ifdef MI_EXP_STRIP_STAMPS  // (timing build: how long a wave works between two barriers and how long it waits at them)
#define STRIP_BARRIER()                               \
    do {                                              \
        const unsigned long long tb_ = wall_clock64(); \
        x_work += tb_ - x_last;                       \
        MI_WG_LDS_BARRIER();                          \
        x_last = wall_clock64();                      \
        x_wait += x_last - tb_;                       \
    } while (0)
#else
#define STRIP_BARRIER() MI_WG_LDS_BARRIER()
#endif
template <bool ALL_DIRTY>
__global__ void __launch_bounds__(STRIP_THREADS, 7) k_propagate_strips(Columns c, TreeArgs a, const StripDesc* __restrict__ strips, const StripRound* __restrict__ rounds) {
    __shared__ float4 lds_g[2][STRIP_W_CAP * 3];  // the GlobalTransforms of the level above / of this level
    __shared__ uint8_t lds_chg[2][STRIP_W_CAP];   // "tick bumped" of the same rows
    __shared__ StripStage st;
    // The strip's rounds, copied into LDS first and read from there.  (Read with scalar loads from memory inside the loops they cost
    // 0.2 us a round: scalar loads share the LDS counter and return out of order, so the first wait for any LDS result waited for them.)
    __shared__ uint4 tab[STRIP_TAB_CAP];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t strip = xcd_contiguous_tile();
    unsigned long long ts[3] = {0, 0, 0};
    if (a.trace) ts[0] = wall_clock64();
    uint32_t first, J, NB;
    bool snap_owner;
    {
        typedef const uint32_t __attribute__((address_space(4))) * const_u32;
        const_u32 src = (const_u32)(uintptr_t)(strips + strip);
        first = src[0];
        const uint32_t nr = src[1];
        J = nr & 0xFFFFu;
        NB = (nr >> 16) & 0xFFu;
        snap_owner = (nr >> 31) != 0u;
    }
    for (uint32_t i = threadIdx.x; i < J + 8u && i < STRIP_TAB_CAP; i += STRIP_THREADS) tab[i] = reinterpret_cast<const uint4*>(rounds + first)[i];
    __syncthreads();
    // ---- a frame in which little moved: the strip's flags first (the workgroup tiles' pretest).  Under the static-scene rule the rows
    // the strip owns keep every GlobalTransform and every tick unless one of them is marked (TransformTreeChanged) or assigned as a
    // root / flat row, or the parent of a top row gets a new tick -- which takes a cone row whose own Transform changed (a row that is
    // merely re-evaluated reproduces its value and set_if_neq leaves it alone, systems.rs:719), or a forest root directly above the top
    // rows that is assigned this frame (its tick is bumped whatever the value, systems.rs:522-530).  Strips that mirror rows into the
    // snapshot take the long way.  (Every wave tests, and they agree.)
    if constexpr (!ALL_DIRTY) {
        if ((a.pretest & 1u) && a.static_opt && !snap_owner) {
            bool hot = false;
            for (uint32_t j = 0; j < J; ++j) {
                const uint4 e = tab[j];
                const bool on = lane < (e.z & 0x7Fu);
                const uint32_t row = e.x + (on ? lane : 0u);
                const bool root_level = (e.z & STRIP_ROOT) != 0u;
                const NodeIn in = node_inputs_raw(a, row, root_level, node_raw<false>(a, row, true));
                const bool own_hot = in.tree_changed || in.root_write;
                const bool cone_hot = row_changed(at32<uint8_t>(a.changed, row), a.changed_gen) || ((e.z & STRIP_ABOVE_TOP) != 0u && in.root_write);
                hot = hot || (on && ((e.z & STRIP_OWNED) ? own_hot : cone_hot));
            }
            if (__ballot(hot) == 0ull) {
                if (wv == 0u)
                    for (uint32_t j = 0; j < J; ++j) {
                        const uint4 e = tab[j];
                        if ((e.z & STRIP_OWNED) && lane < (e.z & 0x7Fu)) at32w<uint8_t>(a.g_changed_bytes, e.x + lane) = 0;
                    }
                return;
            }
        }
    }
#ifdef MI_EXP_STRIP_STAMPS
    unsigned long long x_work = 0, x_wait = 0, x_last = wall_clock64();
#endif
    if (a.trace) ts[1] = wall_clock64();
    // a batch's header: the info word of its first entry (uniform)
    auto header = [&](uint32_t i) { return (uint32_t)__builtin_amdgcn_readfirstlane(tab[i].z); };
    auto first_row = [&](uint32_t i) { return (uint32_t)__builtin_amdgcn_readfirstlane(tab[i].x); };
    auto batch_entries = [](uint32_t info) {
        const uint32_t k = (info >> STRIP_BATCH_SHIFT) & 7u;
        return k ? k : 1u;
    };
    // The barriers: B(k) when batch k is staged and batch k - 1 is through.  NB of them on either side (NB, the strip's batches, is even:
    // the planner pads).  (A third slot, the producer two batches ahead and the consumers fetching batch k + 1's staged inputs under
    // batch k's arithmetic, measured no faster and cost a workgroup per CU.)
    if (wv == STRIP_CONSUMERS) {
        // ================= the producer =================
        // A batch's inputs: a row per lane -- lane i row i of a wide round, or row i % 16 of the batch's narrow level i / 16 --, the
        // lane's table entry from LDS; lanes past the rows re-read the entry's first row.  (Scale, translation and rotation word by
        // word, each offset of the rotation through an empty asm statement so that the loads are not merged back: a merged load lands in
        // a register pair / triple / quad which the loop-carried copy of the ring cannot always be allocated on top of -- the moves that
        // remained at the loop's end waited for every load in flight but the last four.)
        // the lane's table entry of the batch that starts at entry i: its row, the first row of the level above, its flags (zero rows
        // for a lane without a row)
        struct Mine {
            uint32_t row, pstart, info;
        };
        auto mine = [&](uint32_t i, uint32_t info0) {
            const uint32_t cnt = batch_entries(info0);
            const bool wide = (info0 & 0x7Fu) > 16u;
            const uint32_t sub_raw = lane >> 4;
            const uint32_t sub = wide ? 0u : (sub_raw < cnt ? sub_raw : cnt - 1u);
            const uint4 ev = tab[i + sub];
            const uint32_t jrow = wide ? lane : (lane & 15u);
            const bool on = jrow < (ev.z & 0x7Fu) && (wide || sub_raw < cnt);
            return Mine{ev.x + (on ? jrow : 0u), ev.y, on ? ev.z : (ev.z & ~0x7Fu)};
        };
        // (what the loop carries is what the loads wrote, nothing derived from them: a value computed from a load would be waited for
        // at the loop's end)
        auto fetch = [&](uint32_t i, uint32_t info0) {
            StripIn f;
            const uint32_t row = mine(i, info0).row;
            const float* const old_src = (info0 & STRIP_OWNED) ? c.global : a.snap_read;  // (uniform: a batch is cone or own, never both)
            const uint32_t r12 = __umul24(row, 12u), r48 = __umul24(row, 48u);  // (rows of a hierarchy that takes strips are below 2^24)
            f.t = V3{at32<float>(c.translation, r12), at32<float>(c.translation, r12 + 4u), at32<float>(c.translation, r12 + 8u)};
            f.s = V3{at32<float>(c.scale, r12), at32<float>(c.scale, r12 + 4u), at32<float>(c.scale, r12 + 8u)};
            {
                uint32_t q0 = row * 16u, q1 = row * 16u + 4u, q2 = row * 16u + 8u, q3 = row * 16u + 12u;
                asm volatile("" : "+v"(q1));
                asm volatile("" : "+v"(q2));
                asm volatile("" : "+v"(q3));
                f.q = make_float4(at32<float>(c.rotation, q0), at32<float>(c.rotation, q1), at32<float>(c.rotation, q2), at32<float>(c.rotation, q3));
            }
            f.par = at32<uint32_t>(a.parent_idx, row * 4u);
            f.o0 = at32<float4>(old_src, r48);
            f.o1 = at32<float4>(old_src, r48 + 16u);
            f.o2 = at32<float4>(old_src, r48 + 32u);
            f.raw = node_raw<ALL_DIRTY>(a, row, true);
            return f;
        };
        auto stage = [&](const StripIn& in, uint32_t i, uint32_t info0, uint32_t sl) {
            const Mine m = mine(i, info0);
            const bool root_level = (m.info & STRIP_ROOT) != 0u;
            lds_put(st.local[sl], lane, affine_from_srt(in.s, V4{in.q.x, in.q.y, in.q.z, in.q.w}, in.t));
            st.old[sl][lane * 3u] = in.o0;
            st.old[sl][lane * 3u + 1u] = in.o1;
            st.old[sl][lane * 3u + 2u] = in.o2;
            uint32_t ps = in.par - m.pstart;  // the parent's slot in the level above (whatever a root reads there is ignored)
            ps = ps < STRIP_W_CAP ? ps : STRIP_W_CAP - 1u;
            const NodeIn nin = node_inputs_raw(a, m.row, root_level, in.raw);
            const bool snap = snap_owner && (in.raw.nflag & 2u) != 0u;  // (node_flags bit 1: some strip's cone holds the row -- its owner mirrors it into the snapshot)
            st.pin[sl][lane] = ps | (nin.tree_changed ? 0x100u : 0u) | (nin.root_write ? 0x200u : 0u) | (snap ? 0x400u : 0u);
        };
        // the next batch to ask for: its first entry and header (kept inside the table: J + 8 entries are there)
        uint32_t pf = 0, hf = header(0);
        auto advance = [&]() {
            pf += batch_entries(hf);
            pf = pf < J + 4u ? pf : J + 4u;
            hf = header(pf);
        };
        uint32_t i0 = pf, h0 = hf;  // the batch whose loads r0 holds
        StripIn r0 = fetch(pf, hf);
        advance();
        uint32_t i1 = pf, h1 = hf;
        StripIn r1 = fetch(pf, hf);
        advance();
        for (uint32_t k = 0; k < NB; k += 2u) {
            stage(r0, i0, h0, 0u);
            __builtin_amdgcn_sched_barrier(0);
            i0 = pf, h0 = hf;
            r0 = fetch(pf, hf);  // (behind the staging that used them up: the loads land in the registers the loop carries)
            advance();
            STRIP_BARRIER();  // B(k)
            stage(r1, i1, h1, 1u);
            __builtin_amdgcn_sched_barrier(0);
            i1 = pf, h1 = hf;
            r1 = fetch(pf, hf);
            advance();
            STRIP_BARRIER();  // B(k + 1)
        }
#ifdef MI_EXP_STRIP_STAMPS
        if (a.trace && lane == 0) {
            a.trace[(size_t)strip * 8u + 5u] = x_wait;
            a.trace[(size_t)strip * 8u + 6u] = x_work;
        }
#endif
        return;
    }
    // ================= the consumers =================
    const uint32_t q_unit = lane >> 2, q_cc = lane & 3u;
    struct Staged {
        uint32_t pin;
        V3 local_c, old_c;
    };
    // the staged inputs of the lane's row: `sidx` its place in the slot
    // (24-bit multiplies for every offset: rows, slots and their byte offsets are far below 2^24, and a full 32-bit or 64-bit multiply-add
    // issues at a quarter of the rate -- five of them were a tenth of a level's step)
    const uint32_t cc3 = q_cc * 3u;
    auto staged = [&](uint32_t sl, uint32_t sidx) {
        Staged r;
        r.pin = st.pin[sl][sidx];
        const uint32_t o = __umul24(sidx, 12u) + cc3;
        const float* const lp = reinterpret_cast<const float*>(st.local[sl]);
        const float* const op = reinterpret_cast<const float*>(st.old[sl]);
        r.local_c = V3{lp[o], lp[o + 1u], lp[o + 2u]};
        r.old_c = V3{op[o], op[o + 1u], op[o + 2u]};
        return r;
    };
    // where the lane's row of a batch's FIRST round sits in the slot (a wide round: sixteen rows per consumer wave)
    auto first_sidx = [&](uint32_t info0) {
        const bool wide = (info0 & 0x7Fu) > 16u;
        const uint32_t j = (wide ? wv * 16u : 0u) + q_unit;
        return j < (info0 & 0x7Fu) ? j : 0u;
    };
    // one level's rows: j = the lane's row inside the round
    auto level_step = [&](uint32_t row0, uint32_t info, uint32_t j, const Staged& in) {
        const bool on = j < (info & 0x7Fu);
        const uint32_t jj = on ? j : 0u;
        const uint32_t row = row0 + jj;
        const bool root_level = (info & STRIP_ROOT) != 0u;
        const bool owned = (info & STRIP_OWNED) != 0u;
        const uint32_t p = (info & STRIP_PARITY) ? 1u : 0u;
        const uint32_t slot = ((info >> 8) & 0xFFu) + jj;
        const uint32_t ps = in.pin & 0xFFu;
        const Affine gp = lds_affine(lds_g[p ^ 1u], ps);
        const uint32_t pcv = lds_chg[p ^ 1u][ps];
        __builtin_amdgcn_sched_barrier(0);  // (the parent's tick is asked for WITH the parent, not where it is used: a round trip less on the chain)
        const bool pc = pcv != 0u;
        V3 cur_c;
        const bool chg = quad_node_apply(on, root_level, a.static_opt != 0, in.pin >> 8, gp, pc, in.local_c, in.old_c, q_cc, lane, &cur_c);
        if (on) {
            float* const gw = reinterpret_cast<float*>(lds_g[p]) + __umul24(slot, 12u) + cc3;
            gw[0] = cur_c.x, gw[1] = cur_c.y, gw[2] = cur_c.z;
            if (q_cc == 0u) lds_chg[p][slot] = chg ? 1 : 0;
            if (owned) {
                const uint32_t goff = __umul24(row, 48u) + cc3 * 4u;
                if (q_cc == 0u) at32w<uint8_t>(a.g_changed_bytes, row) = chg ? 1 : 0;
                if (chg) at32w<F3>(c.global, goff) = F3{cur_c.x, cur_c.y, cur_c.z};
                if (in.pin & 0x400u) at32w<F3>(a.snap_write, goff) = F3{cur_c.x, cur_c.y, cur_c.z};
            }
        }
    };
    auto batch = [&](uint32_t i, uint32_t info0, uint32_t row0, uint32_t sl) {
        const bool wide = (info0 & 0x7Fu) > 16u;
        if (wide) {
            level_step(row0, info0, wv * 16u + q_unit, staged(sl, first_sidx(info0)));
            return;
        }
        if (wv != 0u) return;  // (the levels of a narrow batch hang on each other: one wave, nothing but LDS between two of them)
        const uint32_t cnt = batch_entries(info0);
        // a level's staged inputs are asked for before the level above is multiplied: they do not hang on anything
        Staged cur = staged(sl, first_sidx(info0));
        uint32_t info = info0, r0w = row0;
        for (uint32_t s2 = 1; s2 <= cnt; ++s2) {
            const uint4 ev = tab[i + (s2 < cnt ? s2 : 0u)];
            const uint32_t info_n = __builtin_amdgcn_readfirstlane(ev.z), row_n = __builtin_amdgcn_readfirstlane(ev.x);
            const Staged nxt = staged(sl, 16u * (s2 < cnt ? s2 : 0u) + (q_unit < (info_n & 0x7Fu) ? q_unit : 0u));
            level_step(r0w, info, q_unit, cur);
            // (the level below reads what this one wrote: LDS executes one wave's operations in order, so the read needs no wait for the
            // write -- only the compiler must keep their order)
            __builtin_amdgcn_wave_barrier();
            cur = nxt, info = info_n, r0w = row_n;
        }
    };
    {
        uint32_t pc_i = 0, hc = header(0), rc0 = first_row(0);  // (the next batch's header and first row are read before the barrier that opens it)
        for (uint32_t k = 0; k < NB; k += 2u) {
            STRIP_BARRIER();  // B(k)
            batch(pc_i, hc, rc0, 0u);
            pc_i += batch_entries(hc);
            pc_i = pc_i < J + 4u ? pc_i : J + 4u;
            hc = header(pc_i), rc0 = first_row(pc_i);
            STRIP_BARRIER();  // B(k + 1)
            batch(pc_i, hc, rc0, 1u);
            pc_i += batch_entries(hc);
            pc_i = pc_i < J + 4u ? pc_i : J + 4u;
            hc = header(pc_i), rc0 = first_row(pc_i);
        }
    }
    if (a.trace && wv == 0u) {
        ts[2] = wall_clock64();
        __builtin_amdgcn_s_waitcnt(0);  // stores drained
        const unsigned long long t7 = wall_clock64();
        if (lane == 0) {  // the eight stamps of the workgroup tiles' trace: start, flags tested, -, -, rounds done, -, -, drained
            unsigned long long* const o = a.trace + (size_t)strip * 8u;
            o[0] = ts[0];
            o[1] = ts[1];
#ifdef MI_EXP_STRIP_STAMPS
            o[2] = x_wait;
            o[3] = x_work;
            o[4] = ts[2];
#else
            o[2] = ts[1];
            o[3] = ts[1];
            o[4] = ts[2];
            o[5] = ts[2];
            o[6] = ts[2];
#endif
            o[7] = t7;
        }
    }
}

#undef STRIP_BARRIER
// ---------------------------------------------------------------------------------------------
// visibility_propagate_system + propagate_recursive (crates/bevy_camera/src/visibility/mod.rs:638-729) as the
// fixpoint they maintain, swept over the same subtree tiles as the transforms:
//   Visible -> true, Hidden -> false, Inherited -> parent's InheritedVisibility (true without a parent or when
//   the parent lacks the visibility components, mod.rs:656-659).  InheritedVisibility is bit0 of flags[]; it is
//   assigned (and the change byte set) only where the value differs (mod.rs:667-669,717-718).
// LDS holds one byte per upper-level row: 0/1 = InheritedVisibility, 2 = "no components" (children fall back to true).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t inherit_rule(uint32_t vis, uint32_t parent_state /*0,1,2; 2 = none*/) {
    return vis == 2u ? 1u : vis == 1u ? 0u : (parent_state == 0u ? 0u : 1u);
}
// returns the state children see: 2 when the row has no components, else its (possibly updated) InheritedVisibility
__device__ __forceinline__ uint32_t inherit_update(uint32_t row, uint32_t vis, uint32_t parent_state, uint8_t* flags,
                                                   uint8_t* inh_changed) {
    if (vis & 0x80u) {
        inh_changed[row] = 0;
        return 2u;
    }
    const uint32_t fl = flags[row];
    const uint32_t nv = inherit_rule(vis, parent_state);
    const bool chg = (fl & 1u) != nv;
    if (chg) flags[row] = (uint8_t)((fl & ~1u) | nv);
    inh_changed[row] = chg ? 1 : 0;
    return nv;
}

__global__ void __launch_bounds__(256) k_inherit_flat(uint32_t n, const uint8_t* __restrict__ visibility, uint8_t* flags,
                                                       uint8_t* inh_changed) {
    const uint32_t row = blockIdx.x * 256u + threadIdx.x;
    if (row < n) inherit_update(row, visibility[row], 2u, flags, inh_changed);  // no parent -> Inherited means visible
}

__global__ void __launch_bounds__(256) k_inherit_level(const uint32_t* __restrict__ parent_idx, uint32_t start, uint32_t count,
                                                        const uint8_t* __restrict__ visibility, uint8_t* flags, uint8_t* inh_changed) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    const uint32_t row = start + i, p = parent_idx[row];
    const uint32_t ps = (visibility[p] & 0x80u) ? 2u : (uint32_t)(flags[p] & 1u);
    inherit_update(row, visibility[row], ps, flags, inh_changed);
}

template <bool ROOTS>
__global__ void __launch_bounds__(256) k_inherit_tiles(const uint32_t* __restrict__ parent_idx,
                                                        const TileDesc* __restrict__ tiles,
                                                        const uint8_t* __restrict__ visibility, uint8_t* flags,
                                                        uint8_t* inh_changed) {
    __shared__ uint8_t lds_state[TILE_INHERIT_UCAP];
    const TileDesc td = load_tile_desc(tiles + blockIdx.x);
    const uint32_t L = td.n_levels;
    const uint32_t tid = threadIdx.x;
    uint32_t ubase[TILE_MAX_LEVELS + 1];
    ubase[0] = 0;
    uint32_t n_lds = 0;
#pragma unroll
    for (uint32_t l = 0; l < TILE_MAX_LEVELS; ++l) {
        const uint32_t cnt = l < L ? td.count[l] : 0u;
        ubase[l + 1] = ubase[l] + cnt;
        if (l + 1 < L && n_lds == l && ubase[l + 1] <= TILE_INHERIT_UCAP) n_lds = l + 1;
    }
    // parent state of a row whose parent is NOT in LDS (first level of a non-root tile, or the LDS fallback)
    auto parent_state_global = [&](uint32_t p) -> uint32_t {
        return (visibility[p] & 0x80u) ? 2u : (uint32_t)(flags[p] & 1u);
    };
    for (uint32_t l = 0; l < L; ++l) {
        uint32_t start = td.start[0], count = td.count[0], lbase = 0, pbase = 0, pstart = 0;
#pragma unroll
        for (uint32_t j = 1; j < TILE_MAX_LEVELS; ++j)
            if (j == l) {
                start = td.start[j];
                count = td.count[j];
                lbase = ubase[j];
                pbase = ubase[j - 1];
                pstart = td.start[j - 1];
            }
        const bool to_lds = l < n_lds;
        const bool parents_in_lds = l > 0 && l - 1 < n_lds;
        for (uint32_t i = tid; i < count; i += 256u) {
            const uint32_t row = start + i;
            uint32_t ps = 2u;
            if (!(ROOTS && l == 0)) {
                const uint32_t p = parent_idx[row];
                ps = parents_in_lds ? (uint32_t)lds_state[pbase + (p - pstart)] : parent_state_global(p);
            }
            const uint32_t st = inherit_update(row, visibility[row], ps, flags, inh_changed);
            if (to_lds) lds_state[lbase + i] = (uint8_t)st;
        }
        if (l + 1 < L && !to_lds) {  // fallback: the next level reads this level's flags back from global memory
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        } else {
            __syncthreads();
        }
    }
}

hipError_t launch_inherit_flat(uint32_t n, const uint8_t* visibility, uint8_t* flags, uint8_t* inh_changed,
                               hipStream_t stream) {
    if (n == 0) return hipSuccess;
    MI_LAUNCH(k_inherit_flat, dim3((n + 255u) / 256u), dim3(256), 0, stream, n, visibility, flags, inh_changed);
    return hipGetLastError();
}
hipError_t launch_inherit_tiles(const uint32_t* parent_idx, const TileDesc* d_tiles, uint32_t n_tiles, bool roots,
                                const uint8_t* visibility, uint8_t* flags, uint8_t* inh_changed, hipStream_t stream) {
    if (n_tiles == 0) return hipSuccess;
    if (roots) MI_LAUNCH(k_inherit_tiles<true>, dim3(n_tiles), dim3(256), 0, stream, parent_idx, d_tiles, visibility, flags, inh_changed);
    else MI_LAUNCH(k_inherit_tiles<false>, dim3(n_tiles), dim3(256), 0, stream, parent_idx, d_tiles, visibility, flags, inh_changed);
    return hipGetLastError();
}

hipError_t launch_inherit_level(const uint32_t* parent_idx, uint32_t start, uint32_t count, const uint8_t* visibility, uint8_t* flags,
                                uint8_t* inh_changed, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    MI_LAUNCH(k_inherit_level, dim3((count + 255u) / 256u), dim3(256), 0, stream, parent_idx, start, count, visibility, flags, inh_changed);
    return hipGetLastError();
}
hipError_t launch_propagate_level(const Columns& c, const uint32_t* parent_idx, uint32_t start, uint32_t count, const uint8_t* changed,
                                  const uint8_t* tree_bytes, uint8_t* g_changed_bytes, bool all_dirty, bool static_opt, hipStream_t stream,
                                  const uint8_t* root_node_flags) {
    if (count == 0) return hipSuccess;
    TreeArgs a{};
    a.parent_idx = parent_idx;
    a.node_flags = root_node_flags;
    a.changed = changed;
    a.tree_bytes = tree_bytes;
    a.g_changed_bytes = g_changed_bytes;
    a.all_dirty = all_dirty ? 1u : 0u;
    a.static_opt = static_opt ? 1u : 0u;
    a.pretest = 0;
    a.changed_gen = c.changed_gen;
    if (root_node_flags) MI_LAUNCH(k_propagate_level<true>, dim3((count + 255u) / 256u), dim3(256), 0, stream, c, a, start, count);
    else MI_LAUNCH(k_propagate_level<false>, dim3((count + 255u) / 256u), dim3(256), 0, stream, c, a, start, count);
    return hipGetLastError();
}

hipError_t launch_mark_dirty(uint32_t n, const uint8_t* changed, uint32_t changed_gen, const uint32_t* parent_idx, uint8_t* tree_bytes,
                             uint32_t* clear_words, uint32_t n_clear_words, hipStream_t stream, const uint32_t* anc) {
    if (n == 0) return hipSuccess;
    MI_LAUNCH(k_mark_dirty, dim3((n + 255u) / 256u), dim3(256), 0, stream, n, changed, changed_gen, parent_idx, tree_bytes, clear_words, n_clear_words, anc);
    return hipGetLastError();
}

hipError_t launch_propagate_narrow(const Columns& c, const uint32_t* parent_idx, const uint32_t* level_offsets, uint32_t n_levels,
                                   const uint8_t* node_flags, const uint8_t* changed, const uint8_t* tree_bytes, uint8_t* g_changed_bytes, bool all_dirty,
                                   bool static_opt, bool quad, hipStream_t stream) {
    if (n_levels == 0) return hipSuccess;
    TreeArgs a{};
    a.changed_gen = c.changed_gen;
    a.parent_idx = parent_idx;
    a.node_flags = node_flags;
    a.changed = changed;
    a.tree_bytes = tree_bytes;
    a.g_changed_bytes = g_changed_bytes;
    a.all_dirty = all_dirty ? 1u : 0u;
    a.static_opt = static_opt ? 1u : 0u;
    if (quad) {
        if (all_dirty) MI_LAUNCH((k_propagate_narrow<true, true>), dim3(1), dim3(64), 0, stream, c, a, level_offsets, n_levels);
        else MI_LAUNCH((k_propagate_narrow<false, true>), dim3(1), dim3(64), 0, stream, c, a, level_offsets, n_levels);
    } else {
        if (all_dirty) MI_LAUNCH((k_propagate_narrow<true, false>), dim3(1), dim3(64), 0, stream, c, a, level_offsets, n_levels);
        else MI_LAUNCH((k_propagate_narrow<false, false>), dim3(1), dim3(64), 0, stream, c, a, level_offsets, n_levels);
    }
    return hipGetLastError();
}

hipError_t launch_propagate_wave_tiles(const Columns& c, const uint32_t* parent_idx, const TileDesc* d_wtiles, uint32_t n_tiles, const uint8_t* node_flags,
                                       const uint8_t* changed, const uint8_t* tree_bytes, uint8_t* g_changed_bytes, bool all_dirty, bool static_opt,
                                       bool quad, bool pretest, hipStream_t stream) {
    if (n_tiles == 0) return hipSuccess;
    TreeArgs a{};
    a.pretest = pretest && changed && tree_bytes ? 1u : 0u;
    a.changed_gen = c.changed_gen;
    a.parent_idx = parent_idx;
    a.node_flags = node_flags;
    a.changed = changed;
    a.tree_bytes = tree_bytes;
    a.g_changed_bytes = g_changed_bytes;
    a.all_dirty = all_dirty ? 1u : 0u;
    a.static_opt = static_opt ? 1u : 0u;
    if (quad) {
        if (all_dirty) MI_LAUNCH((k_propagate_wave_tiles<true, true>), dim3(n_tiles), dim3(64), 0, stream, c, a, d_wtiles);
        else MI_LAUNCH((k_propagate_wave_tiles<false, true>), dim3(n_tiles), dim3(64), 0, stream, c, a, d_wtiles);
    } else {
        if (all_dirty) MI_LAUNCH((k_propagate_wave_tiles<true, false>), dim3(n_tiles), dim3(64), 0, stream, c, a, d_wtiles);
        else MI_LAUNCH((k_propagate_wave_tiles<false, false>), dim3(n_tiles), dim3(64), 0, stream, c, a, d_wtiles);
    }
    return hipGetLastError();
}

hipError_t launch_propagate_strips(const Columns& c, const uint32_t* parent_idx, const StripDesc* d_strips, const StripRound* d_rounds, uint32_t n_strips,
                                   const uint8_t* node_flags, const uint8_t* changed, const uint8_t* tree_bytes, uint8_t* g_changed_bytes, const float* snap_read,
                                   float* snap_write, uint32_t snap_rows, bool all_dirty, bool static_opt, bool pretest, hipStream_t stream, unsigned long long* trace) {
    if (n_strips == 0) return hipSuccess;
    TreeArgs a{};
    a.pretest = pretest && changed && tree_bytes ? 1u : 0u;
    a.changed_gen = c.changed_gen;
    a.snap_read = snap_read ? snap_read : c.global;  // (no cone anywhere: nothing reads it)
    a.snap_write = snap_write;
    a.snap_rows = snap_write ? snap_rows : 0u;
    a.parent_idx = parent_idx;
    a.node_flags = node_flags;
    a.changed = changed;
    a.tree_bytes = tree_bytes;
    a.g_changed_bytes = g_changed_bytes;
    a.all_dirty = all_dirty ? 1u : 0u;
    a.static_opt = static_opt ? 1u : 0u;
    a.trace = trace;
    if (all_dirty) MI_LAUNCH((k_propagate_strips<true>), dim3(n_strips), dim3(STRIP_THREADS), 0, stream, c, a, d_strips, d_rounds);
    else MI_LAUNCH((k_propagate_strips<false>), dim3(n_strips), dim3(STRIP_THREADS), 0, stream, c, a, d_strips, d_rounds);
    return hipGetLastError();
}

hipError_t launch_propagate_tiles(const Columns& c, const uint32_t* parent_idx, const TileDesc* d_tiles, const uint32_t* d_chains,
                                  uint32_t n_tiles, const uint8_t* node_flags, const uint8_t* changed, const uint8_t* tree_bytes,
                                  uint8_t* g_changed_bytes, const float* snap_read, float* snap_write, uint32_t snap_rows, bool all_dirty,
                                  bool static_opt, hipStream_t stream, unsigned long long* trace, bool pretest, const TreeCull* cull, bool deep) {
    if (n_tiles == 0) return hipSuccess;
    TreeArgs a;
    a.pretest = (pretest && changed && tree_bytes ? 1u : 0u) | (c.n >= TREE_NT_MIN_ROWS ? 2u : 0u);
    a.changed_gen = c.changed_gen;
    a.snap_read = snap_read;
    a.snap_write = snap_write;
    a.snap_rows = snap_rows;
    a.parent_idx = parent_idx;
    a.tiles = d_tiles;
    a.node_flags = node_flags;
    a.changed = changed;
    a.tree_bytes = tree_bytes;
    a.g_changed_bytes = g_changed_bytes;
    a.chains = d_chains;
    a.all_dirty = all_dirty ? 1u : 0u;
    a.static_opt = static_opt ? 1u : 0u;
    a.trace = trace;
    if (cull) {
        if (!all_dirty || !c.row_summary) return hipErrorInvalidValue;  // (the host checks: every tile runs)
        if (deep) MI_LAUNCH((k_propagate_fans<true, true, TILE_MAX_LEVELS>), dim3(n_tiles + cull->n_compact), dim3(256), 0, stream, c, a, *cull);
        else MI_LAUNCH((k_propagate_fans<true, true>), dim3(n_tiles + cull->n_compact), dim3(256), 0, stream, c, a, *cull);
    } else if (all_dirty) {
        if (deep) MI_LAUNCH((k_propagate_fans<true, false, TILE_MAX_LEVELS>), dim3(n_tiles), dim3(256), 0, stream, c, a, NoCull{});
        else MI_LAUNCH((k_propagate_fans<true, false>), dim3(n_tiles), dim3(256), 0, stream, c, a, NoCull{});
    } else if (deep) MI_LAUNCH((k_propagate_fans<false, false, TILE_MAX_LEVELS>), dim3(n_tiles), dim3(256), 0, stream, c, a, NoCull{});
    else MI_LAUNCH((k_propagate_fans<false, false>), dim3(n_tiles), dim3(256), 0, stream, c, a, NoCull{});
    return hipGetLastError();
}

}  // namespace mi
