// strip_plan.h -- the host-side planner of the STRIPS (kernels.h, kernels_tree.hip: k_propagate_strips): plain C++ over the hierarchy's
// level offsets and parent indices, no device, no context -- mi_upload_hierarchy calls it (ctx_hierarchy.cpp) and so does the CPU test of
// the plan (mi_debug_plan_strips, host_helpers.cpp; tests/test_strip_plan.py walks the plan on the CPU and compares with the oracle).
//
// The levels are cut into bands bottom-up -- a band grows upwards while its widest level stays within W / 2 rows per row of its first
// level --; inside a band consecutive first-level rows are grouped into a strip while no level of their subtree, and no level of the cone
// of their ancestors, holds more than W rows.  A single row whose subtree is wider than that inside the band keeps the levels that fit;
// the rows below become strips of their own (their cone runs through it).  Every row is OWNED by exactly one strip; a strip's table lists
// the cone of its rows' ancestors first (level 0 downwards, a contiguous row range per level), then the rows it owns.
#pragma once
#include <algorithm>
#include <cstdint>
#include <utility>
#include <vector>

#include "kernels.h"

namespace mi {

struct StripPlan {
    std::vector<StripDesc> strips;
    std::vector<uint32_t> strip_top;  // first own level of each strip
    std::vector<StripRound> rounds;   // every strip's table, then eight padding entries (the producer wave reads ahead)
    uint32_t n_bands = 0;
    uint32_t snap_rows = 0;           // every cone row lies in [0, snap_rows): the rows above the deepest first own level
    std::vector<uint8_t> cone_flags;  // per row: 1 = some strip's cone holds it (its owner mirrors it into the snapshot the cones read)
};

// first_child[row] = first row of the next level whose parent is >= row (n + 1 entries; first_child of the last level's rows = n).
// false: no plan (a cone wider than W, a strip with more table entries than the kernel keeps in LDS, cones out of all proportion).
inline bool plan_strips(uint32_t n, uint32_t n_levels, const uint32_t* level_offsets, const uint32_t* parent_idx, const uint32_t* first_child, uint32_t W,
                        bool narrow_batches, uint32_t max_extra, StripPlan& out) {
    out = StripPlan{};
    if (n == 0 || n_levels == 0 || W == 0 || W > STRIP_W_CAP) return false;
    auto child_begin = [&](uint32_t l, uint32_t row) -> uint32_t {  // first row of level l + 1 whose parent >= row (row in level l, or == its end)
        if (row >= level_offsets[l + 1]) return l + 2 <= n_levels ? level_offsets[l + 2] : n;
        return first_child[row];
    };
    auto level_size = [&](uint32_t lv) -> uint64_t { return level_offsets[lv + 1] - level_offsets[lv]; };
    std::vector<std::pair<uint32_t, uint32_t>> bands;  // [s, e), bottom-up
    for (uint32_t e = n_levels; e > 0;) {
        uint32_t s = e - 1;
        uint64_t widest = level_size(s);
        while (s > 0) {
            const uint64_t w2 = std::max(widest, level_size(s - 1));
            if (w2 > (uint64_t)std::max(1u, W / 2u) * std::max<uint64_t>(1, level_size(s - 1))) break;
            widest = w2;
            --s;
        }
        bands.emplace_back(s, e);
        e = s;
    }
    out.n_bands = (uint32_t)bands.size();
    std::vector<StripDesc>& strips = out.strips;
    std::vector<uint32_t>& strip_top = out.strip_top;
    std::vector<StripRound>& rounds = out.rounds;
    struct Region { uint32_t s, lo, hi, e; };
    std::vector<Region> work;
    bool too_long = false;
    // the levels of [a, b) of level s that fit (every level <= W rows), up to e; 0 = the cone of [a, b) is too wide
    // (`extra`: the rounds beyond one per level -- levels of more than 64 rows -- a strip of several first-level rows may have)
    auto fit_levels = [&](uint32_t s, uint32_t e, uint32_t a, uint32_t b) -> uint32_t {
        uint32_t plo = a, phi = b, extra = 0;
        for (uint32_t l = s; l-- > 0;) {
            const uint32_t nlo = parent_idx[plo], nhi = parent_idx[phi - 1] + 1u;
            if (nhi - nlo > W) return 0;
            extra += (nhi - nlo - 1u) / 64u;
            plo = nlo;
            phi = nhi;
        }
        if (extra > max_extra && b - a > 1u) return 0;
        uint32_t clo = a, chi = b, k = 0;
        while (s + k < e && chi > clo) {
            if (chi - clo > W) return k;
            extra += (chi - clo - 1u) / 64u;
            if (extra > max_extra && b - a > 1u) return k;
            const uint32_t nlo = s + k + 1 < n_levels ? child_begin(s + k, clo) : 0u, nhi = s + k + 1 < n_levels ? child_begin(s + k, chi) : 0u;
            clo = nlo;
            chi = nhi;
            ++k;
        }
        return e - s;  // (all of them: the subtree may end earlier)
    };
    auto emit_level = [&](uint32_t l, uint32_t lo, uint32_t hi, uint32_t pstart, uint32_t bits) {
        for (uint32_t r = 0; lo + 64u * r < hi; ++r) {
            StripRound rd{};
            rd.row0 = lo + 64u * r;
            rd.pstart = pstart;
            const uint32_t cnt = std::min(64u, hi - rd.row0);
            rd.info = cnt | ((64u * r) << 8) | ((l & 1u) ? STRIP_PARITY : 0u) | (l == 0 ? STRIP_ROOT : 0u) | bits;
            rd.level = l;
            rounds.push_back(rd);
        }
    };
    auto emit = [&](uint32_t s, uint32_t a, uint32_t b, uint32_t n_lv) {
        std::vector<std::pair<uint32_t, uint32_t>> cone(s);
        uint32_t plo = a, phi = b;
        for (uint32_t l = s; l-- > 0;) {
            const uint32_t nlo = parent_idx[plo], nhi = parent_idx[phi - 1] + 1u;
            cone[l] = {nlo, nhi};
            plo = nlo;
            phi = nhi;
        }
        StripDesc sd{};
        sd.first_round = (uint32_t)rounds.size();
        uint32_t prev = 0;
        for (uint32_t l = 0; l < s; ++l) {
            emit_level(l, cone[l].first, cone[l].second, prev, l + 1 == s ? STRIP_ABOVE_TOP : 0u);
            prev = cone[l].first;
        }
        uint32_t clo = a, chi = b;
        for (uint32_t k = 0; k < n_lv && chi > clo; ++k) {
            emit_level(s + k, clo, chi, prev, STRIP_OWNED);
            prev = clo;
            const uint32_t nlo = s + k + 1 < n_levels ? child_begin(s + k, clo) : 0u, nhi = s + k + 1 < n_levels ? child_begin(s + k, chi) : 0u;
            clo = nlo;
            chi = nhi;
        }
        // batches: a round of more than 16 rows is a batch of its own; up to four consecutive narrow levels of the same kind
        // (cone / own) share one -- the producer stages them together, one consumer wave walks them back to back
        uint32_t n_batches = 0;
        for (size_t r0 = sd.first_round; r0 < rounds.size();) {
            size_t r1 = r0 + 1;
            if ((rounds[r0].info & 0x7Fu) <= 16u && narrow_batches)
                while (r1 < rounds.size() && r1 - r0 < 4 && (rounds[r1].info & 0x7Fu) <= 16u &&
                       ((rounds[r1].info ^ rounds[r0].info) & STRIP_OWNED) == 0u && ((rounds[r1].info >> 8) & 0xFFu) == 0u)
                    ++r1;
            rounds[r0].info |= (uint32_t)(r1 - r0) << STRIP_BATCH_SHIFT;
            ++n_batches;
            r0 = r1;
        }
        if (n_batches & 1u) {  // (an even number of batches: the kernel's loops turn twice per iteration)
            StripRound pad{};
            pad.info = 1u << STRIP_BATCH_SHIFT;
            rounds.push_back(pad);
            ++n_batches;
        }
        sd.n_rounds = (uint32_t)(rounds.size() - sd.first_round);
        if (sd.n_rounds + 8u > STRIP_TAB_CAP || n_batches > 0xFFu) too_long = true;  // (a hierarchy of more levels than a strip's table holds: the tiles)
        sd.n_rounds |= n_batches << 16;
        strips.push_back(sd);
        strip_top.push_back(s);
    };
    for (auto& bd : bands) work.push_back({bd.first, level_offsets[bd.first], level_offsets[bd.first + 1], bd.second});
    for (size_t wi = 0; wi < work.size(); ++wi) {  // (regions handed down are appended behind)
        const Region rg = work[wi];
        uint32_t a = rg.lo;
        while (a < rg.hi) {
            uint32_t b = a + 1;
            const uint32_t lv = fit_levels(rg.s, rg.e, a, b);  // (>= 1: one row, a cone of one row per level)
            if (lv == 0) return false;
            if (lv < rg.e - rg.s) {
                // the row's own subtree outgrows the width at level s + lv: keep the levels above, hand the rest down
                uint32_t clo = a, chi = b;
                for (uint32_t k = 0; k < lv; ++k) {
                    const uint32_t nlo = child_begin(rg.s + k, clo), nhi = child_begin(rg.s + k, chi);
                    clo = nlo;
                    chi = nhi;
                }
                if (chi > clo) work.push_back({rg.s + lv, clo, chi, rg.e});
            } else {
                uint32_t step = 1;
                while (b < rg.hi) {  // galloping extension while everything still fits
                    const uint32_t nb = (uint32_t)std::min<uint64_t>((uint64_t)b + step, rg.hi);
                    if (fit_levels(rg.s, rg.e, a, nb) == rg.e - rg.s) { b = nb; step *= 2; }
                    else if (step > 1) step = 1;
                    else break;
                }
            }
            emit(rg.s, a, b, lv);
            a = b;
        }
        if (rounds.size() > (size_t)64 * n + (1u << 20)) return false;  // (cones out of all proportion: a hierarchy for the tiles)
    }
    if (too_long || strips.empty()) return false;
    for (uint32_t i = 0; i < 8u; ++i) rounds.push_back(StripRound{});  // (the producer reads up to six entries past a strip's last round)
    uint32_t snap_level = 0;
    for (uint32_t s : strip_top) snap_level = std::max(snap_level, s);
    out.snap_rows = level_offsets[snap_level];
    // Which rows the snapshot has to hold: the cones' rows only -- a few thousand of a lopsided tree's 300 000, although a handed-down
    // strip deep in the tree puts most rows above the deepest first own level --, and which strips own one of them.
    out.cone_flags.assign(n, 0);
    for (const StripRound& rd : rounds)
        if (!(rd.info & STRIP_OWNED))
            for (uint32_t k = 0; k < (rd.info & 0x7Fu); ++k) out.cone_flags[rd.row0 + k] = 1;
    for (size_t i = 0; i < strips.size(); ++i) {
        bool owns = false;
        for (uint32_t j = 0; j < (strips[i].n_rounds & 0xFFFFu) && !owns; ++j) {
            const StripRound& rd = rounds[strips[i].first_round + j];
            if (rd.info & STRIP_OWNED)
                for (uint32_t k = 0; k < (rd.info & 0x7Fu) && !owns; ++k) owns = out.cone_flags[rd.row0 + k] != 0;
        }
        if (owns) strips[i].n_rounds |= 0x80000000u;
    }
    return true;
}

}  // namespace mi
