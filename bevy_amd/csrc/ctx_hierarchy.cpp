// ctx_hierarchy.cpp -- mi_upload_hierarchy: validation of the level-ordered parent array and the subtree-tile plan.
#include "ctx.h"
#include "strip_plan.h"

using namespace mi;
using namespace mi_detail;

extern "C" {

int32_t mi_upload_hierarchy(mi_ctx* ctx, uint32_t n, const uint32_t* parent_idx, const uint32_t* level_offsets,
                            uint32_t n_levels) {
    ENTER(ctx);
    if (n != ctx->n) return fail(ctx, MI_ERR_INVALID_ARG, "mi_upload_hierarchy: n (%u) != live rows (%u)", n, ctx->n);
    trs_written(ctx);  // (rows may have been renumbered with it: nothing fetched ahead of a flat frame applies)
    ctx->changed_maybe = true, ++ctx->marks_serial;  // conservative: the next propagate looks at the rows again
    // marks an upload of this frame climbed for belong to the hierarchy that is being replaced
    if (ctx->marks_in_cur) ctx->tree_clean[ctx->tree_parity] = false;
    ctx->marks_live = ctx->marks_in_cur = ctx->marks_complete = ctx->marks_other_cleared = false;
    if (!parent_idx || n_levels <= 1) {
        ctx->have_hierarchy = false;
        ctx->anc_valid = false;
        ctx->n_levels = 1;
        ctx->level_offsets = {0, n};
        ctx->passes.clear();
        ctx->groups.clear();
        ctx->stream_levels.clear();
        ctx->narrow = false;
        ctx->wave_forest = false;
        return MI_OK;
    }
    if (!level_offsets) return fail(ctx, MI_ERR_INVALID_ARG, "mi_upload_hierarchy: level_offsets NULL");
    if (level_offsets[0] != 0 || level_offsets[n_levels] != n)
        return fail(ctx, MI_ERR_MALFORMED_HIERARCHY, "level_offsets must start at 0 and end at n");
    for (uint32_t l = 0; l < n_levels; ++l)
        if (level_offsets[l + 1] < level_offsets[l]) return fail(ctx, MI_ERR_MALFORMED_HIERARCHY, "level_offsets not monotone");
    // level 0: roots; level l>0: parent in level l-1, parents non-decreasing (BFS order)
    for (uint32_t i = level_offsets[0]; i < level_offsets[1]; ++i)
        if (parent_idx[i] != MI_NO_PARENT)
            return fail(ctx, MI_ERR_MALFORMED_HIERARCHY, "row %u in level 0 has a parent", i);
    for (uint32_t l = 1; l < n_levels; ++l) {
        const uint32_t plo = level_offsets[l - 1], phi = level_offsets[l];
        uint32_t prev = plo;
        for (uint32_t i = level_offsets[l]; i < level_offsets[l + 1]; ++i) {
            const uint32_t p = parent_idx[i];
            if (p < plo || p >= phi)
                return fail(ctx, MI_ERR_MALFORMED_HIERARCHY, "row %u (level %u): parent %u is not in level %u", i, l, p, l - 1);
            if (p < prev)
                return fail(ctx, MI_ERR_MALFORMED_HIERARCHY, "row %u: rows of a level must be ordered by parent (use mi_hierarchy_sort)", i);
            prev = p;
        }
    }
    // node flags + first-child table
    std::vector<uint8_t> nflags(n, 0);
    std::vector<uint32_t> first_child((size_t)n + 1, 0);
    for (uint32_t l = 0; l + 1 < n_levels; ++l) {
        uint32_t ch = level_offsets[l + 1];
        const uint32_t chi = level_offsets[l + 2];
        for (uint32_t p = level_offsets[l]; p < level_offsets[l + 1]; ++p) {
            first_child[p] = ch;
            while (ch < chi && parent_idx[ch] == p) { ++ch; nflags[p] |= 1; }
        }
    }
    for (uint32_t p = level_offsets[n_levels - 1]; p <= n; ++p) first_child[p] = n;
    // first_child[level end] of level l must read as "end of level l+1": patch boundaries
    auto child_begin = [&](uint32_t l, uint32_t row) -> uint32_t {
        // first row of level l+1 whose parent >= row (row in level l, or == end of level l)
        if (row >= level_offsets[l + 1]) return level_offsets[l + 2];
        return first_child[row];
    };

    // ---- tile plan ----
    // A tile = a contiguous row range at its first level plus all the descendants of those rows over the next levels,
    // walked by one workgroup (kernels_tree.hip).  It "fits" when all its levels but the last together hold <= TILE_LIGHT_UCAP
    // rows (they live in LDS); the last level is streamed and may be any size (kept <= TILE_LIGHT_LAST_CAP to spread the work).
    // Levels are cut into BANDS of consecutive levels, bottom-up, so that the bottom band -- where nearly all the rows of a
    // tree are -- gets the deepest tiles the LDS budget allows.  Three kinds of band:
    //   roots      first level = level 0 (forest roots and flat rows); a tile may span many roots;
    //   chain      every tile's first-level rows are children of ONE node whose ancestor chain (<= TILE_MAX_CHAIN nodes)
    //              the tile re-evaluates itself -- same products, same order, hence the same bits as the tiles that own
    //              those ancestors write -- so it depends on nothing another tile produces;
    //   dependent  (fallback: chains too long, or tiles too small to be worth a chain each) first-level parents are read
    //              from global memory, so the band needs its own launch behind the band above.
    // Roots and chain bands are mutually independent: they share ONE launch, whatever the depth of the hierarchy.
    std::vector<TileDesc> tiles;
    std::vector<uint32_t> chains;
    {
        // Short-lived tiles, several rounds of them per CU: the head of one tile (descriptor, ancestor chain, LDS levels) runs
        // under the streaming of its neighbours.
        const uint32_t UCAP = TILE_LIGHT_UCAP, LAST_CAP = TILE_LIGHT_LAST_CAP;
        ctx->passes.clear();
        ctx->groups.clear();
        ctx->stream_levels.clear();
        auto level_size = [&](uint32_t lv) -> uint64_t { return lv < n_levels ? level_offsets[lv + 1] - level_offsets[lv] : 0; };
        // A VERY wide deepest level is not tiled but swept as a stream by k_propagate_level, at the flat kernel's pace, in a
        // launch of its own.  That paid from ~1.4 M nodes up while the tile kernels ran at 4 - 6 workgroups per CU; the light
        // tile kernel of today wins at every size and shape measured, so the thresholds (kernels.h) sit above that range.
        uint32_t n_tile_levels = n_levels;
        const uint32_t stream_last = ctx->tile_mode == 3 ? STREAM_LEVEL_MIN_ROWS_LAST_TEST : STREAM_LEVEL_MIN_ROWS_LAST;
        const uint32_t stream_inner = ctx->tile_mode == 3 ? STREAM_LEVEL_MIN_ROWS_TEST : STREAM_LEVEL_MIN_ROWS;
        while (n_tile_levels > 1 && level_size(n_tile_levels - 1) >= (n_tile_levels == n_levels ? stream_last : stream_inner))
            --n_tile_levels;
        for (uint32_t l = n_tile_levels; l < n_levels; ++l) ctx->stream_levels.emplace_back(level_offsets[l], level_offsets[l + 1] - level_offsets[l]);
        struct Band { uint32_t s, e; bool chain; };
        std::vector<Band> bands;  // bottom-up
        // the shallowest first level `cand` >= lo from which tiles down to level e - 1 fit on average
        auto band_start = [&](uint32_t e, uint32_t lo) {
            for (uint32_t cand = lo; cand + 1 < e; ++cand) {
                // per first-level row, on average: the would-be upper levels must fit the LDS budget and the last level the cap
                uint64_t upper = 0;
                for (uint32_t l = cand; l + 1 < e; ++l) upper += level_size(l);
                const uint64_t firsts = std::max<uint64_t>(1, level_size(cand));
                if (upper <= (uint64_t)UCAP * firsts && level_size(e - 1) <= (uint64_t)LAST_CAP * firsts) return cand;
            }
            return e - 1;  // a band of one level always works
        };
        for (uint32_t e = n_tile_levels; e > 0;) {
            // Up to TILE_FAST_LEVELS levels per tile -- what the LDS rows allow a bushy tree anyway.  Deeper tiles (TILE_MAX_LEVELS, the
            // kernel's second instantiation) where they END the cutting: a forest of small trees whose every tree then is (part of) ONE
            // roots tile instead of a roots tile and several chain tiles (4 000 humanoid rigs of 68 nodes and 13 levels: 12 286 -> 4 000
            // tiles, 43 -> 24 us per all-dirty frame), and where the hierarchy is as narrow as a chain (half the dependent launches).  On
            // lopsided deep trees (transform_hierarchy.rs's large_tree / deep_tree) deeper bands measured SLOWER -- more chain tiles
            // with longer chains: 25.6 -> 32.0 us -- so everything else keeps the short tiles.
            uint32_t s = band_start(e, e > TILE_FAST_LEVELS ? e - TILE_FAST_LEVELS : 0);
            if (s > 0 && e > TILE_FAST_LEVELS) {
                const uint32_t deep = band_start(e, e > TILE_MAX_LEVELS ? e - TILE_MAX_LEVELS : 0);
                uint64_t widest = 0;  // of the deeper band's levels: a chain (or a rope of a few strands) if none exceeds a wave
                for (uint32_t l = deep; l < e; ++l) widest = std::max(widest, level_size(l));
                if (deep == 0 || widest <= 64) s = deep;
            }
            uint64_t rows = 0;
            for (uint32_t l = s; l < e; ++l) rows += level_size(l);
            // worth a chain per tile unless the band is a swarm of tiny tiles: (tiles x chain length) node products are spent on
            // chains; keep that within a few times the band's own rows
            const uint64_t est_tiles = std::max<uint64_t>(std::max<uint64_t>(1, level_size(s - (s ? 1 : 0))), rows / (UCAP + LAST_CAP));
            const bool chain = s > 0 && s <= TILE_MAX_CHAIN && est_tiles * s <= 4 * rows + 4096;
            bands.push_back({s, e, chain});
            e = s;
        }
        // One tile: rows [lo, hi) of level s and their descendants over the next n_lv - 1 levels.  true = it fits.
        auto build_tile = [&](uint32_t s, uint32_t n_lv, uint32_t lo, uint32_t hi, TileDesc& td) -> bool {
            td = TileDesc{};
            uint32_t clo = lo, chi2 = hi;
            for (uint32_t k = 0; k < n_lv; ++k) {
                td.start[k] = clo;
                td.count[k] = chi2 - clo;
                if (chi2 > clo) td.n_levels = k + 1;
                if (k + 1 < n_lv) {
                    const uint32_t nlo = child_begin(s + k, clo), nhi = child_begin(s + k, chi2);
                    clo = nlo;
                    chi2 = nhi;
                }
            }
            uint64_t up = 0;
            for (uint32_t k = 0; k + 1 < td.n_levels; ++k) up += td.count[k];
            return up <= UCAP && (td.n_levels == 0 || td.count[td.n_levels - 1] <= LAST_CAP);
        };
        auto upper_rows = [](const TileDesc& td) {
            uint64_t up = 0;
            for (uint32_t k = 0; k + 1 < td.n_levels; ++k) up += td.count[k];
            return up;
        };
        // What a tile of ONE first-level row could not take: a node with hundreds of children that have children of their own
        // overflows the tile kernel's LDS rows.  Its tile is cut off after the last level that fits, and the rows below -- [lo, hi) of
        // `level`, down to the band's end -- become tiles of a launch of their own behind the one that holds their parents (first-level
        // parents read from global memory), cut again if need be.  (Rounds 1 - 3 sent such hierarchies to a second tile kernel, round 4
        // first to the level sweep; this keeps them on the one tile kernel.)
        struct Region { uint32_t level, lo, hi, e; };
        // rows [rlo, rhi) of level s as tiles over levels [s, e): galloping extension of the first-level range while the tile still fits
        auto cut_rows = [&](uint32_t s, uint32_t e, uint32_t rlo, uint32_t rhi, bool chain, uint32_t kind_base, std::vector<Region>& spill) {
            const uint32_t d = e - s;
            uint32_t a = rlo;
            while (a < rhi) {
                TileDesc best{};
                uint32_t b = a + 1;
                const bool fits = build_tile(s, d, a, b, best);
                // (a last level ONE workgroup would stream for milliseconds -- a node whose few children have 100 000 children -- is not kept
                // either: it is handed down and cut into tiles of its own)
                const bool too_wide = best.n_levels > 1 && best.count[best.n_levels - 1] > 16u * LAST_CAP;
                if (!fits && (upper_rows(best) > UCAP || too_wide)) {
                    // the row's own subtree does not fit: keep the levels that do, hand the rest down
                    uint32_t k = d;
                    while (k > 1 && (build_tile(s, k, a, b, best), upper_rows(best) > UCAP)) --k;
                    build_tile(s, k, a, b, best);
                    if (best.n_levels < k) k = best.n_levels;
                    while (k > 1 && best.count[k - 1] > 16u * LAST_CAP) {
                        --k;
                        build_tile(s, k, a, b, best);
                    }
                    if (best.n_levels == k && k < d) {
                        const uint32_t llo = best.start[k - 1], lhi = llo + best.count[k - 1];
                        const uint32_t clo = child_begin(s + k - 1, llo), chi = child_begin(s + k - 1, lhi);
                        if (chi > clo) spill.push_back({s + k, clo, chi, e});
                    }
                } else {
                    uint32_t step = 1;
                    while (b < rhi) {
                        const uint32_t nb = (uint32_t)std::min<uint64_t>((uint64_t)b + step, rhi);
                        TileDesc cand{};
                        const bool same_parent = !chain || parent_idx[nb - 1] == parent_idx[a];
                        if (same_parent && build_tile(s, d, a, nb, cand)) { best = cand; b = nb; step *= 2; }
                        else if (step > 1) step = 1;
                        else break;
                    }
                }
                if (best.n_levels) {
                    best.kind = kind_base;
                    chains.resize((tiles.size() + 1) * (size_t)TILE_MAX_CHAIN, 0u);
                    if (chain) {
                        uint32_t row = parent_idx[best.start[0]], len = 0;
                        while (row != MI_NO_PARENT && len < TILE_MAX_CHAIN) {
                            chains[tiles.size() * (size_t)TILE_MAX_CHAIN + len++] = row;
                            row = parent_idx[row];
                        }
                        best.kind = len;
                    }
                    tiles.push_back(best);
                }
                a = b;
            }
        };
        std::vector<std::vector<Region>> band_spill(bands.size());
        auto build_band = [&](size_t i, uint32_t kind_base) {
            const Band& bd = bands[i];
            const uint32_t first_tile = (uint32_t)tiles.size();
            cut_rows(bd.s, bd.e, level_offsets[bd.s], level_offsets[bd.s + 1], bd.chain, kind_base, band_spill[i]);
            return std::make_pair(first_tile, (uint32_t)tiles.size() - first_tile);
        };
        // launch 1: the chain bands, bottom band first (the longest tiles start first), then the roots band
        std::vector<std::pair<uint32_t, uint32_t>> band_tiles(bands.size());
        uint32_t n_chain_tiles = 0, owner_rows = 0;
        for (size_t i = 0; i < bands.size(); ++i)
            if (bands[i].chain) {
                band_tiles[i] = build_band(i, 0u);
                n_chain_tiles += band_tiles[i].second;
                owner_rows = std::max(owner_rows, level_offsets[bands[i].s]);  // the snapshot prefix: every row above the deepest chain band
            }
        band_tiles.back() = build_band(bands.size() - 1, TILE_ROOTS);  // bands.back() starts at level 0
        ctx->groups.push_back({0u, (uint32_t)tiles.size(), n_chain_tiles, owner_rows});
        // then, top-down: a dependent band's own launch, and behind any band the launches of what its tiles had to hand down (one per
        // depth of cutting: a handed-down region reads the rows of the launch that cut it)
        std::vector<std::vector<std::pair<uint32_t, uint32_t>>> band_passes(bands.size());
        for (size_t i = bands.size(); i-- > 0;) {
            if (!bands[i].chain && bands[i].s > 0) {
                band_tiles[i] = build_band(i, 0u);
                ctx->groups.push_back({band_tiles[i].first, band_tiles[i].second, 0u, 0u});
            }
            band_passes[i].push_back(band_tiles[i]);
            std::vector<Region> cur;
            cur.swap(band_spill[i]);
            while (!cur.empty()) {
                std::vector<Region> next;
                const uint32_t first_tile = (uint32_t)tiles.size();
                for (const Region& r : cur) cut_rows(r.level, r.e, r.lo, r.hi, false, 0u, next);
                const uint32_t cnt = (uint32_t)tiles.size() - first_tile;
                if (cnt) {
                    ctx->groups.push_back({first_tile, cnt, 0u, 0u});
                    band_passes[i].emplace_back(first_tile, cnt);
                }
                cur.swap(next);
            }
        }
        chains.resize(tiles.size() * (size_t)TILE_MAX_CHAIN, 0u);
        // the bands top-down, for the kernels that sweep level by level with one launch per band (InheritedVisibility): a band's own
        // tiles, then what they handed down, depth by depth
        for (size_t i = bands.size(); i-- > 0;)
            for (auto& ps : band_passes[i]) ctx->passes.push_back(ps);
    }
    // The tile kernel addresses rows with 32-bit byte offsets: a hierarchy beyond 89 M rows is swept level by level, one streaming
    // launch per level (mi_propagate; mi_debug_set_tile_mode(1) forces it).  The tile list stays: the InheritedVisibility sweep walks it.
    for (auto& gr : ctx->groups) {  // which instantiation of the tile kernel a launch takes (kernels.h, TILE_FAST_LEVELS)
        gr.deep = false;
        for (uint32_t t = gr.first; t < gr.first + gr.count; ++t) gr.deep = gr.deep || tiles[t].n_levels > TILE_FAST_LEVELS;
    }
    ctx->by_levels = ctx->tile_mode == 1 || n > 0xFFFFFFFFu / 48u;
    for (const TileDesc& td : tiles) {  // (every tile fits by construction: a subtree that would not was cut, above)
        uint64_t up = 0;
        for (uint32_t k = 0; k + 1 < td.n_levels; ++k) up += td.count[k];
        if (up > TILE_LIGHT_UCAP) ctx->by_levels = true;
    }
    if (ctx->by_levels)
        for (auto& gr : ctx->groups) gr.n_chain = gr.owner_rows = 0;  // (no chain tile runs: no snapshot to keep)
    int32_t rc;
    if ((rc = ensure(ctx, ctx->parent_idx, (size_t)n * 4))) return rc;
    if ((rc = ensure(ctx, ctx->node_flags, n))) return rc;
    if ((rc = ensure(ctx, ctx->tiles, std::max<size_t>(tiles.size(), 1) * sizeof(TileDesc)))) return rc;
    if ((rc = upload(ctx, ctx->parent_idx.p, parent_idx, (size_t)n * 4))) return rc;
    if ((rc = upload(ctx, ctx->node_flags.p, nflags.data(), n))) return rc;
    if ((rc = upload(ctx, ctx->tiles.p, tiles.data(), tiles.size() * sizeof(TileDesc)))) return rc;
    ctx->snap_rows = 0;
    for (auto& gr : ctx->groups) ctx->snap_rows = std::max(ctx->snap_rows, gr.owner_rows);
    ctx->snap_valid = false;
    if (ctx->snap_rows && (rc = ensure(ctx, ctx->snap, 2 * (size_t)ctx->snap_rows * 48))) return rc;
    if ((rc = ensure(ctx, ctx->chains, std::max<size_t>(chains.size(), 1) * 4))) return rc;
    if ((rc = upload(ctx, ctx->chains.p, chains.data(), chains.size() * 4))) return rc;
    ctx->level_offsets.assign(level_offsets, level_offsets + n_levels + 1);
    ctx->n_levels = n_levels;
    // as narrow as a chain, and deeper than one tile: the whole hierarchy is one wave's walk (kernels_tree.hip, k_propagate_narrow)
    ctx->narrow = n_levels > TILE_MAX_LEVELS && ctx->tile_mode != 1;
    ctx->narrow_quad = true;
    for (uint32_t l = 0; l < n_levels && ctx->narrow; ++l) {
        const uint32_t w = level_offsets[l + 1] - level_offsets[l];
        ctx->narrow = w >= 1u && w <= 64u;  // (an empty level -- mi_upload_hierarchy accepts them -- sends the hierarchy to the tiles: the one-wave kernel counts on increasing level ends)
        ctx->narrow_quad = ctx->narrow_quad && w <= 16u;
    }
    if (ctx->narrow) {
        if ((rc = ensure(ctx, ctx->level_offs_dev, ((size_t)n_levels + 1) * 4))) return rc;
        if ((rc = upload(ctx, ctx->level_offs_dev.p, level_offsets, ((size_t)n_levels + 1) * 4))) return rc;
    }
    // A forest of small trees (humanoid rigs: transform_hierarchy.rs:493-561; most of a game's scene): when EVERY root's tree fits a
    // wave tile -- <= TILE_MAX_LEVELS levels, <= WAVE_TILE_ROWS rows, no level wider than a wave -- a wave per tile walks it
    // (k_propagate_wave_tiles): consecutive roots are packed into one tile while the limits hold, first with at most 16 rows to a
    // level (a node per quad of lanes), else with at most 64.  Deep enough for the serial level steps to be what a tile costs
    // (>= 5 levels): a shallow forest is bandwidth, which the workgroup tiles are built for.  mi_debug_set_tile_mode(4) plans without.
    ctx->wave_forest = false;
    ctx->n_wtiles = 0;
    if (n_levels >= 5 && n_levels <= TILE_MAX_LEVELS && !ctx->narrow && (ctx->tile_mode == 0 || ctx->tile_mode == 2 || ctx->tile_mode == 3)) {
        auto wave_tile = [&](uint32_t lo, uint32_t hi, uint32_t max_w, TileDesc& td) -> bool {
            td = TileDesc{};
            uint32_t clo = lo, chi = hi, total = 0;
            for (uint32_t k = 0; k < n_levels && chi > clo; ++k) {
                td.start[k] = clo;
                td.count[k] = chi - clo;
                td.n_levels = k + 1;
                total += chi - clo;
                if (chi - clo > max_w || total > WAVE_TILE_ROWS) return false;
                if (k + 1 < n_levels) {
                    const uint32_t nlo = child_begin(k, clo), nhi = child_begin(k, chi);
                    clo = nlo;
                    chi = nhi;
                } else {
                    clo = chi;
                }
            }
            td.kind = TILE_ROOTS;
            return true;
        };
        const uint32_t n_roots = level_offsets[1];
        for (uint32_t max_w : {16u, 64u}) {
            std::vector<TileDesc> wt;
            bool ok = true;
            uint32_t a0 = 0;
            while (a0 < n_roots && ok) {
                TileDesc best{};
                uint32_t b = a0 + 1;
                ok = wave_tile(a0, b, max_w, best);
                if (!ok) break;
                uint32_t step = 1;
                while (b < n_roots) {  // galloping extension of the root range while the tile still fits
                    const uint32_t nb = (uint32_t)std::min<uint64_t>((uint64_t)b + step, n_roots);
                    TileDesc cand{};
                    if (wave_tile(a0, nb, max_w, cand)) { best = cand; b = nb; step *= 2; }
                    else if (step > 1) step = 1;
                    else break;
                }
                wt.push_back(best);
                a0 = b;
            }
            if (ok && !wt.empty()) {
                if ((rc = ensure(ctx, ctx->wtiles, wt.size() * sizeof(TileDesc)))) return rc;
                if ((rc = upload(ctx, ctx->wtiles.p, wt.data(), wt.size() * sizeof(TileDesc)))) return rc;
                ctx->wave_forest = true;
                ctx->wave_quad = max_w == 16u;
                ctx->n_wtiles = (uint32_t)wt.size();
                break;
            }
        }
    }
    // STRIPS (kernels.h, k_propagate_strips): a hierarchy of >= 4 levels that is neither a forest of wave tiles nor as narrow as a chain
    // takes ONE launch of independent waves, whatever its depth.  The levels are cut into bands bottom-up -- a band grows upwards while
    // its widest level stays within W / 2 rows per row of its first level --; inside a band consecutive first-level rows are grouped
    // into a strip while no level of their subtree, and no level of the cone of their ancestors, holds more than W rows.  A single
    // row whose subtree is wider than that inside the band keeps the levels that fit; the rows below become strips of their own (their
    // cone runs through it).  mi_debug_set_tile_mode(4) plans without, (5) takes strips wherever they can be planned.
    ctx->strip_plan = false;
    ctx->n_strips = 0;
    {
        // The width: 64 rows to a level (one round per level) while every strip of the launch is in flight at once -- five workgroups of
        // five waves per CU --, else 128: a second round at the widest levels costs a strip less than waiting for a free CU does
        // (deep_tree, kernel us of an all-dirty frame: 1 839 strips of 64 29.8, 1 084 of 128 22.9).  MI_STRIP_W: the test knob.
        const uint32_t resident = 5u * 256u;
        std::vector<uint32_t> widths = {64u, 128u};
        if (const char* ev = getenv("MI_STRIP_W")) widths = {std::min((uint32_t)std::max(1, atoi(ev)), STRIP_W_CAP)};
        const bool modes_ok = ctx->tile_mode == 0 || ctx->tile_mode == 2 || ctx->tile_mode == 3;
        // by default: where the tiles need dependent launches (a lopsided tree), or a deep hierarchy has more tiles than the chip holds at
        // once (2 048: the launch then runs two generations of tiles' chains).  Measured, all-dirty frame, kernel us: large_tree 36.3 ->
        // 17.8, deep_tree 26.7 -> 22.9, update_leaves (18 levels, 2 065 tiles) 19.4 -> 16.0; the other way: a full binary tree of 16
        // levels (517 tiles) 8.9 -> 10.1 us per frame, a root with 500 x 500 descendants 10.2 -> 13.5, the 1.4 M-node 4-ary tree 32.9 ->
        // 67.9 (tools/probes/mid_shapes_probe.py, profiles/r06b/).
        const bool wanted = ctx->tile_mode == 5 || (modes_ok && !ctx->wave_forest && !ctx->narrow && !ctx->by_levels && n <= STRIP_MAX_ROWS &&
                                                    (ctx->groups.size() >= 2 || (n_levels >= 16 && !ctx->groups.empty() && ctx->groups.front().count > 2048u)));
        for (size_t wi_ = 0; wi_ < widths.size() && wanted && !ctx->by_levels && n > 0 && n < (1u << 24) && !ctx->strip_plan; ++wi_) {  // (the kernel's 24-bit row offsets)
            const bool last_width = wi_ + 1 == widths.size();
            StripPlan sp;
            if (!plan_strips(n, n_levels, level_offsets, parent_idx, first_child.data(), widths[wi_], true, 1000u, sp)) continue;
            if (!last_width && sp.strips.size() > resident) continue;
            if ((rc = ensure(ctx, ctx->strips, sp.strips.size() * sizeof(StripDesc)))) return rc;
            if ((rc = upload(ctx, ctx->strips.p, sp.strips.data(), sp.strips.size() * sizeof(StripDesc)))) return rc;
            if ((rc = ensure(ctx, ctx->strip_rounds, sp.rounds.size() * sizeof(StripRound)))) return rc;
            if ((rc = upload(ctx, ctx->strip_rounds.p, sp.rounds.data(), sp.rounds.size() * sizeof(StripRound)))) return rc;
            // node_flags bit 1: some strip's cone holds the row (its owner mirrors it into the snapshot the cones read)
            for (uint32_t r = 0; r < n; ++r) nflags[r] = (uint8_t)((nflags[r] & 1u) | (sp.cone_flags[r] ? 2u : 0u));
            if ((rc = upload(ctx, ctx->node_flags.p, nflags.data(), n))) return rc;
            ctx->strip_plan = true;
            if (ctx->tile_mode == 5) ctx->narrow = ctx->wave_forest = false;  // (the test mode: strips whatever else would take the hierarchy)
            ctx->n_strips = (uint32_t)sp.strips.size();
            ctx->n_strip_rounds = (uint32_t)sp.rounds.size();
            ctx->strip_bands = sp.n_bands;
            // the snapshot the cones read: the strips' own prefix (the tile plan's chain tiles do not run)
            ctx->snap_rows = sp.snap_rows;
            ctx->snap_valid = false;
            if (ctx->snap_rows && (rc = ensure(ctx, ctx->snap, 2 * (size_t)ctx->snap_rows * 48))) return rc;
        }
    }
    ctx->have_hierarchy = true;
    // the ancestor table for mark_dirty_trees (kernels.h): level by level on the device, behind the parent_idx upload
    ctx->anc_valid = false;
    if (ensure(ctx, ctx->anc, (size_t)n * ANC_DEPTH * 4) == MI_OK) {
        hipError_t e = hipSuccess;
        for (uint32_t l = 0; l < n_levels && e == hipSuccess; ++l)
            e = launch_build_ancestors((const uint32_t*)ctx->parent_idx.p, level_offsets[l], level_offsets[l + 1] - level_offsets[l], (uint32_t*)ctx->anc.p, ctx->stream);
        ctx->anc_valid = e == hipSuccess;
    }
    return MI_OK;  // (without the table the marks climb along parent_idx)
}

// test / bench hook (not part of the public header): which tile kernel the NEXT mi_upload_hierarchy plans for
// (0 = tiles where they fit, 1 = level by level whatever the shape, 2 = as 0 (the light tiles of earlier rounds), 3 = as 0 with the
// streamed-level thresholds at their test values, 4 = as 0 without the wave tiles of a forest of small trees and without strips,
// 5 = strips wherever they can be planned)
int32_t mi_debug_set_tile_mode(mi_ctx* ctx, int32_t mode) {
    ENTER(ctx);
    if (mode < 0 || mode > 5) return fail(ctx, MI_ERR_INVALID_ARG, "mi_debug_set_tile_mode: mode %d", mode);
    ctx->tile_mode = mode;
    return MI_OK;
}

// development hook: per-tile phase timestamps of the light tile kernel (8 x s_memrealtime, 100 MHz, per tile of the first
// launch).  enable allocates the buffer (mi_propagate then fills it every frame); out != NULL copies it out; enable = 0 with
// out = NULL switches the stamps off again.
int32_t mi_debug_tree_trace(mi_ctx* ctx, int32_t enable, unsigned long long* out, uint32_t n_tiles) {
    ENTER(ctx);
    if (enable) {
        size_t tiles = 1;
        for (auto& g : ctx->groups) tiles = std::max<size_t>(tiles, (size_t)g.first + g.count);  // (every launch's tiles at their own place)
        tiles = std::max<size_t>(tiles, ctx->n_strips);
        int32_t rc = ensure(ctx, ctx->tree_trace, tiles * 64);
        if (rc) return rc;
        HIP_TRY(ctx, hipMemsetAsync(ctx->tree_trace.p, 0, tiles * 64, ctx->stream));
    }
    if (out && ctx->tree_trace.p) return download(ctx, out, ctx->tree_trace.p, std::min<size_t>((size_t)n_tiles * 64, ctx->tree_trace.bytes));
    if (!enable && !out && ctx->tree_trace.p) {  // switch it off again: later launches carry no stamps
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipFree(ctx->tree_trace.p));
        ctx->tree_trace = DevBuf{};
    }
    return MI_OK;
}

// test hook (not part of the public header): the shape of the current tile plan
int32_t mi_debug_tile_plan(mi_ctx* ctx, uint32_t* out_launches, uint32_t* out_tiles, uint32_t* out_chain_tiles, uint32_t* out_bands) {
    ENTER(ctx);
    uint32_t tiles = 0, chain = 0;
    for (auto& g : ctx->groups) { tiles += g.count; chain += g.n_chain; }
    if (out_launches) *out_launches = ctx->by_levels ? ctx->n_levels : ctx->wave_forest ? 1u : (uint32_t)ctx->groups.size();  // (tile launches; by levels: one per level)
    if (ctx->wave_forest && !ctx->by_levels) tiles = ctx->n_wtiles, chain = 0;  // (the wave tiles of a forest of small trees: one launch)
    if (ctx->strip_plan && !ctx->by_levels) {  // (strips: one launch; "chain tiles" = the strips with a cone)
        tiles = ctx->n_strips, chain = ctx->n_strips > 0 ? ctx->n_strips - 1u : 0u;
        if (out_launches) *out_launches = 1u;
    }
    if (out_tiles) *out_tiles = tiles;
    if (out_chain_tiles) *out_chain_tiles = chain;
    if (out_bands) *out_bands = (uint32_t)ctx->passes.size();
    return MI_OK;
}

// development hook: the strips of the current plan -- per strip its rounds (padding included) and cone rounds (tools/strip_trace.py);
// *out_n = strips, *out_total_rounds = the table's rounds
int32_t mi_debug_strip_plan(mi_ctx* ctx, uint32_t* out_rounds, uint32_t* out_cone_rounds, uint32_t cap, uint32_t* out_n, uint32_t* out_total_rounds) {
    ENTER(ctx);
    if (out_n) *out_n = ctx->strip_plan ? ctx->n_strips : 0u;
    if (out_total_rounds) *out_total_rounds = ctx->strip_plan ? ctx->n_strip_rounds : 0u;
    if (!ctx->strip_plan || !cap || (!out_rounds && !out_cone_rounds)) return MI_OK;
    std::vector<StripDesc> sd(ctx->n_strips);
    std::vector<StripRound> rd(ctx->n_strip_rounds);
    int32_t rc = download(ctx, sd.data(), ctx->strips.p, sd.size() * sizeof(StripDesc));
    if (rc) return rc;
    if ((rc = download(ctx, rd.data(), ctx->strip_rounds.p, rd.size() * sizeof(StripRound)))) return rc;
    for (uint32_t i = 0; i < ctx->n_strips && i < cap; ++i) {
        const uint32_t nr = sd[i].n_rounds & 0xFFFFu;
        uint32_t cone = 0;
        for (uint32_t j = 0; j < nr; ++j) {
            const uint32_t info = rd[sd[i].first_round + j].info;
            if ((info & 0x7Fu) && !(info & STRIP_OWNED)) ++cone;
        }
        if (out_rounds) out_rounds[i] = nr;
        if (out_cone_rounds) out_cone_rounds[i] = cone;
    }
    return MI_OK;
}

// development hook: the launches of the current tile plan -- per group (first tile, tiles, chain tiles, deep instantiation) -- and per
// tile (levels, rows, chain length | 0x100 for a roots tile) (tools/shape_trace.py)
int32_t mi_debug_tile_groups(mi_ctx* ctx, uint32_t* out_groups, uint32_t cap_groups, uint32_t* out_n_groups, uint32_t* out_tiles, uint32_t cap_tiles) {
    ENTER(ctx);
    uint32_t ng = 0;
    for (auto& g : ctx->groups) {
        if (out_groups && ng < cap_groups) {
            out_groups[4 * ng] = g.first;
            out_groups[4 * ng + 1] = g.count;
            out_groups[4 * ng + 2] = g.n_chain;
            out_groups[4 * ng + 3] = g.deep ? 1u : 0u;
        }
        ++ng;
    }
    if (out_n_groups) *out_n_groups = ng;
    if (out_tiles && cap_tiles && ctx->tiles.p) {
        const size_t nt = std::min<size_t>(cap_tiles, ctx->tiles.bytes / sizeof(TileDesc));
        std::vector<TileDesc> td(nt);
        int32_t rc = download(ctx, td.data(), ctx->tiles.p, nt * sizeof(TileDesc));
        if (rc) return rc;
        for (size_t i = 0; i < nt; ++i) {
            uint32_t rows = 0;
            for (uint32_t k = 0; k < td[i].n_levels && k < TILE_MAX_LEVELS; ++k) rows += td[i].count[k];
            out_tiles[3 * i] = td[i].n_levels;
            out_tiles[3 * i + 1] = rows;
            out_tiles[3 * i + 2] = (td[i].kind & TILE_CHAIN_MASK) | ((td[i].kind & TILE_ROOTS) ? 0x100u : 0u);
        }
    }
    return MI_OK;
}

}  // extern "C"
