// ctx_hierarchy.cpp -- mi_upload_hierarchy: validation of the level-ordered parent array and the subtree-tile plan.
#include "ctx.h"

using namespace mi;
using namespace mi_detail;

extern "C" {

int32_t mi_upload_hierarchy(mi_ctx* ctx, uint32_t n, const uint32_t* parent_idx, const uint32_t* level_offsets,
                            uint32_t n_levels) {
    ENTER(ctx);
    if (n != ctx->n) return fail(ctx, MI_ERR_INVALID_ARG, "mi_upload_hierarchy: n (%u) != live rows (%u)", n, ctx->n);
    ctx->changed_maybe = true;  // conservative: the next propagate looks at the rows again
    if (!parent_idx || n_levels <= 1) {
        ctx->have_hierarchy = false;
        ctx->n_levels = 1;
        ctx->level_offsets = {0, n};
        ctx->passes.clear();
        ctx->groups.clear();
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
    // A pass = one launch covering `d` consecutive levels; its tiles partition the rows of the level the pass is
    // rooted in.  Pass 0 is rooted in level 0 itself (tile level 0 = a range of roots / flat rows); later passes
    // are rooted in the last level of the previous pass (tile level 0 = the children of a range of its rows).
    // A tile "fits" when all its levels but the last together hold <= TILE_UCAP rows (they live in LDS).
    const char* env_levels = getenv("MI_TILE_LEVELS");
    const uint32_t max_d = env_levels ? std::max(1, std::min((int)TILE_MAX_LEVELS, atoi(env_levels))) : TILE_MAX_LEVELS;
    std::vector<TileDesc> tiles;
    std::vector<uint32_t> chains;
    ctx->passes.clear();
    ctx->groups.clear();
    auto level_size = [&](uint32_t lv) -> uint64_t { return lv < n_levels ? level_offsets[lv + 1] - level_offsets[lv] : 0; };
    // MI_TILE_PLAN="2,5,4": explicit band depths (experiments)
    std::vector<uint32_t> plan;
    if (const char* pe = getenv("MI_TILE_PLAN")) {
        for (const char* q = pe; *q;) {
            plan.push_back((uint32_t)std::max(1, atoi(q)));
            while (*q && *q != ',') ++q;
            if (*q == ',') ++q;
        }
    }
    uint32_t band = 0;
    uint32_t l = 0;  // first level this pass computes
    while (l < n_levels) {
        const bool roots = l == 0;
        // band depth: as deep as possible while an average root range of one row still fits in LDS
        const uint64_t n_roots = std::max<uint64_t>(1, roots ? level_size(0) : level_size(l - 1));
        uint32_t d = 1;
        uint64_t upper = level_size(l);  // rows of the levels that would be non-last if we add one more level
        while (d < max_d && l + d < n_levels && upper <= (uint64_t)TILE_UCAP * n_roots) {
            ++d;
            upper += level_size(l + d - 1);
        }
        if (band < plan.size()) d = std::min<uint32_t>(std::min<uint32_t>(plan[band], TILE_MAX_LEVELS), n_levels - l);
        ++band;
        // [lo,hi) is a row range of the rooting level; returns the tile and whether it fits
        auto build = [&](uint32_t lo, uint32_t hi, TileDesc& td) -> bool {
            uint32_t clo = lo, chi2 = hi;
            td = TileDesc{};
            for (uint32_t k = 0; k < d; ++k) {
                uint32_t nlo, nhi;
                if (roots && k == 0) { nlo = lo; nhi = hi; }
                else {
                    const uint32_t plevel = roots ? k - 1 : l - 1 + k;
                    nlo = child_begin(plevel, clo);
                    nhi = child_begin(plevel, chi2);
                }
                td.start[k] = nlo;
                td.count[k] = nhi - nlo;
                clo = nlo; chi2 = nhi;
                if (nhi > nlo) td.n_levels = k + 1;
            }
            uint64_t up = 0;
            for (uint32_t k = 0; k + 1 < td.n_levels; ++k) up += td.count[k];
            return up <= TILE_UCAP;
        };
        // Chain candidate (see "Tile kinds" below): then every tile hangs below exactly one node, as long as that
        // still gives tiles of a decent size.
        uint64_t pass_rows = 0;
        for (uint32_t k = 0; k < d; ++k) pass_rows += level_size(l + k);
        // (several chained passes may share a launch: chain tiles read the pre-frame snapshot, nobody waits for anybody)
        const bool chain_candidate = !roots && l <= TILE_MAX_CHAIN && !ctx->groups.empty() &&
                                     (!plan.empty() || (ctx->groups.back().n_chain == 0 && ctx->groups.back().count <= 64)) &&
                                     getenv("MI_TILE_NO_CHAIN") == nullptr && n_roots <= 16384 && pass_rows >= 128 * n_roots;
        const uint32_t first_tile = (uint32_t)tiles.size();
        const uint32_t rl = roots ? 0 : l - 1;
        const uint32_t rlo = level_offsets[rl], rhi = level_offsets[rl + 1];
        uint32_t a = rlo;
        while (a < rhi) {
            TileDesc best{};
            uint32_t b = a + 1;
            build(a, b, best);
            uint32_t step = 1;  // galloping extension of the root range
            while (b < rhi && !chain_candidate) {
                const uint32_t nb = (uint32_t)std::min<uint64_t>((uint64_t)b + step, rhi);
                TileDesc cand{};
                // keep tiles small enough to spread over the chip: at most 4 x TILE_UCAP rows in the streamed last level
                if (build(a, nb, cand) && (cand.n_levels == 0 || cand.count[cand.n_levels - 1] <= 4 * TILE_UCAP || nb == a + 1)) { best = cand; b = nb; step *= 2; }
                else if (step > 1) step = 1;
                else break;
            }
            if (best.n_levels) tiles.push_back(best);
            a = b;
        }
        const uint32_t n_pass_tiles = (uint32_t)tiles.size() - first_tile;
        ctx->passes.emplace_back(first_tile, n_pass_tiles);
        // Tile kinds.  Pass 0: roots.  A later pass whose every tile hangs below ONE node of a short enough ancestor
        // chain lets each tile re-evaluate that chain itself (kernels_tree.hip), which makes the pass independent of
        // the one above it: it joins the previous launch.  Otherwise its tiles read their parents from global memory
        // and the pass needs its own launch behind the previous one.
        // (the owners wait for the chain tiles to start, so there must be few of them and only one chained pass per launch)
        bool chainable = chain_candidate;
        for (uint32_t ti = first_tile; chainable && ti < tiles.size(); ++ti) {
            const TileDesc& td = tiles[ti];
            if (td.n_levels == 0) continue;
            const uint32_t p0 = parent_idx[td.start[0]];
            if (parent_idx[td.start[0] + td.count[0] - 1] != p0) chainable = false;  // level 0 of the tile spans several parents
        }
        chains.resize(tiles.size() * (size_t)TILE_MAX_CHAIN, 0u);
        for (uint32_t ti = first_tile; ti < tiles.size(); ++ti) {
            TileDesc& td = tiles[ti];
            td.kind = roots ? TILE_ROOTS : 0u;
            if (chainable && td.n_levels) {
                uint32_t row = parent_idx[td.start[0]], len = 0;
                while (row != MI_NO_PARENT && len < TILE_MAX_CHAIN) {
                    chains[(size_t)ti * TILE_MAX_CHAIN + len++] = row;
                    row = parent_idx[row];
                }
                td.kind = len;
            }
        }
        if (chainable) {
            ctx->groups.back().count += n_pass_tiles;
            ctx->groups.back().n_chain += n_pass_tiles;
            ctx->groups.back().owner_rows = level_offsets[l];  // the snapshot prefix: every row above the deepest chained pass
        } else {
            ctx->groups.push_back({first_tile, n_pass_tiles, 0u, 0u});
        }
        l += d;
    }
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
    ctx->have_hierarchy = true;
    return MI_OK;
}

}  // extern "C"
