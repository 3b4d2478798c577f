// Host-only part of libfep_hip.so (no HIP, plain C++17): the symbolic phase of the assembly and the gather plans of
// the kernels.  Kept free of device code so that the same functions build into a host test binary with
// -fsanitize=address,undefined / thread (tests/host_san.cpp, tests/test_host_sanitizers.py).
//
//   build_symbolic   node graph of the mesh -> CSR pattern of K on node-pair blocks, per block the list of
//                    element-local blocks that sum into it, per node its (element, local node) incidences
//                    (the index side of K = B^T D B, DP:549-595)
//   build_p1_plan    P1 fast path: tiles of whole nodes, per tile the staged element / node lists, gather codes,
//                    packed block descriptors (p1_node_lds_kernel, p1_fused_kernel)
//   build_gn_plan    the same for the opt-in node route of P2 / Q1 / Q2
//   aggregate        greedy aggregation of a node graph (multigrid setup, solver.py)
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <new>
#include <system_error>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/fep.h"

namespace fep_host {

struct U2 { uint32_t x, y; };

inline int worker_count() {
    if (const char* e = std::getenv("FEP_HOST_THREADS")) { const int v = std::atoi(e); if (v >= 1) return std::min(v, 64); }
    return (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
}

// f(lo, hi, worker) over [0, n) in contiguous chunks, one per worker thread.  If a thread cannot be started
// (process / thread limits) its chunk runs on the calling thread: never throws std::system_error to the caller.
// An exception thrown INSIDE a chunk (std::bad_alloc of a work buffer at BASELINE configs[4] sizes on a small host) is
// caught in its thread — an exception leaving a std::thread body is std::terminate — and rethrown here, on the calling
// thread, once every worker has been joined; fep_ctx_create turns it into FEP_ENOMEM.
template <class F> void parallel_chunks(int64_t n, F&& f) {
    const int nw = (int)std::max<int64_t>(1, std::min<int64_t>(worker_count(), n));
    std::vector<std::thread> th;
    th.reserve(nw);
    std::vector<int> inline_chunks;
    std::vector<std::exception_ptr> err((size_t)nw);
    auto guarded = [&f, &err](int64_t lo, int64_t hi, int w) {
        try { f(lo, hi, w); } catch (...) { err[(size_t)w] = std::current_exception(); }
    };
    for (int w = 1; w < nw; ++w) {
        const int64_t lo = n * w / nw, hi = n * (w + 1) / nw;
        try { th.emplace_back([&guarded, lo, hi, w]() { guarded(lo, hi, w); }); }
        catch (const std::system_error&) { inline_chunks.push_back(w); }
    }
    guarded(0, n / nw, 0);
    for (int w : inline_chunks) guarded(n * w / nw, n * (w + 1) / nw, w);
    for (auto& t : th) t.join();
    for (const std::exception_ptr& e : err)
        if (e) std::rethrow_exception(e);
}

// ---------------------------------------------------------------------------------------
// Symbolic phase.  Node graph of the mesh -> (i) CSR pattern of K on node-pair blocks,
// (ii) per block the list of element-local blocks that sum into it,
// (iii) per node the list of (element, local node) pairs for the force gather.
// ---------------------------------------------------------------------------------------
struct Symbolic {
    std::vector<int32_t> iptr, ilist;        // node -> incident (a*n_e + e), ordered by (e, a)
    std::vector<int32_t> nptr, ncol;         // node CSR (sorted neighbour nodes); block id = position in ncol
    std::vector<int32_t> segptr, perm;       // block -> contributions (a*NP+b)*n_e + e
    std::vector<uint32_t> meta;              // block -> (deg << 16) | (diag << 15) | slot
};

inline int build_symbolic(int n_p, int64_t n_e, int64_t n_n, const int32_t* elem, Symbolic& S) {
    if ((int64_t)n_p * n_p * n_e >= (int64_t)INT32_MAX / 2) return FEP_ERANGE;
    for (int64_t i = 0; i < (int64_t)n_p * n_e; ++i)
        if (elem[i] < 0 || elem[i] >= n_n) return FEP_ERANGE;
    // (iii) incidence lists
    S.iptr.assign(n_n + 1, 0);
    for (int64_t i = 0; i < (int64_t)n_p * n_e; ++i) S.iptr[elem[i] + 1]++;
    for (int64_t n = 0; n < n_n; ++n) S.iptr[n + 1] += S.iptr[n];
    S.ilist.resize(S.iptr[n_n]);
    {
        std::vector<int32_t> fill(S.iptr.begin(), S.iptr.end() - 1);
        for (int64_t e = 0; e < n_e; ++e)
            for (int a = 0; a < n_p; ++a) S.ilist[fill[elem[(int64_t)a * n_e + e]]++] = (int32_t)((int64_t)a * n_e + e);
    }
    // (i)+(ii) per node: gather (neighbour, code) pairs, sort by neighbour (stable in (e,a,b) order)
    std::vector<int32_t> deg(n_n, 0);
    auto node_pairs = [&](int64_t n, std::vector<std::pair<int32_t, int32_t>>& buf) {
        buf.clear();
        for (int32_t t = S.iptr[n]; t < S.iptr[n + 1]; ++t) {
            const int64_t code = S.ilist[t];
            const int a = (int)(code / n_e);
            const int64_t e = code - (int64_t)a * n_e;
            for (int b = 0; b < n_p; ++b)
                buf.emplace_back(elem[(int64_t)b * n_e + e], (int32_t)(((int64_t)a * n_p + b) * n_e + e));
        }
        std::stable_sort(buf.begin(), buf.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
    };
    parallel_chunks(n_n, [&](int64_t lo, int64_t hi, int) {                      // pass 1: degrees
        std::vector<std::pair<int32_t, int32_t>> buf;
        for (int64_t n = lo; n < hi; ++n) {
            node_pairs(n, buf);
            int32_t d = 0;
            for (size_t i = 0; i < buf.size(); ++i)
                if (i == 0 || buf[i].first != buf[i - 1].first) ++d;
            deg[n] = d;
        }
    });
    S.nptr.assign(n_n + 1, 0);
    int64_t tot = 0;
    for (int64_t n = 0; n < n_n; ++n) {
        if (deg[n] > 0x7fff) return FEP_ERANGE;
        tot += deg[n];
        if (tot >= INT32_MAX / 4) return FEP_ERANGE;
        S.nptr[n + 1] = (int32_t)tot;
    }
    const int64_t n_blk = tot;
    const int64_t n_contrib = (int64_t)n_p * n_p * n_e;
    S.ncol.resize(n_blk);
    S.meta.resize(n_blk);
    S.segptr.assign(n_blk + 1, 0);
    S.perm.resize(n_contrib);
    // contributions of node n start at n_p * iptr[n] (each incident (e,a) brings n_p pairs)
    parallel_chunks(n_n, [&](int64_t lo, int64_t hi, int) {
        std::vector<std::pair<int32_t, int32_t>> buf;
        for (int64_t n = lo; n < hi; ++n) {
            node_pairs(n, buf);
            int64_t pos = (int64_t)n_p * S.iptr[n];
            int32_t slot = -1;
            for (size_t i = 0; i < buf.size(); ++i) {
                if (i == 0 || buf[i].first != buf[i - 1].first) {
                    ++slot;
                    const int64_t blk = S.nptr[n] + slot;
                    S.ncol[blk] = buf[i].first;
                    S.meta[blk] = ((uint32_t)deg[n] << 16) | (buf[i].first == (int32_t)n ? 0x8000u : 0u) | (uint32_t)slot;
                    S.segptr[blk] = (int32_t)pos;
                }
                S.perm[pos++] = buf[i].second;
            }
        }
    });
    S.segptr[n_blk] = (int32_t)n_contrib;
    return FEP_OK;
}

// Tiles of whole nodes with at most `tile` node-pair blocks, one run of consecutive nodes each (a node's blocks are
// consecutive ids, so the CSR values a tile produces are one contiguous range): csr_reduce_kernel's work units.
inline int row_tiles(const Symbolic& S, int64_t n_n, int tile, std::vector<int32_t>& tstart) {
    tstart.assign(1, 0);
    for (int64_t n = 0, cur = 0; n < n_n; ++n) {
        const int64_t d = S.nptr[n + 1] - S.nptr[n];
        if (d > tile) return FEP_ERANGE;
        if (cur + d > tile) { tstart.push_back(S.nptr[n]); cur = 0; }
        cur += d;
    }
    tstart.push_back(S.nptr[n_n]);
    return FEP_OK;
}

// ---------------------------------------------------------------------------------------
// P1 fast path: gather plan of p1_node_lds_kernel / p1_fused_kernel.
//
// A tile = up to kSegMax SEGMENTS, each a run of consecutive nodes (with all their blocks), at most `tile` blocks and
// 255 nodes in total.  One segment per tile gives the row strips of round 1 (36 consecutive nodes: ~150 staged elements
// for 72 owned ones on a row-numbered structured mesh).  With two segments the second one is the longest run of
// consecutive, still unassigned neighbour ids of the first (on a row-numbered mesh: the row above), which makes the
// tile two rows high: ~114 staged elements and ~80 instead of ~114 staged nodes for the same 256 blocks.  Every tile
// still writes its CSR values / forces as one contiguous range PER SEGMENT.  Which tiling is used is decided by
// counting: the multi-segment one must stage at least 5 % fewer elements over the whole mesh.
//
//   tdesc  kDescInts int32 per tile: [pk_base, n_blocks, n_nodes, n_segments | staged elements << 4 | staged nodes << 16]
//          then kSegMax x [first_block, n_blocks, first_node, n_nodes]  (scalar loads in the kernels)
//   elist_pad / rng_tab   sorted unique elements of the tile, padded to L with its first element / as <= 8 runs
//   codes_pad             gather codes (local element << 4 | a << 2 | b) in tile block order, padded to C
//   pkv                   per lane of the tile (lanes sorted by descending segment length) the packed descriptor
//                         x: code offset:11 | len:4 | deg:8 | slot:8 | diag:1,  y: block index in tile | node index << 8
//   fused step:  nlist_pad / nrng_tab  the nodes the staged elements touch;  elnodes  per staged element its three
//                tile-local node indices (10 bits each) | owner bit 30 (first tile in order that stages the element)
// ---------------------------------------------------------------------------------------
constexpr int kSegMax = 4;
constexpr int kDescInts = 4 * (1 + kSegMax);

struct P1Options {
    int tile = 256;
    int max_segs = 2;
    int staged_cap = 128;                    // multi-segment tiles that would stage more elements are split into their segments
    bool allow_lds = true, allow_rng = true, allow_pk = true, allow_fused = true;
};

struct P1Plan {
    int tile = 256, n_segs = 1;
    int64_t n_wg = 0, staged_total = 0, staged_nodes_total = 0;
    std::vector<int32_t> tdesc;
    std::vector<int32_t> perm2;              // global gather list (e << 4 | a << 2 | b), block order: direct kernel
    bool lds = false, rng = false, pk = false, fused = false, fused_rng = false;
    int L = 0, C = 0, NL = 0;
    std::vector<int32_t> elist_pad, rng_tab, nlist_pad, nrng_tab;
    std::vector<uint16_t> codes_pad;
    std::vector<U2> pkv;
    std::vector<uint32_t> elnodes;
};

struct Tile { int nseg = 0; int32_t fb[kSegMax] = {}, nb[kSegMax] = {}, fn[kSegMax] = {}, nn[kSegMax] = {}; };

inline int make_tiles(const Symbolic& S, int64_t n_n, int tile, int max_segs, std::vector<Tile>& tiles) {
    tiles.clear();
    max_segs = std::max(1, std::min(max_segs, kSegMax));
    std::vector<uint8_t> assigned((size_t)n_n, 0);
    for (int64_t n = 0; n < n_n; ++n) assigned[n] = S.nptr[n + 1] == S.nptr[n];      // nodes of no element: no blocks, no tile
    std::vector<int32_t> nbrs;
    int64_t p = 0;
    while (true) {
        while (p < n_n && assigned[p]) ++p;
        if (p >= n_n) break;
        Tile t;
        int total_nodes = 0, total_blocks = 0;
        int64_t start = p;
        for (int s = 0; s < max_segs; ++s) {
            // an even share of what is left of the tile: the first segment takes tile / max_segs blocks, the last one
            // whatever the others left, so that tiles come out full (their number sets the kernels' time)
            // (but no segment much longer than its even share: a long one stages as much as a whole row strip)
            const int seg_budget = std::min((tile - total_blocks + (max_segs - s) - 1) / (max_segs - s),
                                            max_segs > 1 ? tile / max_segs + 8 : tile);
            int64_t n = start;
            int blocks = 0;
            while (n < n_n && !assigned[n] && blocks + (S.nptr[n + 1] - S.nptr[n]) <= seg_budget && total_nodes + (int)(n - start) < 255) {
                blocks += S.nptr[n + 1] - S.nptr[n];
                ++n;
            }
            if (n == start) {
                if (s > 0) break;
                if (S.nptr[n + 1] - S.nptr[n] > tile) return FEP_ERANGE;            // a single node does not fit a tile
                blocks = S.nptr[n + 1] - S.nptr[n];                                  // alone in its tile
                ++n;
            }
            t.fb[s] = S.nptr[start]; t.nb[s] = blocks; t.fn[s] = (int32_t)start; t.nn[s] = (int32_t)(n - start);
            t.nseg = s + 1;
            total_nodes += (int)(n - start);
            total_blocks += blocks;
            for (int64_t m = start; m < n; ++m) assigned[m] = 1;
            if (s + 1 == max_segs || total_blocks >= tile) break;
            // next segment: the longest run of consecutive unassigned neighbour ids of this one ...
            nbrs.clear();
            for (int32_t b = S.nptr[start]; b < S.nptr[n]; ++b)
                if (!assigned[S.ncol[b]]) nbrs.push_back(S.ncol[b]);
            std::sort(nbrs.begin(), nbrs.end());
            nbrs.erase(std::unique(nbrs.begin(), nbrs.end()), nbrs.end());
            size_t best = 0, best_len = 0;
            for (size_t i = 0; i < nbrs.size();) {
                size_t j = i + 1;
                while (j < nbrs.size() && nbrs[j] == nbrs[j - 1] + 1) ++j;
                if (j - i > best_len) { best_len = j - i; best = i; }
                i = j;
            }
            if (best_len < 2) break;
            // ... started below it when that closes a short gap (what the previous tile's segment left of that row;
            // such leftovers would otherwise make small tiles of their own); never walks into untouched territory
            start = nbrs[best];
            {
                int64_t lo = start;
                int gap = 0;
                while (lo > 0 && !assigned[lo - 1] && gap < 32) { --lo; ++gap; }
                if (gap < 32) start = lo;
            }
        }
        tiles.push_back(t);
    }
    return FEP_OK;
}

inline void runs_of(const std::vector<int32_t>& l, int32_t* d /* 16 ints */, bool& fits) {
    int nr = 0, cum = 0;
    for (size_t i = 0; i < l.size();) {
        size_t j = i + 1;
        while (j < l.size() && l[j] == l[j - 1] + 1) ++j;
        if (nr == 8) { fits = false; break; }
        cum += (int)(j - i);
        d[2 * nr] = l[i]; d[2 * nr + 1] = cum;
        ++nr; i = j;
    }
    for (; nr < 8; ++nr) { d[2 * nr] = l.empty() ? 0 : l[0]; d[2 * nr + 1] = cum; }
}

inline int build_p1_plan_segs(const Symbolic& S, int64_t n_e, int64_t n_n, const int32_t* elem, const P1Options& opt,
                              int max_segs, P1Plan& P) {
    P = P1Plan();
    P.tile = opt.tile;
    const int TILE = opt.tile;
    if (n_e >= (int64_t)1 << 27) return FEP_ERANGE;
    const int64_t n_blk = (int64_t)S.ncol.size();
    P.perm2.resize(S.perm.size());
    for (size_t i = 0; i < S.perm.size(); ++i) {
        const int64_t code = S.perm[i];
        const int64_t ab = code / n_e, e = code - ab * n_e;
        P.perm2[i] = (int32_t)((e << 4) | ((ab / 3) << 2) | (ab % 3));
    }
    std::vector<Tile> tiles;
    const int r = make_tiles(S, n_n, TILE, max_segs, tiles);
    if (r != FEP_OK) return r;
    if (max_segs > 1 && opt.staged_cap > 0) {
        // a multi-segment tile whose segments turned out not to share elements (first rows, junctions of the brick
        // pattern) stages as much as two tiles: make it two, so that the LDS image of EVERY tile stays small
        std::vector<int32_t> count(tiles.size(), 0);
        parallel_chunks((int64_t)tiles.size(), [&](int64_t lo, int64_t hi, int) {
            std::vector<int32_t> l;
            for (int64_t g = lo; g < hi; ++g) {
                const Tile& t = tiles[g];
                if (t.nseg < 2) continue;
                l.clear();
                for (int s = 0; s < t.nseg; ++s)
                    for (int32_t c = S.segptr[t.fb[s]]; c < S.segptr[t.fb[s] + t.nb[s]]; ++c) l.push_back(P.perm2[c] >> 4);
                std::sort(l.begin(), l.end());
                count[g] = (int32_t)(std::unique(l.begin(), l.end()) - l.begin());
            }
        });
        std::vector<Tile> split;
        split.reserve(tiles.size());
        for (size_t g = 0; g < tiles.size(); ++g) {
            const Tile& t = tiles[g];
            if (count[g] <= opt.staged_cap) { split.push_back(t); continue; }
            for (int s = 0; s < t.nseg; ++s) {
                Tile u;
                u.nseg = 1; u.fb[0] = t.fb[s]; u.nb[0] = t.nb[s]; u.fn[0] = t.fn[s]; u.nn[0] = t.nn[s];
                split.push_back(u);
            }
        }
        tiles.swap(split);
    }
    const int64_t n_wg = (int64_t)tiles.size();
    P.n_wg = n_wg;
    P.tdesc.assign((size_t)n_wg * kDescInts, 0);
    int64_t pk_base = 0;
    for (int64_t g = 0; g < n_wg; ++g) {
        const Tile& t = tiles[g];
        int32_t* d = P.tdesc.data() + g * kDescInts;
        int nb = 0, nn = 0;
        for (int s = 0; s < t.nseg; ++s) {
            d[4 + 4 * s] = t.fb[s]; d[5 + 4 * s] = t.nb[s]; d[6 + 4 * s] = t.fn[s]; d[7 + 4 * s] = t.nn[s];
            nb += t.nb[s]; nn += t.nn[s];
        }
        d[0] = (int32_t)pk_base; d[1] = nb; d[2] = nn; d[3] = t.nseg;
        pk_base += nb;
        P.n_segs = std::max(P.n_segs, t.nseg);
    }
    if (pk_base != n_blk) return FEP_EINVAL;                                         // tiles partition the blocks
    if (!opt.allow_lds) return FEP_OK;
    // per tile: blocks in tile order, the staged elements, local gather codes.  The staged list holds the elements the tile
    // OWNS first (the first tile in order that stages an element owns it: it alone writes the element's point outputs in
    // the one-kernel step), ascending, then the others, ascending: the owner lanes of the one-kernel step are then lanes
    // 0 .. n_own - 1 — whole waves store the point outputs instead of a scattered half of every wave.
    std::vector<std::vector<int32_t>> lists(n_wg);
    std::vector<std::vector<uint16_t>> codes(n_wg);
    std::vector<uint8_t> too_big((size_t)n_wg, 0);
    std::vector<int32_t> n_own((size_t)n_wg, 0);
    parallel_chunks(n_wg, [&](int64_t lo, int64_t hi, int) {
        for (int64_t g = lo; g < hi; ++g) {
            const Tile& t = tiles[g];
            std::vector<int32_t>& l = lists[g];
            for (int s = 0; s < t.nseg; ++s)
                for (int32_t c = S.segptr[t.fb[s]]; c < S.segptr[t.fb[s] + t.nb[s]]; ++c) l.push_back(P.perm2[c] >> 4);
            std::sort(l.begin(), l.end());
            l.erase(std::unique(l.begin(), l.end()), l.end());
            if (l.size() > 4095) too_big[g] = 1;
        }
    });
    for (int64_t g = 0; g < n_wg; ++g)
        if (too_big[g]) return FEP_OK;                                               // plan without the LDS route
    {   // owners, in tile order (sequential), then every list: owned elements first
        std::vector<uint8_t> owned((size_t)n_e, 0);
        std::vector<int32_t> rest;
        for (int64_t g = 0; g < n_wg; ++g) {
            std::vector<int32_t>& l = lists[g];
            rest.clear();
            size_t k = 0;
            for (size_t i = 0; i < l.size(); ++i) {
                const int32_t e = l[i];
                if (!owned[e]) { owned[e] = 1; l[k++] = e; } else rest.push_back(e);
            }
            n_own[g] = (int32_t)k;
            std::copy(rest.begin(), rest.end(), l.begin() + (int64_t)k);
        }
    }
    parallel_chunks(n_wg, [&](int64_t lo, int64_t hi, int) {
        std::vector<std::pair<int32_t, int32_t>> where;                              // (element, slot), sorted by element
        for (int64_t g = lo; g < hi; ++g) {
            const Tile& t = tiles[g];
            const std::vector<int32_t>& l = lists[g];
            where.clear();
            for (size_t i = 0; i < l.size(); ++i) where.emplace_back(l[i], (int32_t)i);
            std::sort(where.begin(), where.end());
            std::vector<uint16_t>& cd = codes[g];
            for (int s = 0; s < t.nseg; ++s)
                for (int32_t c = S.segptr[t.fb[s]]; c < S.segptr[t.fb[s] + t.nb[s]]; ++c) {
                    const int32_t e = P.perm2[c] >> 4;
                    const int32_t loc = std::lower_bound(where.begin(), where.end(), std::make_pair(e, (int32_t)0))->second;
                    cd.push_back((uint16_t)((loc << 4) | (P.perm2[c] & 15)));
                }
        }
    });
    size_t lmax = 0, cmax = 0;
    for (int64_t g = 0; g < n_wg; ++g) {
        lmax = std::max(lmax, lists[g].size());
        cmax = std::max(cmax, codes[g].size());
        P.staged_total += (int64_t)lists[g].size();
    }
    for (int64_t g = 0; g < n_wg; ++g) P.tdesc[(size_t)g * kDescInts + 3] |= (int32_t)(lists[g].size() << 4);   // staged elements of the tile
    P.C = (int)((cmax + 7) & ~(size_t)7);
    P.L = (int)((lmax + 1) & ~(size_t)1);
    // 15 doubles per staged element; the staged kernels hold ONE element and <= 4 gather codes per lane in registers.
    // (A tile of B blocks over N nodes stages at most B - N elements — every staged element is a distinct neighbour
    // relation of one of its nodes — and touches at most B nodes, so with B <= tile = threads both always fit.)
    P.lds = lmax <= (size_t)TILE && cmax <= 4 * (size_t)TILE &&
            (size_t)P.L * 15 * sizeof(double) + (size_t)P.C * 2 <= 96 * 1024;
    if (!P.lds) return FEP_OK;
    const int64_t LP = P.L, CP = P.C;
    P.elist_pad.assign((size_t)(n_wg * LP), 0);
    P.codes_pad.assign((size_t)(n_wg * CP), 0);
    P.rng_tab.assign((size_t)n_wg * 16, 0);
    bool fits = opt.allow_rng;
    for (int64_t g = 0; g < n_wg; ++g) {
        const std::vector<int32_t>& l = lists[g];
        std::fill(P.elist_pad.begin() + g * LP, P.elist_pad.begin() + (g + 1) * LP, l.empty() ? 0 : l[0]);
        std::copy(l.begin(), l.end(), P.elist_pad.begin() + g * LP);
        std::copy(codes[g].begin(), codes[g].end(), P.codes_pad.begin() + g * CP);
        if (fits) runs_of(l, P.rng_tab.data() + g * 16, fits);
    }
    P.rng = fits;
    // packed block descriptors, if every field fits its bit width; lanes sorted by descending segment length
    bool pk_ok = opt.allow_pk && CP <= 2047 && TILE <= 256;
    P.pkv.assign((size_t)n_blk, U2{0u, 0u});
    std::vector<uint8_t> pk_bad((size_t)n_wg, 0);
    if (pk_ok)
        parallel_chunks(n_wg, [&](int64_t lo, int64_t hi, int) {
            std::vector<int> order, blk, node_of, off;
            for (int64_t g = lo; g < hi; ++g) {
                const Tile& t = tiles[g];
                const int32_t* d = P.tdesc.data() + g * kDescInts;
                const int nb = d[1];
                if (d[2] > 255) { pk_bad[g] = 1; continue; }
                order.resize(nb); blk.resize(nb); node_of.resize(nb); off.resize(nb);
                int k = 0, node_base = 0, o = 0;
                for (int s = 0; s < t.nseg; ++s) {
                    for (int32_t n = t.fn[s]; n < t.fn[s] + t.nn[s]; ++n)
                        for (int32_t b = S.nptr[n]; b < S.nptr[n + 1]; ++b) {
                            blk[k] = b; node_of[k] = node_base + (n - t.fn[s]); off[k] = o;
                            o += S.segptr[b + 1] - S.segptr[b];
                            ++k;
                        }
                    node_base += t.nn[s];
                }
                if (k != nb) { pk_bad[g] = 1; continue; }
                for (int i = 0; i < nb; ++i) order[i] = i;
                std::stable_sort(order.begin(), order.end(), [&](int x, int y) {
                    return S.segptr[blk[x] + 1] - S.segptr[blk[x]] > S.segptr[blk[y] + 1] - S.segptr[blk[y]];
                });
                for (int lane = 0; lane < nb; ++lane) {
                    const int kk = order[lane];
                    const int32_t b = blk[kk];
                    const uint32_t mt = S.meta[b];
                    const uint32_t len = (uint32_t)(S.segptr[b + 1] - S.segptr[b]), deg = mt >> 16, slot = mt & 0x7fffu;
                    const uint32_t diag = (mt >> 15) & 1u;
                    if (len > 15 || deg > 255 || slot > 255 || off[kk] > 2047) { pk_bad[g] = 1; break; }
                    P.pkv[(size_t)d[0] + lane] = U2{(uint32_t)off[kk] | (len << 11) | (deg << 15) | (slot << 23) | (diag << 31),
                                                    (uint32_t)kk | ((uint32_t)node_of[kk] << 8)};
                }
            }
        });
    for (int64_t g = 0; g < n_wg && pk_ok; ++g) pk_ok = !pk_bad[g];
    P.pk = pk_ok;
    if (!pk_ok) { P.pkv.clear(); return FEP_OK; }
    if (!opt.allow_fused) return FEP_OK;
    // one-kernel step: the nodes every tile's staged elements touch and, per staged element, its tile-local node indices
    std::vector<std::vector<int32_t>> nlists(n_wg);
    P.elnodes.assign((size_t)(n_wg * LP), 0u);
    std::vector<uint8_t> nbad((size_t)n_wg, 0);
    parallel_chunks(n_wg, [&](int64_t lo, int64_t hi, int) {
        for (int64_t g = lo; g < hi; ++g) {
            const std::vector<int32_t>& l = lists[g];
            std::vector<int32_t>& nl = nlists[g];
            nl.reserve(3 * l.size());
            for (int32_t e : l)
                for (int a = 0; a < 3; ++a) nl.push_back(elem[(int64_t)a * n_e + e]);
            std::sort(nl.begin(), nl.end());
            nl.erase(std::unique(nl.begin(), nl.end()), nl.end());
            if (nl.size() > 1023) { nbad[g] = 1; continue; }
            for (size_t i = 0; i < l.size(); ++i) {
                uint32_t word = 0u;
                for (int a = 0; a < 3; ++a) {
                    const int32_t nd = elem[(int64_t)a * n_e + l[i]];
                    word |= (uint32_t)(std::lower_bound(nl.begin(), nl.end(), nd) - nl.begin()) << (10 * a);
                }
                P.elnodes[(size_t)(g * LP) + i] = word;
            }
            for (size_t i = l.size(); i < (size_t)LP; ++i) P.elnodes[(size_t)(g * LP) + i] = P.elnodes[(size_t)(g * LP)];
        }
    });
    size_t nlmax = 0;
    for (int64_t g = 0; g < n_wg; ++g) {
        if (nbad[g]) { P.elnodes.clear(); return FEP_OK; }
        nlmax = std::max(nlmax, nlists[g].size());
        P.staged_nodes_total += (int64_t)nlists[g].size();
    }
    if (nlmax == 0 || nlmax > (size_t)TILE) { P.elnodes.clear(); return FEP_OK; }
    for (int64_t g = 0; g < n_wg; ++g) {                                             // owner bit = slot < n_own; the count rides in the descriptor
        for (int32_t i = 0; i < n_own[g]; ++i) P.elnodes[(size_t)(g * LP) + i] |= 1u << 30;
        P.tdesc[(size_t)g * kDescInts + 2] |= n_own[g] << 8;
    }
    const int64_t NLP = (int64_t)((nlmax + 1) & ~(size_t)1);
    P.NL = (int)NLP;
    P.nlist_pad.assign((size_t)(n_wg * NLP), 0);
    P.nrng_tab.assign((size_t)n_wg * 16, 0);
    bool nfits = P.rng;
    for (int64_t g = 0; g < n_wg; ++g) {
        const std::vector<int32_t>& nl = nlists[g];
        std::fill(P.nlist_pad.begin() + g * NLP, P.nlist_pad.begin() + (g + 1) * NLP, nl[0]);
        std::copy(nl.begin(), nl.end(), P.nlist_pad.begin() + g * NLP);
        if (nfits) runs_of(nl, P.nrng_tab.data() + g * 16, nfits);
    }
    for (int64_t g = 0; g < n_wg; ++g) P.tdesc[(size_t)g * kDescInts + 3] |= (int32_t)(nlists[g].size() << 16);  // staged nodes of the tile
    P.fused_rng = nfits;
    const size_t lds_f = (((size_t)P.L * 15 * sizeof(double) + (size_t)P.C * 2 + 15) & ~(size_t)15) + (size_t)NLP * 32;
    P.fused = lds_f <= 96 * 1024;
    if (!P.fused) P.elnodes.clear();
    return FEP_OK;
}

// The plan the kernels run: the multi-segment tiling when it has every compression the fast kernels need AND stages
// at least 5 % fewer elements than the row strips, else the row strips.
inline int build_p1_plan(const Symbolic& S, int64_t n_e, int64_t n_n, const int32_t* elem, const P1Options& opt, P1Plan& P) {
    int r = build_p1_plan_segs(S, n_e, n_n, elem, opt, 1, P);
    if (r != FEP_OK || opt.max_segs <= 1 || !P.lds || !P.pk) return r;
    P1Plan Q;
    r = build_p1_plan_segs(S, n_e, n_n, elem, opt, opt.max_segs, Q);
    // (and must not lose a table compression the strips have: run-compressed lists are worth more than fewer slots)
    if (r == FEP_OK && Q.lds && Q.pk && (Q.fused || !P.fused) && (Q.rng || !P.rng) && (Q.fused_rng || !P.fused_rng) &&
        Q.staged_total * 100 <= P.staged_total * 95) P = std::move(Q);
    return FEP_OK;
}

// Consistency of a plan with the mesh it was built from (every index the kernels will form stays inside its table or
// LDS region; tiles partition the blocks; every element has exactly one owner): the host test's check, also run by
// fep_ctx_create when FEP_VALIDATE_PLAN is set.  Returns 0 or the number of the first failed check.
inline int validate_p1_plan(const P1Plan& P, const Symbolic& S, int64_t n_e, int64_t n_n, const int32_t* elem) {
    const int64_t n_blk = (int64_t)S.ncol.size();
    if ((int64_t)P.tdesc.size() != P.n_wg * kDescInts) return 1;
    std::vector<uint8_t> seen((size_t)n_blk, 0);
    std::vector<int> owners((size_t)n_e, 0);
    int64_t pk_base = 0;
    for (int64_t g = 0; g < P.n_wg; ++g) {
        const int32_t* d = P.tdesc.data() + g * kDescInts;
        const int nseg = d[3] & 15, n_staged = (d[3] >> 4) & 4095, n_staged_nodes = (d[3] >> 16) & 4095;
        const int nn_desc = d[2] & 255, n_own = d[2] >> 8;                           // (n_own is packed in only by the one-kernel plan)
        if (d[0] != pk_base || nseg < 1 || nseg > kSegMax || d[1] > P.tile || d[1] < 1 || n_own > n_staged) return 2;
        int nb = 0, nn = 0;
        for (int s = 0; s < nseg; ++s) {
            const int32_t fb = d[4 + 4 * s], sb = d[5 + 4 * s], fn = d[6 + 4 * s], sn = d[7 + 4 * s];
            if (fn < 0 || fn + sn > n_n || sn < 1 || S.nptr[fn] != fb || S.nptr[fn + sn] != fb + sb) return 3;
            for (int32_t b = fb; b < fb + sb; ++b) { if (seen[b]) return 4; seen[b] = 1; }
            nb += sb; nn += sn;
        }
        if (nb != d[1] || nn != nn_desc) return 5;
        pk_base += nb;
        if (!P.lds) continue;
        const int32_t* el = P.elist_pad.data() + g * P.L;
        const int n_list = n_staged;
        if (n_list < 1 || n_list > P.L) return 18;
        for (int i = 0; i < P.L; ++i) {
            if (el[i] < 0 || el[i] >= n_e) return 6;
            if (i >= n_list && el[i] != el[0]) return 18;                            // padding repeats the first element
        }
        {   // distinct; with the one-kernel plan: owned elements first, each part ascending
            std::vector<int32_t> srt(el, el + n_list);
            std::sort(srt.begin(), srt.end());
            if (std::adjacent_find(srt.begin(), srt.end()) != srt.end()) return 18;
            if (P.fused)
                for (int i = 1; i < n_list; ++i)
                    if (i != n_own && el[i] <= el[i - 1]) return 18;
        }
        if (P.rng) {
            const int32_t* r = P.rng_tab.data() + g * 16;
            for (int i = 0; i < P.L; ++i) {
                int e = r[0] + i;
                for (int k = 1; k < 8; ++k) e = i >= r[2 * k - 1] ? r[2 * k] + (i - r[2 * k - 1]) : e;
                e = i < r[15] ? e : r[0];
                if (e != el[i]) return 7;
            }
        }
        if (P.pk) {
            std::vector<uint8_t> lane_blk((size_t)nb, 0);
            for (int lane = 0; lane < nb; ++lane) {
                const U2 w = P.pkv[(size_t)d[0] + lane];
                const int beg = (int)(w.x & 2047u), len = (int)((w.x >> 11) & 15u), deg = (int)((w.x >> 15) & 255u);
                const int slot = (int)((w.x >> 23) & 255u), blk = (int)(w.y & 255u), nod = (int)((w.y >> 8) & 255u);
                if (beg + len > P.C || blk >= nb || nod >= nn || lane_blk[blk]) return 8;
                lane_blk[blk] = 1;
                const int rel = 4 * blk - 2 * slot;
                if (rel < 0 || (rel >> 1) + deg + 1 > 2 * nb || slot >= deg) return 9;
                for (int c = beg; c < beg + len; ++c) {
                    const unsigned code = P.codes_pad[(size_t)g * P.C + c];
                    if ((int)(code >> 4) >= n_list || ((code >> 2) & 3) > 2 || (code & 3) > 2) return 10;
                }
            }
        }
        if (P.fused) {
            const int32_t* nl = P.nlist_pad.data() + g * P.NL;
            int n_nl = 1;
            for (int i = 0; i < P.NL; ++i) {
                if (nl[i] < 0 || nl[i] >= n_n) return 11;
                if (i > 0 && nl[i] > nl[i - 1]) n_nl = i + 1;
            }
            if (n_nl != n_staged_nodes) return 19;
            if (P.fused_rng) {
                const int32_t* r = P.nrng_tab.data() + g * 16;
                for (int i = 0; i < P.NL; ++i) {
                    int n = r[0] + i;
                    for (int k = 1; k < 8; ++k) n = i >= r[2 * k - 1] ? r[2 * k] + (i - r[2 * k - 1]) : n;
                    n = i < r[15] ? n : r[0];
                    if (n != nl[i]) return 12;
                }
            }
            for (int i = 0; i < P.L; ++i) {
                const uint32_t w = P.elnodes[(size_t)g * P.L + i];
                for (int a = 0; a < 3; ++a) {
                    const int loc = (int)((w >> (10 * a)) & 1023u);
                    if (loc >= P.NL || nl[loc] != elem[(int64_t)a * n_e + el[i]]) return 13;
                }
                if (((w >> 30) & 1u) != (i < n_own ? 1u : 0u)) return 14;                  // owner bit <=> slot < n_own
                if (i < n_own) owners[el[i]]++;
            }
        }
    }
    if (pk_base != n_blk) return 15;
    for (int64_t b = 0; b < n_blk; ++b) if (!seen[b]) return 16;
    if (P.fused) for (int64_t e = 0; e < n_e; ++e) if (owners[e] != 1) return 17;
    return 0;
}

// ---------------------------------------------------------------------------------------
// Node route of P2 / Q1 / Q2 (ablation build only, FEP_GEN_PATH=node): per tile of `tile` consecutive blocks the sorted unique
// element list and 16-bit gather codes (local element << 8 | a << 4 | b).
// ---------------------------------------------------------------------------------------
struct GnPlan {
    bool ok = false;
    int tile = 256, L = 0, C = 0;
    size_t lds = 0;
    std::vector<int32_t> elist_pad;
    std::vector<uint16_t> codes_pad;
};

inline void build_gn_plan(const Symbolic& S, int n_p, int n_q, int64_t n_e, GnPlan& G) {
    const int64_t n_blk = (int64_t)S.ncol.size();
    for (int TILE : {256, 128}) {
        const int64_t n_wg = (n_blk + TILE - 1) / TILE;
        std::vector<std::vector<int32_t>> lists(n_wg);
        std::vector<uint16_t> perm_l(S.perm.size());
        std::vector<uint8_t> bad((size_t)n_wg, 0);
        parallel_chunks(n_wg, [&](int64_t lo, int64_t hi, int) {
            for (int64_t g = lo; g < hi; ++g) {
                const int64_t b0 = g * TILE, b1 = std::min<int64_t>(n_blk, b0 + TILE);
                const int32_t t0 = S.segptr[b0], t1 = S.segptr[b1];
                std::vector<int32_t>& l = lists[g];
                l.reserve(t1 - t0);
                for (int32_t t = t0; t < t1; ++t) l.push_back((int32_t)(S.perm[t] % n_e));
                std::sort(l.begin(), l.end());
                l.erase(std::unique(l.begin(), l.end()), l.end());
                if (l.size() > 256) { bad[g] = 1; continue; }
                for (int32_t t = t0; t < t1; ++t) {
                    const int64_t ab = S.perm[t] / n_e, e = S.perm[t] % n_e;
                    const int32_t loc = (int32_t)(std::lower_bound(l.begin(), l.end(), (int32_t)e) - l.begin());
                    perm_l[t] = (uint16_t)((loc << 8) | ((int)(ab / n_p) << 4) | (int)(ab % n_p));
                }
            }
        });
        size_t lmax = 0, cmax = 0;
        bool ok = true;
        for (int64_t g = 0; g < n_wg; ++g) {
            ok = ok && !bad[g];
            lmax = std::max(lmax, lists[g].size());
            const int64_t b0 = g * TILE, b1 = std::min<int64_t>(n_blk, b0 + TILE);
            cmax = std::max(cmax, (size_t)(S.segptr[b1] - S.segptr[b0]));
        }
        const int L = (int)lmax, C = (int)((cmax + 7) & ~(size_t)7);
        const size_t lds = ((size_t)(9 + 2 * n_p) * L * n_q + 2 * (size_t)n_p * n_q + n_q + (n_q & 1)) * sizeof(double) +
                           (size_t)C * sizeof(uint16_t);
        if (!ok || lmax > 256 || lds > 64 * 1024) continue;
        G.ok = true; G.tile = TILE; G.L = L; G.C = C; G.lds = lds;
        G.elist_pad.assign((size_t)(n_wg * L), 0);
        G.codes_pad.assign((size_t)(n_wg * C), 0);
        for (int64_t g = 0; g < n_wg; ++g) {
            std::fill(G.elist_pad.begin() + g * L, G.elist_pad.begin() + (g + 1) * L, lists[g].empty() ? 0 : lists[g][0]);
            std::copy(lists[g].begin(), lists[g].end(), G.elist_pad.begin() + g * L);
            const int64_t b0 = g * TILE, b1 = std::min<int64_t>(n_blk, b0 + TILE);
            std::copy(perm_l.begin() + S.segptr[b0], perm_l.begin() + S.segptr[b1], G.codes_pad.begin() + g * C);
        }
        return;
    }
}

// ---------------------------------------------------------------------------------------
// Patch plan of the element route (P2 / Q1 / Q2 / P4, and P1 with FEP_ROUTE=patch): the element matrices never make
// the round trip through HBM.
//
// A patch = the <= `eb` elements one workgroup of element_kernel processes: a chunk of the elements in the order of a
// Hilbert curve through their centroids (compact in 2-D whatever the caller's numbering: what leaves a patch is
// proportional to its perimeter), listed in ascending element id (`pel`, `pnodes`).  Only the grouping is internal: every
// array the caller sees keeps its own element / point / node order.  The kernel leaves the stored half of every K_e
// (sym_block_index) and the element force pairs in LDS; then one lane per ITEM sums, in ascending element order, the
// contributions of ITS patch to one node-pair block of the CSR pattern:
//   closed item  every contribution of the block lies inside the patch (always true for node pairs that share one
//                element only: 12 of the 36 pairs of a P2 element, 153 of the 225 of a P4 element) -> the sum is the
//                CSR block, written straight to its two rows;
//   open item    the block also gets contributions from other patches (node pairs on a patch-boundary edge) -> the sum
//                is a PARTIAL, written to slot `y` of a side buffer (32 bytes per slot, a patch's slots consecutive);
//                fixup_kernel adds a block's partials in ascending patch order and writes the CSR block.
//                K is symmetric block by block (every stored K_e block serves (a, b) and, transposed, (b, a); same
//                summation order), so only the UPPER open blocks (row node <= column node) have items and partials:
//                fixup_kernel writes the block and its transposed mirror.
// Forces: the same with one item per node of the patch (closed: all incident elements inside the patch).
// Deterministic (fixed order, no atomics); the association differs from the flat element order of the COO route
// ((a+b+c) + (d+e+f) instead of a+b+c+d+e+f for a vertex on a patch boundary), so the two routes agree to rounding only.
//
//   pdesc   kPatchDescInts int32 per patch: item_off, n_items, code_off, n_codes, fitem_off, n_fitems, fcode_off, n_fcodes
//   pel     eb int32 per patch: its elements, ascending, padded with -1;   pnodes  [patch][a][local element] node ids (0 padded)
//   items   x: code offset in the patch:13 | (count-1):6 | degree of the row node:12 | open:1
//           y: closed: position of the block's first row in double2 units (2*nptr[n] + slot); open: partial slot
//   codes   uint16 (position of the stored block in the LDS image: patch_image_pos) << 1 | transposed
//   fitems  x: code offset:13 | (count-1):6 | open << 31;  y: node | force partial slot;   fcodes  patch_force_pos
//   fix     per upper open block: x = position (as items.y), y = degree | count << 16, z = first partial slot,
//           w = second partial slot (count <= 2) or offset into plist (count > 2: `count` slots, ascending patch)
//   fixT    its mirror: x = position of the transposed block (0xffffffff: diagonal block, no mirror), y = degree of ITS row node
//   ffix    per open node:  x = node, y = count, z / w as above
// ---------------------------------------------------------------------------------------
struct U4 { uint32_t x, y, z, w; };
constexpr int kPatchDescInts = 8;

struct PatchOptions {
    int order = 2;                                      // 0: consecutive elements, 1: Hilbert curve through the centroids,
                                                        // 2: up to `runs` runs of consecutive elements, each the neighbours of the one before
    int runs = 2;
    int align = 1;                                      // runs start at multiples of `align` elements where they can (full 128-byte
                                                        // lines of the point arrays: 16 / gcd(16, n_q) elements)
};

struct PatchPlan {
    bool ok = false;
    int eb = 0, max_items = 0, max_fitems = 0;
    int64_t n_patch = 0, n_open = 0, n_part = 0, n_fopen = 0, n_fpart = 0;
    std::vector<int32_t> pdesc, plist, pel, pnodes;
    std::vector<U2> items, fitems;
    std::vector<uint16_t> codes, fcodes;
    std::vector<U4> fix, ffix;
    std::vector<U2> fixT;
};

// Where the phase-3 LDS image keeps the stored block (j, a) of local element el, in 16-byte slots of a plane (fep_kernels.hip.h,
// ElemCfg, uses the same formulas).  SKEWED: with the plain (j*n_p + a)*eb + el a step in a is eb slots and a step in j
// n_p*eb slots — both multiples of 16 slots = the 256-byte bank row for every element type (eb is a multiple of 8, n_p*eb of
// 16), so the gather of a CSR row's blocks, which walks j and a at fixed el, put all 16 lanes of a `ds_read_b128` group on ONE
// slot (P4: 49 % of the LDS cycles were conflict cycles).  Odd element stride ebp = eb | 1, and one more slot per j when n_p
// is even, make both steps odd.
inline int patch_ebp(int eb) { return eb | 1; }
inline int patch_image_period(int n_p, int eb) { return n_p * patch_ebp(eb) + ((n_p & 1) ? 0 : 1); }      // slots per j
inline int patch_image_pos(int n_p, int eb, int j, int a, int el) { return j * patch_image_period(n_p, eb) + a * patch_ebp(eb) + el; }
inline int patch_image_slots(int n_p, int eb) { return (n_p / 2 + 1) * patch_image_period(n_p, eb); }
inline int patch_force_pos(int eb, int a, int el) { return a * patch_ebp(eb) + el; }

// where element_kernel keeps the block (a, b) of K_e (same function as fep::sym_block_index, block-major numbering)
inline void patch_block_index(int n_p, int a, int b, int& idx, bool& transposed) {
    const int j = b >= a ? b - a : b - a + n_p;
    const bool direct = 2 * j < n_p || (2 * j == n_p && a < n_p / 2);
    if (direct) { idx = j * n_p + a; transposed = false; }
    else { idx = (n_p - j) * n_p + b; transposed = true; }
}

// position of (x, y), 16 bits each, along the Hilbert curve of order 16
inline uint32_t hilbert_d(uint32_t x, uint32_t y) {
    uint32_t d = 0;
    for (uint32_t s = 1u << 15; s > 0; s >>= 1) {
        const uint32_t rx = (x & s) ? 1u : 0u, ry = (y & s) ? 1u : 0u;
        d += s * s * ((3u * rx) ^ ry);
        if (ry == 0) {
            if (rx == 1) { x = 65535u - x; y = 65535u - y; }
            const uint32_t t = x; x = y; y = t;
        }
    }
    return d;
}

// Patches of up to `runs` RUNS of consecutive element ids (what keeps the point arrays' accesses in long contiguous
// pieces): the first run starts at the lowest unassigned element; every further one is the longest run of consecutive,
// still unassigned ids among the elements that share a node with the run before (on a row-numbered mesh: the same cells
// one row up), so that a patch is `runs` rows high instead of one — its perimeter, hence what it exchanges with its
// neighbours, shrinks accordingly.  A patch that stays short is topped up from its own unassigned neighbours.
inline void patch_runs(const Symbolic& S, int n_p, int64_t n_e, const int32_t* elem, int eb, int runs, int align,
                       std::vector<int32_t>& pel, std::vector<int32_t>& patch_of) {
    align = std::max(1, align);
    runs = std::max(1, std::min(runs, eb));
    pel.clear();
    patch_of.assign((size_t)n_e, -1);
    std::vector<int32_t> cur, nbrs;
    int64_t cursor = 0;
    int32_t p = 0;
    while (true) {
        while (cursor < n_e && patch_of[cursor] >= 0) ++cursor;
        if (cursor >= n_e) break;
        cur.clear();
        int64_t start = cursor;
        for (int r = 0; r < runs && (int)cur.size() < eb; ++r) {
            const int budget = (eb - (int)cur.size() + (runs - r) - 1) / (runs - r);     // an even share of what is left
            const size_t run0 = cur.size();
            // a run is consecutive ids that are also neighbours in the mesh: it ends where element e shares no node with e - 1
            // (the end of a row of a row-numbered mesh — running on into the next row shears every patch of the band behind it)
            for (int64_t e = start; e < n_e && patch_of[e] < 0 && (int)(cur.size() - run0) < budget; ++e) {
                if (e > start) {
                    bool touch = false;
                    for (int a = 0; a < n_p && !touch; ++a)
                        for (int b = 0; b < n_p && !touch; ++b) touch = elem[(int64_t)a * n_e + e] == elem[(int64_t)b * n_e + e - 1];
                    if (!touch) break;
                }
                cur.push_back((int32_t)e); patch_of[e] = p;
            }
            if (r + 1 == runs || (int)cur.size() >= eb) break;
            nbrs.clear();
            for (size_t i = run0; i < cur.size(); ++i)
                for (int a = 0; a < n_p; ++a) {
                    const int32_t nd = elem[(int64_t)a * n_e + cur[i]];
                    for (int32_t t = S.iptr[nd]; t < S.iptr[nd + 1]; ++t) {
                        const int32_t e2 = (int32_t)((int64_t)S.ilist[t] % n_e);
                        if (patch_of[e2] < 0) nbrs.push_back(e2);
                    }
                }
            std::sort(nbrs.begin(), nbrs.end());
            nbrs.erase(std::unique(nbrs.begin(), nbrs.end()), nbrs.end());
            size_t best = 0, best_len = 0;
            for (size_t i = 0; i < nbrs.size();) {
                size_t j = i + 1;
                while (j < nbrs.size() && nbrs[j] == nbrs[j - 1] + 1) ++j;
                if (j - i > best_len) { best_len = j - i; best = i; }
                i = j;
            }
            if (best_len < 2) break;
            start = nbrs[best];
            if (align > 1 && start % align) {           // a start on a line boundary of the point arrays: down if those ids are free, else up
                const int64_t lo = start - start % align;
                bool free_lo = true;
                for (int64_t e = lo; e < start && free_lo; ++e) free_lo = patch_of[e] < 0;
                if (free_lo) start = lo;
                else if (lo + align < (int64_t)nbrs[best] + (int64_t)best_len) start = lo + align;
            }
        }
        if ((int)cur.size() < eb) {                     // top up, with unassigned NEIGHBOURS of the patch only (lowest ids first): a patch
            nbrs.clear();                               // that stays short costs a partly idle workgroup, one that collects far-away
            for (size_t i = 0; i < cur.size(); ++i)     // elements costs every patch around them
                for (int a = 0; a < n_p; ++a) {
                    const int32_t nd = elem[(int64_t)a * n_e + cur[i]];
                    for (int32_t t = S.iptr[nd]; t < S.iptr[nd + 1]; ++t) {
                        const int32_t e2 = (int32_t)((int64_t)S.ilist[t] % n_e);
                        if (patch_of[e2] < 0) nbrs.push_back(e2);
                    }
                }
            std::sort(nbrs.begin(), nbrs.end());
            nbrs.erase(std::unique(nbrs.begin(), nbrs.end()), nbrs.end());
            for (size_t i = 0; i < nbrs.size() && (int)cur.size() < eb; ++i) { cur.push_back(nbrs[i]); patch_of[nbrs[i]] = p; }
        }
        std::sort(cur.begin(), cur.end());
        pel.insert(pel.end(), cur.begin(), cur.end());
        pel.resize((size_t)(p + 1) * eb, -1);
        ++p;
    }
}

// pel / patch_of for the chosen grouping (coords: planar x[n_n], y[n_n]; may be NULL -> consecutive elements)
inline void patch_grouping(const Symbolic& S, int n_p, int64_t n_e, int64_t n_n, const int32_t* elem, const double* coords, int eb,
                           const PatchOptions& opt, std::vector<int32_t>& pel, std::vector<int32_t>& patch_of) {
    const int order = opt.order;
    if (order == 2 && n_e > eb) { patch_runs(S, n_p, n_e, elem, eb, opt.runs, opt.align, pel, patch_of); return; }
    const int64_t n_patch = (n_e + eb - 1) / eb;
    pel.assign((size_t)(n_patch * eb), -1);
    patch_of.resize((size_t)n_e);
    std::vector<int32_t> seq((size_t)n_e);
    for (int64_t e = 0; e < n_e; ++e) seq[e] = (int32_t)e;
    if (order == 1 && coords && n_e > eb) {
        double lo[2] = {coords[0], coords[n_n]}, hi[2] = {coords[0], coords[n_n]};
        for (int64_t n = 0; n < n_n; ++n)
            for (int k = 0; k < 2; ++k) { const double v = coords[k * n_n + n]; lo[k] = std::min(lo[k], v); hi[k] = std::max(hi[k], v); }
        const double ext = std::max(hi[0] - lo[0], hi[1] - lo[1]);
        const double sc = ext > 0 ? 65535.0 / ext : 0.0;                     // one scale for both axes: square cells stay square
        std::vector<uint64_t> key((size_t)n_e);
        parallel_chunks(n_e, [&](int64_t a0, int64_t a1, int) {
            for (int64_t e = a0; e < a1; ++e) {
                double cx = 0.0, cy = 0.0;
                for (int a = 0; a < n_p; ++a) { const int32_t nd = elem[(int64_t)a * n_e + e]; cx += coords[nd]; cy += coords[n_n + nd]; }
                const double qx = (cx / n_p - lo[0]) * sc, qy = (cy / n_p - lo[1]) * sc;
                const uint32_t ix = (uint32_t)std::min(65535.0, std::max(0.0, qx)), iy = (uint32_t)std::min(65535.0, std::max(0.0, qy));
                key[e] = ((uint64_t)hilbert_d(ix, iy) << 32) | (uint32_t)e;
            }
        });
        std::sort(key.begin(), key.end());
        for (int64_t i = 0; i < n_e; ++i) seq[i] = (int32_t)(key[i] & 0xffffffffu);
    }
    for (int64_t p = 0; p < n_patch; ++p) {
        const int64_t i0 = p * eb, i1 = std::min<int64_t>(n_e, i0 + eb);
        std::sort(seq.begin() + i0, seq.begin() + i1);                        // a patch lists its elements in ascending id
        for (int64_t i = i0; i < i1; ++i) { pel[(size_t)i] = seq[i]; patch_of[seq[i]] = (int32_t)p; }
    }
}

inline int build_patch_plan(const Symbolic& S, int n_p, int64_t n_e, int64_t n_n, const int32_t* elem, const double* coords,
                            int eb, const PatchOptions& opt, PatchPlan& P) {
    P = PatchPlan();
    P.eb = eb;
    const int nj = n_p / 2 + 1;
    if (eb < 1 || (int64_t)n_p * n_p * eb > 8192 || patch_image_slots(n_p, eb) >= 32768) return FEP_OK;       // field widths: plan not usable
    const int64_t n_blk = (int64_t)S.ncol.size();
    std::vector<int32_t> patch_of;
    patch_grouping(S, n_p, n_e, n_n, elem, coords, eb, opt, P.pel, patch_of);
    const int64_t n_patch = (int64_t)P.pel.size() / eb;
    P.n_patch = n_patch;
    P.pnodes.assign((size_t)(n_patch * n_p * eb), 0);
    parallel_chunks(n_patch, [&](int64_t lo, int64_t hi, int) {
        for (int64_t p = lo; p < hi; ++p)
            for (int el = 0; el < eb; ++el) {
                const int32_t e = P.pel[(size_t)(p * eb + el)];
                if (e < 0) continue;
                for (int a = 0; a < n_p; ++a) P.pnodes[(size_t)((p * n_p + a) * eb + el)] = elem[(int64_t)a * n_e + e];
            }
    });
    // pass A: how many patches contribute to every node / block
    std::vector<uint8_t> npb((size_t)n_blk, 0), npn((size_t)n_n, 0);
    std::atomic<int> bad{0};
    auto distinct = [&](const int32_t* list, int32_t beg, int32_t end, std::vector<int32_t>& buf) {
        buf.clear();
        for (int32_t t = beg; t < end; ++t) buf.push_back(patch_of[(size_t)((int64_t)list[t] % n_e)]);
        std::sort(buf.begin(), buf.end());
        buf.erase(std::unique(buf.begin(), buf.end()), buf.end());
    };
    parallel_chunks(n_n, [&](int64_t lo, int64_t hi, int) {
        std::vector<int32_t> buf;
        for (int64_t n = lo; n < hi; ++n) {
            distinct(S.ilist.data(), S.iptr[n], S.iptr[n + 1], buf);
            if (buf.size() > 255) { bad.store(1, std::memory_order_relaxed); buf.resize(255); }
            npn[n] = (uint8_t)buf.size();
            for (int32_t b = S.nptr[n]; b < S.nptr[n + 1]; ++b) {
                if (npn[n] < 2) { npb[b] = 1; continue; }                     // all elements of the row node in one patch
                distinct(S.perm.data(), S.segptr[b], S.segptr[b + 1], buf);
                npb[b] = (uint8_t)std::min<size_t>(buf.size(), 255);
            }
        }
    });
    if (bad.load()) return FEP_OK;
    // open blocks / nodes: index, table entries, plist offsets
    // open_idx: -1 closed, -2 lower open block (row node > column node: written by its mirror's fix-up lane), else index into fix
    std::vector<int32_t> open_idx((size_t)n_blk, -1), fopen_idx((size_t)n_n, -1);
    int64_t n_plist = 0;
    for (int64_t n = 0; n < n_n; ++n) {
        const uint32_t deg = (uint32_t)(S.nptr[n + 1] - S.nptr[n]);
        if (deg > 4095) return FEP_OK;
        for (int32_t b = S.nptr[n]; b < S.nptr[n + 1]; ++b) {
            if (npb[b] < 2) continue;
            const int64_t m = S.ncol[b];
            if (m < n) { open_idx[b] = -2; continue; }
            open_idx[b] = (int32_t)P.fix.size();
            U4 f{(uint32_t)(2 * (int64_t)S.nptr[n] + (b - S.nptr[n])), deg | ((uint32_t)npb[b] << 16), 0u, 0u};
            if (npb[b] > 2) { f.w = (uint32_t)n_plist; n_plist += npb[b]; }
            P.fix.push_back(f);
            U2 ft{0xffffffffu, 0u};
            if (m != n) {                                                    // the mirror block (m, n)
                const int32_t* row = S.ncol.data() + S.nptr[m];
                const int64_t slot = std::lower_bound(row, S.ncol.data() + S.nptr[m + 1], (int32_t)n) - row;
                ft = U2{(uint32_t)(2 * (int64_t)S.nptr[m] + slot), (uint32_t)(S.nptr[m + 1] - S.nptr[m])};
            }
            P.fixT.push_back(ft);
        }
        if (npn[n] >= 2) {
            fopen_idx[n] = (int32_t)P.ffix.size();
            U4 f{(uint32_t)n, (uint32_t)npn[n], 0u, 0u};
            if (npn[n] > 2) { f.w = (uint32_t)n_plist; n_plist += npn[n]; }
            P.ffix.push_back(f);
        }
        if (n_plist >= INT32_MAX / 2) return FEP_ERANGE;
    }
    P.n_open = (int64_t)P.fix.size();
    P.n_fopen = (int64_t)P.ffix.size();
    P.plist.assign((size_t)n_plist, -1);
    // pass B: the patches' items, worker-local, then concatenated
    struct Out {
        std::vector<int32_t> pdesc;                      // local offsets
        std::vector<U2> items, fitems;
        std::vector<uint16_t> codes, fcodes;
        std::vector<int32_t> open_b, open_p, fopen_n, fopen_p;   // per local partial slot: its block / node and patch
        int max_items = 0, max_fitems = 0, fail = 0;
    };
    const int nw = (int)std::max<int64_t>(1, std::min<int64_t>(worker_count(), n_patch));
    std::vector<Out> outs((size_t)nw);
    parallel_chunks(n_patch, [&](int64_t lo, int64_t hi, int w) {
        Out& O = outs[(size_t)w];
        struct Tup { int32_t key; uint16_t code; };
        std::vector<Tup> tup, ftup;
        std::vector<U2> open_items;
        std::vector<uint16_t> open_codes;
        std::vector<int32_t> open_keys;
        for (int64_t p = lo; p < hi; ++p) {
            const int32_t* pe = P.pel.data() + p * eb;
            int nel = 0;
            while (nel < eb && pe[nel] >= 0) ++nel;
            tup.clear(); ftup.clear();
            for (int el = 0; el < nel; ++el)
                for (int a = 0; a < n_p; ++a) {
                    const int32_t na = elem[(int64_t)a * n_e + pe[el]];
                    ftup.push_back(Tup{na, (uint16_t)patch_force_pos(eb, a, el)});
                    const int32_t* row = S.ncol.data() + S.nptr[na];
                    const int32_t* row_end = S.ncol.data() + S.nptr[na + 1];
                    for (int b = 0; b < n_p; ++b) {
                        const int32_t nb = elem[(int64_t)b * n_e + pe[el]];
                        const int32_t blk = (int32_t)(S.nptr[na] + (std::lower_bound(row, row_end, nb) - row));
                        int idx; bool tr;
                        patch_block_index(n_p, a, b, idx, tr);
                        tup.push_back(Tup{blk, (uint16_t)((patch_image_pos(n_p, eb, idx / n_p, idx % n_p, el) << 1) | (tr ? 1 : 0))});
                    }
                }
            // (el, a, b) generation order = ascending element id: the stable sorts keep it inside every block / node
            std::stable_sort(tup.begin(), tup.end(), [](const Tup& x, const Tup& y) { return x.key < y.key; });
            std::stable_sort(ftup.begin(), ftup.end(), [](const Tup& x, const Tup& y) { return x.key < y.key; });
            const size_t item0 = O.items.size(), code0 = O.codes.size(), fitem0 = O.fitems.size(), fcode0 = O.fcodes.size();
            open_items.clear(); open_codes.clear(); open_keys.clear();
            // closed items first (their codes too), open items behind them; both in ascending block order
            for (size_t i = 0; i < tup.size();) {
                size_t j = i + 1;
                while (j < tup.size() && tup[j].key == tup[i].key) ++j;
                const int32_t b = tup[i].key;
                const uint32_t cnt = (uint32_t)(j - i);
                if (cnt > 64) { O.fail = 1; break; }
                const bool all_here = (int32_t)cnt == S.segptr[b + 1] - S.segptr[b];
                if ((npb[b] < 2) != all_here) { O.fail = 2; break; }
                if (open_idx[b] == -2) { i = j; continue; }                 // lower open block: its mirror carries the partial
                if (open_idx[b] == -1) {
                    const uint32_t deg = S.meta[b] >> 16, slot = S.meta[b] & 0x7fffu;
                    O.items.push_back(U2{(uint32_t)(O.codes.size() - code0) | ((cnt - 1) << 13) | (deg << 19),
                                         (uint32_t)(2 * (int64_t)(b - (int32_t)slot) + slot)});
                    for (size_t k = i; k < j; ++k) O.codes.push_back(tup[k].code);
                } else {
                    open_items.push_back(U2{(uint32_t)open_codes.size() | ((cnt - 1) << 13) | (1u << 31), 0u});
                    for (size_t k = i; k < j; ++k) open_codes.push_back(tup[k].code);
                    open_keys.push_back(b);
                }
                i = j;
            }
            if (O.fail) break;
            const uint32_t shift = (uint32_t)(O.codes.size() - code0);
            for (size_t i = 0; i < open_items.size(); ++i) {
                U2 it = open_items[i];
                it.x += shift;                                               // code offset: behind the closed items' codes
                it.y = (uint32_t)O.open_b.size();                            // worker-local slot, rebased after the join
                O.items.push_back(it);
                O.open_b.push_back(open_keys[i]);
                O.open_p.push_back((int32_t)p);
            }
            O.codes.insert(O.codes.end(), open_codes.begin(), open_codes.end());
            const int32_t n_codes_p = (int32_t)(O.codes.size() - code0);
            if (O.codes.size() & 1) O.codes.push_back(0);                    // every patch's codes start at an even offset (32-bit loads)
            for (size_t i = 0; i < ftup.size();) {
                size_t j = i + 1;
                while (j < ftup.size() && ftup[j].key == ftup[i].key) ++j;
                const int32_t n = ftup[i].key;
                const uint32_t cnt = (uint32_t)(j - i);
                if (cnt > 64) { O.fail = 1; break; }
                const bool closed = (int32_t)cnt == S.iptr[n + 1] - S.iptr[n];
                if ((npn[n] < 2) != closed) { O.fail = 2; break; }
                uint32_t y = (uint32_t)n;
                if (!closed) { y = (uint32_t)O.fopen_n.size(); O.fopen_n.push_back(n); O.fopen_p.push_back((int32_t)p); }
                O.fitems.push_back(U2{(uint32_t)(O.fcodes.size() - fcode0) | ((cnt - 1) << 13) | (closed ? 0u : 1u << 31), y});
                for (size_t k = i; k < j; ++k) O.fcodes.push_back(ftup[k].code);
                i = j;
            }
            if (O.fail) break;
            const int32_t n_fcodes_p = (int32_t)(O.fcodes.size() - fcode0);
            if (O.fcodes.size() & 1) O.fcodes.push_back(0);
            const int32_t d[kPatchDescInts] = {(int32_t)item0, (int32_t)(O.items.size() - item0), (int32_t)code0, n_codes_p,
                                               (int32_t)fitem0, (int32_t)(O.fitems.size() - fitem0), (int32_t)fcode0, n_fcodes_p};
            O.pdesc.insert(O.pdesc.end(), d, d + kPatchDescInts);
            O.max_items = std::max(O.max_items, d[1]);
            O.max_fitems = std::max(O.max_fitems, d[5]);
            if (O.items.size() >= (size_t)INT32_MAX / 2 || O.codes.size() >= (size_t)INT32_MAX / 2) { O.fail = 3; break; }
        }
    });
    // bases of every worker's pieces
    std::vector<int64_t> bi(nw + 1, 0), bc(nw + 1, 0), bfi(nw + 1, 0), bfc(nw + 1, 0), bs(nw + 1, 0), bfs(nw + 1, 0);
    for (int w = 0; w < nw; ++w) {
        const Out& O = outs[(size_t)w];
        if (O.fail == 3) return FEP_ERANGE;
        if (O.fail == 2) return FEP_EINVAL;              // inconsistent with the symbolic phase: a bug, not a property of the mesh
        if (O.fail) return FEP_OK;                       // a field does not fit: no plan, the caller keeps the COO route
        bi[w + 1] = bi[w] + (int64_t)O.items.size(); bc[w + 1] = bc[w] + (int64_t)O.codes.size();
        bfi[w + 1] = bfi[w] + (int64_t)O.fitems.size(); bfc[w + 1] = bfc[w] + (int64_t)O.fcodes.size();
        bs[w + 1] = bs[w] + (int64_t)O.open_b.size(); bfs[w + 1] = bfs[w] + (int64_t)O.fopen_n.size();
        P.max_items = std::max(P.max_items, O.max_items);
        P.max_fitems = std::max(P.max_fitems, O.max_fitems);
    }
    if (bi[nw] >= INT32_MAX / 2 || bc[nw] >= INT32_MAX / 2 || bs[nw] >= INT32_MAX / 4) return FEP_ERANGE;
    P.items.resize((size_t)bi[nw]); P.codes.resize((size_t)bc[nw] + 8, 0);
    P.fitems.resize((size_t)bfi[nw]); P.fcodes.resize((size_t)bfc[nw] + 8, 0);
    P.pdesc.resize((size_t)n_patch * kPatchDescInts);
    P.n_part = bs[nw]; P.n_fpart = bfs[nw];
    std::vector<std::thread> th;
    auto finish = [&](int w) {
        Out& O = outs[(size_t)w];
        for (U2& it : O.items) if (it.x >> 31) it.y += (uint32_t)bs[w];
        for (U2& it : O.fitems) if (it.x >> 31) it.y += (uint32_t)bfs[w];
        std::copy(O.items.begin(), O.items.end(), P.items.begin() + bi[w]);
        std::copy(O.codes.begin(), O.codes.end(), P.codes.begin() + bc[w]);
        std::copy(O.fitems.begin(), O.fitems.end(), P.fitems.begin() + bfi[w]);
        std::copy(O.fcodes.begin(), O.fcodes.end(), P.fcodes.begin() + bfc[w]);
        const int64_t p0 = n_patch * w / nw;
        for (size_t i = 0; i < O.pdesc.size(); i += kPatchDescInts) {
            int32_t* d = P.pdesc.data() + (p0 * kPatchDescInts + (int64_t)i);
            std::copy(O.pdesc.begin() + i, O.pdesc.begin() + i + kPatchDescInts, d);
            d[0] += (int32_t)bi[w]; d[2] += (int32_t)bc[w]; d[4] += (int32_t)bfi[w]; d[6] += (int32_t)bfc[w];
        }
        // the fix-up tables: slot of patch p in its block's / node's list = rank of p among the contributing patches
        std::vector<int32_t> buf;
        for (size_t s = 0; s < O.open_b.size(); ++s) {
            const int32_t b = O.open_b[s], p = O.open_p[s];
            distinct(S.perm.data(), S.segptr[b], S.segptr[b + 1], buf);
            const int rank = (int)(std::lower_bound(buf.begin(), buf.end(), p) - buf.begin());
            U4& f = P.fix[(size_t)open_idx[b]];
            const uint32_t slot = (uint32_t)(bs[w] + (int64_t)s);
            if ((f.y >> 16) > 2) { P.plist[(size_t)f.w + rank] = (int32_t)slot; if (rank == 0) f.z = slot; }
            else if (rank == 0) f.z = slot; else f.w = slot;
        }
        for (size_t s = 0; s < O.fopen_n.size(); ++s) {
            const int32_t n = O.fopen_n[s], p = O.fopen_p[s];
            distinct(S.ilist.data(), S.iptr[n], S.iptr[n + 1], buf);
            const int rank = (int)(std::lower_bound(buf.begin(), buf.end(), p) - buf.begin());
            U4& f = P.ffix[(size_t)fopen_idx[n]];
            const uint32_t slot = (uint32_t)(bfs[w] + (int64_t)s);
            if (f.y > 2) { P.plist[(size_t)f.w + rank] = (int32_t)slot; if (rank == 0) f.z = slot; }
            else if (rank == 0) f.z = slot; else f.w = slot;
        }
    };
    // (two workers never touch the same fix entry field: a block's partials come from different patches, each patch
    // belongs to one worker and writes its own rank)
    std::vector<std::exception_ptr> ferr((size_t)nw);
    auto finish_guarded = [&](int w) { try { finish(w); } catch (...) { ferr[(size_t)w] = std::current_exception(); } };
    for (int w = 1; w < nw; ++w) {
        try { th.emplace_back(finish_guarded, w); } catch (const std::system_error&) { finish_guarded(w); }
    }
    finish_guarded(0);
    for (auto& t : th) t.join();
    for (const std::exception_ptr& e : ferr)
        if (e) std::rethrow_exception(e);
    P.ok = true;
    return FEP_OK;
}

// Consistency of a patch plan with the symbolic phase: replays every block's / node's sum from the plan (closed items,
// or the partials of its fix-up entry in list order) and compares the (element, a, b) contributions with
// Symbolic::perm / ilist (same multiset; ascending element inside every item); every CSR block, node, partial slot is
// produced exactly once; every element belongs to exactly one patch; every index the kernels form stays inside its
// table or LDS region.  Returns 0 or the number of the first failed check.
inline int validate_patch_plan(const PatchPlan& P, const Symbolic& S, int n_p, int64_t n_e, int64_t n_n, const int32_t* elem) {
    if (!P.ok) return 0;
    const int eb = P.eb, nj = n_p / 2 + 1;
    const int64_t n_blk = (int64_t)S.ncol.size();
    if ((int64_t)P.pdesc.size() != P.n_patch * kPatchDescInts || P.n_patch < (n_e + eb - 1) / eb) return 1;
    if ((int64_t)P.pel.size() != P.n_patch * eb || (int64_t)P.pnodes.size() != P.n_patch * n_p * eb || P.fixT.size() != P.fix.size()) return 1;
    {
        std::vector<uint8_t> owned((size_t)n_e, 0);
        for (int64_t p = 0; p < P.n_patch; ++p)
            for (int el = 0; el < eb; ++el) {
                const int32_t e = P.pel[(size_t)(p * eb + el)];
                if (e < 0) { for (int k = el; k < eb; ++k) if (P.pel[(size_t)(p * eb + k)] >= 0) return 35; break; }
                if (e >= n_e || owned[e] || (el > 0 && e <= P.pel[(size_t)(p * eb + el - 1)])) return 35;
                owned[e] = 1;
                for (int a = 0; a < n_p; ++a) if (P.pnodes[(size_t)((p * n_p + a) * eb + el)] != elem[(int64_t)a * n_e + e]) return 36;
            }
        for (uint8_t v : owned) if (!v) return 37;
        for (int32_t v : P.pnodes) if (v < 0 || v >= n_n) return 36;
    }
    // stored block (idx, transposed) -> (a, b)
    std::vector<int> ab_of((size_t)2 * nj * n_p, -1);
    for (int a = 0; a < n_p; ++a)
        for (int b = 0; b < n_p; ++b) {
            int idx; bool tr;
            patch_block_index(n_p, a, b, idx, tr);
            if (idx < 0 || idx >= nj * n_p) return 2;
            if (a != b || !tr) ab_of[(size_t)2 * idx + (tr ? 1 : 0)] = a * n_p + b;
        }
    std::vector<std::vector<int32_t>> part_seq((size_t)P.n_part), fpart_seq((size_t)P.n_fpart);   // contributions of every partial
    std::vector<uint8_t> part_seen((size_t)P.n_part, 0), fpart_seen((size_t)P.n_fpart, 0);
    std::vector<uint8_t> blk_seen((size_t)n_blk, 0), node_seen((size_t)n_n, 0);
    int64_t total = 0, ftotal = 0;
    auto ascending = [&](const std::vector<int32_t>& q) {                    // by element, then (a, b)
        for (size_t k = 1; k < q.size(); ++k) {
            const int64_t e0 = q[k - 1] % n_e, e1 = q[k] % n_e;
            if (e1 < e0 || (e1 == e0 && q[k] <= q[k - 1])) return false;
        }
        return true;
    };
    auto same_set = [&](std::vector<int32_t> got, const int32_t* ref, size_t n) {
        if (got.size() != n) return false;
        std::vector<int32_t> want(ref, ref + n);
        std::sort(got.begin(), got.end()); std::sort(want.begin(), want.end());
        return got == want;
    };
    for (int64_t p = 0; p < P.n_patch; ++p) {
        const int32_t* d = P.pdesc.data() + p * kPatchDescInts;
        const int32_t* pe = P.pel.data() + p * eb;
        int nel = 0;
        while (nel < eb && pe[nel] >= 0) ++nel;
        if (d[0] < 0 || d[0] + (int64_t)d[1] > (int64_t)P.items.size() || d[2] < 0 || d[2] + (int64_t)d[3] + 8 > (int64_t)P.codes.size()) return 3;
        if (d[4] < 0 || d[4] + (int64_t)d[5] > (int64_t)P.fitems.size() || d[6] < 0 || d[6] + (int64_t)d[7] + 8 > (int64_t)P.fcodes.size()) return 3;
        if (nel < 1 || d[3] > n_p * n_p * nel || d[7] != n_p * nel || d[1] > P.max_items || d[5] > P.max_fitems || (d[2] & 1) || (d[6] & 1)) return 4;
        std::vector<int32_t> seq;
        for (int32_t i = 0; i < d[1]; ++i) {
            const U2 it = P.items[(size_t)d[0] + i];
            const int off = (int)(it.x & 8191u), cnt = (int)((it.x >> 13) & 63u) + 1, deg = (int)((it.x >> 19) & 4095u);
            const bool open = it.x >> 31;
            if (off + cnt > d[3]) return 5;
            seq.clear();
            for (int k = 0; k < cnt; ++k) {
                const unsigned code = P.codes[(size_t)d[2] + off + k];
                const int pos = (int)(code >> 1), per = patch_image_period(n_p, eb), ebp = patch_ebp(eb);
                const int jj = pos / per, rem = pos % per, aa = rem / ebp, el = rem % ebp, idx = jj * n_p + aa;
                if (jj >= nj || aa >= n_p || el >= nel || patch_image_pos(n_p, eb, jj, aa, el) != pos) return 6;
                const int ab = ab_of[(size_t)2 * idx + (code & 1u)];
                if (ab < 0) return 7;
                seq.push_back((int32_t)((int64_t)ab * n_e + pe[el]));
            }
            if (!ascending(seq)) return 38;
            total += cnt;
            if (open) {
                if (it.y >= (uint32_t)P.n_part || part_seen[it.y]) return 8;
                part_seen[it.y] = 1;
                part_seq[it.y] = seq;
            } else {
                // position -> block: row node n with 2*nptr[n] <= y < 2*nptr[n] + deg
                const int64_t y = it.y;
                const int64_t n = std::upper_bound(S.nptr.begin(), S.nptr.end(), (int32_t)(y / 2)) - S.nptr.begin() - 1;
                if (n < 0 || n >= n_n) return 9;
                const int64_t slot = y - 2 * (int64_t)S.nptr[n];
                if (slot < 0 || slot >= S.nptr[n + 1] - S.nptr[n] || deg != S.nptr[n + 1] - S.nptr[n]) return 9;
                const int64_t b = S.nptr[n] + slot;
                if (blk_seen[b]) return 10;
                blk_seen[b] = 1;
                if (!same_set(seq, S.perm.data() + S.segptr[b], (size_t)(S.segptr[b + 1] - S.segptr[b]))) return 12;
            }
        }
        for (int32_t i = 0; i < d[5]; ++i) {
            const U2 it = P.fitems[(size_t)d[4] + i];
            const int off = (int)(it.x & 8191u), cnt = (int)((it.x >> 13) & 63u) + 1;
            const bool open = it.x >> 31;
            if (off + cnt > d[7]) return 13;
            seq.clear();
            for (int k = 0; k < cnt; ++k) {
                const unsigned code = P.fcodes[(size_t)d[6] + off + k];
                const int a = (int)code / patch_ebp(eb), el = (int)code % patch_ebp(eb);
                if (a >= n_p || el >= nel) return 14;
                seq.push_back((int32_t)((int64_t)a * n_e + pe[el]));
            }
            if (!ascending(seq)) return 38;
            ftotal += cnt;
            if (open) {
                if (it.y >= (uint32_t)P.n_fpart || fpart_seen[it.y]) return 15;
                fpart_seen[it.y] = 1;
                fpart_seq[it.y] = seq;
            } else {
                const int64_t n = it.y;
                if (n >= n_n || node_seen[n]) return 16;
                node_seen[n] = 1;
                if (!same_set(seq, S.ilist.data() + S.iptr[n], (size_t)(S.iptr[n + 1] - S.iptr[n]))) return 18;
            }
        }
    }
    if (ftotal != (int64_t)S.ilist.size()) return 19;
    for (uint8_t v : part_seen) if (!v) return 20;
    for (uint8_t v : fpart_seen) if (!v) return 20;
    std::vector<uint8_t> slot_used((size_t)P.n_part, 0), fslot_used((size_t)P.n_fpart, 0);
    auto slots_of = [&](const U4& f, int cnt, std::vector<int64_t>& out) {
        out.clear();
        if (cnt <= 2) { out.push_back(f.z); if (cnt == 2) out.push_back(f.w); return true; }
        if ((int64_t)f.w + cnt > (int64_t)P.plist.size()) return false;
        for (int k = 0; k < cnt; ++k) out.push_back(P.plist[(size_t)f.w + k]);
        return out[0] == (int64_t)f.z;
    };
    std::vector<int64_t> sl;
    std::vector<int32_t> all;
    for (const U4& f : P.fix) {
        const int cnt = (int)(f.y >> 16), deg = (int)(f.y & 0xffffu);
        if (cnt < 1 || !slots_of(f, cnt, sl)) return 21;
        const int64_t n = std::upper_bound(S.nptr.begin(), S.nptr.end(), (int32_t)(f.x / 2)) - S.nptr.begin() - 1;
        if (n < 0 || n >= n_n || deg != S.nptr[n + 1] - S.nptr[n]) return 22;
        const int64_t slot = (int64_t)f.x - 2 * (int64_t)S.nptr[n];
        if (slot < 0 || slot >= deg) return 22;
        const int64_t b = S.nptr[n] + slot;
        if (blk_seen[b]) return 23;
        blk_seen[b] = 1;
        all.clear();
        for (int64_t s : sl) {
            if (s < 0 || s >= P.n_part || slot_used[s]) return 24;
            slot_used[s] = 1;
            all.insert(all.end(), part_seq[(size_t)s].begin(), part_seq[(size_t)s].end());
        }
        if (!same_set(all, S.perm.data() + S.segptr[b], (size_t)(S.segptr[b + 1] - S.segptr[b]))) return 25;
        // the mirror block: same elements, (a, b) swapped
        const U2 ft = P.fixT[(size_t)(&f - P.fix.data())];
        const int64_t m = S.ncol[b];
        if (m == n) { if (ft.x != 0xffffffffu) return 39; continue; }
        if (m < n) return 39;                                                // only upper blocks have entries
        const int64_t slot_t = (int64_t)ft.x - 2 * (int64_t)S.nptr[m];
        if (slot_t < 0 || slot_t >= S.nptr[m + 1] - S.nptr[m] || (int64_t)ft.y != S.nptr[m + 1] - S.nptr[m]) return 40;
        const int64_t bt = S.nptr[m] + slot_t;
        if (S.ncol[bt] != n || blk_seen[bt]) return 41;
        blk_seen[bt] = 1;
        for (int32_t& v : all) { const int64_t ab = v / n_e, e = v % n_e; v = (int32_t)(((ab % n_p) * n_p + ab / n_p) * n_e + e); }
        if (!same_set(all, S.perm.data() + S.segptr[bt], (size_t)(S.segptr[bt + 1] - S.segptr[bt]))) return 42;
        total += (int64_t)all.size();
    }
    if (total != (int64_t)S.perm.size()) return 19;
    for (const U4& f : P.ffix) {
        const int cnt = (int)f.y;
        if (cnt < 2 || !slots_of(f, cnt, sl)) return 27;
        const int64_t n = f.x;
        if (n >= n_n || node_seen[n]) return 28;
        node_seen[n] = 1;
        all.clear();
        for (int64_t s : sl) {
            if (s < 0 || s >= P.n_fpart || fslot_used[s]) return 29;
            fslot_used[s] = 1;
            all.insert(all.end(), fpart_seq[(size_t)s].begin(), fpart_seq[(size_t)s].end());
        }
        if (!same_set(all, S.ilist.data() + S.iptr[n], (size_t)(S.iptr[n + 1] - S.iptr[n]))) return 30;
    }
    for (int64_t b = 0; b < n_blk; ++b) if (!blk_seen[b]) return 32;
    for (int64_t n = 0; n < n_n; ++n) if (!node_seen[n] && S.iptr[n + 1] > S.iptr[n]) return 33;
    for (uint8_t v : slot_used) if (!v) return 34;
    for (uint8_t v : fslot_used) if (!v) return 34;
    return 0;
}

// Greedy aggregation of a node graph in CSR (smoothed-aggregation multigrid setup): agg_out[i] in [0, *n_agg_out).
// A node's neighbour list may hold an id more than once and in any order (solver.py passes the rows of a node's DOFs back to back).
inline int aggregate(int64_t n, const int32_t* indptr, const int32_t* indices, int32_t* agg_out, int64_t* n_agg_out) {
    if (n <= 0 || !indptr || !indices || !agg_out || !n_agg_out) return FEP_EINVAL;
    std::vector<int32_t> agg((size_t)n, -1);
    int32_t na = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (agg[i] >= 0) continue;
        bool free_nb = true;
        for (int32_t t = indptr[i]; t < indptr[i + 1] && free_nb; ++t) {
            if (indices[t] < 0 || indices[t] >= n) return FEP_ERANGE;
            free_nb = agg[indices[t]] < 0;
        }
        if (!free_nb) continue;
        for (int32_t t = indptr[i]; t < indptr[i + 1]; ++t) agg[indices[t]] = na;
        agg[i] = na++;
    }
    std::vector<int32_t> fin(agg);
    for (int64_t i = 0; i < n; ++i) {
        if (agg[i] >= 0) continue;
        int32_t jmin = -1;                               // the aggregated neighbour of smallest id: the list may repeat ids, in any order
        for (int32_t t = indptr[i]; t < indptr[i + 1]; ++t) {
            const int32_t j = indices[t];
            if (j < 0 || j >= n) return FEP_ERANGE;
            if (agg[j] >= 0 && (jmin < 0 || j < jmin)) jmin = j;
        }
        fin[i] = jmin >= 0 ? agg[jmin] : na++;
    }
    std::memcpy(agg_out, fin.data(), (size_t)n * sizeof(int32_t));
    *n_agg_out = na;
    return FEP_OK;
}

// ---------------------------------------------------------------------------------------
// Sparse products on FIXED patterns (multigrid: coarse operators re-projected from every tangent, A' = R A P with the
// transfers of the hierarchy).  The symbolic work is done once on the host: the pattern of X*Y, and for a given output
// pattern the TERMS of every output entry, C[c] = sum_t X[xa[t]] * Y[ya[t]], t in [tptr[c], tptr[c+1]) in ascending order
// of the X entry — the numeric phase on the device is one gather-multiply-add loop per entry, in a fixed order.
// ---------------------------------------------------------------------------------------
// pattern of X*Y (rows of X, columns of Y), column ids ascending per row
inline int product_pattern(int64_t n_rows, int64_t n_mid, int64_t n_cols, const int32_t* Xp, const int32_t* Xi, const int32_t* Yp,
                           const int32_t* Yi, std::vector<int32_t>& Cp, std::vector<int32_t>& Ci) {
    if (n_rows < 0 || n_mid < 0 || n_cols < 0 || !Xp || !Yp) return FEP_EINVAL;
    std::vector<int32_t> cnt((size_t)n_rows, 0);
    std::atomic<int> bad{0};
    auto row_cols = [&](int64_t i, std::vector<int32_t>& buf) {
        buf.clear();
        for (int32_t x = Xp[i]; x < Xp[i + 1]; ++x) {
            const int32_t j = Xi[x];
            if (j < 0 || j >= n_mid) { bad = 1; return; }
            for (int32_t y = Yp[j]; y < Yp[j + 1]; ++y) {
                if (Yi[y] < 0 || Yi[y] >= n_cols) { bad = 1; return; }
                buf.push_back(Yi[y]);
            }
        }
        std::sort(buf.begin(), buf.end());
        buf.erase(std::unique(buf.begin(), buf.end()), buf.end());
    };
    parallel_chunks(n_rows, [&](int64_t lo, int64_t hi, int) {
        std::vector<int32_t> buf;
        for (int64_t i = lo; i < hi; ++i) { row_cols(i, buf); cnt[(size_t)i] = (int32_t)buf.size(); }
    });
    if (bad) return FEP_ERANGE;
    Cp.assign((size_t)n_rows + 1, 0);
    int64_t tot = 0;
    for (int64_t i = 0; i < n_rows; ++i) {
        tot += cnt[(size_t)i];
        if (tot >= INT32_MAX) return FEP_ERANGE;
        Cp[(size_t)i + 1] = (int32_t)tot;
    }
    Ci.resize((size_t)tot);
    parallel_chunks(n_rows, [&](int64_t lo, int64_t hi, int) {
        std::vector<int32_t> buf;
        for (int64_t i = lo; i < hi; ++i) { row_cols(i, buf); std::copy(buf.begin(), buf.end(), Ci.begin() + Cp[(size_t)i]); }
    });
    return FEP_OK;
}

// Numeric sparse product C = X * Y on the host, rows in parallel (the multigrid set-up's Galerkin products: SciPy's are single
// threaded, 0.45 of the hierarchy's 1.3 s at 1 M DOFs).  Two calls: spgemm_count fills Cp (structural row sizes), spgemm_fill
// the column ids (ascending per row) and values; entries whose terms cancel to zero are kept (the caller drops them).
// Row i accumulates its terms in the order of X's entries, each Y row front to back (Gustavson), into a per-thread dense row.
inline int spgemm_count(int64_t n_rows, int64_t n_mid, int64_t n_cols, const int32_t* Xp, const int32_t* Xi, const int32_t* Yp,
                        const int32_t* Yi, int32_t* Cp) {
    if (n_rows < 0 || n_mid < 0 || n_cols < 0 || !Xp || !Yp || !Cp) return FEP_EINVAL;
    std::atomic<int> bad{0};
    std::vector<int32_t> cnt((size_t)n_rows, 0);
    parallel_chunks(n_rows, [&](int64_t lo, int64_t hi, int) {
        std::vector<int64_t> mark((size_t)n_cols, -1);
        for (int64_t i = lo; i < hi && !bad; ++i) {
            int32_t c = 0;
            for (int32_t x = Xp[i]; x < Xp[i + 1]; ++x) {
                const int32_t j = Xi[x];
                if (j < 0 || j >= n_mid) { bad = 1; return; }
                for (int32_t y = Yp[j]; y < Yp[j + 1]; ++y) {
                    const int32_t J = Yi[y];
                    if (J < 0 || J >= n_cols) { bad = 1; return; }
                    if (mark[(size_t)J] != i) { mark[(size_t)J] = i; ++c; }
                }
            }
            cnt[(size_t)i] = c;
        }
    });
    if (bad) return FEP_ERANGE;
    int64_t tot = 0;
    Cp[0] = 0;
    for (int64_t i = 0; i < n_rows; ++i) {
        tot += cnt[(size_t)i];
        if (tot >= INT32_MAX) return FEP_ERANGE;
        Cp[i + 1] = (int32_t)tot;
    }
    return FEP_OK;
}

inline int spgemm_fill(int64_t n_rows, int64_t n_mid, int64_t n_cols, const int32_t* Xp, const int32_t* Xi, const double* Xv,
                       const int32_t* Yp, const int32_t* Yi, const double* Yv, const int32_t* Cp, int32_t* Ci, double* Cv) {
    if (n_rows < 0 || n_mid < 0 || n_cols < 0 || !Xp || !Yp || !Cp || (Cp[n_rows] > 0 && (!Ci || !Cv || !Xv || !Yv))) return FEP_EINVAL;
    std::atomic<int> bad{0};
    parallel_chunks(n_rows, [&](int64_t lo, int64_t hi, int) {
        std::vector<int64_t> mark((size_t)n_cols, -1);
        std::vector<double> acc((size_t)n_cols, 0.0);
        std::vector<int32_t> cols;
        for (int64_t i = lo; i < hi && !bad; ++i) {
            cols.clear();
            for (int32_t x = Xp[i]; x < Xp[i + 1]; ++x) {
                const int32_t j = Xi[x];
                if (j < 0 || j >= n_mid) { bad = 1; return; }
                const double xv = Xv[x];
                for (int32_t y = Yp[j]; y < Yp[j + 1]; ++y) {
                    const int32_t J = Yi[y];
                    if (J < 0 || J >= n_cols) { bad = 1; return; }
                    if (mark[(size_t)J] != i) { mark[(size_t)J] = i; cols.push_back(J); acc[(size_t)J] = xv * Yv[y]; }
                    else acc[(size_t)J] += xv * Yv[y];
                }
            }
            if ((int64_t)cols.size() != (int64_t)Cp[i + 1] - Cp[i]) { bad = 2; return; }      // Cp is not spgemm_count's of these factors
            std::sort(cols.begin(), cols.end());
            int32_t t = Cp[i];
            for (int32_t J : cols) { Ci[t] = J; Cv[t] = acc[(size_t)J]; ++t; }
        }
    });
    return bad == 1 ? FEP_ERANGE : bad == 2 ? FEP_EINVAL : FEP_OK;
}

struct ProductPlan {
    std::vector<int32_t> tptr, xa, ya;       // terms of output entry c: xa / ya [tptr[c] .. tptr[c+1])
};

// terms of C = X*Y on the pattern (Cp, Ci) (column ids ascending per row; `dense_cols` > 0: C is dense with that many columns
// and Cp / Ci are not read).  FEP_EINVAL when a structural term of the product has no entry in the pattern.
inline int product_plan(int64_t n_rows, int64_t n_mid, const int32_t* Xp, const int32_t* Xi, const int32_t* Yp, const int32_t* Yi,
                        const int32_t* Cp, const int32_t* Ci, int64_t dense_cols, ProductPlan& P) {
    if (n_rows < 0 || !Xp || !Yp || (!dense_cols && (!Cp || !Ci))) return FEP_EINVAL;
    const int64_t n_out = dense_cols ? n_rows * dense_cols : Cp[n_rows];
    if (n_out >= INT32_MAX) return FEP_ERANGE;
    std::vector<int32_t> cnt((size_t)n_out + 1, 0);
    std::atomic<int> bad{0};
    int64_t n_cols = dense_cols;
    if (!dense_cols)
        for (int64_t c = 0; c < n_out; ++c) n_cols = std::max<int64_t>(n_cols, (int64_t)Ci[c] + 1);
    // position of column J in row i of C through a per-thread map column -> entry (entries of other rows fail the range test)
    auto walk = [&](int64_t lo, int64_t hi, bool fill) {
        std::vector<int32_t> pos(dense_cols ? 0 : (size_t)n_cols, -1);
        for (int64_t i = lo; i < hi && !bad; ++i) {
            if (!dense_cols)
                for (int32_t c = Cp[i]; c < Cp[i + 1]; ++c) pos[(size_t)Ci[c]] = c;
            for (int32_t x = Xp[i]; x < Xp[i + 1]; ++x) {
                const int32_t j = Xi[x];
                if (j < 0 || j >= n_mid) { bad = 2; return; }
                for (int32_t y = Yp[j]; y < Yp[j + 1]; ++y) {
                    const int32_t J = Yi[y];
                    int64_t c = -1;
                    if (J >= 0 && J < n_cols) c = dense_cols ? i * dense_cols + J : (int64_t)pos[(size_t)J];
                    if (c < 0 || (!dense_cols && (c < Cp[i] || c >= Cp[i + 1]))) { bad = 1; return; }
                    if (!fill) { ++cnt[(size_t)c + 1]; continue; }
                    const int32_t t = P.tptr[(size_t)c] + cnt[(size_t)c]++;
                    P.xa[(size_t)t] = x; P.ya[(size_t)t] = y;
                }
            }
        }
    };
    parallel_chunks(n_rows, [&](int64_t lo, int64_t hi, int) { walk(lo, hi, false); });      // rows own their output entries
    if (bad) return bad == 1 ? FEP_EINVAL : FEP_ERANGE;
    P.tptr.assign((size_t)n_out + 1, 0);
    int64_t tot = 0;
    for (int64_t c = 0; c < n_out; ++c) {
        tot += cnt[(size_t)c + 1];
        if (tot >= INT32_MAX) return FEP_ERANGE;
        P.tptr[(size_t)c + 1] = (int32_t)tot;
    }
    P.xa.resize((size_t)tot); P.ya.resize((size_t)tot);
    std::fill(cnt.begin(), cnt.end(), 0);
    parallel_chunks(n_rows, [&](int64_t lo, int64_t hi, int) { walk(lo, hi, true); });
    return bad ? FEP_EINVAL : FEP_OK;
}

}  // namespace fep_host
