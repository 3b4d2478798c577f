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
#include <cstdint>
#include <cstdlib>
#include <cstring>
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
template <class F> void parallel_chunks(int64_t n, F&& f) {
    const int nw = (int)std::max<int64_t>(1, std::min<int64_t>(worker_count(), n));
    std::vector<std::thread> th;
    th.reserve(nw);
    std::vector<int> inline_chunks;
    for (int w = 1; w < nw; ++w) {
        const int64_t lo = n * w / nw, hi = n * (w + 1) / nw;
        try { th.emplace_back([&f, lo, hi, w]() { f(lo, hi, w); }); }
        catch (const std::system_error&) { inline_chunks.push_back(w); }
    }
    f(0, n / nw, 0);
    for (int w : inline_chunks) f(n * w / nw, n * (w + 1) / nw, w);
    for (auto& t : th) t.join();
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
    // per tile: blocks in tile order, sorted unique element list, local gather codes
    std::vector<std::vector<int32_t>> lists(n_wg);
    std::vector<std::vector<uint16_t>> codes(n_wg);
    std::vector<uint8_t> too_big((size_t)n_wg, 0);
    parallel_chunks(n_wg, [&](int64_t lo, int64_t hi, int) {
        for (int64_t g = lo; g < hi; ++g) {
            const Tile& t = tiles[g];
            std::vector<int32_t>& l = lists[g];
            for (int s = 0; s < t.nseg; ++s)
                for (int32_t c = S.segptr[t.fb[s]]; c < S.segptr[t.fb[s] + t.nb[s]]; ++c) l.push_back(P.perm2[c] >> 4);
            std::sort(l.begin(), l.end());
            l.erase(std::unique(l.begin(), l.end()), l.end());
            if (l.size() > 4095) { too_big[g] = 1; continue; }
            std::vector<uint16_t>& cd = codes[g];
            for (int s = 0; s < t.nseg; ++s)
                for (int32_t c = S.segptr[t.fb[s]]; c < S.segptr[t.fb[s] + t.nb[s]]; ++c) {
                    const int32_t loc = (int32_t)(std::lower_bound(l.begin(), l.end(), P.perm2[c] >> 4) - l.begin());
                    cd.push_back((uint16_t)((loc << 4) | (P.perm2[c] & 15)));
                }
        }
    });
    size_t lmax = 0, cmax = 0;
    for (int64_t g = 0; g < n_wg; ++g) {
        if (too_big[g]) return FEP_OK;                                               // plan without the LDS route
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
    {   // owner bits, in tile order
        std::vector<uint8_t> owned((size_t)n_e, 0);
        for (int64_t g = 0; g < n_wg; ++g)
            for (size_t i = 0; i < lists[g].size(); ++i) {
                const int32_t e = lists[g][i];
                if (!owned[e]) { owned[e] = 1; P.elnodes[(size_t)(g * LP) + i] |= 1u << 30; }
            }
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
        if (d[0] != pk_base || nseg < 1 || nseg > kSegMax || d[1] > P.tile || d[1] < 1 || d[2] > 255) return 2;
        int nb = 0, nn = 0;
        for (int s = 0; s < nseg; ++s) {
            const int32_t fb = d[4 + 4 * s], sb = d[5 + 4 * s], fn = d[6 + 4 * s], sn = d[7 + 4 * s];
            if (fn < 0 || fn + sn > n_n || sn < 1 || S.nptr[fn] != fb || S.nptr[fn + sn] != fb + sb) return 3;
            for (int32_t b = fb; b < fb + sb; ++b) { if (seen[b]) return 4; seen[b] = 1; }
            nb += sb; nn += sn;
        }
        if (nb != d[1] || nn != d[2]) return 5;
        pk_base += nb;
        if (!P.lds) continue;
        const int32_t* el = P.elist_pad.data() + g * P.L;
        int n_list = 1;
        for (int i = 0; i < P.L; ++i) {
            if (el[i] < 0 || el[i] >= n_e) return 6;
            if (i > 0 && el[i] > el[i - 1]) n_list = i + 1;
        }
        if (n_list != n_staged) return 18;
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
                if ((w >> 30) & 1u) { if (i >= n_list) return 14; owners[el[i]]++; }
            }
        }
    }
    if (pk_base != n_blk) return 15;
    for (int64_t b = 0; b < n_blk; ++b) if (!seen[b]) return 16;
    if (P.fused) for (int64_t e = 0; e < n_e; ++e) if (owners[e] != 1) return 17;
    return 0;
}

// ---------------------------------------------------------------------------------------
// Opt-in node route of P2 / Q1 / Q2 (FEP_GEN_PATH=node): per tile of `tile` consecutive blocks the sorted unique
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

// Greedy aggregation of a node graph in CSR (smoothed-aggregation multigrid setup): agg_out[i] in [0, *n_agg_out).
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
        int32_t a = -1;
        for (int32_t t = indptr[i]; t < indptr[i + 1] && a < 0; ++t) {
            if (indices[t] < 0 || indices[t] >= n) return FEP_ERANGE;
            a = agg[indices[t]];
        }
        fin[i] = a >= 0 ? a : na++;
    }
    std::memcpy(agg_out, fin.data(), (size_t)n * sizeof(int32_t));
    *n_agg_out = na;
    return FEP_OK;
}

}  // namespace fep_host
