// Host-only driver of fem-elastoplasticity_amd/csrc/fep_host.h for the sanitizer builds
// (tests/test_host_sanitizers.py: g++ -fsanitize=address,undefined and -fsanitize=thread; CPU only, no HIP).
//   host_san MESHFILE [max_segs]
// MESHFILE: int32 n_p, n_e, n_n, then elements (n_p x n_e, C order), optionally 2 x n_n doubles of coordinates.  Runs the symbolic phase (threaded), the COO
// tiles, the P1 plans with every table option (validated against the mesh), the opt-in node plan of P2/Q1/Q2 and
// the multigrid aggregation on the node graph; prints one summary line per plan; exit code 0 = all consistent.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../fem-elastoplasticity_amd/csrc/fep_host.h"

using namespace fep_host;

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t hdr[3];
    if (std::fread(hdr, sizeof(int32_t), 3, f) != 3) return 2;
    const int n_p = hdr[0];
    const int64_t n_e = hdr[1], n_n = hdr[2];
    std::vector<int32_t> elem((size_t)n_p * n_e);
    if (std::fread(elem.data(), sizeof(int32_t), elem.size(), f) != elem.size()) return 2;
    std::vector<double> coords(2 * (size_t)n_n);          // optional: planar coordinates x[n_n], y[n_n] behind the elements
    const bool have_coords = std::fread(coords.data(), sizeof(double), coords.size(), f) == coords.size();
    std::fclose(f);
    const int max_segs = argc > 2 ? std::atoi(argv[2]) : 2;
    Symbolic S;
    int r = build_symbolic(n_p, n_e, n_n, elem.data(), S);
    if (r != FEP_OK) { std::printf("build_symbolic: %d\n", r); return 1; }
    const int64_t n_blk = (int64_t)S.ncol.size();
    std::printf("symbolic: n_p %d n_e %lld n_n %lld blocks %lld contributions %zu\n", n_p, (long long)n_e, (long long)n_n,
                (long long)n_blk, S.perm.size());
    // pattern invariants: every block of node n lists n's neighbours in ascending order, one diagonal per node with elements
    for (int64_t n = 0; n < n_n; ++n) {
        int diag = 0;
        for (int32_t b = S.nptr[n]; b < S.nptr[n + 1]; ++b) {
            if (b > S.nptr[n] && S.ncol[b] <= S.ncol[b - 1]) return 1;
            if ((S.meta[b] >> 15) & 1u) { ++diag; if (S.ncol[b] != n) return 1; }
            if ((int)(S.meta[b] & 0x7fffu) != b - S.nptr[n] || (int)(S.meta[b] >> 16) != S.nptr[n + 1] - S.nptr[n]) return 1;
        }
        if (diag != (S.nptr[n + 1] > S.nptr[n] ? 1 : 0)) return 1;
    }
    std::vector<int32_t> tstart;
    r = row_tiles(S, n_n, 256, tstart);
    if (r != FEP_OK || tstart.front() != 0 || tstart.back() != n_blk) return 1;
    for (size_t i = 1; i < tstart.size(); ++i)
        if (tstart[i] - tstart[i - 1] > 256 || tstart[i] < tstart[i - 1]) return 1;
    int rc = 0;
    if (n_p == 3) {
        for (int segs = 1; segs <= kSegMax; ++segs) {       // the raw tilings, before build_p1_plan chooses
            P1Plan P;
            r = build_p1_plan_segs(S, n_e, n_n, elem.data(), P1Options(), segs, P);
            const int bad = r == FEP_OK ? validate_p1_plan(P, S, n_e, n_n, elem.data()) : -1;
            std::printf("tiling with <= %d segment(s): rc %d check %d tiles %lld staged %lld (%.3f per element) nodes %lld L %d C %d NL %d "
                        "lds %d rng %d pk %d fused %d/%d\n", segs, r, bad, (long long)P.n_wg, (long long)P.staged_total,
                        (double)P.staged_total / (double)n_e, (long long)P.staged_nodes_total, P.L, P.C, P.NL, (int)P.lds, (int)P.rng,
                        (int)P.pk, (int)P.fused, (int)P.fused_rng);
            if (r != FEP_OK || bad) rc = 1;
        }
        struct Case { const char* name; bool lds, rng, pk, fused; int segs; };
        const Case cases[] = {{"default", true, true, true, true, max_segs}, {"one segment", true, true, true, true, 1},
                              {"lists", true, false, true, true, max_segs}, {"unpacked", true, true, false, true, max_segs},
                              {"two kernels", true, true, true, false, max_segs}, {"direct", false, true, true, true, 1},
                              {"four segments", true, true, true, true, 4}};
        for (const Case& cs : cases) {
            P1Options opt;
            opt.allow_lds = cs.lds; opt.allow_rng = cs.rng; opt.allow_pk = cs.pk; opt.allow_fused = cs.fused; opt.max_segs = cs.segs;
            P1Plan P;
            r = build_p1_plan(S, n_e, n_n, elem.data(), opt, P);
            const int bad = r == FEP_OK ? validate_p1_plan(P, S, n_e, n_n, elem.data()) : -1;
            std::printf("p1 plan [%s]: rc %d check %d tiles %lld segs %d staged %lld (%.3f per element) nodes %lld L %d C %d NL %d "
                        "lds %d rng %d pk %d fused %d/%d\n", cs.name, r, bad, (long long)P.n_wg, P.n_segs, (long long)P.staged_total,
                        (double)P.staged_total / (double)n_e, (long long)P.staged_nodes_total, P.L, P.C, P.NL, (int)P.lds,
                        (int)P.rng, (int)P.pk, (int)P.fused, (int)P.fused_rng);
            if (r != FEP_OK || bad) rc = 1;
        }
    } else {
        const int n_q = n_p == 6 ? 7 : n_p == 4 ? 4 : n_p == 8 ? 9 : 12;
        GnPlan G;
        build_gn_plan(S, n_p, n_q, n_e, G);
        std::printf("node plan: ok %d tile %d L %d C %d lds %zu\n", (int)G.ok, G.tile, G.L, G.C, G.lds);
        if (G.ok) {
            const int64_t n_wg = (n_blk + G.tile - 1) / G.tile;
            if ((int64_t)G.elist_pad.size() != n_wg * G.L || (int64_t)G.codes_pad.size() != n_wg * G.C) rc = 1;
            for (int32_t e : G.elist_pad) if (e < 0 || e >= n_e) rc = 1;
            for (int64_t g = 0; g < n_wg && !rc; ++g) {
                const int64_t b0 = g * G.tile, b1 = std::min<int64_t>(n_blk, b0 + G.tile);
                for (int32_t t = S.segptr[b0]; t < S.segptr[b1]; ++t) {
                    const unsigned code = G.codes_pad[(size_t)(g * G.C + (t - S.segptr[b0]))];
                    const int64_t ab = S.perm[t] / n_e, e = S.perm[t] % n_e;
                    if ((int)(code >> 8) >= G.L || G.elist_pad[(size_t)(g * G.L + (code >> 8))] != e ||
                        (int)((code >> 4) & 15) != ab / n_p || (int)(code & 15) != ab % n_p) rc = 1;
                }
            }
        }
    }
    // patch plans of the element route (every element type runs it; P1 with FEP_ROUTE=patch | coo), at the kernels' patch sizes
    // and at odd ones, both groupings (consecutive elements / Hilbert curve through the centroids; a mesh file without
    // coordinates: node id -> a point of a 2-D lattice stands in) and both classifications of open blocks;
    // replayed against the symbolic phase contribution by contribution
    {
        std::vector<double> xy(2 * (size_t)n_n);
        const int64_t side = (int64_t)std::ceil(std::sqrt((double)n_n));
        for (int64_t n = 0; n < n_n; ++n) { xy[n] = (double)(n % side); xy[n_n + n] = (double)(n / side); }
        if (have_coords) xy = coords;
        const int ebs_np[][4] = {{3, 64, 7, 1}, {6, 56, 28, 5}, {4, 64, 9, 2}, {8, 28, 24, 3}, {15, 16, 4, 1}};
        for (const auto& row : ebs_np) {
            if (row[0] != n_p) continue;
            for (int k = 1; k < 4; ++k)
                for (int variant = 0; variant < 4; ++variant) {
                    PatchOptions opt;
                    opt.order = variant == 3 ? 2 : variant;                  // consecutive, Hilbert, 2 runs, 4 runs
                    opt.runs = variant == 3 ? 4 : 2;
                    opt.align = variant == 2 ? 16 : 1;
                    PatchPlan P;
                    r = build_patch_plan(S, n_p, n_e, n_n, elem.data(), xy.data(), row[k], opt, P);
                    const int bad = r == FEP_OK ? validate_patch_plan(P, S, n_p, n_e, n_n, elem.data()) : -1;
                    std::printf("patch plan eb %d order %d runs %d: rc %d ok %d check %d patches %lld items %zu (<= %d per patch) open blocks %lld "
                                "partials %lld (%.3f per element) open nodes %lld\n", row[k], opt.order, opt.runs, r, (int)P.ok, bad,
                                (long long)P.n_patch, P.items.size(), P.max_items, (long long)P.n_open, (long long)P.n_part,
                                (double)P.n_part / (double)n_e, (long long)P.n_fopen);
                    if (r != FEP_OK || bad || !P.ok) rc = 1;
                }
        }
    }
    // multigrid aggregation on the node graph
    std::vector<int32_t> agg((size_t)n_n);
    int64_t n_agg = 0;
    r = aggregate(n_n, S.nptr.data(), S.ncol.data(), agg.data(), &n_agg);
    if (r != FEP_OK || n_agg <= 0 || n_agg > n_n) rc = 1;
    for (int32_t a : agg) if (a < 0 || a >= n_agg) rc = 1;
    std::printf("aggregate: rc %d aggregates %lld\n", r, (long long)n_agg);
    // sparse products on fixed patterns (the multigrid refresh): A = node graph with made-up values, P = A * (tentative
    // aggregation), C = P^T (A P) through product_pattern / product_plan, against the triple loop
    {
        auto val = [](int64_t i, int64_t j, int k) { return 1.0 + (double)((i * 7 + j * 13 + k) % 11) / 8.0; };
        const int32_t* Ap = S.nptr.data();
        const int32_t* Ai = S.ncol.data();
        std::vector<double> Av(S.ncol.size());
        for (int64_t i = 0; i < n_n; ++i) for (int32_t t = Ap[i]; t < Ap[i + 1]; ++t) Av[(size_t)t] = val(i, Ai[t], 0);
        std::vector<int32_t> Gp((size_t)n_n + 1), Gi((size_t)n_n);                 // tentative: node -> its aggregate
        for (int64_t i = 0; i <= n_n; ++i) Gp[(size_t)i] = (int32_t)i;
        for (int64_t i = 0; i < n_n; ++i) Gi[(size_t)i] = agg[(size_t)i];
        std::vector<int32_t> Pp, Pi, Tp, Ti, Cp, Ci;
        int pr = product_pattern(n_n, n_n, n_agg, Ap, Ai, Gp.data(), Gi.data(), Pp, Pi);
        std::vector<double> Pv(Pi.size());
        for (int64_t i = 0; i < n_n && pr == FEP_OK; ++i) for (int32_t t = Pp[(size_t)i]; t < Pp[(size_t)i + 1]; ++t) Pv[(size_t)t] = val(i, Pi[(size_t)t], 3);
        // R = P^T
        std::vector<int32_t> Rp((size_t)n_agg + 1, 0), Ri(Pi.size()), Rsrc(Pi.size());
        for (int32_t c : Pi) ++Rp[(size_t)c + 1];
        for (int64_t a = 0; a < n_agg; ++a) Rp[(size_t)a + 1] += Rp[(size_t)a];
        { std::vector<int32_t> cur(Rp.begin(), Rp.end() - 1);
          for (int64_t i = 0; i < n_n && pr == FEP_OK; ++i) for (int32_t t = Pp[(size_t)i]; t < Pp[(size_t)i + 1]; ++t) { const int32_t q = cur[(size_t)Pi[(size_t)t]]++; Ri[(size_t)q] = (int32_t)i; Rsrc[(size_t)q] = t; } }
        ProductPlan h1, h2, h3;
        if (pr == FEP_OK) pr = product_pattern(n_n, n_n, n_agg, Ap, Ai, Pp.data(), Pi.data(), Tp, Ti);
        if (pr == FEP_OK) pr = product_plan(n_n, n_n, Ap, Ai, Pp.data(), Pi.data(), Tp.data(), Ti.data(), 0, h1);
        if (pr == FEP_OK) pr = product_pattern(n_agg, n_n, n_agg, Rp.data(), Ri.data(), Tp.data(), Ti.data(), Cp, Ci);
        if (pr == FEP_OK) pr = product_plan(n_agg, n_n, Rp.data(), Ri.data(), Tp.data(), Ti.data(), Cp.data(), Ci.data(), 0, h2);
        const bool small = n_agg <= 1500;
        if (pr == FEP_OK && small) pr = product_plan(n_agg, n_n, Rp.data(), Ri.data(), Tp.data(), Ti.data(), nullptr, nullptr, n_agg, h3);
        double worst = 0.0;
        if (pr == FEP_OK) {
            std::vector<double> Tv(Ti.size()), Cv(Ci.size());
            for (size_t c = 0; c < Tv.size(); ++c) { double a = 0; for (int32_t t = h1.tptr[c]; t < h1.tptr[c + 1]; ++t) a += Av[(size_t)h1.xa[(size_t)t]] * Pv[(size_t)h1.ya[(size_t)t]]; Tv[c] = a; }
            for (size_t c = 0; c < Cv.size(); ++c) { double a = 0; for (int32_t t = h2.tptr[c]; t < h2.tptr[c + 1]; ++t) a += Pv[(size_t)Rsrc[(size_t)h2.xa[(size_t)t]]] * Tv[(size_t)h2.ya[(size_t)t]]; Cv[c] = a; }
            // the triple loop, one coarse row at a time
            std::vector<double> rowv((size_t)n_agg);
            for (int64_t I = 0; I < n_agg; ++I) {
                std::fill(rowv.begin(), rowv.end(), 0.0);
                for (int32_t q = Rp[(size_t)I]; q < Rp[(size_t)I + 1]; ++q) {
                    const int32_t i = Ri[(size_t)q];
                    for (int32_t t = Ap[i]; t < Ap[i + 1]; ++t)
                        for (int32_t u = Pp[(size_t)Ai[t]]; u < Pp[(size_t)Ai[t] + 1]; ++u)
                            rowv[(size_t)Pi[(size_t)u]] += Pv[(size_t)Rsrc[(size_t)q]] * Av[(size_t)t] * Pv[(size_t)u];
                }
                for (int32_t c = Cp[(size_t)I]; c < Cp[(size_t)I + 1]; ++c) {
                    worst = std::max(worst, std::fabs(Cv[(size_t)c] - rowv[(size_t)Ci[(size_t)c]]) / (1.0 + std::fabs(rowv[(size_t)Ci[(size_t)c]])));
                    rowv[(size_t)Ci[(size_t)c]] = 0.0;
                }
                for (double v : rowv) if (v != 0.0) worst = 1.0;                   // a term outside the pattern
                if (small)
                    for (int64_t J = 0; J < n_agg; ++J) {
                        const size_t c = (size_t)(I * n_agg + J);
                        double a = 0; for (int32_t t = h3.tptr[c]; t < h3.tptr[c + 1]; ++t) a += Pv[(size_t)Rsrc[(size_t)h3.xa[(size_t)t]]] * Tv[(size_t)h3.ya[(size_t)t]];
                        const int32_t* it = std::lower_bound(Ci.data() + Cp[(size_t)I], Ci.data() + Cp[(size_t)I + 1], (int32_t)J);
                        const double ref = (it != Ci.data() + Cp[(size_t)I + 1] && *it == J) ? Cv[(size_t)(it - Ci.data())] : 0.0;
                        if (a != ref) worst = 1.0;
                    }
            }
        }
        std::printf("product plans: rc %d terms %zu + %zu worst %.2e\n", pr, h1.xa.size(), h2.xa.size(), worst);
        if (pr != FEP_OK || worst > 1e-13) rc = 1;
        // the threaded numeric product of the set-up (spgemm_count / spgemm_fill) on the same factors: T = A P has the
        // pattern of product_pattern and the values of the plan, summed in the same order (bitwise)
        if (pr == FEP_OK) {
            std::vector<int32_t> Sp((size_t)n_n + 1), Si;
            std::vector<double> Sv, Tv(Ti.size());
            for (size_t c = 0; c < Tv.size(); ++c) { double a = 0; for (int32_t t = h1.tptr[c]; t < h1.tptr[c + 1]; ++t) a += Av[(size_t)h1.xa[(size_t)t]] * Pv[(size_t)h1.ya[(size_t)t]]; Tv[c] = a; }
            int sr = spgemm_count(n_n, n_n, n_agg, Ap, Ai, Pp.data(), Pi.data(), Sp.data());
            if (sr == FEP_OK) { Si.resize((size_t)Sp[(size_t)n_n]); Sv.resize(Si.size()); sr = spgemm_fill(n_n, n_n, n_agg, Ap, Ai, Av.data(), Pp.data(), Pi.data(), Pv.data(), Sp.data(), Si.data(), Sv.data()); }
            const bool same = sr == FEP_OK && Sp == Tp && Si == Ti && Sv == Tv;
            std::printf("spgemm: rc %d entries %zu identical to the plan's product %d\n", sr, Si.size(), (int)same);
            if (!same) rc = 1;
        }
    }
    std::printf("result %s\n", rc ? "FAILED" : "ok");
    return rc;
}
