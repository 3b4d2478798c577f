// Host-only analysis: who writes each 64-byte chunk of the CSR value array in the patch form of the element route?
// (MI355X: a store that covers an aligned 64-byte chunk is a plain write; anything smaller is a read-modify-write of the chunk in
// memory, 2.5x the cost per chunk — tools/probes/partial_write_probe.hip.)  Reads the mesh dump of tests/host_san.cpp
// (int32 n_p, n_e, n_n; elements[n_p][n_e]; optional coordinates), builds the product's patch plan and counts, per chunk, the
// distinct writers (patches with closed items in it, the fix-up kernel) and whether one writer covers it completely.
//   g++ -std=c++17 -O2 -pthread -I fem-elastoplasticity_amd/csrc -o /tmp/chunk_writers tools/probes/chunk_writers.cpp
//   /tmp/chunk_writers mesh.bin <elements per patch> <runs>
#include "../../fem-elastoplasticity_amd/csrc/fep_host.h"
#include <cstdio>
#include <map>
using namespace fep_host;
int main(int argc, char** argv) {
    if (argc < 4) return 2;
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t hdr[3];
    if (std::fread(hdr, 4, 3, f) != 3) return 2;
    const int n_p = hdr[0]; const int64_t n_e = hdr[1], n_n = hdr[2];
    std::vector<int32_t> elem((size_t)n_p * n_e);
    if (std::fread(elem.data(), 4, elem.size(), f) != elem.size()) return 2;
    std::vector<double> xy(2 * (size_t)n_n);
    if (std::fread(xy.data(), 8, xy.size(), f) != xy.size()) return 2;
    std::fclose(f);
    Symbolic S;
    if (build_symbolic(n_p, n_e, n_n, elem.data(), S) != FEP_OK) return 1;
    PatchOptions opt; opt.runs = std::atoi(argv[3]);
    PatchPlan P;
    if (build_patch_plan(S, n_p, n_e, n_n, elem.data(), xy.data(), std::atoi(argv[2]), opt, P) != FEP_OK || !P.ok) return 1;
    const int64_t nnz2 = 2 * (int64_t)S.ncol.size();                 // double2 units (16 bytes): 2 per block
    const int64_t n_chunk = (nnz2 + 3) / 4;
    // per chunk: pieces written by the first patch seen, number of distinct writers (saturating), pieces by the fix-up
    std::vector<int32_t> first((size_t)n_chunk, -1);
    std::vector<uint8_t> writers((size_t)n_chunk, 0), pieces((size_t)n_chunk, 0), fixp((size_t)n_chunk, 0);
    auto put = [&](int64_t pos, int32_t who) {
        const int64_t c = pos / 4;
        ++pieces[c];
        if (who < 0) { ++fixp[c]; return; }
        if (first[c] == who) return;
        if (first[c] < 0) { first[c] = who; writers[c] = 1; } else if (writers[c] < 255) ++writers[c];   // (a third patch counts once per piece: upper bound)
    };
    for (int64_t p = 0; p < P.n_patch; ++p) {
        const int32_t* d = P.pdesc.data() + p * kPatchDescInts;
        for (int32_t i = 0; i < d[1]; ++i) {
            const U2 it = P.items[(size_t)d[0] + i];
            if (it.x >> 31) continue;
            const int deg = (int)((it.x >> 19) & 4095u);
            put(it.y, (int32_t)p); put((int64_t)it.y + deg, (int32_t)p);
        }
    }
    for (size_t i = 0; i < P.fix.size(); ++i) {
        const U4& fx = P.fix[i]; const int deg = (int)(fx.y & 0xffffu);
        put(fx.x, -1); put((int64_t)fx.x + deg, -1);
        if (P.fixT[i].x != 0xffffffffu) { put(P.fixT[i].x, -1); put((int64_t)P.fixT[i].x + P.fixT[i].y, -1); }
    }
    int64_t full1 = 0, multi = 0, with_fix = 0, rmw_elem = 0, rmw_fix = 0, fix_chunks = 0;
    for (int64_t c = 0; c < n_chunk; ++c) {
        const bool last = c == n_chunk - 1 && (nnz2 & 3);
        const int w = writers[c] + (fixp[c] ? 1 : 0);
        if (w == 1 && (pieces[c] == 4 || last)) { ++full1; continue; }
        ++multi;
        rmw_elem += writers[c];                                  // every patch's part of a shared chunk: one partial write
        if (fixp[c]) { ++with_fix; rmw_fix += fixp[c]; ++fix_chunks; }
    }
    std::printf("n_p %d elements %lld patches %lld of <= %d: chunks %lld, one writer and complete %lld (%.1f %%), shared %lld (%.1f %%) of which the "
                "fix-up touches %lld\n  partial writes: element kernel %lld (%.2f per element), fix-up pieces %lld in %lld chunks (%.2f per "
                "element); open blocks %lld\n  cost in plain-chunk writes (partial = 2.5): as is %.0f k, every chunk written once %.0f k (x %.2f)\n",
                n_p, (long long)n_e, (long long)P.n_patch, P.eb, (long long)n_chunk, (long long)full1, 100.0 * full1 / n_chunk,
                (long long)multi, 100.0 * multi / n_chunk, (long long)with_fix, (long long)rmw_elem, (double)rmw_elem / n_e,
                (long long)rmw_fix, (long long)fix_chunks, (double)rmw_fix / n_e, (long long)P.n_open,
                (full1 + 2.5 * (rmw_elem + rmw_fix)) / 1e3, n_chunk / 1e3, (full1 + 2.5 * (rmw_elem + rmw_fix)) / (double)n_chunk);
    return 0;
}
