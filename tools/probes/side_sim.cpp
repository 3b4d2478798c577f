// Host-only what-if: blocks whose 16-byte row pieces share a 64-byte chunk with another writer's pieces go through the side buffer
// ("side" blocks, like the open ones) and fixup_kernel writes them, its list in destination order so that adjacent lanes complete
// chunks.  Counts, per closure round k: side blocks with own partials, fix-up entries, chunks the element kernel still writes
// partially, chunks the fix-up completes / writes partially.
//   g++ -std=c++17 -O2 -pthread -o /tmp/side_sim tools/probes/side_sim.cpp && /tmp/side_sim mesh.bin <elements per patch> <runs>
#include "../../fem-elastoplasticity_amd/csrc/fep_host.h"
#include <cstdio>
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
    const int eb = std::atoi(argv[2]);
    std::vector<int32_t> pel, patch_of;
    patch_grouping(S, n_p, n_e, n_n, elem.data(), xy.data(), eb, opt, pel, patch_of);
    const int64_t n_blk = (int64_t)S.ncol.size(), nnz2 = 2 * n_blk, n_chunk = (nnz2 + 3) / 4;
    // owner of every block: patch id, or -2 when several patches contribute (open)
    std::vector<int32_t> owner((size_t)n_blk), rown((size_t)n_blk);
    for (int64_t n = 0; n < n_n; ++n)
        for (int32_t b = S.nptr[n]; b < S.nptr[n + 1]; ++b) {
            rown[b] = (int32_t)n;
            int32_t o = -1;
            for (int32_t t = S.segptr[b]; t < S.segptr[b + 1]; ++t) {
                const int32_t p = patch_of[(size_t)((int64_t)S.perm[t] % n_e)];
                o = o == -1 ? p : (o == p ? o : -2);
            }
            owner[b] = o;
        }
    auto pos0 = [&](int64_t b) { const int64_t n = rown[b]; return 2 * (int64_t)S.nptr[n] + (b - S.nptr[n]); };
    auto deg = [&](int64_t b) { const int64_t n = rown[b]; return (int64_t)(S.nptr[n + 1] - S.nptr[n]); };
    std::vector<uint8_t> side((size_t)n_blk, 0);
    int64_t n_open = 0;
    for (int64_t b = 0; b < n_blk; ++b) if (owner[b] == -2) { side[b] = 1; ++n_open; }
    for (int round = 0; round <= 3; ++round) {
        // chunk state from the DIRECT writers: -1 nobody, >= 0 one patch, -2 several patches; has_side: a side piece lies in it
        std::vector<int32_t> cw((size_t)n_chunk, -1);
        std::vector<uint8_t> has_side((size_t)n_chunk, 0), n_side((size_t)n_chunk, 0);
        for (int64_t b = 0; b < n_blk; ++b)
            for (int r = 0; r < 2; ++r) {
                const int64_t c = (pos0(b) + r * deg(b)) / 4;
                if (side[b]) { has_side[c] = 1; ++n_side[c]; continue; }
                cw[c] = cw[c] == -1 ? owner[b] : (cw[c] == owner[b] ? cw[c] : -2);
            }
        int64_t elem_partial = 0, fix_full = 0, fix_partial_chunks = 0, fix_pieces = 0, n_side_blocks = 0, own = 0;
        for (int64_t c = 0; c < n_chunk; ++c) {
            const int64_t len = std::min<int64_t>(4, nnz2 - 4 * c);
            if (cw[c] == -2) elem_partial += 2;                          // (at least two partial writes)
            else if (cw[c] >= 0 && has_side[c]) elem_partial += 1;
            if (has_side[c]) { fix_pieces += n_side[c]; if (n_side[c] == len) ++fix_full; else ++fix_partial_chunks; }
        }
        for (int64_t b = 0; b < n_blk; ++b) if (side[b]) { ++n_side_blocks; if (owner[b] != -2 || S.ncol[b] >= rown[b]) ++own; }
        std::printf("round %d: side blocks %.2f per element (own partial slots %.2f; open %.2f), element kernel's partial chunk writes %.2f per "
                    "element, fix-up: %.2f complete chunks + %.2f partial ones per element (%.2f pieces)\n", round,
                    (double)n_side_blocks / n_e, (double)own / n_e, (double)n_open / n_e, (double)elem_partial / n_e, (double)fix_full / n_e,
                    (double)fix_partial_chunks / n_e, (double)fix_pieces / n_e);
        // closure: every block with a piece in a chunk that is not one direct writer's alone becomes a side block
        std::vector<uint8_t> next(side);
        for (int64_t b = 0; b < n_blk; ++b) {
            if (side[b]) continue;
            for (int r = 0; r < 2; ++r) {
                const int64_t c = (pos0(b) + r * deg(b)) / 4;
                if (cw[c] == -2 || has_side[c]) next[b] = 1;
            }
        }
        side.swap(next);
    }
    return 0;
}
