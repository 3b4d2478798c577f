// Probe: what does a scattered store of 16 / 32 / 64 / 128 contiguous bytes cost when the target line is in no cache?
// N pieces at pseudo-random 128-byte-aligned slots of a buffer much larger than L2 + Infinity Cache; PIECE bytes per slot written by
// PIECE/16 adjacent lanes (16 bytes each).  If memory performs a read-modify-write per partially written burst, 16- and 32-byte pieces
// cost as much as (or more than) 64-byte ones per piece.
//   hipcc --offload-arch=gfx950 -O3 -o partial_write_probe partial_write_probe.hip && ./partial_write_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int LANES>   // lanes per piece (16 bytes each)
__global__ void __launch_bounds__(256) scatter(double2* buf, uint64_t n_slots, uint64_t n_pieces, uint64_t mul, int offset16) {
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t piece = gid / LANES, lane = gid % LANES;
    if (piece >= n_pieces) return;
    const uint64_t slot = (piece * mul) % n_slots;             // mul odd and coprime with n_slots: a permutation
    buf[slot * 8 + offset16 + lane] = make_double2((double)piece, (double)lane);     // slot = 128 bytes = 8 double2
}

// Software read-modify-write: the four lanes of a chunk read its 64 bytes (one lane changes its piece) and write the chunk whole
__global__ void __launch_bounds__(256) sw_rmw(double2* buf, uint64_t n_slots, uint64_t n_pieces, uint64_t mul, int offset16) {
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t piece = gid / 4, lane = gid % 4;
    if (piece >= n_pieces) return;
    const uint64_t slot = (piece * mul) % n_slots;
    double2* p = buf + slot * 8 + (offset16 & 4) + lane;            // the 64-byte half of the line that holds the piece
    double2 v = *p;
    if ((int)lane == (offset16 & 3)) v = make_double2((double)piece, v.y + 1.0);
    *p = v;
}

int main() {
    const uint64_t bytes = 3ull << 30, n_slots = bytes / 128 - 1, n_pieces = 4ull << 20;   // 3 GiB, 4 Mi pieces
    double2* buf; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 0, bytes));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const uint64_t mul = 2654435761ull;                          // odd; n_slots is odd-ish: check coprimality loosely by design
    auto run = [&](int lanes, int off, const char* what) -> int {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            // evict: sweep a different 1 GiB region is implicit — pieces land on lines not touched since the memset / last pass
            const uint64_t threads = n_pieces * lanes; const unsigned grid = (unsigned)((threads + 255) / 256);
            CK(hipEventRecord(a));
            switch (lanes) {
                case 1: hipLaunchKernelGGL(scatter<1>, dim3(grid), dim3(256), 0, 0, buf, n_slots, n_pieces, mul + 2 * rep, off); break;
                case 2: hipLaunchKernelGGL(scatter<2>, dim3(grid), dim3(256), 0, 0, buf, n_slots, n_pieces, mul + 2 * rep, off); break;
                case 4: hipLaunchKernelGGL(scatter<4>, dim3(grid), dim3(256), 0, 0, buf, n_slots, n_pieces, mul + 2 * rep, off); break;
                case 8: hipLaunchKernelGGL(scatter<8>, dim3(grid), dim3(256), 0, 0, buf, n_slots, n_pieces, mul + 2 * rep, off); break;
            }
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
        }
        std::printf("%-44s %7.1f us  %6.1f G pieces/s  payload %5.2f TB/s\n", what, best * 1e3, n_pieces / (best * 1e-3) / 1e9,
                    (double)n_pieces * lanes * 16 / (best * 1e-3) / 1e12);
        return 0;
    };
    if (run(1, 0, "16 B per piece (offset 0 of the line)")) return 1;
    if (run(1, 3, "16 B per piece (offset 48)")) return 1;
    if (run(2, 0, "32 B per piece (aligned sector)")) return 1;
    if (run(2, 1, "32 B per piece (offset 16: two sectors)")) return 1;
    if (run(4, 0, "64 B per piece (aligned half line)")) return 1;
    if (run(4, 2, "64 B per piece (offset 32)")) return 1;
    if (run(8, 0, "128 B per piece (whole line)")) return 1;
    {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            const unsigned grid = (unsigned)((n_pieces * 4 + 255) / 256);
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(sw_rmw, dim3(grid), dim3(256), 0, 0, buf, n_slots, n_pieces, mul + 2 * rep + 100, 1);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
        }
        std::printf("%-44s %7.1f us  %6.1f G pieces/s\n", "16 B changed by reading + writing the 64 B", best * 1e3, n_pieces / (best * 1e-3) / 1e9);
    }
    return 0;
}
