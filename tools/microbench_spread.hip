// Follow-up to microbench_placement.hip: is a buffer made of physically SCATTERED pieces faster to stream at random than one made of neighbouring pieces?
// M pieces of 1 GB are allocated one after the other (neighbours in allocation order are, mostly, neighbours in HBM); a "buffer" is a table of 12 of them:
// consecutive pieces, every 2nd, every 4th, every 8th ... piece. Random 1 KB chunks as before, GB/s per selection, three passes.
//   hipcc --offload-arch=gfx950 -O2 tools/microbench_spread.hip -o /tmp/mbs && /tmp/mbs [pieces, default 96]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
constexpr unsigned NSEL = 12;                       // pieces per buffer
constexpr unsigned long long PIECE = 1ULL << 30;    // bytes
struct Table { const v4u *piece[NSEL]; };

__global__ __launch_bounds__(256) void gather(Table t, unsigned iters, unsigned *sink) {
    const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    unsigned long long state = 0x9E3779B97F4A7C15ULL * (wave + 1);
    unsigned acc = 0;
    for (unsigned i = 0; i < iters; i++) {
        state = state * 6364136223846793005ULL + 1442695040888963407ULL;
        const unsigned long long chunk = (state >> 20) % (NSEL * (PIECE / 1024));  // < NSEL x 2^20 chunks of 1 KB
        const v4u *base = t.piece[chunk >> 20];                                    // (piece index < NSEL)
        const v4u v = __builtin_nontemporal_load(&base[(chunk & 0xFFFFF) * 64 + lane]);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char **argv) {
    const unsigned m = argc > 1 ? (unsigned)atoi(argv[1]) : 96;
    if (m < NSEL * 8) { fprintf(stderr, "at least %u pieces\n", NSEL * 8); return 1; }
    std::vector<v4u *> piece(m);
    unsigned *sink;
    CHECK(hipMalloc(&sink, 4));
    for (unsigned b = 0; b < m; b++) {
        CHECK(hipMalloc(&piece[b], PIECE));
        CHECK(hipMemset(piece[b], (int)b + 1, PIECE));
    }
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const unsigned grid = 256 * 5, iters = 4000;
    struct Sel { const char *name; unsigned first, step; };
    const Sel sels[] = {{"consecutive from 0", 0, 1}, {"consecutive from 40", 40, 1}, {"every 2nd", 0, 2}, {"every 4th", 0, 4}, {"every 8th", 0, 8}, {"every 8th from 3", 3, 8}};
    for (int pass = 0; pass < 3; pass++)
        for (const Sel &s : sels) {
            Table t;
            for (unsigned k = 0; k < NSEL; k++) t.piece[k] = piece[s.first + k * s.step];  // (first + 11 * step < m by the check above)
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(gather, dim3(grid), dim3(256), 0, 0, t, iters, sink);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("pass %d  %-22s %7.1f GB/s\n", pass, s.name, (double)grid * 4 * iters * 1024 / ms / 1e6);
        }
    return 0;
}
