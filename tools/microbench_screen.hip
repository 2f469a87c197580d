// Would SCREENING pieces of HBM pay? M pieces of 1 GB fill the card; each is streamed on its own (random 1 KB chunks, as microbench_placement.hip) and ranked;
// then a 72 GB "array" is put together as a table of 72 pieces -- the fastest, the slowest, the first 72 in allocation order -- and streamed the same way.
//   hipcc --offload-arch=gfx950 -O2 tools/microbench_screen.hip -o /tmp/mbsc && /tmp/mbsc [pieces, default 280]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
constexpr unsigned long long PIECE = 1ULL << 30;  // bytes; 2^20 chunks of 1 KB

__global__ __launch_bounds__(256) void gather(const v4u *const *table, unsigned npieces, unsigned iters, unsigned *sink) {
    const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    unsigned long long state = 0x9E3779B97F4A7C15ULL * (wave + 1);
    unsigned acc = 0;
    for (unsigned i = 0; i < iters; i++) {
        state = state * 6364136223846793005ULL + 1442695040888963407ULL;
        const unsigned long long chunk = (state >> 20) % ((unsigned long long)npieces << 20);  // (piece index = chunk >> 20 < npieces)
        const v4u *base = table[chunk >> 20];
        const v4u v = __builtin_nontemporal_load(&base[(chunk & 0xFFFFF) * 64 + lane]);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char **argv) {
    const unsigned m = argc > 1 ? (unsigned)atoi(argv[1]) : 280, take = 72;
    if (m < 2 * take) { fprintf(stderr, "at least %u pieces\n", 2 * take); return 1; }
    std::vector<v4u *> piece(m);
    unsigned *sink;
    const v4u **d_table;
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMalloc(&d_table, m * sizeof(v4u *)));
    for (unsigned b = 0; b < m; b++) {
        CHECK(hipMalloc(&piece[b], PIECE));
        CHECK(hipMemset(piece[b], (int)b + 1, PIECE));
    }
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const unsigned grid = 256 * 5;
    auto run = [&](const std::vector<unsigned> &sel, unsigned iters, double *gbs) -> int {
        std::vector<const v4u *> t(sel.size());
        for (size_t k = 0; k < sel.size(); k++) t[k] = piece[sel[k]];
        CHECK(hipMemcpy(d_table, t.data(), t.size() * sizeof(v4u *), hipMemcpyHostToDevice));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(gather, dim3(grid), dim3(256), 0, 0, d_table, (unsigned)sel.size(), iters, sink);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        *gbs = (double)grid * 4 * iters * 1024 / ms / 1e6;
        return 0;
    };
    std::vector<double> speed(m, 0.0);
    for (int pass = 0; pass < 2; pass++)  // (the second pass counts: the first touches the pages)
        for (unsigned b = 0; b < m; b++)
            if (run({b}, 1000, &speed[b])) return 1;
    std::vector<unsigned> order(m);
    std::iota(order.begin(), order.end(), 0u);
    std::sort(order.begin(), order.end(), [&](unsigned x, unsigned y) { return speed[x] > speed[y]; });
    printf("pieces alone: fastest %.0f, median %.0f, slowest %.0f GB/s\n", speed[order[0]], speed[order[m / 2]], speed[order[m - 1]]);
    std::vector<unsigned> best(order.begin(), order.begin() + take), worst(order.end() - take, order.end()), first(take), mid(order.begin() + (m - take) / 2, order.begin() + (m - take) / 2 + take);
    std::iota(first.begin(), first.end(), 0u);
    for (int pass = 0; pass < 3; pass++) {
        double a, b, c, d;
        if (run(best, 4000, &a) || run(worst, 4000, &b) || run(first, 4000, &c) || run(mid, 4000, &d)) return 1;
        printf("pass %d: 72 GB from the fastest pieces %.0f GB/s, from the slowest %.0f, the first 72 allocated %.0f, the middle of the ranking %.0f\n", pass, a, b, c, d);
    }
    return 0;
}
