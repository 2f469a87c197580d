// Does plain random streaming depend on where hipMalloc put a buffer? N buffers of S GB, all resident; for each, every wave of a full grid reads random 1 KB chunks
// (64 lanes x 16 B, non-temporal -- the shape of the long-list stream of align_kernel<.., HEAVY>) and the GB/s are printed, twice, so that a buffer's figure can be
// told from run-to-run noise.   hipcc --offload-arch=gfx950 -O2 tools/microbench_placement.hip -o /tmp/mbp && /tmp/mbp [S_GB] [N] [contiguous 0/1]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef unsigned int v4u __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void gather(const v4u *buf, unsigned long long nchunks, unsigned iters, unsigned *sink) {
    const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    unsigned long long state = 0x9E3779B97F4A7C15ULL * (wave + 1);
    unsigned acc = 0;
    for (unsigned i = 0; i < iters; i++) {
        state = state * 6364136223846793005ULL + 1442695040888963407ULL;
        const unsigned long long chunk = (state >> 20) % nchunks;  // (chunk < nchunks: 64 uint4 per chunk, all inside the buffer)
        const v4u v = __builtin_nontemporal_load(&buf[chunk * 64 + lane]);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char **argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 12.0;
    const int n = argc > 2 ? atoi(argv[2]) : 8;
    const bool contig = argc > 3 && atoi(argv[3]) != 0;
    const size_t bytes = (size_t)(gb * 1e9) / 1024 * 1024;
    const unsigned long long nchunks = bytes / 1024;
    std::vector<v4u *> bufs(n);
    unsigned *sink;
    CHECK(hipMalloc(&sink, 4));
    for (int b = 0; b < n; b++) {
        if (contig) {  // physically contiguous (best effort); falls back to the ordinary allocation
            if (hipExtMallocWithFlags((void **)&bufs[b], bytes, hipDeviceMallocContiguous) != hipSuccess) {
                (void)hipGetLastError();
                printf("buffer %d: no contiguous block, ordinary allocation\n", b);
                CHECK(hipMalloc(&bufs[b], bytes));
            }
        } else
            CHECK(hipMalloc(&bufs[b], bytes));
        CHECK(hipMemset(bufs[b], b + 1, bytes));
    }
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const unsigned grid = 256 * 5, iters = 4000;  // 5 120 waves x 4 000 chunks x 1 KB = 21 GB per launch
    for (int pass = 0; pass < 3; pass++) {
        printf("pass %d:", pass);
        for (int b = 0; b < n; b++) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(gather, dim3(grid), dim3(256), 0, 0, bufs[b], nchunks, iters, sink);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf(" %7.1f", (double)grid * 4 * iters * 1024 / ms / 1e6);
        }
        printf("  GB/s per buffer (%p ...)\n", (void *)bufs[0]);
    }
    return 0;
}
