// microbench_gather.hip -- how fast can MI355X serve RANDOM small gathers from a multi-GB table?
// This is the access pattern of candidate verification in the BASAL core (one 16-byte read at a
// random 8-byte-aligned offset per lane); it gives the practical ceiling for the kernel's roofline.
// build: hipcc --offload-arch=gfx950 -O3 tools/microbench_gather.hip -o /tmp/mbg ; run: /tmp/mbg [GiB] [loads-in-flight]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

template <int ILP, int BYTES>
__global__ __launch_bounds__(256) void gather(const uint64_t *__restrict__ tab, uint64_t nwords, uint64_t iters, uint64_t *out) {
    uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t acc = 0, st = mix(tid + 1);
    for (uint64_t it = 0; it < iters; it++) {
        uint64_t idx[ILP];
#pragma unroll
        for (int k = 0; k < ILP; k++) { st = mix(st + k + 1); idx[k] = st % (nwords - 8); }
#pragma unroll
        for (int k = 0; k < ILP; k++) {
            acc += tab[idx[k]];
            if (BYTES >= 16) acc += tab[idx[k] + 1];
            if (BYTES >= 64) { acc += tab[idx[k] + 2] + tab[idx[k] + 3] + tab[idx[k] + 4] + tab[idx[k] + 5] + tab[idx[k] + 6] + tab[idx[k] + 7]; }
        }
    }
    if (acc == 0x1234567) out[0] = acc;
}

template <int ILP, int BYTES>
double run(const uint64_t *d, uint64_t nwords, int blocks_per_cu, int cus, uint64_t *d_out) {
    uint64_t iters = 64;
    dim3 grid(cus * blocks_per_cu), block(256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((gather<ILP, BYTES>), grid, block, 0, 0, d, nwords, 4, d_out);
    hipEventRecord(e0);
    hipLaunchKernelGGL((gather<ILP, BYTES>), grid, block, 0, 0, d, nwords, iters, d_out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double loads = (double)grid.x * 256 * iters * ILP;
    return loads / (ms * 1e-3);
}

int main(int argc, char **argv) {
    double gib = argc > 1 ? atof(argv[1]) : 6.0;
    uint64_t nwords = (uint64_t)(gib * (1ull << 30)) / 8;
    uint64_t *d, *d_out;
    hipMalloc(&d, nwords * 8); hipMalloc(&d_out, 8);
    hipMemset(d, 1, nwords * 8);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount;
    printf("table %.1f GiB, %d CUs\n", gib, cus);
    printf("%-28s %14s %14s\n", "config", "G gathers/s", "TB/s @64B");
    for (int bpc : {4, 8}) {
        double r;
        r = run<1, 16>(d, nwords, bpc, cus, d_out); printf("16B x1 in flight, %d blk/CU   %14.2f %14.3f\n", bpc, r / 1e9, r * 64 / 1e12);
        r = run<4, 16>(d, nwords, bpc, cus, d_out); printf("16B x4 in flight, %d blk/CU   %14.2f %14.3f\n", bpc, r / 1e9, r * 64 / 1e12);
        r = run<8, 16>(d, nwords, bpc, cus, d_out); printf("16B x8 in flight, %d blk/CU   %14.2f %14.3f\n", bpc, r / 1e9, r * 64 / 1e12);
        r = run<4, 8>(d, nwords, bpc, cus, d_out);  printf(" 8B x4 in flight, %d blk/CU   %14.2f %14.3f\n", bpc, r / 1e9, r * 64 / 1e12);
        r = run<4, 64>(d, nwords, bpc, cus, d_out); printf("64B x4 in flight, %d blk/CU   %14.2f %14.3f\n", bpc, r / 1e9, r * 64 / 1e12);
    }
    return 0;
}
