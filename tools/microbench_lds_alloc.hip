// How much LDS does a block really take on gfx950? For a dynamic LDS size D, launch more 256-thread blocks than can be resident; every block counts itself in,
// waits a millisecond and reads the count: the first generation of blocks sees exactly the blocks that were resident with it. Resident blocks per CU against D
// shows the allocation unit (the occupancy API assumes none).
//   hipcc --offload-arch=gfx950 -O2 tools/microbench_lds_alloc.hip -o /tmp/lds_alloc && /tmp/lds_alloc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void probe(unsigned *arrived, unsigned *seen, unsigned long long wait_ticks) {
    extern __shared__ unsigned lds[];
    if (threadIdx.x == 0) {
        lds[0] = atomicAdd(arrived, 1u);
        const unsigned long long t0 = __builtin_readcyclecounter();  // s_memtime
        while (__builtin_readcyclecounter() - t0 < wait_ticks) __builtin_amdgcn_s_sleep(32);
        seen[blockIdx.x] = __hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const unsigned grid = (unsigned)cus * 10;  // at most 8 blocks of 256 threads per CU are resident (two waves per SIMD each would be 8 waves)
    unsigned *arrived, *seen;
    CHECK(hipMalloc(&arrived, 4));
    CHECK(hipMalloc(&seen, grid * 4));
    std::vector<unsigned> h(grid);
    CHECK(hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const size_t sizes[] = {20400, 20480, 20481, 20736, 20992, 21760, 21761, 31976, 32000, 32001, 32008, 32256, 32257, 32512, 32768, 32769, 40960, 40961, 53760, 53761, 53960, 54272, 54613, 54614, 65536, 81920, 81921};
    printf("CUs %d, LDS per CU by the API %zu B\n", cus, (size_t)prop.maxSharedMemoryPerMultiProcessor);
    for (size_t D : sizes) {
        CHECK(hipMemset(arrived, 0, 4));
        CHECK(hipMemset(seen, 0, grid * 4));
        hipLaunchKernelGGL(probe, dim3(grid), dim3(256), D, 0, arrived, seen, 200000ULL);  // 100 MHz counter: 2 ms
        CHECK(hipGetLastError());
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h.data(), seen, grid * 4, hipMemcpyDeviceToHost));
        unsigned first_gen = grid;
        for (unsigned v : h) if (v && v < first_gen) first_gen = v;
        int api = 0;
        CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, probe, 256, D));
        printf("dynamic LDS %6zu B: %5u blocks resident = %.2f per CU (occupancy API: %d)\n", D, first_gen, (double)first_gen / cus, api);
    }
    return 0;
}
