// microbench_valu.hip -- issue cost of the integer vector instructions the align kernels are made of (one wave per SIMD, 8 independent chains):
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_valu.hip -o /tmp/mb_valu && /tmp/mb_valu
// Prints shader cycles per wave-instruction. (Is a 64-bit shift one issue slot, or several? The stream filters shift 64-bit planes.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITER 2048
template <int OP>
__global__ __launch_bounds__(64) void k(uint64_t *out, uint32_t sh, uint64_t seed) {
    uint64_t a[8];
    for (int i = 0; i < 8; i++) a[i] = seed * (i + 1) + threadIdx.x;
    uint32_t b[8];
    for (int i = 0; i < 8; i++) b[i] = (uint32_t)(seed >> i) + threadIdx.x;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(a[i]) : "v"(sh));
            if (OP == 1) asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(a[i]));
            if (OP == 2) asm volatile("v_alignbit_b32 %0, %0, %1, 3" : "+v"(b[i]) : "v"(b[(i + 1) & 7]));
            if (OP == 3) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(b[i]) : "v"(b[(i + 1) & 7]), "v"(b[(i + 2) & 7]));
            if (OP == 4) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(b[i]) : "v"(b[(i + 1) & 7]));
            if (OP == 5) asm volatile("v_and_b32 %0, %1, %0" : "+v"(b[i]) : "v"(b[(i + 1) & 7]));
            if (OP == 6) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(b[i]));
            if (OP == 7) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
            if (OP == 8) asm volatile("v_ffbh_u32 %0, %0" : "+v"(b[i]));
            if (OP == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(b[i]) : "v"(b[(i + 1) & 7]));
            if (OP == 10) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(b[i]) : "v"(b[(i + 1) & 7]));
            if (OP == 11) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(b[i]) : "v"(b[(i + 1) & 7]) : "vcc");
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint64_t acc = 0;
    for (int i = 0; i < 8; i++) acc += a[i] + b[i];
    out[(size_t)gridDim.x + (size_t)blockIdx.x * 64 + threadIdx.x] = acc;  // (keeps the chains alive)
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char *name, int waves_per_simd) {
    uint64_t *d;
    const int blocks = 256 * 4 * waves_per_simd;  // one 64-thread block per wave slot
    const size_t words = (size_t)blocks * 65;
    hipMalloc(&d, words * 8);
    hipMemset(d, 0, words * 8);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d, 3u, 0x9E3779B97F4A7C15ull);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d, 3u, 0x9E3779B97F4A7C15ull);
    hipDeviceSynchronize();
    uint64_t h[64];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < 64; i++) m += (double)h[i];
    m /= 64;
    printf("%-28s %d wave(s)/SIMD: %.2f cycles per wave-instruction (per SIMD: %.2f)\n", name, waves_per_simd, m / (ITER * 8.0), m / (ITER * 8.0) / waves_per_simd);
    hipFree(d);
}

// throughput: every SIMD of the chip holds `w` waves of the same instruction stream; wall time by HIP events
template <int OP>
void thr(const char *name, int w) {
    uint64_t *d;
    const int blocks = 256 * 4 * w;
    const size_t words = (size_t)blocks * 65;
    hipMalloc(&d, words * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d, 3u, 0x9E3779B97F4A7C15ull);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 20; r++) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d, 3u, 0x9E3779B97F4A7C15ull);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double inst = 20.0 * blocks * ITER * 8.0;  // wave-instructions
    printf("%-28s %d wave(s)/SIMD: %.3f G wave-instructions/s per SIMD (chip: %.1f G/s); at 2.4 GHz %.2f cycles per wave-instruction per SIMD\n", name, w,
           inst / (ms * 1e-3) / 1024 / 1e9, inst / (ms * 1e-3) / 1e9, 2.4e9 / (inst / (ms * 1e-3) / 1024));
    hipFree(d);
}

int main() {
    for (int w : {1, 2, 4, 8}) { thr<3>("v_bfi_b32", w); thr<0>("v_lshlrev_b64", w); thr<4>("v_bcnt_u32_b32", w); thr<7>("v_lshl_add_u64", w); }

    for (int w : {1, 4}) {
        if (w == 1) {
            run<0>("v_lshlrev_b64 (vgpr amount)", 1); run<1>("v_lshrrev_b64 (imm)", 1); run<2>("v_alignbit_b32", 1); run<3>("v_bfi_b32", 1); run<4>("v_bcnt_u32_b32", 1);
            run<5>("v_and_b32", 1); run<6>("v_lshlrev_b32", 1); run<7>("v_lshl_add_u64", 1); run<8>("v_ffbh_u32", 1); run<9>("v_cndmask_b32", 1); run<10>("v_mul_lo_u32", 1); run<11>("v_add_co_u32", 1);
        } else {
            run<0>("v_lshlrev_b64 (vgpr amount)", 4); run<1>("v_lshrrev_b64 (imm)", 4); run<2>("v_alignbit_b32", 4); run<3>("v_bfi_b32", 4); run<4>("v_bcnt_u32_b32", 4);
            run<5>("v_and_b32", 4); run<7>("v_lshl_add_u64", 4); run<10>("v_mul_lo_u32", 4);
        }
    }
    return 0;
}
