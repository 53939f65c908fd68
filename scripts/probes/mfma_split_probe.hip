// Do an MFMA stream (wave A) and a VALU stream (wave B) on the same SIMD overlap?  Per-wave
// timing with s_memrealtime (100 MHz) around fixed work; separate code paths, no inner branches.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// mode bit0: waves 0..3 run MFMA stream; bit1: waves 4..7 run VALU stream; bit2: waves 4..7 run interleaved M+8V
template <int BURST>
__global__ void __launch_bounds__(512) probe(float* out, unsigned long long* ticks, int iters, int mode, float seed) {
    const int wave = threadIdx.x >> 6;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(threadIdx.x + i); b[i] = (short)(threadIdx.x * 3 + i); }
    f32x16 acc[4] = {};
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = seed + i;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    if (wave < 4) {
        if (mode & 1)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int q = 0; q < 8; ++q) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[q & 3]) : "v"(a), "v"(b));
            }
    } else {
        if (mode & 2)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int v = 0; v < 64; ++v) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[v & 7]) : "v"(seed));
            }
        if (mode & 4)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
#pragma unroll
                    for (int u = 0; u < BURST; ++u) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[(q * BURST + u) & 3]) : "v"(a), "v"(b));
#pragma unroll
                    for (int v = 0; v < 8 * BURST; ++v) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[v & 7]) : "v"(seed));
                }
            }
    }
    const unsigned long long t1 = wall_clock64();
    float s = 0;
    for (int q = 0; q < 4; ++q) for (int i = 0; i < 16; ++i) s += acc[q][i];
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) ticks[wave] = t1 - t0;
}

template <int BURST>
void run(const char* name, int mode, float* out, unsigned long long* ticks) {
    const int iters = 20000;
    probe<BURST><<<256, 512>>>(out, ticks, 100, mode, 1.f);
    probe<BURST><<<256, 512>>>(out, ticks, iters, mode, 1.f);
    (void)hipDeviceSynchronize();
    unsigned long long h[8]; (void)hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
    // wall_clock64 ticks at 100 MHz -> 10 ns per tick
    printf("%-44s waveA(0): %7.2f ns/iter   waveB(4): %7.2f ns/iter\n", name, h[0] * 10.0 / iters, h[4] * 10.0 / iters);
}
int main() {
    float* out; unsigned long long* ticks;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&ticks, 64);
    run<1>("A: 8 MFMA/iter alone", 1, out, ticks);
    run<1>("B: 64 fma/iter alone", 2, out, ticks);
    run<1>("A: 8 MFMA  +  B: 64 fma", 3, out, ticks);
    run<1>("B': 8x(1 MFMA + 8 fma) alone", 4, out, ticks);
    run<1>("A: 8 MFMA  +  B': 8x(1 MFMA + 8 fma)", 5, out, ticks);
    run<4>("B'': 8x(4 MFMA + 32 fma) alone", 4, out, ticks);
    run<4>("A: 8 MFMA  +  B'': 8x(4 MFMA + 32 fma)", 5, out, ticks);
    return 0;
}
