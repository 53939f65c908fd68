// Issue-overlap probe for gfx950: how many independent VALU ops hide under one
// v_mfma_f32_32x32x16_bf16 when finely interleaved in one wave / issued by other waves of the SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_probe mfma_valu_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define VOP(r)                                                                             \
    do {                                                                                   \
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r) : "v"(seed));     \
        else if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(r));                    \
        else asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(r) : "v"(seed));           \
    } while (0)

// MODE 0: every wave runs {MFMA, NV VALU} x 4 per iteration (4 independent accumulators)
// MODE 1: waves 0..3 MFMA only, waves 4..7 VALU only (one of each per SIMD)
// MODE 2: VALU only, MODE 3: MFMA only
template <int NV, int KIND, int MODE, int MF>   // MF 0: 32x32x16, 1: 16x16x32
__global__ void __launch_bounds__(512) probe(float* out, int iters, float seed) {
    const int wave = threadIdx.x >> 6;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(threadIdx.x + i); b[i] = (short)(threadIdx.x * 3 + i); }
    f32x16 acc[4] = {{0}, {0}, {0}, {0}};
    f32x4 acs[4] = {{0}, {0}, {0}, {0}};
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = seed + i;
    const bool do_m = MODE == 0 || MODE == 3 || (MODE == 1 && wave < 4);
    const bool do_v = MODE == 0 || MODE == 2 || (MODE == 1 && wave >= 4);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (do_m) {
                if (MF == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[q]) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acs[q]) : "v"(a), "v"(b));
            }
            if (do_v) {
#pragma unroll
                for (int v = 0; v < NV; ++v) VOP(x[(q * NV + v) & 7]);
            }
        }
    }
    float s = 0;
    for (int q = 0; q < 4; ++q) { for (int i = 0; i < 16; ++i) s += acc[q][i]; for (int i = 0; i < 4; ++i) s += acs[q][i]; }
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, int KIND, int MODE, int MF>
void run(const char* name, float* out, int waves_per_simd = 1) {
    const int iters = 20000, threads = (MODE == 1 ? 512 : 256 * waves_per_simd);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe<NV, KIND, MODE, MF><<<256, threads>>>(out, 100, 1.f);
    (void)hipEventRecord(e0);
    probe<NV, KIND, MODE, MF><<<256, threads>>>(out, iters, 1.f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s NV=%2d waves/simd=%d  ns per (MFMA + NV valu) = %6.2f\n", name, NV, MODE == 1 ? 2 : waves_per_simd, ms * 1e6 / iters / 4);
}

int main() {
    float* out; (void)hipMalloc(&out, 256 * 1024 * 4);
    run<0, 0, 3, 0>("mfma32x32x16 only", out);
    run<0, 0, 3, 0>("mfma32x32x16 only", out, 2);
    run<0, 0, 3, 1>("mfma16x16x32 only", out);
    run<8, 0, 2, 0>("fma only", out);
    run<8, 0, 2, 0>("fma only", out, 2);
    run<8, 1, 2, 0>("exp only", out);
    run<8, 2, 2, 0>("cvt_pk only", out);
    run<2, 0, 0, 0>("interleaved fma", out);
    run<4, 0, 0, 0>("interleaved fma", out);
    run<6, 0, 0, 0>("interleaved fma", out);
    run<8, 0, 0, 0>("interleaved fma", out);
    run<12, 0, 0, 0>("interleaved fma", out);
    run<16, 0, 0, 0>("interleaved fma", out);
    run<4, 0, 0, 0>("interleaved fma", out, 2);
    run<8, 0, 0, 0>("interleaved fma", out, 2);
    run<4, 1, 0, 0>("interleaved exp", out);
    run<8, 1, 0, 0>("interleaved exp", out);
    run<4, 0, 0, 1>("interleaved fma (16x16x32)", out);
    run<8, 0, 0, 1>("interleaved fma (16x16x32)", out);
    run<4, 0, 1, 0>("split waves fma", out);
    run<8, 0, 1, 0>("split waves fma", out);
    run<16, 0, 1, 0>("split waves fma", out);
    return 0;
}
