// Dependent-accumulate probe for gfx950: time per v_mfma_f32_32x32x16_bf16 when NACC accumulators
// rotate (NACC = 1: every MFMA consumes the previous result as SrcC).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NACC>
__global__ void __launch_bounds__(256) probe(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(threadIdx.x + i); b[i] = (short)(threadIdx.x * 3 + i); }
    f32x16 acc[8] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[q % NACC]) : "v"(a), "v"(b));
    }
    float s = 0;
    for (int q = 0; q < 8; ++q) for (int i = 0; i < 16; ++i) s += acc[q][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(float* out, int waves_per_simd) {
    const int iters = 10000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe<NACC><<<256 * waves_per_simd, 256>>>(out, 100);
    (void)hipEventRecord(e0);
    probe<NACC><<<256 * waves_per_simd, 256>>>(out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("NACC=%d waves/simd=%d  ns per MFMA per SIMD = %6.2f\n", NACC, waves_per_simd, ms * 1e6 / iters / 8 / waves_per_simd);
}
int main() {
    float* out; (void)hipMalloc(&out, 1024 * 1024 * 4);
    run<1>(out, 1); run<2>(out, 1); run<4>(out, 1); run<8>(out, 1);
    run<1>(out, 2); run<2>(out, 2); run<1>(out, 3); run<2>(out, 3);
    return 0;
}
