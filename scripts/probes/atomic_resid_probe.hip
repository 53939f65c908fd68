// Residual add into an fp32 stream, two ways: read - add - write (what the RESID epilogue of the GEMM does: 8 B per element through the
// CU's vector memory path) against a no-return global atomic add (4 B per element towards the L2, no load).  Every element receives
// exactly one add per launch, so the atomic form is deterministic and gives the same bits.  Shapes: the hidden stream of the headline
// config, 86016 x 1536 fp32.  (a) whole-chip streaming rate of the two forms; (b) per-CU rate: one workgroup per CU walks 256 x 256
// element tiles in 16-row passes (the epilogue's access pattern), 4 waves.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o atomic_resid_probe atomic_resid_probe.hip && ./atomic_resid_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int64_t M = 86016, N = 1536;

__global__ void __launch_bounds__(256) rmw_stream(float* __restrict__ x, float v, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 a = ((float4*)x)[i];
        a.x += v; a.y += v; a.z += v; a.w += v;
        ((float4*)x)[i] = a;
    }
}
__global__ void __launch_bounds__(256) atomic_stream(float* __restrict__ x, float v, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float* p = x + 4 * i;
#pragma unroll
        for (int j = 0; j < 4; ++j) __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float*)(p + j), v);
    }
}
// lanes of a wave on consecutive floats (256 B per instruction) instead of 16 B per lane
__global__ void __launch_bounds__(256) atomic_stream_lane(float* __restrict__ x, float v, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float*)(x + i), v);
}
// per-CU tile walk: workgroup b owns tiles b, b + grid, ... of the 336 x 6 tile grid; a tile = 256 rows x 256 columns, 16-row passes,
// wave w rows 4 w .. 4 w + 3 of a pass, lane = 16 B (4 floats) of the 1-KiB row piece
template <int MODE>
__global__ void __launch_bounds__(256) tile_walk(float* __restrict__ x, float v) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ntn = (int)(N / 256), ntiles = (int)(M / 256) * ntn;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tm = t / ntn, tn = t - tm * ntn;
        float* base = x + (int64_t)tm * 256 * N + tn * 256 + lane * 4;
        for (int pass = 0; pass < 16; ++pass) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float* p = base + (int64_t)(pass * 16 + wave * 4 + r) * N;
                if (MODE == 0) {
                    float4 a = *(float4*)p;
                    a.x += v; a.y += v; a.z += v; a.w += v;
                    *(float4*)p = a;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float*)(p + j), v);
                }
            }
        }
    }
}

template <class F>
static float timeit(F f, int iters = 10) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f(); f();
    (void)hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) f();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / iters;
}

int main() {
    float* x;
    if (hipMalloc(&x, M * N * sizeof(float)) != hipSuccess) return 1;
    (void)hipMemset(x, 0, M * N * sizeof(float));
    const int64_t n = M * N, n4 = n / 4;
    const double gb = n * 4 / 1e9;
    float t;
    t = timeit([&] { rmw_stream<<<256 * 8, 256>>>(x, 1.f, n4); });
    printf("whole chip  read-add-write (float4)        %8.1f us  %6.0f GB/s moved (8 B / element)\n", t * 1e3, 2 * gb / t * 1e3);
    t = timeit([&] { atomic_stream<<<256 * 8, 256>>>(x, 1.f, n4); });
    printf("whole chip  atomic add, 4 per lane         %8.1f us  %6.0f M elements/us-equivalent GB/s %6.0f (4 B / element)\n", t * 1e3, 0.0, gb / t * 1e3);
    t = timeit([&] { atomic_stream_lane<<<256 * 8, 256>>>(x, 1.f, n); });
    printf("whole chip  atomic add, lane-contiguous    %8.1f us  GB/s %6.0f (4 B / element)\n", t * 1e3, gb / t * 1e3);
    t = timeit([&] { tile_walk<0><<<256, 256>>>(x, 1.f); });
    printf("per-CU walk read-add-write                 %8.1f us  = %5.2f us per 256 x 256 tile per CU\n", t * 1e3, t * 1e3 / (double)(M / 256 * (N / 256)) * 256);
    t = timeit([&] { tile_walk<1><<<256, 256>>>(x, 1.f); });
    printf("per-CU walk atomic add                     %8.1f us  = %5.2f us per 256 x 256 tile per CU\n", t * 1e3, t * 1e3 / (double)(M / 256 * (N / 256)) * 256);
    // the sums: 0 + adds of 1.0 -> every element equals the number of launches that touched it in either form
    float h[4];
    (void)hipMemcpy(h, x, sizeof(h), hipMemcpyDeviceToHost);
    printf("x[0..3] = %.0f %.0f %.0f %.0f (all forms add 1.0 per launch)\n", h[0], h[1], h[2], h[3]);
    return 0;
}
