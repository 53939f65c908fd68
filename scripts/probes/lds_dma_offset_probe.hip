// Does the instruction offset of global_load_lds_dwordx4 move the LDS destination, the global source, or both?
// (gfx950; decides whether one M0 write can serve several 1-KiB LDS-DMA requests of a stage.)
// build: hipcc --offload-arch=gfx950 -O2 -o lds_dma_offset_probe lds_dma_offset_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void probe(const uint32_t* src, uint32_t* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t* l = (uint32_t*)smem;
    for (int i = threadIdx.x; i < 2048; i += 64) l[i] = 0xdeadbeefu;
    __syncthreads();
    const uint32_t lds0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;
    const char* g = (const char*)src + threadIdx.x * 16;
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off offset:1024\n\ts_waitcnt vmcnt(0)" ::"v"(g), "s"(lds0) : "memory", "m0");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 64) out[i] = l[i];
}

int main() {
    std::vector<uint32_t> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = i;          // dword i holds i
    uint32_t *d, *o;
    hipMalloc(&d, 4096 * 4); hipMalloc(&o, 2048 * 4);
    hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 8192, 0, d, o);
    std::vector<uint32_t> r(2048);
    hipMemcpy(r.data(), o, 2048 * 4, hipMemcpyDeviceToHost);
    int first = -1;
    for (int i = 0; i < 2048; ++i) if (r[i] != 0xdeadbeefu) { first = i; break; }
    if (first < 0) { printf("nothing landed\n"); return 1; }
    printf("first written LDS dword %d (byte %d) holds source dword %u (byte %u)\n", first, first * 4, r[first], r[first] * 4);
    printf("=> instruction offset 1024: LDS destination moved by %d bytes, global source by %u bytes\n", first * 4, r[first] * 4);
    return 0;
}
