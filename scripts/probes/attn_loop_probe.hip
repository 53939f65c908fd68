// What bounds the resident attention kernel's tile loop?  One workgroup per CU, NW waves, K / V images in LDS (no global
// memory traffic inside the timed loop), every wave runs the software-pipelined step of attn_res_kernel (S(k+1) || E(k) ||
// PV(k-1), head_dim 64, one query tile) ITERS times.  Template knobs remove one ingredient at a time:
//   MF: the MFMAs   EX: the exponentials (replaced by a mov)   VA: the other VALU of a slice (adds, converts)
//   DS: the fragment reads (fragments stay in registers)   SB: the sched_barrier pinning
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -o p attn_loop_probe.hip && ./p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_hw;
#define DEVINL __device__ __forceinline__
DEVINL uint32_t pack2(float lo, float hi) { f32x2 v = {lo, hi}; bf16x2_hw b = __builtin_convertvector(v, bf16x2_hw); return *reinterpret_cast<uint32_t*>(&b); }

template <bool MF, bool EX, bool VA, bool DS, bool SB>
DEVINL void step(const char* kl, const char* vl, const bf16x8 (&qf)[4], f32x16& s_out, f32x16& s_in, bf16x8 (&p_out)[2], const bf16x8 (&p_in)[2],
                 f32x16 (&ot)[2], f32x2 (&lsum)[2], bf16x8 (&kf)[4], bf16x8 (&vf)[2][2], int l31, int half, int kswz, const int (&vra)[2], const int (&vrb)[2]) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    uint32_t pk[8];
    auto slice = [&](int j) {
        float a = s_in[2 * j], b = s_in[2 * j + 1];
        float pa, pb;
        if (EX) { pa = __builtin_amdgcn_exp2f(a); pb = __builtin_amdgcn_exp2f(b); }
        else { pa = a; pb = b; asm volatile("v_mov_b32 %0, %0\n\tv_mov_b32 %1, %1" : "+v"(pa), "+v"(pb)); if (false) {} }
        if (VA) {
            lsum[j & 1] += (f32x2){pa, pb};
            uint32_t w = pack2(pa, pb);
            asm volatile("" : "+v"(w));
            pk[j] = w;
        } else {
            pk[j] = __float_as_uint(pa) ^ __float_as_uint(pb);
            asm volatile("" : "+v"(pk[j]));
        }
    };
    auto vread = [&](int s2, int dt) {
        if (!DS) return;
        const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vl + vra[dt] + s2 * (16 * 128)));
        const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vl + vrb[dt] + s2 * (16 * 128)));
        vf[s2][dt] = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    constexpr int PD = 4;
    auto request = [&](int m) {
        if (m < 4) { if (DS) kf[m] = *(const bf16x8*)(kl + l31 * 128 + (((2 * m + half) ^ kswz) << 4)); }
        else if (m < 8) vread((m - 4) >> 1, (m - 4) & 1);
    };
#pragma unroll
    for (int m = 0; m < PD; ++m) request(m);
    slice(0);
    if (SB) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        request(m + PD);
        if (MF) {
            if (m < 4) s_out = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[m], qf[m], m == 0 ? zero : s_out, 0, 0, 0);
            else { const int i = m - 4; ot[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[i >> 1][i & 1], p_in[i >> 1], ot[i & 1], 0, 0, 0); }
        }
        if (m < 7) slice(m + 1);
        if (SB) __builtin_amdgcn_sched_barrier(0);
    }
    const uint4 lo = {pk[0], pk[1], pk[2], pk[3]}, hi = {pk[4], pk[5], pk[6], pk[7]};
    p_out[0] = *reinterpret_cast<const bf16x8*>(&lo);
    p_out[1] = *reinterpret_cast<const bf16x8*>(&hi);
}

template <bool MF, bool EX, bool VA, bool DS, bool SB>
__global__ void __launch_bounds__(768, 1) probe(float* out, int iters, int nsub) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    for (int i = tid; i < nsub * 2 * 4096 / 4; i += blockDim.x) ((uint32_t*)smem)[i] = 0x3c003c00u + (i * 2654435761u >> 20 & 0x00ff00ff);   // small finite bf16 pairs
    __syncthreads();
    const char* kimg = smem; const char* vimg = smem + nsub * 4096;
    const int kswz = (lane >> 1) & 7;
    const int tr_u = lane & 15, tr_g = (lane >> 4) & 1;
    int vra[2], vrb[2];
    for (int dt = 0; dt < 2; ++dt) {
        const int dcol = dt * 32 + tr_g * 16 + (tr_u & 3) * 4;
        const int keyA = half * 4 + (tr_u >> 2), keyB = keyA + 8;
        vra[dt] = keyA * 128 + (((dcol >> 3) ^ (((keyA >> 1) & 1) << 2)) << 4) + ((dcol & 7) << 1);
        vrb[dt] = keyB * 128 + (((dcol >> 3) ^ (((keyB >> 1) & 1) << 2)) << 4) + ((dcol & 7) << 1);
    }
    bf16x8 qf[4], kf[4], vf[2][2], pa[2], pb[2];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) { qf[i][e] = (short)(0x3c00 + ((lane * 7 + i * 3 + e) & 63)); kf[i][e] = (short)(0x3c00 + ((lane + e + i) & 31)); }
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int e = 0; e < 8; ++e) vf[a][b][e] = (short)(0x3c00 + ((lane * 3 + e + a + b) & 31));
    for (int a = 0; a < 2; ++a) for (int e = 0; e < 8; ++e) { pa[a][e] = (short)0x3c00; pb[a][e] = (short)0x3c00; }
    f32x16 sa, sb, ot[2];
    for (int r = 0; r < 16; ++r) { sa[r] = sb[r] = -3.f + 0.01f * r; ot[0][r] = ot[1][r] = 0.f; }
    f32x2 lsum[2] = {{0.f, 0.f}, {0.f, 0.f}};
    for (int it = 0; it < iters; ++it) {
        int k = 1;
        for (; k + 2 < nsub; k += 2) {
            step<MF, EX, VA, DS, SB>(kimg + (k + 1) * 4096, vimg + (k - 1) * 4096, qf, sa, sb, pb, pa, ot, lsum, kf, vf, l31, half, kswz, vra, vrb);
            step<MF, EX, VA, DS, SB>(kimg + (k + 2) * 4096, vimg + k * 4096, qf, sb, sa, pa, pb, ot, lsum, kf, vf, l31, half, kswz, vra, vrb);
        }
    }
    float s = lsum[0][0] + lsum[0][1] + lsum[1][0] + lsum[1][1];
    for (int r = 0; r < 16; ++r) s += ot[0][r] + ot[1][r] + sa[r] + sb[r];
    for (int e = 0; e < 8; ++e) s += (float)pa[0][e] + (float)pb[1][e];
    out[blockIdx.x * blockDim.x + tid] = s;
}


// the compiler-scheduled 64-key tile body (S, then E, then PV of the SAME tile; no software pipeline, no pinning)
DEVINL void simple_tile(const char* kl, const char* vl, const bf16x8 (&qf)[4], f32x16 (&ot)[2], f32x2 (&lsum)[2],
                        int l31, int half, int kswz, const int (&vra)[2], const int (&vrb)[2]) {
    f32x16 st[2];
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    bf16x8 kf[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kf[j][ks] = *(const bf16x8*)(kl + (j * 32 + l31) * 128 + (((2 * ks + half) ^ kswz) << 4));
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) st[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[j][ks], qf[ks], ks == 0 ? zero : st[j], 0, 0, 0);
    bf16x8 pf[4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            float pv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) pv[e] = __builtin_amdgcn_exp2f(st[j][s2 * 8 + e]);
#pragma unroll
            for (int e = 0; e < 8; e += 2) lsum[(e >> 1) & 1] += (f32x2){pv[e], pv[e + 1]};
            const uint4 pk = {pack2(pv[0], pv[1]), pack2(pv[2], pv[3]), pack2(pv[4], pv[5]), pack2(pv[6], pv[7])};
            pf[j * 2 + s2] = *reinterpret_cast<const bf16x8*>(&pk);
        }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vl + vra[dt] + s * (16 * 128)));
            const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vl + vrb[dt] + s * (16 * 128)));
            const bf16x8 vf = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
            ot[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[s], ot[dt], 0, 0, 0);
        }
}
__global__ void __launch_bounds__(768, 1) probe_simple(float* out, int iters, int nsub) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    for (int i = tid; i < nsub * 2 * 4096 / 4; i += blockDim.x) ((uint32_t*)smem)[i] = 0x3c003c00u + (i * 2654435761u >> 20 & 0x00ff00ff);
    __syncthreads();
    const char* kimg = smem; const char* vimg = smem + nsub * 4096;
    const int kswz = (lane >> 1) & 7;
    const int tr_u = lane & 15, tr_g = (lane >> 4) & 1;
    int vra[2], vrb[2];
    for (int dt = 0; dt < 2; ++dt) {
        const int dcol = dt * 32 + tr_g * 16 + (tr_u & 3) * 4;
        const int keyA = half * 4 + (tr_u >> 2), keyB = keyA + 8;
        vra[dt] = keyA * 128 + (((dcol >> 3) ^ (((keyA >> 1) & 1) << 2)) << 4) + ((dcol & 7) << 1);
        vrb[dt] = keyB * 128 + (((dcol >> 3) ^ (((keyB >> 1) & 1) << 2)) << 4) + ((dcol & 7) << 1);
    }
    bf16x8 qf[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) qf[i][e] = (short)(0x3c00 + ((lane * 7 + i * 3 + e) & 63));
    f32x16 ot[2];
    for (int r = 0; r < 16; ++r) ot[0][r] = ot[1][r] = 0.f;
    f32x2 lsum[2] = {{0.f, 0.f}, {0.f, 0.f}};
    for (int it = 0; it < iters; ++it)
        for (int kt = 0; kt < 8; ++kt) simple_tile(kimg + kt * 8192, vimg + kt * 8192, qf, ot, lsum, l31, half, kswz, vra, vrb);
    float s = lsum[0][0] + lsum[0][1] + lsum[1][0] + lsum[1][1];
    for (int r = 0; r < 16; ++r) s += ot[0][r] + ot[1][r];
    out[blockIdx.x * blockDim.x + tid] = s;
}
void run_simple(float* out, int nw) {
    const int iters = 200, nsub = 18;
    const size_t lds = (size_t)nsub * 2 * 4096;
    (void)hipFuncSetAttribute((const void*)probe_simple, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe_simple<<<256, nw * 64, lds>>>(out, 20, nsub);
    (void)hipEventRecord(e0);
    probe_simple<<<256, nw * 64, lds>>>(out, iters, nsub);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double steps = (double)iters * 16;          // 8 tiles of 64 keys = 16 sub-tile steps
    printf("%-44s waves/SIMD=%d  ns per wave-step = %7.1f   ns per MFMA slot on a SIMD = %6.2f\n", "simple 64-key tile body (compiler order)", nw / 4,
           ms * 1e6 / steps, ms * 1e6 / (steps * 8 * (nw / 4.0)));
}

template <bool MF, bool EX, bool VA, bool DS, bool SB>
void run(const char* name, float* out, int nw) {
    const int iters = 200, nsub = 18;
    const size_t lds = (size_t)nsub * 2 * 4096;
    (void)hipFuncSetAttribute((const void*)probe<MF, EX, VA, DS, SB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe<MF, EX, VA, DS, SB><<<256, nw * 64, lds>>>(out, 20, nsub);
    (void)hipEventRecord(e0);
    probe<MF, EX, VA, DS, SB><<<256, nw * 64, lds>>>(out, iters, nsub);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double steps = (double)iters * 16;          // steps per wave (k = 1 .. 16)
    const double ns_per_step = ms * 1e6 / steps;      // wall time per step of one wave (all waves run concurrently)
    const double mfma_per_simd = steps * 8 * (nw / 4.0);
    printf("%-44s waves/SIMD=%d  ns per wave-step = %7.1f   ns per MFMA slot on a SIMD = %6.2f  (17.2 = pipe-bound)%s\n", name, nw / 4, ns_per_step,
           ms * 1e6 / mfma_per_simd, hipGetLastError() == hipSuccess ? "" : "  ERROR");
}

int main() {
    float* out; (void)hipMalloc(&out, 256 * 1024 * 4);
    for (int nw : {4, 8, 12}) {
        run_simple(out, nw);
        run<true, true, true, true, true>("full step", out, nw);
        run<true, true, true, true, false>("full step, no sched_barrier pinning", out, nw);
        run<true, false, true, true, true>("no exponentials (mov)", out, nw);
        run<true, true, false, true, true>("no adds / converts", out, nw);
        run<true, false, false, true, true>("no exp, no adds / converts", out, nw);
        run<true, true, true, false, true>("no fragment reads", out, nw);
        run<false, true, true, true, true>("no MFMAs", out, nw);
        run<true, false, false, false, true>("MFMAs only (+ movs)", out, nw);
        run<false, true, true, false, true>("VALU only", out, nw);
    }
    return 0;
}
