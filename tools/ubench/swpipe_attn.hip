// development microbenchmark: does software-pipelining INSIDE a wave — the first-product MFMAs of key tile t+1 issued
// between the softmax VALU work of tile t — hide the VALU phase?  Register-only model of the attention wave-tile
// (8 QK MFMA + 12 PV MFMA, 16 max3 + 32 exp + 16 cvt), 2 or 3 waves per SIMD.
//   build: hipcc --offload-arch=gfx950 -O3 -o swpipe swpipe_attn.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 mfma(const bf16x8& a, const bf16x8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// MODE 0: [QK(t)][softmax(t)][PV(t)] in program order (the product kernel's structure)
// MODE 1: QK(t+1) MFMAs interleaved with softmax(t) by sched_group_barrier, then PV(t)
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    f32x16 o[3], sc[2], scn[2];
    bf16x8 kf[4], vf[4], q[4], pf[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 8; ++e) { kf[i][e] = (bf16)(seed * (i + 1) * 0.01f + e * 0.001f); vf[i] = kf[i]; q[i][e] = (bf16)(0.02f * e); pf[i] = kf[i]; }
    for (int j = 0; j < 3; ++j) for (int i = 0; i < 16; ++i) o[j][i] = 0.f;
    f32x16 cneg;
    for (int i = 0; i < 16; ++i) cneg[i] = -seed;
    auto qk = [&](f32x16 (&s)[2]) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            s[0] = mfma(kf[ks], q[ks], ks ? s[0] : cneg);
            s[1] = mfma(kf[(ks + 1) & 3], q[ks], ks ? s[1] : cneg);
        }
    };
    auto softmax = [&](const f32x16 (&s)[2]) {
        float m = fmaxf(s[0][0], s[0][1]);
#pragma unroll
        for (int i = 1; i < 16; ++i) m = fmaxf(fmaxf(m, s[i >> 3][(2 * i) & 15]), s[i >> 3][(2 * i + 1) & 15]);
        if (__any(m > 1e30f)) cneg[0] += 1.f;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                f32x8 pv;
#pragma unroll
                for (int e = 0; e < 8; ++e) pv[e] = __builtin_amdgcn_exp2f(s[b][8 * u + e]);
                pf[2 * b + u] = __builtin_convertvector(pv, bf16x8);
            }
    };
    auto pv = [&]() {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int d = 0; d < 3; ++d) o[d] = mfma(vf[(kk + d) & 3], pf[kk], o[d]);
    };
    if (MODE == 0) {
        for (int it = 0; it < iters; ++it) {
            qk(sc);
            __builtin_amdgcn_sched_barrier(0);
            softmax(sc);
            pv();
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        qk(sc);
        for (int it = 0; it < iters; ++it) {
            __builtin_amdgcn_sched_barrier(0);
            qk(scn);                 // tile t+1
            softmax(sc);             // tile t
            // interleave: per MFMA of the first product ~7 VALU / transcendental ops of the softmax
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);   // 8 VALU
            }
            __builtin_amdgcn_sched_barrier(0);
            pv();
            sc[0] = scn[0]; sc[1] = scn[1];
        }
    }
    float r = 0.f;
    for (int j = 0; j < 3; ++j) for (int i = 0; i < 16; ++i) r += o[j][i];
    for (int j = 0; j < 2; ++j) for (int i = 0; i < 16; ++i) r += sc[j][i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MODE>
void run(const char* name, float* d, int wgs_per_cu) {
    const int iters = 4000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<grid, 256>>>(d, 10, 1.0f);
    hipEventRecord(e0);
    k<MODE><<<grid, 256>>>(d, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s waves/SIMD=%d: %7.3f us per wave-tile (20 MFMA floor 0.32 us @2 GHz)\n", name, wgs_per_cu, ms * 1e3 / iters / wgs_per_cu);
}
int main() {
    float* d; hipMalloc(&d, 256 * 4 * 256 * sizeof(float));
    for (int w = 1; w <= 3; ++w) { run<0>("program order QK | softmax | PV", d, w); run<1>("QK(t+1) interleaved with softmax(t)", d, w); }
    return 0;
}
