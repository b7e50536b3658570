// development microbenchmark (round 3): issue cadence of v_mfma_f32_32x32x16_bf16 as a function of who issues it.
//   MODE 0  every wave issues 16-MFMA bursts back to back on NACC rotating accumulators, no barrier   (1, 2 or 3 waves / SIMD)
//   MODE 1  two groups of four waves alternate 16-MFMA slots separated by s_barrier (the ping-pong GEMM's skeleton):
//           [MFMA slot | barrier | idle slot | barrier], the second group one barrier behind
//   MODE 2  as 1 but the slots are 32 MFMAs long
//   MODE 3  no alternation: both groups issue their 16 MFMAs in the SAME slot, one barrier per slot
// Reports shader cycles (s_memtime) per MFMA per SIMD.  build: hipcc --offload-arch=gfx950 -O3 -o mfma_cadence mfma_cadence.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int NACC, int NW>
__global__ __launch_bounds__(NW * 64) void k(float* out, long long* cyc, int iters, float seed) {
    f32x16 acc[NACC];
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 8; ++e) {
            a[i][e] = (bf16)(seed * (i + 1) * 0.013f + e * 0.001f + threadIdx.x * 1e-4f);
            b[i][e] = (bf16)(0.3f - e * 0.01f + i * 0.02f);
        }
    for (int j = 0; j < NACC; ++j)
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave >> 2;
    constexpr int SLOT = MODE == 2 ? 32 : 16;
    __syncthreads();
    const long long t0 = clock64();
    if (MODE == 1 || MODE == 2) {
        if (grp == 1) __builtin_amdgcn_s_barrier();
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < SLOT; ++i)
            acc[i % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[(i >> 2) & 3], acc[i % NACC], 0, 0, 0);
        if (MODE >= 1) {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 1 || MODE == 2) {
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (MODE == 1 || MODE == 2) {
        if (grp == 0) __builtin_amdgcn_s_barrier();
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int j = 0; j < NACC; ++j)
        for (int i = 0; i < 16; ++i) s += acc[j][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * NW + wave] = t1 - t0;
}

template <int MODE, int NACC, int NW>
void run(const char* tag) {
    const int blocks = 256, iters = 2000;
    float* out; long long* cyc;
    hipMalloc(&out, blocks * NW * 64 * sizeof(float));
    hipMalloc(&cyc, blocks * NW * sizeof(long long));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, NACC, NW><<<blocks, NW * 64>>>(out, cyc, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE, NACC, NW><<<blocks, NW * 64>>>(out, cyc, iters, 1.0f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    static long long h[256 * 16];
    hipMemcpy(h, cyc, blocks * NW * sizeof(long long), hipMemcpyDeviceToHost);
    long long mx = 0; for (int i = 0; i < blocks * NW; ++i) if (h[i] > mx) mx = h[i];
    constexpr int SLOT = MODE == 2 ? 32 : 16;
    const double per_simd = (double)iters * SLOT * (NW / 4);           // MFMAs per SIMD
    printf("%-58s %7.2f cycles / MFMA / SIMD   wall %7.1f us   (%.2f GHz eff, %6.1f TF/s)\n", tag, mx / per_simd, ms * 1e3,
           mx / (ms * 1e6), 256.0 * 4 * per_simd * 32768.0 / (ms * 1e9));
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0, 4, 4>("1 wave/SIMD, back to back, 4 acc");
    run<0, 8, 4>("1 wave/SIMD, back to back, 8 acc");
    run<0, 2, 4>("1 wave/SIMD, back to back, 2 acc");
    run<0, 4, 8>("2 waves/SIMD, both issuing, 4 acc");
    run<0, 8, 8>("2 waves/SIMD, both issuing, 8 acc");
    run<0, 4, 12>("3 waves/SIMD, all issuing, 4 acc");
    run<1, 4, 8>("ping-pong 16-MFMA slots + barriers, 4 acc");
    run<1, 8, 8>("ping-pong 16-MFMA slots + barriers, 8 acc");
    run<2, 8, 8>("ping-pong 32-MFMA slots + barriers, 8 acc");
    run<3, 4, 8>("both groups same slot, barrier per 16, 4 acc");
    run<3, 8, 8>("both groups same slot, barrier per 16, 8 acc");
    return 0;
}
