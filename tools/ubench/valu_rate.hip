// development microbenchmark (round 3): issue rate of the VALU instructions of the attention V slot, alone and beside a SIMD
// partner that keeps the matrix pipe busy.
//   OP 0  v_exp_f32          OP 1  v_fma_f32          OP 2  v_cvt_pk_bf16_f32 (two inputs -> one packed)          OP 3  v_max3_f32
//   PARTNER 0: every wave runs the VALU stream (NW waves per SIMD)     PARTNER 1: waves 0-3 run the VALU stream, waves 4-7
//   (their SIMD partners) loop v_mfma_f32_32x32x16_bf16 back to back
// Reports shader cycles (s_memtime / clock64) per VALU instruction per wave.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int OP, int PARTNER, int NW>
__global__ __launch_bounds__(NW * 64) void k(float* out, long long* cyc, int iters, float seed) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float x[32];
    for (int i = 0; i < 32; ++i) x[i] = seed * 0.001f * (i + 1) + threadIdx.x * 1e-6f;
    f32x16 acc[4];
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (bf16)(seed * 0.01f + e * 0.001f); b[e] = (bf16)(0.3f - e * 0.01f); }
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    __syncthreads();
    const long long t0 = clock64();
    if (PARTNER == 1 && wave >= 4) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 20; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                if (OP == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
                if (OP == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(x[(i + 1) & 31]));
                if (OP == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x[i]) : "v"(x[(i + 1) & 31]));
                if (OP == 3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(x[(i + 1) & 31]), "v"(x[(i + 2) & 31]));
            }
        }
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 32; ++i) s += x[i];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 16; ++i) s += acc[j][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * NW + wave] = t1 - t0;
}

template <int OP, int PARTNER, int NW>
void run(const char* tag) {
    const int blocks = 256, iters = 2000;
    float* out;
    long long* cyc;
    hipMalloc(&out, blocks * NW * 64 * sizeof(float));
    hipMalloc(&cyc, blocks * NW * sizeof(long long));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<OP, PARTNER, NW>), dim3(blocks), dim3(NW * 64), 0, 0, out, cyc, iters, 1.0f);
    hipDeviceSynchronize();
    static long long h[256 * 16];
    hipMemcpy(h, cyc, blocks * NW * sizeof(long long), hipMemcpyDeviceToHost);
    double valu = 0, mf = 0;
    int nv = 0, nm = 0;
    for (int bl = 0; bl < blocks; ++bl)
        for (int w = 0; w < NW; ++w) {
            if (PARTNER == 1 && w >= 4) { mf += (double)h[bl * NW + w] / (iters * 20.0); ++nm; }
            else                        { valu += (double)h[bl * NW + w] / (iters * 32.0); ++nv; }
        }
    if (PARTNER == 1) printf("%-34s %6.2f clock64 ticks per VALU instruction      partner: %6.2f per MFMA\n", tag, valu / nv, mf / nm);
    else              printf("%-34s %6.2f clock64 ticks per VALU instruction per wave (%d waves per SIMD)\n", tag, valu / nv, NW / 4);
    hipFree(out);
    hipFree(cyc);
}

int main() {
    // clock64 = s_memtime ticks at the constant 100 MHz reference on gfx950?  calibrate against the MFMA: 32x32x16 = 32 shader cycles
    run<1, 1, 8>("(calibration) v_fma beside MFMA");
    run<0, 0, 4>("v_exp_f32, alone, 1 wave/SIMD");
    run<0, 0, 8>("v_exp_f32, alone, 2 waves/SIMD");
    run<1, 0, 4>("v_fma_f32, alone, 1 wave/SIMD");
    run<1, 0, 8>("v_fma_f32, alone, 2 waves/SIMD");
    run<2, 0, 4>("v_cvt_pk_bf16_f32, alone");
    run<3, 0, 4>("v_max3_f32, alone");
    run<0, 1, 8>("v_exp_f32 beside MFMA partner");
    run<1, 1, 8>("v_fma_f32 beside MFMA partner");
    run<2, 1, 8>("v_cvt_pk_bf16_f32 beside MFMA");
    run<3, 1, 8>("v_max3_f32 beside MFMA");
    return 0;
}
