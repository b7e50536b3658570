// development microbenchmark: can one wave's VALU stream and another wave's MFMA stream share a SIMD at full rate?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ROLE of waves 4..7 (waves 0..3 always run the MFMA loop): 0 = idle (exit), 1 = plain VALU (fma), 2 = v_exp, 3 = v_cvt_pk + max3, 4 = MFMA too
template <int ROLE>
__global__ __launch_bounds__(512) void k(float* out, int iters, float seed, float* cyc) {
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    const long long t0 = __builtin_readcyclecounter();
    if (wave < 4 || ROLE == 4) {
        f32x16 acc[4];
        f16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (f16)(seed + i); b[i] = (f16)(seed * 0.5f + i); }
        for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 20; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i & 3], 0, 0, 0);
        }
        for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) r += acc[j][i];
    } else if (ROLE != 0) {
        float v[32];
        for (int i = 0; i < 32; ++i) v[i] = seed * (i + 1) * 1e-3f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 3; ++rep)
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    if (ROLE == 1) v[i] = fmaf(v[i], 0.999f, 1e-6f);
                    if (ROLE == 2) v[i] = __builtin_amdgcn_exp2f(v[i]) * 0.5f;
                    if (ROLE == 3) v[i] = fmaxf(fmaxf(v[i], v[(i + 1) & 31]), 0.25f * v[(i + 7) & 31]);
                }
        }
        for (int i = 0; i < 32; ++i) r += v[i];
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 512 + threadIdx.x] = r;
    if (blockIdx.x == 7 && (threadIdx.x & 63) == 0) cyc[wave] = (float)(t1 - t0) / iters;
}
template <int ROLE>
void run(const char* name, float* d) {
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float* cyc; hipMalloc(&cyc, 8 * sizeof(float)); hipMemset(cyc, 0, 32);
    k<ROLE><<<256, 512>>>(d, 10, 1.0f, cyc);
    hipEventRecord(e0);
    k<ROLE><<<256, 512>>>(d, iters, 1.0f, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    float h[8]; hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
    printf("%-44s: %7.3f us per iteration; cycles per iteration: MFMA wave %6.0f, partner wave %6.0f\n", name, ms * 1e3 / iters, h[0], h[4]);
}
int main() {
    float* d; hipMalloc(&d, 256 * 512 * sizeof(float));
    run<0>("MFMA wave alone on its SIMD", d);
    run<4>("MFMA wave + MFMA wave", d);
    run<1>("MFMA wave + 96 v_fma wave", d);
    run<2>("MFMA wave + 96 (v_exp + v_mul) wave", d);
    run<3>("MFMA wave + 96 x (2 v_max/v_fma) wave", d);
    return 0;
}
