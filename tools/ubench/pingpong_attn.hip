// development microbenchmark: do the MFMA phase of one wave group and the softmax VALU phase of the other overlap on a
// SIMD when the two groups of a workgroup run one barrier apart?   build: hipcc --offload-arch=gfx950 -O3 -o pp pingpong_attn.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int NMFMA>   // MODE 0: every wave [M][V] no barriers; 1: barriers, all waves same phase; 2: staggered groups
__global__ __launch_bounds__(512) void k(float* out, int iters, float seed) {
    const int wave = threadIdx.x >> 6;
    const int grp = wave >> 2;
    f32x16 acc[4], sc[2];
    f16x8 a, b, p[4];
    for (int i = 0; i < 8; ++i) { a[i] = (f16)(seed + i); b[i] = (f16)(seed * 0.5f + i); }
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    for (int j = 0; j < 2; ++j) for (int i = 0; i < 16; ++i) sc[j][i] = seed * i;
    for (int j = 0; j < 4; ++j) p[j] = a;
    auto mphase = [&]() {
#pragma unroll
        for (int i = 0; i < NMFMA; ++i) {
            if (i < 8) sc[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, sc[i & 1], 0, 0, 0);
            else       acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p[i & 3], b, acc[i & 3], 0, 0, 0);
        }
    };
    auto vphase = [&]() {
        float m = sc[0][0];
#pragma unroll
        for (int i = 0; i < 16; ++i) m = fmaxf(m, fmaxf(sc[0][i], sc[1][i]));
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                float e0 = sc[j][i] - m * 1e-9f, e1 = sc[j][i + 1] - m * 1e-9f;
                if (MODE != 5) e0 = __builtin_amdgcn_exp2f(e0);
                if (MODE != 5 && MODE != 4) e1 = __builtin_amdgcn_exp2f(e1);
                p[(j * 2 + (i >> 3)) & 3][(i & 7)] = (f16)e0;
                p[(j * 2 + (i >> 3)) & 3][(i & 7) + 1] = (f16)e1;
            }
    };
    auto bar = [&]() { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); };
    if (MODE == 2 && grp == 1) bar();
    for (int it = 0; it < iters; ++it) {
        mphase();
        if (MODE) bar(); else __builtin_amdgcn_sched_barrier(0);
        if (MODE != 3) vphase();
        if (MODE) bar(); else __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE == 2 && grp == 0) bar();
    float r = 0.f;
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) r += acc[j][i];
    for (int j = 0; j < 2; ++j) for (int i = 0; i < 16; ++i) r += sc[j][i];
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int MODE, int NMFMA>
void run(const char* name, float* d, int wgs_per_cu) {
    const int iters = 2000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, NMFMA><<<grid, 512>>>(d, 10, 1.0f);
    hipEventRecord(e0);
    k<MODE, NMFMA><<<grid, 512>>>(d, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 2 * wgs_per_cu waves, each iters tiles
    const double us_per_wave_tile = ms * 1e3 / iters / (2.0 * wgs_per_cu);
    printf("%-34s NMFMA=%2d wg/cu=%d: %7.3f us per wave-tile  (MFMA floor %.3f us @2.0GHz)\n", name, NMFMA, wgs_per_cu, us_per_wave_tile, NMFMA * 32 / 2.0e3);
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 512 * sizeof(float));
    run<0, 20>("free-running", d, 1); run<2, 20>("barriers, staggered groups", d, 1);
    run<0, 0>("VALU phase only", d, 1); run<0, 0>("VALU phase only", d, 2);
    run<3, 20>("MFMA phase only", d, 1); run<3, 20>("MFMA phase only", d, 2);
    run<4, 20>("free-running, half the exps", d, 1);
    run<5, 20>("free-running, no exps (max+cvt)", d, 1);
    run<0, 20>("free-running", d, 3);
    return 0;
}
