// development microbenchmark (round 2): can ONE wave overlap its own MFMAs with its own softmax VALU work when the source
// order is pinned (sched_barrier(0) between hand-placed groups)?  Register-only model of the d = 64 attention wave-tile
// (32 query rows x 64 keys): 8 QK MFMAs (2 chains), 12 PV MFMAs (3 accumulators), 16 max3 + 32 exp + 16 cvt_pk.
//   MODE 0  MFMAs only (floor)                         MODE 1  VALU only
//   MODE 2  program order  QK(t) | softmax(t) | PV(t)   (the product kernel's structure)
//   MODE 3  cross-tile software pipeline, order pinned:  PV(t-1) MFMAs with max3(t) + first half of exp(t) between them,
//           QK(t+1) MFMAs with the second half of exp(t) between them (FPG fillers per MFMA gap)
//   build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 -o overlap overlap.hip ; ./overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

#define SB() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ f32x16 mfma(const bf16x8& a, const bf16x8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8 cvt(const f32x8& x) { return __builtin_convertvector(x, bf16x8); }

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    f32x16 o[3], sc[2], scn[2];
    bf16x8 kf[4], vf[4], q[4], pf[4], pfn[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 8; ++e) {
            kf[i][e] = (bf16)(seed * (i + 1) * 0.013f + e * 0.001f + threadIdx.x * 1e-4f);
            vf[i][e] = (bf16)(0.3f - e * 0.01f + i * 0.02f);
            q[i][e] = (bf16)(0.02f * e - 0.05f);
            pf[i][e] = (bf16)(0.1f * (e + 1));
        }
    for (int j = 0; j < 3; ++j) for (int i = 0; i < 16; ++i) o[j][i] = 0.f;
    f32x16 cneg;
    for (int i = 0; i < 16; ++i) cneg[i] = -seed;
    float macc = 0.f;
    for (int b = 0; b < 2; ++b) for (int i = 0; i < 16; ++i) sc[b][i] = 0.01f * i - b;

    auto qk_all = [&](f32x16 (&s)[2]) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            s[0] = mfma(kf[ks], q[ks], ks ? s[0] : cneg);
            s[1] = mfma(kf[(ks + 1) & 3], q[ks], ks ? s[1] : cneg);
        }
    };
    auto maxchain = [&](const f32x16 (&s)[2]) {
        float m = fmaxf(s[0][0], s[0][1]);
#pragma unroll
        for (int i = 1; i < 16; ++i) m = fmaxf(fmaxf(m, s[i >> 3][(2 * i) & 15]), s[i >> 3][(2 * i + 1) & 15]);
        return m;
    };
    auto expq = [&](const f32x16 (&s)[2], bf16x8 (&p)[4], int qd) {          // quarter qd of the 32 exponentials (8 exp + 4 cvt)
        f32x8 pv;
#pragma unroll
        for (int e = 0; e < 8; ++e) pv[e] = __builtin_amdgcn_exp2f(s[qd >> 1][8 * (qd & 1) + e]);
        p[qd] = cvt(pv);
    };
    auto pv_all = [&](const bf16x8 (&p)[4]) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int d = 0; d < 3; ++d) o[d] = mfma(vf[(kk + d) & 3], p[kk], o[d]);
    };

    if (MODE == 0) {
        for (int it = 0; it < iters; ++it) { qk_all(sc); SB(); pv_all(pf); SB(); }
    } else if (MODE == 1) {
        for (int it = 0; it < iters; ++it) {
            macc += maxchain(sc);
            for (int qd = 0; qd < 4; ++qd) expq(sc, pf, qd);
            sc[0][0] += (float)pf[0][0] + (float)pf[1][1] + (float)pf[2][2] + (float)pf[3][3];
            SB();
        }
    } else if (MODE == 2) {
        for (int it = 0; it < iters; ++it) {
            qk_all(sc); SB();
            macc += maxchain(sc);
            if (__any(macc > 1e30f)) cneg[0] += 1.f;
            for (int qd = 0; qd < 4; ++qd) expq(sc, pf, qd);
            SB();
            pv_all(pf); SB();
        }
    } else {
        // steady state: sa = scores of tile t (QK done in the previous half-iteration), pa = P of tile t-1.
        // Two half-iterations per loop trip with the register sets swapped (no copies).
        f32x16 sa[2], sb[2];
        bf16x8 pa[4], pb[4];
        for (int i = 0; i < 4; ++i) pa[i] = pf[i];
        qk_all(sa);
        // phase A: PV(P_prev) (12 MFMAs) with max3(S_cur) + exp/cvt of the first 16 scores between them -> P_next[0..1]
        // phase B: QK -> S_next (8 MFMAs) with exp/cvt of the other 16 scores between them -> P_next[2..3]
        auto half = [&](f32x16 (&s_cur)[2], f32x16 (&s_next)[2], bf16x8 (&p_prev)[4], bf16x8 (&p_next)[4]) __attribute__((always_inline)) {
            float m = fmaxf(s_cur[0][0], s_cur[0][1]);
            f32x8 ea, eb;
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                o[i % 3] = mfma(vf[((i / 3) + (i % 3)) & 3], p_prev[i / 3], o[i % 3]);
                SB();
                if (i < 5) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const int u = 1 + 3 * i + j;
                        m = fmaxf(fmaxf(m, s_cur[u >> 3][(2 * u) & 15]), s_cur[u >> 3][(2 * u + 1) & 15]);
                    }
                } else if (i < 9) {                                  // 4 gaps x 4 exp
                    const int g = i - 5;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int e = 4 * g + j;
                        if (e < 8) ea[e] = __builtin_amdgcn_exp2f(s_cur[0][e]); else eb[e - 8] = __builtin_amdgcn_exp2f(s_cur[0][e]);
                    }
                } else if (i == 9) {
                    p_next[0] = cvt(ea);
                } else if (i == 10) {
                    p_next[1] = cvt(eb);
                }
                SB();
            }
            macc += m;
            if (__any(m > 1e30f)) cneg[0] += 1.f;
            SB();
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int ks = i >> 1;
                if (i & 1) s_next[1] = mfma(kf[(ks + 1) & 3], q[ks], ks ? s_next[1] : cneg);
                else       s_next[0] = mfma(kf[ks], q[ks], ks ? s_next[0] : cneg);
                SB();
                if (i < 4) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int e = 4 * i + j;
                        if (e < 8) ea[e] = __builtin_amdgcn_exp2f(s_cur[1][e]); else eb[e - 8] = __builtin_amdgcn_exp2f(s_cur[1][e]);
                    }
                } else if (i == 4) {
                    p_next[2] = cvt(ea);
                } else if (i == 5) {
                    p_next[3] = cvt(eb);
                }
                SB();
            }
        };
        for (int it = 0; it < iters; it += 2) {
            half(sa, sb, pa, pb);
            half(sb, sa, pb, pa);
        }
        sc[0] = sa[0]; sc[1] = sa[1];
        for (int i = 0; i < 4; ++i) pf[i] = pa[i];
    }
    float r = macc;
    for (int j = 0; j < 3; ++j) for (int i = 0; i < 16; ++i) r += o[j][i];
    for (int j = 0; j < 2; ++j) for (int i = 0; i < 16; ++i) r += sc[j][i];
    for (int i = 0; i < 4; ++i) r += (float)pf[i][0];
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
    printf("%-52s waves/SIMD=%d: %7.3f us per wave-tile\n", name, wgs_per_cu, ms * 1e3 / iters / wgs_per_cu);
}
int main() {
    float* d; hipMalloc(&d, 256 * 4 * 256 * sizeof(float));
    for (int w = 1; w <= 3; ++w) {
        run<0>("0 MFMA only (20)", d, w);
        run<1>("1 VALU only (16 max3 + 32 exp + 16 cvt)", d, w);
        run<2>("2 program order QK | softmax | PV", d, w);
        run<3>("3 pinned cross-tile pipeline", d, w);
    }
    return 0;
}
