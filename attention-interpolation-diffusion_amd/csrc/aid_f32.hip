// float32 storage path (AID_DTYPE_F32, ABI v7): the reference's own default for SD1.x — `gradio_src/app.py:62, 78-85, 414` load the
// SD pipelines in float32 ("SDXL will run on float16 while the rest will run on float32") and its CPU path is float32 throughout.
// Same entry points, same semantics as the f16 / bf16 kernels (aid_gemm_nt / aid_attn_fwd / aid_lerp_kv / aid_processor_fwd), float32
// tensors in and out, float32 arithmetic on the matrix pipe: `v_mfma_f32_32x32x2_f32` is exact f32 (an fmaf chain) at the f32 vector
// rate (157 TFLOP/s peak, 1 / 16 of the bf16 rate — /opt/skills/guides/MI355X_MICROARCH.md), so these kernels are correctness-first:
// LDS-tiled, synchronous staging, every edge guarded.  What they buy is the strongest parity statement the repository can make: the
// HIP path against the reference's OWN float32 outputs (tests/golden/*.npz) at rounding-noise level, not through a storage-type
// tolerance.  Probabilities are NOT rounded before the PV product (the reference's float32 `get_attention_scores` does not either).
//
//   aid_gemm_f32_kernel     C = scale * A B^T (+ bias) (+ residual), 64 x 64 x 16 tiles, four waves of one 32 x 32 block; every
//                           AidGemmProblem option (batches, transposed-per-frame output, folded LayerNorm constants)
//                           [attn.to_q / to_k / to_v / to_out[0], interpolation.py:613, 623-624, 666]
//   aid_attn_f32_kernel<D>  flash-style interpolated attention, 128 query rows per workgroup (32 per wave), 32-key tiles through LDS,
//                           swapped products (a lane owns one query row), online softmax in fp32; PLAIN / INNER / OUTER, fused or
//                           pure, riders, maps, accumulate — OUTER as two independent softmax passes like the reference writes it
//                           [interpolation.py:626-664, 760-790]
//   aid_lerp_kv_f32_kernel  K / V^T of the interior frames for INNER [interpolation.py:772-775]
#include "aid_common.hpp"
#include "aid_kernels.hpp"

#include <string.h>

namespace aid {

// D(32x32) += A(32x2) B(2x32): lane l supplies A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31]; the result layout is mfma32's
__device__ __forceinline__ f32x16 mfma32f(float a, float b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// GEMM
// ------------------------------------------------------------------------------------------------
constexpr int FBM = 64, FBN = 64, FBK = 16, FLD = FBK + 1;       // padded LDS rows: (17 r + k) mod 32 is conflict-free over r

__global__ __launch_bounds__(256) void aid_gemm_f32_kernel(const GemmGroup g) {
    __shared__ float As[FBM * FLD];
    __shared__ float Bs[FBN * FLD];
    // ---- block -> (problem, batch, tile)
    int p = 0;
#pragma unroll
    for (int i = 1; i < AID_GEMM_MAX_PROBLEMS; ++i)
        if (i < g.n_problems && (int)blockIdx.x >= g.tile_start[i]) p = i;
    const GemmDesc& P = g.p[p];
    int rem = blockIdx.x - g.tile_start[p];
    const int tiles_n = (P.n + FBN - 1) / FBN, tiles_m = (P.m + FBM - 1) / FBM;
    const int batch = rem / (tiles_m * tiles_n);
    rem -= batch * tiles_m * tiles_n;
    const int m0 = (rem / tiles_n) * FBM, n0 = (rem % tiles_n) * FBN;
    const float* __restrict__ A = reinterpret_cast<const float*>(P.a) + (int64_t)batch * P.stride_a;
    const float* __restrict__ B = reinterpret_cast<const float*>(P.b) + (int64_t)batch * P.stride_b;
    float* __restrict__ C = reinterpret_cast<float*>(P.c) + (int64_t)batch * P.stride_c;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const int srow = tid >> 2, sk = (tid & 3) * 4;                 // staging: 64 rows x 4 quads of k

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    for (int k0 = 0; k0 < P.k; k0 += FBK) {
        float ra[4], rb[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int kk = k0 + sk + e;
            ra[e] = (m0 + srow < P.m && kk < P.k) ? A[(int64_t)(m0 + srow) * P.lda + kk] : 0.f;
            rb[e] = (n0 + srow < P.n && kk < P.k) ? B[(int64_t)(n0 + srow) * P.ldb + kk] : 0.f;
        }
        __syncthreads();                                           // the previous tile has been consumed
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            As[srow * FLD + sk + e] = ra[e];
            Bs[srow * FLD + sk + e] = rb[e];
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < FBK; kk += 2)                        // D[n][m]: lane (m = l31, hi) ends up with n = 8 g + 4 hi + e
            acc = mfma32f(Bs[(wn + l31) * FLD + kk + hi], As[(wm + l31) * FLD + kk + hi], acc);
    }

    // ---- epilogue (every option of AidGemmProblem; fp32 needs no intermediate rounding)
    const int m = m0 + wm + l31;
    if (m >= P.m) return;
    const float* stats = P.ln_stats ? P.ln_stats + 2 * (int64_t)batch * P.stride_stats : nullptr;
    const float* bias = reinterpret_cast<const float*>(P.bias);
    const float* R = reinterpret_cast<const float*>(P.residual);
#pragma unroll
    for (int gq = 0; gq < 4; ++gq)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n = n0 + wn + 8 * gq + 4 * hi + e;
            if (n >= P.n) {                                        // columns [n, round_up(n, 4)) are written with zeros (aid_hip.h)
                if (!P.trans_rows && n < (P.n + 3) / 4 * 4) C[(int64_t)m * P.ldc + n] = 0.f;
                continue;
            }
            float v = acc[4 * gq + e];
            if (stats) {                                           // folded LayerNorm: rstd (x W'^T - mean colsum) + shift
                if (P.ln_side == 1) v = fmaf(stats[2 * m + 1], fmaf(-stats[2 * m], P.ln_colsum[n], v), P.ln_shift[n]);
                else                v = fmaf(stats[2 * n + 1], fmaf(-stats[2 * n], P.ln_colsum[m], v), P.ln_shift[m]);
            }
            v *= P.scale;
            if (bias) v += bias[n];
            int64_t off;
            if (P.trans_rows) off = (int64_t)(m / P.trans_rows) * P.stride_c + (int64_t)n * P.ldc + m % P.trans_rows;
            else              off = (int64_t)m * P.ldc + n;
            if (R) v += R[off];
            C[off] = v;
        }
}

hipError_t gemm_f32_launch(GemmGroup& g, hipStream_t stream) {
    int tiles = 0;
    for (int i = 0; i < g.n_problems; ++i) {
        g.tile_start[i] = tiles;
        tiles += ((g.p[i].m + FBM - 1) / FBM) * ((g.p[i].n + FBN - 1) / FBN) * g.p[i].batch;
    }
    for (int i = g.n_problems; i <= AID_GEMM_MAX_PROBLEMS; ++i) g.tile_start[i] = tiles;
    if (tiles <= 0) return hipSuccess;
    hipLaunchKernelGGL(aid_gemm_f32_kernel, dim3(tiles), dim3(256), 0, stream, g);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// attention
// ------------------------------------------------------------------------------------------------
constexpr int FKT = 32;                    // keys per tile

struct AttnF32Params {
    AidAttnArgs a;
    float c2;                              // softmax_scale * log2(e)  (1 when q is pre-scaled)
};

// key index of accumulator register r in lane half hi (the mfma32 result layout): the r-th key a lane holds
__device__ __forceinline__ int key_of(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

template <int D>
__global__ __launch_bounds__(256) void aid_attn_f32_kernel(const AttnF32Params p) {
    constexpr int DP = (D + 31) / 32 * 32, NDB = DP / 32;          // channels padded to whole 32-blocks of the PV product
    constexpr int KLD = D + 1;                                     // K tile rows [32][D + 1]
    constexpr int VLD = FKT + 1;                                   // V^T tile rows [DP][33]
    __shared__ float Ks[FKT * KLD];
    __shared__ float Vs[DP * VLD];
    const AidAttnArgs& a = p.a;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int nqb = (a.s + 127) / 128;
    int bid = blockIdx.x;
    const int qb = bid % nqb; bid /= nqb;
    const int h = bid % a.heads;
    const int fr = bid / a.heads;
    const int q = qb * 128 + wave * 32 + l31;                       // this lane's query row (both lane halves share it)
    const bool qok = q < a.s;

    const float* __restrict__ Q = reinterpret_cast<const float*>(a.q) + (int64_t)fr * a.q_fs + (int64_t)min(q, a.s - 1) * a.ldq + h * D;
    float qf[D / 2];                                               // Q[q][2 t + hi], pre-multiplied by softmax_scale log2(e)
#pragma unroll
    for (int t = 0; t < D / 2; ++t) qf[t] = Q[2 * t + hi] * p.c2;

    // ---- what this frame attends with (decided per frame from the coefficient, like aid_attn_kernel)
    const float cf = (a.mode != AID_MODE_PLAIN && a.coef) ? a.coef[fr] : -1.f;
    const bool plain = a.mode == AID_MODE_PLAIN || cf < 0.f;       // a negative coefficient marks a PLAIN rider
    const int own = a.kv_map ? a.kv_map[fr] : fr;
    const float* const K0 = reinterpret_cast<const float*>(a.k);
    const float* const V0 = reinterpret_cast<const float*>(a.vt);

    // additive score bias (AidAttnArgs.bias, ABI v8): the row of this lane's query; element j goes with key j of every segment
    const float* const brow = a.bias ? reinterpret_cast<const float*>(a.bias) + (int64_t)fr * a.bias_fs + (int64_t)h * a.bias_hs +
                                           (int64_t)min(q, a.s - 1) * a.bias_rs
                                     : nullptr;

    f32x16 res[NDB];                                               // sum over passes of weight * O^T
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) res[d][r] = 0.f;

    // one softmax pass over up to two key segments; its normalised output is added to `res` with weight w
    auto pass = [&](const float* k1, const float* v1, const float* k2, const float* v2, float w) {
        f32x16 o[NDB];
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
        float mrow = -INFINITY, lsum = 0.f;                         // running reference and this lane half's partial row sum
        for (int seg = 0; seg < 2; ++seg) {
            const float* kp = seg ? k2 : k1;
            const float* vp = seg ? v2 : v1;
            if (!kp) continue;
            for (int t0 = 0; t0 < a.l; t0 += FKT) {
                __syncthreads();                                   // the previous tile has been consumed by every wave
                for (int i = tid; i < FKT * D; i += 256) {         // K tile [key][channel]
                    const int kk = i / D, c = i - kk * D;
                    Ks[kk * KLD + c] = (t0 + kk < a.l) ? kp[(int64_t)(t0 + kk) * a.ldk + h * D + c] : 0.f;
                }
                for (int i = tid; i < DP * FKT; i += 256) {        // V^T tile [channel][key]
                    const int c = i / FKT, kk = i - c * FKT;
                    Vs[c * VLD + kk] = (c < D && t0 + kk < a.l) ? vp[(int64_t)(h * D + c) * a.ldvt + t0 + kk] : 0.f;
                }
                __syncthreads();
                // S^T = K Q'^T: lane (query l31, half hi) receives the scores of keys key_of(r, hi)
                f32x16 sc;
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
                for (int t = 0; t < D / 2; ++t) sc = mfma32f(Ks[l31 * KLD + 2 * t + hi], qf[t], sc);
                if (brow) {                                        // scale q k^T + bias, in the log2 domain (values below -1e30 clamped, aid_attn.hip)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        sc[r] += fmaxf(brow[min(t0 + key_of(r, hi), a.l - 1)], -1e30f) * 1.4426950408889634f;
                }
                float mx = -INFINITY;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (t0 + key_of(r, hi) >= a.l) sc[r] = -INFINITY;
                    mx = fmaxf(mx, sc[r]);
                }
                mx = max_halves(mx);
                const float mnew = fmaxf(mrow, mx);                // finite: every tile holds at least one key
                const float alpha = exp2f(mrow - mnew);            // 0 on the first tile (mrow = -inf)
                mrow = mnew;
                float ps = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    sc[r] = exp2f(sc[r] - mnew);
                    ps += sc[r];
                }
                lsum = lsum * alpha + ps;
                // O^T = alpha O^T + V^T P^T: k-step r contracts keys key_of(r, 0) (lower lane half) and key_of(r, 1) (upper)
#pragma unroll
                for (int d = 0; d < NDB; ++d) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[d] = mfma32f(Vs[(32 * d + l31) * VLD + key_of(r, hi)], sc[r], o[d]);
                }
            }
        }
        const float inv = w / sum_halves(lsum);
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) res[d][r] = fmaf(o[d][r], inv, res[d][r]);
    };

    const float* k_own = K0 + (int64_t)own * a.k_fs;
    const float* v_own = V0 + (int64_t)own * a.vt_fs;
    if (plain) {
        pass(k_own, v_own, nullptr, nullptr, 1.f);
    } else if (a.mode == AID_MODE_INNER) {
        // interpolated keys / values: k2 / vt2 row of the frame for 0 < c < 1, the end-point frames themselves for c = 0 / 1
        const float* km;
        const float* vm;
        if (cf > 0.f && cf < 1.f) {
            km = reinterpret_cast<const float*>(a.k2) + (int64_t)fr * a.k_fs;
            vm = reinterpret_cast<const float*>(a.vt2) + (int64_t)fr * a.vt_fs;
        } else {
            const int e = cf == 0.f ? a.begin : a.end;
            km = K0 + (int64_t)e * a.k_fs;
            vm = V0 + (int64_t)e * a.vt_fs;
        }
        if (a.fused) pass(k_own, v_own, km, vm, 1.f);
        else         pass(km, vm, nullptr, nullptr, 1.f);
    } else {                                                       // OUTER: (1 - c) A(.., begin) + c A(.., end); a zero weight drops its side
        const float* kb = K0 + (int64_t)a.begin * a.k_fs;
        const float* vb = V0 + (int64_t)a.begin * a.vt_fs;
        const float* ke = K0 + (int64_t)a.end * a.k_fs;
        const float* ve = V0 + (int64_t)a.end * a.vt_fs;
        if (cf != 1.f) { if (a.fused) pass(k_own, v_own, kb, vb, 1.f - cf); else pass(kb, vb, nullptr, nullptr, 1.f - cf); }
        if (cf != 0.f) { if (a.fused) pass(k_own, v_own, ke, ve, cf);       else pass(ke, ve, nullptr, nullptr, cf); }
    }

    // ---- out_i = (accumulate ? out_i : 0) + out_scale * frame_scale[i] * O_i; lane (q, hi) holds channels 32 d + 8 g + 4 hi + e
    if (!qok) return;
    const float osc = a.out_scale * (a.frame_scale ? a.frame_scale[fr] : 1.f);
    float* orow = reinterpret_cast<float*>(a.out) + (int64_t)fr * a.o_fs + (int64_t)q * a.ldo + h * D;
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int dv = 32 * d + 8 * gq + 4 * hi + e;
                if (dv < D) {
                    float v = res[d][4 * gq + e] * osc;
                    if (a.accumulate) v += orow[dv];
                    orow[dv] = v;
                }
            }
}

template <int D>
static hipError_t attn_f32_run(const AttnF32Params& p, hipStream_t stream) {
    const int nqb = (p.a.s + 127) / 128;
    hipLaunchKernelGGL(aid_attn_f32_kernel<D>, dim3(nqb * p.a.heads * p.a.n_frames), dim3(256), 0, stream, p);
    return hipGetLastError();
}

hipError_t attn_f32_launch(const AidAttnArgs& a, hipStream_t stream) {
    AttnF32Params p;
    p.a = a;
    p.c2 = a.q_prescaled ? 1.f : a.softmax_scale * 1.4426950408889634f;
    switch (a.d) {
        case 40:  return attn_f32_run<40>(p, stream);
        case 64:  return attn_f32_run<64>(p, stream);
        case 80:  return attn_f32_run<80>(p, stream);
        case 160: return attn_f32_run<160>(p, stream);
        default:  return hipErrorInvalidValue;
    }
}

// ------------------------------------------------------------------------------------------------
// K / V^T of the interior frames (INNER)
// ------------------------------------------------------------------------------------------------
__global__ void aid_lerp_kv_f32_kernel(const float* __restrict__ k, const float* __restrict__ vt, float* __restrict__ k2,
                                       float* __restrict__ vt2, const float* __restrict__ coef, int begin, int end, int64_t k_fs,
                                       int64_t vt_fs) {
    const int fr = blockIdx.y;
    const float c = coef[fr];
    if (!(c > 0.f && c < 1.f)) return;                             // end points and PLAIN riders read the projected tensors themselves
    const int64_t n = k_fs + vt_fs;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const bool isk = i < k_fs;
        const int64_t j = isk ? i : i - k_fs;
        const float* src = isk ? k : vt;
        const int64_t fs = isk ? k_fs : vt_fs;
        const float b = src[(int64_t)begin * fs + j], e = src[(int64_t)end * fs + j];
        (isk ? k2 : vt2)[(int64_t)fr * fs + j] = (1.f - c) * b + c * e;       // interpolation.py:772-775
    }
}

hipError_t lerp_kv_f32_launch(const void* k, const void* vt, void* k2, void* vt2, const float* coef, int n_frames, int begin, int end,
                              int64_t k_fs, int64_t vt_fs, hipStream_t stream) {
    const int64_t n = k_fs + vt_fs;
    const int bx = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(aid_lerp_kv_f32_kernel, dim3(bx > 0 ? bx : 1, n_frames), dim3(256), 0, stream,
                       reinterpret_cast<const float*>(k), reinterpret_cast<const float*>(vt), reinterpret_cast<float*>(k2),
                       reinterpret_cast<float*>(vt2), coef, begin, end, k_fs, vt_fs);
    return hipGetLastError();
}

}  // namespace aid
