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
// GEMM: 128 x 128 x 16 tiles, four waves of 64 x 64 (2 x 2 blocks of 32 x 32), global -> registers -> LDS with the next K tile's
// loads in flight during the current tile's MFMAs (two LDS buffers, one barrier per K tile).
//   * `v_mfma_f32_32x32x2_f32` contracts TWO k per instruction: the lower lane half supplies one, the upper half the other.  Any
//     pairing of k values works as long as both operands use the same one, so inside a group of 8 consecutive k the lower half takes
//     k = 0 .. 3 and the upper half k = 4 .. 7: ONE 16-byte LDS read per lane then feeds FOUR MFMAs (the first version read one
//     float per MFMA: 27 TFLOP/s of the 157 peak).
//   * LDS rows of 16 + 4 floats: row r of a tile starts at bank 20 r mod 64 — 16 distinct multiples of 4 over the 16 rows of a
//     `ds_read_b128` lane group, and over the 2 x 4 (row, quad) lanes of a `ds_write_b128` group: no conflicts either way.
//   * the swapped product D[n][m] leaves four consecutive n of one row m in a lane: 16-byte stores where the options allow.
// ------------------------------------------------------------------------------------------------
// Two tile sizes: 128 x 128 (wave tile 64 x 64) for launches with enough tiles to fill the device, 64 x 64 (wave tile 32 x 32) for
// the small levels of the SD1.5 stack at batch 3 (m = 3072 / 768 / 192: 120 / 60 / 20 tiles of 128 x 128 on 256 CUs) — the fp32 matrix
// instruction takes 64 cycles, so the smaller tile's doubled LDS traffic per MFMA costs nothing and four times the workgroups are in flight.
constexpr int FBK = 16, FLD = FBK + 4;

template <int FBM>
__global__ __launch_bounds__(256) void aid_gemm_f32_kernel(const GemmGroup g) {
    constexpr int FBN = FBM, WT = FBM / 2, NB = WT / 32, RS = FBM / 64;       // wave tile, 32-blocks per side, staging rows per thread
    __shared__ __attribute__((aligned(16))) float smem[2 * (FBM + FBN) * FLD];
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
    // column tiles of one row panel are neighbours in the grid: they share the A panel in L2
    const int m0 = (rem / tiles_n) * FBM, n0 = (rem % tiles_n) * FBN;
    const float* __restrict__ A = reinterpret_cast<const float*>(P.a) + (int64_t)batch * P.stride_a;
    const float* __restrict__ B = reinterpret_cast<const float*>(P.b) + (int64_t)batch * P.stride_b;
    float* __restrict__ C = reinterpret_cast<float*>(P.c) + (int64_t)batch * P.stride_c;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = (wave >> 1) * WT, wn = (wave & 1) * WT;
    // staging: thread -> (row = tid / 4 [+ 64], k quad = tid % 4); rows past the matrix are clamped (their products are never stored)
    const int srow = tid >> 2, sq = (tid & 3) * 4;
    const float* ap[RS];
    const float* bp[RS];
#pragma unroll
    for (int i = 0; i < RS; ++i) {
        ap[i] = A + (int64_t)min(m0 + srow + 64 * i, P.m - 1) * P.lda + sq;
        bp[i] = B + (int64_t)min(n0 + srow + 64 * i, P.n - 1) * P.ldb + sq;
    }

    f32x16 acc[NB][NB];                                    // [n block][m block]
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[RS], rb[RS];
    auto load = [&](int k0) {                              // k is a multiple of 8 (aid_hip.h): a quad is inside the row or past it
        const bool in = k0 + sq < P.k;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < RS; ++i) {
            ra[i] = in ? *reinterpret_cast<const f32x4*>(ap[i] + k0) : z;
            rb[i] = in ? *reinterpret_cast<const f32x4*>(bp[i] + k0) : z;
        }
    };
    auto store = [&](int buf) {
        float* As = smem + buf * (FBM + FBN) * FLD;
        float* Bs = As + FBM * FLD;
#pragma unroll
        for (int i = 0; i < RS; ++i) {
            *reinterpret_cast<f32x4*>(As + (srow + 64 * i) * FLD + sq) = ra[i];
            *reinterpret_cast<f32x4*>(Bs + (srow + 64 * i) * FLD + sq) = rb[i];
        }
    };

    const int nk = (P.k + FBK - 1) / FBK;
    load(0);
    store(0);
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
        if (t + 1 < nk) load((t + 1) * FBK);
        const float* As = smem + (t & 1) * (FBM + FBN) * FLD;
        const float* Bs = As + FBM * FLD;
#pragma unroll
        for (int kg = 0; kg < 2; ++kg) {                   // two groups of 8 k per tile
            f32x4 fa[NB], fb[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                fa[i] = *reinterpret_cast<const f32x4*>(As + (wm + 32 * i + l31) * FLD + 8 * kg + 4 * hi);
                fb[i] = *reinterpret_cast<const f32x4*>(Bs + (wn + 32 * i + l31) * FLD + 8 * kg + 4 * hi);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < NB; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j)           // D[n][m]: lane (m = l31, hi) ends up with n = 8 g + 4 hi + e
                        acc[i][j] = mfma32f(fb[i][e], fa[j][e], acc[i][j]);
        }
        if (t + 1 < nk) store((t + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue (every option of AidGemmProblem; fp32 needs no intermediate rounding)
    const float* stats = P.ln_stats ? P.ln_stats + 2 * (int64_t)batch * P.stride_stats : nullptr;
    const float* bias = reinterpret_cast<const float*>(P.bias);
    const float* R = reinterpret_cast<const float*>(P.residual);
    const bool vec = !P.trans_rows && !stats && P.n % 4 == 0 &&     // whole 16-byte groups of a row: one store each
                     (reinterpret_cast<uintptr_t>(bias) & 15) == 0 && (reinterpret_cast<uintptr_t>(R) & 15) == 0;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int m = m0 + wm + 32 * j + l31;
        if (m >= P.m) continue;
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int nb = n0 + wn + 32 * i + 8 * gq + 4 * hi;
                if (vec) {
                    if (nb >= P.n) continue;
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * gq + e] * P.scale;
                    if (bias) v += *reinterpret_cast<const f32x4*>(bias + nb);
                    const int64_t off = (int64_t)m * P.ldc + nb;
                    if (R) v += *reinterpret_cast<const f32x4*>(R + off);
                    *reinterpret_cast<f32x4*>(C + off) = v;
                    continue;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int n = nb + e;
                    if (n >= P.n) {                                // columns [n, round_up(n, 4)) are written with zeros (aid_hip.h)
                        if (!P.trans_rows && n < (P.n + 3) / 4 * 4) C[(int64_t)m * P.ldc + n] = 0.f;
                        continue;
                    }
                    float v = acc[i][j][4 * gq + e];
                    if (stats) {                                   // folded LayerNorm: rstd (x W'^T - mean colsum) + shift
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
    }
}

static int f32_tiles(GemmGroup& g, int bm) {
    int tiles = 0;
    for (int i = 0; i < g.n_problems; ++i) {
        g.tile_start[i] = tiles;
        tiles += ((g.p[i].m + bm - 1) / bm) * ((g.p[i].n + bm - 1) / bm) * g.p[i].batch;
    }
    for (int i = g.n_problems; i <= AID_GEMM_MAX_PROBLEMS; ++i) g.tile_start[i] = tiles;
    return tiles;
}

hipError_t gemm_f32_launch(GemmGroup& g, hipStream_t stream) {
    // big tiles only when there are enough of them to give every CU two (512 on MI355X); same arithmetic either way (the K order
    // inside a tile does not depend on the tile size: results are bit-identical)
    static PerDevice<int> ncu;
    int* n = ncu.slot();
    if (!n) return hipErrorInvalidDevice;
    if (*n == 0) {
        int dev = 0;
        hipDeviceProp_t pr;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return hipErrorInvalidDevice;
        *n = pr.multiProcessorCount;
    }
    int tiles = f32_tiles(g, 128);
    if (tiles <= 0) return hipSuccess;
    if (tiles >= 2 * *n) {
        hipLaunchKernelGGL(aid_gemm_f32_kernel<128>, dim3(tiles), dim3(256), 0, stream, g);
    } else {
        tiles = f32_tiles(g, 64);
        hipLaunchKernelGGL(aid_gemm_f32_kernel<64>, dim3(tiles), dim3(256), 0, stream, g);
    }
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

// TWO = the launch has frames that take two softmax passes (OUTER): only then is a second accumulator set (the weighted sum of the
// passes) needed beside the running one — PLAIN / INNER launches normalise their single pass in place (d = 160: 80 registers less).
template <int D, bool TWO>
__global__ __launch_bounds__(256, (D == 40 ? (TWO ? 2 : 3) : D == 64 ? 2 : D == 80 ? (TWO ? 1 : 2) : 1)) void aid_attn_f32_kernel(const AttnF32Params p) {
    constexpr int DP = (D + 31) / 32 * 32, NDB = DP / 32;          // channels padded to whole 32-blocks of the PV product
    // LDS rows of D + 4 / 32 + 4 floats: (row stride / 4) is odd, so the 16 rows of a `ds_read_b128` lane group start at 16 distinct
    // multiples of four banks — every fragment read below is one conflict-free 16-byte read that feeds FOUR MFMAs (see the GEMM).
    constexpr int KLD = D + 4;                                     // K tile rows [32][D + 4]
    constexpr int VLD = FKT + 4;                                   // V^T tile rows [DP][36]
    static_assert((KLD / 4) % 2 == 1 && (VLD / 4) % 2 == 1 && D % 8 == 0, "conflict-free 16-byte fragment reads");
    // PF (d <= 80): the NEXT tile's global loads are in flight during the current tile's MFMAs (registers -> the other LDS buffer after
    // the arithmetic, ONE barrier per tile); d = 160 has no registers for it and stages synchronously through one buffer
    constexpr bool PF = D <= 80;
    constexpr int NBUF = PF ? 2 : 1;
    constexpr int KCH = FKT * (D / 4), VCH = DP * (FKT / 4);       // 16-byte chunks of a K / V^T tile
    constexpr int NKC = (KCH + 255) / 256, NVC = (VCH + 255) / 256;
    __shared__ __attribute__((aligned(16))) float Ks_[NBUF * FKT * KLD];
    __shared__ __attribute__((aligned(16))) float Vs_[NBUF * DP * VLD];
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
    // Q[q][8 g + 4 hi + e] at qf[4 g + e], pre-multiplied by softmax_scale log2(e): inside a group of 8 channels the lower lane half
    // contracts channels 0 .. 3 and the upper half 4 .. 7 (the K fragment reads below use the same pairing)
    float qf[D / 2];
#pragma unroll
    for (int gq = 0; gq < D / 8; ++gq) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(Q + 8 * gq + 4 * hi);
#pragma unroll
        for (int e = 0; e < 4; ++e) qf[4 * gq + e] = v[e] * p.c2;
    }

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
        f32x16 otmp[NDB];
        f32x16 (&o)[NDB] = TWO ? otmp : res;                        // one pass per frame: accumulate where the result lives
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
        float mrow = -INFINITY, lsum = 0.f;                         // running reference and this lane half's partial row sum
        const int ntl = (a.l + FKT - 1) / FKT;                      // tiles per segment
        const int nt = (k2 ? 2 : 1) * ntl;                          // (k1 is never null)
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        f32x4 rk[NKC], rv[NVC];
        // tile i of the pass -> registers (chunks past the tile / the keys / the channels: zero)
        auto gload = [&](int i) {
            const float* kp = i >= ntl ? k2 : k1;
            const float* vp = i >= ntl ? v2 : v1;
            const int t0 = (i >= ntl ? i - ntl : i) * FKT;
#pragma unroll
            for (int u = 0; u < NKC; ++u) {                         // K tile [key][channel], 16 bytes at a time
                const int id = tid + 256 * u;
                const int kk = id / (D / 4), c = (id - kk * (D / 4)) * 4;
                rk[u] = (id < KCH && t0 + kk < a.l) ? *reinterpret_cast<const f32x4*>(kp + (int64_t)(t0 + kk) * a.ldk + h * D + c) : z4;
            }
#pragma unroll
            for (int u = 0; u < NVC; ++u) {                         // V^T tile [channel][key]; keys past L and pad channels are zero
                const int id = tid + 256 * u;
                const int c = id / (FKT / 4), kk = (id - c * (FKT / 4)) * 4;
                f32x4 v = z4;
                if (id < VCH && c < D && t0 + kk < a.l) {
                    const float* src = vp + (int64_t)(h * D + c) * a.ldvt + t0 + kk;
                    if (t0 + kk + 3 < a.l) v = *reinterpret_cast<const f32x4*>(src);       // (ldvt, t0 and kk are multiples of 4)
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = (t0 + kk + e < a.l) ? src[e] : 0.f;
                    }
                }
                rv[u] = v;
            }
        };
        auto lstore = [&](int buf) {
            float* Kb = Ks_ + buf * FKT * KLD;
            float* Vb = Vs_ + buf * DP * VLD;
#pragma unroll
            for (int u = 0; u < NKC; ++u) {
                const int id = tid + 256 * u;
                const int kk = id / (D / 4), c = (id - kk * (D / 4)) * 4;
                if (KCH % 256 == 0 || id < KCH) *reinterpret_cast<f32x4*>(Kb + kk * KLD + c) = rk[u];
            }
#pragma unroll
            for (int u = 0; u < NVC; ++u) {
                const int id = tid + 256 * u;
                const int c = id / (FKT / 4), kk = (id - c * (FKT / 4)) * 4;
                if (VCH % 256 == 0 || id < VCH) *reinterpret_cast<f32x4*>(Vb + c * VLD + kk) = rv[u];
            }
        };
        __syncthreads();                                            // the previous pass has been consumed by every wave
        if (PF) {
            gload(0);
            lstore(0);
            __syncthreads();
        }
        for (int it = 0; it < nt; ++it) {
            {
                const int t0 = (it >= ntl ? it - ntl : it) * FKT;
                const float* Ks = Ks_ + (PF ? (it & 1) : 0) * FKT * KLD;
                const float* Vs = Vs_ + (PF ? (it & 1) : 0) * DP * VLD;
                if (PF) {
                    if (it + 1 < nt) gload(it + 1);                 // in flight across this tile's arithmetic
                } else {
                    if (it) __syncthreads();                        // the previous tile has been consumed by every wave
                    gload(it);
                    lstore(0);
                    __syncthreads();
                }
                // S^T = K Q'^T: lane (query l31, half hi) receives the scores of keys key_of(r, hi)
                f32x16 sc;
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
                for (int gq = 0; gq < D / 8; ++gq) {
                    const f32x4 kf = *reinterpret_cast<const f32x4*>(Ks + l31 * KLD + 8 * gq + 4 * hi);
#pragma unroll
                    for (int e = 0; e < 4; ++e) sc = mfma32f(kf[e], qf[4 * gq + e], sc);
                }
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
                // O^T = alpha O^T + V^T P^T: k-step r contracts keys key_of(r, 0) (lower lane half) and key_of(r, 1) (upper);
                // registers 4 j .. 4 j + 3 of a lane are the four CONSECUTIVE keys 8 j + 4 hi + {0 .. 3}: one 16-byte read of the V^T row
                if (__any(alpha != 1.f)) {                          // (most tiles leave every row's reference where it was)
#pragma unroll
                    for (int d = 0; d < NDB; ++d)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int d = 0; d < NDB; ++d) {                // (block-inner: consecutive MFMAs go to different accumulators)
                        const f32x4 vf = *reinterpret_cast<const f32x4*>(Vs + (32 * d + l31) * VLD + 8 * j + 4 * hi);
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[d] = mfma32f(vf[e], sc[4 * j + e], o[d]);
                    }
                if (PF) {
                    if (it + 1 < nt) lstore((it + 1) & 1);          // that buffer was last read a barrier ago
                    __syncthreads();
                }
            }
        }
        const float inv = w / sum_halves(lsum);
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) res[d][r] = TWO ? fmaf(o[d][r], inv, res[d][r]) : o[d][r] * inv;
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
    if (p.a.mode == AID_MODE_OUTER)
        hipLaunchKernelGGL((aid_attn_f32_kernel<D, true>), dim3(nqb * p.a.heads * p.a.n_frames), dim3(256), 0, stream, p);
    else
        hipLaunchKernelGGL((aid_attn_f32_kernel<D, false>), dim3(nqb * p.a.heads * p.a.n_frames), dim3(256), 0, stream, p);
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
