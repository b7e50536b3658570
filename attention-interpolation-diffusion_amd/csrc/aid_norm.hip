// LayerNorm over the channel dimension of [rows, c] activations — the `norm1` / `norm2` that sits in front of the
// attention call inside diffusers' BasicTransformerBlock (the step either side of the path).
// HBM-bound streaming: one wave per row, 16 B per lane per access, the row stays in registers between the
// statistics and the normalisation (read once, written once), fp32 statistics with the two-pass variance
// (sum of squared deviations), one rounding to the storage type — what torch's LayerNorm kernel produces for
// fp16 / bf16 inputs up to the order of the fp32 reduction.
#include "aid_common.hpp"
#include "aid_kernels.hpp"

namespace aid {

constexpr int LN_MAX_CHUNKS = 4;            // 64 lanes x 8 elements x 4 = 2048 channels

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) v += __shfl_xor(v, o, 64);
    return sum_halves(v);
}

template <typename T>
__global__ __launch_bounds__(256) void aid_layernorm_kernel(const T* __restrict__ x, const T* __restrict__ gamma,
                                                            const T* __restrict__ beta, T* __restrict__ y, int64_t rows,
                                                            int c, float eps) {
    typedef typename Vec<T>::v8 T8;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nch = c >> 3;                                     // 16-B chunks per row
    const T* xr = x + row * c;
    f32x8 v[LN_MAX_CHUNKS];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_CHUNKS; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nch) {
            v[i] = up8<T>(*reinterpret_cast<const T8*>(xr + ch * 8));
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[i][e];
        }
    }
    const float mean = wave_sum(s) / (float)c;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_CHUNKS; ++i) {
        if (lane + 64 * i < nch) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = v[i][e] - mean;
                q = fmaf(d, d, q);
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)c + eps);
    T* yr = y + row * c;
#pragma unroll
    for (int i = 0; i < LN_MAX_CHUNKS; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nch) {
            f32x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mean) * rstd;
            if (gamma) {
                const f32x8 g = up8<T>(*reinterpret_cast<const T8*>(gamma + ch * 8));
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] *= g[e];
            }
            if (beta) {
                const f32x8 b = up8<T>(*reinterpret_cast<const T8*>(beta + ch * 8));
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] += b[e];
            }
            *reinterpret_cast<T8*>(yr + ch * 8) = cvt8<T>(o);
        }
    }
}

// ---- LayerNorm folded into the projections that consume it -----------------------------------------------------------
//   LayerNorm(x) W^T = rstd (x W'^T - mean * colsum) + shift,   W' = W * gamma (rounded to the storage type),
//   colsum[n] = sum_k W'[n, k]  (of the ROUNDED W', so the identity holds for the numbers the GEMM multiplies),
//   shift[n]  = sum_k beta[k] W[n, k].
// aid_ln_stats_kernel reads the activations once and writes (mean, rstd) per row — the normalised tensor is never
// materialised (the separate LayerNorm kernel writes and the projection re-reads [rows, c]); aid_ln_fold_kernel prepares
// the weight side once per (weights, gamma, beta).  Same statistics arithmetic as aid_layernorm_kernel above.
// LN_RPW rows per wave, every row's loads issued before the first reduction: a wave's life is one memory latency, and with one row per
// wave (round 2 - 5) the 14336-row launches of the SDXL stack were two rounds of 32 waves per CU, each a latency long (12 us for 36.7 MB).
constexpr int LN_RPW = 4;
template <typename T>
__global__ __launch_bounds__(256) void aid_ln_stats_kernel(const T* __restrict__ x, float* __restrict__ stats, int64_t rows,
                                                           int c, float eps) {
    typedef typename Vec<T>::v8 T8;
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * LN_RPW;
    if (row0 >= rows) return;
    const int nch = c >> 3;
    T8 raw[LN_RPW][LN_MAX_CHUNKS];
#pragma unroll
    for (int r = 0; r < LN_RPW; ++r) {
        const T* xr = x + (row0 + r < rows ? row0 + r : rows - 1) * c;       // (rows past the end: a valid row, its result is dropped)
#pragma unroll
        for (int i = 0; i < LN_MAX_CHUNKS; ++i) {
            const int ch = lane + 64 * i;
            raw[r][i] = ch < nch ? *reinterpret_cast<const T8*>(xr + ch * 8) : zero8<T>();
        }
    }
#pragma unroll
    for (int r = 0; r < LN_RPW; ++r) {
        // the arithmetic of one row, in the order of aid_layernorm_kernel (same sums, same bits as the one-row-per-wave version)
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAX_CHUNKS; ++i) {
            if (lane + 64 * i < nch) {
                const f32x8 v = up8<T>(raw[r][i]);
#pragma unroll
                for (int e = 0; e < 8; ++e) s += v[e];
            }
        }
        const float mean = wave_sum(s) / (float)c;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAX_CHUNKS; ++i) {
            if (lane + 64 * i < nch) {
                const f32x8 v = up8<T>(raw[r][i]);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = v[e] - mean;
                    q = fmaf(d, d, q);
                }
            }
        }
        const float rstd = rsqrtf(wave_sum(q) / (float)c + eps);
        if (lane == 0 && row0 + r < rows) {
            stats[2 * (row0 + r)] = mean;
            stats[2 * (row0 + r) + 1] = rstd;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void aid_ln_fold_kernel(const T* __restrict__ w, const T* __restrict__ gamma,
                                                          const T* __restrict__ beta, T* __restrict__ wf,
                                                          float* __restrict__ colsum, float* __restrict__ shift, int rows, int c) {
    typedef typename Vec<T>::v8 T8;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float cs = 0.f, sh = 0.f;
    for (int ch = lane; ch < (c >> 3); ch += 64) {
        const f32x8 wv = up8<T>(*reinterpret_cast<const T8*>(w + (int64_t)row * c + ch * 8));
        f32x8 g, b;
#pragma unroll
        for (int e = 0; e < 8; ++e) { g[e] = 1.f; b[e] = 0.f; }
        if (gamma) g = up8<T>(*reinterpret_cast<const T8*>(gamma + ch * 8));
        if (beta) b = up8<T>(*reinterpret_cast<const T8*>(beta + ch * 8));
        f32x8 p;
#pragma unroll
        for (int e = 0; e < 8; ++e) p[e] = wv[e] * g[e];
        const T8 pr = cvt8<T>(p);
        *reinterpret_cast<T8*>(wf + (int64_t)row * c + ch * 8) = pr;
        const f32x8 pf = up8<T>(pr);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            cs += pf[e];
            sh = fmaf(b[e], wv[e], sh);
        }
    }
    cs = wave_sum(cs);
    sh = wave_sum(sh);
    if (lane == 0) {
        colsum[row] = cs;
        shift[row] = sh;
    }
}

hipError_t ln_stats_launch(const void* x, float* stats, int64_t rows, int c, float eps, int dtype, hipStream_t stream) {
    if (rows <= 0) return hipSuccess;
    const dim3 grid((unsigned)((rows + 4 * LN_RPW - 1) / (4 * LN_RPW)));
    if (dtype == AID_DTYPE_F16)
        hipLaunchKernelGGL(aid_ln_stats_kernel<f16>, grid, dim3(256), 0, stream, (const f16*)x, stats, rows, c, eps);
    else
        hipLaunchKernelGGL(aid_ln_stats_kernel<bf16>, grid, dim3(256), 0, stream, (const bf16*)x, stats, rows, c, eps);
    return hipGetLastError();
}

hipError_t ln_fold_launch(const void* w, const void* gamma, const void* beta, void* w_folded, float* colsum, float* shift,
                          int rows, int c, int dtype, hipStream_t stream) {
    if (rows <= 0) return hipSuccess;
    const dim3 grid((unsigned)((rows + 3) / 4));
    if (dtype == AID_DTYPE_F16)
        hipLaunchKernelGGL(aid_ln_fold_kernel<f16>, grid, dim3(256), 0, stream, (const f16*)w, (const f16*)gamma,
                           (const f16*)beta, (f16*)w_folded, colsum, shift, rows, c);
    else
        hipLaunchKernelGGL(aid_ln_fold_kernel<bf16>, grid, dim3(256), 0, stream, (const bf16*)w, (const bf16*)gamma,
                           (const bf16*)beta, (bf16*)w_folded, colsum, shift, rows, c);
    return hipGetLastError();
}

bool layernorm_width_supported(int c) { return c >= 8 && c % 8 == 0 && c <= 64 * 8 * LN_MAX_CHUNKS; }

hipError_t layernorm_launch(const void* x, const void* gamma, const void* beta, void* y, int64_t rows, int c, float eps,
                            int dtype, hipStream_t stream) {
    if (rows <= 0) return hipSuccess;
    const dim3 grid((unsigned)((rows + 3) / 4));
    if (dtype == AID_DTYPE_F16)
        hipLaunchKernelGGL(aid_layernorm_kernel<f16>, grid, dim3(256), 0, stream, (const f16*)x, (const f16*)gamma,
                           (const f16*)beta, (f16*)y, rows, c, eps);
    else
        hipLaunchKernelGGL(aid_layernorm_kernel<bf16>, grid, dim3(256), 0, stream, (const bf16*)x, (const bf16*)gamma,
                           (const bf16*)beta, (bf16*)y, rows, c, eps);
    return hipGetLastError();
}

}  // namespace aid
