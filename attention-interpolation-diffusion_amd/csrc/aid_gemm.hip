// Grouped NT GEMM for the q / k / V^T / out projections of an AID attention layer.
//   C[b] = A[b] * B[b]^T (+ bias[n]),   A [m,k], B [n,k] (both K-contiguous), fp32 accumulate.
// Replaces attn.to_q / to_k / to_v / to_out[0] (reference interpolation.py:613, 623-624, 666).
//
// gfx950 mapping: 128x128x64 block tile, 256 threads = 4 waves in a 2x2 grid, each wave owns a
// 64x64 output tile as 2x2 MFMA 32x32x16 blocks (64 fp32 accumulators / lane).  Operands are
// staged global -> registers -> LDS (issue-early / write-late, one barrier per K tile, two LDS
// buffers); LDS rows are padded by 16 B so the ds_read_b128 fragment reads are conflict free
// (row stride = 9 x 16 B, odd).  The MFMA is issued "transposed" (D rows = n, cols = m) so every
// lane ends up with 4 consecutive n of one output row -> 8-byte stores.  The 1-D grid is remapped
// XCD-aware so the blocks that share an A row panel run on one XCD / one L2.
#include "aid_common.hpp"
#include "aid_kernels.hpp"

namespace aid {

constexpr int GBM = 128, GBN = 128, GBK = 64;
constexpr int GLD = GBK + 8;            // padded LDS row (elements)
constexpr int GTHREADS = 256;

template <typename T>
__global__ __launch_bounds__(GTHREADS) void aid_gemm_nt_kernel(const GemmGroup g) {
    typedef typename Vec<T>::v8 T8;
    typedef typename Vec<T>::v4 T4;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* As = reinterpret_cast<T*>(smem_raw);                 // [2][GBM][GLD]
    T* Bs = As + 2 * GBM * GLD;                             // [2][GBN][GLD]

    // ---- which problem / batch / tile ------------------------------------------------
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    int p = 0;
#pragma unroll
    for (int i = 1; i < AID_GEMM_MAX_PROBLEMS; ++i)
        if (i < g.n_problems && lid >= g.tile_start[i]) p = i;
    const GemmDesc& P = g.p[p];
    int rem = lid - g.tile_start[p];
    const int tiles_n = (P.n + GBN - 1) / GBN;
    const int tiles_m = (P.m + GBM - 1) / GBM;
    const int per_batch = tiles_m * tiles_n;
    const int batch = rem / per_batch;
    rem -= batch * per_batch;
    const int m0 = (rem / tiles_n) * GBM;
    const int n0 = (rem % tiles_n) * GBN;

    const T* __restrict__ A = reinterpret_cast<const T*>(P.a) + (int64_t)batch * P.stride_a;
    const T* __restrict__ B = reinterpret_cast<const T*>(P.b) + (int64_t)batch * P.stride_b;
    T* __restrict__ C = reinterpret_cast<T*>(P.c) + (int64_t)batch * P.stride_c;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;   // wave tile origin inside the block tile
    const int l31 = lane & 31, hi = lane >> 5;

    // staging: 128 rows x 8 chunks(16 B) per operand tile = 1024 chunks, 4 per thread
    const int srow = tid >> 3, scol = (tid & 7) * 8;         // + 32 rows per step
    T8 ra[4], rb[4];

    auto stage_load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = srow + 32 * i;
            const bool kin = (k0 + scol) < P.k;
            ra[i] = (kin && (m0 + r) < P.m) ? *reinterpret_cast<const T8*>(A + (int64_t)(m0 + r) * P.lda + k0 + scol)
                                            : zero8<T>();
            rb[i] = (kin && (n0 + r) < P.n) ? *reinterpret_cast<const T8*>(B + (int64_t)(n0 + r) * P.ldb + k0 + scol)
                                            : zero8<T>();
        }
    };
    auto stage_write = [&](int buf) {
        T* as = As + buf * GBM * GLD;
        T* bs = Bs + buf * GBN * GLD;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = srow + 32 * i;
            *reinterpret_cast<T8*>(as + r * GLD + scol) = ra[i];
            *reinterpret_cast<T8*>(bs + r * GLD + scol) = rb[i];
        }
    };

    f32x16 acc[2][2];   // [n block][m block]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (P.k + GBK - 1) / GBK;
    stage_load(0);
    stage_write(0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) stage_load((kt + 1) * GBK);
        const T* as = As + buf * GBM * GLD + (wm + l31) * GLD + hi * 8;
        const T* bs = Bs + buf * GBN * GLD + (wn + l31) * GLD + hi * 8;
#pragma unroll
        for (int ks = 0; ks < GBK / 16; ++ks) {
            T8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i] = *reinterpret_cast<const T8*>(as + i * 32 * GLD + ks * 16);   // rows m
                fb[i] = *reinterpret_cast<const T8*>(bs + i * 32 * GLD + ks * 16);   // rows n
            }
#pragma unroll
            for (int in = 0; in < 2; ++in)
#pragma unroll
                for (int im = 0; im < 2; ++im)
                    acc[in][im] = mfma32(fb[in], fa[im], acc[in][im]);              // D[n][m]
        }
        if (kt + 1 < nk) stage_write(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane (m = l31, hi) holds n = 8*g + 4*hi + {0..3} for g = r>>2 -------------
    const T* __restrict__ bias = reinterpret_cast<const T*>(P.bias);
#pragma unroll
    for (int im = 0; im < 2; ++im) {
        const int m = m0 + wm + im * 32 + l31;
        if (m >= P.m) continue;
#pragma unroll
        for (int in = 0; in < 2; ++in) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int n = n0 + wn + in * 32 + gq * 8 + hi * 4;
                if (n >= P.n) continue;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[in][im][gq * 4 + e];
                if (bias) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < P.n) v[e] += (float)bias[n + e];
                }
                *reinterpret_cast<T4*>(C + (int64_t)m * P.ldc + n) = cvt4<T>(v);
            }
        }
    }
}

template <typename T>
static hipError_t launch_gemm(const GemmGroup& g, int total_tiles, hipStream_t stream) {
    const size_t smem = (size_t)2 * (GBM + GBN) * GLD * sizeof(T);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&aid_gemm_nt_kernel<T>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(aid_gemm_nt_kernel<T>, dim3(total_tiles), dim3(GTHREADS), smem, stream, g);
    return hipGetLastError();
}

hipError_t gemm_group_launch(const GemmGroup& g, int dtype, hipStream_t stream) {
    int total = g.tile_start[g.n_problems];
    if (total <= 0) return hipSuccess;
    return dtype == AID_DTYPE_F16 ? launch_gemm<f16>(g, total, stream) : launch_gemm<bf16>(g, total, stream);
}

int gemm_tiles(int m, int n, int batch) {
    return ((m + GBM - 1) / GBM) * ((n + GBN - 1) / GBN) * batch;
}

}  // namespace aid
