// Grouped NT GEMM for the q / k / V^T / out projections of an AID attention layer.
//   C[b] = A[b] * B[b]^T (+ bias[n]),   A [m,k], B [n,k] (both K-contiguous), fp32 accumulate.
// Replaces attn.to_q / to_k / to_v / to_out[0] (reference interpolation.py:613, 623-624, 666).
//
// gfx950 mapping (both kernels): 128x128 block tile, 256 threads = 4 waves in a 2x2 grid, each wave
// owns a 64x64 output tile as 2x2 MFMA 32x32x16 blocks (64 fp32 accumulators / lane).  The MFMA is
// issued "transposed" (D rows = n, cols = m) so every lane ends up with 4 consecutive n of one output
// row.  The 1-D grid is remapped XCD-aware so the blocks that share an A row panel run on one XCD /
// one L2.
//
//  * aid_gemm_nt_pipe_kernel — main path (every k % BK == 0: all SD1.5 / SDXL projection shapes).
//    Operand tiles go HBM -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave-instruction, no
//    VGPR round trip, no ds_write pass) into an NS-deep ring; NS-1 tiles stay in flight across the
//    (raw) workgroup barrier with a counted s_waitcnt vmcnt, so the DMA latency is covered by NS-1
//    tiles of MFMA work instead of one.  LDS-DMA writes lane-linear images, so rows are unpadded and
//    the bank-conflict-free layout comes from an XOR swizzle applied to the per-lane SOURCE address
//    and again on the fragment read.  The C tile is staged through LDS (re-using the ring) and
//    written as full 16-B-per-lane row segments.
//  * aid_gemm_nt_kernel — edge path for ragged k (tests, odd context widths): register-staged,
//    fully guarded loads, padded LDS rows.
#include "aid_common.hpp"
#include "aid_kernels.hpp"

#include <stdlib.h>

namespace aid {

constexpr int GBM = 128, GBN = 128, GBK = 64;
constexpr int GLD = GBK + 8;            // padded LDS row (elements) of the edge kernel
constexpr int GTHREADS = 256;

struct TileCoord {
    int p, batch, m0, n0;
};

__device__ __forceinline__ TileCoord locate_tile(const GemmGroup& g) {
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
    TileCoord t;
    t.p = p;
    t.batch = rem / per_batch;
    rem -= t.batch * per_batch;
    t.m0 = (rem / tiles_n) * GBM;
    t.n0 = (rem % tiles_n) * GBN;
    return t;
}

// ------------------------------------------------------------------------------------------------
// edge kernel
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(GTHREADS) void aid_gemm_nt_kernel(const GemmGroup g) {
    typedef typename Vec<T>::v8 T8;
    typedef typename Vec<T>::v4 T4;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* As = reinterpret_cast<T*>(smem_raw);                 // [2][GBM][GLD]
    T* Bs = As + 2 * GBM * GLD;                             // [2][GBN][GLD]

    const TileCoord tc = locate_tile(g);
    const GemmDesc& P = g.p[tc.p];
    const int m0 = tc.m0, n0 = tc.n0;
    const T* __restrict__ A = reinterpret_cast<const T*>(P.a) + (int64_t)tc.batch * P.stride_a;
    const T* __restrict__ B = reinterpret_cast<const T*>(P.b) + (int64_t)tc.batch * P.stride_b;
    T* __restrict__ C = reinterpret_cast<T*>(P.c) + (int64_t)tc.batch * P.stride_c;

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

// ------------------------------------------------------------------------------------------------
// main path: NS-stage LDS-DMA ring, BK in {32, 64}
//   row bytes RB = 2*BK; a 16-B chunk c of tile row r is stored at chunk slot c ^ swz(r) with
//   swz(r) = (r >> 1) & 7 for RB = 128 and (r >> 2) & 3 for RB = 64, which makes the 16 rows of a
//   ds_read_b128 lane group hit 16 distinct 16-B slots of the 256-B LDS bank row.
// ------------------------------------------------------------------------------------------------
constexpr int G2_CLD = GBN + 8;                        // staged C row (elements), 272 B

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <typename T, int BK, int NS, int NWV>
__global__ __launch_bounds__(NWV * 64) void aid_gemm_nt_pipe_kernel(const GemmGroup g) {
    typedef typename Vec<T>::v8 T8;
    typedef typename Vec<T>::v4 T4;
    constexpr int RB = BK * 2;                          // bytes per tile row
    constexpr int CPR = RB / 16;                        // 16-B chunks per row (8 or 4)
    constexpr int RPI = 1024 / RB;                      // rows per wave DMA instruction (8 or 16)
    constexpr int NTHR = NWV * 64;
    constexpr int WNW = NWV / 2;                        // waves along n (2 or 4); 2 waves along m
    constexpr int NB = GBN / WNW / 32;                  // 32-wide n blocks per wave (2 or 1); 2 m blocks per wave
    constexpr int IPW = GBM / RPI / NWV;                // DMA instructions per wave per operand tile
    constexpr int STAGE = (GBM + GBN) * RB;             // bytes per stage
    constexpr int DPT = 2 * IPW;                        // DMA instructions per wave per K tile
    static_assert(NS >= 2 && NS <= 4, "ring depth");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];

    const TileCoord tc = locate_tile(g);
    const GemmDesc& P = g.p[tc.p];
    const int m0 = tc.m0, n0 = tc.n0;
    const T* __restrict__ A = reinterpret_cast<const T*>(P.a) + (int64_t)tc.batch * P.stride_a;
    const T* __restrict__ B = reinterpret_cast<const T*>(P.b) + (int64_t)tc.batch * P.stride_b;
    T* __restrict__ C = reinterpret_cast<T*>(P.c) + (int64_t)tc.batch * P.stride_c;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave / WNW) * 64, wn = (wave % WNW) * (32 * NB);
    const int l31 = lane & 31, hi = lane >> 5;

    auto swz = [](int r) { return CPR == 8 ? ((r >> 1) & 7) : ((r >> 2) & 3); };

    // ---- DMA source pointers: wave-instruction j covers tile rows RPI*(IPW*wave+j) .. ; lane -> (row, slot)
    const T* asrc[IPW];
    const T* bsrc[IPW];
#pragma unroll
    for (int j = 0; j < IPW; ++j) {
        const int row = RPI * (IPW * wave + j) + lane / CPR;
        const int c = (lane % CPR) ^ swz(row);                               // logical chunk stored at slot lane%CPR
        const int ra = min(m0 + row, P.m - 1), rb = min(n0 + row, P.n - 1);    // clamp: rows past the edge are never stored
        asrc[j] = A + (int64_t)ra * P.lda + c * 8;
        bsrc[j] = B + (int64_t)rb * P.ldb + c * 8;
    }
    auto dma = [&](int stage, int k0) __attribute__((always_inline)) {
        char* sa = smem_raw + stage * STAGE + wave * (IPW * 1024);
        char* sb = sa + GBM * RB;
#pragma unroll
        for (int j = 0; j < IPW; ++j) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[j] + k0),
                                             (__attribute__((address_space(3))) void*)(sa + j * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc[j] + k0),
                                             (__attribute__((address_space(3))) void*)(sb + j * 1024), 16, 0, 0);
        }
    };

    // ---- fragment read offsets (bytes inside a stage) ----------------------------------------------
    int aoff[2], boff[NB], ax[2], bx[NB];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ra = wm + i * 32 + l31;
        aoff[i] = ra * RB;
        ax[i] = hi ^ swz(ra);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int rb = wn + i * 32 + l31;
        boff[i] = GBM * RB + rb * RB;
        bx[i] = hi ^ swz(rb);
    }

    f32x16 acc[NB][2];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = P.k / BK;
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nk) dma(s, s * BK);

    int stage = 0, fill = NS - 1;              // ring slot of tile kt; slot the next DMA goes to
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt has landed once at most (tiles issued after it) x DPT of this wave's DMAs are outstanding
        if (kt + NS - 2 < nk) wait_vmcnt<(NS - 2) * DPT>();
        else                  wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();          // every wave's part of tile kt is in LDS; slot `fill` is no longer read
        asm volatile("" ::: "memory");
        if (kt + NS - 1 < nk) dma(fill, (kt + NS - 1) * BK);
        const char* st = smem_raw + stage * STAGE;
        // fragment reads run one k-step ahead of the MFMAs that consume them
        T8 fa[2][2], fb[2][NB];
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[0][i] = *reinterpret_cast<const T8*>(st + aoff[i] + ((0 ^ ax[i]) << 4));
#pragma unroll
        for (int i = 0; i < NB; ++i) fb[0][i] = *reinterpret_cast<const T8*>(st + boff[i] + ((0 ^ bx[i]) << 4));
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            if (ks + 1 < BK / 16) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    fa[nxt][i] = *reinterpret_cast<const T8*>(st + aoff[i] + (((2 * ks + 2) ^ ax[i]) << 4));
#pragma unroll
                for (int i = 0; i < NB; ++i)
                    fb[nxt][i] = *reinterpret_cast<const T8*>(st + boff[i] + (((2 * ks + 2) ^ bx[i]) << 4));
            }
#pragma unroll
            for (int in = 0; in < NB; ++in)
#pragma unroll
                for (int im = 0; im < 2; ++im) acc[in][im] = mfma32(fb[cur][in], fa[cur][im], acc[in][im]);
        }
        stage = (stage + 1 == NS) ? 0 : stage + 1;
        fill = (fill + 1 == NS) ? 0 : fill + 1;
    }
    __syncthreads();                           // everyone is done reading the ring

    // ---- epilogue: acc (+bias) -> LDS C tile -> coalesced 16-B row segments -----------------------------
    T* Cs = reinterpret_cast<T*>(smem_raw);    // [GBM][G2_CLD]
    const T* __restrict__ bias = reinterpret_cast<const T*>(P.bias);
    const bool bias_vec = (reinterpret_cast<uintptr_t>(bias) & 7) == 0;
#pragma unroll
    for (int in = 0; in < NB; ++in)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int nl = wn + in * 32 + gq * 8 + hi * 4;
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (bias) {
                if (bias_vec && n0 + nl + 4 <= P.n) {
                    bv = up4<T>(*reinterpret_cast<const T4*>(bias + n0 + nl));     // one 8-B load
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n0 + nl + e < P.n) bv[e] = (float)bias[n0 + nl + e];
                }
            }
#pragma unroll
            for (int im = 0; im < 2; ++im) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[in][im][gq * 4 + e] + bv[e];
                *reinterpret_cast<T4*>(Cs + (wm + im * 32 + l31) * G2_CLD + nl) = cvt4<T>(v);
            }
        }
    __syncthreads();
    const bool vec_ok = (P.ldc % 8 == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0);
#pragma unroll
    for (int it = 0; it < (GBM * GBN / 8) / NTHR; ++it) {
        const int id = tid + it * NTHR;
        const int row = id >> 4, ch = (id & 15) * 8;
        const int m = m0 + row, n = n0 + ch;
        if (m >= P.m || n >= P.n) continue;
        const T8 v = *reinterpret_cast<const T8*>(Cs + row * G2_CLD + ch);
        T* dst = C + (int64_t)m * P.ldc + n;
        if (vec_ok && n + 8 <= P.n) {
            *reinterpret_cast<T8*>(dst) = v;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (n + e < P.n) dst[e] = v[e];
        }
    }
}

template <typename K>
static hipError_t launch_with_smem(K kernel, size_t smem, bool* attr_set, const GemmGroup& g, int total_tiles,
                                   hipStream_t stream, int threads = GTHREADS) {
    if (!*attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        *attr_set = true;
    }
    hipLaunchKernelGGL(kernel, dim3(total_tiles), dim3(threads), smem, stream, g);
    return hipGetLastError();
}

template <typename T>
static hipError_t launch_gemm(const GemmGroup& g, int total_tiles, hipStream_t stream) {
    bool k64 = true, k32 = true;
    for (int i = 0; i < g.n_problems; ++i) {
        k64 = k64 && (g.p[i].k % 64 == 0);
        k32 = k32 && (g.p[i].k % 32 == 0);
    }
    // development knob (tools/kbench.py): AID_GEMM_VARIANT = 0 edge, 1 BK64xNS2, 2 BK64xNS3, 3 BK64xNS4, 4 BK32xNS4,
    // 5 BK32xNS3, 6 BK32xNS2 (4 waves); 7 BK64xNS2, 8 BK64xNS3 (8 waves, 64x32 wave tiles).  Default 7: best on the
    // SD1.5 / SDXL projection shapes (measured, profiles/r01_gemm_variants.txt)
    static const int variant = getenv("AID_GEMM_VARIANT") ? atoi(getenv("AID_GEMM_VARIANT")) : 7;
    static bool s0 = false, s1 = false, s2 = false, s3 = false, s4 = false, s5 = false, s6 = false, s7 = false, s8 = false;
    if (k64 && variant == 7)
        return launch_with_smem(aid_gemm_nt_pipe_kernel<T, 64, 2, 8>, 2 * 32768, &s7, g, total_tiles, stream, 512);
    if (k64 && variant == 8)
        return launch_with_smem(aid_gemm_nt_pipe_kernel<T, 64, 3, 8>, 3 * 32768, &s8, g, total_tiles, stream, 512);
    if (k32 && variant == 5)
        return launch_with_smem(aid_gemm_nt_pipe_kernel<T, 32, 3, 4>, 3 * 16384, &s5, g, total_tiles, stream);
    if (k32 && variant == 6)
        return launch_with_smem(aid_gemm_nt_pipe_kernel<T, 32, 2, 4>, GBM * G2_CLD * 2, &s6, g, total_tiles, stream);
    if (k64 && variant == 1)
        return launch_with_smem(aid_gemm_nt_pipe_kernel<T, 64, 2, 4>, 2 * 32768, &s1, g, total_tiles, stream);
    if (k64 && variant == 2)
        return launch_with_smem(aid_gemm_nt_pipe_kernel<T, 64, 3, 4>, 3 * 32768, &s2, g, total_tiles, stream);
    if (k64 && variant == 3)
        return launch_with_smem(aid_gemm_nt_pipe_kernel<T, 64, 4, 4>, 4 * 32768, &s3, g, total_tiles, stream);
    if (k32 && variant == 4)
        return launch_with_smem(aid_gemm_nt_pipe_kernel<T, 32, 4, 4>, 4 * 16384, &s4, g, total_tiles, stream);
    return launch_with_smem(aid_gemm_nt_kernel<T>, (size_t)2 * (GBM + GBN) * GLD * sizeof(T), &s0, g, total_tiles, stream);
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
