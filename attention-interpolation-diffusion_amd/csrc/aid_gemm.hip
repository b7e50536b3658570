// Grouped NT GEMM for the q / k / V^T / out projections of an AID attention layer.
//   C[b] = A[b] * B[b]^T (+ bias[n]),   A [m,k], B [n,k] (both K-contiguous), fp32 accumulate.
// Replaces attn.to_q / to_k / to_v / to_out[0] (reference interpolation.py:613, 623-624, 666).
//
// gfx950 mapping: BM x BN block tile, waves in a WM x WN grid, each wave owns MB x NB MFMA 32x32x16
// blocks.  The MFMA is issued "transposed" (D rows = n, cols = m) so every lane ends up with 4
// consecutive n of one output row.  The 1-D grid is remapped XCD-aware so the blocks that share an A
// row panel run on one XCD / one L2.
//
//  * aid_gemm_nt_pipe_kernel — lock-step engine, 128 x 128 x 64 tiles, 2 workgroups / CU (every k % 64 == 0 shape
//    can run on it).  Operand tiles go HBM -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave-instruction,
//    no VGPR round trip, no ds_write pass) into a 2-deep ring with a (raw) workgroup barrier and a counted
//    s_waitcnt vmcnt per K tile.  LDS-DMA writes lane-linear images, so rows are unpadded and the
//    bank-conflict-free layout comes from an XOR swizzle applied to the per-lane SOURCE address and again on the
//    fragment read.  Fragment reads run one k-step ahead of the MFMAs.  The C tile is staged through LDS (re-using
//    the ring) and written as full 16-B-per-lane row segments; an optional residual is added there, after the
//    rounding, like the transformer block's separate add.
//  * aid_gemm_nt_pp_kernel — ping-pong engine, 256 x 256 x 64 tiles, 1 workgroup / CU: two groups of four waves run
//    one barrier apart so one group's MFMAs cover the other group's LDS reads (struct PingPong); the ragged last
//    round of big tiles is cut into 128 x 128 tiles that the lock-step engine computes inside the same launch.
//    launch_gemm() picks the engine per launch with a small cost model (long K loops + many tiles -> ping-pong).
//  * aid_gemm_nt_kernel — edge path for ragged k (tests, odd context widths): 128x128 tile,
//    register-staged, fully guarded loads, padded LDS rows.
#include "aid_common.hpp"
#include "aid_kernels.hpp"

#include <stdlib.h>
#include <string.h>

#include <type_traits>

namespace aid {

struct TileCoord {
    int p, batch, m0, n0;
};

// Fence between the MAC loop and an epilogue.  hipcc's hazard recogniser counts the wait states between an MFMA and the
// first VALU access of its result along the fall-through path; with a branch in between (`if (bias)`, `if (ln_side)`, a
// row guard) the taken path can be shorter — the hazard that bit the ragged attention tile (profiles/r02_attn_notes.txt).
// Nothing of the kind was observed here; 20 wait states per tile are cheap insurance.
template <int NB, int MB>
__device__ __forceinline__ void mfma_fence(f32x16 (&acc)[NB][MB]) {
    // volatile asm statements keep their order: every MFMA (producer of an accumulator) sits above the first ties, the
    // wait states between the two rows of ties, every use of an accumulator below the second
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < MB; ++j) asm volatile("" : "+v"(acc[i][j]));
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < MB; ++j) asm volatile("" : "+v"(acc[i][j]));
}

// Folded LayerNorm (GemmDesc.ln_*): acc = x . W'[row] with W' = W * gamma  ->  rstd * (acc - mean * colsum[row]) + shift[row].
// `m`, `n` are the output coordinates of v[0]; v[e] sits at column n + e.  Reads are guarded, out-of-range entries are
// never stored by the callers.
__device__ __forceinline__ void ln_fix4(const GemmDesc& P, const float* __restrict__ stats, int m, int n, f32x4& v) {
    if (P.ln_side == 1) {                               // statistics per output row, weight constants per column
        const float mu = m < P.m ? stats[2 * m] : 0.f, rs = m < P.m ? stats[2 * m + 1] : 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int nn = n + e < P.n ? n + e : P.n - 1;
            float t = fmaf(-mu, P.ln_colsum[nn], v[e]);
            asm volatile("" : "+v"(t));                 // scalar on purpose, see Engine::store_tile
            v[e] = fmaf(rs, t, P.ln_shift[nn]);
        }
    } else {                                            // the activation is the B operand: statistics per column
        const int mm = m < P.m ? m : P.m - 1;
        const float cs = P.ln_colsum[mm], sh = P.ln_shift[mm];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int nn = n + e < P.n ? n + e : P.n - 1;
            float t = fmaf(-stats[2 * nn], cs, v[e]);
            asm volatile("" : "+v"(t));
            v[e] = fmaf(stats[2 * nn + 1], t, sh);
        }
    }
}

// Block -> (problem, batch, tile).  The hardware places block b on XCD b % 8; xcd_remap gives every XCD a
// contiguous range of positions so neighbouring tiles share that XCD's L2.
//  * problems with equal K loops are simply concatenated (each XCD then mostly streams ONE weight matrix);
//  * when the K loops differ (the K = 2048 text-context projections of a cross-attention layer next to its
//    K = 1280 query projection) every XCD gets a contiguous chunk of EACH problem instead: with a plain
//    concatenation all long-K tiles landed on two XCDs and set the launch time (105 -> 84 us measured).
template <int BM, int BN>
__device__ __forceinline__ TileCoord locate_pos(const GemmGroup& g, int pos) {
    constexpr int NX = 8;
    int p = 0, local = 0;
    bool found = false;
    if (!g.interleave) {                                 // equal K loops: plain concatenation keeps ONE weight matrix per XCD
#pragma unroll
        for (int i = 1; i < AID_GEMM_MAX_PROBLEMS; ++i)
            if (i < g.n_problems && pos >= g.tile_start[i]) p = i;
        local = pos - g.tile_start[p];
        found = true;
    }
    for (int x = 0; x < NX && !found; ++x) {
#pragma unroll
        for (int i = 0; i < AID_GEMM_MAX_PROBLEMS; ++i) {
            if (i >= g.n_problems || found) continue;
            const int t = g.tile_start[i + 1] - g.tile_start[i];
            const int q = t / NX, r = t % NX;
            const int share = q + (x < r ? 1 : 0);       // tiles of problem i that belong to XCD x's chunk
            if (pos < share) {
                p = i;
                local = x * q + (x < r ? x : r) + pos;
                found = true;
            } else {
                pos -= share;
            }
        }
    }
    const GemmDesc& P = g.p[p];
    int rem = local;
    const int tiles_n = (P.n + BN - 1) / BN;
    const int tiles_m = (P.m + BM - 1) / BM;
    const int per_batch = tiles_m * tiles_n;
    TileCoord t;
    t.p = p;
    t.batch = rem / per_batch;
    rem -= t.batch * per_batch;
    t.m0 = (rem / tiles_n) * BM;
    t.n0 = (rem % tiles_n) * BN;
    return t;
}

template <int BM, int BN>
__device__ __forceinline__ TileCoord locate_tile(const GemmGroup& g, int idx, int count) {
    return locate_pos<BM, BN>(g, xcd_remap(idx, count));     // position in XCD-major order (bijective)
}

// ------------------------------------------------------------------------------------------------
// edge kernel (ragged k)
// ------------------------------------------------------------------------------------------------
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

    const TileCoord tc = locate_tile<GBM, GBN>(g, blockIdx.x, gridDim.x);
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
    mfma_fence(acc);
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
                if (P.ln_stats) ln_fix4(P, P.ln_stats + 2 * (int64_t)tc.batch * P.stride_stats, m, n, v);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= P.scale;
                if (bias) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < P.n) v[e] += (float)bias[n + e];
                }
                if (P.residual) {                       // added after the first rounding, like the reference's separate add
                    const T* R = reinterpret_cast<const T*>(P.residual) + (int64_t)tc.batch * P.stride_c + (int64_t)m * P.ldc + n;
                    const f32x4 r0 = up4<T>(cvt4<T>(v));
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (n + e < P.n) ? r0[e] + (float)R[e] : 0.f;
                }
                *reinterpret_cast<T4*>(C + (int64_t)m * P.ldc + n) = cvt4<T>(v);
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// Fast epilogue (tile columns inside the matrix, 16-byte aligned rows / pointers): straight-line code.
// The generic epilogue below handles every edge (ragged n, unaligned bias / residual, odd ldc) with per-element guards;
// compiled for a full tile it still carried ~250 exec-mask branches, scalar loads inside the staging loop (each one a full
// lgkmcnt wait = every LDS write drained) and one LDS round trip per global store: 10 us of a 55 us out-projection launch
// next to 6.5 us of actual HBM write time (profiles/r03_gemm_notes.txt).  Here every uniform decision is taken once per
// tile, the bias / LayerNorm constants of a column block are fetched once, and the global side is `store_rows_fast`.
// ------------------------------------------------------------------------------------------------
typedef __amdgpu_buffer_rsrc_t BufRsrc;

// one 32 x 32 accumulator block of the n-major MFMA issue: lane (l31, hi) holds row `row`, columns nl0 + 8 gq + e
template <typename T, int SIDE>
__device__ __forceinline__ void stage_block_fast(T* Cs, int cld, const f32x16& a, int row, int nl0, float scale, const f32x4 (&bv)[4],
                                                 const float* lnr, const float* lnc) {
    typedef typename Vec<T>::v4 T4;
    float lr0 = 0.f, lr1 = 0.f;
    if (SIDE) { lr0 = lnr[2 * row]; lr1 = lnr[2 * row + 1]; }
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        const int nl = nl0 + gq * 8;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = a[gq * 4 + e];
        if (SIDE) {
            const f32x4 p0 = *reinterpret_cast<const f32x4*>(lnc + 2 * nl);          // (c0, c1) of columns nl, nl + 1
            const f32x4 p1 = *reinterpret_cast<const f32x4*>(lnc + 2 * nl + 4);      // nl + 2, nl + 3
            const float c0[4] = {p0[0], p0[2], p1[0], p1[2]}, c1[4] = {p0[1], p0[3], p1[1], p1[3]};
#pragma unroll
            for (int e = 0; e < 4; ++e) {               // scalar FMAs on purpose, see Engine::store_tile
                float t = SIDE == 1 ? fmaf(-lr0, c0[e], v[e]) : fmaf(-c0[e], lr0, v[e]);
                asm volatile("" : "+v"(t));
                v[e] = SIDE == 1 ? fmaf(lr1, t, c1[e]) : fmaf(c1[e], t, lr1);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = fmaf(v[e], scale, bv[gq][e]);
            asm volatile("" : "+v"(t));                 // keeps hipcc from pairing them into v_pk_fma_f32 (slower beside nothing, and see above)
            v[e] = t;
        }
        *reinterpret_cast<T4*>(Cs + row * cld + nl) = cvt4<T>(v);
    }
}

// bias of the 16 columns a lane owns in one 32-column block (columns nl0 + 8 gq + e), fp32; zero without a bias
template <typename T>
__device__ __forceinline__ void bias_block(f32x4 (&bv)[4], const T* bias, int ncol0) {
    typedef typename Vec<T>::v4 T4;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        if (bias) bv[gq] = up4<T>(*reinterpret_cast<const T4*>(bias + ncol0 + gq * 8));
        else      bv[gq] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// LDS C tile [ROWS][CLD] -> global rows, 16 B per lane: all LDS reads of the pass group issued before the first store, store
// addresses = one 32-bit per-thread offset + a scalar pass offset.  R (optional): residual, added after the rounding.
template <typename T, int NTHR, int ROWS, int BN, int CLD>
__device__ __forceinline__ void store_rows_fast(const T* Cs, T* C, const T* R, int m0, int n0, int ldc, int rows_valid, int tid) {
    typedef typename Vec<T>::v8 T8;
    constexpr int CPRW = BN / 8, RPP = NTHR / CPRW, NP = ROWS / RPP;      // chunks per row, rows per pass, passes
    static_assert(NTHR % CPRW == 0 && ROWS % RPP == 0, "rows divide over the passes");
    constexpr int G = NP % 9 == 0 ? 9 : (NP % 8 == 0 ? 8 : (NP % 4 == 0 ? 4 : 1));     // passes in flight together
    const int r0 = tid / CPRW, ch = (tid % CPRW) * 8;
    // (the scalar offset is not part of the descriptor's range check: rows past the matrix are predicated, not "dropped")
    const BufRsrc rc = __builtin_amdgcn_make_buffer_rsrc(C + (int64_t)m0 * ldc + n0, 0, 0x7fffffff, 0x00020000);
    const BufRsrc rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(R ? R : C) + (int64_t)m0 * ldc + n0, 0, 0x7fffffff, 0x00020000);
    const int voff = (r0 * ldc + ch) * 2;
    const int pass = RPP * ldc * 2;                     // bytes between passes (scalar)
    const T* src = Cs + r0 * CLD + ch;
#pragma unroll
    for (int g0 = 0; g0 < NP; g0 += G) {
        T8 v[G], r[G];
        if (R) {
#pragma unroll
            for (int i = 0; i < G; ++i)
                r[i] = (r0 + RPP * (g0 + i) < rows_valid)
                           ? __builtin_bit_cast(T8, __builtin_amdgcn_raw_buffer_load_b128(rr, voff, (g0 + i) * pass, 0)) : zero8<T>();
        }
#pragma unroll
        for (int i = 0; i < G; ++i) v[i] = *reinterpret_cast<const T8*>(src + (g0 + i) * RPP * CLD);
#pragma unroll
        for (int i = 0; i < G; ++i) {
            T8 o = v[i];
            if (R) o = cvt8<T>(up8<T>(v[i]) + up8<T>(r[i]));
            if (r0 + RPP * (g0 + i) < rows_valid)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rc, voff, (g0 + i) * pass, AID_ST_AUX);
        }
    }
}

// may this tile take the fast epilogue?
template <typename T>
__device__ __forceinline__ bool epilogue_fast_ok(const GemmDesc& P, const T* C, const T* R, int n0, int bn) {
    return n0 + bn <= P.n && P.ldc % 8 == 0 && P.ldc < (1 << 20) && (reinterpret_cast<uintptr_t>(C) & 15) == 0 &&
           (reinterpret_cast<uintptr_t>(P.bias) & 7) == 0 && (!R || (reinterpret_cast<uintptr_t>(R) & 15) == 0);
}

// ------------------------------------------------------------------------------------------------
// main path: NS-stage LDS-DMA ring
//   row bytes RB = 2*BK; a 16-B chunk c of tile row r is stored at chunk slot c ^ swz(r) with
//   swz(r) = (r >> 1) & 7 for RB = 128 and (r >> 2) & 3 for RB = 64, which makes the 16 rows of a
//   ds_read_b128 lane group hit 16 distinct 16-B slots of the 256-B LDS bank row.
// ------------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Everything one workgroup needs to produce (part of) one BM x BN output tile: DMA addressing, the pipelined
// MAC loop over a range of K tiles, the LDS-staged store.  All members are force-inlined; the accumulators live in registers.
template <typename T, int BM, int BN, int BK, int NS, int WM, int WN>
struct Engine {
    typedef typename Vec<T>::v8 T8;
    typedef typename Vec<T>::v4 T4;
    static constexpr int NWV = WM * WN, NTHR = NWV * 64;
    static constexpr int MB = BM / WM / 32, NB = BN / WN / 32;      // 32x32 blocks per wave along m / n
    static constexpr int RB = BK * 2;                          // bytes per tile row
    static constexpr int CPR = RB / 16;                        // 16-B chunks per row (8 or 4)
    static constexpr int RPI = 1024 / RB;                      // rows per wave DMA instruction (8 or 16)
    static constexpr int IPA = BM / RPI / NWV, IPB = BN / RPI / NWV;   // DMA instructions per wave per A / B tile
    static constexpr int STAGE = (BM + BN) * RB;               // bytes per stage
    static constexpr int DPT = IPA + IPB;                      // DMA instructions per wave per K tile
    static constexpr int CLD = BN + 8;                         // staged C row (elements)
    static constexpr size_t LNS = (size_t)(BM + BN) * 2 * sizeof(float);      // folded LayerNorm: per-row / per-column pairs of a tile
    static constexpr size_t SMEM = ((size_t)NS * STAGE > (size_t)BM * CLD * 2 + LNS) ? (size_t)NS * STAGE : (size_t)BM * CLD * 2 + LNS;
    static_assert(BM % (WM * 32) == 0 && BN % (WN * 32) == 0, "wave tiles are multiples of 32x32");
    static_assert(BM % (RPI * NWV) == 0 && BN % (RPI * NWV) == 0, "DMA instructions divide evenly over the waves");
    static_assert(NS >= 2 && NS <= 8, "ring depth");

    char* smem;
    int tid, lane, wave, wm, wn, l31, hi;
    int aoff[MB], boff[NB], ax[MB], bx[NB];
    const T* asrc[IPA];
    const T* bsrc[IPB];
    f32x16 acc[NB][MB];

    static __device__ __forceinline__ int swz(int r) { return CPR == 8 ? ((r >> 1) & 7) : ((r >> 2) & 3); }

    __device__ __forceinline__ void init(char* smem_) {
        smem = smem_;
        tid = threadIdx.x;
        lane = tid & 63;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        wm = (wave / WN) * (32 * MB);
        wn = (wave % WN) * (32 * NB);
        l31 = lane & 31;
        hi = lane >> 5;
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            const int ra = wm + i * 32 + l31;
            aoff[i] = ra * RB;
            ax[i] = hi ^ swz(ra);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int rb = wn + i * 32 + l31;
            boff[i] = BM * RB + rb * RB;
            bx[i] = hi ^ swz(rb);
        }
    }

    // DMA source pointers: wave-instruction j covers tile rows RPI*(IP*wave+j) .. ; lane -> (row, slot)
    __device__ __forceinline__ void set_tile(const GemmDesc& P, const T* A, const T* B, int m0, int n0) {
#pragma unroll
        for (int j = 0; j < IPA; ++j) {
            const int row = RPI * (IPA * wave + j) + lane / CPR;
            const int c = (lane % CPR) ^ swz(row);                           // logical chunk stored at slot lane%CPR
            asrc[j] = A + (int64_t)min(m0 + row, P.m - 1) * P.lda + c * 8;     // clamp: rows past the edge are never stored
        }
#pragma unroll
        for (int j = 0; j < IPB; ++j) {
            const int row = RPI * (IPB * wave + j) + lane / CPR;
            const int c = (lane % CPR) ^ swz(row);
            bsrc[j] = B + (int64_t)min(n0 + row, P.n - 1) * P.ldb + c * 8;
        }
    }

    __device__ __forceinline__ void zero_acc() {
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int j = 0; j < MB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }

    __device__ __forceinline__ void dma(int stage, int k0) {
        char* sa = smem + stage * STAGE + wave * (IPA * 1024);
        char* sb = smem + stage * STAGE + BM * RB + wave * (IPB * 1024);
#pragma unroll
        for (int j = 0; j < IPA; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[j] + k0),
                                             (__attribute__((address_space(3))) void*)(sa + j * 1024), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < IPB; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc[j] + k0),
                                             (__attribute__((address_space(3))) void*)(sb + j * 1024), 16, 0, 0);
    }

    // acc += A[:, kb*BK : ke*BK] * B[:, kb*BK : ke*BK]^T through the NS-deep LDS-DMA ring.
    // Precondition: no VMEM operation of this wave outstanding, nobody still reads the ring.
    __device__ __forceinline__ void mac(int kb, int ke) {
        const int nk = ke - kb;
#pragma unroll
        for (int s = 0; s < NS - 1; ++s)
            if (s < nk) dma(s, (kb + s) * BK);
        int stage = 0, fill = NS - 1;          // ring slot of tile kt; slot the next DMA goes to
        for (int kt = 0; kt < nk; ++kt) {
            // tile kt has landed once at most (tiles issued after it) x DPT of this wave's DMAs are outstanding
            if (kt + NS - 2 < nk) wait_vmcnt<(NS - 2) * DPT>();
            else                  wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();      // every wave's part of tile kt is in LDS; slot `fill` is no longer read
            asm volatile("" ::: "memory");
            if (kt + NS - 1 < nk) dma(fill, (kb + kt + NS - 1) * BK);
            const char* st = smem + stage * STAGE;
            // fragment reads run one k-step ahead of the MFMAs that consume them
            T8 fa[2][MB], fb[2][NB];
#pragma unroll
            for (int i = 0; i < MB; ++i) fa[0][i] = *reinterpret_cast<const T8*>(st + aoff[i] + ((0 ^ ax[i]) << 4));
#pragma unroll
            for (int i = 0; i < NB; ++i) fb[0][i] = *reinterpret_cast<const T8*>(st + boff[i] + ((0 ^ bx[i]) << 4));
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                const int cur = ks & 1, nxt = cur ^ 1;
                if (ks + 1 < BK / 16) {
#pragma unroll
                    for (int i = 0; i < MB; ++i)
                        fa[nxt][i] = *reinterpret_cast<const T8*>(st + aoff[i] + (((2 * ks + 2) ^ ax[i]) << 4));
#pragma unroll
                    for (int i = 0; i < NB; ++i)
                        fb[nxt][i] = *reinterpret_cast<const T8*>(st + boff[i] + (((2 * ks + 2) ^ bx[i]) << 4));
                }
#pragma unroll
                for (int in = 0; in < NB; ++in)
#pragma unroll
                    for (int im = 0; im < MB; ++im) acc[in][im] = mfma32(fb[cur][in], fa[cur][im], acc[in][im]);
            }
            stage = (stage + 1 == NS) ? 0 : stage + 1;
            fill = (fill + 1 == NS) ? 0 : fill + 1;
        }
        __syncthreads();                       // everyone is done reading the ring
    }

    template <int SIDE>
    __device__ __forceinline__ void stage_all_fast(const GemmDesc& P, T* Cs, int n0, const float* lnr, const float* lnc) {
        const T* bias = reinterpret_cast<const T*>(P.bias);
        const float scale = P.scale;
#pragma unroll
        for (int in = 0; in < NB; ++in) {
            f32x4 bv[4];
            bias_block<T>(bv, bias, n0 + wn + in * 32 + hi * 4);
#pragma unroll
            for (int im = 0; im < MB; ++im)
                stage_block_fast<T, SIDE>(Cs, CLD, acc[in][im], wm + im * 32 + l31, wn + in * 32 + hi * 4, scale, bv, lnr, lnc);
        }
    }

    // acc (+bias) -> LDS C tile -> coalesced 16-B row segments.  Ends with all stores drained and a barrier.
    // `R` (optional, laid out like C) is added after the rounding to T — the transformer block's residual add.
    __device__ __forceinline__ void store_tile(const GemmDesc& P, T* C, int m0, int n0, const T* R = nullptr,
                                               const float* stats = nullptr) {
        T* Cs = reinterpret_cast<T*>(smem);        // [BM][CLD]
        mfma_fence(acc);
        const T* __restrict__ bias = reinterpret_cast<const T*>(P.bias);
        const bool bias_vec = (reinterpret_cast<uintptr_t>(bias) & 7) == 0;
        if (epilogue_fast_ok<T>(P, C, R, n0, BN)) {             // straight-line epilogue (see stage_block_fast)
            const int side_f = stats ? P.ln_side : 0;
            float* const lnr_f = reinterpret_cast<float*>(smem + (size_t)BM * CLD * 2);
            float* const lnc_f = lnr_f + 2 * BM;
            if (side_f) {
                for (int i = tid; i < BM + BN; i += NTHR) {
                    const bool isrow = i < BM;
                    const int gi = isrow ? min(m0 + i, P.m - 1) : n0 + i - BM;
                    const bool st = isrow == (side_f == 1);
                    lnr_f[2 * i] = st ? stats[2 * gi] : P.ln_colsum[gi];
                    lnr_f[2 * i + 1] = st ? stats[2 * gi + 1] : P.ln_shift[gi];
                }
                __syncthreads();
            }
            if (side_f == 0)      stage_all_fast<0>(P, Cs, n0, lnr_f, lnc_f);
            else if (side_f == 1) stage_all_fast<1>(P, Cs, n0, lnr_f, lnc_f);
            else                  stage_all_fast<2>(P, Cs, n0, lnr_f, lnc_f);
            __syncthreads();
            store_rows_fast<T, NTHR, BM, BN, CLD>(Cs, C, R, m0, n0, P.ldc, min(BM, P.m - m0), tid);
            return;
        }
        // folded LayerNorm: the tile's per-row and per-column pairs go through LDS (behind the C tile) — one coalesced load
        // per thread instead of 8 broadcast loads per 4-column group; ln_side 1: rows carry (mean, rstd), columns
        // (colsum, shift); ln_side 2: the other way round
        float lr0[MB], lr1[MB];
        const int side = stats ? P.ln_side : 0;
        float* const lnr = reinterpret_cast<float*>(smem + (size_t)BM * CLD * 2);     // [BM][2]
        float* const lnc = lnr + 2 * BM;                                              // [BN][2]
        if (side) {
            for (int i = tid; i < BM + BN; i += NTHR) {
                const bool isrow = i < BM;
                const int g = isrow ? min(m0 + i, P.m - 1) : min(n0 + i - BM, P.n - 1);
                float q0, q1;
                if (isrow == (side == 1)) { q0 = stats[2 * g]; q1 = stats[2 * g + 1]; }
                else                      { q0 = P.ln_colsum[g]; q1 = P.ln_shift[g]; }
                lnr[2 * i] = q0;                        // lnc follows lnr: index i runs through both
                lnr[2 * i + 1] = q1;
            }
            __syncthreads();
#pragma unroll
            for (int im = 0; im < MB; ++im) {
                lr0[im] = lnr[2 * (wm + im * 32 + l31)];
                lr1[im] = lnr[2 * (wm + im * 32 + l31) + 1];
            }
        }
#pragma unroll
        for (int in = 0; in < NB; ++in)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int nl = wn + in * 32 + gq * 8 + hi * 4;
                f32x4 lc0 = {0.f, 0.f, 0.f, 0.f}, lc1 = {0.f, 0.f, 0.f, 0.f};
                if (side) {
                    const f32x4 p0 = *reinterpret_cast<const f32x4*>(lnc + 2 * nl);          // pairs of columns nl, nl + 1
                    const f32x4 p1 = *reinterpret_cast<const f32x4*>(lnc + 2 * nl + 4);      // nl + 2, nl + 3
                    lc0[0] = p0[0]; lc1[0] = p0[1]; lc0[1] = p0[2]; lc1[1] = p0[3];
                    lc0[2] = p1[0]; lc1[2] = p1[1]; lc0[3] = p1[2]; lc1[3] = p1[3];
                }
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
                for (int im = 0; im < MB; ++im) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[in][im][gq * 4 + e];
                    // Scalar FMAs on purpose (the empty asm keeps hipcc from pairing them): as v_pk_fma_f32 with an
                    // op_sel broadcast of the odd register of the (lr0[0], lr0[1]) pair, the product of the FIRST FMA
                    // sporadically came out as 0 in lanes 48-63 of the low half — ~2e-5 of the outputs of a
                    // 57344 x 640 x 640 problem, different elements every run (found by the parity test of this epilogue;
                    // wait states, vmcnt(0), uncached loads and LDS staging of the operands all left it in place).
                    if (side == 1) {                    // rstd_m (acc - mean_m colsum_n) + shift_n
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float t = fmaf(-lr0[im], lc0[e], v[e]);
                            asm volatile("" : "+v"(t));
                            v[e] = fmaf(lr1[im], t, lc1[e]);
                        }
                    } else if (side == 2) {             // rstd_n (acc - mean_n colsum_m) + shift_m
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float t = fmaf(-lc0[e], lr0[im], v[e]);
                            asm volatile("" : "+v"(t));
                            v[e] = fmaf(lc1[e], t, lr1[im]);
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], P.scale, bv[e]);
                    *reinterpret_cast<T4*>(Cs + (wm + im * 32 + l31) * CLD + nl) = cvt4<T>(v);
                }
            }
        __syncthreads();
        const bool vec_ok = (P.ldc % 8 == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0);
        constexpr int CPRW = BN / 8;               // 16-B chunks per C row
#pragma unroll
        for (int it = 0; it < (BM * CPRW) / NTHR; ++it) {
            const int id = tid + it * NTHR;
            const int row = id / CPRW, ch = (id % CPRW) * 8;
            const int m = m0 + row, n = n0 + ch;
            if (m >= P.m || n >= P.n) continue;
            T8 v = *reinterpret_cast<const T8*>(Cs + row * CLD + ch);
            T* dst = C + (int64_t)m * P.ldc + n;
            const T* res = R ? R + (int64_t)m * P.ldc + n : nullptr;
            if (vec_ok && n + 8 <= P.n) {
                if (res) {
                    const bool rvec = (reinterpret_cast<uintptr_t>(R) & 15) == 0;
                    T8 r;
                    if (rvec) {
                        r = *reinterpret_cast<const T8*>(res);
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) r[e] = res[e];
                    }
                    const f32x8 a = up8<T>(v), b = up8<T>(r);
                    v = cvt8<T>(a + b);
                }
                *reinterpret_cast<T8*>(dst) = v;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (n + e < P.n) dst[e] = res ? (T)((float)v[e] + (float)res[e]) : v[e];
            }
        }
    }
};

// ---- 256 x 256 tile, two wave groups in ping-pong ---------------------------------------------------------
// 8 waves as 2 (m) x 4 (n), wave tile 128 x 64 = 4 x 2 MFMA blocks.  The four waves with wr = 0 and the four
// with wr = 1 (one of each per SIMD) run the same program ONE BARRIER APART: a K tile is two phases, a phase is
// [fragment reads | barrier | 16 MFMAs on one half (64 x 64) of the wave tile, 4 DMA issues in between | barrier],
// so while one group's MFMAs occupy the matrix pipes the other group's ds_reads use the LDS.  (The lock-step loop of Engine::mac leaves the matrix pipe idle during every read burst: 50 %
// MFMA-busy at 256 x 256, LDS-bound at 128 x 128 — profiles/r01_gemm_variants.txt.)
//   LDS: 2 parities x [A0 A1 B0 B1] half-tiles of 128 rows x 128 B.  A half h holds tile rows with bit 6 == h
//   (the wave's blocks 2h, 2h+1), B half j the columns with bit 5 == j (the wave's block j).
//   DMA stream in read order, half-tiles of 2 instructions per wave:  item s = 4 kt + q, q: 0 = B0, 1 = B1,
//   2 = A0, 3 = A1;  phase P of K tile kt reads  P0: B0, B1, A0   P1: A1  and issues items 4 kt + 2 P + 6, + 7 between
//   its MFMAs (6 items of lead).  The READ slot ends with a counted vmcnt that retires what the NEXT phase reads,
//   followed by the barrier, so every wave's part of a half-tile is both landed and barrier-published before
//   anybody reads it; a buffer is refilled >= 2 slots after the lgkmcnt that completed its last read.
template <typename T>
struct PingPong : Engine<T, 256, 256, 64, 2, 2, 4> {
    typedef Engine<T, 256, 256, 64, 2, 2, 4> Base;
    typedef typename Vec<T>::v8 T8;
    static constexpr int HALF = 128 * 128;          // bytes per half-tile
    static constexpr int STG = 4 * HALF;            // one parity
    static constexpr size_t SMEM = Base::SMEM;
    using Base::smem; using Base::tid; using Base::lane; using Base::wave; using Base::wm; using Base::wn;
    using Base::l31; using Base::hi; using Base::aoff; using Base::boff; using Base::ax; using Base::bx;
    using Base::acc;
    typedef __amdgpu_buffer_rsrc_t Rsrc;
    int wr;
    // DMA source addressing: one buffer resource per operand tile (base = first row of the tile) + a 32-bit byte
    // offset per lane + the K position as scalar offset.  A `buffer_load ... lds` hands the address unit 4 B per
    // lane and needs no VALU; the flat form (64-bit address per lane, v_lshl_add_u64 per DMA) kept the SIMD's issue
    // busy ~29 cycles per DMA — with 16 DMAs per K tile and SIMD that was +20 % on the MFMA-bound loop (PMC:
    // SQ_ACTIVE_INST_ANY, profiles/r01_gemm_variants.txt).
    Rsrc ra, rb;
    int avo[4], bvo[4];

    static __device__ __forceinline__ int swz(int r) { return (r >> 1) & 7; }

    __device__ __forceinline__ void init(char* smem_) {
        smem = smem_;
        tid = threadIdx.x;
        lane = tid & 63;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        wr = wave >> 2;
        wm = wr * 128;
        wn = (wave & 3) * 64;
        l31 = lane & 31;
        hi = lane >> 5;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int lr = wr * 64 + e * 32 + l31;      // row inside an A half-tile
            aoff[e] = lr * 128;
            ax[e] = hi ^ swz(lr);
        }
        const int lb = (wave & 3) * 32 + l31;           // row inside a B half-tile
        boff[0] = 2 * HALF + lb * 128;
        bx[0] = hi ^ swz(lb);
    }

    __device__ __forceinline__ void set_tile(const GemmDesc& P, const T* A, const T* B, int m0, int n0) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int lrow = 8 * (2 * wave + j) + (lane >> 3);
                const int c = (lane & 7) ^ swz(lrow);
                const int row_a = (lrow >> 6) * 128 + h * 64 + (lrow & 63);
                const int row_b = (lrow >> 5) * 64 + h * 32 + (lrow & 31);
                avo[h * 2 + j] = min(row_a, P.m - 1 - m0) * (P.lda * 2) + c * 16;     // clamp: rows past the edge are never stored
                bvo[h * 2 + j] = min(row_b, P.n - 1 - n0) * (P.ldb * 2) + c * 16;
            }
        ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(A + (int64_t)m0 * P.lda), 0, 0x7fffffff, 0x00020000);
        rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(B + (int64_t)n0 * P.ldb), 0, 0x7fffffff, 0x00020000);
    }

    // DMA j (0 / 1) of half-tile Q: 0 = B0, 1 = B1, 2 = A0, 3 = A1
    template <int Q>
    __device__ __forceinline__ void dma_one(int parity, int k0, int j) {
        constexpr bool IS_A = Q >= 2;
        constexpr int H = Q & 1;
        char* dst = smem + parity * STG + (IS_A ? 0 : 2 * HALF) + H * HALF + wave * 2048 + j * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(IS_A ? ra : rb, (__attribute__((address_space(3))) void*)dst, 16,
                                                 IS_A ? avo[H * 2 + j] : bvo[H * 2 + j], k0 * 2, 0, 0);
    }

    static constexpr int E = 6;             // items of lead; a buffer is refilled >= 1 phase after its last read
    template <int S>
    __device__ __forceinline__ void issue_first(int kb, int nk) {        // prologue: items 0 .. E - 1
        if ((S >> 2) < nk) {
            dma_one<(S & 3)>((S >> 2) & 1, (kb + (S >> 2)) * 64, 0);
            dma_one<(S & 3)>((S >> 2) & 1, (kb + (S >> 2)) * 64, 1);
        }
        if constexpr (S + 1 < E) issue_first<S + 1>(kb, nk);
    }
    // MFMA slot of phase P (0 / 1) of K tile kt issues items 4 kt + 2 P + E and + E + 1, one DMA at a time (i = 0..3)
    template <int P>
    __device__ __forceinline__ void issue(int kt, int kb, int nk, int i) {
        constexpr int S0 = 2 * P + E;
        const int s = S0 + (i >> 1);
        const int t = kt + (s >> 2);
        if (t < nk) {
            if ((S0 & 3) == 2) {            // items A0, A1 of one tile
                if (i < 2) dma_one<2>(t & 1, (kb + t) * 64, i & 1);
                else       dma_one<3>(t & 1, (kb + t) * 64, i & 1);
            } else {                        // items B0, B1
                if (i < 2) dma_one<0>(t & 1, (kb + t) * 64, i & 1);
                else       dma_one<1>(t & 1, (kb + t) * 64, i & 1);
            }
        }
    }
    // READ slot of phase P: block until what the NEXT phase reads has landed.  Newest item issued so far is
    // 4 kt + 2 P + E - 1; P = 0 -> next reads A1(kt) = item 4 kt + 3; P = 1 -> next reads B0, B1, A0 of kt + 1 (<= 4 kt + 6)
    template <int P>
    __device__ __forceinline__ void retire(int kt, int nk) {
        constexpr int NEWEST = 2 * P + E - 1;
        constexpr int NEED = P == 0 ? 3 : 6;
        static_assert(NEWEST >= NEED, "lead too short");
        if (kt + (NEWEST >> 2) < nk) wait_vmcnt<2 * (NEWEST - NEED)>();
        else                         wait_vmcnt<0>();
    }

    __device__ __forceinline__ void read_a(T8 (&fa)[2][4], const char* st, int h) {
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                fa[e][ks] = *reinterpret_cast<const T8*>(st + h * HALF + aoff[e] + (((2 * ks) ^ ax[e]) << 4));
    }
    __device__ __forceinline__ void read_b(T8 (&fb)[2][4], const char* st) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                fb[j][ks] = *reinterpret_cast<const T8*>(st + boff[0] + j * HALF + (((2 * ks) ^ bx[0]) << 4));
    }
    // MFMA slot of phase P: acc[0..1][2P .. 2P+1] += B blocks x A blocks of half P; 16 MFMAs on 4 independent
    // accumulators with the phase's four DMAs in between
    template <int P, bool DMA = true, bool PRIO = false>
    __device__ __forceinline__ void half(const T8 (&fb)[2][4], const T8 (&fa)[2][4], int kt, int kb, int nk) {
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ks = i >> 2, in = (i >> 1) & 1, e = i & 1;
            acc[in][2 * P + e] = mfma32(fb[in][ks], fa[e][ks], acc[in][2 * P + e]);
            if (DMA && (i == 2 || i == 5 || i == 8 || i == 11)) {
                __builtin_amdgcn_sched_barrier(0);
                issue<P>(kt, kb, nk, (i - 2) / 3);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
    }
    static __device__ __forceinline__ void slot() {          // slot boundary: nothing moves across
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- variant: the DMAs are issued in the READ slots -----------------------------------------------------------
    // A wave in its MFMA slot issues nothing but MFMAs; the half-tiles are requested by the OTHER group's waves, which sit
    // in their read slot at that time (fragment reads, then four LDS-DMAs, then the waits, then the barrier).  In-order issue
    // means a `buffer_load ... lds` that finds the address path backed up holds everything queued behind it; here that is a
    // wave with no MFMA to issue.  Stream per wave:  B(0) A(0) B(1) | A(1) B(2) | A(2) B(3) | ...   (4 DMAs per half pair)
    //   READ-P0(kt): reads B0 B1 A0 of tile kt, requests A0 A1 of kt + 1 (parity of kt - 1: last read two slots ago),
    //                retires A1(kt) (newer in the stream: B(kt+1), A(kt+1) -> vmcnt 8)
    //   READ-P1(kt): reads A1(kt), requests B0 B1 of kt + 2 (parity of kt: read in READ-P0(kt)), retires B(kt+1) and
    //                A0(kt+1) (newer: A1(kt+1), B(kt+2) -> vmcnt 6)
    // Every read slot ends with lgkmcnt(0) BEFORE its barrier: a buffer may then be refilled by anybody two slots after
    // the slot that read it (the second group runs one barrier behind the first).
    template <int Q>
    __device__ __forceinline__ void dma_half(int t, int kb) {
        dma_one<Q>(t & 1, (kb + t) * 64, 0);
        dma_one<Q>(t & 1, (kb + t) * 64, 1);
    }
    // ABL (timing ablations, results are garbage): bit 0 = no DMAs in the loop, bit 1 = no fragment reads in the loop
    template <bool PRIO, int ABL = 0>
    __device__ __forceinline__ void mac_rd(int kb, int ke) {
        const int nk = ke - kb;
        constexpr bool NODMA = ABL & 1, NORD = ABL & 2;
        dma_half<0>(0, kb); dma_half<1>(0, kb); dma_half<2>(0, kb); dma_half<3>(0, kb);
        if (nk > 1) { dma_half<0>(1, kb); dma_half<1>(1, kb); wait_vmcnt<6>(); }
        else        wait_vmcnt<2>();
        slot();
        if (wr == 1) slot();                // the second group runs one barrier behind
        T8 fa[2][4], fb[2][4];
        if (NORD) { read_b(fb, smem); read_a(fa, smem, 0); }
        auto body = [&](int kt, auto more1_t, auto more2_t) __attribute__((always_inline)) {
            constexpr bool M1 = decltype(more1_t)::value && !NODMA, M2 = decltype(more2_t)::value && !NODMA;   // tile kt + 1 / kt + 2 exists
            const char* st = smem + (kt & 1) * STG;
            if (!NORD) {
                read_b(fb, st);
                read_a(fa, st, 0);
            } else {
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) { asm volatile("" : "+v"(fa[e][ks])); asm volatile("" : "+v"(fb[e][ks])); }
            }
            if (M1) { dma_half<2>(kt + 1, kb); dma_half<3>(kt + 1, kb); }
            if (M1) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            else    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            slot();
            half<0, false, PRIO>(fb, fa, kt, kb, nk);
            slot();
            if (!NORD) {
                read_a(fa, st, 1);
            } else {
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(fa[e][ks]));
            }
            if (M2) { dma_half<0>(kt + 2, kb); dma_half<1>(kt + 2, kb); }
            if (M2)      asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
            else if (M1) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
            else         asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            slot();
            half<1, false, PRIO>(fb, fa, kt, kb, nk);
            slot();
        };
        const std::true_type Y{};
        const std::false_type N{};
        int kt = 0;
        for (; kt + 2 < nk; ++kt) body(kt, Y, Y);           // steady state: no branch inside
        if (kt + 1 < nk) { body(kt, Y, N); ++kt; }
        if (kt < nk) body(kt, N, N);
        if (wr == 0) slot();
        wait_vmcnt<0>();
        __syncthreads();                    // everyone is done reading the ring
    }

    // Precondition: no VMEM operation of this wave outstanding, nobody still reads the LDS.
    __device__ __forceinline__ void mac(int kb, int ke) {
        const int nk = ke - kb;
        issue_first<0>(kb, nk);
        if (nk >= 2) wait_vmcnt<2 * (E - 3)>();             // B0, B1, A0 of the first tile landed
        else         wait_vmcnt<0>();
        slot();
        if (wr == 1) slot();                // the second group runs one barrier behind
        T8 fa[2][4], fb[2][4];
        for (int kt = 0; kt < nk; ++kt) {
            const char* st = smem + (kt & 1) * STG;
            read_b(fb, st);
            read_a(fa, st, 0);
            retire<0>(kt, nk);
            slot();
            half<0>(fb, fa, kt, kb, nk);
            slot();
            read_a(fa, st, 1);
            retire<1>(kt, nk);
            slot();
            half<1>(fb, fa, kt, kb, nk);
            slot();
        }
        if (wr == 0) slot();
        wait_vmcnt<0>();
        __syncthreads();                    // everyone is done reading the ring
    }
};

// The 128 x 128 tiles inside a ping-pong launch (side problems, ragged last round) have the CU to themselves — the launch
// reserves the big tile's 128 KB of LDS per workgroup — so their DMA ring is 4 deep instead of the lock-step kernel's 2:
// a lone workgroup has nobody to hide the L2 round trip of the next K tile behind.
constexpr int PP_SMALL_NS = 4;
static_assert(Engine<bf16, 128, 128, 64, PP_SMALL_NS, 2, 4>::SMEM <= PingPong<bf16>::SMEM, "small tiles use the big tile's LDS");


// ---- 288 x 256 tile on the same eight waves --------------------------------------------------------------------
// The projection launches of the SDXL stack are 1.09 (out projection, cross-attention query: 280 tiles of 256 x 256 on
// 256 CUs) and 3.28 CU rounds (q / k / V^T: 840 tiles): a quarter of the out projection's time is the tail of cut-up tiles,
// an eighth of the grouped launch is the idle part of its last round.  With 288 rows per tile the same launches are
// 250 tiles (0.98 rounds) and 750 (2.93): whole rounds, 9 / 8 of the work per tile.
// The extra 32 rows cost no new wave roles: the strip is 32 x 256 = eight 32 x 32 blocks, ONE per wave, and wave
// (wr, wc) takes the block over ITS OWN B fragment fb[wr] (columns 64 wc + 32 wr), so the strip needs 4 more A fragment
// reads per K tile and wave (rows 256 .. 287, read by everybody), 4 more MFMAs (spread over phase 1, one per four main
// MFMAs: the chain on the ninth accumulator never stalls) and 16 more accumulator registers (144 + 80 fragments).
// LDS: each parity grows by the strip's 32 rows x 128 B (4 DMA pieces per K tile, issued by waves 0 - 3 with the A halves).
template <typename T>
struct PingPongX : PingPong<T> {
    typedef PingPong<T> PP;
    typedef typename Vec<T>::v8 T8;
    typedef typename Vec<T>::v4 T4;
    static constexpr int BMX = 288, XR = 32;
    static constexpr int HALF = PP::HALF;
    static constexpr int STGX = 4 * HALF + XR * 128;            // one parity: A0 A1 B0 B1 + the strip
    static constexpr int CLD = 256 + 8;
    static constexpr size_t LNS = (size_t)(BMX + 256) * 2 * sizeof(float);
    static constexpr size_t SMEMX = (size_t)BMX * CLD * 2 + LNS > (size_t)2 * STGX ? (size_t)BMX * CLD * 2 + LNS : (size_t)2 * STGX;
    static_assert(SMEMX <= 160 * 1024, "one workgroup per CU");
    using PP::smem; using PP::tid; using PP::lane; using PP::wave; using PP::wm; using PP::wn; using PP::wr;
    using PP::l31; using PP::hi; using PP::aoff; using PP::boff; using PP::ax; using PP::bx; using PP::acc;
    using PP::ra; using PP::rb; using PP::avo; using PP::bvo;
    f32x16 accx;
    int xoff, xx, xvo;
    int kmul = 128;             // bytes per K tile in the DMA's scalar offset (development order 4 freezes it at 0: every request an L2 hit)
#ifdef AID_ABLATIONS
    int abl_ = 0;               // development builds: 4 = epilogue without the global stores, 8 = without the staging pass
#endif

    __device__ __forceinline__ void init(char* smem_) {
        PP::init(smem_);
        xoff = 4 * HALF + l31 * 128;                            // strip row l31 inside a parity
        xx = hi ^ PP::swz(l31);
    }
    __device__ __forceinline__ void set_tile(const GemmDesc& P, const T* A, const T* B, int m0, int n0) {
        PP::set_tile(P, A, B, m0, n0);
        const int lrow = 8 * (wave & 3) + (lane >> 3);          // piece `wave` (0 .. 3) of the strip: 8 rows x 128 B
        const int c = (lane & 7) ^ PP::swz(lrow);
        xvo = min(256 + lrow, P.m - 1 - m0) * (P.lda * 2) + c * 16;
    }
    __device__ __forceinline__ void zero_acc() {
        PP::zero_acc();
#pragma unroll
        for (int r = 0; r < 16; ++r) accx[r] = 0.f;
    }
    // the four half-tiles live at the same offsets as in PingPong, parities are STGX apart
    __device__ __forceinline__ void xdma_half(const int Q, int t, int kb) {     // Q is a literal at every call site
        const bool IS_A = Q >= 2;
        const int H = Q & 1;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            char* dst = smem + (t & 1) * STGX + (IS_A ? 0 : 2 * HALF) + H * HALF + wave * 2048 + j * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(IS_A ? ra : rb, (__attribute__((address_space(3))) void*)dst, 16,
                                                     IS_A ? avo[H * 2 + j] : bvo[H * 2 + j], (kb + t) * kmul, 0, 0);
        }
    }
    __device__ __forceinline__ void xdma_one(const int Q, const int j, int t, int kb) {     // piece j (0 / 1) of half-tile Q
        const bool IS_A = Q >= 2;
        const int H = Q & 1;
        char* dst = smem + (t & 1) * STGX + (IS_A ? 0 : 2 * HALF) + H * HALF + wave * 2048 + j * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(IS_A ? ra : rb, (__attribute__((address_space(3))) void*)dst, 16,
                                                 IS_A ? avo[H * 2 + j] : bvo[H * 2 + j], (kb + t) * kmul, 0, 0);
    }
    __device__ __forceinline__ void dma_strip(int t, int kb) {  // waves 0 - 3 only
        char* dst = smem + (t & 1) * STGX + 4 * HALF + wave * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)dst, 16, xvo, (kb + t) * kmul, 0, 0);
    }
    __device__ __forceinline__ void read_x(T8 (&fx)[4], const char* st) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fx[ks] = *reinterpret_cast<const T8*>(st + xoff + (((2 * ks) ^ xx) << 4));
    }
    // counted wait of a read slot: waves 0 - 3 carry one more DMA (the strip piece) per K tile in the A group
    template <int N>
    __device__ __forceinline__ void wait_rd(bool xw) {
        if (xw) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N + 1) : "memory");
        else    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
    }

    // The K loop of PingPong::mac_rd with the strip: stream per wave  B(0) A(0) X(0) B(1) | A(1) X(1) B(2) | A(2) X(2) B(3) ...
    //   READ-P0(kt): reads B0 B1 A0 of kt, requests A0 A1 X of kt + 1, retires A1(kt), X(kt)   (newer: B(kt+1) A(kt+1) X(kt+1))
    //   READ-P1(kt): reads A1(kt), X(kt), requests B0 B1 of kt + 2, retires B(kt+1), A0(kt+1)  (newer: A1(kt+1) X(kt+1) B(kt+2))
    // TR: the problem wants C transposed per frame (GemmDesc.trans_rows): the MFMAs are issued the other way round
    // (D rows = m, columns = n), so a lane ends up with four consecutive ROWS of one output column and the staged tile
    // can be written n-major with the same 8-byte stores.
    // ORD (development build -DAID_PPX_ORDERS, GEMM_PP = 4 .. 6; same arithmetic, same results): where a read slot issues its LDS-DMA
    // requests — 0 behind the fragment reads (round 3, the product), 1 in front of them, 2 one request per four reads; 3 = order 0 with
    // s_setprio 1 on the MFMA slots
    template <int WR, bool TR, int ORD>
    __device__ __forceinline__ void mac_x(int kb, int ke) {
        const int nk = ke - kb;
        const bool xw = wave < 4;
        if (ORD == 4) kmul = 0;             // (timing only, results are garbage: the K loop with every operand request served by the L2)
        xdma_half(0, 0, kb); xdma_half(1, 0, kb); xdma_half(2, 0, kb); xdma_half(3, 0, kb);
        if (xw) dma_strip(0, kb);
        if (nk > 1) {
            xdma_half(0, 1, kb); xdma_half(1, 1, kb);
            if (xw) wait_vmcnt<7>(); else wait_vmcnt<6>();      // B(0), A0(0) landed; A1(0) [X(0)] B(1) may fly
        } else {
            if (xw) wait_vmcnt<3>(); else wait_vmcnt<2>();
        }
        PP::slot();
        if (WR == 1) PP::slot();            // the second group runs one barrier behind
        T8 fa[2][4], fb[2][4], fx[4];
        auto body = [&](int kt, auto more1_t, auto more2_t) __attribute__((always_inline)) {
            constexpr bool M1 = decltype(more1_t)::value, M2 = decltype(more2_t)::value;   // tile kt + 1 / kt + 2 exists
            const char* st = smem + (kt & 1) * STGX;
            auto rdb = [&](int j) __attribute__((always_inline)) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) fb[j][ks] = *reinterpret_cast<const T8*>(st + boff[0] + j * HALF + (((2 * ks) ^ bx[0]) << 4));
            };
            auto rda = [&](int e, int h) __attribute__((always_inline)) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) fa[e][ks] = *reinterpret_cast<const T8*>(st + h * HALF + aoff[e] + (((2 * ks) ^ ax[e]) << 4));
            };
            auto pinb = []() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };
            if (ORD == 1 && M1) { xdma_half(2, kt + 1, kb); xdma_half(3, kt + 1, kb); if (xw) dma_strip(kt + 1, kb); pinb(); }
            if (ORD == 2 && M1) {
                xdma_one(2, 0, kt + 1, kb); pinb(); rdb(0); pinb(); xdma_one(2, 1, kt + 1, kb); pinb(); rdb(1); pinb();
                xdma_one(3, 0, kt + 1, kb); pinb(); rda(0, 0); pinb(); xdma_one(3, 1, kt + 1, kb); pinb(); rda(1, 0); pinb();
                if (xw) dma_strip(kt + 1, kb);
            } else {
                this->read_b(fb, st);
                this->read_a(fa, st, 0);
            }
            if (M1) {
                if (ORD == 0 || ORD >= 3) { xdma_half(2, kt + 1, kb); xdma_half(3, kt + 1, kb); if (xw) dma_strip(kt + 1, kb); }
                wait_rd<8>(xw);
            } else {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            }
            PP::slot();
            if (ORD == 3) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ks = i >> 2, in = (i >> 1) & 1, e = i & 1;
                acc[in][e] = TR ? mfma32(fa[e][ks], fb[in][ks], acc[in][e]) : mfma32(fb[in][ks], fa[e][ks], acc[in][e]);
            }
            if (ORD == 3) __builtin_amdgcn_s_setprio(0);
            PP::slot();
            if (ORD == 1 && M2) { xdma_half(0, kt + 2, kb); xdma_half(1, kt + 2, kb); pinb(); }
            if (ORD == 2 && M2) {
                xdma_one(0, 0, kt + 2, kb); pinb(); rda(0, 1); pinb(); xdma_one(0, 1, kt + 2, kb); pinb(); rda(1, 1); pinb();
                xdma_one(1, 0, kt + 2, kb); pinb(); read_x(fx, st); pinb(); xdma_one(1, 1, kt + 2, kb);
            } else {
                this->read_a(fa, st, 1);
                read_x(fx, st);
            }
            if (M2) {
                if (ORD == 0 || ORD >= 3) { xdma_half(0, kt + 2, kb); xdma_half(1, kt + 2, kb); }
                wait_rd<6>(xw);
            } else if (M1) {
                wait_rd<2>(xw);
            } else {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            }
            PP::slot();
            if (ORD == 3) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ks = i >> 2, in = (i >> 1) & 1, e = i & 1;
                acc[in][2 + e] = TR ? mfma32(fa[e][ks], fb[in][ks], acc[in][2 + e]) : mfma32(fb[in][ks], fa[e][ks], acc[in][2 + e]);
                if ((i & 3) == 3)                                               // the strip's block: this wave's own B fragment
                    accx = TR ? mfma32(fx[ks], fb[WR][ks], accx) : mfma32(fb[WR][ks], fx[ks], accx);
            }
            if (ORD == 3) __builtin_amdgcn_s_setprio(0);
            PP::slot();
        };
        const std::true_type Y{};
        const std::false_type N{};
        int kt = 0;
        for (; kt + 2 < nk; ++kt) body(kt, Y, Y);
        if (kt + 1 < nk) { body(kt, Y, N); ++kt; }
        if (kt < nk) body(kt, N, N);
        if (WR == 0) PP::slot();
        wait_vmcnt<0>();
        __syncthreads();                    // everyone is done reading the ring
    }
    // Whole tile: four copies of (K loop + epilogue) — fb[WR] must be a compile-time register choice and the operand order
    // a compile-time choice; the epilogue sits INSIDE each copy so that no accumulator crosses a control-flow merge (with a
    // common epilogue behind the four loops hipcc spilled 80 accumulator registers at the join).
    template <int ORD>
    __device__ __forceinline__ void run_tile(const GemmDesc& P, T* C, int batch, int m0, int n0
#ifdef AID_ABLATIONS
                                             , int abl = 0
#endif
                                             ) {
        const int nk = P.k / 64;
        const T* R = P.residual ? reinterpret_cast<const T*>(P.residual) + (int64_t)batch * P.stride_c : nullptr;
        const float* st = P.ln_stats ? P.ln_stats + 2 * (int64_t)batch * P.stride_stats : nullptr;
        if (P.trans_rows) {
            if (wr == 0) { mac_x<0, true, ORD>(0, nk); store_tile_t(P, reinterpret_cast<T*>(P.c), m0, n0, P.ln_stats); }
            else         { mac_x<1, true, ORD>(0, nk); store_tile_t(P, reinterpret_cast<T*>(P.c), m0, n0, P.ln_stats); }
        } else {
#ifdef AID_ABLATIONS
            // (timing ablations of development builds — 1 = no epilogue, 2 = no K loop; one call site per instantiation: a second
            //  `mac_x<1, false>` call elsewhere fails to instantiate in hipcc's host pass)
            abl_ = abl;
            if (wr == 0) { if (!(abl & 2)) mac_x<0, false, ORD>(0, nk); if (!(abl & 1)) store_tile(P, C, m0, n0, R, st); }
            else         { if (!(abl & 2)) mac_x<1, false, ORD>(0, nk); if (!(abl & 1)) store_tile(P, C, m0, n0, R, st); }
            if (abl & 1) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
                asm volatile("" ::"v"(accx));
            }
#else
            if (wr == 0) { mac_x<0, false, ORD>(0, nk); store_tile(P, C, m0, n0, R, st); }
            else         { mac_x<1, false, ORD>(0, nk); store_tile(P, C, m0, n0, R, st); }
#endif
        }
    }

    // Transposed epilogue (GemmDesc.trans_rows = rows per frame): element (m, n) goes to
    //   C + (m / trans_rows) * stride_c + n * ldc + m % trans_rows      — V^T[frame][channel][key] from the FLAT product
    // E W^T, so the value projection has the tile count of the q / k projections (SDXL: 250 tiles of 288 rows, not 280 of
    // 256 channel rows) and one launch of all three is 750 tiles = 2.93 CU rounds.  Accumulators hold D rows = m.
    __device__ __forceinline__ void store_tile_t(const GemmDesc& P, T* C, int m0, int n0, const float* stats) {
        constexpr int NTHR = 512, BN = 256, CLDT = BMX + 8;
        static_assert((size_t)BN * CLDT * 2 + LNS <= SMEMX, "transposed C tile fits");
        T* Cs = reinterpret_cast<T*>(smem);        // [BN][CLDT]: n-major
        mfma_fence(acc);
        asm volatile("" : "+v"(accx));
        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
        asm volatile("" : "+v"(accx));
        const T* __restrict__ bias = reinterpret_cast<const T*>(P.bias);
        const int side = stats ? P.ln_side : 0;    // 1 only (the activation is A): statistics per m, weight constants per n
        float* const lnr = reinterpret_cast<float*>(smem + (size_t)BN * CLDT * 2);    // [BMX][2] row statistics
        float* const lnc = lnr + 2 * BMX;                                             // [BN][2]  (colsum, shift)
        if (side) {
            for (int i = tid; i < BMX + BN; i += NTHR) {
                const bool isrow = i < BMX;
                const int gi = isrow ? min(m0 + i, P.m - 1) : min(n0 + i - BMX, P.n - 1);
                lnr[2 * i] = isrow ? stats[2 * gi] : P.ln_colsum[gi];
                lnr[2 * i + 1] = isrow ? stats[2 * gi + 1] : P.ln_shift[gi];
            }
            __syncthreads();
        }
        // one 32 x 32 block: column cb + l31, tile rows rb + 8 gq + 4 hi + e
        auto stage = [&](const f32x16& a, int rb, int cb) __attribute__((always_inline)) {
            const int col = cb + l31;
            const float bv = (bias && n0 + col < P.n) ? (float)bias[n0 + col] : 0.f;
            float cs = 0.f, sh = 0.f;
            if (side) { cs = lnc[2 * col]; sh = lnc[2 * col + 1]; }
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int ml = rb + gq * 8 + hi * 4;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = a[gq * 4 + e];
                if (side) {
                    const f32x4 p0 = *reinterpret_cast<const f32x4*>(lnr + 2 * ml);          // (mean, rstd) of rows ml, ml + 1
                    const f32x4 p1 = *reinterpret_cast<const f32x4*>(lnr + 2 * ml + 4);
                    const float mu[4] = {p0[0], p0[2], p1[0], p1[2]}, rs[4] = {p0[1], p0[3], p1[1], p1[3]};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = fmaf(-mu[e], cs, v[e]);
                        asm volatile("" : "+v"(t));
                        v[e] = fmaf(rs[e], t, sh);
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], P.scale, bv);
                *reinterpret_cast<T4*>(Cs + col * CLDT + ml) = cvt4<T>(v);
            }
        };
#pragma unroll
        for (int in = 0; in < 2; ++in)
#pragma unroll
            for (int im = 0; im < 4; ++im) stage(acc[in][im], wm + im * 32, wn + in * 32);
        stage(accx, 256, wn + wr * 32);
        __syncthreads();
        // Fast path: frames at least as long as the tile (the tile touches at most two frames: one compare instead of a division
        // per chunk), all 18 LDS reads up front, then the stores.  Thread -> (staged row `col` = id / 36, chunk of 8 keys).
        if (P.trans_rows >= BMX && n0 + BN <= P.n && ((reinterpret_cast<uintptr_t>(C) & 15) == 0)) {
            const int f0 = m0 / P.trans_rows;                       // scalar
            const int next = (f0 + 1) * P.trans_rows - m0;          // first tile row of the next frame
            constexpr int NPT = (BN * (BMX / 8)) / NTHR;            // 18
            T8 v[NPT];
            T* dst[NPT];
#pragma unroll
            for (int it = 0; it < NPT; ++it) {
                const int id = tid + it * NTHR;
                const int col = id / (BMX / 8), chm = (id % (BMX / 8)) * 8;
                v[it] = *reinterpret_cast<const T8*>(Cs + col * CLDT + chm);
                const int f = f0 + (chm >= next ? 1 : 0);
                dst[it] = (m0 + chm < P.m) ? C + (int64_t)f * P.stride_c + (int64_t)(n0 + col) * P.ldc + (m0 + chm - f * P.trans_rows)
                                           : nullptr;
            }
#pragma unroll
            for (int it = 0; it < NPT; ++it)
                if (dst[it]) *reinterpret_cast<T8*>(dst[it]) = v[it];
            return;
        }
        constexpr int CPRW = BMX / 8;              // 16-B chunks (8 rows m) per staged n-row
#pragma unroll
        for (int it = 0; it < (BN * CPRW) / NTHR; ++it) {
            const int id = tid + it * NTHR;
            const int col = id / CPRW, ch = (id % CPRW) * 8;
            const int m = m0 + ch, n = n0 + col;
            if (m >= P.m || n >= P.n) continue;                  // trans_rows, m0 and P.m are multiples of 8: whole chunks
            const int f = m / P.trans_rows, key = m - f * P.trans_rows;
            *reinterpret_cast<T8*>(C + (int64_t)f * P.stride_c + (int64_t)n * P.ldc + key) =
                *reinterpret_cast<const T8*>(Cs + col * CLDT + ch);
        }
    }

    template <int SIDE>
    __device__ __forceinline__ void stage_x_fast(const GemmDesc& P, T* Cs, int n0, const float* lnr, const float* lnc) {
        const T* bias = reinterpret_cast<const T*>(P.bias);
        const float scale = P.scale;
        f32x4 bvx[4];
#pragma unroll
        for (int in = 0; in < 2; ++in) {
            f32x4 bv[4];
            bias_block<T>(bv, bias, n0 + wn + in * 32 + hi * 4);
#pragma unroll
            for (int im = 0; im < 4; ++im)
                stage_block_fast<T, SIDE>(Cs, CLD, acc[in][im], wm + im * 32 + l31, wn + in * 32 + hi * 4, scale, bv, lnr, lnc);
        }
        bias_block<T>(bvx, bias, n0 + wn + wr * 32 + hi * 4);                    // the strip's block: columns of B fragment wr
        stage_block_fast<T, SIDE>(Cs, CLD, accx, 256 + l31, wn + wr * 32 + hi * 4, scale, bvx, lnr, lnc);
    }

    // Epilogue of Engine::store_tile for nine blocks per wave and 288 tile rows.
    __device__ __forceinline__ void store_tile(const GemmDesc& P, T* C, int m0, int n0, const T* R = nullptr,
                                               const float* stats = nullptr) {
        constexpr int NTHR = 512, BN = 256;
        T* Cs = reinterpret_cast<T*>(smem);        // [BMX][CLD]
        mfma_fence(acc);
        asm volatile("" : "+v"(accx));
        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
        asm volatile("" : "+v"(accx));
        const T* __restrict__ bias = reinterpret_cast<const T*>(P.bias);
        const bool bias_vec = (reinterpret_cast<uintptr_t>(bias) & 7) == 0;
        const int side = stats ? P.ln_side : 0;
        float* const lnr = reinterpret_cast<float*>(smem + (size_t)BMX * CLD * 2);    // [BMX][2]
        float* const lnc = lnr + 2 * BMX;                                             // [BN][2]
        if (epilogue_fast_ok<T>(P, C, R, n0, BN)
#ifdef AID_ABLATIONS
            && !(abl_ & 12)
#endif
        ) {                                                     // straight-line epilogue (see stage_block_fast)
            if (side) {
                for (int i = tid; i < BMX + BN; i += NTHR) {
                    const bool isrow = i < BMX;
                    const int gi = isrow ? min(m0 + i, P.m - 1) : n0 + i - BMX;
                    const bool st = isrow == (side == 1);
                    lnr[2 * i] = st ? stats[2 * gi] : P.ln_colsum[gi];
                    lnr[2 * i + 1] = st ? stats[2 * gi + 1] : P.ln_shift[gi];
                }
                __syncthreads();
            }
            if (side == 0)      stage_x_fast<0>(P, Cs, n0, lnr, lnc);
            else if (side == 1) stage_x_fast<1>(P, Cs, n0, lnr, lnc);
            else                stage_x_fast<2>(P, Cs, n0, lnr, lnc);
            __syncthreads();
            store_rows_fast<T, NTHR, BMX, BN, CLD>(Cs, C, R, m0, n0, P.ldc, min(BMX, P.m - m0), tid);
            return;
        }
        if (side) {
            for (int i = tid; i < BMX + BN; i += NTHR) {
                const bool isrow = i < BMX;
                const int gi = isrow ? min(m0 + i, P.m - 1) : min(n0 + i - BMX, P.n - 1);
                float q0, q1;
                if (isrow == (side == 1)) { q0 = stats[2 * gi]; q1 = stats[2 * gi + 1]; }
                else                      { q0 = P.ln_colsum[gi]; q1 = P.ln_shift[gi]; }
                lnr[2 * i] = q0;
                lnr[2 * i + 1] = q1;
            }
            __syncthreads();
        }
        // one 32 x 32 accumulator block: tile rows rb + l31, columns cb + 8 gq + 4 hi + e
        auto stage = [&](const f32x16& a, int rb, int cb) __attribute__((always_inline)) {
            const int row = rb + l31;
            float lr0 = 0.f, lr1 = 0.f;
            if (side) { lr0 = lnr[2 * row]; lr1 = lnr[2 * row + 1]; }
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int nl = cb + gq * 8 + hi * 4;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = a[gq * 4 + e];
                if (side) {
                    const f32x4 p0 = *reinterpret_cast<const f32x4*>(lnc + 2 * nl);
                    const f32x4 p1 = *reinterpret_cast<const f32x4*>(lnc + 2 * nl + 4);
                    const float lc0[4] = {p0[0], p0[2], p1[0], p1[2]}, lc1[4] = {p0[1], p0[3], p1[1], p1[3]};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {       // scalar FMAs on purpose (Engine::store_tile)
                        float t = side == 1 ? fmaf(-lr0, lc0[e], v[e]) : fmaf(-lc0[e], lr0, v[e]);
                        asm volatile("" : "+v"(t));
                        v[e] = side == 1 ? fmaf(lr1, t, lc1[e]) : fmaf(lc1[e], t, lr1);
                    }
                }
                f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                if (bias) {
                    if (bias_vec && n0 + nl + 4 <= P.n) {
                        bv = up4<T>(*reinterpret_cast<const T4*>(bias + n0 + nl));
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (n0 + nl + e < P.n) bv[e] = (float)bias[n0 + nl + e];
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], P.scale, bv[e]);
                *reinterpret_cast<T4*>(Cs + row * CLD + nl) = cvt4<T>(v);
            }
        };
#ifdef AID_ABLATIONS
        if (!(abl_ & 8))
#endif
        {
#pragma unroll
        for (int in = 0; in < 2; ++in)
#pragma unroll
            for (int im = 0; im < 4; ++im) stage(acc[in][im], wm + im * 32, wn + in * 32);
        stage(accx, 256, wn + wr * 32);
        }
#ifdef AID_ABLATIONS
        if (abl_ & 8) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) asm volatile("" ::"v"(acc[i][jj]));
            asm volatile("" ::"v"(accx));
        }
        if (abl_ & 4) return;
#endif
        __syncthreads();
        const bool vec_ok = (P.ldc % 8 == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0);
        constexpr int CPRW = BN / 8;               // 16-B chunks per C row
#pragma unroll
        for (int it = 0; it < (BMX * CPRW) / NTHR; ++it) {
            const int id = tid + it * NTHR;
            const int row = id / CPRW, ch = (id % CPRW) * 8;
            const int m = m0 + row, n = n0 + ch;
            if (m >= P.m || n >= P.n) continue;
            T8 v = *reinterpret_cast<const T8*>(Cs + row * CLD + ch);
            T* dst = C + (int64_t)m * P.ldc + n;
            const T* res = R ? R + (int64_t)m * P.ldc + n : nullptr;
            if (vec_ok && n + 8 <= P.n) {
                if (res) {
                    const bool rvec = (reinterpret_cast<uintptr_t>(R) & 15) == 0;
                    T8 r;
                    if (rvec) {
                        r = *reinterpret_cast<const T8*>(res);
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) r[e] = res[e];
                    }
                    const f32x8 a = up8<T>(v), b = up8<T>(r);
                    v = cvt8<T>(a + b);
                }
                *reinterpret_cast<T8*>(dst) = v;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (n + e < P.n) dst[e] = res ? (T)((float)v[e] + (float)res[e]) : v[e];
            }
        }
    }
};
static_assert((288 * 32) % 512 == 0 && (256 * 36) % 512 == 0, "C rows divide over the threads");
static_assert(Engine<bf16, 128, 128, 64, 4, 2, 4>::SMEM <= PingPongX<bf16>::SMEMX, "side tiles use the big tile's LDS");

template <typename T, int ORD = 0>
__global__ __launch_bounds__(512) void aid_gemm_nt_ppx_kernel(const GemmGroup g, const GemmSide sd
#ifdef AID_ABLATIONS
                                                              , const int abl      // development builds only: timing ablations
#endif
                                                              ) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    if ((int)blockIdx.x < sd.pad_tiles) {                          // side problems first (see aid_gemm_nt_pp_kernel)
        const int u = blockIdx.x;
        if (u >= sd.tiles) return;
        int pi = 0;
#pragma unroll
        for (int i = 1; i < 4; ++i)
            if (i < sd.n && u >= sd.tile_start[i]) pi = i;
        const GemmDesc& P = sd.p[pi];
        int rem = u - sd.tile_start[pi];
        const int tiles_n = (P.n + 127) / 128, per_batch = ((P.m + 127) / 128) * tiles_n;
        const int batch = rem / per_batch;
        rem -= batch * per_batch;
        const int m0 = (rem / tiles_n) * 128, n0 = (rem % tiles_n) * 128;
        const T* A = reinterpret_cast<const T*>(P.a) + (int64_t)batch * P.stride_a;
        const T* B = reinterpret_cast<const T*>(P.b) + (int64_t)batch * P.stride_b;
        T* C = reinterpret_cast<T*>(P.c) + (int64_t)batch * P.stride_c;
        Engine<T, 128, 128, 64, 4, 2, 4> e;
        e.init(smem_raw);
        e.set_tile(P, A, B, m0, n0);
        e.zero_acc();
        e.mac(0, P.k / 64);
        e.store_tile(P, C, m0, n0, P.residual ? reinterpret_cast<const T*>(P.residual) + (int64_t)batch * P.stride_c : nullptr,
                     P.ln_stats ? P.ln_stats + 2 * (int64_t)batch * P.stride_stats : nullptr);
        return;
    }
    const int b = blockIdx.x - sd.pad_tiles;
    const TileCoord tc = locate_pos<288, 256>(g, xcd_remap(b, (int)gridDim.x - sd.pad_tiles));
    const GemmDesc& P = g.p[tc.p];
    const T* A = reinterpret_cast<const T*>(P.a) + (int64_t)tc.batch * P.stride_a;
    const T* B = reinterpret_cast<const T*>(P.b) + (int64_t)tc.batch * P.stride_b;
    T* C = reinterpret_cast<T*>(P.c) + (int64_t)tc.batch * P.stride_c;
    PingPongX<T> e;
    e.init(smem_raw);
    e.set_tile(P, A, B, tc.m0, tc.n0);
    e.zero_acc();
#ifdef AID_ABLATIONS
    e.template run_tile<ORD>(P, C, tc.batch, tc.m0, tc.n0, abl);
#else
    e.template run_tile<ORD>(P, C, tc.batch, tc.m0, tc.n0);
#endif
}

template <typename T, int PPV>
__global__ __launch_bounds__(512) void aid_gemm_nt_pp_kernel(const GemmGroup g, const int n_big, const GemmSide sd) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    if ((int)blockIdx.x < sd.pad_tiles) {                          // side problems first: their long K loops start at once
        const int u = blockIdx.x;
        if (u >= sd.tiles) return;
        int pi = 0;
#pragma unroll
        for (int i = 1; i < 4; ++i)
            if (i < sd.n && u >= sd.tile_start[i]) pi = i;
        const GemmDesc& P = sd.p[pi];
        int rem = u - sd.tile_start[pi];
        const int tiles_n = (P.n + 127) / 128, per_batch = ((P.m + 127) / 128) * tiles_n;
        const int batch = rem / per_batch;
        rem -= batch * per_batch;
        const int m0 = (rem / tiles_n) * 128, n0 = (rem % tiles_n) * 128;
        const T* A = reinterpret_cast<const T*>(P.a) + (int64_t)batch * P.stride_a;
        const T* B = reinterpret_cast<const T*>(P.b) + (int64_t)batch * P.stride_b;
        T* C = reinterpret_cast<T*>(P.c) + (int64_t)batch * P.stride_c;
        Engine<T, 128, 128, 64, PP_SMALL_NS, 2, 4> e;
        e.init(smem_raw);
        e.set_tile(P, A, B, m0, n0);
        e.zero_acc();
        e.mac(0, P.k / 64);
        e.store_tile(P, C, m0, n0, P.residual ? reinterpret_cast<const T*>(P.residual) + (int64_t)batch * P.stride_c : nullptr,
                     P.ln_stats ? P.ln_stats + 2 * (int64_t)batch * P.stride_stats : nullptr);
        return;
    }
    const int b = blockIdx.x - sd.pad_tiles;
    if (b < n_big) {
        const TileCoord tc = locate_pos<256, 256>(g, xcd_remap(b, n_big));
        const GemmDesc& P = g.p[tc.p];
        const T* A = reinterpret_cast<const T*>(P.a) + (int64_t)tc.batch * P.stride_a;
        const T* B = reinterpret_cast<const T*>(P.b) + (int64_t)tc.batch * P.stride_b;
        T* C = reinterpret_cast<T*>(P.c) + (int64_t)tc.batch * P.stride_c;
        PingPong<T> e;
        e.init(smem_raw);
        e.set_tile(P, A, B, tc.m0, tc.n0);
        e.zero_acc();
        if (PPV == 1)      e.template mac_rd<false>(0, P.k / 64);
        else if (PPV == 2) e.template mac_rd<true>(0, P.k / 64);
        else if (PPV >= 4) e.template mac_rd<false, PPV - 4>(0, P.k / 64);
        else               e.mac(0, P.k / 64);
        e.store_tile(P, C, tc.m0, tc.n0, P.residual ? reinterpret_cast<const T*>(P.residual) + (int64_t)tc.batch * P.stride_c : nullptr,
                     P.ln_stats ? P.ln_stats + 2 * (int64_t)tc.batch * P.stride_stats : nullptr);
    } else {
        const int u = b - n_big;                                   // n_big is a multiple of 8: u % 8 is still the XCD
        const int n_rest = ((int)gridDim.x - sd.pad_tiles - n_big) >> 2;
        const int quad = u / n_rest, t = u - quad * n_rest;        // quadrant-major: equal quadrants are neighbours
        TileCoord tc = locate_pos<256, 256>(g, n_big + xcd_remap(t, n_rest));
        const GemmDesc& P = g.p[tc.p];
        tc.m0 += (quad >> 1) * 128;
        tc.n0 += (quad & 1) * 128;
        if (tc.m0 >= P.m || tc.n0 >= P.n) return;
        const T* A = reinterpret_cast<const T*>(P.a) + (int64_t)tc.batch * P.stride_a;
        const T* B = reinterpret_cast<const T*>(P.b) + (int64_t)tc.batch * P.stride_b;
        T* C = reinterpret_cast<T*>(P.c) + (int64_t)tc.batch * P.stride_c;
        Engine<T, 128, 128, 64, PP_SMALL_NS, 2, 4> e;
        e.init(smem_raw);
        e.set_tile(P, A, B, tc.m0, tc.n0);
        e.zero_acc();
        e.mac(0, P.k / 64);
        e.store_tile(P, C, tc.m0, tc.n0, P.residual ? reinterpret_cast<const T*>(P.residual) + (int64_t)tc.batch * P.stride_c : nullptr,
                     P.ln_stats ? P.ln_stats + 2 * (int64_t)tc.batch * P.stride_stats : nullptr);
    }
}

// ---- one output tile per workgroup ------------------------------------------------------------------
template <typename T, int BM, int BN, int BK, int NS, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64) void aid_gemm_nt_pipe_kernel(const GemmGroup g) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const TileCoord tc = locate_tile<BM, BN>(g, blockIdx.x, gridDim.x);
    const GemmDesc& P = g.p[tc.p];
    const T* A = reinterpret_cast<const T*>(P.a) + (int64_t)tc.batch * P.stride_a;
    const T* B = reinterpret_cast<const T*>(P.b) + (int64_t)tc.batch * P.stride_b;
    T* C = reinterpret_cast<T*>(P.c) + (int64_t)tc.batch * P.stride_c;
    Engine<T, BM, BN, BK, NS, WM, WN> e;
    e.init(smem_raw);
    e.set_tile(P, A, B, tc.m0, tc.n0);
    e.zero_acc();
    e.mac(0, P.k / BK);
    e.store_tile(P, C, tc.m0, tc.n0, P.residual ? reinterpret_cast<const T*>(P.residual) + (int64_t)tc.batch * P.stride_c : nullptr,
                     P.ln_stats ? P.ln_stats + 2 * (int64_t)tc.batch * P.stride_stats : nullptr);
}

// ------------------------------------------------------------------------------------------------
static int plan_tiles(GemmGroup& g, int bm, int bn) {
    int tiles = 0;
    for (int i = 0; i < g.n_problems; ++i) {
        g.tile_start[i] = tiles;
        tiles += ((g.p[i].m + bm - 1) / bm) * ((g.p[i].n + bn - 1) / bn) * g.p[i].batch;
    }
    for (int i = g.n_problems; i <= AID_GEMM_MAX_PROBLEMS; ++i) g.tile_start[i] = tiles;
    return tiles;
}

template <typename K>
static hipError_t launch_with_smem(K kernel, size_t smem, PerDevice<bool>* attr_set, const GemmGroup& g, int total_tiles,
                                   hipStream_t stream, int threads) {
    if (total_tiles <= 0) return hipSuccess;
    bool* done = attr_set->slot();
    if (!done) return hipErrorInvalidDevice;
    if (!*done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        *done = true;
    }
    hipLaunchKernelGGL(kernel, dim3(total_tiles), dim3(threads), smem, stream, g);
    return hipGetLastError();
}

template <typename T, int BM, int BN, int BK, int NS, int WM, int WN>
static hipError_t launch_pipe(GemmGroup& g, hipStream_t stream) {
    static PerDevice<bool> attr_set;
    return launch_with_smem(aid_gemm_nt_pipe_kernel<T, BM, BN, BK, NS, WM, WN>, Engine<T, BM, BN, BK, NS, WM, WN>::SMEM,
                            &attr_set, g, plan_tiles(g, BM, BN), stream, WM * WN * 64);
}

static PerDevice<int> g_num_cu;

static int num_cu() {                       // CU count of the CURRENT device (cached per device)
    int* n = g_num_cu.slot();
    if (!n) return 0;
    if (*n == 0) {
        int dev = 0;
        hipDeviceProp_t pr;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return 0;
        *n = pr.multiProcessorCount;
    }
    return *n;
}

// Plan of the ping-pong path: n_big 256 x 256 tiles (whole CU rounds) + the ragged rest as 128 x 128 tiles.
struct PpPlan {
    int tiles, n_big, n_small;
    double rounds;                          // cost in units of one big-tile round
};
static PpPlan plan_pp(GemmGroup& g, int ncu, int nk) {
    PpPlan pl;
    pl.tiles = plan_tiles(g, 256, 256);
    int full = pl.tiles / ncu, rest = pl.tiles % ncu;
    pl.n_big = pl.tiles;
    pl.n_small = 0;
    pl.rounds = full + (rest ? 1 : 0);
    if (full > 0 && rest > 0 && rest <= ncu / 2 && ncu % 8 == 0) {   // a last round at <= 50 % occupancy: cut it up
        pl.n_big = full * ncu;
        pl.n_small = 4 * rest;
        pl.rounds = full + ((4 * rest + ncu - 1) / ncu) * (2.0 + 0.6 * nk) / (6.0 + 1.62 * nk);
    }
    return pl;
}

template <typename T, int PPV>
static hipError_t launch_pp_v(GemmGroup& g, hipStream_t stream, const PpPlan& pl, const GemmSide& sd) {
    static PerDevice<bool> attr_set;
    if (pl.tiles <= 0) return hipSuccess;
    if (plan_tiles(g, 256, 256) != pl.tiles) return hipErrorInvalidValue;      // g.tile_start must be in 256 x 256 units
    bool* done = attr_set.slot();
    if (!done) return hipErrorInvalidDevice;
    if (!*done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(aid_gemm_nt_pp_kernel<T, PPV>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)PingPong<T>::SMEM);
        if (e != hipSuccess) return e;
        *done = true;
    }
    hipLaunchKernelGGL((aid_gemm_nt_pp_kernel<T, PPV>), dim3(sd.pad_tiles + pl.n_big + pl.n_small), dim3(512), PingPong<T>::SMEM,
                       stream, g, pl.n_big, sd);
    return hipGetLastError();
}
template <typename T>
static hipError_t launch_pp(GemmGroup& g, hipStream_t stream, const PpPlan& pl, const GemmSide& sd) {
    switch (tune(TUNE_GEMM_PP)) {
        case 1:  return launch_pp_v<T, 1>(g, stream, pl, sd);
        case 0:  return launch_pp_v<T, 0>(g, stream, pl, sd);
        case 2:  return launch_pp_v<T, 2>(g, stream, pl, sd);
#ifdef AID_ABLATIONS
        case 5:  return launch_pp_v<T, 5>(g, stream, pl, sd);      // no DMAs
        case 6:  return launch_pp_v<T, 6>(g, stream, pl, sd);      // no fragment reads
        case 7:  return launch_pp_v<T, 7>(g, stream, pl, sd);      // neither: MFMAs + barriers
#endif
        default: return launch_pp_v<T, 1>(g, stream, pl, sd);
    }
}


template <typename T>
static hipError_t launch_ppx(GemmGroup& g, hipStream_t stream, const GemmSide& sd) {
    static PerDevice<bool> attr_set;
    const int tiles = plan_tiles(g, 288, 256);
    if (tiles <= 0) return hipSuccess;
    bool* done = attr_set.slot();
    if (!done) return hipErrorInvalidDevice;
    if (!*done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(aid_gemm_nt_ppx_kernel<T>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)PingPongX<T>::SMEMX);
        if (e != hipSuccess) return e;
        *done = true;
    }
#ifdef AID_ABLATIONS
    const int abl = tune(TUNE_GEMM_PP) >= 8 ? tune(TUNE_GEMM_PP) - 8 : 0;
    hipLaunchKernelGGL(aid_gemm_nt_ppx_kernel<T>, dim3(sd.pad_tiles + tiles), dim3(512), PingPongX<T>::SMEMX, stream, g, sd, abl);
#else
#ifdef AID_PPX_ORDERS
    // development build (tools/dev/Makefile, libaid_ppxord.so): the read-slot orders of mac_x side by side, GEMM_PP = 4 / 5 / 6
    const int ord = tune(TUNE_GEMM_PP) - 3;
    auto go = [&](auto kern) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PingPongX<T>::SMEMX);
        hipLaunchKernelGGL(kern, dim3(sd.pad_tiles + tiles), dim3(512), PingPongX<T>::SMEMX, stream, g, sd);
    };
    if (ord == 1)      go(aid_gemm_nt_ppx_kernel<T, 1>);
    else if (ord == 2) go(aid_gemm_nt_ppx_kernel<T, 2>);
    else if (ord == 3) go(aid_gemm_nt_ppx_kernel<T, 3>);
    else if (ord == 4) go(aid_gemm_nt_ppx_kernel<T, 4>);      // timing ablation: garbage results
    else               go(aid_gemm_nt_ppx_kernel<T, 0>);
#else
    hipLaunchKernelGGL(aid_gemm_nt_ppx_kernel<T>, dim3(sd.pad_tiles + tiles), dim3(512), PingPongX<T>::SMEMX, stream, g, sd);
#endif
#endif
    return hipGetLastError();
}

// 288-row tiles pay when they turn a ragged last round into whole rounds: cost in rounds of a 256 x 256 tile, a 288-row
// tile counted 9 / 8.  GEMM_TRI: 0 = never, 1 = whenever the shape allows (development knob; the name is historical).
static bool prefer_ppx(GemmGroup& g, int ncu, const PpPlan& pl256) {
    const int knob = tune(TUNE_GEMM_TRI);
    if (knob == 0) return false;
    GemmGroup t = g;
    const int tiles = plan_tiles(t, 288, 256);
    const double r288 = 1.125 * (double)((tiles + ncu - 1) / ncu);
    if (knob == 1) return true;
    return r288 < 0.97 * pl256.rounds;
}

// Two engines serve the k % 64 == 0 shapes; which one a launch gets is decided by a two-line cost model fitted to
// the ten projection launches of the SD1.5 / SDXL stacks (tools/kbench_proj.py, profiles/r01_gemm_variants.txt; it
// picks the measured winner on all ten):
//   lock-step 128 x 128 (2 workgroups / CU):  3 + ceil_half(tiles128 / (2 CUs)) * (4.1 + 1.09 nk)   us
//   ping-pong 256 x 256 (1 workgroup / CU):   5 + rounds * (6 + 1.62 nk)                            us
// The big tiles win on long K loops and many tiles (SDXL C = 1280: 170 -> 145 us, 69 -> 58 us); the small ones on
// short K loops (SD1.5 C = 320), on launches of less than half a round, and whenever the K loops of a group differ
// (the text-context projections of cross-attention: their tiles are mostly padding at 256 x 256).
// A problem with trans_rows (C transposed per frame, the flat value projection E Wv^T -> V^T) runs as such only on the
// 288-row engine.  Everywhere else it is rewritten into the equivalent batched product with swapped operands,
// V^T[f] = Wv E_f^T  (m' = n channels, n' = trans_rows keys, one batch entry per frame) — what the callers issued before.
static bool has_trans(const GemmGroup& g) {
    for (int i = 0; i < g.n_problems; ++i)
        if (g.p[i].trans_rows) return true;
    return false;
}
static void untranspose(GemmGroup& g) {
    for (int i = 0; i < g.n_problems; ++i) {
        GemmDesc& d = g.p[i];
        if (!d.trans_rows) continue;
        const int rows = d.trans_rows, frames = d.m / rows;
        GemmDesc o = d;
        o.a = d.b; o.b = d.a;
        o.m = d.n; o.n = rows;
        o.lda = d.ldb; o.ldb = d.lda;
        o.batch = frames;
        o.stride_a = 0; o.stride_b = (int64_t)rows * d.lda;      // stride_c = frame stride of V^T already
        if (d.ln_stats) { o.ln_side = 2; o.stride_stats = rows; }
        o.trans_rows = 0;
        d = o;
    }
}

template <typename T>
static hipError_t launch_gemm_sel(GemmGroup& g, hipStream_t stream, const char** variant, bool dry, bool* is_ppx, int cu_share);

template <typename T>
static hipError_t launch_gemm(GemmGroup& g, hipStream_t stream, const char** variant, int cu_share) {
    if (has_trans(g)) {
        GemmGroup probe = g;
        bool ppx = false;
        const hipError_t e = launch_gemm_sel<T>(probe, stream, nullptr, true, &ppx, cu_share);
        if (e != hipSuccess) return e;
        if (!ppx) untranspose(g);
    }
    return launch_gemm_sel<T>(g, stream, variant, false, nullptr, cu_share);
}

// dry: pick the engine only (*is_ppx = the 288-row engine would run the main problems), launch nothing
template <typename T>
static hipError_t launch_gemm_sel(GemmGroup& g, hipStream_t stream, const char** variant, bool dry, bool* is_ppx, int cu_share) {
    bool k64 = true;
    for (int i = 0; i < g.n_problems; ++i) k64 = k64 && (g.p[i].k % 64 == 0);
    if (!k64) {
        static PerDevice<bool> s0;
        if (dry) return hipSuccess;
        if (variant) *variant = "edge";
        return launch_with_smem(aid_gemm_nt_kernel<T>, (size_t)2 * (GBM + GBN) * GLD * sizeof(T), &s0, g,
                                plan_tiles(g, GBM, GBN), stream, GTHREADS);
    }
    // development knob (tools/gemm_shapes.py): AID_GEMM_VARIANT=7 / 31 forces the lock-step / ping-pong engine
    const int force = tune(TUNE_GEMM_VARIANT);
    int ncu = num_cu();
    if (ncu <= 0) return hipErrorInvalidDevice;
    // CU_SHARE = n: n launch streams share the device (the two passes of a step on two streams): a launch can count on 1 / n of the
    // CUs, so e.g. 125 tiles of 288 rows are a full round, not half of one (SDXL, two streams: 39.0 -> 37.3 ms/step)
    // (the per-call value — AidGemmProblem.cu_share — wins over the process-wide knob)
    const int share = cu_share > 1 ? cu_share : tune(TUNE_CU_SHARE);
    if (share > 1) ncu = (ncu / share + 7) / 8 * 8;
    // short K, tall shared activation (C = 320 / 640 levels): the row-stationary engine (aid_gemm_rs.hip) — the activation rows stay in
    // registers, the weights stream through LDS; it writes transposed problems itself.  GEMM_RS: 0 never, 1 wherever the shape allows.
    {
        const int rs = tune(TUNE_GEMM_RS);
        // (the size rule looks at the DEVICE's CU count, not at the share the caller's hint leaves: the engine choice between this engine
        //  and the tile engines — whose summation orders differ — must not depend on cu_share; only the split of the slice range does)
        if (rs != 0 && force < 0 && gemm_rs_supported(g, num_cu(), rs == 1)) {
            if (dry) { *is_ppx = true; return hipSuccess; }           // (a transposed problem stays as it is)
            if (variant) *variant = g.p[0].k == 640 ? "rowstat640" : "rowstat320";
            return gemm_rs_launch(g, std::is_same<T, f16>::value ? AID_DTYPE_F16 : AID_DTYPE_BF16, ncu, stream);
        }
    }
    bool pp = false;
    PpPlan pl = {};
    GemmSide sd;
    memset(&sd, 0, sizeof(sd));
    if (g.interleave && g.n_problems > 1 && force != 7) {
        // K loops differ: if a few short problems (<= 15 % of the flops) sit next to main problems of ONE K that the
        // ping-pong engine would win, run the short ones as side tiles of the ping-pong launch
        double fl[AID_GEMM_MAX_PROBLEMS], tot = 0, best = -1;
        int kmain = 0;
        for (int i = 0; i < g.n_problems; ++i) {
            fl[i] = 2.0 * g.p[i].m * g.p[i].n * g.p[i].k * g.p[i].batch;
            tot += fl[i];
            if (fl[i] > best) { best = fl[i]; kmain = g.p[i].k; }
        }
        GemmGroup gm;
        memset(&gm, 0, sizeof(gm));
        double side_fl = 0;
        int ns = 0, st = 0;
        bool ok = true;
        for (int i = 0; i < g.n_problems; ++i) {
            if (g.p[i].k == kmain) {
                gm.p[gm.n_problems++] = g.p[i];
            } else if (ns < 4) {
                sd.p[ns] = g.p[i];
                sd.tile_start[ns] = st;
                st += ((g.p[i].m + 127) / 128) * ((g.p[i].n + 127) / 128) * g.p[i].batch;
                side_fl += fl[i];
                ++ns;
            } else {
                ok = false;
            }
        }
        for (int i = ns; i <= 4; ++i) sd.tile_start[i] = st;
        if (ok && ns > 0 && side_fl <= 0.15 * tot && st <= ncu) {
            const int nk = kmain / 64;
            const int t128 = plan_tiles(gm, 128, 128);
            const PpPlan plm = plan_pp(gm, ncu, nk);
            const double r128 = 0.5 * (double)((2 * t128 + 2 * ncu - 1) / (2 * ncu));
            if (5.0 + plm.rounds * (6.0 + 1.62 * nk) < 0.95 * (3.0 + r128 * (4.1 + 1.09 * nk)) || force == 31) {
                sd.n = ns;
                sd.tiles = st;
                sd.pad_tiles = (st + 7) / 8 * 8;
                if (prefer_ppx(gm, ncu, plm)) {
                    if (dry) {                              // side tiles run on the 128 x 128 engine: no transposed output there
                        *is_ppx = true;
                        for (int i = 0; i < ns; ++i) *is_ppx = *is_ppx && !sd.p[i].trans_rows;
                        return hipSuccess;
                    }
                    if (variant) *variant = "pingpong288+side128";
                    return launch_ppx<T>(gm, stream, sd);
                }
                if (dry) return hipSuccess;
                plan_pp(gm, ncu, nk);                       // back to 256 x 256 tile units
                if (variant) *variant = plm.n_small ? "pingpong256+tail128+side128" : "pingpong256+side128";
                return launch_pp<T>(gm, stream, plm, sd);
            }
        }
        memset(&sd, 0, sizeof(sd));
    }
    if (!g.interleave && g.n_problems > 0) {
        const int nk = g.p[0].k / 64;
        const int t128 = plan_tiles(g, 128, 128);
        pl = plan_pp(g, ncu, nk);                   // last: leaves g.tile_start in 256 x 256 units for launch_pp
        const double r128 = 0.5 * (double)((2 * t128 + 2 * ncu - 1) / (2 * ncu));       // rounds of 2 CUs-fulls, in halves
        const double cost_ls = 3.0 + r128 * (4.1 + 1.09 * nk);
        // (a single round that leaves CUs idle runs its tiles faster — less contention for L2 / HBM, a higher clock: measured 35.5 us at
        //  160 of 256 tiles, 41.3 at 240, 43.4 modelled for a full round, K = 1280; profiles/r05_gemm_shards_ab.txt)
        const int pp_tiles = pl.n_small ? 0 : pl.tiles;
        const double occ = (pp_tiles > 0 && pp_tiles <= ncu) ? (double)pp_tiles / ncu : 1.0;
        const double cost_pp = 5.0 + pl.rounds * (6.0 + 1.62 * nk) * (0.65 + 0.35 * occ);
        pp = cost_pp < 0.95 * cost_ls;
        if (force == 31) pp = true;
    }
    if (force == 7) pp = false;
    if (pp && prefer_ppx(g, ncu, pl)) {
        if (dry) { *is_ppx = true; return hipSuccess; }
        if (variant) *variant = "pingpong288";
        return launch_ppx<T>(g, stream, sd);
    }
    if (dry) return hipSuccess;
    if (pp) plan_pp(g, ncu, g.p[0].k / 64);             // g.tile_start back in 256 x 256 units
    if (variant) *variant = !pp ? "lockstep128" : pl.n_small ? "pingpong256+tail128" : "pingpong256";
    if (pp) return launch_pp<T>(g, stream, pl, sd);
    // 8 waves, 64 x 32 wave tiles.  The ring: a K tile's loads are in flight for ~0.75 us whatever the launch (HBM / L2 latency), so a
    // workgroup that is alone on its CU with ONE tile ahead is latency-bound (0.75 us per K tile against 0.24 us of MFMAs).
    // Measured (profiles/r06_gemm_lockstep_rings.txt): a K tile step costs 0.5 - 0.65 us with one tile ahead and ~0.5 with three ahead —
    // what bounds it is the CU's own fill rate (32 KB per step at 64 B / clk), not the latency; the deeper ring pays 8 - 12 % only
    // where a launch leaves at most one workgroup per CU and has a long K loop (C = 1280 levels of SD1.5, text contexts of SDXL), and
    // costs 10 - 15 % wherever two workgroups share a CU (its 128 KB ring keeps the second one out).  Same bits either way.
    // (32-wide K tiles — four stages in 64 KB, or three in 48 KB for three workgroups per CU — lost everywhere: same file.)
    int ls = tune(TUNE_GEMM_LS);
    if (ls < 0) {
        const int t128 = plan_tiles(g, 128, 128);
        int nk_min = 1 << 30;
        for (int i = 0; i < g.n_problems; ++i) nk_min = g.p[i].k / 64 < nk_min ? g.p[i].k / 64 : nk_min;
        ls = (t128 >= 64 && t128 <= num_cu() && nk_min >= 16) ? 1 : 0;
    }
    if (ls == 1) { if (variant) *variant = "lockstep128x4"; return launch_pipe<T, 128, 128, 64, 4, 2, 4>(g, stream); }
    return launch_pipe<T, 128, 128, 64, 2, 2, 4>(g, stream);       // 2 workgroups / CU
}

hipError_t gemm_group_launch(GemmGroup& g, int dtype, hipStream_t stream, const char** variant, int cu_share) {
    // Longest K loop first: blocks are dispatched in grid order, so the tiles that take longest (the K = 2048
    // text-context projections of a cross-attention layer next to its K = 1280 query projection) start first
    // and finish under the rest instead of forming the tail of the launch (measured: 107 -> 7x us).
    g.interleave = 0;
    for (int i = 1; i < g.n_problems; ++i)
        if (g.p[i].k != g.p[0].k) g.interleave = 1;
    for (int i = 1; i < g.n_problems; ++i)
        for (int j = i; j > 0 && g.p[j].k > g.p[j - 1].k; --j) {
            const GemmDesc t = g.p[j];
            g.p[j] = g.p[j - 1];
            g.p[j - 1] = t;
        }
    return dtype == AID_DTYPE_F16 ? launch_gemm<f16>(g, stream, variant, cu_share) : launch_gemm<bf16>(g, stream, variant, cu_share);
}

}  // namespace aid
