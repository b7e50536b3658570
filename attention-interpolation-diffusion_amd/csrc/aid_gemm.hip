// Grouped NT GEMM for the q / k / V^T / out projections of an AID attention layer.
//   C[b] = A[b] * B[b]^T (+ bias[n]),   A [m,k], B [n,k] (both K-contiguous), fp32 accumulate.
// Replaces attn.to_q / to_k / to_v / to_out[0] (reference interpolation.py:613, 623-624, 666).
//
// gfx950 mapping: BM x BN block tile, waves in a WM x WN grid, each wave owns MB x NB MFMA 32x32x16
// blocks.  The MFMA is issued "transposed" (D rows = n, cols = m) so every lane ends up with 4
// consecutive n of one output row.  The 1-D grid is remapped XCD-aware so the blocks that share an A
// row panel run on one XCD / one L2.
//
//  * aid_gemm_nt_pipe_kernel — main path (every k % BK == 0: all SD1.5 / SDXL projection shapes).
//    Operand tiles go HBM -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave-instruction, no
//    VGPR round trip, no ds_write pass) into an NS-deep ring; NS-1 tiles stay in flight across the
//    (raw) workgroup barrier with a counted s_waitcnt vmcnt.  LDS-DMA writes lane-linear images, so
//    rows are unpadded and the bank-conflict-free layout comes from an XOR swizzle applied to the
//    per-lane SOURCE address and again on the fragment read.  Fragment reads run one k-step ahead of
//    the MFMAs.  The C tile is staged through LDS (re-using the ring) and written as full
//    16-B-per-lane row segments.
//  * aid_gemm_nt_kernel — edge path for ragged k (tests, odd context widths): 128x128 tile,
//    register-staged, fully guarded loads, padded LDS rows.
#include "aid_common.hpp"
#include "aid_kernels.hpp"

#include <stdlib.h>

namespace aid {

struct TileCoord {
    int p, batch, m0, n0;
};

// Block -> (problem, batch, tile).  The hardware places block b on XCD b % 8; xcd_remap gives every XCD a
// contiguous range of positions so neighbouring tiles share that XCD's L2.
//  * problems with equal K loops are simply concatenated (each XCD then mostly streams ONE weight matrix);
//  * when the K loops differ (the K = 2048 text-context projections of a cross-attention layer next to its
//    K = 1280 query projection) every XCD gets a contiguous chunk of EACH problem instead: with a plain
//    concatenation all long-K tiles landed on two XCDs and set the launch time (105 -> 84 us measured).
template <int BM, int BN>
__device__ __forceinline__ TileCoord locate_tile(const GemmGroup& g) {
    constexpr int NX = 8;
    int pos = xcd_remap(blockIdx.x, gridDim.x);          // position in XCD-major order (bijective)
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

    const TileCoord tc = locate_tile<GBM, GBN>(g);
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
                for (int e = 0; e < 4; ++e) v[e] = acc[in][im][gq * 4 + e] * P.scale;
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
// MAC loop over a range of K tiles, the LDS-staged store, and the fp32 partial-tile exchange used by the
// stream-K kernel.  All members are force-inlined; the accumulators live in registers.
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
    static constexpr size_t SMEM = ((size_t)NS * STAGE > (size_t)BM * CLD * 2) ? (size_t)NS * STAGE : (size_t)BM * CLD * 2;
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

    // acc (+bias) -> LDS C tile -> coalesced 16-B row segments.  Ends with all stores drained and a barrier.
    __device__ __forceinline__ void store_tile(const GemmDesc& P, T* C, int m0, int n0) {
        T* Cs = reinterpret_cast<T*>(smem);        // [BM][CLD]
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
                for (int im = 0; im < MB; ++im) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaf(acc[in][im][gq * 4 + e], P.scale, bv[e]);
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
            const T8 v = *reinterpret_cast<const T8*>(Cs + row * CLD + ch);
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

    // fp32 partial tile <-> global scratch, lane-linear (16 B per lane, 1 KiB per wave-instruction); the reader
    // is the same lane of the same wave index in another workgroup, so no layout translation is needed
    __device__ __forceinline__ void write_partial(float* slot) {
        f32x4* dst = reinterpret_cast<f32x4*>(slot) + (size_t)wave * (NB * MB * 4) * 64 + lane;
#pragma unroll
        for (int in = 0; in < NB; ++in)
#pragma unroll
            for (int im = 0; im < MB; ++im)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[in][im][gq * 4 + e];
                    dst[((in * MB + im) * 4 + gq) * 64] = v;
                }
    }
    __device__ __forceinline__ void add_partial(const float* slot) {
        const f32x4* src = reinterpret_cast<const f32x4*>(slot) + (size_t)wave * (NB * MB * 4) * 64 + lane;
#pragma unroll
        for (int in = 0; in < NB; ++in)
#pragma unroll
            for (int im = 0; im < MB; ++im)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const f32x4 v = src[((in * MB + im) * 4 + gq) * 64];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[in][im][gq * 4 + e] += v[e];
                }
    }
};

// ---- one output tile per workgroup ------------------------------------------------------------------
template <typename T, int BM, int BN, int BK, int NS, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64) void aid_gemm_nt_pipe_kernel(const GemmGroup g) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const TileCoord tc = locate_tile<BM, BN>(g);
    const GemmDesc& P = g.p[tc.p];
    const T* A = reinterpret_cast<const T*>(P.a) + (int64_t)tc.batch * P.stride_a;
    const T* B = reinterpret_cast<const T*>(P.b) + (int64_t)tc.batch * P.stride_b;
    T* C = reinterpret_cast<T*>(P.c) + (int64_t)tc.batch * P.stride_c;
    Engine<T, BM, BN, BK, NS, WM, WN> e;
    e.init(smem_raw);
    e.set_tile(P, A, B, tc.m0, tc.n0);
    e.zero_acc();
    e.mac(0, P.k / BK);
    e.store_tile(P, C, tc.m0, tc.n0);
}

// ---- stream-K: a persistent grid (one workgroup per CU) splits the (tile, K-tile) iteration space of the whole
// group evenly, so a launch whose tile count is not a multiple of the CU count (e.g. 560 tiles on 256 CUs)
// has no ragged tail.  A workgroup's range is contiguous: [tail of a tile][whole tiles][head of a tile].
//   * a segment that does not start at k = 0 (only ever the FIRST segment of a workgroup) is a contribution:
//     the fp32 partial tile goes to this workgroup's scratch slot, then its flag is published (agent-scope
//     release);
//   * a segment that starts at k = 0 but stops early makes this workgroup the tile's owner: it waits for the
//     flags of the following workgroup(s) — which computed their contribution FIRST, long before — adds the
//     partials and stores the tile.
// Contributors never wait, dependencies only point to higher workgroup ids: no cycle, no residency assumption
// beyond forward progress.  Every spin is bounded (timeout -> *err = 1, result wrong but no hang).
struct SkArgs {
    GemmGroup g;
    int32_t  iter_start[AID_GEMM_MAX_PROBLEMS + 1];   // prefix sums of tiles * k-tiles per problem
    int32_t  nk[AID_GEMM_MAX_PROBLEMS];
    int32_t* flags;                                   // [gridDim.x], zeroed by a memset node before every launch
    float*   partials;                                // [gridDim.x][BM * BN]
    int32_t* err;
};

template <typename T, int BM, int BN, int BK, int NS, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64) void aid_gemm_nt_sk_kernel(const SkArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    typedef Engine<T, BM, BN, BK, NS, WM, WN> E;
    E e;
    e.init(smem_raw);
    const int G = gridDim.x;
    const int lid = xcd_remap(blockIdx.x, G);          // neighbours in the iteration space share an XCD
    const int64_t total = a.iter_start[a.g.n_problems];
    const int it_begin = (int)(total * lid / G), it_end = (int)(total * (lid + 1) / G);
    int* s_ok = reinterpret_cast<int*>(smem_raw + E::SMEM);   // in the dynamic segment: a second __shared__ object
                                                              // would make hipcc drain vmcnt before every ds_read

    for (int it = it_begin; it < it_end;) {
        int p = 0;
#pragma unroll
        for (int i = 1; i < AID_GEMM_MAX_PROBLEMS; ++i)
            if (i < a.g.n_problems && it >= a.iter_start[i]) p = i;
        const GemmDesc& P = a.g.p[p];
        const int nk = a.nk[p];
        const int rel = it - a.iter_start[p];
        int tile = rel / nk;
        const int k0 = rel - tile * nk;
        const int k1 = min(nk, k0 + (it_end - it));
        const int tiles_n = (P.n + BN - 1) / BN, tiles_m = (P.m + BM - 1) / BM;
        const int batch = tile / (tiles_m * tiles_n);
        tile -= batch * tiles_m * tiles_n;
        const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
        const T* A = reinterpret_cast<const T*>(P.a) + (int64_t)batch * P.stride_a;
        const T* B = reinterpret_cast<const T*>(P.b) + (int64_t)batch * P.stride_b;
        T* C = reinterpret_cast<T*>(P.c) + (int64_t)batch * P.stride_c;

        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stores of the previous segment: the ring counts DMAs only
        __syncthreads();
        e.set_tile(P, A, B, m0, n0);
        e.zero_acc();
        e.mac(k0, k1);
        it += k1 - k0;

        if (k0 > 0) {                                       // contribution to a tile another workgroup owns
            e.write_partial(a.partials + (size_t)lid * (BM * BN));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (e.tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(a.flags + lid, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            continue;
        }
        if (k1 < nk) {                                      // owner: collect the rest of the K range from the successors
            int remaining = nk - k1;
            for (int c = lid + 1; remaining > 0 && c < G; ++c) {
                const int c_iters = (int)(total * (c + 1) / G) - (int)(total * c / G);
                if (c_iters == 0) continue;
                if (e.tid == 0) {
                    int ok = 1;
                    unsigned spins = 0;
                    while (__hip_atomic_load(a.flags + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                        __builtin_amdgcn_s_sleep(4);
                        if (++spins > (1u << 24)) { ok = 0; *a.err = 1; break; }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    *s_ok = ok;
                }
                __syncthreads();
                if (*s_ok) e.add_partial(a.partials + (size_t)c * (BM * BN));
                __syncthreads();
                remaining -= min(remaining, c_iters);
            }
        }
        e.store_tile(P, C, m0, n0);
    }
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
static hipError_t launch_with_smem(K kernel, size_t smem, bool* attr_set, const GemmGroup& g, int total_tiles,
                                   hipStream_t stream, int threads) {
    if (total_tiles <= 0) return hipSuccess;
    if (!*attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        *attr_set = true;
    }
    hipLaunchKernelGGL(kernel, dim3(total_tiles), dim3(threads), smem, stream, g);
    return hipGetLastError();
}

template <typename T, int BM, int BN, int BK, int NS, int WM, int WN>
static hipError_t launch_pipe(GemmGroup& g, hipStream_t stream) {
    static bool attr_set = false;
    return launch_with_smem(aid_gemm_nt_pipe_kernel<T, BM, BN, BK, NS, WM, WN>, Engine<T, BM, BN, BK, NS, WM, WN>::SMEM,
                            &attr_set, g, plan_tiles(g, BM, BN), stream, WM * WN * 64);
}

// persistent scratch of the stream-K path (flags + fp32 partial tiles), one per device, allocated on first use
// (do the first call outside stream capture); calls on different streams of one device must not overlap
struct SkScratch {
    int32_t* flags = nullptr;
    float*   partials = nullptr;
    int32_t* err = nullptr;
    int      grid = 0;
    size_t   tile_elems = 0;
};
static SkScratch g_sk[16];

static hipError_t sk_scratch(int grid, size_t tile_elems, SkScratch** out) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    SkScratch& s = g_sk[dev & 15];
    if (s.grid < grid || s.tile_elems < tile_elems) {
        if (s.flags) { (void)hipFree(s.flags); (void)hipFree(s.partials); }
        s.grid = grid > s.grid ? grid : s.grid;
        s.tile_elems = tile_elems > s.tile_elems ? tile_elems : s.tile_elems;
        e = hipMalloc(&s.flags, (size_t)(s.grid + 16) * sizeof(int32_t));
        if (e != hipSuccess) return e;
        e = hipMalloc(&s.partials, (size_t)s.grid * s.tile_elems * sizeof(float));
        if (e != hipSuccess) return e;
        e = hipMemset(s.flags, 0, (size_t)(s.grid + 16) * sizeof(int32_t));
        if (e != hipSuccess) return e;
        s.err = s.flags + s.grid;
    }
    *out = &s;
    return hipSuccess;
}

static int g_num_cu = 0;

template <typename T, int BM, int BN, int BK, int NS, int WM, int WN>
static hipError_t launch_sk(GemmGroup& g, hipStream_t stream) {
    typedef Engine<T, BM, BN, BK, NS, WM, WN> E;
    static bool attr_set = false;
    SkArgs a;
    plan_tiles(g, BM, BN);
    a.g = g;
    int iters = 0;
    for (int i = 0; i < g.n_problems; ++i) {
        a.iter_start[i] = iters;
        a.nk[i] = g.p[i].k / BK;
        iters += (g.tile_start[i + 1] - g.tile_start[i]) * a.nk[i];
    }
    for (int i = g.n_problems; i <= AID_GEMM_MAX_PROBLEMS; ++i) a.iter_start[i] = iters;
    for (int i = g.n_problems; i < AID_GEMM_MAX_PROBLEMS; ++i) a.nk[i] = 1;
    if (iters <= 0) return hipSuccess;
    if (g_num_cu == 0) {
        int dev = 0;
        hipDeviceProp_t pr;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return hipErrorInvalidDevice;
        g_num_cu = pr.multiProcessorCount;
    }
    const int grid = iters < g_num_cu ? iters : g_num_cu;
    SkScratch* sc = nullptr;
    hipError_t e = sk_scratch(g_num_cu, (size_t)BM * BN, &sc);
    if (e != hipSuccess) return e;
    a.flags = sc->flags;
    a.partials = sc->partials;
    a.err = sc->err;
    if (!attr_set) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(aid_gemm_nt_sk_kernel<T, BM, BN, BK, NS, WM, WN>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)E::SMEM + 16);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    e = hipMemsetAsync(sc->flags, 0, (size_t)grid * sizeof(int32_t), stream);      // flags are per-launch state
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((aid_gemm_nt_sk_kernel<T, BM, BN, BK, NS, WM, WN>), dim3(grid), dim3(WM * WN * 64), E::SMEM + 16, stream, a);
    return hipGetLastError();
}

template <typename T>
static hipError_t launch_gemm(GemmGroup& g, hipStream_t stream) {
    bool k64 = true;
    for (int i = 0; i < g.n_problems; ++i) k64 = k64 && (g.p[i].k % 64 == 0);
    // development knob (tools/kbench.py): AID_GEMM_VARIANT selects the tile configuration; default 7
    // (table: profiles/r01_gemm_variants.txt)
    static const int variant = getenv("AID_GEMM_VARIANT") ? atoi(getenv("AID_GEMM_VARIANT")) : 7;
    if (k64) {
        switch (variant) {
            case 1:  return launch_pipe<T, 128, 128, 64, 2, 2, 2>(g, stream);   // 4 waves, 64x64 wave tiles, 2 WG/CU
            case 10: return launch_pipe<T, 256, 128, 64, 2, 4, 2>(g, stream);   // 8 waves, 64x64 wave tiles, 96 KB
            case 11: return launch_pipe<T, 256, 128, 64, 3, 4, 2>(g, stream);   // same, 3 stages (144 KB)
            case 12: return launch_pipe<T, 256, 256, 64, 2, 2, 4>(g, stream);   // 8 waves, 128x64 wave tiles, 128 KB
            case 13: return launch_pipe<T, 256, 128, 32, 4, 4, 2>(g, stream);   // BK 32, 4 stages (96 KB)
            case 14: return launch_pipe<T, 256, 128, 32, 6, 4, 2>(g, stream);   // BK 32, 6 stages (144 KB)
            case 15: return launch_pipe<T, 128, 128, 64, 2, 1, 2>(g, stream);   // 2 waves, 128x64 wave tiles, 2 WG/CU
            case 16: return launch_pipe<T, 256, 128, 64, 2, 2, 2>(g, stream);   // 4 waves, 128x64 wave tiles, 1 WG/CU
            case 17: return launch_pipe<T, 128, 128, 64, 2, 2, 1>(g, stream);   // 2 waves, 64x128 wave tiles
            case 20: return launch_sk<T, 256, 128, 64, 3, 4, 2>(g, stream);     // stream-K, 256x128 tiles
            case 21: return launch_sk<T, 256, 256, 64, 2, 2, 4>(g, stream);     // stream-K, 256x256 tiles
            case 22: return launch_sk<T, 128, 128, 64, 2, 2, 4>(g, stream);     // stream-K, 128x128 tiles (1 WG/CU)
            default: return launch_pipe<T, 128, 128, 64, 2, 2, 4>(g, stream);   // 7: 8 waves, 64x32 wave tiles, 2 WG/CU
        }
    }
    static bool s0 = false;
    return launch_with_smem(aid_gemm_nt_kernel<T>, (size_t)2 * (GBM + GBN) * GLD * sizeof(T), &s0, g,
                            plan_tiles(g, GBM, GBN), stream, GTHREADS);
}

hipError_t gemm_group_launch(GemmGroup& g, int dtype, hipStream_t stream) {
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
    return dtype == AID_DTYPE_F16 ? launch_gemm<f16>(g, stream) : launch_gemm<bf16>(g, stream);
}

}  // namespace aid
