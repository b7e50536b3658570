// Row-stationary projection GEMM for the SHORT-K levels of the UNets (C = 320: SD1.5 S = 4096; C = 640: SDXL S = 4096) — the
// attn.to_q / to_k / to_v / to_out[0] calls of the reference (interpolation.py:613, 623-624, 666) where the activation matrix is tall
// (M = frames x 4096 rows) and the contraction is short (K = 320 / 640).
//
// Why another engine.  On these shapes the tile engines of aid_gemm.hip pay per TILE (ring prologue + staged epilogue ~ 6 of 11 us at
// five K tiles, profiles/r04_gemm_notes.txt) and re-read the activation panel once per column tile; measured 611 - 758 TF/s at C = 640
// and 275 TF/s at C = 320 where the operand bytes alone allow 2 - 3x that (57344 x 640 x 640: 147 MB = 23 us of HBM time, 47 GF = 19 us
// of matrix-pipe time; measured 76.9 us).  Here the roles are turned round:
//
//   * a wave keeps its 32 activation rows x the WHOLE K in registers (K = 640: 160 VGPRs, K = 320: 80) — read ONCE from HBM, for the
//     q, k and V^T projections of a self-attention layer together (one launch, one pass over x);
//   * the weights stream through LDS in SLICES of 32 output columns x K (40 KB / 20 KB, a contiguous piece of the [out, in] matrix)
//     by LDS-DMA into a 3-deep ring; every wave of the workgroup multiplies the slice against its resident rows: K / 16 MFMAs
//     32x32x16 into ONE 32 x 32 accumulator block, no K loop over tiles, no prologue per tile;
//   * the 32 x 32 result block leaves straight from the registers: two lane swaps (v_permlane32_swap, v_permlane16_swap) turn the four
//     8-byte column groups a lane holds into ONE 16-byte row segment per lane, a store instruction covers 16 rows x 64 B; bias (scalar
//     loads), scale and residual are applied on the way.  No LDS in the epilogue: with eight waves reading 1-KiB fragments an LDS
//     round trip costs 300+ cycles, and a staged patch paid three of them per slice (profiles/r05_gemm_rs_notes.txt).
//     The value projection is issued with the MFMA operands swapped, so the block arrives as V^T[channel][key] (AidGemmProblem.trans_rows).
//
// LDS layout of a slice: the DMA writes lane-linear 1-KiB pieces, so the slice is the flat 40 / 20 KB image of the weight rows with the
// 16-B chunks of every 1280-B "virtual row" (one weight row at K = 640, two at K = 320) XOR-permuted inside their 256-B groups by
// (virtual row & 15) — on the DMA SOURCE address and again on the fragment read; the 16 lanes of every ds_read_b128 lane group then hit
// 16 distinct 16-B slots of the bank row.
//
// Numerics: fp32 accumulation of 16x16x32 products, 32 of k at a time into one accumulator per 16 x 16 block — the same products as the
// tile engines' 32x32x16 form in another summation grouping, so results agree with theirs to fp32 rounding of the sums (a few 1e-7
// relative before the storage rounding; identical storage values except where a sum sits on a rounding boundary), not bit for bit.
// Which engine a launch gets therefore depends on the SHAPE and the device only, never on the cu_share hint (aid_gemm.hip).
// tests/test_hip_gemm_rs.py holds both engines against the fp64 oracle and against each other.
#include "aid_common.hpp"
#include "aid_kernels.hpp"

#include <string.h>

namespace aid {

constexpr int RS_MAXP = 3;
constexpr int RS_NSTG = 3;                  // ring depth (slices)
constexpr int RS_BIASN = 1280;              // widest biased problem (its bias vector is staged in LDS once per problem)

struct RsProblem {
    const void* b;                          // weights [n, K], K-contiguous, ldb == K
    void*       c;
    const void* bias;                       // [n] or NULL
    const void* residual;                   // like c or NULL
    int32_t     n, ldc, trans_rows;
    float       scale;
    int64_t     stride_c;                   // trans_rows > 0: frame stride of V^T
};

struct RsParams {
    const void* a;                          // activation [m, K], lda == K
    int32_t     m, n_problems, n_slices, nsplit;
    RsProblem   p[RS_MAXP];
    int32_t     slice_start[RS_MAXP + 1];   // prefix sums of 32-column slices
};

template <int N>
__device__ __forceinline__ void rs_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// D(16x16, fp32) += A(16x32) * B(32x16).  Lane l supplies A[i = l & 15][8 (l >> 4) .. + 7] and B[8 (l >> 4) .. + 7][j = l & 15];
// lane l receives D[4 (l >> 4) + e][l & 15], e in [0, 4).
__device__ __forceinline__ f32x4 mfma16(const Vec<f16>::v8& a, const Vec<f16>::v8& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(const Vec<bf16>::v8& a, const Vec<bf16>::v8& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// LDS-DMA of one slice: a wave contributes PPW 1-KiB pieces.  `rs_src` resolves the slice (uniform: its problem's weight bytes), a
// piece is one buffer_load ... lds.  (Unlike the global_load_lds builtin the buffer form does not make hipcc drain the whole DMA ring —
// s_waitcnt vmcnt(0) — in front of every later ds_read's first use; the kernel's counted waits are the only ones.  One specialisation
// per KERNEL instantiation — hence the unused T and TAG: the host pass marks a specialisation that holds buffer builtins invalid after
// its first use and silently drops every later kernel instantiation that calls it.)
struct RsSrc {
    __amdgpu_buffer_rsrc_t rw;
    int so;
};
template <typename T, int K, int TAG>
__device__ __forceinline__ RsSrc rs_src(const RsParams& p, int s) {
    int pi = 0;
#pragma unroll
    for (int i = 1; i < RS_MAXP; ++i)
        if (i < p.n_problems && s >= p.slice_start[i]) pi = i;
    RsSrc r;
    r.rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.p[pi].b), 0, 0x7fffffff, 0x00020000);
    r.so = (s - p.slice_start[pi]) * (32 * K * 2);
    return r;
}
template <typename T, int TAG>
__device__ __forceinline__ void rs_piece(const RsSrc& src, char* dst, int voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(src.rw, (__attribute__((address_space(3))) void*)dst, 16, voff, src.so, 0, 0);
}

// The MFMAs of one slice against the resident rows: K / 32 k-steps of FOUR 16x16x32 products (two column halves of the slice x two
// row halves of the wave's 32 rows) into four independent accumulator blocks.  Round 5 measurement (profiles/r05_gemm_rs_notes.txt):
// one 32 x 32 block fed by 32x32x16 products is a chain of K / 16 MFMAs on ONE accumulator, and a chain issues far below the pipe's
// rate whatever sits between its links (2.2 us per slice of 1.2 us of MFMA time, the same with the fragment reads removed); four blocks
// take turns, every accumulator is touched once per four issues, at the same register count (4 x 4 = 16) and the same LDS traffic
// (two 1-KiB fragment reads per four products).  Fragment reads run PF k-steps ahead (a rolling window, pinned by scheduling barriers:
// hipcc would otherwise hoist every read of the slice to its top — K = 640: 160 + 160 registers).
// TRANS swaps the operands of every product: the blocks arrive as D[m][n] (V^T) instead of D[n][m].
// `bias_at` (or NULL): LDS address of this lane's four bias values of column half 0 (half 1: + 32 B); they are read HERE, with the
// first fragment reads, so that the epilogue finds them in registers (an LDS round trip is 200 - 300 cycles with eight waves reading).
template <typename T, int KT, bool TRANS, int PF>
__device__ __forceinline__ void rs_mac(const char* st, const int (&fr)[2][4], const typename Vec<T>::v8 (&xa)[2][KT], f32x4 (&acc)[2][2],
                                       const char* bias_at, uint32_t (&bq)[4]) {
    typedef typename Vec<T>::v8 T8;
    static_assert(PF >= 1 && PF <= KT, "prefetch window");
    T8 w[PF][2];
#pragma unroll
    for (int j = 0; j < PF; ++j)
#pragma unroll
        for (int a = 0; a < 2; ++a) w[j][a] = *reinterpret_cast<const T8*>(st + fr[a][j & 3] + 256 * (j >> 2));
    if (bias_at) {
        const u32x2 q0 = *reinterpret_cast<const u32x2*>(bias_at), q1 = *reinterpret_cast<const u32x2*>(bias_at + 32);
        bq[0] = q0[0]; bq[1] = q0[1]; bq[2] = q1[0]; bq[3] = q1[1];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < KT; ++t) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
                acc[a][b] = TRANS ? mfma16(xa[b][t], w[t % PF][a], acc[a][b]) : mfma16(w[t % PF][a], xa[b][t], acc[a][b]);
        if (t + PF < KT) {
#pragma unroll
            for (int a = 0; a < 2; ++a) w[t % PF][a] = *reinterpret_cast<const T8*>(st + fr[a][(t + PF) & 3] + 256 * ((t + PF) >> 2));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// One workgroup = NW waves x 32 activation rows; it walks the slices [s_lo, s_hi) of the concatenated problems, starting at a slice
// of its own (ROT) and wrapping round, so that the workgroups of an XCD do not all pull the same 40 KB out of its L2 at the same moment.
template <typename T, int K, int NW, int VAR = 0>
__global__ __launch_bounds__(NW * 64) void aid_gemm_rs_kernel(const RsParams p) {
    typedef typename Vec<T>::v8 T8;
    constexpr int KT = K / 32;                      // k-steps (of four 16x16x32 products) per slice
    constexpr int SLICE = 32 * K * 2;               // bytes per slice
    constexpr int PIECES = SLICE / 1024;            // 1-KiB DMA pieces per slice
    constexpr int PPW = PIECES / NW;                // ... per wave
    constexpr int PF = K == 640 ? 3 : 4;            // k-steps of fragment reads in flight ahead of the MFMAs (8 VGPRs each)
    static_assert(K == 320 || K == 640, "slice layout is written for K = 320 / 640");
    static_assert(PIECES % NW == 0, "DMA pieces divide evenly over the waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kg = lane >> 4;
    char* const ring = smem;
    char* const bias_s = smem + RS_NSTG * SLICE;                    // [RS_BIASN] of T

    // ---- work item: (row tile, slice range); the splits of a row tile sit next to each other on one XCD (they share x in its L2)
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = id / p.nsplit, sp = id - tm * p.nsplit;
    const int per = p.n_slices / p.nsplit;
    const int s_lo = sp * per;
    const int rot = (5 * id) % per;                 // first slice of this workgroup's walk
    const int row0 = tm * (32 * NW) + 32 * wave;    // first activation row of this wave
    auto slice_of = [&](int v) { const int r = v + rot; return s_lo + (r >= per ? r - per : r); };     // v in [0, per + 2)

    // ---- DMA source offsets of this wave's pieces (slice-invariant): LDS chunk position pos <- global chunk with the low four bits of
    // its index inside the 80-chunk virtual row XOR-ed by (virtual row & 15)
    int soff[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int pos = (wave * PPW + j) * 64 + lane;
        const int v = pos / 80, cp = pos - v * 80;
        soff[j] = (v * 80 + (cp ^ (v & 15))) * 16;
    }
    // ---- fragment read addresses: column half a, weight row n = 16 a + l15 of the slice, chunk 4 t + kg:
    //      addr(a, t) = fr[a][t & 3] + 256 (t >> 2)      (checked against the DMA image by tools/dev/rs_layout_check.py)
    int fr[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            if (K == 640) {
                fr[a][tt] = (16 * a + l15) * 1280 + (((4 * tt + kg) ^ l15) << 4);
            } else {                                // two weight rows per 1280-B virtual row: v = n >> 1, u = n & 1, chunk 40 u + 4 t + kg
                const int v = 8 * a + (l15 >> 1), u = l15 & 1;
                fr[a][tt] = v * 1280 + 512 * u + (tt >= 2 ? 256 * u : 0) + (((4 * (tt ^ (2 * u)) + kg) ^ v) << 4);
            }
        }

    char* const mine = ring + wave * (PPW * 1024);                  // this wave's pieces inside a ring stage
    {
        const RsSrc s0 = rs_src<T, K, VAR>(p, slice_of(0)), s1 = rs_src<T, K, VAR>(p, slice_of(1));
#pragma unroll
        for (int j = 0; j < PPW; ++j) rs_piece<T, VAR>(s0, mine + j * 1024, soff[j]);
#pragma unroll
        for (int j = 0; j < PPW; ++j) rs_piece<T, VAR>(s1, mine + SLICE + j * 1024, soff[j]);
    }

    // ---- resident activation fragments: xa[b][t], lane (row 16 b + l15, k-group kg) holds x[row][32 t + 8 kg .. + 7].  Loaded behind
    // the first two slices' DMA requests and tied off HERE: hipcc then places its one vmcnt(0) for them in front of the loop instead of
    // a descending chain of counted waits inside it (it cannot see the asm waits below and would drain the DMA ring in every slice step)
    T8 xa[2][KT];
    {
        const T* __restrict__ xr = reinterpret_cast<const T*>(p.a) + (int64_t)(row0 + l15) * K + 8 * kg;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int t = 0; t < KT; ++t) xa[b][t] = *reinterpret_cast<const T8*>(xr + (int64_t)16 * b * K + 32 * t);
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int t = 0; t < KT; ++t) asm volatile("" : "+v"(xa[b][t]));
    }

    // ---- epilogue of the four 16 x 16 blocks of a slice.  Lane (l15, kg) holds, of block (a, b), the four values
    // (n = 16 a + 4 kg + e, m = 16 b + l15) [V^T: (m = 16 b + 4 kg + e, n = 16 a + l15)]: scale, bias, rounding, then per row half two
    // lane swaps so that lane (l15, j) owns the 16-byte segment j of its row —
    //     v_permlane32_swap (X = block column half 0, Y = half 1):  X' = [X.r0 X.r1 Y.r0 Y.r1]   Y' = [X.r2 X.r3 Y.r2 Y.r3]   (r = 16-lane row = kg)
    //     v_permlane16_swap (X', Y'):                                lo = [X.r0 X.r2 Y.r0 Y.r2]   hi = [X.r1 X.r3 Y.r1 Y.r3]
    // i.e. lane row j ends up with (kg = 2 (j & 1), kg = 2 (j & 1) + 1) of column half j >> 1: columns 8 j .. 8 j + 7 — and ONE 16-byte
    // store per row half: 16 rows x 64 B per instruction.  The stores are plain global stores: issued before the step's DMA requests,
    // they are older than every DMA piece that may stay in flight at the next counted wait (vmcnt(PPW) is exact whatever order loads
    // and stores retire in).  Its parameters are uniform scalars: the LATE waves run it one step after the MFMAs.
    struct Epi {
        float scale;
        int   ldc, trans, biased;
        T*    c;                                    // this wave's first output element of the slice
        const T* r;                                 // residual at the same position or NULL
    };
    auto epilogue = [&](f32x4 (&acc)[2][2], const Epi& E, const uint32_t (&bq)[4]) {
        int ln = lane;                              // (opaque: what is derived from it is recomputed per call instead of living in
        asm volatile("" : "+v"(ln));                //  registers the resident rows have taken)
        const int e15 = ln & 15, ekg = ln >> 4;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) asm volatile("" : "+v"(acc[a][b]));
        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");                           // MFMA -> VALU wait states on every path (aid_gemm.hip mfma_fence)
        uint32_t pk[2][2][2];                       // [a][b][dword]: four rounded values of a block, packed
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (E.biased) {                         // the lane's four bias values of this column half (read by rs_mac): 16 a + 4 kg + e
                typedef T T2 __attribute__((ext_vector_type(2)));
                const T2 p0 = __builtin_bit_cast(T2, bq[2 * a]), p1 = __builtin_bit_cast(T2, bq[2 * a + 1]);
                bv = f32x4{(float)p0[0], (float)p0[1], (float)p1[0], (float)p1[1]};
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float tv = fmaf(acc[a][b][e], E.scale, bv[e]);
                    asm volatile("" : "+v"(tv));                                    // no v_pk_fma_f32 pairing (slower beside MFMAs)
                    v[e] = tv;
                }
                const u32x2 q = __builtin_bit_cast(u32x2, cvt4<T>(v));
                pk[a][b][0] = q[0];
                pk[a][b][1] = q[1];
            }
        }
        // row half h2 of the block = b (rows = activation rows) or a (V^T: rows = channels); the column halves are the other index
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            u32x4 o;
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const uint32_t x = E.trans ? pk[h2][0][d] : pk[0][h2][d];           // column half 0
                const uint32_t y = E.trans ? pk[h2][1][d] : pk[1][h2][d];           // column half 1
                const auto s1 = __builtin_amdgcn_permlane32_swap(x, y, false, false);
                const auto s2 = __builtin_amdgcn_permlane16_swap(s1[0], s1[1], false, false);
                o[d] = s2[0];
                o[2 + d] = s2[1];
            }
            const uint32_t off = (uint32_t)((16 * h2 + e15) * E.ldc + 8 * ekg);     // (32 rows of at most 2^20 elements: 32-bit)
            if (E.r) {                                                              // added after the rounding, like the block's separate add
                const f32x8 r8 = up8<T>(*reinterpret_cast<const T8*>(E.r + off));
                f32x8 sum = up8<T>(__builtin_bit_cast(T8, o));
#pragma unroll
                for (int e = 0; e < 8; ++e) sum[e] += r8[e];
                o = __builtin_bit_cast(u32x4, cvt8<T>(sum));
            }
            if (VAR == 4) asm volatile("" :: "v"(o));                               // ablation: no global stores
            else          *reinterpret_cast<u32x4*>(E.c + off) = o;
        }
    };

    // ---- the two waves of a SIMD run HALF A STEP APART: waves 0 .. NW/2 - 1 ("early") finish a slice with its epilogue, waves NW/2 ..
    // ("late", their SIMD partners) carry the accumulators over the barrier and run the epilogue at the START of the next step — so
    // between two barriers one wave of a SIMD does [MFMAs | epilogue] and its partner [epilogue | MFMAs]: the matrix pipe always has a
    // taker (one set of accumulator blocks is live per wave either way: no extra registers).
    const bool late = VAR != 8 && wave >= NW / 2;                   // (8: development, every wave early = lock step)
    Epi ep;
    ep.scale = 1.f; ep.ldc = 0; ep.trans = 0; ep.biased = 0; ep.c = nullptr; ep.r = nullptr;
    bool ep_on = false;
    f32x4 acc[2][2];
    uint32_t bq[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // development (VAR == 9): shader-clock stamps of the step's phases, waves 0 and NW / 2 of the LAST workgroup, written over the first
    // bytes of problem 0's output (its own rows are elsewhere) — [wave][slice][6] int64
    long long* const tl = reinterpret_cast<long long*>(p.p[0].c) + (wave >= NW / 2 ? 6 * 64 : 0);
    const bool tl_on = VAR == 9 && id == (int)gridDim.x - 1 && (wave == 0 || wave == NW / 2) && lane == 0;
#define RS_STAMP(i) do { if (VAR == 9) { const long long t_ = __builtin_readcyclecounter(); if (tl_on && v < 64) tl[6 * v + (i)] = t_; } } while (0)
    // The DMA pieces of slice v + 2 are requested in a wave's NON-MFMA phase (early: behind its epilogue, late: behind the previous
    // slice's epilogue at the step's start): a piece costs its wave 100+ cycles of issue, which then overlap the SIMD partner's MFMAs
    // instead of standing inside the wave's own MFMA stream (1830 instead of 1280 cycles per 80 products, profiles/r05_gemm_rs_notes.txt).
    auto request = [&](int v, int fill) {
        if (v + 2 < per && VAR != 7) {                                  // (7: ablation, no DMA stream)
            const RsSrc src = rs_src<T, K, VAR>(p, slice_of(v + 2));
#pragma unroll
            for (int j = 0; j < PPW; ++j) rs_piece<T, VAR>(src, mine + fill * SLICE + j * 1024, soff[j]);
        }
    };
    int stage = 0, fill = 2, pi_cur = -1;
    for (int v = 0; v < per; ++v) {
        RS_STAMP(0);
        const int s = slice_of(v);
        int pi = 0;                                  // uniform: the problem of slice s
#pragma unroll
        for (int i = 1; i < RS_MAXP; ++i)
            if (i < p.n_problems && s >= p.slice_start[i]) pi = i;
        const RsProblem& P = p.p[pi];
        const bool biased = P.bias != nullptr;
        if (pi != pi_cur) {
            pi_cur = pi;
            // the bias vector goes through LDS once per problem (a per-slice global load would make hipcc drain the DMA ring).  A late
            // wave's pending epilogue already holds its bias values in registers; the barrier keeps a second biased problem off a
            // vector that slower waves still read, the step's barrier below publishes it.
            if (biased) {
                __builtin_amdgcn_s_barrier();
                for (int i = tid; i < P.n / 4; i += NW * 64)
                    reinterpret_cast<u32x2*>(bias_s)[i] = reinterpret_cast<const u32x2*>(P.bias)[i];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (a raw s_barrier does not wait for this wave's LDS writes)
            }
        }
        // slice s has landed once at most the PPW pieces of ONE later slice are outstanding (every store is older than they are).
        // Early waves requested slice s + 1 at the end of the previous step, late waves at its start: for both, exactly one
        // slice's pieces may be in flight here.
        if (v + 1 < per) rs_wait_vm<PPW>();
        else             rs_wait_vm<0>();
        RS_STAMP(1);
        __builtin_amdgcn_s_barrier();               // every wave's pieces of slice s are in LDS; nobody still reads the slice before it
        asm volatile("" ::: "memory");
        RS_STAMP(2);
        if (late) {                                  // the previous slice's blocks, then this wave's requests for slice s + 2
            if (ep_on) { epilogue(acc, ep, bq); ep_on = false; }
            request(v, fill);
        }
        const char* st = ring + stage * SLICE;
        const int n0 = (s - p.slice_start[pi]) * 32;
        const bool trans = P.trans_rows > 0;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        RS_STAMP(3);
        {
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const char* bias_at = biased ? bias_s + (n0 + 4 * (ln >> 4)) * 2 : nullptr;
            if (VAR == 6) asm volatile("" : "+v"(acc[0][0]));                               // ablation: no MFMAs
            else if (trans) rs_mac<T, KT, true, PF>(st, fr, xa, acc, bias_at, bq);
            else            rs_mac<T, KT, false, PF>(st, fr, xa, acc, bias_at, bq);
        }
        if (VAR == 9) asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]));
        RS_STAMP(4);
        ep.scale = P.scale; ep.ldc = P.ldc; ep.trans = trans; ep.biased = biased;
        if (trans) {
            const int frame = row0 / P.trans_rows, key0 = row0 - frame * P.trans_rows;
            ep.r = nullptr;
            ep.c = reinterpret_cast<T*>(P.c) + (int64_t)frame * P.stride_c + (int64_t)n0 * P.ldc + key0;
        } else {
            ep.c = reinterpret_cast<T*>(P.c) + (int64_t)row0 * P.ldc + n0;
            ep.r = P.residual ? reinterpret_cast<const T*>(P.residual) + (int64_t)row0 * P.ldc + n0 : nullptr;
        }
        if (late) {
            ep_on = true;
        } else {
            epilogue(acc, ep, bq);
            if (VAR == 9) { asm volatile("" ::: "memory"); RS_STAMP(3); }      // (development: the early wave's stamp 3 is re-used: epilogue end)
            request(v, fill);
        }
        RS_STAMP(5);
        stage = (stage + 1 == RS_NSTG) ? 0 : stage + 1;
        fill = (fill + 1 == RS_NSTG) ? 0 : fill + 1;
    }
#undef RS_STAMP
    if (ep_on) epilogue(acc, ep, bq);
}

template <typename T, int K, int NW, int VAR = 0>
static hipError_t rs_launch(const RsParams& p, hipStream_t stream) {
    static PerDevice<bool> attr_set;
    constexpr size_t smem = (size_t)RS_NSTG * 32 * K * 2 + (size_t)RS_BIASN * 2;
    static_assert(smem <= 160 * 1024, "one workgroup's LDS");
    bool* done = attr_set.slot();
    if (!done) return hipErrorInvalidDevice;
    if (!*done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(aid_gemm_rs_kernel<T, K, NW, VAR>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        *done = true;
    }
    const int grid = (p.m / (32 * NW)) * p.nsplit;
    hipLaunchKernelGGL((aid_gemm_rs_kernel<T, K, NW, VAR>), dim3(grid), dim3(NW * 64), smem, stream, p);
    return hipGetLastError();
}

static inline bool al(const void* q, uintptr_t a) { return (reinterpret_cast<uintptr_t>(q) & (a - 1)) == 0; }

// Does the row-stationary engine run this group?  One shared activation (the q / k / V^T projections of a self-attention layer, or a
// single projection), K = 320 / 640 with dense operands, whole 32-row wave blocks and 32-column slices, and enough row tiles to
// fill the device (a short activation is better served by the 2-D tiles of aid_gemm.hip).  `ncu` = the CUs the launch may count on.
bool gemm_rs_supported(const GemmGroup& g, int ncu, bool ignore_size) {
    if (g.n_problems < 1 || g.n_problems > RS_MAXP) return false;
    const GemmDesc& p0 = g.p[0];
    const int k = p0.k;
    if (k != 320 && k != 640) return false;
    const int tm = k == 640 ? 256 : 128;
    if (p0.m <= 0 || p0.m % tm || p0.lda != k || !al(p0.a, 16)) return false;
    for (int i = 0; i < g.n_problems; ++i) {
        const GemmDesc& d = g.p[i];
        if (d.a != p0.a || d.m != p0.m || d.k != k || d.lda != k || d.ldb != k || d.batch != 1) return false;
        if (d.n < 32 || d.n % 32 || d.ln_stats || !al(d.b, 16) || !al(d.c, 16) || d.ldc % 8) return false;
        if (d.bias && (!al(d.bias, 8) || d.n > RS_BIASN)) return false;
        if (d.residual && !al(d.residual, 16)) return false;
        if (d.trans_rows) {
            if (d.trans_rows % 32 || d.m % d.trans_rows || d.stride_c % 8 || d.bias || d.residual) return false;
        } else if (d.ldc < d.n) {
            return false;
        }
    }
    // a tall activation: at least 64 rows per CU of the device (K = 640: a quarter of the CUs get a 256-row tile and the slice range is
    // split to cover the rest; K = 320: half of them get a 128-row tile).  A lone K = 320 projection (ten slices: little to amortise the
    // load of the resident rows over) needs three times that — measured 28672 rows: 20.8 us here, 18.5 on the lock-step tiles; 57344
    // rows: 27.8 against 29.1 (profiles/r05_gemm_rs_ab.txt).
    int slices = 0;
    for (int i = 0; i < g.n_problems; ++i) slices += g.p[i].n / 32;
    const int rows_per_cu = (k == 320 && slices < 20) ? 192 : 64;
    if (ignore_size) return true;
    if (p0.m < rows_per_cu * ncu) return false;
    // more row tiles than workgroup slots: whole rounds only.  A 9-frame sequence shard (18 frames batched, 73728 rows at C = 640) is 288
    // tiles on 256 slots = 0.56 of two rounds: 625 TF/s here against 930 at 512 tiles and ~770 on the tile engines
    const int tiles = p0.m / tm, slots = (k == 640 ? 1 : 2) * ncu;
    if (tiles > slots) {
        const int rounds = (tiles + slots - 1) / slots;
        if ((double)tiles < 0.8 * (double)rounds * slots) return false;
    }
    return true;
}

hipError_t gemm_rs_launch(const GemmGroup& g, int dtype, int ncu, hipStream_t stream) {
    RsParams p;
    memset(&p, 0, sizeof(p));
    const GemmDesc& p0 = g.p[0];
    p.a = p0.a;
    p.m = p0.m;
    p.n_problems = g.n_problems;
    int ns = 0;
    for (int i = 0; i < g.n_problems; ++i) {
        const GemmDesc& d = g.p[i];
        RsProblem& q = p.p[i];
        q.b = d.b; q.c = d.c; q.bias = d.bias; q.residual = d.residual;
        q.n = d.n; q.ldc = d.ldc; q.trans_rows = d.trans_rows; q.scale = d.scale; q.stride_c = d.stride_c;
        p.slice_start[i] = ns;
        ns += d.n / 32;
    }
    for (int i = g.n_problems; i <= RS_MAXP; ++i) p.slice_start[i] = ns;
    p.n_slices = ns;
    // split the slice range of a row tile over several workgroups while that fills CUs that would idle otherwise; every part keeps
    // at least eight slices (the resident rows are re-read once per part)
    const int k = p0.k;
    const int tiles = p0.m / (k == 640 ? 256 : 128);
    const int slots = k == 640 ? ncu : 2 * ncu;
    int nsplit = 1;
    for (int c = 2; c <= 8; ++c)
        if (tiles * c <= slots && ns % c == 0 && ns / c >= 8) nsplit = c;
    p.nsplit = nsplit;
#ifdef AID_RS_VARIANTS                          // development builds: timing ablations of the K = 640 bf16 kernel behind GEMM_PP
    if (k == 640 && dtype == AID_DTYPE_BF16) {
        switch (tune(TUNE_GEMM_PP)) {
            case 5: return rs_launch<bf16, 640, 8, 5>(p, stream);      // stores may stay in flight at the counted wait (UNSAFE)
            case 6: return rs_launch<bf16, 640, 8, 6>(p, stream);      // no MFMAs
            case 7: return rs_launch<bf16, 640, 8, 7>(p, stream);      // no DMA stream
            case 3: return rs_launch<bf16, 640, 8, 9>(p, stream);      // phase time stamps over the first output bytes
            case 4: return rs_launch<bf16, 640, 8, 8>(p, stream);      // every wave runs its epilogue right behind its MFMAs (lock step)
            default: break;
        }
    }
#endif
    if (k == 640) return dtype == AID_DTYPE_F16 ? rs_launch<f16, 640, 8>(p, stream) : rs_launch<bf16, 640, 8>(p, stream);
    return dtype == AID_DTYPE_F16 ? rs_launch<f16, 320, 4>(p, stream) : rs_launch<bf16, 320, 4>(p, stream);
}

}  // namespace aid
