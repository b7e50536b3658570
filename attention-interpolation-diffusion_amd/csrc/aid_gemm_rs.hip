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
//   * the 32 x 32 result block leaves through a wave-private LDS patch (two slices = 64 columns = one full 128-B line per row) as
//     16-byte row segments while the workgroup's other waves keep the matrix pipe busy; bias / scale / residual are applied there.
//     The value projection is issued with the MFMA operands swapped, so the block arrives as V^T[channel][key] (AidGemmProblem.trans_rows).
//
// LDS layout of a slice: the DMA writes lane-linear 1-KiB pieces, so the slice is the flat 40 / 20 KB image of the weight rows with the
// 16-B chunks of every 1280-B "virtual row" (one weight row at K = 640, two at K = 320) XOR-permuted inside their 256-B groups by
// (virtual row & 15) — on the DMA SOURCE address and again on the fragment read; the 16 lanes of every ds_read_b128 lane group then hit
// 16 distinct 16-B slots of the bank row.
//
// Bit-compatibility: the accumulation order over k is the tile engines' (ascending, 16 at a time, one fp32 accumulator), so results equal
// theirs bit for bit; tests/test_hip_gemm_rs.py holds both engines against the fp64 oracle and against each other.
#include "aid_common.hpp"
#include "aid_kernels.hpp"

#include <string.h>

namespace aid {

constexpr int RS_MAXP = 3;
constexpr int RS_NSTG = 3;                  // ring depth (slices)
constexpr int RS_CROW = 136;                // bytes per row of the wave-private output patch: 128 + 8 (conflict-free 8-B writes)
constexpr int RS_CSTG = 32 * RS_CROW;       // 4352 B per wave
constexpr int RS_BIASN = 1280;              // widest biased problem (its bias vector is staged in LDS)

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

// The PPW 1-KiB pieces this wave contributes to slice `s` (uniform: the problem of the slice, its weight bytes).
// (buffer_load ... lds: unlike the global_load_lds builtin it does not make hipcc drain the whole DMA ring — s_waitcnt vmcnt(0) — in
// front of every later ds_read's first use; the kernel's counted waits are the only ones.  One specialisation per
// KERNEL instantiation (hence the unused T): the host pass marks a specialisation that holds buffer builtins invalid after its first
// use and silently drops every later kernel instantiation that calls it.)
template <typename T, int K, int PPW>
__device__ __forceinline__ void rs_dma(const RsParams& p, int s, char* dst, const int (&soff)[PPW]) {
    int pi = 0;
#pragma unroll
    for (int i = 1; i < RS_MAXP; ++i)
        if (i < p.n_problems && s >= p.slice_start[i]) pi = i;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.p[pi].b), 0, 0x7fffffff, 0x00020000);
    const int so = (s - p.slice_start[pi]) * (32 * K * 2);
#pragma unroll
    for (int j = 0; j < PPW; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, soff[j], so, 0, 0);
}

// The K / 16 MFMAs of one slice against the resident rows.  Fragment reads run PF k-steps ahead of the MFMAs that consume them (a rolling
// window of PF registers quads: the read of step t + PF goes into the registers step t's MFMA has just consumed); one scheduling barrier
// per k-step pins that order — hipcc would otherwise hoist every read of the slice to its top (K = 640: 160 + 160 registers), and a
// window of two steps measured 2.2x the matrix-pipe time per slice (an LDS round trip per pair of MFMAs, profiles/r05_gemm_rs_notes.txt).
// TRANS swaps the MFMA operands: the block arrives as D[m][n] (lane (n = l31, hi) holds m = 8 g + 4 hi + e) instead of D[n][m].
template <typename T, int KT, bool TRANS, int PF>
__device__ __forceinline__ void rs_mac(const char* st, const int (&fr)[8], const typename Vec<T>::v8 (&xa)[KT], f32x16& acc) {
    typedef typename Vec<T>::v8 T8;
    static_assert(PF >= 2 && PF <= KT, "prefetch window");
    T8 w[PF];
#pragma unroll
    for (int j = 0; j < PF; ++j) w[j] = *reinterpret_cast<const T8*>(st + fr[j & 7] + 256 * (j >> 3));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        acc = TRANS ? mfma32(xa[t], w[t % PF], acc) : mfma32(w[t % PF], xa[t], acc);
        if (t + PF < KT) w[t % PF] = *reinterpret_cast<const T8*>(st + fr[(t + PF) & 7] + 256 * ((t + PF) >> 3));
        __builtin_amdgcn_sched_barrier(0);
    }
}

// One workgroup = NW waves x 32 activation rows; it walks the slices [s_lo, s_hi) of the concatenated problems.
template <typename T, int K, int NW>
__global__ __launch_bounds__(NW * 64) void aid_gemm_rs_kernel(const RsParams p) {
    typedef typename Vec<T>::v8 T8;
    typedef typename Vec<T>::v4 T4;
    constexpr int KT = K / 16;                      // MFMA k-steps per slice
    constexpr int SLICE = 32 * K * 2;               // bytes per slice
    constexpr int PIECES = SLICE / 1024;            // 1-KiB DMA pieces per slice
    constexpr int PPW = PIECES / NW;                // ... per wave
    constexpr int PF = K == 640 ? 6 : 8;            // fragment reads in flight ahead of the MFMAs (register budget: 4 VGPRs each)
    static_assert(K == 320 || K == 640, "slice layout is written for K = 320 / 640");
    static_assert(PIECES % NW == 0, "DMA pieces divide evenly over the waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    char* const ring = smem;
    char* const cst = smem + RS_NSTG * SLICE + wave * RS_CSTG;
    T* const bias_s = reinterpret_cast<T*>(smem + RS_NSTG * SLICE + NW * RS_CSTG);      // [RS_BIASN]

    // ---- work item: (row tile, slice range); the splits of a row tile sit next to each other on one XCD (they share x in its L2)
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = id / p.nsplit, sp = id - tm * p.nsplit;
    const int per = p.n_slices / p.nsplit;
    const int s_lo = sp * per, s_hi = s_lo + per;
    const int row0 = tm * (32 * NW) + 32 * wave;    // first activation row of this wave

    // ---- DMA source offsets of this wave's pieces (slice-invariant): LDS chunk position pos <- global chunk with the low four bits of
    // its index inside the 80-chunk virtual row XOR-ed by (virtual row & 15)
    int soff[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int pos = (wave * PPW + j) * 64 + lane;
        const int v = pos / 80, cp = pos - v * 80;
        soff[j] = (v * 80 + (cp ^ (v & 15))) * 16;
    }
    // ---- fragment read addresses: weight row i = l31 of the slice, chunk 2 t + hi;  addr(t) = fr[t & 7] + 256 (t >> 3)
    int fr[8];
    if (K == 640) {
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) fr[tt] = l31 * 1280 + (((2 * tt + hi) ^ (l31 & 15)) << 4);
    } else {                                        // two weight rows per virtual row: v = i >> 1, u = i & 1, chunk 40 u + 2 t + hi
        const int v = l31 >> 1, u = l31 & 1;
#pragma unroll
        for (int tt = 0; tt < 8; ++tt)
            fr[tt] = v * 1280 + 512 * u + (tt >= 4 ? 256 * u : 0) + (((2 * (tt ^ (4 * u)) + hi) ^ (v & 15)) << 4);
    }

    char* const mine = ring + wave * (PPW * 1024);                  // this wave's pieces inside a ring stage
    rs_dma<T, K, PPW>(p, s_lo, mine, soff);
    if (s_lo + 1 < s_hi) rs_dma<T, K, PPW>(p, s_lo + 1, mine + SLICE, soff);

    // ---- resident activation fragments: lane (row l31, half hi) holds x[row][16 t + 8 hi .. + 7].  Loaded behind the first two slices'
    // DMA requests and tied off HERE: hipcc then places its one vmcnt(0) for them in front of the loop instead of a descending chain
    // of counted waits inside it (it cannot see the asm waits below and would drain the DMA ring to three pieces in every slice step)
    T8 xa[KT];
    {
        const T* __restrict__ xr = reinterpret_cast<const T*>(p.a) + (int64_t)(row0 + l31) * K + 8 * hi;
#pragma unroll
        for (int t = 0; t < KT; ++t) xa[t] = *reinterpret_cast<const T8*>(xr + 16 * t);
#pragma unroll
        for (int t = 0; t < KT; ++t) asm volatile("" : "+v"(xa[t]));
    }


    // ---- deferred flush of the output patch.  The stores of a finished column pair are issued at the START of a later slice step,
    // in front of that step's DMA requests: at the next counted wait every store is then older than the DMA pieces that may stay in
    // flight, so `vmcnt(PPW)` is exact whatever order loads and stores retire in, and a store has a whole slice step to complete.
    int  f_kind = 0;                                // 0: nothing pending, 1: 32 rows x 64 columns, 2: V^T 32 channels x 32 keys
    T*   f_c = nullptr;                             // this wave's first output element of the pending block
    const T* f_r = nullptr;                         // residual at the same position (kind 1) or NULL
    int  f_ldc = 0;
    auto flush = [&]() {
        // (the lane id is made opaque here and in the epilogue: everything derived from it is then recomputed per call — a handful of
        // VALU operations — instead of being hoisted out of the slice loop into registers the resident rows have taken)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        if (f_kind == 1 && !f_r) {
            // (two code paths, with and without a residual, each with its own stores: a merged tail would carry hipcc's vmcnt(0) for
            // the residual loads onto the path that has none — and drain the DMA ring in every flush)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = 8 * q + (ln >> 3), ch = ln & 7;
                const T4 lo = *reinterpret_cast<const T4*>(cst + row * RS_CROW + ch * 16);
                const T4 up = *reinterpret_cast<const T4*>(cst + row * RS_CROW + ch * 16 + 8);
                T8 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) { o[e] = lo[e]; o[4 + e] = up[e]; }
                *reinterpret_cast<T8*>(f_c + (int64_t)row * f_ldc + 8 * ch) = o;
            }
        } else if (f_kind == 1) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {                                        // two row groups at a time (register budget)
                T8 o[2], rs[2];
#pragma unroll
                for (int q = 0; q < 2; ++q)                                         // both residual loads in flight before the first use
                    rs[q] = *reinterpret_cast<const T8*>(f_r + (int64_t)(8 * (2 * h2 + q) + (ln >> 3)) * f_ldc + 8 * (ln & 7));
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int row = 8 * (2 * h2 + q) + (ln >> 3), ch = ln & 7;
                    const T4 lo = *reinterpret_cast<const T4*>(cst + row * RS_CROW + ch * 16);
                    const T4 up = *reinterpret_cast<const T4*>(cst + row * RS_CROW + ch * 16 + 8);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { o[q][e] = lo[e]; o[q][4 + e] = up[e]; }
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) {                                       // added after the rounding, like the block's separate add
                    const f32x8 r8 = up8<T>(rs[q]);
                    f32x8 sum = up8<T>(o[q]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) sum[e] += r8[e];
                    o[q] = cvt8<T>(sum);
                }
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    *reinterpret_cast<T8*>(f_c + (int64_t)(8 * (2 * h2 + q) + (ln >> 3)) * f_ldc + 8 * (ln & 7)) = o[q];
            }
        } else if (f_kind == 2) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int row = 16 * q + (ln >> 2), ch = ln & 3;
                const T4 lo = *reinterpret_cast<const T4*>(cst + row * RS_CROW + ch * 16);
                const T4 up = *reinterpret_cast<const T4*>(cst + row * RS_CROW + ch * 16 + 8);
                T8 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) { o[e] = lo[e]; o[4 + e] = up[e]; }
                *reinterpret_cast<T8*>(f_c + (int64_t)row * f_ldc + 8 * ch) = o;
            }
        }
        f_kind = 0;
    };

    // ---- epilogue of one 32 x 32 block: scale / bias / rounding -> the wave's patch.  Rows of the patch = activation rows (columns
    // [32 (slice & 1), + 32), flushed after the odd slice as 128 B per row) or, for V^T, channels (32 keys = 64 B per channel row,
    // flushed every slice).  Its parameters are uniform scalars: the LATE waves (below) run it one step after the MFMAs.
    struct Epi {
        float scale;
        int   biased, cp, n0, kind, ldc;            // kind: 0 = first half of a column pair (nothing to flush yet), 1 / 2 as f_kind
        T*    c;
        const T* r;
    };
    auto epilogue = [&](f32x16& acc, const Epi& E) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int el31 = ln & 31, ehi = ln >> 5;
        asm volatile("" : "+v"(acc));
        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");                           // MFMA -> VALU wait states on every path (aid_gemm.hip mfma_fence)
        asm volatile("" : "+v"(acc));
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (E.biased) bv = up4<T>(*reinterpret_cast<const T4*>(bias_s + E.n0 + 8 * g + 4 * ehi));
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float tv = fmaf(acc[4 * g + e], E.scale, bv[e]);
                asm volatile("" : "+v"(tv));                                        // no v_pk_fma_f32 pairing (slower beside MFMAs)
                v[e] = tv;
            }
            *reinterpret_cast<T4*>(cst + el31 * RS_CROW + (E.cp + 8 * g + 4 * ehi) * 2) = cvt4<T>(v);
        }
        if (E.kind) { f_kind = E.kind; f_c = E.c; f_r = E.r; f_ldc = E.ldc; }
    };

    // ---- the two waves of a SIMD run HALF A STEP APART: waves 0 .. NW/2 - 1 ("early") finish a slice with its epilogue, waves NW/2 ..
    // ("late", their SIMD partners) carry the accumulator over the barrier and run the epilogue at the START of the next step — so
    // between two barriers one wave of a SIMD does [MFMAs | epilogue] and its partner [epilogue | MFMAs]: the matrix pipe always has a
    // taker (one accumulator block is live per wave either way: no extra registers).
    const bool late = wave >= NW / 2;
    Epi ep;
    ep.scale = 1.f; ep.biased = 0; ep.cp = 0; ep.n0 = 0; ep.kind = 0; ep.ldc = 0; ep.c = nullptr; ep.r = nullptr;
    bool ep_on = false;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    int stage = 0, fill = 2;
    int s = s_lo;
    for (int pi = 0; pi < p.n_problems; ++pi) {
        const int pe = min(p.slice_start[pi + 1], s_hi);
        if (s >= pe) continue;
        const RsProblem& P = p.p[pi];
        // a late wave's pending epilogue belongs to the previous problem (its bias vector is about to be replaced)
        if (ep_on) { epilogue(acc, ep); ep_on = false; }
        // the bias vector goes through LDS once per problem: a per-slice global load would make hipcc drain the DMA ring (vmcnt(0))
        // in every slice step.  The loop's first barrier publishes it; the barrier here keeps a second biased problem off a
        // vector that slower waves still read.
        const bool biased = P.bias != nullptr;
        if (biased) {
            __builtin_amdgcn_s_barrier();
            for (int i = tid; i < P.n / 4; i += NW * 64)
                reinterpret_cast<T4*>(bias_s)[i] = reinterpret_cast<const T4*>(P.bias)[i];
        }
        const T* R = reinterpret_cast<const T*>(P.residual);
        T* C = reinterpret_cast<T*>(P.c);
        const float scale = P.scale;
        const int ldc = P.ldc;
        const bool trans = P.trans_rows > 0;
        const int frame = trans ? row0 / P.trans_rows : 0;
        const int key0 = trans ? row0 - frame * P.trans_rows : 0;
        for (; s < pe; ++s) {
            // slice s has landed once at most the PPW pieces of slice s + 1 are outstanding (every store is older than they are)
            if (s + 1 < s_hi) rs_wait_vm<PPW>();
            else              rs_wait_vm<0>();
            __builtin_amdgcn_s_barrier();           // every wave's pieces of slice s are in LDS; nobody still reads slice s - 1
            asm volatile("" ::: "memory");
            if (ep_on) { epilogue(acc, ep); ep_on = false; }            // late waves: the previous slice's block
            flush();
            if (s + 2 < s_hi) rs_dma<T, K, PPW>(p, s + 2, mine + fill * SLICE, soff);
            const char* st = ring + stage * SLICE;
            const int n0 = (s - p.slice_start[pi]) * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            if (trans) rs_mac<T, KT, true, PF>(st, fr, xa, acc);
            else       rs_mac<T, KT, false, PF>(st, fr, xa, acc);
            ep.scale = scale; ep.biased = biased; ep.n0 = n0; ep.ldc = ldc;
            if (trans) {
                ep.cp = 0; ep.kind = 2; ep.r = nullptr;
                ep.c = C + (int64_t)frame * P.stride_c + (int64_t)n0 * ldc + key0;
            } else {
                ep.cp = n0 & 32; ep.kind = ep.cp ? 1 : 0;
                ep.c = C + (int64_t)row0 * ldc + (n0 - 32);
                ep.r = R ? R + (int64_t)row0 * ldc + (n0 - 32) : nullptr;
            }
            if (late) ep_on = true;
            else      epilogue(acc, ep);
            stage = (stage + 1 == RS_NSTG) ? 0 : stage + 1;
            fill = (fill + 1 == RS_NSTG) ? 0 : fill + 1;
        }
    }
    if (ep_on) epilogue(acc, ep);
    flush();
}

template <typename T, int K, int NW>
static hipError_t rs_launch(const RsParams& p, hipStream_t stream) {
    static PerDevice<bool> attr_set;
    constexpr size_t smem = (size_t)RS_NSTG * 32 * K * 2 + (size_t)NW * RS_CSTG + (size_t)RS_BIASN * 2;
    static_assert(smem <= 160 * 1024, "one workgroup's LDS");
    bool* done = attr_set.slot();
    if (!done) return hipErrorInvalidDevice;
    if (!*done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(aid_gemm_rs_kernel<T, K, NW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        *done = true;
    }
    const int grid = (p.m / (32 * NW)) * p.nsplit;
    hipLaunchKernelGGL((aid_gemm_rs_kernel<T, K, NW>), dim3(grid), dim3(NW * 64), smem, stream, p);
    return hipGetLastError();
}


static inline bool al(const void* q, uintptr_t a) { return (reinterpret_cast<uintptr_t>(q) & (a - 1)) == 0; }

// Does the row-stationary engine run this group?  One shared activation (the q / k / V^T projections of a self-attention layer, or a
// single projection), K = 320 / 640 with dense operands, whole 32-row wave blocks and 64-column slice pairs, and enough row tiles to
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
        if (d.n < 64 || d.n % 64 || d.ln_stats || !al(d.b, 16) || !al(d.c, 16) || d.ldc % 8) return false;
        if (d.bias && (!al(d.bias, 8) || d.n > RS_BIASN)) return false;
        if (d.residual && !al(d.residual, 16)) return false;
        if (d.trans_rows) {
            if (d.trans_rows % 32 || d.m % d.trans_rows || d.stride_c % 8 || d.bias || d.residual) return false;
        } else if (d.ldc < d.n) {
            return false;
        }
    }
    // a tall activation: at least 64 rows per CU the launch may count on (K = 640: a quarter of the CUs get a 256-row tile and the
    // slice range is split to cover the rest; K = 320: half of them get a 128-row tile)
    return ignore_size || p0.m >= 64 * ncu;
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
    // whole slice pairs and at least eight slices (the resident rows are re-read once per part)
    const int k = p0.k;
    const int tiles = p0.m / (k == 640 ? 256 : 128);
    const int slots = k == 640 ? ncu : 2 * ncu;
    int nsplit = 1;
    for (int c = 2; c <= 8; ++c)
        if (tiles * c <= slots && ns % (2 * c) == 0 && ns / c >= 8) nsplit = c;
    p.nsplit = nsplit;
    if (k == 640) return dtype == AID_DTYPE_F16 ? rs_launch<f16, 640, 8>(p, stream) : rs_launch<bf16, 640, 8>(p, stream);
    return dtype == AID_DTYPE_F16 ? rs_launch<f16, 320, 4>(p, stream) : rs_launch<bf16, 320, 4>(p, stream);
}

}  // namespace aid
