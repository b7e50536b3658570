// Fused interpolated-attention core for gfx950 (CDNA4): one kernel for the PLAIN / INNER / OUTER
// modes of the AID processors, with optional fusion of the frame's own keys/values.
// Replaces, per attention layer call, the reference's end-point select + replicate + concat +
// baddbmm + softmax + bmm + batch_to_head_dim + lerp chain (interpolation.py:626-664 outer,
// 760-790 inner; de-activated fallback 581-584) — nothing of [N*H, S, 2L] is ever materialised.
//
// Work decomposition: one workgroup = NW waves x 32 query rows of one (frame, head); the grid is
// ordered [head][frame][q-block] and remapped XCD-aware so all q-blocks of a (frame, head) and the
// frames of one head (which share the two end-point K/V) hit the same per-XCD L2.
//
// Per 64-key tile and wave (MFMA 32x32x16, fp32 accumulate):
//   S^T[key, q]  = K_tile * Q^T      "swapped" product: lane (q = lane&31, half = lane>>5) owns 32
//                                    scores of ONE query row -> the row max / row sum of the online
//                                    softmax are in-lane reductions + one permlane32 swap.
//   O^T[dv, q]  += Vt_tile * P^T     V is consumed TRANSPOSED ([channel][key], produced that way by
//                                    the projection GEMM), so both MFMA operands are plain
//                                    ds_read_b128 row reads; the rescale factor of the online
//                                    softmax is lane-local because q is the lane index here too.
//   The K tile is read with key bits 2<->3 swapped so the P registers a lane holds after the first
//   product are exactly the 8 consecutive keys it must supply as B operand to the second one
//   (no cross-lane shuffle of P).
// The kernel is VALU-ISSUE bound, not MFMA bound (PMC: 168 VALU per 16 MFMA per wave-tile in the first
// version, profiles/r01_attn_notes.txt), so all the LINEAR arithmetic of the online softmax is pushed
// into the matrix pipe:
//   * Q arrives pre-multiplied by softmax_scale*log2(e) (folded into the q-projection GEMM epilogue, before
//     its single rounding), and the accumulator of the first product is INITIALISED to -m (the row's
//     reference, lane-local because a lane owns one query column): the MFMA result IS the exponent
//     argument x = s*c - m — no per-score FMA;
//   * the row sums come out of the second product: V^T carries a row of ones (a spare padded row for d = 40 /
//     80, one extra 32-row block fed from a constant register fragment for d = 64 / 160), so l is a row of
//     the O^T accumulator — no per-score add, and the rescale covers it automatically;
//   * the reference m is LAZY: it is set from the first tile and only raised when some exponent argument
//     of the wave exceeds the head-room of the storage type (2^15 for fp16, 2^60 for bf16); P = 2^x may exceed
//     1, which is exact in floating point.  The check is one v_max3 chain + a wave-uniform branch; the slow
//     path (rare) rescales O and shifts the tile's x in registers.
// What is left per score on the VALU is one v_exp_f32 and half a v_cvt_pk.
// K / Vt tiles go global -> registers -> LDS (issued before the tile's compute, written after it,
// two LDS buffers, one barrier per tile); the loads are branch-free (edge rows are clamped, never
// predicated).  LDS rows are padded by 16 B (odd 16-B stride): all fragment reads are bank-conflict
// free.
// INNER reads the interpolated keys/values of an interior frame from k2 / vt2, which the tiny
// element-wise kernel aid_lerp_kv_kernel below writes once per layer call: lerping the end-point
// tiles inside this kernel was measured at +18 % kernel time because every one of the S/128
// q-blocks of a (frame, head) repeated the same O(L*d) lerp.
// OUTER shares the own-keys segment between its two softmaxes (3 segment passes, not 4): the online
// state after the own segment is snapshotted and continued once with the begin and once with the
// end frame; frames with coefficient exactly 0 / 1 skip the zero-weighted side.
#include <stdlib.h>

#include <type_traits>


#include "aid_common.hpp"
#include "aid_kernels.hpp"

namespace aid {

constexpr int KT = 64;                  // keys per tile
constexpr int VLD = KT + 8;             // padded Vt tile row (elements)

struct AttnKParams {
    AidAttnArgs a;
    int32_t nqb;                        // q blocks per (frame, head)
    float   c2;                         // softmax_scale * log2(e)
    int32_t q_iters;                    // resident variant: q blocks a workgroup works through one after the other
    int32_t skip_single;                // 1: frames with ONE key segment are run by aid_attn_pp_kernel (launched next to this one)
};

// Resident variant (short key sets: the 77 text tokens of cross-attention, IP-Adapter image tokens): every key segment of the
// (frame, head) sits in LDS for the whole life of the workgroup — RES_KEYS key rows of K and V^T rows of RES_KEYS + 8 keys.
constexpr int RES_KEYS = 96;
constexpr int RES_VLD = RES_KEYS + 8;
constexpr int RES_SLACK = 64;           // zeroed elements behind the last segment: the masked half of a ragged tile reads on

__host__ __device__ constexpr bool attn_prefetch(int d, int nw) { return (nw == 4 || nw == 8) && d <= 80; }

// Online-softmax state of one wave: reference m (scaled log2 domain, per query = per lane), O^T blocks, and —
// when the head dim leaves no spare padded row — the extra block whose row 0 accumulates the row sums.
template <int NDB, bool XL, int QB>
struct OState {
    float  m[QB];
    bool   fresh;                       // no tile processed yet: the first tile sets the reference
    bool   mz;                          // LAZY0: the reference of every row of this wave is still exactly 0
    f32x16 o[QB][NDB];
    f32x16 ol[QB];                      // only used when XL
    f32x16 cn[QB];                      // -m in all 16 registers (C operand of the first MFMA), kept when PERSIST_C
};

typedef __amdgpu_buffer_rsrc_t Rsrc;
__device__ __forceinline__ Rsrc make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}

// QB = 32-row query blocks per wave.  QB = 2 (64 query rows per wave): every K / V^T fragment read from LDS and every
// staging pass / barrier serves twice the MFMAs, and the two blocks' independent MFMA chains give the scheduler
// something to put between dependent instructions — at the price of one wave per SIMD (> 256 VGPRs).
// PIPE = software-pipelined main loop (run_pipe below): the MFMAs of P V(t-1) and of Q K(t+1) are issued with the softmax
// VALU work of tile t in the gaps between them, order pinned by sched_barrier — one wave then overlaps its own matrix
// and vector work instead of running QK | softmax | PV back to back (tools/ubench/overlap.hip: 0.404 -> 0.341 us per
// wave-tile in the register-only model at 3 waves / SIMD, 0.560 -> 0.419 at one).
// RES = resident key segments (a.l <= RES_KEYS): the streaming kernel spends a 77-key launch waiting — three segments of two
// tiles, each a global -> register -> LDS round trip behind a barrier, for 128 query rows.  Here a workgroup loads the one to
// three segments of its (frame, head) ONCE (zero-padded to RES_KEYS keys), then every wave runs q block after q block out of
// LDS with no barrier at all; two workgroups per CU overlap one's fill with the other's arithmetic.
// BIAS = additive score bias (AidAttnArgs.bias: diffusers' attention_mask; the reference hands it to get_attention_scores of every
// segment, interpolation.py:651-656, 787): a separate instantiation of the four-wave program-order kernel only, so the kernels of
// the unmasked UNet calls carry neither its registers nor a branch.
template <typename T, int D, int MODE, int NW, int QB, bool PIPE, bool RES, bool BIAS = false>
__global__ __launch_bounds__(NW * 64) void aid_attn_kernel(const AttnKParams p) {
    typedef typename Vec<T>::v8 T8;
    typedef typename Vec<T>::v4 T4;
    constexpr int DK = (D + 15) / 16 * 16;      // contraction length of K Q^T (MFMA k = 16)
    constexpr int DV = (D + 31) / 32 * 32;      // rows of O^T (MFMA m = 32)
    constexpr int KLD = DK + 8;                 // padded K tile row (elements); KLD/8 is odd
    constexpr int NQK = DK / 16, NDB = DV / 32;
    constexpr int NT = NW * 64;
    constexpr int DC = D / 8;                   // 16-B chunks per K row
    constexpr int KCH = KT * DC, VCH = D * (KT / 8);    // chunks per K / Vt tile
    constexpr int NKC = (KCH + NT - 1) / NT;    // K chunks per thread per tile
    constexpr int NVC = (VCH + NT - 1) / NT;    // Vt chunks per thread per tile
    // Prefetch (issue tile t+1's loads before tile t's compute, two LDS buffers) only where the staging
    // registers fit beside the accumulators; otherwise stage synchronously through one buffer.
    constexpr bool PREFETCH = !RES && (attn_prefetch(D, NW) || QB > 1);
    constexpr int NBUF = PREFETCH ? 2 : 1;
    constexpr int VROW = RES ? RES_VLD : VLD;                   // V^T row stride in LDS (elements)
    constexpr int RSEG = RES_KEYS * KLD + DV * RES_VLD;         // elements of one resident segment: K rows, then V^T rows
    static_assert(!RES || (QB == 1 && !PIPE), "resident variant: one q block per wave, program-order tile");
    static_assert(!BIAS || (QB == 1 && !PIPE && !RES), "score bias: program-order streaming kernel, one q block per wave");
    static_assert((RES_VLD / 8) % 2 == 1 && RSEG % 8 == 0, "resident rows: odd number of 16-B slots");
    // row of ones in V^T -> the row sums l come out of the second MFMA as a row of O^T.  A spare padded row
    // (index D) exists for d = 40 / 80; d = 64 / 160 use one eXtra 32-row block fed from a constant fragment.
    constexpr bool XL = (D == DV);
    constexpr bool KPIPE = true;                // K fragment reads two k-steps ahead of their MFMAs (see tile())
    constexpr bool PERSIST_C = !RES && (QB > 1 || (D <= 64 && !(KPIPE && D == 64 && MODE == AID_MODE_PLAIN) &&
                                                   !(MODE == AID_MODE_OUTER && NW == 4) &&
                                                   !(D == 40 && MODE == AID_MODE_INNER && NW == 4 && !PIPE)));   // +16 VGPRs (outer: +32); d64 plain would drop to 2 waves/SIMD
    constexpr int LBLK = D / 32, LREG = ((D % 32) / 8) * 4;     // (block, register) of row D at lanes hi == 0
    static_assert(XL || ((D % 32) % 8 == 0 && (D % 32) < 32), "spare row must sit at a register boundary");
    // head-room (log2) of P = 2^x in the storage type before the row reference has to be raised
    constexpr float XTH = sizeof(T) == 2 && std::is_same<T, f16>::value ? 15.0f : 60.0f;
    // LAZY0 (experiment, off): bf16 has the exponent range of fp32, so the row reference could start at 0 and stay there
    // while every exponent argument is within +- 2^60 — the first product would then start from the constant 0 instead of
    // a copy of the -m block (32 v_mov per tile, a third of the kernel's VALU instructions: profiles/r02_pmc.json).
    // Measured: d64 plain 628 -> 639 us (no gain: the copies are not what the tile waits for), and the second code version
    // of the first product pushes INNER / OUTER past 256 VGPRs (one wave per SIMD: outer 1207 -> 1665 us).  Not adopted.
    constexpr bool LAZY0 = false;
    static_assert(D % 8 == 0, "head dim must be a multiple of 8");
    static_assert((KLD / 8) % 2 == 1 && (VLD / 8) % 2 == 1, "LDS row stride must be an odd number of 16-B slots");

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* Ks = reinterpret_cast<T*>(smem_raw);             // [NBUF][KT][KLD]
    T* Vs = Ks + NBUF * KT * KLD;                       // [NBUF][DV][VLD]

    const AidAttnArgs& a = p.a;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;

    const int n_heavy = a.n_frames - a.n_plain;                 // riders sit at the tail of the batch (hint, see heavy_first)
    const int lid = (MODE != AID_MODE_PLAIN && a.n_plain > 0 && n_heavy > 0)
                        ? heavy_first(blockIdx.x, gridDim.x, n_heavy * p.nqb, a.n_frames * p.nqb)
                        : xcd_remap(blockIdx.x, gridDim.x);
    const int qb = lid % p.nqb;
    const int fr = (lid / p.nqb) % a.n_frames;
    const int h = lid / (p.nqb * a.n_frames);
    int q0 = (qb * (RES ? p.q_iters : 1) * NW + wave) * 32 * QB;
    if (MODE != AID_MODE_PLAIN && p.skip_single) {      // the ping-pong kernel has this frame (same predicate there); before any barrier
        const int kv_ = a.kv_map ? a.kv_map[fr] : fr;
        const float c_ = a.coef[fr];
        if (c_ < 0.f || (a.fused && ((c_ == 0.f && kv_ == a.begin) || (c_ == 1.f && kv_ == a.end)))) return;
    }

    // zero both LDS buffers once where the head dim is padded (d = 40 / 80): pad columns of K / pad rows of V^T are
    // never staged and must be finite.  d = 64 / 160 have no padding that a fragment read touches.
    if (!RES && (DK != D || DV != D)) {
        for (int i = tid; i < NBUF * (KT * KLD + DV * VLD) / 8; i += NT)
            reinterpret_cast<T8*>(Ks)[i] = zero8<T>();
    }
    if (!RES && !XL) {                                  // the ones row (never touched by the staging, which writes rows < D)
        __syncthreads();
        for (int i = tid; i < NBUF * KT; i += NT) Vs[(i / KT) * DV * VLD + D * VLD + (i % KT)] = (T)1.0f;
    }

    // ---- Q fragments (B operand of the swapped product), straight from global -------------
    T8 qf[QB][NQK];
    auto load_q = [&](T8 (&qd)[QB][NQK], int qbase) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < QB; ++j) {
        const int qr = min(qbase + 32 * j + l31, a.s - 1);  // rows past the end are clamped, never stored
        const T* qrow = reinterpret_cast<const T*>(a.q) + (int64_t)fr * a.q_fs + (int64_t)qr * a.ldq + h * D;
#pragma unroll
        for (int ks = 0; ks < NQK; ++ks) {
            const int col = ks * 16 + hi * 8;
            qd[j][ks] = (col < D) ? *reinterpret_cast<const T8*>(qrow + col) : zero8<T>();
            if (!a.q_prescaled) {                       // generic callers: fold softmax_scale*log2(e) into Q here (one
                f32x8 t = up8<T>(qd[j][ks]);            // extra rounding; the processor path does it in the GEMM epilogue)
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] *= p.c2;
                qd[j][ks] = cvt8<T>(t);
            }
        }
    }
    };
    if (!RES || q0 < a.s) load_q(qf, q0);               // resident variant: the first q block's loads fly during the fill
    // constant A fragment of the row-sum block (XL): row 0 = ones, every other row zero
    T8 onesf;
#pragma unroll
    for (int e = 0; e < 8; ++e) onesf[e] = (l31 == 0) ? (T)1.0f : (T)0.0f;
    __syncthreads();

    const int kvf = a.kv_map ? a.kv_map[fr] : fr;
    const float cf = (MODE == AID_MODE_PLAIN || a.coef == nullptr) ? 0.f : a.coef[fr];
    const T* Kg = reinterpret_cast<const T*>(a.k) + h * D;
    const T* Vg = reinterpret_cast<const T*>(a.vt) + (int64_t)(h * D) * a.ldvt;
    const int L = a.l;
    const int Lc8 = ((L - 1) >> 3) << 3;                // first key of the last 8-key chunk holding a valid key
    // score-bias row of this lane's query (BIAS): element j belongs to key j of whichever segment is running
    const T* brow = nullptr;
    if (BIAS)
        brow = reinterpret_cast<const T*>(a.bias) + (int64_t)fr * a.bias_fs + (int64_t)h * a.bias_hs +
               (int64_t)min(q0 + l31, a.s - 1) * a.bias_rs;
    // key bits 2<->3 swapped: MFMA row i of the score block reads LDS key row pi(i)
    const int krow = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);

    // per-lane byte offsets of this thread's staging chunks inside a full K / Vt tile
    int kvo[NKC], vvo[NVC];
#pragma unroll
    for (int i = 0; i < NKC; ++i) {
        const int id = min(tid + i * NT, KCH - 1);
        kvo[i] = (id / DC) * (a.ldk * 2) + (id % DC) * 16;
    }
#pragma unroll
    for (int i = 0; i < NVC; ++i) {
        const int id = min(tid + i * NT, VCH - 1);
        vvo[i] = (id / (KT / 8)) * (a.ldvt * 2) + (id % (KT / 8)) * 16;
    }

    // ---- one segment of keys: online-softmax update of `st` ----------------------------------
    // k0/v0: frame base pointers (already offset to head h)
    // k0/v0: frame base pointers (already offset to head h); reg: LDS region of the segment (resident variant)
    auto run = [&](OState<NDB, XL, QB>& st, const T* k0, const T* v0, int reg) __attribute__((always_inline)) {
        T8 rk[PREFETCH ? NKC : 1], rv[PREFETCH ? NVC : 1];
        // buffer descriptors of the segment's K / Vt (wave-uniform); per-lane byte offsets are 32-bit and the
        // tile advance goes into the scalar offset, so a full tile costs no address VALU at all
        const Rsrc sk0 = make_rsrc(k0), sv0 = make_rsrc(v0);
        auto stage_load = [&](int key0, auto full_tag) __attribute__((always_inline)) {
            constexpr bool FULL = decltype(full_tag)::value;
            if (!PREFETCH) return;
#pragma unroll
            for (int i = 0; i < NKC; ++i) {
                int vo = kvo[i], so = key0 * a.ldk * 2;
                if (!FULL) {                            // ragged tile: clamp rows past the last key (never predicate)
                    const int id = min(tid + i * NT, KCH - 1);
                    vo = min(key0 + id / DC, L - 1) * (a.ldk * 2) + (id % DC) * 16;
                    so = 0;
                }
                rk[i] = __builtin_bit_cast(T8, __builtin_amdgcn_raw_buffer_load_b128(sk0, vo, so, 0));
            }
#pragma unroll
            for (int i = 0; i < NVC; ++i) {
                int vo = vvo[i], so = key0 * 2;
                if (!FULL) {
                    const int id = min(tid + i * NT, VCH - 1);
                    vo = (id / (KT / 8)) * (a.ldvt * 2) + min(key0 + (id % (KT / 8)) * 8, Lc8) * 2;
                    so = 0;
                }
                rv[i] = __builtin_bit_cast(T8, __builtin_amdgcn_raw_buffer_load_b128(sv0, vo, so, 0));
            }
        };
        auto stage_write = [&](int buf, int key0, auto full_tag) __attribute__((always_inline)) {
            constexpr bool WFULL = decltype(full_tag)::value;     // tile written has all 64 keys valid
            if (!PREFETCH) return;
            T* ks = Ks + buf * KT * KLD;
            T* vs = Vs + buf * DV * VLD;
#pragma unroll
            for (int i = 0; i < NKC; ++i) {
                const int id = tid + i * NT;
                if (NKC * NT != KCH && id >= KCH) continue;
                T8 v = rk[i];
                *reinterpret_cast<T8*>(ks + (id / DC) * KLD + (id % DC) * 8) = v;
            }
#pragma unroll
            for (int i = 0; i < NVC; ++i) {
                const int id = tid + i * NT;
                if (NVC * NT != VCH && id >= VCH) continue;
                T8 v = rv[i];
                if (!WFULL) {                           // keys >= L get P = 0; their V must be finite
                    asm volatile("; ragged Vt tile" ::: "memory");   // keeps hipcc from if-converting this into the hot path
                    const int kc = min(key0 + (id % (KT / 8)) * 8, Lc8);
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (kc + e >= L) v[e] = (T)0.0f;
                }
                *reinterpret_cast<T8*>(vs + (id / (KT / 8)) * VLD + (id % (KT / 8)) * 8) = v;
            }
        };

        // Non-prefetch variants (one wave per workgroup, or d = 160): a whole tile would be 40 x 16 B of staging
        // registers per lane, so it is staged synchronously in groups of 4 chunks (rolled loop, bounded registers).
        auto stage_sync = [&](int key0) __attribute__((always_inline)) {
            T* ks = Ks;
            T* vs = Vs;
            constexpr int G = 4;
#pragma nounroll
            for (int i0 = 0; i0 < NKC; i0 += G) {
                T8 r[G];
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    const int id = min(tid + (i0 + j) * NT, KCH - 1);
                    const int vo = min(key0 + id / DC, L - 1) * (a.ldk * 2) + (id % DC) * 16;
                    r[j] = __builtin_bit_cast(T8, __builtin_amdgcn_raw_buffer_load_b128(sk0, vo, 0, 0));
                }
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    const int id = tid + (i0 + j) * NT;
                    if (id < KCH) *reinterpret_cast<T8*>(ks + (id / DC) * KLD + (id % DC) * 8) = r[j];
                }
            }
#pragma nounroll
            for (int i0 = 0; i0 < NVC; i0 += G) {
                T8 r[G];
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    const int id = min(tid + (i0 + j) * NT, VCH - 1);
                    const int kc = min(key0 + (id % (KT / 8)) * 8, Lc8);
                    r[j] = __builtin_bit_cast(T8, __builtin_amdgcn_raw_buffer_load_b128(sv0, (id / (KT / 8)) * (a.ldvt * 2) + kc * 2, 0, 0));
                }
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    const int id = tid + (i0 + j) * NT;
                    if (id >= VCH) continue;
                    const int kc = min(key0 + (id % (KT / 8)) * 8, Lc8);
                    T8 v = r[j];
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (kc + e >= L) v[e] = (T)0.0f;         // keys >= L get P = 0; their V must be finite
                    *reinterpret_cast<T8*>(vs + (id / (KT / 8)) * VLD + (id % (KT / 8)) * 8) = v;
                }
            }
        };

        // ---- compute on one staged tile; FULL = all 64 keys valid ---------------------------------
        auto tile = [&](const T* ksrc, const T* vsrc, int key0, auto full_tag) __attribute__((always_inline)) {
            constexpr bool FULL = decltype(full_tag)::value;
            // A ragged tile runs the SAME straight-line MFMA sequence as a full one and masks afterwards: keys past L
            // have finite K rows / zero V^T columns in LDS.  (Skipping the second 32-key block with a branch saved four
            // MFMAs and cost correctness: on the taken path hipcc left too few wait states between the last MFMA of
            // block 0 and the first VALU read of its result — the score of keys 22 / 30 of the tile lacked its last
            // k-step whenever 16 < L % 64 <= 32.)
            // x^T = K Q'^T - m : the accumulator starts at -m, so the MFMA result is the exponent argument
            // (kept in registers across tiles where the budget allows: re-broadcasting it costs 16 v_mov per tile, a
            // sixth of the VALU instructions of a kernel that is VALU-issue bound — profiles/r01_attn_notes.txt)
            f32x16 cneg[QB];
#pragma unroll
            for (int j = 0; j < QB; ++j) {
                if (PERSIST_C) {
                    cneg[j] = st.cn[j];
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) cneg[j][r] = -st.m[j];
                }
            }
            f32x16 sc[QB][2];
            const T* kt = ksrc + krow * KLD + hi * 8;
            auto qk = [&](auto zero_tag) __attribute__((always_inline)) {
            constexpr bool ZERO = decltype(zero_tag)::value;
            // k-step outer, key-block inner: consecutive MFMAs go to DIFFERENT accumulators, so the dependent
            // chain of one block never stalls the matrix pipe (the block-outer order measured ~45 % of the tile time)
            if (FULL && KPIPE) {
                // K fragments two k-steps ahead of their MFMAs, order pinned: left to itself the compiler reuses ONE
                // fragment register for all 2 NQK reads, i.e. a full LDS round trip in front of every MFMA
                T8 kf[2][2];
                kf[0][0] = *reinterpret_cast<const T8*>(kt);
                kf[0][1] = *reinterpret_cast<const T8*>(kt + 32 * KLD);
                if (NQK > 1) {
                    kf[1][0] = *reinterpret_cast<const T8*>(kt + 16);
                    kf[1][1] = *reinterpret_cast<const T8*>(kt + 32 * KLD + 16);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < NQK; ++ks) {
#pragma unroll
                    for (int j = 0; j < QB; ++j) {
                        sc[j][0] = mfma32(kf[ks & 1][0], qf[j][ks], ks ? sc[j][0] : (ZERO ? zero16() : cneg[j]));
                        sc[j][1] = mfma32(kf[ks & 1][1], qf[j][ks], ks ? sc[j][1] : (ZERO ? zero16() : cneg[j]));
                    }
                    if (ks + 2 < NQK) {
                        kf[ks & 1][0] = *reinterpret_cast<const T8*>(kt + (ks + 2) * 16);
                        kf[ks & 1][1] = *reinterpret_cast<const T8*>(kt + 32 * KLD + (ks + 2) * 16);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < NQK; ++ks) {
                    const T8 k0f = *reinterpret_cast<const T8*>(kt + ks * 16);
                    const T8 k1f = *reinterpret_cast<const T8*>(kt + 32 * KLD + ks * 16);
#pragma unroll
                    for (int j = 0; j < QB; ++j) {
                        sc[j][0] = mfma32(k0f, qf[j][ks], ks ? sc[j][0] : (ZERO ? zero16() : cneg[j]));
                        sc[j][1] = mfma32(k1f, qf[j][ks], ks ? sc[j][1] : (ZERO ? zero16() : cneg[j]));
                    }
                }
            }
            };
            if (LAZY0 && st.mz) qk(std::true_type{});
            else                qk(std::false_type{});
            // lane (q, hi): sc[j][b][r] belongs to key  key0 + 32 b + 16 (r>>3) + 8 hi + (r&7)
            if (BIAS) {
                // scores = scale q k^T + bias (baddbmm(attention_mask, q, k^T, beta = 1, alpha = scale)); the kernel works in the log2
                // domain.  Values below -1e30 (-inf, finfo.min of bf16) are clamped there: the key then weighs exp(-1e30) = 0 like in
                // the reference, and a row whose keys are ALL masked averages them uniformly (the reference: uniform for finfo.min, NaN
                // for -inf).  2-byte loads: a mask row (L elements, any L) has no alignment to offer; correctness-first path.
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {               // eight loads in flight at a time (d = 160 has no registers for 32)
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int key = min(key0 + 32 * b + 16 * u + 8 * hi + e, L - 1);
                            sc[0][b][8 * u + e] += fmaxf((float)brow[key], -1e30f) * 1.4426950408889634f;
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
            }
            if (!FULL) {
#pragma unroll
                for (int j = 0; j < QB; ++j)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (key0 + 32 * b + 16 * (r >> 3) + 8 * hi + (r & 7) >= L) sc[j][b][r] = -1e30f;
            }
            // head-room check: one v_max3 chain over this lane's 32 arguments per query block, wave-uniform decision
            float xm[QB];
#pragma unroll
            for (int j = 0; j < QB; ++j) {
                xm[j] = fmaxf(sc[j][0][0], sc[j][0][1]);
#pragma unroll
                for (int i = 1; i < 16; ++i)
                    xm[j] = fmaxf(fmaxf(xm[j], sc[j][i >> 3][(2 * i) & 15]), sc[j][i >> 3][(2 * i + 1) & 15]);
            }
            float xall = xm[0];
#pragma unroll
            for (int j = 1; j < QB; ++j) xall = fmaxf(xall, xm[j]);
            bool need = __any(xall > XTH);
            if (st.fresh) {
                if (!LAZY0) {
                    need = true;
                } else {                                  // reference 0 is fine unless a whole row sits below the head-room
#pragma unroll
                    for (int j = 0; j < QB; ++j) need = need || __any(max_halves(xm[j]) < -XTH);
                    if (!need) st.fresh = false;
                }
            }
            if (__builtin_expect(need, 0)) {                    // (out of line: the fast path falls through)
                // slow path (first tile of a row, or a score out-grew the head-room): move the reference to the
                // row maximum, rescale O (its ones-row = l included) and shift this tile's arguments in registers
                st.mz = false;
#pragma unroll
                for (int j = 0; j < QB; ++j) {
                    const float rowmax = max_halves(xm[j]);
                    const float shift = st.fresh ? rowmax : fmaxf(rowmax, 0.f);
                    const float alpha = __builtin_amdgcn_exp2f(-shift);
                    st.m[j] += shift;
                    if (PERSIST_C) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) st.cn[j][r] = -st.m[j];
                        asm volatile("" : "+v"(st.cn[j]));     // opaque: keeps the compiler from re-deriving it from m per tile
                    }
#pragma unroll
                    for (int d = 0; d < NDB; ++d)
#pragma unroll
                        for (int r = 0; r < 16; ++r) st.o[j][d][r] *= alpha;
                    if (XL) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) st.ol[j][r] *= alpha;
                    }
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r) sc[j][b][r] -= shift;
                }
                st.fresh = false;
            }
            T8 pf[QB][4];
#pragma unroll
            for (int j = 0; j < QB; ++j)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        f32x8 pv;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            pv[e] = __builtin_amdgcn_exp2f(sc[j][b][8 * u + e]);
                        }
                        pf[j][2 * b + u] = cvt8<T>(pv);
                    }
            // O^T += Vt P^T   (the ones row / ones block accumulates the row sums); a V^T fragment read serves all QB blocks
            const T* vt = vsrc + l31 * VROW + hi * 8;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                for (int d = 0; d < NDB; ++d) {
                    const T8 vf = *reinterpret_cast<const T8*>(vt + d * 32 * VROW + kk * 16);
#pragma unroll
                    for (int j = 0; j < QB; ++j) st.o[j][d] = mfma32(vf, pf[j][kk], st.o[j][d]);
                }
                if (XL) {
#pragma unroll
                    for (int j = 0; j < QB; ++j) st.ol[j] = mfma32(onesf, pf[j][kk], st.ol[j]);
                }
            }
        };


        // ---- software-pipelined main loop over `nfp` FULL tiles (nfp odd, >= 3) ---------------------------------
        // Tile t:  S(t) = K(t) Q'^T - m  (QK),  P(t) = 2^S(t)  (softmax VALU),  O^T += V^T(t) P(t)^T  (PV).
        // Iteration i issues   phase A: the MFMAs of PV(i-1) with max3(S(i)) and the exponentials of key block 0 of
        //                               S(i) (in place) between them;
        //                      [rare] head-room decision for tile i: PV(i-1) is complete, so rescaling O here is safe;
        //                      phase B: the MFMAs of QK(i+1) with the cvt of block 0, exp + cvt of block 1 -> P(i) and the
        //                               LDS writes of the staged tiles between them.
        // S(t) lives in sA / sB by the parity of t, P(t) in pA / pB; K(t) / V^T(t) in LDS buffer t & 1.  K runs one tile
        // further ahead than V^T: iteration i stages K(i+2) and V^T(i) (loads issued at its start, written in phase B),
        // one barrier per iteration.
        auto run_pipe = [&](int nfp) __attribute__((always_inline)) {
            static_assert(!PIPE || PREFETCH, "pipelined loop: register staging");
            constexpr int NDX = NDB + (XL ? 1 : 0);             // O^T blocks incl. the row-sum block
            constexpr int NPV = 4 * NDX * QB;                   // MFMAs of one PV (all query blocks of the wave)
            constexpr int NQM = 2 * NQK * QB;                   // MFMAs of one QK
            constexpr int NOP = 32 * QB;                        // VALU micro-ops per phase
            auto ld_k = [&](int t_) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < NKC; ++i)
                    rk[PREFETCH ? i : 0] = __builtin_bit_cast(T8, __builtin_amdgcn_raw_buffer_load_b128(sk0, kvo[i], t_ * KT * a.ldk * 2, 0));
            };
            auto ld_v = [&](int t_) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < NVC; ++i)
                    rv[PREFETCH ? i : 0] = __builtin_bit_cast(T8, __builtin_amdgcn_raw_buffer_load_b128(sv0, vvo[i], t_ * KT * 2, 0));
            };
            auto wr_k = [&](int buf, int i) __attribute__((always_inline)) {      // chunk i of the staged K tile -> LDS
                const int id = tid + i * NT;
                if (NKC * NT != KCH && id >= KCH) return;
                *reinterpret_cast<T8*>(Ks + buf * KT * KLD + (id / DC) * KLD + (id % DC) * 8) = rk[PREFETCH ? i : 0];
            };
            auto wr_v = [&](int buf, int i) __attribute__((always_inline)) {
                const int id = tid + i * NT;
                if (NVC * NT != VCH && id >= VCH) return;
                *reinterpret_cast<T8*>(Vs + buf * DV * VLD + (id / (KT / 8)) * VLD + (id % (KT / 8)) * 8) = rv[PREFETCH ? i : 0];
            };
            auto cneg_of = [&](int j) __attribute__((always_inline)) {
                f32x16 c;
                if (PERSIST_C) {
                    c = st.cn[j];
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) c[r] = -st.m[j];
                }
                return c;
            };
            typedef f32x16 Sc[QB][2];
            typedef T8 Pf[QB][4];
            // QK of the tile in LDS buffer `buf`, program order (prologue)
            auto qk_plain = [&](int buf, Sc& s_) __attribute__((always_inline)) {
                const T* kt = Ks + buf * KT * KLD + krow * KLD + hi * 8;
#pragma unroll
                for (int ks = 0; ks < NQK; ++ks) {
                    const T8 f0 = *reinterpret_cast<const T8*>(kt + ks * 16);
                    const T8 f1 = *reinterpret_cast<const T8*>(kt + 32 * KLD + ks * 16);
#pragma unroll
                    for (int j = 0; j < QB; ++j) {
                        s_[j][0] = mfma32(f0, qf[j][ks], ks ? s_[j][0] : cneg_of(j));
                        s_[j][1] = mfma32(f1, qf[j][ks], ks ? s_[j][1] : cneg_of(j));
                    }
                }
            };
            // VALU micro-ops, the wave's query blocks interleaved (op k belongs to block k % QB).  Phase A, step u = k / QB:
            // 0..15 = the max3 chain over S, 16..31 = exp of S[0][u-16] in place.
            // the empty asm pins the op where it is written: without it LLVM sinks the (pure) chain to its first use
            // behind the MFMAs and the interleave is gone
            auto op_a = [&](int k, Sc& s_, float (&mx)[QB]) __attribute__((always_inline)) {
                const int j = k % QB, u = k / QB;
                if (u < 16) {
                    if (u == 0) mx[j] = fmaxf(s_[j][0][0], s_[j][0][1]);
                    else        mx[j] = fmaxf(fmaxf(mx[j], s_[j][u >> 3][(2 * u) & 15]), s_[j][u >> 3][(2 * u + 1) & 15]);
                    asm volatile("" : "+v"(mx[j]));
                } else {
                    float e = __builtin_amdgcn_exp2f(s_[j][0][u - 16]);
                    asm volatile("" : "+v"(e));
                    s_[j][0][u - 16] = e;
                }
            };
            // Phase B, step u: 0..7 = cvt_pk of block 0 (exponentiated in phase A) -> P[0..1]; 8..23 = exp of S[1][u-8];
            // 24..31 = cvt_pk of block 1 -> P[2..3]
            auto op_b = [&](int k, Sc& s_, Pf& p_) __attribute__((always_inline)) {
                typedef typename Vec<T>::v2 T2;
                const int j = k % QB, u = k / QB;
                if (u < 8 || u >= 24) {
                    const int b = u < 8 ? 0 : 1, w = u < 8 ? u : u - 24;          // pair w of key block b: elements 2w, 2w+1
                    f32x2 v2;
                    v2[0] = s_[j][b][2 * w];
                    v2[1] = s_[j][b][2 * w + 1];
                    T2 c2 = __builtin_convertvector(v2, T2);
                    asm volatile("" : "+v"(c2));
                    p_[j][2 * b + (w >> 2)][2 * (w & 3)] = c2[0];
                    p_[j][2 * b + (w >> 2)][2 * (w & 3) + 1] = c2[1];
                } else {
                    float e = __builtin_amdgcn_exp2f(s_[j][1][u - 8]);
                    asm volatile("" : "+v"(e));
                    s_[j][1][u - 8] = e;
                }
            };
            // S of tile t straight from global memory in fragment layout (slow path only: the LDS buffer that held K(t) may
            // already be receiving K(t+2) from a faster wave)
            auto qk_global = [&](int t_, Sc& s_) __attribute__((always_inline)) {
                const T* kg = k0 + (int64_t)(t_ * KT + krow) * a.ldk + hi * 8;
#pragma unroll
                for (int ks = 0; ks < NQK; ++ks) {
                    const bool in = ks * 16 + hi * 8 < D;
                    const T8 f0 = in ? *reinterpret_cast<const T8*>(kg + ks * 16) : zero8<T>();
                    const T8 f1 = in ? *reinterpret_cast<const T8*>(kg + (int64_t)32 * a.ldk + ks * 16) : zero8<T>();
#pragma unroll
                    for (int j = 0; j < QB; ++j) {
                        s_[j][0] = mfma32(f0, qf[j][ks], ks ? s_[j][0] : cneg_of(j));
                        s_[j][1] = mfma32(f1, qf[j][ks], ks ? s_[j][1] : cneg_of(j));
                    }
                }
            };
            // head-room slow path for tile i (first tile of the wave, or a score out-grew the storage type's head-room): S(i)
            // was partly exponentiated in place, so it is recomputed, the reference is moved to the row maximum, O is
            // rescaled (PV(i-1) is complete) and block 0 is exponentiated again
            auto slow = [&](int i, Sc& s_) __attribute__((always_inline)) {
                qk_global(i, s_);
#pragma unroll
                for (int j = 0; j < QB; ++j) {
                    float xm = fmaxf(s_[j][0][0], s_[j][0][1]);
#pragma unroll
                    for (int k = 1; k < 16; ++k) xm = fmaxf(fmaxf(xm, s_[j][k >> 3][(2 * k) & 15]), s_[j][k >> 3][(2 * k + 1) & 15]);
                    const float rowmax = max_halves(xm);
                    const float shift = st.fresh ? rowmax : fmaxf(rowmax, 0.f);
                    const float alpha = __builtin_amdgcn_exp2f(-shift);
                    st.m[j] += shift;
                    if (PERSIST_C) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) st.cn[j][r] = -st.m[j];
                        asm volatile("" : "+v"(st.cn[j]));
                    }
#pragma unroll
                    for (int d = 0; d < NDB; ++d)
#pragma unroll
                        for (int r = 0; r < 16; ++r) st.o[j][d][r] *= alpha;
                    if (XL) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) st.ol[j][r] *= alpha;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        s_[j][0][r] = __builtin_amdgcn_exp2f(s_[j][0][r] - shift);
                        s_[j][1][r] -= shift;
                    }
                }
                st.fresh = false;
            };
            // one iteration; HAS_PV: PV(i-1) exists, HAS_KLD: K(i+2) exists, HAS_QK: tile i+1 exists
            auto iter = [&](auto has_pv_t, auto has_kld_t, auto has_qk_t, int i, Sc& s_cur, Sc& s_nxt, Pf& p_prev,
                            Pf& p_cur) __attribute__((always_inline)) {
                constexpr bool HAS_PV = decltype(has_pv_t)::value, HAS_KLD = decltype(has_kld_t)::value,
                               HAS_QK = decltype(has_qk_t)::value;
                const int bc = i & 1;
                if (HAS_KLD) ld_k(i + 2);
                ld_v(i);
                // ---------------- phase A ----------------
                float mx[QB];
#pragma unroll
                for (int j = 0; j < QB; ++j) mx[j] = 0.f;
                if (HAS_PV) {
                    const T* vt = Vs + (bc ^ 1) * DV * VLD + l31 * VLD + hi * 8;
                    T8 vf[2][NDB];
#pragma unroll
                    for (int d = 0; d < NDB; ++d) vf[0][d] = *reinterpret_cast<const T8*>(vt + d * 32 * VLD);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        if (kk + 1 < 4) {
#pragma unroll
                            for (int d = 0; d < NDB; ++d)
                                vf[(kk + 1) & 1][d] = *reinterpret_cast<const T8*>(vt + d * 32 * VLD + (kk + 1) * 16);
                        }
#pragma unroll
                        for (int d = 0; d < NDX; ++d) {
#pragma unroll
                            for (int j = 0; j < QB; ++j) {
                                if (d < NDB) st.o[j][d] = mfma32(vf[kk & 1][d], p_prev[j][kk], st.o[j][d]);
                                else         st.ol[j] = mfma32(onesf, p_prev[j][kk], st.ol[j]);
                                __builtin_amdgcn_sched_barrier(0);
                                const int g = (kk * NDX + d) * QB + j;
#pragma unroll
                                for (int k = g * NOP / NPV; k < (g + 1) * NOP / NPV; ++k) op_a(k, s_cur, mx);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < NOP; ++k) op_a(k, s_cur, mx);
                }
                float mall = mx[0];
#pragma unroll
                for (int j = 1; j < QB; ++j) mall = fmaxf(mall, mx[j]);
                if (__builtin_expect(st.fresh || __any(mall > XTH), 0)) slow(i, s_cur);
                __builtin_amdgcn_sched_barrier(0);
                // ---------------- phase B ----------------
                if (HAS_QK) {
                    f32x16 cneg[QB];
#pragma unroll
                    for (int j = 0; j < QB; ++j) cneg[j] = cneg_of(j);
                    const T* kt = Ks + (bc ^ 1) * KT * KLD + krow * KLD + hi * 8;
                    T8 kf[2][2];
                    kf[0][0] = *reinterpret_cast<const T8*>(kt);
                    kf[0][1] = *reinterpret_cast<const T8*>(kt + 32 * KLD);
                    if (NQK > 1) {
                        kf[1][0] = *reinterpret_cast<const T8*>(kt + 16);
                        kf[1][1] = *reinterpret_cast<const T8*>(kt + 32 * KLD + 16);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ks = 0; ks < NQK; ++ks) {
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
#pragma unroll
                            for (int j = 0; j < QB; ++j) {
                                s_nxt[j][b] = mfma32(kf[ks & 1][b], qf[j][ks], ks ? s_nxt[j][b] : cneg[j]);
                                __builtin_amdgcn_sched_barrier(0);
                                const int g = (2 * ks + b) * QB + j;
#pragma unroll
                                for (int k = g * NOP / NQM; k < (g + 1) * NOP / NQM; ++k) op_b(k, s_cur, p_cur);
                                // the staged tiles go to LDS in the last gaps (their loads were issued a phase and a half ago)
                                if (g >= NQM - NKC - NVC) {
                                    const int w = g - (NQM - NKC - NVC);
                                    if (w < NKC) { if (HAS_KLD) wr_k(bc, w); }
                                    else         wr_v(bc, w - NKC);
                                }
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                        if (ks + 2 < NQK) {
                            kf[ks & 1][0] = *reinterpret_cast<const T8*>(kt + (ks + 2) * 16);
                            kf[ks & 1][1] = *reinterpret_cast<const T8*>(kt + 32 * KLD + (ks + 2) * 16);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < NOP; ++k) op_b(k, s_cur, p_cur);
#pragma unroll
                    for (int w = 0; w < NVC; ++w) wr_v(bc, w);
                }
                __syncthreads();
            };
            static_assert(NKC + NVC <= 2 * NQK, "staging writes must fit the gaps of phase B");

            Sc sA, sB;
            Pf pA, pB;
            // prologue: K(0), K(1) to LDS, S(0)
            ld_k(0);
#pragma unroll
            for (int w = 0; w < NKC; ++w) wr_k(0, w);
            ld_k(1);
#pragma unroll
            for (int w = 0; w < NKC; ++w) wr_k(1, w);
            __syncthreads();
            qk_plain(0, sA);
            const std::true_type Y{};
            const std::false_type N{};
            iter(N, Y, Y, 0, sA, sB, pB, pA);                   // i = 0: S(0) -> P(0) in pA, S(1) in sB
            int i = 1;
            for (; i + 1 <= nfp - 3; i += 2) {                  // steady state, two iterations per trip (register sets swap)
                iter(Y, Y, Y, i, sB, sA, pA, pB);
                iter(Y, Y, Y, i + 1, sA, sB, pB, pA);
            }
            iter(Y, N, Y, nfp - 2, sB, sA, pA, pB);             // odd: no K tile left to stage
            iter(Y, N, N, nfp - 1, sA, sB, pB, pA);             // even: last tile, no QK
            {                                                   // PV(nfp - 1): P in pA, V^T in buffer 0
                const T* vt = Vs + l31 * VLD + hi * 8;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                    for (int d = 0; d < NDB; ++d) {
                        const T8 vf = *reinterpret_cast<const T8*>(vt + d * 32 * VLD + kk * 16);
#pragma unroll
                        for (int j = 0; j < QB; ++j) st.o[j][d] = mfma32(vf, pA[j][kk], st.o[j][d]);
                    }
                    if (XL) {
#pragma unroll
                        for (int j = 0; j < QB; ++j) st.ol[j] = mfma32(onesf, pA[j][kk], st.ol[j]);
                    }
                }
            }
            __syncthreads();
        };

        const int nt = (L + KT - 1) / KT;
        const int nfull = L / KT;
        int t = 0;
        if (RES) {                                      // at most one full and one ragged tile, both already in LDS
            const T* kb = Ks + reg * RSEG;
            const T* vb = kb + RES_KEYS * KLD;
            if (nfull) tile(kb, vb, 0, std::true_type{});
            if (nt > nfull) tile(kb + nfull * KT * KLD, vb + nfull * KT, nfull * KT, std::false_type{});
            return;
        }
        if (PIPE && nfull >= 3) {                       // pipelined over an odd number of full tiles, the rest below
            const int nfp = (nfull & 1) ? nfull : nfull - 1;
            run_pipe(nfp);
            t = nfp;
            if (t == nt) return;
        }
        if (PREFETCH) {
            if (t < nfull) stage_load(t * KT, std::true_type{});
            else           stage_load(t * KT, std::false_type{});
            if (t < nfull) stage_write(t & 1, t * KT, std::true_type{});
            else           stage_write(t & 1, t * KT, std::false_type{});
            __syncthreads();
        }
        for (; t < nfull; ++t) {                        // tiles with all 64 keys valid: no masks, no edge logic
            const int buf = PREFETCH ? (t & 1) : 0, key0 = t * KT;
            if (PREFETCH) {
                if (t + 1 < nfull)   stage_load(key0 + KT, std::true_type{});
                else if (t + 1 < nt) stage_load(key0 + KT, std::false_type{});
            } else {
                stage_sync(key0);
                __syncthreads();
            }
            tile(Ks + buf * KT * KLD, Vs + buf * DV * VLD, key0, std::true_type{});
            if (PREFETCH) {
                if (t + 1 < nfull)   stage_write(buf ^ 1, key0 + KT, std::true_type{});
                else if (t + 1 < nt) stage_write(buf ^ 1, key0 + KT, std::false_type{});
            }
            __syncthreads();
        }
        if (t < nt) {                                   // ragged last tile
            const int buf = PREFETCH ? (t & 1) : 0, key0 = t * KT;
            if (!PREFETCH) {
                stage_sync(key0);
                __syncthreads();
            }
            tile(Ks + buf * KT * KLD, Vs + buf * DV * VLD, key0, std::false_type{});
            __syncthreads();
        }
    };

    typedef OState<NDB, XL, QB> State;
    auto init = [&](State& st) {
        st.fresh = true;
        st.mz = true;
#pragma unroll
        for (int j = 0; j < QB; ++j) {
            st.m[j] = 0.f;
            st.cn[j] = zero16();
            if (PERSIST_C) asm volatile("" : "+v"(st.cn[j]));
#pragma unroll
            for (int d = 0; d < NDB; ++d) st.o[j][d] = zero16();
            st.ol[j] = zero16();
        }
    };
    // Fence between the last MFMAs of a segment and the VALU code that reads the accumulators (finish): the segment
    // loops end in branches (`is there a ragged tile`), and on a taken path hipcc has left too few wait states between an
    // MFMA and the first VALU read of its result before (the ragged-tile hazard above).  20 wait states per q block.
    auto settle = [&](State& st) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < QB; ++j) {
#pragma unroll
            for (int d = 0; d < NDB; ++d) asm volatile("" : "+v"(st.o[j][d]));
            if (XL) asm volatile("" : "+v"(st.ol[j]));
        }
        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
#pragma unroll
        for (int j = 0; j < QB; ++j) {
#pragma unroll
            for (int d = 0; d < NDB; ++d) asm volatile("" : "+v"(st.o[j][d]));
            if (XL) asm volatile("" : "+v"(st.ol[j]));
        }
    };
    // 1 / (row sum) of a finished state: the sum sits in the ones-row of O^T at the lanes of half 0
    auto inv_l = [&](const State& st, int j) __attribute__((always_inline)) {
        const float lv = XL ? st.ol[j][0] : st.o[j][LBLK][LREG];
        const float partner = other_half(lv);          // executed by every lane (cross-half permute)
        return 1.f / (hi ? partner : lv);
    };
    // res (+)= w * O / l  for every query block of the wave
    auto finish = [&](f32x16 (&res)[QB][NDB], const State& st, float w, bool add) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < QB; ++j) {
            const float sc_ = w * inv_l(st, j);
#pragma unroll
            for (int d = 0; d < NDB; ++d) {
                if (add) res[j][d] += st.o[j][d] * sc_;
                else     res[j][d] = st.o[j][d] * sc_;
            }
        }
    };

    const T* k_own = Kg + (int64_t)kvf * a.k_fs;
    const T* v_own = Vg + (int64_t)kvf * a.vt_fs;
    const T* k_beg = Kg + (int64_t)a.begin * a.k_fs;
    const T* v_beg = Vg + (int64_t)a.begin * a.vt_fs;
    const T* k_end = Kg + (int64_t)a.end * a.k_fs;
    const T* v_end = Vg + (int64_t)a.end * a.vt_fs;

    // Which segments this frame runs (the same decisions for every q block of the workgroup):
    //  single — (1) PLAIN; (2) negative coefficient = this frame is PLAIN inside an INNER / OUTER launch (the unconditional
    //           half of a classifier-free-guidance batch rides in the same call); (3) a fused END-POINT frame: its second
    //           segment would be its own keys again ([K_0 ; K_0]) — duplicating every key leaves softmax(QK^T)V unchanged
    //           (SURVEY.md §4 invariant), so one pass over the own keys is the same result with half the work.
    //  INNER  — coefficient exactly 0 / 1: the lerp is the end-point frame itself; interior frames read the interpolated
    //           keys / values that aid_lerp_kv wrote to k2 / vt2.
    const bool single = MODE == AID_MODE_PLAIN || cf < 0.f ||
                        (a.fused && ((cf == 0.f && kvf == a.begin) || (cf == 1.f && kvf == a.end)));
    const bool own_first = single || a.fused;
    const T* k_mix = k_beg;
    const T* v_mix = v_beg;
    if (MODE == AID_MODE_INNER) {
        if (cf == 1.f) { k_mix = k_end; v_mix = v_end; }
        else if (cf != 0.f) {
            k_mix = reinterpret_cast<const T*>(a.k2) + h * D + (int64_t)fr * a.k_fs;
            v_mix = reinterpret_cast<const T*>(a.vt2) + (int64_t)(h * D) * a.ldvt + (int64_t)fr * a.vt_fs;
        }
    }
    const int r_own = 0, r_mix = own_first ? 1 : 0;             // LDS regions (resident variant)
    const int r_beg = r_mix, r_end = r_mix + (cf != 1.f ? 1 : 0);

    if (RES) {
        // ---- fill: every segment this frame needs, once per workgroup ------------------------------------------------
        // One pass writes EVERY 16-B chunk of every region exactly once — key / value data, zeros (key rows and V^T
        // columns past L, pad columns, unused regions: the masked half of a ragged tile reads them), the ones row —
        // with all global loads of the pass in flight before the first LDS write, and one barrier.
        T* const L0 = reinterpret_cast<T*>(smem_raw);
        constexpr int NREG = MODE == AID_MODE_PLAIN ? 1 : MODE == AID_MODE_INNER ? 2 : 3;
        constexpr int KC8 = KLD / 8, VC8 = RES_VLD / 8;             // chunks per K row / V^T row
        constexpr int KCH8 = RES_KEYS * KC8, NCH = RSEG / 8;        // chunks of the K part / of a whole region
        constexpr int NIT = (NCH + NT - 1) / NT;
        const bool both = cf != 0.f && cf != 1.f;
        const T* rk[3] = {own_first ? k_own : (MODE == AID_MODE_INNER ? k_mix : (cf != 1.f ? k_beg : k_end)),
                          own_first ? (single ? nullptr : (MODE == AID_MODE_INNER ? k_mix : (cf != 1.f ? k_beg : k_end)))
                                    : (MODE == AID_MODE_OUTER && both ? k_end : nullptr),
                          (MODE == AID_MODE_OUTER && own_first && !single && both) ? k_end : nullptr};
        const T* rv[3] = {own_first ? v_own : (MODE == AID_MODE_INNER ? v_mix : (cf != 1.f ? v_beg : v_end)),
                          own_first ? (single ? nullptr : (MODE == AID_MODE_INNER ? v_mix : (cf != 1.f ? v_beg : v_end)))
                                    : (MODE == AID_MODE_OUTER && both ? v_end : nullptr),
                          (MODE == AID_MODE_OUTER && own_first && !single && both) ? v_end : nullptr};
        const int nk8 = (L + 7) & ~7;
        T8 one8;
#pragma unroll
        for (int e = 0; e < 8; ++e) one8[e] = (T)1.0f;
        T8 stg[NREG][NIT];
#pragma unroll
        for (int r = 0; r < NREG; ++r) {
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                const int idx = tid + i * NT;
                T8 v = zero8<T>();
                if (idx < KCH8) {
                    const int row = idx / KC8, c = idx % KC8;
                    if (rk[r] != nullptr && row < L && c < DC)
                        v = *reinterpret_cast<const T8*>(rk[r] + (int64_t)row * a.ldk + c * 8);
                } else if (idx < NCH) {
                    const int row = (idx - KCH8) / VC8, kc = ((idx - KCH8) % VC8) * 8;
                    if (rv[r] != nullptr && row < D && kc < nk8) {
                        v = *reinterpret_cast<const T8*>(rv[r] + (int64_t)row * a.ldvt + kc);
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (kc + e >= L) v[e] = (T)0.0f;        // keys >= L get P = 0; their V must be finite
                    } else if (!XL && row == D) {
                        v = one8;                                   // the ones row: row sums out of the second product
                    }
                }
                stg[r][i] = v;
            }
        }
#pragma unroll
        for (int r = 0; r < NREG; ++r)
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                const int idx = tid + i * NT;
                if (idx < NCH) reinterpret_cast<T8*>(L0 + r * RSEG)[idx] = stg[r][i];
            }
        if (tid < RES_SLACK / 8) reinterpret_cast<T8*>(L0 + NREG * RSEG)[tid] = zero8<T>();
        __syncthreads();
    }

    const int n_it = RES ? p.q_iters : 1;
    for (int it = 0; it < n_it; ++it, q0 += NW * 32 * QB) {
    T8 qn[QB][NQK];                                     // resident variant: next q block's fragments, loaded a block ahead
    if (RES) {
        if (q0 >= a.s) break;                           // no barrier below: a wave may leave on its own
        const int qnx = q0 + NW * 32 * QB;
        if (it + 1 < n_it && qnx < a.s) load_q(qn, qnx);
    }
    State st;
    init(st);
    f32x16 res[QB][NDB];

    if (single) {
        run(st, k_own, v_own, r_own);
        settle(st);
        finish(res, st, 1.f, false);
    } else if (MODE == AID_MODE_INNER) {
        if (a.fused) run(st, k_own, v_own, r_own);
        run(st, k_mix, v_mix, r_mix);
        settle(st);
        finish(res, st, 1.f, false);
    } else {
        if (a.fused) {
            run(st, k_own, v_own, r_own);
            settle(st);                                     // the snapshot below copies the accumulators on the VALU
        }
#pragma unroll
        for (int j = 0; j < QB; ++j)
#pragma unroll
            for (int d = 0; d < NDB; ++d) res[j][d] = zero16();
        if (cf != 1.f) {                                    // begin side, weight (1 - c)
            State sb = st;
            run(sb, k_beg, v_beg, r_beg);
            settle(sb);
            finish(res, sb, 1.f - cf, false);
        }
        if (cf != 0.f) {                                    // end side, weight c
            run(st, k_end, v_end, r_end);
            settle(st);
            finish(res, st, cf, true);
        }
    }

    // ---- epilogue: lane (q = l31, hi) holds dv = 32 d + 8 g + 4 hi + {0..3} ------------------------
    // Streaming variants with one q block per wave: the wave's 32 x D block leaves through LDS (the tile buffers, free behind a
    // barrier) as 16-byte stores covering whole rows of the head — 8-byte stores per lane at a row stride are store-ISSUE bound (the
    // guide's T21; profiles/r04_attn_notes.txt): the 77-key launches, little more than a store tail, -9 ... -18 %.
    constexpr bool STAGED = !RES && QB == 1 && D % 8 == 0 && (size_t)NBUF * (KT * KLD + DV * VLD) * sizeof(T) >= (size_t)NW * 32 * D * sizeof(T);
    // (an accumulating launch — the IP-Adapter image branch — takes the direct path: there the old output is added in fp32 and the sum
    // rounded ONCE, whatever the alignment; the staged path would round the new term first: one numeric behaviour)
    const bool staged = STAGED && !a.accumulate && a.ldo % 8 == 0 && a.o_fs % 8 == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0 &&
                        (int64_t)a.n_frames * a.o_fs * 2 < (1ll << 31);
    if (STAGED && staged) {
        __syncthreads();                                        // every wave is done with the K / V^T tiles
        if (q0 < a.s) {
            constexpr int RBY = D * 2, CPR = D / 8;             // bytes / 16-byte chunks per staged row
            constexpr bool SWZ = CPR == 8;                      // d = 64: chunk c of row r at slot c ^ swz(r) (conflict-free read-back)
            const float osc = a.out_scale * (a.frame_scale ? a.frame_scale[fr] : 1.f);
            char* const stg = smem_raw + wave * (32 * RBY);
            const int wsw = SWZ ? (l31 >> 1) & 7 : 0;
#pragma unroll
            for (int d = 0; d < NDB; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int dv = 32 * d + 8 * g + 4 * hi;
                    if (32 * d + 8 * g < D) {                   // (compile-time: D is a multiple of 8, both halves of a chunk exist)
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = res[0][d][4 * g + e] * osc;
                        *reinterpret_cast<T4*>(stg + l31 * RBY + (((dv >> 3) ^ wsw) << 4) + 8 * hi) = cvt4<T>(v);
                    }
                }
            typedef __amdgpu_buffer_rsrc_t ORsrc;
            const ORsrc ro = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, 0x7fffffff, 0x00020000);
            const int so = (int)(((int64_t)fr * a.o_fs + h * D) * 2);
            constexpr int NPASS = (32 * CPR + 63) / 64;
#pragma unroll
            for (int i = 0; i < NPASS; ++i) {
                const int id = lane + 64 * i;
                const int row = id / CPR, cc = id - row * CPR;
                if (32 * CPR % 64 == 0 || id < 32 * CPR) {
                    T8 v = *reinterpret_cast<const T8*>(stg + row * RBY + ((SWZ ? cc ^ ((row >> 1) & 7) : cc) << 4));
                    const int q = q0 + row;
                    const int vo = (min(q, a.s - 1) * a.ldo + cc * 8) * 2;
                    if (q < a.s) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ro, vo, so, AID_ST_AUX);
                }
            }
        }
    } else
#pragma unroll
    for (int j = 0; j < QB; ++j) {
        const int q = q0 + 32 * j + l31;
        if (q < a.s) {
            const float osc = a.out_scale * (a.frame_scale ? a.frame_scale[fr] : 1.f);
            T* orow = reinterpret_cast<T*>(a.out) + (int64_t)fr * a.o_fs + (int64_t)q * a.ldo + h * D;
#pragma unroll
            for (int d = 0; d < NDB; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int dv = 32 * d + 8 * g + 4 * hi;
                    if (dv < D) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = res[j][d][4 * g + e] * osc;
                        if (a.accumulate) {
                            const f32x4 old = up4<T>(*reinterpret_cast<const T4*>(orow + dv));
                            v += old;
                        }
                        *reinterpret_cast<T4*>(orow + dv) = cvt4<T>(v);
                    }
                }
        }
    }
    if (RES) {
#pragma unroll
        for (int j = 0; j < QB; ++j)
#pragma unroll
            for (int ks = 0; ks < NQK; ++ks) qf[j][ks] = qn[j][ks];
    }
    }
}

// ------------------------------------------------------------------------------------------------
template <typename T, int D, int MODE, int NW, int QB, bool PIPE, bool RES = false, bool BIAS = false>
static hipError_t launch_variant(const AttnKParams& p, hipStream_t stream) {
    constexpr int DK = (D + 15) / 16 * 16, DV = (D + 31) / 32 * 32;
    constexpr int NREG = MODE == AID_MODE_PLAIN ? 1 : MODE == AID_MODE_INNER ? 2 : 3;
    const size_t smem = RES ? ((size_t)NREG * (RES_KEYS * (DK + 8) + DV * RES_VLD) + RES_SLACK) * sizeof(T)
                            : (size_t)((attn_prefetch(D, NW) || QB > 1) ? 2 : 1) * (KT * (DK + 8) + DV * VLD) * sizeof(T);
    static PerDevice<bool> attr_set;
    bool* done = attr_set.slot();
    if (!done) return hipErrorInvalidDevice;
    if (!*done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&aid_attn_kernel<T, D, MODE, NW, QB, PIPE, RES, BIAS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        *done = true;
    }
    const int grid = p.nqb * p.a.n_frames * p.a.heads;
    hipLaunchKernelGGL((aid_attn_kernel<T, D, MODE, NW, QB, PIPE, RES, BIAS>), dim3(grid), dim3(NW * 64), smem, stream, p);
    return hipGetLastError();
}

// Four waves per workgroup for every shape: one-wave workgroups were measured slower
// even at S = 64 (profiles/r01_attn_small_shapes.txt) — the K/V staging cost per query row quadruples.
static int attn_qb(const AidAttnArgs& a);

// Waves per workgroup.  Eight waves (256 query rows sharing one K / V^T staging pass and barrier) pay only where the
// kernel runs two waves per SIMD anyway: d = 64 OUTER at S = 4096 (+2 %; at S = 1024 the four-wave kernel — 221 VGPRs since
// it stopped keeping the -m blocks of its two states in registers — is 4 % faster); the three-wave PLAIN kernels lose
// 5 - 19 % and the 77-key cross-attention launches 4 - 8 % (profiles/r02_attn_notes.txt).  Built for d <= 80; development
// knob AID_ATTN_NW = 4 / 8.
static int attn_nw(const AidAttnArgs& a) {
    const int knob = tune(TUNE_ATTN_NW);
    if (a.d > 80) return 4;
    if (attn_qb(a) >= 2) return 4;
    if (knob >= 0) return knob == 8 ? 8 : 4;
    return (a.d == 64 && a.mode == AID_MODE_OUTER && a.l >= 2048) ? 8 : 4;
}

// query blocks (of 32 rows) per wave.  Measured (profiles/r02_attn_notes.txt, tools/kbench_attn_ab.py): 64 rows per wave
// pays only where the kernel still fits two waves per SIMD — d = 40 PLAIN (254 VGPRs, +4 % at S = 4096); every other
// variant needs > 256 VGPRs, runs one wave per SIMD and loses 25-35 %.  Development knob AID_ATTN_QB = 1 / 2 forces it.
static int attn_qb(const AidAttnArgs& a) {
    const int force = tune(TUNE_ATTN_QB);
    if (a.d != 40 || a.mode != AID_MODE_PLAIN) return 1;         // the other 64-row variants are not built (see above)
    if (force == 1 || force == 2) return force;
    return (a.d == 40 && a.mode == AID_MODE_PLAIN && a.s >= 2048 && a.l >= 1024) ? 2 : 1;
}

// software-pipelined main loop.  Built for d = 40 (every mode) and d = 64 PLAIN; measured (profiles/r02_attn_notes.txt):
// +5 % for d = 40 INNER (two waves per SIMD either way), neutral for d = 64 PLAIN (206 VGPRs: two waves per SIMD instead of
// three), slower wherever the extra live tile pushes the kernel to one wave per SIMD (OUTER).  It was the default for d = 40
// INNER until the program-order kernel of that variant got under 168 VGPRs (152: no persistent -m block, one `mix` segment
// instead of three code copies): three waves per SIMD beat the pipelined two (577 vs 588 us at S = 4096).  Not a default
// any more; development knob AID_ATTN_PIPE = 0 / 1 forces it for the built variants.
static bool attn_pipe(const AidAttnArgs& a) {
    const int knob = tune(TUNE_ATTN_PIPE);
    const bool built = a.d == 40 || (a.d == 64 && a.mode == AID_MODE_PLAIN);
    if (!built || a.l < 192) return false;
    return knob > 0;
}

// Resident key segments: short key sets (text tokens, image tokens) at d <= 80.  A workgroup (4 waves) takes one chunk of the
// query rows of its (frame, head) and works through it 128 rows at a time.  These launches are latency chains (start-up ->
// fill -> per q block: scores, softmax, PV, store; 8 % of the MFMA rate), so the number of chunks trades start-up cost per row
// against workgroups in flight: measured best at 4 chunks for S = 1024 and 8 for S = 4096 (1 or 2 chunks: +30 - 130 %,
// one q block per workgroup: +20 %; profiles/r02_attn_notes.txt).  Development knobs AID_ATTN_RES = 0 / 1, AID_ATTN_RES_CHUNKS.
constexpr int RES_CHUNKS_MAX = 8;
static bool attn_res(const AidAttnArgs& a) {
    const int knob = tune(TUNE_ATTN_RES);
    if (a.d > 80 || a.l > RES_KEYS) return false;
    if (knob >= 0) return knob != 0;
    return a.d == 40;           // measured in the stacks: d = 40 -25 % (inner) / -14 % (plain); d = 64 / 80 within +-5 % of streaming
}

template <typename T, int D, int MODE>
static hipError_t launch_nw(AttnKParams& p, hipStream_t stream) {
    p.q_iters = 1;
    if (p.a.bias) {                                             // score bias: the one instantiation that reads it
        p.nqb = (p.a.s + 127) / 128;
        return launch_variant<T, D, MODE, 4, 1, false, false, true>(p, stream);
    }
    if (D <= 80 && attn_res(p.a)) {
        const int nqb = (p.a.s + 127) / 128;                    // 128-row blocks (4 waves x 32 rows)
        int chunks = nqb / 2 < 1 ? 1 : nqb / 2 > RES_CHUNKS_MAX ? RES_CHUNKS_MAX : nqb / 2;
        if (tune(TUNE_ATTN_RES_CHUNKS) > 0) chunks = tune(TUNE_ATTN_RES_CHUNKS);
        p.nqb = nqb < chunks ? nqb : chunks;
        p.q_iters = (nqb + p.nqb - 1) / p.nqb;
        return launch_variant<T, D, MODE, 4, 1, false, (D <= 80)>(p, stream);
    }
    if (D == 40 && MODE == AID_MODE_PLAIN && attn_qb(p.a) == 2) {
        p.nqb = (p.a.s + 255) / 256;
        return launch_variant<T, D, MODE, 4, (D == 40 && MODE == AID_MODE_PLAIN) ? 2 : 1, false>(p, stream);
    }
    p.nqb = (p.a.s + 127) / 128;
    constexpr bool CAN_PIPE = D == 40 || (D == 64 && MODE == AID_MODE_PLAIN);
    if (D <= 80 && attn_nw(p.a) == 8) {
        p.nqb = (p.a.s + 255) / 256;
        return launch_variant<T, D, MODE, (D <= 80) ? 8 : 4, 1, false>(p, stream);
    }
    if (CAN_PIPE && attn_pipe(p.a)) return launch_variant<T, D, MODE, 4, 1, CAN_PIPE>(p, stream);
    return launch_variant<T, D, MODE, 4, 1, false>(p, stream);
}

template <typename T, int D>
static hipError_t launch_mode(AttnKParams& p, hipStream_t stream) {
    switch (p.a.mode) {
        case AID_MODE_PLAIN: return launch_nw<T, D, AID_MODE_PLAIN>(p, stream);
        case AID_MODE_INNER: return launch_nw<T, D, AID_MODE_INNER>(p, stream);
        default:             return launch_nw<T, D, AID_MODE_OUTER>(p, stream);
    }
}

template <typename T>
static hipError_t launch_d(AttnKParams& p, hipStream_t stream) {
    switch (p.a.d) {
        case 40:  return launch_mode<T, 40>(p, stream);
        case 64:  return launch_mode<T, 64>(p, stream);
        case 80:  return launch_mode<T, 80>(p, stream);
        case 160: return launch_mode<T, 160>(p, stream);
        default:  return hipErrorInvalidValue;
    }
}

// ------------------------------------------------------------------------------------------------
// Interpolated keys / values of the interior frames (reference interpolation.py:772-775):
//   k2[i] = (1 - c_i) k[begin] + c_i k[end]   (same for vt), frames with c_i in (0, 1) only — the
// attention kernel reads the end-point frames themselves for c_i == 0 / 1.  Pure streaming
// (HBM-bound): 16 B per lane, fp32 lerp, one rounding to the storage type.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void aid_lerp_kv_kernel(const T* __restrict__ k, const T* __restrict__ vt,
                                                          T* __restrict__ k2, T* __restrict__ vt2,
                                                          const float* __restrict__ coef, int n_frames, int begin,
                                                          int end, int64_t k_fs, int64_t vt_fs) {
    typedef typename Vec<T>::v8 T8;
    const int fr = blockIdx.y;
    const float c = coef[fr];
    if (c <= 0.f || c >= 1.f) return;      // end points and PLAIN-marked frames need no interpolated rows
    const int64_t nk8 = k_fs / 8, nv8 = vt_fs / 8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nk8 + nv8; i += (int64_t)gridDim.x * blockDim.x) {
        const bool isk = i < nk8;
        const int64_t j = isk ? i : i - nk8;
        const T* src = isk ? k : vt;
        const int64_t fs = isk ? k_fs : vt_fs;
        const f32x8 x0 = up8<T>(*reinterpret_cast<const T8*>(src + begin * fs + j * 8));
        const f32x8 x1 = up8<T>(*reinterpret_cast<const T8*>(src + end * fs + j * 8));
        f32x8 y;
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = fmaf(c, x1[e], (1.f - c) * x0[e]);
        *reinterpret_cast<T8*>((isk ? k2 : vt2) + fr * fs + j * 8) = cvt8<T>(y);
    }
}

hipError_t lerp_kv_launch(const void* k, const void* vt, void* k2, void* vt2, const float* coef, int n_frames, int begin,
                          int end, int64_t k_fs, int64_t vt_fs, int dtype, hipStream_t stream) {
    const int64_t n8 = (k_fs + vt_fs) / 8;
    const int bx = (int)((n8 + 255) / 256 < 1024 ? (n8 + 255) / 256 : 1024);
    dim3 grid(bx > 0 ? bx : 1, n_frames);
    if (dtype == AID_DTYPE_F16)
        hipLaunchKernelGGL(aid_lerp_kv_kernel<f16>, grid, dim3(256), 0, stream, (const f16*)k, (const f16*)vt, (f16*)k2,
                           (f16*)vt2, coef, n_frames, begin, end, k_fs, vt_fs);
    else
        hipLaunchKernelGGL(aid_lerp_kv_kernel<bf16>, grid, dim3(256), 0, stream, (const bf16*)k, (const bf16*)vt,
                           (bf16*)k2, (bf16*)vt2, coef, n_frames, begin, end, k_fs, vt_fs);
    return hipGetLastError();
}

bool attn_head_dim_supported(int d) { return d == 40 || d == 64 || d == 80 || d == 160; }

const char* attn_variant_name(const AidAttnArgs& a) {
    static thread_local char name[64];
    static const char* modes[] = {"plain", "inner", "outer"};
    if (a.bias)
        snprintf(name, sizeof(name), "aid_attn<%s,d%d,%s,nw4,bias>", a.dtype == AID_DTYPE_F16 ? "f16" : "bf16", a.d, modes[a.mode]);
    else if (attn_res(a))
        snprintf(name, sizeof(name), "aid_attn<%s,d%d,%s,res>", a.dtype == AID_DTYPE_F16 ? "f16" : "bf16", a.d, modes[a.mode]);
    else if (attn_qb(a) == 2)
        snprintf(name, sizeof(name), "aid_attn<%s,d%d,%s,nw%d,qb2>", a.dtype == AID_DTYPE_F16 ? "f16" : "bf16", a.d,
                 modes[a.mode], attn_nw(a));
    else if (attn_nw(a) != 8 && attn_pipe(a))
        snprintf(name, sizeof(name), "aid_attn<%s,d%d,%s,nw%d,pipe>", a.dtype == AID_DTYPE_F16 ? "f16" : "bf16", a.d,
                 modes[a.mode], attn_nw(a));
    else
        snprintf(name, sizeof(name), "aid_attn<%s,d%d,%s,nw%d>", a.dtype == AID_DTYPE_F16 ? "f16" : "bf16", a.d,
                 modes[a.mode], attn_nw(a));
    return name;
}

hipError_t attn_launch(const AidAttnArgs& a, hipStream_t stream, const char** variant, bool skip_single) {
    AttnKParams p;
    p.a = a;
    p.nqb = 0;
    p.q_iters = 1;
    p.skip_single = skip_single ? 1 : 0;
    if (tune(TUNE_ATTN_ORDER) == 0) p.a.n_plain = 0;         // development knob: plain XCD order for mixed launches
    p.c2 = a.softmax_scale * 1.4426950408889634f;
    hipError_t e = (a.dtype == AID_DTYPE_F16) ? launch_d<f16>(p, stream) : launch_d<bf16>(p, stream);
    if (variant) *variant = attn_variant_name(a);
    return e;
}

}  // namespace aid
