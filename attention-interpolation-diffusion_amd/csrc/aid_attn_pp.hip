// Ping-pong attention core for head dim 64 and whole 64-key tiles — plain attention (de-activated passes, reference
// interpolation.py:581-584), the PLAIN riders of a batched-CFG call, and the interpolated frames of INNER / OUTER calls
// (interpolation.py:626-664, 760-790) as ONE tile stream over the frame's key segments.  Same arithmetic as aid_attn_kernel (swapped
// products, -m folded into the score MFMAs' accumulator, lazy row reference; the row sums here are sums of the ROUNDED P — bf16: eight
// 4x4x4 MFMAs against a ones operand behind the PV MFMAs, f16: v_dot2c in the V slot — not a ones-row block); what differs is WHO does
// what WHEN:
//
//   One workgroup = 8 waves x 32 query rows of one (frame, head); waves w and w + 4 share a SIMD.  Waves 0-3 and waves 4-7
//   run the same program ONE BARRIER APART, and the program alternates two slots per 64-key tile:
//     M(t)   every MFMA of the tile in one burst: S(t) = K(t) Q'^T - m (8), then O^T += V^T(t-1) P(t-1)^T (8); in their shadow,
//            one per MFMA and in pinned program order, the 16 operand-fragment reads: V^T(t-1) beside the score MFMAs, K(t+1)
//            beside the PV MFMAs into the registers the score MFMAs released (bf16: + the eight row-sum MFMAs).  No VALU instruction.
//     V(t)   the VALU half: P(t) = 2^S(t), rounding to the storage type, the head-room test (bf16: on the exponent bits of the packed P,
//            f16: on the tile's row sum, formed here with v_dot2c); plus this wave's two LDS-DMA pieces of tile t + 6 and the counted
//            wait that retires its pieces of tile t + 3
//   so while one wave of a SIMD keeps the matrix pipe busy its partner does the exponentials, and vice versa.  In the
//   program-order kernel the three co-resident waves of a SIMD drift into the same phase and MFMA time and VALU time add up
//   (1100 cycles per wave-tile for 640 of MFMA, profiles/r02_attn_notes.txt); here they are complementary by construction.
//   A frame with several key segments (fused INNER: own + mix; OUTER: own / begin / end with the own-keys state parked in registers
//   and swapped back) is one stream; the DMA addresses walk it on running scalar offsets.
//   K / V^T tiles go HBM -> LDS by DMA (no staging registers, no ds_write pass) into a ring of eight 16 KB stages; rows are
//   unpadded 128 B, bank conflicts are avoided by the XOR swizzle of the GEMM (on the DMA source address and on the read).
//   The main loop is unrolled by eight so the ring stage is a constant: it sits in the offset field of every ds_read.
//
// Measured (profiles/r03_attn_notes.txt, same process): S = 4096 plain 597 us against 658 for the program-order kernel, fused outer
// 1089 against 1213, fused inner 838 against 943; S = 1024 plain 107 against 95 (a 16-tile stream on one workgroup per CU).
// Round 4 (profiles/r04_attn_notes.txt): one instantiation per call mode, a DMA stream that never stops (no tail branches in the V
// slot), head-room test on the row sum instead of a maximum chain: plain -4.2 %, outer -4.6 %, inner -5 % per launch; balanced
// persistent walk for every mode, staged stores; bf16 row sums on the matrix pipe + exponent-bit test, no padding in the loop,
// straight-line fast path: S = 4096 plain 524 - 562 us by box, fused outer 880 - 965, S = 1024 fused outer 132 - 143.
// aid_attn_fwd's default rule: fused OUTER / INNER from 1024 keys, everything else from 2048.
#include <type_traits>

#include "aid_common.hpp"
#include "aid_kernels.hpp"

namespace aid {

namespace {

constexpr int PKT = 64;                 // keys per tile
constexpr int PTILE = 8192;             // bytes per tile: K [64 keys][128 B] or V^T [64 channels][128 B]
constexpr int PSTAGE = 2 * PTILE;       // bytes per ring stage (K tile + V^T tile)
constexpr int PNS = 8;                  // ring depth (128 KB)

struct AttnPPParams {
    AidAttnArgs a;
    int32_t nqb;                        // 256-row q blocks per (frame, head)
    int32_t multi;                      // 1: this launch runs every frame of the call (it is the only launch)
    int32_t persist;                    // 1: one workgroup per CU walks several items (every item is whole 8-tile trips)
    int32_t hv_lo, hv_hi, hv_units;     // persistent walk: frames [hv_lo, hv_hi) are HEAVY (hv_units key segments instead of one) — a
                                        // balance hint from the host's view of the call; results never depend on it
    float   c2;                         // softmax_scale * log2(e)
#ifdef AID_ABLATIONS
    int32_t abl;                        // development builds only: timing ablations, results are garbage
#endif
};

typedef __amdgpu_buffer_rsrc_t Rsrc;

// acc + lo(w) + hi(w) for a register holding two storage-type values (v_dot2c_f32_bf16 / v_dot2c_f32_f16 against (1, 1))
template <typename T>
__device__ __forceinline__ float dot2_ones(uint32_t w, float acc);
template <>
__device__ __forceinline__ float dot2_ones<bf16>(uint32_t w, float acc) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    b2 one;
    one[0] = (__bf16)1.0f; one[1] = (__bf16)1.0f;
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2, w), one, acc, false);
}
template <>
__device__ __forceinline__ float dot2_ones<f16>(uint32_t w, float acc) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    h2 one;
    one[0] = (_Float16)1.0f; one[1] = (_Float16)1.0f;
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, w), one, acc, false);
}

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void slot_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// MODE = the call's mode (AID_MODE_*): a PLAIN instantiation carries none of the segment machinery, an INNER one no parked state — the
// registers and most of the scalar item records
template <typename T, int MODE>
__global__ __launch_bounds__(512) void aid_attn_pp_kernel(const AttnPPParams p) {
    typedef typename Vec<T>::v8 T8;
    typedef typename Vec<T>::v4 T4;
    constexpr int D = 64;
    constexpr float XTH = std::is_same<T, f16>::value ? 15.0f : 60.0f;      // head-room (log2) of P = 2^x in the storage type
    // bf16 (SUMM): P is kept 2^-XB below "1 at the row reference", so that P >= 2 — bit 14 of a bf16, the top exponent bit — IS the
    // head-room test (2^(XB + 1) above the reference); O and the row sums carry the same factor, which the quotient drops.  f16 has
    // no exponent range to spare for the bias (and its PV product would meet subnormal P): it keeps the row-sum test.
    constexpr bool SUMM = std::is_same<T, bf16>::value;
    constexpr float XB = SUMM ? XTH - 1.f : 0.f;
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) char smem[];

#ifdef AID_ABLATIONS
    long long tl[6];                                            // 32: workgroup timeline (shader cycles), written over the output rows
    tl[0] = clock64();
#endif
    const AidAttnArgs& a = p.a;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int l31 = lane & 31, hi = lane >> 5;

    // ---- work items: 256 query rows of one (frame, head).
    // p.persist == 0: one item per workgroup in hardware dispatch order; logical order [head][frame][q block], every XCD a contiguous
    // range of it, the three-segment frames first inside the range.
    // p.persist == 1 (every item is whole 8-tile trips): one workgroup per CU; workgroup b sits on XCD b % 8 and walks a STATIC,
    // BALANCED share of its XCD's items (w = b / 8 of W workgroups per XCD).  An XCD owns a contiguous range of (head, q block) PAIRS,
    // every pair with all its frames — so every XCD holds the same mix of heavy frames (hv_units key segments) and light ones (one
    // segment), and a (frame, head)'s q blocks still meet in one L2 (contiguous ranges of the [head][frame][q block] order cut through
    // heads: S = 4096, 7 + 7 frames put 520 segment-units on one XCD and 440 on another).  Inside the XCD the heavy items, ordered
    // [frame][pair], are dealt cyclically (w, w + W, ...); then the light items first level the workgroups that got one heavy item
    // less (hv_units light items each), and the rest is dealt cyclically again: S = 4096 gives every workgroup exactly 15
    // segment-units, S = 1024 7 or 8 (profiles/r04_attn_notes.txt; round 3's snake order could not balance the 3 : 1 mix of a fused
    // OUTER call, which therefore paid a workgroup's ~5 us start-up per ITEM).  The tile stream, the DMA ring and the slot alternation
    // run on across an item boundary, so the start-up (dispatch, scalar loads, Q and six tiles in flight, first product) is paid once
    // per launch.  The walk is a pure function of (block, j): no atomics, the same result bit for bit run to run.
    const int nt = a.l / PKT;                                   // tiles per key segment
    const int n_items = p.nqb * a.n_frames * a.heads;
    const int n_heavy = a.n_frames - a.n_plain;
    const bool hf = p.multi && a.n_plain > 0 && n_heavy > 0;
    const int xcd = blockIdx.x & 7, wx = blockIdx.x >> 3, WX = gridDim.x >> 3;
    const int nhf = p.hv_hi - p.hv_lo;                          // heavy frames
    int pb = 0, npx = 0;                                        // this XCD's pairs [pb, pb + npx)
    if (p.persist) {
        const int np = a.heads * p.nqb, q = np >> 3, r = np & 7;
        pb = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        npx = q + (xcd < r ? 1 : 0);
    }
    // this workgroup's j-th item as a logical id of the [head][frame][q block] order (>= n_items: none)
    auto item_lid = [&](int j) __attribute__((always_inline)) {
        if (!p.persist) {
            if (j != 0) return n_items;
            return hf ? heavy_first((int)blockIdx.x, n_items, n_heavy * p.nqb, a.n_frames * p.nqb) : xcd_remap((int)blockIdx.x, n_items);
        }
        const int nhx = npx * nhf, nlx = npx * (a.n_frames - nhf);
        const int hq = nhx / WX, hr = nhx % WX;
        const int hc = hq + (wx < hr ? 1 : 0);                  // heavy items of this workgroup
        int t, fr;
        if (j < hc) {
            t = wx + j * WX;                                    // t-th heavy item of the XCD, order [frame][pair]
            fr = p.hv_lo + t / npx;
        } else {
            const int jj = j - hc;
            const int ndef = hr > 0 ? WX - hr : 0;              // workgroups one heavy item short
            const int p2 = min(nlx, p.hv_units * ndef);         // light items [0, p2) level them
            const int c1 = (hr > 0 && wx >= hr) ? max(min(p.hv_units * (wx - hr + 1), p2) - p.hv_units * (wx - hr), 0) : 0;
            if (jj < c1) {
                t = p.hv_units * (wx - hr) + jj;
            } else {
                t = p2 + wx + (jj - c1) * WX;
                if (t >= nlx) return n_items;
            }
            const int fi = t / npx;                             // t-th light item of the XCD
            fr = fi < p.hv_lo ? fi : fi + nhf;
        }
        const int pi = pb + t % npx;
        return ((pi / p.nqb) * a.n_frames + fr) * p.nqb + pi % p.nqb;
    };
    // Key segments of a frame (the same decisions aid_attn_kernel takes, on the same device coefficients):
    //   single  — PLAIN call, negative coefficient (PLAIN rider of a batched-CFG call), fused END-POINT frame: own keys only.
    //   OUTER   — reference interpolation.py:626-664: softmaxes over [own ; begin] and [own ; end] (fused) or over begin and end
    //             (pure), outputs mixed (1 - c) : c.  The walk is own -> begin -> end as ONE tile stream; the state after the own
    //             segment serves both sides: it is PARKED in registers (po / pl / pm) when the begin segment starts and swapped
    //             back in when the end segment starts, while the parked registers take the finished begin side (pure: the parked
    //             state is the empty one).  c == 0 / c == 1: the zero-weighted side is dropped.
    //   INNER   — interpolation.py:760-790: one softmax over [own ; mix] (fused) or mix (pure); mix = the interpolated keys / values
    //             aid_lerp_kv wrote to k2 / vt2 (row = frame), or the begin / end row itself for c == 0 / 1.
    // A launch that does not own the whole call (p.multi == 0: the split launches behind ATTN_V2 = 1) runs single frames only.
    const int row_b = a.begin, row_e = a.end;
    // the item being planned (N_*) and the item being computed (no prefix); adopt() moves the one into the other
    int N_fr = 0, N_h = 0, N_q0 = 0, N_nseg = 1, N_ks0 = 0, N_ks1 = 0, N_ks2 = 0, N_vs0 = 0, N_vs1 = 0, N_vs2 = 0, N_t2 = 0;
    int N_park = -1, N_swap = -1;
    float N_wb = 0.f, N_we = 1.f;
    bool N_skip = false;
    auto plan = [&](int lid) __attribute__((always_inline)) {
        const int qb = lid % p.nqb;
        N_fr = (lid / p.nqb) % a.n_frames;
        N_h = lid / (p.nqb * a.n_frames);
        N_q0 = (qb * 8 + wave) * 32;
        const int kvf = a.kv_map ? a.kv_map[N_fr] : N_fr;
        int seg0 = kvf, seg1 = 0;
        N_nseg = 1; N_t2 = 0; N_park = -1; N_swap = -1; N_wb = 0.f; N_we = 1.f; N_skip = false;
        if (MODE != AID_MODE_PLAIN) {
            const float cf = a.coef[N_fr];
            const bool single = cf < 0.f || (a.fused && ((cf == 0.f && kvf == row_b) || (cf == 1.f && kvf == row_e)));
            if (!single) {
                N_skip = !p.multi;
                // (arithmetic, not `c == 1 ? a.end : a.begin`: a select between two FIELDS of the by-value argument struct becomes a
                //  select between their addresses and hipcc then keeps the whole struct in scratch)
                const bool both = cf != 0.f && cf != 1.f;
                const int side = row_b + (cf == 1.f ? 1 : 0) * (row_e - row_b);       // the end-point row of a one-sided frame
                const int fz = a.fused ? 1 : 0;
                if (MODE == AID_MODE_OUTER) {
                    N_nseg = fz + (both ? 2 : 1);
                    const int first = both ? row_b : side;
                    seg0 = fz * kvf + (1 - fz) * first;
                    seg1 = fz * first + (1 - fz) * row_e;
                    if (both) { N_wb = 1.f - cf; N_we = cf; N_park = fz ? 1 : -1; N_swap = fz + 1; }
                } else {
                    N_nseg = fz + 1;
                    const int mix = both ? N_fr : side;
                    seg0 = fz * kvf + (1 - fz) * mix;
                    seg1 = mix;
                    N_t2 = both ? (fz ? 2 : 1) : 0;
                }
            }
        }
        // one descriptor per tensor; the key / value ROW of a segment and the head ride in the scalar offset (tensors < 2 GB)
        const int kh = N_h * D * 2, vh = N_h * D * a.ldvt * 2;
        N_ks0 = seg0 * (int)a.k_fs * 2 + kh;  N_ks1 = seg1 * (int)a.k_fs * 2 + kh;  N_ks2 = row_e * (int)a.k_fs * 2 + kh;
        N_vs0 = seg0 * (int)a.vt_fs * 2 + vh; N_vs1 = seg1 * (int)a.vt_fs * 2 + vh; N_vs2 = row_e * (int)a.vt_fs * 2 + vh;
    };
    int fr, h, q0, nseg, ks0, ks1, ks2, vs0, vs1, vs2, t2mask, park_at, swap_at, NT;
    float w_b, w_e;
    auto adopt = [&]() __attribute__((always_inline)) {
        fr = N_fr; h = N_h; q0 = N_q0; nseg = N_nseg; ks0 = N_ks0; ks1 = N_ks1; ks2 = N_ks2; vs0 = N_vs0; vs1 = N_vs1; vs2 = N_vs2;
        t2mask = N_t2; park_at = N_park; swap_at = N_swap; w_b = N_wb; w_e = N_we;
        NT = nseg * nt;
    };
    {
        const int first = item_lid(0);
        if (first >= n_items) return;                           // a persistent workgroup the deal left empty (fewer items than the balance needs)
        plan(first);
    }
    if (N_skip) return;                                         // (split launches are never persistent)
    adopt();

    // ---- Q fragments (B operand of the swapped product) -----------------------------------------
    T8 qf[4];
    auto scale_q = [&]() __attribute__((always_inline)) {       // generic callers: fold softmax_scale * log2(e) into Q here (one extra
        if (!a.q_prescaled) {                                   // rounding; the processor path does it in the q projection's epilogue)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                f32x8 t = up8<T>(qf[ks]);
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] *= p.c2;
                qf[ks] = cvt8<T>(t);
            }
        }
    };
    {                                                           // the first item's rows: straight from global memory
        const int qr = min(q0 + l31, a.s - 1);                  // rows past the end are clamped, never stored
        const T* qrow = reinterpret_cast<const T*>(a.q) + (int64_t)fr * a.q_fs + (int64_t)qr * a.ldq + h * D;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const T8*>(qrow + ks * 16 + hi * 8);
        scale_q();
    }

    // ---- DMA addressing: this wave's piece (8 rows x 128 B) of every K tile and of every V^T tile ------------
    // one descriptor per tensor (attn_pp_supported: below 2 GB); INNER: the interpolated keys / values live in their own tensors
    const Rsrc rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.k), 0, 0x7fffffff, 0x00020000);
    const Rsrc rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.vt), 0, 0x7fffffff, 0x00020000);
    const Rsrc rk2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.k2), 0, 0x7fffffff, 0x00020000);
    const Rsrc rv2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.vt2), 0, 0x7fffffff, 0x00020000);
    // the NEXT item's Q rows wait in LDS behind the ring (8 waves x 4 KB, already in fragment order: piece ks of a wave = its 64 lanes'
    // 16 B of k-step ks), fetched by DMA while the current item runs — no registers held for them
    const Rsrc rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.q), 0, 0x7fffffff, 0x00020000);
    char* const qlds = smem + 2 * PNS * PTILE + wave * 4096;
    auto dma_q_next = [&]() __attribute__((always_inline)) {
        const int qr = min(N_q0 + l31, a.s - 1);
        const int vo = qr * (a.ldq * 2) + hi * 16;
        const int so = (int)(((int64_t)N_fr * a.q_fs + N_h * D) * 2);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (__attribute__((address_space(3))) void*)(qlds + ks * 1024), 16, vo, so + ks * 32, 0, 0);
    };
    const int prow = 8 * wave + (lane >> 3);                    // tile row this lane fetches
    const int pch = (lane & 7) ^ ((prow >> 1) & 7);             // logical 16-B chunk stored at slot lane & 7 (XOR swizzle)
    const int kvo = prow * (a.ldk * 2) + pch * 16;              // + key0 * ldk * 2 (scalar)
    const int vvo = prow * (a.ldvt * 2) + pch * 16;             // + key0 * 2       (scalar)
    // The DMA stream walks the frame's segments tile by tile with RUNNING scalar offsets (two s_add per tile; the per-tile
    // segment / tensor arithmetic of a random-access `tile g` cost ~45 SALU instructions in every V slot and 4 % of the kernel):
    // dk / dv = byte offsets of the next tile to request, dleft = tiles left in its segment, dseg = that segment, d2 = it reads k2 / vt2
    int dk = ks0, dv = vs0, dleft = nt, dseg = 0;
    bool d2 = (t2mask & 1) != 0;
    const int kstep = PKT * a.ldk * 2;
    auto dma_next = [&](int st) __attribute__((always_inline)) {    // st: ring stage, a constant wherever the caller's tile index is
        char* dst = smem + st * PTILE + wave * 1024;
        if (MODE == AID_MODE_INNER && d2) {                     // (a branch, not a select between the two descriptors)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rk2, (__attribute__((address_space(3))) void*)dst, 16, kvo, dk, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rv2, (__attribute__((address_space(3))) void*)(dst + PNS * PTILE), 16, vvo, dv, 0, 0);
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (__attribute__((address_space(3))) void*)dst, 16, kvo, dk, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (__attribute__((address_space(3))) void*)(dst + PNS * PTILE), 16, vvo, dv, 0, 0);
        }
        dk += kstep;
        dv += PKT * 2;
        if (__builtin_expect(--dleft == 0, 0)) {                // next segment (arithmetic on values, see above)
            ++dseg;
            dleft = nt;
            if (dseg < nseg) {
                dk = ks1 + (dseg >= 2 ? ks2 - ks1 : 0);
                dv = vs1 + (dseg >= 2 ? vs2 - vs1 : 0);
                d2 = ((t2mask >> dseg) & 1) != 0;
            } else {                                            // the stream runs on into the first segment of the planned item
                dk = N_ks0;
                dv = N_vs0;
                d2 = (N_t2 & 1) != 0;
            }
        }
    };
    bool has_next = false;

    // ---- fragment read offsets: K rows with key bits 2 <-> 3 swapped (so P comes out in B-operand order), V^T rows = channels
    const int krow = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
    const int kx = hi ^ ((krow >> 1) & 7), vx = hi ^ ((l31 >> 1) & 7);      // swz(r + 32) == swz(r)
    const int koff = krow * 128, voff = PNS * PTILE + l31 * 128;
    // LDS: the K tiles of the eight stages in [0, 64 KB), the V^T tiles in [64 KB, 128 KB): per-lane address of k-step ks in the
    // register (< 4 KB, resp. 64 KB + < 4 KB), stage * 8 KB + block * 4 KB in the 16-bit offset field (<= 61440)
    int kad[4], vad[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        kad[ks] = koff + (((2 * ks) ^ kx) << 4);
        vad[ks] = voff + (((2 * ks) ^ vx) << 4);
    }

    // ---- online-softmax state -----------------------------------------------------------------------
    float m = 0.f;
    bool fresh = true;
    f32x16 o[2], sc[2];
    float lsum = 0.f;                   // row sum of the ROUNDED P over the 32 keys per tile this lane holds (the partner lane has the others)
    f32x4 lacc = {0.f, 0.f, 0.f, 0.f};  // SUMM: the same sum as the D block of the 4x4x4 MFMAs (four equal copies; lsum unused)
    s16x4 ones4;                        // SUMM: A operand of those MFMAs, 4 x 4 ones per block
#pragma unroll
    for (int i = 0; i < 4; ++i) ones4[i] = 0x3f80;
    asm volatile("" : "+v"(ones4));
    auto get_l = [&]() __attribute__((always_inline)) -> float { return SUMM ? lacc[0] : lsum; };
    auto set_l = [&](float x) __attribute__((always_inline)) {
        if (SUMM) { lacc[0] = x; lacc[1] = x; lacc[2] = x; lacc[3] = x; } else lsum = x;
    };
    T8 pf[4];
    f32x16 cneg;                        // -m as an accumulator block (C operand of a tile's first score MFMAs), rebuilt when m moves
    f32x16 po[2];                       // parked: O of the own-keys state while the begin side runs, then the finished begin side
    float pl = 0.f, pm = 0.f;           // parked row sum (this lane's register of the row-sum block) and row reference
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; cneg[r] = 0.f; }
    if (MODE == AID_MODE_OUTER) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { po[0][r] = 0.f; po[1][r] = 0.f; }
    }
    asm volatile("" : "+v"(cneg));
#pragma unroll
    for (int i = 0; i < 4; ++i) pf[i] = zero8<T>();

#ifdef AID_ABLATIONS
    // 16: slot timing — shader cycles of [V work | wait at the barrier behind it | M work | wait at the barrier behind it], summed over the
    // tiles and written over the output rows (lane 0 of every wave: four floats = cycles per tile)
    long long tmark = clock64();
    float tacc[4] = {0.f, 0.f, 0.f, 0.f};
    auto stamp = [&](int which) __attribute__((always_inline)) {
        if (p.abl & 16) {
            __builtin_amdgcn_sched_barrier(0);
            const long long now = clock64();
            tacc[which] += (float)(now - tmark);
            tmark = now;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
#define PP_STAMP(i) stamp(i)
#else
#define PP_STAMP(i)
#endif

    // MFMA results and the VALU code that reads them.  The hazard recogniser does not see across the slots' barriers / branches
    // (aid_attn.hip has the history), so the distance is kept by hand: an 8-pass MFMA's result may be read 11 wait states (of four
    // cycles) after its issue.  `sc` is read by the next V slot — behind the eight PV MFMAs that follow the score MFMAs in every M slot
    // and a barrier (>= 48 cycles): no padding needed in the loop, only the compiler is told not to move anything across (tie).
    // `o` / the row-sum block are read by the rare paths only (raise, park, swap_sides, finish): THEY start with the padding (settle)
    // — 20 wait states per M slot were 76 cycles of every 1120-cycle interval (tools/ubench/barrier_cost.hip).  The prologue's S(0)
    // has nothing behind it and is settled as before.
    auto tie = [&]() __attribute__((always_inline)) {
        asm volatile("" : "+v"(sc[0]), "+v"(sc[1]));
        asm volatile("" : "+v"(o[0]), "+v"(o[1]));
        if (SUMM) asm volatile("" : "+v"(lacc));
    };
    auto settle = [&]() __attribute__((always_inline)) {
        tie();
        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
        tie();
    };
    // operand fragments: kf = K(t + 1) for S(t + 1), vf = V^T(t) for O += V^T(t) P(t)^T, all four k-steps each
    T8 kf[4][2], vf[4][2];
    auto lds_k = [&](int st, int ks, int b) __attribute__((always_inline)) {           // st: ring stage, a constant wherever it matters
        return *reinterpret_cast<const T8*>(smem + kad[ks] + (st * PTILE + b * 4096));
    };
    auto lds_v = [&](int st, int kk, int d) __attribute__((always_inline)) {
        return *reinterpret_cast<const T8*>(smem + vad[kk] + (st * PTILE + d * 4096));
    };
    auto pin = []() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };

    // M slot of the main loop — S(t + 1) = K(t + 1) Q'^T - m, then O^T += V^T(t) P(t)^T — and the fragment reads that ride in its
    // shadow, one per MFMA, program order pinned:  V^T(t) (stage sv) beside the eight score MFMAs, which use the K fragments read
    // one M slot ago; K(t + 2) (stage sk) beside PV MFMAs 2 .. 9, into the registers the score MFMAs have released.  No VALU.
    auto mslot = [&](int sv, int sk) __attribute__((always_inline)) {
#ifdef AID_ABLATIONS
        if (p.abl & 4) {                                        // 4: no MFMA slot at all
            asm volatile("" : "+v"(sc[0]), "+v"(sc[1]));
            return;
        }
#endif
        pin();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ks = i >> 1, b = i & 1;
#ifdef AID_ABLATIONS
            if (!((p.abl & 64) && b))                           // 64: half the fragment reads (is the M slot bound by the LDS port?)
#endif
            vf[ks][b] = lds_v(sv, ks, b);
            pin();
            sc[b] = mfma32(kf[ks][b], qf[ks], ks ? sc[b] : cneg);
            pin();
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int kk = i >> 1, w = i & 1;
            o[w] = mfma32(vf[kk][w], pf[kk], o[w]);
            pin();
            if constexpr (SUMM) {                               // l += the lane's four P values of half w of k-step kk (D[i][j] = sum_k B[k][j])
                const u32x4 w4 = __builtin_bit_cast(u32x4, pf[kk]);
                u32x2 h2;
                h2[0] = w4[2 * w]; h2[1] = w4[2 * w + 1];
                lacc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(ones4, __builtin_bit_cast(s16x4, h2), lacc, 0, 0, 0);
                pin();
            }
#ifdef AID_ABLATIONS
            if (!((p.abl & 64) && w))
#endif
            kf[kk][w] = lds_k(sk, kk, w);                       // (kf[kk][w] was released by score MFMA 2 kk + w, eight or more MFMAs ago)
            pin();
        }
    };

    // V slot: P(t) = 2^S(t) (row reference raised first when the head-room of the storage type is exceeded), rounded to the storage
    // type; this wave's two DMA pieces of tile t + LEAD; the counted wait that retires its pieces of tile t + 3.
    // bf16: 32 v_exp + 16 v_cvt_pk + 8 v_or3 / v_bitop3 + 1 compare — the row sums are the M slot's (eight 4x4x4 MFMAs of 8 cycles behind
    // the PV MFMAs; 16 v_dot2c + 2 v_mov + 2 v_add less here: -0.4 ... -4 % per launch, profiles/r04_attn_notes.txt table 5).
    // (The requests stay HERE: issued from the M slot they cost 3 - 18 %, table 4.)
    constexpr int LEAD = 6;
    auto vslot = [&](int t, bool first) __attribute__((always_inline)) {
        // The DMA stream never stops: behind the workgroup's last item it wraps into that item's first segment again (valid memory,
        // nobody reads those stages), so the slot has no "is there a tile t + LEAD" branch and ONE counted wait — the tail logic cost
        // 1.6 % (plain) to 6 % (inner) of the launch (profiles/r04_attn_notes.txt).
#ifdef AID_ABLATIONS
        if (p.abl & 1) {                                        // 1: no VALU work in the V slot
            if (!(p.abl & 2)) dma_next((t + LEAD) & (PNS - 1));
            fresh = false;
            wait_vm<6>();
            return;
        }
        if (!(p.abl & 2))
#endif
        dma_next((t + LEAD) & (PNS - 1));
        // P(t) = 2^S(t) rounded to the storage type (lane (q, hi): sc[b][r] belongs to key 32 b + 16 (r >> 3) + 8 hi + (r & 7) of the tile
        // = k-step 2 b + (r >> 3) of PV), and the tile's row sum of the ROUNDED values: v_dot2c against (1, 1), two keys per instruction,
        // on two accumulators (the matrix pipe's ones-row block cost 4 of 20 MFMAs)
        auto exp_tile = [&]() __attribute__((always_inline)) -> float {
            float t0 = 0.f, t1 = 0.f;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    f32x8 pv;
#pragma unroll
                    for (int e = 0; e < 8; ++e) pv[e] = __builtin_amdgcn_exp2f(sc[b][8 * u + e]);
                    pf[2 * b + u] = cvt8<T>(pv);
                    const u32x4 w4 = __builtin_bit_cast(u32x4, pf[2 * b + u]);
                    t0 = dot2_ones<T>(w4[0], t0); t1 = dot2_ones<T>(w4[1], t1);
                    t0 = dot2_ones<T>(w4[2], t0); t1 = dot2_ones<T>(w4[3], t1);
                }
            return t0 + t1;
        };
        // SUMM: the same without the sums — they are formed from pf in the M slot — returning the OR of the 16 packed words instead
        // (v_or3_b32): a value >= 2 anywhere shows in bit 14 of one of its halves (P >= 0: no other value has that bit)
        auto exp_tile_or = [&]() __attribute__((always_inline)) -> uint32_t {
            uint32_t acc = 0;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    f32x8 pv;
#pragma unroll
                    for (int e = 0; e < 8; ++e) pv[e] = __builtin_amdgcn_exp2f(sc[b][8 * u + e]);
                    pf[2 * b + u] = cvt8<T>(pv);
                    const u32x4 w4 = __builtin_bit_cast(u32x4, pf[2 * b + u]);
                    acc = (b == 0 && u == 0) ? (w4[0] | w4[1] | w4[2]) | w4[3] : (acc | w4[0] | w4[1]) | (w4[2] | w4[3]);
                }
            return acc;
        };
        auto row_max = [&]() __attribute__((always_inline)) -> float {
            float xm = fmaxf(sc[0][0], sc[0][1]);
#ifdef AID_ABLATIONS
            if (!(p.abl & 8))                                   // 8: no head-room check (the maximum chain)
#endif
#pragma unroll
            for (int i = 1; i < 16; ++i) xm = fmaxf(fmaxf(xm, sc[i >> 3][(2 * i) & 15]), sc[i >> 3][(2 * i + 1) & 15]);
            return xm;
        };
        // slow path (first tile of the row, or a score out-grew the head-room of the storage type): move the reference to the row
        // maximum, rescale O and l and shift this tile's arguments; PV(t - 1) is complete, O is at rest
        auto raise = [&](float xm) __attribute__((always_inline)) {
            settle();
            const float rowmax = max_halves(xm) + XB;           // (SUMM: the row maximum lands at -XB)
            const float shift = fresh ? rowmax : fmaxf(rowmax, 0.f);
            const float alpha = __builtin_amdgcn_exp2f(-shift);
            m += shift;
#pragma unroll
            for (int r = 0; r < 16; ++r) cneg[r] = -m;
            asm volatile("" : "+v"(cneg));                      // (opaque: keeps hipcc from rebuilding the block in every M slot)
#pragma unroll
            for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
            if (SUMM) lacc *= alpha; else lsum *= alpha;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[b][r] -= shift;
            fresh = false;
        };
        // Exponentiate first, test afterwards: the scores stay intact in `sc`, so a tile whose row sum shows that an argument out-grew
        // the head-room (sum of this lane's 32 values above 2^XTH; an f32 / storage-type overflow arrives as +inf and tests true as well)
        // is simply redone against the raised reference.  The sixteen v_max3 of a per-tile maximum chain become one compare:
        // -2.7 ... -3.1 % per launch (profiles/r04_attn_notes.txt); every P that reaches the PV product is <= 2^XTH as before.
        // (`fresh` is set where an item or a pure OUTER side starts: tile 0 of a trip.  `first` is a constant in each unrolled copy,
        //  so seven of eight copies carry no test of it; the unlikely paths sit out of line: two taken branches less per slot, -1.4 ... -3.8 %)
        const bool fr = first && fresh;
        if constexpr (SUMM) {
            const uint32_t orw = fr ? 0u : exp_tile_or();
            if (__builtin_expect(fr || __any((orw & 0x40004000u) != 0u), 0)) {
                raise(row_max());
                exp_tile_or();
            }
        } else {
            float ts = fr ? 0.f : exp_tile();
            if (__builtin_expect(fr || __any(ts > __builtin_amdgcn_exp2f(XTH)), 0)) {
                raise(row_max());
                ts = exp_tile();
            }
            lsum += ts;
        }
        wait_vm<6>();                                           // retires tile t + 3: the three tiles behind it stay in flight
    };

    // Segment boundaries of a two-sided frame.  S(t) of the next segment's first tile is in `sc` and PV(t - 1) closed the segment
    // before it: O, l, m are at rest.
    auto park = [&]() __attribute__((always_inline)) {          // own keys done: park the state, the begin side continues on it
        settle();
#pragma unroll
        for (int r = 0; r < 16; ++r) { po[0][r] = o[0][r]; po[1][r] = o[1][r]; }
        pl = get_l();
        pm = m;
    };
    auto swap_sides = [&]() __attribute__((always_inline)) {    // begin side done: keep (1 - c) O_b / l_b, resume the own-keys state
        settle();
        const float lown = get_l();
        const float lrow = lown + other_half(lown);             // (all lanes take part in the exchange)
        const float wb = w_b / lrow;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float rb = o[d][r] * wb;
                o[d][r] = po[d][r];
                po[d][r] = rb;
            }
        set_l(pl);
        const float back = m - pm;                              // S(t) was formed against the begin side's reference (>= the parked one)
        m = pm;
#pragma unroll
        for (int r = 0; r < 16; ++r) cneg[r] = -m;
        asm volatile("" : "+v"(cneg));
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[b][r] += back;
        if (park_at < 0) fresh = true;                          // pure OUTER: the end side starts from the empty state
    };

    // O / l of the finished item; lane (q = l31, hi) holds dv = 32 d + 8 g + 4 hi + {0..3}.  The wave's 32 x 64 block goes through ITS
    // 4 KB of LDS behind the ring (the next item's Q rows were read out of it before the last M slot; a persistent walk requests the
    // item after that only at the top of the next item) and leaves as four 16-byte stores per lane covering whole 128-byte rows: eight
    // 8-byte stores per lane at a row stride are store-ISSUE bound (~1.4 us per item measured on the short-stream kernel; the guide's T21).
    auto finish = [&]() __attribute__((always_inline)) {
        settle();
#ifdef AID_ABLATIONS
        if (p.abl & (16 | 32)) {                                // slot timing / workgroup timeline instead of the result
            if (lane == 0 && q0 < a.s) {
                float* dbg = reinterpret_cast<float*>(reinterpret_cast<T*>(a.out) + (int64_t)fr * a.o_fs + (int64_t)q0 * a.ldo + h * D);
                if (p.abl & 16) for (int i = 0; i < 4; ++i) dbg[i] = tacc[i] / (float)NT;
                else { tl[4] = clock64(); for (int i = 1; i < 5; ++i) dbg[i - 1] = (float)(tl[i] - tl[0]); }
            }
            for (int i = 0; i < 4; ++i) tacc[i] = 0.f;          // per item; the item boundary itself is not in the slots
            tmark = clock64();
            return;
        }
#endif
        const float lown = get_l();
        const float inv = w_e / (lown + other_half(lown));      // (w_e = 1 unless this frame mixes two sides)
        if (q0 >= a.s) return;                                  // (wave-uniform: a wave past the last row)
        const float osc = a.out_scale * (a.frame_scale ? a.frame_scale[fr] : 1.f);
        if (MODE == AID_MODE_OUTER) {
            // (the OUTER instantiation sits at 250 of 256 registers with the parked state: it keeps the 8-byte row-per-lane stores — its
            //  items are two or three key segments long, the store tail is <= 3 % of an item)
            const int q = q0 + l31;
            if (q < a.s) {
                T* orow = reinterpret_cast<T*>(a.out) + (int64_t)fr * a.o_fs + (int64_t)q * a.ldo + h * D;
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int dv_ = 32 * d + 8 * g + 4 * hi;
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = (o[d][4 * g + e] * inv + po[d][4 * g + e]) * osc;      // po: the begin side, or zero
                        if (a.accumulate) v += up4<T>(*reinterpret_cast<const T4*>(orow + dv_));
                        *reinterpret_cast<T4*>(orow + dv_) = cvt4<T>(v);
                    }
            }
        } else {
            // stage: 16-byte chunk c = 4 d + g of row r sits at slot c ^ swz(r) (two-way conflicts at most on the 8-byte writes)
            char* const st = qlds;
            const int wsw = (l31 >> 1) & 7;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = o[d][4 * g + e] * inv * osc;
                    *reinterpret_cast<T4*>(st + l31 * 128 + (((4 * d + g) ^ wsw) << 4) + 8 * hi) = cvt4<T>(v);
                }
            // read back row-major: lane -> (row lane / 8 + 8 i, chunk lane % 8), one whole 128-byte row per eight lanes
            const Rsrc ro = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, 0x7fffffff, 0x00020000);     // (built here: no SGPRs held across the item)
            const int so = (int)(((int64_t)fr * a.o_fs + h * D) * 2);
            const int rr = lane >> 3, cc = lane & 7;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = rr + 8 * i;
                T8 v = *reinterpret_cast<const T8*>(st + row * 128 + ((cc ^ ((row >> 1) & 7)) << 4));
                const int q = q0 + row;
                const int vo = (min(q, a.s - 1) * a.ldo + cc * 8) * 2;
                if (a.accumulate) v = cvt8<T>(up8<T>(v) + up8<T>(__builtin_bit_cast(T8, __builtin_amdgcn_raw_buffer_load_b128(ro, vo, so, 0))));
                if (q < a.s) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ro, vo, so, AID_ST_AUX);
            }
        }
    };

    // Schedule.  Group g runs M(t) in interval 2 t + g and V(t) in 2 t + 1 + g (M(t) computes S(t) and the PV product of tile t - 1).
    // Tile T is read in M(T - 1) (K, for the next slot's scores) and in M(T + 1) (V^T) = intervals 2 T - 2 .. 2 T + 3; every read is
    // complete (lgkmcnt) before its reader's next barrier, so stage T & 7 is free from interval 2 T + 4.  Tile U = T + 8 lands there:
    // requested in V(U - 6) (interval 2 U - 11 + g >= 2 T + 5), retired by the counted wait of V(U - 3) — three tile periods later —
    // and barrier-published at the end of interval 2 U - 4 at the latest, two intervals before M(U - 1) reads it.
    // ---- prologue: pieces of tiles 0 .. 5 requested, 0 .. 2 retired and published; S(0); fragments of K(1) ----------------
#pragma unroll
    for (int t = 0; t < LEAD; ++t)
        dma_next(t);
#ifdef AID_ABLATIONS
    if (p.abl & 32) { tl[1] = clock64(); asm volatile("" :: "v"(qf[0]), "v"(qf[3])); }
#endif
    wait_vm<6>();
#ifdef AID_ABLATIONS
    if (p.abl & 32) tl[2] = clock64();
#endif
    slot_barrier();
    if (grp == 1) slot_barrier();                               // the second group runs one barrier behind
#pragma unroll
    for (int i = 0; i < 8; ++i) kf[i >> 1][i & 1] = lds_k(0, i >> 1, i & 1);
    f32x16 zacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) zacc[r] = 0.f;
#ifdef AID_ABLATIONS
    if (!(p.abl & 4))
#endif
    {
#pragma unroll
        for (int i = 0; i < 8; ++i) sc[i & 1] = mfma32(kf[i >> 1][i & 1], qf[i >> 1], (i >> 1) ? sc[i & 1] : zacc);      // m = 0
        pin();
#pragma unroll
        for (int i = 0; i < 8; ++i) kf[i >> 1][i & 1] = lds_k(1, i >> 1, i & 1);      // nt >= 2
    }
    settle();
    slot_barrier();
#ifdef AID_ABLATIONS
    if (p.abl & 32) tl[3] = clock64();
#endif
    // Items: for (;;) { plan the next item, request its Q rows; walk this item's tiles; finish it; adopt the next }.  Every tile t of an
    // item gets V(t) and M(t + 1); behind the LAST tile "t + 1" is tile 0 of the next item (its Q fragments are read back from LDS
    // right before that M slot) or, for the workgroup's last item, stale ring bytes whose scores nobody reads.
#pragma nounroll
    for (int j = 0;; ++j) {
        const int vbn = item_lid(j + 1);
        has_next = vbn < n_items;
        if (has_next) {
            plan(vbn);
            dma_q_next();
        }
        // One pass per key segment (`nounroll`: one copy of the body); inside, eight tiles per trip: t & 7 — the ring stage of every
        // DMA and fragment read — is a compile-time constant in each copy (segments of a multi-segment frame, and every item of a
        // persistent launch, are whole trips).
#pragma nounroll
        for (int sgi = 0; sgi < (MODE == AID_MODE_PLAIN ? 1 : nseg); ++sgi) {
            if (MODE == AID_MODE_OUTER) {
                if (sgi == park_at) park();
                if (sgi == swap_at) swap_sides();
            }
            const int t_end = (sgi + 1) * nt;
            for (int t8 = sgi * nt; t8 < t_end; t8 += 8) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int t = t8 + i;
                    if (t >= t_end) break;
                    PP_STAMP(3);
                    vslot(t, i == 0);
                    if (i == 7 && __builtin_expect(has_next && t == NT - 1, 0)) {   // (items with a successor are whole trips) the M slot behind the item's last tile forms the next item's S(0):
#pragma unroll                                                  // its Q rows, and a zero row reference like a fresh workgroup's
                        for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const T8*>(qlds + ks * 1024 + lane * 16);
                        scale_q();
                        m = 0.f;                                // (the finished item needs O and the row sums only)
#pragma unroll
                        for (int r = 0; r < 16; ++r) cneg[r] = 0.f;
                        asm volatile("" : "+v"(cneg));
                    }
                    PP_STAMP(0);
                    slot_barrier();
                    PP_STAMP(1);
                    mslot(i, (i + 2) & 7);
                    tie();
                    PP_STAMP(2);
                    slot_barrier();
                }
            }
        }
        finish();
        if (!has_next) break;
        // the next item: its S(0) is in `sc`; everything else starts over
        fresh = true;
        set_l(0.f);
        pl = 0.f;
        pm = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
        if (MODE == AID_MODE_OUTER) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { po[0][r] = 0.f; po[1][r] = 0.f; }
        }
        adopt();
        dseg = 0;                                               // the DMA stream is LEAD tiles into this item's first segment already
    }
    wait_vm<0>();                            // (the wrapped requests behind the last item: nobody reads them)
    if (grp == 0) slot_barrier();                               // both groups pass the same number of barriers
}

}  // namespace

// May the ping-pong kernel run the single-segment frames of this call?  d = 64, whole 64-key tiles, at least two of them.
bool attn_pp_supported(const AidAttnArgs& a) {
    // segment rows, heads and the next item's Q rows ride in 32-bit offsets of descriptors with num_records 0x7fffffff: every tensor
    // the kernel reads through one stays below 2 GB (k / vt hold n_kv rows, k2 / vt2 and q one row per FRAME)
    const int64_t lim = 1ll << 31;
    const int64_t kb = (int64_t)a.n_kv * a.k_fs * 2, vb = (int64_t)a.n_kv * a.vt_fs * 2;
    const int64_t k2b = a.mode == AID_MODE_INNER ? (int64_t)a.n_frames * a.k_fs * 2 : 0;
    const int64_t v2b = a.mode == AID_MODE_INNER ? (int64_t)a.n_frames * a.vt_fs * 2 : 0;
    const int64_t qb = (int64_t)a.n_frames * a.q_fs * 2, ob = (int64_t)a.n_frames * a.o_fs * 2;
    const uintptr_t al = reinterpret_cast<uintptr_t>(a.k) | reinterpret_cast<uintptr_t>(a.vt) | reinterpret_cast<uintptr_t>(a.k2) |
                         reinterpret_cast<uintptr_t>(a.vt2) | reinterpret_cast<uintptr_t>(a.out);   // LDS-DMA / output rows: 16-byte pieces
    return a.d == 64 && a.l % PKT == 0 && a.l >= 2 * PKT && a.ldk % 8 == 0 && a.ldvt % 8 == 0 && a.k_fs % 8 == 0 && a.vt_fs % 8 == 0 &&
           (al & 15) == 0 && kb < lim && vb < lim && k2b < lim && v2b < lim && qb < lim && ob < lim && a.ldo % 8 == 0 && a.o_fs % 8 == 0;
}

hipError_t attn_pp_launch(const AidAttnArgs& a, hipStream_t stream, bool multi) {
    AttnPPParams p;
    p.a = a;
    p.multi = multi ? 1 : 0;
    p.nqb = (a.s + 255) / 256;
    p.c2 = a.softmax_scale * 1.4426950408889634f;
#ifdef AID_ABLATIONS
    p.abl = tune(TUNE_ATTN_RES_CHUNKS) > 100 ? tune(TUNE_ATTN_RES_CHUNKS) - 100 : 0;
#endif
    const size_t smem = (size_t)PNS * PSTAGE + 8 * 4096;        // ring + the next item's Q rows
    static PerDevice<int> attr_set[2];
    const int ti = a.dtype == AID_DTYPE_F16 ? 0 : 1;
    int* done = attr_set[ti].slot();
    if (!done) return hipErrorInvalidDevice;
    if (a.mode < AID_MODE_PLAIN || a.mode > AID_MODE_OUTER) return hipErrorInvalidValue;
    const void* fns[2][3] = {
        {reinterpret_cast<const void*>(&aid_attn_pp_kernel<f16, AID_MODE_PLAIN>), reinterpret_cast<const void*>(&aid_attn_pp_kernel<f16, AID_MODE_INNER>),
         reinterpret_cast<const void*>(&aid_attn_pp_kernel<f16, AID_MODE_OUTER>)},
        {reinterpret_cast<const void*>(&aid_attn_pp_kernel<bf16, AID_MODE_PLAIN>), reinterpret_cast<const void*>(&aid_attn_pp_kernel<bf16, AID_MODE_INNER>),
         reinterpret_cast<const void*>(&aid_attn_pp_kernel<bf16, AID_MODE_OUTER>)}};
    static_assert(AID_MODE_PLAIN == 0 && AID_MODE_INNER == 1 && AID_MODE_OUTER == 2, "mode index");
    const void* fn = fns[ti][a.mode];
    if (!(*done & (1 << a.mode))) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        *done |= 1 << a.mode;
    }
    const int items = p.nqb * a.n_frames * a.heads;
    // persistent: the kernel owns every frame of the call (no early exits), every item is whole 8-tile trips, more items than CUs
    static PerDevice<int> cus;
    int* ncu = cus.slot();
    if (!ncu) return hipErrorInvalidDevice;
    if (*ncu == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) *ncu = 256;
    }
    const int knob = tune(TUNE_ATTN_PIPE);                      // development: 0 = one item per workgroup
    // (measured: plain S = 4096 582 -> 560 us, S = 1024 108 -> 94, inner 808 -> 800 / 129 -> 122; the 3 : 1 item mix of an OUTER call is
    //  balanced as well by the hardware's dynamic dispatch as by the static snake order: 1048 vs 1047, 160 vs 160 — one item per workgroup)
    p.persist = ((multi || a.mode == AID_MODE_PLAIN) && (a.l / PKT) % 8 == 0 && items > *ncu && *ncu % 8 == 0 && knob != 0) ? 1 : 0;
    // which frames walk several key segments (host's view: the AID frames of a call sit in front of its PLAIN riders, a fused call's
    // end-point frames walk their own keys only): a hint for the static balance of the persistent walk
    const int n_aid = a.n_frames - a.n_plain;
    p.hv_lo = p.hv_hi = 0;
    p.hv_units = 1;
    if (a.mode == AID_MODE_OUTER && n_aid > 2) { p.hv_lo = 1; p.hv_hi = n_aid - 1; p.hv_units = a.fused ? 3 : 2; }
    if (a.mode == AID_MODE_INNER && a.fused && n_aid > 2) { p.hv_lo = 1; p.hv_hi = n_aid - 1; p.hv_units = 2; }
    const int grid = p.persist ? *ncu : items;
    void* kargs[] = {const_cast<AttnPPParams*>(&p)};
    return hipLaunchKernel(fn, dim3(grid), dim3(512), kargs, smem, stream);
}

}  // namespace aid
