// Ping-pong attention core for head dim 64 and whole 64-key tiles — plain attention (de-activated passes, reference
// interpolation.py:581-584), the PLAIN riders of a batched-CFG call, and the interpolated frames of INNER / OUTER calls
// (interpolation.py:626-664, 760-790) as ONE tile stream over the frame's key segments.  Same arithmetic as aid_attn_kernel (swapped
// products, -m folded into the score MFMAs' accumulator, lazy row reference; the row sums here are VALU dot products of the rounded P,
// not a ones-row block); what differs is WHO does what WHEN:
//
//   One workgroup = 8 waves x 32 query rows of one (frame, head); waves w and w + 4 share a SIMD.  Waves 0-3 and waves 4-7
//   run the same program ONE BARRIER APART, and the program alternates two slots per 64-key tile:
//     M(t)   every MFMA of the tile in one burst: S(t) = K(t) Q'^T - m (8), then O^T += V^T(t-1) P(t-1)^T (8); in their shadow,
//            one per MFMA and in pinned program order, the 16 operand-fragment reads: V^T(t-1) beside the score MFMAs, K(t+1)
//            beside the PV MFMAs into the registers the score MFMAs released.  No VALU instruction.
//     V(t)   the VALU half: head-room check, P(t) = 2^S(t), rounding to the storage type, row sums (v_dot2c); plus this wave's two
//            LDS-DMA pieces of tile t + 6 and the counted wait that retires its pieces of tile t + 3
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
// aid_attn_fwd's default rule: fused OUTER from 1024 keys, everything else from 2048.
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
    int32_t multi;                      // 1: this launch also runs the three-segment frames of a fused OUTER call (it is the only launch)
    float   c2;                         // softmax_scale * log2(e)
    int32_t abl;                        // development builds (-DAID_ABLATIONS): timing ablations, results are garbage
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

template <typename T>
__global__ __launch_bounds__(512) void aid_attn_pp_kernel(const AttnPPParams p) {
    typedef typename Vec<T>::v8 T8;
    typedef typename Vec<T>::v4 T4;
    constexpr int D = 64;
    constexpr float XTH = std::is_same<T, f16>::value ? 15.0f : 60.0f;      // head-room (log2) of P = 2^x in the storage type
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

    // [head][frame][q block]: a (frame, head)'s blocks share an L2; three-segment frames first inside every XCD's range
    const int n_heavy = a.n_frames - a.n_plain;
    const int lid = (p.multi && a.n_plain > 0 && n_heavy > 0)
                        ? heavy_first(blockIdx.x, gridDim.x, n_heavy * p.nqb, a.n_frames * p.nqb)
                        : xcd_remap(blockIdx.x, gridDim.x);
    const int qb = lid % p.nqb;
    const int fr = (lid / p.nqb) % a.n_frames;
    const int h = lid / (p.nqb * a.n_frames);
    const int q0 = (qb * 8 + wave) * 32;
    const int kvf = a.kv_map ? a.kv_map[fr] : fr;
    // Key segments of this frame (the same decisions aid_attn_kernel takes, on the same device coefficients):
    //   single  — PLAIN call, negative coefficient (PLAIN rider of a batched-CFG call), fused END-POINT frame: own keys only.
    //   OUTER   — reference interpolation.py:626-664: softmaxes over [own ; begin] and [own ; end] (fused) or over begin and end
    //             (pure), outputs mixed (1 - c) : c.  The walk is own -> begin -> end as ONE tile stream; the state after the own
    //             segment serves both sides: it is PARKED in registers (po / pl / pm) when the begin segment starts and swapped
    //             back in when the end segment starts, while the parked registers take the finished begin side (pure: the parked
    //             state is the empty one).  c == 0 / c == 1: the zero-weighted side is dropped.
    //   INNER   — interpolation.py:760-790: one softmax over [own ; mix] (fused) or mix (pure); mix = the interpolated keys / values
    //             aid_lerp_kv wrote to k2 / vt2 (row = frame), or the begin / end row itself for c == 0 / 1.
    // A launch that does not own the whole call (p.multi == 0: the split launches behind ATTN_V2 = 1) runs single frames only.
    int nseg = 1, seg0 = kvf, seg1 = 0, seg2 = 0;               // key / value rows of the segments
    int t2mask = 0;                                             // bit s: segment s reads k2 / vt2
    int park_at = -1, swap_at = -1;                             // segment index in front of which the state is parked / swapped
    const int row_b = a.begin, row_e = a.end;
    float w_b = 0.f, w_e = 1.f;
    if (a.mode != AID_MODE_PLAIN) {
        const float cf = a.coef[fr];
        const bool single = cf < 0.f || (a.fused && ((cf == 0.f && kvf == row_b) || (cf == 1.f && kvf == row_e)));
        if (!single) {
            if (!p.multi) return;
            // (arithmetic, not `c == 1 ? a.end : a.begin`: a select between two FIELDS of the by-value argument struct becomes a
            //  select between their addresses and hipcc then keeps the whole struct in scratch)
            const bool both = cf != 0.f && cf != 1.f;
            const int side = row_b + (cf == 1.f ? 1 : 0) * (row_e - row_b);       // the end-point row of a one-sided frame
            const int fz = a.fused ? 1 : 0;
            if (a.mode == AID_MODE_OUTER) {
                nseg = fz + (both ? 2 : 1);
                const int first = both ? row_b : side;
                seg0 = fz * kvf + (1 - fz) * first;
                seg1 = fz * first + (1 - fz) * row_e;
                seg2 = row_e;
                if (both) { w_b = 1.f - cf; w_e = cf; park_at = fz ? 1 : -1; swap_at = fz + 1; }
            } else {
                nseg = fz + 1;
                const int mix = both ? fr : side;
                seg0 = fz * kvf + (1 - fz) * mix;
                seg1 = mix;
                t2mask = both ? (fz ? 2 : 1) : 0;
            }
        }
    }

    // ---- Q fragments (B operand of the swapped product) -----------------------------------------
    T8 qf[4];
    {
        const int qr = min(q0 + l31, a.s - 1);                  // rows past the end are clamped, never stored
        const T* qrow = reinterpret_cast<const T*>(a.q) + (int64_t)fr * a.q_fs + (int64_t)qr * a.ldq + h * D;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qf[ks] = *reinterpret_cast<const T8*>(qrow + ks * 16 + hi * 8);
            if (!a.q_prescaled) {
                f32x8 t = up8<T>(qf[ks]);
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] *= p.c2;
                qf[ks] = cvt8<T>(t);
            }
        }
    }

    // ---- DMA addressing: this wave's piece (8 rows x 128 B) of every K tile and of every V^T tile ------------
    // one descriptor per tensor (this head's columns / rows); the key / value ROW of a segment goes into the scalar offset
    // (attn_pp_supported: the tensors are below 2 GB)
    const T* Kg = reinterpret_cast<const T*>(a.k) + h * D;
    const T* Vg = reinterpret_cast<const T*>(a.vt) + (int64_t)(h * D) * a.ldvt;
    const Rsrc rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(Kg), 0, 0x7fffffff, 0x00020000);
    const Rsrc rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(Vg), 0, 0x7fffffff, 0x00020000);
    const int nt = a.l / PKT;                                   // tiles per segment
    const int NT = nseg * nt;                                   // tiles of this workgroup's stream
    const int ks0 = seg0 * (int)a.k_fs * 2, ks1 = seg1 * (int)a.k_fs * 2, ks2 = seg2 * (int)a.k_fs * 2;
    const int vs0 = seg0 * (int)a.vt_fs * 2, vs1 = seg1 * (int)a.vt_fs * 2, vs2 = seg2 * (int)a.vt_fs * 2;
    // INNER: the interpolated keys / values live in their own tensors (same layout, row = frame)
    const T* K2g = reinterpret_cast<const T*>(a.k2) + h * D;
    const T* V2g = reinterpret_cast<const T*>(a.vt2) + (int64_t)(h * D) * a.ldvt;
    const Rsrc rk2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(K2g), 0, 0x7fffffff, 0x00020000);
    const Rsrc rv2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(V2g), 0, 0x7fffffff, 0x00020000);
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
        if (d2) {                                               // (a branch, not a select between the two descriptors)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rk2, (__attribute__((address_space(3))) void*)dst, 16, kvo, dk, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rv2, (__attribute__((address_space(3))) void*)(dst + PNS * PTILE), 16, vvo, dv, 0, 0);
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (__attribute__((address_space(3))) void*)dst, 16, kvo, dk, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (__attribute__((address_space(3))) void*)(dst + PNS * PTILE), 16, vvo, dv, 0, 0);
        }
        dk += kstep;
        dv += PKT * 2;
        if (--dleft == 0) {                                     // next segment (arithmetic on values, see above)
            ++dseg;
            dk = ks1 + (dseg >= 2 ? ks2 - ks1 : 0);
            dv = vs1 + (dseg >= 2 ? vs2 - vs1 : 0);
            d2 = ((t2mask >> dseg) & 1) != 0;
            dleft = nt;
        }
    };

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
    T8 pf[4];
    f32x16 cneg;                        // -m as an accumulator block (C operand of a tile's first score MFMAs), rebuilt when m moves
    f32x16 po[2];                       // parked: O of the own-keys state while the begin side runs, then the finished begin side
    float pl = 0.f, pm = 0.f;           // parked row sum (this lane's register of the row-sum block) and row reference
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; po[0][r] = 0.f; po[1][r] = 0.f; cneg[r] = 0.f; }
    asm volatile("" : "+v"(cneg));
#pragma unroll
    for (int i = 0; i < 4; ++i) pf[i] = zero8<T>();

#ifdef AID_ABLATIONS
    // 16: slot timing — shader cycles of [V work | wait at the barrier behind it | M work | wait at the barrier behind it], summed over the
    // tiles and written over the output rows (lane 0 of every wave: four floats = cycles per tile)
    long long tmark = 0;
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
            kf[kk][w] = lds_k(sk, kk, w);                       // (kf[kk][w] was released by score MFMA 2 kk + w, eight or more MFMAs ago)
            pin();
        }
    };

    // V slot: P(t) = 2^S(t) (row reference raised first when the head-room of the storage type is exceeded), rounded to the storage
    // type; this wave's two DMA pieces of tile t + LEAD; the counted wait that retires its pieces of tile t + 3
    constexpr int LEAD = 6;
    auto retire = [&](int r) __attribute__((always_inline)) {   // r = tiles that may stay in flight behind the one being retired
        if (r >= 3)      wait_vm<6>();
        else if (r == 2) wait_vm<4>();
        else if (r == 1) wait_vm<2>();
        else             wait_vm<0>();
    };
    auto vslot = [&](int t) __attribute__((always_inline)) {
        const bool issue = t + LEAD < NT;
#ifdef AID_ABLATIONS
        if (p.abl & 1) {                                        // 1: no VALU work in the V slot
            if (!(p.abl & 2) && issue) dma_next((t + LEAD) & (PNS - 1));
            fresh = false;
            retire(NT - 4 - t);
            return;
        }
#endif
        if (issue
#ifdef AID_ABLATIONS
            && !(p.abl & 2)
#endif
        ) dma_next((t + LEAD) & (PNS - 1));
        float xm = fmaxf(sc[0][0], sc[0][1]);
#ifdef AID_ABLATIONS
        if (!(p.abl & 8))                                       // 8: no head-room check (the maximum chain)
#endif
#pragma unroll
        for (int i = 1; i < 16; ++i) xm = fmaxf(fmaxf(xm, sc[i >> 3][(2 * i) & 15]), sc[i >> 3][(2 * i + 1) & 15]);
        if (fresh || __any(xm > XTH)) {
            // slow path (first tile of the row, or a score out-grew the head-room): move the reference to the row maximum,
            // rescale O (its ones row = l included) and shift this tile's arguments; PV(t - 1) is complete, O is at rest
            const float rowmax = max_halves(xm);
            const float shift = fresh ? rowmax : fmaxf(rowmax, 0.f);
            const float alpha = __builtin_amdgcn_exp2f(-shift);
            m += shift;
#pragma unroll
            for (int r = 0; r < 16; ++r) cneg[r] = -m;
            asm volatile("" : "+v"(cneg));                      // (opaque: keeps hipcc from rebuilding the block in every M slot)
#pragma unroll
            for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
            lsum *= alpha;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[b][r] -= shift;
            fresh = false;
        }
        // lane (q, hi): sc[b][r] belongs to key 32 b + 16 (r >> 3) + 8 hi + (r & 7) of the tile = k-step 2 b + (r >> 3) of PV
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                f32x8 pv;
#pragma unroll
                for (int e = 0; e < 8; ++e) pv[e] = __builtin_amdgcn_exp2f(sc[b][8 * u + e]);
                pf[2 * b + u] = cvt8<T>(pv);
            }
        // row sums of the rounded P: v_dot2c against (1, 1), two keys per instruction (the matrix pipe's ones-row block cost 4 of 20 MFMAs)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32x4 w4 = __builtin_bit_cast(u32x4, pf[i]);
#pragma unroll
            for (int e = 0; e < 4; ++e) lsum = dot2_ones<T>(w4[e], lsum);
        }
        if (issue) wait_vm<6>();                                // steady state: the three tiles behind t + 3 stay in flight
        else       retire(NT - 4 - t);
    };

    // Segment boundaries of a two-sided frame.  S(t) of the next segment's first tile is in `sc` and PV(t - 1) closed the segment
    // before it: O, l, m are at rest.
    auto park = [&]() __attribute__((always_inline)) {          // own keys done: park the state, the begin side continues on it
#pragma unroll
        for (int r = 0; r < 16; ++r) { po[0][r] = o[0][r]; po[1][r] = o[1][r]; }
        pl = lsum;
        pm = m;
    };
    auto swap_sides = [&]() __attribute__((always_inline)) {    // begin side done: keep (1 - c) O_b / l_b, resume the own-keys state
        const float lrow = lsum + other_half(lsum);             // (all lanes take part in the exchange)
        const float wb = w_b / lrow;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float rb = o[d][r] * wb;
                o[d][r] = po[d][r];
                po[d][r] = rb;
            }
        lsum = pl;
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

    // fence between a slot's last MFMAs and the VALU code of the next slot that reads their results (20 wait states; the hazard
    // recogniser does not see across the barrier's asm / branches — aid_attn.hip has the history)
    auto settle = [&]() __attribute__((always_inline)) {
        asm volatile("" : "+v"(sc[0]), "+v"(sc[1]));
        asm volatile("" : "+v"(o[0]), "+v"(o[1]));
        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
        asm volatile("" : "+v"(sc[0]), "+v"(sc[1]));
        asm volatile("" : "+v"(o[0]), "+v"(o[1]));
    };

    // Schedule.  Group g runs M(t) in interval 2 t + g and V(t) in 2 t + 1 + g (M(t) computes S(t) and the PV product of tile t - 1).
    // Tile T is read in M(T - 1) (K, for the next slot's scores) and in M(T + 1) (V^T) = intervals 2 T - 2 .. 2 T + 3; every read is
    // complete (lgkmcnt) before its reader's next barrier, so stage T & 7 is free from interval 2 T + 4.  Tile U = T + 8 lands there:
    // requested in V(U - 6) (interval 2 U - 11 + g >= 2 T + 5), retired by the counted wait of V(U - 3) — three tile periods later —
    // and barrier-published at the end of interval 2 U - 4 at the latest, two intervals before M(U - 1) reads it.
    // ---- prologue: pieces of tiles 0 .. 5 requested, 0 .. 2 retired and published; S(0); fragments of K(1) ----------------
#pragma unroll
    for (int t = 0; t < LEAD; ++t)
        if (t < NT) dma_next(t);
#ifdef AID_ABLATIONS
    if (p.abl & 32) { tl[1] = clock64(); asm volatile("" :: "v"(qf[0]), "v"(qf[3])); }
#endif
    retire(NT - 3);
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
    // One pass per key segment (`nounroll`: one copy of the body); inside, eight tiles per trip: t & 7 — the ring stage of every DMA
    // and fragment read — is a compile-time constant in each copy (segments of a multi-segment frame are multiples of eight tiles).
#pragma nounroll
    for (int sgi = 0; sgi < nseg; ++sgi) {
        if (sgi == park_at) park();
        if (sgi == swap_at) swap_sides();
        const int t_end = min((sgi + 1) * nt, NT - 1);          // V(t) + M(t + 1) for the tiles t of this segment; the stream's last
        for (int t8 = sgi * nt; t8 < t_end; t8 += 8) {          // tile is finished behind the loop
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int t = t8 + j;
                if (t >= t_end) break;
                PP_STAMP(3);
                vslot(t);
                PP_STAMP(0);
                slot_barrier();
                PP_STAMP(1);
                mslot(j, (j + 2) & 7);                          // (the K(t + 2) reads past the last tile fetch stale ring bytes, unused)
                settle();
                PP_STAMP(2);
                slot_barrier();
            }
        }
    }
    vslot(NT - 1);
    slot_barrier();
    {                                                           // O += V^T(NT - 1) P(NT - 1)^T
        const int so = ((NT - 1) & 7) * PTILE;
#pragma unroll
        for (int i = 0; i < 8; ++i) vf[i >> 1][i & 1] = *reinterpret_cast<const T8*>(smem + vad[i >> 1] + so + (i & 1) * 4096);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int d = 0; d < 2; ++d) o[d] = mfma32(vf[kk][d], pf[kk], o[d]);
        }
    }
    settle();
    slot_barrier();
    if (grp == 0) slot_barrier();                               // both groups pass the same number of barriers

    // ---- finish: O / l, lane (q = l31, hi) holds dv = 32 d + 8 g + 4 hi + {0..3} ----------------------
#ifdef AID_ABLATIONS
    if (p.abl & 32) {
        tl[4] = clock64();
        if (lane == 0 && q0 < a.s) {
            float* dbg = reinterpret_cast<float*>(reinterpret_cast<T*>(a.out) + (int64_t)fr * a.o_fs + (int64_t)q0 * a.ldo + h * D);
            for (int i = 1; i < 5; ++i) dbg[i - 1] = (float)(tl[i] - tl[0]);
            dbg[4] = (float)(tl[0] & 0xffffff);
        }
        return;
    }
    if (p.abl & 16) {
        if (lane == 0 && q0 < a.s) {
            float* dbg = reinterpret_cast<float*>(reinterpret_cast<T*>(a.out) + (int64_t)fr * a.o_fs + (int64_t)q0 * a.ldo + h * D);
            for (int i = 0; i < 4; ++i) dbg[i] = tacc[i] / (float)(NT - 1);
        }
        return;
    }
#endif
    const float inv = w_e / (lsum + other_half(lsum));          // (w_e = 1 unless this frame mixes two sides)
    const int q = q0 + l31;
    if (q < a.s) {
        const float osc = a.out_scale * (a.frame_scale ? a.frame_scale[fr] : 1.f);
        T* orow = reinterpret_cast<T*>(a.out) + (int64_t)fr * a.o_fs + (int64_t)q * a.ldo + h * D;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int dv = 32 * d + 8 * g + 4 * hi;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (o[d][4 * g + e] * inv + po[d][4 * g + e]) * osc;      // po: the begin side, or zero
                if (a.accumulate) v += up4<T>(*reinterpret_cast<const T4*>(orow + dv));
                *reinterpret_cast<T4*>(orow + dv) = cvt4<T>(v);
            }
    }
}

}  // namespace

// May the ping-pong kernel run the single-segment frames of this call?  d = 64, whole 64-key tiles, at least two of them.
bool attn_pp_supported(const AidAttnArgs& a) {
    const int64_t kb = (int64_t)a.n_kv * a.k_fs * 2, vb = (int64_t)a.n_kv * a.vt_fs * 2;    // segment rows ride in 32-bit scalar offsets
    const uintptr_t al = reinterpret_cast<uintptr_t>(a.k) | reinterpret_cast<uintptr_t>(a.vt) | reinterpret_cast<uintptr_t>(a.k2) |
                         reinterpret_cast<uintptr_t>(a.vt2);                               // LDS-DMA moves 16-byte pieces
    return a.d == 64 && a.l % PKT == 0 && a.l >= 2 * PKT && a.ldk % 8 == 0 && a.ldvt % 8 == 0 && a.k_fs % 8 == 0 && a.vt_fs % 8 == 0 &&
           (al & 15) == 0 && kb < (1ll << 31) && vb < (1ll << 31);
}

hipError_t attn_pp_launch(const AidAttnArgs& a, hipStream_t stream, bool multi) {
    AttnPPParams p;
    p.a = a;
    p.multi = multi ? 1 : 0;
    p.nqb = (a.s + 255) / 256;
    p.c2 = a.softmax_scale * 1.4426950408889634f;
    p.abl = tune(TUNE_ATTN_RES_CHUNKS) > 100 ? tune(TUNE_ATTN_RES_CHUNKS) - 100 : 0;      // development builds only
    const size_t smem = (size_t)PNS * PSTAGE;
    static PerDevice<bool> attr_set[2];
    const int ti = a.dtype == AID_DTYPE_F16 ? 0 : 1;
    bool* done = attr_set[ti].slot();
    if (!done) return hipErrorInvalidDevice;
    const void* fn = ti == 0 ? reinterpret_cast<const void*>(&aid_attn_pp_kernel<f16>)
                             : reinterpret_cast<const void*>(&aid_attn_pp_kernel<bf16>);
    if (!*done) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        *done = true;
    }
    const int grid = p.nqb * a.n_frames * a.heads;
    if (ti == 0) hipLaunchKernelGGL(aid_attn_pp_kernel<f16>, dim3(grid), dim3(512), smem, stream, p);
    else         hipLaunchKernelGGL(aid_attn_pp_kernel<bf16>, dim3(grid), dim3(512), smem, stream, p);
    return hipGetLastError();
}

}  // namespace aid
