// Ping-pong attention for SHORT key streams, head dim 64: the text cross-attention of the stacks (77 keys per context; reference
// interpolation.py:623-664 with encoder_hidden_states given, :581-584 de-activated) — PLAIN calls, the PLAIN riders of a batched-CFG
// call, and the one / two / three key segments of a fused or pure OUTER frame.
//
// Why a second kernel.  aid_attn_pp.hip streams the key tiles of ONE (frame, head, 256 query rows) item past resident Q rows; with 77
// keys an item is one to six tiles long, and everything that kernel pays per ITEM (scalar planning, coefficient loads, state reset,
// normalise + store) or assumes per item (whole 8-tile trips so that the ring stage is a compile-time constant; a DMA stream that
// runs at most one item ahead) would dominate.  The program-order kernel (aid_attn.hip) runs these launches as 2240 - 4480 short
// workgroups, each one latency chain: 30 - 77 us per launch for 15 us of HBM time (profiles/r03_breakdown_sdxl.txt, 0.10 of the MFMA
// peak).  Here the same slot machinery — two wave groups one barrier apart, M slot = every MFMA of a tile with the fragment reads in
// their shadow, V slot = exponentials, row sums, this wave's LDS-DMA pieces — runs ONE CONTINUOUS TILE STREAM PER WORKGROUP across
// all the items the workgroup owns (the balanced static deal of aid_attn_pp.hip), so a workgroup's start-up is paid once per launch:
//   * the ring stage of a tile is its GLOBAL index in the workgroup's stream mod 8 — a run-time value (eight v_add per M slot) —
//     so items of any length follow each other without gaps;
//   * the DMA walker is independent of the compute walker: it plans items as far ahead as its six-tile lead needs (up to six
//     one-tile items); planned items are queued in the lanes of two VGPRs (v_writelane / v_readlane, wave-uniform);
//   * per-frame data (coefficient, key / value row, output scale) are read ONCE into the lanes of three VGPRs: planning an item costs
//     no memory access; the item deal advances incrementally (no division per item);
//   * keys past L in the last tile of a segment are masked by 16-key groups (a group past L costs eight v_mov, the one straddling L
//     eight compare + select pairs).  The caller provides K with rows and V^T with columns up to the tile boundary
//     (AidAttnArgs.kv_padded: finite K rows, ZERO V^T columns) — the step-invariant text keys / values are projected once per run
//     into that layout (ops.project_kv(padded=True)), every other caller stays on aid_attn_kernel.
// Arithmetic identical to aid_attn_pp.hip (swapped products, -m in the score accumulator, lazy reference with the head-room test on
// the tile's row sum, v_dot2c row sums, OUTER's parked own-keys state).
#include <type_traits>

#include "aid_common.hpp"
#include "aid_kernels.hpp"

namespace aid {

namespace {

constexpr int XKT = 64;                 // keys per tile
constexpr int XTILE = 8192;             // bytes per tile: K [64 keys][128 B] or V^T [64 channels][128 B]
constexpr int XNS = 8;                  // ring depth (K tiles in [0, 64 KB), V^T tiles in [64 KB, 128 KB))
constexpr int XNONE = 0x7fffffff;       // "no item"

struct AttnXSParams {
    AidAttnArgs a;
    int32_t nqb;                        // 256-row q blocks per (frame, head)
    uint32_t nqb_magic;                 // floor(2^32 / nqb) + 1: pair / nqb for pair < 65536 as one s_mul_hi
    int32_t nt, rem;                    // tiles per key segment; valid keys in a segment's last tile (64 = none masked)
    int32_t hv_lo, hv_hi, hv_units;     // frames [hv_lo, hv_hi) are HEAVY (hv_units key segments): balance hint, see aid_attn_pp.hip
    float   c2;                         // softmax_scale * log2(e)
};

typedef __amdgpu_buffer_rsrc_t Rsrc;

template <typename T>
__device__ __forceinline__ float xdot2(uint32_t w, float acc);
template <>
__device__ __forceinline__ float xdot2<bf16>(uint32_t w, float acc) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    b2 one;
    one[0] = (__bf16)1.0f; one[1] = (__bf16)1.0f;
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2, w), one, acc, false);
}
template <>
__device__ __forceinline__ float xdot2<f16>(uint32_t w, float acc) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    h2 one;
    one[0] = (_Float16)1.0f; one[1] = (_Float16)1.0f;
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, w), one, acc, false);
}
template <int N>
__device__ __forceinline__ void xwait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// s_waitcnt vmcnt(n) for a run-time n that is one of 6 + 4 i (i = 0 .. 6) or 2 / 4 / 10 / 12 (the immediate must be a literal)
__device__ __forceinline__ void xwait_vm_dyn(int n) {
    switch (n) {
        case 2:  xwait_vm<2>(); break;
        case 4:  xwait_vm<4>(); break;
        case 6:  xwait_vm<6>(); break;
        case 10: xwait_vm<10>(); break;
        case 12: xwait_vm<12>(); break;
        case 14: xwait_vm<14>(); break;
        case 18: xwait_vm<18>(); break;
        case 22: xwait_vm<22>(); break;
        case 26: xwait_vm<26>(); break;
        case 30: xwait_vm<30>(); break;
        default: xwait_vm<0>(); break;
    }
}
__device__ __forceinline__ void xbarrier() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ int rdlane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ float rdlane(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }

template <typename T, int MODE>
__global__ __launch_bounds__(512) void aid_attn_xs_kernel(const AttnXSParams p) {
    typedef typename Vec<T>::v8 T8;
    typedef typename Vec<T>::v4 T4;
    constexpr int D = 64;
    constexpr float XTH = std::is_same<T, f16>::value ? 15.0f : 60.0f;      // head-room (log2) of P = 2^x in the storage type
    constexpr float NEG = -1.0e30f;                                        // score of a masked key: 2^NEG = 0
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const AidAttnArgs& a = p.a;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int l31 = lane & 31, hi = lane >> 5;
    const int nt = p.nt;
    const bool ragged = p.rem < XKT;

    // ---- per-frame tables in lanes (n_frames <= 64): coefficient, key / value row, output scale ---------------------------------
    float tb_coef, tb_fs;
    int tb_kv;
    {
        const int f = min(lane, a.n_frames - 1);
        tb_coef = (MODE != AID_MODE_PLAIN && a.coef) ? a.coef[f] : -1.f;
        tb_kv = a.kv_map ? a.kv_map[f] : f;
        tb_fs = a.frame_scale ? a.frame_scale[f] : 1.f;
    }

    // ---- the deal (aid_attn_pp.hip: an XCD owns whole (head, q block) pairs; heavy items cyclically, light items level the rest), as an
    // incremental planner: state (phase, count, t = index in the phase's list, fi = t / npx, rm = t % npx) ---------------------------
    const int xcd = blockIdx.x & 7, wx = blockIdx.x >> 3, WX = gridDim.x >> 3;
    const int nhf = p.hv_hi - p.hv_lo;
    int pb, npx;
    {
        const int np = a.heads * p.nqb, q = np >> 3, r = np & 7;
        pb = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        npx = q + (xcd < r ? 1 : 0);
    }
    if (npx == 0) return;                                       // fewer pairs than XCDs: nothing for this workgroup
    const int nhx = npx * nhf, nlx = npx * (a.n_frames - nhf);
    const int hq_ = nhx / WX, hr = nhx % WX;
    const int hc = hq_ + (wx < hr ? 1 : 0);
    const int ndef = hr > 0 ? WX - hr : 0;
    const int p2 = min(nlx, p.hv_units * ndef);
    const int c1 = (hr > 0 && wx >= hr) ? max(min(p.hv_units * (wx - hr + 1), p2) - p.hv_units * (wx - hr), 0) : 0;
    const int WXq = WX / npx, WXr = WX % npx;
    // Every item of this workgroup is planned HERE, once, into the lanes of two VGPRs (at most 62 items per workgroup, checked by
    // the host): qi = fr | h << 8 | qb << 16 (XNONE behind the last one), qs = nseg | seg0 << 4 | seg1 << 12 | two-sided << 20.
    // The walkers below only read lanes (v_readlane with a scalar index); the planner's code and state exist once, outside the stream.
    int qi = XNONE, qs = 0;
    const int row_b = a.begin, row_e = a.end;
    {
        int pl_phase = 0, pl_cnt = 0, pl_t = wx, pl_fi = wx / npx, pl_rm = wx % npx;
#pragma nounroll
        for (int jp = 0; jp < 63; ++jp) {
            int fr = -1;
#pragma nounroll
            for (int round = 0; round < 3 && fr < 0 && pl_phase < 3; ++round) {
                if (pl_phase == 0) {
                    if (pl_cnt < hc) fr = p.hv_lo + pl_fi;
                    else { pl_phase = 1; pl_cnt = 0; pl_t = p.hv_units * (wx - hr); pl_fi = pl_t / npx; pl_rm = pl_t % npx; }
                } else if (pl_phase == 1) {
                    if (pl_cnt < c1) fr = pl_fi < p.hv_lo ? pl_fi : pl_fi + nhf;
                    else { pl_phase = 2; pl_cnt = 0; pl_t = p2 + wx; pl_fi = pl_t / npx; pl_rm = pl_t % npx; }
                } else {
                    if (pl_t < nlx) fr = pl_fi < p.hv_lo ? pl_fi : pl_fi + nhf;
                    else pl_phase = 3;
                }
            }
            if (fr < 0) break;
            const int pi = pb + pl_rm;
            const int h = p.nqb == 1 ? pi : (int)__umulhi((uint32_t)pi, p.nqb_magic);
            const int qb = pi - h * p.nqb;
            const int item = fr | (h << 8) | (qb << 16);
            ++pl_cnt;                                           // advance inside the phase: no division per item
            if (pl_phase == 1) { pl_t += 1; pl_rm += 1; }
            else               { pl_t += WX; pl_rm += WXr; pl_fi += WXq; }
            if (pl_rm >= npx) { pl_rm -= npx; pl_fi += 1; }
            // key segments of the frame (the decisions of aid_attn_kernel / aid_attn_pp_kernel, on the device coefficients)
            const int kvf = rdlane(tb_kv, fr);
            int nseg = 1, seg0 = kvf, seg1 = 0, two = 0;
            if (MODE == AID_MODE_OUTER) {
                const float cf = rdlane(tb_coef, fr);
                const bool single = cf < 0.f || (a.fused && ((cf == 0.f && kvf == row_b) || (cf == 1.f && kvf == row_e)));
                if (!single) {
                    const bool both = cf != 0.f && cf != 1.f;
                    const int side = row_b + (cf == 1.f ? 1 : 0) * (row_e - row_b);
                    const int fz = a.fused ? 1 : 0;
                    nseg = fz + (both ? 2 : 1);
                    const int first = both ? row_b : side;
                    seg0 = fz * kvf + (1 - fz) * first;
                    seg1 = fz * first + (1 - fz) * row_e;
                    two = both ? 1 : 0;
                }
            }
            const int segw = nseg | (seg0 << 4) | (seg1 << 12) | (two << 20);
            const bool mine = lane == jp;                       // (a compare + two selects; v_writelane has no builtin in this toolchain)
            qi = mine ? item : qi;
            qs = mine ? segw : qs;
        }
    }
    auto item_at = [&](int j) __attribute__((always_inline)) { return rdlane(qi, j & 63); };

    // ---- the computed item -------------------------------------------------------------------------------------------------
    int fr, h, q0, nseg, park_at, swap_at;
    float w_b, w_e;
    auto adopt = [&](int j) __attribute__((always_inline)) {
        const int it = rdlane(qi, j & 63), sw = rdlane(qs, j & 63);
        fr = it & 0xff; h = (it >> 8) & 0xff;
        q0 = ((it >> 16) * 8 + wave) * 32;
        nseg = sw & 0xf;
        park_at = -1; swap_at = -1; w_b = 0.f; w_e = 1.f;
        if (MODE == AID_MODE_OUTER && ((sw >> 20) & 1)) {
            const float cf = rdlane(tb_coef, fr);
            const int fz = a.fused ? 1 : 0;
            w_b = 1.f - cf; w_e = cf; park_at = fz ? 1 : -1; swap_at = fz + 1;
        }
    };
    int jc = 0;
    if (item_at(0) == XNONE) return;                            // the deal left this workgroup empty
    adopt(0);

    // ---- Q fragments (B operand of the swapped product) -----------------------------------------
    T8 qf[4];
    auto scale_q = [&]() __attribute__((always_inline)) {
        if (!a.q_prescaled) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                f32x8 t = up8<T>(qf[ks]);
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] *= p.c2;
                qf[ks] = cvt8<T>(t);
            }
        }
    };
    {
        const int qr = min(q0 + l31, a.s - 1);
        const T* qrow = reinterpret_cast<const T*>(a.q) + (int64_t)fr * a.q_fs + (int64_t)qr * a.ldq + h * D;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const T8*>(qrow + ks * 16 + hi * 8);
        scale_q();
    }

    // ---- DMA: this wave's piece (8 rows x 128 B) of every K tile and of every V^T tile; the next item's Q rows behind the ring --------
    const Rsrc rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.k), 0, 0x7fffffff, 0x00020000);
    const Rsrc rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.vt), 0, 0x7fffffff, 0x00020000);
    const Rsrc rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.q), 0, 0x7fffffff, 0x00020000);
    const Rsrc ro = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, 0x7fffffff, 0x00020000);
    char* const qlds = smem + 2 * XNS * XTILE + wave * 4096;
    // VMEM bookkeeping.  CDNA4's vmcnt counts LDS-DMA loads and stores in ONE in-order queue.  A V slot has to retire the pieces of
    // tile t + 3, i.e. everything OLDER than [tile t + 4, what followed it, tile t + 5, what followed it, tile t + 6]; "what followed" a
    // tile's pieces is, at the end of an item, the next-but-one item's four Q pieces and the finished item's eight output stores
    // (ex_cur collects them; ex_p1 / ex_p2 are the two previous slots').  Counting them in keeps a slot from waiting for requests that
    // are one slot old, or for a store's round trip to HBM — with two-tile items that happened in EVERY slot (plain S = 1024: 36.8 ->
    // 2x.x us, profiles/r04_attn_notes.txt).  Every count is exact or low (low = waits longer): buffer stores are single instructions.
    int ex_cur = 0, ex_p1 = 0, ex_p2 = 0;
    auto dma_q = [&](int it) __attribute__((always_inline)) {   // Q rows of packed item `it` -> this wave's 4 KB (fragment order)
        const int nfr = it & 0xff, nh = (it >> 8) & 0xff, nq0 = ((it >> 16) * 8 + wave) * 32;
        const int qr = min(nq0 + l31, a.s - 1);
        const int vo = qr * (a.ldq * 2) + hi * 16;
        const int so = (int)(((int64_t)nfr * a.q_fs + nh * D) * 2);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (__attribute__((address_space(3))) void*)(qlds + ks * 1024), 16, vo, so + ks * 32, 0, 0);
    };
    const int prow = 8 * wave + (lane >> 3);
    const int pch = (lane & 7) ^ ((prow >> 1) & 7);
    const int kvo = prow * (a.ldk * 2) + pch * 16;
    const int vvo = prow * (a.ldvt * 2) + pch * 16;
    const int kstep = XKT * a.ldk * 2;
    const int kfs2 = (int)a.k_fs * 2, vfs2 = (int)a.vt_fs * 2;
    // DMA walker: item jd, its segment dseg of dn, dleft tiles left in it, running offsets dk / dv; gd = tiles requested so far
    int jd = 0, dn = 1, dseg = 0, dleft = nt, dk = 0, dv = 0, gd = 0, dsw = 0, dh = 0;
    auto dma_row = [&]() __attribute__((always_inline)) -> int {    // key / value row of the walker's current segment
        // segments: [seg0, seg1, row_e] (three), [seg0, seg1] (two), [seg0] (one)
        const int s0 = (dsw >> 4) & 0xff, s1 = (dsw >> 12) & 0xff;
        return dseg == 0 ? s0 : (dseg == 1 ? s1 : row_e);
    };
    auto dma_item = [&](int j) __attribute__((always_inline)) { // point the walker at item j (if there is none: stay, re-stream)
        const int it = item_at(j);
        if (it != XNONE) {
            dsw = rdlane(qs, j & 63);
            dn = dsw & 0xf;
            dh = (it >> 8) & 0xff;
        }
        dseg = 0;
        dleft = nt;
        const int row = dma_row();
        dk = row * kfs2 + dh * (D * 2);
        dv = row * vfs2 + dh * D * a.ldvt * 2;
    };
    auto dma_next = [&]() __attribute__((always_inline)) {
        const int st = gd & (XNS - 1);
        char* dst = smem + st * XTILE + wave * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (__attribute__((address_space(3))) void*)dst, 16, kvo, dk, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (__attribute__((address_space(3))) void*)(dst + XNS * XTILE), 16, vvo, dv, 0, 0);
        ++gd;
        dk += kstep;
        dv += XKT * 2;
        if (--dleft == 0) {
            ++dseg;
            if (dseg < dn) {
                dleft = nt;
                const int row = dma_row();
                dk = row * kfs2 + dh * (D * 2);
                dv = row * vfs2 + dh * D * a.ldvt * 2;
            } else {
                ++jd;
                dma_item(jd);
            }
        }
    };
    dma_item(0);

    // ---- fragment read offsets (as aid_attn_pp.hip) ------------------------------------------------------------------------
    const int krow = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
    const int kx = hi ^ ((krow >> 1) & 7), vx = hi ^ ((l31 >> 1) & 7);
    const int koff = krow * 128, voff = XNS * XTILE + l31 * 128;
    int kad[4], vad[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        kad[ks] = koff + (((2 * ks) ^ kx) << 4);
        vad[ks] = voff + (((2 * ks) ^ vx) << 4);
    }
    const int remh = p.rem - 8 * hi;                            // first masked key index of this lane's 8-key halves, relative to 16 g

    // ---- online-softmax state -----------------------------------------------------------------------
    float m = 0.f;
    bool fresh = true;
    f32x16 o[2], sc[2];
    float lsum = 0.f;
    T8 pf[4];
    f32x16 po[2];
    float pl = 0.f, pm = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
    if (MODE == AID_MODE_OUTER) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { po[0][r] = 0.f; po[1][r] = 0.f; }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) pf[i] = zero8<T>();

    T8 kf[4][2], vf[4][2];
    auto pin = []() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };

    // M slot behind global tile g: S(g + 1) = K(g + 1) Q'^T - m, then O^T += V^T(g) P(g)^T; V^T(g) fragments beside the score MFMAs,
    // K(g + 2) fragments beside the PV MFMAs.  Ring stages are run-time values here: one address add per k-step.
    // `first`: the tile is the first of its item — its PV product STARTS the item's O (no 32 v_mov to clear the accumulators at every
    // item boundary).
    auto mslot = [&](int g, bool first) __attribute__((always_inline)) {
        const int sv = (g & (XNS - 1)) * XTILE, sk = ((g + 2) & (XNS - 1)) * XTILE;
        f32x16 c0, zacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) { c0[r] = -m; zacc[r] = 0.f; }
        pin();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ks = i >> 1, b = i & 1;
            vf[ks][b] = *reinterpret_cast<const T8*>(smem + vad[ks] + sv + b * 4096);
            pin();
            sc[b] = mfma32(kf[ks][b], qf[ks], ks ? sc[b] : c0);
            pin();
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int kk = i >> 1, w = i & 1;
            if (kk == 0) {
                if (first) o[w] = mfma32(vf[kk][w], pf[kk], zacc);
                else       o[w] = mfma32(vf[kk][w], pf[kk], o[w]);
            } else {
                o[w] = mfma32(vf[kk][w], pf[kk], o[w]);
            }
            pin();
            kf[kk][w] = *reinterpret_cast<const T8*>(smem + kad[kk] + sk + w * 4096);
            pin();
        }
    };

    // V slot of a tile (mask = it is the ragged last tile of a key segment)
    auto vslot = [&](bool mask) __attribute__((always_inline)) {
        ex_p2 = ex_p1; ex_p1 = ex_cur; ex_cur = 0;
        dma_next();
        if (mask) {
            // 16-key groups g = 2 b + u hold keys 16 g + 8 hi + (0 .. 7) of the tile in sc[b][8 u ..]
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (p.rem <= 16 * g) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) sc[g >> 1][8 * (g & 1) + e] = NEG;
                } else if (p.rem < 16 * g + 16) {
                    int thr = remh - 16 * g;
                    asm volatile("" : "+v"(thr));               // (opaque: hipcc otherwise hoists the 32 compares of the four groups out of
                                                                //  the item loop and keeps their 64 mask SGPRs alive — 120 SGPR spills)
#pragma unroll
                    for (int e = 0; e < 8; ++e) sc[g >> 1][8 * (g & 1) + e] = e >= thr ? NEG : sc[g >> 1][8 * (g & 1) + e];
                }
            }
        }
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
                    t0 = xdot2<T>(w4[0], t0); t1 = xdot2<T>(w4[1], t1);
                    t0 = xdot2<T>(w4[2], t0); t1 = xdot2<T>(w4[3], t1);
                }
            return t0 + t1;
        };
        auto row_max = [&]() __attribute__((always_inline)) -> float {
            float xm = fmaxf(sc[0][0], sc[0][1]);
#pragma unroll
            for (int i = 1; i < 16; ++i) xm = fmaxf(fmaxf(xm, sc[i >> 3][(2 * i) & 15]), sc[i >> 3][(2 * i + 1) & 15]);
            return xm;
        };
        auto raise = [&](float xm) __attribute__((always_inline)) {
            const float rowmax = max_halves(xm);
            const float shift = fresh ? rowmax : fmaxf(rowmax, 0.f);      // (a first tile may move the reference DOWN: all scores far below 0)
            const float alpha = __builtin_amdgcn_exp2f(-shift);
            m += shift;
            if (!fresh) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
                lsum *= alpha;
            }
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[b][r] -= shift;
            fresh = false;
        };
        // The first tile of a row is exponentiated against the reference 0 it was scored with (text scores are O(10): 2^s is in
        // range; no maximum chain, no shift).  The row sum decides as for every tile — above the head-room, or, for a first tile,
        // (almost) nothing left of it — whether the tile is redone against the row maximum.
        float ts = exp_tile();
        if (__any(ts > __builtin_amdgcn_exp2f(XTH) || (fresh && ts < 1e-30f))) {
            raise(row_max());
            ts = exp_tile();
        }
        fresh = false;
        lsum += ts;
        xwait_vm_dyn(6 + ex_p1 + ex_p2);
    };

    auto park = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { po[0][r] = o[0][r]; po[1][r] = o[1][r]; }
        pl = lsum;
        pm = m;
    };
    auto swap_sides = [&]() __attribute__((always_inline)) {
        const float lrow = lsum + other_half(lsum);
        const float wb = w_b / lrow;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float rb = o[d][r] * wb;
                o[d][r] = park_at < 0 ? 0.f : po[d][r];         // (pure OUTER: the end side starts from the EMPTY state)
                po[d][r] = rb;
            }
        lsum = park_at < 0 ? 0.f : pl;
        const float back = m - (park_at < 0 ? 0.f : pm);
        m = park_at < 0 ? 0.f : pm;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[b][r] += back;
        if (park_at < 0) fresh = true;
    };
    // O / l of the finished item -> out.  The wave's 32 x 64 block goes through ITS 4 KB of LDS behind the ring (the next item's Q rows
    // were read out of it before the last M slot; the next-but-one item's are requested after this) and leaves as four 16-byte stores per
    // lane covering whole 128-byte rows — eight 8-byte stores per lane at a row stride were 1.4 us per item (store issue, not bandwidth:
    // the guide's T21).  Returns the number of VMEM instructions issued (bookkeeping; 0 for a wave past the last row or in accumulate
    // mode, where the compiler's own wait on the loads drains the queue: a low count only makes later waits longer).
    auto finish = [&]() __attribute__((always_inline)) -> int {
        const float inv = w_e / (lsum + other_half(lsum));
        if (q0 >= a.s) return 0;                                // (wave-uniform)
        const float osc = a.out_scale * rdlane(tb_fs, fr);
        const bool two = MODE == AID_MODE_OUTER && swap_at >= 0;
        // stage: lane (row l31, hi) holds channels 32 d + 8 g + 4 hi + (0 .. 3); 16-byte chunk c = 4 d + g of row r sits at slot c ^ swz(r)
        char* const st = qlds;
        const int wsw = (l31 >> 1) & 7;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float r_ = o[d][4 * g + e] * inv;
                    if (two) r_ += po[d][4 * g + e];
                    v[e] = r_ * osc;
                }
                *reinterpret_cast<T4*>(st + l31 * 128 + (((4 * d + g) ^ wsw) << 4) + 8 * hi) = cvt4<T>(v);
            }
        // read back row-major: lane -> (row lane / 8 + 8 i, chunk lane % 8)
        const int so = (int)(((int64_t)fr * a.o_fs + h * D) * 2);
        const int rr = lane >> 3, cc = lane & 7;
        int n = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = rr + 8 * i;
            T8 v = *reinterpret_cast<const T8*>(st + row * 128 + ((cc ^ ((row >> 1) & 7)) << 4));
            const int q = q0 + row;
            const int vo = (min(q, a.s - 1) * a.ldo + cc * 8) * 2;
            if (a.accumulate) v = cvt8<T>(up8<T>(v) + up8<T>(__builtin_bit_cast(T8, __builtin_amdgcn_raw_buffer_load_b128(ro, vo, so, 0))));
            if (q < a.s) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ro, vo, so, 0);
            n += q0 + 8 * i < a.s ? 1 : 0;                      // (the instruction issues iff the 8-row group has a live row: uniform)
        }
        return a.accumulate ? 0 : n;
    };
    auto settle = [&]() __attribute__((always_inline)) {
        asm volatile("" : "+v"(sc[0]), "+v"(sc[1]));
        asm volatile("" : "+v"(o[0]), "+v"(o[1]));
        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
        asm volatile("" : "+v"(sc[0]), "+v"(sc[1]));
        asm volatile("" : "+v"(o[0]), "+v"(o[1]));
    };

    // ---- prologue: tiles 0 .. 5 of the stream requested, 0 .. 2 retired and published; S(0); fragments of K(1) ----------------
    constexpr int LEAD = 6;
#pragma nounroll
    for (int t = 0; t < LEAD; ++t) dma_next();
    xwait_vm<6>();
    xbarrier();
    if (grp == 1) xbarrier();                                   // the second group runs one barrier behind
#pragma unroll
    for (int i = 0; i < 8; ++i) kf[i >> 1][i & 1] = *reinterpret_cast<const T8*>(smem + kad[i >> 1] + (i & 1) * 4096);
    {
        f32x16 zacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) zacc[r] = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) sc[i & 1] = mfma32(kf[i >> 1][i & 1], qf[i >> 1], (i >> 1) ? sc[i & 1] : zacc);
        pin();
#pragma unroll
        for (int i = 0; i < 8; ++i) kf[i >> 1][i & 1] = *reinterpret_cast<const T8*>(smem + kad[i >> 1] + XTILE + (i & 1) * 4096);
    }
    settle();
    xbarrier();

    // ---- items ------------------------------------------------------------------------------------------------------------
    // Q rows: item 0's came straight from memory; item j + 1's are requested at the END of item j - 1 (behind its output, which is staged
    // through the same 4 KB) and read back in the last V slot of item j: a whole item of lead whatever its length.
    int gc = 0;                                                 // global index of the tile being computed
    if (item_at(1) != XNONE) { dma_q(item_at(1)); ex_cur += 4; }
#pragma nounroll
    for (;; ++jc) {
        const bool has_next = item_at(jc + 1) != XNONE;
        const int NT = nseg * nt;
#pragma nounroll
        for (int sgi = 0; sgi < (MODE == AID_MODE_PLAIN ? 1 : nseg); ++sgi) {
            if (MODE == AID_MODE_OUTER) {
                if (sgi == park_at) park();
                if (sgi == swap_at) swap_sides();
            }
#pragma nounroll
            for (int tt = 0; tt < nt; ++tt) {
                vslot(ragged && tt == nt - 1);
                if (has_next && sgi == nseg - 1 && tt == nt - 1) {
                    // the M slot behind the item's last tile forms the next item's S(0): its Q rows and a zero row reference.  The Q
                    // pieces are older than three slots for items of three tiles or more (the slot's own wait retired them);
                    // behind them came the previous item's stores and this item's tile pieces
                    if (NT == 1)      xwait_vm<2>();
                    else if (NT == 2) xwait_vm<4>();
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const T8*>(qlds + ks * 1024 + lane * 16);
                    scale_q();
                    m = 0.f;
                }
                xbarrier();
                mslot(gc, sgi == 0 && tt == 0);
                settle();
                xbarrier();
                ++gc;
            }
        }
        ex_cur += finish();
        if (!has_next) break;
        {
            const int nn = item_at(jc + 2);
            if (nn != XNONE) { dma_q(nn); ex_cur += 4; }
        }
        fresh = true;
        lsum = 0.f;
        pl = 0.f;
        pm = 0.f;
        adopt(jc + 1);
    }
    xwait_vm<0>();
    if (grp == 0) xbarrier();
}

}  // namespace

// May the short-stream kernel run this call (alone)?  d = 64, padded keys / values, PLAIN or OUTER, at most four tiles per segment.
bool attn_xs_supported(const AidAttnArgs& a) {
    const int nt = (a.l + XKT - 1) / XKT;
    const int64_t lim = 1ll << 31;
    const int64_t kb = (int64_t)a.n_kv * a.k_fs * 2, vb = (int64_t)a.n_kv * a.vt_fs * 2, qb = (int64_t)a.n_frames * a.q_fs * 2;
    const int64_t ob = (int64_t)a.n_frames * a.o_fs * 2;
    const uintptr_t al = reinterpret_cast<uintptr_t>(a.k) | reinterpret_cast<uintptr_t>(a.vt);
    const int nqb = (a.s + 255) / 256;
    // every workgroup plans all its items up front into 63 lanes: heavy share + levelling + light share (aid_attn_xs_kernel)
    int ncu = 256;
    {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) ncu = 256;
    }
    const int wxn = ncu / 8 > 0 ? ncu / 8 : 1;
    const int npx_max = (a.heads * nqb + 7) / 8;
    const int per_wg = (npx_max * a.n_frames + wxn - 1) / wxn + 4;
    if (per_wg > 60) return false;
    return a.kv_padded == 1 && a.d == 64 && (a.mode == AID_MODE_PLAIN || a.mode == AID_MODE_OUTER) && nt >= 1 && nt <= 4 &&
           a.k_fs >= (int64_t)nt * XKT * a.ldk && a.ldvt >= nt * XKT && a.vt_fs >= (int64_t)a.heads * 64 * a.ldvt &&
           a.n_frames <= 64 && a.heads <= 255 && a.n_kv <= 255 && nqb * a.heads < 65536 && a.ldk % 8 == 0 && a.ldvt % 8 == 0 &&
           a.k_fs % 8 == 0 && a.vt_fs % 8 == 0 && (al & 15) == 0 && kb < lim && vb < lim && qb < lim && ob < lim && a.ldo % 4 == 0 && a.o_fs % 4 == 0;
}

hipError_t attn_xs_launch(const AidAttnArgs& a, hipStream_t stream) {
    AttnXSParams p;
    p.a = a;
    p.nqb = (a.s + 255) / 256;
    p.nqb_magic = (uint32_t)((1ull << 32) / (uint32_t)p.nqb) + 1u;
    p.nt = (a.l + XKT - 1) / XKT;
    p.rem = a.l - (p.nt - 1) * XKT;
    p.c2 = a.softmax_scale * 1.4426950408889634f;
    const int n_aid = a.n_frames - a.n_plain;
    p.hv_lo = p.hv_hi = 0;
    p.hv_units = 1;
    if (a.mode == AID_MODE_OUTER && n_aid > 2) { p.hv_lo = 1; p.hv_hi = n_aid - 1; p.hv_units = a.fused ? 3 : 2; }
    const size_t smem = (size_t)2 * XNS * XTILE + 8 * 4096;
    static PerDevice<int> attr_set[2];
    const int ti = a.dtype == AID_DTYPE_F16 ? 0 : 1;
    int* done = attr_set[ti].slot();
    if (!done) return hipErrorInvalidDevice;
    const int mi = a.mode == AID_MODE_PLAIN ? 0 : 1;
    const void* fns[2][2] = {
        {reinterpret_cast<const void*>(&aid_attn_xs_kernel<f16, AID_MODE_PLAIN>), reinterpret_cast<const void*>(&aid_attn_xs_kernel<f16, AID_MODE_OUTER>)},
        {reinterpret_cast<const void*>(&aid_attn_xs_kernel<bf16, AID_MODE_PLAIN>), reinterpret_cast<const void*>(&aid_attn_xs_kernel<bf16, AID_MODE_OUTER>)}};
    const void* fn = fns[ti][mi];
    if (!(*done & (1 << mi))) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        *done |= 1 << mi;
    }
    static PerDevice<int> cus;
    int* ncu = cus.slot();
    if (!ncu) return hipErrorInvalidDevice;
    if (*ncu == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) *ncu = 256;
    }
    const int grid = (*ncu / 8) * 8 > 0 ? (*ncu / 8) * 8 : 8;   // whole workgroups per XCD
    void* kargs[] = {const_cast<AttnXSParams*>(&p)};
    return hipLaunchKernel(fn, dim3(grid), dim3(512), kargs, smem, stream);
}

}  // namespace aid
