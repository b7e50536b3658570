// Attention over TEXT keys, head dim 64: the cross-attention calls of the SDXL stack (77 keys per context; reference
// interpolation.py:623-664 with encoder_hidden_states given, :581-584 de-activated) — PLAIN calls, the PLAIN riders of a batched-CFG
// call, and the one / two / three key segments of a pure or fused OUTER frame.  The default for these calls since round 5.
//
// These launches move 37 - 147 MB of q / out per call against 10 GFLOP of arithmetic: they are HBM streams (tools/ubench/
// head_stride_copy.hip: a per-head copy of the same rows in the same order runs at 4.8 - 6.0 TB/s = 12 - 15 us for the S = 1024 call).
// aid_attn_kernel runs them as 2240 - 4480 short workgroups, each one latency chain behind machinery made for LONG key streams (tile
// rings, a barrier per tile); with at most 96 keys per segment none of that is needed:
//   * every key segment of the workgroup's (frame, head) sits in LDS for the workgroup's whole life (24 KB per segment: K as
//     [8 chunks][96 keys] x 16 B, V^T as [12 chunks][64 channels] x 16 B — fragment reads are consecutive 16-B words over the lanes,
//     conflict-free without padding; three segments = 72 KB, two workgroups per CU); ONE barrier per workgroup;
//   * a wave owns 32 query rows at a time and is independent of the other waves from then on: Q fragments straight from global
//     memory (the next tile's are requested before the current tile is computed and waited for BEFORE the current tile's stores are
//     issued: loads and stores share vmcnt), one 32-key score tile at a time with an online softmax over the segment's <= 3 tiles
//     (16 score registers: 111 VGPRs = four waves per SIMD for PLAIN launches, 210 and no scratch for INNER / OUTER ones), 16-key halves past
//     L skipped;
//   * the segments of a fused OUTER frame are combined at the end from their maxima and row sums:
//         out = (1 - c) [a1 O_own + b1 O_beg] / (a1 l_own + b1 l_beg) + c [a2 O_own + b2 O_end] / (a2 l_own + b2 l_end),
//         a1 = 2^(m_own - max(m_own, m_beg)), b1 = 2^(m_beg - max(m_own, m_beg)), ... — the same softmax over [own ; begin] and
//         [own ; end] as the reference's two calls, with the own keys' scores and products computed once;
//   * the 32 x 64 output block leaves as 16-byte stores (two v_permlane32_swap per 16 channels bring the halves of a row together).
// Measured -5 ... -17 % against aid_attn_kernel launch by launch; what still separates it from the copy's time is in
// profiles/r05_attn_tx_notes.txt (ablation builds: -DAID_TX_ABL=1 no fill, 2 no exponentials, 3 no arithmetic at all).
#include <string.h>

#include "aid_common.hpp"
#include "aid_kernels.hpp"

namespace aid {

constexpr int TXK = 96;                              // keys per segment, padded (3 score tiles of 32)
constexpr int TX_KBYTES = 8 * TXK * 16;              // K part of a region
constexpr int TX_VBYTES = 12 * 64 * 16;              // V^T part
constexpr int TX_REGION = TX_KBYTES + TX_VBYTES;     // 24576

struct AttnTxParams {
    AidAttnArgs a;
    int32_t chunks;                     // workgroups per (frame, head)
    int32_t tiles_per_chunk;            // 32-row tiles a workgroup works through (4 waves, round robin)
    int32_t na;                         // heavy workgroups per head (balance hint: frames [0, n_frames - n_plain) x chunks)
    float   c2;                         // softmax_scale * log2(e), or 1 when q is pre-scaled
};

__device__ __forceinline__ f32x16 tx_zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}

template <typename T>
__device__ __forceinline__ uint32_t tx_pack2(float x, float y);
template <>
__device__ __forceinline__ uint32_t tx_pack2<f16>(float x, float y) {
    Vec<f16>::v2 v = __builtin_convertvector((f32x2){x, y}, Vec<f16>::v2);
    return __builtin_bit_cast(uint32_t, v);
}
template <>
__device__ __forceinline__ uint32_t tx_pack2<bf16>(float x, float y) {
    Vec<bf16>::v2 v = __builtin_convertvector((f32x2){x, y}, Vec<bf16>::v2);
    return __builtin_bit_cast(uint32_t, v);
}

template <typename T>
__device__ __forceinline__ float tx_dot2(uint32_t w, float acc);
template <>
__device__ __forceinline__ float tx_dot2<bf16>(uint32_t w, float acc) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    b2 one;
    one[0] = (__bf16)1.0f; one[1] = (__bf16)1.0f;
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2, w), one, acc, false);
}
template <>
__device__ __forceinline__ float tx_dot2<f16>(uint32_t w, float acc) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    h2 one;
    one[0] = (_Float16)1.0f; one[1] = (_Float16)1.0f;
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, w), one, acc, false);
}

#if defined(AID_TX_ABL) && AID_TX_ABL == 2
#define TX_EXP(x) (x)                               // development: no exponentials (garbage results, timing only)
#else
#define TX_EXP(x) __builtin_amdgcn_exp2f(x)
#endif

#if defined(AID_TX_ABL) && AID_TX_ABL == 4              // development: shader-clock stamps of one wave (tools/dev/tx_timeline.py)
#define TX_STAMP(i) do { if (tx_trace) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tx_ts[tx_n++] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define TX_STAMP(i) do { } while (0)
#endif

// NSEG = LDS regions: 1 for PLAIN launches, 3 for OUTER launches (whose frames run one to three segments each).
template <typename T, int NSEG>
__global__ __launch_bounds__(256, 2) void aid_attn_tx_kernel(const AttnTxParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tx_smem[];
    typedef typename Vec<T>::v8 T8;
    const AidAttnArgs& a = p.a;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, m = lane & 31, h = lane >> 5;
    const int per = a.n_frames * p.chunks;
    const int id = heavy_first((int)blockIdx.x, (int)gridDim.x, p.na, per);
    const int head = id / per, rem = id - head * per, fr = rem / p.chunks, chunk = rem - fr * p.chunks;
    const int Lk = a.l;
    const int kvf = a.kv_map ? a.kv_map[fr] : fr;
    const float cf = (NSEG == 1 || a.coef == nullptr) ? -1.f : a.coef[fr];
    // which segments this frame runs — the same decisions as aid_attn_kernel (see there): single = PLAIN, a rider (negative
    // coefficient) or a fused END-POINT frame; coefficient exactly 0 / 1 skips the zero-weighted side
    const bool single = NSEG == 1 || cf < 0.f || (a.fused && ((cf == 0.f && kvf == a.begin) || (cf == 1.f && kvf == a.end)));
    const bool has_own = single || a.fused != 0;
    // INNER: ONE interpolated segment [(1 - c) K_begin + c K_end] beside the own keys, one softmax over both — the "begin side" with
    // weight 1; its keys / values are the end-point frame itself for c = 0 / 1, else the rows aid_lerp_kv wrote to k2 / vt2
    const bool inner = NSEG > 1 && a.mode == AID_MODE_INNER;
    const bool has_b = !single && (inner || cf != 1.f), has_e = !single && !inner && cf != 0.f;
    const int nseg = (has_own ? 1 : 0) + (has_b ? 1 : 0) + (has_e ? 1 : 0);
    // role of region r: 0 own, 1 begin side, 2 end side; source of its keys / values (nullptr: region unused)
    const int role0 = has_own ? 0 : has_b ? 1 : 2;
    const int role1 = has_own ? (has_b ? 1 : 2) : 2;
    const T* const Kh = reinterpret_cast<const T*>(a.k) + head * 64;
    const T* const Vh = reinterpret_cast<const T*>(a.vt) + (int64_t)(head * 64) * a.ldvt;
    const bool lerped = inner && cf != 0.f && cf != 1.f;
    const T* const kside = lerped ? reinterpret_cast<const T*>(a.k2) + head * 64 + (int64_t)fr * a.k_fs
                                  : Kh + (int64_t)((inner ? cf == 1.f : !has_b) ? a.end : a.begin) * a.k_fs;
    const T* const vside = lerped ? reinterpret_cast<const T*>(a.vt2) + (int64_t)(head * 64) * a.ldvt + (int64_t)fr * a.vt_fs
                                  : Vh + (int64_t)((inner ? cf == 1.f : !has_b) ? a.end : a.begin) * a.vt_fs;
    // region 0: own keys, or the first side of a pure call; region 1: the first (or only) side behind own keys, or the end side of a
    // pure OUTER call; region 2: the end side of a fused OUTER frame
    const T* const rk[3] = {has_own ? Kh + (int64_t)kvf * a.k_fs : kside,
                            nseg < 2 ? nullptr : has_own ? kside : Kh + (int64_t)a.end * a.k_fs,
                            nseg < 3 ? nullptr : Kh + (int64_t)a.end * a.k_fs};
    const T* const rv[3] = {has_own ? Vh + (int64_t)kvf * a.vt_fs : vside,
                            nseg < 2 ? nullptr : has_own ? vside : Vh + (int64_t)a.end * a.vt_fs,
                            nseg < 3 ? nullptr : Vh + (int64_t)a.end * a.vt_fs};

    const T* Qg = reinterpret_cast<const T*>(a.q) + (int64_t)fr * a.q_fs + head * 64 + 8 * h;
    T* Og = reinterpret_cast<T*>(a.out) + (int64_t)fr * a.o_fs + head * 64 + 8 * h;
    const int ntiles = (a.s + 31) >> 5;
    const int t0 = chunk * p.tiles_per_chunk;
    const int t_end = t0 + p.tiles_per_chunk < ntiles ? t0 + p.tiles_per_chunk : ntiles;

    auto load_q = [&](T8 (&q)[4], int t) {
        const int row = 32 * t + m < a.s ? 32 * t + m : a.s - 1;       // rows past S: any valid row, their results are not stored
        const T* src = Qg + (int64_t)row * a.ldq;
#pragma unroll
        for (int j = 0; j < 4; ++j) q[j] = *reinterpret_cast<const T8*>(src + 16 * j);
    };
    T8 qf[4];
    int t = t0 + wave;
#if defined(AID_TX_ABL) && AID_TX_ABL == 4
    const bool tx_trace = blockIdx.x == gridDim.x / 2 && wave == 0;
    uint64_t tx_ts[40];
    int tx_n = 0;
    TX_STAMP(0);                                        // kernel entry
#endif
    if (t < t_end) load_q(qf, t);                       // in flight across the fill

    // ---- fill: every segment of this (frame, head), once per workgroup ------------------------------------------------------------
#if !defined(AID_TX_ABL) || AID_TX_ABL != 1
    {
        T8 stk[NSEG][3], stv[NSEG][3];
#pragma unroll
        for (int r = 0; r < NSEG; ++r) {
            if (rk[r] == nullptr) continue;
            const T* kb = rk[r];
            const T* vb = rv[r];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int idx = tid + 256 * i;
                const int key = idx >> 3, c = idx & 7;              // K: 8 consecutive lanes = one 128-B key row of the head
                stk[r][i] = key < Lk ? *reinterpret_cast<const T8*>(kb + (int64_t)key * a.ldk + 8 * c) : zero8<T>();
                const int ch = idx / 12, x = idx - 12 * ch;         // V^T: 12 consecutive lanes = the 96 keys of one channel
                T8 v = 8 * x < Lk ? *reinterpret_cast<const T8*>(vb + (int64_t)ch * a.ldvt + 8 * x) : zero8<T>();
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (8 * x + e >= Lk) v[e] = (T)0.0f;             // keys >= L get P = 0; their V must be finite
                stv[r][i] = v;
            }
        }
#pragma unroll
        for (int r = 0; r < NSEG; ++r) {
            if (rk[r] == nullptr) continue;
            unsigned char* reg = tx_smem + r * TX_REGION;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int idx = tid + 256 * i;
                const int key = idx >> 3, c = idx & 7;
                *reinterpret_cast<T8*>(reg + (c * TXK + key) * 16) = stk[r][i];
                // V^T: inside every group of 16 keys the two middle 4-key pieces change places, so that the 16-B word a lane reads for
                // a 16-key step holds exactly the keys of its eight score registers: [16u + 4h .. +3 , 16u + 8 + 4h .. +3]
                const int ch = idx / 12, x = idx - 12 * ch;
                const u32x4 w = __builtin_bit_cast(u32x4, stv[r][i]);
                unsigned char* vb = reg + TX_KBYTES + ch * 16 + ((x & 1) ? 8 : 0);
                const int c0 = x & ~1;
                *reinterpret_cast<u32x2*>(vb + (c0 * 64) * 16) = (u32x2){w[0], w[1]};
                *reinterpret_cast<u32x2*>(vb + ((c0 + 1) * 64) * 16) = (u32x2){w[2], w[3]};
            }
        }
    }
#endif
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(qf[j]));      // the first tile's Q has landed (the fill waited on the same counter)

    const int nkt = (Lk + 31) >> 5;                                  // score tiles that hold a valid key
    const float c2 = p.c2;
    const float osc = a.out_scale * (a.frame_scale ? a.frame_scale[fr] : 1.f);
    const unsigned char* kfrag0 = tx_smem + (h * TXK + m) * 16;                    // + region, + j * 3072 + kt * 512
    const unsigned char* vfrag0 = tx_smem + TX_KBYTES + (h * 64 + m) * 16;         // + region, + (4 kt + 2 s) * 1024 + ct * 512
    const int lim = Lk - 4 * h;                                      // register i of tile kt is a valid key iff 32 kt + 8 (i / 4) + i % 4 < lim

    TX_STAMP(1);                                        // fill done, first Q landed
    // one 32-row tile: scores, softmax, second product of every segment, the combination, the packed output words
    auto compute = [&](const T8 (&qf)[4], u32x4 (&ow)[4]) {
        int L = Lk;
        asm volatile("" : "+s"(L));        // the per-half-tile decisions are re-derived per tile (scalar compares) instead of living in spilled SGPR pairs
        f32x16 o_own[2], res[2] = {tx_zero16(), tx_zero16()};
        float m_own = 0.f, l_own = 1.f;
#if defined(AID_TX_ABL) && AID_TX_ABL == 3            // development: no arithmetic at all — the kernel as a copy of its q rows
        for (int j = 0; j < 4; ++j) { const u32x4 w = __builtin_bit_cast(u32x4, qf[j]); for (int e = 0; e < 4; ++e) res[j >> 1][4 * (j & 1) + e] = __uint_as_float(w[e]); }
        for (int sg = 0; sg < 0; ++sg) {
#else
#pragma unroll 1
        for (int sg = 0; sg < nseg; ++sg) {
#endif
            const unsigned char* kf = kfrag0 + sg * TX_REGION;
            const unsigned char* vf = vfrag0 + sg * TX_REGION;
            // ---- one 32-key score tile at a time: S^T[key, row] = K Q^T, running maximum, probabilities, O^T += V^T P^T ----
            // (online softmax over the <= 3 tiles of the segment: 16 score registers live instead of 48 — that is what lets the PLAIN
            //  instantiation run four waves per SIMD and the OUTER one keep its three output blocks without scratch; the output block
            //  is rescaled only when a later tile raises a row's maximum)
            f32x16 oc[2] = {tx_zero16(), tx_zero16()};
            float mx = -INFINITY, ls = 0.f;
#pragma unroll
            for (int kt = 0; kt < 3; ++kt) {
                if (kt < nkt) {
                    f32x16 sc = tx_zero16();
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        sc = mfma32(*reinterpret_cast<const T8*>(kf + j * 3072 + kt * 512), qf[j], sc);
                    // the 16-key half that straddles L: keys >= L to -inf.  Halves entirely past L are never read.
#pragma unroll
                    for (int s = 0; s < 2; ++s)
                        if (32 * kt + 16 * s < L && 32 * kt + 16 * s + 16 > L) {
#pragma unroll
                            for (int i = 8 * s; i < 8 * s + 8; ++i)
                                if (32 * kt + 8 * (i >> 2) + (i & 3) >= lim) sc[i] = -INFINITY;
                        }
                    float tm = -INFINITY;
#pragma unroll
                    for (int s = 0; s < 2; ++s)
                        if (32 * kt + 16 * s < L) {
#pragma unroll
                            for (int i = 8 * s; i < 8 * s + 8; ++i) tm = fmaxf(tm, sc[i]);
                        }
                    tm = max_halves(tm);
                    if (kt > 0 && __any(tm > mx)) {                       // a row's maximum rises: rescale what was accumulated
                        const float mn = fmaxf(mx, tm);
                        const float alpha = TX_EXP((mx - mn) * c2);
                        oc[0] *= alpha; oc[1] *= alpha; ls *= alpha;
                        mx = mn;
                    } else if (kt == 0) {
                        mx = tm;
                    }
                    const float nm = -mx * c2;
#pragma unroll
                    for (int s = 0; s < 2; ++s)
                        if (32 * kt + 16 * s < L) {
                            uint32_t pk[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const float p0 = TX_EXP(fmaf(sc[8 * s + 2 * u], c2, nm));
                                const float p1 = TX_EXP(fmaf(sc[8 * s + 2 * u + 1], c2, nm));
                                pk[u] = tx_pack2<T>(p0, p1);
                                ls = tx_dot2<T>(pk[u], ls);
                            }
                            const T8 pf = __builtin_bit_cast(T8, (u32x4){pk[0], pk[1], pk[2], pk[3]});
#pragma unroll
                            for (int ct = 0; ct < 2; ++ct)
                                oc[ct] = mfma32(*reinterpret_cast<const T8*>(vf + (4 * kt + 2 * s) * 1024 + ct * 512), pf, oc[ct]);
                        }
                }
            }
            ls = sum_halves(ls);
            TX_STAMP(3);                                // segment done (scores, softmax, second product)
            // ---- combine ----
            const int rl = NSEG == 1 ? 0 : sg == 0 ? role0 : sg == 1 ? role1 : 2;
            if (rl == 0) {
                if (single) {
                    const float w = osc / ls;
                    res[0] = oc[0] * w; res[1] = oc[1] * w;
                } else {
                    o_own[0] = oc[0]; o_own[1] = oc[1];
                    m_own = mx; l_own = ls;
                }
            } else {
                const float side = inner ? 1.f : rl == 1 ? 1.f - cf : cf;
                if (has_own) {
                    const float ms = fmaxf(m_own, mx);
                    const float ao = __builtin_amdgcn_exp2f((m_own - ms) * c2), bo = __builtin_amdgcn_exp2f((mx - ms) * c2);
                    const float w = side * osc / (ao * l_own + bo * ls);
                    const float wa = w * ao, wb = w * bo;
                    res[0] += o_own[0] * wa + oc[0] * wb; res[1] += o_own[1] * wa + oc[1] * wb;
                } else {
                    const float w = side * osc / ls;
                    res[0] += oc[0] * w; res[1] += oc[1] * w;
                }
            }
        }

        // ---- output words: lane (row m, half h) holds channels 32 ct + 8 g + 4 h + {0..3}; after the swaps 32 ct + 16 u + 8 h + {0..7} ----
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const uint32_t x0 = tx_pack2<T>(res[ct][8 * u], res[ct][8 * u + 1]), x1 = tx_pack2<T>(res[ct][8 * u + 2], res[ct][8 * u + 3]);
                const uint32_t y0 = tx_pack2<T>(res[ct][8 * u + 4], res[ct][8 * u + 5]), y1 = tx_pack2<T>(res[ct][8 * u + 6], res[ct][8 * u + 7]);
                const auto s0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
                ow[2 * ct + u] = (u32x4){s0[0], s1[0], s0[1], s1[1]};
            }
    };
    auto store_rows = [&](const u32x4 (&ow)[4], int tt) {
        const int row = 32 * tt + m;
        T* dst = Og + (int64_t)row * a.ldo;
        if (row < a.s) {
#pragma unroll
            for (int w = 0; w < 4; ++w) *reinterpret_cast<u32x4*>(dst + 16 * w) = ow[w];
        }
    };
    for (; t < t_end; t += 4) {
        T8 qn[4];
        const bool more = t + 4 < t_end;
        if (more) load_q(qn, t + 4);
        TX_STAMP(2);                                    // tile start (next Q requested)
        u32x4 ow[4];
        compute(qf, ow);
        // ORDER MATTERS (loads and stores share vmcnt on gfx950 and the compiler waits for both at once): the next tile's Q — requested
        // before this tile's arithmetic — is waited for HERE, before this tile's stores are issued, so that wait covers loads that had
        // the whole tile to land and stores that are a whole tile old; the stores below then have the next tile to complete.
        TX_STAMP(4);                                    // output words ready
        if (more) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                qf[j] = qn[j];
                asm volatile("" : "+v"(qf[j]));
            }
        }
        TX_STAMP(5);                                    // next Q landed (and the previous tile's stores)
        store_rows(ow, t);
        TX_STAMP(6);                                    // stores issued
    }
#if defined(AID_TX_ABL) && AID_TX_ABL == 4
    if (tx_trace) {                                     // (overwrites the head of `out`: development only)
        __builtin_amdgcn_s_waitcnt(0);
        if (lane == 0) {
            uint64_t* dbg = reinterpret_cast<uint64_t*>(a.out);
            dbg[0] = (uint64_t)tx_n;
            for (int i = 0; i < tx_n && i < 40; ++i) dbg[1 + i] = tx_ts[i];
        }
    }
#endif
}

bool attn_tx_supported(const AidAttnArgs& a) {
    if (a.d != 64 || a.l < 1 || a.l > TXK) return false;
    if (a.dtype != AID_DTYPE_F16 && a.dtype != AID_DTYPE_BF16) return false;
    if (a.mode == AID_MODE_INNER && (!a.k2 || !a.vt2)) return false;
    if (a.accumulate) return false;                         // (the IP-Adapter image branch adds into the text result: aid_attn_kernel)
    if (a.ldo % 8 || a.o_fs % 8) return false;              // 16-byte output stores
    return true;
}

template <typename T>
static hipError_t tx_launch(const AttnTxParams& p, hipStream_t stream) {
    const bool outer = p.a.mode != AID_MODE_PLAIN;            // INNER / OUTER: the three-region instantiation
    const size_t smem = (size_t)(outer ? 3 : 1) * TX_REGION;
    const void* fn = outer ? reinterpret_cast<const void*>(&aid_attn_tx_kernel<T, 3>) : reinterpret_cast<const void*>(&aid_attn_tx_kernel<T, 1>);
    static PerDevice<int> attr_set;
    int* done = attr_set.slot();
    if (!done) return hipErrorInvalidDevice;
    if (outer && !*done) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        *done = 1;
    }
    const int grid = p.a.heads * p.a.n_frames * p.chunks;
    void* kargs[] = {const_cast<AttnTxParams*>(&p)};
    return hipLaunchKernel(fn, dim3(grid), dim3(256), kargs, smem, stream);
}

hipError_t attn_tx_launch(const AidAttnArgs& a, hipStream_t stream) {
    AttnTxParams p;
    memset(&p, 0, sizeof(p));
    p.a = a;
    const int ntiles = (a.s + 31) / 32;
    int tpw = tune(TUNE_ATTN_TX_TILES);                      // 32-row tiles per wave (development knob)
    if (tpw <= 0) tpw = ntiles >= 96 ? 5 : 3;                // measured best of 1 ... 16 on the SDXL launches (S = 4096 : 5, S = 1024 : 3; flat within 3 % from 2 to 5)
    int tpc = 4 * tpw;
    if (tpc > ntiles) tpc = ntiles;
    p.tiles_per_chunk = tpc;
    p.chunks = (ntiles + tpc - 1) / tpc;
    const int n_heavy = a.mode != AID_MODE_PLAIN ? a.n_frames - a.n_plain : 0;
    p.na = (n_heavy > 0 && n_heavy < a.n_frames ? n_heavy : 0) * p.chunks;
    p.c2 = a.q_prescaled ? 1.f : a.softmax_scale * 1.4426950408889634f;
    return a.dtype == AID_DTYPE_F16 ? tx_launch<f16>(p, stream) : tx_launch<bf16>(p, stream);
}

}  // namespace aid
