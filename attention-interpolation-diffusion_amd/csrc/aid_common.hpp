// Shared device helpers for the gfx950 kernels (wave64, MFMA 32x32x16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Cache policy of the big output stores (buffer-store aux bits on gfx950: 1 = sc0, 2 = nt, 16 = sc1).  0 = plain write-back stores (the
// product); development builds (tools/dev/Makefile, libaid_st<aux>.so) try the others: a kernel boundary is a release of everything
// the kernel left dirty in the L2s, and write-through stores leave nothing dirty (profiles/r06_notes.md).
#ifndef AID_ST_AUX
#define AID_ST_AUX 0
#endif

namespace aid {

typedef _Float16 f16;
typedef __bf16   bf16;

template <typename T> struct Vec;
template <> struct Vec<f16> {
    typedef f16 v2 __attribute__((ext_vector_type(2)));
    typedef f16 v4 __attribute__((ext_vector_type(4)));
    typedef f16 v8 __attribute__((ext_vector_type(8)));
};
template <> struct Vec<bf16> {
    typedef bf16 v2 __attribute__((ext_vector_type(2)));
    typedef bf16 v4 __attribute__((ext_vector_type(4)));
    typedef bf16 v8 __attribute__((ext_vector_type(8)));
};

typedef float f32x2  __attribute__((ext_vector_type(2)));
typedef float f32x4  __attribute__((ext_vector_type(4)));
typedef float f32x8  __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// D(32x32, fp32) += A(32x16) * B(16x32).  Lane l supplies A[i = l&31][8*(l>>5) .. +7] and
// B[8*(l>>5) .. +7][j = l&31]; lane l receives D[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31], r in [0,16).
__device__ __forceinline__ f32x16 mfma32(const Vec<f16>::v8& a, const Vec<f16>::v8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(const Vec<bf16>::v8& a, const Vec<bf16>::v8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

template <typename T>
__device__ __forceinline__ typename Vec<T>::v8 zero8() {
    typename Vec<T>::v8 z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z[i] = (T)0.0f;
    return z;
}

template <typename T>
__device__ __forceinline__ typename Vec<T>::v8 cvt8(const f32x8& x) {
    return __builtin_convertvector(x, typename Vec<T>::v8);     // v_cvt_pk_{f16,bf16}_f32, RNE
}
template <typename T>
__device__ __forceinline__ f32x8 up8(const typename Vec<T>::v8& x) {
    return __builtin_convertvector(x, f32x8);
}
template <typename T>
__device__ __forceinline__ typename Vec<T>::v4 cvt4(const f32x4& x) {
    return __builtin_convertvector(x, typename Vec<T>::v4);
}
template <typename T>
__device__ __forceinline__ f32x4 up4(const typename Vec<T>::v4& x) {
    return __builtin_convertvector(x, f32x4);
}

// value held by the partner lane (lane ^ 32)
__device__ __forceinline__ float other_half(float x) {
    const uint32_t u = __float_as_uint(x);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    // lanes < 32 find the partner's value in r[1], lanes >= 32 in r[0]
    return (threadIdx.x & 32) ? __uint_as_float(r[0]) : __uint_as_float(r[1]);
}
__device__ __forceinline__ float max_halves(float x) {
    const uint32_t u = __float_as_uint(x);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float sum_halves(float x) {
    const uint32_t u = __float_as_uint(x);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// Bijective XCD-aware remap of a 1-D grid: hardware places block b on XCD b % 8; give every
// XCD a contiguous range of logical ids so neighbouring tiles share that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
    constexpr int NX = 8;
    const int q = nblocks / NX, r = nblocks % NX;
    const int x = bid % NX, j = bid / NX;
    const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + j;
}

// Block -> logical id for a launch that mixes interpolated frames (up to three key segments per q block) with PLAIN riders
// (one segment): inside every XCD's contiguous range the heavy workgroups are listed FIRST, so a range never ends on a
// late-started three-segment workgroup running alone (S = 1024 OUTER with 7 + 7 frames: a workgroup lives 10 - 40 us of a
// 180 us launch).  `per` = workgroups per head, the first `na` of them heavy; a stable partition of the XCD's range, so
// neighbours still share K / V^T in that XCD's L2.  Any (na, per) gives a bijection: a wrong hint only costs balance.
__device__ __forceinline__ int heavy_first(int bid, int nblocks, int na, int per) {
    constexpr int NX = 8;
    const int q = nblocks / NX, r = nblocks % NX;
    const int x = bid % NX, j = bid / NX;
    const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    const int end = base + q + (x < r ? 1 : 0);
    const int nb = per - na;
    const int a0 = (base / per) * na + min(base % per, na);         // heavy ids below `base`
    const int a1 = (end / per) * na + min(end % per, na);
    if (j < a1 - a0) {
        const int t = a0 + j;
        return (t / na) * per + t % na;
    }
    const int t = (base - a0) + (j - (a1 - a0));                     // light ids below `base`, plus the position among the lights
    return (t / nb) * per + na + t % nb;
}


// Host side: state that is per DEVICE, not per process (hipFuncSetAttribute applies to the current device only, and a
// process may drive several GPUs).  slot() returns the current device's entry, or nullptr when there is no device.
template <typename V>
struct PerDevice {
    static constexpr int MAXD = 64;
    V v[MAXD] = {};
    V* slot() {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= MAXD) return nullptr;
        return &v[d];
    }
};

}  // namespace aid
