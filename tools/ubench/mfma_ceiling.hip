// What the matrix pipe of THIS box sustains on the operands the benchmark uses (round 6).
// Dense v_mfma_f32_32x32x16 issue, nothing else: every CU runs four waves (one per SIMD), each wave 64 back-to-back MFMAs per loop
// trip on 16 rotating accumulators (no dependency stall: 32.0 cycles per MFMA and SIMD, tools/ubench/mfma_cadence.hip), operands
// held in registers.  The instruction stream is the same for every data set; what changes is the CLOCK the power budget allows:
//   zeros            the chip stays near its top clock (2.3 - 2.4 GHz under this load): 2.43 - 2.49 PFLOP/s, the data-sheet figure
//   random operands  (N(0, 1) samples rounded to bf16 / f16 — the benchmark's synthetic activations) the clock drops until the power fits
//                    (1.6 - 1.7 GHz): 1.66 - 1.82 PFLOP/s, the POWER-LIMITED dense rate — the ceiling a real GEMM or attention kernel of
//                    this library meets first (MI355X_MICROARCH.md "DVFS give-back"; profiles/r06_notes.md section 6).
// Prints one JSON line; bench.py runs this binary (when present) and reports the rate next to the nominal 2.5 PFLOP/s peak.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_ceiling mfma_ceiling.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <vector>
typedef _Float16 f16;
typedef __bf16 bf16;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

template <bool BF>
__global__ __launch_bounds__(256) void k(const unsigned short* __restrict__ src, float* out, long long* cyc, int iters) {
    typedef bf16 b8 __attribute__((ext_vector_type(8)));
    typedef f16 h8 __attribute__((ext_vector_type(8)));
    // 32 fragments per lane (16 A, 16 B), 16 bytes each, from the data set: every MFMA of a trip reads another operand pair
    u16x8 fr[32];
    const unsigned short* p = src + ((size_t)blockIdx.x * 256 + threadIdx.x) * 256;
#pragma unroll
    for (int i = 0; i < 32; ++i) fr[i] = *reinterpret_cast<const u16x8*>(p + 8 * i);
    f32x16 acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    __syncthreads();
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 64; ++g) {
            const int i = g & 15, j = (g >> 2) & 15;
            if (BF) acc[g & 15] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, fr[16 + j]), __builtin_bit_cast(b8, fr[i]), acc[g & 15], 0, 0, 0);
            else    acc[g & 15] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, fr[16 + j]), __builtin_bit_cast(h8, fr[i]), acc[g & 15], 0, 0, 0);
        }
        // keep the accumulators bounded without touching the instruction mix: nothing — fp32 has the range for iters * 64 * 16 products
    }
    const long long c1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = c1 - c0;
}

struct Res { double tflops, ghz, cyc_per_mfma; };

template <bool BF>
Res run(const unsigned short* data, float* out, long long* cyc, int ncu, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 5; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<BF>, dim3(ncu), dim3(256), 0, 0, data, out, cyc, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it && ms < best) best = ms;
    }
    std::vector<long long> hc(ncu * 4);
    hipMemcpy(hc.data(), cyc, hc.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double cs = 0; for (auto c : hc) cs += (double)c;
    cs /= hc.size();
    const double mfmas = (double)iters * 64;
    Res r;
    r.tflops = mfmas * 4 * ncu * 32768.0 / (best * 1e-3) / 1e12;       // 2 * 32 * 32 * 16 flops per MFMA, four waves per CU
    r.cyc_per_mfma = cs / mfmas;
    r.ghz = cs / (best * 1e-3) * 1e-9;
    return r;
}

int main(int argc, char** argv) {
    int dev = 0, ncu = 256;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    const int iters = 4000;                                                   // ~0.5 - 0.8 ms per launch
    const size_t n = (size_t)ncu * 256 * 256;
    std::vector<unsigned short> h(n);
    unsigned long long x = 88172645463325252ull;
    unsigned short *d_rand_bf, *d_rand_h, *d_zero;
    float* out; long long* cyc;
    hipMalloc(&d_rand_bf, n * 2); hipMalloc(&d_rand_h, n * 2); hipMalloc(&d_zero, n * 2);
    hipMalloc(&out, (size_t)ncu * 256 * 4); hipMalloc(&cyc, (size_t)ncu * 4 * sizeof(long long));
    // standard-normal samples (Box-Muller over a xorshift stream) rounded to bf16 / f16: the statistics of the benchmark's synthetic
    // activations (~N(0, 1)); weights ~N(0, 1 / fan_in) have the same mantissa / sign statistics, other exponents
    auto uni = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return ((x >> 11) + 1) * (1.0 / 9007199254740993.0); };
    auto gauss = [&]() { const double u = uni(), v = uni(); return sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v); };
    for (size_t i = 0; i < n; ++i) {
        const float g = (float)gauss();
        unsigned int b; memcpy(&b, &g, 4);
        b += 0x7fff + ((b >> 16) & 1);                                        // round to nearest even
        h[i] = (unsigned short)(b >> 16);
    }
    hipMemcpy(d_rand_bf, h.data(), n * 2, hipMemcpyHostToDevice);
    for (size_t i = 0; i < n; ++i) {
        const f16 v = (f16)(float)gauss();
        memcpy(&h[i], &v, 2);
    }
    hipMemcpy(d_rand_h, h.data(), n * 2, hipMemcpyHostToDevice);
    hipMemset(d_zero, 0, n * 2);
    const Res zb = run<true>(d_zero, out, cyc, ncu, iters);
    const Res rb = run<true>(d_rand_bf, out, cyc, ncu, iters);
    const Res rh = run<false>(d_rand_h, out, cyc, ncu, iters);
    const Res rb2 = run<true>(d_rand_bf, out, cyc, ncu, iters);              // again: the clock has settled
    printf("{\"cus\": %d, \"bf16_zeros_tflops\": %.1f, \"bf16_zeros_ghz\": %.3f, \"bf16_random_tflops\": %.1f, \"bf16_random_ghz\": %.3f, "
           "\"f16_random_tflops\": %.1f, \"f16_random_ghz\": %.3f, \"cycles_per_mfma\": %.2f, "
           "\"what\": \"dense v_mfma_f32_32x32x16 issue from four waves per CU, operands in registers, nothing else; random = N(0, 1) samples rounded to the storage type\"}\n",
           ncu, zb.tflops, zb.ghz, (rb.tflops < rb2.tflops ? rb.tflops : rb2.tflops), (rb.ghz < rb2.ghz ? rb.ghz : rb2.ghz), rh.tflops, rh.ghz, rb2.cyc_per_mfma);
    return 0;
}
