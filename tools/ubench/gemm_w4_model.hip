// development microbenchmark (round 6): K-loop model of a FOUR-wave, one-wave-per-SIMD 256 x 256 x 64 NT GEMM tile — the structure
// hipBLASLt-class kernels use — to price it against the eight-wave ping-pong loop of csrc/aid_gemm.hip (1.49 us per 288 x 256 x 64
// K tile; 1.32 us per 256 x 256 x 64 at the same rate per flop) BEFORE building an engine around it.
//   wave (wr, wc) owns a 128 x 128 wave tile (4 x 4 blocks of v_mfma_f32_32x32x16_bf16, 256 accumulator registers), reads
//   A half wr / B half wc of the K tile from LDS (8 ds_read_b128 per k-step of 16 MFMAs, one k-step ahead, two register sets),
//   requests its 16 LDS-DMA pieces of the NEXT K tile (other parity) in the first two k-steps, waits for them at the end of the third,
//   one s_barrier per K tile.  Fillers are pinned one per MFMA gap (sched_barrier).
//   ABL bit 0: no DMA requests, bit 1: no fragment reads (timing ablations).
// Real operands (A 14336 x 1280, B 1280 x 1280 bf16, XOR-swizzled 128-byte LDS rows as in the product), result summed so nothing is
// dead.  Reports us per K tile per CU by wall clock (events) over 280 workgroups of one tile each... and the MFMA floor.
// build: hipcc --offload-arch=gfx950 -O3 -o gemm_w4_model gemm_w4_model.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __amdgpu_buffer_rsrc_t Rsrc;

constexpr int HALF = 128 * 128;          // bytes: 128 rows x 128 B (64 bf16 of K)
constexpr int STG = 4 * HALF;            // one parity: A0 A1 B0 B1

__device__ __forceinline__ int swz(int r) { return (r >> 1) & 7; }
__device__ __forceinline__ void pin() { __builtin_amdgcn_sched_barrier(0); }

template <int ABL>
__global__ __launch_bounds__(256) void k(const bf16* __restrict__ A, const bf16* __restrict__ B, float* out, long long* cyc, int nk, int lda,
                                         int ldb, int tiles_n, int reps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int m0 = (blockIdx.x / tiles_n) * 256, n0 = (blockIdx.x % tiles_n) * 256;
    const Rsrc ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(A + (size_t)m0 * lda), 0, 0x7fffffff, 0x00020000);
    const Rsrc rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(B + (size_t)n0 * ldb), 0, 0x7fffffff, 0x00020000);
    // DMA pieces: item q (A0 A1 B0 B1) has 16 pieces of 8 rows; this wave takes pieces 4 wave .. 4 wave + 3 of every item
    int avo[2][4], bvo[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = (4 * wave + j) * 8 + (lane >> 3);                  // row inside the half
            const int c = (lane & 7) ^ swz(row);
            avo[h][j] = (h * 128 + row) * (lda * 2) + c * 16;
            bvo[h][j] = (h * 128 + row) * (ldb * 2) + c * 16;
        }
    auto dma = [&](int idx, int parity, int kt) __attribute__((always_inline)) {      // idx 0 .. 15: item idx >> 2, piece idx & 3
        const int q = idx >> 2, j = idx & 3;
        char* dst = smem + parity * STG + q * HALF + (4 * wave + j) * 1024;
        if (q < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)dst, 16, avo[q][j], kt * 128, 0, 0);
        else       __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)dst, 16, bvo[q - 2][j], kt * 128, 0, 0);
    };
    int aoff[4], boff[4], ax[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 32 * i + l31;
        aoff[i] = wr * HALF + row * 128;
        boff[i] = (2 + wc) * HALF + row * 128;
        ax[i] = hi ^ swz(row);
    }
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8 fa[2][4], fb[2][4];
    auto rd = [&](int set, int parity, int ks, int w) __attribute__((always_inline)) {   // w 0 .. 7: A block w, or B block w - 4
        const char* st = smem + parity * STG;
        if (w < 4) fa[set][w] = *reinterpret_cast<const bf16x8*>(st + aoff[w] + (((2 * ks) ^ ax[w]) << 4));
        else       fb[set][w - 4] = *reinterpret_cast<const bf16x8*>(st + boff[w - 4] + (((2 * ks) ^ ax[w - 4]) << 4));
    };
    const long long c0 = clock64();
    for (int rep = 0; rep < reps; ++rep) {
        // prologue: tile 0 -> parity 0
#pragma unroll
        for (int i = 0; i < 16; ++i) dma(i, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 8; ++w) rd(0, 0, 0, w);
        if (ABL & 2) {                                                         // no reads in the loop: both register sets hold real operands
#pragma unroll
            for (int w = 0; w < 8; ++w) rd(1, 0, 1, w);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        for (int kt = 0; kt < nk; ++kt) {
            const int p = kt & 1;
            const int ktn = kt + 1 < nk ? kt + 1 : 0;                          // (the last tile re-requests tile 0: nobody reads it)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int cur = ks & 1, nxt = cur ^ 1;
                pin();
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int i = g >> 2, j = g & 3;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[cur][j], fa[cur][i], acc[i][j], 0, 0, 0);
                    pin();
                    if (!(ABL & 2) && (g & 1) == 0) {                          // eight reads per k-step: the next k-step's fragments
                        if (ks < 3) rd(nxt, p, ks + 1, g >> 1);
                        else        rd(nxt, p ^ 1, 0, g >> 1);                 // (published by the barrier behind k-step 2)
                        pin();
                    }
                    if (!(ABL & 1) && ks < 2 && (g & 1) == 1) {                // sixteen requests in the first two k-steps
                        dma(8 * ks + (g >> 1), p ^ 1, ktn);
                        pin();
                    }
                }
                if (ks == 2) {
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    pin();
                    __builtin_amdgcn_s_barrier();
                    pin();
                }
            }
        }
        __syncthreads();
    }
    if (tid == 0) cyc[blockIdx.x] = clock64() - c0;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int ABL>
void run(const char* tag, const bf16* A, const bf16* B, float* out, long long* cyc, int m, int n, int kdim) {
    const int tiles_m = m / 256, tiles_n = n / 256, nk = kdim / 64, reps = 8;
    const int grid = tiles_m * tiles_n;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STG);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 6; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<ABL>, dim3(grid), dim3(256), 2 * STG, 0, A, B, out, cyc, nk, kdim, kdim, tiles_n, reps);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it && ms < best) best = ms;
    }
    std::vector<long long> hc(grid);
    hipMemcpy(hc.data(), cyc, grid * sizeof(long long), hipMemcpyDeviceToHost);
    double cs = 0; for (auto c : hc) cs += (double)c;
    const double cyc_tile = cs / grid / (reps * nk);
    const double rounds = (double)((grid + 255) / 256);
    const double us_tile = best * 1e3 / (reps * nk * rounds);
    const double tf = 2.0 * 256 * 256 * 64 * grid * reps * nk / (best * 1e-3) / 1e12;
    printf("%-28s %7.1f us per pass of 20 K tiles  %6.3f us = %5.0f shader cycles per 256x256x64 K tile (2048 = pure MFMA issue)  eff. clock %4.2f GHz  %5.0f TF/s\n",
           tag, best * 1e3 / reps, us_tile, cyc_tile, cyc_tile / us_tile * 1e-3, tf);
}

int main() {
    const int m = 14336, n = 1280, kdim = 1280;
    bf16 *A, *B; float* out;
    hipMalloc(&A, (size_t)m * kdim * 2); hipMalloc(&B, (size_t)n * kdim * 2); hipMalloc(&out, 4096 * 256 * 4);
    long long* cyc; hipMalloc(&cyc, 4096 * sizeof(long long));
    std::vector<unsigned short> h((size_t)m * kdim);
    unsigned long long x = 88172645463325252ull;                              // xorshift: random sign, mantissa and three exponent bits (|v| in [0.06, 16))
    for (size_t i = 0; i < h.size(); ++i) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        h[i] = (unsigned short)(((x >> 20) & 0x83ff) | (((x >> 40) & 7) + 0x7b) << 7);
    }
    hipMemcpy(A, h.data(), (size_t)m * kdim * 2, hipMemcpyHostToDevice);
    hipMemcpy(B, h.data(), (size_t)n * kdim * 2, hipMemcpyHostToDevice);
    printf("four waves, one per SIMD, 256 x 256 x 64 tiles; 13056 x 1280 x 1280 = 255 tiles = ONE round of one tile per CU, 20 K tiles, 8 repetitions inside the launch\n");
    printf("(pure MFMA time of a 256 x 256 x 64 K tile: 2048 cycles per SIMD = 0.98 us at 2.09 GHz, 0.85 us at 2.4 GHz)\n");
    const int m1 = 13056;
    run<0>("full loop", A, B, out, cyc, m1, n, kdim);
    run<1>("no DMA requests", A, B, out, cyc, m1, n, kdim);
    run<2>("no fragment reads", A, B, out, cyc, m1, n, kdim);
    run<3>("MFMAs + barrier only", A, B, out, cyc, m1, n, kdim);
    hipMemset(A, 0, (size_t)m * kdim * 2);
    hipMemset(B, 0, (size_t)n * kdim * 2);
    printf("the same with ALL-ZERO operands (the clock the power budget allows depends on the data):\n");
    run<0>("full loop, zeros", A, B, out, cyc, m1, n, kdim);
    run<3>("MFMAs + barrier only, zeros", A, B, out, cyc, m1, n, kdim);
    return 0;
}
