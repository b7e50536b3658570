// development microbenchmark (round 3): what does writing one CU round of C tiles cost?  256 workgroups x 512 threads, each
// writes a 288 x 256 bf16 tile (147 KB) of a row-major [M, 1280] matrix with 16-B stores — the GEMM epilogue's global side.
//   MODE 0  tile pattern as the epilogue issues it: 18 passes, a wave covers 2 rows x 512 B          (rows 2560 B apart)
//   MODE 1  the same bytes as one flat contiguous 147 KB block per workgroup
//   MODE 2  tile pattern, non-temporal stores
//   MODE 3  tile pattern, each thread's 18 stores issued back to back from registers (no loop-carried address math)
// build: hipcc --offload-arch=gfx950 -O3 -o store_bw store_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned short* c, int ldc, int tiles_n) {
    const int tile = blockIdx.x, tm = tile / tiles_n, tn = tile % tiles_n;
    const int tid = threadIdx.x;
    u32x4 v = {(unsigned)tid, 1u, 2u, (unsigned)tile};
    if (MODE == 1) {
        unsigned short* base = c + (size_t)tile * 288 * 256;
#pragma unroll
        for (int it = 0; it < 18; ++it) *reinterpret_cast<u32x4*>(base + (size_t)(tid + it * 512) * 8) = v;
        return;
    }
    unsigned short* base = c + (size_t)tm * 288 * ldc + tn * 256;
#pragma unroll
    for (int it = 0; it < 18; ++it) {
        const int id = tid + it * 512, row = id >> 5, ch = (id & 31) * 8;
        u32x4* dst = reinterpret_cast<u32x4*>(base + (size_t)row * ldc + ch);
        if (MODE == 2) __builtin_nontemporal_store(v, dst);
        else           *dst = v;
    }
}

template <int MODE>
void run(const char* tag, unsigned short* c) {
    const int tiles_n = 5, tiles = 250, ldc = 1280;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) k<MODE><<<tiles, 512>>>(c, ldc, tiles_n);
    hipDeviceSynchronize();
    const int iters = 20;
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) k<MODE><<<tiles, 512>>>(c, ldc, tiles_n);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / iters, bytes = 250.0 * 288 * 256 * 2;
    printf("%-60s %7.2f us per launch  %6.2f TB/s\n", tag, us, bytes / us / 1e6);
}

int main() {
    unsigned short* c;
    hipMalloc(&c, (size_t)14400 * 1280 * 2 + (1 << 20));
    run<0>("tile pattern (2 rows x 512 B per wave store)", c);
    run<1>("flat 147 KB per workgroup", c);
    run<2>("tile pattern, non-temporal", c);
    run<0>("tile pattern again", c);
    return 0;
}
