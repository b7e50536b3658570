// How fast can the Q rows of ONE head be streamed?  q / out are [rows][C] (C = heads * 64 channels, 2 B each): a (frame, head)
// item reads 128 B out of every 2560-B row.  Copy kernels with the access orders a 77-key cross-attention kernel can have:
//   A  workgroup = 128 rows of one head (4 waves x 32 rows; lane (n, h) reads 4 x 16 B of row n), grid = heads x row blocks
//      (what aid_attn_kernel / aid_attn_xs do), head-major and row-block-major block order
//   B  workgroup = 64 rows x ALL heads, waves loop over heads (each wave: 32 rows x 128 B per trip, head after head)
//   C  linear copy (16 B per lane, consecutive lanes consecutive addresses): the ceiling
// usage: head_stride_copy [rows = 14336] [C = 1280]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void copy_a(const char* __restrict__ q, char* __restrict__ o, int rows, int cb, int heads, int rowmajor) {
    const int nb = rows / 128;
    const int b = blockIdx.x;
    const int head = rowmajor ? b % heads : b / nb, rb = rowmajor ? b / heads : b % nb;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t row = (size_t)rb * 128 + wave * 32 + (lane & 31);
    const char* src = q + row * cb + head * 128 + (lane >> 5) * 16;
    char* dst = o + row * cb + head * 128 + (lane >> 5) * 16;
    u32x4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const u32x4*>(src + 32 * j);
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x4*>(dst + 32 * j) = v[j];
}

__global__ __launch_bounds__(256) void copy_a_lin(const char* __restrict__ q, char* __restrict__ o, int rows, int cb, int heads, int rowmajor) {
    // same items, but a wave instruction covers 8 whole 128-B row segments (lane = 16-B piece of row lane / 8)
    const int nb = rows / 128;
    const int b = blockIdx.x;
    const int head = rowmajor ? b % heads : b / nb, rb = rowmajor ? b / heads : b % nb;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    u32x4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const size_t row = (size_t)rb * 128 + wave * 32 + j * 8 + (lane >> 3);
        v[j] = *reinterpret_cast<const u32x4*>(q + row * cb + head * 128 + (lane & 7) * 16);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const size_t row = (size_t)rb * 128 + wave * 32 + j * 8 + (lane >> 3);
        *reinterpret_cast<u32x4*>(o + row * cb + head * 128 + (lane & 7) * 16) = v[j];
    }
}

__global__ __launch_bounds__(256) void copy_b(const char* __restrict__ q, char* __restrict__ o, int rows, int cb, int heads, int rpb) {
    // workgroup = rpb rows x all heads; wave w takes 32-row groups, loops over heads; one instruction = 8 row segments of 128 B
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t row0 = (size_t)blockIdx.x * rpb;
    for (int g = wave; g < rpb / 32; g += 4)
        for (int head = 0; head < heads; ++head) {
            u32x4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const size_t row = row0 + g * 32 + j * 8 + (lane >> 3);
                v[j] = *reinterpret_cast<const u32x4*>(q + row * cb + head * 128 + (lane & 7) * 16);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const size_t row = row0 + g * 32 + j * 8 + (lane >> 3);
                *reinterpret_cast<u32x4*>(o + row * cb + head * 128 + (lane & 7) * 16) = v[j];
            }
        }
}

__global__ __launch_bounds__(256) void copy_c(const u32x4* __restrict__ q, u32x4* __restrict__ o, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) o[i] = q[i];
}

int main(int argc, char** argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 14336, c = argc > 2 ? atoi(argv[2]) : 1280;
    const int cb = c * 2, heads = c / 64;
    const size_t bytes = (size_t)rows * cb;
    char *q, *o;
    hipMalloc(&q, bytes); hipMalloc(&o, bytes); hipMemset(q, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-58s %8.2f us  %7.1f GB/s (read + write)\n", name, ms * 50, 2.0 * bytes / (ms / 20 * 1e-3) * 1e-9);
    };
    printf("rows %d  C %d  (%d heads)  %.1f MB each way\n", rows, c, heads, bytes * 1e-6);
    timeit("A  head-major blocks, lane = (row, half)", [&] { copy_a<<<heads * rows / 128, 256>>>(q, o, rows, cb, heads, 0); });
    timeit("A  row-block-major blocks, lane = (row, half)", [&] { copy_a<<<heads * rows / 128, 256>>>(q, o, rows, cb, heads, 1); });
    timeit("A' head-major blocks, 8 whole segments / instruction", [&] { copy_a_lin<<<heads * rows / 128, 256>>>(q, o, rows, cb, heads, 0); });
    timeit("A' row-block-major blocks, 8 whole segments / instruction", [&] { copy_a_lin<<<heads * rows / 128, 256>>>(q, o, rows, cb, heads, 1); });
    timeit("B  workgroup = 64 rows x all heads", [&] { copy_b<<<rows / 64, 256>>>(q, o, rows, cb, heads, 64); });
    timeit("B  workgroup = 32 rows x all heads (1 active wave)", [&] { copy_b<<<rows / 32, 256>>>(q, o, rows, cb, heads, 32); });
    timeit("B  workgroup = 128 rows x all heads", [&] { copy_b<<<rows / 128, 256>>>(q, o, rows, cb, heads, 128); });
    timeit("C  linear copy, 2048 blocks", [&] { copy_c<<<2048, 256>>>((const u32x4*)q, (u32x4*)o, bytes / 16); });
    return 0;
}
