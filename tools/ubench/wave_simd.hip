// Which SIMD does wave w of a 512-thread workgroup land on?  (HW_REG_HW_ID: wave_id [3:0], simd_id [5:4], cu_id [11:8], se_id ...)
// The ping-pong kernels assume waves w and w + 4 share a SIMD.   hipcc --offload-arch=gfx950 -O2 wave_simd.hip -o wave_simd
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(unsigned* out, int lds_bytes) {
    extern __shared__ char smem[];
    if (lds_bytes < 0) smem[threadIdx.x] = 0;
    const unsigned id = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
}
int main() {
    unsigned* d;
    const int nb = 512;
    hipMalloc(&d, nb * 8 * 4);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int lds : {0, 160 * 1024}) {
        hipLaunchKernelGGL(k, dim3(nb), dim3(512), lds, 0, d, lds);
        unsigned h[nb * 8];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        int pair_ok = 0, rr = 0, hist[8][4] = {};
        for (int b = 0; b < nb; ++b) {
            bool ok = true, r = true;
            for (int w = 0; w < 8; ++w) {
                const int simd = (h[b * 8 + w] >> 4) & 3;
                hist[w][simd]++;
                if (w >= 4 && simd != (int)((h[b * 8 + w - 4] >> 4) & 3)) ok = false;
                if (simd != (w & 3)) r = false;
            }
            pair_ok += ok; rr += r;
        }
        printf("dynamic LDS %6d B: %d of %d workgroups have waves w and w + 4 on one SIMD; %d are plain round-robin (wave w on SIMD w %% 4)\n", lds, pair_ok, nb, rr);
        for (int w = 0; w < 8; ++w) printf("   wave %d: SIMD histogram %d %d %d %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
        printf("   first workgroup: ");
        for (int w = 0; w < 8; ++w) printf("w%d simd %u wave_id %u cu %u | ", w, (h[w] >> 4) & 3, h[w] & 15, (h[w] >> 8) & 15);
        printf("\n");
    }
    return 0;
}
