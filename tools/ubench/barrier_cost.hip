// What does one s_barrier interval cost a 512-thread workgroup (8 waves, two per SIMD, one workgroup per CU) when nothing else happens?
// clock64 ticks per loop trip of { s_barrier }, { s_barrier ; 20 wait states }, { s_barrier ; scalar loop control as in the attention loop }.
//   hipcc --offload-arch=gfx950 -O2 barrier_cost.hip -o barrier_cost
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ __launch_bounds__(512) void k(long long* out, int n, int lds_bytes) {
    extern __shared__ char smem[];
    if (lds_bytes < 0) smem[threadIdx.x] = 0;
    const int grp = (threadIdx.x >> 6) >> 2;
    __builtin_amdgcn_s_barrier();
    if (KIND == 3 && grp == 1) __builtin_amdgcn_s_barrier();
    const long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
        if (KIND == 1 || KIND == 3) asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
        if (KIND == 2) asm volatile("s_sleep 1" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    const long long t1 = clock64();
    if (KIND == 3 && grp == 0) __builtin_amdgcn_s_barrier();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
template <int KIND>
void run(const char* what, long long* d, int n) {
    hipFuncSetAttribute((const void*)k<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 160 * 1024, 0, d, n, 160 * 1024);
    long long h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < 256; ++i) s += (double)h[i];
    printf("%-44s %7.1f clock64 ticks per barrier interval\n", what, s / 256 / n);
}
int main() {
    long long* d;
    (void)hipMalloc(&d, 256 * 8);
    run<0>("s_barrier only", d, 20000);
    run<1>("s_barrier + 20 wait states", d, 20000);
    run<2>("s_barrier + s_sleep 1", d, 20000);
    run<3>("groups one barrier apart + 20 wait states", d, 20000);
    return 0;
}
