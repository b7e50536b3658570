// Does the matrix pipe keep SUBNORMAL f16 / bf16 inputs?  A = 2^-20 (f16 subnormal) resp. 2^-130 (bf16 subnormal) in every element,
// B = 1: D[i][j] = 16 * A if subnormal inputs are honoured, 0 if they are flushed.  (The attention kernels' P = 2^(s - m) values
// reach the subnormal range of f16 for keys 2^-14 below the row reference.)   hipcc --offload-arch=gfx950 -O2 mfma_denorm.hip -o mfma_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(float* out, unsigned short hbits, unsigned short bbits) {
    h8 a, b; b8 c, d;
    for (int i = 0; i < 8; ++i) {
        a[i] = __builtin_bit_cast(_Float16, hbits); b[i] = (_Float16)1.0f;
        c[i] = __builtin_bit_cast(__bf16, bbits);   d[i] = (__bf16)1.0f;
    }
    f16v z = {0}, r1, r2;
    r1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, z, 0, 0, 0);
    r2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c, d, z, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = r1[0]; out[1] = r2[0]; }
}
int main() {
    float* d; (void)hipMalloc(&d, 8);
    const unsigned short hb[] = {0x0010, 0x0001, 0x03ff, 0x0400};      // 2^-20, 2^-24 (smallest), largest subnormal, smallest normal 2^-14
    const unsigned short bb[] = {0x0008, 0x0001, 0x007f, 0x0080};      // bf16: 2^-130, 2^-133, largest subnormal, smallest normal 2^-126
    for (int i = 0; i < 4; ++i) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, hb[i], bb[i]);
        float h[2]; (void)hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("f16 input bits 0x%04x: sum of 16 = %.6e   |   bf16 input bits 0x%04x: sum of 16 = %.6e\n", hb[i], h[0], bb[i], h[1]);
    }
    printf("expected when subnormals are kept: f16 16 * 2^-20 = %.6e, 16 * 2^-24 = %.6e; bf16 16 * 2^-130 = %.6e\n", 16 * 9.5367431640625e-07, 16 * 5.9604644775390625e-08, 16 * 7.346839692639297e-40);
    return 0;
}
