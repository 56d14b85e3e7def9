// Does v_mfma_f32_32x32x16_f16 honour f16 SUBNORMAL operands?  A = 2^-20 (subnormal) in every slot, B = 1: C should be 16 * 2^-20.
// build: hipcc --offload-arch=gfx950 -o /tmp/mfma_f16_denorm tools/isa_probes/mfma_f16_denorm.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float a, float b) {
    f16x8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = (_Float16)a; B[i] = (_Float16)b; }
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
    float* d; hipMalloc(&d, 4);
    const float cases[][2] = {{9.5367431640625e-07f, 1.0f}, {1.0f, 9.5367431640625e-07f}, {3.0517578125e-05f, 3.0517578125e-05f}, {5.9604644775390625e-08f, 1024.0f}};
    for (auto& c : cases) {
        k<<<1, 64>>>(d, c[0], c[1]);
        float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("a=%g b=%g: mfma %.9g expected %.9g\n", c[0], c[1], h, 16.0 * (double)(float)(_Float16)c[0] * (double)(float)(_Float16)c[1]);
    }
    return 0;
}
