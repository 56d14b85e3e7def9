// Issue / execution rate of the operations the fc1 (NORM + GELU) epilogue is made of, per wave64 and SIMD, at 1 and 2 waves per
// SIMD: full-rate VALU (v_fma_f32), packed f32 (v_pk_fma_f32), the transcendentals (v_exp_f32, v_rcp_f32) and the epilogue's own
// mix (per value pair: 6 packed + 2 v_min + 4 transcendentals).  Eight independent register chains per wave, s_memtime (shader
// clock) around the loop.  Build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 tools/isa_probes/valu_trans_rate_probe.hip -o /tmp/vt && /tmp/vt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float f32x2 __attribute__((ext_vector_type(2)));

#define R8(OP) OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)

template <int MODE>
__global__ void probe(long long* ticks, float* sink, int iters) {
    float a0 = threadIdx.x * 1e-3f + 0.1f, a1 = a0 + 0.01f, a2 = a0 + 0.02f, a3 = a0 + 0.03f, a4 = a0 + 0.04f, a5 = a0 + 0.05f,
          a6 = a0 + 0.06f, a7 = a0 + 0.07f;
    f32x2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = p0 + 1.0f, p5 = p1 + 1.0f, p6 = p2 + 1.0f, p7 = p3 + 1.0f;
    const float c = 0.999f, d = 1e-3f;
    const f32x2 c2 = {c, c}, d2 = {d, d};
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {
#define OP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
            R8(OP)
#undef OP
        } else if constexpr (MODE == 1) {
#define OP(x) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
            R8(OP)
#undef OP
        } else if constexpr (MODE == 2) {
#define OP(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d));
            R8(OP)
#undef OP
        } else if constexpr (MODE == 3) {
#define OP(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c2), "v"(d2));
            OP(p0) OP(p1) OP(p2) OP(p3) OP(p4) OP(p5) OP(p6) OP(p7)
#undef OP
        } else if constexpr (MODE == 4) {
            // the product GELU per value pair x 4 pairs: pk_mul, 2 min, 2 pk_fma, pk_mul, 2 exp, pk_add, 2 rcp, pk_mul  (+ 2 pk_fma norm)
#define PAIR(p, q)                                                                   \
    asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %0, %0, %2, %3\n v_pk_mul_f32 %1, %0, %0" \
                 : "+v"(p), "+v"(q) : "v"(c2), "v"(d2));                             \
    asm volatile("v_min_f32 %0, 0x42480000, %0\n v_min_f32 %1, 0x42480000, %1" : "+v"(q[0]), "+v"(q[1]));          \
    asm volatile("v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_mul_f32 %1, %1, %0\n" \
                 : "+v"(p), "+v"(q) : "v"(c2), "v"(d2));                             \
    asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1" : "+v"(q[0]), "+v"(q[1]));    \
    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(q) : "v"(c2));                     \
    asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1" : "+v"(q[0]), "+v"(q[1]));    \
    asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(q));
            PAIR(p0, p4) PAIR(p1, p5) PAIR(p2, p6) PAIR(p3, p7)
#undef PAIR
        } else if constexpr (MODE == 5) {
            // the same pairs without the transcendentals (what the ordinary VALU alone costs)
#define PAIR(p, q)                                                                   \
    asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %0, %0, %2, %3\n v_pk_mul_f32 %1, %0, %0" \
                 : "+v"(p), "+v"(q) : "v"(c2), "v"(d2));                             \
    asm volatile("v_min_f32 %0, 0x42480000, %0\n v_min_f32 %1, 0x42480000, %1" : "+v"(q[0]), "+v"(q[1]));          \
    asm volatile("v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_mul_f32 %1, %1, %0\n" \
                 : "+v"(p), "+v"(q) : "v"(c2), "v"(d2));                             \
    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(q) : "v"(c2));                     \
    asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(q));
            PAIR(p0, p4) PAIR(p1, p5) PAIR(p2, p6) PAIR(p3, p7)
#undef PAIR
        } else {
            // the transcendentals of four pairs alone
#define OP(x) asm volatile("v_exp_f32 %0, %0\n v_rcp_f32 %0, %0" : "+v"(x));
            R8(OP)
#undef OP
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0[0] + p1[1] + p2[0] + p3[1] + p4[0] + p5[1] + p6[0] + p7[1];
}

template <int MODE>
double run(int threads, int iters, long long* dt, float* sink) {
    const int blocks = 256;
    probe<MODE><<<blocks, threads>>>(dt, sink, iters);
    probe<MODE><<<blocks, threads>>>(dt, sink, iters);
    hipDeviceSynchronize();
    const int n = blocks * (threads / 64);
    std::vector<long long> h(n);
    hipMemcpy(h.data(), dt, n * sizeof(long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    return (double)h[n / 2] / iters;
}

int main() {
    long long* dt; float* sink;
    hipMalloc(&dt, 256 * 8 * sizeof(long long));
    hipMalloc(&sink, 256 * 512 * sizeof(float));
    const int iters = 4000;
    printf("shader-clock ticks per wave (median over waves), 256 workgroups; 256 threads = 1 wave / SIMD, 512 = 2 waves / SIMD\n");
    printf("%-44s %12s %12s\n", "loop body", "1 wave/SIMD", "2 waves/SIMD");
#define ROW(MODE, NAME, PER)                                                          \
    {                                                                                 \
        const double a = run<MODE>(256, iters, dt, sink) / (PER), b = run<MODE>(512, iters, dt, sink) / (PER); \
        printf("%-44s %12.2f %12.2f\n", NAME, a, b);                                  \
    }
    ROW(0, "v_exp_f32 (per instruction)", 8.0)
    ROW(1, "v_rcp_f32 (per instruction)", 8.0)
    ROW(2, "v_fma_f32 (per instruction)", 8.0)
    ROW(3, "v_pk_fma_f32 (per instruction)", 8.0)
    ROW(4, "NORM+GELU mix (per VALUE: 8 values / iter)", 8.0)
    ROW(5, "  ... its ordinary VALU alone (per value)", 8.0)
    ROW(6, "  ... its exp + rcp alone (per value)", 8.0)
    return 0;
}
