#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned* out) {
    unsigned a = threadIdx.x + 100, b = a;
    asm volatile("" : "+v"(b));
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[threadIdx.x * 2] = r[0]; out[threadIdx.x * 2 + 1] = r[1];
}
int main() {
    unsigned* o; (void)hipMalloc(&o, 512);
    k<<<1, 64>>>(o);
    std::vector<unsigned> r(128);
    (void)hipMemcpy(r.data(), o, 512, hipMemcpyDeviceToHost);
    for (int l : {0, 1, 31, 32, 33, 63}) printf("lane %2d: r0=%u r1=%u\n", l, r[l * 2], r[l * 2 + 1]);
    return 0;
}
