#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x;
    // each lane points at 4 contiguous elements: chunk s = l & 15 of a 4-row x 16-col block (row stride 64 elems)
    const int s = l & 15, grp = l >> 4;
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)lds
                          + ((grp * 4 + (s >> 2)) * 64 + 4 * (s & 3)) * 2;
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[l * 4 + 0] = v[0] & 0xffff; out[l * 4 + 1] = v[0] >> 16;
    out[l * 4 + 2] = v[1] & 0xffff; out[l * 4 + 3] = v[1] >> 16;
}
int main() {
    uint16_t* o; hipMalloc(&o, 512);
    k<<<1, 64>>>(o);
    std::vector<uint16_t> r(256);
    hipMemcpy(r.data(), o, 512, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) printf(" (r%d,c%d)", r[l * 4 + j] / 64, r[l * 4 + j] % 64);
        printf("\n");
    }
    return 0;
}
