// experiment: semantics of global_load_lds (LDS-direct loads) for 2-byte elements on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(const uint16_t* __restrict__ src, uint32_t* __restrict__ out, const int* __restrict__ idx) {
    __shared__ uint32_t lds[256];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    const uint16_t* g = src + idx[lane];
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g, (void __attribute__((address_space(3)))*)(lds + 64), 2, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0)
    __syncthreads();
    for (int i = lane; i < 256; i += 64) out[i] = lds[i];
}
int main() {
    uint16_t h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (uint16_t)(0x1000 + i);
    int hidx[64]; for (int i = 0; i < 64; ++i) hidx[i] = (i * 7) % 1000;
    uint16_t* d; uint32_t* o; int* di;
    hipMalloc(&d, sizeof h); hipMalloc(&o, 1024); hipMalloc(&di, sizeof hidx);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice); hipMemcpy(di, hidx, sizeof hidx, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o, di);
    uint32_t r[256]; hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost);
    for (int i = 56; i < 140; ++i) printf("%d:%08x%s", i, r[i], (i % 8 == 7) ? "\n" : " ");
    printf("\n");
    return 0;
}
