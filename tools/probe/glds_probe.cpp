// what does global_load_lds write where?  (probe, not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
__global__ void k(const uint8_t *src, uint8_t *out, int mode)
{
    __shared__ __attribute__((aligned(16))) uint8_t pad[70000];
    __shared__ __attribute__((aligned(16))) uint8_t lds[2048];
    uint8_t *base = mode >= 4 ? pad + 66000 : lds;
    const int lane = threadIdx.x;
    for (int i = lane; i < 2048; i += 64) base[i] = 0xEE;
    __syncthreads();
    if ((mode & 3) == 0) {          // dword, all lanes, lane i reads src dword (63 - i)
        __builtin_amdgcn_global_load_lds((gptr_t)(src + 4 * (63 - lane)), (lptr_t)(base + 16), 4, 0, 0);
    } else if ((mode & 3) == 1) {   // ubyte, all lanes, lane i reads byte 2 i
        __builtin_amdgcn_global_load_lds((gptr_t)(src + 2 * lane), (lptr_t)(base + 16), 1, 0, 0);
    } else if ((mode & 3) == 2) {   // dword, lanes < 48 only
        if (lane < 48) __builtin_amdgcn_global_load_lds((gptr_t)(src + 4 * lane), (lptr_t)(base + 16), 4, 0, 0);
    } else {                        // ubyte, odd lanes only
        if (lane & 1) __builtin_amdgcn_global_load_lds((gptr_t)(src + lane), (lptr_t)(base + 16), 1, 0, 0);
    }
    __syncthreads();
    for (int i = lane; i < 512; i += 64) out[i] = base[i];
}
int main()
{
    std::vector<uint8_t> h(1024);
    for (int i = 0; i < 1024; i++) h[i] = (uint8_t)i;
    uint8_t *d, *o; hipMalloc(&d, 1024); hipMalloc(&o, 512);
    hipMemcpy(d, h.data(), 1024, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 8; mode++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, mode);
        std::vector<uint8_t> r(512); hipMemcpy(r.data(), o, 512, hipMemcpyDeviceToHost);
        printf("mode %d:", mode);
        for (int i = 0; i < 96; i++) printf(" %02x", r[i]);
        printf(" ... [268..283]:");
        for (int i = 268; i < 284; i++) printf(" %02x", r[i]);
        printf("\n");
    }
    return 0;
}
