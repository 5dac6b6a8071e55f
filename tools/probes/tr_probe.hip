// Probe: what does ds_read_b64_tr_b16 return?  LDS is filled with u16 element = its own index; lane l supplies byte address
// (l * 8) [mode 0], or an address pattern [mode 1: row-major 4x16 tile per 16-lane group with row stride 104 halves].
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void probe(unsigned short *out, int mode)
{
    __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    unsigned addr;
    if (mode == 0) addr = (unsigned)(size_t)lds + l * 8;
    else {
        const int t = l & 15, g = l >> 4;
        // chunk t of the group's tile: row t / 4 (stride 104 halves), column group t % 4; groups at column blocks 16 g
        addr = (unsigned)(size_t)lds + (unsigned)((t / 4) * 104 + 16 * g + 4 * (t % 4)) * 2u;
    }
    unsigned lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(*(uint2 *)&lo) : "v"(addr));
    (void)hi;
}
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__global__ void probe2(unsigned short *out, int mode)
{
    __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    unsigned addr;
    if (mode == 0) addr = (unsigned)(size_t)lds + l * 8;
    else {
        const int t = l & 15, g = l >> 4;
        addr = (unsigned)(size_t)lds + (unsigned)((t / 4) * 104 + 16 * g + 4 * (t % 4)) * 2u;
    }
    v2u r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr));
    out[l * 4 + 0] = (unsigned short)(r[0] & 0xffff); out[l * 4 + 1] = (unsigned short)(r[0] >> 16);
    out[l * 4 + 2] = (unsigned short)(r[1] & 0xffff); out[l * 4 + 3] = (unsigned short)(r[1] >> 16);
}
int main()
{
    unsigned short *d, h[256];
    hipMalloc(&d, 512);
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(probe2, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
    }
    return 0;
}
