// probe_addtid.hip -- what is "TID" in ds_read_addtid_b32 / ds_write_addtid_b32 on gfx950: the lane in
// the wave or the work-item in the workgroup?  256-thread workgroup; every wave reads LDS[M0 + tid*4]
// with M0 = 0 and with M0 = 1024 and reports what it saw.
// Build: hipcc -O2 --offload-arch=gfx950 -o tools/scratch/probe_addtid tools/probe_addtid.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out)
{
    __shared__ unsigned lds[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = (unsigned)i;
    __syncthreads();
    unsigned a, b;
    asm volatile("s_mov_b32 m0, 0\n\ts_nop 0\n\tds_read_addtid_b32 %0 offset:0\n\ts_mov_b32 m0, 1024\n\ts_nop 0\n\tds_read_addtid_b32 %1 offset:16\n\ts_waitcnt lgkmcnt(0)"
                 : "=v"(a), "=v"(b) :: "memory");
    out[threadIdx.x * 2] = a;
    out[threadIdx.x * 2 + 1] = b;
    __syncthreads();
    unsigned v = 1000u + threadIdx.x;
    asm volatile("s_mov_b32 m0, 2048\n\ts_nop 0\n\tds_write_addtid_b32 %0 offset:0\n\ts_waitcnt lgkmcnt(0)" :: "v"(v) : "memory");
    __syncthreads();
    out[512 + threadIdx.x] = lds[512 + threadIdx.x];           // dwords 512.. = byte 2048..
}
int main()
{
    unsigned *d, h[1024];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int t : {0, 1, 63, 64, 65, 128, 255})
        printf("thread %3d: read(M0=0) = dword %4u   read(M0=1024, offset 16) = dword %4u   after write(M0=2048): lds[512+%d] = %u\n",
               t, h[2 * t], h[2 * t + 1], t, h[512 + t]);
    return 0;
}
