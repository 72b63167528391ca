// calib_fetch.hip -- calibration of rocprofv3's FETCH_SIZE for the access shape the
// placement kernel uses (8 bytes per lane, 512 contiguous bytes per wave-instruction).
// MI355X_MICROARCH.md (HBM section): FETCH_SIZE is exact only for some widths; calibrate
// on a known byte count in your own access pattern before trusting an absolute.
//
//   seq : every wave reads consecutive 512-B chunks of a 4 GiB buffer, each byte once
//   rnd : every wave reads 512-B chunks at pseudo-random 8-byte-aligned offsets
// Known bytes requested = chunks * 512 in both.  Build: hipcc -O3 --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__global__ void stream_seq(const uint2 *buf, uint64_t chunks, uint2 *sink)
{
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const int lane = threadIdx.x & 63;
    uint2 acc = make_uint2(0, 0);
    for (uint64_t c = wave; c < chunks; c += waves) {
        const uint2 v = buf[c * 64 + lane];
        acc.x ^= v.x;
        acc.y += v.y;
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = acc;
}

__global__ void stream_rnd(const uint2 *buf, uint64_t n_elems, uint64_t chunks, uint2 *sink)
{
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const int lane = threadIdx.x & 63;
    uint2 acc = make_uint2(0, 0);
    for (uint64_t c = wave; c < chunks; c += waves) {
        uint64_t h = c * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29;
        h *= 0xBF58476D1CE4E5B9ull;
        h ^= h >> 32;
        const uint64_t start = h % (n_elems - 64);
        const uint2 v = buf[start + lane];
        acc.x ^= v.x;
        acc.y += v.y;
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = acc;
}

int main()
{
    const uint64_t bytes = 4ull << 30;
    const uint64_t n_elems = bytes / 8;
    uint2 *buf, *sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
    (void)hipMemset(buf, 1, bytes);
    const uint64_t chunks = n_elems / 64;
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
        float ms;
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(stream_seq, dim3(2048), dim3(256), 0, 0, buf, chunks, sink);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        (void)hipEventElapsedTime(&ms, a, b);
        printf("seq: %llu bytes requested, %.3f ms, %.1f GB/s\n", (unsigned long long)(chunks * 512), ms,
               chunks * 512 / ms / 1e6);
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(stream_rnd, dim3(2048), dim3(256), 0, 0, buf, n_elems, chunks, sink);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        (void)hipEventElapsedTime(&ms, a, b);
        printf("rnd: %llu bytes requested, %.3f ms, %.1f GB/s\n", (unsigned long long)(chunks * 512), ms,
               chunks * 512 / ms / 1e6);
    }
    return 0;
}
