// probe_mall.hip -- bandwidth of random 512-byte wave reads (the posting-chunk access shape) as a
// function of the working set: what do the 4 MB L2s, the 256 MB Infinity Cache (MALL) and HBM each
// sustain on MI355X?  Total bytes read are the same for every working set.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/scratch/probe_mall tools/probe_mall.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__global__ void __launch_bounds__(256) stream_rnd(const uint2 *buf, uint64_t n_chunks_in_set, uint64_t chunks, uint2 *sink)
{
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const int lane = threadIdx.x & 63;
    uint2 acc = make_uint2(0, 0);
    for (uint64_t c = wave; c < chunks; c += 4 * waves) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // four independent chunks in flight per wave
            uint64_t h = (c + u * waves) * 0x9E3779B97F4A7C15ull;
            h ^= h >> 29;
            h *= 0xBF58476D1CE4E5B9ull;
            h ^= h >> 32;
            const uint64_t chunk = h & (n_chunks_in_set - 1);  // power of two; 512-byte aligned: four whole lines
            const uint2 v = buf[chunk * 64 + lane];
            acc.x ^= v.x;
            acc.y += v.y;
        }
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = acc;
}

int main()
{
    const uint64_t max_bytes = 4ull << 30;
    uint2 *buf, *sink;
    if (hipMalloc(&buf, max_bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
    (void)hipMemset(buf, 1, max_bytes);
    const uint64_t chunks = (16ull << 30) / 512;  // 16 GiB read per run
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    const uint64_t sets_mb[] = {2, 8, 16, 32, 64, 128, 256, 512, 1024, 4096};  // powers of two
    for (uint64_t mb : sets_mb) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            float ms;
            (void)hipEventRecord(a);
            hipLaunchKernelGGL(stream_rnd, dim3(256 * 8), dim3(256), 0, 0, buf, (mb << 20) / 512, chunks, sink);
            (void)hipEventRecord(b);
            (void)hipEventSynchronize(b);
            (void)hipEventElapsedTime(&ms, a, b);
            best = ms < best ? ms : best;
        }
        printf("working set %5llu MB: %7.3f ms  %6.2f TB/s\n", (unsigned long long)mb, best,
               (double)chunks * 512 / best / 1e9);
    }
    return 0;
}
