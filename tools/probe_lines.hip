// probe_lines.hip -- how many random 128-byte lines per second does MI355X deliver when every LANE asks for 8 bytes
// of a line of its own (the shape of a table or presence-filter lookup: 64 unrelated lines per wave instruction)?
// The chunk probe beside it (probe_mall.hip) reads four consecutive lines per wave; a protein placement is mostly
// lookups, and this is the ceiling they meet.  Lines per run are the same for every working set.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/scratch/probe_lines tools/probe_lines.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int kInFlight>
__global__ void __launch_bounds__(256) lookup_rnd(const uint2 *buf, uint64_t lines_in_set, uint64_t per_thread, uint2 *sink)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint2 acc = make_uint2(0, 0);
    for (uint64_t i = 0; i < per_thread; i += kInFlight) {
#pragma unroll
        for (int u = 0; u < kInFlight; ++u) {  // independent lookups of one lane, all on their way together
            uint64_t h = (tid * per_thread + i + u) * 0x9E3779B97F4A7C15ull;
            h ^= h >> 29;
            h *= 0xBF58476D1CE4E5B9ull;
            h ^= h >> 32;
            const uint64_t line = h & (lines_in_set - 1);  // a power of two
            const uint2 v = buf[line * 16 + ((h >> 40) & 15)];  // 8 of the line's 128 bytes
            acc.x ^= v.x;
            acc.y += v.y;
        }
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = acc;
}

int main()
{
    const uint64_t max_bytes = 16ull << 30;
    uint2 *buf, *sink;
    if (hipMalloc(&buf, max_bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
    (void)hipMemset(buf, 1, max_bytes);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    const uint64_t sets_mb[] = {64, 256, 512, 1024, 4096, 16384};
    const int blocks = 256 * 8, threads = 256;  // eight workgroups of four waves per CU
    const uint64_t per_thread = 256;            // 2^19 threads x 256 = 1.3e8 lines per run
    for (int in_flight : {1, 4, 8}) {
        for (uint64_t mb : sets_mb) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                float ms;
                (void)hipEventRecord(a);
                const uint64_t lines = (mb << 20) / 128;
                if (in_flight == 1)
                    hipLaunchKernelGGL(lookup_rnd<1>, dim3(blocks), dim3(threads), 0, 0, buf, lines, per_thread, sink);
                else if (in_flight == 4)
                    hipLaunchKernelGGL(lookup_rnd<4>, dim3(blocks), dim3(threads), 0, 0, buf, lines, per_thread, sink);
                else
                    hipLaunchKernelGGL(lookup_rnd<8>, dim3(blocks), dim3(threads), 0, 0, buf, lines, per_thread, sink);
                (void)hipEventRecord(b);
                (void)hipEventSynchronize(b);
                (void)hipEventElapsedTime(&ms, a, b);
                best = ms < best ? ms : best;
            }
            const double n = (double)blocks * threads * per_thread;
            printf("%d lookups in flight per lane, working set %6llu MB: %8.3f ms  %6.2f G lines/s  = %5.2f TB/s of 128-byte lines\n",
                   in_flight, (unsigned long long)mb, best, n / best / 1e6, n * 128 / best / 1e9);
        }
    }
    return 0;
}
