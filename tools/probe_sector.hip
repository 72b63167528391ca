// probe_sector.hip -- how many bytes does ONE random 8-byte table lookup pull out of HBM on
// gfx950, and does a cache-policy modifier on the load change it?  (The placement kernel does
// 141 such lookups per read; at 128 B each they are a third of its traffic.)
// Each variant is its own kernel name, so `rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum ...` reports
// them separately.  Build: hipcc -O2 --offload-arch=gfx950 -o tools/scratch/probe_sector ...
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                            \
    do {                                                                 \
        hipError_t e = (x);                                              \
        if (e != hipSuccess) {                                           \
            std::printf("%s -> %s\n", #x, hipGetErrorString(e));         \
            std::exit(1);                                                \
        }                                                                \
    } while (0)

typedef unsigned int v2u __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

template <int kVariant>
__device__ __forceinline__ v2u load8(const uint8_t *base, uint32_t byte_off)
{
    v2u out;
    const uint64_t b = (uint64_t)base;
    if (kVariant == 0) asm volatile("global_load_dwordx2 %0, %1, %2\n\ts_waitcnt vmcnt(0)" : "=&v"(out) : "v"(byte_off), "s"(b) : "memory");
    if (kVariant == 1) asm volatile("global_load_dwordx2 %0, %1, %2 sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(out) : "v"(byte_off), "s"(b) : "memory");
    if (kVariant == 2) asm volatile("global_load_dwordx2 %0, %1, %2 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(out) : "v"(byte_off), "s"(b) : "memory");
    if (kVariant == 3) asm volatile("global_load_dwordx2 %0, %1, %2 sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(out) : "v"(byte_off), "s"(b) : "memory");
    if (kVariant == 4) asm volatile("global_load_dwordx2 %0, %1, %2 nt\n\ts_waitcnt vmcnt(0)" : "=&v"(out) : "v"(byte_off), "s"(b) : "memory");
    if (kVariant == 5) asm volatile("global_load_dwordx2 %0, %1, %2 sc0 sc1 nt\n\ts_waitcnt vmcnt(0)" : "=&v"(out) : "v"(byte_off), "s"(b) : "memory");
    return out;
}

// every lane: kIters dependent-free random 8-byte loads over a 2 GiB table
template <int kVariant>
__global__ void __launch_bounds__(256) lookup_kernel(const uint8_t *table, uint32_t entries_mask, int iters,
                                                      unsigned long long *sink)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (int i = 0; i < iters; i += 4) {
        v2u r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t idx = mix(tid * 977u + (uint32_t)(i + j) * 0x9e3779b9u) & entries_mask;
            r[j] = load8<kVariant>(table, idx * 8u);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc += r[j].x ^ r[j].y;
    }
    if (acc == 0x12345678u) atomicAdd(sink, 1ull);
}

// scalar-cache path: one wave-uniform random 8-byte s_load per iteration
__global__ void __launch_bounds__(256) lookup_scalar_kernel(const uint8_t *table, uint32_t entries_mask, int iters,
                                                             unsigned long long *sink)
{
    const uint32_t wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        const uint32_t idx = __builtin_amdgcn_readfirstlane(mix(wave * 977u + (uint32_t)i * 0x9e3779b9u) & entries_mask);
        const uint64_t a = (uint64_t)table + (uint64_t)idx * 8u;
        unsigned long long v;
        asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(a) : "memory");
        acc += (uint32_t)v ^ (uint32_t)(v >> 32);
    }
    if (acc == 0x12345678u && threadIdx.x == 0) atomicAdd(sink, 1ull);
}

template <int kVariant>
static void run(const uint8_t *table, uint32_t mask, unsigned long long *sink, const char *name)
{
    const int blocks = 256 * 8, iters = 256;
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(lookup_kernel<kVariant>, dim3(blocks), dim3(256), 0, 0, table, mask, iters, sink);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
    }
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    const double lookups = (double)blocks * 256 * iters;
    std::printf("%-12s %8.3f ms  %7.2f G lookups/s  (x128 B = %6.2f TB/s, x32 B = %6.2f TB/s)\n", name, ms,
                lookups / ms / 1e6, lookups * 128 / ms / 1e9, lookups * 32 / ms / 1e9);
}

int main()
{
    const size_t bytes = 2ull << 30;
    uint8_t *table = nullptr;
    unsigned long long *sink = nullptr;
    CK(hipMalloc(reinterpret_cast<void **>(&table), bytes));
    CK(hipMemset(table, 1, bytes));
    CK(hipMalloc(reinterpret_cast<void **>(&sink), 8));
    CK(hipMemset(sink, 0, 8));
    const uint32_t mask = (uint32_t)(bytes / 8 - 1);
    run<0>(table, mask, sink, "plain");
    run<1>(table, mask, sink, "sc0");
    run<2>(table, mask, sink, "sc1");
    run<3>(table, mask, sink, "sc0 sc1");
    run<4>(table, mask, sink, "nt");
    run<5>(table, mask, sink, "sc0 sc1 nt");
    {
        hipEvent_t a, b;
        CK(hipEventCreate(&a));
        CK(hipEventCreate(&b));
        const int blocks = 256 * 8, iters = 256;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(lookup_scalar_kernel, dim3(blocks), dim3(256), 0, 0, table, mask, iters, sink);
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
        }
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        const double lookups = (double)blocks * 4 * iters;
        std::printf("%-12s %8.3f ms  %7.2f G lookups/s\n", "s_load", ms, lookups / ms / 1e6);
    }
    CK(hipDeviceSynchronize());
    return 0;
}
