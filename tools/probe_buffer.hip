// probe_buffer.hip -- can the posting stream use range-checked buffer loads on gfx950?
//   (a) raw V# (stride 0), num_records = 6*cnt bytes, lane l reads a dword at 6*l and a ushort at
//       6*l+4 (6-byte {f32 score, u16 cell} postings, dwords only 2-byte aligned): are the values
//       right, do lanes past the end read 0, and how fast is it next to
//   (b) the current SoA form (f32 score[cnt] then u16 cell[cnt], clamped global loads)?
// Build: hipcc -O2 --offload-arch=gfx950 -o tools/scratch/probe_buffer tools/probe_buffer.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                            \
    do {                                                                 \
        hipError_t e = (x);                                              \
        if (e != hipSuccess) {                                           \
            std::printf("%s -> %s\n", #x, hipGetErrorString(e));         \
            std::exit(1);                                                \
        }                                                                \
    } while (0)

constexpr uint32_t kLists = 1u << 20;   // 512 MiB of lists, 512 B apart
constexpr uint32_t kStride = 512;

__host__ __device__ inline uint32_t list_len(uint32_t list) { return 1u + (list * 2654435761u >> 26); }  // 1..64
__host__ __device__ inline uint32_t mix(uint32_t x)
{
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

// out[0] = checksum over in-range lanes, out[1] = number of out-of-range lanes that did NOT read 0
// kVariant: 0 = SoA, clamped global loads (what the kernel did before)
//           1 = AoS6, one raw V# of 6*cnt bytes, dword at 6*l (2-byte aligned) + ushort at 6*l+4
//           2 = SoA, two raw V#s (4*cnt bytes of scores, 2*cnt bytes of cells)
//           3 = SoA, one raw V# of 6*cnt bytes, the cell load with soffset = 4*cnt
//               (tells whether soffset takes part in the range check)
// kUnroll independent chunks are in flight per wave and trip.
template <int kVariant, int kUnroll>
__global__ void __launch_bounds__(256) stream_kernel(const uint8_t *db, int iters, unsigned long long *out)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    unsigned long long sum = 0, bad = 0;
    for (int i = 0; i < iters; i += kUnroll) {
        uint32_t score[kUnroll], cell[kUnroll], cnt[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const uint32_t list = __builtin_amdgcn_readfirstlane(mix(wave * 8191u + (uint32_t)(i + u)) & (kLists - 1u));
            cnt[u] = list_len(list);
            const uint8_t *base = db + (size_t)list * kStride + (kVariant == 1 ? 2u * (list & 1u) : 0u);
            if (kVariant == 0) {
                const uint32_t l = lane < cnt[u] ? lane : cnt[u] - 1u;
                score[u] = *reinterpret_cast<const uint32_t *>(base + 4u * l);
                cell[u] = *reinterpret_cast<const uint16_t *>(base + 4u * cnt[u] + 2u * l);
            } else if (kVariant == 1) {
                const __amdgpu_buffer_rsrc_t rsrc =
                    __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base), 0, (int)(6u * cnt[u]), 0x00020000);
                score[u] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(6u * lane), 0, 0);
                cell[u] = (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rsrc, (int)(6u * lane + 4u), 0, 0);
            } else if (kVariant == 2) {
                const __amdgpu_buffer_rsrc_t rs =
                    __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base), 0, (int)(4u * cnt[u]), 0x00020000);
                const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<uint8_t *>(base + 4u * cnt[u]), 0, (int)(2u * cnt[u]), 0x00020000);
                score[u] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(4u * lane), 0, 0);
                cell[u] = (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rc, (int)(2u * lane), 0, 0);
            } else {
                const __amdgpu_buffer_rsrc_t rsrc =
                    __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base), 0, (int)(6u * cnt[u]), 0x00020000);
                score[u] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(4u * lane), 0, 0);
                cell[u] = (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rsrc, (int)(2u * lane),
                                                                                   (int)(4u * cnt[u]), 0);
            }
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            if (kVariant != 0 && lane >= cnt[u] && cell[u] != 0) ++bad;  // a lane past the end must see cell 0
            if (lane < cnt[u]) sum += (unsigned long long)(score[u] ^ (cell[u] << 7));
        }
    }
    atomicAdd(&out[0], sum);
    atomicAdd(&out[1], bad);
}

template <int kVariant>
static float run_variant(const uint8_t *d, unsigned long long *d_out, int blocks, int iters, unsigned long long *res)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(d_out, 0, 16));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((stream_kernel<kVariant, 8>), dim3(blocks), dim3(256), 0, 0, d, iters, d_out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    CK(hipMemcpy(res, d_out, 16, hipMemcpyDeviceToHost));
    return best;
}

int main()
{
    const size_t bytes = (size_t)kLists * kStride;
    std::vector<uint8_t> aos(bytes + 512, 0xEE), soa(bytes + 512, 0xEE);   // 0xEE: what a missed range check would read
    for (uint32_t list = 0; list < kLists; ++list) {
        const uint32_t cnt = list_len(list);
        uint8_t *a = aos.data() + (size_t)list * kStride + 2u * (list & 1u), *s = soa.data() + (size_t)list * kStride;
        for (uint32_t j = 0; j < cnt; ++j) {
            const uint32_t score = mix(list * 64u + j);
            const uint16_t cell = (uint16_t)(1u + (mix(list + 77u * j) % 999u));
            std::memcpy(a + 6u * j, &score, 4);
            std::memcpy(a + 6u * j + 4u, &cell, 2);
            std::memcpy(s + 4u * j, &score, 4);
            std::memcpy(s + 4u * cnt + 2u * j, &cell, 2);
        }
    }
    uint8_t *d_aos = nullptr, *d_soa = nullptr;
    unsigned long long *d_out = nullptr;
    CK(hipMalloc(reinterpret_cast<void **>(&d_aos), aos.size()));
    CK(hipMalloc(reinterpret_cast<void **>(&d_soa), soa.size()));
    CK(hipMalloc(reinterpret_cast<void **>(&d_out), 16));
    CK(hipMemcpy(d_aos, aos.data(), aos.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_soa, soa.data(), soa.size(), hipMemcpyHostToDevice));
    const int blocks = 256 * 5, iters = 512;
    unsigned long long res[4][2];
    const char *names[4] = {"SoA, clamped global loads", "AoS6, one raw V# (2-byte aligned dwords)",
                            "SoA, two raw V#s", "SoA, one raw V# + soffset"};
    float ms[4];
    ms[0] = run_variant<0>(d_soa, d_out, blocks, iters, res[0]);
    ms[1] = run_variant<1>(d_aos, d_out, blocks, iters, res[1]);
    ms[2] = run_variant<2>(d_soa, d_out, blocks, iters, res[2]);
    ms[3] = run_variant<3>(d_soa, d_out, blocks, iters, res[3]);
    for (int v = 0; v < 4; ++v)
        std::printf("%-42s %7.3f ms  %6.2f G chunks/s  checksum %016llx  lanes past the end with cell != 0: %llu\n",
                    names[v], ms[v], (double)blocks * 4 * iters / ms[v] / 1e6, res[v][0], res[v][1]);
    // CPU checksum of the same walk
    unsigned long long ref = 0;
    for (uint32_t wave = 0; wave < (uint32_t)blocks * 4u; ++wave)
        for (int i = 0; i < iters; ++i) {
            const uint32_t list = mix(wave * 8191u + (uint32_t)i) & (kLists - 1u);
            const uint32_t cnt = list_len(list);
            for (uint32_t j = 0; j < cnt; ++j) {
                const uint32_t score = mix(list * 64u + j);
                const uint32_t cell = 1u + (mix(list + 77u * j) % 999u);
                ref += (unsigned long long)(score ^ (cell << 7));
            }
        }
    std::printf("CPU checksum %016llx\n", ref);
    return 0;
}
