// pcie_probe.hip -- what the host-buffer entry point (epik_amd_placer_place) can expect from
// the host link: pageable vs pinned copies, the cost of hipHostRegister on caller memory,
// and of a CPU memcpy into a pinned bounce buffer.  Build: hipcc -O2 --offload-arch=gfx950.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e = (x);                                                           \
        if (e != hipSuccess) {                                                        \
            std::printf("%s -> %s\n", #x, hipGetErrorString(e));                      \
            std::exit(1);                                                             \
        }                                                                             \
    } while (0)

static double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main()
{
    const size_t bytes = 160u << 20;
    char *pageable = static_cast<char *>(std::aligned_alloc(4096, bytes));
    std::memset(pageable, 1, bytes);
    char *pinned = nullptr;
    CK(hipHostMalloc(reinterpret_cast<void **>(&pinned), bytes, hipHostMallocDefault));
    std::memset(pinned, 2, bytes);
    char *dev = nullptr;
    CK(hipMalloc(reinterpret_cast<void **>(&dev), bytes));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    for (int rep = 0; rep < 3; ++rep) {
        double t = now();
        CK(hipMemcpyAsync(dev, pageable, bytes, hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
        std::printf("H2D pageable  %6.2f GB/s\n", bytes / (now() - t) / 1e9);
        t = now();
        CK(hipMemcpyAsync(pageable, dev, bytes, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        std::printf("D2H pageable  %6.2f GB/s\n", bytes / (now() - t) / 1e9);
        t = now();
        CK(hipMemcpyAsync(dev, pinned, bytes, hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
        std::printf("H2D pinned    %6.2f GB/s\n", bytes / (now() - t) / 1e9);
        t = now();
        CK(hipMemcpyAsync(pinned, dev, bytes, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        std::printf("D2H pinned    %6.2f GB/s\n", bytes / (now() - t) / 1e9);
        t = now();
        CK(hipHostRegister(pageable, bytes, hipHostRegisterDefault));
        const double t_reg = now() - t;
        t = now();
        CK(hipMemcpyAsync(dev, pageable, bytes, hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
        const double t_cp = now() - t;
        t = now();
        CK(hipHostUnregister(pageable));
        std::printf("register %6.2f ms (%5.2f GB/s), copy registered %6.2f GB/s, unregister %6.2f ms\n",
                    t_reg * 1e3, bytes / t_reg / 1e9, bytes / t_cp / 1e9, (now() - t) * 1e3);
        for (int threads : {1, 2, 4, 8}) {
            t = now();
            std::vector<std::thread> pool;
            const size_t part = bytes / threads;
            for (int i = 0; i < threads; ++i)
                pool.emplace_back([=] { std::memcpy(pinned + i * part, pageable + i * part, part); });
            for (auto &th : pool) th.join();
            std::printf("CPU memcpy pageable->pinned, %d thread(s): %6.2f GB/s\n", threads,
                        bytes / (now() - t) / 1e9);
        }
    }
    return 0;
}
