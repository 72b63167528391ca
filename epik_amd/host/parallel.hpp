// parallel.hpp -- items handed out to a few threads, one at a time (the host stages of the driver: the
// batches of a group are de-duplicated, joined and formatted side by side).
#ifndef EPIK_AMD_HOST_PARALLEL_HPP
#define EPIK_AMD_HOST_PARALLEL_HPP

#include <atomic>
#include <cstddef>
#include <exception>
#include <mutex>
#include <thread>
#include <vector>

namespace epik_amd {

/// fn(i) for every i in [0, n), on up to `num_threads` threads (the calling one among them); the first
/// exception thrown by any of them is thrown again here, after all have stopped.
template <typename F>
void parallel_for(size_t n, size_t num_threads, F&& fn)
{
    if (n == 0) return;
    const size_t workers = num_threads < 2 ? 1 : (num_threads < n ? num_threads : n);
    if (workers == 1) {
        for (size_t i = 0; i < n; ++i) fn(i);
        return;
    }
    std::atomic<size_t> next{0};
    std::exception_ptr error;
    std::mutex error_mutex;
    auto work = [&] {
        try {
            for (size_t i = next.fetch_add(1); i < n; i = next.fetch_add(1)) fn(i);
        } catch (...) {
            std::lock_guard<std::mutex> lock(error_mutex);
            if (!error) error = std::current_exception();
            next.store(n);  // the others stop at their next item
        }
    };
    std::vector<std::thread> threads;
    threads.reserve(workers - 1);
    for (size_t t = 1; t < workers; ++t) threads.emplace_back(work);
    work();
    for (auto& t : threads) t.join();
    if (error) std::rethrow_exception(error);
}

}  // namespace epik_amd
#endif
