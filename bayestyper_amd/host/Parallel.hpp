// Host-side worker threads (-p/--threads): the reference threads its host stages (graph construction VariantFileParser.cpp:1044-1106, genotype
// collection InferenceEngine.cpp:292-310, the writer GenotypeWriter.cpp:84-127); here the GPU does the sampling and the threads share out the
// per-cluster host work either side of it.  parallelFor cuts [0, n) into contiguous ranges, one per thread, so that a result assembled "range
// after range" has the order of a one-thread run.
#pragma once
#include <algorithm>
#include <exception>
#include <functional>
#include <thread>
#include <vector>

namespace bthost {

inline unsigned clampThreads(unsigned long requested) { return (unsigned)std::max<unsigned long>(1, std::min<unsigned long>(requested, 256)); }

// fn(begin, end, part) for `parts` contiguous ranges of [0, n); the first exception of a worker is rethrown
inline void parallelFor(size_t n, unsigned threads, const std::function<void(size_t, size_t, unsigned)> &fn) {
    const unsigned parts = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, n));
    if (parts <= 1) {
        fn(0, n, 0);
        return;
    }
    std::vector<std::thread> pool;
    std::vector<std::exception_ptr> err(parts);
    for (unsigned p = 0; p < parts; p++)
        pool.emplace_back([&, p]() {
            try {
                fn(n * p / parts, n * (p + 1) / parts, p);
            } catch (...) {
                err[p] = std::current_exception();
            }
        });
    for (auto &t : pool) t.join();
    for (auto &e : err)
        if (e) std::rethrow_exception(e);
}

}  // namespace bthost
