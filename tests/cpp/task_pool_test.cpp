// TaskPool (bevy_amd/host/bevy_mi355x_host.hpp): every chunk of every job runs exactly once, whatever the job sizes and however the
// workers straggle; jobs follow each other without a pause.  Host code only (no device): part of the CPU suite.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../bevy_amd/host/bevy_mi355x_host.hpp"

int main(int argc, char** argv) {
    using namespace bevy_mi355x;
    const int jobs = argc > 1 ? std::atoi(argv[1]) : 20000;
    int failed = 0;
    for (unsigned workers : {0u, 1u, 3u, 7u}) {
        TaskPool pool(workers);
        uint64_t rng = 0x9E3779B97F4A7C15ull + workers;
        auto next = [&rng]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
        for (int job = 0; job < jobs; ++job) {
            const uint32_t n = (uint32_t)(next() % 40);  // (0 and 1 run inline)
            std::vector<std::atomic<uint32_t>> hits(n);
            for (auto& h : hits) h.store(0);
            std::atomic<uint64_t> sum{0};
            pool.for_each_chunk(n, [&](uint32_t c) {
                hits[c].fetch_add(1);
                if ((c + job) % 7 == 0) for (volatile int spin = 0; spin < 200; ++spin) {}  // a straggler now and then
                sum.fetch_add(c + 1);
            });
            bool ok = sum.load() == (uint64_t)n * (n + 1) / 2;
            for (auto& h : hits) ok = ok && h.load() == 1;
            if (!ok) {
                ++failed;
                std::printf("FAILED: %u workers, job %d, %u chunks\n", workers, job, n);
                break;
            }
        }
    }
    std::printf("%s\n", failed ? "task pool: FAILED" : "task pool: ok");
    return failed ? 1 : 0;
}
