/* ecs_gather.c -- the ECS side's gather loop of the end-to-end frames (benchlib/end_to_end.py): the rows a Changed<Transform> query
 * yields, copied from the component columns straight into the library's pinned upload window.  In a Bevy app this is a par_iter over
 * the tables (the Rust shim's upload_and_propagate; the C++ host layer's chunked loop in bevy_mi355x_host.hpp); the Python harness
 * has no such loop of its own -- one thread of numpy's take() was 1.9 of the 2.7 ms of a 10 %-dirty frame, a harness artefact the
 * line then carried as if it were the product's.  HARNESS code: not part of the library, not the oracle.  gcc -O2 -fopenmp. */
#include <stdint.h>
#include <string.h>

void ecs_gather_rows(uint32_t k, const uint32_t* rows, const float* t3, const float* r4, const float* s3, uint32_t* out_rows,
                     float* out_t, float* out_r, float* out_s, int threads) {
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t i = 0; i < (int64_t)k; ++i) {
        const uint32_t row = rows[i];
        out_rows[i] = row;
        memcpy(out_t + 3 * i, t3 + 3 * (size_t)row, 12);
        memcpy(out_r + 4 * i, r4 + 4 * (size_t)row, 16);
        memcpy(out_s + 3 * i, s3 + 3 * (size_t)row, 12);
    }
}

/* dense: rows [lo, lo + m) of whichever components the window carries (NULL = not carried) */
void ecs_copy_rows(uint32_t lo, uint32_t m, const float* t3, const float* r4, const float* s3, float* out_t, float* out_r, float* out_s,
                   int threads) {
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t c = 0; c < (int64_t)threads; ++c) {
        const size_t a = (size_t)m * (size_t)c / (size_t)threads, b = (size_t)m * (size_t)(c + 1) / (size_t)threads;
        if (out_t) memcpy(out_t + 3 * a, t3 + 3 * (lo + a), (b - a) * 12);
        if (out_r) memcpy(out_r + 4 * a, r4 + 4 * (lo + a), (b - a) * 16);
        if (out_s) memcpy(out_s + 3 * a, s3 + 3 * (lo + a), (b - a) * 12);
    }
}
