/* ecs_gather.c -- the ECS side's gather loop of the end-to-end frames (benchlib/end_to_end.py): the rows a Changed<Transform> query
 * yields, copied from the component columns straight into the library's pinned upload window.  In a Bevy app this is a par_iter over
 * the tables (the Rust shim's upload_and_propagate; the C++ host layer's chunked loop in bevy_mi355x_host.hpp); the Python harness
 * has no such loop of its own -- one thread of numpy's take() was 1.9 of the 2.7 ms of a 10 %-dirty frame, a harness artefact the
 * line then carried as if it were the product's.  HARNESS code: not part of the library, not the oracle.  gcc -O2 -pthread.
 * A small persistent pool (threads started at the first call, parked on a condition variable between calls: starting eight
 * threads per call was 200 us of a 235 us gather of 11 000 rows; an OpenMP team spun for 25 ms per call in a CPU-throttled
 * container). */
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <string.h>

typedef struct {
    uint32_t a, b, lo;
    const uint32_t* rows;
    const float *t3, *r4, *s3;
    uint32_t* out_rows;
    float *out_t, *out_r, *out_s;
} job_t;

static void* gather_job(void* p) {
    const job_t* j = (const job_t*)p;
    for (uint32_t i = j->a; i < j->b; ++i) {
        const uint32_t row = j->rows[i];
        j->out_rows[i] = row;
        memcpy(j->out_t + 3 * (size_t)i, j->t3 + 3 * (size_t)row, 12);
        memcpy(j->out_r + 4 * (size_t)i, j->r4 + 4 * (size_t)row, 16);
        memcpy(j->out_s + 3 * (size_t)i, j->s3 + 3 * (size_t)row, 12);
    }
    return 0;
}
static void* copy_job(void* p) {
    const job_t* j = (const job_t*)p;
    const size_t a = j->a, n = (size_t)j->b - j->a, src = (size_t)j->lo + a;
    if (j->out_t) memcpy(j->out_t + 3 * a, j->t3 + 3 * src, n * 12);
    if (j->out_r) memcpy(j->out_r + 4 * a, j->r4 + 4 * src, n * 16);
    if (j->out_s) memcpy(j->out_s + 3 * a, j->s3 + 3 * src, n * 12);
    return 0;
}
#define MAX_THREADS 64
static struct {
    pthread_mutex_t mu;
    pthread_cond_t wake, done_cv;
    int started;              /* workers alive (they take job slots 1..started) */
    uint64_t generation;      /* bumped per call */
    int active;               /* job slots of this call (slot 0 is the caller's) */
    int pending;              /* workers of this call still running */
    void* (*fn)(void*);
    job_t jobs[MAX_THREADS];
} pool = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, 0, 0, 0, 0, 0, {{0}}};

static void* worker(void* arg) {
    const int slot = (int)(intptr_t)arg;
    uint64_t seen = 0;
    pthread_mutex_lock(&pool.mu);
    for (;;) {
        while (pool.generation == seen) pthread_cond_wait(&pool.wake, &pool.mu);
        seen = pool.generation;
        if (slot >= pool.active) continue;  /* this call uses fewer threads */
        void* (*fn)(void*) = pool.fn;
        pthread_mutex_unlock(&pool.mu);
        fn(&pool.jobs[slot]);
        pthread_mutex_lock(&pool.mu);
        if (--pool.pending == 0) pthread_cond_signal(&pool.done_cv);
    }
    return 0;
}
static void run(void* (*fn)(void*), job_t* proto, uint32_t k, int threads) {
    if (threads < 1) threads = 1;
    if (threads > MAX_THREADS) threads = MAX_THREADS;
    if ((uint32_t)threads > k / 8192u) threads = (int)(k / 8192u);  /* (a wake-up is ~20 us: at least 8 192 rows per thread) */
    if (threads < 1) threads = 1;
    job_t mine = *proto;
    if (threads == 1) {
        mine.a = 0;
        mine.b = k;
        fn(&mine);
        return;
    }
    pthread_mutex_lock(&pool.mu);
    while (pool.started < threads - 1) {
        pthread_t th;
        const int slot = pool.started + 1;
        if (pthread_create(&th, 0, worker, (void*)(intptr_t)slot) != 0) break;
        pthread_detach(th);
        pool.started = slot;
    }
    if (threads > pool.started + 1) threads = pool.started + 1;
    for (int c = 0; c < threads; ++c) {
        pool.jobs[c] = *proto;
        pool.jobs[c].a = (uint32_t)((uint64_t)k * (uint64_t)c / (uint64_t)threads);
        pool.jobs[c].b = (uint32_t)((uint64_t)k * (uint64_t)(c + 1) / (uint64_t)threads);
    }
    mine = pool.jobs[0];
    pool.fn = fn;
    pool.active = threads;
    pool.pending = threads - 1;
    pool.generation++;
    pthread_cond_broadcast(&pool.wake);
    pthread_mutex_unlock(&pool.mu);
    fn(&mine);
    pthread_mutex_lock(&pool.mu);
    while (pool.pending) pthread_cond_wait(&pool.done_cv, &pool.mu);
    pthread_mutex_unlock(&pool.mu);
}

void ecs_gather_rows(uint32_t k, const uint32_t* rows, const float* t3, const float* r4, const float* s3, uint32_t* out_rows,
                     float* out_t, float* out_r, float* out_s, int threads) {
    job_t j = {0, 0, 0, rows, t3, r4, s3, out_rows, out_t, out_r, out_s};
    run(gather_job, &j, k, threads);
}

/* dense: rows [lo, lo + m) of whichever components the window carries (NULL = not carried) */
void ecs_copy_rows(uint32_t lo, uint32_t m, const float* t3, const float* r4, const float* s3, float* out_t, float* out_r, float* out_s,
                   int threads) {
    job_t j = {0, 0, lo, 0, t3, r4, s3, 0, out_t, out_r, out_s};
    run(copy_job, &j, m, threads);
}
