/* How many cores does this box really give a process?  N threads each do the same fixed amount of register-only work; a box with N free cores
 * takes the time of one thread.  usage: cpu_spin <threads> [iterations]   (prints seconds).  tools/cpu_scaling.py sweeps N. */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

static uint64_t iters = 400000000ull;
static volatile uint64_t sink;

static void *work(void *arg) {
    uint64_t x = (uint64_t)(uintptr_t)arg + 88172645463325252ull, acc = 0;
    for (uint64_t i = 0; i < iters; ++i) {   /* xorshift64: a dependent chain, no memory */
        x ^= x << 13;
        x ^= x >> 7;
        x ^= x << 17;
        acc += x;
    }
    sink = acc;
    return NULL;
}

int main(int argc, char **argv) {
    int n = argc > 1 ? atoi(argv[1]) : 1;
    if (argc > 2) iters = strtoull(argv[2], NULL, 10);
    pthread_t *t = malloc(sizeof(pthread_t) * (size_t)n);
    struct timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    for (int i = 0; i < n; ++i) pthread_create(&t[i], NULL, work, (void *)(uintptr_t)i);
    for (int i = 0; i < n; ++i) pthread_join(t[i], NULL);
    clock_gettime(CLOCK_MONOTONIC, &b);
    printf("%.4f\n", (b.tv_sec - a.tv_sec) + (b.tv_nsec - a.tv_nsec) * 1e-9);
    return 0;
}
