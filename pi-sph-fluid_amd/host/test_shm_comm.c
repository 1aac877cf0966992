/* test_shm_comm.c — the shared-memory transport of the multi-GPU host (sph_shm.h) driven by N THREADS of one process, so
 * that ThreadSanitizer sees every access to the segment (between processes it sees nothing): the protocol of a step of
 * slab_sph_fluid.c --transport host — reduce a word over the ranks, exchange halo messages with both neighbours, now and
 * then a larger collective (statistics, re-balancing histogram) — for a number of steps, checking what arrives.
 *     test_shm_comm_tsan <ranks> <steps> [<failing rank> <at step>]      (make -C pi-sph-fluid_amd host-tsan; tests/test_sanitizers.py)
 * With a failing rank: that rank's `out` callback fails in that step (a halo buffer over its capacity) — every rank must come
 * back from that exchange with an error, none may be left waiting at the barrier (round-4 advisor finding), and every later
 * collective fails too.
 * Test infrastructure: built and run by the tests only. */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdlib.h>

#include "sph_shm.h"

enum { HALO_WORDS = 4096, COLL_WORDS = 512 };

typedef struct rank_arg {
    shm_comm sc;
    int steps, failures;
    int fail_rank, fail_step, saw_failure;      /* injected failure: where, and whether this rank's exchange of that step returned an error */
    uint32_t send[2][HALO_WORDS], recv[2][HALO_WORDS];
    int step;
} rank_arg;

/* message of rank r to its neighbour on `side` in step t: header {count, tag} then a pattern */
static void fill(uint32_t *buf, int r, int side, int t) {
    const uint32_t count = (uint32_t)(64 + (r * 37 + side * 11 + t * 5) % (HALO_WORDS - 66));
    buf[0] = count;
    buf[1] = (uint32_t)t;
    for (uint32_t k = 0; k < count; k++) buf[2 + k] = (uint32_t)r * 0x01000193u + (uint32_t)side * 0x9e3779b9u + (uint32_t)t * 2654435761u + k;
}
static int check(const uint32_t *buf, int r, int side, int t) {
    const uint32_t count = (uint32_t)(64 + (r * 37 + side * 11 + t * 5) % (HALO_WORDS - 66));
    if (buf[0] != count || buf[1] != (uint32_t)t) return 1;
    for (uint32_t k = 0; k < count; k++)
        if (buf[2 + k] != (uint32_t)r * 0x01000193u + (uint32_t)side * 0x9e3779b9u + (uint32_t)t * 2654435761u + k) return 1;
    return 0;
}
static int out_cb(void *user, int side, void *dst) {
    rank_arg *a = (rank_arg *)user;
    if (a->sc.rank == a->fail_rank && a->step == a->fail_step) return 1;      /* (what SPH_E_CAPACITY from sph_slab_copy_out looks like here) */
    memcpy(dst, a->send[side], sizeof(uint32_t) * (2 + a->send[side][0]));      /* (as sph_slab_copy_out: header + used records) */
    return 0;
}
static int in_cb(void *user, int side, const void *src) {
    rank_arg *a = (rank_arg *)user;
    const uint32_t count = ((const uint32_t *)src)[0];
    if (count > HALO_WORDS - 2) return 1;
    memcpy(a->recv[side], src, sizeof(uint32_t) * (2 + count));
    return 0;
}

static void *rank_main(void *p) {
    rank_arg *a = (rank_arg *)p;
    shm_comm *sc = &a->sc;
    const int r = sc->rank, n = sc->nranks;
    for (int t = 0; t < a->steps; t++) {
        a->step = t;
        /* the rebuild word: MAX over the ranks (one rank raises it in some steps) */
        float w = (t % 7 == r % 7) ? 1.0f : 0.0f;
        if (shm_allreduce(sc, &w, 1, 1)) { a->failures++; break; }
        int expect = 0;
        for (int q = 0; q < n; q++) expect |= t % 7 == q % 7;
        if ((w > 0.0f) != (expect != 0)) a->failures++;
        /* the halo exchange */
        fill(a->send[0], r, 0, t);
        fill(a->send[1], r, 1, t);
        if (shm_exchange(sc, r > 0, r + 1 < n, out_cb, NULL, in_cb, a)) {
            if (a->fail_rank >= 0 && t == a->fail_step) {      /* the injected failure: seen by this rank too, and it sticks */
                float again = 0.0f;
                a->saw_failure = shm_allreduce(sc, &again, 1, 1) != 0 && shm_barrier(sc) != 0;
            } else {
                a->failures++;
            }
            break;
        }
        if (r > 0 && check(a->recv[0], r - 1, 1, t)) a->failures++;          /* what the left neighbour sent right */
        if (r + 1 < n && check(a->recv[1], r + 1, 0, t)) a->failures++;
        /* every tenth step a larger collective (column histogram of the re-balancing, statistics) */
        if (t % 10 == 0) {
            long long h[COLL_WORDS];
            for (int k = 0; k < COLL_WORDS; k++) h[k] = (long long)(r + 1) * (k + t);
            if (shm_allreduce(sc, h, COLL_WORDS, 0)) { a->failures++; break; }
            const long long tri = (long long)n * (n + 1) / 2;
            for (int k = 0; k < COLL_WORDS; k++) if (h[k] != tri * (k + t)) { a->failures++; break; }
            if (shm_barrier(sc)) { a->failures++; break; }
        }
    }
    return NULL;
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 4, steps = argc > 2 ? atoi(argv[2]) : 200;
    const int fail_rank = argc > 4 ? atoi(argv[3]) : -1, fail_step = argc > 4 ? atoi(argv[4]) : -1;
    if (n < 1 || n > 64 || steps < 1) { fprintf(stderr, "usage: %s <ranks 1..64> <steps>\n", argv[0]); return 2; }
    char name[128];
    snprintf(name, sizeof name, "/sph_shm_test_%d", (int)getpid());
    if (shm_create(name, n, sizeof(uint32_t) * HALO_WORDS, sizeof(long long) * COLL_WORDS)) return 1;
    rank_arg *args = (rank_arg *)calloc((size_t)n, sizeof(rank_arg));
    pthread_t *th = (pthread_t *)calloc((size_t)n, sizeof(pthread_t));
    if (!args || !th) return 1;
    int rc = 0;
    for (int r = 0; r < n; r++) {          /* ONE mapping shared by all "ranks": ThreadSanitizer tracks virtual addresses, and */
        args[r].sc.rank = r;               /* accesses through different mappings of the same memory would look unrelated to it */
        args[r].sc.nranks = n;
        args[r].steps = steps;
        args[r].fail_rank = fail_rank;
        args[r].fail_step = fail_step;
        if (r == 0) { if (shm_attach(&args[0].sc, name)) { rc = 1; break; } }
        else { args[r].sc.shm = args[0].sc.shm; snprintf(args[r].sc.shm_name, sizeof args[r].sc.shm_name, "%s", name); }
    }
    if (!rc) {
        for (int r = 0; r < n; r++) pthread_create(&th[r], NULL, rank_main, &args[r]);
        for (int r = 0; r < n; r++) pthread_join(th[r], NULL);
        for (int r = 0; r < n; r++) rc |= args[r].failures != 0;
        if (fail_rank >= 0) for (int r = 0; r < n; r++) rc |= !args[r].saw_failure;      /* (they all joined: nobody hung) */
    }
    shm_unlink(name);
    if (rc) { fprintf(stderr, "shared-memory transport: wrong data or a failed call\n"); return 1; }
    if (fail_rank >= 0) printf("ok: %d ranks, the failure of rank %d in step %d reached every rank\n", n, fail_rank, fail_step);
    else printf("ok: %d ranks, %d steps\n", n, steps);
    free(args);
    free(th);
    return 0;
}
