/* sph_shm.h — the shared-memory transport of the multi-GPU host (slab_sph_fluid.c, --transport host | peer): one POSIX
 * shared segment per job holding a process-shared barrier, two sets of per-rank collective slots and two sets of per-rank
 * halo mailboxes (double-buffered by the parity of a sequence number, so that a slot is written again only two collectives
 * later: a barrier lies in between — every collective, a plain barrier included, takes a number and waits once).  Plain C, no HIP: slab_sph_fluid.c includes it, and host/test_shm_comm.c drives the
 * same functions from N threads under ThreadSanitizer (make host-tsan; tests/test_sanitizers.py).
 *
 * No reference counterpart: pi_sph_fluid.c is a single process (its only sharing is the omp team of :610). */
#ifndef SPH_SHM_H
#define SPH_SHM_H

#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

typedef struct shm_hdr {
    pthread_barrier_t bar;
    unsigned long failed;         /* 0, or 1 + the sequence number of the collective in which a rank could not do its part (a callback
                                   * failed, a message too large).  That rank still ARRIVES at the barrier — the others would wait
                                   * there for ever — and every rank finds the word behind it: all of them return 1 from THAT
                                   * collective and from every later one (not from an earlier one a slow rank may still be in: it
                                   * would leave, and the others would wait for it in the failing one) */
    int nranks;
    size_t halo_bytes, coll_bytes, coll_off, mail_off, total;
} shm_hdr;

typedef struct shm_comm {
    int rank, nranks;
    shm_hdr *shm;
    char shm_name[160];
    unsigned long seq;            /* collectives so far (parity: which of the two slots / mailboxes) */
    unsigned long xseq;           /* all-to-all exchanges so far (names of their segments) */
    size_t coll_bytes, halo_bytes;
} shm_comm;

static size_t shm_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static size_t shm_layout(shm_hdr *h, int nranks, size_t halo_bytes, size_t coll_bytes) {
    h->nranks = nranks;
    h->failed = 0;
    h->halo_bytes = shm_align_up(halo_bytes, 64);
    h->coll_bytes = shm_align_up(coll_bytes, 64);
    h->coll_off = shm_align_up(sizeof(shm_hdr), 4096);
    h->mail_off = shm_align_up(h->coll_off + 2 * (size_t)nranks * h->coll_bytes, 4096);
    h->total = h->mail_off + 2 * (size_t)nranks * 2 * h->halo_bytes;
    return h->total;
}
static unsigned char *shm_coll(shm_comm *sc, unsigned long parity, int r) {
    return (unsigned char *)sc->shm + sc->shm->coll_off + ((size_t)(parity & 1ul) * (size_t)sc->nranks + (size_t)r) * sc->shm->coll_bytes;
}
static unsigned char *shm_mail(shm_comm *sc, unsigned long parity, int r, int side) {
    return (unsigned char *)sc->shm + sc->shm->mail_off + (((size_t)(parity & 1ul) * (size_t)sc->nranks + (size_t)r) * 2 + (size_t)side) * sc->shm->halo_bytes;
}

/* create (launcher, or the one in-process rank) or open (a rank) the shared segment */
static int shm_create(const char *name, int nranks, size_t halo_bytes, size_t coll_bytes) {
    shm_hdr tmp;
    const size_t total = shm_layout(&tmp, nranks, halo_bytes, coll_bytes);
    shm_unlink(name);
    const int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) { perror("shm_open"); return 1; }
    if (ftruncate(fd, (off_t)total) != 0) { perror("ftruncate"); close(fd); shm_unlink(name); return 1; }
    shm_hdr *h = (shm_hdr *)mmap(NULL, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (h == MAP_FAILED) { perror("mmap"); shm_unlink(name); return 1; }
    shm_layout(h, nranks, halo_bytes, coll_bytes);
    pthread_barrierattr_t at;
    pthread_barrierattr_init(&at);
    pthread_barrierattr_setpshared(&at, PTHREAD_PROCESS_SHARED);
    const int rc = pthread_barrier_init(&h->bar, &at, (unsigned)nranks);
    pthread_barrierattr_destroy(&at);
    munmap(h, total);
    if (rc != 0) { fprintf(stderr, "pthread_barrier_init: %s\n", strerror(rc)); shm_unlink(name); return 1; }
    return 0;
}
static int shm_attach(shm_comm *sc, const char *name) {
    const int fd = shm_open(name, O_RDWR, 0600);
    if (fd < 0) { fprintf(stderr, "[rank %d] shm_open(%s): %s\n", sc->rank, name, strerror(errno)); return 1; }
    struct stat sb;
    if (fstat(fd, &sb) != 0 || sb.st_size < (off_t)sizeof(shm_hdr)) { close(fd); return 1; }
    sc->shm = (shm_hdr *)mmap(NULL, (size_t)sb.st_size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (sc->shm == MAP_FAILED) { sc->shm = NULL; return 1; }
    if (sc->shm->nranks != sc->nranks || sc->shm->total != (size_t)sb.st_size) { fprintf(stderr, "[rank %d] shared segment %s does not match this run\n", sc->rank, name); return 1; }
    snprintf(sc->shm_name, sizeof sc->shm_name, "%s", name);
    return 0;
}

static void shm_fail(shm_comm *sc, unsigned long par) {
    unsigned long none = 0ul;
    if (sc->shm) (void)__atomic_compare_exchange_n(&sc->shm->failed, &none, par + 1ul, 0, __ATOMIC_RELEASE, __ATOMIC_RELAXED);
}
static int shm_failed(shm_comm *sc, unsigned long par) {
    const unsigned long f = sc->shm ? __atomic_load_n(&sc->shm->failed, __ATOMIC_ACQUIRE) : 0ul;
    return f != 0ul && f <= par + 1ul;
}
/* the barrier of collective number `par` (every collective — reduction, exchange, plain barrier — takes one number, the same on
 * every rank: they all make the same calls in the same order) */
static int shm_wait(shm_comm *sc, unsigned long par) {
    const int rc = pthread_barrier_wait(&sc->shm->bar);
    return (rc != 0 && rc != PTHREAD_BARRIER_SERIAL_THREAD) || shm_failed(sc, par);
}
static int shm_barrier(shm_comm *sc) {
    if (sc->nranks == 1) return 0;
    return shm_wait(sc, sc->seq++);
}

/* element-wise reduction of a small host array over all ranks, in place: op 0 = sum of int64, 1 = max of float */
static int shm_allreduce(shm_comm *sc, void *buf, size_t count, int op) {
    const size_t esz = op == 0 ? sizeof(long long) : sizeof(float), bytes = count * esz;
    if (sc->nranks == 1) return 0;
    const unsigned long par = sc->seq++;
    if (bytes > sc->shm->coll_bytes) {      /* (the others may have passed a size that fits: they are waiting at the barrier) */
        fprintf(stderr, "[rank %d] collective of %zu bytes exceeds the staging size %zu\n", sc->rank, bytes, sc->shm->coll_bytes);
        shm_fail(sc, par);
    } else {
        memcpy(shm_coll(sc, par, sc->rank), buf, bytes);
    }
    if (shm_wait(sc, par)) return 1;      /* (the slot of this parity is written again two collectives later: a barrier lies in between) */
    for (int r = 0; r < sc->nranks; r++) {
        if (r == sc->rank) continue;
        if (op == 0) {
            const long long *o = (const long long *)shm_coll(sc, par, r);
            for (size_t k = 0; k < count; k++) ((long long *)buf)[k] += o[k];
        } else {
            const float *o = (const float *)shm_coll(sc, par, r);
            for (size_t k = 0; k < count; k++) if (o[k] > ((float *)buf)[k]) ((float *)buf)[k] = o[k];
        }
    }
    return 0;
}

/* the halo exchange of a step through the mailboxes: `out(user, side, dst)` writes this rank's message for the neighbour on
 * `side` (0 = left, 1 = right) into its mailbox, `between(user)` is whatever may run while the others write theirs (the
 * interior density pass), `in(user, side, src)` reads what that neighbour sent towards this rank.  Callbacks return 0 on success.
 * A rank whose callback fails in front of the barrier (a halo buffer over its capacity: SPH_E_CAPACITY from sph_slab_copy_out)
 * still arrives at it, with the segment's `failed` word raised: every rank returns 1 instead of waiting for ever (round-4
 * advisor finding); a failure behind the barrier raises the word for the next collective. */
typedef int (*shm_out_fn)(void *user, int side, void *dst);
typedef int (*shm_in_fn)(void *user, int side, const void *src);
typedef int (*shm_mid_fn)(void *user);
static int shm_exchange(shm_comm *sc, int has_left, int has_right, shm_out_fn out, shm_mid_fn between, shm_in_fn in, void *user) {
    const unsigned long par = sc->seq++;
    int bad = 0;
    if (has_left && out(user, 0, shm_mail(sc, par, sc->rank, 0))) bad = 1;
    if (!bad && has_right && out(user, 1, shm_mail(sc, par, sc->rank, 1))) bad = 1;
    if (!bad && between && between(user)) bad = 1;
    if (bad) shm_fail(sc, par);
    if (shm_wait(sc, par)) return 1;
    if (has_left && in(user, 0, shm_mail(sc, par, sc->rank - 1, 1))) bad = 1;       /* what the left neighbour sent right */
    if (!bad && has_right && in(user, 1, shm_mail(sc, par, sc->rank + 1, 0))) bad = 1;
    if (bad) shm_fail(sc, par + 1ul);      /* (behind the barrier: the others have left this collective — they learn of it in the next) */
    return bad;
}

#endif /* SPH_SHM_H */
