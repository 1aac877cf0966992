/* slab_sph_fluid.c — the C multi-GPU host of the MI355X stepper: one process per GPU, x-slab decomposition, halo
 * exchange and rebuild-word reduction over RCCL (xGMI).
 *
 * The reference (pi_sph_fluid.c) has no distributed path: its particle loops (:272, :311) run on one node.  This host
 * shards them by cell column and drives the slab entry points of include/sph.h; per step and rank
 *     sph_slab_step_begin     kick 1/2 + drift of the owned particles (:615-624), may raise the rebuild word
 *     MAX-reduce the word     over all ranks: all slabs rebuild in the same step
 *     sph_slab_step_pack      fills the send buffers (full records on a rebuild step, x/y/u/v updates otherwise)
 *     exchange                send_right -> right neighbour's recv_left, send_left -> left neighbour's recv_right
 *     sph_slab_step_overlap   density of the tiles that stage no ghost particle, beside the exchange
 *     sph_slab_step_end       ingest + sort + lists | ghost update, density of the rest, force + kick (:626-640)
 * and, every 0.1 s of simulated time, the reference's console line (:679-691: ticks/s, max rho error, max speed — the two
 * maxima reduced over the ranks); every K steps (--rebalance-every K) new column ranges from the current per-column
 * particle histogram, the particles shipped to the slabs that now hold their columns, the slab contexts re-created.
 *
 * Transports (--transport):
 *   rccl  (default) ncclAllReduce(max) on the device word itself (sph_slab_flag_buffer), grouped ncclSend / ncclRecv on the
 *         device buffers (sph_slab_buffers) on a second stream; one rank per GPU (RCCL does not share a device)
 *   host  the same protocol with POSIX shared memory between the ranks: the word through sph_slab_flag_get / _set, the
 *         buffers through sph_slab_copy_out / _copy_in, a process-shared barrier between writing and reading.  Ranks may
 *         share a device: rehearsals and tests of THIS file's step loop, re-balancing and statistics on a one-GPU box.
 *   peer  the per-step traffic as plain stores into the neighbours' memory, mapped through hipIpc handles (xGMI is point to
 *         point), and flag words: sph_slab_peer_reduce / _push / _wait (include/sph.h), three small kernels per step on the
 *         one stream instead of an all-reduce and a send / receive kernel.  Set-up and the small collectives (statistics,
 *         re-balancing) go through the shared-memory segment of `host`.  Ranks may share a device (tests); on a node with
 *         one GPU per rank this is the transport without a collective library on the step path.  Up to 8 ranks.
 * The three differ only in the functions under "transport" below; the step loop is one.
 *
 *   slab_sph_fluid --ranks N [--transport rccl|host|peer] [--scene dam|cfg3|cfg4|cfg4slab] [--block NX NY BOXW BOXH] [--origin X0 Y0]
 *                  [--velocity U V] [--steps K] [--warmup W] [--windows N] [--tilt] [--check] [--deterministic] [--skin F]
 *                  [--rebalance-every K] [--capacity N] [--halo-capacity N] [--console] [--frame FILE] [--dump-state FILE] [--dump-accel FILE]
 *                  [--dump-before STATE_FILE ACCEL_FILE]      (the state and accelerations in front of the last step: tests pin ONE slab step to the oracle)
 *                  [--selfcomm] [--exchange-stream serial|main|side] [--breakdown K] [--lean auto|0|1] [--lean-graph 0|1] [--lean-spec 0|1|2] [--one-launch-wgs N] [--verify -1|0|1] [--repair -1|0|1]
 * starts N processes (fork + exec of this program with --rank r, before anything touches a GPU), rank r on device
 * r (rccl) or r mod devices (host).  --ranks 1 without --rank runs the one rank in this process: no fork, no exec (this is
 * what may sit under a profiler; the launcher must not: see profiles/README.md).  The ncclUniqueId travels through a file
 * (--id-file, made by the launcher), the shared memory through its name (--shm).  --scene dam: N lattice blocks of
 * 4000 x 500 (2 000 000 particles per GPU, box 1200 N x 60 m: the cfg2 -> cfg3 weak-scaling family); cfg3 / cfg4: the
 * fixed 8M / 32M scenes; --tilt: gravity from the scripted tilt trace (sph_gravity, the MPU6050 stand-in), sampled every
 * step with its 0.1 s hold.  Every rank generates only the lattice columns it holds.
 * --check (N = 1): the run is repeated with sph_step on a single context and the two final states are compared.
 * --exchange-stream (rccl): where the RCCL calls of a step go.  serial (default) = all-reduce, pack, send / receive and
 * everything else on ONE stream, nothing beside anything; main = the same, with the interior density pass on a side stream
 * from right after sph_slab_step_begin; side = the exchange on the side stream with the interior density pass beside it on
 * the main stream (round 2's order).  Measured with --selfcomm on one MI355X (us per step; 95-101 without any RCCL call):
 * serial 113-114, main 131-137, side 139-144 — a stream that is already waiting when the event it waits for fires resumes
 * ~13-15 us later, and the two overlapping orders pay that twice per step (fork and join), which is more than the ~15 us
 * send / receive kernel they hide; that kernel also takes twice as long (31 us) beside a density pass.
 * --breakdown K: K more steps after the timed ones with HIP events at the phase boundaries, per rank (begin / reduction of the
 * word / pack / exchange / end, us per step), printed per rank in the JSON line with its device, its particle counts and the
 * ranks its communicator counts (ncclCommCount): one run tells an overloaded rank from a slow interconnect.
 * --lean (default auto): the step as ONE call (sph_slab_step: four kernels — head, ghost update or rebuild, density, force — with the
 * update message written by the force pass of the step before and everything between the ranks inside those kernels) where that is
 * possible: the peer transport, or a slab without neighbours, and the device this rank's alone (or --one-launch-wgs N: ranks that
 * share a device cap the grids of their one-launch kernels so that all of them stay resident — tests).  0: the three-call step.
 * --lean-spec 2|1|0 (default 2, round 6): the SPECULATIVE lean step (sph_slab_set_speculative): the rebuild criterion — boxes, verification,
 * list repair — inside the launch of a speculative density pass, as in sph_step; the word goes round in the gate kernel.  One slab of
 * 2 M particles, five windows of 1000 steps of the dam break: 8 759 against 8 172 steps/s (rebuilds 230 against 561); 4 M developed: 4 195
 * against 3 098; 4 M at rest: 6 528 against 6 581.  2: its FUSED form — the head kernel's work (books, push of the update, wait for the
 * neighbours', ghost update) by the first workgroups of the density launch, ghost-staging tiles wait for them: three launches per step as
 * sph_step (2 M: 9 140 against 9 000 for the four-launch form, the first window 11 509 against 11 157; 4 M at rest 6 947 against 6 833).
 * 0: the plain lean step (the criterion's boxes in the head kernel, no verification).
 * --lean-graph 1|0 (default 1, round 6): the lean steps between two things the HOST does (window boundaries, console lines, re-balancing)
 * go to the library as ONE call per run of steps (sph_slab_steps: up to 16 steps per captured graph, their gravity samples in a device
 * array, step number and buffer parity taken from the device); 0: sph_slab_step, one call of four launches per step.
 * --selfcomm (N = 1, rccl; a measurement): the all-reduce and the grouped send / receive of every step are issued anyway,
 * to this rank itself: what the RCCL calls of a step cost (enqueue + their kernels) before any neighbour is waited for.
 */
#define _GNU_SOURCE
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <errno.h>
#include <fcntl.h>
#include <math.h>
#include <pthread.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#include "sph.h"
#include "sph_diag.h"      /* sph_time_kernel, sph_rebuild_reasons: the figures the bench line reports */
#include "sph_host.h"
#include "sph_shm.h"

#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "[rank %d] %s: %s\n", g_rank, #call, hipGetErrorString(e_)); return 1; } } while (0)
#define NCCLCHK(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) { fprintf(stderr, "[rank %d] %s: %s\n", g_rank, #call, ncclGetErrorString(r_)); return 1; } } while (0)
#define SPHCHK(ctx, call) do { int rc_ = (call); if (rc_ != SPH_OK) { fprintf(stderr, "[rank %d] %s: %d (%s)\n", g_rank, #call, rc_, (ctx) ? sph_last_error(ctx) : sph_error_string(rc_)); return 1; } } while (0)
#define CHK(call) do { if ((call) != 0) return 1; } while (0)

static int g_rank = 0;

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

typedef struct {
    float box_w, box_h, x0, y0, u0, v0;
    long nx, ny;
    const char *label;
} scene_t;

/* ------------------------------------------------------------------------------------------------------------------
 * transport: what the step loop needs from the ranks' interconnect
 * ---------------------------------------------------------------------------------------------------------------- */
enum { TR_RCCL = 0, TR_HOST = 1, TR_PEER = 2 };      /* (host and peer: set-up and the small collectives through shared memory) */

typedef struct comm {
    int kind, rank, nranks;
    /* rccl */
    ncclComm_t nccl;
    void *d_coll;                 /* device staging of the small collectives */
    /* host (and the set-up / small collectives of peer): the shared segment, sph_shm.h */
    shm_comm sh;
    size_t coll_bytes, halo_bytes;
    int selfcomm;                 /* measurement (--selfcomm, one rank, rccl): the step's RCCL calls made anyway, to itself */
    int serial;                   /* --exchange-stream: 2 = serial (default), 0 = main, 1 = side */
} comm;

static int comm_barrier(comm *cm) {
    if (cm->nranks == 1) return 0;
    if (cm->kind != TR_RCCL) return shm_barrier(&cm->sh);
    int one = 1;      /* (a 4-byte all-reduce: RCCL has no barrier call) */
    HIPCHK(hipMemcpy(cm->d_coll, &one, sizeof one, hipMemcpyHostToDevice));
    NCCLCHK(ncclAllReduce(cm->d_coll, cm->d_coll, 1, ncclInt32, ncclSum, cm->nccl, NULL));
    HIPCHK(hipStreamSynchronize(NULL));
    return 0;
}

/* element-wise reduction of a small host array over all ranks, in place: op 0 = sum of int64, 1 = max of float */
static int comm_allreduce(comm *cm, void *buf, size_t count, int op) {
    const size_t esz = op == 0 ? sizeof(long long) : sizeof(float), bytes = count * esz;
    if (cm->nranks == 1) return 0;
    if (cm->kind == TR_RCCL) {
        if (bytes > cm->coll_bytes) { fprintf(stderr, "[rank %d] collective of %zu bytes exceeds the staging size %zu\n", cm->rank, bytes, cm->coll_bytes); return 1; }
        HIPCHK(hipMemcpy(cm->d_coll, buf, bytes, hipMemcpyHostToDevice));
        NCCLCHK(ncclAllReduce(cm->d_coll, cm->d_coll, count, op == 0 ? ncclInt64 : ncclFloat32, op == 0 ? ncclSum : ncclMax, cm->nccl, NULL));
        HIPCHK(hipStreamSynchronize(NULL));
        HIPCHK(hipMemcpy(buf, cm->d_coll, bytes, hipMemcpyDeviceToHost));
        return 0;
    }
    return shm_allreduce(&cm->sh, buf, count, op);
}

/* the rebuild word of this step, MAX-reduced over the ranks, on the context's stream */
static int comm_reduce_word(comm *cm, sph_ctx *ctx, void *dev_word, hipStream_t st) {
    if (cm->nranks == 1 && !(cm->selfcomm && cm->kind == TR_RCCL)) return 0;
    if (cm->kind == TR_RCCL) {
        NCCLCHK(ncclAllReduce(dev_word, dev_word, 1, ncclUint32, ncclMax, cm->nccl, st));    /* 4 bytes, on the device word */
        return 0;
    }
    uint32_t w = 0;
    SPHCHK(ctx, sph_slab_flag_get(ctx, &w));
    float f = (float)w;
    CHK(comm_allreduce(cm, &f, 1, 1));
    SPHCHK(ctx, sph_slab_flag_set(ctx, f > 0.0f ? 1u : 0u));
    return 0;
}

static int deterministic_blocks_lean(int deterministic) { (void)deterministic; return 0; }      /* (the lean step keeps the deterministic order: nothing to exclude) */

typedef struct xchg {      /* what the exchange of a step needs besides the communicator */
    void *send_l, *send_r, *recv_l, *recv_r;
    size_t halo_bytes;
    int has_left, has_right;
    hipStream_t st, xst;
    hipEvent_t packed, arrived;
} xchg;

/* the callbacks of shm_exchange (sph_shm.h): the halo buffers of the context to and from the mailboxes */
static int mail_out(void *ctx, int side, void *dst) { SPHCHK((sph_ctx *)ctx, sph_slab_copy_out((sph_ctx *)ctx, side, dst)); return 0; }
static int mail_in(void *ctx, int side, const void *src) { SPHCHK((sph_ctx *)ctx, sph_slab_copy_in((sph_ctx *)ctx, side, src)); return 0; }
static int mail_between(void *ctx) { SPHCHK((sph_ctx *)ctx, sph_slab_step_overlap((sph_ctx *)ctx)); return 0; }

/* the halo exchange of this step with sph_slab_step_overlap beside it; on return the receive buffers are (rccl: will be,
 * in stream order) filled and sph_slab_step_end may follow */
static int comm_exchange(comm *cm, sph_ctx *ctx, const xchg *x) {
    if (cm->selfcomm && cm->kind == TR_RCCL && cm->nranks == 1) {
        /* what the enqueue and the RCCL kernels of a step cost without a neighbour to wait for: both halo buffers sent to
         * this rank itself (the slab has no neighbours: it ignores what arrives) */
        HIPCHK(hipEventRecord(x->packed, x->st));
        HIPCHK(hipStreamWaitEvent(x->xst, x->packed, 0));
        NCCLCHK(ncclGroupStart());
        NCCLCHK(ncclSend(x->send_l, x->halo_bytes, ncclChar, 0, cm->nccl, x->xst));
        NCCLCHK(ncclRecv(x->recv_l, x->halo_bytes, ncclChar, 0, cm->nccl, x->xst));
        NCCLCHK(ncclSend(x->send_r, x->halo_bytes, ncclChar, 0, cm->nccl, x->xst));
        NCCLCHK(ncclRecv(x->recv_r, x->halo_bytes, ncclChar, 0, cm->nccl, x->xst));
        NCCLCHK(ncclGroupEnd());
        HIPCHK(hipEventRecord(x->arrived, x->xst));
        SPHCHK(ctx, sph_slab_step_overlap(ctx));
        HIPCHK(hipStreamWaitEvent(x->st, x->arrived, 0));
        return 0;
    }
    if (!(x->has_left || x->has_right)) return 0;
    if (cm->kind == TR_RCCL) {
        /* the exchange on its own stream, behind the pack; the interior density pass runs beside it */
        HIPCHK(hipEventRecord(x->packed, x->st));
        HIPCHK(hipStreamWaitEvent(x->xst, x->packed, 0));
        NCCLCHK(ncclGroupStart());
        if (x->has_left) {
            NCCLCHK(ncclSend(x->send_l, x->halo_bytes, ncclChar, cm->rank - 1, cm->nccl, x->xst));
            NCCLCHK(ncclRecv(x->recv_l, x->halo_bytes, ncclChar, cm->rank - 1, cm->nccl, x->xst));
        }
        if (x->has_right) {
            NCCLCHK(ncclSend(x->send_r, x->halo_bytes, ncclChar, cm->rank + 1, cm->nccl, x->xst));
            NCCLCHK(ncclRecv(x->recv_r, x->halo_bytes, ncclChar, cm->rank + 1, cm->nccl, x->xst));
        }
        NCCLCHK(ncclGroupEnd());
        HIPCHK(hipEventRecord(x->arrived, x->xst));
        SPHCHK(ctx, sph_slab_step_overlap(ctx));
        HIPCHK(hipStreamWaitEvent(x->st, x->arrived, 0));
        return 0;
    }
    return shm_exchange(&cm->sh, x->has_left, x->has_right, mail_out, mail_between, mail_in, ctx);
}

/* rccl: the exchange on the MAIN stream (the interior density pass runs on the side stream: step_once) */
static int comm_exchange_inline(comm *cm, const xchg *x) {
    const int self = cm->selfcomm && cm->nranks == 1;
    if (!(x->has_left || x->has_right || self)) return 0;
    NCCLCHK(ncclGroupStart());
    if (x->has_left || self) {
        NCCLCHK(ncclSend(x->send_l, x->halo_bytes, ncclChar, self ? 0 : cm->rank - 1, cm->nccl, x->st));
        NCCLCHK(ncclRecv(x->recv_l, x->halo_bytes, ncclChar, self ? 0 : cm->rank - 1, cm->nccl, x->st));
    }
    if (x->has_right || self) {
        NCCLCHK(ncclSend(x->send_r, x->halo_bytes, ncclChar, self ? 0 : cm->rank + 1, cm->nccl, x->st));
        NCCLCHK(ncclRecv(x->recv_r, x->halo_bytes, ncclChar, self ? 0 : cm->rank + 1, cm->nccl, x->st));
    }
    NCCLCHK(ncclGroupEnd());
    return 0;
}

/* every rank sends cnt[q] records of `recw` floats to every rank q (re-balancing): out[q] -> what arrives, concatenated
 * in rank order, in *in (malloc'd) / *n_in */
static int comm_alltoallv(comm *cm, float *const *out, const long long *cnt, int recw, float **in, long long *n_in) {
    const int n = cm->nranks;
    long long *mat = (long long *)calloc((size_t)n * (size_t)n, sizeof(long long));      /* mat[s * n + d] = records s -> d */
    if (!mat) return 1;
    for (int q = 0; q < n; q++) mat[(size_t)cm->rank * n + q] = cnt[q];
    if (comm_allreduce(cm, mat, (size_t)n * (size_t)n, 0)) { free(mat); return 1; }
    long long total = 0;
    for (int s = 0; s < n; s++) total += mat[(size_t)s * n + cm->rank];
    float *dst = (float *)malloc(sizeof(float) * (size_t)recw * (size_t)(total ? total : 1));
    if (!dst) { free(mat); return 1; }
    const size_t rb = sizeof(float) * (size_t)recw;
    int rc = 0;
    if (cm->kind == TR_RCCL && n > 1) {
        long long sent = 0;
        for (int q = 0; q < n; q++) sent += q == cm->rank ? 0 : cnt[q];
        const long long recv = total - mat[(size_t)cm->rank * n + cm->rank];
        float *d_s = NULL, *d_r = NULL;
        if (hipMalloc((void **)&d_s, rb * (size_t)(sent ? sent : 1)) != hipSuccess || hipMalloc((void **)&d_r, rb * (size_t)(recv ? recv : 1)) != hipSuccess) rc = 1;
        long long so = 0;
        for (int q = 0; q < n && !rc; q++) {
            if (q == cm->rank || !cnt[q]) continue;
            if (hipMemcpy((char *)d_s + rb * (size_t)so, out[q], rb * (size_t)cnt[q], hipMemcpyHostToDevice) != hipSuccess) rc = 1;
            so += cnt[q];
        }
        if (!rc) {
            ncclGroupStart();
            long long s_off = 0, r_off = 0;
            for (int q = 0; q < n; q++) {
                if (q == cm->rank) continue;
                const long long a = cnt[q], b = mat[(size_t)q * n + cm->rank];
                if (a) ncclSend((char *)d_s + rb * (size_t)s_off, rb * (size_t)a, ncclChar, q, cm->nccl, NULL);
                if (b) ncclRecv((char *)d_r + rb * (size_t)r_off, rb * (size_t)b, ncclChar, q, cm->nccl, NULL);
                s_off += a;
                r_off += b;
            }
            if (ncclGroupEnd() != ncclSuccess || hipStreamSynchronize(NULL) != hipSuccess) rc = 1;
        }
        long long o = 0, r_off = 0;
        for (int s = 0; s < n && !rc; s++) {      /* rank order: what came from the others out of d_r, the own part from memory */
            const long long b = mat[(size_t)s * n + cm->rank];
            if (s == cm->rank) memcpy((char *)dst + rb * (size_t)o, out[s], rb * (size_t)b);
            else if (b) {
                if (hipMemcpy((char *)dst + rb * (size_t)o, (char *)d_r + rb * (size_t)r_off, rb * (size_t)b, hipMemcpyDeviceToHost) != hipSuccess) rc = 1;
                r_off += b;
            }
            o += b;
        }
        if (d_s) (void)hipFree(d_s);
        if (d_r) (void)hipFree(d_r);
    } else if (n > 1) {      /* host: one shared segment per (source, destination) pair that carries anything */
        const unsigned long seq = cm->sh.xseq++;
        char name[256];
        for (int q = 0; q < n && !rc; q++) {
            if (q == cm->rank || !cnt[q]) continue;
            snprintf(name, sizeof name, "%s.x%lu.%d.%d", cm->sh.shm_name, seq, cm->rank, q);
            shm_unlink(name);
            const int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
            const size_t bytes = rb * (size_t)cnt[q];
            void *m = MAP_FAILED;
            if (fd >= 0 && ftruncate(fd, (off_t)bytes) == 0) m = mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            if (fd >= 0) close(fd);
            if (m == MAP_FAILED) { rc = 1; break; }
            memcpy(m, out[q], bytes);
            munmap(m, bytes);
        }
        if (comm_barrier(cm)) rc = 1;
        long long o = 0;
        for (int s = 0; s < n && !rc; s++) {
            const long long b = mat[(size_t)s * n + cm->rank];
            if (s == cm->rank) memcpy((char *)dst + rb * (size_t)o, out[s], rb * (size_t)b);
            else if (b) {
                snprintf(name, sizeof name, "%s.x%lu.%d.%d", cm->sh.shm_name, seq, s, cm->rank);
                const int fd = shm_open(name, O_RDONLY, 0600);
                const size_t bytes = rb * (size_t)b;
                void *m = fd >= 0 ? mmap(NULL, bytes, PROT_READ, MAP_SHARED, fd, 0) : MAP_FAILED;
                if (fd >= 0) close(fd);
                if (m == MAP_FAILED) { rc = 1; break; }
                memcpy((char *)dst + rb * (size_t)o, m, bytes);
                munmap(m, bytes);
            }
            o += b;
        }
        if (comm_barrier(cm)) rc = 1;
        for (int q = 0; q < n; q++) {
            if (q == cm->rank || !cnt[q]) continue;
            snprintf(name, sizeof name, "%s.x%lu.%d.%d", cm->sh.shm_name, seq, cm->rank, q);
            shm_unlink(name);
        }
    } else {
        memcpy(dst, out[0], rb * (size_t)total);
    }
    free(mat);
    if (rc) { free(dst); return 1; }
    *in = dst;
    *n_in = total;
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------------
 * launcher: N ranks, before anything touches a GPU
 * ---------------------------------------------------------------------------------------------------------------- */
static int launch(int nranks, int argc, char **argv, const char *idfile, const char *shm_name) {
    pid_t *pids = (pid_t *)calloc((size_t)nranks, sizeof(pid_t));
    if (!pids) return 1;
    int worst = 0, started = 0;
    for (int r = 0; r < nranks; r++) {
        pid_t pid = fork();
        if (pid < 0) { perror("fork"); worst = 1; break; }
        if (pid == 0) {
            char **av = (char **)calloc((size_t)argc + 8, sizeof(char *));
            char rbuf[16];
            snprintf(rbuf, sizeof rbuf, "%d", r);
            int k = 0;
            for (int i = 0; i < argc; i++) av[k++] = argv[i];
            av[k++] = "--rank"; av[k++] = rbuf; av[k++] = "--id-file"; av[k++] = (char *)idfile;
            if (shm_name) { av[k++] = "--shm"; av[k++] = (char *)shm_name; }
            av[k] = NULL;
            execv("/proc/self/exe", av);
            perror("execv");
            _exit(127);
        }
        pids[r] = pid;
        started++;
    }
    int left = started, killed = 0, first_fail = 0;
    if (worst) {                                   /* fork failed: the ranks already started would wait for the others */
        for (int r = 0; r < nranks; r++) if (pids[r] > 0) kill(pids[r], SIGTERM);
        killed = 1;
    }
    while (left > 0) {
        int status = 0;
        pid_t pid = wait(&status);
        if (pid < 0) break;
        left--;
        for (int r = 0; r < nranks; r++) if (pids[r] == pid) pids[r] = 0;      /* reaped: never signal this pid again */
        const int rc = WIFEXITED(status) ? WEXITSTATUS(status) : 128 + (WIFSIGNALED(status) ? WTERMSIG(status) : 0);
        if (rc != 0 && !killed) {                  /* one rank failed: the others would wait for it forever */
            first_fail = rc;
            for (int r = 0; r < nranks; r++) if (pids[r] > 0) kill(pids[r], SIGTERM);
            killed = 1;
        }
    }
    if (first_fail) worst = first_fail;
    free(pids);
    return worst;
}

static int exchange_id(const char *idfile, int rank, ncclUniqueId *id) {
    if (rank == 0) {
        if (ncclGetUniqueId(id) != ncclSuccess) return 1;
        char tmp[300];
        snprintf(tmp, sizeof tmp, "%s.tmp", idfile);
        FILE *f = fopen(tmp, "wb");
        if (!f || fwrite(id, sizeof *id, 1, f) != 1) return 1;
        fclose(f);
        return rename(tmp, idfile) != 0;      /* atomic: readers see the whole id or nothing */
    }
    for (int tries = 0; tries < 6000; tries++) {      /* up to 60 s */
        FILE *f = fopen(idfile, "rb");
        if (f) {
            const size_t n = fread(id, sizeof *id, 1, f);
            fclose(f);
            if (n == 1) return 0;
        }
        usleep(10000);
    }
    return 1;
}

static float max_abs_diff(const sph_particle *a, const sph_particle *b, const unsigned *ids, long n, int field) {
    float m = 0;
    for (long k = 0; k < n; k++) {
        const float *pa = (const float *)&a[k], *pb = (const float *)&b[ids ? ids[k] : k];
        const float d = fabsf(pa[field] - pb[field]);
        if (d > m) m = d;
    }
    return m;
}

/* ------------------------------------------------------------------------------------------------------------------
 * one rank
 * ---------------------------------------------------------------------------------------------------------------- */
#define LEAN_RUN_MAX 256      /* steps per sph_slab_steps call (the library cuts them into graphs of 16, 8, 4, 2 and single steps) */
typedef struct rank_state {
    sph_params prm;
    comm cm;
    int device, transport, deterministic, capacity, halo_capacity;
    int own_device;              /* no other rank of this run on this rank's device (peer: as many devices as ranks) */
    int lean;                    /* the step is ONE call, sph_slab_step: four kernels, the exchange inside them (peer transport, or a slab alone) */
    int lean_graph;              /* ... and runs of lean steps go to sph_slab_steps (graphs of up to 16 steps) */
    int lean_spec;               /* ... and the lean step is the speculative one (sph_slab_set_speculative): the criterion inside the density launch */
    int verify, repair;          /* sph_set_verification / sph_set_list_repair: -1 automatic, 0 never, 1 always */
    int one_launch_wgs;          /* > 0: the one-launch kernels of this rank use at most so many workgroups (ranks sharing a device: they must all be resident) */
    sph_particle *walls;
    long nw;
    hipStream_t st, xst;
    hipEvent_t packed, arrived;
    sph_ctx *ctx;
    int c0, c1;                  /* owned columns */
    void *flag;
    xchg x;
    /* peer transport: this rank's block (send / receive buffers, arrival flags, word slots) and its peers' blocks as
     * mapped into this process */
    void *peer_blk, *peer_of[SPH_PEER_MAX_RANKS];
    size_t peer_halo;            /* bytes per halo buffer inside a block */
    uint32_t peer_tag;
    /* --breakdown: events around the parts of a step (serial rccl and peer: everything is on the one stream) */
    hipEvent_t bev[6];
    int bd_on;                   /* record them in this step */
} rank_state;

/* layout of a peer block: [send_l][send_r][recv_l, parity 0][recv_r, 0][recv_l, 1][recv_r, 1][flag from left | flag from right (256 B apart)][slots]
 * (the second pair of receive buffers: the lean step pushes the message of step t into parity t & 1 — see sph_slab_step in sph.h;
 * the three-kernel peer step uses parity 0 only) */
static size_t peer_off_recv2(const rank_state *rs, int side, int parity) { return (size_t)(2 + 2 * parity + side) * rs->peer_halo; }
static size_t peer_off_recv(const rank_state *rs, int side) { return peer_off_recv2(rs, side, 0); }
static size_t peer_off_flag(const rank_state *rs, int side) { return 6 * rs->peer_halo + (size_t)side * 256; }
static size_t peer_off_slots(const rank_state *rs) { return 6 * rs->peer_halo + 512; }
static size_t peer_block_bytes(const rank_state *rs) { return peer_off_slots(rs) + sizeof(uint32_t) * 2 * SPH_PEER_MAX_RANKS + 256; }

/* allocate and export this rank's block, open the others' (collective; once per run: the block outlives re-balancing) */
static int peer_setup(rank_state *rs, size_t halo_bytes) {
    comm *cm = &rs->cm;
    if (cm->nranks > SPH_PEER_MAX_RANKS) { fprintf(stderr, "[rank %d] the peer transport serves up to %d ranks\n", cm->rank, SPH_PEER_MAX_RANKS); return 1; }
    rs->peer_halo = (halo_bytes + 255) / 256 * 256;
    const size_t bytes = peer_block_bytes(rs);
    /* fine-grained device memory where the runtime exports it (what a peer writes must not linger in anybody's cache);
     * SPH_PEER_COARSE=1: plain hipMalloc (the kernels fence at system scope either way) */
    hipError_t e = hipErrorUnknown;
    int e_fine = 0;
    if (!getenv("SPH_PEER_COARSE")) e = hipExtMallocWithFlags(&rs->peer_blk, bytes, hipDeviceMallocFinegrained);
    if (e == hipSuccess) e_fine = 1;
    else { (void)hipGetLastError(); HIPCHK(hipMalloc(&rs->peer_blk, bytes)); }
    HIPCHK(hipMemset(rs->peer_blk, 0, bytes));
    HIPCHK(hipDeviceSynchronize());
    long long hs[SPH_PEER_MAX_RANKS * 8 + SPH_PEER_MAX_RANKS];      /* 64-byte handles, then the ranks' devices */
    memset(hs, 0, sizeof hs);
    if (cm->nranks > 1) {
        hipIpcMemHandle_t h;
        e = hipIpcGetMemHandle(&h, rs->peer_blk);
        if (e != hipSuccess && !getenv("SPH_PEER_COARSE")) {      /* (a runtime that does not export fine-grained memory) */
            (void)hipGetLastError();
            (void)hipFree(rs->peer_blk);
            e_fine = 0;
            HIPCHK(hipMalloc(&rs->peer_blk, bytes));
            HIPCHK(hipMemset(rs->peer_blk, 0, bytes));
            HIPCHK(hipDeviceSynchronize());
            e = hipIpcGetMemHandle(&h, rs->peer_blk);
        }
        if (e != hipSuccess) { fprintf(stderr, "[rank %d] hipIpcGetMemHandle: %s\n", cm->rank, hipGetErrorString(e)); return 1; }
        _Static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
        memcpy(&hs[cm->rank * 8], &h, 64);
        hs[SPH_PEER_MAX_RANKS * 8 + cm->rank] = rs->device;
        CHK(comm_allreduce(cm, hs, SPH_PEER_MAX_RANKS * 8 + SPH_PEER_MAX_RANKS, 0));      /* (sum with zeros: an all-gather) */
    }
    for (int q = 0; q < cm->nranks; q++) {
        if (q == cm->rank) { rs->peer_of[q] = rs->peer_blk; continue; }
        hipIpcMemHandle_t h;
        memcpy(&h, &hs[q * 8], 64);
        const int qdev = (int)hs[SPH_PEER_MAX_RANKS * 8 + q];
        if (qdev != rs->device) {      /* (another GPU of the node: its memory over xGMI) */
            int can = 0;
            (void)hipDeviceCanAccessPeer(&can, rs->device, qdev);
            if (!can) { fprintf(stderr, "[rank %d] device %d cannot map device %d\n", cm->rank, rs->device, qdev); return 1; }
            e = hipDeviceEnablePeerAccess(qdev, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { fprintf(stderr, "[rank %d] hipDeviceEnablePeerAccess(%d): %s\n", cm->rank, qdev, hipGetErrorString(e)); return 1; }
            (void)hipGetLastError();
        }
        e = hipIpcOpenMemHandle(&rs->peer_of[q], h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) { fprintf(stderr, "[rank %d] hipIpcOpenMemHandle(rank %d): %s\n", cm->rank, q, hipGetErrorString(e)); return 1; }
    }
    CHK(comm_barrier(cm));
    fprintf(stderr, "[rank %d] peer block of %zu bytes (%s), %d peer block(s) mapped\n", cm->rank, bytes,
            e_fine ? "fine-grained" : "coarse-grained: system-scope fences only", cm->nranks - 1);
    return 0;
}
static void peer_teardown(rank_state *rs) {
    for (int q = 0; q < rs->cm.nranks; q++)
        if (q != rs->cm.rank && rs->peer_of[q]) (void)hipIpcCloseMemHandle(rs->peer_of[q]);
    if (rs->peer_blk) (void)hipFree(rs->peer_blk);
}

/* a slab context for columns [c0, c1) from the particles given, wired to the rank's streams and transport */
static int make_context(rank_state *rs, int c0, int c1, const sph_particle *loc, const unsigned *ids, long n_loc, float gx, float gy) {
    sph_slab_desc desc = {c0, c1, rs->cm.rank > 0, rs->cm.rank < rs->cm.nranks - 1, rs->halo_capacity, rs->capacity};
    rs->ctx = NULL;
    SPHCHK(rs->ctx, sph_create_slab(&rs->ctx, &rs->prm, &desc, loc, ids, (int)n_loc, rs->walls, (int)rs->nw, gx, gy, rs->device));
    SPHCHK(rs->ctx, sph_set_stream(rs->ctx, rs->st));
    if (rs->verify >= 0) SPHCHK(rs->ctx, sph_set_verification(rs->ctx, rs->verify));
    if (rs->repair >= 0) SPHCHK(rs->ctx, sph_set_list_repair(rs->ctx, rs->repair));
    /* one rank per GPU (rccl): nothing else computes on this device, so what follows the halo exchange may run as one
     * launch with grid barriers (include/sph.h, sph_set_rebuild_launches); ranks that may share a device must not */
    if (rs->one_launch_wgs > 0) SPHCHK(rs->ctx, sph_set_rebuild_launches(rs->ctx, rs->one_launch_wgs));      /* (a capped grid: ranks that share a device) */
    else if (rs->transport == TR_RCCL || rs->lean || (rs->transport == TR_PEER && rs->own_device)) SPHCHK(rs->ctx, sph_set_rebuild_launches(rs->ctx, 1));
    if (rs->transport == TR_PEER) {      /* the context sends from and receives into this rank's block */
        char *b = (char *)rs->peer_blk;
        SPHCHK(rs->ctx, sph_slab_set_buffers(rs->ctx, b, b + rs->peer_halo, b + peer_off_recv(rs, 0), b + peer_off_recv(rs, 1), rs->peer_halo));
    }
    if (rs->lean && rs->transport == TR_PEER && rs->cm.nranks > 1) {      /* the lean step talks to the other ranks' blocks itself */
        const int me = rs->cm.rank, n = rs->cm.nranks;
        sph_peer_links L;
        memset(&L, 0, sizeof L);
        L.me = me;
        L.n_ranks = n;
        for (int q = 0; q < n; q++) L.slots_of_rank[q] = (char *)rs->peer_of[q] + peer_off_slots(rs);
        char *M = (char *)rs->peer_blk;
        for (int par = 0; par < 2; par++) {
            L.my_recv_left[par] = M + peer_off_recv2(rs, 0, par);
            L.my_recv_right[par] = M + peer_off_recv2(rs, 1, par);
            /* my left neighbour receives me on ITS right side, and the other way round */
            if (desc.has_left) L.left_recv[par] = (char *)rs->peer_of[me - 1] + peer_off_recv2(rs, 1, par);
            if (desc.has_right) L.right_recv[par] = (char *)rs->peer_of[me + 1] + peer_off_recv2(rs, 0, par);
        }
        L.my_flag_left = M + peer_off_flag(rs, 0);
        L.my_flag_right = M + peer_off_flag(rs, 1);
        if (desc.has_left) L.left_flag = (char *)rs->peer_of[me - 1] + peer_off_flag(rs, 1);
        if (desc.has_right) L.right_flag = (char *)rs->peer_of[me + 1] + peer_off_flag(rs, 0);
        SPHCHK(rs->ctx, sph_slab_set_peer_links(rs->ctx, &L));
    }
    if (rs->lean && rs->lean_spec) SPHCHK(rs->ctx, sph_slab_set_speculative(rs->ctx, rs->lean_spec));
    SPHCHK(rs->ctx, sph_slab_flag_buffer(rs->ctx, &rs->flag));
    SPHCHK(rs->ctx, sph_slab_buffers(rs->ctx, &rs->x.send_l, &rs->x.send_r, &rs->x.recv_l, &rs->x.recv_r, &rs->x.halo_bytes));
    if (rs->transport == TR_PEER && rs->x.halo_bytes > rs->peer_halo) { fprintf(stderr, "[rank %d] halo buffers outgrew the peer block\n", rs->cm.rank); return 1; }
    if (rs->transport == TR_HOST && rs->cm.nranks > 1 && rs->x.halo_bytes > rs->cm.halo_bytes) {
        fprintf(stderr, "[rank %d] halo buffers of %zu bytes exceed the shared mailboxes (%zu)\n", rs->cm.rank, rs->x.halo_bytes, rs->cm.halo_bytes);
        return 1;
    }
    rs->x.has_left = desc.has_left;
    rs->x.has_right = desc.has_right;
    rs->x.st = rs->st;
    rs->x.xst = rs->xst;
    rs->x.packed = rs->packed;
    rs->x.arrived = rs->arrived;
    rs->c0 = c0;
    rs->c1 = c1;
    return 0;
}

/* every rank's owned particles (records of sph_particle) and / or their accelerations (du_dt, dv_dt: two floats) into one file each, by
 * global id: rank 0 makes the files, then everybody writes its records in place.  Between steps (sph_slab_read recomputes what the fused
 * step does not store — the velocity between steps, the acceleration — without touching what the next step reads).  Collective. */
static int dump_owned(rank_state *rs, const char *state_file, const char *accel_file, long n_total) {
    int n_local = 0, n_owned = 0, n_out = 0;
    SPHCHK(rs->ctx, sph_slab_counts(rs->ctx, &n_local, &n_owned));
    sph_particle *got = (sph_particle *)malloc(sizeof(sph_particle) * (size_t)(n_owned + 1));
    unsigned *gid = (unsigned *)malloc(sizeof(unsigned) * (size_t)(n_owned + 1));
    float *du = (float *)malloc(sizeof(float) * (size_t)(n_owned + 1)), *dv = (float *)malloc(sizeof(float) * (size_t)(n_owned + 1));
    if (!got || !gid || !du || !dv) return 1;
    SPHCHK(rs->ctx, sph_slab_read(rs->ctx, got, gid, du, dv, n_owned, &n_out));
    const char *files[2] = {state_file, accel_file};
    const size_t rec[2] = {sizeof(sph_particle), 2 * sizeof(float)};
    for (int f = 0; f < 2; f++) {
        if (!files[f]) continue;
        if (rs->cm.rank == 0) {
            FILE *fh = fopen(files[f], "wb");
            if (!fh || ftruncate(fileno(fh), (off_t)(rec[f] * (size_t)n_total)) != 0) { fprintf(stderr, "cannot write %s\n", files[f]); return 1; }
            fclose(fh);
        }
        CHK(comm_barrier(&rs->cm));
        const int fd = open(files[f], O_WRONLY);
        if (fd < 0) { fprintf(stderr, "[rank %d] cannot open %s\n", rs->cm.rank, files[f]); return 1; }
        for (int k = 0; k < n_out; k++) {
            const float a2[2] = {du[k], dv[k]};
            const void *src = f == 0 ? (const void *)&got[k] : (const void *)a2;
            if (pwrite(fd, src, rec[f], (off_t)(rec[f] * (size_t)gid[k])) != (ssize_t)rec[f]) return 1;
        }
        close(fd);
        CHK(comm_barrier(&rs->cm));
    }
    free(got); free(gid); free(du); free(dv);
    return 0;
}

/* one time step (:612-641) of this rank's slab */
#define BD(k) do { if (rs->bd_on) HIPCHK(hipEventRecord(rs->bev[k], rs->st)); } while (0)
static int step_once(rank_state *rs, float gx, float gy) {
    BD(0);
    if (rs->lean) {      /* one call, four kernels (head | ghost update or rebuild | density | force): sph.h, "the lean slab step" */
        SPHCHK(rs->ctx, sph_slab_step(rs->ctx, gx, gy));
        BD(1); BD(2); BD(3); BD(4); BD(5);      /* (the breakdown has nothing to tell apart: the whole step shows as `begin`) */
        return 0;
    }
    SPHCHK(rs->ctx, sph_slab_step_begin(rs->ctx, gx, gy));
    BD(1);
    if (rs->cm.kind == TR_PEER) {
        /* everything between the ranks as stores into mapped peer memory and flag words, on the one stream */
        const int me = rs->cm.rank, n = rs->cm.nranks;
        const uint32_t tag = ++rs->peer_tag;
        void *slots[SPH_PEER_MAX_RANKS];
        for (int q = 0; q < n; q++) slots[q] = (char *)rs->peer_of[q] + peer_off_slots(rs);
        if (n > 1 || rs->cm.selfcomm) SPHCHK(rs->ctx, sph_slab_peer_reduce(rs->ctx, slots, me, n, tag));
        BD(2);
        SPHCHK(rs->ctx, sph_slab_step_pack(rs->ctx));
        BD(3);
        if (rs->cm.selfcomm && n == 1) {      /* (a measurement: the three kernels of the step against this rank's own block) */
            char *M = (char *)rs->peer_blk;
            SPHCHK(rs->ctx, sph_slab_peer_push(rs->ctx, M + peer_off_recv(rs, 1), M + peer_off_flag(rs, 1), M + peer_off_recv(rs, 0), M + peer_off_flag(rs, 0), tag));
            SPHCHK(rs->ctx, sph_slab_peer_wait(rs->ctx, M + peer_off_flag(rs, 0), M + peer_off_flag(rs, 1), tag));
        } else if (rs->x.has_left || rs->x.has_right) {
            char *L = rs->x.has_left ? (char *)rs->peer_of[me - 1] : NULL, *R = rs->x.has_right ? (char *)rs->peer_of[me + 1] : NULL;
            /* my left neighbour receives me on ITS right side, and the other way round */
            SPHCHK(rs->ctx, sph_slab_peer_push(rs->ctx, L ? L + peer_off_recv(rs, 1) : NULL, L ? L + peer_off_flag(rs, 1) : NULL,
                                               R ? R + peer_off_recv(rs, 0) : NULL, R ? R + peer_off_flag(rs, 0) : NULL, tag));
            char *M = (char *)rs->peer_blk;
            SPHCHK(rs->ctx, sph_slab_peer_wait(rs->ctx, rs->x.has_left ? M + peer_off_flag(rs, 0) : NULL,
                                               rs->x.has_right ? M + peer_off_flag(rs, 1) : NULL, tag));
        }
        BD(4);
        SPHCHK(rs->ctx, sph_slab_step_end(rs->ctx));
        BD(5);
        return 0;
    }
    if (rs->cm.kind == TR_RCCL && (rs->cm.nranks > 1 || rs->cm.selfcomm) && rs->cm.serial == 2) {
        /* everything on the main stream, nothing beside anything: no cross-stream event at all */
        CHK(comm_reduce_word(&rs->cm, rs->ctx, rs->flag, rs->st));
        BD(2);
        SPHCHK(rs->ctx, sph_slab_step_pack(rs->ctx));
        BD(3);
        CHK(comm_exchange_inline(&rs->cm, &rs->x));
        BD(4);
        SPHCHK(rs->ctx, sph_slab_step_end(rs->ctx));
        BD(5);
        return 0;
    }
    if (rs->cm.kind == TR_RCCL && (rs->cm.nranks > 1 || rs->cm.selfcomm) && !rs->cm.serial) {
        /* The interior density pass on the side stream from here on: beside the all-reduce of the word, the pack and the
         * exchange, which all stay on the main stream (a cross-stream event in FRONT of the RCCL kernels and another behind
         * them cost ~6 + ~15 us of queue latency per step, measured with --selfcomm; waiting for a pass that has long
         * finished costs nothing).  On a step that turns out to rebuild the pass has worked for nothing: see sph.h. */
        HIPCHK(hipEventRecord(rs->x.packed, rs->st));
        HIPCHK(hipStreamWaitEvent(rs->xst, rs->x.packed, 0));
        SPHCHK(rs->ctx, sph_slab_step_overlap_on(rs->ctx, rs->xst));
        HIPCHK(hipEventRecord(rs->x.arrived, rs->xst));
        CHK(comm_reduce_word(&rs->cm, rs->ctx, rs->flag, rs->st));
        SPHCHK(rs->ctx, sph_slab_step_pack(rs->ctx));
        CHK(comm_exchange_inline(&rs->cm, &rs->x));
        HIPCHK(hipStreamWaitEvent(rs->st, rs->x.arrived, 0));
        SPHCHK(rs->ctx, sph_slab_step_end(rs->ctx));
        return 0;
    }
    CHK(comm_reduce_word(&rs->cm, rs->ctx, rs->flag, rs->st));
    BD(2);
    SPHCHK(rs->ctx, sph_slab_step_pack(rs->ctx));
    BD(3);
    CHK(comm_exchange(&rs->cm, rs->ctx, &rs->x));
    BD(4);
    SPHCHK(rs->ctx, sph_slab_step_end(rs->ctx));
    BD(5);
    return 0;
}
#undef BD

/* Dynamic re-balancing: new column ranges at the quantiles of the CURRENT per-column particle histogram
 * (summed over the ranks), every particle shipped to the slab that now holds its column (owned + 2 ghost columns), the
 * slab context re-created from what arrived — it evaluates rho, p, a from (x, v) like the reference's init sequence
 * (:604-607): the run is continued, not bit-continued.  Collective.  *moved = 0 when the largest slab would shrink by
 * less than min_gain (nothing is touched then). */
static int rebalance(rank_state *rs, float gx, float gy, double min_gain, int *moved) {
    comm *cm = &rs->cm;
    const int n = cm->nranks, cols = sph_slab_grid_columns(&rs->prm);
    *moved = 0;
    int n_local = 0, n_owned = 0, n_out = 0;
    SPHCHK(rs->ctx, sph_slab_counts(rs->ctx, &n_local, &n_owned));
    sph_particle *own = (sph_particle *)malloc(sizeof(sph_particle) * (size_t)(n_owned + 1));
    unsigned *ids = (unsigned *)malloc(sizeof(unsigned) * (size_t)(n_owned + 1));
    long long *hist = (long long *)calloc((size_t)cols + (size_t)n, sizeof(long long));      /* per column, then per rank */
    int *cuts = (int *)calloc((size_t)n + 1, sizeof(int)), *pcol = (int *)malloc(sizeof(int) * (size_t)(n_owned + 1));
    if (!own || !ids || !hist || !cuts || !pcol) return 1;
    SPHCHK(rs->ctx, sph_slab_read(rs->ctx, own, ids, NULL, NULL, n_owned, &n_out));
    for (int k = 0; k < n_out; k++) {
        int c = sph_slab_column_of(&rs->prm, own[k].x);
        c = c < 0 ? 0 : (c > cols - 1 ? cols - 1 : c);
        pcol[k] = c;
        hist[c]++;
    }
    hist[cols + cm->rank] = n_out;
    CHK(comm_allreduce(cm, hist, (size_t)cols + (size_t)n, 0));
    if (sph_slab_partition_counts(hist, cols, n, cuts) != SPH_OK) { fprintf(stderr, "[rank %d] re-balancing: the fluid is too narrow for %d slabs\n", cm->rank, n); return 1; }
    long long old_max = 0, new_max = 0, run = 0;
    for (int r = 0; r < n; r++) if (hist[cols + r] > old_max) old_max = hist[cols + r];
    for (int r = 0, c = 0; r < n; r++) {
        for (run = 0; c < cuts[r + 1]; c++) run += hist[c];
        if (run > new_max) new_max = run;
    }
    if ((double)(old_max - new_max) < min_gain * (double)(old_max > 1 ? old_max : 1)) {
        free(own); free(ids); free(hist); free(cuts); free(pcol);
        return 0;
    }
    /* records {x, y, u, v, m, rho, p, id}: to every slab whose columns [c0 - 2, c1 + 2) hold the particle */
    enum { RECW = 8 };
    long long *cnt = (long long *)calloc((size_t)n, sizeof(long long));
    float **out = (float **)calloc((size_t)n, sizeof(float *));
    if (!cnt || !out) return 1;
    for (int pass = 0; pass < 2; pass++) {
        if (pass) for (int q = 0; q < n; q++) { out[q] = (float *)malloc(sizeof(float) * RECW * (size_t)(cnt[q] + 1)); if (!out[q]) return 1; cnt[q] = 0; }
        for (int k = 0; k < n_out; k++)
            for (int q = 0; q < n; q++)
                if (pcol[k] >= cuts[q] - 2 && pcol[k] < cuts[q + 1] + 2) {
                    if (pass) {
                        float *r = out[q] + (size_t)RECW * (size_t)cnt[q];
                        memcpy(r, &own[k], sizeof(sph_particle));
                        memcpy(r + 7, &ids[k], sizeof(unsigned));
                    }
                    cnt[q]++;
                }
    }
    float *in = NULL;
    long long n_in = 0;
    CHK(comm_alltoallv(cm, out, cnt, RECW, &in, &n_in));
    sph_particle *loc = (sph_particle *)malloc(sizeof(sph_particle) * (size_t)(n_in + 1));
    unsigned *lid = (unsigned *)malloc(sizeof(unsigned) * (size_t)(n_in + 1));
    if (!loc || !lid) return 1;
    for (long long k = 0; k < n_in; k++) {
        memcpy(&loc[k], in + (size_t)RECW * (size_t)k, sizeof(sph_particle));
        memcpy(&lid[k], in + (size_t)RECW * (size_t)k + 7, sizeof(unsigned));
    }
    HIPCHK(hipStreamSynchronize(rs->st));
    HIPCHK(hipStreamSynchronize(rs->xst));
    sph_destroy(rs->ctx);
    if (rs->lean && rs->transport == TR_PEER) {
        /* the lean step's tags are step numbers, and the new contexts count from 1 again: this rank's arrival flags and word slots
         * start from 0 too (nobody writes into them before the barrier at the end of this function, which every rank reaches
         * only after this) */
        HIPCHK(hipMemset((char *)rs->peer_blk + peer_off_flag(rs, 0), 0, peer_block_bytes(rs) - peer_off_flag(rs, 0)));
        HIPCHK(hipDeviceSynchronize());
    }
    CHK(make_context(rs, cuts[cm->rank], cuts[cm->rank + 1], loc, lid, (long)n_in, gx, gy));
    HIPCHK(hipStreamSynchronize(rs->st));
    CHK(comm_barrier(cm));      /* (nobody steps before every rank has its new context: see main) */
    *moved = 1;
    for (int q = 0; q < n; q++) free(out[q]);
    free(out); free(cnt); free(in); free(loc); free(lid);
    free(own); free(ids); free(hist); free(cuts); free(pcol);
    return 0;
}

int main(int argc, char **argv) {
    int nranks = 1, rank = -1, steps = 200, warmup = 50, windows = 1, tilt = 0, check = 0, deterministic = 0, transport = TR_RCCL;
    int rebalance_every = 0, capacity = 0, console = 0, selfcomm = 0, xside = 2, breakdown = 0, halo_capacity = -1;
    int repair_opt = -1;                        /* --repair -1 (default: from 4 000 000 particles per slab on) | 0 | 1: sph_set_list_repair */
    int verify_opt = -1;                        /* --verify -1 (default: the library's — slab contexts verify only when asked: 1) | 0 | 1: sph_set_verification */
    int lean_graph_opt = 1, lean_spec_opt = 2;
    int lean_opt = -1, one_launch_wgs = 0;      /* --lean auto (-1) | 0 | 1; --one-launch-wgs N: cap of the one-launch kernels' grid (ranks that share a device) */
    float skin = -1;
    const char *scene_name = "dam", *idfile = NULL, *shm_name = NULL, *frame_file = NULL, *state_file = NULL, *accel_file = NULL, *before_file = NULL, *before_accel_file = NULL;
    scene_t sc = {0, 0, 0.3f, 0.3f, 0, 0, 0, 0, NULL};
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--ranks") && i + 1 < argc) nranks = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--rank") && i + 1 < argc) rank = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--id-file") && i + 1 < argc) idfile = argv[++i];
        else if (!strcmp(argv[i], "--shm") && i + 1 < argc) shm_name = argv[++i];
        else if (!strcmp(argv[i], "--transport") && i + 1 < argc) {
            const char *t = argv[++i];
            if (!strcmp(t, "rccl")) transport = TR_RCCL;
            else if (!strcmp(t, "host")) transport = TR_HOST;
            else if (!strcmp(t, "peer")) transport = TR_PEER;
            else { fprintf(stderr, "unknown transport %s (rccl | host)\n", t); return 2; }
        }
        else if (!strcmp(argv[i], "--scene") && i + 1 < argc) scene_name = argv[++i];
        else if (!strcmp(argv[i], "--steps") && i + 1 < argc) steps = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--warmup") && i + 1 < argc) warmup = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--windows") && i + 1 < argc) windows = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--rebalance-every") && i + 1 < argc) rebalance_every = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--capacity") && i + 1 < argc) capacity = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--halo-capacity") && i + 1 < argc) halo_capacity = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--breakdown") && i + 1 < argc) breakdown = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--frame") && i + 1 < argc) frame_file = argv[++i];
        else if (!strcmp(argv[i], "--dump-state") && i + 1 < argc) state_file = argv[++i];
        else if (!strcmp(argv[i], "--dump-accel") && i + 1 < argc) accel_file = argv[++i];
        else if (!strcmp(argv[i], "--dump-before") && i + 2 < argc) { before_file = argv[++i]; before_accel_file = argv[++i]; }
        else if (!strcmp(argv[i], "--tilt")) tilt = 1;
        else if (!strcmp(argv[i], "--console")) console = 1;
        else if (!strcmp(argv[i], "--selfcomm")) selfcomm = 1;
        else if (!strcmp(argv[i], "--lean") && i + 1 < argc) { i++; lean_opt = !strcmp(argv[i], "auto") ? -1 : atoi(argv[i]); }
        else if (!strcmp(argv[i], "--lean-graph") && i + 1 < argc) lean_graph_opt = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--lean-spec") && i + 1 < argc) lean_spec_opt = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--one-launch-wgs") && i + 1 < argc) one_launch_wgs = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--verify") && i + 1 < argc) verify_opt = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--repair") && i + 1 < argc) repair_opt = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--exchange-stream") && i + 1 < argc) { i++; xside = !strcmp(argv[i], "side") ? 1 : !strcmp(argv[i], "main") ? 0 : 2; }
        else if (!strcmp(argv[i], "--deterministic")) deterministic = 1;
        else if (!strcmp(argv[i], "--skin") && i + 1 < argc) skin = (float)atof(argv[++i]);
        else if (!strcmp(argv[i], "--check")) check = 1;
        else if (!strcmp(argv[i], "--block") && i + 4 < argc) {
            sc.nx = atol(argv[++i]); sc.ny = atol(argv[++i]); sc.box_w = (float)atof(argv[++i]); sc.box_h = (float)atof(argv[++i]);
            sc.label = "custom block";
        }
        else if (!strcmp(argv[i], "--origin") && i + 2 < argc) { sc.x0 = (float)atof(argv[++i]); sc.y0 = (float)atof(argv[++i]); }
        else if (!strcmp(argv[i], "--velocity") && i + 2 < argc) { sc.u0 = (float)atof(argv[++i]); sc.v0 = (float)atof(argv[++i]); }
        else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    if (nranks < 1 || steps < 0 || warmup < 0 || rebalance_every < 0 || windows < 1 || windows > 64) return 2;

    /* ---- scene and parameters (the launcher needs them too: the size of the shared mailboxes) ---- */
    if (!sc.nx) {
        if (!strcmp(scene_name, "dam")) { sc.nx = 4000L * nranks; sc.ny = 500; sc.box_w = 1200.0f * (float)nranks; sc.box_h = 60.0f; sc.label = "dam break, 2 000 000 fluid particles per slab"; }
        else if (!strcmp(scene_name, "cfg3")) { sc.nx = 16000; sc.ny = 500; sc.box_w = 2400.0f; sc.box_h = 60.0f; sc.label = "cfg3: 8M dam break"; }
        else if (!strcmp(scene_name, "cfg4")) { sc.nx = 32000; sc.ny = 1000; sc.box_w = 2400.6f; sc.box_h = 150.0f; sc.label = "cfg4: 32M tank"; }
        else if (!strcmp(scene_name, "cfg4slab")) { sc.nx = 4000; sc.ny = 1000; sc.box_w = 300.6f; sc.box_h = 150.0f; sc.label = "cfg4's slab: 4M tank"; }
        else { fprintf(stderr, "unknown scene %s\n", scene_name); return 2; }
    }
    sph_params prm;
    sph_params_default(&prm);
    prm.deterministic = deterministic;
    if (skin >= 0) prm.skin = prm.skin_min = skin;      /* a fixed skin */
    prm.x_max = sc.box_w;
    prm.y_max = sc.box_h;
    const int cols = sph_slab_grid_columns(&prm);
    /* Records per halo buffer.  RCCL sends a whole buffer every step (the counts live on the device and ncclSend wants its size
     * on the host), so the buffers are cut to the scene instead of the library's default (64 particles per row of cells): the two
     * outermost columns of a lattice block ny high hold 2 ny cell / R particles at rest; 2.5 x that for what a collapse piles up
     * (a rank that needs more ends in SPH_E_CAPACITY, not in wrong results).  --halo-capacity N sets it, 0 = the library's. */
    if (halo_capacity < 0) {
        const double per_col = (double)sc.ny * (double)sph_device_cell(&prm) / (double)prm.r;
        long want = (long)(2.5 * 2.0 * per_col) + 1024, dflt = (long)(sph_slab_halo_bytes(&prm, 0) / (5 * sizeof(uint32_t)));
        halo_capacity = (int)(want < dflt ? want : 0);
    }
    const size_t halo_bytes = sph_slab_halo_bytes(&prm, halo_capacity);
    size_t coll_bytes = sizeof(long long) * ((size_t)cols + (size_t)nranks * (size_t)nranks + 64);      /* histogram + counts ... */
    if (coll_bytes < 16384) coll_bytes = 16384;                                                          /* ... or the 1024 bytes of a frame as int64 */

    char idbuf[256], shmbuf[160];
    if (rank < 0) {
        snprintf(idbuf, sizeof idbuf, "/tmp/slab_sph_fluid.%d.%ld.id", (int)getpid(), (long)time(NULL));
        snprintf(shmbuf, sizeof shmbuf, "/slab_sph_fluid.%d.%ld", (int)getpid(), (long)time(NULL));
        unlink(idbuf);
        if (transport != TR_RCCL && nranks > 1 && shm_create(shmbuf, nranks, halo_bytes, coll_bytes)) return 1;
        if (nranks > 1) {
            const int rc = launch(nranks, argc, argv, idbuf, transport != TR_RCCL ? shmbuf : NULL);
            unlink(idbuf);
            if (transport != TR_RCCL) shm_unlink(shmbuf);
            return rc;
        }
        rank = 0;                      /* one rank: in this process (no fork, no exec) */
        idfile = idbuf;
    }
    g_rank = rank;

    rank_state rs;
    memset(&rs, 0, sizeof rs);
    rs.prm = prm;
    rs.transport = transport;
    rs.deterministic = deterministic;
    rs.capacity = capacity;
    rs.halo_capacity = halo_capacity;
    rs.cm.kind = transport;
    rs.cm.rank = rs.cm.sh.rank = rank;
    rs.cm.nranks = rs.cm.sh.nranks = nranks;
    rs.cm.selfcomm = selfcomm;
    rs.cm.serial = xside;
    rs.cm.coll_bytes = coll_bytes;
    rs.cm.halo_bytes = (halo_bytes + 63) / 64 * 64;

    /* ---- this rank's columns and particles ---- */
    int *cuts = (int *)calloc((size_t)nranks + 1, sizeof(int));
    SPHCHK(NULL, sph_slab_partition_block(&prm, sc.x0, sc.nx, sc.ny, nranks, cuts));
    const int c0 = cuts[rank], c1 = cuts[rank + 1];
    long ib = 0, ie = 0;
    SPHCHK(NULL, sph_slab_block_columns(&prm, sc.x0, sc.nx, c0, c1, &ib, &ie));
    const long n_loc = (ie - ib) * sc.ny, n_total = sc.nx * sc.ny;
    sph_particle *loc = (sph_particle *)malloc(sizeof(sph_particle) * (size_t)(n_loc ? n_loc : 1));
    unsigned *ids = (unsigned *)malloc(sizeof(unsigned) * (size_t)(n_loc ? n_loc : 1));
    rs.nw = sph_scene_walls(&prm, 0, NULL, 0);
    rs.walls = (sph_particle *)malloc(sizeof(sph_particle) * (size_t)rs.nw);
    if (!loc || !ids || !rs.walls) { fprintf(stderr, "[rank %d] out of host memory\n", rank); return 1; }
    if (sph_scene_block_range(&prm, sc.x0, sc.y0, sc.nx, sc.ny, ib, ie, loc, n_loc) != n_loc) return 1;
    for (long k = 0; k < n_loc; k++) { ids[k] = (unsigned)(ib * sc.ny + k); loc[k].u = sc.u0; loc[k].v = sc.v0; }
    sph_scene_walls(&prm, 0, rs.walls, rs.nw);

    sph_gravity grav;
    sph_gravity_init(&grav, tilt ? SPH_GRAVITY_TILT : SPH_GRAVITY_CONSTANT, prm.g);
    float gx, gy, t = 0;
    sph_gravity_sample(&grav, 0.0f, &gx, &gy);

    /* ---- device, streams, communicator, slab context ---- */
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        fprintf(stderr, "[rank %d] no HIP device available (this program has no CPU path)\n", rank);
        return 1;
    }
    if (transport == TR_RCCL && nranks > ndev) {
        fprintf(stderr, "[rank %d] %d ranks need %d GPUs (found %d): RCCL does not share a device between ranks "
                        "(--transport host rehearses the same step loop on fewer)\n", rank, nranks, nranks, ndev);
        return 1;
    }
    rs.device = transport == TR_RCCL ? rank : rank % ndev;
    rs.own_device = nranks <= ndev;
    rs.one_launch_wgs = one_launch_wgs;
    rs.verify = verify_opt;
    rs.repair = repair_opt;
    /* The lean step needs the one-launch kernels (the device is this rank's alone, or the ranks that share it cap their grids so
     * that all stay resident) and a transport that lives inside the step's kernels: peer-mapped memory — or no neighbour at all. */
    {
        const int can = (rs.own_device || one_launch_wgs > 0) && !selfcomm && !deterministic_blocks_lean(deterministic) &&
                        (nranks == 1 || transport == TR_PEER);
        rs.lean = lean_opt < 0 ? can : (lean_opt && can);
        rs.lean_graph = rs.lean && lean_graph_opt != 0;
        rs.lean_spec = rs.lean ? (lean_spec_opt < 0 ? 0 : lean_spec_opt > 2 ? 2 : lean_spec_opt) : 0;
        if (lean_opt > 0 && !can) { fprintf(stderr, "[rank %d] --lean 1 needs --transport peer (or one rank) and the device to itself (or --one-launch-wgs)\n", rank); return 2; }
    }
    HIPCHK(hipSetDevice(rs.device));
    HIPCHK(hipStreamCreateWithFlags(&rs.st, hipStreamNonBlocking));      /* compute stream (adopted by the context) ... */
    HIPCHK(hipStreamCreateWithFlags(&rs.xst, hipStreamNonBlocking));     /* ... and exchange stream */
    HIPCHK(hipEventCreateWithFlags(&rs.packed, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&rs.arrived, hipEventDisableTiming));
    if (transport == TR_RCCL) {
        ncclUniqueId id;
        if (nranks == 1) NCCLCHK(ncclGetUniqueId(&id));
        else if (!idfile || exchange_id(idfile, rank, &id)) { fprintf(stderr, "[rank %d] could not exchange the ncclUniqueId\n", rank); return 1; }
        NCCLCHK(ncclCommInitRank(&rs.cm.nccl, nranks, id, rank));
        if (rank == 0 && nranks > 1 && idfile) unlink(idfile);           /* every rank has joined: the file is spent */
        HIPCHK(hipMalloc(&rs.cm.d_coll, coll_bytes));
    } else if (nranks > 1) {
        if (!shm_name) { fprintf(stderr, "[rank %d] --transport host needs --shm (the launcher passes it)\n", rank); return 1; }
        CHK(shm_attach(&rs.cm.sh, shm_name));
    }
    if (transport == TR_PEER) CHK(peer_setup(&rs, halo_bytes));
    const double t_create = now_s();
    CHK(make_context(&rs, c0, c1, loc, ids, n_loc, gx, gy));
    /* (every rank has its context before anybody steps: the peer transport's first wait of a step is bounded, and a rank whose
     * context creation lags must not run its neighbours into that bound) */
    HIPCHK(hipStreamSynchronize(rs.st));
    CHK(comm_barrier(&rs.cm));
    int n_local = 0, n_owned = 0;
    SPHCHK(rs.ctx, sph_slab_counts(rs.ctx, &n_local, &n_owned));
    fprintf(stderr, "[rank %d] columns [%d,%d) of %d, lattice columns [%ld,%ld), local/owned %d/%d, created in %.2f s, halo buffers %zu B, device %d, %s transport\n",
            rank, c0, c1, cols, ib, ie, n_local, n_owned, now_s() - t_create, rs.x.halo_bytes, rs.device, transport == TR_RCCL ? "rccl" : transport == TR_PEER ? "peer" : "host");
    if (console && rank == 0) {
        printf("dt = %f    (expected ticks/s) %d\n", prm.dt, (int)(1 / prm.dt));      /* :543 */
        printf("n_fluid = %ld\n", n_total);                                            /* :544 */
        printf("n_boundary = %ld\n", rs.nw);                                           /* :545 */
    }

    /* ---- step loop ---- */
    double t0 = 0, last_reported = now_s();
    float dens0_ms = 0, force0_ms = 0;
    const int pre_reps = getenv("SPH_BENCH_PRE_REPS") ? atoi(getenv("SPH_BENCH_PRE_REPS")) : 300;
    float last_t = 0, worst_rho_err = 0, worst_speed = 0;
    int rebalanced = 0;
    double win_t[65];                 /* the clock at the start of every timed window and at the end of the last */
    int n_win = 0;
    for (int s = 0; s < warmup + windows * steps; s++) {
        if (before_file && s == warmup + windows * steps - 1) {           /* --dump-before: the state (and accelerations) in front of the LAST step */
            HIPCHK(hipStreamSynchronize(rs.st));
            CHK(dump_owned(&rs, before_file, before_accel_file, n_total));
        }
        if (s > warmup && steps > 0 && (s - warmup) % steps == 0) {      /* a window ends, the next begins: every rank, like t0 below */
            HIPCHK(hipStreamSynchronize(rs.st));
            HIPCHK(hipStreamSynchronize(rs.xst));
            CHK(comm_barrier(&rs.cm));
            win_t[++n_win] = now_s();
        }
        if (s == warmup) {
            /* the two heavy kernels at the start of the timed region (back-to-back launches on the live state; again at
             * the end: the JSON line reports their mean) — and enough of them that a fresh GPU has reached its running
             * clocks when a short window (the driver's 20 steps) begins: bench.py does the same at N = 1 */
            if (s > 0 && pre_reps > 0) {      /* (EVERY rank: one-sided work between two steps would leave the others waiting in theirs) */
                HIPCHK(hipStreamSynchronize(rs.st));
                CHK(comm_barrier(&rs.cm));
                SPHCHK(rs.ctx, sph_time_kernel(rs.ctx, SPH_K_DENSITY_EOS, pre_reps, &dens0_ms));
                SPHCHK(rs.ctx, sph_time_kernel(rs.ctx, SPH_K_FORCE_KICK, pre_reps, &force0_ms));
            }
            HIPCHK(hipStreamSynchronize(rs.st));
            HIPCHK(hipStreamSynchronize(rs.xst));
            CHK(comm_barrier(&rs.cm));
            t0 = now_s();
            win_t[0] = t0;
        }
        /* lean step, graphed: everything up to the next thing the host does at a step boundary — the end of the warm-up or of a
         * window, a re-balancing, the end of the run — as ONE call; the gravity source is polled for every step of the run up front,
         * with the time of that step, exactly as the loop below does one step at a time (:632, :678) */
        if (rs.lean_graph && !console && !rs.bd_on) {
            int k = warmup + windows * steps - s;
            if (s < warmup) k = warmup - s;
            else if (steps > 0) { const int in_win = (s - warmup) % steps; if (steps - in_win < k) k = steps - in_win; }
            if (rebalance_every) { const int to_rb = rebalance_every - s % rebalance_every; if (to_rb < k) k = to_rb; }
            if (before_file && s + k == warmup + windows * steps) k--;      /* (the last step on its own: the state in front of it is dumped) */
            if (k > LEAN_RUN_MAX) k = LEAN_RUN_MAX;
            if (k >= 2) {
                static float gseq[2 * LEAN_RUN_MAX];
                for (int j = 0; j < k; j++) {
                    gseq[2 * j] = gx;
                    gseq[2 * j + 1] = gy;
                    t += prm.dt;
                    sph_gravity_sample(&grav, t, &gx, &gy);
                }
                SPHCHK(rs.ctx, sph_slab_steps(rs.ctx, gseq, k));
                s += k - 1;
                if (rebalance_every && (s + 1) % rebalance_every == 0 && s + 1 < warmup + windows * steps) goto rebalance_now;
                continue;
            }
        }
        CHK(step_once(&rs, gx, gy));
        t += prm.dt;                                                             /* :678 */
        if (console && t - last_t > 0.1f) {                                      /* the reference's console line, :679-691 */
            float mx[2] = {0, 0};
            SPHCHK(rs.ctx, sph_stats(rs.ctx, &mx[0], &mx[1]));                   /* :657-671 over the owned particles ... */
            CHK(comm_allreduce(&rs.cm, mx, 2, 1));                               /* ... and the maximum over the ranks */
            const double now = now_s();
            const float rho_err = (mx[0] - prm.rho0) / prm.rho0 * 100;
            if (rho_err > worst_rho_err) worst_rho_err = rho_err;
            if (mx[1] > worst_speed) worst_speed = mx[1];
            if (rank == 0) {
                printf("sim time: %.2f, ", t);
                printf("ticks/s: %d, ", (int)(((t - last_t) / prm.dt) / (now - last_reported)));
                printf("max rho error: %.3f%% (worst) %.3f%%, ", rho_err, worst_rho_err);
                printf("max speed: %.1f m/s (worst) %.1f m/s, ", mx[1], worst_speed);
                printf("\n");
                fflush(stdout);
            }
            last_t = t;
            last_reported = now;
        }
        sph_gravity_sample(&grav, t, &gx, &gy);                                  /* 10 Hz hold, :455-461 */
        if (rebalance_every && (s + 1) % rebalance_every == 0 && s + 1 < warmup + windows * steps) {
rebalance_now:;
            int rc = sph_sync(rs.ctx);                                           /* capacity / out-of-domain / NaN so far */
            if (rc) { fprintf(stderr, "[rank %d] sph_sync before re-balancing: %d (%s)\n", rank, rc, sph_last_error(rs.ctx)); return 1; }
            int moved = 0;
            CHK(rebalance(&rs, gx, gy, 0.05, &moved));
            if (moved) {
                rebalanced++;
                SPHCHK(rs.ctx, sph_slab_counts(rs.ctx, &n_local, &n_owned));
                fprintf(stderr, "[rank %d] re-balanced after step %d: columns [%d,%d), local/owned %d/%d\n", rank, s + 1, rs.c0, rs.c1, n_local, n_owned);
            }
        }
    }
    HIPCHK(hipStreamSynchronize(rs.st));
    HIPCHK(hipStreamSynchronize(rs.xst));
    CHK(comm_barrier(&rs.cm));
    win_t[++n_win] = now_s();
    if (warmup + windows * steps == 0 || steps == 0) { t0 = win_t[0] = win_t[1] = now_s(); n_win = 1; }
    /* the run's rate: the MEDIAN window (SURVEY 8d's protocol: `--windows 5 --steps 1000`; one window: exactly --steps timed steps) */
    double win_rate[64], elapsed = win_t[1] - win_t[0];
    for (int w = 0; w < n_win; w++) win_rate[w] = steps > 0 && win_t[w + 1] > win_t[w] ? (double)steps / (win_t[w + 1] - win_t[w]) : 0.0;
    {
        double sorted[64];
        for (int w = 0; w < n_win; w++) {
            int k = w;
            for (; k > 0 && sorted[k - 1] > win_rate[w]; k--) sorted[k] = sorted[k - 1];
            sorted[k] = win_rate[w];
        }
        /* (an even number of windows: the mean of the two middle ones, not the upper of them) */
        const double med = n_win % 2 ? sorted[n_win / 2] : 0.5 * (sorted[n_win / 2 - 1] + sorted[n_win / 2]);
        if (steps > 0 && med > 0) elapsed = (double)steps / med;
    }
    int rc = sph_sync(rs.ctx);                                                   /* capacity / out-of-domain / NaN */
    if (rc) { fprintf(stderr, "[rank %d] sph_sync: %d (%s)\n", rank, rc, sph_last_error(rs.ctx)); return 1; }

    /* ---- conservation: every particle owned exactly once; statistics; the frame ---- */
    SPHCHK(rs.ctx, sph_slab_counts(rs.ctx, &n_local, &n_owned));
    long long owned_total = n_owned, max_owned_ll = n_owned;
    CHK(comm_allreduce(&rs.cm, &owned_total, 1, 0));
    float fm[3] = {0, 0, (float)n_owned};
    SPHCHK(rs.ctx, sph_stats(rs.ctx, &fm[0], &fm[1]));
    CHK(comm_allreduce(&rs.cm, fm, 3, 1));
    max_owned_ll = (long long)fm[2];
    unsigned char page[1024];
    long long page_ll[1024];
    if (frame_file) {                                                            /* draw_metaballs :380-411 on every slab; the pages OR-ed */
        SPHCHK(rs.ctx, sph_render_metaballs(rs.ctx, page));
        for (int k = 0; k < 1024; k++) page_ll[k] = page[k];                     /* (every pixel belongs to one slab: the sum IS the OR) */
        CHK(comm_allreduce(&rs.cm, page_ll, 1024, 0));
        if (rank == 0) {
            for (int k = 0; k < 1024; k++) page[k] = (unsigned char)page_ll[k];
            FILE *fh = fopen(frame_file, "wb");
            if (!fh || fwrite(page, 1, 1024, fh) != 1024) { fprintf(stderr, "cannot write %s\n", frame_file); return 1; }
            fclose(fh);
        }
    }
    long long rebuilds = 0, direct = 0;
    sph_rebuild_stats(rs.ctx, &rebuilds, &direct);
    /* ---- --breakdown K: K more steps (outside the timed region) with an event at every phase boundary of the step, per rank:
     * where a step's time goes — this rank's own kernels, or waiting for the others (the reduction of the word completes
     * when the SLOWEST rank has contributed, the exchange when both neighbours have sent).  One line of diagnosis for a
     * scaling curve: a rank with a long `begin`/`end` is overloaded (re-balance), long `reduce`/`exchange` everywhere is the
     * interconnect or the collective.  Serial rccl and peer transports (everything on the one stream). ---- */
    double bd[6] = {0, 0, 0, 0, 0, 0};      /* begin, reduce, pack, exchange, end (us per step), steps measured */
    const int bd_ok = breakdown > 0 && ((transport == TR_RCCL && xside == 2) || transport == TR_PEER);
    if (bd_ok) {
        for (int k = 0; k < 6; k++) HIPCHK(hipEventCreate(&rs.bev[k]));
        CHK(comm_barrier(&rs.cm));
        for (int s = 0; s < breakdown; s++) {
            rs.bd_on = 1;
            CHK(step_once(&rs, gx, gy));
            rs.bd_on = 0;
            HIPCHK(hipEventSynchronize(rs.bev[5]));
            for (int k = 0; k < 5; k++) {
                float ms = 0;
                HIPCHK(hipEventElapsedTime(&ms, rs.bev[k], rs.bev[k + 1]));
                bd[k] += 1e3 * (double)ms;
            }
            t += prm.dt;
            sph_gravity_sample(&grav, t, &gx, &gy);
        }
        for (int k = 0; k < 5; k++) bd[k] /= (double)breakdown;
        bd[5] = (double)breakdown;
        rc = sph_sync(rs.ctx);
        if (rc) { fprintf(stderr, "[rank %d] sph_sync after the breakdown steps: %d (%s)\n", rank, rc, sph_last_error(rs.ctx)); return 1; }
    }
    /* every rank's row {begin, reduce, pack, exchange, end, owned, local, device, ranks the communicator counts} -> rank 0 */
    enum { BDW = 9 };
    float *rows = (float *)calloc((size_t)nranks * BDW, sizeof(float));
    if (!rows) return 1;
    {
        int seen = nranks;
        if (transport == TR_RCCL) NCCLCHK(ncclCommCount(rs.cm.nccl, &seen));
        float *mine = rows + (size_t)rank * BDW;
        for (int k = 0; k < 5; k++) mine[k] = (float)bd[k];
        mine[5] = (float)n_owned; mine[6] = (float)n_local; mine[7] = (float)rs.device; mine[8] = (float)seen;
        /* (a sum over rows of which all but one are zero on every rank; floats: counts below 2^24 are exact) */
        long long *tmp = (long long *)calloc((size_t)nranks * BDW, sizeof(long long));
        if (!tmp) return 1;
        for (int k = 0; k < nranks * BDW; k++) tmp[k] = (long long)llround((double)rows[k] * 16.0);
        CHK(comm_allreduce(&rs.cm, tmp, (size_t)nranks * BDW, 0));
        for (int k = 0; k < nranks * BDW; k++) rows[k] = (float)((double)tmp[k] / 16.0);
        free(tmp);
    }
    /* the two heavy kernels of this rank's slab, back to back on the live state (HIP events on the context's stream): what
     * bench.py prices against the HBM roofline */
    float dens_ms = 0, force_ms = 0;
    if (steps + warmup > 0) {
        SPHCHK(rs.ctx, sph_time_kernel(rs.ctx, SPH_K_DENSITY_EOS, 20, &dens_ms));
        SPHCHK(rs.ctx, sph_time_kernel(rs.ctx, SPH_K_FORCE_KICK, 20, &force_ms));
        if (dens0_ms > 0 && force0_ms > 0) { dens_ms = 0.5f * (dens_ms + dens0_ms); force_ms = 0.5f * (force_ms + force0_ms); }
    }
    if (rank == 0) {
        const double tps = steps > 0 ? (double)steps / elapsed : 0.0;
        printf("{\"host\": \"slab_sph_fluid (C, %s%s)\", \"workload\": \"%s%s\", \"n_gpus\": %d, \"n_fluid\": %ld, \"n_boundary\": %ld, "
               "\"steps\": %d, \"warmup\": %d, \"windows\": %d, \"ticks_per_s\": %.2f, \"mparticle_steps_per_s\": %.2f, \"ms_per_step\": %.5f, "
               "\"neighbour_rebuilds\": %lld, \"rebalanced\": %d, \"max_owned\": %lld, \"max_rho\": %.3f, \"max_speed\": %.3f, "
               "\"rank0_local\": %d, \"rank0_owned\": %d, \"rank0_density_ms\": %.5f, \"rank0_force_ms\": %.5f, \"particles_conserved\": %s, "
               "\"halo_buffer_bytes\": %zu, \"breakdown_steps\": %d, \"per_rank\": [",
               transport == TR_RCCL ? "RCCL" : transport == TR_PEER ? "peer-mapped memory" : "host-staged shared memory",
               rs.lean ? (rs.lean_spec == 2 ? (rs.lean_graph ? ", lean step: 3 kernels, speculative, head fused into the density launch, graphs of up to 16 steps" : ", lean step: 3 kernels, speculative, head fused into the density launch")
                          : rs.lean_spec ? (rs.lean_graph ? ", lean step: 4 kernels, speculative, graphs of up to 16 steps" : ", lean step: 4 kernels, speculative")
                                       : (rs.lean_graph ? ", lean step: 4 kernels, graphs of up to 16 steps" : ", lean step: 4 kernels")) : "", sc.label, tilt ? ", scripted tilt gravity" : "", nranks, n_total, rs.nw,
               steps, warmup, n_win, tps, tps * (double)n_total / 1e6, steps > 0 ? elapsed / steps * 1e3 : 0.0, rebuilds, rebalanced, max_owned_ll, fm[0], fm[1],
               n_local, n_owned, dens_ms, force_ms, owned_total == (long long)n_total ? "true" : "false", rs.x.halo_bytes, bd_ok ? breakdown : 0);
        for (int r = 0; r < nranks; r++) {
            const float *q = rows + (size_t)r * BDW;
            printf("%s{\"rank\": %d, \"device\": %d, \"ranks_seen\": %d, \"owned\": %.0f, \"local\": %.0f, \"begin_us\": %.1f, \"reduce_us\": %.1f, "
                   "\"pack_us\": %.1f, \"exchange_us\": %.1f, \"end_us\": %.1f}", r ? ", " : "", r, (int)q[7], (int)q[8], q[5], q[6], q[0], q[1], q[2], q[3], q[4]);
        }
        printf("], \"window_ticks_per_s\": [");
        for (int w = 0; w < n_win; w++) printf("%s%.2f", w ? ", " : "", win_rate[w]);
        printf("]}\n");
        fflush(stdout);
    }
    if (owned_total != (long long)n_total) { fprintf(stderr, "[rank %d] particles lost: %lld of %ld\n", rank, owned_total, n_total); return 1; }

    /* ---- --dump-state [--dump-accel]: every rank's owned particles into one file, by global id ---- */
    if (state_file || accel_file) CHK(dump_owned(&rs, state_file, accel_file, n_total));

    /* ---- --check (one rank): the same run through sph_step on a single context ---- */
    if (check) {
        if (nranks != 1) { fprintf(stderr, "--check needs --ranks 1\n"); return 2; }
        sph_particle *got = (sph_particle *)malloc(sizeof(sph_particle) * (size_t)n_total);
        unsigned *gid = (unsigned *)malloc(sizeof(unsigned) * (size_t)n_total);
        int n_out = 0;
        SPHCHK(rs.ctx, sph_slab_read(rs.ctx, got, gid, NULL, NULL, (int)n_total, &n_out));
        if (n_out != n_total) { fprintf(stderr, "sph_slab_read returned %d of %ld particles\n", n_out, n_total); return 1; }
        sph_ctx *one = NULL;
        sph_gravity g2;
        sph_gravity_init(&g2, tilt ? SPH_GRAVITY_TILT : SPH_GRAVITY_CONSTANT, prm.g);
        float hx, hy, t2 = 0;
        sph_gravity_sample(&g2, 0.0f, &hx, &hy);
        SPHCHK(one, sph_create(&one, &prm, loc, (int)n_total, rs.walls, (int)rs.nw, hx, hy, rs.device));
        for (int s = 0; s < warmup + windows * steps; s++) {
            SPHCHK(one, sph_step(one, hx, hy, 1));
            t2 += prm.dt;
            sph_gravity_sample(&g2, t2, &hx, &hy);
        }
        sph_particle *ref = (sph_particle *)malloc(sizeof(sph_particle) * (size_t)n_total);
        SPHCHK(one, sph_read_particles(one, ref));
        const float dx = fmaxf(max_abs_diff(got, ref, gid, n_total, 0), max_abs_diff(got, ref, gid, n_total, 1));
        const float drho = max_abs_diff(got, ref, gid, n_total, 5);
        const float tol = 1e-6f * (1.0f + sc.box_w) * (float)(1 + (warmup + windows * steps) / 20);      /* ulps of x, growing with the run */
        printf("check: slab path vs sph_step after %d steps: max|dx| = %.3e m (tolerance %.1e), max|drho| = %.3e -> %s\n",
               warmup + windows * steps, dx, tol, drho, dx <= tol ? "ok" : "FAILED");
        sph_destroy(one);
        free(got); free(gid); free(ref);
        if (!(dx <= tol)) return 1;
    }
    if (transport == TR_RCCL) { ncclCommDestroy(rs.cm.nccl); (void)hipFree(rs.cm.d_coll); }
    if (transport == TR_PEER) peer_teardown(&rs);
    sph_destroy(rs.ctx);
    free(loc); free(ids); free(rs.walls); free(cuts);
    return 0;
}
