/* slab_sph_fluid.c — the C multi-GPU host of the MI355X stepper: one process per GPU, x-slab decomposition, halo
 * exchange and rebuild-word reduction over RCCL (xGMI).
 *
 * The reference (pi_sph_fluid.c) has no distributed path: its particle loops (:272, :311) run on one node.  This host
 * shards them by cell column and drives the slab entry points of include/sph.h; per step and rank
 *     sph_slab_step_begin     kick 1/2 + drift of the owned particles (:615-624), may raise the rebuild word
 *     ncclAllReduce(max)      on the device word itself (sph_slab_flag_buffer): all slabs rebuild in the same step
 *     sph_slab_step_pack      fills the send buffers (full records on a rebuild step, x/y/u/v updates otherwise)
 *     ncclGroupStart; ncclSend / ncclRecv with each neighbour; ncclGroupEnd        (sph_slab_buffers, device memory)
 *     sph_slab_step_overlap   density of the tiles that stage no ghost particle, enqueued behind the sends on the same
 *                             stream only when there is no neighbour; with neighbours the exchange runs on a second
 *                             stream and this runs beside it
 *     sph_slab_step_end       ingest + sort + lists | ghost update, density of the rest, force + kick (:626-640)
 * Everything is enqueued on HIP streams; the host synchronises only around the timed region.
 *
 *   slab_sph_fluid --ranks N [--scene dam|cfg3|cfg4] [--block NX NY BOXW BOXH] [--steps K] [--warmup W] [--tilt]
 *                  [--check] [--deterministic] [--skin F]
 * starts N processes (fork + exec of this program with --rank r, before anything touches a GPU), rank r on device r.
 * The ncclUniqueId travels through a file (--id-file, made by the launcher).  --scene dam: N lattice blocks of
 * 4000 x 500 (2 000 000 particles per GPU, box 1200 N x 60 m: the cfg2 -> cfg3 weak-scaling family); cfg3 / cfg4: the
 * fixed 8M / 32M scenes; --tilt: gravity from the scripted tilt trace (sph_gravity, the MPU6050 stand-in), sampled every
 * step with its 0.1 s hold.  Every rank generates only the lattice columns it holds.
 * --check (N = 1): the run is repeated with sph_step on a single context and the two final states are compared.
 */
#define _GNU_SOURCE
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <errno.h>
#include <math.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#include "sph.h"
#include "sph_host.h"

#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "[rank %d] %s: %s\n", g_rank, #call, hipGetErrorString(e_)); return 1; } } while (0)
#define NCCLCHK(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) { fprintf(stderr, "[rank %d] %s: %s\n", g_rank, #call, ncclGetErrorString(r_)); return 1; } } while (0)
#define SPHCHK(ctx, call) do { int rc_ = (call); if (rc_ != SPH_OK) { fprintf(stderr, "[rank %d] %s: %d (%s)\n", g_rank, #call, rc_, (ctx) ? sph_last_error(ctx) : sph_error_string(rc_)); return 1; } } while (0)

static int g_rank = 0;

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

typedef struct {
    float box_w, box_h, x0, y0;
    long nx, ny;
    const char *label;
} scene_t;

/* ---- launcher: N ranks, before anything touches a GPU ---- */
static int launch(int nranks, int argc, char **argv) {
    char idfile[256];
    snprintf(idfile, sizeof idfile, "/tmp/slab_sph_fluid.%d.%ld.id", (int)getpid(), (long)time(NULL));
    unlink(idfile);
    pid_t *pids = (pid_t *)calloc((size_t)nranks, sizeof(pid_t));
    for (int r = 0; r < nranks; r++) {
        pid_t pid = fork();
        if (pid < 0) { perror("fork"); return 1; }
        if (pid == 0) {
            char **av = (char **)calloc((size_t)argc + 6, sizeof(char *));
            char rbuf[16];
            snprintf(rbuf, sizeof rbuf, "%d", r);
            int k = 0;
            for (int i = 0; i < argc; i++) av[k++] = argv[i];
            av[k++] = "--rank"; av[k++] = rbuf; av[k++] = "--id-file"; av[k++] = idfile; av[k] = NULL;
            execv("/proc/self/exe", av);
            perror("execv");
            _exit(127);
        }
        pids[r] = pid;
    }
    int worst = 0, left = nranks;
    while (left > 0) {
        int status = 0;
        pid_t pid = wait(&status);
        if (pid < 0) break;
        left--;
        const int rc = WIFEXITED(status) ? WEXITSTATUS(status) : 128 + (WIFSIGNALED(status) ? WTERMSIG(status) : 0);
        if (rc != 0) {                         /* one rank failed: the others would wait for it forever */
            if (rc > worst) worst = rc;
            for (int r = 0; r < nranks; r++)
                if (pids[r] != pid) kill(pids[r], SIGTERM);
        }
    }
    unlink(idfile);
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

int main(int argc, char **argv) {
    int nranks = 1, rank = -1, steps = 200, warmup = 50, tilt = 0, check = 0, deterministic = 0;
    float skin = -1;
    const char *scene_name = "dam", *idfile = NULL;
    scene_t sc = {0, 0, 0.3f, 0.3f, 0, 0, NULL};
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--ranks") && i + 1 < argc) nranks = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--rank") && i + 1 < argc) rank = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--id-file") && i + 1 < argc) idfile = argv[++i];
        else if (!strcmp(argv[i], "--scene") && i + 1 < argc) scene_name = argv[++i];
        else if (!strcmp(argv[i], "--steps") && i + 1 < argc) steps = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--warmup") && i + 1 < argc) warmup = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--tilt")) tilt = 1;
        else if (!strcmp(argv[i], "--deterministic")) deterministic = 1;
        else if (!strcmp(argv[i], "--skin") && i + 1 < argc) skin = (float)atof(argv[++i]);
        else if (!strcmp(argv[i], "--check")) check = 1;
        else if (!strcmp(argv[i], "--block") && i + 4 < argc) {
            sc.nx = atol(argv[++i]); sc.ny = atol(argv[++i]); sc.box_w = (float)atof(argv[++i]); sc.box_h = (float)atof(argv[++i]);
            sc.label = "custom block";
        }
        else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    if (nranks < 1 || steps < 0 || warmup < 0) return 2;
    if (rank < 0) return launch(nranks, argc, argv);
    g_rank = rank;

    /* ---- scene: parameters and this rank's columns ---- */
    if (!sc.nx) {
        if (!strcmp(scene_name, "dam")) { sc.nx = 4000L * nranks; sc.ny = 500; sc.box_w = 1200.0f * (float)nranks; sc.box_h = 60.0f; sc.label = "dam break, 2 000 000 fluid particles per slab"; }
        else if (!strcmp(scene_name, "cfg3")) { sc.nx = 16000; sc.ny = 500; sc.box_w = 2400.0f; sc.box_h = 60.0f; sc.label = "cfg3: 8M dam break"; }
        else if (!strcmp(scene_name, "cfg4")) { sc.nx = 32000; sc.ny = 1000; sc.box_w = 2400.6f; sc.box_h = 150.0f; sc.label = "cfg4: 32M tank"; }
        else { fprintf(stderr, "unknown scene %s\n", scene_name); return 2; }
    }
    sph_params prm;
    sph_params_default(&prm);
    prm.deterministic = deterministic;
    if (skin >= 0) prm.skin = skin;
    prm.x_max = sc.box_w;
    prm.y_max = sc.box_h;
    int *cuts = (int *)calloc((size_t)nranks + 1, sizeof(int));
    SPHCHK(NULL, sph_slab_partition_block(&prm, sc.x0, sc.nx, sc.ny, nranks, cuts));
    const int c0 = cuts[rank], c1 = cuts[rank + 1];
    long ib = 0, ie = 0;
    SPHCHK(NULL, sph_slab_block_columns(&prm, sc.x0, sc.nx, c0, c1, &ib, &ie));
    const long n_loc = (ie - ib) * sc.ny, n_total = sc.nx * sc.ny;
    sph_particle *loc = (sph_particle *)malloc(sizeof(sph_particle) * (size_t)(n_loc ? n_loc : 1));
    unsigned *ids = (unsigned *)malloc(sizeof(unsigned) * (size_t)(n_loc ? n_loc : 1));
    const long nw = sph_scene_walls(&prm, 0, NULL, 0);
    sph_particle *walls = (sph_particle *)malloc(sizeof(sph_particle) * (size_t)nw);
    if (!loc || !ids || !walls) { fprintf(stderr, "[rank %d] out of host memory\n", rank); return 1; }
    if (sph_scene_block_range(&prm, sc.x0, sc.y0, sc.nx, sc.ny, ib, ie, loc, n_loc) != n_loc) return 1;
    for (long k = 0; k < n_loc; k++) ids[k] = (unsigned)(ib * sc.ny + k);
    sph_scene_walls(&prm, 0, walls, nw);

    sph_gravity grav;
    sph_gravity_init(&grav, tilt ? SPH_GRAVITY_TILT : SPH_GRAVITY_CONSTANT, prm.g);
    float gx, gy, t = 0;
    sph_gravity_sample(&grav, 0.0f, &gx, &gy);

    /* ---- device, streams, slab context ---- */
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        fprintf(stderr, "[rank %d] no HIP device available (this program has no CPU path)\n", rank);
        return 1;
    }
    if (nranks > ndev) {
        fprintf(stderr, "[rank %d] %d ranks need %d GPUs (found %d): RCCL does not share a device between ranks\n", rank, nranks, nranks, ndev);
        return 1;
    }
    const int device = rank;
    HIPCHK(hipSetDevice(device));
    hipStream_t st, xst;                       /* compute stream (adopted by the context) and exchange stream */
    hipEvent_t packed, arrived;
    HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&xst, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&packed, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&arrived, hipEventDisableTiming));
    sph_slab_desc desc = {c0, c1, rank > 0, rank < nranks - 1, 0, 0};
    sph_ctx *ctx = NULL;
    const double t_create = now_s();
    SPHCHK(ctx, sph_create_slab(&ctx, &prm, &desc, loc, ids, (int)n_loc, walls, (int)nw, gx, gy, device));
    SPHCHK(ctx, sph_set_stream(ctx, st));
    void *flag = NULL, *send_l = NULL, *send_r = NULL, *recv_l = NULL, *recv_r = NULL;
    size_t halo_bytes = 0;
    SPHCHK(ctx, sph_slab_flag_buffer(ctx, &flag));
    SPHCHK(ctx, sph_slab_buffers(ctx, &send_l, &send_r, &recv_l, &recv_r, &halo_bytes));
    int n_local = 0, n_owned = 0;
    SPHCHK(ctx, sph_slab_counts(ctx, &n_local, &n_owned));
    fprintf(stderr, "[rank %d] columns [%d,%d) of %d, lattice columns [%ld,%ld), local/owned %d/%d, created in %.2f s, halo buffers %zu B\n",
            rank, c0, c1, sph_slab_grid_columns(&prm), ib, ie, n_local, n_owned, now_s() - t_create, halo_bytes);

    /* ---- communicator ---- */
    ncclUniqueId id;
    ncclComm_t comm;
    if (!idfile || exchange_id(idfile, rank, &id)) { fprintf(stderr, "[rank %d] could not exchange the ncclUniqueId\n", rank); return 1; }
    NCCLCHK(ncclCommInitRank(&comm, nranks, id, rank));
    unsigned long long *d_sum = NULL;          /* owned-particle count for the conservation check */
    HIPCHK(hipMalloc((void **)&d_sum, sizeof *d_sum));

    /* ---- step loop ---- */
    const int has_nb = desc.has_left || desc.has_right;
    double t0 = 0;
    for (int s = 0; s < warmup + steps; s++) {
        if (s == warmup) {
            HIPCHK(hipStreamSynchronize(st));
            HIPCHK(hipStreamSynchronize(xst));
            NCCLCHK(ncclAllReduce(d_sum, d_sum, 1, ncclUint64, ncclSum, comm, st));      /* barrier across ranks */
            HIPCHK(hipStreamSynchronize(st));
            t0 = now_s();
        }
        SPHCHK(ctx, sph_slab_step_begin(ctx, gx, gy));
        if (nranks > 1) NCCLCHK(ncclAllReduce(flag, flag, 1, ncclUint32, ncclMax, comm, st));    /* 4 bytes, on the device word */
        SPHCHK(ctx, sph_slab_step_pack(ctx));
        if (has_nb) {
            /* the exchange on its own stream, behind the pack; the interior density pass runs beside it */
            HIPCHK(hipEventRecord(packed, st));
            HIPCHK(hipStreamWaitEvent(xst, packed, 0));
            NCCLCHK(ncclGroupStart());
            if (desc.has_left) {
                NCCLCHK(ncclSend(send_l, halo_bytes, ncclChar, rank - 1, comm, xst));
                NCCLCHK(ncclRecv(recv_l, halo_bytes, ncclChar, rank - 1, comm, xst));
            }
            if (desc.has_right) {
                NCCLCHK(ncclSend(send_r, halo_bytes, ncclChar, rank + 1, comm, xst));
                NCCLCHK(ncclRecv(recv_r, halo_bytes, ncclChar, rank + 1, comm, xst));
            }
            NCCLCHK(ncclGroupEnd());
            HIPCHK(hipEventRecord(arrived, xst));
            SPHCHK(ctx, sph_slab_step_overlap(ctx));
            HIPCHK(hipStreamWaitEvent(st, arrived, 0));
        }
        SPHCHK(ctx, sph_slab_step_end(ctx));
        t += prm.dt;                                                             /* :678 */
        sph_gravity_sample(&grav, t, &gx, &gy);                                  /* 10 Hz hold, :455-461 */
    }
    HIPCHK(hipStreamSynchronize(st));
    NCCLCHK(ncclAllReduce(d_sum, d_sum, 1, ncclUint64, ncclSum, comm, st));
    HIPCHK(hipStreamSynchronize(st));
    const double elapsed = now_s() - t0;
    int rc = sph_sync(ctx);                                                      /* capacity / out-of-domain / NaN */
    if (rc) { fprintf(stderr, "[rank %d] sph_sync: %d (%s)\n", rank, rc, sph_last_error(ctx)); return 1; }

    /* ---- conservation: every particle owned exactly once ---- */
    SPHCHK(ctx, sph_slab_counts(ctx, &n_local, &n_owned));
    unsigned long long owned = (unsigned long long)n_owned, owned_total = 0;
    HIPCHK(hipMemcpy(d_sum, &owned, sizeof owned, hipMemcpyHostToDevice));
    NCCLCHK(ncclAllReduce(d_sum, d_sum, 1, ncclUint64, ncclSum, comm, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipMemcpy(&owned_total, d_sum, sizeof owned_total, hipMemcpyDeviceToHost));
    long long rebuilds = 0, direct = 0;
    sph_rebuild_stats(ctx, &rebuilds, &direct);
    if (rank == 0) {
        const double tps = steps > 0 ? (double)steps / elapsed : 0.0;
        printf("{\"host\": \"slab_sph_fluid (C, RCCL)\", \"workload\": \"%s%s\", \"n_gpus\": %d, \"n_fluid\": %ld, \"n_boundary\": %ld, "
               "\"steps\": %d, \"warmup\": %d, \"ticks_per_s\": %.2f, \"mparticle_steps_per_s\": %.2f, \"ms_per_step\": %.5f, "
               "\"neighbour_rebuilds\": %lld, \"particles_conserved\": %s}\n",
               sc.label, tilt ? ", scripted tilt gravity" : "", nranks, n_total, nw, steps, warmup, tps, tps * (double)n_total / 1e6,
               steps > 0 ? elapsed / steps * 1e3 : 0.0, rebuilds, owned_total == (unsigned long long)n_total ? "true" : "false");
    }
    if (owned_total != (unsigned long long)n_total) { fprintf(stderr, "[rank %d] particles lost: %llu of %ld\n", rank, owned_total, n_total); return 1; }

    /* ---- --check (one rank): the same run through sph_step on a single context ---- */
    if (check) {
        if (nranks != 1) { fprintf(stderr, "--check needs --ranks 1\n"); return 2; }
        sph_particle *got = (sph_particle *)malloc(sizeof(sph_particle) * (size_t)n_total);
        unsigned *gid = (unsigned *)malloc(sizeof(unsigned) * (size_t)n_total);
        int n_out = 0;
        SPHCHK(ctx, sph_slab_read(ctx, got, gid, NULL, NULL, (int)n_total, &n_out));
        if (n_out != n_total) { fprintf(stderr, "sph_slab_read returned %d of %ld particles\n", n_out, n_total); return 1; }
        sph_ctx *one = NULL;
        sph_gravity g2;
        sph_gravity_init(&g2, tilt ? SPH_GRAVITY_TILT : SPH_GRAVITY_CONSTANT, prm.g);
        float hx, hy, t2 = 0;
        sph_gravity_sample(&g2, 0.0f, &hx, &hy);
        SPHCHK(one, sph_create(&one, &prm, loc, (int)n_total, walls, (int)nw, hx, hy, device));
        for (int s = 0; s < warmup + steps; s++) {
            SPHCHK(one, sph_step(one, hx, hy, 1));
            t2 += prm.dt;
            sph_gravity_sample(&g2, t2, &hx, &hy);
        }
        sph_particle *ref = (sph_particle *)malloc(sizeof(sph_particle) * (size_t)n_total);
        SPHCHK(one, sph_read_particles(one, ref));
        const float dx = fmaxf(max_abs_diff(got, ref, gid, n_total, 0), max_abs_diff(got, ref, gid, n_total, 1));
        const float drho = max_abs_diff(got, ref, gid, n_total, 5);
        const float tol = 1e-6f * (1.0f + sc.box_w) * (float)(1 + (warmup + steps) / 20);      /* ulps of x, growing with the run */
        printf("check: slab path vs sph_step after %d steps: max|dx| = %.3e m (tolerance %.1e), max|drho| = %.3e -> %s\n",
               warmup + steps, dx, tol, drho, dx <= tol ? "ok" : "FAILED");
        sph_destroy(one);
        free(got); free(gid); free(ref);
        if (!(dx <= tol)) return 1;
    }
    ncclCommDestroy(comm);
    sph_destroy(ctx);
    (void)hipFree(d_sum);
    free(loc); free(ids); free(walls); free(cuts);
    return 0;
}
