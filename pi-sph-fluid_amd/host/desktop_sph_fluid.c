/* desktop_sph_fluid.c — the C host of the MI355X stepper.
 *
 * Plays the role of the reference's main() (pi_sph_fluid.c:475-704) for the desktop target
 * (Makefile:18-23): builds the scene, initialises the stepper, then loops
 *     step -> (<= 60 Hz) metaball frame -> statistics line every 0.1 s of simulated time
 * with every physics call of main() replaced by one C-ABI call (include/sph.h).  The SSD1306 panel
 * and the MPU6050 are stubbed: frames go to a 128x64 page-format buffer that can be dumped as
 * text (--show) and gravity comes from sph_gravity (constant, scripted tilt, or the sysfs reader).
 *
 *   desktop_sph_fluid [--scene cfg0|cfg1|cfg2|cfg3|cfg4] [--steps N] [--realtime] [--tilt]
 *                     [--tilt-amp DEG] [--tilt-period S] [--tilt-hold S] [--mpu6050 DIR] [--show] [--batch K]
 *                     [--device D] [--skin F] [--deterministic] [--wall-velocity] [--dump-frame FILE] [--dump-state FILE]
 * --dump-frame: the 1024-byte SSD1306 page-format frame of the final state (what ssd1306_drawBufferFast would be
 * handed, :469); --dump-state: the final fluid[] array (struct particle, 28 bytes each, :26-31).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "sph.h"
#include "sph_host.h"

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);                     /* :551 */
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void show_frame(const unsigned char *buf) {           /* stand-in for ssd1306_drawBufferFast (:469) */
    for (int i = 0; i < 64; i += 2) {                        /* two pixel rows per text row */
        char line[129];
        for (int j = 0; j < 128; j++) {
            int a = (buf[i / 8 * 128 + j] >> (i % 8)) & 1, b = (buf[(i + 1) / 8 * 128 + j] >> ((i + 1) % 8)) & 1;
            line[j] = a && b ? '#' : (a ? '"' : (b ? '_' : ' '));
        }
        line[128] = 0;
        puts(line);
    }
}

static int die(sph_ctx *ctx, const char *what, int rc) {
    fprintf(stderr, "%s failed: %d (%s)\n", what, rc, ctx ? sph_last_error(ctx) : sph_error_string(rc));
    if (ctx) sph_destroy(ctx);
    return 1;
}

int main(int argc, char **argv) {
    const char *scene = "cfg0";
    long max_steps = 0;
    int realtime = 0, show = 0, batch = 1, device = 0, gkind = SPH_GRAVITY_CONSTANT;
    const char *mpu_dir = NULL, *dump_frame = NULL, *dump_state = NULL;
    float tilt_amp = -1, tilt_period = -1, tilt_hold = -1, skin = -1;
    int deterministic = 0, wall_velocity = 0;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--scene") && i + 1 < argc) scene = argv[++i];
        else if (!strcmp(argv[i], "--steps") && i + 1 < argc) max_steps = atol(argv[++i]);
        else if (!strcmp(argv[i], "--batch") && i + 1 < argc) batch = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--realtime")) realtime = 1;        /* the REALTIME define, :10 */
        else if (!strcmp(argv[i], "--show")) show = 1;
        else if (!strcmp(argv[i], "--tilt")) gkind = SPH_GRAVITY_TILT;
        else if (!strcmp(argv[i], "--tilt-amp") && i + 1 < argc) tilt_amp = (float)atof(argv[++i]);
        else if (!strcmp(argv[i], "--tilt-period") && i + 1 < argc) tilt_period = (float)atof(argv[++i]);
        else if (!strcmp(argv[i], "--tilt-hold") && i + 1 < argc) tilt_hold = (float)atof(argv[++i]);
        else if (!strcmp(argv[i], "--skin") && i + 1 < argc) skin = (float)atof(argv[++i]);
        else if (!strcmp(argv[i], "--deterministic")) deterministic = 1;
        else if (!strcmp(argv[i], "--wall-velocity")) wall_velocity = 1;    /* README.md:175-176: infer it from the accelerometer */
        else if (!strcmp(argv[i], "--dump-frame") && i + 1 < argc) dump_frame = argv[++i];
        else if (!strcmp(argv[i], "--dump-state") && i + 1 < argc) dump_state = argv[++i];
        else if (!strcmp(argv[i], "--mpu6050") && i + 1 < argc) { gkind = SPH_GRAVITY_MPU6050; mpu_dir = argv[++i]; }
        else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    if (batch < 1) batch = 1;

    /* ---- scene (:484-540) ---- */
    sph_params prm;
    sph_params_default(&prm);
    prm.deterministic = deterministic;
    if (skin >= 0) prm.skin = prm.skin_min = skin;      /* a fixed skin */
    long n_fluid = 0, n_boundary = 0;
    sph_particle *fluid = NULL, *boundary = NULL;
    int accumulate = 0;
    float bx0 = 0.3f, by0 = 0.3f;
    long bnx = 0, bny = 0;
    float dcx = 0, dcy = 0, drad = 0;
    if (!strcmp(scene, "cfg0")) accumulate = 1;
    else if (!strcmp(scene, "cfg1")) { prm.x_max = 409.6f; prm.y_max = 204.8f; dcx = 204.8f; dcy = 30.95f; drad = 21.665f; }
    else if (!strcmp(scene, "cfg2")) { prm.x_max = 1200.0f; prm.y_max = 60.0f; bnx = 4000; bny = 500; }
    else if (!strcmp(scene, "cfg3")) { prm.x_max = 2400.0f; prm.y_max = 60.0f; bnx = 16000; bny = 500; }
    else if (!strcmp(scene, "cfg4")) { prm.x_max = 2400.6f; prm.y_max = 150.0f; bnx = 32000; bny = 1000; }
    else { fprintf(stderr, "unknown scene %s\n", scene); return 2; }
    if (accumulate) n_fluid = sph_scene_default_fluid(&prm, NULL, 0);
    else if (bnx) n_fluid = sph_scene_block(&prm, bx0, by0, bnx, bny, NULL, 0);
    else n_fluid = sph_scene_disc(&prm, dcx, dcy, drad, NULL, 0);
    n_boundary = sph_scene_walls(&prm, accumulate, NULL, 0);
    fluid = (sph_particle *)malloc(sizeof(sph_particle) * (size_t)n_fluid);
    boundary = (sph_particle *)malloc(sizeof(sph_particle) * (size_t)n_boundary);
    if (!fluid || !boundary) return die(NULL, "malloc", SPH_E_NOMEM);
    if (accumulate) sph_scene_default_fluid(&prm, fluid, n_fluid);
    else if (bnx) sph_scene_block(&prm, bx0, by0, bnx, bny, fluid, n_fluid);
    else sph_scene_disc(&prm, dcx, dcy, drad, fluid, n_fluid);
    sph_scene_walls(&prm, accumulate, boundary, n_boundary);

    printf("dt = %f    (expected ticks/s) %d\n", prm.dt, (int)(1 / prm.dt));      /* :543 */
    printf("n_fluid = %ld\n", n_fluid);                                            /* :544 */
    printf("n_boundary = %ld\n", n_boundary);                                      /* :545 */

    /* ---- gravity source (:555-558) ---- */
    sph_gravity grav;
    sph_gravity_init(&grav, gkind, prm.g);
    if (mpu_dir) { strncpy(grav.sysfs_dir, mpu_dir, sizeof grav.sysfs_dir - 1); }
    if (tilt_amp >= 0) grav.amp_deg = tilt_amp;
    if (tilt_period > 0) grav.period_s = tilt_period;
    if (tilt_hold >= 0) grav.hold_s = tilt_hold;
    float gx, gy;
    sph_wall_motion wmotion;
    sph_wall_motion_init(&wmotion);
    int rc = sph_gravity_sample(&grav, 0.0f, &gx, &gy);
    if (rc) return die(NULL, "sph_gravity_sample", rc);

    /* ---- init (:594-607) ---- */
    sph_ctx *ctx = NULL;
    rc = sph_create(&ctx, &prm, fluid, (int)n_fluid, boundary, (int)n_boundary, gx, gy, device);
    if (rc) return die(ctx, "sph_create", rc);
    int rows, cols;
    sph_grid_dims(ctx, &rows, &cols);
    printf("grid = %d x %d cells, device memory %.1f MB\n", rows, cols, (double)sph_device_bytes(ctx) / 1e6);

    unsigned char *draw_buffer = (unsigned char *)calloc(1024, 1);                 /* :563 */
    float worst_max_rho_error_pct = 0, max_max_speed = 0;                          /* :583 */
    float t = 0, last_t = 0;                                                       /* :584 */
    double now = now_s(), last_reported = now, last_drew = now, last_stepped = now;
    long steps = 0;

    /* ---- main loop (:610-703) ---- */
    while (!max_steps || steps < max_steps) {
        int k = batch;
        if (max_steps && steps + k > max_steps) k = (int)(max_steps - steps);
        rc = sph_step(ctx, gx, gy, k);                                             /* :612-641 */
        if (rc) return die(ctx, "sph_step", rc);
        steps += k;
        t += (float)k * prm.dt;                                                    /* :678 */
        now = now_s();
        if (now - last_drew > 1.0 / 60) {                                          /* :648-651 */
            rc = sph_render_metaballs(ctx, draw_buffer);
            if (rc) return die(ctx, "sph_render_metaballs", rc);
            last_drew = now;
        }
        if (t - last_t > 0.1f) {                                                   /* :679-691 */
            float max_rho, max_speed;
            rc = sph_stats(ctx, &max_rho, &max_speed);                             /* :657-671 (true max; :659 is buggy) */
            if (rc) return die(ctx, "sph_stats", rc);
            rc = sph_sync(ctx);
            if (rc) fprintf(stderr, "warning: %s\n", sph_last_error(ctx));
            now = now_s();
            float max_rho_error_pct = (max_rho - prm.rho0) / prm.rho0 * 100;
            if (max_rho_error_pct > worst_max_rho_error_pct) worst_max_rho_error_pct = max_rho_error_pct;
            if (max_speed > max_max_speed) max_max_speed = max_speed;
            double elapsed = now - last_reported;
            int tps = (int)(((t - last_t) / prm.dt) / elapsed);
            printf("sim time: %.2f, ", t);
            printf("ticks/s: %d, ", tps);
            printf("max rho error: %.3f%% (worst) %.3f%%, ", max_rho_error_pct, worst_max_rho_error_pct);
            printf("max speed: %.1f m/s (worst) %.1f m/s, ", max_speed, max_max_speed);
            printf("\n");
            if (show) show_frame(draw_buffer);
            last_t = t;
            last_reported = now;
        }
        rc = sph_gravity_sample(&grav, t, &gx, &gy);                               /* 10 Hz hold, :455-461 */
        if (rc) return die(ctx, "sph_gravity_sample", rc);
        if (wall_velocity) {                                                       /* README.md:175-176 */
            float wvx, wvy;
            sph_wall_motion_update(&wmotion, gx, gy, (float)k * prm.dt, &wvx, &wvy);
            rc = sph_set_boundary_velocity(ctx, wvx, wvy);
            if (rc) return die(ctx, "sph_set_boundary_velocity", rc);
        }
        if (realtime) {                                                            /* :694-701 */
            sph_sync(ctx);
            do { now = now_s(); } while (now - last_stepped < (double)k * prm.dt - 30e-6);
            last_stepped = now;
        }
    }
    rc = sph_sync(ctx);
    if (rc) fprintf(stderr, "warning: %s\n", sph_last_error(ctx));
    float max_rho, max_speed;
    sph_stats(ctx, &max_rho, &max_speed);
    printf("done: %ld steps, sim time %.3f s, max rho %.2f, max speed %.2f m/s\n", steps, t, max_rho, max_speed);
    if (dump_frame) {                                                              /* the frame of the final state */
        rc = sph_render_metaballs(ctx, draw_buffer);
        if (rc) return die(ctx, "sph_render_metaballs", rc);
        FILE *fh = fopen(dump_frame, "wb");
        if (!fh || fwrite(draw_buffer, 1, 1024, fh) != 1024) { fprintf(stderr, "cannot write %s\n", dump_frame); return 1; }
        fclose(fh);
        if (show) show_frame(draw_buffer);
    }
    if (dump_state) {                                                              /* fluid[] as main() would hold it */
        rc = sph_read_particles(ctx, fluid);
        if (rc) return die(ctx, "sph_read_particles", rc);
        FILE *fh = fopen(dump_state, "wb");
        if (!fh || fwrite(fluid, sizeof(sph_particle), (size_t)n_fluid, fh) != (size_t)n_fluid) { fprintf(stderr, "cannot write %s\n", dump_state); return 1; }
        fclose(fh);
    }
    sph_destroy(ctx);
    free(draw_buffer); free(fluid); free(boundary);
    return 0;
}
