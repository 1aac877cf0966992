/* sph_gravity.c — gravity sources (host side, plain C).
 * Mirrors get_gravity :431-445 and the 10 Hz zero-order hold of get_gravity_routine :447-464. */
#include "sph_host.h"

#include <math.h>
#include <stdio.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

void sph_gravity_init(sph_gravity *gs, int kind, float g) {
    memset(gs, 0, sizeof *gs);
    gs->kind = kind;
    gs->g = g;
    gs->amp_deg = 15.0f;      /* cfg4: theta = 15 deg * sin(2 pi t / 8 s) */
    gs->period_s = 8.0f;
    gs->hold_s = 0.1f;        /* 1000000000/10 ns, :459 */
    strcpy(gs->sysfs_dir, "/sys/bus/iio/devices/iio:device0");   /* :436-437 */
    gs->gx = 0; gs->gy = -g;  /* :442-443 */
}

static int read_int_file(const char *dir, const char *name, int *value) {   /* read_file_as_integer :417-428 */
    char path[512];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    int ok = fscanf(f, "%d", value) == 1;
    fclose(f);
    return ok ? 0 : -1;
}

int sph_gravity_sample(sph_gravity *gs, float t, float *gx, float *gy) {
    int due = !gs->primed || (t - gs->last_t) >= gs->hold_s;
    if (due) {
        switch (gs->kind) {
        case SPH_GRAVITY_TILT: {
            double th = gs->amp_deg * (M_PI / 180.0) * sin(2.0 * M_PI * (double)t / gs->period_s);
            gs->gx = (float)(gs->g * sin(th));
            gs->gy = (float)(-gs->g * cos(th));
            break;
        }
        case SPH_GRAVITY_MPU6050: {
            int ax, ay;
            if (read_int_file(gs->sysfs_dir, "in_accel_x_raw", &ax) || read_int_file(gs->sysfs_dir, "in_accel_y_raw", &ay))
                return SPH_E_ARG;
            gs->gx = (float)ay / (1 << 14) * gs->g;      /* :439 */
            gs->gy = -(float)ax / (1 << 14) * gs->g;     /* :440 */
            break;
        }
        default:
            gs->gx = 0; gs->gy = -gs->g;                 /* :442-443 */
        }
        gs->last_t = t;
        gs->primed = 1;
    }
    *gx = gs->gx; *gy = gs->gy;
    return SPH_OK;
}

/* ---- wall velocity from the accelerometer (sph_host.h) ---- */
void sph_wall_motion_init(sph_wall_motion *wm) {
    memset(wm, 0, sizeof *wm);
    wm->tau_tilt = 0.5f;
    wm->tau_leak = 1.0f;
}

void sph_wall_motion_update(sph_wall_motion *wm, float gx, float gy, float dt, float *vx, float *vy) {
    if (!wm->primed) {            /* the first sample is all tilt */
        wm->glx = gx;
        wm->gly = gy;
        wm->primed = 1;
    }
    const float k = wm->tau_tilt > 0 ? fminf(dt / wm->tau_tilt, 1.0f) : 1.0f;
    wm->glx += (gx - wm->glx) * k;
    wm->gly += (gy - wm->gly) * k;
    const float ax = -(gx - wm->glx), ay = -(gy - wm->gly);       /* the box accelerates against the apparent gravity change */
    const float leak = wm->tau_leak > 0 ? expf(-dt / wm->tau_leak) : 0.0f;
    wm->vx = (wm->vx + ax * dt) * leak;
    wm->vy = (wm->vy + ay * dt) * leak;
    *vx = wm->vx;
    *vy = wm->vy;
}
