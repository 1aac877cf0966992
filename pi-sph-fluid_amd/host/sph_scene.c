/* sph_scene.c — scene generators (host side, plain C).
 * Mirrors the reference's main() :484-540 and in_initial_shape :238-240; see include/sph_host.h.
 * Built with -ffp-contract=off so lattice coordinates are the same f32 values on every host. */
#include "sph_host.h"

#include <math.h>
#include <string.h>

static void put(sph_particle *out, long i, float x, float y, float m, float rho0) {
    if (!out) return;
    sph_particle q;
    memset(&q, 0, sizeof q);
    q.x = x; q.y = y; q.m = m; q.rho = rho0;
    out[i] = q;
}

static float dist2d(float xi, float yi, float xj, float yj) {     /* euclid_dist :40-43 */
    float dx = xi - xj, dy = yi - yj;
    return sqrtf(dx * dx + dy * dy);
}

long sph_scene_default_fluid(const sph_params *p, sph_particle *out, long cap) {
    if (!p) return SPH_E_ARG;
    const float cx = p->x_min + (p->x_max - p->x_min) / 2, cy = p->y_min + (p->y_max - p->y_min) / 2;
    const float m = p->rho0 * p->vol;                                   /* :502 */
    long n = 0;
    for (float x = p->x_min; x < p->x_max; x += p->r)                   /* :497 */
        for (float y = p->y_min; y < p->y_max; y += p->r)               /* :498 */
            if (dist2d(x, y, cx, cy) < 0.70) {                          /* :239 (double compare) */
                if (out && n >= cap) return SPH_E_ARG;
                put(out, n, x, y, m, p->rho0);
                n++;
            }
    return n;
}

long sph_scene_walls(const sph_params *p, int accumulate, sph_particle *out, long cap) {
    if (!p) return SPH_E_ARG;
    long n = 0;
    if (accumulate) {
        for (float x = p->x_min; x < p->x_max; x += p->r) {             /* :523-531 */
            if (out && n + 2 > cap) return SPH_E_ARG;
            put(out, n, x, p->y_min, 0, p->rho0);
            put(out, n + 1, x, p->y_max, 0, p->rho0);
            n += 2;
        }
        for (float y = p->y_min; y < p->y_max; y += p->r) {             /* :532-540 */
            if (out && n + 2 > cap) return SPH_E_ARG;
            put(out, n, p->x_min, y, 0, p->rho0);
            put(out, n + 1, p->x_max, y, 0, p->rho0);
            n += 2;
        }
    } else {
        for (long i = 0; ; i++) {
            float x = p->x_min + (float)i * p->r;
            if (!(x < p->x_max)) break;
            if (out && n + 2 > cap) return SPH_E_ARG;
            put(out, n, x, p->y_min, 0, p->rho0);
            put(out, n + 1, x, p->y_max, 0, p->rho0);
            n += 2;
        }
        for (long j = 0; ; j++) {
            float y = p->y_min + (float)j * p->r;
            if (!(y < p->y_max)) break;
            if (out && n + 2 > cap) return SPH_E_ARG;
            put(out, n, p->x_min, y, 0, p->rho0);
            put(out, n + 1, p->x_max, y, 0, p->rho0);
            n += 2;
        }
    }
    return n;
}

long sph_scene_walls_layers(const sph_params *p, float wx0, float wx1, float wy0, float wy1, int layers,
                            sph_particle *out, long cap) {
    if (!p || layers < 1 || !(wx1 > wx0) || !(wy1 > wy0)) return SPH_E_ARG;
    const float r = p->r;
    if (wx0 - (float)(layers - 1) * r < p->x_min || wx1 + (float)(layers - 1) * r > p->x_max ||
        wy0 - (float)(layers - 1) * r < p->y_min || wy1 + (float)(layers - 1) * r > p->y_max)
        return SPH_E_ARG;                                   /* frames must stay inside the neighbour grid */
    long n = 0;
    for (int l = 0; l < layers; l++) {
        const float x0 = wx0 - (float)l * r, x1 = wx1 + (float)l * r, y0 = wy0 - (float)l * r, y1 = wy1 + (float)l * r;
        const long nx = (long)floorf((x1 - x0) / r + 0.5f), ny = (long)floorf((y1 - y0) / r + 0.5f);
        /* bottom and top edges including both corners, then the side edges without them */
        for (long i = 0; i <= nx; i++) {
            const float x = i == nx ? x1 : x0 + (float)i * r;
            if (out && n + 2 > cap) return SPH_E_ARG;
            put(out, n, x, y0, 0, p->rho0);
            put(out, n + 1, x, y1, 0, p->rho0);
            n += 2;
        }
        for (long j = 1; j < ny; j++) {
            const float y = y0 + (float)j * r;
            if (out && n + 2 > cap) return SPH_E_ARG;
            put(out, n, x0, y, 0, p->rho0);
            put(out, n + 1, x1, y, 0, p->rho0);
            n += 2;
        }
    }
    return n;
}

long sph_scene_disc(const sph_params *p, float cx, float cy, float radius, sph_particle *out, long cap) {
    if (!p || !(radius > 0)) return SPH_E_ARG;
    const float m = p->rho0 * p->vol;
    long i0 = (long)floorf((cx - radius - p->x_min) / p->r) - 1, i1 = (long)ceilf((cx + radius - p->x_min) / p->r) + 1;
    long j0 = (long)floorf((cy - radius - p->y_min) / p->r) - 1, j1 = (long)ceilf((cy + radius - p->y_min) / p->r) + 1;
    if (i0 < 0) i0 = 0;
    if (j0 < 0) j0 = 0;
    long n = 0;
    for (long i = i0; i <= i1; i++) {
        float x = p->x_min + (float)i * p->r;
        if (!(x < p->x_max)) break;
        for (long j = j0; j <= j1; j++) {
            float y = p->y_min + (float)j * p->r;
            if (!(y < p->y_max)) break;
            if (dist2d(x, y, cx, cy) < radius) {
                if (out && n >= cap) return SPH_E_ARG;
                put(out, n, x, y, m, p->rho0);
                n++;
            }
        }
    }
    return n;
}

long sph_scene_block(const sph_params *p, float x0, float y0, long nx, long ny, sph_particle *out, long cap) {
    return sph_scene_block_range(p, x0, y0, nx, ny, 0, nx, out, cap);
}

long sph_scene_block_range(const sph_params *p, float x0, float y0, long nx, long ny, long i_begin, long i_end,
                           sph_particle *out, long cap) {
    if (!p || nx < 0 || ny < 0 || i_begin < 0 || i_end > nx || i_begin > i_end) return SPH_E_ARG;
    if (!out) return (i_end - i_begin) * ny;
    if ((i_end - i_begin) * ny > cap) return SPH_E_ARG;
    const float m = p->rho0 * p->vol;
    long n = 0;
    for (long i = i_begin; i < i_end; i++) {
        float x = x0 + (float)i * p->r;
        for (long j = 0; j < ny; j++) {
            float y = y0 + (float)j * p->r;
            put(out, n++, x, y, m, p->rho0);
        }
    }
    return n;
}
