/* sph_oracle.c — TEST INFRASTRUCTURE ONLY (see sph_oracle.h).
 *
 * Restatement of the reference's 2-D WCSPH step.  Every function cites the
 * lines of /root/reference/pi_sph_fluid.c it follows.  Arithmetic mirrors the
 * reference operation by operation, including the places where a C double
 * literal promotes a sub-expression to double (SURVEY.md §8a "double
 * promotion"), and the neighbour summation order (cell rows outer, cell
 * columns inner, ascending particle index inside a cell), so that an `-O2`
 * build is bit-identical to an `-O2` build of the reference.
 *
 * Differences on purpose: int32 indices (-1 = end of list) instead of
 * unsigned short/USHRT_MAX (:78-79,:107); run-time parameters instead of
 * macros (:11-20); a per-thread neighbour scratch that is bounds-checked
 * (the reference's 48-entry one is not, :145); out-of-domain particles are
 * reported instead of corrupting the heap (:111-116).
 */
#include "sph_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#define ORC_SCRATCH 512   /* neighbour scratch per thread (reference: 48, :21) */

/* ---- derived constants, evaluated the way the reference's macros are ---- */
typedef struct {
    float h, two_h, cell;       /* H :12 ; 2*H :144 ; cell length 2*H :596 */
    float nf;                   /* 7/(4*M_PI*H*H), double expression rounded to f32 :46 */
    float w_k2h;                /* W(0.2*H,0,0,0) :325 */
    float rho0, c, dt, B;       /* B = C*C*RHO_0/7 :297 */
    double eps_h2;              /* 0.01*H*H in double :332 */
    double neg_alpha_c;         /* -0.01*C in double :334 */
    double k1;                  /* 0.1 :325 */
    double half_dt;             /* 0.5*DT in double :616 */
} consts;

static float dist2d(float xi, float yi, float xj, float yj) {          /* euclid_dist :40-43 */
    float dx = xi - xj, dy = yi - yj;
    return sqrtf(dx*dx + dy*dy);
}

static float kernel_w(const consts *k, float xi, float yi, float xj, float yj) {   /* W :45-50 */
    float q = dist2d(xi, yi, xj, yj) / k->h;
    float a = 1 - 0.5f*q, b = 1 + 2*q;
    return k->nf * powf(a, 4) * b;
}

static void kernel_grad(const consts *k, float xi, float yi, float xj, float yj, float *gx, float *gy) { /* grad_a_W_ab :52-62 */
    float q = dist2d(xi, yi, xj, yj) / k->h;
    float a = 1 - 0.5f*q;
    float dw_dq = k->nf * (-5) * q * powf(a, 3);
    float dq_dx = (xi - xj) / dist2d(xi, yi, xj, yj) / k->h;
    float dq_dy = (yi - yj) / dist2d(xi, yi, xj, yj) / k->h;
    *gx = dw_dq * dq_dx;
    *gy = dw_dq * dq_dy;
}

static void make_consts(const orc_params *p, consts *k) {
    k->h = p->h;
    k->two_h = 2 * p->h;
    k->cell = 2 * p->h;
    k->nf = (float)(7 / (4 * M_PI * p->h * p->h));
    k->rho0 = p->rho0; k->c = p->c; k->dt = p->dt;
    k->B = p->c * p->c * p->rho0 / 7;
    k->eps_h2 = p->eps * p->h * p->h;
    k->neg_alpha_c = -p->alpha * p->c;
    k->k1 = p->k1;
    k->half_dt = 0.5 * p->dt;
    k->w_k2h = kernel_w(k, (float)(p->k2 * p->h), 0, 0, 0);
}

void orc_params_default(orc_params *p) {            /* :11-20, :595 */
    p->r = 0.0750f;
    p->h = p->r * 1.3f;
    p->rho0 = 1000.0f;
    p->c = 400.0f;
    p->g = 9.81f;
    p->dt = 1.0f * p->h / p->c;
    p->vol = 0.57f * p->h * p->h;
    p->x_min = 0; p->x_max = 4.0f; p->y_min = 0; p->y_max = 2.0f;
    p->alpha = 0.01; p->eps = 0.01; p->k1 = 0.1; p->k2 = 0.2;
}

void orc_constants(const orc_params *p, float *out) {
    consts k; make_consts(p, &k);
    out[0] = p->r; out[1] = p->h; out[2] = p->x_max - p->x_min; out[3] = p->y_max - p->y_min;
    out[4] = p->rho0; out[5] = p->c; out[6] = p->g; out[7] = p->dt; out[8] = p->vol;
    out[9] = p->rho0 * p->vol; out[10] = k.B; out[11] = kernel_w(&k, 0, 0, 0, 0);
    out[12] = k.w_k2h; out[13] = k.cell; out[14] = (float)k.eps_h2; out[15] = (float)ORC_SCRATCH;
}

/* ---- cell linked list (:73-124) ---- */
typedef struct {
    float x_min, y_min, cell;
    int rows, cols, n;          /* n_cells (y), m_cells (x) :93-94 */
    int *head, *tail, *next;
} grid;

static void grid_alloc(grid *g, const orc_params *p, const consts *k, int n) {     /* :82-102 */
    g->x_min = p->x_min; g->y_min = p->y_min; g->cell = k->cell;
    g->rows = (int)((p->y_max - p->y_min) / k->cell) + 1;
    g->cols = (int)((p->x_max - p->x_min) / k->cell) + 1;
    g->n = n;
    g->head = (int*)malloc(sizeof(int) * (size_t)g->rows * g->cols);
    g->tail = (int*)malloc(sizeof(int) * (size_t)g->rows * g->cols);
    g->next = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
}
static void grid_free(grid *g) { free(g->head); free(g->tail); free(g->next); }

/* :104-124; returns the number of particles outside the grid (those are left
 * out of the lists; the reference would write out of bounds) */
static int grid_build(grid *g, const orc_particle *ps) {
    size_t nc = (size_t)g->rows * g->cols;
    for (size_t c = 0; c < nc; c++) g->head[c] = g->tail[c] = -1;
    int bad = 0;
    for (int i = 0; i < g->n; i++) {
        int row = (int)((ps[i].y - g->y_min) / g->cell);
        int col = (int)((ps[i].x - g->x_min) / g->cell);
        g->next[i] = -1;
        if (row < 0 || row >= g->rows || col < 0 || col >= g->cols || ps[i].x != ps[i].x || ps[i].y != ps[i].y) { bad++; continue; }
        int c = row * g->cols + col;
        if (g->head[c] < 0) g->head[c] = g->tail[c] = i;
        else { g->next[g->tail[c]] = i; g->tail[c] = i; }
    }
    return bad;
}

int orc_grid_dims(const orc_params *p, int *n_cells, int *m_cells) {
    consts k; make_consts(p, &k);
    *n_cells = (int)((p->y_max - p->y_min) / k.cell) + 1;
    *m_cells = (int)((p->x_max - p->x_min) / k.cell) + 1;
    return 0;
}

/* neighbours of point (x,y) in set b; `self` = index to reject or -1 (:126-153).
 * Returns the count, or -1 when the scratch would overflow. */
static int find_nbrs(const consts *k, const grid *g, const orc_particle *b, float x, float y, int self, int *out) {
    int n = 0;
    int row0 = (int)((y - g->y_min) / g->cell), col0 = (int)((x - g->x_min) / g->cell);
    for (int row = row0 - 1; row <= row0 + 1; row++) {
        for (int col = col0 - 1; col <= col0 + 1; col++) {
            if (row < 0 || row >= g->rows || col < 0 || col >= g->cols) continue;
            for (int j = g->head[row * g->cols + col]; j >= 0; j = g->next[j]) {
                float d = dist2d(x, y, b[j].x, b[j].y);
                if (d < k->two_h && j != self) {
                    if (n == ORC_SCRATCH) return -1;
                    out[n++] = j;
                }
            }
        }
    }
    return n;
}

/* the reference's AoS -> SoA neighbour gather (:157-182) */
typedef struct {
    int count;
    float x[ORC_SCRATCH], y[ORC_SCRATCH], u[ORC_SCRATCH], v[ORC_SCRATCH], m[ORC_SCRATCH], rho[ORC_SCRATCH], p[ORC_SCRATCH];
} nbr_soa;

static void gather(const orc_particle *ps, const int *idx, int n, nbr_soa *s) {
    s->count = n;
    for (int k = 0; k < n; k++) {
        const orc_particle *q = &ps[idx[k]];
        s->x[k] = q->x; s->y[k] = q->y; s->u[k] = q->u; s->v[k] = q->v; s->m[k] = q->m; s->rho[k] = q->rho; s->p[k] = q->p;
    }
}

/* Σ m_j * quantity_k * W_ij (:200-214, MASS branch — the only one any caller uses) */
static float sph_sum(const consts *k, const float *quantity, float xi, float yi, const nbr_soa *s) {
    float acc = 0;
    for (int n = 0; n < s->count; n++) acc += s->m[n] * quantity[n] * kernel_w(k, xi, yi, s->x[n], s->y[n]);
    return acc;
}

/* Σ m_j * quantity_k * grad_i W_ij (:216-231); optionally Σ |term| for gate G3 */
static void sph_grad(const consts *k, const float *quantity, float xi, float yi, const nbr_soa *s, float *gx, float *gy, double *abs_sum) {
    float ax = 0, ay = 0;
    for (int n = 0; n < s->count; n++) {
        float wx, wy;
        kernel_grad(k, xi, yi, s->x[n], s->y[n], &wx, &wy);
        float tx = s->m[n] * quantity[n] * wx, ty = s->m[n] * quantity[n] * wy;
        ax += tx; ay += ty;
        if (abs_sum) *abs_sum += sqrt((double)tx * tx + (double)ty * ty);
    }
    *gx = ax; *gy = ay;
}

/* ---- physics passes ---- */

/* Akinci pseudo-mass, :242-261 (self excluded because both sets are the same array, :130) */
int orc_psi(const orc_params *p, orc_particle *boundary, int n_boundary) {
    consts k; make_consts(p, &k);
    grid g; grid_alloc(&g, p, &k, n_boundary);
    int bad = grid_build(&g, boundary);
    int overflow = 0;
    int *idx = (int*)malloc(sizeof(int) * ORC_SCRATCH);
    nbr_soa *s = (nbr_soa*)malloc(sizeof(nbr_soa));
    float *psi = (float*)malloc(sizeof(float) * (size_t)(n_boundary > 0 ? n_boundary : 1));
    for (int i = 0; i < n_boundary; i++) {
        int n = find_nbrs(&k, &g, boundary, boundary[i].x, boundary[i].y, i, idx);
        if (n < 0) { overflow = 1; n = 0; }
        gather(boundary, idx, n, s);
        float recip_volume = 0;
        for (int q = 0; q < n; q++) recip_volume += kernel_w(&k, boundary[i].x, boundary[i].y, s->x[q], s->y[q]);
        psi[i] = boundary[i].rho / recip_volume;
    }
    /* the reference writes .m in place while later particles still read .x/.y only — same result */
    for (int i = 0; i < n_boundary; i++) boundary[i].m = psi[i];
    free(psi); free(s); free(idx); grid_free(&g);
    return overflow ? -2 : (bad ? -3 : 0);
}

typedef struct { int *idx; nbr_soa *s; float *tmp; float *ones; } scratch;
static void scratch_alloc(scratch *w) {
    w->idx = (int*)malloc(sizeof(int) * ORC_SCRATCH);
    w->s = (nbr_soa*)malloc(sizeof(nbr_soa));
    w->tmp = (float*)malloc(sizeof(float) * ORC_SCRATCH);
    w->ones = (float*)malloc(sizeof(float) * ORC_SCRATCH);
    for (int i = 0; i < ORC_SCRATCH; i++) w->ones[i] = 1;
}
static void scratch_free(scratch *w) { free(w->idx); free(w->s); free(w->tmp); free(w->ones); }

/* :263-289 — orphaned work-sharing loop, to be entered by a whole team */
static void pass_density(const consts *k, orc_particle *fluid, const orc_particle *boundary, const grid *gf, const grid *gb, scratch *w, int *overflow) {
    #pragma omp for
    for (int i = 0; i < gf->n; i++) {
        const float w_ii = kernel_w(k, 0, 0, 0, 0);
        float self = fluid[i].m * w_ii;
        int n = find_nbrs(k, gf, fluid, fluid[i].x, fluid[i].y, i, w->idx);
        if (n < 0) { *overflow = 1; n = 0; }
        gather(fluid, w->idx, n, w->s);
        float ff = sph_sum(k, w->ones, fluid[i].x, fluid[i].y, w->s);
        n = find_nbrs(k, gb, boundary, fluid[i].x, fluid[i].y, -1, w->idx);
        if (n < 0) { *overflow = 1; n = 0; }
        gather(boundary, w->idx, n, w->s);
        float fb = sph_sum(k, w->ones, fluid[i].x, fluid[i].y, w->s);
        fluid[i].rho = self + ff + fb;
    }
}

/* :294-301 */
static void pass_pressure(const consts *k, orc_particle *fluid, int n) {
    #pragma omp for
    for (int i = 0; i < n; i++) {
        float pr = k->B * (powf(fluid[i].rho / k->rho0, 7) - 1);
        fluid[i].p = (pr > 0) ? pr : 0;
    }
}

/* :303-373 */
static void pass_accel(const consts *k, float *du_dt, float *dv_dt, const orc_particle *fluid, const orc_particle *boundary,
                       const grid *gf, const grid *gb, float gx, float gy, scratch *w, float *sum_abs, int *overflow) {
    #pragma omp for
    for (int i = 0; i < gf->n; i++) {
        const orc_particle pi = fluid[i];
        double abs_acc = 0;
        int n = find_nbrs(k, gf, fluid, pi.x, pi.y, i, w->idx);
        if (n < 0) { *overflow = 1; n = 0; }
        gather(fluid, w->idx, n, w->s);
        for (int q = 0; q < n; q++) {
            const nbr_soa *s = w->s;
            float pressure_ij = (pi.p / (pi.rho * pi.rho) + s->p[q] / (s->rho[q] * s->rho[q]));           /* :321 */
            float w_ij = kernel_w(k, pi.x, pi.y, s->x[q], s->y[q]);                                        /* :324 */
            float artificial_ij = k->k1 * powf(w_ij / k->w_k2h, 4);                                        /* :325 (double product) */
            float u_ij = pi.u - s->u[q], v_ij = pi.v - s->v[q];                                            /* :328 */
            float x_ij = pi.x - s->x[q], y_ij = pi.y - s->y[q];                                            /* :329 */
            float xv = x_ij * u_ij + y_ij * v_ij;                                                          /* :330 */
            float xx = x_ij * x_ij + y_ij * y_ij;                                                          /* :331 */
            float mu_ij = k->h * xv / (xx + k->eps_h2);                                                    /* :332 (double division) */
            float mean_rho = (pi.rho + s->rho[q]) / 2;                                                     /* :333 */
            float viscosity_ij = (xv < 0) ? k->neg_alpha_c * mu_ij / mean_rho : 0;                         /* :334 (double) */
            w->tmp[q] = pressure_ij + artificial_ij + viscosity_ij;                                        /* :336 */
        }
        float ffx, ffy;
        sph_grad(k, w->tmp, pi.x, pi.y, w->s, &ffx, &ffy, sum_abs ? &abs_acc : 0);                         /* :340 */

        n = find_nbrs(k, gb, boundary, pi.x, pi.y, -1, w->idx);                                            /* :343 */
        if (n < 0) { *overflow = 1; n = 0; }
        gather(boundary, w->idx, n, w->s);
        for (int q = 0; q < n; q++) {
            const nbr_soa *s = w->s;
            float pressure_ij = pi.p / (pi.rho * pi.rho);                                                  /* :350 */
            float w_ij = kernel_w(k, pi.x, pi.y, s->x[q], s->y[q]);
            float artificial_ij = k->k1 * powf(w_ij / k->w_k2h, 4);                                        /* :354 */
            float u_ij = pi.u - s->u[q], v_ij = pi.v - s->v[q];
            float x_ij = pi.x - s->x[q], y_ij = pi.y - s->y[q];
            float xv = x_ij * u_ij + y_ij * v_ij;
            float xx = x_ij * x_ij + y_ij * y_ij;
            float mu_ij = k->h * xv / (xx + k->eps_h2);                                                    /* :361 */
            float viscosity_ij = (xv < 0) ? k->neg_alpha_c * mu_ij / pi.rho : 0;                           /* :362 fluid density only */
            w->tmp[q] = pressure_ij + artificial_ij + viscosity_ij;
        }
        float fbx, fby;
        sph_grad(k, w->tmp, pi.x, pi.y, w->s, &fbx, &fby, sum_abs ? &abs_acc : 0);                         /* :368 */

        du_dt[i] = gx - ffx - fbx;                                                                         /* :370 */
        dv_dt[i] = gy - ffy - fby;                                                                         /* :371 */
        if (sum_abs) sum_abs[i] = (float)abs_acc;
    }
}

int orc_max_neighbors(const orc_params *p, const orc_particle *fluid, int n_fluid,
                      const orc_particle *boundary, int n_boundary, int *max_ff, int *max_fb) {
    consts k; make_consts(p, &k);
    grid gf, gb; grid_alloc(&gf, p, &k, n_fluid); grid_alloc(&gb, p, &k, n_boundary);
    grid_build(&gf, fluid); grid_build(&gb, boundary);
    int mff = 0, mfb = 0, overflow = 0;
    #pragma omp parallel reduction(max:mff) reduction(max:mfb) reduction(|:overflow)
    {
        int *idx = (int*)malloc(sizeof(int) * ORC_SCRATCH);
        #pragma omp for
        for (int i = 0; i < n_fluid; i++) {
            int a = find_nbrs(&k, &gf, fluid, fluid[i].x, fluid[i].y, i, idx);
            int b = find_nbrs(&k, &gb, boundary, fluid[i].x, fluid[i].y, -1, idx);
            if (a < 0 || b < 0) overflow = 1;
            if (a > mff) mff = a;
            if (b > mfb) mfb = b;
        }
        free(idx);
    }
    *max_ff = mff; *max_fb = mfb;
    grid_free(&gf); grid_free(&gb);
    return overflow ? -2 : 0;
}

/* :604-607 stage by stage */
int orc_eval(const orc_params *p, orc_particle *fluid, int n_fluid, const orc_particle *boundary, int n_boundary,
             float gx, float gy, int flags, float *du_dt, float *dv_dt, float *sum_abs, int threads) {
    consts k; make_consts(p, &k);
    grid gf, gb; grid_alloc(&gf, p, &k, n_fluid); grid_alloc(&gb, p, &k, n_boundary);
    int bad = grid_build(&gb, boundary) + grid_build(&gf, fluid);
    int overflow = 0;
    if (threads < 1) threads = 1;
    #pragma omp parallel num_threads(threads)
    {
        scratch w; scratch_alloc(&w);
        if (flags & 1) pass_density(&k, fluid, boundary, &gf, &gb, &w, &overflow);
        if (flags & 2) pass_pressure(&k, fluid, n_fluid);
        if (flags & 4) pass_accel(&k, du_dt, dv_dt, fluid, boundary, &gf, &gb, gx, gy, &w, (flags & 8) ? sum_abs : 0, &overflow);
        scratch_free(&w);
    }
    grid_free(&gf); grid_free(&gb);
    return overflow ? -2 : (bad ? -3 : 0);
}

/* the main loop body :612-641, nsteps times; du_dt/dv_dt in = accelerations of
 * the input state, out = accelerations of the output state */
int orc_steps(const orc_params *p, orc_particle *fluid, int n_fluid, const orc_particle *boundary, int n_boundary,
              float gx, float gy, float *du_dt, float *dv_dt, int nsteps, int threads) {
    consts k; make_consts(p, &k);
    grid gf, gb; grid_alloc(&gf, p, &k, n_fluid); grid_alloc(&gb, p, &k, n_boundary);
    int bad = grid_build(&gb, boundary);
    int overflow = 0;
    if (threads < 1) threads = 1;
    #pragma omp parallel num_threads(threads)
    {
        scratch w; scratch_alloc(&w);
        for (int s = 0; s < nsteps; s++) {
            #pragma omp single
            {
                for (int i = 0; i < n_fluid; i++) {                 /* kick :615-618 (double product, rounded on store) */
                    fluid[i].u += k.half_dt * du_dt[i];
                    fluid[i].v += k.half_dt * dv_dt[i];
                }
                for (int i = 0; i < n_fluid; i++) {                 /* drift :621-624 */
                    fluid[i].x += k.dt * fluid[i].u;
                    fluid[i].y += k.dt * fluid[i].v;
                }
                bad += grid_build(&gf, fluid);                      /* :626 */
            }
            pass_density(&k, fluid, boundary, &gf, &gb, &w, &overflow);                               /* :630 */
            pass_pressure(&k, fluid, n_fluid);                                                        /* :631 */
            pass_accel(&k, du_dt, dv_dt, fluid, boundary, &gf, &gb, gx, gy, &w, 0, &overflow);        /* :632 */
            #pragma omp single
            {
                for (int i = 0; i < n_fluid; i++) {                 /* kick :637-640 */
                    fluid[i].u += k.half_dt * du_dt[i];
                    fluid[i].v += k.half_dt * dv_dt[i];
                }
            }
        }
        scratch_free(&w);
    }
    grid_free(&gf); grid_free(&gb);
    return overflow ? -2 : (bad ? -3 : 0);
}

/* ---- default scene: main() :484-540 ---- */
static int in_initial_shape(const orc_params *p, float x, float y) {     /* :238-240 */
    float w = p->x_max - p->x_min, h = p->y_max - p->y_min;
    return dist2d(x, y, w / 2, h / 2) < 0.70;
}

int orc_scene_default(const orc_params *p, orc_particle **fluid_out, int *n_fluid_out,
                      orc_particle **boundary_out, int *n_boundary_out) {
    const float W_ = p->x_max - p->x_min, H_ = p->y_max - p->y_min, R_ = p->r;
    int nf = 0, nb = 0;
    /* lattice by f32 accumulation from 0, x outer / y inner (:486-488) */
    for (float x = 0; x < W_; x += R_) for (float y = 0; y < H_; y += R_) if (in_initial_shape(p, x, y)) nf++;
    for (float x = 0; x < W_; x += R_) nb += 2;          /* :515 */
    for (float y = 0; y < H_; y += R_) nb += 2;          /* :516 */
    orc_particle *f = (orc_particle*)calloc((size_t)nf, sizeof(orc_particle));
    orc_particle *b = (orc_particle*)calloc((size_t)nb, sizeof(orc_particle));
    int c = 0;
    for (float x = 0; x < W_; x += R_)
        for (float y = 0; y < H_; y += R_)
            if (in_initial_shape(p, x, y)) {
                f[c].x = x; f[c].y = y; f[c].m = p->rho0 * p->vol; f[c].rho = p->rho0;   /* :500-502 */
                c++;
            }
    c = 0;
    for (float x = 0; x < W_; x += R_) {                 /* floor and ceiling :523-531 */
        b[c].x = x; b[c].y = 0;  b[c].rho = p->rho0;
        b[c+1].x = x; b[c+1].y = H_; b[c+1].rho = p->rho0;
        c += 2;
    }
    for (float y = 0; y < H_; y += R_) {                 /* left and right walls :532-540 */
        b[c].x = 0;  b[c].y = y; b[c].rho = p->rho0;
        b[c+1].x = W_; b[c+1].y = y; b[c+1].rho = p->rho0;
        c += 2;
    }
    *fluid_out = f; *n_fluid_out = nf; *boundary_out = b; *n_boundary_out = nb;
    return 0;
}

void orc_free(void *ptr) { free(ptr); }

void orc_first_touch_copy(void *dst, const void *src, long n, int rec, int threads) {
    if (threads < 1) threads = 1;
    #pragma omp parallel for num_threads(threads) schedule(static)
    for (long i = 0; i < n; i++) memcpy((char*)dst + (size_t)i * (size_t)rec, (const char*)src + (size_t)i * (size_t)rec, (size_t)rec);
}

/* ---- metaballs :380-411 with the pixel grid of :570-577 ---- */
int orc_metaballs(const orc_params *p, unsigned char *draw_buffer, const orc_particle *fluid, int n_fluid, int threads) {
    consts k; make_consts(p, &k);
    const float W_ = p->x_max - p->x_min, H_ = p->y_max - p->y_min;
    grid gf; grid_alloc(&gf, p, &k, n_fluid);
    grid_build(&gf, fluid);
    int overflow = 0;
    if (threads < 1) threads = 1;
    const float px_width = W_ / 128;                                   /* :399 */
    const float w_half_px = kernel_w(&k, px_width / 2, 0, 0, 0);       /* :401 */
    unsigned char bits[64 * 128];
    #pragma omp parallel num_threads(threads)
    {
        int *idx = (int*)malloc(sizeof(int) * ORC_SCRATCH);
        #pragma omp for collapse(2)
        for (int i = 0; i < 64; i++) {
            for (int j = 0; j < 128; j++) {
                float px = (j + 0.5) * W_ / 128, py = (64 - (i + 0.5)) * H_ / 64;     /* :573 (double) */
                int n = find_nbrs(&k, &gf, fluid, px, py, -1, idx);
                if (n < 0) { overflow = 1; n = 0; }
                float cond = 0;
                for (int q = 0; q < n; q++) {
                    cond += kernel_w(&k, px, py, fluid[idx[q]].x, fluid[idx[q]].y) / w_half_px;
                    if (cond >= 1) break;                              /* :403 */
                }
                bits[i * 128 + j] = (cond >= 1);
            }
        }
        free(idx);
    }
    for (int i = 0; i < 64; i++)                                        /* page format :407-408 */
        for (int j = 0; j < 128; j++) {
            if (bits[i * 128 + j]) draw_buffer[i / 8 * 128 + j] |= (unsigned char)(1 << (i % 8));
            else draw_buffer[i / 8 * 128 + j] &= (unsigned char)~(1 << (i % 8));
        }
    grid_free(&gf);
    return overflow ? -2 : 0;
}
