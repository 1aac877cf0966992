/* sph_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (C99 + OpenMP) of the per-step compute of
 * colonelwatch/pi-sph-fluid (/root/reference/pi_sph_fluid.c:10-373 and the loop
 * body :612-641), with 32-bit indices and run-time parameters so that it also
 * runs the configurations the reference cannot (> 65 534 particles, other
 * boxes).  It is the checker for the HIP path and the timed "port" CPU
 * baseline of bench.py.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product never does.
 *
 * Parity status: PINNED.  Built with `-O2` (no -march, no fast-math) it is
 * bit-identical to the real reference built the same way (oracle/build_ref.sh,
 * oracle/_ref/libpisph_ref_strict.so) on every fixture of tests/golden/
 * (tests/test_oracle_golden.py, tests/test_oracle_vs_ref.py).
 */
#ifndef SPH_ORACLE_H
#define SPH_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* byte-compatible with the reference's `struct particle` (pi_sph_fluid.c:26-31) */
typedef struct { float x, y, u, v, m, rho, p; } orc_particle;

/* the reference's compile-time constants (pi_sph_fluid.c:11-20) as run-time
 * fields; alpha/eps/k1/k2 are double because the reference writes them as
 * double literals inside f32 expressions (:325, :332, :334). */
typedef struct {
    float r;        /* R     :11 */
    float h;        /* H     :12 */
    float rho0;     /* RHO_0 :15 */
    float c;        /* C     :16 */
    float g;        /* G     :17 */
    float dt;       /* DT    :19 */
    float vol;      /* V     :20 */
    float x_min, x_max, y_min, y_max;   /* domain box :595 */
    double alpha;   /* 0.01  :334 */
    double eps;     /* 0.01  :332 */
    double k1;      /* 0.1   :325 */
    double k2;      /* 0.2   :325 */
} orc_params;

void orc_params_default(orc_params *p);                 /* reference defaults, box 4 x 2 */
void orc_constants(const orc_params *p, float *out16);  /* same slots as ref_constants() */
int  orc_grid_dims(const orc_params *p, int *n_cells, int *m_cells);

int  orc_scene_default(const orc_params *p, orc_particle **fluid, int *n_fluid,
                       orc_particle **boundary, int *n_boundary);
void orc_free(void *ptr);

int  orc_psi(const orc_params *p, orc_particle *boundary, int n_boundary);
int  orc_max_neighbors(const orc_params *p, const orc_particle *fluid, int n_fluid,
                       const orc_particle *boundary, int n_boundary, int *max_ff, int *max_fb);
/* flags: bit0 density, bit1 pressure, bit2 acceleration, bit3 also return the
 * per-particle sum of |m_j*temp_ij*gradW_ij| (gate G3's scale) in sum_abs */
int  orc_eval(const orc_params *p, orc_particle *fluid, int n_fluid, const orc_particle *boundary, int n_boundary,
              float gx, float gy, int flags, float *du_dt, float *dv_dt, float *sum_abs, int threads);
int  orc_steps(const orc_params *p, orc_particle *fluid, int n_fluid, const orc_particle *boundary, int n_boundary,
               float gx, float gy, float *du_dt, float *dv_dt, int nsteps, int threads);
/* 128x64 1-bpp SSD1306 page-format bitmap (1024 bytes), pi_sph_fluid.c:380-411 + :570-577 */
int  orc_metaballs(const orc_params *p, unsigned char *draw_buffer, const orc_particle *fluid, int n_fluid, int threads);
/* timing aid (bench.py's cpu_baseline): copy `n` records of `rec` bytes with the static schedule of the physics loops, so that
 * every page of a freshly allocated destination is first touched by the thread that will work on it (NUMA first touch; the
 * reference gets the same from its own serial initialisation only by luck of the allocator: no counterpart) */
void orc_first_touch_copy(void *dst, const void *src, long n, int rec, int threads);

#ifdef __cplusplus
}
#endif
#endif
