/* sph_host.h — host-side C helpers of the stepper (libsph_host.so, plain C, no GPU).
 *
 * These mirror the parts of the reference's main() that sit on either side of the hot
 * path: scene generation (pi_sph_fluid.c:484-540, :238-240) and the gravity source
 * (:431-464).  They produce / consume the same `struct particle` arrays main() does,
 * so the C host (pi-sph-fluid_amd/host/desktop_sph_fluid.c), the tests and bench.py
 * all build their inputs through one implementation.
 */
#ifndef SPH_HOST_H
#define SPH_HOST_H

#include "sph.h"

#ifdef __cplusplus
extern "C" {
#endif

/* All generators follow one protocol: with out == NULL they return the number of particles
 * the scene has; otherwise they fill out[0..cap) and return the number written (or
 * SPH_E_ARG when cap is too small).  u = v = 0, rho = rho0, p = 0; fluid m = rho0*vol,
 * boundary m = 0 (psi is computed by sph_create). */

/* The exact default scene of the reference: lattice positions by f32 accumulation from 0
 * (x outer, y inner, :486-488/:497-498), kept where euclid_dist(x,y,WIDTH/2,HEIGHT/2) < 0.70
 * (:238-240).  With the default parameters: 269 particles. */
long sph_scene_default_fluid(const sph_params *prm, sph_particle *out, long cap);

/* The wall generator of :523-540 for the box of prm: for every lattice x: (x,y_min),(x,y_max);
 * for every lattice y: (x_min,y),(x_max,y).  accumulate != 0 reproduces the reference's f32
 * accumulation (default scene: 162 particles, corner (0,0) duplicated, corner (W,H) absent);
 * accumulate == 0 uses x = x_min + i*R (large boxes). */
long sph_scene_walls(const sph_params *prm, int accumulate, sph_particle *out, long cap);

/* Multi-layer walls (README.md:171-181 lists them as not implemented; Akinci's psi handles any wall sampling, :242-261):
 * `layers` nested rectangular frames of wall particles with spacing R, the innermost on the rectangle
 * [wx0,wx1] x [wy0,wy1], each further one R further out.  The frames must lie inside the domain box of prm (the
 * neighbour grid), so the caller makes the box at least layers*R larger than the inner rectangle on every side.
 * Corners are sampled once; the single-layer case differs from sph_scene_walls only in covering all four corners. */
long sph_scene_walls_layers(const sph_params *prm, float wx0, float wx1, float wy0, float wy1, int layers,
                            sph_particle *out, long cap);

/* Disc of lattice points (x = i*R, y = j*R, i outer / j inner) with distance < radius from
 * (cx,cy): the "drop on dry surface" scene scaled up (cfg1). */
long sph_scene_disc(const sph_params *prm, float cx, float cy, float radius, sph_particle *out, long cap);

/* nx x ny lattice block with its lower-left particle at (x0,y0), x = x0 + i*R (i outer, j inner):
 * the dam-break scenes (cfg2-4). */
long sph_scene_block(const sph_params *prm, float x0, float y0, long nx, long ny, sph_particle *out, long cap);
/* lattice columns [i_begin, i_end) of the same block, bit-identical to the corresponding part of sph_scene_block
 * (particle k of the range is particle i_begin*ny + k of the block): a slab host generates only what it holds. */
long sph_scene_block_range(const sph_params *prm, float x0, float y0, long nx, long ny, long i_begin, long i_end,
                           sph_particle *out, long cap);

/* ---- x-slab decomposition, host side (the reference has no distributed path) ---- */
/* columns of the device grid (cell = 2H + skin 2H) and the column of a position, in the device's own f32 arithmetic */
int  sph_slab_grid_columns(const sph_params *prm);
int  sph_slab_column_of(const sph_params *prm, float x);
/* cuts[0..world]: rank r owns cell columns [cuts[r], cuts[r+1]) of the lattice block (x0, nx, ny): contiguous ranges
 * with ~equal particle counts (quantiles of the per-column histogram), every range at least 4 columns, the first
 * beginning at column 0 and the last ending at the box edge (a dam-break front never leaves the decomposition). */
int  sph_slab_partition_block(const sph_params *prm, float x0, long nx, long ny, int world, int *cuts);
/* the same cuts from a per-column particle histogram hist[0..cols) (re-balancing: the histogram of the current state,
 * summed over the ranks) */
int  sph_slab_partition_counts(const long long *hist, int cols, int world, int *cuts);
/* lattice columns [i_begin, i_end) of the block that lie in cell columns [col_begin - 2, col_end + 2): what the rank
 * owning [col_begin, col_end) generates (sph_scene_block_range); global id of its k-th particle = i_begin * ny + k */
int  sph_slab_block_columns(const sph_params *prm, float x0, long nx, int col_begin, int col_end, long *i_begin, long *i_end);

/* ---- gravity source: get_gravity / get_gravity_routine (:431-464) ---- */
typedef enum sph_gravity_kind {
    SPH_GRAVITY_CONSTANT = 0,   /* (0,-G): the non-MPU6050 branch :442-443 */
    SPH_GRAVITY_TILT = 1,       /* scripted trace g = G(sin th, -cos th), th = amp*sin(2 pi t/period) (cfg4) */
    SPH_GRAVITY_MPU6050 = 2     /* sysfs IIO reader, :436-440 */
} sph_gravity_kind;

typedef struct sph_gravity {
    int kind;
    float g;               /* magnitude G */
    float amp_deg;         /* tilt amplitude, degrees */
    float period_s;        /* tilt period, seconds of simulated time */
    float hold_s;          /* zero-order hold (the 10 Hz poll of :455-461): 0.1 */
    char  sysfs_dir[256];  /* MPU6050: directory holding in_accel_{x,y}_raw */
    /* state */
    float last_t, gx, gy;
    int   primed;
} sph_gravity;

void sph_gravity_init(sph_gravity *gs, int kind, float g);
/* gravity vector at simulated time t, re-sampled at most every hold_s of simulated time
 * (the reference polls every 100 ms of wall time, which equals simulated time under REALTIME).
 * Returns 0, or SPH_E_ARG when the MPU6050 files cannot be read (reference: exit(1), :419-422). */
int  sph_gravity_sample(sph_gravity *gs, float t, float *gx, float *gy);

/* ---- wall velocity inferred from the accelerometer (the reference's README, "What's not implemented?" item 2,
 * README.md:175-176; the velocity enters the wall viscosity term pi_sph_fluid.c:357 through sph_set_boundary_velocity) ----
 * The accelerometer reports gravity as seen from the box.  Its slow part is the tilt (a first-order low pass with time
 * constant tau_tilt follows it); what remains is the linear acceleration of the box, which a leaky integrator (time
 * constant tau_leak: without the leak sensor bias would make the velocity drift) turns into the box's velocity:
 *     g_lp += (g - g_lp) dt / tau_tilt;   a = -(g - g_lp);   v = (v + a dt) exp(-dt / tau_leak)                     */
typedef struct sph_wall_motion {
    float tau_tilt;   /* [s], default 0.5 */
    float tau_leak;   /* [s], default 1.0 */
    float glx, gly;   /* low-passed gravity */
    float vx, vy;     /* inferred velocity of the box */
    int   primed;
} sph_wall_motion;
void sph_wall_motion_init(sph_wall_motion *wm);
/* one gravity sample (gx, gy) valid for the last dt seconds -> the box's velocity */
void sph_wall_motion_update(sph_wall_motion *wm, float gx, float gy, float dt, float *vx, float *vy);

#ifdef __cplusplus
}
#endif
#endif /* SPH_HOST_H */
