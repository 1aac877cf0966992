/* sph_slab_host.c — host-side helpers of the x-slab decomposition (plain C, no GPU): which cell columns a rank owns and
 * which lattice columns of a block scene it must generate.  The reference has no distributed path; this is the host
 * logic of SURVEY.md 8e (quantiles of the per-column particle histogram), shared by the C multi-GPU host
 * (slab_sph_fluid.c) and, through ctypes, checked against the Python partitioner (pi-sph-fluid_amd/slab.py).
 * Built with -ffp-contract=off: the column of a position must be the device's own f32 arithmetic (cell_of in
 * csrc/sph_kernels.hip): (int)((x - x_min) * (1 / cell)), cell = 2H + skin * 2H. */
#include "sph_host.h"

#include <stdlib.h>

static float device_cell(const sph_params *p) {
    const float two_h = 2 * p->h;
    return two_h + p->skin * two_h;
}

int sph_slab_grid_columns(const sph_params *p) {
    if (!p) return SPH_E_ARG;
    return (int)((p->x_max - p->x_min) / device_cell(p)) + 1;
}

int sph_slab_column_of(const sph_params *p, float x) {
    const float inv = 1.0f / device_cell(p);
    return (int)((x - p->x_min) * inv);
}

int sph_slab_partition_counts(const long long *hist, int cols, int world, int *cuts) {
    if (!hist || !cuts || cols < 1 || world < 1) return SPH_E_ARG;
    double *cum = (double *)calloc((size_t)cols, sizeof(double));
    if (!cum) return SPH_E_NOMEM;
    for (int c = 0; c < cols; c++) cum[c] = (double)hist[c] + (c ? cum[c - 1] : 0.0);
    const double total = cum[cols - 1];
    cuts[0] = 0;                                          /* the slabs tile the whole box */
    for (int r = 1; r < world; r++) {
        const double target = total * (double)r / (double)world;
        int lo = 0, hi = cols;                            /* first column with cum >= target */
        while (lo < hi) {
            const int mid = (lo + hi) / 2;
            if (cum[mid] < target) lo = mid + 1; else hi = mid;
        }
        int c = lo + 1;                                   /* first column boundary at or past the quantile */
        if (c < cuts[r - 1] + 4) c = cuts[r - 1] + 4;     /* a slab owns at least 4 columns */
        cuts[r] = c;
    }
    cuts[world] = cols;
    for (int r = world - 1; r >= 1; r--)
        if (cuts[r] > cuts[r + 1] - 4) cuts[r] = cuts[r + 1] - 4;
    free(cum);
    for (int r = 0; r < world; r++)
        if (cuts[r + 1] - cuts[r] < 4 || cuts[r] < 0) return SPH_E_ARG;      /* scene too narrow for that many slabs */
    return SPH_OK;
}

int sph_slab_partition_block(const sph_params *p, float x0, long nx, long ny, int world, int *cuts) {
    if (!p || !cuts || nx <= 0 || ny <= 0 || world < 1) return SPH_E_ARG;
    const int cols = sph_slab_grid_columns(p);
    long long *hist = (long long *)calloc((size_t)cols, sizeof(long long));
    if (!hist) return SPH_E_NOMEM;
    for (long i = 0; i < nx; i++) {                       /* per-column particle counts follow from the lattice */
        int c = sph_slab_column_of(p, x0 + (float)i * p->r);
        if (c < 0) c = 0;
        if (c > cols - 1) c = cols - 1;
        hist[c] += (long long)ny;
    }
    const int rc = sph_slab_partition_counts(hist, cols, world, cuts);
    free(hist);
    return rc;
}

int sph_slab_block_columns(const sph_params *p, float x0, long nx, int col_begin, int col_end, long *i_begin, long *i_end) {
    if (!p || !i_begin || !i_end || nx < 0) return SPH_E_ARG;
    long b = nx, e = nx;                                  /* lattice columns are monotone in x: one contiguous range */
    for (long i = 0; i < nx; i++) {
        const int c = sph_slab_column_of(p, x0 + (float)i * p->r);
        if (c >= col_begin - 2 && c < col_end + 2) {
            if (b == nx) b = i;
            e = i + 1;
        }
    }
    if (b == nx) { b = 0; e = 0; }
    *i_begin = b;
    *i_end = e;
    return SPH_OK;
}
