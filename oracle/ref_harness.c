/* ref_harness.c — TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Thin driver appended, by oracle/build_ref.sh, AFTER the hot-path section of
 * the reference translation unit (/root/reference/pi_sph_fluid.c lines 10-411,
 * piped straight from where it lies; nothing of it is copied into this repo).
 * Everything named here without a definition (struct particle, struct
 * neighbors_context, alloc_neighbors_context, update_neighbors_context,
 * calculate_*, draw_metaballs, the R/H/DT/... macros) is the reference's own.
 *
 * The harness only (a) gives the reference's functions arrays and a domain box
 * to work on and (b) exposes them under ref_* names with a plain C ABI so that
 * oracle/gen_golden.py can produce tests/golden/ fixtures and tests can pin
 * oracle/sph_oracle.c against the real thing.  The two functions that need
 * lines of the reference's main() (scene generation :476-540 and one time step
 * :612-644) get those lines spliced in by build_ref.sh as ref_scene_body() and
 * ref_step_body(); see that script.
 */

typedef struct { float x_min, x_max, y_min, y_max; } ref_box;

static struct neighbors_context *ref_ctx(int n, const ref_box *b) {
    /* cell length 2*H exactly as pi_sph_fluid.c:596-597 */
    return alloc_neighbors_context(n, b->x_min, b->x_max, b->y_min, b->y_max, 2*H);
}

static void ref_ctx_free(struct neighbors_context *c) {
    free(c->cells_head); free(c->cells_tail); free(c->particles_next); free(c);
}

/* constants as the reference's macros/expressions evaluate them */
void ref_constants(float *out) {
    const float B = C*C*RHO_0/7;                 /* :297 */
    out[0] = R; out[1] = H; out[2] = WIDTH; out[3] = HEIGHT; out[4] = RHO_0;
    out[5] = C; out[6] = G; out[7] = DT; out[8] = V; out[9] = RHO_0*V;   /* :502 */
    out[10] = B; out[11] = W(0, 0, 0, 0);        /* :274 */
    out[12] = W(0.2*H, 0, 0, 0);                 /* :325 */
    out[13] = 2*H;                               /* :596 */
    out[14] = (float)(0.01*H*H);                 /* :332 */
    out[15] = (float)MAX_POSSIBLE_NEIGHBORS;
}

int ref_grid_dims(const ref_box *b, int *n_cells, int *m_cells) {
    struct neighbors_context *c = ref_ctx(1, b);
    *n_cells = c->n_cells; *m_cells = c->m_cells;
    ref_ctx_free(c);
    return 0;
}

/* :600-601 — boundary list + Akinci pseudo-mass (writes boundary[i].m) */
int ref_psi(struct particle *boundary, int n_boundary, const ref_box *b) {
    if (n_boundary >= USHRT_MAX) return -1;
    struct neighbors_context *cb = ref_ctx(n_boundary, b);
    update_neighbors_context(cb, boundary);
    calculate_boundary_pseudomass(boundary, cb);   /* orphaned omp for: serial, as in the reference */
    ref_ctx_free(cb);
    return 0;
}

/* maximum neighbour count over all fluid particles (fluid list and boundary
 * list separately) — the reference overflows its 48-entry scratch silently
 * (:145), so fixtures are only valid when this stays <= 48. */
int ref_max_neighbors(struct particle *fluid, int n_fluid, struct particle *boundary, int n_boundary,
                      const ref_box *b, int *max_ff, int *max_fb) {
    if (n_fluid >= USHRT_MAX || n_boundary >= USHRT_MAX) return -1;
    struct neighbors_context *cf = ref_ctx(n_fluid, b), *cb = ref_ctx(n_boundary, b);
    update_neighbors_context(cf, fluid);
    update_neighbors_context(cb, boundary);
    /* count with the reference's own acceptance test but without its buffer */
    int mff = 0, mfb = 0;
    for (int i = 0; i < n_fluid; i++) {
        for (int pass = 0; pass < 2; pass++) {
            struct neighbors_context *c = pass ? cb : cf;
            struct particle *pb = pass ? boundary : fluid;
            int ic = (int)((fluid[i].y - c->y_min) / c->cell_length), jc = (int)((fluid[i].x - c->x_min) / c->cell_length);
            int cnt = 0;
            for (int a = ic-1; a <= ic+1; a++) for (int d = jc-1; d <= jc+1; d++) {
                if (a < 0 || a >= c->n_cells || d < 0 || d >= c->m_cells) continue;
                for (unsigned short j = c->cells_head[a*c->m_cells+d]; j != USHRT_MAX; j = c->particles_next[j])
                    if (euclid_dist(fluid[i].x, fluid[i].y, pb[j].x, pb[j].y) < 2*H && (pass || i != j)) cnt++;
            }
            if (pass) { if (cnt > mfb) mfb = cnt; } else { if (cnt > mff) mff = cnt; }
        }
    }
    *max_ff = mff; *max_fb = mfb;
    ref_ctx_free(cf); ref_ctx_free(cb);
    return 0;
}

/* stage evaluation, each stage optional (flags bit0 density, bit1 pressure,
 * bit2 acceleration); mirrors :604-607. boundary[].m must already hold psi. */
int ref_eval(struct particle *fluid, int n_fluid, struct particle *boundary, int n_boundary,
             const ref_box *b, float gx, float gy, int flags, float *du_dt, float *dv_dt, int threads) {
    if (n_fluid >= USHRT_MAX || n_boundary >= USHRT_MAX) return -1;
    struct neighbors_context *cf = ref_ctx(n_fluid, b), *cb = ref_ctx(n_boundary, b);
    update_neighbors_context(cb, boundary);
    update_neighbors_context(cf, fluid);
    #pragma omp parallel num_threads(threads)
    {
        if (flags & 1) calculate_density(fluid, boundary, cf, cb);
        if (flags & 2) calculate_particle_pressure(fluid, n_fluid);
        if (flags & 4) calculate_accelerations(du_dt, dv_dt, fluid, boundary, cf, cb, gx, gy);
    }
    ref_ctx_free(cf); ref_ctx_free(cb);
    return 0;
}

/* one time step = the reference's own loop body (:612-644), spliced in by
 * build_ref.sh as ref_step_body(); must be entered by every thread of a team. */
void ref_step_body(int n_fluid, struct particle *fluid, float *du_dt, float *dv_dt, struct particle *boundary,
                   struct neighbors_context *ctx_fluid, struct neighbors_context *ctx_boundary, float2 g);

/* nsteps of the reference's main loop, starting from a state whose du_dt/dv_dt
 * are the accelerations at that state (i.e. after ref_eval(...,7,...)). */
int ref_steps(struct particle *fluid, int n_fluid, struct particle *boundary, int n_boundary,
              const ref_box *b, float gx, float gy, float *du_dt, float *dv_dt, int nsteps, int threads) {
    if (n_fluid >= USHRT_MAX || n_boundary >= USHRT_MAX) return -1;
    struct neighbors_context *cf = ref_ctx(n_fluid, b), *cb = ref_ctx(n_boundary, b);
    update_neighbors_context(cb, boundary);
    update_neighbors_context(cf, fluid);
    float2 g = { gx, gy };
    #pragma omp parallel num_threads(threads)
    for (int s = 0; s < nsteps; s++)
        ref_step_body(n_fluid, fluid, du_dt, dv_dt, boundary, cf, cb, g);
    ref_ctx_free(cf); ref_ctx_free(cb);
    return 0;
}

/* the default scene = the reference's own main() lines :476-540, spliced in by
 * build_ref.sh as ref_scene_body(). Caller frees with ref_free(). */
int ref_scene_body(struct particle **fluid_out, int *n_fluid_out, struct particle **boundary_out, int *n_boundary_out);
int ref_scene(struct particle **fluid_out, int *n_fluid_out, struct particle **boundary_out, int *n_boundary_out) {
    return ref_scene_body(fluid_out, n_fluid_out, boundary_out, n_boundary_out);
}
void ref_free(void *p) { free(p); }

/* metaball rasterisation (:380-411) with the pixel grid of :570-577 spliced in
 * by build_ref.sh as ref_pixels_body(); draw_buffer is 1024 bytes, page format. */
void ref_pixels_body(struct particle *pixel_pseudoparticles);
int ref_metaballs(unsigned char *draw_buffer, struct particle *fluid, int n_fluid, const ref_box *b, int threads) {
    if (n_fluid >= USHRT_MAX) return -1;
    struct particle *px = (struct particle*)calloc(64*128, sizeof(struct particle));
    ref_pixels_body(px);
    struct neighbors_context *cf = ref_ctx(n_fluid, b);
    update_neighbors_context(cf, fluid);
    #pragma omp parallel num_threads(threads)
    draw_metaballs(draw_buffer, px, fluid, cf);
    ref_ctx_free(cf); free(px);
    return 0;
}
