/* sph_params.c — reference defaults of the run-time parameters (pi_sph_fluid.c:11-20, :595).
 * Compiled into both libsph_hip.so and libsph_host.so. */
#include "sph.h"

void sph_params_default(sph_params *p) {
    p->r = 0.0750f;                  /* :11 */
    p->h = p->r * 1.3f;              /* :12 */
    p->rho0 = 1000.0f;               /* :15 */
    p->c = 400.0f;                   /* :16 */
    p->g = 9.81f;                    /* :17 */
    p->dt = 1.0f * p->h / p->c;      /* :19 */
    p->vol = 0.57f * p->h * p->h;    /* :20 */
    p->x_min = 0.0f; p->x_max = 4.0f;    /* WIDTH  :13 */
    p->y_min = 0.0f; p->y_max = 2.0f;    /* HEIGHT :14 */
    p->alpha = 0.01f; p->eps = 0.01f; p->k1 = 0.1f; p->k2 = 0.2f;   /* :325, :332, :334 */
    p->deterministic = 0;
    p->skin = 0.30f;                 /* neighbour-structure reuse: the skin adapts between skin_min and skin (fixed skins on the 2M-particle */
    p->skin_min = 0.08f;             /* dam break: 0.15 is best while the fluid is at rest, 0.30 once the flow has developed); skin_min: where lists live long; 1.5 x that otherwise (adapt_skin) */
}

int sph_abi_version(void) { return SPH_ABI_VERSION; }

const char *sph_error_string(int err) {
    switch (err) {
        case SPH_OK: return "ok";
        case SPH_E_ARG: return "bad argument";
        case SPH_E_HIP: return "HIP failure / no usable gfx950 device";
        case SPH_E_OUT_OF_DOMAIN: return "particles left the domain (clamped into edge cells)";
        case SPH_E_NAN: return "particle position became NaN/Inf";
        case SPH_E_NOMEM: return "out of memory";
        case SPH_E_CAPACITY: return "slab/halo capacity exceeded";
        case SPH_E_STATE: return "call not valid in this state";
        default: return "unknown error";
    }
}
