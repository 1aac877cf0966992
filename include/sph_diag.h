/* sph_diag.h — measurement and diagnostic entry points of libsph_hip.so.
 *
 * NOT part of the drop-in boundary (that is sph.h: the calls the reference's main() makes, pi_sph_fluid.c:594-641, plus
 * the slab and metaball sections).  Nothing here changes results; bench.py, the profiling scripts and the tests use
 * these to time single kernels, to count why the neighbour structure was rebuilt and to select A/B variants.  A host
 * that only steps and reads back never includes this file.
 */
#ifndef SPH_DIAG_H
#define SPH_DIAG_H

#include "sph.h"

#ifdef __cplusplus
extern "C" {
#endif

/* names of the per-step kernels, in launch order (index into sph_kernel_times.ms).  The kernels marked [R] are
 * the rebuild of the neighbour structure (the reference's update_neighbors_context :104-124 + find_neighbors
 * :126-153): they return at once unless a rebuild was asked for.  sph_step launches them as ONE kernel with grid
 * barriers between the phases (sph_set_rebuild_launches); sph_profile_steps, which times them one by one, as four. */
enum {
    SPH_K_KICK_DRIFT     = 0,  /* :615-624 in place; requests a rebuild when a particle moved > skin/2 since the last */
    SPH_K_KEY_HIST       = 1,  /* [R] cell index of :111-113 + histogram (counting sort pass 1)      */
    SPH_K_SCAN           = 2,  /* [R] counting sort: exclusive scan -> cell_start                    */
    SPH_K_REORDER        = 3,  /* [R] counting sort: scatter to cell-contiguous order (replaces the linked list of :104-124) */
    SPH_K_BUILD_LIST     = 4,  /* [R] find_neighbors :126-153 once per rebuild: per-particle neighbour lists */
    SPH_K_DENSITY_EOS    = 5,  /* :263-289 + :294-301                            */
    SPH_K_FORCE_KICK     = 6,  /* :303-373 + :637-640                            */
    SPH_K_HALO           = 7,  /* end-of-step marker (slab mode: halo pack/ingest) */
    SPH_K_COUNT          = 8,
    /* sph_time_kernel only: the density launch as sph_step issues it on a single-GPU context — SPECULATIVE, with the check and
     * verify jobs of the rebuild criterion as extra workgroups of the same launch (see "neighbour-structure reuse" in sph.h) —
     * followed by a one-thread launch that does for those jobs what the step's gate does when nothing is rebuilt (completion
     * count, queue length and rebuild word back to where they were): the figure to hold against the profiler's
     * k_density_list<1, 0, true>, a launch boundary included.  SPH_K_DENSITY_EOS times the plain pass (the tiles alone).
     * SPH_E_STATE on contexts whose step does not launch it (slabs, the direct variant, contexts that share their device). */
    SPH_K_DENSITY_SPEC   = 9
};
typedef struct sph_kernel_times {
    float ms[SPH_K_COUNT];     /* mean device time per step of each kernel, HIP events on the context's stream */
    float step_ms;             /* mean device time of one whole step */
    int   nsteps;
    int   rebuilds;            /* how many of the nsteps rebuilt the neighbour structure */
} sph_kernel_times;

/* ---- per-kernel timing ---- */
/* run nsteps steps eagerly with HIP events around every kernel (same kernels as sph_step) */
int  sph_profile_steps(sph_ctx *ctx, float gx, float gy, int nsteps, sph_kernel_times *out);
/* mean device time [ms] of `reps` back-to-back launches of ONE per-step kernel on the live state, between two HIP
 * events on the context's stream.  Only the idempotent kernels (SPH_K_DENSITY_EOS, SPH_K_DENSITY_SPEC, SPH_K_FORCE_KICK: same
 * inputs -> same outputs, nothing they read is overwritten) can be timed this way; others give SPH_E_ARG.
 * SPH_K_FORCE_KICK re-does the kick of the last step: valid only after at least one sph_step since creation / upload /
 * sph_eval_accel — SPH_E_STATE otherwise: the velocities would be kicked a second time.  SPH_K_BUILD_LIST (single-GPU
 * contexts) rebuilds the lists on the sort that is there and leaves the rebuild request raised, so the next step
 * redoes the whole neighbour structure.  SPH_K_DENSITY_SPEC: for the duration of the measurement the criterion's jobs are told not to
 * repair lists (a pair they find missing only raises the rebuild word, which is put back): the state is left as it was found. */
int  sph_time_kernel(sph_ctx *ctx, int kernel, int reps, float *ms);
/* select kernel variant for density/force: 0 = default (best), others for A/B measurements */
int  sph_set_variant(sph_ctx *ctx, int variant);

/* ---- counters of the neighbour-structure reuse (see "neighbour-structure reuse" in sph.h) ---- */
/* why list builds put tiles on the direct path so far, as counts: [0] the tile touches more column pairs than the build's
 * tables hold, [1] more rows between its first and last particle than its row bitmap, [2] more runs of rows or cell-table
 * entries, [3] more candidates than the LDS tile, [4] a candidate window longer than a list byte can index, [5] a
 * neighbour list longer than the list capacity; and [6] workgroups of one-launch rebuilds that did not run on the XCD of
 * their grid-barrier leader and took the slow path (measurement / diagnostics) */
int   sph_direct_tile_reasons(sph_ctx *ctx, long long why[7]);
/* pairs of box groups whose particles were checked one by one (instead of a rebuild) because their boxes had moved more
 * than the skin relative to each other (single-GPU contexts; see k_check in csrc/sph_kernels.hip) */
int  sph_verify_stats(sph_ctx *ctx, long long *pairs);
/* who asked for the rebuilds so far (requests, several may ask for the same rebuild): why[0] box pairs that could not be verified
 * (too many failing neighbours of one group, the queue full, or a mode without verification), why[1] the verification found a
 * pair missing from the lists, why[2] a particle drifted H + skin from its sort position, why[3] unused (0) */
int  sph_rebuild_reasons(sph_ctx *ctx, long long why[4]);
/* list repair (single-GPU contexts, default order): out[0] pairs that the verification found inside the support and in nobody's list
 * and that were APPENDED to the two lists instead of asking for a rebuild; out[1..3] repairs that were not possible (the rebuild was
 * asked for after all): the partner not staged within reach of the lane's window bytes / no padding byte left in the lane's rows /
 * more repaired tiles in one step than the queue holds.  SPH_NO_LIST_REPAIR in the environment at creation switches it off (A/B). */
int   sph_repair_stats(sph_ctx *ctx, long long out[4]);
/* steps so far in which somebody was beyond skin/2 and the relative-motion check had to run */
int   sph_check_stats(sph_ctx *ctx, long long *checks);

/* ---- box calibration (round 6) ----
 * The boxes of a GPU pool differ (+-5-8 % in steps/s for one build of this library): a figure from one box cannot be held against a
 * figure from another without knowing what the box itself delivers.  Two micro-measurements on `device`, ~50 ms each, nothing to do
 * with any context:
 *   copy_gbs     a streaming copy kernel over 1 GiB (read + write bytes / time): the memory system as a simple kernel finds it
 *   valu_cycles  SIMD-cycles per wave-instruction of an independent v_fma_f32 stream with every SIMD saturated, priced at the
 *                NOMINAL 2.4 GHz (tools/ubench_valu's figure: 2.67 there; it moves with the clock the box really runs)
 *   clock_ghz    the shader clock under that load: ticks of the wave's cycle counter (s_memtime) over the 100 MHz constant
 *                counter (s_memrealtime) across the kernel; 0 if the two counters turn out to be the same clock
 * bench.py puts them into its JSON line (`box`) next to fractions taken against the MEASURED copy bandwidth. */
typedef struct sph_box_calibration {
    float copy_gbs, valu_cycles, clock_ghz;
    float copy_ms, valu_ms;      /* duration of the two measurements */
} sph_box_calibration;
int   sph_box_calibrate(int device, sph_box_calibration *out);

#ifdef __cplusplus
}
#endif
#endif /* SPH_DIAG_H */
