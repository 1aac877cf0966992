/* sph.h — C ABI of the MI355X-native 2-D WCSPH stepper (libsph_hip.so).
 *
 * This is the drop-in boundary for the per-step compute of
 * colonelwatch/pi-sph-fluid.  The reference has no FFI seam: main() owns raw
 * arrays and calls the physics functions directly.  Each entry point below
 * names the reference call site(s) it replaces (file = pi_sph_fluid.c).
 * Plain C types only; no HIP, torch or C++ types cross this boundary.
 *
 * Conventions
 *   - every function returning int returns SPH_OK (0) or a negative sph_error;
 *     sph_last_error() gives the text of the last failure of that context
 *     (reference convention is exit(1) :419-422, printf warnings :546-547, or UB)
 *   - the library owns all device memory; the caller owns every buffer it
 *     passes in or receives results in (reference: bare malloc'd arrays, :491-493)
 *   - one sph_ctx is driven by one host thread at a time (reference: all
 *     threads of the omp team enter the physics functions, :610, :630-632)
 *   - work is enqueued on the context's HIP stream; sph_step() returns without
 *     waiting, read-backs and sph_sync() wait
 *   - there is NO CPU fallback: without a usable MI355X every compute entry
 *     point fails with SPH_E_HIP
 */
#ifndef SPH_H
#define SPH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPH_ABI_VERSION 8      /* 8 (round 6): sph_slab_steps (runs of lean steps as graphs); 7 (round 5): the halo buffers' header (update written by the force pass), sph_slab_step + sph_slab_set_peer_links,
                                 * sph_set_rebuild_launches(ctx, N > 1) = a capped grid, verification on slab contexts */

typedef enum sph_error {
    SPH_OK = 0,
    SPH_E_ARG = -1,            /* bad argument (NULL, negative count, non-uniform fluid mass, ...) */
    SPH_E_HIP = -2,            /* HIP runtime failure or no gfx950 device */
    SPH_E_OUT_OF_DOMAIN = -3,  /* particles left [x_min,x_max]x[y_min,y_max]; they were clamped into edge cells (reference: heap overflow, :111-116) */
    SPH_E_NAN = -4,            /* a particle position became NaN/Inf */
    SPH_E_NOMEM = -5,
    SPH_E_CAPACITY = -6,       /* slab mode: local particle or halo capacity exceeded */
    SPH_E_STATE = -7           /* call not valid in the context's current state */
} sph_error;

/* byte-compatible with the reference's `struct particle` (:26-31): 7 x f32, 28 B, no padding.
 * fluid: m = RHO_0*V (:502); boundary: m holds the Akinci pseudo-mass psi (:259), rho = RHO_0 (:526). */
typedef struct sph_particle { float x, y, u, v, m, rho, p; } sph_particle;

/* the reference's compile-time constants (:11-20) and domain box (:595) as run-time parameters */
typedef struct sph_params {
    float r;      /* R      initial particle spacing [m]       :11 */
    float h;      /* H      smoothing length = 1.3 R           :12 */
    float rho0;   /* RHO_0  reference density                  :15 */
    float c;      /* C      numerical speed of sound           :16 */
    float g;      /* G      gravitational acceleration         :17 */
    float dt;     /* DT     time step = H/C                    :19 */
    float vol;    /* V      fluid particle volume = 0.57 H^2   :20 */
    float x_min, x_max, y_min, y_max;   /* neighbour-grid domain, :595 ({0,WIDTH,0,HEIGHT}) */
    float alpha;  /* 0.01   artificial viscosity               :334 */
    float eps;    /* 0.01   viscosity singularity guard        :332 */
    float k1;     /* 0.1    artificial pressure strength       :325 */
    float k2;     /* 0.2    artificial pressure reference q    :325 */
    float skin;   /* 0.30   Verlet skin of the neighbour structure as a fraction of 2H (no reference counterpart: the
                            reference rebuilds every step, :626 = skin 0): the LARGEST skin the lists are built with, and what
                            the device grid is sized for (cell = 2H (1 + skin)).  Per context; see "neighbour-structure reuse" */
    int deterministic;   /* 0 (default): the particles of a grid cell are kept in the order in which the binning atomics
                            arrived, which differs from run to run (results agree to rounding: the summation order of a
                            particle's neighbours follows it).  1: in particle-id order (one more pass per rebuild, < 1 % of
                            a step): results are bit-identical from run to run (the reference is deterministic too: one
                            fixed traversal order), and the same bits on one GPU and in any number of slabs
                            as long as both rebuild in the same steps (the order follows the cells at the last rebuild):
                            always with skin = 0; with a skin, a slab may rebuild a step earlier than a single context
                            would (waves next to ghost particles use the absolute criterion) */
    float skin_min;      /* 0.08   the SMALLEST skin (ABI v6; round 5: 0.08, was 0.12 — and reached only by lists that follow lists which
                            lived 100 steps or longer: a fluid at rest or sloshing gently; lists that died sooner are followed by
                            lists of at least 1.5 skin_min = the old 0.12; the first lists of a context have skin_min).
                            skin_min < skin: the skin adapts — every rebuild looks at how many
                            steps the last lists lasted (how fast the flow uses up a skin) and picks the skin that minimises
                            list-walking cost + rebuild cost per step for that rate: larger in a violent flow (a rebuild costs
                            more than two steps), smaller in a calm one (shorter lists, cheaper steps).  skin_min >= skin (or
                            skin = 0): the skin is fixed at `skin`.  Results do not depend on the skin beyond summation order. */
} sph_params;

typedef struct sph_ctx sph_ctx;

/* reference defaults (:11-20), box 4 x 2 */
void sph_params_default(sph_params *prm);
int  sph_abi_version(void);
const char *sph_error_string(int err);
/* number of usable HIP devices (0 when there is none; never initialises a context) */
int  sph_device_count(void);

/* Replaces the init sequence :594-607 (alloc_neighbors_context x2, update_neighbors_context,
 * calculate_boundary_pseudomass, first calculate_density/_particle_pressure/_accelerations).
 * Copies fluid[0..n_fluid) (x,y,u,v; m must equal rho0*vol) and boundary[0..n_boundary) (x,y, and u,v: the
 * fluid-boundary viscosity term reads the wall particle's stored velocity, :357; walls themselves do not move),
 * computes psi, bins everything, evaluates rho, p and a at t = 0 under gravity (gx,gy).
 * device = HIP device ordinal. */
int  sph_create(sph_ctx **out, const sph_params *prm,
                const sph_particle *fluid, int n_fluid,
                const sph_particle *boundary, int n_boundary,
                float gx, float gy, int device);
void sph_destroy(sph_ctx *ctx);
const char *sph_last_error(const sph_ctx *ctx);

/* Replaces the loop body :612-641, nsteps times: kick 1/2, drift, rebuild the neighbour
 * structure, density, pressure, acceleration under (gx,gy), kick 1/2.  The reference re-reads
 * g every step (:632); here it is sampled once per call.  Asynchronous. */
int  sph_step(sph_ctx *ctx, float gx, float gy, int nsteps);
/* wait for all enqueued work; reports SPH_E_OUT_OF_DOMAIN / SPH_E_NAN seen since the last sync */
int  sph_sync(sph_ctx *ctx);

/* Read-back in ORIGINAL particle order (what main() reads from fluid[] at :649, :657-671).
 * out[i] = {x,y,u,v,m,rho,p} of the particle given as fluid[i] to sph_create. */
int  sph_read_particles(sph_ctx *ctx, sph_particle *out);
/* du_dt[], dv_dt[] of :492-493, original order.  (The step loop does not keep them in memory — its force pass uses them for its
 * kicks and moves on; this call, and the velocity of sph_read_particles / sph_stats, evaluates them once more on the state the last
 * step left: one force pass, then the copy.) */
int  sph_read_accel(sph_ctx *ctx, float *du_dt, float *dv_dt);
/* boundary particles, original order, psi in .m (:259) */
int  sph_read_boundary(sph_ctx *ctx, sph_particle *out);
/* Walls that move (README.md:175-176 lists "boundary velocity" as not implemented; the hook is the wall particle's
 * stored u, v in the fluid-boundary viscosity term, :357).  Overwrites x, y, u, v of every wall particle (original
 * order, n_boundary of them) and re-bins the walls; psi is kept (:259: rigid motion does not change a wall particle's own
 * wall neighbourhood).  Call between steps; the next step rebuilds the fluid's neighbour structure against the new bins.
 * The walls must stay inside the domain box. */
int  sph_update_boundary(sph_ctx *ctx, const sph_particle *boundary);
/* every wall particle moves with (u, v) from now on: the velocity the wall viscosity term (:357) sees.  Positions stay
 * (sph_update_boundary moves walls); nothing is re-binned or rebuilt, so this is cheap enough for every step, e.g. with
 * the velocity a host infers from its accelerometer (sph_wall_motion, include/sph_host.h; README.md:175-176). */
int  sph_set_boundary_velocity(sph_ctx *ctx, float u, float v);
/* the statistics of :657-671 as device reductions: max rho and max sqrt(u^2+v^2) over fluid (a slab context: over the
 * particles it owns; the host takes the maximum over the ranks) */
int  sph_stats(sph_ctx *ctx, float *max_rho, float *max_speed);

int  sph_n_fluid(const sph_ctx *ctx);
int  sph_n_boundary(const sph_ctx *ctx);
/* n_cells (rows, y) and m_cells (columns, x) as :93-94 computes them (cell length 2H) */
int  sph_grid_dims(const sph_ctx *ctx, int *n_cells, int *m_cells);
/* the device's own neighbour grid: rows, columns and cell length 2H + skin (slab mode: the local grid) */
int  sph_device_grid(const sph_ctx *ctx, int *rows, int *cols, float *cell);

/* ---- neighbour-structure reuse (Verlet skin) ----
 * The reference rebuilds its linked list every step (:626) and searches it four times per particle (:278, :283,
 * :314, :343).  Here one rebuild (counting sort + per-particle neighbour lists holding every pair closer than
 * 2H + skin) serves density and force, and stays in use while it is provably complete: as long as every particle is
 * within skin/2 of where it was at the rebuild, or — single GPU — as long as no two particles whose cells were at
 * most two cells apart have moved more than the skin RELATIVE to each other (checked on per-wave displacement boxes,
 * so a jet moving as a whole keeps its lists; where two boxes fail that bound, their particles are checked pair by pair
 * before a rebuild is asked for: sph_set_verification) and nobody has moved more than H + skin.  Until then no unlisted pair
 * can be inside the support 2H, and listed pairs beyond 2H contribute exactly 0.  Results do not depend on the skin
 * (beyond summation order).  skin = 0: a rebuild whenever neighbouring particles moved relative to each other at all.
 * The skin is a fraction of 2H between sph_params.skin_min and sph_params.skin (0 <= skin <= 1), chosen by the device at
 * every rebuild from how long the last lists lasted; skin_min >= skin fixes it.  sph_current_skin() reads it. */
float sph_current_skin(sph_ctx *ctx);      /* the skin (fraction of 2H) the present lists were built with */
/* cell length of the device grid for these parameters: 2H (1 + skin) — what slab hosts must bin with */
float sph_device_cell(const sph_params *prm);
/* ask for a rebuild of the neighbour structure in the next step whatever the displacement criterion says (measurement;
 * hosts that want the reference's rebuild-every-step behaviour for a while without re-creating the context) */
int   sph_request_rebuild(sph_ctx *ctx);
/* rebuilds since creation, and tiles that list builds have put on the direct (no list) path */
int   sph_rebuild_stats(sph_ctx *ctx, long long *rebuilds, long long *direct_tiles);
/* Verification of failing box pairs (jobs inside the launch of the density pass of sph_step): mode -1 = automatic (from 500 000
 * particles on, where a rebuild costs far more than checking a few thousand particle pairs), 0 = never (two boxes that have moved
 * more than the skin relative to each other ask for the rebuild), 1 = always.  Contexts with skin > 0.  A slab context (round 5)
 * verifies only in mode 1 — with blocks of its step's head kernel, which is on the step's critical path (sph_step's jobs ride inside
 * its density launch): worth it where rebuilds are frequent and expensive — for the groups whose neighbourhood holds owned particles
 * only (a group that can meet ghosts keeps the absolute criterion: its partners' boxes are not known before the exchange); every
 * rank may choose for itself: the rebuild word is MAX-reduced, so all ranks rebuild in the same steps whoever verifies. */
int  sph_set_verification(sph_ctx *ctx, int mode);
/* List repair (round 5; contexts whose verification is on, default particle order): a pair that the verification finds
 * inside the support and in nobody's list is APPENDED to the two lists it is missing from — when its partner is staged within reach
 * of the lane's window bytes and the lane (or, by half a row, its wave) has room — instead of asking for the rebuild of everything;
 * the density of the repaired tiles is repeated in the same step.  Exact either way (tests/test_gpu_verlet.py: lists against the
 * walk over the cell ranges).  mode -1 = automatic: from 4 000 000 particles on (a rebuild of 32 M particles costs 4.2 ms, of 2 M
 * 0.28 ms; a step with repairs ~10 us), 0 = never, 1 = always.  Switching it on asks for one rebuild (lists built while it was off
 * carry neither the spare row of padding nor the remembered partners a repair needs).  Counters: sph_diag.h. */
int  sph_set_list_repair(sph_ctx *ctx, int mode);
/* total particles clamped into the domain so far (0 in a healthy run) */
long long sph_out_of_domain_count(sph_ctx *ctx);

/* ---- stage entry points: the individual calculate_* calls, for staged parity gates ---- */
/* overwrite x,y,u,v AND rho,p of every fluid particle (original order) and re-bin (:604).  The stored accelerations
 * stay aligned with their particles (reference: du_dt[] is indexed like fluid[], :616), so
 * sph_upload_state(sph_read_particles()) followed by sph_step() continues the run it was read from. */
int  sph_upload_state(sph_ctx *ctx, const sph_particle *fluid);
/* overwrite du_dt[], dv_dt[] (original order): with sph_upload_state this restores a checkpoint taken with
 * sph_read_particles + sph_read_accel */
int  sph_upload_accel(sph_ctx *ctx, const float *du_dt, const float *dv_dt);
int  sph_eval_density(sph_ctx *ctx);                       /* calculate_density :263-289, rho only */
int  sph_eval_pressure(sph_ctx *ctx);                      /* calculate_particle_pressure :294-301, from the stored rho */
int  sph_eval_accel(sph_ctx *ctx, float gx, float gy);     /* calculate_accelerations :303-373, from the stored x,y,u,v,rho,p */

/* ---- streams, memory, launch structure ---- */
/* adopt an existing hipStream_t (e.g. the host framework's current stream); NULL = own stream */
int  sph_set_stream(sph_ctx *ctx, void *hip_stream);
/* device bytes held by the context */
size_t sph_device_bytes(const sph_ctx *ctx);
/* How a step launches its rebuild chain (binning, scan, scatter, lists): one_launch = 1 (default of single-GPU contexts):
 * ONE kernel with grid barriers between the phases, sized to what the device holds at once, so that the many steps
 * that rebuild nothing pay for one empty launch instead of four.  It needs the device to itself: sph_create tries its
 * barriers once and falls back to one kernel per phase where they do not complete (compute units masked off or held
 * by another process); should that happen later, a barrier gives up after a few seconds and the next call that checks
 * the flags returns SPH_E_STATE.  one_launch = 0: one kernel per phase (what a single-GPU context does anyway while
 * another context of the same process lives on its device: A/B comparisons).
 * Slab contexts do the same with what follows their halo exchange (ghost update / ingest, scan, scatter, canonical
 * order, lists), but only when their host asks for it (default: one kernel per phase): several slabs may share a device,
 * in one process or in several, and the library cannot see the other processes.  A host that calls this with
 * one_launch = 1 on a slab context vouches that nothing else computes on that device while the slab steps: one rank per
 * GPU (the C multi-GPU host over RCCL does), or slabs of one device stepped strictly one after the other with a
 * synchronisation in between.  Results are the same either way.  one_launch > 1: the same with at most that many workgroups
 * (a multiple of 8) — ranks that DO share a device may cap their grids so that all of them are resident together (the grid
 * barriers need that; a barrier that cannot complete gives up after a few seconds: SPH_E_STATE): rehearsals of the multi-rank
 * step on one GPU. */
int  sph_set_rebuild_launches(sph_ctx *ctx, int one_launch);
int  sph_get_rebuild_launches(const sph_ctx *ctx);      /* 1: one launch, 0: one kernel per phase (as of the last step) */

/* ---- multi-GPU: x-slab domain decomposition, one process per GPU ----
 * The reference has no distributed path; this is the sharding of its particle loops (:272, :311) by cell column.
 * A slab owns the cell columns [col_begin, col_end) of the device grid (sph_device_cell) and keeps 2 ghost columns
 * per side.  One halo exchange per step, as fixed-capacity buffers: uint32 header[4] = {update count, update step, record
 * count, record step}, then ONE payload that holds either
 *   records (a step that rebuilds the neighbour structure): 5 words {x, y, u, v, id} each — every particle now inside the
 *          neighbour's reach (this slab's 2 outermost owned columns + anything that migrated across since the last rebuild);
 *          ownership follows position: after the sort a slab owns whatever lies in its columns.  Appended by
 *          sph_slab_step_pack; header words 2 and 3 (zeroed at the start of every step) count them and name the step;
 *   an update (any other step): 4 words {x, y, u, v} per particle of the 2 outermost owned columns in array order, which is
 *          the order of the neighbour's ghost columns (interface cells are kept sorted by particle id).  Written — header
 *          words 0 and 1 included — by the kernel that drifted the particles: the force pass of the step BEFORE (it has the
 *          next step's positions and velocities in registers; round 5), after creation the stand-alone kick / drift.
 *          sph_slab_step_pack has nothing to do on such a step.
 * The receiver checks count and step of what it consumes (SPH_E_STATE: neighbouring slabs out of step).
 * All slabs must rebuild in the same step, so the rebuild request is ONE 32-bit word per slab that the host
 * MAX-reduces over all ranks each step (RCCL all-reduce on the device word, or any other transport):
 *     sph_slab_step_begin()   kick 1/2 + drift of the owned particles; raises the word when lists may be stale (:615-624)
 *     -- reduce the word over all ranks --
 *     sph_slab_step_pack()    a rebuild step: fills the send buffers with records (any other step: the update is there already)
 *     -- move send_right -> right neighbour's recv_left, send_left -> left neighbour's recv_right --
 *     sph_slab_step_overlap() optional, while the buffers move: density of the tiles that stage no ghost particle
 *                             (or sph_slab_step_overlap_on(): the same on a stream of the host's, from right after
 *                             sph_slab_step_begin — beside the reduction of the word too)
 *     sph_slab_step_end()     ingest + sort + lists (records) or ghost update, density + EOS + force + kick (:626-640);
 *                             the force pass leaves the NEXT step's update message in the send buffers */
typedef struct sph_slab_desc {
    int col_begin, col_end;      /* owned global cell columns [begin, end), at least 4 */
    int has_left, has_right;     /* a neighbouring slab exists */
    int halo_capacity;           /* records per halo buffer (0 = default 64 x rows) */
    int particle_capacity;       /* local particle capacity incl. ghosts (0 = default) */
} sph_slab_desc;

/* fluid[0..n_fluid) = every particle inside columns [col_begin-2, col_end+2) with its global id;
 * boundary_all = ALL wall particles of the scene (psi needs the full wall set; the slab keeps its part). */
int  sph_create_slab(sph_ctx **out, const sph_params *prm, const sph_slab_desc *desc,
                     const sph_particle *fluid, const uint32_t *ids, int n_fluid,
                     const sph_particle *boundary_all, int n_boundary_all, float gx, float gy, int device);
int  sph_slab_step_begin(sph_ctx *ctx, float gx, float gy);
int  sph_slab_step_pack(sph_ctx *ctx);
int  sph_slab_step_overlap(sph_ctx *ctx);
/* The interior density pass on `hip_stream` (a hipStream_t of the caller's on the context's device; NULL = the context's
 * own stream) at any point after sph_slab_step_begin: it needs the drifted positions and nothing of this step's halo.
 * The CALLER orders the streams: `hip_stream` must wait for what sph_slab_step_begin enqueued on the context's stream
 * (an event), and the context's stream must wait for this pass before sph_slab_step_end.  The pass reads the rebuild word
 * as it finds it (reduced or not): on a step that turns out to rebuild its work is discarded, never wrong. */
int  sph_slab_step_overlap_on(sph_ctx *ctx, void *hip_stream);
int  sph_slab_step_end(sph_ctx *ctx);
/* the rebuild word: its device address (library-owned unless replaced), adopting a word of the host framework
 * (e.g. a 1-element int32 torch tensor handed to RCCL; NULL = back to the library's own), host-staged access */
int  sph_slab_flag_buffer(sph_ctx *ctx, void **dev_word);
int  sph_slab_set_flag_buffer(sph_ctx *ctx, void *dev_word);
int  sph_slab_flag_get(sph_ctx *ctx, uint32_t *value);
int  sph_slab_flag_set(sph_ctx *ctx, uint32_t value);
/* device addresses and byte size of the four halo buffers (library-owned unless replaced below) */
int  sph_slab_buffers(sph_ctx *ctx, void **send_left, void **send_right, void **recv_left, void **recv_right, size_t *bytes);
/* adopt device buffers of the host framework (e.g. torch tensors handed to RCCL); each >= the size above.  Between steps only;
 * what the send buffers hold (the next step's update message) moves along */
int  sph_slab_set_buffers(sph_ctx *ctx, void *send_left, void *send_right, void *recv_left, void *recv_right, size_t bytes);
/* host-staged transport (tests, non-RCCL hosts): side 0 = left, 1 = right */
int  sph_slab_copy_out(sph_ctx *ctx, int side, void *host_bytes);
int  sph_slab_copy_in(sph_ctx *ctx, int side, const void *host_bytes);
/* ---- peer-mapped transport (ranks of ONE node; xGMI is point to point and every GPU can map its peers' memory) ----
 * Instead of a collective library on the per-step path: the HOST allocates, per rank, the two receive buffers
 * (sph_slab_set_buffers), two arrival flags and a slot array uint32[2][SPH_PEER_MAX_RANKS], exports them (hipIpcGetMemHandle)
 * and opens its peers'; per step, all on the context's stream, with `tag` = a number that grows by one per step:
 *     sph_slab_step_begin
 *     sph_slab_peer_reduce   stores this rank's rebuild word into its slot of every rank's array, waits for theirs: MAX
 *     sph_slab_step_pack
 *     sph_slab_peer_push     copies the send buffers into the neighbours' receive buffers, then raises their arrival flags
 *     sph_slab_peer_wait     waits for this rank's two arrival flags
 *     sph_slab_step_end
 * One receive buffer per side is enough: a neighbour can only push step t + 1 after its own peer_reduce of step t + 1,
 * which needs this rank's word of step t + 1, which this rank's stream stores after its sph_slab_step_end of step t.
 * All waits are bounded: a peer that never arrives ends in SPH_E_STATE at the next call that reads the flags.
 * slots_of_rank[q] = the address (in this process) of rank q's slot array, q = 0 .. n_ranks - 1 (entry `me`: its own);
 * a missing neighbour: NULL buffer and flag. */
#define SPH_PEER_MAX_RANKS 8
int  sph_slab_peer_reduce(sph_ctx *ctx, void *const *slots_of_rank, int me, int n_ranks, uint32_t tag);
int  sph_slab_peer_push(sph_ctx *ctx, void *left_recv_right, void *left_flag, void *right_recv_left, void *right_flag, uint32_t tag);
int  sph_slab_peer_wait(sph_ctx *ctx, const void *flag_from_left, const void *flag_from_right, uint32_t tag);
/* ---- the lean slab step (round 5): ONE call, FOUR kernels, nothing of the exchange as a launch of its own ----
 *     head      k_check's work (count the step, this step's gravity, the boxes when somebody is beyond skin/2) and, with links,
 *               the push of this step's update message (the last force pass left it in the send buffers) into the neighbours'
 *               receive buffers + their arrival flags, and the MAX of the rebuild word over the ranks (slot arrays, as above)
 *     update /  a step that does not rebuild: waits for the neighbours' flags (with links), updates the ghosts.  A rebuild step:
 *     rebuild   the pack (keys + histogram of the owned range, records for the neighbours), with links the records pushed to the
 *               neighbours and theirs awaited between two grid barriers, ingest, scan, scatter, canonical order, lists — ONE
 *               launch: the context must have been given its device (sph_set_rebuild_launches(ctx, 1)) — SPH_E_STATE otherwise
 *     density   rho + EOS of every tile
 *     force     a, kick, the NEXT step's kick 1/2 + drift — and its update message into the send buffers
 * (pi_sph_fluid.c:612-641; the three-call step above is six to eight launches.)  Without links (sph_slab_set_peer_links never
 * called, or NULL): a slab without neighbours.  The links name memory of the other ranks as mapped into this process
 * (hipIpcOpenMemHandle: the peer-mapped transport above); every rank's block holds TWO receive buffers per side — the message
 * of step t goes to parity t & 1: a neighbour may push step t + 1 while this rank still reads step t — one arrival flag per side
 * (tag 2 t for the update of step t, 2 t + 1 for the records of a rebuild step t; a flag only grows and a wait is for
 * "at least": the neighbour's next head kernel may have raised it to 2 (t + 1) already) and the slot array of the word exchange.
 * All waits are bounded (SPH_E_STATE at the next call that reads the flags; the error string names the wait that gave up first). */
typedef struct sph_peer_links {
    int   me, n_ranks;
    void *slots_of_rank[SPH_PEER_MAX_RANKS];   /* every rank's slot array uint32[2][SPH_PEER_MAX_RANKS] (entry `me`: this rank's own) */
    void *left_recv[2],  *left_flag;            /* the LEFT neighbour's receive buffers for what comes from ITS right (parity 0, 1) and that flag; NULL: no neighbour */
    void *right_recv[2], *right_flag;           /* the RIGHT neighbour's receive buffers for what comes from its left */
    void *my_recv_left[2], *my_recv_right[2];   /* this rank's own receive buffers (device memory of its exported block) */
    void *my_flag_left,  *my_flag_right;        /* this rank's own arrival flags */
} sph_peer_links;
int  sph_slab_set_peer_links(sph_ctx *ctx, const sph_peer_links *links);      /* between steps; NULL: none */
int  sph_slab_step(sph_ctx *ctx, float gx, float gy);
/* nsteps lean steps in one call (round 6, ABI v8): step s runs under gravity (gravity_xy[2 s], gravity_xy[2 s + 1]) — the host polls its
 * gravity source for every step up front (the reference re-reads g every step, :632; a 10 Hz source changes it every ~400 steps).
 * Runs of 16 / 8 / 4 / 2 steps are replayed as captured graphs — the four launches of a step take the step's number, the parity of its
 * receive buffers and its gravity from device memory, so nothing in a launch changes from step to step — and what is left goes through
 * sph_slab_step (also the first step after creation or an upload).  Same kernels, same results as nsteps calls of sph_slab_step (bitwise
 * with sph_params.deterministic); with links every rank must call it with the same nsteps.  Measured (one slab of 2 M particles through
 * the C host): the same rate as one call per step, 7 888 against 7 867 steps/s — a kernel trace shows no gaps between a step's launches
 * either way; what graphs save is the host's work per step. */
int  sph_slab_steps(sph_ctx *ctx, const float *gravity_xy, int nsteps);
/* the SPECULATIVE lean step (round 6; on = 1; default 0): what makes sph_step's neighbour lists last — failing boxes verified particle
 * by particle, missing pairs appended — costs sph_step nothing because those jobs ride in the launch of a density pass that assumes the
 * lists valid.  The plain lean step cannot do that (its head kernel must know the verdict before the word goes round), so a slab rebuilt
 * 2.3 x as often as sph_step on the dam break (0.104 against 0.045 rebuilds per step: the whole of its deficit at 2 M particles).  With
 * this switch the four launches of sph_slab_step / sph_slab_steps become
 *     head      the books, the push of the update message, the wait for the neighbours' and the ghost update
 *     density   speculative, the criterion's check / verify jobs in its launch (sph_set_verification decides as on a single context)
 *     gate      the MAX of the rebuild word over the ranks (by its first workgroup; the others wait for it), then nothing — or the
 *               rebuild of the plain lean step AND the density pass again on the new lists
 *     force     as before
 * on = 2: the FUSED form — the head's work (books, push, wait, ghost update) by the first workgroups of the density launch itself; tiles
 * that stage ghost particles wait for them, every other tile starts at once: THREE launches per step, as sph_step, and the exchange of
 * the update message hides behind the interior tiles.  Step number, buffer parity and gravity then always come from device memory (the
 * gravity through a ring of 16 samples that sph_slab_step / sph_slab_steps keep fed).
 * Every rank of a run must use the same setting (the word is exchanged by different kernels). */
int  sph_slab_set_speculative(sph_ctx *ctx, int on);
/* bytes of one halo buffer of a slab created with these parameters and this halo_capacity (0 = default): what a host
 * needs to know before it creates the context (shared-memory transports size their mailboxes with it) */
size_t sph_slab_halo_bytes(const sph_params *prm, int halo_capacity);
/* owned particles (any order) with their global ids and accelerations; *n_out = owned count */
int  sph_slab_read(sph_ctx *ctx, sph_particle *out, uint32_t *ids, float *du_dt, float *dv_dt, int cap, int *n_out);
int  sph_slab_counts(sph_ctx *ctx, int *n_local, int *n_owned);

/* ---- metaball renderer (next row f1): draw_metaballs :380-411 + pixel grid :570-577 ----
 * 128 x 64 1-bpp SSD1306 page-format bitmap, 1024 bytes: bit (i%8) of byte (i/8)*128+j.
 * A slab context renders the pixels whose centres lie in the cell columns it owns and leaves the others 0: the bitwise
 * OR of the pages of all slabs is the frame (1 KB per rank crosses PCIe; an 8-bit OR all-reduce or a host OR joins them). */
int  sph_render_metaballs(sph_ctx *ctx, unsigned char *draw_buffer_1024);

#ifdef __cplusplus
}
#endif
#endif /* SPH_H */
