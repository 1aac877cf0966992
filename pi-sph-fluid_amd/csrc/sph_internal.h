// sph_internal.h — shared between the C-ABI layer (sph_abi.hip) and the kernels (sph_kernels.hip).
// gfx950 only; not part of the public interface (that is include/sph.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sph.h"
#include "sph_diag.h"

namespace sph {

// Derived constants, evaluated on the host the way the reference's macros are (double where the
// reference's expression is double, then rounded to f32), passed to kernels by value.
struct Consts {
    // kernel maths (pi_sph_fluid.c:45-62)
    float h, inv_h;          // H, 1/H
    float cut2;              // (2H)^2 : support test of :144 on squared distance
    // The skin of the neighbour lists is a device-side variable (Arrays::dyn, DYN_*): it adapts between skin_min and
    // skin_max at every rebuild (adapt_skin).  The grid is sized for skin_max.
    float two_h;             // 2H : the support radius (:144)
    float skin_min, skin_max;   // [m]
    float cap2;              // (H+skin_max)^2 : nobody may be further than this from its rebuild position (the walks that use
                             //              no lists look at 5x5 sort cells / 3x3 wall cells around it)
    float nf;                // 7/(4 pi H^2)          :46
    float grad_c;            // 5 nf / H^2 : -dW/dq / (d H) = grad_c * (1-q/2)^3   (:56-59 with q/d = 1/H)
    float inv_w_k2h;         // 1 / W(0.2 H)          :325
    float k1;                // 0.1                   :325
    float eps_h2;            // 0.01 H^2              :332
    float visc_c;            // 0.01 C H  (alpha c h) :332,:334
    // the list force pass's folded constants (pair_coef_ff, sph_list.inc), evaluated ONCE on the host: as expressions of the fields
    // above they were two correctly rounded square roots and a division — ~50 vector instructions — at the top of every wave
    float pair_k4;           // k1^(1/4) nf / W(0.2 H):  (pair_k4 W / nf)^4 = k1 (W / W(0.2H))^4   :325
    float pair_tq4;          // 2 pair_k4 / H
    float rho_scale;         // 1 / (-2 alpha c h): the staged densities of a tile are pre-divided (-inf without viscosity: the term is exactly 0)
    // particle / EOS (:294-301, :502)
    float m_fluid;           // RHO_0 V
    float rho0, inv_rho0;
    float B;                 // C^2 RHO_0 / 7         :297
    // integrator (:615-624, :637-640)
    float dt, half_dt;
    // neighbour grid (:82-124); cells are linearised COLUMN-major on the device:
    // cell = col * rows + row, so that a column of cells (fixed x range) is one contiguous
    // range of the sorted arrays (3 contiguous candidate ranges per particle; slab halos are
    // contiguous).  The reference is row-major (:113); the pair sets are identical.
    float x_min, y_min, cell; // cell length 2H :596, plus the skin
    float inv_cell;           // 1/cell: the device bins with a multiply (see cell_of)
    int rows, cols;          // n_cells (y), m_cells (x) :93-94 — in slab mode: the LOCAL column count (owned + 2*ghost)
    int n_cells;             // rows * cols
    // slab decomposition (single GPU: col_off = 0, ghost = 0, owned = cols, no neighbours)
    int col_off;             // global column of local column 0
    int ghost;               // ghost columns on each side (2: one exchange per step)
    int owned;               // owned columns: local columns [ghost, ghost + owned)
    int has_left, has_right; // a neighbouring slab exists on that side
    int halo_cap;            // records per halo buffer
};

// Per-context device arrays. "S" = cell-sorted state, "T" = staging written by kick/drift.
// What the rebuild-criterion jobs inside the launch of the speculative density pass need (spec_check_job, spec_verify_job:
// sph_list.inc).  In device memory behind ONE kernel argument: as a dozen arguments of their own these pointers were fetched
// at the top of the kernel by every workgroup and their scalar registers lived across the whole walk of the tiles' waves.
struct SlabFuse;
struct SpecJobs {
    const float4 *wbox;
    const uint32_t *wnbr;
    const float *dyn;
    uint32_t *flags, *vq, *rebuild;
    const uint32_t *check, *dn;
    const uint2 *lrec;
    const float2 *pos, *pos_ref;
    const float2 *vel;      // (pos and vel of ONE orientation of the two sets)
    float *uref;            // the reference displacement of this step's force pass (sample_uref; null: the absolute criterion)
    // list repair (round 5; null rq: off — slab contexts, the deterministic order): a pair the verification finds inside the support
    // and in nobody's list is APPENDED to both lists instead of asking for the rebuild (list_add, sph_list.inc)
    uint32_t *tiles;        // TileInfo records
    uint32_t *nlist;
    const unsigned short *stab;
    const uint32_t *xranges;
    uint32_t *rq;           // tiles whose lists were repaired in this step: the gate repeats their density (RQ_CAP entries); null: no repairs
                            // — or (rq_none below) a slab context: its density pass runs AFTER the head kernel's repairs, nothing to repeat
    uint2 *xpair;           // per tile lane (like lrec): up to two partners (sorted indices) of pairs this lane's particle was repaired
                            // with since the last rebuild (0xffffffff: none) — how the verification of the following steps knows, from one
                            // coalesced load, the pairs it has dealt with (they break its rule "listed exactly if within the cut-off then")
    const struct SlabFuse *fuse;   // the fused speculative slab step (sph_slab_set_speculative(ctx, 2)): what the head blocks of the density launch need; else null
    uint32_t repair;        // 0: a missing pair asks for the rebuild; 1: list repair, repaired tiles queued in rq (single-GPU contexts: the
                            // speculative density pass of the same launch may have read the lists as they were); 2: list repair, no queue
};

struct Arrays {
    // fluid, sorted (S)
    float2 *pos;       // x,y
    float2 *vel;       // u,v after the first half kick (what the force pass reads)
    float2 *pos2, *vel2; // the alternate set: the fused force kernel writes the NEXT step's kick 1/2 + drift there and
                         // sph_abi.hip swaps the two sets at the start of that step
    uint32_t *id;      // original index
    float2 *rp;        // rho, p/rho^2
    float *prs;        // p
    float2 *acc;       // du_dt, dv_dt
    uint32_t *skey;    // sorted cell keys
    float2 *pos_ref;   // x,y at the last rebuild (the positions the neighbour lists were built from)
    uint32_t *tiles;   // one TILE_WORDS-word TileInfo record per 256-particle workgroup (sph_list.inc)
    uint32_t *nlist;   // neighbour lists: per tile LROWS4 rows x 256 lanes of 4 x 8-bit entries (sph_list.inc)
    uint2 *lrec;           // per tile lane (tile * 256 + lane): .x = the lane's particle (index into the sorted arrays),
                           // .y = first LDS slot of its candidate window (list entries are relative to it) | its own LDS slot << 16
    uint32_t *tstart;      // per tile 4 words {column pair, row, index in column A, index in column B}: where the tile begins
                           // in the tile order (sph_list.inc); entry [tiles] = the end of the last tile
    uint32_t *pext;        // per column pair 2 words: first / last row that holds particles (valid for pairs that hold any)
    uint32_t *xranges;     // per tile XRANGE_WORDS: range table of the tiles with more than 8 candidate ranges
    unsigned short *stab;  // staging table: per tile STAB_ENTRIES_PER_TILE LDS slots, one per staged candidate
    // staging (T)
    float4 *pk;        // x, y, id bits, cell key bits (after kick/drift, before the sort)
    float2 *velt;      // u,v after the second half kick, sorted order: the velocity between steps (vel: after the first)
    float2 *velk;      // u,v after the first half kick, staging order (source of the sort)
    uint32_t *slot;    // arrival rank of the particle inside its cell
    // grid
    uint32_t *count;      // per-cell histogram (zero between sorts)
    uint32_t *dirty;      // one flag per scan tile (2048 cells): a histogram atomic touched it since the last scan
    uint32_t *cell_start; // n_cells + 1, exclusive scan of count
    uint32_t *block_sums; // scan scratch
    // boundary, sorted once at init
    float2 *bpos;
    float2 *bvel;       // the wall particles' stored u, v (the viscosity term reads them, :357; 0 in the reference's scenes)
    float *bpsi;
    uint32_t *bid;
    uint32_t *bcell_start; // n_cells + 1
    unsigned char *bnear;  // per cell: a wall particle lies in the 3x3 cells around it (0 -> a fluid particle there skips the wall walk)
    // misc
    float2 *grav;       // gravity vector read by the force kernel
    uint32_t *flags;    // see FLAG_*
    uint32_t *gbar;     // k_rebuild's grid barrier: GBAR_WORDS words, GBAR_STRIDE apart (arrivals, one per XCD, releases)
    uint32_t *wcast;    // the speculative slab step: WCAST_COPIES copies (GBAR_STRIDE words apart) of "step << 2 | reduced rebuild word", written by
                        // workgroup 0 of k_rebuild_slab, polled by the others — a few pollers per word (1 535 waves reading ONE word past their
                        // L2s are served one after the other: ~36 us per step, measured)
    float4 *wbox;       // per box group (BOXG consecutive lanes of a tile: a wave): bounding box of displacement since the last rebuild
    uint32_t *wnbr;     // per group: WNBR_WORDS words = 6 x {first, last} group whose particles may come near this group's
    uint32_t *latch;    // slab mode: copy of the reduced rebuild word of the current step (flags + FLAG_LATCH)
    uint32_t *check;    // word: somebody moved more than skin/2 -> k_check compares the wave boxes (single GPU)
    uint32_t *rebuild;  // the rebuild request word: flags + FLAG_REBUILD, or (slab mode) a word of the host framework
                        // that it MAX-reduces over all ranks between kick/drift and the halo pack
    SpecJobs *djobs[2]; // see SpecJobs (single-GPU contexts): [0] with pos = pos_first, [1] with the alternate set (filled once at creation)
    float2 *pos_first;  // what pos pointed at when the context was created
    uint32_t *rq;       // tiles whose lists this step's verification repaired (SpecJobs::rq; RQ_CAP entries; nullptr: no repairs)
    uint2 *xpair;       // SpecJobs::xpair (reset by the list build; nullptr: no repairs)
    uint32_t *vq;       // verification queue (spec_check_job -> spec_verify_job): [0] count, [2 + 2 e ..] = (group, failing neighbour group); nullptr: slabs
    float *uref;        // where the density pass leaves the reference displacement (dyn + DYN_UREF_X; nullptr: slab contexts)
    float *dyn;         // DYN_COUNT floats: the list cut-off and the rebuild thresholds that follow from the current skin
    uint32_t *dn;       // live counts: [0] particles in the sorted/staging arrays, [1] owned particles after kick/drift
    // slab halo buffers: uint32 header[4] (HALO_*) + halo_cap records of 5 words (x, y, u, v, id)
    uint32_t *send[2], *recv[2];   // [0] = left neighbour, [1] = right neighbour
};

enum {
    FLAG_OOB = 0,           // particles clamped into edge cells (count)
    FLAG_NAN = 1,           // NaN/Inf positions (count)
    FLAG_MAXRHO = 2,        // sph_stats: max rho bits
    FLAG_MAXSPEED = 3,      // sph_stats: max speed bits
    FLAG_CAPACITY = 4,      // slab mode: capacity overflow (count)
    FLAG_REBUILD = 5,       // set by kick/drift (or the host): the rebuild kernels of this step run; cleared by density
    FLAG_NREBUILD = 6,      // rebuilds so far
    FLAG_DIRECT_TILES = 7,  // tiles put on the direct path by list builds so far
    FLAG_MISMATCH = 8,      // slab mode: a halo message did not match the step (kind or length): ranks out of step
    FLAG_CHECK = 9,         // set by the drifting kernel: a particle is beyond skin/2, the wave boxes need comparing
    FLAG_NCHECK = 10,       // steps in which k_check ran
    FLAG_LATCH = 11,        // slab mode: the reduced rebuild word of this step, latched by k_halo_in for the final density pass
    FLAG_BAR_TIMEOUT = 12,  // k_rebuild: a grid barrier gave up waiting (its workgroups were not all resident)
    FLAG_WHY_DIRECT = 13,   // + k: tiles put on the direct path because of (k = 0) more column pairs than RMAX, (1) more rows between
                            // the first and last own row than the bitmap holds, (2) more runs or cell-table entries than fit,
                            // (3) more candidates than the LDS tile holds, (4) a window no byte can index, (5) a list longer than LROWS
    FLAG_OFF_XCD = 19,      // workgroups of one-launch rebuilds so far that did not run on the XCD of their barrier leader
    FLAG_STEP = 20,         // steps so far (k_check counts them)
    FLAG_LAST_REBUILD = 21, // FLAG_STEP at the last rebuild
    FLAG_NVERIFY = 23,      // pairs of box groups that the density pass verified particle by particle instead of asking for a rebuild
    FLAG_PEER_DONE = 22,    // workgroups of k_peer_push that have finished (grows: the last one of a launch raises the flags)
    FLAG_BAR_EPOCH = 24,    // launches that used grid barriers so far (the barrier words only grow: 8 values per such launch)
    FLAG_WHY_REBUILD = 25,  // who asked for the rebuilds (4 counters of requests, not of rebuilds): 0 the box check could not be
                            //   verified (a lane with too many failing neighbours, the queue full, no verification in this mode),
                            //   1 the verification found a pair missing from the lists, 2 somebody drifted H + skin from its sort
                            //   position (the cap), 3 rest mode: somebody beyond skin/2
    FLAG_CHECK_DONE = 29,   // check jobs of the running density launch that have finished (spec_check_job; k_rebuild clears it)
    FLAG_SAVED_WORD = 30,   // sph_time_kernel(SPH_K_DENSITY_SPEC): the rebuild word as it was before the timed launches
    FLAG_HEAD_DONE = 31,    // push blocks of k_slab_head that have finished (grows: the last one of a launch raises the flags)
    FLAG_COUNT = 32,        // (what the host reads back)
    FLAG_VERIFY_DONE = 32,  // verify blocks of the running k_slab_head that have finished (the launch after it clears it)
    FLAG_NREPAIR = 33,      // tiles queued in Arrays::rq by this step's list repairs (the first check job of the next density launch clears it)
    FLAG_REPAIRS = 34,      // pairs appended to the lists so far instead of a rebuild (list_add: two entries each)
    FLAG_REPAIR_FAIL = 35,  // + k: repairs that were not possible and asked for the rebuild after all: (0) the partner is not staged within
                            //   reach of the lane's window bytes, (1) no free byte in the lane's rows, (2) the queue of repaired tiles is full
    FLAG_HEAD_GAVE_UP = 38, // k_slab_head: its exchange block gave up waiting for the criterion's blocks of its own launch (count); that step
                            //   went out as "rebuild": exact, reported by check_flags with a message of its own
    FLAG_STEP_DONE = 39,    // slab contexts: FLAG_STEP as the last k_rebuild_slab found it = the step that launch belonged to.  Nobody writes it
                            //   while a head kernel runs, so every block of k_slab_head can derive ITS step (this + 1) from the device alone:
                            //   what lets 2^k lean steps be captured as one graph (sph_slab_steps)
    FLAG_GHOSTS_READY = 45, // the fused speculative slab step: the step whose ghost update the update blocks of the density launch have completed (grows)
    FLAG_UPD_DONE = 46,     //   ... and their completion count (grows by the number of update blocks per launch)
    FLAG_PEER_DIAG = 40,    // + 0..3: the first peer wait that gave up: site (1 k_peer_reduce, 2 k_peer_wait, 3 head: rebuild word, 4 lean: update, 5 lean: records)
                            //   << 8 | side or rank, the tag it waited for, the word it saw last, this rank's step count
    FLAG_WORDS = 64
};
// Arrays::dyn
enum {
    DYN_CUT_LIST2 = 0,      // (2H + skin)^2 : a pair enters a neighbour list below this distance (at rebuild time)
    DYN_LIM2 = 1,           // (skin/2)^2 : while nobody is further than this from its rebuild position the lists are valid
    DYN_SKIN2 = 2,          // skin^2     : ... and beyond that, while neighbouring waves moved less than this RELATIVE to each other (k_check)
    DYN_SKIN = 3,           // the skin [m] of the present lists
    DYN_UREF_X = 4,         // the reference displacement of this step (single-GPU contexts; slabs: 0): criterion (0) is
    DYN_UREF_Y = 5,         //   |u_i - U| <= skin/2 for ANY common U (then |u_i - u_k| <= skin) — see drift_verdict
    DYN_COUNT = 6
};
// the rebuild word: 0 = no rebuild; REBUILD_CRITERION = the displacement criterion asked for it (the interval since the last
// rebuild then steers the skin); REBUILD_HOST = the host did (creation, upload, sph_request_rebuild: says nothing about the flow)
enum { REBUILD_CRITERION = 1, REBUILD_HOST = 2 };
// skin controller (adapt_skin, sph_kernels.hip): cost of a rebuild / (2 x the list-dependent cost of a step at skin 0)
#ifndef SPH_ADAPT_RATIO
#define SPH_ADAPT_RATIO 5.0f
#endif
constexpr float ADAPT_RATIO = SPH_ADAPT_RATIO;
constexpr int TILE_WORDS = 32;           // 32-bit words per tile record
constexpr int WNBR_WORDS = 12;           // words per box group in Arrays::wnbr
#ifndef SPH_BOX_GROUP
#define SPH_BOX_GROUP 64                 // consecutive sorted particles per displacement box (power of two, <= 64;
                                         // 16 / 32 / 64 measured the same rebuild rates on the dam break: one per wave)
#endif
constexpr int BOXG = SPH_BOX_GROUP;
#ifndef SPH_TILE_PARTICLES
#define SPH_TILE_PARTICLES 256           // particles per tile (= threads per workgroup of the list kernels)
#endif
#ifndef SPH_XCD_CHUNKS
#define SPH_XCD_CHUNKS 8                 // chunks of consecutive tiles per XCD (xcd_tile_of, sph_list.inc); tile counts are padded to 8 x this
#endif
constexpr int XCD_CHUNKS = SPH_XCD_CHUNKS;
constexpr int LIST_WORDS_PER_TILE = 12 * SPH_TILE_PARTICLES;   // LROWS4 x TP (sph_list.inc static_asserts this)
constexpr int XRANGE_WORDS = 260;                              // 4 RMAX + 1 prefix sums, 4 RMAX first particles (sph_list.inc)
constexpr int STAB_ENTRIES_PER_TILE = 896;                     // staging-table entries per tile (sph_list.inc)
constexpr float FAR_AWAY = 1.0e9f;       // coordinate of the dummy particle list padding points at (finite: no NaN)
#ifndef SPH_VQ_CAP
#define SPH_VQ_CAP 4096
#endif
constexpr int VQ_CAP = SPH_VQ_CAP;    // pairs of groups the verification queue holds (more: rebuild) ...
// ... in contexts of up to VQ_LARGE_FROM particles; 8 x that beyond (round 5: cfg4's developed flow, 32 M particles, 608 -> 619 steps/s with
// 32 768; at 2 M particles a larger queue LOSES — 16 384: -19 % in the first window: the verify jobs take on what they cannot finish
// inside the launch).  The capacity lives in vq[1].
constexpr int VQ_LARGE_FROM = 8000000;
inline int vq_capacity(long long n) { return n >= VQ_LARGE_FROM ? 8 * VQ_CAP : VQ_CAP; }
constexpr int RQ_CAP = 2048;          // tiles one step's list repairs may queue for a repeat of their density (more: rebuild)
constexpr int HALO_HDR = 4;     // header words of a halo buffer (below)
constexpr int HALO_REC = 5;     // words per halo record
// the header: [0] particles in the UPDATE message {x, y, u, v} and [1] the step it is for — written by the kernel that drifted the
// particles (the force pass of the step before; k_kick_drift after creation / uploads); [2] full RECORDS {x, y, u, v, id} appended
// and [3] the step they are for — zeroed by k_check at the start of every step, filled by the pack of a rebuild step.  One payload.
enum { HALO_UPD_COUNT = 0, HALO_UPD_STEP = 1, HALO_REC_COUNT = 2, HALO_REC_STEP = 3 };

constexpr int GBAR_MAX_WGS = 2048, GBAR_COPIES = 32, GBAR_STRIDE = 32;      // (words: 128 bytes apart)
constexpr int WCAST_COPIES = 64;
constexpr int GBAR_WORDS = GBAR_MAX_WGS + 8 + 8 * GBAR_COPIES + 8;      // arrivals, written back (per XCD), go (per XCD, in copies), leaders' XCDs
constexpr int SCAN_ITEMS = 8;            // items per thread in the scan kernels
constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_TILE = SCAN_ITEMS * SCAN_BLOCK;   // 2048 cells per block
constexpr int SCAN_SPREAD = 8;           // counters per scan tile that the binning kernels add the tile's total into

// ---- launchers (sph_kernels.hip); all asynchronous on `st` ----
void launch_set_gravity(hipStream_t st, const Arrays &a, float gx, float gy);
// In every per-step launcher `cap` is the launch capacity (grid size, array stride); the live particle count is
// read on the device from a.dn[0], so slab mode (count changes every step) and single mode share the kernels.
// set / clear flags[FLAG_REBUILD] from the host side of the stream
void launch_set_rebuild(hipStream_t st, const Arrays &a, bool on);
// the whole rebuild chain of a step (binning, scan, scatter, tile records + lists) as ONE launch with grid barriers
// between the phases: `grid` workgroups, all resident at once (rebuild_grid).  A no-op unless the rebuild word is set.
int rebuild_grid(int device, int cap);
// selftest: only the barriers (sph_create checks that they complete on this device before it relies on them)
// spec: the launch follows the speculative density pass of the step (launch_density(..., spec = true)): it counts the step,
// clears the check word and, when the rebuild word is raised, rebuilds AND repeats the density pass on the new lists
void launch_rebuild(hipStream_t st, const Consts &c, const Arrays &a, int cap, int grid, bool selftest = false, bool deterministic = false,
                    bool spec = false);
// slab mode, what follows the halo exchange, as one launch: ghost update, or (rebuild step) ingest -> scan -> scatter ->
// canonical order of the interface cells -> tile records + lists
// the lean slab step (sph_slab_step): what its two kernels that talk to the other ranks need of the links (by value)
// Round 6: both carry the buffers of BOTH parities and may leave the step number to the device (step = 0: k_slab_head takes
// flags[FLAG_STEP_DONE] + 1, k_rebuild_slab flags[FLAG_STEP] — the head kernel of its step has counted it) — nothing in them then
// changes from step to step, and a run of steps can be captured as one graph (sph_slab_steps).
struct PeerHead {      // k_slab_head: k_check + the push of this step's update message + the MAX of the rebuild word
    uint32_t *slots_of_rank[SPH_PEER_MAX_RANKS];      // every rank's slot array (as in sph_slab_peer_reduce)
    const uint32_t *my_slots;
    uint32_t *remote_l[2], *remote_r[2];             // the neighbours' receive buffers by parity of the step (null: no neighbour)
    uint32_t *flag_l, *flag_r;                       // ... and their arrival flags
    int me, nranks;
    uint32_t step;                                   // this step's number (flags[FLAG_STEP] once the kernel has counted it); 0: from the device
};
// Round 6, the SPECULATIVE lean step (sph_slab_set_speculative): head = bookkeeping + push + the wait for the neighbours' update and the
// ghost update (no criterion, no word exchange) | density, speculative, with the criterion's jobs in its launch as in sph_step | the
// gate: MAX of the word over the ranks, then nothing — or the rebuild AND the density pass again | force.  PeerLinks then also carries
// what the word exchange needs (slots).
struct PeerLinks {     // k_rebuild_slab, lean: wait for the update | exchange the records of a rebuild step inside the launch
    uint32_t *remote_l[2], *remote_r[2], *flag_l, *flag_r;
    uint32_t *recv_l[2], *recv_r[2];                 // this rank's own receive buffers by parity (null: Arrays::recv)
    const uint32_t *my_flag_l, *my_flag_r;           // this rank's own arrival flags (what its neighbours raise)
    uint32_t step;                                   // 0: from the device
    uint32_t *slots_of_rank[SPH_PEER_MAX_RANKS];     // (speculative step: the word exchange happens in k_rebuild_slab) as in PeerHead
    const uint32_t *my_slots;
    int me, nranks;
};
// Round 6, the FUSED speculative lean step (sph_slab_set_speculative(ctx, 2)): the head kernel's work — the books, the push of the update
// message, the wait for the neighbours' and the ghost update — as the FIRST workgroups of the speculative density launch; tiles that stage
// ghost particles (TileInfo::ghost, set by the list build) wait for FLAG_GHOSTS_READY, every other tile starts at once: three launches
// per step, as sph_step, and the exchange of the update hides behind the interior tiles.  Everything per step comes from the device:
// the step number (FLAG_STEP_DONE + 1), the parity of the buffers, the gravity (a ring of 16 samples: the sample of step s at s & 15).
struct SlabFuse {
    PeerHead ph;                                     // (step = 0: from the device)
    uint32_t *recv_l[2], *recv_r[2];                 // this rank's own receive buffers by parity
    const uint32_t *my_flag_l, *my_flag_r;           // this rank's own arrival flags
    uint32_t *send_l, *send_r;
    float2 *grav;                                    // where the force pass reads the step's gravity
    const float *gring;                              // 2 x 16 floats
    int npush, nupd;                                 // head blocks of the density launch: push, update (0: no neighbours)
};
constexpr int GRAV_RING = 16;
// gravity: the step's (gx, gy) as launch arguments — or gravity_dev (device memory, two floats) when the launch is a node of a graph
// spec: the head of the speculative lean step — no criterion blocks, no word exchange; with links it also waits for the neighbours' update
// and updates the ghosts (links: the receive buffers and own flags)
void launch_slab_head(hipStream_t st, const Consts &c, const Arrays &a, int cap, const float *gravity, const PeerHead &ph, bool verify, int verify_most = 0,
                      const float *gravity_dev = nullptr, bool spec = false, const PeerLinks *links = nullptr);
void launch_peer_reduce(hipStream_t st, const Arrays &a, void *const *slots_of_rank, const void *mine, int me, int nranks, uint32_t tag);
void launch_peer_push(hipStream_t st, const Consts &c, const Arrays &a, void *remote_l, void *flag_l, void *remote_r, void *flag_r, uint32_t tag);
void launch_peer_wait(hipStream_t st, const Arrays &a, const void *flag_l, const void *flag_r, uint32_t tag);
// lean (sph_slab_step): bit 0 = the pack of a rebuild step (keys + histogram of the owned range, records into the send buffers) is
// this launch's first phase instead of k_halo_out's; bit 1 = peer transport: the launch waits for the neighbours' update itself
// and, on a rebuild step, pushes its records and waits for theirs between two of its grid barriers (links)
// lean bit 2 (speculative lean step): the launch follows the speculative density pass — it exchanges the rebuild word itself (links->slots),
// the ghost update has been done by the head kernel, and a rebuild ends with the density pass on the new lists
void launch_rebuild_slab(hipStream_t st, const Consts &c, const Arrays &a, int cap, int grid, bool deterministic = false, int lean = 0,
                         const PeerLinks *links = nullptr);
// raise the rebuild request: the next step rebuilds the neighbour structure
void launch_request_rebuild(hipStream_t st, const Arrays &a);
// measurement (sph_time_kernel): what the gate does for the criterion jobs of a speculative density launch when nothing is
// rebuilt — their completion count and queue length back to 0.  mode 0: before the timed launches (notes the rebuild word and clears
// it: a raised word sends the tiles of a speculative pass home), 1: between them, 2: after the last (the word as it was)
void launch_spec_reset(hipStream_t st, const Arrays &a, int mode);
// first half kick + drift in place (:615-624; slab mode: the owned range); requests a rebuild when the lists may be stale
void launch_kick_drift(hipStream_t st, const Consts &c, const Arrays &a, int cap, bool slab);
// slab mode: fill the send buffers — rebuild step: keys + histogram of the owned range into the staging arrays + full
// records; other steps: position/velocity updates of the interface columns
void launch_halo_out(hipStream_t st, const Consts &c, const Arrays &a, int cap);
// slab mode: consume the receive buffers — rebuild step: records join the staging arrays; other steps: ghosts updated in place
void launch_halo_in(hipStream_t st, const Consts &c, const Arrays &a, int stage_cap);
// slab mode, rebuild step: canonical (by id) particle order inside the cells of the interface columns
void launch_canon(hipStream_t st, const Consts &c, const Arrays &a);
// rebuild: keys + histogram of (pos, vsrc, id) as they are, into the staging arrays
void launch_key_only(hipStream_t st, const Consts &c, const Arrays &a, int cap, const float2 *vsrc);
// slab mode: owned particles (sorted order) -> compact AoS + ids; count left in dn[1]... see sph_abi.hip
void launch_export_owned(hipStream_t st, const Consts &c, const Arrays &a, int cap, sph_particle *out_dev, uint32_t *ids_dev,
                         float *du, float *dv);
// single GPU: if the check word is set, compare the displacement boxes of neighbouring waves; raise the rebuild word
// when two of them moved more than the skin relative to each other
// gravity != nullptr: (gx, gy) of this step, written to the device by the same launch (slab step)
void launch_check(hipStream_t st, const Consts &c, const Arrays &a, int cap, const float *gravity = nullptr);
// rebuild kernels (no-ops unless flags[FLAG_REBUILD]): scan, scatter, tile records + neighbour lists
void launch_scan(hipStream_t st, const Consts &c, uint32_t *count, uint32_t *dirty, uint32_t *cell_start,
                 uint32_t *block_sums, const uint32_t *rebuild, bool reduce);
// deterministic: the order of the particles inside a cell is by particle id (one more gated launch: the members of every
// cell are written out first) instead of the order in which the binning atomics arrived
void launch_reorder(hipStream_t st, const Consts &c, const Arrays &a, int cap, bool deterministic = false);
void launch_build_list(hipStream_t st, const Consts &c, const Arrays &a, int cap);
// variant: 0 = LDS-tiled neighbour lists (default), 1 = direct global loads over the cell ranges (A/B measurements)
// mode: what the density pass writes
enum { DENS_RHO = 0, DENS_RHO_EOS = 1 };
// consume_rebuild: this is the (final) density pass of a step, it clears the rebuild word (the rebuild kernels in front
// of it have served the request).  pass (slab mode: overlap with the halo exchange): DENS_ALL; DENS_INTERIOR = only the
// tiles that stage no ghost, and nothing at all on a rebuild step; DENS_REST = what DENS_INTERIOR left out.
enum { DENS_ALL = 0, DENS_INTERIOR = 1, DENS_REST = 2 };
// store_p = false (the passes of a step): p itself is not written (4 bytes per particle that only a read-back looks at;
// p / rho^2 is); the caller marks it stale and launch_eos restores it from rho, with the same arithmetic, on demand.
// spec = true (single-GPU step, list kernels): the pass runs BEFORE the gate of the rebuild, assumes the lists valid and
// evaluates the rebuild criterion of every box group on the way (what k_check + k_verify do as launches of their own); it
// raises the rebuild word and clears nothing: launch_rebuild(..., spec = true) must follow.  verify (spec only): two boxes that
// have moved more than the skin relative to each other are checked particle by particle; false: they ask for the rebuild.
// head_blocks (spec, slab contexts in the fused speculative step): so many workgroups in front of everything else do the slab head's work
// (a multiple of 4, at most 508; SpecJobs::fuse says what)
void launch_density(hipStream_t st, const Consts &c, const Arrays &a, int cap, int mode, int variant, bool consume_rebuild,
                    int pass = DENS_ALL, bool store_p = true, bool spec = false, bool verify = true, int head_blocks = 0);
int slab_fuse_blocks(const Consts &c, bool peers, int *npush, int *nupd);      // the head blocks a slab context's fused density launch needs
// what the force pass writes besides a: nothing / velt (second half kick) / velt + the next step's kick 1/2 + drift
// into pos2, vel2 + the next step's rebuild request
enum { FORCE_EVAL = 0, FORCE_KICK = 1, FORCE_KICK_DRIFT = 2 };
void launch_eos(hipStream_t st, const Consts &c, const Arrays &a, int cap, bool from_prs);
void launch_force(hipStream_t st, const Consts &c, const Arrays &a, int cap, int mode, int variant);
// boundary init: bin + pseudo-mass (:600-601, :242-261)
void launch_boundary_key(hipStream_t st, const Consts &c, const float2 *bpos_in, uint32_t *key, uint32_t *slot,
                         uint32_t *count, uint32_t *dirty, uint32_t *flags, int nb);
void launch_boundary_reorder(hipStream_t st, const float2 *bpos_in, const uint32_t *key, const uint32_t *slot,
                             const uint32_t *cell_start, float2 *bpos, uint32_t *bid, int nb, const float2 *bvel_in,
                             float2 *bvel, uint32_t *cell_ids_tmp);
void launch_boundary_psi(hipStream_t st, const Consts &c, const Arrays &a, int nb);
// per-cell flag "walls within reach" from the wall bins (init; again whenever the walls are re-binned)
void launch_boundary_near(hipStream_t st, const Consts &c, const Arrays &a);
void launch_fill_float2(hipStream_t st, float2 *dst, float x, float y, int n);
void launch_boundary_gather_psi(hipStream_t st, const Arrays &a, const float *psi_in_original_order, int nb);
void launch_boundary_unsort_psi(hipStream_t st, const Arrays &a, float *psi_out_original_order, int nb);
// read-back helpers
void launch_unsort_particles(hipStream_t st, const Consts &c, const Arrays &a, int n, sph_particle *out_dev);
void launch_unsort_accel(hipStream_t st, const Arrays &a, int n, float *du, float *dv);
void launch_gather_accel(hipStream_t st, const Arrays &a, int n, const float *du, const float *dv);
void launch_unsort_boundary(hipStream_t st, const Consts &c, const Arrays &a, int nb, sph_particle *out_dev);
void launch_upload_state(hipStream_t st, const Arrays &a, int n, const sph_particle *in_dev);
void launch_gather_rho_p(hipStream_t st, const Consts &c, const Arrays &a, int n, const sph_particle *in_dev);
void launch_stats(hipStream_t st, const Consts &c, const Arrays &a, int n, bool slab);
// velt := vel + dt/2 acc (the velocity between steps, which the fused force pass does not store)
void launch_refresh_velt(hipStream_t st, const Consts &c, const Arrays &a, int cap);
void launch_metaballs(hipStream_t st, const Consts &c, const Arrays &a, float width, float height,
                      unsigned char *page_bytes_dev /* 1024 bytes, SSD1306 page format */);

}  // namespace sph
