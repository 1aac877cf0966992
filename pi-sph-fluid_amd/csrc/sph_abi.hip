// sph_abi.hip — the C ABI of include/sph.h on top of the gfx950 kernels.
//
// Host-side orchestration only: device memory, the HIP stream, the captured step graph,
// read-backs.  All arithmetic on particles happens in sph_kernels.hip.  There is no CPU
// compute path: every entry point that needs a GPU fails with SPH_E_HIP when none is usable.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <unistd.h>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <atomic>

#include "sph.h"
#include "sph_diag.h"
#include "sph_internal.h"

using namespace sph;

struct sph_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    sph_params prm{};
    Consts c{};
    Arrays a{};
    int n = 0, nb = 0;          // single mode: fluid / boundary counts; slab mode: n = particles passed at creation
    int cap = 0;                // capacity of the particle arrays (== n in single mode)
    bool slab = false;
    int slab_phase = 0;         // 0 idle, 1 after sph_slab_step_begin, 2 after sph_slab_step_pack
    bool slab_overlapped = false; // sph_slab_step_overlap ran in this step: step_end's density does the rest only
    bool has_links = false;      // sph_slab_set_peer_links: the lean step (sph_slab_step) talks to the other ranks itself
    sph_peer_links links{};
    uint32_t lean_step = 0;      // steps taken by sph_slab_step / counted by the device (FLAG_STEP): the tags of its messages
    // sph_slab_steps: the gravity samples of a graphed run of steps (device: 2 x MULTI_STEPS floats, read by the head kernels of the
    // graph's nodes; host: a ring of pinned staging slots, an event per slot; what the device holds now)
    float *d_gseq = nullptr, *h_gseq = nullptr;
    float h_gseq_last[2 * 64] = {};
    bool gseq_valid = false, gseq_used[8] = {};
    hipEvent_t gseq_ev[8] = {};
    unsigned gseq_slot = 0;
    int lean_spec = 0;           // sph_slab_set_speculative: 1 = the lean step with the criterion inside the launch of a speculative density pass;
                                 // 2 = ... and the head kernel's work as the first workgroups of that launch (three launches per step)
    SlabFuse *d_fuse = nullptr;  // mode 2: what those workgroups need (device record) and the gravity ring they read (2 x GRAV_RING floats)
    float *d_gring = nullptr;
    float h_gring[2 * GRAV_RING] = {};      // the ring as the device holds it (or will, in stream order)
    bool gring_valid = false;
    int fuse_blocks = 0;
    bool step_done_synced = true;   // FLAG_STEP_DONE == FLAG_STEP between steps (k_rebuild_slab keeps it; the per-phase kernels do not)
    bool own_halo = false;      // halo buffers allocated by the library (else adopted from the host framework)
    size_t halo_bytes = 0;
    uint32_t *d_ids = nullptr;  // slab read-back staging
    int variant = 0;
    float skin = 0.0f;          // absolute skin [m] of this context's neighbour lists
    sph_particle *d_aos = nullptr;    // n  : read-back / upload staging, original order
    sph_particle *d_baos = nullptr;   // nb
    float *d_du = nullptr, *d_dv = nullptr;
    float2 *d_bpos_in = nullptr, *d_bvel_in = nullptr;   // nb : wall particles as given (original order), staging of the wall bins
    uint32_t *d_bkey = nullptr;       // nb : their cells
    float *d_bpsi0 = nullptr;         // nb : pseudo-mass in original order (kept across sph_update_boundary)
    uint32_t *d_bcell_ids = nullptr;   // wall particles: the members of every cell in slot order (deterministic order inside a cell)
    unsigned char *d_bits = nullptr;  // metaball frame, 1024 bytes of SSD1306 page format
    std::vector<void *> allocs;
    size_t bytes = 0;
    // Single-GPU step with the list kernels: the force pass of step s also does step s+1's kick 1/2 + drift into the
    // alternate arrays (a.pos2, a.vel2); step s+1 starts by swapping the two sets.  primed: the alternate set holds
    // that look-ahead (false after creation / upload / eval_accel / variant change: the next step then starts with
    // the stand-alone kick/drift kernel instead).  One captured graph per orientation of the two sets.
    bool primed = false;
    bool p_stale = false;        // the density passes of a step do not store p: refresh_p() before use
    bool velt_stale = false;     // the fused force pass does not store the velocity between steps: refresh_velt() before use
    bool acc_stale = false;      // ... nor the acceleration: refresh_acc() before use, and before anything it depends on changes
    bool stepped = false;        // a step has run since creation / upload / sph_eval_accel (sph_time_kernel needs it)
    float2 *pos_a = nullptr;     // the array a.pos pointed at when the context was created (graph index 0)
    hipGraph_t graph[16] = {};         // [0,1]: one step; [2 j, 2 j + 1]: 2^j steps, j = 1 .. 6 (x the two orientations)
    hipGraphExec_t gexec[16] = {};
    int slab_verify_most = 0;          // slab contexts: the most queued group pairs the head kernel verifies (more: the rebuild); 0 = the queue's capacity ($SPH_SLAB_VERIFY_MOST)
    int verify_mode = -1;              // failing box pairs verified particle by particle (spec_verify_job): -1 = from VERIFY_MIN_PARTICLES on, 0 = never (they ask for the rebuild), 1 = always (sph_set_verification)
    bool use_graph = true;
    int rebuild_wgs = 0;         // > 0: the rebuild chain of a step is ONE launch of this many workgroups (k_rebuild)
    bool one_launch_asked = false;  // sph_set_rebuild_launches(ctx, 1): the host vouches that nothing else computes on the device meanwhile
    bool counted = false;        // this context is in g_live_contexts
    bool deterministic = false;  // particles of a cell in id order (sph_set_deterministic)
    int repair_mode = -1;        // list repair (sph_set_list_repair): -1 = from REPAIR_MIN_PARTICLES on, 0 = never, 1 = always (default order only)
    hipEvent_t ev[SPH_K_COUNT + 2] = {};
    long long oob_total = 0, nan_total = 0;
    std::string err;
};

namespace {

// Contexts of this process per device.  k_rebuild needs all its workgroups resident at once; two of them running
// concurrently (two contexts stepped from one host thread, each on its own stream) could each hold half the device and
// wait for the other half: a context that finds company on its device goes back to one kernel per phase.
constexpr int MAX_DEVICES = 64;
constexpr int GSEQ_SLOTS = 8;      // pinned staging slots of sph_slab_steps' gravity samples
std::atomic<int> g_live_contexts[MAX_DEVICES];
bool device_shared(const sph_ctx *ctx);
int upload_fuse(sph_ctx *ctx);      // (the fused speculative slab step: defined with the lean step below)

int fail(sph_ctx *ctx, int code, const char *what, hipError_t e = hipSuccess) {
    if (ctx) {
        char buf[512];
        if (e != hipSuccess) snprintf(buf, sizeof buf, "%s: %s (%s)", what, hipGetErrorString(e), sph_error_string(code));
        else snprintf(buf, sizeof buf, "%s (%s)", what, sph_error_string(code));
        ctx->err = buf;
    }
    return code;
}

#define HIPCHK(ctx, call)                                                         \
    do {                                                                          \
        hipError_t e_ = (call);                                                   \
        if (e_ != hipSuccess) return fail((ctx), SPH_E_HIP, #call, e_);           \
    } while (0)

template <typename T>
int dalloc(sph_ctx *ctx, T **p, size_t count) {
    size_t b = (count ? count : 1) * sizeof(T);
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, b);
    if (e != hipSuccess) return fail(ctx, e == hipErrorOutOfMemory ? SPH_E_NOMEM : SPH_E_HIP, "hipMalloc", e);
    ctx->allocs.push_back(q);
    ctx->bytes += b;
    *p = static_cast<T *>(q);
    return SPH_OK;
}

// derived constants, evaluated like the reference's macros (double where its expression is double)
int make_consts(const sph_params &p, Consts &c) {
    const float skin_frac = p.skin;      // per context (sph_params.skin), a fraction of 2H
    if (!(skin_frac >= 0.0f && skin_frac <= 1.0f)) return SPH_E_ARG;
    if (!(p.h > 0) || !(p.r > 0) || !(p.rho0 > 0) || !(p.c > 0) || !(p.dt > 0) || !(p.vol > 0)) return SPH_E_ARG;
    if (!(p.x_max > p.x_min) || !(p.y_max > p.y_min)) return SPH_E_ARG;
    const double H = p.h;
    const double nf = 7.0 / (4.0 * M_PI * H * H);                       // :46
    c.h = p.h;
    c.inv_h = 1.0f / p.h;
    const float two_h = 2 * p.h;                                         // :144
    c.cut2 = two_h * two_h;
    c.two_h = two_h;
    const float skin = skin_frac * two_h;                                // the largest skin: what the grid is sized for
    c.skin_max = skin;
    c.skin_min = p.skin_min >= 0.0f && p.skin_min < skin_frac ? p.skin_min * two_h : skin;      // (>= skin: fixed)
    c.cap2 = (p.h + skin) * (p.h + skin) * 0.999f;                       // (rounding of the squared distances stays on the safe side)
    c.nf = (float)nf;
    c.grad_c = (float)(5.0 * nf / (H * H));                              // :56-59
    const double q = p.k2;                                               // W(0.2 H): q = 0.2   :325
    const double t = 1.0 - 0.5 * q;
    c.inv_w_k2h = (float)(1.0 / (nf * t * t * t * t * (1.0 + 2.0 * q)));
    c.k1 = p.k1;
    c.eps_h2 = (float)((double)p.eps * H * H);                           // :332
    c.visc_c = (float)((double)p.alpha * (double)p.c * H);               // :332, :334
    c.pair_k4 = sqrtf(sqrtf(c.k1)) * (c.nf * c.inv_w_k2h);               // (f32, the expressions the kernel used to evaluate per wave)
    c.pair_tq4 = 2.0f * c.inv_h * c.pair_k4;
    c.rho_scale = 1.0f / (-2.0f * c.visc_c);
    c.m_fluid = p.rho0 * p.vol;                                          // :502
    c.rho0 = p.rho0;
    c.inv_rho0 = 1.0f / p.rho0;
    c.B = p.c * p.c * p.rho0 / 7;                                        // :297
    c.dt = p.dt;
    c.half_dt = (float)(0.5 * (double)p.dt);                             // :616
    c.x_min = p.x_min;
    c.y_min = p.y_min;
    c.cell = two_h + skin;                                               // :596 (2H) + skin
    c.inv_cell = 1.0f / c.cell;
    double rows = (double)(int)((p.y_max - p.y_min) / c.cell) + 1;       // :93
    double cols = (double)(int)((p.x_max - p.x_min) / c.cell) + 1;       // :94
    if (rows * cols > 1.0e9) return SPH_E_ARG;
    c.rows = (int)rows;
    c.cols = (int)cols;
    c.n_cells = c.rows * c.cols;
    c.col_off = 0;
    c.ghost = 0;
    c.owned = c.cols;
    c.has_left = c.has_right = 0;
    c.halo_cap = 0;
    return SPH_OK;
}

size_t padded_items(const Consts &c) {
    size_t n_items = (size_t)c.n_cells + 1;
    return (n_items + SCAN_TILE - 1) / SCAN_TILE * SCAN_TILE;
}

bool fused(const sph_ctx *ctx) { return ctx->variant == 0; }

bool device_shared(const sph_ctx *ctx) {
    return ctx->device >= 0 && ctx->device < MAX_DEVICES && g_live_contexts[ctx->device].load(std::memory_order_relaxed) > 1;
}

// the velocity after the second half kick (:638-639): the fused step leaves it to be recomputed on demand
// p from the stored rho, as the density pass of the last step computed it (and p / rho^2 with it: the same value again)
void refresh_p(sph_ctx *ctx) {
    if (ctx->p_stale) launch_eos(ctx->stream, ctx->c, ctx->a, ctx->cap, false);
    ctx->p_stale = false;
}
// a of the last step (:303-373): the fused force pass has used it for its kicks and has not stored it.  The state it was
// computed from — positions, half-kicked velocities, rho, p, walls, gravity, lists — is untouched until the next step or until
// a caller changes one of them (every entry point that does calls this first), and the same kernel in its FORCE_EVAL form
// walks the same lists in the same order.
void refresh_acc(sph_ctx *ctx) {
    if (ctx->acc_stale) launch_force(ctx->stream, ctx->c, ctx->a, ctx->cap, FORCE_EVAL, ctx->variant);
    ctx->acc_stale = false;
}
void refresh_velt(sph_ctx *ctx) {
    refresh_acc(ctx);
    if (ctx->velt_stale) launch_refresh_velt(ctx->stream, ctx->c, ctx->a, ctx->cap);
    ctx->velt_stale = false;
}

// What a step launches after its kick/drift.
//   The step of sph_step (single GPU, list kernels, the one-launch rebuild available): THREE kernels —
//     density + EOS, speculative: it assumes the lists valid and evaluates the rebuild criterion of every box group on the way
//       (check_inline / verify_inline in sph_list.inc: what k_check and k_verify do as launches of their own);
//     k_rebuild: the gate — nothing to do in ~95 % of the steps; else binning, scan, scatter, lists AND the density pass again;
//     force + kick + the next step's kick 1/2 + drift (FORCE_KICK_DRIFT), which leaves the displacement boxes for the next check.
//   The legacy order (ev != nullptr: the profiled step, an event before each kernel in SPH_K_* order; contexts that may share
//   their device; the direct variant): k_check [+ k_verify], the rebuild as one kernel per phase, density, force.
constexpr int REPAIR_MIN_PARTICLES = 4000000;     // list repair instead of a rebuild (list_add, sph_list.inc): from this many particles on (sph_set_list_repair)
constexpr int VERIFY_MIN_PARTICLES = 500000;      // the verify jobs of the density launch: from this many particles on (sph_set_verification)
bool speculative(const sph_ctx *ctx) { return !ctx->slab && fused(ctx) && ctx->rebuild_wgs > 0; }
// Slab contexts: failing boxes verified particle by particle by blocks of the head kernel (k_slab_head) — ONLY when the host asks
// (sph_set_verification(ctx, 1)).  In sph_step the criterion's jobs ride inside the density launch and cost the step nothing; the slab
// step's head kernel is on its critical path: measured with one slab through the C host (bench.py, slab_overhead, 2 M particles), the
// verification made the collapse windows 5 % faster (fewer rebuilds) and every other window 10 - 15 % slower.  Every rank may choose
// for itself: the rebuild word is MAX-reduced, all ranks rebuild in the same steps whoever verifies.
bool slab_verifies(const sph_ctx *ctx) { return fused(ctx) && ctx->verify_mode > 0; }
bool list_repair(const sph_ctx *ctx) { return !ctx->deterministic && (ctx->repair_mode < 0 ? ctx->n >= REPAIR_MIN_PARTICLES : ctx->repair_mode > 0); }
void enqueue_step_body(sph_ctx *ctx, hipEvent_t *ev) {
    hipStream_t st = ctx->stream;
    if (speculative(ctx) && !ev) {
        // (failing box pairs verified particle by particle: where a rebuild is expensive, i.e. with many particles — 262 144
        // particles, cfg1: the verify jobs took the launch from 6 to 29 us to save rebuilds of 50 us in one step of a hundred)
        launch_density(st, ctx->c, ctx->a, ctx->cap, DENS_RHO_EOS, ctx->variant, false, DENS_ALL, false, true,
                       ctx->verify_mode < 0 ? ctx->n >= VERIFY_MIN_PARTICLES : ctx->verify_mode > 0);
        Arrays ga = ctx->a;
        if (!list_repair(ctx)) {      // no repairs: no queue of repaired tiles (the gate's workgroups do not look for one), and the list build
            ga.rq = nullptr;          // writes neither the remembered partners (8 B per particle) nor the spare row of padding (4 B): at
            ga.xpair = nullptr;       // 2 M particles 24 MB per rebuild that only a repair would read
        }
#ifndef SPH_UNSAFE_NO_GATE      // (measurement build, results NOT exact once a rebuild is due — the bound for any scheme that drops the gate's launch
                                // from steps that do not rebuild: make variant NAME=nogate VFLAGS=-DSPH_UNSAFE_NO_GATE)
        launch_rebuild(st, ctx->c, ga, ctx->cap, ctx->rebuild_wgs, false, ctx->deterministic, true);
#endif
        launch_force(st, ctx->c, ctx->a, ctx->cap, FORCE_KICK_DRIFT, ctx->variant);
        return;
    }
    if (ev) (void)hipEventRecord(ev[SPH_K_KEY_HIST], st);
    // beyond skin/2: do neighbouring groups still move together?  (boxes only: two that have moved more than the skin relative
    // to each other ask for the rebuild)
    if (!ctx->slab) launch_check(st, ctx->c, ctx->a, ctx->cap, nullptr);
    // (one kernel per phase)
    launch_key_only(st, ctx->c, ctx->a, ctx->cap, ctx->a.vel);
    if (ev) (void)hipEventRecord(ev[SPH_K_SCAN], st);
    launch_scan(st, ctx->c, ctx->a.count, ctx->a.dirty, ctx->a.cell_start, ctx->a.block_sums, ctx->a.rebuild, false);
    if (ev) (void)hipEventRecord(ev[SPH_K_REORDER], st);
    launch_reorder(st, ctx->c, ctx->a, ctx->cap, ctx->deterministic);
    if (ev) (void)hipEventRecord(ev[SPH_K_BUILD_LIST], st);
    launch_build_list(st, ctx->c, ctx->a, ctx->cap);
    if (ev) (void)hipEventRecord(ev[SPH_K_DENSITY_EOS], st);
    launch_density(st, ctx->c, ctx->a, ctx->cap, DENS_RHO_EOS, ctx->variant, true, DENS_ALL, false);
    if (ev) (void)hipEventRecord(ev[SPH_K_FORCE_KICK], st);
    launch_force(st, ctx->c, ctx->a, ctx->cap, fused(ctx) ? FORCE_KICK_DRIFT : FORCE_KICK, ctx->variant);
    if (ev) (void)hipEventRecord(ev[SPH_K_HALO], st);   // = end of step
}

void drop_graph(sph_ctx *ctx) {
    for (int k = 0; k < 16; k++) {
        if (ctx->gexec[k]) { (void)hipGraphExecDestroy(ctx->gexec[k]); ctx->gexec[k] = nullptr; }
        if (ctx->graph[k]) { (void)hipGraphDestroy(ctx->graph[k]); ctx->graph[k] = nullptr; }
    }
}

// the step body captured into a graph for the current orientation of the two position/velocity sets
// (launch-latency bound at small N; replay costs one submission)
hipGraphExec_t step_graph(sph_ctx *ctx) {
    if (!ctx->use_graph) return nullptr;
    const int k = ctx->a.pos == ctx->pos_a ? 0 : 1;
    if (ctx->gexec[k]) return ctx->gexec[k];
    if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        ctx->use_graph = false;
        return nullptr;
    }
    enqueue_step_body(ctx, nullptr);
    hipError_t e = hipStreamEndCapture(ctx->stream, &ctx->graph[k]);
    if (e == hipSuccess) e = hipGraphInstantiate(&ctx->gexec[k], ctx->graph[k], nullptr, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        drop_graph(ctx);
        ctx->use_graph = false;
        return nullptr;
    }
    return ctx->gexec[k];
}

// MULTI_STEPS consecutive steps of the fused (primed) loop as ONE graph: a replay has a fixed cost of several
// microseconds whatever it holds, and sph_step(nsteps) usually asks for many steps.  An even number of steps leaves the
// orientation of the two position / velocity sets as it was.
#ifndef SPH_MULTI_STEPS
#define SPH_MULTI_STEPS 16
#endif
constexpr int MULTI_STEPS = SPH_MULTI_STEPS;      // the largest (a power of two, at most 64); the smaller powers serve the remainder of a call (20 steps = 16 + 4: two replays).  Round 5, same box: 8 -> 16: cfg1 48.0k -> 49.0k steps/s, cfg0 +1.3 %, cfg2 +0.3 %, the 20-step window unchanged; 32: no more
static_assert(MULTI_STEPS >= 2 && MULTI_STEPS <= 64 && (MULTI_STEPS & (MULTI_STEPS - 1)) == 0, "MULTI_STEPS");
hipGraphExec_t multi_graph(sph_ctx *ctx, int steps = MULTI_STEPS) {
    if (!ctx->use_graph || steps < 2 || steps > MULTI_STEPS || (steps & (steps - 1)) != 0) return nullptr;
    const int k = 2 * (31 - __builtin_clz((unsigned)steps)) + (ctx->a.pos == ctx->pos_a ? 0 : 1);
    if (ctx->gexec[k]) return ctx->gexec[k];
    if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        ctx->use_graph = false;
        return nullptr;
    }
    for (int s = 0; s < steps; s++) {
        std::swap(ctx->a.pos, ctx->a.pos2);
        std::swap(ctx->a.vel, ctx->a.vel2);
        enqueue_step_body(ctx, nullptr);
    }
    hipError_t e = hipStreamEndCapture(ctx->stream, &ctx->graph[k]);
    if (e == hipSuccess) e = hipGraphInstantiate(&ctx->gexec[k], ctx->graph[k], nullptr, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        drop_graph(ctx);
        ctx->use_graph = false;
        return nullptr;
    }
    return ctx->gexec[k];
}

// The step graphs for both orientations of the position / velocity sets, captured, instantiated and uploaded at context
// creation: the first sph_step calls of a host are often the ones it times, and instantiating a 24-kernel graph inside
// them cost ~0.3 ms.  Not for contexts that cannot use them as they are: slabs, the direct variant, and contexts that find
// company on their device (their first sph_step falls back to one kernel per phase and drops the graphs).
void prebuild_graphs(sph_ctx *ctx) {
    if (ctx->slab || !ctx->use_graph || !fused(ctx) || device_shared(ctx)) return;
    for (int o = 0; o < 2 && ctx->use_graph; o++) {      // both orientations
        hipGraphExec_t g = step_graph(ctx);
        if (g) (void)hipGraphUpload(g, ctx->stream);
        for (int m = MULTI_STEPS; m >= 2; m >>= 1) {
            g = multi_graph(ctx, m);
            if (g) (void)hipGraphUpload(g, ctx->stream);
        }
        std::swap(ctx->a.pos, ctx->a.pos2);
        std::swap(ctx->a.vel, ctx->a.vel2);
    }
    (void)hipGetLastError();
}

// one time step (:612-641).  Kick 1/2 + drift: already done by the previous step's force pass (swap the sets), or
// the stand-alone kernel.
int run_step(sph_ctx *ctx, hipEvent_t *ev) {
    hipStream_t st = ctx->stream;
    if (ev) (void)hipEventRecord(ev[SPH_K_KICK_DRIFT], st);
    if (fused(ctx) && ctx->primed) {
        std::swap(ctx->a.pos, ctx->a.pos2);
        std::swap(ctx->a.vel, ctx->a.vel2);
    } else {
        refresh_velt(ctx);
        launch_kick_drift(st, ctx->c, ctx->a, ctx->cap, false);
    }
    ctx->primed = fused(ctx);
    ctx->velt_stale = fused(ctx);
    ctx->acc_stale = fused(ctx);
    ctx->p_stale = true;
    ctx->stepped = true;
    hipGraphExec_t g = ev ? nullptr : step_graph(ctx);
    if (g) HIPCHK(ctx, hipGraphLaunch(g, st));
    else enqueue_step_body(ctx, ev);
    return SPH_OK;
}

// rebuild from the current (pos, velt, id) state: keys + histogram, scan, scatter, neighbour lists; then velt := sorted vel
int resort_state(sph_ctx *ctx) {
    hipStream_t st = ctx->stream;
    launch_set_rebuild(st, ctx->a, true);
    launch_key_only(st, ctx->c, ctx->a, ctx->cap, ctx->a.velt);
    launch_scan(st, ctx->c, ctx->a.count, ctx->a.dirty, ctx->a.cell_start, ctx->a.block_sums, ctx->a.rebuild, false);
    launch_reorder(st, ctx->c, ctx->a, ctx->cap, ctx->deterministic);
    if (ctx->slab) launch_canon(st, ctx->c, ctx->a);
    launch_build_list(st, ctx->c, ctx->a, ctx->cap);
    launch_set_rebuild(st, ctx->a, false);
    HIPCHK(ctx, hipMemcpyAsync(ctx->a.velt, ctx->a.vel, sizeof(float2) * (size_t)ctx->cap, hipMemcpyDeviceToDevice, st));
    return SPH_OK;
}

// what the rebuild-criterion jobs read (SpecJobs, sph_internal.h), one record per orientation of the two position / velocity sets:
// at creation, and again when a slab adopts a rebuild word of its host's
int upload_jobs(sph_ctx *ctx, bool no_repair = false) {      // no_repair: sph_time_kernel's launches must not change the lists
    Arrays &a = ctx->a;
    if (!a.djobs[0]) return SPH_OK;
    float2 *const first = a.pos_first, *const other = a.pos == a.pos_first ? a.pos2 : a.pos;
    float2 *const vfirst = a.pos == a.pos_first ? a.vel : a.vel2, *const vother = a.pos == a.pos_first ? a.vel2 : a.vel;
    // (list repair: single-GPU contexts in their default order — appended entries land where the races of the verify waves put them:
    // not for a context that promises the same bits every run)
    // From REPAIR_MIN_PARTICLES on by default.  What a repair costs does not depend on the size of the scene — the gate repeats the density
    // of the repaired tiles, ~10 us in a step that has any — what it saves does: a rebuild of 2 M particles is 280 us, of 32 M 4.2 ms.
    // Measured (tools/ab_env5.sh, same box): cfg4's developed flow 458 -> 583 steps/s (rebuilds in 17 % -> 6 % of the steps); cfg2's
    // collapse (steps 1200-2200, repairs in nearly every step) 8 660 -> 7 940, its other windows unchanged.
    // Slab contexts repair too (the verification blocks of their head kernel, k_slab_head): their density pass runs after the head
    // kernel, on the repaired lists — there is nothing to repeat and no queue.
    const bool repair = list_repair(ctx) && !no_repair;
    const bool queued = !ctx->slab || ctx->lean_spec != 0;      // (the density pass runs beside the repairs: their tiles are queued for a repeat)
    uint32_t *rq = repair && queued ? a.rq : nullptr;
    const uint32_t repair_kind = !repair ? 0u : queued ? 1u : 2u;
    const SpecJobs j0 = {a.wbox, a.wnbr, a.dyn, a.flags, a.vq, a.rebuild, a.check, a.dn, a.lrec, first, a.pos_ref, vfirst, a.uref,
                         a.tiles, a.nlist, a.stab, a.xranges, rq, a.xpair, ctx->slab && ctx->lean_spec == 2 ? ctx->d_fuse : nullptr, repair_kind};
    SpecJobs j1 = j0;
    j1.pos = other;
    j1.vel = vother;
    HIPCHK(ctx, hipMemcpyAsync(a.djobs[0], &j0, sizeof j0, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(a.djobs[1], &j1, sizeof j1, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));      // (locals)
    return SPH_OK;
}

int check_flags(sph_ctx *ctx) {
    uint32_t h[FLAG_HEAD_GAVE_UP + 1] = {0};
    static_assert(FLAG_HEAD_GAVE_UP + 1 >= FLAG_COUNT && FLAG_HEAD_GAVE_UP < FLAG_WORDS, "one read-back for every word checked here");
    HIPCHK(ctx, hipMemcpyAsync(h, ctx->a.flags, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (h[FLAG_HEAD_GAVE_UP]) {      // (not a grid barrier of the rebuild: rebuild_wgs and the graphs stay as they are)
        HIPCHK(ctx, hipMemsetAsync(ctx->a.flags + FLAG_HEAD_GAVE_UP, 0, sizeof(uint32_t), ctx->stream));
        return fail(ctx, SPH_E_STATE, "slab step: the exchange block of the head kernel gave up waiting for the rebuild criterion's blocks of its own "
                                      "launch (were they not all resident: is another process computing on this device?); those steps went out as "
                                      "'rebuild' and are exact - the state is valid, the step is slower than it should be");
    }
    if (h[FLAG_BAR_TIMEOUT]) {
        HIPCHK(ctx, hipMemsetAsync(ctx->a.flags + FLAG_BAR_TIMEOUT, 0, sizeof(uint32_t), ctx->stream));
        if (h[FLAG_BAR_TIMEOUT] & 2u) {      // (the peer transport's waits have a bit of their own: nothing is wrong with this rank's launches)
            uint32_t d[4] = {0, 0, 0, 0};
            HIPCHK(ctx, hipMemcpy(d, ctx->a.flags + FLAG_PEER_DIAG, sizeof d, hipMemcpyDeviceToHost));
            static const char *const site[] = {"?", "rebuild word (sph_slab_peer_reduce), rank", "halo (sph_slab_peer_wait), side", "rebuild word (head of sph_slab_step), rank",
                                               "ghost update (sph_slab_step), side", "records (sph_slab_step), side"};
            char buf[384];
            snprintf(buf, sizeof buf, "peer transport: a neighbouring rank's rebuild word or halo did not arrive within the time limit (is every rank "
                     "stepping?) - first to give up: the wait for the %s %u, tag 0x%x, word last seen 0x%x, step %u; the state is invalid, upload it again",
                     site[(d[0] >> 8) < 6u ? (d[0] >> 8) : 0u], d[0] & 255u, d[1], d[2], d[3]);
            return fail(ctx, SPH_E_STATE, buf);
        }
        ctx->rebuild_wgs = 0;
        drop_graph(ctx);
        return fail(ctx, SPH_E_STATE, "the one-launch rebuild gave up at a grid barrier: its workgroups were not all resident (is another "
                                      "process computing on this device?); the state is invalid, upload it again");
    }
    if (h[FLAG_OOB] | h[FLAG_NAN] | h[FLAG_CAPACITY] | h[FLAG_MISMATCH]) {
        HIPCHK(ctx, hipMemsetAsync(ctx->a.flags, 0, 2 * sizeof(uint32_t), ctx->stream));
        HIPCHK(ctx, hipMemsetAsync(ctx->a.flags + FLAG_CAPACITY, 0, sizeof(uint32_t), ctx->stream));
        HIPCHK(ctx, hipMemsetAsync(ctx->a.flags + FLAG_MISMATCH, 0, sizeof(uint32_t), ctx->stream));
        ctx->oob_total += h[FLAG_OOB];
        ctx->nan_total += h[FLAG_NAN];
        if (h[FLAG_NAN]) return fail(ctx, SPH_E_NAN, "particle positions became NaN/Inf");
        // (capacity first: a slab that dropped particles also puts its neighbours out of step)
        if (h[FLAG_CAPACITY]) return fail(ctx, SPH_E_CAPACITY, "slab particle or halo capacity exceeded");
        if (h[FLAG_MISMATCH])
            return fail(ctx, SPH_E_STATE, "halo message does not match this step (neighbouring slabs out of step, or the rebuild "
                                          "word was not reduced over all ranks before sph_slab_step_pack)");
        return fail(ctx, SPH_E_OUT_OF_DOMAIN, ctx->slab ? "particles left the slab's local grid and were clamped into edge cells"
                                                        : "particles left the domain and were clamped into edge cells");
    }
    return SPH_OK;
}

// The one-launch rebuild needs all its workgroups resident at once.  Try its barriers once, now: where they do not
// complete (compute units masked off or held by somebody else), this context rebuilds with one kernel per phase.
int selftest_one_launch(sph_ctx *ctx) {
    // (in company a single-GPU context does not use it: sph_step; a slab context uses it because its host said so)
    if (ctx->rebuild_wgs <= 0 || (device_shared(ctx) && !(ctx->slab && ctx->one_launch_asked))) return SPH_OK;
    hipStream_t st = ctx->stream;
    launch_rebuild(st, ctx->c, ctx->a, ctx->cap, ctx->rebuild_wgs, true);
    uint32_t timed_out = 0;
    HIPCHK(ctx, hipMemcpyAsync(&timed_out, ctx->a.flags + FLAG_BAR_TIMEOUT, sizeof timed_out, hipMemcpyDeviceToHost, st));
    // (the next launch waits for the same barrier values again: the rebuild count has not moved)
    HIPCHK(ctx, hipMemsetAsync(ctx->a.gbar, 0, sizeof(uint32_t) * (size_t)GBAR_WORDS * GBAR_STRIDE, st));
    HIPCHK(ctx, hipMemsetAsync(ctx->a.flags + FLAG_BAR_TIMEOUT, 0, sizeof(uint32_t), st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    if (timed_out) ctx->rebuild_wgs = 0;
    return SPH_OK;
}

int select_device(sph_ctx *ctx, int device) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        return fail(ctx, SPH_E_HIP, "no HIP device available (this library has no CPU path)", e);
    }
    if (device < 0 || device >= count) return fail(ctx, SPH_E_ARG, "device ordinal out of range");
    HIPCHK(ctx, hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(ctx, hipGetDeviceProperties(&prop, device));
    if (!strstr(prop.gcnArchName, "gfx950"))
        return fail(ctx, SPH_E_HIP, "device is not gfx950 (MI355X); kernels are built for gfx950 only");
    ctx->device = device;
    return SPH_OK;
}

}  // namespace

extern "C" {

int sph_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return count;
}

const char *sph_last_error(const sph_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

void sph_destroy(sph_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->counted) g_live_contexts[ctx->device].fetch_sub(1);
    drop_graph(ctx);
    for (auto &e : ctx->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto &e : ctx->gseq_ev)
        if (e) (void)hipEventDestroy(e);
    if (ctx->h_gseq) (void)hipHostFree(ctx->h_gseq);
    for (void *p : ctx->allocs) (void)hipFree(p);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

}  // extern "C"

namespace {

struct SlabSpec {          // internal form of sph_slab_desc
    int col_begin, col_end, has_left, has_right, halo_cap, particle_cap;
};

// Shared by sph_create and sph_create_slab: allocate, bin the boundary (psi given or computed), upload the fluid,
// bin it, evaluate rho, p, a at t = 0 (:594-607).  ids == nullptr: ids are 0..n-1.
int init_context(sph_ctx *ctx, const sph_params *prm, const sph_particle *fluid, const uint32_t *ids, int n_fluid,
                 const sph_particle *boundary, int n_boundary, bool psi_given, float gx, float gy, int device,
                 const SlabSpec *slab) {
    ctx->prm = *prm;
    if (make_consts(*prm, ctx->c) != SPH_OK) return fail(ctx, SPH_E_ARG, "invalid parameters (skin must be within [0, 1]) or grid too large");
    if (prm->deterministic != 0 && prm->deterministic != 1) return fail(ctx, SPH_E_ARG, "sph_params.deterministic must be 0 or 1");
    ctx->deterministic = prm->deterministic == 1;
    if (getenv("SPH_SLAB_VERIFY_MOST")) ctx->slab_verify_most = atoi(getenv("SPH_SLAB_VERIFY_MOST"));
    if (getenv("SPH_NO_LIST_REPAIR")) ctx->repair_mode = 0;      // (A/B measurements: tools/ab_env5.sh)
    else if (getenv("SPH_LIST_REPAIR")) ctx->repair_mode = 1;
    ctx->skin = ctx->c.cell - 2 * prm->h;
    if (slab) {
        Consts &c = ctx->c;
        const int G = 2;
        if (slab->col_begin < 0 || slab->col_end > c.cols || slab->col_end - slab->col_begin < 4)
            return fail(ctx, SPH_E_ARG, "slab must own at least 4 cell columns inside the grid");
        c.col_off = slab->col_begin - G;
        c.ghost = G;
        c.owned = slab->col_end - slab->col_begin;
        c.cols = c.owned + 2 * G;
        if ((double)c.rows * c.cols > 1.0e9) return fail(ctx, SPH_E_ARG, "local grid too large");
        c.n_cells = c.rows * c.cols;
        c.has_left = slab->has_left;
        c.has_right = slab->has_right;
        c.halo_cap = slab->halo_cap;
        ctx->slab = true;
    }
    for (int i = 0; i < n_fluid; i++)
        if (std::fabs(fluid[i].m - ctx->c.m_fluid) > 1e-6f * ctx->c.m_fluid)
            return fail(ctx, SPH_E_ARG, "fluid mass must be uniform and equal rho0*vol (pi_sph_fluid.c:502)");
    int rc = select_device(ctx, device);
    if (rc) return rc;
    ctx->n = n_fluid;
    ctx->nb = n_boundary;
    ctx->cap = slab ? slab->particle_cap : n_fluid;
    if (ctx->cap < n_fluid) return fail(ctx, SPH_E_ARG, "particle capacity smaller than the initial particle count");

    HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    ctx->own_stream = true;
    for (auto &e : ctx->ev) HIPCHK(ctx, hipEventCreate(&e));

    Arrays &a = ctx->a;
    const size_t n = (size_t)ctx->cap, nb = (size_t)n_boundary, pad = padded_items(ctx->c);
    const size_t tiles = pad / SCAN_TILE;
#define ALLOC(ptr, cnt) if ((rc = dalloc(ctx, &(ptr), (cnt))) != SPH_OK) return rc
    ALLOC(a.pos, n); ALLOC(a.vel, n); ALLOC(a.pos2, n); ALLOC(a.vel2, n); ALLOC(a.id, n); ALLOC(a.rp, n); ALLOC(a.prs, n); ALLOC(a.acc, n);
    ALLOC(a.pk, n); ALLOC(a.velt, n); ALLOC(a.velk, n); ALLOC(a.skey, n); ALLOC(a.pos_ref, n);
    // (tile counts are padded to a multiple of 8 XCD_CHUNKS by the launchers: padded_grid, sph_list.inc)
    const size_t ntiles = ((n + SPH_TILE_PARTICLES - 1) / SPH_TILE_PARTICLES + 8 * XCD_CHUNKS - 1) / (8 * XCD_CHUNKS) * (8 * XCD_CHUNKS) + 9;
    ALLOC(a.tiles, TILE_WORDS * ntiles); ALLOC(a.nlist, (size_t)LIST_WORDS_PER_TILE * ntiles);
    ALLOC(a.lrec, ntiles * SPH_TILE_PARTICLES); ALLOC(a.stab, (size_t)STAB_ENTRIES_PER_TILE * ntiles);
    ALLOC(a.xpair, ntiles * SPH_TILE_PARTICLES);      // (written by the list build next to lrec; read by the list repair: single-GPU contexts)
    ALLOC(a.xranges, (size_t)XRANGE_WORDS * ntiles);
    ALLOC(a.tstart, 4 * (ntiles + 1)); ALLOC(a.pext, (size_t)ctx->c.cols + 4);
    const size_t nwaves = ntiles * (SPH_TILE_PARTICLES / BOXG);      // box groups
    ALLOC(a.wbox, nwaves); ALLOC(a.wnbr, (size_t)WNBR_WORDS * nwaves);
    a.vq = nullptr;
    ALLOC(a.vq, 2 + 2 * (size_t)vq_capacity(ctx->cap));      // (slab contexts too: the verification blocks of their head kernel, k_slab_head)
    a.rq = nullptr;
    ALLOC(a.rq, RQ_CAP);      // (slab contexts: the speculative lean step queues repaired tiles like sph_step)
    a.djobs[0] = a.djobs[1] = nullptr;
    a.pos_first = a.pos;
    ALLOC(a.djobs[0], 1); ALLOC(a.djobs[1], 1);
    ALLOC(a.slot, n > nb ? n : nb);
    ALLOC(a.count, pad); ALLOC(a.cell_start, pad); ALLOC(a.block_sums, tiles * SCAN_SPREAD); ALLOC(a.bcell_start, pad); ALLOC(a.bnear, pad);
    ALLOC(a.dirty, tiles);
    ALLOC(a.bpos, nb); ALLOC(a.bvel, nb); ALLOC(a.bpsi, nb); ALLOC(a.bid, nb);
    ALLOC(a.grav, 1); ALLOC(a.flags, FLAG_WORDS); ALLOC(a.dn, 4); ALLOC(a.dyn, DYN_COUNT);
    ALLOC(a.gbar, (size_t)GBAR_WORDS * GBAR_STRIDE);
    ALLOC(a.wcast, (size_t)WCAST_COPIES * GBAR_STRIDE);
    ALLOC(ctx->d_aos, n); ALLOC(ctx->d_baos, nb); ALLOC(ctx->d_du, n); ALLOC(ctx->d_dv, n); ALLOC(ctx->d_bits, 1024);
    ALLOC(ctx->d_ids, n);
    float2 *&bpos_in = ctx->d_bpos_in, *&bvel_in = ctx->d_bvel_in;
    uint32_t *&bkey = ctx->d_bkey;
    ALLOC(bpos_in, nb); ALLOC(bvel_in, nb); ALLOC(bkey, nb); ALLOC(ctx->d_bpsi0, nb); ALLOC(ctx->d_bcell_ids, nb);
    a.send[0] = a.send[1] = a.recv[0] = a.recv[1] = nullptr;
    if (slab) {
        ALLOC(ctx->d_gseq, 2 * MULTI_STEPS);
        ALLOC(ctx->d_fuse, 1);
        ALLOC(ctx->d_gring, 2 * GRAV_RING);
        HIPCHK(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->h_gseq), sizeof(float) * 2 * MULTI_STEPS * GSEQ_SLOTS, hipHostMallocDefault));
        for (auto &e : ctx->gseq_ev) HIPCHK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->halo_bytes = sizeof(uint32_t) * (HALO_HDR + (size_t)HALO_REC * slab->halo_cap);
        for (int k = 0; k < 2; k++) { ALLOC(a.send[k], ctx->halo_bytes / 4); ALLOC(a.recv[k], ctx->halo_bytes / 4); }
        ctx->own_halo = true;
    }
#undef ALLOC
    if (ctx->device < MAX_DEVICES) {
        g_live_contexts[ctx->device].fetch_add(1);
        ctx->counted = true;
    }
    // one launch for the rebuild chain: single-GPU contexts by default; a slab context only when its host asks for it
    // (sph_set_rebuild_launches): several slabs may share a device, in one process or in several
    ctx->rebuild_wgs = slab ? 0 : rebuild_grid(ctx->device, ctx->cap);
    if (const char *e = getenv("SPH_REBUILD_WGS_MOST")) {      // (measurements: tools/ab_env_small.sh)
        const int most = atoi(e) - atoi(e) % 8;
        if (most >= 8 && ctx->rebuild_wgs > most) ctx->rebuild_wgs = most;
    }
    hipStream_t st = ctx->stream;
    HIPCHK(ctx, hipMemsetAsync(a.count, 0, pad * sizeof(uint32_t), st));
    HIPCHK(ctx, hipMemsetAsync(a.dirty, 0, tiles * sizeof(uint32_t), st));
    HIPCHK(ctx, hipMemsetAsync(a.block_sums, 0, tiles * SCAN_SPREAD * sizeof(uint32_t), st));
    HIPCHK(ctx, hipMemsetAsync(a.flags, 0, FLAG_WORDS * sizeof(uint32_t), st));
    HIPCHK(ctx, hipMemsetAsync(a.gbar, 0, sizeof(uint32_t) * (size_t)GBAR_WORDS * GBAR_STRIDE, st));
    HIPCHK(ctx, hipMemsetAsync(a.wcast, 0, sizeof(uint32_t) * (size_t)WCAST_COPIES * GBAR_STRIDE, st));
    a.rebuild = a.flags + FLAG_REBUILD;
    a.check = a.flags + FLAG_CHECK;
    a.latch = a.flags + FLAG_LATCH;
    ctx->pos_a = a.pos;
    HIPCHK(ctx, hipMemsetAsync(a.acc, 0, (n ? n : 1) * sizeof(float2), st));
    if (slab)
        for (int k = 0; k < 2; k++) {
            HIPCHK(ctx, hipMemsetAsync(a.send[k], 0, ctx->halo_bytes, st));
            HIPCHK(ctx, hipMemsetAsync(a.recv[k], 0, ctx->halo_bytes, st));
        }
    const uint32_t hdn[4] = {(uint32_t)n_fluid, (uint32_t)n_fluid, 0u, 0u};
    HIPCHK(ctx, hipMemcpyAsync(a.dn, hdn, sizeof hdn, hipMemcpyHostToDevice, st));
    {   // the skin the first lists are built with (adapt_skin derives the thresholds from it at the first rebuild): the
        // smallest — most scenes start at rest, and the first rebuild the flow itself asks for corrects it
        const Consts &c = ctx->c;
        const float hdyn[DYN_COUNT] = {0.0f, 0.0f, 0.0f, c.skin_min, 0.0f, 0.0f};
        a.uref = ctx->slab ? nullptr : a.dyn + DYN_UREF_X;      // (slabs: the absolute criterion — their references would differ)
        HIPCHK(ctx, hipMemcpyAsync(a.dyn, hdyn, sizeof hdyn, hipMemcpyHostToDevice, st));
        if (a.vq) {
            const uint32_t head[2] = {0u, (uint32_t)vq_capacity(ctx->cap)};
            HIPCHK(ctx, hipMemsetAsync(a.vq, 0, sizeof(uint32_t) * (2 + 2 * (size_t)vq_capacity(ctx->cap)), st));
            HIPCHK(ctx, hipMemcpyAsync(a.vq, head, sizeof head, hipMemcpyHostToDevice, st));
            HIPCHK(ctx, hipStreamSynchronize(st));      // (a local)
        }
        if ((rc = upload_jobs(ctx)) != SPH_OK) return rc;
    }

    // boundary: bin once, pseudo-mass once (:600-601)
    std::vector<float2> hb(nb ? nb : 1), hbv(nb ? nb : 1);
    std::vector<float> hpsi(nb ? nb : 1);
    for (size_t i = 0; i < nb; i++) {
        hb[i] = make_float2(boundary[i].x, boundary[i].y);
        hbv[i] = make_float2(boundary[i].u, boundary[i].v);      // read by the viscosity term (:357); walls do not move
        hpsi[i] = boundary[i].m;
    }
    HIPCHK(ctx, hipMemcpyAsync(bpos_in, hb.data(), nb * sizeof(float2), hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(bvel_in, hbv.data(), nb * sizeof(float2), hipMemcpyHostToDevice, st));
    launch_set_rebuild(st, a, true);        // the scan is a rebuild kernel
    launch_boundary_key(st, ctx->c, bpos_in, bkey, a.slot, a.count, a.dirty, a.flags, n_boundary);
    launch_scan(st, ctx->c, a.count, a.dirty, a.bcell_start, a.block_sums, a.rebuild, true);
    launch_boundary_reorder(st, bpos_in, bkey, a.slot, a.bcell_start, a.bpos, a.bid, n_boundary, bvel_in, a.bvel, ctx->d_bcell_ids);
    launch_boundary_near(st, ctx->c, a);
    if (psi_given) {
        // psi was computed on the full wall set (a slab sees only its part of the walls): scatter it to bin order
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_bpsi0, hpsi.data(), nb * sizeof(float), hipMemcpyHostToDevice, st));
        launch_boundary_gather_psi(st, a, ctx->d_bpsi0, n_boundary);
    } else {
        launch_boundary_psi(st, ctx->c, a, n_boundary);
        launch_boundary_unsort_psi(st, a, ctx->d_bpsi0, n_boundary);      // original order, for sph_update_boundary
    }

    // fluid: upload, bin, then rho, p, a at t = 0 (:604-607)
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_aos, fluid, (size_t)n_fluid * sizeof(sph_particle), hipMemcpyHostToDevice, st));
    launch_upload_state(st, a, n_fluid, ctx->d_aos);
    if (ids) HIPCHK(ctx, hipMemcpyAsync(a.id, ids, (size_t)n_fluid * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    if ((rc = resort_state(ctx)) != SPH_OK) return rc;
    launch_set_gravity(st, a, gx, gy);
    launch_density(st, ctx->c, a, ctx->cap, DENS_RHO_EOS, ctx->variant, false);
    launch_force(st, ctx->c, a, ctx->cap, FORCE_EVAL, ctx->variant);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(st));   // also keeps hb / fluid alive until the copies are done
    rc = check_flags(ctx);
    if (rc) return rc;
    rc = selftest_one_launch(ctx);
    if (rc == SPH_OK) {
        prebuild_graphs(ctx);
        HIPCHK(ctx, hipStreamSynchronize(st));
    }
    return rc;
}

}  // namespace

extern "C" {

int sph_create(sph_ctx **out, const sph_params *prm, const sph_particle *fluid, int n_fluid,
               const sph_particle *boundary, int n_boundary, float gx, float gy, int device) {
    if (!out) return SPH_E_ARG;
    *out = nullptr;
    sph_ctx *ctx = new sph_ctx();
    *out = ctx;   // returned even on failure so that sph_last_error() can be read; caller destroys it
    if (!prm || n_fluid < 0 || n_boundary < 0 || (n_fluid > 0 && !fluid) || (n_boundary > 0 && !boundary))
        return fail(ctx, SPH_E_ARG, "sph_create: null pointer or negative count");
    return init_context(ctx, prm, fluid, nullptr, n_fluid, boundary, n_boundary, false, gx, gy, device, nullptr);
}

int sph_create_slab(sph_ctx **out, const sph_params *prm, const sph_slab_desc *desc, const sph_particle *fluid,
                    const uint32_t *ids, int n_fluid, const sph_particle *boundary_all, int n_boundary_all, float gx,
                    float gy, int device) {
    if (!out) return SPH_E_ARG;
    *out = nullptr;
    sph_ctx *ctx = new sph_ctx();
    *out = ctx;
    if (!prm || !desc || n_fluid < 0 || n_boundary_all < 0 || (n_fluid > 0 && (!fluid || !ids)) ||
        (n_boundary_all > 0 && !boundary_all))
        return fail(ctx, SPH_E_ARG, "sph_create_slab: null pointer or negative count");
    // Akinci psi needs every wall neighbour of a wall particle; a slab holds only its part of the walls, so psi
    // is evaluated once on the full wall set (a throw-away single-mode context without fluid), then filtered.
    std::vector<sph_particle> ball(n_boundary_all ? n_boundary_all : 1);
    {
        sph_ctx *tmp = nullptr;
        int rc = sph_create(&tmp, prm, nullptr, 0, boundary_all, n_boundary_all, gx, gy, device);
        if (rc == SPH_OK) rc = sph_read_boundary(tmp, ball.data());
        if (rc != SPH_OK) {
            ctx->err = std::string("sph_create_slab: wall pseudo-mass pass failed: ") + (tmp ? tmp->err : "");
            sph_destroy(tmp);
            return rc;
        }
        sph_destroy(tmp);
    }
    Consts cg;
    if (make_consts(*prm, cg) != SPH_OK) return fail(ctx, SPH_E_ARG, "sph_create_slab: invalid parameters");
    const int lo = desc->col_begin - 2, hi = desc->col_end + 2;   // local columns [lo, hi)
    std::vector<sph_particle> bloc;
    for (int i = 0; i < n_boundary_all; i++) {
        const float fc = (ball[i].x - cg.x_min) * cg.inv_cell;     // same arithmetic as the device's cell_of
        const int col = (int)fc;
        if (fc >= 0.0f && col >= lo && col < hi) bloc.push_back(ball[i]);
    }
    SlabSpec sp;
    sp.col_begin = desc->col_begin;
    sp.col_end = desc->col_end;
    sp.has_left = desc->has_left != 0;
    sp.has_right = desc->has_right != 0;
    sp.halo_cap = desc->halo_capacity > 0 ? desc->halo_capacity : 4 * cg.rows * 16;      // (sph_slab_halo_bytes says the same)
    sp.particle_cap = desc->particle_capacity > 0 ? desc->particle_capacity : n_fluid + n_fluid / 4 + 2 * sp.halo_cap + 1024;
    return init_context(ctx, prm, fluid, ids, n_fluid, bloc.data(), (int)bloc.size(), true, gx, gy, device, &sp);
}

int sph_step(sph_ctx *ctx, float gx, float gy, int nsteps) {
    if (!ctx || nsteps < 0) return SPH_E_ARG;
    if (!ctx->stream) return fail(ctx, SPH_E_STATE, "context not initialised");
    if (ctx->slab) return fail(ctx, SPH_E_STATE, "slab context: use sph_slab_step_begin / exchange / sph_slab_step_end");
    (void)hipSetDevice(ctx->device);
    if (ctx->rebuild_wgs > 0 && device_shared(ctx)) {      // another context of this process on the device: see g_live_contexts
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        drop_graph(ctx);
        ctx->rebuild_wgs = 0;
    }
    // No step, no new gravity: the vector on the device belongs to the last force pass until the next one runs — the
    // acceleration of the last step is recomputed from it on demand (refresh_acc), so a host that polls its gravity source
    // with sph_step(ctx, g_new, 0) and then reads back must still get a(g_old).  (With nsteps > 0 nothing stale can be read
    // afterwards: whenever the look-ahead is dropped — uploads, variant change — the acceleration was refreshed first.)
    if (nsteps == 0) return SPH_OK;
    launch_set_gravity(ctx->stream, ctx->a, gx, gy);
    int s = 0;
    while (s < nsteps) {
        if (fused(ctx) && ctx->primed && nsteps - s >= 2) {
            int m = MULTI_STEPS;
            while (m > nsteps - s) m >>= 1;
            hipGraphExec_t g = multi_graph(ctx, m);
            if (g) {
                HIPCHK(ctx, hipGraphLaunch(g, ctx->stream));
                ctx->velt_stale = true;
                ctx->acc_stale = true;
                ctx->p_stale = true;
                s += m;
                continue;
            }
        }
        int rc = run_step(ctx, nullptr);
        if (rc) return rc;
        s++;
    }
    HIPCHK(ctx, hipGetLastError());
    return SPH_OK;
}

int sph_sync(sph_ctx *ctx) {
    if (!ctx || !ctx->stream) return SPH_E_ARG;
    (void)hipSetDevice(ctx->device);
    return check_flags(ctx);
}

int sph_read_particles(sph_ctx *ctx, sph_particle *out) {
    if (!ctx || !ctx->stream || (!out && ctx->n)) return SPH_E_ARG;
    if (ctx->slab) return fail(ctx, SPH_E_STATE, "slab context: use sph_slab_read");
    (void)hipSetDevice(ctx->device);
    refresh_velt(ctx);
    refresh_p(ctx);
    launch_unsort_particles(ctx->stream, ctx->c, ctx->a, ctx->n, ctx->d_aos);
    HIPCHK(ctx, hipMemcpyAsync(out, ctx->d_aos, (size_t)ctx->n * sizeof(sph_particle), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SPH_OK;
}

int sph_read_accel(sph_ctx *ctx, float *du_dt, float *dv_dt) {
    if (!ctx || !ctx->stream || ((!du_dt || !dv_dt) && ctx->n)) return SPH_E_ARG;
    if (ctx->slab) return fail(ctx, SPH_E_STATE, "slab context: use sph_slab_read");
    (void)hipSetDevice(ctx->device);
    refresh_acc(ctx);
    launch_unsort_accel(ctx->stream, ctx->a, ctx->n, ctx->d_du, ctx->d_dv);
    HIPCHK(ctx, hipMemcpyAsync(du_dt, ctx->d_du, (size_t)ctx->n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(dv_dt, ctx->d_dv, (size_t)ctx->n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SPH_OK;
}

int sph_read_boundary(sph_ctx *ctx, sph_particle *out) {
    if (!ctx || !ctx->stream || (!out && ctx->nb)) return SPH_E_ARG;
    (void)hipSetDevice(ctx->device);
    launch_unsort_boundary(ctx->stream, ctx->c, ctx->a, ctx->nb, ctx->d_baos);
    HIPCHK(ctx, hipMemcpyAsync(out, ctx->d_baos, (size_t)ctx->nb * sizeof(sph_particle), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SPH_OK;
}

int sph_update_boundary(sph_ctx *ctx, const sph_particle *boundary) {
    if (!ctx || !ctx->stream || (!boundary && ctx->nb)) return SPH_E_ARG;
    if (ctx->slab) return fail(ctx, SPH_E_STATE, "sph_update_boundary is single-GPU only");
    (void)hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    Arrays &a = ctx->a;
    refresh_acc(ctx);      // (a of the last step belongs to the walls as they were)
    const size_t nb = (size_t)ctx->nb;
    std::vector<float2> hb(nb ? nb : 1), hbv(nb ? nb : 1);
    for (size_t i = 0; i < nb; i++) {
        hb[i] = make_float2(boundary[i].x, boundary[i].y);
        hbv[i] = make_float2(boundary[i].u, boundary[i].v);
    }
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_bpos_in, hb.data(), nb * sizeof(float2), hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_bvel_in, hbv.data(), nb * sizeof(float2), hipMemcpyHostToDevice, st));
    // re-bin the walls (the histogram array is zero between sorts; the scan is gated by the rebuild word), pseudo-mass
    // follows its particle (rigid motion leaves the wall's own neighbourhood, hence psi :259, unchanged)
    launch_set_rebuild(st, a, true);
    launch_boundary_key(st, ctx->c, ctx->d_bpos_in, ctx->d_bkey, a.slot, a.count, a.dirty, a.flags, ctx->nb);
    launch_scan(st, ctx->c, a.count, a.dirty, a.bcell_start, a.block_sums, a.rebuild, true);
    launch_boundary_reorder(st, ctx->d_bpos_in, ctx->d_bkey, a.slot, a.bcell_start, a.bpos, a.bid, ctx->nb, ctx->d_bvel_in, a.bvel, ctx->d_bcell_ids);
    launch_boundary_gather_psi(st, a, ctx->d_bpsi0, ctx->nb);
    launch_boundary_near(st, ctx->c, a);
    // the rebuild word stays raised: the tile records count the wall particles in reach of each tile, so the next step
    // rebuilds the fluid's neighbour structure against the new wall bins (this also keeps a request the last force
    // pass may have left for that step)
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(st));      // hb / hbv are free again
    return check_flags(ctx);
}

int sph_set_boundary_velocity(sph_ctx *ctx, float u, float v) {
    if (!ctx || !ctx->stream) return SPH_E_ARG;
    if (!(std::isfinite(u) && std::isfinite(v))) return fail(ctx, SPH_E_ARG, "sph_set_boundary_velocity: velocity not finite");
    if (ctx->slab && ctx->slab_phase != 0) return fail(ctx, SPH_E_STATE, "sph_set_boundary_velocity mid-step");
    (void)hipSetDevice(ctx->device);
    refresh_acc(ctx);      // (a of the last step belongs to the wall velocity as it was)
    launch_fill_float2(ctx->stream, ctx->a.bvel, u, v, ctx->nb);            // bin order: what the force pass reads
    launch_fill_float2(ctx->stream, ctx->d_bvel_in, u, v, ctx->nb);         // original order: what sph_update_boundary re-bins
    HIPCHK(ctx, hipGetLastError());
    return SPH_OK;
}

int sph_stats(sph_ctx *ctx, float *max_rho, float *max_speed) {
    if (!ctx || !ctx->stream) return SPH_E_ARG;
    (void)hipSetDevice(ctx->device);
    HIPCHK(ctx, hipMemsetAsync(ctx->a.flags + FLAG_MAXRHO, 0, 2 * sizeof(uint32_t), ctx->stream));
    refresh_velt(ctx);
    launch_stats(ctx->stream, ctx->c, ctx->a, ctx->slab ? ctx->cap : ctx->n, ctx->slab);      // (a slab: its owned particles)
    uint32_t h[2] = {0, 0};
    HIPCHK(ctx, hipMemcpyAsync(h, ctx->a.flags + FLAG_MAXRHO, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    float r, s;
    memcpy(&r, &h[0], 4);
    memcpy(&s, &h[1], 4);
    if (max_rho) *max_rho = r;
    if (max_speed) *max_speed = s;
    return SPH_OK;
}

int sph_n_fluid(const sph_ctx *ctx) { return ctx ? ctx->n : SPH_E_ARG; }
int sph_n_boundary(const sph_ctx *ctx) { return ctx ? ctx->nb : SPH_E_ARG; }
int sph_grid_dims(const sph_ctx *ctx, int *n_cells, int *m_cells) {
    if (!ctx) return SPH_E_ARG;
    const sph_params &p = ctx->prm;
    const float cell = 2 * p.h;                                                        // :596
    if (n_cells) *n_cells = (int)((p.y_max - p.y_min) / cell) + 1;                     // :93
    if (m_cells) *m_cells = (int)((p.x_max - p.x_min) / cell) + 1;                     // :94
    return SPH_OK;
}
int sph_device_grid(const sph_ctx *ctx, int *rows, int *cols, float *cell) {
    if (!ctx) return SPH_E_ARG;
    if (rows) *rows = ctx->c.rows;
    if (cols) *cols = ctx->c.cols;
    if (cell) *cell = ctx->c.cell;
    return SPH_OK;
}
float sph_device_cell(const sph_params *prm) {
    if (!prm || !(prm->skin >= 0.0f && prm->skin <= 1.0f)) return 0.0f;
    const float two_h = 2 * prm->h;
    return two_h + prm->skin * two_h;      // the arithmetic of make_consts
}
float sph_current_skin(sph_ctx *ctx) {
    if (!ctx || !ctx->stream) return -1.0f;
    (void)hipSetDevice(ctx->device);
    float s = 0.0f;
    if (hipMemcpyAsync(&s, ctx->a.dyn + DYN_SKIN, sizeof s, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess)
        return -1.0f;
    return s / (2 * ctx->prm.h);
}
int sph_request_rebuild(sph_ctx *ctx) {
    if (!ctx || !ctx->stream) return SPH_E_ARG;
    if (ctx->slab) return fail(ctx, SPH_E_STATE, "slab context: raise the word with sph_slab_flag_set on every rank");
    (void)hipSetDevice(ctx->device);
    launch_request_rebuild(ctx->stream, ctx->a);
    HIPCHK(ctx, hipGetLastError());
    return SPH_OK;
}
int sph_rebuild_stats(sph_ctx *ctx, long long *rebuilds, long long *direct_tiles) {
    if (!ctx || !ctx->stream) return SPH_E_ARG;
    (void)hipSetDevice(ctx->device);
    uint32_t h[FLAG_COUNT] = {0};
    HIPCHK(ctx, hipMemcpyAsync(h, ctx->a.flags, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (rebuilds) *rebuilds = h[FLAG_NREBUILD];
    if (direct_tiles) *direct_tiles = h[FLAG_DIRECT_TILES];
    return SPH_OK;
}
int sph_direct_tile_reasons(sph_ctx *ctx, long long why[7]) {
    if (!ctx || !ctx->stream || !why) return SPH_E_ARG;
    (void)hipSetDevice(ctx->device);
    uint32_t h[7] = {0};
    static_assert(FLAG_OFF_XCD == FLAG_WHY_DIRECT + 6, "the diagnostics are read as one block");
    HIPCHK(ctx, hipMemcpyAsync(h, ctx->a.flags + FLAG_WHY_DIRECT, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < 7; k++) why[k] = h[k];
    return SPH_OK;
}
int sph_set_verification(sph_ctx *ctx, int mode) {
    if (!ctx || !ctx->stream || mode < -1 || mode > 1) return SPH_E_ARG;
    (void)hipSetDevice(ctx->device);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->verify_mode = mode;
    drop_graph(ctx);      // (an argument of the captured density launches)
    return SPH_OK;
}

int sph_set_list_repair(sph_ctx *ctx, int mode) {
    if (!ctx || !ctx->stream || mode < -1 || mode > 1) return SPH_E_ARG;
    (void)hipSetDevice(ctx->device);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const bool before = list_repair(ctx);
    ctx->repair_mode = mode;
    drop_graph(ctx);      // (an argument of the captured gate launch)
    // lists built while repairs were off have neither the spare row nor the remembered partners: the next step rebuilds them
    if (list_repair(ctx) && !before) launch_request_rebuild(ctx->stream, ctx->a);
    return upload_jobs(ctx);
}

int sph_rebuild_reasons(sph_ctx *ctx, long long why[4]) {
    if (!ctx || !ctx->stream || !why) return SPH_E_ARG;
    (void)hipSetDevice(ctx->device);
    uint32_t h[4] = {0, 0, 0, 0};
    HIPCHK(ctx, hipMemcpyAsync(h, ctx->a.flags + FLAG_WHY_REBUILD, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < 4; k++) why[k] = h[k];
    return SPH_OK;
}

int sph_repair_stats(sph_ctx *ctx, long long out[4]) {
    if (!ctx || !ctx->stream || !out) return SPH_E_ARG;
    (void)hipSetDevice(ctx->device);
    uint32_t h[4] = {0, 0, 0, 0};
    static_assert(FLAG_REPAIR_FAIL == FLAG_REPAIRS + 1, "read as one block");
    HIPCHK(ctx, hipMemcpyAsync(h, ctx->a.flags + FLAG_REPAIRS, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < 4; k++) out[k] = h[k];
    return SPH_OK;
}

int sph_verify_stats(sph_ctx *ctx, long long *pairs) {
    if (!ctx || !ctx->stream || !pairs) return SPH_E_ARG;
    (void)hipSetDevice(ctx->device);
    uint32_t h = 0;
    HIPCHK(ctx, hipMemcpyAsync(&h, ctx->a.flags + FLAG_NVERIFY, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const long long sum = h;
    *pairs = sum;
    return SPH_OK;
}

int sph_check_stats(sph_ctx *ctx, long long *checks) {
    if (!ctx || !ctx->stream || !checks) return SPH_E_ARG;
    (void)hipSetDevice(ctx->device);
    uint32_t h = 0;
    HIPCHK(ctx, hipMemcpyAsync(&h, ctx->a.flags + FLAG_NCHECK, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *checks = h;
    return SPH_OK;
}
long long sph_out_of_domain_count(sph_ctx *ctx) {
    if (!ctx) return SPH_E_ARG;
    if (ctx->stream) (void)check_flags(ctx);
    return ctx->oob_total;
}
size_t sph_device_bytes(const sph_ctx *ctx) { return ctx ? ctx->bytes : 0; }

int sph_set_rebuild_launches(sph_ctx *ctx, int one_launch) {
    if (!ctx || !ctx->stream) return SPH_E_ARG;
    (void)hipSetDevice(ctx->device);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    drop_graph(ctx);
    ctx->one_launch_asked = one_launch != 0;
    ctx->rebuild_wgs = one_launch ? rebuild_grid(ctx->device, ctx->cap) : 0;
    if (one_launch && ctx->rebuild_wgs <= 0) return fail(ctx, SPH_E_HIP, "occupancy query for the one-launch rebuild failed");
    if (one_launch > 1) {      // a cap on the grid: several contexts whose hosts vouch that ALL their one-launch grids fit the device together
        const int capg = one_launch - one_launch % 8;
        if (capg < 8) return fail(ctx, SPH_E_ARG, "sph_set_rebuild_launches: a grid cap below 8 workgroups");
        if (ctx->rebuild_wgs > capg) ctx->rebuild_wgs = capg;
    }
    if (one_launch) {
        int rc = selftest_one_launch(ctx);
        if (rc) return rc;
        if (ctx->rebuild_wgs <= 0) return fail(ctx, SPH_E_STATE, "the one-launch rebuild's workgroups are not all resident on this device");
    }
    return SPH_OK;
}

int sph_get_rebuild_launches(const sph_ctx *ctx) { return ctx ? (ctx->rebuild_wgs > 0 ? 1 : 0) : SPH_E_ARG; }

int sph_set_variant(sph_ctx *ctx, int variant) {
    if (!ctx || variant < 0 || variant > 1) return SPH_E_ARG;
    if (variant != ctx->variant) {
        if (ctx->stream) {
            (void)hipSetDevice(ctx->device);
            refresh_velt(ctx);
            (void)hipStreamSynchronize(ctx->stream);
        }
        drop_graph(ctx);
        ctx->variant = variant;      // both variants work from the same sorted state
        ctx->primed = false;         // a look-ahead kick/drift of the list kernels is dropped, the current state kept
    }
    return SPH_OK;
}

int sph_set_stream(sph_ctx *ctx, void *hip_stream) {
    if (!ctx || !ctx->stream) return SPH_E_ARG;
    (void)hipSetDevice(ctx->device);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    drop_graph(ctx);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    if (hip_stream) {
        ctx->stream = static_cast<hipStream_t>(hip_stream);
        ctx->own_stream = false;
    } else {
        HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        ctx->own_stream = true;
    }
    return SPH_OK;
}

// ---- stage entry points ----
int sph_upload_state(sph_ctx *ctx, const sph_particle *fluid) {
    if (!ctx || !ctx->stream || (!fluid && ctx->n)) return SPH_E_ARG;
    if (ctx->slab) return fail(ctx, SPH_E_STATE, "stage entry points are single-GPU only");
    (void)hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    // du_dt, dv_dt stay with their particles (reference: index-aligned arrays, :616): out to original order before
    // the arrays are rewritten, back in through the new sort order afterwards
    refresh_acc(ctx);
    launch_unsort_accel(st, ctx->a, ctx->n, ctx->d_du, ctx->d_dv);
    ctx->velt_stale = false;      // velt is rewritten below
    ctx->p_stale = false;         // ... and so are rho and p
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_aos, fluid, (size_t)ctx->n * sizeof(sph_particle), hipMemcpyHostToDevice, st));
    launch_upload_state(st, ctx->a, ctx->n, ctx->d_aos);
    ctx->primed = false;
    ctx->stepped = false;
    int rc = resort_state(ctx);
    if (rc) return rc;
    launch_gather_rho_p(st, ctx->c, ctx->a, ctx->n, ctx->d_aos);
    launch_gather_accel(st, ctx->a, ctx->n, ctx->d_du, ctx->d_dv);
    HIPCHK(ctx, hipGetLastError());
    return check_flags(ctx);
}

int sph_upload_accel(sph_ctx *ctx, const float *du_dt, const float *dv_dt) {
    if (!ctx || !ctx->stream || ((!du_dt || !dv_dt) && ctx->n)) return SPH_E_ARG;
    if (ctx->slab) return fail(ctx, SPH_E_STATE, "stage entry points are single-GPU only");
    (void)hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    refresh_velt(ctx);            // with the accelerations of the last step, before they are replaced
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_du, du_dt, (size_t)ctx->n * sizeof(float), hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_dv, dv_dt, (size_t)ctx->n * sizeof(float), hipMemcpyHostToDevice, st));
    launch_gather_accel(st, ctx->a, ctx->n, ctx->d_du, ctx->d_dv);
    ctx->primed = false;         // a look-ahead kick/drift made with the old du_dt is dropped
    ctx->stepped = false;
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(st));      // the caller's buffers are free again
    return SPH_OK;
}

int sph_eval_density(sph_ctx *ctx) {
    if (!ctx || !ctx->stream) return SPH_E_ARG;
    (void)hipSetDevice(ctx->device);
    refresh_acc(ctx);             // a of the last step, while the rho and p it was computed from are still there
    refresh_p(ctx);               // the stored p, while the rho it belongs to is still there
    launch_density(ctx->stream, ctx->c, ctx->a, ctx->cap, DENS_RHO, ctx->variant, false);
    launch_eos(ctx->stream, ctx->c, ctx->a, ctx->cap, true);   // keep p/rho^2 consistent with the new rho and the stored p
    HIPCHK(ctx, hipGetLastError());
    return SPH_OK;
}

int sph_eval_pressure(sph_ctx *ctx) {
    if (!ctx || !ctx->stream) return SPH_E_ARG;
    (void)hipSetDevice(ctx->device);
    refresh_acc(ctx);
    launch_eos(ctx->stream, ctx->c, ctx->a, ctx->cap, false);
    ctx->p_stale = false;
    HIPCHK(ctx, hipGetLastError());
    return SPH_OK;
}

int sph_eval_accel(sph_ctx *ctx, float gx, float gy) {
    if (!ctx || !ctx->stream) return SPH_E_ARG;
    (void)hipSetDevice(ctx->device);
    refresh_velt(ctx);            // with the accelerations of the last step, before they are replaced
    launch_set_gravity(ctx->stream, ctx->a, gx, gy);
    launch_force(ctx->stream, ctx->c, ctx->a, ctx->cap, FORCE_EVAL, ctx->variant);
    ctx->primed = false;         // the next step kicks with THIS du_dt (:616), not with a look-ahead made before it
    ctx->stepped = false;
    HIPCHK(ctx, hipGetLastError());
    return SPH_OK;
}

// ---- measurement ----
int sph_profile_steps(sph_ctx *ctx, float gx, float gy, int nsteps, sph_kernel_times *out) {
    if (!ctx || !ctx->stream || !out || nsteps <= 0) return SPH_E_ARG;
    if (ctx->slab) return fail(ctx, SPH_E_STATE, "sph_profile_steps is single-GPU only");
    (void)hipSetDevice(ctx->device);
    memset(out, 0, sizeof *out);
    long long r0 = 0, r1 = 0;
    int rc = sph_rebuild_stats(ctx, &r0, nullptr);
    if (rc) return rc;
    launch_set_gravity(ctx->stream, ctx->a, gx, gy);
    double acc[SPH_K_COUNT] = {0}, total = 0;
    for (int s = 0; s < nsteps; s++) {
        int rs = run_step(ctx, ctx->ev);
        if (rs) return rs;
        HIPCHK(ctx, hipEventSynchronize(ctx->ev[SPH_K_HALO]));
        for (int k = 0; k < SPH_K_HALO; k++) {
            float ms = 0;
            HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev[k], ctx->ev[k + 1]));
            acc[k] += ms;
        }
        float ms = 0;
        HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[SPH_K_HALO]));
        total += ms;
    }
    for (int k = 0; k < SPH_K_HALO; k++) out->ms[k] = (float)(acc[k] / nsteps);
    out->step_ms = (float)(total / nsteps);
    out->nsteps = nsteps;
    rc = sph_rebuild_stats(ctx, &r1, nullptr);
    if (rc) return rc;
    out->rebuilds = (int)(r1 - r0);
    HIPCHK(ctx, hipGetLastError());
    return SPH_OK;
}

int sph_time_kernel(sph_ctx *ctx, int kernel, int reps, float *ms) {
    if (!ctx || !ctx->stream || !ms || reps <= 0) return SPH_E_ARG;
    if (kernel != SPH_K_DENSITY_EOS && kernel != SPH_K_FORCE_KICK && kernel != SPH_K_BUILD_LIST && kernel != SPH_K_DENSITY_SPEC)
        return fail(ctx, SPH_E_ARG, "sph_time_kernel: kernel is not idempotent");
    if (kernel == SPH_K_DENSITY_SPEC && (!speculative(ctx) || device_shared(ctx)))
        return fail(ctx, SPH_E_STATE, "sph_time_kernel(SPH_K_DENSITY_SPEC): this context's step does not launch the speculative density pass");
    if (kernel == SPH_K_BUILD_LIST && ctx->slab) return fail(ctx, SPH_E_STATE, "sph_time_kernel(SPH_K_BUILD_LIST): not on a slab context");
    // the force pass of the step writes velt = vel + dt/2 a: a repeat of the last step's kick only once a step has
    // kicked vel; before that vel == velt and the launch would kick the velocities a second time
    if (kernel == SPH_K_FORCE_KICK && !ctx->stepped)
        return fail(ctx, SPH_E_STATE, "sph_time_kernel(SPH_K_FORCE_KICK) needs at least one sph_step since creation / upload");
    (void)hipSetDevice(ctx->device);
    // the list build on the sort that is there, with the positions of now: what it leaves is replaced by the next step,
    // which finds the rebuild word raised
    if (kernel == SPH_K_BUILD_LIST) launch_request_rebuild(ctx->stream, ctx->a);
    // the speculative launch as the step issues it (its criterion jobs read the boxes the last force pass left: whatever they find,
    // the words they raise are put back after every launch, where the step's gate would clear them or rebuild)
    const bool verify = ctx->verify_mode < 0 ? ctx->n >= VERIFY_MIN_PARTICLES : ctx->verify_mode > 0;
    // (the verify jobs of the timed launches must not repair lists — appended entries, FLAG_NREPAIR, xpair would outlive the measurement:
    // for its duration the jobs' record says "no repairs"; a missing pair then only raises the word, which is put back)
    const bool mute_repair = kernel == SPH_K_DENSITY_SPEC && list_repair(ctx);
    if (mute_repair) { int rc = upload_jobs(ctx, true); if (rc) return rc; }
    if (kernel == SPH_K_DENSITY_SPEC) launch_spec_reset(ctx->stream, ctx->a, 0);
    HIPCHK(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    for (int r = 0; r < reps; r++) {
        if (kernel == SPH_K_DENSITY_SPEC) {
            launch_density(ctx->stream, ctx->c, ctx->a, ctx->cap, DENS_RHO_EOS, ctx->variant, false, DENS_ALL, false, true, verify);
            launch_spec_reset(ctx->stream, ctx->a, r + 1 < reps ? 1 : 2);
        } else
        if (kernel == SPH_K_BUILD_LIST) launch_build_list(ctx->stream, ctx->c, ctx->a, ctx->cap);
        else if (kernel == SPH_K_DENSITY_EOS) launch_density(ctx->stream, ctx->c, ctx->a, ctx->cap, DENS_RHO_EOS, ctx->variant, false);
        else launch_force(ctx->stream, ctx->c, ctx->a, ctx->cap, fused(ctx) ? FORCE_KICK_DRIFT : FORCE_KICK, ctx->variant);
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    // (tiles of a speculative launch whose jobs raised the word have left early, or computed on a half-staged tile: in a step the gate
    // rebuilds and repeats the pass — here the plain pass puts rho and p / rho^2 back, outside the timed interval)
    if (kernel == SPH_K_DENSITY_SPEC) launch_density(ctx->stream, ctx->c, ctx->a, ctx->cap, DENS_RHO_EOS, ctx->variant, false);
    HIPCHK(ctx, hipEventSynchronize(ctx->ev[1]));
    float t = 0;
    HIPCHK(ctx, hipEventElapsedTime(&t, ctx->ev[0], ctx->ev[1]));
    *ms = t / reps;
    if (mute_repair) { int rc = upload_jobs(ctx); if (rc) return rc; }
    HIPCHK(ctx, hipGetLastError());
    return SPH_OK;
}

// ---- box calibration (sph_diag.h) ----
namespace {
// (tools/ubench_copy on MI355X: four non-temporal loads in flight per thread and 8192 workgroups reach 5.1-5.2 TB/s, hipMemcpyDtoD 5.2; one
// plain load per thread 4.5-4.7, four plain loads 4.3-4.5)
typedef float cal_v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_cal_copy(const cal_v4f *__restrict__ src, cal_v4f *__restrict__ dst, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const cal_v4f a = __builtin_nontemporal_load(&src[i]), b = __builtin_nontemporal_load(&src[i + stride]),
                      c = __builtin_nontemporal_load(&src[i + 2 * stride]), d = __builtin_nontemporal_load(&src[i + 3 * stride]);
        __builtin_nontemporal_store(a, &dst[i]);
        __builtin_nontemporal_store(b, &dst[i + stride]);
        __builtin_nontemporal_store(c, &dst[i + 2 * stride]);
        __builtin_nontemporal_store(d, &dst[i + 3 * stride]);
    }
    for (; i < n4; i += stride) dst[i] = src[i];
}
constexpr int CAL_ITERS = 4096, CAL_BLOCKS = 2048;
__global__ __launch_bounds__(256) void k_cal_valu(float *out, float seed, unsigned long long *clocks) {
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = seed + i + threadIdx.x * 1e-3f;
    const float c1 = seed * 0.999f, c2 = seed * 1e-3f;
    const bool stamp = blockIdx.x == 0 && threadIdx.x == 0;
    unsigned long long t0 = 0, w0 = 0;
    if (stamp) { t0 = clock64(); w0 = wall_clock64(); }
    for (int it = 0; it < CAL_ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) a[i] = fmaf(a[i], c1, c2);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (stamp) { clocks[0] = clock64() - t0; clocks[1] = wall_clock64() - w0; }
}
}  // namespace

int sph_box_calibrate(int device, sph_box_calibration *out) {
    if (!out) return SPH_E_ARG;
    memset(out, 0, sizeof *out);
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) { (void)hipGetLastError(); return SPH_E_HIP; }
    if (hipSetDevice(device) != hipSuccess) return SPH_E_HIP;
    const size_t bytes = (size_t)1 << 30, n4 = bytes / sizeof(cal_v4f);
    cal_v4f *src = nullptr, *dst = nullptr;
    float *vout = nullptr;
    unsigned long long *clk = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipStream_t st = nullptr;
    int rc = SPH_E_HIP;
    do {
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) break;
        if (hipMalloc(&src, bytes) != hipSuccess || hipMalloc(&dst, bytes) != hipSuccess) { rc = SPH_E_NOMEM; break; }
        if (hipMalloc(&vout, (size_t)CAL_BLOCKS * 256 * sizeof(float)) != hipSuccess || hipMalloc(&clk, 2 * sizeof(unsigned long long)) != hipSuccess) { rc = SPH_E_NOMEM; break; }
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) break;
        if (hipMemsetAsync(src, 0x3c, bytes, st) != hipSuccess || hipMemsetAsync(dst, 0, bytes, st) != hipSuccess) break;
        // copy: two untimed passes, then as many as fill ~50 ms
        const int grid = 8192;
        for (int k = 0; k < 2; k++) hipLaunchKernelGGL(k_cal_copy, dim3(grid), dim3(256), 0, st, src, dst, n4);
        const int reps = 96;      // 96 x 2 GiB at ~4-5 TB/s ~ 45 ms
        (void)hipEventRecord(e0, st);
        for (int k = 0; k < reps; k++) hipLaunchKernelGGL(k_cal_copy, dim3(grid), dim3(256), 0, st, src, dst, n4);
        (void)hipEventRecord(e1, st);
        if (hipEventSynchronize(e1) != hipSuccess) break;
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        out->copy_ms = ms;
        out->copy_gbs = ms > 0 ? (float)(2.0 * (double)bytes * reps / (ms * 1e-3) / 1e9) : 0.0f;
        // VALU issue: one untimed launch, then ~50 ms of launches (one launch: 2048 x 4 waves x 32768 v_fma ~ 0.3 ms)
        hipLaunchKernelGGL(k_cal_valu, dim3(CAL_BLOCKS), dim3(256), 0, st, vout, 1.0001f, clk);
        const int vreps = 160;
        (void)hipEventRecord(e0, st);
        for (int k = 0; k < vreps; k++) hipLaunchKernelGGL(k_cal_valu, dim3(CAL_BLOCKS), dim3(256), 0, st, vout, 1.0001f, clk);
        (void)hipEventRecord(e1, st);
        if (hipEventSynchronize(e1) != hipSuccess) break;
        (void)hipEventElapsedTime(&ms, e0, e1);
        out->valu_ms = ms;
        int cus = 256;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        const double instr = (double)CAL_BLOCKS * 4.0 * CAL_ITERS * 8.0 * vreps;
        out->valu_cycles = (float)(ms * 1e-3 * 2.4e9 * (4.0 * cus) / instr);
        unsigned long long h[2] = {0, 0};
        if (hipMemcpyAsync(h, clk, sizeof h, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) break;
        // (s_memrealtime counts at 100 MHz; a ratio of ~1 means both counters are that clock on this stack: no figure)
        const double ratio = h[1] ? (double)h[0] / (double)h[1] : 0.0;
        out->clock_ghz = ratio > 1.5 ? (float)(ratio * 0.1) : 0.0f;
        rc = hipGetLastError() == hipSuccess ? SPH_OK : SPH_E_HIP;
    } while (0);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (src) (void)hipFree(src);
    if (dst) (void)hipFree(dst);
    if (vout) (void)hipFree(vout);
    if (clk) (void)hipFree(clk);
    if (st) (void)hipStreamDestroy(st);
    if (rc != SPH_OK) (void)hipGetLastError();
    return rc;
}

// ---- slab decomposition ----
int sph_slab_step_begin(sph_ctx *ctx, float gx, float gy) {
    if (!ctx || !ctx->stream) return SPH_E_ARG;
    if (!ctx->slab || ctx->slab_phase != 0) return fail(ctx, SPH_E_STATE, "sph_slab_step_begin: not a slab context or already mid-step");
    (void)hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    // kick 1/2 + drift of the owned range: done by the previous step's force pass (swap the sets; the ghost entries of
    // the swapped-in set are refreshed by this step's halo exchange) or by the stand-alone kernel; either has raised
    // the rebuild word if the lists may be stale
    if (fused(ctx) && ctx->primed) {
        std::swap(ctx->a.pos, ctx->a.pos2);
        std::swap(ctx->a.vel, ctx->a.vel2);
    } else {
        refresh_velt(ctx);
        launch_kick_drift(st, ctx->c, ctx->a, ctx->cap, true);
    }
    ctx->primed = fused(ctx);
    const float gravity[2] = {gx, gy};
    // beyond skin/2 somewhere: compare the boxes, verify the failing ones particle by particle; may raise the rebuild word
    PeerHead none = {};
    none.nranks = 1;
    launch_slab_head(st, ctx->c, ctx->a, ctx->cap, gravity, none, slab_verifies(ctx), ctx->slab_verify_most);
    ctx->lean_step++;      // (the device counts the same steps: the tags of the lean step's messages, should the host switch to it)
    HIPCHK(ctx, hipGetLastError());
    ctx->slab_phase = 1;
    return SPH_OK;
}

int sph_slab_step_pack(sph_ctx *ctx) {
    if (!ctx || !ctx->stream) return SPH_E_ARG;
    if (!ctx->slab || ctx->slab_phase != 1) return fail(ctx, SPH_E_STATE, "sph_slab_step_pack without sph_slab_step_begin");
    (void)hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    // rebuild step: full records (ghosts + migration); other steps: x, y, u, v of the interface columns
    launch_halo_out(st, ctx->c, ctx->a, ctx->cap);
    HIPCHK(ctx, hipGetLastError());
    ctx->slab_phase = 2;
    return SPH_OK;
}

int sph_slab_step_overlap(sph_ctx *ctx) {
    if (!ctx || !ctx->stream) return SPH_E_ARG;
    if (!ctx->slab || ctx->slab_phase != 2) return fail(ctx, SPH_E_STATE, "sph_slab_step_overlap without sph_slab_step_pack");
    (void)hipSetDevice(ctx->device);
    // density of the tiles that stage no ghost particle (nothing on a rebuild step): independent of the incoming halo
    launch_density(ctx->stream, ctx->c, ctx->a, ctx->cap, DENS_RHO_EOS, ctx->variant, false, DENS_INTERIOR, false);
    HIPCHK(ctx, hipGetLastError());
    ctx->slab_overlapped = true;
    return SPH_OK;
}

int sph_slab_step_overlap_on(sph_ctx *ctx, void *hip_stream) {
    if (!ctx || !ctx->stream) return SPH_E_ARG;
    if (!ctx->slab || ctx->slab_phase < 1) return fail(ctx, SPH_E_STATE, "sph_slab_step_overlap_on without sph_slab_step_begin");
    (void)hipSetDevice(ctx->device);
    hipStream_t st = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->stream;
    launch_density(st, ctx->c, ctx->a, ctx->cap, DENS_RHO_EOS, ctx->variant, false, DENS_INTERIOR, false);
    HIPCHK(ctx, hipGetLastError());
    ctx->slab_overlapped = true;
    return SPH_OK;
}

int sph_slab_step_end(sph_ctx *ctx) {
    if (!ctx || !ctx->stream) return SPH_E_ARG;
    if (!ctx->slab || ctx->slab_phase != 2) return fail(ctx, SPH_E_STATE, "sph_slab_step_end without sph_slab_step_pack");
    (void)hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    if (ctx->rebuild_wgs > 0) {      // the host asked for it (sph_set_rebuild_launches): nothing else computes on the device meanwhile
        launch_rebuild_slab(st, ctx->c, ctx->a, ctx->cap, ctx->rebuild_wgs, ctx->deterministic);
    } else {
        ctx->step_done_synced = false;                              // (FLAG_STEP_DONE is k_rebuild_slab's to keep: sph_slab_steps catches up)
        launch_halo_in(st, ctx->c, ctx->a, ctx->cap);               // rebuild step: ingest (then the next four); else ghost update
        launch_scan(st, ctx->c, ctx->a.count, ctx->a.dirty, ctx->a.cell_start, ctx->a.block_sums, ctx->a.rebuild, false);
        launch_reorder(st, ctx->c, ctx->a, ctx->cap, ctx->deterministic);
        launch_canon(st, ctx->c, ctx->a);
        launch_build_list(st, ctx->c, ctx->a, ctx->cap);
    }
    launch_density(st, ctx->c, ctx->a, ctx->cap, DENS_RHO_EOS, ctx->variant, true, ctx->slab_overlapped ? DENS_REST : DENS_ALL, false);
    launch_force(st, ctx->c, ctx->a, ctx->cap, fused(ctx) ? FORCE_KICK_DRIFT : FORCE_KICK, ctx->variant);
    ctx->velt_stale = fused(ctx);
    ctx->acc_stale = fused(ctx);
    ctx->p_stale = true;
    ctx->stepped = true;
    HIPCHK(ctx, hipGetLastError());
    ctx->slab_overlapped = false;
    ctx->slab_phase = 0;
    return SPH_OK;
}

int sph_slab_set_peer_links(sph_ctx *ctx, const sph_peer_links *links) {
    if (!ctx || !ctx->stream || !ctx->slab) return SPH_E_ARG;
    if (ctx->slab_phase != 0) return fail(ctx, SPH_E_STATE, "sph_slab_set_peer_links mid-step");
    (void)hipSetDevice(ctx->device);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    drop_graph(ctx);      // (the links are arguments of the captured head / rebuild launches: sph_slab_steps)
    if (!links) { ctx->has_links = false; return SPH_OK; }
    if (links->n_ranks < 1 || links->n_ranks > SPH_PEER_MAX_RANKS || links->me < 0 || links->me >= links->n_ranks)
        return fail(ctx, SPH_E_ARG, "sph_slab_set_peer_links: rank out of range");
    for (int q = 0; q < links->n_ranks; q++)
        if (!links->slots_of_rank[q]) return fail(ctx, SPH_E_ARG, "sph_slab_set_peer_links: a rank's slot array is missing");
    const bool l = ctx->c.has_left != 0, r = ctx->c.has_right != 0;
    if ((l && !(links->left_recv[0] && links->left_recv[1] && links->left_flag && links->my_recv_left[0] && links->my_recv_left[1] && links->my_flag_left)) ||
        (r && !(links->right_recv[0] && links->right_recv[1] && links->right_flag && links->my_recv_right[0] && links->my_recv_right[1] && links->my_flag_right)))
        return fail(ctx, SPH_E_ARG, "sph_slab_set_peer_links: a neighbour's buffers or flags are missing");
    ctx->links = *links;
    ctx->has_links = true;
    if (ctx->lean_spec == 2) {
        int rc = upload_fuse(ctx);
        if (rc) return rc;
        if (ctx->fuse_blocks > 508) { ctx->lean_spec = 1; return upload_jobs(ctx); }
    }
    return SPH_OK;
}

// Test hook, compiled with -DSPH_TEST_HOOKS only (tests/test_slab_c_host.py): $SPH_TEST_STALL_AFTER_HEAD = "rank:microseconds:every" holds that rank's HOST for so long
// between the head kernel and the rest of every `every`-th step — what time-slicing does to ranks that share a device.  Its
// neighbours run a launch ahead meanwhile (their next head kernel pushes its update and raises the arrival flag past the
// value this rank is about to wait for: peer_wait).
#ifdef SPH_TEST_HOOKS      // (`make stress`: libsph_hip_co.so + host/slab_sph_fluid_stress; the product build has no such hook)
static void test_stall_after_head(int me, uint32_t step) {
    static int rank = -2, us = 0, every = 1;
    if (rank == -2) {
        rank = -1;
        const char *e = getenv("SPH_TEST_STALL_AFTER_HEAD");
        if (e && sscanf(e, "%d:%d:%d", &rank, &us, &every) < 2) rank = -1;
        if (every < 1) every = 1;
    }
    if (me == rank && us > 0 && step % (uint32_t)every == 0u) usleep((useconds_t)us);
}
#else
static inline void test_stall_after_head(int, uint32_t) {}
#endif

namespace {
// what the two kernels of the lean step that talk to the other ranks need of the links.  step = 0: a node of a graph — the kernels take
// the step's number (and with it the parity of the buffers) from the device (FLAG_STEP_DONE / FLAG_STEP, sph_internal.h).
bool fill_peer(sph_ctx *ctx, uint32_t step, PeerHead &ph, PeerLinks &pl) {
    const sph_peer_links &L = ctx->links;
    const bool peer = ctx->has_links && L.n_ranks > 1;
    ph = PeerHead{};
    pl = PeerLinks{};
    ph.nranks = 1;
    ph.step = pl.step = step;
    if (!peer) return false;
    for (int q = 0; q < L.n_ranks; q++) ph.slots_of_rank[q] = pl.slots_of_rank[q] = static_cast<uint32_t *>(L.slots_of_rank[q]);
    ph.my_slots = pl.my_slots = static_cast<const uint32_t *>(L.slots_of_rank[L.me]);
    ph.me = pl.me = L.me;
    ph.nranks = pl.nranks = L.n_ranks;
    for (int par = 0; par < 2; par++) {
        if (ctx->c.has_left) {
            ph.remote_l[par] = pl.remote_l[par] = static_cast<uint32_t *>(L.left_recv[par]);
            pl.recv_l[par] = static_cast<uint32_t *>(L.my_recv_left[par]);
        }
        if (ctx->c.has_right) {
            ph.remote_r[par] = pl.remote_r[par] = static_cast<uint32_t *>(L.right_recv[par]);
            pl.recv_r[par] = static_cast<uint32_t *>(L.my_recv_right[par]);
        }
    }
    if (ctx->c.has_left) {
        ph.flag_l = pl.flag_l = static_cast<uint32_t *>(L.left_flag);
        pl.my_flag_l = static_cast<const uint32_t *>(L.my_flag_left);
    }
    if (ctx->c.has_right) {
        ph.flag_r = pl.flag_r = static_cast<uint32_t *>(L.right_flag);
        pl.my_flag_r = static_cast<const uint32_t *>(L.my_flag_right);
    }
    return true;
}

// mode 2: the record the head blocks of the fused density launch read (after the links or the mode have changed)
int upload_fuse(sph_ctx *ctx) {
    if (!ctx->slab || !ctx->d_fuse) return SPH_OK;
    SlabFuse F = {};
    PeerLinks pl;
    const bool peer = fill_peer(ctx, 0u, F.ph, pl);
    for (int par = 0; par < 2; par++) { F.recv_l[par] = pl.recv_l[par]; F.recv_r[par] = pl.recv_r[par]; }
    F.my_flag_l = pl.my_flag_l;
    F.my_flag_r = pl.my_flag_r;
    F.send_l = ctx->a.send[0];
    F.send_r = ctx->a.send[1];
    F.grav = ctx->a.grav;
    F.gring = ctx->d_gring;
    ctx->fuse_blocks = slab_fuse_blocks(ctx->c, peer, &F.npush, &F.nupd);
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_fuse, &F, sizeof F, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));      // (a local)
    return SPH_OK;
}

// mode 2: the gravity of steps first .. first + n - 1 into the ring (the sample of step s at s mod GRAV_RING), uploaded when it differs from
// what the device holds
int feed_gravity_ring(sph_ctx *ctx, uint32_t first, const float *gravity_xy, int n) {
    float want[2 * GRAV_RING];
    memcpy(want, ctx->h_gring, sizeof want);
    for (int k = 0; k < n; k++) {
        const uint32_t at = (first + (uint32_t)k) % (uint32_t)GRAV_RING;
        want[2 * at] = gravity_xy[2 * k];
        want[2 * at + 1] = gravity_xy[2 * k + 1];
    }
    if (ctx->gring_valid && memcmp(want, ctx->h_gring, sizeof want) == 0) return SPH_OK;
    const int slot = ctx->gseq_slot++ % GSEQ_SLOTS;
    if (ctx->gseq_used[slot]) HIPCHK(ctx, hipEventSynchronize(ctx->gseq_ev[slot]));
    float *h = ctx->h_gseq + (size_t)slot * 2 * MULTI_STEPS;
    static_assert(GRAV_RING == MULTI_STEPS, "a run of MULTI_STEPS steps fills the ring once, and a pinned staging slot holds a whole ring");
    memcpy(h, want, sizeof want);
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_gring, h, sizeof want, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipEventRecord(ctx->gseq_ev[slot], ctx->stream));
    ctx->gseq_used[slot] = true;
    memcpy(ctx->h_gring, want, sizeof want);
    ctx->gring_valid = true;
    return SPH_OK;
}

// the four launches of one lean step; gravity: this step's (gx, gy) as launch arguments (one call per step), or gravity_dev: where the
// head kernel finds them in device memory (a node of a graph)
void enqueue_lean_step(sph_ctx *ctx, const float *gravity, const float *gravity_dev, uint32_t step, bool stall_hook) {
    hipStream_t st = ctx->stream;
    PeerHead ph;
    PeerLinks pl;
    const bool peer = fill_peer(ctx, step, ph, pl);
    if (ctx->lean_spec == 2) {
        // the FUSED speculative lean step: three launches, as sph_step — the head's work by the first workgroups of the density launch
        // (everything per step from the device: SlabFuse), the gate, the force pass
        launch_density(st, ctx->c, ctx->a, ctx->cap, DENS_RHO_EOS, ctx->variant, false, DENS_ALL, false, true,
                       ctx->verify_mode < 0 ? ctx->n >= VERIFY_MIN_PARTICLES : ctx->verify_mode > 0, ctx->fuse_blocks > 0 ? ctx->fuse_blocks : 4);      // 1
        Arrays ga = ctx->a;
        if (!list_repair(ctx)) { ga.rq = nullptr; ga.xpair = nullptr; }
        pl.step = 0u;
        launch_rebuild_slab(st, ctx->c, ga, ctx->cap, ctx->rebuild_wgs, ctx->deterministic, peer ? 7 : 5, peer ? &pl : nullptr);       // 2
        launch_force(st, ctx->c, ctx->a, ctx->cap, FORCE_KICK_DRIFT, ctx->variant);                                                    // 3
        return;
    }
    if (ctx->lean_spec) {
        // the speculative lean step: books + push + ghost update | density with the criterion's jobs in its launch (as sph_step) | the gate:
        // MAX of the word over the ranks, then nothing, or the rebuild and the density pass again | force
        launch_slab_head(st, ctx->c, ctx->a, ctx->cap, gravity, ph, false, 0, gravity_dev, true, peer ? &pl : nullptr);                  // 1
        if (peer && stall_hook) test_stall_after_head(ctx->links.me, step);
        launch_density(st, ctx->c, ctx->a, ctx->cap, DENS_RHO_EOS, ctx->variant, false, DENS_ALL, false, true,
                       ctx->verify_mode < 0 ? ctx->n >= VERIFY_MIN_PARTICLES : ctx->verify_mode > 0);                                   // 2
        Arrays ga = ctx->a;
        if (!list_repair(ctx)) { ga.rq = nullptr; ga.xpair = nullptr; }
        launch_rebuild_slab(st, ctx->c, ga, ctx->cap, ctx->rebuild_wgs, ctx->deterministic, peer ? 7 : 5, peer ? &pl : nullptr);       // 3
        launch_force(st, ctx->c, ctx->a, ctx->cap, FORCE_KICK_DRIFT, ctx->variant);                                                    // 4
        return;
    }
    launch_slab_head(st, ctx->c, ctx->a, ctx->cap, gravity, ph, slab_verifies(ctx), ctx->slab_verify_most, gravity_dev);                 // 1
    if (peer && stall_hook) test_stall_after_head(ctx->links.me, step);
    launch_rebuild_slab(st, ctx->c, ctx->a, ctx->cap, ctx->rebuild_wgs, ctx->deterministic, peer ? 3 : 1, peer ? &pl : nullptr);   // 2
    launch_density(st, ctx->c, ctx->a, ctx->cap, DENS_RHO_EOS, ctx->variant, true, DENS_ALL, false);       // 3
    launch_force(st, ctx->c, ctx->a, ctx->cap, FORCE_KICK_DRIFT, ctx->variant);                            // 4
}

int lean_step_ready(sph_ctx *ctx, const char *who) {
    if (!ctx->slab || ctx->slab_phase != 0) return fail(ctx, SPH_E_STATE, (std::string(who) + ": not a slab context, or mid-step").c_str());
    if (ctx->rebuild_wgs <= 0)
        return fail(ctx, SPH_E_STATE, (std::string(who) + " needs the one-launch rebuild: sph_set_rebuild_launches(ctx, 1) (nothing else may compute on the device)").c_str());
    if (!fused(ctx)) return fail(ctx, SPH_E_STATE, (std::string(who) + ": list kernels only (variant 0)").c_str());
    if ((ctx->c.has_left || ctx->c.has_right) && !ctx->has_links)
        return fail(ctx, SPH_E_STATE, (std::string(who) + ": this slab has neighbours: sph_slab_set_peer_links first (or the three-call step with a transport of the host's)").c_str());
    return SPH_OK;
}

// 2^k consecutive lean steps (primed loop) as ONE graph for the current orientation of the two position / velocity sets: every launch
// takes its step number, the parity of its buffers and its gravity from device memory (gravity of step s of the run: d_gseq[2 s ..])
hipGraphExec_t lean_graph(sph_ctx *ctx, int steps) {
    if (!ctx->use_graph || steps < 2 || steps > MULTI_STEPS || (steps & (steps - 1)) != 0) return nullptr;
    const int k = 2 * (31 - __builtin_clz((unsigned)steps)) + (ctx->a.pos == ctx->pos_a ? 0 : 1);
    if (ctx->gexec[k]) return ctx->gexec[k];
    if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        ctx->use_graph = false;
        return nullptr;
    }
    for (int s = 0; s < steps; s++) {
        std::swap(ctx->a.pos, ctx->a.pos2);
        std::swap(ctx->a.vel, ctx->a.vel2);
        enqueue_lean_step(ctx, nullptr, ctx->d_gseq + 2 * s, 0u, false);
    }
    hipError_t e = hipStreamEndCapture(ctx->stream, &ctx->graph[k]);
    if (e == hipSuccess) e = hipGraphInstantiate(&ctx->gexec[k], ctx->graph[k], nullptr, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        drop_graph(ctx);
        ctx->use_graph = false;
        return nullptr;
    }
    return ctx->gexec[k];
}
}  // namespace

int sph_slab_set_speculative(sph_ctx *ctx, int on) {
    if (!ctx || !ctx->stream || !ctx->slab || on < 0 || on > 2) return SPH_E_ARG;
    if (ctx->slab_phase != 0) return fail(ctx, SPH_E_STATE, "sph_slab_set_speculative mid-step");
    (void)hipSetDevice(ctx->device);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    drop_graph(ctx);
    ctx->lean_spec = on;
    if (on == 2) {
        int rc = upload_fuse(ctx);
        if (rc) return rc;
        if (ctx->fuse_blocks > 508) ctx->lean_spec = 1;      // (more head blocks than the launch's last argument can say: the four-launch form)
    }
    return upload_jobs(ctx);      // (repaired tiles are queued for a repeat of their density only where a density pass runs beside the repairs)
}

int sph_slab_step(sph_ctx *ctx, float gx, float gy) {
    if (!ctx || !ctx->stream) return SPH_E_ARG;
    int rc = lean_step_ready(ctx, "sph_slab_step");
    if (rc) return rc;
    (void)hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    // kick 1/2 + drift of the owned range: the previous step's force pass made it (swap the sets), else the stand-alone kernel;
    // either has left this step's update message in the send buffers
    if (ctx->primed) {
        std::swap(ctx->a.pos, ctx->a.pos2);
        std::swap(ctx->a.vel, ctx->a.vel2);
    } else {
        refresh_velt(ctx);
        launch_kick_drift(st, ctx->c, ctx->a, ctx->cap, true);
    }
    ctx->primed = true;
    const uint32_t step = ++ctx->lean_step, par = step & 1u;
    if (ctx->has_links && ctx->links.n_ranks > 1) {      // (what sph_slab_buffers reports: the receive buffers of this step's parity)
        if (ctx->c.has_left) ctx->a.recv[0] = static_cast<uint32_t *>(ctx->links.my_recv_left[par]);
        if (ctx->c.has_right) ctx->a.recv[1] = static_cast<uint32_t *>(ctx->links.my_recv_right[par]);
    }
    const float gravity[2] = {gx, gy};
    if (ctx->lean_spec == 2) {      // (everything per step from the device: the step number from FLAG_STEP_DONE, the gravity from the ring)
        rc = feed_gravity_ring(ctx, step, gravity, 1);
        if (rc) return rc;
        if (!ctx->step_done_synced) {
            HIPCHK(ctx, hipMemcpyAsync(ctx->a.flags + FLAG_STEP_DONE, ctx->a.flags + FLAG_STEP, sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
            ctx->step_done_synced = true;
        }
        enqueue_lean_step(ctx, nullptr, nullptr, 0u, false);
    } else
    enqueue_lean_step(ctx, gravity, nullptr, step, true);
    ctx->velt_stale = true;
    ctx->acc_stale = true;
    ctx->p_stale = true;
    ctx->stepped = true;
    HIPCHK(ctx, hipGetLastError());
    return SPH_OK;
}

int sph_slab_steps(sph_ctx *ctx, const float *gravity_xy, int nsteps) {
    if (!ctx || !ctx->stream || nsteps < 0 || (nsteps > 0 && !gravity_xy)) return SPH_E_ARG;
    int rc = lean_step_ready(ctx, "sph_slab_steps");
    if (rc) return rc;
    (void)hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    int s = 0;
    while (s < nsteps) {
        int m = MULTI_STEPS;
        while (m > nsteps - s) m >>= 1;
        hipGraphExec_t g = ctx->primed && m >= 2 ? lean_graph(ctx, m) : nullptr;
        if (!g) {      // the first step after creation / an upload (no look-ahead yet), a remainder of one step, or no graphs on this stream
            rc = sph_slab_step(ctx, gravity_xy[2 * s], gravity_xy[2 * s + 1]);
            if (rc) return rc;
            s++;
            continue;
        }
        // this run's gravity samples -> device (skipped while they repeat what is there: a 10 Hz gravity source changes every ~400 steps)
        const float *gs = gravity_xy + 2 * s;
        if (ctx->lean_spec == 2) {      // (the fused step reads the ring: the sample of step t at t mod GRAV_RING)
            rc = feed_gravity_ring(ctx, ctx->lean_step + 1u, gs, m);
            if (rc) return rc;
        } else
        if (!ctx->gseq_valid || memcmp(ctx->h_gseq_last, gs, sizeof(float) * 2 * (size_t)m) != 0) {
            const int slot = ctx->gseq_slot++ % GSEQ_SLOTS;
            if (ctx->gseq_used[slot]) HIPCHK(ctx, hipEventSynchronize(ctx->gseq_ev[slot]));      // (the copy that read this slot last has run)
            float *h = ctx->h_gseq + (size_t)slot * 2 * MULTI_STEPS;
            memcpy(h, gs, sizeof(float) * 2 * (size_t)m);
            HIPCHK(ctx, hipMemcpyAsync(ctx->d_gseq, h, sizeof(float) * 2 * (size_t)m, hipMemcpyHostToDevice, st));
            HIPCHK(ctx, hipEventRecord(ctx->gseq_ev[slot], st));
            ctx->gseq_used[slot] = true;
            memcpy(ctx->h_gseq_last, gs, sizeof(float) * 2 * (size_t)m);
            // (entries beyond m keep whatever an earlier, longer run left: a shorter run that matches the first m entries is still right,
            // a longer one compares all of its own)
            for (int k = 2 * m; k < 2 * MULTI_STEPS; k++) ctx->h_gseq_last[k] = NAN;
            ctx->gseq_valid = true;
        }
        if (!ctx->step_done_synced) {      // (the three-call step with one kernel per phase does not keep FLAG_STEP_DONE)
            HIPCHK(ctx, hipMemcpyAsync(ctx->a.flags + FLAG_STEP_DONE, ctx->a.flags + FLAG_STEP, sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
            ctx->step_done_synced = true;
        }
        HIPCHK(ctx, hipGraphLaunch(g, st));
        ctx->lean_step += (uint32_t)m;
        if (ctx->has_links && ctx->links.n_ranks > 1) {      // (an even number of steps: the parity of the last one)
            const uint32_t par = ctx->lean_step & 1u;
            if (ctx->c.has_left) ctx->a.recv[0] = static_cast<uint32_t *>(ctx->links.my_recv_left[par]);
            if (ctx->c.has_right) ctx->a.recv[1] = static_cast<uint32_t *>(ctx->links.my_recv_right[par]);
        }
        ctx->velt_stale = true;
        ctx->acc_stale = true;
        ctx->p_stale = true;
        ctx->stepped = true;
        s += m;
    }
    HIPCHK(ctx, hipGetLastError());
    return SPH_OK;
}

size_t sph_slab_halo_bytes(const sph_params *prm, int halo_capacity) {
    Consts cg;
    if (!prm || make_consts(*prm, cg) != SPH_OK) return 0;
    const size_t cap = halo_capacity > 0 ? (size_t)halo_capacity : (size_t)4 * cg.rows * 16;      // the default of sph_create_slab
    return sizeof(uint32_t) * (HALO_HDR + (size_t)HALO_REC * cap);
}

int sph_slab_flag_buffer(sph_ctx *ctx, void **dev_word) {
    if (!ctx || !ctx->slab || !dev_word) return SPH_E_ARG;
    *dev_word = ctx->a.rebuild;
    return SPH_OK;
}

int sph_slab_set_flag_buffer(sph_ctx *ctx, void *dev_word) {
    if (!ctx || !ctx->slab || !ctx->stream) return SPH_E_ARG;
    if (ctx->slab_phase != 0) return fail(ctx, SPH_E_STATE, "sph_slab_set_flag_buffer mid-step");
    (void)hipSetDevice(ctx->device);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->a.rebuild = dev_word ? static_cast<uint32_t *>(dev_word) : ctx->a.flags + FLAG_REBUILD;
    HIPCHK(ctx, hipMemsetAsync(ctx->a.rebuild, 0, sizeof(uint32_t), ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return upload_jobs(ctx);      // (the criterion's jobs of the head kernel raise the word: they must know where it is)
}

int sph_slab_flag_get(sph_ctx *ctx, uint32_t *value) {
    if (!ctx || !ctx->slab || !ctx->stream || !value) return SPH_E_ARG;
    (void)hipSetDevice(ctx->device);
    HIPCHK(ctx, hipMemcpyAsync(value, ctx->a.rebuild, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SPH_OK;
}

int sph_slab_flag_set(sph_ctx *ctx, uint32_t value) {
    if (!ctx || !ctx->slab || !ctx->stream) return SPH_E_ARG;
    (void)hipSetDevice(ctx->device);
    launch_set_rebuild(ctx->stream, ctx->a, value != 0u);
    HIPCHK(ctx, hipGetLastError());
    return SPH_OK;
}

int sph_slab_buffers(sph_ctx *ctx, void **send_left, void **send_right, void **recv_left, void **recv_right, size_t *bytes) {
    if (!ctx || !ctx->slab) return SPH_E_ARG;
    if (send_left) *send_left = ctx->a.send[0];
    if (send_right) *send_right = ctx->a.send[1];
    if (recv_left) *recv_left = ctx->a.recv[0];
    if (recv_right) *recv_right = ctx->a.recv[1];
    if (bytes) *bytes = ctx->halo_bytes;
    return SPH_OK;
}

int sph_slab_set_buffers(sph_ctx *ctx, void *send_left, void *send_right, void *recv_left, void *recv_right, size_t bytes) {
    if (!ctx || !ctx->slab || !send_left || !send_right || !recv_left || !recv_right) return SPH_E_ARG;
    if (bytes < ctx->halo_bytes) return fail(ctx, SPH_E_ARG, "sph_slab_set_buffers: buffers smaller than the halo capacity");
    if (ctx->slab_phase != 0) return fail(ctx, SPH_E_STATE, "sph_slab_set_buffers mid-step");
    (void)hipSetDevice(ctx->device);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    // (the send buffers hold the NEXT step's update message, written by the last force pass: it moves along)
    uint32_t *const old_send[2] = {ctx->a.send[0], ctx->a.send[1]};
    ctx->a.send[0] = static_cast<uint32_t *>(send_left);
    ctx->a.send[1] = static_cast<uint32_t *>(send_right);
    ctx->a.recv[0] = static_cast<uint32_t *>(recv_left);
    ctx->a.recv[1] = static_cast<uint32_t *>(recv_right);
    ctx->own_halo = false;
    for (int k = 0; k < 2; k++) {
        if (old_send[k] && old_send[k] != ctx->a.send[k])
            HIPCHK(ctx, hipMemcpyAsync(ctx->a.send[k], old_send[k], ctx->halo_bytes, hipMemcpyDeviceToDevice, ctx->stream));
        else if (!old_send[k]) HIPCHK(ctx, hipMemsetAsync(ctx->a.send[k], 0, ctx->halo_bytes, ctx->stream));
        HIPCHK(ctx, hipMemsetAsync(ctx->a.recv[k], 0, ctx->halo_bytes, ctx->stream));
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    drop_graph(ctx);      // (the buffers are arguments of captured launches)
    if (ctx->lean_spec == 2) return upload_fuse(ctx);      // (... and fields of the fused step's device record)
    return SPH_OK;
}

int sph_slab_copy_out(sph_ctx *ctx, int side, void *host) {
    if (!ctx || !ctx->slab || side < 0 || side > 1 || !host) return SPH_E_ARG;
    (void)hipSetDevice(ctx->device);
    HIPCHK(ctx, hipMemcpyAsync(host, ctx->a.send[side], ctx->halo_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SPH_OK;
}

int sph_slab_copy_in(sph_ctx *ctx, int side, const void *host) {
    if (!ctx || !ctx->slab || side < 0 || side > 1 || !host) return SPH_E_ARG;
    (void)hipSetDevice(ctx->device);
    HIPCHK(ctx, hipMemcpyAsync(ctx->a.recv[side], host, ctx->halo_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SPH_OK;
}

int sph_slab_peer_reduce(sph_ctx *ctx, void *const *slots_of_rank, int me, int n_ranks, uint32_t tag) {
    if (!ctx || !ctx->stream || !ctx->slab || !slots_of_rank || n_ranks < 1 || n_ranks > SPH_PEER_MAX_RANKS || me < 0 || me >= n_ranks)
        return SPH_E_ARG;
    if (ctx->slab_phase != 1) return fail(ctx, SPH_E_STATE, "sph_slab_peer_reduce: between sph_slab_step_begin and sph_slab_step_pack");
    for (int q = 0; q < n_ranks; q++)
        if (!slots_of_rank[q]) return SPH_E_ARG;
    (void)hipSetDevice(ctx->device);
    launch_peer_reduce(ctx->stream, ctx->a, slots_of_rank, slots_of_rank[me], me, n_ranks, tag);
    HIPCHK(ctx, hipGetLastError());
    return SPH_OK;
}

int sph_slab_peer_push(sph_ctx *ctx, void *left_recv_right, void *left_flag, void *right_recv_left, void *right_flag, uint32_t tag) {
    if (!ctx || !ctx->stream || !ctx->slab || (!left_recv_right) != (!left_flag) || (!right_recv_left) != (!right_flag)) return SPH_E_ARG;
    if (ctx->slab_phase != 2) return fail(ctx, SPH_E_STATE, "sph_slab_peer_push without sph_slab_step_pack");
    (void)hipSetDevice(ctx->device);
    launch_peer_push(ctx->stream, ctx->c, ctx->a, left_recv_right, left_flag, right_recv_left, right_flag, tag);
    HIPCHK(ctx, hipGetLastError());
    return SPH_OK;
}

int sph_slab_peer_wait(sph_ctx *ctx, const void *flag_from_left, const void *flag_from_right, uint32_t tag) {
    if (!ctx || !ctx->stream || !ctx->slab) return SPH_E_ARG;
    (void)hipSetDevice(ctx->device);
    launch_peer_wait(ctx->stream, ctx->a, flag_from_left, flag_from_right, tag);
    HIPCHK(ctx, hipGetLastError());
    return SPH_OK;
}

int sph_slab_read(sph_ctx *ctx, sph_particle *out, uint32_t *ids, float *du_dt, float *dv_dt, int cap, int *n_out) {
    if (!ctx || !ctx->stream || !ctx->slab || !n_out) return SPH_E_ARG;
    if (ctx->slab_phase != 0) return fail(ctx, SPH_E_STATE, "sph_slab_read mid-step");
    (void)hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    refresh_velt(ctx);
    refresh_p(ctx);
    launch_export_owned(st, ctx->c, ctx->a, ctx->cap, ctx->d_aos, ctx->d_ids, ctx->d_du, ctx->d_dv);
    uint32_t hdn[4] = {0, 0, 0, 0};
    HIPCHK(ctx, hipMemcpyAsync(hdn, ctx->a.dn, sizeof hdn, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    const int n = (int)hdn[2];
    *n_out = n;
    if (n > cap) return fail(ctx, SPH_E_CAPACITY, "sph_slab_read: output capacity too small");
    if (out) HIPCHK(ctx, hipMemcpyAsync(out, ctx->d_aos, (size_t)n * sizeof(sph_particle), hipMemcpyDeviceToHost, st));
    if (ids) HIPCHK(ctx, hipMemcpyAsync(ids, ctx->d_ids, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    if (du_dt) HIPCHK(ctx, hipMemcpyAsync(du_dt, ctx->d_du, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, st));
    if (dv_dt) HIPCHK(ctx, hipMemcpyAsync(dv_dt, ctx->d_dv, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    return SPH_OK;
}

int sph_slab_counts(sph_ctx *ctx, int *n_local, int *n_owned) {
    if (!ctx || !ctx->stream || !ctx->slab) return SPH_E_ARG;
    (void)hipSetDevice(ctx->device);
    // local = everything in the sorted arrays; owned = the range of the owned columns (as of the last rebuild)
    uint32_t hdn[4] = {0, 0, 0, 0}, lo = 0, hi = 0;
    const Consts &c = ctx->c;
    HIPCHK(ctx, hipMemcpyAsync(hdn, ctx->a.dn, sizeof hdn, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(&lo, ctx->a.cell_start + (size_t)c.ghost * c.rows, sizeof lo, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(&hi, ctx->a.cell_start + (size_t)(c.ghost + c.owned) * c.rows, sizeof hi, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (n_local) *n_local = (int)hdn[0];
    if (n_owned) *n_owned = (int)(hi - lo);
    return SPH_OK;
}

// ---- metaballs (next row f1) ----
int sph_render_metaballs(sph_ctx *ctx, unsigned char *draw_buffer) {
    if (!ctx || !ctx->stream || !draw_buffer) return SPH_E_ARG;
    if (ctx->slab && ctx->slab_phase != 0) return fail(ctx, SPH_E_STATE, "sph_render_metaballs mid-step");
    (void)hipSetDevice(ctx->device);
    // the kernel emits the SSD1306 page format itself (:407-408): 1 KB crosses PCIe, straight into the caller's buffer
    launch_metaballs(ctx->stream, ctx->c, ctx->a, ctx->prm.x_max - ctx->prm.x_min, ctx->prm.y_max - ctx->prm.y_min,
                     ctx->d_bits);
    HIPCHK(ctx, hipMemcpyAsync(draw_buffer, ctx->d_bits, 1024, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SPH_OK;
}

}  // extern "C"
