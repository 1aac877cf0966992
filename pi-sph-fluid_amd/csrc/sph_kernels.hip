// sph_kernels.hip — hand-written gfx950 kernels of the 2-D WCSPH step.
//
// What each kernel restates (file = /root/reference/pi_sph_fluid.c):
//   k_kick_drift       kick 1/2 + drift :615-624, in place; asks for a neighbour-structure rebuild when a particle
//                      has moved more than half the skin since the last one
//   k_kick_drift_key   cell index :111-113 + histogram (counting sort pass 1); slab mode: with kick 1/2 + drift
//   k_scan_*           exclusive scan of the cell histogram (counting sort pass 2)
//   k_reorder          scatter into cell-contiguous order (replaces the linked list :104-124)
//   k_build_list       find_neighbors :126-153 once per rebuild: per-particle neighbour lists (sph_list.inc)
//   k_density_*        calculate_density :263-289 (+ calculate_particle_pressure :294-301 when fused)
//   k_force_*          calculate_accelerations :303-373 (+ kick 1/2 :637-640 when fused)
//   k_boundary_psi     calculate_boundary_pseudomass :242-261
//
// Data layout: SoA, cell-sorted at every rebuild of the neighbour structure (the reference rebuilds its
// linked list every step, :626; here a rebuild happens when the lists could have gone stale — every step with
// skin = 0).  Cells are linearised column-major (cell = col*rows + row), so the 3x3 neighbourhood of a particle is
// THREE contiguous ranges of the sorted arrays (one per column: rows r-1..r+1).  The kernels marked "rebuild" start
// with `if (!flags[FLAG_REBUILD]) return;` so that one captured graph serves every step.  No MFMA anywhere: there
// is no dense contraction in this workload; the kernels are HBM/LDS/VALU work.
#include <cstdlib>

#include "sph_internal.h"

namespace sph {

#define DEV __device__ __forceinline__

constexpr int BLK = 256;   // 4 waves of 64
// grid of a kernel that returns at once in most steps (the rebuild kernels): small, striding over its work
constexpr int GATED_GRID_MAX = 2048;
static inline int gated_grid(int work_blocks) { return work_blocks < GATED_GRID_MAX ? (work_blocks > 0 ? work_blocks : 1) : GATED_GRID_MAX; }

DEV bool finite_bits(float x) { return (__float_as_uint(x) & 0x7fffffffu) < 0x7f800000u; }

// cell of a position, clamped into the grid; out-of-range and NaN are reported through `oob` / `bad`.
// The reference bins with (int)((y - y_min) / cell_length) (:111-112).  The device multiplies by 1/cell instead:
// an IEEE f32 division costs ~20 instructions plus two FP-mode switches per call here (it was 37 us of the force
// kernel at 2M particles).  A position within 1 ulp of a cell edge may land in the adjacent cell; that cannot lose
// a neighbour that matters: a pair missed that way is at distance > 2H(1 - 1e-7), where W and grad W vanish like
// (1 - q/2)^4 and (1 - q/2)^3.  Every kernel uses this one function, so keys, tiles and ranges agree.
DEV void cell_of(const Consts &c, float x, float y, int &row, int &col, bool &oob, bool &bad) {
    bad = !(finite_bits(x) && finite_bits(y));
    float fr = (y - c.y_min) * c.inv_cell, fc = (x - c.x_min) * c.inv_cell;
    row = bad ? 0 : (int)fr;
    col = bad ? 0 : (int)fc - c.col_off;          // local column (slab mode: col_off = first local column)
    oob = (fr < 0.0f) | (fc < 0.0f) | (col < 0) | (row >= c.rows) | (col >= c.cols);
    row = min(max(row, 0), c.rows - 1);
    col = min(max(col, 0), c.cols - 1);
}

// cell (row, col) of a sorted particle from its key (cells are column-major).  Used where the CURRENT position must
// not be used: between rebuilds a particle may have left the cell it is sorted under.
DEV void cell_of_key(const Consts &c, uint32_t key, int &row, int &col) {
    col = (int)(key / (uint32_t)c.rows);
    row = (int)(key - (uint32_t)col * (uint32_t)c.rows);
}

// cell_start with out-of-grid cells reading as empty: an index below 0 (a column left of the grid) reads cell_start[0] = 0,
// one beyond the grid (a column right of it) cell_start[n_cells] = n
DEV uint32_t cs_ext(const uint32_t *__restrict__ cs, int cell, int n_cells) { return cs[min(max(cell, 0), n_cells)]; }
// first sorted particle of cell (row, col); col may be -1 .. cols + 1 (empty columns), row 0 .. rows
DEV uint32_t cs_at(const Consts &c, const uint32_t *__restrict__ cs, int col, int row) { return cs_ext(cs, col * c.rows + row, c.n_cells); }
// Tile order.  The sorted arrays are column-major; the list kernels cut them into tiles TWO cell columns wide: columns
// 2P ("A") and 2P + 1 ("B") form pair P, and the tile order runs through a pair row by row — cell (row, A), cell (row, B),
// cell (row + 1, A) ... — then through the next pair.  Only the ORDER is virtual: a tile (256 consecutive particles of it)
// is, per pair it touches, one contiguous range of column A and one of column B.  Rank of the first particle of pair-row
// (P, row) in that order (equal to its index in the sorted arrays at the pair boundaries):
DEV int n_pairs(const Consts &c) { return (c.cols + 1) >> 1; }
DEV uint32_t tile_rank(const Consts &c, const uint32_t *__restrict__ cs, int pair, int row) {
    return cs_at(c, cs, 2 * pair, row) + cs_at(c, cs, 2 * pair + 1, row) - cs_at(c, cs, 2 * pair + 1, 0);
}

// Wendland C2 without its normalising factor: (1 - q/2)^4 (1 + 2q), q = d/H   (:45-50)
DEV float w_shape(const Consts &c, float d2) {
    float d = __builtin_amdgcn_sqrtf(d2);
    float q = d * c.inv_h;
    float a = fmaf(-0.5f, q, 1.0f);
    float a2 = a * a;
    return a2 * a2 * fmaf(2.0f, q, 1.0f);
}

// ------------------------------------------------------------------------------------------
// gravity lives in device memory so that a captured step graph can be replayed under a
// changing gravity vector (the reference re-reads g every step, :632)
__global__ void k_set_gravity(float2 *grav, float gx, float gy) { *grav = make_float2(gx, gy); }

void launch_set_gravity(hipStream_t st, const Arrays &a, float gx, float gy) {
    hipLaunchKernelGGL(k_set_gravity, dim3(1), dim3(1), 0, st, a.grav, gx, gy);
}

__global__ void k_set_word(uint32_t *word, uint32_t value) { *word = value; }
void launch_request_rebuild(hipStream_t st, const Arrays &a) {
    hipLaunchKernelGGL(k_set_word, dim3(1), dim3(1), 0, st, a.rebuild, (uint32_t)REBUILD_HOST);
}
__global__ void k_spec_reset(uint32_t *flags, uint32_t *vq, uint32_t *rebuild, int mode) {
    if (mode == 0) flags[FLAG_SAVED_WORD] = *rebuild;      // before the timed launches: note the word (the last force pass may have raised it) ...
    flags[FLAG_CHECK_DONE] = 0u;
    if (vq) vq[0] = 0u;
    *rebuild = mode == 2 ? flags[FLAG_SAVED_WORD] : 0u;    // ... keep it clear while they run (a raised word sends the tiles home), put it back afterwards
}
void launch_spec_reset(hipStream_t st, const Arrays &a, int mode) {
    hipLaunchKernelGGL(k_spec_reset, dim3(1), dim3(1), 0, st, a.flags, a.vq, a.rebuild, mode);
}
void launch_set_rebuild(hipStream_t st, const Arrays &a, bool on) {
    hipLaunchKernelGGL(k_set_word, dim3(1), dim3(1), 0, st, a.rebuild, on ? (uint32_t)REBUILD_HOST : 0u);
    if (!on) hipLaunchKernelGGL(k_set_word, dim3(1), dim3(1), 0, st, a.check, 0u);
}

// ------------------------------------------------------------------------------------------
// When must the neighbour structure be rebuilt?  The lists hold every pair closer than L = 2H + skin at the rebuild
// positions pos_ref; the sort cells are L wide.  Let u_i = pos_i - pos_ref_i.  A pair that is NOT listed
// (|r_ref| >= L) can only come inside the support 2H if |u_i - u_k| > |r_ref| - 2H.  Hence the lists are valid while
//   (0) every |u_i| <= skin/2                                        (then |u_i - u_k| <= skin for every pair), or
//   (1) |u_i - u_k| <= skin for every pair whose sort cells are at most two cells apart (|r_ref| >= L otherwise
//       irrelevant), AND |u_i| <= H + skin for everybody (pairs three or more cells apart have |r_ref| >= 2L and would
//       need |u_i - u_k| > 2H + 2 skin).
// (0) is absolute: one fast jet makes everybody rebuild.  (1) is Galilean-invariant where it matters: a coherent jet
// keeps its lists.  The drifting kernel evaluates (0) and the cap of (1) per particle and leaves the bounding box of
// u over each group of BOXG consecutive sorted particles (~2 cells) in wbox; only if (0) fails, k_check evaluates the
// pair part of (1) on the boxes of the groups k_build_list found to be within two cells of each other (conservative).
// The direct walks (fallback tiles, variant 1, metaballs) look at 5x5 sort cells, which is exact under the cap alone.
// Slab mode: a ghost's box is not known before the halo exchange, so groups that can meet ghosts (their ranges touch
// the ghost slots) must satisfy (0); a pair across an interface has both partners in such groups, each held to
// skin/2 by its owner.  Everybody else is compared as on a single GPU (all the boxes involved are owned ones).
DEV float group_min(float v) {      // over the BOXG lanes of this lane's group
#pragma unroll
    for (int d = BOXG / 2; d >= 1; d >>= 1) v = fminf(v, __shfl_xor(v, d, 64));
    return v;
}
DEV float group_max(float v) {
#pragma unroll
    for (int d = BOXG / 2; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
    return v;
}
// The displacement box of a whole wave (BOXG == 64): x0 = min, y0 = min, x1 = max, y1 = max over the 64 lanes, valid in EVERY lane.
// __shfl_xor is ds_bpermute_b32 on this target: the four butterflies above were 24 trips through the LDS pipe plus ~80 vector
// instructions at the very end of every wave of the force pass (round 5: ~9 % of its instructions).  Here: four DPP steps inside
// the rows of 16 lanes (quad_perm xor 1, xor 2, row_half_mirror, row_mirror: min and max are idempotent, so mirrors do as well as
// butterflies) — v_min / v_max with a DPP operand, one instruction per step and value — then the four rows through v_readlane.
// (A DPP operand must not be read within two wait states of the instruction that wrote it and nothing tracks that inside an asm
// block: the four values are interleaved, so every result is three instructions old when it is read again; one s_nop in front.)
DEV void wave_box64(float &x0, float &y0, float &x1, float &y1) {
#define SPH_DPP_STEP(CTRL)                                                                   \
    "v_min_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                        \
    "v_min_f32_dpp %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                        \
    "v_max_f32_dpp %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                        \
    "v_max_f32_dpp %3, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf\n\t"
    asm volatile("s_nop 1\n\t" SPH_DPP_STEP("quad_perm:[1,0,3,2]") SPH_DPP_STEP("quad_perm:[2,3,0,1]") SPH_DPP_STEP("row_half_mirror")
                 SPH_DPP_STEP("row_mirror") "s_nop 1"
                 : "+v"(x0), "+v"(y0), "+v"(x1), "+v"(y1));
#undef SPH_DPP_STEP
#define SPH_ROWS(v, OP)                                                                                                     \
    OP(OP(__uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 0)),                                 \
          __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 16))),                               \
       OP(__uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 32)),                                \
          __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 48))))
    x0 = SPH_ROWS(x0, fminf);
    y0 = SPH_ROWS(y0, fminf);
    x1 = SPH_ROWS(x1, fmaxf);
    y1 = SPH_ROWS(y1, fmaxf);
#undef SPH_ROWS
}
// all 64 lanes of the wave must call this (live = the lane holds a particle this rank integrates: in slab mode the
// boxes cover the owned particles only); group = the wave's box group (tile * 4 + wave of the tile: sph_list.inc).  A group
// without a live lane leaves an empty box (zero displacement): in slab mode k_check may look at it (see there).
DEV void drift_verdict(const Consts &c, float ux, float uy, bool live, int group, float4 *__restrict__ wbox,
                       uint32_t *__restrict__ check, uint32_t *__restrict__ rebuild, const float *__restrict__ dyn) {
    static_assert(BOXG == 64 || BOXG == 32 || BOXG == 16, "a box group is a wave or an aligned part of one");
    const float d2 = fmaf(ux, ux, uy * uy);
    // Criterion (0) relative to a displacement U that is the same for every particle of this launch: if everybody is within
    // skin/2 of U, no two particles have moved more than the skin relative to each other — whatever U is.  U = 0 is the
    // absolute form; the density pass of the step leaves the (predicted) displacement of three sampled particles' median
    // there, so a fluid that moves as a whole (a falling drop) does not raise the check word step after step.
    const float rx = ux - dyn[DYN_UREF_X], ry = uy - dyn[DYN_UREF_Y];
    const bool over = live && !(fmaf(rx, rx, ry * ry) <= dyn[DYN_LIM2]);      // true for NaN too
    const bool capped = live && !(d2 <= c.cap2);
    const float inf = __builtin_huge_valf();
    float x0 = live ? ux : inf, y0 = live ? uy : inf, x1 = live ? ux : -inf, y1 = live ? uy : -inf;
    if (BOXG == 64) {
        wave_box64(x0, y0, x1, y1);
    } else {
        x0 = group_min(x0); y0 = group_min(y0); x1 = group_max(x1); y1 = group_max(y1);
    }
    const unsigned long long any_over = __ballot(over), any_cap = __ballot(capped);
    const unsigned long long gmask = (BOXG == 64 ? ~0ull : ((1ull << (BOXG & 63)) - 1ull)) << (threadIdx.x & 63 & ~(BOXG - 1));
    const unsigned long long any_live = __ballot(live) & gmask;      // (the live lanes of THIS group)
    if ((threadIdx.x & (BOXG - 1)) == 0) {
        wbox[group] = any_live != 0ull ? make_float4(x0, y0, x1, y1) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (any_over != 0ull) *check = 1u;
        if (any_cap != 0ull) {
            atomicMax(rebuild, (uint32_t)REBUILD_CRITERION);
            atomicAdd(check + ((int)FLAG_WHY_REBUILD + 2 - (int)FLAG_CHECK), 1u);      // (the check word always lives in flags[])
        }
    }
}

// The update message of the NEXT step, written by the kernel that has just drifted the particles (round 5; rounds 1-4 packed it
// with a kernel of its own at the start of that step): particle i (sorted index) of this slab's two outermost owned columns on a
// side with a neighbour -> entry i - (first particle of those columns) of that side's send buffer.  `first` = one thread of the
// launch: the headers (count, the step the message is for).
DEV void halo_update_write(const Consts &c, const uint32_t *__restrict__ cs, const int i, const bool mine, const float2 p, const float2 v,
                           uint32_t *__restrict__ send_l, uint32_t *__restrict__ send_r, uint32_t *__restrict__ flags,
                           const uint32_t for_step, const bool first) {
#pragma unroll
    for (int side = 0; side < 2; side++) {
        if (side == 0 ? !c.has_left : !c.has_right) continue;
        const int col0 = side == 0 ? c.ghost : c.ghost + c.owned - 2;
        const int beg = (int)cs[col0 * c.rows], n = (int)cs[(col0 + 2) * c.rows] - beg;
        uint32_t *buf = side == 0 ? send_l : send_r;
        if (first) {
            buf[HALO_UPD_COUNT] = (uint32_t)min(n, c.halo_cap);
            buf[HALO_UPD_STEP] = for_step;
            if (n > c.halo_cap) atomicAdd(&flags[FLAG_CAPACITY], 1u);
        }
        const int t = i - beg;
        if (mine && t >= 0 && t < n && t < c.halo_cap) reinterpret_cast<float4 *>(buf + HALO_HDR)[t] = make_float4(p.x, p.y, v.x, v.y);
    }
}

// kick 1/2 + drift in place (:615-624), 48 B/particle: the stand-alone form (the first step after creation / upload —
// afterwards the force pass does this for the next step, sph_list.inc).  One thread per tile lane (lrec: the lane's
// particle), so that a wave is one box group here as in the force pass.
template <bool SLAB>
__global__ __launch_bounds__(BLK) void k_kick_drift(Consts c, float2 *__restrict__ pos, const float2 *__restrict__ pos_ref,
                                                    const float2 *__restrict__ acc, const float2 *__restrict__ velt,
                                                    float2 *__restrict__ vel, const uint32_t *__restrict__ cs,
                                                    const uint2 *__restrict__ lrec, float4 *__restrict__ wbox,
                                                    uint32_t *__restrict__ check, uint32_t *__restrict__ rebuild,
                                                    const uint32_t *__restrict__ dn, const float *__restrict__ dyn,
                                                    uint32_t *__restrict__ send_l, uint32_t *__restrict__ send_r) {
    // slab mode: only the OWNED range of the sorted arrays moves (cell_start of the last rebuild); the ghosts are
    // refreshed from their owners by the halo exchange of this step
    const int n = (int)dn[0];
    const int own_lo = SLAB ? (int)cs[c.ghost * c.rows] : 0;
    const int own_hi = SLAB ? (int)cs[(c.ghost + c.owned) * c.rows] : n;
    const int t = blockIdx.x * BLK + threadIdx.x;      // rank in the tile order
    if ((int)(blockIdx.x * BLK) >= n) return;          // whole workgroup beyond the live count
    const int i = t < n ? (int)lrec[t].x : n;
    const bool live = t < n && i >= own_lo && i < own_hi;
    float ux = 0.0f, uy = 0.0f;
    float2 v = make_float2(0.0f, 0.0f), p = v;
    if (live) {
        const float2 a = acc[i], r = pos_ref[i];
        v = velt[i];
        p = pos[i];
        v.x = fmaf(c.half_dt, a.x, v.x);   // u += 0.5*DT*du_dt   :616
        v.y = fmaf(c.half_dt, a.y, v.y);
        p.x = fmaf(c.dt, v.x, p.x);        // x += DT*u           :622
        p.y = fmaf(c.dt, v.y, p.y);
        pos[i] = p;
        vel[i] = v;
        ux = p.x - r.x;
        uy = p.y - r.y;
    }
    // slab mode: this step's update message for the neighbours (k_check, which counts the step, runs after this kernel)
    if (SLAB) halo_update_write(c, cs, i, live, p, v, send_l, send_r, check - (int)FLAG_CHECK, check[(int)FLAG_STEP - (int)FLAG_CHECK] + 1u, t == 0);
    drift_verdict(c, ux, uy, live, t / BOXG, wbox, check, rebuild, dyn);
}

void launch_kick_drift(hipStream_t st, const Consts &c, const Arrays &a, int cap, bool slab) {
    if (cap <= 0) return;
    dim3 g((cap + BLK - 1) / BLK), b(BLK);
    if (slab) hipLaunchKernelGGL(k_kick_drift<true>, g, b, 0, st, c, a.pos, a.pos_ref, a.acc, a.velt, a.vel, a.cell_start, a.lrec, a.wbox, a.check, a.rebuild, a.dn, a.dyn, a.send[0], a.send[1]);
    else hipLaunchKernelGGL(k_kick_drift<false>, g, b, 0, st, c, a.pos, a.pos_ref, a.acc, a.velt, a.vel, a.cell_start, a.lrec, a.wbox, a.check, a.rebuild, a.dn, a.dyn, nullptr, nullptr);
}

// ------------------------------------------------------------------------------------------
// P1: key + histogram into the staging arrays (rebuild kernel).  44 B/particle + 4 B slot.
// !SLAB (single GPU; init and upload in either mode): entries 0..dn[0]-1 of (pos, vsrc, id) as they are.
// SLAB (a slab's rebuild step): the OWNED range of the sorted arrays (cell_start of the previous rebuild) becomes
// staging entries 0..n_own-1, and every particle now inside a neighbour's reach (this slab's two outermost owned
// columns, plus the column it may have migrated into since the last rebuild) is appended to that neighbour's halo
// buffer as a full record: ONE exchange carries both the ghosts and the ownership migration.
// Halo buffer header (4 words): [0] particles in the UPDATE message, [1] the step it is for (k_check's count of steps) — written
// by the kernel that drifts the particles (the force pass of the step before; k_kick_drift after creation / uploads);
// [2] full RECORDS appended so far, [3] the step they are for — zeroed by k_check at the start of every step, filled by the pack
// of a rebuild step.  The payload behind the header is shared: update entries {x, y, u, v} or records {x, y, u, v, id}.
DEV void halo_append(uint32_t *__restrict__ buf, int cap, float2 p, float2 v, uint32_t id, uint32_t *__restrict__ flags) {
    const uint32_t k = atomicAdd(&buf[HALO_REC_COUNT], 1u);
    if (k < (uint32_t)cap) {
        uint32_t *r = buf + HALO_HDR + (size_t)k * HALO_REC;
        r[0] = __float_as_uint(p.x); r[1] = __float_as_uint(p.y);
        r[2] = __float_as_uint(v.x); r[3] = __float_as_uint(v.y);
        r[4] = id;
    } else {
        atomicAdd(&flags[FLAG_CAPACITY], 1u);
    }
}

// The skin of the next lists, chosen by ONE thread when a rebuild begins (the binning is its first phase).  The lists
// that are being replaced had the skin s0 (as a fraction of 2H) and lasted T steps, so the flow eats s0 / T of skin per
// step (the rebuild criterion is a bound on relative displacements, which grow linearly until something hits them).  A
// step with skin s costs about K (1 + s)^2 (the two list walkers: list length ~ area of the cut-off disc) plus R / T(s)
// (a rebuild every T(s) = T s / s0 steps): minimal where  s^2 (1 + s) = (R / 2K) (s0 / T).  On paper R / 2K is 2.9 on
// MI355X (a rebuild ~200 us, the part of density + force that scales with the lists ~35 us at s = 0, 2M particles; both
// scale with the particle count); measured (tools/skin_sweep_gpu.py, variants with 2 / 2.9 / 4 / 5.5 / 8): 4 and above are
// equally good and better than 2.9, hence ADAPT_RATIO = 5.  The next skin is half-way from the old one to that optimum, clamped to [skin_min,
// skin_max] (the grid is sized for skin_max).  Measured on the 2M-particle dam break with FIXED skins: best 0.15 - 0.19
// while most of the fluid is at rest, 0.30 in the developed flow.  A rebuild the host asked for says nothing about the
// flow and leaves the skin alone.  Everything that depends on the skin is written here: the list cut-off (build_tile) and
// the two thresholds of the rebuild criterion (drift_verdict, k_check).
#ifndef SPH_QUIET_STEPS
#define SPH_QUIET_STEPS 100
#endif
constexpr int QUIET_STEPS = SPH_QUIET_STEPS;      // lists that lived at least this long may be followed by lists with skin_min itself
DEV void adapt_skin(const Consts &c, const uint32_t word, uint32_t *__restrict__ flags, float *__restrict__ dyn) {
    // (the step counter past the caches: in rest mode k_rebuild itself has counted this step a moment ago, after reading the same line)
    const uint32_t step = __hip_atomic_load(&flags[FLAG_STEP], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), last = flags[FLAG_LAST_REBUILD];
    float skin = dyn[DYN_SKIN];
    if (c.skin_min < c.skin_max && word == (uint32_t)REBUILD_CRITERION && last != 0u) {
        const float T = fmaxf((float)(step - last), 1.0f);
        // (a skin of 0 — skin_min = 0, or a caller that zeroed the struct and set only `skin` — would stay 0 for ever: rhs = 0;
        // the rate at which the flow uses up a skin is then taken from a floor of 2 % of 2H)
        const float rhs = ADAPT_RATIO * (fmaxf(skin, 0.02f * c.two_h) / c.two_h) / T;
        float s = sqrtf(rhs);                                   // s^2 (1 + s) = rhs, Newton from s^2 = rhs
#pragma unroll
        for (int it = 0; it < 4; it++) s -= (s * s * (1.0f + s) - rhs) / (s * (2.0f + 3.0f * s) + 1e-12f);
        skin = 0.5f * (skin + s * c.two_h);
    }
    // The floor.  skin_min itself only for lists that follow lists which lived long (a fluid at rest or sloshing gently: there the
    // shorter lists pay every step and a rebuild is rare); lists that died sooner than QUIET_STEPS are followed by lists of at least
    // 1.5 skin_min — in an accelerating flow the next lists die sooner than the last, and the estimate above lags behind (measured in
    // round 4: the floor at 0.08 x 2H everywhere cost the collapse 3 %, at rest it gained 4 %).  Host-requested rebuilds (creation,
    // uploads) keep the skin they find: the first lists of a context have skin_min.
    float floor_ = c.skin_min;
    if (c.skin_min < c.skin_max && word == (uint32_t)REBUILD_CRITERION && last != 0u && step - last < (uint32_t)QUIET_STEPS)
        floor_ = fminf(1.5f * c.skin_min, c.skin_max);
    skin = fminf(fmaxf(skin, floor_), c.skin_max);
    flags[FLAG_LAST_REBUILD] = step;
    dyn[DYN_SKIN] = skin;
    dyn[DYN_CUT_LIST2] = (c.two_h + skin) * (c.two_h + skin);
    // (rounding of the squared distances stays on the safe side)
    dyn[DYN_LIM2] = (0.5f * skin) * (0.5f * skin) * (skin > 0.0f ? 0.999f : 1.0f);
    dyn[DYN_SKIN2] = skin * skin * (skin > 0.0f ? 0.999f : 1.0f);
}

template <bool SLAB>
DEV void key_hist_body(const Consts &c, const float2 *__restrict__ pos, const uint32_t *__restrict__ id,
                       const float2 *__restrict__ vsrc, const uint32_t *__restrict__ cs, float2 *__restrict__ velk,
                       float4 *__restrict__ pk, uint32_t *__restrict__ slot, uint32_t *__restrict__ count,
                       uint32_t *__restrict__ dirty, uint32_t *__restrict__ flags, uint32_t *__restrict__ dn,
                       uint32_t *__restrict__ send_l, uint32_t *__restrict__ send_r, int chunk,
                       uint32_t *__restrict__ block_sums, float *__restrict__ dyn, const uint32_t word) {
    const int t = chunk * BLK + threadIdx.x;      // chunk = 256 consecutive array slots
    if (t == 0) adapt_skin(c, word, flags, dyn);  // a rebuild begins: the skin of the lists it will build
    const int lane = threadIdx.x & 63;
    int src0 = 0, n;
    if (SLAB) {
        src0 = (int)cs[c.ghost * c.rows];
        n = (int)cs[(c.ghost + c.owned) * c.rows] - src0;
        if (t == 0) {
            dn[1] = (uint32_t)n;
            const uint32_t step = flags[FLAG_STEP];      // (counted by k_check at the start of this step)
            if (send_l) send_l[HALO_REC_STEP] = step;
            if (send_r) send_r[HALO_REC_STEP] = step;
        }
    } else {
        n = (int)dn[0];
    }
    const bool active = t < n;
    uint32_t key = 0xffffffffu, pid = 0u;
    float2 p = make_float2(0.0f, 0.0f), v = make_float2(0.0f, 0.0f);
    bool oob = false, bad = false;
    int col = 0;
    if (active) {
        const int i = src0 + t;
        p = pos[i];
        v = vsrc[i];
        pid = id[i];
        int row;
        cell_of(c, p.x, p.y, row, col, oob, bad);
        key = (uint32_t)(col * c.rows + row);
    }
    // The array is in the last rebuild's cell order and a particle moves << one cell between rebuilds, so
    // neighbouring lanes mostly share their new cell.  One histogram atomic per run of equal
    // keys (run head adds the run length, members take consecutive slots) instead of one per
    // particle: ~7x fewer atomics and no same-address serialisation inside the wave.
    const uint32_t prev = __shfl_up(key, 1, 64);
    const bool head = active && (lane == 0 || prev != key);
    const unsigned long long hm = __ballot(head);
    const unsigned long long am = __ballot(active);
    {   // the scan's per-tile totals (2048 cells each), accumulated here so that the scan needs no reduction launch of
        // its own: one atomic per run of equal TILES in the wave (a wave's keys span one or two tiles; one atomic per
        // run of equal keys cost 400 us: ~300 to an address, each ~280 ns)
        const uint32_t tkey = active ? key / SCAN_TILE : 0xffffffffu;
        const uint32_t tprev = __shfl_up(tkey, 1, 64);
        const bool thead = active && (lane == 0 || tprev != tkey);
        const unsigned long long thm = __ballot(thead);
        if (thead) {
            const unsigned long long above = lane == 63 ? 0ull : (thm >> (lane + 1));
            int nxt = above ? lane + 1 + __builtin_ctzll(above) : 64;
            nxt = min(nxt, 64 - __builtin_clzll(am));
            // (SCAN_SPREAD counters per tile, chosen by wave: a tile inside the fluid receives ~170 of these adds)
            atomicAdd(&block_sums[tkey * SCAN_SPREAD + (((uint32_t)chunk * (BLK / 64) + (threadIdx.x >> 6)) & (SCAN_SPREAD - 1))],
                      (uint32_t)(nxt - lane));
        }
    }
    if (active) {
        const unsigned long long below = hm & ((2ull << lane) - 1ull);       // run heads at or below this lane
        const int start = 63 - __builtin_clzll(below);
        const unsigned long long above = (start == 63) ? 0ull : (hm >> (start + 1));
        int next = above ? start + __builtin_ffsll((long long)above) : 64;
        next = min(next, 64 - __builtin_clzll(am));                          // active lanes are a prefix of the wave
        uint32_t base = 0;
        if (lane == start) {
            base = atomicAdd(&count[key], (uint32_t)(next - start));
            dirty[key / SCAN_TILE] = 1u;      // the scan skips histogram tiles nobody touched (most of a dry box)
        }
        base = __shfl(base, start, 64);
        slot[t] = base + (uint32_t)(lane - start);
        pk[t] = make_float4(p.x, p.y, __uint_as_float(pid), __uint_as_float(key));
        velk[t] = v;
        if (bad) atomicAdd(&flags[FLAG_NAN], 1u);
        else if (oob) atomicAdd(&flags[FLAG_OOB], 1u);
        if (SLAB && !bad) {
            if (c.has_left && col < c.ghost + 2) halo_append(send_l, c.halo_cap, p, v, pid, flags);
            if (c.has_right && col >= c.ghost + c.owned - 2) halo_append(send_r, c.halo_cap, p, v, pid, flags);
        }
    }
}

// single GPU, init, upload: keys + histogram of entries 0..dn[0]-1 as they are
__global__ __launch_bounds__(BLK) void k_key_hist(Consts c, const float2 *__restrict__ pos, const uint32_t *__restrict__ id,
                                                  const float2 *__restrict__ vsrc, const uint32_t *__restrict__ cs,
                                                  float2 *__restrict__ velk, float4 *__restrict__ pk,
                                                  uint32_t *__restrict__ slot, uint32_t *__restrict__ count,
                                                  uint32_t *__restrict__ dirty, uint32_t *__restrict__ flags,
                                                  const uint32_t *__restrict__ rebuild, uint32_t *__restrict__ dn,
                                                  int nchunks, uint32_t *__restrict__ block_sums, float *__restrict__ dyn) {
    const uint32_t word = *rebuild;
    if (word == 0u) return;      // rebuild kernel
    // a small grid striding over the chunks: in most steps this launch returns at once, and what that costs grows with
    // the grid (1.5 us up to 2048 workgroups, 2.8 us at 8192: tools/ubench_launch)
    for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x)
        key_hist_body<false>(c, pos, id, vsrc, cs, velk, pk, slot, count, dirty, flags, dn, nullptr, nullptr, chunk, block_sums, dyn, word);
}

// the pair part of criterion (1): every box group against every group k_build_list listed for it, eight threads per group
// (one per range of groups, two idle: the loads of a group are a chain of dependent latencies when one thread does them all)
constexpr int CHECK_LANES = 8;
// (This kernel serves the steps that launch one kernel per phase: slab contexts, the profiled step, contexts that share their
// device.  The step of sph_step runs the same comparison as workgroups inside the launch of its density pass, and there two
// boxes that fail are not the last word: their particles are checked one by one — spec_check_job / spec_verify_job, sph_list.inc.)
// the failing neighbours a lane of the check remembers for the particle-by-particle verification (spec_check_job, sph_list.inc);
// measured with 4 / 8 / 12: 8 404 / 8 790 / 8 793 steps/s in the protocol's median window
#ifndef SPH_VERIFY_MAX
#define SPH_VERIFY_MAX 8
#endif
constexpr int VERIFY_MAX = SPH_VERIFY_MAX;
DEV bool check_group(const Consts &c, const float4 *__restrict__ wbox, const uint32_t *__restrict__ wnbr,
                     const int w, const int k, const int own_lo, const int own_hi, const int own_safe, const float *__restrict__ dyn,
                     const bool verify, uint32_t (&fail_h)[VERIFY_MAX], int &nfail);
// what k_check does, for workgroup `block` of `nblocks` (returns true when it had boxes to compare: the check word was set)
DEV bool check_body(const Consts &c, const float4 *__restrict__ wbox, const uint32_t *__restrict__ wnbr,
                    const uint32_t *__restrict__ cs, const uint32_t *__restrict__ check,
                    uint32_t *__restrict__ rebuild, uint32_t *__restrict__ flags,
                    const uint32_t *__restrict__ dn, uint32_t *__restrict__ send_l,
                    uint32_t *__restrict__ send_r, int nw, float2 *__restrict__ grav, float gx, float gy,
                    const float *__restrict__ dyn, const int block, const int nblocks) {
    // the first kernel of every step: it counts them (the skin controller measures how many steps a set of lists lasted)
    if (block == 0 && threadIdx.x == 0) atomicAdd(&flags[FLAG_STEP], 1u);
    // (slab mode: the gravity of this step rides along instead of taking a launch of its own)
    if (grav && block == 0 && threadIdx.x == 0) *grav = make_float2(gx, gy);
    // slab mode: this is the first kernel of a step; it also clears the record half of the send buffers' headers (count, step) for
    // the pack that follows the reduction of the rebuild word on a rebuild step
    if (send_l && block == 0 && threadIdx.x < 4) ((threadIdx.x & 2u) ? send_r : send_l)[HALO_REC_COUNT + (threadIdx.x & 1u)] = 0u;
    if (*check == 0u) return false;
    if (block == 0 && threadIdx.x == 0) atomicAdd(&flags[FLAG_NCHECK], 1u);
    const int n = (int)dn[0];
    // The ranks (tile order, see tile_rank) this rank integrates; single GPU: all of them.  A slab's owned columns begin
    // at a pair boundary (two ghost columns); they end at one when their number is even.  Otherwise the last owned column
    // shares its pair with the first ghost column, and the ranks of that pair are a mix: own_hi = the end of that pair
    // (every group that holds an owned particle lies below it), own_safe = its beginning (a group whose neighbourhood
    // stays below it meets owned particles only).
    int own_lo = 0, own_hi = n, own_safe = n;
    if (c.ghost) {
        own_lo = (int)cs[c.ghost * c.rows];
        own_hi = (int)cs_ext(cs, (c.ghost + c.owned + (c.owned & 1)) * c.rows, c.n_cells);
        own_safe = (int)cs[(c.ghost + c.owned - (c.owned & 1)) * c.rows];
    }
    // (a small grid striding over the groups: see k_key_hist)
    for (int t = block * BLK + threadIdx.x; t < nw * CHECK_LANES; t += nblocks * BLK) {
        uint32_t fail_h[VERIFY_MAX] = {};
        int nfail = 0;
        if (check_group(c, wbox, wnbr, t / CHECK_LANES, t % CHECK_LANES, own_lo, own_hi, own_safe, dyn, false, fail_h, nfail)) {
            atomicMax(rebuild, (uint32_t)REBUILD_CRITERION);      // (never lowers a host's request)
            atomicAdd(&flags[FLAG_WHY_REBUILD + 0], 1u);
        }
    }
    return true;
}
__global__ __launch_bounds__(BLK) void k_check(Consts c, const float4 *__restrict__ wbox, const uint32_t *__restrict__ wnbr,
                                               const uint32_t *__restrict__ cs, const uint32_t *__restrict__ check,
                                               uint32_t *__restrict__ rebuild, uint32_t *__restrict__ flags,
                                               const uint32_t *__restrict__ dn, uint32_t *__restrict__ send_l,
                                               uint32_t *__restrict__ send_r, int nw, float2 *__restrict__ grav, float gx, float gy,
                                               const float *__restrict__ dyn) {
    (void)check_body(c, wbox, wnbr, cs, check, rebuild, flags, dn, send_l, send_r, nw, grav, gx, gy, dyn, (int)blockIdx.x, (int)gridDim.x);
}
// The boxes of group w against those of the groups in its range k.  Returns true: rebuild (two boxes have moved more than the skin
// relative to each other and there is no verification, or more of them than a lane remembers); false: fine, or up to VERIFY_MAX
// failing neighbour groups in fail_h (verify: the caller has their particles checked one by one)
DEV bool check_group(const Consts &c, const float4 *__restrict__ wbox, const uint32_t *__restrict__ wnbr,
                     const int w, const int k, const int own_lo, const int own_hi, const int own_safe, const float *__restrict__ dyn,
                     const bool verify, uint32_t (&fail_h)[VERIFY_MAX], int &nfail) {
    const float skin2 = dyn[DYN_SKIN2];
    if (w * BOXG >= own_hi || (w + 1) * BOXG <= own_lo) return false;      // no owned particle in this group
    const float4 b = wbox[w];
    const uint32_t *nb = wnbr + (size_t)w * WNBR_WORDS;
    bool meets_ghosts = false;
    if (c.ghost) {
        meets_ghosts = w * BOXG < own_lo || (w + 1) * BOXG > own_safe;
#pragma unroll
        for (int j = 0; j < WNBR_WORDS / 2; j++) {
            const uint32_t first = nb[2 * j], last = nb[2 * j + 1];
            if (first <= last) meets_ghosts |= (int)(first * BOXG) < own_lo || (int)((last + 1u) * BOXG) > own_safe;
        }
    }
    bool bad = false;
    if (meets_ghosts) {
        const float mx = fmaxf(fabsf(b.x), fabsf(b.z)), my = fmaxf(fabsf(b.y), fabsf(b.w));
        bad = k == 0 && !(fmaf(mx, mx, my * my) <= dyn[DYN_LIM2]);
    } else if (k < WNBR_WORDS / 2) {
        const uint32_t first = nb[2 * k], last = nb[2 * k + 1];
        // (first > last: nobody in that range.  Four boxes per trip, their loads in flight together: one by one they are
        // a chain of memory latencies, and this kernel is nothing but latency)
        for (uint32_t o = first; o <= last && o != 0xffffffffu; o += 4u) {
            float4 q[4];
#pragma unroll
            for (int j = 0; j < 4; j++) q[j] = wbox[min(o + (uint32_t)j, last)];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float rx = fmaxf(b.z - q[j].x, q[j].z - b.x), ry = fmaxf(b.w - q[j].y, q[j].w - b.y);
                const bool f = o + (uint32_t)j <= last && !(fmaf(rx, rx, ry * ry) <= skin2);
                if (f && verify && nfail < VERIFY_MAX) {
#pragma unroll
                    for (int i = 0; i < VERIFY_MAX; i++)
                        if (i == nfail) fail_h[i] = o + (uint32_t)j;      // (a register array: no dynamic index)
                    nfail++;
                } else {
                    bad |= f;      // no verification, or a lane with more failing neighbours than it can remember
                }
            }
        }
    }
    return bad;
}
void launch_check(hipStream_t st, const Consts &c, const Arrays &a, int cap, const float *gravity) {
    if (cap <= 0) return;
    const int nw = (cap + BOXG - 1) / BOXG;
    hipLaunchKernelGGL(k_check, dim3(gated_grid((nw * CHECK_LANES + BLK - 1) / BLK)), dim3(BLK), 0, st, c, a.wbox, a.wnbr, a.cell_start, a.check,
                       a.rebuild, a.flags, a.dn, a.send[0], a.send[1], nw, gravity ? a.grav : nullptr, gravity ? gravity[0] : 0.0f,
                       gravity ? gravity[1] : 0.0f, a.dyn);
}

void launch_key_only(hipStream_t st, const Consts &c, const Arrays &a, int cap, const float2 *vsrc) {
    if (cap <= 0) return;
    const int nchunks = (cap + BLK - 1) / BLK;
    hipLaunchKernelGGL(k_key_hist, dim3(gated_grid(nchunks)), dim3(BLK), 0, st, c, a.pos, a.id, vsrc, a.cell_start, a.velk,
                       a.pk, a.slot, a.count, a.dirty, a.flags, a.rebuild, a.dn, nchunks, a.block_sums, a.dyn);
}

// ------------------------------------------------------------------------------------------
// slab mode, steps WITHOUT a rebuild: the local particle set and its order are unchanged, so a ghost only needs its
// owner's new position and velocity.  Both sides keep the cells of the interface columns in a canonical order (by
// particle id, k_canon), so the owner's two outermost owned columns and the neighbour's two ghost columns are the
// SAME sequence of particles: the update is a plain copy of a contiguous range, 16 bytes per particle.
// Buffer header: {count, kind, 0, 0}, kind 0 = full records (rebuild step), 1 = update.
// expect_step: the step the message must be for — flags[FLAG_STEP] once the step's first kernel has counted it (0: read it), or the
// number k_slab_head has derived for itself (its own block 0 counts the step while its other blocks run)
DEV void unpack_update_body(const Consts &c, int side, int t, float2 *__restrict__ pos, float2 *__restrict__ vel,
                            const uint32_t *__restrict__ cs, uint32_t *__restrict__ flags, const uint32_t *__restrict__ recv_l,
                            const uint32_t *__restrict__ recv_r, const uint32_t expect_step = 0u) {
    if (side == 0 ? !c.has_left : !c.has_right) return;
    const int col0 = side == 0 ? 0 : c.ghost + c.owned;
    const int beg = (int)cs[col0 * c.rows], n = (int)cs[(col0 + c.ghost) * c.rows] - beg;
    const uint32_t *buf = side == 0 ? recv_l : recv_r;
    // the neighbour must have sent an update of exactly my ghost range (same rebuild step, same canonical order), for THIS step
    if (t == 0 && (buf[HALO_UPD_COUNT] != (uint32_t)n || buf[HALO_UPD_STEP] != (expect_step ? expect_step : flags[FLAG_STEP]))) atomicAdd(&flags[FLAG_MISMATCH], 1u);
    if (t >= n || t >= (int)buf[HALO_UPD_COUNT]) return;
    const float4 q = reinterpret_cast<const float4 *>(buf + HALO_HDR)[t];
    pos[beg + t] = make_float2(q.x, q.y);
    vel[beg + t] = make_float2(q.z, q.w);
}

// What a slab sends on a REBUILD step (the reduced word is set): keys + histogram of the owned range with the full-record
// halo pack; the record half of the send headers was zeroed by k_check at the start of the step.  On any other step
// there is nothing to do: the update message was written by the kernel that drifted the particles (halo_update_write).
__global__ __launch_bounds__(BLK) void k_halo_out(Consts c, const float2 *__restrict__ pos, const uint32_t *__restrict__ id,
                                                  const float2 *__restrict__ vel, const uint32_t *__restrict__ cs,
                                                  float2 *__restrict__ velk, float4 *__restrict__ pk,
                                                  uint32_t *__restrict__ slot, uint32_t *__restrict__ count,
                                                  uint32_t *__restrict__ dirty, uint32_t *__restrict__ flags,
                                                  const uint32_t *__restrict__ rebuild, uint32_t *__restrict__ dn,
                                                  uint32_t *__restrict__ send_l, uint32_t *__restrict__ send_r,
                                                  int part_blocks, int halo_blocks, uint32_t *__restrict__ block_sums,
                                                  float *__restrict__ dyn) {
    // (a small grid striding over the work: on most steps only the two update packs run; see k_key_hist)
    const uint32_t word = *rebuild;
    if (word != 0u) {
        for (int vb = (int)blockIdx.x; vb < part_blocks; vb += (int)gridDim.x)
            key_hist_body<true>(c, pos, id, vel, cs, velk, pk, slot, count, dirty, flags, dn, send_l, send_r, vb, block_sums, dyn, word);
    }
    // (any other step: the update message is in the send buffers already — halo_update_write, by the kernel that drifted)
    (void)halo_blocks;
}

void launch_halo_out(hipStream_t st, const Consts &c, const Arrays &a, int cap) {
    if (cap <= 0) return;
    const int part_blocks = (cap + BLK - 1) / BLK, halo_blocks = (c.halo_cap + BLK - 1) / BLK;
    const int grid = part_blocks > 2 * halo_blocks ? part_blocks : 2 * halo_blocks;
    hipLaunchKernelGGL(k_halo_out, dim3(gated_grid(grid)), dim3(BLK), 0, st, c, a.pos, a.id, a.vel, a.cell_start, a.velk, a.pk, a.slot,
                       a.count, a.dirty, a.flags, a.rebuild, a.dn, a.send[0], a.send[1], part_blocks, halo_blocks, a.block_sums, a.dyn);
}

// slab mode, rebuild step, after the scatter: put every cell of the interface columns (two ghost + two owned columns
// on each side) into ascending particle-id order.  Both neighbours hold the same particles in those cells, so after
// this their sequences agree and updates can be exchanged as contiguous ranges.  One thread per cell, selection sort
// in place (a cell holds ~7 particles).
DEV void canon_cell(const Consts &c, const int t, float2 *__restrict__ pos, float2 *__restrict__ pos_ref, float2 *__restrict__ vel,
                    uint32_t *__restrict__ id, const uint32_t *__restrict__ cs) {
    const int span = (c.ghost + 2) * c.rows;          // cells per side
    if (t >= 2 * span) return;
    const bool right = t >= span;
    if (right ? !c.has_right : !c.has_left) return;
    const int cell = right ? (c.ghost + c.owned - 2) * c.rows + (t - span) : t;
    const uint32_t beg = cs[cell], end = cs[cell + 1];
    for (uint32_t a = beg; a + 1 < end; a++) {
        uint32_t m = a, mid = id[a];
        for (uint32_t b = a + 1; b < end; b++) {
            const uint32_t v = id[b];
            if (v < mid) { mid = v; m = b; }
        }
        if (m != a) {
            const float2 p = pos[a], r = pos_ref[a], v = vel[a];
            const uint32_t ia = id[a];
            pos[a] = pos[m]; pos_ref[a] = pos_ref[m]; vel[a] = vel[m]; id[a] = mid;
            pos[m] = p; pos_ref[m] = r; vel[m] = v; id[m] = ia;
        }
    }
}
__global__ __launch_bounds__(BLK) void k_canon(Consts c, float2 *__restrict__ pos, float2 *__restrict__ pos_ref,
                                               float2 *__restrict__ vel, uint32_t *__restrict__ id,
                                               const uint32_t *__restrict__ cs, const uint32_t *__restrict__ rebuild) {
    if (*rebuild == 0u) return;
    canon_cell(c, blockIdx.x * BLK + threadIdx.x, pos, pos_ref, vel, id, cs);
}
void launch_canon(hipStream_t st, const Consts &c, const Arrays &a) {
    if (!(c.has_left || c.has_right)) return;
    const int work = 2 * (c.ghost + 2) * c.rows;
    hipLaunchKernelGGL(k_canon, dim3((work + BLK - 1) / BLK), dim3(BLK), 0, st, c, a.pos, a.pos_ref, a.vel, a.id, a.cell_start,
                       a.rebuild);
}

// slab mode: the records received from the two neighbours join the staging arrays behind the owned particles
DEV void ingest_body(const Consts &c, const uint32_t *__restrict__ recv_l, const uint32_t *__restrict__ recv_r,
                     float2 *__restrict__ velk, float4 *__restrict__ pk, uint32_t *__restrict__ slot,
                     uint32_t *__restrict__ count, uint32_t *__restrict__ dirty, uint32_t *__restrict__ flags,
                     uint32_t *__restrict__ dn, int stage_cap, uint32_t *__restrict__ block_sums, const int vblock) {
    const int n_own = (int)dn[1];
    const int nl = c.has_left ? min((int)recv_l[HALO_REC_COUNT], c.halo_cap) : 0;
    const int nr = c.has_right ? min((int)recv_r[HALO_REC_COUNT], c.halo_cap) : 0;
    const int t = vblock * BLK + threadIdx.x;
    if (t == 0) {
        int total = n_own + nl + nr;
        if (total > stage_cap) { atomicAdd(&flags[FLAG_CAPACITY], 1u); total = stage_cap; }
        if ((c.has_left && (int)recv_l[HALO_REC_COUNT] > c.halo_cap) || (c.has_right && (int)recv_r[HALO_REC_COUNT] > c.halo_cap))
            atomicAdd(&flags[FLAG_CAPACITY], 1u);
        dn[0] = (uint32_t)total;
        // (records of THIS step: a neighbour that did not rebuild in it has left the record half of its header zeroed)
        const uint32_t step = flags[FLAG_STEP];
        if ((c.has_left && recv_l[HALO_REC_STEP] != step) || (c.has_right && recv_r[HALO_REC_STEP] != step)) atomicAdd(&flags[FLAG_MISMATCH], 1u);
    }
    const bool mine = t < nl + nr && n_own + t < stage_cap;
    {   // the scan's per-tile totals: the records of a wave fall into one or two scan tiles -> one atomic per tile and wave
        uint32_t tkey = 0xffffffffu;
        if (mine) {
            const uint32_t *rr = (t < nl) ? recv_l + HALO_HDR + (size_t)t * HALO_REC : recv_r + HALO_HDR + (size_t)(t - nl) * HALO_REC;
            int row, col;
            bool oob, bad;
            cell_of(c, __uint_as_float(rr[0]), __uint_as_float(rr[1]), row, col, oob, bad);
            tkey = (uint32_t)(col * c.rows + row) / SCAN_TILE;
        }
        unsigned long long todo = __ballot(mine);
        while (todo) {
            const int lead = __builtin_ctzll(todo);
            const uint32_t tk = (uint32_t)__shfl((int)tkey, lead, 64);
            const unsigned long long same = __ballot(mine && tkey == tk) & todo;
            if ((int)(threadIdx.x & 63) == lead)
                atomicAdd(&block_sums[tk * SCAN_SPREAD + (((uint32_t)vblock * (BLK / 64) + (threadIdx.x >> 6)) & (SCAN_SPREAD - 1))],
                          (uint32_t)__builtin_popcountll(same));
            todo &= ~same;
        }
    }
    if (!mine) return;
    const int dst = n_own + t;
    const uint32_t *r = (t < nl) ? recv_l + HALO_HDR + (size_t)t * HALO_REC : recv_r + HALO_HDR + (size_t)(t - nl) * HALO_REC;
    const float2 p = make_float2(__uint_as_float(r[0]), __uint_as_float(r[1]));
    int row, col;
    bool oob, bad;
    cell_of(c, p.x, p.y, row, col, oob, bad);
    const uint32_t key = (uint32_t)(col * c.rows + row);
    pk[dst] = make_float4(p.x, p.y, __uint_as_float(r[4]), __uint_as_float(key));
    velk[dst] = make_float2(__uint_as_float(r[2]), __uint_as_float(r[3]));
    slot[dst] = atomicAdd(&count[key], 1u);
    dirty[key / SCAN_TILE] = 1u;
    if (bad) atomicAdd(&flags[FLAG_NAN], 1u);
    else if (oob) atomicAdd(&flags[FLAG_OOB], 1u);
}

// What a slab does with what it received, one launch: on a rebuild step the records join the staging arrays behind
// the owned particles (then dn[0] = owned + received); otherwise the updates overwrite the ghost ranges in place.
__global__ __launch_bounds__(BLK) void k_halo_in(Consts c, const uint32_t *__restrict__ recv_l,
                                                 const uint32_t *__restrict__ recv_r, float2 *__restrict__ velk,
                                                 float4 *__restrict__ pk, uint32_t *__restrict__ slot,
                                                 uint32_t *__restrict__ count, uint32_t *__restrict__ dirty,
                                                 uint32_t *__restrict__ flags, const uint32_t *__restrict__ rebuild,
                                                 uint32_t *__restrict__ dn, int stage_cap, float2 *__restrict__ pos,
                                                 float2 *__restrict__ vel, const uint32_t *__restrict__ cs, int halo_blocks,
                                                 uint32_t *__restrict__ block_sums, uint32_t *__restrict__ vq) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        flags[FLAG_LATCH] = *rebuild;      // for the final density pass (DENS_REST)
        flags[FLAG_CHECK_DONE] = 0u;       // (the criterion's blocks of this step's head kernel: k_slab_head)
        flags[FLAG_VERIFY_DONE] = 0u;
        if (vq) vq[0] = 0u;
    }
    if (*rebuild != 0u) {
        ingest_body(c, recv_l, recv_r, velk, pk, slot, count, dirty, flags, dn, stage_cap, block_sums, (int)blockIdx.x);
    } else {
        const int side = (int)blockIdx.x / halo_blocks;
        unpack_update_body(c, side, ((int)blockIdx.x - side * halo_blocks) * BLK + (int)threadIdx.x, pos, vel, cs, flags, recv_l, recv_r);
    }
}

void launch_halo_in(hipStream_t st, const Consts &c, const Arrays &a, int stage_cap) {
    const int halo_blocks = (c.halo_cap + BLK - 1) / BLK > 0 ? (c.halo_cap + BLK - 1) / BLK : 1;
    hipLaunchKernelGGL(k_halo_in, dim3(2 * halo_blocks), dim3(BLK), 0, st, c, a.recv[0], a.recv[1], a.velk, a.pk, a.slot,
                       a.count, a.dirty, a.flags, a.rebuild, a.dn, stage_cap, a.pos, a.vel, a.cell_start, halo_blocks, a.block_sums, a.vq);
}

// ------------------------------------------------------------------------------------------
// P2/P3: exclusive scan over n_cells+1 histogram entries (padded to SCAN_TILE with zeros).
DEV uint32_t wave_incl_scan(uint32_t v) {
    int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

DEV uint32_t block_sum_256(uint32_t v, uint32_t *lds4) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) lds4[w] = v;
    __syncthreads();
    uint32_t t = lds4[0] + lds4[1] + lds4[2] + lds4[3];
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_reduce(const uint32_t *__restrict__ count,
                                                            const uint32_t *__restrict__ dirty,
                                                            uint32_t *__restrict__ block_sums,
                                                            const uint32_t *__restrict__ rebuild) {
    __shared__ uint32_t red[4];
    if (*rebuild == 0u) return;      // rebuild kernel
    if (dirty[blockIdx.x] == 0u) {      // untouched since it was last zeroed: all counts are 0
        if (threadIdx.x == 0) block_sums[blockIdx.x * SCAN_SPREAD] = 0u;
        return;
    }
    const uint4 *src = reinterpret_cast<const uint4 *>(count + (size_t)blockIdx.x * SCAN_TILE) + threadIdx.x * 2;
    uint4 a = src[0], b = src[1];
    uint32_t s = a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) block_sums[blockIdx.x * SCAN_SPREAD] = s;      // (the tile's other counters are zero)
}

// one tile (SCAN_TILE entries) of the exclusive scan, by a workgroup of SCAN_BLOCK threads
DEV void scan_apply_tile(const int tile, uint32_t *__restrict__ count, const uint32_t *__restrict__ block_sums,
                         uint32_t *__restrict__ dirty, uint32_t *__restrict__ cell_start, int n_items) {
    __shared__ uint32_t red[4];
    __shared__ uint32_t wave_tot[4];
    // offset of this tile = sum of the tiles before it (<= a few thousand L2-resident words)
    uint32_t off = 0;
    for (int k = threadIdx.x; k < tile * SCAN_SPREAD; k += SCAN_BLOCK) off += block_sums[k];
    off = block_sum_256(off, red);

    size_t base = (size_t)tile * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
    if (dirty[tile] == 0u) {      // empty tile: every cell starts at the running offset; nothing to read or zero
        if (base + SCAN_ITEMS <= (size_t)n_items) {
            uint4 *dst = reinterpret_cast<uint4 *>(cell_start + base);
            dst[0] = make_uint4(off, off, off, off);
            dst[1] = make_uint4(off, off, off, off);
        } else {
#pragma unroll
            for (int k = 0; k < SCAN_ITEMS; k++)
                if (base + k < (size_t)n_items) cell_start[base + k] = off;
        }
        return;
    }
    __syncthreads();                    // every thread has read the flag before it is cleared
    if (threadIdx.x == 0) dirty[tile] = 0u;
    uint4 *src = reinterpret_cast<uint4 *>(count + base);
    uint4 a = src[0], b = src[1];
    uint32_t v[SCAN_ITEMS] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint32_t tsum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) tsum += v[k];
    uint32_t incl = wave_incl_scan(tsum);
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) wave_tot[w] = incl;
    __syncthreads();
    uint32_t woff = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) woff += (k < w) ? wave_tot[k] : 0u;
    uint32_t run = off + woff + incl - tsum;
    // the histogram is consumed: leave it zeroed for the next sort
    src[0] = make_uint4(0, 0, 0, 0);
    src[1] = make_uint4(0, 0, 0, 0);
    uint32_t o[SCAN_ITEMS];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) { o[k] = run; run += v[k]; }
    if (base + SCAN_ITEMS <= (size_t)n_items) {
        uint4 *dst = reinterpret_cast<uint4 *>(cell_start + base);
        dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
        dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
    } else {
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; k++)
            if (base + k < (size_t)n_items) cell_start[base + k] = o[k];
    }
}

__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_apply(uint32_t *__restrict__ count,
                                                           const uint32_t *__restrict__ block_sums,
                                                           uint32_t *__restrict__ dirty,
                                                           uint32_t *__restrict__ cell_start, int n_items,
                                                           const uint32_t *__restrict__ rebuild) {
    if (*rebuild == 0u) return;      // rebuild kernel
    scan_apply_tile((int)blockIdx.x, count, block_sums, dirty, cell_start, n_items);
}

__global__ void k_zero_words(uint32_t *w, int n) {
    const int i = blockIdx.x * BLK + threadIdx.x;
    if (i < n) w[i] = 0u;
}
// reduce = false (the fluid sort of a step): the per-tile totals were accumulated by the binning kernels
// (key_hist_body, ingest_body) and are zeroed again by k_reorder: ONE launch.  reduce = true (wall bins, at init or after
// sph_update_boundary): the totals are computed here and zeroed afterwards.
void launch_scan(hipStream_t st, const Consts &c, uint32_t *count, uint32_t *dirty, uint32_t *cell_start,
                 uint32_t *block_sums, const uint32_t *rebuild, bool reduce) {
    int n_items = c.n_cells + 1;
    int tiles = (n_items + SCAN_TILE - 1) / SCAN_TILE;
    if (reduce) hipLaunchKernelGGL(k_scan_reduce, dim3(tiles), dim3(SCAN_BLOCK), 0, st, count, dirty, block_sums, rebuild);
    hipLaunchKernelGGL(k_scan_apply, dim3(tiles), dim3(SCAN_BLOCK), 0, st, count, block_sums, dirty, cell_start, n_items, rebuild);
    if (reduce) hipLaunchKernelGGL(k_zero_words, dim3((tiles * SCAN_SPREAD + BLK - 1) / BLK), dim3(BLK), 0, st, block_sums, tiles * SCAN_SPREAD);
}

// ------------------------------------------------------------------------------------------
// P4: scatter to cell order (rebuild kernel).  The sorted positions are the reference positions of the new lists.
// Deterministic order (optional): the members of every cell in slot order, so that the scatter can rank a particle by
// its id among them.  The slots are handed out in the order the binning atomics arrive, which differs from run to run.
DEV void cell_ids_body(const float4 *__restrict__ pk, const uint32_t *__restrict__ slot, const uint32_t *__restrict__ cell_start,
                       uint32_t *__restrict__ cell_ids, const int n) {
    for (int i = blockIdx.x * BLK + threadIdx.x; i < n; i += gridDim.x * BLK) {
        const float4 q = pk[i];
        cell_ids[cell_start[__float_as_uint(q.w)] + slot[i]] = __float_as_uint(q.z);
    }
}
__global__ __launch_bounds__(BLK) void k_cell_ids(const float4 *__restrict__ pk, const uint32_t *__restrict__ slot,
                                                  const uint32_t *__restrict__ cell_start, uint32_t *__restrict__ cell_ids,
                                                  const uint32_t *__restrict__ dn, const uint32_t *__restrict__ rebuild) {
    if (*rebuild == 0u) return;
    cell_ids_body(pk, slot, cell_start, cell_ids, (int)dn[0]);
}

// cell_ids != nullptr: deterministic order inside a cell (by particle id)
DEV void reorder_body(const float4 *__restrict__ pk, const float2 *__restrict__ velk, const uint32_t *__restrict__ slot,
                      const uint32_t *__restrict__ cell_start, float2 *__restrict__ pos, float2 *__restrict__ pos_ref,
                      float2 *__restrict__ vel, uint32_t *__restrict__ id, uint32_t *__restrict__ skey, const int n,
                      uint32_t *__restrict__ block_sums, const int scan_tiles, const uint32_t *__restrict__ cell_ids) {
    // the scan has consumed its per-tile totals: leave them zero for the binning kernels of the next sort
    for (int k = blockIdx.x * BLK + threadIdx.x; k < scan_tiles; k += gridDim.x * BLK) block_sums[k] = 0u;
    for (int i = blockIdx.x * BLK + threadIdx.x; i < n; i += gridDim.x * BLK) {      // small grid: see k_key_hist
        float4 q = pk[i];
        const uint32_t key = __float_as_uint(q.w);
        uint32_t dst;
        if (cell_ids) {
            const uint32_t beg = cell_start[key], end = cell_start[key + 1], mine = __float_as_uint(q.z);
            dst = beg;
            for (uint32_t j = beg; j < end; j++) dst += cell_ids[j] < mine ? 1u : 0u;
        } else {
            dst = cell_start[key] + slot[i];
        }
        pos[dst] = make_float2(q.x, q.y);
        pos_ref[dst] = make_float2(q.x, q.y);
        vel[dst] = velk[i];
        id[dst] = __float_as_uint(q.z);
        skey[dst] = key;        // sorted keys: tile records, and the sort cell of a particle between rebuilds
    }
}

// Where the tiles of the list kernels begin (tile order: tile_rank).  One thread per pair-row: the pair-row in which rank
// 256 T falls records, for tile T, {pair, row, index in column A, index in column B} of the tile's first particle (the
// particles of a pair-row count in the order cell A, cell B); the tile ends where tile T + 1 begins.  Also per pair: the
// first and the last row that hold particles.  Part of the scatter phase of a rebuild: it reads cell_start only.
DEV void tile_starts_body(const Consts &c, const uint32_t *__restrict__ cs, uint32_t *__restrict__ tstart, uint32_t *__restrict__ pext,
                          const int n) {
    const int np = n_pairs(c), total = np * c.rows;
    for (int t = blockIdx.x * BLK + threadIdx.x; t < total; t += gridDim.x * BLK) {
        const int P = t / c.rows, r = t - P * c.rows;
        const uint32_t a0 = cs_at(c, cs, 2 * P, r), a1 = cs_at(c, cs, 2 * P, r + 1);
        const uint32_t b0 = cs_at(c, cs, 2 * P + 1, r), b1 = cs_at(c, cs, 2 * P + 1, r + 1);
        if (a1 + b1 == a0 + b0) continue;                      // nobody in this pair-row
        const uint32_t base_a = cs_at(c, cs, 2 * P, 0), base_b = cs_at(c, cs, 2 * P + 1, 0), end = cs_at(c, cs, 2 * P + 2, 0);
        if (a0 == base_a && b0 == base_b) pext[2 * P] = (uint32_t)r;
        if (a1 == base_b && b1 == end) pext[2 * P + 1] = (uint32_t)r;
        const uint32_t m0 = a0 + b0 - base_b, m1 = a1 + b1 - base_b, ca = a1 - a0;      // ranks [m0, m1)
        if (m1 == (uint32_t)n)      // the last pair-row that holds anybody: the last tile ends behind this pair
            reinterpret_cast<uint4 *>(tstart)[(n + SPH_TILE_PARTICLES - 1) / SPH_TILE_PARTICLES] = make_uint4((uint32_t)P + 1u, 0u, (uint32_t)n, (uint32_t)n);
        for (uint32_t T = (m0 + (uint32_t)SPH_TILE_PARTICLES - 1u) / (uint32_t)SPH_TILE_PARTICLES; T * (uint32_t)SPH_TILE_PARTICLES < m1; T++) {
            const uint32_t o = T * (uint32_t)SPH_TILE_PARTICLES - m0;
            reinterpret_cast<uint4 *>(tstart)[T] = make_uint4((uint32_t)P, (uint32_t)r, o < ca ? a0 + o : a1, o < ca ? b0 : b0 + (o - ca));
        }
    }
}

__global__ __launch_bounds__(BLK) void k_reorder(Consts c, const float4 *__restrict__ pk, const float2 *__restrict__ velk,
                                                 const uint32_t *__restrict__ slot,
                                                 const uint32_t *__restrict__ cell_start, float2 *__restrict__ pos,
                                                 float2 *__restrict__ pos_ref, float2 *__restrict__ vel,
                                                 uint32_t *__restrict__ id, uint32_t *__restrict__ skey,
                                                 const uint32_t *__restrict__ dn, const uint32_t *__restrict__ rebuild,
                                                 uint32_t *__restrict__ block_sums, int scan_tiles,
                                                 const uint32_t *__restrict__ cell_ids, uint32_t *__restrict__ tstart,
                                                 uint32_t *__restrict__ pext) {
    if (*rebuild == 0u) return;
    reorder_body(pk, velk, slot, cell_start, pos, pos_ref, vel, id, skey, (int)dn[0], block_sums, scan_tiles, cell_ids);
    tile_starts_body(c, cell_start, tstart, pext, (int)dn[0]);
}

void launch_reorder(hipStream_t st, const Consts &c, const Arrays &a, int cap, bool deterministic) {
    if (cap <= 0) return;
    const int scan_tiles = (c.n_cells + 1 + SCAN_TILE - 1) / SCAN_TILE * SCAN_SPREAD;      // counters to zero
    uint32_t *cell_ids = deterministic ? a.nlist : nullptr;      // (the lists are rebuilt after the scatter: free until then)
    if (deterministic)
        hipLaunchKernelGGL(k_cell_ids, dim3(gated_grid((cap + BLK - 1) / BLK)), dim3(BLK), 0, st, a.pk, a.slot, a.cell_start, cell_ids, a.dn, a.rebuild);
    hipLaunchKernelGGL(k_reorder, dim3(gated_grid((cap + BLK - 1) / BLK)), dim3(BLK), 0, st, c, a.pk, a.velk, a.slot, a.cell_start, a.pos,
                       a.pos_ref, a.vel, a.id, a.skey, a.dn, a.rebuild, a.block_sums, scan_tiles, cell_ids, a.tstart, a.pext);
}

// ------------------------------------------------------------------------------------------
// Tait EOS, clamped at zero (:294-301): p = max(0, B((rho/rho0)^7 - 1)); also p/rho^2 for the force pass.
// (rho / rho0 as a multiplication by 1 / rho0 and p / rho^2 through v_rcp_f32 — one ulp each, 1e-7 of gates that are 1e-5 wide —
// instead of two IEEE divisions: ~25 of the density pass's ~450 vector instructions per wave)
// -DSPH_EOS_IEEE (`make variant`: the bisect of tools/rho_gate_chaos.py's statistic on the GPU): the two IEEE divisions back.
#ifdef SPH_EOS_IEEE
DEV float eos_p_over_rho2(float p, float rho) { return p / (rho * rho); }
#else
DEV float eos_p_over_rho2(float p, float rho) { return p * __builtin_amdgcn_rcpf(rho * rho); }
#endif
DEV void eos(const Consts &c, float rho, float &p, float &p_over_rho2) {
#ifdef SPH_EOS_IEEE
    float r = rho / c.rho0;
#else
    float r = rho * c.inv_rho0;
#endif
    float r2 = r * r, r4 = r2 * r2;
    float r7 = r4 * r2 * r;
    p = fmaxf(c.B * (r7 - 1.0f), 0.0f);
    p_over_rho2 = eos_p_over_rho2(p, rho);
}

// ------------------------------------------------------------------------------------------
// P5 (variant 1, "direct": A/B measurements and the tests' exact reference walk): one thread per particle, no lists.
// Fluid neighbours: the 5x5 block of SORT cells around the particle's sort cell, exact support test on the current
// positions — valid whenever nobody is further than H + skin from where it was sorted (the cap of the rebuild
// criterion, see drift_verdict).  Boundary neighbours: the 3x3 cells around the particle's CURRENT cell (walls do not
// move and are binned in the same grid).
template <bool EOS>
__global__ __launch_bounds__(BLK) void k_density_direct(Consts c, const float2 *__restrict__ pos,
                                                        const uint32_t *__restrict__ skey,
                                                        const uint32_t *__restrict__ cs, const float2 *__restrict__ bpos,
                                                        const float *__restrict__ bpsi, const uint32_t *__restrict__ bcs,
                                                        float2 *__restrict__ rp, float *__restrict__ prs,
                                                        uint32_t *__restrict__ rebuild, uint32_t *__restrict__ check,
                                                        const uint32_t *__restrict__ dn) {
    int i = blockIdx.x * BLK + threadIdx.x;
    if (rebuild && i == 0) { *rebuild = 0u; *check = 0u; }      // served by the kernels before this one
    if (i >= (int)dn[0]) return;
    float2 pi = pos[i];
    int row, col;
    cell_of_key(c, skey[i], row, col);
    const int r0 = max(row - 2, 0), r1 = min(row + 2, c.rows - 1);
    float sf = 0.0f, sb = 0.0f;
    for (int cc = max(col - 2, 0); cc <= min(col + 2, c.cols - 1); cc++) {
        const uint32_t beg = cs[cc * c.rows + r0], end = cs[cc * c.rows + r1 + 1];
        for (uint32_t j = beg; j < end; j++) {      // includes j == i: W(0) = nf is the self term of :274-275
            float2 pj = pos[j];
            float dx = pi.x - pj.x, dy = pi.y - pj.y;
            float d2 = fmaf(dx, dx, dy * dy);
            float w = w_shape(c, d2);
            sf += (d2 < c.cut2) ? w : 0.0f;
        }
    }
    bool oob, bad;
    cell_of(c, pi.x, pi.y, row, col, oob, bad);
    const int b0 = max(row - 1, 0), b1 = min(row + 1, c.rows - 1);
    for (int cc = max(col - 1, 0); cc <= min(col + 1, c.cols - 1); cc++) {
        const uint32_t bbeg = bcs[cc * c.rows + b0], bend = bcs[cc * c.rows + b1 + 1];
        for (uint32_t j = bbeg; j < bend; j++) {
            float2 pj = bpos[j];
            float dx = pi.x - pj.x, dy = pi.y - pj.y;
            float d2 = fmaf(dx, dx, dy * dy);
            float w = w_shape(c, d2) * bpsi[j];
            sb += (d2 < c.cut2) ? w : 0.0f;
        }
    }
    float rho = c.nf * fmaf(c.m_fluid, sf, sb);     // m W(0) + sum m W + sum psi W   :287
    if (EOS) {
        float p, pr2;
        eos(c, rho, p, pr2);
        rp[i] = make_float2(rho, pr2);
        prs[i] = p;
    } else {
        rp[i].x = rho;
    }
}

// EOS alone (stage entry point): from the stored rho, or (from_prs) only refresh p/rho^2 from stored rho and p
template <bool FROM_PRS>
__global__ __launch_bounds__(BLK) void k_eos(Consts c, float2 *__restrict__ rp, float *__restrict__ prs,
                                             const uint32_t *__restrict__ dn) {
    int i = blockIdx.x * BLK + threadIdx.x;
    if (i >= (int)dn[0]) return;
    float rho = rp[i].x;
    if (FROM_PRS) {
        rp[i].y = eos_p_over_rho2(prs[i], rho);      // (as eos())
    } else {
        float p, pr2;
        eos(c, rho, p, pr2);
        rp[i].y = pr2;
        prs[i] = p;
    }
}

void launch_eos(hipStream_t st, const Consts &c, const Arrays &a, int cap, bool from_prs) {
    if (cap <= 0) return;
    dim3 g((cap + BLK - 1) / BLK), b(BLK);
    if (from_prs) hipLaunchKernelGGL(k_eos<true>, g, b, 0, st, c, a.rp, a.prs, a.dn);
    else hipLaunchKernelGGL(k_eos<false>, g, b, 0, st, c, a.rp, a.prs, a.dn);
}

// ------------------------------------------------------------------------------------------
// pair term of calculate_accelerations (:317-337 / :346-365) on squared distance.
//   returns coef such that  -(m_j temp_ij grad_i W_ij) = grad_c * m_j * coef * (dx,dy)
//   (grad_i W_ij = -5 nf (1-q/2)^3 (dx,dy)/H^2 because q/d = 1/H, :56-59)
DEV float pair_coef(const Consts &c, float d2, float xv, float pr_sum, float rho_mean) {
    float d = __builtin_amdgcn_sqrtf(d2);
    float q = d * c.inv_h;
    float a = fmaf(-0.5f, q, 1.0f);
    float a2 = a * a;
    float a3 = a2 * a;
    float w = a3 * a * fmaf(2.0f, q, 1.0f);                  // W_ij / nf            :324
    float s = w * (c.nf * c.inv_w_k2h);                       // W_ij / W(0.2H)
    float s2 = s * s;
    float art = c.k1 * s2 * s2;                               // artificial pressure  :325
    float inv = __builtin_amdgcn_rcpf((d2 + c.eps_h2) * rho_mean);
    float visc = (xv < 0.0f) ? -c.visc_c * xv * inv : 0.0f;   // artificial viscosity :332-334
    return a3 * (pr_sum + art + visc);                        // temp_ij * (1-q/2)^3  :336
}

// P6 (variant 1, "direct")
template <bool KICK>
__global__ __launch_bounds__(BLK) void k_force_direct(Consts c, const float2 *__restrict__ pos,
                                                      const float2 *__restrict__ vel, const float2 *__restrict__ rp,
                                                      const uint32_t *__restrict__ skey,
                                                      const uint32_t *__restrict__ cs, const float2 *__restrict__ bpos,
                                                      const float *__restrict__ bpsi, const uint32_t *__restrict__ bcs,
                                                      const float2 *__restrict__ grav, float2 *__restrict__ acc,
                                                      float2 *__restrict__ velt, const uint32_t *__restrict__ dn,
                                                      const float2 *__restrict__ bvel) {
    int i = blockIdx.x * BLK + threadIdx.x;
    if (i >= (int)dn[0]) return;
    float2 pi = pos[i], vi = vel[i], rpi = rp[i];
    int row, col;
    cell_of_key(c, skey[i], row, col);
    const int r0 = max(row - 2, 0), r1 = min(row + 2, c.rows - 1);      // 5x5 sort cells, see k_density_direct
    float fx = 0.0f, fy = 0.0f, bx = 0.0f, by = 0.0f;
    for (int cc = max(col - 2, 0); cc <= min(col + 2, c.cols - 1); cc++) {
        const uint32_t beg = cs[cc * c.rows + r0], end = cs[cc * c.rows + r1 + 1];
        for (uint32_t j = beg; j < end; j++) {
            float2 pj = pos[j];
            float dx = pi.x - pj.x, dy = pi.y - pj.y;
            float d2 = fmaf(dx, dx, dy * dy);
            // j == i and coincident particles (d = 0, 0/0 in the reference :58-59) exert no force
            if (d2 < c.cut2 && d2 > 0.0f) {
                float2 vj = vel[j], rpj = rp[j];
                float xv = fmaf(dx, vi.x - vj.x, dy * (vi.y - vj.y));              // :330
                float cf = pair_coef(c, d2, xv, rpi.y + rpj.y, 0.5f * (rpi.x + rpj.x));
                fx = fmaf(cf, dx, fx);
                fy = fmaf(cf, dy, fy);
            }
        }
    }
    bool oob, bad;
    cell_of(c, pi.x, pi.y, row, col, oob, bad);                          // walls: 3x3 around the CURRENT cell
    const int b0 = max(row - 1, 0), b1 = min(row + 1, c.rows - 1);
    for (int cc = max(col - 1, 0); cc <= min(col + 1, c.cols - 1); cc++) {
        const uint32_t bbeg = bcs[cc * c.rows + b0], bend = bcs[cc * c.rows + b1 + 1];
        for (uint32_t j = bbeg; j < bend; j++) {
            float2 pj = bpos[j];
            float dx = pi.x - pj.x, dy = pi.y - pj.y;
            float d2 = fmaf(dx, dx, dy * dy);
            if (d2 < c.cut2 && d2 > 0.0f) {
                float2 vb = bvel[j];                                               // the wall particle's stored u, v :357
                float xv = fmaf(dx, vi.x - vb.x, dy * (vi.y - vb.y));
                float cf = bpsi[j] * pair_coef(c, d2, xv, rpi.y, rpi.x);           // :350, :362
                bx = fmaf(cf, dx, bx);
                by = fmaf(cf, dy, by);
            }
        }
    }
    float2 g = *grav;
    float ax = fmaf(c.grad_c, fmaf(c.m_fluid, fx, bx), g.x);                       // :370
    float ay = fmaf(c.grad_c, fmaf(c.m_fluid, fy, by), g.y);                       // :371
    acc[i] = make_float2(ax, ay);
    if (KICK) velt[i] = make_float2(fmaf(c.half_dt, ax, vi.x), fmaf(c.half_dt, ay, vi.y));   // :638-639
}

// ------------------------------------------------------------------------------------------
// Peer transport of the slab step (include/sph.h, "peer-mapped transport"): the ranks of one node map each other's receive
// buffers and flag words (hipIpc handles, opened by the HOST) and the per-step traffic is plain stores over xGMI plus
// flag words — three small kernels on the context's stream instead of an all-reduce and a send / receive of a collective
// library.  Everything that crosses between ranks is written with system-scope release stores behind a system-scope
// fence (the data has left this device's caches) and read with system-scope acquire loads in a kernel of its own (the
// kernels that follow start with freshly invalidated caches).  All waits are bounded (FLAG_BAR_TIMEOUT -> SPH_E_STATE).
constexpr uint32_t PEER_SPINS = 1u << 23;      // x ~0.2 us per poll: a few seconds
DEV bool peer_wait(const uint32_t *__restrict__ word, const uint32_t tag, const uint32_t shift, uint32_t *__restrict__ flags, uint32_t &value,
                   const uint32_t site = 0u) {
    uint32_t spins = 0u;
    for (;;) {
        value = __hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        // an arrival flag (shift 0) only grows — 2 x step for the update, 2 x step + 1 for the records — and the neighbour may be a
        // launch AHEAD: the push blocks of its next head kernel do not wait for that launch's exchange block, so a rank whose launch was
        // held up (ranks sharing a device are time-sliced) finds 2 x (step + 1) where it waits for 2 x step.  What it then reads is
        // still there: receive buffers alternate with the step's parity, and the neighbour cannot begin step + 2 without this rank's
        // word for step + 1 (k_slab_head).  The rebuild-word slots (shift 2) carry a payload under the step: equality, safe for the same reason.
#ifdef SPH_PEER_WAIT_EQUAL      // (what rounds 3-5 shipped, for tools/lean_peer_probe.sh: the stalled rank gives up)
        if ((value >> shift) == tag) return true;
#else
        if (shift == 0u ? (int)(value - tag) >= 0 : (value >> shift) == tag) return true;
#endif
        __builtin_amdgcn_s_sleep(4);
        if (++spins > PEER_SPINS) {
            // (bit 1: a peer never arrived; bit 0 is the grid barrier's.)  The first to give up says where and over what
            if ((atomicOr(&flags[FLAG_BAR_TIMEOUT], 2u) & 2u) == 0u) {
                flags[FLAG_PEER_DIAG + 0] = site;
                flags[FLAG_PEER_DIAG + 1] = tag << shift;
                flags[FLAG_PEER_DIAG + 2] = value;
                flags[FLAG_PEER_DIAG + 3] = flags[FLAG_STEP];
            }
            return false;
        }
    }
}
struct PeerSlots { uint32_t *of_rank[SPH_PEER_MAX_RANKS]; };
// MAX over the ranks of the rebuild word.  Every rank owns slots[2][SPH_PEER_MAX_RANKS] (parity of the step x sender):
// thread q stores (tag << 1 | my word) into MY slot of rank q's array, then waits for rank q's entry in this rank's array.
__global__ __launch_bounds__(64) void k_peer_reduce(PeerSlots peers, const uint32_t *__restrict__ mine, uint32_t *__restrict__ rebuild,
                                                    uint32_t *__restrict__ flags, int me, int nranks, uint32_t tag) {
    const int q = (int)threadIdx.x;
    const uint32_t par = (tag & 1u) * (uint32_t)SPH_PEER_MAX_RANKS;
    const uint32_t w = *rebuild;
    uint32_t got = w;
    if (q < nranks && q != me) {
        __hip_atomic_store(peers.of_rank[q] + par + (uint32_t)me, (tag << 2) | (w & 3u), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        uint32_t v = 0u;
        if (peer_wait(mine + par + (uint32_t)q, tag, 2u, flags, v, (1u << 8) | (uint32_t)q)) got = v & 3u;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) got = max(got, (uint32_t)__shfl_xor((int)got, d, 64));
    if (q == 0) *rebuild = got;
}
// The two send buffers of this step into the neighbours' receive buffers (as many bytes as the header says), then —
// by the last workgroup to finish, behind a system-scope fence in every workgroup — the neighbours' arrival flags.
__global__ __launch_bounds__(BLK) void k_peer_push(const uint32_t *__restrict__ send_l, const uint32_t *__restrict__ send_r,
                                                   uint32_t *__restrict__ remote_l, uint32_t *__restrict__ remote_r,
                                                   uint32_t *__restrict__ flag_l, uint32_t *__restrict__ flag_r,
                                                   uint32_t *__restrict__ done, int halo_cap, uint32_t tag) {
    for (int side = 0; side < 2; side++) {
        const uint32_t *src = side == 0 ? send_l : send_r;
        uint32_t *dst = side == 0 ? remote_l : remote_r;
        if (!dst) continue;
        const bool records = src[HALO_REC_STEP] != 0u;      // (a rebuild step's pack has stamped it; zeroed by k_check otherwise)
        const uint32_t count = min(src[records ? HALO_REC_COUNT : HALO_UPD_COUNT], (uint32_t)halo_cap);
        const uint32_t words = (uint32_t)HALO_HDR + count * (records ? (uint32_t)HALO_REC : 4u), quads = (words + 3u) / 4u;
        const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
        uint4 *d4 = reinterpret_cast<uint4 *>(dst);
        for (uint32_t k = blockIdx.x * BLK + threadIdx.x; k < quads; k += gridDim.x * BLK) d4[k] = s4[k];
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t arrived = atomicAdd(done, 1u);      // (grows by gridDim.x per launch: never reset)
        if ((arrived + 1u) % gridDim.x == 0u) {
            __threadfence_system();
            if (flag_l) __hip_atomic_store(flag_l, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            if (flag_r) __hip_atomic_store(flag_r, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
__global__ __launch_bounds__(64) void k_peer_wait(const uint32_t *__restrict__ flag_l, const uint32_t *__restrict__ flag_r,
                                                  uint32_t *__restrict__ flags, uint32_t tag) {
    uint32_t v;
    if (threadIdx.x == 0 && flag_l) (void)peer_wait(flag_l, tag, 0u, flags, v, 2u << 8);
    if (threadIdx.x == 1 && flag_r) (void)peer_wait(flag_r, tag, 0u, flags, v, (2u << 8) | 1u);
}
void launch_peer_reduce(hipStream_t st, const Arrays &a, void *const *slots_of_rank, const void *mine, int me, int nranks, uint32_t tag) {
    PeerSlots ps;
    for (int q = 0; q < SPH_PEER_MAX_RANKS; q++) ps.of_rank[q] = q < nranks ? static_cast<uint32_t *>(slots_of_rank[q]) : nullptr;
    hipLaunchKernelGGL(k_peer_reduce, dim3(1), dim3(64), 0, st, ps, static_cast<const uint32_t *>(mine), a.rebuild, a.flags, me, nranks, tag);
}
void launch_peer_push(hipStream_t st, const Consts &c, const Arrays &a, void *remote_l, void *flag_l, void *remote_r, void *flag_r, uint32_t tag) {
    constexpr int PUSH_WGS = 32;
    hipLaunchKernelGGL(k_peer_push, dim3(PUSH_WGS), dim3(BLK), 0, st, a.send[0], a.send[1], static_cast<uint32_t *>(remote_l),
                       static_cast<uint32_t *>(remote_r), static_cast<uint32_t *>(flag_l), static_cast<uint32_t *>(flag_r),
                       a.flags + FLAG_PEER_DONE, c.halo_cap, tag);
}
void launch_peer_wait(hipStream_t st, const Arrays &a, const void *flag_l, const void *flag_r, uint32_t tag) {
    hipLaunchKernelGGL(k_peer_wait, dim3(1), dim3(64), 0, st, static_cast<const uint32_t *>(flag_l), static_cast<const uint32_t *>(flag_r), a.flags, tag);
}

}  // namespace sph

#include "sph_list.inc"

namespace sph {

void launch_density(hipStream_t st, const Consts &c, const Arrays &a, int cap, int mode, int variant, bool consume_rebuild,
                    int pass, bool store_p, bool spec, bool verify, int head_blocks) {
    if (cap <= 0) return;
    if (variant == 0) { launch_density_list(st, c, a, cap, mode, consume_rebuild, pass, store_p, spec, verify, head_blocks); return; }
    if (pass == DENS_INTERIOR) return;      // the direct variant is not split: everything in the final pass
    dim3 g((cap + BLK - 1) / BLK), b(BLK);
    uint32_t *rb = consume_rebuild ? a.rebuild : nullptr;
    if (mode == DENS_RHO_EOS)
        hipLaunchKernelGGL(k_density_direct<true>, g, b, 0, st, c, a.pos, a.skey, a.cell_start, a.bpos, a.bpsi, a.bcell_start,
                           a.rp, a.prs, rb, a.check, a.dn);
    else
        hipLaunchKernelGGL(k_density_direct<false>, g, b, 0, st, c, a.pos, a.skey, a.cell_start, a.bpos, a.bpsi, a.bcell_start,
                           a.rp, a.prs, rb, a.check, a.dn);
}

// mode: FORCE_EVAL / FORCE_KICK / FORCE_KICK_DRIFT (the latter with the list kernels only: sph_abi.hip never asks the
// direct variant for it)
void launch_force(hipStream_t st, const Consts &c, const Arrays &a, int cap, int mode, int variant) {
    if (cap <= 0) return;
    if (variant == 0) { launch_force_list(st, c, a, cap, mode); return; }
    dim3 g((cap + BLK - 1) / BLK), b(BLK);
    if (mode != FORCE_EVAL)
        hipLaunchKernelGGL(k_force_direct<true>, g, b, 0, st, c, a.pos, a.vel, a.rp, a.skey, a.cell_start, a.bpos, a.bpsi,
                           a.bcell_start, a.grav, a.acc, a.velt, a.dn, a.bvel);
    else
        hipLaunchKernelGGL(k_force_direct<false>, g, b, 0, st, c, a.pos, a.vel, a.rp, a.skey, a.cell_start, a.bpos, a.bpsi,
                           a.bcell_start, a.grav, a.acc, a.velt, a.dn, a.bvel);
}

// slab mode read-back: the owned particles (sorted order) as compact AoS + their global ids (+ accelerations)
__global__ __launch_bounds__(BLK) void k_export_owned(Consts c, const float2 *__restrict__ pos,
                                                      const float2 *__restrict__ velt, const uint32_t *__restrict__ id,
                                                      const float2 *__restrict__ rp, const float *__restrict__ prs,
                                                      const float2 *__restrict__ acc, const uint32_t *__restrict__ cs,
                                                      uint32_t *__restrict__ dn, sph_particle *__restrict__ out,
                                                      uint32_t *__restrict__ ids, float *__restrict__ du,
                                                      float *__restrict__ dv) {
    const int beg = (int)cs[c.ghost * c.rows], n = (int)cs[(c.ghost + c.owned) * c.rows] - beg;
    const int t = blockIdx.x * BLK + threadIdx.x;
    if (t == 0) dn[2] = (uint32_t)n;
    if (t >= n) return;
    const int i = beg + t;
    sph_particle q;
    q.x = pos[i].x; q.y = pos[i].y; q.u = velt[i].x; q.v = velt[i].y; q.m = c.m_fluid; q.rho = rp[i].x; q.p = prs[i];
    out[t] = q;
    ids[t] = id[i];
    du[t] = acc[i].x;
    dv[t] = acc[i].y;
}

void launch_export_owned(hipStream_t st, const Consts &c, const Arrays &a, int cap, sph_particle *out_dev, uint32_t *ids_dev,
                         float *du, float *dv) {
    if (cap <= 0) return;
    hipLaunchKernelGGL(k_export_owned, dim3((cap + BLK - 1) / BLK), dim3(BLK), 0, st, c, a.pos, a.velt, a.id, a.rp, a.prs,
                       a.acc, a.cell_start, a.dn, out_dev, ids_dev, du, dv);
}

// ------------------------------------------------------------------------------------------
// boundary: bin once (:600) and Akinci pseudo-mass (:242-261)
__global__ __launch_bounds__(BLK) void k_boundary_key(Consts c, const float2 *__restrict__ bpos_in,
                                                      uint32_t *__restrict__ key, uint32_t *__restrict__ slot,
                                                      uint32_t *__restrict__ count, uint32_t *__restrict__ dirty,
                                                      uint32_t *__restrict__ flags, int nb) {
    int i = blockIdx.x * BLK + threadIdx.x;
    if (i >= nb) return;
    float2 p = bpos_in[i];
    int row, col;
    bool oob, bad;
    cell_of(c, p.x, p.y, row, col, oob, bad);
    uint32_t k = (uint32_t)(col * c.rows + row);
    key[i] = k;
    slot[i] = atomicAdd(&count[k], 1u);
    dirty[k / SCAN_TILE] = 1u;
    if (bad) atomicAdd(&flags[FLAG_NAN], 1u);
    else if (oob) atomicAdd(&flags[FLAG_OOB], 1u);
}

__global__ __launch_bounds__(BLK) void k_boundary_reorder(const float2 *__restrict__ bpos_in,
                                                          const uint32_t *__restrict__ key,
                                                          const uint32_t *__restrict__ slot,
                                                          const uint32_t *__restrict__ cell_start,
                                                          float2 *__restrict__ bpos, uint32_t *__restrict__ bid, int nb,
                                                          const float2 *__restrict__ bvel_in, float2 *__restrict__ bvel,
                                                          const uint32_t *__restrict__ cell_ids) {
    int i = blockIdx.x * BLK + threadIdx.x;
    if (i >= nb) return;
    // the order inside a cell: by index (the slots were handed out in the order the atomics arrived)
    const uint32_t k = key[i], beg = cell_start[k], end = cell_start[k + 1];
    uint32_t dst = beg;
    for (uint32_t j = beg; j < end; j++) dst += cell_ids[j] < (uint32_t)i ? 1u : 0u;
    bpos[dst] = bpos_in[i];
    bvel[dst] = bvel_in[i];
    bid[dst] = (uint32_t)i;
}
// the members of every cell, in slot order (input of the deterministic rank above / in reorder_body)
__global__ __launch_bounds__(BLK) void k_boundary_cell_ids(const uint32_t *__restrict__ key, const uint32_t *__restrict__ slot,
                                                           const uint32_t *__restrict__ cell_start, uint32_t *__restrict__ cell_ids, int nb) {
    int i = blockIdx.x * BLK + threadIdx.x;
    if (i < nb) cell_ids[cell_start[key[i]] + slot[i]] = (uint32_t)i;
}

__global__ __launch_bounds__(BLK) void k_boundary_psi(Consts c, const float2 *__restrict__ bpos,
                                                      const uint32_t *__restrict__ bcs, float *__restrict__ bpsi, int nb) {
    int i = blockIdx.x * BLK + threadIdx.x;
    if (i >= nb) return;
    float2 pi = bpos[i];
    int row, col;
    bool oob, bad;
    cell_of(c, pi.x, pi.y, row, col, oob, bad);
    int r0 = max(row - 1, 0), r1 = min(row + 1, c.rows - 1);
    float s = 0.0f;
    for (int cc = max(col - 1, 0); cc <= min(col + 1, c.cols - 1); cc++) {
        int base = cc * c.rows;
        uint32_t beg = bcs[base + r0], end = bcs[base + r1 + 1];
        for (uint32_t j = beg; j < end; j++) {
            float2 pj = bpos[j];
            float dx = pi.x - pj.x, dy = pi.y - pj.y;
            float d2 = fmaf(dx, dx, dy * dy);
            // self excluded by index (:130,:144); a coincident duplicate (d = 0) does count, W = nf
            if (j != (uint32_t)i && d2 < c.cut2) s += w_shape(c, d2);
        }
    }
    bpsi[i] = c.rho0 / (c.nf * s);      // psi = rho_0 / sum W   :259
}

// walls are few and static: most fluid particles have none within reach, and finding that out from the bins costs six
// dependent loads per particle and pass.  One byte per cell answers it: is there a wall particle in the 3x3 cells around?
__global__ __launch_bounds__(BLK) void k_boundary_near(Consts c, const uint32_t *__restrict__ bcs, unsigned char *__restrict__ bnear) {
    const int cell = blockIdx.x * BLK + threadIdx.x;
    if (cell >= c.n_cells) return;
    int row, col;
    cell_of_key(c, (uint32_t)cell, row, col);
    const int r0 = max(row - 1, 0), r1 = min(row + 1, c.rows - 1);
    uint32_t cnt = 0u;
    for (int cc = max(col - 1, 0); cc <= min(col + 1, c.cols - 1); cc++) cnt += bcs[cc * c.rows + r1 + 1] - bcs[cc * c.rows + r0];
    bnear[cell] = cnt ? 1 : 0;
}
__global__ __launch_bounds__(BLK) void k_fill_float2(float2 *__restrict__ dst, float x, float y, int n) {
    const int i = blockIdx.x * BLK + threadIdx.x;
    if (i < n) dst[i] = make_float2(x, y);
}
void launch_fill_float2(hipStream_t st, float2 *dst, float x, float y, int n) {
    if (n > 0) hipLaunchKernelGGL(k_fill_float2, dim3((n + BLK - 1) / BLK), dim3(BLK), 0, st, dst, x, y, n);
}
void launch_boundary_near(hipStream_t st, const Consts &c, const Arrays &a) {
    hipLaunchKernelGGL(k_boundary_near, dim3((c.n_cells + BLK - 1) / BLK), dim3(BLK), 0, st, c, a.bcell_start, a.bnear);
}

__global__ __launch_bounds__(BLK) void k_boundary_gather_psi(const float *__restrict__ psi_in, const uint32_t *__restrict__ bid,
                                                             float *__restrict__ bpsi, int nb) {
    int i = blockIdx.x * BLK + threadIdx.x;
    if (i >= nb) return;
    bpsi[i] = psi_in[bid[i]];
}
__global__ __launch_bounds__(BLK) void k_boundary_unsort_psi(const float *__restrict__ bpsi, const uint32_t *__restrict__ bid,
                                                             float *__restrict__ psi_out, int nb) {
    int i = blockIdx.x * BLK + threadIdx.x;
    if (i >= nb) return;
    psi_out[bid[i]] = bpsi[i];
}
void launch_boundary_unsort_psi(hipStream_t st, const Arrays &a, float *psi_out_original_order, int nb) {
    if (nb <= 0) return;
    hipLaunchKernelGGL(k_boundary_unsort_psi, dim3((nb + BLK - 1) / BLK), dim3(BLK), 0, st, a.bpsi, a.bid, psi_out_original_order, nb);
}
void launch_boundary_gather_psi(hipStream_t st, const Arrays &a, const float *psi_in, int nb) {
    if (nb <= 0) return;
    hipLaunchKernelGGL(k_boundary_gather_psi, dim3((nb + BLK - 1) / BLK), dim3(BLK), 0, st, psi_in, a.bid, a.bpsi, nb);
}

void launch_boundary_key(hipStream_t st, const Consts &c, const float2 *bpos_in, uint32_t *key, uint32_t *slot,
                         uint32_t *count, uint32_t *dirty, uint32_t *flags, int nb) {
    if (nb <= 0) return;
    hipLaunchKernelGGL(k_boundary_key, dim3((nb + BLK - 1) / BLK), dim3(BLK), 0, st, c, bpos_in, key, slot, count, dirty,
                       flags, nb);
}
void launch_boundary_reorder(hipStream_t st, const float2 *bpos_in, const uint32_t *key, const uint32_t *slot,
                             const uint32_t *cell_start, float2 *bpos, uint32_t *bid, int nb, const float2 *bvel_in,
                             float2 *bvel, uint32_t *cell_ids_tmp) {
    if (nb <= 0) return;
    hipLaunchKernelGGL(k_boundary_cell_ids, dim3((nb + BLK - 1) / BLK), dim3(BLK), 0, st, key, slot, cell_start, cell_ids_tmp, nb);
    hipLaunchKernelGGL(k_boundary_reorder, dim3((nb + BLK - 1) / BLK), dim3(BLK), 0, st, bpos_in, key, slot, cell_start,
                       bpos, bid, nb, bvel_in, bvel, cell_ids_tmp);
}
void launch_boundary_psi(hipStream_t st, const Consts &c, const Arrays &a, int nb) {
    if (nb <= 0) return;
    hipLaunchKernelGGL(k_boundary_psi, dim3((nb + BLK - 1) / BLK), dim3(BLK), 0, st, c, a.bpos, a.bcell_start, a.bpsi, nb);
}

// ------------------------------------------------------------------------------------------
// read-back / upload helpers (original particle order <-> cell order through the carried id)
__global__ __launch_bounds__(BLK) void k_unsort_particles(Consts c, const float2 *__restrict__ pos,
                                                          const float2 *__restrict__ velt,
                                                          const uint32_t *__restrict__ id, const float2 *__restrict__ rp,
                                                          const float *__restrict__ prs, sph_particle *__restrict__ out,
                                                          int n) {
    int i = blockIdx.x * BLK + threadIdx.x;
    if (i >= n) return;
    float2 p = pos[i], v = velt[i];
    sph_particle q;
    q.x = p.x; q.y = p.y; q.u = v.x; q.v = v.y; q.m = c.m_fluid; q.rho = rp[i].x; q.p = prs[i];
    out[id[i]] = q;
}
__global__ __launch_bounds__(BLK) void k_unsort_accel(const float2 *__restrict__ acc, const uint32_t *__restrict__ id,
                                                      float *__restrict__ du, float *__restrict__ dv, int n) {
    int i = blockIdx.x * BLK + threadIdx.x;
    if (i >= n) return;
    float2 a = acc[i];
    uint32_t k = id[i];
    du[k] = a.x;
    dv[k] = a.y;
}
__global__ __launch_bounds__(BLK) void k_unsort_boundary(Consts c, const float2 *__restrict__ bpos,
                                                         const float2 *__restrict__ bvel,
                                                         const float *__restrict__ bpsi, const uint32_t *__restrict__ bid,
                                                         sph_particle *__restrict__ out, int nb) {
    int i = blockIdx.x * BLK + threadIdx.x;
    if (i >= nb) return;
    sph_particle q;
    q.x = bpos[i].x; q.y = bpos[i].y; q.u = bvel[i].x; q.v = bvel[i].y; q.m = bpsi[i]; q.rho = c.rho0; q.p = 0;
    out[bid[i]] = q;
}
__global__ __launch_bounds__(BLK) void k_upload_state(const sph_particle *__restrict__ in, float2 *__restrict__ pos,
                                                      float2 *__restrict__ velt, uint32_t *__restrict__ id, int n) {
    int i = blockIdx.x * BLK + threadIdx.x;
    if (i >= n) return;
    sph_particle q = in[i];
    pos[i] = make_float2(q.x, q.y);
    velt[i] = make_float2(q.u, q.v);
    id[i] = (uint32_t)i;
}
__global__ __launch_bounds__(BLK) void k_gather_rho_p(const sph_particle *__restrict__ in, const uint32_t *__restrict__ id,
                                                      float2 *__restrict__ rp, float *__restrict__ prs, int n) {
    int i = blockIdx.x * BLK + threadIdx.x;
    if (i >= n) return;
    sph_particle q = in[id[i]];
    rp[i] = make_float2(q.rho, eos_p_over_rho2(q.p, q.rho));      // (as eos())
    prs[i] = q.p;
}

void launch_unsort_particles(hipStream_t st, const Consts &c, const Arrays &a, int n, sph_particle *out_dev) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_unsort_particles, dim3((n + BLK - 1) / BLK), dim3(BLK), 0, st, c, a.pos, a.velt, a.id, a.rp, a.prs,
                       out_dev, n);
}
void launch_unsort_accel(hipStream_t st, const Arrays &a, int n, float *du, float *dv) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_unsort_accel, dim3((n + BLK - 1) / BLK), dim3(BLK), 0, st, a.acc, a.id, du, dv, n);
}
// du_dt, dv_dt given in original order -> the sorted acc array (inverse of k_unsort_accel)
__global__ __launch_bounds__(BLK) void k_gather_accel(const float *__restrict__ du, const float *__restrict__ dv,
                                                      const uint32_t *__restrict__ id, float2 *__restrict__ acc, int n) {
    int i = blockIdx.x * BLK + threadIdx.x;
    if (i >= n) return;
    const uint32_t k = id[i];
    acc[i] = make_float2(du[k], dv[k]);
}
void launch_gather_accel(hipStream_t st, const Arrays &a, int n, const float *du, const float *dv) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_gather_accel, dim3((n + BLK - 1) / BLK), dim3(BLK), 0, st, du, dv, a.id, a.acc, n);
}
void launch_unsort_boundary(hipStream_t st, const Consts &c, const Arrays &a, int nb, sph_particle *out_dev) {
    if (nb <= 0) return;
    hipLaunchKernelGGL(k_unsort_boundary, dim3((nb + BLK - 1) / BLK), dim3(BLK), 0, st, c, a.bpos, a.bvel, a.bpsi, a.bid, out_dev, nb);
}
void launch_upload_state(hipStream_t st, const Arrays &a, int n, const sph_particle *in_dev) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_upload_state, dim3((n + BLK - 1) / BLK), dim3(BLK), 0, st, in_dev, a.pos, a.velt, a.id, n);
}
void launch_gather_rho_p(hipStream_t st, const Consts &c, const Arrays &a, int n, const sph_particle *in_dev) {
    (void)c;
    if (n <= 0) return;
    hipLaunchKernelGGL(k_gather_rho_p, dim3((n + BLK - 1) / BLK), dim3(BLK), 0, st, in_dev, a.id, a.rp, a.prs, n);
}

// the velocity between steps (after the second half kick, :638-639) from the half-kicked velocity and the acceleration
// of the last force pass: what the fused force pass would have stored, same fma
__global__ __launch_bounds__(BLK) void k_refresh_velt(Consts c, const float2 *__restrict__ vel, const float2 *__restrict__ acc,
                                                      float2 *__restrict__ velt, const uint32_t *__restrict__ dn) {
    const int i = blockIdx.x * BLK + threadIdx.x;
    if (i >= (int)dn[0]) return;
    const float2 v = vel[i], a = acc[i];
    velt[i] = make_float2(fmaf(c.half_dt, a.x, v.x), fmaf(c.half_dt, a.y, v.y));
}
void launch_refresh_velt(hipStream_t st, const Consts &c, const Arrays &a, int cap) {
    if (cap <= 0) return;
    hipLaunchKernelGGL(k_refresh_velt, dim3((cap + BLK - 1) / BLK), dim3(BLK), 0, st, c, a.vel, a.acc, a.velt, a.dn);
}

// ------------------------------------------------------------------------------------------
// statistics of :657-671 as device max-reductions (non-negative floats order like their bit patterns)
// slab: the owned range of the sorted arrays only (the ghosts belong to the neighbours' statistics)
__global__ __launch_bounds__(BLK) void k_stats(Consts c, const float2 *__restrict__ rp, const float2 *__restrict__ velt,
                                               const uint32_t *__restrict__ cs, uint32_t *__restrict__ flags, int n, int slab) {
    float mr = 0.0f, ms = 0.0f;
    const int lo = slab ? (int)cs[c.ghost * c.rows] : 0, hi = slab ? (int)cs[(c.ghost + c.owned) * c.rows] : n;
    for (int i = lo + blockIdx.x * BLK + threadIdx.x; i < hi; i += gridDim.x * BLK) {
        mr = fmaxf(mr, rp[i].x);
        float2 v = velt[i];
        ms = fmaxf(ms, fmaf(v.x, v.x, v.y * v.y));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        mr = fmaxf(mr, __shfl_xor(mr, d, 64));
        ms = fmaxf(ms, __shfl_xor(ms, d, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&flags[FLAG_MAXRHO], __float_as_uint(fmaxf(mr, 0.0f)));
        atomicMax(&flags[FLAG_MAXSPEED], __float_as_uint(sqrtf(ms)));
    }
}
void launch_stats(hipStream_t st, const Consts &c, const Arrays &a, int n, bool slab) {
    if (n <= 0) return;
    int blocks = min((n + BLK - 1) / BLK, 2048);
    hipLaunchKernelGGL(k_stats, dim3(blocks), dim3(BLK), 0, st, c, a.rp, a.velt, a.cell_start, a.flags, n, slab ? 1 : 0);
}

// ------------------------------------------------------------------------------------------
// metaballs (:380-411): one thread per pixel of the 128x64 panel, pixel centres as :573.  The eight pixels of one
// output byte (SSD1306 page format: bit i%8 of byte (i/8)*128 + j, :407-408) sit in eight consecutive lanes, so a wave
// ballot holds eight finished bytes.
__global__ __launch_bounds__(BLK) void k_metaballs(Consts c, const float2 *__restrict__ pos,
                                                   const uint32_t *__restrict__ cs, float width, float height,
                                                   float inv_w_half_px, unsigned char *__restrict__ page_bytes) {
    const int t = blockIdx.x * BLK + threadIdx.x;      // grid covers exactly 64 * 128 pixels
    const int byte = t >> 3, bit = t & 7;
    const int i = (byte >> 7) * 8 + bit, j = byte & 127;
    float px = c.x_min + (float)((j + 0.5) * (double)width / 128.0);
    float py = c.y_min + (float)((64 - (i + 0.5)) * (double)height / 64.0);
    int row, col;
    bool oob, bad;
    cell_of(c, px, py, row, col, oob, bad);
    // slab mode: a slab renders the pixels whose centres lie in its owned columns (every pixel belongs to exactly one slab;
    // the host ORs the pages); the two ghost columns per side are what the 5x5 walk below needs
    const bool mine = c.ghost == 0 || (!oob && col >= c.ghost && col < c.ghost + c.owned);
    // 5x5 sort cells: a particle may be up to H + skin from where it was sorted (see drift_verdict)
    int r0 = max(row - 2, 0), r1 = min(row + 2, c.rows - 1);
    float s = 0.0f;
    for (int cc = max(col - 2, 0); mine && cc <= min(col + 2, c.cols - 1); cc++) {
        int base = cc * c.rows;
        uint32_t beg = cs[base + r0], end = cs[base + r1 + 1];
        for (uint32_t k = beg; k < end; k++) {
            float2 pj = pos[k];
            float dx = px - pj.x, dy = py - pj.y;
            float d2 = fmaf(dx, dx, dy * dy);
            if (d2 < c.cut2) s += w_shape(c, d2);
        }
    }
    const unsigned long long lit = __ballot(s * c.nf * inv_w_half_px >= 1.0f);     // sum W / W(px/2) >= 1   :401-407
    if (bit == 0) page_bytes[byte] = (unsigned char)((lit >> (threadIdx.x & 63)) & 0xffull);
}
void launch_metaballs(hipStream_t st, const Consts &c, const Arrays &a, float width, float height, unsigned char *page_bytes_dev) {
    // W(px_width/2) with px_width = WIDTH/128 (:399-401), evaluated like the device W
    float half_px = width / 128.0f / 2.0f;
    float q = half_px / c.h;
    float t = 1.0f - 0.5f * q;
    float w = c.nf * (t * t) * (t * t) * (1.0f + 2.0f * q);
    hipLaunchKernelGGL(k_metaballs, dim3(64 * 128 / BLK), dim3(BLK), 0, st, c, a.pos, a.cell_start, width, height, 1.0f / w,
                       page_bytes_dev);
}

}  // namespace sph

#ifdef SPH_SPEC_TRACE
// measurement builds only: the begin / end clock of the first `wgs` workgroups of the last speculative density launch (100 MHz
// ticks; 2 words each) and the verify jobs' counters (4 words, summed over their waves), read and reset.  The caller has synchronised.
extern "C" int sph_spec_trace(unsigned long long *out, int wgs, unsigned int *counts) {
    if (wgs > sph::SPEC_TRACE_WGS) wgs = sph::SPEC_TRACE_WGS;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(sph::g_spec_trace), sizeof(unsigned long long) * 2 * (size_t)wgs) != hipSuccess) return -1;
    static unsigned int per_wave[sph::SPEC_TRACE_WGS][4];
    if (hipMemcpyFromSymbol(per_wave, HIP_SYMBOL(sph::g_spec_trace_n), sizeof per_wave) != hipSuccess) return -1;
    counts[0] = counts[1] = counts[2] = counts[3] = 0u;
    for (int w = 0; w < sph::SPEC_TRACE_WGS; w++)
        for (int k = 0; k < 3; k++) counts[k] += per_wave[w][k];
    for (int w = 0; w < sph::SPEC_TRACE_WGS; w++) per_wave[w][0] = per_wave[w][1] = per_wave[w][2] = per_wave[w][3] = 0u;
    if (hipMemcpyToSymbol(HIP_SYMBOL(sph::g_spec_trace_n), per_wave, sizeof per_wave) != hipSuccess) return -1;
    return wgs;
}
#endif
