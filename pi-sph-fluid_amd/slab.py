"""Host side of the x-slab domain decomposition: one process per GPU.

The reference has no distributed path.  Its particle loops (pi_sph_fluid.c:272, :311) shard by cell
column: a slab owns the global cell columns [c0, c1) and keeps two ghost columns per side, so ONE
exchange per step (after kick+drift) carries both the ghosts and the ownership migration.  All slabs
rebuild their neighbour structure in the same step: each raises a 32-bit "rebuild" word when its
lists may be stale, the host MAX-reduces the word over all ranks, and the halo message of the step
is a set of full records (rebuild step) or a plain position/velocity update of the interface
columns (any other step).  This module is backend-agnostic host logic:

  partition_columns()  column ranges with ~equal particle counts (quantiles of the per-column histogram)
  local_subset()       the particles a slab starts with (owned + ghosts) and their global ids
  GpuSlab              the C-ABI slab context (sph_create_slab ... sph_slab_read)
  LocalTransport       in-process exchange between several slabs on one device (tests, 1 GPU)
  TorchTransport       torch.distributed P2P (backend "nccl" = RCCL over xGMI on the GPU box; "gloo" on CPU)
  SlabRunner           step loop: begin -> reduce the rebuild word -> pack -> exchange -> end

Halo buffer format (include/sph.h): uint32 header[4] = {update count, update step, record count, record step}
then one payload: records of 5 words {x, y, u, v, id} (a rebuild step) or update entries of 4 words
{x, y, u, v} (any other step; written by the force pass of the step before); fixed capacity, always sent
whole (messages are 0.1-1 MB: latency-bound on xGMI).
"""
import ctypes as C

import numpy as np

HALO_HDR = 4
HALO_REC = 5
GHOST = 2


_cell_fn = None


def set_cell_fn(fn):
    """fn(prm) -> cell length of the device grid (2H + skin); the package sets this to the C ABI's sph_device_cell."""
    global _cell_fn
    _cell_fn = fn


def device_cell(prm):
    if _cell_fn is not None:
        return np.float32(_cell_fn(prm))
    return np.float32(2) * np.float32(prm.h)


def global_columns(prm, x):
    """global cell column of positions x — the device's arithmetic (cell_of): (int)((x - x_min) * (1/cell)), f32."""
    inv = np.float32(1.0) / device_cell(prm)
    return ((np.asarray(x, np.float32) - np.float32(prm.x_min)) * inv).astype(np.int64)


def grid_columns(prm):
    return int((np.float32(prm.x_max) - np.float32(prm.x_min)) / device_cell(prm)) + 1


def grid_rows(prm):
    return int((np.float32(prm.y_max) - np.float32(prm.y_min)) / device_cell(prm)) + 1


def _cuts_from_histogram(hist, world, lo, hi):
    """column boundaries [lo = cuts[0] < ... < cuts[world] = hi] at the quantiles of the per-column histogram, every
    slab at least 4 columns wide."""
    cum = np.cumsum(hist)
    total = int(cum[-1])
    cuts = [lo]
    for r in range(1, world):
        target = total * r / world
        c = int(np.searchsorted(cum, target)) + 1        # first column boundary at or past the quantile
        c = max(c, cuts[-1] + 4)                           # a slab owns at least 4 columns
        cuts.append(c)
    cuts.append(hi)
    for r in range(world - 1, 0, -1):                      # keep >= 4 columns walking back from the end
        cuts[r] = min(cuts[r], cuts[r + 1] - 4)
    if any(cuts[r + 1] - cuts[r] < 4 for r in range(world)) or cuts[0] < 0:
        raise ValueError("scene too narrow for %d slabs" % world)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def partition_columns(prm, fluid, world, slack=None):
    """[(c0, c1)] * world: contiguous column ranges holding ~equal numbers of particles.  By default the slabs tile
    the whole box: the first begins at column 0 and the last ends at the box edge, so a dam-break front can run
    through the dry part of the box without ever leaving the decomposition (the cells of a dry column cost 12 bytes
    each and the scan skips them).  slack = k restricts the decomposition to [first occupied - k, last occupied + k]
    (small local grids for tests; a particle leaving that range is reported as out of domain)."""
    cols = grid_columns(prm)
    gc = np.clip(global_columns(prm, fluid["x"]), 0, cols - 1)
    hist = np.bincount(gc, minlength=cols)
    occ = np.nonzero(hist)[0]
    lo = 0 if slack is None else max(0, int(occ[0]) - slack)
    hi = cols if slack is None else min(cols, int(occ[-1]) + 1 + slack)
    return _cuts_from_histogram(hist, world, lo, hi)


def block_lattice_columns(prm, spec):
    """global cell column of every lattice column i of a block scene spec = (box, x0, y0, nx, ny): x_i = x0 + i R in
    f32, exactly as sph_scene_block computes it."""
    _, x0, _, nx, _ = spec
    xi = np.float32(x0) + np.arange(nx, dtype=np.float32) * np.float32(prm.r)
    return global_columns(prm, xi)


def partition_block(prm, spec, world):
    """partition_columns for a lattice block without generating it (per-column counts follow from the lattice)."""
    ny = spec[4]
    cols = grid_columns(prm)
    hist = np.bincount(np.clip(block_lattice_columns(prm, spec), 0, cols - 1), minlength=cols) * ny
    return _cuts_from_histogram(hist, world, 0, cols)


def local_block_subset(pkg, prm, spec, c0, c1):
    """the particles of a block scene inside columns [c0 - 2, c1 + 2) and their global ids, generating only those
    lattice columns (sph_scene_block_range): what local_subset(prm, scene_block(...), c0, c1) would return."""
    _, x0, y0, nx, ny = spec
    gc = block_lattice_columns(prm, spec)
    sel = np.nonzero((gc >= c0 - GHOST) & (gc < c1 + GHOST))[0]
    if len(sel) == 0:
        return np.zeros(0, pkg.PARTICLE), np.zeros(0, np.uint32)
    i0, i1 = int(sel[0]), int(sel[-1]) + 1                 # lattice columns are monotone in x: a contiguous range
    loc = pkg.block_range(prm, x0, y0, nx, ny, i0, i1)
    ids = (np.arange(i0 * ny, i1 * ny, dtype=np.int64)).astype(np.uint32)
    return loc, ids


def local_subset(prm, fluid, c0, c1):
    """particles of columns [c0 - 2, c1 + 2) and their global ids (index into `fluid`)."""
    gc = global_columns(prm, fluid["x"])
    ids = np.nonzero((gc >= c0 - GHOST) & (gc < c1 + GHOST))[0].astype(np.uint32)
    return np.ascontiguousarray(fluid[ids]), ids


def halo_words(capacity):
    return HALO_HDR + HALO_REC * capacity


def default_halo_capacity(prm):
    return 64 * grid_rows(prm)


class GpuSlab:
    """One slab on one GPU through the C ABI (sph_create_slab ...)."""

    def __init__(self, pkg, prm, fluid, boundary_all, c0, c1, has_left, has_right, gx=0.0, gy=-9.81, device=0,
                 halo_capacity=0, particle_capacity=0, local=None):
        """fluid = the whole scene (the slab keeps its part), or None with local = (particles, global ids) of the
        columns [c0 - 2, c1 + 2) when the host generated only those."""
        self.pkg, self.L = pkg, pkg.hip_lib()
        self.c0, self.c1 = c0, c1
        loc, ids = local if local is not None else local_subset(prm, fluid, c0, c1)
        loc, ids = np.ascontiguousarray(loc, pkg.PARTICLE), np.ascontiguousarray(ids, np.uint32)
        self.halo_capacity = halo_capacity or default_halo_capacity(prm)
        self.particle_capacity = particle_capacity or (len(loc) + len(loc) // 4 + 2 * self.halo_capacity + 1024)
        desc = pkg.SlabDesc(c0, c1, int(has_left), int(has_right), self.halo_capacity, self.particle_capacity)
        boundary_all = np.ascontiguousarray(boundary_all, pkg.PARTICLE)
        self.h = C.c_void_p()
        rc = self.L.sph_create_slab(C.byref(self.h), C.byref(prm), C.byref(desc), loc.ctypes.data_as(C.c_void_p),
                                    ids.ctypes.data_as(C.c_void_p), len(loc), boundary_all.ctypes.data_as(C.c_void_p),
                                    len(boundary_all), gx, gy, device)
        if rc:
            msg = self.L.sph_last_error(self.h).decode() if self.h else "sph_create_slab failed"
            self.close()
            raise pkg.SphError(rc, msg)
        self.words = halo_words(self.halo_capacity)
        self._torch_bufs = None

    def _chk(self, rc):
        if rc:
            raise self.pkg.SphError(rc, self.L.sph_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.sph_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def step_begin(self, gx, gy):
        self._chk(self.L.sph_slab_step_begin(self.h, gx, gy))

    def step_pack(self):
        self._chk(self.L.sph_slab_step_pack(self.h))

    def step_overlap(self):
        self._chk(self.L.sph_slab_step_overlap(self.h))

    def step_end(self):
        self._chk(self.L.sph_slab_step_end(self.h))

    def step_lean(self, gx, gy):
        """the whole step as ONE call (sph_slab_step: four kernels); a slab without neighbours, or with peer links set by its host;
        needs set_rebuild_launches(True)"""
        self.L.sph_slab_step.argtypes = [C.c_void_p, C.c_float, C.c_float]
        self._chk(self.L.sph_slab_step(self.h, gx, gy))

    # the rebuild word
    def flag_get(self):
        v = C.c_uint32()
        self._chk(self.L.sph_slab_flag_get(self.h, C.byref(v)))
        return int(v.value)

    def flag_set(self, value):
        self._chk(self.L.sph_slab_flag_set(self.h, int(value)))

    def flag_tensor(self, torch, device):
        """a 1-element int32 tensor of the host framework adopted as the rebuild word (all-reduced by the transport)."""
        if getattr(self, "_flag_t", None) is None:
            t = torch.zeros(1, dtype=torch.int32, device=device)
            self._chk(self.L.sph_slab_set_flag_buffer(self.h, C.c_void_p(t.data_ptr())))
            self._flag_t = t
        return self._flag_t

    def sync(self):
        self._chk(self.L.sph_sync(self.h))

    def set_stream(self, hip_stream):
        self._chk(self.L.sph_set_stream(self.h, C.c_void_p(hip_stream)))

    # host-staged transport
    def copy_out(self, side):
        buf = np.zeros(self.words, np.uint32)
        self._chk(self.L.sph_slab_copy_out(self.h, side, buf.ctypes.data_as(C.c_void_p)))
        return buf

    def copy_in(self, side, buf):
        buf = np.ascontiguousarray(buf, np.uint32)
        assert len(buf) == self.words
        self._chk(self.L.sph_slab_copy_in(self.h, side, buf.ctypes.data_as(C.c_void_p)))

    # device-resident transport: torch owns the buffers, the library packs into / ingests from them
    def halo_tensors(self, torch, device):
        if self._torch_bufs is None:
            bufs = [torch.zeros(self.words, dtype=torch.int32, device=device) for _ in range(4)]
            self._chk(self.L.sph_slab_set_buffers(self.h, *[C.c_void_p(b.data_ptr()) for b in bufs], self.words * 4))
            self._torch_bufs = bufs
        return self._torch_bufs      # send_left, send_right, recv_left, recv_right

    def read(self):
        """owned particles, their global ids and accelerations."""
        cap = self.particle_capacity
        out = np.zeros(cap, self.pkg.PARTICLE)
        ids = np.zeros(cap, np.uint32)
        du, dv = np.zeros(cap, np.float32), np.zeros(cap, np.float32)
        n = C.c_int()
        self._chk(self.L.sph_slab_read(self.h, out.ctypes.data_as(C.c_void_p), ids.ctypes.data_as(C.c_void_p),
                                       du.ctypes.data_as(C.c_void_p), dv.ctypes.data_as(C.c_void_p), cap, C.byref(n)))
        k = n.value
        return out[:k], ids[:k], du[:k], dv[:k]

    def counts(self):
        a, b = C.c_int(), C.c_int()
        self._chk(self.L.sph_slab_counts(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def stats(self):
        """max rho and max speed over the particles this slab owns (the host takes the maximum over the ranks)."""
        a, b = C.c_float(), C.c_float()
        self._chk(self.L.sph_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def render(self):
        """this slab's part of the 128 x 64 metaball frame (1024 bytes, SSD1306 page format): the pixels whose centres lie in
        its owned columns; the bitwise OR over all slabs is the frame."""
        buf = np.zeros(1024, np.uint8)
        self._chk(self.L.sph_render_metaballs(self.h, buf.ctypes.data_as(C.c_void_p)))
        return buf

    def set_rebuild_launches(self, one_launch):
        """True: what follows the halo exchange runs as ONE kernel with grid barriers.  Only when nothing else computes on
        the device while this slab steps (one rank per GPU, or slabs stepped strictly one after the other)."""
        self._chk(self.L.sph_set_rebuild_launches(self.h, 1 if one_launch else 0))

    def diagnostics(self):
        a = (C.c_longlong * 7)()
        self._chk(self.L.sph_direct_tile_reasons(self.h, a))
        return tuple(int(x) for x in a)

    def rebuilds(self):
        """rebuilds of the neighbour structure since creation (the same number on every slab)."""
        a, b = C.c_longlong(), C.c_longlong()
        self._chk(self.L.sph_rebuild_stats(self.h, C.byref(a), C.byref(b)))
        return a.value


class LocalTransport:
    """All slabs live in this process (one device): what slab r sends right is what slab r+1 receives left."""

    def __init__(self, slabs):
        self.slabs = slabs

    # re-balancing (SlabRunner.rebalance): the "ranks" are the slabs of this process
    def rebind(self, slabs):
        self.slabs = slabs

    def sum_over_ranks(self, per_slab_arrays):
        return np.sum(per_slab_arrays, axis=0)

    def redistribute(self, outgoing):
        """outgoing[r][q] = record array slab r sends to slab q  ->  incoming[q] = everything sent to q"""
        n = len(outgoing)
        return [np.concatenate([outgoing[r][q] for r in range(n)]) for q in range(n)]

    def reduce_flag(self):
        s = self.slabs
        if len(s) > 1:
            any_set = max(x.flag_get() for x in s)
            for x in s:
                x.flag_set(any_set)

    def exchange_start(self):
        self.exchange()

    def exchange_finish(self, handle):
        pass

    def exchange(self):
        s = self.slabs
        for r in range(len(s) - 1):
            right_of_r = s[r].copy_out(1)
            left_of_next = s[r + 1].copy_out(0)
            s[r + 1].copy_in(0, right_of_r)
            s[r].copy_in(1, left_of_next)


class TorchTransport:
    """One slab per process; neighbours exchange whole halo buffers by torch.distributed P2P
    (backend "nccl" is RCCL over xGMI; "gloo" on CPU).  A slab has at most two neighbours, so only
    two of a GPU's seven xGMI links carry traffic and the messages are latency-bound."""

    def __init__(self, torch, dist, slab, rank, world, device):
        self.torch, self.dist, self.rank, self.world, self.device = torch, dist, rank, world, device
        self.rebind([slab])

    # re-balancing (SlabRunner.rebalance): one slab per rank
    def rebind(self, slabs):
        slab = slabs[0]
        self.send_l, self.send_r, self.recv_l, self.recv_r = slab.halo_tensors(self.torch, self.device)
        self.flag = slab.flag_tensor(self.torch, self.device)

    def _dev(self, t):
        return t.to(self.device) if str(self.device) != "cpu" else t

    def sum_over_ranks(self, per_slab_arrays):
        t = self._dev(self.torch.from_numpy(np.ascontiguousarray(per_slab_arrays[0], np.int64)))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.cpu().numpy()

    def redistribute(self, outgoing):
        """outgoing[0][q] = float32 record array (k x RECW) this rank sends to rank q -> [everything sent to this rank]"""
        t, d = self.torch, self.dist
        mine = outgoing[0]
        counts = t.tensor([len(a) for a in mine], dtype=t.int64)
        allc = [self._dev(t.zeros(self.world, dtype=t.int64)) for _ in range(self.world)]
        d.all_gather(allc, self._dev(counts))
        allc = [c.cpu() for c in allc]                      # allc[r][q] = records r sends to q
        width = mine[0].shape[1] if mine[0].ndim == 2 else 0
        recv = [t.zeros((int(allc[r][self.rank]), width), dtype=t.float32) for r in range(self.world)]
        recv = [self._dev(x) for x in recv]
        ops, keep = [], []
        for q in range(self.world):
            if q == self.rank:
                continue
            if len(mine[q]):
                buf = self._dev(t.from_numpy(np.ascontiguousarray(mine[q], np.float32)))
                keep.append(buf)
                ops.append(d.P2POp(d.isend, buf, q))
            if recv[q].shape[0]:
                ops.append(d.P2POp(d.irecv, recv[q], q))
        for w in (d.batch_isend_irecv(ops) if ops else []):
            w.wait()
        parts = [mine[self.rank]] + [recv[q].cpu().numpy() for q in range(self.world) if q != self.rank]
        return [np.concatenate(parts)]

    def reduce_flag(self):
        """all slabs rebuild in the same step: MAX over ranks of the 4-byte rebuild word (stays on the device)."""
        if self.world > 1:
            self.dist.all_reduce(self.flag, op=self.dist.ReduceOp.MAX)

    def exchange_start(self):
        """post the sends / receives; they run beside whatever is enqueued next (the interior density pass)."""
        d, ops = self.dist, []
        if self.rank > 0:
            ops.append(d.P2POp(d.isend, self.send_l, self.rank - 1))
            ops.append(d.P2POp(d.irecv, self.recv_l, self.rank - 1))
        if self.rank < self.world - 1:
            ops.append(d.P2POp(d.isend, self.send_r, self.rank + 1))
            ops.append(d.P2POp(d.irecv, self.recv_r, self.rank + 1))
        return d.batch_isend_irecv(ops) if ops else []

    def exchange_finish(self, handle):
        for w in handle:
            w.wait()

    def exchange(self):
        self.exchange_finish(self.exchange_start())


class SlabRunner:
    """nsteps of: kick/drift -> reduce the rebuild word -> halo pack -> exchange (beside it: density of the interior
    tiles) -> (ingest + sort + lists | ghost update) + density of the rest + force (pi_sph_fluid.c:612-641)."""

    def __init__(self, slabs, transport, overlap=None, factory=None, prm=None, rank0=0, world=None, serialize=False):
        """factory(c0, c1, has_left, has_right, particles, ids, gx, gy) -> a new slab: needed only by rebalance(), with
        prm (the scene's parameters), rank0 (global index of this process's first slab) and world (slabs in total).
        serialize: the slabs of this process finish their steps strictly one after the other (a synchronisation after
        every sph_slab_step_end): what slabs that share a device owe each other when they run what follows the halo
        exchange as one launch with grid barriers (GpuSlab.set_rebuild_launches)."""
        self.serialize = serialize
        self.slabs = slabs if isinstance(slabs, (list, tuple)) else [slabs]
        self.transport = transport
        self.factory, self.prm, self.rank0 = factory, prm, rank0
        self.world = world if world is not None else len(self.slabs)
        # splitting the density pass costs one more launch: worth it only when there is an exchange to hide
        self.overlap = (len(self.slabs) > 1 or getattr(transport, "world", 1) > 1) if overlap is None else overlap

    def step(self, nsteps=1, gx=0.0, gy=-9.81, gravity=None):
        """gravity: optional callable k -> (gx, gy) sampled before every step (the reference re-reads g every step,
        :632); otherwise the constant (gx, gy)."""
        for k in range(nsteps):
            if gravity is not None:
                gx, gy = gravity(k)
            for s in self.slabs:
                s.step_begin(gx, gy)
            self.transport.reduce_flag()
            for s in self.slabs:
                s.step_pack()
            handle = self.transport.exchange_start()
            if self.overlap:
                for s in self.slabs:
                    s.step_overlap()         # density of the tiles that stage no ghost, beside the halo exchange
            self.transport.exchange_finish(handle)
            for s in self.slabs:
                s.step_end()
                if self.serialize:
                    s.sync()

    RECW = 8      # words per particle in a re-balancing message: x, y, u, v, m, rho, p, id (bits)

    def rebalance(self, gx=0.0, gy=-9.81, min_gain=0.05):
        """Dynamic re-balancing: new column ranges from the CURRENT per-column particle histogram
        (summed over all ranks), every particle moved to the slab that now holds its column (owned + 2 ghost columns),
        the slab contexts re-created from them.  Collective: every rank calls it between two steps.  A re-created
        context evaluates rho, p, a from (x, v) like the reference's init sequence (:604-607) — the checkpoint / resume
        semantics of SURVEY.md 5 — so the run is continued, not bit-continued.  Returns the new column ranges, or None
        when the largest slab would shrink by less than min_gain (nothing is touched then)."""
        if self.factory is None or self.prm is None:
            raise ValueError("SlabRunner.rebalance needs factory= and prm=")
        prm, world = self.prm, self.world
        cols = grid_columns(prm)
        owned = [s.read()[:2] for s in self.slabs]                                   # (particles, ids) per local slab
        hists = [np.bincount(np.clip(global_columns(prm, p["x"]), 0, cols - 1), minlength=cols).astype(np.int64) for p, _ in owned]
        per_slab_counts = []
        for k, (p, _) in enumerate(owned):
            v = np.zeros(world, np.int64)
            v[self.rank0 + k] = len(p)
            per_slab_counts.append(v)
        hist = np.asarray(self.transport.sum_over_ranks(hists)).reshape(-1)          # particles per global column, all ranks
        counts = np.asarray(self.transport.sum_over_ranks(per_slab_counts)).reshape(-1)      # owned particles per slab
        parts = _cuts_from_histogram(hist, world, 0, cols)
        cum = np.concatenate([[0], np.cumsum(hist)])
        new_counts = np.array([cum[c1] - cum[c0] for c0, c1 in parts])
        if counts.max() - new_counts.max() < min_gain * max(counts.max(), 1):
            return None
        outgoing = []
        for p, ids in owned:
            gc = global_columns(prm, p["x"])
            rec = np.zeros((len(p), self.RECW), np.float32)
            for j, fld in enumerate(("x", "y", "u", "v", "m", "rho", "p")):
                rec[:, j] = p[fld]
            rec[:, 7] = ids.astype(np.uint32).view(np.float32)
            outgoing.append([rec[(gc >= c0 - GHOST) & (gc < c1 + GHOST)] for c0, c1 in parts])
        incoming = self.transport.redistribute(outgoing)
        for s in self.slabs:
            s.close()
        new = []
        for k, rec in enumerate(incoming):
            r = self.rank0 + k
            c0, c1 = parts[r]
            loc = np.zeros(len(rec), owned[0][0].dtype)
            for j, fld in enumerate(("x", "y", "u", "v", "m", "rho", "p")):
                loc[fld] = rec[:, j]
            ids = np.ascontiguousarray(rec[:, 7]).view(np.uint32).copy()
            new.append(self.factory(c0, c1, r > 0, r < world - 1, loc, ids, gx, gy))
        self.slabs = new
        self.transport.rebind(new)
        return parts

    def gather_local(self, n_total, particle_dtype):
        """all slabs of THIS process merged back into original order (ids index the global arrays)."""
        out = np.zeros(n_total, particle_dtype)
        du, dv = np.zeros(n_total, np.float32), np.zeros(n_total, np.float32)
        seen = np.zeros(n_total, np.int32)
        for s in self.slabs:
            p, ids, a, b = s.read()
            out[ids], du[ids], dv[ids] = p, a, b
            seen[ids] += 1
        return out, du, dv, seen
