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

Halo buffer format (include/sph.h): uint32 header[4] = {count, kind, 0, 0} then `count` records of
5 words {x, y, u, v, id} (kind 0) or 4 words {x, y, u, v} (kind 1); fixed capacity, always sent
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

    def rebuilds(self):
        """rebuilds of the neighbour structure since creation (the same number on every slab)."""
        a, b = C.c_longlong(), C.c_longlong()
        self._chk(self.L.sph_rebuild_stats(self.h, C.byref(a), C.byref(b)))
        return a.value


class LocalTransport:
    """All slabs live in this process (one device): what slab r sends right is what slab r+1 receives left."""

    def __init__(self, slabs):
        self.slabs = slabs

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
        self.torch, self.dist, self.rank, self.world = torch, dist, rank, world
        self.send_l, self.send_r, self.recv_l, self.recv_r = slab.halo_tensors(torch, device)
        self.flag = slab.flag_tensor(torch, device)

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

    def __init__(self, slabs, transport, overlap=None):
        self.slabs = slabs if isinstance(slabs, (list, tuple)) else [slabs]
        self.transport = transport
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
