"""Z-slab decomposition of the per-scale feature path across the GPUs of one node.

The reference has no distributed code; this is the host-side
orchestration the MI355X build adds.  One process per GPU (torch.distributed; backend
"nccl" is RCCL over xGMI).  Rank r owns a contiguous range of planes of the volume (any
cut with at least 4 planes per rank) and the same planes of every output, so the
concatenation of the ranks' outputs in rank order IS the reference's voxel order.

Everything is slab-local except two things:

  * the Z recursion of ITK's recursive Gaussian, which runs the full length of every Z
    line.  It is kept exactly sequential by handing the recursion STATE across the slab
    boundaries (C-ABI `ife_stage_z_sweep` / `ife_stage_z_combine`): the causal chain
    travels rank 0 -> W-1, the anticausal chain W-1 -> 0, both at once, as records of
    4 doubles per line and (scale, field): the last four outputs of the recursion.  The
    input samples such a state refers to do not travel: a rank's slab of the RAW data is
    cut with an overlap of 3 planes below and 4 above (`overlap(rank, world)`), which every
    rank can take from wherever its data comes from, and the prepass runs over the
    overlap too.  Each rank then rebuilds both recursions of its slab from checkpoints.
    Stitched, this is bit for bit the single-device result.
  * the +-1 plane stencil of the feature kernel: one boundary plane of numerator and
    denominator per scale to each Z neighbour.

Per step and rank (S scales, nf = 2 fields with a mask, G line groups per scale):

  prepare                       tc = image*mask, cf = float(mask)                (local)
  boundary chains, items (scale group q, line group g) in expected-arrival order:
      chain stream: the direction whose state reaches this rank FIRST (causal in the lower half
      of the ranks, anticausal in the upper) as a lean sweep: [recv state] one recursion step
      per sample, checkpoints [send state];
      fused stream: the other direction FUSED with the combine: [recv state] that recursion
      carried through the slab, the first one rebuilt from its checkpoints, the Z output
      written [send state].
      Three recursion steps per sample and rank (round 2: two sweeps and a combine, four).
      Two streams, and `depth` sets of the per-step buffers: the fused sweeps of step t wait for
      the state that arrives last, the lean sweeps of steps t+1 .. t+depth-1 start the next
      chains meanwhile (_Streams).
  bulk stream, per scale group (all scales up to four ranks, else one scale):
      X pass, Y pass (stores numerator / denominator), halo exchange, features  (local)

Bytes over xGMI per boundary, direction and step: 32 B x nx*ny x S x nf (512^2, 3 scales,
2 fields: 50 MB) + one plane of the smoothed value per scale (3 MB); neighbours only.

The order of the sweeps on a rank is static: sorted by the hop count after which the state
can arrive (causal item i at rank r: r + i; anticausal: W-1-r + i; ties causal first).
Every dependency points to an event with a smaller key on the neighbour, so blocking waits
in that order cannot form a cycle -- the schedule is deadlock-free whether a wait blocks a
stream (RCCL) or the host (gloo in the tests).

Nothing here touches the oracle: `stages` is the C-ABI (HipStages).  Tests substitute
their own stage object to exercise this orchestration on CPU with gloo.
"""
import contextlib

import numpy as np

STATE_BYTES_PER_LINE = 32      # 4 doubles (include/ife_hip.h: IFE_Z_STATE_BYTES)
OVERLAP_LO, OVERLAP_HI = 3, 4  # neighbour planes in front of / behind a slab's input (IFE_Z_OVERLAP_*)


def overlap(rank, world):
    """(planes below, planes above) that rank's slab of the raw data carries beyond its own."""
    return (OVERLAP_LO if rank > 0 else 0, OVERLAP_HI if rank < world - 1 else 0)


def slab_bounds(nz, world):
    """Default cut: nz // world planes each, the remainder spread over the first ranks."""
    q, rem = divmod(nz, world)
    b = [0]
    for r in range(world):
        b.append(b[-1] + q + (1 if r < rem else 0))
    return b


def sweep_schedule(rank, world, n_items):
    """[(direction, item)] in the order this rank runs its sweeps (module docstring)."""
    ev = []
    for i in range(n_items):
        ev.append((rank + i, 0, i))
        ev.append((world - 1 - rank + i, 1, i))
    ev.sort()
    return [(d, i) for _, d, i in ev]


class HipStages:
    """Stage calls through the C-ABI on torch device tensors.  `ctx` runs the bulk work,
    `chain_ctx` (bound to another stream; may be the same context) the prepass and the lean
    boundary sweeps, `fused_ctx` (a third stream; default: chain_ctx) the fused sweeps."""

    def __init__(self, pkg, ctx, chain_ctx=None, fused_ctx=None):
        self.pkg, self.ctx, self.chain_ctx = pkg, ctx, chain_ctx or ctx
        self.fused_ctx = fused_ctx or self.chain_ctx

    def ck_bytes(self, slab_shape):
        return self.ctx.stage_z_ck_bytes(slab_shape)

    def prepare(self, img, mask, tc, cf):
        """Runs with the boundary sweeps (chain context): the next step's prepass and sweeps
        then overlap the bulk work of the current one."""
        pkg = self.pkg
        idt = pkg.F32 if img.element_size() == 4 else pkg.I16
        mdt = pkg.U8 if mask is None or mask.element_size() == 1 else pkg.U16
        self.chain_ctx.stage_prepare(img.data_ptr(), idt, mask.data_ptr() if mask is not None else None,
                               mdt, tuple(img.shape), tc.data_ptr(),
                               cf.data_ptr() if cf is not None else None, 1)

    def z_sweep(self, direction, srcs_ext, pad_lo, nzl, spacing, sigmas, line0, nlines, has_neighbour,
                state_in, state_out, cks):
        """srcs_ext: the fields WITH their overlap planes; the slab's own planes start at pad_lo."""
        own = [t[pad_lo:pad_lo + nzl] for t in srcs_ext]  # data_ptr() of a view is its first plane
        self.chain_ctx.stage_z_sweep(direction, [t.data_ptr() for t in own], tuple(own[0].shape),
                                     spacing, line0, nlines, sigmas, has_neighbour,
                                     state_in.data_ptr() if has_neighbour else None,
                                     state_out.data_ptr(), [c.data_ptr() for c in cks])

    def z_fused(self, direction, srcs_ext, pad_lo, nzl, dsts, spacing, sigmas, line0, nlines, has_lo, has_hi,
                state_in, state_out, cks):
        """The sweep of `direction` and the combine in one launch (C-ABI ife_stage_z_fused): for the
        direction whose state arrives last; needs the other direction's checkpoints in cks."""
        own = [t[pad_lo:pad_lo + nzl] for t in srcs_ext]
        has_nb = has_lo if direction == 0 else has_hi
        self.fused_ctx.stage_z_fused(direction, [t.data_ptr() for t in own], [t.data_ptr() for t in dsts],
                                     tuple(own[0].shape), spacing, line0, nlines, sigmas, has_lo, has_hi,
                                     state_in.data_ptr() if has_nb else None, state_out.data_ptr(),
                                     [c.data_ptr() for c in cks])

    def gaussian_quotient(self, nums, dens, dsts, spacing, axis, sigmas):
        """Last axis pass of the normalized convolution: dsts[j] = G(nums[j]) / G(dens[j])."""
        self.ctx.stage_recursive_gaussian_quotient([t.data_ptr() for t in nums], [t.data_ptr() for t in dens],
                                                   [t.data_ptr() for t in dsts], tuple(dsts[0].shape),
                                                   spacing, axis, sigmas)

    def z_combine(self, srcs_ext, pad_lo, nzl, dsts, spacing, sigmas, has_lo, has_hi, cks):
        own = [t[pad_lo:pad_lo + nzl] for t in srcs_ext]
        shape = tuple(own[0].shape)
        self.ctx.stage_z_combine([t.data_ptr() for t in own], [t.data_ptr() for t in dsts], shape,
                                 spacing, 0, shape[1] * shape[2], sigmas, has_lo, has_hi,
                                 [c.data_ptr() for c in cks])

    def gaussian_axis_batch(self, srcs, dsts, spacing, axis, sigmas):
        """One launch over several float volumes (jobs) shaped like dsts[0]."""
        self.ctx.stage_recursive_gaussian_batch([t.data_ptr() for t in srcs],
                                                [t.data_ptr() for t in dsts],
                                                tuple(dsts[0].shape), spacing, axis, sigmas, 1)

    def features(self, num, den, mask, slab_shape, spacing, halo_lo, halo_hi, out, layout):
        mdt = self.pkg.U8 if mask is None or mask.element_size() == 1 else self.pkg.U16
        self.ctx.stage_features(num.data_ptr(), den.data_ptr() if den is not None else None,
                                mask.data_ptr() if mask is not None else None, mdt, slab_shape,
                                spacing, halo_lo, halo_hi, out.data_ptr(), layout)


class _Xfer:
    """A pending point-to-point transfer; wait() blocks the current stream (RCCL) or the host
    (gloo) and then runs the completion hook (host staging copies back to the device)."""

    def __init__(self, work, after=None, keep=None):
        self.work, self.after, self.keep = work, after, keep

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        if self.after is not None:
            self.after()
            self.after = None
        self.keep = None


class TorchComm:
    """Neighbour exchanges over torch.distributed.  Every edge (r, r+1) has one communicator
    per traffic class -- causal states up, anticausal states down, stencil planes -- so that
    no class queues behind another on a shared communicator stream.  `host_staging` moves
    data through host memory (for backends that cannot take device tensors: gloo with the
    ranks of a test sharing one GPU)."""

    def __init__(self, dist, rank, world, host_staging=False, per_edge_groups=True, device=None):
        """device: where the exchanged buffers live (a torch.device for RCCL; None = host)."""
        self.dist, self.rank, self.world, self.host = dist, rank, world, host_staging
        self.up, self.down, self.halo_g = [], [], []
        # False: all traffic of an edge shares one communicator.  The engine then keeps the lean
        # and the fused sweeps in ONE stream: with a communicator per class the lean chain may
        # run steps ahead of the fused one, on a shared one that reorders the two ranks' operations
        # against each other and they block for good.
        self.per_class = True
        try:
            if not per_edge_groups:
                raise RuntimeError("disabled by the caller")
            for e in range(world - 1):  # collective: every rank creates every group, same order
                self.up.append(dist.new_group([e, e + 1]))
                self.down.append(dist.new_group([e, e + 1]))
                self.halo_g.append(dist.new_group([e, e + 1]))
        except Exception as exc:  # noqa: BLE001 -- a backend without sub-groups
            # Everything then shares the default group.  Still correct: every rank issues its
            # transfers of one direction of an edge in the same order (chain items, then stencil
            # planes), only a class may now queue behind another.
            import sys
            print("slab.TorchComm: per-edge communicators unavailable (%s); using the default group"
                  % exc, file=sys.stderr)
            self.up = self.down = self.halo_g = [None] * max(world - 1, 0)
            self.per_class = False
        self._handshake(device)

    def _handshake(self, device):
        """One tiny transfer per edge and traffic class, in an order no rank can block in:
        even edges first, then odd ones, so the two ranks of an edge always meet.  RCCL builds a
        communicator at its first use and that rendezvous blocks the host; left to the first
        step, rank r would sit in its first send up (waiting for r+1 to receive) while r+1 sits
        in its first send down (waiting for r).  Also leaves every communicator warm before
        bench.py starts its clock."""
        import torch
        dist, r, W = self.dist, self.rank, self.world
        dev = "cpu" if (self.host or device is None) else device
        tok = lambda v: torch.full((4,), v, dtype=torch.uint8, device=dev)
        for parity in (0, 1):
            for e in (r - 1, r):
                if e < 0 or e >= W - 1 or e % 2 != parity:
                    continue
                lower = r == e              # I am the lower rank of edge e
                peer = e + 1 if lower else e
                got_u, got_d, got_h = tok(0), tok(0), tok(0)
                if lower:
                    dist.isend(tok(1), peer, group=self.up[e]).wait()
                    dist.irecv(got_d, peer, group=self.down[e]).wait()
                else:
                    dist.irecv(got_u, peer, group=self.up[e]).wait()
                    dist.isend(tok(2), peer, group=self.down[e]).wait()
                for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, tok(3), peer, group=self.halo_g[e]),
                                                 dist.P2POp(dist.irecv, got_h, peer, group=self.halo_g[e])]):
                    w.wait()
                if dev != "cpu":
                    torch.cuda.synchronize(device)
                want = [(got_d, 2), (got_h, 3)] if lower else [(got_u, 1), (got_h, 3)]
                for t, v in want:
                    if not bool((t == v).all()):
                        raise RuntimeError("slab.TorchComm: handshake on edge %d returned wrong data" % e)

    def _isend(self, buf, dst, group):
        if self.host:
            h = buf.cpu()
            return _Xfer(self.dist.isend(h, dst, group=group), keep=h)
        return _Xfer(self.dist.isend(buf, dst, group=group))

    def _irecv(self, buf, src, group):
        if self.host:
            import torch
            h = torch.empty(buf.shape, dtype=buf.dtype, device="cpu")
            return _Xfer(self.dist.irecv(h, src, group=group), after=lambda: buf.copy_(h), keep=h)
        return _Xfer(self.dist.irecv(buf, src, group=group))

    # causal states travel up (r -> r+1), anticausal states down (r -> r-1)
    def isend_up(self, buf):
        return self._isend(buf, self.rank + 1, self.up[self.rank])

    def irecv_up(self, buf):
        return self._irecv(buf, self.rank - 1, self.up[self.rank - 1])

    def isend_down(self, buf):
        return self._isend(buf, self.rank - 1, self.down[self.rank - 1])

    def irecv_down(self, buf):
        return self._irecv(buf, self.rank + 1, self.down[self.rank])

    def halo(self, first_planes, last_planes, lo_halos, hi_halos):
        """Send my first planes to rank-1 (its hi halos) and my last planes to rank+1 (its lo
        halos); receive mine.  Lists of equal length (one entry per field and scale); edge
        ranks skip the missing side.  One batched call per neighbour."""
        dist = self.dist
        works, post = [], []
        for nb, snd, rcv in ((self.rank - 1, first_planes, lo_halos),
                             (self.rank + 1, last_planes, hi_halos)):
            if nb < 0 or nb >= self.world:
                continue
            g = self.halo_g[min(nb, self.rank)]
            ops = []
            for s, r in zip(snd, rcv):
                if self.host:
                    import torch
                    sh, rh = s.cpu(), torch.empty(r.shape, dtype=r.dtype, device="cpu")
                    post.append((r, rh, sh))
                    s, r = sh, rh
                ops += [dist.P2POp(dist.isend, s, nb, group=g), dist.P2POp(dist.irecv, r, nb, group=g)]
            works += dist.batch_isend_irecv(ops)
        for w in works:
            w.wait()
        for dst, src, _ in post:
            dst.copy_(src)


class NullComm:
    """Timing proxy only (bench.py --proxy-rank): the interface of TorchComm with every transfer
    left out, so that ONE GPU can run the local work of rank r of W -- neighbours on the sides
    that rank has, hence the interior forms of the Z kernels and the halo planes of the feature
    pass -- on whatever the state buffers hold.  The numbers it produces mean nothing."""

    per_class = True

    def isend_up(self, buf):
        return _Xfer(None)

    irecv_up = isend_down = irecv_down = isend_up

    def halo(self, first_planes, last_planes, lo_halos, hi_halos):
        pass


class _Streams:
    """The HIP streams of a rank on a GPU, or nothing on CPU: `bulk` (X, Y, features), `chain`
    (prepass and the lean sweeps), `fused` (the fused sweeps), `post[d]` (the receives of
    direction d are posted from it).  The lean and the fused sweeps have streams of their own
    because the fused sweeps of step t wait for the state that arrives LAST -- on the end ranks
    after W-1 hops -- and the lean sweeps of step t+1, which START the next chain, must not queue
    behind that wait: in one in-order stream the two end ranks would hand the chains back and
    forth and a step would take a whole chain latency however little work it holds."""

    def __init__(self, torch, dev, two_streams):
        self.torch = torch
        self.gpu = dev is not None and dev.type == "cuda"
        self.bulk = torch.cuda.current_stream(dev) if self.gpu else None
        self.chain = (torch.cuda.Stream(dev, priority=-1) if two_streams else self.bulk) if self.gpu else None
        self.fused = (torch.cuda.Stream(dev, priority=-1) if two_streams else self.bulk) if self.gpu else None
        # one per direction: a stream of event waits is in order too, and the receive of a lean
        # sweep must not stand behind the wait for a fused sweep of the step before
        self.post = [torch.cuda.Stream(dev) if self.gpu and two_streams else self.bulk for _ in range(2)]

    def on(self, stream):
        return self.torch.cuda.stream(stream) if self.gpu else contextlib.nullcontext()

    def record(self, stream):
        if not self.gpu:
            return None
        e = self.torch.cuda.Event()
        e.record(stream)
        return e

    def wait(self, stream, event):
        if self.gpu and event is not None:
            stream.wait_event(event)


class SlabEngine:
    """Runs all scales of the feature path on this rank's Z-slab."""

    def __init__(self, stages, comm, shape_zyx, spacing, sigmas, rank, world, alloc, layout,
                 has_mask=True, line_groups=None, bounds=None, streams=None,
                 scales_per_item=None, depth=4):
        """alloc(shape, dtype_name) -> tensor ('float32' or 'uint8') on the compute device;
        bounds: W+1 plane indices (default: slab_bounds); line_groups: items per scale group on
        the boundary chains (default 4); scales_per_item: scales whose sweeps share a launch and a
        message (default: all of them, up to the four a launch takes -- the jobs of a launch share
        their input through L2 and the 64-plane launches of one scale leave most of the device
        idle: 512^3 on a 64-plane slab 1.39 ms per step against 1.43 with one scale per item);
        streams: a _Streams (GPU); depth: sets of the per-step buffers (samples with overlap,
        checkpoints, Z output), i.e. how many steps the lean chain may run ahead of the bulk work:
        the chain of a step needs W-1 hops from end to end, and with d sets a step can be as short
        as that latency / d (eight ranks: ~2.4 ms predicted at 70 GB/s per link, 3.4 at 45; local
        work 1.33 ms; scripts/experiments/slab_chain_sim.py: 3 sets suffice at 70 GB/s, 4 at 45)."""
        if depth < 2:
            raise ValueError("depth must be at least 2")
        self.depth = int(depth)
        nz, ny, nx = shape_zyx
        self.bounds = list(bounds) if bounds is not None else slab_bounds(nz, world)
        if len(self.bounds) != world + 1 or self.bounds[0] != 0 or self.bounds[-1] != nz:
            raise ValueError("bounds must run from 0 to nz in world+1 steps")
        if min(b1 - b0 for b0, b1 in zip(self.bounds, self.bounds[1:])) < 4:
            raise ValueError("every Z-slab needs at least 4 planes (nz=%d over %d ranks: %s)"
                             % (nz, world, self.bounds))
        if min(ny, nx) < 4:
            raise ValueError("the recursive Gaussian needs at least 4 voxels along every axis")
        self.st, self.comm, self.sync = stages, comm, streams
        self.nz, self.ny, self.nx, self.W, self.rank = nz, ny, nx, world, rank
        self.z0, self.z1 = self.bounds[rank], self.bounds[rank + 1]
        self.nzl = nzl = self.z1 - self.z0
        self.spacing, self.sigmas, self.layout = tuple(spacing), list(sigmas), layout
        self.has_mask = has_mask
        self.nf = nf = 2 if has_mask else 1
        S = len(self.sigmas)
        L = ny * nx
        # line groups pipeline a boundary: the transfer of one group travels while the next is
        # swept, and a chain's start-up (W-1 hops of kernel + wire + latency) shrinks with the
        # item.  Four groups carrying all scales at every world size: with 50 MB per boundary and
        # direction (512^2, 3 scales, 2 fields) an item is 12.6 MB, a chain of eight ranks takes
        # 7 x (0.18 ms of wire at 70 GB/s + kernel + latency) + 3 x 0.18 = 2.4 ms from its first
        # sweep to its last state -- hidden behind the `depth` steps the engine keeps in flight -- at
        # 16 point-to-point calls and 8 chain launches per rank and step.  (Round 2 ran eight
        # ranks with one scale per item and three groups, 36 calls: the host thread then needs
        # about as long per step as the device, scripts/experiments/slab_p2p_host_time.py.)
        G = line_groups if line_groups is not None else (1 if world == 1 else 4)
        G = max(1, min(G, (L + 255) // 256))
        per = ((L + G - 1) // G + 255) // 256 * 256  # whole workgroups of 256 lines
        self.groups = [(l0, min(L, l0 + per) - l0) for l0 in range(0, L, per)]
        max_jobs = max(1, 8 // nf)                     # jobs of one launch (IIR_MAX_JOBS = 8)
        spi = scales_per_item if scales_per_item else S
        spi = max(1, min(spi, max_jobs, S))
        # scale groups: their sweeps share launches and messages; the bulk phase follows them
        self.scale_groups = [list(range(s0, min(S, s0 + spi))) for s0 in range(0, S, spi)]
        self.items = [(q, g) for q in range(len(self.scale_groups)) for g in range(len(self.groups))]
        # the bulk phase follows the chains group by group, except that everything behind the
        # first group is merged into launches of up to `max_jobs` jobs: a one-scale launch on a
        # thin slab fills a third of the device (64 planes: 0.122 ms per scale against 0.29 for three)
        self.bulk_groups = [[0]]
        for q in range(1, len(self.scale_groups)):
            last = self.bulk_groups[-1]
            n_sc = sum(len(self.scale_groups[k]) for k in last) + len(self.scale_groups[q])
            if len(self.bulk_groups) > 1 and n_sc <= max_jobs:
                last.append(q)
            else:
                self.bulk_groups.append([q])
        self.schedule = sweep_schedule(rank, world, len(self.items))
        f = lambda *shp: alloc(shp, "float32")
        self.pad_lo, self.pad_hi = overlap(rank, world)
        # tc, cf with overlap, and the checkpoints: `depth` sets, used by successive steps in
        # turn, so that the prepass and the lean sweeps of the next steps (chain stream) may run
        # while the fused sweeps and the bulk work of step t still read their own -- in a stream of
        # volumes the chains' start-up (W-1 hops) then hides behind the previous volumes' work
        D = self.depth
        self.src = [[f(self.pad_lo + nzl + self.pad_hi, ny, nx) for _ in range(nf)] for _ in range(D)]
        # Z-pass output: written on the fused stream while the X pass of an earlier step may
        # still read its own, so it rotates between steps like src and ck
        self.zo = [[[f(nzl, ny, nx) for _ in range(nf)] for _ in range(S)] for _ in range(D)]
        self.xo = [[f(nzl, ny, nx) for _ in range(nf)] for _ in range(S)]   # X-pass output
        self.pad = [f(nzl + 2, ny, nx) for _ in range(S)]  # smoothed value S (Y output) + halo planes
        # the direction whose state reaches this rank first runs as a lean sweep, the other fused
        self.lean = 0 if rank <= world - 1 - rank else 1
        # run() relies on it: the lean sweep of an item stands before its fused sweep
        order = {di: k for k, di in enumerate(self.schedule)}
        assert all(order[(self.lean, i)] < order[(1 - self.lean, i)] for i in range(len(self.items)))
        ckb = stages.ck_bytes((nzl, ny, nx))
        self.ck = [[[alloc((ckb,), "uint8") for _ in range(nf)] for _ in range(S)] for _ in range(D)]
        self.step = 0
        self.free = [None] * D    # events (bulk stream): the X pass that read zo[p] has run
        self.fdone = [None] * D   # events (fused stream): the fused sweeps that read src[p] / ck[p] have run
        sb = lambda q, g: alloc((len(self.scale_groups[q]) * nf * STATE_BYTES_PER_LINE
                                 * self.groups[g][1],), "uint8")
        self.c_in = [sb(q, g) for q, g in self.items]
        self.c_out = [sb(q, g) for q, g in self.items]
        self.a_in = [sb(q, g) for q, g in self.items]
        self.a_out = [sb(q, g) for q, g in self.items]
        n = len(self.items)
        self.sent = [[None] * n, [None] * n]      # pending sends of the previous step
        self.consumed = [[None] * n, [None] * n]  # events: in-state read by its sweep

    # ---- one step -------------------------------------------------------------------
    def run(self, img_slab, mask_slab, out):
        """img_slab [pad_lo + nzl + pad_hi][ny][nx] f32|i16: the rank's planes of the raw volume
        with the overlap of `overlap(rank, world)`; mask_slab the same planes, u8|u16, or None;
        out [S][nzl][ny][nx][8] (or [S][8][nzl][ny][nx] planar) float32: the rank's own planes."""
        if img_slab.shape[0] != self.pad_lo + self.nzl + self.pad_hi:
            raise ValueError("the slab of rank %d needs %d + %d + %d planes (overlap below, own, above), got %d"
                             % (self.rank, self.pad_lo, self.nzl, self.pad_hi, img_slab.shape[0]))
        st, comm, sy = self.st, self.comm, self.sync
        nf, sp, W, r = self.nf, self.spacing, self.W, self.rank
        has_lo, has_hi = r > 0, r < W - 1
        slab_shape = (self.nzl, self.ny, self.nx)
        on = sy.on if sy is not None else (lambda s: contextlib.nullcontext())
        chain = sy.chain if sy is not None else None
        fstream = sy.fused if sy is not None else None
        bulk = sy.bulk if sy is not None else None
        post = sy.post if sy is not None else (None, None)
        rec = (lambda s: sy.record(s)) if sy is not None else (lambda s: None)
        wait = (lambda s, e: sy.wait(s, e)) if sy is not None else (lambda s, e: None)
        n = len(self.items)

        par = self.step % self.depth
        self.step += 1
        src, ck, zo = self.src[par], self.ck[par], self.zo[par]

        # A receive is ENQUEUED where its sweep stands in the schedule, never earlier: the
        # schedule is a topological order of the whole job (every dependency has a smaller key
        # on the neighbour), so even if the runtime ran all streams of a rank through one
        # in-order hardware queue -- HIP shares a few queues among all streams -- a receive
        # that spins for its sender can only hold back work that comes after it in that order.
        # It is issued from the direction's `post` stream, which holds nothing but event waits: the
        # communicator's stream then waits for the sweep that read the buffer in the previous
        # step and not for the sweeps queued on the chain stream, so the transfer of item i+1
        # still travels under the sweep of item i.
        swept = [[None] * n, [None] * n]
        with on(chain):
            # this set was last used `depth` steps ago: its fused sweeps have read src and ck
            # (its lean sweeps stand earlier in this very stream)
            wait(chain, self.fdone[par])
            st.prepare(img_slab, mask_slab if self.has_mask else None, src[0],
                       src[1] if self.has_mask else None)
        wait(fstream, self.free[par])  # ... and its X pass has read the Z output
        last_fused = None
        for d, i in self.schedule:
            q, g = self.items[i]
            ss = self.scale_groups[q]
            l0, nl = self.groups[g]
            has_nb = has_lo if d == 0 else has_hi
            sin = (self.c_in if d == 0 else self.a_in)[i]
            sout = (self.c_out if d == 0 else self.a_out)[i]
            mine = chain if d == self.lean else fstream
            with on(mine):
                if has_nb:
                    with on(post[d]):
                        wait(post[d], self.consumed[d][i])
                        rx = comm.irecv_up(sin) if d == 0 else comm.irecv_down(sin)
                    rx.wait()   # this sweep's stream (RCCL) or the host (gloo) waits for the state
                if self.sent[d][i] is not None:  # last step's send still reads sout
                    self.sent[d][i].wait()
                    self.sent[d][i] = None
                srcs = [src[k] for _ in ss for k in range(nf)]
                sgs = [self.sigmas[s] for s in ss for _ in range(nf)]
                cks = [ck[s][k] for s in ss for k in range(nf)]
                if d == self.lean:
                    st.z_sweep(d, srcs, self.pad_lo, self.nzl, sp, sgs, l0, nl, has_nb, sin, sout, cks)
                else:  # its state arrives last: carried through the slab, the other from checkpoints
                    # the lean sweep of the same item (it stands earlier in the schedule) has left
                    # its checkpoints, and with it the prepass has run
                    wait(fstream, swept[self.lean][i])
                    st.z_fused(d, srcs, self.pad_lo, self.nzl, [zo[s][k] for s in ss for k in range(nf)],
                               sp, sgs, l0, nl, has_lo, has_hi, sin, sout, cks)
                swept[d][i] = rec(mine)
                self.consumed[d][i] = swept[d][i]
                if d != self.lean:
                    last_fused = swept[d][i]
                if d == 0 and has_hi:
                    self.sent[d][i] = comm.isend_up(sout)
                elif d == 1 and has_lo:
                    self.sent[d][i] = comm.isend_down(sout)
        self.fdone[par] = last_fused

        first = 0 if has_lo else 1
        fat = 1 - self.lean
        for qs in self.bulk_groups:
            ss = [s for q in qs for s in self.scale_groups[q]]
            for i, (qi, g) in enumerate(self.items):
                if qi in qs:
                    wait(bulk, swept[fat][i])  # the fused kernels have written this group's Z output
            sg = [self.sigmas[s] for s in ss for _ in range(nf)]
            jobs = lambda bufs: [bufs[s][k] for s in ss for k in range(nf)]
            st.gaussian_axis_batch(jobs(zo), jobs(self.xo), sp, 0, sg)
            if qs is self.bulk_groups[-1]:
                self.free[par] = rec(bulk)  # the last reader of this step's Z output
            own = [self.pad[s][1:self.nzl + 1] for s in ss]
            if nf == 2:  # the last pass stores numerator / denominator: one field from here on
                st.gaussian_quotient([self.xo[s][0] for s in ss], [self.xo[s][1] for s in ss], own, sp, 1,
                                     [self.sigmas[s] for s in ss])
            else:
                st.gaussian_axis_batch([self.xo[s][0] for s in ss], own, sp, 1, [self.sigmas[s] for s in ss])
            pads = [self.pad[s] for s in ss]
            comm.halo([p[1] for p in pads], [p[self.nzl] for p in pads],
                      [p[0] for p in pads], [p[self.nzl + 1] for p in pads])
            for s in ss:
                st.features(self.pad[s][first:], None,
                            mask_slab[self.pad_lo:self.pad_lo + self.nzl] if self.has_mask else None,
                            slab_shape, sp,
                            1 if has_lo else 0, 1 if has_hi else 0, out[s], self.layout)

    def finish(self):
        """Wait for the sends of the last step (call before tearing the process group down)."""
        for d in (0, 1):
            for i, x in enumerate(self.sent[d]):
                if x is not None:
                    x.wait()
                    self.sent[d][i] = None


class SlabRunner:
    """bench.py's N>1 runner: builds this rank's slab of the synthetic volume in HBM and
    steps the engine."""

    def __init__(self, pkg, synth, shape, sigmas, seed, mask_kind, layout, rank, world, dev, args):
        import torch
        import torch.distributed as dist
        proxy = getattr(args, "proxy_world", None)
        if proxy:  # one GPU standing in for rank `proxy_rank` of `proxy_world`: local work only
            rank, world = int(getattr(args, "proxy_rank", 0) or 0), int(proxy)
        nz, ny, nx = shape
        bounds = slab_bounds(nz, world)
        z0, nzl = bounds[rank], bounds[rank + 1] - bounds[rank]
        i16 = bool(getattr(args, "i16", False))
        spacing = tuple(getattr(args, "spacing", (1.0, 1.0, 1.0)))
        lo, hi = overlap(rank, world)  # the raw slab carries the neighbours' adjacent planes
        ze, nze = z0 - lo, lo + nzl + hi
        img = (synth.volume_i16 if i16 else synth.volume_f32)((nze, ny, nx), seed, z0=ze)
        if mask_kind == "ellipsoids":
            mask = np.minimum(synth.mask_ellipsoids((nze, ny, nx), z0=ze, nz_total=nz), 1)
            mask = mask.astype(np.uint8)
        else:
            mask = np.ones((nze, ny, nx), np.uint8)
        self.d_img = torch.from_numpy(img).to(dev)
        self.d_mask = None if mask_kind == "none" else torch.from_numpy(mask).to(dev)
        oshape = ((len(sigmas), nzl, ny, nx, 8) if layout == pkg.INTERLEAVED
                  else (len(sigmas), 8, nzl, ny, nx))
        self.d_out = torch.empty(oshape, dtype=torch.float32, device=dev)
        # the one-rank form (bench.py --force-slab: a rank's local work without neighbours) keeps
        # the two streams of a real rank
        import os
        two = (world > 1 or bool(getattr(args, "force_slab", False))) and not os.environ.get("IFE_SLAB_ONE_STREAM")
        comm = NullComm() if proxy else TorchComm(dist, rank, world, device=dev)
        self.streams = _Streams(torch, dev, two_streams=two)  # IFE_SLAB_ONE_STREAM: diagnostics (clean per-kernel times)
        # Two ranks: a chain is one hop long, nothing to run ahead of -- and lean and fused sweeps
        # running side by side cost the local work 5-7 % (interior rank of eight: 1.46-1.50 ms per
        # step against 1.38 in one stream; stream priorities make no difference), which only a
        # longer chain pays back (simulated: 2.65 -> 1.45 ms per step at eight ranks, 2.9 -> 2.65 at four).
        if not comm.per_class or world <= 2 or os.environ.get("IFE_SLAB_ONE_CHAIN"):  # (the variable: diagnostics)
            self.streams.fused = self.streams.chain
        self.ctx = pkg.Context(dev.index or 0)
        self.ctx.set_stream(self.streams.bulk.cuda_stream)
        self.chain_ctx = self.fused_ctx = self.ctx
        if self.streams.chain is not self.streams.bulk:
            self.chain_ctx = pkg.Context(dev.index or 0)
            self.chain_ctx.set_stream(self.streams.chain.cuda_stream)
            self.fused_ctx = pkg.Context(dev.index or 0)
            self.fused_ctx.set_stream(self.streams.fused.cuda_stream)
        for c in self.contexts():
            c.set_option(pkg.OPT_TRIG_MODE, args.trig)
            if getattr(args, "iir_block", None):
                c.set_option(pkg.OPT_IIR_BLOCK, args.iir_block)
            if getattr(args, "iir_ckpt", None):
                c.set_option(pkg.OPT_IIR_CKPT, args.iir_ckpt)
            if getattr(args, "iir_fma", False):
                c.set_option(pkg.OPT_IIR_FMA, 1)
            if getattr(args, "zchunk", None):
                c.set_option(pkg.OPT_ZCHUNK, args.zchunk)
            c.set_option(pkg.OPT_CONST_LINES, getattr(args, "const_lines", 0))
        dt = {"float32": torch.float32, "uint8": torch.uint8}
        alloc = lambda shp, d: torch.empty(shp, dtype=dt[d], device=dev)
        self.engine = SlabEngine(HipStages(pkg, self.ctx, self.chain_ctx, self.fused_ctx),
                                 comm, shape, spacing, sigmas, rank, world,
                                 alloc, layout, has_mask=self.d_mask is not None,
                                 streams=self.streams,
                                 line_groups=getattr(args, "line_groups", None),
                                 scales_per_item=getattr(args, "scales_per_item", None),
                                 depth=getattr(args, "slab_depth", None) or 4)
        # what was actually built, for the bench line
        if proxy:  # the state buffers are never received into: give them finite contents
            for b in self.engine.c_in + self.engine.a_in:
                b.zero_()
        self.config = {"input": "int16" if i16 else "float32", "spacing": list(spacing),
                       "slab_planes": nzl, "line_groups": len(self.engine.groups), "depth": self.engine.depth,
                       "scales_per_item": len(self.engine.scale_groups[0])}

    def step(self):
        self.engine.run(self.d_img, self.d_mask, self.d_out)

    def contexts(self):
        return list({id(c): c for c in (self.ctx, self.chain_ctx, self.fused_ctx)}.values())

    def finish(self):
        self.engine.finish()
