"""Z-slab decomposition of the per-scale feature path across the GPUs of one node.

The reference has no distributed code; this is the host-side
orchestration the MI355X build adds.  One process per GPU (torch.distributed; backend
"nccl" is RCCL over xGMI).  Rank g owns planes [g*nz/W, (g+1)*nz/W) of the volume and its
slab of every output, so the concatenation of the ranks' outputs in rank order IS the
reference's voxel order.

Per step (all scales):
  exchange #0      all-to-all: Z-slabs -> Y-slabs of the raw image and mask (once per step)
  prepare          Y-slab   tc = image*mask, cf = float(mask)              (local)
  per scale:
    Z pass         Y-slab   every Z line is whole inside a Y-slab          (local)
    exchange #1    all-to-all: Y-slabs -> Z-slabs of the Z-pass output     (per field)
    X, Y passes    Z-slab   lines are slab-local                           (local)
    exchange #2    one boundary plane of num and den to each Z neighbour   (halo)
    features       Z-slab   divide + gradient + Hessian + eigen + mask     (local)

The Z recursion of ITK's recursive Gaussian runs the full length of every Z line, so it
is the one stage a Z-slab cut cannot keep local.  Re-cutting the two smoothing inputs
along Y for that pass keeps the arithmetic exactly the sequential recursion of the
single-GPU path (results are bit-identical to it), at the price of moving 8 B/voxel/scale
across xGMI; each rank talks to all W-1 peers at once, so all seven links carry
traffic.  The Z pass of scale 0 runs first and its exchange
starts at once; the Z passes of the other scales run batched in one launch (a slab has few
lines per field) behind it, and their exchanges travel while scale 0 computes.

Nothing here touches the oracle: `stages` is the C-ABI (HipStages).  Tests substitute
their own stage object to exercise this orchestration on CPU with gloo.
"""
import numpy as np


class HipStages:
    """Stage calls through the C-ABI on torch device tensors."""

    def __init__(self, pkg, ctx):
        self.pkg, self.ctx = pkg, ctx

    def prepare(self, img, mask, tc, cf, y_chunks):
        """tc/cf are written in all-to-all send order [y_chunks][nzl][ny/y_chunks][nx]."""
        pkg = self.pkg
        idt = pkg.F32 if img.element_size() == 4 else pkg.I16
        mdt = pkg.U8 if mask is None or mask.element_size() == 1 else pkg.U16
        self.ctx.stage_prepare(img.data_ptr(), idt, mask.data_ptr() if mask is not None else None,
                               mdt, tuple(img.shape), tc.data_ptr(),
                               cf.data_ptr() if cf is not None else None, y_chunks)

    def gaussian_axis_batch(self, srcs, dsts, spacing, axis, sigmas, in_y_chunks=1):
        """One launch over several float volumes (jobs) shaped like dsts[0].  With
        in_y_chunks = W the sources are [W][nzl][ny/W][nx] as an all-to-all left them."""
        self.ctx.stage_recursive_gaussian_batch([t.data_ptr() for t in srcs],
                                                [t.data_ptr() for t in dsts],
                                                tuple(dsts[0].shape), spacing, axis, sigmas,
                                                in_y_chunks)

    def features(self, num, den, mask, slab_shape, spacing, halo_lo, halo_hi, out, layout):
        mdt = self.pkg.U8 if mask is None or mask.element_size() == 1 else self.pkg.U16
        self.ctx.stage_features(num.data_ptr(), den.data_ptr() if den is not None else None,
                                mask.data_ptr() if mask is not None else None, mdt, slab_shape,
                                spacing, halo_lo, halo_hi, out.data_ptr(), layout)


class TorchComm:
    """The two exchanges, over torch.distributed.  `host_staging` moves data through host
    memory (for backends that cannot take device tensors, e.g. gloo in the tests)."""

    def __init__(self, dist, rank, world, host_staging=False):
        self.dist, self.rank, self.world, self.host = dist, rank, world, host_staging
        # The halo planes travel on their own communicator: on the default one they would
        # queue behind the all-to-alls of the later scales, which are issued up front.
        self.halo_group = dist.new_group(list(range(world))) if world > 1 else None

    def all_to_all(self, send, recv, async_op=False):
        """send/recv: contiguous [world, ...]; chunk h of send goes to rank h."""
        dist = self.dist
        if self.host:
            s, r = send.cpu(), recv.cpu()
            ops = []
            for h in range(self.world):
                if h == self.rank:
                    r[h].copy_(s[h])
                else:
                    ops.append(dist.P2POp(dist.isend, s[h], h))
                    ops.append(dist.P2POp(dist.irecv, r[h], h))
            for w in (dist.batch_isend_irecv(ops) if ops else []):
                w.wait()
            recv.copy_(r)
            return None
        return dist.all_to_all_single(recv, send, async_op=async_op)

    def halo(self, first_plane, last_plane, lo_halo, hi_halo):
        """Send my first plane to rank-1 (its hi halo) and my last plane to rank+1 (its lo
        halo); receive mine.  Edge ranks skip the missing side."""
        dist = self.dist
        ops, post = [], []
        lo, hi = self.rank - 1, self.rank + 1
        if self.host:
            fp, lp = first_plane.cpu(), last_plane.cpu()
            lo_b, hi_b = lo_halo.cpu(), hi_halo.cpu()
        else:
            fp, lp, lo_b, hi_b = first_plane, last_plane, lo_halo, hi_halo
        g = self.halo_group
        if lo >= 0:
            ops += [dist.P2POp(dist.isend, fp, lo, group=g), dist.P2POp(dist.irecv, lo_b, lo, group=g)]
            post.append((lo_halo, lo_b))
        if hi < self.world:
            ops += [dist.P2POp(dist.isend, lp, hi, group=g), dist.P2POp(dist.irecv, hi_b, hi, group=g)]
            post.append((hi_halo, hi_b))
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        if self.host:
            for dst, src in post:
                dst.copy_(src)


class SlabEngine:
    """Runs all scales of the feature path on this rank's Z-slab."""

    def __init__(self, stages, comm, shape_zyx, spacing, sigmas, rank, world, empty, layout,
                 has_mask=True, overlap=True):
        nz, ny, nx = shape_zyx
        if nz % world or ny % world:
            raise ValueError("nz=%d and ny=%d must be multiples of the number of slabs %d"
                             % (nz, ny, world))
        if min(nz, ny, nx) < 4:
            raise ValueError("the recursive Gaussian needs at least 4 voxels along every axis")
        self.st, self.comm = stages, comm
        self.nz, self.ny, self.nx, self.W, self.rank = nz, ny, nx, world, rank
        self.nzl, self.nyl = nz // world, ny // world
        self.spacing, self.sigmas, self.layout = tuple(spacing), list(sigmas), layout
        self.has_mask, self.overlap = has_mask, overlap
        nzl, nyl, W = self.nzl, self.nyl, world
        f = lambda *shp: empty(shp)  # float32 buffers
        nf = 2 if has_mask else 1
        S = len(self.sigmas)
        self.group = 4 if nf == 2 else 8              # scales per line-kernel launch (<= 8 jobs)
        if nyl % 64 and world > 1:
            raise ValueError("ny/world = %d must be a multiple of 64 (wave-aligned Y chunks)" % nyl)
        self.raw = None                                           # raw image / mask exchange buffers
        self.src_y = [f(W, nzl, nyl, nx) for _ in range(nf)]      # tc, cf (Y-slab == [nz][nyl][nx])
        # per scale: Z-pass output (Y-slab) and its Y-chunked Z-slab image after exchange #1
        self.zy = [[f(W, nzl, nyl, nx) for _ in range(nf)] for _ in range(S)]
        self.zz = [[f(W, nzl, nyl, nx) for _ in range(nf)] for _ in range(S)]
        self.b = [f(nzl, ny, nx) for _ in range(nf)]              # X-pass output
        self.pad = [f(nzl + 2, ny, nx) for _ in range(nf)]        # Y-pass output + halo planes

    def run(self, img_slab, mask_slab, out):
        """img_slab [nzl][ny][nx] f32|i16, mask_slab same shape u8|u16 or None,
        out [S][nzl][ny][nx][8] (or [S][8][nzl][ny][nx] planar) float32."""
        st, comm = self.st, self.comm
        nf = 2 if self.has_mask else 1
        yshape = (self.nz, self.nyl, self.nx)
        sp = self.spacing
        S = len(self.sigmas)
        # exchange #0 moves the RAW slab (image 4 or 2 B + mask 1 or 2 B per voxel, not the
        # 8 B of two float fields) re-cut along Y; Cast + Multiply then run on the Y-slab
        if self.raw is None:
            import torch
            mk = lambda t: (torch.empty((self.W, self.nzl, self.nyl, self.nx), dtype=t.dtype,
                                        device=t.device),
                            torch.empty((self.W, self.nzl, self.nyl, self.nx), dtype=t.dtype,
                                        device=t.device))
            self.raw = [mk(img_slab), mk(mask_slab) if self.has_mask else None]
        for t, bufs in ((img_slab, self.raw[0]), (mask_slab if self.has_mask else None, self.raw[1])):
            if t is None:
                continue
            send, recv = bufs
            send.copy_(t.view(self.nzl, self.W, self.nyl, self.nx).permute(1, 0, 2, 3))
            comm.all_to_all(send, recv)
        st.prepare(self.raw[0][1].view(yshape),
                   self.raw[1][1].view(yshape) if self.has_mask else None,
                   self.src_y[0].view(yshape), self.src_y[1].view(yshape) if self.has_mask else None, 1)
        # Z passes and exchanges #1.  Scale 0 goes alone so that its exchange is on the wire
        # while the remaining scales run their Z pass in ONE launch (a slab has few lines per
        # field: one job per launch would leave most of the device idle); the exchanges of
        # scales 1.. then travel while scale 0 runs its X/Y/feature kernels.
        groups = [[0]] + [list(range(s0, min(S, s0 + self.group)))
                          for s0 in range(1, S, self.group)]
        pending = [None] * S
        for ss in groups:
            st.gaussian_axis_batch([self.src_y[k].view(yshape) for s in ss for k in range(nf)],
                                   [self.zy[s][k].view(yshape) for s in ss for k in range(nf)],
                                   sp, 2, [self.sigmas[s] for s in ss for k in range(nf)])
            for s in ss:
                pending[s] = [comm.all_to_all(self.zy[s][k], self.zz[s][k],
                                              async_op=self.overlap) for k in range(nf)]
        lo = 1 if self.rank > 0 else 0
        hi = 1 if self.rank < self.W - 1 else 0
        first = 0 if lo else 1
        for s in range(S):                                    # X, Y, halo, features of scale s
            for w in pending[s]:
                if w is not None:
                    w.wait()
            sg = [self.sigmas[s]] * nf
            # the x pass reads the Y-chunked buffers the exchange left and writes plain slabs
            st.gaussian_axis_batch(self.zz[s][:nf], self.b[:nf], sp, 0, sg, self.W)
            st.gaussian_axis_batch(self.b[:nf], [p[1:self.nzl + 1] for p in self.pad[:nf]], sp, 1, sg)
            for k in range(nf):                               # exchange #2
                p = self.pad[k]
                comm.halo(p[1], p[self.nzl], p[0], p[self.nzl + 1])
            st.features(self.pad[0][first:], self.pad[1][first:] if self.has_mask else None,
                        mask_slab if self.has_mask else None, (self.nzl, self.ny, self.nx), sp,
                        lo, hi, out[s], self.layout)


class SlabRunner:
    """bench.py's N>1 runner: builds this rank's slab of the synthetic volume in HBM and
    steps the engine."""

    def __init__(self, pkg, synth, shape, sigmas, seed, mask_kind, layout, rank, world, dev, args):
        import torch
        import torch.distributed as dist
        nz, ny, nx = shape
        nzl = nz // world
        z0 = rank * nzl
        img = synth.volume_f32((nzl, ny, nx), seed, z0=z0)
        if mask_kind == "ellipsoids":
            mask = np.minimum(synth.mask_ellipsoids((nzl, ny, nx), z0=z0, nz_total=nz), 1)
            mask = mask.astype(np.uint8)
        else:
            mask = np.ones((nzl, ny, nx), np.uint8)
        self.d_img = torch.from_numpy(img).to(dev)
        self.d_mask = None if mask_kind == "none" else torch.from_numpy(mask).to(dev)
        oshape = ((len(sigmas), nzl, ny, nx, 8) if layout == pkg.INTERLEAVED
                  else (len(sigmas), 8, nzl, ny, nx))
        self.d_out = torch.empty(oshape, dtype=torch.float32, device=dev)
        self.ctx = pkg.Context(dev.index or 0)
        self.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        self.ctx.set_option(pkg.OPT_TRIG_MODE, args.trig)
        if args.iir_block:
            self.ctx.set_option(pkg.OPT_IIR_BLOCK, args.iir_block)
        if args.zchunk:
            self.ctx.set_option(pkg.OPT_ZCHUNK, args.zchunk)
        empty = lambda shp: torch.empty(shp, dtype=torch.float32, device=dev)
        self.engine = SlabEngine(HipStages(pkg, self.ctx), TorchComm(dist, rank, world), shape,
                                 (1.0, 1.0, 1.0), sigmas, rank, world, empty, layout,
                                 has_mask=self.d_mask is not None)

    def step(self):
        self.engine.run(self.d_img, self.d_mask, self.d_out)
