"""Host-side cost of one engine step INCLUDING its point-to-point calls, priced against gloo:
W ranks on CPU, the real SlabEngine and TorchComm (the branch RCCL takes: tensors handed to
isend / irecv directly), stage calls replaced by no-ops and tiny fields (8 KB states), so that
what is timed is Python + torch.distributed bookkeeping per step and rank, not arithmetic or
wire time.  Usage: python scripts/experiments/slab_p2p_host_time.py [W ...]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
PKG = "image-feature-extraction_amd"


class NullStages:
    def ck_bytes(self, shape): return 8
    def prepare(self, *a): pass
    def z_sweep(self, *a): pass
    def z_fused(self, *a): pass
    def gaussian_axis_batch(self, *a): pass
    def gaussian_quotient(self, *a): pass
    def features(self, *a): pass


def worker(rank, world, port, steps, q):
    import torch, torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = importlib.import_module(PKG)
    slab = importlib.import_module(PKG + ".slab")
    shape = (8 * world, 64, 64)
    dt = {"float32": torch.float32, "uint8": torch.uint8}
    alloc = lambda shp, d: torch.zeros(shp, dtype=dt[d])
    eng = slab.SlabEngine(NullStages(), slab.TorchComm(dist, rank, world), shape, (1, 1, 1), [1.0, 2.0, 4.0],
                          rank, world, alloc, pkg.INTERLEAVED, has_mask=True)
    lo, hi = slab.overlap(rank, world)
    img = torch.zeros((lo + 8 + hi, 64, 64)); mask = torch.ones((lo + 8 + hi, 64, 64), dtype=torch.uint8)
    out = torch.zeros((3, 8, 64, 64, 8))
    # split the step's host time: issuing point-to-point calls / blocked in their waits (under
    # gloo a wait blocks the HOST until the neighbour's data is there -- chain latency, which
    # under RCCL is a stream wait of a few microseconds) / everything else (Python, events)
    acc = {"issue": 0.0, "wait": 0.0, "n_issue": 0}
    def timed(fn, key):
        def w(*a, **k):
            t = time.perf_counter(); r = fn(*a, **k); acc[key] += time.perf_counter() - t
            if key == "issue": acc["n_issue"] += 1
            return r
        return w
    comm = eng.comm
    comm._isend, comm._irecv = timed(comm._isend, "issue"), timed(comm._irecv, "issue")
    comm.halo = timed(comm.halo, "wait")   # batched and waited for inside: counted as blocking
    slab._Xfer.wait = timed(slab._Xfer.wait, "wait")
    for _ in range(5): eng.run(img, mask, out)
    dist.barrier()
    acc.update(issue=0.0, wait=0.0, n_issue=0)
    t0 = time.perf_counter()
    for _ in range(steps): eng.run(img, mask, out)
    t = (time.perf_counter() - t0) / steps * 1e3
    eng.finish(); dist.barrier()
    calls = len(eng.items) * 2  # sweeps per step; each with one receive and one send where a neighbour exists
    if rank == world // 2:
        q.put((world, len(eng.items), len(eng.scale_groups), len(eng.groups), t, acc["issue"] / steps * 1e3,
               acc["wait"] / steps * 1e3, acc["n_issue"] / steps))
    dist.destroy_process_group()


if __name__ == "__main__":
    import socket
    import torch.multiprocessing as mp
    for W in [int(a) for a in sys.argv[1:]] or [2, 4, 8]:
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        q = mp.get_context("spawn").SimpleQueue()
        mp.spawn(worker, args=(W, port, 50, q), nprocs=W, join=True)
        w, items, sg, lg, t, ti, tw, ni = q.get()
        print("W=%d, interior rank, %d items per direction (%d scale groups x %d line groups): step %.3f ms of host "
              "time = %.3f issuing %d isend/irecv (%.1f us each) + %.3f blocked in waits and the halo exchange "
              "(gloo: the host waits for the neighbour) + %.3f engine (Python, no-op stages)"
              % (w, items, sg, lg, t, ti, ni, ti / max(ni, 1) * 1e3, tw, t - ti - tw))
