#!/usr/bin/env python3
"""bench.py -- throughput of the per-voxel Hessian feature path on MI355X.

Metric (BASELINE.json): "Mvoxels/sec Hessian+eig, 512^3 fp32 3-scale @1/2/4/8 GPU;
%HBM roofline".  One step = one pass of ImageToEmphysemaFeaturesFilter over the whole
512^3 float32 synthetic volume at sigma = 1, 2, 4 (BASELINE configs[2] at N=1,
configs[3] = the same volume cut into N Z-slabs at N>1), 8 output components per
voxel per scale, inputs resident in HBM when the timed region starts and outputs
left in HBM.

  value = nx*ny*nz * n_scales / t_step / 1e6          [Mvoxels/s, "voxel-scales"]

Launch: `python bench.py --gpus 1`; for N>1 either
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (the ranks read
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment) or plain
`python bench.py --gpus N`: with no WORLD_SIZE in the environment the process becomes a
launcher that never touches the GPU, starts the N ranks as child processes, relays rank 0's
line and exits with the worst exit code (launch_ranks).  Prints ONE JSON line.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ALG_BYTES_PER_VOXEL_SCALE = 37  # 4 (image) + 1 (mask) + 8*4 (out)

# Slab engine: a rank drives eleven streams (bulk, lean chain, fused chain, two for posting
# receives, and RCCL's own stream per edge and traffic class).  HIP multiplexes streams onto
# GPU_MAX_HW_QUEUES hardware queues (default 4), and two streams that share a queue are in order
# with each other: a receive that waits for its neighbour would hold back whatever shares its
# queue, and the lean chain could no longer run ahead of the fused one.  The engine's enqueue
# order keeps that free of deadlock (slab.py); a queue per stream keeps it free of false waits
# (scripts/experiments/stream_independence_probe.py: independent at 16, not at 4).  Read when
# the HIP runtime starts, so it is set before anything touches the GPU; the caller's own value
# wins.
if int(os.environ.get("WORLD_SIZE", "1")) > 1 or any(a in sys.argv for a in ("--force-slab", "--proxy-world")):
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PKG = "image-feature-extraction_amd"

# compulsory bytes per voxel of ONE field pass / feature launch of each kernel kind (DESIGN.md
# "Kernels"): a line-kernel launch covers several (scale, field) jobs.  Two entries depend on
# the options of the run (kernel_alg_bytes): with the quotient stored by the last axis pass
# (the default with a mask) that pass writes half a value per field and the feature kernel
# reads one field, 4 + 1 + 32 = 37 B; with two fields out of the last pass it reads both, 41 B.
KERNEL_ALG_BYTES = {"iir_z": 8.0, "iir_x": 8.0, "iir_y": 8.0, "features": 37.0, "prep": 13.0,
                    "zslab_sweep": 4.0, "zslab_combine": 8.0}


def kernel_alg_bytes(args):
    k = dict(KERNEL_ALG_BYTES)
    masked = args.mask != "none"
    fused = masked and not args.no_fused_divide and (args.iir_ckpt or 2) == 2
    if masked and fused:
        k["iir_y"] = 6.0       # 4 in, the quotient shared by the two fields out
    elif masked:
        k["features"] = 41.0   # numerator and denominator in
    if not masked:
        k["features"] = 36.0   # no mask byte
    return k


# Second roofline (DESIGN.md "Where the time goes"): on CDNA4 a vector instruction of a wave64
# occupies its SIMD's issue for 4 cycles, so a kernel cannot finish before
#   instructions x 4 / (256 CUs x 4 SIMDs x sustained clock).
# The instruction counts are SQ_INSTS_VALU per launch from the committed PMC passes of this
# workload (profiles/r03_traffic.json, written by scripts/make_traffic.py), read by kernel kind.
SIMDS, SUSTAINED_GHZ = 1024, 2.03  # GRBM_GUI_ACTIVE / 8 / duration under this load
PROFILE_JSON = "r03_traffic.json"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)   # 0.2 s of device time; the first steps after
    ap.add_argument("--warmup", type=int, default=5)   # start-up run 2 % slower (clocks settle)
    ap.add_argument("--size", type=int, nargs=3, default=[512, 512, 512], metavar=("NZ", "NY", "NX"))
    ap.add_argument("--sigmas", type=float, nargs="+", default=[1.0, 2.0, 4.0])
    ap.add_argument("--mask", choices=["ones", "ellipsoids", "none"], default="ones",
                    help="ones: explicit all-ones uint8 mask (every voxel pays the full path); "
                         "ellipsoids: ~20%% foreground like a lung mask")
    ap.add_argument("--layout", choices=["interleaved", "planar"], default="interleaved")
    ap.add_argument("--spacing", type=float, nargs=3, default=[1.0, 1.0, 1.0], metavar=("SX", "SY", "SZ"))
    ap.add_argument("--i16", action="store_true", help="int16 CT-like input (BASELINE configs[4])")
    ap.add_argument("--trig", type=int, default=2,
                    help="IFE_OPT_TRIG_MODE: 2 (library default) float acos/cos inside the 1e-5 "
                         "bar; 0 double evaluation, bit-faithful to the oracle")
    ap.add_argument("--iir-block", type=int, default=None)
    ap.add_argument("--iir-ckpt", type=int, default=None)
    ap.add_argument("--iir-fma", action="store_true",
                    help="the RELAXED line, never the headline: fused multiply-adds in the line "
                         "recurrences (IFE_OPT_IIR_FMA, not the reference's arithmetic).  The run then "
                         "also compares the whole output with the oracle at full size and prints the "
                         "MAXIMUM eigenvalue error (config.iir_fma = 1, `relaxed` in the JSON line)")
    ap.add_argument("--relaxed-bar", type=float, default=None,
                    help="--iir-fma: fail if the maximum eigenvalue error / |lambda1| exceeds this")
    ap.add_argument("--zchunk", type=int, default=None)
    ap.add_argument("--const-lines", type=int, default=0, choices=[0, 1],
                    help="IFE_OPT_CONST_LINES.  The headline keeps it OFF: with the all-ones mask of "
                         "this workload every denominator line is the constant 1 and would be copied "
                         "instead of filtered; off, every voxel pays the full path.  (The library "
                         "default is on; the time with it on is reported beside the headline.)")
    ap.add_argument("--no-shortcut-leg", action="store_true",
                    help="skip the four extra steps that time the constant-line shortcut beside the "
                         "headline (the profiling scripts use it: every profiled launch is a headline launch)")
    ap.add_argument("--no-fused-divide", action="store_true",
                    help="IFE_OPT_FUSED_DIVIDE=0: two fields out of the last axis pass (A/B)")
    ap.add_argument("--feat-ring", type=int, default=1, choices=[0, 1],
                    help="IFE_OPT_FEAT_RING (A/B): 1 planes by LDS-DMA into a ring (default), 0 register-staged")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stream-probe", action="store_true",
                    help="skip the 4-GiB fill / copy that measures the box's streaming rates")
    ap.add_argument("--proxy-world", type=int, default=None,
                    help="timing proxy (implies --force-slab, never a headline): this one GPU runs the "
                         "LOCAL work of rank --proxy-rank of this many Z-slabs of --size, transfers left "
                         "out (slab.NullComm); `value` is then the whole volume over this rank's step")
    ap.add_argument("--proxy-rank", type=int, default=0)
    ap.add_argument("--slab-depth", type=int, default=None,
                    help="slab engine: sets of per-step buffers = steps the boundary chain may run ahead (default 4)")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="diagnostic: leave the per-kernel hipEvents out of the timed region (no kernel "
                         "table, no roofline.achieved from device time)")
    ap.add_argument("--force-slab", action="store_true",
                    help="run the Z-slab engine (RCCL exchanges) even with one rank")
    ap.add_argument("--cpu-sample", type=int, default=512, help="edge of the CPU baseline cube")
    ap.add_argument("--scales-per-item", type=int, default=None,
                    help="N>1: scales whose boundary sweeps share a launch and a message (slab.py)")
    ap.add_argument("--line-groups", type=int, default=None,
                    help="N>1: items per scale on the boundary-state chains (slab.py)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="launcher rehearsal without a GPU: the ranks meet over gloo, agree on a "
                         "sum and rank 0 prints a stub line (tests/test_bench_launch.py)")
    ap.add_argument("--launch-timeout", type=float, default=900.0,
                    help="self-launch: seconds after which the launcher ends its ranks")
    a = ap.parse_args()
    if a.proxy_world:
        if a.gpus != 1 or not 0 <= a.proxy_rank < a.proxy_world:
            ap.error("--proxy-world needs --gpus 1 and 0 <= --proxy-rank < --proxy-world")
        a.force_slab = True
    return a


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks ourselves.

    This process makes no GPU call (it never imports torch): a parent that had initialised
    the GPU could not hand it to its children.  Each rank is a fresh interpreter running this
    file with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, exactly what
    torch.distributed.run would give it.  Rank 0's stdout is the job's stdout (one JSON
    line); every rank's stderr passes through.  A rank that fails takes the others down (they
    would otherwise sit in a collective until the process-group timeout), and the launcher's
    exit code is the worst of the ranks'."""
    import socket
    import subprocess
    n = args.gpus
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in env:
        with socket.socket() as sk:  # a free port of this host
            sk.bind(("127.0.0.1", 0))
            env["MASTER_PORT"] = str(sk.getsockname()[1])
    env["WORLD_SIZE"] = env["LOCAL_WORLD_SIZE"] = str(n)
    env.setdefault("GPU_MAX_HW_QUEUES", "16")  # see the top of this file
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                      stdin=subprocess.DEVNULL, text=True))
    deadline = time.time() + args.launch_timeout
    codes = [None] * n
    failed = False
    while any(c is None for c in codes):
        for r, pr in enumerate(procs):
            if codes[r] is None:
                codes[r] = pr.poll()
                if codes[r] not in (None, 0):
                    failed = True
        if failed or time.time() > deadline:
            for r, pr in enumerate(procs):  # exactly the processes started here, by pid
                if codes[r] is None:
                    pr.terminate()
            t_end = time.time() + 10
            for r, pr in enumerate(procs):
                if codes[r] is None:
                    try:
                        codes[r] = pr.wait(timeout=max(0.1, t_end - time.time()))
                    except subprocess.TimeoutExpired:
                        pr.kill()
                        codes[r] = pr.wait()
                    if not failed:
                        codes[r] = codes[r] or 124  # ended by the launcher's timeout
            break
        time.sleep(0.05)
    out = procs[0].stdout.read() if procs[0].stdout else ""
    for ln in out.splitlines():  # the contract is ONE line: library banners go to stderr
        (sys.stdout if ln.lstrip().startswith("{") else sys.stderr).write(ln + "\n")
    sys.stdout.flush()
    worst = max((abs(c) for c in codes if c), default=0)
    if worst:
        sys.stderr.write("bench.py launcher: rank exit codes %s\n" % codes)
    return worst if worst < 256 else 1


def dry_launch_rank(args):
    """Launcher rehearsal (no GPU): rendezvous over gloo, one all-reduce, one line."""
    import datetime
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"dry_launch": True, "n_gpus": world, "rank_sum": t.item(),
                          "local_ranks": "0..%d" % (world - 1)}), flush=True)
    dist.destroy_process_group()


def cpu_baseline(synth, seed, sigmas, edge):
    """The oracle ("port") timed on this host's cores on a bounded sample: an edge^3
    corner of the same synthetic volume, same sigmas.  Reported, never the target."""
    from oracle import pyoracle
    pyoracle.build()
    threads = min(os.cpu_count() or 1, 16)
    pyoracle.set_threads(threads)
    shape = (edge, edge, edge)
    img = synth.volume_f32(shape, seed)
    mask = np.ones(shape, np.uint8)
    pyoracle.emphysema_features(img[:32, :32, :32].copy(), mask[:32, :32, :32].copy(), 1.0)  # warm
    t0 = time.perf_counter()
    for s in sigmas:
        pyoracle.emphysema_features(img, mask, float(s))
    dt = time.perf_counter() - t0
    # one-thread figure on a 128^3 corner (SURVEY.md 8d asks for both)
    pyoracle.set_threads(1)
    e1 = min(edge, 128)
    img1, mask1 = img[:e1, :e1, :e1].copy(), mask[:e1, :e1, :e1].copy()
    t1 = time.perf_counter()
    for s in sigmas:
        pyoracle.emphysema_features(img1, mask1, float(s))
    dt1 = time.perf_counter() - t1
    pyoracle.set_threads(threads)
    return {"value": round(edge ** 3 * len(sigmas) / dt / 1e6, 3), "unit": "Mvoxels/s",
            "cores": threads, "kind": "port",
            "one_thread_value": round(e1 ** 3 * len(sigmas) / dt1 / 1e6, 3),
            "sample": "%d^3 corner of the same synthetic volume, sigmas %s, all-ones mask, "
                      "%.1f s of CPU work" % (edge, list(sigmas), dt)}


def relaxed_error(runner, synth, seed, sigmas, args):
    """What the fused recurrences cost in accuracy, over EVERY voxel of the bench volume at every
    scale, against the oracle (exact arithmetic, double trigonometry): the maximum, not a
    quantile.  The smoothed value may differ by float ulps; the second differences amplify
    that, so the eigenvalue error relative to |lambda1| is what is printed (and asserted
    against --relaxed-bar)."""
    from oracle import parity, pyoracle
    pyoracle.build()
    pyoracle.set_threads(min(os.cpu_count() or 1, 16))
    shape = runner.shape
    img = synth.volume_i16(shape, seed) if runner.i16 else synth.volume_f32(shape, seed)
    mask = (np.minimum(synth.mask_ellipsoids(shape), 1).astype(np.uint8) if args.mask == "ellipsoids"
            else np.ones(shape, np.uint8))
    res = {"iir_fma": 1, "compared": "every voxel of the volume, all scales, against the oracle",
           "scales": []}
    worst = 0.0
    for s, sigma in enumerate(sigmas):
        got = runner.d_out[s].cpu().numpy()
        ref = pyoracle.emphysema_features(img.astype(np.float32), mask, float(sigma), tuple(args.spacing))
        p = parity.eig_parity(got, ref, tie_tol=2e-5)
        ds = np.abs(got[..., 0].astype(np.float64) - ref[..., 0])
        scale = np.maximum(np.abs(ref[..., 0]).astype(np.float64), 1e-30)
        lam = np.maximum(np.abs(ref[..., 2]).astype(np.float64), 1e-30)
        ee = np.abs(np.sort(got[..., 2:5].astype(np.float64), -1) - np.sort(ref[..., 2:5].astype(np.float64), -1)).max(-1) / lam
        res["scales"].append({"sigma": sigma, "max_eig_err_rel_lambda1": p["max_err"],
                              "voxels_beyond_1e-5": int((ee > 1e-5).sum()), "voxels": int(ee.size),
                              "smoothed_value_differs_at": int((got[..., 0] != ref[..., 0]).sum()),
                              "max_smoothed_value_rel_err": float((ds / scale).max()),
                              "order_diff": p["order_diff"]})
        worst = max(worst, p["max_err"])
        del got, ref, ee, ds
    res["max_eig_err_rel_lambda1"] = worst
    res["bar"] = args.relaxed_bar
    if args.relaxed_bar is not None and not worst <= args.relaxed_bar:
        raise SystemExit("relaxed line: maximum eigenvalue error %.3g |lambda1| exceeds the bar %.3g"
                         % (worst, args.relaxed_bar))
    return res


def measured_stream_gbs(torch, ctx, dev, nbytes=4 << 30, reps=5):
    """The box's own streaming rates with this library's access shape (16 B per lane,
    non-temporal, grid-stride; ife_measure_stream): a 4-GiB fill (bytes written per second) and
    a 4-GiB copy (bytes read + written per second).  Printed beside the 8 TB/s peak as what a
    pure stream achieves here; far larger than the 256 MB Infinity Cache."""
    a = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    b = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    fill = ctx.measure_stream(0, a.data_ptr(), None, nbytes, reps)
    copy = ctx.measure_stream(1, b.data_ptr(), a.data_ptr(), nbytes, reps)
    del a, b
    return fill, copy


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))  # before anything touches the GPU
    if args.dry_launch:
        return dry_launch_rank(args)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (unset WORLD_SIZE to let bench.py start its own ranks)"
                         % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_slab
    if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
        os.environ["NCCL_DEBUG"] = ""  # the RCCL version banner goes to stdout; keep it to the JSON line
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # a transfer that has not completed after three minutes is a deadlock, not a slow link:
        # the watchdog then ends the job instead of leaving kernels spinning on the GPUs
        import datetime
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev,
                                timeout=datetime.timedelta(seconds=180))

    pkg = importlib.import_module(PKG)
    synth = importlib.import_module(PKG + ".synthetic")
    nz, ny, nx = args.size
    sigmas = list(args.sigmas)
    seed = synth.SEED_CONFIG[3]
    layout = pkg.INTERLEAVED if args.layout == "interleaved" else pkg.PLANAR

    if use_dist:
        slab = importlib.import_module(PKG + ".slab")
        runner = slab.SlabRunner(pkg, synth, (nz, ny, nx), sigmas, seed, args.mask, layout,
                                 rank, world, dev, args)
    else:
        runner = SingleGpuRunner(pkg, synth, (nz, ny, nx), sigmas, seed, args.mask, layout, dev,
                                 args)

    def barrier():
        if use_dist:
            dist.barrier()

    for _ in range(args.warmup):
        runner.step()
    # Per-kernel hipEvents (two records per launch).  One GPU: inside the timed region -- eight
    # launches of 0.3-2 ms per step, the records cost nothing measurable (8.86 vs 8.88 ms).  Slab
    # engine: ~26 launches of 20-60 us per step on two streams, where the records cost 2.7 % of
    # the step (scripts/experiments/r3_events_cost.sh: 1.367 -> 1.330 ms); there the K timed
    # steps run without them and the kernel table comes from extra steps behind the timed region.
    events_in_timed = not use_dist and not args.no_kernel_events
    for c in runner.contexts():
        c.set_option(pkg.OPT_PROFILE, 1 if events_in_timed else 0)
        c.reset_kernel_times()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        runner.step()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    ksteps = args.steps  # steps the kernel table below is summed over
    if use_dist and not args.no_kernel_events:
        ksteps = min(args.steps, 5)
        for c in runner.contexts():
            c.set_option(pkg.OPT_PROFILE, 1)
            c.reset_kernel_times()
        for _ in range(ksteps):
            runner.step()
        torch.cuda.synchronize()
        barrier()
    ktimes = {}
    for c in runner.contexts():
        for name, (n, ms) in c.kernel_times().items():
            n0, ms0 = ktimes.get(name, (0, 0.0))
            ktimes[name] = (n0 + n, ms0 + ms)
        c.set_option(pkg.OPT_PROFILE, 0)
    # beside the headline: the same step with the library's default constant-line shortcut
    shortcut_ms = None
    if not use_dist and not args.const_lines and not args.no_shortcut_leg:
        runner.ctx.set_option(pkg.OPT_CONST_LINES, 1)
        runner.step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            runner.step()
        torch.cuda.synchronize()
        shortcut_ms = (time.perf_counter() - t1) / 3 * 1e3
        runner.ctx.set_option(pkg.OPT_CONST_LINES, 0)
    relaxed = None
    if args.iir_fma and not use_dist:
        relaxed = relaxed_error(runner, synth, seed, sigmas, args)
    fill_gbs = copy_gbs = None
    if rank == 0 and not use_dist and not args.no_stream_probe:
        runner.release_outputs()  # the output volumes go back to the allocator before the probe's 8 GiB
        fill_gbs, copy_gbs = measured_stream_gbs(torch, runner.ctx, dev)

    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    nvox = nz * ny * nx
    share = args.proxy_world or world  # slabs the volume is cut into (this rank's work: nvox / share)
    t_step = dt / args.steps
    value = nvox * len(sigmas) / t_step / 1e6

    # ---- roofline: whole hot path, per step, from HIP events around every kernel ----
    # HBM bytes and vector instructions per launch come from the committed PMC passes of the
    # DEFAULT workload (PMC counters cannot be collected from inside this process)
    default_cfg = (not use_dist and [nz, ny, nx] == [512, 512, 512] and sigmas == [1.0, 2.0, 4.0]
                   and args.mask == "ones" and args.layout == "interleaved" and args.trig == 2
                   and not args.i16 and list(args.spacing) == [1.0, 1.0, 1.0] and not args.iir_fma
                   and not args.iir_block and not args.no_fused_divide and args.feat_ring == 1)
    tpath = os.path.join(ROOT, "profiles", PROFILE_JSON)
    prof = json.load(open(tpath)) if default_cfg and os.path.exists(tpath) else None
    prof_kinds = (prof or {}).get("kinds", {})
    alg = kernel_alg_bytes(args)
    kern = {}
    dev_ms_step = 0.0
    issue_floor_step = 0.0
    nfields = 1 if args.mask == "none" else 2
    for name, (n, ms) in ktimes.items():
        per_step_ms = ms / ksteps
        dev_ms_step += per_step_ms
        e = {"launches_per_step": n / ksteps, "avg_ms": round(ms / n, 4),
             "ms_per_step": round(per_step_ms, 4)}
        if name in alg:
            # units of work per step: field passes for the line kernels, scales for the rest
            units = {"prep": 1, "features": len(sigmas)}.get(name, len(sigmas) * nfields)
            gbs = alg[name] * units * (nvox / share) / (per_step_ms * 1e-3) / 1e9
            e["alg_bytes_per_voxel"] = alg[name]
            e["units_per_step"] = units
            e["achieved_GBs"] = round(gbs, 1)
            e["frac"] = round(gbs / HBM_PEAK_GBS, 4)
        if name in prof_kinds and prof_kinds[name].get("valu_insts"):
            floor = prof_kinds[name]["valu_insts"] * 4.0 / (SIMDS * SUSTAINED_GHZ * 1e9) * 1e3
            e["valu_insts_per_launch"] = prof_kinds[name]["valu_insts"]
            e["issue_floor_ms"] = round(floor, 4)  # per launch, like avg_ms
            issue_floor_step += floor * n / ksteps
        kern[name] = e
    alg_bytes_step_rank = ALG_BYTES_PER_VOXEL_SCALE * (nvox / share) * len(sigmas)
    # one GPU: the kernels of a step run back to back, their hipEvent durations add up to the
    # device time.  Slab engine: two streams per rank and successive steps overlap, so the sum
    # exceeds the step; the step's wall time is the denominator there
    scope_ms = t_step * 1e3 if use_dist else dev_ms_step
    achieved = alg_bytes_step_rank / (scope_ms * 1e-3) / 1e9 if scope_ms > 0 else 0.0
    dominant = max(kern, key=lambda k: kern[k]["ms_per_step"]) if kern else None
    traffic = prof.get("traffic_bytes_per_step") if prof else None
    issue = None
    if prof and issue_floor_step > 0:
        issue = {"floor_ms_per_step": round(issue_floor_step, 3),
                 "frac_of_issue_roofline": round(issue_floor_step / dev_ms_step, 4),
                 "rule": "SQ_INSTS_VALU per launch x 4 cycles / (%d SIMDs x %.2f GHz), summed over the "
                         "launches of a step; measured: the sum of hipEvent durations" % (SIMDS, SUSTAINED_GHZ),
                 "source": "profiles/%s (rocprofv3 PMC, SQ_INSTS_VALU per launch and kernel)" % PROFILE_JSON}
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "algorithmic_bytes_per_step": int(alg_bytes_step_rank),
                "measured_fill_GBs": round(fill_gbs, 1) if fill_gbs else None,
                "measured_copy_GBs": round(copy_gbs, 1) if copy_gbs else None,
                "measured_stream_note": "4-GiB float4 fill (bytes written / s) and copy (bytes read + written / s) "
                                        "with this library's access shape, on this box, after the timed region"
                if fill_gbs else None,
                "traffic_source": "profiles/%s (rocprofv3 PMC, bytes per step)" % PROFILE_JSON
                if traffic else None,
                "issue": issue,
                "scope": ("this rank's share of one step over the step's wall time (%.3f ms; the kernels "
                          "of its two streams overlap: their hipEvent durations, taken over %d extra steps "
                          "behind the timed region, sum to %.3f ms); "
                          % (scope_ms, ksteps, dev_ms_step) if use_dist else
                          "all kernels of one step (sum of hipEvent durations %.3f ms); " % dev_ms_step)
                         + "algorithmic bytes = 37 B x voxels x scales",
                "dominant_kernel": dominant, "kernels": kern}

    out = {
        "metric": "Mvoxels/sec Hessian+eig, 512^3 fp32 3-scale",
        "value": round(value, 1), "unit": "Mvoxels/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(t_step * 1e3, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32 storage / f64 line recurrences", "data": "synthetic",
        # described from what the runner actually built
        "config": {"workload": "%dx%dx%d %s volume, sigma=%s, 8 features/voxel/scale "
                               "(ImageToEmphysemaFeaturesFilter), uint8 mask=%s, %s output, "
                               "%s%s" % (nx, ny, nz, runner.config["input"], sigmas, args.mask,
                                         args.layout,
                                         "1 GPU" if not use_dist else
                                         "%d Z-slabs, boundary-state hand-off, %d line groups per scale"
                                         % (share, runner.config.get("line_groups", 1)),
                                         "" if runner.config["spacing"] == [1.0, 1.0, 1.0]
                                         else ", spacing %s" % runner.config["spacing"]),
                   "slab_engine": ({k: runner.config[k] for k in ("slab_planes", "line_groups", "scales_per_item", "depth")}
                                   if use_dist else None),
                   "iir_fma": 1 if args.iir_fma else 0,
                   "const_lines": args.const_lines,
                   "const_lines_meaning": "0: every line is filtered (every voxel pays the full path); "
                                          "1 (library default): lines that are all 0 or all 1 are copied",
                   "trig_mode": args.trig,
                   "trig_mode_meaning": {0: "double acos/cos (bit-faithful to the oracle)",
                                         1: "float overloads, correctly rounded",
                                         2: "float polynomials, max error 4.5e-7 |lambda1| over all of 512^3 x 3 scales "
                                            "(bar: 1e-5)"}.get(args.trig)},
        "volume_level_Mvoxels_per_s": round(nvox / t_step / 1e6, 1),
        "ms_per_step_with_constant_line_shortcut": round(shortcut_ms, 3) if shortcut_ms else None,
        "roofline": roofline,
    }
    if relaxed is not None:
        out["relaxed"] = relaxed
        out["headline"] = False
    if args.proxy_world:
        out["headline"] = False
        out["config"]["proxy"] = ("timing proxy: the local work of rank %d of %d Z-slabs on one GPU, transfers "
                                  "left out, state buffers zero -- `value` = the whole volume over this rank's "
                                  "step, an upper bound on what %d such GPUs could reach"
                                  % (args.proxy_rank, args.proxy_world, args.proxy_world))
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(synth, seed, sigmas, args.cpu_sample)
        print(json.dumps(out), flush=True)
    if use_dist:
        runner.finish()
        dist.barrier()
        dist.destroy_process_group()


class SingleGpuRunner:
    def __init__(self, pkg, synth, shape, sigmas, seed, mask_kind, layout, dev, args):
        import torch
        self.pkg, self.shape, self.sigmas, self.layout = pkg, shape, sigmas, layout
        self.i16, self.spacing = args.i16, tuple(args.spacing)
        nz, ny, nx = shape
        img = synth.volume_i16(shape, seed) if args.i16 else synth.volume_f32(shape, seed)
        if mask_kind == "ellipsoids":
            mask = np.minimum(synth.mask_ellipsoids(shape), 1).astype(np.uint8)
        else:
            mask = np.ones(shape, np.uint8)
        self.d_img = torch.from_numpy(img).to(dev)
        self.d_mask = torch.from_numpy(mask).to(dev)
        self.mask_ptr = None if mask_kind == "none" else self.d_mask.data_ptr()
        self.d_out = torch.empty((len(sigmas), nz, ny, nx, 8), dtype=torch.float32, device=dev)
        del img, mask
        self.ctx = pkg.Context(dev.index or 0)
        self.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        self.ctx.set_option(pkg.OPT_TRIG_MODE, args.trig)
        if args.iir_block:
            self.ctx.set_option(pkg.OPT_IIR_BLOCK, args.iir_block)
        if args.iir_ckpt:
            self.ctx.set_option(pkg.OPT_IIR_CKPT, args.iir_ckpt)
        if args.iir_fma:
            self.ctx.set_option(pkg.OPT_IIR_FMA, 1)
        if args.zchunk:
            self.ctx.set_option(pkg.OPT_ZCHUNK, args.zchunk)
        if args.no_fused_divide:
            self.ctx.set_option(pkg.OPT_FUSED_DIVIDE, 0)
        self.ctx.set_option(pkg.OPT_CONST_LINES, args.const_lines)
        self.ctx.set_option(pkg.OPT_FEAT_RING, args.feat_ring)
        self.ctx.reserve(shape)
        self.config = {"input": "int16" if args.i16 else "float32", "spacing": list(self.spacing)}

    def contexts(self):
        return [self.ctx]

    def release_outputs(self):
        self.d_out = None

    def step(self):
        self.ctx.emphysema_features_device(
            self.d_img.data_ptr(), self.pkg.I16 if self.i16 else self.pkg.F32, self.mask_ptr,
            self.pkg.U8, self.shape, self.spacing, self.sigmas, self.d_out.data_ptr(), self.layout)


if __name__ == "__main__":
    main()
