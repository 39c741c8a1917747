"""Steady-state step period of the slab engine's dependency structure (no GPU, no torch): a
list-scheduling simulation of W ranks with the kernel durations measured on one MI355X
(profiles/r03_rank_proxy_kernel_stats.csv) and an assumed link.

Per rank and step t: prepass, the lean and the fused sweep of every item in the engine's
schedule order, then X and the rest of the bulk work.  Every stream is in order; all kernels of
a rank share its GPU (one at a time, the chain streams first); a state record travels one link
per edge and direction (one at a time) and can only leave once the receiver has consumed the
record of the step before.  Buffer sets: prepass(t) after the fused sweeps of step t - depth,
the first fused sweep of step t after X(t - depth).

    python scripts/experiments/slab_chain_sim.py            # the table in DESIGN.md section 6
"""
import sys


def schedule(rank, world, n):
    ev = []
    for i in range(n):
        ev.append((rank + i, 0, i))
        ev.append((world - 1 - rank + i, 1, i))
    ev.sort()
    return [(d, i) for _, d, i in ev]


def simulate(W=8, steps=40, depth=3, split=True, G=4, wire=0.18, lat=0.03,
             t_prep=0.038, t_lean=0.027, t_fused=0.057, t_x=0.285, t_rest=0.75, concurrent=False, halo=0.1, post_split=True, skew=False):
    """concurrent=False: all kernels of a rank one at a time, not preempted (a sweep may wait
    behind a whole bulk kernel: pessimistic for the chain's latency); True: the chain kernels run
    beside the bulk kernels at no cost and the bulk kernels are stretched so that they alone fill
    the measured step of an overlapped rank (1.33 of 1.39 ms: optimistic for the latency)."""
    if concurrent:
        k = (t_prep + G * (t_lean + t_fused) + t_x + t_rest) * (1.33 / 1.39) / (t_x + t_rest)
        t_x, t_rest = t_x * k, t_rest * k
    chain_res = (lambda r: ("gpu-chain", r)) if concurrent else (lambda r: ("gpu", r))
    ops = {}       # name -> dict(dur, res, deps, stream, prio)
    order = []     # enqueue order (per stream in-order constraint is derived from it)
    last_in_stream = {}

    def add(name, dur, res, deps, stream, prio):
        deps = [d for d in deps if d is not None]   # filtered below, once every op exists
        if stream is not None:
            prev = last_in_stream.get(stream)
            if prev is not None:
                deps.append(prev)
            last_in_stream[stream] = name
        ops[name] = dict(dur=dur, res=res, deps=deps, prio=prio)
        order.append(name)

    def lean_of(r):
        return 0 if r <= W - 1 - r else 1

    for t in range(steps):
        for r in range(W):
            lean = lean_of(r)
            sL = ("L", r)
            sF = ("F", r) if split else sL
            # skew (one chain stream, software-pipelined by one step): the stream holds, per step,
            # the prepass and the lean sweeps of step t+1 interleaved with the fused sweeps of step t
            if skew and t == 0:
                add(("prep", r, 0), t_prep, chain_res(r), [], sL, 0)
            tp = t + 1 if skew else t
            if tp < steps:
                add(("prep", r, tp), t_prep, chain_res(r), [("sweep", r, tp - depth, 1 - lean, G - 1)], sL, 0)
            sched = schedule(r, W, G)
            if skew and t == 0:   # the very first lean sweeps
                sched = [(d, i, 0) for d, i in sched if d == lean] + [(d, i, (1 if d == lean else 0)) for d, i in sched]
            else:
                sched = [(d, i, (t + 1 if (skew and d == lean) else t)) for d, i in sched]
            for d, i, t_ in sched:
                if t_ >= steps:
                    continue
                _emit_sweep(add, ops, r, t_, d, i, lean, W, G, depth, sL, sF, chain_res, t_lean, t_fused, wire, lat, post_split)
            add(("x", r, t), t_x, ("gpu", r), [("sweep", r, t, 1 - lean, i) for i in range(G)], ("B", r), 1)
            _emit_rest(add, ops, r, t, W, t_rest, halo)
    for o in ops.values():   # dependencies on steps before the first one do not exist
        o["deps"] = [d for d in o["deps"] if d in ops]
    finish, res_free, pending = {}, {}, list(order)
    while pending:
        best = None
        for name in pending[:4000]:
            o = ops[name]
            if any(d not in finish for d in o["deps"]):
                continue
            start = max([finish[d] for d in o["deps"]] + [res_free.get(o["res"], 0.0)])
            key = (start, o["prio"])
            if best is None or key < best[0]:
                best = (key, name, start)
        _, name, start = best
        o = ops[name]
        res_free[o["res"]] = start + o["dur"]
        finish[name] = start + o["dur"] + o.get("lat", 0.0)
        pending.remove(name)
    done = [max(finish[("rest", r, t)] for r in range(W)) for t in range(steps)]
    k = steps // 2
    return (done[-1] - done[k]) / (steps - 1 - k)


def _emit_sweep(add, ops, r, t, d, i, lean, W, G, depth, sL, sF, chain_res, t_lean, t_fused, wire, lat, post_split):
    src = r - 1 if d == 0 else r + 1          # where the state comes from
    has_nb = 0 <= src < W
    if has_nb:
        # the receive is posted from an in-order stream of event waits: once the sweep of the step
        # before has consumed the buffer (post_split: one such stream per direction, else one for
        # both -- a lean receive then queues behind the wait for a fused sweep of the step before)
        add(("post", r, t, d, i), 0.0, ("none", r, t, d, i), [("sweep", r, t - 1, d, i)],
            ("P", r, d if post_split else 0), 0)
    deps = [("xfer", src, t, d, i)] if has_nb else []
    deps.append(("xfer", r, t - 1, d, i))     # the record of the step before has left the out buffer
    if d == lean:
        add(("sweep", r, t, d, i), t_lean, chain_res(r), deps, sL, 0)
    else:
        deps += [("sweep", r, t, lean, i), ("x", r, t - depth)]
        add(("sweep", r, t, d, i), t_fused, chain_res(r), deps, sF, 0)
    dst = r + 1 if d == 0 else r - 1
    if 0 <= dst < W:   # the record leaves once swept and once the receiver has posted its receive
        add(("xfer", r, t, d, i), wire, ("link", min(r, dst), d),
            [("sweep", r, t, d, i), ("post", dst, t, d, i)], ("N", r, d), 0)
        ops[("xfer", r, t, d, i)]["lat"] = lat


def _emit_rest(add, ops, r, t, W, t_rest, halo):
    # Y, then the stencil planes of both neighbours' Y output (halo: one message each way, its own
    # communicator), then the feature launches: four kernels of the bulk stream
    add(("rest0", r, t), t_rest / 4, ("gpu", r), [], ("B", r), 1)
    add(("halo", r, t), 0.0, ("nic", r), [("rest0", q, t) for q in (r - 1, r + 1) if 0 <= q < W], ("B", r), 1)
    ops[("halo", r, t)]["lat"] = halo
    for k in (1, 2):
        add(("rest%d" % k, r, t), t_rest / 4, ("gpu", r), [], ("B", r), 1)
    add(("rest", r, t), t_rest / 4, ("gpu", r), [], ("B", r), 1)


if __name__ == "__main__":
    print("ms per step in steady state: kernels of a rank one at a time / chain kernels beside the bulk ones")
    for W in ((8,) if "--w8" in sys.argv else (2, 4, 8)):
        # kernel times scale with the slab (64 planes at W = 8), and so does nothing else
        f = 8 / W
        kw = dict(t_prep=0.038 * f, t_lean=0.027 * f, t_fused=0.057 * f, t_x=0.285 * f, t_rest=0.75 * f)
        for wire in (0.18, 0.28):
            row = []
            for split, depth, ps in ((False, 2, False), (True, 4, False), (True, 2, True), (True, 3, True), (True, 4, True)):
                a = simulate(W=W, depth=depth, split=split, wire=wire, post_split=ps, **kw)
                b = simulate(W=W, depth=depth, split=split, wire=wire, concurrent=True, post_split=ps, **kw)
                row.append("%s, %s, depth %d: %.2f / %.2f" % ("two chain streams" if split else "one chain stream ",
                                                             "receives posted per direction" if ps else "one posting stream", depth, a, b))
            print("W=%d  item on the wire %.2f ms | " % (W, wire) + " | ".join(row))
    sys.stdout.flush()
