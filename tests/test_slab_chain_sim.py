"""The reasoning behind the slab engine's stream layout, kept executable: the dependency
structure of eight ranks, list-scheduled with the measured kernel times
(scripts/experiments/slab_chain_sim.py).  No test of this pool can show it on hardware -- RCCL
refuses two ranks on one GPU and gloo's waits block the host -- so the simulation is what pins
the three in-order traps found in round 3."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sim():
    spec = importlib.util.spec_from_file_location(
        "slab_chain_sim", os.path.join(ROOT, "scripts", "experiments", "slab_chain_sim.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_one_chain_stream_costs_a_chain_latency_per_step():
    sim = _sim()
    kw = dict(W=8, steps=16, wire=0.18)
    one = sim.simulate(depth=2, split=False, post_split=False, **kw)
    two = sim.simulate(depth=4, split=True, post_split=True, **kw)
    local = 0.038 + 4 * (0.027 + 0.057) + 0.285 + 0.75 + 0.1   # kernels of a step + the exposed stencil exchange
    assert one > 2.2, one                   # the end ranks hand the chains back and forth
    assert two < 1.1 * local, (two, local)  # the local work bounds the step again
    # software pipelining inside ONE stream is no substitute
    assert sim.simulate(depth=3, split=False, skew=True, **kw) > 2.0


def test_buffer_sets_and_posting_streams_matter_on_slower_links():
    sim = _sim()
    kw = dict(W=8, steps=16, wire=0.28, split=True)
    assert sim.simulate(depth=2, **kw) > 1.25 * sim.simulate(depth=4, **kw)
    assert sim.simulate(depth=4, post_split=False, **kw) > 1.15 * sim.simulate(depth=4, post_split=True, **kw)
