"""bench.py's output contract (one JSON line; the keys and meanings the driver reads), on a
small volume so that it takes seconds.  GPU only: the bench has no CPU path."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                       # exactly one line on stdout
    return json.loads(lines[0])


def test_single_gpu_line():
    d = run_bench("--gpus", "1", "--steps", "2", "--warmup", "1", "--size", "64", "96", "128",
                  "--cpu-sample", "48")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"].startswith("Mvoxels/sec Hessian+eig") and d["unit"] == "Mvoxels/s"
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - 64 * 96 * 128 * 3 / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 1e-3
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None and r["issue"] is None         # PMC figures exist only for the default workload
    assert r["measured_fill_GBs"] > 1000 and r["measured_copy_GBs"] > 1000   # the box's own stream rates
    assert r["kernels"]["features"]["alg_bytes_per_voxel"] == 37.0 and r["kernels"]["iir_y"]["alg_bytes_per_voxel"] == 6.0
    assert set(r["kernels"]) >= {"iir_z", "iir_x", "iir_y", "features", "prep"}
    assert r["kernels"]["features"]["launches_per_step"] == 3.0
    assert r["kernels"]["iir_z"]["launches_per_step"] == 1.0   # all (scale, field) jobs in one launch
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c


def test_slab_engine_line_with_one_rank():
    """The multi-GPU engine (one rank: no neighbour, so no transfer) prints the same contract,
    and its description comes from what the runner built."""
    d = run_bench("--gpus", "1", "--steps", "1", "--warmup", "1", "--size", "64", "64", "64",
                  "--force-slab", "--no-cpu-baseline", "--i16", "--spacing", "0.7", "0.7", "1.0")
    assert d["n_gpus"] == 1 and d["value"] > 0
    w = d["config"]["workload"]
    assert "1 Z-slabs" in w and "int16" in w and "spacing [0.7, 0.7, 1.0]" in w
    assert set(d["roofline"]["kernels"]) >= {"zslab_sweep", "zslab_combine", "iir_x", "iir_y", "features"}
    assert d["roofline"]["issue"] is None                      # PMC instruction counts exist only for the default workload
    assert "cpu_baseline" not in d


def test_per_rank_timing_proxy_is_marked_as_such():
    """--proxy-world: one GPU runs the local work of a rank with neighbours (no transfers);
    the line must say that it is no headline."""
    d = run_bench("--gpus", "1", "--steps", "1", "--warmup", "1", "--size", "64", "64", "64",
                  "--proxy-world", "4", "--proxy-rank", "1", "--no-cpu-baseline")
    assert d["headline"] is False and "rank 1 of 4" in d["config"]["proxy"]
    assert "4 Z-slabs" in d["config"]["workload"] and d["n_gpus"] == 1
    assert set(d["roofline"]["kernels"]) >= {"zslab_sweep", "zslab_combine", "iir_x", "iir_y", "features"}
