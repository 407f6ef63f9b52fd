"""GPU: the bench.py contract the driver depends on -- one JSON line on stdout with the metric,
the roofline and cpu_baseline objects -- on the small workload, at N=1 and as two bond-sharded
ranks sharing the test GPU (gloo-staged collectives)."""

import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_bench_single_gpu_contract():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "C2", "--steps", "4", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    o = _line(p.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in o, k
    assert o["metric"] == "tdvp_sweeps_per_sec" and o["unit"] == "sweeps/s" and o["n_gpus"] == 1 and o["steps"] == 4
    assert o["higher_is_better"] is True and o["vs_baseline"] is None and o["data"] == "synthetic" and o["dtype"] == "c128"
    assert "workload" in o["config"] and "model" not in o["config"]
    assert abs(o["value"] - 1e3 / o["ms_per_step"]) < 1e-6 * o["value"]
    r = o["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c = o["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "sweeps/s" and c["sample"]
    assert abs(o["config"]["norm_after"] - 1) < 1e-10


def test_bench_two_ranks_bond_sharded():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", MITDVP_DIST_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "C2", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline"]
    procs = [subprocess.Popen(cmd, env=dict(env, RANK=str(r), LOCAL_RANK="0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True, cwd=ROOT) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[1][-1500:] for o in outs)
    o = _line(outs[0][0])
    assert not [l for l in outs[1][0].splitlines() if l.startswith("{")]  # only rank 0 prints
    assert o["n_gpus"] == 2 and o["scaling"] == "strong" and o["config"]["collectives"] > 0
    assert "bond-sharded" in o["config"]["parallelism"] and abs(o["config"]["norm_after"] - 1) < 1e-10
