"""GPU: the bench.py contract the driver depends on -- one JSON line on stdout with the metric,
the roofline and cpu_baseline objects.  Covered: the small workload at N=1; the driver's EXACT
command line (default workload C4, --steps 20 --warmup 5) under a small wall budget; the plain
`python bench.py --gpus 2` form, which starts its own ranks (two bond-sharded ranks sharing the
test GPU, gloo-staged collectives); and the same two ranks started by a launcher."""

import json
import os
import socket
import subprocess
import sys
import time

import pytest

from helpers.ranks import run_ranks

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline")


class _Done:
    def __init__(self, returncode, stdout, stderr):
        self.returncode, self.stdout, self.stderr = returncode, stdout, stderr


def _run_bench(cmd, timeout, env=None):
    """subprocess.run(capture_output=True) with the child's progress notes (stderr) ALSO appended to
    gpurun_out/bench_contract_progress.txt as they come: a run of several minutes under captured output looks hung to a
    watchdog that only sees stdout, stderr and that directory."""
    import threading

    prog_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(prog_dir, exist_ok=True)
    err = []
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, env=env)

    def pump():
        with open(os.path.join(prog_dir, "bench_contract_progress.txt"), "a") as f:
            for line in p.stderr:
                err.append(line)
                f.write(line)
                f.flush()

    th = threading.Thread(target=pump, daemon=True)
    th.start()
    killer = threading.Timer(timeout, p.kill)  # stdout is one line: read to the end of the run, bounded by the timer
    killer.start()
    try:
        out = p.stdout.read()
        p.wait()
    finally:
        killer.cancel()
    th.join(10)
    return _Done(p.returncode, out, "".join(err))


def _line(out):
    # stdout carries the JSON line and NOTHING else (library banners -- gloo, RCCL -- are sent to stderr)
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out
    return json.loads(lines[0])


def _check_common(o):
    for k in KEYS:
        assert k in o, k
    assert o["metric"] == "tdvp_sweeps_per_sec" and o["unit"] == "sweeps/s"
    assert o["higher_is_better"] is True and o["vs_baseline"] is None and o["data"] == "synthetic" and o["dtype"] == "c128"
    assert "workload" in o["config"] and "model" not in o["config"]
    units = o["n_gpus"] if o["scaling"] == "weak" else 1  # replicas: every rank runs its own sweeps
    assert abs(o["value"] - units * 1e3 / o["ms_per_step"]) < 1e-6 * o["value"]
    r = o["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert 0 < r["frac"] <= 1.0, r  # a utilisation: executed work over the peak, never above 1
    assert 1 <= o["steps"] <= o["steps_requested"] and 1 <= o["warmup"]


def _check_secondary(rec, name):
    """one short leg of another BASELINE config under the `secondary` key: value, ms_per_step, roofline, cpu_baseline"""
    assert "error" not in rec, rec
    assert rec["workload"].startswith(name) and rec["unit"] == "sweeps/s" and rec["steps"] >= 2
    assert abs(rec["value"] - 1e3 / rec["ms_per_step"]) < 1e-6 * rec["value"]
    r = rec["roofline"]
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] <= 1.0
    assert "traffic_source" in r
    c = rec["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["unit"] == "sweeps/s" and c["sample"] and "cpu_model" in c
    assert c["pinned"] is True and c["socket"] is not None and c["cores"] >= 1  # one socket, its physical cores
    cfg = rec["config"]
    if cfg["integrator"] == "lanczos":  # unitary: the norm stays 1 and the energy where it was
        assert abs(cfg["norm_after"] - 1) < 1e-10
        assert abs(cfg["energy_after"] - cfg["energy_before"]) < 1e-7 * max(1.0, abs(cfg["energy_before"]))
    b = rec["breakdown_ms"]
    assert b["wall"] > 0 and b["launches"] > 0


def test_bench_single_gpu_contract():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "C2", "--steps", "4", "--warmup", "1",
                        "--secondary", "C3", "--secondary-seconds", "40"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    o = _line(p.stdout)
    _check_common(o)
    assert o["n_gpus"] == 1 and o["steps"] == 4 and o["warmup"] == 1
    c = o["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "sweeps/s" and c["sample"]
    assert "cpu_model" in c and c["pinned"] is True and c["affinity_cpus"] >= c["cores"] >= 1
    assert abs(o["config"]["norm_after"] - 1) < 1e-10
    # the C2 traffic figure comes from a pass of the one-launch kernel, the C3 one from a pass of the form that ran
    assert o["roofline"]["traffic_source"].endswith("c2_traffic.json")
    assert o["secondary"]["C3"]["roofline"]["apply_form"] in o["secondary"]["C3"]["roofline"]["traffic_source"]
    assert o["config"]["secondary_summary"]["C3"]["value"] == pytest.approx(o["secondary"]["C3"]["value"], rel=1e-4)
    # energy conservation of the run itself, on one GPU as well
    assert abs(o["config"]["energy_after"] - o["config"]["energy_before"]) < 1e-7 * max(1.0, abs(o["config"]["energy_before"]))
    assert list(o["secondary"]) == ["C3"]
    _check_secondary(o["secondary"]["C3"], "C3")


def test_bench_driver_command_line_fits_its_wall_budget():
    """`python3 bench.py --gpus 1 --steps 20 --warmup 5` is what the driver runs (600 s limit).  On
    the default workload (C4: about a minute per sweep) 25 sweeps do not fit: the run must cut
    them to the wall budget, say so, and still print the full line.  The budget is set low here
    (one warm-up + one timed sweep + the CPU sample) to keep the test short."""
    t0 = time.time()
    p = _run_bench([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5"],
                   timeout=660, env=dict(os.environ, MITDVP_BENCH_BUDGET="150"))
    wall = time.time() - t0
    assert p.returncode == 0, p.stderr[-2000:]
    o = _line(p.stdout)
    _check_common(o)
    assert o["config"]["workload"].startswith("C4") and o["config"]["D"] == 1024 and o["config"]["L"] == 64
    assert o["steps_requested"] == 20 and o["warmup_requested"] == 5 and o["steps"] < 20
    assert o["roofline"]["bound"] == "mfma" and o["roofline"]["unit"] == "TFLOP/s"
    assert o["roofline"]["algorithmic_tflops"] >= o["roofline"]["achieved"]
    assert o["cpu_baseline"]["value"] > 0
    assert abs(o["config"]["norm_after"] - 1) < 1e-10
    assert abs(o["config"]["energy_after"] - o["config"]["energy_before"]) < 1e-7 * max(1.0, abs(o["config"]["energy_before"]))
    assert o["roofline"]["traffic_source"] and o["roofline"]["traffic_source"].startswith("profiles/")
    # the dominant kernel by itself: the two large launches of the apply, executed flop over their own HIP-event time
    st = o["roofline"]["stage_executed_tflops"]
    assert len(st) == 3 and 0 < st[0] <= 78.6 and 0 < st[2] <= 78.6
    assert abs(o["roofline"]["dominant_kernel_frac"] - max(st[0], st[2]) / 78.6) < 1e-12
    assert o["roofline"]["dominant_kernel_frac"] >= o["roofline"]["frac"] - 1e-12  # the whole apply cannot beat its best stage
    # the other BASELINE configs ride on the same line, driver-observed
    assert sorted(o["secondary"]) == ["C2", "C3", "C5"]
    for w in ("C2", "C3", "C5"):
        _check_secondary(o["secondary"][w], w)
        assert o["config"]["secondary_summary"][w]["value"] == pytest.approx(o["secondary"][w]["value"], rel=1e-4)
        src = o["secondary"][w]["roofline"]["traffic_source"]
        assert src is None or w.lower() in src.lower() or "D%d_" % o["secondary"][w]["config"]["D"] in src, src
    # the driver keeps the parsed `config` and a tail of stdout: the summary rides in `config`, the line stays short
    assert len(json.dumps(o["secondary"])) < 8192 and len(p.stdout) < 14000, (len(json.dumps(o["secondary"])), len(p.stdout))
    assert wall < 520, wall  # budget 150 s + at most one sweep + the CPU sample + the secondary legs (100 s)


def test_bench_plain_multi_gpu_command_launches_its_own_ranks():
    """`python bench.py --gpus 2`: starts its own ranks; the default partitioning is by site ranges."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "C2", "--steps", "2", "--warmup", "2"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    o = _line(p.stdout)
    _check_common(o)
    assert o["n_gpus"] == 2 and o["scaling"] == "strong", p.stderr[-3000:]
    assert "site ranges" in o["config"]["parallelism"] and o["config"]["halo_messages"] > 0
    assert abs(o["config"]["norm_after"] - 1) < 1e-3  # the site-sharded scheme keeps the norm to O(dt^2)


def test_bench_plain_multi_gpu_command_bond_sharded():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "C2", "--steps", "2", "--warmup", "1",
           "--parallel", "tp"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    o = _line(p.stdout)
    _check_common(o)
    assert o["n_gpus"] == 2 and o["scaling"] == "strong" and o["config"]["collectives"] > 0, p.stderr[-3000:]
    assert "bond-sharded" in o["config"]["parallelism"] and abs(o["config"]["norm_after"] - 1) < 1e-10


def test_bench_two_ranks_bond_sharded():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", MITDVP_DIST_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "C2", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--parallel", "tp"]
    rcs, outs = run_ranks([cmd] * 2, [dict(env, RANK=str(r), LOCAL_RANK="0") for r in range(2)], timeout=600, cwd=ROOT,
                          split_stderr=True)
    assert rcs == [0, 0], "\n".join(o[1][-1500:] for o in outs)
    o = _line(outs[0][0])
    assert not [l for l in outs[1][0].splitlines() if l.startswith("{")]  # only rank 0 prints
    assert o["n_gpus"] == 2 and o["scaling"] == "strong" and o["config"]["collectives"] > 0
    assert "bond-sharded" in o["config"]["parallelism"] and abs(o["config"]["norm_after"] - 1) < 1e-10
