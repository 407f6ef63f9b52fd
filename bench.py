#!/usr/bin/env python3
"""bench.py -- TDVP sweeps/s of the MI355X engine on BASELINE.json's configs.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C4] [--max-seconds S]

One "step" = one half-sweep of one-site TDVP over all L sites of a synthetic
full-rank MPS under a synthetic Hermitian MPO (SURVEY.md 8d inputs), i.e. L
local exp(-i H_eff dt/2), L-1 QR gauge moves, L-1 environment updates, L-1 local
exp(+i K_eff dt/2) (reference: one directional pass of propagate_along_sweep,
_mps_cls.py:482-500, :798-1014).  Sweeps alternate direction (forward, backward,
...), two of them are one PyTDSCF time step.  All tensors are generated on /
resident in HBM before the timed region.

Wall budget.  A C4 sweep takes about a minute, so the requested --steps /
--warmup are an upper bound: after the first warm-up sweep the run knows the
sweep time and fits `warmup + steps` into --max-seconds (default 400 s, env
MITDVP_BENCH_BUDGET; the CPU-baseline leg and the start-up are counted).  The
JSON line reports the sweeps actually run ("steps", "warmup") next to
"steps_requested" / "warmup_requested"; at least one warm-up and one timed
sweep always run.

N > 1: either launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE
in the environment) or plainly as `python bench.py --gpus N`, in which case this
process starts the N ranks itself (fresh children, before anything touches the
GPU) and forwards rank 0's JSON line.  --parallel:
  sites     (default when every rank gets >= 4 sites) contiguous site ranges per
            rank, "scaling": "strong": all ranks sweep their blocks at once, the two
            sites facing each other across a rank boundary are updated together
            through the pseudo-inverse of the joint bond matrix; ONE library call
            per time step and rank (mitdvp_shard_step): the halo is the library's own
            grouped ncclSend / ncclRecv of device buffers between chain neighbours
            (RCCL over xGMI), no collective; torch.distributed (gloo) is the control
            plane only.  The reference's real-space parallel TDVP (_mps_parallel.py):
            approximate, deviation from the serial sweep ~ dt^2; the JSON line carries
            "approximate": true and the norm / energy drift of the run ("accuracy").
  tp        (when D % N == 0) ONE sweep shared by all GPUs, "scaling":
            "strong": every H_eff / K_eff apply and environment update is sharded
            over the bra-side bond index (each rank contracts D/N rows of the
            environment block), combined by one RCCL all-gather / all-reduce per
            contraction chain over xGMI, issued by the library itself on the
            engine's stream (falls back to the torch.distributed callback);
            Krylov algebra and QR are replicated.  Exact.
  replicas  N independent trajectories (SURVEY 8e "fallback"), "scaling": "weak",
            no data-path collective.

Prints ONE JSON line on rank 0.
"""

from __future__ import annotations

import time

T_PROCESS_START = time.perf_counter()

import argparse  # noqa: E402
import json  # noqa: E402
import os  # noqa: E402
import socket  # noqa: E402
import subprocess  # noqa: E402
import sys  # noqa: E402


def one_socket_physical_cores():
    """(socket id, one logical CPU per physical core of that socket) among the CPUs this process may use: the socket
    that offers the most cores (the CPU baseline runs pinned to one socket, all its physical cores)."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return None, []
    by_pkg = {}
    for c in allowed:
        base = f"/sys/devices/system/cpu/cpu{c}/topology/"
        try:
            pkg = int(open(base + "physical_package_id").read())
            core = int(open(base + "core_id").read())
        except (OSError, ValueError):
            pkg, core = 0, c
        by_pkg.setdefault(pkg, {}).setdefault(core, c)  # first logical CPU of each physical core
    if not by_pkg:
        return None, []
    pkg = max(by_pkg, key=lambda k: (len(by_pkg[k]), -k))
    return pkg, sorted(by_pkg[pkg].values())


CPU_WORKER_FLAG = "--cpu-baseline-worker"
CPU_PLACEMENT = None
if CPU_WORKER_FLAG in sys.argv:
    # The CPU-baseline leg runs in a process of its own, pinned BEFORE NumPy / OpenBLAS start their thread pool (threads
    # inherit the affinity of the thread that creates them; the pool of an already running process cannot be re-pinned).
    _pkg, _cpus = one_socket_physical_cores()
    if _cpus:
        os.sched_setaffinity(0, _cpus)
        for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
            os.environ[_v] = str(len(_cpus))
    CPU_PLACEMENT = {"pinned": bool(_cpus), "socket": _pkg, "affinity_cpus": len(_cpus),
                     "note": "one logical CPU per physical core of one socket (os.sched_setaffinity before NumPy / OpenBLAS "
                             "start their threads; OPENBLAS_NUM_THREADS = that count)"}

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X FP64 matrix peak (spec), 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E (MI355X_MICROARCH.md)

WORKLOADS = {
    # name: (L, d, D, M, dt_au, integrator, description)
    "C2": (10, 10, 32, 6, 2.0, "lanczos", "Henon-Heiles-like L=10 d=10 D=32 M=6"),
    "C3": (6, 32, 128, 16, 1.0, "lanczos", "H2CO-like grid MPO L=6 d=32 D=128 M=16"),
    "C4": (64, 16, 1024, 32, 0.5, "lanczos", "synthetic exciton chain L=64 d=16 D=1024 M=32"),
    "C5": (128, 4, 512, 16, 0.5, "arnoldi", "Liouville-space spin chain L=128 d=4=2x2 D=512 M=16, non-Hermitian, Arnoldi, conserve_norm=False"),
}


def flops_heff(dl, d, dr, ml, mr):
    return 8.0 * (dl * dl * ml * d * dr + dl * dr * ml * mr * d * d + dl * dr * dr * mr * d)


def sweep_flops_estimate(L, d, D, M, kh=7.0):
    from pytdscf_amd.mps import bond_dims

    return sum(flops_heff(a, d, b, M, M) for a, b in bond_dims([d] * L, D)) * (kh + 1.0)


def cpu_baseline_is_sampled(L, d, D, M):
    """Large workloads time single kernels of the oracle and extrapolate; small ones run it end to end."""
    return sweep_flops_estimate(L, d, D, M) > 2e12


def cpu_baseline_seconds_estimate(L, d, D, M, short=False):
    """Wall time the cpu_baseline leg needs on the GPU box's host (reserved out of the budget)."""
    if short and not cpu_baseline_is_sampled(L, d, D, M):
        return 12.0
    if cpu_baseline_is_sampled(L, d, D, M):
        # one chunked H_eff apply + one K_eff apply + one QR at the interior shape: ~25 s at C4 on 64 BLAS threads
        return 10.0 + 3.0 * flops_heff(D, d, D, M, M) / 1e12
    return 20.0


def cpu_baseline(L, d, D, M, kh, kk, n_threads, dt, budget_s=15.0, short=False):
    """Oracle (NumPy/OpenBLAS zgemm + LAPACK QR) timed on the host cores on a bounded
    sample: one H_eff apply, one K_eff apply and one QR gauge move at the interior
    site shape; extrapolated over the chain with per-site flop ratios and the
    Krylov counts measured on the GPU run."""
    from oracle import tdvp_oracle as orc

    bd = orc.bond_dims([d] * L, D)
    dl, dr = max(b[0] for b in bd), max(b[1] for b in bd)
    ml = mr = M
    rng = np.random.default_rng(0)

    def crandn(*s):
        return rng.standard_normal(s) + 1j * rng.standard_normal(s)

    f_int = flops_heff(dl, d, dr, ml, mr)
    # (short: the secondary legs of the bench line have seconds, not tens of seconds: mid-size workloads are sampled too)
    if cpu_baseline_is_sampled(L, d, D, M) or (short and sweep_flops_estimate(L, d, D, M) > 2e11):  # an end-to-end oracle sweep would take minutes to hours on the host
        # chunk the apply over the left bond so the intermediates stay ~1 GB
        Lb, Rb, psi = crandn(dl, ml, dl), crandn(dr, mr, dr), crandn(dl, d, dr)
        W = crandn(ml, d, d, mr)
        ch = max(1, min(dl, int(6.4e7 // (ml * d * dr))))
        t0 = time.perf_counter()
        orc.heff_apply_chunked(Lb, W, Rb, psi, ch)
        t_h = time.perf_counter() - t0
        sv = crandn(dl, dr)
        t0 = time.perf_counter()
        orc.keff_apply(Lb, Rb, sv)
        t_k = time.perf_counter() - t0
        t0 = time.perf_counter()
        orc.qr_psi2Asigma(psi)
        t_q = time.perf_counter() - t0
        sample = f"1 H_eff apply ({t_h:.1f}s) + 1 K_eff apply ({t_k:.1f}s) + 1 QR ({t_q:.1f}s) at ({dl},{d},{dr}), M={M}; extrapolated over L={L}"
        tot = 0.0
        for p, (a, b) in enumerate(bd):
            mlp = 1 if p == 0 else M
            mrp = 1 if p == L - 1 else M
            r = flops_heff(a, d, b, mlp, mrp) / f_int
            tot += (kh + 1.0) * t_h * r  # k_H applies + one environment update
            if p < L - 1:
                tot += kk * t_k * (b / dr) ** 3 + t_q * (a * d * b * b) / (dl * d * dr * dr)
        return dict(value=1.0 / tot, unit="sweeps/s", cores=n_threads, kind="port", sample=sample)
    # small workloads: run the oracle end to end
    mpo = orc.synthetic_mpo(L, d, M, seed=0)
    mps = orc.synthetic_mps([d] * L, D, seed=1)
    st = orc.OracleMPS(mps, mpo)
    st.build_right_envs()
    if not short:
        st.sweep(dt, True)
        st.sweep(dt, False)
    n = 0
    t0 = time.perf_counter()
    while True:
        st.sweep(dt, True)
        st.sweep(dt, False)
        n += 2
        if time.perf_counter() - t0 > budget_s or n >= 20:
            break
    el = time.perf_counter() - t0
    return dict(value=n / el, unit="sweeps/s", cores=n_threads, kind="port", sample=f"{n} full sweeps of the oracle, {el:.1f}s")


def workload_name(L, d, D, M):
    for k, v in WORKLOADS.items():
        if v[:4] == (L, d, D, M):
            return k
    return None


def committed_traffic(name, L, d, D, M, form="chain"):
    """(HBM-side bytes per H_eff apply, the committed file they come from) -- PMC passes run beside the bench, NOT a
    measurement of this run (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs, FETCH doubled per
    MI355X_MICROARCH.md).  Newest round first, and only files of the apply FORM that ran (`form`: "chain" = the
    three-stage chain, "edge" = the two-product form; the one-launch small-bond kernel has its own files); (None, None)
    when no pass of that form and shape is committed."""
    rounds = range(9, 0, -1)
    if name == "heff":
        wl = workload_name(L, d, D, M)
        cands = ["r%02d_heff_traffic_%s_%s.json" % (r, wl, form) for r in rounds] if wl else []
        if form == "chain":  # older files: per shape (round 3) or the C4 default (rounds 1-3), all of the chain form
            cands += ["r%02d_heff_traffic_D%d_d%d_M%d.json" % (r, D, d, M) for r in (4, 3)]
            cands += ["r%02d_heff_traffic.json" % r for r in (3, 2, 1)]
    else:
        cands = ["r%02d_%s_traffic.json" % (r, name) for r in rounds]
    for fn in cands:
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", fn)))
        except (OSError, ValueError):
            continue
        if (t.get("shape") in ({"D": D, "d": d, "M": M}, {"L": L, "D": D, "d": d, "M": M})
                or f"L={L} d={d} D={D} M={M}" in str(t.get("workload", ""))):
            for k in ("total_bytes", "bytes_per_heff_apply", "zgemm_NN_NT_bytes_per_heff_apply_upper_bound"):
                if k in t:
                    return t[k], "profiles/" + fn
    return None, None


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def cpu_baseline_pinned(L, d, D, M, kh, kk, dt, budget_s=15.0, short=False, timeout_s=240.0):
    """The cpu_baseline leg in a child process pinned to one socket's physical cores (see CPU_WORKER_FLAG above); the
    child never touches the GPU.  Returns the record of `cpu_baseline` plus placement, thread count and CPU model."""
    req = json.dumps({"L": L, "d": d, "D": D, "M": M, "kh": kh, "kk": kk, "dt": dt, "budget_s": budget_s, "short": short})
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), CPU_WORKER_FLAG, req], capture_output=True, text=True,
                           timeout=timeout_s, cwd=ROOT)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            raise RuntimeError(f"rc={r.returncode}: {r.stderr.strip()[-300:]}")
        return json.loads(lines[-1])
    except Exception as e:  # noqa: BLE001 -- the GPU numbers stand on their own
        return {"error": f"cpu baseline worker failed ({type(e).__name__}: {e})"}


def cpu_worker_main(req_json):
    q = json.loads(req_json)
    nthr = blas_threads()
    rec = cpu_baseline(q["L"], q["d"], q["D"], q["M"], q["kh"], q["kk"], nthr, q["dt"], budget_s=q["budget_s"], short=q["short"])
    rec["host_cpus"] = os.cpu_count()
    rec["cpu_model"] = cpu_model()
    rec.update(CPU_PLACEMENT or {"pinned": False})
    sys.stdout.write(json.dumps(rec) + "\n")


def blas_threads():
    nthr = os.cpu_count() or 1
    try:  # the threads the BLAS behind NumPy actually runs (OpenBLAS caps at its build-time maximum)
        import threadpoolctl

        blas = [x["num_threads"] for x in threadpoolctl.threadpool_info() if x.get("user_api") == "blas"]
        if blas:
            nthr = max(blas)
    except Exception:
        pass
    return nthr


def roofline_report(cnt, L, d, D, M, gemm_mode, el_s, profile_in_timed, tp=False):
    """(roofline, breakdown_ms, mean Krylov dimension of the site / bond exponentials) from the engine's counters."""
    kh = cnt["n_heff"] / max(cnt["n_exp_site"], 1)
    kk = cnt["n_keff"] / max(cnt["n_exp_bond"], 1)
    alg = cnt["heff_flops"] / max(cnt["heff_ms"], 1e-9) / 1e9  # algorithmic TFLOP/s (8 flop per complex MAC)
    # the 3M (Karatsuba) complex product executes 6 real flop per complex MAC, the 4M product all 8:
    # the roofline fraction is what the matrix cores actually execute over their peak
    # zero (c, t) blocks of W are skipped by the block-sparse W stage: those flops are not executed either
    done_share = 1.0 - cnt.get("heff_flops_skipped", 0.0) / max(cnt["heff_flops"], 1.0)
    executed = alg * done_share * (0.75 if gemm_mode == "3m" else 1.0)
    # small-bond regime (C2, D < 128): the apply is memory / latency bound, the
    # roofline that applies is HBM: algorithmic bytes B_H per apply over the apply time
    small = D < 128
    bytes_apply = 16.0 * (2 * D * d * D + 2 * D * D * M + M * d * d * M)  # interior site, SURVEY 8d B_H
    ach_gbs = bytes_apply * cnt["n_heff"] / max(cnt["heff_ms"], 1e-9) / 1e6  # GB/s
    form = "edge" if cnt.get("n_heff_edge", 0.0) > 0.5 * max(cnt["n_heff"], 1) else "chain"
    traffic, traffic_src = committed_traffic("c2" if small else "heff", L, d, D, M, form)
    if small:
        roof = {
            "bound": "hbm",
            "kernel": "k_small_site (one launch per local exponential: H_eff applies, Krylov algebra, k x k exponential, convergence test)",
            "achieved": ach_gbs,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": ach_gbs / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": traffic_src,
            "bytes_per_apply": bytes_apply,
            "ms_per_apply": cnt["heff_ms"] / max(cnt["n_heff"], 1),
            "n_apply": cnt["n_heff"],
            "tflops": alg,
            "note": ("latency bound: a site is a few hundred KB; achieved = algorithmic bytes B_H per H_eff apply x applies / HIP-event "
                     "time of the site exponentials (which also holds their Krylov algebra and grid-wide exchanges)"),
        }
    else:
        roof = {
            "bound": "mfma",
            "kernel": ("zgemm_kernel (H_eff apply, edge form = 2 launches with a reducing epilogue: psi.R^T (x) W, L.psi (x) W)" if form == "edge"
                       else "zgemm_kernel (H_eff apply = 3 launches: L.psi, W., .R)") + (" -- per GPU, this rank's bond shard" if tp else ""),
            "apply_form": form,
            "achieved": executed,
            "peak": FP64_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": executed / FP64_MFMA_PEAK_TFLOPS,
            "algorithmic_tflops": alg,
            "achieved_algorithmic": alg,
            "traffic": traffic,
            "traffic_source": traffic_src,
            "traffic_note": "bytes per apply from the committed PMC passes named in traffic_source, not measured in this run",
            "flops_per_apply": cnt["heff_flops"] / max(cnt["n_heff"], 1),
            "ms_per_apply": cnt["heff_ms"] / max(cnt["n_heff"], 1),
            "stage_ms_per_apply": [x / max(cnt["n_heff"], 1) for x in cnt["heff_stage_ms"]],
            # the roofline of each stage's kernel by itself: what that stage executes (identity blocks trimmed, zero blocks
            # skipped, tile padding counted; x 6 / 8 for the 3M product) over its own HIP-event time.  Stages 0 and 2 are
            # the two large launches of zgemm_kernel (the dominant kernel); stage 1 is the block-sparse W stage (chain
            # form) or the transpose (edge form)
            "stage_executed_tflops": [f * (0.75 if gemm_mode == "3m" else 1.0) / max(ms, 1e-9) / 1e9
                                      for f, ms in zip(cnt.get("heff_stage_flops", [0, 0, 0]), cnt["heff_stage_ms"])],
            "dominant_kernel_frac": max([f * (0.75 if gemm_mode == "3m" else 1.0) / max(ms, 1e-9) / 1e9
                                         for f, ms in zip(cnt.get("heff_stage_flops", [0, 0, 0]), cnt["heff_stage_ms"])][::2]) / FP64_MFMA_PEAK_TFLOPS,
            "n_apply": cnt["n_heff"],
            "complex_product": gemm_mode,
            "executed_share_of_algorithmic": done_share * (0.75 if gemm_mode == "3m" else 1.0),
            "w_stage": ("block-sparse: zero blocks of the finite-state-machine MPO skipped; identity blocks of the two "
                        "environments short-circuited in stages S1 / S3" if done_share < 0.999 else "dense"),
            "note": ("achieved / frac = flop the matrix cores EXECUTE per second: the 3M (Karatsuba) complex product "
                     "runs 6 real flop per complex MAC, the W stage skips the zero blocks of the MPO and stages S1 / S3 the identity "
                     "blocks of the environments; "
                     "algorithmic_tflops counts the 8 flop of the textbook dense product (SURVEY 8d F_H) over the "
                     "same HIP-event time") if gemm_mode == "3m"
                    else "4M complex product: executed = algorithmic flops",
        }
    brk = {
        "heff": cnt["heff_ms"], "env": cnt["env_ms"], "keff": cnt["keff_ms"], "qr": cnt["qr_ms"],
        "krylov_vec": cnt["krylov_vec_ms"], "wall": 1e3 * el_s,
        "phases_from": "the timed sweeps" if profile_in_timed else "a profiled repeat of the timed sweeps (event overhead kept out of the timed region)",
        "env_tflops": cnt["env_flops"] / max(cnt["env_ms"], 1e-9) / 1e9,
        "keff_tflops": cnt["keff_flops"] / max(cnt["keff_ms"], 1e-9) / 1e9,
        "qr_tflops": cnt["qr_flops"] / max(cnt["qr_ms"], 1e-9) / 1e9,
        "launches": cnt["n_launch"],
        "host_waits": cnt.get("n_host_waits", 0.0),
    }
    return roof, brk, kh, kk


# Per-phase HIP-event timing (roofline / breakdown) costs two event records per phase: nothing at C4 (0.01 %), a third
# of the run in the launch-bound small-bond regime, 15 % at C3 (D = 128), 1.7 % at C5 (D = 512; 0.4614 vs 0.4691 sweeps/s).
# Below this bond dimension the timed region runs unprofiled and the same sweeps are repeated afterwards, profiled, only
# for the breakdown.
PROFILE_IN_TIMED_MIN_D = int(os.environ.get("MITDVP_BENCH_PROFILE_MIN_D", "1024"))


def ensemble_leg(name, device, n_steps):
    """Aggregate throughput of B independent replicas of a small-bond workload on one GPU (TDVPEnsemble: every replica
    on its own slice of the compute units).  HBM roofline: algorithmic bytes B_H per apply x applies of ALL replicas / wall."""
    from pytdscf_amd import synthetic as syn
    from pytdscf_amd import TDVPEnsemble

    L, d, D, M, dt, integ, desc = WORKLOADS[name]
    mpo = syn.synthetic_mpo(L, d, M, seed=0)
    bytes_apply = 16.0 * (2 * D * d * D + 2 * D * D * M + M * d * d * M)
    out = {"unit": "sweeps/s", "steps_per_replica": 2 * n_steps, "replicas": {}}
    for B in (8, 16):
        ens = TDVPEnsemble(B, L, device=device, integrator=integ)
        try:
            ens.set_mpo(mpo)
            for r, e in enumerate(ens.engines):
                e.init_random([d] * L, D, seed=1 + r)
            ens.propagate(dt, 2)
            for e in ens.engines:
                e.counters_reset()
            t0 = time.perf_counter()
            ens.propagate(dt, n_steps)  # returns with every replica's stream drained
            el = time.perf_counter() - t0
            napply = sum(e.counters()["n_heff"] for e in ens.engines)
            out["replicas"][str(B)] = {
                "value": B * 2 * n_steps / el, "per_replica": 2 * n_steps / el, "cu_per_replica": ens.cu_per_replica,
                "hbm_GBs_algorithmic": bytes_apply * napply / el / 1e9,
                "norm_minus_1_max": max(abs(e.norm() - 1.0) for e in ens.engines),
            }
        finally:
            ens.close()
    best = max(out["replicas"].values(), key=lambda r: r["value"])
    out["value"] = best["value"]
    out["note"] = ("independent trajectories on disjoint compute-unit ranges of one GPU (hipExtStreamCreateWithCUMask), one "
                   "mitdvp_ensemble_step call per batch of time steps; bit-identical to the same engines run one at a time")
    return out


def energy_at_centre(eng):
    """<Psi|H|Psi> = <psi_c | H_eff psi_c> at the centre site, through the sweep's own apply kernels and the environment
    blocks the last sweep left around the centre (valid at either end of the chain; `expectation` rebuilds a whole chain
    of right blocks and wants the centre at site 0 like the reference's, _mps_cls.py:540-612)."""
    sig, _ = eng.heff_apply_center()
    c = next(p for p in range(eng.nsite) if eng.get_site_shape(p)[3] == 0)
    return float(np.vdot(eng.get_site(c), sig).real)


def secondary_leg(name, device, gemm_mode, seconds, steps_req, with_cpu, note):
    """One short single-GPU run of another BASELINE config after the headline one: the same measurement (warm-up, timed
    sweeps between synchronisations, counters -> roofline / breakdown, the oracle on the host cores), a few seconds of
    sweeps.  Returns the record for the `secondary` key of the JSON line."""
    from pytdscf_amd import synthetic as syn
    from pytdscf_amd import TDVPEngine

    t_leg = time.perf_counter()
    L, d, D, M, dt, integ, desc = WORKLOADS[name]
    liouville = integ == "arnoldi"
    mpo_cores = syn.synthetic_liouvillian_mpo(L, M, seed=0, gamma=0.002) if liouville else syn.synthetic_mpo(L, d, M, seed=0)
    eng = TDVPEngine(L, device=device, integrator=integ, conserve_norm=not liouville)
    try:
        eng.set_mpo(mpo_cores)
        eng.init_random([d] * L, D, seed=1)
        e0 = eng.expectation().real
        fwd = [True]

        def sweep():
            eng.sweep(dt, fwd[0])
            fwd[0] = not fwd[0]

        t1 = time.perf_counter()
        sweep()
        eng.norm()
        t_first = time.perf_counter() - t1
        t1 = time.perf_counter()
        sweep()
        eng.norm()
        t_sweep = time.perf_counter() - t1
        warm = 2
        profile_in_timed = D >= PROFILE_IN_TIMED_MIN_D
        reserve = (cpu_baseline_seconds_estimate(L, d, D, M, short=True) if with_cpu else 0.0) + 2.0
        left = seconds - (time.perf_counter() - t_leg) - reserve
        afford = int(left / max(t_sweep, 1e-9) / (1 if profile_in_timed else 2))
        steps = max(2, min(steps_req, (afford // 2) * 2))
        eng.counters_reset()
        eng.set_profiling(profile_in_timed)
        eng.norm()
        t0 = time.perf_counter()
        for _ in range(steps):
            sweep()
        nrm = eng.norm()
        el = time.perf_counter() - t0
        if profile_in_timed:
            cnt = eng.counters()
        eng.set_profiling(False)
        e1 = energy_at_centre(eng)  # the state after exactly `steps` timed sweeps (the profiled repeat below moves it on)
        if not profile_in_timed:
            eng.counters_reset()
            eng.set_profiling(True)
            for _ in range(steps):
                sweep()
            eng.norm()
            cnt = eng.counters()
            eng.set_profiling(False)
        roof, brk, kh, kk = roofline_report(cnt, L, d, D, M, gemm_mode, el, profile_in_timed)
        rec = {
            "workload": f"{name}: {desc}",
            "value": steps / el, "unit": "sweeps/s", "steps": steps, "warmup": warm, "ms_per_step": 1e3 * el / steps,
            "first_sweep_s": t_first,
            "config": {"L": L, "d": d, "D": D, "M": M, "dt_au": dt, "thresh_sil": 1e-9, "integrator": integ,
                       "mean_krylov_site": round(kh, 2), "mean_krylov_bond": round(kk, 2),
                       "norm_after": nrm, "energy_before": e0, "energy_after": e1},
            "roofline": roof,
            "breakdown_ms": brk,
        }
    finally:
        eng.close()
    if D < 128 and seconds - (time.perf_counter() - t_leg) > 10.0:
        # small-bond regime: the data-parallel axis is the ensemble of trajectories: B replicas on
        # disjoint compute-unit ranges, one library call per batch of time steps (mitdvp_ensemble_step)
        try:
            rec["ensemble"] = ensemble_leg(name, device, n_steps=max(10, steps // 2))
        except Exception as e:  # noqa: BLE001 -- the single-engine record stands on its own
            rec["ensemble"] = {"error": f"{type(e).__name__}: {e}"}
    if with_cpu:
        note(f"{name}: timing the CPU baseline (oracle on the host cores)")
        rec["cpu_baseline"] = cpu_baseline_pinned(L, d, D, M, kh, kk, dt, budget_s=6.0, short=True, timeout_s=90.0)
    rec["leg_s"] = time.perf_counter() - t_leg
    return rec


def compact_record(rec):
    """A secondary leg's record without its prose (the headline `roofline` carries the notes once): the whole `secondary`
    object stays under 8 KB so that the JSON line fits the driver's stdout tail."""
    drop = {"note", "traffic_note", "w_stage", "kernel", "phases_from", "sample"}

    def strip(x):
        if isinstance(x, dict):
            return {k: strip(v) for k, v in x.items() if k not in drop}
        if isinstance(x, float):
            return float(f"{x:.9g}")
        if isinstance(x, list):
            return [strip(v) for v in x]
        return x

    out = strip(rec)
    roof = out.get("roofline") if isinstance(out, dict) else None
    if isinstance(roof, dict) and roof.get("peak"):  # keep the identity frac = achieved / peak exact after the rounding
        roof["frac"] = roof["achieved"] / roof["peak"]
    if isinstance(rec, dict) and isinstance(rec.get("cpu_baseline"), dict) and "sample" in rec["cpu_baseline"]:
        out["cpu_baseline"]["sample"] = rec["cpu_baseline"]["sample"][:90]
    return out


def secondary_summary(sec):
    """{workload: {value, frac, ...}} of the secondary legs, carried inside the headline `config` (the driver's parsed
    record keeps `config`; extra top-level keys are not guaranteed to survive)."""
    out = {}
    for w, r in sec.items():
        if not isinstance(r, dict) or "value" not in r:
            out[w] = {"error": str(r.get("error", "failed"))[:80] if isinstance(r, dict) else "failed"}
            continue
        roof = r.get("roofline", {})
        out[w] = {"value": float(f"{r['value']:.5g}"), "unit": "sweeps/s", "frac": float(f"{roof.get('frac', 0.0):.4g}"),
                  "bound": roof.get("bound"), "steps": r.get("steps")}
        cb = r.get("cpu_baseline")
        if isinstance(cb, dict) and "value" in cb:
            out[w]["cpu_value"] = float(f"{cb['value']:.4g}")
        ens = r.get("ensemble")
        if isinstance(ens, dict) and "replicas" in ens:
            out[w + "_ensemble"] = {b: float(f"{v['value']:.5g}") for b, v in ens["replicas"].items()}
    return out


def plan_sweeps(budget_s, elapsed_s, t_sweep, warm_done, warm_req, steps_req, reserve_s):
    """How many more warm-up sweeps and how many timed sweeps fit the wall budget.

    `elapsed_s` since process start, `t_sweep` the (max over ranks) time of the last sweep,
    `reserve_s` what must stay for the CPU baseline and the tail.  Always >= 1 timed sweep."""
    afford = int((budget_s - elapsed_s - reserve_s) // max(t_sweep, 1e-9))
    more_warm = max(0, warm_req - warm_done)
    if afford >= more_warm + steps_req:
        return more_warm, steps_req
    return 0, max(1, min(steps_req, afford))  # the warm-up gives way first


def self_launch(args, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh children (this
    process has not imported torch or touched the GPU), forward rank 0's stdout, exit with the
    worst child return code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    base = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(args.gpus))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    base["MITDVP_BENCH_T0"] = repr(time.time() - (time.perf_counter() - T_PROCESS_START))
    procs = []
    try:
        for r in range(args.gpus):
            env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, cwd=os.getcwd(),
                                          stdout=None if r == 0 else subprocess.DEVNULL))
        rc = 0
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0:
                    rc = rc or code
                    for q in pending:  # a dead rank leaves the others blocked in a collective
                        q.kill()
            time.sleep(0.2)
        return rc
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
            p.wait()


def main():
    if CPU_WORKER_FLAG in sys.argv:
        cpu_worker_main(sys.argv[sys.argv.index(CPU_WORKER_FLAG) + 1])
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("MITDVP_WORKLOAD", "C4"), choices=sorted(WORKLOADS))
    ap.add_argument("--dt", type=float, default=None, help="time step in a.u. (default per workload)")
    ap.add_argument("--max-seconds", type=float, default=float(os.environ.get("MITDVP_BENCH_BUDGET", "400")),
                    help="wall budget for the whole run; steps / warmup are cut to fit (0 = no limit)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--secondary", default=os.environ.get("MITDVP_BENCH_SECONDARY", "auto"),
                    help="other BASELINE configs run as short single-GPU legs after the headline one and reported under "
                         "the `secondary` key: auto (C2, C3, C5), none, or a comma list")
    ap.add_argument("--secondary-seconds", type=float, default=float(os.environ.get("MITDVP_BENCH_SECONDARY_BUDGET", "100")),
                    help="wall budget of all secondary legs together (beside --max-seconds)")
    ap.add_argument("--parallel", default=os.environ.get("MITDVP_PARALLEL", "auto"), choices=["auto", "sites", "tp", "replicas"])
    args = ap.parse_args()
    if args.steps < 1 or args.warmup < 0 or args.gpus < 1:
        ap.error("need --steps >= 1, --warmup >= 0, --gpus >= 1")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args, sys.argv[1:]))

    # stdout carries ONE JSON line and nothing else: whatever the libraries below print on file descriptor 1
    # (gloo's "Rank 0 is connected to ..." banner, RCCL notices) is sent to stderr, the line goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    # wall clock of the whole job: a self-launched rank inherits the parent's start
    t_start = T_PROCESS_START
    if "MITDVP_BENCH_T0" in os.environ:
        t_start = time.perf_counter() - (time.time() - float(os.environ["MITDVP_BENCH_T0"]))

    def elapsed():
        return time.perf_counter() - t_start

    from pytdscf_amd.dist import Comm, replica_throughput

    # Site-range sharding moves its halo with the library's own RCCL communicator (ncclSend / ncclRecv between chain
    # neighbours, csrc/shard.hip); torch.distributed is the control plane only (rendezvous, the ncclUniqueId, barriers,
    # scalar reductions) and runs over gloo there, so that the process holds ONE RCCL instance on its GPU.
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    L_w = WORKLOADS[args.workload][0]
    if env_world > 1 and (args.parallel == "sites" or (args.parallel == "auto" and L_w >= 4 * env_world)):
        os.environ.setdefault("MITDVP_DIST_BACKEND", "gloo")
    comm = Comm()  # RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment
    rank, world = comm.rank, comm.world
    if comm.gpu is None:
        sys.exit("bench.py needs a GPU: the MI355X engine has no CPU fallback")
    args.gpus = world

    from pytdscf_amd import synthetic as syn  # synthetic inputs; oracle/ is imported by the cpu_baseline leg only
    from pytdscf_amd import TDVPEngine
    from pytdscf_amd.engine import get_gemm_mode

    gemm_mode = get_gemm_mode()

    L, d, D, M, dt, integ, desc = WORKLOADS[args.workload]
    if args.dt is not None:
        dt = args.dt

    def note(msg):
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')} +{elapsed():.0f}s] {msg}", file=sys.stderr, flush=True)

    liouville = integ == "arnoldi"
    mpo_cores = syn.synthetic_liouvillian_mpo(L, M, seed=0, gamma=0.002) if liouville else syn.synthetic_mpo(L, d, M, seed=0)
    mode = args.parallel
    if world == 1:
        mode = "single"
    elif mode == "auto":
        # site ranges (the partitioning north_star names; approximate like the reference's) when every rank
        # gets at least four sites, else exact bond sharding, else independent replicas
        if L >= 4 * world:
            mode = "sites"
        elif D % world == 0 and D >= 8 * world:
            mode = "tp"
        else:
            mode = "replicas"
    if comm.shared_gpu:  # several processes on one GPU (rehearsals): no persistent kernels, they need every CU
        os.environ.setdefault("MITDVP_SMALL_KERNELS", "0")
    ss = None
    if mode == "sites":
        from pytdscf_amd.parallel_sites import SiteShardedTDVP

        ok = 1.0
        try:
            ss = SiteShardedTDVP(comm, mpo_cores, dims=[d] * L, bond_dim=D, seed=1, integrator=integ, conserve_norm=not liouville)
            if not ss.selftest():
                raise RuntimeError("neighbour send / recv self-test failed")
        except Exception as e:  # noqa: BLE001 -- the verdict below is common to all ranks
            note(f"site sharding unavailable ({type(e).__name__}: {e})")
            ok = 0.0
        if comm.min_over_ranks(ok) < 0.5:
            if ss is not None:
                ss.close()
                ss = None
            mode = "tp" if (D % world == 0 and D >= 8 * world) else "replicas"
            note(f"falling back to --parallel {mode}")
    eng = None
    collectives = None
    e0 = None
    if mode != "sites":
        eng = TDVPEngine(L, device=comm.gpu, integrator=integ, conserve_norm=not liouville)
        eng.set_mpo(mpo_cores)
        # tp: every rank holds the same replicated state (same seed); replicas: one trajectory per rank
        eng.init_random([d] * L, D, seed=1 + (rank if mode == "replicas" else 0))
    if mode == "tp":
        from pytdscf_amd.dist import attach_parallel, attach_parallel_native

        # default: the library's own RCCL calls on the engine's stream (no host synchronisation per
        # collective); the torch.distributed callback is the fallback, and the only form when the
        # ranks share one GPU (gloo, staged through the host)
        want = os.environ.get("MITDVP_COLLECTIVES", "native" if comm.backend == "nccl" else "torch")
        try:
            if want == "native":
                try:
                    attach_parallel_native(eng, comm)
                    collectives = "rccl-native"
                except Exception as e:  # noqa: BLE001 -- any failure: the verdict below is common to all ranks
                    note(f"native RCCL collectives unavailable ({e}); using the torch.distributed callback")
                    want = "torch"
            ok_native = comm.min_over_ranks(1.0 if collectives == "rccl-native" else 0.0) > 0.5
            if not ok_native:
                attach_parallel(eng, comm)  # includes a collective self-test with a verdict common to all ranks
                collectives = f"torch.distributed/{comm.backend}"
        except RuntimeError as e:
            # every rank gets here together: run N independent trajectories instead
            note(f"{e}; falling back to independent replicas")
            mode = "replicas"
            eng.init_random([d] * L, D, seed=1 + rank)
    if eng is not None:
        e0 = eng.expectation().real  # also builds nothing persistent; forces setup to finish
    elif ss is not None:
        e0 = ss.expectation().real  # folded rank by rank on the devices (collective), outside the timed region

    # one "unit" of work: a half-sweep; the site-sharded step is two half-sweeps + the junction updates
    unit = 2 if mode == "sites" else 1
    state = {"forward": True}

    def run_unit():
        if mode == "sites":
            ss.step(dt)
        else:
            eng.sweep(dt, state["forward"])
            state["forward"] = not state["forward"]

    def sync_norm():  # synchronises the engine's stream; the centre tensor's norm (= the state's, every block carries it)
        return ss.block.norm() if mode == "sites" else eng.norm()

    meas = ss.block if mode == "sites" else eng  # the engine whose counters feed the roofline (rank 0's block)

    note(f"{args.workload} set up on device {comm.gpu} (L={L} d={d} D={D} M={M}), mode {mode}" + (f", <H>={e0:.9f}" if e0 is not None else ""))

    barrier = comm.barrier
    budget = args.max_seconds if args.max_seconds > 0 else float("inf")
    with_cpu = not args.no_cpu_baseline and world == 1
    reserve = (cpu_baseline_seconds_estimate(L, d, D, M) if with_cpu else 0.0) + 8.0

    # ---- warm-up: at least one unit, which also measures the sweep time ----
    steps_req = max(unit, (args.steps // unit) * unit)
    warm_done, steps = 0, steps_req
    warm_target = max(unit, (args.warmup // unit) * unit)
    while warm_done < warm_target:
        t1 = time.perf_counter()
        run_unit()
        sync_norm()
        t_sweep = comm.max_over_ranks(time.perf_counter() - t1) / unit
        warm_done += unit
        note(f"warm-up sweep {warm_done} done in {t_sweep:.2f}s per sweep")
        more, steps = plan_sweeps(budget, comm.max_over_ranks(elapsed()), t_sweep, warm_done, warm_target, steps_req, reserve)
        more, steps = (more // unit) * unit, max(unit, (steps // unit) * unit)
        warm_target = warm_done + more
    if steps != args.steps or warm_done != args.warmup:
        note(f"wall budget {budget:.0f}s: running {warm_done} warm-up + {steps} timed sweeps "
             f"(requested {args.warmup} + {args.steps})")
    sync_norm()
    meas.counters_reset()
    profile_in_timed = D >= PROFILE_IN_TIMED_MIN_D  # below: an unprofiled timed region + a profiled repeat
    meas.set_profiling(profile_in_timed)
    barrier()
    t0 = time.perf_counter()
    for i in range(0, steps, unit):
        run_unit()
        if steps > unit and t_sweep > 5.0:
            sync_norm()
            note(f"timed sweep {i + unit}/{steps} done")
    nrm = sync_norm()
    barrier()
    el_rank = time.perf_counter() - t0
    if mode in ("tp", "sites"):  # one shared job: units are counted once
        el = comm.max_over_ranks(el_rank)
        value = steps / el
    else:
        value, el = replica_throughput(comm, float(steps), el_rank)
    e1 = None
    if profile_in_timed:
        cnt = meas.counters()
    meas.set_profiling(False)
    # norm / energy of the state after exactly the timed sweeps, before the profiled repeat moves it on
    if ss is not None:  # the sharded state's own norm and energy after the run (collective; not timed)
        nrm = ss.norm()
        e1 = ss.expectation().real
    elif mode == "single":  # energy conservation of the run itself, from the blocks the last sweep left around the centre
        e1 = energy_at_centre(eng)
    if not profile_in_timed:
        meas.counters_reset()
        meas.set_profiling(True)
        for i in range(0, steps, unit):
            run_unit()
        sync_norm()
        cnt = meas.counters()
        meas.set_profiling(False)
    halo = ss.traffic() if ss is not None else (0, 0)
    per_rank = None
    if ss is not None:  # every rank's share of the time step: block half-sweeps vs junction updates (incl. waiting for them)
        bm, jm, ns = ss.phase_times()
        mine = [bm / max(ns, 1), jm / max(ns, 1), ss.setup_s]
        if comm.dist is not None:
            import torch

            t = torch.tensor(mine, dtype=torch.float64, device=comm.device)
            allr = [torch.zeros_like(t) for _ in range(world)]
            comm.dist.all_gather(allr, t)
            per_rank = [[float(x) for x in a.tolist()] for a in allr]
        else:
            per_rank = [mine]

    if rank == 0:
        roof, brk, kh, kk = roofline_report(cnt, L, d, D, M, gemm_mode, el, profile_in_timed, tp=(mode == "tp"))
        out = {
            "metric": "tdvp_sweeps_per_sec",
            "value": value,
            "unit": "sweeps/s",
            "n_gpus": args.gpus,
            "steps": steps,
            "warmup": warm_done,
            "steps_requested": args.steps,
            "warmup_requested": args.warmup,
            "ms_per_step": 1e3 * el / steps,
            "higher_is_better": True,
            "scaling": "strong" if mode in ("tp", "sites") else "weak",
            "vs_baseline": None,
            "dtype": "c128",
            "data": "synthetic",
            "config": {
                "workload": f"{args.workload}: {desc}",
                "L": L, "d": d, "D": D, "M": M, "dt_au": dt, "thresh_sil": 1e-9,
                "integrator": integ,
                "step": "one half-sweep (L site exponentials, L-1 QR, L-1 env updates, L-1 bond exponentials)",
                "mean_krylov_site": round(kh, 2),
                "mean_krylov_bond": round(kk, 2),
                "norm_after": nrm,
                "energy_before": e0,
                "energy_after": e1,
                "parallelism": {"single": "single GPU", "replicas": f"{args.gpus} independent replicas",
                                "tp": f"bond-sharded over {args.gpus} GPUs (all-gather / all-reduce via {collectives})",
                                "sites": f"site ranges over {args.gpus} GPUs (L/N sites per rank, one library call per time step: "
                                         "half-sweeps + joint two-site update through pinv(X) "
                                         f"[{(ss.junction + ': both ranks of a junction, bond-sharded') if ss is not None and ss.junction == 'pair' else 'left rank of a junction'}] "
                                         f"+ neighbour halo [{ss.transport if ss is not None else None}]); control plane torch.distributed/{comm.backend}; "
                                         "roofline / breakdown are rank 0's block"}[mode],
                "halo_GB": halo[0] / 1e9,
                "halo_messages": halo[1],
                "per_rank_ms_per_step": ({"block_sweeps": [r[0] for r in per_rank], "junctions": [r[1] for r in per_rank]}
                                         if per_rank else None),
                "setup_s_per_rank": [r[2] for r in per_rank] if per_rank else None,
                "halo_path": (("library RCCL: grouped ncclSend / ncclRecv of device buffers on the engine's stream"
                               if ss.transport == "rccl" else f"callback transport: torch.distributed/{comm.backend}, host-staged")
                              if ss is not None and world > 1 else None),
                # the site-sharded scheme is an approximation (the reference's own; error ~ dt^2 per junction): how far
                # the sharded state drifted over the run
                "approximate": mode == "sites",
                "accuracy": ({"norm_minus_1": nrm - 1.0, "energy_drift_rel": (e1 - e0) / abs(e0) if e0 else None}
                             if ss is not None else None),
                "collectives": int(cnt["n_collectives"]),
                "collective_GB": cnt["collective_bytes"] / 1e9,
                "wall_budget_s": args.max_seconds,
            },
            "roofline": roof,
            "breakdown_ms": brk,
        }
        if with_cpu:
            note("timing the CPU baseline (oracle on the host cores)")
            out["cpu_baseline"] = cpu_baseline_pinned(L, d, D, M, kh, kk, dt)
        # ---- the other BASELINE configs, driver-observed: short single-GPU legs after the headline run ----
        if mode == "single" and args.secondary != "none":
            names = [w for w in ("C2", "C3", "C5") if w != args.workload] if args.secondary == "auto" else \
                    [w for w in args.secondary.split(",") if w in WORKLOADS and w != args.workload]
            eng.close()
            eng = None
            out["secondary"] = {}
            share = {"C2": 0.15, "C3": 0.25, "C5": 0.60}
            tot_share = sum(share.get(w, 0.3) for w in names) or 1.0
            for w in names:
                secs = args.secondary_seconds * share.get(w, 0.3) / tot_share
                note(f"secondary leg {w} ({secs:.0f}s)")
                try:
                    out["secondary"][w] = secondary_leg(w, comm.gpu, gemm_mode, secs, args.steps if w != "C2" else max(args.steps, 100),
                                                        not args.no_cpu_baseline, note)
                except Exception as e:  # noqa: BLE001 -- the headline number stands on its own
                    out["secondary"][w] = {"error": f"{type(e).__name__}: {e}"}
            out["config"]["secondary_summary"] = secondary_summary(out["secondary"])
            out["secondary"] = {w: compact_record(r) for w, r in out["secondary"].items()}
        out["wall_s"] = elapsed()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if eng is not None:
        eng.close()
    if ss is not None:
        ss.close()
    comm.close()


if __name__ == "__main__":
    main()
