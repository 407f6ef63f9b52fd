#!/usr/bin/env python3
"""bench.py -- TDVP sweeps/s of the MI355X engine on BASELINE.json's configs.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C4]

One "step" = one half-sweep of one-site TDVP over all L sites of a synthetic
full-rank MPS under a synthetic Hermitian MPO (SURVEY.md 8d inputs), i.e. L
local exp(-i H_eff dt/2), L-1 QR gauge moves, L-1 environment updates, L-1 local
exp(+i K_eff dt/2).  Sweeps alternate direction (forward, backward, ...), two of
them are one PyTDSCF time step.  All tensors are generated on / resident in HBM
before the timed region.

N > 1 (launched by torch.distributed.run, one rank per GPU), --parallel:
  tp        (default when D % N == 0) ONE sweep shared by all GPUs, "scaling":
            "strong": every H_eff / K_eff apply and environment update is sharded
            over the bra-side bond index (each rank contracts D/N rows of the
            environment block), combined by one RCCL all-gather / all-reduce per
            contraction chain over xGMI; Krylov algebra and QR are replicated.
            Exact: same results as one GPU up to summation order.
  replicas  N independent trajectories (SURVEY 8e "fallback"), "scaling": "weak",
            no data-path collective.

Prints ONE JSON line on rank 0.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

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


def cpu_baseline(L, d, D, M, kh, kk, n_threads, budget_s=30.0):
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
    sweep_flops = sum(flops_heff(a, d, b, M, M) for a, b in bd) * (kh + 1.0)
    if sweep_flops > 2e12:  # an end-to-end oracle sweep would take minutes on the host
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
    st.sweep(WORK_DT[0], True)
    st.sweep(WORK_DT[0], False)
    n = 0
    t0 = time.perf_counter()
    while True:
        st.sweep(WORK_DT[0], True)
        st.sweep(WORK_DT[0], False)
        n += 2
        if time.perf_counter() - t0 > min(budget_s, 15.0) or n >= 20:
            break
    el = time.perf_counter() - t0
    return dict(value=n / el, unit="sweeps/s", cores=n_threads, kind="port", sample=f"{n} full sweeps of the oracle, {el:.1f}s")


WORK_DT = [0.0]


def heff_traffic(d, D, M):
    """HBM-side bytes per H_eff apply from the committed PMC passes (separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of tools/heff_probe.py, FETCH
    doubled per MI355X_MICROARCH.md); None for shapes that were not measured."""
    f = os.path.join(ROOT, "profiles", "r01_heff_traffic.json")
    try:
        t = json.load(open(f))
    except OSError:
        return None
    if t["shape"] == {"D": D, "d": d, "M": M}:
        return t["total_bytes"]
    return None


def small_regime_traffic(L, d, D, M):
    """Fabric-side bytes per H_eff apply in the small-bond regime from the committed PMC passes over
    tools/small_trace.py (profiles/r01_c2_traffic.json: FETCH doubled + WRITE of the NN / NT GEMM
    launches, divided by the applies -- an upper bound, those launches also serve the environment
    updates); None for shapes that were not measured."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "r01_c2_traffic.json")))
    except OSError:
        return None
    if (L, d, D, M) == (10, 10, 32, 6):
        return t["zgemm_NN_NT_bytes_per_heff_apply_upper_bound"]
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("MITDVP_WORKLOAD", "C4"), choices=sorted(WORKLOADS))
    ap.add_argument("--dt", type=float, default=None, help="time step in a.u. (default per workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--parallel", default=os.environ.get("MITDVP_PARALLEL", "auto"), choices=["auto", "tp", "replicas"])
    args = ap.parse_args()

    from pytdscf_amd.dist import Comm, replica_throughput

    comm = Comm()  # RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment
    rank, local_rank, world = comm.rank, comm.local_rank, comm.world
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        args.gpus = world

    from pytdscf_amd import synthetic as syn  # synthetic inputs; oracle/ is imported by the cpu_baseline leg only
    from pytdscf_amd import TDVPEngine
    from pytdscf_amd.engine import get_gemm_mode

    gemm_mode = get_gemm_mode()

    L, d, D, M, dt, integ, desc = WORKLOADS[args.workload]
    if args.dt is not None:
        dt = args.dt
    WORK_DT[0] = dt

    liouville = integ == "arnoldi"
    eng = TDVPEngine(L, device=local_rank, integrator=integ, conserve_norm=not liouville)
    eng.set_mpo(syn.synthetic_liouvillian_mpo(L, M, seed=0, gamma=0.002) if liouville else syn.synthetic_mpo(L, d, M, seed=0))
    mode = args.parallel
    if mode == "auto":
        mode = "tp" if (world > 1 and D % world == 0 and D >= 8 * world) else "replicas"
    if world == 1:
        mode = "single"
    # tp: every rank holds the same replicated state (same seed); replicas: one trajectory per rank
    eng.init_random([d] * L, D, seed=1 + (rank if mode == "replicas" else 0))
    if mode == "tp":
        from pytdscf_amd.dist import attach_parallel

        try:
            if os.environ.get("MITDVP_COLLECTIVES", "torch") == "native":
                from pytdscf_amd.dist import attach_parallel_native

                attach_parallel_native(eng, comm)  # the library's own RCCL calls on the engine's stream
            else:
                attach_parallel(eng, comm)  # includes a collective self-test with a verdict common to all ranks
        except RuntimeError as e:
            # every rank gets here together: run N independent trajectories instead
            if rank == 0:
                print(f"[bench] {e}; falling back to independent replicas", file=sys.stderr, flush=True)
            mode = "replicas"
            eng.init_random([d] * L, D, seed=1 + rank)
    e0 = eng.expectation().real  # also builds nothing persistent; forces setup to finish

    def note(msg):
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    note(f"{args.workload} set up on device (L={L} d={d} D={D} M={M}), <H>={e0:.9f}")

    barrier = comm.barrier

    forward = True
    for i in range(args.warmup):
        eng.sweep(dt, forward)
        eng.norm()
        note(f"warm-up sweep {i + 1}/{args.warmup} done")
        forward = not forward
    if args.warmup == 0:
        # the right environments are state, not sweep work: build them before timing
        eng.sweep(0.0, True)
        eng.sweep(0.0, False)
    eng.norm()
    eng.counters_reset()
    # Per-phase HIP-event timing (roofline / breakdown) costs two event records per phase:
    # nothing at C4 (0.01 %), a third of the run in the launch-bound small-bond regime.  There
    # the timed region runs unprofiled and the same number of sweeps is repeated afterwards,
    # profiled, only for the breakdown.
    profile_in_timed = D >= 128
    eng.set_profiling(profile_in_timed)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        eng.sweep(dt, forward)
        forward = not forward
        if args.steps > 1 and L * D >= 32768:
            eng.norm()
            note(f"timed sweep {i + 1}/{args.steps} done")
    nrm = eng.norm()  # synchronises the engine's stream
    barrier()
    el_rank = time.perf_counter() - t0
    if mode == "tp":  # one shared job: units are counted once
        el = comm.max_over_ranks(el_rank)
        value = args.steps / el
    else:
        value, el = replica_throughput(comm, float(args.steps), el_rank)
    if not profile_in_timed:
        eng.counters_reset()
        eng.set_profiling(True)
        for i in range(args.steps):
            eng.sweep(dt, forward)
            forward = not forward
        eng.norm()
    cnt = eng.counters()
    eng.set_profiling(False)

    if rank == 0:
        kh = cnt["n_heff"] / max(cnt["n_exp_site"], 1)
        kk = cnt["n_keff"] / max(cnt["n_exp_bond"], 1)
        ach = cnt["heff_flops"] / max(cnt["heff_ms"], 1e-9) / 1e9  # TFLOP/s
        # small-bond regime (C2, D < 128): the apply is memory / latency bound, the
        # roofline that applies is HBM: algorithmic bytes B_H per apply over the apply time
        small = D < 128
        bytes_apply = 16.0 * (2 * D * d * D + 2 * D * D * M + M * d * d * M)  # interior site, SURVEY 8d B_H
        ach_gbs = bytes_apply * cnt["n_heff"] / max(cnt["heff_ms"], 1e-9) / 1e6  # GB/s
        out = {
            "metric": "tdvp_sweeps_per_sec",
            "value": value,
            "unit": "sweeps/s",
            "n_gpus": args.gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * el / args.steps,
            "higher_is_better": True,
            "scaling": "strong" if mode == "tp" else "weak",
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
                "parallelism": {"single": "single GPU", "replicas": f"{args.gpus} independent replicas",
                                "tp": f"bond-sharded over {args.gpus} GPUs (RCCL all-gather / all-reduce)"}[mode],
                "collectives": int(cnt["n_collectives"]),
                "collective_GB": cnt["collective_bytes"] / 1e9,
            },
            "roofline": ({
                "bound": "hbm",
                "kernel": "H_eff apply (3 zgemm launches) in the small-bond regime",
                "achieved": ach_gbs,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": ach_gbs / HBM_PEAK_GBS,
                "traffic": small_regime_traffic(L, d, D, M),
                "bytes_per_apply": bytes_apply,
                "ms_per_apply": cnt["heff_ms"] / max(cnt["n_heff"], 1),
                "n_apply": cnt["n_heff"],
                "tflops": ach,
                "note": "launch/latency bound: a site is a few hundred KB, every apply is three dependent launches of a few microseconds",
            } if small else {
                "bound": "mfma",
                "kernel": "zgemm_kernel (H_eff apply = 3 launches: L.psi, W., .R)" + (" -- per GPU, this rank's bond shard" if mode == "tp" else ""),
                "achieved": ach,
                "peak": FP64_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": ach / FP64_MFMA_PEAK_TFLOPS,
                "traffic": heff_traffic(d, D, M),
                "flops_per_apply": cnt["heff_flops"] / max(cnt["n_heff"], 1),
                "ms_per_apply": cnt["heff_ms"] / max(cnt["n_heff"], 1),
                "stage_ms_per_apply": [x / max(cnt["n_heff"], 1) for x in cnt["heff_stage_ms"]],
                "n_apply": cnt["n_heff"],
                "complex_product": gemm_mode,
                "note": ("3M (Karatsuba) complex product: 6 real flop executed per 8 algorithmic; executed-MFMA "
                         "rate = 0.75 x achieved") if gemm_mode == "3m" else "4M complex product: executed = algorithmic flops",
            }),
            "breakdown_ms": {
                "heff": cnt["heff_ms"], "env": cnt["env_ms"], "keff": cnt["keff_ms"], "qr": cnt["qr_ms"],
                "krylov_vec": cnt["krylov_vec_ms"], "wall": 1e3 * el,
                "phases_from": "the timed sweeps" if profile_in_timed else "a profiled repeat of the timed sweeps (event overhead kept out of the timed region)",
                "env_tflops": cnt["env_flops"] / max(cnt["env_ms"], 1e-9) / 1e9,
                "keff_tflops": cnt["keff_flops"] / max(cnt["keff_ms"], 1e-9) / 1e9,
                "qr_tflops": cnt["qr_flops"] / max(cnt["qr_ms"], 1e-9) / 1e9,
                "launches": cnt["n_launch"],
            },
        }
        if not args.no_cpu_baseline and args.gpus == 1:
            nthr = os.cpu_count() or 1
            try:  # the threads the BLAS behind NumPy actually runs (OpenBLAS caps at its build-time maximum)
                import numpy  # noqa: F401
                import threadpoolctl

                blas = [x["num_threads"] for x in threadpoolctl.threadpool_info() if x.get("user_api") == "blas"]
                if blas:
                    nthr = max(blas)
            except Exception:
                pass
            out["cpu_baseline"] = cpu_baseline(L, d, D, M, kh, kk, nthr)
            out["cpu_baseline"]["host_cpus"] = os.cpu_count()
        print(json.dumps(out), flush=True)
    eng.close()
    comm.close()


if __name__ == "__main__":
    main()
