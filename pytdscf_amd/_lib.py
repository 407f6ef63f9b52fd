"""ctypes binding of ``libmitdvp.so`` (C ABI declared in ``include/mitdvp.h``).

The product path has NO CPU fallback: if the HIP library is missing or no GPU
is visible every compute entry point raises.  Loading the library itself does
not touch the GPU, so symbol checks can run on a CPU-only box.
"""

from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MITDVP_LIB") or os.path.join(_HERE, "csrc", "libmitdvp.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "mitdvp.h")

OK, EINVAL, EHIP, ENOTCONV, ESTATE, ENOMEM = 0, -1, -2, -3, -4, -5
LANCZOS, ARNOLDI = 0, 1
GAUGE_PSI, GAUGE_A, GAUGE_B, GAUGE_C = 0, 1, 2, -1


class MitdvpError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [
        ("nsite", C.c_int),
        ("device", C.c_int),
        ("integrator", C.c_int),
        ("conserve_norm", C.c_int),
        ("relax", C.c_int),
        ("thresh", C.c_double),
        ("max_krylov", C.c_int),
        ("lanczos_variant", C.c_int),
        ("max_diag_krylov", C.c_int),
        ("cu_first", C.c_int),
        ("cu_count", C.c_int),
        ("reserved", C.c_int * 5),
    ]


class Counters(C.Structure):
    _fields_ = [
        ("heff_flops", C.c_double),
        ("heff_ms", C.c_double),
        ("env_flops", C.c_double),
        ("env_ms", C.c_double),
        ("keff_flops", C.c_double),
        ("keff_ms", C.c_double),
        ("qr_flops", C.c_double),
        ("qr_ms", C.c_double),
        ("krylov_vec_ms", C.c_double),
        ("n_heff", C.c_longlong),
        ("n_keff", C.c_longlong),
        ("n_env", C.c_longlong),
        ("n_qr", C.c_longlong),
        ("n_exp_site", C.c_longlong),
        ("n_exp_bond", C.c_longlong),
        ("n_launch", C.c_longlong),
        ("heff_stage_ms", C.c_double * 3),
        ("n_collectives", C.c_double),
        ("collective_bytes", C.c_double),
        ("heff_flops_skipped", C.c_double),
        ("n_host_waits", C.c_double),
        ("n_heff_edge", C.c_double),
        ("heff_stage_flops", C.c_double * 3),
    ]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k not in ("reserved", "heff_stage_ms", "heff_stage_flops")}
        d["heff_stage_ms"] = list(self.heff_stage_ms)
        d["heff_stage_flops"] = list(self.heff_stage_flops)
        return d


COLLECTIVE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t)
P2P_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t)  # mitdvp_p2p_fn

_lib = None


def declared_symbols() -> list[str]:
    """Every function name declared in include/mitdvp.h."""
    with open(HEADER_PATH) as f:
        txt = f.read()
    return sorted(set(re.findall(r"\b(mitdvp_[a-z_0-9]+)\s*\(", txt)))


def load() -> C.CDLL:
    """Load the library (no GPU needed for loading). Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MitdvpError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C pytdscf_amd/csrc`.  There is no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p
    i, d = C.c_int, C.c_double
    sig = {
        "mitdvp_create": (i, [C.POINTER(Config), C.POINTER(vp)]),
        "mitdvp_destroy": (None, [vp]),
        "mitdvp_last_error": (C.c_char_p, [vp]),
        "mitdvp_version": (C.c_char_p, []),
        "mitdvp_device_count": (i, [ip]),
        "mitdvp_device_sync": (i, [i]),
        "mitdvp_set_site": (i, [vp, i, dp, i, i, i, i]),
        "mitdvp_get_site_shape": (i, [vp, i, ip, ip, ip, ip]),
        "mitdvp_get_site": (i, [vp, i, dp]),
        "mitdvp_init_random": (i, [vp, ip, i, C.c_uint64]),
        "mitdvp_init_random_block": (i, [vp, ip, i, i, i, C.c_uint64, i]),
        "mitdvp_canonicalize": (i, [vp, d]),
        "mitdvp_set_mpo_core": (i, [vp, i, i, dp, i, i, i, i]),
        "mitdvp_set_shift": (i, [vp, i, d, d]),
        "mitdvp_step": (i, [vp, d]),
        "mitdvp_sweep": (i, [vp, d, i]),
        "mitdvp_invalidate_env": (i, [vp]),
        "mitdvp_set_boundary_env": (i, [vp, i, dp, i, i]),
        "mitdvp_replace_site": (i, [vp, i, dp, i]),
        "mitdvp_get_env": (i, [vp, i, i, dp, ip, ip]),
        "mitdvp_build_envs": (i, [vp, i]),
        "mitdvp_site_exp": (i, [vp, d]),
        "mitdvp_split_center": (i, [vp, i]),
        "mitdvp_bond_exp": (i, [vp, d]),
        "mitdvp_absorb_bond": (i, [vp, i]),
        "mitdvp_get_bond": (i, [vp, dp, ip]),
        "mitdvp_set_bond": (i, [vp, i, dp, i]),
        "mitdvp_fold_block": (i, [vp, i, i, i, dp, i, i, dp]),
        "mitdvp_set_pointer_mode": (i, [vp, i]),
        "mitdvp_fold_block_range": (i, [vp, i, i, i, i, i, dp, i, i, dp]),
        "mitdvp_site_rdm_blocks": (i, [vp, i, dp, dp, dp]),
        "mitdvp_expect": (i, [vp, i, dp]),
        "mitdvp_autocorr": (i, [vp, dp]),
        "mitdvp_norm": (i, [vp, dp]),
        "mitdvp_site_rdm": (i, [vp, i, dp]),
        "mitdvp_reduced_density": (i, [vp, ip, i, dp, C.POINTER(C.c_size_t)]),
        "mitdvp_truncate_bond": (i, [vp, d, i, ip, dp]),
        "mitdvp_svd": (i, [i, dp, i, i, dp, dp, dp, ip]),
        "mitdvp_set_adaptive": (i, [vp, i, i, i, d]),
        "mitdvp_set_gate": (i, [vp, i, dp, i]),
        "mitdvp_save_reference": (i, [vp]),
        "mitdvp_overlap_reference": (i, [vp, dp]),
        "mitdvp_rccl_unique_id": (i, [C.c_char_p]),
        "mitdvp_set_parallel_rccl": (i, [vp, i, i, C.c_char_p]),
        "mitdvp_rccl_selftest": (i, [vp, ip]),
        "mitdvp_operate": (i, [vp, i, i, d, dp, ip]),
        "mitdvp_ms_configure": (i, [vp, i]),
        "mitdvp_ms_set_site": (i, [vp, i, i, dp, i, i, i, i]),
        "mitdvp_ms_get_site_shape": (i, [vp, i, i, ip, ip, ip, ip]),
        "mitdvp_ms_get_site": (i, [vp, i, i, dp]),
        "mitdvp_ms_canonicalize": (i, [vp, i, d]),
        "mitdvp_ms_set_mpo_core": (i, [vp, i, i, i, i, dp, i, i, i, i]),
        "mitdvp_ms_set_coupleJ": (i, [vp, i, i, i, d, d]),
        "mitdvp_ms_step": (i, [vp, d]),
        "mitdvp_ms_expect": (i, [vp, i, dp]),
        "mitdvp_ms_autocorr": (i, [vp, dp]),
        "mitdvp_ms_pops": (i, [vp, dp]),
        "mitdvp_ms_operate": (i, [vp, i, i, d, dp, ip]),
        "mitdvp_set_kraus": (i, [vp, i, i, dp, i, i]),
        "mitdvp_apply_kraus": (i, [vp]),
        "mitdvp_clock_probe": (i, [i, C.c_long, dp]),
        "mitdvp_apply_gates": (i, [vp]),
        "mitdvp_thin_to_full": (i, [i, i, dp, i, i, i, i, dp]),
        "mitdvp_set_trace_op_core": (i, [vp, i, i, dp, i, i, i]),
        "mitdvp_expect_trace": (i, [vp, i, dp]),
        "mitdvp_partial_trace": (i, [vp, ip, i, dp, C.POINTER(C.c_size_t)]),
        "mitdvp_set_subspace": (i, [vp, i, i, ip, i]),
        "mitdvp_hermitise": (i, [vp]),
        "mitdvp_krylov_stats": (i, [vp, ip]),
        "mitdvp_counters_get": (i, [vp, C.POINTER(Counters)]),
        "mitdvp_counters_reset": (i, [vp]),
        "mitdvp_set_profiling": (i, [vp, i]),
        "mitdvp_set_parallel": (i, [vp, i, i, COLLECTIVE_FN, vp]),
        "mitdvp_heff_apply": (i, [i, dp, dp, dp, dp, i, i, i, i, i, dp, i, dp]),
        "mitdvp_keff_apply": (i, [i, dp, dp, dp, i, i, i, dp]),
        "mitdvp_env_update": (i, [i, i, dp, dp, dp, i, i, i, i, i, dp]),
        "mitdvp_gauge_trf": (i, [i, i, dp, i, i, i, dp, dp]),
        "mitdvp_expm_dense": (i, [i, i, i, i, dp, i, dp, d, d, d, i, dp, ip]),
        "mitdvp_zgemm": (i, [i, i, i, i, i, i, i, i, dp, dp, dp, dp, dp, i, i, dp]),
        "mitdvp_bench_heff": (i, [i, i, i, i, i, i, i, i, dp]),
        "mitdvp_heff_selfcheck": (i, [i, i, i, i, i, i, dp]),
        "mitdvp_set_gemm_mode": (i, [i]),
        "mitdvp_get_gemm_mode": (i, []),
        "mitdvp_mfma_peak_probe": (i, [i, dp]),
        "mitdvp_mfma_layout_probe": (i, [i, ip]),
        "mitdvp_shard_create": (i, [C.POINTER(Config), i, i, i, i, C.POINTER(vp)]),
        "mitdvp_shard_destroy": (None, [vp]),
        "mitdvp_shard_last_error": (C.c_char_p, [vp]),
        "mitdvp_shard_engine": (i, [vp, i, C.POINTER(vp)]),
        "mitdvp_shard_set_options": (i, [vp, i, d]),
        "mitdvp_shard_enable_pair": (i, [vp, i]),
        "mitdvp_shard_set_joint": (i, [vp, dp, i]),
        "mitdvp_shard_get_joint": (i, [vp, dp, ip]),
        "mitdvp_shard_set_transport": (i, [vp, P2P_FN, vp]),
        "mitdvp_shard_attach_rccl": (i, [vp, C.c_char_p]),
        "mitdvp_shard_selftest": (i, [vp, ip]),
        "mitdvp_shard_self_sendrecv": (i, [vp, C.c_size_t, ip]),
        "mitdvp_shard_step": (i, [vp, d]),
        "mitdvp_shard_sweep": (i, [vp, d, i, i]),
        "mitdvp_shard_junctions": (i, [vp, d, i]),
        "mitdvp_shard_traffic": (i, [vp, dp, C.POINTER(C.c_long)]),
        "mitdvp_shard_phase_times": (i, [vp, dp, dp, C.POINTER(C.c_long)]),
        "mitdvp_heff_apply_center": (i, [vp, dp, dp, ip]),
        "mitdvp_set_qr_fast": (i, [i]),
        "mitdvp_get_qr_fast": (i, []),
        "mitdvp_bench_qr": (i, [i, i, i, i, dp, C.POINTER(C.c_long)]),
        "mitdvp_qr_thin": (i, [i, dp, i, i, i, dp, dp, i, dp, C.POINTER(C.c_long), ip]),
        "mitdvp_get_krylov_memory": (i, [vp, i, ip]),
        "mitdvp_set_krylov_memory": (i, [vp, i, i]),
        "mitdvp_set_small_kernels": (i, [vp, i]),
        "mitdvp_ensemble_step": (i, [C.POINTER(vp), i, d, i, ip]),
        "mitdvp_device_cu_count": (i, [i, ip]),
        "mitdvp_cu_mask_probe": (i, [i, C.POINTER(C.c_uint), i, i, C.c_size_t, i, ip]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, handle=None, shard: bool = False):
    if rc == OK:
        return
    lib = load()
    msg = (lib.mitdvp_shard_last_error if shard else lib.mitdvp_last_error)(handle).decode(errors="replace")
    if rc == ENOTCONV:
        # same exception type as the reference (_integrator.py:430, :653)
        raise ValueError(msg)
    if rc == EINVAL:
        raise ValueError(msg)
    raise MitdvpError(f"mitdvp error {rc}: {msg}")
