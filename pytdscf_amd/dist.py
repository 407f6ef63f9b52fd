"""Multi-GPU plumbing: one process per GPU under ``torch.distributed``
(backend "nccl" = RCCL on ROCm; "gloo" on CPU-only hosts for tests).

This module is the whole N>1 surface: rendezvous, barrier, max-over-ranks of
the elapsed time, the replica throughput aggregation used by bench.py, and
``attach_parallel``: the collectives of the bond-sharded (tensor-parallel)
execution of one sweep over several GPUs."""

from __future__ import annotations

import os


class Comm:
    """Rendezvous from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*.  At world size 1 nothing but the
    HIP library is touched (no torch import); with several ranks ``torch.distributed`` carries the
    barrier and the scalar reductions.  ``device`` is the HIP ordinal the engine runs on:
    LOCAL_RANK modulo the number of visible devices, so several ranks can share one GPU in tests."""

    def __init__(self, n_devices: int | None = None, backend: str | None = None):
        """``backend``: "nccl" / "gloo" for the process group this Comm creates (default: MITDVP_DIST_BACKEND, else nccl
        with one GPU per rank and gloo otherwise); ignored when the process group exists already."""
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        self.device = None
        self.backend = None
        self._owns_group = True
        torch = None
        if self.world > 1:
            # dmabuf IPC is the only form this driver stack supports; without it RCCL's cross-process
            # buffer sharing fails (hipIpcGetMemHandle: invalid argument).  Normally exported already.
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            # torch brings its own HIP runtime: it must initialise BEFORE libmitdvp.so pulls in the
            # system one (the other order leaves torch with "No HIP GPUs are available")
            import torch

            if n_devices is None:
                n_devices = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if n_devices is None:
            try:
                from .engine import device_count

                n_devices = device_count()
            except Exception:  # library not built: CPU-only plumbing tests
                n_devices = 0
        self.n_devices = n_devices
        self.gpu = self.local_rank % n_devices if n_devices > 0 else None
        self.shared_gpu = self.world > 1 and n_devices > 0 and self.world > n_devices
        if self.world > 1:
            import torch.distributed as dist

            # RCCL refuses two ranks on one device: ranks sharing a GPU (rehearsals, tests) use gloo
            backend = backend or os.environ.get("MITDVP_DIST_BACKEND") or ("nccl" if n_devices > 0 and not self.shared_gpu else "gloo")
            if n_devices > 0:
                torch.cuda.set_device(self.gpu)
            if dist.is_initialized():
                # the user's script (or an earlier Comm) has set the process group up: reuse it, never a second
                # init_process_group; its backend decides where the scalars of the control plane live
                backend = str(dist.get_backend()).lower()
                self.device = torch.device("cuda", self.gpu) if "nccl" in backend and n_devices > 0 else torch.device("cpu")
                self._owns_group = False
            elif backend == "nccl":
                self.device = torch.device("cuda", self.gpu)
                dist.init_process_group("nccl", device_id=self.device)
            else:  # gloo: CPU-only hosts, or several test ranks sharing one GPU
                self.device = torch.device("cpu")
                dist.init_process_group("gloo")
            self.dist = dist
            self.backend = backend

    def device_sync(self):
        if self.gpu is not None:
            from .engine import device_sync

            device_sync(self.gpu)

    def barrier(self):
        """device synchronise, barrier over the ranks, device synchronise."""
        self.device_sync()
        if self.dist is not None:
            self.dist.barrier()
            self.device_sync()

    def max_over_ranks(self, x: float) -> float:
        if self.dist is None:
            return float(x)
        import torch

        t = torch.tensor([float(x)], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def min_over_ranks(self, x: float) -> float:
        if self.dist is None:
            return float(x)
        import torch

        t = torch.tensor([float(x)], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return float(t.item())

    def sum_over_ranks(self, x: float) -> float:
        if self.dist is None:
            return float(x)
        import torch

        t = torch.tensor([float(x)], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            if self._owns_group:
                self.dist.destroy_process_group()
            self.dist = None


_WORLD = None


def world_comm(site_sharding: bool = False) -> Comm:
    """The process-wide rendezvous (torch.distributed can be initialised once): what the shell API uses.

    ``site_sharding``: the caller moves its halo with the library's own RCCL communicator
    (``mitdvp_shard_attach_rccl``); torch.distributed is then the control plane only and runs over gloo unless the
    user chose otherwise, so that the process holds ONE RCCL instance on its GPU -- the configuration bench.py and
    every test of the sharded sweep run."""
    global _WORLD
    if _WORLD is None or (_WORLD.world > 1 and _WORLD.dist is None):
        # the preference is handed to this Comm only (the environment is left alone: a later, non-sharded Comm of the
        # same process still gets its own default)
        want = (os.environ.get("MITDVP_DIST_BACKEND") or "gloo") if site_sharding else None
        _WORLD = Comm(backend=want)
    elif site_sharding and _WORLD.world > 1 and _WORLD.backend and "nccl" in _WORLD.backend:
        import warnings

        warnings.warn("pytdscf_amd: the existing torch.distributed process group runs over nccl (RCCL); the site-sharded "
                      "sweep brings its own RCCL communicator for the halo, so this process will hold two RCCL instances "
                      "on its GPU.  Initialise the process group with gloo (or set MITDVP_DIST_BACKEND=gloo) before the "
                      "first use to avoid that.", RuntimeWarning, stacklevel=2)
    return _WORLD


def replica_throughput(comm: Comm, units_this_rank: float, elapsed_this_rank: float) -> tuple[float, float]:
    """Whole-job throughput of independent replicas: (sum of units over ranks) /
    (max elapsed over ranks).  Returns (throughput, max_elapsed)."""
    tmax = comm.max_over_ranks(elapsed_this_rank)
    units = comm.sum_over_ranks(units_this_rank)
    return units / tmax, tmax


# ---------------------------------------------------------------------------
# bond-sharded (tensor-parallel) execution: the collectives the engine asks for
# ---------------------------------------------------------------------------
class _DevPtr:
    """Zero-copy view of engine-owned device memory for torch (CUDA array interface)."""

    def __init__(self, ptr: int, n_f64: int):
        self.__cuda_array_interface__ = {
            "shape": (n_f64,),
            "typestr": "<f8",
            "data": (int(ptr), False),
            "version": 2,
        }


def attach_parallel_native(engine, comm: Comm):
    """Bond-sharded mode with the library's own RCCL collectives on the engine's stream
    (``mitdvp_set_parallel_rccl``): torch.distributed only carries the 128-byte ncclUniqueId
    from rank 0 to the others and the verdict of the self-test."""
    import ctypes as C

    from . import _lib

    lib = _lib.load()
    if comm.world == 1:
        return
    ident = C.create_string_buffer(128)
    if comm.rank == 0:
        _lib.check(lib.mitdvp_rccl_unique_id(ident))
    box = [ident.raw]
    comm.dist.broadcast_object_list(box, src=0)
    _lib.check(lib.mitdvp_set_parallel_rccl(engine._h, comm.world, comm.rank, box[0]), engine._h)
    bad = C.c_int(-1)
    _lib.check(lib.mitdvp_rccl_selftest(engine._h, C.byref(bad)), engine._h)
    if comm.min_over_ranks(1.0 if bad.value == 0 else 0.0) < 1.0:
        raise RuntimeError("bond-sharded mode (native RCCL): the collective self-test failed on at least one rank")


def attach_parallel(engine, comm: Comm, host_staged: bool | None = None):
    """Put ``engine`` (a TDVPEngine living on this rank's GPU) into bond-sharded
    mode over ``comm``.  All ranks must hold the same replicated state and issue
    the same calls.  Collectives run on the engine's device buffers:

    * backend nccl (= RCCL over xGMI): ``all_gather_into_tensor`` / ``all_reduce``
      directly on the device memory;
    * backend gloo (tests, several ranks sharing one GPU): staged through the host.
    """
    import ctypes as C

    import torch

    from . import _lib

    if comm.world == 1:
        return None
    dist = comm.dist
    if host_staged is None:
        host_staged = dist.get_backend() != "nccl"
    rank, world = comm.rank, comm.world
    dev = torch.device("cuda", engine.device)

    def cb(user, op, ptr, nbytes):
        try:
            n = nbytes // 8
            t = torch.as_tensor(_DevPtr(ptr, n), device=dev)
            if op == 0:  # in-place all-gather of equal shards
                chunk = n // world
                mine = t[rank * chunk : (rank + 1) * chunk]
                if host_staged:
                    parts = [torch.empty(chunk, dtype=torch.float64) for _ in range(world)]
                    dist.all_gather(parts, mine.cpu())
                    t.copy_(torch.cat(parts))
                else:
                    dist.all_gather_into_tensor(t, mine.clone())
            elif op == 1:  # in-place all-reduce (sum)
                if host_staged:
                    h = t.cpu()
                    dist.all_reduce(h)
                    t.copy_(h)
                else:
                    dist.all_reduce(t)
            else:
                return 2
            torch.cuda.synchronize(dev)
            return 0
        except Exception as e:  # never let an exception unwind through the C frame
            import sys
            import traceback

            traceback.print_exc(file=sys.stderr)
            return 1

    # self-test of both collectives on a small device buffer before the engine relies on them;
    # every rank learns the common verdict, so all take the same decision afterwards
    probe = torch.full((world * 8,), -1.0, dtype=torch.float64, device=dev)
    probe[rank * 8 : (rank + 1) * 8] = float(rank + 1)
    ok = cb(None, 0, probe.data_ptr(), probe.numel() * 8) == 0
    want = torch.arange(1, world + 1, dtype=torch.float64).repeat_interleave(8)
    ok = ok and bool(torch.equal(probe.cpu(), want))
    probe.fill_(float(rank + 1))
    ok = ok and cb(None, 1, probe.data_ptr(), probe.numel() * 8) == 0
    ok = ok and bool(torch.all(probe.cpu() == world * (world + 1) / 2))
    if comm.min_over_ranks(1.0 if ok else 0.0) < 1.0:
        raise RuntimeError("bond-sharded mode: the collective self-test failed on at least one rank")

    fn = _lib.COLLECTIVE_FN(cb)
    engine._collective_cb = fn  # keep the trampoline alive as long as the engine
    _lib.check(_lib.load().mitdvp_set_parallel(engine._h, world, rank, fn, None), engine._h)
    return fn
