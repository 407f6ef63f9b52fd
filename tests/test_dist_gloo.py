"""CPU: the N>1 path (rendezvous, barrier, max-over-ranks timing, replica
aggregation) with world_size = 2 over gloo on 127.0.0.1."""

import os
import socket
import subprocess
import sys
import textwrap

from helpers.ranks import run_ranks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_replica_aggregation(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(
        textwrap.dedent(
            f"""
            import sys, time
            sys.path.insert(0, {ROOT!r})
            from pytdscf_amd.dist import Comm, replica_throughput
            c = Comm()
            assert c.world == 2
            c.barrier()
            el = 1.0 + 0.5 * c.rank          # rank 1 is the slow replica
            thr, tmax = replica_throughput(c, 3.0, el)
            assert abs(tmax - 1.5) < 1e-12 and abs(thr - 6.0 / 1.5) < 1e-12, (thr, tmax)
            assert c.max_over_ranks(c.rank) == 1.0 and c.sum_over_ranks(1.0) == 2.0
            c.barrier()
            if c.rank == 0:
                print("OK", thr)
            c.close()
            """
        )
    )
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="2", CUDA_VISIBLE_DEVICES="")
    rcs, outs = run_ranks([[sys.executable, str(script)]] * 2, [dict(env, RANK=str(r), LOCAL_RANK=str(r)) for r in range(2)],
                          timeout=120)
    assert rcs == [0, 0], outs
    assert "OK 4.0" in outs[0]


def test_two_rank_shape_exchange_of_the_adaptive_sharded_state(tmp_path):
    """Adaptive ranks across junctions: after every step the ranks share the new site shapes of the whole chain
    (SiteShardedTDVP._refresh_shapes / bond_dims; gathers and folds post their receives with them).  Host logic only:
    the block engine is a stand-in that reports shapes."""
    script = tmp_path / "w.py"
    script.write_text(
        textwrap.dedent(
            f"""
            import sys
            sys.path.insert(0, {ROOT!r})
            from pytdscf_amd.dist import Comm
            from pytdscf_amd.parallel_sites import SiteShardedTDVP
            c = Comm(n_devices=0)
            class Block:
                def __init__(self, shapes): self.shapes = shapes
                def get_site_shape(self, i): return self.shapes[i] + (0,)
            sh = object.__new__(SiteShardedTDVP)
            sh.comm, sh.rank, sh.world = c, c.rank, c.world
            mine = [[(1, 4, 3), (3, 4, 7), (7, 4, 5)], [(5, 4, 6), (6, 4, 2), (2, 4, 1)]][c.rank]
            sh.block, sh.n = Block(mine), 3
            sh._refresh_shapes()
            assert sh.shapes == [(1, 4, 3), (3, 4, 7), (7, 4, 5), (5, 4, 6), (6, 4, 2), (2, 4, 1)], sh.shapes
            assert sh.bond_dims() == [3, 7, 5, 6, 2]
            c.barrier()
            if c.rank == 0:
                print("SHAPES OK")
            c.close()
            """
        )
    )
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="2", CUDA_VISIBLE_DEVICES="",
               MITDVP_DIST_BACKEND="gloo")
    rcs, outs = run_ranks([[sys.executable, str(script)]] * 2, [dict(env, RANK=str(r), LOCAL_RANK=str(r)) for r in range(2)],
                          timeout=120)
    assert rcs == [0, 0], outs
    assert "SHAPES OK" in outs[0]
