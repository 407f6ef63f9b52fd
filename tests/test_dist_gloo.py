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
