"""Start several rank processes and collect them without leaving any behind.

stdout+stderr of every rank go to a temporary file (no pipe to fill up while we poll); the
first rank that exits non-zero, or the deadline, kills the others -- peers of a dead rank
would otherwise sit in a collective holding the GPU."""

from __future__ import annotations

import subprocess
import tempfile
import time


def run_ranks(cmds, envs, timeout=300, cwd=None, split_stderr=False):
    """cmds[i] / envs[i]: argv and environment of rank i.  Returns (returncodes, outputs);
    outputs[i] = combined text, or (stdout, stderr) with split_stderr.  A rank still alive at
    the deadline (or after a peer failed) is killed and reported with returncode -9."""
    files, procs = [], []
    try:
        for cmd, env in zip(cmds, envs):
            fo = tempfile.TemporaryFile(mode="w+")
            fe = tempfile.TemporaryFile(mode="w+") if split_stderr else None
            files.append((fo, fe))
            procs.append(subprocess.Popen(cmd, env=env, cwd=cwd, stdout=fo, stderr=fe if split_stderr else subprocess.STDOUT,
                                          text=True))
        deadline = time.monotonic() + timeout
        while any(p.poll() is None for p in procs):
            if time.monotonic() > deadline or any(p.poll() not in (None, 0) for p in procs):
                break
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
    outs = []
    for fo, fe in files:
        fo.seek(0)
        if fe is not None:
            fe.seek(0)
            outs.append((fo.read(), fe.read()))
            fe.close()
        else:
            outs.append(fo.read())
        fo.close()
    return [p.returncode for p in procs], outs
