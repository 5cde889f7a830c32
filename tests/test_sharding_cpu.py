"""N>1 path on CPU: shard planner, file rendezvous, and a world_size-2 gloo run of the
sharded FIR driver (halo exchange of Ntaps-1 samples) checked against the oracle."""
import multiprocessing as mp
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from sk_dsp_comm_amd import sharding

HERE = os.path.dirname(os.path.abspath(__file__))


def test_shard_bounds_cover_and_align():
    for n in (0, 1, 7, 1000, 2 ** 20 + 3):
        for world in (1, 2, 3, 8):
            for mult in (1, 3, 12):
                b = sharding.shard_bounds(n, world, mult)
                assert len(b) == world and b[0][0] == 0 and b[-1][1] == n
                for (s0, s1), (t0, t1) in zip(b[:-1], b[1:]):
                    assert s1 == t0 and s0 <= s1
                    assert s1 % mult == 0 or s1 == n
    assert sharding.shard_bounds(2 ** 30, 8) == [(i * 2 ** 27, (i + 1) * 2 ** 27) for i in range(8)]
    with pytest.raises(ValueError):
        sharding.shard_bounds(10, 0)


def _rdzv_worker(rank, world, root, q):
    r = sharding.FileRendezvous(rank, world, tag="pytest_%d" % os.getppid(), root=root, timeout=30)
    blob = r.broadcast("id", b"\x01\x02" * 64 if rank == 0 else None)
    r.barrier("up")
    q.put((rank, blob))
    r.cleanup()


def test_file_rendezvous_two_processes(tmp_path):
    ctx = mp.get_context("fork")
    q = ctx.Queue()
    ps = [ctx.Process(target=_rdzv_worker, args=(r, 2, str(tmp_path), q)) for r in range(2)]
    for p in ps:
        p.start()
    got = dict(q.get(timeout=60) for _ in range(2))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert got[0] == got[1] == b"\x01\x02" * 64
    assert not os.path.exists(os.path.join(str(tmp_path), "skdsp_rdzv_pytest_%d" % os.getpid()))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_gloo_world2_sharded_fir(tmp_path):
    pytest.importorskip("torch")
    port = str(_free_port())
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_gloo_worker.py"), str(r), "2", port, str(tmp_path)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    spans = []
    for r in range(2):
        err, ok_hist, tmax, s0, s1 = open(os.path.join(str(tmp_path), "rank%d.txt" % r)).read().split()
        assert float(err) < 1e-12        # sharded == whole-vector oracle (float64 arithmetic)
        assert ok_hist == "True"
        assert float(tmax) == 2.0
        spans.append((int(s0), int(s1)))
    assert spans[0][0] == 0 and spans[0][1] == spans[1][0] and spans[1][1] == 20011
