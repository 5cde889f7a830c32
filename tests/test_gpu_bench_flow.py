"""bench.py's N > 1 branches, end to end, on ONE GPU.  `bench.py --gpus N` has only ever met 1-GPU boxes, so the Python around a
sharded run -- per-rank tables, the per-rank parity check and its reduction, BASELINE config 5 behind the weak-scaling line, the
bounded final line with its N > 1 members, the exit code -- is run here as what RANK 0 OF A 2-RANK JOB executes: a live 1-rank RCCL
communicator (SKDSP_DIST_FORCE_COMM) serves every collective, bench.py is told the world has two ranks, and rank 0's shard is the
head of the signal (zero history), so every parity figure is the real one.  What it cannot show is the peer hop itself."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r"""
import os, sys, json, io, contextlib, ctypes
ROOT = %r
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'scikit-dsp-comm_amd'))
os.environ.update(SKDSP_DIST_FORCE_COMM='1', WORLD_SIZE='2', RANK='0', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT='29555')
import numpy as np
from sk_dsp_comm_amd import _ffi, sharding
import bench

Base = sharding.RcclTransport

class Rank0Of2(Base):
    def __init__(self, rank, world, local):
        assert (rank, world, local) == (0, 2, 0), (rank, world, local)
        self.rank, self.world, self._rdzv = 0, 2, None
        _ffi.init(0)
        L = _ffi.load()
        buf = ctypes.create_string_buffer(128)
        _ffi.check(L.skdsp_dist_unique_id(buf))
        _ffi.check(L.skdsp_dist_init(0, 1, ctypes.c_char_p(buf.raw)))
    def allgather_state(self, vec):   # one row arrives (the communicator has one rank); the second rank's row is a copy
        self.world = 1
        try:
            tab = Base.allgather_state(self, vec)
        finally:
            self.world = 2
        return np.tile(tab, (2, 1))

sharding.RcclTransport = Rank0Of2
sys.argv = ['bench.py', '--gpus', '2', '--steps', '6', '--warmup', '3'] + sys.argv[1:]
bench.main()
"""


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["fir1024", "iir8"])
def test_bench_main_as_rank_0_of_2(workload):
    if workload == "iir8":   # a reader of both streams as one (2>&1): the line is the LAST thing it sees, whatever librccl wrote through C stdio
        m = subprocess.run([sys.executable, "-c", CODE % ROOT, "--workload", workload], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900, cwd=ROOT)
        assert m.returncode == 0, m.stdout.decode()[-3000:]
        ml = [ln for ln in m.stdout.decode().split("\n") if ln.strip()]
        assert any("Librccl" in ln for ln in ml[:-1]) and json.loads(ml[-1])["n_gpus"] == 2, ml[-3:]
    r = subprocess.run([sys.executable, "-c", CODE % ROOT, "--workload", workload], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().split("\n") if ln.strip()]
    # ONE line on the process's stdout -- whatever the libraries it loaded print there (librccl announces its path through C stdio, flushed at exit: behind the result)
    assert len(lines) == 1, lines
    line = lines[0]
    assert len(line) <= 8000
    assert "Librccl" not in line
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 6 and d["warmup"] == 3
    assert d["config"]["n_ranks_rccl"] == 1            # (the live communicator: one rank here, N in a real run)
    assert d["value"] > 2 * 1e5                        # two ranks' samples over the slowest rank's time
    assert d["parity_ok"] is True and max(d["parity_halo_max_err"], d["parity_interior_max_err"]) < 1e-6
    assert len(d["per_rank"]["step_ms"]) == 2 and d["cpu_baseline"] is None
    assert d["roofline"]["frac"] > (0.4 if workload == "fir1024" else 0.15)   # (the sharded IIR carries states across calls: the cascade kernels, not the parallel form)
    if workload != "fir1024":      # (the state hand-off of the sharded IIR: no halo, no config-5 leg)
        assert "config5" not in d and "state hand-off" in d["config"]["sharding"]
        return
    c5 = d["config5"]
    assert "error" not in c5 and "skipped" not in c5, c5
    assert c5["parity_ok"] is True and c5["n_gpus"] == 2 and c5["total_samples"] == 1 << 30
