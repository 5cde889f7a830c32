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
        err, ok_hist, tmax, s0, s1, iir_err = open(os.path.join(str(tmp_path), "rank%d.txt" % r)).read().split()
        assert float(err) < 1e-12        # sharded == whole-vector oracle (float64 arithmetic)
        assert float(iir_err) < 1e-12    # sharded IIR (state hand-off over gloo) == whole-vector sosfilt
        assert ok_hist == "True"
        assert float(tmax) == 2.0
        spans.append((int(s0), int(s1)))
    assert spans[0][0] == 0 and spans[0][1] == spans[1][0] and spans[1][1] == 20011
    # bench.py's self-verification of an N > 1 run (shard_parity / reduce_parity), driven through the same gloo group:
    # a correct halo passes on every rank, a result computed without the halo is caught on every rank (max-reduced),
    # the per-rank fallback flags arrive in rank order
    import json
    for r in range(2):
        par = json.load(open(os.path.join(str(tmp_path), "parity%d.json" % r)))
        hmax, imax, flags, ok, states = par["good"]
        assert ok and hmax < 1e-12 and imax < 1e-12 and flags == [0, 0]
        hmax, imax, flags, ok, states = par["bad"]
        assert not ok and hmax > 1e-3 and imax < 1e-12 and flags == [0, 1]
        hmax, imax, flags, ok, states = par["good_iir"]
        assert ok and hmax < 1e-12 and imax < 1e-12
        # the config-5 record of an N > 1 run: whole-job rate over the slowest rank's step, per-GPU roofline fraction, speed-up
        # over the committed N = 1 line (withheld when that line is stale), the reduced parity verdict
        c5 = par["c5"]
        assert c5["n_gpus"] == 2 and c5["total_samples"] == 20000 and c5["samples_per_gpu"] == 10000 and c5["ms"] == 2.0
        assert abs(c5["value"] - 20000 / 2e-3 / 1e6) < 1e-9 and abs(c5["speedup_vs_n1"] - 1.5) < 1e-12 and c5["parity_ok"]
        assert abs(c5["frac_per_gpu"] - 16.0 * 10000 / 2e-3 / 1e9 / 8000.0) < 1e-12
        assert par["c5_stale"]["speedup_vs_n1"] is None and not par["c5_stale"]["parity_ok"]


class _Mailbox:
    """Emulated all-gather for ShardedIIR: ranks run one after the other in this process; rank r
    only folds rows k < r, which the earlier ranks of the sweep have already contributed."""
    def __init__(self, world):
        self.world, self.rank, self.rows = world, 0, {}

    def allgather_state(self, vec):
        self.rows[self.rank] = np.array(vec, copy=True)
        return np.stack([self.rows.get(k, np.zeros_like(vec)) for k in range(self.world)])


@pytest.mark.parametrize("world", [1, 3, 8])
def test_sharded_iir_chain_emulated(world):
    """s_{r+1} = f_r + A^{n_r} s_r over `world` ragged shards, with a non-zero zi for the whole
    signal; per-shard kernel = scipy's sosfilt (the reference's own IIR engine)."""
    from scipy import signal
    from sk_dsp_comm_amd import sharding
    rng = np.random.default_rng(7)
    n = 40_003
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    for sos in (signal.cheby1(5, 0.05, 0.8 / 12, output="sos"), np.array([[0.5, 0.5, 0.0, 1.0, -1.0, 0.0]])):
        nsec = sos.shape[0]
        zi = rng.standard_normal((nsec, 2)) + 1j * rng.standard_normal((nsec, 2))
        want = signal.sosfilt(sos, x, zi=zi)[0]
        def kern(xl, z):
            return signal.sosfilt(sos, xl, zi=np.zeros((nsec, 2), dtype=complex) if z is None else z)
        tr = _Mailbox(world)
        got = []
        for r, (a, b) in enumerate(sharding.shard_bounds(n, world)):
            tr.rank = r
            iir = sharding.ShardedIIR(sos, tr, dtype=np.complex128, kernel=kern, head_quantum=256)
            got.append(iir.filter_local_host(x[a:b], zi))
            if sos.shape[0] > 1 and r > 0:
                assert iir.head_length(b - a) < b - a  # the decaying filter re-filters only a head
        assert np.max(np.abs(np.concatenate(got) - want)) / np.max(np.abs(want)) < 1e-12
        assert len(tr.rows) == world


def test_sos_state_matrix_matches_sosfilt():
    from scipy import signal
    from sk_dsp_comm_amd import sharding
    sos = signal.ellip(7, 0.3, 50, 0.3, output="sos")
    A = sharding.sos_state_matrix(sos)
    z = np.random.default_rng(3).standard_normal((sos.shape[0], 2))
    _, zf = signal.sosfilt(sos, np.zeros(5), zi=z)
    assert np.allclose(np.linalg.matrix_power(A, 5) @ z.ravel(), zf.ravel(), rtol=1e-13, atol=1e-15)


def test_bench_reads_board_power_and_clock_from_rocm_smi(monkeypatch):
    """bench.py's board leg parses `rocm-smi --showpower --showmaxpower --showclocks`; the text below is what the tool prints on an MI355X box
    (profiles/r04/power_probe.txt).  A box without the tool yields (None, None, None) and the leg reports itself as skipped."""
    import types
    import bench
    text = ("============================ ROCm System Management Interface ============================\n"
            "GPU[0]\t\t: fclk clock level: 0: (1250Mhz)\nGPU[0]\t\t: mclk clock level: 0: (2000Mhz)\n"
            "GPU[0]\t\t: sclk clock level: 1: (2121Mhz)\nGPU[0]\t\t: socclk clock level: S: (38Mhz)\n"
            "GPU[0]\t\t: Max Graphics Package Power (W): 1400.0\n"
            "GPU[0]\t\t: Current Socket Graphics Package Power (W): 1394.0\n")
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: types.SimpleNamespace(stdout=text))
    assert bench._smi_sample() == (1394.0, 1400.0, 2121.0)

    def missing(*a, **k):
        raise FileNotFoundError("rocm-smi")
    monkeypatch.setattr(subprocess, "run", missing)
    assert bench._smi_sample() == (None, None, None)


def test_bench_config5_n1_reference_is_hash_checked(tmp_path, monkeypatch):
    """bench.py's N > 1 line quotes its speed-up over the committed 1-GPU run of the same 2^30 samples
    (profiles/rNN/bench_fir1024_2p30_one_gpu.json); a line recorded with other kernel sources is withheld."""
    import json
    import bench
    real_root = bench.ROOT
    d = tmp_path / "profiles" / "r99"
    d.mkdir(parents=True)
    line = {"ms_per_step": 3.7, "value": 290000.0, "n_gpus": 1, "config": {"total_samples": 1 << 30},
            "kernel_source_sha256": bench.source_hashes("fir1024")}
    (d / "bench_fir1024_2p30_one_gpu.json").write_text(json.dumps(line) + "\n")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    # (source_hashes reads the kernel sources under ROOT: keep it on the real tree)
    monkeypatch.setattr(bench, "source_hashes", lambda w, _r=real_root, _f=bench.source_hashes: _with_root(bench, _r, _f, w))
    ref = bench.config5_n1_reference()
    assert ref["ms"] == 3.7 and "stale" not in ref
    line["kernel_source_sha256"] = {"fir_ols.hip": "0" * 16, "ols_core.hpp": "0" * 16}
    (d / "bench_fir1024_2p30_one_gpu.json").write_text(json.dumps(line) + "\n")
    ref = bench.config5_n1_reference()
    assert ref["ms"] is None and "stale" in ref
    line["config"]["total_samples"] = 1 << 26
    (d / "bench_fir1024_2p30_one_gpu.json").write_text(json.dumps(line) + "\n")
    assert bench.config5_n1_reference()["ms"] is None


def _with_root(bench, root, fn, w):
    saved = bench.ROOT
    bench.ROOT = root
    try:
        return fn(w)
    finally:
        bench.ROOT = saved
