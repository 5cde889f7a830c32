"""BASELINE config 5 with a real PEER: the RCCL halo of the sharded FIR (Ntaps-1 samples rank r -> r+1) between two or more
HIP devices.  Skipped unless the box exposes more than one logical device (an 8-GPU node, or ONE MI355X in CPX partition
mode: profiles/r05/cpx_probe/).  The ranks are the same `bench.py --gpus N` processes the driver launches; every rank checks the
outputs that consumed its neighbour's halo against the CPU oracle (bench.py: shard_parity)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _device_count():
    from sk_dsp_comm_amd import _ffi
    return _ffi.load().skdsp_device_count()


def _bench(n_ranks, extra, env=None):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n_ranks), "--steps", "6", "--warmup", "2",
           "--settle-seconds", "0.05", "--launch-timeout", "400", "--no-cpu-baseline"] + extra
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    e.update(env or {})
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=500, env=e)
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
    assert out.returncode == 0 and lines, (out.returncode, out.stdout.decode()[-1500:], out.stderr.decode()[-1500:])
    return json.loads(lines[-1])


@pytest.mark.parametrize("two_launches", [0, 1])
def test_sharded_fir_halo_meets_a_peer(two_launches):
    nd = _device_count()
    if nd < 2:
        pytest.skip("one logical HIP device on this box: the halo has no peer to meet (profiles/r05/cpx_probe/)")
    world = 8 if nd >= 8 else 2
    j = _bench(world, ["--scaling", "strong", "--total-log2n", "25"], {"SKDSP_SHARD_TWO_LAUNCHES": str(two_launches)})
    assert j["config"]["n_ranks_rccl"] == world
    assert j["parity_ok"] is True, j
    assert j["parity_halo_max_err"] <= 1e-6 and j["parity_interior_max_err"] <= 1e-6
    assert len(j["halo_fallback_used"]) == world
    if two_launches:
        assert all(j["halo_fallback_used"])


def test_sharded_iir_state_handoff_meets_a_peer():
    nd = _device_count()
    if nd < 2:
        pytest.skip("one logical HIP device on this box")
    j = _bench(2, ["--workload", "iir8", "--log2n", "22"])
    assert j["config"]["n_ranks_rccl"] == 2
    assert j["parity_ok"] is True, j
