"""bench.py's N-rank paths rehearsed on one GPU: two ranks (gloo, sharing the device) split the
work as they would on two GPUs, and every frame rank 0 (or, for frame-parallel rounds, its owner)
ends up with must equal the single-GPU frame bit for bit.  On an 8-GPU node the same code runs with
RCCL (backend nccl) and one GPU per rank."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _bench(*extra, ranks=2):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", str(ranks), "--backend", "gloo", "--check", "--no-cpu-baseline", *extra]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=400, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def test_two_ranks_assemble_the_single_gpu_frame():
    """Headline config, interleaved row tiles + gather.  Five steps: two frames in flight, all
    three buffers reused."""
    res = _bench("--steps", "5", "--warmup", "1")
    assert res["n_gpus"] == 2 and res["frame_matches_single_gpu"] is True
    assert res["config"]["rays_per_frame"] > 5.0e8 and res["value"] > 0
    assert "roofline" in res and res["roofline"]["frac"] > 0
    assert "k % N" in res["config"]["partition"]                      # step k's frame is assembled on rank k % N ...
    res = _bench("--steps", "3", "--warmup", "1", "--gather-root", "0", "--spp", "16")
    assert res["frame_matches_single_gpu"] is True and "rank 0" in res["config"]["partition"]   # ... or always on rank 0


def test_two_ranks_cornell_tiles_and_sample_split():
    """configs[2] (Cornell box): row tiles, and sample-split with integer sums + int64 reduce; the
    reduced frame equals the one-GPU frame bit for bit (64 spp here instead of 1000)."""
    res = _bench("--config", "c3", "--spp", "64", "--steps", "3", "--warmup", "1")
    assert res["frame_matches_single_gpu"] is True and "row-tiles" in res["config"]["partition"]
    res = _bench("--config", "c3", "--spp", "65", "--partition", "samples", "--steps", "4", "--warmup", "1")
    assert res["frame_matches_single_gpu"] is True and "sample-split" in res["config"]["partition"]
    assert res["config"]["rays_per_frame"] > 1920 * 1080 * 65 * 2


def test_ranks_share_the_camera_path_by_frames_and_samples():
    """configs[4] (camera path): five frames on two ranks are two frame-parallel rounds plus one
    frame sample-split over both ranks; on three ranks, four frames are one round plus one
    sample-split frame whose windows are unequal (32 = 11 + 11 + 10)."""
    res = _bench("--config", "c5", "--spp", "32", "--steps", "5", "--warmup", "0")
    assert res["frame_matches_single_gpu"] is True and "4 of 5 steps" in res["config"]["partition"]
    res = _bench("--config", "c5", "--spp", "32", "--steps", "4", "--warmup", "1", ranks=3)
    assert res["frame_matches_single_gpu"] is True and "3 of 4 steps" in res["config"]["partition"]


def test_rccl_collectives_among_one_rank():
    """The collectives as an 8-GPU run issues them (backend nccl = RCCL, device tensors, a
    communicator bound to the device), among the one rank a one-GPU box can hold: row-tile gather +
    row index, int64 reduce of the integer sums + resolve; the frames equal the directly rendered
    ones bit for bit."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for extra in (("--config", "c2", "--partition", "tiles", "--spp", "8"),
                  ("--config", "c3", "--partition", "samples", "--spp", "9")):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--rehearse-collectives", "--check", "--no-cpu-baseline",
               "--steps", "4", "--warmup", "1", *extra]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=400, env=env, cwd=ROOT)
        assert out.returncode == 0, out.stderr[-3000:]
        res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        assert res["n_gpus"] == 1 and res["frame_matches_single_gpu"] is True
        assert "RCCL" in res["config"]["partition"]
