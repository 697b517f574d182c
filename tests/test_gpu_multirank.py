"""bench.py's N-rank path rehearsed on one GPU: two ranks (gloo, sharing the device) render
interleaved row tiles, rank 0 gathers them and the frame must equal the single-GPU frame bit for
bit.  On an 8-GPU node the same code runs with RCCL (backend nccl) and one GPU per rank."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_assemble_the_single_gpu_frame():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29517", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "5", "--warmup", "1", "--backend", "gloo", "--check"]   # five steps: two frames in flight, all three buffers reused
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["frame_matches_single_gpu"] is True
    assert res["config"]["rays_per_frame"] > 5.0e8 and res["value"] > 0
    assert "roofline" in res and res["roofline"]["frac"] > 0
