"""GPU (-m gpu): bench.py's multi-rank path rehearsed on ONE GPU.  Two ranks share the device
(NOVA_BENCH_BACKEND=gloo: collectives through host memory instead of RCCL, everything else as in
the driver's N > 1 runs): tile partition by rank, per-rank counters summed, max-over-ranks timing,
one gather of the finished tiles — and the gathered frame must equal the frame one rank renders
alone, byte for byte."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_gpu_gather_the_same_frame():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, NOVA_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "3", "--warmup", "1", "--grid", "96", "--width", "448", "--height", "256",
           "--check-frame"]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak"
    assert out["frame_check"] is True
    assert out["value"] > 0 and out["rays_per_step"] >= 448 * 256
    assert out["config"]["tiles"].startswith("64x64")
