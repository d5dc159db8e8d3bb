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


def run_bench(ranks, extra):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, NOVA_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", str(ranks), "--steps", "3", "--warmup", "1", "--width", "448", "--height", "256",
           "--check-frame"] + extra
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_two_ranks_on_one_gpu_gather_the_same_frame():
    out = run_bench(2, ["--scaling", "weak", "--grid", "96"])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak"
    assert out["frame_check"] is True
    assert out["value"] > 0 and out["rays_per_step"] >= 448 * 256
    assert out["config"]["tiles"].startswith("64x64")


def test_three_ranks_strong_scaling_with_fxaa_gather_the_same_frame():
    """The default N > 1 run (BASELINE configs[2]: room stand-in, fixed frame, "strong") with FXAA on: the
    float4 means are gathered and rank 0 post-processes the whole frame — bytes equal to one GPU's."""
    out = run_bench(3, ["--anti-aliasing"])
    assert out["n_gpus"] == 3 and out["scaling"] == "strong"
    assert out["frame_check"] is True, out.get("frame_bytes_differing")
    assert "fireplace_room stand-in" in out["config"]["workload"]
    one = out["same_workload_one_gpu"]
    assert one["ms_per_step"] > 0 and abs(one["rays_per_step"] - out["rays_per_step"]) < 0.02 * out["rays_per_step"]


def test_bench_starts_its_own_ranks_when_no_launcher_did():
    """`python bench.py --gpus 2` — no torchrun, WORLD_SIZE unset — must start its ranks itself (a
    child torch.distributed.run made before the parent touches the GPU), pass the JSON line through and return the
    child's exit code.  On this one-GPU box the child is the rehearsal mode (two ranks share the device, gloo),
    and the line says which devices the ranks saw."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "NOVA_BENCH_BACKEND")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--width", "448", "--height", "256", "--check-frame"]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert res.returncode == 0, res.stderr[-2000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["frame_check"] is True
    r = out["rccl"]
    assert r["world_size"] == 2 and r["ranks_seen"] == 2 and len(r["device_per_rank"]) == 2
    assert r["distinct_devices"] == 1 and "rehearsal" in r["note"]  # one GPU here; 8 distinct bus ids on the node
    assert r["gather_ms"] > 0


def test_the_rccl_path_of_the_bench_on_a_world_of_one_rank():
    """What an 8-GPU node runs, as far as one GPU can: backend nccl (RCCL) for the process group, the library's own
    communicator from a unique id that torch.distributed hands round, and the gather through the C ABI
    (nova_hip_tiles_allgather) inside the timed region — with a world of ONE rank under the launcher
    (NOVA_BENCH_FORCE_DIST=1).  FXAA on, so the float4 means travel and the frame is post-processed after the gather."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k != "NOVA_BENCH_BACKEND"}
    env.update(NOVA_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", "1", "--steps", "2", "--warmup", "1", "--width", "448", "--height", "256", "--check-frame",
           "--anti-aliasing"]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert out["frame_check"] is True and out["scaling"] == "strong"
    r = out["rccl"]
    assert r["backend"].startswith("nccl") and r["gather"].startswith("nova_hip_tiles_allgather")
    assert r["library_comm_ranks"] == [1] and r["rccl_version"] > 0 and r["distinct_devices"] == 1
