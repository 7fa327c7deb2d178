"""bench.py's N > 1 path (barrier + synchronize on both sides, MAX over ranks, rank 0 prints ONE JSON line) exercised
with two ranks on the single GPU of the test box: `torch.distributed.run --nproc-per-node 2` with
OSM_BENCH_BACKEND=gloo (both ranks drive cuda:0; on a multi-GPU node the same code runs one rank per GPU over RCCL).
Tiny UNet: a plumbing check of the contract, not a benchmark."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_one_json_line():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, OSM_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--tiny", "--image-size", "32", "--cpu-steps", "0"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["value"] > 0 and abs(d["value"] - 2 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-2 * d["value"]
    assert d["higher_is_better"] is True and "roofline" in d and "cpu_baseline" not in d


def test_bench_eight_ranks_full_size_dry_run(tmp_path):
    """The launcher at the world size the driver's SCALE run uses, on the one GPU of the test box: plain
    `bench.py --gpus 8` (it re-executes itself under torch.distributed.run), OSM_BENCH_BACKEND=gloo so that the 8 ranks
    share cuda:0 (8 x (weight images + 2.8 GB of activations) fits 288 GB).  Full-size model, 2 timed steps: one JSON line,
    n_gpus 8, value = 8 ranks x steps / max-rank time, every rank worked on ITS image (images[rank::8]) and rank 0's
    result is bit-identical to a 1-rank run of the same command."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = dict(os.environ, OSM_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", OSM_BENCH_DUMP=str(tmp_path))
    tail = ["--steps", "2", "--warmup", "1", "--cpu-steps", "0", "--secondary-steps", "0"]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"] + tail, env=env, capture_output=True,
                         text=True, timeout=1500, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["steps"] == 2 and d["scaling"] == "weak" and d["config"]["finite_outputs"]
    assert abs(d["value"] - 8 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-2 * d["value"]
    assert "cpu_baseline" not in d and "secondary" not in d and "roofline" in d
    ranks = [json.load(open(tmp_path / f"rank{r}_of_8.json")) for r in range(8)]
    assert [r["image_index"] for r in ranks] == list(range(8)) and all(r["finite"] for r in ranks)
    assert len({r["sha1"] for r in ranks}) == 8                       # eight different images, eight different results
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + tail, env=env, capture_output=True,
                         text=True, timeout=900, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-3000:]
    single = json.load(open(tmp_path / "rank0_of_1.json"))
    assert single["image_index"] == 0 and single["sha1"] == ranks[0]["sha1"]
