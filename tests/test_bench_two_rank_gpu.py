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
    tail = ["--steps", "2", "--warmup", "1", "--cpu-steps", "0", "--secondary-steps", "0", "--pmc", "off", "--scale-only"]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"] + tail, env=env, capture_output=True,
                         text=True, timeout=1500, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["steps"] == 2 and d["scaling"] == "weak" and d["config"]["finite_outputs"]
    assert abs(d["value"] - 8 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-2 * d["value"]
    assert "cpu_baseline" not in d and "secondary" not in d and "roofline" in d and "full_chain" not in d and "long_window" not in d
    # a scaling line says that every rank reported and which device each one drove (here the eight ranks
    # share cuda:0 by construction -- OSM_BENCH_BACKEND=gloo -- so `devices_distinct` is False and that is not an error), and the
    # transport negotiation is visible on stderr as soon as it is known
    assert d["ranks_seen"] == 8 and d["ranks_complete"] is True and d["per_rank_device"] == [0] * 8 and d["devices_distinct"] is False
    assert out.stderr.count("[RankSync rank") >= 8 and "transport = gloo" in out.stderr
    # start-up of the eight ranks of ONE node: per-rank seconds from process start to the timed window -- host
    # initialisation of 552.8 M parameters (a per-parameter-seeded thread pool since round 5: it was 2 x 8 s of one sequential
    # generator), the 2.2 GB upload, weight-image packing (0.06 s of device kernels: no cache needed), plan recording + graph capture
    setup = d["per_rank_setup_s"]
    print("per_rank_setup_s at world = 8 (one GPU, one host):", setup)
    assert len(setup) == 8 and max(setup) < 90.0
    ranks = [json.load(open(tmp_path / f"rank{r}_of_8.json")) for r in range(8)]
    assert [r["image_index"] for r in ranks] == list(range(8)) and all(r["finite"] for r in ranks)
    assert len({r["sha1"] for r in ranks}) == 8                       # eight different images, eight different results
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + tail, env=env, capture_output=True,
                         text=True, timeout=900, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-3000:]
    single = json.load(open(tmp_path / "rank0_of_1.json"))
    assert single["image_index"] == 0 and single["sha1"] == ranks[0]["sha1"]


def test_bench_single_gpu_line_with_the_whole_chain_legs():
    """The N = 1 line the driver records (BENCH_rNN.json), with the slow side legs switched off (no PMC child runs, no CPU baseline,
    one step of the config 3 / 5 legs): the contract fields, `long_window`, `full_chain` (one complete Osmosis image) and
    `rgb_guidance_chain` (one complete image of the shipped rgb-guidance config: fused, FINITE outputs -- clip_denoised bounds the chain)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--cpu-steps", "0", "--pmc", "off",
                          "--secondary-steps", "1", "--long-window", "8"], capture_output=True, text=True, timeout=1500, cwd=ROOT,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["unit"] == "denoise-steps/sec" and d["vs_baseline"] is None
    assert abs(d["value"] - 4 / (d["ms_per_step"] * 4e-3)) < 1e-2 * d["value"] and 30 < d["value"] < 90
    assert d["long_window"]["steps"] == 8 and d["long_window"]["finite_outputs"]
    assert d["full_chain"]["steps"] == 1000 and 10 < d["full_chain"]["wall_s"] < 40
    rg = d["rgb_guidance_chain"]
    assert "error" not in rg, rg
    assert rg["fused_loop"] is True and rg["finite_outputs"] is True and rg["steps"] == 1000 and rg["max_abs_sample"] < 1.5
    assert 10 < rg["wall_s"] < 40 and abs(rg["ms_per_step"] - rg["wall_s"]) < 1e-6 * 1000
    assert [s["workload"][:8] for s in d["secondary"]] == ["config 4", "config 3", "config 5"]


def _bench_two(tmp, env_extra, tag):
    d = tmp / tag
    d.mkdir()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OSM_BENCH_DUMP=str(d), OSM_SYNC_TIMEOUT_S="45",
               OSM_SYNC_DIR=str(d / "sync"), **env_extra)
    if "OSM_BENCH_BACKEND" not in env_extra:
        env.pop("OSM_BENCH_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--tiny", "--image-size", "32", "--cpu-steps", "0"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    ranks = [json.load(open(d / f"rank{r}_of_2.json")) for r in range(2)]
    return json.loads(lines[0]), ranks


def test_bench_survives_rccl_failure(tmp_path):
    """the first real multi-GPU run must be un-losable.  Two ranks on the ONE GPU of the test box with the
    DEFAULT backend: RCCL is really tried and really fails (two ranks on one device) -- the failure branch the driver's 8-GPU
    box would take on an RCCL problem.  Every rank must fall back to gloo and still produce ONE JSON line whose per-rank
    results are those of the gloo run; with gloo failing too (forced), the per-rank-files transport carries the run."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    if torch.cuda.device_count() > 1:
        pytest.skip("RCCL would work here: the natural failure needs a single-GPU box")
    ref, ref_ranks = _bench_two(tmp_path, {"OSM_BENCH_BACKEND": "gloo"}, "gloo")
    assert ref["collective"] == "gloo" and ref["ranks_seen"] == 2 and len(ref["per_rank_ms"]) == 2
    got, got_ranks = _bench_two(tmp_path, {}, "default")
    assert got["collective"] == "gloo" and "rccl" in got["collective_failures"], got.get("collective_failures")
    files, files_ranks = _bench_two(tmp_path, {"OSM_SYNC_FORCE_FAIL": "rccl,gloo"}, "files")
    assert files["collective"] == "files" and set(files["collective_failures"]) == {"rccl", "gloo"}
    for d, ranks in ((got, got_ranks), (files, files_ranks)):
        assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["per_rank_image"] == [0, 1] and d["scaling"] == "weak"
        assert abs(d["value"] - 2 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-2 * d["value"]
        assert max(d["per_rank_ms"]) <= d["ms_per_step"] * 1.001
        assert [r["sha1"] for r in ranks] == [r["sha1"] for r in ref_ranks]        # same images, same bits, whatever the transport


def test_bench_two_ranks_full_size_with_the_config4_leg(tmp_path):
    """What the driver's SCALE run executes at N > 1 with the default flags, on the one GPU of the test box (gloo, two ranks share
    cuda:0): the headline leg, then the config-4 leg on EVERY rank (a batch of `--images-per-gpu` chains per rank through the same
    model / weight images, barriers through RankSync), one JSON line whose first `secondary` entry is that leg."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = dict(os.environ, OSM_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", OSM_BENCH_DUMP=str(tmp_path))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--cpu-steps", "0",
                          "--secondary-steps", "1", "--images-per-gpu", "2"], env=env, capture_output=True, text=True, timeout=1200,
                         cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["collective"] == "gloo" and d["ranks_seen"] == 2 and d["per_rank_image"] == [0, 1]
    assert "cpu_baseline" not in d and len(d["secondary"]) == 1          # N > 1: only the leg every rank runs
    c4 = d["secondary"][0]
    assert "error" not in c4, c4
    assert c4["workload"].startswith("config 4") and c4["images_per_gpu"] == 2 and c4["n_gpus"] == 2 and c4["ranks_seen"] == 2
    assert c4["finite_outputs"] and abs(c4["image_steps_per_s"] - 2 * 2 * 1 / (c4["ms_per_step"] * 1e-3)) < 1e-2 * c4["image_steps_per_s"]
    assert len({json.load(open(tmp_path / f"rank{r}_of_2.json"))["sha1"] for r in range(2)}) == 2
