"""SURVEY.md section 4 tier 5 / section 8e on the HIP path: the same per-image result on 1 rank, on 2 ranks and on 8 ranks
(the world size of the driver's scaling run).  W processes (gloo rendezvous on 127.0.0.1, all on cuda:0 -- the box has one
GPU) each restore images[rank::W] with the fused HIP sampler; the parent restores all images in one process and compares
bit for bit."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,batch_size,n_images", [(2, 1, 4), (2, 2, 4), (8, 1, 8)])
def test_ranks_equal_one_rank(tmp_path, world, batch_size, n_images):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import two_rank_worker as W
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "two_rank_worker.py"), str(tmp_path),
                                       str(batch_size), str(n_images)], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    one = tmp_path / "single"
    one.mkdir()
    W.run(0, 1, str(one), batch_size=1, n_images=n_images)
    ref = dict(np.load(one / "rank0.npz"))
    assert len(ref) == 3 * n_images
    seen = set()
    for r in range(world):
        got = dict(np.load(tmp_path / f"rank{r}.npz"))
        assert sorted(int(k.split("_")[-1]) for k in got if k.startswith("x0_")) == list(range(n_images))[r::world]
        for k, v in got.items():
            assert np.isfinite(v).all()
            if batch_size == 1:
                assert np.array_equal(v, ref[k]), f"rank {r} {k}: max diff {np.abs(v - ref[k]).max()}"
            else:       # a batch of 2 runs different tile shapes than batch 1: equal to rounding, not bit for bit
                assert np.abs(v - ref[k]).max() < 1e-5, k
            seen.add(k)
    assert seen == set(ref)
    g = np.load(tmp_path / "gathered.npz")
    assert g["losses"].shape == (n_images,) and np.isfinite(g["losses"]).all() and float(g["tmax"]) == float(world)
