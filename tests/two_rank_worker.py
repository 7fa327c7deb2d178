"""Worker of tests/test_two_rank_gpu.py: one rank of a world_size-W job whose ranks all drive cuda:0 (gloo for the
bookkeeping collectives; on a multi-GPU node the same code runs one rank per GPU over RCCL).  Restores its shard
images[rank::W] with the HIP sampler and writes the per-image results to <out>/rank<r>.npz."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def make_inputs(n=4, size=32):
    g = torch.Generator().manual_seed(77)
    return [torch.rand(1, 3, size, size, generator=g) * 1.6 - 0.8 for _ in range(n)]


def make_model(dev):
    import baseline_configs as BC
    from oracle import unet_ref as U
    from osmosis_diffusion_code_amd.guided_diffusion import unet
    cfg = U.UNetConfig.from_create_model_kwargs(**BC.TINY_UNET)
    m = unet.create_model(**BC.TINY_UNET)
    m.load_state_dict(U.seeded_state_dict(cfg, 1234), strict=True)
    return m.to(dev).eval()


def run(rank, world, out_dir, batch_size=1, n_images=4):
    import baseline_configs as BC
    from osmosis_diffusion_code_amd import sampling
    from osmosis_diffusion_code_amd.sharding import gather_per_image, max_over_ranks
    dev = "cuda:0"
    model = make_model(dev)
    cfg = BC.with_unet(BC.SAMPLE, BC.TINY_UNET)
    images = make_inputs(n_images)
    res = sampling.restore_images(model, images, cfg, rank=rank, world=world, device=dev, batch_size=batch_size,
                                  index_range=(2, 0), x_scale=0.05)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"),
             **{f"x0_{i}": r["pred_xstart"].numpy() for i, r in res.items()},
             **{f"img_{i}": r["sample"].numpy() for i, r in res.items()},
             **{f"phi_inf_{i}": r["phi"]["phi_inf"].numpy() for i, r in res.items()})
    # the only cross-rank traffic of the path: bookkeeping (per-image norm loss gathered in image order, MAX of a time)
    losses = gather_per_image([r["norm_loss_final"] for r in res.values()], len(images))
    tmax = max_over_ranks(float(rank + 1))
    if rank == 0:
        np.savez(os.path.join(out_dir, "gathered.npz"), losses=np.asarray(losses), tmax=tmax)


if __name__ == "__main__":
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    run(rank, world, sys.argv[1], batch_size=int(sys.argv[2]) if len(sys.argv) > 2 else 1,
        n_images=int(sys.argv[3]) if len(sys.argv) > 3 else 4)
    dist.barrier()
    dist.destroy_process_group()
