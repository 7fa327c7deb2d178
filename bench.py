#!/usr/bin/env python3
"""Benchmark of the Osmosis guided-diffusion hot path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): osmosis_sample_config.yaml -- the 552.8 M-parameter RGBD UNet
(256 ch, mult 1,1,2,2,4,4, attention at 32/16/8), 1x4x256x256 per GPU, 1000-step DDPM (linear
schedule, epsilon mean, learned-range variance), `underwater_physical_revised` operator, 'osmosis'
guidance with n_iter=20 inner phi iterations, scale 7,7,7,0.9, clip 0.005, aux losses 0.5/20.
One "step" = one iteration of the reference loop (gaussian_diffusion.py:213): UNet forward, posterior,
20x (physics loss + phi SGD), UNet input-gradient, guidance update, noise.  Timed steps are taken
from the phi-update regime (t <= 0.7 T, the expensive 70 % of the chain), starting at t = 0.3 T from a bounded
x_t so that the un-trained network's x0 prediction stays inside the physical model's range.  Weights are seeded
synthetic (no checkpoint is available offline), inputs synthetic; timing is value independent.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the Winograd 3x3 convolution): `frac` is the
EXECUTED fraction -- MFMA flops the matrix cores ran per launch (algorithmic flops x 16/36 x MFMAs per product) / the
kernel's duration / the dense MFMA peak of the operand type -- with the duration taken from a kernel trace of a child run
of this script (`kernel_only_avg_us`) and, beside it, from HIP events around the launches on the launch stream
(`frac_hip_event`, `avg_launch_ms`: kernel + split-K combine); the algorithmic figure is `algorithmic_frac_of_direct_roof`.
`cpu_baseline` times the CPU oracle (oracle/, torch-CPU fp32 restatement of the reference) on the host cores, rank 0, N=1.
`--scale-only`: the N = 2 / 4 / 8 form (headline leg + config-4 leg, nothing else).
"""
import argparse
import json
import os
import sys
import time

# the host driver only supports dmabuf IPC: without this RCCL's first exchange fails with `hipIpcGetMemHandle: invalid argument`
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

UNET_KW = dict(image_size=256, num_channels=256, num_res_blocks=2, channel_mult="", learn_sigma=True,
               class_cond=False, use_checkpoint=False, attention_resolutions="32, 16, 8", num_heads=4,
               num_head_channels=64, num_heads_upsample=-1, use_scale_shift_norm=True, dropout=0.0,
               resblock_updown=True, use_fp16=False, use_new_attention_order=False, model_path="",
               pretrain_model="osmosis")
DIFFUSION = dict(sampler="ddpm", steps=1000, noise_schedule="linear", model_mean_type="epsilon",
                 model_var_type="learned_range", dynamic_threshold=False, clip_denoised=False,
                 rescale_timesteps=False, timestep_respacing=1000)
OPERATOR = dict(optimizer="sgd", depth_type="gamma", value="1.4,1.4,1", phi_a="1.1,0.95,0.95", phi_a_eta="1e-5",
                phi_a_learn_flag=True, phi_b="0.95, 0.8, 0.8", phi_b_eta="1e-5", phi_b_learn_flag=True,
                phi_inf="0.14, 0.29, 0.49", phi_inf_eta="1e-5", phi_inf_learn_flag=True)
COND = dict(loss_function="norm", loss_weight="depth", weight_function="gamma,1.4,1.4,1", scale="7,7,7,0.9",
            gradient_x_prev=True, gradient_clip="True,0.005")
PATTERN = dict(pattern="pcgs", update_start=0.7, update_end=0, global_N=1, local_M=1, s_start=1, s_end=0, n_iter=20,
               start_guidance=1, stop_guidance=0)
AUX = {"avrg_loss": 0.5, "val_loss": 20}
# configs/rgb_guidance_sample_config.yaml as the reference parses it (the keys sampling.restore_image reads; pinned to
# tests/golden/configs.json by tests/test_postprocess.py): ddpm, `ps` conditioning, clip_denoised True, gaussian noiser with sigma 0
RGB_GUIDANCE = dict(
    manual_seed=0, degamma_input=False, rgb_guidance=True, sample_pattern=PATTERN, unet_model=dict(pretrain_model="osmosis"),
    diffusion=dict(DIFFUSION, clip_denoised=True),
    conditioning=dict(method="ps", params=dict(loss_function="norm", loss_weight="depth", weight_function="gamma,1.4,1.4,1",
                                               scale="3,3,3,0.1", gradient_x_prev=True, gradient_clip="False,0.001")),
    aux_loss=dict(aux_loss=None),
    measurement=dict(operator=dict(name="rgb_guidance"), noise=dict(name="gaussian", sigma=0)))
FP32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 2.4 GHz
BF16_MFMA_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA (no sparsity)


def shard(n_items: int, rank: int, world: int):
    """images[rank::world] -- every image is an independent chain."""
    from osmosis_diffusion_code_amd.sharding import shard_indices
    return shard_indices(n_items, rank, world)


def synthetic_inputs(image_index: int, B: int, size: int):
    g = torch.Generator().manual_seed(1000 + image_index)
    x_T = torch.randn(B, 4, size, size, generator=g)
    y = torch.rand(B, 3, size, size, generator=g) * 1.6 - 0.8
    return x_T, y


def seeded_weights(model, seed=1234):
    """Deterministic non-degenerate weights for the product model (no oracle import on this path)."""
    model.reset_parameters(seed)


# the two other single-GPU configurations BASELINE.json names, timed in the same run (a few steps each) and reported under
# "secondary" of the one JSON line: config 3 = osmosis_simulation_sample_config.yaml as one batch of 8; config 5 =
# osmosis_haze_sample_config.yaml as BASELINE.json quotes it (batch 32, 250-step respacing, use_fp16)
SECONDARY = [
    dict(workload="config 3: osmosis_simulation_sample_config.yaml, B=8 underwater_physical, 1000-step DDPM + guidance, fp32 storage",
         batch=8, unet=dict(), diffusion=dict(), dtype="f32",
         operator=("underwater_physical", dict(optimizer="sgd", depth_type="original", value="1.4,1.4,1", phi_ab="1.1,0.95,0.95",
                                               phi_ab_eta="1e-5", phi_ab_learn_flag=True, phi_inf="0.2,0.4,0.7",
                                               phi_inf_eta="1e-5", phi_inf_learn_flag=True)),
         cond=dict(loss_function="norm", loss_weight="depth", weight_function="gamma,1.4,1.4,1", scale="4,4,4,1",
                   gradient_x_prev=True, gradient_clip="True,0.001"), aux={"val_loss": 40}),
    dict(workload="config 5: osmosis_haze_sample_config.yaml, B=32 haze_physical, 250-step respaced DDPM + guidance, use_fp16",
         batch=32, unet=dict(use_fp16=True), diffusion=dict(timestep_respacing="250"), dtype="f16",
         operator=("haze_physical", dict(optimizer="sgd", depth_type="gamma", value="1.4,1.4,1", phi_ab=1.0, phi_ab_eta="1e-5",
                                         phi_ab_learn_flag=True, phi_inf="0.14, 0.29, 0.49", phi_inf_eta="1e-5",
                                         phi_inf_learn_flag=True)),
         cond=dict(COND), aux=dict(AUX)),
]


def build_case(args, dev, batch, unet_kw=None, diffusion_kw=None, operator=None, cond_kw=None, aux=None, conv_mode=None,
               model=None):
    """(model, sampler, conditioner) of one configuration; the headline run is build_case(args, dev, args.batch).
    `model`: reuse this UNet (and its packed weight images) instead of building one."""
    from osmosis_diffusion_code_amd.guided_diffusion import condition_methods as CM
    from osmosis_diffusion_code_amd.guided_diffusion import gaussian_diffusion as gd
    from osmosis_diffusion_code_amd.guided_diffusion import measurements as M
    from osmosis_diffusion_code_amd.guided_diffusion import unet

    kw = dict(UNET_KW, **(unet_kw or {}))
    if args.tiny:
        kw.update(num_channels=32, num_res_blocks=1, channel_mult="1,2,2", attention_resolutions="128,64",
                  num_head_channels=16)
    import contextlib
    import io
    if model is None:
        with contextlib.redirect_stdout(io.StringIO()):     # create_model prints its "no checkpoint" notice
            model = unet.create_model(**kw)
        seeded_weights(model)
        model = model.to(dev).eval()
        if conv_mode is not None and not kw.get("use_fp16"):
            model.conv_mode = conv_mode
    sampler = gd.create_sampler(**dict(DIFFUSION, **(diffusion_kw or {})))
    opname, opkw = operator or ("underwater_physical_revised", OPERATOR)
    op = M.get_operator(opname, device=dev, batch_size=batch, **opkw)
    cond = CM.get_conditioning_method("osmosis", op, M.get_noise("clean"), **(cond_kw or COND), **PATTERN,
                                      aux_loss=aux if aux is not None else AUX)
    return model, sampler, cond


def timed_steps(args, dev, model, sampler, cond, batch, image_index, steps, warmup, world=1, sync=None, dump=False):
    """`warmup` untimed + `steps` timed guided steps; returns (seconds of the timed steps, outputs finite?).
    `sync` (sharding.RankSync) supplies the barrier of the N > 1 contract over whatever transport works on this node."""
    x_T, y = synthetic_inputs(image_index, batch, args.image_size)
    x_T, y = x_T.to(dev), y.to(dev)
    T = sampler.num_timesteps
    # Timed window inside the phi-update regime (t <= 0.7 T), started at t = 0.3 T from a bounded x_t: the seeded
    # synthetic weights do not denoise, so from t = 0.7 T the x0 prediction (x_t / sqrt(alpha_bar) ...) leaves the
    # physical model's range and the squared residual overflows fp32.  Kernel time is value-independent;
    # this only keeps the reported outputs finite.
    first = min(int(0.3 * T) - 1, T - 1)
    x_T = 0.5 * x_T
    torch.manual_seed(4242 + image_index)       # the per-step noise of this image's chain (device generator)

    def run(n_steps, start):
        return sampler.p_sample_loop(model=model, x_start=x_T, measurement=y, measurement_cond_fn=cond.conditioning,
                                     record=False, save_root=None, pretrain_model="osmosis", rgb_guidance=False,
                                     sample_pattern=PATTERN, index_range=(start, start - n_steps + 1),
                                     reference_rng_order=False)

    if warmup > 0:
        run(warmup, first)
    dsync = sync.device_synchronize if sync is not None else torch.cuda.synchronize
    dsync()                     # this rank's warm-up is done before it enters the barrier
    if world > 1:
        sync.barrier()
    dsync()
    t0 = time.perf_counter()
    out = run(steps, first - warmup)
    dsync()
    t_own = time.perf_counter() - t0        # this rank alone (reported per rank; the job time is the MAX below)
    if world > 1:
        sync.barrier()
    dt = time.perf_counter() - t0
    timed_steps.last_own_s = t_own
    dump = os.environ.get("OSM_BENCH_DUMP") if dump else None   # tests: per-rank checksum of the headline leg's final x_t
    if dump:
        import hashlib
        x = out[0].detach().cpu().contiguous()
        with open(os.path.join(dump, f"rank{int(os.environ.get('RANK', '0'))}_of_{world}.json"), "w") as f:
            json.dump({"image_index": image_index, "sha1": hashlib.sha1(x.numpy().tobytes()).hexdigest(),
                       "abs_sum": float(x.double().abs().sum()), "finite": bool(torch.isfinite(x).all())}, f)
    return dt, bool(torch.isfinite(out[0]).all())


def run_gpu(args, rank, world, dev, sync):
    """The headline leg on this rank.  Returns (model, job seconds = MAX over ranks of barrier-to-barrier time, finite,
    per-rank rows [rank, image index, own ms per step, barrier-to-barrier ms per step, setup s, finite, device index])."""
    t_setup = time.perf_counter()
    model, sampler, cond = build_case(args, dev, args.batch, conv_mode=args.conv_mode)
    image_index = shard(world, rank, world)[0]
    # weight packing (device kernels, ~14 GB of images per GPU) happens at the first engine build inside the warm-up:
    # every rank packs on ITS OWN GPU; what the ranks share is host memory bandwidth for the 2.2 GB parameter upload
    dt, finite = timed_steps(args, dev, model, sampler, cond, args.batch, image_index, args.steps, args.warmup, world, sync,
                             dump=True)
    setup_s = time.perf_counter() - t_setup - dt
    rows = sync.all_gather([rank, image_index, 1e3 * timed_steps.last_own_s / args.steps, 1e3 * dt / args.steps, setup_s,
                            1.0 if finite else 0.0, float(torch.cuda.current_device())])
    dt = max(r[3] for r in rows) * args.steps / 1e3
    return model, dt, all(r[5] == 1.0 for r in rows), rows


def run_config4(args, dev, model, rank, world, sync):
    """BASELINE config 4's real shape -- osmosis_sample_config.yaml on a 64-image set, 8 images per GPU carried as ONE
    batch of 8 independent chains -- on every rank of this job (same model / weight images as the headline leg): whole-job
    image-steps/s = world x 8 x steps / MAX over ranks of the barrier-to-barrier time."""
    B = args.images_per_gpu
    steps, warmup = max(1, args.secondary_steps), 1
    # The leg has a FIXED round schedule -- two barriers inside timed_steps, one gather -- and a rank whose leg raises (an
    # out-of-memory at B = 8 on one GPU only ...) still enters every one of them with a "failed" row: no rank is left waiting
    # in a round its peer skipped, and all ranks report the failure instead of some hanging until a timeout.
    round0 = sync.round
    err, own_ms, dt_ms, finite = None, float("nan"), float("nan"), False
    try:
        _, sampler, cond = build_case(args, dev, B, model=model)
        dt, finite = timed_steps(args, dev, model, sampler, cond, B, 100 + rank, steps, warmup, world, sync)
        own_ms, dt_ms = 1e3 * timed_steps.last_own_s / steps, 1e3 * dt / steps
    except Exception as e:      # a secondary line must never cost the headline number
        err = f"{type(e).__name__}: {e}"[:300]
    if world > 1:
        while sync.round < round0 + 2:      # the barriers of timed_steps this rank did not reach
            sync.barrier()
    rows = sync.all_gather([rank, own_ms, dt_ms, 1.0 if finite else 0.0, 0.0 if err is None else 1.0])
    try:
        model._engines = {k: v for k, v in model._engines.items() if k[0] != B}     # give the B = 8 activations back
        torch.cuda.empty_cache()
    except Exception:
        pass
    failed = [int(r[0]) for r in rows if r[4] != 0.0]
    if failed:
        return {"workload": "config 4", "error": err or f"failed on rank(s) {failed}", "failed_ranks": failed}
    dtm = max(r[2] for r in rows) * steps / 1e3
    return {"workload": f"config 4: osmosis_sample_config.yaml, 64-image set sharded {B} per GPU (one batch of {B} chains per rank), "
                        "underwater_physical_revised, 1000-step DDPM + guidance, fp32 storage",
            "images_per_gpu": B, "n_gpus": world, "dtype": "f32", "conv_arithmetic": model.conv_mode, "steps": steps,
            "warmup": warmup, "ms_per_step": round(1e3 * dtm / steps, 2),
            "image_steps_per_s": round(world * B * steps / dtm, 2), "ms_per_image_step": round(1e3 * dtm / steps / B, 3),
            "finite_outputs": all(r[3] == 1.0 for r in rows),
            "per_rank_ms": [round(r[1], 3) for r in sorted(rows)], "ranks_seen": len(rows)}


def full_chain(args, dev, model):
    """ONE COMPLETE image of the headline configuration through the per-image driver (`sampling.restore_image`: fresh operator /
    conditioner / sampler, manual_seed, x_T ~ N(0, I), all 1000 steps of gaussian_diffusion.py:213 -- the frozen-phi regime
    t > 0.7 T and the phi-update regime -- and the host post-processing): north_star's unit, images/s at 1000 DDPM steps,
    MEASURED rather than extrapolated from the timed window.  Seeded weights do not denoise, so the chain's values leave the
    physical model's range from t ~ 0.7 T (the reference does the same); kernel time is value independent."""
    from osmosis_diffusion_code_amd import sampling
    cfg = dict(measurement=dict(operator=dict(name="underwater_physical_revised", **OPERATOR), noise=dict(name="clean")),
               conditioning=dict(method="osmosis", params=COND), sample_pattern=PATTERN, aux_loss=dict(aux_loss=AUX),
               diffusion=dict(DIFFUSION), unet_model=dict(pretrain_model="osmosis"), manual_seed=0, degamma_input=False,
               rgb_guidance=False)
    _, y = synthetic_inputs(0, args.batch, args.image_size)
    y = y.to(dev)
    T = int(DIFFUSION["steps"])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = sampling.restore_image(model, y, cfg)[0]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"what": "one complete 1000-step image through sampling.restore_image (operator / sampler construction, every step of the "
                    "chain, host post-processing), wall clock with device synchronisation on both sides",
            "steps": T, "images": args.batch, "wall_s": round(dt, 3), "denoise_steps_per_sec": round(T * args.batch / dt, 3),
            "images_per_sec": round(args.batch / dt, 6), "conv_arithmetic": model.conv_mode,
            "finite_outputs": bool(torch.isfinite(res["pred_xstart"]).all() and torch.isfinite(res["sample"]).all()),
            "finite_note": "seeded synthetic weights: the free-running chain leaves the physical model's range from t ~ 0.7 T in "
                           "every arithmetic and in the reference itself (SURVEY F10); timing is value independent"}


def rgb_guidance_chain(args, dev, model):
    """ONE COMPLETE image of the reference's SHIPPED rgb-guidance configuration (configs/rgb_guidance_sample_config.yaml: DDPM.p_sample
    + `ps` conditioning on the identity operator, clip_denoised True) through `sampling.restore_image` on the headline
    network: all 1000 steps on the fused kernels (UNet forward and input gradient, osm_posterior_typed with the clamp, osm_phys_*
    kind 3, osm_clamp_bwd, osm_guide_update_rng).  The clamp bounds pred_xstart, so this chain stays FINITE on seeded weights from
    x_T ~ N(0, I) to the end -- the one complete full-size chain of the bench line whose outputs are finite."""
    from osmosis_diffusion_code_amd import sampling
    from osmosis_diffusion_code_amd.guided_diffusion import gaussian_diffusion as gd
    _, y = synthetic_inputs(0, 1, args.image_size)
    y = y.to(dev)
    fell_back = []
    orig = gd.GaussianDiffusion._generic_loop

    def spy(self, *a, **k):
        fell_back.append(1)
        return orig(self, *a, **k)
    gd.GaussianDiffusion._generic_loop = spy
    try:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = sampling.restore_image(model, y, RGB_GUIDANCE, noise_seed=0)[0]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        gd.GaussianDiffusion._generic_loop = orig
    T = int(DIFFUSION["steps"])
    s = res["sample"]
    return {"workload": "rgb_guidance_sample_config.yaml: one complete 1000-step image (ddpm + ps, clip_denoised), B = 1, seeded weights",
            "steps": T, "wall_s": round(dt, 3), "denoise_steps_per_sec": round(T / dt, 3), "ms_per_step": round(1e3 * dt / T, 3),
            "conv_arithmetic": model.conv_mode, "fused_loop": not fell_back, "finite_outputs": bool(torch.isfinite(s).all()),
            "max_abs_sample": round(float(s.abs().max()), 4),
            "rel_residual_rgb": round(float((y.cpu()[0] - res["rgb"]).norm() / y.cpu().norm()), 4)}


def run_secondary(args, dev):
    """Configs 3 and 5 on this GPU, a few steps each: image-steps/s of the whole guided step + the 3x3-conv class."""
    out = []
    for c in SECONDARY:
        try:
            model, sampler, cond = build_case(args, dev, c["batch"], c["unet"], c["diffusion"], c["operator"], c["cond"],
                                              c["aux"], conv_mode=args.conv_mode)
            steps, warmup = args.secondary_steps, 1
            dt, finite = timed_steps(args, dev, model, sampler, cond, c["batch"], 7, steps, warmup)
            a2 = argparse.Namespace(**dict(vars(args), batch=c["batch"], conv_mode=model.conv_mode, dump_layers=""))
            rl, br = roofline(model, a2, reps=1)
            out.append({"workload": c["workload"], "images_per_gpu": c["batch"], "dtype": c["dtype"],
                        "conv_arithmetic": model.conv_mode, "steps": steps, "warmup": warmup,
                        "ms_per_step": round(1e3 * dt / steps, 2),
                        "image_steps_per_s": round(c["batch"] * steps / dt, 2), "finite_outputs": finite,
                        "roofline_kernel": rl["kernel"], "roofline_frac": rl["frac"], "roofline_achieved_tflops": rl["achieved"],
                        "roofline_frac_is": "executed MFMA flops / HIP-event launch time / dense MFMA peak",
                        "roofline_algorithmic_frac_of_direct_roof": rl["algorithmic_frac_of_direct_roof"],
                        "kernel_breakdown_ms_per_step": {k: round(v["ms_per_step"], 2) for k, v in br.items() if not k.startswith("_")}})
            del model, sampler, cond
            torch.cuda.empty_cache()
        except Exception as e:      # a secondary line must never cost the headline number
            out.append({"workload": c["workload"], "error": f"{type(e).__name__}: {e}"[:300]})
    return out


CONV_KINDS = {0: "f32", 1: "igemm", 2: "direct", 3: "direct8", 4: "direct", 5: "winograd"}   # osm_conv_kernel_kind
# dominant-kernel facts per (class, arithmetic): kernel name (as rocprofv3 prints it), MFMAs per product
KERNELS = {
    ("conv3x3_winograd", "f16x3"): ("conv3_wino8_kernel<2,*,true>", 3, "f16"),
    ("conv3x3_winograd", "bf16x6"): ("conv3_wino8_kernel<3,*,false>", 6, "bf16"),
    ("conv3x3_winograd", "bf16x3"): ("conv3_wino8_kernel<2,*,false>", 3, "bf16"),
    ("conv3x3_winograd", "f16"): ("conv3_wino8_kernel<1,*,false>", 1, "f16"),
    ("conv3x3_direct", "f16x3"): ("conv3_halo_bf16s_kernel<3,*>", 6, "bf16"),
    ("conv3x3_direct", "bf16x6"): ("conv3_halo_bf16s_kernel<3,*>", 6, "bf16"),
    ("conv3x3_direct", "bf16x3"): ("conv3_halo_bf16s_kernel<2,*>", 3, "bf16"),
    ("conv3x3_direct", "f16"): ("conv3_halo_bf16s_kernel<1,*>", 1, "f16"),
    ("conv3x3_f32", "f32"): ("igemm_f32_kernel<9,false>", 0, "f32"),
}


def roofline(model, args, reps=3):
    """HIP-event timing of every launch of one guided step's UNet plans, classified by the kernel the library really
    launches for each call (osm_conv_kernel_kind).  The roofline object is for the class that takes the most time."""
    from osmosis_diffusion_code_amd import ops as _ops
    B, S = args.batch, args.image_size
    eng = model.engine(B, S, S)

    def select(name, a):
        if name.endswith("_h"):          # fp16-storage family: same operations
            name = name[:-2]
        if name == "osm_conv2d_nhwc":
            d = a[0]._obj
            fl = 2.0 * d.B * d.H * d.W * d.Cin * d.Cout * d.ksize * d.ksize
            shape = (d.B, d.H, d.W, d.Cin, d.Cout, d.ksize, d.splitk)
            per_shape.setdefault(shape, [0.0, 0, fl])
            kind = CONV_KINDS[_ops.query("osm_conv_kernel_kind", d.B, d.H, d.W, d.Cin, d.Cout, d.ksize, d.wfmt)]
            if d.ksize != 3:
                tag = "conv1x1" if kind != "f32" else "conv1x1_f32"
            else:
                tag = {"winograd": "conv3x3_winograd", "direct": "conv3x3_direct", "direct8": "conv3x3_8x8",
                       "igemm": "conv3x3_tapchunked", "f32": "conv3x3_f32"}[kind]
            # what THIS launch has to move, given its tiling and its fused epilogue (the PMC traffic is compared with it):
            # x once, its halo overlap (patches of 16 x 16 / 8 x 16 / 8 x 8 pixels stage a one-pixel border), x again for every group
            # of column tiles beyond the first (Winograd: groups of <= 4 tiles of 64 columns share a patch in time), the weight
            # image once, the result (split-K: fp32 partials), the residual / previous output, the GroupNorm input of the fused
            # backward reductions
            esz_ = 2.0 if d.wfmt & 0xf == 1 else 4.0
            Mx = float(d.B * d.H * d.W)
            wf_ = d.wfmt & 0xf
            wbytes_per = {0: 4.0, 1: 2.0, 2: 4.0, 3: 6.0, 4: 4.0}[wf_]
            if kind == "winograd":
                ph, pw, taps_img = min(d.H, 16), min(d.W, 16), 16
                ntile = (d.Cout + 63) // 64
                nb1_ = max(g for g in (1, 2, 3, 4) if ntile % g == 0)
                groups = ntile // nb1_
                # the eight XCDs have private L2s: an XCD fetches the weight fragments of every column tile one of ITS workgroups
                # works on (ids are dealt to the XCDs in contiguous ranges; id -> (group, M-tile, column tile in the group) as in
                # conv3_wino8_kernel).  Beyond the first fetch these come from the Infinity Cache, and FETCH_SIZE counts them
                mt_ = d.B * ((d.H + 15) // 16) * ((d.W + 15) // 16)
                nt_, gsz_ = mt_ * ntile, nb1_ * mt_
                q_, r_ = divmod(nt_, 8)
                fetched_tiles, lo_ = 0, 0
                for xcd_ in range(8):
                    n_ = q_ + (1 if xcd_ < r_ else 0)
                    fetched_tiles += len({(i_ // gsz_) * nb1_ + i_ % nb1_ for i_ in range(lo_, lo_ + n_)})
                    lo_ += n_
                wino_xcd_refetch = max(0, fetched_tiles - ntile) / float(ntile)
            elif kind in ("direct", "direct8"):
                ph, pw, taps_img, groups = 8, (16 if d.W >= 16 else 8), d.ksize * d.ksize, 1   # column tiles of a patch: adjacent ids, one XCD
            else:
                ph, pw, taps_img, groups = 0, 0, d.ksize * d.ksize, 1
            halo = ((ph + 2) * (pw + 2) / float(ph * pw) - 1.0) if (ph and d.ksize == 3) else 0.0
            comp = {"x": esz_ * Mx * d.Cin, "x_halo_overlap": esz_ * Mx * d.Cin * halo,
                    "x_again_per_column_tile_group": esz_ * Mx * d.Cin * (1.0 + halo) * (groups - 1),
                    "weights": wbytes_per * taps_img * d.Cin * d.Cout,
                    "weights_again_per_xcd": wbytes_per * taps_img * d.Cin * d.Cout * (wino_xcd_refetch if kind == "winograd" else 0.0),
                    "result": (4.0 * d.splitk if d.splitk > 1 else esz_) * Mx * d.Cout,
                    "residual_or_previous_output": esz_ * Mx * d.Cout * ((1 if d.res else 0) + (1 if d.accumulate else 0)) * (d.splitk <= 1),
                    "groupnorm_input_of_fused_backward_sums": esz_ * Mx * d.Cout * (1 if (d.colsum and d.stat_mode == 2 and d.splitk <= 1) else 0)}
            ob = opbytes.setdefault(tag, {})
            for k_, v_ in comp.items():
                ob[k_] = ob.get(k_, 0.0) + v_
            return (tag, fl, shape)
        if name == "osm_maxabs":
            return ("maxabs", 0.0, None)
        if name == "osm_gemm":
            d = a[0]._obj
            fl = 2.0 * d.M * d.N * d.K * d.nb1 * d.nb2
            shape = ("osm_gemm", d.M, d.N, d.K, d.nb1 * d.nb2, int(d.b_kn))
            per_shape.setdefault(shape, [0.0, 0, fl])
            return ("attn_gemm", fl, shape)
        if name in ("osm_attn_flash_fwd", "osm_attn_flash_bwd", "osm_attn_small_fwd", "osm_attn_small_bwd"):
            d = a[0]._obj
            # algorithmic flops of the core: forward 2 GEMMs (QK^T, PV), backward 5 (S, dP, dq, dk, dv) of 2 T^2 ch each
            fl = 2.0 * d.B * d.heads * d.T * d.T * d.ch * (2 if name.endswith("fwd") else 5)
            shape = (name, d.B, d.T, d.heads, d.ch)
            per_shape.setdefault(shape, [0.0, 0, fl])
            return ("attention_core", fl, shape)
        if name in ("osm_softmax_rows", "osm_softmax_rows_bwd"):
            i0 = 3 if name == "osm_softmax_rows" else 4
            shape = (name, int(a[i0]), int(a[i0 + 1]))
            per_shape.setdefault(shape, [0.0, 0, 4.0 * shape[1] * shape[2] * shape[2] * (3 if i0 == 3 else 4)])
            return ("softmax", 0.0, shape)
        if name == "osm_gn_finalize_cols":
            shape = (name, int(a[2]), int(a[1]), int(a[4]))        # B, chunks, C
            per_shape.setdefault(shape, [0.0, 0, 8.0 * shape[1] * shape[2] * shape[3]])
            return ("groupnorm", 0.0, shape)
        if name.startswith("osm_gn"):
            i0 = {"osm_gn_stats": 2, "osm_gn_apply": 4, "osm_gn_bwd": 10, "osm_gn_bwd_apply": 10, "osm_gn_fwd": 4,
                  "osm_gn_prep": 2}[name]     # position of (B, HW, C) in the C argument list
            shape = (name, int(a[i0]), int(a[i0 + 1]), int(a[i0 + 2]))
            passes = {"osm_gn_stats": 1, "osm_gn_apply": 2, "osm_gn_fwd": 3, "osm_gn_prep": 1,
                      "osm_gn_bwd": 4 + bool(a[6]) + bool(a[8]), "osm_gn_bwd_apply": 3 + bool(a[6]) + bool(a[8])}[name]
            per_shape.setdefault(shape, [0.0, 0, 4.0 * shape[1] * shape[2] * shape[3] * passes])
            return ("groupnorm", 0.0, shape)
        return ("other", 0.0, None)

    agg = {}
    per_shape = {}
    shape_tag = {}
    opbytes = {}
    for _ in range(reps):
        for plan in (eng._fwd_plan, eng._bwd_plan):
            for (tag, fl, shape), ms in plan.replay_timed(select):
                if shape is not None:
                    per_shape[shape][0] += ms
                    per_shape[shape][1] += 1
                    shape_tag[shape] = tag
                a = agg.setdefault(tag, [0.0, 0.0, 0])
                a[0] += ms
                a[1] += fl
                a[2] += 1
    out = {k: {"ms_per_step": v[0] / reps, "gflop_per_step": v[1] / reps / 1e9, "launches_per_step": v[2] // reps}
           for k, v in agg.items()}
    if args.dump_layers:
        rows = [{"B,H,W,Cin,Cout,k,splitk": list(k), "class": shape_tag.get(k), "launches_per_step": v[1] // reps,
                 "ms_per_launch": v[0] / v[1], "tflops": v[2] / (v[0] / v[1]) / 1e9} for k, v in per_shape.items()]   # GN rows: "tflops" = TB/s
        rows.sort(key=lambda r: -r["ms_per_launch"] * r["launches_per_step"])
        with open(args.dump_layers, "w") as f:
            json.dump(rows, f, indent=0)
    # the levels at <= 32 x 32 pixels (weight-stream / launch bound: < 3 % of the flops): their convolutions' share of the step
    conv_shapes = {k: v for k, v in per_shape.items() if len(k) == 7 and isinstance(k[0], int)}
    lowres_ms = sum(v[0] for k, v in conv_shapes.items() if k[1] <= 32) / reps
    out["_lowres"] = {"convs_le_32x32_ms_per_step": round(lowres_ms, 3),
                      "launches_per_step": sum(v[1] for k, v in conv_shapes.items() if k[1] <= 32) // reps}

    # ---- the bandwidth-bound class (north_star: "achieved HBM GB/s"): GroupNorm (+ SiLU / FiLM) passes.  Bytes = the ALGORITHMIC
    # bytes of every launch (tensor bytes x the passes that launch makes over it: select() above), time = HIP events around
    # the same launches in the replayed step.  `streaming` = the launches that move >= 8 MB (the ones a bandwidth roof applies
    # to); the rest are latency-floor kernels (a dependent launch costs ~5 us whatever it moves).
    esz_act = 2.0 if args.conv_mode == "f16" else 4.0
    gshapes = {k: v for k, v in per_shape.items() if shape_tag.get(k) == "groupnorm"}

    def _bw(items):
        by = sum(v[2] * (esz_act / 4.0 if k[0] != "osm_gn_finalize_cols" else 1.0) * v[1] for k, v in items) / reps
        ms = sum(v[0] for k, v in items) / reps
        n = sum(v[1] for k, v in items) // reps
        return by, ms, n
    if gshapes:
        by_all, ms_all, n_all = _bw(gshapes.items())
        big = [(k, v) for k, v in gshapes.items() if v[2] * (esz_act / 4.0) >= 8e6]
        by_big, ms_big, n_big = _bw(big) if big else (0.0, 0.0, 0)
        out["_hbm"] = {
            "bound": "hbm", "class": "GroupNorm (+ SiLU, FiLM, residual) passes of the UNet forward and input-gradient",
            "kernel": "gn_apply_kernel / gn_bwd_apply_kernel / gn_reduce_kernel / gn_reg_kernel (csrc/norm.hip)",
            "achieved": round(by_big / max(ms_big, 1e-9) / 1e6, 1), "peak": 8000.0, "unit": "GB/s",
            "frac": round(by_big / max(ms_big, 1e-9) / 1e6 / 8000.0, 4),
            "achieved_is": "algorithmic bytes of the streaming launches (>= 8 MB each) / their HIP-event time in the replayed step",
            "streaming_launches_per_step": n_big, "streaming_ms_per_step": round(ms_big, 3),
            "streaming_bytes_per_step": round(by_big),
            "whole_class_gbps": round(by_all / max(ms_all, 1e-9) / 1e6, 1), "whole_class_ms_per_step": round(ms_all, 3),
            "whole_class_launches_per_step": n_all, "whole_class_bytes_per_step": round(by_all),
            "copy_yardstick_gbps": 4800.0,
            "note": "peak = 8 TB/s HBM3E (MI355X_MICROARCH.md); a plain float4 copy on this pool's boxes reaches 4.8 TB/s at the "
                    "same grid size (tools/ubench/copy_bw.hip): the 256^2 x 256-channel passes run at ~0.9 of that copy, the 8-32 MB "
                    "ones below it; the launches under 8 MB are bound by the ~5 us floor of a dependent launch, not by bandwidth"}

    conv3 = {k: v for k, v in out.items() if k.startswith("conv3x3")}
    dom = max(conv3, key=lambda k: conv3[k]["ms_per_step"])
    c = out[dom]
    achieved = c["gflop_per_step"] / c["ms_per_step"]  # GFLOP/ms == TFLOP/s
    kname, nmfma, mtype = KERNELS.get((dom, args.conv_mode), (dom, 6, "bf16"))
    wino = dom == "conv3x3_winograd"
    if nmfma == 0:
        peak, note = FP32_MFMA_PEAK_TFLOPS, "exact-fp32 MFMA v_mfma_f32_32x32x2_f32"
    else:
        peak = BF16_MFMA_PEAK_TFLOPS / nmfma
        note = {("f16", 3): "fp32 operands scaled into the fp16 range by powers of two and split into 2 IEEE-half terms (~22-bit "
                            "operands), 3 fp16 MFMAs per product, fp32 accumulation (fp32-class accuracy: tests vs fp64 hold it to "
                            "the same 4e-6 as bf16x6)",
                ("bf16", 6): "fp32 operands split exactly into 3 bf16 terms, 6 bf16 MFMAs per fp32 product (fp32-class accuracy)",
                ("bf16", 3): "fp32 operands split into 2 bf16 terms, 3 bf16 MFMAs per product (~2^-16 relative)",
                ("f16", 1): "fp16 activations x fp16 weights, fp32 accumulation (the reference's use_fp16): one fp16 MFMA per product",
                }.get((mtype, nmfma), "") + f": peak = dense {mtype} MFMA peak 2500 TFLOP/s / {nmfma}; achieved counts ALGORITHMIC flops"
    # `frac` is the EXECUTED fraction: MFMA flops the matrix cores really ran (algorithmic flops x 16/36 for
    # Winograd F(2x2,3x3) x the MFMAs of one product in this arithmetic) / kernel time / the DENSE peak of the MFMA operand type.
    # <= 1 by construction.  The algorithmic figure against the roof a direct convolution in this arithmetic would have
    # (peak / MFMAs per product) stays beside it as `algorithmic_frac_of_direct_roof` (it tops out at 36/16 for Winograd).
    mfma_peak = FP32_MFMA_PEAK_TFLOPS if nmfma == 0 else BF16_MFMA_PEAK_TFLOPS
    exec_ratio = (16.0 / 36.0 if wino else 1.0) * max(nmfma, 1)
    executed = achieved * exec_ratio
    extra = {"executed_tflops": round(executed, 2), "executed_frac": round(executed / mfma_peak, 4),
             "executed_is": f"algorithmic flops x {'16/36 (Winograd F(2x2,3x3)) x ' if wino else ''}{max(nmfma, 1)} MFMA(s) per product",
             "mfma_dense_peak_tflops": mfma_peak,
             "algorithmic_tflops": round(achieved, 2), "algorithmic_roof_tflops": round(peak, 1),
             "algorithmic_frac_of_direct_roof": round(achieved / peak, 4)}
    if wino:
        note += ("; the Winograd F(2x2,3x3) kernel EXECUTES 16/36 of the algorithmic multiply-adds: `achieved` / `frac` count what the "
                 "matrix cores did, `algorithmic_tflops` what a direct convolution would have to do")
    extra["share_of_3x3_time"] = round(c["ms_per_step"] / sum(v["ms_per_step"] for v in conv3.values()), 4)
    # HBM bytes per launch of the dominant kernel RELAYED from the committed PMC summary (rocprofv3 --pmc FETCH_SIZE /
    # WRITE_SIZE in separate passes, tools/pmc_summary.py): a builder-side measurement, not something this run observed
    traffic, traffic_src = None, None
    import glob
    pref = kname.split("<")[0]
    targs = kname.split("<")[1].rstrip(">").split(",") if "<" in kname else []
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_hbm_traffic.json")))[-1:]:
        try:
            def match(k):
                kn = k["kernel"].replace(" ", "")
                if not kn.startswith(pref + "<"):
                    return False
                ka = kn.split("<")[1].rstrip(">").split(",")
                return all(t == "*" or (i < len(ka) and ka[i] == t) for i, t in enumerate(targs))
            hit = [k for k in json.load(open(path))["kernels"] if match(k)]
            if hit:   # launch-weighted mean over the template instances of the dominant kernel
                traffic = round(sum(k["hbm_bytes_per_launch"] * k["launches"] for k in hit) / sum(k["launches"] for k in hit))
                traffic_src = os.path.relpath(path, ROOT)
        except Exception:
            pass
    traffic_how = None if traffic is None else "builder profile (committed PMC summary), relayed -- not observed by this run"
    traffic_detail = None
    if getattr(args, "pmc", "off") == "auto" and reps > 1:          # the headline leg only (secondary legs pass reps=1)
        live, detail = live_traffic(args, kname)
        if live is not None:
            traffic, traffic_src, traffic_detail = live, "this run", detail
            traffic_how = "measured in this run (child processes of bench.py under rocprofv3 --pmc)"
        else:
            traffic_detail = {"live_measurement_failed": detail}
    dshapes = {k: v for k, v in per_shape.items() if shape_tag.get(k) == dom}
    esz = 2.0 if args.conv_mode == "f16" else 4.0
    alg_bytes = sum(esz * (k[0] * k[1] * k[2] * (k[3] + k[4]) + 9 * k[3] * k[4]) * v[1]
                    for k, v in dshapes.items()) / max(1, sum(v[1] for v in dshapes.values()))
    att = {k: out[k] for k in ("attention_core", "attn_gemm", "softmax") if k in out}
    if att:
        ams = sum(v["ms_per_step"] for v in att.values())
        agf = sum(v["gflop_per_step"] for v in att.values())
        att_f16x3 = args.conv_mode == "f16x3" and os.environ.get("OSM_ATTN_F16X3", "1") != "0"
        n_att = 1 if args.conv_mode == "f16" else (3 if att_f16x3 else 6)
        peak_att = BF16_MFMA_PEAK_TFLOPS / n_att
        out["_attention"] = {
            "what": "attention cores (QK^T, softmax, PV and their gradients; T = 64 / 256 / 1024, 64-wide heads)",
            "arithmetic": {1: "f16 (one fp16 MFMA per product)", 3: "f16x3 (two half terms per operand, three fp16 MFMAs per product; "
                           "operand ranges found in the kernels)", 6: "bf16x6 (six bf16 MFMAs per product)"}[n_att],
            "ms_per_step": round(ams, 3), "achieved_tflops": round(agf / max(ams, 1e-9), 2),
            "executed_mfma_tflops": round(n_att * agf / max(ams, 1e-9), 2),
            "peak_tflops": round(peak_att, 1), "mfma_util": round(agf / max(ams, 1e-9) / peak_att, 4),
            "note": f"algorithmic flops / HIP-event time of the launches / (dense MFMA peak 2500 TF / {n_att} MFMAs per product); "
                    "B = 1 has 8-16 (image, head) pairs of <= 1024 tokens: chains of dependent L2 round trips on one round of workgroups -- "
                    "halving the MFMAs (bf16x6 -> f16x3, round 6) moved the class by 2 % (profiles/NOTES_r06.md section 3); the counter-based "
                    "matrix-pipe busy fraction of these kernels is in profiles/r06_pmc_mfma_busy.json (0.094-0.098 at T = 1024)"}
    lt = c["ms_per_step"] / c["launches_per_step"]
    # kernel-only duration of the same kernel from a kernel trace taken in THIS run (a child run of this script under
    # `rocprofv3 --kernel-trace`, one guided step): what profiles/rNN_rocprofv3_kernel_stats.csv reports.  The HIP-event time
    # brackets one osm_conv2d_nhwc call, i.e. the kernel AND the split-K combine of the launches that have one.
    ktrace = None
    if getattr(args, "pmc", "off") == "auto" and reps > 1:
        ktrace = live_kernel_time(args, kname)
    k_us = ktrace.get("avg_us") if isinstance(ktrace, dict) else None
    fl_launch = c["gflop_per_step"] * 1e9 / c["launches_per_step"]
    exec_event = fl_launch * exec_ratio / (lt * 1e-3) / 1e12
    exec_kernel = None if not k_us else fl_launch * exec_ratio / (k_us * 1e-6) / 1e12
    best = exec_kernel if exec_kernel is not None else exec_event
    return {"bound": "mfma", "kernel": kname + " (3x3 conv fwd + dgrad)", "class": dom, "arithmetic": note,
            "achieved": round(best, 2), "peak": mfma_peak, "unit": "TFLOP/s",
            "frac": round(best / mfma_peak, 4),
            "frac_is": ("EXECUTED MFMA flops per launch / kernel-only time of that kernel (kernel trace of a child run of this script, "
                        "this box) / dense MFMA peak of the operand type" if exec_kernel is not None else
                        "EXECUTED MFMA flops per launch / HIP-event time of the launch (kernel + split-K combine) / dense MFMA peak"),
            "frac_hip_event": round(exec_event / mfma_peak, 4),
            "frac_kernel_trace": None if exec_kernel is None else round(exec_kernel / mfma_peak, 4),
            "kernel_only_avg_us": None if not k_us else round(k_us, 2), "kernel_trace": ktrace,
            **extra,
            "frac_of_fp32_mfma_peak": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),   # algorithmic, against 157.3 TF (exact-fp32 MFMA / vector peak)
            "traffic": traffic, "traffic_source": traffic_src,
            "traffic_measured_in": traffic_how, "traffic_detail": traffic_detail,
            # achieved HBM GB/s of the conv kernel (north_star asks for it; the kernel is MFMA / power bound, not HBM-bound):
            # relayed PMC bytes and algorithmic bytes, each / the live HIP-event launch time
            "hbm_gbps": None if traffic is None else round(traffic / lt / 1e6, 1),
            "algorithmic_gbps": round(alg_bytes / lt / 1e6, 1),
            "algorithmic_bytes_per_launch_avg": round(alg_bytes),
            # the bytes the launches of this class have to move given their tiling and fused epilogues, by component (average per
            # launch), and the counter traffic over their sum (FETCH_SIZE counts L2 misses that the Infinity Cache serves as well:
            # the per-XCD weight fetches and the column-group re-reads of x are of that kind, not HBM reads)
            "operand_bytes_per_launch_avg": {k_: round(v_ / reps / c["launches_per_step"]) for k_, v_ in opbytes.get(dom, {}).items()},
            "operand_bytes_per_launch_total": round(sum(opbytes.get(dom, {}).values()) / reps / c["launches_per_step"]),
            "traffic_over_operand_bytes": None if traffic is None else round(
                traffic / max(1.0, sum(opbytes.get(dom, {}).values()) / reps / c["launches_per_step"]), 3),
            "flop_per_launch_avg": fl_launch, "executed_mfma_flop_per_launch_avg": fl_launch * exec_ratio,
            "avg_launch_ms": lt, "avg_launch_ms_is": "HIP events around one osm_conv2d_nhwc call (kernel + split-K combine where there is one)",
            "launches_per_step": c["launches_per_step"]}, out


def _kernel_matcher(kname):
    pref = kname.split("<")[0]
    targs = kname.split("<")[1].rstrip(">").split(",") if "<" in kname else []

    def match(name):
        kn = name.replace(" ", "")
        i = kn.find(pref + "<")
        if i < 0:
            return False
        ka = kn[i + len(pref) + 1:].split(">")[0].split(",")
        return all(t == "*" or (j < len(ka) and ka[j] == t) for j, t in enumerate(targs))
    return match


def live_kernel_time(args, kname):
    """Kernel-only duration of the dominant kernel, MEASURED IN THIS RUN: one child run of this script (1 warm-up + 1 guided
    step) under `rocprofv3 --kernel-trace` (no counters, no other trace domain); the average of End - Start over the kernel's
    dispatches -- the quantity `rocprofv3 --kernel-trace --stats` prints as AverageNs.  Returns a dict or {"error": ...}."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return {"error": "rocprofv3 not on PATH"}
    match = _kernel_matcher(kname)
    d = tempfile.mkdtemp(prefix="osm_kt_", dir="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
           "--steps", "1", "--warmup", "1", "--cpu-steps", "0", "--secondary-steps", "0", "--pmc", "off",
           "--conv-mode", args.conv_mode, "--batch", str(args.batch), "--image-size", str(args.image_size)]
    try:
        env = {k: v for k, v in os.environ.items() if k != "OSM_BENCH_DUMP"}
        subprocess.run(cmd, cwd="/tmp", env=dict(env, TMPDIR="/tmp"), stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL, timeout=args.pmc_timeout)
        tot, n, names = 0.0, 0, set()
        for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if match(r["Kernel_Name"]):
                    tot += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                    n += 1
                    kn_ = r["Kernel_Name"]
                    i_ = kn_.find(kname.split("<")[0])
                    names.add(kn_[i_:kn_.find(">", i_) + 1] if i_ >= 0 else kn_[:60])
        if n == 0:
            return {"error": f"no kernel-trace rows for {kname}"}
        return {"avg_us": tot / n / 1e3, "dispatches": n, "instances": sorted(names)[:4],
                "how": "rocprofv3 --kernel-trace, child run of this script (1 warm-up + 1 guided step); average End - Start of the "
                       "kernel's dispatches (both UNet plans, every template instance)"}
    except Exception as e:          # a profiler problem must never cost the bench line
        return {"error": f"{type(e).__name__}: {e}"[:200]}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def live_traffic(args, kname):
    """HBM-side bytes per launch of the dominant kernel, MEASURED IN THIS RUN: two child runs of this script (one guided step
    each) under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` -- separate passes, counters only, no trace domains, as
    MI355X_MICROARCH.md prescribes -- summarised like tools/pmc_summary.py (read side x 2: the gfx950 half-count of 16 B/lane
    streaming reads).  Returns (bytes per launch, detail dict) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    pref = kname.split("<")[0]
    targs = kname.split("<")[1].rstrip(">").split(",") if "<" in kname else []

    def match(name):
        kn = name.replace(" ", "")
        i = kn.find(pref + "<")
        if i < 0:
            return False
        ka = kn[i + len(pref) + 1:].split(">")[0].split(",")
        return all(t == "*" or (j < len(ka) and ka[j] == t) for j, t in enumerate(targs))

    got = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="osm_pmc_", dir="/tmp")
        cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
               "--steps", "1", "--warmup", "1", "--cpu-steps", "0", "--secondary-steps", "0", "--pmc", "off",
               "--conv-mode", args.conv_mode, "--batch", str(args.batch), "--image-size", str(args.image_size)]
        try:
            env = {k: v for k, v in os.environ.items() if k != "OSM_BENCH_DUMP"}
            subprocess.run(cmd, cwd="/tmp", env=dict(env, TMPDIR="/tmp"), stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL, timeout=args.pmc_timeout)
            tot, n = 0.0, 0
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                for r in csv.DictReader(open(f)):
                    if r["Counter_Name"] == counter and match(r["Kernel_Name"]):
                        tot += float(r["Counter_Value"])
                        n += 1
            if n == 0:
                return None, f"no {counter} rows for {kname}"
            got[counter] = (tot * 1024.0 / n, n)
        except Exception as e:          # a profiler problem must never cost the bench line
            return None, f"{counter} pass failed: {type(e).__name__}: {e}"[:200]
        finally:
            shutil.rmtree(d, ignore_errors=True)
    rd, wr = 2.0 * got["FETCH_SIZE"][0], got["WRITE_SIZE"][0]
    return round(rd + wr), {"read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
                            "launches_counted": got["FETCH_SIZE"][1],
                            "how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate child runs of this script (1 warm-up + 1 "
                                   "guided step each), read side x 2 (gfx950 half-count of 16 B/lane reads), KiB -> bytes"}


def cpu_baseline(args):
    """Time the CPU oracle (torch-CPU fp32 restatement of the reference path) on the host cores."""
    from oracle import diffusion_ref as D
    from oracle import unet_ref as U
    kw = dict(UNET_KW)
    if args.tiny:
        kw.update(num_channels=32, num_res_blocks=1, channel_mult="1,2,2", attention_resolutions="128,64",
                  num_head_channels=16)
    cfg = U.UNetConfig.from_create_model_kwargs(**kw)
    sd = U.seeded_state_dict(cfg, 1234)
    # thread sweep: oneDNN on a 128-thread host is FASTER with fewer threads than cores for these shapes (round 1
    # ran it oversubscribed: 18 s/step at 128 threads vs 9.7 s on 8 cores).  One UNet forward per candidate, best wins.
    ncpu = os.cpu_count() or 1
    cand = sorted({c for c in (8, 16, 32, 48, 64) if c <= ncpu} or {ncpu})   # > 64 threads: slower every time (measured)
    x_probe = torch.randn(1, 4, args.image_size, args.image_size)
    sweep = {}
    with torch.no_grad():
        for c in cand:
            torch.set_num_threads(c)
            U.unet_forward(sd, cfg, x_probe[:, :, :64, :64], torch.tensor([10.0]))     # warm the thread pool
            t0 = time.perf_counter()
            U.unet_forward(sd, cfg, x_probe, torch.tensor([10.0]))
            sweep[c] = time.perf_counter() - t0
    cores = min(sweep, key=sweep.get)
    torch.set_num_threads(cores)
    tb = D.make_tables(1000, "linear", 1000)
    rop = D.PhysOperator("underwater_physical_revised", batch_size=1, depth_type="gamma", value="1.4,1.4,1",
                         phi_a="1.1,0.95,0.95", phi_b="0.95, 0.8, 0.8", phi_inf="0.14, 0.29, 0.49")
    rg = D.OsmosisGuidance(rop, n_iter=20, scale=COND["scale"], gradient_clip=COND["gradient_clip"], aux=AUX)
    x_T, y = synthetic_inputs(0, 1, args.image_size)
    n_timed = args.cpu_steps
    sub = D.Tables(D.named_beta_schedule("linear", 1000), range(0, 1 + n_timed))   # t = n_timed .. 0 : phi-update regime
    noises = [torch.randn(1, 4, args.image_size, args.image_size) for _ in range(1 + n_timed)]
    model = lambda x, t: U.unet_forward(sd, cfg, x, t)  # noqa: E731
    times = []

    class Timed(list):
        def append(self, rec):
            times.append(time.perf_counter())
            super().append(rec)

    t0 = time.perf_counter()
    D.p_sample_loop(model, sub, 0.5 * x_T, y, rg, PATTERN, noises, Timed())
    steps = [b - a for a, b in zip([t0] + times[:-1], times)]
    timed = steps[1:]                                   # first step = warm-up
    sps = len(timed) / sum(timed)
    return {"value": round(sps, 5), "unit": "denoise-steps/sec", "cores": cores, "kind": "port",
            "thread_sweep_unet_fwd_s": {str(k): round(v, 2) for k, v in sweep.items()},
            "sample": f"{len(timed)} guided steps (B=1, 256x256, n_iter=20) after 1 warm-up step, "
                      f"torch-CPU fp32 oracle, {cores} threads (best of a {cand} sweep on one UNet forward, "
                      f"host has {ncpu} hardware threads); {sum(timed) / len(timed):.2f} s/step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed guided steps (>= 200: 3.7 s, box-to-box noise stops dominating)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--long-window", type=int, default=200,
                    help="N = 1: when --steps is smaller than this, the SAME timed leg is repeated over this many steps and "
                         "reported as `long_window` (0 = skip)")
    ap.add_argument("--full-chain", type=int, default=1,
                    help="N = 1: one complete 1000-step image through sampling.restore_image, reported as `full_chain` (0 = skip)")
    ap.add_argument("--batch", type=int, default=1, help="images per GPU (reference config: 1)")
    ap.add_argument("--image-size", type=int, default=256)
    ap.add_argument("--cpu-steps", type=int, default=2, help="timed CPU-oracle steps for cpu_baseline (0 = skip)")
    ap.add_argument("--conv-mode", default=os.environ.get("OSM_CONV_MODE", "f16x3"), choices=["f32", "bf16x6", "bf16x3", "f16", "f16x3"],
                    help="conv arithmetic: f16x3 (default: Winograd 3x3 layers on 3 fp16 MFMAs per product, the rest bf16x6), "
                         "exact-fp32 MFMA, or fp32 split into 3 / 2 bf16 terms (6 / 3 bf16 MFMAs)")
    ap.add_argument("--dump-layers", default="", help="write per-conv-shape timings (JSON) to this path")
    ap.add_argument("--tiny", action="store_true", help="tiny UNet (plumbing check only; NOT a valid bench)")
    ap.add_argument("--pmc", default="auto", choices=["auto", "off"],
                    help="auto: roofline.traffic is measured in this run (two short child runs under rocprofv3 --pmc, rank 0, N = 1); "
                         "off: relayed from the committed profile")
    ap.add_argument("--pmc-timeout", type=float, default=150.0, help="seconds allowed per rocprofv3 child run")
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 5],
                    help="2: the headline workload (default).  3 / 5: time ONLY that BASELINE configuration (profiling target) and "
                         "print its compact line")
    ap.add_argument("--images-per-gpu", type=int, default=8,
                    help="batch of the config-4 secondary leg (BASELINE config 4: 64 images sharded 8 per GPU); runs at every N")
    ap.add_argument("--secondary-steps", type=int, default=3,
                    help="timed steps of each secondary configuration (BASELINE configs 3 and 5; N = 1 only; 0 = skip)")
    ap.add_argument("--scale-only", action="store_true",
                    help="scaling runs (N = 2 / 4 / 8): the headline leg and the config-4 leg only -- no rocprofv3 child runs, no CPU "
                         "baseline, no complete chain, no long window, no bf16x6 / config 3 / config 5 legs: < 60 s per N after setup")
    args = ap.parse_args()
    if args.scale_only:
        args.pmc, args.cpu_steps, args.full_chain, args.long_window = "off", 0, 0, 0

    if args.gpus > 1 and "RANK" not in os.environ:
        # started plainly with --gpus N: become the one-process-per-GPU job the contract describes
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.execvp(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                                   f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
                                   "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product path has no CPU fallback)")
    # One rank per GPU.  OSM_BENCH_BACKEND=gloo lets the N > 1 path be exercised on a single-GPU box (ranks share device 0
    # and RCCL is not tried); the driver's multi-GPU runs use the default: RCCL first, and if it fails to initialise, to
    # all-reduce or to answer in time on ANY rank, every rank falls back to gloo, then to per-rank files (sharding.RankSync).
    backend = os.environ.get("OSM_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend != "nccl" or local >= ndev:
        local = local % ndev
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    from osmosis_diffusion_code_amd.sharding import RankSync
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    sync = RankSync(rank, world, device=dev, probe_timeout_s=float(os.environ.get("OSM_SYNC_TIMEOUT_S", "120")),
                    force_fail=() if backend == "nccl" else ("rccl",))

    if world > 1 or args.tiny:
        args.pmc = "off"
    if args.config != 2:        # profiling target: one secondary configuration alone
        c = SECONDARY[0 if args.config == 3 else 1]
        model, sampler, cond = build_case(args, dev, c["batch"], c["unet"], c["diffusion"], c["operator"], c["cond"], c["aux"],
                                          conv_mode=args.conv_mode)
        dt, finite = timed_steps(args, dev, model, sampler, cond, c["batch"], 7, args.steps, args.warmup)
        print(json.dumps({"workload": c["workload"], "images_per_gpu": c["batch"], "conv_arithmetic": model.conv_mode,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
                          "image_steps_per_s": round(c["batch"] * args.steps / dt, 2), "finite_outputs": finite}), flush=True)
        return
    model, dt, finite, rank_rows = run_gpu(args, rank, world, dev, sync)
    units = world * args.batch * args.steps
    line = {
        "metric": "denoise-steps/sec (256x256 RGBD, 1000-step DDPM+guidance)",
        "value": round(units / dt, 4), "unit": "denoise-steps/sec", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": "f16" if args.conv_mode == "f16" else ("f32" if args.conv_mode == "f32" else f"f32 ({args.conv_mode} conv)"),
        "data": "synthetic",
        "config": {"workload": "osmosis_sample_config.yaml: 1 underwater 256x256 image per GPU, 1000-step DDPM "
                               "+ osmosis guidance (n_iter=20); the timed window starts at t = 0.3 T, inside the phi-update regime "
                               "(t <= 0.7 T: 20 phi iterations per step, the expensive 70 % of the chain)",
                   "images_per_gpu": args.batch, "image_size": args.image_size, "unet_params": 552821000,
                   "weights": "seeded synthetic", "conv_arithmetic": args.conv_mode,
                   "headline_arithmetic": (f"`value` / `ms_per_step` / `roofline` are measured in {args.conv_mode}"
                                           + ("; `same_workload_bf16x6` is the identical run with every contraction in the exact "
                                              "three-term bf16 split (strictly fp32-class operands) -- quote both"
                                              if args.conv_mode == "f16x3" else "")),
                   "conv_arithmetic_note": ("f16x3: the Winograd 3x3 layers and the 1x1 layers at >= 64x64 multiply ~22-bit operands (two IEEE-half terms per fp32 "
                                            "value after a power-of-two scaling) with three fp16 MFMAs per product and fp32 accumulation; every other "
                                            "contraction is bf16x6 (exact three-term bf16 split, six MFMAs).  Tests hold both to the same "
                                            "4e-6 against fp64; `same_workload_bf16x6` is this run in bf16x6 everywhere")
                   if args.conv_mode == "f16x3" else None, "parallelism": f"images[rank::{world}] (no collective on the path)",
                   "groupnorm_reductions": os.environ.get("OSM_FUSE_STATS", "wino") + " (OSM_FUSE_STATS; wino = forward statistics and the "
                                           "backward reductions of GroupNorm from the Winograd convolutions' epilogues)",
                   "finite_outputs": finite},
        "images_per_sec_at_1000_steps": round(units / dt / 1000.0, 6),
        # how the ranks met (barrier, gather of per-rank times): "rccl" | "gloo" | "files" ("none" at N = 1); the data path has
        # no collective.  per_rank_ms: each rank's own synchronize-to-synchronize time per step; the job time is the MAX over
        # ranks of the barrier-to-barrier time
        "collective": sync.transport, "collective_failures": sync.failures or None,
        "ranks_seen": len(rank_rows), "per_rank_device": [int(r[6]) for r in sorted(rank_rows)],
        "per_rank_ms": [round(r[2], 3) for r in sorted(rank_rows)],
        "per_rank_image": [int(r[1]) for r in sorted(rank_rows)],
        "per_rank_setup_s": [round(r[4], 1) for r in sorted(rank_rows)],
    }
    if world > 1:
        # a scaling line is only a scaling line if every rank reported and (one rank per GPU) every rank drove its own device;
        # OSM_BENCH_BACKEND=gloo (ranks sharing device 0 on a one-GPU box: tests) is the declared exception
        line["ranks_complete"] = line["ranks_seen"] == world
        line["devices_distinct"] = len(set(line["per_rank_device"])) == world
        if rank == 0 and not line["ranks_complete"]:
            raise SystemExit(f"bench.py: {line['ranks_seen']} of {world} ranks reported -- not a valid N = {world} line")
        if rank == 0 and backend == "nccl" and ndev >= world and not line["devices_distinct"]:
            raise SystemExit(f"bench.py: ranks shared a device ({line['per_rank_device']}) on a node with {ndev} GPUs")
    rl = breakdown = None
    long_window = chain = rgb_chain = None
    if rank == 0:               # replays the headline engine's plans: before the config-4 leg replaces that engine
        rl, breakdown = roofline(model, args)
        if world == 1 and not args.tiny and args.secondary_steps > 0 and not args.scale_only:     # (--secondary-steps 0 = the headline leg alone: profiling runs)
            # same model, same engine, same start state as the headline leg -- only longer (the driver passes --steps 20)
            if 0 < args.steps < args.long_window:
                try:
                    _, s_lw, c_lw = build_case(args, dev, args.batch, model=model)
                    dt_lw, fin_lw = timed_steps(args, dev, model, s_lw, c_lw, args.batch, 0, args.long_window, args.warmup)
                    long_window = {"steps": args.long_window, "warmup": args.warmup, "ms_per_step": round(1e3 * dt_lw / args.long_window, 3),
                                   "value": round(args.batch * args.long_window / dt_lw, 4), "unit": "denoise-steps/sec",
                                   "seconds": round(dt_lw, 3), "finite_outputs": fin_lw,
                                   "what": "the headline leg over a longer timed window (same model / engine / start state)"}
                except Exception as e:
                    long_window = {"error": f"{type(e).__name__}: {e}"[:300]}
            if args.full_chain:
                try:
                    chain = full_chain(args, dev, model)
                except Exception as e:
                    chain = {"error": f"{type(e).__name__}: {e}"[:300]}
                try:
                    rgb_chain = rgb_guidance_chain(args, dev, model)
                except Exception as e:
                    rgb_chain = {"error": f"{type(e).__name__}: {e}"[:300]}
    cfg4 = run_config4(args, dev, model, rank, world, sync) if (args.secondary_steps > 0 and not args.tiny) else None
    if rank == 0:
        line["roofline"] = rl
        att = breakdown.pop("_attention", None)
        if att:
            line["attention"] = att
            line["attention_mfma_util"] = att["mfma_util"]
        line["lowres_levels"] = breakdown.pop("_lowres", None)
        line["roofline_hbm"] = breakdown.pop("_hbm", None)
        line["kernel_breakdown_ms_per_step"] = {k: round(v["ms_per_step"], 3) for k, v in breakdown.items()}
        if long_window is not None:
            line["long_window"] = long_window
        if rgb_chain is not None:
            line["rgb_guidance_chain"] = rgb_chain
        if chain is not None:
            line["full_chain"] = chain
            if "images_per_sec" in chain:      # north_star's unit, measured over a whole chain (the extrapolation stays beside it)
                line["images_per_sec_at_1000_steps_measured"] = chain["images_per_sec"]
        line["achieved_tflops_whole_step"] = round(
            sum(v["gflop_per_step"] for v in breakdown.values()) / (1e3 * dt / args.steps), 2)
        if world == 1 and args.secondary_steps > 0 and not args.tiny and not args.scale_only:
            del model
            model = None
            torch.cuda.empty_cache()
            if args.conv_mode == "f16x3":
                # the same workload in round 2's arithmetic (every fp32 operand split EXACTLY into three bf16 terms, six
                # MFMAs per product), measured in the same run: what the headline would be without the f16x3 Winograd images
                try:
                    m2, s2, c2 = build_case(args, dev, args.batch, conv_mode="bf16x6")
                    dt2, fin2 = timed_steps(args, dev, m2, s2, c2, args.batch, 0, args.steps, args.warmup)
                    line["same_workload_bf16x6"] = {"conv_arithmetic": "bf16x6", "value": round(args.batch * args.steps / dt2, 4),
                                                    "ms_per_step": round(1e3 * dt2 / args.steps, 3), "steps": args.steps,
                                                    "warmup": args.warmup, "finite_outputs": fin2}
                    del m2, s2, c2
                    torch.cuda.empty_cache()
                except Exception as e:
                    line["same_workload_bf16x6"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            line["secondary"] = run_secondary(args, dev)
        if cfg4 is not None:
            line.setdefault("secondary", []).insert(0, cfg4)
        if world == 1 and args.cpu_steps > 0:
            del model
            line["cpu_baseline"] = cpu_baseline(args)
            line["speedup_vs_cpu_baseline"] = round(line["value"] / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line), flush=True)
    if world > 1:
        sync.barrier()
    sync.close()
    if not sync.device_sync_safe:
        os._exit(0)             # an abandoned (hung) RCCL probe thread must not keep the process alive


if __name__ == "__main__":
    main()
