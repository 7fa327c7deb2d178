"""The parsed-YAML dictionaries of the BASELINE.json configurations (values as in the reference's
configs/osmosis_sample_config.yaml, osmosis_simulation_sample_config.yaml, osmosis_haze_sample_config.yaml),
in the shape `sampling.restore_image(s)` takes.  Data only; shared by the full-size GPU tests."""
import copy

UNET = dict(image_size=256, num_channels=256, num_res_blocks=2, channel_mult="", learn_sigma=True,
            class_cond=False, use_checkpoint=False, attention_resolutions="32, 16, 8", num_heads=4,
            num_head_channels=64, num_heads_upsample=-1, use_scale_shift_norm=True, dropout=0.0,
            resblock_updown=True, use_fp16=False, use_new_attention_order=False, model_path="",
            pretrain_model="osmosis")
TINY_UNET = dict(image_size=256, num_channels=32, num_res_blocks=1, channel_mult="1,2,2",
                 attention_resolutions="128,64", num_head_channels=16, num_heads=4, learn_sigma=True,
                 use_scale_shift_norm=True, resblock_updown=True, pretrain_model="osmosis")
PATTERN = dict(pattern="pcgs", update_start=0.7, update_end=0, global_N=1, local_M=1, s_start=1, s_end=0,
               n_iter=20, start_guidance=1, stop_guidance=0)


def _diffusion(respacing):
    return dict(sampler="ddpm", steps=1000, noise_schedule="linear", model_mean_type="epsilon",
                model_var_type="learned_range", dynamic_threshold=False, clip_denoised=False,
                rescale_timesteps=False, timestep_respacing=respacing)


# config 2 / 4: osmosis_sample_config.yaml (one underwater image; 64-image set sharded 8 per GPU)
SAMPLE = dict(
    manual_seed=0, degamma_input=False, rgb_guidance=False, sample_pattern=PATTERN, unet_model=UNET,
    diffusion=_diffusion(1000),
    conditioning=dict(method="osmosis", params=dict(
        loss_function="norm", loss_weight="depth", weight_function="gamma,1.4,1.4,1", scale="7,7,7,0.9",
        gradient_x_prev=True, gradient_clip="True,0.005")),
    aux_loss=dict(aux_loss={"avrg_loss": 0.5, "val_loss": 20}),
    measurement=dict(
        operator=dict(name="underwater_physical_revised", optimizer="sgd", depth_type="gamma", value="1.4,1.4,1",
                      phi_a="1.1,0.95,0.95", phi_a_eta="1e-5", phi_a_learn_flag=True,
                      phi_b="0.95, 0.8, 0.8", phi_b_eta="1e-5", phi_b_learn_flag=True,
                      phi_inf="0.14, 0.29, 0.49", phi_inf_eta="1e-5", phi_inf_learn_flag=True),
        noise=dict(name="clean")))

# config 3: osmosis_simulation_sample_config.yaml (simulated NYUv2 with ground truth, run here as one batch of 8)
SIMULATION = dict(
    manual_seed=0, degamma_input=False, rgb_guidance=False, sample_pattern=PATTERN, unet_model=UNET,
    diffusion=_diffusion(1000),
    conditioning=dict(method="osmosis", params=dict(
        loss_function="norm", loss_weight="depth", weight_function="gamma,1.4,1.4,1", scale="4,4,4,1",
        gradient_x_prev=True, gradient_clip="True,0.001")),
    aux_loss=dict(aux_loss={"val_loss": 40}),
    measurement=dict(
        operator=dict(name="underwater_physical", optimizer="sgd", depth_type="original", value="1.4,1.4,1",
                      phi_ab="1.1,0.95,0.95", phi_ab_eta="1e-5", phi_ab_learn_flag=True,
                      phi_inf="0.2,0.4,0.7", phi_inf_eta="1e-5", phi_inf_learn_flag=True),
        noise=dict(name="clean")))

# config 5: osmosis_haze_sample_config.yaml as BASELINE.json quotes it (batch 32, 250-step respacing, fp16)
HAZE = dict(
    manual_seed=0, degamma_input=True, rgb_guidance=False, sample_pattern=PATTERN,
    unet_model=dict(UNET, use_fp16=True), diffusion=_diffusion("250"),
    conditioning=dict(method="osmosis", params=dict(
        loss_function="norm", loss_weight="depth", weight_function="gamma,1.4,1.4,1", scale="7,7,7,0.9",
        gradient_x_prev=True, gradient_clip="True,0.005")),
    aux_loss=dict(aux_loss={"avrg_loss": 0.5, "val_loss": 20}),
    measurement=dict(
        operator=dict(name="haze_physical", optimizer="sgd", depth_type="gamma", value="1.4,1.4,1",
                      phi_ab=1.0, phi_ab_eta="1e-5", phi_ab_learn_flag=True,
                      phi_inf="0.14, 0.29, 0.49", phi_inf_eta="1e-5", phi_inf_learn_flag=True),
        noise=dict(name="clean")))


def with_unet(cfg, unet_kw):
    c = copy.deepcopy(cfg)
    c["unet_model"] = dict(unet_kw)
    return c
