"""MI355X-native Osmosis sampler hot path (gfx950 HIP kernels behind the reference's registry surface).

Drop-in module layout (same import paths below this package as in the reference repo):
    guided_diffusion.unet.create_model
    guided_diffusion.gaussian_diffusion.create_sampler
    guided_diffusion.measurements.get_operator / get_noise
    guided_diffusion.condition_methods.get_conditioning_method
    osmosis_utils.losses / osmosis_utils.utils (hot-path helpers only)
"""
__version__ = "0.1.0"
