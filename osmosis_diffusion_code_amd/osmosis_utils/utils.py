"""Host-side helpers of the hot path (mirror of the reference's osmosis_utils/utils.py subset:
str2bool :384-395, get_depth_value :529-541, convert_depth :544-566, is_freeze_phi :571-590,
set_alternate_length :595-630, set_loss_weight :674-700).  Same names, arguments and error behaviour."""
import argparse

import numpy as np
import torch

DEPTH_TYPE_CODE = {None: 0, "original": 0, "gamma": 1, "move": 2}


def str2bool(v):
    if isinstance(v, bool):
        return v
    s = v.lower()
    if s in ("yes", "true", "t", "y", "1"):
        return True
    if s in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("boolean value expected")


def get_depth_value(value_raw, **kwargs):
    if isinstance(value_raw, float):
        return value_raw
    if isinstance(value_raw, int):
        return float(value_raw)
    if isinstance(value_raw, str):
        return np.array([float(p) for p in value_raw.split(",")], dtype=float)
    if isinstance(value_raw, (np.ndarray, np.generic)):
        return value_raw
    raise NotImplementedError


def depth_code_and_values(depth_type, value):
    """(code, (v0,v1,v2)) as the kernels want them; raises like convert_depth for unknown types."""
    if depth_type not in DEPTH_TYPE_CODE:
        raise NotImplementedError
    code = DEPTH_TYPE_CODE[depth_type]
    v = get_depth_value(value) if value is not None else 0.0
    vals = [0.0, 1.0, 1.0]
    if code == 1:
        vals = [float(v[0]), float(v[1]), float(v[2])]
    elif code == 2:
        vals = [float(v if np.isscalar(v) else np.asarray(v).ravel()[0]), 1.0, 1.0]
    return code, vals


def convert_depth(depth, depth_type, **kwargs):
    """Tensor version (host/visualisation use; the sampler's hot path evaluates this inside the
    physics kernels)."""
    value = get_depth_value(kwargs.get("value", None))   # value=None raises NotImplementedError (utils.py:551-552)
    if depth_type == "move":
        return depth + value
    if depth_type == "gamma":
        return torch.pow((depth + value[0]) * value[1], value[2])
    if depth_type is None or depth_type == "original":
        return 0.5 * (depth + 1.0)
    raise NotImplementedError


def parse_weight_function(weight_function):
    """'gamma,1.4,1.4,1' -> ('gamma', array([1.4,1.4,1.]))"""
    if not isinstance(weight_function, str):
        return "none", None
    parts = weight_function.split(",")
    value = None
    if len(parts) > 1:
        value = np.asarray(parts[1:]).astype(float)
        value = value.item() if value.shape[0] == 1 else value
    return parts[0], value


def set_loss_weight(loss_weight_type, weight_function=None, degraded_image=None, x_0_hat=None):
    fn, value = parse_weight_function(weight_function)
    if loss_weight_type == "none" or loss_weight_type is None:
        return 1
    if loss_weight_type == "depth":
        return convert_depth(depth=x_0_hat.detach()[:, 3:4], depth_type=fn, value=value)
    raise NotImplementedError


def _outside_guidance(p, idx, T):
    return idx > p["start_guidance"] * T or idx < p["stop_guidance"] * T


def _outside_update(p, idx, T):
    return idx > p["update_start"] * T or idx < p["update_end"] * T


def is_freeze_phi(sample_pattern, time_index, num_timesteps):
    if sample_pattern is None or sample_pattern["pattern"] == "original":
        return False
    return bool(_outside_guidance(sample_pattern, time_index, num_timesteps)
                or _outside_update(sample_pattern, time_index, num_timesteps))


def set_alternate_length(sample_pattern, time_index, num_timesteps):
    if sample_pattern["pattern"] != "original" and sample_pattern is not None:
        assert sample_pattern["update_start"] > sample_pattern["update_end"]
        assert sample_pattern["s_start"] > sample_pattern["s_end"]
        if sample_pattern["local_M"] > 1:
            assert sample_pattern["update_start"] >= sample_pattern["s_start"]
            assert sample_pattern["s_end"] >= sample_pattern["update_end"]
    if sample_pattern is None or sample_pattern["pattern"] == "original":
        return 1
    T = num_timesteps
    if _outside_guidance(sample_pattern, time_index, T) or _outside_update(sample_pattern, time_index, T):
        return 1
    if time_index > sample_pattern["s_start"] * T or time_index < sample_pattern["s_end"] * T:
        return 1
    return sample_pattern["local_M"]


# ----------------------------------------------------------------------------- output post-processing (N2)
# Host-side (CPU tensors, like the reference: the driver works on `pred_xstart.cpu()`).
def _min_max(img):
    """(min, max) per image for [B,C,H,W], global for [C,H,W] (utils.py:46-62)."""
    if img.dim() == 4:
        flat = img.reshape(img.size(0), -1)
        return flat.min(dim=1)[0].view(-1, 1, 1, 1), flat.max(dim=1)[0].view(-1, 1, 1, 1)
    if img.dim() == 3:
        return img.min(), img.max()
    raise NotImplementedError


def _rescale(img, vmin, vmax, is_uint8):
    lo, hi = _min_max(img)
    if lo == hi:            # a batch (numel > 1) raises here exactly like the reference's `if img_min == img_max`
        out = torch.zeros_like(img)
    else:
        out = (img - lo) * ((float(vmax) - float(vmin)) / (hi - lo)) + float(vmin)
    if is_uint8:
        out = (255 * out).to(torch.uint8)
    return out


def min_max_norm_range(img, vmin=0, vmax=1, is_uint8=False):
    """utils.py:46-74: affine map of [min, max] onto [vmin, vmax]; a constant image maps to zeros."""
    return _rescale(img, vmin, vmax, is_uint8)


def min_max_norm_range_percentile(img, vmin=0, vmax=1, percent_low=0., percent_high=1., is_uint8=False):
    """utils.py:77-114: clamp to the [percent_low, percent_high] quantiles first, then min-max normalise."""
    lo, hi = torch.quantile(img, q=percent_low), torch.quantile(img, q=percent_high)
    return _rescale(torch.clamp(img, lo, hi), vmin, vmax, is_uint8)


def depth_tensor_to_color_image(tensor_image, colormap="viridis"):
    """utils.py:748-763: [H,W] / [1,H,W] / [1,1,H,W] depth in [0,1] -> [3,H,W] colour-mapped (float64 like matplotlib)."""
    import matplotlib
    cm = matplotlib.colormaps[colormap] if hasattr(matplotlib, "colormaps") else __import__("matplotlib.pyplot").pyplot.get_cmap(colormap)
    if tensor_image.dim() == 4:
        tensor_image = tensor_image.squeeze()
    if tensor_image.dim() == 3:
        tensor_image = tensor_image[0]
    assert tensor_image.dim() == 2
    rgba = cm(tensor_image.numpy())
    return torch.tensor(rgba[:, :, 0:3]).permute(2, 0, 1)


def psnr(img, ref, data_range=1.0):
    """Peak signal-to-noise ratio in dB per image ([B,C,H,W] -> [B], [C,H,W] -> scalar tensor).  The reference
    computes no quality metric; SURVEY.md 8(f) N2 asks for one for the simulated config."""
    d = (img.double() - ref.double()) ** 2
    mse = d.flatten(1).mean(dim=1) if d.dim() == 4 else d.mean()
    return 10.0 * torch.log10((float(data_range) ** 2) / mse)


# ----------------------------------------------------------------------------- driver-side helpers the sampling scripts call
def clip_image(img, scale=True, move=True, is_uint8=True):
    """utils.py:138-159: [C,H,W] (or [H,W]) tensor, optionally (img + 1) / 2, to uint8 by clamp-and-truncate or to [0, 1].  The
    reference multiplies IN PLACE when is_uint8 (a caller's tensor is scaled by 255 unless move / scale made a copy first); this
    one never touches its argument."""
    if img.dim() == 2:
        img = img.unsqueeze(0)
    if move:
        img = img + 1
    if scale:
        img = 0.5 * img
    if is_uint8:
        return (img * 255).clamp(0, 255).to(torch.uint8)
    return img.clamp(0, 1)


def load_yaml(file_path: str) -> dict:
    """utils.py:357-360 (yaml.FullLoader: '1e-5' and '32, 16, 8' stay strings)."""
    import yaml
    with open(file_path) as f:
        return yaml.load(f, Loader=yaml.FullLoader)


def arguments_from_file(config_file_path: str) -> argparse.Namespace:
    """utils.py:466-476: the YAML file's top-level keys as attributes of a Namespace (nested sections stay dictionaries) -- what
    osmosis_sampling.py / RGBD_prior_sampling.py read as `args.<section>[...]`.  `sampling.load_config` is the dictionary form."""
    args = argparse.Namespace()
    for k, v in load_yaml(config_file_path).items():
        setattr(args, k, v)
    return args


def change_input_output_unet(model, in_channels=4, out_channels=8):
    """utils.py:265-288 for this package's UNetModel: a new stem convolution (in_channels -> model channels) and a new head
    convolution (-> out_channels), freshly initialised the way `nn.Conv2d` initialises (the reference swaps in new nn.Conv2d layers:
    kaiming-uniform weights with a = sqrt(5), uniform bias), every other parameter kept.  `create_model(pretrain_model='osmosis')`
    builds the 4 -> 8 network directly; this function serves callers that construct `UNetModel(in_channels=3, out_channels=6, ...)`
    first, as RGBD_prior_sampling.py:62-69 does.  Returns the model."""
    import math

    from ..guided_diffusion.unet import _Slot
    stem, head = model.input_blocks[0].at(0), model.out.at(2)
    dev = stem.weight.device

    def conv_slot(cout, cin):
        slot = _Slot((cout, cin, 3, 3), (cout,))
        torch.nn.init.kaiming_uniform_(slot.weight, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(cin * 9)
        torch.nn.init.uniform_(slot.bias, -bound, bound)
        return slot.to(dev)
    model.input_blocks[0].add_module("0", conv_slot(stem.weight.shape[0], in_channels))
    model.out.add_module("2", conv_slot(out_channels, head.weight.shape[1]))
    model.in_channels, model.out_channels = in_channels, out_channels
    model._engines, model._weights = {}, {}              # packed weight images and plans of the old shapes
    return model


def get_optimizer(optimizer_name, model_parameters, **kwargs):
    """utils.py:494-524: the torch optimizer of that (case-insensitive) name over `model_parameters` (parameter groups with their own
    `lr`), None for 'GD' / '' (plain gradient descent done by hand), ValueError for an unknown name.  For third-party operators written
    against the reference API; the package's own physical operators step phi inside osm_phys_finalize (measurements.OPTIMIZER_CODES)."""
    name = optimizer_name.lower()
    if name in ("gd", ""):
        return None
    table = {"adam": "Adam", "sgd": "SGD", "rmsprop": "RMSprop", "adagrad": "Adagrad", "adadelta": "Adadelta", "adamw": "AdamW",
             "sparseadam": "SparseAdam", "adamax": "Adamax", "asgd": "ASGD", "lbfgs": "LBFGS", "rprop": "Rprop"}
    if name not in table:
        raise ValueError(f"Optimizer '{name}' is not supported.")
    return getattr(torch.optim, table[name])(model_parameters, **kwargs)
