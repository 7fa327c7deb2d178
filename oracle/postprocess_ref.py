"""ORACLE (test infrastructure only -- imported by tests/, never by the product path).

numpy restatement of what the reference does with the sampler's outputs for one image:
  * depth normalisation helpers  osmosis_utils/utils.py:46-74 (min_max_norm_range), :77-114
    (min_max_norm_range_percentile; torch.quantile = linear interpolation between order statistics);
  * convert_depth               osmosis_utils/utils.py:544-566;
  * forward-image recomposition osmosis_sampling.py:199-300 (inline code of the driver, restated as a function):
        rgb01 = (x0[:3]+1)/2, d = convert_depth(x0[3]),  B = phi_inf (1 - exp(-phi_b d)),  A = exp(-phi_a d),
        I = rgb01 A + B,  degraded = 2 I - 1,  norm = round(||degraded - y||_2, 3),  recon = exp(phi_a d) (y01 - B).
Pinned by tests/golden/postprocess.npz (outputs of the reference's own helper functions on seeded inputs).
"""
import numpy as np


def min_max_norm_range(img, vmin=0.0, vmax=1.0, is_uint8=False):
    img = np.asarray(img, dtype=np.float32)
    if img.ndim == 4:
        lo = img.reshape(img.shape[0], -1).min(1).reshape(-1, 1, 1, 1)
        hi = img.reshape(img.shape[0], -1).max(1).reshape(-1, 1, 1, 1)
    elif img.ndim == 3:
        lo, hi = img.min(), img.max()
    else:
        raise NotImplementedError
    if np.all(lo == hi):
        out = np.zeros_like(img)
    else:
        scale = np.float32(float(vmax) - float(vmin)) / (hi - lo)
        out = (img - lo) * scale + np.float32(vmin)
    return (255 * out).astype(np.uint8) if is_uint8 else out


def quantile(img, q):
    """torch.quantile default ('linear') on the flattened tensor, computed in fp32 like torch."""
    v = np.sort(np.asarray(img, dtype=np.float32).ravel())
    pos = np.float32(q) * np.float32(v.size - 1)
    lo = int(np.floor(pos))
    hi = min(lo + 1, v.size - 1)
    w = np.float32(pos - np.float32(lo))
    return np.float32(v[lo] + (v[hi] - v[lo]) * w)


def min_max_norm_range_percentile(img, vmin=0.0, vmax=1.0, percent_low=0.0, percent_high=1.0, is_uint8=False):
    img = np.asarray(img, dtype=np.float32)
    clip = np.clip(img, quantile(img, percent_low), quantile(img, percent_high))
    return min_max_norm_range(clip, vmin, vmax, is_uint8)


def convert_depth(depth, depth_type, value):
    depth = np.asarray(depth, dtype=np.float32)
    if isinstance(value, str):
        value = np.array([float(p) for p in value.split(",")], dtype=float)
    elif value is None:
        raise NotImplementedError
    if depth_type == "move":
        return (depth + value).astype(np.float32)
    if depth_type == "gamma":
        return np.power((depth + np.float32(value[0])) * np.float32(value[1]), np.float32(value[2])).astype(np.float32)
    if depth_type is None or depth_type == "original":
        return np.float32(0.5) * (depth + np.float32(1.0))
    raise NotImplementedError


def recompose(x0, phi, ref, operator_name, depth_type, value):
    """x0 [4,H,W], ref [3,H,W] in [-1,1], phi dict of [3] / [1] arrays -> dict of the driver's derived images."""
    x0 = np.asarray(x0, dtype=np.float32)
    ref = np.asarray(ref, dtype=np.float32)
    rgb01 = np.float32(0.5) * (x0[:3] + 1)
    d = convert_depth(np.repeat(x0[3:4], 3, axis=0), depth_type, value)
    inf = np.asarray(phi["phi_inf"], dtype=np.float32).reshape(-1, 1, 1)
    if "underwater_physical_revised" in operator_name:
        a = np.asarray(phi["phi_a"], dtype=np.float32).reshape(-1, 1, 1)
        b = np.asarray(phi["phi_b"], dtype=np.float32).reshape(-1, 1, 1)
    elif "haze" in operator_name or "underwater_physical" in operator_name:
        a = b = np.asarray(phi["phi_ab"], dtype=np.float32).reshape(-1, 1, 1)
    else:
        raise NotImplementedError
    ones = np.ones_like(rgb01)
    back = (inf * ones) * (1 - np.exp(-(b * ones) * d))
    att = np.exp(-(a * ones) * d)
    fwd = rgb01 * att + back
    degraded = 2 * fwd - 1
    return {"rgb_01": rgb01, "rgb_01_clip": np.clip(rgb01, 0, 1), "depth_calc": d, "backscatter": back,
            "attenuation": att, "forward_predicted": fwd, "degraded": degraded,
            "norm_loss_final": float(np.round(np.linalg.norm((degraded - ref).astype(np.float32).ravel()), 3)),
            "rgb_recon": np.exp((a * ones) * d) * (np.float32(0.5) * (ref + 1) - back)}


def psnr(img, ref, data_range=1.0):
    mse = np.mean((np.asarray(img, dtype=np.float64) - np.asarray(ref, dtype=np.float64)) ** 2)
    return 10.0 * np.log10(data_range ** 2 / mse)
