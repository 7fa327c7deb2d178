"""Auxiliary-loss registry (mirror of the reference's osmosis_utils/losses.py:9-83).

On the hot path these are not evaluated as modules: PosteriorSamplingOsmosis reads the
coefficients (`AuxiliaryLoss.losses_dictionary`) and the fused physics kernels compute
avrg_loss / val_loss and their gradients.  The torch `forward`s are kept for API parity
(per-image semantics are identical at the reference's batch size 1)."""
import torch
import torch.nn as nn

__LOSS__ = {}


def register_loss(name: str):
    def wrapper(cls):
        if __LOSS__.get(name, None):
            raise NameError(f"Name {name} is already registered!")
        __LOSS__[name] = cls
        return cls
    return wrapper


def get_loss(name: str, **kwargs):
    if __LOSS__.get(name, None) is None:
        raise NameError(f"Name {name} is not defined.")
    return __LOSS__[name](**kwargs)


@register_loss(name="avrg_loss")
class Average_Loss(nn.Module):
    kernel_slot = "gamma_avrg"

    def forward(self, x):
        return torch.sum(torch.abs(torch.mean(x[:, 0:3], dim=(2, 3))))


@register_loss(name="val_loss")
class Value_Loss(nn.Module):
    kernel_slot = "gamma_val"

    def __init__(self, device=None, **kwargs):
        super().__init__()

    def forward(self, rgbd, **kwargs):
        thr = kwargs.get("value", 0.7)
        rgb = rgbd[:, 0:3]
        return (torch.clamp(rgb.abs() - thr, min=0) ** 2).mean()


class AuxiliaryLoss(nn.Module):
    def __init__(self, losses_dictionary):
        super().__init__()
        self.losses_dictionary = losses_dictionary
        self.losses_list = [get_loss(k) for k in losses_dictionary.keys()]
        self.loss_gammas = [torch.tensor(v) for v in losses_dictionary.values()]

    def kernel_coefficients(self):
        """{'gamma_avrg': g, 'gamma_val': g} for the fused physics kernels."""
        out = {"gamma_avrg": 0.0, "gamma_val": 0.0}
        for mod, g in zip(self.losses_list, self.loss_gammas):
            slot = getattr(mod, "kernel_slot", None)
            if slot is None:
                raise NotImplementedError(f"auxiliary loss {type(mod).__name__} has no HIP implementation")
            out[slot] += float(g)
        return out

    def forward(self, x):
        total, parts = 0, {}
        for g, mod, name in zip(self.loss_gammas, self.losses_list, self.losses_dictionary):
            cur = mod.forward(x)
            total = total + g.to(x.device) * cur
            parts[name] = cur.detach().cpu()
        return total, parts
