"""Measurement operators and noisers -- registry surface of the reference's
guided_diffusion/measurements.py (register_operator/get_operator :19-38, register_noise/get_noise
:444-459) with the learnable physical operators re-designed for the fused HIP guidance kernels.

Reference semantics kept:
  * `get_operator(name, device=, batch_size=, **cfg)` builds a fresh operator (phi re-initialised
    per image) and sets `operator.__name__ = name`; unknown / duplicate names raise NameError.
  * phi values arrive as strings ("1.1,0.95,0.95"), etas as strings or floats ("1e-5"),
    `phi_*_learn_flag=False` freezes a variable (eta 0)            (measurements.py:213-249)
  * plain SGD `phi <- phi - eta * dL/dphi` (`optimizer: sgd`, or the 'GD' branch: same math), or any elementwise torch.optim
    class the reference's factory knows (adam, adamw, adamax, rmsprop, adagrad, adadelta, asgd, rprop: torch defaults, lr = eta
    per parameter group, state per operator instance = per image), all stepped on device inside osm_phys_finalize
    (measurements.py:266-303).  'sparseadam' / 'lbfgs' raise (they cannot step these parameters in the reference either).

Device state: `phi` is ONE fp32 device tensor [B][9] = phi_a[3] | phi_b[3] | phi_inf[3]
(`underwater_physical` and `haze_physical` keep phi_ab in the phi_a slots; haze's scalar is
replicated).  It is read and updated in place by osm_phys_* kernels (csrc/guidance.hip); the
Python object never syncs with the device unless the caller asks for values.
"""
from abc import ABC, abstractmethod

import numpy as np
import torch

from .._lib import PhysDesc
from ..osmosis_utils import utils as utilso

__OPERATOR__ = {}


def register_operator(name: str):
    def wrapper(cls):
        if __OPERATOR__.get(name, None):
            raise NameError(f"Name {name} is already registered!")
        __OPERATOR__[name] = cls
        return cls
    return wrapper


def get_operator(name: str, **kwargs):
    if __OPERATOR__.get(name, None) is None:
        raise NameError(f"Name {name} is not defined.")
    operator = __OPERATOR__[name](**kwargs)
    operator.__name__ = name
    return operator


class LinearOperator(ABC):
    @abstractmethod
    def forward(self, data, **kwargs):
        pass

    @abstractmethod
    def transpose(self, data, **kwargs):
        pass

    def ortho_project(self, data, **kwargs):
        return data - self.transpose(self.forward(data, **kwargs), **kwargs)

    def project(self, data, measurement, **kwargs):
        return self.ortho_project(measurement, **kwargs) - self.forward(data, **kwargs)


class _IdentityOperator(LinearOperator):
    def __init__(self, device, batch_size=1, **kwargs):
        self.device = device
        self.batch_size = batch_size

    def forward(self, data, **kwargs):
        return data

    def transpose(self, data, **kwargs):
        return data

    def ortho_project(self, data, **kwargs):
        return data

    def project(self, data, **kwargs):
        return data


@register_operator(name="noise")
class DenoiseOperator(_IdentityOperator):
    pass


@register_operator(name="rgb_guidance")
class RGBGuidanceOperator(_IdentityOperator):
    pass


class LearnableOperator(ABC):
    @abstractmethod
    def forward(self, data, **kwargs):
        pass


def _vec(s, n=3):
    a = np.array([float(p) for p in str(s).split(",")], dtype=np.float32)
    if a.size == 1 and n == 3:
        a = np.repeat(a, 3)
    if a.size != n:
        raise ValueError(f"expected {n} comma separated values, got {s!r}")
    return a


# optimizer name (utils.py:494-524) -> osm_phys_desc.optimizer
OPTIMIZER_CODES = {"": 0, "gd": 0, "sgd": 0, "adam": 1, "adamw": 2, "adamax": 3, "rmsprop": 4, "adagrad": 5, "adadelta": 6,
                   "asgd": 7, "rprop": 8}


def _check_optimizer(name):
    n = (name or "").lower()
    if n in OPTIMIZER_CODES:
        return n
    if n in ("sparseadam", "lbfgs"):
        # the reference builds these, and its optimize() then raises: SparseAdam refuses dense gradients, LBFGS.step() needs a
        # closure (measurements.py:296-297 calls step() without one)
        raise NotImplementedError(f"optimizer '{name}' cannot step the phi parameters (in the reference either: SparseAdam needs "
                                  f"sparse gradients, LBFGS a closure)")
    raise ValueError(f"Optimizer '{name}' is not supported.")


class _PhysicalOperator(LearnableOperator):
    """Shared machinery of the three image-formation models
    I = 0.5(rgb+1) exp(-phi_a d) + phi_inf (1 - exp(-phi_b d)),  d = convert_depth(x[:,3])."""
    KIND = -1
    VARS = ()

    def __init__(self, device, batch_size=1, **kwargs):
        self.device = torch.device(device) if not isinstance(device, torch.device) else device
        self.batch_size = batch_size
        self.depth_type = kwargs.get("depth_type", None)
        self.value = utilso.get_depth_value(kwargs.get("value", None)) if kwargs.get("value", None) is not None else None
        self.depth_code, self.depth_vals = utilso.depth_code_and_values(self.depth_type, kwargs.get("value", None))
        self.optimizer = _check_optimizer(kwargs.get("optimizer", None))
        self._requires_grad = {v: False for v in self.VARS}

    # -- state ------------------------------------------------------------------------------
    def _init_phi(self, a, b, inf):
        row = np.concatenate([a, b, inf]).astype(np.float32)
        self.phi = torch.from_numpy(np.tile(row, (self.batch_size, 1))).to(self.device).contiguous()
        # The reference's parameter tensors (`self.phi_a`, ... [B,3,1,1]; haze `phi_ab` [B,1,1,1]) as VIEWS of the [B][9] block the
        # kernels step: leaves of autograd for code written against the reference API (a third-party conditioning method that calls
        # `operator.forward` under autograd, `loss.backward(inputs=[x_prev] + operator.get_variable_list())`, `operator.optimize()`),
        # and always the current values whichever side moved them.
        self._leaves = {}
        for name, (lo, n) in self._slots().items():
            leaf = self.phi[:, lo:lo + n].unflatten(1, (n, 1, 1))
            self._leaves[name] = leaf
            setattr(self, name, leaf)
        self._torch_optimizer = None

    def _slots(self):
        """{variable name: (first column of self.phi, width)}."""
        raise NotImplementedError

    def eta3(self):
        raise NotImplementedError

    def fill_desc(self, d: PhysDesc):
        d.kind = self.KIND
        d.depth_type = self.depth_code
        for i in range(3):
            d.dval[i] = self.depth_vals[i]
            d.eta[i] = self.eta3()[i]

    def _slot(self, lo, n=3):
        return self.phi[:, lo:lo + n].detach().clone().reshape(self.batch_size, n, 1, 1)

    # -- reference API ----------------------------------------------------------------------
    def forward(self, data, **kwargs):
        """Image formation on torch tensors (measurements.py:138-151, :251-264, :363-376), differentiable w.r.t. `data` AND the
        parameter tensors (`get_variable_list()`), as in the reference.  The sampler's own hot path evaluates the model inside the
        osm_phys_* kernels; this is the API for visualisation and for conditioning methods written against the reference."""
        rgb01 = 0.5 * (data[:, 0:-1] + 1)
        d = utilso.convert_depth(depth=data[:, -1:], depth_type=self.depth_type, value=self.value)
        lv = {k: (v if v.device == data.device else v.to(data.device)) for k, v in self._leaves.items()}
        pa = lv["phi_a"] if "phi_a" in lv else lv["phi_ab"]
        pb = lv["phi_b"] if "phi_b" in lv else pa
        return rgb01 * torch.exp(-pa * d) + lv["phi_inf"] * (1 - torch.exp(-pb * d))

    def optimize(self, **kwargs):
        """measurements.py:266-303.  The package's conditioning method steps phi on the device (osm_phys_finalize) and calls this
        for the variables dictionary only.  A caller that back-propagated into the parameter tensors itself (their `.grad` is set)
        gets the reference's step here: plain gradient descent phi -= eta * grad for 'GD' / 'sgd' / '', else the torch optimizer of
        `optimizer:` over one parameter group per variable (lr = eta), then the gradients are zeroed."""
        if not kwargs.get("freeze_phi", False) and any(v.grad is not None for v in self._leaves.values()):
            etas = dict(zip(("phi_a", "phi_b", "phi_inf"), self.eta3()))
            etas["phi_ab"] = etas["phi_a"]
            if OPTIMIZER_CODES[self.optimizer] == 0:
                with torch.no_grad():
                    for name, v in self._leaves.items():
                        if v.requires_grad and v.grad is not None:
                            v.add_(v.grad, alpha=-etas[name])
            else:
                if self._torch_optimizer is None:
                    self._torch_optimizer = utilso.get_optimizer(self.optimizer, [{"params": v, "lr": etas[n]} for n, v in self._leaves.items()])
                self._torch_optimizer.step()
            for v in self._leaves.values():
                if v.grad is not None:
                    v.grad.zero_()
        return self.variables()

    def set_variable_gradients(self, value=None, **kwargs):
        if value is None:
            raise ValueError("A value should be specified (True or False for general or dictionary)")
        for v in self.VARS:
            self._requires_grad[v] = bool(value[v] if isinstance(value, dict) else value)
            self._leaves[v].requires_grad_(self._requires_grad[v])

    def get_variable_gradients(self, **kwargs):
        return {v: self._leaves[v].requires_grad for v in self.VARS}

    def get_variable_list(self, **kwargs):
        return [self._leaves[v] for v in self.VARS]


@register_operator(name="underwater_physical_revised")
class UnderWaterPhysicalRevisedOperator(_PhysicalOperator):
    KIND = 0
    VARS = ("phi_a", "phi_b", "phi_inf")

    def __init__(self, device, phi_a, phi_b, phi_inf, phi_a_eta=1e-5, phi_b_eta=1e-5, phi_inf_eta=1e-5,
                 phi_a_learn_flag=True, phi_b_learn_flag=True, phi_inf_learn_flag=True, batch_size=1, **kwargs):
        super().__init__(device, batch_size, **kwargs)
        self._init_phi(_vec(phi_a), _vec(phi_b), _vec(phi_inf))
        self.phi_a_eta = float(phi_a_eta) if phi_a_learn_flag else 0.0
        self.phi_b_eta = float(phi_b_eta) if phi_b_learn_flag else 0.0
        self.phi_inf_eta = float(phi_inf_eta) if phi_inf_learn_flag else 0.0

    def eta3(self):
        return (self.phi_a_eta, self.phi_b_eta, self.phi_inf_eta)

    def _slots(self):
        return {"phi_a": (0, 3), "phi_b": (3, 3), "phi_inf": (6, 3)}

    def variables(self):
        return {"phi_a": self._slot(0), "phi_b": self._slot(3), "phi_inf": self._slot(6)}


class _ABOperator(_PhysicalOperator):
    VARS = ("phi_ab", "phi_inf")

    def eta3(self):
        return (self.phi_ab_eta, 0.0, self.phi_inf_eta)


@register_operator(name="underwater_physical")
class UnderWaterPhysicalOperator(_ABOperator):
    KIND = 1

    def __init__(self, device, phi_ab, phi_inf, phi_ab_eta=1e-5, phi_inf_eta=1e-5, phi_ab_learn_flag=True,
                 phi_inf_learn_flag=True, batch_size=1, **kwargs):
        super().__init__(device, batch_size, **kwargs)
        ab = _vec(phi_ab)
        self._init_phi(ab, ab, _vec(phi_inf))
        self.phi_ab_eta = float(phi_ab_eta) if phi_ab_learn_flag else 0.0
        self.phi_inf_eta = float(phi_inf_eta) if phi_inf_learn_flag else 0.0

    def _slots(self):
        return {"phi_ab": (0, 3), "phi_inf": (6, 3)}

    def variables(self):
        return {"phi_ab": self._slot(0), "phi_inf": self._slot(6)}


@register_operator(name="haze_physical")
class HazePhysicalOperator(_ABOperator):
    KIND = 2

    def __init__(self, device, phi_ab, phi_inf, phi_ab_eta=1e-5, phi_inf_eta=1e-5, phi_ab_learn_flag=True,
                 phi_inf_learn_flag=True, batch_size=1, **kwargs):
        super().__init__(device, batch_size, **kwargs)
        ab = np.repeat(np.float32(float(phi_ab)), 3)
        self._init_phi(ab, ab, _vec(phi_inf))
        self.phi_ab_eta = float(phi_ab_eta) if phi_ab_learn_flag else 0.0
        self.phi_inf_eta = float(phi_inf_eta) if phi_inf_learn_flag else 0.0

    def _slots(self):
        return {"phi_ab": (0, 1), "phi_inf": (6, 3)}

    def variables(self):
        return {"phi_ab": self._slot(0, 1), "phi_inf": self._slot(6)}


# ----------------------------------------------------------------------------- noisers
__NOISE__ = {}


def register_noise(name: str):
    def wrapper(cls):
        if __NOISE__.get(name, None):
            raise NameError(f"Name {name} is already defined!")
        __NOISE__[name] = cls
        return cls
    return wrapper


def get_noise(name: str, **kwargs):
    if __NOISE__.get(name, None) is None:
        raise NameError(f"Name {name} is not defined.")
    noiser = __NOISE__[name](**kwargs)
    noiser.__name__ = name
    return noiser


class Noise(ABC):
    def __call__(self, data):
        return self.forward(data)

    @abstractmethod
    def forward(self, data):
        pass


@register_noise(name="clean")
class Clean(Noise):
    def forward(self, data):
        return data


@register_noise(name="gaussian")
class GaussianNoise(Noise):
    def __init__(self, sigma):
        self.sigma = sigma

    def forward(self, data):
        return data + torch.randn_like(data) * self.sigma


@register_noise(name="poisson")
class PoissonNoise(Noise):
    def __init__(self, rate):
        self.rate = rate

    def forward(self, data):
        dev = data.device
        d01 = ((data + 1.0) / 2.0).clamp(0, 1).detach().cpu()
        noisy = torch.from_numpy(np.random.poisson(d01 * 255.0 * self.rate) / 255.0 / self.rate)
        return (noisy * 2.0 - 1.0).clamp(-1, 1).to(dev)
