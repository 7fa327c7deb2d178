"""CPU restatement of the reference's unconditional sampler (TEST INFRASTRUCTURE, see oracle/__init__.py).
Follows osmosis_utils/diffusion.py:26-47 (schedule) and :59-130 (`inverse`, numeric path only)."""
import numpy as np
import torch


def schedule_linear(T):
    beta = np.linspace(1e-4, 2e-2, T)
    alpha = 1 - beta
    return beta, alpha, np.cumprod(alpha)


def inverse(net, x, T, steps, noises, start_t=None, trace=None):
    """x [1,C,H,W]; noises[k] = the randn_like(x) drawn at loop iteration k (t > 1 only).
    Runs t = start_t .. start_t-steps+1 of the T-step chain (truncation, not respacing: SURVEY F11)."""
    beta, alpha, alphabar = schedule_linear(T)
    C = x.shape[1]
    start_t = T if start_t is None else start_t
    k = 0
    x0 = None
    for t in range(start_t, start_t - steps, -1):
        if trace is not None:
            trace.append(x.clone())
        at, atbar = alpha[t - 1], alphabar[t - 1]
        if t > 1:
            z = noises[k]
            k += 1
            beta_tilde = beta[t - 1] * (1 - alphabar[t - 2]) / (1 - atbar)
        else:
            z = torch.zeros_like(x)
            beta_tilde = 0
        with torch.no_grad():
            pred = net(x, torch.tensor([t]).float())[:, :C]
        x0 = (1 / np.sqrt(atbar)) * (x - (np.sqrt(1 - atbar) * pred))
        x = (1 / np.sqrt(at)) * (x - ((1 - at) / np.sqrt(1 - atbar)) * pred) + np.sqrt(beta_tilde) * z
    return x, x0
