"""Stein variational gradient descent with an AMSGrad update, on torch tensors.

The reference delegates this to third-party code: ``blackjax.svgd(grad(log_density),
optax.amsgrad(lr))`` (src/phlash/mcmc.py:178-199, 279; blackjax==1.2.5, optax==0.2.6 pinned in
uv.lock).  Neither package is in the reference tree, and the only reference test that reaches them
asserts types and lengths (tests/test_mcmc.py:10-32), so this restatement of their published
algorithms is **parity unpinned** (SURVEY.md section 8a row A14): it is validated against analytic
posteriors instead (tests/test_svgd.py).

* kernel: RBF k(x, y) = exp(-|x-y|^2 / h); h starts at 1 and after every step is set to
  median(pairwise distances)^2 / log(n) (blackjax ``update_median_heuristic``);
* functional gradient fed to the optimiser as a loss gradient:
  phi(x_j) = mean_i [ -k(x_i, x_j) grad log p(x_i) - grad_{x_i} k(x_i, x_j) ];
* optax.amsgrad: b1 = 0.9, b2 = 0.999, eps = 1e-8, bias-corrected moments, running max of the
  corrected second moment, update = -lr * m_hat / (sqrt(v_max) + eps).

The functions below are the definition (plain torch, CPU-testable).  On the GPU ``step`` runs the same
update as a handful of HIP launches (``phk_svgd_step``, csrc/svgd_step.hip) instead of ~70 small torch kernels;
``tests/test_kernel_api.py::test_svgd_step_kernel_matches_the_torch_definition`` holds the two together.
"""

from __future__ import annotations

import ctypes
import dataclasses
import math

import torch


@dataclasses.dataclass
class SVGDState:
    particles: torch.Tensor  # [n, D]
    length_scale: torch.Tensor  # 0-dim tensor on the particles' device (no host sync per step)
    mu: torch.Tensor
    nu: torch.Tensor
    nu_max: torch.Tensor
    count: int


def init(particles: torch.Tensor) -> SVGDState:
    z = torch.zeros_like(particles)
    one = torch.ones((), dtype=particles.dtype, device=particles.device)
    return SVGDState(particles=particles, length_scale=one, mu=z, nu=z.clone(), nu_max=z.clone(), count=0)


def functional_gradient(x: torch.Tensor, grad_logp: torch.Tensor, h) -> torch.Tensor:
    """phi [n, D] as defined in the module docstring."""
    n = x.shape[0]
    diff = x[:, None, :] - x[None, :, :]  # [i, j] = x_i - x_j
    k = torch.exp(-(diff**2).sum(-1) / h)  # [i, j]
    grad_k_i = (-2.0 / h) * diff * k[..., None]  # d k(x_i, x_j) / d x_i
    return (-(k.T @ grad_logp) - grad_k_i.sum(0)) / n


def median_heuristic(x: torch.Tensor) -> torch.Tensor:
    """median(pairwise distances)^2 / log(n), as a 0-dim tensor (stays on the device)."""
    n = x.shape[0]
    if n < 2:
        return torch.ones((), dtype=x.dtype, device=x.device)
    d = torch.cdist(x, x, compute_mode="donot_use_mm_for_euclid_dist")  # differences, not |x|^2 + |y|^2 - 2xy
    iu = torch.tril_indices(n, n, offset=-1, device=x.device)
    med = torch.quantile(d[iu[0], iu[1]], 0.5)
    return med**2 / math.log(n)


def amsgrad_update(state: SVGDState, g: torch.Tensor, lr: float, b1=0.9, b2=0.999, eps=1e-8):
    count = state.count + 1
    mu = b1 * state.mu + (1 - b1) * g
    nu = b2 * state.nu + (1 - b2) * g * g
    mu_hat = mu / (1 - b1**count)
    nu_hat = nu / (1 - b2**count)
    nu_max = torch.maximum(state.nu_max, nu_hat)
    upd = -lr * mu_hat / (torch.sqrt(nu_max) + eps)
    return upd, mu, nu, nu_max, count


def step_torch(state: SVGDState, grad_logp: torch.Tensor, lr: float) -> SVGDState:
    """One SVGD / AMSGrad update, the definition (any device)."""
    phi = functional_gradient(state.particles, grad_logp, state.length_scale)
    upd, mu, nu, nu_max, count = amsgrad_update(state, phi, lr)
    x = state.particles + upd
    return SVGDState(particles=x, length_scale=median_heuristic(x), mu=mu, nu=nu, nu_max=nu_max, count=count)


def step_hip(state: SVGDState, grad_logp: torch.Tensor, lr: float, b1=0.9, b2=0.999, eps=1e-8) -> SVGDState:
    """The same update on the GPU through the C ABI (three launches up to 256 particles, seven beyond: the median of
    up to 8.4 million pairwise distances is an exact bucket select over the whole chip; no host synchronisation, no
    sort)."""
    from . import _lib

    x = state.particles.contiguous()
    assert x.is_cuda and x.dtype == torch.float64 and x.ndim == 2
    B, D = x.shape
    g = grad_logp.to(dtype=torch.float64, device=x.device).contiguous()
    mu, nu, nu_max = state.mu.clone(), state.nu.clone(), state.nu_max.clone()
    x_out = torch.empty_like(x)
    h_in = state.length_scale.to(dtype=torch.float64, device=x.device).reshape(1).contiguous()
    h_out = torch.empty(1, dtype=torch.float64, device=x.device)
    lib = _lib.load()
    ws = torch.empty(int(lib.phk_svgd_workspace_doubles(B)), dtype=torch.float64, device=x.device)
    count = state.count + 1
    stream = torch.cuda.current_stream(x.device).cuda_stream
    _lib.check(lib.phk_svgd_step(
        x.device.index, B, D, x.data_ptr(), g.data_ptr(), mu.data_ptr(), nu.data_ptr(), nu_max.data_ptr(),
        h_in.data_ptr(), h_out.data_ptr(), x_out.data_ptr(), ws.data_ptr(), count, float(lr),
        b1, b2, eps, ctypes.c_void_p(stream)))
    return SVGDState(particles=x_out, length_scale=h_out.reshape(()), mu=mu, nu=nu, nu_max=nu_max, count=count)


_MEDIAN_IN_KERNEL = 32768  # pairwise distances (256 particles) up to which ONE workgroup selects the median; beyond: the chip-wide select (csrc/svgd_step.hip)


def step(state: SVGDState, grad_logp: torch.Tensor, lr: float) -> SVGDState:
    """One update: the HIP kernels for particles on the GPU (within their limits: <= 4,096 particles of
    <= 72 coordinates), the torch definition otherwise."""
    x = state.particles
    if x.is_cuda and x.dtype == torch.float64 and x.shape[0] <= 4096 and x.shape[1] <= 72:
        return step_hip(state, grad_logp, lr)
    return step_torch(state, grad_logp, lr)
