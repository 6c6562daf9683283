"""Oracle for the sampler's update: SVGD with an RBF kernel and the median heuristic, fed to AMSGrad.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  **Parity unpinned**: the reference delegates this step to
``blackjax.svgd(grad(log_density), optax.amsgrad(learning_rate))`` (src/phlash/mcmc.py:178-199, 279; blackjax==1.2.5,
optax==0.2.6 in uv.lock:163-164, 1797-1798).  Neither package is in the reference tree or installable here, and the
only reference test that reaches them checks types and lengths (tests/test_mcmc.py:10-32).  This file restates the
two published algorithms from their definitions, with explicit loops over particles and coordinates, and shares no
code with ``phlash_amd/svgd.py`` or ``csrc/svgd_step.hip`` -- it is what those two are compared against.

blackjax 1.2.5, ``blackjax/vi/svgd.py``:
  * state = (particles, kernel_parameters = {"length_scale": 1.0} at init, optimiser state);
  * ``rbf_kernel(x, y, length_scale) = exp(-(1/length_scale) * sum((x - y)**2))``;
  * step: for every particle p_j:  phi*(p_j) = mean over i of [ -(k(p_i, p_j) * grad_logp(p_i)) - d k(p_i, p_j)/d p_i ],
    handed to ``optimizer.update`` as if it were a loss gradient, ``particles = optax.apply_updates(particles, updates)``;
  * after the step ``update_median_heuristic``: length_scale = median(d_ij, i > j)**2 / log(n) with d the Euclidean
    distances between the NEW particles (``jnp.median``: mean of the two middle order statistics for an even count).
optax 0.2.6, ``scale_by_amsgrad`` then ``scale(-learning_rate)`` (b1 = 0.9, b2 = 0.999, eps = 1e-8, eps_root = 0):
  mu = b1 mu + (1-b1) g;  nu = b2 nu + (1-b2) g^2;  count += 1;  mu_hat = mu / (1 - b1^count);  nu_hat = nu / (1 - b2^count);
  nu_max = max(nu_max, nu_hat);  update = -lr * mu_hat / (sqrt(nu_max + eps_root) + eps).
"""

from __future__ import annotations

import math

import numpy as np


class State:
    def __init__(self, particles):
        self.particles = np.array(particles, dtype=np.float64)
        self.length_scale = 1.0
        self.mu = np.zeros_like(self.particles)
        self.nu = np.zeros_like(self.particles)
        self.nu_max = np.zeros_like(self.particles)
        self.count = 0


def rbf(x, y, h):
    s = 0.0
    for a, b in zip(x, y):
        s += (a - b) ** 2
    return math.exp(-(1.0 / h) * s)


def phi_star(particles, grads, h):
    """[n, D]: the functional gradient blackjax hands to the optimiser (note its sign: minus the SVGD direction)."""
    n, D = particles.shape
    out = np.zeros((n, D))
    for j in range(n):
        acc = np.zeros(D)
        for i in range(n):
            k = rbf(particles[i], particles[j], h)
            for d in range(D):
                dk_dxi = -(2.0 / h) * (particles[i, d] - particles[j, d]) * k  # d k(x_i, x_j) / d x_i[d]
                acc[d] += -(k * grads[i, d]) - dk_dxi
        out[j] = acc / n
    return out


def median_heuristic(particles):
    n = particles.shape[0]
    if n < 2:
        return 1.0  # (log 1 = 0: the reference never runs a single particle; the product keeps 1)
    d = []
    for i in range(n):
        for j in range(i):
            d.append(math.sqrt(sum((particles[i, q] - particles[j, q]) ** 2 for q in range(particles.shape[1]))))
    d.sort()
    m = len(d)
    med = d[m // 2] if m % 2 else 0.5 * (d[m // 2 - 1] + d[m // 2])
    return med * med / math.log(n)


def step(state: State, grad_logp, lr: float, b1=0.9, b2=0.999, eps=1e-8) -> State:
    g = phi_star(state.particles, np.asarray(grad_logp, float), state.length_scale)
    new = State(state.particles)
    new.count = state.count + 1
    n, D = g.shape
    for i in range(n):
        for d in range(D):
            mu = b1 * state.mu[i, d] + (1.0 - b1) * g[i, d]
            nu = b2 * state.nu[i, d] + (1.0 - b2) * g[i, d] * g[i, d]
            mu_hat = mu / (1.0 - b1**new.count)
            nu_hat = nu / (1.0 - b2**new.count)
            nmax = max(state.nu_max[i, d], nu_hat)
            new.mu[i, d], new.nu[i, d], new.nu_max[i, d] = mu, nu, nmax
            new.particles[i, d] = state.particles[i, d] - lr * mu_hat / (math.sqrt(nmax) + eps)
    new.length_scale = median_heuristic(new.particles)
    return new
