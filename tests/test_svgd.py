"""SVGD / AMSGrad restatement (parity unpinned: blackjax / optax sources are not in the reference
tree) validated against analytic answers instead.  CPU only."""

import math

import numpy as np
import torch

from phlash_amd import svgd

F64 = torch.float64


def test_functional_gradient_matches_definition():
    rng = np.random.default_rng(0)
    x = torch.tensor(rng.normal(size=(7, 3)))
    g = torch.tensor(rng.normal(size=(7, 3)))
    h = 0.7
    phi = svgd.functional_gradient(x, g, h)
    # brute force: phi(x_j) = mean_i [ -k(x_i,x_j) g_i - d k(x_i,x_j)/d x_i ] with autograd for dk
    want = torch.zeros_like(x)
    for j in range(7):
        acc = torch.zeros(3, dtype=F64)
        for i in range(7):
            xi = x[i].clone().requires_grad_(True)
            k = torch.exp(-((xi - x[j]) ** 2).sum() / h)
            (dk,) = torch.autograd.grad(k, xi)
            acc += -k.detach() * g[i] - dk
        want[j] = acc / 7
    np.testing.assert_allclose(phi, want, rtol=1e-12, atol=1e-14)


def test_median_heuristic():
    x = torch.tensor([[0.0], [1.0], [3.0]], dtype=F64)  # pairwise distances 1, 2, 3 -> median 2
    np.testing.assert_allclose(svgd.median_heuristic(x), 4.0 / math.log(3))
    x = torch.tensor([[0.0], [1.0], [3.0], [7.0]], dtype=F64)  # 1,2,3,4,6,7 -> median 3.5
    np.testing.assert_allclose(svgd.median_heuristic(x), 3.5**2 / math.log(4))


def test_amsgrad_first_steps():
    st = svgd.init(torch.zeros(1, 2, dtype=F64))
    g = torch.tensor([[1.0, -2.0]], dtype=F64)
    upd, mu, nu, nu_max, count = svgd.amsgrad_update(st, g, lr=0.1)
    # first step of a bias-corrected Adam-family update is -lr * sign(g) (up to eps)
    np.testing.assert_allclose(upd, [[-0.1, 0.1]], rtol=1e-6)
    st = svgd.SVGDState(st.particles, st.length_scale, mu, nu, nu_max, count)
    upd2, *_ = svgd.amsgrad_update(st, 0.1 * g, lr=0.1)
    # second moment max is kept: the step cannot grow when the gradient shrinks
    assert float(upd2.abs().max()) < 0.1


def test_svgd_recovers_gaussian_posterior():
    """Particles driven by the score of N(mu, diag(s^2)) must end up with that mean and roughly
    that spread."""
    torch.manual_seed(0)
    mu = torch.tensor([1.5, -2.0], dtype=F64)
    s = torch.tensor([0.5, 2.0], dtype=F64)
    x = torch.randn(100, 2, dtype=F64) * 3.0
    st = svgd.init(x)
    for _ in range(1500):
        score = -(st.particles - mu) / s**2
        st = svgd.step(st, score, lr=0.05)
    m = st.particles.mean(0)
    sd = st.particles.std(0)
    np.testing.assert_allclose(m, mu, atol=0.1)
    np.testing.assert_allclose(sd, s, rtol=0.25)
