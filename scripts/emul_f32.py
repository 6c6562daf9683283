"""numpy emulation of the HIP kernels' arithmetic in a chosen dtype (dev tool: separates
algorithmic precision limits from kernel bugs)."""
import numpy as np


def run(P, row, W, dt=np.float32):
    K = P.shape[1]
    b, d, u, v, e0, e1, pi = (P[i].astype(dt) for i in range(7))
    T = len(row)
    a = pi.copy()
    alphas = [a.copy()]
    scales = []
    E = 0
    llW = 0.0
    invW = dt(0)
    one = np.ones(K, dt)
    cs = dt(1)
    for t in range(T):
        ux = (u * a).astype(dt)
        pre = np.concatenate([[dt(0)], np.cumsum(ux[:-1], dtype=dt)]).astype(dt)
        suf = np.concatenate([np.cumsum(a[::-1][:-1], dtype=dt)[::-1], [dt(0)]]).astype(dt)
        p = (d * a + v * pre + b * suf).astype(dt)
        ob = row[t]
        e = one if ob < 0 else (e1 if ob >= 1 else e0)
        p = (p * e).astype(dt)
        c = p.sum(dtype=dt)
        ex = int(np.frexp(c)[1])
        s = dt(2.0) ** dt(-ex)
        a = (p * s).astype(dt)
        cs = dt(c * s)
        E += ex
        scales.append(s)
        alphas.append(a.copy())
        if t + 1 == W:
            llW = np.log(float(cs)) + E * np.log(2.0)
            invW = dt(1.0 / float(cs))
    ll = np.log(float(cs)) + E * np.log(2.0) - llW
    beta = np.full(K, dt(1.0 / float(cs)), dt)
    g = np.zeros((7, K), np.float64)
    for t in range(T, 0, -1):
        if t == W:
            beta = (beta - invW).astype(dt)
        ob = row[t - 1]
        ap, aq = alphas[t - 1], alphas[t]
        s = scales[t - 1]
        m = (aq * beta).astype(dt)
        if ob == 0:
            g[4] += m
        elif ob >= 1:
            g[5] += m
        e = one if ob < 0 else (e1 if ob >= 1 else e0)
        w = (beta * s * e).astype(dt)
        ux = (u * ap).astype(dt)
        pre = np.concatenate([[dt(0)], np.cumsum(ux[:-1], dtype=dt)]).astype(dt)
        suf = np.concatenate([np.cumsum(ap[::-1][:-1], dtype=dt)[::-1], [dt(0)]]).astype(dt)
        vw = (v * w).astype(dt)
        bw = (b * w).astype(dt)
        svw = np.concatenate([np.cumsum(vw[::-1][:-1], dtype=dt)[::-1], [dt(0)]]).astype(dt)
        pbw = np.concatenate([[dt(0)], np.cumsum(bw[:-1], dtype=dt)]).astype(dt)
        g[0] += (w * suf).astype(dt)
        g[1] += (w * ap).astype(dt)
        g[3] += (w * pre).astype(dt)
        g[2] += (ap * svw).astype(dt)
        beta = (d * w + pbw + u * svw).astype(dt)
    g[4] /= e0
    g[5] /= e1
    g[6] = beta
    return ll, g


if __name__ == "__main__":
    import sys
    sys.path.insert(0, ".")
    sys.path.insert(0, "tests")
    from oracle import cport
    from test_hip_parity import _params
    rng = np.random.default_rng(0)
    data = (rng.uniform(size=(6, 700)) < 0.08).astype(np.int8)
    data.flat[rng.integers(0, data.size, 40)] = -1
    P = _params(4, 2, 1, seed=3)
    inds = np.array([5, 0, 3, 3])
    np.set_printoptions(linewidth=200, precision=3)
    for W in (0, 64):
        ll_ref, g_ref = cport.batch(P, data, inds, W)
        for bb in range(2):
            for si, s in enumerate(inds[:3]):
                ll, g = run(P[bb, 0], data[s], W)
                ll64, g64 = run(P[bb, 0], data[s], W, np.float64)
                sc = np.abs(g_ref[bb, si]).max(-1, keepdims=True)
                print(W, bb, s, "ll err f32", abs(ll - ll_ref[bb, si]) / abs(ll_ref[bb, si]), "f64", abs(ll64 - ll_ref[bb, si]),
                      "g err f32", (np.abs(g - g_ref[bb, si]) / sc).max(), "f64", (np.abs(g64 - g_ref[bb, si]) / sc).max())
                if (np.abs(g - g_ref[bb, si]) / sc).max() > 1e-3:
                    print((np.abs(g - g_ref[bb, si]) / sc))
                    print(g_ref[bb, si]); print(P[bb, 0])
