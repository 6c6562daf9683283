// Particle -> PSMCParams on the device, with its Jacobian, in ONE launch.
//
// What is computed (reference: jthlab/phlash v1.0.6, paths relative to that repo):
//   src/phlash/params.py:94-127   MCMCParams.to_dm      x = [t_tr(2), c_tr(P), rho_over_theta_tr]
//                                  -> t = [0, geomspace(t1, tM, K-1)], c = pattern(softplus(c_tr)),
//                                     rho = (0.1 + 9.9 sigmoid(.)) theta
//   src/phlash/size_history.py:17-22,123-138,170-193    _expm1inv, surv, pi, ect
//   src/phlash/transition.py:9-85                        _expQ (incl. its u<1e-6 branch, quirk Q8),
//                                                        the running 3x3 products, L / D / U
//   src/phlash/params.py:33-55    PSMCParams.from_dm     clips, (b, d, u, v) factorisation
// The reference runs this as ~10^3 XLA ops per particle and differentiates it with jax.grad; the
// torch restatement in phlash_amd/{params,transition,size_history}.py is ~1,400 kernel launches per
// SVGD step.  Here one workgroup per particle evaluates the map in float64 with forward-mode dual
// numbers: thread j carries d/dx_j, every thread carries the value, so the block writes the [7,K]
// parameter block and its [7K, D] Jacobian (D = P + 3 <= 66) with coalesced stores.  The VJP the
// sampler needs is then a single batched mat-vec (phlash_amd/param_map.py).
//
// Only row 0 of the running 3x3 products is carried (the reference reads nothing else,
// transition.py:58-71) and the upper triangle is assembled from running products in O(K)
// (transition.py:76-83 builds it with an O(K^3) masked power).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "step_args.h"

namespace phk {

struct Dual {
    double v, d;
};
__device__ __forceinline__ Dual mk(double v, double d = 0.0) { return Dual{v, d}; }
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { return mk(a.v + b.v, a.d + b.d); }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { return mk(a.v - b.v, a.d - b.d); }
__device__ __forceinline__ Dual operator-(Dual a) { return mk(-a.v, -a.d); }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) { return mk(a.v * b.v, a.d * b.v + a.v * b.d); }
__device__ __forceinline__ Dual operator/(Dual a, Dual b) {
    const double q = a.v / b.v;
    return mk(q, (a.d - q * b.d) / b.v);
}
__device__ __forceinline__ Dual operator+(Dual a, double b) { return mk(a.v + b, a.d); }
__device__ __forceinline__ Dual operator-(Dual a, double b) { return mk(a.v - b, a.d); }
__device__ __forceinline__ Dual operator*(Dual a, double b) { return mk(a.v * b, a.d * b); }
__device__ __forceinline__ Dual operator*(double b, Dual a) { return mk(a.v * b, a.d * b); }
__device__ __forceinline__ Dual operator/(Dual a, double b) { return mk(a.v / b, a.d / b); }
__device__ __forceinline__ Dual operator/(double a, Dual b) {
    const double q = a / b.v;
    return mk(q, -q * b.d / b.v);
}
__device__ __forceinline__ Dual operator-(double a, Dual b) { return mk(a - b.v, -b.d); }
__device__ __forceinline__ Dual operator+(double a, Dual b) { return mk(a + b.v, b.d); }
__device__ __forceinline__ Dual dexp(Dual a) {
    const double e = exp(a.v);
    return mk(e, e * a.d);
}
__device__ __forceinline__ Dual dexpm1(Dual a) { return mk(expm1(a.v), exp(a.v) * a.d); }
__device__ __forceinline__ Dual dlog(Dual a) { return mk(log(a.v), a.d / a.v); }
__device__ __forceinline__ Dual dsqrt(Dual a) {
    const double s = sqrt(a.v);
    return mk(s, a.d / (2.0 * s));
}
// clip with the sub-gradient torch / jax use: 1 inside [lo, hi], 0 outside
__device__ __forceinline__ Dual dclamp(Dual a, double lo, double hi) {
    if (a.v < lo) return mk(lo);
    if (a.v > hi) return mk(hi);
    return a;
}
__device__ __forceinline__ Dual dsoftplus(Dual a) {
    // log(1 + e^x), threshold 20 as torch.nn.functional.softplus
    if (a.v > 20.0) return a;
    const double e = exp(a.v);
    return mk(log1p(e), a.d * e / (1.0 + e));
}
__device__ __forceinline__ Dual dsigmoid(Dual a) {
    const double s = 1.0 / (1.0 + exp(-a.v));
    return mk(s, a.d * s * (1.0 - s));
}
// 1 / expm1(x), large-x safe  (size_history.py:17-22)
__device__ __forceinline__ Dual dexpm1inv(Dual x) {
    if (x.v > 10.0) return -dexp(-x) / dexpm1(-x);
    return 1.0 / dexpm1(x);
}
__device__ __forceinline__ bool is_close0(double a) { return fabs(a) <= 1e-8; }  // isclose(a, 0): atol 1e-8

struct Row3 {
    Dual r0, r1, r2;
};

// row <- row @ expQ(r, c, n = 2)   (transition.py:9-34; only the row-vector product is needed)
__device__ __forceinline__ Row3 step_expQ(Row3 row, Dual r, Dual c) {
    const double n = 2.0;
    const Dual cn = c * n;
    const Dual u = dsqrt(cn * cn - 2.0 * c * (n - 2.0) * r + r * r) / 2.0;
    const Dual v = (r + cn) / 2.0;
    const Dual w = (r - cn) / 2.0;
    const Dual t1 = (dexp(u - v) + dexp(-(u + v))) / 2.0;
    Dual t2;
    if (u.v < 1e-6) {
        // quirk Q8: the reference's series branch is evaluated at u_safe = 1 (transition.py:18-21)
        t2 = dexp(-v) * (1.0 + 1.0 / 6.0);
    } else {
        t2 = (dexp(u - v) - dexp(-(u + v))) / 2.0 / u;
    }
    const Dual P11 = t1 - w * t2, P12 = r * t2, P21 = c * t2, P22 = t1 + w * t2;
    const Dual P13 = 1.0 - P11 - P12, P23 = 1.0 - P21 - P22;
    Row3 o;
    o.r0 = row.r0 * P11 + row.r1 * P21;
    o.r1 = row.r0 * P12 + row.r1 * P22;
    o.r2 = row.r0 * P13 + row.r1 * P23 + row.r2;
    return o;
}

// One pass over the hidden states with O(1) state per thread: round 1's version kept t, c, ect, the transition
// factors and the survival differences in per-thread arrays of dual numbers indexed by a run-time k (9 KB of
// scratch per thread), and spent most of its 130 us per launch on scratch traffic.  Same operations in the
// same order, so the same bits.
__global__ __launch_bounds__(128) void param_map_kernel(PMArgs A) {
    const int K = A.K, D = A.D;
    const int64_t bidx = blockIdx.x;
    const int j = threadIdx.x;  // tangent carried by this thread (threads >= D carry a zero tangent)
    const double* x = A.x + bidx * D;
    auto X = [&](int i) { return mk(x[i], i == j ? 1.0 : 0.0); };
    const double lo = 1e-20, hi = 1.0 - 1e-20;
    double* out = A.params + bidx * 7 * K;
    double* jac = A.jac ? A.jac + bidx * 7 * K * D : nullptr;
    auto put = [&](int row, int k, Dual val) {
        if (j == 0) {
            out[row * K + k] = val.v;
            if (A.params_f32) A.params_f32[bidx * 7 * K + row * K + k] = (float)val.v;
        }
        if (jac && j < D) jac[(size_t)(row * K + k) * D + j] = val.d;
    };

    // ---- MCMCParams.to_dm (params.py:94-127): t = [0, geomspace(t1, tM, K-1)], c by epoch, rho -------------
    const Dual t1 = dexp(X(0));
    const Dual tM = t1 + dexp(X(1));
    const Dual lt1 = dlog(t1), ltM = dlog(tM);
    auto t_of = [&](int k) { return k == 0 ? mk(0.0) : dexp(lt1 + (ltM - lt1) * ((double)(k - 1) / (double)(K - 2))); };
    const Dual rho = (0.1 + 9.9 * dsigmoid(X(2 + A.P))) * A.theta;

    Dual tk = mk(0.0);                                  // t[k]
    Dual H = mk(0.0), Sprev = mk(1.0), csum = mk(0.0);  // survival (size_history.py:123-138)
    Row3 row{mk(1.0), mk(0.0), mk(0.0)};                // row 0 of the running 3x3 products (transition.py:52)
    Dual Rt2_prev = mk(0.0);                            // ... its last entry at t[k]
    Dual p1_first = mk(0.0), p1_prev = mk(0.0);         // p1[0], p1[k-1]
    Dual cum = mk(1.0), A01 = mk(1.0);                  // prod_{0<l<k} p2[l], A[0][1]
    put(3, 0, mk(0.0));
    for (int k = 0; k < K; ++k) {
        const bool last = k == K - 1;
        const Dual tk1 = last ? tk : t_of(k + 1);  // t[k+1]
        const Dual ck = dsoftplus(X(2 + A.epoch[k]));
        // ---- SizeHistory.ect (size_history.py:170-193) --------------------------------------
        Dual e;
        if (!last) {
            const Dual dt = tk1 - tk;
            if (is_close0(ck.v)) e = (tk + tk1) / 2.0;
            else if (isinf(ck.v) || ck.v > 100.0) e = tk;
            else e = 1.0 / ck + tk - dt * dexpm1inv(ck * dt);
        } else {
            e = tk + 1.0 / ck;
        }
        if (e.v < 1e-20) e = mk(1e-20);
        // ---- emissions (params.py:36-43) ------------------------------------------------------
        {
            const Dual uu = e * A.theta;
            put(4, k, dclamp(dexp(-uu), lo, hi));
            put(5, k, dclamp(-dexpm1(-uu), lo, hi));
        }
        // ---- pi (size_history.py:123-138): pi[k] = surv[k-1] - surv[k] for k >= 1 ----------------
        if (!last) {
            H = H + ck * (tk1 - tk);
            const Dual S = dexp(-H);
            if (k > 0) {
                const Dual Ci = Sprev - S;
                csum = csum + Ci;
                put(6, k, dclamp(Ci, lo, hi));
            }
            Sprev = S;
        } else {
            csum = csum + Sprev;  // surv[K-2] - 0
            put(6, k, dclamp(Sprev, lo, hi));
        }
        // ---- transition factors (transition.py:37-83) -----------------------------------------
        // augmented grid [t0, e0, t1, e1, ...]: t_k -> ect_k with rate c_k
        {
            const Dual dt = e - tk;
            if (!is_close0(dt.v)) row = step_expQ(row, 2.0 * dt * rho, dt * ck);
        }
        const Row3 Re = row;
        const Dual c_adj = ck;  // c * (n - 1), n = 2
        Dual q, keep, p2k, p3k;
        if (!last) {
            const Dual gap = (tk1 - e) * c_adj;
            q = -dexpm1(-gap);
            keep = dexp(-gap);
            const Dual dtk = tk1 - tk;
            p2k = dclamp(dexp(-dtk * c_adj), 1e-8, 1.0 - 1e-8);
            p3k = dclamp(-dexpm1(-dtk * c_adj), 1e-8, 1.0 - 1e-8);
        } else {
            q = mk(1.0);
            keep = mk(0.0);
            p2k = mk(1e-8);
            p3k = mk(1.0 - 1e-8);
        }
        put(1, k, dclamp(Re.r0 + Re.r1 * q + Re.r2 - Rt2_prev, lo, hi));  // diagonal, transition.py:60-67
        const Dual p1k = dclamp(Re.r1 * keep, 1e-8, 1.0 - 1e-8);
        // ect_k -> t_{k+1} with rate c_k (the last interval ends in the absorbing Pinf)
        if (!last) {
            const Dual dt = tk1 - e;
            if (!is_close0(dt.v)) row = step_expQ(row, 2.0 * dt * rho, dt * ck);
            put(0, k, dclamp(row.r2 - Rt2_prev, lo, hi));  // sub-diagonal, transition.py:58
            Rt2_prev = row.r2;
        } else {
            put(0, k, mk(0.0));
        }
        // ---- (u, v) of params.py:44-55: first row above the diagonal A[0][k] = p1[0] prod_{0<l<k} p2[l] p3[k]
        if (k == 0) {
            p1_first = p1k;
        } else {
            const Dual A0k = dclamp(p1_first * cum * p3k, lo, hi);
            if (k == 1) A01 = A0k;
            const Dual vk = A0k / A01;
            put(3, k, vk);
            // u[k-1] = A[k-1][k] / v[k],  A[i][i+1] = p1[i] * p3[i+1]
            put(2, k - 1, dclamp(p1_prev * p3k, lo, hi) / vk);
            cum = cum * p2k;
        }
        p1_prev = p1k;
        tk = tk1;
    }
    put(2, K - 1, mk(0.0));
    put(6, 0, dclamp(1.0 - csum, lo, hi));
}

// log_prior of a whole population with its gradient (model.py:11-21): one thread per particle.
//   value = logN(log(rho/theta); 0, 1) - alpha sum_i (log c_{i+1} - log c_i)^2 - beta |x|^2,
//   rho/theta = 0.1 + 9.9 sigmoid(x[P+2]) (params.py:110-112), c = softplus(x[2 .. 2+P)) per epoch.
__device__ double log_prior_one(int P, double alpha, double beta, const double* __restrict__ x, double* __restrict__ g) {
    const int D = P + 3;
    double xx = 0.0;
    for (int i = 0; i < D; ++i) xx += x[i] * x[i];
    const double r = x[P + 2];
    const double sg = 1.0 / (1.0 + exp(-r));
    const double rot = 0.1 + 9.9 * sg;
    const double z = log(rot);
    double ret = -0.5 * z * z - 0.91893853320467274178;  // 0.5 log(2 pi)
    // softplus as torch evaluates it (threshold 20): y > 20 ? y : log1p(exp(y)), derivative 1 : sigmoid(y)
    auto lc_of = [&](int i, double& dlc) {
        const double y = x[2 + i];
        const double sp = y > 20.0 ? y : log1p(exp(y));
        const double dsp = y > 20.0 ? 1.0 : 1.0 / (1.0 + exp(-y));
        dlc = dsp / sp;
        return log(sp);
    };
    double dprev = 0.0, dl_prev = 0.0, lc_prev = 0.0, rough = 0.0;
    for (int i = 0; i < P; ++i) {
        double dl;
        const double lc = lc_of(i, dl);
        const double dcur = i > 0 ? lc - lc_prev : 0.0;  // lc_i - lc_{i-1}
        if (i > 0) {
            rough += dcur * dcur;
            // d/d lc_{i-1} of -alpha sum diff^2 = -2 alpha (d_{i-1} - d_i)
            if (g) g[2 + i - 1] = -2.0 * alpha * (dprev - dcur) * dl_prev - 2.0 * beta * x[2 + i - 1];
        }
        dprev = dcur;
        dl_prev = dl;
        lc_prev = lc;
    }
    if (g) {
        g[2 + P - 1] = -2.0 * alpha * dprev * dl_prev - 2.0 * beta * x[2 + P - 1];
        g[0] = -2.0 * beta * x[0];
        g[1] = -2.0 * beta * x[1];
        g[P + 2] = -z * 9.9 * sg * (1.0 - sg) / rot - 2.0 * beta * r;
    }
    return ret - alpha * rough - beta * xx;
}

__global__ void log_prior_kernel(int P, double alpha, double beta, const double* __restrict__ X, int64_t B,
                                 double* __restrict__ value, double* __restrict__ grad) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int D = P + 3;
    value[b] = log_prior_one(P, alpha, beta, X + b * D, grad ? grad + b * D : nullptr);
}

// ---------------------------------------------------------------------------------------------
// The tail of a sampler step between the likelihood kernels and the SVGD update, in two launches (it was ~25 torch
// launches of 5-20 us: sums over the chunk axis, buffer assembly, the flag hand-over, prior, the autograd chain):
//   reduce_chunks : ll [B, S] f64 and d ll / d params [B, S, J = 7K] (float or double)  ->  buf [B + 1, 1 + J] f64,
//                   row b = [sum_s ll, sum_s grad], row B = the kernel object's flags (underflow, bad index; the
//                   device word is cleared) -- the buffer that is all-reduced over the ranks (parallel.py);
//   chain_rule    : buf (reduced), the Jacobian d params / d x [B, J, D] of phk_param_map and the particles x
//                   [B, D]  ->  logp [B] = c_prior log_prior(x) + c_hmm sum ll (+ c_extra extra_val) and its gradient
//                   [B, D] = c_prior d log_prior + c_hmm J^T sum grad (+ c_extra extra_grad); a particle whose
//                   logp is not finite gets -inf and a zero gradient (model.py:73 under jax.grad).
// Every sum has a fixed order (bit-reproducible).
// ---------------------------------------------------------------------------------------------
constexpr int RC_NT = 1024;

template <typename real>
__global__ __launch_bounds__(RC_NT) void reduce_chunks_kernel(const double* __restrict__ ll, const real* __restrict__ g,
                                                              int64_t B, int64_t S, int J, double* __restrict__ buf,
                                                              int* flags) {
    __shared__ double red[RC_NT];
    __shared__ int fw;
    const int64_t b = blockIdx.x;
    const int t = threadIdx.x;
    double* row = buf + b * (1 + J);
    if (b == B) {  // the flag row
        if (t == 0) fw = flags ? atomicExch(flags, 0) : 0;
        __syncthreads();
        const int w = fw;
        // (bit 2 -- a loop out of its iteration budget, psmc_kernels.hip FLAG_OVERRUN -- rides in the second slot with the weight 4096)
        for (int i = t; i < 1 + J; i += RC_NT)
            row[i] = i == 0 ? ((w & 1) ? 1.0 : 0.0) : (i == 1 ? ((w & 2) ? 1.0 : 0.0) + ((w & 4) ? 4096.0 : 0.0) : 0.0);
        return;
    }
    // gradient: lane j of part p adds chunks p, p + parts, ...; the parts are then added in part order
    const int parts = RC_NT / J;
    const int part = t / J, j = t - part * J;
    double acc = 0.0;
    if (part < parts) {
        const real* src = g + (b * S) * (int64_t)J + j;
        for (int64_t s = part; s < S; s += parts) acc += (double)src[s * J];
        red[t] = acc;
    }
    __syncthreads();
    if (t < J) {
        double tot = 0.0;
        for (int p = 0; p < parts; ++p) tot += red[p * J + t];
        row[1 + t] = tot;
    }
    __syncthreads();
    // log-likelihood: strided partial sums, then a fixed tree
    double l = 0.0;
    for (int64_t s = t; s < S; s += RC_NT) l += ll[b * S + s];
    red[t] = l;
    __syncthreads();
    for (int off = RC_NT / 2; off > 0; off >>= 1) {
        if (t < off) red[t] += red[t + off];
        __syncthreads();
    }
    if (t == 0) row[0] = red[0];
}

hipError_t launch_reduce_chunks(const double* ll, const void* g, bool g_f64, int64_t B, int64_t S, int J, double* buf, int* flags,
                                hipStream_t st) {
    if (g_f64)
        hipLaunchKernelGGL(reduce_chunks_kernel<double>, dim3((unsigned)(B + 1)), dim3(RC_NT), 0, st, ll, (const double*)g, B, S, J, buf, flags);
    else
        hipLaunchKernelGGL(reduce_chunks_kernel<float>, dim3((unsigned)(B + 1)), dim3(RC_NT), 0, st, ll, (const float*)g, B, S, J, buf, flags);
    return hipGetLastError();
}

constexpr int CR_MAXJ = 7 * PM_MAXK, CR_MAXD = PM_MAXK + 3;

__global__ __launch_bounds__(128) void chain_rule_kernel(CRArgs A) {
    __shared__ double G[CR_MAXJ];
    __shared__ double part[8][CR_MAXD];  // partial sums of J^T G over eight slices of j, added in slice order
    __shared__ double pg[CR_MAXD];
    __shared__ double pv;
    const int64_t b = blockIdx.x;
    const int t = threadIdx.x, J = A.J, D = A.D;
    const double* row = A.buf + b * (1 + J);
    for (int j = t; j < J; j += 128) G[j] = row[1 + j];
    // log_prior (log_prior_one above, same arithmetic in the same order, so the same bits) with the per-epoch
    // transcendentals spread over the lanes: one lane doing all of them in turn was 30 of this kernel's 48 us
    {
        __shared__ double lcs[PM_MAXK], dls[PM_MAXK];
        const int P = A.P;
        const double* x = A.x + b * D;
        if (t < P) {
            const double y = x[2 + t];
            const double sp = y > 20.0 ? y : log1p(exp(y));
            const double dsp = y > 20.0 ? 1.0 : 1.0 / (1.0 + exp(-y));
            dls[t] = dsp / sp;
            lcs[t] = log(sp);
        }
        __syncthreads();
        if (t < P) {  // d/d lc_t of -alpha sum_i (lc_i - lc_{i-1})^2 = -2 alpha (d_t - d_{t+1}), d_0 = d_P = 0
            const double dt = t > 0 ? lcs[t] - lcs[t - 1] : 0.0;
            const double dn = t + 1 < P ? lcs[t + 1] - lcs[t] : 0.0;
            pg[2 + t] = (t + 1 < P ? -2.0 * A.alpha * (dt - dn) : -2.0 * A.alpha * dt) * dls[t] - 2.0 * A.beta * x[2 + t];
        }
        if (t == 127) {
            double xx = 0.0;
            for (int i = 0; i < D; ++i) xx += x[i] * x[i];
            const double r = x[P + 2];
            const double sg = 1.0 / (1.0 + exp(-r));
            const double rot = 0.1 + 9.9 * sg;
            const double z = log(rot);
            double rough = 0.0;
            for (int i = 1; i < P; ++i) {
                const double dcur = lcs[i] - lcs[i - 1];
                rough += dcur * dcur;
            }
            pg[0] = -2.0 * A.beta * x[0];
            pg[1] = -2.0 * A.beta * x[1];
            pg[P + 2] = -z * 9.9 * sg * (1.0 - sg) / rot - 2.0 * A.beta * r;
            pv = (-0.5 * z * z - 0.91893853320467274178) - A.alpha * rough - A.beta * xx;
        }
    }
    __syncthreads();
    // thread (slice, d): consecutive d read consecutive addresses of a Jacobian row; a slice takes a contiguous range of
    // rows j, and the slices are added in slice order: the order of the sum is fixed (bit-reproducible)
    const int lanes = D <= 16 ? 16 : (D <= 32 ? 32 : 64);
    const int slices = 128 / lanes;  // 8, 4 or 2 (D <= 67: at least 2... D > 64 falls back to one slice of 128 lanes)
    const int sl = t / lanes, d = t - sl * lanes;
    if (D <= 64) {
        const int per = (J + slices - 1) / slices;
        const int j0 = sl * per, j1 = j0 + per < J ? j0 + per : J;
        double acc = 0.0;
        if (d < D) {
            const double* jc = A.jac + b * (int64_t)J * D + d;
            for (int j = j0; j < j1; ++j) acc = fma(G[j], jc[(int64_t)j * D], acc);
            part[sl][d] = acc;
        }
    } else if (t < D) {
        double acc = 0.0;
        const double* jc = A.jac + b * (int64_t)J * D + t;
        for (int j = 0; j < J; ++j) acc = fma(G[j], jc[(int64_t)j * D], acc);
        part[0][t] = acc;
    }
    __syncthreads();
    double lp = A.c_prior * pv + A.c_hmm * row[0];
    if (A.extra_val) lp += A.c_extra * A.extra_val[b];
    const bool fin = isfinite(lp);
    if (t < D) {
        double acc = part[0][t];
        if (D <= 64)
            for (int q = 1; q < slices; ++q) acc += part[q][t];
        double gx = A.c_prior * pg[t] + A.c_hmm * acc;
        if (A.extra_grad) gx += A.c_extra * A.extra_grad[b * D + t];
        A.grad[b * D + t] = fin ? gx : 0.0;
    }
    if (t == 0) A.logp[b] = fin ? lp : -INFINITY;
}

hipError_t launch_chain_rule(const CRArgs& a, int64_t B, hipStream_t st) {
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(chain_rule_kernel, dim3((unsigned)B), dim3(128), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_log_prior(int P, double alpha, double beta, const double* x, int64_t B, double* value, double* grad,
                            hipStream_t st) {
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(log_prior_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, st, P, alpha, beta, x, B, value, grad);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// AFS term of the objective (model.py:58-68): sum_m xlogy((T afs)_m, (T esfs)_m), esfs = etbl / sum(etbl),
// etbl = W etjj (size_history.py:224-226), etjj_k = int_0^inf exp(-k(k-1)/2 R(t)) dt for k = 2..n as the closed form
// over the size history's pieces (size_history.py:217-222 via JaxPPoly.exp_integral, jax_ppoly.py:44-84).  It depends on
// the particle only through t = [0, geomspace(t1, tM, K-1)] and c = softplus(c_tr) by epoch (params.py:94-127).
// One workgroup per particle; thread j carries the tangent d / d x_j through the whole evaluation (forward-mode duals,
// as param_map_kernel): ~(n - 1)(K + m) dual operations per thread -- microseconds, against the ~60 launches of the
// autograd graph this replaces in the fused step (3 ms at n = 20, round 4).
// T W, 1^T W and T afs are constants of a run, prepared once on the host.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void afs_term_kernel(AFArgs A) {
    const int K = A.K, D = A.D, n1 = A.n1;
    const int64_t bidx = blockIdx.x;
    const int j = threadIdx.x;
    const double* x = A.x + bidx * D;
    auto X = [&](int i) { return mk(x[i], i == j ? 1.0 : 0.0); };
    const Dual t1 = dexp(X(0));
    const Dual tM = t1 + dexp(X(1));
    const Dual lt1 = dlog(t1), ltM = dlog(tM);
    auto t_of = [&](int k) { return k == 0 ? mk(0.0) : dexp(lt1 + (ltM - lt1) * ((double)(k - 1) / (double)(K - 2))); };
    // numerator (T W etjj)_m and denominator 1^T W etjj accumulated k by k: etjj_k is used once
    Dual num[AF_MAXN];
    for (int m = 0; m < A.m; ++m) num[m] = mk(0.0);
    Dual den = mk(0.0);
    for (int kk = 0; kk < n1; ++kk) {
        const double kap = 0.5 * (double)(kk + 2) * (double)(kk + 1);  // k (k - 1) / 2, k = kk + 2
        Dual I = mk(0.0), tk = mk(0.0), e = mk(0.0);
        for (int s = 0; s < K; ++s) {
            const Dual a = dsoftplus(X(2 + A.epoch[s])) * kap;
            if (s < K - 1) {
                const Dual tn = t_of(s + 1);
                const Dual xdt = a * (tn - tk);
                e = e + dexp(-I) * (-dexpm1(-xdt)) / a;
                I = I + xdt;
                tk = tn;
            } else {
                e = e + dexp(-I) / a;
            }
        }
        den = den + e * A.w1[kk];
        for (int m = 0; m < A.m; ++m) num[m] = num[m] + e * A.tw[(size_t)m * n1 + kk];
    }
    Dual val = mk(0.0);
    const Dual lden = dlog(den);
    for (int m = 0; m < A.m; ++m) {
        const double ym = A.y[m];
        if (ym != 0.0) val = val + (dlog(num[m]) - lden) * ym;  // xlogy(0, .) = 0
    }
    if (j == 0) A.value[bidx] = val.v;
    if (j < D) A.grad[bidx * D + j] = val.d;
}

hipError_t launch_afs_term(const AFArgs& a, hipStream_t st) {
    if (a.B <= 0) return hipSuccess;
    hipLaunchKernelGGL(afs_term_kernel, dim3((unsigned)a.B), dim3(128), 0, st, a);
    return hipGetLastError();
}

// One thread per (block, state): the float32 rounding of the seven rows, and the float32 kernels' pre-folded factors.
// The float32 kernels run on the model written with hom emission 1 (psmc_kernels.hip, "Folded model"): (b, d, v) <- emis0 .* (b, d, v)
// and emission rows 1, emis1 / emis0, 1 / emis0.  Formed inside the kernels from the float32-rounded rows, every folded factor
// carries three roundings (the factor, emis0, their product) that are the same at every site of a row; formed here in float64
// and rounded once it carries one -- the rounding any float32 parameter has.
// crel (optional, [n, 7, K] float64): what the rounding to float32 did to the MODEL, as coefficients of a first-order correction of
// the log-likelihood.  Whatever the kernels compute, they compute for the model with factors q_eff = fl(q) instead of q, and to first
// order  ll(q) - ll(q_eff) = sum_q (d ll / d log q) (q - q_eff) / q.  The factors of the folded model are b' = emis0 b, d', v', u,
// the ratio rows emis1 / emis0 and 1 / emis0, and pi; their log-derivatives are linear in the rows the gradient kernel returns:
// b d ll/d b etc., mass(het) = emis1 d ll/d emis1, mass(hom) = emis0 d ll/d emis0, and mass(missing) = b g_b + d g_d + v g_v -
// mass(hom) - mass(het) (every site's posterior mass is the sum of the three products, state by state).  Collected per caller row r:
//     ll(q) - ll(q_eff)  =  sum_{r, k} theta_{r,k} (d ll / d theta_{r,k}) crel_{r,k}  +  O(eps^2 L)
// (phk_ll_first_order applies it).  The error that is left is the arithmetic's, which does not add up coherently along a row.
__global__ void prefold_kernel(int K, const double* __restrict__ params, int64_t n, float* __restrict__ p32, float* __restrict__ pf,
                               double* __restrict__ crel) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * K) return;
    const int64_t q = idx / K;
    const int k = (int)(idx - q * K);
    const double* p = params + q * 7 * K;
    float* o = p32 + q * 7 * K;
#pragma unroll
    for (int r = 0; r < 7; ++r) o[r * K + k] = (float)p[r * K + k];
    const double e0 = p[4 * K + k], e1 = p[5 * K + k];
    float* f = pf + q * 5 * K;
    const double fb = e0 * p[0 * K + k], fd = e0 * p[1 * K + k], fv = e0 * p[3 * K + k];
    // (a block whose float32 emis0 is too small to fold is not folded by the kernels, whatever stands here: Lane::emissions_foldable)
    const double rh = e0 > 0.0 ? e1 / e0 : 1.0, rm = e0 > 0.0 ? 1.0 / e0 : 1.0;
    f[0 * K + k] = (float)fb;
    f[1 * K + k] = (float)fd;
    f[2 * K + k] = (float)fv;
    f[3 * K + k] = (float)rh;
    f[4 * K + k] = (float)rm;
    if (crel == nullptr) return;
    // relative residual of one rounding (0 for a zero, which float32 keeps exactly)
    auto res = [](double x) { return x != 0.0 ? (x - (double)(float)x) / x : 0.0; };
    bool folds = true;  // the kernels' own test (Lane::emissions_foldable): every float32 emis0 of the block above 2^-64
    for (int j = 0; j < K; ++j) folds = folds && (float)p[4 * K + j] > 0x1p-64f;
    double* c = crel + q * 7 * K;
    if (folds) {
        const double em = res(rm);
        c[0 * K + k] = res(fb) + em;
        c[1 * K + k] = res(fd) + em;
        c[2 * K + k] = res(p[2 * K + k]);
        c[3 * K + k] = res(fv) + em;
        c[4 * K + k] = -em;
        c[5 * K + k] = res(rh) - em;
        c[6 * K + k] = res(p[6 * K + k]);
    } else {
#pragma unroll
        for (int r = 0; r < 7; ++r) c[r * K + k] = res(p[r * K + k]);
    }
}

hipError_t launch_prefold(int K, const double* params, int64_t n, float* params_f32, float* prefold_f32, double* crel, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const int64_t total = n * K;
    hipLaunchKernelGGL(prefold_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, K, params, n, params_f32, prefold_f32, crel);
    return hipGetLastError();
}

// ll[b, s] += sum_j theta[b(,s), j] g[b, s, j] crel[b(,s), j]   (g = d ll / d theta, or theta d ll / d theta if dlog).  Sixteen lanes per
// sequence, each a strided share of the 7K terms (coalesced reads of the gradient row), then a butterfly over the sixteen: one thread
// per sequence walked its 112 terms one dependent fma and three uncoalesced loads at a time, 40 us at any batch size.
__global__ __launch_bounds__(256) void ll_first_order_kernel(double* __restrict__ ll, const float* __restrict__ g, const double* __restrict__ params,
                                                             const double* __restrict__ crel, int64_t stride_b, int64_t stride_s, int64_t B, int64_t S, int J,
                                                             int dlog) {
    const int64_t idx = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int r = threadIdx.x & 15;
    double acc = 0.0;
    if (idx < B * S) {
        const int64_t b = idx / S, s = idx - b * S;
        const float* gr = g + idx * J;
        const double* th = params + b * stride_b + s * stride_s;
        const double* cr = crel + b * stride_b + s * stride_s;
        for (int j = r; j < J; j += 16) {
            const double c = cr[j];
            if (c != 0.0) acc = fma((double)gr[j] * (dlog ? 1.0 : th[j]), c, acc);
        }
    }
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 16);  // (every lane of the wave takes part: idle sequences carry 0)
    if (r == 0 && idx < B * S) ll[idx] += acc;
}

hipError_t launch_ll_first_order(double* ll, const float* g, const double* params, const double* crel, int64_t stride_b, int64_t stride_s,
                                 int64_t B, int64_t S, int K, int dlog, hipStream_t st) {
    const int64_t n = B * S;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(ll_first_order_kernel, dim3((unsigned)((n * 16 + 255) / 256)), dim3(256), 0, st, ll, g, params, crel, stride_b, stride_s, B, S,
                       7 * K, dlog);
    return hipGetLastError();
}

hipError_t launch_param_map(const PMArgs& a, hipStream_t st) {
    if (a.B <= 0) return hipSuccess;
    hipLaunchKernelGGL(param_map_kernel, dim3((unsigned)a.B), dim3(a.D <= 64 ? 64 : 128), 0, st, a);
    return hipGetLastError();
}

}  // namespace phk
