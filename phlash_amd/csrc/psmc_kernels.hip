// PSMC / SMC' coalescent-HMM forward recursion and its reverse-mode gradient for gfx950 (MI355X).
//
// What is computed (reference: jthlab/phlash v1.0.6, paths relative to that repo):
//   * src/phlash/hmm.py:52-65 / src/phlash/gpu.py:504-522 -- the O(K) product x*A with the SMC'
//     matrix A[i][j] = b[j] (i>j), d[j] (i==j), u[i]*v[j] (i<j);
//   * src/phlash/hmm.py:68-82 / gpu.py:529-573 -- forward log-likelihood: transition, then
//     emission (code "missing" -> 1), running normalisation;
//   * gpu.py:575-692 -- the gradient with respect to all 7*K parameters.  The reference
//     propagates 7*K forward-mode tangents (O(7K^2) per site); here it is a scaled
//     forward-backward sweep (O(K) per site) with block checkpointing.
//
// Mapping onto CDNA4 (no line of this follows the CUDA kernel's thread layout):
//   * one sequence = one (particle b, chunk s) pair; R adjacent lanes of a 16-lane DPP row own
//     one sequence, each lane holding SPL = K/R consecutive hidden states in registers, so a
//     wave64 carries 64/R sequences.  The two running sums of the mat-vec (prefix of u.*x, suffix
//     of x) are serial inside a lane and cross lanes with row_shr / row_shl DPP steps; the
//     normaliser is a quad_perm / row_mirror butterfly.  No LDS or MFMA on that path.
//   * the 7*K parameters of a sequence live in registers for the whole scan.
//   * observations are re-packed on upload to 2 bits per site (16 sites per dword).
//   * normalisation is by an exact power of two (v_frexp_exp / v_ldexp): the scaled state stays
//     in [0.5,1) and the integer exponents are summed, so ll = E*ln2 + log(sum) takes one log per
//     sequence and loses nothing to f32 accumulation of 60,000 log terms.
//   * gradient: kernel 1 (forward) stores alpha every T sites to HBM (coalesced, K reals per
//     sequence per block); kernel 2 walks the blocks backwards: re-runs the T forward sites of a
//     block into LDS (alpha_{t-1} and the scale per site), then sweeps them in reverse
//     accumulating the six parameter rows in registers (f32 partial sums are flushed to f64 every
//     FLUSH_SITES sites).
//
// Numerics contract: every arithmetic step goes through explicit fma / mul / add in ONE inline
// step function used by both kernels and the file is compiled with -ffp-contract=off, so the
// re-run of a block reproduces the forward pass bit for bit (the backward sweep relies on the
// re-run ending in the same scaling as the next checkpoint).

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace phk {

constexpr int NT_MAX = 256;        // max threads per workgroup (4 waves); the launch picks <= this
constexpr int FLUSH_SITES = 512;   // f32 gradient partial sums are folded into f64 this often

// ---------------------------------------------------------------------------------------------
// scalar helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ int frexp_exp_(float x) { return __builtin_amdgcn_frexp_expf(x); }
__device__ __forceinline__ int frexp_exp_(double x) { return __builtin_amdgcn_frexp_exp(x); }
__device__ __forceinline__ float ldexp_(float x, int e) { return __builtin_ldexpf(x, e); }
__device__ __forceinline__ double ldexp_(double x, int e) { return __builtin_ldexp(x, e); }

// DPP move with zero fill for lanes whose source is outside the 16-lane row.
template <int CTRL>
__device__ __forceinline__ float dpp_(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ double dpp_(double x) {
    const uint64_t u = __builtin_bit_cast(uint64_t, x);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)u, CTRL, 0xf, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(u >> 32), CTRL, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}

constexpr int QP(int a, int b, int c, int d) { return a | (b << 2) | (c << 4) | (d << 6); }
constexpr int ROW_SHL(int n) { return 0x100 + n; }  // lane i <- lane i+n
constexpr int ROW_SHR(int n) { return 0x110 + n; }  // lane i <- lane i-n
constexpr int ROW_MIRROR = 0x140;
constexpr int ROW_HALF_MIRROR = 0x141;

// ---------------------------------------------------------------------------------------------
// R-lane group collectives (R in {1,2,4,8,16}; a group never straddles a 16-lane DPP row)
// ---------------------------------------------------------------------------------------------
template <typename real, int R>
struct Group {
    real up1, up2, up4, up8;  // 1.0 where rank >= n, else 0.0
    real dn1, dn2, dn4, dn8;  // 1.0 where rank + n < R, else 0.0

    __device__ __forceinline__ void init(int rank) {
        up1 = rank >= 1 ? real(1) : real(0);
        up2 = rank >= 2 ? real(1) : real(0);
        up4 = rank >= 4 ? real(1) : real(0);
        up8 = rank >= 8 ? real(1) : real(0);
        dn1 = rank + 1 < R ? real(1) : real(0);
        dn2 = rank + 2 < R ? real(1) : real(0);
        dn4 = rank + 4 < R ? real(1) : real(0);
        dn8 = rank + 8 < R ? real(1) : real(0);
    }

    // all-reduce: every lane of the group gets the same bits (each step adds a commutative pair)
    __device__ __forceinline__ real sum(real x) const {
        if constexpr (R >= 2) x = x + dpp_<QP(1, 0, 3, 2)>(x);
        if constexpr (R >= 4) x = x + dpp_<QP(2, 3, 0, 1)>(x);
        if constexpr (R >= 8) x = x + dpp_<ROW_HALF_MIRROR>(x);
        if constexpr (R >= 16) x = x + dpp_<ROW_MIRROR>(x);
        return x;
    }
    // exclusive prefix over the group: lane r gets sum of x over lanes < r
    __device__ __forceinline__ real excl_prefix(real x) const {
        if constexpr (R == 1) return real(0);
        real y = dpp_<ROW_SHR(1)>(x) * up1;
        if constexpr (R > 2) y = fma_(dpp_<ROW_SHR(1)>(y), up1, y);
        if constexpr (R > 3) y = fma_(dpp_<ROW_SHR(2)>(y), up2, y);
        if constexpr (R > 5) y = fma_(dpp_<ROW_SHR(4)>(y), up4, y);
        if constexpr (R > 9) y = fma_(dpp_<ROW_SHR(8)>(y), up8, y);
        return y;
    }
    // exclusive suffix: lane r gets sum of x over lanes > r
    __device__ __forceinline__ real excl_suffix(real x) const {
        if constexpr (R == 1) return real(0);
        real y = dpp_<ROW_SHL(1)>(x) * dn1;
        if constexpr (R > 2) y = fma_(dpp_<ROW_SHL(1)>(y), dn1, y);
        if constexpr (R > 3) y = fma_(dpp_<ROW_SHL(2)>(y), dn2, y);
        if constexpr (R > 5) y = fma_(dpp_<ROW_SHL(4)>(y), dn4, y);
        if constexpr (R > 9) y = fma_(dpp_<ROW_SHL(8)>(y), dn8, y);
        return y;
    }
};

// ---------------------------------------------------------------------------------------------
// per-lane slice of one sequence's parameters + the forward / backward site steps
// ---------------------------------------------------------------------------------------------
template <typename real, int K, int R>
struct Lane {
    static constexpr int SPL = K / R;
    static_assert(SPL * R == K, "R must divide K");
    real b[SPL], d[SPL], u[SPL], v[SPL], e0[SPL], e1[SPL];
    Group<real, R> g;

    // p: [7,K] rows b,d,u,v,emis0,emis1,pi (gpu.py:189 stacking order)
    __device__ __forceinline__ void load(const real* __restrict__ p, int rank, real (&pi)[SPL]) {
        g.init(rank);
        const real* q = p + rank * SPL;
#pragma unroll
        for (int i = 0; i < SPL; ++i) {
            b[i] = q[0 * K + i];
            d[i] = q[1 * K + i];
            u[i] = q[2 * K + i];
            v[i] = q[3 * K + i];
            e0[i] = q[4 * K + i];
            e1[i] = q[5 * K + i];
            pi[i] = q[6 * K + i];
        }
    }

    // exclusive prefix of u.*x and exclusive suffix of x over the K states of the sequence
    __device__ __forceinline__ void scans(const real (&x)[SPL], real (&pre_ux)[SPL], real (&suf_x)[SPL]) const {
        real tu = real(0);
#pragma unroll
        for (int i = 0; i < SPL; ++i) {
            pre_ux[i] = tu;
            tu = fma_(u[i], x[i], tu);
        }
        real ta = real(0);
#pragma unroll
        for (int i = SPL - 1; i >= 0; --i) {
            suf_x[i] = ta;
            ta = ta + x[i];
        }
        if constexpr (R > 1) {
            const real cu = g.excl_prefix(tu);
            const real ca = g.excl_suffix(ta);
#pragma unroll
            for (int i = 0; i < SPL; ++i) {
                pre_ux[i] = pre_ux[i] + cu;
                suf_x[i] = suf_x[i] + ca;
            }
        }
    }

    // One forward site (hmm.py:74-79): a <- ((a A) .* e_code) * 2^-ex with ex = exponent of the
    // sum, so that sum(a) is in [0.5,1).  Returns ex; csum = the scaled sum.
    // code: 0 hom, 1 het, 2 missing (emission 1; hmm.py:70-71)
    __device__ __forceinline__ int fwd_site(real (&a)[SPL], int code, real& csum) const {
        real pre[SPL], suf[SPL], p[SPL];
        scans(a, pre, suf);
        const bool het = code == 1, miss = code == 2;
        real c = real(0);
#pragma unroll
        for (int i = 0; i < SPL; ++i) {
            real t = d[i] * a[i];
            t = fma_(v[i], pre[i], t);
            t = fma_(b[i], suf[i], t);
            const real e = het ? e1[i] : e0[i];
            const real te = t * e;
            p[i] = miss ? t : te;
            c = c + p[i];
        }
        c = g.sum(c);
        const int ex = frexp_exp_(c);
#pragma unroll
        for (int i = 0; i < SPL; ++i) a[i] = ldexp_(p[i], -ex);
        csum = ldexp_(c, -ex);
        return ex;
    }

    // One backward site.  In: ap = alpha before the site, aq = alpha after it, beta = d ll/d aq,
    // s = the 2^-ex applied at the site.  Out: beta = d ll / d ap; gradient rows accumulated:
    //   gb += w.*suf(ap)   gd += w.*ap   gu += ap.*suf(v.*w)   gv += w.*pre(u.*ap)
    //   g0/g1 += aq.*beta  (divided by emis0/emis1 at the end)         with w = e.*beta*s
    __device__ __forceinline__ void bwd_site(const real (&ap)[SPL], const real (&aq)[SPL], real (&beta)[SPL], int code,
                                             real s, real (&gb)[SPL], real (&gd)[SPL], real (&gu)[SPL],
                                             real (&gv)[SPL], real (&g0)[SPL], real (&g1)[SPL]) const {
        const bool het = code == 1, miss = code == 2;
        const real f1 = het ? real(1) : real(0);
        const real f0 = code == 0 ? real(1) : real(0);
        real pre[SPL], suf[SPL], w[SPL];
        scans(ap, pre, suf);
#pragma unroll
        for (int i = 0; i < SPL; ++i) {
            const real m = aq[i] * beta[i];
            g1[i] = fma_(f1, m, g1[i]);
            g0[i] = fma_(f0, m, g0[i]);
            const real bs = beta[i] * s;
            const real e = het ? e1[i] : e0[i];
            const real be = bs * e;
            w[i] = miss ? bs : be;
        }
        // suffix of v.*w and prefix of b.*w
        real svw[SPL], pbw[SPL];
        real tv = real(0);
#pragma unroll
        for (int i = SPL - 1; i >= 0; --i) {
            svw[i] = tv;
            tv = fma_(v[i], w[i], tv);
        }
        real tb = real(0);
#pragma unroll
        for (int i = 0; i < SPL; ++i) {
            pbw[i] = tb;
            tb = fma_(b[i], w[i], tb);
        }
        if constexpr (R > 1) {
            const real cv = g.excl_suffix(tv);
            const real cb = g.excl_prefix(tb);
#pragma unroll
            for (int i = 0; i < SPL; ++i) {
                svw[i] = svw[i] + cv;
                pbw[i] = pbw[i] + cb;
            }
        }
#pragma unroll
        for (int i = 0; i < SPL; ++i) {
            gb[i] = fma_(w[i], suf[i], gb[i]);
            gd[i] = fma_(w[i], ap[i], gd[i]);
            gv[i] = fma_(w[i], pre[i], gv[i]);
            gu[i] = fma_(ap[i], svw[i], gu[i]);
            real nb = d[i] * w[i];
            nb = nb + pbw[i];
            beta[i] = fma_(u[i], svw[i], nb);
        }
    }
};

// 2-bit observation codes: 16 sites per dword; site t of a row -> bits [2*(t%16), +2) of word t/16
__device__ __forceinline__ uint32_t block_codes(const uint32_t* __restrict__ words, int64_t t0) {
    return words[t0 >> 4] >> (2 * (int)(t0 & 15));
}

struct SeqAux {      // written by the forward kernel, read by the backward kernel
    double inv_end;  // 1 / sum(alpha) after the last site (scaled state)
    double inv_w;    // 1 / sum(alpha) after the W-th site (0 if W == 0)
};

struct KArgs {
    const uint32_t* packed;  // [N, Lw] 2-bit codes
    int64_t Lw;              // dwords per row
    int64_t Ltot;            // sites per row (warm-up + scored)
    int64_t W;               // leading sites that are not scored
    const int64_t* inds;     // [S] row of each chunk
    const void* params;      // [B, S|1, 7, K] real
    int64_t pstride_b;       // element strides of params
    int64_t pstride_s;       // 0: one block per particle, broadcast over chunks
    int64_t B, S;
    double* ll;              // [B*S]
    void* ckpt;              // [nblk, B*S, K] real  (null: forward only)
    SeqAux* aux;             // [B*S]
    void* grad;              // [B*S, 7, K] real
    double* gacc;            // [B*S, 6, K] f64 partial sums (f32 kernels), zeroed before launch
    int grad_dlog;           // 1: return theta * d ll/d theta (what the reference kernel returns)
};

constexpr double LN2 = 0.693147180559945309417232121458;

// ---------------------------------------------------------------------------------------------
// kernel 1: forward pass.  ll per sequence; optionally alpha checkpoints every T sites.
// ---------------------------------------------------------------------------------------------
template <typename real, int K, int R, int T, bool CKPT>
__global__ __launch_bounds__(NT_MAX) void fwd_kernel(KArgs A) {
    constexpr int SPL = K / R;
    static_assert(T <= 16 && 16 % T == 0, "a block's codes must sit in one dword");
    const int64_t nseq = A.B * A.S;
    const int64_t gid = (int64_t)blockIdx.x * (blockDim.x / R) + threadIdx.x / R;
    const bool active = gid < nseq;
    const int64_t seq = active ? gid : nseq - 1;
    const int rank = threadIdx.x & (R - 1);
    const int64_t bb = seq / A.S, ss = seq - bb * A.S;

    Lane<real, K, R> lane;
    real a[SPL];
    lane.load((const real*)A.params + bb * A.pstride_b + ss * A.pstride_s, rank, a);
    const uint32_t* words = A.packed + A.inds[ss] * A.Lw;

    int E = 0;
    real csum = real(1);
    double llW = 0.0, invW = 0.0;
    const int64_t nblk = (A.Ltot + T - 1) / T;
    real* ck = (real*)A.ckpt;
    for (int64_t blk = 0; blk < nblk; ++blk) {
        const int64_t t0 = blk * T;
        if constexpr (CKPT) {
            if (active) {
                real* dst = ck + (blk * nseq + seq) * K + rank * SPL;
#pragma unroll
                for (int i = 0; i < SPL; ++i) dst[i] = a[i];
            }
        }
        uint32_t codes = block_codes(words, t0);
        const int ns = (int)((A.Ltot - t0) < T ? (A.Ltot - t0) : T);
        for (int i = 0; i < ns; ++i) {
            E += lane.fwd_site(a, codes & 3, csum);
            codes >>= 2;
            if (t0 + i + 1 == A.W) {
                llW = log((double)csum) + (double)E * LN2;
                invW = 1.0 / (double)csum;
            }
        }
    }
    if (active && rank == 0) {
        // Ltot == 0: csum = 1, E = 0 -> ll = 0
        A.ll[seq] = log((double)csum) + (double)E * LN2 - llW;
        if constexpr (CKPT) {
            A.aux[seq].inv_end = 1.0 / (double)csum;
            A.aux[seq].inv_w = invW;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// kernel 2: backward sweep over the checkpointed blocks.
// LDS: per site of the block, the SPL alpha values entering the site and the site's scale.
// ---------------------------------------------------------------------------------------------
template <typename real, int K, int R, int T>
__global__ __launch_bounds__(NT_MAX) void bwd_kernel(KArgs A) {
    constexpr int SPL = K / R;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    real* smem = (real*)smem_raw;  // [T][SPL+1][blockDim.x]
    const int tid = threadIdx.x;
    const int64_t nseq = A.B * A.S;
    const int NT = blockDim.x;
    const int64_t gid = (int64_t)blockIdx.x * (NT / R) + tid / R;
    const bool active = gid < nseq;
    const int64_t seq = active ? gid : nseq - 1;
    const int rank = tid & (R - 1);
    const int64_t bb = seq / A.S, ss = seq - bb * A.S;

    Lane<real, K, R> lane;
    real pi[SPL];
    const real* prm = (const real*)A.params + bb * A.pstride_b + ss * A.pstride_s;
    lane.load(prm, rank, pi);
    const uint32_t* words = A.packed + A.inds[ss] * A.Lw;
    const real* ck = (const real*)A.ckpt;

    real beta[SPL], gb[SPL], gd[SPL], gu[SPL], gv[SPL], g0[SPL], g1[SPL];
    const real inv_end = (real)A.aux[seq].inv_end;
    const real inv_w = (real)A.aux[seq].inv_w;
#pragma unroll
    for (int i = 0; i < SPL; ++i) {
        beta[i] = inv_end;
        gb[i] = gd[i] = gu[i] = gv[i] = g0[i] = g1[i] = real(0);
    }
    constexpr bool F64ACC = sizeof(real) == 4;  // f32 kernels fold partial sums into f64
    double* gacc = A.gacc + seq * 6 * K + rank * SPL;
    int since_flush = 0;

    const int64_t nblk = (A.Ltot + T - 1) / T;
    real a[SPL], anext[SPL];
    if (nblk > 0) {
        const real* src = ck + ((nblk - 1) * nseq + seq) * K + rank * SPL;
#pragma unroll
        for (int i = 0; i < SPL; ++i) anext[i] = src[i];
    }
    for (int64_t blk = nblk - 1; blk >= 0; --blk) {
        const int64_t t0 = blk * T;
#pragma unroll
        for (int i = 0; i < SPL; ++i) a[i] = anext[i];
        if (blk > 0) {  // prefetch the previous block's checkpoint under this block's arithmetic
            const real* src = ck + ((blk - 1) * nseq + seq) * K + rank * SPL;
#pragma unroll
            for (int i = 0; i < SPL; ++i) anext[i] = src[i];
        }
        const uint32_t codes = block_codes(words, t0);
        const int ns = (int)((A.Ltot - t0) < T ? (A.Ltot - t0) : T);
        // re-run the block forward, keeping alpha_{t-1} and the scale of every site
        for (int i = 0; i < ns; ++i) {
            real* row = smem + (size_t)i * (SPL + 1) * NT + tid;
#pragma unroll
            for (int j = 0; j < SPL; ++j) row[j * NT] = a[j];
            real csum;
            const int ex = lane.fwd_site(a, (codes >> (2 * i)) & 3, csum);
            row[SPL * NT] = ldexp_(real(1), -ex);
        }
        // sweep it backwards; a = alpha after site i
        for (int i = ns - 1; i >= 0; --i) {
            if (t0 + i + 1 == A.W) {
#pragma unroll
                for (int j = 0; j < SPL; ++j) beta[j] = beta[j] - inv_w;
            }
            const real* row = smem + (size_t)i * (SPL + 1) * NT + tid;
            real ap[SPL];
#pragma unroll
            for (int j = 0; j < SPL; ++j) ap[j] = row[j * NT];
            const real s = row[SPL * NT];
            lane.bwd_site(ap, a, beta, (codes >> (2 * i)) & 3, s, gb, gd, gu, gv, g0, g1);
#pragma unroll
            for (int j = 0; j < SPL; ++j) a[j] = ap[j];
        }
        if constexpr (F64ACC) {
            since_flush += ns;
            if (since_flush >= FLUSH_SITES || blk == 0) {
                since_flush = 0;
                if (active) {
#pragma unroll
                    for (int i = 0; i < SPL; ++i) {
                        gacc[0 * K + i] += (double)gb[i];
                        gacc[1 * K + i] += (double)gd[i];
                        gacc[2 * K + i] += (double)gu[i];
                        gacc[3 * K + i] += (double)gv[i];
                        gacc[4 * K + i] += (double)g0[i];
                        gacc[5 * K + i] += (double)g1[i];
                    }
                }
#pragma unroll
                for (int i = 0; i < SPL; ++i) gb[i] = gd[i] = gu[i] = gv[i] = g0[i] = g1[i] = real(0);
            }
        }
    }
    if (!active) return;
    // d ll / d theta (or theta * that), rows b,d,u,v,emis0,emis1,pi
    real* out = (real*)A.grad + seq * 7 * K + rank * SPL;
    const bool dl = A.grad_dlog != 0;
#pragma unroll
    for (int i = 0; i < SPL; ++i) {
        double vb, vd, vu, vv, v0, v1;
        if constexpr (F64ACC) {
            vb = gacc[0 * K + i]; vd = gacc[1 * K + i]; vu = gacc[2 * K + i];
            vv = gacc[3 * K + i]; v0 = gacc[4 * K + i]; v1 = gacc[5 * K + i];
        } else {
            vb = gb[i]; vd = gd[i]; vu = gu[i]; vv = gv[i]; v0 = g0[i]; v1 = g1[i];
        }
        out[0 * K + i] = (real)(dl ? vb * (double)lane.b[i] : vb);
        out[1 * K + i] = (real)(dl ? vd * (double)lane.d[i] : vd);
        out[2 * K + i] = (real)(dl ? vu * (double)lane.u[i] : vu);
        out[3 * K + i] = (real)(dl ? vv * (double)lane.v[i] : vv);
        out[4 * K + i] = (real)(dl ? v0 : v0 / (double)lane.e0[i]);
        out[5 * K + i] = (real)(dl ? v1 : v1 / (double)lane.e1[i]);
        out[6 * K + i] = (real)(dl ? (double)beta[i] * (double)pi[i] : (double)beta[i]);
    }
}

// ---------------------------------------------------------------------------------------------
// upload-time re-pack: int8 {-1,0,1,(>1 clipped to 1; gpu.py:108-110)} -> 2-bit codes
// (a plain, non-template kernel: emitted only in the translation unit that defines PHK_WITH_PACK)
// ---------------------------------------------------------------------------------------------
#ifdef PHK_WITH_PACK
__global__ void pack_kernel(const int8_t* __restrict__ data, int64_t N, int64_t L, uint32_t* __restrict__ out,
                            int64_t Lw) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * Lw) return;
    const int64_t row = idx / Lw, w = idx - row * Lw;
    const int8_t* src = data + row * L;
    uint32_t word = 0;
    for (int j = 0; j < 16; ++j) {
        const int64_t t = w * 16 + j;
        uint32_t code = 2;  // padding decodes as "missing"; never read by the scan (t >= L)
        if (t < L) {
            const int o = src[t];
            code = o < 0 ? 2u : (o >= 1 ? 1u : 0u);
        }
        word |= code << (2 * j);
    }
    out[idx] = word;
}
#endif  // PHK_WITH_PACK

}  // namespace phk
