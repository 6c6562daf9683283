// The SVGD / AMSGrad update of the sampler's inner step on the device, in three launches.
//
// Reference: jthlab/phlash v1.0.6 src/phlash/mcmc.py:178-199, 279 delegates this to
// blackjax.svgd(grad(log_density), optax.amsgrad(lr)) (blackjax==1.2.5, optax==0.2.6; sources absent, so
// the definition this kernel is tested against is the torch restatement in phlash_amd/svgd.py):
//   k(x_i, x_j) = exp(-|x_i - x_j|^2 / h)
//   phi_j       = (1/n) sum_i [ -k_ij g_i + (2/h) (x_i - x_j) k_ij ]          (g = grad log p)
//   AMSGrad     : mu = b1 mu + (1-b1) phi, nu = b2 nu + (1-b2) phi^2, bias-corrected, running max of nu_hat,
//                 x <- x - lr mu_hat / (sqrt(nu_max) + eps)
//   h_next      = median(pairwise distances of the new x)^2 / log n   (median = torch.quantile(., 0.5))
// The torch restatement is ~70 launches of a few microseconds each per step (about 0.4 ms, i.e. 1 % of a cfg2
// step and 5 % of a step at the reference's production shape); here it is
//   svgd_update_kernel : one workgroup per particle j: its kernel row k(., x_j) into LDS, then one lane per
//                        coordinate sums over i in index order and applies the AMSGrad update (float64);
//   pair_dist_kernel   : the n (n - 1) / 2 pairwise distances of the new particles;
//   median_kernel      : ONE workgroup: exact bucket select of the two middle order statistics, h_next.  One
//                        workgroup reads at one CU's rate, so this is for populations of up to 256 particles
//                        (cfg2: 100 -> 4,950 distances, 25 us; 200: the whole step 54 us against 86 with a sort);
//   beyond that (the reference's default is 500 particles = 124,750 distances, mcmc.py:193; the limit here is 4,096
//   = 8.4 million) the same bucket select runs over the whole chip, three more launches (round 4; rounds 1-3 sorted
//   the distances with rocPRIM through torch.sort):
//   pair_dist_kernel   : also reduces min / max of the distances' bit patterns (workgroup, then one atomic each);
//   hist_kernel        : every workgroup histograms its share over 2,048 linear buckets of [min, max] in LDS and adds
//                        its non-empty buckets to the global histogram (integer atomics: order-independent);
//   gather_kernel      : every workgroup finds the bucket(s) holding the two middle ranks from the global histogram
//                        (the same prefix sum everywhere) and appends its elements of those buckets to a candidate list;
//   median_cand_kernel : one workgroup: the exact order statistics among the candidates (select_kth again), h_next.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "step_args.h"

namespace phk {

constexpr int SV_NT = 256;

__global__ __launch_bounds__(SV_NT) void svgd_update_kernel(SVArgs A) {
    __shared__ double xj[SV_MAXD];
    __shared__ double kj[SV_MAXB];  // k(x_i, x_j) for all i
    __shared__ double red[SV_NT];
    const int j = blockIdx.x, t = threadIdx.x, D = A.D;
    const double h = *A.h_in;
    for (int d = t; d < D; d += SV_NT) xj[d] = A.x[(int64_t)j * D + d];
    __syncthreads();
    for (int64_t i = t; i < A.B; i += SV_NT) {
        const double* xi = A.x + i * D;
        double d2 = 0.0;
        for (int d = 0; d < D; ++d) {
            const double df = xi[d] - xj[d];
            d2 = fma(df, df, d2);
        }
        kj[i] = exp(-d2 / h);
    }
    __syncthreads();
    // sum over i: the workgroup is cut into PARTS groups of D lanes (lane = coordinate: consecutive lanes read
    // consecutive addresses of x and g); part p takes i = p, p + PARTS, ...; the parts are then added in part
    // order by one lane per coordinate, so the result does not depend on timing (bit-reproducible)
    const int parts = SV_NT / D;
    const int part = t / D, d = t - part * D;
    const double two_over_h = 2.0 / h;
    double s = 0.0;
    if (part < parts) {
        for (int64_t i = part; i < A.B; i += parts) s += kj[i] * (two_over_h * (A.x[i * D + d] - xj[d]) - A.g[i * D + d]);
        red[part * D + d] = s;
    }
    __syncthreads();
    if (t < D) {
        double tot = 0.0;
        for (int p = 0; p < parts; ++p) tot += red[p * D + t];
        const double phi = tot / (double)A.B;
        const int64_t o = (int64_t)j * D + t;
        const double mu = A.b1 * A.mu[o] + (1.0 - A.b1) * phi;
        const double nu = A.b2 * A.nu[o] + (1.0 - A.b2) * phi * phi;
        const double mu_hat = mu / A.den1, nu_hat = nu / A.den2;
        const double nm = fmax(A.nu_max[o], nu_hat);
        A.mu[o] = mu;
        A.nu[o] = nu;
        A.nu_max[o] = nm;
        A.x_out[o] = xj[t] - A.lr * mu_hat / (sqrt(nm) + A.eps);
    }
}

// Scratch of the chip-wide select, behind the two double arrays of the workspace (see svgd_ws_doubles); set up by
// sel_init_kernel at the start of every step.
struct SelGlobal {
    unsigned long long mn, mx;  // min / max bit pattern of the distances (non-negative doubles order like their bits)
    int ncand;                  // candidates gathered
    int pad_;
    long long below;            // elements in buckets before the first gathered one
    int hist[2048];
};

// distances of the strict lower triangle, row-major: pair p <-> (i, j), i > j, p = i (i - 1) / 2 + j.  One workgroup
// per 16 x 16 tile of (i, j) with its 32 particles in LDS (round 3 had one thread per pair recover (i, j) from p
// with a float64 square root and read 2 D strided doubles: 51 us at 500 particles).
__global__ __launch_bounds__(256) void pair_dist_kernel(const double* __restrict__ x, int64_t B, int D, double* __restrict__ out,
                                                        SelGlobal* G) {
    __shared__ double xi[16][SV_MAXD + 1], xj[16][SV_MAXD + 1];
    const int ti = blockIdx.y, tj = blockIdx.x;
    if (tj > ti) return;  // (workgroup-uniform)
    const int t = threadIdx.x, a = t >> 4, c = t & 15;
    for (int e = t; e < 16 * D; e += 256) {
        const int r = e / D, d = e - r * D;
        const int64_t gi = 16 * (int64_t)ti + r, gj = 16 * (int64_t)tj + r;
        xi[r][d] = gi < B ? x[gi * D + d] : 0.0;
        xj[r][d] = gj < B ? x[gj * D + d] : 0.0;
    }
    __syncthreads();
    const int64_t i = 16 * (int64_t)ti + a, j = 16 * (int64_t)tj + c;
    unsigned long long key_mn = ~0ull, key_mx = 0ull;
    if (i < B && j < i) {
        double d2 = 0.0;
        for (int d = 0; d < D; ++d) {
            const double df = xi[a][d] - xj[c][d];
            d2 = fma(df, df, d2);
        }
        const double dist = sqrt(d2);
        out[i * (i - 1) / 2 + j] = dist;
        key_mn = key_mx = (unsigned long long)__double_as_longlong(dist);
    }
    if (G == nullptr) return;  // (a kernel argument: uniform)
    // min / max over the workgroup (wave shuffles, then LDS), one pair of atomics per workgroup: with one pair per wave
    // the two contended addresses were most of this kernel's 50 us at 500 particles
    __shared__ unsigned long long wmn[4], wmx[4];
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long p = __shfl_xor(key_mn, off), q = __shfl_xor(key_mx, off);
        key_mn = p < key_mn ? p : key_mn;
        key_mx = q > key_mx ? q : key_mx;
    }
    if ((t & 63) == 0) {
        wmn[t >> 6] = key_mn;
        wmx[t >> 6] = key_mx;
    }
    __syncthreads();
    if (t == 0) {
        for (int w = 1; w < 4; ++w) {
            key_mn = wmn[w] < key_mn ? wmn[w] : key_mn;
            key_mx = wmx[w] > key_mx ? wmx[w] : key_mx;
        }
        if (key_mn != ~0ull) {
            atomicMin(&G->mn, key_mn);
            atomicMax(&G->mx, key_mx);
        }
    }
}

// k-th smallest (0-based) of n non-negative doubles, one workgroup, exact.  Bucket select: min and max of the
// candidates; a monotone map of [min, max] onto SEL_BINS buckets (so the buckets partition the candidates in
// order) and a histogram of it -- pairwise distances spread over the buckets, so the LDS atomics do not pile up
// on one address the way a radix digit of their nearly identical bit patterns would (a radix select took 380
// us here, 55 us per pass); the bucket holding the k-th element is gathered into LDS and ranked by counting.
// A bucket too large to gather (many equal values) becomes the new candidate range.
constexpr int SEL_BINS = 2048, SEL_CAP = 4096;

struct SelShared {
    int hist[SEL_BINS];
    double cand[SEL_CAP];
    unsigned long long mn, mx;
    int ncand, bucket;
    long long k_in;
    double result;
    int wsum[16];  // per-wave totals of the histogram prefix
    int cnt;
};

__device__ double select_kth(const double* __restrict__ v, int64_t n, int64_t k, SelShared& S) {
    const int t = threadIdx.x, nt = blockDim.x;
    unsigned long long lo_key = 0ull, hi_key = ~0ull;  // candidates: lo_key <= key <= hi_key
    long long k_rel = k;                               // rank of the wanted element among the candidates
    for (int round = 0; round < 64; ++round) {
        // min / max of the candidates
        if (t == 0) {
            S.mn = ~0ull;
            S.mx = 0ull;
        }
        __syncthreads();
        unsigned long long mn = ~0ull, mx = 0ull;
        for (int64_t e = t; e < n; e += nt) {
            const unsigned long long key = (unsigned long long)__double_as_longlong(v[e]);
            if (key >= lo_key && key <= hi_key) {
                mn = key < mn ? key : mn;
                mx = key > mx ? key : mx;
            }
        }
        atomicMin(&S.mn, mn);
        atomicMax(&S.mx, mx);
        __syncthreads();
        const unsigned long long kmn = S.mn, kmx = S.mx;
        if (kmn == kmx) return __longlong_as_double((long long)kmn);
        const double vmn = __longlong_as_double((long long)kmn), vmx = __longlong_as_double((long long)kmx);
        const double scale = (double)(SEL_BINS - 1) / (vmx - vmn);
        for (int b = t; b < SEL_BINS; b += nt) S.hist[b] = 0;
        if (t == 0) S.ncand = 0;
        __syncthreads();
        for (int64_t e = t; e < n; e += nt) {
            const double x = v[e];
            const unsigned long long key = (unsigned long long)__double_as_longlong(x);
            if (key >= kmn && key <= kmx) {
                int b = (int)((x - vmn) * scale);
                b = b > SEL_BINS - 1 ? SEL_BINS - 1 : b;
                atomicAdd(&S.hist[b], 1);
            }
        }
        __syncthreads();
        // the bucket holding rank k_rel: exclusive prefix of the histogram over the workgroup (thread t owns BPT
        // consecutive buckets; a serial walk by one thread was 35 of this kernel's 65 us per select)
        {
            constexpr int BPT = SEL_BINS / 1024;
            static_assert(SEL_BINS % 1024 == 0, "median_kernel runs 1024 threads");
            int own = 0;
#pragma unroll
            for (int i = 0; i < BPT; ++i) own += S.hist[t * BPT + i];
            const int lane = t & 63, wv = t >> 6;
            int inc = own;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int y = __shfl_up(inc, off);
                if (lane >= off) inc += y;
            }
            if (lane == 63) S.wsum[wv] = inc;
            __syncthreads();
            long long before = inc - own;
            for (int i = 0; i < wv; ++i) before += S.wsum[i];
            if (k_rel >= before && k_rel < before + own) {  // exactly one thread
                long long kk = k_rel - before;
                int bk = t * BPT;
                for (; bk < t * BPT + BPT - 1; ++bk) {
                    if (kk < S.hist[bk]) break;
                    kk -= S.hist[bk];
                }
                S.bucket = bk;
                S.k_in = kk;
            }
        }
        __syncthreads();
        const int bsel = S.bucket;
        const long long k_in = S.k_in;
        const int c = S.hist[bsel];
        if (c <= SEL_CAP) {
            for (int64_t e = t; e < n; e += nt) {
                const double x = v[e];
                const unsigned long long key = (unsigned long long)__double_as_longlong(x);
                if (key >= kmn && key <= kmx) {
                    int b = (int)((x - vmn) * scale);
                    b = b > SEL_BINS - 1 ? SEL_BINS - 1 : b;
                    if (b == bsel) S.cand[atomicAdd(&S.ncand, 1)] = x;
                }
            }
            __syncthreads();
            for (int e = t; e < c; e += nt) {
                const double x = S.cand[e];
                int rank = 0;
                for (int f = 0; f < c; ++f) rank += (S.cand[f] < x || (S.cand[f] == x && f < e)) ? 1 : 0;
                if (rank == (int)k_in) S.result = x;
            }
            __syncthreads();
            return S.result;
        }
        // too many candidates in that bucket: narrow the range to it (its own min / max next round)
        unsigned long long bmn = ~0ull, bmx = 0ull;
        for (int64_t e = t; e < n; e += nt) {
            const double x = v[e];
            const unsigned long long key = (unsigned long long)__double_as_longlong(x);
            if (key >= kmn && key <= kmx) {
                int b = (int)((x - vmn) * scale);
                b = b > SEL_BINS - 1 ? SEL_BINS - 1 : b;
                if (b == bsel) {
                    bmn = key < bmn ? key : bmn;
                    bmx = key > bmx ? key : bmx;
                }
            }
        }
        if (t == 0) {
            S.mn = ~0ull;
            S.mx = 0ull;
        }
        __syncthreads();
        atomicMin(&S.mn, bmn);
        atomicMax(&S.mx, bmx);
        __syncthreads();
        lo_key = S.mn;
        hi_key = S.mx;
        k_rel = k_in;
        __syncthreads();
    }
    return __longlong_as_double((long long)lo_key);  // (not reached: every round shrinks the range)
}

__global__ __launch_bounds__(1024) void median_kernel(const double* __restrict__ v, int64_t n, int64_t B, double* h_out) {
    __shared__ SelShared S;
    if (n <= 0) {  // a single particle: the length scale stays at its initial value 1
        if (threadIdx.x == 0) *h_out = 1.0;
        return;
    }
    // torch.quantile(v, 0.5), linear interpolation: position 0.5 (n - 1) between two order statistics
    const double pos = 0.5 * (double)(n - 1);
    const int64_t lo = (int64_t)floor(pos), hi = (int64_t)ceil(pos);
    const double vlo = select_kth(v, n, lo, S);
    __syncthreads();
    double vhi = vlo;
    if (hi != lo) {
        // the next order statistic without a second select: vlo again if more than lo + 1 elements are <= vlo, else
        // the smallest element above it (distances are >= 0: their bit patterns order like their values)
        if (threadIdx.x == 0) {
            S.cnt = 0;
            S.mn = ~0ull;
        }
        __syncthreads();
        const unsigned long long klo = (unsigned long long)__double_as_longlong(vlo);
        int c = 0;
        unsigned long long nx = ~0ull;
        for (int64_t e = threadIdx.x; e < n; e += blockDim.x) {
            const unsigned long long key = (unsigned long long)__double_as_longlong(v[e]);
            if (key <= klo) ++c;
            else nx = key < nx ? key : nx;
        }
        atomicAdd(&S.cnt, c);
        atomicMin(&S.mn, nx);
        __syncthreads();
        vhi = (int64_t)S.cnt >= hi + 1 ? vlo : __longlong_as_double((long long)S.mn);
    }
    if (threadIdx.x == 0) {
        const double med = vlo + (vhi - vlo) * (pos - (double)lo);
        *h_out = med * med / log((double)B);
    }
}

// ---- the select over the whole chip (more than SEL_ONE_WG distances) -------------------------------------------
constexpr int64_t SEL_ONE_WG = 32768;  // distances (256 particles) up to which the single-workgroup select is used (it wins up to ~300 particles)

__device__ __forceinline__ int sel_bucket(double x, double vmn, double scale) {
    int b = (int)((x - vmn) * scale);
    return b > SEL_BINS - 1 ? SEL_BINS - 1 : b;
}

__global__ __launch_bounds__(256) void hist_kernel(const double* __restrict__ v, int64_t n, SelGlobal* G) {
    __shared__ int h[SEL_BINS];
    const unsigned long long kmn = G->mn, kmx = G->mx;
    if (kmn == kmx) return;  // all distances equal: nothing to narrow
    const double vmn = __longlong_as_double((long long)kmn), vmx = __longlong_as_double((long long)kmx);
    const double scale = (double)(SEL_BINS - 1) / (vmx - vmn);
    for (int b = threadIdx.x; b < SEL_BINS; b += 256) h[b] = 0;
    __syncthreads();
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) atomicAdd(&h[sel_bucket(v[e], vmn, scale)], 1);
    __syncthreads();
    for (int b = threadIdx.x; b < SEL_BINS; b += 256)
        if (h[b]) atomicAdd(&G->hist[b], h[b]);
}

// bucket holding global rank k, and the number of elements in the buckets before it (every thread of the workgroup
// gets both): 256 threads x 8 buckets
__device__ void rank_bucket(const int* __restrict__ hist, long long k, int* sh, int& bucket, long long& before) {
    const int t = threadIdx.x;
    int own = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) own += hist[t * 8 + i];
    sh[t] = own;
    __syncthreads();
    if (t == 0) {  // 256 partial sums: a serial walk, once per workgroup
        long long acc = 0;
        int th = 0;
        for (; th < 255; ++th) {
            if (k < acc + sh[th]) break;
            acc += sh[th];
        }
        int b = th * 8;
        for (; b < th * 8 + 7; ++b) {
            if (k < acc + hist[b]) break;
            acc += hist[b];
        }
        sh[256] = b;
        sh[257] = (int)(acc & 0xffffffffll);
        sh[258] = (int)(acc >> 32);
    }
    __syncthreads();
    bucket = sh[256];
    before = ((long long)sh[258] << 32) | (unsigned int)sh[257];
    __syncthreads();
}

__global__ __launch_bounds__(256) void gather_kernel(const double* __restrict__ v, int64_t n, SelGlobal* G, double* __restrict__ cand) {
    __shared__ int sh[260];
    const unsigned long long kmn = G->mn, kmx = G->mx;
    if (kmn == kmx) return;
    const double vmn = __longlong_as_double((long long)kmn), vmx = __longlong_as_double((long long)kmx);
    const double scale = (double)(SEL_BINS - 1) / (vmx - vmn);
    const double pos = 0.5 * (double)(n - 1);
    const long long lo = (long long)floor(pos), hi = (long long)ceil(pos);
    int b_lo, b_hi;
    long long before_lo, before_hi;
    rank_bucket(G->hist, lo, sh, b_lo, before_lo);
    rank_bucket(G->hist, hi, sh, b_hi, before_hi);
    if (blockIdx.x == 0 && threadIdx.x == 0) G->below = before_lo;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const double x = v[e];
        const int b = sel_bucket(x, vmn, scale);
        if (b >= b_lo && b <= b_hi) cand[atomicAdd(&G->ncand, 1)] = x;  // (buckets strictly between the two are empty)
    }
}

// one workgroup: median of the n distances from the candidate list (all elements of the bucket(s) holding the two middle
// ranks; `below` elements lie in earlier buckets), h_out, and the scratch back in its resting state
__global__ __launch_bounds__(1024) void median_cand_kernel(const double* __restrict__ cand, int64_t n, int64_t B, SelGlobal* G, double* h_out) {
    __shared__ SelShared S;
    const unsigned long long kmn = G->mn, kmx = G->mx;
    const int nc = G->ncand;
    const long long below = G->below;
    __syncthreads();
    const double pos = 0.5 * (double)(n - 1);
    const int64_t lo = (int64_t)floor(pos), hi = (int64_t)ceil(pos);
    double vlo, vhi;
    if (kmn == kmx) {
        vlo = vhi = __longlong_as_double((long long)kmn);
    } else {
        vlo = select_kth(cand, nc, lo - below, S);
        __syncthreads();
        vhi = vlo;
        if (hi != lo) {  // (as in median_kernel, on the candidates: ranks shifted by `below`)
            if (threadIdx.x == 0) {
                S.cnt = 0;
                S.mn = ~0ull;
            }
            __syncthreads();
            const unsigned long long klo = (unsigned long long)__double_as_longlong(vlo);
            int c = 0;
            unsigned long long nx = ~0ull;
            for (int64_t e = threadIdx.x; e < nc; e += blockDim.x) {
                const unsigned long long key = (unsigned long long)__double_as_longlong(cand[e]);
                if (key <= klo) ++c;
                else nx = key < nx ? key : nx;
            }
            atomicAdd(&S.cnt, c);
            atomicMin(&S.mn, nx);
            __syncthreads();
            vhi = below + (int64_t)S.cnt >= hi + 1 ? vlo : __longlong_as_double((long long)S.mn);
        }
    }
    if (threadIdx.x == 0) {
        const double med = vlo + (vhi - vlo) * (pos - (double)lo);
        *h_out = med * med / log((double)B);
    }
}

// (every step: the workspace is the caller's and may have held anything in between)
__global__ void sel_init_kernel(SelGlobal* G) {
    for (int b = threadIdx.x; b < SEL_BINS; b += blockDim.x) G->hist[b] = 0;
    if (threadIdx.x == 0) {
        G->mn = ~0ull;
        G->mx = 0ull;
        G->ncand = 0;
        G->below = 0;
    }
}

// doubles of workspace phk_svgd_step needs for B particles: the distances; beyond the single-workgroup limit as many
// again for the candidate list (a bucket can hold all of them when they are all equal but one) and the SelGlobal block
int64_t svgd_ws_doubles(int64_t B) {
    const int64_t n = B * (B - 1) / 2;
    if (n <= SEL_ONE_WG) return n > 0 ? n : 1;
    return 2 * n + (int64_t)((sizeof(SelGlobal) + 7) / 8);
}

hipError_t launch_svgd_step(const SVArgs& a, double* dist_ws, double* h_out, hipStream_t st) {
    if (a.B <= 0) return hipSuccess;
    hipLaunchKernelGGL(svgd_update_kernel, dim3((unsigned)a.B), dim3(SV_NT), 0, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int64_t n = a.B * (a.B - 1) / 2;
    const bool chip = h_out != nullptr && n > SEL_ONE_WG;
    SelGlobal* G = chip ? (SelGlobal*)(dist_ws + 2 * n) : nullptr;
    if (chip) hipLaunchKernelGGL(sel_init_kernel, dim3(1), dim3(256), 0, st, G);
    if (n > 0) {
        const unsigned tiles = (unsigned)((a.B + 15) / 16);
        hipLaunchKernelGGL(pair_dist_kernel, dim3(tiles, tiles), dim3(256), 0, st, (const double*)a.x_out, a.B, a.D, dist_ws, G);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (h_out == nullptr) return hipGetLastError();
    if (!chip) {
        hipLaunchKernelGGL(median_kernel, dim3(1), dim3(1024), 0, st, (const double*)dist_ws, n, a.B, h_out);
        return hipGetLastError();
    }
    const unsigned wgs = (unsigned)std::min<int64_t>((n + 256 * 8 - 1) / (256 * 8), 2048);
    hipLaunchKernelGGL(hist_kernel, dim3(wgs), dim3(256), 0, st, (const double*)dist_ws, n, G);
    hipLaunchKernelGGL(gather_kernel, dim3(wgs), dim3(256), 0, st, (const double*)dist_ws, n, G, dist_ws + n);
    hipLaunchKernelGGL(median_cand_kernel, dim3(1), dim3(1024), 0, st, (const double*)(dist_ws + n), n, a.B, G, h_out);
    return hipGetLastError();
}

}  // namespace phk
