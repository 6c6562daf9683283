// C ABI of libphlash_hip.so (declared in include/phlash_hip.h): handle life cycle, upload-time
// re-pack, scratch management, variant selection and the two-kernel launch sequence.
// Host side of what the reference does in _PSMCKernelBase (src/phlash/gpu.py:101-325), minus the
// host<->device copies per call: every per-call buffer here is a device pointer.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <vector>

#include "../../include/phlash_hip.h"
#define PHK_WITH_PACK 1
#include "psmc_kernels.hip"

namespace phk {
#define PHK_DECL(tag)                                                                                 \
    hipError_t launch_fwd_##tag(int R, int T, int nrm, bool ckpt, const KArgs& a, int nt, hipStream_t st); \
    hipError_t launch_bwd_##tag(int R, int T, int nrm, const KArgs& a, int nt, hipStream_t st);
PHK_DECL(f32_4) PHK_DECL(f32_8) PHK_DECL(f32_16) PHK_DECL(f32_32) PHK_DECL(f32_64)
PHK_DECL(f64_4) PHK_DECL(f64_8) PHK_DECL(f64_16) PHK_DECL(f64_32) PHK_DECL(f64_64)
#undef PHK_DECL

constexpr int PM_MAXK = 64;
struct PMArgs {  // must match param_map.hip
    int K, P, D;
    double theta;
    int8_t epoch[PM_MAXK];
    const double* x;
    double* params;
    double* jac;
    int64_t B;
};
hipError_t launch_param_map(const PMArgs& a, hipStream_t st);
}  // namespace phk

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(e_ == hipErrorOutOfMemory ? PHK_ENOMEM : PHK_EHIP, "%s: %s", #expr, \
                        hipGetErrorString(e_));                                            \
    } while (0)

constexpr int DEFAULT_NRM = 4;  // rescale every 4th site unless told otherwise (measured: 1 -> 5.0e10, 2 -> 5.3e10, 4 -> 5.4e10; same parity)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return PHK_OK;
        if (p) {
            hipError_t e = hipFree(p);  // synchronises with outstanding work on the buffer
            p = nullptr;
            cap = 0;
            if (e != hipSuccess) return fail(PHK_EHIP, "hipFree: %s", hipGetErrorString(e));
        }
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) {
            p = nullptr;
            return fail(PHK_ENOMEM, "while trying to allocate %zu bytes on GPU: %s", bytes, hipGetErrorString(e));
        }
        cap = bytes;
        return PHK_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

}  // namespace

struct phk_handle {
    int K = 0, device = 0, dbl = 0;
    int64_t N = 0, L = 0, Lw = 0;
    uint32_t* packed = nullptr;
    DevBuf ckpt, aux, gacc, tune_ll, tune_grad;
    int64_t last_total = -1;  // B*S of the last phk_loglik and the variant it ran with
    int last_R = 0, last_T = 0;
    int autotune = 1;                                               // time the candidate variants once per batch shape
    std::map<std::pair<int64_t, int>, std::pair<int, int>> tuned;   // (sequences per launch, grad?) -> (R, T)
    int64_t ws_limit = 0;
    int force_R = 0, force_T = 0, nrm = DEFAULT_NRM;
    int profiling = 0;
    std::vector<hipEvent_t> ev;  // triples (start, mid, end) per launch pair since the last timing query
    int n_launches = 0;          // launch pairs recorded since the last query
    int n_last = 0;              // ... of which by the last call
};

namespace {

bool valid_R(int K, int R) { return R >= 1 && R <= 16 && (R & (R - 1)) == 0 && R <= K && K % R == 0 && K / R <= 16; }

size_t real_size(const phk_handle* h) { return h->dbl ? 8 : 4; }

// T = 16 keeps 17 alpha vectors in registers: only compiled where a lane owns <= 4 states
bool valid_T(int K, int R, int T) { return T == 8 || (T == 16 && K / R <= 4); }

void choose_variant(const phk_handle* h, int64_t nseq, int* R, int* T, int want_grad = 1) {
    int r = h->force_R, t = h->force_T;
    if (!r && !t) {
        auto it = h->tuned.find({nseq, want_grad});
        if (it != h->tuned.end()) {
            *R = it->second.first;
            *T = it->second.second;
            return;
        }
    }
    if (!t) t = 8;
    if (!r) {
        // smallest R (fewest cross-lane steps, fewest instructions per site.particle) that still
        // gives every one of the 1024 SIMDs a wave; otherwise the largest valid R.  Measured at
        // cfg2 (50,000 sequences, K=16, f32): R=2 3.6e10, R=1 3.2e10, R=4 2.8e10 site.particle/s.
        int best = 0;
        for (int c = 1; c <= 16; c <<= 1) {
            if (!valid_R(h->K, c)) continue;
            if (!valid_T(h->K, c, t)) continue;
            best = c;
            if (nseq * c / 64 >= 1024) break;
        }
        r = best ? best : 1;
    }
    *R = r;
    *T = t;
}

typedef hipError_t (*fwd_fn)(int, int, int, bool, const phk::KArgs&, int, hipStream_t);
typedef hipError_t (*bwd_fn)(int, int, int, const phk::KArgs&, int, hipStream_t);

bool pick_launchers(const phk_handle* h, fwd_fn* f, bwd_fn* b) {
#define PHK_CASE(k)                                                        \
    case k:                                                                \
        *f = h->dbl ? phk::launch_fwd_f64_##k : phk::launch_fwd_f32_##k;   \
        *b = h->dbl ? phk::launch_bwd_f64_##k : phk::launch_bwd_f32_##k;   \
        return true;
    switch (h->K) {
        PHK_CASE(4) PHK_CASE(8) PHK_CASE(16) PHK_CASE(32) PHK_CASE(64)
    }
#undef PHK_CASE
    return false;
}

// Time every compiled (R, T) on the first `tune_sites` sites of this very batch (scratch outputs)
// and remember the fastest for this (sequence count, gradient?) shape.  One-off, a few tens of ms.
int autotune_variant(phk_handle* h, const phk::KArgs& proto, bool want_grad, fwd_fn lf, bwd_fn lb, hipStream_t st) {
    const int K = h->K;
    const size_t rs = real_size(h);
    const int64_t nseq = proto.B * proto.S;
    const int64_t tune_sites = std::min<int64_t>(h->L, 2048);
    int rc;
    if ((rc = h->tune_ll.ensure((size_t)nseq * sizeof(double))) != PHK_OK) return rc;
    if (want_grad && (rc = h->tune_grad.ensure((size_t)nseq * 7 * K * rs)) != PHK_OK) return rc;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    float best = 0.f;
    int bestR = 0, bestT = 0;
    for (int R = 1; R <= 16; R <<= 1) {
        if (!valid_R(K, R)) continue;
        for (int T = 8; T <= 16; T += 8) {
            if (!valid_T(K, R, T)) continue;
            phk::KArgs a = proto;
            a.Ltot = tune_sites;
            a.W = std::min<int64_t>(proto.W, tune_sites);
            a.ll = (double*)h->tune_ll.p;
            a.grad = want_grad ? h->tune_grad.p : nullptr;
            float ms = 0.f;
            for (int rep = 0; rep < 2; ++rep) {  // rep 0 loads the code object / warms the caches
                if (want_grad && !h->dbl) HIP_TRY(hipMemsetAsync(h->gacc.p, 0, (size_t)nseq * 6 * K * sizeof(double), st));
                HIP_TRY(hipEventRecord(e0, st));
                hipError_t e = lf(R, T, h->nrm, want_grad, a, 256, st);
                if (e == hipSuccess && want_grad) e = lb(R, T, h->nrm, a, 256, st);
                if (e != hipSuccess) return fail(PHK_EHIP, "autotune launch (K=%d R=%d T=%d): %s", K, R, T, hipGetErrorString(e));
                HIP_TRY(hipEventRecord(e1, st));
                HIP_TRY(hipEventSynchronize(e1));
                HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
            }
            if (!bestR || ms < best) {
                best = ms;
                bestR = R;
                bestT = T;
            }
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (bestR) h->tuned[{nseq, want_grad ? 1 : 0}] = {bestR, bestT};
    return PHK_OK;
}

}  // namespace

extern "C" {

int phk_version(void) { return 1000; }

const char* phk_last_error(void) { return g_err; }

int phk_device_count(int* n) {
    if (!n) return fail(PHK_EINVAL, "n is NULL");
    HIP_TRY(hipGetDeviceCount(n));
    return PHK_OK;
}

int phk_create(phk_handle** out, int K, const int8_t* data, int64_t N, int64_t L, int data_on_device,
               int double_precision, int device) {
    if (!out) return fail(PHK_EINVAL, "out is NULL");
    *out = nullptr;
    if (!(K == 4 || K == 8 || K == 16 || K == 32 || K == 64))
        return fail(PHK_EUNSUPPORTED, "K=%d not compiled in (supported: 4, 8, 16, 32, 64)", K);
    if (!data || N <= 0 || L <= 0) return fail(PHK_EINVAL, "data must be a non-empty [N, L] int8 matrix");
    if (!data_on_device) {
        // the reference's checks (gpu.py:106-113): min >= -1, no all-missing row
        for (int64_t n = 0; n < N; ++n) {
            int mx = -128;
            const int8_t* row = data + n * L;
            for (int64_t t = 0; t < L; ++t) {
                if (row[t] < -1) return fail(PHK_EINVAL, "data[%lld][%lld] = %d < -1", (long long)n, (long long)t, (int)row[t]);
                mx = std::max(mx, (int)row[t]);
            }
            if (mx <= -1) return fail(PHK_EINVAL, "data contains observations with all missing values (row %lld)", (long long)n);
        }
    }
    HIP_TRY(hipSetDevice(device));
    phk_handle* h = new (std::nothrow) phk_handle();
    if (!h) return fail(PHK_ENOMEM, "host allocation failed");
    h->K = K;
    h->device = device;
    h->dbl = double_precision ? 1 : 0;
    h->N = N;
    h->L = L;
    h->Lw = ((L + 15) / 16 + 3) & ~(int64_t)3;  // dwords per row, rows 16-byte aligned
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) h->ws_limit = (int64_t)(free_b / 2);
    else h->ws_limit = (int64_t)32 << 30;
    if (const char* env = std::getenv("PHK_AUTOTUNE")) h->autotune = std::atoi(env) != 0;

    int rc = PHK_OK;
    int8_t* staged = nullptr;
    const int8_t* dsrc = data;
    hipError_t e = hipMalloc((void**)&h->packed, (size_t)N * h->Lw * 4);
    if (e != hipSuccess) {
        rc = fail(PHK_ENOMEM, "while trying to allocate %zu bytes on GPU: %s", (size_t)N * h->Lw * 4, hipGetErrorString(e));
    }
    if (rc == PHK_OK && !data_on_device) {
        e = hipMalloc((void**)&staged, (size_t)N * L);
        if (e != hipSuccess) rc = fail(PHK_ENOMEM, "while trying to allocate %zu bytes on GPU: %s", (size_t)N * L, hipGetErrorString(e));
        if (rc == PHK_OK) {
            e = hipMemcpy(staged, data, (size_t)N * L, hipMemcpyHostToDevice);
            if (e != hipSuccess) rc = fail(PHK_EHIP, "hipMemcpy: %s", hipGetErrorString(e));
        }
        dsrc = staged;
    }
    if (rc == PHK_OK) {
        const int64_t total = N * h->Lw;
        hipLaunchKernelGGL(phk::pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, dsrc, N, L, h->packed, h->Lw);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) rc = fail(PHK_EHIP, "pack kernel: %s", hipGetErrorString(e));
    }
    if (staged) (void)hipFree(staged);
    if (rc != PHK_OK) {
        if (h->packed) (void)hipFree(h->packed);
        delete h;
        return rc;
    }
    *out = h;
    return PHK_OK;
}

int phk_destroy(phk_handle* h) {
    if (!h) return PHK_OK;
    (void)hipSetDevice(h->device);
    for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
    h->ckpt.release();
    h->aux.release();
    h->gacc.release();
    h->tune_ll.release();
    h->tune_grad.release();
    if (h->packed) (void)hipFree(h->packed);
    delete h;
    return PHK_OK;
}

int phk_set_variant(phk_handle* h, int R, int T) {
    if (!h) return fail(PHK_EINVAL, "handle is NULL");
    if (R != 0 && !valid_R(h->K, R)) return fail(PHK_EINVAL, "R=%d invalid for K=%d", R, h->K);
    if (T != 0 && T != 8 && T != 16) return fail(PHK_EINVAL, "T must be 0, 8 or 16");
    if (R != 0 && !valid_T(h->K, R, T ? T : 8)) return fail(PHK_EINVAL, "R=%d T=%d not available (T=16 needs K/R <= 4)", R, T);
    h->force_R = R;
    h->force_T = T;
    return PHK_OK;
}

int phk_set_autotune(phk_handle* h, int on) {
    if (!h) return fail(PHK_EINVAL, "handle is NULL");
    h->autotune = on ? 1 : 0;
    if (!on) h->tuned.clear();
    return PHK_OK;
}

int phk_set_rescale_interval(phk_handle* h, int nrm) {
    if (!h) return fail(PHK_EINVAL, "handle is NULL");
    if (nrm == 0) nrm = DEFAULT_NRM;
    if (nrm != 1 && nrm != 2 && nrm != 4) return fail(PHK_EINVAL, "rescale interval must be 1, 2 or 4 (0 = default)");
    h->nrm = nrm;
    return PHK_OK;
}

int phk_get_variant(phk_handle* h, int64_t B, int64_t S, int* R, int* T) {
    if (!h || !R || !T) return fail(PHK_EINVAL, "NULL argument");
    if (B * S == h->last_total && h->last_R) {  // what the last call of this shape actually ran (slabs included)
        *R = h->last_R;
        *T = h->last_T;
        return PHK_OK;
    }
    choose_variant(h, B * S, R, T);
    return PHK_OK;
}

int phk_set_workspace_limit(phk_handle* h, int64_t bytes) {
    if (!h || bytes <= 0) return fail(PHK_EINVAL, "bad workspace limit");
    h->ws_limit = bytes;
    return PHK_OK;
}

int64_t phk_workspace_bytes(phk_handle* h) { return h ? (int64_t)(h->ckpt.cap + h->aux.cap + h->gacc.cap) : 0; }

int phk_set_profiling(phk_handle* h, int on) {
    if (!h) return fail(PHK_EINVAL, "handle is NULL");
    h->profiling = on ? 1 : 0;
    return PHK_OK;
}

int phk_last_timing(phk_handle* h, float* fwd_ms, float* bwd_ms, int* n_launches) {
    if (!h) return fail(PHK_EINVAL, "handle is NULL");
    float f = 0.f, b = 0.f;
    for (int i = h->n_launches - h->n_last; i < h->n_launches; ++i) {
        hipEvent_t e0 = h->ev[3 * i], e1 = h->ev[3 * i + 1], e2 = h->ev[3 * i + 2];
        HIP_TRY(hipEventSynchronize(e2));
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, e0, e1));
        f += t;
        HIP_TRY(hipEventElapsedTime(&t, e1, e2));
        b += t;
    }
    if (fwd_ms) *fwd_ms = f;
    if (bwd_ms) *bwd_ms = b;
    if (n_launches) *n_launches = h->n_last;
    return PHK_OK;
}

int phk_timing_totals(phk_handle* h, double* fwd_ms, double* bwd_ms, int* n_launches) {
    if (!h) return fail(PHK_EINVAL, "handle is NULL");
    double f = 0.0, b = 0.0;
    for (int i = 0; i < h->n_launches; ++i) {
        hipEvent_t e0 = h->ev[3 * i], e1 = h->ev[3 * i + 1], e2 = h->ev[3 * i + 2];
        HIP_TRY(hipEventSynchronize(e2));
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, e0, e1));
        f += t;
        HIP_TRY(hipEventElapsedTime(&t, e1, e2));
        b += t;
    }
    if (fwd_ms) *fwd_ms = f;
    if (bwd_ms) *bwd_ms = b;
    if (n_launches) *n_launches = h->n_launches;
    h->n_launches = 0;  // events are recycled from here on
    h->n_last = 0;
    return PHK_OK;
}

int phk_param_map(int device, int K, int P, const int32_t* epoch_of_state, double theta, const double* x, int64_t B,
                  double* params, double* jac, void* stream) {
    if (K < 3 || K > phk::PM_MAXK) return fail(PHK_EUNSUPPORTED, "K=%d outside [3, %d]", K, phk::PM_MAXK);
    if (P < 1 || P > K) return fail(PHK_EINVAL, "P=%d epochs for K=%d states", P, K);
    if (!epoch_of_state || !x || !params) return fail(PHK_EINVAL, "NULL argument");
    if (B < 0) return fail(PHK_EINVAL, "B must be >= 0");
    phk::PMArgs a;
    a.K = K;
    a.P = P;
    a.D = P + 3;
    a.theta = theta;
    for (int k = 0; k < K; ++k) {
        if (epoch_of_state[k] < 0 || epoch_of_state[k] >= P) return fail(PHK_EINVAL, "epoch_of_state[%d] = %d outside [0, %d)", k, epoch_of_state[k], P);
        a.epoch[k] = (int8_t)epoch_of_state[k];
    }
    a.x = x;
    a.params = params;
    a.jac = jac;
    a.B = B;
    HIP_TRY(hipSetDevice(device));
    hipError_t e = phk::launch_param_map(a, (hipStream_t)stream);
    if (e != hipSuccess) return fail(PHK_EHIP, "param_map kernel launch: %s", hipGetErrorString(e));
    return PHK_OK;
}

int phk_loglik(phk_handle* h, const void* params, int64_t pstride_b, int64_t pstride_s, const int64_t* inds,
               int64_t B, int64_t S, int64_t W, double* ll, void* grad, int grad_dlog, void* stream) {
    if (!h) return fail(PHK_EINVAL, "handle is NULL");
    if (!params || !inds || !ll) return fail(PHK_EINVAL, "params, inds and ll must be non-NULL device pointers");
    if (B < 0 || S < 0) return fail(PHK_EINVAL, "B and S must be >= 0");
    if (W < 0 || W > h->L) return fail(PHK_EINVAL, "W=%lld outside [0, L=%lld]", (long long)W, (long long)h->L);
    if (B == 0 || S == 0) return PHK_OK;
    fwd_fn lf = nullptr;
    bwd_fn lb = nullptr;
    if (!pick_launchers(h, &lf, &lb)) return fail(PHK_EUNSUPPORTED, "K=%d not compiled in", h->K);
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    const size_t rs = real_size(h);
    const int K = h->K;
    const bool want_grad = grad != nullptr;
    h->n_last = 0;
    if (h->n_launches > 4096) h->n_launches = 0;  // nobody is collecting: recycle the event pool

    const int nt_b = 256, nt_f = 256;

    // slab the (particle, chunk) grid so that the checkpoint store (sized for T = 8, the densest
    // spacing) stays under the workspace limit
    int64_t Bs = B, Ss = S;
    if (want_grad) {
        const int64_t per_seq = ((h->L + 7) / 8) * K * (int64_t)rs;
        int64_t max_seq = std::max<int64_t>(1, h->ws_limit / std::max<int64_t>(per_seq, 1));
        if (B * S > max_seq) {
            if (max_seq >= S) {
                Bs = max_seq / S;
            } else {
                Bs = 1;
                Ss = max_seq;
            }
        }
        const int64_t slab = Bs * Ss;
        int rc;
        if ((rc = h->ckpt.ensure((size_t)slab * per_seq)) != PHK_OK) return rc;
        if ((rc = h->aux.ensure((size_t)slab * sizeof(phk::SeqAux))) != PHK_OK) return rc;
        if (!h->dbl && (rc = h->gacc.ensure((size_t)slab * 6 * K * sizeof(double))) != PHK_OK) return rc;
    }

    // kernel variant for this launch shape: forced, tuned earlier, tuned now, or the static rule
    const int64_t nseq_launch = std::min(Bs, B) * std::min(Ss, S);
    if (h->autotune && !h->force_R && !h->force_T && h->L >= 512 && nseq_launch >= 64 &&
        !h->tuned.count({nseq_launch, want_grad ? 1 : 0})) {
        phk::KArgs a;
        a.packed = h->packed;
        a.Lw = h->Lw;
        a.Ltot = h->L;
        a.W = W;
        a.inds = inds;
        a.params = params;
        a.pstride_b = pstride_b;
        a.pstride_s = pstride_s;
        a.B = std::min(Bs, B);
        a.S = std::min(Ss, S);
        a.ll = nullptr;
        a.ckpt = want_grad ? h->ckpt.p : nullptr;
        a.aux = (phk::SeqAux*)h->aux.p;
        a.grad = nullptr;
        a.gacc = (double*)h->gacc.p;
        a.grad_dlog = grad_dlog;
        int rc = autotune_variant(h, a, want_grad, lf, lb, st);
        if (rc != PHK_OK) return rc;
    }
    int R = 1, T = 8;
    choose_variant(h, nseq_launch, &R, &T, want_grad ? 1 : 0);
    if (!valid_R(K, R)) return fail(PHK_EINVAL, "no valid lanes-per-sequence for K=%d", K);
    if (!valid_T(K, R, T)) return fail(PHK_EINVAL, "R=%d T=%d not available for K=%d", R, T, K);
    h->last_total = B * S;
    h->last_R = R;
    h->last_T = T;

    for (int64_t b0 = 0; b0 < B; b0 += Bs) {
        const int64_t nb = std::min(Bs, B - b0);
        for (int64_t s0 = 0; s0 < S; s0 += Ss) {
            const int64_t ns = std::min(Ss, S - s0);
            phk::KArgs a;
            a.packed = h->packed;
            a.Lw = h->Lw;
            a.Ltot = h->L;
            a.W = W;
            a.inds = inds + s0;
            a.params = (const char*)params + (size_t)(b0 * pstride_b + s0 * pstride_s) * rs;
            a.pstride_b = pstride_b;
            a.pstride_s = pstride_s;
            a.B = nb;
            a.S = ns;
            // with Ss < S the slab is one particle (nb == 1): rows b0*S + s0 .. are contiguous
            a.ll = ll + b0 * S + s0;
            a.ckpt = want_grad ? h->ckpt.p : nullptr;
            a.aux = (phk::SeqAux*)h->aux.p;
            a.grad = want_grad ? (char*)grad + (size_t)(b0 * S + s0) * 7 * K * rs : nullptr;
            a.gacc = (double*)h->gacc.p;
            a.grad_dlog = grad_dlog;

            if (want_grad && !h->dbl)
                HIP_TRY(hipMemsetAsync(h->gacc.p, 0, (size_t)nb * ns * 6 * K * sizeof(double), st));
            hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
            if (h->profiling) {
                while ((int)h->ev.size() < 3 * (h->n_launches + 1)) {
                    hipEvent_t e;
                    HIP_TRY(hipEventCreate(&e));
                    h->ev.push_back(e);
                }
                e0 = h->ev[3 * h->n_launches];
                e1 = h->ev[3 * h->n_launches + 1];
                e2 = h->ev[3 * h->n_launches + 2];
                HIP_TRY(hipEventRecord(e0, st));
            }
            hipError_t e = lf(R, T, h->nrm, want_grad, a, nt_f, st);
            if (e != hipSuccess) return fail(PHK_EHIP, "forward kernel launch (K=%d R=%d T=%d): %s", K, R, T, hipGetErrorString(e));
            if (h->profiling) HIP_TRY(hipEventRecord(e1, st));
            if (want_grad) {
                e = lb(R, T, h->nrm, a, nt_b, st);
                if (e != hipSuccess) return fail(PHK_EHIP, "backward kernel launch (K=%d R=%d T=%d): %s", K, R, T, hipGetErrorString(e));
            }
            if (h->profiling) {
                HIP_TRY(hipEventRecord(e2, st));
                h->n_launches++;
                h->n_last++;
            }
        }
    }
    return PHK_OK;
}

}  // extern "C"
