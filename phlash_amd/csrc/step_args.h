// Argument blocks and host launchers of the kernels above the scan (param_map.hip, svgd_step.hip), shared with the
// C ABI (phk_api.hip): one definition instead of copies that "must match".
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace phk {

constexpr int PM_MAXK = 64;

struct PMArgs {
    int K, P, D;
    double theta;
    int8_t epoch[PM_MAXK];  // epoch index of every hidden state (pattern expansion, util.py:35-37)
    const double* x;        // [B, D]
    double* params;         // [B, 7, K]
    double* jac;            // [B, 7K, D] or null
    int64_t B;
    float* params_f32;      // [B, 7, K] the same block rounded to float32 (what the float32 kernels take), or null
};
hipError_t launch_param_map(const PMArgs& a, hipStream_t st);
// params [n, 7, K] float64 -> the same block rounded to float32 + the float32 kernels' pre-folded factors [n, 5, K]
// (rows fl(emis0 b), fl(emis0 d), fl(emis0 v), fl(emis1 / emis0), fl(1 / emis0): products and quotients in float64, rounded once)
hipError_t launch_prefold(int K, const double* params, int64_t n, float* params_f32, float* prefold_f32, double* crel, hipStream_t st);
// ll [B, S] += sum_j theta_j (d ll / d theta_j) crel_j: the first-order effect of the float32 rounding of the model, taken back
hipError_t launch_ll_first_order(double* ll, const float* g, const double* params, const double* crel, int64_t stride_b, int64_t stride_s,
                                 int64_t B, int64_t S, int K, int dlog, hipStream_t st);
hipError_t launch_log_prior(int P, double alpha, double beta, const double* x, int64_t B, double* value, double* grad,
                            hipStream_t st);

// the tail of a step (param_map.hip): chunk sums + flag hand-over, then prior + chain rule
hipError_t launch_reduce_chunks(const double* ll, const void* g, bool g_f64, int64_t B, int64_t S, int J, double* buf, int* flags,
                                hipStream_t st);
struct CRArgs {
    int J, D, P;
    double alpha, beta, c_prior, c_hmm, c_extra;
    const double* x;           // [B, D]
    const double* buf;         // [>= B, 1 + J]
    const double* jac;         // [B, J, D]
    const double* extra_val;   // [B] or null
    const double* extra_grad;  // [B, D] or null
    double* logp;              // [B]
    double* grad;              // [B, D]
};
hipError_t launch_chain_rule(const CRArgs& a, int64_t B, hipStream_t st);

// the AFS term of the objective (param_map.hip): value and gradient w.r.t. the particle, one launch
constexpr int AF_MAXN = 128;   // sample size n <= this (n - 1 expected branch lengths)
struct AFArgs {
    int K, P, D;
    int8_t epoch[PM_MAXK];
    int n1;               // n - 1
    int m;                // rows of the transform T
    const double* x;      // [B, D]
    const double* tw;     // [m, n1] = T W   (W: Polanski-Kimmel matrix of size_history.py:350-369)
    const double* w1;     // [n1] = column sums of W
    const double* y;      // [m] = T afs
    double* value;        // [B]
    double* grad;         // [B, D]
    int64_t B;
};
hipError_t launch_afs_term(const AFArgs& a, hipStream_t st);

constexpr int SV_MAXD = 72;    // P + 3 <= 67
constexpr int SV_MAXB = 4096;  // particles (the kernel row of one particle lives in LDS)

struct SVArgs {
    int64_t B;
    int D;
    const double* x;      // [B, D]
    const double* g;      // [B, D] grad log p
    double* mu;           // [B, D] in/out
    double* nu;
    double* nu_max;
    const double* h_in;   // device scalar
    double* x_out;        // [B, D]
    double den1, den2;    // 1 - b1^count, 1 - b2^count
    double lr, b1, b2, eps;
};
// dist_ws: svgd_ws_doubles(B) doubles, see phk_svgd_step in include/phlash_hip.h
int64_t svgd_ws_doubles(int64_t B);
hipError_t launch_svgd_step(const SVArgs& a, double* dist_ws, double* h_out, hipStream_t st);

}  // namespace phk
