// C ABI of libphlash_hip.so (declared in include/phlash_hip.h): handle life cycle, upload-time
// re-pack, scratch management, variant selection and the two-kernel launch sequence.
// Host side of what the reference does in _PSMCKernelBase (src/phlash/gpu.py:101-325), minus the
// host<->device copies per call: every per-call buffer here is a device pointer.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <map>
#include <new>
#include <vector>

#include "../../include/phlash_hip.h"
#define PHK_WITH_PACK 1
#include "psmc_kernels.hip"
#include "step_args.h"

namespace phk {
#define PHK_DECL(tag)                                                                                                  \
    hipError_t launch_fwd_##tag(int R, int T, int nrm, bool ckpt, const KArgs& a, int nt, hipStream_t st);              \
    hipError_t launch_bwd_##tag(int R, int T, int nrm, const KArgs& a, int units, int nt, hipStream_t st);              \
    hipError_t launch_bscan_##tag(int R, int nrm, const KArgs& a, int64_t seg_sites, void* bseg, int32_t* fseg, int nt, \
                                  hipStream_t st);                                                                      \
    hipError_t launch_finalize_##tag(const KArgs& a, int units, hipStream_t st);
PHK_DECL(f32_4) PHK_DECL(f32_8) PHK_DECL(f32_16) PHK_DECL(f32_32) PHK_DECL(f32_64)
PHK_DECL(f64_4) PHK_DECL(f64_8) PHK_DECL(f64_16) PHK_DECL(f64_32) PHK_DECL(f64_64)
#undef PHK_DECL

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

constexpr int DEFAULT_NRM = 4;   // rescale every 4th site unless told otherwise (measured: 1 -> 5.0e10, 2 -> 5.3e10, 4 -> 5.4e10; same parity)
// sites per segment of the segmented backward: 512 (64 blocks at T = 8, 32 at T = 16).  Builds with -DPHK_DEV_OVERRIDES
// (experiments only) read PHK_SEG_SITES from the environment once, when the library is loaded, and say so on stderr; the
// shipped library ignores the variable: a stray one must not change plans or the bits of a "deterministic" run.
static int seg_sites_from_env() {
#ifdef PHK_DEV_OVERRIDES
    const char* e = std::getenv("PHK_SEG_SITES");
    const int v = e ? std::atoi(e) : 0;
    if (v >= 64 && v % 64 == 0 && v <= 65536) {
        std::fprintf(stderr, "phk: developer override PHK_SEG_SITES=%d is active\n", v);
        return v;
    }
#endif
    return 512;
}
static const int SEG_SITES = seg_sites_from_env();
// threads per workgroup of the forward kernel (developer builds: PHK_FWD_NT = 64 | 128 | 256; smaller workgroups lose 55 %:
// their waves are packed onto one SIMD, profiles/r05_ab_experiments.txt item 18)
static int fwd_nt_from_env() {
#ifdef PHK_DEV_OVERRIDES
    const char* e = std::getenv("PHK_FWD_NT");
    const int v = e ? std::atoi(e) : 0;
    if (v == 64 || v == 128 || v == 256) {
        std::fprintf(stderr, "phk: developer override PHK_FWD_NT=%d is active\n", v);
        return v;
    }
#endif
    return 256;
}
static const int FWD_NT = fwd_nt_from_env();
inline int seg_blocks(int T) { return SEG_SITES / T; }
constexpr int TUNE_SITES = 2048;  // sites of the real batch the variant tuner times
constexpr double DENSE_SCAN_MAX_NONHOM = 0.13;  // static plan: share of het + missing sites above which the hybrid plan's beta scan is the structured one
constexpr double MASK_RUNS_MIN_SHARE = 0.005;  // share of the sites in all-missing 8-site halves above which the one-state-per-lane kernels step over such halves with one operator (fwd_kernel_mr)
constexpr double SCAN_PRIO_MIN_NONHOM = 0.045;  // share of het + missing sites above which the dense beta scan's waves outrank the forward kernel's

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

// How one launch shape is evaluated.
//   serial   : forward kernel (R1 or R, T) -> backward kernel (R, T), one sweep per sequence.  Large
//              batches.  The two kernels may be different variants: the forward kernel records the
//              exponent it took out of every block and the backward kernel reconciles its re-run.
//   segmented: forward kernel (R1, T) on the call's stream || beta scan (R2) on a second stream ->
//              backward kernel (R3, T) over independent segments -> finalize.  Small batches, where
//              a serial sweep leaves most of the chip idle.
struct Plan {
    int segmented = 0;
    int R = 2, T = 8;    // serial: backward kernel; segmented: R = R3 of the segment sweep
    int R1 = 0, R2 = 0;  // forward kernel (0: same as R) / beta scan (segmented only)
    // hybrid (serial plan only): sequences [0, hybrid_first) are swept serially by (R, T); the rest is
    // swept by segments with variant R3 (seeds from a beta scan with R2), concurrently, so that the
    // many short segment units fill the wave slots the serial sweep leaves empty (at cfg2: 1,563
    // serial waves on 2,048 slots).  0: no hybrid.
    int64_t hybrid_first = 0;
    int R3 = 0;
};

}  // namespace

struct phk_handle {
    int K = 0, device = 0, dbl = 0;
    int64_t N = 0, L = 0, Lw = 0;
    uint32_t* packed = nullptr;
    DevBuf ckpt, aux, gacc, eblk, eseg, bseg, fseg, bpi, part, tune_ll, tune_grad, risk, ops;
    int64_t ws_limit = 0;
    int force_R = 0, force_T = 0, nrm = DEFAULT_NRM;
    int mode = -1;     // -1 auto, 0 serial, 1 segmented
    int has_forced_plan = 0;  // phk_set_plan: overrides everything above
    Plan forced_plan;
    int autotune = 1;  // time the candidate plans once per launch shape
    int deterministic = 0;  // plan from the static rule only (never from a timing): same inputs -> same bits
    hipStream_t last_stream = nullptr;  // stream of the last phk_loglik (the flag word is read behind it)
    std::map<std::pair<int64_t, int>, Plan> tuned;  // (sequences per launch, grad?) -> plan
    int64_t last_Bs = 0, last_Ss = 0;  // particle x chunk slab of the last phk_loglik
    int64_t last_total = -1;  // B*S of the last phk_loglik and the plan it ran with
    Plan last_plan;
    hipStream_t side = nullptr;  // second stream of the segmented plan
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_fwd = nullptr;
    int profiling = 0;
    double nonhom_frac = 0.0;  // share of the observation matrix that is het or missing (counted by pack_kernel)
    int mask_runs = 0;  // the rows hold runs of missing sites (pack_kernel: all-missing 8-site halves): KArgs::mask_runs
    int asm_run = 0;  // phk_set_asm_run (developer builds with -DPHK_ASM_RUN=1)
    int budget_num[3] = {1, 1, 1}, budget_den[3] = {1, 1, 1};  // phk_set_loop_budget_scale (tests): forward kernel, backward kernel, beta scan
    int poison = 0;  // diagnostic: fill the scratch buffers with this byte before every launch sequence (PHK_POISON=255: NaN patterns)
    std::vector<hipEvent_t> ev;  // triples (start, mid, end) per launch since the last timing query
    int n_launches = 0;          // launches recorded since the last query
    int n_last = 0;              // ... of which by the last call
};

namespace {

typedef hipError_t (*fwd_fn)(int, int, int, bool, const phk::KArgs&, int, hipStream_t);
typedef hipError_t (*bwd_fn)(int, int, int, const phk::KArgs&, int, int, hipStream_t);
typedef hipError_t (*bscan_fn)(int, int, const phk::KArgs&, int64_t, void*, int32_t*, int, hipStream_t);
typedef hipError_t (*fin_fn)(const phk::KArgs&, int, hipStream_t);
struct Launchers {
    fwd_fn fwd = nullptr;
    bwd_fn bwd = nullptr;
    bscan_fn bscan = nullptr;
    fin_fn fin = nullptr;
};

bool pick_launchers(const phk_handle* h, Launchers* l) {
#define PHK_CASE(k)                                                                    \
    case k:                                                                            \
        l->fwd = h->dbl ? phk::launch_fwd_f64_##k : phk::launch_fwd_f32_##k;           \
        l->bwd = h->dbl ? phk::launch_bwd_f64_##k : phk::launch_bwd_f32_##k;           \
        l->bscan = h->dbl ? phk::launch_bscan_f64_##k : phk::launch_bscan_f32_##k;     \
        l->fin = h->dbl ? phk::launch_finalize_f64_##k : phk::launch_finalize_f32_##k; \
        return true;
    switch (h->K) {
        PHK_CASE(4) PHK_CASE(8) PHK_CASE(16) PHK_CASE(32) PHK_CASE(64)
    }
#undef PHK_CASE
    return false;
}

bool valid_R(int K, int R) { return R >= 1 && R <= 16 && (R & (R - 1)) == 0 && R <= K && K % R == 0 && K / R <= 16; }
// Variants that are not compiled (launch.hip mirrors these rules).  float64 kernels whose lanes own many
// states need more than 256 registers; hipcc (ROCm 7.2.0) then keeps part of the state in AGPR copies AND
// in scratch, and in that regime it has produced wrong code: fwd_kernel<double, 64, 4, 8, 2, true> reloaded
// only the low half of a split 64-bit spill ("A register-allocation miscompile").
// Kernels that combine AGPR copies with scratch are therefore not shipped (tests/test_layout.py checks
// the build's resource log): float64 forward / scan variants with 16 states per lane beyond K = 16, and
// float64 sweeps with more than 4 states per lane.
#ifdef PHK_EXP_F64_SPL16  // diagnostic builds only (scripts/diag_fenced_variants.py)
bool valid_Rf(const phk_handle* h, int R) { return valid_R(h->K, R); }
#else
bool valid_Rf(const phk_handle* h, int R) { return valid_R(h->K, R) && (!h->dbl || h->K / R <= 8 || h->K == 16); }
#endif
// The backward kernel keeps T + 1 state vectors of K/R reals per lane in registers.  In float64 with
// 16 states per lane that is 288 VGPRs for them alone (1.6 KB of scratch on top), with 8 states per lane
// 256 VGPRs + 256 AGPR copies + scratch: never the fastest, and in the regime described above.
#ifdef PHK_EXP_F64_SPL16  // diagnostic builds only (scripts/diag_fenced_variants.py)
bool valid_Rb(const phk_handle* h, int R) { return valid_R(h->K, R); }
#else
bool valid_Rb(const phk_handle* h, int R) { return valid_R(h->K, R) && (!h->dbl || h->K / R <= 4); }
#endif
// ... as the segment sweep (SEG = true instantiations: 8 states per lane in float64 at K = 16 only, see launch.hip)
#ifdef PHK_EXP_F64_SPL16
bool valid_Rs(const phk_handle* h, int R) { return valid_R(h->K, R); }
#else
bool valid_Rs(const phk_handle* h, int R) { return valid_R(h->K, R) && (!h->dbl || h->K / R <= 4 || (h->K == 16 && h->K / R == 8)); }
#endif
// T = 16 keeps 17 alpha vectors in registers: only compiled where a lane owns <= 4 states
bool valid_T(int K, int R, int T) { return T == 8 || (T == 16 && K / R <= 4); }
size_t real_size(const phk_handle* h) { return h->dbl ? 8 : 4; }

// smallest R (fewest instructions per site.particle) that still gives every SIMD `waves` waves
int throughput_R(const phk_handle* h, int64_t units, int T, int waves) {
    int best = 0;
    for (int c = 1; c <= 16; c <<= 1) {
        if (!valid_Rb(h, c) || !valid_T(h->K, c, T)) continue;
        best = c;
        if (units * c / 64 >= 1024 * (int64_t)waves) break;
    }
    return best ? best : 1;
}
int largest_R(const phk_handle* h, int T) {
    int best = 1;
    for (int c = 1; c <= 16; c <<= 1)
        if (valid_Rf(h, c) && valid_T(h->K, c, T)) best = c;
    return best;
}

int smallest_scan_R(const phk_handle* h) {  // fewest lanes per sequence (>= 2) a forward kernel / beta scan exists for
    for (int c = 2; c <= 16; c <<= 1)
        if (valid_Rf(h, c)) return c;
    return 0;
}

int64_t n_units(const phk_handle* h, int T, int64_t W) {
    const int64_t nblk = (h->L + T - 1) / T;
    const int64_t nseg = (nblk + seg_blocks(T) - 1) / seg_blocks(T);
    const int64_t segW = W > 0 ? ((W - 1) / T) / seg_blocks(T) : 0;
    return nseg - segW;
}

// The one-state-per-lane beta scan with its dense steps (K = 16, float32, rescale interval 4) pays inside the hybrid
// plan only if the four sequences of a wave read the SAME chunk (the codes are then scalars and every run of sites one
// dense step).  Sequences are stored chunk-major (psmc_kernels.hip, SeqMap), so that holds for a segment-swept range of
// whole chunks: a hybrid plan that asks for R2 = 16 gets its split rounded down to a multiple of B -- unless that takes
// more than 2 % of the serial sweep's sequences away (then R2 = 2, the structured scan, with the split as it was).
bool dense_scan_ok(const phk_handle* h) { return h->K == 16 && !h->dbl && h->nrm == 4 && valid_Rf(h, 16); }
Plan adjust_hybrid(const phk_handle* h, Plan p, int64_t B) {
    if (p.segmented || p.hybrid_first <= 0 || p.R2 != 16) return p;
    const int64_t rect = B > 0 ? (p.hybrid_first / B) * B : 0;
    if (dense_scan_ok(h) && rect > 0 && (p.hybrid_first - rect) * 50 <= p.hybrid_first) p.hybrid_first = rect;
    else p.R2 = valid_Rf(h, 2) ? 2 : p.R2;
    return p;
}

// the plan when nothing was forced or tuned
Plan static_plan(const phk_handle* h, int64_t nseq, int64_t W) {
    Plan p;
    p.T = h->force_T ? h->force_T : 8;
    p.R = h->force_R ? h->force_R : throughput_R(h, nseq, p.T, 1);
    // float32: never 16 states per lane in the sweep -- the beta-first body (block partly in LDS) exists for 8 states
    // per lane and below and is the fastest per site.particle however large the batch (the tuner picks 8 per lane at
    // 50,000 and at 500,000 sequences alike, cfg2 / cfg3, and at K = 32, cfg5; with 16 per lane the static plan ran
    // cfg4 at 289 ms and cfg5 at 551 against the tuner's 169 and 371)
    if (!h->force_R && !h->dbl && h->K / p.R == 16 && valid_Rb(h, 2 * p.R)) p.R *= 2;
    const int64_t units = n_units(h, 8, W);
    // small batch: even with the most lanes per sequence a serial sweep would not give every SIMD four
    // waves -- the sweep is then a chain of L dependent sites per wave on a mostly idle chip
    const bool small = nseq * largest_R(h, 8) / 64 < 4096 && units >= 8;
    p.segmented = h->mode == 1 || (h->mode < 0 && small && !h->force_R && !h->force_T);
    if (p.segmented) {
        // what the tuner picks for such shapes here: the latency-bound forward kernel and beta scan with
        // the most lanes per sequence (K = 16 float32: the dense hom-run kernels), the segment sweep with
        // <= 4 states per lane, where 16-site blocks fit in registers (half the checkpoint traffic)
        p.T = 8;
        p.R1 = p.R2 = h->force_R ? h->force_R : largest_R(h, 8);
        p.R = h->force_R ? h->force_R : throughput_R(h, nseq * units, 8, 2);
        if (!h->force_R && !h->force_T) {
            int Rs = std::max(1, h->K / 4);
            while (Rs > 1 && !valid_Rb(h, Rs)) Rs >>= 1;
            if (valid_Rb(h, Rs) && valid_T(h->K, Rs, 16) && valid_T(h->K, p.R1, 16)) {
                p.R = Rs;
                p.T = 16;
            }
            // ... unless the batch gives the 8-states-per-lane float32 sweep (beta-first body, block partly in LDS:
            // the fewest instructions per site) a chip full of units: then that sweep on 8-site blocks, which is what
            // the tuner picks at the reference's production shape (500 x 5 x 100,000: sweep 2.2 ms against 3.7 with
            // R = 4, T = 16; profiles/r04_ab_experiments.txt)
            if (h->K == 16 && !h->dbl && valid_Rs(h, 2) && nseq * units * 2 / 64 >= 2048) {
                p.R = 2;
                p.T = 8;
            }
        }
        return p;
    }
    if (!h->force_R && !h->force_T && h->mode < 0) {
        // What the tuner finds on this hardware, as a rule (the deterministic mode runs on it): the
        // forward kernel with half the lanes per sequence of the sweep (its waves are alone on their
        // SIMDs either way, and fewer lanes mean fewer instructions per site), and the sequences
        // beyond whole rounds of 1,024 waves swept by segments (hybrid, see Plan).
        if (p.R >= 2 && valid_Rf(h, p.R / 2) && nseq * (p.R / 2) / 64 >= 512) p.R1 = p.R / 2;
        const int64_t per_round = 1024 * (int64_t)(64 / p.R);
        const int64_t first = (nseq / per_round) * per_round;
        // segment sweep: the serial sweep's own variant where that is the 8-states-per-lane float32 kernel (the folded
        // beta-first body: fewest instructions per site, no scratch in its block loop), else 4 lanes per sequence;
        // beta scan: the dense one-state-per-lane scan where it exists (K = 16), else the fewest lanes per sequence
        // the state count allows (2 up to K = 32, 4 at K = 64: round 5 -- the rule used to ask for 2 and K = 64 never
        // got a hybrid plan, 157 ms at cfg4 against the tuner's 146)
        const int Rseg = (!h->dbl && h->K / p.R == 8 && valid_Rs(h, p.R)) ? p.R : 4;
        // ... unless the rows are very dense in het / missing sites: the dense scan's time grows with them (6.6 ms at 2 % non-hom
        // sites, 12-14 at 11 % at cfg2, where the forward kernel beside it takes 9-10) while the structured scan's does not (13).
        // Once the dense scan outlasts the forward kernel its 4,328 waves hold the wave slots the sweeps want: a cliff between 6 %
        // and 8 % non-hom sites (cfg2: 30.5 -> 36.9 ms per step) that raising the scan's wave priority (scan_prio_for) moves out
        // to where the structured scan is the faster one anyway -- 16 % (profiles/r06_ab_experiments.txt item 10).
        const bool dense_rows = h->nonhom_frac > DENSE_SCAN_MAX_NONHOM;
        const int Rscan = (dense_scan_ok(h) && !dense_rows) ? 16 : smallest_scan_R(h);
        if (p.T == 8 && h->L >= 8192 && units >= 8 && first > 0 && nseq - first > per_round / 20 && valid_Rs(h, Rseg) &&
            valid_T(h->K, Rseg, 8) && Rscan > 0) {
            p.hybrid_first = first;
            p.R3 = Rseg;
            p.R2 = Rscan;  // (16: see adjust_hybrid, applied once the launch shape is known)
        }
    }
    return p;
}

Plan choose_plan(const phk_handle* h, int64_t nseq, int64_t W, int want_grad) {
    if (const char* env = std::getenv("PHK_HYBRID")) {  // developer override: "R:Rf:first:R3:R2" (the plan is reported by phk_get_plan*)
        Plan p;
        long long first = 0;
        if (want_grad && std::sscanf(env, "%d:%d:%lld:%d:%d", &p.R, &p.R1, &first, &p.R3, &p.R2) == 5) {
            static bool said = false;
            if (!said) std::fprintf(stderr, "phk: developer override PHK_HYBRID=%s is active\n", env);
            said = true;
            p.hybrid_first = first;
            return p;
        }
    }
    if (h->has_forced_plan) {
        Plan p = h->forced_plan;
        if (!want_grad) {
            p.R = p.segmented ? p.R1 : p.R;
            p.segmented = 0;
        }
        return p;
    }
    if (!h->force_R && !h->force_T && h->mode < 0 && !h->deterministic) {
        auto it = h->tuned.find({nseq, want_grad});
        if (it != h->tuned.end()) return it->second;
    }
    Plan p = static_plan(h, nseq, W);
    if (!want_grad) {
        p.segmented = 0;
        p.hybrid_first = 0;
    }
    return p;
}

// The one-state-per-lane kernels (K = 16, float32, rescale interval 4) take their dense hom-run operators from a table
// built once per launch sequence, one [2 forms][10 operators][16][16] block per parameter block of the launch (20 KB; round 6: + the missing-run operator A^8);
// one per particle when the chunks share it, one per (particle, chunk) otherwise).
bool dense_capable(const phk_handle* h) { return h->K == 16 && !h->dbl && h->nrm == 4; }
bool plan_uses_dense(const phk_handle* h, const Plan& p) { return dense_capable(h) && (p.R == 16 || p.R1 == 16 || p.R2 == 16); }
int64_t param_blocks(const phk::KArgs& a) { return a.B * (a.pstride_s != 0 ? a.S : 1); }
int build_dense_ops(phk_handle* h, phk::KArgs* a, hipStream_t st) {
    const int64_t nblk = param_blocks(*a);
    if (int rc = h->ops.ensure((size_t)nblk * 2 * phk::DENSE_OPS_FLOATS * sizeof(float)); rc != PHK_OK) return rc;
    float* f = (float*)h->ops.p;
    float* b = f + nblk * phk::DENSE_OPS_FLOATS;
    hipLaunchKernelGGL(phk::dense_ops_kernel, dim3((unsigned)nblk), dim3(256), 0, st, (const float*)a->params, a->pstride_b, a->pstride_s,
                       a->pstride_s != 0 ? a->S : (int64_t)1, f, b, a->prefold, a->pfstride_b, a->pfstride_s);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(PHK_EHIP, "dense operator table launch: %s", hipGetErrorString(e));
    a->ops_f = f;
    a->ops_b = b;
    return PHK_OK;
}

// scratch shared by both plans, sized for one launch of `nseq` sequences
int ensure_scratch(phk_handle* h, int64_t nseq) {
    const int K = h->K;
    const size_t rs = real_size(h);
    const int64_t nblk8 = (h->L + 7) / 8;
    const int64_t nsegp = (nblk8 + seg_blocks(8) - 1) / seg_blocks(8) + 1;
    int rc;
    if ((rc = h->ckpt.ensure((size_t)nseq * nblk8 * K * rs)) != PHK_OK) return rc;
    if ((rc = h->aux.ensure((size_t)nseq * sizeof(phk::SeqAux))) != PHK_OK) return rc;
    if ((rc = h->gacc.ensure((size_t)nseq * 6 * K * sizeof(double))) != PHK_OK) return rc;
    if ((rc = h->eblk.ensure((size_t)nseq * nblk8 * sizeof(int16_t))) != PHK_OK) return rc;
    if ((rc = h->eseg.ensure((size_t)nseq * nsegp * sizeof(int32_t))) != PHK_OK) return rc;
    if ((rc = h->fseg.ensure((size_t)nseq * nsegp * sizeof(int32_t))) != PHK_OK) return rc;
    if ((rc = h->bseg.ensure((size_t)nseq * nsegp * K * rs)) != PHK_OK) return rc;
    if ((rc = h->bpi.ensure((size_t)nseq * K * sizeof(double))) != PHK_OK) return rc;
    return PHK_OK;
}

// partial sums of the segment sweep: one slot of [6, K] reals per (unit >= 1, sequence of the swept range)
int ensure_part(phk_handle* h, int64_t nloc, int64_t units) {
    return h->part.ensure((size_t)std::max<int64_t>(units - 1, 1) * (size_t)nloc * 6 * h->K * real_size(h));
}

// s_setprio of the beta scan's waves (KArgs::scan_prio).  They share SIMDs with the forward kernel's, which run at PHK_FWD_PRIO = 1
// because on rows like the headline's (2 % non-hom sites) the forward kernel is the longer of the two (9.1 ms against 6.6).  The dense
// scan's time grows with the non-hom sites, and from about 7 % on it is the one that ends last -- and holds the segment sweep up.
// Measured at cfg2, ms per step, dense scan at priority 0 / 2 and the structured scan (priority 0; at 2 it starves the forward
// kernel: +2 ms): 2 % hets 29.4 / 29.7 / 31.4, 5 % 30.5 / 30.7 / 31.7, 7 % 36.9 / 30.7 / 31.8, 10 % 37.2 / 31.9 / 32.2,
// 15 % 39.5 / 33.0 / 33.0, 20 % 41.5 / 36.8 / 33.1 (profiles/r06_ab_experiments.txt item 10).
int scan_prio_for(const phk_handle* h, const Plan& plan) {
    if (const char* env = std::getenv("PHK_SCAN_PRIO")) {  // developer override "dense:structured" for A/B runs
        int d = 0, s = 0;
        if (std::sscanf(env, "%d:%d", &d, &s) == 2) return plan.R2 == 16 ? d : s;
    }
    return (plan.R2 == 16 && !plan.segmented && h->nonhom_frac > SCAN_PRIO_MIN_NONHOM) ? 2 : 0;
}

// Enqueue one evaluation of `a` (ll, grad and scratch pointers set by the caller) under `plan`.
// e_mid, if given, is recorded after the forward kernel.
int enqueue(phk_handle* h, const Launchers& l, phk::KArgs a, const Plan& plan, bool want_grad, hipStream_t st,
            hipEvent_t e_mid) {
    const int K = h->K;
    const int nt = 256;
    const int64_t nseq = a.B * a.S;
    hipError_t e;
    a.seg_blocks = seg_blocks(plan.T);
    {   // Iteration budgets (KArgs::loop_budget): twice what a healthy wave needs, from the row length alone.  The forward kernel
        // and the beta scan make one outer iteration per 64-site piece (the ragged pieces at the warm-up boundary and the row's
        // end a few more: one per block / word), a serial sweep one per block, a unit of the segment sweep one per block of its
        // segments (unit 0: every segment up to the one holding the warm-up boundary).
        const int64_t nblk = (h->L + plan.T - 1) / plan.T, npieces = h->Lw / 4, G = a.seg_blocks;
        const int64_t segW = a.W > 0 ? ((a.W - 1) / plan.T) / G : 0;
        const int64_t need[4] = {npieces + 2 * (64 / plan.T) + 8, nblk + 8, npieces + 16, G * (segW + 1) + 8};
        const int scale_of[4] = {0, 1, 2, 1};
        for (int i = 0; i < 4; ++i)
            a.loop_budget[i] = (int32_t)std::min<int64_t>(2 * need[i] * h->budget_num[scale_of[i]] / h->budget_den[scale_of[i]], INT32_MAX);
    }
    if (!want_grad) {
        e = l.fwd(plan.R1 && !plan.segmented ? plan.R1 : plan.R, plan.T, h->nrm, false, a, nt, st);
        if (e != hipSuccess) return fail(PHK_EHIP, "forward kernel launch (K=%d R=%d T=%d): %s", K, plan.R, plan.T, hipGetErrorString(e));
        if (e_mid) HIP_TRY(hipEventRecord(e_mid, st));
        return PHK_OK;
    }
    if (!h->dbl || plan.segmented) HIP_TRY(hipMemsetAsync(h->gacc.p, 0, (size_t)nseq * 6 * K * sizeof(double), st));
    if (!plan.segmented && plan.hybrid_first > 0 && plan.hybrid_first < nseq && n_units(h, plan.T, a.W) >= 2) {
        // hybrid: forward kernel over everything || beta scan over the tail range; then the serial
        // sweep of the head range || the segment sweep + finalize of the tail range
        phk::KArgs a1 = a, a2 = a;
        a1.seq_begin = 0;
        a1.seq_end = plan.hybrid_first;
        a2.seq_begin = plan.hybrid_first;
        a2.seq_end = nseq;
        a2.scan_prio = scan_prio_for(h, plan);
        const int Rf = plan.R1 ? plan.R1 : plan.R;
        const int units = (int)n_units(h, plan.T, a.W);
        if (int rc = ensure_part(h, nseq - plan.hybrid_first, units); rc != PHK_OK) return rc;
        a2.part = h->part.p;
        HIP_TRY(hipMemsetAsync(h->gacc.p, 0, (size_t)nseq * 6 * K * sizeof(double), st));  // (float64 too: unit 0 of a segment sweep adds into it)
        HIP_TRY(hipEventRecord(h->ev_fork, st));
        HIP_TRY(hipStreamWaitEvent(h->side, h->ev_fork, 0));
        // (forward kernel first: its 784 lone waves are to be on their SIMDs before the scan's 4,376 fill the slots)
        e = l.fwd(Rf, plan.T, h->nrm, true, a, FWD_NT, st);
        if (e != hipSuccess) return fail(PHK_EHIP, "forward kernel launch (K=%d R=%d T=%d): %s", K, Rf, plan.T, hipGetErrorString(e));
        e = l.bscan(plan.R2, h->nrm, a2, (int64_t)SEG_SITES, h->bseg.p, (int32_t*)h->fseg.p, nt, h->side);
        if (e != hipSuccess) return fail(PHK_EHIP, "beta-scan kernel launch (hybrid, K=%d R=%d): %s", K, plan.R2, hipGetErrorString(e));
        if (e_mid) HIP_TRY(hipEventRecord(e_mid, st));
        HIP_TRY(hipEventRecord(h->ev_fwd, st));
        HIP_TRY(hipStreamWaitEvent(h->side, h->ev_fwd, 0));
        e = l.bwd(plan.R3, plan.T, h->nrm, a2, units, nt, h->side);
        if (e == hipSuccess) e = l.fin(a2, units, h->side);
        if (e != hipSuccess) return fail(PHK_EHIP, "segment sweep launch (hybrid, K=%d R=%d T=%d): %s", K, plan.R3, plan.T, hipGetErrorString(e));
        HIP_TRY(hipEventRecord(h->ev_join, h->side));
        e = l.bwd(plan.R, plan.T, h->nrm, a1, 0, nt, st);
        if (e != hipSuccess) return fail(PHK_EHIP, "backward kernel launch (K=%d R=%d T=%d): %s", K, plan.R, plan.T, hipGetErrorString(e));
        HIP_TRY(hipStreamWaitEvent(st, h->ev_join, 0));
        return PHK_OK;
    }
    if (!plan.segmented) {
        const int Rf = plan.R1 ? plan.R1 : plan.R;
        e = l.fwd(Rf, plan.T, h->nrm, true, a, FWD_NT, st);
        if (e != hipSuccess) return fail(PHK_EHIP, "forward kernel launch (K=%d R=%d T=%d): %s", K, Rf, plan.T, hipGetErrorString(e));
        if (e_mid) HIP_TRY(hipEventRecord(e_mid, st));
        e = l.bwd(plan.R, plan.T, h->nrm, a, 0, nt, st);
        if (e != hipSuccess) return fail(PHK_EHIP, "backward kernel launch (K=%d R=%d T=%d): %s", K, plan.R, plan.T, hipGetErrorString(e));
        return PHK_OK;
    }
    // segmented: the beta scan needs nothing from the forward kernel -> second stream
    const int64_t seg_sites = SEG_SITES;
    const int units = (int)n_units(h, plan.T, a.W);
    if (int rc = ensure_part(h, nseq, units); rc != PHK_OK) return rc;
    a.part = h->part.p;
    HIP_TRY(hipEventRecord(h->ev_fork, st));
    HIP_TRY(hipStreamWaitEvent(h->side, h->ev_fork, 0));
    e = l.fwd(plan.R1, plan.T, h->nrm, true, a, nt, st);
    if (e != hipSuccess) return fail(PHK_EHIP, "forward kernel launch (K=%d R=%d T=%d): %s", K, plan.R1, plan.T, hipGetErrorString(e));
    if (e_mid) HIP_TRY(hipEventRecord(e_mid, st));
    phk::KArgs as = a;
    as.scan_prio = scan_prio_for(h, plan);
    e = l.bscan(plan.R2, h->nrm, as, seg_sites, h->bseg.p, (int32_t*)h->fseg.p, nt, h->side);
    if (e != hipSuccess) return fail(PHK_EHIP, "beta-scan kernel launch (K=%d R=%d): %s", K, plan.R2, hipGetErrorString(e));
    HIP_TRY(hipEventRecord(h->ev_join, h->side));
    HIP_TRY(hipStreamWaitEvent(st, h->ev_join, 0));
    e = l.bwd(plan.R, plan.T, h->nrm, a, units, nt, st);
    if (e != hipSuccess) return fail(PHK_EHIP, "segment kernel launch (K=%d R=%d T=%d units=%d): %s", K, plan.R, plan.T, units, hipGetErrorString(e));
    e = l.fin(a, units, st);
    if (e != hipSuccess) return fail(PHK_EHIP, "finalize kernel launch: %s", hipGetErrorString(e));
    return PHK_OK;
}

// Time candidate plans on this very batch (scratch outputs) and remember the fastest for this
// (sequence count, gradient?) shape.  Serial candidates and the two latency-bound kernels of the
// segmented plan are timed on the first TUNE_SITES sites (their cost is linear in L); the segmented
// plan as a whole is timed once at full length because its parallelism depends on L.
int autotune(phk_handle* h, const Launchers& l, const phk::KArgs& proto, bool want_grad, hipStream_t st) {
    const int K = h->K;
    const size_t rs = real_size(h);
    const int64_t nseq = proto.B * proto.S;
    const int64_t tune_sites = std::min<int64_t>(h->L, TUNE_SITES);
    int rc;
    if ((rc = h->tune_ll.ensure((size_t)nseq * sizeof(double))) != PHK_OK) return rc;
    if (want_grad && (rc = h->tune_grad.ensure((size_t)nseq * 7 * K * rs)) != PHK_OK) return rc;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    phk::KArgs a = proto;
    a.ll = (double*)h->tune_ll.p;
    a.grad = want_grad ? h->tune_grad.p : nullptr;
    if (dense_capable(h) && (rc = build_dense_ops(h, &a, st)) != PHK_OK) return rc;
    auto timed = [&](const phk::KArgs& ka, const Plan& p, bool grad, float* ms) -> int {
        for (int rep = 0; rep < 2; ++rep) {  // rep 0 loads the code objects / warms the caches
            HIP_TRY(hipEventRecord(e0, st));
            int r = enqueue(h, l, ka, p, grad, st, nullptr);
            if (r != PHK_OK) return r;
            HIP_TRY(hipEventRecord(e1, st));
            HIP_TRY(hipEventSynchronize(e1));
            HIP_TRY(hipEventElapsedTime(ms, e0, e1));
        }
        return PHK_OK;
    };
    phk::KArgs at = a;  // truncated problem
    at.Ltot = tune_sites;
    at.W = std::min<int64_t>(proto.W, tune_sites);
    Plan best;
    float best_ms = 0.f;
    const bool verbose = std::getenv("PHK_TUNE_VERBOSE") != nullptr;  // diagnostic: the tuner's timings on stderr
    auto time_launch = [&](auto&& launch, float* ms) -> int {
        for (int rep = 0; rep < 2; ++rep) {
            HIP_TRY(hipEventRecord(e0, st));
            hipError_t e = launch();
            if (e != hipSuccess) return fail(PHK_EHIP, "autotune launch: %s", hipGetErrorString(e));
            HIP_TRY(hipEventRecord(e1, st));
            HIP_TRY(hipEventSynchronize(e1));
            HIP_TRY(hipEventElapsedTime(ms, e0, e1));
        }
        return PHK_OK;
    };
    for (int T = 8; T <= 16; T += 8) {
        at.seg_blocks = seg_blocks(T);
        // the forward and the backward kernel are timed separately (any forward variant leaves valid
        // checkpoints for any backward variant of the same T) and the fastest of each is kept
        int bf = 0, bb = 0;
        float tf = 0.f, tb = 0.f;
        float fms[5] = {0.f, 0.f, 0.f, 0.f, 0.f};  // forward variants R = 1, 2, 4, 8, 16
        for (int R = 1, ri = 0; R <= 16; R <<= 1, ++ri) {
            if (!valid_Rf(h, R) || !valid_T(K, R, T)) continue;
            float ms = 0.f;
            if ((rc = time_launch([&] { return l.fwd(R, T, h->nrm, want_grad, at, 256, st); }, &ms)) != PHK_OK) return rc;
            if (verbose) std::fprintf(stderr, "phk tune: nseq %lld forward R=%d T=%d on %lld sites: %.3f ms\n", (long long)nseq, R, T, (long long)tune_sites, ms);
            fms[ri] = ms;
            if (!bf || ms < tf) { tf = ms; bf = R; }
        }
        // near-ties go to the variant with fewer lanes per sequence (fewer instructions per site): on the truncated
        // problem the checkpoint stores still land in the caches, at full length they do not, and the variant with the
        // fewest instructions gains (K = 32, 159,000 sequences: R = 2 / 4 time 2.70 / 2.83 ms on 2,048 sites but
        // 49 / 60 ms at full length; the plain minimum picked R = 4 in most runs)
        for (int R = 1, ri = 0; R < bf; R <<= 1, ++ri) {
            if (fms[ri] > 0.f && fms[ri] <= 1.08f * tf) {
                bf = R;
                tf = fms[ri];
                break;
            }
        }
        if (want_grad) {
            float bms[5] = {0.f, 0.f, 0.f, 0.f, 0.f};  // sweep variants R = 1, 2, 4, 8, 16
            for (int R = 1, ri = 0; R <= 16; R <<= 1, ++ri) {
                if (!valid_Rb(h, R) || !valid_T(K, R, T)) continue;
                float ms = 0.f;
                if (!h->dbl) HIP_TRY(hipMemsetAsync(h->gacc.p, 0, (size_t)nseq * 6 * K * sizeof(double), st));
                if ((rc = time_launch([&] { return l.bwd(R, T, h->nrm, at, 0, 256, st); }, &ms)) != PHK_OK) return rc;
                if (verbose) std::fprintf(stderr, "phk tune: nseq %lld sweep   R=%d T=%d on %lld sites: %.3f ms\n", (long long)nseq, R, T, (long long)tune_sites, ms);
                bms[ri] = ms;
                if (!bb || ms < tb) { tb = ms; bb = R; }
            }
            // near-ties go to fewer lanes per sequence here too: at K = 64 the 8- and the 4-states-per-lane sweeps time
            // 4.10 and 4.14 ms on the tuner's 2,048 sites, on a slow box the other way round, and at full length the
            // first is 15 % faster (cfg4: 149 against 168 ms per step, profiles/r05_ab_experiments.txt item 11)
            for (int R = 1, ri = 0; R < bb; R <<= 1, ++ri) {
                if (bms[ri] > 0.f && bms[ri] <= 1.08f * tb) {
                    bb = R;
                    tb = bms[ri];
                    break;
                }
            }
        } else {
            bb = bf;
        }
        if (bf && bb && (best_ms == 0.f || tf + tb < best_ms)) {
            best_ms = tf + tb;
            best = Plan();
            best.T = T;
            best.R = bb;
            best.R1 = bf;
        }
    }
    // hybrid: the serial sweep runs whole waves of 64/R sequences, two per SIMD at most.  Whatever does
    // not fill a round of 1,024 waves either leaves wave slots empty for the whole sweep (cfg2: 1,563
    // waves on 2,048 slots) or runs as a thin extra round.  Give that remainder to the segment sweep,
    // whose many short units fill whatever slots are free, concurrently with the serial sweep of the rest.
    if (want_grad && best.T == 8 && h->L >= 4 * TUNE_SITES && n_units(h, 8, proto.W) >= 8 && valid_Rb(h, best.R)) {
        const int64_t per_round = 1024 * (int64_t)(64 / best.R);
        const int64_t first = (nseq / per_round) * per_round;
        if (first > 0 && nseq - first > per_round / 20) {
            Plan hbest_plan;
            float hbest = 0.f;
            // (segment sweep, beta scan) lanes per sequence; the static rule's choice first, and a later candidate
            // has to beat the best so far by 3 % (the candidates' single timings scatter by about 2 %)
            // (the serial sweep's own variant first where it is the 8-states-per-lane kernel: K = 32 / 64 had no such
            // candidate before round 5)
            const int Rmin = smallest_scan_R(h);
            std::vector<std::pair<int, int>> cand;
            auto add = [&](int r3, int r2) {
                if (r2 <= 0 || std::find(cand.begin(), cand.end(), std::make_pair(r3, r2)) != cand.end()) return;
                cand.push_back({r3, r2});
            };
            if (K / best.R == 8) { add(best.R, 16); add(best.R, Rmin); }
            add(2, 16); add(4, 16); add(2, Rmin); add(4, Rmin); add(4, 4);
            for (const auto& cc : cand) {
                const int c[2] = {cc.first, cc.second};
                if (!valid_Rs(h, c[0]) || !valid_T(K, c[0], 8) || !valid_Rf(h, c[1])) continue;
                if (c[1] == 16 && !dense_scan_ok(h)) continue;
                Plan hyb = best;
                hyb.hybrid_first = first;
                hyb.R3 = c[0];
                hyb.R2 = c[1];
                hyb = adjust_hybrid(h, hyb, a.B);
                if (c[1] == 16 && hyb.R2 != 16) continue;  // (the split cannot be a range of whole chunks here)
                float ms = 0.f;
                if ((rc = timed(a, hyb, true, &ms)) != PHK_OK) return rc;
                if (verbose) std::fprintf(stderr, "phk tune: nseq %lld hybrid first %lld sweep R=%d scan R=%d at full length: %.3f ms\n", (long long)nseq, (long long)hyb.hybrid_first, hyb.R3, hyb.R2, ms);
                if (hbest == 0.f || ms < 0.97f * hbest) { hbest = ms; hbest_plan = hyb; }
            }
            float full_ms = 0.f;  // the serial plan at full length, same conditions
            if ((rc = timed(a, best, true, &full_ms)) != PHK_OK) return rc;
            if (verbose) std::fprintf(stderr, "phk tune: nseq %lld serial plan at full length: %.3f ms\n", (long long)nseq, full_ms);
            if (hbest > 0.f && hbest < full_ms) best = hbest_plan;
        }
    }
    // segmented plan: only worth a look where the serial sweep cannot fill the chip
    const int64_t units = n_units(h, 8, proto.W);
    if (want_grad && h->L >= 4 * TUNE_SITES && units >= 8 && nseq * best.R / 64 < 1024) {
        const float serial_full = best_ms * (float)h->L / (float)tune_sites;
        // The forward kernel and the beta scan run concurrently and share the SIMDs: two variants
        // that are each the fastest alone (e.g. 625 + 625 one-state-per-lane waves on 1,024 SIMDs) can
        // be slower together than two variants with fewer waves.  So: rank the variants of each kernel
        // alone on the truncated problem, keep the two fastest of each, and time the whole plan at
        // full length for every pair (the segment sweep is throughput-bound and its parallelism
        // depends on L, so it is timed at full length as well).
        auto two_fastest = [&](auto&& launch_R, int T, int (&out)[2]) -> int {
            float t[2] = {0.f, 0.f};
            out[0] = out[1] = 0;
            for (int R = 1; R <= 16; R <<= 1) {
                if (!valid_Rf(h, R) || (T && !valid_T(K, R, T))) continue;
                float ms = 0.f;
                int r = time_launch([&] { return launch_R(R); }, &ms);
                if (r != PHK_OK) return r;
                if (!out[0] || ms < t[0]) {
                    out[1] = out[0]; t[1] = t[0];
                    out[0] = R; t[0] = ms;
                } else if (!out[1] || ms < t[1]) {
                    out[1] = R; t[1] = ms;
                }
            }
            return PHK_OK;
        };
        int R2s[2];
        if ((rc = two_fastest([&](int R) { return l.bscan(R, h->nrm, at, (int64_t)SEG_SITES, h->bseg.p, (int32_t*)h->fseg.p, 256, st); }, 0, R2s)) != PHK_OK) return rc;
        float seg_best = 0.f;
        Plan seg_plan;
        for (int T = 8; T <= 16; T += 8) {
            int R1s[2];
            at.seg_blocks = seg_blocks(T);
            if ((rc = two_fastest([&](int R) { return l.fwd(R, T, h->nrm, true, at, 256, st); }, T, R1s)) != PHK_OK) return rc;
            if (!R1s[0] || !R2s[0]) continue;
            // sweep variant with the individually fastest pair, then the pairs with that sweep variant
            Plan tbest;
            float tbest_ms = 0.f;
            for (int R = 1; R <= 8; R <<= 1) {
                if (!valid_Rs(h, R) || !valid_T(K, R, T)) continue;
                Plan cand;
                cand.segmented = 1;
                cand.T = T;
                cand.R = R;
                cand.R1 = R1s[0];
                cand.R2 = R2s[0];
                float seg_ms = 0.f;
                if ((rc = timed(a, cand, true, &seg_ms)) != PHK_OK) return rc;
                if (tbest_ms == 0.f || seg_ms < tbest_ms) { tbest_ms = seg_ms; tbest = cand; }
            }
            if (tbest_ms == 0.f) continue;
            for (int i1 = 0; i1 < 2; ++i1) {
                for (int i2 = 0; i2 < 2; ++i2) {
                    if ((i1 == 0 && i2 == 0) || !R1s[i1] || !R2s[i2]) continue;
                    Plan cand = tbest;
                    cand.R1 = R1s[i1];
                    cand.R2 = R2s[i2];
                    float seg_ms = 0.f;
                    if ((rc = timed(a, cand, true, &seg_ms)) != PHK_OK) return rc;
                    if (seg_ms < tbest_ms) { tbest_ms = seg_ms; tbest = cand; }
                }
            }
            if (seg_best == 0.f || tbest_ms < seg_best) {
                seg_best = tbest_ms;
                seg_plan = tbest;
            }
        }
        if (seg_best > 0.f && seg_best < serial_full) best = seg_plan;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (verbose)
        std::fprintf(stderr, "phk tune: nseq %lld grad %d -> segmented %d R %d T %d R_forward %d R_scan %d hybrid_first %lld R_sweep %d\n", (long long)nseq,
                     (int)want_grad, best.segmented, best.R, best.T, best.R1, best.R2, (long long)best.hybrid_first, best.R3);
    h->tuned[{nseq, want_grad ? 1 : 0}] = best;
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
    if (const char* env = std::getenv("PHK_DETERMINISTIC")) h->deterministic = std::atoi(env) != 0;
    if (const char* env = std::getenv("PHK_POISON")) h->poison = std::atoi(env);
#if PHK_ASM_RUN
    if (const char* env = std::getenv("PHK_ASM_RUN")) h->asm_run = std::atoi(env) != 0;
#endif
    if (h->risk.ensure(4 * sizeof(int)) != PHK_OK || hipMemset(h->risk.p, 0, 4 * sizeof(int)) != hipSuccess) {  // flag word + (kernel, sequence, block) of an overrun
        delete h;
        return fail(PHK_ENOMEM, "could not allocate the underflow flag");
    }
    // The second stream must run concurrently with the caller's: HIP maps streams onto a few hardware
    // queues in creation order, and two streams that share a queue serialise (seen with RCCL
    // initialised in the process: the side stream landed on the compute stream's queue and the hybrid /
    // segmented plans lost their overlap).  Streams of another priority level get queues of their own.
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    // LOWEST priority: what runs on the side stream (beta scan, segment sweep) is filler beside the long-lived waves of the
    // caller's stream (forward kernel, serial sweep), which have to get their wave slots first: a forward wave that finds
    // both slots of its SIMD taken by beta-scan waves starts a scan wave's lifetime late (forward phase 11.9 instead of
    // 11.0 ms at cfg2 in about every other run, profiles/r03_ab_experiments.txt item 14).  PHK_SIDE_PRIO=high: round 2's
    // choice, for A/B runs.
    const char* sp = std::getenv("PHK_SIDE_PRIO");
    const int side_prio = (sp && sp[0] == 'h') ? prio_greatest : prio_least;
    if (hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, side_prio) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_fwd, hipEventDisableTiming) != hipSuccess) {
        delete h;
        return fail(PHK_EHIP, "could not create the side stream");
    }

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
        // [0] sites that are not hom: what the static plan's choice of the beta scan depends on; [1] 8-site halves of a word that
        // are missing throughout: what decides whether the one-state-per-lane kernels run in their *_mr form (mask_runs)
        unsigned long long* d_cnt = nullptr;
        unsigned long long cnt[2] = {0, 0};
        if (hipMalloc((void**)&d_cnt, sizeof(cnt)) != hipSuccess || hipMemset(d_cnt, 0, sizeof(cnt)) != hipSuccess) d_cnt = nullptr;
        hipLaunchKernelGGL(phk::pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, dsrc, N, L, h->packed, h->Lw, d_cnt);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e == hipSuccess && d_cnt) e = hipMemcpy(cnt, d_cnt, sizeof(cnt), hipMemcpyDeviceToHost);
        if (d_cnt) (void)hipFree(d_cnt);
        if (e != hipSuccess) rc = fail(PHK_EHIP, "pack kernel: %s", hipGetErrorString(e));
        h->nonhom_frac = (double)cnt[0] / ((double)N * (double)L);
        h->mask_runs = (double)cnt[1] * 8.0 > MASK_RUNS_MIN_SHARE * (double)N * (double)L ? 1 : 0;
        if (const char* env = std::getenv("PHK_MASK_RUNS")) h->mask_runs = std::atoi(env) != 0;  // developer override, A/B runs
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
    (void)hipDeviceSynchronize();
    for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->ev_fwd) (void)hipEventDestroy(h->ev_fwd);
    if (h->side) (void)hipStreamDestroy(h->side);
    for (DevBuf* b : {&h->ckpt, &h->aux, &h->gacc, &h->eblk, &h->eseg, &h->bseg, &h->fseg, &h->bpi, &h->part, &h->tune_ll, &h->tune_grad, &h->risk, &h->ops})
        b->release();
    if (h->packed) (void)hipFree(h->packed);
    delete h;
    return PHK_OK;
}

int phk_set_variant(phk_handle* h, int R, int T) {
    if (!h) return fail(PHK_EINVAL, "handle is NULL");
    if (R != 0 && !valid_R(h->K, R)) return fail(PHK_EINVAL, "R=%d invalid for K=%d", R, h->K);
    if (R != 0 && !valid_Rb(h, R)) return fail(PHK_EINVAL, "R=%d: the float64 backward kernel needs K/R <= 4 (K=%d)", R, h->K);
    if (T != 0 && T != 8 && T != 16) return fail(PHK_EINVAL, "T must be 0, 8 or 16");
    if (R != 0 && !valid_T(h->K, R, T ? T : 8)) return fail(PHK_EINVAL, "R=%d T=%d not available (T=16 needs K/R <= 4)", R, T);
    h->force_R = R;
    h->force_T = T;
    return PHK_OK;
}

int phk_set_backward_mode(phk_handle* h, int mode) {
    if (!h) return fail(PHK_EINVAL, "handle is NULL");
    if (mode < -1 || mode > 1) return fail(PHK_EINVAL, "mode must be -1 (auto), 0 (serial) or 1 (segmented)");
    h->mode = mode;
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
    Plan p = (B * S == h->last_total) ? h->last_plan : choose_plan(h, B * S, 0, 1);
    *R = p.R;
    *T = p.T;
    return PHK_OK;
}

int phk_set_plan(phk_handle* h, int segmented, int R, int T, int R_forward, int R_scan) {
    if (!h) return fail(PHK_EINVAL, "handle is NULL");
    if (segmented < 0) {  // back to automatic
        h->has_forced_plan = 0;
        return PHK_OK;
    }
    Plan p;
    p.segmented = segmented ? 1 : 0;
    p.R = R;
    p.T = T;
    p.R1 = R_forward;
    p.R2 = R_scan;
    if (!(p.segmented ? valid_Rs(h, p.R) : valid_Rb(h, p.R)) || !valid_T(h->K, p.R, p.T))
        return fail(PHK_EINVAL, "R=%d T=%d not available for K=%d", R, p.T, h->K);
    if (p.segmented && (!valid_Rf(h, p.R1) || !valid_Rf(h, p.R2)))
        return fail(PHK_EINVAL, "segmented plan needs valid R_forward and R_scan (got %d, %d)", R_forward, R_scan);
    if (!p.segmented && p.R1 != 0 && !valid_Rf(h, p.R1)) return fail(PHK_EINVAL, "R_forward=%d not available for K=%d", R_forward, h->K);
    h->forced_plan = p;
    h->has_forced_plan = 1;
    return PHK_OK;
}

int phk_set_plan_hybrid(phk_handle* h, int64_t first, int R_sweep, int R_scan) {
    if (!h) return fail(PHK_EINVAL, "handle is NULL");
    if (!h->has_forced_plan || h->forced_plan.segmented)
        return fail(PHK_EINVAL, "phk_set_plan_hybrid extends a forced serial plan: call phk_set_plan(h, 0, R, T, R_forward, 0) first");
    if (first < 0) return fail(PHK_EINVAL, "first must be >= 0");
    if (first > 0 && (!valid_Rs(h, R_sweep) || !valid_T(h->K, R_sweep, h->forced_plan.T) || !valid_Rf(h, R_scan)))
        return fail(PHK_EINVAL, "hybrid plan needs a valid R_sweep and R_scan for K=%d (got %d, %d)", h->K, R_sweep, R_scan);
    if (first > 0 && R_scan == 16 && !dense_scan_ok(h)) return fail(PHK_EINVAL, "R_scan=16 (dense hom-run scan) needs K=16, float32, rescale interval 4");
    h->forced_plan.hybrid_first = first;
    h->forced_plan.R3 = first > 0 ? R_sweep : 0;
    h->forced_plan.R2 = first > 0 ? R_scan : 0;
    return PHK_OK;
}

int phk_get_plan(phk_handle* h, int* segmented, int* R, int* T, int* R_forward, int* R_scan) {
    if (!h) return fail(PHK_EINVAL, "handle is NULL");
    const Plan& p = h->last_plan;
    if (segmented) *segmented = p.segmented;
    if (R) *R = p.R;
    if (T) *T = p.T;
    if (R_forward) *R_forward = p.R1 ? p.R1 : p.R;
    if (R_scan) *R_scan = p.segmented ? p.R2 : 0;
    return PHK_OK;
}

int phk_get_plan_hybrid(phk_handle* h, int64_t* first, int* R_sweep, int* R_scan) {
    if (!h) return fail(PHK_EINVAL, "handle is NULL");
    const Plan& p = h->last_plan;
    const bool on = !p.segmented && p.hybrid_first > 0;
    if (first) *first = on ? p.hybrid_first : 0;
    if (R_sweep) *R_sweep = on ? p.R3 : 0;
    if (R_scan) *R_scan = on ? p.R2 : 0;
    return PHK_OK;
}

int phk_underflow_risk(phk_handle* h, int* flag) {
    if (!h || !flag) return fail(PHK_EINVAL, "NULL argument");
    HIP_TRY(hipSetDevice(h->device));
    // the flag word is written by kernels on the stream of the last phk_loglik (pool streams of
    // PyTorch are non-blocking: the null stream would not wait for them), so read it behind that stream
    int rec[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(rec, h->risk.p, sizeof(rec), hipMemcpyDeviceToHost, h->last_stream));
    HIP_TRY(hipStreamSynchronize(h->last_stream));
    const int word = rec[0];
    if (word) {
        HIP_TRY(hipMemsetAsync(h->risk.p, 0, sizeof(rec), h->last_stream));
        HIP_TRY(hipStreamSynchronize(h->last_stream));
    }
    *flag = (word & phk::FLAG_UNDERFLOW) ? 1 : 0;
    if (word & phk::FLAG_OVERRUN) {
        static const char* const names[] = {"?", "fwd_kernel", "bwd_kernel (serial sweep)", "bwd_kernel (segment sweep)", "bscan_kernel"};
        return fail(PHK_EOVERRUN, "%s ran out of its loop budget at sequence %d, block/word %d (L=%lld): the call's results are invalid",
                    names[rec[1] >= 1 && rec[1] <= 4 ? rec[1] : 0], rec[2], rec[3], (long long)h->L);
    }
    if (word & phk::FLAG_BAD_INDEX) return fail(PHK_EINVAL, "a chunk index passed to phk_loglik was outside [0, N=%lld)", (long long)h->N);
    return PHK_OK;
}

int phk_take_flags_async(phk_handle* h, double* dst, void* stream) {
    if (!h || !dst) return fail(PHK_EINVAL, "NULL argument");
    HIP_TRY(hipSetDevice(h->device));
    hipLaunchKernelGGL(phk::take_flags_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (int*)h->risk.p, dst);
    HIP_TRY(hipGetLastError());
    return PHK_OK;
}

int phk_set_asm_run(phk_handle* h, int on) {
    if (!h) return fail(PHK_EINVAL, "handle is NULL");
#if !PHK_ASM_RUN
    if (on) return fail(PHK_EUNSUPPORTED, "this library was built without the hand-written block-run sequence (-DPHK_ASM_RUN=1: developer builds)");
#endif
    h->asm_run = on ? 1 : 0;
    return PHK_OK;
}

int phk_set_loop_budget_scale(phk_handle* h, int kernels, int num, int den) {
    if (!h) return fail(PHK_EINVAL, "handle is NULL");
    if (num < 0 || den <= 0) return fail(PHK_EINVAL, "scale must be num / den with num >= 0, den > 0");
    for (int i = 0; i < 3; ++i) {
        if (kernels & (1 << i)) {
            h->budget_num[i] = num;
            h->budget_den[i] = den;
        }
    }
    return PHK_OK;
}

int phk_set_deterministic(phk_handle* h, int on) {
    if (!h) return fail(PHK_EINVAL, "handle is NULL");
    h->deterministic = on ? 1 : 0;
    return PHK_OK;
}

int phk_set_workspace_limit(phk_handle* h, int64_t bytes) {
    if (!h || bytes <= 0) return fail(PHK_EINVAL, "bad workspace limit");
    h->ws_limit = bytes;
    return PHK_OK;
}

int phk_get_slab(phk_handle* h, int64_t* particles, int64_t* chunks) {
    if (!h) return fail(PHK_EINVAL, "handle is NULL");
    if (particles) *particles = h->last_Bs;
    if (chunks) *chunks = h->last_Ss;
    return PHK_OK;
}

#ifdef PHK_DEBUG_EXPORTS  // diagnostic builds only: the checkpoint store of the last gradient call, copied to the host
int phk_debug_copy_ckpt(phk_handle* h, void* host, int64_t bytes) {
    if (!h || !host) return fail(PHK_EINVAL, "NULL argument");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(host, h->ckpt.p, (size_t)std::min<int64_t>(bytes, (int64_t)h->ckpt.cap), hipMemcpyDeviceToHost));
    return PHK_OK;
}
// ... and the beta scan's segment seeds [nseg + 1, nseq, K] reals with their exponents [nseg + 1, nseq] int32
int phk_debug_copy_seeds(phk_handle* h, void* bseg_host, int64_t bseg_bytes, void* fseg_host, int64_t fseg_bytes) {
    if (!h || !bseg_host || !fseg_host) return fail(PHK_EINVAL, "NULL argument");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(bseg_host, h->bseg.p, (size_t)std::min<int64_t>(bseg_bytes, (int64_t)h->bseg.cap), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(fseg_host, h->fseg.p, (size_t)std::min<int64_t>(fseg_bytes, (int64_t)h->fseg.cap), hipMemcpyDeviceToHost));
    return PHK_OK;
}
#endif

int64_t phk_workspace_bytes(phk_handle* h) {
    return h ? (int64_t)(h->ckpt.cap + h->aux.cap + h->gacc.cap + h->eblk.cap + h->eseg.cap + h->bseg.cap + h->fseg.cap + h->bpi.cap + h->part.cap + h->ops.cap) : 0;
}

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
    return phk_param_map_rounded(device, K, P, epoch_of_state, theta, x, B, params, jac, nullptr, stream);
}

int phk_param_map_rounded(int device, int K, int P, const int32_t* epoch_of_state, double theta, const double* x, int64_t B,
                          double* params, double* jac, float* params_f32, void* stream) {
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
    a.params_f32 = params_f32;
    HIP_TRY(hipSetDevice(device));
    hipError_t e = phk::launch_param_map(a, (hipStream_t)stream);
    if (e != hipSuccess) return fail(PHK_EHIP, "param_map kernel launch: %s", hipGetErrorString(e));
    return PHK_OK;
}

int phk_log_prior(int device, int P, double alpha, double beta, const double* x, int64_t B, double* value, double* grad,
                  void* stream) {
    if (P < 1 || P > phk::PM_MAXK) return fail(PHK_EINVAL, "P=%d epochs outside [1, %d]", P, phk::PM_MAXK);
    if (!x || !value) return fail(PHK_EINVAL, "NULL argument");
    if (B < 0) return fail(PHK_EINVAL, "B must be >= 0");
    HIP_TRY(hipSetDevice(device));
    hipError_t e = phk::launch_log_prior(P, alpha, beta, x, B, value, grad, (hipStream_t)stream);
    if (e != hipSuccess) return fail(PHK_EHIP, "log_prior kernel launch: %s", hipGetErrorString(e));
    return PHK_OK;
}

int phk_afs_term(int device, int K, int P, const int32_t* epoch_of_state, const double* x, int64_t B, int n, int m,
                 const double* tw, const double* w1, const double* y, double* value, double* grad, void* stream) {
    if (K < 3 || K > phk::PM_MAXK) return fail(PHK_EUNSUPPORTED, "K=%d outside [3, %d]", K, phk::PM_MAXK);
    if (P < 1 || P > K) return fail(PHK_EINVAL, "P=%d epochs for K=%d states", P, K);
    if (n < 3 || n > phk::AF_MAXN) return fail(PHK_EUNSUPPORTED, "n=%d samples outside [3, %d] (n = 2 has no AFS term: model.py:58-68 gives 0)", n, phk::AF_MAXN);
    if (m < 1 || m > phk::AF_MAXN) return fail(PHK_EUNSUPPORTED, "m=%d transform rows outside [1, %d]", m, phk::AF_MAXN);
    if (!epoch_of_state || !x || !tw || !w1 || !y || !value || !grad) return fail(PHK_EINVAL, "NULL argument");
    if (B < 0) return fail(PHK_EINVAL, "B must be >= 0");
    phk::AFArgs a;
    a.K = K;
    a.P = P;
    a.D = P + 3;
    for (int k = 0; k < K; ++k) {
        if (epoch_of_state[k] < 0 || epoch_of_state[k] >= P) return fail(PHK_EINVAL, "epoch_of_state[%d] = %d outside [0, %d)", k, epoch_of_state[k], P);
        a.epoch[k] = (int8_t)epoch_of_state[k];
    }
    a.n1 = n - 1;
    a.m = m;
    a.x = x;
    a.tw = tw;
    a.w1 = w1;
    a.y = y;
    a.value = value;
    a.grad = grad;
    a.B = B;
    HIP_TRY(hipSetDevice(device));
    hipError_t e = phk::launch_afs_term(a, (hipStream_t)stream);
    if (e != hipSuccess) return fail(PHK_EHIP, "afs_term kernel launch: %s", hipGetErrorString(e));
    return PHK_OK;
}

int phk_reduce_chunks(phk_handle* h, const double* ll, const void* grad, int64_t B, int64_t S, double* buf, void* stream) {
    if (!h) return fail(PHK_EINVAL, "handle is NULL");
    if (!ll || !grad || !buf) return fail(PHK_EINVAL, "NULL argument");
    if (B < 0 || S < 0) return fail(PHK_EINVAL, "B and S must be >= 0");
    HIP_TRY(hipSetDevice(h->device));
    hipError_t e = phk::launch_reduce_chunks(ll, grad, h->dbl != 0, B, S, 7 * h->K, buf, (int*)h->risk.p, (hipStream_t)stream);
    if (e != hipSuccess) return fail(PHK_EHIP, "reduce_chunks kernel launch: %s", hipGetErrorString(e));
    return PHK_OK;
}

int phk_chain_rule(int device, int K, int P, double alpha, double beta, const double* x, const double* buf, const double* jac,
                   int64_t B, double c_prior, double c_hmm, const double* extra_val, const double* extra_grad, double c_extra,
                   double* logp, double* grad, void* stream) {
    if (K < 3 || K > phk::PM_MAXK) return fail(PHK_EUNSUPPORTED, "K=%d outside [3, %d]", K, phk::PM_MAXK);
    if (P < 1 || P > K) return fail(PHK_EINVAL, "P=%d epochs for K=%d states", P, K);
    if (!x || !buf || !jac || !logp || !grad) return fail(PHK_EINVAL, "NULL argument");
    if (B < 0) return fail(PHK_EINVAL, "B must be >= 0");
    phk::CRArgs a;
    a.J = 7 * K;
    a.D = P + 3;
    a.P = P;
    a.alpha = alpha;
    a.beta = beta;
    a.c_prior = c_prior;
    a.c_hmm = c_hmm;
    a.c_extra = c_extra;
    a.x = x;
    a.buf = buf;
    a.jac = jac;
    a.extra_val = extra_val;
    a.extra_grad = extra_grad;
    a.logp = logp;
    a.grad = grad;
    HIP_TRY(hipSetDevice(device));
    hipError_t e = phk::launch_chain_rule(a, B, (hipStream_t)stream);
    if (e != hipSuccess) return fail(PHK_EHIP, "chain_rule kernel launch: %s", hipGetErrorString(e));
    return PHK_OK;
}

int64_t phk_svgd_workspace_doubles(int64_t B) { return B > 0 ? phk::svgd_ws_doubles(B) : 1; }

int phk_svgd_step(int device, int64_t B, int D, const double* x, const double* grad_logp, double* mu, double* nu,
                  double* nu_max, const double* h_in, double* h_out, double* x_out, double* dist_ws, int64_t count,
                  double lr, double b1, double b2, double eps, void* stream) {
    if (!x || !grad_logp || !mu || !nu || !nu_max || !h_in || !x_out || !dist_ws) return fail(PHK_EINVAL, "NULL argument");
    if (B < 1 || B > phk::SV_MAXB) return fail(PHK_EUNSUPPORTED, "B=%lld particles outside [1, %d]", (long long)B, phk::SV_MAXB);
    if (D < 1 || D > phk::SV_MAXD) return fail(PHK_EUNSUPPORTED, "D=%d outside [1, %d]", D, phk::SV_MAXD);
    if (count < 1) return fail(PHK_EINVAL, "count must be >= 1 (the step number, starting at 1)");
    if (x_out == x) return fail(PHK_EINVAL, "x_out must not alias x (every workgroup reads all particles)");
    phk::SVArgs a;
    a.B = B;
    a.D = D;
    a.x = x;
    a.g = grad_logp;
    a.mu = mu;
    a.nu = nu;
    a.nu_max = nu_max;
    a.h_in = h_in;
    a.x_out = x_out;
    a.den1 = 1.0 - std::pow(b1, (double)count);
    a.den2 = 1.0 - std::pow(b2, (double)count);
    a.lr = lr;
    a.b1 = b1;
    a.b2 = b2;
    a.eps = eps;
    HIP_TRY(hipSetDevice(device));
    hipError_t e = phk::launch_svgd_step(a, dist_ws, h_out, (hipStream_t)stream);
    if (e != hipSuccess) return fail(PHK_EHIP, "svgd step kernels: %s", hipGetErrorString(e));
    return PHK_OK;
}

int phk_ll_first_order(int device, int K, double* ll, const float* grad, int grad_dlog, const double* params, const double* crel,
                       int64_t pstride_b, int64_t pstride_s, int64_t B, int64_t S, void* stream) {
    if (K < 1 || K > phk::PM_MAXK) return fail(PHK_EUNSUPPORTED, "K=%d outside [1, %d]", K, phk::PM_MAXK);
    if (!ll || !grad || !params || !crel) return fail(PHK_EINVAL, "NULL argument");
    if (B < 0 || S < 0) return fail(PHK_EINVAL, "B and S must be >= 0");
    HIP_TRY(hipSetDevice(device));
    hipError_t e = phk::launch_ll_first_order(ll, grad, params, crel, pstride_b, pstride_s, B, S, K, grad_dlog, (hipStream_t)stream);
    if (e != hipSuccess) return fail(PHK_EHIP, "first-order correction kernel launch: %s", hipGetErrorString(e));
    return PHK_OK;
}

int phk_prefold(int device, int K, const double* params, int64_t nblocks, float* params_f32, float* prefold_f32, double* crel, void* stream) {
    if (K < 1 || K > phk::PM_MAXK) return fail(PHK_EUNSUPPORTED, "K=%d outside [1, %d]", K, phk::PM_MAXK);
    if (!params || !params_f32 || !prefold_f32) return fail(PHK_EINVAL, "NULL argument");
    if (nblocks < 0) return fail(PHK_EINVAL, "nblocks must be >= 0");
    if (nblocks == 0) return PHK_OK;
    HIP_TRY(hipSetDevice(device));
    hipError_t e = phk::launch_prefold(K, params, nblocks, params_f32, prefold_f32, crel, (hipStream_t)stream);
    if (e != hipSuccess) return fail(PHK_EHIP, "prefold kernel launch: %s", hipGetErrorString(e));
    return PHK_OK;
}

static int loglik_impl(phk_handle* h, const void* params, int64_t pstride_b, int64_t pstride_s, const float* prefold,
                       const int64_t* inds, int64_t B, int64_t S, int64_t W, double* ll, void* grad, int grad_dlog, void* stream);

int phk_loglik(phk_handle* h, const void* params, int64_t pstride_b, int64_t pstride_s, const int64_t* inds,
               int64_t B, int64_t S, int64_t W, double* ll, void* grad, int grad_dlog, void* stream) {
    return loglik_impl(h, params, pstride_b, pstride_s, nullptr, inds, B, S, W, ll, grad, grad_dlog, stream);
}

int phk_loglik_prefolded(phk_handle* h, const void* params, int64_t pstride_b, int64_t pstride_s, const float* prefold,
                         const int64_t* inds, int64_t B, int64_t S, int64_t W, double* ll, void* grad, int grad_dlog, void* stream) {
    if (!h) return fail(PHK_EINVAL, "handle is NULL");
    if (!prefold) return fail(PHK_EINVAL, "prefold is NULL (phk_loglik takes blocks without pre-folded factors)");
    if (h->dbl) return fail(PHK_EINVAL, "pre-folded factors belong to the float32 kernels (the float64 kernels do not fold)");
    if (pstride_b % 7 != 0 || pstride_s % 7 != 0) return fail(PHK_EINVAL, "parameter strides must be multiples of 7 (whole [7, K] blocks)");
    return loglik_impl(h, params, pstride_b, pstride_s, prefold, inds, B, S, W, ll, grad, grad_dlog, stream);
}

static int loglik_impl(phk_handle* h, const void* params, int64_t pstride_b, int64_t pstride_s, const float* prefold,
                       const int64_t* inds, int64_t B, int64_t S, int64_t W, double* ll, void* grad, int grad_dlog, void* stream) {
    if (!h) return fail(PHK_EINVAL, "handle is NULL");
    if (!params || !inds || !ll) return fail(PHK_EINVAL, "params, inds and ll must be non-NULL device pointers");
    if (B < 0 || S < 0) return fail(PHK_EINVAL, "B and S must be >= 0");
    if (W < 0 || W > h->L) return fail(PHK_EINVAL, "W=%lld outside [0, L=%lld]", (long long)W, (long long)h->L);
    if (B == 0 || S == 0) return PHK_OK;
    Launchers l;
    if (!pick_launchers(h, &l)) return fail(PHK_EUNSUPPORTED, "K=%d not compiled in", h->K);
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    h->last_stream = st;
    const size_t rs = real_size(h);
    const int K = h->K;
    const bool want_grad = grad != nullptr;
    h->n_last = 0;
    if (h->n_launches > 4096) h->n_launches = 0;  // nobody is collecting: recycle the event pool

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
        int rc = ensure_scratch(h, Bs * Ss);
        if (rc != PHK_OK) return rc;
    }

    auto make_args = [&](int64_t b0, int64_t nb, int64_t s0, int64_t ns) {
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
        a.eblk = (int16_t*)h->eblk.p;
        a.eseg = (int32_t*)h->eseg.p;
        a.seg_blocks = seg_blocks(8);  // enqueue() sets the plan's value
        a.bseg = h->bseg.p;
        a.fseg = (const int32_t*)h->fseg.p;
        a.bpi = (double*)h->bpi.p;
        a.risk = (int*)h->risk.p;
        a.seq_begin = a.seq_end = 0;
        a.N = h->N;
        a.part = h->part.p;
        a.ops_f = a.ops_b = nullptr;  // build_dense_ops sets them where the plan runs a one-state-per-lane kernel
        // the pre-folded blocks are laid out like the parameter blocks, five rows instead of seven
        for (int i = 0; i < 4; ++i) a.loop_budget[i] = INT32_MAX;  // enqueue() sets the plan's values
        a.asm_run = h->asm_run;
        a.scan_prio = 0;  // enqueue() sets the plan's value
        a.mask_runs = h->mask_runs;
        a.pfstride_b = pstride_b / 7 * 5;
        a.pfstride_s = pstride_s / 7 * 5;
        a.prefold = prefold ? prefold + (b0 * a.pfstride_b + s0 * a.pfstride_s) : nullptr;
        return a;
    };

    // plan for this launch shape: forced, tuned earlier, tuned now, or the static rule
    const int64_t nseq_launch = std::min(Bs, B) * std::min(Ss, S);
    if (h->autotune && !h->deterministic && !h->has_forced_plan && !h->force_R && !h->force_T && h->mode < 0 && h->L >= 512 && nseq_launch >= 64 &&
        !h->tuned.count({nseq_launch, want_grad ? 1 : 0})) {
        int rc = autotune(h, l, make_args(0, std::min(Bs, B), 0, std::min(Ss, S)), want_grad, st);
        if (rc != PHK_OK) return rc;
    }
    const Plan plan = adjust_hybrid(h, choose_plan(h, nseq_launch, W, want_grad ? 1 : 0), std::min(Bs, B));
    if (!(want_grad ? (plan.segmented ? valid_Rs(h, plan.R) : valid_Rb(h, plan.R)) : valid_Rf(h, plan.R)) || !valid_T(K, plan.R, plan.T))
        return fail(PHK_EINVAL, "R=%d T=%d not available for K=%d", plan.R, plan.T, K);
    if (want_grad && !plan.segmented && plan.hybrid_first > 0 && !valid_Rs(h, plan.R3)) return fail(PHK_EINVAL, "segment sweep R=%d not available for K=%d", plan.R3, K);
    if (plan.segmented && (!valid_Rf(h, plan.R1) || !valid_Rf(h, plan.R2))) return fail(PHK_EINVAL, "invalid segmented plan for K=%d", K);
    if (!plan.segmented && plan.R1 != 0 && !valid_Rf(h, plan.R1)) return fail(PHK_EINVAL, "forward variant R=%d not available for K=%d", plan.R1, K);
    if (plan_uses_dense(h, plan)) {  // sized before anything of this call is enqueued (growing it later would synchronise the device mid-step)
        const int64_t blocks = std::min(Bs, B) * (pstride_s != 0 ? std::min(Ss, S) : 1);
        if (int rc = h->ops.ensure((size_t)blocks * 2 * phk::DENSE_OPS_FLOATS * sizeof(float)); rc != PHK_OK) return rc;
    }
    h->last_total = B * S;
    h->last_plan = plan;
    h->last_Bs = std::min(Bs, B);
    h->last_Ss = std::min(Ss, S);

    for (int64_t b0 = 0; b0 < B; b0 += Bs) {
        const int64_t nb = std::min(Bs, B - b0);
        for (int64_t s0 = 0; s0 < S; s0 += Ss) {
            const int64_t ns = std::min(Ss, S - s0);
            phk::KArgs a = make_args(b0, nb, s0, ns);
            if (plan_uses_dense(h, plan)) {
                if (pstride_s == 0 && s0 > 0) {  // one block per particle: the table of this particle range is built already
                    a.ops_f = (const float*)h->ops.p;
                    a.ops_b = a.ops_f + nb * phk::DENSE_OPS_FLOATS;
                } else if (int rc = build_dense_ops(h, &a, st); rc != PHK_OK) {
                    return rc;
                }
            }
            if (h->poison) {
                // diagnostic (environment PHK_POISON=<byte>, e.g. 255 = NaN patterns): fill every scratch buffer with that byte before every
                // launch sequence, so that a kernel reading what no kernel of this launch sequence wrote shows up as
                // NaN whatever ran before (an earlier call, an earlier slab of this call)
                int bit = 0;  // PHK_POISON_MASK (default all): bit i selects the i-th buffer of this list
                const char* menv = std::getenv("PHK_POISON_MASK");
                const int mask = menv ? std::atoi(menv) : 0x1ff;
                for (DevBuf* b : {&h->ckpt, &h->aux, &h->eblk, &h->eseg, &h->bseg, &h->fseg, &h->bpi, &h->part, &h->gacc}) {
                    if (b->p && ((mask >> bit) & 1)) HIP_TRY(hipMemsetAsync(b->p, h->poison & 0xFF, b->cap, st));
                    ++bit;
                }
            }
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
            int rc = enqueue(h, l, a, plan, want_grad, st, e1);
            if (rc != PHK_OK) return rc;
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
