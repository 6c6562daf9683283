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
//   * normalisation is by an exact power of two (v_frexp_exp, multiply by 2^-e): the scaled state
//     returns to [0.5,1) and the integer exponents are summed, so ll = E*ln2 + log(sum) takes one log per
//     sequence and loses nothing to f32 accumulation of 60,000 log terms.
//   * gradient: kernel 1 (forward) stores alpha every T sites to HBM (coalesced, K reals per
//     sequence per block); kernel 2 walks the blocks backwards: re-runs the T forward sites of a
//     block keeping every alpha in REGISTERS (both site loops fully unrolled), then sweeps them in
//     reverse accumulating the six parameter rows in registers (f32 partial sums are flushed to
//     f64 every FLUSH_SITES sites).
//   * the element-wise part of a site runs on packed pairs (v_pk_fma_f32 ...); the emission row of
//     a site is fetched from a per-thread LDS table [hom, het, ones] with one ds_read per pair
//     instead of two v_cndmask per state.
//   * rescaling may be done every NRM-th site only (NRM in {1,2,4}); NRM = 1 is the reference's
//     per-site normalisation.
//
// Numerics contract: every arithmetic step goes through explicit fma / mul / add in ONE inline
// step function used by both kernels and the file is compiled with -ffp-contract=off, so the
// re-run of a block reproduces the forward pass bit for bit (the backward sweep relies on the
// re-run ending in the same scaling as the next checkpoint).

#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>

#ifndef PHK_EXP_LAND_F64
#define PHK_EXP_LAND_F64 0  // diagnostic builds only: 1 applies the float32 kernels' piece-landing asm to float64 too
#endif
// A/B switches of scripts/ab_build.sh (defaults = the shipped behaviour)
#ifndef PHK_SERIAL_PRIO
#define PHK_SERIAL_PRIO 0  // s_setprio of the serial backward sweep's waves (0..3)
#endif
#ifndef PHK_SEG_PRIO
#define PHK_SEG_PRIO 0  // s_setprio of the segment sweep's waves (they share SIMDs with the serial sweep's in the hybrid plan)
#endif
#ifndef PHK_FWD_PRIO
#define PHK_FWD_PRIO 1  // s_setprio of the forward kernel's waves where they are the step's critical path (checkpointing, more than one state per lane): the beta scan's waves of the hybrid plan share their SIMDs and finish 3 ms earlier; cfg2 forward 9.39 -> 9.10 ms, 3 = 1 (r05_ab_experiments.txt item 17)
#endif
#ifndef PHK_FWD_BIG_PIECES
#define PHK_FWD_BIG_PIECES 0  // forward kernel with checkpoints: request observation words 256 sites at a time
#endif
#ifndef PHK_EMIS_AHEAD
#define PHK_EMIS_AHEAD 1  // forward kernel: emission rows requested this many sites ahead (1 or 2)
#endif
#ifndef PHK_SWEEP_V2
#define PHK_SWEEP_V2 1  // backward kernel, full blocks: beta pass first (storing w = e.*beta), then the forward re-run accumulates
#endif
#ifndef PHK_PARK
#define PHK_PARK 1  // PHK_SWEEP_V2 with 8 float32 states per lane: this many of a block's 8 w vectors live in LDS, not registers.
                    // Round 3 needed 3 to keep the block loop free of scratch; the folded body (round 5) has no emission rows
                    // in flight and its mass rows in LDS, and 1 is enough -- and it keeps the slice of a 256-thread workgroup
                    // under the 64 KB a launch gets without asking: with 2 or 3 the launcher falls back to 128-thread
                    // workgroups, whose two waves land on one SIMD (cfg2 backward phase 25.1 instead of 20.1 ms,
                    // profiles/r05_ab_experiments.txt item 2)
#endif
#ifndef PHK_HET_REGS
#define PHK_HET_REGS 1  // folded sweeps (round 6): the het ratio row emis1 / emis0 and the het posterior mass live in REGISTERS in the hot
                        // body (two more parked w vectors pay for them), het lanes are handled under the exec mask, the block's
                        // non-hom mask comes from two readlanes where the wave holds at most two observation rows; only missing
                        // sites still go through the LDS rows.  0 = round 5's body (every non-hom site through LDS)
#endif
#ifndef PHK_FWD_HET_REGS
#define PHK_FWD_HET_REGS 0  // A/B: 1 = forward kernels with several states per lane, folded waves: the het / missing ratio rows in registers
                           // instead of the LDS table.  Measured and not adopted (profiles/r06_ab_experiments.txt item 3): the one-lane
                           // K = 16 kernel goes from 236 registers to 255 + 14 AGPR copies, 0.37 ms slower at 1 % hets, level at 10 %
#endif
#ifndef PHK_EXP_HET_MODE
#define PHK_EXP_HET_MODE 0  // timing-only diagnostic builds of the PHK_HET_REGS body: 1 = no site is het or missing, 2 = branches taken, bodies empty
#endif
#ifndef PHK_EXP_NO_STEEP
#define PHK_EXP_NO_STEEP 0
#endif
#ifndef PHK_EXP_NO_CKPT_STORE
#define PHK_EXP_NO_CKPT_STORE 0  // timing-only diagnostic builds: the forward kernel does not store its checkpoints (results are wrong)
#endif
#ifndef PHK_DS_FIRST
#define PHK_DS_FIRST 0  // beta-first body: ask the scheduler to issue a site's LDS reads (next emission row, next parked w) before its arithmetic
#endif
#ifndef PHK_FWD_SITE_BARRIER
#define PHK_FWD_SITE_BARRIER 1  // scheduling barrier after every site of the forward kernel's straight-line block
#endif

namespace phk {

constexpr int NT_MAX = 256;        // max threads per workgroup (4 waves); the launch picks <= this
constexpr int FLUSH_SITES = 2048;  // f32 gradient partial sums are folded into f64 this often

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
// BANKS: 4-bit mask over the four 4-lane banks of a row; lanes of a disabled bank also get zero.
template <int CTRL, int BANKS = 0xf>
__device__ __forceinline__ float dpp_(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, BANKS, false));
}
template <int CTRL, int BANKS = 0xf>
__device__ __forceinline__ double dpp_(double x) {
    const uint64_t u = __builtin_bit_cast(uint64_t, x);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)u, CTRL, 0xf, BANKS, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(u >> 32), CTRL, 0xf, BANKS, false);
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
    real up1, up2, up3;  // 1.0 where rank >= n, else 0.0   (shifts by 4 and 8 are masked by DPP itself)
    real dn1, dn2, dn3;  // 1.0 where rank + n < R, else 0.0

    __device__ __forceinline__ void init(int rank) {
        up1 = rank >= 1 ? real(1) : real(0);
        up2 = rank >= 2 ? real(1) : real(0);
        up3 = rank >= 3 ? real(1) : real(0);
        dn1 = rank + 1 < R ? real(1) : real(0);
        dn2 = rank + 2 < R ? real(1) : real(0);
        dn3 = rank + 3 < R ? real(1) : real(0);
    }

    // all-reduce: every lane of the group gets the same bits (each step adds a commutative pair)
    __device__ __forceinline__ real sum(real x) const {
        if constexpr (R >= 2) x = x + dpp_<QP(1, 0, 3, 2)>(x);
        if constexpr (R >= 4) x = x + dpp_<QP(2, 3, 0, 1)>(x);
        if constexpr (R >= 8) x = x + dpp_<ROW_HALF_MIRROR>(x);
        if constexpr (R >= 16) x = x + dpp_<ROW_MIRROR>(x);
        return x;
    }
    // exclusive prefix over the group: lane r gets sum of x over lanes < r.
    //   R = 16: the group is the DPP row, row_shr zero-fills, every step is one v_add_f32_dpp;
    //   R = 8 : Hillis-Steele; the shift by 4 is masked by the DPP bank mask (lanes 4..7 = banks 1,3);
    //   R = 4 : the group is one quad: three independent quad_perm gathers, depth 3 instead of 5;
    //   R = 2 : one masked neighbour read.
    __device__ __forceinline__ real excl_prefix(real x) const {
        if constexpr (R == 1) {
            return real(0);
        } else if constexpr (R == 2) {
            return dpp_<ROW_SHR(1)>(x) * up1;
        } else if constexpr (R == 4) {
            real y = dpp_<QP(0, 0, 1, 2)>(x) * up1;
            y = fma_(dpp_<QP(0, 0, 0, 1)>(x), up2, y);
            return fma_(dpp_<QP(0, 0, 0, 0)>(x), up3, y);
        } else if constexpr (R == 8) {
            real y = dpp_<ROW_SHR(1)>(x) * up1;
            y = fma_(dpp_<ROW_SHR(1)>(y), up1, y);
            y = fma_(dpp_<ROW_SHR(2)>(y), up2, y);
            return y + dpp_<ROW_SHR(4), 0xA>(y);
        } else {
            real y = dpp_<ROW_SHR(1)>(x);
            y = y + dpp_<ROW_SHR(1)>(y);
            y = y + dpp_<ROW_SHR(2)>(y);
            y = y + dpp_<ROW_SHR(4)>(y);
            return y + dpp_<ROW_SHR(8)>(y);
        }
    }
    // exclusive suffix: lane r gets sum of x over lanes > r
    __device__ __forceinline__ real excl_suffix(real x) const {
        if constexpr (R == 1) {
            return real(0);
        } else if constexpr (R == 2) {
            return dpp_<ROW_SHL(1)>(x) * dn1;
        } else if constexpr (R == 4) {
            real y = dpp_<QP(1, 2, 3, 3)>(x) * dn1;
            y = fma_(dpp_<QP(2, 3, 3, 3)>(x), dn2, y);
            return fma_(dpp_<QP(3, 3, 3, 3)>(x), dn3, y);
        } else if constexpr (R == 8) {
            real y = dpp_<ROW_SHL(1)>(x) * dn1;
            y = fma_(dpp_<ROW_SHL(1)>(y), dn1, y);
            y = fma_(dpp_<ROW_SHL(2)>(y), dn2, y);
            return y + dpp_<ROW_SHL(4), 0x5>(y);
        } else {
            real y = dpp_<ROW_SHL(1)>(x);
            y = y + dpp_<ROW_SHL(1)>(y);
            y = y + dpp_<ROW_SHL(2)>(y);
            y = y + dpp_<ROW_SHL(4)>(y);
            return y + dpp_<ROW_SHL(8)>(y);
        }
    }
};

// ---------------------------------------------------------------------------------------------
// dense 16 x 16 operators for the latency-bound regime (K = 16, one state per lane, float).
// A batch too small to fill the chip (the reference's production shape: 500 particles x <= 5 chunks)
// is bound by the latency of ONE site step of ONE sequence times L.  Runs of homozygous sites are
// the rule (>= 90 % of sites), so the product of 2 (or 4) consecutive hom steps, M_h^2 (M_h^4) with
// M_h = A diag(emis0), is precomputed per sequence as a dense matrix -- 16 registers per lane, lane i
// holding column i (forward) or row i (beta scan) -- and applied in ONE step of 16 multiply-adds on
// lane-broadcast operands (DPP row_newbcast), 6 dependent instructions deep, instead of 2 (4)
// structured steps of ~11 dependent instructions each.  A wave takes the dense step when all four of
// its sequences see only hom sites in the group (wave vote); otherwise the structured step.
// ---------------------------------------------------------------------------------------------
template <typename real, int K, int R>
constexpr bool has_dense() { return K == 16 && R == 16 && sizeof(real) == 4; }
// floats per thread of the *_mr dense kernels' LDS slice holding the missing-run operator (DenseOps::q8), behind the workgroup's
// emission tables: 16 slots, padded to an odd number of 16-byte units (conflict-free ds_read_b128, see Lane::ETAB_STRIDE)
constexpr int DENSE_Q8_STRIDE = 20;

template <int J>
__device__ __forceinline__ float row_share(float x) {  // every lane of a 16-lane row reads lane J of its row
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x150 + J, 0xf, 0xf, true));
}

// y = sum_j x[lane j of this row] * d[j]: four accumulator chains.  One asm block: the compiler
// neither fuses the broadcast into v_fmac_f32 nor knows the DPP read-after-VALU-write hazard inside
// asm, so the block starts with the two wait states that hazard needs (x may be freshly written).
__device__ __forceinline__ float dense16(float x, const float (&d)[16]) {
    float c0, c1, c2, c3;
    asm volatile(
        "s_nop 1\n\t"
        "v_mul_f32_dpp %0, %4, %5 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %4, %6 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %2, %4, %7 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %3, %4, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %4, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %4, %10 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %2, %4, %11 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %3, %4, %12 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %4, %13 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %4, %14 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %2, %4, %15 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %3, %4, %16 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %4, %17 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %4, %18 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %2, %4, %19 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %3, %4, %20 row_newbcast:15 row_mask:0xf bank_mask:0xf"
        : "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(c3)
        : "v"(x), "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]), "v"(d[4]), "v"(d[5]), "v"(d[6]), "v"(d[7]), "v"(d[8]),
          "v"(d[9]), "v"(d[10]), "v"(d[11]), "v"(d[12]), "v"(d[13]), "v"(d[14]), "v"(d[15]));
    return (c0 + c1) + (c2 + c3);
}

// Round 4: runs that END in a het or a missing site are dense steps too.  A diag(emis1) = M_h diag(emis1 / emis0) and
// A = M_h diag(1 / emis0), so "n - 1 hom sites, then a het" is M_h^n followed by ONE multiply by a per-state ratio
// (forward form; the beta scan multiplies first).  With every power M_h^1 .. M_h^8 in registers (128 VGPRs; these
// kernels are compiled for two waves per SIMD = 256) a run of any length up to 8 is one dense step, chosen by a
// wave-uniform switch on the run length: the four sequences of a wave read the SAME observation row by construction
// (see the particle-major mapping in fwd_kernel), so the codes are scalars.  Before, a het cost two structured steps
// (~35 instructions each for a lone wave), a broken hom run and two rescales: the reference's production shape lost
// 26 % (5 % hets) to 55 % (10 %) of its rate against 1 %-het rows (profiles/r03_ab_experiments.txt item 18).
#ifndef PHK_DENSE_RESCALE_SITES
#define PHK_DENSE_RESCALE_SITES 64  // dense steps: rescale once the debt (1 per hom site, 16 per het / missing site of the scalar-code path) reaches this
#endif
// The het / missing ratios emis1 / emis0 and 1 / emis0 exist for the scalar-code path while emis0 is not absurdly small
// (the reference clips emissions at 1e-20 = 2^-66): at 2^-64 the ratio is 2^64, a state with alpha >= 2^-62 of the
// total keeps its het contribution exactly, and a smaller one carries nothing in float32.  (2^-30 in the first
// version sent every wave holding a particle with theta x t_M > 21 to the wave-vote path -- one in a few hundred
// particles of fit()'s initial population at 5 % hets, and the slowest wave sets the kernel's time: +20 %.)
constexpr float RATIO_MIN_EMIS0 = 0x1p-64f;
#ifndef PHK_DENSE_UNI
#define PHK_DENSE_UNI 1  // A/B: 0 = waves with one observation row take the wave-vote path like any other
#endif
#ifndef PHK_FOLD
#define PHK_FOLD 1  // A/B: 0 = no float32 kernel folds its hom emission into the factors
#endif
#ifndef PHK_ASM_RUN
#define PHK_ASM_RUN 0  // 1 (developer builds: make -C phlash_amd/csrc OUT=exp/libphk_asm.so OBJDIR=/tmp/build_asm EXTRA=-DPHK_ASM_RUN=1, run with PHK_LIB): the K = 16, R = 2 float32 sweeps run their
                       // hot blocks through the generated instruction sequence sweep_run_k16r2.inc (scripts/gen_sweep_asm.py) when the
                       // handle asks for it (phk_set_asm_run).  Bit-identical to the C++ body and 1-2 % SLOWER than it
                       // (profiles/r06_ab_experiments.txt item 8), hence not in the shipped library
#endif
#ifndef PHK_FOLD_F64
#define PHK_FOLD_F64 1  // A/B: 0 = the float64 kernels keep their emissions in the table (rounds 1-5); 1: folded like the float32 ones (round 6: cfg2 76 -> 67 ms)
#endif
#ifndef PHK_SWEEP_FOLD
#define PHK_SWEEP_FOLD PHK_FOLD  // A/B: 0 = the sweeps' hot body keeps its per-site emission rows (the model is folded all the same)
#endif
#ifndef PHK_FWD_FOLD
#define PHK_FWD_FOLD PHK_FOLD  // A/B: 0 = the forward kernels with several states per lane have no emission-free block
#endif
#ifndef PHK_UNI_SLOAD
#define PHK_UNI_SLOAD 1  // A/B: 0 = such waves read their observation words by vector loads like the others
#endif
#ifndef PHK_DENSE_UNI_SCAN
#define PHK_DENSE_UNI_SCAN PHK_DENSE_UNI  // ... the beta scan alone
#endif
template <bool ON>
struct DenseOps {};
template <>
struct DenseOps<true> {
    float P[8][16];  // P[n-1] = M_h^n, n = 1..8: slot j of lane i = [j][i] (forward) or [i][j] (beta scan)
    float D16[16];   // M_h^16
    // (M_h diag(1 / emis0))^8 = A^8, eight MISSING sites in a row (an accessibility mask leaves runs of hundreds): this lane's
    // sixteen slots in LDS, read when such a half comes up.  In registers (16 more of 221 / 237) the operator cost the forward
    // kernel of the reference's production shape 0.45 of its 1.93 ms on rows without a single run (r06_ab_experiments.txt item 14).
    const float* q8;
    float rhet, rmis;  // this lane's state: emis1 / emis0 and 1 / emis0
};

// ---------------------------------------------------------------------------------------------
// packed pairs: the element-wise part of every site step runs on 2-vectors so that the f32
// instantiation issues v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 (one issue slot, two states).
// A wave can issue one VALU instruction every 4 cycles; at the <= 2 waves per SIMD this problem
// offers, halving the instruction count is worth more than anything else.
// ---------------------------------------------------------------------------------------------
template <typename real>
using vec2 = real __attribute__((ext_vector_type(2)));

template <typename real>
__device__ __forceinline__ vec2<real> fma2(vec2<real> a, vec2<real> b, vec2<real> c) {
    return __builtin_elementwise_fma(a, b, c);
}
template <typename real>
__device__ __forceinline__ vec2<real> splat(real x) {
    vec2<real> r = {x, x};
    return r;
}

// ---------------------------------------------------------------------------------------------
// per-lane slice of one sequence's parameters + the forward / backward site steps
// ---------------------------------------------------------------------------------------------
template <typename real, int K, int R>
struct Lane : DenseOps<has_dense<real, K, R>()> {
    static constexpr int SPL = K / R;        // states owned by this lane
    static constexpr int NP = (SPL + 1) / 2;  // packed pairs (odd SPL: the last pair's .y is padding)
    static constexpr int EROW = 2 * NP;       // reals per emission row in the LDS table
    // reals per thread in the LDS table: 3 rows, padded so that the per-thread stride in 16-byte
    // units is odd -- then the 16 lanes of a ds_read_b128 group fall on 16 different bank quads
    // (with an even stride, lanes l and l+8 collide: SQ_LDS_BANK_CONFLICT was 6-9 % of wave cycles)
    static constexpr int ETAB_RAW = 3 * EROW;
    static constexpr int ETAB_DW = ETAB_RAW * (int)(sizeof(real) / 4);
    static constexpr int ETAB_STRIDE =
        (ETAB_DW % 4 == 0 && (ETAB_DW / 4) % 2 == 0) ? ETAB_RAW + 4 / (int)(sizeof(real) / 4) : ETAB_RAW;
    static_assert(SPL * R == K, "R must divide K");
    using V = vec2<real>;

    V b[NP], d[NP], u[NP], v[NP];
    const real* etab;  // LDS: this thread's emission rows [3][EROW]: hom, het, missing (= ones)
    Group<real, R> g;

    // Split-halves layout (SPLIT): packed pair h of a lane holds its states h (.x) and h + NP (.y).  The two
    // running sums of the mat-vec then advance through BOTH halves of the lane's states with one
    // packed instruction per pair (NP dependent steps instead of SPL scalar ones); the .y half gets
    // the .x half's total through the same carry add that brings in the other lanes' totals.
    // Used where it pays (measured at K = 16 on the backward kernel: SPL = 8 -6.5 %; SPL = 16 +2 %, the
    // instruction count is the same there; SPL = 4 pushes the straight-line path over its register
    // budget); elsewhere pair h holds the adjacent states 2h, 2h + 1 and the sums run state by state.
    static constexpr bool SPLIT = SPL == 8;
    static constexpr int PH(int i) { return SPLIT ? i % NP : i >> 1; }  // pair holding state i of the lane
    static constexpr int HF(int i) { return SPLIT ? i / NP : i & 1; }   // ... and which half of it
    static constexpr int SLOT(int i) { return 2 * PH(i) + HF(i); }    // position in an emission-table row
    static __device__ __forceinline__ real get(const V (&x)[NP], int i) { return x[PH(i)][HF(i)]; }
    static __device__ __forceinline__ void set(V (&x)[NP], int i, real val) { x[PH(i)][HF(i)] = val; }

    // Checkpoint layout in HBM: [block][K/4 pieces][sequence][4 states].  One store instruction of a wave (one 16-byte
    // piece per lane) then writes 64 consecutive pieces = 1 KB of consecutive addresses whatever the kernel's lanes per
    // sequence, and so does a load of the backward kernel.  (Rounds 1-2 kept [block][sequence][K]: a lane's 64 bytes
    // contiguous, so that every dwordx4 store of a wave touched all 32 lines of a 4 KB region in 16-byte pieces; with
    // more than one forward wave per SIMD -- cfg3, cfg5 -- the forward kernel ran 25 % slower than without its
    // stores, profiles/r03_ab_experiments.txt item 12.)
    // ck_lane: offset of this lane's first state inside a block; state i of the lane sits at ck_lane + ck_elem(i, nseq).
    static __device__ __forceinline__ int64_t ck_lane(int64_t nseq, int64_t seq, int rank) {
        const int k0 = rank * SPL;
        return ((int64_t)(k0 / 4) * nseq + seq) * 4 + (SPL < 4 ? k0 % 4 : 0);
    }
    static __device__ __forceinline__ int64_t ck_elem(int i, int64_t nseq) { return (int64_t)(i / 4) * nseq * 4 + (i % 4); }

    // p: [7,K] rows b,d,u,v,emis0,emis1,pi (gpu.py:189 stacking order); padding lanes hold zeros
    // (ones in the emission rows) so that they stay exactly 0 through every step.
    __device__ __forceinline__ void load(const real* __restrict__ p, int rank, real* etab_thread, V (&pi)[NP]) {
        g.init(rank);
        etab = etab_thread;
        const real* q = p + rank * SPL;
#pragma unroll
        for (int h = 0; h < NP; ++h) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int i = SPLIT ? h + c * NP : 2 * h + c;  // state of the lane in half c of pair h
                const bool ok = i < SPL;
                b[h][c] = ok ? q[0 * K + i] : real(0);
                d[h][c] = ok ? q[1 * K + i] : real(0);
                u[h][c] = ok ? q[2 * K + i] : real(0);
                v[h][c] = ok ? q[3 * K + i] : real(0);
                pi[h][c] = ok ? q[6 * K + i] : real(0);
                etab_thread[0 * EROW + 2 * h + c] = ok ? q[4 * K + i] : real(1);
                etab_thread[1 * EROW + 2 * h + c] = ok ? q[5 * K + i] : real(1);
                etab_thread[2 * EROW + 2 * h + c] = real(1);
            }
        }
    }

    // Forward kernel, float32: alpha' = e .* (d .* alpha + v .* pre + b .* suf), so the hom emission can live inside
    // the factors (b, d, v) <- emis0 .* (b, d, v) and the table rows become what is left to multiply by: 1 for a hom
    // site, emis1 / emis0 for a het, 1 / emis0 for a missing one.  Every code path of the kernel stays what it is (it
    // multiplies by its site's row), and a site known to be hom for the whole wave multiplies by nothing and reads no
    // row at all (fwd_site<false>).  Only where the ratios exist: see RATIO_MIN_EMIS0.
    // (decided per SEQUENCE -- all R lanes of the group agree -- so that what is computed for a sequence does not
    // depend on which other sequences share its wave)
    __device__ __forceinline__ bool emissions_foldable() const {
        bool ok = true;
#pragma unroll
        for (int i = 0; i < EROW; ++i) ok = ok && etab[i] > (real)RATIO_MIN_EMIS0;
        return g.sum(ok ? real(0) : real(1)) == real(0);
    }
    // pf (round 6, KArgs::prefold): this lane's slice of the sequence's PRE-FOLDED block [5, K] -- rows fl(emis0 b),
    // fl(emis0 d), fl(emis0 v), fl(emis1 / emis0), fl(1 / emis0), formed in float64 from the float64 parameters and
    // rounded ONCE (phk_prefold).  Folding here, from the float32-rounded factors, rounds each of b, d, v three times
    // (b, emis0, their product), and the error of a factor is the same at every site: it adds up along the row.
    __device__ __forceinline__ void fold_emissions(const real* pf) {
        real* t = const_cast<real*>(etab);
#pragma unroll
        for (int h = 0; h < NP; ++h) {
            const V e0 = ((const V*)t)[h];
            const V e1 = ((const V*)(t + EROW))[h];
            if (pf == nullptr) {
                b[h] = b[h] * e0;
                d[h] = d[h] * e0;
                v[h] = v[h] * e0;
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int i = SPLIT ? h + c * NP : 2 * h + c;  // state of the lane in half c of pair h
                t[0 * EROW + 2 * h + c] = real(1);
                if (pf != nullptr) {
                    const bool ok = i < SPL;
                    b[h][c] = ok ? pf[0 * K + i] : real(0);
                    d[h][c] = ok ? pf[1 * K + i] : real(0);
                    v[h][c] = ok ? pf[2 * K + i] : real(0);
                    t[1 * EROW + 2 * h + c] = ok ? pf[3 * K + i] : real(1);
                    t[2 * EROW + 2 * h + c] = ok ? pf[4 * K + i] : real(1);
                } else {
                    t[1 * EROW + 2 * h + c] = (real)((double)e1[c] / (double)e0[c]);
                    t[2 * EROW + 2 * h + c] = (real)(1.0 / (double)e0[c]);
                }
            }
        }
    }
    // Every float32 kernel -- forward, beta scan, sweeps, every variant -- runs on the folded model whenever the
    // sequence's ratios exist: the kernels of one evaluation must agree on the factors to the last bit, or the forward
    // kernel's alpha and the sweep's beta belong to two HMMs that differ by one rounding per factor, a difference
    // that is the same at every site and adds up along the row (60,000 sites: sum(alpha .* beta) drifts 6e-4 from 1
    // and takes every gradient row with it).  float64 kernels keep their emissions in the table.
    __device__ __forceinline__ bool try_fold(const real* pf = nullptr) {
        if constexpr ((sizeof(real) == 4 || PHK_FOLD_F64 != 0) && PHK_FOLD != 0) {
            const bool ok = emissions_foldable();
            if (ok) fold_emissions(pf);
            return ok;
        } else {
            return false;
        }
    }

    __device__ __forceinline__ void emis(int code, V (&e)[NP]) const {
        const V* row = (const V*)(etab + code * EROW);
#pragma unroll
        for (int h = 0; h < NP; ++h) e[h] = row[h];
    }

    // In-lane part of an exclusive prefix of w.*x (PREFIX) or exclusive suffix (else) over the lane's
    // states, both halves at once; returns the inclusive totals of the two halves in `tot`.
    template <bool PREFIX, bool WEIGHTED>
    __device__ __forceinline__ void half_scans(const V (&w)[NP], const V (&x)[NP], V (&out)[NP], V& tot) const {
        V t = splat<real>(real(0));
        if constexpr (PREFIX) {
#pragma unroll
            for (int h = 0; h < NP; ++h) {
                out[h] = t;
                t = WEIGHTED ? fma2<real>(w[h], x[h], t) : t + x[h];
            }
        } else {
#pragma unroll
            for (int h = NP - 1; h >= 0; --h) {
                out[h] = t;
                t = WEIGHTED ? fma2<real>(w[h], x[h], t) : t + x[h];
            }
        }
        tot = t;
    }
    // carry added to every pair: what precedes (PREFIX) / follows the lane's states in other lanes,
    // plus, for the half that needs it, the total of the lane's other half
    template <bool PREFIX>
    __device__ __forceinline__ V carry(const V& tot) const {
        const real lane_total = SPL > 1 ? tot[0] + tot[1] : tot[0];
        real c = real(0);
        if constexpr (R > 1) c = PREFIX ? g.excl_prefix(lane_total) : g.excl_suffix(lane_total);
        V r;
        if constexpr (PREFIX) {
            r[0] = c;
            r[1] = SPL > 1 ? c + tot[0] : c;  // the .y states also follow every .x state
        } else {
            r[0] = SPL > 1 ? c + tot[1] : c;  // the .x states are also followed by every .y state
            r[1] = c;
        }
        return r;
    }
    // adjacent-pairs layout: exclusive prefix of wp.*xp into `pre` and exclusive suffix of xs (SUFW:
    // of v.*xs) into `suf`, state by state inside the lane, then the other lanes' totals
    __device__ __forceinline__ void serial_scans(const V (&wp)[NP], const V (&xp)[NP], V (&pre)[NP], const V (&xs)[NP],
                                                 V (&suf)[NP], const bool SUFW) const {
        real tp = real(0);
#pragma unroll
        for (int i = 0; i < 2 * NP; ++i) {
            pre[i >> 1][i & 1] = tp;
            if (i < SPL) tp = fma_(wp[i >> 1][i & 1], xp[i >> 1][i & 1], tp);
        }
        real ts = real(0);
#pragma unroll
        for (int i = 2 * NP - 1; i >= 0; --i) {
            suf[i >> 1][i & 1] = ts;
            if (i < SPL) ts = SUFW ? fma_(v[i >> 1][i & 1], xs[i >> 1][i & 1], ts) : ts + xs[i >> 1][i & 1];
        }
        if constexpr (R > 1) {
            const V cp = splat<real>(g.excl_prefix(tp));
            const V cs = splat<real>(g.excl_suffix(ts));
#pragma unroll
            for (int h = 0; h < NP; ++h) {
                pre[h] = pre[h] + cp;
                suf[h] = suf[h] + cs;
            }
        }
    }
    // exclusive prefix of u.*x and exclusive suffix of x over the K states of the sequence
    __device__ __forceinline__ void scans(const V (&x)[NP], V (&pre_ux)[NP], V (&suf_x)[NP]) const {
        if constexpr (!SPLIT) {
            serial_scans(u, x, pre_ux, x, suf_x, false);
            return;
        }
        V tu, ta;
        half_scans<true, true>(u, x, pre_ux, tu);
        half_scans<false, false>(u, x, suf_x, ta);
        if constexpr (R > 1 || SPL > 1) {
            const V cu = carry<true>(tu);
            const V ca = carry<false>(ta);
#pragma unroll
            for (int h = 0; h < NP; ++h) {
                pre_ux[h] = pre_ux[h] + cu;
                suf_x[h] = suf_x[h] + ca;
            }
        }
    }
    // exclusive suffix of v.*w (complete) and exclusive prefix of b.*w (the adjoint mat-vec).  The
    // prefix is only ever added to d.*w, so its in-lane part and its carry `cb` are handed back
    // separately: fma(d, w, pbw) + cb costs one instruction per pair less than (pbw + cb) + d*w.
    __device__ __forceinline__ void scans_adj(const V (&w)[NP], V (&svw)[NP], V (&pbw)[NP], V& cb) const {
        if constexpr (!SPLIT) {
            real tv = real(0);
#pragma unroll
            for (int i = 2 * NP - 1; i >= 0; --i) {
                svw[i >> 1][i & 1] = tv;
                if (i < SPL) tv = fma_(v[i >> 1][i & 1], w[i >> 1][i & 1], tv);
            }
            real tb = real(0);
#pragma unroll
            for (int i = 0; i < 2 * NP; ++i) {
                pbw[i >> 1][i & 1] = tb;
                if (i < SPL) tb = fma_(b[i >> 1][i & 1], w[i >> 1][i & 1], tb);
            }
            cb = splat<real>(real(0));
            if constexpr (R > 1) {
                const V cv = splat<real>(g.excl_suffix(tv));
                cb = splat<real>(g.excl_prefix(tb));
#pragma unroll
                for (int h = 0; h < NP; ++h) svw[h] = svw[h] + cv;
            }
        } else {
            V tv, tb;
            half_scans<false, true>(v, w, svw, tv);
            half_scans<true, true>(b, w, pbw, tb);
            const V cv = carry<false>(tv);
            cb = carry<true>(tb);
#pragma unroll
            for (int h = 0; h < NP; ++h) svw[h] = svw[h] + cv;
        }
    }
    // exclusive suffix of v.*w alone (complete, other lanes' totals included): the one scan of the adjoint side a
    // gradient row needs (gu += a .* suf(v.*w)); pre(b.*w) only feeds the beta recursion
    __device__ __forceinline__ void suffix_vw(const V (&w)[NP], V (&svw)[NP]) const {
        if constexpr (!SPLIT) {
            real tv = real(0);
#pragma unroll
            for (int i = 2 * NP - 1; i >= 0; --i) {
                svw[i >> 1][i & 1] = tv;
                if (i < SPL) tv = fma_(v[i >> 1][i & 1], w[i >> 1][i & 1], tv);
            }
            if constexpr (R > 1) {
                const V cv = splat<real>(g.excl_suffix(tv));
#pragma unroll
                for (int h = 0; h < NP; ++h) svw[h] = svw[h] + cv;
            }
        } else {
            V tv;
            half_scans<false, true>(v, w, svw, tv);
            const V cv = carry<false>(tv);
#pragma unroll
            for (int h = 0; h < NP; ++h) svw[h] = svw[h] + cv;
        }
    }
    // beta_prev = d.*w + pre(b.*w) + u.*suf(v.*w)
    __device__ __forceinline__ V beta_prev(int h, const V (&w)[NP], const V (&svw)[NP], const V (&pbw)[NP], const V& cb) const {
        V nb = fma2<real>(d[h], w[h], pbw[h]);
        if constexpr (R > 1 || SPLIT) nb = nb + cb;
        return fma2<real>(u[h], svw[h], nb);
    }

    // sum over the K states of the sequence (all lanes of the group get the same bits)
    __device__ __forceinline__ real total(const V (&x)[NP]) const {
        // One state per lane: the pair's second half is padding.  It stays exactly 0 in the forward recursion, but
        // the adjoint recursion (beta_prev) leaves pre(b.*w) + carry in it -- recomputed by every structured step and
        // read by none, so harmless there; the dense steps do not touch it, and a stale value from the last
        // structured step would take over the normaliser once the real state has decayed (found in round 4, when
        // het sites became dense steps: the scan stopped rescaling and beta underflowed).
        if constexpr (SPL == 1) return g.sum(x[0][0]);
        V acc = x[0];
#pragma unroll
        for (int h = 1; h < NP; ++h) acc = acc + x[h];
        return g.sum(acc[0] + acc[1]);
    }

    // The dense hom operators of this lane's sequence, from the table dense_ops_kernel built for the launch (see there):
    // slot j of lane i = (M_h^n)[j][i] in the forward (row-vector) form, (M_h^n)[i][j] in the beta-scan (column-vector)
    // form; the table holds lane i's 16 slots of a power contiguously in either form.  Rounds 1-3 built the powers in
    // the prologue of every wave (float64 products over DPP broadcasts: ~5,000 instructions, 1.4 KB of scratch per
    // lane); with all eight powers that prologue outgrew the 2,048-site problem the tuner times these kernels on.
    template <bool NEED16>
    __device__ __forceinline__ void load_dense(const float* __restrict__ ops, int rank, const bool folded, float* q8_lds) {
        if constexpr (has_dense<real, K, R>()) {
            const float4* src = (const float4*)(ops + rank * 16);
#pragma unroll
            for (int n = 0; n < 8; ++n) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v4 = src[n * 64 + q];
                    this->P[n][4 * q + 0] = v4.x;
                    this->P[n][4 * q + 1] = v4.y;
                    this->P[n][4 * q + 2] = v4.z;
                    this->P[n][4 * q + 3] = v4.w;
                }
            }
            if constexpr (NEED16) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v4 = src[8 * 64 + q];
                    this->D16[4 * q + 0] = v4.x;
                    this->D16[4 * q + 1] = v4.y;
                    this->D16[4 * q + 2] = v4.z;
                    this->D16[4 * q + 3] = v4.w;
                }
            }
            if (q8_lds != nullptr) {  // (the *_mr kernels only: see half_step)
#pragma unroll
                for (int q = 0; q < 4; ++q) ((float4*)q8_lds)[q] = src[9 * 64 + q];
            }
            this->q8 = q8_lds;
            // (a folded lane's table already holds the ratios: rows 1 and 2)
            const double e0 = (double)etab[0], e1 = (double)etab[EROW];
            this->rhet = folded ? etab[EROW] : (float)(e1 / e0);
            this->rmis = folded ? etab[2 * EROW] : (float)(1.0 / e0);
            // (pinned: left alone the compiler sinks the divisions into the site loop to save two registers)
            asm volatile("" : "+v"(this->rhet), "+v"(this->rmis));
        }
    }
    // x <- x M_h^n (forward form) / M_h^n x (beta-scan form) for a wave-uniform n in [1, NMAX], NMAX = 8 or 16: one
    // dense step for n <= 8 and n = 16, two for 9..15
    template <int NMAX>
    __device__ __forceinline__ float hom_power(float x, int n) const {
        if constexpr (has_dense<real, K, R>()) {
            if constexpr (NMAX > 8) {
                if (n == 16) return dense16(x, this->D16);
                if (n > 8) {
                    x = dense16(x, this->P[7]);
                    n -= 8;
                }
            }
            if constexpr (NMAX <= 4) {
                if (n == 4) return dense16(x, this->P[3]);
                if (n & 2) return (n & 1) ? dense16(x, this->P[2]) : dense16(x, this->P[1]);
                return dense16(x, this->P[0]);
            }
            if (n == 8) return dense16(x, this->P[7]);
            if (n & 4) {
                if (n & 2) return (n & 1) ? dense16(x, this->P[6]) : dense16(x, this->P[5]);
                return (n & 1) ? dense16(x, this->P[4]) : dense16(x, this->P[3]);
            }
            if (n & 2) return (n & 1) ? dense16(x, this->P[2]) : dense16(x, this->P[1]);
            return dense16(x, this->P[0]);
        } else {
            return x;
        }
    }

    // ---- eight sites whose codes are a wave-uniform 16-bit word h (site i = bits [2i, 2i+2)) ----------------------
    // Measured on the MI355X (scripts/microbench/latency.hip, one wave alone on its SIMD): a VALU instruction issues
    // every ~6.5 cycles whether or not it depends on the one before, a dense step is 94 cycles, a rescale 72, and a
    // TAKEN branch ~40 -- six instructions' worth.  So the shapes that matter get straight-line code behind ONE
    // dispatch: all hom (one dense step), and exactly one het / missing site at position p (two dense steps and a
    // multiply, 8-way switch on p).  Two or more such sites: the same by quarters of four sites, and run by run (a
    // dispatch per run) only inside a quarter that holds two or more.
    // Forward form (FWD): sites 0 .. 7 in order, x <- x M_h^(p+1) .* ratio, then M_h^(7-p).
    // Beta-scan form: sites 7 .. 0, x <- M_h^(p+1) (ratio .* (M_h^(7-p) x)).
    // Returns the rescale debt of the half (1 per hom site, 16 per het / missing site); `resc(x)` is called after a run
    // of the generic loop whenever the running debt (debt0 + so far) reaches the threshold: it rescales x, books the
    // exponent and returns the debt it leaves (0).
    template <bool FWD, bool MR, typename Resc>
    __device__ __forceinline__ int half_step(float& x, const uint32_t h, const int debt0, Resc&& resc) const {
        if constexpr (has_dense<real, K, R>()) {
            if (__builtin_expect(h == 0u, 1)) {
                x = dense16(x, this->P[7]);
                return debt0 + 8;
            }
            const uint32_t m = (h | (h >> 1)) & 0x5555u;  // bit 2i set: site i is not hom
            if (__builtin_expect((m & (m - 1u)) == 0u, 1)) {
                const int p = __builtin_ctz(m) >> 1;
                const float r = ((h >> (2 * p)) & 3u) == 1u ? this->rhet : this->rmis;
                // `a` sites before the ratio, `b` after it (forward: a = p + 1, b = 7 - p; scan: a = 7 - p, b = p + 1,
                // and the ratio goes BEFORE the site's own step)
                switch (FWD ? p : 7 - p) {
#define PHK_ONE(q)                                                                            \
    case q:                                                                                   \
        if (FWD) {                                                                            \
            x = dense16(x, this->P[q]) * r;                                                   \
            if (q < 7) x = dense16(x, this->P[q < 7 ? 6 - q : 0]);                            \
        } else {                                                                              \
            if (q > 0) x = dense16(x, this->P[q > 0 ? q - 1 : 0]);                            \
            x = dense16(x * r, this->P[7 - q]);                                               \
        }                                                                                     \
        break;
                    PHK_ONE(0) PHK_ONE(1) PHK_ONE(2) PHK_ONE(3) PHK_ONE(4) PHK_ONE(5) PHK_ONE(6)
                    default:
                    PHK_ONE(7)
#undef PHK_ONE
                }
                return debt0 + 8 + 16;
            }
            // eight missing sites: one step by A^8 = (M_h diag(1 / emis0))^8 (either form: the ratio sits between the hom
            // steps).  Masked stretches of a genome are runs of hundreds of missing windows; site by site they cost one
            // dense step EACH, eight to sixteen times a hom site (500 x 5 x 100,000 at 7 % hets with a quarter of every
            // row masked: forward phase 7.2 instead of 3.2 ms).  Only in the kernels
            // launched for rows that hold such runs (MR: fwd_kernel_mr / bscan_kernel_mr).  A stochastic matrix takes no mass away: debt as for hom sites.
            if (MR && h == 0xAAAAu) {
                float Q[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v4 = ((const float4*)this->q8)[q];
                    Q[4 * q + 0] = v4.x;
                    Q[4 * q + 1] = v4.y;
                    Q[4 * q + 2] = v4.z;
                    Q[4 * q + 3] = v4.w;
                }
                x = dense16(x, Q);
                return debt0 + 8;
            }
            // two or more het / missing sites: by quarters of four sites -- all hom: M_h^4; one such site: 4-way
            // dispatch; more: run by run (a dispatch per run)
            int debt = debt0;
#pragma nounroll
            for (int qi = 0; qi < 2; ++qi) {
                const uint32_t hq = (h >> (8 * (FWD ? qi : 1 - qi))) & 0xffu;
                if (hq == 0u) {
                    x = dense16(x, this->P[3]);
                    debt += 4;
                    continue;
                }
                const uint32_t mq = (hq | (hq >> 1)) & 0x55u;
                if ((mq & (mq - 1u)) == 0u) {
                    const int p = __builtin_ctz(mq) >> 1;
                    const float r = ((hq >> (2 * p)) & 3u) == 1u ? this->rhet : this->rmis;
                    switch (FWD ? p : 3 - p) {
#define PHK_ONE4(q)                                                                           \
    case q:                                                                                   \
        if (FWD) {                                                                            \
            x = dense16(x, this->P[q]) * r;                                                   \
            if (q < 3) x = dense16(x, this->P[q < 3 ? 2 - q : 0]);                            \
        } else {                                                                              \
            if (q > 0) x = dense16(x, this->P[q > 0 ? q - 1 : 0]);                            \
            x = dense16(x * r, this->P[3 - q]);                                               \
        }                                                                                     \
        break;
                        PHK_ONE4(0) PHK_ONE4(1) PHK_ONE4(2)
                        default:
                        PHK_ONE4(3)
#undef PHK_ONE4
                    }
                    debt += 4 + 16;
                } else if (FWD) {
                    uint32_t rem = hq;
                    int left = 4;
#pragma nounroll
                    do {
                        const int tz = rem != 0u ? (__builtin_ctz(rem) >> 1) : 32;
                        const bool stop = tz < left;  // the run ends in a het / missing site
                        const int run = stop ? tz + 1 : left;
                        x = hom_power<4>(x, run);
                        if (stop) {
                            x *= ((rem >> (2 * tz)) & 3u) == 1u ? this->rhet : this->rmis;
                            debt += 16;
                        }
                        debt += run;
                        rem >>= 2 * run;
                        left -= run;
                        if (debt >= PHK_DENSE_RESCALE_SITES) debt = resc(x);
                    } while (left > 0);
                } else {
                    int left = 4;  // sites [0, left) of the quarter are still to do; the next one is left - 1
#pragma nounroll
                    do {
                        const uint32_t top = (hq >> (2 * (left - 1))) & 3u;
                        if (top != 0u) {
                            x *= top == 1u ? this->rhet : this->rmis;
                            debt += 16;
                        }
                        const uint32_t below = hq & ((1u << (2 * (left - 1))) - 1u);
                        const int s = below != 0u ? (31 - __builtin_clz(below)) >> 1 : -1;  // next het / missing site to the left
                        const int run = left - 1 - s;
                        x = hom_power<4>(x, run);
                        debt += run;
                        left = s + 1;
                        if (debt >= PHK_DENSE_RESCALE_SITES) debt = resc(x);
                    } while (left > 0);
                }
                if (debt >= PHK_DENSE_RESCALE_SITES) debt = resc(x);
            }
            return debt;
        } else {
            return debt0;
        }
    }

    // power-of-two rescale of a one-state-per-lane vector held in a scalar register per lane; returns the exponent removed
    __device__ __forceinline__ int rescale1(float& x) const {
        const float c = g.sum(x);
        const int ex = frexp_exp_(c);
        x *= ldexp_(1.0f, -ex);
        return ex;
    }
    // power-of-two rescale of a state vector on its own (the SCALE part of fwd_site / bt_site)
    __device__ __forceinline__ int rescale(V (&x)[NP]) const {
        const real c = total(x);
        const int ex = frexp_exp_(c);
        const V s2 = splat<real>(ldexp_(real(1), -ex));
#pragma unroll
        for (int h = 0; h < NP; ++h) x[h] = x[h] * s2;
        return ex;
    }

    // One forward site (hmm.py:74-79): a <- (a A) .* e_code, then, if SCALE, a *= 2^-ex with ex the
    // exponent of the sum (so that sum(a) lands in [0.5,1)); returns ex (0 if !SCALE) and the scale.
    // code: 0 hom, 1 het, 2 missing (emission 1; hmm.py:70-71)
    // (EMIS = false: a hom site of a lane whose hom emission is folded into b, d, v -- fold_emissions -- multiplies
    // by nothing)
    template <bool EMIS = true>
    __device__ __forceinline__ int fwd_site(V (&a)[NP], const V (&e)[NP], real& scale, const bool SCALE) const {
        V pre[NP], suf[NP];
        scans(a, pre, suf);
#pragma unroll
        for (int h = 0; h < NP; ++h) {
            V t = d[h] * a[h];
            t = fma2<real>(v[h], pre[h], t);
            t = fma2<real>(b[h], suf[h], t);
            a[h] = EMIS ? t * e[h] : t;
        }
        if (SCALE) {
            const real c = total(a);
            const int ex = frexp_exp_(c);
            scale = ldexp_(real(1), -ex);  // exact power of two: the products below do not round
            const V s2 = splat<real>(scale);
#pragma unroll
            for (int h = 0; h < NP; ++h) a[h] = a[h] * s2;
            return ex;
        } else {
            scale = real(1);
            return 0;
        }
    }

    // One backward site.  In: ap = alpha before the site, aq = alpha after it, beta = d ll/d aq,
    // s = the scale applied at the site (SCALE) .  Out: beta = d ll / d ap; accumulated:
    //   gb += w.*suf(ap)   gd += w.*ap   gu += ap.*suf(v.*w)   gv += w.*pre(u.*ap)
    //   g0/g1 += aq.*beta  (divided by emis0/emis1 at the end)         with w = e.*beta*s
    __device__ __forceinline__ void bwd_site(const V (&ap)[NP], const V (&aq)[NP], V (&beta)[NP], const V (&e)[NP],
                                             int code, real s, const bool SCALE, V (&gb)[NP], V (&gd)[NP], V (&gu)[NP],
                                             V (&gv)[NP], V (&g0)[NP], V (&g1)[NP], const int g0code = 0) const {
        const V f1 = splat<real>(code == 1 ? real(1) : real(0));
        const V f0 = splat<real>(code == g0code ? real(1) : real(0));  // (folded form: g0 collects the mass at MISSING sites)
        V pre[NP], suf[NP], w[NP];
        scans(ap, pre, suf);
#pragma unroll
        for (int h = 0; h < NP; ++h) {
            const V m = aq[h] * beta[h];
            g1[h] = fma2<real>(f1, m, g1[h]);
            g0[h] = fma2<real>(f0, m, g0[h]);
            V t = beta[h] * e[h];
            if (SCALE) t = t * splat<real>(s);
            w[h] = t;
        }
        // suffix of v.*w and prefix of b.*w
        V svw[NP], pbw[NP], cb;
        scans_adj(w, svw, pbw, cb);
#pragma unroll
        for (int h = 0; h < NP; ++h) {
            gb[h] = fma2<real>(w[h], suf[h], gb[h]);
            gd[h] = fma2<real>(w[h], ap[h], gd[h]);
            gv[h] = fma2<real>(w[h], pre[h], gv[h]);
            gu[h] = fma2<real>(ap[h], svw[h], gu[h]);
            beta[h] = beta_prev(h, w, svw, pbw, cb);
        }
    }

    // One step of the unnormalised backward recursion b*_{t-1} = A (e_t .* b*_t), then (SCALE) a
    // power-of-two rescale of its own; returns the exponent removed.  Used by the beta scan only.
    __device__ __forceinline__ int bt_site(V (&beta)[NP], const V (&e)[NP], const bool SCALE) const {
        V w[NP], svw[NP], pbw[NP];
#pragma unroll
        for (int h = 0; h < NP; ++h) w[h] = beta[h] * e[h];
        V cb;
        scans_adj(w, svw, pbw, cb);
#pragma unroll
        for (int h = 0; h < NP; ++h) beta[h] = beta_prev(h, w, svw, pbw, cb);
        if (SCALE) {
            const real c = total(beta);
            const int ex = frexp_exp_(c);
            const V s2 = splat<real>(ldexp_(real(1), -ex));
#pragma unroll
            for (int h = 0; h < NP; ++h) beta[h] = beta[h] * s2;
            return ex;
        }
        return 0;
    }
};

// 2-bit observation codes: 16 sites per dword; site t of a row -> bits [2*(t%16), +2) of word t/16

struct SeqAux {      // written by the forward kernel, read by the backward kernel
    double inv_end;  // 1 / sum(alpha) after the last site (scaled state)
    int32_t e_end;   // exponent total E after the last site: true alpha_L = alpha * 2^E
    int32_t eb_min;  // smallest exponent total of any checkpoint block of the sequence (<= 0).  REQUIRED of every producer
                     // of checkpoints: bwd_kernel decides from it whether the wave may take its unscaled hot body
    int32_t folded;  // written by the segment sweep: 0 = its partial sums are d ll / d(parameter) as they stand; 1 / 2 = they
                     // are those of the folded model (see bwd_kernel; 1: hom mass = remainder, 2: booked directly) and
                     // grad_unfold_kernel converts them after grad_finalize_kernel
    int32_t pad_;
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
    void* ckpt;              // [nblk][K/4 pieces][B*S][4 states] real: Lane::ck_lane / ck_elem  (null: forward only)
    SeqAux* aux;             // [B*S]
    void* grad;              // [B*S, 7, K] real
    double* gacc;            // [B*S, 6, K] f64 partial sums (f32 kernels), zeroed before launch
    int grad_dlog;           // 1: return theta * d ll/d theta (what the reference kernel returns)
    // scale bookkeeping written by the forward kernel (lets the backward kernel use another variant)
    int16_t* eblk;           // [nblk, B*S] exponent total of every block
    int32_t* eseg;           // [nseg+1, B*S] exponent total before the first site of every segment
    // segmented (small-batch) backward: independent units of seg_blocks blocks per sequence
    int seg_blocks;          // blocks per segment (0: one serial sweep per sequence)
    const void* bseg;        // [nseg+1, B*S, K] real: scaled b*_t = A(e.b*) recursion at segment starts
    const int32_t* fseg;     // [nseg+1, B*S] its exponents
    double* bpi;             // [B*S, K] d ll / d pi (segmented mode; the serial kernel writes grad itself)
    int* risk;               // set to 1 if a rescale ever found the mass below 2^RISK_EXP (see below)
    // Sub-range [seq_begin, seq_end) of the sequences (chunk-major storage index s * B + b: see SeqMap) this launch covers;
    // seq_end == 0 means all.  The hybrid plan sweeps one range serially and the other by segments;
    // every per-sequence array keeps its full B*S layout.
    int64_t seq_begin, seq_end;
    int64_t N;               // rows of the observation matrix (chunk indices outside [0, N) raise FLAG_BAD_INDEX)
    // segment sweep: unit y >= 1 of sequence seq writes its partial sums ONCE to
    // part[((y - 1) * (seq_hi - seq_begin) + (seq - seq_begin)) * 6 * K ...] (real); unit 0 adds into
    // gacc (it alone touches the sequence's row there); grad_finalize_kernel adds them up in unit order.
    void* part;
    // dense hom-run operators of the one-state-per-lane kernels (K = 16, float32; dense_ops_kernel): per parameter block
    // [10 operators M_h^1..8, M_h^16, (M_h diag(1 / emis0))^8][lane 16][slot 16] floats, forward form and beta-scan form; null where those kernels are not used
    const float* ops_f;
    const float* ops_b;
    // pre-folded factors of every parameter block (float32 kernels; phk_prefold / phk_loglik_prefolded): [B, S|1, 5, K]
    // with the element strides below, or null (the kernels then fold the float32 rows themselves: Lane::fold_emissions)
    const float* prefold;
    int64_t pfstride_b, pfstride_s;
    // Iteration budget of every loop whose trip count derives from these arguments (round 6): an upper bound, computed on the
    // host from the row length, on the loop iterations one wave of each kernel can legitimately make.  A kernel that exhausts it
    // raises FLAG_OVERRUN, records where, and returns -- so that no inconsistency of the arguments (or of this file) can make a
    // wave spin until the watchdog takes the GPU away.  [0] forward kernel (outer iterations: one per 64-site piece or per
    // block of a ragged piece), [1] serial sweep (blocks), [2] beta scan (pieces / words), [3] one unit of the segment sweep (blocks).
    int32_t loop_budget[4];
    int32_t scan_prio;  // s_setprio of the beta scan's waves (0..3)
    int32_t mask_runs;  // 1: the rows hold runs of missing sites; the one-state-per-lane kernels are launched in their *_mr form
    int32_t asm_run;  // 1: the K = 16, R = 2 float32 sweeps run their hot blocks through the hand-written sequence (0: the C++ body; tests)
};

// this sequence's pre-folded block (lane slice added by the caller), or null
template <typename real>
__device__ __forceinline__ const real* prefold_block(const KArgs& A, int64_t bb, int64_t ss) {
    if constexpr (sizeof(real) == 4) return A.prefold != nullptr ? (const real*)A.prefold + bb * A.pfstride_b + ss * A.pfstride_s : nullptr;
    else return nullptr;
}
constexpr int DENSE_NPOW = 10;  // M_h^1 .. M_h^8, M_h^16, (M_h diag(1 / emis0))^8
constexpr int DENSE_OPS_FLOATS = DENSE_NPOW * 256;  // per parameter block and form

// bits of the sticky device flag word (KArgs::risk)
constexpr int FLAG_UNDERFLOW = 1;  // a rescale found the mass below 2^RISK_EXP (see below)
constexpr int FLAG_BAD_INDEX = 2;  // a chunk index outside [0, N): the row was clamped to 0, the result is garbage
constexpr int FLAG_OVERRUN = 4;    // a block / piece loop ran out of the iteration budget the host gave it (KArgs::loop_budget): the kernel
                                   // returned early, the call's results are garbage; risk[1..3] name the kernel, sequence and block

// first reporter records (kernel id, sequence, block or word) behind the flag word
__device__ __forceinline__ void report_overrun(const KArgs& A, int kernel_id, int64_t seq, int where) {
    if (A.risk == nullptr) return;
    if ((atomicOr(A.risk, FLAG_OVERRUN) & FLAG_OVERRUN) == 0) {
        A.risk[1] = kernel_id;
        A.risk[2] = (int)seq;
        A.risk[3] = where;
    }
}

// row of the observation matrix for chunk ss, range-checked (gpu.py:197-199 asserts this on the host;
// here the indices live on the device, so the check does too)
__device__ __forceinline__ int64_t checked_row(const KArgs& A, int64_t ss) {
    int64_t row = A.inds[ss];
    if (row < 0 || row >= A.N) {
        if (A.risk != nullptr) atomicOr(A.risk, FLAG_BAD_INDEX);
        row = 0;
    }
    return row;
}

// With rescaling only every NRM-th site the unscaled mass must survive NRM sites.  A rescale that
// finds the total below 2^RISK_EXP means the parameters are extreme enough (emissions near the
// reference's 1e-20 clip on a run of such sites) that float32 could have lost states or underflowed in
// between; the forward kernel then raises a flag and the host re-evaluates with per-site rescaling
// (NRM = 1, the reference's schedule), which is always safe.
constexpr int RISK_EXP_F32 = -64;
constexpr int RISK_EXP_F64 = -600;
// The dense steps of the one-state-per-lane kernels defer their rescale over up to 64 hom sites (or 4 het / missing
// sites): the exponent such a rescale removes is the decay of up to 80 sites, not of 4, and a harmless 1 bit per site
// would trip the threshold above -- flipping the kernel object to per-site rescaling for good.  Those rescales are
// held against this threshold instead: the total still sits 2^30 above the smallest normal float, and a state that far
// below the total carries nothing.
constexpr int RISK_EXP_DEFERRED_F32 = -96;
// The backward kernel runs a whole checkpoint block unscaled in its hot body (PHK_SWEEP_V2: both passes of a block
// start from the checkpoint and only the block's exponent TOTAL is applied, to beta, at the block's edge).  A block out
// of which the forward kernel took more than this many binary orders takes the general body instead (it rescales as
// it goes): decided per block from the recorded exponent, no flag, no fallback of the whole kernel object.
constexpr int HOT_BLOCK_MIN_EXP_F32 = -64;
constexpr int HOT_BLOCK_MIN_EXP_F64 = -600;

constexpr double LN2 = 0.693147180559945309417232121458;

// SCALE arguments of the site steps are compile-time constants after unrolling (the site loops
// below are fully unrolled and the steps force-inlined), so the untaken side folds away.
// Rescaling schedule: site t (0-based) is followed by a rescale iff t % NRM == NRM - 1.  T is a
// multiple of NRM and blocks start at multiples of T, so the schedule is a function of the index
// inside the block.  NRM = 1 is the reference's "normalise every site" (hmm.py:77-79).
template <int NRM>
__device__ __forceinline__ constexpr bool rescale_after(int i) { return (i % NRM) == NRM - 1; }

// The backward kernel has a straight-line path for full blocks (no per-site branches, next
// emission row prefetched).  It needs a few more live registers than the guarded loop; where the
// T alpha vectors already fill the 256-VGPR budget of 2 waves/SIMD (f32: more than 4 states per
// lane) it spills and runs 4x slower (measured, R=2: 153 ms vs 40 ms), so it is compiled only
// where it fits (measured, R=4: 47 ms vs 56 ms).
template <typename real, int K, int R, int T>
constexpr bool bwd_straight_line() { return T * (K / R) * (int)sizeof(real) <= 128; }

// ---------------------------------------------------------------------------------------------
// kernel 1: forward pass.  ll per sequence; optionally alpha checkpoints every T sites.
// LDS: the per-thread emission table only.
// ---------------------------------------------------------------------------------------------
// Waves per SIMD the latency-bound (one state per lane, dense hom-run operators) kernels are compiled for.
// Left to itself the compiler gives them all 512 registers (the float64 construction of the dense
// operators in the prologue has long live ranges) -- one wave per SIMD, so that the forward kernel and
// the beta scan of the segmented plan, 625 waves each at the reference's production shape, could not
// share the 1,024 SIMDs and ran one after the other.  Their loops need fewer than 100 registers.
// Measured at 500 x 5 x 100,000 (interleaved A/B, round 2): 1 wave per SIMD 11.8-12.4
// ms per step, 2 waves 9.3, 4 waves 9.3-9.4; one 100,000-site sequence: 5.56 / 5.33 / 5.60 ms.
#ifndef PHK_DENSE_WAVES
#define PHK_DENSE_WAVES 2
#endif
#ifndef PHK_DENSE8
#define PHK_DENSE8 1  // A/B: 0 = no M_h^8 step (groups of four sites only)
#endif
#ifndef PHK_DENSE16
#define PHK_DENSE16 1  // A/B: 0 = no M_h^16 step
#endif
#ifndef PHK_FWD_LEAN
#define PHK_FWD_LEAN 1  // A/B: 0 = lean piece loops in the one-state-per-lane kernels only (1: in every forward kernel and beta scan)
#endif
#ifndef PHK_DENSE_LEAN
#define PHK_DENSE_LEAN 1  // A/B: 0 = no lean piece loops in the one-state-per-lane kernels
#endif
template <typename real, int K, int R>
constexpr int scan_waves_per_simd() { return has_dense<real, K, R>() ? PHK_DENSE_WAVES : 1; }

// Which sequence a lane group of the forward kernel / beta scan works on.  Sequences are independent and everything
// stored is indexed by seq = b * S + s, so the assignment is free.  In the one-state-per-lane layout a range of whole
// particles is taken chunk-major, every chunk's particles padded to a multiple of FOUR groups: the four sequences of a
// wave then read the same observation row (scalar codes: Lane::half_step; wave votes that succeed as often as one
// sequence alone would), whatever the particle count (round 3 mapped without the padding: with 250, 125 or 63
// particles -- a rank's share in particle mode -- the waves straddling two chunks took the slow path and set the
// kernel's time).  Padding groups repeat the chunk's last particle: same row, same parameters, hence the same bits
// to the same addresses in the store-by-every-lane loops.
// Observation pieces of a wave-uniform row through the scalar data path: a pointer into the constant address space
// whose value sits in SGPRs (readfirstlane), so that the compiler emits s_load_dwordx4 (lgkmcnt) instead of a vector
// load (vmcnt, shared with the stores).  The packed rows are written by pack_kernel before any of these kernels starts.
typedef uint32_t PieceWords __attribute__((ext_vector_type(4)));  // (a builtin vector: copyable out of another address space)
typedef const PieceWords __attribute__((address_space(4)))* ScalarPieces;
__device__ __forceinline__ ScalarPieces scalar_pieces(const uint4* p) {
    const uint64_t a = (uint64_t)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
    return (ScalarPieces)(((uint64_t)hi << 32) | lo);
}

// Sequence order.  Every per-sequence array the kernels keep for themselves (checkpoints, block and segment
// exponents, segment seeds, partial sums, aux) is indexed CHUNK-MAJOR: seq = s * B + b, and launches cover ranges
// [seq_begin, seq_end) of that order.  Adjacent lanes then hold adjacent particles of ONE chunk: a wave reads one
// observation row (two where it crosses a chunk boundary), which is what lets the forward kernel take its codes as
// scalars (uni2 below), and the checkpoint pieces of a wave stay neighbours in memory.  Only what the caller reads --
// ll [B, S], grad [B, S, 7, K] -- is particle-major: oseq = b * S + s.
__device__ __forceinline__ ScalarPieces scalar_pieces_of_lane(const uint4* p, const int lane_id) {
    const uint64_t a = (uint64_t)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)a, lane_id);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(a >> 32), lane_id);
    return (ScalarPieces)(((uint64_t)hi << 32) | lo);
}

// The checkpoint stream is written once and read once 10-20 ms later, 24 GB at cfg2: non-temporal stores (the `nt` bit: no
// allocation in the caches on the way out).  What the stores cost the forward kernel at all: 1.1 of its 9.0 ms at cfg2 (a
// timing-only build without them, profiles/r05_ab_experiments.txt item 16) -- back-pressure of a 2.7 TB/s write stream, not
// instructions; nt takes 0.13 ms of that back, non-temporal loads in the sweeps nothing.
#ifndef PHK_CK_NT
#define PHK_CK_NT 1  // A/B: 0 = plain stores; 2 = non-temporal loads in the sweeps as well
#endif
template <typename real>
__device__ __forceinline__ void ck_store(real* p, real v) {
#if PHK_CK_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
template <typename real>
__device__ __forceinline__ real ck_load(const real* p) {
#if PHK_CK_NT >= 2
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}

struct SeqMap {
    int64_t bb, ss, seq, oseq;
    bool active;     // this group holds a sequence of its own
    bool idle_wave;  // no group of this wave does: the wave must leave (see fwd_kernel)
};
// The one-state-per-lane kernels (four lane groups per wave) want the four sequences of a wave on ONE chunk: over a
// range of whole chunks every chunk's particles are padded to a multiple of four groups (the padding groups repeat the
// chunk's last particle).
template <typename real, int K, int R>
__host__ __device__ inline int64_t launch_groups(const KArgs& A) {  // lane groups a launch needs (grid = this / groups per workgroup)
    const int64_t seq_hi = A.seq_end > 0 ? A.seq_end : A.B * A.S;
    if (has_dense<real, K, R>() && A.B > 0 && (A.B & 3) != 0 && A.seq_begin % A.B == 0 && seq_hi % A.B == 0) {
        const int64_t ns = (seq_hi - A.seq_begin) / A.B;
        return ns * ((A.B + 3) & ~int64_t(3));
    }
    return seq_hi - A.seq_begin;
}
template <typename real, int K, int R>
__device__ __forceinline__ SeqMap map_group(const KArgs& A) {
    const int64_t seq_hi = A.seq_end > 0 ? A.seq_end : A.B * A.S;
    const int64_t g = (int64_t)blockIdx.x * (blockDim.x / R) + threadIdx.x / R;               // this group, within the launch
    const int64_t gw = (int64_t)blockIdx.x * (blockDim.x / R) + (int64_t)((threadIdx.x & ~63u) / R);  // the wave's first group
    SeqMap m;
    if (has_dense<real, K, R>() && (A.B & 3) != 0 && A.seq_begin % A.B == 0 && seq_hi % A.B == 0) {  // whole chunks, padded
        const int64_t s0 = A.seq_begin / A.B, ns = (seq_hi - A.seq_begin) / A.B, nbp = (A.B + 3) & ~int64_t(3);
        const int64_t sc = g / nbp, r = g - sc * nbp;
        m.idle_wave = gw >= ns * nbp;
        m.active = sc < ns && r < A.B;
        m.ss = s0 + (sc < ns ? sc : ns - 1);
        m.bb = m.active ? r : A.B - 1;
    } else {
        m.idle_wave = A.seq_begin + gw >= seq_hi;
        m.active = A.seq_begin + g < seq_hi;
        const int64_t lin = m.active ? A.seq_begin + g : seq_hi - 1;
        m.ss = lin / A.B;
        m.bb = lin - m.ss * A.B;
    }
    m.seq = m.ss * A.B + m.bb;
    m.oseq = m.bb * A.S + m.ss;
    return m;
}

// (MR: the one-state-per-lane kernels of a handle whose rows hold runs of missing sites -- an accessibility mask -- step over eight
// missing sites with one operator, Lane::half_step.  Kernels of their own, picked by the launcher from KArgs::mask_runs, compiled
// from the same text: fwd_kernel_body.inc / bscan_kernel_body.inc.)
template <typename real, int K, int R, int T, int NRM, bool CKPT>
__global__ __launch_bounds__(NT_MAX, (scan_waves_per_simd<real, K, R>())) void fwd_kernel(KArgs A) {
#define PHK_KERNEL_MR false
#include "fwd_kernel_body.inc"
#undef PHK_KERNEL_MR
}
template <typename real, int K, int R, int T, int NRM, bool CKPT>
__global__ __launch_bounds__(NT_MAX, (scan_waves_per_simd<real, K, R>())) void fwd_kernel_mr(KArgs A) {
#define PHK_KERNEL_MR true
#include "fwd_kernel_body.inc"
#undef PHK_KERNEL_MR
}

// ---------------------------------------------------------------------------------------------
// kernel 2: backward sweep over the checkpointed blocks.  The T alpha vectors of a block (and the
// scales of its rescale sites) live in REGISTERS: both site loops are fully unrolled, so there is
// no LDS block store and no address arithmetic; LDS holds only the emission table.
// ---------------------------------------------------------------------------------------------
// waves per SIMD the backward kernel is compiled for: 2 where T*SPL alphas + state fit 256 VGPRs
#ifndef PHK_SEG_WAVES
#define PHK_SEG_WAVES 0  // A/B: waves per SIMD the segment sweep is compiled for where it owns <= 4 states per lane (0: as the serial sweep)
#endif
template <typename real, int K, int R, int T, bool SEG = false>
constexpr int bwd_waves_per_simd() {
    if (SEG && PHK_SEG_WAVES > 0 && T == 8 && (K / R) * (int)sizeof(real) <= 16) return PHK_SEG_WAVES;
    return (T * (K / R) * (int)sizeof(real) <= 256) ? 2 : 1;
}

// PHK_SWEEP_V2 keeps w_i = e_i .* beta_{i+1} of the T sites of a block until the forward re-run has used them.  With 32
// bytes of state per lane (8 float32 states: K = 16 at R = 2, K = 32 at R = 4, the throughput variants; or 4 float64
// states) those 64 registers are what pushes
// the kernel over its 256-VGPR budget (two waves per SIMD), and the compiler's answer is scratch inside the block loop.
// PARKED of the T vectors therefore live in LDS instead: explicit 16-byte stores in the beta pass, loads one site ahead in
// the forward pass, in the thread's own slice behind its emission table (no barrier: nothing is shared).
// Sweeps in the folded form (see bwd_kernel, SFOLD): the beta-first body in float32.  Their posterior-mass rows --
// touched at het / missing sites only -- live in LDS behind the emission table, [3][EROW] reals indexed by the site's
// code like the table itself.
template <typename real, int K, int R, int T, int NRM>
constexpr bool sweep_folds() { return PHK_SWEEP_FOLD != 0 && (sizeof(real) == 4 || PHK_FOLD_F64 != 0) && PHK_SWEEP_V2 != 0 && NRM > 1 && T == 8 && T * (K / R) * (int)sizeof(real) <= 256; }
#ifndef PHK_PARK_HREG
#define PHK_PARK_HREG 2  // ... of the folded bodies with PHK_HET_REGS: 68 floats of LDS per thread = 69,632 B per 256-thread workgroup, two
                         // workgroups per CU (the launcher raises the kernel's dynamic-LDS limit: launch.hip, bwd_rtn).  Every parked
                         // vector costs ~0.25 ms of cfg2's backward phase (1 / 2 / 3 / 5 parked: spills in the block loop / 19.2 / 19.4 /
                         // 20.9 ms, profiles/r06_ab_experiments.txt item 2); 2 is the fewest that keeps the block loop out of scratch
#endif
#ifndef PHK_PARK_HREG2
#define PHK_PARK_HREG2 5  // ... with PHK_HET_REGS == 2 (the missing row and its mass in registers as well, no mass rows in LDS): 28 + 40 floats
#endif
#ifndef PHK_PARK_UNFOLDED
#define PHK_PARK_UNFOLDED 3  // ... of the bodies that keep their emission rows (float64): round 3's figure
#endif
template <typename real, int K, int R, int T, int NRM>
constexpr int sweep_parked() {
    if (!(PHK_SWEEP_V2 != 0 && NRM > 1 && T == 8 && (K / R) * (int)sizeof(real) == 32)) return 0;
    return sweep_folds<real, K, R, T, NRM>() ? (PHK_HET_REGS == 2 ? PHK_PARK_HREG2 : PHK_HET_REGS != 0 ? PHK_PARK_HREG : PHK_PARK) : PHK_PARK_UNFOLDED;
}
// reals per thread of the backward kernel's LDS slice: emission table + mass rows + parked vectors, the stride in 16-byte
// units odd (the 16 lanes of a ds_read_b128 group then fall on 16 different bank quads, see Lane::ETAB_STRIDE)
template <typename real, int K, int R, int T, int NRM>
constexpr int sweep_lds_stride() {
    using L = Lane<real, K, R>;
    constexpr int park = sweep_parked<real, K, R, T, NRM>() * 2 * L::NP;
    constexpr int gtab = (sweep_folds<real, K, R, T, NRM>() && PHK_HET_REGS != 2) ? 3 * L::EROW : 0;
    if (park + gtab == 0) return L::ETAB_STRIDE;
    constexpr int per16 = 16 / (int)sizeof(real);
    int n = L::ETAB_STRIDE + gtab + park;
    n = (n + per16 - 1) / per16 * per16;
    if ((n / per16) % 2 == 0) n += per16;
    return n;
}

// SEG = false: one unit per sequence sweeps all blocks (blockIdx.y == 0) and writes the gradient.
// SEG = true : blockIdx.y picks a unit of A.seg_blocks blocks; it starts from the beta-scan's value
//   at its right edge, stores its partial sums once in its own slot of KArgs::part (no atomics: the
//   sums of a sequence are added up in unit order by grad_finalize_kernel, so the result does not
//   depend on the order the units ran in), and grad_finalize_kernel writes the gradient.  Unit 0 also covers every segment up to the one holding the warm-up
//   boundary (the correction there makes those segments depend on each other).
template <typename real, int K, int R, int T, int NRM, bool SEG>
__global__ __launch_bounds__(NT_MAX, (bwd_waves_per_simd<real, K, R, T, SEG>())) void bwd_kernel(KArgs A) {
    using L = Lane<real, K, R>;
    using V = typename L::V;
    constexpr int SPL = L::SPL, NP = L::NP;
    static_assert(T <= 16 && 16 % T == 0 && T % NRM == 0, "block / rescale schedule");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x;
#if PHK_SERIAL_PRIO
    if constexpr (!SEG) __builtin_amdgcn_s_setprio(PHK_SERIAL_PRIO);
#endif
#if PHK_SEG_PRIO
    if constexpr (SEG) __builtin_amdgcn_s_setprio(PHK_SEG_PRIO);
#endif
    const int64_t nseq = A.B * A.S;
    const int64_t seq_hi = A.seq_end > 0 ? A.seq_end : nseq;
    const int64_t gid = A.seq_begin + (int64_t)blockIdx.x * (blockDim.x / R) + tid / R;
    const bool active = gid < seq_hi;
    const int64_t seq = active ? gid : seq_hi - 1;
    const int rank = tid & (R - 1);
    const int64_t ss = seq / A.B, bb = seq - ss * A.B;  // chunk-major order (see SeqMap)
    const int64_t oseq = bb * A.S + ss;                   // ... the caller's, for grad

    L lane;
    V pi[NP];
    const real* prm = (const real*)A.params + bb * A.pstride_b + ss * A.pstride_s;
    constexpr int PARKN = sweep_parked<real, K, R, T, NRM>();
    constexpr bool SFOLD = sweep_folds<real, K, R, T, NRM>();
    real* etab = (real*)smem_raw + (size_t)tid * sweep_lds_stride<real, K, R, T, NRM>();
    real* const gtab = etab + L::ETAB_STRIDE;                     // SFOLD: posterior-mass rows [3][EROW], by code
    real* const park = gtab + ((SFOLD && PHK_HET_REGS != 2) ? 3 * L::EROW : 0);  // parked w vectors
    lane.load(prm, rank, etab, pi);
    if constexpr (SFOLD && PHK_HET_REGS != 2) {
#pragma unroll
        for (int j = 0; j < 3 * L::EROW; ++j) gtab[j] = real(0);
    }
    // Folded form (kernels with the beta-first body: SFOLD).  The emission of a site multiplies the COLUMN index of A,
    // the one b, d and v carry: with (b, d, v) <- emis0 .* (b, d, v) and the table rows 1, emis1 / emis0, 1 / emis0
    // (Lane::fold_emissions) both recursions keep their form -- it is the same HMM written with hom emission 1 -- and
    //   gb, gd, gv   accumulate the gradient w.r.t. the FOLDED factors (d/db_j = emis0_j d/db'_j, applied at the end),
    //   gu           is unchanged (suf(v' .* w_hat) = suf(v .* w)),
    //   the posterior mass p .* w of a site is booked at het and missing sites only (rows 1 and 2 of gtab), and the hom
    //                row is the remainder: the mass summed over ALL sites is b' gb + d' gd + v' gv, state by state.
    // A site that is hom in every lane of the wave -- with sequences stored chunk-major a wave reads one or two
    // observation rows, so that is 96-98 % of the sites at 1 % hets -- then needs no emission row, no row .* beta, no
    // row .* (A alpha) and no mass at all.  A sequence folds whenever its own ratios exist (Lane::emissions_foldable), so
    // what is computed for it does not depend on its wave; one that cannot keeps its emissions in the table and books
    // its hom mass in row 0, and its wave treats every site as "not hom" (the folded lanes then multiply by rows that are
    // exactly 1: the same bits).  The flag goes to aux for the segment sweep's finalize.
    // Kernels without that body (16-site blocks, per-site rescaling, 16 states per lane) run on the folded model all the
    // same (Lane::try_fold) and book the hom mass directly; only the conversion at the end differs (aux.folded = 2).
    const real* pfb = prefold_block<real>(A, bb, ss);
    const bool folded = lane.try_fold(pfb != nullptr ? pfb + rank * SPL : nullptr);  // (per sequence)
    const int g0code = (SFOLD && folded) ? 2 : 0;
    const bool wave_folded = SFOLD && __all(folded) != 0;
    // HREG (round 6): what a het site needs in the hot body -- its ratio row and the row its posterior mass is booked in --
    // lives in registers; see the hot body.  A wave that holds a sequence which cannot fold takes the general body.
    constexpr bool HREG = SFOLD && PHK_HET_REGS != 0;
    constexpr bool HREG2 = SFOLD && PHK_HET_REGS == 2;  // ... and so does what a missing site needs: no mass rows in LDS at all
    V rhet[NP], rmis[NP];
    bool two_rows = false;  // every lane of the wave reads the observation row of lane 0 or that of lane 63
    if constexpr (HREG) {
#pragma unroll
        for (int h = 0; h < NP; ++h) {
            rhet[h] = *(const V*)(etab + L::EROW + 2 * h);
            rmis[h] = *(const V*)(etab + 2 * L::EROW + 2 * h);
        }
        const int sA = __builtin_amdgcn_readfirstlane((int)ss), sB = __builtin_amdgcn_readlane((int)ss, 63);
        two_rows = __all((int)ss == sA || (int)ss == sB) != 0;
    }
    if constexpr (SEG) {
        if (active && rank == 0 && blockIdx.y == 0) A.aux[seq].folded = folded ? (SFOLD ? 1 : 2) : 0;
    }
    const uint32_t* words = A.packed + checked_row(A, ss) * A.Lw;
    const real* ck = (const real*)A.ckpt;

    V beta[NP], gb[NP], gd[NP], gu[NP], gv[NP], g0[NP], g1[NP];
    const real inv_end = (real)A.aux[seq].inv_end;
    // (block bookkeeping in 32-bit wave-uniform integers and stepped pointers, as in fwd_kernel: a block of the folded body
    // is ~720 vector instructions, and 64-bit index products per block were 5 % of that and the segment sweep's last
    // scratch accesses)
    const int nblk = (int)((A.Ltot + T - 1) / T);
    // block range [blk_lo, blk_hi) of this unit
    int blk_lo = 0, blk_hi = nblk;
    if constexpr (SEG) {
        const int G = A.seg_blocks;
        const int segW = A.W > 0 ? (int)((A.W - 1) / T) / G : 0;  // segment holding the warm-up boundary
        const int seg = segW + (int)blockIdx.y;
        blk_lo = blockIdx.y == 0 ? 0 : seg * G;
        blk_hi = (seg + 1) * G < nblk ? (seg + 1) * G : nblk;
    }
#pragma unroll
    for (int h = 0; h < NP; ++h) {
        beta[h] = splat<real>(inv_end);
        gb[h] = gd[h] = gu[h] = gv[h] = g0[h] = g1[h] = splat<real>(real(0));
    }
    if constexpr (SEG) {
        if (blk_hi < nblk) {
            // adjoint of the scaled alpha at the unit's right edge t: b*_t 2^{E_t} / P,
            // b*_t = bseg 2^{fseg},  P = sum(alpha_L) 2^{E_L}
            const int64_t sb = blk_hi / A.seg_blocks;
            const real* src = (const real*)A.bseg + (sb * nseq + seq) * K + rank * SPL;
            const int ex = A.fseg[sb * nseq + seq] + A.eseg[sb * nseq + seq] - A.aux[seq].e_end;
            const real f = ldexp_(inv_end, ex);
#pragma unroll
            for (int h = 0; h < NP; ++h) beta[h] = splat<real>(real(0));
#pragma unroll
            for (int i = 0; i < SPL; ++i) L::set(beta, i, src[i] * f);
            // b* (beta scan) and alpha (forward kernel) come from two differently rounded recursions,
            // so sum_i alpha_t[i] beta_t[i] is 1 only up to their accumulated round-off -- a global
            // factor on beta that does not decay along the sweep and shows up undamped in d ll/d pi
            // after the warm-up correction.  Normalise the seed against the forward kernel's own
            // alpha at this edge (its checkpoint): the identity then holds exactly where the sweep starts.
            const real* ca = ck + (int64_t)blk_hi * nseq * K + L::ck_lane(nseq, seq, rank);
            real dot = real(0);
#pragma unroll
            for (int i = 0; i < SPL; ++i) dot = fma_(ca[L::ck_elem(i, nseq)], L::get(beta, i), dot);
            dot = lane.g.sum(dot);
            // (a seed whose product with the checkpoint has underflowed cannot be normalised: the same remedy as for
            // the forward kernel's rescale interval -- raise the flag, the host re-evaluates with per-site rescaling)
            if (!(dot > real(0)) && active && rank == 0 && A.risk != nullptr) atomicOr(A.risk, FLAG_UNDERFLOW);
            const V inv = splat<real>(dot > real(0) ? real(1) / dot : real(1));
#pragma unroll
            for (int h = 0; h < NP; ++h) beta[h] = beta[h] * inv;
        }
    }
    constexpr bool F64ACC = sizeof(real) == 4 || SEG;  // fold partial sums into the f64 buffer
    double* gacc = A.gacc + seq * 6 * K + rank * SPL;
    real* part = nullptr;  // this unit's own slot of partial sums (units >= 1 of the segment sweep)
    if constexpr (SEG) {
        if (blockIdx.y > 0)
            part = (real*)A.part + (((int64_t)blockIdx.y - 1) * (seq_hi - A.seq_begin) + (seq - A.seq_begin)) * 6 * K + rank * SPL;
    }
    int since_flush = 0;

    real anext[SPL];
    // observation words, one word ahead of the (descending) block that needs it
    int widx = -1;
    uint32_t wcur = 0, wprev = 0;
    int e_next = 0;  // block exponent, requested one block ahead like the checkpoint (the beta-first body needs it first thing)
    const int64_t ck_step = nseq * K;
    const real* ckq = ck + L::ck_lane(nseq, seq, rank);  // this lane's piece of the checkpoint that is requested NEXT
    const int16_t* ebq = A.eblk + seq;                   // ... and its block exponent
    if (blk_hi > blk_lo) {
        ckq += (int64_t)(blk_hi - 1) * ck_step;
        ebq += (int64_t)(blk_hi - 1) * nseq;
#pragma unroll
        for (int i = 0; i < SPL; ++i) anext[i] = ck_load(&ckq[L::ck_elem(i, nseq)]);
        e_next = *ebq;
        ckq -= ck_step;
        ebq -= nseq;

        widx = ((blk_hi - 1) * T) >> 4;
        wcur = words[widx];
        wprev = words[widx > 0 ? widx - 1 : 0];
    }
    // Two kinds of blocks.  Nearly all are full blocks with no warm-up boundary inside: they take a straight-line body
    // (HOT).  The partial last block of a row and the block holding the warm-up boundary take the general body with its
    // per-site tests.  The hot blocks run in a loop of their OWN, in runs that end where a general block or a fold of the
    // partial sums into float64 is due: with both bodies and the fold in one loop the register allocator paid for their
    // union on the hot path (moves and scratch reloads at the join of the paths, once per block).
    // (kernels whose lanes own more states -- 16 in float32, 8 in float64 -- keep the general body: a block of w vectors
    // would take them past 256 registers into AGPR copies plus scratch, the regime tests/test_layout.py keeps out)
    constexpr bool HOT_V2 = PHK_SWEEP_V2 != 0 && NRM > 1 && T == 8 && T * SPL * (int)sizeof(real) <= 256;
    constexpr bool HOT_SL = !HOT_V2 && bwd_straight_line<real, K, R, T>() && bwd_waves_per_simd<real, K, R, T, SEG>() <= 2;
    constexpr bool HOT = HOT_V2 || HOT_SL;
    const int blkW = A.W > 0 ? (int)((A.W - 1) / T) : -1;          // block holding the warm-up boundary
    const int blk_part = (A.Ltot % T) != 0 ? nblk - 1 : -1;      // partial block (the row's last), if any
    // block prologue: step the observation words, take the block's checkpoint (requested one block earlier) and
    // request the next one, fetch the exponent total the forward kernel took out of the block
    auto enter = [&](const int blk, V (&al0)[NP], int& e_fwd, uint32_t& codes) {
        const int t0 = blk * T;  // (site index inside the row: < 2^31)
        if ((t0 >> 4) != widx) {  // stepped into the previous word
            widx = t0 >> 4;
            wcur = wprev;
            wprev = words[widx > 0 ? widx - 1 : 0];
        }
#pragma unroll
        for (int h = 0; h < NP; ++h) al0[h] = splat<real>(real(0));
#pragma unroll
        for (int i = 0; i < SPL; ++i) L::set(al0, i, anext[i]);
        e_fwd = e_next;
        if (blk > blk_lo) {  // prefetch the previous block's checkpoint and exponent under this block's arithmetic
                             // (two blocks ahead: +2 ms at cfg2, profiles/r05_ab_experiments.txt item 13)
#pragma unroll
            for (int i = 0; i < SPL; ++i) anext[i] = ck_load(&ckq[L::ck_elem(i, nseq)]);
            e_next = *ebq;
            ckq -= ck_step;
            ebq -= nseq;
        }
        codes = wcur >> (2 * (t0 & 15));
    };
    // fold the partial sums into float64 (units >= 1 of the segment sweep: store them, exactly once, at their left edge)
    auto flush = [&]() {
        since_flush = 0;
        if constexpr (SFOLD && !HREG2) {  // the mass rows of the folded form live in LDS (see the hot body): row g0code and the het row
#pragma unroll
            for (int h = 0; h < NP; ++h) {
                g0[h] = *(const V*)(gtab + g0code * L::EROW + 2 * h);
                if constexpr (HREG) g1[h] = g1[h] + *(const V*)(gtab + 1 * L::EROW + 2 * h);  // (registers: hot body; LDS row: general body)
                else g1[h] = *(const V*)(gtab + 1 * L::EROW + 2 * h);
            }
#pragma unroll
            for (int j = 0; j < 3 * L::EROW; ++j) gtab[j] = real(0);
        }
        if (active) {
#pragma unroll
            for (int i = 0; i < SPL; ++i) {
                if constexpr (SEG) {
                    if (blockIdx.y > 0) {  // one flush per unit (a unit is shorter than FLUSH_SITES): plain stores
                        part[0 * K + i] = L::get(gb, i);
                        part[1 * K + i] = L::get(gd, i);
                        part[2 * K + i] = L::get(gu, i);
                        part[3 * K + i] = L::get(gv, i);
                        part[4 * K + i] = L::get(g0, i);
                        part[5 * K + i] = L::get(g1, i);
                        continue;
                    }
                }
                {
                    gacc[0 * K + i] += (double)L::get(gb, i);
                    gacc[1 * K + i] += (double)L::get(gd, i);
                    gacc[2 * K + i] += (double)L::get(gu, i);
                    gacc[3 * K + i] += (double)L::get(gv, i);
                    gacc[4 * K + i] += (double)L::get(g0, i);
                    gacc[5 * K + i] += (double)L::get(g1, i);
                }
            }
        }
#pragma unroll
        for (int h = 0; h < NP; ++h) gb[h] = gd[h] = gu[h] = gv[h] = g0[h] = g1[h] = splat<real>(real(0));
    };
    // A sequence with a block too steep for the unscaled hot body (see HOT_BLOCK_MIN_EXP) makes its whole wave take
    // the general body for every block: decided once per wave from what the forward kernel recorded, so that the hot
    // loop itself carries no test (a per-block test with an exit from the hot loop cost 1.6-2.9 ms at cfg2: the exit
    // edge brought scratch accesses back into the block)
    constexpr int HOT_MIN_EXP = sizeof(real) == 4 ? HOT_BLOCK_MIN_EXP_F32 : HOT_BLOCK_MIN_EXP_F64;
#if PHK_EXP_NO_STEEP  // timing-only diagnostic builds: every wave takes the hot body (results of steep sequences are wrong)
    const bool wave_steep = false;
#else
    const bool wave_steep = HOT_V2 && __any(A.aux[seq].eb_min < HOT_MIN_EXP);
#endif
    int blk = blk_hi - 1;
    int budget = A.loop_budget[SEG ? 3 : 1];  // blocks this wave may still sweep (see KArgs::loop_budget)
    while (blk >= blk_lo) {
        if (__builtin_expect(budget <= 0, 0)) {
            report_overrun(A, SEG ? 3 : 2, seq, blk);
            return;
        }
        if (!HOT || wave_steep || (HREG && !wave_folded) || blk == blkW || blk == blk_part) {
            --budget;
            const int64_t t0 = (int64_t)blk * T;
            V al[T + 1][NP];  // al[i] = alpha entering site i of the block; al[ns] = alpha leaving it
            real sc[T / NRM];
            int e_fwd, e_run = 0;  // exponent total of the block: forward kernel's, and the re-run's below
            uint32_t codes;
            enter(blk, al[0], e_fwd, codes);
            const int ns = (int)((A.Ltot - t0) < T ? (A.Ltot - t0) : T);
#pragma unroll
            for (int i = 0; i < T; ++i) {
                if (i < ns) {
#pragma unroll
                    for (int h = 0; h < NP; ++h) al[i + 1][h] = al[i][h];
                    real s;
                    V e[NP];
                    lane.emis((codes >> (2 * i)) & 3, e);
                    e_run += lane.fwd_site(al[i + 1], e, s, rescale_after<NRM>(i));
                    if (rescale_after<NRM>(i)) sc[i / NRM] = s;
                }
            }
            {
                const V fx = splat<real>(ldexp_(real(1), e_run - e_fwd));
#pragma unroll
                for (int h = 0; h < NP; ++h) beta[h] = beta[h] * fx;
            }
            V m0[NP], m1[NP];  // SFOLD: this block's mass rows (g0code and het), added to the LDS rows below
#pragma unroll
            for (int h = 0; h < NP; ++h) m0[h] = m1[h] = splat<real>(real(0));
#pragma unroll
            for (int i = T - 1; i >= 0; --i) {
                if (i < ns) {
                    if (t0 + i + 1 == A.W) {
                        // warm-up boundary: ll = log P(o_1..L) - log P(o_1..W); al[i+1] is alpha_W in
                        // this kernel's own scaling, beta = d log P(o_1..L) / d alpha_W, and the second
                        // term contributes -1 / sum(alpha_W).  In exact arithmetic sum_i alpha_W beta = 1,
                        // so the difference has no component along the direction that survives the
                        // remaining W sites; in floating point beta carries a global factor 1 + eps
                        // (round-off of L - W steps) whose image does NOT decay and would dominate
                        // d ll / d pi.  Dividing beta by the measured sum restores the identity.
                        V prod[NP];
#pragma unroll
                        for (int h = 0; h < NP; ++h) prod[h] = al[i + 1][h] * beta[h];
                        const real sab = lane.total(prod), sa = lane.total(al[i + 1]);
                        const V is = splat<real>(sab > real(0) ? real(1) / sab : real(1));
                        const V ic = splat<real>(real(1) / sa);
#pragma unroll
                        for (int h = 0; h < NP; ++h) beta[h] = beta[h] * is - ic;
                        if (A.W == A.Ltot) {  // nothing is scored: the two terms are the same number
#pragma unroll
                            for (int h = 0; h < NP; ++h) beta[h] = splat<real>(real(0));
                        }
                    }
                    const bool SC = rescale_after<NRM>(i);
                    V e[NP];
                    lane.emis((codes >> (2 * i)) & 3, e);
                    if constexpr (SFOLD && !HREG2)
                        lane.bwd_site(al[i], al[i + 1], beta, e, (codes >> (2 * i)) & 3, SC ? sc[i / NRM] : real(1), SC,
                                      gb, gd, gu, gv, m0, m1, g0code);
                    else
                        lane.bwd_site(al[i], al[i + 1], beta, e, (codes >> (2 * i)) & 3, SC ? sc[i / NRM] : real(1), SC,
                                      gb, gd, gu, gv, g0, g1, g0code);
                }
            }
            if constexpr (SFOLD && !HREG2) {
#pragma unroll
                for (int h = 0; h < NP; ++h) {
                    V* r0 = (V*)(gtab + g0code * L::EROW + 2 * h);
                    V* r1 = (V*)(gtab + 1 * L::EROW + 2 * h);
                    *r0 = *r0 + m0[h];
                    *r1 = *r1 + m1[h];
                }
            }
            if constexpr (F64ACC) {
                since_flush += ns;
                if ((since_flush >= FLUSH_SITES && part == nullptr) || blk == blk_lo) flush();
            }
            --blk;
            continue;
        }
        if constexpr (HOT) {
            // a run of hot blocks: down to the unit's left edge, the warm-up block or the next fold, whichever comes first
            int stop = blk_lo;
            if (blkW >= blk_lo && blkW < blk) stop = blkW + 1;
            if constexpr (F64ACC) {
                if (part == nullptr) {
                    const int left = since_flush < FLUSH_SITES ? (FLUSH_SITES - since_flush + T - 1) / T : 1;
                    if (blk - left + 1 > stop) stop = blk - left + 1;
                }
            }
            if (blk - stop + 1 > budget) stop = blk - budget + 1;  // (the budget bounds the run: no test inside the block loop)
            budget -= blk - stop + 1;
            const int first = blk;
#if PHK_ASM_RUN
            // K = 16, R = 2, float32: the whole run of hot blocks as ONE hand-written instruction sequence (scripts/gen_sweep_asm.py):
            // an all-hom block runs without the sixteen per-site tests of the body below, a mixed block with them, under one
            // register plan -- what the compiler cannot do at the join of two C++ bodies.  Same arithmetic, operation for
            // operation; waves that hold more than two observation rows keep the C++ loop.
            if constexpr (HREG && !HREG2 && sizeof(real) == 4 && K == 16 && R == 2) {
                if (A.asm_run != 0 && two_rows && blk >= stop) {
                    // (the statement's scalar operands must be SGPRs to the compiler's divergence analysis, which cannot see that
                    // `stop` -- it depends on whether this unit owns a partial-sum slot -- is the same in every lane)
                    auto uni64 = [](int64_t x) {
                        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)x);
                        const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)x >> 32));
                        return ((uint64_t)hi << 32) | lo;
                    };
                    uint64_t asm_ckq = (uint64_t)ckq, asm_ebq = (uint64_t)ebq;
                    const uint64_t asm_words = (uint64_t)words;
                    const uint32_t asm_lds = (uint32_t)(uintptr_t)etab;
                    const uint64_t asm_ckstep = uni64(-(ck_step * (int64_t)sizeof(real)));
                    const uint64_t asm_ebstep = uni64(-(nseq * (int64_t)sizeof(int16_t)));
                    const uint64_t asm_ckpiece = uni64(nseq * 4 * (int64_t)sizeof(real));
                    int asm_blk = __builtin_amdgcn_readfirstlane(blk), asm_widx = __builtin_amdgcn_readfirstlane(widx);
                    const int asm_stop = __builtin_amdgcn_readfirstlane(stop), asm_blklo = __builtin_amdgcn_readfirstlane(blk_lo);
#include "sweep_run_k16r2.inc"
                    blk = asm_blk;
                    widx = asm_widx;
                    ckq = (const real*)asm_ckq;
                    ebq = (const int16_t*)asm_ebq;
                }
            }
#endif
            for (; blk >= stop; --blk) {
                int e_fwd;
                uint32_t codes;
                if constexpr (HOT_V2) {
                    V al0[NP];
                    enter(blk, al0, e_fwd, codes);
                    // Full block, no warm-up boundary inside, beta pass FIRST.  Of the four scans a site's gradient rows and
                    // recursions need on the alpha side (pre(u.*a), suf(a): both also needed by the forward step itself) and
                    // on the beta side (suf(v.*w): needed by gu; pre(b.*w): by the beta recursion only), the order
                    // "re-run alpha, then sweep beta" computes the alpha-side pair twice.  Here beta is swept through the
                    // block first, keeping w_i = e_i .* beta_{i+1} of every site in registers; the forward re-run then
                    // accumulates all six rows as it goes and recomputes only suf(v.*w_i):
                    //     gd += w.*a   gb += w.*suf(a)   gv += w.*pre(u.*a)   gu += a.*suf(v.*w)   g0/g1 += p.*w  (= a_next.*beta_next)
                    // Both passes run UNSCALED from the checkpoint a_0: with a'_i = a_i 2^{-E_i} (E_i: what the forward kernel's
                    // rescales took out before site i of the block; E_0 = 0) the adjoint of a'_i is beta_i 2^{E_i}, every product
                    // above is unchanged, and only beta at the block's right edge needs the block total: a_T = a'_T 2^{-E_T}, so
                    // beta'_T = beta_T 2^{-E_T} (E_T = e_fwd <= 0 as recorded; the forward kernel flags blocks whose total would
                    // leave the float range).
                    {
                        const V fx = splat<real>(ldexp_(real(1), -e_fwd));
#pragma unroll
                        for (int h = 0; h < NP; ++h) beta[h] = beta[h] * fx;
                    }
                    // w[i] of site i: sites [T - PARKN, T) in LDS (produced first by the beta pass, used last by the forward
                    // pass: the longest lifetimes), the others in registers
                    constexpr int NREG = T - PARKN;
                    V w[NREG > 0 ? NREG : 1][NP];
                    if constexpr (HREG) {
                        // Folded form, round 6.  As below (ONE body, a wave-uniform branch per site and pass on one bit of the
                        // block's non-hom mask), but what a HET site needs is in registers: its ratio row `rhet` and the
                        // row `g1` its posterior mass goes to.  Round 5 fetched the ratio row from the LDS table and booked
                        // the mass by a read-modify-write of an LDS row: three LDS round trips in the dependency chain of
                        // every het site, which cost such a site as much again as the 90 vector instructions of the site
                        // itself (cfg2: 20.1 ms at 1 % hets, 23.8 at 10 %).  Het lanes now multiply and accumulate under
                        // the exec mask (a lane whose row is hom at this site does nothing: two-row waves), and only
                        // MISSING sites -- runs of masked windows in real data, 1 % of the bench rows -- still go through
                        // the LDS rows, behind a second wave-uniform bit.  The sixteen registers are paid for by parking
                        // three of the block's w vectors in LDS instead of one (sweep_parked).
                        // The block's masks: with sequences stored chunk-major a wave holds one or two observation rows
                        // whenever B >= the wave's sequence count, so the OR over the wave of the block's 16 code bits is
                        // the OR of lane 0's and lane 63's -- two readlanes instead of round 5's eight ballots.
                        uint32_t u16;
                        {
                            const uint32_t c16 = codes & 0xffffu;
                            if (__builtin_expect(two_rows, 1)) {
                                u16 = (uint32_t)__builtin_amdgcn_readfirstlane((int)c16) | (uint32_t)__builtin_amdgcn_readlane((int)c16, 63);
                            } else {
                                u16 = 0u;
                                if (__any(c16 != 0u)) {
#pragma unroll
                                    for (int i = 0; i < 2 * T; ++i) u16 |= __any((c16 >> i) & 1u) ? (1u << i) : 0u;
                                }
                            }
                        }
#if PHK_EXP_HET_MODE == 1  // timing-only diagnostic builds (results are wrong): no site is treated as het or missing
                        u16 = 0u;
#endif
                        const uint32_t nm = (u16 >> 1) & 0x5555u;  // bit 2i: site i is missing in some lane
                        const uint32_t nh = (u16 | (u16 >> 1)) & 0x5555u;  // bit 2i: site i is not hom in some lane
#pragma unroll
                        for (int i = T - 1; i >= 0; --i) {
                            V wi[NP];
#pragma unroll
                            for (int h = 0; h < NP; ++h) wi[h] = beta[h];
                            // (a use on the hot path: read only in branches the compiler knows to be cold, the ratio row is
                            // its first choice for scratch -- reloaded inside every het branch, worse than the LDS table)
#pragma unroll
                            for (int h = 0; h < NP; ++h) {
                                asm volatile("" : : "v"(rhet[h]));
                                if constexpr (HREG2) asm volatile("" : : "v"(rmis[h]));
                            }
                            if (__builtin_expect((nh >> (2 * i)) & 1u, 0)) {
                                const uint32_t c = (codes >> (2 * i)) & 3u;
#if PHK_EXP_HET_MODE == 2  // timing-only: the branch is taken, its body is one instruction
                                asm volatile("s_nop 0");
#else
                                if (c == 1u) {
#pragma unroll
                                    for (int h = 0; h < NP; ++h) wi[h] = beta[h] * rhet[h];
                                }
#endif
                                if (PHK_EXP_HET_MODE != 2 && __builtin_expect((nm >> (2 * i)) & 1u, 0)) {
                                    if constexpr (HREG2) {
                                        if (c == 2u) {
#pragma unroll
                                            for (int h = 0; h < NP; ++h) wi[h] = beta[h] * rmis[h];
                                        }
                                    } else {
                                        V e[NP];
                                        lane.emis(c == 2u ? 2 : 0, e);  // (row 0 of a folded lane: ones)
#pragma unroll
                                        for (int h = 0; h < NP; ++h) wi[h] = wi[h] * e[h];
                                    }
                                }
                            }
                            if (i >= NREG) {
#pragma unroll
                                for (int h = 0; h < NP; ++h) *(V*)(park + ((i - NREG) * NP + h) * 2) = wi[h];
                            } else {
#pragma unroll
                                for (int h = 0; h < NP; ++h) w[i][h] = wi[h];
                            }
                            V svw[NP], pbw[NP], cb;
                            lane.scans_adj(wi, svw, pbw, cb);
#pragma unroll
                            for (int h = 0; h < NP; ++h) beta[h] = lane.beta_prev(h, wi, svw, pbw, cb);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        V a[NP];
#pragma unroll
                        for (int h = 0; h < NP; ++h) a[h] = al0[h];
                        V wc[NP];  // w of the current site; a parked one is requested one site ahead
                        if (NREG > 0) {
#pragma unroll
                            for (int h = 0; h < NP; ++h) wc[h] = w[0][h];
                        } else {
#pragma unroll
                            for (int h = 0; h < NP; ++h) wc[h] = *(const V*)(park + h * 2);
                        }
                        // (the forward pass tests copies of the masks the compiler cannot connect with the beta pass's: it
                        // otherwise keeps every site's bit as a 64-bit lane mask from one pass to the other, seven scalar
                        // instructions per site for the two tests instead of four)
                        uint32_t nh_f = nh, nm_f = nm;
                        asm volatile("" : "+s"(nh_f), "+s"(nm_f));
#pragma unroll
                        for (int i = 0; i < T; ++i) {
                            V wn[NP];
                            if (i + 1 < T) {
                                if (i + 1 >= NREG) {
#pragma unroll
                                    for (int h = 0; h < NP; ++h) wn[h] = *(const V*)(park + ((i + 1 - NREG) * NP + h) * 2);
                                } else {
#pragma unroll
                                    for (int h = 0; h < NP; ++h) wn[h] = w[i + 1 < NREG ? i + 1 : 0][h];
                                }
                            }
                            V pre[NP], suf[NP], svw[NP], t[NP];
                            lane.scans(a, pre, suf);
                            lane.suffix_vw(wc, svw);
#pragma unroll
                            for (int h = 0; h < NP; ++h) {
                                gd[h] = fma2<real>(wc[h], a[h], gd[h]);
                                gb[h] = fma2<real>(wc[h], suf[h], gb[h]);
                                gv[h] = fma2<real>(wc[h], pre[h], gv[h]);
                                gu[h] = fma2<real>(a[h], svw[h], gu[h]);
                                V tt = lane.d[h] * a[h];
                                tt = fma2<real>(lane.v[h], pre[h], tt);
                                t[h] = fma2<real>(lane.b[h], suf[h], tt);
                            }
                            // (the four rows are pinned here: see the round-5 body below; the mass rows and the ratio rows for the
                            // reason given in the beta pass)
#pragma unroll
                            for (int h = 0; h < NP; ++h) {
                                asm volatile("" : "+v"(gd[h]), "+v"(gb[h]), "+v"(gv[h]), "+v"(gu[h]), "+v"(g1[h]) : "v"(rhet[h]));
                                if constexpr (HREG2) asm volatile("" : "+v"(g0[h]) : "v"(rmis[h]));
                            }
                            if (__builtin_expect((nh_f >> (2 * i)) & 1u, 0)) {
                                const uint32_t c = (codes >> (2 * i)) & 3u;
#if PHK_EXP_HET_MODE == 2
                                asm volatile("s_nop 0");
#else
                                if (c == 1u) {  // het lanes: the site's posterior mass p .* w, then the ratio row
#pragma unroll
                                    for (int h = 0; h < NP; ++h) {
                                        g1[h] = fma2<real>(t[h], wc[h], g1[h]);
                                        t[h] = t[h] * rhet[h];
                                    }
                                }
#endif
                                if (PHK_EXP_HET_MODE != 2 && HREG2 && __builtin_expect((nm_f >> (2 * i)) & 1u, 0)) {
                                    if (c == 2u) {  // missing lanes: the same with their own rows
#pragma unroll
                                        for (int h = 0; h < NP; ++h) {
                                            g0[h] = fma2<real>(t[h], wc[h], g0[h]);
                                            t[h] = t[h] * rmis[h];
                                        }
                                    }
                                }
                                if (PHK_EXP_HET_MODE != 2 && !HREG2 && __builtin_expect((nm_f >> (2 * i)) & 1u, 0)) {
                                    // a missing site: mass into LDS row 2 and the row 1 / emis0 (a lane that is not missing
                                    // here adds to row 0, which nobody reads, and multiplies by ones)
                                    const int row = c == 2u ? 2 : 0;
                                    real* grow = gtab + row * L::EROW;
                                    const V* erow = (const V*)(etab + row * L::EROW);
#pragma unroll
                                    for (int h = 0; h < NP; ++h) {
                                        V* gr = (V*)(grow + 2 * h);
                                        *gr = fma2<real>(t[h], wc[h], *gr);
                                        t[h] = t[h] * erow[h];
                                    }
                                }
                            }
                            if (i + 1 < T) {
#pragma unroll
                                for (int h = 0; h < NP; ++h) {
                                    a[h] = t[h];
                                    wc[h] = wn[h];
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    } else if constexpr (SFOLD) {
                        // Folded form: ONE body.  A site that is hom in every lane of the wave multiplies by no emission
                        // row and books no posterior mass (its row is 1 and its mass is part of the remainder, see where
                        // the wave folds its emissions); any other site does both behind a wave-uniform branch on one bit
                        // of `nh`.  With sequences stored chunk-major a wave reads one or two observation rows, so at 1 %
                        // hets 98 % of the sites take the fall-through.  (Round 4 compiled the hom block as a second
                        // instance of the whole body: the allocator paid for the union of the two at their join, 712 ->
                        // 964 B of scratch and reloads in the hot loop; the branches here enclose a few instructions each.)
                        uint32_t nh = 0xffu;  // bit i: site i of the block is not hom in some lane (a wave with an unfolded sequence: every site)
                        if (wave_folded) {
                            const uint32_t m16 = ((codes | (codes >> 1)) & 0x5555u);  // this lane: bit 2i set = site i not hom
                            nh = 0u;
                            if (__builtin_expect(__any(m16 != 0u), 0)) {
#pragma unroll
                                for (int i = 0; i < T; ++i) nh |= __any((m16 >> (2 * i)) & 1u) ? (1u << i) : 0u;
                            }
                        }
#pragma unroll
                        for (int i = T - 1; i >= 0; --i) {
                            V wi[NP];
#pragma unroll
                            for (int h = 0; h < NP; ++h) wi[h] = beta[h];
                            if (__builtin_expect((nh >> i) & 1u, 0)) {
                                V e[NP];
                                lane.emis((codes >> (2 * i)) & 3, e);
#pragma unroll
                                for (int h = 0; h < NP; ++h) wi[h] = beta[h] * e[h];
                            }
                            if (i >= NREG) {
#pragma unroll
                                for (int h = 0; h < NP; ++h) *(V*)(park + ((i - NREG) * NP + h) * 2) = wi[h];
                            } else {
#pragma unroll
                                for (int h = 0; h < NP; ++h) w[i][h] = wi[h];
                            }
                            V svw[NP], pbw[NP], cb;
                            lane.scans_adj(wi, svw, pbw, cb);
#pragma unroll
                            for (int h = 0; h < NP; ++h) beta[h] = lane.beta_prev(h, wi, svw, pbw, cb);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        V a[NP];
#pragma unroll
                        for (int h = 0; h < NP; ++h) a[h] = al0[h];
                        V wc[NP];  // w of the current site; a parked one is requested one site ahead
                        if (NREG > 0) {
#pragma unroll
                            for (int h = 0; h < NP; ++h) wc[h] = w[0][h];
                        } else {
#pragma unroll
                            for (int h = 0; h < NP; ++h) wc[h] = *(const V*)(park + h * 2);
                        }
#pragma unroll
                        for (int i = 0; i < T; ++i) {
                            V wn[NP];
                            if (i + 1 < T) {
                                if (i + 1 >= NREG) {
#pragma unroll
                                    for (int h = 0; h < NP; ++h) wn[h] = *(const V*)(park + ((i + 1 - NREG) * NP + h) * 2);
                                } else {
#pragma unroll
                                    for (int h = 0; h < NP; ++h) wn[h] = w[i + 1 < NREG ? i + 1 : 0][h];
                                }
                            }
                            V pre[NP], suf[NP], svw[NP], t[NP];
                            lane.scans(a, pre, suf);
                            lane.suffix_vw(wc, svw);
#pragma unroll
                            for (int h = 0; h < NP; ++h) {
                                gd[h] = fma2<real>(wc[h], a[h], gd[h]);
                                gb[h] = fma2<real>(wc[h], suf[h], gb[h]);
                                gv[h] = fma2<real>(wc[h], pre[h], gv[h]);
                                gu[h] = fma2<real>(a[h], svw[h], gu[h]);
                                V tt = lane.d[h] * a[h];
                                tt = fma2<real>(lane.v[h], pre[h], tt);
                                t[h] = fma2<real>(lane.b[h], suf[h], tt);
                            }
                            // (the four rows are pinned here: left alone the compiler sinks their updates past the branch
                            // below to the end of the block and keeps every site's scans in scratch until then)
#pragma unroll
                            for (int h = 0; h < NP; ++h) asm volatile("" : "+v"(gd[h]), "+v"(gb[h]), "+v"(gv[h]), "+v"(gu[h]));
                            if (__builtin_expect((nh >> i) & 1u, 0)) {
                                // the site's posterior mass p .* w into the row of its code (a hom lane of a folded wave
                                // adds to row 0, which nobody reads), then the emission row
                                const int code = (codes >> (2 * i)) & 3;
                                // (plain C++ since the gradient rows are pinned above: both rows are requested at once; round 5
                                // first wrote this pair by pair in inline asm, suspecting the sixteen registers of the two rows
                                // in flight of the spills that were the compiler's sinking -- 0.8 ms slower at 10 % hets)
                                real* grow = gtab + code * L::EROW;
                                const V* erow = (const V*)(etab + code * L::EROW);
#pragma unroll
                                for (int h = 0; h < NP; ++h) {
                                    V* gr = (V*)(grow + 2 * h);
                                    *gr = fma2<real>(t[h], wc[h], *gr);
                                    t[h] = t[h] * erow[h];
                                }
                            }
                            if (i + 1 < T) {
#pragma unroll
                                for (int h = 0; h < NP; ++h) {
                                    a[h] = t[h];
                                    wc[h] = wn[h];
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    } else {
                    V ec[NP];
                    lane.emis((codes >> (2 * (T - 1))) & 3, ec);
#pragma unroll
                    for (int i = T - 1; i >= 0; --i) {
                        V en[NP];
                        if (i > 0) lane.emis((codes >> (2 * (i - 1))) & 3, en);
                        V wi[NP];
#pragma unroll
                        for (int h = 0; h < NP; ++h) wi[h] = beta[h] * ec[h];
                        if (i >= NREG) {
#pragma unroll
                            for (int h = 0; h < NP; ++h) *(V*)(park + ((i - NREG) * NP + h) * 2) = wi[h];
                        } else {
#pragma unroll
                            for (int h = 0; h < NP; ++h) w[i][h] = wi[h];
                        }
                        V svw[NP], pbw[NP], cb;
                        lane.scans_adj(wi, svw, pbw, cb);
#pragma unroll
                        for (int h = 0; h < NP; ++h) beta[h] = lane.beta_prev(h, wi, svw, pbw, cb);
                        if (i > 0) {
#pragma unroll
                            for (int h = 0; h < NP; ++h) ec[h] = en[h];
                        }
#if PHK_DS_FIRST
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 400, 0);
#endif
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    V a[NP];
#pragma unroll
                    for (int h = 0; h < NP; ++h) a[h] = al0[h];
                    lane.emis(codes & 3, ec);
                    V wc[NP];  // w of the current site; a parked one is requested one site ahead
                    if (NREG > 0) {
#pragma unroll
                        for (int h = 0; h < NP; ++h) wc[h] = w[0][h];
                    } else {
#pragma unroll
                        for (int h = 0; h < NP; ++h) wc[h] = *(const V*)(park + h * 2);
                    }
#pragma unroll
                    for (int i = 0; i < T; ++i) {
                        V en[NP], wn[NP];
                        if (i + 1 < T) {
                            lane.emis((codes >> (2 * (i + 1))) & 3, en);
                            if (i + 1 >= NREG) {
#pragma unroll
                                for (int h = 0; h < NP; ++h) wn[h] = *(const V*)(park + ((i + 1 - NREG) * NP + h) * 2);
                            } else {
#pragma unroll
                                for (int h = 0; h < NP; ++h) wn[h] = w[i + 1 < NREG ? i + 1 : 0][h];
                            }
                        }
                        const int code = (codes >> (2 * i)) & 3;
                        const V f1 = splat<real>(code == 1 ? real(1) : real(0));
                        const V f0 = splat<real>(code == 0 ? real(1) : real(0));
                        V pre[NP], suf[NP], svw[NP];
                        lane.scans(a, pre, suf);
                        lane.suffix_vw(wc, svw);
#pragma unroll
                        for (int h = 0; h < NP; ++h) {
                            gd[h] = fma2<real>(wc[h], a[h], gd[h]);
                            gb[h] = fma2<real>(wc[h], suf[h], gb[h]);
                            gv[h] = fma2<real>(wc[h], pre[h], gv[h]);
                            gu[h] = fma2<real>(a[h], svw[h], gu[h]);
                            V t = lane.d[h] * a[h];
                            t = fma2<real>(lane.v[h], pre[h], t);
                            t = fma2<real>(lane.b[h], suf[h], t);
                            const V m = t * wc[h];
                            g1[h] = fma2<real>(f1, m, g1[h]);
                            g0[h] = fma2<real>(f0, m, g0[h]);
                            if (i + 1 < T) a[h] = t * ec[h];
                        }
                        if (i + 1 < T) {
#pragma unroll
                            for (int h = 0; h < NP; ++h) {
                                ec[h] = en[h];
                                wc[h] = wn[h];
                            }
                        }
#if PHK_DS_FIRST
                        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 400, 0);
#endif
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    }
                } else {
                    V al[T + 1][NP];
                    real sc[T / NRM];
                    int e_run = 0;
                    enter(blk, al[0], e_fwd, codes);
                    // full block, no warm-up boundary inside: straight-line code for all 2T site steps.
                    // re-run the block forward (bit-identical to kernel 1), keeping every alpha; the
                    // emission row of the NEXT step is always in flight while the current one computes,
                    // and a scheduling barrier after every step keeps the live ranges of its temporaries
                    // from being stretched over its neighbours (the kernel sits at the 256-VGPR budget).
                    V ec[NP];
                    lane.emis(codes & 3, ec);
#pragma unroll
                    for (int i = 0; i < T; ++i) {
                        V en[NP];
                        if (i + 1 < T) lane.emis((codes >> (2 * (i + 1))) & 3, en);
#pragma unroll
                        for (int h = 0; h < NP; ++h) al[i + 1][h] = al[i][h];
                        real s;
                        e_run += lane.fwd_site(al[i + 1], ec, s, rescale_after<NRM>(i));
                        if (rescale_after<NRM>(i)) sc[i / NRM] = s;
                        if (i + 1 < T) {  // (the last forward step and the first backward step share a row)
#pragma unroll
                            for (int h = 0; h < NP; ++h) ec[h] = en[h];
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    {   // beta is the adjoint of the forward kernel's alpha at the block end; the re-run's
                        // alpha is that times 2^(e_fwd - e_run) (0 when both kernels are the same variant)
                        const V fx = splat<real>(ldexp_(real(1), e_run - e_fwd));
#pragma unroll
                        for (int h = 0; h < NP; ++h) beta[h] = beta[h] * fx;
                    }
                    // sweep it backwards
#pragma unroll
                    for (int i = T - 1; i >= 0; --i) {
                        V en[NP];
                        if (i > 0) lane.emis((codes >> (2 * (i - 1))) & 3, en);
                        const bool SC = rescale_after<NRM>(i);
                        lane.bwd_site(al[i], al[i + 1], beta, ec, (codes >> (2 * i)) & 3, SC ? sc[i / NRM] : real(1), SC, gb,
                                      gd, gu, gv, g0, g1);
                        if (i > 0) {
#pragma unroll
                            for (int h = 0; h < NP; ++h) ec[h] = en[h];
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if constexpr (F64ACC) {
                since_flush += (int)(first - blk) * T;
                if ((since_flush >= FLUSH_SITES && part == nullptr) || blk < blk_lo) flush();
            }
        }
    }
    if (!active) return;
    if constexpr (SEG) {
        if (blk_lo == 0) {
#pragma unroll
            for (int i = 0; i < SPL; ++i) A.bpi[seq * K + rank * SPL + i] = (double)L::get(beta, i);
        }
        return;  // grad_finalize_kernel turns gacc / bpi into the gradient
    }
    if constexpr (SFOLD && !HREG2 && !F64ACC) {  // (never flushed: the mass rows are still in LDS)
#pragma unroll
        for (int h = 0; h < NP; ++h) {
            g0[h] = *(const V*)(gtab + g0code * L::EROW + 2 * h);
            if constexpr (HREG) g1[h] = g1[h] + *(const V*)(gtab + 1 * L::EROW + 2 * h);  // (registers: hot body; LDS row: general body)
            else g1[h] = *(const V*)(gtab + 1 * L::EROW + 2 * h);
        }
    }
    // d ll / d theta (or theta * that), rows b,d,u,v,emis0,emis1,pi
    real* out = (real*)A.grad + oseq * 7 * K + rank * SPL;
    const bool dl = A.grad_dlog != 0;
#pragma unroll
    for (int i = 0; i < SPL; ++i) {
        double vb, vd, vu, vv, v0, v1;
        if constexpr (F64ACC) {
            vb = gacc[0 * K + i]; vd = gacc[1 * K + i]; vu = gacc[2 * K + i];
            vv = gacc[3 * K + i]; v0 = gacc[4 * K + i]; v1 = gacc[5 * K + i];
        } else {
            vb = L::get(gb, i); vd = L::get(gd, i); vu = L::get(gu, i);
            vv = L::get(gv, i); v0 = L::get(g0, i); v1 = L::get(g1, i);
        }
        if (folded) {  // folded form -> the caller's (see where the sequence folds its emissions)
            const double e0 = (double)prm[4 * K + rank * SPL + i], e1 = (double)prm[5 * K + rank * SPL + i];
            const double lb = vb * (double)L::get(lane.b, i), ld = vd * (double)L::get(lane.d, i), lv = vv * (double)L::get(lane.v, i);
            // hom row: all sites - het sites - missing sites where the hot body booked those (SFOLD), else booked directly
            const double mhom = SFOLD ? lb + ld + lv - v1 - v0 : v0;
            out[0 * K + i] = (real)(dl ? lb : vb * e0);
            out[1 * K + i] = (real)(dl ? ld : vd * e0);
            out[2 * K + i] = (real)(dl ? vu * (double)L::get(lane.u, i) : vu);
            out[3 * K + i] = (real)(dl ? lv : vv * e0);
            out[4 * K + i] = (real)(dl ? mhom : mhom / e0);
            out[5 * K + i] = (real)(dl ? v1 : v1 / e1);
        } else {
        out[0 * K + i] = (real)(dl ? vb * (double)L::get(lane.b, i) : vb);
        out[1 * K + i] = (real)(dl ? vd * (double)L::get(lane.d, i) : vd);
        out[2 * K + i] = (real)(dl ? vu * (double)L::get(lane.u, i) : vu);
        out[3 * K + i] = (real)(dl ? vv * (double)L::get(lane.v, i) : vv);
        out[4 * K + i] = (real)(dl ? v0 : v0 / (double)etab[0 * L::EROW + L::SLOT(i)]);
        out[5 * K + i] = (real)(dl ? v1 : v1 / (double)etab[1 * L::EROW + L::SLOT(i)]);
        }
        // pi is re-read here rather than kept in registers through the sweep (the kernel sits at its VGPR budget)
        out[6 * K + i] = (real)(dl ? (double)L::get(beta, i) * (double)prm[6 * K + rank * SPL + i] : (double)L::get(beta, i));
    }
}

// ---------------------------------------------------------------------------------------------
// kernel 3 (small batches): the unnormalised backward recursion alone, b*_L = 1,
// b*_{t-1} = A (e_t .* b*_t), with its own power-of-two scaling; the state at every segment start is
// kept.  It does not depend on the forward kernel, so the two run concurrently on two streams, and
// the segments of a sequence can then be swept in parallel by bwd_kernel<..., SEG = true>.
// ---------------------------------------------------------------------------------------------
template <typename real, int K, int R, int NRM>
__global__ __launch_bounds__(NT_MAX, (scan_waves_per_simd<real, K, R>())) void bscan_kernel(KArgs A, int64_t seg_sites, void* bseg_out, int32_t* fseg_out) {
#define PHK_KERNEL_MR false
#include "bscan_kernel_body.inc"
#undef PHK_KERNEL_MR
}
template <typename real, int K, int R, int NRM>
__global__ __launch_bounds__(NT_MAX, (scan_waves_per_simd<real, K, R>())) void bscan_kernel_mr(KArgs A, int64_t seg_sites, void* bseg_out, int32_t* fseg_out) {  // (see fwd_kernel_mr)
#define PHK_KERNEL_MR true
#include "bscan_kernel_body.inc"
#undef PHK_KERNEL_MR
}

// gacc [B*S, 6, K] (unit 0) + part [units - 1, range, 6, K] (the other units, added in unit order:
// the result does not depend on the order in which the units ran) + bpi [B*S, K] -> grad [B*S, 7, K]
template <typename real>
struct alignas(4 * sizeof(real)) Quad {
    real x[4];
};
template <typename real>
__global__ void grad_finalize_kernel(KArgs A, int K, int units) {
    // one thread per FOUR consecutive gradient entries (sequence, row, states k .. k + 3; K is a multiple of 4): consecutive
    // threads read consecutive 16-byte pieces of every unit's slot.  The partial sums are what bounds this kernel (cfg2: 777 MB,
    // prod: 188 MB); with one entry per thread (rounds 3-5; round 3 had one thread per (sequence, state) walking six rows) it
    // read them at 3.1 TB/s.  Each entry is summed in unit order as before: the same bits.
    const int64_t seq_hi = A.seq_end > 0 ? A.seq_end : A.B * A.S;
    const int64_t nloc = seq_hi - A.seq_begin;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int Q = 7 * K / 4;  // quads per sequence
    if (idx >= nloc * Q) return;
    const int64_t sl = idx / Q;
    const int rk = 4 * (int)(idx - sl * Q);  // r * K + k of the first entry
    const int64_t seq = A.seq_begin + sl;
    const int64_t ss = seq / A.B, bb = seq - ss * A.B;  // chunk-major order (see SeqMap); grad in the caller's
    const real* p = (const real*)A.params + bb * A.pstride_b + ss * A.pstride_s;
    real* out = (real*)A.grad + (bb * A.S + ss) * 7 * K;
    const bool dl = A.grad_dlog != 0;
    Quad<real> o;
    if (rk >= 6 * K) {  // the pi row
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double bp = A.bpi[seq * K + (rk - 6 * K) + i];
            o.x[i] = (real)(dl ? bp * (double)p[rk + i] : bp);
        }
        *(Quad<real>*)(out + rk) = o;
        return;
    }
    double sum[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) sum[i] = A.gacc[seq * 6 * K + rk + i];
    const real* pt = (const real*)A.part + sl * 6 * K + rk;
#pragma unroll 4
    for (int u = 1; u < units; ++u) {
        const Quad<real> q = *(const Quad<real>*)pt;
#pragma unroll
        for (int i = 0; i < 4; ++i) sum[i] += (double)q.x[i];
        pt += nloc * 6 * K;
    }
    const bool folded = A.aux[seq].folded != 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (folded) o.x[i] = (real)sum[i];  // sums of the folded form: grad_unfold_kernel converts them in place
        else if (rk < 4 * K) o.x[i] = (real)(dl ? sum[i] * (double)p[rk + i] : sum[i]);
        else o.x[i] = (real)(dl ? sum[i] : sum[i] / (double)p[rk + i]);
    }
    *(Quad<real>*)(out + rk) = o;
}

// Segment sweep, folded form (see bwd_kernel): the six rows grad_finalize_kernel left as plain sums -> the caller's
// gradient, one thread per (sequence, state).
template <typename real>
__global__ void grad_unfold_kernel(KArgs A, int K) {
    const int64_t seq_hi = A.seq_end > 0 ? A.seq_end : A.B * A.S;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (seq_hi - A.seq_begin) * K) return;
    const int64_t seq = A.seq_begin + idx / K;
    const int k = (int)(idx % K);
    if (!A.aux[seq].folded) return;
    const int64_t ss = seq / A.B, bb = seq - ss * A.B;
    const real* p = (const real*)A.params + bb * A.pstride_b + ss * A.pstride_s;
    real* out = (real*)A.grad + (bb * A.S + ss) * 7 * K;
    const bool dl = A.grad_dlog != 0;
    const double e0 = (double)p[4 * K + k], e1 = (double)p[5 * K + k];
    const double vb = (double)out[0 * K + k], vd = (double)out[1 * K + k], vu = (double)out[2 * K + k], vv = (double)out[3 * K + k];
    const double vm = (double)out[4 * K + k], v1 = (double)out[5 * K + k];
    // folded factors exactly as the kernels formed them (float products)
    // (... or took from the pre-folded block)
    const real* pf = prefold_block<real>(A, bb, ss);
    const double fb = pf != nullptr ? (double)pf[0 * K + k] : (double)(p[0 * K + k] * p[4 * K + k]);
    const double fd = pf != nullptr ? (double)pf[1 * K + k] : (double)(p[1 * K + k] * p[4 * K + k]);
    const double fv = pf != nullptr ? (double)pf[2 * K + k] : (double)(p[3 * K + k] * p[4 * K + k]);
    const double lb = vb * fb, ld = vd * fd, lv = vv * fv;
    const double mhom = A.aux[seq].folded == 1 ? lb + ld + lv - v1 - vm : vm;  // (2: the hom mass was booked directly)
    out[0 * K + k] = (real)(dl ? lb : vb * e0);
    out[1 * K + k] = (real)(dl ? ld : vd * e0);
    out[2 * K + k] = (real)(dl ? vu * (double)p[2 * K + k] : vu);
    out[3 * K + k] = (real)(dl ? lv : vv * e0);
    out[4 * K + k] = (real)(dl ? mhom : mhom / e0);
    out[5 * K + k] = (real)(dl ? v1 : v1 / e1);
}

// ---------------------------------------------------------------------------------------------
// upload-time re-pack: int8 {-1,0,1,(>1 clipped to 1; gpu.py:108-110)} -> 2-bit codes
// (a plain, non-template kernel: emitted only in the translation unit that defines PHK_WITH_PACK)
// ---------------------------------------------------------------------------------------------
#ifdef PHK_WITH_PACK
constexpr double FLAG_OVERRUN_WEIGHT = 4096.0;
// stream-ordered hand-over of the flag word, one double per bit (so that the bits survive the SUM
// all-reduce of the float64 buffer they ride in): dst[0] = underflow risk, dst[1] = bad index; flags = 0
__global__ void take_flags_kernel(int* flags, double* dst) {
    const int w = atomicExch(flags, 0);
    dst[0] = (w & FLAG_UNDERFLOW) ? 1.0 : 0.0;
    // (a loop that ran out of its budget rides in the second slot with a weight of its own: sums over ranks stay decodable)
    dst[1] = ((w & FLAG_BAD_INDEX) ? 1.0 : 0.0) + ((w & FLAG_OVERRUN) ? FLAG_OVERRUN_WEIGHT : 0.0);
}

// Dense hom-run operators for the one-state-per-lane kernels (K = 16, float32): one workgroup per parameter block,
// thread (r, c) owns entry [r][c].  M_h = A diag(emis0), A[r][c] = b[c] (r > c), d[c] (r == c), u[r] v[c] (r < c)
// (hmm.py:52-65 written out as a matrix); powers 1..8 and 16 (and the eighth power of the missing-site step) accumulated in float64 from float64 factors and rounded
// once -- the same operator is applied thousands of times along a row, so an error in it acts like a perturbation of
// the parameters (coherent over the sequence), not like round-off.  Both forms are written lane-major:
//   ops_f[blk][n][i][j] = (M_h^n)[j][i]   (forward kernel: lane i holds column i)
//   ops_b[blk][n][i][j] = (M_h^n)[i][j]   (beta scan: lane i holds row i)
__global__ __launch_bounds__(256) void dense_ops_kernel(const float* __restrict__ params, int64_t pstride_b, int64_t pstride_s,
                                                        int64_t S_blocks, float* __restrict__ ops_f, float* __restrict__ ops_b,
                                                        const float* __restrict__ prefold, int64_t pfstride_b, int64_t pfstride_s) {
    __shared__ double Mx[5][16][17];  // M, M^2, M^3, M^4, M^8
    const int64_t q = blockIdx.x;
    const int64_t bb = q / S_blocks, ss = q - bb * S_blocks;
    const float* p = params + bb * pstride_b + ss * pstride_s;
    const int t = threadIdx.x, r = t >> 4, c = t & 15;
    const double ur = p[2 * 16 + r];
    // (from the factors as the structured steps of the same sequence use them: folded in float32 where the sequence
    // folds -- Lane::try_fold -- so that dense and structured steps advance the same model)
    const float e0f = p[4 * 16 + c];
    const bool fold = PHK_FOLD != 0 && __syncthreads_and(e0f > RATIO_MIN_EMIS0);
    // (... or took pre-folded: the same bits the structured steps of the sequence run on)
    const float* pf = prefold != nullptr ? prefold + bb * pfstride_b + ss * pfstride_s : nullptr;
    const double bc = fold ? (pf ? (double)pf[0 * 16 + c] : (double)(p[0 * 16 + c] * e0f)) : (double)p[0 * 16 + c] * (double)e0f;
    const double dc = fold ? (pf ? (double)pf[1 * 16 + c] : (double)(p[1 * 16 + c] * e0f)) : (double)p[1 * 16 + c] * (double)e0f;
    const double vc = fold ? (pf ? (double)pf[2 * 16 + c] : (double)(p[3 * 16 + c] * e0f)) : (double)p[3 * 16 + c] * (double)e0f;
    const double m = r > c ? bc : (r == c ? dc : ur * vc);
    float* of = ops_f + q * DENSE_OPS_FLOATS;
    float* ob = ops_b + q * DENSE_OPS_FLOATS;
    auto emit = [&](int n, double val) {
        of[(n * 16 + c) * 16 + r] = (float)val;
        ob[(n * 16 + r) * 16 + c] = (float)val;
    };
    auto mul = [&](int x, int y) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) acc = __builtin_fma(Mx[x][r][k], Mx[y][k][c], acc);
        return acc;
    };
    Mx[0][r][c] = m;
    emit(0, m);
    __syncthreads();
    const double p2 = mul(0, 0);
    Mx[1][r][c] = p2;
    emit(1, p2);
    __syncthreads();
    const double p3 = mul(1, 0), p4 = mul(1, 1);
    Mx[2][r][c] = p3;
    Mx[3][r][c] = p4;
    emit(2, p3);
    emit(3, p4);
    __syncthreads();
    const double p5 = mul(3, 0), p6 = mul(3, 1), p7 = mul(3, 2), p8 = mul(3, 3);
    Mx[4][r][c] = p8;
    emit(4, p5);
    emit(5, p6);
    emit(6, p7);
    emit(7, p8);
    __syncthreads();
    emit(8, mul(4, 4));
    // eight missing sites: (M_h diag(rmis))^8 with the ratio as the kernels hold it (Lane::load_dense / fold_emissions)
    const double rm = (fold && pf) ? (double)pf[4 * 16 + c] : (double)(float)(1.0 / (double)e0f);
    __syncthreads();
    Mx[0][r][c] = m * rm;
    __syncthreads();
    const double a2 = mul(0, 0);
    Mx[1][r][c] = a2;
    __syncthreads();
    const double a4 = mul(1, 1);
    Mx[2][r][c] = a4;
    __syncthreads();
    emit(9, mul(2, 2));
}

// (nonhom: optional device counters [2] -- the sites that are not hom, the one property of the data the static plan looks at
// (phk_api.hip static_plan), and the 8-site halves that are missing throughout (phk_api.hip mask_runs))
__global__ void pack_kernel(const int8_t* __restrict__ data, int64_t N, int64_t L, uint32_t* __restrict__ out,
                            int64_t Lw, unsigned long long* __restrict__ nonhom) {
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
    if (nonhom != nullptr) {
        const int64_t left = L - w * 16;  // sites of this word inside the row
        uint32_t nh = (word | (word >> 1)) & 0x55555555u;
        if (left < 16) nh &= left > 0 ? ((1u << (2 * left)) - 1u) : 0u;
        const int n = __builtin_popcount(nh);
        if (n) atomicAdd(nonhom, (unsigned long long)n);
        // [1]: halves that are missing throughout (inside the row: padding decodes as missing too)
        const int nm = ((word & 0xffffu) == 0xAAAAu && left >= 8 ? 1 : 0) + ((word >> 16) == 0xAAAAu && left >= 16 ? 1 : 0);
        if (nm) atomicAdd(nonhom + 1, (unsigned long long)nm);
    }
}
#endif  // PHK_WITH_PACK

}  // namespace phk
