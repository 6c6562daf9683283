// Two translation units per (real, K): compiled with -DPHK_REAL=float|double -DPHK_K=<K> -DPHK_SUFFIX=f32_16 ... and
// -DPHK_PART=1 (forward kernel) or -DPHK_PART=2 (beta scan, backward kernel, finalize), so
// that the variants build in parallel and each part gets its own compiler flags (see the Makefile); for float K = 16
// a third, -DPHK_PART=3, holds the one-state-per-lane forward kernel and beta scan (see PHK_LAT_PART below).  Without
// PHK_PART everything is compiled into one object.  Dispatches (R, T) to the template instantiations of psmc_kernels.hip.
#include "psmc_kernels.hip"

#ifndef PHK_REAL
#error "compile with -DPHK_REAL=float|double -DPHK_K=<K> -DPHK_SUFFIX=<tag>"
#endif

#ifndef PHK_PART
#define PHK_PART 0
#endif
#define PHK_FWD_PART (PHK_PART == 0 || PHK_PART == 1)
#define PHK_BWD_PART (PHK_PART == 0 || PHK_PART == 2)
// PHK_PART == 3 (float, K = 16 only; -DPHK_LAT_SPLIT=1 tells parts 1 and 2 that it exists): the one-state-per-lane
// forward kernel and beta scan, the latency-bound pair of the small-batch plans.  Their control flow is wave-uniform
// (scalar observation codes) and branch-heavy, and a taken branch costs a lone wave ~40 cycles, so this part is
// compiled with -mllvm -structurizecfg-skip-uniform-regions (see the Makefile): the structurizer otherwise turns
// every multi-way uniform branch into a chain of flag tests with several taken branches per case.
#ifndef PHK_LAT_SPLIT
#define PHK_LAT_SPLIT 0
#endif
#define PHK_LAT_PART (PHK_PART == 3)

#define PHK_CAT2(a, b) a##b
#define PHK_CAT(a, b) PHK_CAT2(a, b)

namespace phk {

using real_t = PHK_REAL;
constexpr int KK = PHK_K;

static size_t lds_bytes(int R, int nt, bool mask_runs = false) {
    const int spl = KK / R;
    const int erow = 2 * ((spl + 1) / 2);
    const int w = (int)(sizeof(real_t) / 4);
    int raw = 3 * erow;  // per-thread emission table, padded as Lane::ETAB_STRIDE
    const int dw = raw * w;
    if (dw % 4 == 0 && (dw / 4) % 2 == 0) raw += 4 / w;
    size_t bytes = (size_t)raw * nt * sizeof(real_t);
    if (mask_runs && has_dense<real_t, KK, 16>() && R == 16) bytes += (size_t)nt * DENSE_Q8_STRIDE * sizeof(float);  // DenseOps::q8 (the *_mr kernels)
    return bytes;
}

#if PHK_FWD_PART || PHK_LAT_PART
template <int R, int T, int NRM>
static hipError_t fwd_rtn(bool ckpt, const KArgs& a, int nt, hipStream_t st) {
    const int64_t nseq = launch_groups<real_t, KK, R>(a);  // (the one-state-per-lane layout pads every chunk's particles: map_group)
    const int spb = nt / R;  // sequences per workgroup
    const dim3 grid((unsigned)((nseq + spb - 1) / spb)), block(nt);
    const size_t lds = lds_bytes(R, nt);
    if constexpr (has_dense<real_t, KK, R>() && NRM == 4) {  // rows with runs of missing sites: the kernels that step over them (see fwd_kernel_mr)
        if (a.mask_runs) {
            const size_t lds_mr = lds_bytes(R, nt, true);
            if (ckpt)
                hipLaunchKernelGGL((fwd_kernel_mr<real_t, KK, R, T, NRM, true>), grid, block, lds_mr, st, a);
            else
                hipLaunchKernelGGL((fwd_kernel_mr<real_t, KK, R, T, NRM, false>), grid, block, lds_mr, st, a);
            return hipGetLastError();
        }
    }
    if (ckpt)
        hipLaunchKernelGGL((fwd_kernel<real_t, KK, R, T, NRM, true>), grid, block, lds, st, a);
    else
        hipLaunchKernelGGL((fwd_kernel<real_t, KK, R, T, NRM, false>), grid, block, lds, st, a);
    return hipGetLastError();
}

#endif
// float64 sweeps that are not compiled (256 VGPRs + AGPR copies + scratch: see phk_api.hip, valid_Rf): the
// serial sweep with more than 4 states per lane, the segment sweep with more than 4 (8 at K = 16, where the compiler's
// report shows no scratch; at K = 32 the same layout came back with AGPR copies plus 20 B of scratch in round 3)
#ifdef PHK_EXP_F64_SPL16  // diagnostic builds only (scripts/diag_fenced_variants.py): compile every variant
template <int R, bool SEG>
constexpr bool f64_sweep_ok() { return true; }
#else
template <int R, bool SEG>
constexpr bool f64_sweep_ok() { return sizeof(real_t) == 4 || KK / R <= 4 || (SEG && KK == 16 && KK / R == 8); }
#endif

#if PHK_BWD_PART
template <int R, int T, int NRM>
static hipError_t bwd_rtn(const KArgs& a, int units, int nt, hipStream_t st) {
    const int64_t nseq = (a.seq_end > 0 ? a.seq_end : a.B * a.S) - a.seq_begin;
    // variants that park part of a block in LDS (psmc_kernels.hip, sweep_parked) run as workgroups of two waves: four
    // of them share a CU's 160 KB and a workgroup stays under the 64 KB a launch gets without asking
    // A slice above the 64 KB a launch gets without asking (round 6: the folded float32 sweeps park three w vectors, 76
    // floats per thread = 77,824 B per 256-thread workgroup) is asked for per kernel, once: two such workgroups still fit
    // a CU's 160 KB, i.e. two waves per SIMD as the kernels are compiled for.  Round 5 fell back to 128-thread workgroups
    // there, whose two waves land on ONE SIMD (profiles/r05_ab_experiments.txt item 2).
    constexpr int stride = sweep_lds_stride<real_t, KK, R, T, NRM>();
    constexpr size_t lds256 = (size_t)stride * 256 * sizeof(real_t);
    constexpr bool big_lds = lds256 > 65536 && 2 * lds256 <= 163840;
    if ((size_t)stride * nt * sizeof(real_t) > 65536 && !big_lds) nt = 128;
    const size_t lds = (size_t)stride * nt * sizeof(real_t);
    const int spb = nt / R;
    const dim3 block(nt);
    if (units <= 0) {  // one serial sweep per sequence
        if constexpr (f64_sweep_ok<R, false>()) {
            if constexpr (big_lds) {
                static const hipError_t attr = hipFuncSetAttribute((const void*)bwd_kernel<real_t, KK, R, T, NRM, false>,
                                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds256);
                if (attr != hipSuccess) return attr;
            }
            const dim3 grid((unsigned)((nseq + spb - 1) / spb));
            hipLaunchKernelGGL((bwd_kernel<real_t, KK, R, T, NRM, false>), grid, block, lds, st, a);
        } else {
            return hipErrorInvalidValue;
        }
    } else {  // `units` independent segments per sequence
        if constexpr (f64_sweep_ok<R, true>()) {
            if constexpr (big_lds) {
                static const hipError_t attr = hipFuncSetAttribute((const void*)bwd_kernel<real_t, KK, R, T, NRM, true>,
                                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds256);
                if (attr != hipSuccess) return attr;
            }
            const dim3 grid((unsigned)((nseq + spb - 1) / spb), (unsigned)units);
            hipLaunchKernelGGL((bwd_kernel<real_t, KK, R, T, NRM, true>), grid, block, lds, st, a);
        } else {
            return hipErrorInvalidValue;
        }
    }
    return hipGetLastError();
}

#endif
#if PHK_BWD_PART || PHK_LAT_PART
template <int R, int NRM>
static hipError_t bscan_rn(const KArgs& a, int64_t seg_sites, void* bseg, int32_t* fseg, int nt, hipStream_t st) {
    const int64_t nseq = launch_groups<real_t, KK, R>(a);
    const int spb = nt / R;
    const dim3 grid((unsigned)((nseq + spb - 1) / spb)), block(nt);
    if constexpr (has_dense<real_t, KK, R>() && NRM == 4) {
        if (a.mask_runs) {
            hipLaunchKernelGGL((bscan_kernel_mr<real_t, KK, R, NRM>), grid, block, lds_bytes(R, nt, true), st, a, seg_sites, bseg, fseg);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL((bscan_kernel<real_t, KK, R, NRM>), grid, block, lds_bytes(R, nt), st, a, seg_sites, bseg, fseg);
    return hipGetLastError();
}

#endif
// T = 16 keeps 17 alpha vectors in registers: only offered where a lane owns <= 4 states.  float64 with 16
// states per lane beyond K = 16: not compiled (more than 256 registers plus scratch: see phk_api.hip, valid_Rf)
#ifdef PHK_EXP_F64_SPL16  // diagnostic builds only (scripts/diag_fenced_variants.py): compile every variant
constexpr bool f64_fwd_ok(int) { return true; }
#else
constexpr bool f64_fwd_ok(int R) { return sizeof(real_t) == 4 || KK / R <= 8 || KK == 16; }
#endif
template <int R, int T>
constexpr bool variant_ok() { return KK % R == 0 && KK / R <= 16 && R <= KK && (T == 8 || KK / R <= 4) && f64_fwd_ok(R); }

#if PHK_FWD_PART || PHK_LAT_PART
template <int R, int T>
static hipError_t fwd_rt(int nrm, bool ckpt, const KArgs& a, int nt, hipStream_t st) {
    if constexpr (!variant_ok<R, T>()) {
        return hipErrorInvalidValue;
    } else {
        if (nrm == 1) return fwd_rtn<R, T, 1>(ckpt, a, nt, st);
        if (nrm == 2) return fwd_rtn<R, T, 2>(ckpt, a, nt, st);
        if (nrm == 4) return fwd_rtn<R, T, 4>(ckpt, a, nt, st);
        return hipErrorInvalidValue;
    }
}
#endif
// float64 with more than 4 states per lane: not compiled (256 VGPRs + AGPR copies + scratch; see phk_api.hip, valid_Rb)
template <int R, int T>
constexpr bool bwd_variant_ok() { return variant_ok<R, T>() && (f64_sweep_ok<R, false>() || f64_sweep_ok<R, true>()); }

#if PHK_BWD_PART
template <int R, int T>
static hipError_t bwd_rt(int nrm, const KArgs& a, int units, int nt, hipStream_t st) {
    if constexpr (!bwd_variant_ok<R, T>()) {
        return hipErrorInvalidValue;
    } else {
        if (nrm == 1) return bwd_rtn<R, T, 1>(a, units, nt, st);
        if (nrm == 2) return bwd_rtn<R, T, 2>(a, units, nt, st);
        if (nrm == 4) return bwd_rtn<R, T, 4>(a, units, nt, st);
        return hipErrorInvalidValue;
    }
}
#endif
#if PHK_BWD_PART || PHK_LAT_PART
template <int R>
static hipError_t bscan_r(int nrm, const KArgs& a, int64_t seg_sites, void* bseg, int32_t* fseg, int nt, hipStream_t st) {
    if constexpr (!variant_ok<R, 8>()) {
        return hipErrorInvalidValue;
    } else {
        if (nrm == 1) return bscan_rn<R, 1>(a, seg_sites, bseg, fseg, nt, st);
        if (nrm == 2) return bscan_rn<R, 2>(a, seg_sites, bseg, fseg, nt, st);
        if (nrm == 4) return bscan_rn<R, 4>(a, seg_sites, bseg, fseg, nt, st);
        return hipErrorInvalidValue;
    }
}

#endif
#if PHK_FWD_PART || PHK_LAT_PART
template <int R>
static hipError_t fwd_r(int T, int nrm, bool ckpt, const KArgs& a, int nt, hipStream_t st) {
    if (T == 8) return fwd_rt<R, 8>(nrm, ckpt, a, nt, st);
    if (T == 16) return fwd_rt<R, 16>(nrm, ckpt, a, nt, st);
    return hipErrorInvalidValue;
}
#endif
#if PHK_BWD_PART
template <int R>
static hipError_t bwd_r(int T, int nrm, const KArgs& a, int units, int nt, hipStream_t st) {
    if (T == 8) return bwd_rt<R, 8>(nrm, a, units, nt, st);
    if (T == 16) return bwd_rt<R, 16>(nrm, a, units, nt, st);
    return hipErrorInvalidValue;
}

#endif
#if PHK_LAT_SPLIT || PHK_LAT_PART
hipError_t PHK_CAT(launch_fwd_lat_, PHK_SUFFIX)(int T, int nrm, bool ckpt, const KArgs& a, int nt, hipStream_t st);
hipError_t PHK_CAT(launch_bscan_lat_, PHK_SUFFIX)(int nrm, const KArgs& a, int64_t seg_sites, void* bseg, int32_t* fseg, int nt,
                                                  hipStream_t st);
#endif
#if PHK_LAT_PART
hipError_t PHK_CAT(launch_fwd_lat_, PHK_SUFFIX)(int T, int nrm, bool ckpt, const KArgs& a, int nt, hipStream_t st) {
    return fwd_r<16>(T, nrm, ckpt, a, nt, st);
}
hipError_t PHK_CAT(launch_bscan_lat_, PHK_SUFFIX)(int nrm, const KArgs& a, int64_t seg_sites, void* bseg, int32_t* fseg, int nt,
                                                  hipStream_t st) {
    return bscan_r<16>(nrm, a, seg_sites, bseg, fseg, nt, st);
}
#endif
#if PHK_FWD_PART
hipError_t PHK_CAT(launch_fwd_, PHK_SUFFIX)(int R, int T, int nrm, bool ckpt, const KArgs& a, int nt, hipStream_t st) {
    switch (R) {
        case 1: return fwd_r<1>(T, nrm, ckpt, a, nt, st);
        case 2: return fwd_r<2>(T, nrm, ckpt, a, nt, st);
        case 4: return fwd_r<4>(T, nrm, ckpt, a, nt, st);
        case 8: return fwd_r<8>(T, nrm, ckpt, a, nt, st);
#if PHK_LAT_SPLIT
        case 16: return PHK_CAT(launch_fwd_lat_, PHK_SUFFIX)(T, nrm, ckpt, a, nt, st);
#else
        case 16: return fwd_r<16>(T, nrm, ckpt, a, nt, st);
#endif
    }
    return hipErrorInvalidValue;
}
#endif
#if PHK_BWD_PART
hipError_t PHK_CAT(launch_bwd_, PHK_SUFFIX)(int R, int T, int nrm, const KArgs& a, int units, int nt, hipStream_t st) {
    switch (R) {
        case 1: return bwd_r<1>(T, nrm, a, units, nt, st);
        case 2: return bwd_r<2>(T, nrm, a, units, nt, st);
        case 4: return bwd_r<4>(T, nrm, a, units, nt, st);
        case 8: return bwd_r<8>(T, nrm, a, units, nt, st);
        case 16: return bwd_r<16>(T, nrm, a, units, nt, st);
    }
    return hipErrorInvalidValue;
}
#endif
#if PHK_BWD_PART
hipError_t PHK_CAT(launch_bscan_, PHK_SUFFIX)(int R, int nrm, const KArgs& a, int64_t seg_sites, void* bseg, int32_t* fseg,
                                              int nt, hipStream_t st) {
    switch (R) {
        case 1: return bscan_r<1>(nrm, a, seg_sites, bseg, fseg, nt, st);
        case 2: return bscan_r<2>(nrm, a, seg_sites, bseg, fseg, nt, st);
        case 4: return bscan_r<4>(nrm, a, seg_sites, bseg, fseg, nt, st);
        case 8: return bscan_r<8>(nrm, a, seg_sites, bseg, fseg, nt, st);
#if PHK_LAT_SPLIT
        case 16: return PHK_CAT(launch_bscan_lat_, PHK_SUFFIX)(nrm, a, seg_sites, bseg, fseg, nt, st);
#else
        case 16: return bscan_r<16>(nrm, a, seg_sites, bseg, fseg, nt, st);
#endif
    }
    return hipErrorInvalidValue;
}
#endif
#if PHK_BWD_PART
hipError_t PHK_CAT(launch_finalize_, PHK_SUFFIX)(const KArgs& a, int units, hipStream_t st) {
    const int64_t n = ((a.seq_end > 0 ? a.seq_end : a.B * a.S) - a.seq_begin) * 7 * KK;
    static_assert(KK % 4 == 0, "grad_finalize_kernel takes four states per thread");
    hipLaunchKernelGGL((grad_finalize_kernel<real_t>), dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, a, KK, units);
    {  // (sequences whose waves swept in the folded form: plain sums -> the caller's gradient)
        const int64_t m = n / 7;
        hipLaunchKernelGGL((grad_unfold_kernel<real_t>), dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, a, KK);
    }
    return hipGetLastError();
}

#endif
}  // namespace phk
