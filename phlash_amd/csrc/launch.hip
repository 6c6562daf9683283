// One translation unit per (real, K): compiled with -DPHK_REAL=float|double -DPHK_K=<K>
// -DPHK_SUFFIX=f32_16 ... so that the variants build in parallel.  Dispatches (R, T) to the
// template instantiations of psmc_kernels.hip.
#include "psmc_kernels.hip"

#ifndef PHK_REAL
#error "compile with -DPHK_REAL=float|double -DPHK_K=<K> -DPHK_SUFFIX=<tag>"
#endif

#define PHK_CAT2(a, b) a##b
#define PHK_CAT(a, b) PHK_CAT2(a, b)

namespace phk {

using real_t = PHK_REAL;
constexpr int KK = PHK_K;

template <int R, int T>
static hipError_t fwd_rt(bool ckpt, const KArgs& a, int nt, hipStream_t st) {
    const int64_t nseq = a.B * a.S;
    const int spb = nt / R;  // sequences per workgroup
    const dim3 grid((unsigned)((nseq + spb - 1) / spb)), block(nt);
    if (ckpt)
        hipLaunchKernelGGL((fwd_kernel<real_t, KK, R, T, true>), grid, block, 0, st, a);
    else
        hipLaunchKernelGGL((fwd_kernel<real_t, KK, R, T, false>), grid, block, 0, st, a);
    return hipGetLastError();
}

template <int R, int T>
static hipError_t bwd_rt(const KArgs& a, int nt, hipStream_t st) {
    const int64_t nseq = a.B * a.S;
    const int spb = nt / R;
    const dim3 grid((unsigned)((nseq + spb - 1) / spb)), block(nt);
    const size_t lds = (size_t)T * (KK / R + 1) * nt * sizeof(real_t);
    auto kern = bwd_kernel<real_t, KK, R, T>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, grid, block, lds, st, a);
    return hipGetLastError();
}

template <int R>
static hipError_t fwd_r(int T, bool ckpt, const KArgs& a, int nt, hipStream_t st) {
    if constexpr (KK % R != 0 || KK / R > 16 || R > KK) {
        return hipErrorInvalidValue;
    } else {
        if (T == 8) return fwd_rt<R, 8>(ckpt, a, nt, st);
        if (T == 16) return fwd_rt<R, 16>(ckpt, a, nt, st);
        return hipErrorInvalidValue;
    }
}
template <int R>
static hipError_t bwd_r(int T, const KArgs& a, int nt, hipStream_t st) {
    if constexpr (KK % R != 0 || KK / R > 16 || R > KK) {
        return hipErrorInvalidValue;
    } else {
        if (T == 8) return bwd_rt<R, 8>(a, nt, st);
        if (T == 16) return bwd_rt<R, 16>(a, nt, st);
        return hipErrorInvalidValue;
    }
}

hipError_t PHK_CAT(launch_fwd_, PHK_SUFFIX)(int R, int T, bool ckpt, const KArgs& a, int nt, hipStream_t st) {
    switch (R) {
        case 1: return fwd_r<1>(T, ckpt, a, nt, st);
        case 2: return fwd_r<2>(T, ckpt, a, nt, st);
        case 4: return fwd_r<4>(T, ckpt, a, nt, st);
        case 8: return fwd_r<8>(T, ckpt, a, nt, st);
        case 16: return fwd_r<16>(T, ckpt, a, nt, st);
    }
    return hipErrorInvalidValue;
}
hipError_t PHK_CAT(launch_bwd_, PHK_SUFFIX)(int R, int T, const KArgs& a, int nt, hipStream_t st) {
    switch (R) {
        case 1: return bwd_r<1>(T, a, nt, st);
        case 2: return bwd_r<2>(T, a, nt, st);
        case 4: return bwd_r<4>(T, a, nt, st);
        case 8: return bwd_r<8>(T, a, nt, st);
        case 16: return bwd_r<16>(T, a, nt, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace phk
