// Dev microbenchmark: single-wave dependent-chain latencies on gfx950 (shader cycles via s_memtime,
// and the shader clock in MHz against the 100 MHz wall clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int CTRL>
__device__ __forceinline__ float dppf(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}
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
__device__ __forceinline__ float rescale(float x, int& E) {
    float c = x;
    c = c + dppf<0xB1>(c);   // quad_perm 1,0,3,2
    c = c + dppf<0x4E>(c);   // quad_perm 2,3,0,1
    c = c + dppf<0x141>(c);  // row_half_mirror
    c = c + dppf<0x140>(c);  // row_mirror
    const int ex = __builtin_amdgcn_frexp_expf(c);
    E += ex;
    return x * __builtin_ldexpf(1.0f, -ex);
}

// mode 0: dependent v_fma chain; 1: dependent dpp-add chain; 2: dense16 only; 3: dense16 + rescale;
// 4: dense16 + rescale + wave vote on a loaded word (as the kernel's fast path does)
__global__ void k(int mode, int iters, float* out, long long* cyc, long long* wall, const uint32_t* words) {
    float x = 0.5f + 0.001f * threadIdx.x;
    float d[16];
    for (int j = 0; j < 16; ++j) d[j] = (j == (int)(threadIdx.x & 15) ? 0.9f : 0.00625f);
    int E = 0;
    const long long w0 = wall_clock64();
    const long long t0 = __builtin_readcyclecounter();
    if (mode == 0) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) x = __builtin_fmaf(x, 0.999f, 1e-6f);
        }
    } else if (mode == 1) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) x = x * 0.5f + dppf<0x111>(x);  // row_shr:1
        }
    } else if (mode == 2) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) x = dense16(x, d);
        }
    } else if (mode == 3) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) x = rescale(dense16(x, d), E);
        }
    } else if (mode == 5) {  // a loop whose body is one dependent add: cost of the taken back-edge
#pragma nounroll
        for (int i = 0; i < iters * 4; ++i) x = x + 1e-6f;
    } else if (mode == 6) {  // ... with a skipped forward branch per iteration (two taken branches)
#pragma nounroll
        for (int i = 0; i < iters * 4; ++i) {
            if (__builtin_amdgcn_readfirstlane(i) & 0x40000000) x = x * 0.5f;
            x = x + 1e-6f;
        }
    } else if (mode == 7) {  // 16 independent v_fmac_f32_dpp row_newbcast (4 chains), no combine: the dense step's issue cost
        float c0 = 0, c1 = 0, c2 = 0, c3 = 0;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                asm volatile("v_fmac_f32_dpp %0, %4, %5 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                             "v_fmac_f32_dpp %1, %4, %6 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                             "v_fmac_f32_dpp %2, %4, %7 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                             "v_fmac_f32_dpp %3, %4, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(x), "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]));
            }
        }
        x = (c0 + c1) + (c2 + c3);
    } else if (mode == 8) {  // 16 independent plain v_fmac_f32 (no DPP)
        float c0 = 0, c1 = 0, c2 = 0, c3 = 0;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                asm volatile("v_fmac_f32 %0, %4, %5\n\tv_fmac_f32 %1, %4, %6\n\tv_fmac_f32 %2, %4, %7\n\tv_fmac_f32 %3, %4, %8"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(x), "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]));
            }
        }
        x = (c0 + c1) + (c2 + c3);
    } else if (mode == 9) {  // 16 SALU adds in a dependent chain
        int s = __builtin_amdgcn_readfirstlane(iters);
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) asm volatile("s_add_i32 %0, %0, 3" : "+s"(s));
        }
        E = s;
    } else if (mode == 10) {  // dense16 + one checkpoint-like store per step
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                out[64 * ((4 * i + u) & 1023) + threadIdx.x] = x;
                x = dense16(x, d);
            }
        }
    } else {
        for (int i = 0; i < iters; ++i) {
            const uint32_t w = words[i & 1023];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (__all(((w >> (8 * u)) & 0xffu) == 0u)) x = dense16(x, d);
                else x = x * 0.5f;
                x = rescale(x, E);
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    const long long w1 = wall_clock64();
    out[threadIdx.x] = x + E;
    if (threadIdx.x == 0) { *cyc = t1 - t0; *wall = w1 - w0; }
}

int main() {
    float* out; long long *cyc, *wall; uint32_t* words;
    hipMalloc(&out, 64 * 1024 * 4 + 256); hipMalloc(&cyc, 8); hipMalloc(&wall, 8); hipMalloc(&words, 4096);
    hipMemset(words, 0, 4096);
    const char* names[] = {"16 dependent v_fma_f32", "16 dependent (mul + dpp add)", "4 x dense16", "4 x (dense16 + rescale)", "word load + 4 x (vote + dense16 + rescale)",
                           "loop: 1 add + taken back-edge", "loop: 1 add + skipped forward branch + back-edge", "16 v_fmac_f32_dpp row_newbcast, 4 chains",
                           "16 v_fmac_f32, 4 chains", "16 dependent s_add_i32", "4 x (store + dense16)"};
    const int per[] = {16, 16, 4, 4, 4, 4, 4, 1, 1, 1, 4};
    const int order[] = {0, 1, 2, 3, 11, 5, 6, 7, 8, 9, 10};
    for (int mi = 0; mi < 11; ++mi) {
        const int mode = order[mi];
        for (int rep = 0; rep < 2; ++rep) {
            const int iters = 20000;
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, mode, iters, out, cyc, wall, words);
            hipDeviceSynchronize();
            long long c, w;
            hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(&w, wall, 8, hipMemcpyDeviceToHost);
            if (rep) printf("%-52s: %7.1f cycles per step (%lld cycles, %.1f us wall, shader clock %.0f MHz)\n", names[mode == 11 ? 4 : mode],
                            (double)c / iters / per[mode == 11 ? 4 : mode], c, w / 100.0, c / (w / 100.0));
        }
    }
    return 0;
}
