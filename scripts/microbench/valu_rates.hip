// Micro-benchmark (developer tool, not part of the product): issue cost of the VALU instruction kinds the forward kernel is
// made of, for ONE wave per SIMD (the forward kernel's regime) and for two.  Prints cycles per instruction (s_memtime ticks
// are scaled by a calibration against wall-clock-free instruction counts: only ratios matter).
//   hipcc --offload-arch=gfx950 -O2 valu_rates.hip -o valu_rates && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2 __attribute__((ext_vector_type(2)));

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

template <int KIND>
__global__ void bench(float* out, long long* cyc, int iters) {
    v2 a = {out[threadIdx.x], out[threadIdx.x + 1]}, b = {1.0001f, 0.9999f}, c = {1e-3f, 1e-3f};
    v2 x0 = a, x1 = a + c, x2 = a - c, x3 = a * b;
    float s0 = a.x, s1 = a.y, s2 = c.x, s3 = c.y;
    float x = a.x + 1.0f, m[16];
    for (int k = 0; k < 16; ++k) m[k] = 0.0625f + 1e-4f * (float)k * a.y;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if constexpr (KIND == 0) {  // dependent v_fma_f32
            asm volatile(REP64("v_fma_f32 %0, %0, %1, %2\n") : "+v"(s0) : "v"(b.x), "v"(c.x));
        } else if constexpr (KIND == 1) {  // 4 independent chains of v_fma_f32
            asm volatile(REP16("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n")
                         : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3) : "v"(b.x), "v"(c.x));
        } else if constexpr (KIND == 2) {  // dependent v_pk_fma_f32 with the s_nop the compiler inserts
            asm volatile(REP64("v_pk_fma_f32 %0, %0, %1, %2\n s_nop 0\n") : "+v"(x0) : "v"(b), "v"(c));
        } else if constexpr (KIND == 3) {  // 4 independent chains of v_pk_fma_f32
            asm volatile(REP16("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b), "v"(c));
        } else if constexpr (KIND == 4) {  // the same with swapped halves of the chained operand (op_sel)
            asm volatile(REP16("v_pk_fma_f32 %0, %0, %4, %5 op_sel:[1,0,0] op_sel_hi:[0,1,1]\n v_pk_fma_f32 %1, %1, %4, %5 op_sel:[1,0,0] op_sel_hi:[0,1,1]\n"
                               "v_pk_fma_f32 %2, %2, %4, %5 op_sel:[1,0,0] op_sel_hi:[0,1,1]\n v_pk_fma_f32 %3, %3, %4, %5 op_sel:[1,0,0] op_sel_hi:[0,1,1]\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b), "v"(c));
        } else if constexpr (KIND == 5) {  // 2 independent chains of v_pk_fma_f32 (one other instruction between dependent ones)
            asm volatile(REP16("v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3\n")
                         : "+v"(x0), "+v"(x1) : "v"(b), "v"(c));
        } else if constexpr (KIND == 6) {  // s_nop 0 alone
            asm volatile(REP64("s_nop 0\n"));
        } else if constexpr (KIND == 7) {  // pk chain alternating with an independent scalar fma (filler instead of the nop)
            asm volatile(REP16("v_pk_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %4, %5\n")
                         : "+v"(x0), "+v"(s0) : "v"(b), "v"(c), "v"(b.x), "v"(c.x));
        } else if constexpr (KIND == 8) {  // 2 independent chains of v_fma_f32
            asm volatile(REP16("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n")
                         : "+v"(s0), "+v"(s1) : "v"(b.x), "v"(c.x));
        } else if constexpr (KIND == 9) {  // 4 independent v_pk_mul_f32 / v_pk_add_f32 mixed
            asm volatile(REP16("v_pk_mul_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %5\n v_pk_mul_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %5\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b), "v"(c));
        } else if constexpr (KIND == 10) {  // dense 16 x 16 step of the one-state-per-lane kernels, 4 accumulator chain(s); 4 steps per iteration, each on the result of the one before
            for (int rep = 0; rep < 4; ++rep) {
                float c0, c1 = 0.f, c2 = 0.f, c3 = 0.f;
                asm volatile("s_nop 1\n"
                "v_mul_f32_dpp %0, %4, %5 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
                "v_mul_f32_dpp %1, %4, %6 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                "v_mul_f32_dpp %2, %4, %7 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
                "v_mul_f32_dpp %3, %4, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %0, %4, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %1, %4, %10 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %2, %4, %11 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %3, %4, %12 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %0, %4, %13 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %1, %4, %14 row_newbcast:9 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %2, %4, %15 row_newbcast:10 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %3, %4, %16 row_newbcast:11 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %0, %4, %17 row_newbcast:12 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %1, %4, %18 row_newbcast:13 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %2, %4, %19 row_newbcast:14 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %3, %4, %20 row_newbcast:15 row_mask:0xf bank_mask:0xf\n"
                    : "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(c3) : "v"(x), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[8]), "v"(m[9]), "v"(m[10]), "v"(m[11]), "v"(m[12]), "v"(m[13]), "v"(m[14]), "v"(m[15]));
                x = (c0 + c1) + (c2 + c3);
            }
        } else if constexpr (KIND == 11) {  // dense 16 x 16 step of the one-state-per-lane kernels, 1 accumulator chain(s); 4 steps per iteration, each on the result of the one before
            for (int rep = 0; rep < 4; ++rep) {
                float c0, c1 = 0.f, c2 = 0.f, c3 = 0.f;
                asm volatile("s_nop 1\n"
                "v_mul_f32_dpp %0, %4, %5 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %0, %4, %6 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %0, %4, %7 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %0, %4, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %0, %4, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %0, %4, %10 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %0, %4, %11 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %0, %4, %12 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %0, %4, %13 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %0, %4, %14 row_newbcast:9 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %0, %4, %15 row_newbcast:10 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %0, %4, %16 row_newbcast:11 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %0, %4, %17 row_newbcast:12 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %0, %4, %18 row_newbcast:13 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %0, %4, %19 row_newbcast:14 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %0, %4, %20 row_newbcast:15 row_mask:0xf bank_mask:0xf\n"
                    : "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(c3) : "v"(x), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[8]), "v"(m[9]), "v"(m[10]), "v"(m[11]), "v"(m[12]), "v"(m[13]), "v"(m[14]), "v"(m[15]));
                x = c0;
            }
        } else if constexpr (KIND == 12) {  // dense 16 x 16 step of the one-state-per-lane kernels, 2 accumulator chain(s); 4 steps per iteration, each on the result of the one before
            for (int rep = 0; rep < 4; ++rep) {
                float c0, c1 = 0.f, c2 = 0.f, c3 = 0.f;
                asm volatile("s_nop 1\n"
                "v_mul_f32_dpp %0, %4, %5 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
                "v_mul_f32_dpp %1, %4, %6 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %0, %4, %7 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %1, %4, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %0, %4, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %1, %4, %10 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %0, %4, %11 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %1, %4, %12 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %0, %4, %13 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %1, %4, %14 row_newbcast:9 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %0, %4, %15 row_newbcast:10 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %1, %4, %16 row_newbcast:11 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %0, %4, %17 row_newbcast:12 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %1, %4, %18 row_newbcast:13 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %0, %4, %19 row_newbcast:14 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f32_dpp %1, %4, %20 row_newbcast:15 row_mask:0xf bank_mask:0xf\n"
                    : "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(c3) : "v"(x), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[8]), "v"(m[9]), "v"(m[10]), "v"(m[11]), "v"(m[12]), "v"(m[13]), "v"(m[14]), "v"(m[15]));
                x = c0 + c1;
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    v2 r = x0 + x1 + x2 + x3;
    out[threadIdx.x + blockIdx.x * blockDim.x] = r.x + r.y + s0 + s1 + s2 + s3 + x;
    if (threadIdx.x % 64 == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND>
static void run(const char* name, int waves_per_simd, float* out, long long* cyc) {
    const int iters = 2000, nblk = 256 * waves_per_simd;  // one 256-thread workgroup per CU (and wave slot): one wave per SIMD each
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(bench<KIND>, dim3(nblk), dim3(256), 0, 0, out, cyc, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(bench<KIND>, dim3(nblk), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(nblk * 4);
    hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += (double)v; avg /= h.size();
    const double n = 64.0 * iters;  // instructions of the measured kind per wave (KIND 2: + as many s_nop; KIND 7: pairs counted once each)
    printf("%-58s waves/SIMD %d: %7.2f counter ticks / instr, %8.3f ns / instr (kernel %.3f ms)\n", name, waves_per_simd, avg / n, ms * 1e6 / n, ms);
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * 2 * sizeof(float)); hipMemset(out, 0, 256 * 512 * 2 * sizeof(float));
    hipMalloc(&cyc, 4096 * sizeof(long long));
    for (int w = 1; w <= 2; ++w) {
        run<0>("dependent v_fma_f32", w, out, cyc);
        run<8>("2 chains of v_fma_f32", w, out, cyc);
        run<1>("4 chains of v_fma_f32", w, out, cyc);
        run<2>("dependent v_pk_fma_f32 + s_nop 0 (per pair)", w, out, cyc);
        run<5>("2 chains of v_pk_fma_f32", w, out, cyc);
        run<3>("4 chains of v_pk_fma_f32", w, out, cyc);
        run<4>("4 chains of v_pk_fma_f32, swapped halves (op_sel)", w, out, cyc);
        run<9>("4 chains of v_pk_mul_f32 / v_pk_add_f32", w, out, cyc);
        run<7>("v_pk_fma_f32 chain + independent v_fma_f32 (per instr)", w, out, cyc);
        run<6>("s_nop 0", w, out, cyc);
        run<10>("dense16 step, 4 accumulators (per 16 steps of 64: x16)", w, out, cyc);
        run<11>("dense16 step, 1 accumulator", w, out, cyc);
        run<12>("dense16 step, 2 accumulators", w, out, cyc);
    }
    return 0;
}
