// Micro-benchmark (developer tool): does the placement of a VALU instruction's three source registers matter to a lone wave?
// Four independent chains of v_fma_f32 / v_pk_fma_f32 on hard-coded registers: sources spread over register numbers that differ
// mod 4, or all equal mod 4.      hipcc --offload-arch=gfx950 -O2 vgpr_banks.hip -o vgpr_banks && ./vgpr_banks
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define CLOB "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71"
#define INIT "v_mov_b32 v40, 1.0\n v_mov_b32 v41, 1.0\n v_mov_b32 v42, 1.0\n v_mov_b32 v43, 1.0\n v_mov_b32 v44, 1.0\n v_mov_b32 v45, 1.0\n v_mov_b32 v46, 1.0\n v_mov_b32 v47, 1.0\n" \
             "v_mov_b32 v48, 0.5\n v_mov_b32 v49, 0.5\n v_mov_b32 v50, 0.5\n v_mov_b32 v51, 0.5\n v_mov_b32 v52, 0.5\n v_mov_b32 v53, 0.5\n v_mov_b32 v54, 0.5\n v_mov_b32 v55, 0.5\n" \
             "v_mov_b32 v56, 0.5\n v_mov_b32 v57, 0.5\n v_mov_b32 v58, 0.5\n v_mov_b32 v59, 0.5\n v_mov_b32 v60, 0.5\n v_mov_b32 v61, 0.5\n v_mov_b32 v62, 0.5\n v_mov_b32 v63, 0.5\n" \
             "v_mov_b32 v64, 0.5\n v_mov_b32 v65, 0.5\n v_mov_b32 v66, 0.5\n v_mov_b32 v67, 0.5\n v_mov_b32 v68, 0.5\n v_mov_b32 v69, 0.5\n v_mov_b32 v70, 0.5\n v_mov_b32 v71, 0.5\n"
template <int KIND>
__global__ void bench(float* out, int iters) {
    asm volatile(INIT ::: CLOB);
    for (int i = 0; i < iters; ++i) {
        if constexpr (KIND == 0)  // accumulators v40..v43; the two other sources in registers that differ from it and from each other mod 4
            asm volatile(REP16("v_fma_f32 v40, v40, v49, v50\n v_fma_f32 v41, v41, v50, v51\n v_fma_f32 v42, v42, v51, v48\n v_fma_f32 v43, v43, v48, v49\n") ::: CLOB);
        else if constexpr (KIND == 1)  // all three sources equal mod 4
            asm volatile(REP16("v_fma_f32 v40, v40, v48, v52\n v_fma_f32 v41, v41, v49, v53\n v_fma_f32 v42, v42, v50, v54\n v_fma_f32 v43, v43, v51, v55\n") ::: CLOB);
        else if constexpr (KIND == 2)  // two of the three equal mod 4
            asm volatile(REP16("v_fma_f32 v40, v40, v48, v49\n v_fma_f32 v41, v41, v49, v50\n v_fma_f32 v42, v42, v50, v51\n v_fma_f32 v43, v43, v51, v48\n") ::: CLOB);
        else if constexpr (KIND == 3)  // packed: pairs v[40:41].. accumulators; sources pairs whose first registers differ mod 4 where pairs allow (0 / 2)
            asm volatile(REP16("v_pk_fma_f32 v[40:41], v[40:41], v[50:51], v[56:57]\n v_pk_fma_f32 v[42:43], v[42:43], v[48:49], v[58:59]\n v_pk_fma_f32 v[44:45], v[44:45], v[54:55], v[60:61]\n v_pk_fma_f32 v[46:47], v[46:47], v[52:53], v[62:63]\n") ::: CLOB);
        else if constexpr (KIND == 4)  // packed, every pair starting at the same register number mod 4
            asm volatile(REP16("v_pk_fma_f32 v[40:41], v[40:41], v[48:49], v[56:57]\n v_pk_fma_f32 v[44:45], v[44:45], v[52:53], v[60:61]\n v_pk_fma_f32 v[64:65], v[64:65], v[48:49], v[56:57]\n v_pk_fma_f32 v[68:69], v[68:69], v[52:53], v[60:61]\n") ::: CLOB);
        else if constexpr (KIND == 5)  // two sources only (v_mul_f32), different mod 4
            asm volatile(REP16("v_mul_f32 v40, v40, v49\n v_mul_f32 v41, v41, v50\n v_mul_f32 v42, v42, v51\n v_mul_f32 v43, v43, v48\n") ::: CLOB);
        else if constexpr (KIND == 6)  // two sources only, equal mod 4
            asm volatile(REP16("v_mul_f32 v40, v40, v48\n v_mul_f32 v41, v41, v49\n v_mul_f32 v42, v42, v50\n v_mul_f32 v43, v43, v51\n") ::: CLOB);
    }
    float r;
    asm volatile("v_add_f32 %0, v40, v41\n v_add_f32 %0, %0, v42\n v_add_f32 %0, %0, v43\n v_add_f32 %0, %0, v44\n v_add_f32 %0, %0, v46\n v_add_f32 %0, %0, v64\n v_add_f32 %0, %0, v68" : "=v"(r) :: CLOB);
    out[threadIdx.x + blockIdx.x * blockDim.x] = r;
}
template <int KIND>
static void run(const char* name, int w, float* out) {
    const int iters = 2000, nblk = 256 * w;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(bench<KIND>, dim3(nblk), dim3(256), 0, 0, out, 10);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(bench<KIND>, dim3(nblk), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-78s waves/SIMD %d: %6.3f ns / instr\n", name, w, ms * 1e6 / (64.0 * iters));
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 512 * sizeof(float));
    for (int w = 1; w <= 2; ++w) {
        run<0>("v_fma_f32, 4 chains, the three sources in registers that differ mod 4", w, out);
        run<2>("v_fma_f32, 4 chains, two of the three sources equal mod 4", w, out);
        run<1>("v_fma_f32, 4 chains, all three sources equal mod 4", w, out);
        run<5>("v_mul_f32, 4 chains, sources differ mod 4", w, out);
        run<6>("v_mul_f32, 4 chains, sources equal mod 4", w, out);
        run<3>("v_pk_fma_f32, 4 chains, source pairs start at 0 / 2 mod 4 alternately", w, out);
        run<4>("v_pk_fma_f32, 4 chains, every source pair starts at the same register mod 4", w, out);
    }
    return 0;
}
