/* A plain-C client of libphlash_hip.so: no Python, no torch -- only the C ABI (include/phlash_hip.h)
 * and the HIP runtime for device memory.  It does what one call of the reference's kernel object does
 * (_PSMCKernelBase.__call__, src/phlash/gpu.py:182-325): observations and a [B, S, 7, K] parameter
 * block in, log-likelihoods and d ll / d log(theta) out.
 *
 *   c_abi_client <input.bin> <double_precision 0|1>
 *
 * input.bin (little endian): int64 K, N, L, B, S; int8 data[N*L]; int64 inds[S]; float64 params[B*S*7*K].
 * Output (text): one line "ll b s value", then "dlog b s row k value" lines.
 *
 * Build (plain C compiler; the HIP runtime is used for device memory only):
 *   gcc -O2 -std=c11 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/c_abi_client.c \
 *       -o c_abi_client -L phlash_amd -lphlash_hip -L /opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,$PWD/phlash_amd -Wl,-rpath,/opt/rocm/lib
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "phlash_hip.h"

#define HIPCHK(x)                                                            \
    do {                                                                     \
        hipError_t e_ = (x);                                                 \
        if (e_ != hipSuccess) {                                              \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));          \
            return 2;                                                        \
        }                                                                    \
    } while (0)
#define PHKCHK(x)                                                            \
    do {                                                                     \
        int rc_ = (x);                                                       \
        if (rc_ != PHK_OK) {                                                 \
            fprintf(stderr, "%s: code %d: %s\n", #x, rc_, phk_last_error()); \
            return 3;                                                        \
        }                                                                    \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s input.bin double_precision\n", argv[0]);
        return 1;
    }
    const int dbl = atoi(argv[2]);
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 1;
    int64_t hdr[5];
    if (fread(hdr, sizeof(int64_t), 5, f) != 5) return 1;
    const int64_t K = hdr[0], N = hdr[1], L = hdr[2], B = hdr[3], S = hdr[4];
    const size_t npar = (size_t)(B * S * 7 * K);
    int8_t* data = (int8_t*)malloc((size_t)(N * L));
    int64_t* inds = (int64_t*)malloc((size_t)S * sizeof(int64_t));
    double* params = (double*)malloc(npar * sizeof(double));
    if (fread(data, 1, (size_t)(N * L), f) != (size_t)(N * L) || fread(inds, sizeof(int64_t), (size_t)S, f) != (size_t)S ||
        fread(params, sizeof(double), npar, f) != npar)
        return 1;
    fclose(f);

    phk_handle* h = NULL;
    PHKCHK(phk_create(&h, (int)K, data, N, L, /*data_on_device=*/0, dbl, /*device=*/0));

    /* parameter block in the handle's element type, on the device */
    const size_t esz = dbl ? sizeof(double) : sizeof(float);
    void* hpar = malloc(npar * esz);
    for (size_t i = 0; i < npar; ++i) {
        if (dbl) ((double*)hpar)[i] = params[i];
        else ((float*)hpar)[i] = (float)params[i];
    }
    void *d_par = NULL, *d_grad = NULL;
    int64_t* d_inds = NULL;
    double* d_ll = NULL;
    HIPCHK(hipMalloc(&d_par, npar * esz));
    HIPCHK(hipMalloc(&d_grad, npar * esz));
    HIPCHK(hipMalloc((void**)&d_inds, (size_t)S * sizeof(int64_t)));
    HIPCHK(hipMalloc((void**)&d_ll, (size_t)(B * S) * sizeof(double)));
    HIPCHK(hipMemcpy(d_par, hpar, npar * esz, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_inds, inds, (size_t)S * sizeof(int64_t), hipMemcpyHostToDevice));

    /* one [7,K] block per (particle, chunk): strides S*7*K and 7*K; no warm-up; d/dlog output */
    PHKCHK(phk_loglik(h, d_par, S * 7 * K, 7 * K, d_inds, B, S, /*W=*/0, d_ll, d_grad, /*grad_dlog=*/1, /*stream=*/NULL));
    HIPCHK(hipDeviceSynchronize());

    double* ll = (double*)malloc((size_t)(B * S) * sizeof(double));
    void* grad = malloc(npar * esz);
    HIPCHK(hipMemcpy(ll, d_ll, (size_t)(B * S) * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(grad, d_grad, npar * esz, hipMemcpyDeviceToHost));
    for (int64_t b = 0; b < B; ++b)
        for (int64_t s = 0; s < S; ++s) printf("ll %lld %lld %.17g\n", (long long)b, (long long)s, ll[b * S + s]);
    for (int64_t b = 0; b < B; ++b)
        for (int64_t s = 0; s < S; ++s)
            for (int r = 0; r < 7; ++r)
                for (int64_t k = 0; k < K; ++k) {
                    const size_t i = (size_t)(((b * S + s) * 7 + r) * K + k);
                    printf("dlog %lld %lld %d %lld %.9g\n", (long long)b, (long long)s, r, (long long)k,
                           dbl ? ((double*)grad)[i] : (double)((float*)grad)[i]);
                }
    PHKCHK(phk_destroy(h));
    hipFree(d_par); hipFree(d_grad); hipFree(d_inds); hipFree(d_ll);
    free(data); free(inds); free(params); free(hpar); free(ll); free(grad);
    return 0;
}
