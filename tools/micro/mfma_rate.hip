// Issue rate of the bf16 MFMAs used by net_x3.hip.h on one SIMD (one wave per CU): cycles per instruction, 4 accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ void __launch_bounds__(64) k_rate(long long *out, float *sink) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    bf16x8 x, y;
    s16x4 u = {1, 2, 3, 4}, v = {5, 6, 7, 8};
    for (int i = 0; i < 8; i++) { x[i] = (__bf16)(1.0f + threadIdx.x); y[i] = (__bf16)0.5f; }
    long long t0 = clock64();
    for (int i = 0; i < 1000; i++) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (KIND == 0) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a3, 0, 0, 0);
            } else {
                a0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(u, v, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(u, v, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(u, v, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(u, v, a3, 0, 0, 0);
            }
        }
    }
    long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 64 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}
int main() {
    long long *out; float *sink;
    CHK(hipMalloc(&out, 8 * 256)); CHK(hipMalloc(&sink, 4 * 64 * 256));
    for (int kind = 0; kind < 2; kind++) {
        if (kind == 0) k_rate<0><<<256, 64>>>(out, sink); else k_rate<1><<<256, 64>>>(out, sink);
        CHK(hipDeviceSynchronize());
        long long h[256]; CHK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
        double s = 0; for (int i = 0; i < 256; i++) s += (double)h[i];
        printf("%s: %.1f cycles per instruction\n", kind == 0 ? "v_mfma_f32_16x16x32_bf16" : "v_mfma_f32_16x16x16_bf16", s / 256 / 16000);
    }
    return 0;
}
