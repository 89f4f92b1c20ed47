// Numerical behaviour of v_mfma_f32_16x16x32_bf16 / 16x16x16: does C + sum of small exact products keep float32 accuracy?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// A: [16][32] bf16 bits row-major, B: [32][16], C: [16][16]; assumed layout: lane (i = l & 15, g = l >> 4) holds k = 8g .. 8g + 7
__global__ void k32(const unsigned short *A, const unsigned short *B, const float *C, float *D, int alt) {
    int l = threadIdx.x, i = l & 15, g = l >> 4;
    unsigned short a[8], b[8];
    for (int e = 0; e < 8; e++) {
        int k = alt ? (e < 4 ? 4 * g + e : 16 + 4 * g + (e - 4)) : 8 * g + e;
        a[e] = A[i * 32 + k];
        b[e] = B[k * 16 + i];
    }
    bf16x8 av, bv;
    memcpy(&av, a, 16);
    memcpy(&bv, b, 16);
    f32x4 c;
    for (int r = 0; r < 4; r++) c[r] = C[(4 * g + r) * 16 + i];
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) D[(4 * g + r) * 16 + i] = c[r];
}
static unsigned short bf(float v) { unsigned u; memcpy(&u, &v, 4); u += 0x7FFF + ((u >> 16) & 1); return u >> 16; }
static float fb(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
int main() {
    unsigned short hA[512], hB[512];
    float hC[256], hD[256];
    srand(1);
    for (int scale_pow = 0; scale_pow <= 20; scale_pow += 10) {
        float sc = ldexpf(1.0f, -scale_pow);
        for (int i = 0; i < 512; i++) { hA[i] = bf(sc * ((rand() % 2001) / 1000.0f - 1.0f)); hB[i] = bf((float)(rand() % 3 - 1)); }
        for (int i = 0; i < 256; i++) hC[i] = (rand() % 2001) / 1000.0f - 1.0f;
        unsigned short *dA, *dB; float *dC, *dD;
        CHK(hipMalloc(&dA, 1024)); CHK(hipMalloc(&dB, 1024)); CHK(hipMalloc(&dC, 1024)); CHK(hipMalloc(&dD, 1024));
        CHK(hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice)); CHK(hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice)); CHK(hipMemcpy(dC, hC, 1024, hipMemcpyHostToDevice));
        for (int alt = 0; alt < 2; alt++) {
            k32<<<1, 64>>>(dA, dB, dC, dD, alt);
            CHK(hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost));
            double maxerr = 0;
            for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) {
                double s = hC[i * 16 + j];
                for (int k = 0; k < 32; k++) s += (double)fb(hA[i * 32 + k]) * (double)fb(hB[k * 16 + j]);
                maxerr = fmax(maxerr, fabs(s - hD[i * 16 + j]));
            }
            printf("A scale 2^-%d, layout %s: max |D - exact| = %.3e\n", scale_pow, alt ? "k = {4g..4g+3, 16+4g..}" : "k = 8g..8g+7", maxerr);
        }
    }
    return 0;
}
