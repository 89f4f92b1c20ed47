// Does a wave's VALU work slow down when another wave of the same SIMD issues MFMAs back to back?  (gfx950)
// One workgroup of 8 waves per CU: waves w and w + 4 share a SIMD.  Waves 0..3 run the probe chain (f32 / f64 / int /
// lds), waves 4..7 either idle or loop v_mfma_f32_16x16x4_f32 on four independent accumulators.
// usage: coissue            prints cycles per dependent instruction for each probe, alone and beside the MFMA wave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ void __launch_bounds__(512) k_probe(int iters, int mfma_on, int prio, long long *out, float *sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __shared__ float lds[64 * 8];
    __shared__ int done;
    if (threadIdx.x == 0) done = 0;
    lds[threadIdx.x] = (float)lane;
    __syncthreads();
    if (wave < 4) {
        if (prio) __builtin_amdgcn_s_setprio(3);
        long long t0 = clock64();
        float x = 1.0f + lane * 1e-3f;
        double y = 1.0 + lane * 1e-3;
        int z = lane + 1;
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                if (KIND == 0) x = __builtin_fmaf(x, 0.999f, 0.001f);
                if (KIND == 1) y = __builtin_fma(y, 0.999, 0.001);
                if (KIND == 2) z = z * 3 + 1;
                if (KIND == 3) x = lds[((__float_as_int(x) >> 3) & 63) + wave * 64] + 1.0f;
                if (KIND == 4) x = __shfl_xor(x, 1, 64) + 1.0f; // dpp / permute
            }
        }
        long long t1 = clock64();
        if (lane == 0) {
            out[blockIdx.x * 4 + wave] = t1 - t0;
            atomicAdd(&done, 1);
        }
        sink[blockIdx.x * 512 + threadIdx.x] = x + (float)y + (float)z;
    } else {
        f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        float av = 1.0f + lane, bv = 0.5f;
        if (mfma_on == 1) {
            while (*(volatile int *)&done < 4) {
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, a2, 0, 0, 0);
                    a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, a3, 0, 0, 0);
                }
            }
        } else if (mfma_on == 2) {
            bf16x8 ab, bb;
            for (int i = 0; i < 8; i++) { ab[i] = (__bf16)(1.0f + lane); bb[i] = (__bf16)0.5f; }
            while (*(volatile int *)&done < 4) {
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, a2, 0, 0, 0);
                    a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, a3, 0, 0, 0);
                }
            }
        }
        sink[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
    }
}

template <int KIND>
static void run(const char *name, int prio, long long *out, float *sink) {
    const int blocks = 256, iters = 2000;
    double res[3];
    for (int on = 0; on < 3; on++) {
        for (int rep = 0; rep < 2; rep++) {
            k_probe<KIND><<<blocks, 512>>>(iters, on, prio, out, sink);
            CHK(hipDeviceSynchronize());
        }
        long long h[256 * 4];
        CHK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
        double s = 0;
        for (int i = 0; i < blocks * 4; i++) s += (double)h[i];
        res[on] = s / (blocks * 4) / ((double)iters * 16);
    }
    printf("%-22s prio %d: alone %6.1f cycles per dependent instruction, beside an f32 16x16x4 MFMA wave %6.1f (x%.2f), beside a bf16 16x16x32 MFMA wave %6.1f (x%.2f)\n", name, prio ? 3 : 0, res[0], res[1], res[1] / res[0], res[2], res[2] / res[0]);
}

int main() {
    long long *out;
    float *sink;
    CHK(hipMalloc(&out, 256 * 4 * 8));
    CHK(hipMalloc(&sink, 256 * 512 * 4));
    for (int prio = 0; prio < 1; prio++) {
        run<0>("v_fma_f32 chain", prio, out, sink);
        run<1>("v_fma_f64 chain", prio, out, sink);
        run<3>("ds_read chain", prio, out, sink);
        run<4>("dpp/swizzle chain", prio, out, sink);
    }
    return 0;
}
