// LDS operand-read patterns of net_x3.hip.h: cycles per ds_read_b128 (one wave per CU, 64 lanes) for the tile addressing
// lane (n = lane & 15, g = lane >> 4) -> slot(n) * SB + (g >> 1) * SB + (g & 1) * 16, Connect4 row tiles (14 pixels + 2
// repeats), against a lane-linear read.  usage: lds_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(64) k_lds(int sb, int mode, int width, long long *out, unsigned *sink) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    const int lane = threadIdx.x, n = lane & 15, g = lane >> 4;
    for (int i = lane; i < 65536 / 4; i += 64) ((unsigned *)lds)[i] = i;
    __syncthreads();
    int addr;
    if (mode == 0) addr = lane * width;                                     // lane-linear
    else {
        const int W = 7, t = mode - 1;                                      // tile t = board rows 2t, 2t + 1
        int q = t * 14 + (n < 14 ? n : 13), y = q / W, x = q % W;
        int slot = (y + 1) * (W + 1) + (x + 1) - 9;
        addr = width == 16 ? slot * sb + (g >> 1) * sb + (g & 1) * 16 : slot * sb + g * 8;
    }
    unsigned acc = 0;
    long long t0 = clock64();
    for (int i = 0; i < 4096; i++) {
        const int off = (i & 3) * 32 % (sb > 96 ? 96 : 64);
        if (width == 16) {
            u32x4 v = *(volatile u32x4 *)(lds + addr + off);
            acc += v[0] + v[3];
        } else {
            u32x2 v = *(volatile u32x2 *)(lds + addr + off);
            acc += v[0] + v[1];
        }
    }
    long long t1 = clock64();
    if (lane == 0) out[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 64 + lane] = acc;
}

int main() {
    long long *out;
    unsigned *sink;
    CHK(hipMalloc(&out, 8 * 256));
    CHK(hipMalloc(&sink, 4 * 64 * 256));
    auto run = [&](int sb, int mode, int width) {
        k_lds<<<256, 64>>>(sb, mode, width, out, sink);
        CHK(hipDeviceSynchronize());
        long long h[256];
        CHK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
        double s = 0;
        for (int i = 0; i < 256; i++) s += (double)h[i];
        return s / 256 / 4096;
    };
    printf("lane-linear b128: %.1f cycles per read; b64: %.1f\n", run(96, 0, 16), run(96, 0, 8));
    for (int sb : {96, 112, 128, 144, 160, 176, 208, 240}) {
        printf("slot stride %3d B: b128 tiles %.1f %.1f %.1f | b64 tiles %.1f %.1f %.1f\n", sb, run(sb, 1, 16), run(sb, 2, 16), run(sb, 3, 16),
               run(sb, 1, 8), run(sb, 2, 8), run(sb, 3, 8));
    }
    return 0;
}
