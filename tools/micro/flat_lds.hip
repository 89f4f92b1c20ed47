// micro-test: read-after-write through generic (flat) pointers into LDS at high offsets
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ unsigned long long launder(const void *p) {
    unsigned long long a = (unsigned long long)p;
    asm volatile("" : "+v"(a));
    return a;
}
__global__ void __launch_bounds__(768) k(int *out, int iters, int g0) {
    __shared__ float big[38000];           // 152 KB
    __shared__ int counters[16];
    if (threadIdx.x < 16) counters[threadIdx.x] = 1000000;
    for (int i = threadIdx.x; i < 38000; i += 768) big[i] = 0.f;
    __syncthreads();
    int *c = (int *)(launder(counters) - (unsigned long long)g0 * 4);
    int wave = threadIdx.x >> 6, l64 = threadIdx.x & 63;
    int grp = l64 >> 3, lane = l64 & 7;
    int g = g0 + wave;          // one counter per wave for waves 0..11; groups of the wave all use it? no: use (wave) only group 0
    int bad = 0;
    if (grp == 0 && wave < 12) {
        for (int it = 0; it < iters; it++) {
            if (lane == 0) c[g] -= 1;
            __threadfence_block();
            int v = c[g];
            if (v != 1000000 - (it + 1)) bad++;
            __threadfence_block();
        }
    }
    if (bad) atomicAdd(&out[0], bad);
    if (threadIdx.x == 0) out[1] = (int)big[5];
    __syncthreads();
    if (threadIdx.x < 12) out[2 + threadIdx.x] = counters[threadIdx.x];
}
int main() {
    int *d; hipMalloc(&d, 64 * 4); hipMemset(d, 0, 64 * 4);
    k<<<4, 768>>>(d, 10000, 4080);
    hipError_t e = hipDeviceSynchronize();
    int h[16]; hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    printf("err=%d bad=%d counters:", (int)e, h[0]);
    for (int i = 0; i < 12; i++) printf(" %d", h[2 + i]);
    printf("\n");
    return 0;
}
