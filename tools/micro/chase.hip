// Dependent-load latency on this GPU: one wave per workgroup follows a random cycle of 256-byte rows spread over
// `span` bytes.  Prints cycles (s_memtime) per hop.  usage: chase <span MiB> <rows (power of 2)> <blocks> [hops] [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void k_init(unsigned *rows, size_t stride_u32, unsigned n_rows) {
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_rows) rows[(size_t)i * stride_u32] = (i * 1664525u + 1013904223u) & (n_rows - 1);
}

__global__ void k_chase(const unsigned *rows, size_t stride_u32, int hops, unsigned start_stride, unsigned n_rows, long long *out) {
    unsigned cur = (blockIdx.x * start_stride) % n_rows;
    int lane = threadIdx.x & 63;
    long long t0 = clock64();
    unsigned acc = 0;
    for (int i = 0; i < hops; i++) {
        const unsigned *r = rows + (size_t)cur * stride_u32;
        unsigned nxt = r[0];        // every lane reads the link (same address) ...
        acc += r[(8 + lane) & 63];       // ... and its own word of the row, as the tree's PUCT step does
        cur = nxt;
    }
    long long t1 = clock64();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = acc + cur; }
}

int main(int argc, char **argv) {
    size_t span_mib = argc > 1 ? atol(argv[1]) : 64;
    unsigned n_rows = argc > 2 ? atoi(argv[2]) : 65536;
    int blocks = argc > 3 ? atoi(argv[3]) : 256;
    int hops = argc > 4 ? atoi(argv[4]) : 2000;
    size_t span = span_mib << 20;
    size_t stride = (span / n_rows) / 256 * 256;
    if (stride < 256) stride = 256;
    // n_rows must be a power of two: the link of row i is the full-period LCG step (i * 1664525 + 1013904223) mod n_rows
    if (n_rows & (n_rows - 1)) { printf("rows must be a power of two\n"); return 1; }
    unsigned *d;
    CHK(hipMalloc(&d, stride * n_rows));
    CHK(hipMemset(d, 0, stride * n_rows));
    k_init<<<(n_rows + 255) / 256, 256>>>(d, stride / 4, n_rows);
    CHK(hipDeviceSynchronize());
    long long *out;
    CHK(hipMalloc(&out, blocks * 16));
    int reps = argc > 5 ? atoi(argv[5]) : 1; // 1: every row is touched for the first time since the init pass swept the span
    for (int rep = 0; rep < reps; rep++) {
        k_chase<<<blocks, 64>>>(d, stride / 4, hops, n_rows / blocks + 1, n_rows, out);
        CHK(hipDeviceSynchronize());
    }
    std::vector<long long> h(2 * blocks);
    CHK(hipMemcpy(h.data(), out, blocks * 16, hipMemcpyDeviceToHost));
    double s = 0;
    for (int b = 0; b < blocks; b++) s += (double)h[2 * b];
    printf("span %zu MiB, %u rows (stride %zu B), %d waves: %.0f cycles per hop\n", span_mib, n_rows, stride, blocks, s / blocks / hops);
    return 0;
}
