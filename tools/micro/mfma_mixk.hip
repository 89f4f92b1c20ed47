// Minimal repro of the hazard met while writing net_x3.hip.h (ROCm 7.2, gfx950): a K = 32 bf16 MFMA directly followed by a
// K = 16 bf16 MFMA on the SAME accumulator.  If the second reads its SrcC before the first has written it, the first
// product is lost.  The kernel computes  acc = A32.B32 (K = 32, all ones: 32 per element)  then  acc += A16.B16 (K = 16, all
// ones: 16) and prints what came out (48 = correct, 16 = the K = 32 product was lost) for the two instructions back to back
// and with an s_nop 15 between them; the host also prints the instructions between the two MFMAs as compiled
// (llvm-objdump of its own code object: `hipcc --save-temps` or `llvm-objdump -d` on the bundle shows whether the
// compiler's hazard recognizer put wait states there).  Run it once after a ROCm update; net_x3.hip.h / gnet_x3.hip.h use
// one MFMA kind throughout and poison the K = 16 builtin, so the product does not depend on the answer.
//   hipcc -O3 --offload-arch=gfx950 -o mfma_mixk mfma_mixk.hip && ./mfma_mixk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int GAP>
__global__ void __launch_bounds__(64) k_mix(float *out) {
    bf16x8 a32, b32;
    for (int i = 0; i < 8; i++) { a32[i] = (__bf16)1.0f; b32[i] = (__bf16)1.0f; }
    const short one = 0x3f80; // bf16 1.0
    s16x4 a16 = {one, one, one, one}, b16 = a16;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a32, b32, acc, 0, 0, 0);
    if (GAP) asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc));
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a16, b16, acc, 0, 0, 0);
    out[threadIdx.x] = acc[0];
}

int main() {
    float *d, h[64];
    CHK(hipMalloc(&d, sizeof h));
    for (int gap = 0; gap < 2; gap++) {
        if (gap) k_mix<1><<<1, 64>>>(d); else k_mix<0><<<1, 64>>>(d);
        CHK(hipDeviceSynchronize());
        CHK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
        bool same = true;
        for (int i = 1; i < 64; i++) same = same && h[i] == h[0];
        printf("K=32 then K=16 on one accumulator, %s: D[0][0] = %g on lane 0%s (48 = both products, 16 = the K = 32 product lost)\n",
               gap ? "two s_nop 15 between" : "back to back", h[0], same ? ", all lanes alike" : ", lanes differ");
    }
    return 0;
}
