// rng.hip.h -- counter-based RNG + the validation evaluator's hash (device + host).
// Spec (shared with the CPU oracle, DESIGN.md "RNG streams"):
//   Philox4x32-10, key = engine seed, counter = (game_id, index, purpose tag, sub-index)
//   move sampling : (game_id, ply, 'MOVE', 0)          -> 53-bit uniform from words 0,1
//   prior noise   : (game_id, node, 'NOIS', a*64+trial) -> two 24-bit uniforms
//   rollouts      : (game_id, sim,  'ROLL', step)       -> word0 * n >> 32
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define BB_TAG_MOVE 0x4D4F5645u
#define BB_TAG_NOISE 0x4E4F4953u
#define BB_TAG_ROLL 0x524F4C4Cu

struct Philox4 {
    uint32_t x[4];
};

__host__ __device__ __forceinline__ Philox4 philox4x32_10(uint64_t key, uint32_t c0, uint32_t c1, uint32_t c2,
                                                         uint32_t c3) {
    uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    Philox4 o;
    o.x[0] = c0; o.x[1] = c1; o.x[2] = c2; o.x[3] = c3;
    return o;
}

__host__ __device__ __forceinline__ double bb_u53(uint64_t key, uint32_t game_id, uint32_t ply) {
    Philox4 r = philox4x32_10(key, game_id, ply, BB_TAG_MOVE, 0);
    return ((double)(r.x[0] >> 5) * 67108864.0 + (double)(r.x[1] >> 6)) / 9007199254740992.0;
}

__host__ __device__ __forceinline__ uint64_t bb_splitmix(uint64_t z) {
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

// FNV-1a over the AsInputArray bytes, finalised with splitmix64
struct HashAcc {
    uint64_t h;
    __host__ __device__ __forceinline__ explicit HashAcc(uint64_t salt) : h(0xcbf29ce484222325ull ^ salt) {}
    __host__ __device__ __forceinline__ void byte(int8_t b) {
        h ^= (uint8_t)b;
        h *= 0x100000001b3ull;
    }
    __host__ __device__ __forceinline__ uint64_t final() const { return bb_splitmix(h); }
};
__host__ __device__ __forceinline__ float bb_hash_value(uint64_t z) {
    return (float)(int32_t)(z >> 40) * (1.0f / 8388608.0f) - 1.0f;
}
__host__ __device__ __forceinline__ float bb_hash_policy(uint64_t z, int a) {
    uint64_t za = bb_splitmix(z + (uint64_t)(a + 1) * 0x9e3779b97f4a7c15ull);
    return (float)(int32_t)(1 + (za >> 44));
}

// Beta(alpha, 1-alpha) by Johnk's method (NetworkFactory.py:176-180: Dirichlet([a,1-a]) first coordinate):
// X = U^(1/alpha), Y = V^(1/(1-alpha)); accept when X+Y <= 1; return X/(X+Y).
// Trial t takes its two uniforms from Philox counter (game, node, 'NOIS', action*64 + t/2), words 2(t&1), 2(t&1)+1.
// bb_beta_pair evaluates trials 2k and 2k+1 (one Philox call) and returns the first accepted draw, or -1.
__device__ __forceinline__ float bb_beta_pair(uint64_t key, uint32_t game_id, uint32_t node, uint32_t action, float ia,
                                              float ib, uint32_t k) {
    Philox4 r = philox4x32_10(key, game_id, node, BB_TAG_NOISE, action * 64u + k);
    float out = -1.0f;
#pragma unroll
    for (int h = 1; h >= 0; h--) {
        float u = ((float)(r.x[2 * h] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        float v = ((float)(r.x[2 * h + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        // u^(1/a) = exp2(log2(u) / a) on the hardware transcendentals (v_log_f32 / v_exp_f32, ~1e-6 relative error:
        // far below what a prior-noise draw needs).  NOT __powf: that is the full-precision library pow, several
        // hundred instructions per call -- four of them per trial pair were most of the noise cost
        float X = __builtin_amdgcn_exp2f(ia * __builtin_amdgcn_logf(u)), Y = __builtin_amdgcn_exp2f(ib * __builtin_amdgcn_logf(v));
        if (X + Y <= 1.0f && X + Y > 0.0f) out = X / (X + Y); // h = 0 (the earlier trial) is written last
    }
    return out;
}

__device__ __forceinline__ float bb_beta_noise(uint64_t key, uint32_t game_id, uint32_t node, uint32_t action,
                                               float alpha) {
    float ia = 1.0f / alpha, ib = 1.0f / (1.0f - alpha);
    for (uint32_t k = 0; k < 32; k++) {
        float r = bb_beta_pair(key, game_id, node, action, ia, ib, k);
        if (r >= 0.0f) return r;
    }
    return alpha;
}
