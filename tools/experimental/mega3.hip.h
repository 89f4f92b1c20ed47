// mega3.hip.h -- persistent self-play kernel whose network evaluations are computed by TEAMS of waves.
//
// k_selfplay_queue (mega2.hip.h) gives every leaf to one wave, and that wave needs ~30 us for it: the 3 MFMA
// tiles of a 7x6 board run one after the other on one SIMD's matrix pipe.  With 16 games per CU the self-play
// rate is set by the per-game round trip (tree descent -> evaluation -> apply), not by any pipe's throughput,
// so here the evaluation itself is spread over a TEAM of T = 4 waves, one on each SIMD of the CU:
//     a team takes 1..3 queued leaves at a time; wave tw owns the 16-pixel tiles tw, tw+4 of their 3 / 6 / 8 tiles;
//     activations live in the team's shared LDS region (one layout, sized for 3 positions); after every conv layer
//     the team synchronises through per-wave epoch words in LDS (no atomics, no workgroup barrier);
//     after the tower, wave 0 computes the value head, wave 1 the policy head + softmax, wave 2 draws the
//     Beta(alpha, 1-alpha) prior noise, and wave 1 mixes it in.
// Two teams (waves 0-3 and 4-7) and four tree waves (8-11) put exactly one wave of each kind on every SIMD, so
// the waves of a team progress at the same pace.  Every output pixel is still one wave's own MFMA chain in the
// oracle's K order, so results are bit-identical to net_body's (tools/queue_cmp.py, GPU tests with BB_MEGA_QUEUE=2).
#pragma once
#include "mega2.hip.h"

#ifndef T_TREE_PRIO
#define T_TREE_PRIO 3 // issue priorities of the two kinds of waves (tuning)
#endif
#ifndef T_NET_PRIO
#define T_NET_PRIO 3
#endif
#define T_PWT 3 // positions a team takes per pass (8 tiles = 2 per wave)

struct TeamCtl {
    int flag[4]; // flag[tw] = number of synchronisation points wave tw has passed
    int slot[4]; // engine slots of the positions being evaluated
    int li[4];   // their workgroup-local game indices
    int n;       // 1..3 positions, -1 nothing queued, -2 finished
    int pad[3];
};

// publish "wave tw reached its next synchronisation point" (all lanes store the same word: no divergence)
__device__ __forceinline__ void team_bump(TeamCtl *tc, int tw, int &epoch) {
    epoch++;
    __threadfence_block(); // this wave's LDS writes first
    *(volatile int *)&tc->flag[tw] = epoch;
}
// wait until wave k has reached synchronisation point `epoch`; false = aborted (wall-clock limit)
__device__ __forceinline__ bool team_wait(TeamCtl *tc, int k, int epoch, QueueCtl *qc, long long t_start, long long t_limit) {
    int spins = 0;
    for (;;) {
        int v = *(volatile int *)&tc->flag[k];
        if (__builtin_amdgcn_readfirstlane(v - epoch) >= 0) break;
        if ((++spins & 0x3ff) == 0) {
            int late = wall_clock64() - t_start > t_limit || lds_load(&qc->abort_flag);
            if (__builtin_amdgcn_readfirstlane(late)) {
                qc->abort_flag = 1;
                return false;
            }
        }
    }
    __threadfence_block();
    return true;
}
template <int T>
__device__ __forceinline__ bool team_sync(TeamCtl *tc, int tw, int &epoch, QueueCtl *qc, long long t_start, long long t_limit) {
    team_bump(tc, tw, epoch);
#pragma unroll
    for (int k = 0; k < T; k++)
        if (k != tw && !team_wait(tc, k, epoch, qc, t_start, t_limit)) return false;
    return true;
}

#ifdef BB_STAMPS
#define TSTAMP(i) do { long long _t = clock64(); if ((threadIdx.x & 63) == 0 && tst) tst[i] += _t - _ts; _ts = clock64(); } while (0)
#define TS_ARG , long long *tst = nullptr
#define TS_PASS , tst
#define TS_INIT long long _ts = clock64();
#else
#define TSTAMP(i) do {} while (0)
#define TS_ARG
#define TS_PASS
#define TS_INIT
#endif

// One evaluation of n <= PW positions by a team of T waves; `tl` is the team's LDS region
// (NetGeom<G,PW>::WAVE_FLOATS floats, zeroed once per launch; the layout does not depend on n, so the zero halo
// survives any mix of batch sizes).  NTW = tiles this wave computes (0: it only keeps the team's synchronisation
// points).  Returns false when the launch was aborted.
template <class G, int PW, int T, int NTW>
__device__ __forceinline__ bool net_team_tower(const NetDev &nd, int n, float *tl, TeamCtl *tc, int tw, int &epoch,
                                               QueueCtl *qc, long long t_start, long long t_limit TS_ARG) {
    TS_INIT
    using NG = NetGeom<G, PW>;
    constexpr int W = NG::W, CIN = NG::CIN, HW = NG::HW, SLOTS = NG::SLOTS, CP = NG::CP, STEPS0 = NG::STEPS0, ACT = NG::ACT,
                  PLANE = NG::PLANE;
    const int lane = threadIdx.x & 63;
    const int j = lane >> 4, nn = lane & 15;
    float *actA = tl;
    float *actB = actA + ACT;
    const float *inp = actB + ACT;
    int aoff[NTW > 0 ? NTW : 1], ioff[NTW > 0 ? NTW : 1];
    bool valid[NTW > 0 ? NTW : 1];
#pragma unroll
    for (int k = 0; k < NTW; k++) {
        int q = (tw + T * k) * 16 + nn;
        valid[k] = q < n * HW;
        int qq = q < PW * HW ? q : 0;
        int pp = qq / HW, cell = qq % HW, y = cell / W, x = cell % W;
        int slot = (y + 1) * (W + 1) + (x + 1);
        aoff[k] = j * PLANE + (pp * SLOTS + slot) * 4;
        ioff[k] = (pp * SLOTS + slot) * CP;
    }
    f32x4 acc[NTW > 0 ? NTW : 1];
    if constexpr (NTW > 0) { // ---- first conv (this wave's tiles only)
        float w0r[STEPS0];
#pragma unroll
        for (int s = 0; s < STEPS0; s++) w0r[s] = nd.w0[s * 64 + lane];
        const f32x4 bias0 = *(const f32x4 *)(nd.epi + 4 * j), scale0 = *(const f32x4 *)(nd.epi + 16 + 4 * j),
                    shift0 = *(const f32x4 *)(nd.epi + 32 + 4 * j);
#pragma unroll
        for (int k = 0; k < NTW; k++) acc[k] = bias0;
#pragma unroll
        for (int s = 0; s < STEPS0; s++) {
            int kk0 = 4 * s + j;
            int kk = kk0 < 9 * CIN ? kk0 : 9 * CIN - 1;
            int tap = kk / CIN, c = kk % CIN;
            int toff = ((tap / 3 - 1) * (W + 1) + (tap % 3 - 1)) * CP + c;
            float a = w0r[s];
#pragma unroll
            for (int k = 0; k < NTW; k++) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, inp[ioff[k] + toff], acc[k], 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < NTW; k++) {
            f32x4 y;
#pragma unroll
            for (int r = 0; r < 4; r++) y[r] = fmaxf(__builtin_fmaf(acc[k][r], scale0[r], shift0[r]), 0.f);
            if (valid[k]) *(f32x4 *)(actA + aoff[k]) = y;
        }
    }
    TSTAMP(1);
    if (!team_sync<T>(tc, tw, epoch, qc, t_start, t_limit)) return false;
    TSTAMP(2);
    // ---- residual tower: one team synchronisation per layer (the output buffer of layer l+1 is the input of
    // layer l, which every wave finished reading before it reached the previous synchronisation) -----------
    const int L = (nd.dbg & 2) ? 0 : 2 * nd.R;
    for (int l = 0; l < L; l++) {
        if constexpr (NTW > 0) {
            const float *in = (l & 1) ? actB : actA;
            float *out = (l & 1) ? actA : actB;
            const float *ep = nd.epi + (size_t)(1 + l) * 48;
            f32x4 bias = *(const f32x4 *)(ep + 4 * j), scale = *(const f32x4 *)(ep + 16 + 4 * j),
                  shift = *(const f32x4 *)(ep + 32 + 4 * j);
            f32x4 w[9];
#pragma unroll
            for (int tap = 0; tap < 9; tap++) w[tap] = nd.wt[((size_t)l * 9 + tap) * 64 + lane];
#pragma unroll
            for (int k = 0; k < NTW; k++) acc[k] = bias;
#pragma unroll
            for (int tap = 0; tap < 9; tap++) {
                const int toff = ((tap / 3 - 1) * (W + 1) + (tap % 3 - 1)) * 4;
                f32x4 b[NTW];
#pragma unroll
                for (int k = 0; k < NTW; k++) b[k] = *(const f32x4 *)(in + aoff[k] + toff);
#pragma unroll
                for (int r = 0; r < 4; r++)
#pragma unroll
                    for (int k = 0; k < NTW; k++)
                        acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[tap][r], b[k][r], acc[k], 0, 0, 0);
            }
            const bool skip = (l & 1) != 0;
#pragma unroll
            for (int k = 0; k < NTW; k++) {
                f32x4 y;
                f32x4 sk = {0.f, 0.f, 0.f, 0.f};
                if (skip) sk = *(const f32x4 *)(out + aoff[k]);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float v = __builtin_fmaf(acc[k][r], scale[r], shift[r]);
                    if (skip) v = v + sk[r];
                    y[r] = fmaxf(v, 0.f);
                }
                if (valid[k]) *(f32x4 *)(out + aoff[k]) = y;
            }
        }
        TSTAMP(3);
        if (!team_sync<T>(tc, tw, epoch, qc, t_start, t_limit)) return false;
        TSTAMP(2);
    }
    return true;
}

template <class G, int PW, int T>
__device__ __forceinline__ bool net_team_body(const NetDev &nd, int n, const int *slot_list, float *tl, TeamCtl *tc, int tw,
                                              int &epoch, const typename G::State *states, const uint32_t *game_id,
                                              const int32_t *serial, int noise, float *value_out, float *policy_out,
                                              int pstride, QueueCtl *qc, long long t_start, long long t_limit TS_ARG) {
    TS_INIT
    static_assert(T == 4 || T == 2, "head roles are written for teams of four or two waves");
    // head roles: teams of four = value / policy / noise / (idle); teams of two = value + noise / policy
    constexpr int ROLE_VALUE = 0, ROLE_POLICY = 1, ROLE_NOISE = T == 4 ? 2 : 0;
    using NG = NetGeom<G, PW>;
    constexpr int W = NG::W, CIN = NG::CIN, A = NG::A, HW = NG::HW, SLOTS = NG::SLOTS, CP = NG::CP, ACT = NG::ACT,
                  PLANE = NG::PLANE;
    const int lane = threadIdx.x & 63;
    float *actA = tl;
    float *actB = actA + ACT;
    float *inp = actB + ACT;

    // ---- prologue: every wave of the team stages the same boards and planes (identical stores, no sync needed)
    {
        const int my_pp = lane % PW;
        const typename G::State my_state = states[slot_list[my_pp < n ? my_pp : 0]];
        typename G::State *sst = (typename G::State *)(inp + PW * SLOTS * CP);
        if (lane < PW) sst[lane] = my_state;
        for (int q = lane; q < PW * HW; q += 64) {
            int pp = q / HW, cell = q % HW, y = cell / W, x = cell % W;
            if (pp >= n) continue;
            float *dst = inp + (pp * SLOTS + (y + 1) * (W + 1) + (x + 1)) * CP;
            int8_t v[CIN];
            G::encode_cell(sst[pp], y, x, v);
#pragma unroll
            for (int c = 0; c < CIN; c++) dst[c] = (float)v[c];
        }
    }
    TSTAMP(0);
    // ---- tower: tiles tw and tw+4 of the ceil(n*HW/16) tiles that hold real pixels
    const int ntiles = (n * HW + 15) / 16;
    constexpr int NTWMAX = (NG::NT + T - 1) / T;
    const int mine = (NTWMAX >= 3 && ntiles > tw + 2 * T) ? 3 : (ntiles > tw + T ? 2 : (ntiles > tw ? 1 : 0));
    bool ok;
    if constexpr (NTWMAX >= 3) {
        if (mine == 3) {
            ok = net_team_tower<G, PW, T, 3>(nd, n, tl, tc, tw, epoch, qc, t_start, t_limit TS_PASS);
            goto tower_done;
        }
    }
    if (mine == 2) ok = net_team_tower<G, PW, T, 2>(nd, n, tl, tc, tw, epoch, qc, t_start, t_limit TS_PASS);
    else if (mine == 1) ok = net_team_tower<G, PW, T, 1>(nd, n, tl, tc, tw, epoch, qc, t_start, t_limit TS_PASS);
    else ok = net_team_tower<G, PW, T, 0>(nd, n, tl, tc, tw, epoch, qc, t_start, t_limit TS_PASS);
tower_done:
    if (!ok) return false;
#ifdef BB_STAMPS
    _ts = clock64();
#endif
    // ---- heads: tower output is in actA; actB and inp are scratch.  One role per wave. ------------------------
    const float *hp = nd.head;
    const int D = nd.D;
    float *rv = actB;             // [PW*HW]    value-conv output           (wave 0)
    float *rp = actB + PW * HW;   // [PW*HW][2] policy-conv output          (wave 1)
    float *sd = rp + 2 * PW * HW; // [PW][D]                                (wave 0)
    float *lg = sd + PW * D;      // [PW][A]                                (wave 1)
    float *nz = lg + PW * A;      // [PW][A]    Beta draws                  (wave 2 writes, wave 1 reads)
    if (tw == ROLE_VALUE) {
        const float *vk = hp + nd.off_vk, *v3 = hp + nd.off_v3;
        for (int q = lane; q < n * HW; q += 64) {
            int pp = q / HW, cell = q % HW, y = cell / W, x = cell % W;
            const float *xp = actA + (pp * SLOTS + (y + 1) * (W + 1) + (x + 1)) * 4;
            float av = v3[0];
#pragma unroll
            for (int c4 = 0; c4 < 4; c4++) {
                f32x4 xv = *(const f32x4 *)(xp + c4 * PLANE);
#pragma unroll
                for (int r = 0; r < 4; r++) av = __builtin_fmaf(xv[r], vk[4 * c4 + r], av);
            }
            rv[q] = fmaxf(__builtin_fmaf(av, v3[1], v3[2]), 0.f);
        }
        const float *d1k = hp + nd.off_d1k, *d1b = hp + nd.off_d1b;
        for (int q = lane; q < n * D; q += 64) {
            int pp = q / D, dd = q % D;
            float s = 0.f, wk = d1k[dd], wb = d1b[dd];
#pragma unroll
            for (int p = 0; p < HW; p++) s += __builtin_fmaf(rv[pp * HW + p], wk, wb);
            sd[q] = fmaxf(s, 0.f);
        }
        if (lane < n) {
            const float *d2k = hp + nd.off_d2k, *d2b = hp + nd.off_d2b;
            float e = d2b[0];
            for (int dd = 0; dd < D; dd++) e = __builtin_fmaf(sd[lane * D + dd], d2k[dd], e);
            value_out[slot_list[lane]] = tanhf(e);
        }
        for (int i = lane; i < PW * HW; i += 64) rv[i] = 0.f; // the scratch overlays halo slots: restore the zeros
        for (int i = lane; i < PW * D; i += 64) sd[i] = 0.f;
    }
    if (tw == ROLE_POLICY) {
        const float *pk = hp + nd.off_pk, *p6 = hp + nd.off_p6;
        for (int q = lane; q < n * HW; q += 64) {
            int pp = q / HW, cell = q % HW, y = cell / W, x = cell % W;
            const float *xp = actA + (pp * SLOTS + (y + 1) * (W + 1) + (x + 1)) * 4;
            float a0 = p6[0], a1 = p6[1];
#pragma unroll
            for (int c4 = 0; c4 < 4; c4++) {
                f32x4 xv = *(const f32x4 *)(xp + c4 * PLANE);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    int c = 4 * c4 + r;
                    a0 = __builtin_fmaf(xv[r], pk[2 * c], a0);
                    a1 = __builtin_fmaf(xv[r], pk[2 * c + 1], a1);
                }
            }
            rp[2 * q] = fmaxf(__builtin_fmaf(a0, p6[2], p6[4]), 0.f);
            rp[2 * q + 1] = fmaxf(__builtin_fmaf(a1, p6[3], p6[5]), 0.f);
        }
        const float *pdk = hp + nd.off_pdk, *pdb = hp + nd.off_pdb;
        for (int q = lane; q < n * A; q += 64) {
            int pp = q / A, a = q % A;
            float s = 0.f, k0 = pdk[a], k1 = pdk[A + a], kb = pdb[a];
#pragma unroll
            for (int p = 0; p < HW; p++)
                s += __builtin_fmaf(rp[2 * (pp * HW + p) + 1], k1, __builtin_fmaf(rp[2 * (pp * HW + p)], k0, kb));
            lg[q] = s;
        }
    }
    if (tw == ROLE_NOISE) {
        if (noise) { // two lanes per (position, action): lane pair (2i, 2i+1) tries Philox pairs k and k+1 side by side
            const float ia = 1.0f / nd.alpha, ib = 1.0f / (1.0f - nd.alpha);
            for (int base = 0; base < n * A; base += 32) {
                int q = base + (lane >> 1), sub = lane & 1;
                bool live = q < n * A;
                int a = live ? q % A : 0;
                int pos = live ? slot_list[q / A] : slot_list[0];
                uint32_t gid = game_id[pos], ser = (uint32_t)serial[pos];
                float r = -1.0f;
                for (uint32_t k = 0; k < 32 && __any(live && r < 0.0f); k += 2) {
                    float mine_r = (live && r < 0.0f) ? bb_beta_pair(nd.seed, gid, ser, (uint32_t)a, ia, ib, k + sub) : -1.0f;
                    float other = __shfl_xor(mine_r, 1, 64);
                    float first = sub ? other : mine_r, second = sub ? mine_r : other; // pair k before pair k+1
                    if (r < 0.0f) r = first >= 0.0f ? first : second;
                }
                if (live && sub == 0) nz[q] = r >= 0.0f ? r : nd.alpha;
            }
        }
    }
    TSTAMP(4);
    // the policy wave needs the noise wave's draws; nobody else waits here
    team_bump(tc, tw, epoch);
    if (tw == ROLE_POLICY) {
        if (!team_wait(tc, ROLE_NOISE, epoch, qc, t_start, t_limit)) return false;
        if (lane < n) { // one lane finishes each position (sequential, oracle order)
            const int pp = lane, pos = slot_list[lane];
            float m = -INFINITY;
            for (int a = 0; a < A; a++) m = fmaxf(m, lg[pp * A + a]);
            float pr[A];
            float tot = 0.f;
#pragma unroll
            for (int a = 0; a < A; a++) {
                pr[a] = expf(lg[pp * A + a] - m);
                tot += pr[a];
            }
#pragma unroll
            for (int a = 0; a < A; a++) pr[a] = pr[a] / tot;
            if (noise) { // policy = (1-eps)*softmax + eps*Beta(alpha,1-alpha); policy /= sum(policy)
                float t2 = 0.f;
#pragma unroll
                for (int a = 0; a < A; a++) {
                    pr[a] = (1.0f - nd.eps) * pr[a] + nd.eps * nz[pp * A + a];
                    t2 += pr[a];
                }
#pragma unroll
                for (int a = 0; a < A; a++) pr[a] = pr[a] / t2;
            }
#pragma unroll
            for (int a = 0; a < A; a++) policy_out[(size_t)pos * pstride + a] = pr[a];
        }
        for (int i = lane; i < 2 * PW * HW; i += 64) rp[i] = 0.f;
        for (int i = lane; i < 2 * PW * A; i += 64) lg[i] = 0.f; // logits and the noise draws behind them
    }
    release_global_then_lds(); // value / policy stores have landed before the final synchronisation word
    team_bump(tc, tw, epoch);
    TSTAMP(5);
    return true;
}

template <class G, int NTEAMS, int T = 4, int PWT = T_PWT>
__global__ void __launch_bounds__(MEGA2_THREADS) k_selfplay_team(TreeDev dg, NetDev nd, int visits, int noise_on, int limit_s) {
    constexpr int S = G::S, GW = 16, NETW = NTEAMS * T, TREEW = 12 - NETW, GPT = (GW + TREEW - 1) / TREEW;
    static_assert(GPT * S <= 64, "a tree wave holds at most 64 / S games");
    using NG = NetGeom<G, PWT>;
    constexpr int RMAX = MEGA_RMAX, STEPS0 = NG::STEPS0;
    constexpr int WT_F = 2 * RMAX * 9 * 64 * 4, W0_F = STEPS0 * 64, EPI_F = (1 + 2 * RMAX) * 48, HEAD_F = MEGA_HEAD_FLOATS;
    __shared__ __attribute__((aligned(16))) float lds[NTEAMS * NG::WAVE_FLOATS];
    __shared__ __attribute__((aligned(16))) float wlds[WT_F + W0_F + EPI_F + HEAD_F];
    __shared__ QueueCtl qc;
    __shared__ GameShadow<G, GW> shadow;
    __shared__ TeamCtl tcs[NTEAMS];
    __shared__ int gstate[GW]; // 0 owned by its tree wave, 1 leaf queued / being evaluated, 2 result published
#ifdef BB_STAMPS
    __shared__ long long tstamps[12][8];
    if (threadIdx.x < 96) ((long long *)tstamps)[threadIdx.x] = 0;
#endif
    const int wave = threadIdx.x >> 6, l64 = threadIdx.x & 63;
    const int g0 = blockIdx.x * GW;
    const int n_mine = dg.n_slots - g0 < GW ? dg.n_slots - g0 : GW;
    shadow.load(dg, g0, n_mine, MEGA2_THREADS);
    const TreeDev d = shadow.rebased(dg, g0); // everything below works on the LDS copies
    if (threadIdx.x == 0) {
        qc.head = 0;
        qc.tail = 0;
        qc.tree_done = 0;
        qc.abort_flag = 0;
    }
    if (threadIdx.x < GW) gstate[threadIdx.x] = 0;
    if (threadIdx.x < MEGA2_QCAP) qc.q[threadIdx.x] = -1;
    for (int i = threadIdx.x; i < NTEAMS * (int)(sizeof(TeamCtl) / 4); i += MEGA2_THREADS) ((int *)tcs)[i] = 0;
    for (int i = threadIdx.x; i < NTEAMS * NG::WAVE_FLOATS; i += MEGA2_THREADS) lds[i] = 0.f;
    NetDev ndl = nd;
    {
        const float *gwt = (const float *)nd.wt;
        for (int i = threadIdx.x; i < 2 * nd.R * 9 * 64 * 4; i += MEGA2_THREADS) wlds[i] = gwt[i];
        for (int i = threadIdx.x; i < W0_F; i += MEGA2_THREADS) wlds[WT_F + i] = nd.w0[i];
        for (int i = threadIdx.x; i < (1 + 2 * nd.R) * 48; i += MEGA2_THREADS) wlds[WT_F + W0_F + i] = nd.epi[i];
        for (int i = threadIdx.x; i < nd.head_floats; i += MEGA2_THREADS) wlds[WT_F + W0_F + EPI_F + i] = nd.head[i];
        ndl.wt = (const f32x4 *)wlds;
        ndl.w0 = wlds + WT_F;
        ndl.epi = wlds + WT_F + W0_F;
        ndl.head = wlds + WT_F + W0_F + EPI_F;
    }
    __syncthreads();
    const long long t_start = wall_clock64();
    const long long t_limit = 100000000ll * limit_s; // wall clock runs at 100 MHz

    if (wave >= NETW) { // ---------------- tree waves ----------------
        const int tw = wave - NETW;
        __builtin_amdgcn_s_setprio(T_TREE_PRIO);
        const int li = (l64 / S) * TREEW + tw, lane = l64 % S; // games are dealt round-robin to the tree waves
        const bool mine = l64 < GPT * S && li < GW && g0 + li < d.n_slots;
        const int g = g0 + li;
        int left = mine ? visits : 0;
        for (;;) {
            int stt = mine ? lds_load(&gstate[li]) : 1;
            bool ready = mine && left > 0 && stt != 1;
            bool busy = mine && (left > 0 || stt == 1); // still owes visits, or a leaf of mine is in flight
            if (!__any(busy)) break;
            if (__any(ready)) {
                __threadfence_block(); // acquire: the network team's results for state 2
                bool posted = false;
                if (ready) {
                    posted = async_game<G>(d, g, lane);
                    left--;
                    if (d.game_lid[g] < 0) left = 0; // slot ran out of games
                }
                queue_push(&qc, &gstate[li < GW ? li : 0], ready && lane == 0, posted, li);
            } else {
                __builtin_amdgcn_s_sleep(4);
                int late = wall_clock64() - t_start > t_limit || lds_load(&qc.abort_flag);
                if (__builtin_amdgcn_readfirstlane(late)) {
                    qc.abort_flag = 1;
                    break;
                }
            }
        }
        if (l64 == 0) atomicAdd(&qc.tree_done, 1);
    } else { // ---------------- network teams ----------------
        __builtin_amdgcn_s_setprio(T_NET_PRIO);
        const int team = wave / T, tw = wave % T;
        TeamCtl *tc = &tcs[team];
        float *tl = lds + team * NG::WAVE_FLOATS;
        int epoch = 0;
#ifdef BB_STAMPS
        long long t_work = 0, t_all0 = clock64(), n_evals = 0, n_pass = 0;
#endif
        for (;;) {
            if (tw == 0) { // the team leader takes up to PWT queued leaves and tells the others
                int got[PWT];
                int cnt = 0, first = -1;
#pragma unroll
                for (int k = 0; k < PWT; k++) {
                    int a = (k == 0 || cnt == k) ? __builtin_amdgcn_readfirstlane(queue_pop(&qc, t_start, t_limit, TREEW)) : -1;
                    if (k == 0) first = a;
                    got[k] = a;
                    if (a >= 0) cnt++;
                }
#pragma unroll
                for (int k = 0; k < PWT; k++) {
                    *(volatile int *)&tc->li[k] = got[k];
                    *(volatile int *)&tc->slot[k] = g0 + (got[k] >= 0 ? got[k] : 0);
                }
                *(volatile int *)&tc->n = cnt > 0 ? cnt : first; // first is -1 (nothing queued) or -2 (finished)
            }
            if (!team_sync<T>(tc, tw, epoch, &qc, t_start, t_limit)) break;
            const int n = __builtin_amdgcn_readfirstlane(*(volatile int *)&tc->n);
            if (n == -2) break;
            if (n < 0) {
                __builtin_amdgcn_s_sleep(2);
                // the leader must not overwrite tc->n before the others have read it
                if (!team_sync<T>(tc, tw, epoch, &qc, t_start, t_limit)) break;
                continue;
            }
#ifdef BB_STAMPS
            long long ts = clock64();
            n_evals += n;
            n_pass++;
#endif
            bool ok = net_team_body<G, PWT, T>(ndl, n, tc->slot, tl, tc, tw, epoch, (const typename G::State *)d.leaf_state,
                                               d.leaf_game_id, d.leaf_serial, noise_on, d.eval_value, d.eval_policy, S, &qc,
                                               t_start, t_limit
#ifdef BB_STAMPS
                                               , tstamps[wave]
#endif
                                               );
            if (!ok) break;
            // everybody waits for everybody: the results are published, and the leader may reuse the control block
            bool fine = true;
#pragma unroll
            for (int k = 0; k < T; k++)
                if (k != tw && !team_wait(tc, k, epoch, &qc, t_start, t_limit)) fine = false;
            if (!fine) break;
            if (tw == 0)
                for (int k = 0; k < n; k++) *(volatile int *)&gstate[tc->li[k]] = 2;
#ifdef BB_STAMPS
            t_work += clock64() - ts;
#endif
        }
#ifdef BB_STAMPS
        if (l64 == 0) {
            for (int i = 0; i < 4; i++) atomicAdd(&g_net_stamps[i], (unsigned long long)tstamps[wave][i]);
            atomicAdd(&g_net_stamps[4 + (tw < 3 ? tw : 2)], (unsigned long long)tstamps[wave][4]); // head time per role
            atomicAdd(&g_net_stamps[7], (unsigned long long)tstamps[wave][5]);
        }
        if (tw == 0 && l64 == 0 && d.stamps) {
            atomicAdd(&d.stamps[0], (unsigned long long)t_work);
            atomicAdd(&d.stamps[1], (unsigned long long)(clock64() - t_all0));
            atomicAdd(&d.stamps[4], 1ull);
            atomicAdd(&d.stamps[15], (unsigned long long)n_evals);
            atomicAdd(&d.stamps[11], (unsigned long long)n_pass);
        }
#endif
    }
    __syncthreads();
    if (threadIdx.x == 0 && qc.abort_flag) d.ctr[(size_t)g0 * 8 + 6] += 1; // surfaces as bb_counters.overflow
    __syncthreads();
    shadow.store(dg, g0, n_mine, MEGA2_THREADS); // hand the per-game state back to HBM
}
