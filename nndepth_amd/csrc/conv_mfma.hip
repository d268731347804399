// Implicit-GEMM 2-D convolution (stride 1, zero "same" padding) on the gfx950 fp32 MFMA.
//
//   out[b, co, y, x] = epilogue( bias[co] + sum_{ci,dy,dx} W[co,ci,dy,dx] * in[b, ci, y+dy-ph, x+dx-pw] )
//
// Mapping (D = A*B with v_mfma_f32_32x32x2_f32, an exact fp32 fmaf chain):
//   A = weights     A[i = co (32 per wave)][k]     streamed global -> VGPR, pre-packed on the host in
//                                                  fragment order (one coalesced 1 KiB dwordx4 load per wave per
//                                                  8 input channels of one tap); never touches LDS
//   B = activations B[k][j = pixel (32 per MFMA)]  staged once per workgroup as a zero-filled halo patch in
//                                                  LDS, shared by all waves and all taps
//   D[i = co][j = pixel]: a lane holds ONE pixel and 16 output channels, so NCHW stores are 32 consecutive
//   pixels per (register, half-wave) and the GRU gate math is a pure per-lane epilogue.
// The 32 pixels of an MFMA column block form a 4-row x 8-col sub-tile (tiles 68x120 exactly: 17 x 15); a
// wave owns P such sub-tiles side by side (P accumulators).
// Workgroup = wco x ks waves.  Wave (cbi, kj) owns output-channel block blockIdx.y*wco + cbi and K-slice kj:
// of every super-chunk of ks*CI_T input channels staged in LDS it multiplies channels [kj*CI_T, (kj+1)*CI_T);
// the ks partial accumulators are summed through LDS at the end (intra-workgroup split-K) — this is what lets
// 255 pixel tiles x ncb channel blocks load 1024 SIMDs evenly at batch 1.
// Pipeline: a chunk is walked in steps of (tap, 32-channel group) = 16*P MFMAs; the A fragments and the B
// operands of step s+1 are fetched (global / LDS) before the MFMAs of step s issue; the patch of the next
// super-chunk is fetched into registers at the top of a chunk and written to the other LDS buffer at its end
// (one barrier per super-chunk, i.e. per 16*P*taps*CI_T/32 MFMAs per wave).
//
// Replaces the nn.Conv2d calls of nndepth/blocks/update_block.py:57-65,26-36,97-112 and
// nndepth/blocks/gru.py:22-37,53-61 (reference files; semantics restated in oracle/torch_ref.py).
#include "common.h"
#include "conv_epilogue.h"
#include "layout.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace nnd {

#ifndef NND_INTERLEAVE
#define NND_INTERLEAVE 0
#endif

#ifndef NND_SC
#define NND_SC 8
#endif
// LDS row stride: smallest odd multiple of SC that holds PC columns (lanes (r, c) then hit 32 distinct banks)
__host__ __device__ constexpr int patch_stride(int PC, int SC) {
    int s = SC;
    while (s < PC) s += 2 * SC;
    return s;
}

#ifdef NND_DBG_STAMPS
// debug build only: per-workgroup phase timestamps (s_memrealtime, 100 MHz) for scripts/stamps.py
__device__ unsigned long long g_stamps[4096 * 8];
#define NND_STAMP(i)                                                                                   \
    do {                                                                                               \
        if (threadIdx.x == 0) {                                                                        \
            const unsigned lin_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);      \
            if (lin_ < 4096) g_stamps[lin_ * 8 + (i)] = __builtin_amdgcn_s_memrealtime();              \
        }                                                                                              \
    } while (0)
#else
#define NND_STAMP(i)
#endif

// NE: patch elements staged per thread per super-chunk (the thread owns one patch position and NE channels)
template <int KH, int KW, int CI_T, int P, int NE, int STR = 1>
__global__ void __launch_bounds__((P == 1 ? 768 : 512)) __attribute__((amdgpu_waves_per_eu((P == 1 && NE == 8 ? 4 : 1))))
conv_mfma_kernel(ConvArgs a) {
    constexpr int NT = KH * KW;
    constexpr int SG = CI_T < 32 ? CI_T : 32;  // channels per pipeline step
    constexpr int NGRP = CI_T / SG;            // channel groups per tap
    constexpr int NS = NT * NGRP;              // steps per chunk
    constexpr int AQ = SG / 8;                 // float4 A fragments per lane per step
    constexpr int NB = SG / 2 * P;             // MFMAs (= B operands) per step
    constexpr int PH = KH / 2, PW = KW / 2;
    extern __shared__ float lds[];
    NND_STAMP(0);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform -> scalar address math
    const int wco = a.wco, ks = a.ks;
    const int cbi = wave % wco, kj = wave / wco;
    const int h2 = lane >> 5, l31 = lane & 31;
    constexpr int SC = NND_SC, SR = 32 / NND_SC;  // MFMA column block = SR rows x SC cols of pixels
    const int r = l31 / SC, c = l31 % SC;
    const int tx0 = (blockIdx.x % a.tiles_x) * (P * SC);
    const int ty0 = (blockIdx.x / a.tiles_x) * SR;
    const int cb = blockIdx.y * wco + cbi;
    const bool active = cb * 32 < a.Cout;  // trailing waves of the last workgroup only help staging
    const int b = blockIdx.z;
    const int Hin = a.Hin, Win = a.Win;
    const long SP = a.ls.plane;  // channel stride of the source tensors
    // LDS patch geometry is compile-time so every B-operand read is base + immediate offset.
    // Row stride S is an odd multiple of SC: lanes (r, c) then hit 32 distinct banks.
    // Stride 2: the input patch is stored split into its 4 (row, col) parity phases, so that the operand of tap
    // (dy, dx) for output pixel (r, c) — input (2r+dy, 2c+dx) — is again lane_base + an immediate, with unit lane stride.
    constexpr int PR = (SR - 1) * STR + KH, PC = (P * SC - 1) * STR + KW;
    constexpr int PRH = (PR + STR - 1) / STR, PCH = (PC + STR - 1) / STR;  // rows / cols of one phase (== PR, PC for stride 1)
    constexpr int S = patch_stride(PCH, SC), PHASE = PRH * S, PATCH = STR * STR * PHASE;
    const int SCH = ks * CI_T;  // channels per super-chunk

    // ---- staging role of this thread: patch position `pos`, channels cg, cg+ngroups, ... (NE of them)
    const int npos = a.npos, ngroups = a.ngroups;
    const int pos = tid % npos, cg = tid / npos;
    const bool stager = cg < ngroups;
    const int pr = pos / PC, pc = pos - pr * PC;
    const int gy = ty0 * STR + pr - PH, gx = tx0 * STR + pc - PW;
    const bool inimg = stager && gy >= 0 && gy < Hin && gx >= 0 && gx < Win;
    const int goff0 = inimg ? (int)pix_off(a.ls, gy, gx) : 0;
    const int loff0 = ((pr % STR) * STR + (pc % STR)) * PHASE + (pr / STR) * S + pc / STR;
    const int trash = 2 * SCH * PATCH;  // one spare LDS word swallows the stores of non-staging threads

    f32x16 acc[P];
#pragma unroll
    for (int pp = 0; pp < P; ++pp)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[pp][i] = 0.f;

    const int nchunks = a.nchunks;
    const int nsuper = (nchunks + ks - 1) / ks;
    // uniform (SGPR) base + 32-bit lane offset: the loads use the saddr form, no per-load VGPR address math
    const float4* wbase = reinterpret_cast<const float4*>(a.wpk) + (size_t)(active ? cb : 0) * nchunks * (NS * AQ * 64);
    float stage[NE];

    auto chunk_src = [&](int K, const float*& src, int& climit) {
        int cbase = K * SCH;
        if (cbase < a.c0) {
            src = a.src0 + b * a.bs0 + (long)cbase * SP;
            climit = a.c0 - cbase;
        } else {
            int cc = cbase - a.c0;
            src = a.src1 + b * a.bs1 + (long)cc * SP;
            climit = a.c1 - cc;
        }
    };
    // raw loads only (clamped to element 0 when masked, so they are unconditional and hipcc counts vmcnt
    // exactly); the zero-fill select is applied in store_x, after the chunk's MFMAs
    // channel of staging slot j: strided over the thread groups for planar sources; NE consecutive channels per thread for
    // 4-channel-interleaved sources (layout.h), which are then NE/4 16-B loads
    const bool c4s = a.ls.ci == 4;
    auto slot_ch = [&](int j) { return c4s ? cg * NE + j : cg + j * ngroups; };
    auto load_x = [&](int K) {
        const float* src;
        int climit;
        chunk_src(K, src, climit);
        if (!inimg) climit = 0;
        if (c4s) {
#pragma unroll
            for (int q = 0; q < NE / 4; ++q) {
                const int ci = cg * NE + 4 * q;  // multiple of 4: channel group ci/4 starts at ci*SP, like a planar channel
                const float4 t = *reinterpret_cast<const float4*>(src + (ci < climit ? (unsigned)(ci * (int)SP + goff0) : 0u));
                stage[4 * q] = t.x; stage[4 * q + 1] = t.y; stage[4 * q + 2] = t.z; stage[4 * q + 3] = t.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < NE; ++j) {
                const int ci = cg + j * ngroups;
                stage[j] = src[ci < climit ? (unsigned)(ci * (int)SP + goff0) : 0u];
            }
        }
    };
    auto store_x = [&](int K) {
        const float* src;
        int climit;
        chunk_src(K, src, climit);
        if (!inimg) climit = 0;
        const int boff = (K & 1) * (SCH * PATCH);
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            const int ci = slot_ch(j);
            const bool own = stager && ci < SCH;
            lds[own ? boff + ci * PATCH + loff0 : trash] = ci < climit ? stage[j] : 0.f;
        }
    };

    // A fragments of this wave's chunk in super-chunk K (waves past the last chunk re-read chunk 0: harmless)
    auto a_ptr = [&](int K) {
        int ch = K * ks + kj;
        ch = ch < nchunks ? ch : 0;
        return wbase + (size_t)ch * (NS * AQ * 64);
    };
    auto load_a = [&](float4* dstv, const float4* wc, int s) {
        const float4* ws = wc + s * (AQ * 64);  // scalar base per step; q selects an immediate offset
#pragma unroll
        for (int q = 0; q < AQ; ++q) dstv[q] = ws[(unsigned)(q * 64 + lane)];
    };

    const int lane_base = kj * (CI_T * PATCH) + h2 * PATCH + r * S + c;

    // A fragments ping-pong between two register sets by step parity (rolling prefetch, also across chunk
    // boundaries); `par` = parity of the first step of the chunk, a compile-time constant per call site.
    float4 abuf[2][AQ];
    load_a(abuf[0], a_ptr(0), 0);
    load_x(0);
    store_x(0);
    __syncthreads();
    NND_STAMP(1);

    auto chunk = [&](int K, auto par_c) {
        constexpr int par = decltype(par_c)::value;
        const bool more = (K + 1 < nsuper);
#ifndef NND_DBG_NO_STAGE
        if (more) load_x(K + 1);
#endif
        const float4* wc = a_ptr(K);
        const float4* wn = a_ptr(more ? K + 1 : K);
        const bool mine = K * ks + kj < nchunks;
        const float* xb = lds + (K & 1) * (SCH * PATCH) + lane_base;
        float bq[2][NB];
        // B operands of step s, group g (8 LDS reads): operand index i = pair*P + pp
        auto read_group = [&](int s, int g, float* dst) {
            const int t = s / NGRP, grp = s % NGRP;
            const int dy = t / KW, dx = t % KW;
#pragma unroll
            for (int i = g * 8; i < g * 8 + 8; ++i) {
                const int pair = i / P, pp = i % P;
                dst[i] = xb[(grp * SG + pair * 2) * PATCH + ((dy % STR) * STR + dx % STR) * PHASE + (dy / STR) * S + dx / STR + pp * SC];
            }
        };
        constexpr int NRG = NB / 8;  // read groups per step
#pragma unroll
        for (int g = 0; g < NRG; ++g) read_group(0, g, bq[0]);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            float4* ac = abuf[(par + s) & 1];
            float4* an = abuf[(par + s + 1) & 1];
#ifndef NND_DBG_NO_ALOAD
            if (s + 1 < NS) load_a(an, wc, s + 1);
            else load_a(an, wn, 0);
#else
            for (int q = 0; q < AQ; ++q) an[q] = ac[q];
#endif
            __builtin_amdgcn_sched_barrier(0);
            // MFMAs of step s with the LDS reads of step s+1 slotted in 8 at a time, late enough that at most 8
            // reads are in flight whenever an MFMA that depends on older reads issues (lgkmcnt is in-order and
            // saturates at 15: a burst of 16 reads in front of the MFMAs would stall each step ~100 cycles)
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                if (s + 1 < NS) {
#pragma unroll
                    for (int g = 0; g < NRG; ++g) {
                        const int at = (8 * (g + 1) < NB - 4) ? 8 * (g + 1) : NB - 4 - 4 * (NRG - 1 - g);
                        if (i == at) {
                            __builtin_amdgcn_sched_barrier(0);
                            read_group(s + 1, g, bq[(s + 1) & 1]);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
                if (mine) {
                    const int pair = i / P, pp = i % P;
                    const float4 av = ac[pair / 4];
                    const float a_s = (pair % 4 == 0) ? av.x : (pair % 4 == 1) ? av.y : (pair % 4 == 2) ? av.z : av.w;
                    acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_s, bq[s & 1][i], acc[pp], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#ifndef NND_DBG_NO_STAGE
        if (more) store_x(K + 1);
#endif
#ifndef NND_DBG_NO_BARRIER
        __syncthreads();
#endif
    };
    if constexpr (NS % 2 == 0) {
        for (int K = 0; K < nsuper; ++K) chunk(K, std::integral_constant<int, 0>{});
    } else {
        for (int K = 0; K < nsuper; K += 2) {
            chunk(K, std::integral_constant<int, 0>{});
            if (K + 1 < nsuper) chunk(K + 1, std::integral_constant<int, 1>{});
        }
    }

    NND_STAMP(2);
    // ---- intra-workgroup split-K reduction through LDS (the patch buffers are free after the last barrier).
    // Every K-slice wave publishes its partial tile; slice kj then owns registers [kj*16/ks, (kj+1)*16/ks) of the
    // tile for the epilogue, so the gate math / stores of a tile are spread over all ks waves.
    constexpr int TS = P * 1024;  // floats per partial tile in LDS
    if (ks > 1) {
        if (active) {
            float* red = lds + (size_t)(cbi * ks + kj) * TS + lane;
#pragma unroll
            for (int pp = 0; pp < P; ++pp)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) red[pp * 1024 + reg * 64] = acc[pp][reg];
        }
        __syncthreads();
    }
    if (!active) return;
#ifdef NND_DBG_NO_EPI
    if (acc[0][0] != 123.456f) return;
#endif
    const int nreg = 16 / ks, reg0 = kj * nreg;  // ks is 1 or 2 (or 4): this wave's share of the tile
    if (ks > 1) {
        const float* red = lds + (size_t)(cbi * ks) * TS + lane;
#pragma unroll
        for (int pp = 0; pp < P; ++pp)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                if (reg < reg0 || reg >= reg0 + nreg) continue;
                float sum = red[pp * 1024 + reg * 64];
                for (int j = 1; j < ks; ++j) sum += red[(size_t)j * TS + pp * 1024 + reg * 64];
                acc[pp][reg] = sum;
            }
    }

    NND_STAMP(3);
    {   // ---- epilogue (conv_epilogue.h): lane holds pixel (y, x_pp) and 16 output channels
        int ys[P], xs[P];
#pragma unroll
        for (int pp = 0; pp < P; ++pp) {
            ys[pp] = ty0 + r;
            xs[pp] = tx0 + pp * SC + c;
        }
        conv_epilogue<P>(a, acc, cb, b, h2, reg0, nreg, ys, xs);
    }
#ifdef NND_DBG_STAMPS
    __builtin_amdgcn_s_waitcnt(0);
#endif
    NND_STAMP(4);
}

// 1x1 stride-1 convs with 64..128 input channels on tile-major tensors (the projection shortcuts of the encoder's residual
// blocks, cnet_proj): no halo, so nothing needs staging — the 32 pixels of a sub-tile are one 128-B line per channel and
// are loaded straight into the MFMA B operand.  One wave = one 32-channel output block, its NQ*4 weight fragments held in
// registers, looping over TPW sub-tiles; the waves of the other output blocks of the same sub-tiles are its neighbours
// (their loads hit L1).  HBM-bound by construction (2 flop/B at 64 -> 64): the conv_mfma path spent 85 us on the
// 134 MB of a 64 -> 64 shortcut at 272x480x2 (1.6 TB/s), mostly per-workgroup staging / barrier overhead.
template <int NQ, int TPW>
__global__ void __launch_bounds__(256) conv1x1_stream_kernel(ConvArgs a, int ntiles, int ncb, int KQ) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h2 = lane >> 5, l31 = lane & 31;
    const long gw = (long)blockIdx.x * 4 + wave;
    const int cb = (int)(gw % ncb), b = blockIdx.z;
    const int t0 = (int)(gw / ncb) * TPW;
    if (t0 >= ntiles) return;
    const long SP = a.ls.plane, DP = a.ld.plane;
    float4 afr[NQ];
    const float4* wq = reinterpret_cast<const float4*>(a.wpk) + (size_t)cb * KQ * 64 + lane;
#pragma unroll
    for (int q = 0; q < NQ; ++q) afr[q] = wq[q * 64];
    const int epi = a.epi;
    float bias_r[16], sc_r[16];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int co = cb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h2;
        bias_r[reg] = co < a.Cout ? a.bias[co] : 0.f;
        sc_r[reg] = (epi == EPI_AFFINE && co < a.Cout) ? a.cscale[co] : 1.f;
    }
    const int tend = min(t0 + TPW, ntiles);
    for (int t = t0; t < tend; ++t) {
        const int y = (t / a.tiles_x) * 4 + (l31 >> 3), x = (t % a.tiles_x) * 8 + (l31 & 7);
        const bool pix_ok = y < a.H && x < a.W;
        const float* src = a.src0 + b * a.bs0 + (pix_ok ? pix_off(a.ls, y, x) : 0);  // tile-major: t*32 + l31, one line per channel
        float bv[NQ * 4];
        if (a.ls.ci == 4) {
            // 4 channels interleaved: the lower half-wave loads channel quad 2j of its pixel, the upper one quad 2j + 1 (one 16-B load
            // per lane and 8 channels; every byte requested once).  The B operand of k pair (c, c + 1) wants channel c in the lower
            // and c + 1 in the upper half: v_permlane32_swap exchanges the upper half of one register with the lower half of
            // another — (a0|b0),(a1|b1) -> (a0|a1),(b0|b1) — so two swaps per 8 channels put every pair in place.  Same k order
            // as the planar loads: bit-identical results.
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                const float4 q4 = *reinterpret_cast<const float4*>(src + (long)(2 * j + h2) * 4 * SP);
                const auto s01 = __builtin_amdgcn_permlane32_swap(__float_as_uint(q4.x), __float_as_uint(q4.y), false, false);
                const auto s23 = __builtin_amdgcn_permlane32_swap(__float_as_uint(q4.z), __float_as_uint(q4.w), false, false);
                bv[4 * j + 0] = __uint_as_float(s01[0]);  // (a0 | a1): channels 8j, 8j+1
                bv[4 * j + 1] = __uint_as_float(s23[0]);  // (a2 | a3)
                bv[4 * j + 2] = __uint_as_float(s01[1]);  // (b0 | b1): channels 8j+4, 8j+5
                bv[4 * j + 3] = __uint_as_float(s23[1]);  // (b2 | b3)
            }
        } else {
#pragma unroll
            for (int kp = 0; kp < NQ * 4; ++kp) bv[kp] = src[(long)(2 * kp + h2) * SP];
        }
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int kp = 0; kp < NQ * 4; ++kp) {
            const float4 av = afr[kp / 4];
            const float a_s = (kp % 4 == 0) ? av.x : (kp % 4 == 1) ? av.y : (kp % 4 == 2) ? av.z : av.w;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_s, bv[kp], acc, 0, 0, 0);
        }
        if (!pix_ok) continue;
        const long pix = pix_off(a.ld, y, x);
        if (a.ld.ci == 4) {  // registers 4q..4q+3 are 4 consecutive channels: one 16-B residual load / store per group (Cout % 4 == 0)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int co0 = cb * 32 + 8 * q + 4 * h2;
                if (co0 >= a.Cout) continue;
                float4 r4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (epi == EPI_AFFINE && a.aux0) r4 = *reinterpret_cast<const float4*>(a.aux0 + b * a.abs0 + (long)co0 * DP + pix);
                const float rr[4] = {r4.x, r4.y, r4.z, r4.w};
                float o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int reg = 4 * q + i;
                    float v;
                    if (epi == EPI_AFFINE) {
                        v = fmaf(acc[reg], sc_r[reg], bias_r[reg]);
                        if (a.flags & 4) v = v > 0.f ? v : a.scale * v;
                        if (a.flags & 1) v = fmaxf(v, 0.f);
                        if (a.aux0) v = rr[i] + v;
                        if (a.flags & 2) v = fmaxf(v, 0.f);
                    } else {
                        v = acc[reg] + bias_r[reg];
                        if (epi == EPI_RELU) v = fmaxf(v, 0.f);
                        else if (epi == EPI_SCALE) v = a.scale * v;
                    }
                    o[i] = v;
                }
                *reinterpret_cast<float4*>(a.out0 + b * a.obs0 + (long)co0 * DP + pix) = make_float4(o[0], o[1], o[2], o[3]);
            }
            continue;
        }
        float res[16];
        if (epi == EPI_AFFINE && a.aux0) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int co = cb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h2;
                res[reg] = co < a.Cout ? a.aux0[b * a.abs0 + co * DP + pix] : 0.f;
            }
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int co = cb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h2;
            if (co >= a.Cout) continue;
            float v;
            if (epi == EPI_AFFINE) {
                v = fmaf(acc[reg], sc_r[reg], bias_r[reg]);
                if (a.flags & 4) v = v > 0.f ? v : a.scale * v;
                if (a.flags & 1) v = fmaxf(v, 0.f);
                if (a.aux0) v = res[reg] + v;
                if (a.flags & 2) v = fmaxf(v, 0.f);
            } else {
                v = acc[reg] + bias_r[reg];
                if (epi == EPI_RELU) v = fmaxf(v, 0.f);
                else if (epi == EPI_SCALE) v = a.scale * v;
            }
            a.out0[b * a.obs0 + co * DP + pix] = v;
        }
    }
}

template <int NQ>
static void launch_conv1x1_stream(const ConvArgs& a, int ntiles, int ncb, int KQ, int B, hipStream_t stream) {
    // one sub-tile per wave: two per wave (weight fragments amortised, 172 VGPRs) measured 110 vs 45 us on the 64 -> 64 shortcut
    hipLaunchKernelGGL((conv1x1_stream_kernel<NQ, 1>), dim3((unsigned)cdiv64((long)ncb * ntiles, 4), 1, B), dim3(256), 0, stream, a, ntiles, ncb, KQ);
}

// --------------------------------------------------------------------------- host side
struct TileCfg {
    int P, wco, ks, tiles_x, tiles_y, npos, ngroups, ne;
    size_t lds;
};

// Chooses the pixel sub-tile shape (SR x SC), P sub-tiles per wave and wco x ks waves per workgroup.
// Cost model: every wave issues unit = ceil(nchunks/ks) * P MFMA streams; waves spread evenly over the 1024
// SIMDs of the chip, so the busiest SIMD runs ceil(waves/1024) * unit (quantisation is what matters at
// batch 1: 68x120 = 255 tiles of 4x8 pixels).
static bool pick_tile(const ConvLayer& L, int c0, int c1, int B, int H, int W, TileCfg* out) {
    const int force_p = switches().conv_p, force_ks = switches().conv_ks, force_wco = switches().conv_wco;
    double best = 1e30;
    bool found = false;
    for (int ks : {1, 2})  // ks = 4 measured slower on every layer (scripts/sweep_conv.py)
    for (int wco : {8, 6, 4, 3, 2, 1}) {
        if (force_ks > 0 && ks != force_ks) continue;
        if (force_wco > 0 && wco != force_wco) continue;
        if (ks > L.nchunks || wco * ks > 12) continue;  // P = 1 kernels are built for <= 768 threads, P = 2 for <= 512
        if (c1 > 0 && c0 % (ks * L.CI_T) != 0) continue;
        if (wco > 1 && cdiv(L.ncb, wco) * wco >= L.ncb + wco) continue;  // a whole workgroup of idle waves
        const int nthreads = 64 * wco * ks;
        {
            for (int P : {1}) {  // P = 2 (two sub-tiles per wave) measured slower on every config: 256 VGPRs + scratch
                if (force_p > 0 && P != force_p) continue;
                if (P == 2 && wco * ks > 8) continue;
                const int SC = NND_SC, SR = 32 / NND_SC;
                int tx = cdiv(W, P * SC), ty = cdiv(H, SR);
                const int st = L.stride;
                if (st == 2 && P != 1) continue;  // stride-2 kernels are instantiated for P = 1 only
                int PR = (SR - 1) * st + L.KH, PC = (P * SC - 1) * st + L.KW;
                int PRH = (PR + st - 1) / st, PCH = (PC + st - 1) / st;
                int S = patch_stride(PCH, SC);
                int npos = PR * PC;
                if (npos > nthreads) continue;
                int ngroups = nthreads / npos;
                int ne = cdiv(ks * L.CI_T, ngroups);
                if (ne > 16) continue;
                size_t lds = ((size_t)2 * ks * L.CI_T * st * st * PRH * S + 1) * sizeof(float);
                size_t red = ks > 1 ? (size_t)wco * ks * P * 1024 * sizeof(float) : 0;
                if (red > lds) lds = red;
                if (lds > 160 * 1024) continue;
                int occ = P == 1 ? (ne <= 8 ? 4 : 3) : 2;  // resident waves per SIMD the register budget allows
                int wg_per_cu = (int)((160 * 1024) / lds);
                int occ_lds = cdiv(wg_per_cu * wco * ks, 4);
                if (occ_lds < occ) occ = occ_lds;
                if (occ < 1) occ = 1;
                double waves = (double)tx * ty * B * cdiv(L.ncb, wco) * wco * ks;
                double unit = (double)cdiv(L.nchunks, ks) * P;
                double per_simd = std::ceil(waves / 1024.0);
                double t = (per_simd <= occ) ? per_simd * unit : std::ceil(waves / (1024.0 * occ)) * occ * unit;
                // a workgroup spreads its wco*ks waves round-robin over the 4 SIMDs of its CU: 6 waves load them (2,2,1,1), and
                // measured launches of such workgroups run like 8 waves each (flow_head.conv1+mask.0: 78 vs 64 us)
                t *= std::ceil(wco * ks / 4.0) * 4.0 / (wco * ks);
                if (waves <= 1024.0) t *= 1.10;             // one wave per SIMD cannot hide its own stalls
                t *= 1.0 + 0.02 * (4 - wco);                 // fewer waves share one staged patch
                t *= 1.0 - 0.02 * (P - 1);                   // larger P: fewer weight bytes per flop
                t *= 1.0 + 0.01 * (ks - 1);                  // reduction cost
                if (t < best) {
                    best = t;
                    *out = {P, wco, ks, tx, ty, npos, ngroups, ne, lds};
                    found = true;
                }
            }
        }
    }
    return found;
}

template <int KH, int KW, int CI_T, int P, int NE, int STR = 1>
static int launch_one(const ConvArgs& a, dim3 grid, dim3 block, size_t lds, hipStream_t stream) {
    auto kern = conv_mfma_kernel<KH, KW, CI_T, P, NE, STR>;
    if (lds > 64 * 1024) {
        static std::atomic<unsigned> raised{0};
        if (int rc = raise_lds_limit(reinterpret_cast<const void*>(kern), raised)) return rc;
    }
    hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
    return NND_OK;
}

template <int KH, int KW, int CI_T>
static int launch_shape(const ConvArgs& a, const TileCfg& cfg, dim3 grid, dim3 block, hipStream_t stream) {
    if (cfg.ne <= 8) return launch_one<KH, KW, CI_T, 1, 8>(a, grid, block, cfg.lds, stream);
    return launch_one<KH, KW, CI_T, 1, 16>(a, grid, block, cfg.lds, stream);
}

int launch_conv(const ConvLayer& L, const float* blob, const ConvIO& io, int epi, int B, int H, int W,
                hipStream_t stream) {
    if (L.arith != 0) return launch_conv_split(L, blob, io, epi, B, H, W, stream);
    NND_REQUIRE(io.src0.C + io.src1.C == L.Cin, "conv: source channels %d+%d != Cin %d", io.src0.C, io.src1.C, L.Cin);
    NND_REQUIRE(io.src1.C == 0 || io.src0.C % L.CI_T == 0, "conv: first source (%d ch) must be a multiple of %d", io.src0.C, L.CI_T);
    const int Hin = io.Hin > 0 ? io.Hin : H, Win = io.Win > 0 ? io.Win : W;
    NND_REQUIRE(L.stride == 1 || L.stride == 2, "conv: stride %d not supported", L.stride);
    NND_REQUIRE(H == (Hin + L.stride - 1) / L.stride && W == (Win + L.stride - 1) / L.stride,
                "conv: output %dx%d does not match input %dx%d at stride %d", H, W, Hin, Win, L.stride);
    NND_REQUIRE((long)(L.Cin + 2 * L.CI_T) * tiled_plane(Hin, Win) < (1L << 31), "conv: plane offsets exceed 32 bits");
    TileCfg cfg;
    NND_REQUIRE(pick_tile(L, io.src0.C, io.src1.C, B, H, W, &cfg), "conv: no tile configuration for %dx%d Cin=%d", L.KH, L.KW, L.Cin);
    {   // LDS sizing rule, re-derived independently of pick_tile (DESIGN.md §4 "staging bounds"): two patch buffers of
        // ks*CI_T channels + the spare word that swallows the stores of non-staging threads, and — aliasing them after the
        // last barrier — one 32x32 partial tile per wave for the split-K exchange; every staging thread needs a slot.
        const int SR_ = 32 / NND_SC, st = L.stride;
        const int PR_ = (SR_ - 1) * st + L.KH, PC_ = (cfg.P * NND_SC - 1) * st + L.KW;
        const size_t patch = (size_t)st * st * ((PR_ + st - 1) / st) * patch_stride((PC_ + st - 1) / st, NND_SC);
        NND_REQUIRE(cfg.lds >= ((size_t)2 * cfg.ks * L.CI_T * patch + 1) * sizeof(float) &&
                        (cfg.ks == 1 || cfg.lds >= (size_t)cfg.wco * cfg.ks * cfg.P * 1024 * sizeof(float)) &&
                        cfg.lds <= 160 * 1024,
                    "conv: LDS plan %zu B does not cover the patch buffers / split-K tiles", cfg.lds);
        NND_REQUIRE(cfg.npos == PR_ * PC_ && cfg.npos * cfg.ngroups <= 64 * cfg.wco * cfg.ks &&
                        cfg.ngroups * cfg.ne >= cfg.ks * L.CI_T && cfg.ne <= 16,
                    "conv: staging plan (%d positions x %d groups x %d) does not cover %d channels", cfg.npos, cfg.ngroups,
                    cfg.ne, cfg.ks * L.CI_T);
    }
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.src0 = io.src0.ptr; a.bs0 = io.src0.bstride; a.c0 = io.src0.C;
    a.src1 = io.src1.ptr; a.bs1 = io.src1.bstride; a.c1 = io.src1.C;
    a.wpk = blob + L.w_off;
    a.bias = blob + L.b_off;
    a.out0 = io.out0.ptr; a.obs0 = io.out0.bstride;
    a.out1 = io.out1.ptr; a.obs1 = io.out1.bstride;
    a.aux0 = io.aux0.ptr; a.abs0 = io.aux0.bstride;
    a.aux1 = io.aux1.ptr; a.abs1 = io.aux1.bstride;
    a.bmap = io.bmap.ptr; a.bmbs = io.bmap.bstride;
    a.ls = make_lay(Hin, Win, io.src_tiled, io.src_c4);
    a.ld = make_lay(H, W, io.dst_tiled, io.dst_c4);
    NND_REQUIRE(!io.src_c4 || (io.src_tiled && L.stride == 1 && io.src0.C % 4 == 0 && io.src1.C % 4 == 0),
                "conv: c4 sources need stride 1 and channel counts %% 4 == 0");
    // the streaming 1x1 kernel: tile-major sources, planar or 4-channel-interleaved (then Cout % 4 == 0 for a c4 destination)
    NND_REQUIRE(!io.dst_c4 || (io.dst_tiled && (L.Cout % 4 == 0 || (!io.bmap.ptr && !io.aux0.ptr && !io.aux1.ptr && !io.out1.ptr))),
                "conv: c4 destination with per-pixel operands needs Cout %% 4 == 0");
    a.H = H; a.W = W; a.Cout = L.Cout; a.nchunks = L.nchunks; a.epi = epi; a.hidden = io.hidden;
    a.Hin = Hin; a.Win = Win; a.flags = io.flags;
    a.cscale = L.s_off >= 0 ? blob + L.s_off : nullptr;
    NND_REQUIRE(epi != EPI_AFFINE || a.cscale, "conv: EPI_AFFINE needs a packed scale vector");
    a.tiles_x = cfg.tiles_x; a.wco = cfg.wco; a.ks = cfg.ks;
    a.npos = cfg.npos; a.ngroups = cfg.ngroups;
    a.scale = io.scale;
    const bool no_stream = switches().no_conv1x1_stream;
    if (!no_stream && L.KH == 1 && L.KW == 1 && L.stride == 1 && io.src1.C == 0 && io.src_tiled && (!io.dst_c4 || L.Cout % 4 == 0) && !io.bmap.ptr &&
        (L.Cin == 64 || L.Cin == 96 || L.Cin == 128) &&
        (epi == EPI_LINEAR || epi == EPI_RELU || epi == EPI_SCALE || epi == EPI_AFFINE)) {
        const int ntiles = cfg.tiles_x * cfg.tiles_y, KQ = L.nchunks * L.CI_T / 8;
        if (L.Cin == 64) launch_conv1x1_stream<8>(a, ntiles, L.ncb, KQ, B, stream);
        else if (L.Cin == 96) launch_conv1x1_stream<12>(a, ntiles, L.ncb, KQ, B, stream);
        else launch_conv1x1_stream<16>(a, ntiles, L.ncb, KQ, B, stream);
        NND_LAUNCH_CHECK();
        return NND_OK;
    }
    dim3 grid(cfg.tiles_x * cfg.tiles_y, cdiv(L.ncb, cfg.wco), B), block(64 * cfg.wco * cfg.ks);
    const bool verbose = switches().conv_verbose;
    if (verbose)
        fprintf(stderr, "[nnd] conv %dx%d Cin=%d Cout=%d CI_T=%d: P=%d, wco=%d, ks=%d, ne=%d, grid %ux%ux%u, lds %zu B\n",
                L.KH, L.KW, L.Cin, L.Cout, L.CI_T, cfg.P, cfg.wco, cfg.ks, cfg.ne, grid.x, grid.y, grid.z, cfg.lds);
    int rc = NND_ERR_UNSUPPORTED;
    if (L.stride == 2) {
        if (L.KH == 3 && L.KW == 3 && L.CI_T == 16) rc = launch_one<3, 3, 16, 1, 16, 2>(a, grid, block, cfg.lds, stream);
        else if (L.KH == 1 && L.KW == 1 && L.CI_T == 16) rc = launch_one<1, 1, 16, 1, 16, 2>(a, grid, block, cfg.lds, stream);
        else set_error("conv %dx%d stride 2 CI_T=%d not instantiated", L.KH, L.KW, L.CI_T);
    } else
    if (L.KH == 1 && L.KW == 1 && L.CI_T == 128) rc = launch_shape<1, 1, 128>(a, cfg, grid, block, stream);
    else if (L.KH == 1 && L.KW == 1 && L.CI_T == 32) rc = launch_shape<1, 1, 32>(a, cfg, grid, block, stream);
    else if (L.KH == 3 && L.KW == 3 && L.CI_T == 32) rc = launch_shape<3, 3, 32>(a, cfg, grid, block, stream);
    else if (L.KH == 3 && L.KW == 3 && L.CI_T == 16) rc = launch_shape<3, 3, 16>(a, cfg, grid, block, stream);
    else if (L.KH == 1 && L.KW == 5 && L.CI_T == 32) rc = launch_shape<1, 5, 32>(a, cfg, grid, block, stream);
    else if (L.KH == 5 && L.KW == 1 && L.CI_T == 32) rc = launch_shape<5, 1, 32>(a, cfg, grid, block, stream);
    else if (L.KH == 1 && L.KW == 5 && L.CI_T == 64) rc = launch_shape<1, 5, 64>(a, cfg, grid, block, stream);
    else if (L.KH == 5 && L.KW == 1 && L.CI_T == 64) rc = launch_shape<5, 1, 64>(a, cfg, grid, block, stream);
    else set_error("conv %dx%d CI_T=%d not instantiated", L.KH, L.KW, L.CI_T);
    if (rc != NND_OK) return rc;
    NND_LAUNCH_CHECK();
    return NND_OK;
}

// input channels per K-chunk for a layer shape (the host packer and the kernels must agree)
int conv_ci_t(int KH, int KW, int Cin, int stride, int Cout) {
    if (stride == 2) return 16;  // the phase-split stride-2 patch is 4x larger per channel
    // shallow 3x3 layers (encoder layer1, Cin = 64): half-size chunks halve the LDS patch and the staging registers (NE 8
    // instead of 16, 125 instead of 165 VGPRs), so twice as many of their small workgroups fit on a CU: 72 -> 89 TFLOP/s
    if (KH == 3 && KW == 3 && Cin <= 64) return 16;
    // wide GRU gate convs (1x5 / 5x1, z and r in one launch): 64-channel chunks halve the barriers per MFMA and let the
    // picker use 8 output-channel waves without split-K (ne stays 8): 63.5 -> 60.9 / 61.4 -> 57.7 us at 68x120
    if (KH * KW == 5 && Cin >= 256 && Cout >= 256) return 64;
    return (KH == 1 && KW == 1 && Cin >= 128) ? 128 : 32;
}

// The packed layer may take only a subset of the source tensor's input channels: packed channel ci reads source
// channel ci_map[ci] of a (cout, cin_src, KH, KW) tensor (ci_map == nullptr: identity, cin_src = L.Cin).
// bvec[part] == nullptr packs a zero bias (the caller adds it elsewhere).
void pack_conv(const ConvLayer& L, int nparts, const float* const* w, const float* const* bvec,
               const int* cout, float* blob, const int* ci_map, int cin_src) {
    if (L.arith != 0) return pack_conv_split(L, nparts, w, bvec, cout, blob, ci_map, cin_src);
    if (!ci_map) cin_src = L.Cin;
    const int NT = L.KH * L.KW, NQ = L.CI_T / 8;
    float* wp = blob + L.w_off;
    float* bp = blob + L.b_off;
    memset(wp, 0, sizeof(float) * L.w_floats());
    memset(bp, 0, sizeof(float) * L.b_floats());
    // blob order: [cb][chunk][tap][q][lane][j]; lane = h2*32 + (co%32); channel in chunk = (q*4+j)*2 + h2.
    // For CI_T >= 32 the kernel walks (tap, 32-channel group) steps, i.e. q = g*4 + q' — the same linear order.
    int co0 = 0;
    for (int part = 0; part < nparts; ++part) {
        for (int col = 0; col < cout[part]; ++col) {
            int co = co0 + col;
            int cb = co / 32, i = co % 32;
            bp[co] = bvec[part] ? bvec[part][col] : 0.f;
            for (int ci = 0; ci < L.Cin; ++ci) {
                int chunk = ci / L.CI_T, cl = ci % L.CI_T;
                int pair = cl / 2, h2 = cl % 2;
                int q = pair / 4, j = pair % 4;
                int lane = h2 * 32 + i;
                for (int t = 0; t < NT; ++t) {
                    size_t idx = (((((size_t)cb * L.nchunks + chunk) * NT + t) * NQ + q) * 64 + lane) * 4 + j;
                    wp[idx] = w[part][((size_t)col * cin_src + (ci_map ? ci_map[ci] : ci)) * NT + t];
                }
            }
        }
        co0 += cout[part];
    }
}

#ifdef NND_DBG_STAMPS
extern "C" int nnd_debug_read_stamps(unsigned long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -1;
}
#endif
}  // namespace nnd
