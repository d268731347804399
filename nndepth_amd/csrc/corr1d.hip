// 1-D all-pairs correlation pyramid (build) and multi-radius lookup for RAFT-Stereo style models.
//
// Replaces nndepth/models/raft_stereo/cost_volume.py:12-61 and nndepth/models/raft_stereo/utils.py:4-27
// (reference; semantics restated in oracle/torch_ref.py: corr1d_build / corr1d_lookup).
//
// build  : per image row h, D[w1][w2] = sum_c f1[c,h,w1] * f2[c,h,w2] on the fp32 MFMA (32x32x2).
//          NCHW puts 32 consecutive w of one channel in one 128-B segment, which is exactly the
//          A (f1, i = w1) and B (f2, j = w2) fragment of a k-step, so both operands stream
//          global -> VGPR fully coalesced with no LDS; the avg-pool levels are produced from the
//          accumulator with lane shuffles (2^l consecutive w2 live in 2^l consecutive lanes) and all
//          levels are written in the same pass (HBM-bound: reads 2*C*H*W*4 B, writes ~1.9*H*W*W*4 B).
// lookup : one thread per (level, tap, pixel) output element; consecutive lanes = consecutive pixels,
//          so the (B,36,H,W) output is written coalesced; the two gathers per element hit the pixel's
//          own pyramid row (<= 2 cache lines per level).
//
// This file is compiled with -ffp-contract=off so the lookup reproduces the reference's
// mul/mul/add rounding sequence bit for bit.
#include <algorithm>
#include "common.h"
#include "flow_branch.h"
#include "layout.h"

namespace nnd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int MAX_LEVELS = 8;

struct PyrLayout {
    long off[MAX_LEVELS];
    int width[MAX_LEVELS];
    int nlev;  // stored levels
};

static void make_layout(int B, int H, int W, int stored_levels, PyrLayout* L, int64_t* total) {
    long off = 0;
    int w = W;
    L->nlev = stored_levels;
    for (int l = 0; l < stored_levels; ++l) {
        L->off[l] = off;
        L->width[l] = w;
        off += (long)B * H * W * w;
        w /= 2;
    }
    if (total) *total = off;
}

// grid: (ceil(W/32) w1-blocks, H, B); block: 256 threads = 4 waves, wave t takes w2 tiles t, t+4, ...
// Group-wise variant (IGEV, igev_stereo/cost_volume.py:81-98): blockIdx.z = b*G + g, the dot product runs over
// the C channels [g*C, (g+1)*C) of a Ctot-channel map; pyramid rows are ordered (b, g, h, w1).  RAFT: G = 1.
template <int KB>
__global__ void __launch_bounds__(256) corr1d_build_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                           float* __restrict__ pyr, PyrLayout L, int C, int H, int W,
                                                           float rscale_div, int Ctot, int G) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, h2 = lane >> 5;
    const int w1_0 = blockIdx.x * 32, h = blockIdx.y, b = blockIdx.z;
    const long HW = (long)H * W;
    const long chan0 = (long)(b / G) * Ctot + (long)(b % G) * C;
    const float* a_base = f1 + chan0 * HW + (long)h * W;
    const float* b_base = f2 + chan0 * HW + (long)h * W;
    const int w1 = w1_0 + l31;
    const bool a_ok = w1 < W;
    const int ntile = (W + 31) / 32;
    for (int t = wave; t < ntile; t += 4) {
        const int w2 = t * 32 + l31;
        const bool b_ok = w2 < W;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        // K loop in batches of KB k-steps (2 channels each), double-buffered: the 2*KB loads of batch i+1 are in flight
        // while the KB MFMAs of batch i issue (one wave per SIMD here, so latency is hidden by depth, not occupancy).
        // Loads are unconditional on clamped indices; the zero-fill select happens when the batch is consumed.
        const int w1c = min(w1, W - 1), w2c = min(w2, W - 1);
        const int nk = (C + 1) / 2, nbatch = (nk + KB - 1) / KB;
        float av[2][KB], bv[2][KB];
        auto load = [&](int bt, float* a, float* bb) {
#pragma unroll
            for (int i = 0; i < KB; ++i) {
                const int cc = min((bt * KB + i) * 2 + h2, C - 1);
                a[i] = a_base[cc * HW + w1c];
                bb[i] = b_base[cc * HW + w2c];
            }
        };
        auto mma = [&](int bt, const float* a, const float* bb) {
#pragma unroll
            for (int i = 0; i < KB; ++i) {
                const bool k_ok = (bt * KB + i) * 2 + h2 < C;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a_ok && k_ok) ? a[i] : 0.f, (b_ok && k_ok) ? bb[i] : 0.f, acc, 0, 0, 0);
            }
        };
        load(0, av[0], bv[0]);
        for (int bt = 0; bt < nbatch; bt += 2) {
            if (bt + 1 < nbatch) load(bt + 1, av[1], bv[1]);
            mma(bt, av[0], bv[0]);
            if (bt + 1 < nbatch) {
                if (bt + 2 < nbatch) load(bt + 2, av[0], bv[0]);
                mma(bt + 1, av[1], bv[1]);
            }
        }
        // D[i = w1 row][j = w2 col]: lane holds column w2, rows (reg&3)+8*(reg>>2)+4*h2
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h2;
            const int ww1 = w1_0 + row;
            float v = acc[reg] / rscale_div;
            const long prow = ((long)b * H + h) * W + ww1;  // pyramid row of pixel (b,h,w1)
            const bool row_ok = ww1 < W;
            if (row_ok && b_ok) pyr[L.off[0] + prow * L.width[0] + w2] = v;
            int wcur = w2;
#pragma unroll
            for (int l = 1; l < MAX_LEVELS; ++l) {
                if (l >= L.nlev) break;
                float other = __shfl_xor(v, 1 << (l - 1));
                v = (v + other) * 0.5f;
                wcur >>= 1;
                const bool owner = (l31 & ((1 << l) - 1)) == 0;
                if (row_ok && owner && wcur < L.width[l]) pyr[L.off[l] + prow * L.width[l] + wcur] = v;
            }
        }
    }
}

// Same result (same MFMA sequence per output, same pooling arithmetic: bit-identical) with both operands staged through
// LDS and the output block written through LDS:
//   * loads: the 32-channel chunk of the f1 block (32 columns) and of the whole f2 row are fetched with 16-byte coalesced
//     loads into registers while the previous chunk is multiplied (the kernel above issues one 4-byte load per lane and
//     MFMA operand, 2 per MFMA, and is bound by the load-issue rate: 0.53 TB/s at 68x120, 0.63 TB/s at batch 8);
//   * stores: the 32 pyramid rows of a workgroup are contiguous in every level, so after pooling in LDS each level leaves
//     as one linear coalesced copy (the kernel above stores 16 + 4x16 partial rows per tile from the accumulator layout).
// grid (ceil(W/32), H, B*G), 256 threads; wave t multiplies the w2 tiles t, t+4, ... (NT of them).
constexpr int CB_MAXLD = 12;
template <int NT>
__global__ void __launch_bounds__(256) corr1d_build_lds_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                               float* __restrict__ pyr, PyrLayout L, int C, int H, int W,
                                                               float rscale_div, int Ctot, int G, int KC) {
    extern __shared__ float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h2 = lane >> 5;
    const int w1_0 = blockIdx.x * 32, h = blockIdx.y, b = blockIdx.z;
    const long HW = (long)H * W;
    const long chan0 = (long)(b / G) * Ctot + (long)(b % G) * C;
    const float* a_base = f1 + chan0 * HW + (long)h * W;
    const float* b_base = f2 + chan0 * HW + (long)h * W;
    const int ntile = (W + 31) / 32, WP = ntile * 32;
    float* sA = sm;             // [KC][32]
    float* sB = sm + KC * 32;   // [KC][WP]
    const bool vec = (W & 3) == 0;
    const int UPC = 8 + WP / 4, nunits = KC * UPC;  // float4 units per channel: 8 of the f1 block, WP/4 of the f2 row
    float4 st[CB_MAXLD];
    int ucode[CB_MAXLD];  // chunk-invariant decomposition of this thread's units: channel << 16 | column << 1 | isA
#pragma unroll
    for (int k = 0; k < CB_MAXLD; ++k) {
        const int u = tid + 256 * k, ch = u / UPC, r = u - ch * UPC;
        ucode[k] = (ch << 16) | ((r < 8 ? 4 * r : 4 * (r - 8)) << 1) | (r < 8 ? 1 : 0);
    }
    auto unit = [&](int k, int& ch, int& col, bool& isA) {
        ch = ucode[k] >> 16;
        col = (ucode[k] & 0xffff) >> 1;
        isA = ucode[k] & 1;
    };
    auto fetch = [&](int c0) {
#pragma unroll
        for (int k = 0; k < CB_MAXLD; ++k) {
            const int u = tid + 256 * k;
            if (u < nunits) {
                int ch, col;
                bool isA;
                unit(k, ch, col, isA);
                const int c = c0 + ch, gcol = isA ? w1_0 + col : col;
                const float* src = (isA ? a_base : b_base) + (long)min(c, C - 1) * HW;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (c < C) {
                    if (vec) {
                        if (gcol < W) v = *reinterpret_cast<const float4*>(src + gcol);
                    } else {
                        if (gcol + 0 < W) v.x = src[gcol + 0];
                        if (gcol + 1 < W) v.y = src[gcol + 1];
                        if (gcol + 2 < W) v.z = src[gcol + 2];
                        if (gcol + 3 < W) v.w = src[gcol + 3];
                    }
                }
                st[k] = v;
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int k = 0; k < CB_MAXLD; ++k) {
            const int u = tid + 256 * k;
            if (u < nunits) {
                int ch, col;
                bool isA;
                unit(k, ch, col, isA);
                *reinterpret_cast<float4*>((isA ? sA + ch * 32 : sB + ch * WP) + col) = st[k];
            }
        }
    };
    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    const int nchunk = (C + KC - 1) / KC;
    fetch(0);
    for (int ck = 0; ck < nchunk; ++ck) {
        __syncthreads();
        commit();
        __syncthreads();
        if (ck + 1 < nchunk) fetch((ck + 1) * KC);
        const int nstep = min(KC, C - ck * KC + 1) / 2;  // k-steps of 2 channels that hold real channels
        // GS k-steps at a time with all their LDS reads issued before the first MFMA (at most 16 reads: lgkmcnt saturates at 15)
        constexpr int GS = NT == 1 ? 8 : 1;  // measured: batching pays only with one tile per wave (68x120: 47.6 -> 34.8 us; 8x48x156: 245 -> 305 us)
        for (int s0 = 0; s0 < nstep; s0 += GS) {
            float av[GS], bv[GS][NT];
#pragma unroll
            for (int ss = 0; ss < GS; ++ss) {
                const int row = (2 * min(s0 + ss, nstep - 1) + h2);
                av[ss] = sA[row * 32 + l31];
#pragma unroll
                for (int j = 0; j < NT; ++j) bv[ss][j] = sB[row * WP + min(wave + 4 * j, ntile - 1) * 32 + l31];
            }
#pragma unroll
            for (int ss = 0; ss < GS; ++ss)
                if (s0 + ss < nstep) {
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        if (wave + 4 * j < ntile) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ss], bv[ss][j], acc[j], 0, 0, 0);
                }
        }
    }
    __syncthreads();  // the operand buffers become the output block: level 0 [32][W], level l behind it [32][W_l]
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int t = wave + 4 * j, col = t * 32 + l31;
        if (t < ntile && col < W)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) sm[((reg & 3) + 8 * (reg >> 2) + 4 * h2) * W + col] = acc[j][reg] / rscale_div;
    }
    __syncthreads();
    int lo = 0;
    for (int l = 1; l < L.nlev; ++l) {
        const int wp = L.width[l - 1], wl = L.width[l];
        const float* src = sm + lo;
        float* dst = sm + lo + 32 * wp;
        for (int idx = tid; idx < 32 * wl; idx += 256) {
            const int i = idx / wl, jx = idx - i * wl;
            dst[idx] = (src[i * wp + 2 * jx] + src[i * wp + 2 * jx + 1]) * 0.5f;
        }
        lo += 32 * wp;
        __syncthreads();
    }
    const int nrows = min(32, W - w1_0);
    const long prow0 = ((long)b * H + h) * W + w1_0;  // first pyramid row of the block
    lo = 0;
    for (int l = 0; l < L.nlev; ++l) {
        const int wl = L.width[l];
        const long g0 = L.off[l] + prow0 * wl;
        float* g = pyr + g0;
        const int cnt = nrows * wl;
        if (((g0 | cnt) & 3) == 0) {  // the block's rows of a level are one contiguous run: 16-byte stores when it is aligned
            for (int idx = tid; idx < cnt >> 2; idx += 256)
                reinterpret_cast<float4*>(g)[idx] = *reinterpret_cast<const float4*>(sm + lo + 4 * idx);
        } else {
            for (int idx = tid; idx < cnt; idx += 256) g[idx] = sm[lo + idx];
        }
        lo += 32 * wl;
    }
}

// Small problems (one pair at 1/4 resolution: 272 workgroups of the kernel above = one wave per SIMD, every load latency
// exposed): one 32 x 32 output tile per workgroup and the channel range split over its 4 waves, 4 x the waves in flight.
// Operands go global -> register (lane l of a k-step holds f[2s + l/32][x0 + l%32]: one 128-byte run per half wave), all
// loads of a KB-step batch issued before its first MFMA; the four partial tiles meet in LDS and are summed in wave order
// (q = 0..3: the result differs from the full-K kernels in the last bits, deterministically).  Pooling stays inside the
// tile (tile origins are multiples of 32 >= 2^levels) through lane exchanges; every level leaves as row segments.
// 1-D grid, remapped so that the tiles of one image row (which share their f1 / f2 row) run on one XCD's L2.
template <int KB>
__global__ void __launch_bounds__(256) corr1d_build_ksplit_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                                  float* __restrict__ pyr, PyrLayout L, int C, int H, int W,
                                                                  float rscale_div, int Ctot, int G, int nwg) {
    __shared__ float red[4][32][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h2 = lane >> 5;
    const int per_xcd = (nwg + 7) >> 3;
    const int wg = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (wg >= nwg) return;
    const int nt = (W + 31) / 32;
    const int t2 = wg % nt, t1 = (wg / nt) % nt, h = (wg / (nt * nt)) % H, b = wg / (nt * nt * H);
    const long HW = (long)H * W;
    const long chan0 = (long)(b / G) * Ctot + (long)(b % G) * C;
    const float* a_base = f1 + chan0 * HW + (long)h * W;
    const float* b_base = f2 + chan0 * HW + (long)h * W;
    const int w1_0 = t1 * 32, w2_0 = t2 * 32;
    const int w1 = w1_0 + l31, w2 = w2_0 + l31;
    const bool a_ok = w1 < W, b_ok = w2 < W;
    const int w1c = min(w1, W - 1), w2c = min(w2, W - 1);
    const int nk = (C + 1) / 2, per = (nk + 3) / 4;
    const int s_begin = min(wave * per, nk), s_end = min(s_begin + per, nk);
    const int nbatch = (s_end - s_begin + KB - 1) / KB;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float av[2][KB], bv[2][KB];
    auto load = [&](int bt, float* a, float* bb) {
#pragma unroll
        for (int i = 0; i < KB; ++i) {
            const int cc = min((s_begin + bt * KB + i) * 2 + h2, C - 1);
            a[i] = a_base[cc * HW + w1c];
            bb[i] = b_base[cc * HW + w2c];
        }
    };
    auto mma = [&](int bt, const float* a, const float* bb) {
#pragma unroll
        for (int i = 0; i < KB; ++i) {
            const int s = s_begin + bt * KB + i;
            const bool k_ok = s < s_end && s * 2 + h2 < C;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a_ok && k_ok) ? a[i] : 0.f, (b_ok && k_ok) ? bb[i] : 0.f, acc, 0, 0, 0);
        }
    };
    if (nbatch > 0) load(0, av[0], bv[0]);
    for (int bt = 0; bt < nbatch; bt += 2) {
        if (bt + 1 < nbatch) load(bt + 1, av[1], bv[1]);
        mma(bt, av[0], bv[0]);
        if (bt + 1 < nbatch) {
            if (bt + 2 < nbatch) load(bt + 2, av[0], bv[0]);
            mma(bt + 1, av[1], bv[1]);
        }
    }
    // partial tile of wave q -> red[q][row'][col], row' = row with bit 0 ^= bit 2: the two half waves of a store (rows r, r + 4)
    // land in different halves of the 64 banks
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h2;
        red[wave][row ^ ((row >> 2) & 1)][l31] = acc[reg];
    }
    __syncthreads();
    const int col = tid & 31, rg = tid >> 5;  // thread: column `col` of rows 4 rg .. 4 rg + 3; a half wave = one row group
    const int wcol = w2_0 + col;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = 4 * rg + j, rp = row ^ ((row >> 2) & 1);
        float v = ((red[0][rp][col] + red[1][rp][col]) + red[2][rp][col]) + red[3][rp][col];
        v = v / rscale_div;
        const int ww1 = w1_0 + row;
        const bool row_ok = ww1 < W;
        const long prow = ((long)b * H + h) * W + ww1;  // pyramid row of pixel (b,h,w1)
        if (row_ok && wcol < W) pyr[L.off[0] + prow * L.width[0] + wcol] = v;
        int wcur = wcol;
#pragma unroll
        for (int l = 1; l < MAX_LEVELS; ++l) {
            if (l >= L.nlev) break;
            const float other = __shfl_xor(v, 1 << (l - 1));
            v = (v + other) * 0.5f;
            wcur >>= 1;
            const bool owner = (col & ((1 << l) - 1)) == 0;
            if (row_ok && owner && wcur < L.width[l]) pyr[L.off[l] + prow * L.width[l] + wcur] = v;
        }
    }
}

// launches the k-split kernel when the problem is small enough that the LDS-staged kernel leaves the SIMDs with one wave
// each (fewer than 4 of its workgroups per CU) and the channel range is long enough to split; NND_CORR_BUILD_NO_KSPLIT: never
static bool corr1d_build_ksplit_launch(const float* f1, const float* f2, float* pyr, const PyrLayout& L, int C, int H, int W, int B,
                                       float div, int Ctot, int G, hipStream_t stream, int* rc) {
    const long nt = cdiv(W, 32), lds_wgs = nt * H * B * G, nwg = lds_wgs * nt;
    if (switches().corr_build_v1 || switches().corr_build_no_ksplit || C < 32 || lds_wgs >= 1024 || nwg > (1 << 24)) return false;
    const unsigned grid = (unsigned)(8 * cdiv((int)nwg, 8));
    // batches of 8 k-steps, two in flight (32 loads per lane): 96 VGPRs = 5 waves per SIMD, so the 4.25 workgroups per CU of one
    // pair at 68x120 are resident at once (batches of 16: 116 VGPRs, a fifth workgroup waits for a slot: 21.3 vs 19.7 us)
    hipLaunchKernelGGL(corr1d_build_ksplit_kernel<8>, dim3(grid), dim3(256), 0, stream, f1, f2, pyr, L, C, H, W, div, Ctot, G, (int)nwg);
    *rc = hipGetLastError() == hipSuccess ? NND_OK : NND_ERR_HIP;
    return true;
}

// launches corr1d_build_lds_kernel when the shape fits its staging plan; false: the caller uses corr1d_build_kernel
static bool corr1d_build_lds_launch(const float* f1, const float* f2, float* pyr, const PyrLayout& L, int C, int H, int W, int B,
                                    float div, int Ctot, int G, hipStream_t stream, int* rc) {
    if (corr1d_build_ksplit_launch(f1, f2, pyr, L, C, H, W, B, div, Ctot, G, stream, rc)) return true;
    const int ntile = cdiv(W, 32), WP = ntile * 32, NT = cdiv(ntile, 4);
    int KC = C >= 32 ? 32 : ((C + 1) / 2) * 2;
    if (C >= 64 && cdiv(64 * (8 + WP / 4), 256) <= CB_MAXLD) KC = 64;  // fewer, longer chunks: the next chunk's loads get more cover
    if (NT > 4 || cdiv(KC * (8 + WP / 4), 256) > CB_MAXLD || switches().corr_build_v1) return false;
    long outf = 0;
    for (int l = 0; l < L.nlev; ++l) outf += 32L * L.width[l];
    const size_t lds = sizeof(float) * (size_t)std::max<long>((long)KC * (32 + WP), outf);
    if (lds > 160 * 1024) return false;
    dim3 grid(cdiv(W, 32), H, B * G), block(256);
    {  // > 64 KB of dynamic LDS needs the opt-in, once per instantiation and device
        static std::atomic<unsigned> raised[4];
        const void* kerns[4] = {reinterpret_cast<const void*>(corr1d_build_lds_kernel<1>), reinterpret_cast<const void*>(corr1d_build_lds_kernel<2>),
                                reinterpret_cast<const void*>(corr1d_build_lds_kernel<3>), reinterpret_cast<const void*>(corr1d_build_lds_kernel<4>)};
        if (raise_lds_limit(kerns[NT - 1], raised[NT - 1]) != NND_OK) {
            *rc = NND_ERR_HIP;
            return true;
        }
    }
    if (NT == 1) hipLaunchKernelGGL(corr1d_build_lds_kernel<1>, grid, block, lds, stream, f1, f2, pyr, L, C, H, W, div, Ctot, G, KC);
    else if (NT == 2) hipLaunchKernelGGL(corr1d_build_lds_kernel<2>, grid, block, lds, stream, f1, f2, pyr, L, C, H, W, div, Ctot, G, KC);
    else if (NT == 3) hipLaunchKernelGGL(corr1d_build_lds_kernel<3>, grid, block, lds, stream, f1, f2, pyr, L, C, H, W, div, Ctot, G, KC);
    else hipLaunchKernelGGL(corr1d_build_lds_kernel<4>, grid, block, lds, stream, f1, f2, pyr, L, C, H, W, div, Ctot, G, KC);
    *rc = hipGetLastError() == hipSuccess ? NND_OK : NND_ERR_HIP;
    return true;
}

struct LookupArgs {
    PyrLayout L;
    int B, H, W, num_levels, radius;
    Lay lay;  // layout of coords (in) and of the sampled features (out): NCHW for the C-ABI, tile-major in the loop
};

__global__ void __launch_bounds__(256) corr1d_lookup_kernel(const float* __restrict__ pyr, const float* __restrict__ coords,
                                                            float* __restrict__ out, LookupArgs a) {
    const long HW = (long)a.H * a.W;
    const int ntap = 2 * a.radius + 1;
    const long total = (long)a.B * a.num_levels * ntap * HW;
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const long pix = idx % HW;
    const int ch = (int)((idx / HW) % (a.num_levels * ntap));
    const int b = (int)(idx / (HW * a.num_levels * ntap));
    const int lvl = ch / ntap, k = ch - lvl * ntap;
    const int w2 = a.L.width[lvl];
    const float* row = pyr + a.L.off[lvl] + ((long)b * HW + pix) * w2;
    const long po = pix_off(a.lay, (int)(pix / a.W), (int)(pix % a.W));
    float x = (float)(k - a.radius) + coords[(long)b * a.lay.plane + po] / (float)(1 << lvl);
    const float wm1 = (float)(w2 - 1);
    x = x / wm1;
    x = fminf(fmaxf(x, 0.f), 1.f);
    x = x * wm1;
    const float f0 = floorf(x), f1 = ceilf(x);
    const float v0 = row[(int)f0], v1 = row[(int)f1];
    const float coef = f1 - x;
    out[((long)b * (a.num_levels * ntap) + ch) * a.lay.plane + po] = coef * v0 + (1.0f - coef) * v1;
}

// level l = avg_pool1d(level l-1, 2) (floor on odd widths) for a pyramid whose level 0 came from elsewhere
// (IGEV's regularised volume, igev_stereo/cost_volume.py:46-52)
__global__ void __launch_bounds__(256) avg_pool_level_kernel(const float* __restrict__ src, float* __restrict__ dst, long rows,
                                                             int w_src, int w_dst) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * w_dst) return;
    const long row = idx / w_dst;
    const int j = (int)(idx - row * w_dst);
    const float* p = src + row * w_src + 2 * j;
    dst[idx] = (p[0] + p[1]) * 0.5f;
}

// IGEV combined lookup (igev_stereo/cost_volume.py:54-79): both volumes, G groups sharing the pixel's coordinate;
// out channel = i*(2*G*T) + v*(G*T) + g*T + k  (v = 0 feature volume, 1 geometry volume; T = 2r+1)
__global__ void __launch_bounds__(256) igev_lookup_kernel(const float* __restrict__ feat, const float* __restrict__ geo,
                                                          const float* __restrict__ coords, float* __restrict__ out, LookupArgs a,
                                                          int G) {
    const long HW = (long)a.H * a.W;
    const int ntap = 2 * a.radius + 1;
    const int nch = a.num_levels * 2 * G * ntap;
    const long total = (long)a.B * nch * HW;
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const long pix = idx % HW;
    const int ch = (int)((idx / HW) % nch);
    const int b = (int)(idx / (HW * nch));
    const int k = ch % ntap, g = (ch / ntap) % G, v = (ch / (ntap * G)) % 2, lvl = ch / (ntap * G * 2);
    const int w2 = a.L.width[lvl];
    const float* row = (v ? geo : feat) + a.L.off[lvl] + (((long)b * G + g) * HW + pix) * w2;
    const long po = pix_off(a.lay, (int)(pix / a.W), (int)(pix % a.W));
    float x = coords[(long)b * a.lay.plane + po] / (float)(1 << lvl) + (float)(k - a.radius);
    const float wm1 = (float)(w2 - 1);
    x = x / wm1;
    x = fminf(fmaxf(x, 0.f), 1.f);
    x = x * wm1;
    const float f0 = floorf(x), f1 = ceilf(x);
    const float v0 = row[(int)f0], v1 = row[(int)f1];
    const float coef = f1 - x;
    out[((long)b * nch + ch) * a.lay.plane + po] = coef * v0 + (1.0f - coef) * v1;
}

// out[b, c, r*h + i, r*w + j] = sum_k softmax_k(mask[b, k*r*r + i*r + j, h, w]) * (r * flow)[b, c, h+ky-1, w+kx-1]
// thread = (b, h, i, w); loops j (r consecutive outputs -> contiguous store) and c.
template <int RATE>
__global__ void __launch_bounds__(256) convex_upsample_kernel(const float* __restrict__ flow, const float* __restrict__ mask,
                                                              float* __restrict__ out, int B, int C, int H, int W, Lay lay) {
    const long total = (long)B * H * RATE * W;
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int w = (int)(idx % W);
    const int i = (int)((idx / W) % RATE);
    const int h = (int)((idx / ((long)W * RATE)) % H);
    const int b = (int)(idx / ((long)W * RATE * H));
    const long HW = lay.plane;
    const float* mb = mask + (long)b * 9 * RATE * RATE * HW + pix_off(lay, h, w);
    for (int c = 0; c < C; ++c) {
        float nb[9];
        const float* fb = flow + ((long)b * C + c) * HW;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            int yy = h + k / 3 - 1, xx = w + k % 3 - 1;
            nb[k] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? (float)RATE * fb[pix_off(lay, yy, xx)] : 0.f;
        }
        float res[RATE];
#pragma unroll
        for (int j = 0; j < RATE; ++j) {
            float m[9];
            float mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                m[k] = mb[(long)(k * RATE * RATE + i * RATE + j) * HW];
                mx = fmaxf(mx, m[k]);
            }
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                m[k] = expf(m[k] - mx);
                s += m[k];
            }
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) acc += (m[k] / s) * nb[k];
            res[j] = acc;
        }
        float* o = out + (((long)b * C + c) * (H * RATE) + (long)h * RATE + i) * ((long)W * RATE) + (long)w * RATE;
#pragma unroll
        for (int j = 0; j < RATE; ++j) o[j] = res[j];
    }
}

int corr1d_lookup_launch(const float* pyr, const float* coords, float* out, int B, int H, int W, int num_levels,
                         int radius, hipStream_t stream, bool tiled) {
    NND_REQUIRE(num_levels >= 1 && num_levels < MAX_LEVELS, "lookup: num_levels %d out of range", num_levels);
    LookupArgs a;
    make_layout(B, H, W, num_levels + 1, &a.L, nullptr);
    a.B = B; a.H = H; a.W = W; a.num_levels = num_levels; a.radius = radius;
    a.lay = make_lay(H, W, tiled);
    NND_REQUIRE(a.L.width[num_levels - 1] >= 2, "lookup: level %d has width %d < 2", num_levels - 1, a.L.width[num_levels - 1]);
    long total = (long)B * num_levels * (2 * radius + 1) * H * W;
    hipLaunchKernelGGL(corr1d_lookup_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, stream, pyr, coords, out, a);
    NND_LAUNCH_CHECK();
    return NND_OK;
}

// GroupCorrBlock1D.__call__ (raft_stereo/cost_volume.py:94-113; the correlation of Coarse2FineGroupRepViTRAFTStereo), quirk Q6
// included: per level the (B*G*H*W, 2r+1) samples — row (b,g,h,w) sampled at coords[b,h,w] / 2^i + dx — are VIEWED as
// (B, H, W, G*(2r+1)) without moving the group axis, so output pixel (y, x), channel j of level i holds
//   sample k = j % T of row r = (y*W + x)*G + j / T of batch b's G*H*W rows   (T = 2r+1; r -> (g', h', w') = (r / HW, r % HW / W, r % W)),
// i.e. the samples of G consecutive (group, pixel) rows, usually of another pixel and group.  Output channel = i*G*T + j.
// pyr: the group pyramid of nnd_group_corr_build(_scaled), rows ordered (b,g,h,w1).
__global__ void __launch_bounds__(256) group_lookup_flat_kernel(const float* __restrict__ pyr, const float* __restrict__ coords,
                                                                float* __restrict__ out, LookupArgs a, int G) {
    const long HW = (long)a.H * a.W;
    const int ntap = 2 * a.radius + 1, nch = a.num_levels * G * ntap;
    const long total = (long)a.B * nch * HW;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const long pix = idx % HW;
    const int ch = (int)((idx / HW) % nch);
    const int b = (int)(idx / (HW * nch));
    const int lvl = ch / (G * ntap), j = ch - lvl * (G * ntap);
    const long r = pix * G + j / ntap;  // row of the (g, h, w) enumeration whose samples land here
    const int k = j % ntap;
    const long spix = r % HW;           // its pixel: the coordinate it was sampled at
    const int w2 = a.L.width[lvl];
    const float* row = pyr + a.L.off[lvl] + ((long)b * G * HW + r) * w2;
    const long so = pix_off(a.lay, (int)(spix / a.W), (int)(spix % a.W)), po = pix_off(a.lay, (int)(pix / a.W), (int)(pix % a.W));
    float x = (float)(k - a.radius) + coords[(long)b * a.lay.plane + so] / (float)(1 << lvl);
    const float wm1 = (float)(w2 - 1);
    x = x / wm1;
    x = fminf(fmaxf(x, 0.f), 1.f);
    x = x * wm1;
    const float f0 = floorf(x), f1 = ceilf(x);
    const float v0 = row[(int)f0], v1 = row[(int)f1];
    const float coef = f1 - x;
    out[((long)b * nch + ch) * a.lay.plane + po] = coef * v0 + (1.0f - coef) * v1;
}

int group_lookup_flat_launch(const float* pyr, const float* coords, float* out, int B, int G, int H, int W, int num_levels, int radius,
                             hipStream_t stream, bool tiled) {
    NND_REQUIRE(num_levels >= 1 && num_levels < MAX_LEVELS && G >= 1, "group lookup: num_levels %d / groups %d out of range", num_levels, G);
    LookupArgs a;
    make_layout(B * G, H, W, num_levels + 1, &a.L, nullptr);
    a.B = B; a.H = H; a.W = W; a.num_levels = num_levels; a.radius = radius;
    a.lay = make_lay(H, W, tiled);
    NND_REQUIRE(a.L.width[num_levels - 1] >= 2, "group lookup: level %d has width %d < 2", num_levels - 1, a.L.width[num_levels - 1]);
    const long total = (long)B * num_levels * G * (2 * radius + 1) * H * W;
    hipLaunchKernelGGL(group_lookup_flat_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, stream, pyr, coords, out, a, G);
    NND_LAUNCH_CHECK();
    return NND_OK;
}

// relu(acc + bias) of one lane's pixel and 16 channels into c1: planar tile-major, or (out_c4) the 4-channel-interleaved
// tile-major layout of layout.h, where registers 4q..4q+3 are one 16-B store
// the lane's 16 bias values (accumulator register order), loaded by the callers BEFORE their MFMA loops: fetched inside the
// store they sat on the kernel's critical tail (the flow-branch kernel lost 5 us that way)
__device__ __forceinline__ void load_c1_bias(const float* __restrict__ bias, int cb, int h2, float (&br)[16]) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) br[reg] = bias[cb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h2];
}
__device__ __forceinline__ void store_c1(float* __restrict__ out_b, const Lay& lay, int y, int x, int cb, int h2, const f32x16& acc,
                                         const float (&br)[16], int out_c4) {
    const long po = pix_off(lay, y, x);  // lay: the planar tile-major layout of coords (ci = 1)
    if (out_c4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int co0 = cb * 32 + 8 * q + 4 * h2;
            *reinterpret_cast<float4*>(out_b + (long)co0 * lay.plane + po * 4) =
                make_float4(fmaxf(acc[4 * q] + br[4 * q], 0.f), fmaxf(acc[4 * q + 1] + br[4 * q + 1], 0.f),
                            fmaxf(acc[4 * q + 2] + br[4 * q + 2], 0.f), fmaxf(acc[4 * q + 3] + br[4 * q + 3], 0.f));
        }
    } else {
        float* o = out_b + po;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int co = cb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h2;
            o[(long)co * lay.plane] = fmaxf(acc[reg] + br[reg], 0.f);
        }
    }
}

// Fused lookup + encoder.convc1 (1x1, cor_planes -> 256, ReLU) for the refinement loops: one workgroup = one 4x8 pixel
// tile x all 256 output channels (8 waves).  The sampled correlation features are produced 32 channels at a time
// straight into the LDS B-operand (same arithmetic and op order as corr1d_lookup_kernel / igev_lookup_kernel; the two
// gathers of chunk c+1 are in flight while the MFMAs of chunk c issue), and every wave multiplies them with the packed
// convc1 weights (conv_mfma fragment order, CI_T = 32).  Saves a launch and the (B,cor_planes,H,W) round trip — 75 MB
// per iteration for IGEV's 576 planes.  The accumulation order is conv_mfma's with ks = 1.
// IGEV = true: two pyramids, channel = lvl*(2*G*T) + v*(G*T) + g*T + k (igev_stereo/cost_volume.py:54-79).
// (the workgroup's work as a device function of its (tile, batch) index: lookup_convc1_kernel launches it alone,
//  flow_branch_lookup_kernel beside the flow-branch workgroups of the same iteration)
struct LookupC1Args {
    const float* pyr;
    const float* geo;
    const float* coords;
    const float* wpk;
    const float* bias;
    float* out;
    long obs;
    LookupArgs a;
    int G, tiles_x, cb_stride, out_c4;
};

template <bool IGEV>
__device__ __forceinline__ void lookup_convc1_body(const LookupC1Args& q, float (*xs)[32 * 32], const int bx, const int b) {
    const float* __restrict__ pyr = q.pyr;
    const float* __restrict__ geo = q.geo;
    const float* __restrict__ coords = q.coords;
    const float* __restrict__ wpk = q.wpk;
    const float* __restrict__ bias = q.bias;
    float* __restrict__ out = q.out;
    const long obs = q.obs;
    const LookupArgs& a = q.a;
    const int G = q.G, tiles_x = q.tiles_x, cb_stride = q.cb_stride, out_c4 = q.out_c4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h2 = lane >> 5, l31 = lane & 31;
    const int tx0 = (bx % tiles_x) * 8, ty0 = (bx / tiles_x) * 4;
    const int ntap = 2 * a.radius + 1;
    const int nch = a.num_levels * ntap * (IGEV ? 2 * G : 1), nchunks = (nch + 31) / 32;
    const long HW = (long)a.H * a.W;
    // staging role: pixel px of the tile, channels cs and cs + 16 of every chunk
    const int px = tid & 31, cs = tid >> 5;
    const int py = ty0 + (px >> 3), pxx = tx0 + (px & 7);
    const bool pin = py < a.H && pxx < a.W;
    const long pix = pin ? (long)py * a.W + pxx : 0;
    const float cval = pin ? coords[(long)b * a.lay.plane + pix_off(a.lay, py, pxx)] : 0.f;
    float v0[2], v1[2], cf[2];
    auto fetch = [&](int chunk) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ch = chunk * 32 + cs + 16 * j;
            const bool ok = pin && ch < nch;
            const int chc = ok ? ch : 0;
            int k, lvl;
            const float* base;
            if (IGEV) {
                k = chc % ntap;
                const int g = (chc / ntap) % G, v = (chc / (ntap * G)) % 2;
                lvl = chc / (ntap * G * 2);
                base = (v ? geo : pyr) + a.L.off[lvl] + (((long)b * G + g) * HW + pix) * a.L.width[lvl];
            } else {
                lvl = chc / ntap;
                k = chc - lvl * ntap;
                base = pyr + a.L.off[lvl] + ((long)b * HW + pix) * a.L.width[lvl];
            }
            float x = cval / (float)(1 << lvl) + (float)(k - a.radius);
            const float wm1 = (float)(a.L.width[lvl] - 1);
            x = x / wm1;
            x = fminf(fmaxf(x, 0.f), 1.f);
            x = x * wm1;
            const float f0 = floorf(x), f1 = ceilf(x);
            v0[j] = base[(int)f0];
            v1[j] = base[(int)f1];
            cf[j] = ok ? f1 - x : 2.0f;  // 2.0 marks "no such channel / pixel": the store below writes 0
        }
    };
    auto put = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
            xs[buf][(cs + 16 * j) * 32 + px] = cf[j] == 2.0f ? 0.f : cf[j] * v0[j] + (1.0f - cf[j]) * v1[j];
    };
    // what does not depend on the coordinates is requested first: this wave's bias and the weight fragments of chunk 0 (the packed
    // layer's K is padded to whole 32-channel chunks, so all 4 fragments of a chunk exist)
    const int cb = wave;  // 8 waves = 256 output channels
    // cb_stride: float4s per output-channel block of the packed layer (its K is padded to whole CI_T chunks)
    const float4* wb = reinterpret_cast<const float4*>(wpk) + (size_t)cb * cb_stride;
    float br[16];
    load_c1_bias(bias, cb, h2, br);
    float4 aw[4], an[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) aw[q] = wb[q * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);
    fetch(0);
    put(0);
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const bool more = chunk + 1 < nchunks;
        if (more) {
#pragma unroll
            for (int q = 0; q < 4; ++q) an[q] = wb[(size_t)(chunk + 1) * (4 * 64) + q * 64 + lane];
            fetch(chunk + 1);
        }
        const int npair = min(16, (nch - chunk * 32 + 1) / 2);  // k-pairs that hold real channels
        const float* xb = xs[chunk & 1] + h2 * 32 + l31;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float as[4] = {aw[q].x, aw[q].y, aw[q].z, aw[q].w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (q * 4 + j < npair) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(as[j], xb[(q * 4 + j) * 64], acc, 0, 0, 0);
        }
        if (more) {
            put((chunk + 1) & 1);
#pragma unroll
            for (int q = 0; q < 4; ++q) aw[q] = an[q];
        }
        __syncthreads();
    }
    const int y = ty0 + (l31 >> 3), x = tx0 + (l31 & 7);
    if (y >= a.H || x >= a.W) return;
    store_c1(out + (long)b * obs, a.lay, y, x, cb, h2, acc, br, out_c4);
}

template <bool IGEV>
__global__ void __launch_bounds__(512) lookup_convc1_kernel(LookupC1Args q) {
    __shared__ float xs[2][32 * 32];  // [buffer][channel of the chunk][pixel of the tile]
    lookup_convc1_body<IGEV>(q, xs, (int)blockIdx.x, (int)blockIdx.z);
}

// Round 3: the motion encoder's flow branch and lookup + convc1 of one iteration as ONE launch of two kinds of workgroups.  The
// two are independent (both read only the state the previous iteration left; convc2 / conv need both) and each alone is a
// latency chain on one workgroup per CU (12.4 and 7.6 us at 68x120); both are 512-thread programs of 120 / 97 VGPRs whose LDS
// (69 KB with fp16x2 pieces + 8 KB, allocated per workgroup: 2 x 69 KB) fits a CU twice, so a flow-branch and a lookup workgroup
// run side by side on every CU.  Workgroups [0, nfb) of a batch item are flow-branch tiles, [nfb, 2 nfb) lookup tiles.  Same code
// paths as the two kernels: bit-identical (tests/test_gpu_split.py).  RAFT-Stereo pyramids, arithmetic 2 (with 3 bf16 pieces
// the flow branch's patch makes 2 x 82 KB, more than a CU has: the two launches stay).
template <int FC, int NS>
__global__ void __launch_bounds__(512) flow_branch_lookup_kernel(FlowBranchArgs fa, LookupC1Args q) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fbl_lds[];
    const int nfb = fa.c.npos;
    if ((int)blockIdx.x < nfb) flow_branch_body<FC, NS>(fa, fbl_lds, (int)blockIdx.x, (int)blockIdx.z);
    else lookup_convc1_body<false>(q, reinterpret_cast<float(*)[32 * 32]>(fbl_lds), (int)blockIdx.x - nfb, (int)blockIdx.z);
}

// coords (tile-major) -> c1 = relu(convc1(lookup(coords))) (tile-major, 256 channels); wpk / bias: the packed convc1 layer.
// geo == nullptr: RAFT-Stereo pyramid; otherwise the IGEV feature + geometry pyramids with G groups.
static int lookup_convc1_launch_impl(const float* pyr, const float* geo, int G, const float* coords, const ConvLayer& L, const float* blob,
                                    float* c1, int64_t c1_bs, int B, int H, int W, int num_levels, int radius, hipStream_t stream,
                                    bool c1_c4, const FlowBranchArgs* fb, int fb_fc, int fb_ns) {
    NND_REQUIRE(L.KH == 1 && L.KW == 1 && L.Cout == 256 && L.CI_T % 32 == 0 && L.arith == 0 &&
                    L.Cin == num_levels * (2 * radius + 1) * (geo ? 2 * G : 1),
                "lookup_convc1: layer %dx%d %d->%d (arithmetic %d) does not match the lookup", L.KH, L.KW, L.Cin, L.Cout, L.arith);
    const float* wpk = blob + L.w_off;
    const float* bias = blob + L.b_off;
    const int cb_stride = L.nchunks * (L.CI_T / 8) * 64;
    NND_REQUIRE(num_levels >= 1 && num_levels < MAX_LEVELS, "lookup_convc1: num_levels %d out of range", num_levels);
    LookupArgs a;
    make_layout(geo ? B * G : B, H, W, num_levels + 1, &a.L, nullptr);
    a.B = B; a.H = H; a.W = W; a.num_levels = num_levels; a.radius = radius;
    a.lay = make_lay(H, W, true);
    NND_REQUIRE(a.L.width[num_levels - 1] >= 2, "lookup: level %d has width %d < 2", num_levels - 1, a.L.width[num_levels - 1]);
    const int tiles_x = cdiv(W, 8);
    dim3 grid(tiles_x * cdiv(H, 4), 1, B), block(512);
    LookupC1Args q{pyr, geo, coords, wpk, bias, c1, (long)c1_bs, a, geo ? G : 1, tiles_x, cb_stride, c1_c4 ? 1 : 0};
    if (fb) {  // the flow-branch workgroups of the same iteration in the same launch
        NND_REQUIRE(!geo && flow_branch_lookup_supported(fb_ns), "lookup_convc1: the merged flow-branch launch is built for RAFT-Stereo pyramids and arithmetic 2");
        NND_REQUIRE(fb->c.npos == (int)grid.x && fb->c.tiles_x == tiles_x, "lookup_convc1: flow-branch tiling differs");
        grid.x *= 2;
        const size_t lds = fb_fc == 1 ? fb_lds_bytes<1, 2>() : fb_lds_bytes<2, 2>();
        if (fb_fc == 1) {
            static std::atomic<unsigned> raised{0};
            if (int rc = raise_lds_limit(reinterpret_cast<const void*>(flow_branch_lookup_kernel<1, 2>), raised)) return rc;
            hipLaunchKernelGGL((flow_branch_lookup_kernel<1, 2>), grid, block, lds, stream, *fb, q);
        } else {
            static std::atomic<unsigned> raised{0};
            if (int rc = raise_lds_limit(reinterpret_cast<const void*>(flow_branch_lookup_kernel<2, 2>), raised)) return rc;
            hipLaunchKernelGGL((flow_branch_lookup_kernel<2, 2>), grid, block, lds, stream, *fb, q);
        }
        NND_LAUNCH_CHECK();
        return NND_OK;
    }
    if (geo) hipLaunchKernelGGL(lookup_convc1_kernel<true>, grid, block, 0, stream, q);
    else hipLaunchKernelGGL(lookup_convc1_kernel<false>, grid, block, 0, stream, q);
    NND_LAUNCH_CHECK();
    return NND_OK;
}

bool flow_branch_lookup_supported(int arith) { return arith == 2; }

int lookup_convc1_launch(const float* pyr, const float* geo, int G, const float* coords, const ConvLayer& L, const float* blob,
                         float* c1, int64_t c1_bs, int B, int H, int W, int num_levels, int radius, hipStream_t stream, bool c1_c4) {
    return lookup_convc1_launch_impl(pyr, geo, G, coords, L, blob, c1, c1_bs, B, H, W, num_levels, radius, stream, c1_c4, nullptr, 0, 0);
}

int flow_branch_lookup_launch(const ConvLayer& f2, const float* blob, const float* w7t, const float* b7, const float* flow, int64_t fbs,
                              int fc, const ConvIO& io, const float* pyr, const float* coords, const ConvLayer& Lc1, float* c1,
                              int64_t c1_bs, int B, int H, int W, int num_levels, int radius, hipStream_t stream, bool c1_c4) {
    FlowBranchArgs fa;
    int rc = make_flow_branch_args(f2, blob, w7t, b7, flow, fbs, fc, io, B, H, W, &fa);
    if (rc != NND_OK) return rc;
    return lookup_convc1_launch_impl(pyr, nullptr, 1, coords, Lc1, blob, c1, c1_bs, B, H, W, num_levels, radius, stream, c1_c4, &fa, fc,
                                     f2.arith);
}

// ---------------------------------------------------------------------------------------------------------------
// IGEV, group-interleaved pyramids.  In the reference layout (rows (b, g, h, w1) of w2 floats) the 2*G*(2r+1) samples a
// pixel takes at one level sit in 2*G different rows, each touched for ~1.4 of its 128-B lines: ~90 lines per pixel and
// iteration for 4 levels, 375 MB per sample at 136x240.  The interleaved copy keeps, per level,
//     il[((b*H*W + pix) * w2 + x) * 2G + v*G + g]            v = 0 feature volume / 1 geometry volume
// so the same samples are one contiguous run of (2r+2)*2G floats: ~22 lines per pixel and iteration.  It is written
// once per pair (igev_interleave_kernel: a 2G x 64 LDS transpose per block) and read by every iteration of the loop.
struct ILayout {
    long off[MAX_LEVELS];
    int width[MAX_LEVELS];
};

static void make_il_layout(int B, int G, int H, int W, int num_levels, ILayout* L, int64_t* total) {
    long off = 0;
    int w = W;
    for (int l = 0; l < MAX_LEVELS; ++l) {
        L->off[l] = off;
        L->width[l] = l < num_levels ? w : 0;
        if (l < num_levels) off += (long)B * H * W * w * 2 * G;
        w /= 2;
    }
    if (total) *total = off;
}

// grid (B*H*W, ceil(w2/64)); block 256.  src rows: feat / geo [(b*G + g)*HW + pix][w2]; dst: see above.
__global__ void __launch_bounds__(256) igev_interleave_kernel(const float* __restrict__ feat, const float* __restrict__ geo,
                                                              float* __restrict__ dst, int G, long HW, int w2) {
    __shared__ float sm[32 * 68];
    const int VG = 2 * G;  // <= 32
    const long bp = blockIdx.x, b = bp / HW, pix = bp - b * HW;
    const int x0 = blockIdx.y * 64, nx = min(64, w2 - x0);
    for (int i = threadIdx.x; i < VG * 64; i += 256) {
        const int vg = i >> 6, x = i & 63;
        if (x < nx) {
            const int v = vg / G, g = vg - v * G;
            sm[vg * 68 + x] = (v ? geo : feat)[((b * G + g) * HW + pix) * w2 + x0 + x];
        }
    }
    __syncthreads();
    float* o = dst + (bp * w2 + x0) * VG;
    for (int i = threadIdx.x; i < VG * nx; i += 256) o[i] = sm[(i % VG) * 68 + i / VG];
}

// The same interleaved levels straight from the two level-0 volumes: one workgroup = one pixel's 2G rows of w2 floats, read once
// (16-byte loads), avg-pooled level by level in LDS with the pyramid's own arithmetic ((a + b) * 0.5f, floor on odd widths:
// bit-identical to pooling first and interleaving after), every level written as one contiguous run of w_l * 2G floats while the
// next one is pooled.  Replaces pooling the geometry pyramid (levels 0-3 read, 1-4 written), the pooled levels of the feature
// pyramid and the re-read of levels 1-3 by the kernel above: 1.44 GB instead of 2.8 GB per sample at 136x240, G = 8.
// LDS rows of level l have stride il_pool_stride(w_l): a multiple of 4 that is 4 or 12 mod 16, so the 4 rows a lane gathers
// for one 16-byte store and the 16 columns of a wave fall into distinct banks.
__host__ __device__ static inline int il_pool_stride(int w) {
    int s = (w + 3) & ~3;
    if ((s & 7) == 0) s += 4;
    return s;
}
template <bool V4>
__global__ void __launch_bounds__(256) igev_pool_interleave_kernel(const float* __restrict__ feat0, const float* __restrict__ geo0,
                                                                   float* __restrict__ dst, ILayout IL, int G, long HW, int w2,
                                                                   int nlev) {
    extern __shared__ float sm[];
    const int VG = 2 * G, tid = threadIdx.x;
    const long bp = blockIdx.x, b = bp / HW, pix = bp - b * HW;
    const int S0 = il_pool_stride(w2);
    if (V4) {
        // IL_LD loads in flight per thread before their LDS stores; (row, quad) of unit tid + 256 k advance without divisions
        constexpr int IL_LD = 4;
        const int q4 = w2 >> 2, total = VG * q4, qs = 256 / q4, ms = 256 - qs * q4;
        int vg = tid / q4, x4 = tid - vg * q4;
        for (int base = tid; base < total; base += 256 * IL_LD) {
            float4 val[IL_LD];
            int dsti[IL_LD];
#pragma unroll
            for (int k = 0; k < IL_LD; ++k) {
                const bool in = base + 256 * k < total;
                const int v = vg >= G, g = vg - v * G;
                dsti[k] = in ? vg * S0 + 4 * x4 : -1;
                val[k] = in ? *reinterpret_cast<const float4*>((v ? geo0 : feat0) + ((b * G + g) * HW + pix) * w2 + 4 * x4)
                            : make_float4(0.f, 0.f, 0.f, 0.f);
                x4 += ms;
                vg += qs;
                if (x4 >= q4) x4 -= q4, ++vg;
            }
#pragma unroll
            for (int k = 0; k < IL_LD; ++k)
                if (dsti[k] >= 0) *reinterpret_cast<float4*>(sm + dsti[k]) = val[k];
        }
    } else {
        for (int i = tid; i < VG * w2; i += 256) {
            const int vg = i / w2, x = i - vg * w2, v = vg / G, g = vg - v * G;
            sm[vg * S0 + x] = ((v ? geo0 : feat0) + ((b * G + g) * HW + pix) * w2)[x];
        }
    }
    __syncthreads();
    int lo = 0, wl = w2, Sl = S0;
    for (int l = 0; l < nlev; ++l) {
        const float* cur = sm + lo;
        float* o = dst + IL.off[l] + bp * wl * VG;
        if (V4) {
            const int vq = VG >> 2;
            for (int u = tid; u < wl * vq; u += 256) {
                const int x = u / vq, q = u - x * vq;
                const float* c = cur + 4 * q * Sl + x;
                *reinterpret_cast<float4*>(o + 4 * u) = make_float4(c[0], c[Sl], c[2 * Sl], c[3 * Sl]);
            }
        } else {
            for (int i = tid; i < wl * VG; i += 256) o[i] = cur[(i % VG) * Sl + i / VG];
        }
        if (l + 1 == nlev) break;
        const int wn = wl >> 1, Sn = il_pool_stride(wn);
        float* nxt = sm + lo + VG * Sl;
        for (int i = tid; i < VG * wn; i += 256) {
            const int vg = i / wn, j = i - vg * wn;
            const float2 p = *reinterpret_cast<const float2*>(cur + vg * Sl + 2 * j);
            nxt[vg * Sn + j] = (p.x + p.y) * 0.5f;
        }
        __syncthreads();
        lo += VG * Sl;
        wl = wn;
        Sl = Sn;
    }
}

// lookup + convc1 over the interleaved pyramids (G = 8, radius 4: VG = 16 runs of NTAP = 9 taps, 144 planes per level).
// One workgroup = two 4x8 pixel sub-tiles (64 pixels), 8 waves = 8 x 32 output channels; every weight fragment read
// from L2 feeds two MFMAs.  K is walked level by level: while the MFMAs of level l run out of one LDS buffer, the gathers
// of level l+1 are in flight (thread = (pixel, v*G+g): its 9 taps are consecutive 2G-float steps of one contiguous run;
// the 16 lanes of a pixel read 64 contiguous bytes), then blended into the other buffer.  The weights are the packed
// convc1 layer (conv_mfma fragment order; one float4 per lane and group of 4 k-pairs, prefetched one group ahead).
// Same arithmetic and K order as lookup_convc1_kernel<true>: bit-identical output.
constexpr int IL_VG = 16, IL_NTAP = 9, IL_LC = IL_VG * IL_NTAP, IL_S = 68;  // IL_S: LDS row stride (64 pixels + 4: conflict-free)
__global__ void __launch_bounds__(512, 2) igev_lookup_convc1_il_kernel(const float* __restrict__ il, const float* __restrict__ coords,
                                                                      const float* __restrict__ wpk, const float* __restrict__ bias,
                                                                      float* __restrict__ out, long obs, ILayout IL, Lay lay,
                                                                      int H, int W, int num_levels, int ntiles, int tiles_x,
                                                                      int cb_stride, int out_c4) {
    extern __shared__ float xs[];  // [2][IL_LC][IL_S]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h2 = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.z;
    const long HW = (long)H * W;
    float br[16];  // this wave's bias, requested before anything that depends on the coordinates
    load_c1_bias(bias, wave, h2, br);
    // gather role: items tid and tid + 512 of the 64 pixels x 16 runs
    const int vg = tid & 15;
    long pixo[2];   // float offset of the pixel's run at level 0 divided by w2*VG, i.e. the row index b*HW + pix
    float cval[2];
    bool pin[2];
    int pcol[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int px64 = (tid >> 4) + 32 * i, t = blockIdx.x * 2 + (px64 >> 5), px = px64 & 31;
        const int py = (t / tiles_x) * 4 + (px >> 3), pxx = (t % tiles_x) * 8 + (px & 7);
        pin[i] = t < ntiles && py < H && pxx < W;
        const long pix = pin[i] ? (long)py * W + pxx : 0;
        pixo[i] = (long)b * HW + pix;
        cval[i] = pin[i] ? coords[(long)b * lay.plane + pix_off(lay, py, pxx)] : 0.f;
        pcol[i] = px64;
    }
    float v0[2][IL_NTAP], v1[2][IL_NTAP];
    auto fetch = [&](int lvl) {
        const int w2 = IL.width[lvl];
        const float wm1 = (float)(w2 - 1);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float* base = il + IL.off[lvl] + pixo[i] * w2 * IL_VG + vg;
#pragma unroll
            for (int k = 0; k < IL_NTAP; ++k) {
                float x = cval[i] / (float)(1 << lvl) + (float)(k - IL_NTAP / 2);
                x = x / wm1;
                x = fminf(fmaxf(x, 0.f), 1.f);
                x = x * wm1;
                v0[i][k] = base[(int)floorf(x) * IL_VG];
                v1[i][k] = base[(int)ceilf(x) * IL_VG];
            }
        }
    };
    auto put = [&](int lvl, float* buf) {
        const float wm1 = (float)(IL.width[lvl] - 1);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int k = 0; k < IL_NTAP; ++k) {
                float x = cval[i] / (float)(1 << lvl) + (float)(k - IL_NTAP / 2);
                x = x / wm1;
                x = fminf(fmaxf(x, 0.f), 1.f);
                x = x * wm1;
                const float cf = ceilf(x) - x;
                buf[(vg * IL_NTAP + k) * IL_S + pcol[i]] = pin[i] ? cf * v0[i][k] + (1.0f - cf) * v1[i][k] : 0.f;
            }
    };
    fetch(0);
    put(0, xs);
    __syncthreads();
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc0[i] = 0.f, acc1[i] = 0.f;
    const float4* wq = reinterpret_cast<const float4*>(wpk) + (size_t)wave * cb_stride + lane;  // + qq*64: group qq of 4 k-pairs
    constexpr int NQ = IL_LC / 8;                                                              // groups per level
    float4 av = wq[0];
    for (int lvl = 0; lvl < num_levels; ++lvl) {
        const bool more = lvl + 1 < num_levels;
        if (more) fetch(lvl + 1);
        const float* xb = xs + (lvl & 1) * (IL_LC * IL_S) + h2 * IL_S + l31;
#pragma unroll 2
        for (int q = 0; q < NQ; ++q) {
            const int qq = lvl * NQ + q;
            const float4 an = wq[(size_t)min(qq + 1, num_levels * NQ - 1) * 64];
            const float as[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float* xr = xb + (q * 4 + j) * 2 * IL_S;
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(as[j], xr[0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(as[j], xr[32], acc1, 0, 0, 0);
            }
            av = an;
        }
        if (more) put(lvl + 1, xs + ((lvl + 1) & 1) * (IL_LC * IL_S));
        __syncthreads();
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int t = blockIdx.x * 2 + p;
        const int y = (t / tiles_x) * 4 + (l31 >> 3), x = (t % tiles_x) * 8 + (l31 & 7);
        if (t >= ntiles || y >= H || x >= W) continue;
        store_c1(out + (long)b * obs, lay, y, x, wave, h2, p ? acc1 : acc0, br, out_c4);
    }
}

// The same with convc1 in a split arithmetic (round 3; NS = 2: fp16x2, NS = 3: bf16x3, split_arith.h): convc1 of IGEV is a
// 576 -> 256 GEMM (9.6 GFLOP per iteration at 136x240 — the largest of the IGEV iteration, 61 us at the fp32-MFMA peak); on the
// 16-bit MFMA with split operands it is a fifth of that and the kernel is left with its gathers.  The blended samples are
// split while they are written to LDS, in the B-fragment order of v_mfma_f32_32x32x16_* ([16-channel chunk][piece][k half]
// [pixel][8 channels]: one conflict-free ds_read_b128 per piece and sub-tile); the weights are the layer in pack_conv_split's
// order ([cb][chunk][piece][lane]), two chunks ahead in registers.  K order: level by level, chunk by chunk, products with
// i + j descending — conv_split's order for ks = 1.
template <int NS>
__global__ void __launch_bounds__(512, 2) igev_lookup_convc1_il_split_kernel(const float* __restrict__ il, const float* __restrict__ coords,
                                                                            const float* __restrict__ wpk, const float* __restrict__ bias,
                                                                            float* __restrict__ out, long obs, ILayout IL, Lay lay,
                                                                            int H, int W, int num_levels, int ntiles, int tiles_x,
                                                                            int out_c4) {
    constexpr int NCHL = IL_LC / 16;                     // 16-channel chunks per level (9)
    constexpr int LVB = NCHL * NS * 2 * 64 * 16;         // bytes of one level's B image: [chunk][piece][k half][64 pixels][8 x 16 bit]
    extern __shared__ __attribute__((aligned(16))) unsigned char xsb[];  // [2][LVB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h2 = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.z;
    const long HW = (long)H * W;
    float br[16];
    load_c1_bias(bias, wave, h2, br);
    float oscale = 1.f, xscale = 1.f;
    if constexpr (NS == 2) {  // behind the 8 x 32 bias values of the packed layer (split_arith.h)
        oscale = bias[256 + SPLIT_TAIL_OSCALE];
        xscale = bias[256 + SPLIT_TAIL_XSCALE];
    }
    const int vg = tid & 15;
    long pixo[2];
    float cval[2];
    bool pin[2];
    int pcol[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int px64 = (tid >> 4) + 32 * i, t = blockIdx.x * 2 + (px64 >> 5), px = px64 & 31;
        const int py = (t / tiles_x) * 4 + (px >> 3), pxx = (t % tiles_x) * 8 + (px & 7);
        pin[i] = t < ntiles && py < H && pxx < W;
        const long pix = pin[i] ? (long)py * W + pxx : 0;
        pixo[i] = (long)b * HW + pix;
        cval[i] = pin[i] ? coords[(long)b * lay.plane + pix_off(lay, py, pxx)] : 0.f;
        pcol[i] = px64;
    }
    float v0[2][IL_NTAP], v1[2][IL_NTAP];
    auto fetch = [&](int lvl) {
        const int w2 = IL.width[lvl];
        const float wm1 = (float)(w2 - 1);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float* base = il + IL.off[lvl] + pixo[i] * w2 * IL_VG + vg;
#pragma unroll
            for (int k = 0; k < IL_NTAP; ++k) {
                float x = cval[i] / (float)(1 << lvl) + (float)(k - IL_NTAP / 2);
                x = x / wm1;
                x = fminf(fmaxf(x, 0.f), 1.f);
                x = x * wm1;
                v0[i][k] = base[(int)floorf(x) * IL_VG];
                v1[i][k] = base[(int)ceilf(x) * IL_VG];
            }
        }
    };
    auto put = [&](int lvl, unsigned char* buf) {
        const float wm1 = (float)(IL.width[lvl] - 1);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int k = 0; k < IL_NTAP; ++k) {
                float x = cval[i] / (float)(1 << lvl) + (float)(k - IL_NTAP / 2);
                x = x / wm1;
                x = fminf(fmaxf(x, 0.f), 1.f);
                x = x * wm1;
                const float cf = ceilf(x) - x;
                float res = pin[i] ? cf * v0[i][k] + (1.0f - cf) * v1[i][k] : 0.f;  // the sample, in the reference's op order
                if constexpr (NS == 2) res = res * xscale;
                const int ch = vg * IL_NTAP + k;                              // channel inside the level
                unsigned short* dst = reinterpret_cast<unsigned short*>(buf + (((ch >> 4) * NS * 2 + ((ch >> 3) & 1)) * 64 + pcol[i]) * 16) + (ch & 7);
#pragma unroll
                for (int sp = 0; sp < NS; ++sp) {  // pieces: round-to-nearest of the running residual (split_arith.h)
                    unsigned short bits;
                    if constexpr (NS == 3) {
                        const __bf16 pv = (__bf16)res;
                        res -= (float)pv;
                        bits = __builtin_bit_cast(unsigned short, pv);
                    } else {
                        const _Float16 pv = (_Float16)res;
                        res -= (float)pv;
                        bits = __builtin_bit_cast(unsigned short, pv);
                    }
                    dst[sp * (2 * 64 * 8)] = bits;  // next piece: 2 k halves x 64 pixels x 8 values further
                }
            }
    };
    fetch(0);
    put(0, xsb);
    __syncthreads();
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc0[i] = 0.f, acc1[i] = 0.f;
    // weights: uint4 index ((cb * nchunks + chunk) * NS + piece) * 64 + lane, a ring of 3 chunks
    const int nchunks = num_levels * NCHL;
    const uint4* wq = reinterpret_cast<const uint4*>(wpk) + (size_t)wave * nchunks * NS * 64 + lane;
    auto load_a = [&](uint4 (&dst)[NS], int ch) {
        const uint4* w = wq + (size_t)min(ch, nchunks - 1) * NS * 64;
#pragma unroll
        for (int sp = 0; sp < NS; ++sp) dst[sp] = w[sp * 64];
    };
    uint4 ab[3][NS];
    load_a(ab[0], 0);
    load_a(ab[1], 1);
    for (int lvl = 0; lvl < num_levels; ++lvl) {
        const bool more = lvl + 1 < num_levels;
        if (more) fetch(lvl + 1);
        const unsigned char* xb = xsb + (lvl & 1) * LVB + (h2 * 64 + l31) * 16;
#pragma unroll
        for (int q = 0; q < NCHL; ++q) {  // NCHL = 9 = 3 x 3: the ring phase (lvl * NCHL + q) % 3 == q % 3 is a compile-time constant
            load_a(ab[(q + 2) % 3], lvl * NCHL + q + 2);
            uint4 b0[NS], b1[NS];
#pragma unroll
            for (int sp = 0; sp < NS; ++sp) {
                b0[sp] = *reinterpret_cast<const uint4*>(xb + ((q * NS + sp) * 2 * 64) * 16);
                b1[sp] = *reinterpret_cast<const uint4*>(xb + ((q * NS + sp) * 2 * 64 + 32) * 16);
            }
            __builtin_amdgcn_sched_barrier(0);
            split_mfma_step<NS>(ab[q % 3], b0, acc0);
            split_mfma_step<NS>(ab[q % 3], b1, acc1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) put(lvl + 1, xsb + ((lvl + 1) & 1) * LVB);
        __syncthreads();
    }
    if constexpr (NS == 2) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc0[i] *= oscale, acc1[i] *= oscale;
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int t = blockIdx.x * 2 + p;
        const int y = (t / tiles_x) * 4 + (l31 >> 3), x = (t % tiles_x) * 8 + (l31 & 7);
        if (t >= ntiles || y >= H || x >= W) continue;
        store_c1(out + (long)b * obs, lay, y, x, wave, h2, p ? acc1 : acc0, br, out_c4);
    }
}

// false: shape not covered by the kernel above (the caller falls back to lookup_convc1_launch on the reference layout)
bool igev_lookup_convc1_il_supported(int G, int num_levels, int radius) {
    return 2 * G == IL_VG && 2 * radius + 1 == IL_NTAP && num_levels >= 1 && num_levels < MAX_LEVELS;
}

int igev_lookup_convc1_il_launch(const float* il, int G, const float* coords, const ConvLayer& L, const float* blob, float* c1,
                                 int64_t c1_bs, int B, int H, int W, int num_levels, int radius, hipStream_t stream, bool c1_c4) {
    NND_REQUIRE(igev_lookup_convc1_il_supported(G, num_levels, radius), "igev_lookup_convc1: groups %d / radius %d not built", G, radius);
    NND_REQUIRE(L.KH == 1 && L.KW == 1 && L.Cout == 256 && L.Cin == num_levels * IL_LC && (L.arith != 0 ? L.CI_T == 16 : L.CI_T % 32 == 0),
                "igev_lookup_convc1: layer %dx%d %d->%d does not match the lookup", L.KH, L.KW, L.Cin, L.Cout);
    ILayout IL;
    make_il_layout(B, G, H, W, num_levels, &IL, nullptr);
    NND_REQUIRE(IL.width[num_levels - 1] >= 2, "igev_lookup_convc1: level %d has width %d < 2", num_levels - 1, IL.width[num_levels - 1]);
    if (L.arith != 0) {  // convc1 packed for a split arithmetic
        const int tiles_x = cdiv(W, 8), ntiles = tiles_x * cdiv(H, 4);
        const size_t ldsb = (size_t)2 * (IL_LC / 16) * L.arith * 2 * 64 * 16;
        dim3 grid(cdiv(ntiles, 2), 1, B), block(512);
        if (L.arith == 2) {
            static std::atomic<unsigned> raised2{0};
            if (int rc = raise_lds_limit(reinterpret_cast<const void*>(igev_lookup_convc1_il_split_kernel<2>), raised2)) return rc;
            hipLaunchKernelGGL(igev_lookup_convc1_il_split_kernel<2>, grid, block, ldsb, stream, il, coords, blob + L.w_off, blob + L.b_off, c1,
                               (long)c1_bs, IL, make_lay(H, W, true), H, W, num_levels, ntiles, tiles_x, c1_c4 ? 1 : 0);
        } else {
            static std::atomic<unsigned> raised3{0};
            if (int rc = raise_lds_limit(reinterpret_cast<const void*>(igev_lookup_convc1_il_split_kernel<3>), raised3)) return rc;
            hipLaunchKernelGGL(igev_lookup_convc1_il_split_kernel<3>, grid, block, ldsb, stream, il, coords, blob + L.w_off, blob + L.b_off, c1,
                               (long)c1_bs, IL, make_lay(H, W, true), H, W, num_levels, ntiles, tiles_x, c1_c4 ? 1 : 0);
        }
        NND_LAUNCH_CHECK();
        return NND_OK;
    }
    const size_t lds = 2 * IL_LC * IL_S * sizeof(float);
    static std::atomic<unsigned> raised{0};
    if (int rc = raise_lds_limit(reinterpret_cast<const void*>(igev_lookup_convc1_il_kernel), raised)) return rc;
    const int tiles_x = cdiv(W, 8), ntiles = tiles_x * cdiv(H, 4);
    const int cb_stride = L.nchunks * (L.CI_T / 8) * 64;
    hipLaunchKernelGGL(igev_lookup_convc1_il_kernel, dim3(cdiv(ntiles, 2), 1, B), dim3(512), lds, stream, il, coords, blob + L.w_off,
                       blob + L.b_off, c1, (long)c1_bs, IL, make_lay(H, W, true), H, W, num_levels, ntiles, tiles_x, cb_stride, c1_c4 ? 1 : 0);
    NND_LAUNCH_CHECK();
    return NND_OK;
}

int convex_upsample_launch(const float* flow, const float* mask, float* out, int B, int C, int H, int W, int rate,
                           hipStream_t stream, bool tiled) {
    const Lay lay = make_lay(H, W, tiled);
    long total = (long)B * H * rate * W;
    dim3 grid((unsigned)cdiv64(total, 256)), block(256);
    if (rate == 8)
        hipLaunchKernelGGL(convex_upsample_kernel<8>, grid, block, 0, stream, flow, mask, out, B, C, H, W, lay);
    else if (rate == 4)
        hipLaunchKernelGGL(convex_upsample_kernel<4>, grid, block, 0, stream, flow, mask, out, B, C, H, W, lay);
    else if (rate == 2)
        hipLaunchKernelGGL(convex_upsample_kernel<2>, grid, block, 0, stream, flow, mask, out, B, C, H, W, lay);
    else {
        set_error("convex_upsample: rate %d not supported (2, 4, 8)", rate);
        return NND_ERR_UNSUPPORTED;
    }
    NND_LAUNCH_CHECK();
    return NND_OK;
}

}  // namespace nnd

using namespace nnd;

extern "C" {

int nnd_corr1d_pyramid_layout(int B, int H, int W, int num_levels, int64_t* level_offsets, int32_t* level_widths,
                              int64_t* total_floats) {
    NND_REQUIRE(B > 0 && H > 0 && W > 0 && num_levels >= 1 && num_levels < MAX_LEVELS, "pyramid_layout: bad shape");
    PyrLayout L;
    int64_t total;
    make_layout(B, H, W, num_levels + 1, &L, &total);
    for (int l = 0; l <= num_levels; ++l) {
        if (level_offsets) level_offsets[l] = L.off[l];
        if (level_widths) level_widths[l] = L.width[l];
    }
    if (total_floats) *total_floats = total;
    return NND_OK;
}

int nnd_corr1d_build(const float* fmap1, const float* fmap2, float* pyramid, int B, int C, int H, int W, int num_levels,
                     void* stream) {
    NND_REQUIRE(fmap1 && fmap2 && pyramid, "corr1d_build: null pointer");
    NND_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && num_levels >= 1 && num_levels < MAX_LEVELS, "corr1d_build: bad shape");
    NND_REQUIRE(num_levels <= 5, "corr1d_build: at most 5 pooled levels (32-lane tiles)");
    PyrLayout L;
    make_layout(B, H, W, num_levels + 1, &L, nullptr);
    dim3 grid(cdiv(W, 32), H, B), block(256);
    float div = (float)sqrt((double)C);
    int rc_lds = NND_OK;
    if (corr1d_build_lds_launch(fmap1, fmap2, pyramid, L, C, H, W, B, div, C, 1, (hipStream_t)stream, &rc_lds)) {
        if (rc_lds != NND_OK) set_error("corr1d_build: launch failed");
        return rc_lds;
    }
    if (C >= 64)
        hipLaunchKernelGGL(corr1d_build_kernel<32>, grid, block, 0, (hipStream_t)stream, fmap1, fmap2, pyramid, L, C, H, W, div, C, 1);
    else
        hipLaunchKernelGGL(corr1d_build_kernel<4>, grid, block, 0, (hipStream_t)stream, fmap1, fmap2, pyramid, L, C, H, W, div, C, 1);
    NND_LAUNCH_CHECK();
    return NND_OK;
}

int nnd_group_corr_build_scaled(const float* fmap1, const float* fmap2, float* pyramid, int B, int Ctot, int H, int W,
                                int num_groups, int group_channels, int num_levels, float divisor, void* stream) {
    NND_REQUIRE(fmap1 && fmap2 && pyramid, "group_corr_build: null pointer");
    NND_REQUIRE(B > 0 && H > 0 && W > 0 && num_groups > 0 && group_channels > 0 && num_groups * group_channels <= Ctot,
                "group_corr_build: bad shape (groups %d x %d channels > %d)", num_groups, group_channels, Ctot);
    NND_REQUIRE(num_levels >= 0 && num_levels <= 5, "group_corr_build: 0..5 pooled levels");
    NND_REQUIRE(divisor > 0.f, "group_corr_build: divisor must be positive");
    PyrLayout L;
    make_layout(B * num_groups, H, W, num_levels + 1, &L, nullptr);
    dim3 grid(cdiv(W, 32), H, B * num_groups), block(256);
    const float div = divisor;
    int rc_lds = NND_OK;
    if (corr1d_build_lds_launch(fmap1, fmap2, pyramid, L, group_channels, H, W, B, div, Ctot, num_groups, (hipStream_t)stream, &rc_lds)) {
        if (rc_lds != NND_OK) set_error("group_corr_build: launch failed");
        return rc_lds;
    }
    if (group_channels >= 64)
        hipLaunchKernelGGL(corr1d_build_kernel<32>, grid, block, 0, (hipStream_t)stream, fmap1, fmap2, pyramid, L, group_channels,
                           H, W, div, Ctot, num_groups);
    else
        hipLaunchKernelGGL(corr1d_build_kernel<4>, grid, block, 0, (hipStream_t)stream, fmap1, fmap2, pyramid, L, group_channels,
                           H, W, div, Ctot, num_groups);
    NND_LAUNCH_CHECK();
    return NND_OK;
}

int nnd_group_corr_build(const float* fmap1, const float* fmap2, float* pyramid, int B, int Ctot, int H, int W,
                         int num_groups, int group_channels, int num_levels, void* stream) {
    return nnd_group_corr_build_scaled(fmap1, fmap2, pyramid, B, Ctot, H, W, num_groups, group_channels, num_levels,
                                       group_channels > 0 ? (float)sqrt((double)group_channels) : 1.f, stream);
}

int nnd_group_corr1d_lookup(const float* pyramid, const float* coords, float* out, int B, int G, int H, int W, int num_levels,
                            int radius, void* stream) {
    NND_REQUIRE(pyramid && coords && out, "group_corr1d_lookup: null pointer");
    NND_REQUIRE(B > 0 && G > 0 && H > 0 && W > 0 && radius >= 0, "group_corr1d_lookup: bad shape");
    return group_lookup_flat_launch(pyramid, coords, out, B, G, H, W, num_levels, radius, (hipStream_t)stream, false);
}

int nnd_pyramid_from_level0(float* pyramid, int B, int H, int W, int num_levels, void* stream) {
    NND_REQUIRE(pyramid && B > 0 && H > 0 && W > 0 && num_levels >= 1 && num_levels < MAX_LEVELS, "pyramid_from_level0: bad argument");
    PyrLayout L;
    make_layout(B, H, W, num_levels + 1, &L, nullptr);
    const long rows = (long)B * H * W;
    for (int l = 1; l <= num_levels; ++l) {
        if (L.width[l] == 0) break;
        long total = rows * L.width[l];
        hipLaunchKernelGGL(avg_pool_level_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream,
                           pyramid + L.off[l - 1], pyramid + L.off[l], rows, L.width[l - 1], L.width[l]);
        NND_LAUNCH_CHECK();
    }
    return NND_OK;
}


int64_t nnd_igev_interleaved_floats(int B, int G, int H, int W, int num_levels) {
    if (B <= 0 || G <= 0 || H <= 0 || W <= 0 || num_levels < 1 || num_levels >= MAX_LEVELS) return 0;
    ILayout IL;
    int64_t total;
    make_il_layout(B, G, H, W, num_levels, &IL, &total);
    return total;
}

int nnd_igev_interleave_pyramids(const float* feat_pyramid, const float* geo_pyramid, float* interleaved, int B, int G, int H,
                                 int W, int num_levels, void* stream) {
    NND_REQUIRE(feat_pyramid && geo_pyramid && interleaved, "igev_interleave_pyramids: null pointer");
    NND_REQUIRE(B > 0 && G > 0 && 2 * G <= 32 && H > 0 && W > 0 && num_levels >= 1 && num_levels < MAX_LEVELS,
                "igev_interleave_pyramids: bad shape");
    PyrLayout L;
    make_layout(B * G, H, W, num_levels + 1, &L, nullptr);
    ILayout IL;
    make_il_layout(B, G, H, W, num_levels, &IL, nullptr);
    const long HW = (long)H * W;
    NND_REQUIRE((long)B * HW < (1L << 31), "igev_interleave_pyramids: too many rows");
    for (int l = 0; l < num_levels; ++l) {
        if (L.width[l] == 0) break;
        hipLaunchKernelGGL(igev_interleave_kernel, dim3((unsigned)(B * HW), cdiv(L.width[l], 64)), dim3(256), 0, (hipStream_t)stream,
                           feat_pyramid + L.off[l], geo_pyramid + L.off[l], interleaved + IL.off[l], G, HW, L.width[l]);
        NND_LAUNCH_CHECK();
    }
    return NND_OK;
}

static size_t il_pool_lds_bytes(int G, int W, int num_levels) {
    size_t f = 0;
    int w = W;
    for (int l = 0; l < num_levels; ++l, w /= 2) f += (size_t)2 * G * il_pool_stride(w);
    return f * sizeof(float);
}

int nnd_igev_interleave_level0_supported(int G, int W, int num_levels) {
    return G > 0 && 2 * G <= 32 && W > 0 && num_levels >= 1 && num_levels < MAX_LEVELS && (W >> (num_levels - 1)) > 0 &&
           il_pool_lds_bytes(G, W, num_levels) <= 160 * 1024;
}

int nnd_igev_interleave_level0(const float* feat_level0, const float* geo_level0, float* interleaved, int B, int G, int H, int W,
                               int num_levels, void* stream) {
    NND_REQUIRE(feat_level0 && geo_level0 && interleaved, "igev_interleave_level0: null pointer");
    NND_REQUIRE(B > 0 && H > 0 && nnd_igev_interleave_level0_supported(G, W, num_levels),
                "igev_interleave_level0: bad shape (2*G <= 32, every level at least 1 wide, the pixel's levels within 160 KB of LDS)");
    ILayout IL;
    make_il_layout(B, G, H, W, num_levels, &IL, nullptr);
    const long HW = (long)H * W;
    NND_REQUIRE((long)B * HW < (1L << 31), "igev_interleave_level0: too many rows");
    const size_t lds = il_pool_lds_bytes(G, W, num_levels);
    const bool v4 = (W & 3) == 0 && (G & 1) == 0;
    static std::atomic<unsigned> raised[2];
    const void* kern = v4 ? reinterpret_cast<const void*>(igev_pool_interleave_kernel<true>)
                          : reinterpret_cast<const void*>(igev_pool_interleave_kernel<false>);
    if (lds > 64 * 1024 && raise_lds_limit(kern, raised[v4]) != NND_OK) return NND_ERR_HIP;
    if (v4)
        hipLaunchKernelGGL(igev_pool_interleave_kernel<true>, dim3((unsigned)(B * HW)), dim3(256), lds, (hipStream_t)stream, feat_level0,
                           geo_level0, interleaved, IL, G, HW, W, num_levels);
    else
        hipLaunchKernelGGL(igev_pool_interleave_kernel<false>, dim3((unsigned)(B * HW)), dim3(256), lds, (hipStream_t)stream, feat_level0,
                           geo_level0, interleaved, IL, G, HW, W, num_levels);
    NND_LAUNCH_CHECK();
    return NND_OK;
}
}  // extern "C"

namespace nnd {
int igev_lookup_launch(const float* feat_pyramid, const float* geo_pyramid, const float* coords, float* out, int B, int G, int H,
                       int W, int num_levels, int radius, hipStream_t stream, bool tiled) {
    NND_REQUIRE(feat_pyramid && geo_pyramid && coords && out, "igev_lookup: null pointer");
    NND_REQUIRE(B > 0 && G > 0 && H > 0 && W > 0 && radius >= 0 && num_levels >= 1 && num_levels < MAX_LEVELS, "igev_lookup: bad shape");
    LookupArgs a;
    make_layout(B * G, H, W, num_levels + 1, &a.L, nullptr);
    a.B = B; a.H = H; a.W = W; a.num_levels = num_levels; a.radius = radius;
    a.lay = make_lay(H, W, tiled);
    NND_REQUIRE(a.L.width[num_levels - 1] >= 2, "igev_lookup: level %d has width %d < 2", num_levels - 1, a.L.width[num_levels - 1]);
    long total = (long)B * num_levels * 2 * G * (2 * radius + 1) * H * W;
    hipLaunchKernelGGL(igev_lookup_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, stream, feat_pyramid,
                       geo_pyramid, coords, out, a, G);
    NND_LAUNCH_CHECK();
    return NND_OK;
}
}  // namespace nnd

// IGEV initial disparity (igev_stereo/model.py:92-95,145-146): out[b,0,h,w] = -sum_d d * softmax_d(logits[b,d,h,w]).
// Evaluated in the REFERENCE's fp32 order (round 4; scripts/study/igev_init_order.py, profiles/r04_igev_init_order_study.txt): the
// expectation over 240 candidates reaches ~120, where the order of the additions decides the last 1e-4 — the more accurate
// (sum d e_d) / (sum e_d) with pairwise sums that rounds 1-3 used lies 2e-5 from the float64 value but 1.8e-4 from what the
// reference computes; ATen's own order lies 3e-5 from it.  That order: softmax over a non-innermost dim accumulates
// exp(x - max) strictly in order of d and divides every term (SoftMaxKernel.cpp: vec_softmax); `disp * p` is an elementwise
// product; sum(dim=1) is cascade_sum / multi_row_sum (SumKernel.cpp): rows are added in order in blocks of 16, a block's sum is
// added to a second accumulator, every 16 blocks that one is added to a third; the tail rows (D % 16) stay in the first
// accumulator and the levels are added lowest first.  `aten_expectation` is that last step for one pixel.
template <typename F>
__device__ __forceinline__ float aten_cascade_sum(int D, F&& term) {
    float lvl1 = 0.f, lvl2 = 0.f;
    int d = 0;
    for (int blk = 0; d + 16 <= D; ++blk) {
        float b = 0.f;
        for (int j = 0; j < 16; ++j, ++d) b += term(d);
        lvl1 += b;
        if (((blk + 1) & 15) == 0) {
            lvl2 += lvl1;
            lvl1 = 0.f;
        }
    }
    float tail = 0.f;
    for (; d < D; ++d) tail += term(d);
    return (tail + lvl1) + lvl2;
}

// One thread per pixel, three passes over the D candidates (max, sum of exp, expectation); consecutive lanes are
// consecutive pixels, so every pass reads coalesced rows of the (B,D,H,W) volume.
__global__ void __launch_bounds__(256) softargmin_kernel(const float* __restrict__ logits, float* __restrict__ out, int D,
                                                         long HW, long total) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const long b = idx / HW, p = idx - b * HW;
    const float* x = logits + b * D * HW + p;
    float mx = -INFINITY;
    for (int d = 0; d < D; ++d) mx = fmaxf(mx, x[(long)d * HW]);
    float sum = 0.f;
    for (int d = 0; d < D; ++d) sum += expf(x[(long)d * HW] - mx);
    out[idx] = -aten_cascade_sum(D, [&](int d) { return (float)d * (expf(x[(long)d * HW] - mx) / sum); });
}

// IGEV cv_squeezer + initial disparity in one pass over the regularised volume (igev_stereo/model.py:144-146):
//   logits[b,d,h,w] = bias + sum_{g,kd,kh,kw} Wt[g][kd][kh][kw] * geo[b,g,h+kh-1,w+kw-1,d+kd-1]   (Conv3d(G,1,3,1,1), zero pad)
//   out[b,0,h,w]    = -sum_d d * softmax_d(logits)
// geo = level 0 of the geometry pyramid, rows (b,g,h,w1) of D = W2 floats, read in place (the reference permutes it to
// (B,G,W2,H,W1) for the Conv3d; here the candidate axis stays the contiguous one).  One workgroup = SQ_PX pixels of one
// image row; thread = candidate d (DPT per thread).  Per group g the 3 x (SQ_PX+2) neighbour rows go through LDS
// (zero-padded by one candidate either side), every LDS value feeds up to 9 FMAs from registers; the logits never
// reach HBM.  Weights sit in the kernel arguments (scalar registers).
constexpr int SQ_PX = 8, SQ_MAXG = 8;
struct SqueezeArgs {
    float w[SQ_MAXG * 27];  // [g][kd][kh][kw]
    float bias;
};

// Soft-argmin over the candidates of SQ_PX pixels whose logits (without the bias) sit in acc[j][p], candidate tid + 256 j:
// block reductions for the maxima, then the reference's evaluation order (see softargmin_kernel) through `es` = SQ_PX x D
// floats of LDS that no thread reads any more (the function opens with a barrier).  All 256 threads call it.
template <int DPT>
__device__ __forceinline__ void squeeze_softargmin_tail(float (&acc)[DPT][SQ_PX], float bias, float* es, float* __restrict__ out_row,
                                                        int w0, int W, int D) {
    __shared__ float red[4][SQ_PX];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // soft-argmin over the candidates: block reductions (max, sum of exp, sum of d*exp) for the SQ_PX pixels at once
    auto block_reduce = [&](float (&v)[SQ_PX], bool is_max) {
#pragma unroll
        for (int p = 0; p < SQ_PX; ++p) {
            float x = v[p];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float y = __shfl_xor(x, o);
                x = is_max ? fmaxf(x, y) : x + y;
            }
            v[p] = x;
        }
        __syncthreads();
        if (lane == 0)
#pragma unroll
            for (int p = 0; p < SQ_PX; ++p) red[wave][p] = v[p];
        __syncthreads();
#pragma unroll
        for (int p = 0; p < SQ_PX; ++p)
            v[p] = is_max ? fmaxf(fmaxf(red[0][p], red[1][p]), fmaxf(red[2][p], red[3][p])) : (red[0][p] + red[1][p]) + (red[2][p] + red[3][p]);
    };
    float mx[SQ_PX];
#pragma unroll
    for (int p = 0; p < SQ_PX; ++p) {
        mx[p] = -INFINITY;
#pragma unroll
        for (int j = 0; j < DPT; ++j)
            if (tid + 256 * j < D) mx[p] = fmaxf(mx[p], acc[j][p] + bias);
    }
    block_reduce(mx, true);
    // the reference's evaluation order (see softargmin_kernel): the staged rows are dead, `sm` now holds e[p][d] = exp(logit - max)
    __shared__ float ssum[SQ_PX], bsum[SQ_PX][33];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < DPT; ++j) {
        const int d = tid + 256 * j;
        if (d < D)
#pragma unroll
            for (int p = 0; p < SQ_PX; ++p) es[p * D + d] = expf(acc[j][p] + bias - mx[p]);
    }
    __syncthreads();
    if (tid < SQ_PX) {  // softmax denominator: strictly in order of d
        float sden = 0.f;
        for (int d = 0; d < D; ++d) sden += es[tid * D + d];
        ssum[tid] = sden;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < DPT; ++j) {
        const int d = tid + 256 * j;
        if (d < D)
#pragma unroll
            for (int p = 0; p < SQ_PX; ++p) es[p * D + d] = (float)d * (es[p * D + d] / ssum[p]);  // disp * softmax: a product, then summed
    }
    __syncthreads();
    {  // block sums of 16 consecutive candidates, in order (thread = pixel x block; D <= 512: at most 32 blocks)
        const int p = tid >> 5, blk = tid & 31;
        if (blk * 16 + 16 <= D) {
            float bs = 0.f;
            for (int j = 0; j < 16; ++j) bs += es[p * D + blk * 16 + j];
            bsum[p][blk] = bs;
        }
    }
    __syncthreads();
    if (tid < SQ_PX && w0 + tid < W) {  // the levels of ATen's cascade: block sums in order, every 16 blocks into the next level
        const int nb = D >> 4;
        float lvl1 = 0.f, lvl2 = 0.f, tail = 0.f;
        for (int blk = 0; blk < nb; ++blk) {
            lvl1 += bsum[tid][blk];
            if (((blk + 1) & 15) == 0) {
                lvl2 += lvl1;
                lvl1 = 0.f;
            }
        }
        for (int d = nb * 16; d < D; ++d) tail += es[tid * D + d];
        out_row[w0 + tid] = -((tail + lvl1) + lvl2);
    }
}

template <int DPT>
__global__ void __launch_bounds__(256) igev_squeeze_softargmin_kernel(const float* __restrict__ geo, float* __restrict__ out,
                                                                      SqueezeArgs a, int G, int H, int W, int D) {
    extern __shared__ float sm[];  // [3][SQ_PX + 2][D + 2]
    const int tid = threadIdx.x;
    // XCD-aware mapping: consecutive workgroup ids go round-robin to the 8 XCDs (one L2 each); XCD x gets the band of
    // image rows [x*band, (x+1)*band), so the three rows a workgroup shares with its vertical neighbours stay in one L2
    const int nx = (W + SQ_PX - 1) / SQ_PX, band = (H + 7) / 8;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int h = xcd * band + slot / nx, w0 = (slot % nx) * SQ_PX, b = blockIdx.z;
    if (h >= H) return;
    const int DS = D + 2;
    float acc[DPT][SQ_PX];
#pragma unroll
    for (int j = 0; j < DPT; ++j)
#pragma unroll
        for (int p = 0; p < SQ_PX; ++p) acc[j][p] = 0.f;
    // rows of group g: 3 x (SQ_PX+2) rows of D floats.  D % 4 == 0: 16-byte loads, the rows of group g+1 are fetched into
    // registers while group g is multiplied (latency hidden behind the FMAs); else scalar loads straight into LDS.
    constexpr int NROW = 3 * (SQ_PX + 2), NL = (NROW * (DPT * 64) + 255) / 256;
    const bool vec = (D & 3) == 0;
    const int nq = D >> 2;
    float4 stage[NL];
    auto row_src = [&](int g, int r, bool& ok) {
        const int kh = r / (SQ_PX + 2), c = r - kh * (SQ_PX + 2);
        const int hh = h + kh - 1, ww = w0 + c - 1;
        ok = hh >= 0 && hh < H && ww >= 0 && ww < W;
        return geo + ((((long)b * G + g) * H + (ok ? hh : 0)) * W + (ok ? ww : 0)) * D;
    };
    auto fetch = [&](int g) {
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int i = tid + 256 * k;
            const bool in = i < NROW * nq;
            const int r = in ? i / nq : 0, q = in ? i - r * nq : 0;
            bool ok;
            const float* src = row_src(g, r, ok);
            const float4 v = *reinterpret_cast<const float4*>(src + 4 * q);
            stage[k] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int i = tid + 256 * k;
            if (i < NROW * nq) {
                const int r = i / nq, q = i - r * nq;
                float* dst = sm + r * DS + 1 + 4 * q;
                dst[0] = stage[k].x; dst[1] = stage[k].y; dst[2] = stage[k].z; dst[3] = stage[k].w;
            }
        }
    };
    for (int r = tid; r < NROW; r += 256) sm[r * DS] = 0.f, sm[r * DS + D + 1] = 0.f;  // candidate -1 and D: zero padding
    if (vec) fetch(0);
    for (int g = 0; g < G; ++g) {
        __syncthreads();
        if (vec) {
            commit();
        } else {
            for (int r = 0; r < NROW; ++r) {
                bool ok;
                const float* src = row_src(g, r, ok);
                for (int d = tid; d < D; d += 256) sm[r * DS + 1 + d] = ok ? src[d] : 0.f;
            }
        }
        __syncthreads();
        if (vec && g + 1 < G) fetch(g + 1);
        const float* wg = a.w + g * 27;
#pragma unroll
        for (int j = 0; j < DPT; ++j) {
            const int d = tid + 256 * j;
            if (d < D) {
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int c = 0; c < SQ_PX + 2; ++c) {
                        const float* row = sm + (kh * (SQ_PX + 2) + c) * DS + d;
                        const float r0 = row[0], r1 = row[1], r2 = row[2];
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) {
                            const int p = c - kw;
                            if (p >= 0 && p < SQ_PX)
                                acc[j][p] += wg[0 * 9 + kh * 3 + kw] * r0 + wg[1 * 9 + kh * 3 + kw] * r1 + wg[2 * 9 + kh * 3 + kw] * r2;
                        }
                    }
            }
        }
    }
    squeeze_softargmin_tail<DPT>(acc, a.bias, sm, out + ((long)b * H + h) * W, w0, W, D);  // the staged rows are dead: sm holds e[p][d]
}

// The same squeezer, walking down the image: the kernel above fetches 3 x (SQ_PX + 2) neighbour rows per group for the SQ_PX
// pixels of ONE image row (3.75 x the volume through L1) and runs one soft-argmin (with its two strictly sequential sums over
// the candidates, 8 busy lanes) per row.  Here a workgroup owns SQ_PX columns x R image rows and visits the R + 2 input rows
// in order; each (row, group) slab of SQ_PX + 2 columns x D candidates is staged ONCE and feeds the three output rows it
// neighbours (kh = 2, 1, 0) from three accumulator sets, which rotate when a row is done: (1 + 2/SQ_PX)(1 + 2/R) x the
// volume, every LDS value read feeds 27 FMAs instead of 9.  Slab s + 1 is fetched into registers while slab s is multiplied
// and lands in the other LDS buffer: one barrier per slab.  Finished rows park their logits in LDS and the soft-argmin runs
// for SQW_TB rows at once (SQW_TB x SQ_PX sequential chains side by side instead of SQ_PX).  D <= 256, D % 4 == 0.
// The sum over (g, kh, kw, kd) runs in the order (kh, g, column, kw, kd), one FMA per term, instead of (g, kh, column, kw,
// [kd]) with the three kd terms added first: last-bit differences in the logits; the soft-argmin arithmetic is the same.
constexpr int SQW_TB = 4, SQW_T = 512, SQW_PXT = SQ_PX / 2;
// one slab into the accumulators of a thread: candidate d, pixels p0 .. p0 + SQW_PXT - 1 of the strip (slab columns p0 .. p0 + SQW_PXT + 1).
// Two pixels share a packed FMA: for the pixel pair (2j, 2j+1) and slab column c = 2j + e the taps are kw = e and e - 1, so the pair
// takes (w1, w0) at e = 1 and (w2, w1) at e = 2 as ONE v_pk_fma_f32 with the candidate's value in both halves, and single FMAs at
// e = 0 (w0, first pixel) and e = 3 (w2, second pixel).  The weight quads (w2, w1, w1, w0) of a (group, kh, kd) sit in LDS (wq:
// [kh][kd]), one broadcast ds_read_b128 each: fetched from the kernel arguments the compiler rebuilt every pair with v_mov (50 of
// the 104 VALU instructions of a slab) behind scalar loads issued at the top of every slab.  Per accumulator the terms arrive in the
// order (column, kd) = the order of the scalar form: bit-identical logits.
typedef float sq_f2 __attribute__((ext_vector_type(2)));
template <int MASK>
__device__ __forceinline__ void squeeze_walk_slab(const float* __restrict__ cols, int DS, const float4* __restrict__ wq,
                                                  sq_f2 (&acc)[3][SQW_PXT / 2]) {
    static_assert(SQW_PXT % 2 == 0, "pixel pairs");
    float rr[SQW_PXT + 2][3];  // candidates d - 1, d, d + 1 of the slab columns p0 ..: read once, used by the three accumulator sets
#pragma unroll
    for (int c = 0; c < SQW_PXT + 2; ++c) {
        const float* row = cols + c * DS;
        rr[c][0] = row[0], rr[c][1] = row[1], rr[c][2] = row[2];
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        if (!((MASK >> t) & 1)) continue;
        float4 w[3];  // (w2, w1, w1, w0) per kd; set t sees this input row as its kh = 2 - t
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) w[kd] = wq[(2 - t) * 3 + kd];
#pragma unroll
        for (int c = 0; c < SQW_PXT + 2; ++c) {
#pragma unroll
            for (int j = 0; j < SQW_PXT / 2; ++j) {
                const int e = c - 2 * j;
                if (e < 0 || e > 3) continue;
#pragma unroll
                for (int kd = 0; kd < 3; ++kd) {
                    const float4 q = w[kd];
                    const float r = rr[c][kd];
                    if (e == 0) acc[t][j].x = fmaf(q.w, r, acc[t][j].x);
                    else if (e == 1) acc[t][j] = __builtin_elementwise_fma(sq_f2{q.z, q.w}, sq_f2{r, r}, acc[t][j]);
                    else if (e == 2) acc[t][j] = __builtin_elementwise_fma(sq_f2{q.x, q.y}, sq_f2{r, r}, acc[t][j]);
                    else acc[t][j].y = fmaf(q.x, r, acc[t][j].y);
                }
            }
        }
    }
}

// soft-argmin of `nrow` parked rows (lg[(row * SQ_PX + p) * DP + d] = logit incl. bias) in the reference's evaluation order
// (softargmin_kernel): max, e = exp(l - max), sum of e strictly in order of d, terms d * (e / sum), ATen's cascade over them.
// The two ordered sums are one chain per (row, pixel), spread over the eight waves.  All threads call it; opens with a barrier.
__device__ __forceinline__ void squeeze_walk_softargmin(float* lg, int DP, int D, int nrow, float* __restrict__ out_row0, int W, int w0) {
    __shared__ float mxs[SQW_TB * SQ_PX], sums[SQW_TB * SQ_PX];
    constexpr int NW = SQW_T / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, dl = tid & 255, half = tid >> 8;
    const int npair = nrow * SQ_PX;
    __syncthreads();
    for (int q = wave; q < npair; q += NW) {
        float m = -INFINITY;
        for (int dd = lane; dd < D; dd += 64) m = fmaxf(m, lg[q * DP + dd]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0) mxs[q] = m;
    }
    __syncthreads();
    if (dl < D)
        for (int q = half; q < npair; q += 2) lg[q * DP + dl] = expf(lg[q * DP + dl] - mxs[q]);
    __syncthreads();
    const int cq = lane * NW + wave;  // chain of this thread (the first lanes of every wave)
    const bool chain = lane < SQW_TB * SQ_PX / NW && cq < npair;
    if (chain) {  // 4 x 16 bytes per step: the LDS latency is paid once per 16 ordered adds
        const float4* e4 = reinterpret_cast<const float4*>(lg + cq * DP);
        const int n4 = D >> 2;
        float sden = 0.f;
        for (int i = 0; i < n4; i += 4) {
            float4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = e4[min(i + k, n4 - 1)];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (i + k < n4) sden += v[k].x, sden += v[k].y, sden += v[k].z, sden += v[k].w;
        }
        sums[cq] = sden;
    }
    __syncthreads();
    if (dl < D)
        for (int q = half; q < npair; q += 2) lg[q * DP + dl] = (float)dl * (lg[q * DP + dl] / sums[q]);
    __syncthreads();
    if (chain) {
        const int p = cq % SQ_PX, row = cq / SQ_PX;
        const float4* t4 = reinterpret_cast<const float4*>(lg + cq * DP);
        float lvl1 = 0.f, lvl2 = 0.f, tail = 0.f;  // aten_cascade_sum with the 16 terms of a block read as 4 x 16 bytes
        const int nb = D >> 4;
        for (int blk = 0; blk < nb; ++blk) {
            float4 x[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) x[k] = t4[4 * blk + k];
            float bs = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) bs += x[k].x, bs += x[k].y, bs += x[k].z, bs += x[k].w;
            lvl1 += bs;
            if (((blk + 1) & 15) == 0) {
                lvl2 += lvl1;
                lvl1 = 0.f;
            }
        }
        for (int i = 4 * nb; i < (D >> 2); ++i) {
            const float4 x = t4[i];
            tail += x.x, tail += x.y, tail += x.z, tail += x.w;
        }
        const float v = (tail + lvl1) + lvl2;
        if (w0 + p < W) out_row0[(long)row * W + w0 + p] = -v;
    }
}

template <int GS, int RP>
__global__ void __launch_bounds__(SQW_T, 2) igev_squeeze_walk_kernel(const float* __restrict__ geo, float* __restrict__ out, SqueezeArgs a,
                                                                  int G, int H, int W, int D, int R, int nwg) {
    extern __shared__ float sm[];  // slabs [2][(SQ_PX + 2) * DS + 4], then the parked logits [SQW_TB * SQ_PX][DP]
    const int tid = threadIdx.x, dl = tid & 255, half = tid >> 8;  // thread: candidate dl, pixels 4 half .. 4 half + 3
    const int per_xcd = (nwg + 7) >> 3;  // consecutive logical ids (strips of one band, then the next band) share one XCD's L2
    const int wg_id = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (wg_id >= nwg) return;
    const int nx = (W + SQ_PX - 1) / SQ_PX, nband = (H + R - 1) / R;
    const int w0 = (wg_id % nx) * SQ_PX, band = (wg_id / nx) % nband, b = wg_id / (nx * nband);
    const int h0 = band * R, hend = min(h0 + R, H);
    // slab row c: [c * DS, c * DS + 4) pad (candidate -1 at +3), then D candidates; candidate D = the next row's first pad float
    constexpr int NROW = SQ_PX + 2, NL = (NROW * 64 + SQW_T - 1) / SQW_T;
    static_assert(RP % 2 == 0, "the LDS buffer of a slab is its ring slot's parity");
    const int DS = D + 4, nq = D >> 2, SUB = NROW * DS + 4, SLAB = GS * SUB;  // a slab = GS groups of one input row (G % GS == 0)
    const int DP = ((D >> 2) & 1) ? D : D + 4;  // odd number of 16-byte units per parked row: the chains' b128 reads spread over the banks
    float* lg = sm + 2 * SLAB;
    float4* wl = reinterpret_cast<float4*>(lg + SQW_TB * SQ_PX * DP);  // [G][kh][kd] weight quads (w2, w1, w1, w0) over kw
    sq_f2 acc[3][SQW_PXT / 2];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int j = 0; j < SQW_PXT / 2; ++j) acc[t][j] = sq_f2{0.f, 0.f};
    const int rlo = max(h0 - 1, 0), rhi = min(hend, H - 1);  // input rows that exist
    const int nslab = (rhi - rlo + 1) * (G / GS);
    // RP slabs are on their way or parked in registers at any time; the register ring is indexed statically, so the slab
    // loop is unrolled RP times.  The 16-byte units of a thread: slab-invariant offsets.
    float4 stage[RP][GS * NL];
    int src_off[NL], dst_off[NL];  // floats from the slab's (row, group) base / from the LDS slab; -1: column outside the image or no unit
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        const int i = tid + SQW_T * k;
        const bool in = i < NROW * nq;
        const int c = in ? i / nq : 0, q = in ? i - c * nq : 0;
        const int ww = w0 + c - 1;
        src_off[k] = (in && ww >= 0 && ww < W) ? ww * D + 4 * q : -1;
        dst_off[k] = in ? c * DS + 4 + 4 * q : -1;
    }
    const long row_floats = (long)W * D;
    auto fetch = [&](int r, int g, float4 (&st)[GS * NL]) {
#pragma unroll
        for (int gs = 0; gs < GS; ++gs) {
            const float* base = geo + (((long)b * G + g + gs) * H + r) * row_floats;
#pragma unroll
            for (int k = 0; k < NL; ++k) {
                const float4 v = *reinterpret_cast<const float4*>(base + max(src_off[k], 0));
                st[gs * NL + k] = src_off[k] >= 0 ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    auto commit = [&](int parity, const float4 (&st)[GS * NL]) {
        float* buf = sm + parity * SLAB;
#pragma unroll
        for (int gs = 0; gs < GS; ++gs)
#pragma unroll
            for (int k = 0; k < NL; ++k)
                if (dst_off[k] >= 0) *reinterpret_cast<float4*>(buf + gs * SUB + dst_off[k]) = st[gs * NL + k];
    };
    for (int i = tid; i < G * 9; i += SQW_T) {  // a.w is [g][kd][kh][kw]
        const int g_ = i / 9, kh = (i % 9) / 3, kd = i % 3;
        const float* w3 = a.w + g_ * 27 + kd * 9 + kh * 3;
        wl[i] = make_float4(w3[2], w3[1], w3[1], w3[0]);
    }
    for (int i = tid; i < 2 * GS * (NROW + 1); i += SQW_T) {  // the pads: candidates -1 and D of every slab row
        float* pad = sm + (i / (NROW + 1)) * SUB + (i % (NROW + 1)) * DS;
        pad[0] = pad[1] = pad[2] = pad[3] = 0.f;
    }
    int fr = rlo, fg = 0;  // (row, group) of the next slab to fetch
    auto fetch_next = [&](float4 (&st)[GS * NL]) {
        fetch(fr, fg, st);
        if ((fg += GS) == G) fg = 0, ++fr;
    };
#pragma unroll
    for (int u = 0; u < RP; ++u)
        if (u < nslab) fetch_next(stage[u]);
    commit(0, stage[0]);
    if (RP < nslab) fetch_next(stage[0]);
    __syncthreads();
    const int cols_off = (SQW_PXT * half) * DS + 3 + min(dl, D - 1);  // threads beyond D multiply a copy of the last candidate; not parked
    int parked = 0, park_h = h0;  // rows waiting for their soft-argmin, the first of them
    int r = rlo, g = 0;           // (row, group) of slab s
    for (int s0 = 0; s0 < nslab; s0 += RP) {
#pragma unroll
        for (int u = 0; u < RP; ++u) {
            const int s = s0 + u;  // buffer s & 1 = u & 1 holds it; ring slots u + 1 .. hold slabs s + 1 .., slot u slab s + RP
            if (s >= nslab) break;
            const float* cols = sm + (u & 1) * SLAB + cols_off;
            // rows whose three neighbours are not all in the band: the band's first / last input row feed one output row;
            // its first / last own row (mask 6 / 3) takes the full walk, the set outside the band is never parked
            const bool t0 = r - 1 >= h0 && r - 1 < hend, t1 = r >= h0 && r < hend, t2 = r + 1 >= h0 && r + 1 < hend;
#pragma unroll
            for (int gs = 0; gs < GS; ++gs) {
                const float4* wg = wl + (g + gs) * 9;
                const float* cg = cols + gs * SUB;
                if (t2 && !t0 && !t1) squeeze_walk_slab<4>(cg, DS, wg, acc);
                else if (t0 && !t1 && !t2) squeeze_walk_slab<1>(cg, DS, wg, acc);
                else squeeze_walk_slab<7>(cg, DS, wg, acc);
            }
            if (s + 1 < nslab) commit((u + 1) & 1, stage[(u + 1) % RP]);
            __syncthreads();
            if (s + 1 + RP < nslab) fetch_next(stage[(u + 1) % RP]);
            if (g + GS == G) {  // input row r is done: output row r - 1 is complete, and on the image's last row so is row r (no row H)
                const int nfin = (r == H - 1 && r < hend) ? 2 : 1;
                for (int f = 0; f < nfin; ++f) {
                    const int h = r - 1 + f;
                    if (h >= h0 && h < hend) {
                        if (dl < D)
#pragma unroll
                            for (int p = 0; p < SQW_PXT; ++p) lg[(parked * SQ_PX + SQW_PXT * half + p) * DP + dl] = acc[0][p >> 1][p & 1] + a.bias;
                        ++parked;
                    }
#pragma unroll
                    for (int j = 0; j < SQW_PXT / 2; ++j) acc[0][j] = acc[1][j], acc[1][j] = acc[2][j], acc[2][j] = sq_f2{0.f, 0.f};
                    if (parked == SQW_TB || (parked > 0 && h == hend - 1)) {
                        squeeze_walk_softargmin(lg, DP, D, parked, out + ((long)b * H + park_h) * W, W, w0);
                        park_h += parked;
                        parked = 0;
                        __syncthreads();  // the chains read lg to the end
                    }
                }
            }
            if ((g += GS) == G) g = 0, ++r;
        }
    }
}

extern "C" {
int nnd_softargmin_disparity(const float* logits, float* out, int B, int D, int H, int W, void* stream) {
    NND_REQUIRE(logits && out && B > 0 && D > 0 && H > 0 && W > 0, "softargmin_disparity: bad argument");
    const long HW = (long)H * W, total = (long)B * HW;
    hipLaunchKernelGGL(softargmin_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, logits, out, D, HW, total);
    NND_LAUNCH_CHECK();
    return NND_OK;
}

int nnd_igev_init_disparity(const float* geo_level0, const float* weight, const float* bias, float* out, int B, int G, int H,
                            int W, int D, void* stream) {
    NND_REQUIRE(geo_level0 && weight && out, "igev_init_disparity: null pointer");
    NND_REQUIRE(B > 0 && G > 0 && H > 0 && W > 0 && D > 0, "igev_init_disparity: bad shape");
    if (G > SQ_MAXG || D > 512 || B > 65535) {
        set_error("igev_init_disparity: groups %d (max %d) / candidates %d (max 512) not built", G, SQ_MAXG, D);
        return NND_ERR_UNSUPPORTED;
    }
    SqueezeArgs a;  // weight / bias are HOST pointers: the 27*G weights and the bias travel as kernel arguments
    for (int i = 0; i < G * 27; ++i) a.w[i] = weight[i];
    for (int i = G * 27; i < SQ_MAXG * 27; ++i) a.w[i] = 0.f;
    a.bias = bias ? bias[0] : 0.f;
    // the walking kernel, bands of 8 rows, when those fill the chip (2 workgroups of 512 per CU); smaller problems: measured slower
    // than the one-row kernel (68x120x120: 68 vs 54 us with bands of 2 rows), which also takes D > 256 and D % 4 != 0
    // (NND_IGEV_SQUEEZE_WALK: whenever it can run — the tests' small shapes)
    if (D <= 256 && (D & 3) == 0 && !switches().igev_squeeze_v1 &&
        ((long)cdiv(W, SQ_PX) * cdiv(H, 8) * B >= 384 || switches().igev_squeeze_walk)) {
        const int R = 8;
        const long nwg = (long)cdiv(W, SQ_PX) * cdiv(H, R) * B;
        NND_REQUIRE(nwg < (1L << 30), "igev_init_disparity: grid too large");
        // one group per slab, 4 slabs in the register ring.  (Two groups per slab and a ring of 2 — half the barriers, the same bytes
        // in flight, 72 KB of LDS — measured 158 us against 130: igev_squeeze_walk_kernel<2, 2> stays instantiable, not launched.)
        const size_t lds_walk = sizeof(float) * ((size_t)2 * ((SQ_PX + 2) * (D + 4) + 4) + (size_t)SQW_TB * SQ_PX * (D + 4) + (size_t)SQ_MAXG * 9 * 4);
        hipLaunchKernelGGL((igev_squeeze_walk_kernel<1, 4>), dim3((unsigned)(8 * cdiv64(nwg, 8))), dim3(SQW_T), lds_walk, (hipStream_t)stream,
                           geo_level0, out, a, G, H, W, D, R, (int)nwg);
        NND_LAUNCH_CHECK();
        return NND_OK;
    }
    const size_t lds = (size_t)3 * (SQ_PX + 2) * (D + 2) * sizeof(float);
    dim3 grid(cdiv(W, SQ_PX) * 8 * cdiv(H, 8), 1, B), block(256);  // 8 bands of ceil(H/8) rows, see the kernel
    if (D <= 256)
        hipLaunchKernelGGL(igev_squeeze_softargmin_kernel<1>, grid, block, lds, (hipStream_t)stream, geo_level0, out, a, G, H, W, D);
    else
        hipLaunchKernelGGL(igev_squeeze_softargmin_kernel<2>, grid, block, lds, (hipStream_t)stream, geo_level0, out, a, G, H, W, D);
    NND_LAUNCH_CHECK();
    return NND_OK;
}

int nnd_igev_lookup(const float* feat_pyramid, const float* geo_pyramid, const float* coords, float* out, int B, int G, int H,
                    int W, int num_levels, int radius, void* stream) {
    return igev_lookup_launch(feat_pyramid, geo_pyramid, coords, out, B, G, H, W, num_levels, radius, (hipStream_t)stream, false);
}

int nnd_corr1d_lookup(const float* pyramid, const float* coords, float* out, int B, int H, int W, int num_levels,
                      int radius, void* stream) {
    NND_REQUIRE(pyramid && coords && out, "corr1d_lookup: null pointer");
    NND_REQUIRE(B > 0 && H > 0 && W > 0 && radius >= 0, "corr1d_lookup: bad shape");
    return corr1d_lookup_launch(pyramid, coords, out, B, H, W, num_levels, radius, (hipStream_t)stream, false);
}

int nnd_convex_upsample(const float* flow, const float* mask, float* out, int B, int C, int H, int W, int rate,
                        void* stream) {
    NND_REQUIRE(flow && mask && out, "convex_upsample: null pointer");
    NND_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "convex_upsample: bad shape");
    return convex_upsample_launch(flow, mask, out, B, C, H, W, rate, (hipStream_t)stream, false);
}
}
