// Implicit-GEMM 2-D convolution (stride 1, zero "same" padding) on the gfx950 fp32 MFMA.
//
//   out[b, co, y, x] = epilogue( bias[co] + sum_{ci,dy,dx} W[co,ci,dy,dx] * in[b, ci, y+dy-ph, x+dx-pw] )
//
// Mapping (D = A*B with v_mfma_f32_32x32x2_f32, exact fp32 fmaf chain):
//   A = weights   A[i = co (32 per wave)][k]      streamed global -> VGPR, pre-packed on the host in
//                                                 fragment order (one coalesced 1 KiB dwordx4 load per
//                                                 wave per (chunk, tap, 8 channels)); never touches LDS
//   B = activations B[k][j = pixel (32 per MFMA)] staged once per workgroup as a zero-filled halo
//                                                 patch in LDS and shared by all waves / all taps
//   D[i = co][j = pixel]: a lane holds ONE pixel and 16 output channels, so NCHW stores are
//   32 consecutive pixels per (register, half-wave) and the GRU gate math is a pure per-lane epilogue.
// The 32 pixels of an MFMA column block form an SR x SC sub-tile (SR*SC = 32; 4x8 tiles 68x120 exactly),
// each wave owns P such sub-tiles side by side (P accumulators), the waves of a workgroup own
// consecutive 32-channel output blocks and share the pixel tile.
// K is walked in chunks of CI_T input channels x all taps; the patch of chunk k+1 and the A fragments
// of chunk k+1 are prefetched into registers while chunk k is multiplied (one barrier per chunk).
//
// Replaces the nn.Conv2d calls of nndepth/blocks/update_block.py:57-65,26-36,97-112 and
// nndepth/blocks/gru.py:22-37,53-61 (reference files; semantics restated in oracle/torch_ref.py).
#include "common.h"

#include <cmath>
#include <cstdlib>
#include <cstring>

namespace nnd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int MAX_NE = 8;   // patch elements staged per thread per chunk (host checks)

struct ConvArgs {
    const float* src0;
    const float* src1;
    long bs0, bs1;
    int c0, c1;
    const float* wpk;
    const float* bias;
    float* out0;
    float* out1;
    long obs0, obs1;
    const float* aux0;
    const float* aux1;
    long abs0, abs1;
    int H, W, Cout, nchunks, epi, hidden;
    int log2_sc, tiles_x, S, wco, ks;
    float scale;
};

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

// Workgroup = wco x ks waves.  Wave (cbi, kj): output-channel block cb = blockIdx.y*wco + cbi, and K-slice kj:
// of every "super-chunk" of ks*CI_T input channels staged in LDS it multiplies channels [kj*CI_T, (kj+1)*CI_T).
// The ks partial accumulators of a tile are summed through LDS at the end (intra-workgroup split-K): this is
// what lets 255 pixel tiles x ncb channel blocks fill 1024 SIMDs evenly at batch 1.
template <int KH, int KW, int CI_T, int P>
__global__ void __launch_bounds__(512) conv_mfma_kernel(ConvArgs a) {
    constexpr int NT = KH * KW;
    constexpr int NQ = CI_T / 8;  // float4 A fragments per lane per (chunk, tap)
    extern __shared__ float lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nthreads = blockDim.x;
    const int wco = a.wco, ks = a.ks;
    const int cbi = wave % wco, kj = wave / wco;
    const int h2 = lane >> 5, l31 = lane & 31;
    const int SC = 1 << a.log2_sc, SR = 32 >> a.log2_sc;
    const int r = l31 >> a.log2_sc, c = l31 & (SC - 1);
    const int tx0 = (blockIdx.x % a.tiles_x) * (P * SC);
    const int ty0 = (blockIdx.x / a.tiles_x) * SR;
    const int cb = blockIdx.y * wco + cbi;
    const bool active = cb * 32 < a.Cout;  // trailing waves of the last workgroup only help staging
    const int b = blockIdx.z;
    const int H = a.H, W = a.W;
    const long HW = (long)H * W;
    const int PR = SR + KH - 1, PC = P * SC + KW - 1, S = a.S, PATCH = PR * S;
    constexpr int PH = KH / 2, PW = KW / 2;
    const int SCH = ks * CI_T;  // channels per super-chunk

    // ---- per-thread staging descriptors (identical for every super-chunk).  Loads are unconditional
    // (clamped to element 0 of the source when masked) so that hipcc counts vmcnt exactly.
    int goff[MAX_NE], loff[MAX_NE], cflag[MAX_NE];
    const int total = SCH * PR * PC;
#pragma unroll
    for (int i = 0; i < MAX_NE; ++i) {
        int e = tid + i * nthreads;
        int ci = e / (PR * PC);
        int rem = e - ci * (PR * PC);
        int pr = rem / PC, pc = rem - pr * PC;
        int gy = ty0 + pr - PH, gx = tx0 + pc - PW;
        bool inimg = (e < total) && gy >= 0 && gy < H && gx >= 0 && gx < W;
        goff[i] = inimg ? (int)(ci * HW + (long)gy * W + gx) : 0;
        loff[i] = (e < total) ? ci * PATCH + pr * S + pc : -1;
        cflag[i] = inimg ? ci : 0x7fff;  // 0x7fff: outside the image / not owned -> never < climit
    }

    f32x16 acc[P];
#pragma unroll
    for (int pp = 0; pp < P; ++pp)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[pp][i] = 0.f;

    const int nchunks = a.nchunks;
    const int nsuper = (nchunks + ks - 1) / ks;
    const float4* wp = reinterpret_cast<const float4*>(a.wpk) + (size_t)(active ? cb : 0) * nchunks * (NT * NQ * 64) + lane;
    float4 a0[NT * NQ], a1[NT * NQ];
    float stage[MAX_NE];

    auto chunk_src = [&](int K, const float*& src, int& climit) {
        int cbase = K * SCH;
        if (cbase < a.c0) {
            src = a.src0 + b * a.bs0 + (long)cbase * HW;
            climit = a.c0 - cbase;
        } else {
            int cc = cbase - a.c0;
            src = a.src1 + b * a.bs1 + (long)cc * HW;
            climit = a.c1 - cc;
        }
    };
    // raw loads only: the zero-fill select is applied in store_x, AFTER the chunk's MFMAs, so no
    // s_waitcnt lands between the prefetch and the multiply
    auto load_x = [&](int K) {
        const float* src;
        int climit;
        chunk_src(K, src, climit);
#pragma unroll
        for (int i = 0; i < MAX_NE; ++i) stage[i] = src[cflag[i] < climit ? goff[i] : 0];
    };
    auto store_x = [&](int K) {
        const float* src;
        int climit;
        chunk_src(K, src, climit);
        float* dst = lds + (K & 1) * (SCH * PATCH);
#pragma unroll
        for (int i = 0; i < MAX_NE; ++i)
            if (loff[i] >= 0) dst[loff[i]] = cflag[i] < climit ? stage[i] : 0.f;
    };
    // chunk index of this wave inside super-chunk K; waves past the last real chunk load chunk 0 (harmless)
    auto load_a = [&](float4* dstv, int K) {
        int ch = K * ks + kj;
        ch = ch < nchunks ? ch : 0;
#pragma unroll
        for (int t = 0; t < NT * NQ; ++t) dstv[t] = wp[(size_t)(ch * (NT * NQ) + t) * 64];
    };

    const int lane_base = kj * (CI_T * PATCH) + h2 * PATCH + r * S + c;

    // one super-chunk: prefetch K+1 (A fragments -> nxt, patch -> stage), multiply this wave's slice of K from LDS
    auto chunk = [&](int K, const float4* cur, float4* nxt) {
        const bool more = (K + 1 < nsuper);
        if (more) {
            load_a(nxt, K + 1);
            load_x(K + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (K * ks + kj < nchunks) {
            const float* xb = lds + (K & 1) * (SCH * PATCH) + lane_base;
            // B operands are read one tap ahead of the MFMAs that consume them (register double buffer),
            // so the LDS latency of tap t+1 hides under the 4*NQ*P MFMAs of tap t.
            float bq[2][NQ * 4 * P];
            auto read_tap = [&](int t, float* dst) {
                const int dy = t / KW, dx = t % KW;
#pragma unroll
                for (int pair = 0; pair < NQ * 4; ++pair)
#pragma unroll
                    for (int pp = 0; pp < P; ++pp) dst[pair * P + pp] = xb[(pair * 2) * PATCH + dy * S + dx + pp * SC];
            };
            read_tap(0, bq[0]);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (t + 1 < NT) read_tap(t + 1, bq[(t + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const float4 av = cur[t * NQ + q];
                    const float avs[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int pp = 0; pp < P; ++pp)
                            acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(avs[j], bq[t & 1][(q * 4 + j) * P + pp], acc[pp], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more) store_x(K + 1);
        __syncthreads();
    };

    load_a(a0, 0);
    load_x(0);
    store_x(0);
    __syncthreads();
    for (int K = 0; K < nsuper; K += 2) {
        chunk(K, a0, a1);
        if (K + 1 < nsuper) chunk(K + 1, a1, a0);
    }

    // ---- intra-workgroup split-K reduction through LDS (the patch buffers are free after the last barrier)
    if (ks > 1) {
        if (kj > 0 && active) {
            float* red = lds + (size_t)((cbi * (ks - 1) + (kj - 1)) * P) * 1024 + lane;
#pragma unroll
            for (int pp = 0; pp < P; ++pp)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) red[pp * 1024 + reg * 64] = acc[pp][reg];
        }
        __syncthreads();
        if (kj == 0 && active) {
            for (int j = 1; j < ks; ++j) {
                const float* red = lds + (size_t)((cbi * (ks - 1) + (j - 1)) * P) * 1024 + lane;
#pragma unroll
                for (int pp = 0; pp < P; ++pp)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) acc[pp][reg] += red[pp * 1024 + reg * 64];
            }
        }
    }

    // ---- epilogue: lane holds pixel (y, x_pp) and 16 output channels
    if (!active || kj != 0) return;
    const int y = ty0 + r;
    const int epi = a.epi;
#pragma unroll
    for (int pp = 0; pp < P; ++pp) {
        const int x = tx0 + pp * SC + c;
        if (y >= H || x >= W) continue;
        const long pix = (long)y * W + x;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int co = cb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h2;
            if (co >= a.Cout) continue;
            float v = acc[pp][reg] + a.bias[co];
            if (epi == EPI_RELU) {
                a.out0[b * a.obs0 + co * HW + pix] = fmaxf(v, 0.f);
            } else if (epi == EPI_LINEAR) {
                a.out0[b * a.obs0 + co * HW + pix] = v;
            } else if (epi == EPI_SCALE) {
                a.out0[b * a.obs0 + co * HW + pix] = a.scale * v;
            } else if (epi == EPI_GRU_ZR) {
                float s = sigmoidf_(v);
                if (co < a.hidden) {
                    a.out0[b * a.obs0 + co * HW + pix] = s;
                } else {
                    int cc = co - a.hidden;
                    a.out1[b * a.obs1 + cc * HW + pix] = s * a.aux0[b * a.abs0 + cc * HW + pix];
                }
            } else {  // EPI_GRU_Q
                float q = tanhf(v);
                float hprev = a.aux0[b * a.abs0 + co * HW + pix];
                float z = a.aux1[b * a.abs1 + co * HW + pix];
                a.out0[b * a.obs0 + co * HW + pix] = (1.0f - z) * hprev + z * q;
            }
        }
    }
}

// --------------------------------------------------------------------------- host side
struct TileCfg {
    int log2_sc, P, wco, ks, tiles_x, tiles_y, S;
    size_t lds;
};

// Chooses the pixel sub-tile shape (SR x SC), P sub-tiles per wave, wco x ks waves per workgroup.
// Cost model: every wave issues unit = ceil(nchunks/ks) * P MFMA streams; waves spread evenly over the
// 1024 SIMDs of the chip, so the busiest SIMD runs ceil(waves/1024) * unit (quantisation is what matters
// at batch 1: 68x120 = 255 tiles of 4x8 pixels).
static bool pick_tile(const ConvLayer& L, int c0, int c1, int B, int H, int W, TileCfg* out) {
    int force_sc = -1, force_p = -1, force_ks = -1;
    if (const char* e = getenv("NND_CONV_CFG")) sscanf(e, "%d,%d,%d", &force_sc, &force_p, &force_ks);
    double best = 1e30;
    bool found = false;
    for (int ks : {1, 2, 4}) {
        if (force_ks > 0 && ks != force_ks) continue;
        if (ks > L.nchunks) continue;
        if (c1 > 0 && c0 % (ks * L.CI_T) != 0) continue;
        // waves per workgroup along Cout: fewest total wave slots, ties -> more sharing of the patch
        int wco = 1, best_slots = 1 << 30;
        for (int w : {4, 3, 2, 1}) {
            if (w * ks > 8) continue;
            int slots = cdiv(L.ncb, w) * w;
            if (slots < best_slots) {
                best_slots = slots;
                wco = w;
            }
        }
        if (L.ncb == 1 && ks == 1) wco = 2;  // one idle wave helps stage the patch
        for (int log2_sc : {3, 4, 5, 2}) {
            for (int P : {1, 2, 3}) {
                if (force_sc >= 0 && log2_sc != force_sc) continue;
                if (force_p > 0 && P != force_p) continue;
                int SC = 1 << log2_sc, SR = 32 >> log2_sc;
                int tx = cdiv(W, P * SC), ty = cdiv(H, SR);
                int PR = SR + L.KH - 1, PC = P * SC + L.KW - 1;
                int S = SC;  // smallest odd multiple of SC >= PC (bank-conflict-free B reads)
                while (S < PC) S += 2 * SC;
                if (cdiv(ks * L.CI_T * PR * PC, 64 * wco * ks) > MAX_NE) continue;
                size_t lds = (size_t)2 * ks * L.CI_T * PR * S * sizeof(float);
                size_t red = (size_t)wco * (ks - 1) * P * 1024 * sizeof(float);
                if (red > lds) lds = red;
                if (lds > 64 * 1024) continue;
                const int occ = P == 1 ? 4 : 2;  // resident waves per SIMD the register budget allows
                double waves = (double)tx * ty * B * cdiv(L.ncb, wco) * wco * ks;
                double unit = (double)cdiv(L.nchunks, ks) * P;
                double per_simd = std::ceil(waves / 1024.0);
                double t = (per_simd <= occ) ? per_simd * unit : std::ceil(waves / (1024.0 * occ)) * occ * unit;
                if (waves <= 1024.0) t *= 1.10;             // one wave per SIMD cannot hide its own stalls
                t *= 1.0 - 0.02 * (P - 1);                   // larger P: fewer weight bytes per flop
                t *= 1.0 + 0.01 * (5 - log2_sc);             // wider rows coalesce better
                t *= 1.0 + 0.01 * (ks - 1);                  // reduction cost
                if (t < best) {
                    best = t;
                    *out = {log2_sc, P, wco, ks, tx, ty, S, lds};
                    found = true;
                }
            }
        }
    }
    return found;
}

#define NND_CONV_CASE(KH_, KW_, CI_)                                                              \
    if (L.KH == KH_ && L.KW == KW_ && L.CI_T == CI_) {                                            \
        if (cfg.P == 1)                                                                           \
            hipLaunchKernelGGL((conv_mfma_kernel<KH_, KW_, CI_, 1>), grid, block, lds, stream, a); \
        else if (cfg.P == 2)                                                                      \
            hipLaunchKernelGGL((conv_mfma_kernel<KH_, KW_, CI_, 2>), grid, block, lds, stream, a); \
        else                                                                                      \
            hipLaunchKernelGGL((conv_mfma_kernel<KH_, KW_, CI_, 3>), grid, block, lds, stream, a); \
        launched = true;                                                                          \
    }

int launch_conv(const ConvLayer& L, const float* blob, const ConvIO& io, int epi, int B, int H, int W,
                hipStream_t stream) {
    NND_REQUIRE(io.src0.C + io.src1.C == L.Cin, "conv: source channels %d+%d != Cin %d", io.src0.C, io.src1.C, L.Cin);
    NND_REQUIRE(io.src1.C == 0 || io.src0.C % L.CI_T == 0, "conv: first source (%d ch) must be a multiple of %d", io.src0.C, L.CI_T);
    NND_REQUIRE((long)L.Cin * H * W < (1L << 31), "conv: plane offsets exceed 32 bits");
    TileCfg cfg;
    NND_REQUIRE(pick_tile(L, io.src0.C, io.src1.C, B, H, W, &cfg), "conv: no tile configuration for %dx%d Cin=%d", L.KH, L.KW, L.Cin);
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
    a.H = H; a.W = W; a.Cout = L.Cout; a.nchunks = L.nchunks; a.epi = epi; a.hidden = io.hidden;
    a.log2_sc = cfg.log2_sc; a.tiles_x = cfg.tiles_x; a.S = cfg.S; a.wco = cfg.wco; a.ks = cfg.ks;
    a.scale = io.scale;
    const size_t lds = cfg.lds;
    dim3 grid(cfg.tiles_x * cfg.tiles_y, cdiv(L.ncb, cfg.wco), B), block(64 * cfg.wco * cfg.ks);
    if (getenv("NND_CONV_VERBOSE"))
        fprintf(stderr, "[nnd] conv %dx%d Cin=%d Cout=%d: tile 2^%d cols, P=%d, wco=%d, ks=%d, grid %ux%ux%u, lds %zu B\n", L.KH,
                L.KW, L.Cin, L.Cout, cfg.log2_sc, cfg.P, cfg.wco, cfg.ks, grid.x, grid.y, grid.z, lds);
    bool launched = false;
    NND_CONV_CASE(1, 1, 8)
    NND_CONV_CASE(1, 1, 32)
    NND_CONV_CASE(3, 3, 8)
    NND_CONV_CASE(1, 5, 8)
    NND_CONV_CASE(5, 1, 8)
    if (!launched) {
        set_error("conv %dx%d CI_T=%d not instantiated", L.KH, L.KW, L.CI_T);
        return NND_ERR_UNSUPPORTED;
    }
    NND_LAUNCH_CHECK();
    return NND_OK;
}

void pack_conv(const ConvLayer& L, int nparts, const float* const* w, const float* const* bvec,
               const int* cout, float* blob) {
    const int NT = L.KH * L.KW, NQ = L.CI_T / 8;
    float* wp = blob + L.w_off;
    float* bp = blob + L.b_off;
    memset(wp, 0, sizeof(float) * L.w_floats());
    memset(bp, 0, sizeof(float) * L.b_floats());
    int co0 = 0;
    for (int part = 0; part < nparts; ++part) {
        for (int col = 0; col < cout[part]; ++col) {
            int co = co0 + col;
            int cb = co / 32, i = co % 32;
            bp[co] = bvec[part][col];
            for (int ci = 0; ci < L.Cin; ++ci) {
                int chunk = ci / L.CI_T, cl = ci % L.CI_T;
                int pair = cl / 2, h2 = cl % 2;
                int q = pair / 4, j = pair % 4;
                int lane = h2 * 32 + i;
                for (int t = 0; t < NT; ++t) {
                    size_t idx = (((((size_t)cb * L.nchunks + chunk) * NT + t) * NQ + q) * 64 + lane) * 4 + j;
                    wp[idx] = w[part][((size_t)col * L.Cin + ci) * NT + t];
                }
            }
        }
        co0 += cout[part];
    }
}

}  // namespace nnd
