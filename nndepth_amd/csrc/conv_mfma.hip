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

constexpr int MAX_NE = 16;  // patch elements staged per thread per chunk (host checks)

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
    int log2_sc, tiles_x, S, ne;
    float scale;
};

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

template <int KH, int KW, int CI_T, int P>
__global__ void __launch_bounds__(256) conv_mfma_kernel(ConvArgs a) {
    constexpr int NT = KH * KW;
    constexpr int NQ = CI_T / 8;  // float4 A fragments per lane per (chunk, tap)
    extern __shared__ float lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nthreads = blockDim.x;
    const int h2 = lane >> 5, l31 = lane & 31;
    const int SC = 1 << a.log2_sc, SR = 32 >> a.log2_sc;
    const int r = l31 >> a.log2_sc, c = l31 & (SC - 1);
    const int tx0 = (blockIdx.x % a.tiles_x) * (P * SC);
    const int ty0 = (blockIdx.x / a.tiles_x) * SR;
    const int cb = blockIdx.y * (nthreads >> 6) + wave;
    const int b = blockIdx.z;
    const int H = a.H, W = a.W;
    const long HW = (long)H * W;
    const int PR = SR + KH - 1, PC = P * SC + KW - 1, S = a.S, PATCH = PR * S;
    constexpr int PH = KH / 2, PW = KW / 2;

    // ---- per-thread staging descriptors (identical for every chunk)
    int goff[MAX_NE], meta[MAX_NE];
    const int total = CI_T * PR * PC;
#pragma unroll
    for (int i = 0; i < MAX_NE; ++i) {
        int e = tid + i * nthreads;
        int ci = e / (PR * PC);
        int rem = e - ci * (PR * PC);
        int pr = rem / PC, pc = rem - pr * PC;
        int gy = ty0 + pr - PH, gx = tx0 + pc - PW;
        bool inimg = (e < total) && gy >= 0 && gy < H && gx >= 0 && gx < W;
        goff[i] = (int)(ci * HW + (long)gy * W + gx);
        int loff = ci * PATCH + pr * S + pc;
        meta[i] = (e < total) ? (loff | ((inimg ? ci : 127) << 24)) : -1;
    }

    f32x16 acc[P];
#pragma unroll
    for (int pp = 0; pp < P; ++pp)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[pp][i] = 0.f;

    const float4* wp = reinterpret_cast<const float4*>(a.wpk) + (size_t)cb * a.nchunks * (NT * NQ * 64) + lane;
    float4 a_cur[NT * NQ], a_nxt[NT * NQ];
    float stage[MAX_NE];

    auto load_x = [&](int k) {
        int cbase = k * CI_T;
        const float* src;
        int climit;
        if (cbase < a.c0) {
            src = a.src0 + b * a.bs0 + (long)cbase * HW;
            climit = a.c0 - cbase;
        } else {
            int cc = cbase - a.c0;
            src = a.src1 + b * a.bs1 + (long)cc * HW;
            climit = a.c1 - cc;
        }
        climit = climit < CI_T ? climit : CI_T;  // marker 127 (outside the image / not owned) never passes
#pragma unroll
        for (int i = 0; i < MAX_NE; ++i) {
            int ci = (meta[i] >> 24) & 127;
            stage[i] = (ci < climit) ? src[goff[i]] : 0.f;
        }
    };
    auto store_x = [&](int buf) {
        float* dst = lds + buf * (CI_T * PATCH);
#pragma unroll
        for (int i = 0; i < MAX_NE; ++i)
            if (meta[i] != -1) dst[meta[i] & 0xFFFFFF] = stage[i];
    };
    auto load_a = [&](float4* dstv, int k) {
#pragma unroll
        for (int t = 0; t < NT * NQ; ++t) dstv[t] = wp[(size_t)(k * (NT * NQ) + t) * 64];
    };

    load_a(a_cur, 0);
    load_x(0);
    store_x(0);
    __syncthreads();

    const int lane_base = h2 * PATCH + r * S + c;
    const int nchunks = a.nchunks;
    for (int k = 0; k < nchunks; ++k) {
        const bool more = (k + 1 < nchunks);
        if (more) {
            load_x(k + 1);
            load_a(a_nxt, k + 1);
        }
        const float* xb = lds + (k & 1) * (CI_T * PATCH) + lane_base;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int dy = t / KW, dx = t % KW;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const float4 av = a_cur[t * NQ + q];
                const float avs[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int pair = q * 4 + j;
#pragma unroll
                    for (int pp = 0; pp < P; ++pp) {
                        float bv = xb[(pair * 2) * PATCH + dy * S + dx + pp * SC];
                        acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(avs[j], bv, acc[pp], 0, 0, 0);
                    }
                }
            }
        }
        if (more) {
            store_x((k + 1) & 1);
#pragma unroll
            for (int t = 0; t < NT * NQ; ++t) a_cur[t] = a_nxt[t];
        }
        __syncthreads();
    }

    // ---- epilogue: lane holds pixel (y, x_pp) and 16 output channels
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
    int log2_sc, P, wco, tiles_x, tiles_y, S, ne;
};

static bool pick_tile(const ConvLayer& L, int B, int H, int W, TileCfg* out) {
    int wco = 1;
    for (int w : {4, 3, 2, 1})
        if (L.ncb % w == 0) {
            wco = w;
            break;
        }
    int force_sc = -1, force_p = -1;
    if (const char* e = getenv("NND_CONV_CFG")) sscanf(e, "%d,%d", &force_sc, &force_p);
    double best = 1e30;
    bool found = false;
    for (int log2_sc : {3, 4, 5, 2}) {
        for (int P : {1, 2, 3}) {
            if (force_sc >= 0 && log2_sc != force_sc) continue;
            if (force_p >= 0 && P != force_p) continue;
            int SC = 1 << log2_sc, SR = 32 >> log2_sc;
            int tx = cdiv(W, P * SC), ty = cdiv(H, SR);
            int PR = SR + L.KH - 1, PC = P * SC + L.KW - 1;
            int S = SC;  // smallest odd multiple of SC >= PC (bank-conflict-free B reads)
            while (S < PC) S += 2 * SC;
            int ne = cdiv(L.CI_T * PR * PC, 64 * wco);
            if (ne > MAX_NE) continue;
            size_t lds = (size_t)2 * L.CI_T * PR * S * sizeof(float);
            if (lds > 64 * 1024) continue;
            double waves = (double)tx * ty * B * L.ncb;
            double rounds = std::ceil(waves / 1024.0);
            double t = rounds * P;                       // MFMA streams on the busiest SIMD
            t *= 1.0 - 0.02 * (P - 1);                   // larger P: fewer weight bytes per flop
            t *= 1.0 + 0.01 * (5 - log2_sc);             // wider rows coalesce better
            if (t < best) {
                best = t;
                *out = {log2_sc, P, wco, tx, ty, S, ne};
                found = true;
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
    NND_REQUIRE(pick_tile(L, B, H, W, &cfg), "conv: no tile configuration for %dx%d Cin=%d", L.KH, L.KW, L.Cin);
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
    a.log2_sc = cfg.log2_sc; a.tiles_x = cfg.tiles_x; a.S = cfg.S; a.ne = cfg.ne;
    a.scale = io.scale;
    const int SR = 32 >> cfg.log2_sc;
    size_t lds = (size_t)2 * L.CI_T * (SR + L.KH - 1) * cfg.S * sizeof(float);
    dim3 grid(cfg.tiles_x * cfg.tiles_y, L.ncb / cfg.wco, B), block(64 * cfg.wco);
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
