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
#include "common.h"

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
__global__ void __launch_bounds__(256) corr1d_build_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                           float* __restrict__ pyr, PyrLayout L, int C, int H, int W,
                                                           float rscale_div) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, h2 = lane >> 5;
    const int w1_0 = blockIdx.x * 32, h = blockIdx.y, b = blockIdx.z;
    const long HW = (long)H * W;
    const float* a_base = f1 + ((long)b * C) * HW + (long)h * W;
    const float* b_base = f2 + ((long)b * C) * HW + (long)h * W;
    const int w1 = w1_0 + l31;
    const bool a_ok = w1 < W;
    const int ntile = (W + 31) / 32;
    for (int t = wave; t < ntile; t += 4) {
        const int w2 = t * 32 + l31;
        const bool b_ok = w2 < W;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll 8
        for (int c = 0; c < C; c += 2) {
            const int cc = c + h2;
            float av = (a_ok && cc < C) ? a_base[cc * HW + w1] : 0.f;
            float bv = (b_ok && cc < C) ? b_base[cc * HW + w2] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
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

struct LookupArgs {
    PyrLayout L;
    int B, H, W, num_levels, radius;
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
    float x = (float)(k - a.radius) + coords[(long)b * HW + pix] / (float)(1 << lvl);
    const float wm1 = (float)(w2 - 1);
    x = x / wm1;
    x = fminf(fmaxf(x, 0.f), 1.f);
    x = x * wm1;
    const float f0 = floorf(x), f1 = ceilf(x);
    const float v0 = row[(int)f0], v1 = row[(int)f1];
    const float coef = f1 - x;
    out[idx] = coef * v0 + (1.0f - coef) * v1;
}

// out[b, c, r*h + i, r*w + j] = sum_k softmax_k(mask[b, k*r*r + i*r + j, h, w]) * (r * flow)[b, c, h+ky-1, w+kx-1]
// thread = (b, h, i, w); loops j (r consecutive outputs -> contiguous store) and c.
template <int RATE>
__global__ void __launch_bounds__(256) convex_upsample_kernel(const float* __restrict__ flow, const float* __restrict__ mask,
                                                              float* __restrict__ out, int B, int C, int H, int W) {
    const long total = (long)B * H * RATE * W;
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int w = (int)(idx % W);
    const int i = (int)((idx / W) % RATE);
    const int h = (int)((idx / ((long)W * RATE)) % H);
    const int b = (int)(idx / ((long)W * RATE * H));
    const long HW = (long)H * W;
    const float* mb = mask + (long)b * 9 * RATE * RATE * HW + (long)h * W + w;
    for (int c = 0; c < C; ++c) {
        float nb[9];
        const float* fb = flow + ((long)b * C + c) * HW;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            int yy = h + k / 3 - 1, xx = w + k % 3 - 1;
            nb[k] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? (float)RATE * fb[(long)yy * W + xx] : 0.f;
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
                         int radius, hipStream_t stream) {
    NND_REQUIRE(num_levels >= 1 && num_levels < MAX_LEVELS, "lookup: num_levels %d out of range", num_levels);
    LookupArgs a;
    make_layout(B, H, W, num_levels + 1, &a.L, nullptr);
    a.B = B; a.H = H; a.W = W; a.num_levels = num_levels; a.radius = radius;
    NND_REQUIRE(a.L.width[num_levels - 1] >= 2, "lookup: level %d has width %d < 2", num_levels - 1, a.L.width[num_levels - 1]);
    long total = (long)B * num_levels * (2 * radius + 1) * H * W;
    hipLaunchKernelGGL(corr1d_lookup_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, stream, pyr, coords, out, a);
    NND_LAUNCH_CHECK();
    return NND_OK;
}

int convex_upsample_launch(const float* flow, const float* mask, float* out, int B, int C, int H, int W, int rate,
                           hipStream_t stream) {
    long total = (long)B * H * rate * W;
    dim3 grid((unsigned)cdiv64(total, 256)), block(256);
    if (rate == 8)
        hipLaunchKernelGGL(convex_upsample_kernel<8>, grid, block, 0, stream, flow, mask, out, B, C, H, W);
    else if (rate == 4)
        hipLaunchKernelGGL(convex_upsample_kernel<4>, grid, block, 0, stream, flow, mask, out, B, C, H, W);
    else if (rate == 2)
        hipLaunchKernelGGL(convex_upsample_kernel<2>, grid, block, 0, stream, flow, mask, out, B, C, H, W);
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
    hipLaunchKernelGGL(corr1d_build_kernel, grid, block, 0, (hipStream_t)stream, fmap1, fmap2, pyramid, L, C, H, W, div);
    NND_LAUNCH_CHECK();
    return NND_OK;
}

int nnd_corr1d_lookup(const float* pyramid, const float* coords, float* out, int B, int H, int W, int num_levels,
                      int radius, void* stream) {
    NND_REQUIRE(pyramid && coords && out, "corr1d_lookup: null pointer");
    NND_REQUIRE(B > 0 && H > 0 && W > 0 && radius >= 0, "corr1d_lookup: bad shape");
    return corr1d_lookup_launch(pyramid, coords, out, B, H, W, num_levels, radius, (hipStream_t)stream);
}

int nnd_convex_upsample(const float* flow, const float* mask, float* out, int B, int C, int H, int W, int rate,
                        void* stream) {
    NND_REQUIRE(flow && mask && out, "convex_upsample: null pointer");
    NND_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "convex_upsample: bad shape");
    return convex_upsample_launch(flow, mask, out, B, C, H, W, rate, (hipStream_t)stream);
}
}
