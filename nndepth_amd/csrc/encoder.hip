// Feature encoder of RAFT-Stereo on the MFMA convolution (the row next to the refinement loop).
//
// Replaces nndepth/encoders/basic_encoder.py:71-93 (BasicEncoder, norm_fn = "batch" in eval mode or "none"),
// nndepth/blocks/residual_block.py:53-60 (ResidualBlock: the 1x1 projection shortcut is ALWAYS applied, Q3) and the
// cnet_proj conv of nndepth/models/raft_stereo/model.py:53-55 (reference; restated in oracle/torch_ref.py:
// basic_encoder / residual_block).
//
//   conv1 7x7 s2 (3 -> 64) + norm + ReLU            stem_kernel (its own implicit GEMM on the MFMA: K = 147 is too ragged for conv_mfma)
//   6 residual blocks                                conv_mfma, 3 launches per block:
//       y  = ReLU(norm1(conv1 3x3 (stride s)(x)))        EPI_AFFINE | relu
//       sc = norm3(conv 1x1 (stride s)(x))               EPI_AFFINE
//       x' = ReLU(sc + ReLU(norm2(conv2 3x3(y))))        EPI_AFFINE | relu | residual | relu
//   conv2 1x1 (128 -> output_dim)                    conv_mfma, writes the NCHW result
//   cnet_proj 3x3 (output_dim -> ctx + hid) + ReLU   conv_mfma on the first half of the batch (the left frames)
//
// norm = 2 (InstanceNorm2d(affine=False), CREStereo's encoder, cre_stereo/model.py:70-72): the statistics depend on the
// sample, so every conv writes its raw output (+ bias), `in_stats_kernel` reduces each (sample, channel) plane to
// (1/sqrt(var + eps), -mean/sqrt(var + eps)) in double precision, and `in_apply_kernel` normalises in place with the ReLU
// — for the block's last conv together with the normalised shortcut, the add and the final ReLU (one pass instead of four).
//
// Eval-mode BatchNorm is folded at pack time (host, double precision) into a per-channel scale and shift applied in the
// conv epilogue: y = acc * (gamma / sqrt(var + eps)) + ((bias - mean) * gamma / sqrt(var + eps) + beta).
// Intermediate activations live in the tile-major workspace layout (layout.h); the input frames and the outputs are NCHW.
#include "common.h"
#include "layout.h"

#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace nnd {

// ------------------------------------------------------------------------------------------ generic conv + norm
// arith != 0 asks for the split-bf16 MFMA kernel (conv_split.hip); shapes it does not build fall back to the exact fp32 kernel
static int conv_layer(const nnd_conv_desc* d, ConvLayer* L, int64_t* total, int arith = 0) {
    NND_REQUIRE(d, "conv: null descriptor");
    NND_REQUIRE(d->Cout > 0 && d->Cin > 0, "conv: bad channel counts");
    NND_REQUIRE(d->stride == 1 || d->stride == 2, "conv: stride %d not supported (1, 2)", d->stride);
    const bool k11 = d->KH == 1 && d->KW == 1, k33 = d->KH == 3 && d->KW == 3;
    if (d->stride == 2)
        NND_REQUIRE(k11 || k33, "conv: stride 2 is built for 1x1 and 3x3 kernels");
    else
        NND_REQUIRE(k11 || k33 || (d->KH == 1 && d->KW == 5) || (d->KH == 5 && d->KW == 1),
                    "conv: kernel %dx%d not built (1x1, 3x3, 1x5, 5x1)", d->KH, d->KW);
    ConvLayer l;
    l.KH = d->KH; l.KW = d->KW; l.Cin = d->Cin; l.Cout = d->Cout; l.stride = d->stride;
    // the stride-1 1x1 projections stay on the streaming fp32 kernel (HBM-bound); 3x3 layers (stride 1 and 2) and the stride-2
    // 1x1 shortcuts may take the split kernel
    if (arith != 0 && ((k11 && d->stride == 1) || !conv_split_supported(d->KH, d->KW, d->Cin, d->stride, arith, d->Cout))) arith = 0;
    l.arith = arith;
    l.CI_T = arith ? 16 : conv_ci_t(d->KH, d->KW, d->Cin, d->stride, d->Cout);
    l.nchunks = cdiv(d->Cin, l.CI_T);
    l.ncb = cdiv(d->Cout, 32);
    int64_t off = 0;
    l.w_off = off; off += l.w_floats();
    l.b_off = off; off += l.b_floats();
    l.s_off = off; off += l.b_floats();
    *L = l;
    if (total) *total = off;
    return NND_OK;
}

// packs `w` (Cout,Cin,KH,KW) / bias and the folded norm into blob + base (layer offsets are relative to base)
static void pack_conv_norm(const ConvLayer& L, const float* w, const float* bias, const float* gamma, const float* beta,
                           const float* mean, const float* var, float eps, float* base) {
    const float* ws[1] = {w};
    const float* bs[1] = {nullptr};
    int co[1] = {L.Cout};
    pack_conv(L, 1, ws, bs, co, base);
    float* shift = base + L.b_off;
    float* scale = base + L.s_off;
    for (int c = 0; c < L.ncb * 32; ++c) {
        double sc = 1.0, sh = 0.0;
        if (c < L.Cout) {
            const double b = bias ? (double)bias[c] : 0.0;
            if (gamma) {
                sc = (double)gamma[c] / std::sqrt((double)var[c] + (double)eps);
                sh = (b - (double)mean[c]) * sc + (double)beta[c];
            } else {
                sh = b;
            }
        }
        scale[c] = (float)sc;
        shift[c] = (float)sh;
    }
}

// (x_c4 / y_c4: tile-major tensors with 4 channels interleaved, layout.h; the residual follows y's layout)
static int run_conv_norm(const ConvLayer& L, const float* base, const float* x, int64_t xbs, bool x_tiled, const float* res,
                         int64_t rbs, float* y, int64_t ybs, bool y_tiled, int flags, int B, int Hin, int Win, hipStream_t s,
                         bool x_c4 = false, bool y_c4 = false) {
    const int H = (Hin + L.stride - 1) / L.stride, W = (Win + L.stride - 1) / L.stride;
    ConvIO io{};
    io.src0 = Act{const_cast<float*>(x), xbs, L.Cin};
    io.out0 = Act{y, ybs, L.Cout};
    if (res) io.aux0 = Act{const_cast<float*>(res), rbs, L.Cout};
    io.Hin = Hin; io.Win = Win;
    io.flags = flags;
    io.src_tiled = x_tiled;
    io.dst_tiled = y_tiled;
    io.src_c4 = x_tiled && x_c4;
    io.dst_c4 = y_tiled && y_c4;
    return launch_conv(L, base, io, EPI_AFFINE, B, H, W, s);
}

// ------------------------------------------------------------------------------------------ stem: 7x7 stride 2, 3 -> 64
// Implicit GEMM on the fp32 MFMA with K = 3*7*7 = 147 (+1 zero row): one workgroup = an 8x32 output region (8 sub-tiles of
// 4x8 pixels, tile-major stores of two 128-B lines per wave and channel) x all 64 output channels; wave w multiplies
// sub-tiles 2w, 2w+1 against both 32-channel blocks (4 accumulators).  The 3 x 21 x 69 input patch and the weights,
// pre-transposed on the host to [k][co], sit in LDS: the A operand a[k][co] is a conflict-free row read, the B operand of
// tap k = (c, dy, dx) for lane (r, cc) is patch[c][2r + dy][2cc + dx] — lane stride 2 / 144 floats, 32 distinct banks.
// (The VALU version this replaces kept the 147 taps in registers and streamed the weights through the scalar cache:
// 144 us for 2 x 544x960 frames; K is too ragged for conv_mfma's chunked staging.)
// grid (regions, 1, N)
constexpr int STEM_K = 148;  // 147 taps + one zero row
typedef float stem_f32x16 __attribute__((ext_vector_type(16)));
// (x1 / nsplit: samples nsplit .. N-1 are read from x1 — the right frames of a pair batch — so the caller need not concatenate)
__global__ void __launch_bounds__(256) stem_kernel(const float* __restrict__ x, const float* __restrict__ x1, int nsplit, const float* __restrict__ w,
                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                   float* __restrict__ out, long obs, int Hin, int Win, int H, int W,
                                                   int tiles_x, Lay lay, int relu) {
    __shared__ float patch[3 * 21 * 72];
    __shared__ float wl[STEM_K * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h2 = lane >> 5, l31 = lane & 31;
    const int tx0 = (blockIdx.x % tiles_x) * 32, ty0 = (blockIdx.x / tiles_x) * 8;
    const int n = blockIdx.z;
    const long HWin = (long)Hin * Win;
    const float* src = n < nsplit ? x + (long)n * 3 * HWin : x1 + (long)(n - nsplit) * 3 * HWin;
    for (int e = tid; e < 3 * 21 * 69; e += 256) {
        const int c = e / (21 * 69), rem = e % (21 * 69);
        const int pr = rem / 69, pc = rem % 69;
        const int gy = ty0 * 2 + pr - 3, gx = tx0 * 2 + pc - 3;
        patch[(c * 21 + pr) * 72 + pc] = (gy >= 0 && gy < Hin && gx >= 0 && gx < Win) ? src[c * HWin + (long)gy * Win + gx] : 0.f;
    }
    for (int e = tid; e < STEM_K * 16; e += 256) reinterpret_cast<float4*>(wl)[e] = reinterpret_cast<const float4*>(w)[e];
    __syncthreads();
    stem_f32x16 acc[2][2];  // [sub-tile of the wave][channel block]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // lane base of the B operand for the wave's two sub-tiles: st = 2*wave + i -> rows (st>>2)*4 + r, cols (st&3)*8 + cc
    const int r = l31 >> 3, cc = l31 & 7;
    int base[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int st = 2 * wave + i;
        base[i] = (2 * ((st >> 2) * 4 + r)) * 72 + 2 * ((st & 3) * 8 + cc);
    }
#pragma unroll
    for (int kp = 0; kp < STEM_K / 2; ++kp) {
        // tap offsets of k = 2kp and 2kp + 1 inside the patch (compile-time), selected by the lane's k parity
        const int k0 = 2 * kp, k1 = 2 * kp + 1;
        const int o0 = (k0 / 49 * 21 + (k0 % 49) / 7) * 72 + k0 % 7;
        const int o1 = k1 < 147 ? (k1 / 49 * 21 + (k1 % 49) / 7) * 72 + k1 % 7 : 0;  // k = 147: zero weight row, any address
        const int off = h2 ? o1 : o0;
        const float a0 = wl[(2 * kp + h2) * 64 + l31], a1 = wl[(2 * kp + h2) * 64 + 32 + l31];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float bv = patch[base[i] + off];
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv, acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv, acc[i][1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int st = 2 * wave + i;
        const int y = ty0 + (st >> 2) * 4 + r, xx = tx0 + (st & 3) * 8 + cc;
        if (y >= H || xx >= W) continue;
        float* o = out + (long)n * obs + pix_off(lay, y, xx);
        if (lay.ci == 4) {  // 4 channels interleaved: registers 4q..4q+3 of a block are one 16-B store
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int co0 = j * 32 + 8 * q + 4 * h2;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float t = fmaf(acc[i][j][4 * q + e], scale[co0 + e], shift[co0 + e]);
                        v[e] = relu ? fmaxf(t, 0.f) : t;
                    }
                    *reinterpret_cast<float4*>(o + (long)co0 * lay.plane) = make_float4(v[0], v[1], v[2], v[3]);
                }
            continue;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int co = j * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h2;
                const float v = fmaf(acc[i][j][reg], scale[co], shift[co]);
                o[(long)co * lay.plane] = relu ? fmaxf(v, 0.f) : v;
            }
    }
}

// ------------------------------------------------------------------------------------------ instance norm
// stats[(n*C + c)*2 + {0,1}] = (alpha, beta) with alpha = 1/sqrt(var + eps), beta = -mean*alpha (biased variance, like
// F.instance_norm).  Two deterministic stages with double-precision sums: IN_CHUNKS workgroups per (channel, sample)
// plane of the tile-major tensor write partial (sum, sum of squares); one thread per plane combines them in order.
constexpr int IN_CHUNKS = 32;
__global__ void __launch_bounds__(256) in_partial_kernel(const float* __restrict__ x, long bs, int C, int H, int W, Lay lay,
                                                         double* __restrict__ partial) {
    __shared__ double sh[2][256];
    const int k = blockIdx.x, c = blockIdx.y, n = blockIdx.z;
    const float* p = x + (long)n * bs + (long)c * lay.plane;
    double s = 0.0, q = 0.0;
    // the plane as it lies in memory (tile-major: 4x8-pixel tiles of 32 floats), 16 bytes per thread and step: group g = 4
    // consecutive x of row (g % 8) / 2 of tile g / 8; the padding of the edge tiles holds no data and is skipped
    // (first version: one pixel per thread through pix_off(y, x): 2.1 TB/s on the 540x960 maps of CREStereo's first stage)
    const long ngroups = lay.plane / 4;
    for (long g = (long)k * 256 + threadIdx.x; g < ngroups; g += IN_CHUNKS * 256) {
        const long t = g >> 3;
        const int y = (int)(t / lay.TX) * 4 + (int)((g & 7) >> 1), x0 = (int)(t % lay.TX) * 8 + (int)(g & 1) * 4;
        if (y >= H || x0 >= W) continue;
        const float4 v4 = *reinterpret_cast<const float4*>(p + g * 4);
        const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (x0 + e < W) {
                const double v = (double)vv[e];
                s += v;
                q += v * v;
            }
    }
    sh[0][threadIdx.x] = s;
    sh[1][threadIdx.x] = q;
    __syncthreads();
    for (int j = 128; j > 0; j >>= 1) {
        if ((int)threadIdx.x < j) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + j];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + j];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double* o = partial + (((long)n * C + c) * IN_CHUNKS + k) * 2;
        o[0] = sh[0][0];
        o[1] = sh[1][0];
    }
}

// The same for a tensor with 4 channels interleaved (layout.h): workgroup = (chunk, channel quad, sample); a thread visits the same
// pixel groups in the same order as above and keeps four pairs of sums, so every channel's partials are bit for bit those of the
// planar kernel.
__global__ void __launch_bounds__(256) in_partial_c4_kernel(const float* __restrict__ x, long bs, int C, int H, int W, Lay lay,
                                                            double* __restrict__ partial) {
    __shared__ double sh[2][4][256];
    const int k = blockIdx.x, cq = blockIdx.y, n = blockIdx.z;
    const float* p = x + (long)n * bs + (long)cq * 4 * lay.plane;
    double s[4] = {0.0, 0.0, 0.0, 0.0}, q[4] = {0.0, 0.0, 0.0, 0.0};
    const long ngroups = lay.plane / 4;
    for (long g = (long)k * 256 + threadIdx.x; g < ngroups; g += IN_CHUNKS * 256) {
        const long t = g >> 3;
        const int y = (int)(t / lay.TX) * 4 + (int)((g & 7) >> 1), x0 = (int)(t % lay.TX) * 8 + (int)(g & 1) * 4;
        if (y >= H || x0 >= W) continue;
        float4 v4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v4[e] = *reinterpret_cast<const float4*>(p + (g * 4 + e) * 4);  // pixel 4g + e: its 4 channels
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (x0 + e < W) {
                const float vv[4] = {v4[e].x, v4[e].y, v4[e].z, v4[e].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const double v = (double)vv[j];
                    s[j] += v;
                    q[j] += v * v;
                }
            }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        sh[0][j][threadIdx.x] = s[j];
        sh[1][j][threadIdx.x] = q[j];
    }
    __syncthreads();
    for (int r = 128; r > 0; r >>= 1) {
        if ((int)threadIdx.x < r)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sh[0][j][threadIdx.x] += sh[0][j][threadIdx.x + r];
                sh[1][j][threadIdx.x] += sh[1][j][threadIdx.x + r];
            }
        __syncthreads();
    }
    if (threadIdx.x < 4) {
        double* o = partial + (((long)n * C + cq * 4 + threadIdx.x) * IN_CHUNKS + k) * 2;
        o[0] = sh[0][threadIdx.x][0];
        o[1] = sh[1][threadIdx.x][0];
    }
}

__global__ void __launch_bounds__(256) in_final_kernel(const double* __restrict__ partial, int NC, int HW, float eps,
                                                       float* __restrict__ stats) {
    const int i = blockIdx.x * 256 + threadIdx.x;  // plane index n*C + c
    if (i >= NC) return;
    double s = 0.0, q = 0.0;
    for (int k = 0; k < IN_CHUNKS; ++k) {
        s += partial[((long)i * IN_CHUNKS + k) * 2];
        q += partial[((long)i * IN_CHUNKS + k) * 2 + 1];
    }
    const double mean = s / HW, var = fmax(q / HW - mean * mean, 0.0);
    const double alpha = 1.0 / sqrt(var + (double)eps);
    stats[(long)i * 2] = (float)alpha;
    stats[(long)i * 2 + 1] = (float)(-mean * alpha);
}

// y = relu(x*alpha + beta) in place; with a shortcut: y = relu((sc*alpha_s + beta_s) + relu(x*alpha + beta)).
// One thread per V consecutive pixels of a (channel, sample) plane (tile-major incl. its padding: harmless, never read as data);
// V = 4: 16-byte loads and stores (planes are multiples of 32 floats; the launcher checks the base alignment), V = 1 otherwise.
template <int V>
__global__ void __launch_bounds__(256) in_apply_kernel(float* __restrict__ x, long bs, const float* __restrict__ stats,
                                                       const float* __restrict__ sc, long sbs, const float* __restrict__ sstats,
                                                       int C, long plane, int relu) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * V;
    if (i >= plane) return;
    const int c = blockIdx.y, n = blockIdx.z;
    const float a = stats[((long)n * C + c) * 2], b = stats[((long)n * C + c) * 2 + 1];
    float* p = x + (long)n * bs + (long)c * plane + i;
    float v[V], sv[V];
    if (V == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1 % V] = t.y; v[2 % V] = t.z; v[3 % V] = t.w;
    } else {
        v[0] = *p;
    }
    float as = 0.f, bsft = 0.f;
    if (sc) {
        as = sstats[((long)n * C + c) * 2];
        bsft = sstats[((long)n * C + c) * 2 + 1];
        const float* q = sc + (long)n * sbs + (long)c * plane + i;
        if (V == 4) {
            const float4 t = *reinterpret_cast<const float4*>(q);
            sv[0] = t.x; sv[1 % V] = t.y; sv[2 % V] = t.z; sv[3 % V] = t.w;
        } else {
            sv[0] = *q;
        }
    }
#pragma unroll
    for (int k = 0; k < V; ++k) {
        float r = fmaf(v[k], a, b);
        if (relu) r = fmaxf(r, 0.f);
        if (sc) r = fmaxf(fmaf(sv[k], as, bsft) + r, 0.f);
        v[k] = r;
    }
    if (V == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1 % V], v[2 % V], v[3 % V]);
    else *p = v[0];
}

// The same for 4 channels interleaved: one thread = the 4 channels of one pixel, each with its own (alpha, beta)
__global__ void __launch_bounds__(256) in_apply_c4_kernel(float* __restrict__ x, long bs, const float* __restrict__ stats,
                                                          const float* __restrict__ sc, long sbs, const float* __restrict__ sstats,
                                                          int C, long plane, int relu) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;  // pixel slot of the tile-major plane (padding included: never read as data)
    if (i >= plane) return;
    const int cq = blockIdx.y, n = blockIdx.z;
    const float4 st0 = *reinterpret_cast<const float4*>(stats + ((long)n * C + cq * 4) * 2);      // (a, b) of channels 0, 1
    const float4 st1 = *reinterpret_cast<const float4*>(stats + ((long)n * C + cq * 4 + 2) * 2);  // ... 2, 3
    const float a[4] = {st0.x, st0.z, st1.x, st1.z}, b[4] = {st0.y, st0.w, st1.y, st1.w};
    float* p = x + (long)n * bs + (long)cq * 4 * plane + i * 4;
    const float4 t = *reinterpret_cast<const float4*>(p);
    float v[4] = {t.x, t.y, t.z, t.w}, sv[4] = {0.f, 0.f, 0.f, 0.f}, as[4] = {0.f, 0.f, 0.f, 0.f}, bsft[4] = {0.f, 0.f, 0.f, 0.f};
    if (sc) {
        const float4 s0 = *reinterpret_cast<const float4*>(sstats + ((long)n * C + cq * 4) * 2);
        const float4 s1 = *reinterpret_cast<const float4*>(sstats + ((long)n * C + cq * 4 + 2) * 2);
        as[0] = s0.x; as[1] = s0.z; as[2] = s1.x; as[3] = s1.z;
        bsft[0] = s0.y; bsft[1] = s0.w; bsft[2] = s1.y; bsft[3] = s1.w;
        const float4 u = *reinterpret_cast<const float4*>(sc + (long)n * sbs + (long)cq * 4 * plane + i * 4);
        sv[0] = u.x; sv[1] = u.y; sv[2] = u.z; sv[3] = u.w;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float r = fmaf(v[k], a[k], b[k]);
        if (relu) r = fmaxf(r, 0.f);
        if (sc) r = fmaxf(fmaf(sv[k], as[k], bsft[k]) + r, 0.f);
        v[k] = r;
    }
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}

static int in_stats(const float* x, int64_t bs, int N, int C, int H, int W, float eps, float* stats, double* partial, hipStream_t s,
                    bool c4 = false) {
    // in_partial_kernel reads 16 bytes per thread: a tile-major plane is a multiple of 32 floats, so only the base decides
    NND_REQUIRE(reinterpret_cast<uintptr_t>(x) % 16 == 0 && bs % 4 == 0, "encoder: instance-norm statistics need a 16-byte aligned workspace");
    if (c4) {
        NND_REQUIRE(C % 4 == 0, "encoder: 4-channel-interleaved statistics need C %% 4 == 0");
        hipLaunchKernelGGL(in_partial_c4_kernel, dim3(IN_CHUNKS, C / 4, N), dim3(256), 0, s, x, (long)bs, C, H, W, make_lay(H, W, true), partial);
    } else {
        hipLaunchKernelGGL(in_partial_kernel, dim3(IN_CHUNKS, C, N), dim3(256), 0, s, x, (long)bs, C, H, W, make_lay(H, W, true), partial);
    }
    NND_LAUNCH_CHECK();
    hipLaunchKernelGGL(in_final_kernel, dim3(cdiv(N * C, 256)), dim3(256), 0, s, (const double*)partial, N * C, H * W, eps, stats);
    NND_LAUNCH_CHECK();
    return NND_OK;
}

static int in_apply(float* x, int64_t bs, const float* stats, const float* sc, int64_t sbs, const float* sstats, int N, int C, int H,
                    int W, bool relu, hipStream_t s, bool c4 = false) {
    const long plane = tiled_plane(H, W);
    if (c4) {
        NND_REQUIRE(C % 4 == 0 && bs % 4 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(stats) % 16 == 0 &&
                        (!sc || (sbs % 4 == 0 && reinterpret_cast<uintptr_t>(sc) % 16 == 0 && reinterpret_cast<uintptr_t>(sstats) % 16 == 0)),
                    "encoder: 4-channel-interleaved instance norm needs 16-byte aligned tensors and C %% 4 == 0");
        hipLaunchKernelGGL(in_apply_c4_kernel, dim3((unsigned)cdiv64(plane, 256), C / 4, N), dim3(256), 0, s, x, (long)bs, stats, sc, (long)sbs,
                           sstats, C, plane, relu ? 1 : 0);
        NND_LAUNCH_CHECK();
        return NND_OK;
    }
    const bool v4 = plane % 4 == 0 && bs % 4 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0 &&
                    (!sc || (sbs % 4 == 0 && reinterpret_cast<uintptr_t>(sc) % 16 == 0));
    if (v4)
        hipLaunchKernelGGL(in_apply_kernel<4>, dim3((unsigned)cdiv64(plane, 1024), C, N), dim3(256), 0, s, x, (long)bs, stats, sc, (long)sbs,
                           sstats, C, plane, relu ? 1 : 0);
    else
        hipLaunchKernelGGL(in_apply_kernel<1>, dim3((unsigned)cdiv64(plane, 256), C, N), dim3(256), 0, s, x, (long)bs, stats, sc, (long)sbs,
                           sstats, C, plane, relu ? 1 : 0);
    NND_LAUNCH_CHECK();
    return NND_OK;
}

// ------------------------------------------------------------------------------------------ encoder plan
constexpr int ENC_BLOCKS = 6;
struct EncPlan {
    nnd_encoder_desc d;
    int64_t stem_w, stem_scale, stem_shift;  // (64,3,7,7) weights transposed to [k = c*49 + dy*7 + dx (148 rows, last zero)][co] + folded norm
    ConvLayer c1[ENC_BLOCKS], c2[ENC_BLOCKS], ds[ENC_BLOCKS];
    int64_t base1[ENC_BLOCKS], base2[ENC_BLOCKS], based[ENC_BLOCKS];
    ConvLayer out;  // conv2 1x1
    int64_t base_out;
    ConvLayer cnet;  // cnet_proj 3x3 (optional)
    int64_t base_cnet;
    int64_t total;
    int planes[ENC_BLOCKS], strides[ENC_BLOCKS], inpl[ENC_BLOCKS];
};

static int make_enc_plan(const nnd_encoder_desc* d, EncPlan* p) {
    NND_REQUIRE(d, "encoder: null descriptor");
    NND_REQUIRE(d->struct_size == (int32_t)sizeof(nnd_encoder_desc), "encoder: descriptor of %d bytes, this library expects %d (struct_size)",
                d->struct_size, (int)sizeof(nnd_encoder_desc));
    NND_REQUIRE((d->flags & ~NND_FLAG_CALIBRATE) == 0, "encoder: unknown flags 0x%x", d->flags);
    NND_REQUIRE(d->output_dim > 0 && d->cnet_dim >= 0, "encoder: bad output_dim / cnet_dim");
    NND_REQUIRE(d->norm >= 0 && d->norm <= 2, "encoder: norm must be 0 (none), 1 (batch, eval) or 2 (instance); group norm is not built");
    NND_REQUIRE(d->arithmetic == 0 || d->arithmetic == 3 || d->arithmetic == 2, "encoder: arithmetic must be 0 (fp32 MFMA), 3 (bf16x3) or 2 (fp16x2)");
    p->d = *d;
    int64_t off = 0;
    p->stem_w = off; off += STEM_K * 64;
    p->stem_scale = off; off += 64;
    p->stem_shift = off; off += 64;
    const int dims[3] = {64, 96, 128}, strd[3] = {1, 2, 2};
    int cin = 64;
    for (int i = 0; i < ENC_BLOCKS; ++i) {
        const int dim = dims[i / 2], st = (i % 2 == 0) ? strd[i / 2] : 1;
        p->planes[i] = dim; p->strides[i] = st; p->inpl[i] = cin;
        int64_t t;
        nnd_conv_desc a{dim, cin, 3, 3, st}, b{dim, dim, 3, 3, 1}, c{dim, cin, 1, 1, st};
        int rc;
        if ((rc = conv_layer(&a, &p->c1[i], &t, d->arithmetic)) != NND_OK) return rc;
        p->base1[i] = off; off += t;
        if ((rc = conv_layer(&b, &p->c2[i], &t, d->arithmetic)) != NND_OK) return rc;
        p->base2[i] = off; off += t;
        if ((rc = conv_layer(&c, &p->ds[i], &t, st == 2 ? d->arithmetic : 0)) != NND_OK) return rc;
        p->based[i] = off; off += t;
        cin = dim;
    }
    int64_t t;
    nnd_conv_desc o{d->output_dim, 128, 1, 1, 1};
    int rc = conv_layer(&o, &p->out, &t);
    if (rc != NND_OK) return rc;
    p->base_out = off; off += t;
    p->base_cnet = off;
    if (d->cnet_dim > 0) {
        nnd_conv_desc c{d->cnet_dim, d->output_dim, 3, 3, 1};
        if ((rc = conv_layer(&c, &p->cnet, &t, d->arithmetic)) != NND_OK) return rc;
        off += t;
    }
    p->total = off;
    return NND_OK;
}

static int64_t enc_buf_floats(int N, int H, int W) {  // one activation buffer: 64 channels at 1/2 resolution
    const int h2 = (H + 1) / 2, w2 = (W + 1) / 2;
    return ((int64_t)N * 64 * tiled_plane(h2, w2) + 63) / 64 * 64;
}

}  // namespace nnd

using namespace nnd;

extern "C" {

int64_t nnd_conv_packed_floats(const nnd_conv_desc* desc) {
    ConvLayer L;
    int64_t total;
    if (conv_layer(desc, &L, &total) != NND_OK) return NND_ERR_INVALID;
    return total;
}

int nnd_conv_pack(const nnd_conv_desc* desc, const float* w, const float* bias, const float* bn_gamma, const float* bn_beta,
                  const float* bn_mean, const float* bn_var, float bn_eps, float* packed_host) {
    ConvLayer L;
    int rc = conv_layer(desc, &L, nullptr);
    if (rc != NND_OK) return rc;
    NND_REQUIRE(w && packed_host, "conv_pack: null pointer");
    NND_REQUIRE(!bn_gamma || (bn_beta && bn_mean && bn_var), "conv_pack: incomplete batch-norm parameters");
    pack_conv_norm(L, w, bias, bn_gamma, bn_beta, bn_mean, bn_var, bn_eps, packed_host);
    return NND_OK;
}

int nnd_conv_forward(const nnd_conv_desc* desc, const float* packed_dev, const float* x, const float* residual, float* y,
                     int B, int Hin, int Win, int relu, int relu_after_residual, void* stream) {
    ConvLayer L;
    int rc = conv_layer(desc, &L, nullptr);
    if (rc != NND_OK) return rc;
    NND_REQUIRE(packed_dev && x && y && B > 0 && Hin > 0 && Win > 0, "conv_forward: bad argument");
    const int H = (Hin + L.stride - 1) / L.stride, W = (Win + L.stride - 1) / L.stride;
    return run_conv_norm(L, packed_dev, x, (int64_t)L.Cin * Hin * Win, false, residual, (int64_t)L.Cout * H * W, y,
                         (int64_t)L.Cout * H * W, false, (relu ? 1 : 0) | (relu_after_residual ? 2 : 0), B, Hin, Win,
                         (hipStream_t)stream);
}

int nnd_encoder_num_tensors(const nnd_encoder_desc* desc) {
    EncPlan p;
    if (make_enc_plan(desc, &p) != NND_OK) return NND_ERR_INVALID;
    return 6 * (1 + 3 * ENC_BLOCKS + 1 + (desc->cnet_dim > 0 ? 1 : 0));
}

int64_t nnd_encoder_packed_floats(const nnd_encoder_desc* desc) {
    EncPlan p;
    if (make_enc_plan(desc, &p) != NND_OK) return NND_ERR_INVALID;
    return p.total;
}

int64_t nnd_encoder_workspace_floats(const nnd_encoder_desc* desc, int N, int H, int W) {
    EncPlan p;
    if (make_enc_plan(desc, &p) != NND_OK || N <= 0 || H <= 0 || W <= 0) return NND_ERR_INVALID;
    // + instance-norm statistics (3 slots) and the partial sums of one reduction (doubles)
    return 4 * enc_buf_floats(N, H, W) + (desc->norm == 2 ? (int64_t)3 * N * 256 + (int64_t)N * 128 * IN_CHUNKS * 2 * 2 : 0);
}

// tensors: units of 6 pointers (weight, bias, norm gamma, norm beta, running mean, running var; the last four NULL
// when the conv has no norm) in this order: conv1 | per block (layer1.0 ... layer3.1): conv1, conv2, downsample.0 |
// conv2 | cnet_proj.0 (only if cnet_dim > 0).
int nnd_encoder_pack(const nnd_encoder_desc* desc, const float* const* t, float bn_eps, float* packed_host) {
    EncPlan p;
    int rc = make_enc_plan(desc, &p);
    if (rc != NND_OK) return rc;
    NND_REQUIRE(t && packed_host, "encoder_pack: null pointer");
    const int units = 1 + 3 * ENC_BLOCKS + 1 + (desc->cnet_dim > 0 ? 1 : 0);
    for (int u = 0; u < units; ++u) {
        NND_REQUIRE(t[6 * u] && t[6 * u + 1], "encoder_pack: unit %d: null weight / bias", u);
        const bool has_norm = t[6 * u + 2] != nullptr;
        NND_REQUIRE(!has_norm || (t[6 * u + 3] && t[6 * u + 4] && t[6 * u + 5]), "encoder_pack: unit %d: incomplete norm", u);
        const bool want_norm = desc->norm == 1 && u < units - 1 - (desc->cnet_dim > 0 ? 1 : 0);
        NND_REQUIRE(has_norm == want_norm, "encoder_pack: unit %d: norm parameters %s", u, want_norm ? "missing" : "unexpected");
    }
    memset(packed_host, 0, sizeof(float) * p.total);
    auto unit = [&](int u) { return t + 6 * u; };
    {  // stem: raw weights + folded norm
        const float* const* q = unit(0);
        for (int co = 0; co < 64; ++co)
            for (int k = 0; k < 147; ++k) packed_host[p.stem_w + (int64_t)k * 64 + co] = q[0][co * 147 + k];  // row 147 stays zero
        for (int c = 0; c < 64; ++c) {
            double sc = 1.0, sh = q[1][c];
            if (q[2]) {
                sc = (double)q[2][c] / std::sqrt((double)q[5][c] + (double)bn_eps);
                sh = ((double)q[1][c] - (double)q[4][c]) * sc + (double)q[3][c];
            }
            packed_host[p.stem_scale + c] = (float)sc;
            packed_host[p.stem_shift + c] = (float)sh;
        }
    }
    int u = 1;
    for (int i = 0; i < ENC_BLOCKS; ++i) {
        const float* const* a = unit(u++);
        pack_conv_norm(p.c1[i], a[0], a[1], a[2], a[3], a[4], a[5], bn_eps, packed_host + p.base1[i]);
        const float* const* b = unit(u++);
        pack_conv_norm(p.c2[i], b[0], b[1], b[2], b[3], b[4], b[5], bn_eps, packed_host + p.base2[i]);
        const float* const* c = unit(u++);
        pack_conv_norm(p.ds[i], c[0], c[1], c[2], c[3], c[4], c[5], bn_eps, packed_host + p.based[i]);
    }
    const float* const* o = unit(u++);
    pack_conv_norm(p.out, o[0], o[1], nullptr, nullptr, nullptr, nullptr, bn_eps, packed_host + p.base_out);
    if (desc->cnet_dim > 0) {
        const float* const* c = unit(u++);
        pack_conv_norm(p.cnet, c[0], c[1], nullptr, nullptr, nullptr, nullptr, bn_eps, packed_host + p.base_cnet);
    }
    return NND_OK;
}

// frames (N,3,H,W) NCHW -> fmap (N,output_dim,H8,W8) NCHW, H8 = ceil(ceil(ceil(H/2)/2)/2); cnet_out (optional, needs
// cnet_dim > 0): ReLU(cnet_proj(fmap[:n_cnet])) (n_cnet,cnet_dim,H8,W8).
int nnd_encoder_calibration_finish(const nnd_encoder_desc* desc, float* packed_dev, int32_t* status_dev, void* stream) {
    EncPlan p;
    int rc = make_enc_plan(desc, &p);
    if (rc != NND_OK) return rc;
    NND_REQUIRE(packed_dev, "encoder_calibration_finish: null blob");
    int64_t offs[3 * ENC_BLOCKS + 1];
    int n = 0;
    for (int i = 0; i < ENC_BLOCKS; ++i) {
        if (p.c1[i].arith == 2) offs[n++] = p.base1[i] + p.c1[i].tail_off();
        if (p.c2[i].arith == 2) offs[n++] = p.base2[i] + p.c2[i].tail_off();
        if (p.ds[i].arith == 2 && (p.strides[i] != 1 || p.inpl[i] != p.planes[i])) offs[n++] = p.based[i] + p.ds[i].tail_off();
    }
    if (desc->cnet_dim > 0 && p.cnet.arith == 2) offs[n++] = p.base_cnet + p.cnet.tail_off();
    return calib_finish(packed_dev, offs, n, status_dev, (hipStream_t)stream);
}

int nnd_encoder_forward(const nnd_encoder_desc* desc, const float* packed, const float* frames, float* fmap, float* cnet_out,
                        int n_cnet, float* workspace, int N, int H, int W, void* stream) {
    return nnd_encoder_forward2(desc, packed, frames, nullptr, N, fmap, cnet_out, n_cnet, workspace, N, H, W, stream);
}

int nnd_encoder_forward2(const nnd_encoder_desc* desc, const float* packed, const float* frames, const float* frames_b, int nsplit,
                         float* fmap, float* cnet_out, int n_cnet, float* workspace, int N, int H, int W, void* stream) {
    EncPlan p;
    int rc = make_enc_plan(desc, &p);
    if (rc != NND_OK) return rc;
    NND_REQUIRE(packed && frames && fmap && workspace && N > 0 && H > 0 && W > 0, "encoder_forward: bad argument");
    NND_REQUIRE(!frames_b || (nsplit > 0 && nsplit < N), "encoder_forward2: a second frame tensor needs 0 < nsplit < N");
    NND_REQUIRE(!cnet_out || (desc->cnet_dim > 0 && n_cnet > 0 && n_cnet <= N), "encoder_forward: cnet_out needs cnet_dim > 0 and 0 < n_cnet <= N");
    CalibScope calib((desc->flags & NND_FLAG_CALIBRATE) && desc->arithmetic == 2);
    hipStream_t s = (hipStream_t)stream;
    const int64_t bufsz = enc_buf_floats(N, H, W);
    float* buf[4] = {workspace, workspace + bufsz, workspace + 2 * bufsz, workspace + 3 * bufsz};
    int h = (H + 1) / 2, w = (W + 1) / 2;
    const bool inorm = desc->norm == 2;
    const float eps = 1e-5f;  // nn.InstanceNorm2d default, as constructed by the reference (basic_encoder.py:33-34)
    float* st_a = workspace + 4 * bufsz;  // instance-norm statistics: 3 slots of N * 128 * 2 floats
    float* st_b = st_a + (int64_t)N * 256;
    float* st_c = st_b + (int64_t)N * 256;
    double* part = reinterpret_cast<double*>(st_c + (int64_t)N * 256);  // 8-byte aligned: every offset above is a multiple of 64 floats
    // Round 4: with a split arithmetic (every 3x3 and stride-2 layer on conv_split, the stride-1 shortcuts and the output conv on
    // the streaming 1x1 kernel) the activations between the layers keep 4 channels interleaved (layout.h "c4"): a staging unit is two
    // 16-byte loads instead of eight 4-byte ones, an epilogue lane stores / reads its residual as 16 bytes — the layer1 convs were
    // bound by the CU's vector-memory instruction rate (about 400 wave-instructions per 64-pixel workgroup, epilogue 17 of 37 us
    // per workgroup: scripts/stamps_encoder.py).  The exact arithmetic (stride-2 layers on conv_mfma: planar sources only) and the
    // same tensors under NND_ENC_NO_C4 (diagnostic) stay planar; the instance-norm statistics / apply kernels have c4 variants that visit the
    // pixels in the planar kernels' order (bit-identical statistics).
    const bool c4 = desc->arithmetic != 0 && !switches().enc_no_c4;
    {  // stem -> buf[0]
        const Lay lay = make_lay(h, w, true, c4);
        const int tiles_x = cdiv(w, 32), tiles_y = cdiv(h, 8);
        hipLaunchKernelGGL(stem_kernel, dim3(tiles_x * tiles_y, 1, N), dim3(256), 0, s, frames, frames_b, frames_b ? nsplit : N, packed + p.stem_w,
                           packed + p.stem_scale, packed + p.stem_shift, buf[0], (long)(64 * lay.plane), H, W, h, w, tiles_x, lay,
                           inorm ? 0 : 1);
        NND_LAUNCH_CHECK();
    }
    int cur = 0;  // buffer holding the block input
#define NND_TRY(x)                    \
    do {                              \
        if ((rc = (x)) != NND_OK) return rc; \
    } while (0)
    if (inorm) {  // norm1 + ReLU of the stem
        NND_TRY(in_stats(buf[0], 64 * tiled_plane(h, w), N, 64, h, w, eps, st_a, part, s, c4));
        NND_TRY(in_apply(buf[0], 64 * tiled_plane(h, w), st_a, nullptr, 0, nullptr, N, 64, h, w, true, s, c4));
    }
    for (int i = 0; i < ENC_BLOCKS; ++i) {
        const int st = p.strides[i], cin = p.inpl[i], dim = p.planes[i];
        const int ho = (h + st - 1) / st, wo = (w + st - 1) / st;
        const int64_t pin = tiled_plane(h, w), pout = tiled_plane(ho, wo);
        float* x = buf[cur];
        float* y = buf[(cur + 1) & 3];
        float* sc = buf[(cur + 2) & 3];
        float* o = buf[(cur + 3) & 3];
        if (!inorm) {
            NND_TRY(run_conv_norm(p.c1[i], packed + p.base1[i], x, cin * pin, true, nullptr, 0, y, dim * pout, true, 1, N, h, w, s, c4, c4));
            NND_TRY(run_conv_norm(p.ds[i], packed + p.based[i], x, cin * pin, true, nullptr, 0, sc, dim * pout, true, 0, N, h, w, s, c4, c4));
            NND_TRY(run_conv_norm(p.c2[i], packed + p.base2[i], y, dim * pout, true, sc, dim * pout, o, dim * pout, true, 3, N, ho, wo, s, c4, c4));
        } else {  // raw convs + per-sample statistics; the shortcut is normalised inside the block's final apply pass
            NND_TRY(run_conv_norm(p.c1[i], packed + p.base1[i], x, cin * pin, true, nullptr, 0, y, dim * pout, true, 0, N, h, w, s, c4, c4));
            NND_TRY(in_stats(y, dim * pout, N, dim, ho, wo, eps, st_a, part, s, c4));
            NND_TRY(in_apply(y, dim * pout, st_a, nullptr, 0, nullptr, N, dim, ho, wo, true, s, c4));
            NND_TRY(run_conv_norm(p.ds[i], packed + p.based[i], x, cin * pin, true, nullptr, 0, sc, dim * pout, true, 0, N, h, w, s, c4, c4));
            NND_TRY(in_stats(sc, dim * pout, N, dim, ho, wo, eps, st_b, part, s, c4));
            NND_TRY(run_conv_norm(p.c2[i], packed + p.base2[i], y, dim * pout, true, nullptr, 0, o, dim * pout, true, 0, N, ho, wo, s, c4, c4));
            NND_TRY(in_stats(o, dim * pout, N, dim, ho, wo, eps, st_c, part, s, c4));
            NND_TRY(in_apply(o, dim * pout, st_c, sc, dim * pout, st_b, N, dim, ho, wo, true, s, c4));
        }
        cur = (cur + 3) & 3;
        h = ho; w = wo;
    }
    const int64_t pl = tiled_plane(h, w);
    NND_TRY(run_conv_norm(p.out, packed + p.base_out, buf[cur], 128 * pl, true, nullptr, 0, fmap, (int64_t)desc->output_dim * h * w,
                          false, 0, N, h, w, s, c4, false));
    if (cnet_out)
        NND_TRY(run_conv_norm(p.cnet, packed + p.base_cnet, fmap, (int64_t)desc->output_dim * h * w, false, nullptr, 0, cnet_out,
                              (int64_t)desc->cnet_dim * h * w, false, 1, n_cnet, h, w, s));
#undef NND_TRY
    return NND_OK;
}

}  // extern "C"
