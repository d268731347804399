// 3x3x3 Conv3d (+ folded eval BatchNorm3d + LeakyReLU) on the 2-D MFMA convolution, for IGEV's cost-volume regulariser
// (SURVEY §8 row a15 / §8f-2): ConvBn3D and the conv of Upsampler3D, nndepth/models/igev_stereo/cost_volume.py:101-130.
//
// Layout trick: volumes are kept DEPTH-MAJOR, (N, D+2, C, H, W) with a zero slice before and after the D real ones.  For an
// output slice d the 3 depth taps x Cin channels are then 3*Cin CONSECUTIVE (H,W) planes starting at padded slice d, so
//     y[n, d] = conv2d_3x3( x[n, d : d+3] viewed as 3*Cin channels,  W'[co][kd*Cin + ci][kh][kw] )
// is exactly one launch of conv_mfma with "batch" = the D output slices and batch stride = Cin*H*W (2*Cin*H*W for stride 2,
// where the 2-D kernel also strides by 2).  A channel concat of two volumes (proj_2 / proj_3) is the kernel's virtual
// concat of two such windows.  No im2col, no extra copy; the zero slices provide the depth padding.
#include "common.h"

#include <cmath>
#include <cstring>
#include <vector>

namespace nnd {

static int conv3d_layer(const nnd_conv3d_desc* d, ConvLayer* L, int64_t* total) {
    NND_REQUIRE(d, "conv3d: null descriptor");
    NND_REQUIRE(d->Cout > 0 && d->Cin0 > 0 && d->Cin1 >= 0, "conv3d: bad channel counts");
    NND_REQUIRE(d->stride == 1 || d->stride == 2, "conv3d: stride %d not supported (1, 2)", d->stride);
    ConvLayer l;
    l.KH = l.KW = 3;
    l.Cin = 3 * (d->Cin0 + d->Cin1);
    l.Cout = d->Cout;
    l.stride = d->stride;
    l.CI_T = 16;  // windows of 3*Cin planes: 24, 48, 96, 192 -> 16-channel chunks keep the two sources chunk-aligned
    NND_REQUIRE(d->Cin1 == 0 || (3 * d->Cin0) % l.CI_T == 0, "conv3d: first input of a concat needs 3*Cin0 %% 16 == 0 (Cin0 = %d)", d->Cin0);
    l.nchunks = cdiv(l.Cin, l.CI_T);
    l.ncb = cdiv(l.Cout, 32);
    int64_t off = 0;
    l.w_off = off; off += l.w_floats();
    l.b_off = off; off += l.b_floats();
    l.s_off = off; off += l.b_floats();
    *L = l;
    if (total) *total = off;
    return NND_OK;
}

// (N, C, D, H, W) -> depth-major (N, D+2, C, H, W), zero slices at both ends.  One thread per element of the output.
__global__ void __launch_bounds__(256) to_depth_major_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int D,
                                                             long HW) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= HW) return;
    const int c = blockIdx.y % C, dp = blockIdx.y / C, n = blockIdx.z;  // dp: padded slice index 0..D+1
    const bool real = dp >= 1 && dp <= D;
    const float v = real ? src[(((long)n * C + c) * D + (dp - 1)) * HW + i] : 0.f;
    dst[(((long)n * (D + 2) + dp) * C + c) * HW + i] = v;
}

__global__ void __launch_bounds__(256) from_depth_major_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int D,
                                                               long HW) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= HW) return;
    const int c = blockIdx.y % C, d = blockIdx.y / C, n = blockIdx.z;
    dst[(((long)n * C + c) * D + d) * HW + i] = src[(((long)n * (D + 2) + d + 1) * C + c) * HW + i];
}

}  // namespace nnd

using namespace nnd;

extern "C" {

int64_t nnd_conv3d_packed_floats(const nnd_conv3d_desc* desc) {
    ConvLayer L;
    int64_t total;
    if (conv3d_layer(desc, &L, &total) != NND_OK) return NND_ERR_INVALID;
    return total;
}

// w (Cout, Cin0+Cin1, 3, 3, 3) [kd, kh, kw]; bias may be NULL (the reference's ConvBn3D has bias=False); bn_* may be NULL
int nnd_conv3d_pack(const nnd_conv3d_desc* desc, const float* w, const float* bias, const float* bn_gamma, const float* bn_beta,
                    const float* bn_mean, const float* bn_var, float bn_eps, float* packed_host) {
    ConvLayer L;
    int rc = conv3d_layer(desc, &L, nullptr);
    if (rc != NND_OK) return rc;
    NND_REQUIRE(w && packed_host, "conv3d_pack: null pointer");
    NND_REQUIRE(!bn_gamma || (bn_beta && bn_mean && bn_var), "conv3d_pack: incomplete batch-norm parameters");
    const int C0 = desc->Cin0, C1 = desc->Cin1, Ct = C0 + C1, Co = desc->Cout;
    // 2-D weight (Cout, 3*Ct, 3, 3): channel order = [window of input 0: kd-major, ci] [window of input 1: kd-major, ci]
    std::vector<float> w2((size_t)Co * 3 * Ct * 9);
    for (int co = 0; co < Co; ++co)
        for (int ci = 0; ci < Ct; ++ci)
            for (int kd = 0; kd < 3; ++kd) {
                const int c2 = ci < C0 ? kd * C0 + ci : 3 * C0 + kd * C1 + (ci - C0);
                for (int t = 0; t < 9; ++t)
                    w2[((size_t)co * 3 * Ct + c2) * 9 + t] = w[(((size_t)co * Ct + ci) * 3 + kd) * 9 + t];
            }
    const float* ws[1] = {w2.data()};
    const float* bs[1] = {nullptr};
    int cc[1] = {Co};
    pack_conv(L, 1, ws, bs, cc, packed_host);
    float* shift = packed_host + L.b_off;
    float* scale = packed_host + L.s_off;
    for (int c = 0; c < L.ncb * 32; ++c) {
        double sc = 1.0, sh = 0.0;
        if (c < Co) {
            const double b = bias ? (double)bias[c] : 0.0;
            if (bn_gamma) {
                sc = (double)bn_gamma[c] / std::sqrt((double)bn_var[c] + (double)bn_eps);
                sh = (b - (double)bn_mean[c]) * sc + (double)bn_beta[c];
            } else {
                sh = b;
            }
        }
        scale[c] = (float)sc;
        shift[c] = (float)sh;
    }
    return NND_OK;
}

// x0 (N, D+2, Cin0, H, W), x1 (N, D+2, Cin1, H, W) or NULL, y (N, Do+2, Cout, Ho, Wo) — all depth-major with zero end slices
// (y's end slices are written by this call); Do = ceil(D/stride) etc.  leaky_slope: LeakyReLU negative slope (1 = none).
int nnd_conv3d_forward(const nnd_conv3d_desc* desc, const float* packed, const float* x0, const float* x1, float* y, int N, int D,
                       int H, int W, float leaky_slope, void* stream) {
    ConvLayer L;
    int rc = conv3d_layer(desc, &L, nullptr);
    if (rc != NND_OK) return rc;
    NND_REQUIRE(packed && x0 && y && (desc->Cin1 == 0 || x1), "conv3d_forward: null pointer");
    NND_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0, "conv3d_forward: bad shape");
    const int st = desc->stride;
    const int Do = (D + st - 1) / st, Ho = (H + st - 1) / st, Wo = (W + st - 1) / st;
    NND_REQUIRE(Do <= 65535, "conv3d_forward: depth %d exceeds the grid limit", Do);
    hipStream_t s = (hipStream_t)stream;
    const int64_t hw = (int64_t)H * W, hwo = (int64_t)Ho * Wo;
    for (int n = 0; n < N; ++n) {
        float* yn = y + (int64_t)n * (Do + 2) * desc->Cout * hwo;
        NND_HIP_CHECK(hipMemsetAsync(yn, 0, sizeof(float) * desc->Cout * hwo, s));
        NND_HIP_CHECK(hipMemsetAsync(yn + (int64_t)(Do + 1) * desc->Cout * hwo, 0, sizeof(float) * desc->Cout * hwo, s));
        ConvIO io{};
        // output slice d reads padded input slices st*d .. st*d+2 (= real slices st*d-1 .. st*d+1)
        io.src0 = Act{const_cast<float*>(x0) + (int64_t)n * (D + 2) * desc->Cin0 * hw, (int64_t)st * desc->Cin0 * hw, 3 * desc->Cin0};
        if (desc->Cin1 > 0)
            io.src1 = Act{const_cast<float*>(x1) + (int64_t)n * (D + 2) * desc->Cin1 * hw, (int64_t)st * desc->Cin1 * hw, 3 * desc->Cin1};
        io.out0 = Act{yn + desc->Cout * hwo, (int64_t)desc->Cout * hwo, desc->Cout};
        io.Hin = H; io.Win = W;
        io.flags = leaky_slope != 1.0f ? 4 : 0;
        io.scale = leaky_slope;
        rc = launch_conv(L, packed, io, EPI_AFFINE, Do, Ho, Wo, s);
        if (rc != NND_OK) return rc;
    }
    return NND_OK;
}

int nnd_volume_to_depth_major(const float* x, float* y, int N, int C, int D, int H, int W, void* stream) {
    NND_REQUIRE(x && y && N > 0 && C > 0 && D > 0 && H > 0 && W > 0 && (long)C * (D + 2) <= 65535, "volume_to_depth_major: bad argument");
    const long HW = (long)H * W;
    hipLaunchKernelGGL(to_depth_major_kernel, dim3((unsigned)cdiv64(HW, 256), C * (D + 2), N), dim3(256), 0, (hipStream_t)stream, x, y, C,
                       D, HW);
    NND_LAUNCH_CHECK();
    return NND_OK;
}

int nnd_depth_major_to_volume(const float* x, float* y, int N, int C, int D, int H, int W, void* stream) {
    NND_REQUIRE(x && y && N > 0 && C > 0 && D > 0 && H > 0 && W > 0 && (long)C * D <= 65535, "depth_major_to_volume: bad argument");
    const long HW = (long)H * W;
    hipLaunchKernelGGL(from_depth_major_kernel, dim3((unsigned)cdiv64(HW, 256), C * D, N), dim3(256), 0, (hipStream_t)stream, x, y, C, D,
                       HW);
    NND_LAUNCH_CHECK();
    return NND_OK;
}

}  // extern "C"
