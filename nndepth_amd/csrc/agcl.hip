// CREStereo adaptive group correlation layer (AGCL) and its zero-padded bilinear sampler.
//
// Replaces nndepth/models/cre_stereo/cost_volume.py:28-154 and nndepth/models/cre_stereo/utils.py:5-20,34-107
// (reference; semantics restated in oracle/cre_ref.py: bilinear_sampler / agcl_corr_iter / agcl_corr_att_offset).
//
// The kernels are gather + short dot products: HBM/L2-bound, no MFMA.  Consecutive lanes are consecutive
// pixels, so the left-feature loads and the output stores are coalesced and the four gathers of a tap land in the
// one or two cache lines the neighbouring lanes touch as well (the flow field is smooth).
//   sample      : out[n,c,p] = bilinear(img[n,c], coords[n,p])                 thread = point, loops channels
//   window_corr : iter mode, second pass: the right features are warped ONCE by (grid + flow) with `sample` (4 gathers
//                 per value instead of 36), then out[n,g*9+k,y,x] = mean_c f1_g[c,y,x] * warped_g[c, clamp(y+dy_k),
//                 clamp(x+dx_k)] with coalesced loads only            thread = (pixel, group, channel slice)
//   offset_corr : out[n,g*9+k,y,x] = mean_c f1_g[c,y,x] * bilinear(f2_g[c], (x,y) + flow + window_k + extra_k)
//                 9 tap sets per pixel prepared once                   thread = (pixel, group, channel slice)
//
// Compiled with -ffp-contract=off: the coordinate round trip pixel -> [-1,1] -> pixel and the
// Ia*wa + Ib*wb + Ic*wc + Id*wd sum keep the reference's rounding sequence.
#include "common.h"
#include "layout.h"

namespace nnd {

struct Taps {
    int o00, o01, o10, o11;    // element offsets inside one channel plane (clamped into the image)
    float w00, w01, w10, w11;  // a tap outside the image (the zero border of the reference) gets weight 0 instead of
                               // value 0: the product is the same 0 for finite features, and it costs no flag registers
};

// x, y: pixel coordinates.  utils.py:9-10 maps them to [-1,1], utils.py:55-56 (align_corners=True) maps back.
__device__ __forceinline__ Taps make_taps(float x, float y, int H, int W) {
    const float wm = (float)(W - 1), hm = (float)(H - 1);
    x = ((2.0f * x / wm - 1.0f) + 1.0f) / 2.0f * wm;
    y = ((2.0f * y / hm - 1.0f) + 1.0f) / 2.0f * hm;
    const float fx0 = floorf(x), fy0 = floorf(y);
    const float fx1 = fx0 + 1.0f, fy1 = fy0 + 1.0f;
    Taps t;
    t.w00 = (fx1 - x) * (fy1 - y);
    t.w01 = (fx1 - x) * (y - fy0);
    t.w10 = (x - fx0) * (fy1 - y);
    t.w11 = (x - fx0) * (y - fy0);
    // integer taps; far-away / non-finite coordinates collapse onto "outside"
    const int x0 = (int)fminf(fmaxf(fx0, -2.0f), (float)W), y0 = (int)fminf(fmaxf(fy0, -2.0f), (float)H);
    const int x1 = x0 + 1, y1 = y0 + 1;
    const bool xin0 = x0 >= 0 && x0 < W, xin1 = x1 >= 0 && x1 < W;
    const bool yin0 = y0 >= 0 && y0 < H, yin1 = y1 >= 0 && y1 < H;
    const int cx0 = min(max(x0, 0), W - 1), cx1 = min(max(x1, 0), W - 1);
    const int cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y1, 0), H - 1);
    t.o00 = cy0 * W + cx0;
    t.o01 = cy1 * W + cx0;
    t.o10 = cy0 * W + cx1;
    t.o11 = cy1 * W + cx1;
    if (!(xin0 && yin0)) t.w00 = 0.f;
    if (!(xin0 && yin1)) t.w01 = 0.f;
    if (!(xin1 && yin0)) t.w10 = 0.f;
    if (!(xin1 && yin1)) t.w11 = 0.f;
    return t;
}

__device__ __forceinline__ float tap_sum(const float* __restrict__ plane, const Taps& t) {
    return plane[t.o00] * t.w00 + plane[t.o01] * t.w01 + plane[t.o10] * t.w10 + plane[t.o11] * t.w11;
}

// coords != nullptr: explicit sample points (N, Hg*Wg, 2).  coords == nullptr: warp by flow (N,2,H,W) on the
// image's own grid (Hg*Wg == H*W): point (x, y) samples (x + flow_x, y + flow_y).
// grid: (ceil(P/256), ceil(C/CPB), N)
constexpr int SAMPLE_CPB = 8;  // channels per block: 32 gathers in flight per thread
__global__ void __launch_bounds__(256) sample_kernel(const float* __restrict__ img, const float* __restrict__ coords,
                                                     const float* __restrict__ flow, float* __restrict__ out, int C,
                                                     int H, int W, int P, Lay fl_lay) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const int n = blockIdx.z;
    float x, y;
    if (coords) {
        const float2 xy = reinterpret_cast<const float2*>(coords)[(long)n * P + p];
        x = xy.x;
        y = xy.y;
    } else {
        const int py = p / W, px = p - py * W;
        const long fo = pix_off(fl_lay, py, px);
        x = (float)px + flow[((long)n * 2 + 0) * fl_lay.plane + fo];
        y = (float)py + flow[((long)n * 2 + 1) * fl_lay.plane + fo];
    }
    const Taps t = make_taps(x, y, H, W);
    const int c0 = blockIdx.y * SAMPLE_CPB, c1 = min(c0 + SAMPLE_CPB, C);
    const float* plane = img + ((long)n * C + c0) * H * W;
    float* o = out + ((long)n * C + c0) * P + p;
#pragma unroll 8
    for (int c = c0; c < c1; ++c) {
        *o = tap_sum(plane, t);
        plane += (long)H * W;
        o += P;
    }
}

__device__ __forceinline__ void window_offset(int k, bool small_patch, int& dy, int& dx) {
    if (small_patch) {  // 3x3, dy outer / dx inner (cost_volume.py:41-46)
        dy = k / 3 - 1;
        dx = k % 3 - 1;
    } else {  // 1x9
        dy = 0;
        dx = k - 4;
    }
}

// ITER mode, second pass (the right features were warped once by `sample_kernel` into a scratch map):
//   out[n,g*9+k,y,x] = mean_c f1_g[c,y,x] * warped_g[c, clamp(y+dy_k), clamp(x+dx_k)]     (replicate padding = index clamp)
// Block = 64 consecutive pixels x 4 waves; blockIdx.y = channel group; the 4 waves split the group's channels into 4
// slices and their 9 partial sums meet in LDS.  All loads are coalesced (a wave reads 64 consecutive pixels, shifted).
// grid: (ceil(HW/64), 4 groups, N)
__global__ void __launch_bounds__(256) window_corr_kernel(const float* __restrict__ f1, const float* __restrict__ warped,
                                                          float* __restrict__ out, int C, int H, int W, int small_patch,
                                                          Lay out_lay) {
    __shared__ float red[3][9][64];
    const int HW = H * W;
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const bool p_ok = (int)blockIdx.x * 64 + lane < HW;
    const int p = min((int)blockIdx.x * 64 + lane, HW - 1);  // tail lanes redo the last pixel: no exit before the barrier
    const int g = blockIdx.y, n = blockIdx.z, G = C / 4;
    const int y = p / W, x = p - y * W;
    int off[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        int dy, dx;
        window_offset(k, small_patch != 0, dy, dx);
        off[k] = min(max(y + dy, 0), H - 1) * W + min(max(x + dx, 0), W - 1);
    }
    const int per = (G + 3) / 4, c0 = slice * per, c1 = min(c0 + per, G);
    const float* l = f1 + ((long)n * C + (long)g * G + c0) * HW + p;
    const float* r = warped + ((long)n * C + (long)g * G + c0) * HW;
    float acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0.f;
#pragma unroll 4
    for (int c = c0; c < c1; ++c) {
        const float lv = *l;
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[k] += lv * r[off[k]];
        l += HW;
        r += HW;
    }
    if (slice > 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) red[slice - 1][k][lane] = acc[k];
    }
    __syncthreads();
    if (slice == 0 && p_ok) {
        const float inv = (float)G;
        float* o = out + ((long)n * 36 + g * 9) * out_lay.plane + pix_off(out_lay, y, x);
#pragma unroll
        for (int k = 0; k < 9; ++k) o[(long)k * out_lay.plane] = (((acc[k] + red[0][k][lane]) + red[1][k][lane]) + red[2][k][lane]) / inv;
    }
}

// OFFSET mode: position k samples (x, y) + flow(x, y) + window_k + extra_k(x, y): every (pixel, k) has its own
// sub-pixel position, so nothing can be shared — 9*4 gathers per pixel and channel, bound by the texture-address
// rate, not by HBM.  Same block shape as above; a thread prepares the 9 tap sets of its pixel once.
// grid: (ceil(HW/64), 4 groups, N)
__global__ void __launch_bounds__(256, 3) offset_corr_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                             const float* __restrict__ flow, const float* __restrict__ extra,
                                                             float* __restrict__ out, int C, int H, int W, int small_patch,
                                                             Lay fl_lay, Lay out_lay) {
    __shared__ float red[3][9][64];
    const int HW = H * W;
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const bool p_ok = (int)blockIdx.x * 64 + lane < HW;
    const int p = min((int)blockIdx.x * 64 + lane, HW - 1);
    const int g = blockIdx.y, n = blockIdx.z, G = C / 4;
    const int y = p / W, x = p - y * W;
    const float* fl = flow + (long)n * 2 * fl_lay.plane + pix_off(fl_lay, y, x);
    const float flx = fl[0], fly = fl[fl_lay.plane];
    Taps t[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        int dy, dx;
        window_offset(k, small_patch != 0, dy, dx);
        // reference order (cost_volume.py:136-141): offsets = window + extra; coords = (grid + flow) + offsets
        const float ex = extra[((long)n * 18 + 2 * k) * HW + p], ey = extra[((long)n * 18 + 2 * k + 1) * HW + p];
        const float sx = ((float)x + flx) + ((float)dx + ex);
        const float sy = ((float)y + fly) + ((float)dy + ey);
        t[k] = make_taps(sx, sy, H, W);
    }
    const int per = (G + 3) / 4, c0 = slice * per, c1 = min(c0 + per, G);
    const float* l = f1 + ((long)n * C + (long)g * G + c0) * HW + p;
    const float* r = f2 + ((long)n * C + (long)g * G + c0) * HW;
    float acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0.f;
#pragma unroll 1
    for (int c = c0; c < c1; ++c) {
        const float lv = *l;
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[k] += lv * tap_sum(r, t[k]);
        l += HW;
        r += HW;
    }
    if (slice > 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) red[slice - 1][k][lane] = acc[k];
    }
    __syncthreads();
    if (slice == 0 && p_ok) {
        const float inv = (float)G;
        float* o = out + ((long)n * 36 + g * 9) * out_lay.plane + pix_off(out_lay, y, x);
#pragma unroll
        for (int k = 0; k < 9; ++k) o[(long)k * out_lay.plane] = (((acc[k] + red[0][k][lane]) + red[1][k][lane]) + red[2][k][lane]) / inv;
    }
}

static int check_agcl(const char* what, int N, int C, int H, int W) {
    NND_REQUIRE(N > 0 && C > 0 && H > 1 && W > 1, "%s: bad shape (N=%d C=%d H=%d W=%d; H, W must be > 1)", what, N, C, H, W);
    NND_REQUIRE(C % 4 == 0, "%s: channels %d not divisible by the 4 correlation groups", what, C);
    NND_REQUIRE((long)C * H * W < (1L << 31), "%s: plane offsets exceed 32 bits", what);
    return NND_OK;
}


int agcl_iter_launch(const float* f1, const float* f2, const float* flow, float* warped, float* out, int N, int C, int H,
                     int W, int small_patch, hipStream_t s, bool tiled) {
    const int HW = H * W;
    const Lay lay = make_lay(H, W, tiled);
    hipLaunchKernelGGL(sample_kernel, dim3(cdiv(HW, 256), cdiv(C, SAMPLE_CPB), N), dim3(256), 0, s, f2, (const float*)nullptr,
                       flow, warped, C, H, W, HW, lay);
    NND_LAUNCH_CHECK();
    hipLaunchKernelGGL(window_corr_kernel, dim3(cdiv(HW, 64), 4, N), dim3(256), 0, s, f1, (const float*)warped, out, C, H, W,
                       small_patch, lay);
    NND_LAUNCH_CHECK();
    return NND_OK;
}

int agcl_offset_launch(const float* f1, const float* f2, const float* flow, const float* extra, float* out, int N, int C, int H,
                       int W, int small_patch, hipStream_t s, bool tiled) {
    const Lay lay = make_lay(H, W, tiled);
    hipLaunchKernelGGL(offset_corr_kernel, dim3(cdiv(H * W, 64), 4, N), dim3(256), 0, s, f1, f2, flow, extra, out, C, H, W,
                       small_patch, lay, lay);
    NND_LAUNCH_CHECK();
    return NND_OK;
}

int agcl_check(const char* what, int N, int C, int H, int W) { return check_agcl(what, N, C, H, W); }

}  // namespace nnd

using namespace nnd;

extern "C" {

int nnd_bilinear_sample(const float* img, const float* coords, float* out, int N, int C, int H, int W, int Hg, int Wg,
                        void* stream) {
    NND_REQUIRE(img && coords && out, "bilinear_sample: null pointer");
    NND_REQUIRE(N > 0 && C > 0 && H > 1 && W > 1 && Hg > 0 && Wg > 0, "bilinear_sample: bad shape");
    const int P = Hg * Wg;
    dim3 grid(cdiv(P, 256), cdiv(C, SAMPLE_CPB), N);
    hipLaunchKernelGGL(sample_kernel, grid, dim3(256), 0, (hipStream_t)stream, img, coords, (const float*)nullptr, out, C, H,
                       W, P, make_lay(H, W, false));
    NND_LAUNCH_CHECK();
    return NND_OK;
}

int nnd_agcl_corr_iter(const float* fmap1, const float* fmap2, const float* flow, float* warped, float* out, int N, int C,
                       int H, int W, int small_patch, void* stream) {
    NND_REQUIRE(fmap1 && fmap2 && flow && warped && out, "agcl_corr_iter: null pointer");
    int rc = check_agcl("agcl_corr_iter", N, C, H, W);
    if (rc != NND_OK) return rc;
    return agcl_iter_launch(fmap1, fmap2, flow, warped, out, N, C, H, W, small_patch, (hipStream_t)stream, false);
}

int nnd_agcl_corr_offset(const float* fmap1, const float* fmap2, const float* flow, const float* extra_offset, float* out,
                         int N, int C, int H, int W, int small_patch, void* stream) {
    NND_REQUIRE(fmap1 && fmap2 && flow && extra_offset && out, "agcl_corr_offset: null pointer");
    int rc = check_agcl("agcl_corr_offset", N, C, H, W);
    if (rc != NND_OK) return rc;
    return agcl_offset_launch(fmap1, fmap2, flow, extra_offset, out, N, C, H, W, small_patch, (hipStream_t)stream, false);
}

}  // extern "C"
