// CREStereo adaptive group correlation layer (AGCL) and its zero-padded bilinear sampler.
//
// Replaces nndepth/models/cre_stereo/cost_volume.py:28-154 and nndepth/models/cre_stereo/utils.py:5-20,34-107
// (reference; semantics restated in oracle/cre_ref.py: bilinear_sampler / agcl_corr_iter / agcl_corr_att_offset).
//
// The kernels are gather + short dot products: HBM/L2-bound, no MFMA.  Consecutive lanes are consecutive
// pixels, so the left-feature loads and the output stores are coalesced and the four gathers of a tap land in the
// one or two cache lines the neighbouring lanes touch as well (the flow field is smooth).
//   sample      : out[n,c,p] = bilinear(img[n,c], coords[n,p])                 thread = point, loops channels; the two
//                 x-adjacent taps of a row are ONE 8-byte load
//   window_corr : iter mode, second pass: the right features are warped ONCE by (grid + flow) with `sample` (4 gathers
//                 per value instead of 36), then out[n,g*9+k,y,x] = mean_c f1_g[c,y,x] * warped_g[c, clamp(y+dy_k),
//                 clamp(x+dx_k)] with coalesced loads only            thread = (pixel, group, channel slice);
//                 window_corr_x4 (W % 4 == 0): thread = 4 consecutive pixels, 16-byte loads — the vector-memory pipe
//                 spends 16 cycles per wave-instruction whatever its payload, so 4 loads per channel and 4 pixels
//                 instead of 40 is what takes the kernel from the address rate to the HBM rate
//   offset_corr : out[n,g*9+k,y,x] = mean_c f1_g[c,y,x] * bilinear(f2_g[c], (x,y) + flow + window_k + extra_k)
//                 9 tap sets per pixel prepared once                   thread = (pixel, group, channel slice)
//   offset_corr_cl : the same on channels-last copies of the two maps (made once per stage: the maps are constant over
//                 the iterations): one wave = one pixel, lane = 4 channels, a tap is one 1-KB line read by the whole wave
//
// Compiled with -ffp-contract=off: the coordinate round trip pixel -> [-1,1] -> pixel and the
// Ia*wa + Ib*wb + Ic*wc + Id*wd sum keep the reference's rounding sequence.
#include "common.h"
#include "layout.h"
#include <cstdlib>
#include <type_traits>

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

// The same four taps with the two x-neighbours of a row fetched by one 8-byte load at column clamp(x0, 0, W-2): hi0 / hi1
// say which half of the pair tap (x0, .) / (x1, .) is (x0 = -1 and x0 = W-1 have one tap outside: weight 0, any finite half).
struct PairTaps {
    int r0, r1;  // element offsets of the pairs in rows y0 and y1 (clamped)
    bool hi0, hi1;
    float w00, w01, w10, w11;
};
struct __attribute__((packed, aligned(4))) Pair {
    float lo, hi;
};

__device__ __forceinline__ PairTaps make_pair_taps(float x, float y, int H, int W) {
    const Taps t = make_taps(x, y, H, W);
    const int cy0 = t.o00 / W, cx0 = t.o00 - cy0 * W, cx1 = t.o10 - cy0 * W, cy1 = t.o01 / W;
    const int bx = min(cx0, W - 2);
    PairTaps q;
    q.r0 = cy0 * W + bx;
    q.r1 = cy1 * W + bx;
    q.hi0 = cx0 != bx;
    q.hi1 = cx1 != bx;
    q.w00 = t.w00, q.w01 = t.w01, q.w10 = t.w10, q.w11 = t.w11;
    return q;
}

__device__ __forceinline__ float tap_sum(const float* __restrict__ plane, const PairTaps& t) {
    const Pair a = *reinterpret_cast<const Pair*>(plane + t.r0), b = *reinterpret_cast<const Pair*>(plane + t.r1);
    const float v00 = t.hi0 ? a.hi : a.lo, v10 = t.hi1 ? a.hi : a.lo;
    const float v01 = t.hi0 ? b.hi : b.lo, v11 = t.hi1 ? b.hi : b.lo;
    return v00 * t.w00 + v01 * t.w01 + v10 * t.w10 + v11 * t.w11;
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
    const PairTaps t = make_pair_taps(x, y, H, W);
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

// ITER mode, second pass, 4 consecutive pixels per lane (W % 4 == 0; warped rows are then 16-byte aligned).
//   1x9 window: the 12 values x0-4 .. x0+7 of a channel row are three aligned 16-byte loads (the clamped neighbours of an
//               edge run are the run's own first / last value: replicate padding), and pixel j, position k reads value j+k;
//   3x3 window: per clamped row one 16-byte load + the two clamped side values.
// Same expression order as window_corr_kernel (products added channel by channel, slices summed 0+1+2+3): bit-identical.
// grid: (ceil(H*W/4/64), 4 groups, N)
template <bool SMALL>
__global__ void __launch_bounds__(256) window_corr_x4_kernel(const float* __restrict__ f1, const float* __restrict__ warped,
                                                             float* __restrict__ out, int C, int H, int W, Lay out_lay) {
    __shared__ float red[3][36][64];
    const int W4 = W >> 2, HW = H * W, HW4 = H * W4;
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const bool q_ok = (int)blockIdx.x * 64 + lane < HW4;
    const int qi = min((int)blockIdx.x * 64 + lane, HW4 - 1);
    const int g = blockIdx.y, n = blockIdx.z, G = C / 4;
    const int y = qi / W4, x0 = (qi - y * W4) * 4;
    const int per = (G + 3) / 4, c0 = slice * per, c1 = min(c0 + per, G);
    const float* l = f1 + ((long)n * C + (long)g * G + c0) * HW + y * W + x0;
    const float* r = warped + ((long)n * C + (long)g * G + c0) * HW;
    float acc[4][9];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[j][k] = 0.f;
    auto ld4 = [](const float* p) { return *reinterpret_cast<const float4*>(p); };
    if (!SMALL) {
        const int oL = y * W + max(x0 - 4, 0), oM = y * W + x0, oR = y * W + min(x0 + 4, W - 4);
        const bool eL = x0 == 0, eR = x0 == W - 4;
#pragma unroll 2
        for (int c = c0; c < c1; ++c) {
            const float4 lv = ld4(l);
            float4 L = ld4(r + oL), M = ld4(r + oM), R = ld4(r + oR);
            if (eL) L = make_float4(M.x, M.x, M.x, M.x);
            if (eR) R = make_float4(M.w, M.w, M.w, M.w);
            const float v[12] = {L.x, L.y, L.z, L.w, M.x, M.y, M.z, M.w, R.x, R.y, R.z, R.w};
            const float lj[4] = {lv.x, lv.y, lv.z, lv.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int k = 0; k < 9; ++k) acc[j][k] += lj[j] * v[j + k];
            l += HW;
            r += HW;
        }
    } else {
        int orow[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) orow[d] = min(max(y + d - 1, 0), H - 1) * W;
        const int xl = max(x0 - 1, 0), xr = min(x0 + 4, W - 1);
#pragma unroll 2
        for (int c = c0; c < c1; ++c) {
            const float4 lv = ld4(l);
            const float lj[4] = {lv.x, lv.y, lv.z, lv.w};
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const float4 M = ld4(r + orow[d] + x0);
                const float v[6] = {r[orow[d] + xl], M.x, M.y, M.z, M.w, r[orow[d] + xr]};
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 3; ++e) acc[j][d * 3 + e] += lj[j] * v[j + e];
            }
            l += HW;
            r += HW;
        }
    }
    if (slice > 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 9; ++k) red[slice - 1][j * 9 + k][lane] = acc[j][k];
    }
    __syncthreads();
    if (slice == 0 && q_ok) {
        const float inv = (float)G;
        // x0 % 4 == 0: the 4 pixels are contiguous in NCHW and inside one 8-wide sub-tile row of the tile-major layout
        float* o = out + ((long)n * 36 + g * 9) * out_lay.plane + pix_off(out_lay, y, x0);
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            float t[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                t[j] = (((acc[j][k] + red[0][j * 9 + k][lane]) + red[1][j * 9 + k][lane]) + red[2][j * 9 + k][lane]) / inv;
            *reinterpret_cast<float4*>(o + (long)k * out_lay.plane) = make_float4(t[0], t[1], t[2], t[3]);
        }
    }
}

// (N, C, P) -> (N, P, C): the channels-last copies of the feature maps for offset_corr_cl_kernel.  64 x 64 tiles through
// LDS, both sides coalesced.   grid: (ceil(P/64), ceil(C/64), N), 256 threads
__global__ void __launch_bounds__(256) nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int P) {
    __shared__ float tile[64][65];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64, n = blockIdx.z;
#pragma unroll 4
    for (int i = w; i < 64; i += 4) {
        const int c = c0 + i, p = p0 + lane;
        tile[i][lane] = (c < C && p < P) ? in[((long)n * C + c) * P + p] : 0.f;
    }
    __syncthreads();
#pragma unroll 4
    for (int i = w; i < 64; i += 4) {
        const int p = p0 + i, c = c0 + lane;
        if (p < P && c < C) out[((long)n * P + p) * C + c] = tile[lane][i];
    }
}

// sum over the 16 lanes of a DPP row (every lane of the row gets the total)
__device__ __forceinline__ float row16_sum(float v) {
    auto dpp = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xF, 0xF, true));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]
    v += dpp(v, std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]
    v += dpp(v, std::integral_constant<int, 0x141>{});  // row_half_mirror
    v += dpp(v, std::integral_constant<int, 0x140>{});  // row_mirror
    return v;
}

// OFFSET mode on channels-last maps f1c, f2c (N, H*W, 256): block = PB consecutive pixels, 4 waves.
//   phase A: the PB x 9 tap sets (clamped pixel offsets + zero-bordered weights), one per thread, into LDS;
//   phase B: wave w takes pixels w*PB/4 .. one at a time: lane = channels 4*lane .. 4*lane+3 (16 lanes = one of the 4
//            correlation groups), so each of the 4 taps of a position is ONE 1-KB line read by the whole wave as 16-byte
//            loads; per-lane partial dot products over 4 channels, then a 16-lane DPP sum;
//   phase C: the 36 x PB results leave through LDS with the pixel index fastest.
// A wave's pixels are a serial chain of load rounds, so PB is small (16, or 8 for maps of a few thousand pixels: the
// grid must put many waves on every CU to cover the L2 latency).
// grid: (ceil(HW/PB) rounded up to a multiple of 8, 1, N)
constexpr int CL_C = 256;
template <int PB>
__global__ void __launch_bounds__(256) offset_corr_cl_kernel(const float* __restrict__ f1c, const float* __restrict__ f2c,
                                                             const float* __restrict__ flow, const float* __restrict__ extra,
                                                             float* __restrict__ out, int H, int W, int small_patch, Lay fl_lay,
                                                             Lay out_lay) {
    static_assert(PB % 4 == 0 && 9 * PB <= 256 * 3, "pixels per block");
    __shared__ int4 tap_o[9][PB];
    __shared__ float4 tap_w[9][PB];
    __shared__ float res[36][PB + 1];
    // consecutive workgroup ids go round the 8 XCDs: XCD x takes the x-th eighth of the map (a band of rows), so the lines
    // its pixels sample stay in its own L2 (the whole 33-MB map does not fit, a band does).  gridDim.x is a multiple of 8.
    const int HW = H * W, n = blockIdx.z;
    const int p0 = ((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) * PB;
    if (p0 >= HW) return;  // whole workgroup, before any barrier
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 9 * PB; i += 256) {
        const int k = i / PB, pl = i - k * PB;
        const int p = min(p0 + pl, HW - 1);
        const int y = p / W, x = p - y * W;
        const float* fl = flow + (long)n * 2 * fl_lay.plane + pix_off(fl_lay, y, x);
        const float flx = fl[0], fly = fl[fl_lay.plane];
        int dy, dx;
        window_offset(k, small_patch != 0, dy, dx);
        const float ex = extra[((long)n * 18 + 2 * k) * HW + p], ey = extra[((long)n * 18 + 2 * k + 1) * HW + p];
        const float sx = ((float)x + flx) + ((float)dx + ex);
        const float sy = ((float)y + fly) + ((float)dy + ey);
        const Taps t = make_taps(sx, sy, H, W);
        tap_o[k][pl] = make_int4(t.o00, t.o01, t.o10, t.o11);
        tap_w[k][pl] = make_float4(t.w00, t.w01, t.w10, t.w11);
    }
    __syncthreads();
    const float4* f1n = reinterpret_cast<const float4*>(f1c + (long)n * HW * CL_C);
    const float4* f2n = reinterpret_cast<const float4*>(f2c + (long)n * HW * CL_C);
    constexpr int PW_ = PB / 4;
    const int npix = min(PW_, HW - (p0 + wv * PW_));  // wave-uniform; <= 0 for waves past the end
    for (int pi = 0; pi < npix; ++pi) {
        const int pl = wv * PW_ + pi;
        const float4 lv = f1n[(unsigned)(p0 + pl) * (CL_C / 4) + lane];
#pragma unroll 3
        for (int k = 0; k < 9; ++k) {
            const int4 o = tap_o[k][pl];
            const float4 w = tap_w[k][pl];
            const float4 a = f2n[(unsigned)o.x * (CL_C / 4) + lane], b = f2n[(unsigned)o.y * (CL_C / 4) + lane];
            const float4 c = f2n[(unsigned)o.z * (CL_C / 4) + lane], d = f2n[(unsigned)o.w * (CL_C / 4) + lane];
            // per channel the reference's Ia*wa + Ib*wb + Ic*wc + Id*wd, then f1 * sample added channel by channel
            float part = lv.x * (a.x * w.x + b.x * w.y + c.x * w.z + d.x * w.w);
            part += lv.y * (a.y * w.x + b.y * w.y + c.y * w.z + d.y * w.w);
            part += lv.z * (a.z * w.x + b.z * w.y + c.z * w.z + d.z * w.w);
            part += lv.w * (a.w * w.x + b.w * w.y + c.w * w.z + d.w * w.w);
            const float tot = row16_sum(part);
            if ((lane & 15) == 0) res[(lane >> 4) * 9 + k][pl] = tot;
        }
    }
    __syncthreads();
    const float inv = (float)(CL_C / 4);
    for (int i = tid; i < 36 * PB; i += 256) {
        const int ch = i / PB, pl = i - ch * PB, p = p0 + pl;
        if (p < HW) {
            const int y = p / W, x = p - y * W;
            out[((long)n * 36 + ch) * out_lay.plane + pix_off(out_lay, y, x)] = res[ch][pl] / inv;
        }
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
    if (W % 4 == 0 && !switches().agcl_v1) {
        const dim3 grid(cdiv(HW / 4, 64), 4, N);
        if (small_patch) hipLaunchKernelGGL(window_corr_x4_kernel<true>, grid, dim3(256), 0, s, f1, (const float*)warped, out, C, H, W, lay);
        else hipLaunchKernelGGL(window_corr_x4_kernel<false>, grid, dim3(256), 0, s, f1, (const float*)warped, out, C, H, W, lay);
    } else {
        hipLaunchKernelGGL(window_corr_kernel, dim3(cdiv(HW, 64), 4, N), dim3(256), 0, s, f1, (const float*)warped, out, C, H, W,
                           small_patch, lay);
    }
    NND_LAUNCH_CHECK();
    return NND_OK;
}

int nchw_to_nhwc_launch(const float* in, float* out, int N, int C, int P, hipStream_t s) {
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(cdiv(P, 64), cdiv(C, 64), N), dim3(256), 0, s, in, out, C, P);
    NND_LAUNCH_CHECK();
    return NND_OK;
}

bool agcl_offset_cl_supported(int C) { return C == CL_C; }

int agcl_offset_cl_launch(const float* f1c, const float* f2c, const float* flow, const float* extra, float* out, int N, int C,
                          int H, int W, int small_patch, hipStream_t s, bool tiled) {
    NND_REQUIRE(agcl_offset_cl_supported(C), "agcl offset (channels-last): built for %d channels, got %d", CL_C, C);
    NND_REQUIRE((long)H * W * (CL_C / 4) < (1L << 31), "agcl offset (channels-last): offsets exceed 32 bits");
    const Lay lay = make_lay(H, W, tiled);
    const int force_pb = switches().agcl_pb;  // tuning only
    if (force_pb == 16)
        hipLaunchKernelGGL(offset_corr_cl_kernel<16>, dim3(cdiv(cdiv(H * W, 16), 8) * 8, 1, N), dim3(256), 0, s, f1c, f2c, flow, extra,
                           out, H, W, small_patch, lay, lay);
    else
        hipLaunchKernelGGL(offset_corr_cl_kernel<8>, dim3(cdiv(cdiv(H * W, 8), 8) * 8, 1, N), dim3(256), 0, s, f1c, f2c, flow, extra,
                           out, H, W, small_patch, lay, lay);
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

int nnd_nchw_to_nhwc(const float* in, float* out, int N, int C, int H, int W, void* stream) {
    NND_REQUIRE(in && out && in != out, "nchw_to_nhwc: null pointer / in-place");
    NND_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0, "nchw_to_nhwc: bad shape");
    return nchw_to_nhwc_launch(in, out, N, C, H * W, (hipStream_t)stream);
}

int nnd_agcl_corr_offset_nhwc(const float* fmap1_nhwc, const float* fmap2_nhwc, const float* flow, const float* extra_offset,
                              float* out, int N, int C, int H, int W, int small_patch, void* stream) {
    NND_REQUIRE(fmap1_nhwc && fmap2_nhwc && flow && extra_offset && out, "agcl_corr_offset_nhwc: null pointer");
    int rc = check_agcl("agcl_corr_offset_nhwc", N, C, H, W);
    if (rc != NND_OK) return rc;
    return agcl_offset_cl_launch(fmap1_nhwc, fmap2_nhwc, flow, extra_offset, out, N, C, H, W, small_patch, (hipStream_t)stream, false);
}

}  // extern "C"
