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
#include <cstdlib>
#include <cstring>
#include <vector>

namespace nnd {

// Thin layers (Cout 8 / 16 use only a quarter / half of the 32 MFMA rows): J adjacent output slices are computed by one
// launch element — their (J+2)-slice input windows overlap, so they become J*Cout output channels of a conv over
// (J+2)*Cin planes whose weights are zero where tap kd - j falls outside 0..2.  Half the K is zeros for J = 4, but the
// rows are full: 2x fewer MFMAs for Cout = 8, 1.5x fewer for Cout = 16.  Stride 1 only; needs D % J == 0.
static int group_of(const nnd_conv3d_desc* d) {
    if (d->stride != 1) return 1;
    return d->Cout <= 8 ? 4 : (d->Cout <= 16 ? 2 : 1);
}

static int conv3d_layer(const nnd_conv3d_desc* d, int J, ConvLayer* L, int64_t* total) {
    NND_REQUIRE(d, "conv3d: null descriptor");
    NND_REQUIRE(d->struct_size == (int32_t)sizeof(nnd_conv3d_desc), "conv3d: descriptor of %d bytes, this library expects %d (struct_size)",
                d->struct_size, (int)sizeof(nnd_conv3d_desc));
    NND_REQUIRE((d->flags & ~NND_FLAG_CALIBRATE) == 0, "conv3d: unknown flags 0x%x", d->flags);
    NND_REQUIRE(d->Cout > 0 && d->Cin0 > 0 && d->Cin1 >= 0, "conv3d: bad channel counts");
    NND_REQUIRE(d->stride == 1 || d->stride == 2, "conv3d: stride %d not supported (1, 2)", d->stride);
    NND_REQUIRE(d->arithmetic == 0 || d->arithmetic == 3 || d->arithmetic == 2, "conv3d: arithmetic must be 0 (fp32 MFMA), 3 (bf16x3) or 2 (fp16x2)");
    ConvLayer l;
    l.KH = l.KW = 3;
    l.Cin = (J + 2) * (d->Cin0 + d->Cin1);
    l.Cout = J * d->Cout;
    l.stride = d->stride;
    // split arithmetics: layers (stride 1 and 2) whose plane count is a multiple of 16 on the 16-bit MFMA kernel (conv_split)
    l.arith = (d->arithmetic != 0 && conv_split_supported(3, 3, l.Cin, d->stride, d->arithmetic, l.Cout) &&
               (d->Cin1 == 0 || ((J + 2) * d->Cin0) % 16 == 0)) ? d->arithmetic : 0;
    l.CI_T = 16;  // windows of 3*Cin planes: 24, 48, 96, 192 -> 16-channel chunks keep the two sources chunk-aligned
    NND_REQUIRE(d->Cin1 == 0 || ((J + 2) * d->Cin0) % l.CI_T == 0, "conv3d: first input of a concat needs (J+2)*Cin0 %% 16 == 0 (Cin0 = %d)", d->Cin0);
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

// The same two conversions for a volume whose candidate axis is the CONTIGUOUS one, (N, C, H, W, D) — the rows of IGEV's
// correlation pyramids (level 0 = (B, G, H, W1, W2)): an LDS-tiled transpose between the (h,w) and d axes, so the
// regulariser reads the feature volume and writes the geometry volume where the pyramids keep them (no permuted copies).
// grid (ceil(HW/64), ceil(D/64), N*C), block 256: 64x64 tiles — a wave reads / writes 256 contiguous bytes per row and every thread
// has its 16 loads in flight before the barrier (32x32 tiles: 128-byte rows, 4 loads per thread, 3.4 TB/s)
constexpr int RDM_T = 64;
template <bool TO_DM>
__global__ void __launch_bounds__(256) rows_depth_major_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int D,
                                                               long HW) {
    __shared__ float tile[RDM_T][RDM_T + 1];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const long p0 = (long)blockIdx.x * RDM_T;
    const int d0 = blockIdx.y * RDM_T, n = blockIdx.z / C, c = blockIdx.z % C;
    const long rows_base = ((long)n * C + c) * HW;                 // row index of pixel 0 in the (N,C,HW,D) volume
    const long dm_base = ((long)n * (D + 2) + 1) * C + c;          // plane index of slice d = 0 in the depth-major volume
    float v[RDM_T / 4];
#pragma unroll
    for (int k = 0; k < RDM_T / 4; ++k) {
        const int r = ty + 4 * k;
        v[k] = 0.f;
        if (TO_DM) {  // read rows: lanes along d
            const long p = p0 + r;
            const int d = d0 + tx;
            if (p < HW && d < D) v[k] = src[(rows_base + p) * D + d];
        } else {      // read planes: lanes along p
            const int d = d0 + r;
            const long p = p0 + tx;
            if (p < HW && d < D) v[k] = src[(dm_base + (long)d * C) * HW + p];
        }
    }
#pragma unroll
    for (int k = 0; k < RDM_T / 4; ++k) tile[ty + 4 * k][tx] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RDM_T / 4; ++k) {
        const int r = ty + 4 * k;
        if (TO_DM) {  // write planes: lanes along p
            const int d = d0 + r;
            const long p = p0 + tx;
            if (p < HW && d < D) dst[(dm_base + (long)d * C) * HW + p] = tile[tx][r];
        } else {      // write rows: lanes along d
            const long p = p0 + r;
            const int d = d0 + tx;
            if (p < HW && d < D) dst[(rows_base + p) * D + d] = tile[tx][r];
        }
    }
}

// ATen compute_source_index_and_lambda for align_corners = true: real = scale * dst, scale = (in - 1) / (out - 1)
__device__ __forceinline__ void src_index_ac(float scale, int dst, int in_size, int& i0, int& i1, float& l0, float& l1) {
#pragma clang fp contract(off)  // the product is rounded before the subtraction as in ATen: contracted into an FMA, the weight would carry up to half an ulp of `real` (4e-6 at 120; HIP's __fmul_rn is a plain product and does not prevent it)
    const float real = scale * (float)dst;
    i0 = min((int)floorf(real), in_size - 1);
    l1 = fminf(fmaxf(real - (float)i0, 0.f), 1.f);
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l0 = 1.f - l1;
}

// F.interpolate(x, scale_factor=2, mode="trilinear", align_corners=True) of Upsampler3D (cost_volume.py:128) on depth-major
// volumes: x (N, D+2, C, H, W) -> y (N, 2D+2, C, 2H, 2W) incl. its zero end slices.  One workgroup = UP_R output rows of one
// (n, slice, channel) plane: the <= UP_R/2+2 source rows of the two source slices are staged in LDS with coalesced loads
// (0.8 global loads per output instead of 8 cached gathers, which made the kernel texture-address bound), then every
// output takes its 8 taps from LDS with ATen's weights and summation order (thread = output column, rows looped).
// A band of UP_R = 32 output rows per workgroup (8 when the staged rows would not fit 64 KB of LDS): a workgroup's life is the
// serial chain load -> barrier -> 8 LDS taps per output -> store, and with 8-row bands (7.7 KB written per workgroup, 131 k
// workgroups per sample at 120x68x120) the kernel was bound by that chain's latency, not by HBM (1.6 TB/s); thread = two
// adjacent output columns (one 8-byte store).
template <int UP_R>
__global__ void __launch_bounds__(256) trilinear_up2_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int D, int H,
                                                            int W) {
    constexpr int UP_SR = UP_R / 2 + 2;
    extern __shared__ float sm[];  // [2 slices][UP_SR rows][W]
    const int Do = 2 * D, Ho = 2 * H, Wo = 2 * W;
    const int oy0 = blockIdx.x * UP_R, nrow = min(UP_R, Ho - oy0);
    const int c = blockIdx.y % C, dp = blockIdx.y / C, n = blockIdx.z;  // dp: padded output slice 0..Do+1
    float* o = y + ((((long)n * (Do + 2) + dp) * C + c) * Ho + oy0) * Wo;
    if (dp == 0 || dp == Do + 1) {
        for (int i = threadIdx.x; i < nrow * W; i += 256) reinterpret_cast<float2*>(o)[i] = make_float2(0.f, 0.f);
        return;
    }
    int d0, d1, ys, yt;
    float ld0, ld1, lt0, lt1;
    const float sy = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f, sx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    src_index_ac(Do > 1 ? (float)(D - 1) / (float)(Do - 1) : 0.f, dp - 1, D, d0, d1, ld0, ld1);
    src_index_ac(sy, oy0, H, ys, yt, lt0, lt1);  // ys: first source row of the band
    const long HW = (long)H * W;
    const float* p0 = x + (((long)n * (D + 2) + d0 + 1) * C + c) * HW;
    const float* p1 = x + (((long)n * (D + 2) + d1 + 1) * C + c) * HW;
    const int nsr = min(UP_SR, nrow / 2 + 2);  // source rows the band can touch
    {  // staging: UP_LD loads in flight per thread before their LDS stores (one load per round trip left the kernel bound by load
       // latency: ~17 serial round trips per workgroup); (row, column) of element i = tid + 256 k advance without divisions
        constexpr int UP_LD = 9;
        const int total = 2 * nsr * W, qs = 256 / W, ms = 256 - qs * W;
        int r = (int)threadIdx.x / W, xx = (int)threadIdx.x - r * W;
        for (int base = threadIdx.x; base < total; base += 256 * UP_LD) {
            float v[UP_LD];
            int dsti[UP_LD];
#pragma unroll
            for (int k = 0; k < UP_LD; ++k) {
                const bool in = base + 256 * k < total;
                const int sl = r >= nsr, rr = r - sl * nsr;
                dsti[k] = in ? (sl * UP_SR + rr) * W + xx : -1;
                v[k] = in ? (sl ? p1 : p0)[(long)min(ys + rr, H - 1) * W + xx] : 0.f;
                xx += ms;
                r += qs;
                if (xx >= W) xx -= W, ++r;
            }
#pragma unroll
            for (int k = 0; k < UP_LD; ++k)
                if (dsti[k] >= 0) sm[dsti[k]] = v[k];
        }
    }
    __syncthreads();
    // thread = a pair of output columns (their x taps are fixed); narrow planes: 256 / W row groups share the block
    const int nrg = W < 256 ? 256 / W : 1, rg = W < 256 ? (int)threadIdx.x / W : 0;
    for (int op = W < 256 ? (int)threadIdx.x % W : (int)threadIdx.x; op < W && rg < nrg; op += 256) {
        int xa0, xa1, xb0, xb1;
        float la0, la1, lb0, lb1;
        src_index_ac(sx, 2 * op, W, xa0, xa1, la0, la1);
        src_index_ac(sx, 2 * op + 1, W, xb0, xb1, lb0, lb1);
        for (int r = rg; r < nrow; r += nrg) {
            int y0, y1;
            float ly0, ly1;
            src_index_ac(sy, oy0 + r, H, y0, y1, ly0, ly1);
            const float* r0 = sm + (y0 - ys) * W;
            const float* r1 = sm + (y1 - ys) * W;
            auto plane = [&](const float* a, const float* b, int x0, int x1, float lx0, float lx1) {
                const float t0 = lx0 * a[x0] + lx1 * a[x1];
                const float t1 = lx0 * b[x0] + lx1 * b[x1];
                return ly0 * t0 + ly1 * t1;
            };
            float2 v;
            v.x = ld0 * plane(r0, r1, xa0, xa1, la0, la1) + ld1 * plane(r0 + UP_SR * W, r1 + UP_SR * W, xa0, xa1, la0, la1);
            v.y = ld0 * plane(r0, r1, xb0, xb1, lb0, lb1) + ld1 * plane(r0 + UP_SR * W, r1 + UP_SR * W, xb0, xb1, lb0, lb1);
            *reinterpret_cast<float2*>(o + r * Wo + 2 * op) = v;
        }
    }
}

// FeatureGuidedBlock (cost_volume.py:133-147): vol[n, d, c, h, w] *= sigmoid(logit[n, c, h, w]) for every depth slice, in place
// grid (ceil(HW/256), C * nchunk, N): a block scales GATE_DS = 8 consecutive slices of its channel (8 loads in flight per thread,
// then 8 stores); one block per channel walking all D slices left 2 workgroups per CU (3.0 TB/s)
constexpr int GATE_DS = 8;
__global__ void __launch_bounds__(256) gate_kernel(float* __restrict__ vol, const float* __restrict__ logit, int C, int D, long HW) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= HW) return;
    const int c = blockIdx.y % C, d0 = (blockIdx.y / C) * GATE_DS, n = blockIdx.z;
    const float g = 1.0f / (1.0f + expf(-logit[((long)n * C + c) * HW + i]));
    float* p = vol + (((long)n * (D + 2) + 1 + d0) * C + c) * HW + i;
    const long st = (long)C * HW;
    float v[GATE_DS];
#pragma unroll
    for (int k = 0; k < GATE_DS; ++k)
        if (d0 + k < D) v[k] = p[k * st];
#pragma unroll
    for (int k = 0; k < GATE_DS; ++k)
        if (d0 + k < D) p[k * st] = v[k] * g;
}

}  // namespace nnd

using namespace nnd;

extern "C" {

int64_t nnd_conv3d_packed_floats(const nnd_conv3d_desc* desc) {
    ConvLayer L;
    int64_t t1, tj = 0;
    if (conv3d_layer(desc, 1, &L, &t1) != NND_OK) return NND_ERR_INVALID;
    const int J = group_of(desc);
    if (J > 1 && conv3d_layer(desc, J, &L, &tj) != NND_OK) return NND_ERR_INVALID;
    const int64_t tt = thin3d_supported(desc->Cout, desc->stride) ? thin3d_packed_floats(desc->Cout, desc->Cin0 + desc->Cin1) : 0;
    const int64_t ts = slab3d_supported(desc->Cout, desc->Cin0, desc->Cin1, desc->stride, desc->arithmetic)
                           ? slab3d_packed_floats(desc->Cout, desc->Cin0 + desc->Cin1, desc->stride) : 0;
    return t1 + tj + tt + ts;  // [plain layer | J-slice grouped layer | thin-layer VALU kernel | depth-marching MFMA kernel]
}

// Which formulation a thin (Cout 8 / 16) layer takes.  fp16x2 at stride 1 (conv1.1, conv2_up, proj_2, conv1_up, final_conv):
// use_slab above — the depth-marching MFMA kernel (round 3: 99 / 151 / 153 / 457 / 216 us against the numbers below).  Exact arithmetic: the direct VALU kernel of thin3d.hip on all of them
// (NND_NO_THIN3D, diagnostic: the MFMA formulations above).  Split arithmetics: the J-slice grouped layer on the 16-bit MFMA
// (conv_split, FAST regime on the depth-major slabs) wherever it is the faster one — measured per layer at 544x960 with fp16x2
// (profiles/r03_igev_regulariser_layers_*.txt, us thin / MFMA): conv1.1 16->16 288 / 198, conv2_up 32->16 540 / 268, proj_2
// 32->16 547 / 277, conv1_up 16->8 1020 / 876, but final_conv 8->8 533 / 600 (K = 6 x 8 x 9 is too short for the workgroup's
// fixed phases) and the stride-2 layers 330 / 1041 (no stride-2 split kernel): those two kinds stay on the VALU kernel.
// fp16x2, the regulariser's thin (Cin, Cout, stride) triples: the depth-marching MFMA kernel of slab3d.hip (NND_NO_SLAB3D: the rules below)
static bool use_slab(const nnd_conv3d_desc* d) {
    return !switches().no_slab3d && slab3d_supported(d->Cout, d->Cin0, d->Cin1, d->stride, d->arithmetic);
}

static bool use_thin(const nnd_conv3d_desc* d) {
    if (switches().no_thin3d || !thin3d_supported(d->Cout, d->stride)) return false;
    if (d->arithmetic == 0 || d->stride != 1) return true;
    return d->Cin0 + d->Cin1 <= 8;
}

// 2-D weights of the J-slice grouped layer: (J*Cout, (J+2)*Ct, 3, 3) with [window of input 0: slice-major, ci][window of input 1]
static void pack_one(const nnd_conv3d_desc* desc, const ConvLayer& L, int J, const float* w, const float* bias, const float* g,
                     const float* be, const float* mean, const float* var, float eps, float* base) {
    const int C0 = desc->Cin0, C1 = desc->Cin1, Ct = C0 + C1, Co = desc->Cout, S = J + 2;
    std::vector<float> w2((size_t)J * Co * S * Ct * 9, 0.f);
    for (int j = 0; j < J; ++j)
        for (int co = 0; co < Co; ++co)
            for (int ci = 0; ci < Ct; ++ci)
                for (int kd = 0; kd < 3; ++kd) {
                    const int sl = j + kd;  // slice of the window this tap reads
                    const int c2 = ci < C0 ? sl * C0 + ci : S * C0 + sl * C1 + (ci - C0);
                    for (int t = 0; t < 9; ++t)
                        w2[((size_t)(j * Co + co) * S * Ct + c2) * 9 + t] = w[(((size_t)co * Ct + ci) * 3 + kd) * 9 + t];
                }
    const float* ws[1] = {w2.data()};
    const float* bs[1] = {nullptr};
    int cc[1] = {J * Co};
    pack_conv(L, 1, ws, bs, cc, base);
    float* shift = base + L.b_off;
    float* scale = base + L.s_off;
    for (int c = 0; c < L.ncb * 32; ++c) {
        double sc = 1.0, sh = 0.0;
        if (c < J * Co) {
            const int co = c % Co;
            const double b = bias ? (double)bias[co] : 0.0;
            if (g) {
                sc = (double)g[co] / std::sqrt((double)var[co] + (double)eps);
                sh = (b - (double)mean[co]) * sc + (double)be[co];
            } else {
                sh = b;
            }
        }
        scale[c] = (float)sc;
        shift[c] = (float)sh;
    }
}

// w (Cout, Cin0+Cin1, 3, 3, 3) [kd, kh, kw]; bias may be NULL (the reference's ConvBn3D has bias=False); bn_* may be NULL
int nnd_conv3d_pack(const nnd_conv3d_desc* desc, const float* w, const float* bias, const float* bn_gamma, const float* bn_beta,
                    const float* bn_mean, const float* bn_var, float bn_eps, float* packed_host) {
    ConvLayer L1, LJ;
    int64_t t1;
    int rc = conv3d_layer(desc, 1, &L1, &t1);
    if (rc != NND_OK) return rc;
    NND_REQUIRE(w && packed_host, "conv3d_pack: null pointer");
    NND_REQUIRE(!bn_gamma || (bn_beta && bn_mean && bn_var), "conv3d_pack: incomplete batch-norm parameters");
    pack_one(desc, L1, 1, w, bias, bn_gamma, bn_beta, bn_mean, bn_var, bn_eps, packed_host);
    const int J = group_of(desc);
    int64_t tj = 0;
    if (J > 1) {
        if ((rc = conv3d_layer(desc, J, &LJ, &tj)) != NND_OK) return rc;
        pack_one(desc, LJ, J, w, bias, bn_gamma, bn_beta, bn_mean, bn_var, bn_eps, packed_host + t1);
    }
    const bool thin = thin3d_supported(desc->Cout, desc->stride);
    const bool slab = slab3d_supported(desc->Cout, desc->Cin0, desc->Cin1, desc->stride, desc->arithmetic);
    if (thin || slab) {  // same folded affine; weights in [ci][tap][co] order (VALU kernel) / as A fragments (depth-marching kernel)
        std::vector<float> sc(desc->Cout), sh(desc->Cout);
        for (int co = 0; co < desc->Cout; ++co) {
            const double b = bias ? (double)bias[co] : 0.0;
            double s1 = 1.0, s0 = b;
            if (bn_gamma) {
                s1 = (double)bn_gamma[co] / std::sqrt((double)bn_var[co] + (double)bn_eps);
                s0 = (b - (double)bn_mean[co]) * s1 + (double)bn_beta[co];
            }
            sc[co] = (float)s1;
            sh[co] = (float)s0;
        }
        const int64_t tt = thin ? thin3d_packed_floats(desc->Cout, desc->Cin0 + desc->Cin1) : 0;
        if (thin) thin3d_pack(desc->Cout, desc->Cin0 + desc->Cin1, w, sc.data(), sh.data(), packed_host + t1 + tj);
        if (slab) slab3d_pack(desc->Cout, desc->Cin0 + desc->Cin1, desc->stride, w, sc.data(), sh.data(), packed_host + t1 + tj + tt);
    }
    return NND_OK;
}

// x0 (N, D+2, Cin0, H, W), x1 (N, D+2, Cin1, H, W) or NULL, y (N, Do+2, Cout, Ho, Wo) — all depth-major with zero end slices
// (y's end slices are written by this call); Do = ceil(D/stride) etc.  leaky_slope: LeakyReLU negative slope (1 = none).
// every formulation of the layer that is packed in fp16x2 has its own slot; the one the forwards did not take reports "staged nothing"
int nnd_conv3d_calibration_finish(const nnd_conv3d_desc* desc, float* packed_dev, int32_t* status_dev, void* stream) {
    ConvLayer L1, LJ;
    int64_t t1, tj = 0;
    int rc = conv3d_layer(desc, 1, &L1, &t1);
    if (rc != NND_OK) return rc;
    NND_REQUIRE(packed_dev, "conv3d_calibration_finish: null blob");
    const int J = group_of(desc);
    if (J > 1 && (rc = conv3d_layer(desc, J, &LJ, &tj)) != NND_OK) return rc;
    int64_t offs[3];
    int n = 0;
    if (L1.arith == 2) offs[n++] = L1.tail_off();
    if (J > 1 && LJ.arith == 2) offs[n++] = t1 + LJ.tail_off();
    if (slab3d_supported(desc->Cout, desc->Cin0, desc->Cin1, desc->stride, desc->arithmetic)) {
        const int64_t tt = thin3d_supported(desc->Cout, desc->stride) ? thin3d_packed_floats(desc->Cout, desc->Cin0 + desc->Cin1) : 0;
        offs[n++] = t1 + tj + tt + slab3d_packed_floats(desc->Cout, desc->Cin0 + desc->Cin1, desc->stride) - 4;
    }
    return calib_finish(packed_dev, offs, n, status_dev, (hipStream_t)stream);
}

int nnd_conv3d_forward(const nnd_conv3d_desc* desc, const float* packed, const float* x0, const float* x1, float* y, int N, int D,
                       int H, int W, float leaky_slope, void* stream) {
    ConvLayer L;
    int64_t t1;
    int rc = conv3d_layer(desc, 1, &L, &t1);
    if (rc != NND_OK) return rc;
    NND_REQUIRE(packed && x0 && y && (desc->Cin1 == 0 || x1), "conv3d_forward: null pointer");
    NND_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0, "conv3d_forward: bad shape");
    CalibScope calib((desc->flags & NND_FLAG_CALIBRATE) && desc->arithmetic == 2);
    hipStream_t s = (hipStream_t)stream;
    if (use_slab(desc) || use_thin(desc)) {
        const int st0 = desc->stride;
        const int Do0 = (D + st0 - 1) / st0, Ho0 = (H + st0 - 1) / st0, Wo0 = (W + st0 - 1) / st0;
        const int64_t hwo0 = (int64_t)Ho0 * Wo0;
        for (int n = 0; n < N; ++n) {  // the zero end slices of the output
            float* yn = y + (int64_t)n * (Do0 + 2) * desc->Cout * hwo0;
            NND_HIP_CHECK(hipMemsetAsync(yn, 0, sizeof(float) * desc->Cout * hwo0, s));
            NND_HIP_CHECK(hipMemsetAsync(yn + (int64_t)(Do0 + 1) * desc->Cout * hwo0, 0, sizeof(float) * desc->Cout * hwo0, s));
        }
        int64_t tj0 = 0;
        const int J0 = group_of(desc);
        ConvLayer LJ0;
        if (J0 > 1 && (rc = conv3d_layer(desc, J0, &LJ0, &tj0)) != NND_OK) return rc;
        if (use_slab(desc)) {
            const int64_t tt0 = thin3d_supported(desc->Cout, st0) ? thin3d_packed_floats(desc->Cout, desc->Cin0 + desc->Cin1) : 0;
            return slab3d_forward(desc->Cout, desc->Cin0, desc->Cin1, st0, packed + t1 + tj0 + tt0, x0, x1, y, N, D, H, W, leaky_slope, s);
        }
        return thin3d_forward(desc->Cout, desc->Cin0, desc->Cin1, st0, packed + t1 + tj0, x0, x1, y, N, D, H, W, leaky_slope, s);
    }
    int J = group_of(desc);
    if (D % J != 0) J = 1;  // the grouped layer needs whole groups of slices; the plain one is always packed as well
    const float* blob = packed;
    if (J > 1) {
        if ((rc = conv3d_layer(desc, J, &L, nullptr)) != NND_OK) return rc;
        blob = packed + t1;
    }
    const int st = desc->stride;
    const int Do = (D + st - 1) / st, Ho = (H + st - 1) / st, Wo = (W + st - 1) / st;
    NND_REQUIRE(Do / J <= 65535, "conv3d_forward: depth %d exceeds the grid limit", Do);
    const int64_t hw = (int64_t)H * W, hwo = (int64_t)Ho * Wo;
    for (int n = 0; n < N; ++n) {
        float* yn = y + (int64_t)n * (Do + 2) * desc->Cout * hwo;
        NND_HIP_CHECK(hipMemsetAsync(yn, 0, sizeof(float) * desc->Cout * hwo, s));
        NND_HIP_CHECK(hipMemsetAsync(yn + (int64_t)(Do + 1) * desc->Cout * hwo, 0, sizeof(float) * desc->Cout * hwo, s));
        ConvIO io{};
        // launch element g computes output slices g*J .. g*J+J-1 from padded input slices st*g*J .. (+J+1)
        io.src0 = Act{const_cast<float*>(x0) + (int64_t)n * (D + 2) * desc->Cin0 * hw, (int64_t)st * J * desc->Cin0 * hw, (J + 2) * desc->Cin0};
        if (desc->Cin1 > 0)
            io.src1 = Act{const_cast<float*>(x1) + (int64_t)n * (D + 2) * desc->Cin1 * hw, (int64_t)st * J * desc->Cin1 * hw, (J + 2) * desc->Cin1};
        io.out0 = Act{yn + desc->Cout * hwo, (int64_t)J * desc->Cout * hwo, J * desc->Cout};
        io.Hin = H; io.Win = W;
        io.flags = leaky_slope != 1.0f ? 4 : 0;
        io.scale = leaky_slope;
        rc = launch_conv(L, blob, io, EPI_AFFINE, Do / J, Ho, Wo, s);
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

int nnd_volume_rows_to_depth_major(const float* x, float* y, int N, int C, int D, int H, int W, void* stream) {
    NND_REQUIRE(x && y && N > 0 && C > 0 && D > 0 && H > 0 && W > 0 && (long)N * C <= 65535, "volume_rows_to_depth_major: bad argument");
    const long HW = (long)H * W;
    hipStream_t s = (hipStream_t)stream;
    for (int n = 0; n < N; ++n) {  // the two zero end slices
        NND_HIP_CHECK(hipMemsetAsync(y + (long)n * (D + 2) * C * HW, 0, sizeof(float) * C * HW, s));
        NND_HIP_CHECK(hipMemsetAsync(y + ((long)n * (D + 2) + D + 1) * C * HW, 0, sizeof(float) * C * HW, s));
    }
    hipLaunchKernelGGL(rows_depth_major_kernel<true>, dim3((unsigned)cdiv64(HW, RDM_T), cdiv(D, RDM_T), N * C), dim3(256), 0, s, x, y, C, D, HW);
    NND_LAUNCH_CHECK();
    return NND_OK;
}

int nnd_depth_major_to_volume_rows(const float* x, float* y, int N, int C, int D, int H, int W, void* stream) {
    NND_REQUIRE(x && y && N > 0 && C > 0 && D > 0 && H > 0 && W > 0 && (long)N * C <= 65535, "depth_major_to_volume_rows: bad argument");
    const long HW = (long)H * W;
    hipLaunchKernelGGL(rows_depth_major_kernel<false>, dim3((unsigned)cdiv64(HW, RDM_T), cdiv(D, RDM_T), N * C), dim3(256), 0,
                       (hipStream_t)stream, x, y, C, D, HW);
    NND_LAUNCH_CHECK();
    return NND_OK;
}

int nnd_volume_upsample2x(const float* x, float* y, int N, int C, int D, int H, int W, void* stream) {
    NND_REQUIRE(x && y && N > 0 && C > 0 && D > 0 && H > 0 && W > 0 && (long)C * (2 * D + 2) <= 65535, "volume_upsample2x: bad argument");
    NND_REQUIRE((size_t)2 * 6 * W * sizeof(float) <= 64 * 1024, "volume_upsample2x: rows of %d floats do not fit the LDS staging", W);
    if ((size_t)2 * 18 * W * sizeof(float) <= 64 * 1024)
        hipLaunchKernelGGL(trilinear_up2_kernel<32>, dim3(cdiv(2 * H, 32), C * (2 * D + 2), N), dim3(256), 2 * 18 * W * sizeof(float),
                           (hipStream_t)stream, x, y, C, D, H, W);
    else
        hipLaunchKernelGGL(trilinear_up2_kernel<8>, dim3(cdiv(2 * H, 8), C * (2 * D + 2), N), dim3(256), 2 * 6 * W * sizeof(float),
                           (hipStream_t)stream, x, y, C, D, H, W);
    NND_LAUNCH_CHECK();
    return NND_OK;
}

int nnd_volume_gate(float* vol, const float* logits, int N, int C, int D, int H, int W, void* stream) {
    NND_REQUIRE(vol && logits && N > 0 && C > 0 && D > 0 && H > 0 && W > 0 && (long)C * cdiv(D, GATE_DS) <= 65535, "volume_gate: bad argument");
    const long HW = (long)H * W;
    hipLaunchKernelGGL(gate_kernel, dim3((unsigned)cdiv64(HW, 256), C * cdiv(D, GATE_DS), N), dim3(256), 0, (hipStream_t)stream, vol, logits, C, D,
                       HW);
    NND_LAUNCH_CHECK();
    return NND_OK;
}

}  // extern "C"
