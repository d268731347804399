// Pre- / post-processing of the inference and evaluation scripts on the device: no host round trip
// between decoding a frame and the first convolution, or between the last upsample and the metric.
//
//   resize_normalize : preprocess_frame  nndepth/models/raft_stereo/scripts/inference.py:55-60
//                      F.interpolate(frame, HW, mode="bilinear") (align_corners=False, no antialias) then (x - 127.5) / 127.5;
//                      source either float NCHW or the decoded uint8 HWC image itself
//   replicate_pad    : Padder.pad / unpad   nndepth/data/dataloaders/utils.py:5-21 (F.pad mode="replicate"; crop)
//   epe_metrics      : EvalCriterion.__call__  nndepth/models/raft_stereo/scripts/evaluate.py:48-83
//                      epe = sqrt(sum_c (pred - gt)^2), valid = |gt| < max_flow (& mask), mean EPE and fraction of
//                      valid pixels with epe > threshold_k
// All HBM-bound element-wise / reduction kernels.  Compiled with -ffp-contract=off (the bilinear weights follow
// ATen's compute_source_index_and_lambda step by step).
#include "common.h"

namespace nnd {

// ATen: area_pixel_compute_source_index (align_corners = false, not cubic) + guard_index_and_lambda
__device__ __forceinline__ void src_index(float scale, int dst, int in_size, int& i0, int& i1, float& l0, float& l1) {
    float real = fmaf(scale, (float)dst + 0.5f, -0.5f);  // ATen's x86 build contracts scale*(dst+0.5)-0.5 into one fma; the
                                                         // weight is sensitive to that rounding (measured against torch CPU)
    real = real < 0.f ? 0.f : real;
    i0 = min((int)floorf(real), in_size - 1);
    l1 = fminf(fmaxf(real - (float)i0, 0.f), 1.f);
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l0 = 1.f - l1;
}

// one thread per output pixel, loops the C channels; u8hwc: src is (B,h,w,C) uint8, else (B,C,h,w) float
__global__ void __launch_bounds__(256) resize_normalize_kernel(const void* __restrict__ src, float* __restrict__ dst, int C, int h,
                                                               int w, int H, int W, float sub, float div, int u8hwc) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)H * W) return;
    const int b = blockIdx.y;
    const int y = (int)(idx / W), x = (int)(idx - (long)y * W);
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    src_index(sy, y, h, y0, y1, ly0, ly1);
    src_index(sx, x, w, x0, x1, lx0, lx1);
    for (int c = 0; c < C; ++c) {
        float v00, v01, v10, v11;
        if (u8hwc) {
            const unsigned char* s = (const unsigned char*)src + (long)b * h * w * C;
            v00 = (float)s[((long)y0 * w + x0) * C + c];
            v01 = (float)s[((long)y0 * w + x1) * C + c];
            v10 = (float)s[((long)y1 * w + x0) * C + c];
            v11 = (float)s[((long)y1 * w + x1) * C + c];
        } else {
            const float* s = (const float*)src + ((long)b * C + c) * h * w;
            v00 = s[(long)y0 * w + x0];
            v01 = s[(long)y0 * w + x1];
            v10 = s[(long)y1 * w + x0];
            v11 = s[(long)y1 * w + x1];
        }
        // ATen's interpolate<2> accumulates `out += src * weight` per tap, which its AVX2 build contracts to fma
        const float t0 = fmaf(v01, lx1, v00 * lx0);
        const float t1 = fmaf(v11, lx1, v10 * lx0);
        const float v = fmaf(t1, ly1, t0 * ly0);
        dst[((long)b * C + c) * H * W + idx] = (v - sub) / div;
    }
}

// dst (B*C, Ho, Wo) <- src (B*C, H, W): dst[y, x] = src[clamp(y - top), clamp(x - left)]; a negative pad crops
__global__ void __launch_bounds__(256) replicate_pad_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int W,
                                                            int Ho, int Wo, int left, int top) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)Ho * Wo) return;
    const int y = (int)(idx / Wo), x = (int)(idx - (long)y * Wo);
    const int sy = min(max(y - top, 0), H - 1), sx = min(max(x - left, 0), W - 1);
    dst[(long)blockIdx.y * Ho * Wo + idx] = src[(long)blockIdx.y * H * W + (long)sy * W + sx];
}

constexpr int MAX_THR = 4;
struct EpeArgs {
    float thr[MAX_THR];
    int nthr;
    float max_flow;
};

// pass 1: per-block partial sums (double) of [epe, count, count(epe > thr_k)...]; pass 2: one block adds them in index order
__global__ void __launch_bounds__(256) epe_partial_kernel(const float* __restrict__ gt, const float* __restrict__ pred,
                                                          const unsigned char* __restrict__ mask, int C, long HW, long total,
                                                          EpeArgs a, double* __restrict__ partial) {
    __shared__ double sh[256][2 + MAX_THR];
    double acc[2 + MAX_THR];
#pragma unroll
    for (int i = 0; i < 2 + MAX_THR; ++i) acc[i] = 0.0;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long b = idx / HW, p = idx - b * HW;
        float se = 0.f, sg = 0.f;
        for (int c = 0; c < C; ++c) {
            const float g = gt[(b * C + c) * HW + p], d = pred[(b * C + c) * HW + p] - g;
            se += d * d;
            sg += g * g;
        }
        const float epe = sqrtf(se);
        bool valid = sqrtf(sg) < a.max_flow;
        if (mask) valid = valid && mask[idx] != 0;
        if (valid) {
            acc[0] += (double)epe;
            acc[1] += 1.0;
            for (int k = 0; k < a.nthr; ++k) acc[2 + k] += epe > a.thr[k] ? 1.0 : 0.0;
        }
    }
#pragma unroll
    for (int i = 0; i < 2 + MAX_THR; ++i) sh[threadIdx.x][i] = acc[i];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
#pragma unroll
            for (int i = 0; i < 2 + MAX_THR; ++i) sh[threadIdx.x][i] += sh[threadIdx.x + s][i];
        __syncthreads();
    }
    if (threadIdx.x < 2 + MAX_THR) partial[(long)blockIdx.x * (2 + MAX_THR) + threadIdx.x] = sh[0][threadIdx.x];
}

__global__ void epe_final_kernel(const double* __restrict__ partial, int nblocks, int nthr, float* __restrict__ out) {
    const int i = threadIdx.x;  // 0: epe sum, 1: count, 2..: exceed counts
    if (i >= 2 + nthr) return;
    double s = 0.0;
    for (int b = 0; b < nblocks; ++b) s += partial[(long)b * (2 + MAX_THR) + i];
    __shared__ double tot[2 + MAX_THR];
    tot[i] = s;
    __syncthreads();
    if (i == 0) out[0] = (float)(tot[0] / tot[1]);
    else if (i == 1) out[1] = (float)tot[1];
    else out[i] = (float)(tot[i] / tot[1]);
}

constexpr int EPE_BLOCKS = 512;

}  // namespace nnd

using namespace nnd;

extern "C" {

int nnd_resize_normalize(const void* src, int src_is_u8_hwc, float* dst, int B, int C, int h, int w, int H, int W, float sub,
                         float div, void* stream) {
    NND_REQUIRE(src && dst, "resize_normalize: null pointer");
    NND_REQUIRE(B > 0 && C > 0 && h > 0 && w > 0 && H > 0 && W > 0 && div != 0.f, "resize_normalize: bad argument");
    hipLaunchKernelGGL(resize_normalize_kernel, dim3((unsigned)cdiv64((int64_t)H * W, 256), B), dim3(256), 0, (hipStream_t)stream, src,
                       dst, C, h, w, H, W, sub, div, src_is_u8_hwc);
    NND_LAUNCH_CHECK();
    return NND_OK;
}

int nnd_replicate_pad(const float* src, float* dst, int B, int C, int H, int W, int left, int right, int top, int bottom,
                      void* stream) {
    NND_REQUIRE(src && dst && B > 0 && C > 0 && H > 0 && W > 0, "replicate_pad: bad argument");
    const int Ho = H + top + bottom, Wo = W + left + right;
    NND_REQUIRE(Ho > 0 && Wo > 0, "replicate_pad: padding (%d,%d,%d,%d) leaves nothing of a %dx%d map", left, right, top, bottom, H, W);
    hipLaunchKernelGGL(replicate_pad_kernel, dim3((unsigned)cdiv64((int64_t)Ho * Wo, 256), B * C), dim3(256), 0, (hipStream_t)stream,
                       src, dst, H, W, Ho, Wo, left, top);
    NND_LAUNCH_CHECK();
    return NND_OK;
}

int64_t nnd_epe_metrics_workspace_bytes(void) { return (int64_t)EPE_BLOCKS * (2 + MAX_THR) * sizeof(double); }

int nnd_epe_metrics(const float* disp_gt, const float* disp_pred, const unsigned char* valid_mask, int B, int C, int H, int W,
                    float max_flow, const float* thresholds, int num_thresholds, void* workspace, float* out, void* stream) {
    NND_REQUIRE(disp_gt && disp_pred && workspace && out, "epe_metrics: null pointer");
    NND_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "epe_metrics: bad shape");
    NND_REQUIRE(num_thresholds >= 0 && num_thresholds <= MAX_THR && (num_thresholds == 0 || thresholds),
                "epe_metrics: at most %d thresholds", MAX_THR);
    EpeArgs a;
    a.nthr = num_thresholds;
    a.max_flow = max_flow;
    for (int k = 0; k < MAX_THR; ++k) a.thr[k] = k < num_thresholds ? thresholds[k] : 0.f;
    const long HW = (long)H * W, total = (long)B * HW;
    const int nblocks = (int)(cdiv64(total, 256) < EPE_BLOCKS ? cdiv64(total, 256) : EPE_BLOCKS);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(epe_partial_kernel, dim3(nblocks), dim3(256), 0, s, disp_gt, disp_pred, valid_mask, C, HW, total, a,
                       (double*)workspace);
    NND_LAUNCH_CHECK();
    hipLaunchKernelGGL(epe_final_kernel, dim3(1), dim3(64), 0, s, (const double*)workspace, nblocks, num_thresholds, out);
    NND_LAUNCH_CHECK();
    return NND_OK;
}

}  // extern "C"
