// fp16x2 activation-range calibration (round 4; split_arith.h, DESIGN.md §4 "activation range").
//
// The fp16x2 arithmetic stages every fp32 activation as two fp16 pieces of x * 2^xs.  With a fixed xs the pieces only carry their
// 22 bits for |x| within a fixed window (round 3: [0.06, 16376)); a tensor of small values lost its low piece to fp16's subnormals
// and a large one overflowed.  Now xs is a property of the LAYER, stored in the layer's slot of the packed blob on the device and
// read by the kernels at run time (one scalar load), and it is chosen from data:
//   1. a forward with NND_FLAG_CALIBRATE: every launcher of a kernel that stages fp16x2 activations first runs one of the amax
//      kernels below over exactly what that launch will stage (the conv's sources in their layout; the fused flow branch's internal
//      128-channel map is made by the stand-alone convf1 kernel for the purpose; the IGEV lookup's samples are bounded by the
//      interleaved pyramid they are interpolated from) and atomically maxes the bit pattern of |x| into the slot — over all
//      iterations of a refinement loop, over the batch, over as many calibration forwards as the caller runs;
//   2. nnd_*_calibration_finish: xs = 11 - ceil(log2(max)), i.e. the largest value seen lands in [2^10, 2^11) of fp16's range:
//      a factor 32 of headroom above it before 65504, 13 octaves below it with all 22 bits, and below those an absolute error of
//      2^-36 of the maximum.  oscale is rewritten with it (both exact powers of two).
// Kernels on the product path are unchanged in cost: the scale was a constant multiplication, it is now a multiplication by a
// wave-uniform register.  The calibration kernels run only under the flag.
#include "common.h"
#include "layout.h"
#include "split_arith.h"

#include <cmath>

namespace nnd {

static thread_local int t_calib_depth = 0;
bool calibrating() { return t_calib_depth > 0; }
CalibScope::CalibScope(bool o) : on(o) {
    if (on) ++t_calib_depth;
}
CalibScope::~CalibScope() {
    if (on) --t_calib_depth;
}

namespace {

// bit pattern of |v|: for non-negative floats the unsigned order is the float order, with inf above every finite value and NaN
// above inf — a non-finite activation is therefore never lost by the maximum
__device__ __forceinline__ unsigned abs_bits(float v) { return __float_as_uint(v) & 0x7fffffffu; }

__device__ __forceinline__ void block_max_to_slot(unsigned m, unsigned* slot) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned o = (unsigned)__shfl_xor((int)m, off, 64);
        m = o > m ? o : m;
    }
    __shared__ unsigned wmax[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wmax[wave] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned r = wmax[0];
        for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = wmax[i] > r ? wmax[i] : r;
        if (r != 0) atomicMax(slot, r);
    }
}

// a channel slice of an activation in any of the three layouts (layout.h): only the H x W image pixels are read (the padding of a
// tile-major plane is never written by the kernels and may hold anything)
__global__ void __launch_bounds__(256) amax_act_kernel(const float* __restrict__ src, long bs, int C, Lay lay, int B, int H, int W,
                                                       unsigned* __restrict__ slot) {
    const long HW = (long)H * W, total = (long)B * C * HW;
    unsigned m = 0;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long pix = idx % HW;
        const int c = (int)((idx / HW) % C), b = (int)(idx / (HW * C));
        const unsigned v = abs_bits(src[b * bs + chan_off(lay, c) + pix_off(lay, (int)(pix / W), (int)(pix % W))]);
        m = v > m ? v : m;
    }
    block_max_to_slot(m, slot);
}

__global__ void __launch_bounds__(256) amax_flat_kernel(const float* __restrict__ x, long n, unsigned* __restrict__ slot) {
    unsigned m = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const unsigned v = abs_bits(x[i]);
        m = v > m ? v : m;
    }
    block_max_to_slot(m, slot);
}

constexpr int CALIB_MAX_LAYERS = 64;
struct CalibOffs {
    long off[CALIB_MAX_LAYERS];
};

__global__ void calib_finish_kernel(float* blob, CalibOffs offs, int n, int* status) {
    const int i = threadIdx.x;
    if (i >= n) return;
    float* tail = blob + offs.off[i];
    const unsigned bits = __float_as_uint(tail[SPLIT_TAIL_AMAX]);
    tail[SPLIT_TAIL_AMAX] = 0.f;
    if (bits == 0) {  // the layer staged nothing (not on this forward's path) or only zeros: keep its scale
        if (status) atomicOr(status, 2);
        return;
    }
    int xs;
    if (bits >= 0x7f800000u) {  // inf / NaN reached the layer: the calibration forward itself overflowed upstream or here
        int e_old;
        (void)frexpf(tail[SPLIT_TAIL_XSCALE], &e_old);  // xscale = 2^(e_old - 1)
        xs = (e_old - 1) - 12;
        if (status) atomicOr(status, 1);
    } else {
        int e;
        (void)frexpf(__uint_as_float(bits), &e);  // max = m * 2^e, m in [0.5, 1): max * 2^(T - e) in [2^(T-1), 2^T)
        xs = SPLIT_CALIB_TARGET_EXP - e;
    }
    xs = xs < -60 ? -60 : (xs > 60 ? 60 : xs);  // oscale = 2^-(s + xs) stays a normal fp32 number for any weight scale
    tail[SPLIT_TAIL_XSCALE] = ldexpf(1.f, xs);
    tail[SPLIT_TAIL_OSCALE] = tail[SPLIT_TAIL_WSINV] * ldexpf(1.f, -xs);
}

unsigned* amax_slot(const float* tail) { return reinterpret_cast<unsigned*>(const_cast<float*>(tail)) + SPLIT_TAIL_AMAX; }

}  // namespace

int calib_amax_act(const Act& a, const Lay& lay, int B, int H, int W, const float* tail, hipStream_t s) {
    if (a.C <= 0) return NND_OK;
    NND_REQUIRE(a.ptr && tail && B > 0 && H > 0 && W > 0, "calibration: null activation / slot");
    const long total = (long)B * a.C * H * W;
    const unsigned grid = (unsigned)std::min<long>(cdiv64(total, 256), 4096);
    hipLaunchKernelGGL(amax_act_kernel, dim3(grid), dim3(256), 0, s, (const float*)a.ptr, (long)a.bstride, a.C, lay, B, H, W, amax_slot(tail));
    NND_LAUNCH_CHECK();
    return NND_OK;
}

int calib_amax_flat(const float* x, int64_t n, const float* tail, hipStream_t s) {
    if (n <= 0) return NND_OK;
    NND_REQUIRE(x && tail, "calibration: null tensor / slot");
    const unsigned grid = (unsigned)std::min<long>(cdiv64(n, 256), 8192);
    hipLaunchKernelGGL(amax_flat_kernel, dim3(grid), dim3(256), 0, s, x, (long)n, amax_slot(tail));
    NND_LAUNCH_CHECK();
    return NND_OK;
}

int calib_finish(float* blob, const int64_t* tail_offs, int n, int32_t* status_dev, hipStream_t s) {
    NND_REQUIRE(blob && (tail_offs || n == 0) && n >= 0, "calibration_finish: null pointer");
    CalibOffs o;
    int m = 0;
    for (int i = 0; i < n; ++i) {  // a layer may appear twice in a plan (conv_gru shares the GRU layers of both passes)
        bool dup = false;
        for (int j = 0; j < m; ++j) dup = dup || o.off[j] == (long)tail_offs[i];
        if (dup) continue;
        NND_REQUIRE(m < CALIB_MAX_LAYERS, "calibration_finish: more than %d fp16x2 layers in one blob", CALIB_MAX_LAYERS);
        o.off[m++] = (long)tail_offs[i];
    }
    if (m == 0) return NND_OK;
    hipLaunchKernelGGL(calib_finish_kernel, dim3(1), dim3(CALIB_MAX_LAYERS), 0, s, blob, o, m, (int*)status_dev);
    NND_LAUNCH_CHECK();
    return NND_OK;
}

}  // namespace nnd
