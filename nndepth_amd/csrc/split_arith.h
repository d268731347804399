// The two split arithmetics of the 16-bit MFMA convolutions (conv_split.hip, mask_upsample.hip), selected per layer by
// ConvLayer::arith = the number of pieces NS an fp32 operand is carried as:
//
//   NS = 3  "bf16x3"   x = x0 + x1 + x2 in bf16 (3 x 8 = 24 significand bits), the 6 products x_i*w_j with i + j <= 2 on
//                      v_mfma_f32_32x32x16_bf16; what is dropped is <= 2^-24 |x||w|                              (round 2)
//   NS = 2  "fp16x2"   x = x0 + x1 in fp16 (2 x 11 = 22 significand bits), the 3 products x0*w0, x0*w1, x1*w0 on
//                      v_mfma_f32_32x32x16_f16: HALF the matrix work and 2/3 of the operand bytes of bf16x3.      (round 3)
//                      fp16 has 5 exponent bits, so both operands are range-scaled by exact powers of two that the kernel undoes
//                      after the K loop (acc *= oscale): weights x 2^s per layer at pack time with max|w| * 2^s in [2^13, 2^14),
//                      activations x 2^xs PER LAYER while they are staged (round 4).  xs lives in the layer's slot of the packed
//                      blob (SPLIT_TAIL_* below): 2 until the layer has been calibrated, then chosen from the largest
//                      |activation| the layer staged during a calibration forward (calib.hip) so that it lands in [2^10, 2^11).
//                      An element keeps all 22 bits while |x| * 2^xs >= 2^-2 — 13 octaves below the calibrated maximum —; below
//                      that its ABSOLUTE error is <= 2^-25 in scaled units = 2^-36 of the calibrated maximum (negligible against
//                      the fp32 accumulation rounding of the terms that dominate the dot product).  Valid while
//                      |x| < 65504 / 2^xs, i.e. up to 32 x the calibrated maximum (uncalibrated: 16376); beyond that the
//                      conversion gives inf and the output is NaN/inf (visible, never silently wrong).  Per-op error vs
//                      float64: representation 5.6e-8 rms at |y| <= 4 on the convc2 shape, 20-30x below the fp32 accumulation
//                      rounding both arithmetics share (scripts/study/).
// Every product of two pieces is an exact fp32 number (8 x 8 or 11 x 11 significand bits) and the MFMA accumulates in fp32.
#pragma once
#include <hip/hip_runtime.h>

namespace nnd {

typedef float f32x16s __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8s __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8s __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4s __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4s __attribute__((ext_vector_type(4)));

constexpr int SPLIT_F16_XSHIFT = 2;  // activation shift of a layer that has not been calibrated: x 4 before the fp16 split
// Tail of a packed fp16x2 layer: 4 floats behind its bias vector (ConvLayer::b_floats; slab3d: behind scale | shift)
constexpr int SPLIT_TAIL_OSCALE = 0;  // 2^-(s + xs): what the kernels multiply their accumulators by after the K loop
constexpr int SPLIT_TAIL_XSCALE = 1;  // 2^xs: the activation scale applied while staging
constexpr int SPLIT_TAIL_AMAX = 2;    // calibration accumulator: bit pattern of the largest |activation| staged (atomicMax on uint)
constexpr int SPLIT_TAIL_WSINV = 3;   // 2^-s: the weight part of oscale (calibration recomputes oscale = wsinv / xscale)
constexpr int SPLIT_CALIB_TARGET_EXP = 11;  // calibration puts the largest |activation| * 2^xs in [2^10, 2^11): x 32 of headroom
__host__ __device__ constexpr int split_nprod(int NS) { return NS * (NS + 1) / 2; }

// N (4 or 8) fp32 -> NS pieces of N 16-bit values each: round-to-nearest of the running residual (the subtractions are exact)
// (xscale: the layer's activation scale, a power of two — NS == 2 only)
template <int NS, int N, typename OutT>
__device__ __forceinline__ void split_pieces_n(const float (&x)[N], OutT (&out)[NS], const float xscale) {
    static_assert(NS == 2 || NS == 3, "2 fp16 pieces or 3 bf16 pieces");
    static_assert(sizeof(OutT) == 2 * N, "one 16-bit value per input");
    float res[N];
#pragma unroll
    for (int j = 0; j < N; ++j) res[j] = NS == 2 ? x[j] * xscale : x[j];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if constexpr (NS == 3) {
            typedef __bf16 vt __attribute__((ext_vector_type(N)));
            vt v;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                v[j] = (__bf16)res[j];
                res[j] -= (float)v[j];
            }
            out[s] = __builtin_bit_cast(OutT, v);
        } else {
            typedef _Float16 vt __attribute__((ext_vector_type(N)));
            vt v;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                v[j] = (_Float16)res[j];
                res[j] -= (float)v[j];
            }
            out[s] = __builtin_bit_cast(OutT, v);
        }
    }
}
template <int NS>
__device__ __forceinline__ void split_pieces(const float (&x)[8], uint4 (&out)[NS], const float xscale) {
    split_pieces_n<NS, 8, uint4>(x, out, xscale);
}

// acc += A(32 x 16) * B(16 x 32) for one pair of pieces
template <int NS>
__device__ __forceinline__ f32x16s split_mfma(const uint4& a, const uint4& b, const f32x16s& c) {
    if constexpr (NS == 3) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8s, a), __builtin_bit_cast(bf16x8s, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8s, a), __builtin_bit_cast(f16x8s, b), c, 0, 0, 0);
}

// all products of one K step: x_i * w_j with i + j descending (small products first, so they are not absorbed one by one into
// a large partial sum any earlier than necessary); aw = weight pieces, bx = activation pieces
template <int NS>
__device__ __forceinline__ void split_mfma_step(const uint4 (&aw)[NS], const uint4 (&bx)[NS], f32x16s& acc) {
#pragma unroll
    for (int sum = NS - 1; sum >= 0; --sum)
#pragma unroll
        for (int i = 0; i <= sum; ++i) acc = split_mfma<NS>(aw[sum - i], bx[i], acc);
}

}  // namespace nnd
