// Shared by the convolution kernels (conv_mfma.hip: exact fp32 MFMA; conv_split.hip: split 16-bit MFMA): the kernel
// argument block, the branch-free ~1-ulp sigmoid / tanh, and the epilogue.  Both kernels produce the standard 32x32 MFMA
// accumulator layout — lane = pixel (l & 31) and half h2 = l >> 5, register `reg` = output channel
// cb*32 + (reg & 3) + 8*(reg >> 2) + 4*h2 — so the fused epilogues (ReLU, x0.25, folded norm + residual, GRU z / r*h,
// GRU blend, per-pixel bias map) are written once.
#pragma once
#include "common.h"
#include "layout.h"

namespace nnd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvArgs {
    const float* src0;
    const float* src1;
    long bs0, bs1;
    int c0, c1;
    const float* wpk;
    const float* bias;
    float* out0;
    float* out1;
    long obs0, obs1;
    const float* aux0;
    const float* aux1;
    long abs0, abs1;
    const float* bmap;  // optional per-pixel bias (B, Cout, H, W) added instead of bias[co]
    long bmbs;
    Lay ls, ld;         // layout of the sources / of every destination-side tensor (out*, aux*, bmap)
    int H, W, Cout, nchunks, epi, hidden;  // H, W: OUTPUT size
    int Hin, Win;                          // input size (== H, W for stride 1)
    int tiles_x, wco, ks, npos, ngroups;
    int flags;                             // EPI_AFFINE: bit 0 ReLU after the affine, bit 1 ReLU after the residual add,
                                           // bit 2 LeakyReLU (negative slope = scale) after the affine
    const float* cscale;                   // EPI_AFFINE: per-channel scale (folded norm), shift comes in through `bias`
    float scale;
    int dbg_stamp;                         // -DNND_DBG_STAMPS builds only: this launch records its phase stamps
};

// exp(x) to ~1 ulp without libm's special-case branches: x*log2(e) is split into the rounded product t and its
// exact residual r (fma), so exp(x) = exp2(t) * 2^r ~= exp2(t) * (1 + r*ln2); v_exp_f32 itself is ~1 ulp.
__device__ __forceinline__ float exp_acc(float x) {
    const float L2E = 1.44269504088896341f;
    x = fminf(fmaxf(x, -87.0f), 88.0f);
    const float t = x * L2E;
    const float r = fmaf(x, 1.92596299e-8f, fmaf(x, L2E, -t));  // + x * (log2e - (float)log2e); explicit fma: no context-dependent contraction
    const float e = __builtin_amdgcn_exp2f(t);
    return fmaf(e, r * 0.693147180559945f, e);
}
// 1/d with one Newton step on v_rcp_f32 (<= 1 ulp)
__device__ __forceinline__ float rcp_acc(float d) {
    float y = __builtin_amdgcn_rcpf(d);
    return fmaf(fmaf(-d, y, 1.0f), y, y);
}
__device__ __forceinline__ float sigmoidf_(float v) { return rcp_acc(1.0f + exp_acc(-v)); }
// tanh(x) = sign(x) * (1 - 2/(exp(2|x|) + 1)); for |x| < 0.04 the odd series avoids the cancellation
__device__ __forceinline__ float tanhf_(float v) {
    const float ax = fabsf(v);
    const float big = fmaf(-2.0f, rcp_acc(exp_acc(2.0f * ax) + 1.0f), 1.0f);
    const float x2 = ax * ax;
    const float small = ax * fmaf(x2, fmaf(x2, 0.133333333f, -0.333333333f), 1.0f);
    return copysignf(ax < 0.04f ? small : big, v);
}

// Epilogue of one wave: acc[pp] = the 32 (channel) x 32 (pixel) tile of sub-tile pp, this lane's pixel of it at (ys[pp], xs[pp]);
// the wave stores registers [reg0, reg0 + nreg) (its share after an intra-workgroup split-K exchange; 0, 16 otherwise).
// All loads (bias, h, z) are issued before any store: out0 may alias aux0 (GRU blend in place), which would otherwise
// serialise load / store.
template <int P>
__device__ __forceinline__ void conv_epilogue_planar(const ConvArgs& a, const f32x16 (&acc)[P], const int cb, const int b, const int h2,
                                                     const int reg0, const int nreg, const int (&ys)[P], const int (&xs)[P]) {
    const int H = a.H, W = a.W;
    const long DP = a.ld.plane;
    const int epi = a.epi;
    float bias_r[16];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int co = cb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h2;
        bias_r[reg] = (!a.bmap && reg >= reg0 && reg < reg0 + nreg && co < a.Cout) ? a.bias[co] : 0.f;
    }
#pragma unroll
    for (int pp = 0; pp < P; ++pp) {
        const int y = ys[pp], x = xs[pp];
        const bool pix_ok = (y < H && x < W);
        const long pix = pix_ok ? pix_off(a.ld, y, x) : 0;
        float h_r[16], z_r[16];
        if (a.bmap) {  // precomputed context term of the GRU convs (constant over the iterations of a pair)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int co = cb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h2;
                const bool ok = pix_ok && reg >= reg0 && reg < reg0 + nreg && co < a.Cout;
                bias_r[reg] = ok ? a.bmap[b * a.bmbs + co * DP + pix] : 0.f;
            }
        }
        if (epi == EPI_AFFINE) {  // z_r <- per-channel scale, h_r <- residual (dst layout), bias_r = shift
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int co = cb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h2;
                const bool ok = pix_ok && reg >= reg0 && reg < reg0 + nreg && co < a.Cout;
                z_r[reg] = ok ? a.cscale[co] : 0.f;
                h_r[reg] = (ok && a.aux0) ? a.aux0[b * a.abs0 + co * DP + pix] : 0.f;
            }
        }
        if (epi == EPI_GRU_ZR || epi == EPI_GRU_Q) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int co = cb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h2;
                const bool ok = pix_ok && reg >= reg0 && reg < reg0 + nreg && co < a.Cout;
                h_r[reg] = 0.f;
                z_r[reg] = 0.f;
                if (epi == EPI_GRU_Q) {
                    if (ok) {
                        h_r[reg] = a.aux0[b * a.abs0 + co * DP + pix];
                        z_r[reg] = a.aux1[b * a.abs1 + co * DP + pix];
                    }
                } else if (ok && co >= a.hidden) {
                    h_r[reg] = a.aux0[b * a.abs0 + (co - a.hidden) * DP + pix];
                }
            }
        }
        if (!pix_ok) continue;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            if (reg < reg0 || reg >= reg0 + nreg) continue;
            const int co = cb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h2;
            if (co >= a.Cout) continue;
            const float v = acc[pp][reg] + bias_r[reg];
            if (epi == EPI_RELU) {
                a.out0[b * a.obs0 + co * DP + pix] = fmaxf(v, 0.f);
            } else if (epi == EPI_LINEAR) {
                a.out0[b * a.obs0 + co * DP + pix] = v;
            } else if (epi == EPI_SCALE) {
                a.out0[b * a.obs0 + co * DP + pix] = a.scale * v;
            } else if (epi == EPI_SIGMOID_RANGE) {
                a.out0[b * a.obs0 + co * DP + pix] = a.scale * (sigmoidf_(v) - 0.5f) * 2.0f;
            } else if (epi == EPI_AFFINE) {  // folded norm: y = acc*scale + shift; optional ReLU, residual add, ReLU
                float y2 = fmaf(acc[pp][reg], z_r[reg], bias_r[reg]);
                if (a.flags & 4) y2 = y2 > 0.f ? y2 : a.scale * y2;  // LeakyReLU (slope in `scale`), Conv3d path
                if (a.flags & 1) y2 = fmaxf(y2, 0.f);
                if (a.aux0) y2 = h_r[reg] + y2;
                if (a.flags & 2) y2 = fmaxf(y2, 0.f);
                a.out0[b * a.obs0 + co * DP + pix] = y2;
            } else if (epi == EPI_GRU_ZR) {
                const float sg = sigmoidf_(v);
                if (co < a.hidden) a.out0[b * a.obs0 + co * DP + pix] = sg;
                else a.out1[b * a.obs1 + (co - a.hidden) * DP + pix] = sg * h_r[reg];
            } else {  // EPI_GRU_Q
                const float q = tanhf_(v);
                const float hn = fmaf(z_r[reg], q, (1.0f - z_r[reg]) * h_r[reg]);  // explicit: both layouts' epilogues round alike
                a.out0[b * a.obs0 + co * DP + pix] = hn;
                if (a.out1) a.out1[b * a.obs1 + co * DP + pix] = hn;
            }
        }
    }
}

// The same for destination-side tensors in the 4-channel-interleaved tile-major layout (layout.h): registers 4q..4q+3 of a
// lane are 4 consecutive channels of its pixel = one 16-B access per tensor (bmap / h / z loads, stores) instead of four
// 4-B ones.  reg0 and nreg are multiples of 4 (ks = 1, 2, 4).  A last group that Cout cuts (Cout = 127 of the motion
// encoder: its 4th channel is the flow, written by another kernel) is stored channel by channel.
template <int P>
__device__ __forceinline__ void conv_epilogue_c4(const ConvArgs& a, const f32x16 (&acc)[P], const int cb, const int b, const int h2,
                                                 const int reg0, const int nreg, const int (&ys)[P], const int (&xs)[P]) {
    const int H = a.H, W = a.W;
    const long DP = a.ld.plane;
    const int epi = a.epi;
    auto ld4 = [](const float* p) { return *reinterpret_cast<const float4*>(p); };
    auto get = [](const float4& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); };
    float4 bias4[4], scale4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int co0 = cb * 32 + 8 * q + 4 * h2;
        const bool on = 4 * q >= reg0 && 4 * q < reg0 + nreg;
        float bv[4], sv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bv[i] = (!a.bmap && on && co0 + i < a.Cout) ? a.bias[co0 + i] : 0.f;
            sv[i] = (epi == EPI_AFFINE && on && co0 + i < a.Cout) ? a.cscale[co0 + i] : 0.f;
        }
        bias4[q] = make_float4(bv[0], bv[1], bv[2], bv[3]);
        scale4[q] = make_float4(sv[0], sv[1], sv[2], sv[3]);
    }
#pragma unroll
    for (int pp = 0; pp < P; ++pp) {
        const int y = ys[pp], x = xs[pp];
        const bool pix_ok = (y < H && x < W);
        const long pix = pix_ok ? pix_off(a.ld, y, x) : 0;  // already x4
        float4 bm[4], hv[4], zv[4];
        // ---- every load of this sub-tile first
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int co0 = cb * 32 + 8 * q + 4 * h2;
            const bool ok = pix_ok && 4 * q >= reg0 && 4 * q < reg0 + nreg && co0 + 3 < a.Cout;  // whole groups only
            const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
            bm[q] = bias4[q];
            hv[q] = zero;
            zv[q] = zero;
            if (a.bmap && ok) bm[q] = ld4(a.bmap + b * a.bmbs + (long)co0 * DP + pix);
            if (epi == EPI_AFFINE) {
                if (a.aux0 && ok) hv[q] = ld4(a.aux0 + b * a.abs0 + (long)co0 * DP + pix);
            } else if (epi == EPI_GRU_Q) {
                if (ok) {
                    hv[q] = ld4(a.aux0 + b * a.abs0 + (long)co0 * DP + pix);
                    zv[q] = ld4(a.aux1 + b * a.abs1 + (long)co0 * DP + pix);
                }
            } else if (epi == EPI_GRU_ZR) {
                if (ok && co0 >= a.hidden) hv[q] = ld4(a.aux0 + b * a.abs0 + (long)(co0 - a.hidden) * DP + pix);
            }
        }
        if (!pix_ok) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (4 * q < reg0 || 4 * q >= reg0 + nreg) continue;
            const int co0 = cb * 32 + 8 * q + 4 * h2;
            if (co0 >= a.Cout) continue;
            const bool full = co0 + 3 < a.Cout;
            float r0[4], r1[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float ac = acc[pp][4 * q + i];
                const float v = ac + get(bm[q], i);
                float o0 = v, o1 = 0.f;
                if (epi == EPI_RELU) o0 = fmaxf(v, 0.f);
                else if (epi == EPI_SCALE) o0 = a.scale * v;
                else if (epi == EPI_SIGMOID_RANGE) o0 = a.scale * (sigmoidf_(v) - 0.5f) * 2.0f;
                else if (epi == EPI_AFFINE) {
                    float y2 = fmaf(ac, get(scale4[q], i), get(bm[q], i));
                    if (a.flags & 4) y2 = y2 > 0.f ? y2 : a.scale * y2;
                    if (a.flags & 1) y2 = fmaxf(y2, 0.f);
                    if (a.aux0) y2 = get(hv[q], i) + y2;
                    if (a.flags & 2) y2 = fmaxf(y2, 0.f);
                    o0 = y2;
                } else if (epi == EPI_GRU_ZR) {
                    const float sg = sigmoidf_(v);
                    o0 = sg;
                    o1 = sg * get(hv[q], i);
                } else if (epi == EPI_GRU_Q) {
                    const float qq = tanhf_(v);
                    o0 = fmaf(get(zv[q], i), qq, (1.0f - get(zv[q], i)) * get(hv[q], i));
                    o1 = o0;
                }
                r0[i] = o0;
                r1[i] = o1;
            }
            // EPI_GRU_ZR: channel groups below `hidden` are z (out0), the others r*h (out1, channel - hidden)
            const bool to1 = epi == EPI_GRU_ZR && co0 >= a.hidden;
            float* d0 = to1 ? a.out1 + b * a.obs1 + (long)(co0 - a.hidden) * DP + pix : a.out0 + b * a.obs0 + (long)co0 * DP + pix;
            float rr[4];  // element-wise select: a pointer select between the two arrays would put both in scratch memory
#pragma unroll
            for (int i = 0; i < 4; ++i) rr[i] = to1 ? r1[i] : r0[i];
            if (full) {
                *reinterpret_cast<float4*>(d0) = make_float4(rr[0], rr[1], rr[2], rr[3]);
                if (epi == EPI_GRU_Q && a.out1) *reinterpret_cast<float4*>(a.out1 + b * a.obs1 + (long)co0 * DP + pix) = make_float4(r1[0], r1[1], r1[2], r1[3]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (co0 + i < a.Cout) {
                        d0[i] = rr[i];
                        if (epi == EPI_GRU_Q && a.out1) a.out1[b * a.obs1 + (long)co0 * DP + pix + i] = r1[i];
                    }
            }
        }
    }
}

// The c4 epilogue in two halves over a COMPACT register set, for a kernel with an intra-workgroup split-K exchange
// (conv_split.hip): the wave stores NG = 4/ks groups of 4 registers per sub-tile, groups q0 .. q0+NG-1 (q0 = kj*NG is a
// run-time value, used in addresses only).  `load` is issued before the exchange — the fragment registers are dead — and its
// 16-B loads (per-pixel bias map, h, z) fly while the partial sums cross LDS; `store` takes the summed accumulators as
// float4 cacc[pp][j].  Same per-element arithmetic, in the same order, as conv_epilogue_c4: bit-identical results.
// All loads precede all stores of the wave (out0 may alias aux0).
template <int P, int NG>
struct EpiOpsC4 {
    float4 bm[P][NG], hv[P][NG], zv[P][NG];  // bias (per channel or per pixel), h / residual, z
    float4 scale4[NG];
    long pix[P];  // already x4
    bool pix_ok[P];
};

template <int P, int NG>
__device__ __forceinline__ void epi_c4_load(const ConvArgs& a, const int cb, const int b, const int h2, const int q0,
                                            const int (&ys)[P], const int (&xs)[P], EpiOpsC4<P, NG>& e) {
    const long DP = a.ld.plane;
    const int epi = a.epi;
    auto ld4 = [](const float* p) { return *reinterpret_cast<const float4*>(p); };
    float4 bias4[NG];
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        const int co0 = cb * 32 + 8 * (q0 + j) + 4 * h2;
        float bv[4], sv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bv[i] = (!a.bmap && co0 + i < a.Cout) ? a.bias[co0 + i] : 0.f;
            sv[i] = (epi == EPI_AFFINE && co0 + i < a.Cout) ? a.cscale[co0 + i] : 0.f;
        }
        bias4[j] = make_float4(bv[0], bv[1], bv[2], bv[3]);
        e.scale4[j] = make_float4(sv[0], sv[1], sv[2], sv[3]);
    }
#pragma unroll
    for (int pp = 0; pp < P; ++pp) {
        const int y = ys[pp], x = xs[pp];
        const bool pix_ok = (y < a.H && x < a.W);
        const long pix = pix_ok ? pix_off(a.ld, y, x) : 0;
        e.pix_ok[pp] = pix_ok;
        e.pix[pp] = pix;
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int co0 = cb * 32 + 8 * (q0 + j) + 4 * h2;
            const bool ok = pix_ok && co0 + 3 < a.Cout;  // whole groups only
            const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
            e.bm[pp][j] = bias4[j];
            e.hv[pp][j] = zero;
            e.zv[pp][j] = zero;
            if (a.bmap && ok) e.bm[pp][j] = ld4(a.bmap + b * a.bmbs + (long)co0 * DP + pix);
            if (epi == EPI_AFFINE) {
                if (a.aux0 && ok) e.hv[pp][j] = ld4(a.aux0 + b * a.abs0 + (long)co0 * DP + pix);
            } else if (epi == EPI_GRU_Q) {
                if (ok) {
                    e.hv[pp][j] = ld4(a.aux0 + b * a.abs0 + (long)co0 * DP + pix);
                    e.zv[pp][j] = ld4(a.aux1 + b * a.abs1 + (long)co0 * DP + pix);
                }
            } else if (epi == EPI_GRU_ZR) {
                if (ok && co0 >= a.hidden) e.hv[pp][j] = ld4(a.aux0 + b * a.abs0 + (long)(co0 - a.hidden) * DP + pix);
            }
        }
    }
}

template <int P, int NG>
__device__ __forceinline__ void epi_c4_store(const ConvArgs& a, const float4 (&cacc)[P][NG], const int cb, const int b, const int h2,
                                             const int q0, const EpiOpsC4<P, NG>& e) {
    const long DP = a.ld.plane;
    const int epi = a.epi;
    auto get = [](const float4& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); };
#pragma unroll
    for (int pp = 0; pp < P; ++pp) {
        if (!e.pix_ok[pp]) continue;
        const long pix = e.pix[pp];
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int co0 = cb * 32 + 8 * (q0 + j) + 4 * h2;
            if (co0 >= a.Cout) continue;
            const bool full = co0 + 3 < a.Cout;
            float r0[4], r1[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float ac = get(cacc[pp][j], i);
                const float v = ac + get(e.bm[pp][j], i);
                float o0 = v, o1 = 0.f;
                if (epi == EPI_RELU) o0 = fmaxf(v, 0.f);
                else if (epi == EPI_SCALE) o0 = a.scale * v;
                else if (epi == EPI_SIGMOID_RANGE) o0 = a.scale * (sigmoidf_(v) - 0.5f) * 2.0f;
                else if (epi == EPI_AFFINE) {
                    float y2 = fmaf(ac, get(e.scale4[j], i), get(e.bm[pp][j], i));
                    if (a.flags & 4) y2 = y2 > 0.f ? y2 : a.scale * y2;
                    if (a.flags & 1) y2 = fmaxf(y2, 0.f);
                    if (a.aux0) y2 = get(e.hv[pp][j], i) + y2;
                    if (a.flags & 2) y2 = fmaxf(y2, 0.f);
                    o0 = y2;
                } else if (epi == EPI_GRU_ZR) {
                    const float sg = sigmoidf_(v);
                    o0 = sg;
                    o1 = sg * get(e.hv[pp][j], i);
                } else if (epi == EPI_GRU_Q) {
                    const float qq = tanhf_(v);
                    o0 = fmaf(get(e.zv[pp][j], i), qq, (1.0f - get(e.zv[pp][j], i)) * get(e.hv[pp][j], i));
                    o1 = o0;
                }
                r0[i] = o0;
                r1[i] = o1;
            }
            const bool to1 = epi == EPI_GRU_ZR && co0 >= a.hidden;
            float* d0 = to1 ? a.out1 + b * a.obs1 + (long)(co0 - a.hidden) * DP + pix : a.out0 + b * a.obs0 + (long)co0 * DP + pix;
            float rr[4];  // element-wise select: a pointer select between the two arrays would put both in scratch memory
#pragma unroll
            for (int i = 0; i < 4; ++i) rr[i] = to1 ? r1[i] : r0[i];
            if (full) {
                *reinterpret_cast<float4*>(d0) = make_float4(rr[0], rr[1], rr[2], rr[3]);
                if (epi == EPI_GRU_Q && a.out1) *reinterpret_cast<float4*>(a.out1 + b * a.obs1 + (long)co0 * DP + pix) = make_float4(r1[0], r1[1], r1[2], r1[3]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (co0 + i < a.Cout) {
                        d0[i] = rr[i];
                        if (epi == EPI_GRU_Q && a.out1) a.out1[b * a.obs1 + (long)co0 * DP + pix + i] = r1[i];
                    }
            }
        }
    }
}

// Epilogue of one wave: acc[pp] = the 32 (channel) x 32 (pixel) tile of sub-tile pp, this lane's pixel of it at (ys[pp], xs[pp]);
// the wave stores registers [reg0, reg0 + nreg) (its share after an intra-workgroup split-K exchange; 0, 16 otherwise).
template <int P>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, const f32x16 (&acc)[P], const int cb, const int b, const int h2,
                                              const int reg0, const int nreg, const int (&ys)[P], const int (&xs)[P]) {
    if (a.ld.ci == 4) conv_epilogue_c4<P>(a, acc, cb, b, h2, reg0, nreg, ys, xs);
    else conv_epilogue_planar<P>(a, acc, cb, b, h2, reg0, nreg, ys, xs);
}

}  // namespace nnd
