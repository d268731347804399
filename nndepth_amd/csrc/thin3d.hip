// Thin 3x3x3 Conv3d layers of IGEV's cost-volume regulariser (ConvBn3D / Upsampler3D with 8 or 16 output channels,
// nndepth/models/igev_stereo/cost_volume.py:101-130, 176-190) as a direct fp32 VALU convolution on depth-major volumes.
//
// Why not the MFMA path (conv3d.hip): with Cout = 8 / 16 a 32-row MFMA tile is filled only by grouping J = 4 / 2 output
// slices, which makes half / a quarter of K structural zeros, and each 32-pixel tile then carries the full prologue /
// epilogue / staging overhead of the generic kernel for very little K: conv1_up + final_conv ran at 26 / 19 TFLOP/s and took
// 3.4 ms of the regulariser's 8.3 ms per 544x960 sample (profiles/r02_igev_regulariser_layers_before.txt).  Here:
//   thread  = 4 consecutive x outputs x 1 y x 1 depth slice x all COUT channels (4*COUT accumulators, an exact fmaf chain),
//   weights = [ci][kd][kh][kw][co]: the chunk's 4 x 27 x COUT values are staged in LDS next to the slab and read back as
//             broadcast ds_read_b128 (SGPR operands through the scalar cache were measured first: scalar loads return out of
//             order, so every wait on them is lgkmcnt(0) and drains the LDS reads of the inputs too — 13 full drains per
//             channel, 55 TFLOP/s; with both operands in LDS the waits are counted),
//   input   = a (TZ+2) x 4-channel x (TY+2) x (TX+2) slab per channel chunk in LDS (32 KB, 4 workgroups per CU); a thread reads
//             its 6 consecutive x values per (channel, kd, kh) as ds_read_b128 + ds_read_b64 and reuses them for 3 kw x 4 outputs,
// 8 FMAs per LDS word, no structural zeros.  Two sources = the virtual channel concat of proj_2 / proj_3.
// Stride 2 (conv1.0 .. conv3.0): the same with the slab 2x larger per axis and outputs 2 apart.
#include "common.h"

namespace nnd {

struct Thin3dArgs {
    const float* x0;  // (N, D+2, C0, H, W) depth-major, zero end slices
    const float* x1;  // (N, D+2, C1, H, W) or null
    const float* w;   // [C0+C1][3][3][3][COUT]
    const float* scale;
    const float* shift;
    float* y;         // (N, Do+2, COUT, Ho, Wo); the end slices are zeroed by the host wrapper
    int C0, C1, D, H, W, Do, Ho, Wo;
    float slope;
};

constexpr int T3_TX = 16, T3_TY = 8, T3_TZ = 8;  // outputs per workgroup
__host__ __device__ constexpr int t3_cc(int STR) { return STR == 1 ? 4 : 1; }  // channels per LDS chunk (slab <= 48 KB)

template <int COUT, int STR>
__global__ void __launch_bounds__(256) thin_conv3d_kernel(Thin3dArgs a) {
    constexpr int SX = (T3_TX - 1) * STR + 3, SY = (T3_TY - 1) * STR + 3, SZ = (T3_TZ - 1) * STR + 3;
    constexpr int RS = (SX + 3) / 4 * 4 + 4;  // slab row stride in floats: 16-B aligned rows, +4 so that rows shift banks
    constexpr int PLANE = SY * RS, CHS = SZ * PLANE, T3_CC = t3_cc(STR);
    extern __shared__ __attribute__((aligned(16))) float slab[];  // [T3_CC][SZ][SY][RS] inputs | [T3_CC][27][COUT] weights
    float* wl = slab + T3_CC * CHS;
    const int tid = threadIdx.x;
    const int tx = tid & 3, ty = (tid >> 2) & 7, tz = tid >> 5;
    const int tiles_x = (a.Wo + T3_TX - 1) / T3_TX;
    const int ox0 = (blockIdx.x % tiles_x) * T3_TX, oy0 = (blockIdx.x / tiles_x) * T3_TY, oz0 = blockIdx.y * T3_TZ;
    const int n = blockIdx.z;
    const long HW = (long)a.H * a.W;
    const int ix0 = ox0 * STR - 1, iy0 = oy0 * STR - 1, iz0 = oz0 * STR;  // padded slice index of input slice (oz*STR - 1) is oz*STR

    float acc[4][COUT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < COUT; ++c) acc[i][c] = 0.f;

    const int Ct = a.C0 + a.C1;
    for (int cbase = 0; cbase < Ct; cbase += T3_CC) {
        __syncthreads();  // the previous chunk's reads are done
        // ---- stage T3_CC channels of the slab (zero outside the volume; the depth padding is in the tensor itself);
        // consecutive threads take consecutive x of a slab row; loads in batches of 8 before their LDS stores
        constexpr int NEL = T3_CC * SZ * SY * SX, NB = 4;
        for (int e0 = tid; e0 < NEL; e0 += 256 * NB) {
            float v[NB];
            int lo[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const int e = e0 + 256 * k;
                const int sx = e % SX, r1 = e / SX;
                const int sy = r1 % SY, r2 = r1 / SY;
                const int sz = r2 % SZ, cc = r2 / SZ;
                const int gx = ix0 + sx, gy = iy0 + sy, gz = iz0 + sz;  // gz: padded slice index 0..D+1
                const int ci = cbase + cc;
                const bool ok = e < NEL && ci < Ct && gx >= 0 && gx < a.W && gy >= 0 && gy < a.H && gz <= a.D + 1;
                const float* src = a.x0;
                long off = 0;
                if (ok) {
                    if (ci < a.C0) {
                        off = (((long)n * (a.D + 2) + gz) * a.C0 + ci) * HW + (long)gy * a.W + gx;
                    } else {
                        src = a.x1;
                        off = (((long)n * (a.D + 2) + gz) * a.C1 + (ci - a.C0)) * HW + (long)gy * a.W + gx;
                    }
                }
                const float t = src[off];  // unconditional (element 0 when masked): the loads of a batch are all in flight
                v[k] = ok ? t : 0.f;
                lo[k] = e < NEL ? cc * CHS + sz * PLANE + sy * RS + sx : -1;
            }
#pragma unroll
            for (int k = 0; k < NB; ++k)
                if (lo[k] >= 0) slab[lo[k]] = v[k];
        }
        for (int e = tid; e < T3_CC * 27 * COUT; e += 256) {
            const int ci = cbase + e / (27 * COUT);
            wl[e] = ci < Ct ? a.w[(long)cbase * 27 * COUT + e] : 0.f;
        }
        __syncthreads();
#pragma unroll 1
        for (int cc = 0; cc < T3_CC; ++cc) {
            if (cbase + cc >= Ct) break;
            constexpr int NV = 3 * STR + 3;            // input x values feeding 4 outputs: 6 (stride 1) or 9 (stride 2)
            constexpr int NVP = (NV + 3) / 4 * 4;
            const float* wc = wl + cc * 27 * COUT;
            auto depth_tap = [&](int kd) {
                float xv[3][NVP];  // the 3 input rows of this depth tap
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const float* row = slab + cc * CHS + (tz * STR + kd) * PLANE + (ty * STR + kh) * RS + tx * 4 * STR;
#pragma unroll
                    for (int q = 0; q < NVP / 4; ++q) {
                        const float4 t = *reinterpret_cast<const float4*>(row + 4 * q);
                        xv[kh][4 * q] = t.x; xv[kh][4 * q + 1] = t.y; xv[kh][4 * q + 2] = t.z; xv[kh][4 * q + 3] = t.w;
                    }
                }
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        float wv[COUT];
                        const float4* wt = reinterpret_cast<const float4*>(wc + ((kd * 3 + kh) * 3 + kw) * COUT);
#pragma unroll
                        for (int q = 0; q < COUT / 4; ++q) {
                            const float4 t = wt[q];  // same address in every lane: LDS broadcast
                            wv[4 * q] = t.x; wv[4 * q + 1] = t.y; wv[4 * q + 2] = t.z; wv[4 * q + 3] = t.w;
                        }
#pragma unroll
                        for (int c = 0; c < COUT; ++c)
#pragma unroll
                            for (int i = 0; i < 4; ++i) acc[i][c] = fmaf(wv[c], xv[kh][i * STR + kw], acc[i][c]);
                    }
            };
            if constexpr (STR == 1 || COUT <= 8) {
#pragma unroll
                for (int kd = 0; kd < 3; ++kd) depth_tap(kd);
            } else {  // 16 channels at stride 2: unrolled, the hoisted LDS reads of three depth taps need 256 VGPRs (1 wave / SIMD)
#pragma unroll 1
                for (int kd = 0; kd < 3; ++kd) depth_tap(kd);
            }
        }
    }
    // ---- epilogue: folded BatchNorm affine + LeakyReLU, 4 consecutive x per store
    const int oz = oz0 + tz, oy = oy0 + ty, ox = ox0 + tx * 4;
    if (oz >= a.Do || oy >= a.Ho) return;
    const long HWo = (long)a.Ho * a.Wo;
#pragma unroll
    for (int c = 0; c < COUT; ++c) {
        const float sc = a.scale[c], sh = a.shift[c];
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v = fmaf(acc[i][c], sc, sh);
            o[i] = v > 0.f ? v : a.slope * v;
        }
        float* dst = a.y + (((long)n * (a.Do + 2) + oz + 1) * COUT + c) * HWo + (long)oy * a.Wo + ox;
        if (ox + 3 < a.Wo && (a.Wo & 3) == 0) {
            *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (ox + i < a.Wo) dst[i] = o[i];
        }
    }
}

template <int COUT, int STR>
static int launch_thin(const Thin3dArgs& a, int N, hipStream_t s) {
    constexpr int SX = (T3_TX - 1) * STR + 3, SY = (T3_TY - 1) * STR + 3, SZ = (T3_TZ - 1) * STR + 3;
    constexpr int RS = (SX + 3) / 4 * 4 + 4;
    constexpr size_t lds = ((size_t)t3_cc(STR) * SZ * SY * RS + (size_t)t3_cc(STR) * 27 * COUT) * sizeof(float);
    static_assert(lds <= 160 * 1024, "thin3d slab");
    auto kern = thin_conv3d_kernel<COUT, STR>;
    if (lds > 64 * 1024) {
        static std::atomic<unsigned> raised{0};
        if (int rc = raise_lds_limit(reinterpret_cast<const void*>(kern), raised)) return rc;
    }
    dim3 grid(cdiv(a.Wo, T3_TX) * cdiv(a.Ho, T3_TY), cdiv(a.Do, T3_TZ), N);
    NND_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "thin_conv3d: volume too deep for the grid");
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
    NND_LAUNCH_CHECK();
    return NND_OK;
}

bool thin3d_supported(int Cout, int stride) { return (Cout == 8 || Cout == 16) && (stride == 1 || stride == 2); }

// packed: [w (Ct*27*Cout) | scale (Cout) | shift (Cout)]
int64_t thin3d_packed_floats(int Cout, int Ct) { return (int64_t)Ct * 27 * Cout + 2 * Cout; }

void thin3d_pack(int Cout, int Ct, const float* w /* (Cout, Ct, 3,3,3) */, const float* scale, const float* shift, float* out) {
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Ct; ++ci)
            for (int t = 0; t < 27; ++t) out[((long)ci * 27 + t) * Cout + co] = w[((long)co * Ct + ci) * 27 + t];
    for (int co = 0; co < Cout; ++co) {
        out[(long)Ct * 27 * Cout + co] = scale[co];
        out[(long)Ct * 27 * Cout + Cout + co] = shift[co];
    }
}

int thin3d_forward(int Cout, int C0, int C1, int stride, const float* packed, const float* x0, const float* x1, float* y, int N, int D,
                   int H, int W, float slope, hipStream_t s) {
    NND_REQUIRE(thin3d_supported(Cout, stride), "thin_conv3d: Cout %d / stride %d not built", Cout, stride);
    NND_REQUIRE((long)(C0 + C1) * H * W < (1L << 31), "thin_conv3d: plane offsets exceed 32 bits");
    Thin3dArgs a;
    a.x0 = x0; a.x1 = x1; a.C0 = C0; a.C1 = C1;
    a.w = packed;
    a.scale = packed + (int64_t)(C0 + C1) * 27 * Cout;
    a.shift = a.scale + Cout;
    a.y = y;
    a.D = D; a.H = H; a.W = W;
    a.Do = (D + stride - 1) / stride; a.Ho = (H + stride - 1) / stride; a.Wo = (W + stride - 1) / stride;
    a.slope = slope;
    if (Cout == 8 && stride == 1) return launch_thin<8, 1>(a, N, s);
    if (Cout == 16 && stride == 1) return launch_thin<16, 1>(a, N, s);
    if (Cout == 8 && stride == 2) return launch_thin<8, 2>(a, N, s);
    return launch_thin<16, 2>(a, N, s);
}

}  // namespace nnd
