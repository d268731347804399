// Thin 3x3x3 Conv3d layers of IGEV's cost-volume regulariser (ConvBn3D / Upsampler3D with 8 or 16 output channels,
// nndepth/models/igev_stereo/cost_volume.py:101-130, 176-190) as a direct fp32 VALU convolution on depth-major volumes.
//
// Why not the MFMA path (conv3d.hip): with Cout = 8 / 16 a 32-row MFMA tile is filled only by grouping J = 4 / 2 output
// slices, which makes half / a quarter of K structural zeros, and each 32-pixel tile then carries the full prologue /
// epilogue / staging overhead of the generic kernel for very little K: conv1_up + final_conv ran at 26 / 19 TFLOP/s and took
// 3.4 ms of the regulariser's 8.3 ms per 544x960 sample (profiles/r02_igev_regulariser_layers_before.txt).  Here:
//   thread  = 4 consecutive x outputs x 1 y x 1 depth slice x all COUT channels (4*COUT accumulators, an exact fmaf chain),
//   weights = wave-uniform: [ci][kd][kh][kw][co] in global memory, read through the scalar cache as SGPR operands of v_fma
//             (the whole layer is <= 27.6 KB),
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
    extern __shared__ __attribute__((aligned(16))) float slab[];  // [T3_CC][SZ][SY][RS]
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
        // ---- stage T3_CC channels of the slab (zero outside the volume; the depth padding is in the tensor itself)
        for (int e = tid; e < T3_CC * SZ * SY * SX; e += 256) {
            const int sx = e % SX, r1 = e / SX;
            const int sy = r1 % SY, r2 = r1 / SY;
            const int sz = r2 % SZ, cc = r2 / SZ;
            const int gx = ix0 + sx, gy = iy0 + sy, gz = iz0 + sz;  // gz: padded slice index 0..D+1
            const int ci = cbase + cc;
            float v = 0.f;
            if (ci < Ct && gx >= 0 && gx < a.W && gy >= 0 && gy < a.H && gz <= a.D + 1) {
                const float* src = ci < a.C0 ? a.x0 + (((long)n * (a.D + 2) + gz) * a.C0 + ci) * HW
                                             : a.x1 + (((long)n * (a.D + 2) + gz) * a.C1 + (ci - a.C0)) * HW;
                v = src[(long)gy * a.W + gx];
            }
            slab[cc * CHS + sz * PLANE + sy * RS + sx] = v;
        }
        __syncthreads();
#pragma unroll 1
        for (int cc = 0; cc < T3_CC; ++cc) {
            const int ci = cbase + cc;
            if (ci >= Ct) break;
            const float* wc = a.w + (long)ci * 27 * COUT;  // wave-uniform: scalar loads
#pragma unroll
            for (int kd = 0; kd < 3; ++kd)
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const float* row = slab + cc * CHS + (tz * STR + kd) * PLANE + (ty * STR + kh) * RS + tx * 4 * STR;
                    constexpr int NV = 3 * STR + 3;  // input x values feeding 4 outputs: 6 (stride 1) or 9 (stride 2)
                    float xv[NV + (4 - NV % 4) % 4];
#pragma unroll
                    for (int q = 0; q < (NV + 3) / 4; ++q) {
                        const float4 t = *reinterpret_cast<const float4*>(row + 4 * q);
                        xv[4 * q] = t.x; xv[4 * q + 1] = t.y; xv[4 * q + 2] = t.z; xv[4 * q + 3] = t.w;
                    }
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const float* wt = wc + ((kd * 3 + kh) * 3 + kw) * COUT;
#pragma unroll
                        for (int c = 0; c < COUT; ++c) {
                            const float wv = wt[c];
#pragma unroll
                            for (int i = 0; i < 4; ++i) acc[i][c] = fmaf(wv, xv[i * STR + kw], acc[i][c]);
                        }
                    }
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
    constexpr size_t lds = (size_t)t3_cc(STR) * SZ * SY * RS * sizeof(float);
    static_assert(lds <= 160 * 1024, "thin3d slab");
    auto kern = thin_conv3d_kernel<COUT, STR>;
    if (lds > 64 * 1024) {
        static bool raised = false;
        if (!raised) {
            NND_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            raised = true;
        }
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
