// Stand-alone reproducer attempt for the round-3 co-residency finding: does `v_pk_fma_f32` with operand-half
// selection (op_sel / op_sel_hi) return wrong values when waves of ANOTHER kernel running v_mfma_f32_32x32x16_f16 are resident on
// the same SIMD?  Victim: 8-wave workgroups (<= 128 VGPRs, like flow_branch_body / flow_head2_kernel<2>) that run chains of the
// instruction on known data and store the results; the host recomputes every chain with fmaf and counts mismatches.  Aggressor:
// 2-wave workgroups of dependent fp16 MFMAs on a second stream, small enough to share every CU with the victim.
//   hipcc -O2 -ffp-contract=off -Wno-unused-value --offload-arch=gfx950 scripts/ubench/pk_fma_hazard.hip -o scripts/ubench/pk_fma_hazard && scripts/ubench/pk_fma_hazard
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int ITERS = 256, NV = 4;  // NV independent chains per thread

template <int MODE>
__global__ void __launch_bounds__(512) victim(float* out, int reps) {
    const int t = blockIdx.x * 512 + threadIdx.x;
    for (int r = 0; r < reps; ++r) {
        f32x2 acc[NV], w[NV], v;
        for (int k = 0; k < NV; ++k) {
            acc[k] = f32x2{0.f, 0.f};
            w[k] = f32x2{1.0f + 0.001f * (float)((t + k) % 97), 0.5f + 0.002f * (float)((t + 3 * k) % 89)};
        }
        v = f32x2{0.75f + 0.0001f * (float)(t % 1013), 1.25f - 0.0002f * (float)(t % 911)};
        for (int i = 0; i < ITERS; ++i) {
#pragma unroll
            for (int k = 0; k < NV; ++k) {  // acc.lo = w.lo * v.X + acc.lo ; acc.hi = w.hi * v.Y + acc.hi
                if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc[k]) : "v"(w[k]), "v"(v));  // broadcast v.hi
                if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(acc[k]) : "v"(w[k]), "v"(v));  // broadcast v.lo
                if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "+v"(acc[k]) : "v"(w[k]), "v"(v));  // swapped halves
                if (MODE == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(w[k]), "v"(v));                                     // plain
            }
            v = f32x2{v[0] * 0.999f, v[1] * 1.0005f};  // plain VALU between the packed ones; exact same ops on the host
        }
        for (int k = 0; k < NV; ++k) {
            out[((size_t)t * NV + k) * 2 + 0] = acc[k][0];
            out[((size_t)t * NV + k) * 2 + 1] = acc[k][1];
        }
    }
}

__global__ void __launch_bounds__(128) aggressor(float* sink, int iters) {
    f32x16 a0 = {}, a1 = {};
    f16x8 x, y;
    for (int j = 0; j < 8; ++j) { x[j] = (_Float16)(0.01f * (float)(threadIdx.x + j)); y[j] = (_Float16)(0.02f * (float)(threadIdx.x ^ j)); }
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, a1, 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < 16; ++j) s += a0[j] + a1[j];
    if (s == 123.456f) sink[0] = s;
}

static void expect(int mode, int t, int k, float* lo, float* hi) {
    float alo = 0.f, ahi = 0.f, wlo = 1.0f + 0.001f * (float)((t + k) % 97), whi = 0.5f + 0.002f * (float)((t + 3 * k) % 89);
    float v0 = 0.75f + 0.0001f * (float)(t % 1013), v1 = 1.25f - 0.0002f * (float)(t % 911);
    for (int i = 0; i < ITERS; ++i) {
        const float xl = mode == 0 ? v1 : (mode == 1 ? v0 : (mode == 2 ? v1 : v0)), xh = mode == 0 ? v1 : (mode == 1 ? v0 : (mode == 2 ? v0 : v1));
        alo = fmaf(wlo, xl, alo);
        ahi = fmaf(whi, xh, ahi);
        v0 = v0 * 0.999f;
        v1 = v1 * 1.0005f;
    }
    *lo = alo; *hi = ahi;
}

template <int MODE>
static long run(bool with_aggressor, hipStream_t sv, hipStream_t sa, float* dout, float* sink, std::vector<float>& h, int nwg, float* ms) {
    hipMemsetAsync(dout, 0, h.size() * 4, sv);
    hipStreamSynchronize(sv);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    if (with_aggressor) hipLaunchKernelGGL(aggressor, dim3(256 * 8), dim3(128), 0, sa, sink, 60000);   // ~ tens of ms of MFMA waves on every SIMD
    hipEventRecord(e0, sv);
    for (int l = 0; l < 20; ++l) hipLaunchKernelGGL(victim<MODE>, dim3(nwg), dim3(512), 0, sv, dout, 20);
    hipEventRecord(e1, sv);
    hipStreamSynchronize(sv);
    hipEventElapsedTime(ms, e0, e1);  // the victim's launches take longer while they share the SIMDs with the aggressor's waves
    hipMemcpy(h.data(), dout, h.size() * 4, hipMemcpyDeviceToHost);
    hipStreamSynchronize(sa);
    long bad = 0;
    for (int t = 0; t < nwg * 512; ++t)
        for (int k = 0; k < NV; ++k) {
            float lo, hi;
            expect(MODE, t, k, &lo, &hi);
            bad += (h[((size_t)t * NV + k) * 2] != lo) + (h[((size_t)t * NV + k) * 2 + 1] != hi);
        }
    return bad;
}

int main() {
    const int nwg = 512;
    std::vector<float> h((size_t)nwg * 512 * NV * 2);
    float *dout, *sink;
    hipMalloc(&dout, h.size() * 4);
    hipMalloc(&sink, 64);
    hipStream_t sv, sa;
    hipStreamCreate(&sv);
    hipStreamCreate(&sa);
    const char* names[4] = {"op_sel:[0,1,0] op_sel_hi:[1,1,1] (v.hi broadcast)", "op_sel:[0,0,0] op_sel_hi:[1,0,1] (v.lo broadcast)",
                            "op_sel:[0,1,0] op_sel_hi:[1,0,1] (halves swapped)", "no operand selection"};
    for (int pass = 0; pass < 2; ++pass) {
        float ms[4];
        long b[4] = {run<0>(pass, sv, sa, dout, sink, h, nwg, ms), run<1>(pass, sv, sa, dout, sink, h, nwg, ms + 1),
                     run<2>(pass, sv, sa, dout, sink, h, nwg, ms + 2), run<3>(pass, sv, sa, dout, sink, h, nwg, ms + 3)};
        for (int m = 0; m < 4; ++m)
            printf("%-28s v_pk_fma_f32 %-52s mismatches %ld of %zu   (victim launches: %.2f ms)\n", pass ? "beside fp16-MFMA aggressor:" : "alone:", names[m], b[m],
                   h.size(), ms[m]);
    }
    return 0;
}
