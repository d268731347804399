// Micro-benchmark for DESIGN.md §7 (a): at the same MFMA work per cycle, does the chip sustain a higher clock (= more TFLOP/s under its
// power limit) with v_mfma_f32_16x16x32_f16 than with v_mfma_f32_32x32x16_f16?  Per wave NCH independent accumulator chains of one
// shape, operands in registers (no memory traffic in the loop), random-ish fp16 data (zeros would flatter the power); 1, 2 and 3 waves
// per SIMD on every CU; 20 ms of work per point; sustained TFLOP/s from HIP events and the shader clock from s_memtime / s_memrealtime.
//   hipcc -O3 --offload-arch=gfx950 scripts/ubench/mfma_shape_clock.hip -o scripts/ubench/mfma_shape_clock && scripts/ubench/mfma_shape_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE, int NCH>
__global__ void __launch_bounds__(256) chains(float* out, unsigned long long* clk, int iters, unsigned seed) {
    f16x8 a, b;
    unsigned s = seed ^ (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
    for (int i = 0; i < 8; ++i) {
        s = s * 1664525u + 1013904223u;
        a[i] = (_Float16)(((int)(s >> 20) & 1023) / 512.f - 1.f);
        s = s * 1664525u + 1013904223u;
        b[i] = (_Float16)(((int)(s >> 20) & 1023) / 512.f - 1.f);
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
    if constexpr (SHAPE == 32) {
        f32x16 acc[NCH];
        for (int c = 0; c < NCH; ++c)
            for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int c = 0; c < NCH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[c], 0, 0, 0);
        for (int c = 0; c < NCH; ++c)
            for (int i = 0; i < 16; ++i) sum += acc[c][i];
    } else {
        f32x4 acc[2 * NCH];  // two 16x16x32 = the FLOPs and the cycles of one 32x32x16
        for (int c = 0; c < 2 * NCH; ++c)
            for (int i = 0; i < 4; ++i) acc[c][i] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int c = 0; c < 2 * NCH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[c], 0, 0, 0);
        for (int c = 0; c < 2 * NCH; ++c)
            for (int i = 0; i < 4; ++i) sum += acc[c][i];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (threadIdx.x == 0 && blockIdx.x < 1024) {
        clk[2 * blockIdx.x] = t1 - t0;
        clk[2 * blockIdx.x + 1] = r1 - r0;
    }
}

template <int SHAPE, int NCH = 4>
static void run(int waves_per_simd, float* out, unsigned long long* clk) {
    const int wgs = 256 * waves_per_simd;  // 4 waves per workgroup: one per SIMD; `waves_per_simd` workgroups per CU
    const double flop_per_it = (double)wgs * 4 * NCH * 2.0 * 32 * 32 * 16;  // per loop iteration, whole grid
    int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {  // the second and third launch are timed long enough for the clock to settle
        hipEventRecord(e0);
        hipLaunchKernelGGL((chains<SHAPE, NCH>), dim3(wgs), dim3(256), 0, 0, out, clk, iters, 12345u + rep);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (rep == 0) iters = (int)(iters * 20.0 / ms);
    }
    std::vector<unsigned long long> h(2 * 1024);
    hipMemcpy(h.data(), clk, sizeof(unsigned long long) * 2 * 1024, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    const int n = wgs < 1024 ? wgs : 1024;
    for (int i = 0; i < n; ++i) cyc += (double)h[2 * i], rt += (double)h[2 * i + 1];
    const double ghz = cyc / rt * 0.1;  // s_memrealtime: 100 MHz
    printf("  %s x %d tiles  %d wave(s)/SIMD: %7.1f TFLOP/s   shader clock %.2f GHz   MFMA pipe busy %.0f %% of the cycles at that clock   (%d iterations, %.1f ms)\n",
           SHAPE == 32 ? "v_mfma_f32_32x32x16_f16" : "v_mfma_f32_16x16x32_f16", SHAPE == 32 ? NCH : 2 * NCH, waves_per_simd, flop_per_it * iters / ms / 1e9, ghz,
           100.0 * (flop_per_it * iters / ms / 1e9) / (1024.0 * ghz * 1.024), iters, ms);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

int main() {
    float* out;
    unsigned long long* clk;
    hipMalloc(&out, sizeof(float) * 256 * 1024);
    hipMalloc(&clk, sizeof(unsigned long long) * 2 * 1024);
    printf("dependent fp16 MFMA chains, independent accumulator tiles per wave as printed, operands in registers, no memory traffic in the loop\n");
    for (int w = 1; w <= 3; ++w) {
        run<32>(w, out, clk);
        run<16>(w, out, clk);
    }
    printf("more independent tiles per wave (is the 16x16x32 figure a dependency stall?)\n");
    run<32, 2>(2, out, clk);
    run<16, 8>(2, out, clk);
    run<16, 12>(1, out, clk);
    return 0;
}
