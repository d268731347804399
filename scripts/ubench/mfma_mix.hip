// Micro-benchmark: how fast does one SIMD issue dependent v_mfma_f32_32x32x2_f32 chains when each group of
// 16 MFMAs is accompanied by the fetch work of the conv kernel (4 x global_load_dwordx4 of weights through
// L2, 16 x ds_read_b32 of activations)?   hipcc -O3 --offload-arch=gfx950 mfma_mix.hip -o mfma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <type_traits>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int GL, int DS, int SPLIT>
__global__ void __launch_bounds__(256) mix(const float4* __restrict__ w, float* out, int steps, int wstride) {
    __shared__ float lds[5 * 1024];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 5 * 1024; i += 256) lds[i] = i * 1e-6f;
    __syncthreads();
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const float4* wp = w + (size_t)((blockIdx.x * 4 + wave) % 64) * wstride;
    float4 a[2][4];
    float b[2][16];
    for (int q = 0; q < 4; ++q) a[0][q] = wp[q * 64 + lane];
    for (int i = 0; i < 16; ++i) b[0][i] = lds[i * 288 + lane];
    const float* xb = lds + lane;
    auto step = [&](int s, auto cur_c) {
        constexpr int cur = decltype(cur_c)::value, nxt = cur ^ 1;
        const float4* ws = wp + (size_t)((s + 1) & 1023) * 256;
        if (GL) {
#pragma unroll
            for (int q = 0; q < 4; ++q) a[nxt][q] = ws[q * 64 + lane];
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) a[nxt][q] = a[cur][q];
        }
        if (!SPLIT) {
#pragma unroll
            for (int i = 0; i < 16; ++i) b[nxt][i] = DS ? xb[i * 288 + (s & 7)] : b[cur][i];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (SPLIT && (i == 8 || i == 12)) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = (i == 8 ? 0 : 8); j < (i == 8 ? 8 : 16); ++j) b[nxt][j] = DS ? xb[j * 288 + (s & 7)] : b[cur][j];
                __builtin_amdgcn_sched_barrier(0);
            }
            const float4 av = a[cur][i / 4];
            const float as = (i % 4 == 0) ? av.x : (i % 4 == 1) ? av.y : (i % 4 == 2) ? av.z : av.w;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(as, b[cur][i], acc, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int s = 0; s < steps; s += 2) {
        step(s, std::integral_constant<int, 0>{});
        step(s + 1, std::integral_constant<int, 1>{});
    }
    float sum = 0.f;
    for (int i = 0; i < 16; ++i) sum += acc[i];
    if (sum == 1.2345f) out[0] = sum;
}

template <int GL, int DS, int SPLIT>
void run(const char* name, const float4* w, float* out, int wps) {
    const int steps = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid(256 * wps), block(256);
    hipLaunchKernelGGL((mix<GL, DS, SPLIT>), grid, block, 0, 0, w, out, 10, 1024 * 256);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((mix<GL, DS, SPLIT>), grid, block, 0, 0, w, out, steps, 1024 * 256);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = 2.0 * 32 * 32 * 2 * 16.0 * steps * 4.0 * 256 * wps;
    printf("%-34s waves/SIMD %d : %7.1f TFLOP/s  (%.3f ms)\n", name, wps, fl / ms / 1e9, ms);
}

int main() {
    float4* w; float* out;
    size_t n = (size_t)64 * 1024 * 256;  // 64 weight streams x 4 MB
    hipMalloc(&w, n * sizeof(float4)); hipMalloc(&out, 64);
    hipMemset(w, 0, n * sizeof(float4));
    for (int wps : {1, 2, 3}) {
        run<0, 0, 0>("mfma only", w, out, wps);
        run<0, 1, 0>("mfma + 16 ds_read (burst)", w, out, wps);
        run<0, 1, 1>("mfma + 16 ds_read (8+8 slotted)", w, out, wps);
        run<1, 0, 0>("mfma + 4 dwordx4 L2", w, out, wps);
        run<1, 1, 0>("mfma + both (burst)", w, out, wps);
        run<1, 1, 1>("mfma + both (slotted)", w, out, wps);
    }
    return 0;
}
