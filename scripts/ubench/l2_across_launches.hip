// Micro-benchmark: does a per-XCD L2 keep data across kernel launches on one stream?  A single wave walks a pointer chain (one dependent
// 4-byte load per step, 64-byte stride, `n` steps inside a buffer of `n * 64` bytes) and reports cycles per load:
//   (1) cold (buffer just written by the host copy), (2) again in the SAME launch (second pass: L2 / L1 warm),
//   (3) in the NEXT launch on the same stream, the previous launch having walked the same chain on the same workgroup slot,
//   (4) in the next launch after another kernel wrote the buffer from all CUs (the producer -> consumer case of the conv loop).
// hipcc -O3 --offload-arch=gfx950 scripts/ubench/l2_across_launches.hip -o scripts/ubench/l2_across_launches
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void walk(const int* buf, int n, int passes, unsigned long long* out, int* sink) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int p = 0;
    for (int pass = 0; pass < passes; ++pass) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < n; ++i) p = buf[p];
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        out[pass] = t1 - t0;
    }
    *sink = p;
}
__global__ void rewrite(int* buf, int n_ints, int stride_ints, int n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_ints) {
        const int slot = (int)(i / stride_ints);
        if (i % stride_ints == 0) buf[i] = ((slot * 7 + 3) % n) * stride_ints;  // the same permutation the host wrote
    }
}
int main() {
    const int stride = 16;  // ints = 64 bytes
    for (int kb : {256, 2048}) {
        const int n = kb * 1024 / 64, n_ints = n * stride;
        std::vector<int> h(n_ints, 0);
        for (int s = 0; s < n; ++s) h[(size_t)s * stride] = ((s * 7 + 3) % n) * stride;
        int *buf, *sink;
        unsigned long long* out;
        hipMalloc(&buf, sizeof(int) * n_ints);
        hipMalloc(&sink, 4);
        hipMalloc(&out, 64);
        hipMemcpy(buf, h.data(), sizeof(int) * n_ints, hipMemcpyHostToDevice);
        unsigned long long r[4];
        hipLaunchKernelGGL(walk, dim3(1), dim3(64), 0, 0, buf, n, 2, out, sink);
        hipDeviceSynchronize();
        hipMemcpy(r, out, 16, hipMemcpyDeviceToHost);
        const double cold = (double)r[0] / n, warm_same = (double)r[1] / n;
        hipLaunchKernelGGL(walk, dim3(1), dim3(64), 0, 0, buf, n, 1, out, sink);
        hipDeviceSynchronize();
        hipMemcpy(r, out, 8, hipMemcpyDeviceToHost);
        const double next_launch = (double)r[0] / n;
        hipLaunchKernelGGL(rewrite, dim3((n_ints + 255) / 256), dim3(256), 0, 0, buf, n_ints, stride, n);
        hipLaunchKernelGGL(walk, dim3(1), dim3(64), 0, 0, buf, n, 1, out, sink);
        hipDeviceSynchronize();
        hipMemcpy(r, out, 8, hipMemcpyDeviceToHost);
        const double after_producer = (double)r[0] / n;
        printf("%5d KB chain: cycles per dependent load (s_memtime ticks): cold %.0f | second pass, same launch %.0f | next launch %.0f | next launch after a "
               "producer kernel rewrote it %.0f\n", kb, cold, warm_same, next_launch, after_producer);
        hipFree(buf); hipFree(sink); hipFree(out);
    }
    return 0;
}
