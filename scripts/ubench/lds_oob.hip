// Does the hardware confine DS accesses to the workgroup's own LDS allocation?  Every workgroup fills its 40 KB of dynamic LDS
// with its id, spins a while (so that several workgroups are resident on a CU), then the even workgroups WRITE 0xdeadbeef at
// offsets 40 KB .. 160 KB (outside their allocation) and READ from there; finally every workgroup checks its own 40 KB.
//   hipcc --offload-arch=gfx950 -O2 -o lds_oob lds_oob.hip && ./lds_oob
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) k(unsigned* bad, unsigned* seen) {
    extern __shared__ unsigned lds[];
    const int n = 40 * 1024 / 4;
    for (int i = threadIdx.x; i < n; i += 256) lds[i] = blockIdx.x + 1;
    __syncthreads();
    for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(32);
    if ((blockIdx.x & 1) == 0) {
        unsigned acc = 0;
        for (int i = n + threadIdx.x; i < 160 * 1024 / 4; i += 256) {
            volatile unsigned* p = lds + i;
            acc |= *p;            // out-of-range read: what comes back?
            *p = 0xdeadbeefu;     // out-of-range write
        }
        if (acc) atomicOr(seen, acc ? 1u : 0u);
    }
    for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(32);
    __syncthreads();
    unsigned c = 0;
    for (int i = threadIdx.x; i < n; i += 256) c += lds[i] != blockIdx.x + 1;
    if (c) atomicAdd(bad, c);
}
int main() {
    unsigned *bad, *seen, h[2] = {0, 0};
    hipMalloc(&bad, 4); hipMalloc(&seen, 4);
    hipMemset(bad, 0, 4); hipMemset(seen, 0, 4);
    hipLaunchKernelGGL(k, dim3(1024), dim3(256), 40 * 1024, 0, bad, seen);
    hipDeviceSynchronize();
    hipMemcpy(&h[0], bad, 4, hipMemcpyDeviceToHost); hipMemcpy(&h[1], seen, 4, hipMemcpyDeviceToHost);
    printf("LDS words of OTHER workgroups overwritten by out-of-range writes: %u; out-of-range reads returned non-zero data: %u\n", h[0], h[1]);
    return 0;
}
