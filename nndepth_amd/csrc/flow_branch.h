// Flow branch of the motion encoder as one workgroup program (arithmetics 2 / 3): device body shared by flow_branch_kernel
// (conv_split.hip) and flow_branch_lookup_kernel (corr1d.hip).  Design notes: conv_split.hip.
#pragma once
#include "conv_split_kernel.h"

namespace nnd {

#ifdef NND_DBG_STAMPS
static __device__ unsigned long long g_fb_stamps[4096 * 8];
#define NND_FSTAMP(i)                                                                                        \
    do {                                                                                                     \
        if (threadIdx.x == 0 && bx < 4096) g_fb_stamps[bx * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define NND_FSTAMP(i)
#endif

struct FlowBranchArgs {
    const float* flow;  // (B, FC, H, W) planar tile-major
    long fbs;
    const float* w7t;   // [FC*49][128]: convf1's weights tap-major (transposed at pack time)
    const float* b7;    // [128]
    ConvArgs c;         // convf2: wpk (pack_conv_split order), bias, out0 / obs0, ld, H, W, Cout = 64, epi = EPI_RELU, tiles_x, npos
    Lay lf;             // layout of `flow`
};

constexpr int FB_C1 = 128, FB_NCH = FB_C1 / 16, FB_PR = 6, FB_PC = 10, FB_WR = FB_PR + 6, FB_WC = FB_PC + 6;
__host__ __device__ constexpr int fb_rowb(int NS) { return split_row_bytes(FB_PC, NS); }
__host__ __device__ constexpr int fb_subb(int NS) { return FB_PR * fb_rowb(NS); }
// LDS: [flow window | 7x7 weights | their bias] [convf2's patch]; the exchange buffer of phase 3 (3 slices x 2 blocks x 4 KB) aliases
// the first region, which is dead once phase 1 has finished (a barrier separates them)
template <int FC, int NS>
__host__ __device__ constexpr int fb_lds_bytes() {
    constexpr int head = 4 * (FC * FB_WR * FB_WC + FC * 49 * FB_C1 + FB_C1), red = 2 * 3 * 1024 * 4;
    return (head > red ? head : red) + FB_NCH * fb_subb(NS);
}

// the workgroup's work as a device function of its (tile, batch) index: flow_branch_kernel launches it alone,
// flow_branch_lookup_kernel (corr1d.hip) beside the lookup + convc1 workgroups of the same iteration
template <int FC, int NS>
__device__ __forceinline__ void flow_branch_body(const FlowBranchArgs& a, unsigned char* lds_raw, const int bx, const int b) {
    constexpr int PS = split_pos_bytes(NS), ROWB = fb_rowb(NS), SUBB = fb_subb(NS);
    constexpr int HEAD = 4 * (FC * FB_WR * FB_WC + FC * 49 * FB_C1 + FB_C1), RED = 2 * 3 * 1024 * 4;
    float* win = reinterpret_cast<float*>(lds_raw);                 // [FC][12][16]
    float* w7l = win + FC * FB_WR * FB_WC;                          // [FC*49][128]
    float* b7l = w7l + FC * 49 * FB_C1;                             // [128]
    unsigned char* patch = lds_raw + (HEAD > RED ? HEAD : RED);     // [8 chunks][6][ROWB]
    float* red = reinterpret_cast<float*>(lds_raw);                 // [3 slices][2 blocks][16 regs][64 lanes]: aliases win / w7l / b7l

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = a.c.H, W = a.c.W;
    const int ty0 = (bx / a.c.tiles_x) * 4, tx0 = (bx % a.c.tiles_x) * 8;
    const long FP = a.lf.plane;
    NND_FSTAMP(0);

    // wave = (output block, K slice) of phase 2 — conv_split's wave = kj * wco + cbi with wco = 2, ks = 4.  Its first FB_AD weight
    // fragments are requested NOW: they depend on nothing and arrive while phases 0 and 1 run.
    const int cbi = wave & 1, kj = wave >> 1;
    const uint4* wq = reinterpret_cast<const uint4*>(a.c.wpk) + (size_t)cbi * FB_NCH * (9 * NS * 64) + lane;
    float oscale = 1.f, xscale = 1.f;
    if constexpr (NS == 2) {  // undoes the fp16 range scaling / the layer's activation scale (split_arith.h)
        const float* tail = a.c.bias + (((a.c.Cout + 31) >> 5) << 5);
        oscale = tail[SPLIT_TAIL_OSCALE];
        xscale = tail[SPLIT_TAIL_XSCALE];
    }
    constexpr int NSTEP = (FB_NCH / 4) * 9;  // (chunk, tap) steps of a K slice
    constexpr int FB_AD = 5, NA = FB_AD + 1;
    uint4 ab[NA][NS];
    auto load_a = [&](uint4 (&dst)[NS], int st) {
        const int ch = 4 * (st / 9) + kj, t = st % 9;
        const uint4* w = wq + (size_t)(ch * 9 + t) * (NS * 64);
#pragma unroll
        for (int sp = 0; sp < NS; ++sp) dst[sp] = w[sp * 64];
    };
#pragma unroll
    for (int st = 0; st < FB_AD; ++st) load_a(ab[st], st);
    __builtin_amdgcn_sched_barrier(0);

    // ---- phase 0
    for (int e = tid; e < FC * FB_WR * FB_WC; e += 512) {
        const int cch = e / (FB_WR * FB_WC), rr = (e / FB_WC) % FB_WR, cc = e % FB_WC;
        const int gy = ty0 + rr - 4, gx = tx0 + cc - 4;  // window row rr = patch row rr - 3 = image row ty0 - 1 + (rr - 3)
        win[e] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? a.flow[b * a.fbs + cch * FP + pix_off(a.lf, gy, gx)] : 0.f;
    }
    for (int e = tid; e < FC * 49 * FB_C1 / 4; e += 512)
        reinterpret_cast<float4*>(w7l)[e] = reinterpret_cast<const float4*>(a.w7t)[e];
    if (tid < FB_C1) b7l[tid] = a.b7[tid];
    __syncthreads();
    NND_FSTAMP(1);

    // ---- phase 1
    {
        const int pp = tid & 31, gh = tid >> 5;  // position pair (30 of them), (16-channel chunk, 8-channel half): all 8 waves
        const int g = gh >> 1, h = gh & 1;
        if (pp < 30) {
            const int pr = pp / 5, pc0 = (pp % 5) * 2;
            float acc[2][8];
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[p][j] = 0.f;
#pragma unroll 1
            for (int cch = 0; cch < FC; ++cch)
#pragma unroll 1
                for (int dy = 0; dy < 7; ++dy) {  // rolled: one window row (8 values) and one tap's 8 weights live at a time
                    float v[8];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {  // pc0 is even: 8-byte aligned pairs
                        const float2 t2 = *reinterpret_cast<const float2*>(win + (cch * FB_WR + pr + dy) * FB_WC + pc0 + 2 * q);
                        v[2 * q] = t2.x; v[2 * q + 1] = t2.y;
                    }
#pragma unroll
                    for (int dx = 0; dx < 7; ++dx) {
                        const float4* wt = reinterpret_cast<const float4*>(w7l + ((cch * 49 + dy * 7 + dx) * FB_C1 + g * 16 + h * 8));
                        const float4 w0 = wt[0], w1 = wt[1];
                        const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
                        for (int p = 0; p < 2; ++p)
#pragma unroll
                            for (int j = 0; j < 8; ++j) fmac_scalar(acc[p][j], wv[j], v[dx + p]);  // never a packed FMA (common.h)
                    }
                }
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int pc = pc0 + p;
                const int gy = ty0 + pr - 1, gx = tx0 + pc - 1;
                const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
                float val[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) val[j] = in ? fmaxf(acc[p][j] + b7l[g * 16 + 8 * h + j], 0.f) : 0.f;
                uint4 pieces[NS];
                split_pieces<NS>(val, pieces, xscale);
#pragma unroll
                for (int sp = 0; sp < NS; ++sp)
                    *reinterpret_cast<uint4*>(patch + g * SUBB + pr * ROWB + pc * PS + sp * 32 + h * 16) = pieces[sp];
            }
        }
    }
    __syncthreads();
    NND_FSTAMP(2);

    // ---- phase 2
    const int h2 = lane >> 5, l31 = lane & 31;
    const int pxl = lane_pixel(l31), r = pxl >> 3, c = pxl & 7;
    f32x16 acc[1];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[0][i] = 0.f;
    const unsigned char* xb = patch + r * ROWB + c * PS + h2 * 16;
    // convf2's bias for this lane's 16 output channels, requested before the walk (the shared epilogue would fetch it after the
    // K-slice sum, on the kernel's critical tail)
    float bias_r[16];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) bias_r[reg] = a.c.bias[cbi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h2];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
        const int ch = 4 * (st / 9) + kj, t = st % 9;
        if (st + FB_AD < NSTEP) load_a(ab[(st + FB_AD) % NA], st + FB_AD);
        uint4 bq[NS];
#pragma unroll
        for (int sp = 0; sp < NS; ++sp) bq[sp] = *reinterpret_cast<const uint4*>(xb + ch * SUBB + (t / 3) * ROWB + (t % 3) * PS + sp * 32);
        __builtin_amdgcn_sched_barrier(0);  // keep the prefetch where it is: the scheduler otherwise sinks every load to its use
        split_mfma_step<NS>(ab[st % NA], bq, acc[0]);
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (NS == 2) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) acc[0][reg] *= oscale;
    }

    NND_FSTAMP(3);
    // ---- phase 3
    if (kj > 0) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) red[(((kj - 1) * 2 + cbi) * 16 + reg) * 64 + lane] = acc[0][reg];
    }
    __syncthreads();
    if (kj > 0) return;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        float sum = acc[0][reg];
#pragma unroll
        for (int sl = 0; sl < 3; ++sl) sum += red[((sl * 2 + cbi) * 16 + reg) * 64 + lane];
        acc[0][reg] = sum;
    }
    // epilogue: relu(sum + bias), the arithmetic of conv_epilogue's EPI_RELU; 16-B stores in the c4 layout
    const int y = ty0 + r, x = tx0 + c;
    if (y < H && x < W) {
        const long DP = a.c.ld.plane, pix = pix_off(a.c.ld, y, x);
        float* o = a.c.out0 + b * a.c.obs0 + pix;
        if (a.c.ld.ci == 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int co0 = cbi * 32 + 8 * q + 4 * h2;
                float4 v4;
                v4.x = fmaxf(acc[0][4 * q] + bias_r[4 * q], 0.f);
                v4.y = fmaxf(acc[0][4 * q + 1] + bias_r[4 * q + 1], 0.f);
                v4.z = fmaxf(acc[0][4 * q + 2] + bias_r[4 * q + 2], 0.f);
                v4.w = fmaxf(acc[0][4 * q + 3] + bias_r[4 * q + 3], 0.f);
                *reinterpret_cast<float4*>(o + (long)co0 * DP) = v4;
            }
        } else {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int co = cbi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h2;
                o[(long)co * DP] = fmaxf(acc[0][reg] + bias_r[reg], 0.f);
            }
        }
    }
#ifdef NND_DBG_STAMPS
    __builtin_amdgcn_s_waitcnt(0);
#endif
    NND_FSTAMP(4);
}

template <int FC, int NS>
__global__ void __launch_bounds__(512) flow_branch_kernel(FlowBranchArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fb_lds[];
    flow_branch_body<FC, NS>(a, fb_lds, (int)blockIdx.x, (int)blockIdx.z);
}


// fills the kernel arguments of the flow branch (conv_split.hip); launch_flow_branch and flow_branch_lookup_launch (corr1d.hip) use it
int make_flow_branch_args(const ConvLayer& f2, const float* blob, const float* w7t, const float* b7, const float* flow, int64_t fbs,
                          int fc, const ConvIO& io, int B, int H, int W, FlowBranchArgs* out);

}  // namespace nnd
