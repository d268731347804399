// Fused mask head tail + convex upsample:   up = convex_upsample(flow, 0.25 * conv1x1(x))
//   x = relu(mask.0(h)) (B, CIN, H, W)  ->  mask (9*R*R channels, NEVER written to HBM)  ->  softmax over the 9
//   neighbours  ->  up (B, 1, R*H, R*W).
// Replaces nndepth/blocks/update_block.py:97-101,111 (mask.2 and the 0.25 scale) followed by
// nndepth/models/raft_stereo/model.py:93-105 (convex_upsample); saves writing + re-reading the 18.8 MB mask per
// iteration ("K8 fused with the 1x1 mask conv's epilogue").
//
// One workgroup = one 4x8 pixel tile x ALL 9*R*R mask channels:
//   * the whole x tile (CIN x 32 px) is staged in LDS once (no per-chunk barriers in the K loop);
//   * G x 2 waves: wave (g, kj) accumulates CBW output-channel blocks over half of K on the fp32 MFMA with the
//     same packed-weight stream as conv_mfma (A: global -> VGPR in fragment order, B: LDS);
//   * the two K-halves are summed into an LDS mask tile [9*R*R][33]; then every thread produces output pixels:
//     9 logits from LDS -> softmax -> weighted sum of the 3x3 flow neighbourhood -> R*8 contiguous floats per
//     (tile row, sub-row) so the HBM stores are whole 256-B runs.
// NS = 3 (arithmetic 3, round 2): the same GEMM with both operands as 3 bf16 pieces and 6 products per fp32 product on
// v_mfma_f32_32x32x16_bf16 (conv_split.hip's arithmetic; weights in its packing); NS = 2 (arithmetic 2, round 3): 2 range-scaled
// fp16 pieces, 3 products on v_mfma_f32_32x32x16_f16 (split_arith.h); NS = 0: the exact fp32 MFMA.  Phase stamps of the exact kernel at 68x120
// (scripts/stamps_mu.py): stage 3.8 us, K loop 22 (2.4 GFLOP on the fp32 MFMA = 15.3 us at peak: the kernel is matrix-bound),
// softmax + store 3.2.  The x tile is staged as [16-ch chunk][piece][k half][pixel][8 bf16], so a B fragment is one
// conflict-free ds_read_b128 per piece, shared by the wave's 3 output-channel blocks.
#include "common.h"

#include <type_traits>
#include "layout.h"
#include "split_arith.h"

namespace nnd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifdef NND_DBG_STAMPS
// debug build only (scripts/build_ablate.sh): per-workgroup phase timestamps, s_memrealtime at 100 MHz
__device__ unsigned long long g_mu_stamps[4096 * 8];
#define NND_MSTAMP(i)                                                                             \
    do {                                                                                          \
        if (threadIdx.x == 0 && blockIdx.x < 4096) g_mu_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define NND_MSTAMP(i)
#endif

struct MaskUpArgs {
    const float* x;
    long xbs;
    const float* wpk;
    const float* bias;
    const float* flow;  // (B,FC,H,W)
    float* out;         // (B,FC,R*H,R*W)
    int H, W, tiles_x, fc;  // fc: flow channels, 1 (RAFT/IGEV disparity) or 2 (CREStereo flow)
    Lay lay;  // layout of x and flow (tile-major inside the loop, NCHW through the C-ABI)
    int x_c4;  // x keeps 4 channels interleaved (tile-major c4, layout.h); flow stays planar
    int lds_floats;  // dynamic LDS of this launch; the flow patch sits in its last 128 floats
    MaskUpFlowHead fh;  // fh.x != nullptr: flow_head.conv2 + the recurrence update run in this launch (below)
};

template <int RATE, int CIN>
struct MaskUpCfg {
    static constexpr int COUT = 9 * RATE * RATE;
    static constexpr int NCB = (COUT + 31) / 32;
    static constexpr int CBW = NCB >= 12 ? 3 : 1;        // channel blocks per wave
    static constexpr int G = (NCB + CBW - 1) / CBW;      // wave groups along Cout
    static constexpr int NWAVES = 2 * G;                 // x 2 K-halves
    static constexpr int NT = 64 * NWAVES;
    static constexpr int NST = CIN / 32;                 // 32-channel steps over K
    static constexpr int MT_STRIDE = 33;
    static constexpr int XS_FLOATS = CIN * 32 * 3 / 2;   // x tile: fp32 (CIN*32 floats) or 3 bf16 pieces (1.5x that)
    static constexpr int LDS_FLOATS = (XS_FLOATS > COUT * MT_STRIDE ? XS_FLOATS : COUT * MT_STRIDE) + 128;
};

// FHW: extra waves (beyond the GEMM's NT threads) that run the folded flow head CONCURRENTLY with the staging and the K loop of the
// mask GEMM — the flow head is a latency chain (hidden map from HBM -> 60 dot products -> new state) that nothing in the GEMM waits
// for until the upsample reads the flow patch; its LDS lies behind the GEMM's.  The four barriers of the GEMM path (x tile staged |
// K loop done | first K half in the mask tile | second) are the flow head's: patch staged | partial sums | new flow patch written |
// (nothing).  Instantiated for fp16x2 (105 VGPRs: 16 waves per CU fit; the other arithmetics' GEMM waves need more than 128).  The
// same work by all threads in front of the x tile was measured too: never faster than the separate launch, 5 % slower at batch 8.
template <int RATE, int CIN, int NS = 0, int FHW = 0>
__global__ void __launch_bounds__((MaskUpCfg<RATE, CIN>::NT + 64 * FHW)) mask_upsample_kernel(MaskUpArgs a) {
    constexpr bool SPLIT = NS != 0;
    using Cfg = MaskUpCfg<RATE, CIN>;
    constexpr int COUT = Cfg::COUT, NCB = Cfg::NCB, CBW = Cfg::CBW, G = Cfg::G, NT = Cfg::NT, NST = Cfg::NST;
    constexpr int CI_T = 128;                 // packing of a 1x1 conv with Cin >= 128 (conv_ci_t)
    constexpr int NCHUNK = CIN / CI_T;
    constexpr int MTS = Cfg::MT_STRIDE;
    static_assert(CIN % 128 == 0 && NST % 2 == 0, "CIN must be a multiple of 128");
    extern __shared__ float lds[];
    NND_MSTAMP(0);
    float* xs = lds;                          // [CIN][32]   (K loop)
    float* mt = lds;                          // [COUT][33]  (after the K loop; aliases xs)
    float* fp = lds + a.lds_floats - 128;     // [fc][6][10] flow patch, zero outside the image

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave % G, kj = wave / G;
    const int h2 = lane >> 5, l31 = lane & 31;
    const int tx0 = (blockIdx.x % a.tiles_x) * 8, ty0 = (blockIdx.x / a.tiles_x) * 4;
    const int b = blockIdx.z;
    const int H = a.H, W = a.W;
    const long HW = (long)H * W;
    const long XP = a.lay.plane;
    float oscale = 1.f, xscale = 1.f;  // fp16x2: undoes the power-of-two range scaling of both operands / the layer's activation scale (packed behind the bias, split_arith.h)
    if constexpr (NS == 2) {
        oscale = a.bias[Cfg::NCB * 32 + SPLIT_TAIL_OSCALE];
        xscale = a.bias[Cfg::NCB * 32 + SPLIT_TAIL_XSCALE];
    }

    // ---- folded flow_head.conv2 + recurrence update (MaskUpFlowHead): run by FHW extra waves (below); their LDS (patch, weights,
    // partial sums) at `fbase`, behind the GEMM's.  Last step: the 16 slice sums + bias, the new state, the flow patch.
    float fh_state = 0.f, fh_bias = 0.f;
    auto fh_finish = [&](float* fbase, int t0) {
        if (t0 >= 60) return;
        const float* part = fbase + a.fh.hid * 105;
        const int pr = t0 / 10, pc = t0 % 10;
        const int y = ty0 + pr - 1, x = tx0 + pc - 1;
        float sum = 0.f;
#pragma unroll
        for (int sl = 0; sl < 16; ++sl) sum += part[sl * 64 + t0];
        sum += fh_bias;
        const bool in = y >= 0 && y < H && x >= 0 && x < W;
        const float cnew = fh_state + sum;
        const float fl = a.fh.absolute ? cnew : cnew - (float)x;
        fp[t0] = in ? fl : 0.f;
        if (in && pr >= 1 && pr <= 4 && pc >= 1 && pc <= 8) {  // the tile's own pixels
            const long pix = pix_off(a.lay, y, x);
            a.fh.delta_out[b * XP + pix] = sum;
            a.fh.coords_out[b * XP + pix] = cnew;
            a.fh.flow_out[b * XP + pix] = fl;
            a.fh.hx_flow[b * a.fh.hx_bs + pix * a.fh.hx_pm] = fl;
        }
    };
    // the x tile's global loads of the split arithmetics are issued first: they are in flight while the folded flow head works
    constexpr int NQ4 = (CIN / 4 * 32 + NT - 1) / NT;
    float4 v4[SPLIT ? NQ4 : 1];
    if constexpr (SPLIT) {  // (the flow-head waves, tid >= NT, load clamped addresses and never use them)
        const float* src = a.x + b * a.xbs;
#pragma unroll
        for (int i = 0; i < NQ4; ++i) {
            const int e = min(tid, NT - 1) + i * NT;
            const int qd = e >> 5, px = e & 31;
            const int y = ty0 + (px >> 3), x = tx0 + (px & 7);
            const bool ok = e < CIN / 4 * 32 && y < H && x < W;
            if (a.x_c4) {
                v4[i] = *reinterpret_cast<const float4*>(src + (ok ? (unsigned)(qd * 4 * (int)XP + 4 * (int)pix_off(a.lay, y, x)) : 0u));
            } else {
                const unsigned o = ok ? (unsigned)(qd * 4 * (int)XP + (int)pix_off(a.lay, y, x)) : 0u;
                v4[i] = make_float4(src[o], src[ok ? o + (unsigned)XP : 0u], src[ok ? o + 2u * (unsigned)XP : 0u], src[ok ? o + 3u * (unsigned)XP : 0u]);
            }
        }
    }
    if constexpr (FHW > 0) {
        if (tid >= NT) {
            // concurrent fold: flow-head wave w stages and multiplies its own hid / FHW channels (slices 4w .. 4w + 3).  Its 12 patch loads
            // per lane are ISSUED here, in front of the first barrier — beside the x tile's, before the GEMM's weight stream takes the
            // vector-memory pipe — but that barrier is a bare s_barrier for these waves (they publish nothing there; __syncthreads would
            // wait for the loads and with it hold the GEMM back: staging phase 2.8 -> 6.6 us).  The data is waited for behind it, goes to
            // LDS, is multiplied (wave-local: no workgroup barrier in between) while the GEMM runs its K loop; the 16 slice sums meet
            // behind the K loop's barrier, the flow patch is there one barrier later.
            float* fbase = lds + Cfg::LDS_FLOATS;
            const int hid = a.fh.hid, fw_ = __builtin_amdgcn_readfirstlane((tid - NT) >> 6), cn = hid / FHW, c0 = fw_ * cn;
            float* patch = fbase;
            float* fw = fbase + hid * 96;
            const float* src = a.fh.x + b * a.fh.xbs;
            constexpr int NPL = 12, NWL = 5;  // per lane: patch items (hid <= 128: 96 * (hid / 16) / 64) and weights (hid / 4 * 9 / 64)
            const int total = 96 * (cn / 4), nwt = cn * 9;
            float4 pv[NPL];
            bool pok[NPL];
            float wreg[NWL];
#pragma unroll
            for (int k = 0; k < NWL; ++k) wreg[k] = a.fh.w[c0 * 9 + min(lane + 64 * k, nwt - 1)];
            if (tid - NT < 60) {
                fh_bias = a.fh.bias[0];
                const int y = ty0 + (tid - NT) / 10 - 1, x = tx0 + (tid - NT) % 10 - 1;
                if (y >= 0 && y < H && x >= 0 && x < W) fh_state = a.fh.coords_in[b * XP + pix_off(a.lay, y, x)];
            }
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                const int e = min(lane + 64 * j, total - 1);
                const int pos = e % 96, q = c0 / 4 + e / 96;
                const int gy = ty0 + pos / 12 - 2, gx = tx0 + pos % 12 - 2;
                pok[j] = gy >= 0 && gy < H && gx >= 0 && gx < W;
                const long off = pok[j] ? pix_off(a.lay, gy, gx) : 0;
                if (a.x_c4) pv[j] = *reinterpret_cast<const float4*>(src + (long)q * 4 * XP + 4 * off);
                else pv[j] = make_float4(src[(long)(4 * q) * XP + off], src[(long)(4 * q + 1) * XP + off], src[(long)(4 * q + 2) * XP + off], src[(long)(4 * q + 3) * XP + off]);
            }
            __builtin_amdgcn_s_barrier();  // (x tile staged) — bare: the loads above stay in flight
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                const int e = lane + 64 * j;
                if (e < total) {
                    float* pp = patch + (c0 + 4 * (e / 96)) * 96 + e % 96;
                    pp[0] = pok[j] ? pv[j].x : 0.f;
                    pp[96] = pok[j] ? pv[j].y : 0.f;
                    pp[192] = pok[j] ? pv[j].z : 0.f;
                    pp[288] = pok[j] ? pv[j].w : 0.f;
                }
            }
#pragma unroll
            for (int k = 0; k < NWL; ++k)
                if (lane + 64 * k < nwt) fw[c0 * 9 + lane + 64 * k] = wreg[k];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the wave's own LDS writes before its own reads
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            {   // the wave's 4 slices x 60 positions: lane = position, the 4 slices side by side; the 36 values and 36 (wave-uniform:
                // broadcast) weights of a channel step are read from LDS together: 8 LDS round trips per lane — under the K loop's
                // ds_read_b128 stream a round trip is long, and 24 of them (a tap row at a time) kept the K loop's barrier waiting for
                // these waves.  (Weights straight from global memory became vector loads queued behind the GEMM's weight stream.)
                const int cps = hid / 16, s0 = c0 / cps;
                const int pos = min(lane, 59), po = (pos / 10) * 12 + pos % 10;
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
                for (int cc = 0; cc < cps; ++cc) {
                    float xv[4][9];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float* pp = patch + ((s0 + k) * cps + cc) * 96 + po;
#pragma unroll
                        for (int t = 0; t < 9; ++t) xv[k][t] = pp[(t / 3) * 12 + t % 3];
                    }
                    float wv[4][9];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float* wk = fw + ((s0 + k) * cps + cc) * 9;
#pragma unroll
                        for (int t = 0; t < 9; ++t) wv[k][t] = wk[t];
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int t = 0; t < 9; ++t) fmac_scalar(acc[k], wv[k][t], xv[k][t]);
                }
                float* part = fbase + hid * 105;
                if (lane < 60)
#pragma unroll
                    for (int k = 0; k < 4; ++k) part[(s0 + k) * 64 + pos] = acc[k];
            }
            __syncthreads();  // (K loop done)
            fh_finish(fbase, tid - NT);
            __syncthreads();
            __syncthreads();
            return;
        }
    }
    // ---- stage the x tile (all CIN channels) and the flow patch
    {
        const float* src = a.x + b * a.xbs;
        if constexpr (SPLIT) {
            // item = (pixel, 4 consecutive channels): one 16-B load (c4) or four 4-B loads (issued above, in front of the folded flow head),
            // split into NS x 4 16-bit pieces = NS 8-B stores
            unsigned char* xsb = reinterpret_cast<unsigned char*>(lds);
#pragma unroll
            for (int i = 0; i < NQ4; ++i) {
                const int e = tid + i * NT;
                const int qd = e >> 5, px = e & 31;
                const bool ok = (ty0 + (px >> 3)) < H && (tx0 + (px & 7)) < W;
                if (e < CIN / 4 * 32) {
                    const float res[4] = {ok ? v4[i].x : 0.f, ok ? v4[i].y : 0.f, ok ? v4[i].z : 0.f, ok ? v4[i].w : 0.f};
                    const int chunk = qd >> 2, sub = qd & 3;
                    constexpr int NP = NS == 0 ? 3 : NS;
                    uint2 pv[NP];
                    split_pieces_n<NP, 4, uint2>(res, pv, xscale);  // round-to-nearest of the running residual (split_arith.h)
#pragma unroll
                    for (int sp = 0; sp < NP; ++sp)
                        *reinterpret_cast<uint2*>(xsb + ((((chunk * NP + sp) * 2 + (sub >> 1)) * 32 + px) * 16 + (sub & 1) * 8)) = pv[sp];
                }
            }
        } else
        if (a.x_c4) {  // one 16-B load per (pixel, 4 channels)
            constexpr int NQ4 = (CIN / 4 * 32 + NT - 1) / NT;
            float4 v4[NQ4];
#pragma unroll
            for (int i = 0; i < NQ4; ++i) {
                const int e = tid + i * NT;
                const int qd = e >> 5, px = e & 31;
                const int y = ty0 + (px >> 3), x = tx0 + (px & 7);
                const bool ok = e < CIN / 4 * 32 && y < H && x < W;
                v4[i] = *reinterpret_cast<const float4*>(src + (ok ? (unsigned)(qd * 4 * (int)XP + 4 * (int)pix_off(a.lay, y, x)) : 0u));
            }
#pragma unroll
            for (int i = 0; i < NQ4; ++i) {
                const int e = tid + i * NT;
                const int qd = e >> 5, px = e & 31;
                const bool ok = (ty0 + (px >> 3)) < H && (tx0 + (px & 7)) < W;
                if (e < CIN / 4 * 32) {
                    xs[(qd * 4 + 0) * 32 + px] = ok ? v4[i].x : 0.f;
                    xs[(qd * 4 + 1) * 32 + px] = ok ? v4[i].y : 0.f;
                    xs[(qd * 4 + 2) * 32 + px] = ok ? v4[i].z : 0.f;
                    xs[(qd * 4 + 3) * 32 + px] = ok ? v4[i].w : 0.f;
                }
            }
        } else {
        constexpr int NLD = (CIN * 32 + NT - 1) / NT;
        float v[NLD];
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = tid + i * NT;
            const int ci = e >> 5, px = e & 31;
            const int y = ty0 + (px >> 3), x = tx0 + (px & 7);
            const bool ok = e < CIN * 32 && y < H && x < W;
            v[i] = src[ok ? (unsigned)(ci * (int)XP + (int)pix_off(a.lay, y, x)) : 0u];
        }
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = tid + i * NT;
            const int px = e & 31;
            const bool ok = (ty0 + (px >> 3)) < H && (tx0 + (px & 7)) < W;
            if (e < CIN * 32) xs[e] = ok ? v[i] : 0.f;
        }
        }
        if (!a.fh.x && tid < 60 * a.fc) {
            const int f = tid / 60, pos = tid % 60;
            const int pr = pos / 10, pc = pos % 10;
            const int y = ty0 + pr - 1, x = tx0 + pc - 1;
            fp[tid] = (y >= 0 && y < H && x >= 0 && x < W) ? a.flow[(b * a.fc + f) * XP + pix_off(a.lay, y, x)] : 0.f;
        }
    }
    __syncthreads();
    NND_MSTAMP(1);

    // ---- K loop: units u = (step, cbi); A of unit u+1 is prefetched during unit u, B of step s+1 during step s
    f32x16 acc[CBW];
#pragma unroll
    for (int i = 0; i < CBW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    if constexpr (SPLIT) {
        // weights in pack_conv_split order: uint4 index (((cb * NCH16 + chunk) * NP + piece) * 64 + lane), lane = k-half * 32 + (co % 32)
        constexpr int NP = NS == 0 ? 3 : NS;
        constexpr int NCH16 = CIN / 16, CPW = NCH16 / 2;  // 16-channel chunks: all / per wave (its K half)
        const uint4* wq4 = reinterpret_cast<const uint4*>(a.wpk);
        const unsigned char* xsb = reinterpret_cast<const unsigned char*>(lds);
        const int c0k = kj * CPW;
        auto a_ptr = [&](int cbi, int ch) {
            int cb = g * CBW + cbi;
            cb = cb < NCB ? cb : NCB - 1;  // padded group member: re-reads a valid block, result discarded
            return wq4 + (size_t)((cb * NCH16 + c0k + ch) * NP) * 64 + lane;
        };
        // a unit (chunk, block) is only 6 MFMAs = 192 cycles: the weight fragments run AD units ahead through a ring of AD + 1
        // register sets.  (AD = 1 measured the same 22.4 us: the 255 workgroups pull 864 KB of weight pieces each = 220 MB per
        // launch through the L2s, ~18 TB/s during the K loop — the stream is throughput-bound, not latency-bound.)
        constexpr int AD = 3, NA = AD + 1, NUNIT = CPW * CBW;
        uint4 ab[NA][NP], bq[2][NP];
        auto load_a = [&](uint4* dst, int u) {
            const uint4* w = a_ptr(u % CBW, u / CBW);
#pragma unroll
            for (int sp = 0; sp < NP; ++sp) dst[sp] = w[sp * 64];
        };
        auto load_b = [&](uint4* dst, int ch) {
#pragma unroll
            for (int sp = 0; sp < NP; ++sp)
                dst[sp] = *reinterpret_cast<const uint4*>(xsb + (((((c0k + ch) * NP + sp) * 2 + h2) * 32 + l31) * 16));
        };
#pragma unroll
        for (int u = 0; u < AD && u < NUNIT; ++u) load_a(ab[u], u);
        load_b(bq[0], 0);
#pragma unroll
        for (int u = 0; u < NUNIT; ++u) {
            const int ch = u / CBW, cbi = u % CBW;
            if (u + AD < NUNIT) load_a(ab[(u + AD) % NA], u + AD);
            if (cbi == 0 && ch + 1 < CPW) load_b(bq[(ch + 1) & 1], ch + 1);
            __builtin_amdgcn_sched_barrier(0);
            split_mfma_step<NP>(ab[u % NA], bq[ch & 1], acc[cbi]);  // x_i * w_j with i + j descending: the small products first
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (NS == 2) {
#pragma unroll
            for (int cbi = 0; cbi < CBW; ++cbi)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[cbi][r] *= oscale;
        }
    } else {
        constexpr int SPW = NST / 2;  // steps per wave
        const int s0 = kj * SPW;
        const float4* wq = reinterpret_cast<const float4*>(a.wpk);
        auto a_base = [&](int cbi) {
            int cb = g * CBW + cbi;
            cb = cb < NCB ? cb : NCB - 1;  // padded group member: re-reads a valid block, result discarded
            return wq + (size_t)cb * (NCHUNK * 16 * 64);
        };
        float4 ab[2][4];
        float bq[2][16];
        auto load_a = [&](float4* dst, int u) {  // u = local unit index
            const int s = s0 + u / CBW, cbi = u % CBW;
            const float4* ws = a_base(cbi) + s * (4 * 64);
    #pragma unroll
            for (int q = 0; q < 4; ++q) dst[q] = ws[(unsigned)(q * 64 + lane)];
        };
        auto load_b = [&](float* dst, int sl) {
            const float* xb = xs + ((s0 + sl) * 32 + h2) * 32 + l31;
    #pragma unroll
            for (int pair = 0; pair < 16; ++pair) dst[pair] = xb[pair * 64];
        };
        load_a(ab[0], 0);
        load_b(bq[0], 0);
    #pragma unroll
        for (int u = 0; u < SPW * CBW; ++u) {
            const int sl = u / CBW, cbi = u % CBW;
            if (u + 1 < SPW * CBW) load_a(ab[(u + 1) & 1], u + 1);
            if (cbi == 0 && sl + 1 < SPW) load_b(bq[(sl + 1) & 1], sl + 1);
            __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
            for (int pair = 0; pair < 16; ++pair) {
                const float4 av = ab[u & 1][pair / 4];
                const float as = (pair % 4 == 0) ? av.x : (pair % 4 == 1) ? av.y : (pair % 4 == 2) ? av.z : av.w;
                acc[cbi] = __builtin_amdgcn_mfma_f32_32x32x2f32(as, bq[sl & 1][pair], acc[cbi], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
}
    NND_MSTAMP(2);
    __syncthreads();  // everyone is done reading xs: the mask tile may overwrite it

    // ---- sum the two K-halves into the LDS mask tile [COUT][33]
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if (kj == pass) {
#pragma unroll
            for (int cbi = 0; cbi < CBW; ++cbi) {
                const int cb = g * CBW + cbi;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int co = cb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h2;
                    if (cb < NCB && co < COUT) {
                        float* p = mt + co * MTS + l31;
                        *p = pass == 0 ? acc[cbi][reg] : *p + acc[cbi][reg];
                    }
                }
            }
        }
        __syncthreads();
    }

    NND_MSTAMP(3);
    // ---- softmax over the 9 neighbours + convex combination; item = (tile row rr, sub-row i) x (col c, sub-col j)
    constexpr int ROWLEN = 8 * RATE;          // contiguous output floats per item row
    constexpr int NITEM = 4 * RATE * ROWLEN;  // outputs per tile
    const long OW = (long)W * RATE;
    for (int e = tid; e < NITEM; e += NT) {
        const int seg = e / ROWLEN, within = e % ROWLEN;
        const int rr = seg / RATE, i = seg % RATE;
        const int c = within / RATE, j = within % RATE;
        const int px = rr * 8 + c;
        const int y = ty0 + rr, x = tx0 + c;
        float m[9];
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int co = k * RATE * RATE + i * RATE + j;
            m[k] = 0.25f * (mt[co * MTS + px] + a.bias[co]);
            mx = fmaxf(mx, m[k]);
        }
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            m[k] = expf(m[k] - mx);
            sum += m[k];
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) m[k] = m[k] / sum;
        for (int f = 0; f < a.fc; ++f) {
            float o = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) o += m[k] * ((float)RATE * fp[f * 60 + (rr + k / 3) * 10 + c + k % 3]);
            if (y < H && x < W) a.out[((long)b * a.fc + f) * HW * RATE * RATE + ((long)y * RATE + i) * OW + (long)x * RATE + j] = o;
        }
    }
#ifdef NND_DBG_STAMPS
    __builtin_amdgcn_s_waitcnt(0);
#endif
    NND_MSTAMP(4);
}

constexpr int MU_FHW = 4;  // flow-head waves of the concurrent fold (fp16x2: 105 VGPRs, 16 waves per CU fit)
template <int RATE, int CIN, int NS, int FHW>
static int launch_mu_fhw(MaskUpArgs a, int B, hipStream_t stream) {
    using Cfg = MaskUpCfg<RATE, CIN>;
    auto kern = mask_upsample_kernel<RATE, CIN, NS, FHW>;
    a.lds_floats = Cfg::LDS_FLOATS;
    const int fh_floats = a.fh.x ? a.fh.hid * (96 + 9) + 16 * 64 : 0;  // the folded flow head's patch, weights, partial sums
    if (FHW > 0) a.lds_floats = Cfg::LDS_FLOATS + fh_floats + 128;      // behind the GEMM's LDS
    const size_t lds = a.lds_floats * sizeof(float);
    if (lds > 64 * 1024) {
        static std::atomic<unsigned> raised{0};
        if (int rc = raise_lds_limit(reinterpret_cast<const void*>(kern), raised)) return rc;
    }
    dim3 grid(a.tiles_x * cdiv(a.H, 4), 1, B), block(Cfg::NT + 64 * FHW);
    hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
    NND_LAUNCH_CHECK();
    return NND_OK;
}
template <int RATE, int CIN, int NS = 0>
static int launch_mu(const MaskUpArgs& a, int B, hipStream_t stream) {
    if (a.fh.x) {
        if constexpr (NS == 2) return launch_mu_fhw<RATE, CIN, NS, MU_FHW>(a, B, stream);
        else NND_REQUIRE(false, "mask_upsample: the folded flow head is built for the fp16x2 arithmetic");
    }
    return launch_mu_fhw<RATE, CIN, NS, 0>(a, B, stream);
}

// the folded flow head exists for this mask layer / hidden size (the caller decides whether it pays: small grids)
bool mask_upsample_fold_supported(const ConvLayer& L, int hid) { return L.arith == 2 && hid % 16 == 0 && hid <= 128; }

bool mask_upsample_supported(int rate, int cin, int flow_channels) {
    return (flow_channels == 1 || flow_channels == 2) && ((rate == 8 && (cin == 256 || cin == 128)) || (rate == 4 && (cin == 256 || cin == 128)));
}

int mask_upsample_launch(const ConvLayer& L, const float* blob, const float* x, int64_t xbs, const float* flow, float* out,
                         int B, int H, int W, int rate, hipStream_t stream, bool tiled, int flow_channels, bool x_c4,
                         const MaskUpFlowHead* fh) {
    NND_REQUIRE(L.KH == 1 && L.KW == 1 && L.CI_T == (L.arith ? 16 : 128) && L.Cout == 9 * rate * rate && (L.arith == 0 || L.arith == 3 || L.arith == 2),
                "mask_upsample: layer shape / packing");
    NND_REQUIRE(mask_upsample_supported(rate, L.Cin, flow_channels), "mask_upsample: rate %d / Cin %d / %d flow channels not built",
                rate, L.Cin, flow_channels);
    MaskUpArgs a;
    a.x = x; a.xbs = xbs; a.wpk = blob + L.w_off; a.bias = blob + L.b_off; a.flow = flow; a.out = out;
    a.H = H; a.W = W; a.tiles_x = cdiv(W, 8); a.fc = flow_channels;
    a.lay = make_lay(H, W, tiled);
    a.x_c4 = (tiled && x_c4) ? 1 : 0;
    a.lds_floats = 0;
    a.fh = MaskUpFlowHead{};
    if (fh && fh->x) {
        NND_REQUIRE(flow_channels == 1 && mask_upsample_fold_supported(L, fh->hid) && fh->w && fh->bias && fh->coords_in && fh->coords_out && fh->flow_out &&
                        fh->delta_out && fh->hx_flow && fh->coords_in != fh->coords_out,
                    "mask_upsample: folded flow head needs one flow channel, hid %% 16 == 0 and distinct old / new state buffers");
        a.fh = *fh;
    }
    if (L.arith == 3) {  // split-bf16 arithmetic: weights in pack_conv_split's order
        if (rate == 8 && L.Cin == 256) return launch_mu<8, 256, 3>(a, B, stream);
        if (rate == 8 && L.Cin == 128) return launch_mu<8, 128, 3>(a, B, stream);
        if (rate == 4 && L.Cin == 256) return launch_mu<4, 256, 3>(a, B, stream);
        return launch_mu<4, 128, 3>(a, B, stream);
    }
    if (L.arith == 2) {  // range-scaled fp16 pieces
        if (calibrating()) {  // record the largest |x| this launch stages (calib.hip)
            const Act xa{const_cast<float*>(x), xbs, L.Cin};
            if (int rc = calib_amax_act(xa, make_lay(H, W, tiled, x_c4), B, H, W, blob + L.tail_off(), stream)) return rc;
        }
        if (rate == 8 && L.Cin == 256) return launch_mu<8, 256, 2>(a, B, stream);
        if (rate == 8 && L.Cin == 128) return launch_mu<8, 128, 2>(a, B, stream);
        if (rate == 4 && L.Cin == 256) return launch_mu<4, 256, 2>(a, B, stream);
        return launch_mu<4, 128, 2>(a, B, stream);
    }
    if (rate == 8 && L.Cin == 256) return launch_mu<8, 256>(a, B, stream);
    if (rate == 8 && L.Cin == 128) return launch_mu<8, 128>(a, B, stream);
    if (rate == 4 && L.Cin == 256) return launch_mu<4, 256>(a, B, stream);
    return launch_mu<4, 128>(a, B, stream);
}

#ifdef NND_DBG_STAMPS
extern "C" int nnd_debug_read_mu_stamps(unsigned long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_mu_stamps), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -1;
}
#endif

}  // namespace nnd
