// RAFT update operator (motion encoder -> SepConvGRU / ConvGRU -> flow head + mask head) and the
// fused per-pair refinement loop, as a fixed sequence of launches of the fp32-MFMA implicit-GEMM
// convolution (conv_mfma.hip) plus three small VALU kernels.
//
// Replaces nndepth/blocks/update_block.py:57-65 (BasicMotionEncoder), :26-36 (FlowHead), :97-112
// (BasicUpdateBlock.forward + mask head), nndepth/blocks/gru.py:22-37 (SepConvGRU), :53-61 (ConvGRU) and
// the loop of nndepth/models/raft_stereo/model.py:126-137 (reference; restated in oracle/torch_ref.py).
//
// torch.cat is never materialised: tensors that the reference concatenates live side by side in one
// workspace buffer (hx = [h | inp | motion | flow], cf = [cor | flo], fm = [flow_head.conv1 | mask.0]) and the
// convolutions write straight into their channel slice.  convz/convr are packed as one 2*hidden-channel
// conv whose epilogue emits z and r*h; convq's epilogue does the GRU blend in place.  The fused loops enqueue every
// kernel of every iteration on the caller's stream (enqueue_refine).
#include "common.h"
#include "layout.h"

#include <cstdlib>
#include <cstring>
#include <vector>

namespace nnd {

// ------------------------------------------------------------------------------------ plan
// C_*X: GRU convs of the fused loop — input = [h | motion+flow] only; the contribution of the context features
// `inp` (constant over the iterations of a pair) is precomputed once per pair by the C_*C convs and enters as a
// per-pixel bias map.  C_ZR1..C_Q2 are the full convs used by the single-call API (nnd_update_block_forward).
enum ConvId { C_C1 = 0, C_C2, C_F2, C_CV, C_ZR1X, C_Q1X, C_ZR2X, C_Q2X, C_FHM, C_M2, C_LOOP_COUNT,
              C_ZR1 = C_LOOP_COUNT, C_Q1, C_ZR2, C_Q2, C_ZR1C, C_Q1C, C_ZR2C, C_Q2C, C_COUNT };
static const char* kConvNames[C_COUNT] = {"encoder.convc1", "encoder.convc2", "encoder.convf2", "encoder.conv",
                                          "gru.convz1+convr1[h,motion]", "gru.convq1[rh,motion]", "gru.convz2+convr2[h,motion]",
                                          "gru.convq2[rh,motion]", "flow_head.conv1+mask.0", "mask.2",
                                          "gru.convz1+convr1", "gru.convq1", "gru.convz2+convr2", "gru.convq2",
                                          "gru.convz1+convr1[inp]", "gru.convq1[inp]", "gru.convz2+convr2[inp]", "gru.convq2[inp]"};

struct Plan {
    nnd_update_block_desc d;
    bool sep;
    ConvLayer L[C_COUNT];
    int64_t f1_w, f1_b;  // raw encoder.convf1 weights [128][fc][49] + bias [128]
    int64_t f1_wt;       // the same weights tap-major [fc*49][128] for the fused flow-branch kernel
    int64_t fc2_w, fc2_b;  // raw flow_head.conv2 weights [fc][hid][9] + bias [fc] (VALU kernel)
    int64_t total;
};

// arith != 0: the layer runs on conv_split.hip (fp32 operands as split bf16 pieces), if that kernel is built for its shape
static ConvLayer mk(int KH, int KW, int Cin, int Cout, int64_t* off, int arith = 0) {
    ConvLayer l;
    if (arith != 0 && !conv_split_supported(KH, KW, Cin, 1, arith)) arith = 0;
    const int CI_T = arith ? 16 : conv_ci_t(KH, KW, Cin, 1, Cout);
    l.arith = arith;
    l.KH = KH; l.KW = KW; l.Cin = Cin; l.Cout = Cout; l.CI_T = CI_T;
    l.nchunks = cdiv(Cin, CI_T);
    l.ncb = cdiv(Cout, 32);
    l.w_off = *off;
    *off += l.w_floats();
    l.b_off = *off;
    *off += l.b_floats();
    return l;
}

static int make_plan(const nnd_update_block_desc* d, Plan* p) {
    NND_REQUIRE(d, "update_block: null descriptor");
    NND_REQUIRE(d->struct_size == (int32_t)sizeof(nnd_update_block_desc), "update_block: descriptor of %d bytes, this library expects %d (struct_size)",
                d->struct_size, (int)sizeof(nnd_update_block_desc));
    NND_REQUIRE((d->flags & ~NND_FLAG_CALIBRATE) == 0, "update_block: unknown flags 0x%x", d->flags);
    const int hid = d->hidden_dim, ctx = d->context_dim, cp = d->cor_planes, fc = d->flow_channels, mc = d->mask_channels;
    NND_REQUIRE(hid > 0 && hid % 32 == 0, "update_block: hidden_dim %d must be a positive multiple of 32", hid);
    NND_REQUIRE(ctx > 0 && ctx % 8 == 0, "update_block: context_dim %d must be a positive multiple of 8", ctx);
    NND_REQUIRE(cp > 0 && (fc == 1 || fc == 2) && mc > 0 && mc % 9 == 0, "update_block: bad cor_planes/flow_channels/mask_channels");
    NND_REQUIRE(d->gru_kind == 0 || d->gru_kind == 1, "update_block: gru_kind must be 0 (sep_conv) or 1 (conv_gru)");
    NND_REQUIRE(d->arithmetic == 0 || d->arithmetic == 3 || d->arithmetic == 2,
                "update_block: arithmetic must be 0 (fp32 MFMA), 3 (bf16x3 split) or 2 (fp16x2 split)");
    // arithmetic == 3: every MFMA conv whose shape conv_split.hip builds takes the split-bf16 kernel (mask.2 inside the fused
    // mask + upsample kernel; convf2 together with convf1 in flow_branch_kernel), except convc1 (its weights are consumed by the
    // fused lookup kernels in the fp32 packing); desc.split_layers (diagnostic) restricts it to a subset, bit = ConvId (e.g. 2 =
    // encoder.convc2 only) — part of the descriptor, so the blob layout is a function of the descriptor alone
    const unsigned split_mask = d->split_layers ? (unsigned)d->split_layers : ~0u;
    auto ar = [&](int id) { return (d->arithmetic != 0 && ((split_mask >> id) & 1u)) ? d->arithmetic : 0; };
    p->d = *d;
    p->sep = d->gru_kind == 0;
    int64_t off = 0;
    const int gin = hid + ctx + hid;
    // convc1: split only where its K is whole 16-channel chunks (IGEV's 576 correlation planes: a 9.6-GFLOP GEMM per iteration at
    // 136x240, fused with the interleaved lookup in igev_lookup_convc1_il_split_kernel); RAFT-Stereo's and CREStereo's 36 planes
    // stay fp32 inside the fused lookup kernels (mk() falls back: 36 % 16 != 0)
    p->L[C_C1] = mk(1, 1, cp, 256, &off, ar(C_C1));
    p->L[C_C2] = mk(3, 3, 256, 192, &off, ar(C_C2));
    p->f1_w = off; off += (int64_t)128 * fc * 49;
    p->f1_b = off; off += 128;
    p->f1_wt = off; off += (int64_t)128 * fc * 49;
    p->L[C_F2] = mk(3, 3, 128, 64, &off, ar(C_F2));  // split arithmetic: ONE launch with convf1 (run_flow_branch)
    p->L[C_CV] = mk(3, 3, 256, hid - fc, &off, ar(C_CV));
    for (int pass = 0; pass < (p->sep ? 2 : 1); ++pass) {
        const int kh = p->sep ? (pass == 0 ? 1 : 5) : 3, kw = p->sep ? (pass == 0 ? 5 : 1) : 3;
        const int zr = pass == 0 ? C_ZR1 : C_ZR2, q = pass == 0 ? C_Q1 : C_Q2;
        p->L[zr] = mk(kh, kw, gin, 2 * hid, &off, ar(zr));
        p->L[q] = mk(kh, kw, gin, hid, &off, ar(q));
        p->L[zr - C_ZR1 + C_ZR1X] = mk(kh, kw, 2 * hid, 2 * hid, &off, ar(zr - C_ZR1 + C_ZR1X));
        p->L[q - C_ZR1 + C_ZR1X] = mk(kh, kw, 2 * hid, hid, &off, ar(q - C_ZR1 + C_ZR1X));
        p->L[zr - C_ZR1 + C_ZR1C] = mk(kh, kw, ctx, 2 * hid, &off, ar(zr - C_ZR1 + C_ZR1C));
        p->L[q - C_ZR1 + C_ZR1C] = mk(kh, kw, ctx, hid, &off, ar(q - C_ZR1 + C_ZR1C));
    }
    if (!p->sep)
        for (int base : {(int)C_ZR1X, (int)C_ZR1, (int)C_ZR1C}) {
            p->L[base + 2] = p->L[base];
            p->L[base + 3] = p->L[base + 1];
        }
    p->L[C_FHM] = mk(3, 3, hid, 3 * hid, &off, ar(C_FHM));  // flow_head.conv1 (hid) and mask.0 (2*hid): same input h, both ReLU -> one conv
    p->fc2_w = off; off += (int64_t)fc * hid * 9;
    p->fc2_b = off; off += 4;  // keep 16-B alignment of what follows
    p->L[C_M2] = mk(1, 1, 2 * hid, mc, &off, ar(C_M2));  // consumed by the fused mask + upsample kernel in the same arithmetic
    p->total = off;
    return NND_OK;
}

// ------------------------------------------------------------------------------- workspace
struct Bufs {
    float *c1, *cf, *f1, *hx, *z, *rh, *fm, *corr, *mask, *delta, *coords, *flow, *ctxb;
    int64_t total;
};

static void carve(const Plan& p, int B, int H, int W, float* base, Bufs* b) {
    const int64_t n = (int64_t)B * tiled_plane(H, W);  // every workspace tensor is tile-major (layout.h)
    const int hid = p.d.hidden_dim, ctx = p.d.context_dim, fc = p.d.flow_channels;
    int64_t off = 0;
    auto take = [&](int C) {
        float* ptr = base ? base + off : nullptr;
        off += ((int64_t)C * n + 63) / 64 * 64;  // keep every buffer 256-B aligned
        return ptr;
    };
    b->c1 = take(256);
    b->cf = take(256);
    b->f1 = take(128);
    b->hx = take(2 * hid + ctx);
    b->z = take(hid);
    b->rh = take(hid);
    b->fm = take(3 * hid);
    b->corr = take(p.d.cor_planes);
    b->mask = take(p.d.mask_channels);
    b->delta = take(fc);
    b->coords = take(1);
    b->flow = take(fc);
    b->ctxb = take(6 * hid);   // context terms of the GRU convs: [zr1 2h | q1 h | zr2 2h | q2 h]
    b->total = off;
}

// ------------------------------------------------------------------------- small kernels
// encoder.convf1: 7x7, Cin = FC (1 or 2) -> 128, ReLU.  K = 49*FC is too shallow for the MFMA path;
// each thread keeps its 49*FC-tap neighbourhood in registers and the (wave-uniform) weights come in
// through the scalar cache.  grid (regions of 8x32 px, 128/16 channel groups, B), 256 threads; a region is walked as
// 8 sub-tiles of 4x8 pixels so that a wave's stores are two full 128-B lines of the tile-major layout.
template <int FC>
__global__ void __launch_bounds__(256) convf1_kernel(const float* __restrict__ flow, long fbs, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ out, long obs,
                                                     int H, int W, int tiles_x, Lay lay) {
    __shared__ float patch[FC][14][40];
    // the workgroup's 16 x FC*49 weights, read back as LDS broadcasts (round 1 read them through the scalar cache as SGPR
    // operands: 16 dependent s_load batches per thread, each a scalar-cache miss)
    __shared__ __attribute__((aligned(16))) float wl[16 * (FC * 49 + 3)];
    constexpr int WS = FC * 49 + 3;  // row stride: multiple of 4 floats
    const int tid = threadIdx.x;
    const int tx0 = (blockIdx.x % tiles_x) * 32, ty0 = (blockIdx.x / tiles_x) * 8;
    const int b = blockIdx.z, co0 = blockIdx.y * 16;
    const long HW = lay.plane;
    for (int e = tid; e < FC * 14 * 38; e += 256) {
        int c = e / (14 * 38), rem = e % (14 * 38);
        int pr = rem / 38, pc = rem % 38;
        int gy = ty0 + pr - 3, gx = tx0 + pc - 3;
        patch[c][pr][pc] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? flow[b * fbs + c * HW + pix_off(lay, gy, gx)] : 0.f;
    }
    for (int e = tid; e < 16 * FC * 49; e += 256) wl[(e / (FC * 49)) * WS + e % (FC * 49)] = w[(long)co0 * (FC * 49) + e];
    __syncthreads();
    const int st = tid >> 5, within = tid & 31;  // sub-tile (2 rows x 4 cols of them), pixel inside it
    const int ty = (st >> 2) * 4 + (within >> 3), tx = (st & 3) * 8 + (within & 7);
    float v[FC * 49];
#pragma unroll
    for (int c = 0; c < FC; ++c)
#pragma unroll
        for (int dy = 0; dy < 7; ++dy)
#pragma unroll
            for (int dx = 0; dx < 7; ++dx) v[c * 49 + dy * 7 + dx] = patch[c][ty + dy][tx + dx];
    const int y = ty0 + ty, x = tx0 + tx;
    const bool ok = y < H && x < W;
#pragma unroll 4
    for (int cc = 0; cc < 16; ++cc) {
        const float* wc = wl + cc * WS;  // same address in every lane: broadcast reads
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < FC * 49; ++i) acc = fmaf(wc[i], v[i], acc);
        acc += bias[co0 + cc];
        if (ok) out[b * obs + (co0 + cc) * HW + pix_off(lay, y, x)] = fmaxf(acc, 0.f);
    }
}

// coords = x (+ disp_init); flow = coords - x
// absolute != 0 (IGEV, Q5): the update block and the upsample receive the coordinate itself, not coords - x
// hx_flow points at the flow channel of the GRU input buffer, whose pixels are hx_pm floats apart (4 in the 4-channel-
// interleaved layout of that buffer, 1 otherwise); coords / flow themselves are planar tile-major (`lay`)
__global__ void init_coords_kernel(float* __restrict__ coords, float* __restrict__ flow, float* __restrict__ hx_flow,
                                   long hx_bs, int hx_pm, const float* __restrict__ disp_init, int B, int H, int W, int absolute, Lay lay) {
    const long HW = (long)H * W;
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * HW) return;
    const int b = (int)(idx / HW);
    const long pix = idx % HW;
    const int x = (int)(pix % W);
    const long po = pix_off(lay, (int)(pix / W), x);
    float c = (float)x + (disp_init ? disp_init[idx] : 0.f);  // disp_init is a C-ABI tensor: NCHW
    float f = absolute ? c : c - (float)x;
    coords[b * lay.plane + po] = c;
    flow[b * lay.plane + po] = f;
    hx_flow[b * hx_bs + po * hx_pm] = f;
}

// NCHW (C-ABI) <-> tile-major (workspace) copies of a C-channel tensor; dst/src batch strides in floats
// `c0`: first channel of the slice inside the workspace tensor that `dst` / `src` points at (its channel 0); src == nullptr
// writes zeros
__global__ void to_tiled_kernel(const float* __restrict__ src, float* __restrict__ dst, long dbs, int c0, int B, int C, int H, int W, Lay lay) {
    const long HW = (long)H * W;
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * C * HW) return;
    const long pix = idx % HW;
    const int c = (int)((idx / HW) % C), b = (int)(idx / (HW * C));
    dst[b * dbs + chan_off(lay, c0 + c) + pix_off(lay, (int)(pix / W), (int)(pix % W))] = src ? src[idx] : 0.f;
}
__global__ void from_tiled_kernel(const float* __restrict__ src, long sbs, int c0, float* __restrict__ dst, int B, int C, int H, int W, Lay lay) {
    const long HW = (long)H * W;
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * C * HW) return;
    const long pix = idx % HW;
    const int c = (int)((idx / HW) % C), b = (int)(idx / (HW * C));
    dst[idx] = src[b * sbs + chan_off(lay, c0 + c) + pix_off(lay, (int)(pix / W), (int)(pix % W))];
}

// flow_head.conv2: 3x3, hid -> FC (1 or 2) outputs.  One output channel wastes 31/32 of an MFMA tile, so this is a
// VALU kernel: a workgroup stages the 6x10 halo patch of all `hid` channels of a 4x8 pixel tile in LDS, thread
// (slice, px) accumulates hid/8 channels x 9 taps, the 8 slices are summed through LDS.  With `advance` the
// recurrence update of nndepth/models/raft_stereo/model.py:134-135 is fused in: coords += delta; flow = coords - x
// (or the coordinate itself for IGEV, `absolute`), mirrored into the GRU input buffer.
template <int FC>
__global__ void __launch_bounds__(512) flow_head2_kernel(const float* __restrict__ x, long xbs, int hid, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ delta,
                                                         float* __restrict__ coords, float* __restrict__ flow,
                                                         float* __restrict__ hx_flow, long hx_bs, int hx_cs, int hx_pm, int x_c4, int H,
                                                         int W, int tiles_x, int advance, int absolute, Lay lay) {
    // One 4x8 pixel tile per workgroup, 512 threads.  Staging: thread = (patch position, channel group of 8), its
    // loads are independent and issued back to back (clamped addresses, zero-filled by select).  Compute: thread =
    // (pixel, slice of hid/16 channels); the 16 partial sums per pixel meet in LDS.
    extern __shared__ float sm[];
    float* patch = sm;                  // [hid][64] (60 used)
    float* wl = sm + hid * 64;          // [FC][hid][9]
    float* part = wl + FC * hid * 9;    // [16][FC][32]
    const int tid = threadIdx.x;
    const int tx0 = (blockIdx.x % tiles_x) * 8, ty0 = (blockIdx.x / tiles_x) * 4, b = blockIdx.z;
    const long HW = lay.plane;
    const float* src = x + b * xbs;
    // the finishing threads (tid < 32*FC) request what their last step reads — bias and the state they advance — now, not
    // behind the reduction at the end of a 7-us kernel
    float tail_bias = 0.f, tail_state = 0.f;
    if (tid < 32 * FC) {
        const int f = tid >> 5, p2 = tid & 31;
        const int y = ty0 + (p2 >> 3), xx = tx0 + (p2 & 7);
        tail_bias = bias[f];
        if (y < H && xx < W) {
            const long pix = pix_off(lay, y, xx);
            if (advance == 2) tail_state = flow[(b * FC + f) * HW + pix];
            else if (advance) tail_state = coords[b * HW + pix];
        }
    }
    {
        const int pos = tid & 63, cg = tid >> 6;  // 8 channel groups
        const int gy = ty0 + pos / 10 - 1, gx = tx0 + pos % 10 - 1;
        const bool ok = pos < 60 && gy >= 0 && gy < H && gx >= 0 && gx < W;
        const long off = ok ? pix_off(lay, gy, gx) : 0;
        if (x_c4) {  // x keeps 4 channels interleaved: one 16-B load per channel quad
            for (int q0 = cg; q0 < hid / 4; q0 += 32) {  // 4 loads in flight per thread
                float4 v4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v4[j] = *reinterpret_cast<const float4*>(src + (long)min(q0 + 8 * j, hid / 4 - 1) * 4 * HW + 4 * off);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (q0 + 8 * j < hid / 4) {
                        float* pp = patch + (q0 + 8 * j) * 4 * 64 + pos;
                        pp[0] = ok ? v4[j].x : 0.f;
                        pp[64] = ok ? v4[j].y : 0.f;
                        pp[128] = ok ? v4[j].z : 0.f;
                        pp[192] = ok ? v4[j].w : 0.f;
                    }
            }
        } else
        for (int c0 = cg; c0 < hid; c0 += 64) {  // 8 loads in flight per thread
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = src[(long)min(c0 + 8 * j, hid - 1) * HW + off];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (c0 + 8 * j < hid) patch[(c0 + 8 * j) * 64 + pos] = ok ? v[j] : 0.f;
        }
    }
    for (int e = tid; e < FC * hid * 9; e += 512) wl[e] = w[e];
    __syncthreads();
    const int px = tid & 31, slice = tid >> 5;  // 16 slices
    const int r = px >> 3, c = px & 7;
    const int cps = hid / 16;
    float acc[FC];
#pragma unroll
    for (int f = 0; f < FC; ++f) acc[f] = 0.f;
    for (int ci = slice * cps; ci < (slice + 1) * cps; ++ci) {
        const float* pp = patch + ci * 64 + r * 10 + c;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float v = pp[(t / 3) * 10 + t % 3];
#pragma unroll
            for (int f = 0; f < FC; ++f) fmac_scalar(acc[f], wl[(f * hid + ci) * 9 + t], v);  // never a packed FMA (common.h)
        }
    }
#pragma unroll
    for (int f = 0; f < FC; ++f) part[(slice * FC + f) * 32 + px] = acc[f];
    __syncthreads();
    if (tid < 32 * FC) {
        const int f = tid >> 5, p2 = tid & 31;
        float sum = 0.f;
#pragma unroll
        for (int sl = 0; sl < 16; ++sl) sum += part[(sl * FC + f) * 32 + p2];
        sum += tail_bias;
        const int y = ty0 + (p2 >> 3), xx = tx0 + (p2 & 7);
        if (y < H && xx < W) {
            const long pix = pix_off(lay, y, xx);
            delta[(b * FC + f) * HW + pix] = sum;
            if (advance == 2) {  // CREStereo: the state is the flow itself (cre_stereo/model.py:281), any FC
                const float fl = tail_state + sum;
                flow[(b * FC + f) * HW + pix] = fl;
                hx_flow[b * hx_bs + f * (long)hx_cs + pix * hx_pm] = fl;  // hx_cs: floats between the flow channels of the GRU input
            } else if (advance) {  // FC == 1 on this path
                const float cnew = tail_state + sum;
                const float fl = absolute ? cnew : cnew - (float)xx;
                coords[b * HW + pix] = cnew;
                flow[b * HW + pix] = fl;
                hx_flow[b * hx_bs + pix * hx_pm] = fl;
            }
        }
    }
}

// ------------------------------------------------------------------------------ sequencing
static Act act(float* p, int64_t bs, int C) { return Act{p, bs, C}; }

// 4-channel-interleaved layout of the conv-only workspace tensors (layout.h); NND_NO_C4 (diagnostic) keeps them planar
static bool ws_c4() { return !switches().no_c4; }
// float offset of channel c / floats between consecutive pixels of a workspace tensor with plane size n
static int64_t ws_chan(int c, int64_t n) { return ws_c4() ? (int64_t)(c / 4) * 4 * n + c % 4 : (int64_t)c * n; }
static int ws_pm() { return ws_c4() ? 4 : 1; }

// IO of conv `id` given the workspace; `corr` / `flow_src` are the external inputs.
static ConvIO conv_io(const Plan& p, const Bufs& w, int id, Act corr, int64_t n /*plane*/, float* mask_dst, float* delta_dst) {
    const int hid = p.d.hidden_dim, ctx = p.d.context_dim, fc = p.d.flow_channels;
    const int hxC = 2 * hid + ctx;
    ConvIO io{};
    switch (id) {
        case C_C1: io.src0 = corr; io.out0 = act(w.c1, 256 * n, 256); break;
        case C_C2: io.src0 = act(w.c1, 256 * n, 256); io.out0 = act(w.cf, 256 * n, 192); break;
        case C_F2: io.src0 = act(w.f1, 128 * n, 128); io.out0 = act(w.cf + 192 * n, 256 * n, 64); break;
        case C_CV: io.src0 = act(w.cf, 256 * n, 256); io.out0 = act(w.hx + (hid + ctx) * n, hxC * n, hid - fc); break;
        case C_ZR1:
        case C_ZR2:
            io.src0 = act(w.hx, hxC * n, hxC);
            io.out0 = act(w.z, hid * n, hid);
            io.out1 = act(w.rh, hid * n, hid);
            io.aux0 = act(w.hx, hxC * n, hid);
            io.hidden = hid;
            break;
        case C_Q1:
        case C_Q2:
            io.src0 = act(w.rh, hid * n, hid);
            io.src1 = act(w.hx + hid * n, hxC * n, ctx + hid);
            io.out0 = act(w.hx, hxC * n, hid);
            io.aux0 = act(w.hx, hxC * n, hid);
            io.aux1 = act(w.z, hid * n, hid);
            break;
        case C_ZR1X:
        case C_ZR2X: {  // [h | motion+flow] + context bias map
            io.src0 = act(w.hx, hxC * n, hid);
            io.src1 = act(w.hx + (hid + ctx) * n, hxC * n, hid);
            io.out0 = act(w.z, hid * n, hid);
            io.out1 = act(w.rh, hid * n, hid);
            io.aux0 = act(w.hx, hxC * n, hid);
            io.bmap = act(w.ctxb + (id == C_ZR1X ? 0 : 3 * hid) * n, 6 * hid * n, 2 * hid);
            io.hidden = hid;
            break;
        }
        case C_Q1X:
        case C_Q2X:
            io.src0 = act(w.rh, hid * n, hid);
            io.src1 = act(w.hx + (hid + ctx) * n, hxC * n, hid);
            io.out0 = act(w.hx, hxC * n, hid);
            io.aux0 = act(w.hx, hxC * n, hid);
            io.aux1 = act(w.z, hid * n, hid);
            io.bmap = act(w.ctxb + (id == C_Q1X ? 2 * hid : 5 * hid) * n, 6 * hid * n, hid);
            break;
        case C_ZR1C:
        case C_ZR2C:  // context term: conv over `inp` only (+ the conv's bias), linear
            io.src0 = act(w.hx + hid * n, hxC * n, ctx);
            io.out0 = act(w.ctxb + (id == C_ZR1C ? 0 : 3 * hid) * n, 6 * hid * n, 2 * hid);
            break;
        case C_Q1C:
        case C_Q2C:
            io.src0 = act(w.hx + hid * n, hxC * n, ctx);
            io.out0 = act(w.ctxb + (id == C_Q1C ? 2 * hid : 5 * hid) * n, 6 * hid * n, hid);
            break;
        case C_FHM: io.src0 = act(w.hx, hxC * n, hid); io.out0 = act(w.fm, 3 * hid * n, 3 * hid); break;
        case C_M2:
            io.src0 = act(w.fm + hid * n, 3 * hid * n, 2 * hid);
            io.out0 = act(mask_dst, p.d.mask_channels * n, p.d.mask_channels);
            io.scale = 0.25f;
            break;
    }
    io.src_tiled = io.dst_tiled = true;  // workspace tensors are tile-major (layout.h)
    // ... and c1, cf, hx, z, rh, fm, ctxb additionally keep 4 channels interleaved (ws_c4): the tensors that the MFMA convs,
    // the fused lookup + convc1 kernels, flow_head.conv2 and the fused mask / upsample kernel produce and consume.
    // f1 (convf1), corr, mask, flow, coords stay planar.
    if (ws_c4()) {
        switch (id) {
            case C_C1: io.dst_c4 = true; break;                                  // corr (planar) -> c1
            case C_C2: io.src_c4 = io.dst_c4 = true; break;                      // c1 -> cf
            case C_F2: io.dst_c4 = true; break;                                  // f1 (planar) -> cf
            case C_CV: io.src_c4 = io.dst_c4 = true; break;                      // cf -> hx[motion]
            case C_FHM: io.src_c4 = io.dst_c4 = true; break;                     // hx[h] -> fm
            case C_M2: io.src_c4 = true; break;                                  // fm -> mask (planar; unfused path only)
            default: io.src_c4 = io.dst_c4 = true; break;                        // GRU convs and their context terms
        }
    }
    return io;
}

static int conv_epi(int id) {
    switch (id) {
        case C_ZR1: case C_ZR2: case C_ZR1X: case C_ZR2X: return EPI_GRU_ZR;
        case C_Q1: case C_Q2: case C_Q1X: case C_Q2X: return EPI_GRU_Q;
        case C_ZR1C: case C_ZR2C: case C_Q1C: case C_Q2C: return EPI_LINEAR;
        case C_M2: return EPI_SCALE;
        default: return EPI_RELU;
    }
}

// NND_DEBUG_SYNC=1: synchronise after every launch and name it on stderr (fault localisation only).
// NND_DEBUG_LDS_POISON=<32-bit pattern> (diagnostic): after every launch of the update block a kernel that fills all of a CU's
// LDS with the pattern runs on every CU, so a kernel that reads LDS it did not write itself computes with the pattern
// instead of what the previous workgroup on that CU happened to leave (scripts/poison_lds.py)
__global__ void __launch_bounds__(256) lds_poison_kernel(unsigned pattern) {
    extern __shared__ unsigned poison_lds[];
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 256) poison_lds[i] = pattern;
    __syncthreads();
    if (poison_lds[(threadIdx.x * 97) % (160 * 1024 / 4)] != pattern) __builtin_trap();
}

static int debug_sync(const char* what, hipStream_t s) {
    if (switches().lds_poison_on) {
        static std::atomic<unsigned> raised{0};
        if (int rc = raise_lds_limit(reinterpret_cast<const void*>(lds_poison_kernel), raised)) return rc;
        hipLaunchKernelGGL(lds_poison_kernel, dim3(2048), dim3(256), 160 * 1024, s, switches().lds_poison);
    }
    if (!switches().debug_sync) return NND_OK;
    fprintf(stderr, "[nnd] %s ...", what);
    fflush(stderr);
    NND_HIP_CHECK(hipStreamSynchronize(s));
    fprintf(stderr, " ok\n");
    fflush(stderr);
    return NND_OK;
}

static int run_conv(const Plan& p, const float* blob, const Bufs& w, int id, Act corr, float* mask_dst, float* delta_dst,
                    int B, int H, int W, hipStream_t s) {
    ConvIO io = conv_io(p, w, id, corr, tiled_plane(H, W), mask_dst, delta_dst);
    int rc = launch_conv(p.L[id], blob, io, conv_epi(id), B, H, W, s);
    if (rc != NND_OK) return rc;
    return debug_sync(kConvNames[id], s);
}

static int run_convf1(const Plan& p, const float* blob, const float* flow, int64_t fbs, float* out, int B, int H, int W,
                      hipStream_t s) {
    const int tiles_x = cdiv(W, 32), tiles_y = cdiv(H, 8);
    dim3 grid(tiles_x * tiles_y, 8, B), block(256);
    const Lay lay = make_lay(H, W, true);
    const int64_t n = lay.plane;
    if (p.d.flow_channels == 1)
        hipLaunchKernelGGL(convf1_kernel<1>, grid, block, 0, s, flow, (long)fbs, blob + p.f1_w, blob + p.f1_b, out, (long)(128 * n), H, W, tiles_x, lay);
    else
        hipLaunchKernelGGL(convf1_kernel<2>, grid, block, 0, s, flow, (long)fbs, blob + p.f1_w, blob + p.f1_b, out, (long)(128 * n), H, W, tiles_x, lay);
    NND_LAUNCH_CHECK();
    return debug_sync("encoder.convf1", s);
}

// advance: 0 = delta only, 1 = coords += delta and flow = coords (- x) (fc == 1), 2 = flow += delta (any fc)
static int run_fc2(const Plan& p, const float* blob, const Bufs& w, float* delta_dst, int advance, bool absolute, int B, int H,
                   int W, hipStream_t s) {
    const int hid = p.d.hidden_dim, ctx = p.d.context_dim, fc = p.d.flow_channels, hxC = 2 * hid + ctx;
    const Lay lay = make_lay(H, W, true);
    const int64_t n = lay.plane;
    const int tiles_x = cdiv(W, 8);
    dim3 grid(tiles_x * cdiv(H, 4), 1, B), block(512);
    const size_t lds = (size_t)(hid * 64 + fc * hid * 9 + 16 * fc * 32) * sizeof(float);
    NND_REQUIRE(hid % 16 == 0 && lds <= 64 * 1024, "flow_head.conv2: hidden_dim %d not supported", hid);
    NND_REQUIRE(advance != 1 || fc == 1, "flow_head.conv2: the coordinate advance needs flow_channels == 1");
    float* hx_flow = w.hx + ws_chan(hxC - fc, n);
    const int hx_cs = (int)(ws_chan(hxC - fc + 1, n) - ws_chan(hxC - fc, n));  // floats between the flow channels inside hx
    if (fc == 1)
        hipLaunchKernelGGL(flow_head2_kernel<1>, grid, block, lds, s, w.fm, (long)(3 * hid * n), hid, blob + p.fc2_w, blob + p.fc2_b,
                           delta_dst, w.coords, w.flow, hx_flow, (long)(hxC * n), hx_cs, ws_pm(), ws_c4() ? 1 : 0, H, W, tiles_x, advance, absolute ? 1 : 0, lay);
    else
        hipLaunchKernelGGL(flow_head2_kernel<2>, grid, block, lds, s, w.fm, (long)(3 * hid * n), hid, blob + p.fc2_w, blob + p.fc2_b,
                           delta_dst, w.coords, w.flow, hx_flow, (long)(hxC * n), hx_cs, ws_pm(), ws_c4() ? 1 : 0, H, W, tiles_x, advance == 2 ? 2 : 0, 0, lay);
    NND_LAUNCH_CHECK();
    return debug_sync("flow_head.conv2", s);
}

// Calibration of the fused flow branch (calib.hip): what convf2 stages there is relu(convf1(flow)), which the fused kernel keeps in
// LDS — under NND_FLAG_CALIBRATE the stand-alone convf1 kernel writes it to w.f1 (unused otherwise on this path) and its largest
// value is recorded in convf2's slot.
static int calib_flow_branch(const Plan& p, const float* blob, const Bufs& w, const float* flow, int B, int H, int W, hipStream_t s) {
    if (!calibrating() || p.L[C_F2].arith != 2) return NND_OK;
    const int64_t n = tiled_plane(H, W);
    int rc = run_convf1(p, blob, flow, (int64_t)p.d.flow_channels * n, w.f1, B, H, W, s);
    if (rc != NND_OK) return rc;
    return calib_amax_act(Act{w.f1, 128 * n, 128}, make_lay(H, W, true), B, H, W, blob + p.L[C_F2].tail_off(), s);
}

// Flow branch of the motion encoder: convf1 (7x7 on the flow) -> convf2 (3x3) -> cf[192:256].  Split arithmetic: one launch
// (conv_split.hip: flow_branch_kernel), the 128-channel intermediate never reaches HBM; exact arithmetic, or
// NND_NO_FUSED_FLOW_BRANCH (which the parity test toggles): convf1_kernel into w.f1, then the conv.
static int run_flow_branch(const Plan& p, const float* blob, const Bufs& w, const float* flow, Act corr, int B, int H, int W,
                           hipStream_t s) {
    const int fc = p.d.flow_channels;
    const int64_t n = tiled_plane(H, W);
    if (flow_branch_supported(p.L[C_F2], fc) && !switches().no_fused_flow_branch) {
        if (int rc0 = calib_flow_branch(p, blob, w, flow, B, H, W, s)) return rc0;
        const ConvIO io = conv_io(p, w, C_F2, corr, n, nullptr, nullptr);
        int rc = launch_flow_branch(p.L[C_F2], blob, blob + p.f1_wt, blob + p.f1_b, flow, (int64_t)fc * n, fc, io, B, H, W, s);
        if (rc != NND_OK) return rc;
        return debug_sync("encoder.convf1+convf2", s);
    }
    int rc = run_convf1(p, blob, flow, (int64_t)fc * n, w.f1, B, H, W, s);
    if (rc != NND_OK) return rc;
    return run_conv(p, blob, w, C_F2, corr, nullptr, nullptr, B, H, W, s);
}

// One application of the update block on workspace state: expects h/inp/flow already in w.hx,
// `flow` = (B,fc,H,W) dense.  Writes new h into w.hx[0:hid], delta, and (optionally) the mask.  Single stream.
static int run_update(const Plan& p, const float* blob, const Bufs& w, Act corr, const float* flow, float* mask_dst,
                      float* delta_dst, int B, int H, int W, hipStream_t s) {
    int rc;
#define NND_TRY(x)                    \
    do {                              \
        if ((rc = (x)) != NND_OK) return rc; \
    } while (0)
    NND_TRY(run_conv(p, blob, w, C_C1, corr, nullptr, nullptr, B, H, W, s));
    NND_TRY(run_conv(p, blob, w, C_C2, corr, nullptr, nullptr, B, H, W, s));
    NND_TRY(run_flow_branch(p, blob, w, flow, corr, B, H, W, s));
    NND_TRY(run_conv(p, blob, w, C_CV, corr, nullptr, nullptr, B, H, W, s));
    NND_TRY(run_conv(p, blob, w, C_ZR1, corr, nullptr, nullptr, B, H, W, s));
    NND_TRY(run_conv(p, blob, w, C_Q1, corr, nullptr, nullptr, B, H, W, s));
    if (p.sep) {
        NND_TRY(run_conv(p, blob, w, C_ZR2, corr, nullptr, nullptr, B, H, W, s));
        NND_TRY(run_conv(p, blob, w, C_Q2, corr, nullptr, nullptr, B, H, W, s));
    }
    NND_TRY(run_conv(p, blob, w, C_FHM, corr, nullptr, nullptr, B, H, W, s));
    NND_TRY(run_fc2(p, blob, w, delta_dst, 0, false, B, H, W, s));
    if (mask_dst) NND_TRY(run_conv(p, blob, w, C_M2, corr, mask_dst, nullptr, B, H, W, s));
    return NND_OK;
}

// NCHW <-> workspace tensor `dst` / `src` (its channel 0), slice of C channels starting at channel c0; c4: that tensor is in
// the 4-channel-interleaved layout; src == nullptr (to_tiled) writes zeros
static int to_tiled(const float* src, float* dst, int64_t dbs, int B, int C, int H, int W, hipStream_t s, int c0 = 0, bool c4 = false) {
    const long total = (long)B * C * H * W;
    hipLaunchKernelGGL(to_tiled_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, s, src, dst, (long)dbs, c0, B, C, H, W,
                       make_lay(H, W, true, c4));
    NND_LAUNCH_CHECK();
    return NND_OK;
}
static int from_tiled(const float* src, int64_t sbs, float* dst, int B, int C, int H, int W, hipStream_t s, int c0 = 0, bool c4 = false) {
    const long total = (long)B * C * H * W;
    hipLaunchKernelGGL(from_tiled_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, s, src, (long)sbs, c0, dst, B, C, H, W,
                       make_lay(H, W, true, c4));
    NND_LAUNCH_CHECK();
    return NND_OK;
}


}  // namespace nnd

using namespace nnd;

extern "C" {

int nnd_update_block_num_tensors(const nnd_update_block_desc* desc) {
    Plan p;
    if (make_plan(desc, &p) != NND_OK) return NND_ERR_INVALID;
    return p.sep ? 30 : 24;
}

int64_t nnd_update_block_packed_floats(const nnd_update_block_desc* desc) {
    Plan p;
    if (make_plan(desc, &p) != NND_OK) return NND_ERR_INVALID;
    return p.total;
}

int nnd_update_block_pack(const nnd_update_block_desc* desc, const float* const* t, float* out) {
    Plan p;
    int rc = make_plan(desc, &p);
    if (rc != NND_OK) return rc;
    NND_REQUIRE(t && out, "update_block_pack: null pointer");
    const int n = p.sep ? 30 : 24;
    for (int i = 0; i < n; ++i) NND_REQUIRE(t[i], "update_block_pack: tensor %d is null", i);
    const int hid = p.d.hidden_dim, fc = p.d.flow_channels;
    memset(out, 0, sizeof(float) * p.total);
    auto one = [&](int id, int ti) {
        const float* w[1] = {t[ti]};
        const float* b[1] = {t[ti + 1]};
        int co[1] = {p.L[id].Cout};
        pack_conv(p.L[id], 1, w, b, co, out);
    };
    auto two = [&](int id, int ta, int tb, int ca, int cb) {
        const float* w[2] = {t[ta], t[tb]};
        const float* b[2] = {t[ta + 1], t[tb + 1]};
        int co[2] = {ca, cb};
        pack_conv(p.L[id], 2, w, b, co, out);
    };
    one(C_C1, 0);
    one(C_C2, 2);
    memcpy(out + p.f1_w, t[4], sizeof(float) * 128 * fc * 49);
    memcpy(out + p.f1_b, t[5], sizeof(float) * 128);
    for (int ch = 0; ch < 128; ++ch)
        for (int k = 0; k < fc * 49; ++k) out[p.f1_wt + (int64_t)k * 128 + ch] = t[4][(int64_t)ch * fc * 49 + k];
    one(C_F2, 6);
    one(C_CV, 8);
    // GRU convs three ways: full (API path), [h | motion+flow] channels only (loop), `inp` channels only
    // (per-pair context term, carries the bias).  Source channel order of the reference: [h | inp | motion+flow].
    const int ctx = p.d.context_dim, gin = 2 * hid + ctx;
    std::vector<int> map_x(2 * hid), map_c(ctx);
    for (int i = 0; i < hid; ++i) {
        map_x[i] = i;
        map_x[hid + i] = hid + ctx + i;
    }
    for (int i = 0; i < ctx; ++i) map_c[i] = hid + i;
    auto gru = [&](int pass, int tz, int tr, int tq) {
        const int zr = pass == 0 ? C_ZR1 : C_ZR2, q = pass == 0 ? C_Q1 : C_Q2;
        two(zr, tz, tr, hid, hid);
        one(q, tq);
        const float* wzr[2] = {t[tz], t[tr]};
        const float* bzr[2] = {t[tz + 1], t[tr + 1]};
        const float* nob[2] = {nullptr, nullptr};
        int czr[2] = {hid, hid};
        pack_conv(p.L[zr - C_ZR1 + C_ZR1X], 2, wzr, nob, czr, out, map_x.data(), gin);
        pack_conv(p.L[zr - C_ZR1 + C_ZR1C], 2, wzr, bzr, czr, out, map_c.data(), gin);
        const float* wq[1] = {t[tq]};
        const float* bq[1] = {t[tq + 1]};
        int cq[1] = {hid};
        pack_conv(p.L[q - C_ZR1 + C_ZR1X], 1, wq, nob, cq, out, map_x.data(), gin);
        pack_conv(p.L[q - C_ZR1 + C_ZR1C], 1, wq, bq, cq, out, map_c.data(), gin);
    };
    gru(0, 10, 12, 14);
    int k = 16;
    if (p.sep) {
        gru(1, 16, 18, 20);
        k = 22;
    }
    two(C_FHM, k, k + 4, hid, 2 * hid);
    memcpy(out + p.fc2_w, t[k + 2], sizeof(float) * fc * hid * 9);
    memcpy(out + p.fc2_b, t[k + 3], sizeof(float) * fc);
    one(C_M2, k + 6);
    return NND_OK;
}

int nnd_update_block_scale_slots(const nnd_update_block_desc* desc, int64_t* offsets, int n) {
    Plan p;
    int rc = make_plan(desc, &p);
    if (rc != NND_OK) return rc;
    for (int i = 0; offsets && i < n && i < C_COUNT; ++i) offsets[i] = p.L[i].arith == 2 ? p.L[i].tail_off() : -1;
    return C_COUNT;
}

int nnd_update_block_calibration_finish(const nnd_update_block_desc* desc, float* packed_dev, int32_t* status_dev, void* stream) {
    Plan p;
    int rc = make_plan(desc, &p);
    if (rc != NND_OK) return rc;
    NND_REQUIRE(packed_dev, "update_block_calibration_finish: null blob");
    int64_t offs[C_COUNT];
    int n = 0;
    for (int i = 0; i < C_COUNT; ++i)
        if (p.L[i].arith == 2 && (p.sep || (i != C_ZR2 && i != C_Q2 && i != C_ZR2X && i != C_Q2X && i != C_ZR2C && i != C_Q2C))) offs[n++] = p.L[i].tail_off();
    return calib_finish(packed_dev, offs, n, status_dev, (hipStream_t)stream);
}

int64_t nnd_update_block_workspace_floats(const nnd_update_block_desc* desc, int B, int H, int W) {
    Plan p;
    if (make_plan(desc, &p) != NND_OK || B <= 0 || H <= 0 || W <= 0) return NND_ERR_INVALID;
    Bufs b;
    carve(p, B, H, W, nullptr, &b);
    return b.total;
}

int nnd_update_block_forward(const nnd_update_block_desc* desc, const float* packed, const float* net, const float* inp,
                             const float* corr, const float* flow, float* net_out, float* mask_out, float* delta_out,
                             float* workspace, int B, int H, int W, void* stream) {
    Plan p;
    int rc = make_plan(desc, &p);
    if (rc != NND_OK) return rc;
    NND_REQUIRE(packed && net && inp && corr && flow && net_out && delta_out && workspace, "update_block_forward: null pointer");
    NND_REQUIRE(B > 0 && H > 0 && W > 0, "update_block_forward: bad shape");
    CalibScope calib((p.d.flags & NND_FLAG_CALIBRATE) && p.d.arithmetic == 2);
    hipStream_t s = (hipStream_t)stream;
    Bufs w;
    carve(p, B, H, W, workspace, &w);
    const int hid = p.d.hidden_dim, ctx = p.d.context_dim, fc = p.d.flow_channels, hxC = 2 * hid + ctx;
    const int64_t n = tiled_plane(H, W);
    // C-ABI tensors are NCHW, the workspace is tile-major: convert on the way in and out
    NND_TRY(to_tiled(net, w.hx, hxC * n, B, hid, H, W, s, 0, ws_c4()));
    NND_TRY(to_tiled(inp, w.hx, hxC * n, B, ctx, H, W, s, hid, ws_c4()));
    NND_TRY(to_tiled(flow, w.hx, hxC * n, B, fc, H, W, s, hxC - fc, ws_c4()));
    NND_TRY(to_tiled(flow, w.flow, fc * n, B, fc, H, W, s));
    NND_TRY(to_tiled(corr, w.corr, p.d.cor_planes * n, B, p.d.cor_planes, H, W, s));
    Act c = act(w.corr, p.d.cor_planes * n, p.d.cor_planes);
    NND_TRY(run_update(p, packed, w, c, w.flow, mask_out ? w.mask : nullptr, w.delta, B, H, W, s));
    NND_TRY(from_tiled(w.hx, hxC * n, net_out, B, hid, H, W, s, 0, ws_c4()));
    NND_TRY(from_tiled(w.delta, fc * n, delta_out, B, fc, H, W, s));
    if (mask_out) NND_TRY(from_tiled(w.mask, p.d.mask_channels * n, mask_out, B, p.d.mask_channels, H, W, s));
    return NND_OK;
}

// CREStereo variant of the loop (cre_stereo/model.py:221-284): the correlation features come from the AGCL kernels
// (iter mode: extra == nullptr, needs the `warped` scratch of B*C*H*W floats; offset mode otherwise: with a scratch of
// 2*B*C*H*W floats the maps are copied channels-last once and sampled line by line), the state is a 2-channel flow
// and iteration `it` searches a 3x3 window when it is odd, 1x9 when even.
struct CreArgs {
    const float* f1;
    const float* f2;
    const float* extra;
    float* warped;           // caller-owned scratch
    int64_t warped_floats;   // its size: >= B*C*H*W in iter mode; >= 2*B*C*H*W selects the channels-last offset kernel
    int C;
};

// Profiling hook of nnd_profile_loop_conv: when set (thread-local, so only the calling thread's loops are affected), every
// iteration of the fused loop brackets conv `which` on the caller's stream with a pair of timing events.
struct LoopProbe {
    int which;
    bool empty = false;          // calibration run: both events in FRONT of the conv, nothing between them
    int marks = 0;
    std::vector<hipEvent_t> ev;  // 2 per iteration
};
static thread_local LoopProbe* t_probe = nullptr;
static void probe_mark(int id, hipStream_t s) {
    if (!t_probe || t_probe->which != id) return;
    const bool opening = (t_probe->marks++ & 1) == 0;
    if (t_probe->empty && !opening) return;
    for (int k = 0; k < (t_probe->empty ? 2 : 1); ++k) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return;
        (void)hipEventRecord(e, s);
        t_probe->ev.push_back(e);
    }
}

// geo_pyramid != nullptr selects the IGEV variant: combined two-volume lookup over `groups` groups and
// absolute coordinates into the update block / upsample (igev_stereo/model.py:152-158).  cre != nullptr selects the
// CREStereo variant (then pyramid is unused and disp_init is the initial 2-channel flow, NCHW, or null for zero).
static int enqueue_refine(const nnd_update_block_desc* desc, const float* packed, const float* pyramid, int num_levels,
                          int radius, const float* net, const float* inp, const float* disp_init, float* up_out,
                          int64_t up_iter_stride, float* low_out, float* net_out, float* workspace, int B, int H, int W,
                          int rate, int iters, void* stream, const float* geo_pyramid = nullptr, int groups = 1,
                          const CreArgs* cre = nullptr, const float* interleaved = nullptr, int flat_groups = 0) {
    // flat_groups > 0: GroupCorrBlock1D's lookup (Coarse2Fine RAFT-Stereo): `pyramid` holds flat_groups group volumes per sample
    Plan p;
    int rc = make_plan(desc, &p);
    if (rc != NND_OK) return rc;
    NND_REQUIRE(packed && (pyramid || cre) && net && inp && up_out && workspace, "refine: null pointer");
    const bool igev = geo_pyramid != nullptr;
    const int fc = p.d.flow_channels;
    if (cre) {
        NND_REQUIRE(fc == 2 && p.d.cor_planes == 36, "cre_stereo_refine: needs flow_channels 2 and cor_planes 36");
        NND_REQUIRE(cre->f1 && cre->f2 && (cre->extra || cre->warped), "cre_stereo_refine: null feature map / scratch");
        NND_REQUIRE(cre->extra || cre->warped_floats >= (int64_t)B * cre->C * H * W,
                    "cre_stereo_refine: iter mode needs a scratch of B*C*H*W = %lld floats, got %lld", (long long)B * cre->C * H * W,
                    (long long)cre->warped_floats);
    } else {
        NND_REQUIRE(fc == 1, "raft_stereo_refine: flow_channels must be 1");
        NND_REQUIRE(p.d.cor_planes == num_levels * (2 * radius + 1) * (igev ? 2 * groups : (flat_groups > 0 ? flat_groups : 1)),
                    "refine: cor_planes %d does not match levels*(2r+1)%s", p.d.cor_planes, igev ? "*2*groups" : (flat_groups > 0 ? "*groups" : ""));
        NND_REQUIRE(!(igev && flat_groups > 0), "refine: the IGEV and the group-RAFT lookups exclude each other");
    }

    NND_REQUIRE(p.d.mask_channels == 9 * rate * rate, "raft_stereo_refine: mask_channels %d != 9*rate^2", p.d.mask_channels);
    NND_REQUIRE(B > 0 && H > 0 && W > 0 && iters > 0, "raft_stereo_refine: bad shape");
    CalibScope calib((p.d.flags & NND_FLAG_CALIBRATE) && p.d.arithmetic == 2);
    hipStream_t s = (hipStream_t)stream;
    Bufs w;
    carve(p, B, H, W, workspace, &w);
    const int hid = p.d.hidden_dim, ctx = p.d.context_dim, hxC = 2 * hid + ctx;
    const Lay lay = make_lay(H, W, true);
    const int64_t n = lay.plane;  // channel stride of every (tile-major) workspace tensor
    NND_TRY(to_tiled(net, w.hx, hxC * n, B, hid, H, W, s, 0, ws_c4()));
    NND_TRY(to_tiled(inp, w.hx, hxC * n, B, ctx, H, W, s, hid, ws_c4()));
    float* hx_flow = w.hx + ws_chan(hxC - fc, n);
    const unsigned eg = (unsigned)cdiv64((int64_t)B * H * W, 256);
    if (cre) {
        if (disp_init) {
            NND_TRY(to_tiled(disp_init, w.flow, fc * n, B, fc, H, W, s));
            NND_TRY(to_tiled(disp_init, w.hx, hxC * n, B, fc, H, W, s, hxC - fc, ws_c4()));
        } else {
            NND_HIP_CHECK(hipMemsetAsync(w.flow, 0, sizeof(float) * B * fc * n, s));
            NND_TRY(to_tiled(nullptr, w.hx, hxC * n, B, fc, H, W, s, hxC - fc, ws_c4()));  // zero flow channels of the GRU input
        }
    } else {
        hipLaunchKernelGGL(init_coords_kernel, dim3(eg), dim3(256), 0, s, w.coords, w.flow, hx_flow, (long)(hxC * n), ws_pm(), disp_init, B, H,
                           W, igev ? 1 : 0, lay);
        NND_LAUNCH_CHECK();
    }
    // offset mode with a scratch: channels-last copies of the two maps, made once for all iterations of the stage
    // (explicit opt-in by size: a scratch smaller than 2*B*C*H*W floats — e.g. a caller reusing its iter-mode scratch — takes the
    //  planar offset kernel instead of being written past its end)
    const bool cre_cl = cre && cre->extra && cre->warped && cre->warped_floats >= 2 * (int64_t)B * cre->C * H * W &&
                        agcl_offset_cl_supported(cre->C) && !switches().agcl_v1;
    if (cre_cl) {
        NND_TRY(nchw_to_nhwc_launch(cre->f1, cre->warped, B, cre->C, H * W, s));
        NND_TRY(nchw_to_nhwc_launch(cre->f2, cre->warped + (int64_t)B * cre->C * H * W, B, cre->C, H * W, s));
    }
    Act c = act(w.corr, p.d.cor_planes * n, p.d.cor_planes);
    auto lookup = [&](hipStream_t st_, int it) -> int {
        if (cre) {
            if (cre->extra && cre_cl)
                return agcl_offset_cl_launch(cre->warped, cre->warped + (int64_t)B * cre->C * H * W, w.flow, cre->extra, w.corr, B, cre->C,
                                             H, W, it & 1, st_, true);
            if (cre->extra) return agcl_offset_launch(cre->f1, cre->f2, w.flow, cre->extra, w.corr, B, cre->C, H, W, it & 1, st_, true);
            return agcl_iter_launch(cre->f1, cre->f2, w.flow, cre->warped, w.corr, B, cre->C, H, W, it & 1, st_, true);
        }
        if (igev) return igev_lookup_launch(pyramid, geo_pyramid, w.coords, w.corr, B, groups, H, W, num_levels, radius, st_, true);
        if (flat_groups > 0) return group_lookup_flat_launch(pyramid, w.coords, w.corr, B, flat_groups, H, W, num_levels, radius, st_, true);
        return corr1d_lookup_launch(pyramid, w.coords, w.corr, B, H, W, num_levels, radius, st_, true);
    };
    // per-pair context terms of the GRU convs (inp is constant over the iterations)
    for (int id : {(int)C_ZR1C, (int)C_Q1C, (int)C_ZR2C, (int)C_Q2C})
        if (p.sep || id < C_ZR2C) NND_TRY(run_conv(p, packed, w, id, c, nullptr, nullptr, B, H, W, s));
    // Per-iteration schedule, all on the caller's stream:
    //   convf1 + convf2 (flow branch of the motion encoder: one launch in split arithmetic), lookup (+convc1), convc2, conv, zr1, q1, zr2, q2,
    //   flow_head.conv1+mask.0, flow_head.conv2+advance, mask.2 + convex upsample (fused)
    // Rounds 1-2 ran the flow branch on a low-priority side stream beside lookup / convc2 (per-call fork/join events).
    // Measured on MI355X, same box, ms per 544x960 pair: side stream 11.79, in line 11.82 (KITTI batch 8 70.5 / 71.0,
    // CREStereo 39.8 / 39.9, IGEV batch 8 190.1 / 190.5): the overlap it bought (convc2 61 us beside the branch instead
    // of 44 + 20 + 20 in line) is what its 2 event records + 2 waits per iteration cost in launch gaps (0.47 -> 0.01 ms
    // per pair).  Equal speed, so the version without cross-stream state stays: nothing to join on error paths, no
    // shared stream pool, and the in-loop time of a kernel is its stand-alone time.  Earlier measurements (round 1, ms
    // per pair): a third stream for the mask branch 18.0 vs 17.6 | the linear schedule as a hipGraph 17.6 vs 17.6 |
    // the 3-stream DAG as a hipGraph 34.
    const bool no_fuse_up = switches().no_fused_upsample;  // the parity tests toggle these (nnd_reload_switches)
    const bool no_fuse_lk = switches().no_fused_lookup;
    const bool fused_up = !no_fuse_up && mask_upsample_supported(rate, 2 * hid, fc);
    // flow_head.conv2 folded into the fused mask / upsample launch: 1-channel flow (not CREStereo); NND_NO_FOLDED_FLOW_HEAD: two launches.
    // The second (coordinate, flow) buffer pair lives in the mask buffer, which the fused upsample never writes.
    // Only where it pays: at most one workgroup (4x8-pixel tile) per CU — measured at 68x120 batch 1 (255 tiles) -2.1 ... -2.8 us per
    // iteration, at 48x156 batch 64 -0.5 %: the fold's 134 KB of LDS leave one workgroup per CU where two of the plain kernel fit, so
    // beyond 256 tiles it would serialise what runs side by side (profiles/r04_folded_flow_head.txt).
    const bool fold_fh = fused_up && !cre && fc == 1 && mask_upsample_fold_supported(p.L[C_M2], hid) &&
                         (long)B * cdiv(H, 4) * cdiv(W, 8) <= 256 && !switches().no_folded_flow_head;
    float* coords_alt = w.mask;
    float* flow_alt = w.mask + ((int64_t)B * n + 63) / 64 * 64;
    const bool fused_lk = !cre && !no_fuse_lk && flat_groups == 0;  // (the group-RAFT lookup gathers other pixels' rows: stand-alone kernel)
    const bool merged_fbl = fused_lk && !igev && !switches().no_merged_fb_lookup && !switches().no_fused_flow_branch &&
                            flow_branch_supported(p.L[C_F2], fc) && flow_branch_lookup_supported(p.L[C_F2].arith);
    // a conv of the recurrence on the caller's stream (bracketed by timing events when nnd_profile_loop_conv asks for it)
    auto loop_conv = [&](int id) -> int {
        probe_mark(id, s);
        const int rc_ = run_conv(p, packed, w, id, c, nullptr, nullptr, B, H, W, s);
        probe_mark(id, s);
        return rc_;
    };
    // IGEV, fp16x2 convc1 inside the interleaved lookup: its operands are interpolated from the interleaved pyramid, whose largest
    // value bounds them (the sampler is a convex combination of two entries)
    if (calibrating() && fused_lk && interleaved && p.L[C_C1].arith == 2 && igev_lookup_convc1_il_supported(groups, num_levels, radius))
        NND_TRY(calib_amax_flat(interleaved, nnd_igev_interleaved_floats(B, groups, H, W, num_levels), packed + p.L[C_C1].tail_off(), s));
    for (int it = 0; it < iters; ++it) {
        // RAFT-Stereo, arithmetic 2: the flow branch and lookup + convc1 — independent of each other — as ONE launch of two
        // kinds of workgroups that share every CU (corr1d.hip: flow_branch_lookup_kernel)
        if (merged_fbl) {
            probe_mark(C_F2, s);
            NND_TRY(calib_flow_branch(p, packed, w, w.flow, B, H, W, s));
            const ConvIO fio = conv_io(p, w, C_F2, c, n, nullptr, nullptr);
            NND_TRY(flow_branch_lookup_launch(p.L[C_F2], packed, packed + p.f1_wt, packed + p.f1_b, w.flow, (int64_t)fc * n, fc, fio, pyramid,
                                              w.coords, p.L[C_C1], w.c1, 256 * n, B, H, W, num_levels, radius, s, ws_c4()));
            NND_TRY(debug_sync("encoder.convf1+convf2 | lookup+convc1", s));
            probe_mark(C_F2, s);
        } else {
        probe_mark(C_F2, s);
        NND_TRY(run_flow_branch(p, packed, w, w.flow, c, B, H, W, s));
        probe_mark(C_F2, s);
        if (fused_lk && interleaved && igev_lookup_convc1_il_supported(groups, num_levels, radius)) {  // IGEV over the group-interleaved copy of both pyramids
            NND_TRY(igev_lookup_convc1_il_launch(interleaved, groups, w.coords, p.L[C_C1], packed, w.c1, 256 * n, B, H, W, num_levels,
                                                 radius, s, ws_c4()));
        } else if (fused_lk && p.L[C_C1].arith == 0) {  // lookup + convc1 in one kernel, the sampled features never reach HBM
            NND_TRY(lookup_convc1_launch(pyramid, geo_pyramid, groups, w.coords, p.L[C_C1], packed, w.c1, 256 * n, B, H, W,
                                         num_levels, radius, s, ws_c4()));
        } else {
            NND_TRY(lookup(s, it));
            NND_TRY(run_conv(p, packed, w, C_C1, c, nullptr, nullptr, B, H, W, s));
        }
        }
        NND_TRY(loop_conv(C_C2));
        NND_TRY(loop_conv(C_CV));
        NND_TRY(loop_conv(C_ZR1X));
        NND_TRY(loop_conv(C_Q1X));
        if (p.sep) {
            NND_TRY(loop_conv(C_ZR2X));
            NND_TRY(loop_conv(C_Q2X));
        }
        NND_TRY(loop_conv(C_FHM));  // flow_head.conv1 and mask.0 in one launch
        float* up_it = up_out + (int64_t)it * up_iter_stride;
        if (fold_fh) {
            // flow_head.conv2 + the recurrence update inside the mask / upsample launch (mask_upsample.hip: MaskUpFlowHead): it reads
            // the old coordinate and writes the new state into the other buffer pair (a workgroup's halo belongs to its neighbours)
            MaskUpFlowHead fh;
            fh.x = w.fm; fh.xbs = (int64_t)3 * hid * n; fh.hid = hid;
            fh.w = packed + p.fc2_w; fh.bias = packed + p.fc2_b;
            fh.coords_in = w.coords; fh.coords_out = coords_alt; fh.flow_out = flow_alt; fh.delta_out = w.delta;
            fh.hx_flow = hx_flow; fh.hx_bs = (long)(hxC * n); fh.hx_pm = ws_pm(); fh.absolute = igev ? 1 : 0;
            NND_TRY(mask_upsample_launch(p.L[C_M2], packed, w.fm + hid * n, (int64_t)3 * hid * n, w.flow, up_it, B, H, W, rate, s,
                                         true, fc, ws_c4(), &fh));
            NND_TRY(debug_sync("flow_head.conv2 + mask.2 + upsample", s));
            std::swap(w.coords, coords_alt);
            std::swap(w.flow, flow_alt);
            continue;
        }
        NND_TRY(run_fc2(p, packed, w, w.delta, cre ? 2 : 1, igev, B, H, W, s));
        if (fused_up) {  // mask.2 + softmax + upsample in one kernel: the 9*r*r-channel mask never reaches HBM
            NND_TRY(mask_upsample_launch(p.L[C_M2], packed, w.fm + hid * n, (int64_t)3 * hid * n, w.flow, up_it, B, H, W, rate, s,
                                         true, fc, ws_c4()));
        } else {
            NND_TRY(run_conv(p, packed, w, C_M2, c, w.mask, nullptr, B, H, W, s));
            NND_TRY(convex_upsample_launch(w.flow, w.mask, up_it, B, fc, H, W, rate, s, true));
        }
    }
    if (low_out) NND_TRY(from_tiled(w.flow, fc * n, low_out, B, fc, H, W, s));
    if (net_out) NND_TRY(from_tiled(w.hx, hxC * n, net_out, B, hid, H, W, s, 0, ws_c4()));
    return NND_OK;
}

static int conv2d_layer(int Cout, int Cin, int KH, int KW, int arithmetic, ConvLayer* L, int64_t* total) {
    NND_REQUIRE(Cout > 0 && Cin > 0, "conv2d: bad channel counts");
    NND_REQUIRE((KH == 1 && KW == 1) || (KH == 3 && KW == 3) || (KH == 1 && KW == 5) || (KH == 5 && KW == 1),
                "conv2d: kernel %dx%d not built (1x1, 3x3, 1x5, 5x1)", KH, KW);
    NND_REQUIRE(arithmetic == 0 || conv_split_supported(KH, KW, Cin, 1, arithmetic),
                "conv2d: arithmetic %d not built for %dx%d with %d input channels (3 = bf16x3 split, Cin %% 16 == 0)", arithmetic, KH, KW, Cin);
    int64_t off = 0;
    *L = mk(KH, KW, Cin, Cout, &off, arithmetic);
    if (total) *total = off;
    return NND_OK;
}

int64_t nnd_conv2d_packed_floats_ex(int Cout, int Cin, int KH, int KW, int arithmetic) {
    ConvLayer L;
    int64_t total;
    if (conv2d_layer(Cout, Cin, KH, KW, arithmetic, &L, &total) != NND_OK) return NND_ERR_INVALID;
    return total;
}
int64_t nnd_conv2d_packed_floats(int Cout, int Cin, int KH, int KW) { return nnd_conv2d_packed_floats_ex(Cout, Cin, KH, KW, 0); }

int nnd_conv2d_pack_ex(const float* w_host, const float* b_host, int Cout, int Cin, int KH, int KW, int arithmetic, float* packed_host) {
    ConvLayer L;
    int rc = conv2d_layer(Cout, Cin, KH, KW, arithmetic, &L, nullptr);
    if (rc != NND_OK) return rc;
    NND_REQUIRE(w_host && b_host && packed_host, "conv2d_pack: null pointer");
    const float* w[1] = {w_host};
    const float* b[1] = {b_host};
    int co[1] = {Cout};
    pack_conv(L, 1, w, b, co, packed_host);
    return NND_OK;
}
int nnd_conv2d_pack(const float* w_host, const float* b_host, int Cout, int Cin, int KH, int KW, float* packed_host) {
    return nnd_conv2d_pack_ex(w_host, b_host, Cout, Cin, KH, KW, 0, packed_host);
}

int nnd_conv2d_forward_ex(const float* packed_dev, const float* x, float* y, int B, int Cin, int H, int W, int Cout, int KH,
                          int KW, int relu, int arithmetic, void* stream) {
    ConvLayer L;
    int rc = conv2d_layer(Cout, Cin, KH, KW, arithmetic, &L, nullptr);
    if (rc != NND_OK) return rc;
    NND_REQUIRE(packed_dev && x && y && B > 0 && H > 0 && W > 0, "conv2d_forward: bad argument");
    const int64_t n = (int64_t)H * W;
    ConvIO io{};
    io.src0 = act(const_cast<float*>(x), Cin * n, Cin);
    io.out0 = act(y, Cout * n, Cout);
    return launch_conv(L, packed_dev, io, relu ? EPI_RELU : EPI_LINEAR, B, H, W, (hipStream_t)stream);
}
int nnd_conv2d_calibrate_ex(float* packed_dev, const float* x, int B, int Cin, int H, int W, int Cout, int KH, int KW, int arithmetic,
                            int32_t* status_dev, void* stream) {
    ConvLayer L;
    int rc = conv2d_layer(Cout, Cin, KH, KW, arithmetic, &L, nullptr);
    if (rc != NND_OK) return rc;
    NND_REQUIRE(packed_dev && x && B > 0 && H > 0 && W > 0, "conv2d_calibrate: bad argument");
    if (L.arith != 2) return NND_OK;
    const int64_t n = (int64_t)H * W;
    rc = calib_amax_act(act(const_cast<float*>(x), Cin * n, Cin), make_lay(H, W, false), B, H, W, packed_dev + L.tail_off(), (hipStream_t)stream);
    if (rc != NND_OK) return rc;
    const int64_t off = L.tail_off();
    return calib_finish(packed_dev, &off, 1, status_dev, (hipStream_t)stream);
}

int nnd_conv2d_forward(const float* packed_dev, const float* x, float* y, int B, int Cin, int H, int W, int Cout, int KH,
                       int KW, int relu, void* stream) {
    return nnd_conv2d_forward_ex(packed_dev, x, y, B, Cin, H, W, Cout, KH, KW, relu, 0, stream);
}

int nnd_conv2d_offset_forward(const float* packed_dev, const float* x, float* y, int B, int Cin, int H, int W, int Cout, int KH,
                              int KW, float range, void* stream) {
    ConvLayer L;
    int rc = conv2d_layer(Cout, Cin, KH, KW, 0, &L, nullptr);
    if (rc != NND_OK) return rc;
    NND_REQUIRE(packed_dev && x && y && B > 0 && H > 0 && W > 0, "conv2d_offset_forward: bad argument");
    const int64_t n = (int64_t)H * W;
    ConvIO io{};
    io.src0 = act(const_cast<float*>(x), Cin * n, Cin);
    io.out0 = act(y, Cout * n, Cout);
    io.scale = range;
    return launch_conv(L, packed_dev, io, EPI_SIGMOID_RANGE, B, H, W, (hipStream_t)stream);
}

int nnd_raft_stereo_refine(const nnd_update_block_desc* desc, const float* packed, const float* pyramid, int num_levels,
                           int radius, const float* net, const float* inp, const float* disp_init, float* up_out,
                           int64_t up_iter_stride, float* low_out, float* net_out, float* workspace, int B, int H, int W,
                           int rate, int iters, void* stream) {
    // (hipGraph replay of the loop was measured and dropped: no faster than direct launches for the linear schedule,
    //  2x slower for a multi-stream DAG on ROCm 7.2 — see the schedule note in enqueue_refine.)
    return enqueue_refine(desc, packed, pyramid, num_levels, radius, net, inp, disp_init, up_out, up_iter_stride, low_out,
                          net_out, workspace, B, H, W, rate, iters, stream);
}

int nnd_raft_stereo_group_refine(const nnd_update_block_desc* desc, const float* packed, const float* group_pyramid, int num_groups,
                                 int num_levels, int radius, const float* net, const float* inp, const float* disp_init, float* up_out,
                                 int64_t up_iter_stride, float* low_out, float* net_out, float* workspace, int B, int H, int W, int rate,
                                 int iters, void* stream) {
    NND_REQUIRE(num_groups > 0, "raft_stereo_group_refine: num_groups must be positive");
    return enqueue_refine(desc, packed, group_pyramid, num_levels, radius, net, inp, disp_init, up_out, up_iter_stride, low_out, net_out,
                          workspace, B, H, W, rate, iters, stream, nullptr, 1, nullptr, nullptr, num_groups);
}

int nnd_igev_refine_reads_interleaved(int num_groups, int num_levels, int radius) {
    return !switches().no_fused_lookup && igev_lookup_convc1_il_supported(num_groups, num_levels, radius);
}

int nnd_igev_stereo_refine(const nnd_update_block_desc* desc, const float* packed, const float* feat_pyramid,
                           const float* geo_pyramid, const float* interleaved, int num_groups, int num_levels, int radius,
                           const float* net, const float* inp,
                           const float* disp_init, float* up_out, int64_t up_iter_stride, float* low_out, float* net_out,
                           float* workspace, int B, int H, int W, int rate, int iters, void* stream) {
    NND_REQUIRE(geo_pyramid && num_groups > 0, "igev_stereo_refine: geometry pyramid / groups missing");
    return enqueue_refine(desc, packed, feat_pyramid, num_levels, radius, net, inp, disp_init, up_out, up_iter_stride, low_out,
                          net_out, workspace, B, H, W, rate, iters, stream, geo_pyramid, num_groups, nullptr, interleaved);
}

int nnd_cre_stereo_refine(const nnd_update_block_desc* desc, const float* packed, const float* fmap1, const float* fmap2,
                          int C, const float* extra_offset, float* scratch, int64_t scratch_floats, const float* net, const float* inp,
                          const float* flow_init, float* up_out, int64_t up_iter_stride, float* low_out, float* net_out,
                          float* workspace, int B, int H, int W, int rate, int iters, void* stream) {
    NND_REQUIRE(fmap1 && fmap2, "cre_stereo_refine: null feature map");
    NND_REQUIRE(extra_offset || scratch, "cre_stereo_refine: iter mode (extra_offset == NULL) needs the scratch for the warped map");
    NND_REQUIRE(scratch_floats >= 0 && (scratch || scratch_floats == 0), "cre_stereo_refine: scratch_floats without a scratch");
    int rc = agcl_check("cre_stereo_refine", B, C, H, W);
    if (rc != NND_OK) return rc;
    CreArgs cre{fmap1, fmap2, extra_offset, scratch, scratch_floats, C};
    return enqueue_refine(desc, packed, nullptr, 0, 0, net, inp, flow_init, up_out, up_iter_stride, low_out, net_out, workspace,
                          B, H, W, rate, iters, stream, nullptr, 1, &cre);
}

int nnd_mask_upsample_forward(const float* packed_dev, const float* x, const float* flow, float* out, int B, int Cin, int H,
                              int W, int rate, void* stream) {
    ConvLayer L;
    int rc = conv2d_layer(9 * rate * rate, Cin, 1, 1, 0, &L, nullptr);
    if (rc != NND_OK) return rc;
    NND_REQUIRE(packed_dev && x && flow && out && B > 0 && H > 0 && W > 0, "mask_upsample_forward: bad argument");
    NND_REQUIRE(mask_upsample_supported(rate, Cin, 1), "mask_upsample_forward: rate %d / Cin %d not built", rate, Cin);
    return mask_upsample_launch(L, packed_dev, x, (int64_t)Cin * H * W, flow, out, B, H, W, rate, (hipStream_t)stream, false);
}

int nnd_num_convs(const nnd_update_block_desc* desc) {
    Plan p;
    if (make_plan(desc, &p) != NND_OK) return NND_ERR_INVALID;
    return C_LOOP_COUNT;
}

const char* nnd_conv_name(const nnd_update_block_desc* desc, int which) {
    (void)desc;
    return (which >= 0 && which < C_LOOP_COUNT) ? kConvNames[which] : "";
}

int nnd_profile_conv(const nnd_update_block_desc* desc, const float* packed, float* workspace, int B, int H, int W, int which,
                     int reps, void* stream, float* ms_out, double* flops_out) {
    Plan p;
    int rc = make_plan(desc, &p);
    if (rc != NND_OK) return rc;
    NND_REQUIRE(packed && workspace && ms_out && flops_out, "profile_conv: null pointer");
    NND_REQUIRE(which >= 0 && which < C_LOOP_COUNT && reps > 0, "profile_conv: bad conv index / reps");
    NND_REQUIRE(p.sep || (which != C_ZR2X && which != C_Q2X), "profile_conv: conv_gru has no second GRU pass");
    hipStream_t s = (hipStream_t)stream;
    Bufs w;
    carve(p, B, H, W, workspace, &w);
    const int64_t n = tiled_plane(H, W);
    Act c = act(w.corr, p.d.cor_planes * n, p.d.cor_planes);
    hipEvent_t e0, e1;
    NND_HIP_CHECK(hipEventCreate(&e0));
    NND_HIP_CHECK(hipEventCreate(&e1));
    rc = run_conv(p, packed, w, which, c, w.mask, w.delta, B, H, W, s);  // warm
    if (rc == NND_OK) {
        (void)hipEventRecord(e0, s);
        for (int i = 0; i < reps && rc == NND_OK; ++i) rc = run_conv(p, packed, w, which, c, w.mask, w.delta, B, H, W, s);
        (void)hipEventRecord(e1, s);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        *ms_out = ms / reps;
        *flops_out = p.L[which].flops(B, H, W);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}
}

static int profile_loop(const nnd_update_block_desc* desc, const float* packed, const float* pyramid, int num_levels, int radius,
                        const float* net, const float* inp, float* up_out, float* workspace, int B, int H, int W, int rate, int iters,
                        int which, void* stream, float* ms_out, bool empty) {
    NND_REQUIRE(ms_out && which >= 0 && which < C_LOOP_COUNT && which != C_C1 && which != C_M2 && iters > 0,
                "profile_loop_conv: conv %d is not a stand-alone launch of the recurrence", which);
    LoopProbe probe;
    probe.which = which;
    probe.empty = empty;
    t_probe = &probe;
    int rc = enqueue_refine(desc, packed, pyramid, num_levels, radius, net, inp, nullptr, up_out, 0, nullptr, nullptr, workspace, B,
                            H, W, rate, iters, stream);
    t_probe = nullptr;
    if (rc == NND_OK && hipStreamSynchronize((hipStream_t)stream) != hipSuccess) rc = NND_ERR_HIP;
    double sum = 0.0;
    int n = 0;
    for (size_t i = 0; i + 1 < probe.ev.size(); i += 2) {
        float ms = 0.f;
        // the first iteration starts behind the per-pair prologue: leave it out of the average
        if (rc == NND_OK && i >= 2 && hipEventElapsedTime(&ms, probe.ev[i], probe.ev[i + 1]) == hipSuccess) {
            sum += ms;
            ++n;
        }
    }
    for (hipEvent_t e : probe.ev) (void)hipEventDestroy(e);
    if (rc != NND_OK) return rc;
    NND_REQUIRE(n > 0, "profile_loop_conv: needs at least 2 iterations");
    *ms_out = (float)(sum / n);
    return NND_OK;
}

extern "C" int nnd_profile_loop_conv(const nnd_update_block_desc* desc, const float* packed, const float* pyramid, int num_levels,
                                     int radius, const float* net, const float* inp, float* up_out, float* workspace, int B, int H,
                                     int W, int rate, int iters, int which, void* stream, float* ms_out) {
    return profile_loop(desc, packed, pyramid, num_levels, radius, net, inp, up_out, workspace, B, H, W, rate, iters, which, stream, ms_out,
                        false);
}

extern "C" int nnd_profile_loop_event_pair(const nnd_update_block_desc* desc, const float* packed, const float* pyramid, int num_levels,
                                           int radius, const float* net, const float* inp, float* up_out, float* workspace, int B, int H,
                                           int W, int rate, int iters, int which, void* stream, float* ms_out) {
    return profile_loop(desc, packed, pyramid, num_levels, radius, net, inp, up_out, workspace, B, H, W, rate, iters, which, stream, ms_out,
                        true);
}

// ------------------------------------------------------------------------------ MFMA peak probe
// Dependent chains of v_mfma_f32_32x32x2_f32 with no memory traffic: what the chip sustains on this box at
// `waves_per_simd` resident waves (clock under load included).  Diagnostic used to put the conv numbers in
// context; not on any product path.
namespace nnd {
typedef float f32x16_ __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(256) mfma_probe_kernel(float* out, int iters, float seed) {
    f32x16_ acc0, acc1;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    float av = seed + threadIdx.x * 1e-3f, bv = seed * 0.5f + threadIdx.x * 2e-3f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, av, acc1, 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    if (s == 12345.678f) out[0] = s;  // keep the chain live without a real store
}
}  // namespace nnd

namespace nnd {
// the same for the 16-bit MFMA of the split arithmetics: 4 independent v_mfma_f32_32x32x16_{f16,bf16} chains per wave on pseudo-random
// operands (zeros would flatter the power draw), nothing else in flight — what the chip sustains under its power limit on this box
typedef _Float16 f16x8_ __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8_ __attribute__((ext_vector_type(8)));
template <bool BF16>
__global__ void __launch_bounds__(256) mfma16_probe_kernel(float* out, unsigned long long* clk, int iters, unsigned seed) {
    unsigned s = seed ^ (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
    float av[8], bv[8];
    for (int i = 0; i < 8; ++i) {
        s = s * 1664525u + 1013904223u;
        av[i] = ((int)(s >> 20) & 1023) / 512.f - 1.f;
        s = s * 1664525u + 1013904223u;
        bv[i] = ((int)(s >> 20) & 1023) / 512.f - 1.f;
    }
    f32x16_ acc[4];
    for (int c = 0; c < 4; ++c)
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if constexpr (BF16) {
        bf16x8_ a, b;
        for (int i = 0; i < 8; ++i) a[i] = (__bf16)av[i], b[i] = (__bf16)bv[i];
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
    } else {
        f16x8_ a, b;
        for (int i = 0; i < 8; ++i) a[i] = (_Float16)av[i], b[i] = (_Float16)bv[i];
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[c], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
    for (int c = 0; c < 4; ++c)
        for (int i = 0; i < 16; ++i) sum += acc[c][i];
    if (sum == 12345.678f) out[0] = sum;  // keep the chains live without a real store
    if (threadIdx.x == 0 && blockIdx.x < 256) {
        clk[2 * blockIdx.x] = t1 - t0;
        clk[2 * blockIdx.x + 1] = r1 - r0;
    }
}
}  // namespace nnd

// scratch_dev: >= 1 float; clk_dev: >= 512 64-bit words.  One calibration launch, then launches of about `target_ms` (the clock settles
// under load within a few milliseconds); the last one is reported: TFLOP/s and the shader clock it ran at (s_memtime / s_memrealtime).
extern "C" int nnd_profile_mfma16_peak(int bf16, int waves_per_simd, float target_ms, void* stream, float* scratch_dev,
                                       unsigned long long* clk_dev, float* tflops_out, float* ghz_out) {
    NND_REQUIRE(waves_per_simd >= 1 && waves_per_simd <= 4 && target_ms > 0.f && target_ms <= 100.f && scratch_dev && clk_dev && tflops_out && ghz_out,
                "mfma16_peak: bad argument");
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t e0, e1;
    NND_HIP_CHECK(hipEventCreate(&e0));
    NND_HIP_CHECK(hipEventCreate(&e1));
    dim3 grid(256 * waves_per_simd), block(256);
    int iters = 2000;
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, s);
        if (bf16) hipLaunchKernelGGL(nnd::mfma16_probe_kernel<true>, grid, block, 0, s, scratch_dev, clk_dev, iters, 12345u + rep);
        else hipLaunchKernelGGL(nnd::mfma16_probe_kernel<false>, grid, block, 0, s, scratch_dev, clk_dev, iters, 12345u + rep);
        (void)hipEventRecord(e1, s);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep == 0) iters = (int)(iters * target_ms / (ms > 1e-3f ? ms : 1e-3f));
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    unsigned long long h[512];
    NND_HIP_CHECK(hipMemcpyAsync(h, clk_dev, sizeof(h), hipMemcpyDeviceToHost, s));
    NND_HIP_CHECK(hipStreamSynchronize(s));
    double cyc = 0, rt = 0;
    for (int i = 0; i < 256; ++i) cyc += (double)h[2 * i], rt += (double)h[2 * i + 1];
    const double flops = 2.0 * 32 * 32 * 16 * 4.0 * iters * 4.0 * 256.0 * waves_per_simd;
    *tflops_out = (float)(flops / (ms * 1e-3) / 1e12);
    *ghz_out = (float)(rt > 0 ? cyc / rt * 0.1 : 0.0);  // s_memrealtime counts at 100 MHz
    return NND_OK;
}

extern "C" int nnd_profile_mfma_peak(int waves_per_simd, int iters, void* stream, float* scratch_dev, float* tflops_out) {
    NND_REQUIRE(waves_per_simd >= 1 && waves_per_simd <= 8 && iters > 0 && scratch_dev && tflops_out, "mfma_peak: bad argument");
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t e0, e1;
    NND_HIP_CHECK(hipEventCreate(&e0));
    NND_HIP_CHECK(hipEventCreate(&e1));
    dim3 grid(256 * waves_per_simd), block(256);
    hipLaunchKernelGGL(nnd::mfma_probe_kernel, grid, block, 0, s, scratch_dev, 16, 1.0f);
    (void)hipEventRecord(e0, s);
    hipLaunchKernelGGL(nnd::mfma_probe_kernel, grid, block, 0, s, scratch_dev, iters, 1.0f);
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    double flops = 2.0 * 32 * 32 * 2 * 16.0 * iters * 4.0 * 256.0 * waves_per_simd;
    *tflops_out = (float)(flops / (ms * 1e-3) / 1e12);
    return NND_OK;
}
