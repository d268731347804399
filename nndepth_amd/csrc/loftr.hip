// LoFTR encoder layer with linear attention, as used by CREStereo's self / cross attention.
//
// Replaces LoFTREncoderLayer.forward  nndepth/blocks/transformer.py:39-66  and LinearAttention.forward
// nndepth/blocks/attn_block.py:23-58 (no masks); restated in oracle/cre_ref.py: loftr_layer / linear_attention.
//
// The reference flattens the (N,C,H,W) maps to (N, H*W, C) tokens and back; every Linear of the layer is therefore a
// 1x1 convolution of the map, and the layer runs here on NCHW without a transpose:
//   q = Wq x, k = Wk s, v = Wv s                         conv_mfma (1x1, no bias)
//   Q = elu(q)+1, K = elu(k)+1, per head h (D = C/nhead channels):
//   KV[h] = sum_p K[h,:,p] (v[h,:,p] / S)^T  (D x D),  Ksum[h] = sum_p K[h,:,p]        attn_kv_kernel (two stages)
//   msg[h,:,p] = (Q[h,:,p]^T KV[h]) * S / (Q[h,:,p] . Ksum[h] + eps)                    attn_msg_kernel
//   m = LayerNorm1(Wm msg)                               conv_mfma + layernorm_kernel
//   m = LayerNorm2(W2 relu(W1 [x | m]))                  conv_mfma (virtual concat, ReLU epilogue) x 2 + layernorm_kernel
//   out = x + m                                          fused into the second layernorm_kernel
#include "common.h"

#include <cmath>
#include <cstring>

namespace nnd {

__device__ __forceinline__ float elu1(float x) { return (x > 0.f ? x : expm1f(x)) + 1.0f; }

constexpr int KV_PIX = 256;  // pixels per workgroup of the first reduction stage

// stage 1: partial[n][h][chunk][D*D + D] over KV_PIX source pixels.  grid (chunks, nhead, N), 256 threads.
// thread t owns outputs (d = t / 8 .. , e-range) : D = 32 -> 1024 KV entries = 4 per thread (d = t >> 3, e = (t & 7)*4..+3)
__global__ void __launch_bounds__(256) attn_kv_partial_kernel(const float* __restrict__ k, const float* __restrict__ v, long bs,
                                                              int S, float inv_s, float* __restrict__ partial, int nchunks) {
    constexpr int D = 32;
    __shared__ float ks[D][KV_PIX + 1], vs[D][KV_PIX + 1];
    const int chunk = blockIdx.x, h = blockIdx.y, n = blockIdx.z;
    const int p0 = chunk * KV_PIX;
    const float* kb = k + (long)n * bs + (long)h * D * S;
    const float* vb = v + (long)n * bs + (long)h * D * S;
    for (int e = threadIdx.x; e < D * KV_PIX; e += 256) {
        const int d = e / KV_PIX, p = e % KV_PIX;
        const bool ok = p0 + p < S;
        ks[d][p] = ok ? elu1(kb[(long)d * S + p0 + p]) : 0.f;
        vs[d][p] = ok ? vb[(long)d * S + p0 + p] * inv_s : 0.f;
    }
    __syncthreads();
    const int d = threadIdx.x >> 3, e0 = (threadIdx.x & 7) * 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < KV_PIX; ++p) {
        const float kd = ks[d][p];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = fmaf(kd, vs[e0 + j][p], acc[j]);
    }
    float* o = partial + (((long)n * gridDim.y + h) * nchunks + chunk) * (D * D + D);
#pragma unroll
    for (int j = 0; j < 4; ++j) o[d * D + e0 + j] = acc[j];
    if (threadIdx.x < D) {
        float s = 0.f;
        for (int p = 0; p < KV_PIX; ++p) s += ks[threadIdx.x][p];
        o[D * D + threadIdx.x] = s;
    }
}

// stage 2: kv[n][h][D*D + D] = sum over chunks, in order.  grid (nhead, N), 256 threads
__global__ void __launch_bounds__(256) attn_kv_final_kernel(const float* __restrict__ partial, int nchunks, float* __restrict__ kv) {
    constexpr int D = 32, E = D * D + D;
    const long base = ((long)blockIdx.y * gridDim.x + blockIdx.x);
    for (int i = threadIdx.x; i < E; i += 256) {
        float s = 0.f;
        for (int c = 0; c < nchunks; ++c) s += partial[(base * nchunks + c) * E + i];
        kv[base * E + i] = s;
    }
}

// msg[n, h*D + e, p] = (sum_d Q[d] KV[d][e]) * (S / (sum_d Q[d] Ksum[d] + eps)); thread = pixel, workgroup = 256 pixels of a head
__global__ void __launch_bounds__(256) attn_msg_kernel(const float* __restrict__ q, long bs, const float* __restrict__ kv, int L,
                                                       float s_len, float eps, float* __restrict__ msg, long mbs) {
    constexpr int D = 32, E = D * D + D;
    __shared__ float kvs[E];
    const int h = blockIdx.y, n = blockIdx.z;
    for (int i = threadIdx.x; i < E; i += 256) kvs[i] = kv[((long)n * gridDim.y + h) * E + i];
    __syncthreads();
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= L) return;
    const float* qb = q + (long)n * bs + (long)h * D * L + p;
    float Q[D];
    float den = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        Q[d] = elu1(qb[(long)d * L]);
        den = fmaf(Q[d], kvs[D * D + d], den);
    }
    const float z = 1.0f / (den + eps);
    float* mo = msg + (long)n * mbs + (long)h * D * L + p;
#pragma unroll 4
    for (int e = 0; e < D; ++e) {
        float acc = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) acc = fmaf(Q[d], kvs[d * D + e], acc);
        mo[(long)e * L] = acc * z * s_len;
    }
}

// y[n,:,p] = LayerNorm_C(x[n,:,p]) * gamma + beta (+ res[n,:,p]).  Workgroup = 32 pixels x 8 channel slices: a half-wave reads
// 32 consecutive pixels of a channel (one 128-B line), a thread keeps its C/8 values in registers (C <= 8*LN_MAXC) so x is read
// once; mean and variance (two-pass, as torch) meet in LDS.  The maps here have a few thousand pixels: one thread per pixel
// walking all C channels three times (round 1) put 8 workgroups on the chip and took 168 us at 33x60x2, C = 256.
constexpr int LN_MAXC = 64;
__global__ void __launch_bounds__(256) layernorm_kernel(const float* __restrict__ x, long bs, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ res, long rbs,
                                                        float* __restrict__ y, long ybs, int C, int L, float eps) {
    __shared__ float part[8][33];
    const int px = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int p = blockIdx.x * 32 + px, n = blockIdx.y;
    const bool ok = p < L;
    const int per = (C + 7) / 8, c0 = sl * per, c1 = min(c0 + per, C);
    const float* xb = x + (long)n * bs + (ok ? p : 0);
    float v[LN_MAXC];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        v[i] = (c0 + i < c1) ? xb[(long)(c0 + i) * L] : 0.f;
        sum += v[i];
    }
    part[sl][px] = sum;
    __syncthreads();
    float mean = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) mean += part[j][px];
    mean /= (float)C;
    __syncthreads();
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const float d = (c0 + i < c1) ? v[i] - mean : 0.f;
        var = fmaf(d, d, var);
    }
    part[sl][px] = var;
    __syncthreads();
    var = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) var += part[j][px];
    const float rstd = 1.0f / sqrtf(var / (float)C + eps);
    if (!ok) return;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = c0 + i;
        if (c < c1) {
            float o = (v[i] - mean) * rstd * gamma[c] + beta[c];
            if (res) o = res[(long)n * rbs + (long)c * L + p] + o;
            y[(long)n * ybs + (long)c * L + p] = o;
        }
    }
}

// ------------------------------------------------------------------------------------------ plan
struct LoftrPlan {
    int C, nhead;
    ConvLayer q, k, v, merge, mlp0, mlp2;
    int64_t bq, bk, bv, bm, b0, b2;  // blob offsets of the six packed 1x1 convs
    int64_t ln1, ln2;                // gamma | beta (2*C floats each)
    int64_t total;
};

static ConvLayer lin(int Cout, int Cin, int64_t* off, int64_t* base) {
    ConvLayer l;
    l.KH = l.KW = 1; l.Cin = Cin; l.Cout = Cout; l.stride = 1;
    l.CI_T = conv_ci_t(1, 1, Cin);
    l.nchunks = cdiv(Cin, l.CI_T);
    l.ncb = cdiv(Cout, 32);
    int64_t o = 0;
    l.w_off = o; o += l.w_floats();
    l.b_off = o; o += l.b_floats();
    *base = *off;
    *off += o;
    return l;
}

static int make_loftr_plan(int d_model, int nhead, LoftrPlan* p) {
    NND_REQUIRE(d_model > 0 && nhead > 0 && d_model % nhead == 0 && d_model / nhead == 32,
                "loftr: d_model %d / nhead %d: the attention kernels are built for 32 channels per head", d_model, nhead);
    NND_REQUIRE(d_model <= 8 * LN_MAXC, "loftr: d_model %d above the LayerNorm kernel's %d channels", d_model, 8 * LN_MAXC);
    p->C = d_model; p->nhead = nhead;
    int64_t off = 0;
    p->q = lin(d_model, d_model, &off, &p->bq);
    p->k = lin(d_model, d_model, &off, &p->bk);
    p->v = lin(d_model, d_model, &off, &p->bv);
    p->merge = lin(d_model, d_model, &off, &p->bm);
    p->mlp0 = lin(2 * d_model, 2 * d_model, &off, &p->b0);
    p->mlp2 = lin(d_model, 2 * d_model, &off, &p->b2);
    p->ln1 = off; off += 2 * d_model;
    p->ln2 = off; off += 2 * d_model;
    p->total = off;
    return NND_OK;
}

static int run_lin(const ConvLayer& L, const float* base, const float* x0, int c0, const float* x1, int c1, int64_t xbs0,
                   int64_t xbs1, float* y, int64_t ybs, bool relu, int N, int H, int W, hipStream_t s) {
    ConvIO io{};
    io.src0 = Act{const_cast<float*>(x0), xbs0, c0};
    if (x1) io.src1 = Act{const_cast<float*>(x1), xbs1, c1};
    io.out0 = Act{y, ybs, L.Cout};
    return launch_conv(L, base, io, relu ? EPI_RELU : EPI_LINEAR, N, H, W, s);
}

}  // namespace nnd

using namespace nnd;

extern "C" {

int64_t nnd_loftr_packed_floats(int d_model, int nhead) {
    LoftrPlan p;
    if (make_loftr_plan(d_model, nhead, &p) != NND_OK) return NND_ERR_INVALID;
    return p.total;
}

// floats: q, k, v, msg, t (2C-wide mlp hidden) maps + attention partials
int64_t nnd_loftr_workspace_floats(int d_model, int nhead, int N, int H, int W) {
    LoftrPlan p;
    if (make_loftr_plan(d_model, nhead, &p) != NND_OK || N <= 0 || H <= 0 || W <= 0) return NND_ERR_INVALID;
    const int64_t L = (int64_t)H * W, map = (int64_t)N * d_model * L;
    const int64_t nchunks = cdiv64(L, KV_PIX);
    return 6 * map + (int64_t)N * nhead * (nchunks + 1) * (32 * 32 + 32) + 64;
}

// tensors (host): q_proj.weight, k_proj.weight, v_proj.weight, merge.weight (C,C), mlp.0.weight (2C,2C), mlp.2.weight (C,2C),
// norm1.weight, norm1.bias, norm2.weight, norm2.bias — the state_dict order of the reference layer.
int nnd_loftr_pack(int d_model, int nhead, const float* const* t, float* packed_host) {
    LoftrPlan p;
    int rc = make_loftr_plan(d_model, nhead, &p);
    if (rc != NND_OK) return rc;
    NND_REQUIRE(t && packed_host, "loftr_pack: null pointer");
    for (int i = 0; i < 10; ++i) NND_REQUIRE(t[i], "loftr_pack: tensor %d is null", i);
    memset(packed_host, 0, sizeof(float) * p.total);
    const ConvLayer* Ls[6] = {&p.q, &p.k, &p.v, &p.merge, &p.mlp0, &p.mlp2};
    const int64_t bases[6] = {p.bq, p.bk, p.bv, p.bm, p.b0, p.b2};
    for (int i = 0; i < 6; ++i) {
        const float* w[1] = {t[i]};
        const float* b[1] = {nullptr};
        int co[1] = {Ls[i]->Cout};
        pack_conv(*Ls[i], 1, w, b, co, packed_host + bases[i]);
    }
    memcpy(packed_host + p.ln1, t[6], sizeof(float) * d_model);
    memcpy(packed_host + p.ln1 + d_model, t[7], sizeof(float) * d_model);
    memcpy(packed_host + p.ln2, t[8], sizeof(float) * d_model);
    memcpy(packed_host + p.ln2 + d_model, t[9], sizeof(float) * d_model);
    return NND_OK;
}

// x, source, out: (N, d_model, H, W) maps (= the reference's (N, H*W, C) tokens, transposed); out may alias neither input.
int nnd_loftr_layer_forward(int d_model, int nhead, const float* packed, const float* x, const float* source, float* out,
                            float* workspace, int N, int H, int W, void* stream) {
    LoftrPlan p;
    int rc = make_loftr_plan(d_model, nhead, &p);
    if (rc != NND_OK) return rc;
    NND_REQUIRE(packed && x && source && out && workspace && N > 0 && H > 0 && W > 0, "loftr_layer_forward: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const int C = d_model, L = H * W;
    const int64_t bs = (int64_t)C * L, map = (int64_t)N * bs;
    float* q = workspace;
    float* k = q + map;
    float* v = k + map;
    float* msg = v + map;
    float* t = msg + map;  // 2 maps wide
    const int nchunks = (int)cdiv64(L, KV_PIX);
    float* partial = t + 2 * map;
    float* kv = partial + (int64_t)N * nhead * nchunks * (32 * 32 + 32);
#define NND_TRY(x)                    \
    do {                              \
        if ((rc = (x)) != NND_OK) return rc; \
    } while (0)
    NND_TRY(run_lin(p.q, packed + p.bq, x, C, nullptr, 0, bs, 0, q, bs, false, N, H, W, s));
    NND_TRY(run_lin(p.k, packed + p.bk, source, C, nullptr, 0, bs, 0, k, bs, false, N, H, W, s));
    NND_TRY(run_lin(p.v, packed + p.bv, source, C, nullptr, 0, bs, 0, v, bs, false, N, H, W, s));
    hipLaunchKernelGGL(attn_kv_partial_kernel, dim3(nchunks, nhead, N), dim3(256), 0, s, k, v, (long)bs, L, 1.0f / (float)L, partial,
                       nchunks);
    NND_LAUNCH_CHECK();
    hipLaunchKernelGGL(attn_kv_final_kernel, dim3(nhead, N), dim3(256), 0, s, (const float*)partial, nchunks, kv);
    NND_LAUNCH_CHECK();
    hipLaunchKernelGGL(attn_msg_kernel, dim3(cdiv(L, 256), nhead, N), dim3(256), 0, s, q, (long)bs, (const float*)kv, L, (float)L, 1e-6f,
                       msg, (long)bs);
    NND_LAUNCH_CHECK();
    // merge + norm1 (q is free again: holds the merged message, then its normalised version in place)
    NND_TRY(run_lin(p.merge, packed + p.bm, msg, C, nullptr, 0, bs, 0, q, bs, false, N, H, W, s));
    hipLaunchKernelGGL(layernorm_kernel, dim3(cdiv(L, 32), N), dim3(256), 0, s, (const float*)q, (long)bs, packed + p.ln1,
                       packed + p.ln1 + C, (const float*)nullptr, 0L, q, (long)bs, C, L, 1e-5f);
    NND_LAUNCH_CHECK();
    // mlp on the virtual concat [x | message]
    NND_TRY(run_lin(p.mlp0, packed + p.b0, x, C, q, C, bs, bs, t, 2 * bs, true, N, H, W, s));
    NND_TRY(run_lin(p.mlp2, packed + p.b2, t, 2 * C, nullptr, 0, 2 * bs, 0, k, bs, false, N, H, W, s));
    hipLaunchKernelGGL(layernorm_kernel, dim3(cdiv(L, 32), N), dim3(256), 0, s, (const float*)k, (long)bs, packed + p.ln2,
                       packed + p.ln2 + C, x, (long)bs, out, (long)bs, C, L, 1e-5f);
    NND_LAUNCH_CHECK();
#undef NND_TRY
    return NND_OK;
}

}  // extern "C"
