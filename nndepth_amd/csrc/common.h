// Internal helpers shared by the HIP translation units (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include "../../include/nndepth_amd.h"

namespace nnd {

void set_error(const char* fmt, ...);

#define NND_HIP_CHECK(expr)                                                              \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            nnd::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return NND_ERR_HIP;                                                          \
        }                                                                                \
    } while (0)

#define NND_LAUNCH_CHECK()                                                               \
    do {                                                                                 \
        hipError_t _e = hipGetLastError();                                               \
        if (_e != hipSuccess) {                                                          \
            nnd::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
            return NND_ERR_HIP;                                                          \
        }                                                                                \
    } while (0)

#define NND_REQUIRE(cond, ...)                                                           \
    do {                                                                                 \
        if (!(cond)) {                                                                   \
            nnd::set_error(__VA_ARGS__);                                                 \
            return NND_ERR_INVALID;                                                      \
        }                                                                                \
    } while (0)

// Diagnostic / tuning switches (NND_* environment variables, listed in include/nndepth_amd.h): read ONCE when the library is
// loaded and again only by nnd_reload_switches(); the hot path never calls getenv.  They select between kernels that the parity
// tests prove equivalent, never a non-HIP path.
struct Switches {
    bool no_fused_upsample, no_fused_lookup, no_fused_flow_branch, no_c4, agcl_v1, no_thin3d, corr_build_v1, corr_build_no_ksplit, igev_squeeze_v1, igev_squeeze_walk, no_folded_flow_head, no_conv1x1_stream,
        conv_verbose, debug_sync;
    int split_ny, split_ks, split_p;  // NND_SPLIT_CFG=ny,ks[,P] (<= 0: the picker decides)
    bool split_no_fast;               // NND_SPLIT_NO_FAST: the generic conv_split kernel also where the FAST regime applies
    bool no_merged_fb_lookup;         // NND_NO_MERGED_FB_LOOKUP: flow branch and lookup + convc1 as two launches
    int conv_p, conv_ks, conv_wco;    // NND_CONV_CFG=p,ks,wco / NND_CONV_P
    int agcl_pb;                      // NND_AGCL_PB
    bool lds_poison_on;               // NND_DEBUG_LDS_POISON=<pattern>: fill every CU's LDS with the pattern between the update block's launches
    unsigned lds_poison;
    int lds_slack;                    // NND_DEBUG_LDS_SLACK=<bytes>: conv_split launches ask for that much more dynamic LDS (diagnostic)
    bool enc_no_c4;                   // NND_ENC_NO_C4: the encoder's activations planar tile-major also with a split arithmetic
    bool no_slab3d;                   // NND_NO_SLAB3D: thin Conv3d layers on the round-2 formulations (conv_split / thin3d)
    int slab3d_rounds;                // NND_SLAB3D_ROUNDS: depth segments so that the grid is about this many resident sets
};
const Switches& switches();

// Kernels that use more than 64 KB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize raised on the device's copy of
// the kernel: once per (call site = kernel instantiation, device).  `done` is the call site's own static bit mask over device ids.
// (`bytes`: kernels that also declare static __shared__ arrays ask for less than the 160 KB: static + dynamic must fit)
static inline int raise_lds_limit(const void* kern, std::atomic<unsigned>& done, int bytes = 160 * 1024) {
    int dev = 0;
    NND_HIP_CHECK(hipGetDevice(&dev));
    const unsigned bit = 1u << (dev & 31);
    if (!(done.load(std::memory_order_relaxed) & bit)) {
        NND_HIP_CHECK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        done.fetch_or(bit, std::memory_order_relaxed);
    }
    return NND_OK;
}

// ---------------------------------------------------------------------------------------
// fp16x2 activation-range calibration (calib.hip, split_arith.h): an entry point called with NND_FLAG_CALIBRATE opens a
// CalibScope on the calling thread; while it is open every launcher of a kernel that stages fp16x2 activations first measures
// the largest |activation| of what the launch will stage into the layer's slot of the packed blob (calib_amax_*), and
// nnd_*_calibration_finish turns the slots into the layers' activation scales (calib_finish).  Thread-local: a forward running
// on another host thread is not affected.
// ---------------------------------------------------------------------------------------
bool calibrating();
struct CalibScope {
    explicit CalibScope(bool on);
    ~CalibScope();
    CalibScope(const CalibScope&) = delete;
    CalibScope& operator=(const CalibScope&) = delete;
    bool on;
};
struct Act;
struct Lay;
// `tail`: the layer's 4-float slot (SPLIT_TAIL_*) inside the packed blob on the device (written: the AMAX element)
int calib_amax_act(const Act& a, const Lay& lay, int B, int H, int W, const float* tail, hipStream_t s);
int calib_amax_flat(const float* x, int64_t n, const float* tail, hipStream_t s);
// turns the accumulated maxima of `n` layers (float offsets of their tails inside `blob`) into activation scales; *status_dev
// (optional, device int32) |= 1 if a maximum was inf / NaN (the calibration forward itself overflowed: that layer's scale is lowered
// by 2^12, calibrate again), |= 2 if a layer staged nothing (scale unchanged)
int calib_finish(float* blob, const int64_t* tail_offs, int n, int32_t* status_dev, hipStream_t s);

// acc = a * b + acc as ONE scalar v_fmac_f32, whatever the vectorisers would make of the surrounding loop.  Used by the VALU
// convolutions of the update block (flow_branch_body phase 1, flow_head2_kernel): their results depended on what another
// stream's kernel ran on the same CU whenever the compiler had packed their FMAs into v_pk_fma_f32 with operand-half selection
// (round 3).  The stand-alone reproducer of that instruction beside fp16-MFMA waves does NOT miscompute
// (scripts/ubench/pk_fma_hazard.hip, profiles/r04_pk_fma_hazard_ubench.txt), so the mechanism is not identified; what is
// established is that the scalar form never failed, and this makes it a property of the source instead of a compiler flag.
#ifdef __HIPCC__
__device__ __forceinline__ void fmac_scalar(float& acc, float a, float b) { asm("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b)); }
#endif

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------------------
// Implicit-GEMM convolution on the fp32 MFMA (conv_mfma.hip)
// ---------------------------------------------------------------------------------------
enum ConvEpilogue {
    EPI_LINEAR = 0,  // out0 = v
    EPI_RELU = 1,    // out0 = max(v, 0)
    EPI_GRU_ZR = 2,  // co <  hidden: out0[co] = sigmoid(v)                (z)
                     // co >= hidden: out1[co-hidden] = sigmoid(v) * aux0  (r*h)
    EPI_GRU_Q = 3,   // q = tanh(v); out0 = (1 - aux1) * aux0 + aux1 * q   (aux0 = h, aux1 = z); out1 (optional) = copy
    EPI_SCALE = 4,   // out0 = scale * v
    EPI_AFFINE = 5,  // y = acc * cscale[co] + shift[co] (folded norm; shift in the bias slot); flags bit 0: ReLU;
                     // aux0 (optional): y = aux0 + y; flags bit 1: ReLU          (encoder / residual blocks)
    EPI_SIGMOID_RANGE = 6  // out0 = scale * (sigmoid(v) - 0.5) * 2    (CREStereo's learned search offsets, cre_stereo/model.py:158-159)
};

// One convolution layer inside a packed parameter blob.
struct ConvLayer {
    int KH, KW, Cin, Cout, CI_T;  // CI_T: input channels per K-chunk (32, 128 for wide 1x1, 16 for stride 2)
    int stride = 1;               // 1 or 2 ("same" padding K/2; output = ceil(input / stride) for odd K)
    int arith = 0;                // 0: exact fp32 MFMA (conv_mfma.hip); 3: fp32 carried as 3 bf16 pieces on the bf16 MFMA; 2: as 2
                                  //    range-scaled fp16 pieces on the fp16 MFMA (conv_split.hip, split_arith.h; CI_T = 16, weights
                                  //    packed as 16-bit fragments)
    int64_t s_off = -1;           // float offset of the per-channel scale (EPI_AFFINE), ncb*32 floats; -1: none
    int nchunks;                  // ceil(Cin / CI_T)
    int ncb;                      // ceil(Cout / 32) output-channel blocks
    int64_t w_off, b_off;         // float offsets in the blob
    int64_t w_floats() const {
        return arith ? (int64_t)ncb * nchunks * KH * KW * arith * 256  // [cb][chunk][tap][piece] x 64 lanes x 16 B
                     : (int64_t)ncb * nchunks * KH * KW * CI_T * 32;
    }
    int64_t b_floats() const { return (int64_t)ncb * 32 + (arith == 2 ? 4 : 0); }  // fp16x2: + the output scale (split_arith.h)
    double flops(int B, int H, int W) const { return 2.0 * B * H * W * (double)Cout * Cin * KH * KW; }
    int64_t tail_off() const { return b_off + (int64_t)ncb * 32; }  // fp16x2: float offset of the SPLIT_TAIL_* slot (split_arith.h)
};

// A source / destination activation: channel-slice of an NCHW tensor (channel stride = H*W).
struct Act {
    float* ptr;     // first channel of the slice, batch 0
    int64_t bstride;  // floats between batches
    int C;          // channels in the slice
};

struct ConvIO {
    Act src0, src1;      // virtual concat [src0, src1] along channels (src1.C may be 0)
    Act out0, out1;      // see ConvEpilogue
    Act aux0, aux1;
    Act bmap;            // optional per-pixel bias map (replaces the per-channel bias)
    int hidden = 0;
    float scale = 1.f;
    int Hin = 0, Win = 0;  // input size when it differs from the output size (stride 2); 0: same
    int flags = 0;         // EPI_AFFINE flags: 1 ReLU, 2 ReLU after the residual, 4 LeakyReLU with slope `scale`
    bool src_tiled = false;  // layout.h: sources are tile-major (internal workspace) instead of NCHW
    bool dst_tiled = false;  // ... out0/out1/aux0/aux1/bmap
    bool src_c4 = false;     // tile-major with 4 channels interleaved (layout.h); channel counts / slice starts % 4 == 0
    bool dst_c4 = false;
};

// input channels per K-chunk for a layer shape (host packer and kernels must agree)
int conv_ci_t(int KH, int KW, int Cin, int stride = 1, int Cout = 0);

int launch_conv(const ConvLayer& L, const float* blob, const ConvIO& io, int epi,
                int B, int H, int W, hipStream_t stream);

// Host-side packing of one (virtually concatenated along Cout) conv into A-fragment order.
// w[i]: (cout[i], Cin, KH, KW) row-major, b[i]: (cout[i]).
void pack_conv(const ConvLayer& L, int nparts, const float* const* w, const float* const* b,
               const int* cout, float* blob, const int* ci_map = nullptr, int cin_src = 0);

// conv_split.hip: the same convolution with fp32 operands carried as split bf16 pieces on the 16-bit MFMA (L.arith != 0);
// launch_conv / pack_conv dispatch to these.
bool conv_split_supported(int KH, int KW, int Cin, int stride, int arith, int Cout = 0);  // Cout matters for stride 2 only
int launch_conv_split(const ConvLayer& L, const float* blob, const ConvIO& io, int epi, int B, int H, int W, hipStream_t stream);
// motion encoder flow branch in one launch (conv_split.hip): f2 = convf2's split-packed layer, w7t = convf1's weights tap-major
// ([fc*49][128]), b7 its bias
bool flow_branch_supported(const ConvLayer& f2, int fc);
int launch_flow_branch(const ConvLayer& f2, const float* blob, const float* w7t, const float* b7, const float* flow, int64_t fbs,
                       int fc, const ConvIO& io, int B, int H, int W, hipStream_t stream);
void pack_conv_split(const ConvLayer& L, int nparts, const float* const* w, const float* const* b, const int* cout, float* blob,
                     const int* ci_map, int cin_src);

// thin3d.hip: direct fp32 VALU Conv3d for the regulariser's 8- / 16-output-channel layers (depth-major volumes)
bool thin3d_supported(int Cout, int stride);
// slab3d.hip: the regulariser's thin layers (stride 1 and 2) on the 16-bit MFMA (fp16x2), depth-marching with each input slice staged once
bool slab3d_supported(int Cout, int C0, int C1, int stride, int arith);
int64_t slab3d_packed_floats(int Cout, int Ct, int stride);
void slab3d_pack(int Cout, int Ct, int stride, const float* w, const float* scale, const float* shift, float* out);
int slab3d_forward(int Cout, int C0, int C1, int stride, const float* packed, const float* x0, const float* x1, float* y, int N, int D,
                   int H, int W, float slope, hipStream_t s);
int64_t thin3d_packed_floats(int Cout, int Ct);
void thin3d_pack(int Cout, int Ct, const float* w, const float* scale, const float* shift, float* out);
int thin3d_forward(int Cout, int C0, int C1, int stride, const float* packed, const float* x0, const float* x1, float* y, int N, int D,
                   int H, int W, float slope, hipStream_t s);

// corr1d.hip
// `tiled`: coords / sampled features (resp. flow / mask) are tile-major workspace tensors (layout.h), else NCHW
int corr1d_lookup_launch(const float* pyr, const float* coords, float* out, int B, int H, int W, int num_levels,
                         int radius, hipStream_t stream, bool tiled);
int group_lookup_flat_launch(const float* pyr, const float* coords, float* out, int B, int G, int H, int W, int num_levels, int radius,
                             hipStream_t stream, bool tiled);
int igev_lookup_launch(const float* feat_pyramid, const float* geo_pyramid, const float* coords, float* out, int B, int G, int H,
                       int W, int num_levels, int radius, hipStream_t stream, bool tiled);
int convex_upsample_launch(const float* flow, const float* mask, float* out, int B, int C, int H, int W, int rate,
                           hipStream_t stream, bool tiled);

// mask_upsample.hip: fused mask.2 (1x1, x0.25) + softmax + convex upsample (mask never written)
// CREStereo AGCL (agcl.hip).  tiled: `flow` (in) and `out` are tile-major workspace tensors; fmaps / extra / warped are NCHW.
int agcl_iter_launch(const float* f1, const float* f2, const float* flow, float* warped, float* out, int N, int C, int H,
                     int W, int small_patch, hipStream_t s, bool tiled);
int agcl_offset_launch(const float* f1, const float* f2, const float* flow, const float* extra, float* out, int N, int C, int H,
                       int W, int small_patch, hipStream_t s, bool tiled);
int agcl_check(const char* what, int N, int C, int H, int W);
// offset mode on channels-last copies (N, H*W, C) of the maps (C == 256), and the copy itself
bool agcl_offset_cl_supported(int C);
int agcl_offset_cl_launch(const float* f1c, const float* f2c, const float* flow, const float* extra, float* out, int N, int C,
                          int H, int W, int small_patch, hipStream_t s, bool tiled);
int nchw_to_nhwc_launch(const float* in, float* out, int N, int C, int P, hipStream_t s);
int lookup_convc1_launch(const float* pyr, const float* geo, int G, const float* coords, const ConvLayer& L, const float* blob,
                         float* c1, int64_t c1_bs, int B, int H, int W, int num_levels, int radius, hipStream_t stream, bool c1_c4);
// the flow branch and lookup + convc1 of one iteration as ONE launch of two kinds of workgroups (corr1d.hip: flow_branch_lookup_kernel)
bool flow_branch_lookup_supported(int arith);
int flow_branch_lookup_launch(const ConvLayer& f2, const float* blob, const float* w7t, const float* b7, const float* flow, int64_t fbs,
                              int fc, const ConvIO& io, const float* pyr, const float* coords, const ConvLayer& Lc1, float* c1,
                              int64_t c1_bs, int B, int H, int W, int num_levels, int radius, hipStream_t stream, bool c1_c4);
bool igev_lookup_convc1_il_supported(int G, int num_levels, int radius);
int igev_lookup_convc1_il_launch(const float* il, int G, const float* coords, const ConvLayer& L, const float* blob, float* c1,
                                 int64_t c1_bs, int B, int H, int W, int num_levels, int radius, hipStream_t stream, bool c1_c4);
// flow_head.conv2 (3x3, hid -> 1) and the recurrence update of raft_stereo/model.py:134-135 folded into this kernel (round 4; fc == 1):
// the workgroup needs the NEW flow on the 6x10 patch around its tile, so it computes delta = conv2(relu(conv1(h))) on those 60
// positions itself — from the 8x12 patch of the flow head's hidden map, with flow_head2_kernel's arithmetic (16 slices of hid/16
// channels x 9 taps with v_fmac_f32 each, summed in slice order, + bias: the same bits) — and adds it to the OLD coordinate.  Old
// and new state live in two buffer pairs that the caller swaps every iteration (a neighbour workgroup may already have advanced
// the pixels of this one's halo); the 32 pixels of the tile itself are written: new coordinate, new flow (also into the GRU's
// input tensor) and delta.  Four extra waves do it beside the mask GEMM (fp16x2 kernels): the launch it saves is 8.7 us, 2-3 us of them
// are gained — on grids of at most two workgroups per CU; on larger ones the separate launch is as fast or faster and is kept.
struct MaskUpFlowHead {
    const float* x;       // flow head's hidden map relu(flow_head.conv1(h)): hid channels, layout `lay` / x_c4 of the mask input
    long xbs;
    int hid;
    const float* w;       // (1, hid, 3, 3) and bias (1), as flow_head2_kernel reads them
    const float* bias;
    const float* coords_in;  // old state, (B,1,H,W) in `lay`
    float *coords_out, *flow_out, *delta_out;
    float* hx_flow;       // flow channel of the GRU input tensor
    long hx_bs;
    int hx_pm;            // floats between consecutive pixels there (4: c4)
    int absolute;         // IGEV: the state handed on is the coordinate itself
};

bool mask_upsample_supported(int rate, int cin, int flow_channels);
bool mask_upsample_fold_supported(const ConvLayer& L, int hid);
int mask_upsample_launch(const ConvLayer& L, const float* blob, const float* x, int64_t xbs, const float* flow, float* out,
                         int B, int H, int W, int rate, hipStream_t stream, bool tiled, int flow_channels = 1, bool x_c4 = false,
                         const MaskUpFlowHead* fh = nullptr);

}  // namespace nnd
