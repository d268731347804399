/*
 * nndepth_amd — C-ABI of the MI355X (gfx950) stereo-disparity hot path.
 *
 * The reference (anhtu293/nndepth) is pure Python/PyTorch and has no FFI layer; its seams
 * for this path are three duck-typed Python objects on the model instance.
 * Each entry point below replaces the ATen-op composition behind one of those seams and
 * cites the reference lines it stands in for (paths relative to the reference root).
 *
 * Conventions
 *   - all tensors are fp32, NCHW, contiguous, resident in device (HBM) memory unless a
 *     parameter says "host";
 *   - the caller owns every buffer including workspaces — the library never allocates or
 *     frees device memory and never synchronises;
 *   - all work is enqueued on the caller-supplied hipStream_t (`stream`, may be NULL);
 *   - return value: 0 on success, negative nnd_status on failure (never throws);
 *     nnd_last_error() gives a thread-local message for the last failure;
 *   - threading: every entry point may be called concurrently from several host threads as long as the calls
 *     use distinct streams and distinct output / workspace buffers.  The library keeps no mutable global state:
 *     every kernel of a call — the fused loops' too — is enqueued on the caller's `stream` and nowhere else
 *     (rounds 1-2 ran one branch of the loop on an internal side stream; measured equal, removed);
 *   - diagnostic environment switches (read ONCE when the library is loaded and again only by nnd_reload_switches(), never
 *     on the hot path; they select between kernels that the parity tests prove equivalent, never a non-HIP path; none of them
 *     changes the layout of a packed blob — which convolutions of the update block take a split arithmetic is part of the
 *     descriptor, nnd_update_block_desc.split_layers): NND_NO_FUSED_UPSAMPLE (mask.2 and convex upsample as two launches), NND_NO_FOLDED_FLOW_HEAD (flow_head.conv2 + the recurrence update as their own launch in front of the fused mask / upsample launch instead of inside it),
 *     NND_NO_FUSED_LOOKUP (lookup and convc1 as two launches), NND_NO_MERGED_FB_LOOKUP (flow branch and lookup + convc1 as two launches instead of one launch of two kinds of workgroups), NND_DEBUG_LDS_POISON / NND_DEBUG_LDS_SLACK (diagnostics of the round-3 reproducibility study: pattern-fill every CU's LDS between the loop's launches / ask for more dynamic LDS), NND_SPLIT_NO_FAST (the generic conv_split kernel wherever one exists: every stride-1 shape; the stride-2 kernels are FAST-only and keep running), NND_NO_FUSED_FLOW_BRANCH (convf1 and convf2 as two launches when
 *     arithmetic = 3), NND_SPLIT_CFG (force the workgroup shape of the split kernel), NND_NO_CONV1X1_STREAM (1x1 shortcuts through the
 *     staged conv kernel), NND_CORR_BUILD_V1 (register-operand correlation build), NND_CORR_BUILD_NO_KSPLIT (small grids through the LDS-staged build instead of the k-split one), NND_IGEV_SQUEEZE_V1 / NND_IGEV_SQUEEZE_WALK (cv_squeezer + soft-argmin: the one-row kernel always / the row-walking kernel whenever it can run), NND_AGCL_V1 (one-pixel-per-lane AGCL kernels), NND_AGCL_PB (pixels per workgroup of the channels-last offset kernel), NND_CONV_CFG / NND_CONV_P
 *     (force a tile configuration), NND_CONV_VERBOSE (print the chosen configuration), NND_DEBUG_SYNC
 *     (synchronise and name every launch of the update block on stderr), NND_NO_THIN3D (the regulariser's 8- / 16-channel
 *     Conv3d layers through the MFMA formulation instead of csrc/thin3d.hip), NND_NO_SLAB3D (the same layers at stride 1 with
 *     arithmetic = 2 through the round-2 formulations instead of the depth-marching MFMA kernel csrc/slab3d.hip),
 *     NND_SLAB3D_ROUNDS (depth segments of that kernel: grid of about this many resident sets), NND_NO_C4 (planar instead of 4-channel-
 *     interleaved layout of the update block's conv-only workspace tensors), NND_ENC_NO_C4 (the same for the encoder's activations
 *     with a split arithmetic).
 */
#ifndef NNDEPTH_AMD_H
#define NNDEPTH_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NND_VERSION 104 /* 0.1.4: 102 descriptors carry struct_size and flags, per-layer fp16x2 activation scales + calibration; 103 group-RAFT entry points; 104 nnd_profile_mfma16_peak */

/* Descriptors start with `struct_size` = sizeof(the descriptor type) of the header the caller was compiled against; every entry
 * point that takes one refuses another size (NND_ERR_INVALID), so a caller and a library of different versions cannot
 * misread each other's fields.  `flags` is a bit set of NND_FLAG_*.                                                        */
#define NND_FLAG_CALIBRATE 1 /* fp16x2 only: besides its normal work, the call records — for every layer it runs in that
                                arithmetic — the largest |activation| the layer stages, in the layer's slot of the packed blob
                                (`packed_dev` is WRITTEN by such a call; the extra kernels run on `stream` like everything
                                else).  nnd_*_calibration_finish then fixes the layers' activation scales.  See "fp16x2
                                activation range" below.                                                                    */

typedef enum {
    NND_OK = 0,
    NND_ERR_INVALID = -1,     /* bad argument / unsupported shape */
    NND_ERR_HIP = -2,         /* a HIP runtime call or launch failed */
    NND_ERR_UNSUPPORTED = -3, /* configuration not built */
    NND_ERR_NO_DEVICE = -4
} nnd_status;

int nnd_version(void);
const char* nnd_last_error(void);
/* number of visible HIP devices (0 when none / on a CPU-only host); never raises */
int nnd_device_count(void);
/* re-reads the NND_* diagnostic switches from the environment (tests and tuning scripts that toggle them in-process);
 * not to be called concurrently with other entry points */
int nnd_reload_switches(void);

/* ------------------------------------------------------------------ correlation pyramid
 * Replaces CorrBlock1D.__init__ + CorrBlock1D.corr
 *   nndepth/models/raft_stereo/cost_volume.py:12-34,55-61
 * level 0: corr[b,h,w1,w2] = sum_c f1[b,c,h,w1]*f2[b,c,h,w2] / sqrt(C);
 * level i+1 = avg_pool1d(level i, 2) (floor on odd widths).  `num_levels`+1 levels are
 * stored (the reference builds one more than it reads).
 * Layout of `pyramid`: level l is a dense (B*H*W, W_l) row-major matrix starting at float
 * offset level_offsets[l] (W_0 = W, W_{l+1} = W_l / 2).                                  */
int nnd_corr1d_pyramid_layout(int B, int H, int W, int num_levels,
                              int64_t* level_offsets /* [num_levels+1] */,
                              int32_t* level_widths /* [num_levels+1] */,
                              int64_t* total_floats);
int nnd_corr1d_build(const float* fmap1, const float* fmap2, float* pyramid,
                     int B, int C, int H, int W, int num_levels, void* stream);

/* Replaces CorrBlock1D.__call__ + linear_sampler
 *   nndepth/models/raft_stereo/cost_volume.py:36-53, nndepth/models/raft_stereo/utils.py:4-27
 * coords (B,1,H,W) -> out (B, num_levels*(2*radius+1), H, W); border-clamped (Q2).        */
int nnd_corr1d_lookup(const float* pyramid, const float* coords, float* out,
                      int B, int H, int W, int num_levels, int radius, void* stream);

/* ------------------------------------------------------------ IGEV geometry-encoding volume
 * Replaces GeometryAwareCostVolume.build_cost_volume + the avg-pool pyramids + forward (combined lookup)
 *   nndepth/models/igev_stereo/cost_volume.py:81-98, :40-52, :54-79
 * Group g correlates channels [g*group_channels, (g+1)*group_channels) of the Ctot-channel maps (the reference
 * uses only the first num_groups chunks of num_groups channels) and divides by sqrt(group_channels).
 * Pyramids use the layout of nnd_corr1d_pyramid_layout(B*num_groups, H, W, num_levels): rows ordered (b,g,h,w1),
 * so level 0 viewed as (B,G,H,W1,W2) is the tensor the 3-D regulariser (nnd_conv3d_*) consumes.
 * num_levels = 0 writes level 0 alone (the fused refinement loop reads the interleaved copy below, which pools for itself).
 * nnd_pyramid_from_level0 fills levels 1..num_levels of a pyramid whose level 0 was written by the caller (the
 * regularised volume, or a feature volume built with num_levels = 0).  nnd_igev_lookup: coords (B,1,H,W) -> out (B, num_levels*2*G*(2r+1), H, W), channel
 * = i*2*G*T + v*G*T + g*T + k with v = 0 feature / 1 geometry volume.                                   */
int nnd_group_corr_build(const float* fmap1, const float* fmap2, float* pyramid, int B, int Ctot, int H, int W,
                         int num_groups, int group_channels, int num_levels, void* stream);
/* GroupCorrBlock1D of Coarse2FineGroupRepViTRAFTStereo (nndepth/models/raft_stereo/cost_volume.py:64-128; widening):
 * nnd_group_corr_build_scaled = nnd_group_corr_build with the divisor given by the caller — GroupCorrBlock1D.corr (:115-128) splits the
 * maps into chunks of `num_groups` channels, correlates the first `num_groups` chunks and divides by sqrt(C_total):
 * group_channels = num_groups, divisor = sqrt(Ctot).
 * nnd_group_corr1d_lookup = GroupCorrBlock1D.__call__ (:94-113) INCLUDING its view without the group permute (Q6): coords (B,1,H,W) ->
 * out (B, num_levels*G*(2r+1), H, W); pixel (y,x), channel i*G*T + j holds sample j % T of row (y*W + x)*G + j/T of the sample's
 * G*H*W (group, pixel) rows at level i, each row sampled at ITS OWN pixel's coordinate.                                        */
int nnd_group_corr_build_scaled(const float* fmap1, const float* fmap2, float* pyramid, int B, int Ctot, int H, int W,
                                int num_groups, int group_channels, int num_levels, float divisor, void* stream);
int nnd_group_corr1d_lookup(const float* pyramid, const float* coords, float* out, int B, int G, int H, int W, int num_levels,
                            int radius, void* stream);
int nnd_pyramid_from_level0(float* pyramid, int B, int H, int W, int num_levels, void* stream);
int nnd_igev_lookup(const float* feat_pyramid, const float* geo_pyramid, const float* coords, float* out,
                    int B, int G, int H, int W, int num_levels, int radius, void* stream);
/* Group-interleaved copy of the levels 0..num_levels-1 of both pyramids, the layout the refinement loop gathers from:
 * per level  il[((b*H*W + h*W + w1) * w2 + x) * 2G + v*G + g]  (v = 0 feature / 1 geometry volume), so that the
 * 2G*(2r+1) samples of GeometryAwareCostVolume.forward (igev_stereo/cost_volume.py:54-79) for one pixel and level are
 * one contiguous run instead of 2G rows.  Written once per pair; nnd_igev_interleaved_floats = its size.           */
int64_t nnd_igev_interleaved_floats(int B, int G, int H, int W, int num_levels);
int nnd_igev_interleave_pyramids(const float* feat_pyramid, const float* geo_pyramid, float* interleaved,
                                 int B, int G, int H, int W, int num_levels, void* stream);
/* The same interleaved levels from the two level-0 volumes alone (rows (b,g,h,w1) of W floats each): the avg_pool1d cascade
 * of cost_volume.py:46-52 runs in LDS with the pyramid's arithmetic (bit-identical to pooling first), so neither pyramid's
 * pooled levels have to exist.  _supported: 2*G <= 32, every level at least 1 wide, one pixel's levels within 160 KB of LDS. */
int nnd_igev_interleave_level0_supported(int G, int W, int num_levels);
int nnd_igev_interleave_level0(const float* feat_level0, const float* geo_level0, float* interleaved,
                               int B, int G, int H, int W, int num_levels, void* stream);

/* ------------------------------------------------- CREStereo adaptive group correlation (AGCL)
 * Replaces AGCL.corr_iter / get_correlation / corr_att_offset and bilinear_sampler / bilinear_grid_sample
 *   nndepth/models/cre_stereo/cost_volume.py:28-79, :81-154;  nndepth/models/cre_stereo/utils.py:5-20,34-107
 * Channels are split into 4 groups of C/4; the 9 search positions are a 1x9 window (small_patch = 0) or a 3x3
 * window (small_patch = 1, dy outer / dx inner); out (N,36,H,W), channel = g*9 + k.
 * nnd_bilinear_sample : img (N,C,H,W), coords (N,Hg,Wg,2) = (x,y) in pixels -> out (N,C,Hg,Wg); taps outside the
 *                       image read zero (grid_sample zeros/align_corners=True semantics incl. the reference's
 *                       pixel -> [-1,1] -> pixel round trip).
 * nnd_agcl_corr_iter  : flow (N,2,H,W); the right features are warped by grid+flow into `warped` (caller-owned
 *                       scratch, N*C*H*W floats), then correlated over the window of the replicate-padded warped map.
 * nnd_agcl_corr_offset: each search position samples fmap2 at grid + flow + window + extra_offset, extra_offset
 *                       (N,18,H,W) with channel 2k = x, 2k+1 = y of position k.  The optional cross attention of the
 *                       reference (att != None) is applied by the caller to fmap1/fmap2 beforehand.             */
int nnd_bilinear_sample(const float* img, const float* coords, float* out, int N, int C, int H, int W, int Hg, int Wg,
                        void* stream);
int nnd_agcl_corr_iter(const float* fmap1, const float* fmap2, const float* flow, float* warped, float* out,
                       int N, int C, int H, int W, int small_patch, void* stream);
int nnd_agcl_corr_offset(const float* fmap1, const float* fmap2, const float* flow, const float* extra_offset, float* out,
                         int N, int C, int H, int W, int small_patch, void* stream);
/* The same on channels-last copies (N,H,W,C) of the two maps, C == 256 (the maps are constant over the iterations of a
 * cascade stage, the copy is made once): one wave per pixel, every bilinear tap is one 1-KB line.  Sums the 64 channels
 * of a group in a different order than nnd_agcl_corr_offset (<= 1e-7 apart).  nnd_nchw_to_nhwc makes the copy.        */
int nnd_agcl_corr_offset_nhwc(const float* fmap1_nhwc, const float* fmap2_nhwc, const float* flow, const float* extra_offset,
                              float* out, int N, int C, int H, int W, int small_patch, void* stream);
int nnd_nchw_to_nhwc(const float* in, float* out, int N, int C, int H, int W, void* stream);

/* IGEV initial disparity: regress_disparity(softmax over the candidate axis)
 *   nndepth/models/igev_stereo/model.py:92-95,145-146
 * logits (B,D,H,W) = the squeezed geometry volume -> out (B,1,H,W) = -sum_d d * softmax_d(logits).              */
int nnd_softargmin_disparity(const float* logits, float* out, int B, int D, int H, int W, void* stream);
/* cv_squeezer + regress_disparity in one pass (nndepth/models/igev_stereo/model.py:63,144-146): geo_level0 = level 0 of
 * the geometry pyramid, rows (b,g,h,w1) of D floats (GeometryAwareCostVolume.geo_aware_cv[0]); weight = the
 * Conv3d(G,1,3,1,1) kernel (1,G,3,3,3) over (candidate, h, w1), bias (1) or NULL — both HOST pointers (they travel as
 * kernel arguments); out (B,1,H,W) = -sum_d d * softmax_d(conv3d(geo)).  G <= 8, D <= 512, else NND_ERR_UNSUPPORTED. */
int nnd_igev_init_disparity(const float* geo_level0, const float* weight_host, const float* bias_host, float* out,
                            int B, int G, int H, int W, int D, void* stream);

/* ----------------------------------------------------------------------- convex upsample
 * Replaces RAFTStereo.convex_upsample  nndepth/models/raft_stereo/model.py:93-105
 * (IGEV copy igev_stereo/model.py:103-115; 2-channel CRE copy cre_stereo/model.py:110-122)
 * flow (B,C,H,W), mask (B,9*rate*rate,H,W) -> out (B,C,rate*H,rate*W).                    */
int nnd_convex_upsample(const float* flow, const float* mask, float* out,
                        int B, int C, int H, int W, int rate, void* stream);

/* -------------------------------------------------------------------------- update block
 * Replaces BasicUpdateBlock.forward (+BasicMotionEncoder, SepConvGRU/ConvGRU, FlowHead, mask head)
 *   nndepth/blocks/update_block.py:57-65,26-36,97-112 ; nndepth/blocks/gru.py:22-37,53-61   */
/* fp16x2 activation range (arithmetic = 2; csrc/split_arith.h, csrc/calib.hip).  An fp32 activation x is carried as two fp16
 * pieces of x * 2^xs with xs PER LAYER, stored in the packed blob and read by the kernels at run time.
 *   - a freshly packed blob has xs = 2 for every layer: all 22 significand bits for 0.06 <= |x| < 16376, graceful below
 *     (absolute error <= 2^-27), inf / NaN in the output above (never a silently wrong value);
 *   - after a calibration — one or more forwards with NND_FLAG_CALIBRATE on representative inputs, then
 *     nnd_*_calibration_finish — each layer has xs = 11 - ceil(log2(M)), M = the largest |activation| that layer staged
 *     during those forwards: all 22 bits for M * 2^-13 <= |x| <= M, absolute error <= M * 2^-36 below, valid (finite) up to
 *     |x| < 32 * M, inf / NaN in the output beyond;
 *   - nnd_update_block_scale_slots gives the float offsets of the layers' slots in the blob: slot[1] = 2^xs, so the valid range
 *     of layer i is |x| < 65504 / slot[1] (the Python engines expose it as `activation_ranges()`).
 * The model classes of nndepth_amd calibrate on their first forward (one extra forward, once).                              */
typedef struct {
    int32_t struct_size;   /* sizeof(nnd_update_block_desc) */
    int32_t hidden_dim;    /* 128 */
    int32_t context_dim;   /* 64 (YAML) or 128 (class default) */
    int32_t cor_planes;    /* num_levels*(2r+1) = 36; 576 for IGEV */
    int32_t flow_channels; /* 1 (RAFT/IGEV) or 2 (CRE) */
    int32_t mask_channels; /* 9*rate*rate: 576 (/8) or 144 (/4) */
    int32_t gru_kind;      /* 0 = "sep_conv" (1x5 then 5x1), 1 = "conv_gru" (3x3) */
    int32_t arithmetic;    /* MFMA convolutions of the update block except convc1 (fused with the lookup, exact fp32): mask.2 runs
                              inside the fused mask + upsample kernel and convf1 -> convf2 as one launch in the same arithmetic.
                              0 = exact fp32 (v_mfma_f32_32x32x2_f32); 3 = every fp32 operand carried as 3 bf16 pieces, the 6
                              products x_i*w_j (i + j <= 2) on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (dropped terms
                              <= 2^-24 |x||w|): per-op error vs float64 below the exact kernel's, RAFT-Stereo 544x960 / 32
                              iterations vs the reference 4.3e-5 (exact: 3.6e-5); 2 = every fp32 operand carried as 2 fp16
                              pieces (22 significand bits), the 3 products x0*w0, x0*w1, x1*w0 on v_mfma_f32_32x32x16_f16 with
                              fp32 accumulation — half the matrix work of 3; both operands are range-scaled by exact powers of two
                              (activations x4 while staged, weights per layer at pack time) that the kernel undoes after the K
                              loop, so the low pieces stay out of fp16's subnormals; the activation scale is per layer and
                              calibrated from data, see "fp16x2 activation range" above; csrc/split_arith.h.
                              The packed blob is specific to the value. */
    int32_t split_layers;  /* 0: every convolution the split kernels are built for takes `arithmetic`; else bit i = convolution i
                              (nnd_conv_name) may take it, the others stay exact fp32 (diagnostic: bisecting a difference).  Part
                              of the blob layout, like `arithmetic`: use the same value to pack and to run. */
    int32_t flags;         /* NND_FLAG_* */
} nnd_update_block_desc;

/* Number of weight/bias tensors expected by nnd_update_block_pack, in the order of the
 * reference module's state_dict(): encoder.{convc1,convc2,convf1,convf2,conv},
 * gru.{convz1,convr1,convq1[,convz2,convr2,convq2]}, flow_head.{conv1,conv2}, mask.{0,2};
 * each as (weight, bias).  30 for sep_conv, 24 for conv_gru.                              */
int nnd_update_block_num_tensors(const nnd_update_block_desc* desc);
/* size (floats) of the packed parameter blob */
int64_t nnd_update_block_packed_floats(const nnd_update_block_desc* desc);
/* HOST function (no GPU needed): re-orders PyTorch-layout (Cout,Cin,KH,KW) fp32 weights into
 * the MFMA A-fragment order the kernels stream (see DESIGN.md "packed weights").
 * `tensors_host`: array of host pointers in the order above; `packed_host`: output.       */
int nnd_update_block_pack(const nnd_update_block_desc* desc, const float* const* tensors_host,
                          float* packed_host);
/* workspace (floats) needed by nnd_update_block_forward / nnd_raft_stereo_refine          */
int64_t nnd_update_block_workspace_floats(const nnd_update_block_desc* desc, int B, int H, int W);
/* fp16x2 calibration (see "fp16x2 activation range"): after one or more calls of nnd_update_block_forward / nnd_*_refine with
 * NND_FLAG_CALIBRATE on this blob, turns the recorded maxima into the layers' activation scales (one tiny kernel on `stream`, no
 * synchronisation) and clears the records.  status_dev: optional device int32 that receives |= 1 if a recorded maximum was
 * inf / NaN (the calibration forward itself overflowed at the old scale: that layer's scale was lowered by 2^12 — calibrate
 * again), |= 2 if some fp16x2 layer of the blob staged nothing (not on the path of those calls: its scale is unchanged).
 * No-op for the other arithmetics.
 * nnd_update_block_scale_slots: float offsets inside the blob of the 4-float slots {2^-(s+xs), 2^xs, record, 2^-s} of the
 * convolutions 0 .. n-1 in nnd_conv_name order (-1 where a convolution is not fp16x2); returns how many convolutions the
 * descriptor has (call with offsets = NULL to size the array).                                                            */
int nnd_update_block_calibration_finish(const nnd_update_block_desc* desc, float* packed_dev, int32_t* status_dev, void* stream);
int nnd_update_block_scale_slots(const nnd_update_block_desc* desc, int64_t* offsets, int n);

/* (net, inp, corr, flow) -> (net_out, mask_out, delta_out); same shapes as the reference:
 * net (B,hidden,H,W) inp (B,context,H,W) corr (B,cor_planes,H,W) flow (B,fc,H,W)
 * mask_out (B,mask_channels,H,W) [already x0.25]  delta_out (B,fc,H,W).
 * mask_out may be NULL (mask head skipped).                                               */
int nnd_update_block_forward(const nnd_update_block_desc* desc, const float* packed_dev,
                             const float* net, const float* inp, const float* corr, const float* flow,
                             float* net_out, float* mask_out, float* delta_out,
                             float* workspace, int B, int H, int W, void* stream);

/* ------------------------------------------------------------------- generic convolution
 * The implicit-GEMM fp32-MFMA convolution the update block is made of, exposed on its own:
 * stride 1, zero "same" padding (KH/2, KW/2), kernels 1x1, 3x3, 1x5, 5x1.  Stands in for one
 * nn.Conv2d (+ optional ReLU) call, e.g. nndepth/blocks/update_block.py:58.
 * nnd_conv2d_pack is a HOST function: w (Cout,Cin,KH,KW), b (Cout) -> packed blob.            */
int64_t nnd_conv2d_packed_floats(int Cout, int Cin, int KH, int KW);
int nnd_conv2d_pack(const float* w_host, const float* b_host, int Cout, int Cin, int KH, int KW,
                    float* packed_host);
int nnd_conv2d_forward(const float* packed_dev, const float* x, float* y, int B, int Cin, int H, int W,
                       int Cout, int KH, int KW, int relu, void* stream);
/* The same with the arithmetic chosen (see nnd_update_block_desc.arithmetic): 0 = exact fp32 MFMA (identical to the
 * functions above), 3 = fp32 operands as 3 bf16 pieces on the bf16 MFMA, 2 = as 2 range-scaled fp16 pieces on the fp16 MFMA
 * (csrc/conv_split.hip; both need Cin % 16 == 0).
 * The packed blob is specific to the arithmetic it was packed for.                                                  */
int64_t nnd_conv2d_packed_floats_ex(int Cout, int Cin, int KH, int KW, int arithmetic);
int nnd_conv2d_pack_ex(const float* w_host, const float* b_host, int Cout, int Cin, int KH, int KW, int arithmetic,
                       float* packed_host);
int nnd_conv2d_forward_ex(const float* packed_dev, const float* x, float* y, int B, int Cin, int H, int W,
                          int Cout, int KH, int KW, int relu, int arithmetic, void* stream);
/* fp16x2 calibration of such a single layer (arithmetic == 2; no-op otherwise): sets the blob's activation scale from the
 * largest |x| of this input (see "fp16x2 activation range").  status_dev as nnd_update_block_calibration_finish.          */
int nnd_conv2d_calibrate_ex(float* packed_dev, const float* x, int B, int Cin, int H, int W, int Cout, int KH, int KW,
                            int arithmetic, int32_t* status_dev, void* stream);

/* conv + CREStereo's search-offset activation in the epilogue: y = range * (sigmoid(conv(x)) - 0.5) * 2
 *   nndepth/models/cre_stereo/model.py:158-159,171-172 (conv_offset_16 / conv_offset_8); blob of nnd_conv2d_pack.      */
int nnd_conv2d_offset_forward(const float* packed_dev, const float* x, float* y, int B, int Cin, int H, int W,
                              int Cout, int KH, int KW, float range, void* stream);

/* ------------------------------------------------------------- glue operators of the model forwards (csrc/cascade.hip)
 * nnd_split_tanh_relu    : x (B, Cnet+Cinp, H, W) -> net = tanh(x[:, :Cnet]), inp = relu(x[:, Cnet:])
 *                          nndepth/models/raft_stereo/model.py:119-122 (igev_stereo/model.py:129-131, cre_stereo/model.py:148-151)
 * nnd_avg_pool_2x_4x     : out2 = F.avg_pool2d(x, 2, stride=2) (N,C,H/2,W/2) and out4 = F.avg_pool2d(x, 4, stride=4)
 *                          (N,C,H/4,W/4) in one pass                           cre_stereo/model.py:154-177
 * nnd_resize_bilinear_ac : y (N,C,H,W) = mul * F.interpolate(x (N,C,h,w), (H,W), mode="bilinear", align_corners=True)
 *                          cre_stereo/model.py:205-212,235-241,259-265 (flow hand-over between cascade stages)
 * nnd_pos_enc_sine_add   : y = x + pe[:, :, :H, :W], PositionEncodingSine.forward  nndepth/blocks/pos_enc.py:22-42 (used at
 *                          cre_stereo/model.py:180-196): the table is generated on the fly, pe[4k+j] = sin / cos (j odd) of
 *                          pos * exp(2k * rate), pos = column + 1 (j < 2) or row + 1, rate = the reference's
 *                          `-log(1e4) / d_model // 2` with Python's precedence (= -1; temp_bug_fix != 0: -log(1e4) / (d_model // 2)).
 *                          x1 / y1 (a second map of the same shape, may both be NULL) get the same table in the same launch.    */
int nnd_split_tanh_relu(const float* x, float* net, float* inp, int B, int Cnet, int Cinp, int H, int W, void* stream);
int nnd_pos_enc_sine_add(const float* x0, const float* x1, float* y0, float* y1, int N, int C, int H, int W, int temp_bug_fix,
                         void* stream);
int nnd_avg_pool_2x_4x(const float* x, float* out2, float* out4, int N, int C, int H, int W, void* stream);
int nnd_resize_bilinear_ac(const float* x, float* y, int N, int C, int h, int w, int H, int W, float mul, void* stream);

/* Convolution + folded eval-mode BatchNorm + ReLU / residual epilogue (the building block of the encoder):
 *   y = conv(x; w, stride, "same" padding K/2) ; y = (y + bias - mean) * gamma / sqrt(var + eps) + beta   [norm optional]
 *   if relu: y = max(y, 0);  if residual: y = residual + y;  if relu_after_residual: y = max(y, 0)
 * Stands in for nn.Conv2d [+ nn.BatchNorm2d(eval)] [+ ReLU] and the `relu(x + y)` of
 * nndepth/blocks/residual_block.py:53-60.  Kernels 1x1 / 3x3 at stride 1 or 2 (1x5 / 5x1 at stride 1);
 * x (B,Cin,Hin,Win) -> y (B,Cout,ceil(Hin/stride),ceil(Win/stride)), residual like y.  nnd_conv_pack is a HOST function
 * (the norm is folded in double precision into a per-channel scale / shift); bn_* may be NULL (no norm).          */
typedef struct nnd_conv_desc {
    int Cout, Cin, KH, KW, stride;
} nnd_conv_desc;
int64_t nnd_conv_packed_floats(const nnd_conv_desc* desc);
int nnd_conv_pack(const nnd_conv_desc* desc, const float* w_host, const float* bias_host, const float* bn_gamma,
                  const float* bn_beta, const float* bn_mean, const float* bn_var, float bn_eps, float* packed_host);
int nnd_conv_forward(const nnd_conv_desc* desc, const float* packed_dev, const float* x, const float* residual, float* y,
                     int B, int Hin, int Win, int relu, int relu_after_residual, void* stream);

/* ------------------------------------------------------------------------- feature encoder
 * Replaces BasicEncoder.forward  nndepth/encoders/basic_encoder.py:71-93 (norm_fn "batch" in eval mode, or "none"),
 * ResidualBlock.forward nndepth/blocks/residual_block.py:53-60 and, optionally, the context projection
 * `cnet_proj` of nndepth/models/raft_stereo/model.py:53-55 applied to the first n_cnet maps (the left frames).
 * norm: 0 = none, 1 = BatchNorm2d (running statistics, folded at pack time), 2 = InstanceNorm2d(affine=False) (CREStereo,
 * cre_stereo/model.py:70-72; per-sample statistics computed on the device).  cnet_dim: channels of cnet_proj, 0 = absent.
 * nnd_encoder_pack (HOST): `tensors` = units of 6 pointers {weight, bias, norm weight, norm bias, running_mean,
 * running_var} (the last four NULL where there is no norm) in the order: conv1 | for each residual block layer1.0,
 * layer1.1, layer2.0, layer2.1, layer3.0, layer3.1: conv1, conv2, downsample.0 (its norm = norm3) | conv2 | cnet_proj.0.
 * nnd_encoder_forward: frames (N,3,H,W) -> fmap (N,output_dim,H/8,W/8); cnet_out (n_cnet,cnet_dim,H/8,W/8) or NULL.
 * workspace: nnd_encoder_workspace_floats(desc, N, H, W) floats, caller-owned.                                     */
typedef struct nnd_encoder_desc {
    int32_t struct_size; /* sizeof(nnd_encoder_desc) */
    int32_t output_dim, norm, cnet_dim;
    int32_t arithmetic; /* 0 = exact fp32 MFMA; 3 / 2 = the 3x3 convolutions at stride 1 and 2, the stride-2 1x1 shortcuts and
                           cnet_proj on the 16-bit MFMA with 3 bf16 / 2 fp16 split operands (see nnd_update_block_desc.arithmetic);
                           the stem, the stride-1 1x1 shortcuts and the 1x1 output conv stay exact fp32 */
    int32_t flags;      /* NND_FLAG_* */
} nnd_encoder_desc;
int nnd_encoder_num_tensors(const nnd_encoder_desc* desc);
int64_t nnd_encoder_packed_floats(const nnd_encoder_desc* desc);
int64_t nnd_encoder_workspace_floats(const nnd_encoder_desc* desc, int N, int H, int W);
int nnd_encoder_pack(const nnd_encoder_desc* desc, const float* const* tensors_host, float bn_eps, float* packed_host);
int nnd_encoder_forward(const nnd_encoder_desc* desc, const float* packed_dev, const float* frames, float* fmap,
                        float* cnet_out, int n_cnet, float* workspace, int N, int H, int W, void* stream);
/* The same with the batch in two tensors: samples 0 .. nsplit-1 from `frames`, nsplit .. N-1 from `frames_b` — the left and the
 * right frames of a pair batch where they lie, instead of the torch.cat([frame1, frame2]) copy the reference makes
 * (nndepth/encoders/basic_encoder.py:74-76).  frames_b == NULL: as nnd_encoder_forward.                               */
int nnd_encoder_forward2(const nnd_encoder_desc* desc, const float* packed_dev, const float* frames, const float* frames_b, int nsplit,
                         float* fmap, float* cnet_out, int n_cnet, float* workspace, int N, int H, int W, void* stream);
/* fp16x2 calibration of the encoder's layers after forwards with NND_FLAG_CALIBRATE (as nnd_update_block_calibration_finish) */
int nnd_encoder_calibration_finish(const nnd_encoder_desc* desc, float* packed_dev, int32_t* status_dev, void* stream);

/* ------------------------------------------------------------- pre- / post-processing on the device
 * nnd_resize_normalize : preprocess_frame  nndepth/models/raft_stereo/scripts/inference.py:55-60
 *     dst (B,C,H,W) = (bilinear_resize(src) - sub) / div, bilinear as F.interpolate(mode="bilinear") (align_corners=False);
 *     src is float (B,C,h,w), or — src_is_u8_hwc != 0 — the decoded image itself, uint8 (B,h,w,C).
 * nnd_replicate_pad    : Padder.pad / unpad  nndepth/data/dataloaders/utils.py:5-21
 *     dst (B,C,H+top+bottom,W+left+right) = F.pad(src, (left,right,top,bottom), mode="replicate"); negative values crop.
 * nnd_epe_metrics      : EvalCriterion.__call__  nndepth/models/raft_stereo/scripts/evaluate.py:48-83 (equal-size inputs)
 *     epe = sqrt(sum_c (pred-gt)^2); valid = sqrt(sum_c gt^2) < max_flow (and valid_mask != 0 if given, (B,H,W) bytes);
 *     out[0] = mean epe over valid, out[1] = number of valid pixels, out[2+k] = fraction of valid pixels with
 *     epe > thresholds[k].  `thresholds` is a HOST array (<= 4 entries); `workspace` = nnd_epe_metrics_workspace_bytes()
 *     device bytes; `out` = 2 + num_thresholds device floats.  Deterministic (two-stage double-precision reduction).   */
int nnd_resize_normalize(const void* src, int src_is_u8_hwc, float* dst, int B, int C, int h, int w, int H, int W,
                         float sub, float div, void* stream);
int nnd_replicate_pad(const float* src, float* dst, int B, int C, int H, int W, int left, int right, int top, int bottom,
                      void* stream);
int64_t nnd_epe_metrics_workspace_bytes(void);
int nnd_epe_metrics(const float* disp_gt, const float* disp_pred, const unsigned char* valid_mask, int B, int C, int H, int W,
                    float max_flow, const float* thresholds_host, int num_thresholds, void* workspace, float* out, void* stream);

/* ------------------------------------------------------------- LoFTR layer with linear attention (CREStereo)
 * Replaces LoFTREncoderLayer.forward  nndepth/blocks/transformer.py:39-66 with LinearAttention  nndepth/blocks/attn_block.py:23-58
 * (no masks).  The reference's (N, H*W, C) tokens are the (N,C,H,W) maps transposed, so every nn.Linear is a 1x1 conv of
 * the map: x, source, out are (N, d_model, H, W); out = x + norm2(mlp([x | norm1(merge(attention(x, source)))])).
 * d_model / nhead must be 32.  nnd_loftr_pack (HOST): tensors = q_proj.weight, k_proj.weight, v_proj.weight,
 * merge.weight, mlp.0.weight, mlp.2.weight, norm1.weight, norm1.bias, norm2.weight, norm2.bias (state_dict order).      */
int64_t nnd_loftr_packed_floats(int d_model, int nhead);
int64_t nnd_loftr_workspace_floats(int d_model, int nhead, int N, int H, int W);
int nnd_loftr_pack(int d_model, int nhead, const float* const* tensors_host, float* packed_host);
int nnd_loftr_layer_forward(int d_model, int nhead, const float* packed_dev, const float* x, const float* source, float* out,
                            float* workspace, int N, int H, int W, void* stream);

/* ------------------------------------------------------------- 3x3x3 Conv3d (IGEV cost-volume regulariser, row a15)
 * ConvBn3D / the conv of Upsampler3D:  nndepth/models/igev_stereo/cost_volume.py:101-130
 *   y = LeakyReLU_slope( BatchNorm3d_eval( conv3d(cat(x0, x1); w, stride, padding 1) + bias ) )        (norm / bias optional)
 * Volumes are DEPTH-MAJOR, (N, D+2, C, H, W) with one zero slice before and after the D real ones
 * (nnd_volume_to_depth_major / nnd_depth_major_to_volume convert from / to the usual (N, C, D, H, W)); a Conv3d is then one
 * launch of the 2-D MFMA convolution per sample over 3*C consecutive planes (csrc/conv3d.hip); the thin layers (8 / 16 output
 * channels) run on a direct VALU kernel (csrc/thin3d.hip; exact arithmetic, stride 2) or, with arithmetic = 2 at stride 1, on the
 * depth-marching 16-bit-MFMA kernel (csrc/slab3d.hip).  Cin1 = 0: single input.
 * nnd_conv3d_pack (HOST): w (Cout, Cin0+Cin1, 3, 3, 3), bias / bn_* may be NULL.  leaky_slope 1 = no activation.          */
typedef struct nnd_conv3d_desc {
    int32_t struct_size; /* sizeof(nnd_conv3d_desc) */
    int32_t Cout, Cin0, Cin1, stride;
    int32_t arithmetic; /* 0 = exact fp32 MFMA; 3 / 2 = 16-bit MFMA with 3 bf16 / 2 fp16 split operands
                           (see nnd_update_block_desc.arithmetic) */
    int32_t flags;      /* NND_FLAG_* */
} nnd_conv3d_desc;
int64_t nnd_conv3d_packed_floats(const nnd_conv3d_desc* desc);
int nnd_conv3d_pack(const nnd_conv3d_desc* desc, const float* w_host, const float* bias_host, const float* bn_gamma,
                    const float* bn_beta, const float* bn_mean, const float* bn_var, float bn_eps, float* packed_host);
int nnd_conv3d_forward(const nnd_conv3d_desc* desc, const float* packed_dev, const float* x0, const float* x1, float* y,
                       int N, int D, int H, int W, float leaky_slope, void* stream);
/* fp16x2 calibration of the layer after forwards with NND_FLAG_CALIBRATE (as nnd_update_block_calibration_finish) */
int nnd_conv3d_calibration_finish(const nnd_conv3d_desc* desc, float* packed_dev, int32_t* status_dev, void* stream);
int nnd_volume_to_depth_major(const float* x, float* y, int N, int C, int D, int H, int W, void* stream);
int nnd_depth_major_to_volume(const float* x, float* y, int N, int C, int D, int H, int W, void* stream);
/* the same for a volume stored (N, C, H, W, D), candidate axis contiguous — level 0 of the IGEV pyramids, rows
 * (b, g, h, w1) of w2 floats (igev_stereo/cost_volume.py:40-52): the regulariser reads / writes the pyramids in place */
int nnd_volume_rows_to_depth_major(const float* x, float* y, int N, int C, int D, int H, int W, void* stream);
int nnd_depth_major_to_volume_rows(const float* x, float* y, int N, int C, int D, int H, int W, void* stream);
/* Upsampler3D's F.interpolate(scale_factor=2, mode="trilinear", align_corners=True) (cost_volume.py:128): depth-major
 * x (N,D+2,C,H,W) -> y (N,2D+2,C,2H,2W).  FeatureGuidedBlock (cost_volume.py:133-147): vol *= sigmoid(logits (N,C,H,W)),
 * broadcast over the depth slices, in place.                                                                              */
int nnd_volume_upsample2x(const float* x, float* y, int N, int C, int D, int H, int W, void* stream);
int nnd_volume_gate(float* vol, const float* logits, int N, int C, int D, int H, int W, void* stream);

/* Fused tail of the mask head + convex upsample (the (B, 9*rate^2, H, W) mask is never written):
 *   out = convex_upsample(flow, 0.25 * conv1x1(x; W, b))      x (B,Cin,H,W), flow (B,1,H,W), out (B,1,rate*H,rate*W)
 * Replaces update_block.py:97-101,111 (mask.2, x0.25) + raft_stereo/model.py:93-105.  `packed_dev` is the blob of
 * nnd_conv2d_pack for the (9*rate^2, Cin, 1, 1) weight.  Built for rate 4 / 8 and Cin 128 / 256.            */
int nnd_mask_upsample_forward(const float* packed_dev, const float* x, const float* flow, float* out,
                              int B, int Cin, int H, int W, int rate, void* stream);

/* ------------------------------------------------------------------ fused refinement loop
 * Replaces the `for _ in range(self.iters)` loop of RAFTStereo.forward
 *   nndepth/models/raft_stereo/model.py:126-137 (+ initialize_coords :87-91)
 * net/inp are the tanh/relu halves of cnet (model.py:119-122); `pyramid` comes from
 * nnd_corr1d_build.  coords start at arange(W) (+ disp_init if non-NULL).  Every iteration:
 * lookup -> update block -> coords += delta -> convex upsample of (coords - arange).
 * up_out: iteration i writes (B,1,rate*H,rate*W) at up_out + i*up_iter_stride floats
 *         (stride 0 keeps only the last); low_out (optional) receives the final 1/rate-res
 *         disparity (B,1,H,W); net_out (optional) the final hidden state.
 * Only flow_channels == 1 (RAFT-Stereo / IGEV-style 1-D disparity).                       */
int nnd_raft_stereo_refine(const nnd_update_block_desc* desc, const float* packed_dev,
                           const float* pyramid, int num_levels, int radius,
                           const float* net, const float* inp, const float* disp_init,
                           float* up_out, int64_t up_iter_stride, float* low_out, float* net_out,
                           float* workspace, int B, int H, int W, int rate, int iters, void* stream);
/* One cascade stage of Coarse2FineGroupRepViTRAFTStereo.forward (nndepth/models/raft_stereo/model.py:297-311): the same loop with
 * nnd_group_corr1d_lookup over `group_pyramid` (nnd_group_corr_build_scaled); disp_init = the previous stage's up_disp (its
 * resolution is this stage's: init_coords = org_coords + up_disp, :316).  The lookup runs as its own kernel (it gathers other
 * pixels' rows, Q6), convc1 behind it; everything else as in nnd_raft_stereo_refine.                                            */
int nnd_raft_stereo_group_refine(const nnd_update_block_desc* desc, const float* packed_dev, const float* group_pyramid, int num_groups,
                                 int num_levels, int radius, const float* net, const float* inp, const float* disp_init, float* up_out,
                                 int64_t up_iter_stride, float* low_out, float* net_out, float* workspace, int B, int H, int W, int rate,
                                 int iters, void* stream);

/* IGEV variant of the loop (nndepth/models/igev_stereo/model.py:148-158): lookup = nnd_igev_lookup over both
 * pyramids, and — reference quirk Q5 — the update block and the convex upsample receive the ABSOLUTE coordinate
 * coords1 = arange(W) + disp_init + sum(delta), not the disparity.  disp_init = the soft-argmin initial disparity
 * (computed by the caller: Conv3d squeezer + softmax, PyTorch).  up_out / low_out hold coordinates accordingly.
 * interleaved: optional (may be NULL) output of nnd_igev_interleave_pyramids for the same pyramids; when given, the
 * per-iteration lookup gathers from it (same values, ~4x fewer HBM lines).  nnd_igev_refine_reads_interleaved = 1 when,
 * given that copy, the loop reads nothing else of the two pyramids (groups / radius the fused lookup is built for and the
 * fused lookup not switched off): their pooled levels may then be left unwritten.                                 */
int nnd_igev_refine_reads_interleaved(int num_groups, int num_levels, int radius);
int nnd_igev_stereo_refine(const nnd_update_block_desc* desc, const float* packed_dev,
                           const float* feat_pyramid, const float* geo_pyramid, const float* interleaved,
                           int num_groups, int num_levels, int radius,
                           const float* net, const float* inp, const float* disp_init,
                           float* up_out, int64_t up_iter_stride, float* low_out, float* net_out,
                           float* workspace, int B, int H, int W, int rate, int iters, void* stream);

/* CREStereo variant of the loop — one stage of the cascade (nndepth/models/cre_stereo/model.py:221-236, 246-259, 270-284):
 * every iteration: AGCL correlation of (fmap1, fmap2) at the current flow -> update block (flow_channels = 2) ->
 * flow += delta -> 2-channel convex upsample.  Iteration i searches a 1x9 window when i is even, 3x3 when odd.
 * `scratch` = caller-owned device buffer of `scratch_floats` floats (the size is part of the contract: the library checks it
 * and never writes past it).
 * extra_offset == NULL: iter mode (nnd_agcl_corr_iter); the scratch receives the warped right map and must hold
 * B*C*H*W floats (else NND_ERR_INVALID);
 * extra_offset (B,18,H,W): offset mode (nnd_agcl_corr_offset; fmap1/fmap2 already attended by the caller); with
 * scratch_floats >= 2*B*C*H*W and C == 256 the two maps are copied channels-last into the scratch once per call and
 * sampled line by line (nnd_agcl_corr_offset_nhwc); with a smaller / NULL scratch or C != 256: the planar kernel, same
 * results to 1e-7, slower.  flow_init (B,2,H,W) or NULL for zero.
 * up_out: iteration i writes (B,2,rate*H,rate*W) at up_out + i*up_iter_stride; low_out (optional) the final flow (B,2,H,W); net_out (optional) the hidden state.   */
int nnd_cre_stereo_refine(const nnd_update_block_desc* desc, const float* packed_dev,
                          const float* fmap1, const float* fmap2, int C, const float* extra_offset,
                          float* scratch, int64_t scratch_floats, const float* net, const float* inp, const float* flow_init,
                          float* up_out, int64_t up_iter_stride, float* low_out, float* net_out,
                          float* workspace, int B, int H, int W, int rate, int iters, void* stream);

/* ------------------------------------------------------------------------------ profiling
 * Times `reps` back-to-back launches of ONE hot-path conv (selected by `which`, see
 * nnd_conv_name) on `stream` with hipEvents recorded on that same stream and returns the
 * average milliseconds per launch in *ms_out and its algorithmic FLOPs in *flops_out.
 * Used by bench.py for the roofline object; inputs are whatever the workspace holds.      */
int nnd_profile_conv(const nnd_update_block_desc* desc, const float* packed_dev, float* workspace,
                     int B, int H, int W, int which, int reps, void* stream,
                     float* ms_out, double* flops_out);
/* The same conv timed where it runs in production: inside the fused RAFT-Stereo loop (arguments as
 * nnd_raft_stereo_refine, only the last up_disp is kept), bracketed by hipEvents on `stream` in every iteration.  *ms_out = average over iterations 2..iters (event-to-event, so it
 * includes the launch gap in front of the kernel).  Synchronises `stream`.  `which` must be one of the stand-alone
 * launches of the loop (not convc1 / mask.2, which are fused into their neighbours).                    */
int nnd_profile_loop_conv(const nnd_update_block_desc* desc, const float* packed_dev, const float* pyramid,
                          int num_levels, int radius, const float* net, const float* inp, float* up_out,
                          float* workspace, int B, int H, int W, int rate, int iters, int which, void* stream,
                          float* ms_out);
/* Calibration of the above: the same loop with BOTH events of every iteration recorded in front of conv `which`, nothing
 * between them.  *ms_out = what an event pair alone measures at that place of the stream (the part of nnd_profile_loop_conv's
 * figure that is not the launch): bench.py reports it beside the raw figure.                                         */
int nnd_profile_loop_event_pair(const nnd_update_block_desc* desc, const float* packed_dev, const float* pyramid,
                                int num_levels, int radius, const float* net, const float* inp, float* up_out,
                                float* workspace, int B, int H, int W, int rate, int iters, int which, void* stream,
                                float* ms_out);
/* Diagnostic: sustained fp32-MFMA rate of this device (dependent v_mfma_f32_32x32x2 chains, no memory
 * traffic) at `waves_per_simd` resident waves; `scratch_dev` is any device buffer of >= 1 float.        */
int nnd_profile_mfma_peak(int waves_per_simd, int iters, void* stream, float* scratch_dev, float* tflops_out);
/* Diagnostic: the same for the 16-bit MFMA of the split arithmetics (v_mfma_f32_32x32x16_f16, or _bf16 with bf16 != 0): 4 independent
 * chains per wave on pseudo-random operands, about `target_ms` of nothing else — the rate the chip SUSTAINS under its power limit on
 * this box and the shader clock it settles at (nominal: 2500 TFLOP/s at 2.4 GHz).  clk_dev: device buffer of >= 512 64-bit words. */
int nnd_profile_mfma16_peak(int bf16, int waves_per_simd, float target_ms, void* stream, float* scratch_dev,
                            unsigned long long* clk_dev, float* tflops_out, float* ghz_out);
int nnd_num_convs(const nnd_update_block_desc* desc);
const char* nnd_conv_name(const nnd_update_block_desc* desc, int which);

#ifdef __cplusplus
}
#endif
#endif /* NNDEPTH_AMD_H */
