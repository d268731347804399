// Implicit-GEMM 2-D convolution (stride 1, zero "same" padding) on the gfx950 16-bit MFMA with fp32 operands carried
// as SPLIT bf16 pieces — the route past the 157 TFLOP/s fp32-MFMA ceiling:
//
//   x = x0 + x1 + x2,  w = w0 + w1 + w2     (bf16 pieces by successive round-to-nearest of the residual: 3 x 8 = 24 bits)
//   x*w ~= sum_{i+j<=2} x_i*w_j             (6 products; the dropped ones are <= 2^-24 |x||w|, an fp32 rounding)
//   every product is an exact fp32 value (8 x 8 significand bits) and v_mfma_f32_32x32x16_bf16 accumulates in fp32.
//
// 6 MFMAs at 16x the fp32-MFMA rate = 0.375x the matrix time of conv_mfma.hip for the same result to fp32 rounding level
// (measured error per op and drift over the 32-iteration recurrence: DESIGN.md §4, tests/test_gpu_split.py).  The exact
// fp32 kernel stays the default; this one is selected per layer by ConvLayer::arith (nnd_update_block_desc.arithmetic).
//
// Mapping (D = A*B, same accumulator layout as the fp32 kernel, so conv_epilogue.h is shared):
//   A = weights     pre-split on the host, packed in fragment order [cb][16-channel chunk][tap][piece][lane][8 x bf16]:
//                   one coalesced 1 KiB dwordx4 load per wave per (tap, piece); never touches LDS
//   B = activations fp32 in HBM (tile-major or NCHW); the staging threads split them while they build the halo patch in
//                   LDS as [K-slice][sub-tile][row][col][piece][16 channels] bf16: a lane's B fragment (8 consecutive
//                   channels of one pixel and piece) is one ds_read_b128 at lane_base + immediate
// Workgroup = wco x ks waves on P = 2 independent 4x8-pixel sub-tiles (linear sub-tile index 2*blockIdx.x + pp, so 255
// sub-tiles pair into 128 workgroups without a ragged rectangle): wave (cbi, kj) owns output-channel block
// blockIdx.y*wco + cbi, both sub-tiles (each A fragment feeds two MFMAs — at 16x the MFMA rate the weight stream, not the
// matrix pipe, is what limits a 32-pixel tile) and K-slice kj: of every super-chunk of ks*16 input channels staged in LDS
// it multiplies channels [kj*16, kj*16+16); the ks partial tiles are summed through LDS at the end.
// LDS image: position stride PS = pieces*32 + 16 bytes (an odd number of 16-B units) and row stride ROWB with
// ROWB/16 = 8 (mod 16); with the lane -> pixel permutation of lane_pixel() the 16 lanes that one ds_read_b128 cycle
// serves (lanes {0-3,12-15,20-27}, {4-11,16-19,28-31} of each half-wave) hit 16 different 16-B bank groups for every tap.
//
// Replaces (when selected) the same nn.Conv2d calls as conv_mfma.hip: nndepth/blocks/update_block.py:57-65,26-36,97-112,
// nndepth/blocks/gru.py:22-37,53-61.
#include "conv_split_kernel.h"
#include "flow_branch.h"

#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace nnd {

// --------------------------------------------------------------------------- host side
namespace {
// fp16x2 pieces on the host: clang's _Float16 conversion is IEEE round-to-nearest-even including subnormal results
uint16_t f16_rn(float x) {
    const _Float16 h = (_Float16)x;
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}
float f16_to_f(uint16_t u) {
    _Float16 h;
    memcpy(&h, &u, 2);
    return (float)h;
}
uint16_t bf16_rn(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
float bf16_to_f(uint16_t h) {
    const uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// fast = the kernel's FAST regime is possible for this launch (c4 sources; checked per candidate: full super-chunks inside one
// source).  P (sub-tiles per wave) is a search dimension there: each weight fragment feeds P MFMA groups, so a workgroup of
// a output-channel blocks x P sub-tiles moves a x (weights of a block) + P x (patch of a sub-tile) through its CU's L2 path —
// at 68x120 the 12-wave shapes of round 2 (3 x 2, 4 x 2, 6 x 2 blocks x sub-tiles) become 2 x 3, 2 x 4, 4 x 3.
bool pick_split(const ConvLayer& L, int c0, int c1, int B, int H, int W, bool fast_ok, SplitCfg* out) {
    const int NS = L.arith, STR = L.stride;
    // patch geometry of one sub-tile (SplitGeom in conv_split_kernel.h): positions staged and bytes in LDS
    const int PRI = 3 * STR + L.KH, PCI = 7 * STR + L.KW, k11 = L.KH * L.KW == 1;
    const int nph = STR == 1 ? 1 : (k11 ? 1 : 4);
    const int prp = STR == 1 ? PRI : (L.KH == 1 ? 4 : (PRI + 1) / 2), pcp = STR == 1 ? PCI : (L.KW == 1 ? 8 : (PCI + 1) / 2);
    const int npos = (STR == 2 && k11) ? 32 : PRI * PCI;
    const size_t subb = (size_t)nph * prp * split_row_bytes(pcp, NS);
    const int tiles_x = cdiv(W, 8), ntiles = tiles_x * cdiv(H, 4);
    int force_ny = switches().split_ny, force_ks = switches().split_ks, force_p = switches().split_p;
    const bool forced = force_ny > 0 || force_ks > 0 || force_p > 0;
    const long px_wgs2 = (long)cdiv(ntiles, 2) * B;
    // Short K (Cin <= 64: at most 4 chunks, the encoder's first residual stage): a workgroup is mostly prologue, exchange and
    // epilogue, so the best shape is the smallest one — no split K, all output-channel blocks in one workgroup (the patch is
    // staged once), 6 two-wave workgroups per CU covering each other's fixed phases.  Measured, 64 -> 64 3x3 at 272x480x2
    // (scripts/sweep_split_encoder.py): (ny,ks) = (1,1) 158 us | (1,2) 226 | (1,4) 326 | (2,1) 333; exact fp32 kernel 254.
    bool rule = false;
    if (L.nchunks <= 4 && !forced && L.ncb <= SPLIT_MAX_WAVES) {
        // (at most 4 blocks per workgroup: IGEV's 64 -> 192 head conv as two groups of 3 — 136x240 41.6 -> 33.5 us, batch 8 304 -> 206)
        force_ny = 1;
        while (force_ny < L.ncb && (L.ncb % force_ny != 0 || L.ncb / force_ny > 4)) ++force_ny;
        force_ks = 1;
        force_p = 2;
        rule = true;
    }
    // Many workgroup columns (>= 448: 136x240 maps, batch 8 at KITTI size, the encoder's first stages): the chip is filled
    // several times over, so no split K (no exchange, small LDS, several workgroups per CU covering each other's prologue and
    // epilogue); with a long K (>= 16 chunks) at most 4 output-channel blocks per workgroup, with a short K all of them (the
    // patch staging then dominates and is done once).  Measured with scripts/sweep_split.py at 48x156 batch 8 against the
    // cost model below: zr 312 -> 280 us, q 202 -> 165, convc2 412 -> 379, conv 246 -> 228 (profiles/r02_split_wg_shape_sweep.txt).
    bool try_ks2 = false;
    if (!rule && px_wgs2 >= 448 && !forced) {
        rule = true;
        force_ks = 1;
        force_ny = 1;
        force_p = 2;
        const int max_wco = L.nchunks >= 16 ? 4 : SPLIT_MAX_WAVES;
        while (force_ny < L.ncb && (L.ncb % force_ny != 0 || L.ncb / force_ny > max_wco)) ++force_ny;
        // ... unless such workgroups leave most of the chip's wave slots empty (IGEV at 136x240, batch 1: 510 columns of 2-wave
        // workgroups = 1020 waves for 3072 slots): then 2-way split K (4-wave workgroups).  Measured (scripts/sweep_split.py
        // 136 240 1 fp16x2 igev, profiles/r03_split_shape_sweep_igev_136x240.txt): conv 256 -> 64 57.0 -> 42.7 us, q 28.4 -> 22.8;
        // the same at batch 8 (4080 columns) and the 4-block zr conv (2040 waves) are not faster with it.
        try_ks2 = px_wgs2 * force_ny * (L.ncb / force_ny) < 256 * SPLIT_MAX_WAVES / 2;
    }
    auto search = [&](int f_ny, int f_ks, int f_p) {
        double best = 1e30;
        bool found = false;
        for (int P : {2, 3, 4}) {
            // P = 3 / 4 only when forced (NND_SPLIT_CFG): measured at 68x120 with fp16x2 (profiles/r03_split_shape_sweep_68x120.txt)
            // the 8-wave P = 3 / 4 shapes lose to the 12-wave P = 2 ones on every layer (convc2 29.4 vs 26.2 us, zr 26.3 vs 26.1,
            // fhm 29.4 vs 27.0): with the weight ring pinned, the L2 path is no longer what the K loop waits for
            if ((f_p > 0 && P != f_p) || (f_p <= 0 && P != 2)) continue;
            const long px_wgs = (long)cdiv(ntiles, P) * B;
            for (int ny = 1; ny <= L.ncb; ++ny) {
                if (L.ncb % ny != 0 || (f_ny > 0 && ny != f_ny)) continue;
                const int wco = L.ncb / ny;
                for (int ks : {1, 2, 4}) {
                    if (f_ks > 0 && ks != f_ks) continue;
                    const int waves = wco * ks;
                    if (waves > SPLIT_MAX_WAVES || ks > L.nchunks) continue;
                    if (c1 > 0 && c0 % (ks * 16) != 0) continue;
                    const bool fast = fast_ok && L.nchunks % ks == 0;
                    if (P > 2) continue;  // P = 3 / 4 are not instantiated (see launch_split_shape; also measured for the encoder's
                                          // 2-wave workgroups: 64 -> 64 at 272x480x2 takes 178 us with P = 4 against 111 with P = 2,
                                          // profiles/r03_split_encoder_shapes.txt)
                    if (STR == 2 && !fast) continue;  // the stride-2 kernels exist in the FAST regime only
                    const int nu = cdiv(P * npos * 2 * ks, 64 * waves);  // staging units per thread
                    if (nu > 4 || (P == 3 && nu > 3)) continue;
                    size_t lds = (size_t)2 * ks * P * subb;
                    const size_t red = ks > 1 ? (size_t)waves * P * 4096 : 0;
                    if (red > lds) lds = red;
                    if (lds > 160 * 1024) continue;
                    const int max_waves = P > 2 ? 8 : SPLIT_MAX_WAVES;
                    int wg_per_cu = (int)((160 * 1024) / lds);
                    if (wg_per_cu > max_waves / waves) wg_per_cu = max_waves / waves;
                    if (wg_per_cu < 1) wg_per_cu = 1;
                    const double rounds = std::ceil((double)px_wgs * ny / (256.0 * wg_per_cu));
                    const double simd_waves = std::ceil(waves * wg_per_cu / 4.0);
                    // matrix time of the busiest SIMD, in units of one sub-tile x one 16-channel chunk
                    double t = rounds * simd_waves * cdiv(L.nchunks, ks) * P / 2.0;
                    t *= 1.0 + 0.03 * (ks - 1);       // split-K exchange
                    t *= 1.0 + 0.02 * (4 - (wco < 4 ? wco : 4));  // fewer waves share one staged patch
                    if (nu > 2 && P == 2) t *= 1.2;                // register-heavy staging variant
                    if (t < best) {
                        best = t;
                        *out = {ny, wco, ks, P, ntiles, tiles_x, P == 2 ? (nu <= 2 ? 2 : 4) : (nu <= 3 ? 3 : 4), fast, lds, STR};
                        found = true;
                    }
                }
            }
        }
        return found;
    };
    if (try_ks2 && search(force_ny, 2, force_p)) return true;
    if (search(force_ny, force_ks, force_p)) return true;
    return rule && search(-1, -1, -1);  // the regime rule's shape does not exist for this layer: the cost model decides
}
}  // namespace

// the kernel instantiations live in conv_split_ns2.hip / conv_split_ns3.hip
template <>
int launch_split_ns<2>(const ConvArgs& a, const SplitCfg& cfg, int KH, int KW, dim3 grid, dim3 block, hipStream_t stream);
template <>
int launch_split_ns<3>(const ConvArgs& a, const SplitCfg& cfg, int KH, int KW, dim3 grid, dim3 block, hipStream_t stream);

bool conv_split_supported(int KH, int KW, int Cin, int stride, int arith, int Cout) {
    if ((arith != 3 && arith != 2) || (stride != 1 && stride != 2) || Cin % 16 != 0) return false;
    // stride 2: planar sources only (checked at launch), and at least 3 output-channel blocks: the 9 x 17-position patch of a
    // sub-tile is staged by the workgroup's output-channel waves, fewer of them would need more than 4 staging units per thread
    if (stride == 2) return ((KH == 3 && KW == 3) || (KH == 1 && KW == 1)) && Cout >= 96;
    return (KH == 3 && KW == 3) || (KH == 1 && KW == 5) || (KH == 5 && KW == 1) || (KH == 1 && KW == 1);
}

int launch_conv_split(const ConvLayer& L, const float* blob, const ConvIO& io, int epi, int B, int H, int W, hipStream_t stream) {
    NND_REQUIRE(conv_split_supported(L.KH, L.KW, L.Cin, L.stride, L.arith, L.Cout), "conv_split: %dx%d Cin=%d Cout=%d stride %d arith %d not built",
                L.KH, L.KW, L.Cin, L.Cout, L.stride, L.arith);
    NND_REQUIRE(io.src0.C + io.src1.C == L.Cin, "conv_split: source channels %d+%d != Cin %d", io.src0.C, io.src1.C, L.Cin);
    NND_REQUIRE(L.CI_T == 16 && L.nchunks * 16 == L.Cin, "conv_split: layer was not planned for 16-channel chunks");
    const int Hin = io.Hin > 0 ? io.Hin : H, Win = io.Win > 0 ? io.Win : W;  // input size (stride 2: the caller passes it)
    NND_REQUIRE(L.stride == 1 ? (Hin == H && Win == W) : (H == (Hin + 1) / 2 && W == (Win + 1) / 2),
                "conv_split: output %dx%d does not match input %dx%d at stride %d", H, W, Hin, Win, L.stride);
    NND_REQUIRE(L.stride == 1 || io.src1.C == 0, "conv_split: stride 2 is built for one source");
    NND_REQUIRE((long)(L.Cin + 64) * tiled_plane(Hin, Win) < (1L << 31), "conv_split: plane offsets exceed 32 bits");
    SplitCfg cfg;
    // FAST regime: any source layout the kernel addresses as plane + pixel offset — c4 tile-major (the refinement loops' own
    // tensors), planar tile-major (the encoder's) or NCHW (C-ABI tensors: cnet_proj reads the feature map, the Conv3d layers
    // their depth-major volumes); per candidate shape the picker also requires full super-chunks that never straddle the two
    // sources.  NND_SPLIT_NO_FAST (diagnostic) keeps the generic kernel.
    // (the stride-2 kernels exist in the FAST regime only: the diagnostic switch does not apply to them)
    const bool fast_ok = !switches().split_no_fast || L.stride == 2;
    NND_REQUIRE(pick_split(L, io.src0.C, io.src1.C, B, H, W, fast_ok, &cfg), "conv_split: no configuration for %dx%d Cin=%d (%d+%d)", L.KH,
                L.KW, L.Cin, io.src0.C, io.src1.C);
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.src0 = io.src0.ptr; a.bs0 = io.src0.bstride; a.c0 = io.src0.C;
    a.src1 = io.src1.ptr; a.bs1 = io.src1.bstride; a.c1 = io.src1.C;
    a.wpk = blob + L.w_off;
    a.bias = blob + L.b_off;
    a.out0 = io.out0.ptr; a.obs0 = io.out0.bstride;
    a.out1 = io.out1.ptr; a.obs1 = io.out1.bstride;
    a.aux0 = io.aux0.ptr; a.abs0 = io.aux0.bstride;
    a.aux1 = io.aux1.ptr; a.abs1 = io.aux1.bstride;
    a.bmap = io.bmap.ptr; a.bmbs = io.bmap.bstride;
    a.ls = make_lay(Hin, Win, io.src_tiled, io.src_c4);
    a.ld = make_lay(H, W, io.dst_tiled, io.dst_c4);
    NND_REQUIRE(!io.src_c4 || (io.src_tiled && io.src0.C % 4 == 0 && io.src1.C % 4 == 0), "conv_split: c4 sources need channel counts %% 4 == 0");
    NND_REQUIRE(!io.dst_c4 || (io.dst_tiled && (L.Cout % 4 == 0 || (!io.bmap.ptr && !io.aux0.ptr && !io.aux1.ptr && !io.out1.ptr))),
                "conv_split: c4 destination with per-pixel operands needs Cout %% 4 == 0");
    a.H = H; a.W = W; a.Cout = L.Cout; a.nchunks = L.nchunks; a.epi = epi; a.hidden = io.hidden;
    a.Hin = Hin; a.Win = Win; a.flags = io.flags;
    a.cscale = L.s_off >= 0 ? blob + L.s_off : nullptr;
    NND_REQUIRE(epi != EPI_AFFINE || a.cscale, "conv_split: EPI_AFFINE needs a packed scale vector");
    a.tiles_x = cfg.tiles_x; a.wco = cfg.wco; a.ks = cfg.ks; a.npos = cfg.ntiles;
    a.scale = io.scale;
#ifdef NND_DBG_STAMPS  // NND_DBG_STAMP_LAUNCH=n: only the n-th conv_split launch of the process records its stamps (default: every launch)
    {
        static std::atomic<int> launches{0};
        static const int only = getenv("NND_DBG_STAMP_LAUNCH") ? atoi(getenv("NND_DBG_STAMP_LAUNCH")) : -1;
        const int idx = launches.fetch_add(1);
        a.dbg_stamp = only < 0 || idx == only;
    }
#endif
    if (calibrating() && L.arith == 2) {  // record the largest |activation| this launch stages (calib.hip)
        const float* tail = blob + L.tail_off();
        if (int rc = calib_amax_act(io.src0, a.ls, B, Hin, Win, tail, stream)) return rc;
        if (int rc = calib_amax_act(io.src1, a.ls, B, Hin, Win, tail, stream)) return rc;
    }
    dim3 grid(cdiv(cfg.ntiles, cfg.P), cfg.ny, B), block(64 * cfg.wco * cfg.ks);
    const bool verbose = switches().conv_verbose;
    if (verbose)
        fprintf(stderr, "[nnd] conv_split %dx%d Cin=%d Cout=%d pieces=%d: ny=%d, wco=%d, ks=%d, P=%d, nu=%d%s, grid %ux%ux%u, lds %zu B\n", L.KH,
                L.KW, L.Cin, L.Cout, L.arith, cfg.ny, cfg.wco, cfg.ks, cfg.P, cfg.nu, cfg.fast ? ", fast" : "", grid.x, grid.y, grid.z, cfg.lds);
    int rc = L.arith == 3 ? launch_split_ns<3>(a, cfg, L.KH, L.KW, grid, block, stream)
                          : launch_split_ns<2>(a, cfg, L.KH, L.KW, grid, block, stream);
    NND_REQUIRE(rc != NND_ERR_UNSUPPORTED, "conv_split: shape %dx%d P=%d nu=%d %s is not instantiated", L.KH, L.KW, cfg.P, cfg.nu,
                cfg.fast ? "fast" : "generic");
    if (rc != NND_OK) return rc;
    NND_LAUNCH_CHECK();
    return NND_OK;
}

// Host packer: same (cout, cin_src, KH, KW) inputs as pack_conv; blob order [cb][chunk][tap][piece][lane][8] 16-bit values with
// lane = h*32 + (co % 32) holding channels chunk*16 + 8h + 0..7 (the A operand of v_mfma_f32_32x32x16_{bf16,f16}).
// fp16x2 (arith 2): the pieces are those of w * 2^s, s per layer such that max|w| * 2^s lies in [2^13, 2^14), and the 4 floats
// behind the bias vector (bias[ncb*32 ...], split_arith.h: SPLIT_TAIL_*) are oscale = 2^-(s + xs), which the kernels multiply their
// accumulators by, the activation scale 2^xs (xs = SPLIT_F16_XSHIFT until the layer is calibrated), the calibration accumulator
// and 2^-s.
void pack_conv_split(const ConvLayer& L, int nparts, const float* const* w, const float* const* bvec, const int* cout, float* blob,
                     const int* ci_map, int cin_src) {
    if (!ci_map) cin_src = L.Cin;
    const int NT = L.KH * L.KW, NS = L.arith;
    uint16_t* wp = reinterpret_cast<uint16_t*>(blob + L.w_off);
    float* bp = blob + L.b_off;
    memset(wp, 0, sizeof(float) * L.w_floats());
    memset(bp, 0, sizeof(float) * L.b_floats());
    float wscale = 1.f;
    if (NS == 2) {
        float wmax = 0.f;
        for (int part = 0; part < nparts; ++part)
            for (int col = 0; col < cout[part]; ++col)
                for (int ci = 0; ci < L.Cin; ++ci)
                    for (int t = 0; t < NT; ++t) {
                        const float v = std::fabs(w[part][((size_t)col * cin_src + (ci_map ? ci_map[ci] : ci)) * NT + t]);
                        if (std::isfinite(v) && v > wmax) wmax = v;
                    }
        int e = 0;
        if (wmax > 0.f) std::frexp(wmax, &e);  // wmax = m * 2^e, m in [0.5, 1)  ->  wmax * 2^(14 - e) in [2^13, 2^14)
        const int s = wmax > 0.f ? 14 - e : 0;
        wscale = std::ldexp(1.f, s);
        float* tail = bp + L.ncb * 32;  // split_arith.h: SPLIT_TAIL_*
        tail[SPLIT_TAIL_OSCALE] = std::ldexp(1.f, -(s + SPLIT_F16_XSHIFT));
        tail[SPLIT_TAIL_XSCALE] = std::ldexp(1.f, SPLIT_F16_XSHIFT);  // until the layer is calibrated (calib.hip)
        tail[SPLIT_TAIL_AMAX] = 0.f;
        tail[SPLIT_TAIL_WSINV] = std::ldexp(1.f, -s);
    }
    int co0 = 0;
    for (int part = 0; part < nparts; ++part) {
        for (int col = 0; col < cout[part]; ++col) {
            const int co = co0 + col, cb = co / 32, i = co % 32;
            bp[co] = bvec[part] ? bvec[part][col] : 0.f;
            for (int ci = 0; ci < L.Cin; ++ci) {
                const int chunk = ci / 16, cl = ci % 16, h = cl / 8, j = cl % 8, lane = h * 32 + i;
                for (int t = 0; t < NT; ++t) {
                    float res = w[part][((size_t)col * cin_src + (ci_map ? ci_map[ci] : ci)) * NT + t] * wscale;
                    for (int s = 0; s < NS; ++s) {
                        const uint16_t piece = NS == 2 ? f16_rn(res) : bf16_rn(res);
                        res -= NS == 2 ? f16_to_f(piece) : bf16_to_f(piece);
                        wp[((((((size_t)cb * L.nchunks + chunk) * NT + t) * NS + s) * 64 + lane) * 8) + j] = piece;
                    }
                }
            }
        }
        co0 += cout[part];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Flow branch of the motion encoder as ONE launch (arithmetic 3):  out = relu(conv3x3_{128->64}(relu(conv7x7_{FC->128}(flow))))
//   nndepth/blocks/update_block.py:26-36, 57-65 (BasicMotionEncoder.convf1 / convf2)
// The two launches it replaces cost 9.8 + 21.4 us of a 322-us iteration for 1.3 GFLOP.  One workgroup = one 4x8 sub-tile, 8 waves:
//   phase 0  the 12x16 flow window of the sub-tile, the 7x7 weights (tap-major copy made at pack time: 4 consecutive channels
//            = one 16-B read) and their bias into LDS;
//   phase 1  f1 on the 6x10 halo patch: a thread = 2 horizontally adjacent positions x 8 channels (one 16-B slot of the patch) —
//            a window row in registers, each 16-B weight read feeds 8 FMAs (the first attempt at this fusion, inside the generic
//            kernel's staging units, did one LDS read per FMA and took 51 us) — per channel in convf1_kernel's tap order;
//            bias, ReLU, zero outside the image (convf2's padding), split into 3 bf16 pieces, written in conv_split's patch
//            layout (so the 128-channel map never exists in HBM);
//   phase 2  conv_split's MFMA walk: wave = (output block, K slice of 4), slice kj takes chunks kj and kj + 4; 6 products per
//            step, small ones first; with 2 waves per SIMD nothing else hides the L2 latency of the weight fragments, so they run
//            FB_AD steps ahead through a register ring (one step ahead: 35.8 us for the launch);
//   phase 3  the four K slices meet in LDS (slice 0 + 1 + 2 + 3), shared epilogue (bias, ReLU, c4 or planar store).
// Arithmetic and summation order are those of convf1_kernel followed by conv_split with ks = 4: bit-identical to that pair
// (tests/test_gpu_split.py).
template <int FC, int NS>
static int launch_fb(const FlowBranchArgs& a, dim3 grid, dim3 block, hipStream_t stream) {
    auto kern = flow_branch_kernel<FC, NS>;
    static std::atomic<unsigned> raised{0};
    if (int rc = raise_lds_limit(reinterpret_cast<const void*>(kern), raised)) return rc;
    const size_t lds = (size_t)(fb_lds_bytes<FC, NS>()) + switches().lds_slack;
    hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
    return NND_OK;
}

bool flow_branch_supported(const ConvLayer& f2, int fc) {
    return (f2.arith == 3 || f2.arith == 2) && f2.KH == 3 && f2.KW == 3 && f2.Cin == FB_C1 && f2.Cout == 64 && f2.stride == 1 && (fc == 1 || fc == 2);
}

int make_flow_branch_args(const ConvLayer& f2, const float* blob, const float* w7t, const float* b7, const float* flow, int64_t fbs,
                          int fc, const ConvIO& io, int B, int H, int W, FlowBranchArgs* out) {
    NND_REQUIRE(flow_branch_supported(f2, fc), "flow_branch: built for convf1 7x7 (1 or 2 -> 128) + convf2 3x3 (128 -> 64) in split arithmetic");
    NND_REQUIRE(io.dst_tiled && f2.CI_T == 16 && f2.nchunks == FB_NCH, "flow_branch: tile-major destination, 16-channel chunks");
    NND_REQUIRE((long)(fc + 64) * tiled_plane(H, W) < (1L << 31), "flow_branch: plane offsets exceed 32 bits");
    FlowBranchArgs& a = *out;
    memset(&a, 0, sizeof(a));
    a.flow = flow; a.fbs = (long)fbs; a.w7t = w7t; a.b7 = b7;
    a.lf = make_lay(H, W, true);
    a.c.wpk = blob + f2.w_off;
    a.c.bias = blob + f2.b_off;
    a.c.out0 = io.out0.ptr; a.c.obs0 = io.out0.bstride;
    a.c.ld = make_lay(H, W, true, io.dst_c4);
    a.c.H = H; a.c.W = W; a.c.Hin = H; a.c.Win = W; a.c.Cout = f2.Cout; a.c.epi = EPI_RELU;
    a.c.tiles_x = cdiv(W, 8);
    a.c.npos = a.c.tiles_x * cdiv(H, 4);
    a.c.scale = 1.f;
    return NND_OK;
}

int launch_flow_branch(const ConvLayer& f2, const float* blob, const float* w7t, const float* b7, const float* flow, int64_t fbs,
                       int fc, const ConvIO& io, int B, int H, int W, hipStream_t stream) {
    FlowBranchArgs a;
    int rc0 = make_flow_branch_args(f2, blob, w7t, b7, flow, fbs, fc, io, B, H, W, &a);
    if (rc0 != NND_OK) return rc0;
    dim3 grid(a.c.npos, 1, B), block(512);
    int rc;
    if (fc == 1) rc = f2.arith == 3 ? launch_fb<1, 3>(a, grid, block, stream) : launch_fb<1, 2>(a, grid, block, stream);
    else rc = f2.arith == 3 ? launch_fb<2, 3>(a, grid, block, stream) : launch_fb<2, 2>(a, grid, block, stream);
    if (rc != NND_OK) return rc;
    NND_LAUNCH_CHECK();
    return NND_OK;
}

#ifdef NND_DBG_STAMPS
extern "C" int nnd_debug_read_fb_stamps(unsigned long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_fb_stamps), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -1;
}
#endif
}  // namespace nnd
