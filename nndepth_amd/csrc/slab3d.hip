// Thin 3x3x3 Conv3d layers of IGEV's cost-volume regulariser (ConvBn3D / Upsampler3D with 8 or 16 output channels at stride 1,
// the stride-2 layers conv1.0 8->16 and conv2.0 16->32; nndepth/models/igev_stereo/cost_volume.py:101-130, 176-190) on the
// 16-bit MFMA in the fp16x2 split arithmetic (split_arith.h), as a DEPTH-MARCHING kernel on depth-major volumes (conv3d.hip).
//
// Why a third formulation.  On the 2-D split kernel (conv3d.hip: J output slices folded into 32 output channels) these layers
// are a K of (J+2)*Cin planes with ONE output-channel block: a workgroup is a single wave that stages (fp32 load, split into
// two fp16 pieces, LDS write) every input element (J+2)/J x 1.9 (halo of a 4x8 sub-tile) times for 32 MFMA rows of use, and
// pulls the whole weight set from L2 per 64 pixels.  Measured at 544x960 (profiles/r03_igev_regulariser_ablation.txt): conv1_up
// 16->8 takes 860 us, 470 without the staging, 625 without the weight loads — and the same 860 without its MFMAs.  Here:
//   workgroup = a TY x 32 pixel column of the volume, marching along the depth axis over a segment of output slices;
//   LDS       = a ring of 4 input slices, each staged ONCE per workgroup ((TY+2) x 34 positions x Cin channels as two fp16
//               pieces, [piece][8-channel group][position] x 16 B) and used by the 3 depth taps it takes part in, plus the whole
//               weight set as A fragments;
//   MFMA      = v_mfma_f32_16x16x32_f16, N = 16 consecutive pixels of a row, M = 16 output channels (Cout 16) or 2 adjacent
//               output slices x 8 channels (Cout 8: K then runs over 4 input slices, a quarter of it structural zeros — against
//               a half on the 32-row tile).  K is walked in units of 8 channels x one tap x one slice: the 4 lane groups of a
//               fragment each read their own unit (ds_read_b128, 16 lanes = 256 contiguous bytes), so any Cin that is a multiple
//               of 8 packs K densely.  A wave = 2 rows of the column (4 pixel groups; TY = 4: one row), the A fragments of a K
//               step are read once per wave and used for all of them; 3 products per step (x0*w1, x1*w0, x0*w0), all fragments
//               read one K step ahead;
//   pipeline  = the global loads of the next step's new slice(s) are issued before the MFMA walk of the current step and
//               written to LDS after it; one barrier per step (Cout 16) or two (Cout 8: the two new slices replace slices the
//               walk has just read).
// Stride 2: one output slice per step from 3 input slices (2 new ones per step); the slab keeps the even and the odd columns
// of a row apart ([row][column parity][column / 2]), so the 16 pixels of a fragment — 2 columns apart in the input — are again
// 256 contiguous bytes for every tap; Cout 32 = two M tiles that share every B fragment.
// Epilogue per output slice: acc * oscale, folded BatchNorm affine, LeakyReLU — conv3d's EPI_AFFINE.
#include "common.h"
#include "split_arith.h"

#include <cmath>
#include <cstring>

namespace nnd {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));

struct Slab3dArgs {
    const float* x0;  // (N, D+2, C0, H, W) depth-major, zero end slices
    const float* x1;  // (N, D+2, C1, H, W) or null
    const uint4* wq;  // [KT][M tile][2 pieces][64 lanes] A fragments
    const float* scale;  // [COUT], then shift [COUT], then oscale
    const float* shift;
    float* y;  // (N, Do+2, COUT, Ho, Wo); the end slices are zeroed by the host wrapper
    int C0, C1, D, H, W, Do, Ho, Wo, tiles_x, tiles_y, zseg, nseg, total;  // total = tiles_x * tiles_y * nseg * N workgroups
    float slope;
};

template <int CIN, int COUT, int STR>
struct Slab3dShape {
    static constexpr int CIN_ = CIN, COUT_ = COUT, STR_ = STR;
    static constexpr int OD = (STR == 1 && COUT == 8) ? 2 : 1;   // output slices per step
    static constexpr int MT = STR == 1 ? 1 : COUT / 16;          // M tiles of 16 output channels (stride 1: 16 channels or 2 slices x 8)
    static constexpr int NSL = STR == 1 ? OD + 2 : 3;            // input slices a step reads
    static constexpr int NEW = STR == 1 ? OD : 2;                // of them new in the next step
    static constexpr int NCG = CIN / 8;                          // 8-channel groups
    static constexpr int NQ = NSL * 9 * NCG;                     // K units (slice, tap, channel group)
    static constexpr int KT = (NQ + 3) / 4;                      // K steps of 4 units
    static constexpr int TX = (STR == 2 && CIN >= 16) ? 16 : 32; // output columns of the pixel column
    static constexpr int TY = STR == 1 ? (CIN >= 32 ? 4 : 8) : 4;  // output rows (LDS: 4 slices + weights <= 160 KB)
    static constexpr int ROWS = STR == 1 ? TY + 2 : 2 * TY + 1;  // slab rows / columns (an even number of columns: staged in pairs)
    static constexpr int COLS = STR == 1 ? TX + 2 : 2 * TX + 2;
    static constexpr int HC = COLS / 2;
    static constexpr int NPOS = ROWS * COLS;                     // staged positions per slice
    static constexpr int PIECE = NCG * NPOS * 16;                // bytes of one piece of a slice
    static constexpr int SLAB = 2 * PIECE;                       // bytes of a slice in the ring
    static constexpr int WBYTES = KT * MT * 2 * 64 * 16;         // A fragments
    static constexpr int LDS = 4 * SLAB + WBYTES;
    static constexpr int MINW = LDS <= 80 * 1024 ? 2 : 1;        // waves per SIMD the register budget is set for (2: two workgroups per CU)
    static constexpr int NG = TY * (TX / 16) / 4;                // pixel groups (16 px) per wave
    static constexpr int NPAIR = ROWS * HC;                      // staging pairs per slice and channel
    static_assert(LDS <= 160 * 1024, "slab3d: LDS");
    static_assert(NG >= 1 && COLS % 2 == 0, "slab3d: tile");
    // byte offset of slab position (row r, column c) inside a (piece, channel group) block
    __host__ __device__ static constexpr int pos(int r, int c) { return (STR == 1 ? r * COLS + c : (r * 2 + (c & 1)) * HC + (c >> 1)) * 16; }
};

template <int CIN, int COUT, int STR, bool TWO>
__global__ void __launch_bounds__(256, (Slab3dShape<CIN, COUT, STR>::MINW)) slab3d_kernel(Slab3dArgs a) {
    using S = Slab3dShape<CIN, COUT, STR>;
    constexpr int OD = S::OD, MT = S::MT, NSL = S::NSL, NEW = S::NEW, NCG = S::NCG, NQ = S::NQ, KT = S::KT, TX = S::TX, TY = S::TY,
                  NPOS = S::NPOS, NG = S::NG, HC = S::HC;
    constexpr int NSRC = TWO ? 2 : 1, CS = CIN / NSRC;        // channels per source (a concat splits in the middle)
    constexpr int NUS = S::NPAIR * (CS / 4);                  // staging units per slice and source
    constexpr int NRS = (NUS + 255) / 256, NR = NSRC * NRS;   // rounds per source / per slice
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // [4 slices][2 pieces][NCG][NPOS] x 16 B | weights
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, l15 = lane & 15;
    // XCD-aware order: workgroup ids go round-robin over the 8 XCDs (each with its own L2), so XCD j takes the j-th eighth of the
    // columns in (x fastest, y, segment, sample) order: columns that share halo lines / rows run on the same L2 at about the same time
    const int per_xcd = (a.total + 7) / 8;
    int wg = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (wg >= a.total) return;
    const int x0 = (wg % a.tiles_x) * TX;  // output coordinates of the column
    wg /= a.tiles_x;
    const int y0 = (wg % a.tiles_y) * TY;
    wg /= a.tiles_y;
    const int seg = wg % a.nseg, n = wg / a.nseg;
    const int z0 = seg * a.zseg, z1 = min(z0 + a.zseg, a.Do);  // output slices [z0, z1)
    const long HW = (long)a.H * a.W, HWo = (long)a.Ho * a.Wo;

    // ---- weights -> LDS (stay for the whole march).  In registers they would save a fifth of the LDS reads, but 8 VGPRs per K
    // step (144 .. 216) on top of the B fragments and the slice in flight exceed the 256 architectural VGPRs: the allocator then
    // parks every staged value in an AGPR behind its own s_waitcnt vmcnt(0) (measured: slower than this).
    const uint4* wl = reinterpret_cast<const uint4*>(lds + 4 * S::SLAB) + lane;
    for (int i = tid; i < KT * MT * 2 * 64; i += 256) reinterpret_cast<uint4*>(lds + 4 * S::SLAB)[i] = a.wq[i];

    // ---- staging role: unit = (4-channel group, a pair of adjacent columns 2k, 2k+1 of a slab row); round r of source r / NRS
    // takes units tid + 256 (r % NRS).  A unit is 4 buffer_load_dwordx2 (descriptor = this sample's volume of the source,
    // soffset = the (slice, channel j) plane: SGPRs; voffset = [4-channel group, position]: one VGPR).  A position outside the
    // image reads a clamped row / the neighbouring elements (out of the descriptor's range: 0) and is zeroed by the scale of the
    // split (x * 0 instead of x * 2^XSHIFT): no select, no divergent branch, no 64-bit address math.  Slices beyond D + 1 (odd
    // D) read the zero end slice.
    unsigned uoff[NR];  // byte offset [4-channel group][clamped row][x of the pair's first position] inside a slice of the source
    float usc[NR][2];   // 2^XSHIFT, or 0 outside the image, per position of the pair
    int ulds[NR];       // byte offset of the pair's first position inside a piece ([8-channel group][position] x 16 B, + 8 B for
                        // the odd 4-channel group)
    bool ush[NR];       // the pair starts one column left of the image: it is loaded one element to the right (a voffset of -4
                        // bytes would put the whole dwordx2 out of the descriptor's range and lose column 0 with it)
    bool uok[NR];       // there is a unit
    const unsigned plane = 4u * (unsigned)HW, plane_o = 4u * (unsigned)HWo;
    const float xscale = a.scale[2 * COUT + SPLIT_TAIL_XSCALE];  // the layer's activation scale (split_arith.h)
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        // unit order: pairs along the slab rows of one channel plane fastest (a wave-wide load then touches ~10 cache lines; with
        // the two 4-channel halves of a packet in adjacent lanes — conflict-free LDS writes — it touches ~15 and the issue of a
        // step's loads went from 3.4 k to 5.5 k cycles)
        const int u = tid + 256 * (r % NRS), src = r / NRS;
        const int qg = u / S::NPAIR, pr = u % S::NPAIR, py = pr / HC, px = 2 * (pr % HC);  // px: slab column of the first position
        const int gy = STR * y0 - 1 + py, gx = STR * x0 - 1 + px;
        const bool oky = u < NUS && gy >= 0 && gy < a.H;
        ush[r] = gx < 0;
        uok[r] = u < NUS;
        uoff[r] = (unsigned)qg * 4u * plane + 4u * (unsigned)(min(max(gy, 0), a.H - 1) * a.W + max(gx, 0));
        const int c = src * CS + qg * 4;  // channel of the concat
        ulds[r] = (c >> 3) * NPOS * 16 + S::pos(py, px) + ((c >> 2) & 1) * 8;
#pragma unroll
        for (int e = 0; e < 2; ++e) usc[r][e] = oky && gx + e >= 0 && gx + e < a.W ? xscale : 0.f;
    }
    constexpr int ESTEP = S::pos(0, 1) - S::pos(0, 0);  // from the pair's first position to its second
    const auto rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x0) + (long)n * (a.D + 2) * CS * HW, 0,
                                                       (int)((unsigned)(a.D + 2) * CS * plane), 0x00020000);
    const auto rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(TWO ? a.x1 : a.x0) + (long)n * (a.D + 2) * CS * HW, 0,
                                                       (int)((unsigned)(a.D + 2) * CS * plane), 0x00020000);
    // one unit of slice p: issue its 4 loads / split what they returned into the two fp16 pieces (in place: 8 registers either
    // way) / write the pieces to the ring
    struct Unit {
        uint2 w[4];  // loaded: channel j -> (position 0, position 1) as fp32; split: [2 * position + piece] -> 4 channels as fp16
    };
    auto load_unit = [&](Unit& un, int r, int p) {
        const unsigned so = (unsigned)min(p, a.D + 1) * CS * plane;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            un.w[j] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(r / NRS ? rs1 : rs0, uoff[r], so + j * plane, 0));
    };
    auto split_unit = [&](Unit& un, int r) {
        f32x2v v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = __builtin_bit_cast(f32x2v, un.w[j]);
            v[j][1] = ush[r] ? v[j][0] : v[j][1];  // loaded one element to the right: column 0 is the pair's second position
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            // round-to-nearest of the running residual (split_arith.h), two channels per instruction: v_pk_mul_f32,
            // v_cvt_pk_f16_f32, v_pk_add_f32
            f16x2v hi[2], lo[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                f32x2v x2 = f32x2v{v[2 * k][e], v[2 * k + 1][e]} * usc[r][e];
                hi[k] = __builtin_convertvector(x2, f16x2v);
                x2 -= __builtin_convertvector(hi[k], f32x2v);
                lo[k] = __builtin_convertvector(x2, f16x2v);
            }
            un.w[2 * e + 0] = __builtin_bit_cast(uint2, hi);
            un.w[2 * e + 1] = __builtin_bit_cast(uint2, lo);
        }
    };
    auto write_unit = [&](const Unit& un, int r, int p) {
        unsigned char* dst = lds + (p & 3) * S::SLAB + ulds[r];
        if (!uok[r]) return;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            *reinterpret_cast<uint2*>(dst + e * ESTEP) = un.w[2 * e + 0];
            *reinterpret_cast<uint2*>(dst + e * ESTEP + S::PIECE) = un.w[2 * e + 1];
        }
    };

    // ---- MFMA role: K unit of this lane group at step t -> (slice, byte offset inside a piece); pixel groups of the wave
    int ksl[KT], koff[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        const int qu = min(4 * t + q, NQ - 1);  // units past the end (e.g. Cin 16 -> Cout 16: 54 units in 14 steps) carry zero weights
        const int cg = qu % NCG, tap = (qu / NCG) % 9, kh = tap / 3, kw = tap % 3;
        ksl[t] = qu / (9 * NCG);
        // the pixel l15 of a group sits STR columns further per pixel: in the parity-split slab of stride 2 that is one position
        koff[t] = cg * NPOS * 16 + S::pos(kh, kw) + l15 * 16;
    }
    int goff[NG];   // byte offset of pixel group g of this wave: output row grow[g], columns gcol[g] .. + 15
    int grow[NG], gcol[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int gi = wave * NG + g;
        grow[g] = gi / (TX / 16);
        gcol[g] = (gi % (TX / 16)) * 16;
        goff[g] = S::pos(STR * grow[g], STR * gcol[g]);
    }

    // per-lane epilogue constants: rows q*4 + i of M tile mt -> (output slice od, channel co)
    float sc[MT][4], sh[MT][4];
    unsigned orow[MT][4];
    const float oscale = a.scale[2 * COUT + SPLIT_TAIL_OSCALE];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = q * 4 + i, od = OD == 2 ? (m >> 3) : 0, co = OD == 2 ? (m & 7) : mt * 16 + m;
            sc[mt][i] = a.scale[co];
            sh[mt][i] = a.shift[co];
            orow[mt][i] = (unsigned)(od * COUT + co) * plane_o;
        }
    unsigned opix[NG];
    bool gok[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int gy = y0 + grow[g], gx = x0 + gcol[g] + l15;
        gok[g] = gy < a.Ho && gx < a.Wo;
        opix[g] = 4u * (unsigned)(min(gy, a.Ho - 1) * a.Wo + min(gx, a.Wo - 1));
    }
    const auto rsy = __builtin_amdgcn_make_buffer_rsrc(a.y + (long)n * (a.Do + 2) * COUT * HWo, 0, (int)((unsigned)(a.Do + 2) * COUT * plane_o),
                                                       0x00020000);

    Unit un[NEW][NR];
    // prologue: all NSL slices of the first step (padded input slices STR * z0 ..)
    for (int p = STR * z0; p < STR * z0 + NSL; p += NEW) {
#pragma unroll
        for (int o = 0; o < NEW; ++o)
#pragma unroll
            for (int r = 0; r < NR; ++r) load_unit(un[o][r], r, p + o);
#pragma unroll
        for (int o = 0; o < NEW; ++o) {
            if (p + o >= STR * z0 + NSL) break;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                split_unit(un[o][r], r);
                write_unit(un[o][r], r, p + o);
            }
        }
    }
    __syncthreads();

#ifdef NND_SLAB3D_STAMPS
    long long* stamps = reinterpret_cast<long long*>(a.y);  // debug build: phase stamps of workgroup 9 into the zero end slice
    const bool stamp = blockIdx.x == 9 && tid == 0;
    int sidx = 0;
#define S3_STAMP() do { if (stamp) stamps[sidx++] = clock64(); } while (0)
#else
#define S3_STAMP() do {} while (0)
#endif
    // Order of a step: MFMA walk | (barrier where the new slices replace slices the walk has read) | split + LDS writes of the
    // slices loaded during the walk | loads of the step after the next | affine + stores of this step's outputs | barrier.  The
    // loads are issued behind the writes that free their registers and in front of the output stores, and have the whole next
    // walk to return.  (Hanging the loads, the stores and the splits INTO the walk, one unit per K step, was measured slower —
    // 13.8 k cycles for the walk against 5.5 k + 3.4 k + 1.3 k + 1.6 k apart: with one wave per SIMD every stall at a full
    // vector-memory queue also stops the wave's LDS reads and MFMAs, scripts/stamps_slab3d.py.)
    if (z0 + OD < z1) {
#pragma unroll
        for (int o = 0; o < NEW; ++o)
#pragma unroll
            for (int r = 0; r < NR; ++r) load_unit(un[o][r], r, STR * z0 + NSL + o);
    }
    for (int d = z0; d < z1; d += OD) {
        const int base = STR * d;       // first padded input slice of the step; the next step's new slices: base + NSL + o
        const bool more = d + OD < z1;  // `un` holds them (in flight during the walk)
        S3_STAMP();
        f32x4 acc[MT][NG];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[mt][g] = f32x4{0.f, 0.f, 0.f, 0.f};
        // all fragments one K step ahead in a second register set; the order is pinned (left alone, the scheduler issues the reads
        // of a step right in front of its MFMAs and every MFMA waits for LDS)
        f16x8s bx[2][2][NG], aw[2][MT][2];
        auto read_ab = [&](int t, f16x8s (&b)[2][NG]) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                aw[t & 1][mt][0] = __builtin_bit_cast(f16x8s, wl[((t * MT + mt) * 2 + 0) * 64]);
                aw[t & 1][mt][1] = __builtin_bit_cast(f16x8s, wl[((t * MT + mt) * 2 + 1) * 64]);
            }
            const unsigned char* bp = lds + ((base + ksl[t]) & 3) * S::SLAB + koff[t];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                b[0][g] = __builtin_bit_cast(f16x8s, *reinterpret_cast<const uint4*>(bp + goff[g]));
                b[1][g] = __builtin_bit_cast(f16x8s, *reinterpret_cast<const uint4*>(bp + goff[g] + S::PIECE));
            }
        };
        read_ab(0, bx[0]);
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            if (t + 1 < KT) read_ab(t + 1, bx[(t + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const f16x8s(&b)[2][NG] = bx[t & 1];
            // products with i + j descending (split_arith.h), interleaved over the M tiles and pixel groups
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[mt][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aw[t & 1][mt][1], b[0][g], acc[mt][g], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[mt][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aw[t & 1][mt][0], b[1][g], acc[mt][g], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[mt][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aw[t & 1][mt][0], b[0][g], acc[mt][g], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        S3_STAMP();
        if (more) {
            if (NSL + NEW > 4) __syncthreads();  // the new slices replace slices this step's walk has read
            S3_STAMP();
#pragma unroll
            for (int o = 0; o < NEW; ++o)
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    split_unit(un[o][r], r);
                    write_unit(un[o][r], r, base + NSL + o);
                }
            S3_STAMP();
            if (d + 2 * OD < z1) {
#pragma unroll
                for (int o = 0; o < NEW; ++o)
#pragma unroll
                    for (int r = 0; r < NR; ++r) load_unit(un[o][r], r, base + NEW + NSL + o);
            }
        }
        S3_STAMP();
        // ---- epilogue: D[row q*4 + i][col l15] -> acc * oscale, folded BatchNorm affine, LeakyReLU
        const unsigned yo = (unsigned)(d + 1) * COUT * plane_o;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int od = OD == 2 ? ((q * 4 + i) >> 3) : 0;
                    float r = acc[mt][g][i] * oscale;
                    r = fmaf(r, sc[mt][i], sh[mt][i]);
                    r = r > 0.f ? r : a.slope * r;
                    if (gok[g] && d + od < z1) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, r), rsy, orow[mt][i] + opix[g], yo, 0);
                }
        S3_STAMP();
        __syncthreads();
    }
    S3_STAMP();
}

template <int CIN, int COUT, int STR>
int launch_slab3d(const Slab3dArgs& a0, int N, hipStream_t s) {
    using S = Slab3dShape<CIN, COUT, STR>;
    Slab3dArgs a = a0;
    const bool two = a.C1 > 0;
    NND_REQUIRE(!two || (CIN == 32 && STR == 1 && a.C0 == a.C1), "slab3d: a concat is built for 16 + 16 channels at stride 1");
    a.tiles_x = cdiv(a.Wo, S::TX);
    a.tiles_y = cdiv(a.Ho, S::TY);
    const int tiles = a.tiles_x * a.tiles_y;
    // depth segments: every segment re-stages 2 slices (+ its prologue), and the grid runs in ceil(workgroups / resident set)
    // rounds (LDS: 1 workgroup per CU, 2 for the smallest layers) of a segment's length each — the segment count with the
    // smallest product (NND_SLAB3D_ROUNDS, diagnostic: about that many rounds instead)
    const int resident = 256 * std::min(S::MINW, (int)((160 * 1024) / S::LDS));
    int nseg = 1;
    if (switches().slab3d_rounds > 0) {
        nseg = (int)std::min<long>(cdiv(a.Do, S::OD), std::max<long>(1, ((long)switches().slab3d_rounds * resident) / ((long)tiles * N)));
    } else {
        long best = -1;
        for (int c = 1; c <= 16 && c <= cdiv(a.Do, S::OD); ++c) {
            const int zs = cdiv(cdiv(a.Do, c), S::OD) * S::OD;
            const long wgs = (long)tiles * N * cdiv(a.Do, zs), cost = cdiv64(wgs, resident) * (zs / S::OD + 3);
            if (best < 0 || cost < best) best = cost, nseg = c;
        }
    }
    a.zseg = cdiv(cdiv(a.Do, nseg), S::OD) * S::OD;
    a.nseg = cdiv(a.Do, a.zseg);
    const long total = (long)tiles * a.nseg * N;
    NND_REQUIRE(total < (1L << 30), "slab3d: grid limit");
    a.total = (int)total;
    const dim3 grid(8 * cdiv((int)total, 8));
    if constexpr (CIN == 32 && STR == 1) {
        if (two) {
            static std::atomic<unsigned> raised2{0};
            if (int rc = raise_lds_limit(reinterpret_cast<const void*>(slab3d_kernel<CIN, COUT, STR, true>), raised2)) return rc;
            hipLaunchKernelGGL((slab3d_kernel<CIN, COUT, STR, true>), grid, dim3(256), S::LDS, s, a);
            NND_LAUNCH_CHECK();
            return NND_OK;
        }
    }
    static std::atomic<unsigned> raised{0};
    if (int rc = raise_lds_limit(reinterpret_cast<const void*>(slab3d_kernel<CIN, COUT, STR, false>), raised)) return rc;
    hipLaunchKernelGGL((slab3d_kernel<CIN, COUT, STR, false>), grid, dim3(256), S::LDS, s, a);
    NND_LAUNCH_CHECK();
    return NND_OK;
}

uint16_t f16_bits(float x) {  // round to nearest even, overflow -> inf (weights are range-scaled below that)
    _Float16 h = (_Float16)x;
    uint16_t b;
    std::memcpy(&b, &h, 2);
    return b;
}
float f16_val(uint16_t b) {
    _Float16 h;
    std::memcpy(&h, &b, 2);
    return (float)h;
}

template <int CIN, int COUT, int STR>
void pack_slab3d(const float* w, float wscale, uint16_t* out) {
    using S = Slab3dShape<CIN, COUT, STR>;
    for (int t = 0; t < S::KT; ++t)
        for (int mt = 0; mt < S::MT; ++mt)
            for (int lane = 0; lane < 64; ++lane) {
                const int m = lane & 15, qq = lane >> 4, qu = 4 * t + qq;
                for (int j = 0; j < 8; ++j) {
                    float val = 0.f;
                    if (qu < S::NQ) {
                        const int cg = qu % S::NCG, tap = (qu / S::NCG) % 9, sl = qu / (9 * S::NCG);
                        const int od = S::OD == 2 ? (m >> 3) : 0, co = S::OD == 2 ? (m & 7) : mt * 16 + m, kd = sl - od, ci = cg * 8 + j;
                        if (kd >= 0 && kd <= 2) val = w[(((size_t)co * CIN + ci) * 3 + kd) * 9 + tap] * wscale;
                    }
                    float res = val;
                    for (int pc = 0; pc < 2; ++pc) {
                        const uint16_t b = f16_bits(res);
                        res -= f16_val(b);
                        out[((((size_t)t * S::MT + mt) * 2 + pc) * 64 + lane) * 8 + j] = b;
                    }
                }
            }
}

template <typename F>
bool slab3d_dispatch(int Cout, int Ct, int stride, F&& f) {
    if (stride == 1) {
        if (Ct == 16 && Cout == 8) return f(Slab3dShape<16, 8, 1>{}), true;
        if (Ct == 8 && Cout == 8) return f(Slab3dShape<8, 8, 1>{}), true;
        if (Ct == 32 && Cout == 16) return f(Slab3dShape<32, 16, 1>{}), true;
        if (Ct == 16 && Cout == 16) return f(Slab3dShape<16, 16, 1>{}), true;
    } else if (stride == 2) {
        if (Ct == 8 && Cout == 16) return f(Slab3dShape<8, 16, 2>{}), true;
        if (Ct == 16 && Cout == 32) return f(Slab3dShape<16, 32, 2>{}), true;
    }
    return false;
}

}  // namespace

// the (Cin, Cout) pairs of the regulariser's thin layers — stride 1: conv1_up 16->8, final_conv 8->8, conv2_up / proj_2 32->16
// (a concat of 16 + 16), conv1.1 16->16; stride 2: conv1.0 8->16, conv2.0 16->32
bool slab3d_supported(int Cout, int C0, int C1, int stride, int arith) {
    if (arith != 2 || C0 % 8 != 0 || C1 % 8 != 0 || (C1 > 0 && !(stride == 1 && C0 == 16 && C1 == 16))) return false;
    return slab3d_dispatch(Cout, C0 + C1, stride, [](auto) {});
}

// packed: [A fragments (KT x MT x 2 x 64 x 16 B) | scale (Cout) | shift (Cout) | oscale, xscale, amax, 2^-s (split_arith.h: SPLIT_TAIL_*)]
int64_t slab3d_packed_floats(int Cout, int Ct, int stride) {
    int64_t frag = 0;
    slab3d_dispatch(Cout, Ct, stride, [&](auto S) { frag = (int64_t)decltype(S)::KT * decltype(S)::MT; });
    return frag ? frag * 2 * 64 * 4 + 2 * Cout + 4 : 0;
}

void slab3d_pack(int Cout, int Ct, int stride, const float* w /* (Cout, Ct, 3,3,3) */, const float* scale, const float* shift, float* out) {
    float wmax = 0.f;
    for (size_t i = 0; i < (size_t)Cout * Ct * 27; ++i)
        if (std::isfinite(w[i]) && std::fabs(w[i]) > wmax) wmax = std::fabs(w[i]);
    int e = 0;
    if (wmax > 0.f) std::frexp(wmax, &e);  // as pack_conv_split: max|w| * 2^s in [2^13, 2^14)
    const int sft = wmax > 0.f ? 14 - e : 0;
    const float wscale = std::ldexp(1.f, sft);
    int64_t frag = 0;
    slab3d_dispatch(Cout, Ct, stride, [&](auto S) {
        using T = decltype(S);
        pack_slab3d<T::CIN_, T::COUT_, T::STR_>(w, wscale, reinterpret_cast<uint16_t*>(out));
        frag = (int64_t)T::KT * T::MT;
    });
    float* tail = out + frag * 2 * 64 * 4;
    for (int co = 0; co < Cout; ++co) {
        tail[co] = scale[co];
        tail[Cout + co] = shift[co];
    }
    float* st = tail + 2 * Cout;  // split_arith.h: SPLIT_TAIL_*
    st[SPLIT_TAIL_OSCALE] = std::ldexp(1.f, -(sft + SPLIT_F16_XSHIFT));
    st[SPLIT_TAIL_XSCALE] = std::ldexp(1.f, SPLIT_F16_XSHIFT);
    st[SPLIT_TAIL_AMAX] = 0.f;
    st[SPLIT_TAIL_WSINV] = std::ldexp(1.f, -sft);
}

int slab3d_forward(int Cout, int C0, int C1, int stride, const float* packed, const float* x0, const float* x1, float* y, int N, int D,
                   int H, int W, float slope, hipStream_t s) {
    NND_REQUIRE(slab3d_supported(Cout, C0, C1, stride, 2), "slab3d: %d+%d -> %d stride %d not built", C0, C1, Cout, stride);
    const int Do = (D + stride - 1) / stride, Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
    NND_REQUIRE((long)(D + 2) * std::max(C0, C1) * H * W * 4 < (1L << 32) && (long)(Do + 2) * Cout * Ho * Wo * 4 < (1L << 32),
                "slab3d: a sample's volume exceeds the 4 GB of a buffer descriptor");
    const int Ct = C0 + C1;
    int64_t frag = 0;
    slab3d_dispatch(Cout, Ct, stride, [&](auto S) { frag = (int64_t)decltype(S)::KT * decltype(S)::MT; });
    Slab3dArgs a{};
    a.x0 = x0; a.x1 = x1; a.C0 = C0; a.C1 = C1;
    a.wq = reinterpret_cast<const uint4*>(packed);
    a.scale = packed + frag * 2 * 64 * 4;
    a.shift = a.scale + Cout;
    a.y = y;
    a.D = D; a.H = H; a.W = W;
    a.Do = Do; a.Ho = Ho; a.Wo = Wo;
    a.slope = slope;
    if (calibrating()) {  // record the largest |x| of the volumes this launch stages (calib.hip); the zero end slices do not matter
        const float* tail = a.scale + 2 * Cout;
        if (int rc2 = calib_amax_flat(x0, (int64_t)N * (D + 2) * C0 * H * W, tail, s)) return rc2;
        if (C1 > 0)
            if (int rc2 = calib_amax_flat(x1, (int64_t)N * (D + 2) * C1 * H * W, tail, s)) return rc2;
    }
    int rc = NND_ERR_INVALID;
    slab3d_dispatch(Cout, Ct, stride, [&](auto S) { rc = launch_slab3d<decltype(S)::CIN_, decltype(S)::COUT_, decltype(S)::STR_>(a, N, s); });
    return rc;
}

}  // namespace nnd
