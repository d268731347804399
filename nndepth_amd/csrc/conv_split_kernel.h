// conv_split_kernel: implicit-GEMM 2-D convolution (stride 1, zero "same" padding) on the gfx950 16-bit MFMA with fp32
// operands carried as split 16-bit pieces (split_arith.h).  Design notes: conv_split.hip (header comment), DESIGN.md §4.
// This header holds the kernel template and its launcher; it is instantiated per arithmetic in conv_split_ns2.hip (fp16x2)
// and conv_split_ns3.hip (bf16x3) so that the two sets compile in parallel.
//
// Template parameters
//   KH, KW  kernel shape (3x3, 1x5, 5x1, 1x1)
//   NS      pieces per fp32 operand (3: bf16, 2: range-scaled fp16)
//   P       4x8-pixel sub-tiles per wave (each weight fragment feeds P MFMA groups): 2, 3 or 4
//   NU      staging units per thread (unit = patch position x 8 channels)
//   FAST    decided by the host: every super-chunk of ks*16 channels full and inside one source (SRC4: sources tile-major with
//           4 channels interleaved — the refinement loops' tensors; else planar, tile-major or NCHW — the encoder's, the C-ABI's).
//           Then nothing in the K loop is
//           a run-time branch — the walk over (tap, sub-tile) is straight-line code whose waits the compiler can count exactly
//           (round 2's kernel guarded every MFMA group with a wave-uniform `mine` test and chose the load form per unit at run
//           time: the walk was cut into ~20 basic blocks, each entered with conservative waits; round 3, profiles/r03_*)
//   AD      steps (taps) of lookahead of the weight-fragment ring
//   MAXT    launch bound (768: 3 waves per SIMD / 168 VGPRs; 512: 2 waves per SIMD / 256 VGPRs for the P >= 3 shapes)
//   STR     stride 1 or 2 (2: 3x3 and 1x1, FAST regime with planar sources only — the encoder's and the regulariser's down-sampling
//           layers).  The LDS patch of a stride-2 sub-tile is stored as its 4 (row, column) parity phases, each a dense grid with the
//           stride-1 row / position strides: the operand of tap (dy, dx) for output pixel (r, c) — input (2r + dy, 2c + dx) — is
//           then position (r + dy/2, c + dx/2) of phase (dy%2, dx%2): lane base + immediate with unit lane stride, conflict-free
//           like the stride-1 patch.  A 1x1 stride-2 conv touches phase (0, 0) only and stages just those 4x8 positions.
#pragma once
#include "common.h"
#include "conv_epilogue.h"
#include "layout.h"
#include "split_arith.h"

#include <type_traits>
#include <utility>

namespace nnd {

#ifndef NND_SPLIT_BSLOTS
#define NND_SPLIT_BSLOTS 2  // register sets of the activation fragments (lookahead = sets - 1 units of NPROD MFMAs)
#endif

#ifdef NND_DBG_STAMPS
// debug build only: per-workgroup phase timestamps (s_memrealtime, 100 MHz) for scripts/ablate_split.py; one copy per
// instantiation unit (read back through nnd_debug_read_split_stamps_ns2 / _ns3)
static __device__ unsigned long long g_split_stamps[4096 * 8];
#define NND_SSTAMP(i)                                                                                  \
    do {                                                                                               \
        if (threadIdx.x == 0 && a.dbg_stamp) {                                                         \
            const unsigned lin_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);      \
            if (lin_ < 4096) g_split_stamps[lin_ * 8 + (i)] = __builtin_amdgcn_s_memrealtime();        \
        }                                                                                              \
    } while (0)
// shader-clock stamps (s_memtime) in slots 5 / 6 next to the real-time stamps 1 / 2: in-kernel clock of the K loop
#define NND_SCLOCK(i)                                                                                  \
    do {                                                                                               \
        if (threadIdx.x == 0 && a.dbg_stamp) {                                                         \
            const unsigned lin_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);      \
            if (lin_ < 4096) g_split_stamps[lin_ * 8 + (i)] = __builtin_amdgcn_s_memtime();            \
        }                                                                                              \
    } while (0)
#else
#define NND_SSTAMP(i)
#define NND_SCLOCK(i)
#endif
// -DNND_DBG_STAMPS -DNND_DBG_PROLOGUE (scripts/stamps_prologue.py): slots 5 / 6 / 7 take real-time stamps INSIDE the prologue — the
// units decoded | weight ring and first patch requested | first patch split and stored (in front of the barrier) — instead of the
// K loop's shader-clock pair
#if defined(NND_DBG_STAMPS) && defined(NND_DBG_PROLOGUE)
#undef NND_SCLOCK
#define NND_SCLOCK(i)
#define NND_PSTAMP(i) NND_SSTAMP(i)
#else
#define NND_PSTAMP(i)
#endif

__host__ __device__ constexpr int split_pos_bytes(int NS) { return NS * 32 + 16; }
__host__ __device__ constexpr int split_row_bytes(int PC, int NS) {
    int rb = (PC * split_pos_bytes(NS) + 15) / 16;
    while (rb % 16 != 8) ++rb;
    return rb * 16;
}

// lane (0..31 of a half-wave) -> pixel index r*8 + c of the 4x8 sub-tile: the two lane groups that ds_read_b128 serves in
// separate cycles get rows {0,1} and rows {2,3}
__device__ __forceinline__ int lane_pixel(int l31) {
    const bool g0 = (l31 < 4) || (l31 >= 12 && l31 < 16) || (l31 >= 20 && l31 < 28);
    const int idx = g0 ? (l31 < 4 ? l31 : (l31 < 16 ? l31 - 8 : l31 - 12)) : (l31 < 12 ? l31 - 4 : (l31 < 20 ? l31 - 8 : l31 - 16));
    return (g0 ? 0 : 16) + idx;
}

__host__ __device__ constexpr int split_gcd(int a, int b) { return b == 0 ? a : split_gcd(b, a % b); }

template <int N, typename F, int... I>
__device__ __forceinline__ void split_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void split_static_for(F&& f) {
    split_static_for_impl<N>(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// LDS geometry of one sub-tile's patch (bytes): stride 1: PR x PC positions; stride 2: 4 phases (1 for a 1x1) of PRP x PCP
template <int KH, int KW, int NS, int STR>
struct SplitGeom {
    static constexpr int PRI = 3 * STR + KH, PCI = 7 * STR + KW;                 // input positions covered by a 4x8 output sub-tile
    static constexpr int NPH = STR == 1 ? 1 : (KH * KW == 1 ? 1 : 4);            // parity phases stored
    static constexpr int PRP = STR == 1 ? PRI : (KH == 1 ? 4 : (PRI + 1) / 2);   // rows / columns of a phase grid
    static constexpr int PCP = STR == 1 ? PCI : (KW == 1 ? 8 : (PCI + 1) / 2);
    static constexpr int NPOS = STR == 1 ? PRI * PCI : (KH * KW == 1 ? 32 : PRI * PCI);  // positions staged per sub-tile
    static constexpr int PS = split_pos_bytes(NS), ROWB = split_row_bytes(PCP, NS), PHB = PRP * ROWB, SUBB = NPH * PHB;
};

// Schedule of the PIPELINED walk (round 4, FAST regime, kernels with >= 5 taps): the staging of super-chunk K + 1 is cut into
// micro-step slots that hang behind the MFMA groups of the units [U0, NUNIT - 2] of super-chunk K, SPU slots per unit; slot q
// converts one (unit, channel pair) — q < 4 NU — and stores one 16-byte piece of the unit converted four slots earlier (one
// ds_write_b128 per slot: 13 cycles of the CU's store path each, so a workgroup's 48 stores never pile up in front of the
// barrier); UL = the unit behind whose MFMAs the global loads of K + 1 are issued.
__host__ __device__ constexpr int split_pipe_spu(int nunit, int nsteps, int ul) {
    int v = 1;
    while (v < nsteps && nunit - 1 - (nsteps + v - 1) / v < ul + 3) ++v;
    return v;
}
template <int NT, int P, int NU>
struct SplitPipe {
    static constexpr int NUNIT = NT * P, NSTEPS = NU * 4 + 3, UL = 1;  // slots: the last unit's pieces (<= 3) follow its conversions
    static constexpr int SPU = split_pipe_spu(NUNIT, NSTEPS, UL), U0 = NUNIT - 1 - (NSTEPS + SPU - 1) / SPU;
    static constexpr bool ON = NT >= 5 && P == 2;  // NUNIT even: the B-fragment slot rotation continues across super-chunks
};

template <int KH, int KW, int NS, int P, int NU, bool FAST, bool SRC4, int AD_, int MAXT, int STR = 1>
__global__ void __launch_bounds__(MAXT) conv_split_kernel(ConvArgs a) {
    static_assert(STR == 1 || (FAST && (KH * KW == 9 || KH * KW == 1)), "stride 2: 3x3 / 1x1, FAST regime");
    using Geo = SplitGeom<KH, KW, NS, STR>;
    constexpr int NT = KH * KW, PH = KH / 2, PW = KW / 2;
    constexpr int PC = Geo::PCI, NPOS = Geo::NPOS;
#ifndef NND_SPLIT_NO_PIPE
    // the pipelined walk below (round 4) — for 4-channel-interleaved sources (a staging unit is two 16-byte loads).  With planar
    // sources a unit is eight 4-byte loads: issued as one burst behind unit UL they queue in front of the weight ring, and the
    // regulariser's grouped layers (NU = 4: 32 loads per thread) measured slower than with round 3's walk (conv3_up 142 -> 185 us,
    // conv3.1 72 -> 82: profiles/r04_igev_regulariser_layers_fp16x2.txt vs r03_*), so those keep it.
    constexpr bool PIPE = FAST && SRC4 && SplitPipe<NT, P, NU>::ON;
#else
    constexpr bool PIPE = false;
#endif
    constexpr int PS = Geo::PS, ROWB = Geo::ROWB, SUBB = Geo::SUBB, PHB = Geo::PHB;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    NND_SSTAMP(0);

    const int tid = threadIdx.x, lane = tid & 63, nthreads = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = a.wco, ks = a.ks;
    const int cbi = wave % wco, kj = wave / wco;
    const int h2 = lane >> 5, l31 = lane & 31;
    const int pxl = lane_pixel(l31), r = pxl >> 3, c = pxl & 7;
    const int cb = blockIdx.y * wco + cbi;
    const bool active = cb * 32 < a.Cout;
    const int b = blockIdx.z;
    const int Hin = a.Hin, Win = a.Win;
    const long SP = a.ls.plane;
    const int SCH = ks * 16;  // channels per super-chunk
    // fp16x2: the power-of-two scale that undoes the range scaling of both operands (packed behind the bias), requested now
    float oscale = 1.f, xscale = 1.f;  // xscale: the layer's activation scale (calibrated per layer, split_arith.h)
    if constexpr (NS == 2) {
        const float* tail = a.bias + (((a.Cout + 31) >> 5) << 5);
        oscale = tail[SPLIT_TAIL_OSCALE];
        xscale = tail[SPLIT_TAIL_XSCALE];
    }

    int ty0[P], tx0[P];
#pragma unroll
    for (int pp = 0; pp < P; ++pp) {
        const int t = blockIdx.x * P + pp;
        const bool valid = t < a.npos;  // npos = number of sub-tiles of one image
        ty0[pp] = valid ? (t / a.tiles_x) * 4 : (1 << 20);  // an absent sub-tile lies outside the image: staged as zeros, never stored
        tx0[pp] = valid ? (t % a.tiles_x) * 8 : 0;
    }

    // ---- staging units of this thread: unit = (sub-tile, patch position, 8 consecutive channels of the super-chunk)
    const int nunits = P * NPOS * 2 * ks;
    int goff[NU], loff[NU], cho[NU];
    bool inimg[NU], own[NU];
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        // PIPE: a thread past the last unit repeats the last one (the same values to the same LDS address: harmless), so that the
        // staging micro-steps are branch-free — behind an `own` test the compiler sinks all conversions down to the guarded LDS write
        const int u = PIPE ? min(tid + i * nthreads, nunits - 1) : tid + i * nthreads;
        own[i] = u < nunits;
        const int pos = u % NPOS, rest = u / NPOS;
        const int pp = rest % P, oct = rest / P;
        int pr, pc, lpos;  // input position relative to the sub-tile's first input row / column, and its byte offset in the patch
        if constexpr (STR == 1) {
            pr = pos / PC;
            pc = pos - pr * PC;
            lpos = pr * ROWB + pc * PS;
        } else if constexpr (KH * KW == 1) {  // only the even (row, column) positions exist: phase (0, 0)
            pr = (pos >> 3) * 2;
            pc = (pos & 7) * 2;
            lpos = (pos >> 3) * ROWB + (pos & 7) * PS;
        } else {
            pr = pos / PC;
            pc = pos - pr * PC;
            lpos = ((pr & 1) * 2 + (pc & 1)) * PHB + (pr >> 1) * ROWB + (pc >> 1) * PS;
        }
        int gy = 0, gx = 0;
#pragma unroll
        for (int q = 0; q < P; ++q)
            if (q == pp) {
                gy = ty0[q] * STR + pr - PH;
                gx = tx0[q] * STR + pc - PW;
            }
        inimg[i] = own[i] && gy >= 0 && gy < Hin && gx >= 0 && gx < Win;
        goff[i] = inimg[i] ? (int)pix_off(a.ls, gy, gx) : 0;
        loff[i] = (oct >> 1) * (P * SUBB) + pp * SUBB + lpos + (oct & 1) * 16;
        cho[i] = oct * 8;
        if constexpr (FAST) goff[i] = inimg[i] ? cho[i] * (int)SP + goff[i] : 0;  // whole offset inside the super-chunk
    }

    NND_PSTAMP(5);
    f32x16 acc[P];
#pragma unroll
    for (int pp = 0; pp < P; ++pp)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[pp][i] = 0.f;

    const int nchunks = a.nchunks;  // 16-channel chunks
    const int nsuper = (nchunks + ks - 1) / ks;
    const uint4* wbase = reinterpret_cast<const uint4*>(a.wpk) + (size_t)(active ? cb : 0) * nchunks * (NT * NS * 64) + lane;
    float stage[NU][8];

    auto chunk_src = [&](int K, const float*& src, int& climit) {
        const int cbase = K * SCH;
        if constexpr (FAST) {  // every super-chunk lies inside one source: a select, no branch; climit is not needed
            const bool first = cbase < a.c0;
            const float* s0 = a.src0 + b * a.bs0 + (long)cbase * SP;
            const float* s1 = a.src1 + b * a.bs1 + (long)(cbase - a.c0) * SP;
            src = first ? s0 : s1;
            climit = SCH;
        } else if (cbase < a.c0) {
            src = a.src0 + b * a.bs0 + (long)cbase * SP;
            climit = a.c0 - cbase;
        } else {
            const int cc = cbase - a.c0;
            src = a.src1 + b * a.bs1 + (long)cc * SP;
            climit = a.c1 - cc;
        }
    };
    // unconditional loads with clamped addresses (element 0 when masked); the zero fill is a select in store_unit.
    const bool c4s = FAST ? SRC4 : a.ls.ci == 4;  // 4-channel-interleaved source: a unit's 8 channels are two 16-B loads
    auto load_unit = [&](int K, int i) {
        const float* src;
        int climit;
        chunk_src(K, src, climit);
        if constexpr (FAST && SRC4) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const float4 t = *reinterpret_cast<const float4*>(src + (unsigned)(goff[i] + (inimg[i] ? 4 * g * (int)SP : 0)));
                stage[i][4 * g] = t.x; stage[i][4 * g + 1] = t.y; stage[i][4 * g + 2] = t.z; stage[i][4 * g + 3] = t.w;
            }
        } else if constexpr (FAST) {  // planar tile-major source: 8 channel planes, one 4-B load each
            const int step = inimg[i] ? (int)SP : 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) stage[i][j] = src[(unsigned)(goff[i] + j * step)];
        } else if (c4s) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int ci = cho[i] + 4 * g;  // multiple of 4: channel group ci/4 starts at ci*SP, like a planar channel
                const float4 t = *reinterpret_cast<const float4*>(src + ((inimg[i] && ci < climit) ? (unsigned)(ci * (int)SP + goff[i]) : 0u));
                stage[i][4 * g] = t.x; stage[i][4 * g + 1] = t.y; stage[i][4 * g + 2] = t.z; stage[i][4 * g + 3] = t.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int ci = cho[i] + j;
                stage[i][j] = src[(inimg[i] && ci < climit) ? (unsigned)(ci * (int)SP + goff[i]) : 0u];
            }
        }
    };
    auto store_unit = [&](int K, int i) {
        const float* src;
        int climit;
        chunk_src(K, src, climit);
        unsigned char* buf = lds_raw + (K & 1) * (ks * P * SUBB);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (inimg[i] && (FAST || cho[i] + j < climit)) ? stage[i][j] : 0.f;
        uint4 pieces[NS];
        split_pieces<NS>(v, pieces, xscale);
        if (own[i]) {
#pragma unroll
            for (int s = 0; s < NS; ++s) *reinterpret_cast<uint4*>(buf + loff[i] + s * 32) = pieces[s];
        }
    };
    auto load_x = [&](int K) {
#pragma unroll
        for (int i = 0; i < NU; ++i) load_unit(K, i);
    };
    auto store_x = [&](int K) {
#pragma unroll
        for (int i = 0; i < NU; ++i) store_unit(K, i);
    };
    auto a_ptr = [&](int K) {
        int ch = K * ks + kj;
        ch = ch < nchunks ? ch : 0;  // waves past the last chunk re-read chunk 0: harmless, their MFMAs are skipped
        return wbase + (size_t)ch * (NT * NS * 64);
    };
    auto load_a = [&](uint4 (&dst)[NS], const uint4* wc, int t) {
#pragma unroll
        for (int s = 0; s < NS; ++s) dst[s] = wc[(t * NS + s) * 64];
    };

    const int lane_base = kj * (P * SUBB) + r * ROWB + c * PS + h2 * 16;

    // A fragments: a ring of AD + 1 register sets indexed by the global step g = K*NT + t; set g % NA holds step g, the
    // loads of step g + AD are issued when step g starts (AD steps of MFMAs cover the L2 latency of the weight stream)
    constexpr int AD = (AD_ < NT ? AD_ : NT), NA = AD + 1;
    uint4 abuf[NA][NS];
#pragma unroll
    for (int g = 0; g < AD; ++g) load_a(abuf[g], a_ptr(0), g);
    load_x(0);
    NND_PSTAMP(6);
    store_x(0);
    NND_PSTAMP(7);
    __syncthreads();
    NND_SSTAMP(1);
    NND_SCLOCK(5);

    // ---- PIPELINED walk (round 4).  What the assembly of round 3's FAST loop showed at every super-chunk boundary: the global loads
    // of the next patch were issued at the top of the chunk and the register allocator, short of registers, copied two of their
    // destination registers right behind them (s_waitcnt vmcnt(2) directly after the issue: a full L2 round trip with every wave
    // of the workgroup parked); the whole split of the next patch (about 90 VALU instructions per wave) ran behind the last MFMA
    // of the chunk with all 12 waves in lock step, the matrix pipes idle; then the barrier, then the LDS latency of the first
    // B fragments.  Now a super-chunk has no seam: the B fragment of the next unit — also across the chunk boundary — is always
    // read one unit ahead; the next patch's loads are issued behind the MFMAs of unit UL; its split runs as micro-steps behind
    // the MFMA groups of units U0 .. NUNIT-2 (the VALU port is free for ~3/4 of an MFMA's 32 cycles); and the barrier sits
    // between the last two units: every LDS read of the current buffer and every write of the other one has been issued (and,
    // by the barrier's lgkmcnt(0), completed) by then, while each wave still has a unit of MFMAs to issue behind it.
    // Same values in the same order into the same accumulators as the walk below: bit-identical results.
    uint4 bq2[2][NS];
    uint32_t pw[NU][NS][4];
    auto read_b2 = [&](const unsigned char* xb, int u, uint4 (&dst)[NS]) {
        const int t = u / P, pp = u % P;
        const int dy = t / KW, dx = t % KW;
        const int toff = STR == 1 ? dy * ROWB + dx * PS : ((dy & 1) * 2 + (dx & 1)) * PHB + (dy >> 1) * ROWB + (dx >> 1) * PS;
#pragma unroll
        for (int s = 0; s < NS; ++s) dst[s] = *reinterpret_cast<const uint4*>(xb + pp * SUBB + toff + s * 32);
    };
    auto micro = [&](int q, unsigned char* wbuf) {  // slot q: compile-time after unrolling
        using PP = SplitPipe<NT, P, NU>;
        if (q >= PP::NSTEPS) return;
        if (q < NU * 4) {
            const int i = q / 4, pr = q % 4;
            if constexpr (NS == 2) {
                // 4 instructions per channel pair: hi = rn_f16(s * x) for both channels into the two halves of one register, then
                // lo = rn_f16(s * x - hi) reading hi's halves as the fp16 addend (v_fma_mix*: fp32 FMA with fp16 operands / result;
                // s * x and s * x - hi are exact in fp32, so each piece is rounded once — the values of split_pieces_n).  The
                // compiler's own selection for the C++ form below took 9 (it formed every hi piece twice).  Positions outside the
                // image are zeroed through the scale (the clamped load returned a finite value), no per-element select.
                const float sc = inimg[i] ? xscale : 0.f;
                uint32_t hi, lo;
                asm("v_fma_mixlo_f16 %0, %2, %3, 0\n\t"
                    "v_fma_mixhi_f16 %0, %2, %4, 0\n\t"
                    "v_fma_mixlo_f16 %1, %2, %3, -%0 op_sel_hi:[0,0,1]\n\t"
                    "v_fma_mixhi_f16 %1, %2, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                    : "=&v"(hi), "=&v"(lo)
                    : "v"(sc), "v"(stage[i][2 * pr]), "v"(stage[i][2 * pr + 1]));
                pw[i][0][pr] = hi;
                pw[i][1][pr] = lo;
            } else {
                float r0 = inimg[i] ? stage[i][2 * pr] : 0.f, r1 = inimg[i] ? stage[i][2 * pr + 1] : 0.f;
#pragma unroll
                for (int sp = 0; sp < NS; ++sp) {
                    typedef __bf16 v2 __attribute__((ext_vector_type(2)));
                    v2 v;
                    v[0] = (__bf16)r0; v[1] = (__bf16)r1;
                    r0 -= (float)v[0]; r1 -= (float)v[1];
                    pw[i][sp][pr] = __builtin_bit_cast(uint32_t, v);
                }
            }
        }
        const int w = q - 4;  // piece w % 4 of unit w / 4: its conversions ended with slot 4 * (w / 4) + 3 (every thread owns its units here, see `own`)
        if (w >= 0 && w / 4 < NU && w % 4 < NS) {
#ifdef NND_SPLIT_NO_LDSW
            if (pw[w / 4][0][0] == 0x12345678u)
#endif
            *reinterpret_cast<uint4*>(wbuf + loff[w / 4] + (w % 4) * 32) =
                make_uint4(pw[w / 4][w % 4][0], pw[w / 4][w % 4][1], pw[w / 4][w % 4][2], pw[w / 4][w % 4][3]);
        }
    };
    auto chunk_pipe = [&](int K, auto par_c, auto more_c) {
        using PP = SplitPipe<NT, P, NU>;
        constexpr int par = decltype(par_c)::value;
        constexpr bool more = decltype(more_c)::value;
        constexpr int NUNIT = PP::NUNIT;
        const int BUFB = ks * P * SUBB;
        const uint4* wc = a_ptr(K);
        const uint4* wn = a_ptr(more ? K + 1 : K);  // past the end: re-reads the last chunk, never used
        const unsigned char* xb = lds_raw + (K & 1) * BUFB + lane_base;
        const unsigned char* xbn = lds_raw + ((K + 1) & 1) * BUFB + lane_base;
        unsigned char* wbuf = lds_raw + ((K + 1) & 1) * BUFB;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            uint4(&ac)[NS] = abuf[(par + t) % NA];
            uint4(&an)[NS] = abuf[(par + t + AD) % NA];
#ifndef NND_SPLIT_NO_ALOAD  // (timing-only ablation builds: scripts/build_ablate.sh)
            if (t + AD < NT) load_a(an, wc, t + AD);
            else load_a(an, wn, t + AD - NT);
#else
            for (int s = 0; s < NS; ++s) an[s] = ac[s];
#endif
#pragma unroll
            for (int pp = 0; pp < P; ++pp) {
                const int u = t * P + pp;
#ifndef NND_SPLIT_NO_BREAD
                if (u + 1 < NUNIT) read_b2(xb, u + 1, bq2[(u + 1) & 1]);
                else if (more) read_b2(xbn, 0, bq2[0]);  // unit 0 of the next super-chunk (the barrier below has passed)
#endif
                __builtin_amdgcn_sched_barrier(0);
#ifndef NND_SPLIT_NO_MFMA
                split_mfma_step<NS>(ac, bq2[u & 1], acc[pp]);
#else
                if (acc[0][0] == 123.f) split_mfma_step<NS>(ac, bq2[u & 1], acc[pp]);
#endif
#ifndef NND_SPLIT_NO_STAGE
                if (more) {
#ifndef NND_SPLIT_NO_XLOAD
                    if (u == PP::UL) load_x(K + 1);
#endif
#ifndef NND_SPLIT_NO_MICRO
                    if (u >= PP::U0 && u <= NUNIT - 2) {
#pragma unroll
                        for (int q = 0; q < PP::SPU; ++q) micro((u - PP::U0) * PP::SPU + q, wbuf);
                    }
#endif
                }
#endif
                __builtin_amdgcn_sched_barrier(0);
                if (u == NUNIT - 2) __syncthreads();
            }
        }
    };

    // `mine_c`: this wave's K slice exists in super-chunk K (always, in the FAST regime)
    auto chunk_body = [&](int K, auto par_c, auto mine_c) {
        constexpr int par = decltype(par_c)::value;
        constexpr bool mine = decltype(mine_c)::value;
        const bool more = (K + 1 < nsuper);
#ifndef NND_SPLIT_NO_STAGE
        if (more) load_x(K + 1);
#endif
        const uint4* wc = a_ptr(K);
        const uint4* wn = a_ptr(more ? K + 1 : K);  // past the end: re-reads the last chunk, never used
        const unsigned char* xb = lds_raw + (K & 1) * (ks * P * SUBB) + lane_base;
        // B fragments rotate through NSLOT slots at (tap, sub-tile) granularity: unit u = t*P + pp lives in slot u % NSLOT and
        // the reads of unit u + NSLOT - 1 are issued when unit u starts (its slot was freed by unit u - 1)
        constexpr int NUNIT = NT * P, NSLOT = NND_SPLIT_BSLOTS;
        uint4 bq[NSLOT][NS];
        auto read_b = [&](int u, uint4 (&dst)[NS]) {
            const int t = u / P, pp = u % P;
            const int dy = t / KW, dx = t % KW;
            const int toff = STR == 1 ? dy * ROWB + dx * PS : ((dy & 1) * 2 + (dx & 1)) * PHB + (dy >> 1) * ROWB + (dx >> 1) * PS;
#pragma unroll
            for (int s = 0; s < NS; ++s)
                dst[s] = *reinterpret_cast<const uint4*>(xb + pp * SUBB + toff + s * 32);
        };
#pragma unroll
        for (int u = 0; u < NSLOT - 1 && u < NUNIT; ++u) read_b(u, bq[u]);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            uint4(&ac)[NS] = abuf[(par + t) % NA];
            uint4(&an)[NS] = abuf[(par + t + AD) % NA];
#ifndef NND_SPLIT_NO_ALOAD
            if (t + AD < NT) load_a(an, wc, t + AD);
            else load_a(an, wn, t + AD - NT);
#else
            for (int s = 0; s < NS; ++s) an[s] = ac[s];
#endif
#pragma unroll
            for (int pp = 0; pp < P; ++pp) {
                const int u = t * P + pp;
                if (u + NSLOT - 1 < NUNIT) read_b(u + NSLOT - 1, bq[(u + NSLOT - 1) % NSLOT]);
#ifndef NND_SPLIT_NO_PIN
                // keep the prefetches (weight fragments of step t + AD, activation fragments of the next unit) ABOVE the MFMAs
                // they were written before: unpinned, the scheduler sinks every load down to its first use to save registers
                // (global_load ... s_waitcnt vmcnt(0) ... v_mfma) and the ring hides nothing
                if constexpr (FAST) __builtin_amdgcn_sched_barrier(0);
#endif
#ifndef NND_SPLIT_NO_MFMA
                if constexpr (mine)
#else
                if (mine && acc[0][0] == 123.f)
#endif
                    split_mfma_step<NS>(ac, bq[u % NSLOT], acc[pp]);  // small products first (split_arith.h)
#ifndef NND_SPLIT_NO_PIN
                if constexpr (FAST) __builtin_amdgcn_sched_barrier(0);
#endif
            }
        }
#ifndef NND_SPLIT_NO_STAGE
        if (more) store_x(K + 1);
#endif
        __syncthreads();
    };
    auto chunk = [&](int K, auto par_c) {
        if constexpr (FAST) {
            chunk_body(K, par_c, std::true_type{});
        } else {
            if (K * ks + kj < nchunks) chunk_body(K, par_c, std::true_type{});
            else chunk_body(K, par_c, std::false_type{});
        }
    };
    // the ring phase of a chunk's first step, (K*NT) % NA, must be a compile-time constant: walk the chunks in periods
    constexpr int STEP = NT % NA, PERIOD = NA / split_gcd(STEP, NA);
    if constexpr (PIPE) read_b2(lds_raw + lane_base, 0, bq2[0]);  // unit 0 of super-chunk 0 (behind the prologue's barrier)
    if constexpr (PIPE) {  // all super-chunks but the last stage their successor; the last one is its own code copy (no runtime test inside a chunk)
        const int last = nsuper - 1;
        for (int K0 = 0; K0 < last; K0 += PERIOD)
            split_static_for<PERIOD>([&](auto j) {
                constexpr int jj = decltype(j)::value;
                if (jj == 0 || K0 + jj < last) chunk_pipe(K0 + jj, std::integral_constant<int, (jj * STEP) % NA>{}, std::true_type{});
            });
        split_static_for<PERIOD>([&](auto j) {
            constexpr int jj = decltype(j)::value;
            if (PERIOD == 1 || last % PERIOD == jj) chunk_pipe(last, std::integral_constant<int, (jj * STEP) % NA>{}, std::false_type{});
        });
    } else {
        for (int K0 = 0; K0 < nsuper; K0 += PERIOD)
            split_static_for<PERIOD>([&](auto j) {
                constexpr int jj = decltype(j)::value;
                if (jj == 0 || K0 + jj < nsuper) chunk(K0 + jj, std::integral_constant<int, (jj * STEP) % NA>{});
            });
    }

    NND_SSTAMP(2);
    NND_SCLOCK(6);
    if constexpr (NS == 2) {  // exact: a power of two
#pragma unroll
        for (int pp = 0; pp < P; ++pp)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[pp][i] *= oscale;
    }
    int ys[P], xs[P];
#pragma unroll
    for (int pp = 0; pp < P; ++pp) {
        ys[pp] = ty0[pp] + r;
        xs[pp] = tx0[pp] + c;
    }
    float* red_all = reinterpret_cast<float*>(lds_raw);
    constexpr int TS = P * 1024;
    // intra-workgroup split-K reduction through LDS (the patch buffers are free after the last barrier): every wave leaves
    // its partial tile, slice kj then owns registers [kj*16/ks, (kj+1)*16/ks) of the summed tile for the epilogue
    auto exchange = [&]() {
        if (active) {
            float* red = red_all + (size_t)(cbi * ks + kj) * TS + lane;
#pragma unroll
            for (int pp = 0; pp < P; ++pp)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) red[pp * 1024 + reg * 64] = acc[pp][reg];
        }
        __syncthreads();
    };
    // c4 destination: NG = 4/ks register groups per sub-tile.  What the epilogue reads (per-pixel bias map, h, z) is requested
    // BEFORE the exchange — the fragment registers are dead — and arrives while the partial sums cross LDS.
    auto finish_c4 = [&](auto ng_c) {
        constexpr int NG = decltype(ng_c)::value;  // == 4 / ks, ks = 2 or 4
        const int q0 = kj * NG;
        EpiOpsC4<P, NG> eo;
#ifndef NND_SPLIT_NO_EPI
        if (active) epi_c4_load<P, NG>(a, cb, b, h2, q0, ys, xs, eo);
#endif
        exchange();
        if (!active) return;
        float4 cacc[P][NG];
#pragma unroll
        for (int pp = 0; pp < P; ++pp)
#pragma unroll
            for (int j = 0; j < NG; ++j) {
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float* red = red_all + (size_t)(cbi * ks) * TS + pp * 1024 + (4 * (q0 + j) + i) * 64 + lane;
                    float sum = red[0];
#pragma unroll
                    for (int sl = 1; sl < 4 / NG; ++sl) sum += red[(size_t)sl * TS];
                    v[i] = sum;
                }
                cacc[pp][j] = make_float4(v[0], v[1], v[2], v[3]);
            }
        NND_SSTAMP(3);
#ifdef NND_SPLIT_NO_EPI
        if (cacc[0][0].x != 123.456f) return;
#endif
        epi_c4_store<P, NG>(a, cacc, cb, b, h2, q0, eo);
    };
    if (a.ld.ci == 4) {
        if (ks == 4) finish_c4(std::integral_constant<int, 1>{});
        else if (ks == 2) finish_c4(std::integral_constant<int, 2>{});
        else {  // no exchange to hide the loads behind
            if (!active) return;
            NND_SSTAMP(3);
#ifdef NND_SPLIT_NO_EPI
            if (acc[0][0] != 123.456f) return;
#endif
            conv_epilogue_c4<P>(a, acc, cb, b, h2, 0, 16, ys, xs);
        }
    } else {
        if (ks > 1) exchange();
        if (!active) return;
        const int nreg = 16 / ks, reg0 = kj * nreg;
        if (ks > 1) {
            const float* red = red_all + (size_t)(cbi * ks) * TS + lane;
#pragma unroll
            for (int pp = 0; pp < P; ++pp)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    if (reg < reg0 || reg >= reg0 + nreg) continue;
                    float sum = red[pp * 1024 + reg * 64];
                    for (int j = 1; j < ks; ++j) sum += red[(size_t)j * TS + pp * 1024 + reg * 64];
                    acc[pp][reg] = sum;
                }
        }
        NND_SSTAMP(3);
#ifdef NND_SPLIT_NO_EPI
        if (acc[0][0] != 123.456f) return;
#endif
        conv_epilogue_planar<P>(a, acc, cb, b, h2, reg0, nreg, ys, xs);
    }
#ifdef NND_DBG_STAMPS
    __builtin_amdgcn_s_waitcnt(0);
#endif
    NND_SSTAMP(4);
}

// --------------------------------------------------------------------------- host side shared by the instantiation units
struct SplitCfg {
    int ny, wco, ks, P, ntiles, tiles_x, nu;
    bool fast;
    size_t lds;
    int stride;
};

constexpr int SPLIT_MAX_WAVES = 12;

template <int KH, int KW, int NS, int P, int NU, bool FAST, bool SRC4, int AD, int MAXT, int STR = 1>
int launch_split_kernel(const ConvArgs& a, dim3 grid, dim3 block, size_t lds, hipStream_t stream) {
    auto kern = conv_split_kernel<KH, KW, NS, P, NU, FAST, SRC4, AD, MAXT, STR>;
    if ((int)block.x > MAXT) {
        set_error("conv_split: %u threads exceed the %d-thread bound of this instantiation", block.x, MAXT);
        return NND_ERR_INVALID;
    }
    lds += switches().lds_slack;
    if (lds > 64 * 1024) {
        static std::atomic<unsigned> raised{0};
        if (int rc = raise_lds_limit(reinterpret_cast<const void*>(kern), raised)) return rc;
    }
    hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
    return NND_OK;
}

// weight-ring depth per kernel shape in the FAST regime (steps of lookahead; NA = AD + 1 register sets of NS x 16 B):
// 3x3: NA = 3 divides the 9 taps (one code copy of the chunk body); 1x5 / 5x1: NA = 5 divides the 5 taps for fp16x2 (40 VGPRs),
// 3 sets for bf16x3 (the body is then instantiated for 3 ring phases); 1x1: 2 sets
template <int KH, int KW, int NS>
constexpr int split_fast_ad() {
#ifdef NND_SPLIT_FAST_AD
    return NND_SPLIT_FAST_AD;
#else
    return KH * KW == 9 ? 2 : (KH * KW == 5 ? (NS == 2 ? 4 : 2) : 1);
#endif
}

// every instantiation of one (KH, KW, NS): generic and FAST (c4 / planar tile-major sources), P = 2, NU 2 | 4, 768 threads.
// (P = 3 / 4 with 8 waves were measured at 68x120 and lose to P = 2 with 12 waves on every layer — profiles/r03_split_shape_sweep_68x120.txt
//  — so they are not instantiated; the kernel template itself stays general in P.)
template <int KH, int KW, int NS>
int launch_split_shape(const ConvArgs& a, const SplitCfg& cfg, dim3 grid, dim3 block, hipStream_t stream) {
    constexpr int FAD = split_fast_ad<KH, KW, NS>();
    if (cfg.P != 2) return NND_ERR_UNSUPPORTED;
    if (cfg.stride == 2) {  // FAST regime only; planar or 4-channel-interleaved (round 4: the encoder's tensors) sources
        if constexpr (KH * KW == 9 || KH * KW == 1) {
            if (!cfg.fast) return NND_ERR_UNSUPPORTED;
            if (a.ls.ci == 4) {
                if (cfg.nu <= 2) return launch_split_kernel<KH, KW, NS, 2, 2, true, true, FAD, 768, 2>(a, grid, block, cfg.lds, stream);
                return launch_split_kernel<KH, KW, NS, 2, 4, true, true, FAD, 768, 2>(a, grid, block, cfg.lds, stream);
            }
            if (cfg.nu <= 2) return launch_split_kernel<KH, KW, NS, 2, 2, true, false, FAD, 768, 2>(a, grid, block, cfg.lds, stream);
            return launch_split_kernel<KH, KW, NS, 2, 4, true, false, FAD, 768, 2>(a, grid, block, cfg.lds, stream);
        } else {
            return NND_ERR_UNSUPPORTED;
        }
    }
    if (!cfg.fast) {
        if (cfg.nu <= 2) return launch_split_kernel<KH, KW, NS, 2, 2, false, false, 1, 768>(a, grid, block, cfg.lds, stream);
        return launch_split_kernel<KH, KW, NS, 2, 4, false, false, 1, 768>(a, grid, block, cfg.lds, stream);
    }
    if (a.ls.ci == 4) {
        if (cfg.nu <= 2) return launch_split_kernel<KH, KW, NS, 2, 2, true, true, FAD, 768>(a, grid, block, cfg.lds, stream);
        return launch_split_kernel<KH, KW, NS, 2, 4, true, true, FAD, 768>(a, grid, block, cfg.lds, stream);
    }
    if (cfg.nu <= 2) return launch_split_kernel<KH, KW, NS, 2, 2, true, false, FAD, 768>(a, grid, block, cfg.lds, stream);
    return launch_split_kernel<KH, KW, NS, 2, 4, true, false, FAD, 768>(a, grid, block, cfg.lds, stream);
}

template <int NS>
int launch_split_ns(const ConvArgs& a, const SplitCfg& cfg, int KH, int KW, dim3 grid, dim3 block, hipStream_t stream);

#define NND_SPLIT_DEFINE_NS(NS)                                                                                                      \
    template <>                                                                                                                      \
    int launch_split_ns<NS>(const ConvArgs& a, const SplitCfg& cfg, int KH, int KW, dim3 grid, dim3 block, hipStream_t stream) {     \
        if (KH == 3 && KW == 3) return launch_split_shape<3, 3, NS>(a, cfg, grid, block, stream);                                    \
        if (KH == 1 && KW == 5) return launch_split_shape<1, 5, NS>(a, cfg, grid, block, stream);                                    \
        if (KH == 5 && KW == 1) return launch_split_shape<5, 1, NS>(a, cfg, grid, block, stream);                                    \
        if (KH == 1 && KW == 1) return launch_split_shape<1, 1, NS>(a, cfg, grid, block, stream);                                    \
        return NND_ERR_UNSUPPORTED;                                                                                                  \
    }

}  // namespace nnd
