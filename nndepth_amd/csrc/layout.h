// Activation layouts of the hot path.
//   NCHW        : (c, y, x) -> c*H*W + y*W + x                         (the reference's layout; every C-ABI tensor)
//   tile-major  : (c, y, x) -> c*PLANE + ((y/4)*TX + x/8)*32 + (y%4)*8 + x%8,  TX = ceil(W/8), PLANE = ceil(H/4)*TX*32
//   tile-major, 4 channels interleaved ("c4", round 2):
//                 (c, y, x) -> (c/4)*4*PLANE + (((y/4)*TX + x/8)*32 + (y%4)*8 + x%8)*4 + c%4
//                 the 4 output channels a lane of the MFMA epilogue holds in registers 4q..4q+3 are one 16-B store, and a
//                 staging unit (pixel, 8 channels) is two 16-B loads instead of eight 4-B ones: the vector-memory pipe spends
//                 16 cycles of address processing per wave-instruction whatever its payload.  A channel
//                 slice that starts at a multiple of 4 channels starts at the same float offset as in the planar layouts.
// Internal workspace buffers of the update block are tile-major: the 32 pixels of an MFMA column block (a 4x8
// sub-tile) are then ONE 128-B line per channel, so a half-wave's store / gate-operand load in the conv epilogue
// is a full line instead of four 32-B row segments (measured: the NCHW epilogue cost 6-10 us of a 45-65 us launch).
#pragma once
#include <hip/hip_runtime.h>

namespace nnd {

struct Lay {
    int W, TX, tiled;
    long plane;  // floats per channel per batch item
    int ci;      // channels interleaved innermost: 1 (planar) or 4 (tile-major only)
};

__host__ __device__ inline long tiled_plane(int H, int W) { return (long)((H + 3) / 4) * ((W + 7) / 8) * 32; }

__host__ inline Lay make_lay(int H, int W, bool tiled, bool c4 = false) {
    Lay l;
    l.W = W;
    l.TX = (W + 7) / 8;
    l.tiled = tiled ? 1 : 0;
    l.plane = tiled ? tiled_plane(H, W) : (long)H * W;
    l.ci = (tiled && c4) ? 4 : 1;
    return l;
}

// float offset of pixel (y, x) inside a channel (group): already scaled by the channel interleave
__device__ __forceinline__ long pix_off(const Lay& l, int y, int x) {
    const long p = l.tiled ? ((long)((y >> 2) * l.TX + (x >> 3)) * 32 + (y & 3) * 8 + (x & 7)) : (long)y * l.W + x;
    return l.ci == 1 ? p : p * 4;
}
// float offset of channel c (add pix_off): c*plane for the planar layouts
__device__ __forceinline__ long chan_off(const Lay& l, int c) {
    return l.ci == 1 ? (long)c * l.plane : (long)(c >> 2) * (4 * l.plane) + (c & 3);
}

}  // namespace nnd
