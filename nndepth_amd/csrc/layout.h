// Activation layouts of the hot path.
//   NCHW        : (c, y, x) -> c*H*W + y*W + x                         (the reference's layout; every C-ABI tensor)
//   tile-major  : (c, y, x) -> c*PLANE + ((y/4)*TX + x/8)*32 + (y%4)*8 + x%8,  TX = ceil(W/8), PLANE = ceil(H/4)*TX*32
// Internal workspace buffers of the update block are tile-major: the 32 pixels of an MFMA column block (a 4x8
// sub-tile) are then ONE 128-B line per channel, so a half-wave's store / gate-operand load in the conv epilogue
// is a full line instead of four 32-B row segments (measured: the NCHW epilogue cost 6-10 us of a 45-65 us launch).
#pragma once
#include <hip/hip_runtime.h>

namespace nnd {

struct Lay {
    int W, TX, tiled;
    long plane;  // floats per channel per batch item
};

__host__ __device__ inline long tiled_plane(int H, int W) { return (long)((H + 3) / 4) * ((W + 7) / 8) * 32; }

__host__ inline Lay make_lay(int H, int W, bool tiled) {
    Lay l;
    l.W = W;
    l.TX = (W + 7) / 8;
    l.tiled = tiled ? 1 : 0;
    l.plane = tiled ? tiled_plane(H, W) : (long)H * W;
    return l;
}

__device__ __forceinline__ long pix_off(const Lay& l, int y, int x) {
    return l.tiled ? ((long)((y >> 2) * l.TX + (x >> 3)) * 32 + (y & 3) * 8 + (x & 7)) : (long)y * l.W + x;
}

}  // namespace nnd
