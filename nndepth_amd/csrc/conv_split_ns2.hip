// conv_split_kernel instantiations for arithmetic 2 ("fp16x2": 2 range-scaled fp16 pieces, 3 products on v_mfma_f32_32x32x16_f16).
#include "conv_split_kernel.h"

namespace nnd {
NND_SPLIT_DEFINE_NS(2)
#ifdef NND_DBG_STAMPS
extern "C" int nnd_debug_read_split_stamps_ns2(unsigned long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_split_stamps), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -1;
}
#endif
}  // namespace nnd
