// Thread-local error string + trivial entry points of the C-ABI.
#include "common.h"

#include <cstdlib>

namespace nnd {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static Switches g_sw;
static void load_switches() {
    Switches s{};
    auto on = [](const char* n) { return getenv(n) != nullptr; };
    s.no_fused_upsample = on("NND_NO_FUSED_UPSAMPLE");
    s.no_fused_lookup = on("NND_NO_FUSED_LOOKUP");
    s.no_fused_flow_branch = on("NND_NO_FUSED_FLOW_BRANCH");
    s.no_c4 = on("NND_NO_C4");
    s.agcl_v1 = on("NND_AGCL_V1");
    s.no_thin3d = on("NND_NO_THIN3D");
    s.corr_build_v1 = on("NND_CORR_BUILD_V1");
    s.corr_build_no_ksplit = on("NND_CORR_BUILD_NO_KSPLIT");
    s.igev_squeeze_v1 = on("NND_IGEV_SQUEEZE_V1");
    s.igev_squeeze_walk = on("NND_IGEV_SQUEEZE_WALK");
    s.no_folded_flow_head = on("NND_NO_FOLDED_FLOW_HEAD");
    s.no_conv1x1_stream = on("NND_NO_CONV1X1_STREAM");
    s.conv_verbose = on("NND_CONV_VERBOSE");
    s.debug_sync = on("NND_DEBUG_SYNC");
    s.split_ny = s.split_ks = s.split_p = -1;
    if (const char* e = getenv("NND_SPLIT_CFG")) sscanf(e, "%d,%d,%d", &s.split_ny, &s.split_ks, &s.split_p);
    s.split_no_fast = on("NND_SPLIT_NO_FAST");
    s.no_merged_fb_lookup = on("NND_NO_MERGED_FB_LOOKUP");
    s.conv_p = s.conv_ks = s.conv_wco = -1;
    if (const char* e = getenv("NND_CONV_CFG")) sscanf(e, "%d,%d,%d", &s.conv_p, &s.conv_ks, &s.conv_wco);
    if (const char* e = getenv("NND_CONV_P")) s.conv_p = atoi(e);
    s.agcl_pb = getenv("NND_AGCL_PB") ? atoi(getenv("NND_AGCL_PB")) : 0;
    s.lds_poison_on = on("NND_DEBUG_LDS_POISON");
    s.lds_poison = s.lds_poison_on ? (unsigned)strtoul(getenv("NND_DEBUG_LDS_POISON"), nullptr, 0) : 0u;
    s.lds_slack = getenv("NND_DEBUG_LDS_SLACK") ? atoi(getenv("NND_DEBUG_LDS_SLACK")) : 0;
    s.enc_no_c4 = on("NND_ENC_NO_C4");
    s.no_slab3d = on("NND_NO_SLAB3D");
    s.slab3d_rounds = getenv("NND_SLAB3D_ROUNDS") ? atoi(getenv("NND_SLAB3D_ROUNDS")) : 0;
    g_sw = s;
}
namespace {
struct SwitchInit {
    SwitchInit() { load_switches(); }
} g_switch_init;  // at library load
}  // namespace
const Switches& switches() { return g_sw; }
}  // namespace nnd

extern "C" {
int nnd_reload_switches(void) {
    nnd::load_switches();
    return NND_OK;
}
int nnd_version(void) { return NND_VERSION; }
const char* nnd_last_error(void) { return nnd::g_err; }
int nnd_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}
}
