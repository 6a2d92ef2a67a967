// Library-level entry points of liblp_hip.so.
#include "lp_common.h"

#include <stdlib.h>

#include <atomic>

extern "C" int lp_version(void) { return LP_HIP_ABI_VERSION; }   // (include/lp_hip.h: checked by every binding before its first call)

namespace lp {
static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return (e == nullptr || *e == 0) ? dflt : atoi(e);
}
static LpSwitches read_switches() {
    LpSwitches s;
    s.conv_pipe = env_int("LP_CONV_PIPE", 1);       // 0: every convolution on conv_igemm_kernel / conv_wgrad_kernel
    s.conv_halo = env_int("LP_CONV_HALO", 1);       // 0: the 3x3 layers on the per-tap ring
    s.conv_res2d = env_int("LP_CONV_RES2D", 1);     // 0: layer1's 64 -> 64 3x3 layers on the HALO form
    s.infer_pipe = env_int("LP_INFER_PIPE", 1);     // 0: lp_conv_fwd_act on conv_igemm_kernel<infer>
    s.gemm_pipe = env_int("LP_GEMM_PIPE", 1);       // 0: the Linear layers on conv_igemm_kernel
    s.wgrad_pipe = env_int("LP_WGRAD_PIPE", 1);     // 0: weight gradients on conv_wgrad_kernel; 2: the pipelined kernel wherever it can run
    s.stem_2d = env_int("LP_STEM_2D", 1);           // 0: the stem on conv_igemm_kernel<64, stem>
    s.stem_wgrad_nb = env_int("LP_STEM_WGRAD_NB", 1);   // 0: the stem's weight gradient on conv_wgrad_kernel<64, stem>
    s.pool_v2 = env_int("LP_POOL_V2", 1);           // 0: the stem's pool backward on the first kernel
    s.bn_bwd_wgs_per_cu = env_int("LP_BN_BWD_WGS_PER_CU", 5);   // 1 .. 5: workgroups per CU of the BatchNorm backward walk (5 = one resident round)
    if (s.bn_bwd_wgs_per_cu < 1 || s.bn_bwd_wgs_per_cu > 5) s.bn_bwd_wgs_per_cu = 5;
    s.conv_max_wgs = env_int("LP_CONV_MAX_WGS", 0); // > 0: cap of the persistent grids (tests: several tiles per workgroup on small problems)
    return s;
}
// The table is published through an atomic pointer: a call that reads it sees ONE consistent table, and lp_config_reload_env() swaps in a new
// one (the old tables stay alive - a few dozen bytes per reload - so a reader that already holds one never sees it change or freed).  An entry
// point may read the pointer more than once (workspace query + launch), so a reload must still not run concurrently with other calls (lp_hip.h).
static std::atomic<const LpSwitches*> g_switches{new LpSwitches(read_switches())};     // (dynamic initialisation at library load)
const LpSwitches& lp_switches() { return *g_switches.load(std::memory_order_acquire); }
}  // namespace lp

extern "C" int lp_config_reload_env(void) {
    lp::g_switches.store(new lp::LpSwitches(lp::read_switches()), std::memory_order_release);
    return LP_OK;
}

extern "C" const char* lp_strerror(int code) {
    if (code == LP_OK) return "ok";
    if (code == LP_ERR_ARGUMENT) return "invalid argument (null pointer or non-positive dimension)";
    if (code == LP_ERR_UNSUPPORTED) return "shape not covered by the instantiated kernels";
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "unknown lp_hip error";
}
