// Library-level entry points of liblp_hip.so.
#include "lp_common.h"

extern "C" int lp_version(void) { return 120; }  // 0.2.0: + two BatchNorm segments per launch (lp_bn_fuse.seg_images, lp_bn_finalize2), fp32 validation path (lp_f32_*)

extern "C" const char* lp_strerror(int code) {
    if (code == LP_OK) return "ok";
    if (code == LP_ERR_ARGUMENT) return "invalid argument (null pointer or non-positive dimension)";
    if (code == LP_ERR_UNSUPPORTED) return "shape not covered by the instantiated kernels";
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "unknown lp_hip error";
}
