// api.hip -- ABI bookkeeping of libspt_hip.so (version, error strings).
#include "spt_common.h"

extern "C" int spt_abi_version(void) { return SPT_ABI_VERSION; }

extern "C" const char *spt_strerror(int code) {
    switch (code) {
        case SPT_OK: return "ok";
        case SPT_EINVAL: return "invalid argument (null pointer or non-positive size)";
        case SPT_ESHAPE: return "shape precondition violated";
        case SPT_EUNSUP: return "unsupported size combination";
        default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "unknown error";
}
