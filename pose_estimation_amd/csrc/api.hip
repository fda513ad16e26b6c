// api.hip — version / status strings of the C ABI (include/posehot.h).
#include "common.hpp"

extern "C" int ph_version(void) { return PH_VERSION; }

extern "C" const char* ph_status_string(int status) {
    switch (status) {
        case PH_OK: return "ok";
        case PH_ERR_INVALID_ARG: return "invalid argument";
        case PH_ERR_UNSUPPORTED: return "unsupported size or dtype";
        case PH_ERR_WORKSPACE: return "workspace missing, misaligned or too small";
        case PH_ERR_LAUNCH: return "HIP launch failed";
        case PH_ERR_TOO_LARGE: return "size exceeds the static scratch of the no-allocation entry point";
        default: return "unknown status";
    }
}
