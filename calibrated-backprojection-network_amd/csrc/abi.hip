// abi.hip -- version / status entry points of the C ABI (include/kbnet_hip.h).
#include "kbn_common.h"

extern "C" {

int kbn_version(void) { return KBN_ABI_VERSION; }

const char* kbn_status_string(int status) {
    switch (status) {
        case KBN_OK: return "ok";
        case KBN_ERR_INVALID_ARGUMENT: return "invalid argument";
        case KBN_ERR_UNSUPPORTED: return "configuration outside the kernel limits";
        case KBN_ERR_WORKSPACE: return "buffer too small";
        case KBN_ERR_LAUNCH: return "HIP launch error";
        default: return "unknown status";
    }
}

}  // extern "C"
