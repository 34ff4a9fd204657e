// Version / status entry points of libglare_hip.so.
#include "common.h"

extern "C" int glare_version(void) { return 120; }  // 1.20: + IEEE-half twin library, hi / lo residual stream, grouped conv launches, single-pass DCN,
                                                    //       ActNorm initialisation, guarded Adam / GradScaler (1.10: training step, DCN v1, device-side harness)

extern "C" const char* glare_status_string(int status) {
  switch (status) {
    case GLARE_OK: return "ok";
    case GLARE_ERR_INVALID: return "invalid argument";
    case GLARE_ERR_LAUNCH: return "HIP launch failure";
    case GLARE_ERR_WORKSPACE: return "workspace too small";
    case GLARE_ERR_UNSUPPORTED: return "unsupported configuration";
    default: return "unknown status";
  }
}
