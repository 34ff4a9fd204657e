// Version / status entry points of libglare_hip.so.
#include "common.h"

extern "C" int glare_version(void) { return 110; }  // 1.10: + training step (backward kernels, optimizer, losses), DCN v1, device-side harness

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
