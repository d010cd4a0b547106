// common.hip — ABI version and error strings of libalignq_hip.so
#include <hip/hip_runtime.h>

#include "../../include/alignq.h"

extern "C" {

int alignq_abi_version(void) { return ALIGNQ_ABI_VERSION; }

const char* alignq_strerror(int code) {
  if (code == 0) return "ok";
  if (code == ALIGNQ_EINVAL) return "alignq: invalid argument";
  if (code == ALIGNQ_EUNSUPPORTED) return "alignq: shape not supported by the gfx950 kernels";
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "alignq: unknown error";
}

}  // extern "C"
