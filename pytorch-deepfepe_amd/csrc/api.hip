// Library-level entry points of libdfepe_hip.so (see include/dfepe.h).
#include "dfepe_common.h"

extern "C" int dfepe_version(void) { return DFEPE_VERSION; }

extern "C" int dfepe_save_floats(void) { return DFEPE_SAVE_FLOATS; }

extern "C" const char* dfepe_strerror(int code) {
  switch (code) {
    case DFEPE_OK: return "ok";
    case DFEPE_ERR_INVALID_ARG: return "invalid argument (null pointer, bad size, misaligned buffer or bad flags)";
    case DFEPE_ERR_HIP: return "HIP runtime error while configuring or launching a kernel";
    case DFEPE_ERR_UNSUPPORTED: return "request not supported by this build";
    default: return "unknown dfepe error code";
  }
}
