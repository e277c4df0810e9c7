// status strings / version of libvilco_hip.so
#include "common.h"

extern "C" const char* vilco_status_str(int status) {
  switch (status) {
    case VILCO_OK: return "ok";
    case VILCO_ERR_BADARG: return "vilco_hip: bad argument (null pointer, negative size or misaligned operand)";
    case VILCO_ERR_UNSUPPORTED: return "vilco_hip: shape not supported by the gfx950 kernels";
    case VILCO_ERR_LAUNCH: return "vilco_hip: kernel launch failed";
    case VILCO_ERR_WORKSPACE: return "vilco_hip: workspace too small";
    default: return "vilco_hip: unknown status";
  }
}

extern "C" const char* vilco_version(void) { return "vilco_hip 0.1 gfx950"; }
