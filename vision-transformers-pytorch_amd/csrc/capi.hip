// Library-level C ABI entry points (include/vtx.h): error strings and ABI version.
#include "vtx_common.h"

extern "C" {

const char* vtx_strerror(int code) {
  switch (code) {
    case VTX_OK: return "ok";
    case VTX_ERR_SHAPE: return "unsupported or inconsistent shape";
    case VTX_ERR_DTYPE: return "unsupported dtype (expected VTX_F32 or VTX_BF16)";
    case VTX_ERR_ALIGN: return "dimension / leading dimension not a multiple of 8 elements";
    case VTX_ERR_LAUNCH: return "HIP kernel launch failed";
    case VTX_ERR_WORKSPACE: return "workspace too small";
    case VTX_ERR_NULL: return "required pointer is NULL";
    default: return "unknown vtx error";
  }
}

int vtx_abi_version(void) { return 5; }

}  // extern "C"
