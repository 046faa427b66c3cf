// Library self-description: ABI version and the digest of the sources this binary was compiled from (set by the in-tree
// builder, tcar_amd._lib.build, as -DTCAR_BUILD_ID="<hex>"); this is the only translation unit that depends on it.
#include "../../include/tcar_hip.h"

#ifndef TCAR_BUILD_ID
#define TCAR_BUILD_ID "unknown"
#endif

extern "C" int tcar_abi_version(void) { return TCAR_ABI_VERSION; }

extern "C" const char* tcar_build_id(void) {
  static const char id[] = "TCAR_BUILD_ID=" TCAR_BUILD_ID;   // the marker lets a loader read the id from the file itself
  return id + 14;
}
