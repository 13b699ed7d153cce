// libmod16hip.so -- host side of the C ABI declared in include/mod16_hip.h, as ONE translation
// unit: the entry-point families of capi/ included one after the other (single-command builds:
// kernel variants, tests/host_asan, hipcc -S listings). mod16_amd/csrc/build.py compiles the
// families side by side and links them into the same library.
#include "capi/context.hip"
#include "capi/forward.hip"
#include "capi/methods.hip"
#include "capi/raw.hip"
#include "capi/calibration.hip"
#include "capi/diagnostics.hip"
#include "capi/tiled.hip"
