// Library identity + error string of libcmk_hip.so.
#include "cmk_common.hpp"

namespace cmk {
thread_local char g_err[512] = "";
}

extern "C" int cmk_version(void) { return 5; }
extern "C" const char* cmk_arch(void) { return "gfx950"; }
extern "C" const char* cmk_last_error(void) { return cmk::g_err; }
