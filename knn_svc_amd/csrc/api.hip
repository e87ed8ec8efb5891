#include "common.h"
thread_local char g_knnsvc_err[512] = "";
extern "C" int knnsvc_abi_version(void) { return KNNSVC_ABI_VERSION; }
extern "C" const char* knnsvc_last_error(void) { return g_knnsvc_err; }
