// Version and error reporting of the C ABI (include/iq.h).
#include "iq_common.h"

extern "C" int iq_version(void) { return 100; }  // 0.1.0

extern "C" const char* iq_last_error(void) { return iq::err_buf(); }
