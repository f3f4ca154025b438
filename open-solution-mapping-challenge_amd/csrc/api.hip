// error reporting + version of the C ABI (include/msc.h)
#include <stdarg.h>

#include "common.h"
#include "msc_internal.h"

thread_local char msc_err_buf[512] = "";

int msc_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(msc_err_buf, sizeof(msc_err_buf), fmt, ap);
    va_end(ap);
    return code;
}

int msc_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return msc_fail(MSC_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    return MSC_OK;
}

extern "C" const char* msc_last_error(void) { return msc_err_buf; }
extern "C" int msc_abi_version(void) { return 11; }
