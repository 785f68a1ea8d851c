/* defines.h -- shim under the reference's header name (device/lib/defines.h): the handful of
 * macros callers of the lower surface use.  The reference's compile-time configuration matrix is
 * not reproduced (default configuration only, user_defines.h:37-117). */
#ifndef SEAMD_SHIM_DEFINES_H
#define SEAMD_SHIM_DEFINES_H
#include <assert.h>
#include "../seal_embedded_amd.h"
#define SE_USE_MALLOC
#define SE_UNUSED(x) (void)(x)
#define se_assert(x) assert(x)
#endif
