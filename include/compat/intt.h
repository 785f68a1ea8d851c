/* intt.h -- shim under the reference's header name (device/lib/intt.h): callers written against
 * SEAL-Embedded's own headers compile unchanged against libseal_embedded_amd.so.  Everything is
 * declared in seal_embedded_amd.h / seal_embedded_amd_lower.h. */
#ifndef SEAMD_SHIM_INTT_H
#define SEAMD_SHIM_INTT_H
#include "../seal_embedded_amd.h"
#endif
