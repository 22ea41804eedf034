/* dmz_hip_test.h -- entry points of libdmz_hip.so that exist for the test suite and the developer tools only.  Not part of
 * the drop-in boundary (include/dmz_hip.h): nothing in the reference corresponds to them and no caller should bind them. */
#ifndef DMZ_HIP_TEST_H
#define DMZ_HIP_TEST_H

#include "dmz_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Overwrite the WHOLE LDS of every CU with `word` (asynchronous, on the context's stream): workgroups that together take
 * hipDeviceAttributeMaxSharedMemoryPerMultiprocessor bytes per CU, enough of them to cover every CU several times over.  A
 * kernel that reads LDS words it never wrote sees them afterwards -- tests/test_gpu_hseg.py runs the scan behind 0xFFFFFFFF (a
 * NaN pattern), tools/dev/determinism.py alternates two patterns. */
int dmz_hip_debug_fill_lds(dmz_hip_context *ctx, uint32_t word);

#ifdef __cplusplus
}
#endif
#endif
