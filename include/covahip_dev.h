/*
 * covahip_dev.h -- developer switches of libcovahip.so.  NOT part of the drop-in boundary
 * (include/covahip.h): used by tools/ and tests/ only, may change or disappear between builds.
 */
#ifndef COVAHIP_DEV_H
#define COVAHIP_DEV_H

#include "covahip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Encoder implementation of the loaded model: 1 = one kernel per encoder level, 2 = levels 0 and 1 in
 * one kernel.  Both compute identical bits (tests/test_gpu_blobnet.py); the library picks its default. */
int covahip_blobnet_set_impl(covahip_ctx *ctx, int impl);

#ifdef __cplusplus
}
#endif
#endif /* COVAHIP_DEV_H */
