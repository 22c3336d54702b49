// internal (non-exported) helpers shared between translation units
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

size_t ramp_internal_group_by_ws(int E);
int ramp_internal_group_by(const int64_t *keys, int E, int64_t key_bound, int32_t *order,
                           int32_t *gid, int32_t *seg_start, int64_t *ukeys, int32_t *ngroups,
                           void *ws, size_t ws_bytes, hipStream_t st);
