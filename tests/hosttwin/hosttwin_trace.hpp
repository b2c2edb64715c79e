// TEST-ONLY: records which precomputed-table entries the device templates read (ECGPU_TABLE_TOUCH hook, mp32.hpp).
// Included before the product headers by every hosttwin translation unit.
#pragma once
#include <stddef.h>
extern "C" void ht_trace_push(int idx);
#define ECGPU_TABLE_TOUCH(idx) ht_trace_push((int)(idx))
