// tohc_sort.h -- canonical ordering of the result slab on the device (sort.hip; kept in its own translation unit
// because the rocPRIM radix sort is slow to compile).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/trajopt_hipcc.h"

namespace tohc {
// limits of the packed 64-bit sort key: [set 21][cast 1][scene-wide child number of A 21][of B 21]
enum { SORT_MAX_SETS = 1 << 21, SORT_MAX_CHILDREN = 1 << 21 };

// bytes of temporary storage sortRecords needs for n records
hipError_t sortTempBytes(size_t n, size_t *bytes);
// keys / idx only, no gather: afterwards keys + n holds the sorted keys and idx + n the slab position of every canonical
// position (what the hinge assembly reads through, tohc_hinge.h)
hipError_t sortKeys(const tohc_record *records, size_t n, int T, const int *obj_off, uint64_t *keys, uint32_t *idx, void *tmp, size_t tmpBytes, hipStream_t s);
// order: (problem, t, cctype != 0, idA, childA, idB, childB) ascending -- the order tohc_fetch documents.
// obj_off[object] = scene-wide number of its first child (device pointer; must ascend with the object id).
// keys/idx: 2*n each (double buffers).  Writes records[idx[i]] -> out_records[i] and, when grads != null, the 2K
// gradient doubles of each record -> out_grads.
hipError_t sortRecords(const tohc_record *records, const double *grads, size_t n, int T, int K, const int *obj_off, uint64_t *keys, uint32_t *idx, void *tmp,
                       size_t tmpBytes, tohc_record *out_records, double *out_grads, hipStream_t s);
}  // namespace tohc
