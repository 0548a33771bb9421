// sort.hip -- puts the result slab into the canonical order of tohc_fetch on the device: a 64-bit key per record,
// rocPRIM's radix sort on (key, index), then a gather of records and gradient rows.  No counterpart in the reference
// (its contact order follows Bullet's DBVT traversal and is history dependent); the order exists so that
// results are reproducible and comparable.
#include <hip/hip_runtime.h>
#include <string.h>

#include <rocprim/rocprim.hpp>

#include "tohc_sort.h"

namespace tohc {

__global__ void k_sort_keys(const tohc_record *__restrict__ r, unsigned int n, int T, const int *__restrict__ obj_off, uint64_t *__restrict__ keys,
                            uint32_t *__restrict__ idx) {
  const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const tohc_record &a = r[i];
  uint64_t k = (uint64_t)((int64_t)a.problem * T + a.t);
  k = (k << 1) | (uint64_t)(a.cctype != 0);
  // children are numbered object by object, so the scene-wide child number orders like (object, child)
  k = (k << 21) | (uint64_t)(obj_off[a.idA] + a.childA);
  k = (k << 21) | (uint64_t)(obj_off[a.idB] + a.childB);
  keys[i] = k;
  idx[i] = i;
}

// one 8-byte word per thread: records are 19 words, gradient rows 2K words
__global__ void k_sort_gather(const uint64_t *__restrict__ rec, const uint64_t *__restrict__ grads, const uint32_t *__restrict__ idx, unsigned int n,
                              int wordsRec, int wordsGrad, uint64_t *__restrict__ outRec, uint64_t *__restrict__ outGrads) {
  const size_t per = (size_t)wordsRec + wordsGrad;
  const size_t total = (size_t)n * per;
  for (size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x; w < total; w += (size_t)gridDim.x * blockDim.x) {
    const size_t i = w / per;
    const int k = (int)(w % per);
    const size_t src = idx[i];
    if (k < wordsRec) outRec[i * wordsRec + k] = rec[src * wordsRec + k];
    else outGrads[i * wordsGrad + (k - wordsRec)] = grads[src * wordsGrad + (k - wordsRec)];
  }
}

hipError_t sortTempBytes(size_t n, size_t *bytes) {
  *bytes = 0;
  return rocprim::radix_sort_pairs(nullptr, *bytes, (const uint64_t *)nullptr, (uint64_t *)nullptr, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                   n, 0, 64, (hipStream_t)0);
}

hipError_t sortKeys(const tohc_record *records, size_t n, int T, const int *obj_off, uint64_t *keys, uint32_t *idx, void *tmp, size_t tmpBytes, hipStream_t s) {
  if (n == 0) return hipSuccess;
  const unsigned int nn = (unsigned int)n;
  hipLaunchKernelGGL(k_sort_keys, dim3((nn + 255) / 256), dim3(256), 0, s, records, nn, T, obj_off, keys, idx);
  return rocprim::radix_sort_pairs(tmp, tmpBytes, (const uint64_t *)keys, keys + n, (const uint32_t *)idx, idx + n, n, 0, 64, s);
}

hipError_t sortRecords(const tohc_record *records, const double *grads, size_t n, int T, int K, const int *obj_off, uint64_t *keys, uint32_t *idx, void *tmp,
                       size_t tmpBytes, tohc_record *out_records, double *out_grads, hipStream_t s) {
  static_assert(sizeof(tohc_record) % 8 == 0, "records are copied as 8-byte words");
  if (n == 0) return hipSuccess;
  const unsigned int nn = (unsigned int)n;
  hipError_t e = sortKeys(records, n, T, obj_off, keys, idx, tmp, tmpBytes, s);
  if (e != hipSuccess) return e;
  const int wordsRec = (int)(sizeof(tohc_record) / 8), wordsGrad = grads ? 2 * K : 0;
  const size_t total = n * (size_t)(wordsRec + wordsGrad);
  const unsigned int blocks = (unsigned int)std::min<size_t>((total + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(k_sort_gather, dim3(blocks), dim3(256), 0, s, (const uint64_t *)records, (const uint64_t *)grads, idx + n, nn, wordsRec, wordsGrad,
                     (uint64_t *)out_records, (uint64_t *)out_grads);
  return hipGetLastError();
}

}  // namespace tohc
