// D2H bandwidth of hipMemcpyAsync into pinned memory for a few allocation flags / stream kinds (diagnostic tool)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t n = 64u << 20;
  void *d; CK(hipMalloc(&d, n)); CK(hipMemset(d, 1, n));
  hipStream_t s0, s1; CK(hipStreamCreate(&s0)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  unsigned flags[] = {hipHostMallocDefault, hipHostMallocNonCoherent, hipHostMallocCoherent, hipHostMallocPortable | hipHostMallocMapped};
  const char *fn[] = {"default", "noncoherent", "coherent", "portable|mapped"};
  for (int f = 0; f < 4; ++f) {
    void *h; CK(hipHostMalloc(&h, n, flags[f]));
    hipStream_t ss[] = {0, s0, s1}; const char *sn[] = {"null", "created", "nonblocking"};
    for (int k = 0; k < 3; ++k) {
      for (int w = 0; w < 2; ++w) { CK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, ss[k])); CK(hipStreamSynchronize(ss[k])); }
      double t0 = now();
      for (int r = 0; r < 5; ++r) CK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, ss[k]));
      CK(hipStreamSynchronize(ss[k]));
      double dt = (now() - t0) / 5;
      printf("%-16s stream %-12s D2H %.2f GB/s\n", fn[f], sn[k], n / dt / 1e9);
    }
    CK(hipHostFree(h));
  }
  return 0;
}
