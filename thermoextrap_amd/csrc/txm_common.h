// txm_common.h -- shared host/device helpers for libtxmom (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>

#include "../../include/txmom.h"

#define TXM_MAXK (TXM_MAX_ORDER + 1)
#define TXM_WAVE 64

namespace txm {

// thread-local last error
void set_error(const char *fmt, ...);
int hip_fail(hipError_t e, const char *what, const char *file, int line);

#define TXM_HIP(call)                                                     \
  do {                                                                    \
    hipError_t _e = (call);                                               \
    if (_e != hipSuccess) return txm::hip_fail(_e, #call, __FILE__, __LINE__); \
  } while (0)

#define TXM_LAUNCH_CHECK()                                                \
  do {                                                                    \
    hipError_t _e = hipGetLastError();                                    \
    if (_e != hipSuccess) return txm::hip_fail(_e, "kernel launch", __FILE__, __LINE__); \
  } while (0)

#define TXM_REQUIRE(cond, ...)                                            \
  do {                                                                    \
    if (!(cond)) {                                                        \
      txm::set_error(__VA_ARGS__);                                        \
      return TXM_ERR_INVALID;                                             \
    }                                                                     \
  } while (0)

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

// Number of CUs of the current device (cached by txm_init; 256 on MI355X).
int num_cus();

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to (function, device): set it once per device the calling
// process launches on (`mask` is the call site's static bit set of devices already served).
#define TXM_SET_MAX_LDS(func, bytes)                                                                   \
  do {                                                                                                 \
    static std::atomic<uint64_t> lds_mask_{0};                                                         \
    int dev_ = 0;                                                                                      \
    TXM_HIP(hipGetDevice(&dev_));                                                                      \
    const uint64_t bit_ = (uint64_t)1 << (dev_ & 63);                                                  \
    if (!(lds_mask_.load(std::memory_order_relaxed) & bit_)) {                                         \
      TXM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(func),                                \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)));          \
      lds_mask_.fetch_or(bit_, std::memory_order_relaxed);                                             \
    }                                                                                                  \
  } while (0)

// ---- pivot-sum -> central moment shift (device + host) --------------------
// S0[j] = sum w du^j, S1[j] = sum w dx du^j about the pivot (pu, px), j < K.
// Writes one cmomy state [2][K]:  [0][0]=W, [0][1]=<u>, [1][0]=<x>,
// [a][b] = <(x-<x>)^a (u-<u>)^b>.
template <int K>
__host__ __device__ inline void pivot_sums_to_state(const double *S0, const double *S1, double pu,
                                                    double px, double *state) {
  const double W = S0[0];
  const double inv = 1.0 / W;
  const double du = (K > 1) ? S0[1] * inv : 0.0;  // <u> - pu
  const double dx = S1[0] * inv;                  // <x> - px
  double m0[K], m1[K];
#pragma unroll
  for (int j = 0; j < K; ++j) {
    m0[j] = S0[j] * inv;
    m1[j] = S1[j] * inv;
  }
  // binomial shift of the u variable by -du:  sum_j C(b,j) (-du)^(b-j) m[j]
#pragma unroll
  for (int b = K - 1; b >= 0; --b) {
    double a0 = 0.0, a1 = 0.0, p = 1.0, c = 1.0;  // p = (-du)^(b-j), c = C(b, j)
    // iterate j from b down to 0
#pragma unroll
    for (int j = b; j >= 0; --j) {
      a0 += c * p * m0[j];
      a1 += c * p * (m1[j] - dx * m0[j]);
      p *= -du;
      c = c * (double)j / (double)(b - j + 1);  // C(b, j-1) = C(b,j) * j / (b-j+1)
    }
    state[b] = a0;
    state[K + b] = a1;
  }
  state[0] = W;
  if (K > 1) state[1] = pu + du;
  state[K] = px + dx;
}

}  // namespace txm
