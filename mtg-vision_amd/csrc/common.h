// mtgv - MI355X-native recognition hot path. Shared host/device helpers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <stdexcept>

namespace mtgv {

// ---- error plumbing: C-ABI returns int status, text via mtgv_last_error() ----
void set_last_error(const std::string& s);

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& s) : std::runtime_error(s), code(c) {}
};

enum Status : int {
  OK = 0,
  ERR_INVALID = 1,   // bad argument / shape (reference: AssertionError)
  ERR_KEY = 2,       // unknown name (reference: KeyError)
  ERR_RUNTIME = 3,   // HIP failure or bad state (reference: RuntimeError)
};

#define MTGV_CHECK(cond, code, ...)                                        \
  do {                                                                     \
    if (!(cond)) {                                                         \
      char _b[512];                                                        \
      snprintf(_b, sizeof(_b), __VA_ARGS__);                               \
      throw ::mtgv::Error(code, std::string(_b) + " [" #cond "] at " __FILE__ ":" + std::to_string(__LINE__)); \
    }                                                                      \
  } while (0)

#define HIP_OK(expr)                                                       \
  do {                                                                     \
    hipError_t _e = (expr);                                                \
    if (_e != hipSuccess)                                                  \
      throw ::mtgv::Error(::mtgv::ERR_RUNTIME, std::string(#expr) + ": " + hipGetErrorString(_e) + " at " __FILE__ ":" + std::to_string(__LINE__)); \
  } while (0)

template <class F>
static inline int guarded(F&& f) {
  try {
    f();
    return OK;
  } catch (const Error& e) {
    set_last_error(e.what());
    return e.code;
  } catch (const std::exception& e) {
    set_last_error(e.what());
    return ERR_RUNTIME;
  }
}

// ---- exact unsigned division by a runtime constant (n * d < 2^40) ----
struct FastDiv {
  uint64_t mul;  // ceil(2^40 / d)
  uint32_t d;
};
static inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.d = d ? d : 1;
  f.mul = ((1ull << 40) + f.d - 1) / f.d;
  return f;
}
__host__ __device__ static inline uint32_t fdiv(uint32_t n, const FastDiv& f) {
  return (uint32_t)(((uint64_t)n * f.mul) >> 40);
}

// Per-device lazily created state (zero pages, function attributes, scratch words) is kept in small tables indexed by
// the HIP device ordinal: a handle is bound to the device current at its creation and several may coexist in a process.
constexpr int MTGV_MAX_DEVICES = 64;
static inline int current_device() {
  int d = 0;
  HIP_OK(hipGetDevice(&d));
  MTGV_CHECK(d >= 0 && d < MTGV_MAX_DEVICES, ERR_RUNTIME, "device ordinal %d outside [0, %d)", d, MTGV_MAX_DEVICES);
  return d;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

enum Act : int { ACT_NONE = 0, ACT_GELU = 1, ACT_MISH = 2, ACT_SILU = 3, ACT_SIGMOID = 4 };

}  // namespace mtgv
