// Library-level C ABI entry points (include/vtx.h): error strings and ABI version.
#include <atomic>
#include <stdlib.h>

#include "options.h"
#include "vtx_common.h"

namespace {
struct OptDef { const char* env; int dflt; };
const OptDef kOptDefs[VTX_OPT_COUNT] = {
    {"VTX_GEMM_GLDS", 1},   {"VTX_GLDS_BM", 0},          {"VTX_GLDS_WAVES", 8},   {"VTX_WGRAD_GLDS", 1},
    {"VTX_WG_WAVES", 8},    {"VTX_WGRAD_BLOCKS", 512},   {"VTX_SATTN", 1},        {"VTX_WATTN_FWD_WAVES", 4096},
    {"VTX_WATTN_WAVES", 2048}, {"VTX_SRATTN_WGS", 2048}, {"VTX_WATTN_XCD_MAJOR", 1},
    {"VTX_LN_FIT", 1},      {"VTX_GLDS_EPI", 1},    {"VTX_SATTN_WAVES", 8},
    {"VTX_WATTN_BWD4", 1},  {"VTX_GEMM_SKINNY", 1}, {"VTX_GEMM_ASTAT", 1},
    {"VTX_TWINS_SUB_LDS", 1}, {"VTX_WGRAD_WIDE", 1}, {"VTX_GEMM_PP", 1},
    {"VTX_LN_ROWS", 0},     {"VTX_SKINNY_WAVES", 4}, {"VTX_WATTN_FAST", 3}, {"VTX_WATTN_FWD4", 1},
    {"VTX_MLP_FUSED", 1}, {"VTX_LN_FOLD", 15},
};
struct OptTable {
  std::atomic<int> v[VTX_OPT_COUNT];
  OptTable() {
    for (int i = 0; i < VTX_OPT_COUNT; ++i) {
      const char* e = getenv(kOptDefs[i].env);
      v[i].store(e ? atoi(e) : kOptDefs[i].dflt, std::memory_order_relaxed);
    }
  }
};
OptTable& opt_table() { static OptTable t; return t; }   // initialised on first use (thread-safe static)
}  // namespace

int vtx_cu_count_cached() {
  static std::atomic<int> n{0};
  int v = n.load(std::memory_order_relaxed);
  if (v == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    n.store(v, std::memory_order_relaxed);
  }
  return v;
}

int vtx_opt(int id) { return opt_table().v[id].load(std::memory_order_relaxed); }

__global__ __launch_bounds__(256) void lds_poison_kernel(unsigned pattern, int nwords) {
  extern __shared__ unsigned lds_words[];
  for (int i = threadIdx.x; i < nwords; i += 256) lds_words[i] = pattern;
  __syncthreads();
  for (int d = 0; d < 64; ++d) __builtin_amdgcn_s_sleep(64);        // hold the CU while the other workgroups of the round start elsewhere
  if (lds_words[(threadIdx.x * 97) % nwords] != pattern) __builtin_trap();
}

// MFMA rate probe (bench.py `measured_peaks`): every wave issues `iters` x 4 independent v_mfma_f32_32x32x16_bf16 on register operands --
// no memory traffic, four accumulator tiles so that no MFMA waits for the one before it.  The caller times the launch.
typedef __attribute__((ext_vector_type(8))) __bf16 peak_bf16x8;
typedef __attribute__((ext_vector_type(16))) float peak_f32x16;
__global__ __launch_bounds__(256) void mfma_peak_kernel(int iters, float* sink) {
  peak_bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (float)((threadIdx.x + i) & 7)); b[i] = (__bf16)(0.002f * (float)((threadIdx.x * 3 + i) & 7)); }
  peak_f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  for (int it = 0; it < iters; ++it) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  if (s == 12345.678f) sink[0] = s;          // (keeps the chain alive; never true for these operands)
}

// One wavefront that holds its hardware queue for `ticks` of the 100-MHz wall clock and touches nothing (vtx_debug_spin): two of them on
// two streams take one period when the streams are served concurrently and two when they share a hardware queue.
__global__ __launch_bounds__(64) void spin_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

extern "C" {

int vtx_option_count(void) { return VTX_OPT_COUNT; }
const char* vtx_option_name(int id) { return (id >= 0 && id < VTX_OPT_COUNT) ? kOptDefs[id].env : nullptr; }
int vtx_get_option(int id) { return (id >= 0 && id < VTX_OPT_COUNT) ? vtx_opt(id) : -1; }
int vtx_set_option(int id, int value) {
  if (id < 0 || id >= VTX_OPT_COUNT) return VTX_ERR_SHAPE;
  opt_table().v[id].store(value, std::memory_order_relaxed);
  return VTX_OK;
}

const char* vtx_strerror(int code) {
  switch (code) {
    case VTX_OK: return "ok";
    case VTX_ERR_SHAPE: return "unsupported or inconsistent shape";
    case VTX_ERR_DTYPE: return "unsupported dtype (expected VTX_F32 or VTX_BF16)";
    case VTX_ERR_ALIGN: return "dimension / leading dimension not a multiple of 8 elements";
    case VTX_ERR_LAUNCH: return "HIP kernel launch failed";
    case VTX_ERR_WORKSPACE: return "workspace too small";
    case VTX_ERR_NULL: return "required pointer is NULL";
    default: return "unknown vtx error";
  }
}

int vtx_abi_version(void) { return 27; }

int vtx_cu_count(void) { return vtx_cu_count_cached(); }

/* Test helper: fills the LDS of every CU with `pattern` (e.g. 0x7fc07fc0: bf16 / fp32 NaNs).  LDS is not cleared between
 * workgroups: a kernel that reads LDS it never wrote sees what the previous workgroup on that CU left there.  Workgroups of 160 KB,
 * one per CU at a time, `rounds` per CU, each holding its CU for a few microseconds so that every CU gets one. */
int vtx_debug_lds_poison(unsigned pattern, int rounds, void* stream) {
  const int bytes = 160 * 1024;
  if (hipFuncSetAttribute((const void*)lds_poison_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return VTX_ERR_LAUNCH;
  if (rounds < 1) rounds = 1;
  hipLaunchKernelGGL(lds_poison_kernel, dim3(vtx_cu_count_cached() * rounds), dim3(256), bytes, (hipStream_t)stream, pattern, bytes / 4);
  return vtx_check_launch();
}

/* Measurement helper (bench.py `measured_peaks`): one launch of `waves_per_cu` / 4 256-thread workgroups per CU, every wave issuing
 * 4 * iters dense bf16 MFMAs (32 x 32 x 16: 32 768 FLOP each) on register operands.  *flops receives the FLOPs of the launch; the caller
 * brackets the call with events on `stream`.  `sink` = any device buffer of >= 4 bytes (never written in practice). */
int vtx_debug_mfma_peak(int iters, int waves_per_cu, void* sink, double* flops, void* stream) {
  if (!sink || iters <= 0 || waves_per_cu < 4 || waves_per_cu > 32 || waves_per_cu % 4) return VTX_ERR_SHAPE;
  const int grid = vtx_cu_count_cached() * (waves_per_cu / 4);
  hipLaunchKernelGGL(mfma_peak_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, iters, (float*)sink);
  if (flops) *flops = (double)grid * 4.0 * (double)iters * 4.0 * 32768.0;
  return vtx_check_launch();
}

/* Measurement helper: one idle wavefront that occupies `stream` for `microseconds` (wall clock) -- vtx.functional uses two of them to check
 * that the side stream of the weight gradients is really served CONCURRENTLY with the compute stream (HIP multiplexes streams onto a few
 * hardware queues: a side stream that shares the compute stream's queue serialises, profiles/round6_side_stream_queue.md). */
int vtx_debug_spin(int microseconds, void* stream) {
  if (microseconds <= 0 || microseconds > 100000) return VTX_ERR_SHAPE;
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long)microseconds * 100);
  return vtx_check_launch();
}

}  // extern "C"
