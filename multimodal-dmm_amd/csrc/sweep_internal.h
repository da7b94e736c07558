// Internal launch interface between the C-ABI entry points (sweep_api.hip) and the kernel
// families.  A launcher returns MDMM_UNSUPPORTED when the shape is outside its family; the
// entry point then tries the next one (generic SIMT kernels accept every shape).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/mdmm_hip.h"

#define MDMM_UNSUPPORTED (-100)

int mdmm_simt_sweep_fwd(const mdmm_sweep_t* a, hipStream_t stream);
int mdmm_simt_sweep_bwd(const mdmm_sweep_t* a, hipStream_t stream);
int mdmm_mfma_sweep_fwd(const mdmm_sweep_t* a, hipStream_t stream);
int mdmm_mfma_sweep_bwd(const mdmm_sweep_t* a, hipStream_t stream);
int mdmm_mfma_coop_bwd(const mdmm_sweep_t* a, hipStream_t stream);   /* sweep_coop.hip */
int mdmm_wide_sweep_fwd(const mdmm_sweep_t* a, hipStream_t stream);    /* sweep_wide.hip */
int mdmm_wide_sweep_bwd(const mdmm_sweep_t* a, hipStream_t stream);
int mdmm_wide_sweep_fwd_long(const mdmm_sweep_t* a, hipStream_t stream);   /* sweep_wide_long.hip */
int mdmm_wide_bwd_supported(const mdmm_sweep_t* a);
int mdmm_wide_sweep_bwd4(const mdmm_sweep_t* a, hipStream_t stream);      /* sweep_wide_bwd4.hip */
int mdmm_wide_bwd4_supported(const mdmm_sweep_t* a);
int64_t mdmm_wide_bwd4_ws_bytes(const mdmm_sweep_t* a);
int64_t mdmm_wide_fwd_park_bytes(const mdmm_sweep_t* a);       /* 0: the shape is not the one-round backward's */
int mdmm_wide_bwd4_shape(const mdmm_sweep_t* a);                /* the shape alone (with or without a park) */
int mdmm_wide_trans(const mdmm_sweep_t* a, int bwd, hipStream_t stream);   /* trans_wide.hip */
int mdmm_gru_wide(const mdmm_gru_t* a, int bwd, hipStream_t stream);     /* dks_wide.hip */
int mdmm_dks_wide(const mdmm_dks_t* a, int bwd, hipStream_t stream);
int mdmm_sweep_check_args(const mdmm_sweep_t* a, int bwd);
int mdmm_mfma_bwd_supported(const mdmm_sweep_t* a);
int mdmm_mfma_dw_width(int D, int H);
int64_t mdmm_mfma_dw_rows(const mdmm_sweep_t* a);

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device attribute: `done` is one slot array per
// kernel (a function-local static of the caller), indexed by the current device.
struct MdmmLdsGuard { size_t bytes[64] = {}; };
inline int mdmm_lds_attr(MdmmLdsGuard& g, const void* kern, size_t bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (g.bytes[dev] >= bytes) return 0;
  hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return (int)e;
  g.bytes[dev] = bytes;
  return 0;
}

// The same keyed by the kernel's address, for launchers that are templates over the kernel (a
// function-local static there would be shared by every kernel of the same signature).
#include <map>
#include <mutex>
#include <utility>
inline int mdmm_lds_attr_fn(const void* kern, size_t bytes) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, size_t> seen;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  std::lock_guard<std::mutex> lock(mu);
  size_t& have = seen[std::make_pair(kern, dev)];
  if (have >= bytes) return 0;
  hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return (int)e;
  have = bytes;
  return 0;
}
