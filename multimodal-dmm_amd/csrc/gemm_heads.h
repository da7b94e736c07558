// Shape-specialised GEMM kernels of the plug-ins' Linear heads (gemm_heads.hip), taken by mdmm_gemm_bf16
// (gemm_tiles.hip) when a call has their shape; internal to the library.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/mdmm_hip.h"

namespace heads {
bool expand_ok(const mdmm_gemm_t* g);
int expand_launch(const mdmm_gemm_t* g, hipStream_t st);
bool contract_ok(const mdmm_gemm_t* g);
int contract_split(const mdmm_gemm_t* g);
int contract_launch(const mdmm_gemm_t* g, hipStream_t st);
bool wgrad_ok(const mdmm_gemm_t* g);
int wgrad_split(const mdmm_gemm_t* g);
int wgrad_launch(const mdmm_gemm_t* g, hipStream_t st);
}  // namespace heads
