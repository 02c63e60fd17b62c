// What gemm.hip and gemm_panel.hip share: the epilogue term masks of the specialised instantiations and the launcher of
// the panel-resident kernel.
#pragma once
#include "common.hpp"

// bit mask of the epilogue terms an instantiation carries (alpha = 1); EPI < 0 in gemm.hip: decided at run time
constexpr int PE_BIAS = 1, PE_RELU = 2, PE_DROP = 4, PE_RES = 8, PE_GATE = 16, PE_LNF = 32, PE_STATS = 64, PE_DOT = 128;
constexpr int LNF_GROUPS = 8;  // LayerNorm fold: row length 512 = 8 groups of 64 columns

// gemm_panel.hip: C[M,N] (bf16) = epilogue(A[M,K] B[N,K]^T) with a 96-column panel of B resident in LDS.
// Returns JS2T_OK / an error code when it took the product, -1 when the product is not one of its shapes (the caller goes on).
int launch_bf16_pan96(const js2t_gemm_desc& d, int epi_mask, hipStream_t s);
