// conv_gemm_host.h -- host-side pieces the GEMM translation units share (conv_gemm.hip, conv_gemm_split.hip); not part of common.h's engine-facing API.
#pragma once
#include <atomic>
#include "common.h"

extern std::atomic<long> g_conv_chain_launches[2];   // GEMM launches with a fused chain: [0] compiled epilogue, [1] interpreted (engines may run on several host threads)
int conv_gemm_num_cus();
// Chain of a launch: operand prefetch plan, compiled signature.  Returns 0, or why the launch cannot carry its chain (conv_gemm_refusal).
int conv_gemm_plan_chain(ConvParams& q);
void conv_gemm_warn_interpreted(const ConvParams& q);
// K17 (conv_gemm_split.hip).  false: the launch is not one the bf16x6 kernel covers (or its planes could not be built) -- the caller takes the fp32 kernel
bool conv_gemm_launch_split(const ConvParams& p, hipStream_t s);
bool conv_gemm_split_wanted(const ConvParams& p);       // a covered layer AND a grid worth it
