// api_util.hpp -- host-side error plumbing for the C-ABI (status code + thread-local message).
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

int seeme_fail(const char* msg);                 // records msg, returns 1
int seeme_fail_hip(const char* what, hipError_t e);  // records "what: hipErrorString", returns 2
int seeme_check_launch(const char* kernel);      // hipGetLastError after a launch

#define SEEME_HIP(expr)                                          \
    do {                                                         \
        hipError_t _e = (expr);                                  \
        if (_e != hipSuccess) return seeme_fail_hip(#expr, _e);  \
    } while (0)

// shared between translation units
#include "../../include/seeme_hip.h"
struct LinearKArgs {
    SeemeLinearArgs a;
    // optional row remapping (sequence-structured tensors)
    int seq_in;      // logical rows per sequence (0 = identity mapping everywhere)
    int in_stride;   // physical rows per sequence of A   (A row = (m/seq_in)*in_stride + m%seq_in + in_off)
    int in_off;
    int out_stride;  // physical rows per sequence of Y
    int out_off;
    int res_mode;      // 0: residual row = output row; 1: (m % seq_in) + res_off (positional embedding add);
                       // 2: m / seq_in (one row per sequence: broadcast vector)
    int res_off;
    // optional batch of independent problems over blockIdx.z (element strides per z; nz = 0 or 1: single problem)
    int nz;
    long zs_a, zs_w, zs_b, zs_y, zs_ln;
    int pre_ln_zmin; // batched problems: the pre-LayerNorm applies to members z >= pre_ln_zmin only (0: to all)
    int narrow;      // (set by the launcher) 64 output columns per workgroup, 16 per wave: few-row problems spread over 4x the CUs
};
int seeme_launch_linear(const LinearKArgs& ka, hipStream_t st);
int seeme_linear_simple(hipStream_t st, const float* A, int lda, const float* W, int ldw, const float* bias,
                        float* Y, int ldy, int M, int N, int K, int act, int pre_act,
                        const float* pre_ln_w, const float* pre_ln_b);

int seeme_vae_encode_h16(const SeemeVaeWeights* w, const float* features, const int32_t* lengths, int B, int T, float* mu,
                         void* workspace, hipStream_t st);
int seeme_vae_decode_h16(const SeemeVaeWeights* w, const float* z, const int32_t* lengths, int B, int T, float* feats,
                         void* workspace, hipStream_t st);
