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
int seeme_linear_simple(hipStream_t st, const float* A, int lda, const float* W, int ldw, const float* bias,
                        float* Y, int ldy, int M, int N, int K, int act, int pre_act,
                        const float* pre_ln_w, const float* pre_ln_b);
