// handoff_test.hip -- what does it cost to hand a small vector from one CU to another through L2?  (Design question behind the
// sampling kernel: a sample's dependent GEMV chain could be split over k CUs to multiply the per-CU weight-stream rate, at the
// price of one exchange per GEMV.)  Two workgroups ping-pong: A writes `floats` values + a flag, B polls the flag, reads the
// values, writes its own and its flag, A polls ... `iters` round trips; all other workgroups of the launch exit at once.
// blockIdx a / b pick the pair (consecutive block ids land on consecutive XCDs, ids 8 apart on the same XCD).
#include <hip/hip_runtime.h>
#include "api_util.hpp"

__global__ __launch_bounds__(256) void k_handoff(int* flags, float* buf, int a, int b, int iters, int floats, float* out) {
    const int me = blockIdx.x == a ? 0 : blockIdx.x == b ? 1 : -1;
    if (me < 0) return;
    volatile int* my_flag = flags + 64 * me;
    volatile int* peer_flag = flags + 64 * (1 - me);
    float* my_buf = buf + 1024 * me;
    const float* peer_buf = buf + 1024 * (1 - me);
    const int tid = threadIdx.x;
    float acc = 0.f;
    for (int it = 1; it <= iters; ++it) {
        if (me == 0 || it > 0) {
            if (me == 1 || it > 1) {                       // wait for the peer's message of this round
                const int want = me == 0 ? it - 1 : it;
                if (tid == 0) while (__hip_atomic_load((int*)peer_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {}
                __syncthreads();
                if (tid < floats) acc += __builtin_nontemporal_load(peer_buf + tid);
            }
            if (tid < floats) my_buf[tid] = acc + (float)it;
            __threadfence();
            __syncthreads();
            if (tid == 0) __hip_atomic_store((int*)my_flag, it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (me == 0) {                                          // the last reply
        if (tid == 0) while (__hip_atomic_load((int*)peer_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < iters) {}
        __syncthreads();
    }
    if (tid == 0) out[me] = acc;
}

extern "C" int seeme_debug_handoff(int* flags, float* buf, int a, int b, int iters, int floats, int blocks, float* out, void* stream) {
    if (a == b || a < 0 || b < 0 || a >= blocks || b >= blocks || floats > 256 || iters < 1) return seeme_fail("debug_handoff: bad arguments");
    hipLaunchKernelGGL(k_handoff, dim3(blocks), dim3(256), 0, (hipStream_t)stream, flags, buf, a, b, iters, floats, out);
    return seeme_check_launch("k_handoff");
}
