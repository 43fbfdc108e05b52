// xchg_test.hip -- what does an all-reduce of a 256-float vector among the C workgroups (CUs) of one cluster cost INSIDE a launch?
// (Design question behind k_den_cluster: a sample's dependent GEMV chain split over C CUs pays one such exchange per
// column-split -> row-split GEMV pair.)  Protocol = cdna_hip_programming.md Guideline 16, R2: the data is the flag -- every
// value travels as one naturally aligned 8-byte {tag = epoch, value} granule written by ONE sc1 (write-through) store; the
// consumer re-reads its granules with sc1 loads until every tag equals the epoch.  No flag, no fence.
//
// One iteration = every wave of every workgroup publishes its 32 partial values (as the row-split GEMV's accumulator lanes
// would), wave 0 of every workgroup sweeps the C x 256 granules of its cluster, adds them in a fixed order, checks the sum,
// then the workgroup's barrier.  Two granule buffers alternate (a buffer is only rewritten after every reader has passed the
// following exchange).  `stream_kb` > 0: every wave also issues that many KiB / 8 of 16-byte loads per iteration from a read-only
// buffer (the weight stream that shares the CU's vector-memory queue with the polls).
// mode 0: the C workgroups of a cluster have equal blockIdx % 8 (one XCD under round-robin placement); mode 1: consecutive
// blockIdx (C different XCDs for C <= 8).  Placement is speed only: the protocol does not depend on it.
#include <hip/hip_runtime.h>
#include "api_util.hpp"

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// kind 0: sc1 (write-through: leaves the XCD's L2, any placement); kind 1: plain store (stays in the writer's XCD L2: only a
// reader on the SAME XCD sees it through an sc1 load -- legal only after the cluster has checked its XCC ids)
__device__ __forceinline__ void store_granule(unsigned long long* g, unsigned epoch, float v, int kind) {
    const unsigned long long x = ((unsigned long long)epoch << 32) | __float_as_uint(v);
    if (kind == 0) __hip_atomic_store((gu64*)g, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *(volatile gu64*)g = x;
}

template <int C>
__global__ __launch_bounds__(512) void k_xchg(unsigned long long* gran, const u32x4* wbuf, int wbuf_vecs, int mode, int iters, int stream_ld,
                                              unsigned* err, float* out, int store_kind) {
    extern __shared__ __attribute__((aligned(16))) float pad_lds[];   // 96 KiB requested: one workgroup per CU
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (iters < 0) pad_lds[tid] = 0.f;
    int cl, c;
    if (mode == 0) { const int x = blockIdx.x & 7, j = blockIdx.x >> 3; cl = x * (gridDim.x / 8 / C) + j / C; c = j % C; }
    else { cl = blockIdx.x / C; c = blockIdx.x % C; }
    // per cluster: 2 buffers x C publishers x 256 granules
    unsigned long long* base = gran + (size_t)cl * 2 * C * 256;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, 2 * C * 256 * 8, 0x00020000);
    const __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(wbuf), 0, wbuf_vecs * 16, 0x00020000);
    unsigned bad = 0, spins_max = 0;
    float acc = 0.f;
    unsigned wsum = 0;
    unsigned woff = (unsigned)((blockIdx.x * 512 + tid) * 16) % (unsigned)(wbuf_vecs * 16);
    bool dead = false;
    for (int it = 0; it < iters; ++it) {
        const unsigned epoch = (unsigned)it + 1u;
        const int buf = it & 1;
        // background "weight stream": stream_ld 16-byte loads per lane per iteration
        u32x4 wv[8];
        for (int s = 0; s < stream_ld; s += 8) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { wv[i] = __builtin_amdgcn_raw_buffer_load_b128(ws, woff, 0, 0); woff += 512 * 16 * 251; if (woff >= (unsigned)(wbuf_vecs * 16)) woff -= (unsigned)(wbuf_vecs * 16); }
#pragma unroll
            for (int i = 0; i < 8; ++i) wsum ^= wv[i].x;
        }
        // publish: this wave's 32 outputs (lanes 0..31), value depends on (it, c, index)
        if (lane < 32) {
            const int idx = wave * 32 + lane;
            const float v = (float)((it * 7 + c * 3 + idx) & 1023) * 0.25f;
            store_granule(base + ((size_t)buf * C + c) * 256 + idx, epoch, v, store_kind);
        }
        if (wave == 0) {
            // sweep: lane owns outputs 4 lane .. 4 lane + 3 of every publisher: 2 x 16-byte sc1 loads per publisher
            float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
            unsigned spins = 0;
            for (;;) {
                u32x4 g[C][2];
#pragma unroll
                for (int p = 0; p < C; ++p) {
                    const unsigned off = (unsigned)(((buf * C + p) * 256 + 4 * lane) * 8);
                    g[p][0] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);        // aux 16 = sc1
                    g[p][1] = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16, 0, 16);
                }
                bool ok = true;
                sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int p = 0; p < C; ++p) {
                    ok &= (g[p][0].y == epoch) & (g[p][0].w == epoch) & (g[p][1].y == epoch) & (g[p][1].w == epoch);
                    sum.x += __uint_as_float(g[p][0].x); sum.y += __uint_as_float(g[p][0].z);
                    sum.z += __uint_as_float(g[p][1].x); sum.w += __uint_as_float(g[p][1].z);
                }
                if (__all(ok) || dead) break;
                if (++spins > (1u << 22)) { dead = true; if (lane == 0) atomicOr(err, 1u); break; }
            }
            spins_max = spins > spins_max ? spins : spins_max;
            // expected: sum over p of ((it*7 + p*3 + idx) & 1023) / 4
            float ex[4] = {0.f, 0.f, 0.f, 0.f};
            for (int p = 0; p < C; ++p)
                for (int i = 0; i < 4; ++i) ex[i] += (float)((it * 7 + p * 3 + 4 * lane + i) & 1023) * 0.25f;
            if (!dead && (sum.x != ex[0] || sum.y != ex[1] || sum.z != ex[2] || sum.w != ex[3])) ++bad;
            acc += sum.x + sum.y + sum.z + sum.w;
        }
        __syncthreads();
    }
    if (wave == 0) {
        for (int o = 32; o; o >>= 1) bad += __shfl_xor(bad, o);
        if (lane == 0) { if (bad) atomicAdd(err + 1, bad); atomicMax(err + 2, spins_max); out[blockIdx.x] = acc + (float)(wsum & 1u); out[gridDim.x + blockIdx.x] = (float)(__builtin_amdgcn_s_getreg(0x1814) & 15); }
    }
}

// gran: clusters * 2 * C * 256 u64 (zeroed by the caller before EVERY launch); err: 4 u32 (zeroed); out: 2 * grid floats (checksum, XCC id)
extern "C" int seeme_debug_xchg(void* gran, const void* wbuf, int wbuf_vecs, int C, int clusters, int mode, int iters, int stream_ld,
                                void* err, float* out, int store_kind, void* stream) {
    const int grid = clusters * C;
    if (grid > 256 || grid % 8 != 0 || (grid / 8) % C != 0 || stream_ld % 8 != 0) return seeme_fail("debug_xchg: grid must be <= 256, a multiple of 8 C");
    hipStream_t st = (hipStream_t)stream;
#define XL(CC) SEEME_HIP(hipFuncSetAttribute((const void*)k_xchg<CC>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)); hipLaunchKernelGGL((k_xchg<CC>), dim3(grid), dim3(512), 96 * 1024, st, (unsigned long long*)gran, (const u32x4*)wbuf, wbuf_vecs, mode, iters, stream_ld, (unsigned*)err, out, store_kind)
    if (C == 2) { XL(2); } else if (C == 4) { XL(4); } else if (C == 8) { XL(8); }
    else return seeme_fail("debug_xchg: C must be 2, 4 or 8");
#undef XL
    return seeme_check_launch("k_xchg");
}
