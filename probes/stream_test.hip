// stream_test.hip -- micro-benchmark / correctness probe for the per-wave LDS-DMA weight ring used by
// the sampling kernel (debug entry point, not part of the product path).
//
// Every workgroup streams the SAME `bytes` (like the shared weight image): wave w consumes 1 KiB
// pieces j*NW + w.  mode 0: plain register loads (8 deep); mode 1: private LDS ring filled by
// global_load_lds_dwordx4 (inline asm, hand-counted vmcnt), R pieces ahead.
#include "common.hpp"
#include "api_util.hpp"
#include "lds_ring.hpp"

template <int R>
__global__ __launch_bounds__(512) void k_stream_ring(const float4* __restrict__ src, long npw, int reps,
                                                     float* __restrict__ out) {
    const long npieces_per_wave = npw * reps;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int NW = blockDim.x >> 6;
    float* ring = smem + wave * (R * 256);                       // R slots x 1 KiB per wave
    const uint32_t ring_lds = lds_addr_of(ring);
    const float4* my = src + (size_t)wave * 64 + lane;            // piece j -> my + j*NW*64
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // prologue: R pieces in flight
    for (int j = 0; j < R; ++j)
        if (j < npieces_per_wave) lds_dma_1k(my + (size_t)(j % npw) * NW * 64, ring_lds + (uint32_t)j * 1024u);
    for (long j = 0; j < npieces_per_wave; ++j) {
        const int slot = (int)(j % R);
        // pieces issued after piece j: min(R-1, remaining) -> wait until at most that many are outstanding
        const long younger = (npieces_per_wave - 1 - j) < (R - 1) ? (npieces_per_wave - 1 - j) : (R - 1);
        wait_vmcnt_dyn<R>((int)younger);
        const float4 w = *reinterpret_cast<const float4*>(ring + slot * 256 + lane * 4);
        acc.x += w.x; acc.y += w.y; acc.z += w.z; acc.w += w.w;
        if (j + R < npieces_per_wave)
            lds_dma_1k_after_read(my + (size_t)((j + R) % npw) * NW * 64, ring_lds + (uint32_t)slot * 1024u);
    }
    wait_vmcnt0();
    out[(size_t)blockIdx.x * blockDim.x + tid] = (acc.x + acc.y) + (acc.z + acc.w);
}

__global__ __launch_bounds__(512) void k_stream_regs(const float4* __restrict__ src, long npieces_per_wave, int reps,
                                                     float* __restrict__ out) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int NW = blockDim.x >> 6;
    const float4* my = src + (size_t)wave * 64 + lane;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int rep = 0; rep < reps; ++rep) {
    long j = 0;
#pragma unroll 1
    for (; j + 16 <= npieces_per_wave; j += 16) {
        float4 w[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) w[i] = my[(size_t)(j + i) * NW * 64];
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc.x += w[i].x; acc.y += w[i].y; acc.z += w[i].z; acc.w += w[i].w; }
    }
    for (; j < npieces_per_wave; ++j) { const float4 w = my[(size_t)j * NW * 64]; acc.x += w.x; acc.y += w.y; acc.z += w.z; acc.w += w.w; }
  }
    out[(size_t)blockIdx.x * blockDim.x + tid] = (acc.x + acc.y) + (acc.z + acc.w);
}

// mode 2: the GEMV pattern of the sampling kernel -- double-buffered chunks of CH x 16 B per lane through raw
// buffer loads, a workgroup barrier + a short wave-local epilogue every `per_gemv` chunks.
template <int CH>
__global__ __launch_bounds__(512) void k_stream_gemv(const float4* __restrict__ src, int bytes, long npw, int reps,
                                                     int per_gemv, int epi_iters, float* __restrict__ out) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    __shared__ float red[512];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(src), 0, bytes, 0x00020000);
    const unsigned voff = (unsigned)(wave * 64 + lane) * 16u;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const long nchunks = npw * reps / CH;
    u32x4 A[CH], B[CH];
    auto soff = [&](long c, int i) { return (unsigned)((((c * CH + i) % npw) * 8) * 1024); };
#pragma unroll
    for (int i = 0; i < CH; ++i) A[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff(0, i), 0);
#pragma unroll
    for (int i = 0; i < CH; ++i) B[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff(1, i), 0);
    for (long c = 0; c + 1 < nchunks; c += 2) {
#pragma unroll
        for (int i = 0; i < CH; ++i) { acc.x += __uint_as_float(A[i].x); acc.y += __uint_as_float(A[i].y); acc.z += __uint_as_float(A[i].z); acc.w += __uint_as_float(A[i].w); }
#pragma unroll
        for (int i = 0; i < CH; ++i) A[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff(c + 2, i), 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < CH; ++i) { acc.x += __uint_as_float(B[i].x); acc.y += __uint_as_float(B[i].y); acc.z += __uint_as_float(B[i].z); acc.w += __uint_as_float(B[i].w); }
#pragma unroll
        for (int i = 0; i < CH; ++i) B[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff(c + 3, i), 0);
        __builtin_amdgcn_sched_barrier(0);
        if (per_gemv > 0 && ((c + 2) % per_gemv) == 0) {
            red[tid] = acc.x;
            __syncthreads();
            float v = red[(tid + 64) & 511];
            for (int e = 0; e < epi_iters; ++e) v = wave_sum(v) * 0.015625f + 1e-9f;
            acc.w += v * 1e-30f;
        }
    }
    out[(size_t)blockIdx.x * blockDim.x + tid] = (acc.x + acc.y) + (acc.z + acc.w);
    asm volatile("" ::"v"(A[0].x), "v"(B[0].x));
}

extern "C" int seeme_debug_stream_gemv(const float* src, long bytes, int reps, int ch, int per_gemv, int epi_iters,
                                       int blocks, float* out, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const long npw = bytes / 1024 / 8;
    if (ch == 8) hipLaunchKernelGGL((k_stream_gemv<8>), dim3(blocks), dim3(512), 0, st, (const float4*)src, (int)bytes, npw, reps, per_gemv, epi_iters, out);
    else if (ch == 4) hipLaunchKernelGGL((k_stream_gemv<4>), dim3(blocks), dim3(512), 0, st, (const float4*)src, (int)bytes, npw, reps, per_gemv, epi_iters, out);
    else if (ch == 16) hipLaunchKernelGGL((k_stream_gemv<16>), dim3(blocks), dim3(512), 0, st, (const float4*)src, (int)bytes, npw, reps, per_gemv, epi_iters, out);
    else return seeme_fail("debug_stream_gemv: ch must be 4, 8 or 16");
    return seeme_check_launch("k_stream_gemv");
}

extern "C" int seeme_debug_stream(const float* src, long bytes, int reps, int mode, int ring, int blocks, float* out, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int NW = 8;
    const long npw = bytes / 1024 / NW;
    if (npw <= 0) return seeme_fail("debug_stream: too few bytes");
    if (mode == 0) {
        hipLaunchKernelGGL(k_stream_regs, dim3(blocks), dim3(512), 0, st, (const float4*)src, npw, reps, out);
    } else if (ring == 8) {
        const size_t lds = (size_t)NW * 8 * 1024;
        hipLaunchKernelGGL((k_stream_ring<8>), dim3(blocks), dim3(512), lds, st, (const float4*)src, npw, reps, out);
    } else if (ring == 16) {
        const size_t lds = (size_t)NW * 16 * 1024;
        SEEME_HIP(hipFuncSetAttribute((const void*)k_stream_ring<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_stream_ring<16>), dim3(blocks), dim3(512), lds, st, (const float4*)src, npw, reps, out);
    } else {
        return seeme_fail("debug_stream: ring must be 8 or 16");
    }
    return seeme_check_launch("k_stream");
}

#define PIN() do { asm volatile("" : "+v"(acc.x), "+v"(acc.y), "+v"(acc.z), "+v"(acc.w) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
// mode 3: register ring of R chunks (8 x 16 B per lane each) -- the in-flight depth experiment behind the
// sampling kernel's weight pipeline.  Consumes chunk t, re-fills its slot with chunk t + R; a workgroup
// barrier + wave-local epilogue every `per_gemv` chunks (per_gemv % R == 0 or 0).
template <int R>
__global__ __launch_bounds__(512) void k_stream_rr(const float4* __restrict__ src, int bytes, long nchunks,
                                                   int per_gemv, int epi_iters, float* __restrict__ out) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    constexpr int CH = 8;
    __shared__ float red[512];
    const int tid = threadIdx.x;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(src), 0, bytes, 0x00020000);
    const unsigned voff = (unsigned)tid * 16u;
    const unsigned chunk_bytes = CH * 512 * 16;
    const unsigned wrap = (unsigned)bytes / chunk_bytes * chunk_bytes;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    u32x4 ring[R][CH];
    unsigned soff = 0;
#pragma unroll
    for (int s = 0; s < R; ++s) {
#pragma unroll
        for (int i = 0; i < CH; ++i) ring[s][i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff + i * 8192u, 0);
        soff += chunk_bytes; if (soff >= wrap) soff = 0;
    }
    int since = 0;
#pragma unroll 1
    for (long c = 0; c < nchunks; c += R) {
#pragma unroll
        for (int s = 0; s < R; ++s) {
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                acc.x += __uint_as_float(ring[s][i].x); acc.y += __uint_as_float(ring[s][i].y);
                acc.z += __uint_as_float(ring[s][i].z); acc.w += __uint_as_float(ring[s][i].w);
            }
            PIN();   // the re-fill reuses the registers just consumed
#pragma unroll
            for (int i = 0; i < CH; ++i) ring[s][i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff + i * 8192u, 0);
            soff += chunk_bytes; if (soff >= wrap) soff = 0;
            PIN();
            if (per_gemv > 0 && ++since == per_gemv) {
                since = 0;
                red[tid] = acc.x;
                __syncthreads();
                float v = red[(tid + 64) & 511];
                for (int e = 0; e < epi_iters; ++e) v = wave_sum(v) * 0.015625f + 1e-9f;
                acc.w += v * 1e-30f;
            }
        }
    }
    out[(size_t)blockIdx.x * blockDim.x + tid] = (acc.x + acc.y) + (acc.z + acc.w);
#pragma unroll
    for (int s = 0; s < R; ++s) asm volatile("" ::"v"(ring[s][0].x));
}

extern "C" int seeme_debug_stream_rr(const float* src, long bytes, long nchunks, int ring, int per_gemv, int epi_iters,
                                     int blocks, float* out, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (ring == 2) hipLaunchKernelGGL((k_stream_rr<2>), dim3(blocks), dim3(512), 0, st, (const float4*)src, (int)bytes, nchunks, per_gemv, epi_iters, out);
    else if (ring == 3) hipLaunchKernelGGL((k_stream_rr<3>), dim3(blocks), dim3(512), 0, st, (const float4*)src, (int)bytes, nchunks, per_gemv, epi_iters, out);
    else if (ring == 4) hipLaunchKernelGGL((k_stream_rr<4>), dim3(blocks), dim3(512), 0, st, (const float4*)src, (int)bytes, nchunks, per_gemv, epi_iters, out);
    else if (ring == 6) hipLaunchKernelGGL((k_stream_rr<6>), dim3(blocks), dim3(512), 0, st, (const float4*)src, (int)bytes, nchunks, per_gemv, epi_iters, out);
    else return seeme_fail("debug_stream_rr: ring must be 2, 3, 4 or 6");
    return seeme_check_launch("k_stream_rr");
}

// mode 4: role-specialised variant of mode 3 (ring of 4).  Per "GEMV" of C chunks: every wave consumes C chunks,
// re-filling only from the third consume on; barrier A; wave 0 runs the epilogue while waves 1..7 issue the two
// withheld chunks (so the L1 fill path stays busy during the epilogue); barrier B; wave 0 issues its own two.
template <int C, bool VOL>
__global__ __launch_bounds__(512) void k_stream_spec(const float4* __restrict__ src, int bytes, long ngemv,
                                                     int epi_iters, float* __restrict__ out) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    constexpr int CH = 8, R = 4;
    constexpr int AUX = VOL ? (int)0x80000000 : 0;
    static_assert(C % R == 0 || C == 2, "slot phase must repeat per GEMV");
    __shared__ float red[512];
    const int tid = threadIdx.x, wave = tid >> 6;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(src), 0, bytes, 0x00020000);
    const unsigned voff = (unsigned)tid * 16u;
    const unsigned chunk_bytes = CH * 512 * 16;
    const unsigned wrap = (unsigned)bytes / chunk_bytes * chunk_bytes;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    u32x4 ring[R][CH];
    unsigned soff = 0;
#pragma unroll
    for (int s = 0; s < R; ++s) {
#pragma unroll
        for (int i = 0; i < CH; ++i) ring[s][i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff + i * 8192u, AUX);
        soff += chunk_bytes; if (soff >= wrap) soff = 0;
    }
    constexpr int UN = (C == 2) ? 2 : 1;   // GEMVs per loop trip so that slot indices are compile-time
#pragma unroll 1
    for (long g = 0; g < ngemv; g += UN) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            unsigned soff_b[2];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int s = (u * C + c) % R;
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    acc.x += __uint_as_float(ring[s][i].x); acc.y += __uint_as_float(ring[s][i].y);
                    acc.z += __uint_as_float(ring[s][i].z); acc.w += __uint_as_float(ring[s][i].w);
                }
                PIN();
                if (c >= 2) {   // re-fill the slot consumed two chunks ago
                    const int s2 = (u * C + c - 2) % R;
#pragma unroll
                    for (int i = 0; i < CH; ++i) ring[s2][i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff + i * 8192u, AUX);
                    soff += chunk_bytes; if (soff >= wrap) soff = 0;
                }
                PIN();
            }
            soff_b[0] = soff; soff += chunk_bytes; if (soff >= wrap) soff = 0;
            soff_b[1] = soff; soff += chunk_bytes; if (soff >= wrap) soff = 0;
            red[tid] = acc.x;
            __syncthreads();                                         // barrier A: partial sums published
            if (wave == 0) __builtin_amdgcn_s_setprio(3);           // the epilogue wave's requests go first
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int s2 = (u * C + C - 2 + b) % R;
#pragma unroll
                for (int i = 0; i < CH; ++i) ring[s2][i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff_b[b] + i * 8192u, AUX);
            }
            PIN();
            if (wave == 0) {
                float v = red[(tid + 64) & 511];
                for (int e = 0; e < epi_iters; ++e) v = wave_sum(v) * 0.015625f + 1e-9f;
                red[tid] = v;
            }
            __syncthreads();                                         // barrier B: next input vector published
            acc.w += red[tid & 63] * 1e-30f;
            PIN();
        }
    }
    out[(size_t)blockIdx.x * blockDim.x + tid] = (acc.x + acc.y) + (acc.z + acc.w);
#pragma unroll
    for (int s = 0; s < R; ++s) asm volatile("" ::"v"(ring[s][0].x));
}

extern "C" int seeme_debug_stream_spec(const float* src, long bytes, long ngemv, int c, int vol, int epi_iters,
                                       int blocks, float* out, void* stream) {
    hipStream_t st = (hipStream_t)stream;
#define SPEC_LAUNCH(CC, VV) hipLaunchKernelGGL((k_stream_spec<CC, VV>), dim3(blocks), dim3(512), 0, st, (const float4*)src, (int)bytes, ngemv, epi_iters, out)
    if (c == 2 && vol) SPEC_LAUNCH(2, true); else if (c == 2) SPEC_LAUNCH(2, false);
    else if (c == 4 && vol) SPEC_LAUNCH(4, true); else if (c == 4) SPEC_LAUNCH(4, false);
    else if (c == 8 && vol) SPEC_LAUNCH(8, true); else if (c == 8) SPEC_LAUNCH(8, false);
    else return seeme_fail("debug_stream_spec: c must be 2, 4 or 8");
#undef SPEC_LAUNCH
    return seeme_check_launch("k_stream_spec");
}
