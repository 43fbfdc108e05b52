// den_kernels.hip -- the conditional latent denoiser and its sampling loop as ONE persistent kernel.
//
// Reference: MldDenoiser.forward (mld/models/architectures/mld_denoiser.py:151-244) with
// LinearTemporalDiffusionTransformerDecoderLayer (mdiff_transformer.py:286-304) inside
// SkipTransformerEncoder (cross_attention.py:67-83), driven by MLD._diffusion_reverse
// (mld/models/modeltype/mld.py:467-497).
//
// Design (DESIGN.md section "denoiser"): the denoiser sees 2+N tokens and keeps only token 0, so one
// step is a chain of ~47 dependent 256-wide GEMVs per sample.  One workgroup (512 threads) owns one
// sample (or one CFG pair) for the WHOLE loop: no launches, no inter-workgroup traffic; the packed
// weight image is streamed from L2 / Infinity Cache with 1 KiB-per-wave coalesced reads, all
// workgroups walking it in the same order.  Step-invariant pieces (K/V of the condition tokens,
// linear-attention keys/values) and batch-invariant pieces (time embedding, K/V of the time token,
// AdaLN scale/shift) are precomputed into tables by seeme_denoiser_{cond,time}_tables.
#include "common.hpp"
#include "api_util.hpp"
#include "den_layout.h"
#include <utility>

#define DEN_THREADS 512
#define DEN_MAXTOK 6   // 1 latent + N<=4 condition tokens + 1 time token
#ifndef DEN_CHMAX
#define DEN_CHMAX 8    // 16-B vectors per lane per chunk (two chunks in flight: 2 x 8 x 4 = 64 VGPRs)
#endif

// ------------------------------------------------------------------ weight element types
struct WF32 { typedef float T; static constexpr int KV = 4; static constexpr bool HALF = false; };      // 16-B vector = 4 weights
struct WBF16 { typedef uint16_t T; static constexpr int KV = 8; static constexpr bool HALF = false; };  // 16-B vector = 8 bf16 weights
struct WF16 { typedef uint16_t T; static constexpr int KV = 8; static constexpr bool HALF = true; };    // 16-B vector = 8 fp16 weights

typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// One chunk of this thread's weight stream: up to DEN_CHMAX x 16 B per lane.
struct WBuf { u32x4 r[DEN_CHMAX]; };
// Where the NEXT GEMV's first chunk of this thread lives (weights do not depend on activations, so
// it is requested before the current GEMV's epilogue and lands while the reductions/barriers run).
// Addresses are buffer-load style: per-thread byte offset (VGPR) + wave-uniform byte offset (SGPR),
// so no 64-bit per-chunk pointers are kept in vector registers.
struct NextPre { unsigned voff; unsigned soff; int stride; int ch; };

// GEMV work split over the 512 threads.  A PyTorch [N,K] matrix is stored as [K/KV][N] 16-B vectors;
// an "item" = (k-slice ks, output n); thread tid owns items tid, tid+512, ...
template <typename WT, int K, int N>
struct GemvShape {
    static constexpr int KS = (N >= 512) ? ((N % 512 == 0) ? 1 : 2) : (512 / N);   // k-slices
    static constexpr int NQ = K / WT::KV / KS;                                      // vectors per item
    static constexpr int CH = (NQ / 2 < DEN_CHMAX) ? (NQ / 2) : DEN_CHMAX;                       // vectors per chunk
    static constexpr int CPI = NQ / CH;                                             // chunks per item
    static constexpr int IT = N * KS / DEN_THREADS;                                 // items per thread
    static constexpr int TOT = IT * CPI;                                            // chunks per thread
    static_assert(N * KS % DEN_THREADS == 0 && NQ % CH == 0 && TOT % 2 == 0 && CH >= 1, "unsupported GEMV shape");
    // per-thread byte offset of item `it` (vector (ks*NQ)*N + n)
    __device__ static __forceinline__ unsigned item_voff(int tid, int it) {
        const int idx = tid + it * DEN_THREADS;
        const int ks = idx / N, n = idx - ks * N;
        return (unsigned)(ks * NQ * N + n) * 16u;
    }
    // wave-uniform byte offset of chunk c of an item, relative to the matrix start
    static constexpr unsigned chunk_soff(int c) { return (unsigned)(c * CH) * N * 16u; }
    __device__ static __forceinline__ NextPre pre(int tid, long long w_elem_off) {
        return NextPre{item_voff(tid, 0), (unsigned)(w_elem_off * (long long)sizeof(typename WT::T)), N * 16, CH};
    }
};

__device__ __forceinline__ void issue_rt(WBuf& b, __amdgpu_buffer_rsrc_t rsrc, const NextPre np) {
#pragma unroll
    for (int i = 0; i < DEN_CHMAX; ++i)
        if (i < np.ch) b.r[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, np.voff, np.soff + (unsigned)(i * np.stride), 0);
}
template <int CH>
__device__ __forceinline__ void issue_n(WBuf& b, __amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, int stride) {
#ifdef DEN_DBG_NOLOAD   // timing probe: arithmetic and epilogues without the weight stream
    return;
#endif
#pragma unroll
    for (int i = 0; i < CH; ++i) b.r[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff + (unsigned)(i * stride), 0);
}
template <int CH, int STRIDE>
__device__ __forceinline__ void issue_ct(WBuf& b, __amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
#ifdef DEN_DBG_NOLOAD
    return;
#endif
#pragma unroll
    for (int i = 0; i < CH; ++i) b.r[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff + (unsigned)(i * STRIDE), 0);
}

__device__ __forceinline__ float bf_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

// acc[s] += W-chunk . x[s][k0 ...]   (x in LDS, broadcast reads; packed fp32 FMAs)
template <typename WT, int CH, int MS>
__device__ __forceinline__ void consume(const WBuf& b, const float* __restrict__ x, int ldx, int k0, f2 (&acc)[MS][2]) {
#ifdef DEN_DBG_NOFMA   // timing probe: keep the weight stream, drop the arithmetic (results are garbage)
#pragma unroll
    for (int i = 0; i < CH; ++i) acc[0][0].x += __uint_as_float(b.r[i].x ^ b.r[i].y ^ b.r[i].z ^ b.r[i].w);
    return;
#endif
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const u32x4 u = b.r[i];
#pragma unroll
        for (int s = 0; s < MS; ++s) {
            const float* xp = x + s * ldx + k0 + WT::KV * i;
            const float4 x0 = *reinterpret_cast<const float4*>(xp);
            if constexpr (WT::KV == 4) {
                acc[s][0] += f2{__uint_as_float(u.x), __uint_as_float(u.y)} * f2{x0.x, x0.y};
                acc[s][1] += f2{__uint_as_float(u.z), __uint_as_float(u.w)} * f2{x0.z, x0.w};
            } else if constexpr (WT::HALF) {
                // fp16 weights: v_fma_mix_f32 takes the half operand directly (no unpack instructions)
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                const float4 x1 = *reinterpret_cast<const float4*>(xp + 4);
                const unsigned ux = u.x, uy = u.y, uz = u.z, uw = u.w;   // (bit_cast straight from a vector element picks element 0)
                const h2 w0 = __builtin_bit_cast(h2, ux), w1 = __builtin_bit_cast(h2, uy), w2 = __builtin_bit_cast(h2, uz), w3 = __builtin_bit_cast(h2, uw);
                acc[s][0].x = fmaf((float)w0.x, x0.x, acc[s][0].x); acc[s][0].y = fmaf((float)w0.y, x0.y, acc[s][0].y);
                acc[s][1].x = fmaf((float)w1.x, x0.z, acc[s][1].x); acc[s][1].y = fmaf((float)w1.y, x0.w, acc[s][1].y);
                acc[s][0].x = fmaf((float)w2.x, x1.x, acc[s][0].x); acc[s][0].y = fmaf((float)w2.y, x1.y, acc[s][0].y);
                acc[s][1].x = fmaf((float)w3.x, x1.z, acc[s][1].x); acc[s][1].y = fmaf((float)w3.y, x1.w, acc[s][1].y);
            } else {
                const float4 x1 = *reinterpret_cast<const float4*>(xp + 4);
                acc[s][0] += f2{bf_lo(u.x), bf_hi(u.x)} * f2{x0.x, x0.y};
                acc[s][1] += f2{bf_lo(u.y), bf_hi(u.y)} * f2{x0.z, x0.w};
                acc[s][0] += f2{bf_lo(u.z), bf_hi(u.z)} * f2{x1.x, x1.y};
                acc[s][1] += f2{bf_lo(u.w), bf_hi(u.w)} * f2{x1.z, x1.w};
            }
        }
    }
}

// One pipeline step t of a GEMV.  Chunk t lives in A for even t and in B for odd t (TOT is even).
// Two chunks are always in flight: on entry chunks 0 (A) and 1 (B) of THIS GEMV have already been
// requested by the previous GEMV; step t consumes chunk t and re-fills its buffer with chunk t+2, or,
// for the last two steps, with chunks 0 / 1 of the NEXT GEMV -- those land while the epilogue of
// this GEMV runs, so the weight stream never drains at a GEMV boundary.
template <typename WT, int K, int N, int MS, int NCH, int T>
__device__ __forceinline__ void gemv_step(int tid, __amdgpu_buffer_rsrc_t rsrc, unsigned wbase, const float* __restrict__ x, int ldx,
                                          WBuf& A, WBuf& B, const NextPre& next, float* __restrict__ part,
                                          f2 (&acc)[MS][2]) {
    typedef GemvShape<WT, K, N> G;
    constexpr int it = T / G::CPI, c = T % G::CPI;
    const int idx = tid + it * DEN_THREADS;
    const int ks = idx / N, n = idx - ks * N;
    if constexpr (c == 0) {
#pragma unroll
        for (int s = 0; s < MS; ++s) { acc[s][0] = f2{0.f, 0.f}; acc[s][1] = f2{0.f, 0.f}; }
    }
    WBuf& cur = (T & 1) ? B : A;
    consume<WT, G::CH, MS>(cur, x, ldx, (ks * G::NQ + c * G::CH) * WT::KV, acc);
    if constexpr (T + 2 < G::TOT) {
        constexpr int it2 = (T + 2) / G::CPI, c2 = (T + 2) % G::CPI;
        issue_ct<G::CH, N * 16>(cur, rsrc, G::item_voff(tid, it2), wbase + G::chunk_soff(c2));
    } else if constexpr (T + 2 == G::TOT) {
        issue_n<NCH>(cur, rsrc, next.voff, next.soff, next.stride);                                   // next GEMV, chunk 0 -> A
    } else {
        issue_n<NCH>(cur, rsrc, next.voff, next.soff + (unsigned)(NCH * next.stride), next.stride);   // next GEMV, chunk 1 -> B
    }
    if constexpr (c == G::CPI - 1) {
#pragma unroll
        for (int s = 0; s < MS; ++s) part[(ks * MS + s) * N + n] = (acc[s][0].x + acc[s][0].y) + (acc[s][1].x + acc[s][1].y);
    }
    // keep the machine scheduler from hoisting later chunks' loads above this point: the register
    // budget is sized for exactly two chunks
    __builtin_amdgcn_sched_barrier(0);
}
template <typename WT, int K, int N, int MS, int NCH, int... Ts>
__device__ __forceinline__ void gemv_steps(int tid, __amdgpu_buffer_rsrc_t rsrc, unsigned wbase, const float* __restrict__ x, int ldx,
                                           WBuf& A, WBuf& B, const NextPre& next, float* __restrict__ part,
                                           f2 (&acc)[MS][2], std::integer_sequence<int, Ts...>) {
    (gemv_step<WT, K, N, MS, NCH, Ts>(tid, rsrc, wbase, x, ldx, A, B, next, part, acc), ...);
}

// part[ks][s][n] = partial dot products of W x[s]; A/B hold this GEMV's chunks 0/1 on entry and the
// NEXT GEMV's chunks 0/1 (in flight) on exit.  Ends with a barrier (partials visible).
// w_elem_off: element offset of the matrix inside the packed image.
template <typename WT, int K, int N, int MS, int NCH>
__device__ __forceinline__ void gemv_run(int tid, __amdgpu_buffer_rsrc_t rsrc, long long w_elem_off, const float* __restrict__ x,
                                         int ldx, WBuf& A, WBuf& B, const NextPre next, float* __restrict__ part) {
    f2 acc[MS][2];
    const unsigned wbase = (unsigned)(w_elem_off * (long long)sizeof(typename WT::T));
    gemv_steps<WT, K, N, MS, NCH>(tid, rsrc, wbase, x, ldx, A, B, next, part, acc,
                             std::make_integer_sequence<int, GemvShape<WT, K, N>::TOT>{});
    __syncthreads();
}

// out[s][n] = act(sum_ks part + bias).  Ends with a barrier.
template <typename WT, int K, int N, int MS>
__device__ __forceinline__ void combine_lds(int tid, const float* __restrict__ part, const float* __restrict__ bias, int act,
                                            float* __restrict__ out, int ldo) {
    constexpr int KS = GemvShape<WT, K, N>::KS;
    for (int idx = tid; idx < N * MS; idx += DEN_THREADS) {
        const int s = idx / N, n = idx - s * N;
        float v = bias[n];
#pragma unroll
        for (int k2 = 0; k2 < KS; ++k2) v += part[(k2 * MS + s) * N + n];
        out[s * ldo + n] = act_apply(v, act);
    }
    __syncthreads();
}
// N = 256 (KS = 2): the owner thread (sample ms = tid>>8, dim d = tid&255) picks up its own sum.
template <int MS>
__device__ __forceinline__ float owner256(int tid, const float* __restrict__ part) {
    if (tid >= 256 * MS) return 0.f;
    const int ms = tid >> 8, n = tid & 255;
    return part[ms * 256 + n] + part[(MS + ms) * 256 + n];
}

// ------------------------------------------------------------------ wave-local vector algebra
// Between two GEMVs every wave redundantly owns the WHOLE 256-vector, 4 consecutive dims per lane
// (dims 4*lane .. 4*lane+3).  LayerNorm / softmax / dot products are then wave reductions (DPP
// butterflies) instead of workgroup reductions, so the only workgroup barrier per GEMV is the one
// that publishes the partial sums.
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4_scale(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float4 f4_fma(float s, float4 a, float4 acc) {
    return make_float4(fmaf(s, a.x, acc.x), fmaf(s, a.y, acc.y), fmaf(s, a.z, acc.z), fmaf(s, a.w, acc.w));
}
__device__ __forceinline__ float f4_dot(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// reduce over the lanes of one head segment (64/H lanes = 256/H dims); every lane gets the result
template <bool IS_MAX>
__device__ __forceinline__ float seg_reduce(float v, int seg_lanes) {
    if (seg_lanes == 64) return wave_reduce<IS_MAX>(v);
    v = row16_reduce<IS_MAX>(v);
    if (seg_lanes == 16) return v;
    // 32-lane heads: rows {0,1} and {2,3}
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (threadIdx.x & 32) ? dpp_combine<IS_MAX>(r2, r3) : dpp_combine<IS_MAX>(r0, r1);
}
// LayerNorm over 256 dims held 4 per lane (two-pass, torch semantics); w/b pointers in LDS
__device__ __forceinline__ float4 wave_ln(float4 v, const float* __restrict__ w, const float* __restrict__ b, int lane) {
    const float mean = wave_sum(v.x + v.y + v.z + v.w) * (1.f / 256.f);
    const float4 c = make_float4(v.x - mean, v.y - mean, v.z - mean, v.w - mean);
    const float rs = 1.f / sqrtf(wave_sum(f4_dot(c, c)) * (1.f / 256.f) + 1e-5f);
    const float4 wv = ld4(w + 4 * lane), bv = ld4(b + 4 * lane);
    return make_float4(c.x * rs * wv.x + bv.x, c.y * rs * wv.y + bv.y, c.z * rs * wv.z + bv.z, c.w * rs * wv.w + bv.w);
}
__device__ __forceinline__ float4 f4_silu(float4 v) {
    return make_float4(act_apply(v.x, SEEME_ACT_SILU), act_apply(v.y, SEEME_ACT_SILU), act_apply(v.z, SEEME_ACT_SILU),
                       act_apply(v.w, SEEME_ACT_SILU));
}
// sum of the two k-slice partials of a 256-output GEMV for sample s, dims 4*lane..
template <int MS>
__device__ __forceinline__ float4 part256(const float* __restrict__ part, int s, int lane) {
    return f4_add(ld4(part + s * 256 + 4 * lane), ld4(part + (MS + s) * 256 + 4 * lane));
}

// ------------------------------------------------------------------ the persistent sampling kernel
#define FF_SA 1024   // sa_block feed-forward, hard-coded in the reference (mdiff_transformer.py:279)
#define FF_D 128     // ffn_dim (configs/modules/denoiser.yaml:5)
#define VP_LAYER (256 + 768 + 256 * 3 + FF_SA + 256 * 9 + FF_D + 256 * 4)   // floats of vector params per layer
#define XB_LD 1024   // row stride of the GEMV input buffers

struct DenKArgs {
    const void* wg; int wg_bytes; const float* vp;
    const DenLayout* lay;      // device copy of the layout (scalar loads on demand; keeps SGPR pressure low)
    int nhead;
    SeemeSampleArgs s;
};

template <typename WT, int MS>
__global__ __launch_bounds__(DEN_THREADS) void k_den_sample(const DenKArgs ka) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // wave-uniform buffer descriptor of the packed weight image (raw buffer loads: VGPR offset + SGPR offset)
    const __amdgpu_buffer_rsrc_t wg = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ka.wg), 0, ka.wg_bytes, 0x00020000);
    const float* __restrict__ vp = ka.vp;
    const DenLayout* __restrict__ lay = ka.lay;
    const SeemeSampleArgs& A = ka.s;
    const int tid0 = threadIdx.x;
    const int b = blockIdx.x, N = A.N, NS = N + 2, H = ka.nhead;
    const int seg = 64 / H;                      // lanes per attention head

    float* VP = smem;                            // [VP_LAYER]      this layer's biases / LayerNorm params
    float* TTS = VP + VP_LAYER;                  // [1536]          time-token K|V (512) + AdaLN rows (1024)
    float* CTS = TTS + 1536;                     // [MS][4][1024]   condition K|V (sa 512 | ca 512) per token
    float* XB = CTS + MS * 4 * 1024;             // [2][MS][XB_LD]  GEMV inputs, ping-pong
    float* PART = XB + 2 * MS * XB_LD;           // [2][2*MS*768]   GEMV partial sums, ping-pong
    const int PART_SZ = 2 * MS * 768;
    int pp = 0;

    typedef GemvShape<WT, 512, 256> G_SKIP;
    typedef GemvShape<WT, 256, 768> G_INP;
    typedef GemvShape<WT, 256, 256> G_SQ;
    typedef GemvShape<WT, 256, FF_SA> G_L1;
    typedef GemvShape<WT, FF_SA, 256> G_L2;
    typedef GemvShape<WT, 256, FF_D> G_F1;
    typedef GemvShape<WT, FF_D, 256> G_F2;

    const float sa_scale = 1.f / sqrtf((float)(256 / H));
    float4 lat = ld4(A.latents + (size_t)b * 256 + 4 * (tid0 & 63));   // every wave holds the latent, 4 dims per lane
    WBuf Abuf, Bbuf;                             // the two weight chunks in flight across GEMV boundaries
    {
        const NextPre p0 = G_INP::pre(tid0, lay->L[0].inp);
        issue_n<G_INP::CH>(Abuf, wg, p0.voff, p0.soff, p0.stride);
        issue_n<G_INP::CH>(Bbuf, wg, p0.voff, p0.soff + (unsigned)(G_INP::CH * p0.stride), p0.stride);
    }

#pragma unroll 1
    for (int step = 0; step < A.steps; ++step) {
        const int row = A.trow_per_sample ? A.trow[b] : A.trow[step];
        const float* __restrict__ tt = A.ttab + (size_t)row * SEEME_TROW;
        float4 xr[MS], sk0[MS], sk1[MS];
        {
            const float4 pe = ld4(vp + lay->pe0 + 4 * (tid0 & 63));       // mld_denoiser.py:210
#pragma unroll
            for (int s = 0; s < MS; ++s) { xr[s] = f4_add(lat, pe); sk0[s] = xr[s]; sk1[s] = xr[s]; }
        }

#pragma unroll 1
        for (int l = 0; l < SEEME_DEN_NL; ++l) {
            const DenLayerOff* __restrict__ L = &lay->L[l];
            // Launder the thread id once per layer: every address derived from it is then recomputed inside
            // the layer body instead of being hoisted out of the loops (which costs >100 live VGPRs and spills).
            int tid = tid0;
            asm volatile("" : "+v"(tid));
            const int lane = tid & 63;
            // ---- stage this layer's small operands in LDS (one latency per layer instead of one per epilogue:
            //      an ordinary load issued behind a weight chunk returns behind it)
            __syncthreads();                                   // everyone is done with the previous layer's copies
            for (int i = tid; i < VP_LAYER / 4; i += DEN_THREADS) st4(VP + 4 * i, ld4(vp + L->skip_b + 4 * i));
            for (int i = tid; i < 1536 / 4; i += DEN_THREADS)
                st4(TTS + 4 * i, i < 128 ? ld4(tt + l * 512 + 4 * i) : ld4(tt + 2560 + l * 1024 + 4 * (i - 128)));
            for (int i = tid; i < MS * N * 256; i += DEN_THREADS) {     // float4 index over [MS][N][1024]
                const int s = i / (N * 256), r = i - s * (N * 256), n = r >> 8, c = r & 255;
                const int bc = (MS == 2 && s == 1) ? A.B + b : b;       // CFG: s 0 = uncond (first half), s 1 = cond
                const float* src = A.ctab + ((size_t)bc * N + n) * SEEME_CROW + (c < 128 ? l * 512 + 4 * c : 2560 + l * 512 + 4 * (c - 128));
                st4(CTS + (s * 4 + n) * 1024 + 4 * c, ld4(src));
            }
            __syncthreads();
            // offsets inside VP (relative to skip_b)
            const float* v_skip_b = VP;
            const float* v_in_b = VP + (L->in_b - L->skip_b);
            const float* v_out_b = VP + (L->out_b - L->skip_b);
            const float* v_n1w = VP + (L->n1w - L->skip_b), *v_n1b = VP + (L->n1b - L->skip_b);
            const float* v_l1b = VP + (L->l1b - L->skip_b), *v_l2b = VP + (L->l2b - L->skip_b);
            const float* v_n2w = VP + (L->n2w - L->skip_b), *v_n2b = VP + (L->n2b - L->skip_b);
            const float* v_cnw = VP + (L->cnw - L->skip_b), *v_cnb = VP + (L->cnb - L->skip_b);
            const float* v_caq_b = VP + (L->caq_b - L->skip_b);
            const float* v_csnw = VP + (L->csnw - L->skip_b), *v_csnb = VP + (L->csnb - L->skip_b);
            const float* v_cao_b = VP + (L->cao_b - L->skip_b);
            const float* v_f1b = VP + (L->f1b - L->skip_b), *v_f2b = VP + (L->f2b - L->skip_b);
            const float* v_fsnw = VP + (L->fsnw - L->skip_b), *v_fsnb = VP + (L->fsnb - L->skip_b);
            const float* v_fo_b = VP + (L->fo_b - L->skip_b);

            // ---- skip connection: Linear(cat[x, xs.pop()])  (cross_attention.py:77-79)
            if (l >= 3) {
                float* xb = XB + pp * MS * XB_LD;
#pragma unroll
                for (int s = 0; s < MS; ++s) {
                    st4(xb + s * XB_LD + 4 * lane, xr[s]);
                    st4(xb + s * XB_LD + 256 + 4 * lane, l == 3 ? sk1[s] : sk0[s]);
                }
                gemv_run<WT, 512, 256, MS, G_INP::CH>(tid, wg, L->skip, xb, XB_LD, Abuf, Bbuf, G_INP::pre(tid, L->inp), PART + pp * PART_SZ);
#pragma unroll
                for (int s = 0; s < MS; ++s) xr[s] = f4_add(part256<MS>(PART + pp * PART_SZ, s, lane), ld4(v_skip_b + 4 * lane));
                pp ^= 1;
            }
            // ---- sa_block: post-norm encoder layer over [x, xf.., emb]; only token 0 is kept
            //      (mdiff_transformer.py:292-297); K/V of xf and emb come from the tables.
            {
                float* xb = XB + pp * MS * XB_LD;
#pragma unroll
                for (int s = 0; s < MS; ++s) st4(xb + s * XB_LD + 4 * lane, xr[s]);
                gemv_run<WT, 256, 768, MS, G_SQ::CH>(tid, wg, L->inp, xb, XB_LD, Abuf, Bbuf, G_SQ::pre(tid, L->outp), PART + pp * PART_SZ);
                const float* P = PART + pp * PART_SZ;
                pp ^= 1;
                float* xn = XB + pp * MS * XB_LD;
#pragma unroll
                for (int s = 0; s < MS; ++s) {
                    // in_proj output n in [0,768): q | k | v, two k-slices
                    const float4 q = f4_add(f4_add(ld4(P + s * 768 + 4 * lane), ld4(P + (MS + s) * 768 + 4 * lane)), ld4(v_in_b + 4 * lane));
                    const float4 k0 = f4_add(f4_add(ld4(P + s * 768 + 256 + 4 * lane), ld4(P + (MS + s) * 768 + 256 + 4 * lane)), ld4(v_in_b + 256 + 4 * lane));
                    const float4 v0 = f4_add(f4_add(ld4(P + s * 768 + 512 + 4 * lane), ld4(P + (MS + s) * 768 + 512 + 4 * lane)), ld4(v_in_b + 512 + 4 * lane));
                    float sc[DEN_MAXTOK];
                    sc[0] = seg_reduce<false>(f4_dot(q, k0), seg) * sa_scale;
                    float mx = sc[0];
#pragma unroll
                    for (int j = 0; j < DEN_MAXTOK - 2; ++j)
                        if (j < N) { sc[1 + j] = seg_reduce<false>(f4_dot(q, ld4(CTS + (s * 4 + j) * 1024 + 4 * lane)), seg) * sa_scale; mx = fmaxf(mx, sc[1 + j]); }
                    // the time token is the LAST of the sequence (mdiff_transformer.py:295)
                    const float st = seg_reduce<false>(f4_dot(q, ld4(TTS + 4 * lane)), seg) * sa_scale;
                    mx = fmaxf(mx, st);
                    float e0 = expf(sc[0] - mx), et = expf(st - mx), sum = e0 + et;
#pragma unroll
                    for (int j = 0; j < DEN_MAXTOK - 2; ++j) if (j < N) { sc[1 + j] = expf(sc[1 + j] - mx); sum += sc[1 + j]; }
                    const float inv = 1.f / sum;
                    float4 att = f4_scale(v0, e0 * inv);
#pragma unroll
                    for (int j = 0; j < DEN_MAXTOK - 2; ++j)
                        if (j < N) att = f4_fma(sc[1 + j] * inv, ld4(CTS + (s * 4 + j) * 1024 + 256 + 4 * lane), att);
                    att = f4_fma(et * inv, ld4(TTS + 256 + 4 * lane), att);
                    st4(xn + s * XB_LD + 4 * lane, att);
                }
                (void)NS;
            }
            {   // out_proj + residual + norm1
                float* xb = XB + pp * MS * XB_LD;
                gemv_run<WT, 256, 256, MS, G_L1::CH>(tid, wg, L->outp, xb, XB_LD, Abuf, Bbuf, G_L1::pre(tid, L->l1), PART + pp * PART_SZ);
                const float* P = PART + pp * PART_SZ;
                pp ^= 1;
                float* xn = XB + pp * MS * XB_LD;
#pragma unroll
                for (int s = 0; s < MS; ++s) {
                    const float4 v = f4_add(xr[s], f4_add(part256<MS>(P, s, lane), ld4(v_out_b + 4 * lane)));
                    xr[s] = wave_ln(v, v_n1w, v_n1b, lane);
                    st4(xn + s * XB_LD + 4 * lane, xr[s]);
                }
            }
            {   // linear1 + relu  (N = 1024, one k-slice: outputs 4*lane + 256*j)
                float* xb = XB + pp * MS * XB_LD;
                gemv_run<WT, 256, FF_SA, MS, G_L2::CH>(tid, wg, L->l1, xb, XB_LD, Abuf, Bbuf, G_L2::pre(tid, L->l2), PART + pp * PART_SZ);
                const float* P = PART + pp * PART_SZ;
                pp ^= 1;
                float* xn = XB + pp * MS * XB_LD;
#pragma unroll
                for (int s = 0; s < MS; ++s)
#pragma unroll
                    for (int j = 0; j < FF_SA / 256; ++j) {
                        const float4 h = f4_add(ld4(P + s * FF_SA + 256 * j + 4 * lane), ld4(v_l1b + 256 * j + 4 * lane));
                        st4(xn + s * XB_LD + 256 * j + 4 * lane, make_float4(fmaxf(h.x, 0.f), fmaxf(h.y, 0.f), fmaxf(h.z, 0.f), fmaxf(h.w, 0.f)));
                    }
            }
            {   // linear2 + residual + norm2, then ca_block.norm -> query input
                float* xb = XB + pp * MS * XB_LD;
                gemv_run<WT, FF_SA, 256, MS, G_SQ::CH>(tid, wg, L->l2, xb, XB_LD, Abuf, Bbuf, G_SQ::pre(tid, L->caq), PART + pp * PART_SZ);
                const float* P = PART + pp * PART_SZ;
                pp ^= 1;
                float* xn = XB + pp * MS * XB_LD;
#pragma unroll
                for (int s = 0; s < MS; ++s) {
                    const float4 v = f4_add(xr[s], f4_add(part256<MS>(P, s, lane), ld4(v_l2b + 4 * lane)));
                    xr[s] = wave_ln(v, v_n2w, v_n2b, lane);
                    st4(xn + s * XB_LD + 4 * lane, wave_ln(xr[s], v_cnw, v_cnb, lane));
                }
            }
            {   // ca_block: linear cross-attention + AdaLN (mdiff_transformer.py:219-239, 152-163)
                float* xb = XB + pp * MS * XB_LD;
                gemv_run<WT, 256, 256, MS, G_SQ::CH>(tid, wg, L->caq, xb, XB_LD, Abuf, Bbuf, G_SQ::pre(tid, L->cao), PART + pp * PART_SZ);
                const float* P = PART + pp * PART_SZ;
                pp ^= 1;
                float* xn = XB + pp * MS * XB_LD;
#pragma unroll
                for (int s = 0; s < MS; ++s) {
                    const float4 qv = f4_add(part256<MS>(P, s, lane), ld4(v_caq_b + 4 * lane));
                    const float mx = seg_reduce<true>(fmaxf(fmaxf(qv.x, qv.y), fmaxf(qv.z, qv.w)), seg);
                    const float4 e = make_float4(expf(qv.x - mx), expf(qv.y - mx), expf(qv.z - mx), expf(qv.w - mx));
                    const float inv = 1.f / seg_reduce<false>(e.x + e.y + e.z + e.w, seg);
                    const float4 qc = f4_scale(e, inv);                               // softmax over head_dim (:231)
                    // keys: softmax over the N tokens, per dim (:232)
                    float4 kr[DEN_MAXTOK - 2];
                    float4 kmx = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
                    for (int j = 0; j < DEN_MAXTOK - 2; ++j)
                        if (j < N) {
                            kr[j] = ld4(CTS + (s * 4 + j) * 1024 + 512 + 4 * lane);
                            kmx = make_float4(fmaxf(kmx.x, kr[j].x), fmaxf(kmx.y, kr[j].y), fmaxf(kmx.z, kr[j].z), fmaxf(kmx.w, kr[j].w));
                        }
                    float4 ks = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int j = 0; j < DEN_MAXTOK - 2; ++j)
                        if (j < N) {
                            kr[j] = make_float4(expf(kr[j].x - kmx.x), expf(kr[j].y - kmx.y), expf(kr[j].z - kmx.z), expf(kr[j].w - kmx.w));
                            ks = f4_add(ks, kr[j]);
                        }
                    float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int j = 0; j < DEN_MAXTOK - 2; ++j)
                        if (j < N) {
                            const float4 kc = make_float4(kr[j].x / ks.x, kr[j].y / ks.y, kr[j].z / ks.z, kr[j].w / ks.w);
                            const float dot = seg_reduce<false>(f4_dot(qc, kc), seg);                      // q . k_n per head
                            y = f4_fma(dot, ld4(CTS + (s * 4 + j) * 1024 + 768 + 4 * lane), y);            // (q k^T) v  (:236-237)
                        }
                    float4 hh = wave_ln(y, v_csnw, v_csnb, lane);
                    const float4 scl = ld4(TTS + 512 + 4 * lane), shf = ld4(TTS + 768 + 4 * lane);
                    hh = make_float4(hh.x * (1.f + scl.x) + shf.x, hh.y * (1.f + scl.y) + shf.y, hh.z * (1.f + scl.z) + shf.z, hh.w * (1.f + scl.w) + shf.w);
                    st4(xn + s * XB_LD + 4 * lane, f4_silu(hh));
                }
            }
            {   // proj_out.out_layers + residual
                float* xb = XB + pp * MS * XB_LD;
                gemv_run<WT, 256, 256, MS, G_F1::CH>(tid, wg, L->cao, xb, XB_LD, Abuf, Bbuf, G_F1::pre(tid, L->f1), PART + pp * PART_SZ);
                const float* P = PART + pp * PART_SZ;
                pp ^= 1;
                float* xn = XB + pp * MS * XB_LD;
#pragma unroll
                for (int s = 0; s < MS; ++s) {
                    xr[s] = f4_add(xr[s], f4_add(part256<MS>(P, s, lane), ld4(v_cao_b + 4 * lane)));
                    st4(xn + s * XB_LD + 4 * lane, xr[s]);
                }
            }
            {   // ffn.linear1 + gelu  (N = 128, four k-slices: lanes 0..31 hold 4 outputs each)
                float* xb = XB + pp * MS * XB_LD;
                gemv_run<WT, 256, FF_D, MS, G_F2::CH>(tid, wg, L->f1, xb, XB_LD, Abuf, Bbuf, G_F2::pre(tid, L->f2), PART + pp * PART_SZ);
                const float* P = PART + pp * PART_SZ;
                pp ^= 1;
                float* xn = XB + pp * MS * XB_LD;
                if (lane < FF_D / 4) {
#pragma unroll
                    for (int s = 0; s < MS; ++s) {
                        float4 g = ld4(v_f1b + 4 * lane);
#pragma unroll
                        for (int k2 = 0; k2 < G_F1::KS; ++k2) g = f4_add(g, ld4(P + (k2 * MS + s) * FF_D + 4 * lane));
                        st4(xn + s * XB_LD + 4 * lane, make_float4(act_apply(g.x, SEEME_ACT_GELU), act_apply(g.y, SEEME_ACT_GELU),
                                                                   act_apply(g.z, SEEME_ACT_GELU), act_apply(g.w, SEEME_ACT_GELU)));
                    }
                }
            }
            {   // ffn.linear2 -> AdaLN
                float* xb = XB + pp * MS * XB_LD;
                gemv_run<WT, FF_D, 256, MS, G_SQ::CH>(tid, wg, L->f2, xb, XB_LD, Abuf, Bbuf, G_SQ::pre(tid, L->fo), PART + pp * PART_SZ);
                const float* P = PART + pp * PART_SZ;
                pp ^= 1;
                float* xn = XB + pp * MS * XB_LD;
#pragma unroll
                for (int s = 0; s < MS; ++s) {
                    const float4 y2 = f4_add(part256<MS>(P, s, lane), ld4(v_f2b + 4 * lane));
                    float4 hh = wave_ln(y2, v_fsnw, v_fsnb, lane);
                    const float4 scl = ld4(TTS + 1024 + 4 * lane), shf = ld4(TTS + 1280 + 4 * lane);
                    hh = make_float4(hh.x * (1.f + scl.x) + shf.x, hh.y * (1.f + scl.y) + shf.y, hh.z * (1.f + scl.z) + shf.z, hh.w * (1.f + scl.w) + shf.w);
                    st4(xn + s * XB_LD + 4 * lane, f4_silu(hh));
                }
            }
            {   // ffn.proj_out.out_layers + residual; the chunk requested now belongs to the next layer
                //  (or to layer 0 of the next step)
                float* xb = XB + pp * MS * XB_LD;
                const int ln = (l + 1 < SEEME_DEN_NL) ? l + 1 : 0;
                const NextPre nx = (ln >= 3) ? G_SKIP::pre(tid, lay->L[ln].skip) : G_INP::pre(tid, lay->L[ln].inp);
                static_assert(G_SKIP::CH == G_INP::CH, "the two possible successors of the last GEMV of a layer must chunk alike");
                gemv_run<WT, 256, 256, MS, G_INP::CH>(tid, wg, L->fo, xb, XB_LD, Abuf, Bbuf, nx, PART + pp * PART_SZ);
                const float* P = PART + pp * PART_SZ;
                pp ^= 1;
#pragma unroll
                for (int s = 0; s < MS; ++s) {
                    xr[s] = f4_add(xr[s], f4_add(part256<MS>(P, s, lane), ld4(v_fo_b + 4 * lane)));
                    if (l == 0) sk0[s] = xr[s];
                    if (l == 1) sk1[s] = xr[s];
                }
            }
        }
        // ---- stack norm -> model output (cross_attention.py:82-83; mld_denoiser.py:222)
        const int lane = tid0 & 63;
        float4 e = wave_ln(xr[0], vp + lay->fnw, vp + lay->fnb, lane);
        if (MS == 2) {   // classifier-free guidance (mld.py:488-492), uncond first
            const float4 ec = wave_ln(xr[MS - 1], vp + lay->fnw, vp + lay->fnb, lane);
            const float g = A.guidance_scale;
            e = make_float4(e.x + g * (ec.x - e.x), e.y + g * (ec.y - e.y), e.z + g * (ec.z - e.z), e.w + g * (ec.w - e.w));
        }
        if (A.sched == SEEME_SCHED_NONE) { lat = e; break; }   // steps == 1 by contract: the output is the model output
        // ---- scheduler.step (mld.py:495-497; scalars prepared by seeme_amd/schedulers.py)
        {
            const float* __restrict__ c = A.coef + (size_t)step * 8;
            const float c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3], c4 = c[4], c5 = c[5], clip = c[6], ptype = c[7];
            float4 nz = make_float4(0.f, 0.f, 0.f, 0.f);
            if (A.noise != nullptr) nz = ld4(A.noise + ((size_t)step * A.B + b) * 256 + 4 * lane);
            const float xs[4] = {lat.x, lat.y, lat.z, lat.w}, es[4] = {e.x, e.y, e.z, e.w}, ns[4] = {nz.x, nz.y, nz.z, nz.w};
            float o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float x0, ep;
                if (ptype == 0.f) { ep = es[i]; x0 = (xs[i] - c1 * ep) / c0; }
                else              { x0 = es[i]; ep = (xs[i] - c0 * x0) / c1; }
                if (clip != 0.f) x0 = fminf(fmaxf(x0, -1.f), 1.f);
                o[i] = c2 * x0 + c3 * ep + c5 * xs[i] + c4 * ns[i];
            }
            lat = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    if (tid0 < 64) st4(A.out + (size_t)b * 256 + 4 * tid0, lat);
    // the last requested chunk is never consumed: keep it from being optimised into a dangling load
    asm volatile("" ::"v"(Abuf.r[0].x), "v"(Bbuf.r[0].x));
}

static size_t den_lds_bytes(int MS) {
    return (size_t)(VP_LAYER + 1536 + MS * 4 * 1024 + 2 * MS * XB_LD + 2 * (2 * MS * 768)) * sizeof(float);
}

template <typename WT, int MS>
static int launch_den(const DenKArgs& ka, hipStream_t st) {
    const size_t lds = den_lds_bytes(MS);
    SEEME_HIP(hipFuncSetAttribute((const void*)k_den_sample<WT, MS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_den_sample<WT, MS>), dim3(ka.s.B), dim3(DEN_THREADS), lds, st, ka);
    return seeme_check_launch("k_den_sample");
}

extern "C" int seeme_denoiser_sample(const SeemeDenoiserWeights* w, const SeemeSampleArgs* a, void* stream) {
    if (a->B <= 0) return seeme_fail("denoiser_sample: B must be > 0");
    if (a->N < 1 || a->N > DEN_MAXTOK - 2) return seeme_fail("denoiser_sample: 1 <= N <= 4 condition tokens");
    if (w->nhead != 1 && w->nhead != 2 && w->nhead != 4) return seeme_fail("denoiser_sample: nhead must be 1, 2 or 4");
    if (w->ff_sa != FF_SA || w->ff != FF_D) return seeme_fail("denoiser_sample: built for sa ff 1024 / ffn_dim 128 (all reference configs)");
    if (w->layout == nullptr) return seeme_fail("denoiser_sample: layout table missing");
    if (a->sched == SEEME_SCHED_NONE && a->steps != 1) return seeme_fail("denoiser_sample: SCHED_NONE needs steps == 1");
    if (a->steps < 1) return seeme_fail("denoiser_sample: steps must be >= 1");
    DenKArgs ka;
    ka.wg = w->wg; ka.vp = w->vp; ka.lay = reinterpret_cast<const DenLayout*>(w->layout);
    {
        const DenLayout hl = seeme_make_den_layout(FF_SA, FF_D);
        ka.wg_bytes = (int)(hl.wg_total * (w->wdtype == 0 ? 4 : 2));
    }
    ka.nhead = w->nhead; ka.s = *a;
    hipStream_t st = (hipStream_t)stream;
    if (w->wdtype == 0) return a->cfg ? launch_den<WF32, 2>(ka, st) : launch_den<WF32, 1>(ka, st);
    if (w->wdtype == 1) return a->cfg ? launch_den<WBF16, 2>(ka, st) : launch_den<WBF16, 1>(ka, st);
    if (w->wdtype == 2) return a->cfg ? launch_den<WF16, 2>(ka, st) : launch_den<WF16, 1>(ka, st);
    return seeme_fail("denoiser_sample: wdtype must be 0 (fp32), 1 (bf16) or 2 (fp16)");
}

extern "C" int seeme_den_layout(int ff_sa, int ff, int64_t* out, int cap) {
    if (cap < SEEME_DEN_LAYOUT_FIELDS) return seeme_fail("seeme_den_layout: output too small");
    const DenLayout lay = seeme_make_den_layout(ff_sa, ff);
    int k = 0;
    for (int l = 0; l < SEEME_DEN_NL; ++l) {
        const int64_t* f = reinterpret_cast<const int64_t*>(&lay.L[l]);
        for (int i = 0; i < SEEME_DEN_LAYER_FIELDS; ++i) out[k++] = f[i];
    }
    out[k++] = lay.pe0; out[k++] = lay.fnw; out[k++] = lay.fnb; out[k++] = lay.wg_total; out[k++] = lay.vp_total;
    return 0;
}

// ------------------------------------------------------------------ table builders (fused linears)
extern "C" size_t seeme_denoiser_workspace_bytes(int n_rows, int B, int N) {
    (void)B; (void)N;
    return (size_t)n_rows * 256 * 2 * sizeof(float) + 256;
}

extern "C" int seeme_denoiser_time_tables(const SeemeDenoiserWeights* w, const float* tfeat, int n_rows,
                                          float* ttab, void* workspace, size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (ws_bytes < seeme_denoiser_workspace_bytes(n_rows, 0, 0)) return seeme_fail("time_tables: workspace too small");
    float* t1 = (float*)workspace;
    float* temb = t1 + (size_t)n_rows * 256;
    int rc;
    // TimestepEmbedding: linear_2(silu(linear_1(feat)))   (tools/embeddings.py:298-305)
    if ((rc = seeme_linear_simple(st, tfeat, 256, w->time_w1, 256, w->time_b1, t1, 256, n_rows, 256, 256, SEEME_ACT_SILU, 0, nullptr, nullptr))) return rc;
    if ((rc = seeme_linear_simple(st, t1, 256, w->time_w2, 256, w->time_b2, temb, 256, n_rows, 256, 256, 0, 0, nullptr, nullptr))) return rc;
    // K|V of the time token for the 5 sa_blocks (it is re-fed unchanged to every layer, mdiff_transformer.py:294-295)
    if ((rc = seeme_linear_simple(st, temb, 256, w->kv_cat_w, 256, w->kv_cat_b, ttab, SEEME_TROW, n_rows, 2560, 256, 0, 0, nullptr, nullptr))) return rc;
    // AdaLN (scale|shift) = emb_layers(emb) = Linear(SiLU(emb)) for ca and ffn of the 5 layers (mdiff_transformer.py:158-160)
    return seeme_linear_simple(st, temb, 256, w->style_cat_w, 256, w->style_cat_b, ttab + 2560, SEEME_TROW, n_rows, 5120, 256, 0, SEEME_ACT_SILU, nullptr, nullptr);
}

extern "C" int seeme_denoiser_cond_tables(const SeemeDenoiserWeights* w, const float* cond, int Bc, int N,
                                          float* ctab, void* workspace, size_t ws_bytes, void* stream) {
    (void)workspace; (void)ws_bytes;
    hipStream_t st = (hipStream_t)stream;
    const int M = Bc * N;
    int rc;
    // sa_block K|V of the condition tokens, all layers at once
    if ((rc = seeme_linear_simple(st, cond, 256, w->kv_cat_w, 256, w->kv_cat_b, ctab, SEEME_CROW, M, 2560, 256, 0, 0, nullptr, nullptr))) return rc;
    // ca_block: key|value of text_norm(xf) for all layers in one GEMM -- the per-layer LayerNorm affine is
    // folded into the weights, so the input is the affine-free LayerNorm of the condition
    // (mdiff_transformer.py:230,234); the token softmax of the keys is applied in-kernel
    if (w->ca_fold_w == nullptr) return seeme_fail("cond_tables: folded key/value weights missing");
    if ((rc = seeme_linear_simple(st, cond, 256, w->ca_fold_w, 256, w->ca_fold_b, ctab + 2560, SEEME_CROW, M, 2560, 256, 0, 0,
                                  w->ln_ones, w->ln_zeros))) return rc;
    return 0;
}
