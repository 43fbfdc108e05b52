// vae_h16.hip -- the VAE row-tile kernels with fp16 MFMA operands (v_mfma_f32_16x16x32_f16, fp32
// accumulation): the throughput mode of MldVae.encode / decode.  Same structure as vae_kernels.hip (32-row tiles,
// 4 waves x 64 columns); what changes is the operand path:
//   * activations entering a GEMM are rounded to fp16 in LDS (row stride K+16 halves: conflict-free b128 reads);
//   * weights are fp16 copies packed in MFMA fragment order (one 1 KiB contiguous load per wave and k-block);
//   * the QKV projection writes q|k as fp16 rows and V TRANSPOSED ([B][256][Sp]) so that the P.V contraction
//     (over keys) also reads 16 contiguous bytes per lane;
//   * the residual stream, LayerNorm, softmax and all accumulators stay fp32.
// Error vs the fp32 path is measured in tests/test_gpu_parity.py (fp16: 10-bit mantissa operands).
#include "common.hpp"
#include "api_util.hpp"

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
#define HPAD 16

__device__ __forceinline__ unsigned short f2h(float x) { return __builtin_bit_cast(unsigned short, (_Float16)x); }
// The fp16 path rounds its MFMA operands to 11 bits anyway: hardware transcendentals (1 ulp) and an erf polynomial
// (Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7) cost a fraction of the IEEE library forms on the vector ALU.
__device__ __forceinline__ float h_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float h_gelu(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    return 0.5f * x * (1.f + copysignf(1.f - poly * h_exp(-z * z), x));
}
__device__ __forceinline__ float h_act(float v, int act) { return act == SEEME_ACT_GELU ? h_gelu(v) : act_apply(v, act); }

#ifdef H16_DBG_TIMES
// debug build only: cycle stamps of one workgroup of one kernel (H16_DBG_KERNEL: 1 attention, 2 linear, 3 ffn, 4 qkv) at its
// phase boundaries; the last launch of the pass wins (scripts/h16_times.py)
__device__ unsigned long long h16_dbg_times[32];
#ifndef H16_DBG_KERNEL
#define H16_DBG_KERNEL 1
#endif
#define H16_DBG(k, i) do { if (H16_DBG_KERNEL == (k) && blockIdx.x == 3 && blockIdx.y == ((k) == 1 ? 5 : 0) && threadIdx.x == 0) h16_dbg_times[i] = __builtin_readcyclecounter(); } while (0)
extern "C" int seeme_debug_h16_times(unsigned long long* host, int n) {
    SEEME_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(h16_dbg_times), sizeof(unsigned long long) * (size_t)(n < 32 ? n : 32)));
    return 0;
}
#else
#define H16_DBG(k, i) do {} while (0)
#endif
#ifndef H16_PF
#define H16_PF 4      // k-blocks of weight fragments in flight in the tile GEMMs (2 -> 4: 1 % of a B=256 pass; the tiles are bound by L2 -> L1 bytes)
#endif
#ifndef H16_ALA
#define H16_ALA 1      // A fragments (LDS) requested this many k-blocks ahead of their MFMAs: 1 or 2
#endif
#ifndef FFN_PF
#define FFN_PF H16_PF  // the FFN's first GEMM (2 n-tiles per wave)
#endif
// The first PF k-blocks of a GEMM's B fragments, requested EARLY -- at kernel start, next to the activation rows -- so
// that the cold miss of the weights (every launch finds them evicted from its XCD's L2 by the sampling kernel's 9 MB
// stream) overlaps the staging phase instead of following it.
template <int NTL, int PF>
struct BRing { uint4 br[PF][NTL]; };
template <int NTL, int PF, typename LoadB>
__device__ __forceinline__ void ring_prime(BRing<NTL, PF>& ring, int K32, LoadB loadb) {
#pragma unroll
    for (int u = 0; u < PF; ++u)
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) ring.br[u][nt] = loadb(nt, u < K32 ? u : K32 - 1);
}

template <int MTL, int NTL, int PF, typename LoadB>
__device__ __forceinline__ void tile_gemm_h16(const unsigned short* __restrict__ As, int lda, int K32, LoadB loadb,
                                              f32x4 (&acc)[MTL][NTL], BRing<NTL, PF>& ring) {
    const int lane = threadIdx.x & 63, r = lane & 15, kq = lane >> 4;
    const unsigned short* ap = As + r * lda + 8 * kq;
    // PF k-blocks of B fragments in flight; a slot is re-filled right after the MFMAs that read it.  The scheduler is
    // fenced per k-block: left alone it sinks the re-fills to just before their use (a vmcnt(0) per k-block) and the
    // pipeline collapses -- these tiles are latency-bound, the depth of this pipeline is their speed.
    uint4 (&br)[PF][NTL] = ring.br;                          // primed by the caller (ring_prime)
    // A fragments H16_ALA k-blocks ahead (slots indexed by the ring slot u, so PF must be a multiple of H16_ALA + 1 rounded up to
    // a power of two): read ONE block ahead, a fragment is requested right before an MFMA block of 128 cycles and needed right
    // after it -- a ds_read_b128 of a 2-way-conflicted operand row under four waves' traffic takes longer than that.
    constexpr int AS = H16_ALA == 1 ? 2 : 4;
    static_assert(PF % AS == 0, "A-fragment slots follow the ring slot");
    uint4 ab[AS][MTL];
#pragma unroll
    for (int d = 0; d < H16_ALA; ++d)
#pragma unroll
        for (int mt = 0; mt < MTL; ++mt) ab[d][mt] = *reinterpret_cast<const uint4*>(ap + mt * 16 * lda + (d < K32 ? d : K32 - 1) * 32);
    // Every k-block of the main loop re-fills its slot UNCONDITIONALLY, the last PF k-blocks (peeled) re-fill nothing: with
    // the re-fill under `if (kb + PF < K32)` the number of loads in flight at the next wait is not a compile-time fact, the
    // compiler assumes the smaller one and emits vmcnt(NTL-1 .. 0) -- which, vmcnt being in-order, waits for the re-fill just
    // issued: one full L2 round trip per k-block whatever the ring depth.
    const int K32r = (K32 + PF - 1) / PF * PF;
    int kb0 = 0;
    for (; kb0 + PF < K32r; kb0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int kb = kb0 + u;                          // < K32 - 1 here
            const int ka = kb + H16_ALA < K32 ? kb + H16_ALA : K32 - 1;
#ifndef H16_DBG_NOAREAD        // timing-only ablation: the A fragments of the first k-blocks are reused (wrong results)
#pragma unroll
            for (int mt = 0; mt < MTL; ++mt) ab[(u + H16_ALA) % AS][mt] = *reinterpret_cast<const uint4*>(ap + mt * 16 * lda + ka * 32);
#endif
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < MTL; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTL; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, ab[u % AS][mt]), __builtin_bit_cast(h16x8, br[u][nt]), acc[mt][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#ifndef H16_DBG_NOREFILL       // timing-only ablation: the MFMAs keep reading the primed k-blocks (wrong results)
            const int kn = kb + PF < K32 ? kb + PF : K32 - 1;
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt) br[u][nt] = loadb(nt, kn);
#endif
        }
    }
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        const int kb = kb0 + u;
        const int ka = (kb + H16_ALA < K32) ? kb + H16_ALA : K32 - 1;
#pragma unroll
        for (int mt = 0; mt < MTL; ++mt) ab[(u + H16_ALA) % AS][mt] = *reinterpret_cast<const uint4*>(ap + mt * 16 * lda + ka * 32);
        __builtin_amdgcn_sched_barrier(0);
        if (kb < K32) {
#pragma unroll
            for (int mt = 0; mt < MTL; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTL; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, ab[u % AS][mt]), __builtin_bit_cast(h16x8, br[u][nt]), acc[mt][nt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}
// B fragments from fragment-packed weights: Wp[(ntile*kstride + kb)*64 + lane]; n-tiles clamped to ntiles-1.  Buffer loads:
// one descriptor per matrix, the lane's 16-byte offset in a VGPR, (n-tile, k-block) as a SCALAR byte offset -- no 64-bit
// per-lane address arithmetic and half the address traffic of global_load (PACKED_GLOBAL_LOADS=1 keeps the old form).
#ifndef PACKED_GLOBAL_LOADS
#define PACKED_GLOBAL_LOADS 0
#endif
struct PackedSrc {
    __amdgpu_buffer_rsrc_t rs; const uint4* wp; int kstride, ntile0, ntiles; unsigned voff;
    __device__ __forceinline__ PackedSrc(const uint4* __restrict__ Wp, int kstride_, int ntile0_, int ntiles_)
        : rs(__builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(Wp), 0, 0x7FFFFFF0, 0x00020000)), wp(Wp), kstride(kstride_),
          ntile0(__builtin_amdgcn_readfirstlane(ntile0_)), ntiles(ntiles_), voff((threadIdx.x & 63) * 16u) {}
    __device__ __forceinline__ uint4 operator()(int nt, int kb) const {
        int t = ntile0 + nt; t = t < ntiles ? t : ntiles - 1;
#if PACKED_GLOBAL_LOADS
        return wp[((size_t)t * kstride + kb) * 64 + (threadIdx.x & 63)];
#else
        return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (unsigned)(t * kstride + kb) * 1024u, 0));
#endif
    }
};
template <int MTL, int NTL, int PF = H16_PF>
__device__ __forceinline__ void gemm_packed(const unsigned short* As, int lda, const uint4* __restrict__ Wp, int kstride,
                                            int ntile0, int ntiles, int K32, f32x4 (&acc)[MTL][NTL], BRing<NTL, PF>& ring) {
    tile_gemm_h16<MTL, NTL, PF>(As, lda, K32, PackedSrc(Wp, kstride, ntile0, ntiles), acc, ring);
}
template <int NTL, int PF = H16_PF>
__device__ __forceinline__ void prime_packed(BRing<NTL, PF>& ring, const uint4* __restrict__ Wp, int kstride, int ntile0, int ntiles, int K32) {
    ring_prime(ring, K32, PackedSrc(Wp, kstride, ntile0, ntiles));
}
template <int MTL, int NTL, int PF = H16_PF>
__device__ __forceinline__ void gemm_packed(const unsigned short* As, int lda, const uint4* __restrict__ Wp, int kstride,
                                            int ntile0, int ntiles, int K32, f32x4 (&acc)[MTL][NTL]) {
    BRing<NTL, PF> ring;
    prime_packed(ring, Wp, kstride, ntile0, ntiles, K32);
    gemm_packed<MTL, NTL, PF>(As, lda, Wp, kstride, ntile0, ntiles, K32, acc, ring);
}
// B fragments from a row-major fp16 matrix Bm[n][k] (rows clamped to n_valid-1)
template <int MTL, int NTL, int PF = H16_PF>
__device__ __forceinline__ void gemm_rows(const unsigned short* As, int lda, const unsigned short* __restrict__ Bm, int ldb,
                                          int n0, int n_valid, int K32, f32x4 (&acc)[MTL][NTL]) {
    const int lane = threadIdx.x & 63, r = lane & 15, kq = lane >> 4;
    auto loadb = [&](int nt, int kb) {
        int n = n0 + nt * 16 + r; n = n < n_valid ? n : n_valid - 1;
        return *reinterpret_cast<const uint4*>(Bm + (size_t)n * ldb + kb * 32 + 8 * kq);
    };
    BRing<NTL, PF> ring;
    ring_prime(ring, K32, loadb);
    tile_gemm_h16<MTL, NTL, PF>(As, lda, K32, loadb, acc, ring);
}
// accumulators -> fp16 LDS tile (the next GEMM's A operand), + bias, activation
template <int MTL, int NTL>
__device__ __forceinline__ void acc_store_h16(const f32x4 (&acc)[MTL][NTL], unsigned short* __restrict__ Hs, int ldh, int c0,
                                              const float* __restrict__ bias, int act) {
    const int lane = threadIdx.x & 63, r = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt) {
        const int c = c0 + nt * 16 + r;
        const float bv = bias ? bias[c] : 0.f;
#pragma unroll
        for (int mt = 0; mt < MTL; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) Hs[(mt * 16 + 4 * kq + i) * ldh + c] = f2h(h_act(acc[mt][nt][i] + bv, act));
    }
}

template <int MTL, int NTL>
__device__ __forceinline__ void acc_store_h16(const f32x4 (&acc)[MTL][NTL], unsigned short* __restrict__ Hs, int ldh, int c0,
                                              const BiasRegs<NTL>& bias, int act) {
    const int lane = threadIdx.x & 63, r = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt) {
        const int c = c0 + nt * 16 + r;
#pragma unroll
        for (int mt = 0; mt < MTL; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) Hs[(mt * 16 + 4 * kq + i) * ldh + c] = f2h(h_act(acc[mt][nt][i] + bias.v[nt], act));
    }
}

// ---------------------------------------------------------------------------------------------
struct LinearHArgs {
    LinearKArgs k;              // same meaning as the fp32 kernel (k.a.W unused)
    const uint4* wp; int kstride; int ntiles;
    int qkv_mode;               // 1: N = 768 -> q|k fp16 rows [M][512] and V transposed [B][256][spv]
    unsigned short* qk; unsigned short* vt; int S, spv;
    int fast;                   // (set by the launcher) float4 staging straight to fp16: no fp32 staging tile in LDS
};
__host__ __device__ static inline bool linear_h_fast(const SeemeLinearArgs& a) {
    return (a.pre_ln_w == nullptr) && ((a.K & 3) == 0) && ((a.K1 & 3) == 0) && ((a.lda & 3) == 0) &&
           ((reinterpret_cast<size_t>(a.A) & 15) == 0) &&
           (a.A2 == nullptr || (((a.lda2 & 3) == 0) && ((reinterpret_cast<size_t>(a.A2) & 15) == 0)));
}

__global__ __launch_bounds__(256) void k_linear_h(const LinearHArgs ha) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const LinearKArgs& ka = ha.k;
    const SeemeLinearArgs& a = ka.a;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int Kp = (a.K + 31) & ~31;
    const int lda_f = Kp + LDS_PAD, lda_h = Kp + HPAD;
    const bool fast = ha.fast != 0;                                    // 1: plain float4 staging, 2: + LayerNorm over K = 256 in registers
    float* Af = smem;                                                  // [32][Kp+8] fp32 staging (pre-LN / act); absent in the fast path
    float* Cs = fast ? smem : Af + TILE_M * lda_f;                     // [32][264]
    unsigned short* Ah = reinterpret_cast<unsigned short*>(Cs + TILE_M * (CH_N + LDS_PAD));   // [32][Kp+16] fp16
    const int ldc = CH_N + LDS_PAD;
    const int m0 = blockIdx.x * TILE_M, cn0 = blockIdx.y * CH_N;
    H16_DBG(2, 0);
    const int n0 = cn0 + wave * 64;
    const BiasRegs<4> bias = bias_load<4>(a.bias, n0, a.N);
    BRing<4, H16_PF> ring;
    if (n0 < a.N) prime_packed(ring, ha.wp, ha.kstride, n0 >> 4, ha.ntiles, Kp >> 5);

    if (ha.fast == 1) {   // float4 in, 4 halves out, no fp32 staging tile
        // Batches of 8 guarded loads per thread, ALL issued before the first is converted.  Written as
        // "if (valid) v = load" per element the compiler emitted branch + load + vmcnt(0) per iteration: 8-16
        // serialized memory round trips per tile, most of the time of these latency-bound kernels.  The guard is a
        // select on the address (invalid lanes read the tile's first word) and on the value.
        const int Kp4 = Kp >> 2, K4 = a.K >> 2, total = TILE_M * Kp4;
        for (int base = 0; base < total; base += 256 * 8) {
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int idx = base + tid + j * 256, row = idx / Kp4, c4 = idx - row * Kp4, m = m0 + row;
                const bool valid = idx < total && m < a.M && c4 < K4;
                int prow = valid ? m : 0;
                if (ka.seq_in > 0) prow = (prow / ka.seq_in) * ka.in_stride + (prow % ka.seq_in) + ka.in_off;
                const bool first = 4 * c4 < a.K1 || a.A2 == nullptr;
                const float* p = first ? a.A + (size_t)prow * a.lda + 4 * c4 : a.A2 + (size_t)prow * a.lda2 + (4 * c4 - a.K1);
                v[j] = *reinterpret_cast<const float4*>(valid ? p : a.A);
                if (!valid) v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int idx = base + tid + j * 256, row = idx / Kp4, c4 = idx - row * Kp4;
                if (idx < total) {
                    float4 w = v[j];
                    if (a.pre_act != SEEME_ACT_NONE)
                        w = make_float4(act_apply(w.x, a.pre_act), act_apply(w.y, a.pre_act), act_apply(w.z, a.pre_act), act_apply(w.w, a.pre_act));
                    if (!(m0 + row < a.M && c4 < K4)) w = make_float4(0.f, 0.f, 0.f, 0.f);
                    const unsigned lo = (unsigned)f2h(w.x) | ((unsigned)f2h(w.y) << 16), hi = (unsigned)f2h(w.z) | ((unsigned)f2h(w.w) << 16);
                    *reinterpret_cast<uint2*>(Ah + row * lda_h + 4 * c4) = make_uint2(lo, hi);
                }
            }
        }
        __syncthreads();
    } else if (ha.fast == 2) {   // pre-LayerNorm over K = 256: a wave normalises its 8 rows in registers (4 columns per lane)
        const LnParams lp = ln_params256(a.pre_ln_w, a.pre_ln_b);
        float4 xv[8];
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const int m = m0 + wave * 8 + rr;
            int prow = m < a.M ? m : 0;
            if (ka.seq_in > 0) prow = (prow / ka.seq_in) * ka.in_stride + (prow % ka.seq_in) + ka.in_off;
            xv[rr] = *reinterpret_cast<const float4*>(a.A + (size_t)prow * a.lda + lane * 4);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const int row = wave * 8 + rr;
            float4 v = wave_layernorm256(xv[rr], lp, a.eps);
            if (a.pre_act != SEEME_ACT_NONE)
                v = make_float4(act_apply(v.x, a.pre_act), act_apply(v.y, a.pre_act), act_apply(v.z, a.pre_act), act_apply(v.w, a.pre_act));
            if (m0 + row >= a.M) v = make_float4(0.f, 0.f, 0.f, 0.f);
            const unsigned lo = (unsigned)f2h(v.x) | ((unsigned)f2h(v.y) << 16), hi = (unsigned)f2h(v.z) | ((unsigned)f2h(v.w) << 16);
            *reinterpret_cast<uint2*>(Ah + row * lda_h + lane * 4) = make_uint2(lo, hi);
        }
        __syncthreads();
    } else {
        for (int idx = tid; idx < TILE_M * Kp; idx += 256) {
            const int row = idx / Kp, c = idx - row * Kp, m = m0 + row;
            float v = 0.f;
            if (m < a.M && c < a.K) {
                int prow = m;
                if (ka.seq_in > 0) prow = (m / ka.seq_in) * ka.in_stride + (m % ka.seq_in) + ka.in_off;
                v = (c < a.K1) ? a.A[(size_t)prow * a.lda + c] : a.A2[(size_t)prow * a.lda2 + (c - a.K1)];
            }
            Af[row * lda_f + c] = v;
        }
        __syncthreads();
        if (a.pre_ln_w != nullptr) {
            for (int rr = 0; rr < 8; ++rr) {
                const int row = wave * 8 + rr;
                float s = 0.f;
                for (int c = lane; c < a.K; c += 64) s += Af[row * lda_f + c];
                const float mean = wave_sum(s) / (float)a.K;
                float q = 0.f;
                for (int c = lane; c < a.K; c += 64) { const float d = Af[row * lda_f + c] - mean; q += d * d; }
                const float rs = 1.f / sqrtf(wave_sum(q) / (float)a.K + a.eps);
                for (int c = lane; c < a.K; c += 64) Af[row * lda_f + c] = (Af[row * lda_f + c] - mean) * rs * a.pre_ln_w[c] + a.pre_ln_b[c];
            }
            __syncthreads();
        }
        for (int idx = tid; idx < TILE_M * Kp; idx += 256) {
            const int row = idx / Kp, c = idx - row * Kp;
            Ah[row * lda_h + c] = f2h(c < a.K ? act_apply(Af[row * lda_f + c], a.pre_act) : 0.f);
        }
        __syncthreads();
    }

    H16_DBG(2, 1);
    f32x4 acc[2][4];
    acc_zero(acc);
    if (n0 < a.N) gemm_packed<2, 4>(Ah, lda_h, ha.wp, ha.kstride, n0 >> 4, ha.ntiles, Kp >> 5, acc, ring);
    H16_DBG(2, 2);
    acc_store_lds<2, 4>(acc, Cs, ldc, wave * 64, bias, a.act);
    __syncthreads();
    H16_DBG(2, 3);

    if (ha.qkv_mode) {
        if (blockIdx.y < 2) {          // q | k -> fp16 rows [M][512]
            for (int idx = tid; idx < TILE_M * 64; idx += 256) {
                const int row = idx >> 6, c4 = (idx & 63) * 4, m = m0 + row;
                if (m >= a.M) continue;
                const float4 v = *reinterpret_cast<const float4*>(Cs + row * ldc + c4);
                const unsigned lo = (unsigned)f2h(v.x) | ((unsigned)f2h(v.y) << 16), hi = (unsigned)f2h(v.z) | ((unsigned)f2h(v.w) << 16);
                *reinterpret_cast<uint2*>(ha.qk + (size_t)m * 512 + blockIdx.y * 256 + c4) = make_uint2(lo, hi);
            }
        } else {                       // v -> transposed [b][d][s]: lanes <-> rows (consecutive s), loop over d
            const int row = tid & 31, dg = tid >> 5, m = m0 + row;
            if (m < a.M) {
                const int b = m / ha.S, s = m - b * ha.S;
                unsigned short* base = ha.vt + ((size_t)b * 256) * ha.spv + s;
                for (int j = 0; j < 32; ++j) {
                    const int d = dg * 32 + j;
                    base[(size_t)d * ha.spv] = f2h(Cs[row * ldc + d]);
                }
            }
        }
        return;
    }
    const bool vec_ok = ((a.ldy & 3) == 0) && ((a.N & 3) == 0);
    // residual rows as float4, all 8 of the wave requested together (when rows are 16-byte aligned and whole)
    const bool res_vec = a.res != nullptr && ((a.ldr & 3) == 0) && ((a.N & 3) == 0) && ((reinterpret_cast<size_t>(a.res) & 15) == 0);
    const LnParams lp = ln_params256(a.ln_w, a.ln_b);
    float4 rv[8];
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        rv[rr] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (res_vec) {
            const int m = m0 + wave * 8 + rr, mc = m < a.M ? m : 0, g = cn0 + lane * 4;
            int orow = mc;
            if (ka.seq_in > 0) orow = (mc / ka.seq_in) * ka.out_stride + (mc % ka.seq_in) + ka.out_off;
            const size_t rrow = ka.res_mode == 1 ? (size_t)((mc % ka.seq_in) + ka.res_off)
                              : ka.res_mode == 2 ? (size_t)(mc / ka.seq_in) : (size_t)orow;
            rv[rr] = *reinterpret_cast<const float4*>(a.res + rrow * a.ldr + (g + 3 < a.N ? g : 0));
            if (g + 3 >= a.N) rv[rr] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        const int row = wave * 8 + rr, m = m0 + row;
        if (m >= a.M) continue;
        int orow = m;
        if (ka.seq_in > 0) orow = (m / ka.seq_in) * ka.out_stride + (m % ka.seq_in) + ka.out_off;
        const int c = lane * 4, g = cn0 + c;
        float4 v = *reinterpret_cast<const float4*>(Cs + row * ldc + c);
        if (res_vec) {
            v = make_float4(v.x + rv[rr].x, v.y + rv[rr].y, v.z + rv[rr].z, v.w + rv[rr].w);
        } else if (a.res != nullptr) {
            const size_t rrow = ka.res_mode == 1 ? (size_t)((m % ka.seq_in) + ka.res_off)
                              : ka.res_mode == 2 ? (size_t)(m / ka.seq_in) : (size_t)orow;
            const float* rp = a.res + rrow * a.ldr + g;
            if (g + 0 < a.N) v.x += rp[0];
            if (g + 1 < a.N) v.y += rp[1];
            if (g + 2 < a.N) v.z += rp[2];
            if (g + 3 < a.N) v.w += rp[3];
        }
        if (a.ln_w != nullptr) v = wave_layernorm256(v, lp, a.eps);
        float* yp = a.Y + (size_t)orow * a.ldy + g;
        if (vec_ok && g + 3 < a.N) {
            *reinterpret_cast<float4*>(yp) = v;
        } else {
            if (g + 0 < a.N) yp[0] = v.x;
            if (g + 1 < a.N) yp[1] = v.y;
            if (g + 2 < a.N) yp[2] = v.z;
            if (g + 3 < a.N) yp[3] = v.w;
        }
    }
    H16_DBG(2, 4);
}

static int launch_linear_h(const LinearHArgs& ha, hipStream_t st) {
    const SeemeLinearArgs& a = ha.k.a;
    const int Kp = (a.K + 31) & ~31;
    if (Kp > 512) return seeme_fail("linear_h: K > 512 not supported");
    if (a.ln_w && a.N != 256) return seeme_fail("linear_h: fused LayerNorm needs N == 256");
    LinearHArgs h2 = ha;
    const bool ln_fast = a.pre_ln_w != nullptr && a.K == 256 && a.K1 == 256 && a.A2 == nullptr && ((a.lda & 3) == 0) &&
                         ((reinterpret_cast<size_t>(a.A) & 15) == 0);
    h2.fast = linear_h_fast(a) ? 1 : (ln_fast ? 2 : 0);
    // the fast path needs no fp32 staging tile: 51 KB (K = 256) instead of 85 KB, i.e. 3 workgroups per CU instead of 1
    const size_t lds = (size_t)((h2.fast ? 0 : TILE_M * (Kp + LDS_PAD)) + TILE_M * (CH_N + LDS_PAD)) * 4 + (size_t)TILE_M * (Kp + HPAD) * 2;
    dim3 grid((a.M + TILE_M - 1) / TILE_M, (a.N + CH_N - 1) / CH_N);
    SEEME_HIP(hipFuncSetAttribute((const void*)k_linear_h, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_linear_h, grid, dim3(256), lds, st, h2);
    return seeme_check_launch("k_linear_h");
}

// ---------------------------------------------------------------------------------------------
// k_qkv_h: the QKV projection of one transformer layer, [M,256] fp32 -> q | k fp16 rows [M][512] and V^T [B][256][spv],
// as a 128-row x 256-column tile per workgroup (8 waves = 2 row halves x 4 column quarters).  Against the generic
// 32-row tile: 4x fewer re-reads of the 128 KB weight chunk from L2, two waves per SIMD, no fp32 staging or epilogue
// tile (the fp16 output tile reuses the operand tile's LDS).
#define QKV_MT 128
// ALL3 = false: grid (row tiles, 3), a workgroup produces one of q | k | v for its 128 rows -- three times as many
// workgroups for the small launches, which are latency-bound.  ALL3 = true: grid (row tiles), q, k and v from ONE staged
// operand tile -- at large batch the kernel is HBM-bound (B=512: 460 MB per launch at 4 TB/s) and two thirds of its reads
// were the same fp32 rows fetched by the three column blocks.
template <bool ALL3>
__global__ __launch_bounds__(512) void k_qkv_h(const LinearHArgs ha) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const SeemeLinearArgs& a = ha.k.a;
    const int tid = threadIdx.x, wave = tid >> 6;
    const int mh = wave >> 2, nq = wave & 3;
    constexpr int LDH = 256 + HPAD;
    unsigned short* Ah = reinterpret_cast<unsigned short*>(smem);            // [128][272] fp16 operand tile (ALL3 = false: later the output tile)
    unsigned short* Oh = ALL3 ? Ah + QKV_MT * LDH : Ah;                      // output tile
    const int m0 = blockIdx.x * QKV_MT;
    const int y_begin = ALL3 ? 0 : (int)blockIdx.y, y_end = ALL3 ? 3 : (int)blockIdx.y + 1;
    BRing<4, H16_PF> ring;                                                   // weights of the first part, requested next to the rows
    prime_packed(ring, ha.wp, ha.kstride, y_begin * 16 + nq * 4, ha.ntiles, 8);
    H16_DBG(4, 0);
    {   // stage: 128 rows x 64 float4, all of a thread's loads in flight together
        constexpr int NIT = QKV_MT * 64 / 512;
        float4 v[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * 512, row = idx >> 6, c4 = idx & 63, m = m0 + row;
            // guard as a select on address and value (see k_linear_h: a branch per load serialises the round trips)
            v[it] = *reinterpret_cast<const float4*>(a.A + (m < a.M ? (size_t)m * a.lda + 4 * c4 : (size_t)0));
            if (m >= a.M) v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * 512, row = idx >> 6, c4 = idx & 63;
            const unsigned lo = (unsigned)f2h(v[it].x) | ((unsigned)f2h(v[it].y) << 16), hi = (unsigned)f2h(v[it].z) | ((unsigned)f2h(v[it].w) << 16);
            *reinterpret_cast<uint2*>(Ah + row * LDH + 4 * c4) = make_uint2(lo, hi);
        }
    }
    __syncthreads();
    H16_DBG(4, 1);
    for (int y = y_begin; y < y_end; ++y) {
        f32x4 acc[4][4];
        acc_zero(acc);
        const BiasRegs<4> bias = bias_load<4>(a.bias, y * 256 + nq * 64, 768);
        gemm_packed<4, 4>(Ah + mh * 64 * LDH, LDH, ha.wp, ha.kstride, y * 16 + nq * 4, ha.ntiles, 8, acc, ring);
        if (y + 1 < y_end) prime_packed(ring, ha.wp, ha.kstride, (y + 1) * 16 + nq * 4, ha.ntiles, 8);   // under this part's stores
        H16_DBG(4, 2);
        __syncthreads();                                                      // ALL3 = false: operand tile consumed, it becomes the output tile;
                                                                              // ALL3 = true: the previous part's stores have read the output tile
        acc_store_h16<4, 4>(acc, Oh + mh * 64 * LDH, LDH, nq * 64, bias, SEEME_ACT_NONE);
        __syncthreads();
        H16_DBG(4, 3);
        if (y < 2) {            // q | k: fp16 rows, 16-byte coalesced stores
            for (int idx = tid; idx < QKV_MT * 32; idx += 512) {
                const int row = idx >> 5, c8 = (idx & 31) * 8, m = m0 + row;
                if (m < a.M) *reinterpret_cast<uint4*>(ha.qk + (size_t)m * 512 + y * 256 + c8) = *reinterpret_cast<const uint4*>(Oh + row * LDH + c8);
            }
        } else {                // v -> transposed [b][d][s]: lanes <-> rows (consecutive s), loop over d
            const int row = tid & 127, dg = tid >> 7, m = m0 + row;
            if (m < a.M) {
                const int b = m / ha.S, s2 = m - b * ha.S;
                unsigned short* base = ha.vt + ((size_t)b * 256) * ha.spv + s2;
#pragma unroll 8
                for (int j = 0; j < 64; ++j) {
                    const int d = dg * 64 + j;
                    base[(size_t)d * ha.spv] = Oh[row * LDH + d];
                }
            }
        }
    }
    H16_DBG(4, 4);
}
#ifndef QKV_ALL3_TILES
#define QKV_ALL3_TILES 256   // from this many 128-row tiles on (one per CU), q | k | v come from one staged operand tile
#endif
static int launch_qkv_h(const LinearHArgs& ha, hipStream_t st) {
    const SeemeLinearArgs& a = ha.k.a;
    const int tiles = (a.M + QKV_MT - 1) / QKV_MT;
    if (tiles >= QKV_ALL3_TILES) {
        const size_t lds = (size_t)2 * QKV_MT * (256 + HPAD) * 2;
        SEEME_HIP(hipFuncSetAttribute((const void*)k_qkv_h<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_qkv_h<true>, dim3(tiles), dim3(512), lds, st, ha);
    } else {
        const size_t lds = (size_t)QKV_MT * (256 + HPAD) * 2;
        SEEME_HIP(hipFuncSetAttribute((const void*)k_qkv_h<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_qkv_h<false>, dim3(tiles, 3), dim3(512), lds, st, ha);
    }
    return seeme_check_launch("k_qkv_h");
}

// ---------------------------------------------------------------------------------------------
struct AttnHArgs {
    const unsigned short* qk;     // [B*S][512] fp16 q | k
    const unsigned short* vt;     // [B][256][spv] fp16 V^T (zero beyond S)
    const float* res; const uint4* wo; const float* bo; const float* ln_w; const float* ln_b;
    float* out; const int32_t* lengths;
    int S, n_prefix, q_rows, Sp, spv;
    float scale, eps;
};

#ifndef ATT_PF
#define ATT_PF 4    // k-blocks of B operands (K rows, V^T rows, W_o) in flight: the tile is latency-bound, not MFMA-bound
#endif
template <int NC, int RW = 8>      // NC = Sp / 64 score columns per lane; RW rows per wave
__device__ __forceinline__ void attn_softmax_rows(const float* Ps, int ldp, unsigned short* Ph, int ldph, int wave, int lane) {
    // Ph MAY ALIAS Ps (no __restrict__): the fp16 probabilities of a row overwrite the head of its own fp32 score row, which
    // is legal because a wave reads all of its 8 rows into registers before it writes any, and no other wave touches them
    float v[RW][NC], mx[RW], sum[RW];
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
        const float* prow = Ps + (wave * RW + rr) * ldp + lane;
#pragma unroll
        for (int j = 0; j < NC; ++j) v[rr][j] = prow[64 * j];
        mx[rr] = v[rr][0];
#pragma unroll
        for (int j = 1; j < NC; ++j) mx[rr] = fmaxf(mx[rr], v[rr][j]);
    }
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) mx[rr] = wave_max(mx[rr]);
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
        sum[rr] = 0.f;
#pragma unroll
        for (int j = 0; j < NC; ++j) { v[rr][j] = h_exp(v[rr][j] - mx[rr]); sum[rr] += v[rr][j]; }
    }
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) sum[rr] = wave_sum(sum[rr]);
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
        const float inv = __builtin_amdgcn_rcpf(sum[rr]);
        unsigned short* hrow = Ph + (wave * RW + rr) * ldph + lane;
#pragma unroll
        for (int j = 0; j < NC; ++j) hrow[64 * j] = f2h(v[rr][j] * inv);
    }
}

__global__ __launch_bounds__(256) void k_attn_block_h(const AttnHArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 15, kq = lane >> 4;
    const int ldp = a.Sp + LDS_PAD, ldq = 256 + HPAD, ldph = 2 * ldp;
    float* Ps = smem;                                                       // [32][Sp+8] fp32 scores, later the out_proj tile
    unsigned short* Qh = reinterpret_cast<unsigned short*>(Ps + TILE_M * ldp);   // [32][272] Q, later O
    unsigned short* Ph = reinterpret_cast<unsigned short*>(Ps);             // probabilities: fp16 rows in place of the score rows
                                                                            // (50 KB instead of 68 KB per workgroup: 3 per CU)
    int b = blockIdx.y, qt = blockIdx.x;
    if ((gridDim.y & 7) == 0) {    // all query tiles of one sequence on one XCD (speed only)
        const int L = blockIdx.x + gridDim.x * blockIdx.y, rr = L & 7, q = L >> 3;
        b = rr + 8 * (q / (int)gridDim.x);
        qt = q % (int)gridDim.x;
    }
    const int q0 = qt * TILE_M;
    const size_t base = (size_t)b * a.S;
    const int n_valid_keys = min(a.S, a.n_prefix + a.lengths[b]);
    H16_DBG(1, 0);
    const BiasRegs<4> bias_o = bias_load<4>(a.bo, wave * 64, 256);
    BRing<4, ATT_PF> ring_o;
    prime_packed(ring_o, a.wo, 8, wave * 4, 16, 8);

    {   // Q tile, 8 halves per thread x 4, requested together (guard = select on address and value, no branch per load)
        uint4 qv[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = tid + it * 256, row = idx >> 5, c8 = (idx & 31) * 8, s = q0 + row;
            qv[it] = *reinterpret_cast<const uint4*>(a.qk + (base + (s < a.q_rows ? s : 0)) * 512 + c8);
            if (s >= a.q_rows) qv[it] = make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = tid + it * 256, row = idx >> 5, c8 = (idx & 31) * 8;
            *reinterpret_cast<uint4*>(Qh + row * ldq + c8) = qv[it];
        }
    }
    __syncthreads();
    H16_DBG(1, 1);
    const unsigned short* Kmat = a.qk + base * 512 + 256;
    for (int c0 = 0; c0 < a.Sp; c0 += CH_N) {
        const int n0 = c0 + wave * 64;
        f32x4 acc[2][4];
        acc_zero(acc);
        if (n0 < a.S) gemm_rows<2, 4, ATT_PF>(Qh, ldq, Kmat, 512, n0, a.S, 8, acc);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int c = n0 + nt * 16 + r;
            const bool ok = c < n_valid_keys;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int i = 0; i < 4; ++i) Ps[(mt * 16 + 4 * kq + i) * ldp + c] = ok ? acc[mt][nt][i] * a.scale : -INFINITY;
        }
    }
    __syncthreads();
    H16_DBG(1, 2);
    // softmax (fp32) -> fp16 probabilities, one exp per element.  The wave's 8 rows are held in registers and reduced
    // TOGETHER: row after row, every row paid two dependent wave reductions and three LDS passes in sequence
    // (16.8 k of the tile's 47 k cycles); unrolled, the eight chains interleave.
    if (a.Sp == 256) attn_softmax_rows<4>(Ps, ldp, Ph, ldph, wave, lane);
    else attn_softmax_rows<8>(Ps, ldp, Ph, ldph, wave, lane);
    __syncthreads();
    H16_DBG(1, 3);
    {   // O = P V : contraction over keys, V^T rows are the B operand
        f32x4 acc[2][4];
        acc_zero(acc);
        const int K32 = (n_valid_keys + 31) >> 5;
        gemm_rows<2, 4, ATT_PF>(Ph, ldph, a.vt + (size_t)b * 256 * a.spv, a.spv, wave * 64, 256, K32, acc);
        acc_store_h16<2, 4>(acc, Qh, ldq, wave * 64, nullptr, SEEME_ACT_NONE);
    }
    __syncthreads();
    H16_DBG(1, 4);
    float* Cs = Ps;
    const int ldc = 256 + LDS_PAD;
    {
        f32x4 acc[2][4];
        acc_zero(acc);
        gemm_packed<2, 4, ATT_PF>(Qh, ldq, a.wo, 8, wave * 4, 16, 8, acc, ring_o);
        acc_store_lds<2, 4>(acc, Cs, ldc, wave * 64, bias_o, SEEME_ACT_NONE);
    }
    __syncthreads();
    H16_DBG(1, 5);
    {   // residual rows and LayerNorm parameters requested together, then one pass of LN + store per row
        const LnParams lp = ln_params256(a.ln_w, a.ln_b);
        float4 xr[8];
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const int s = q0 + wave * 8 + rr;
            xr[rr] = *reinterpret_cast<const float4*>(a.res + (base + (s < a.q_rows ? s : 0)) * 256 + lane * 4);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const int row = wave * 8 + rr, s = q0 + row;
            if (s >= a.q_rows) continue;
            float4 v = *reinterpret_cast<const float4*>(Cs + row * ldc + lane * 4);
            v = make_float4(v.x + xr[rr].x, v.y + xr[rr].y, v.z + xr[rr].z, v.w + xr[rr].w);
            v = wave_layernorm256(v, lp, a.eps);
            *reinterpret_cast<float4*>(a.out + (base + s) * 256 + lane * 4) = v;
        }
    }
    H16_DBG(1, 6);
}

static int launch_attn_h(const AttnHArgs& a_in, int B, hipStream_t st) {
    AttnHArgs a = a_in;
    if (a.S <= 0 || a.S > 512) return seeme_fail("attention_h: S must be in 1..512");
    a.Sp = (a.S + CH_N - 1) / CH_N * CH_N;
    const size_t lds = (size_t)TILE_M * (a.Sp + LDS_PAD) * 4 + (size_t)TILE_M * (256 + HPAD) * 2;
    dim3 grid((a.q_rows + TILE_M - 1) / TILE_M, B);
    SEEME_HIP(hipFuncSetAttribute((const void*)k_attn_block_h, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_attn_block_h, grid, dim3(256), lds, st, a);
    return seeme_check_launch("k_attn_block_h");
}

// ---------------------------------------------------------------------------------------------
struct FfnHArgs {
    const float* x; float* out;
    const uint4* w1; const float* b1; const uint4* w2; const float* b2;
    const float* ln_w; const float* ln_b;
    const float* cvec; int cvec_ld; const float* lnc_w; const float* lnc_b;
    const float* fin_w; const float* fin_b;
    int M, FF, act, seq_rows, seq_stride, out_mode;
    float eps;
};

__global__ __launch_bounds__(256) void k_ffn_block_h(const FfnHArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int ld = 256 + LDS_PAD, ldh = 256 + HPAD;
    float* Xs = smem;                                                      // [32][264] fp32 block input (residual)
    unsigned short* Hh = reinterpret_cast<unsigned short*>(Xs + TILE_M * ld);   // [32][FF+16] hidden (FF = 128)
    float* Cs = reinterpret_cast<float*>(Hh + TILE_M * (a.FF + HPAD));     // [32][264] fp32 output tile ...
    unsigned short* Xh = reinterpret_cast<unsigned short*>(Cs);            // ... whose head first holds the [32][272] fp16 A operand
                                                                           // (dead once the hidden layer is formed: 77 KB, 2 workgroups/CU)
    const int m0 = blockIdx.x * TILE_M;
    const BiasRegs<2> bias1 = bias_load<2>(a.b1, wave * 32, a.FF);
    const BiasRegs<4> bias2 = bias_load<4>(a.b2, wave * 64, 256);
    BRing<2, FFN_PF> ring1;
    BRing<4, H16_PF> ring2;
    prime_packed(ring1, a.w1, 8, wave * 2, a.FF >> 4, 8);
    prime_packed(ring2, a.w2, a.FF >> 5, wave * 4, 16, a.FF >> 5);
    H16_DBG(3, 0);
    {   // the wave's 8 rows (and their cross-attention vectors) are requested together, then normalised / converted
        float4 xv[8], cv[8];
        const LnParams lc = ln_params256(a.lnc_w, a.lnc_b);
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const int m = m0 + wave * 8 + rr, mc = m < a.M ? m : 0;
            const int seq = mc / a.seq_rows;
            const size_t prow = (size_t)seq * a.seq_stride + (mc % a.seq_rows);
            xv[rr] = *reinterpret_cast<const float4*>(a.x + prow * 256 + lane * 4);
            cv[rr] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.cvec != nullptr) cv[rr] = *reinterpret_cast<const float4*>(a.cvec + (size_t)seq * a.cvec_ld + lane * 4);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const int row = wave * 8 + rr, m = m0 + row;
            float4 v = xv[rr];
            if (a.cvec != nullptr) {
                v = make_float4(v.x + cv[rr].x, v.y + cv[rr].y, v.z + cv[rr].z, v.w + cv[rr].w);
                v = wave_layernorm256(v, lc, a.eps);
            }
            if (m >= a.M) v = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(Xs + row * ld + lane * 4) = v;
            const unsigned lo = (unsigned)f2h(v.x) | ((unsigned)f2h(v.y) << 16), hi = (unsigned)f2h(v.z) | ((unsigned)f2h(v.w) << 16);
            *reinterpret_cast<uint2*>(Xh + row * ldh + lane * 4) = make_uint2(lo, hi);
        }
    }
    __syncthreads();
    H16_DBG(3, 1);
    const int ldhh = a.FF + HPAD;
    {   // hidden = act(W1 x + b1): FF = 128 -> 32 columns per wave
        f32x4 acc1[2][2];
        acc_zero(acc1);
        gemm_packed<2, 2, FFN_PF>(Xh, ldh, a.w1, 8, wave * 2, a.FF >> 4, 8, acc1, ring1);
        acc_store_h16<2, 2>(acc1, Hh, ldhh, wave * 32, bias1, a.act);
    }
    __syncthreads();
    H16_DBG(3, 2);
    {
        f32x4 acc2[2][4];
        acc_zero(acc2);
        gemm_packed<2, 4>(Hh, ldhh, a.w2, a.FF >> 5, wave * 4, 16, a.FF >> 5, acc2, ring2);
        acc_store_lds<2, 4>(acc2, Cs, ld, wave * 64, bias2, SEEME_ACT_NONE);
    }
    __syncthreads();
    H16_DBG(3, 3);
    const LnParams lp = ln_params256(a.ln_w, a.ln_b), lf = ln_params256(a.fin_w, a.fin_b);
#pragma unroll 2
    for (int rr = 0; rr < 8; ++rr) {
        const int row = wave * 8 + rr, m = m0 + row;
        if (m >= a.M) continue;
        float4 v = *reinterpret_cast<const float4*>(Cs + row * ld + lane * 4);
        const float4 x = *reinterpret_cast<const float4*>(Xs + row * ld + lane * 4);
        v = make_float4(v.x + x.x, v.y + x.y, v.z + x.z, v.w + x.w);
        v = wave_layernorm256(v, lp, a.eps);
        if (a.fin_w != nullptr) v = wave_layernorm256(v, lf, a.eps);
        const int seq = m / a.seq_rows, sr = m % a.seq_rows;
        size_t orow = (size_t)seq * a.seq_stride + sr;
        if (a.out_mode == 1) orow = (size_t)sr * (a.M / a.seq_rows) + seq;
        *reinterpret_cast<float4*>(a.out + orow * 256 + lane * 4) = v;
    }
    H16_DBG(3, 4);
}

static int launch_ffn_h(const FfnHArgs& a, hipStream_t st) {
    if (a.FF != 128) return seeme_fail("ffn_h: FF must be 128 (hard-coded in the reference VAE, mld_vae.py:53)");
    const size_t lds = (size_t)2 * TILE_M * (256 + LDS_PAD) * 4 + (size_t)TILE_M * (a.FF + HPAD) * 2;
    dim3 grid((a.M + TILE_M - 1) / TILE_M);
    SEEME_HIP(hipFuncSetAttribute((const void*)k_ffn_block_h, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_ffn_block_h, grid, dim3(256), lds, st, a);
    return seeme_check_launch("k_ffn_block_h");
}

// ---------------------------------------------------------------------------------------------
// k_layer_h: ONE kernel per transformer layer.  Everything after the QKV projection of a layer is local to a 32-row tile
// (attention reads all keys / values of its sequence, but writes only its own rows), and so is the QKV projection of the
// NEXT layer -- so a workgroup runs, for its 32 query rows:
//   attention -> out_proj -> +residual -> LN1 [-> +cross-attention vector -> LN (decoder)] -> FFN -> +residual -> LN
//   [-> stack LN] [-> skip linear W.cat(x, skip rows) + b] [-> q | k | V^T of the next layer] [-> final projection]
// and the next launch finds q | k | V^T of all rows in the OTHER buffer set.  Against attn + ffn + qkv launches: a third of
// the launches, the intermediate rows (mid, x) never leave the CU, and no V^T memset (the padding columns are written here).
// LDS: R0 = [32][Sp+8] fp32 (scores -> out_proj tile -> hidden operand -> FFN output tile -> tail tiles), R1 = [32][272]
// fp16 (Q -> O -> x operand -> hidden -> x operand): 51 KB at Sp = 256, three workgroups per CU.  LN1's output rows stay in
// registers (8 float4 per lane) as the FFN residual.
#ifndef LAYER_KV_PF
#define LAYER_KV_PF 4    // k-blocks of K rows / V^T rows in flight in k_layer_h's two attention GEMMs
#endif
#ifndef LAYER_W8_PF
#define LAYER_W8_PF 4    // k-blocks in flight in the 8-wave instantiation (4 or 8)
#endif
#ifndef LAYER_WPE
#define LAYER_WPE 2      // workgroups per CU the register allocation aims at (3 would need <= 168 VGPRs)
#endif
struct LayerHArgs {
    const unsigned short* q; const uint4* kp; const uint4* vp; const float* res;   // q rows [B*S][256]; keys / values fragment-packed (below)
    const uint4* wo; const float* bo; const float* n1_w; const float* n1_b;
    const int32_t* lengths;
    int S, n_prefix, q_rows, Sp, spv; float scale, eps;
    const float* cvec; int cvec_ld; const float* lnc_w; const float* lnc_b;
    const uint4* w1; const float* b1; const uint4* w2; const float* b2; const float* n2_w; const float* n2_b;
    const float* fin_w; const float* fin_b;
    float* out; int out_mode, B;
    int in_shared;     // 1: q, kp, vp and res are ONE sequence's, the same for every batch member (the decoder's first layer: its
                       // queries are zeros + positional rows, mld_vae.py:213-222 -- nothing in them depends on the sample)
    const uint4* skip_w; const float* skip_b; const float* skip_src; float* xnext;
    const uint4* qkv_w; const float* qkv_b; unsigned short* q_out; uint4* kp_out; uint4* vp_out;
    const uint4* proj_w; const float* proj_b; float* feats; int F;
};

__device__ __forceinline__ void rows_to_h16(unsigned short* Xh, int ldh, int row, int lane, float4 v) {
    const unsigned lo = (unsigned)f2h(v.x) | ((unsigned)f2h(v.y) << 16), hi = (unsigned)f2h(v.z) | ((unsigned)f2h(v.w) << 16);
    *reinterpret_cast<uint2*>(Xh + row * ldh + lane * 4) = make_uint2(lo, hi);
}

// q | K | V of a 32-row tile (fp16 operand rows in Xh) for the layer that runs next.  q: fp16 rows [B*S][256].  Keys and values
// leave in the order the attention GEMMs of the next launch read them as MFMA B operands -- one contiguous KiB per wave-load:
//   kp[((b*NT16 + key/16)*8 + d/32)*64 + ((d%32)/8)*16 + key%16]   = K[key][8 dims from 8*(d/8)]            (NT16 = ceil(S/16))
//   vp[((b*16 + d/16)*KB + key/32)*64 + ((key%32)/8)*16 + d%16]    = V[8 keys from 8*(key/8)][d]            (KB = spv/32)
// (row-major K rows / V^T rows made every wave-load touch 16 rows x 64 B; the V^T rows were written two bytes at a time).
// Value granules of keys in [S, spv) are written as zeros by the tiles that cover them (P = 0 there, but 0 x garbage is not);
// key rows in [S, 16 NT16) stay unwritten: their score columns are masked by a select.
#ifndef LAYER_TAIL_STORES_LAST
#define LAYER_TAIL_STORES_LAST 1
#endif
template <int WAVES, int RW>
__device__ __forceinline__ void tail_store_q(const unsigned short* Oh, int ld, unsigned short* __restrict__ q_out, size_t base, int q0, int S) {
    constexpr int NT = 64 * WAVES;
    const int tid = threadIdx.x;
#pragma unroll
    for (int it = 0; it < RW / 2; ++it) {
        const int idx = tid + it * NT, row = idx >> 5, c8 = (idx & 31) * 8, s = q0 + row;
        if (s < S) *reinterpret_cast<uint4*>(q_out + (base + s) * 256 + c8) = *reinterpret_cast<const uint4*>(Oh + row * ld + c8);
    }
}
template <int WAVES, int RW>
__device__ __forceinline__ void tail_store_k(const unsigned short* Oh, int ld, uint4* __restrict__ kp_out, int b, int q0, int S, int NT16) {
    constexpr int NT = 64 * WAVES;
    const int tid = threadIdx.x;
    uint4* kb_out = kp_out + ((size_t)b * NT16 + (q0 >> 4)) * 512;
#pragma unroll
    for (int it = 0; it < RW / 2; ++it) {
        const int idx = tid + it * NT, r = idx & 15, kq = (idx >> 4) & 3, kb = (idx >> 6) & 7, half = idx >> 9, row = half * 16 + r;
        if (q0 + row < S) kb_out[(half * 8 + kb) * 64 + kq * 16 + r] = *reinterpret_cast<const uint4*>(Oh + row * ld + kb * 32 + kq * 8);
    }
}
template <int WAVES, int RW>
__device__ __forceinline__ void tail_store_v(const unsigned short* Oh, int ld, uint4* __restrict__ vp_out, int b, int q0, int S, int KB) {
    constexpr int NT = 64 * WAVES;
    const int tid = threadIdx.x;
    uint4* vb_out = vp_out + (size_t)b * 16 * KB * 64;
#pragma unroll
    for (int it = 0; it < RW / 2; ++it) {
        const int idx = tid + it * NT, r = idx & 15, g = (idx >> 4) & 3, nt = (idx >> 6) & 15, kh = idx >> 10;   // kh: 32-key block inside the tile
        const int row0 = 32 * kh + 8 * g, kb = (q0 >> 5) + kh;
        const unsigned short* src = Oh + row0 * ld + nt * 16 + r;
        unsigned w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int s0 = q0 + row0 + 2 * j;
            const unsigned lo = s0 < S ? src[(2 * j) * ld] : 0u, hi = s0 + 1 < S ? src[(2 * j + 1) * ld] : 0u;
            w[j] = lo | (hi << 16);
        }
        if (kb < KB) vb_out[((size_t)nt * KB + kb) * 64 + g * 16 + r] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}
template <int WAVES, int RW, int PF>
__device__ __forceinline__ void tail_qkv_rows(const unsigned short* Xh, unsigned short* Oh, const uint4* __restrict__ qkv_w,
                                              const float* __restrict__ qkv_b, unsigned short* __restrict__ q_out,
                                              uint4* __restrict__ kp_out, uint4* __restrict__ vp_out, int b, int q0, int S, int spv,
                                              BRing<16 / WAVES, PF>& ring_t) {
    constexpr int ROWS = RW * WAVES, MTL = ROWS / 16, NTL = 16 / WAVES, CW = 256 / WAVES;
    const int tid = threadIdx.x, wave = tid >> 6;
    const int ldq = 256 + HPAD;
    const size_t base = (size_t)b * S;
    const int NT16 = (S + 15) >> 4, KB = spv >> 5;
    // All three products first, every global store last: vmcnt is one in-order counter for loads and stores, so a part's B-fragment re-fills
    // queued behind the previous part's stores waited for their write acknowledgements (round 2 measured 14-22 k cycles for this tail against
    // 2.7 k for the out_proj of the same shape).  q and K wait in the two halves of the output region (the second without row padding: together
    // they are exactly the fp32 tile's 32 x 264 x 4 bytes), V in its accumulators.
#if !LAYER_TAIL_STORES_LAST
    for (int y = 0; y < 3; ++y) {
        f32x4 acc[MTL][NTL];
        acc_zero(acc);
        const BiasRegs<NTL> bias = bias_load<NTL>(qkv_b, y * 256 + wave * CW, 768);
        gemm_packed<MTL, NTL, PF>(Xh, ldq, qkv_w, 8, y * 16 + wave * NTL, 48, 8, acc, ring_t);
        if (y < 2) prime_packed(ring_t, qkv_w, 8, (y + 1) * 16 + wave * NTL, 48, 8);
        __syncthreads();                    // the previous part's stores have read the output tile
        acc_store_h16<MTL, NTL>(acc, Oh, ldq, wave * CW, bias, SEEME_ACT_NONE);
        __syncthreads();
        if (y == 0) tail_store_q<WAVES, RW>(Oh, ldq, q_out, base, q0, S);
        else if (y == 1) tail_store_k<WAVES, RW>(Oh, ldq, kp_out, b, q0, S, NT16);
        else tail_store_v<WAVES, RW>(Oh, ldq, vp_out, b, q0, S, KB);
    }
#else
    constexpr int ldk = 256;                                   // K's staging rows: unpadded (see above)
    unsigned short* const OhK = Oh + ROWS * ldq;
    {
        f32x4 acc[MTL][NTL];
        acc_zero(acc);
        const BiasRegs<NTL> bias = bias_load<NTL>(qkv_b, 0 * 256 + wave * CW, 768);
        gemm_packed<MTL, NTL, PF>(Xh, ldq, qkv_w, 8, 0 * 16 + wave * NTL, 48, 8, acc, ring_t);
        prime_packed(ring_t, qkv_w, 8, 1 * 16 + wave * NTL, 48, 8);
        __syncthreads();                    // the earlier phases have read the output region
        acc_store_h16<MTL, NTL>(acc, Oh, ldq, wave * CW, bias, SEEME_ACT_NONE);
    }
    {
        f32x4 acc[MTL][NTL];
        acc_zero(acc);
        const BiasRegs<NTL> bias = bias_load<NTL>(qkv_b, 1 * 256 + wave * CW, 768);
        gemm_packed<MTL, NTL, PF>(Xh, ldq, qkv_w, 8, 1 * 16 + wave * NTL, 48, 8, acc, ring_t);
        prime_packed(ring_t, qkv_w, 8, 2 * 16 + wave * NTL, 48, 8);
        acc_store_h16<MTL, NTL>(acc, OhK, ldk, wave * CW, bias, SEEME_ACT_NONE);
    }
    {
        f32x4 acc[MTL][NTL];
        acc_zero(acc);
        const BiasRegs<NTL> bias = bias_load<NTL>(qkv_b, 2 * 256 + wave * CW, 768);
        gemm_packed<MTL, NTL, PF>(Xh, ldq, qkv_w, 8, 2 * 16 + wave * NTL, 48, 8, acc, ring_t);
        __syncthreads();                    // q and K tiles complete
        tail_store_q<WAVES, RW>(Oh, ldq, q_out, base, q0, S);
        tail_store_k<WAVES, RW>(OhK, ldk, kp_out, b, q0, S, NT16);
        __syncthreads();                    // ... and read
        acc_store_h16<MTL, NTL>(acc, Oh, ldq, wave * CW, bias, SEEME_ACT_NONE);
        __syncthreads();
        tail_store_v<WAVES, RW>(Oh, ldq, vp_out, b, q0, S, KB);
    }
#endif
}

// debug stamps of a layer that has the next layer's QKV as its tail (the last launch of those in a pass is what is read back)
#define LAYER_DBG(i) do { if (a.qkv_w != nullptr) H16_DBG(5, i); } while (0)
// <WAVES, RW>: RW rows per wave.  <4, 8>: 32 query rows per workgroup, 256 threads, LAYER_WPE workgroups per CU (large launches);
// <8, 4>: the same 32 rows on 512 threads, one workgroup per CU -- two waves per SIMD inside every phase (launches that do not fill
// the chip twice over); <8, 8>: 64 rows on 512 threads (SEEME_LAYER_ROWS=64, measured slower).  All three give identical bits.
template <int WAVES, int RW>
__global__ __launch_bounds__(64 * WAVES, WAVES == 4 ? LAYER_WPE : 1) void k_layer_h(const LayerHArgs a) {
    constexpr int ROWS = RW * WAVES, MTL = ROWS / 16, NTL = 16 / WAVES, CW = 256 / WAVES, N1 = 8 / WAVES, NT = 64 * WAVES;
    // k-blocks of B fragments in flight: the 8-wave / 4-row form has the registers to request a whole K = 256 operand at once
    constexpr int PFW = (WAVES == 8 && RW == 4) ? LAYER_W8_PF : H16_PF, PFA = (WAVES == 8 && RW == 4) ? LAYER_W8_PF : ATT_PF,
                  PFK = (WAVES == 8 && RW == 4) ? LAYER_W8_PF : LAYER_KV_PF, PFF = (WAVES == 8 && RW == 4) ? LAYER_W8_PF : FFN_PF;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 15, kq = lane >> 4;
    const int ldp = a.Sp + LDS_PAD, ldq = 256 + HPAD, ldph = 2 * ldp, ldc = 256 + LDS_PAD;
    float* R0 = smem;
    unsigned short* R1 = reinterpret_cast<unsigned short*>(R0 + ROWS * ldp);
    float* Ps = R0;
    unsigned short* Ph = reinterpret_cast<unsigned short*>(R0);
    unsigned short* Qh = R1;
    int b = blockIdx.y, qt = blockIdx.x;
    if ((gridDim.y & 7) == 0) {    // all query tiles of one sequence on one XCD (speed only)
        const int L = blockIdx.x + gridDim.x * blockIdx.y, rr = L & 7, q = L >> 3;
        b = rr + 8 * (q / (int)gridDim.x);
        qt = q % (int)gridDim.x;
    }
    const int q0 = qt * ROWS;
    const size_t base = (size_t)b * a.S;
    const size_t bin = a.in_shared ? 0 : (size_t)b, base_in = bin * a.S;      // batch member whose inputs this sequence reads
    const int n_valid_keys = min(a.S, a.n_prefix + a.lengths[b]);
    const BiasRegs<NTL> bias_o = bias_load<NTL>(a.bo, wave * CW, 256);
    BRing<NTL, PFA> ring_o;
    prime_packed(ring_o, a.wo, 8, wave * NTL, 16, 8);
    LAYER_DBG(0);
    {   // Q tile
        uint4 qv[4];
#pragma unroll
        for (int it = 0; it < RW / 2; ++it) {
            const int idx = tid + it * NT, row = idx >> 5, c8 = (idx & 31) * 8, s = q0 + row;
            qv[it] = *reinterpret_cast<const uint4*>(a.q + (base_in + (s < a.q_rows ? s : 0)) * 256 + c8);
            if (s >= a.q_rows) qv[it] = make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int it = 0; it < RW / 2; ++it) {
            const int idx = tid + it * NT, row = idx >> 5, c8 = (idx & 31) * 8;
            *reinterpret_cast<uint4*>(Qh + row * ldq + c8) = qv[it];
        }
    }
    __syncthreads();
    LAYER_DBG(1);
    const int NT16 = (a.S + 15) >> 4;
    const uint4* Kp = a.kp + bin * NT16 * 512;
    for (int c0 = 0; c0 < a.Sp; c0 += CH_N) {
        const int n0 = c0 + wave * CW;
        f32x4 acc[MTL][NTL];
        acc_zero(acc);
        if (n0 < a.S) gemm_packed<MTL, NTL, PFK>(Qh, ldq, Kp, 8, n0 >> 4, NT16, 8, acc);
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) {
            const int c = n0 + nt * 16 + r;
            const bool ok = c < n_valid_keys;
#pragma unroll
            for (int mt = 0; mt < MTL; ++mt)
#pragma unroll
                for (int i = 0; i < 4; ++i) Ps[(mt * 16 + 4 * kq + i) * ldp + c] = ok ? acc[mt][nt][i] * a.scale : -INFINITY;
        }
    }
    __syncthreads();
    LAYER_DBG(2);
    if (a.Sp == 256) attn_softmax_rows<4, RW>(Ps, ldp, Ph, ldph, wave, lane);
    else attn_softmax_rows<8, RW>(Ps, ldp, Ph, ldph, wave, lane);
    __syncthreads();
    LAYER_DBG(3);
    {   // O = P V
        f32x4 acc[MTL][NTL];
        acc_zero(acc);
        const int K32 = (n_valid_keys + 31) >> 5;
        gemm_packed<MTL, NTL, PFK>(Ph, ldph, a.vp + bin * 16 * (a.spv >> 5) * 64, a.spv >> 5, wave * NTL, 16, K32, acc);
        acc_store_h16<MTL, NTL>(acc, Qh, ldq, wave * CW, nullptr, SEEME_ACT_NONE);
    }
    __syncthreads();
    LAYER_DBG(4);
    float* Cs = R0;
    {
        f32x4 acc[MTL][NTL];
        acc_zero(acc);
        gemm_packed<MTL, NTL, PFA>(Qh, ldq, a.wo, 8, wave * NTL, 16, 8, acc, ring_o);
#ifdef H16_DBG_TIMES
        asm volatile("" : "+v"(acc[0][0][0]), "+v"(acc[MTL - 1][NTL - 1][3]));
        LAYER_DBG(23);
#endif
        acc_store_lds<MTL, NTL>(acc, Cs, ldc, wave * CW, bias_o, SEEME_ACT_NONE);
    }
    LAYER_DBG(24);
    // FFN weights requested now: their cold miss overlaps LN1
    const BiasRegs<N1> bias1 = bias_load<N1>(a.b1, wave * (128 / WAVES), 128);
    const BiasRegs<NTL> bias2 = bias_load<NTL>(a.b2, wave * CW, 256);
    BRing<N1, PFF> ring1;
    BRing<NTL, PFW> ring2;
    prime_packed(ring1, a.w1, 8, wave * N1, 8, 8);
    prime_packed(ring2, a.w2, 4, wave * NTL, 16, 4);
    LAYER_DBG(25);
    __syncthreads();
    LAYER_DBG(5);
    unsigned short* Xh = R1;
    float4 x1[RW];
    {   // +residual, LN1 [, + cross-attention vector, LN]; rows stay in registers, fp16 copy is the FFN operand
        const LnParams lp = ln_params256_opt(a.n1_w, a.n1_b, a.n1_w), lc = ln_params256_opt(a.lnc_w, a.lnc_b, a.n1_w);
        float4 xr[RW];
        const float4 cv = *reinterpret_cast<const float4*>((a.cvec != nullptr ? a.cvec + (size_t)b * a.cvec_ld : a.n1_w) + lane * 4);
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) {
            const int s = q0 + wave * RW + rr;
            xr[rr] = *reinterpret_cast<const float4*>(a.res + (base_in + (s < a.q_rows ? s : 0)) * 256 + lane * 4);
        }
        __builtin_amdgcn_sched_barrier(0);
        LAYER_DBG(16);
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) {
            const float4 v = *reinterpret_cast<const float4*>(Cs + (wave * RW + rr) * ldc + lane * 4);
            x1[rr] = make_float4(v.x + xr[rr].x, v.y + xr[rr].y, v.z + xr[rr].z, v.w + xr[rr].w);
        }
        LAYER_DBG(17);
        wave_layernorm256_rows<RW>(x1, lp, a.eps);
        LAYER_DBG(18);
        if (a.cvec != nullptr) {
#pragma unroll
            for (int rr = 0; rr < RW; ++rr) x1[rr] = make_float4(x1[rr].x + cv.x, x1[rr].y + cv.y, x1[rr].z + cv.z, x1[rr].w + cv.w);
            wave_layernorm256_rows<RW>(x1, lc, a.eps);
        }
        LAYER_DBG(19);
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) {
            const int row = wave * RW + rr;
            if (q0 + row >= a.q_rows) x1[rr] = make_float4(0.f, 0.f, 0.f, 0.f);
            rows_to_h16(Xh, ldq, row, lane, x1[rr]);
        }
        LAYER_DBG(20);
    }
    __syncthreads();
    LAYER_DBG(6);
    unsigned short* Hh = R1;                     // hidden [32][144] overwrites the operand once every wave has read it
    const int ldhh = 128 + HPAD;
    {
        f32x4 acc1[MTL][N1];
        acc_zero(acc1);
        gemm_packed<MTL, N1, PFF>(Xh, ldq, a.w1, 8, wave * N1, 8, 8, acc1, ring1);
        __syncthreads();
        acc_store_h16<MTL, N1>(acc1, Hh, ldhh, wave * (128 / WAVES), bias1, SEEME_ACT_GELU);
    }
    __syncthreads();
    LAYER_DBG(7);
    {
        f32x4 acc2[MTL][NTL];
        acc_zero(acc2);
        gemm_packed<MTL, NTL, PFW>(Hh, ldhh, a.w2, 4, wave * NTL, 16, 4, acc2, ring2);
        acc_store_lds<MTL, NTL>(acc2, Cs, ldc, wave * CW, bias2, SEEME_ACT_NONE);
    }
    // tail weights requested now
    const bool has_tail = a.skip_w != nullptr || a.qkv_w != nullptr || a.proj_w != nullptr;
    BRing<NTL, PFW> ring_t;
    if (a.skip_w != nullptr) prime_packed(ring_t, a.skip_w, 16, wave * NTL, 16, 8);
    else if (a.qkv_w != nullptr) prime_packed(ring_t, a.qkv_w, 8, wave * NTL, 48, 8);
    else if (a.proj_w != nullptr) prime_packed(ring_t, a.proj_w, 8, wave * NTL, (a.F + 15) >> 4, 8);
    __syncthreads();
    LAYER_DBG(8);
    {   // +residual, LN [, stack LN]; layer output rows
        const LnParams lp = ln_params256_opt(a.n2_w, a.n2_b, a.n2_w), lf = ln_params256_opt(a.fin_w, a.fin_b, a.n2_w);
        float4 v[RW];
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) {
            const float4 c = *reinterpret_cast<const float4*>(Cs + (wave * RW + rr) * ldc + lane * 4);
            v[rr] = make_float4(c.x + x1[rr].x, c.y + x1[rr].y, c.z + x1[rr].z, c.w + x1[rr].w);
        }
        wave_layernorm256_rows<RW>(v, lp, a.eps);
        if (a.fin_w != nullptr) wave_layernorm256_rows<RW>(v, lf, a.eps);
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) {
            const int row = wave * RW + rr, s = q0 + row;
            if (s >= a.q_rows) v[rr] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.out != nullptr && s < a.q_rows) {
                const size_t orow = a.out_mode == 1 ? (size_t)s * a.B + b : base + s;
                *reinterpret_cast<float4*>(a.out + orow * 256 + lane * 4) = v[rr];
            }
            if (has_tail) rows_to_h16(Xh, ldq, row, lane, v[rr]);
        }
    }
    LAYER_DBG(9);
    if (!has_tail) return;
    __syncthreads();
    if (a.skip_w != nullptr) {   // x = W_skip . cat(x, skip rows) + b : two K = 256 halves, the second operand tile in R0
        unsigned short* Sh = reinterpret_cast<unsigned short*>(R0);
        {
            float4 sv[RW];
#pragma unroll
            for (int rr = 0; rr < RW; ++rr) {
                const int s = q0 + wave * RW + rr;
                sv[rr] = *reinterpret_cast<const float4*>(a.skip_src + (base + (s < a.S ? s : 0)) * 256 + lane * 4);
                if (s >= a.S) sv[rr] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int rr = 0; rr < RW; ++rr) rows_to_h16(Sh, ldq, wave * RW + rr, lane, sv[rr]);
        }
        const BiasRegs<NTL> bias_s = bias_load<NTL>(a.skip_b, wave * CW, 256);
        __syncthreads();
        f32x4 acc[MTL][NTL];
        acc_zero(acc);
        gemm_packed<MTL, NTL, PFW>(Xh, ldq, a.skip_w, 16, wave * NTL, 16, 8, acc, ring_t);
        gemm_packed<MTL, NTL, PFW>(Sh, ldq, a.skip_w + 8 * 64, 16, wave * NTL, 16, 8, acc);
        if (a.qkv_w != nullptr) prime_packed(ring_t, a.qkv_w, 8, wave * NTL, 48, 8);
        __syncthreads();
        acc_store_lds<MTL, NTL>(acc, Cs, ldc, wave * CW, bias_s, SEEME_ACT_NONE);
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) {
            const int row = wave * RW + rr, s = q0 + row;
            float4 v = *reinterpret_cast<const float4*>(Cs + row * ldc + lane * 4);
            if (s >= a.S) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (s < a.S) *reinterpret_cast<float4*>(a.xnext + (base + s) * 256 + lane * 4) = v;
            rows_to_h16(Xh, ldq, row, lane, v);
        }
        __syncthreads();
    }
    LAYER_DBG(10);
    if (a.qkv_w != nullptr) {    // q | k | V^T of the next layer for these rows, into the other buffer set
        tail_qkv_rows<WAVES, RW, PFW>(Xh, reinterpret_cast<unsigned short*>(R0), a.qkv_w, a.qkv_b, a.q_out, a.kp_out, a.vp_out, b, q0, a.S, a.spv, ring_t);
        LAYER_DBG(11);
        return;
    }
    if (a.proj_w != nullptr) {   // final projection to the feature width (decoder): the stack LN was applied above
        const int ntiles = (a.F + 15) >> 4;
        const BiasRegs<NTL> bias = bias_load<NTL>(a.proj_b, wave * CW, a.F);
        f32x4 acc[MTL][NTL];
        acc_zero(acc);
        if (wave * NTL < ntiles) gemm_packed<MTL, NTL, PFW>(Xh, ldq, a.proj_w, 8, wave * NTL, ntiles, 8, acc, ring_t);
        acc_store_lds<MTL, NTL>(acc, Cs, ldc, wave * CW, bias, SEEME_ACT_NONE);
        __syncthreads();
        for (int idx = tid; idx < ROWS * 256; idx += NT) {
            const int row = idx >> 8, c = idx & 255, s = q0 + row;
            if (s < a.S && c < a.F) a.feats[(base + s) * a.F + c] = Cs[row * ldc + c];
        }
    }
}

// 32 query rows per workgroup.  The 64-row instantiation (SEEME_LAYER_ROWS=64; Sp <= 256 only: 102 KB of LDS) pulls half the
// weight / K / V bytes per row from L2 and measures 1-2 % SLOWER at B = 256 / 512.  Launches of at most
// SEEME_LAYER_W8_MAX workgroups (default 256 = one per CU: B = 32 at T = 196 is 224) run their 32 rows on 8 waves instead of 4.
template <int WAVES, int RW>
static int launch_layer_w(const LayerHArgs& a, hipStream_t st) {
    constexpr int ROWS = RW * WAVES;
    const size_t lds = (size_t)ROWS * (a.Sp + LDS_PAD) * 4 + (size_t)ROWS * (256 + HPAD) * 2;
    dim3 grid((a.q_rows + ROWS - 1) / ROWS, a.B);
    SEEME_HIP(hipFuncSetAttribute((const void*)k_layer_h<WAVES, RW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_layer_h<WAVES, RW>), grid, dim3(64 * WAVES), lds, st, a);
    return seeme_check_launch("k_layer_h");
}
static int launch_layer_h(const LayerHArgs& a_in, hipStream_t st) {
    LayerHArgs a = a_in;
    if (a.S <= 0 || a.S > 512) return seeme_fail("layer_h: S must be in 1..512");
    a.Sp = (a.S + CH_N - 1) / CH_N * CH_N;
    static int rows = -1, w8_max = -1;
    if (rows < 0) { const char* e = getenv("SEEME_LAYER_ROWS"); rows = e ? atoi(e) : 32; }
    if (w8_max < 0) { const char* e = getenv("SEEME_LAYER_W8_MAX"); w8_max = e ? atoi(e) : 256; }
    if (rows == 64 && a.Sp <= 256 && a.q_rows > 32) return launch_layer_w<8, 8>(a, st);
    const long wgs = (long)a.B * ((a.q_rows + 31) / 32);
    return wgs <= w8_max ? launch_layer_w<8, 4>(a, st) : launch_layer_w<4, 8>(a, st);
}


// k_vae_pro_h: the input rows of the first layer and its q | k | V^T in one launch.  mode 1 (encoder): rows 0,1 = distribution
// tokens + PE, rows 2.. = skel_embedding(features) + PE (mld_vae.py:143-160); mode 2 (decoder): the learned query PE rows
// (:213-222, the queries are zeros + PE).
struct ProHArgs {
    int mode, B, S, T, F, spv;
    const float* features; const uint4* emb_w; const float* emb_b; const float* token; const float* pe;
    float* x;
    const uint4* qkv_w; const float* qkv_b; unsigned short* q_out; uint4* kp_out; uint4* vp_out;
};
__global__ __launch_bounds__(256) void k_vae_pro_h(const ProHArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int ldq = 256 + HPAD, ldc = 256 + LDS_PAD;
    float* Cs = smem;
    unsigned short* R1 = reinterpret_cast<unsigned short*>(Cs + TILE_M * ldc);
    const int b = blockIdx.y, q0 = blockIdx.x * TILE_M;
    const size_t base = (size_t)b * a.S;
    BRing<4, H16_PF> ring_t;
    if (a.mode == 1) {
        const int Kp = (a.F + 31) & ~31, lda_h = Kp + HPAD;
        const BiasRegs<4> bias = bias_load<4>(a.emb_b, wave * 64, 256);
        for (int idx = tid; idx < TILE_M * Kp; idx += 256) {       // features of frame s - 2 (guard: select on address and value)
            const int row = idx / Kp, c = idx - row * Kp, s = q0 + row;
            const bool ok = s >= 2 && s < a.S && c < a.F;
            float v = a.features[ok ? ((size_t)b * a.T + (s - 2)) * a.F + c : 0];
            R1[row * lda_h + c] = f2h(ok ? v : 0.f);
        }
        __syncthreads();
        f32x4 acc[2][4];
        acc_zero(acc);
        gemm_packed<2, 4>(R1, lda_h, a.emb_w, Kp >> 5, wave * 4, 16, Kp >> 5, acc);
        acc_store_lds<2, 4>(acc, Cs, ldc, wave * 64, bias, SEEME_ACT_NONE);
    }
    prime_packed(ring_t, a.qkv_w, 8, wave * 4, 48, 8);
    __syncthreads();
    unsigned short* Xh = R1;
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        const int row = wave * 8 + rr, s = q0 + row, sc = s < a.S ? s : 0;
        float4 v = *reinterpret_cast<const float4*>(a.pe + (size_t)sc * 256 + lane * 4);
        if (a.mode == 1) {
            const float4 e = s < 2 ? *reinterpret_cast<const float4*>(a.token + (size_t)sc * 256 + lane * 4)
                                   : *reinterpret_cast<const float4*>(Cs + row * ldc + lane * 4);
            v = make_float4(v.x + e.x, v.y + e.y, v.z + e.z, v.w + e.w);
        }
        if (s >= a.S) v = make_float4(0.f, 0.f, 0.f, 0.f);
        else *reinterpret_cast<float4*>(a.x + (base + s) * 256 + lane * 4) = v;
        rows_to_h16(Xh, ldq, row, lane, v);
    }
    __syncthreads();
    tail_qkv_rows<4, 8, H16_PF>(Xh, reinterpret_cast<unsigned short*>(Cs), a.qkv_w, a.qkv_b, a.q_out, a.kp_out, a.vp_out, b, q0, a.S, a.spv, ring_t);
}
static int launch_pro_h(const ProHArgs& a, hipStream_t st) {
    if (a.mode == 1 && a.F > 256) return seeme_fail("vae_pro_h: nfeats > 256");
    const size_t lds = (size_t)TILE_M * (256 + LDS_PAD) * 4 + (size_t)TILE_M * (256 + HPAD) * 2;
    dim3 grid((a.S + TILE_M - 1) / TILE_M, a.B);
    SEEME_HIP(hipFuncSetAttribute((const void*)k_vae_pro_h, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_vae_pro_h, grid, dim3(256), lds, st, a);
    return seeme_check_launch("k_vae_pro_h");
}

// ---------------------------------------------------------------------------------------------
// host sequencing (mirrors vae_kernels.hip)
__global__ void k_enc_tokens_h(const float* __restrict__ token, const float* __restrict__ pe, float* __restrict__ x, int B, int S) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * 2 * 256) return;
    const int d = idx & 255, i = (idx >> 8) & 1, b = idx >> 9;
    x[((size_t)b * S + i) * 256 + d] = token[i * 256 + d] + pe[i * 256 + d];
}
__global__ void k_bcast_rows_h(const float* __restrict__ src, float* __restrict__ dst, int B, int rows) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t per = (size_t)rows * 64;
    if (idx >= per * B) return;
    reinterpret_cast<float4*>(dst)[idx] = reinterpret_cast<const float4*>(src)[idx % per];
}

struct WsH { float *x, *y, *sk0, *sk1, *cvec; unsigned short *qk, *vt; int spv; uint4* kp[2]; };
static WsH carve_h(void* ws, int B, int S) {
    const size_t R = (size_t)B * S;
    float* p = (float*)ws;
    WsH w;
    w.x = p; p += R * 256; w.y = p; p += R * 256; w.sk0 = p; p += R * 256; w.sk1 = p; p += R * 256;
    unsigned short* h = (unsigned short*)p;          // the fp32 path's qkv region [R][768] floats
    w.qk = h; h += R * 512;
    w.spv = (S + 31) & ~31;
    w.vt = h;                                        // B*256*spv halves <= R*1024 halves
    p += R * 768;
    w.cvec = p;
    p += (size_t)B * 256 * (SEEME_NLAYERS + 1) + 64;     // (vae_ws_floats) the two sets of fragment-packed keys of the one-kernel-per-layer path
    w.kp[0] = reinterpret_cast<uint4*>(p);
    w.kp[1] = reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(p) + (size_t)B * ((S + 15) & ~15) * 256);
    return w;
}

static int lin_h(hipStream_t st, const float* A, int lda, const uint16_t* wp, int K, int N, const float* bias, float* Y, int ldy,
                 int M, const float* pre_ln_w = nullptr, const float* pre_ln_b = nullptr) {
    LinearHArgs ha{};
    ha.k.a.A = A; ha.k.a.lda = lda; ha.k.a.K1 = K; ha.k.a.K = K; ha.k.a.bias = bias; ha.k.a.Y = Y; ha.k.a.ldy = ldy;
    ha.k.a.M = M; ha.k.a.N = N; ha.k.a.eps = 1e-5f; ha.k.a.pre_ln_w = pre_ln_w; ha.k.a.pre_ln_b = pre_ln_b;
    ha.wp = (const uint4*)wp; ha.kstride = ((K + 31) & ~31) >> 5; ha.ntiles = (N + 15) >> 4;
    return launch_linear_h(ha, st);
}

static int run_layer_h(hipStream_t st, const SeemeXfLayer& L, const SeemeXfLayerH& H, float* cur, float* mid, float* out, WsH& ws,
                       const int32_t* lengths, int B, int S, int n_prefix, int q_rows, const float* cvec, int cvec_ld,
                       const float* fin_w, const float* fin_b, int out_mode) {
    LinearHArgs q{};
    q.k.a.A = cur; q.k.a.lda = 256; q.k.a.K1 = 256; q.k.a.K = 256; q.k.a.bias = L.in_b; q.k.a.M = B * S; q.k.a.N = 768; q.k.a.eps = 1e-5f;
    q.wp = (const uint4*)H.in_w; q.kstride = 8; q.ntiles = 48; q.qkv_mode = 1; q.qk = ws.qk; q.vt = ws.vt; q.S = S; q.spv = ws.spv;
    int rc = ((reinterpret_cast<size_t>(cur) & 15) == 0) ? launch_qkv_h(q, st) : launch_linear_h(q, st);
    if (rc) return rc;
    AttnHArgs at{};
    at.qk = ws.qk; at.vt = ws.vt; at.res = cur; at.wo = (const uint4*)H.out_w; at.bo = L.out_b; at.ln_w = L.n1_w; at.ln_b = L.n1_b;
    at.out = mid; at.lengths = lengths; at.S = S; at.n_prefix = n_prefix; at.q_rows = q_rows; at.spv = ws.spv;
    at.scale = 1.0f / 16.0f; at.eps = 1e-5f;
    if ((rc = launch_attn_h(at, B, st))) return rc;
    FfnHArgs f{};
    f.x = mid; f.out = out; f.w1 = (const uint4*)H.l1_w; f.b1 = L.l1_b; f.w2 = (const uint4*)H.l2_w; f.b2 = L.l2_b;
    f.M = B * q_rows; f.FF = 128; f.act = SEEME_ACT_GELU; f.seq_rows = q_rows; f.seq_stride = S; f.out_mode = out_mode; f.eps = 1e-5f;
    f.fin_w = fin_w; f.fin_b = fin_b;
    if (cvec != nullptr) { f.cvec = cvec; f.cvec_ld = cvec_ld; f.lnc_w = L.n2_w; f.lnc_b = L.n2_b; f.ln_w = L.n3_w; f.ln_b = L.n3_b; }
    else { f.ln_w = L.n2_w; f.ln_b = L.n2_b; }
    return launch_ffn_h(f, st);
}

static int skip_lin_h(hipStream_t st, const float* a1, const float* a2, const uint16_t* wp, const float* b, float* y, int M) {
    LinearHArgs ha{};
    ha.k.a.A = a1; ha.k.a.lda = 256; ha.k.a.A2 = a2; ha.k.a.lda2 = 256; ha.k.a.K1 = 256; ha.k.a.K = 512; ha.k.a.bias = b;
    ha.k.a.Y = y; ha.k.a.ldy = 256; ha.k.a.M = M; ha.k.a.N = 256; ha.k.a.eps = 1e-5f;
    ha.wp = (const uint4*)wp; ha.kstride = 16; ha.ntiles = 16;
    return launch_linear_h(ha, st);
}

static int vae_encode_h16_unfused(const SeemeVaeWeights* w, const float* features, const int32_t* lengths, int B, int T, float* mu,
                         void* workspace, hipStream_t st) {
    const SeemeVaeWeightsH* H = w->h16;
    const int S = T + 2, F = w->nfeats;
    WsH ws = carve_h(workspace, B, S);
    const SeemeSkipStack& E = w->enc;
    SEEME_HIP(hipMemsetAsync(ws.vt, 0, (size_t)B * 256 * ws.spv * 2, st));
    hipLaunchKernelGGL(k_enc_tokens_h, dim3((B * 512 + 255) / 256), dim3(256), 0, st, w->token, w->pe_enc, ws.x, B, S);
    int rc = seeme_check_launch("k_enc_tokens_h");
    if (rc) return rc;
    {
        LinearHArgs ha{};
        ha.k.a.A = features; ha.k.a.lda = F; ha.k.a.K1 = F; ha.k.a.K = F; ha.k.a.bias = w->emb_b; ha.k.a.res = w->pe_enc; ha.k.a.ldr = 256;
        ha.k.a.Y = ws.x; ha.k.a.ldy = 256; ha.k.a.M = B * T; ha.k.a.N = 256; ha.k.a.eps = 1e-5f;
        ha.k.seq_in = T; ha.k.in_stride = T; ha.k.out_stride = S; ha.k.out_off = 2; ha.k.res_mode = 1; ha.k.res_off = 2;
        ha.wp = (const uint4*)H->emb_w; ha.kstride = ((F + 31) & ~31) >> 5; ha.ntiles = 16;
        if ((rc = launch_linear_h(ha, st))) return rc;
    }
    if ((rc = run_layer_h(st, E.layer[0], H->enc[0], ws.x, ws.y, ws.sk0, ws, lengths, B, S, 2, S, nullptr, 0, nullptr, nullptr, 0))) return rc;
    if ((rc = run_layer_h(st, E.layer[1], H->enc[1], ws.sk0, ws.y, ws.sk1, ws, lengths, B, S, 2, S, nullptr, 0, nullptr, nullptr, 0))) return rc;
    if ((rc = run_layer_h(st, E.layer[2], H->enc[2], ws.sk1, ws.y, ws.x, ws, lengths, B, S, 2, S, nullptr, 0, nullptr, nullptr, 0))) return rc;
    if ((rc = skip_lin_h(st, ws.x, ws.sk1, H->enc_skip[0], E.skip_b[0], ws.x, B * S))) return rc;
    if ((rc = run_layer_h(st, E.layer[3], H->enc[3], ws.x, ws.y, ws.x, ws, lengths, B, S, 2, S, nullptr, 0, nullptr, nullptr, 0))) return rc;
    if ((rc = skip_lin_h(st, ws.x, ws.sk0, H->enc_skip[1], E.skip_b[1], ws.x, B * S))) return rc;
    return run_layer_h(st, E.layer[4], H->enc[4], ws.x, ws.y, mu, ws, lengths, B, S, 2, 2, nullptr, 0, E.norm_w, E.norm_b, 1);
}

static int vae_decode_h16_unfused(const SeemeVaeWeights* w, const float* z, const int32_t* lengths, int B, int T, float* feats,
                         void* workspace, hipStream_t st) {
    const SeemeVaeWeightsH* H = w->h16;
    const int S = T, F = w->nfeats;
    WsH ws = carve_h(workspace, B, S + 2);
    ws.spv = (S + 31) & ~31;
    const SeemeSkipStack& Dk = w->dec;
    int rc;
    SEEME_HIP(hipMemsetAsync(ws.vt, 0, (size_t)B * 256 * ws.spv * 2, st));
    const int CL = SEEME_NLAYERS * 256;
    if ((rc = lin_h(st, z, 256, H->ca_fold_w, 256, CL, w->ca_fold_b, ws.cvec, CL, B))) return rc;
    {
        const size_t n4 = (size_t)B * S * 64;
        hipLaunchKernelGGL(k_bcast_rows_h, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, w->pe_dec, ws.x, B, S);
        if ((rc = seeme_check_launch("k_bcast_rows_h"))) return rc;
    }
    if ((rc = run_layer_h(st, Dk.layer[0], H->dec[0], ws.x, ws.y, ws.sk0, ws, lengths, B, S, 0, S, ws.cvec + 0 * 256, CL, nullptr, nullptr, 0))) return rc;
    if ((rc = run_layer_h(st, Dk.layer[1], H->dec[1], ws.sk0, ws.y, ws.sk1, ws, lengths, B, S, 0, S, ws.cvec + 1 * 256, CL, nullptr, nullptr, 0))) return rc;
    if ((rc = run_layer_h(st, Dk.layer[2], H->dec[2], ws.sk1, ws.y, ws.x, ws, lengths, B, S, 0, S, ws.cvec + 2 * 256, CL, nullptr, nullptr, 0))) return rc;
    if ((rc = skip_lin_h(st, ws.x, ws.sk1, H->dec_skip[0], Dk.skip_b[0], ws.x, B * S))) return rc;
    if ((rc = run_layer_h(st, Dk.layer[3], H->dec[3], ws.x, ws.y, ws.x, ws, lengths, B, S, 0, S, ws.cvec + 3 * 256, CL, nullptr, nullptr, 0))) return rc;
    if ((rc = skip_lin_h(st, ws.x, ws.sk0, H->dec_skip[1], Dk.skip_b[1], ws.x, B * S))) return rc;
    if ((rc = run_layer_h(st, Dk.layer[4], H->dec[4], ws.x, ws.y, ws.x, ws, lengths, B, S, 0, S, ws.cvec + 4 * 256, CL, nullptr, nullptr, 0))) return rc;
    return lin_h(st, ws.x, 256, H->fin_w, 256, F, w->fin_b, feats, F, B * S, Dk.norm_w, Dk.norm_b);
}

// ---------------------------------------------------------------------------------------------
// One kernel per layer (k_layer_h); SEEME_VAE_FUSED=0 selects the three-kernels-per-layer sequence above.
static bool vae_fused_enabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("SEEME_VAE_FUSED"); v = (e && e[0] == '0') ? 0 : 1; }
    return v != 0;
}
struct KvSet { unsigned short* q; uint4* kp; uint4* vp; };
static void kv_sets(const WsH& ws, int B, KvSet (&kv)[2]) {
    kv[0] = KvSet{ws.qk, ws.kp[0], reinterpret_cast<uint4*>(ws.vt)};
    kv[1] = KvSet{reinterpret_cast<unsigned short*>(ws.y), ws.kp[1], reinterpret_cast<uint4*>(ws.vt + (size_t)B * 256 * ws.spv)};
}

static LayerHArgs layer_args(const SeemeXfLayer& L, const SeemeXfLayerH& H, const KvSet& in, const float* res, const int32_t* lengths,
                             int B, int S, int spv, int n_prefix, int q_rows, const float* cvec, int cvec_ld) {
    LayerHArgs a{};
    a.q = in.q; a.kp = in.kp; a.vp = in.vp; a.res = res; a.wo = (const uint4*)H.out_w; a.bo = L.out_b; a.n1_w = L.n1_w; a.n1_b = L.n1_b;
    a.lengths = lengths; a.S = S; a.n_prefix = n_prefix; a.q_rows = q_rows; a.spv = spv; a.scale = 1.0f / 16.0f; a.eps = 1e-5f;
    a.w1 = (const uint4*)H.l1_w; a.b1 = L.l1_b; a.w2 = (const uint4*)H.l2_w; a.b2 = L.l2_b; a.B = B;
    if (cvec != nullptr) { a.cvec = cvec; a.cvec_ld = cvec_ld; a.lnc_w = L.n2_w; a.lnc_b = L.n2_b; a.n2_w = L.n3_w; a.n2_b = L.n3_b; }
    else { a.n2_w = L.n2_w; a.n2_b = L.n2_b; }
    return a;
}
static void tail_qkv(LayerHArgs& a, const SeemeXfLayer& Ln, const SeemeXfLayerH& Hn, const KvSet& out) {
    a.qkv_w = (const uint4*)Hn.in_w; a.qkv_b = Ln.in_b; a.q_out = out.q; a.kp_out = out.kp; a.vp_out = out.vp;
}
static void tail_skip(LayerHArgs& a, const uint16_t* wp, const float* b, const float* src, float* xnext) {
    a.skip_w = (const uint4*)wp; a.skip_b = b; a.skip_src = src; a.xnext = xnext;
}

int seeme_vae_encode_h16(const SeemeVaeWeights* w, const float* features, const int32_t* lengths, int B, int T, float* mu,
                         void* workspace, hipStream_t st) {
    const SeemeVaeWeightsH* H = w->h16;
    const int S = T + 2, F = w->nfeats;
    WsH ws = carve_h(workspace, B, S);
    if (!vae_fused_enabled() || ws.spv > 2 * S) return vae_encode_h16_unfused(w, features, lengths, B, T, mu, workspace, st);
    const SeemeSkipStack& E = w->enc;
    KvSet kv[2];
    kv_sets(ws, B, kv);
    int rc;
    {
        ProHArgs p{};
        p.mode = 1; p.B = B; p.S = S; p.T = T; p.F = F; p.spv = ws.spv; p.features = features; p.emb_w = (const uint4*)H->emb_w;
        p.emb_b = w->emb_b; p.token = w->token; p.pe = w->pe_enc; p.x = ws.x;
        p.qkv_w = (const uint4*)H->enc[0].in_w; p.qkv_b = E.layer[0].in_b; p.q_out = kv[0].q; p.kp_out = kv[0].kp; p.vp_out = kv[0].vp;
        if ((rc = launch_pro_h(p, st))) return rc;
    }
    LayerHArgs a = layer_args(E.layer[0], H->enc[0], kv[0], ws.x, lengths, B, S, ws.spv, 2, S, nullptr, 0);
    a.out = ws.sk0; tail_qkv(a, E.layer[1], H->enc[1], kv[1]);
    if ((rc = launch_layer_h(a, st))) return rc;
    a = layer_args(E.layer[1], H->enc[1], kv[1], ws.sk0, lengths, B, S, ws.spv, 2, S, nullptr, 0);
    a.out = ws.sk1; tail_qkv(a, E.layer[2], H->enc[2], kv[0]);
    if ((rc = launch_layer_h(a, st))) return rc;
    a = layer_args(E.layer[2], H->enc[2], kv[0], ws.sk1, lengths, B, S, ws.spv, 2, S, nullptr, 0);
    tail_skip(a, H->enc_skip[0], E.skip_b[0], ws.sk1, ws.x); tail_qkv(a, E.layer[3], H->enc[3], kv[1]);
    if ((rc = launch_layer_h(a, st))) return rc;
    a = layer_args(E.layer[3], H->enc[3], kv[1], ws.x, lengths, B, S, ws.spv, 2, S, nullptr, 0);
    tail_skip(a, H->enc_skip[1], E.skip_b[1], ws.sk0, ws.x); tail_qkv(a, E.layer[4], H->enc[4], kv[0]);
    if ((rc = launch_layer_h(a, st))) return rc;
    a = layer_args(E.layer[4], H->enc[4], kv[0], ws.x, lengths, B, S, ws.spv, 2, 2, nullptr, 0);
    a.out = mu; a.out_mode = 1; a.fin_w = E.norm_w; a.fin_b = E.norm_b;
    return launch_layer_h(a, st);
}

int seeme_vae_decode_h16(const SeemeVaeWeights* w, const float* z, const int32_t* lengths, int B, int T, float* feats,
                         void* workspace, hipStream_t st) {
    const SeemeVaeWeightsH* H = w->h16;
    const int S = T, F = w->nfeats;
    WsH ws = carve_h(workspace, B, S + 2);
    ws.spv = (S + 31) & ~31;
    if (!vae_fused_enabled() || ws.spv > 2 * S || F > 256) return vae_decode_h16_unfused(w, z, lengths, B, T, feats, workspace, st);
    const SeemeSkipStack& Dk = w->dec;
    KvSet kv[2];
    kv_sets(ws, B, kv);
    int rc;
    const int CL = SEEME_NLAYERS * 256;
    if ((rc = lin_h(st, z, 256, H->ca_fold_w, 256, CL, w->ca_fold_b, ws.cvec, CL, B))) return rc;
    {
        ProHArgs p{};
        // the decoder's first-layer rows and their q | K | V do not depend on the sample: ONE sequence's worth, read by every batch member
        p.mode = 2; p.B = 1; p.S = S; p.T = T; p.F = F; p.spv = ws.spv; p.pe = w->pe_dec; p.x = ws.x;
        p.qkv_w = (const uint4*)H->dec[0].in_w; p.qkv_b = Dk.layer[0].in_b; p.q_out = kv[0].q; p.kp_out = kv[0].kp; p.vp_out = kv[0].vp;
        if ((rc = launch_pro_h(p, st))) return rc;
    }
    LayerHArgs a = layer_args(Dk.layer[0], H->dec[0], kv[0], ws.x, lengths, B, S, ws.spv, 0, S, ws.cvec + 0 * 256, CL);
    a.in_shared = 1;
    a.out = ws.sk0; tail_qkv(a, Dk.layer[1], H->dec[1], kv[1]);
    if ((rc = launch_layer_h(a, st))) return rc;
    a = layer_args(Dk.layer[1], H->dec[1], kv[1], ws.sk0, lengths, B, S, ws.spv, 0, S, ws.cvec + 1 * 256, CL);
    a.out = ws.sk1; tail_qkv(a, Dk.layer[2], H->dec[2], kv[0]);
    if ((rc = launch_layer_h(a, st))) return rc;
    a = layer_args(Dk.layer[2], H->dec[2], kv[0], ws.sk1, lengths, B, S, ws.spv, 0, S, ws.cvec + 2 * 256, CL);
    tail_skip(a, H->dec_skip[0], Dk.skip_b[0], ws.sk1, ws.x); tail_qkv(a, Dk.layer[3], H->dec[3], kv[1]);
    if ((rc = launch_layer_h(a, st))) return rc;
    a = layer_args(Dk.layer[3], H->dec[3], kv[1], ws.x, lengths, B, S, ws.spv, 0, S, ws.cvec + 3 * 256, CL);
    tail_skip(a, H->dec_skip[1], Dk.skip_b[1], ws.sk0, ws.x); tail_qkv(a, Dk.layer[4], H->dec[4], kv[0]);
    if ((rc = launch_layer_h(a, st))) return rc;
    a = layer_args(Dk.layer[4], H->dec[4], kv[0], ws.x, lengths, B, S, ws.spv, 0, S, ws.cvec + 4 * 256, CL);
    a.fin_w = Dk.norm_w; a.fin_b = Dk.norm_b; a.proj_w = (const uint4*)H->fin_w; a.proj_b = w->fin_b; a.feats = feats; a.F = F;
    return launch_layer_h(a, st);
}
