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
#include <atomic>
#include "common.hpp"
#include "api_util.hpp"
#include "den_layout.h"
#include "lds_ring.hpp"
#include "den_train.h"
#include <utility>

#define DEN_THREADS 512
#define SEEME_DEN_PAIR_ABOVE 256   // batches above this run two independent samples per workgroup
#define DEN_MAXTOK 6   // 1 latent + N<=4 condition tokens + 1 time token
#define DEN_R 4        // register ring: chunks of the weight stream in flight per lane
#define DEN_CH 8       // 16-B vectors per lane per chunk (ring = 4 x 8 x 4 = 128 VGPRs)

// ------------------------------------------------------------------ weight element types
// fp32 weights run on the vector ALU (exact fp32 products).  16-bit weights run on the matrix cores: the
// weight vector a lane streams IS its MFMA operand (16 outputs x 32 k per wave-load), and the fp32 input
// vector enters as PARTS 16-bit rows (hi + lo [+ mid]) whose products are re-added in fp32, so the input is
// not rounded to 16 bits -- only the weights are.
struct WF32 { typedef float T; static constexpr int KV = 4; static constexpr bool HALF = false, MFMA = false; static constexpr int PARTS = 1; };
struct WBF16 { typedef uint16_t T; static constexpr int KV = 8; static constexpr bool HALF = false, MFMA = true; static constexpr int PARTS = 3; };
struct WF16 { typedef uint16_t T; static constexpr int KV = 8; static constexpr bool HALF = true, MFMA = true; static constexpr int PARTS = 2; };
#define DEN_F16_LO_SCALE 2048.f   // fp16 input split: lo = (x - hi) * 2^11 stays in the normal range

typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Ring { u32x4 r[DEN_R][DEN_CH]; };

#define FF_SA 1024   // sa_block feed-forward, hard-coded in the reference (mdiff_transformer.py:279)
#define FF_D 128     // ffn_dim (configs/modules/denoiser.yaml:5)
#define XB_LD 1024   // row stride of the GEMV input buffer

// ------------------------------------------------------------------ the static weight-stream program
// GEMV ids = index of the matrix offset inside DenLayerOff (den_layout.h).
enum { G_SKIP = 0, G_INP, G_OUTP, G_L1, G_L2, G_CAQ, G_CAO, G_F1, G_F2, G_FO, G_NUM,
       // transposed GEMVs of the backward chain (dX = W^T dY): K = out-features, N = in-features of the forward matrix
       GB_FO = G_NUM, GB_F2, GB_F1, GB_CAO, GB_CAQ, GB_L2, GB_L1, GB_OUTP, GB_INP, GB_SKIP, G_ALL };
constexpr int den_gK(int g) {
    constexpr int k[G_ALL] = {512, 256, 256, 256, FF_SA, 256, 256, 256, FF_D, 256,
                              256, 256, FF_D, 256, 256, 256, FF_SA, 256, 768, 256};
    return k[g];
}
constexpr int den_gN(int g) {
    constexpr int n[G_ALL] = {256, 768, 256, FF_SA, 256, 256, 256, FF_D, 256, 256,
                              256, FF_D, 256, 256, 256, FF_SA, 256, 256, 256, 512};
    return n[g];
}

// GEMV work split over the 512 threads.  A PyTorch [N,K] matrix is stored as [K/KV][N] 16-B vectors.  An
// "item" = (k-slice ks, output pair n0 / n0 + N/2); thread tid owns items tid, tid+512, ...  Both outputs of
// an item read the same input values, so every LDS read of x feeds two weight vectors.  An item is streamed
// in chunks of 4 k-vectors x 2 outputs = DEN_CH coalesced 16-B loads per lane, 64 KiB per workgroup.
constexpr int den_gcd(int a, int b) { return b == 0 ? a : den_gcd(b, a % b); }
template <typename WT, int G>
struct GS {
    static constexpr int K = den_gK(G), N = den_gN(G);
    static constexpr int G0 = N / 2;                                                // output pairs
    static constexpr int KS = DEN_THREADS / den_gcd(G0, DEN_THREADS);               // k-slices
    static constexpr int IT = G0 / den_gcd(G0, DEN_THREADS);                        // items per thread
    static constexpr int NQ = K / WT::KV / KS;                                      // k-vectors per item
    static constexpr int CPI = NQ / 4;                                              // chunks per item
    static constexpr int TOT = IT * CPI;                                            // chunks per thread
    static_assert(G0 * KS == DEN_THREADS * IT && NQ * KS * WT::KV == K && NQ % 4 == 0 && TOT >= 1, "unsupported GEMV shape");
    // matrix-core mapping (16-bit weights): wave w owns the TW output tiles (16 outputs each) w*TW .. w*TW+TW-1
    // for ALL k; its stream of 1-KiB wave-loads is k-block major
    // in groups of TG tiles (accumulators live at a time): wave-load i = (group i / (TG*KB), k-block, tile in group)
    static constexpr int TW = N / 16 / (DEN_THREADS / 64), KB = K / 32;
    static constexpr int TG = (TW % 4 == 0) ? 4 : ((TW % 3 == 0) ? 3 : ((TW % 2 == 0) ? 2 : 1));
    static constexpr int GL = TG * KB;                                              // wave-loads per group
    static_assert(!WT::MFMA || (TW * 16 * (DEN_THREADS / 64) == N && KB * 32 == K && TW * KB == TOT * DEN_CH && GL % DEN_CH == 0),
                  "unsupported MFMA GEMV shape");
    __device__ static __forceinline__ void item(int tid, int it, int& ks, int& n0) {
        const int idx = tid + it * DEN_THREADS;
        ks = idx / G0; n0 = idx - ks * G0;
    }
};

// Kernel variants (template parameter V):
//   V_CAQ  -- more than one condition token: the linear cross-attention needs its query GEMV and proj_out GEMV.
//             With ONE token the key softmax over tokens is exactly 1 and the query softmax sums to 1 per head, so
//             q (k^T v) = v whatever the query is (mdiff_transformer.py:231-237): the ca_block's contribution no
//             longer depends on x and comes from a precomputed table (seeme_denoiser_ca_tables).
//   V_FOLD -- one attention head: out_proj is folded into the value projection (W_o W_v, softmax weights sum to
//             1 so the bias folds too); in_proj then yields q | k | W_o v and the out_proj GEMV disappears.
#define V_CAQ 1
#define V_FOLD 2
// One layer streams P0 = inp, [outp,] l1, l2, [caq, cao,] f1, f2, fo (chunk indices 0 .. SUM-1, padded with
// no-op chunks to a multiple of the ring size so that chunk t always lives in ring slot t % DEN_R);
// layers 3 and 4 stream the skip linear first (its chunk count is a multiple of DEN_R too).
template <typename WT, int V>
struct Prog {
    static constexpr bool CAQ = (V & V_CAQ) != 0, FOLD = (V & V_FOLD) != 0;
    static constexpr int NG = 6 + (CAQ ? 2 : 0) + (FOLD ? 0 : 1);
    static constexpr int gid(int i) {
        constexpr int all[9] = {G_INP, G_OUTP, G_L1, G_L2, G_CAQ, G_CAO, G_F1, G_F2, G_FO};
        int k = 0;
        for (int j = 0; j < 9; ++j) {
            const int g = all[j];
            if ((g == G_OUTP && FOLD) || ((g == G_CAQ || g == G_CAO) && !CAQ)) continue;
            if (k == i) return g;
            ++k;
        }
        return G_FO;
    }
    static constexpr int tot(int g) {
        switch (g) {
            case G_SKIP: return GS<WT, G_SKIP>::TOT; case G_INP: return GS<WT, G_INP>::TOT; case G_OUTP: return GS<WT, G_OUTP>::TOT;
            case G_L1: return GS<WT, G_L1>::TOT; case G_L2: return GS<WT, G_L2>::TOT; case G_CAQ: return GS<WT, G_CAQ>::TOT;
            case G_CAO: return GS<WT, G_CAO>::TOT; case G_F1: return GS<WT, G_F1>::TOT; case G_F2: return GS<WT, G_F2>::TOT;
            default: return GS<WT, G_FO>::TOT;
        }
    }
    // chunk numbering: GEMV i starts at start(i); the PAD no-op chunks that round the program up to a multiple of
    // the ring size sit right behind in_proj (inside a long, stream-bound stage) -- at the end of the layer they
    // would take the re-fill slots of the short FFN stages and leave the fill path idle there
    static constexpr int sum_real() { int s = 0; for (int j = 0; j < NG; ++j) s += tot(gid(j)); return s; }
    static constexpr int T = (sum_real() + DEN_R - 1) / DEN_R * DEN_R;
    static constexpr int PAD = T - sum_real();
    static constexpr int PAD0 = tot(G_INP);                                          // first pad chunk
    static constexpr bool is_pad(int t) { return t >= PAD0 && t < PAD0 + PAD; }
    static constexpr int start(int i) { int s = (i >= 1 ? PAD : 0); for (int j = 0; j < i; ++j) s += tot(gid(j)); return s; }
    static constexpr int TS = tot(G_SKIP);
    static constexpr int find(int t) { int i = 0; while (i + 1 < NG && start(i + 1) <= t) ++i; return i; }
    static constexpr int pos(int g) { int i = 0; while (gid(i) != g) ++i; return i; }   // position of GEMV g in P0
    static_assert(gid(0) == G_INP, "in_proj opens the layer program");
    static_assert(TS % DEN_R == 0 && TS >= DEN_R, "skip linear must fill whole ring turns");
    static_assert(tot(G_INP) >= DEN_R, "the first GEMV of a layer must cover the cross-layer prefetch");
};

#ifdef DEN_DBG_TIMES
// debug build only: cycle stamps of workgroup 0 / thread 0 at every barrier of one step (scripts/den_times.py)
__device__ unsigned long long den_dbg_times[512];
__device__ __forceinline__ void den_dbg(int mode) {   // 0: stamp, 1: arm (reset), 2: disarm, 3: dump
    __shared__ unsigned long long buf[512];
    __shared__ int cnt, armed;
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    if (mode == 1) { cnt = 0; armed = 1; }
    else if (mode == 2) { armed = 0; }
    else if (mode == 0) { if (armed && cnt < 511) { buf[cnt] = __builtin_readcyclecounter(); cnt = cnt + 1; } }
    else {   // (a launch of fewer than 3 steps never armed: cnt is then whatever the LDS held)
        const int n = (armed == 0 && cnt >= 0 && cnt <= 511) ? cnt : 0;
        for (int i = 0; i < n; ++i) den_dbg_times[i + 1] = buf[i];
        den_dbg_times[0] = (unsigned long long)n;
    }
}
#define DEN_DBG(m) den_dbg(m)
#else
#define DEN_DBG(m) do {} while (0)
#endif

// ---- issue: chunk C of matrix G (base = byte offset of the matrix in the image) -> ring slot
template <typename WT, int G, int C>
__device__ __forceinline__ void issue_mat(u32x4 (&slot)[DEN_CH], int tid, __amdgpu_buffer_rsrc_t rsrc, unsigned mat_bytes) {
#ifdef DEN_DBG_NOLOAD   // timing probe: arithmetic and epilogues without the weight stream
    return;
#endif
    typedef GS<WT, G> S;
    if constexpr (WT::MFMA) {
        const unsigned voff = (unsigned)(tid >> 6) * (unsigned)(S::TW * S::KB * 1024) + (unsigned)(tid & 63) * 16u;
        const unsigned soff = mat_bytes + (unsigned)(C * DEN_CH) * 1024u;
#pragma unroll
        for (int i = 0; i < DEN_CH; ++i) slot[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff + (unsigned)(i * 1024), 0);
        return;
    }
    constexpr int it = C / S::CPI, cc = C % S::CPI;
    int ks, n0;
    S::item(tid, it, ks, n0);
    const unsigned voff = (unsigned)(ks * S::NQ * S::N + n0) * 16u;
    const unsigned soff = mat_bytes + (unsigned)(cc * 4) * S::N * 16u;
#pragma unroll
    for (int i = 0; i < DEN_CH; ++i)   // vector i = k-vector i/2, output half i%2
        slot[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff + (unsigned)(((i >> 1) * S::N + (i & 1) * S::G0) * 16), 0);
}
// Byte offsets of one layer's matrices inside the packed image, plus the first matrix of the NEXT layer's
// stream.  Read once per layer into scalar registers: the program-order pins below are memory clobbers, so
// anything still addressed through the layout table would be re-loaded (a scalar-cache round trip) per chunk.
struct MatOffs { unsigned m[G_NUM]; unsigned next; };
template <typename WT>
__device__ __forceinline__ MatOffs load_mat_offs(const DenLayerOff* __restrict__ L, const DenLayerOff* __restrict__ Ln, bool nskip) {
    MatOffs o;
#pragma unroll
    for (int g = 0; g < G_NUM; ++g) o.m[g] = (unsigned)(reinterpret_cast<const int64_t*>(L)[g] * (long long)sizeof(typename WT::T));
    o.next = (unsigned)((nskip ? Ln->skip : Ln->inp) * (long long)sizeof(typename WT::T));
    return o;
}

// chunk T of P0 of the current layer
template <typename WT, int V, int T>
__device__ __forceinline__ void issue_p0(Ring& ring, int tid, __amdgpu_buffer_rsrc_t rsrc, const MatOffs& mo) {
    typedef Prog<WT, V> P;
    static_assert(T >= 0 && T < P::T && !P::is_pad(T), "chunk outside the layer program");
    constexpr int i = P::find(T), g = P::gid(i), c = T - P::start(i);
    issue_mat<WT, g, c>(ring.r[T % DEN_R], tid, rsrc, mo.m[g]);
}
// chunk T counted from the start of the current layer's P0; T >= P::T addresses the next layer, whose stream
// starts with its skip linear when nskip is set and with in_proj otherwise (mo.next either way)
template <typename WT, int V, int T>
__device__ __forceinline__ void issue_rel(Ring& ring, int tid, __amdgpu_buffer_rsrc_t rsrc, const MatOffs& mo, bool nskip) {
    typedef Prog<WT, V> P;
    if constexpr (T < P::T) {
        if constexpr (!P::is_pad(T)) issue_p0<WT, V, T>(ring, tid, rsrc, mo);   // (padding chunk: nothing to load)
    } else {
        constexpr int j = T - P::T;
        static_assert(j < DEN_R && j < GS<WT, G_INP>::TOT && j < P::TS, "prefetch reaches too far into the next layer");
        if (nskip) issue_mat<WT, G_SKIP, j>(ring.r[j % DEN_R], tid, rsrc, mo.next);
        else issue_mat<WT, G_INP, j>(ring.r[j % DEN_R], tid, rsrc, mo.next);
    }
}

__device__ __forceinline__ float bf_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

// acc[s][j] += W-chunk(output half j) . x[s][k0 ...]   (x in LDS, broadcast reads; packed fp32 FMAs)
template <typename WT, int MS>
__device__ __forceinline__ void consume(const u32x4 (&b)[DEN_CH], const float* __restrict__ x, int ldx, int k0, f2 (&acc)[MS][2][2]) {
#ifdef DEN_DBG_NOFMA   // timing probe: keep the weight stream, drop the arithmetic (results are garbage)
#pragma unroll
    for (int i = 0; i < DEN_CH; ++i) acc[0][0][0].x += __uint_as_float(b[i].x ^ b[i].y ^ b[i].z ^ b[i].w);
    return;
#endif
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int s = 0; s < MS; ++s) {
            const float* xp = x + s * ldx + k0 + WT::KV * kk;
            const float4 x0 = *reinterpret_cast<const float4*>(xp);
            float4 x1 = x0;
            if constexpr (WT::KV == 8) x1 = *reinterpret_cast<const float4*>(xp + 4);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const u32x4 u = b[2 * kk + j];
                f2 (&a)[2] = acc[s][j];
                if constexpr (WT::KV == 4) {
                    a[0] += f2{__uint_as_float(u.x), __uint_as_float(u.y)} * f2{x0.x, x0.y};
                    a[1] += f2{__uint_as_float(u.z), __uint_as_float(u.w)} * f2{x0.z, x0.w};
                } else if constexpr (WT::HALF) {
                    // fp16 weights: v_fma_mix_f32 takes the half operand directly (no unpack instructions)
                    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                    const unsigned ux = u.x, uy = u.y, uz = u.z, uw = u.w;   // (bit_cast straight from a vector element picks element 0)
                    const h2 w0 = __builtin_bit_cast(h2, ux), w1 = __builtin_bit_cast(h2, uy), w2 = __builtin_bit_cast(h2, uz), w3 = __builtin_bit_cast(h2, uw);
                    a[0].x = fmaf((float)w0.x, x0.x, a[0].x); a[0].y = fmaf((float)w0.y, x0.y, a[0].y);
                    a[1].x = fmaf((float)w1.x, x0.z, a[1].x); a[1].y = fmaf((float)w1.y, x0.w, a[1].y);
                    a[0].x = fmaf((float)w2.x, x1.x, a[0].x); a[0].y = fmaf((float)w2.y, x1.y, a[0].y);
                    a[1].x = fmaf((float)w3.x, x1.z, a[1].x); a[1].y = fmaf((float)w3.y, x1.w, a[1].y);
                } else {
                    a[0] += f2{bf_lo(u.x), bf_hi(u.x)} * f2{x0.x, x0.y};
                    a[1] += f2{bf_lo(u.y), bf_hi(u.y)} * f2{x0.z, x0.w};
                    a[0] += f2{bf_lo(u.z), bf_hi(u.z)} * f2{x1.x, x1.y};
                    a[1] += f2{bf_lo(u.w), bf_hi(u.w)} * f2{x1.z, x1.w};
                }
            }
        }
    }
}

// Where a GEMV reads its input: fp32 path = the fp32 vector(s) in LDS (broadcast reads); matrix-core path = this
// lane's row of the 16-bit input fragments, [k-block][k-group 4][row 4*MS][8 halves], row 4s+p = part p of
// sample s.  Lanes whose operand row (lane % 16) has no part simply re-read another row: operand rows are
// independent in a matrix product, so they only fill accumulator rows nobody reads.  The k-block stride is a
// compile-time constant (an immediate offset of the LDS read).
struct XIn { const float* xf; const char* fbase; int foff; };   // fragment row of this lane = fbase (LDS) + foff bytes
#define XFRAG_STRIDE(MS) (4 * 4 * (MS) * 16)

// Accumulators of one GEMV.  fp32 path: [sample][output half][2] packed pairs.  Matrix-core path: one 16x16
// tile per owned output tile; lanes 16s .. 16s+15 end up with the PARTS partial products of sample s for the
// tile's 16 outputs in elements 0 .. PARTS-1.
template <typename WT, int MS, int G>
struct Acc {
    f2 v[WT::MFMA ? 1 : MS][2][2];
    f32x4 m[WT::MFMA ? GS<WT, G>::TG : 1];
    __device__ __forceinline__ void zero() {
        if constexpr (WT::MFMA) {
#pragma unroll
            for (int t = 0; t < GS<WT, G>::TG; ++t) m[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
#pragma unroll
            for (int s = 0; s < MS; ++s)
#pragma unroll
                for (int j = 0; j < 2; ++j) { v[s][j][0] = f2{0.f, 0.f}; v[s][j][1] = f2{0.f, 0.f}; }
        }
    }
    // Pin the program order: the accumulators pass through a side-effecting asm with a memory clobber, so the
    // arithmetic that read a ring slot stays above and the loads that re-fill the slot stay below it (otherwise
    // the instruction selector hoists the re-fill into fresh registers and the ring silently doubles).
    __device__ __forceinline__ void pin() {
        if constexpr (WT::MFMA) {
#pragma unroll
            for (int t = 0; t < GS<WT, G>::TG; ++t) asm volatile("" : "+v"(m[t]) :: "memory");
        } else {
#pragma unroll
            for (int s = 0; s < MS; ++s)
                asm volatile("" : "+v"(v[s][0][0]), "+v"(v[s][0][1]), "+v"(v[s][1][0]), "+v"(v[s][1][1]) :: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
    }
};

// One chunk step of a GEMV of the layer program.  REL0 = index of the GEMV's first chunk relative to the
// layer's P0 (negative for the skip linear), C = local chunk, NC = chunks incl. padding.  Consuming chunk t
// frees ring slot t % DEN_R, which is re-filled with chunk t + DEN_R -- at once, except for the last BURST
// chunks of the GEMV, whose re-fill is withheld until the epilogue runs (gemv_stream).
template <typename WT, int V, int G, int REL0, int C>
__device__ __forceinline__ void refill(Ring& ring, int tid, __amdgpu_buffer_rsrc_t rsrc, const MatOffs& mo, bool nskip) {
    constexpr int TGT = REL0 + C + DEN_R;
    if constexpr (TGT < 0) issue_mat<WT, G_SKIP, TGT + Prog<WT, V>::TS>(ring.r[((TGT % DEN_R) + DEN_R) % DEN_R], tid, rsrc, mo.m[G_SKIP]);
    else issue_rel<WT, V, TGT>(ring, tid, rsrc, mo, nskip);
}
// vector-ALU consume of chunk C of GEMV G: accumulate, and publish the item's partial sums at its last chunk
template <typename WT, int MS, int G, int C>
__device__ __forceinline__ void consume_chunk_valu(const u32x4 (&slot)[DEN_CH], int tid, const float* __restrict__ xf,
                                                   float* __restrict__ part, Acc<WT, MS, G>& acc) {
    typedef GS<WT, G> S;
    constexpr int it = C / S::CPI, cc = C % S::CPI;
    int ks, n0;
    S::item(tid, it, ks, n0);
    if constexpr (cc == 0) acc.zero();
    consume<WT, MS>(slot, xf, XB_LD, (ks * S::NQ + cc * 4) * WT::KV, acc.v);
    if constexpr (cc == S::CPI - 1) {
#pragma unroll
        for (int s = 0; s < MS; ++s)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                part[(ks * MS + s) * S::N + n0 + j * S::G0] = (acc.v[s][j][0].x + acc.v[s][j][0].y) + (acc.v[s][j][1].x + acc.v[s][j][1].y);
    }
}
template <typename WT, int V, int MS, int G, int REL0, int NC, int BURST, int C>
__device__ __forceinline__ void gemv_chunk(Ring& ring, int tid, __amdgpu_buffer_rsrc_t rsrc, const MatOffs& mo, bool nskip,
                                           const XIn& x, float* __restrict__ part, Acc<WT, MS, G>& acc) {
    typedef GS<WT, G> S;
    constexpr int SLOT = ((REL0 + C) % DEN_R + DEN_R) % DEN_R;
    if constexpr (C < S::TOT && WT::MFMA) {
        // 8 wave-loads = 8 (k-block, tile) operands; the input fragment changes once per k-block
        uint4 a4 = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int j = 0; j < DEN_CH; ++j) {
            const int i = C * DEN_CH + j, grp = i / S::GL, ii = i % S::GL, kb = ii / S::TG, t = ii % S::TG;
            if (ii == 0) acc.zero();
            if (j == 0 || t == 0) a4 = *reinterpret_cast<const uint4*>(x.fbase + x.foff + kb * XFRAG_STRIDE(MS));
#ifndef DEN_DBG_NOFMA
            if constexpr (WT::HALF) acc.m[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, a4), __builtin_bit_cast(h16x8, ring.r[SLOT][j]), acc.m[t], 0, 0, 0);
            else acc.m[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a4), __builtin_bit_cast(bf16x8, ring.r[SLOT][j]), acc.m[t], 0, 0, 0);
#else
            acc.m[t].x += __uint_as_float(ring.r[SLOT][j].x ^ ring.r[SLOT][j].y ^ ring.r[SLOT][j].z ^ ring.r[SLOT][j].w ^ a4.x);
#endif
            if (ii == S::GL - 1) {
                // lanes 16s .. 16s+15: outputs (wave*TW + grp*TG + t)*16 + lane%16 of sample s = sum of the input parts
                const int lane = tid & 63, sl = lane >> 4;
                if (sl < MS) {
#pragma unroll
                    for (int t2 = 0; t2 < S::TG; ++t2) {
                        float val;
                        if constexpr (WT::HALF) val = fmaf(acc.m[t2].y, 1.f / DEN_F16_LO_SCALE, acc.m[t2].x);
                        else val = acc.m[t2].x + (acc.m[t2].y + acc.m[t2].z);
                        part[sl * S::N + ((tid >> 6) * S::TW + grp * S::TG + t2) * 16 + (lane & 15)] = val;
                    }
                }
            }
        }
    } else if constexpr (C < S::TOT) {
        consume_chunk_valu<WT, MS, G, C>(ring.r[SLOT], tid, x.xf, part, acc);
    }
    acc.pin();
    if constexpr (C < NC - BURST) {
        refill<WT, V, G, REL0, C>(ring, tid, rsrc, mo, nskip);
        acc.pin();
    }
}
template <typename WT, int V, int MS, int G, int REL0, int NC, int BURST, int... Cs>
__device__ __forceinline__ void gemv_chunks(Ring& ring, int tid, __amdgpu_buffer_rsrc_t rsrc, const MatOffs& mo, bool nskip,
                                            const XIn& x, float* __restrict__ part, Acc<WT, MS, G>& acc,
                                            std::integer_sequence<int, Cs...>) {
    (gemv_chunk<WT, V, MS, G, REL0, NC, BURST, Cs>(ring, tid, rsrc, mo, nskip, x, part, acc), ...);
}

// Streaming half of GEMV G: every wave consumes its chunks (ring slots were requested earlier), publishes the
// partial sums (barrier A) and then requests the withheld chunks, which keeps the L1 fill path busy while the
// epilogue wave(s) turn the partial sums into the next GEMV's input.  The caller runs the epilogue and then
// barrier B.  On entry and exit DEN_R chunks are in flight ahead of the consumer.
template <typename WT, int V, int MS, int G, bool LAST>
__device__ __forceinline__ void gemv_stream(Ring& ring, int tid, __amdgpu_buffer_rsrc_t rsrc, const MatOffs& mo, bool nskip,
                                            const XIn& xin_, float* __restrict__ part, bool epi) {
    typedef Prog<WT, V> P;
    constexpr int REL0 = (G == G_SKIP) ? -P::TS : P::start(P::pos(G));
    constexpr int NC = GS<WT, G>::TOT + (G == G_INP ? P::PAD : 0);   // in_proj also "consumes" the padding chunks
    (void)LAST;
    constexpr int BURST = NC < 2 ? NC : 2;
    asm volatile("" : "+v"(tid));   // addresses are recomputed per GEMV, not kept live across the layer body
    XIn x = xin_;
    asm volatile("" : "+v"(x.foff));   // (an int, so the LDS address space of fbase stays visible to the compiler)
    Acc<WT, MS, G> acc;
    acc.zero();
    gemv_chunks<WT, V, MS, G, REL0, NC, BURST>(ring, tid, rsrc, mo, nskip, x, part, acc, std::make_integer_sequence<int, NC>{});
    __syncthreads();                                                  // barrier A: partial sums visible
    DEN_DBG(0);
#ifdef DEN_SLEEP
    if (!epi) __builtin_amdgcn_s_sleep(DEN_SLEEP);                    // let the epilogue wave's re-fills enter the fill queue first
#endif
    refill<WT, V, G, REL0, NC - BURST>(ring, tid, rsrc, mo, nskip);
    if constexpr (BURST == 2) refill<WT, V, G, REL0, NC - 1>(ring, tid, rsrc, mo, nskip);
    acc.pin();
}

// ------------------------------------------------------------------ epilogue math
// The chain of dependent epilogues runs on ONE wave per sample, so its instruction count is latency of the
// whole loop.  Hardware transcendental forms (1 ulp): v_exp_f32, v_rcp_f32, v_rsq_f32; erf by Abramowitz &
// Stegun 7.1.26 (|error| <= 1.5e-7) -- all far inside the 1e-4 parity gate.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fast_silu(float x) { return x * fast_rcp(1.f + fast_exp(-x)); }
__device__ __forceinline__ float fast_gelu(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = fast_rcp(fmaf(0.3275911f, z, 1.f));
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    const float erfa = 1.f - poly * fast_exp(-z * z);                   // erf(|x| / sqrt 2)
    return 0.5f * x * (1.f + copysignf(erfa, x));
}

// ------------------------------------------------------------------ wave-local vector algebra
// The epilogue wave of a sample owns the WHOLE 256-vector, 4 consecutive dims per lane (dims 4*lane ..
// 4*lane+3), so LayerNorm / softmax / dot products are wave reductions (DPP butterflies).
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
    const float rs = fast_rsq(wave_sum(f4_dot(c, c)) * (1.f / 256.f) + 1e-5f);
    const float4 wv = ld4(w + 4 * lane), bv = ld4(b + 4 * lane);
    return make_float4(c.x * rs * wv.x + bv.x, c.y * rs * wv.y + bv.y, c.z * rs * wv.z + bv.z, c.w * rs * wv.w + bv.w);
}
// same, also returning the normalised vector and the reciprocal standard deviation (saved for the backward pass)
__device__ __forceinline__ float4 wave_ln_stats(float4 v, const float* __restrict__ w, const float* __restrict__ b, int lane,
                                                float4& xhat, float& rs) {
    const float mean = wave_sum(v.x + v.y + v.z + v.w) * (1.f / 256.f);
    const float4 c = make_float4(v.x - mean, v.y - mean, v.z - mean, v.w - mean);
    rs = fast_rsq(wave_sum(f4_dot(c, c)) * (1.f / 256.f) + 1e-5f);
    xhat = f4_scale(c, rs);
    const float4 wv = ld4(w + 4 * lane), bv = ld4(b + 4 * lane);
    return make_float4(xhat.x * wv.x + bv.x, xhat.y * wv.y + bv.y, xhat.z * wv.z + bv.z, xhat.w * wv.w + bv.w);
}
__device__ __forceinline__ float4 f4_silu(float4 v) { return make_float4(fast_silu(v.x), fast_silu(v.y), fast_silu(v.z), fast_silu(v.w)); }
__device__ __forceinline__ float4 f4_adaln(float4 h, float4 scl, float4 shf) {
    return make_float4(h.x * (1.f + scl.x) + shf.x, h.y * (1.f + scl.y) + shf.y, h.z * (1.f + scl.z) + shf.z, h.w * (1.f + scl.w) + shf.w);
}
// sum of the KS k-slice partials of an N-output GEMV for sample s, outputs off + 4*lane ..
template <int MS, int KS, int N>
__device__ __forceinline__ float4 part_sum(const float* __restrict__ part, int s, int off, int lane) {
    float4 v = ld4(part + s * N + off + 4 * lane);
#pragma unroll
    for (int k2 = 1; k2 < KS; ++k2) v = f4_add(v, ld4(part + (k2 * MS + s) * N + off + 4 * lane));
    return v;
}
// outputs off + 4*lane .. of GEMV G for sample s (matrix-core path: no k-slices, [sample][N])
template <typename WT, int MS, int G>
__device__ __forceinline__ float4 gemv_out(const float* __restrict__ part, int s, int off, int lane) {
    if constexpr (WT::MFMA) return ld4(part + s * GS<WT, G>::N + off + 4 * lane);
    else return part_sum<MS, GS<WT, G>::KS, GS<WT, G>::N>(part, s, off, lane);
}
template <typename WT, int MS>
__device__ __forceinline__ float4 part256(const float* __restrict__ part, int s, int lane) { return gemv_out<WT, MS, G_OUTP>(part, s, 0, lane); }

// The epilogue wave of sample s publishes input values k0 + 4*lane .. k0 + 4*lane + 3 of the next GEMV.
template <typename WT, int MS>
__device__ __forceinline__ void put_x(float* __restrict__ xb, int s, int k0, int lane, float4 v) {
    if constexpr (!WT::MFMA) {
        st4(xb + s * XB_LD + k0 + 4 * lane, v);
    } else {
        const int k = k0 + 4 * lane;
        char* dst = reinterpret_cast<char*>(xb) + ((((k >> 5) * 4 + ((k >> 3) & 3)) * (4 * MS) + 4 * s) * 16) + (k & 7) * 2;
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        typedef __bf16 b4 __attribute__((ext_vector_type(4)));
        if constexpr (WT::HALF) {
            const h4 hi = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
            const h4 lo = {(_Float16)((v.x - (float)hi.x) * DEN_F16_LO_SCALE), (_Float16)((v.y - (float)hi.y) * DEN_F16_LO_SCALE),
                           (_Float16)((v.z - (float)hi.z) * DEN_F16_LO_SCALE), (_Float16)((v.w - (float)hi.w) * DEN_F16_LO_SCALE)};
            *reinterpret_cast<h4*>(dst) = hi;
            *reinterpret_cast<h4*>(dst + 16) = lo;
        } else {
            const b4 hi = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
            const float4 r1 = make_float4(v.x - (float)hi.x, v.y - (float)hi.y, v.z - (float)hi.z, v.w - (float)hi.w);
            const b4 mid = {(__bf16)r1.x, (__bf16)r1.y, (__bf16)r1.z, (__bf16)r1.w};
            const b4 lo = {(__bf16)(r1.x - (float)mid.x), (__bf16)(r1.y - (float)mid.y), (__bf16)(r1.z - (float)mid.z), (__bf16)(r1.w - (float)mid.w)};
            *reinterpret_cast<b4*>(dst) = hi;
            *reinterpret_cast<b4*>(dst + 16) = mid;
            *reinterpret_cast<b4*>(dst + 32) = lo;
        }
    }
}

// ------------------------------------------------------------------ the persistent sampling kernel
#define VP_LAYER (256 + 768 + 256 * 3 + FF_SA + 256 * 9 + FF_D + 256 * 4)   // floats of vector params per layer
#define STG_TT 1536  // time-token K|V (512) + AdaLN rows (1024) of one layer
// floats of the GEMV input buffer: fp32 vectors, or the 16-bit fragments (8 KiB per sample)
#define XB_FLOATS(mfma, MS) ((mfma) ? (MS) * 2048 : (MS) * XB_LD)

struct DenKArgs {
    const void* wg; int wg_bytes; const float* vp;
    DenLayout lay;             // by value: offsets are read with scalar loads from the kernel-argument segment (through a
                               // device pointer the compiler issues vector loads, whose waits drain the weight ring)
    int nhead;
    SeemeSampleArgs s;
    int xcds, chains;          // workgroups are dealt to the 8 XCDs round-robin; xcds < 8 packs the `chains` working ones onto the first xcds
};

// Per-layer small operands (biases / LayerNorm params, the time token's K|V and AdaLN rows, the condition
// tokens' K|V) travel global -> LDS by LDS-DMA (no registers), one layer AHEAD of their use, into the other
// half of a double buffer: requested during layer l (in its ffn.linear1 stage) for layer l+1, 1 KiB per
// wave-instruction spread over the non-epilogue waves.  vmcnt retires in issue order, so once a wave has consumed any weight chunk it requested
// later, its copies have landed; the barriers of layer l then publish them.  (hipcc does not count the asm
// DMA: its own vmcnt(N) waits only become slightly longer, never shorter.)
// Which rows of the per-sample tables / latents the MS samples of workgroup b use.  CFG pair (A.cfg): both share
// latent b; sample 0 = unconditional branch (first half of ctab, mld.py:489), sample 1 = conditional (second half).
// Independent samples (MS = 2 without CFG, batches larger than the chip): samples MS*b + s, clamped to B - 1.
template <int MS>
__device__ __forceinline__ int den_lat_index(const SeemeSampleArgs& A, int b, int s) {
    if (MS == 1 || A.cfg) return b;
    const int i = MS * b + s;
    return i < A.B ? i : A.B - 1;
}
template <int MS>
__device__ __forceinline__ int den_cond_index(const SeemeSampleArgs& A, int b, int s) {
    if (MS == 2 && A.cfg) return s == 1 ? A.B + b : b;
    return den_lat_index<MS>(A, b, s);
}

template <int MS, bool CAQ, int W0, int PART>   // PART 0: everything, 1: the vector params, 2: the table rows
__device__ __forceinline__ void stage_dma(int wave, int lane, float* __restrict__ stg, const float* __restrict__ vpg,
                                          const DenLayerOff* __restrict__ L, const float* __restrict__ tt_row, int l,
                                          const SeemeSampleArgs& A, int b, int N, int ca_r, int ca_R) {
    const uint32_t base = lds_addr_of(stg);
    const int ncond = 31 + 4 * MS * N;
    const int total = ncond + (CAQ ? 0 : MS);
    const int c_lo = (PART == 2) ? 25 : 0, c_hi = (PART == 1) ? 25 : total;
#pragma unroll 1
    for (int c = c_lo + wave - W0; c < c_hi; c += DEN_THREADS / 64 - W0) {   // waves W0 .. 7 share the copies
        const float* src;
        int dst, lanes = 64;
        if (c < 25) {                       // VP_LAYER = 24.5 KiB of vector params
            src = vpg + L->skip_b + c * 256; dst = c * 256; if (c == 24) lanes = (VP_LAYER - 24 * 256) / 4;
        } else if (c < 27) {                // time token K|V of this layer's sa_block
            src = tt_row + l * 512 + (c - 25) * 256; dst = VP_LAYER + (c - 25) * 256;
        } else if (c < 31) {                // AdaLN scale|shift rows (ca, ffn)
            src = tt_row + 2560 + l * 1024 + (c - 27) * 256; dst = VP_LAYER + 512 + (c - 27) * 256;
        } else if (!CAQ && c >= ncond) {    // one condition token: the tabulated ca_block term of (sample, row, layer)
            const int sidx = c - ncond;
            const int bc = den_cond_index<MS>(A, b, sidx);
            src = A.catab + (((size_t)bc * ca_R + ca_r) * SEEME_DEN_NL + l) * 256; dst = VP_LAYER + STG_TT + MS * N * 1024 + sidx * 256;
        } else {                            // condition tokens: sa K|V (512) | ca key|value (512) per (sample, token)
            const int j = c - 31, sn = j >> 2, q = j & 3;
            const int s = sn / N, n = sn - s * N;
            const int bc = den_cond_index<MS>(A, b, s);
            src = A.ctab + ((size_t)bc * N + n) * SEEME_CROW + (q < 2 ? l * 512 + q * 256 : 2560 + l * 512 + (q - 2) * 256);
            dst = VP_LAYER + STG_TT + sn * 1024 + q * 256;
        }
        if (lane < lanes) lds_dma_1k(src + 4 * lane, base + (uint32_t)dst * 4u);
    }
}

template <typename WT, int MS, int V>
__global__ __launch_bounds__(DEN_THREADS) void k_den_sample(const DenKArgs ka) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef Prog<WT, V> P;
    constexpr bool CAQ = P::CAQ, FOLD = P::FOLD;
    // wave-uniform buffer descriptor of the packed weight image (raw buffer loads: VGPR offset + SGPR offset)
    const __amdgpu_buffer_rsrc_t wg = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ka.wg), 0, ka.wg_bytes, 0x00020000);
    const float* __restrict__ vp = ka.vp;
    const DenLayout* __restrict__ lay = &ka.lay;
    const SeemeSampleArgs& A = ka.s;
    const int tid0 = threadIdx.x;
    int b = blockIdx.x;
    if (ka.xcds < 8) {         // fewer XCDs: fewer L2s that each pull the whole weight image from the Infinity Cache every step
        const int x = blockIdx.x & 7;
        b = (blockIdx.x >> 3) * ka.xcds + x;
        if (x >= ka.xcds || b >= ka.chains) return;
    }
    const int N = A.N, H = ka.nhead;
    const int seg = 64 / H;                      // lanes per attention head
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int lane = tid0 & 63;
#ifdef DEN_DBG_NOEPI   // timing probe: no epilogues (results are garbage)
    const bool epi = false;
#else
    const bool epi = wave < MS;                  // wave s turns sample s's partial sums into the next input vector
#endif
    const int es = epi ? wave : 0;
    const int stg_sz = VP_LAYER + STG_TT + MS * N * 1024 + (CAQ ? 0 : MS * 256);
    const int ca_R = A.trow_per_sample ? 1 : A.steps;

    float* CONSTV = smem;                        // [768]            query_pos.pe[0], encoder.norm.{weight,bias}
    float* KEEP = CONSTV + 768;                  // [MS][3][256]     per sample: latent, skip inputs of layers 0 / 1 (epilogue wave's own)
    float* STG = KEEP + MS * 768;                   // [2][stg_sz]      per-layer operands, double-buffered
    float* XB = STG + 2 * stg_sz;                // fp32 path: [MS][XB_LD] GEMV input; matrix-core path: 16-bit input
                                                 // fragments [32 k-blocks][4][4*MS rows][8]
    float* PART = XB + XB_FLOATS(WT::MFMA, MS);  // fp32 path: [4*MS*768] partial sums (k-slice major); matrix-core: [MS][1024]
    XIn xin;
    xin.xf = XB;
    {
        const int row = (lane & 15) % (4 * MS), kg = lane >> 4;
        xin.fbase = reinterpret_cast<const char*>(XB);
        xin.foff = (kg * 4 * MS + row) * 16;
    }

    const float sa_scale = 1.f / sqrtf((float)(256 / H));   // (once per kernel)
    if (epi) __builtin_amdgcn_s_setprio(3);      // the chain of dependent epilogues is the critical path
    // training forward: every intermediate the backward kernel needs goes to A.save (den_train.h); only the unfolded,
    // query-GEMV variant is differentiable (the folds merge parameters), and only it carries the stores
    constexpr bool SAVE = CAQ && !FOLD && MS == 1;
    float* const sv0 = (SAVE && A.save != nullptr && epi) ? A.save + (size_t)den_lat_index<MS>(A, b, es) * DT_TOTAL : nullptr;
    // training-mode dropout (keep-masks drawn by the host, one block per sample): only with the saving forward
    const unsigned char* const dm0 = (SAVE && sv0 && A.drop != nullptr) ? A.drop + (size_t)den_lat_index<MS>(A, b, es) * DM_TOTAL : nullptr;
    const float dsc = A.drop_scale;
    auto drop4 = [&](float4 v, const unsigned char* m) {          // m: this lane's 4 mask bytes
        const uchar4 k = *reinterpret_cast<const uchar4*>(m);
        return make_float4(k.x ? v.x * dsc : 0.f, k.y ? v.y * dsc : 0.f, k.z ? v.z * dsc : 0.f, k.w ? v.w * dsc : 0.f);
    };
    float* const keep = KEEP + es * 768 + 4 * lane;   // [0] latent, [256] layer-0 output, [512] layer-1 output
    const int bl = den_lat_index<MS>(A, b, es);          // this epilogue wave's row of latents / noise / out
    float4 xr = ld4(A.latents + (size_t)bl * 256 + 4 * lane);
    if (epi) st4(keep, xr);

    // ---- prologue: constants, layer 0 operands, first input vector, first DEN_R chunks
    // per-step scalars (table row, scheduler coefficients) through the constant address space: scalar loads, which do
    // not queue behind (and whose waits do not drain) the vector-memory weight ring
    typedef const __attribute__((address_space(4))) int32_t* CI32;
    typedef const __attribute__((address_space(4))) float* CF32;
    const CI32 trow_c = (CI32)(uintptr_t)A.trow;
    const CF32 coef_c = (CF32)(uintptr_t)A.coef;
    int row = A.trow_per_sample ? trow_c[b] : trow_c[0];
    {
        for (int i = tid0; i < 192; i += DEN_THREADS)
            st4(CONSTV + 4 * i, i < 64 ? ld4(vp + lay->pe0 + 4 * i) : (i < 128 ? ld4(vp + lay->fnw + 4 * (i - 64)) : ld4(vp + lay->fnb + 4 * (i - 128))));
        stage_dma<MS, CAQ, 0, 0>(wave, lane, STG, vp, &lay->L[0], A.ttab + (size_t)row * SEEME_TROW, 0, A, b, N, 0, ca_R);
        if (WT::MFMA) for (int i = tid0; i < XB_FLOATS(true, MS) / 4; i += DEN_THREADS) st4(XB + 4 * i, make_float4(0.f, 0.f, 0.f, 0.f));
        wait_vmcnt0();
        __syncthreads();
        if (epi) { xr = f4_add(xr, ld4(CONSTV + 4 * lane)); put_x<WT, MS>(XB, es, 0, lane, xr); }   // mld_denoiser.py:210
    }
    Ring ring;
    {
        const MatOffs m0 = load_mat_offs<WT>(&lay->L[0], &lay->L[1], false);
        issue_p0<WT, V, 0>(ring, tid0, wg, m0); issue_p0<WT, V, 1>(ring, tid0, wg, m0);
        issue_p0<WT, V, 2>(ring, tid0, wg, m0); issue_p0<WT, V, 3>(ring, tid0, wg, m0);
        static_assert(DEN_R == 4, "prologue issues four chunks");
    }
    __syncthreads();
    int cur = 0;                                 // which half of STG holds the current layer

#pragma unroll 1
    for (int step = 0; step < A.steps; ++step) {
        if (step == 2) DEN_DBG(1);
        if (step == 3) DEN_DBG(2);
        // table row of the NEXT step (the last layer stages layer 0 of the next step)
        const int step_next = step + 1 < A.steps ? step + 1 : step;
        const int row_next = A.trow_per_sample ? row : trow_c[step_next];
#pragma unroll 1
        for (int l = 0; l < SEEME_DEN_NL; ++l) {
            const DenLayerOff* __restrict__ L = &lay->L[l];
            const int ln = (l + 1 < SEEME_DEN_NL) ? l + 1 : 0;
            const DenLayerOff* __restrict__ Ln = &lay->L[ln];
            const bool nskip = ln >= 3;
            const MatOffs mo = load_mat_offs<WT>(L, Ln, nskip);
            // Launder the thread id once per layer: every address derived from it is then recomputed inside
            // the layer body instead of being hoisted out of the loops.
            int tid = tid0;
            asm volatile("" : "+v"(tid));
            const float* VP = STG + cur * stg_sz;            // this layer's biases / LayerNorm params
            const float* TTS = VP + VP_LAYER;                // time-token K|V (512) + AdaLN rows (1024)
            const float* CTS = TTS + STG_TT;                 // [MS][N][1024] condition K|V (sa 512 | ca 512)
            const float* CT = CTS + es * N * 1024;
            float* const sv = sv0 ? sv0 + l * DT_LAYER : nullptr;
            const unsigned char* const dm = dm0 ? dm0 + l * DM_LAYER : nullptr;
            const float* CA_ADD = CTS + MS * N * 1024 + es * 256;   // (one condition token) tabulated ca_block term
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

            // ---- skip connection: Linear(cat[x, xs.pop()])  (cross_attention.py:77-79); input staged by the
            //      previous layer's last epilogue
            if (l >= 3) {
                gemv_stream<WT, V, MS, G_SKIP, false>(ring, tid, wg, mo, nskip, xin, PART, epi);
                if (epi) {
                    xr = f4_add(part256<WT, MS>(PART, es, lane), ld4(v_skip_b + 4 * lane));
                    put_x<WT, MS>(XB, es, 0, lane, xr);
                }
                __syncthreads(); DEN_DBG(0);
            }
            // ---- sa_block: post-norm encoder layer over [x, xf.., emb]; only token 0 is kept
            //      (mdiff_transformer.py:292-297); K/V of xf and emb come from the tables.
            gemv_stream<WT, V, MS, G_INP, false>(ring, tid, wg, mo, nskip, xin, PART, epi);
            if (epi) {
                // in_proj output n in [0,768): q | k | v
                const float4 q = f4_add(gemv_out<WT, MS, G_INP>(PART, es, 0, lane), ld4(v_in_b + 4 * lane));
                const float4 k0 = f4_add(gemv_out<WT, MS, G_INP>(PART, es, 256, lane), ld4(v_in_b + 256 + 4 * lane));
                const float4 v0 = f4_add(gemv_out<WT, MS, G_INP>(PART, es, 512, lane), ld4(v_in_b + 512 + 4 * lane));
                if (SAVE && sv) {
                    st4(sv + DT_X + 4 * lane, xr);
                    st4(sv + DT_QKV + 4 * lane, q); st4(sv + DT_QKV + 256 + 4 * lane, k0); st4(sv + DT_QKV + 512 + 4 * lane, v0);
                }
                float sc[DEN_MAXTOK];
                sc[0] = seg_reduce<false>(f4_dot(q, k0), seg) * sa_scale;
                float mx = sc[0];
#pragma unroll
                for (int j = 0; j < DEN_MAXTOK - 2; ++j)
                    if (j < N) { sc[1 + j] = seg_reduce<false>(f4_dot(q, ld4(CT + j * 1024 + 4 * lane)), seg) * sa_scale; mx = fmaxf(mx, sc[1 + j]); }
                // the time token is the LAST of the sequence (mdiff_transformer.py:295)
                const float st = seg_reduce<false>(f4_dot(q, ld4(TTS + 4 * lane)), seg) * sa_scale;
                mx = fmaxf(mx, st);
                float e0 = fast_exp(sc[0] - mx), et = fast_exp(st - mx), sum = e0 + et;
#pragma unroll
                for (int j = 0; j < DEN_MAXTOK - 2; ++j) if (j < N) { sc[1 + j] = fast_exp(sc[1 + j] - mx); sum += sc[1 + j]; }
                const float inv = fast_rcp(sum);
                // (dropout on the attention weights: the saved probabilities stay the un-dropped ones, the backward re-applies the mask)
                auto pk = [&](int j) { return (SAVE && dm) ? (dm[DM_P + j] ? dsc : 0.f) : 1.f; };
                float4 att = f4_scale(v0, e0 * inv * pk(0));
#pragma unroll
                for (int j = 0; j < DEN_MAXTOK - 2; ++j)
                    if (j < N) att = f4_fma(sc[1 + j] * inv * pk(1 + j), ld4(CT + j * 1024 + 256 + 4 * lane), att);
                att = f4_fma(et * inv * pk(1 + N), ld4(TTS + 256 + 4 * lane), att);
                if (SAVE && sv) {
                    st4(sv + DT_A + 4 * lane, att);
                    if (lane == 0) {
                        sv[DT_P] = e0 * inv;
#pragma unroll
                        for (int j = 0; j < DEN_MAXTOK - 2; ++j) if (j < N) sv[DT_P + 1 + j] = sc[1 + j] * inv;
                        sv[DT_P + 1 + N] = et * inv;
                    }
                }
                if constexpr (FOLD) {   // the "values" already carry out_proj: residual + norm1 right here
                    xr = wave_ln(f4_add(xr, att), v_n1w, v_n1b, lane);
                    put_x<WT, MS>(XB, es, 0, lane, xr);
                } else {
                    put_x<WT, MS>(XB, es, 0, lane, att);
                }
            }
            __syncthreads(); DEN_DBG(0);
            if constexpr (!FOLD) {
                // ---- out_proj + residual + norm1
                gemv_stream<WT, V, MS, G_OUTP, false>(ring, tid, wg, mo, nskip, xin, PART, epi);
                if (epi) {
                    float4 o = f4_add(part256<WT, MS>(PART, es, lane), ld4(v_out_b + 4 * lane));
                    if (SAVE && dm) o = drop4(o, dm + DM_1 + 4 * lane);
                    const float4 v = f4_add(xr, o);
                    float4 xh; float rs;
                    xr = wave_ln_stats(v, v_n1w, v_n1b, lane, xh, rs);
                    if (SAVE && sv) { st4(sv + DT_XH1 + 4 * lane, xh); if (lane == 0) sv[DT_RS + 0] = rs; }
                    put_x<WT, MS>(XB, es, 0, lane, xr);
                }
                __syncthreads(); DEN_DBG(0);
            }
            // ---- linear1 + relu  (N = 1024, one k-slice)
            gemv_stream<WT, V, MS, G_L1, false>(ring, tid, wg, mo, nskip, xin, PART, epi);
            if (epi) {
#pragma unroll
                for (int j = 0; j < FF_SA / 256; ++j) {
                    const float4 h = f4_add(gemv_out<WT, MS, G_L1>(PART, es, 256 * j, lane), ld4(v_l1b + 256 * j + 4 * lane));
                    float4 hr = make_float4(fmaxf(h.x, 0.f), fmaxf(h.y, 0.f), fmaxf(h.z, 0.f), fmaxf(h.w, 0.f));
                    if (SAVE && dm) hr = drop4(hr, dm + DM_H + 256 * j + 4 * lane);
                    if (SAVE && sv) st4(sv + DT_H + 256 * j + 4 * lane, hr);
                    put_x<WT, MS>(XB, es, 256 * j, lane, hr);
                }
            }
            __syncthreads(); DEN_DBG(0);
            // ---- linear2 + residual + norm2, then ca_block (mdiff_transformer.py:219-239, 152-163)
            gemv_stream<WT, V, MS, G_L2, false>(ring, tid, wg, mo, nskip, xin, PART, epi);
            if (epi) {
                float4 o2 = f4_add(part256<WT, MS>(PART, es, lane), ld4(v_l2b + 4 * lane));
                if (SAVE && dm) o2 = drop4(o2, dm + DM_2 + 4 * lane);
                const float4 v = f4_add(xr, o2);
                float4 xh2; float rs2;
                xr = wave_ln_stats(v, v_n2w, v_n2b, lane, xh2, rs2);
                if constexpr (CAQ) {
                    float4 xhc; float rsc;
                    const float4 qn = wave_ln_stats(xr, v_cnw, v_cnb, lane, xhc, rsc);       // ca_block.norm -> query input
                    if (SAVE && sv) {
                        st4(sv + DT_XH2 + 4 * lane, xh2); st4(sv + DT_XHC + 4 * lane, xhc);
                        if (lane == 0) { sv[DT_RS + 1] = rs2; sv[DT_RS + 2] = rsc; }
                    }
                    put_x<WT, MS>(XB, es, 0, lane, qn);
                } else {
                    // ONE condition token: x + Stylization(v) with a term that does not depend on x (table)
                    xr = f4_add(xr, ld4(CA_ADD + 4 * lane));
                    put_x<WT, MS>(XB, es, 0, lane, xr);
                }
            }
            __syncthreads(); DEN_DBG(0);
            if constexpr (CAQ) {
                gemv_stream<WT, V, MS, G_CAQ, false>(ring, tid, wg, mo, nskip, xin, PART, epi);
                if (epi) {
                    const float4 qv = f4_add(part256<WT, MS>(PART, es, lane), ld4(v_caq_b + 4 * lane));
                    const float mx = seg_reduce<true>(fmaxf(fmaxf(qv.x, qv.y), fmaxf(qv.z, qv.w)), seg);
                    const float4 e = make_float4(fast_exp(qv.x - mx), fast_exp(qv.y - mx), fast_exp(qv.z - mx), fast_exp(qv.w - mx));
                    const float inv = fast_rcp(seg_reduce<false>(e.x + e.y + e.z + e.w, seg));
                    const float4 qc = f4_scale(e, inv);                               // softmax over head_dim (:231)
                    // keys: softmax over the N tokens, per dim (:232)
                    float4 kr[DEN_MAXTOK - 2];
                    float4 kmx = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
                    for (int j = 0; j < DEN_MAXTOK - 2; ++j)
                        if (j < N) {
                            kr[j] = ld4(CT + j * 1024 + 512 + 4 * lane);
                            kmx = make_float4(fmaxf(kmx.x, kr[j].x), fmaxf(kmx.y, kr[j].y), fmaxf(kmx.z, kr[j].z), fmaxf(kmx.w, kr[j].w));
                        }
                    float4 ks = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int j = 0; j < DEN_MAXTOK - 2; ++j)
                        if (j < N) {
                            kr[j] = make_float4(fast_exp(kr[j].x - kmx.x), fast_exp(kr[j].y - kmx.y), fast_exp(kr[j].z - kmx.z), fast_exp(kr[j].w - kmx.w));
                            ks = f4_add(ks, kr[j]);
                        }
                    const float4 rks = make_float4(fast_rcp(ks.x), fast_rcp(ks.y), fast_rcp(ks.z), fast_rcp(ks.w));
                    float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int j = 0; j < DEN_MAXTOK - 2; ++j)
                        if (j < N) {
                            const float4 kc = make_float4(kr[j].x * rks.x, kr[j].y * rks.y, kr[j].z * rks.z, kr[j].w * rks.w);
                            const float dot = seg_reduce<false>(f4_dot(qc, kc), seg);                      // q . k_n per head
                            if (SAVE && sv && lane == 0) sv[DT_RS + 8 + j] = dot;
                            y = f4_fma(dot, ld4(CT + j * 1024 + 768 + 4 * lane), y);                       // (q k^T) v  (:236-237)
                        }
                    float4 xhy; float rsy;
                    const float4 hh = f4_adaln(wave_ln_stats(y, v_csnw, v_csnb, lane, xhy, rsy), ld4(TTS + 512 + 4 * lane), ld4(TTS + 768 + 4 * lane));
                    if (SAVE && sv) {
                        st4(sv + DT_QC + 4 * lane, qc); st4(sv + DT_XHY + 4 * lane, xhy); st4(sv + DT_U + 4 * lane, hh);
                        if (lane == 0) sv[DT_RS + 3] = rsy;
                    }
                    put_x<WT, MS>(XB, es, 0, lane, (SAVE && dm) ? drop4(f4_silu(hh), dm + DM_C + 4 * lane) : f4_silu(hh));
                }
                __syncthreads(); DEN_DBG(0);
                // ---- proj_out.out_layers + residual
                gemv_stream<WT, V, MS, G_CAO, false>(ring, tid, wg, mo, nskip, xin, PART, epi);
                if (epi) {
                    xr = f4_add(xr, f4_add(part256<WT, MS>(PART, es, lane), ld4(v_cao_b + 4 * lane)));
                    if (SAVE && sv) st4(sv + DT_X3 + 4 * lane, xr);
                    put_x<WT, MS>(XB, es, 0, lane, xr);
                }
                __syncthreads(); DEN_DBG(0);
            }
            // ---- ffn.linear1 + gelu  (N = 128, four k-slices: lanes 0..31 hold 4 outputs each)
            gemv_stream<WT, V, MS, G_F1, false>(ring, tid, wg, mo, nskip, xin, PART, epi);
            // request the next layer's operands (LDS-DMA into the other half of the double buffer) here, in the short
            // stages where the fill path has slack; the epilogue waves are busy and take no part
            if (!epi) stage_dma<MS, CAQ, MS, 1>(wave, lane, STG + (cur ^ 1) * stg_sz, vp, Ln, A.ttab + (size_t)(ln == 0 ? row_next : row) * SEEME_TROW, ln, A, b, N,
                                                A.trow_per_sample ? 0 : (ln == 0 ? step_next : step), ca_R);
            if (epi && lane < FF_D / 4) {
                const float4 g = f4_add(gemv_out<WT, MS, G_F1>(PART, es, 0, lane), ld4(v_f1b + 4 * lane));
                if (SAVE && sv) st4(sv + DT_Z1 + 4 * lane, g);
                const float4 gg = make_float4(fast_gelu(g.x), fast_gelu(g.y), fast_gelu(g.z), fast_gelu(g.w));
                put_x<WT, MS>(XB, es, 0, lane, (SAVE && dm) ? drop4(gg, dm + DM_F + 4 * lane) : gg);
            }
            __syncthreads(); DEN_DBG(0);
            // ---- ffn.linear2 -> AdaLN
            gemv_stream<WT, V, MS, G_F2, false>(ring, tid, wg, mo, nskip, xin, PART, epi);
            if (!epi) stage_dma<MS, CAQ, MS, 2>(wave, lane, STG + (cur ^ 1) * stg_sz, vp, Ln, A.ttab + (size_t)(ln == 0 ? row_next : row) * SEEME_TROW, ln, A, b, N,
                                                A.trow_per_sample ? 0 : (ln == 0 ? step_next : step), ca_R);
            if (epi) {
                const float4 y2 = f4_add(part256<WT, MS>(PART, es, lane), ld4(v_f2b + 4 * lane));
                float4 xhy2; float rsy2;
                const float4 hh = f4_adaln(wave_ln_stats(y2, v_fsnw, v_fsnb, lane, xhy2, rsy2), ld4(TTS + 1024 + 4 * lane), ld4(TTS + 1280 + 4 * lane));
                if (SAVE && sv) { st4(sv + DT_XHY2 + 4 * lane, xhy2); st4(sv + DT_U2 + 4 * lane, hh); if (lane == 0) sv[DT_RS + 4] = rsy2; }
                put_x<WT, MS>(XB, es, 0, lane, (SAVE && dm) ? drop4(f4_silu(hh), dm + DM_O + 4 * lane) : f4_silu(hh));
            }
            __syncthreads(); DEN_DBG(0);
            // ---- ffn.proj_out.out_layers + residual; its epilogue also prepares the input of the next layer
            //      (or, after the last layer, runs the stack norm and the scheduler step)
            gemv_stream<WT, V, MS, G_FO, true>(ring, tid, wg, mo, nskip, xin, PART, epi);
            if (l + 1 < SEEME_DEN_NL) {
                if (epi) {
                    xr = f4_add(xr, f4_add(part256<WT, MS>(PART, es, lane), ld4(v_fo_b + 4 * lane)));
                    if (SAVE && sv) st4(sv + DT_X4 + 4 * lane, xr);
                    if (l < 2) st4(keep + 256 + 256 * l, xr);                                   // xs.append(x) (cross_attention.py:70-72)
                    put_x<WT, MS>(XB, es, 0, lane, xr);
                    if (nskip) put_x<WT, MS>(XB, es, 256, lane, ld4(keep + (ln == 3 ? 512 : 256)));   // xs.pop(): layer 3 <- layer 1, layer 4 <- layer 0
                }
                __syncthreads(); DEN_DBG(0);
            } else {
                // ---- stack norm -> model output (cross_attention.py:82-83; mld_denoiser.py:222)
                float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
                if (epi) {
                    xr = f4_add(xr, f4_add(part256<WT, MS>(PART, es, lane), ld4(v_fo_b + 4 * lane)));
                    float4 xhf; float rsf;
                    e = wave_ln_stats(xr, CONSTV + 256, CONSTV + 512, lane, xhf, rsf);
                    if (SAVE && sv) {
                        st4(sv + DT_X4 + 4 * lane, xr);
                        st4(sv0 + DT_FIN + 4 * lane, xhf);
                        if (lane == 0) sv0[DT_FIN + 256] = rsf;
                    }
                    if (MS == 2) st4(PART + es * 256 + 4 * lane, e);
                }
                if (MS == 2) {   // classifier-free guidance (mld.py:488-492), uncond (sample 0) first
                    __syncthreads(); DEN_DBG(0);
                    if (epi && A.cfg) {
                        const float4 eu = ld4(PART + 4 * lane), ec = ld4(PART + 256 + 4 * lane);
                        const float g = A.guidance_scale;
                        e = make_float4(eu.x + g * (ec.x - eu.x), eu.y + g * (ec.y - eu.y), eu.z + g * (ec.z - eu.z), eu.w + g * (ec.w - eu.w));
                    }
                }
                if (A.sched == SEEME_SCHED_NONE) {   // steps == 1 by contract: the output is the model output
                    if (epi) st4(keep, e);
                } else if (epi) {
                    // ---- scheduler.step (mld.py:495-497; scalars prepared by seeme_amd/schedulers.py)
                    const CF32 c = coef_c + (size_t)step * 8;
                    const float c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3], c4 = c[4], c5 = c[5], clip = c[6], ptype = c[7];
                    float4 nz = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (A.noise != nullptr) nz = ld4(A.noise + ((size_t)step * A.B + bl) * 256 + 4 * lane);
                    const float4 lat = ld4(keep);
                    const float xs[4] = {lat.x, lat.y, lat.z, lat.w}, es4[4] = {e.x, e.y, e.z, e.w}, ns[4] = {nz.x, nz.y, nz.z, nz.w};
                    float o[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float x0, ep;
                        if (ptype == 0.f) { ep = es4[i]; x0 = (xs[i] - c1 * ep) / c0; }
                        else              { x0 = es4[i]; ep = (xs[i] - c0 * x0) / c1; }
                        if (clip != 0.f) x0 = fminf(fmaxf(x0, -1.f), 1.f);
                        o[i] = c2 * x0 + c3 * ep + c5 * xs[i] + c4 * ns[i];
                    }
                    const float4 nl = make_float4(o[0], o[1], o[2], o[3]);
                    st4(keep, nl);
                    xr = f4_add(nl, ld4(CONSTV + 4 * lane));                         // next step: sample + query_pos
                    put_x<WT, MS>(XB, es, 0, lane, xr);
                }
                __syncthreads(); DEN_DBG(0);
            }
            cur ^= 1;
        }
        row = row_next;
    }
    DEN_DBG(3);
    // CFG pair: one guided latent (both waves hold it, wave 0 writes); independent samples: one row each
    if (epi && (es == 0 || (!A.cfg && MS * b + es < A.B))) st4(A.out + (size_t)bl * 256 + 4 * lane, ld4(keep));
    // the last requested chunks are never consumed: keep them from being optimised into dangling loads
#pragma unroll
    for (int s = 0; s < DEN_R; ++s) asm volatile("" ::"v"(ring.r[s][0].x));
}

static size_t den_lds_bytes(int MS, int N, bool mfma) {
    return (size_t)(768 + MS * 768 + 2 * (VP_LAYER + STG_TT + MS * N * 1024 + (N > 1 ? 0 : MS * 256)) + XB_FLOATS(mfma, MS) + (mfma ? MS * 1024 : 4 * MS * 768)) * sizeof(float);
}

// XCD packing (SeemeSampleArgs.xcds): workgroups go to the 8 XCDs round-robin (SPX mode), and every XCD's L2 (4 MB) pulls the whole weight
// image (9 MB at 16 bit) from the Infinity Cache once per step whether 4 or 16 of its CUs consume it.  ~16 chains per XCD measured best
// (B=32: 2 XCDs 4.64 ms, 4: 4.67, 8: 4.74, 1: 4.86 -- 32 CUs on one L2 run into its bandwidth).  It pays for a launch that has the chip
// to itself: two streams packed onto the same two XCDs 12.7 k seqs/s vs 13.3 k dealt out, eight graph-replayed streams 13.1 k vs 16.1 k --
// which only the caller knows, so the count is an argument (seeme_amd/_lib.py default_xcds) and the library keeps no stream state.
// One mapping for the sampling and the backward kernel: grid = 8 * ceil(chains / k) workgroups, blockIdx % 8 >= k return at once.
static int den_xcd_grid(int chains, int want, int* xcds) {
    int k = (want >= 1 && want <= 7) ? want : 8;
    if (chains > 32 * k) k = 8;                                  // one workgroup per CU, 32 CUs per XCD
    *xcds = k;
    return k < 8 ? (chains + k - 1) / k * 8 : chains;
}
template <typename WT, int MS, int V>
static int launch_den(const DenKArgs& ka, hipStream_t st) {
    const int chains = (MS == 2 && !ka.s.cfg) ? (ka.s.B + 1) / 2 : ka.s.B;
    const size_t lds = den_lds_bytes(MS, ka.s.N, WT::MFMA);
    if (lds > 160 * 1024) return seeme_fail("denoiser_sample: LDS budget exceeded");
    SEEME_HIP(hipFuncSetAttribute((const void*)k_den_sample<WT, MS, V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    DenKArgs k2 = ka;
    k2.chains = chains;
    const int grid = den_xcd_grid(chains, ka.s.xcds, &k2.xcds);
    hipLaunchKernelGGL((k_den_sample<WT, MS, V>), dim3(grid), dim3(DEN_THREADS), lds, st, k2);
    return seeme_check_launch("k_den_sample");
}
template <typename WT, int MS>
static int launch_den_ms(const DenKArgs& ka, bool fold, hipStream_t st) {
    const bool caq = ka.s.N > 1 || ka.s.force_query;
    if (caq) return fold ? launch_den<WT, MS, V_CAQ | V_FOLD>(ka, st) : launch_den<WT, MS, V_CAQ>(ka, st);
    return fold ? launch_den<WT, MS, V_FOLD>(ka, st) : launch_den<WT, MS, 0>(ka, st);
}
template <typename WT>
static int launch_den_wt(const DenKArgs& ka, bool fold, hipStream_t st) {
    // two samples per workgroup: a CFG pair, or -- when there are more samples than CUs and all share the step's
    // timestep -- two independent samples behind one weight stream (the matrix-core operand has rows to spare)
    const bool pair = ka.s.cfg || (ka.s.B > SEEME_DEN_PAIR_ABOVE && !ka.s.trow_per_sample);
    return pair ? launch_den_ms<WT, 2>(ka, fold, st) : launch_den_ms<WT, 1>(ka, fold, st);
}

extern "C" int seeme_denoiser_sample(const SeemeDenoiserWeights* w, const SeemeSampleArgs* a, void* stream) {
    if (a->B <= 0) return seeme_fail("denoiser_sample: B must be > 0");
    if (a->N < 1 || a->N > DEN_MAXTOK - 2) return seeme_fail("denoiser_sample: 1 <= N <= 4 condition tokens");
    if (w->nhead != 1 && w->nhead != 2 && w->nhead != 4) return seeme_fail("denoiser_sample: nhead must be 1, 2 or 4");
    if (w->ff_sa != FF_SA || w->ff != FF_D) return seeme_fail("denoiser_sample: built for sa ff 1024 / ffn_dim 128 (all reference configs)");
    if (w->layout == nullptr) return seeme_fail("denoiser_sample: layout table missing");
    if (a->sched == SEEME_SCHED_NONE && a->steps != 1) return seeme_fail("denoiser_sample: SCHED_NONE needs steps == 1");
    if (a->steps < 1) return seeme_fail("denoiser_sample: steps must be >= 1");
    if (a->save != nullptr && (a->steps != 1 || a->cfg || w->sa_fold || w->wdtype != 0 || !(a->N > 1 || a->force_query) || !a->trow_per_sample))
        return seeme_fail("denoiser_sample: save needs one step, no CFG, the unfolded fp32 image, the query GEMV and per-sample rows");
    DenKArgs ka;
    ka.wg = w->wg; ka.vp = w->vp; ka.lay = seeme_make_den_layout(FF_SA, FF_D);
    ka.wg_bytes = (int)(ka.lay.wg_total * (w->wdtype == 0 ? 4 : 2));
    ka.nhead = w->nhead; ka.s = *a;
    hipStream_t st = (hipStream_t)stream;
    const bool fold = w->sa_fold != 0;
    if (fold && w->nhead != 1) return seeme_fail("denoiser_sample: sa_fold needs nhead == 1");
    if (a->N == 1 && a->catab == nullptr && !a->force_query) return seeme_fail("denoiser_sample: N == 1 needs the ca table (seeme_denoiser_ca_tables)");
    if (w->wdtype == 0) return launch_den_wt<WF32>(ka, fold, st);
    if (w->wdtype == 1) return launch_den_wt<WBF16>(ka, fold, st);
    if (w->wdtype == 2) return launch_den_wt<WF16>(ka, fold, st);
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
    // Two GEMMs on the same rows as ONE batched launch (z = 0, 1):
    //   z = 0: sa_block K|V of the condition tokens, all layers at once -> ctab[:, 0:2560]
    //   z = 1: ca_block key|value of text_norm(xf) for all layers -- the per-layer LayerNorm affine is folded into the weights, so
    //          the input is the affine-free LayerNorm of the condition (mdiff_transformer.py:230,234); the token softmax of the
    //          keys is applied in-kernel -> ctab[:, 2560:5120]
    if (w->ca_fold_w == nullptr) return seeme_fail("cond_tables: folded key/value weights missing");
    LinearKArgs ka{};
    ka.a.A = cond; ka.a.lda = 256; ka.a.K1 = 256; ka.a.K = 256; ka.a.W = w->kv_cat_w; ka.a.ldw = 256; ka.a.bias = w->kv_cat_b;
    ka.a.Y = ctab; ka.a.ldy = SEEME_CROW; ka.a.M = M; ka.a.N = 2560; ka.a.eps = 1e-5f;
    ka.a.pre_ln_w = w->ln_ones; ka.a.pre_ln_b = w->ln_zeros;
    ka.nz = 2; ka.zs_a = 0; ka.zs_w = (long)(w->ca_fold_w - w->kv_cat_w); ka.zs_b = (long)(w->ca_fold_b - w->kv_cat_b); ka.zs_y = 2560;
    ka.zs_ln = 0; ka.pre_ln_zmin = 1;
    return seeme_launch_linear(ka, st);
}

// ------------------------------------------------------------------ ca_block table for ONE condition token
// H[l][(bc*R + r)][256] = SiLU( LN(value_l(cond_bc); proj_out.norm) * (1 + scale) + shift ), one wave per row
__global__ __launch_bounds__(256) void k_ca_rows(const float* __restrict__ ctab, const float* __restrict__ ttab,
                                                 const int32_t* __restrict__ trow, int per_sample, int n_trow, int R, int rows,
                                                 const float* __restrict__ nw, const float* __restrict__ nb, float* __restrict__ H) {
    const int lane = threadIdx.x & 63, l = blockIdx.y;
    const int r0 = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r0 >= rows) return;
    const int bc = r0 / R, r = r0 - bc * R;
    const int row = per_sample ? trow[bc % n_trow] : trow[r];
    const float4 v = ld4(ctab + (size_t)bc * SEEME_CROW + 2560 + l * 512 + 256 + 4 * lane);
    const float4 hn = wave_ln(v, nw + l * 256, nb + l * 256, lane);
    const float* tt = ttab + (size_t)row * SEEME_TROW + 2560 + l * 1024;
    st4(H + ((size_t)l * rows + r0) * 256 + 4 * lane, f4_silu(f4_adaln(hn, ld4(tt + 4 * lane), ld4(tt + 256 + 4 * lane))));
}

extern "C" int seeme_denoiser_ca_tables(const SeemeDenoiserWeights* w, const float* ctab, const float* ttab, const int32_t* trow,
                                        int trow_per_sample, int n_trow, int Bc, float* catab,
                                        void* workspace, size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (Bc <= 0 || n_trow <= 0) return seeme_fail("ca_tables: empty batch / row list");
    const int R = trow_per_sample ? 1 : n_trow;
    const int rows = Bc * R;
    if (ws_bytes < (size_t)SEEME_DEN_NL * rows * 256 * sizeof(float)) return seeme_fail("ca_tables: workspace too small");
    if (w->ca_pn_w == nullptr || w->ca_po_w == nullptr) return seeme_fail("ca_tables: proj_out weights missing");
    float* H = (float*)workspace;
    hipLaunchKernelGGL(k_ca_rows, dim3((rows + 3) / 4, SEEME_DEN_NL), dim3(256), 0, st, ctab, ttab, trow, trow_per_sample, n_trow, R, rows,
                       w->ca_pn_w, w->ca_pn_b, H);
    int rc = seeme_check_launch("k_ca_rows");
    if (rc) return rc;
    // proj_out.out_layers of the five layers as one batched launch: Linear(256,256) -> catab[:, :, l, :]
    LinearKArgs ka{};
    ka.a.A = H; ka.a.lda = 256; ka.a.K1 = 256; ka.a.W = w->ca_po_w; ka.a.ldw = 256; ka.a.bias = w->ca_po_b;
    ka.a.Y = catab; ka.a.ldy = SEEME_DEN_NL * 256; ka.a.M = rows; ka.a.N = 256; ka.a.K = 256; ka.a.eps = 1e-5f;
    ka.nz = SEEME_DEN_NL; ka.zs_a = (long)rows * 256; ka.zs_w = 256 * 256; ka.zs_b = 256; ka.zs_y = 256;
    return seeme_launch_linear(ka, st);
}

#ifdef DEN_DBG_TIMES
extern "C" int seeme_debug_den_times(unsigned long long* host, int n) {
    SEEME_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(den_dbg_times), sizeof(unsigned long long) * (size_t)(n < 512 ? n : 512)));
    return 0;
}
#endif

#include "den_train.inc.hip"
#include "den_cluster.inc.hip"
#include "den_cluster_ms.inc.hip"
