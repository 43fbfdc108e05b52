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

#define DEN_THREADS 512
#define DEN_MAXTOK 6   // 1 latent + N<=4 condition tokens + 1 time token

// ------------------------------------------------------------------ weight element types
struct WF32 { typedef float T; static constexpr int KV = 4; };
struct WBF16 { typedef uint16_t T; static constexpr int KV = 8; };

// partial dot products of output n over k-groups [q0, q0+nq) for MS samples
template <int MS>
__device__ __forceinline__ void gemv_part(const float* __restrict__ Wp, int N, int n, int q0, int nq,
                                          const float* __restrict__ x, int ldx, float (&sum)[MS]) {
    const float4* wp = reinterpret_cast<const float4*>(Wp) + (size_t)q0 * N + n;
    float acc[MS][4];
#pragma unroll
    for (int s = 0; s < MS; ++s) acc[s][0] = acc[s][1] = acc[s][2] = acc[s][3] = 0.f;
#pragma unroll 8
    for (int q = 0; q < nq; ++q) {
        const float4 w = wp[(size_t)q * N];
#pragma unroll
        for (int s = 0; s < MS; ++s) {
            const float4 xv = *reinterpret_cast<const float4*>(x + s * ldx + 4 * (q0 + q));
            acc[s][0] = fmaf(w.x, xv.x, acc[s][0]);
            acc[s][1] = fmaf(w.y, xv.y, acc[s][1]);
            acc[s][2] = fmaf(w.z, xv.z, acc[s][2]);
            acc[s][3] = fmaf(w.w, xv.w, acc[s][3]);
        }
    }
#pragma unroll
    for (int s = 0; s < MS; ++s) sum[s] = (acc[s][0] + acc[s][1]) + (acc[s][2] + acc[s][3]);
}

__device__ __forceinline__ float bf_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

template <int MS>
__device__ __forceinline__ void gemv_part(const uint16_t* __restrict__ Wp, int N, int n, int q0, int nq,
                                          const float* __restrict__ x, int ldx, float (&sum)[MS]) {
    const uint4* wp = reinterpret_cast<const uint4*>(Wp) + (size_t)q0 * N + n;
    float acc[MS][4];
#pragma unroll
    for (int s = 0; s < MS; ++s) acc[s][0] = acc[s][1] = acc[s][2] = acc[s][3] = 0.f;
#pragma unroll 8
    for (int q = 0; q < nq; ++q) {
        const uint4 w = wp[(size_t)q * N];
#pragma unroll
        for (int s = 0; s < MS; ++s) {
            const float* xp = x + s * ldx + 8 * (q0 + q);
            const float4 x0 = *reinterpret_cast<const float4*>(xp);
            const float4 x1 = *reinterpret_cast<const float4*>(xp + 4);
            acc[s][0] = fmaf(bf_lo(w.x), x0.x, acc[s][0]);
            acc[s][1] = fmaf(bf_hi(w.x), x0.y, acc[s][1]);
            acc[s][2] = fmaf(bf_lo(w.y), x0.z, acc[s][2]);
            acc[s][3] = fmaf(bf_hi(w.y), x0.w, acc[s][3]);
            acc[s][0] = fmaf(bf_lo(w.z), x1.x, acc[s][0]);
            acc[s][1] = fmaf(bf_hi(w.z), x1.y, acc[s][1]);
            acc[s][2] = fmaf(bf_lo(w.w), x1.z, acc[s][2]);
            acc[s][3] = fmaf(bf_hi(w.w), x1.w, acc[s][3]);
        }
    }
#pragma unroll
    for (int s = 0; s < MS; ++s) sum[s] = (acc[s][0] + acc[s][1]) + (acc[s][2] + acc[s][3]);
}

// y[s][n] = act(W x[s] + bias)[n] -> LDS out[s*ldo + n].  Ends with a barrier.
template <typename WT, int MS>
__device__ __forceinline__ void gemv_lds(const typename WT::T* __restrict__ Wp, int K, int N,
                                         const float* __restrict__ x, int ldx, const float* __restrict__ bias,
                                         int act, float* __restrict__ out, int ldo, float* __restrict__ part) {
    const int tid = threadIdx.x;
    const int nq_total = K / WT::KV;
    if (N >= DEN_THREADS) {
        for (int n = tid; n < N; n += DEN_THREADS) {
            float s[MS];
            gemv_part<MS>(Wp, N, n, 0, nq_total, x, ldx, s);
#pragma unroll
            for (int ms = 0; ms < MS; ++ms) out[ms * ldo + n] = act_apply(s[ms] + bias[n], act);
        }
        __syncthreads();
    } else {
        const int KS = DEN_THREADS / N, ks = tid / N, n = tid - ks * N, nq = nq_total / KS;
        float s[MS];
        gemv_part<MS>(Wp, N, n, ks * nq, nq, x, ldx, s);
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) part[(ks * MS + ms) * N + n] = s[ms];
        __syncthreads();
        for (int idx = tid; idx < N * MS; idx += DEN_THREADS) {
            const int ms = idx / N, nn = idx - ms * N;
            float v = bias[nn];
            for (int k2 = 0; k2 < KS; ++k2) v += part[(k2 * MS + ms) * N + nn];
            out[ms * ldo + nn] = act_apply(v, act);
        }
        __syncthreads();
    }
}

// N = 256 GEMV whose result goes straight to the owner thread (sample ms = tid>>8, dim d = tid&255).
// One barrier inside; the caller must have a barrier between this call and the next write to `part`.
template <typename WT, int MS>
__device__ __forceinline__ float gemv256_owner(const typename WT::T* __restrict__ Wp, int K,
                                               const float* __restrict__ x, int ldx, float* __restrict__ part) {
    const int tid = threadIdx.x, ks = tid >> 8, n = tid & 255;
    const int nq = K / WT::KV / 2;
    float s[MS];
    gemv_part<MS>(Wp, 256, n, ks * nq, nq, x, ldx, s);
#pragma unroll
    for (int ms = 0; ms < MS; ++ms) part[(ks * MS + ms) * 256 + n] = s[ms];
    __syncthreads();
    float r = 0.f;
    if (tid < 256 * MS) {
        const int ms = tid >> 8;
        r = part[ms * 256 + n] + part[(MS + ms) * 256 + n];
    }
    return r;
}

// ------------------------------------------------------------------ group reductions
// A "group" = the 256 owner threads (4 waves) of one sample; a head segment = 4/H consecutive waves.
// `red` holds 2 x 8 waves x 8 values; `cnt` alternates the half so ONE barrier per reduction suffices.
template <int NV, bool IS_MAX>
__device__ __forceinline__ void group_seg_reduce(float (&v)[NV], int nv, int H, float* __restrict__ red, int& cnt) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    float* buf = red + (cnt & 1) * 64;
    ++cnt;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (i < nv) {
            const float w = IS_MAX ? wave_max(v[i]) : wave_sum(v[i]);
            if (lane == 0) buf[wave * 8 + i] = w;
        }
    __syncthreads();
    const int wph = 4 / H;                       // waves per head
    const int w0 = (wave & ~3) + ((wave & 3) / wph) * wph;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (i < nv) {
            float r = buf[w0 * 8 + i];
            for (int j = 1; j < wph; ++j) r = IS_MAX ? fmaxf(r, buf[(w0 + j) * 8 + i]) : r + buf[(w0 + j) * 8 + i];
            v[i] = r;
        }
}
__device__ __forceinline__ float group_sum(float v, float* red, int& cnt) {
    float a[1] = {v};
    group_seg_reduce<1, false>(a, 1, 1, red, cnt);
    return a[0];
}
// LayerNorm over the 256 dims of each group (two-pass, torch semantics); value per owner thread.
__device__ __forceinline__ float group_ln(float v, const float* __restrict__ w, const float* __restrict__ b,
                                          int d, float* red, int& cnt) {
    const float mean = group_sum(v, red, cnt) * (1.f / 256.f);
    const float c = v - mean;
    const float var = group_sum(c * c, red, cnt) * (1.f / 256.f);
    return c * (1.f / sqrtf(var + 1e-5f)) * w[d] + b[d];
}

// ------------------------------------------------------------------ the persistent sampling kernel
struct DenKArgs {
    const void* wg; const float* vp;
    DenLayout lay;
    int nhead, ff_sa, ff;
    SeemeSampleArgs s;
};

template <typename WT, int MS>
__global__ __launch_bounds__(DEN_THREADS) void k_den_sample(const DenKArgs ka) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef typename WT::T wt_t;
    const wt_t* __restrict__ wg = reinterpret_cast<const wt_t*>(ka.wg);
    const float* __restrict__ vp = ka.vp;
    const SeemeSampleArgs& A = ka.s;
    const int tid = threadIdx.x, ms = tid >> 8, d = tid & 255;
    const bool own = tid < 256 * MS;
    const int b = blockIdx.x, N = A.N, NS = N + 2, H = ka.nhead;
    const int ff_sa = ka.ff_sa, ff = ka.ff;
    const int vmax = ff_sa > 512 ? ff_sa : 512;

    float* X = smem;                       // [MS][256]
    float* LAT = X + MS * 256;             // [256]
    float* SK = LAT + 256;                 // [2][MS][256]
    float* VA = SK + 2 * MS * 256;         // [MS][vmax]
    float* VB = VA + MS * vmax;            // [MS][256]
    float* QKV = VB + MS * 256;            // [MS][768]
    float* PART = QKV + MS * 768;          // [4*MS*256]  (KS*MS*N <= 512*MS... sized for KS=4,N=128 / KS=2,N=256)
    float* RED = PART + 4 * MS * 256;      // [2][8][8]
    int cnt = 0;

    // condition tables of this workgroup's sample(s): CFG -> ms 0 = uncond (first half), ms 1 = cond
    const int bc = (MS == 2) ? (ms == 0 ? b : A.B + b) : b;
    const float* __restrict__ ct = A.ctab + (size_t)(own ? bc : b) * N * SEEME_CROW;
    const float sa_scale = 1.f / sqrtf((float)(256 / H));

    if (tid < 256) LAT[tid] = A.latents[(size_t)b * 256 + tid];
    __syncthreads();

    for (int step = 0; step < A.steps; ++step) {
        const int row = A.trow_per_sample ? A.trow[b] : A.trow[step];
        const float* __restrict__ tt = A.ttab + (size_t)row * SEEME_TROW;

        float xr = 0.f;  // owner's current value of token 0
        if (own) { xr = LAT[d] + vp[ka.lay.pe0 + d]; X[ms * 256 + d] = xr; }   // mld_denoiser.py:210
        __syncthreads();

        for (int l = 0; l < SEEME_DEN_NL; ++l) {
            const DenLayerOff& L = ka.lay.L[l];
            // ---- skip connection: Linear(cat[x, xs.pop()])  (cross_attention.py:77-79)
            if (l >= 3) {
                if (own) { VA[ms * 512 + d] = xr; VA[ms * 512 + 256 + d] = SK[((4 - l) * MS + ms) * 256 + d]; }
                __syncthreads();
                const float r = gemv256_owner<WT, MS>(wg + L.skip, 512, VA, 512, PART);
                if (own) { xr = r + vp[L.skip_b + d]; X[ms * 256 + d] = xr; }
                __syncthreads();
            }
            // ---- sa_block: post-norm encoder layer over [x, xf.., emb]; only token 0 is kept
            //      (mdiff_transformer.py:292-297); K/V of xf and emb come from the tables.
            gemv_lds<WT, MS>(wg + L.inp, 256, 768, X, 256, vp + L.in_b, SEEME_ACT_NONE, QKV, 768, PART);
            float sc[DEN_MAXTOK];
            {
                const float q = own ? QKV[ms * 768 + d] : 0.f;
                sc[0] = own ? q * QKV[ms * 768 + 256 + d] : 0.f;
#pragma unroll
                for (int j = 0; j < DEN_MAXTOK - 2; ++j)
                    if (j < N) sc[1 + j] = q * ct[(size_t)j * SEEME_CROW + l * 512 + d];
                // time token is the LAST of the sequence (mdiff_transformer.py:295)
#pragma unroll
                for (int j = 1; j < DEN_MAXTOK; ++j)
                    if (j == N + 1) sc[j] = q * tt[l * 512 + d];
            }
            group_seg_reduce<DEN_MAXTOK, false>(sc, NS, H, RED, cnt);
            if (own) {
                float mx = -INFINITY;
#pragma unroll
                for (int j = 0; j < DEN_MAXTOK; ++j) if (j < NS) { sc[j] *= sa_scale; mx = fmaxf(mx, sc[j]); }
                float sum = 0.f;
#pragma unroll
                for (int j = 0; j < DEN_MAXTOK; ++j) if (j < NS) { sc[j] = expf(sc[j] - mx); sum += sc[j]; }
                const float inv = 1.f / sum;
                float att = sc[0] * inv * QKV[ms * 768 + 512 + d];
#pragma unroll
                for (int j = 0; j < DEN_MAXTOK - 2; ++j)
                    if (j < N) att = fmaf(sc[1 + j] * inv, ct[(size_t)j * SEEME_CROW + l * 512 + 256 + d], att);
#pragma unroll
                for (int j = 1; j < DEN_MAXTOK; ++j)
                    if (j == N + 1) att = fmaf(sc[j] * inv, tt[l * 512 + 256 + d], att);
                VB[ms * 256 + d] = att;
            }
            __syncthreads();
            {
                const float r = gemv256_owner<WT, MS>(wg + L.outp, 256, VB, 256, PART);
                float v = xr + r + (own ? vp[L.out_b + d] : 0.f);
                xr = group_ln(v, vp + L.n1w, vp + L.n1b, d, RED, cnt);
                if (own) X[ms * 256 + d] = xr;
                __syncthreads();
            }
            gemv_lds<WT, MS>(wg + L.l1, 256, ff_sa, X, 256, vp + L.l1b, SEEME_ACT_RELU, VA, ff_sa, PART);
            {
                const float r = gemv256_owner<WT, MS>(wg + L.l2, ff_sa, VA, ff_sa, PART);
                float v = xr + r + (own ? vp[L.l2b + d] : 0.f);
                xr = group_ln(v, vp + L.n2w, vp + L.n2b, d, RED, cnt);
            }
            // ---- ca_block: linear cross-attention + AdaLN (mdiff_transformer.py:219-239, 152-163)
            {
                const float xn = group_ln(xr, vp + L.cnw, vp + L.cnb, d, RED, cnt);
                if (own) VB[ms * 256 + d] = xn;
                __syncthreads();
                const float r = gemv256_owner<WT, MS>(wg + L.caq, 256, VB, 256, PART);
                float qv[1] = {own ? r + vp[L.caq_b + d] : -INFINITY};
                float mx[1] = {qv[0]};
                group_seg_reduce<1, true>(mx, 1, H, RED, cnt);
                float e[1] = {own ? expf(qv[0] - mx[0]) : 0.f};
                float sm[1] = {e[0]};
                group_seg_reduce<1, false>(sm, 1, H, RED, cnt);
                const float qc = e[0] / sm[0];                          // softmax over head_dim (:231)
                float kr[DEN_MAXTOK - 2], dots[DEN_MAXTOK];
                float kmx = -INFINITY;
#pragma unroll
                for (int j = 0; j < DEN_MAXTOK - 2; ++j)
                    if (j < N) { kr[j] = ct[(size_t)j * SEEME_CROW + 2560 + l * 512 + d]; kmx = fmaxf(kmx, kr[j]); }
                float ksum = 0.f;
#pragma unroll
                for (int j = 0; j < DEN_MAXTOK - 2; ++j) if (j < N) { kr[j] = expf(kr[j] - kmx); ksum += kr[j]; }
#pragma unroll
                for (int j = 0; j < DEN_MAXTOK - 2; ++j) if (j < N) dots[j] = own ? qc * (kr[j] / ksum) : 0.f;  // softmax over tokens (:232)
                group_seg_reduce<DEN_MAXTOK, false>(dots, N, H, RED, cnt);
                float y = 0.f;
#pragma unroll
                for (int j = 0; j < DEN_MAXTOK - 2; ++j)
                    if (j < N) y = fmaf(dots[j], ct[(size_t)j * SEEME_CROW + 2560 + l * 512 + 256 + d], y);  // q (k^T v)  (:236-237)
                float hh = group_ln(y, vp + L.csnw, vp + L.csnb, d, RED, cnt);
                hh = hh * (1.f + tt[2560 + l * 1024 + d]) + tt[2560 + l * 1024 + 256 + d];
                __syncthreads();   // PART / VB reuse guard
                if (own) VB[ms * 256 + d] = act_apply(hh, SEEME_ACT_SILU);
                __syncthreads();
                const float r2 = gemv256_owner<WT, MS>(wg + L.cao, 256, VB, 256, PART);
                if (own) { xr = xr + r2 + vp[L.cao_b + d]; X[ms * 256 + d] = xr; }
                __syncthreads();
            }
            // ---- ffn + AdaLN (mdiff_transformer.py:251-254)
            gemv_lds<WT, MS>(wg + L.f1, 256, ff, X, 256, vp + L.f1b, SEEME_ACT_GELU, VA, ff, PART);
            {
                const float r = gemv256_owner<WT, MS>(wg + L.f2, ff, VA, ff, PART);
                const float y2 = r + (own ? vp[L.f2b + d] : 0.f);
                float hh = group_ln(y2, vp + L.fsnw, vp + L.fsnb, d, RED, cnt);
                hh = hh * (1.f + tt[2560 + l * 1024 + 512 + d]) + tt[2560 + l * 1024 + 768 + d];
                __syncthreads();
                if (own) VB[ms * 256 + d] = act_apply(hh, SEEME_ACT_SILU);
                __syncthreads();
                const float r2 = gemv256_owner<WT, MS>(wg + L.fo, 256, VB, 256, PART);
                if (own) {
                    xr = xr + r2 + vp[L.fo_b + d];
                    X[ms * 256 + d] = xr;
                    if (l < 2) SK[(l * MS + ms) * 256 + d] = xr;
                }
                __syncthreads();
            }
        }
        // ---- stack norm -> model output (cross_attention.py:82-83; mld_denoiser.py:222)
        float e = group_ln(xr, vp + ka.lay.fnw, vp + ka.lay.fnb, d, RED, cnt);
        if (MS == 2) {   // classifier-free guidance (mld.py:488-492), uncond first
            __syncthreads();
            if (own) VB[ms * 256 + d] = e;
            __syncthreads();
            e = VB[d] + A.guidance_scale * (VB[256 + d] - VB[d]);
        }
        if (A.sched == SEEME_SCHED_NONE) {
            if (tid < 256) A.out[(size_t)b * 256 + d] = e;
            return;   // steps == 1 by contract
        }
        // ---- scheduler.step (mld.py:495-497; scalars prepared by seeme_amd/schedulers.py)
        if (tid < 256) {
            const float* __restrict__ c = A.coef + (size_t)step * 8;
            const float x = LAT[d];
            float x0, ep;
            if (c[7] == 0.f) { ep = e; x0 = (x - c[1] * ep) / c[0]; }
            else             { x0 = e; ep = (x - c[0] * x0) / c[1]; }
            if (c[6] != 0.f) x0 = fminf(fmaxf(x0, -1.f), 1.f);
            float prev = c[2] * x0 + c[3] * ep + c[5] * x;
            if (A.noise != nullptr) prev += c[4] * A.noise[((size_t)step * A.B + b) * 256 + d];
            LAT[d] = prev;
        }
        __syncthreads();
    }
    if (tid < 256) A.out[(size_t)b * 256 + d] = LAT[d];
}

static size_t den_lds_bytes(int MS, int ff_sa) {
    const int vmax = ff_sa > 512 ? ff_sa : 512;
    return (size_t)(MS * 256 + 256 + 2 * MS * 256 + MS * vmax + MS * 256 + MS * 768 + 4 * MS * 256 + 128) * sizeof(float);
}

template <typename WT, int MS>
static int launch_den(const DenKArgs& ka, hipStream_t st) {
    const size_t lds = den_lds_bytes(MS, ka.ff_sa);
    SEEME_HIP(hipFuncSetAttribute((const void*)k_den_sample<WT, MS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_den_sample<WT, MS>), dim3(ka.s.B), dim3(DEN_THREADS), lds, st, ka);
    return seeme_check_launch("k_den_sample");
}

extern "C" int seeme_denoiser_sample(const SeemeDenoiserWeights* w, const SeemeSampleArgs* a, void* stream) {
    if (a->B <= 0) return seeme_fail("denoiser_sample: B must be > 0");
    if (a->N < 1 || a->N > DEN_MAXTOK - 2) return seeme_fail("denoiser_sample: 1 <= N <= 4 condition tokens");
    if (w->nhead != 1 && w->nhead != 2 && w->nhead != 4) return seeme_fail("denoiser_sample: nhead must be 1, 2 or 4");
    if (w->ff_sa % 512 != 0 || w->ff % 64 != 0 || w->ff > 512) return seeme_fail("denoiser_sample: unsupported ff sizes");
    if (a->sched == SEEME_SCHED_NONE && a->steps != 1) return seeme_fail("denoiser_sample: SCHED_NONE needs steps == 1");
    if (a->steps < 1) return seeme_fail("denoiser_sample: steps must be >= 1");
    DenKArgs ka;
    ka.wg = w->wg; ka.vp = w->vp; ka.lay = seeme_make_den_layout(w->ff_sa, w->ff);
    ka.nhead = w->nhead; ka.ff_sa = w->ff_sa; ka.ff = w->ff; ka.s = *a;
    hipStream_t st = (hipStream_t)stream;
    if (w->wdtype == 0) return a->cfg ? launch_den<WF32, 2>(ka, st) : launch_den<WF32, 1>(ka, st);
    if (w->wdtype == 1) return a->cfg ? launch_den<WBF16, 2>(ka, st) : launch_den<WBF16, 1>(ka, st);
    return seeme_fail("denoiser_sample: wdtype must be 0 (fp32) or 1 (bf16)");
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
    // ca_block: key|value of text_norm(xf) per layer (mdiff_transformer.py:230,234); the token softmax is applied in-kernel
    for (int l = 0; l < SEEME_NLAYERS; ++l)
        if ((rc = seeme_linear_simple(st, cond, 256, w->ca_kv_w[l], 256, w->ca_kv_b[l], ctab + 2560 + l * 512, SEEME_CROW,
                                      M, 512, 256, 0, 0, w->ca_tn_w[l], w->ca_tn_b[l]))) return rc;
    return 0;
}
