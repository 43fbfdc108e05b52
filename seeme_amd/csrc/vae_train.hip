// vae_train.hip -- row kernels of the hand-written STAGE-1 (VAE) training forward / backward; the GEMMs of that step
// (projections, per-sequence Q K^T, P V and their gradients) run on the grouped fp32 GEMM of glue_kernels.hip.
// Reference: the autograd graph of MldVae.encode / decode in training (mld_vae.py:128-256; post-norm layers and skip
// stacks of cross_attention.py:41-147,281-367) -- what loss.backward() walks in train_vae_forward (mld.py:633-885).
// Layout: token rows [B*S][256] fp32, one wave per row (4 features per lane).
#include "common.hpp"
#include "api_util.hpp"

// y = LayerNorm(sub + res) * gamma + beta ; xhat and rstd are kept for the backward.  sub_seq_rows > 0: `sub` has one row
// per sequence (the cross-attention vector of a single memory token, broadcast over the sequence's rows).
__global__ __launch_bounds__(256) void k_vt_add_ln(SeemeVtLn a) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + wave;
    if (row >= a.M) return;
    const long srow = a.sub_seq_rows > 0 ? row / a.sub_seq_rows : row;
    float4 v = *reinterpret_cast<const float4*>(a.sub + srow * 256 + lane * 4);
    if (a.res != nullptr) {
        const float4 r = *reinterpret_cast<const float4*>(a.res + row * 256 + lane * 4);
        v = make_float4(v.x + r.x, v.y + r.y, v.z + r.z, v.w + r.w);
    }
    const float mean = wave_sum(v.x + v.y + v.z + v.w) * (1.f / 256.f);
    const float4 c = make_float4(v.x - mean, v.y - mean, v.z - mean, v.w - mean);
    const float var = wave_sum(c.x * c.x + c.y * c.y + c.z * c.z + c.w * c.w) * (1.f / 256.f);
    const float rs = 1.f / sqrtf(var + a.eps);
    const float4 xh = make_float4(c.x * rs, c.y * rs, c.z * rs, c.w * rs);
    const float4 g = *reinterpret_cast<const float4*>(a.gamma + lane * 4), be = *reinterpret_cast<const float4*>(a.beta + lane * 4);
    *reinterpret_cast<float4*>(a.y + row * 256 + lane * 4) = make_float4(xh.x * g.x + be.x, xh.y * g.y + be.y, xh.z * g.z + be.z, xh.w * g.w + be.w);
    *reinterpret_cast<float4*>(a.xhat + row * 256 + lane * 4) = xh;
    if (lane == 0) a.rstd[row] = rs;
}

// dpre (+)= rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma ; dgamma += sum_rows dy * xhat ; dbeta += sum_rows dy.
// 32 rows per block; the column sums go through LDS and one atomic per column and block.
__global__ __launch_bounds__(256) void k_vt_ln_bwd(SeemeVtLnBwd a) {
    __shared__ float red[2][4][256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float4 g4 = *reinterpret_cast<const float4*>(a.gamma + lane * 4);
    float4 sg = make_float4(0.f, 0.f, 0.f, 0.f), sb = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int rr = 0; rr < 8; ++rr) {
        const long row = (long)blockIdx.x * 32 + wave * 8 + rr;
        if (row >= a.M) break;
        float4 dy = *reinterpret_cast<const float4*>(a.dy + row * 256 + lane * 4);
        if (a.dy2 != nullptr) {
            const float4 d2 = *reinterpret_cast<const float4*>(a.dy2 + row * 256 + lane * 4);
            dy = make_float4(dy.x + d2.x, dy.y + d2.y, dy.z + d2.z, dy.w + d2.w);
        }
        const float4 xh = *reinterpret_cast<const float4*>(a.xhat + row * 256 + lane * 4);
        const float4 g = make_float4(dy.x * g4.x, dy.y * g4.y, dy.z * g4.z, dy.w * g4.w);
        const float m1 = wave_sum(g.x + g.y + g.z + g.w) * (1.f / 256.f);
        const float m2 = wave_sum(g.x * xh.x + g.y * xh.y + g.z * xh.z + g.w * xh.w) * (1.f / 256.f);
        const float rs = a.rstd[row];
        float4 d = make_float4(rs * (g.x - m1 - xh.x * m2), rs * (g.y - m1 - xh.y * m2), rs * (g.z - m1 - xh.z * m2), rs * (g.w - m1 - xh.w * m2));
        float* dp = a.dpre + row * 256 + lane * 4;
        if (a.accumulate) {
            const float4 o = *reinterpret_cast<const float4*>(dp);
            d = make_float4(d.x + o.x, d.y + o.y, d.z + o.z, d.w + o.w);
        }
        *reinterpret_cast<float4*>(dp) = d;
        sg = make_float4(sg.x + dy.x * xh.x, sg.y + dy.y * xh.y, sg.z + dy.z * xh.z, sg.w + dy.w * xh.w);
        sb = make_float4(sb.x + dy.x, sb.y + dy.y, sb.z + dy.z, sb.w + dy.w);
    }
    *reinterpret_cast<float4*>(&red[0][wave][lane * 4]) = sg;
    *reinterpret_cast<float4*>(&red[1][wave][lane * 4]) = sb;
    __syncthreads();
    const int c = threadIdx.x;
    atomicAdd(a.dgamma + c, red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
    atomicAdd(a.dbeta + c, red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
}

// scores [B][S][S] (q k^T, unscaled) -> probabilities in place: softmax over the valid keys of scale * s, zeros elsewhere.
__global__ __launch_bounds__(256) void k_vt_softmax_fwd(float* __restrict__ s, const int32_t* __restrict__ lengths, int B, int S,
                                                        int n_prefix, float scale) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + wave;
    if (row >= (long)B * S) return;
    const int b = (int)(row / S);
    const int n = min(S, n_prefix + lengths[b]);
    float* p = s + row * S;
    float v[8];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = lane + 64 * j;
        v[j] = c < n ? p[c] * scale : -INFINITY;
        mx = fmaxf(mx, v[j]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { v[j] = (lane + 64 * j) < n ? expf(v[j] - mx) : 0.f; sum += v[j]; }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int c = lane + 64 * j; if (c < S) p[c] = v[j] * inv; }
}
// dP (in place) -> dS = scale * P * (dP - sum_k dP_k P_k)
__global__ __launch_bounds__(256) void k_vt_softmax_bwd(float* __restrict__ dp, const float* __restrict__ p, long rows, int S, float scale) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + wave;
    if (row >= rows) return;
    float* d = dp + row * S;
    const float* pr = p + row * S;
    float dv[8], pv[8], dot = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = lane + 64 * j;
        dv[j] = c < S ? d[c] : 0.f;
        pv[j] = c < S ? pr[c] : 0.f;
        dot += dv[j] * pv[j];
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int c = lane + 64 * j; if (c < S) d[c] = scale * pv[j] * (dv[j] - dot); }
}

// exact GELU (torch default): h = 0.5 x (1 + erf(x / sqrt 2)) ; backward dpre = dh * (Phi(x) + x phi(x))
__global__ void k_vt_gelu_fwd(const float* __restrict__ pre, float* __restrict__ h, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = pre[i];
    h[i] = 0.5f * x * (1.f + erff(x * 0.70710678118654752440f));
}
__global__ void k_vt_gelu_bwd(const float* __restrict__ dh, const float* __restrict__ pre, float* __restrict__ dpre, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = pre[i];
    const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
    dpre[i] = dh[i] * (cdf + x * pdf);
}

// out[b][c] (+)= sum_s w[b][s] d[b][s][c]   (gradient of a per-sequence vector that was broadcast over the rows; w = the
// dropped-out single-key attention weight mask[b][s] * scale, or 1 without dropout)
__global__ __launch_bounds__(256) void k_vt_seq_sum(const float* __restrict__ d, float* __restrict__ out, int S, int accumulate,
                                                    const unsigned char* __restrict__ wmask, float scale) {
    const int b = blockIdx.x, c = threadIdx.x;
    const float* p = d + (size_t)b * S * 256 + c;
    float s = 0.f;
    for (int i = 0; i < S; ++i) s += (wmask ? (wmask[(size_t)b * S + i] ? scale : 0.f) : 1.f) * p[(size_t)i * 256];
    out[(size_t)b * 256 + c] = accumulate ? out[(size_t)b * 256 + c] + s : s;
}

// inverted dropout with a given keep-mask: out = x * mask * scale (in place allowed); x / out 16-byte, mask 4-byte aligned
__global__ void k_vt_dropout(const float* __restrict__ x, const unsigned char* __restrict__ m, float scale, float* __restrict__ out, long n) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    if (i + 3 < n) {
        const float4 v = *reinterpret_cast<const float4*>(x + i);
        const uchar4 k = *reinterpret_cast<const uchar4*>(m + i);
        *reinterpret_cast<float4*>(out + i) = make_float4(k.x ? v.x * scale : 0.f, k.y ? v.y * scale : 0.f, k.z ? v.z * scale : 0.f, k.w ? v.w * scale : 0.f);
    } else {
        for (long j = i; j < n; ++j) out[j] = m[j] ? x[j] * scale : 0.f;
    }
}

// The decoder's cross-attention to the single latent token WITH dropout (cross_attention.py:357-362): the softmax over one key
// is 1, attention dropout turns it into w[b,s] in {0, scale}, dropout2 masks the projected output:
//   out[b,s,:] = (w[b,s] * cvn[b,:] + bo) * m2[b,s,:] * scale,   cvn = out_proj.weight (W_v z + b_v)
__global__ __launch_bounds__(256) void k_vt_cross_rows(const float* __restrict__ cvn, const float* __restrict__ bo,
                                                       const unsigned char* __restrict__ wmask, const unsigned char* __restrict__ m2,
                                                       float scale, int S, float* __restrict__ out) {
    const size_t row = blockIdx.x;
    const int c = threadIdx.x;
    const size_t b = row / S;
    const float w = wmask[row] ? scale : 0.f;
    out[row * 256 + c] = m2[row * 256 + c] ? (w * cvn[b * 256 + c] + bo[c]) * scale : 0.f;
}

// ------------------------------------------------------------------ C-ABI
extern "C" int seeme_vt_add_ln(const SeemeVtLn* a, void* stream) {
    if (!a || a->M < 1 || !a->sub || !a->gamma || !a->beta || !a->y || !a->xhat || !a->rstd) return seeme_fail("seeme_vt_add_ln: bad arguments");
    hipLaunchKernelGGL(k_vt_add_ln, dim3((unsigned)((a->M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, *a);
    return seeme_check_launch("k_vt_add_ln");
}
extern "C" int seeme_vt_ln_bwd(const SeemeVtLnBwd* a, void* stream) {
    if (!a || a->M < 1 || !a->dy || !a->xhat || !a->rstd || !a->gamma || !a->dpre || !a->dgamma || !a->dbeta) return seeme_fail("seeme_vt_ln_bwd: bad arguments");
    hipLaunchKernelGGL(k_vt_ln_bwd, dim3((unsigned)((a->M + 31) / 32)), dim3(256), 0, (hipStream_t)stream, *a);
    return seeme_check_launch("k_vt_ln_bwd");
}
extern "C" int seeme_vt_softmax_fwd(float* scores, const int32_t* lengths, int B, int S, int n_prefix, float scale, void* stream) {
    if (!scores || !lengths || B < 1 || S < 1 || S > 512) return seeme_fail("seeme_vt_softmax_fwd: S must be in 1..512");
    hipLaunchKernelGGL(k_vt_softmax_fwd, dim3((unsigned)(((long)B * S + 3) / 4)), dim3(256), 0, (hipStream_t)stream, scores, lengths, B, S, n_prefix, scale);
    return seeme_check_launch("k_vt_softmax_fwd");
}
extern "C" int seeme_vt_softmax_bwd(float* dp, const float* p, long rows, int S, float scale, void* stream) {
    if (!dp || !p || rows < 1 || S < 1 || S > 512) return seeme_fail("seeme_vt_softmax_bwd: S must be in 1..512");
    hipLaunchKernelGGL(k_vt_softmax_bwd, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, dp, p, rows, S, scale);
    return seeme_check_launch("k_vt_softmax_bwd");
}
extern "C" int seeme_vt_gelu(const float* pre, const float* dh, float* out, long n, void* stream) {
    if (!pre || !out || n < 1) return seeme_fail("seeme_vt_gelu: bad arguments");
    if (dh == nullptr) hipLaunchKernelGGL(k_vt_gelu_fwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pre, out, n);
    else hipLaunchKernelGGL(k_vt_gelu_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dh, pre, out, n);
    return seeme_check_launch("k_vt_gelu");
}
extern "C" int seeme_vt_seq_sum(const float* d, float* out, int B, int S, int accumulate, const unsigned char* wmask, float scale, void* stream) {
    if (!d || !out || B < 1 || S < 1) return seeme_fail("seeme_vt_seq_sum: bad arguments");
    hipLaunchKernelGGL(k_vt_seq_sum, dim3(B), dim3(256), 0, (hipStream_t)stream, d, out, S, accumulate, wmask, scale);
    return seeme_check_launch("k_vt_seq_sum");
}
extern "C" int seeme_vt_dropout(const float* x, const unsigned char* mask, float scale, float* out, long n, void* stream) {
    if (!x || !mask || !out || n < 1 || (reinterpret_cast<size_t>(mask) & 3) || (reinterpret_cast<size_t>(x) & 15) || (reinterpret_cast<size_t>(out) & 15))
        return seeme_fail("seeme_vt_dropout: n >= 1, x / out 16-byte aligned, mask 4-byte aligned");
    hipLaunchKernelGGL(k_vt_dropout, dim3((unsigned)(((n + 3) / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, mask, scale, out, n);
    return seeme_check_launch("k_vt_dropout");
}
extern "C" int seeme_vt_cross_rows(const float* cvn, const float* bo, const unsigned char* wmask, const unsigned char* m2, float scale,
                                   int B, int S, float* out, void* stream) {
    if (!cvn || !bo || !wmask || !m2 || !out || B < 1 || S < 1) return seeme_fail("seeme_vt_cross_rows: bad arguments");
    hipLaunchKernelGGL(k_vt_cross_rows, dim3((unsigned)((size_t)B * S)), dim3(256), 0, (hipStream_t)stream, cvn, bo, wmask, m2, scale, S, out);
    return seeme_check_launch("k_vt_cross_rows");
}
