// vae_kernels.hip -- row-tile kernels (fused linear, attention block, FFN block) and the host-side
// sequencing of MldVae.encode / MldVae.decode on one HIP stream.
//
// Everything is fp32 with v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains), so results track the
// reference's fp32 PyTorch path to ~1e-6.  One workgroup = 256 threads = 4 waves owns a tile of
// TILE_M = 32 token rows; a sequence's rows are contiguous ([B][S][256], batch-major).
#include "common.hpp"
#include "api_util.hpp"

// ---------------------------------------------------------------------------------------------
// k_linear: Y = LN?( act( pre(A|A2) W^T + bias ) + res ), one 32-row x 256-col tile per workgroup.


__global__ __launch_bounds__(256) void k_linear(const LinearKArgs ka) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    SeemeLinearArgs a = ka.a;
    if (ka.nz > 1) {   // batch of independent problems over blockIdx.z
        const long z = blockIdx.z;
        a.A += z * ka.zs_a; a.W += z * ka.zs_w; a.Y += z * ka.zs_y;
        if (a.bias) a.bias += z * ka.zs_b;
        if (a.pre_ln_w) { a.pre_ln_w += z * ka.zs_ln; a.pre_ln_b += z * ka.zs_ln; }
        if (z < ka.pre_ln_zmin) { a.pre_ln_w = nullptr; a.pre_ln_b = nullptr; }
    }
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int Kp = (a.K + 15) & ~15;
    const int lda_s = Kp + LDS_PAD;
    float* As = smem;                       // [32][Kp+8]
    float* Cs = smem + TILE_M * lda_s;      // [32][256+8]
    const int ldc = CH_N + LDS_PAD;
    const int m0 = blockIdx.x * TILE_M;
    const int cn0 = blockIdx.y * (ka.narrow ? 64 : CH_N);

    // ---- stage A tile (zero padded), optional pre-LN and pre-activation
    const bool a_vec = (a.A2 == nullptr) && ((a.K & 3) == 0) && ((a.lda & 3) == 0) && ((reinterpret_cast<size_t>(a.A) & 15) == 0);
    if (a_vec) {   // float4 path: K/4 vectors per row, rows spread over the 256 threads
        // batches of 8 guarded loads per thread, all issued before the first is stored (guard = select on address and
        // value; a branch per load made every load a full round trip)
        const int K4 = a.K >> 2, Kp4 = Kp >> 2, total = TILE_M * Kp4;
        for (int base = 0; base < total; base += 256 * 8) {
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int idx = base + tid + j * 256, row = idx / Kp4, c4 = idx - row * Kp4, m = m0 + row;
                const bool valid = idx < total && m < a.M && c4 < K4;
                int prow = valid ? m : 0;
                if (ka.seq_in > 0) prow = (prow / ka.seq_in) * ka.in_stride + (prow % ka.seq_in) + ka.in_off;
                v[j] = *reinterpret_cast<const float4*>(valid ? a.A + (size_t)prow * a.lda + 4 * c4 : a.A);
                if (!valid) v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int idx = base + tid + j * 256, row = idx / Kp4, c4 = idx - row * Kp4;
                if (idx < total) *reinterpret_cast<float4*>(As + row * lda_s + 4 * c4) = v[j];
            }
        }
    } else {
        for (int idx = tid; idx < TILE_M * Kp; idx += 256) {
            const int row = idx / Kp, c = idx - row * Kp;
            const int m = m0 + row;
            float v = 0.f;
            if (m < a.M && c < a.K) {
                int prow = m;
                if (ka.seq_in > 0) prow = (m / ka.seq_in) * ka.in_stride + (m % ka.seq_in) + ka.in_off;
                v = (c < a.K1) ? a.A[(size_t)prow * a.lda + c] : a.A2[(size_t)prow * a.lda2 + (c - a.K1)];
            }
            As[row * lda_s + c] = v;
        }
    }
    __syncthreads();
    if (a.pre_ln_w != nullptr) {  // LayerNorm over K (any K), one wave per 8 rows
        for (int rr = 0; rr < 8; ++rr) {
            const int row = wave * 8 + rr;
            float s = 0.f;
            for (int c = lane; c < a.K; c += 64) s += As[row * lda_s + c];
            const float mean = wave_sum(s) / (float)a.K;
            float q = 0.f;
            for (int c = lane; c < a.K; c += 64) { float d = As[row * lda_s + c] - mean; q += d * d; }
            const float rs = 1.f / sqrtf(wave_sum(q) / (float)a.K + a.eps);
            for (int c = lane; c < a.K; c += 64)
                As[row * lda_s + c] = (As[row * lda_s + c] - mean) * rs * a.pre_ln_w[c] + a.pre_ln_b[c];
        }
        __syncthreads();
    }
    if (a.pre_act != SEEME_ACT_NONE) {
        for (int idx = tid; idx < TILE_M * Kp; idx += 256) {
            const int row = idx / Kp, c = idx - row * Kp;
            if (c < a.K) As[row * lda_s + c] = act_apply(As[row * lda_s + c], a.pre_act);
        }
        __syncthreads();
    }

    // ---- GEMM: wave w owns tile columns [64w, 64w+64)
    if (ka.narrow) {    // one n-tile per wave: a 32-row problem with thousands of columns is matrix-core time on N / 256 CUs otherwise
        f32x4 acc[2][1];
        acc_zero(acc);
        const int n0 = cn0 + wave * 16;
        if (n0 < a.N) tile_gemm_f32<2, 1>(As, lda_s, a.W, a.ldw, n0, a.N, Kp >> 4, acc);
        acc_store_lds<2, 1>(acc, Cs, ldc, wave * 16, a.bias, cn0, a.N, a.act);
    } else {
        f32x4 acc[2][4];
        acc_zero(acc);
        const int n0 = cn0 + wave * 64;
        if (n0 < a.N) tile_gemm_f32<2, 4>(As, lda_s, a.W, a.ldw, n0, a.N, Kp >> 4, acc);
        acc_store_lds<2, 4>(acc, Cs, ldc, wave * 64, a.bias, cn0, a.N, a.act);
    }
    __syncthreads();

    // ---- row pass: residual, LayerNorm, coalesced store.  wave w -> rows 8w..8w+7, lane -> 4 cols
    const bool vec_ok = ((a.ldy & 3) == 0) && ((a.N & 3) == 0);
    // residual rows as float4, all 8 of the wave requested together (when rows are 16-byte aligned and whole); LayerNorm
    // parameters once per phase
    const bool res_vec = a.res != nullptr && ((a.ldr & 3) == 0) && ((a.N & 3) == 0) && ((reinterpret_cast<size_t>(a.res) & 15) == 0);
    const LnParams lp = ln_params256(a.ln_w, a.ln_b);
    float4 rv[8];
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        rv[rr] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (res_vec) {
            const int m = m0 + wave * 8 + rr, mc = m < a.M ? m : 0, g = (ka.narrow && lane >= 16) ? a.N : cn0 + lane * 4;
            int orow = mc;
            if (ka.seq_in > 0) orow = (mc / ka.seq_in) * ka.out_stride + (mc % ka.seq_in) + ka.out_off;
            const size_t rrow = ka.res_mode == 1 ? (size_t)((mc % ka.seq_in) + ka.res_off)
                              : ka.res_mode == 2 ? (size_t)(mc / ka.seq_in) : (size_t)orow;
            rv[rr] = *reinterpret_cast<const float4*>(a.res + rrow * a.ldr + (g + 3 < a.N ? g : 0));
            if (g + 3 >= a.N) rv[rr] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        const int row = wave * 8 + rr;
        const int m = m0 + row;
        if (m >= a.M) continue;  // wave-uniform
        int orow = m;
        if (ka.seq_in > 0) orow = (m / ka.seq_in) * ka.out_stride + (m % ka.seq_in) + ka.out_off;
        const int c = (ka.narrow && lane >= 16) ? 0 : lane * 4, g = (ka.narrow && lane >= 16) ? a.N : cn0 + c;
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
        if (a.ln_w != nullptr) v = wave_layernorm256(v, lp, a.eps);  // N == 256, cn0 == 0
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
}

#ifndef SEEME_NARROW_MIN_N
#define SEEME_NARROW_MIN_N 512
#endif
#ifndef SEEME_NARROW_MAX
#define SEEME_NARROW_MAX 64
#endif
int seeme_launch_linear(const LinearKArgs& ka, hipStream_t st) {
    const SeemeLinearArgs& a = ka.a;
    if (a.M <= 0 || a.N <= 0 || a.K <= 0) return seeme_fail("seeme_linear: empty problem");
    if (a.ln_w && a.N != 256) return seeme_fail("seeme_linear: fused LayerNorm needs N == 256");
    if (a.A2 == nullptr && a.K1 != a.K) return seeme_fail("seeme_linear: K1 must equal K without A2");
    const int Kp = (a.K + 15) & ~15;
    if (a.ldw < Kp) return seeme_fail("seeme_linear: ldw < roundup16(K) (weights must be zero padded)");
    if ((a.ldw & 3) != 0) return seeme_fail("seeme_linear: ldw must be a multiple of 4");
    if (Kp > 1024) return seeme_fail("seeme_linear: K > 1024 not supported");
    const size_t lds = (size_t)(TILE_M * (Kp + LDS_PAD) + TILE_M * (CH_N + LDS_PAD)) * sizeof(float);
    LinearKArgs k2 = ka;
    const long wide = (long)((a.M + TILE_M - 1) / TILE_M) * ((a.N + CH_N - 1) / CH_N) * (ka.nz > 1 ? ka.nz : 1);
    k2.narrow = (a.ln_w == nullptr && a.N >= SEEME_NARROW_MIN_N && wide <= SEEME_NARROW_MAX) ? 1 : 0;     // too few 256-column workgroups to fill 256 CUs
    const int cw = k2.narrow ? 64 : CH_N;
    dim3 grid((a.M + TILE_M - 1) / TILE_M, (a.N + cw - 1) / cw, ka.nz > 1 ? ka.nz : 1);
    SEEME_HIP(hipFuncSetAttribute((const void*)k_linear, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_linear, grid, dim3(256), lds, st, k2);
    return seeme_check_launch("k_linear");
}

extern "C" int seeme_linear(const SeemeLinearArgs* args, void* stream) {
    LinearKArgs ka{};
    ka.a = *args;
    if (ka.a.A2 == nullptr) ka.a.K1 = ka.a.K;
    return seeme_launch_linear(ka, (hipStream_t)stream);
}

// convenience for the host sequencers
static int linear_simple(hipStream_t st, const float* A, int lda, const float* W, int ldw, const float* bias,
                         float* Y, int ldy, int M, int N, int K, int act = SEEME_ACT_NONE,
                         int pre_act = SEEME_ACT_NONE, const float* pre_ln_w = nullptr,
                         const float* pre_ln_b = nullptr) {
    LinearKArgs ka{};
    ka.a.A = A; ka.a.lda = lda; ka.a.K1 = K; ka.a.W = W; ka.a.ldw = ldw; ka.a.bias = bias;
    ka.a.Y = Y; ka.a.ldy = ldy; ka.a.M = M; ka.a.N = N; ka.a.K = K; ka.a.act = act; ka.a.pre_act = pre_act;
    ka.a.pre_ln_w = pre_ln_w; ka.a.pre_ln_b = pre_ln_b; ka.a.eps = 1e-5f;
    return seeme_launch_linear(ka, st);
}
int seeme_linear_simple(hipStream_t st, const float* A, int lda, const float* W, int ldw, const float* bias,
                        float* Y, int ldy, int M, int N, int K, int act, int pre_act,
                        const float* pre_ln_w, const float* pre_ln_b) {
    return linear_simple(st, A, lda, W, ldw, bias, Y, ldy, M, N, K, act, pre_act, pre_ln_w, pre_ln_b);
}

// ---------------------------------------------------------------------------------------------
// k_attn_block: single-head (head_dim 256) self-attention for one 32-query tile of one sequence,
// fused with out_proj + residual + LayerNorm:   out = LN(res + Wo softmax(QK^T/16 + mask) V + bo)
// (nn.MultiheadAttention + norm1, cross_attention.py:286-290 / :353-357).
struct AttnKArgs {
    const float* qkv;      // [B*S][768]  q | k | v
    const float* res;      // [B*S][256]
    const float* wo; const float* bo; const float* ln_w; const float* ln_b;
    float* out;            // [B*S][256]
    const int32_t* lengths;  // [B] valid frames
    int S;                 // tokens per sequence
    int n_prefix;          // always-valid leading tokens (2 distribution tokens in the encoder)
    int q_rows;            // query rows processed per sequence (S, or 2 for the encoder's last layer)
    int Sp;                // S rounded up to 256 (score tile width)
    float scale;           // 1/sqrt(head_dim)
    float eps;
};

template <int NC>      // NC = Sp / 64 score columns per lane
__device__ __forceinline__ void attn_softmax_rows_f32(float* Ps, int ldp, int wave, int lane) {
    float v[8][NC], mx[8], sum[8];
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        const float* prow = Ps + (wave * 8 + rr) * ldp + lane;
        mx[rr] = -INFINITY;
#pragma unroll
        for (int j = 0; j < NC; ++j) { v[rr][j] = prow[64 * j]; mx[rr] = fmaxf(mx[rr], v[rr][j]); }
    }
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) mx[rr] = wave_max(mx[rr]);
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        sum[rr] = 0.f;
#pragma unroll
        for (int j = 0; j < NC; ++j) { v[rr][j] = expf(v[rr][j] - mx[rr]); sum[rr] += v[rr][j]; }
    }
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) sum[rr] = wave_sum(sum[rr]);
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        const float inv = 1.f / sum[rr];
        float* prow = Ps + (wave * 8 + rr) * ldp + lane;
#pragma unroll
        for (int j = 0; j < NC; ++j) prow[64 * j] = v[rr][j] * inv;
    }
}

__global__ __launch_bounds__(256) void k_attn_block(const AttnKArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 15, kq = lane >> 4;
    const int ldq = 256 + LDS_PAD, ldp = a.Sp + LDS_PAD;
    float* Qs = smem;                  // [32][264]  Q tile, later the O tile
    float* Ps = smem + TILE_M * ldq;   // [32][Sp+8] scores / probabilities, later the out_proj tile
    // XCD-aware mapping: blocks L and L+8 share an XCD (and its L2); all query tiles of one sequence re-read
    // that sequence's K/V, so they are placed on the same XCD (speed only -- any placement is correct)
    int b = blockIdx.y, qt = blockIdx.x;
    if ((gridDim.y & 7) == 0) {
        const int L = blockIdx.x + gridDim.x * blockIdx.y, r = L & 7, q = L >> 3;
        b = r + 8 * (q / (int)gridDim.x);
        qt = q % (int)gridDim.x;
    }
    const int q0 = qt * TILE_M;
    const size_t base = (size_t)b * a.S;
    const int n_valid_keys = min(a.S, a.n_prefix + a.lengths[b]);

    {   // Q tile, 8 float4 per thread requested together (the guard is a select on address and value: written as a branch
        // per load the compiler waits for every load before issuing the next -- eight serialised round trips)
        float4 qv[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = tid + it * 256, row = idx >> 6, c4 = (idx & 63) * 4, s = q0 + row;
            qv[it] = *reinterpret_cast<const float4*>(a.qkv + (base + (s < a.q_rows ? s : 0)) * 768 + c4);
            if (s >= a.q_rows) qv[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = tid + it * 256, row = idx >> 6, c4 = (idx & 63) * 4;
            *reinterpret_cast<float4*>(Qs + row * ldq + c4) = qv[it];
        }
    }
    __syncthreads();

    // ---- scores = scale * Q K^T, masked
    const float* Kmat = a.qkv + base * 768 + 256;
    for (int c0 = 0; c0 < a.Sp; c0 += CH_N) {
        const int n0 = c0 + wave * 64;
        f32x4 acc[2][4];
        acc_zero(acc);
        if (n0 < a.S) tile_gemm_f32<2, 4>(Qs, ldq, Kmat, 768, n0, a.S, 16, acc);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int c = n0 + nt * 16 + r;
            const bool ok = c < n_valid_keys;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    Ps[(mt * 16 + 4 * kq + i) * ldp + c] = ok ? acc[mt][nt][i] * a.scale : -INFINITY;
        }
    }
    __syncthreads();

    // ---- softmax over keys, wave w -> rows 8w..8w+7
    // (the wave's 8 rows are held in registers and reduced together: same per-row arithmetic and summation order as
    // the row-by-row loop, without its 16 dependent reductions in sequence)
    if (a.Sp == 256) attn_softmax_rows_f32<4>(Ps, ldp, wave, lane);
    else attn_softmax_rows_f32<8>(Ps, ldp, wave, lane);
    __syncthreads();

    // ---- O = P V   (contraction over keys; P is exactly 0 beyond n_valid_keys)
    {
        f32x4 acc[2][4];
        acc_zero(acc);
        const float* Vmat = a.qkv + base * 768 + 512;
        const int K16 = (n_valid_keys + 15) >> 4;
        tile_gemm_f32_kn<2, 4>(Ps, ldp, Vmat, 768, wave * 64, a.S, K16, acc);
        acc_store_lds<2, 4>(acc, Qs, ldq, wave * 64, nullptr, 0, 256, SEEME_ACT_NONE);
    }
    __syncthreads();

    // ---- out_proj into the (now free) Ps region, then residual + LayerNorm
    float* Cs = Ps;
    const int ldc = 256 + LDS_PAD;
    {
        f32x4 acc[2][4];
        acc_zero(acc);
        tile_gemm_f32<2, 4>(Qs, ldq, a.wo, 256, wave * 64, 256, 16, acc);
        __syncthreads();  // every wave finished reading Ps as probabilities long ago; Qs reads done here
        acc_store_lds<2, 4>(acc, Cs, ldc, wave * 64, a.bo, 0, 256, SEEME_ACT_NONE);
    }
    __syncthreads();
    {   // residual rows and LayerNorm parameters requested together, then one pass of LN + store per row
        const LnParams lp = ln_params256(a.ln_w, a.ln_b);
        float4 xr[8];
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const int s = q0 + wave * 8 + rr;
            xr[rr] = *reinterpret_cast<const float4*>(a.res + (base + (s < a.q_rows ? s : 0)) * 256 + lane * 4);
        }
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
}

static int launch_attn(const AttnKArgs& a_in, int B, hipStream_t st) {
    AttnKArgs a = a_in;
    if (a.S <= 0 || a.S > 512) return seeme_fail("attention: S must be in 1..512 (learned PE has 500 rows)");
    a.Sp = (a.S + CH_N - 1) / CH_N * CH_N;
    const size_t lds = (size_t)(TILE_M * (256 + LDS_PAD) + TILE_M * (a.Sp + LDS_PAD)) * sizeof(float);
    dim3 grid((a.q_rows + TILE_M - 1) / TILE_M, B);
    SEEME_HIP(hipFuncSetAttribute((const void*)k_attn_block, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_attn_block, grid, dim3(256), lds, st, a);
    return seeme_check_launch("k_attn_block");
}

// ---------------------------------------------------------------------------------------------
// k_ffn_block: out = LN( x' + W2 act(W1 x' + b1) + b2 ),  x' = x  or  LNc(x + cvec[seq])
// (cross_attention.py:291-293; decoder: the 1-token cross-attention collapses to a per-sequence
//  vector, SURVEY.md App. E1, so :358-366 is the prologue here).
struct FfnKArgs {
    const float* x; float* out;     // [rows][256]
    const float* w1; const float* b1; const float* w2; const float* b2;
    const float* ln_w; const float* ln_b;
    const float* cvec;              // optional cross-attention vector per sequence, row stride cvec_ld
    int cvec_ld;
    const float* lnc_w; const float* lnc_b;
    const float* fin_w; const float* fin_b;   // optional extra LayerNorm after the block (stack norm)
    int M;            // logical rows
    int FF;           // hidden width (multiple of 128)
    int act;
    int seq_rows;     // logical rows per sequence
    int seq_stride;   // physical rows per sequence
    int out_mode;     // 0: same physical row; 1: out row = (m % seq_rows) * (M/seq_rows) + m / seq_rows
    float eps;
};

__global__ __launch_bounds__(256) void k_ffn_block(const FfnKArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int ld = 256 + LDS_PAD;
    float* Xs = smem;                 // [32][264] block input (also the residual)
    float* Hs = smem + TILE_M * ld;   // [32][264] hidden chunk, later the output tile
    const int m0 = blockIdx.x * TILE_M;

    {   // wave w stages rows 8w..8w+7 (one float4 per lane): rows and cross-attention vectors requested together
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
        }
    }
    __syncthreads();

    f32x4 acc2[2][4];
    acc_zero(acc2);
    const int FFC = a.FF < 256 ? a.FF : 256;  // hidden columns per chunk
    for (int c0 = 0; c0 < a.FF; c0 += FFC) {
        if (FFC == 256) {
            f32x4 acc1[2][4];
            acc_zero(acc1);
            tile_gemm_f32<2, 4>(Xs, ld, a.w1 + (size_t)c0 * 256, 256, wave * 64, FFC, 16, acc1);
            acc_store_lds<2, 4>(acc1, Hs, ld, wave * 64, a.b1 + c0, 0, FFC, a.act);
        } else {  // FF == 128: 32 hidden columns per wave
            f32x4 acc1[2][2];
            acc_zero(acc1);
            tile_gemm_f32<2, 2>(Xs, ld, a.w1 + (size_t)c0 * 256, 256, wave * 32, FFC, 16, acc1);
            acc_store_lds<2, 2>(acc1, Hs, ld, wave * 32, a.b1 + c0, 0, FFC, a.act);
        }
        __syncthreads();
        tile_gemm_f32<2, 4>(Hs, ld, a.w2 + c0, a.FF, wave * 64, 256, FFC >> 4, acc2);
        __syncthreads();
    }
    acc_store_lds<2, 4>(acc2, Hs, ld, wave * 64, a.b2, 0, 256, SEEME_ACT_NONE);
    __syncthreads();

    const LnParams lp = ln_params256(a.ln_w, a.ln_b), lf = ln_params256(a.fin_w, a.fin_b);
    for (int rr = 0; rr < 8; ++rr) {
        const int row = wave * 8 + rr, m = m0 + row;
        if (m >= a.M) continue;
        float4 v = *reinterpret_cast<const float4*>(Hs + row * ld + lane * 4);
        const float4 x = *reinterpret_cast<const float4*>(Xs + row * ld + lane * 4);
        v = make_float4(v.x + x.x, v.y + x.y, v.z + x.z, v.w + x.w);
        v = wave_layernorm256(v, lp, a.eps);
        if (a.fin_w != nullptr) v = wave_layernorm256(v, lf, a.eps);
        const int seq = m / a.seq_rows, sr = m % a.seq_rows;
        size_t orow = (size_t)seq * a.seq_stride + sr;
        if (a.out_mode == 1) orow = (size_t)sr * (a.M / a.seq_rows) + seq;
        *reinterpret_cast<float4*>(a.out + orow * 256 + lane * 4) = v;
    }
}

static int launch_ffn(const FfnKArgs& a, hipStream_t st) {
    if (a.FF != 128 && (a.FF % 256) != 0) return seeme_fail("ffn: FF must be 128 or a multiple of 256");
    const size_t lds = (size_t)(2 * TILE_M * (256 + LDS_PAD)) * sizeof(float);
    dim3 grid((a.M + TILE_M - 1) / TILE_M);
    SEEME_HIP(hipFuncSetAttribute((const void*)k_ffn_block, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_ffn_block, grid, dim3(256), lds, st, a);
    return seeme_check_launch("k_ffn_block");
}

// ---------------------------------------------------------------------------------------------
// small fills
__global__ void k_enc_tokens(const float* __restrict__ token, const float* __restrict__ pe, float* __restrict__ x,
                             int B, int S) {
    // x[b][i] = global_motion_token[i] + pe[i], i = 0,1   (mld_vae.py:154,164,171)
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * 2 * 256) return;
    const int d = idx & 255, i = (idx >> 8) & 1, b = idx >> 9;
    x[((size_t)b * S + i) * 256 + d] = token[i * 256 + d] + pe[i * 256 + d];
}
__global__ void k_bcast_rows(const float* __restrict__ src, float* __restrict__ dst, int B, int rows) {
    // dst[b][t] = src[t]   (decoder queries = zeros + pe[:T], mld_vae.py:198,232)
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // float4 index
    const size_t per = (size_t)rows * 64;
    if (idx >= per * B) return;
    reinterpret_cast<float4*>(dst)[idx] = reinterpret_cast<const float4*>(src)[idx % per];
}

// ---------------------------------------------------------------------------------------------
// host sequencing
struct VaeWs {
    float *x, *y, *sk0, *sk1, *qkv, *cvec, *tmp;
};
static size_t vae_ws_floats(int B, int S) {
    const size_t R = (size_t)B * S;
    // + the fp16 path's two sets of fragment-packed keys (vae_h16.hip): 2 x B x ceil16(S) x 256 halves
    return R * 256 * 4 + R * 768 + (size_t)B * 256 * (SEEME_NLAYERS + 1) + 64 + (size_t)B * ((S + 15) & ~15) * 256;
}
extern "C" size_t seeme_vae_workspace_bytes(int B, int T) { return vae_ws_floats(B, T + 2) * sizeof(float); }

static VaeWs carve(void* ws, int B, int S) {
    const size_t R = (size_t)B * S;
    float* p = (float*)ws;
    VaeWs w;
    w.x = p; p += R * 256;
    w.y = p; p += R * 256;
    w.sk0 = p; p += R * 256;
    w.sk1 = p; p += R * 256;
    w.qkv = p; p += R * 768;
    w.cvec = p; p += (size_t)B * 256 * SEEME_NLAYERS;
    w.tmp = p;
    return w;
}

// One post-norm layer of a skip stack on `rows` = B*S token rows.
//   cur (in) -> out;  decoder layers additionally take the per-sequence cross vector.
static int run_layer(hipStream_t st, const SeemeXfLayer& L, int ff, int act, float* cur, float* mid, float* out,
                     float* qkv, const int32_t* lengths, int B, int S, int n_prefix, int q_rows,
                     const float* cvec, const float* fin_w, const float* fin_b, int out_mode, int cvec_ld = 256) {
    int rc = linear_simple(st, cur, 256, L.in_w, 256, L.in_b, qkv, 768, B * S, 768, 256);
    if (rc) return rc;
    AttnKArgs at{};
    at.qkv = qkv; at.res = cur; at.wo = L.out_w; at.bo = L.out_b; at.ln_w = L.n1_w; at.ln_b = L.n1_b;
    at.out = mid; at.lengths = lengths; at.S = S; at.n_prefix = n_prefix; at.q_rows = q_rows;
    at.scale = 1.0f / 16.0f; at.eps = 1e-5f;
    rc = launch_attn(at, B, st);
    if (rc) return rc;
    FfnKArgs f{};
    f.x = mid; f.out = out; f.w1 = L.l1_w; f.b1 = L.l1_b; f.w2 = L.l2_w; f.b2 = L.l2_b;
    f.M = B * q_rows; f.FF = ff; f.act = act; f.seq_rows = q_rows; f.seq_stride = S; f.out_mode = out_mode;
    f.eps = 1e-5f; f.fin_w = fin_w; f.fin_b = fin_b;
    if (cvec != nullptr) {
        f.cvec = cvec; f.cvec_ld = cvec_ld; f.lnc_w = L.n2_w; f.lnc_b = L.n2_b; f.ln_w = L.n3_w; f.ln_b = L.n3_b;
    } else {
        f.ln_w = L.n2_w; f.ln_b = L.n2_b;
    }
    return launch_ffn(f, st);
}

static int skip_linear(hipStream_t st, const float* a1, const float* a2, const float* w, const float* b, float* y, int M) {
    LinearKArgs ka{};
    ka.a.A = a1; ka.a.lda = 256; ka.a.A2 = a2; ka.a.lda2 = 256; ka.a.K1 = 256; ka.a.K = 512;
    ka.a.W = w; ka.a.ldw = 512; ka.a.bias = b; ka.a.Y = y; ka.a.ldy = 256; ka.a.M = M; ka.a.N = 256; ka.a.eps = 1e-5f;
    return seeme_launch_linear(ka, st);
}

extern "C" int seeme_vae_encode(const SeemeVaeWeights* w, const float* features, const int32_t* lengths,
                                int B, int T, float* mu, float* logvar, void* workspace, size_t ws_bytes,
                                void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (B <= 0 || T <= 0 || T + 2 > 500) return seeme_fail("vae_encode: need B>0 and 0 < T <= 498");
    if (mu + (size_t)B * 256 != logvar) return seeme_fail("vae_encode: mu/logvar must be one [2,B,256] buffer");
    const int S = T + 2, F = w->nfeats;
    if (ws_bytes < seeme_vae_workspace_bytes(B, T)) return seeme_fail("vae_encode: workspace too small");
    if (w->h16 != nullptr) return seeme_vae_encode_h16(w, features, lengths, B, T, mu, workspace, st);
    VaeWs ws = carve(workspace, B, S);
    const SeemeSkipStack& E = w->enc;

    // tokens + frame embedding + learned PE  (mld_vae.py:147-171)
    hipLaunchKernelGGL(k_enc_tokens, dim3((B * 512 + 255) / 256), dim3(256), 0, st, w->token, w->pe_enc, ws.x, B, S);
    int rc = seeme_check_launch("k_enc_tokens");
    if (rc) return rc;
    {
        LinearKArgs ka{};
        ka.a.A = features; ka.a.lda = F; ka.a.K1 = F; ka.a.K = F; ka.a.W = w->emb_w; ka.a.ldw = w->emb_ldw;
        ka.a.bias = w->emb_b; ka.a.res = w->pe_enc; ka.a.ldr = 256; ka.a.Y = ws.x; ka.a.ldy = 256;
        ka.a.M = B * T; ka.a.N = 256; ka.a.eps = 1e-5f;
        ka.seq_in = T; ka.in_stride = T; ka.in_off = 0; ka.out_stride = S; ka.out_off = 2;
        ka.res_mode = 1; ka.res_off = 2;
        rc = seeme_launch_linear(ka, st);
        if (rc) return rc;
    }
    const int ff = w->ff, act = SEEME_ACT_GELU;
    // SkipTransformerEncoder.forward (cross_attention.py:46-65)
    if ((rc = run_layer(st, E.layer[0], ff, act, ws.x, ws.y, ws.sk0, ws.qkv, lengths, B, S, 2, S, nullptr, nullptr, nullptr, 0))) return rc;
    if ((rc = run_layer(st, E.layer[1], ff, act, ws.sk0, ws.y, ws.sk1, ws.qkv, lengths, B, S, 2, S, nullptr, nullptr, nullptr, 0))) return rc;
    if ((rc = run_layer(st, E.layer[2], ff, act, ws.sk1, ws.y, ws.x, ws.qkv, lengths, B, S, 2, S, nullptr, nullptr, nullptr, 0))) return rc;
    if ((rc = skip_linear(st, ws.x, ws.sk1, E.skip_w[0], E.skip_b[0], ws.x, B * S))) return rc;
    if ((rc = run_layer(st, E.layer[3], ff, act, ws.x, ws.y, ws.x, ws.qkv, lengths, B, S, 2, S, nullptr, nullptr, nullptr, 0))) return rc;
    if ((rc = skip_linear(st, ws.x, ws.sk0, E.skip_w[1], E.skip_b[1], ws.x, B * S))) return rc;
    // last layer: only rows 0,1 (mu, logvar) are consumed (mld_vae.py:172-173,186-187); output [2,B,256]
    return run_layer(st, E.layer[4], ff, act, ws.x, ws.y, mu, ws.qkv, lengths, B, S, 2, 2, nullptr, E.norm_w, E.norm_b, 1);
}

extern "C" int seeme_vae_decode(const SeemeVaeWeights* w, const float* z, const int32_t* lengths,
                                int B, int T, float* feats, void* workspace, size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (B <= 0 || T <= 0 || T > 500) return seeme_fail("vae_decode: need B>0 and 0 < T <= 500");
    if (ws_bytes < seeme_vae_workspace_bytes(B, T)) return seeme_fail("vae_decode: workspace too small");
    const int S = T, F = w->nfeats;
    if (w->h16 != nullptr) return seeme_vae_decode_h16(w, z, lengths, B, T, feats, workspace, st);
    VaeWs ws = carve(workspace, B, S + 2);
    const SeemeSkipStack& Dk = w->dec;
    int rc;
    // cross-attention to the single memory token: softmax over one key == 1, so the block adds
    // c_l[b] = out_proj(W_v z_b + b_v) to every query (cross_attention.py:358-361)
    if (w->ca_fold_w == nullptr) return seeme_fail("vae_decode: folded cross-attention weights missing");
    const int CL = SEEME_NLAYERS * 256;
    if ((rc = linear_simple(st, z, 256, w->ca_fold_w, 256, w->ca_fold_b, ws.cvec, CL, B, CL, 256))) return rc;
    {
        const size_t n4 = (size_t)B * S * 64;
        hipLaunchKernelGGL(k_bcast_rows, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, w->pe_dec, ws.x, B, S);
        if ((rc = seeme_check_launch("k_bcast_rows"))) return rc;
    }
    const int ff = w->ff, act = SEEME_ACT_GELU;
    // SkipTransformerDecoder.forward (cross_attention.py:118-147)
    if ((rc = run_layer(st, Dk.layer[0], ff, act, ws.x, ws.y, ws.sk0, ws.qkv, lengths, B, S, 0, S, ws.cvec + 0 * 256, nullptr, nullptr, 0, CL))) return rc;
    if ((rc = run_layer(st, Dk.layer[1], ff, act, ws.sk0, ws.y, ws.sk1, ws.qkv, lengths, B, S, 0, S, ws.cvec + 1 * 256, nullptr, nullptr, 0, CL))) return rc;
    if ((rc = run_layer(st, Dk.layer[2], ff, act, ws.sk1, ws.y, ws.x, ws.qkv, lengths, B, S, 0, S, ws.cvec + 2 * 256, nullptr, nullptr, 0, CL))) return rc;
    if ((rc = skip_linear(st, ws.x, ws.sk1, Dk.skip_w[0], Dk.skip_b[0], ws.x, B * S))) return rc;
    if ((rc = run_layer(st, Dk.layer[3], ff, act, ws.x, ws.y, ws.x, ws.qkv, lengths, B, S, 0, S, ws.cvec + 3 * 256, nullptr, nullptr, 0, CL))) return rc;
    if ((rc = skip_linear(st, ws.x, ws.sk0, Dk.skip_w[1], Dk.skip_b[1], ws.x, B * S))) return rc;
    if ((rc = run_layer(st, Dk.layer[4], ff, act, ws.x, ws.y, ws.x, ws.qkv, lengths, B, S, 0, S, ws.cvec + 4 * 256, nullptr, nullptr, 0, CL))) return rc;
    // stack norm + final_layer (mld_vae.py:251)
    return linear_simple(st, ws.x, 256, w->fin_w, 256, w->fin_b, feats, F, B * S, F, 256, SEEME_ACT_NONE,
                         SEEME_ACT_NONE, Dk.norm_w, Dk.norm_b);
}
