// den_train.inc.hip -- stage-2 training of the denoiser on the HIP path (included by den_kernels.hip).
//
// Reference: MLD._diffusion_process (mld/models/modeltype/mld.py:582-631) + loss backward through
// MldDenoiser.forward (mld_denoiser.py:151-244, mdiff_transformer.py:152-304).
//
// The forward is the sampling kernel itself (one step, per-sample timesteps, unfolded fp32 image, SeemeSampleArgs.save):
// it dumps every intermediate the chain rule needs.  The backward below is the same chain walked in reverse, again ONE
// workgroup per sample: every linear y = W x contributes a GEMV dx = W^T dy streamed from a transposed copy of the
// image, and the wave-local epilogues are the hand-derived derivatives of LayerNorm, the 2+N-token softmax attention,
// the linear cross-attention with its two softmaxes, AdaLN, SiLU / GELU / ReLU.  What is a reduction over the BATCH
// (dW = sum_b dy_b x_b^T, bias and LayerNorm parameter gradients) is not done here: the kernel writes x and dy of every
// linear per sample (den_train.h) and the host reduces them with a handful of batched GEMMs -- those are GEMM-shaped and
// independent, the chain is not.  Gradients w.r.t. the condition / time tables go back to the host's (differentiable)
// table builders.  A dependent 64-row GEMM launch costs 15-25 us, a GEMV stage inside this kernel ~1 us.

// ------------------------------------------------------------------ weight images from the parameter tensors
struct PackDesc { const float* src; int N, K; long long dst_f, dst_b; };   // PyTorch [N,K] -> forward and transposed GEMV layouts
#define PACK_MAX 64
struct PackArgs { PackDesc d[PACK_MAX]; int n; float* img_f; float* img_b; };

// forward layout [K/4][N][4]: vector (k4, n) = W[n][4 k4 .. 4 k4+3];  transposed layout [N/4][K][4]: vector (n4, k) = W[4 n4 + j][k]
__global__ __launch_bounds__(256) void k_den_pack(const PackArgs a) {
    const PackDesc& d = a.d[blockIdx.y];
    const int nv = d.N * d.K / 4;
    for (int v = blockIdx.x * 256 + threadIdx.x; v < nv; v += gridDim.x * 256) {
        {
            const int k4 = v / d.N, n = v - k4 * d.N;
            st4(a.img_f + d.dst_f + 4 * (size_t)v, ld4(d.src + (size_t)n * d.K + 4 * k4));
        }
        if (a.img_b != nullptr) {
            const int n4 = v / d.K, k = v - n4 * d.K;
            const float* s = d.src + (size_t)(4 * n4) * d.K + k;
            st4(a.img_b + d.dst_b + 4 * (size_t)v, make_float4(s[0], s[d.K], s[2 * (size_t)d.K], s[3 * (size_t)d.K]));
        }
    }
}
struct VecDesc { const float* src; int n; long long dst; };
#define VPACK_MAX 128
struct VPackArgs { VecDesc d[VPACK_MAX]; int n; float* vp; };
__global__ __launch_bounds__(256) void k_den_vpack(const VPackArgs a) {
    const VecDesc& d = a.d[blockIdx.x];
    for (int i = threadIdx.x; i < d.n; i += 256) a.vp[d.dst + i] = d.src[i];
}

// element offsets of the transposed image: per layer FO, F2, F1, CAO, CAQ, L2, L1, OUTP, INP, SKIP (layers 3, 4)
struct DenLayoutB { long long m[SEEME_DEN_NL][10]; long long total; };
static inline DenLayoutB seeme_make_den_layout_bwd() {
    DenLayoutB lb;
    long long w = 0;
    for (int l = 0; l < SEEME_DEN_NL; ++l)
        for (int i = 0; i < 10; ++i) {
            const int g = GB_FO + i;
            if (g == GB_SKIP && l < 3) { lb.m[l][i] = -1; continue; }
            lb.m[l][i] = w;
            w += (long long)den_gK(g) * den_gN(g);
        }
    lb.total = w;
    return lb;
}

extern "C" int seeme_den_train_layout(int64_t* out, int cap) {
    if (cap < 5 * 10 + 4) return seeme_fail("seeme_den_train_layout: output too small");
    const DenLayoutB lb = seeme_make_den_layout_bwd();
    int k = 0;
    for (int l = 0; l < SEEME_DEN_NL; ++l) for (int i = 0; i < 10; ++i) out[k++] = lb.m[l][i];
    out[k++] = lb.total; out[k++] = DT_TOTAL; out[k++] = DB_TOTAL; out[k++] = DB_LAYER;
    return 0;
}

// srcs: device pointers of the 10 matrices per layer in DenLayerOff order (skip, inp, outp, l1, l2, caq, cao, f1, f2, fo;
// skip NULL for layers 0-2), [5][10]; vec_src / vec_n / vec_dst: the vector parameters and their offsets in vp.
extern "C" int seeme_den_train_pack(const float* const* mats, float* img_f, float* img_b, const float* const* vec_src,
                                    const int* vec_n, const int64_t* vec_dst, int n_vec, float* vp, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const DenLayout lf = seeme_make_den_layout(FF_SA, FF_D);
    const DenLayoutB lb = seeme_make_den_layout_bwd();
    // forward matrix id -> backward slot
    const int bslot[10] = {9 /*skip*/, 8 /*inp*/, 7 /*outp*/, 6 /*l1*/, 5 /*l2*/, 4 /*caq*/, 3 /*cao*/, 2 /*f1*/, 1 /*f2*/, 0 /*fo*/};
    PackArgs pa;
    pa.n = 0; pa.img_f = img_f; pa.img_b = img_b;
    for (int l = 0; l < SEEME_DEN_NL; ++l)
        for (int g = 0; g < 10; ++g) {
            const float* src = mats[l * 10 + g];
            if (src == nullptr) continue;
            PackDesc& d = pa.d[pa.n++];
            d.src = src; d.N = den_gN(g); d.K = den_gK(g);
            d.dst_f = reinterpret_cast<const int64_t*>(&lf.L[l])[g];
            d.dst_b = lb.m[l][bslot[g]];
            if (d.dst_f < 0 || d.dst_b < 0) return seeme_fail("den_train_pack: matrix given for a slot the layer does not have");
        }
    hipLaunchKernelGGL(k_den_pack, dim3(64, pa.n), dim3(256), 0, st, pa);
    int rc = seeme_check_launch("k_den_pack");
    if (rc) return rc;
    if (n_vec > VPACK_MAX) return seeme_fail("den_train_pack: too many vectors");
    VPackArgs va;
    va.n = n_vec; va.vp = vp;
    for (int i = 0; i < n_vec; ++i) { va.d[i].src = vec_src[i]; va.d[i].n = vec_n[i]; va.d[i].dst = vec_dst[i]; }
    hipLaunchKernelGGL(k_den_vpack, dim3(n_vec), dim3(256), 0, st, va);
    return seeme_check_launch("k_den_vpack");
}

// ------------------------------------------------------------------ backward chain
struct DenBwdArgs {
    const void* wb; int wb_bytes;      // transposed image
    DenLayoutB lb;
    const float* vp; DenLayout lay;    // vector params (LayerNorm weights) and their offsets
    int B, N, xcds;                    // xcds < 8: the B working workgroups sit on the first xcds XCDs (see launch_den)
    const float* save;                 // [B, DT_TOTAL] from the forward
    const unsigned char* drop;         // [B, DM_TOTAL] dropout keep-masks of the forward, or NULL
    float drop_scale;
    const float* ctab; const float* ttab; const int32_t* trow;   // the forward's tables (per-sample rows)
    const float* dout;                 // [B,256] gradient of the model output
    float* gout;                       // [B, DB_TOTAL]
    float* dctab; float* dttab;        // [B,N,SEEME_CROW], [B,SEEME_TROW] (row b of dttab belongs to sample b)
};

// plain double-buffered GEMV (no cross-GEMV prefetch): part = partial sums of W^T x; ends with a barrier
template <int G, int... Cs>
__device__ __forceinline__ void gemv_plain_steps(int tid, __amdgpu_buffer_rsrc_t rsrc, unsigned mat_bytes, const float* __restrict__ x,
                                                 float* __restrict__ part, u32x4 (&b0)[DEN_CH], u32x4 (&b1)[DEN_CH],
                                                 Acc<WF32, 1, G>& acc, std::integer_sequence<int, Cs...>) {
    typedef GS<WF32, G> S;
    ((([&] {
        if constexpr (Cs + 1 < S::TOT) issue_mat<WF32, G, (Cs + 1 < S::TOT ? Cs + 1 : 0)>((Cs & 1) ? b0 : b1, tid, rsrc, mat_bytes);
        consume_chunk_valu<WF32, 1, G, Cs>((Cs & 1) ? b1 : b0, tid, x, part, acc);
        acc.pin();
    })()), ...);
}
template <int G>
__device__ __forceinline__ void gemv_plain(int tid, __amdgpu_buffer_rsrc_t rsrc, long long mat_elems, const float* __restrict__ x,
                                           float* __restrict__ part) {
    asm volatile("" : "+v"(tid));
    u32x4 b0[DEN_CH], b1[DEN_CH];
    const unsigned mat_bytes = (unsigned)(mat_elems * 4);
    issue_mat<WF32, G, 0>(b0, tid, rsrc, mat_bytes);
    Acc<WF32, 1, G> acc;
    acc.zero();
    gemv_plain_steps<G>(tid, rsrc, mat_bytes, x, part, b0, b1, acc, std::make_integer_sequence<int, GS<WF32, G>::TOT>{});
    __syncthreads();
}

__device__ __forceinline__ float4 f4_mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 f4_sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float dsilu(float u) { const float s = fast_rcp(1.f + fast_exp(-u)); return s * (1.f + u * (1.f - s)); }
__device__ __forceinline__ float4 f4_dsilu(float4 u) { return make_float4(dsilu(u.x), dsilu(u.y), dsilu(u.z), dsilu(u.w)); }
__device__ __forceinline__ float dgelu(float z) {   // Phi(z) + z phi(z)
    const float cdf = 0.5f * (1.f + erff(z * 0.70710678118654752440f));
    return cdf + z * 0.39894228040143267794f * fast_exp(-0.5f * z * z);
}
// y = xhat * w + b (xhat = (v - mean) rs): returns dv; writes the per-sample parameter gradients dy*xhat | dy
__device__ __forceinline__ float4 ln_bwd(float4 dy, const float* __restrict__ w, float4 xhat, float rs, int lane, float* __restrict__ g) {
    const float4 gw = f4_mul(dy, ld4(w + 4 * lane));
    const float m1 = wave_sum(gw.x + gw.y + gw.z + gw.w) * (1.f / 256.f);
    const float m2 = wave_sum(f4_dot(gw, xhat)) * (1.f / 256.f);
    if (g != nullptr) { st4(g + 4 * lane, f4_mul(dy, xhat)); st4(g + 256 + 4 * lane, dy); }
    return make_float4(rs * (gw.x - m1 - xhat.x * m2), rs * (gw.y - m1 - xhat.y * m2), rs * (gw.z - m1 - xhat.z * m2), rs * (gw.w - m1 - xhat.w * m2));
}

__global__ __launch_bounds__(DEN_THREADS) void k_den_bwd(const DenBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float XB[1024];     // GEMV input
    __shared__ __attribute__((aligned(16))) float PART[1024];   // GEMV partial sums (k-slice major)
    const __amdgpu_buffer_rsrc_t wb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wb), 0, a.wb_bytes, 0x00020000);
    const int tid = threadIdx.x, lane = tid & 63;
    const bool w0 = (tid >> 6) == 0;                            // every wave runs the epilogue math; wave 0 writes
    int b = blockIdx.x;
    if (a.xcds < 8) {
        const int x = blockIdx.x & 7;
        b = (blockIdx.x >> 3) * a.xcds + x;
        if (x >= a.xcds || b >= a.B) return;
    }
    const int N = a.N;
    const float* __restrict__ sv0 = a.save + (size_t)b * DT_TOTAL;
    float* __restrict__ go0 = a.gout + (size_t)b * DB_TOTAL;
    const float* __restrict__ tt = a.ttab + (size_t)a.trow[b] * SEEME_TROW;
    float* __restrict__ dtt = a.dttab + (size_t)b * SEEME_TROW;
    const float* __restrict__ vp = a.vp;
    const float scale = 1.f / 16.f;                             // one head of 256 dims
    const unsigned char* const dm0 = a.drop ? a.drop + (size_t)b * DM_TOTAL : nullptr;
    const float dsc = a.drop_scale;
    auto drop4 = [&](float4 v, const unsigned char* m) {        // the forward's dropout mask on a gradient / a saved activation
        const uchar4 k = *reinterpret_cast<const uchar4*>(m);
        return make_float4(k.x ? v.x * dsc : 0.f, k.y ? v.y * dsc : 0.f, k.z ? v.z * dsc : 0.f, k.w ? v.w * dsc : 0.f);
    };

    // ---- encoder.norm
    float4 dx;
    {
        const float4 dy = ld4(a.dout + (size_t)b * 256 + 4 * lane);
        dx = ln_bwd(dy, vp + a.lay.fnw, ld4(sv0 + DT_FIN + 4 * lane), sv0[DT_FIN + 256], lane, w0 ? go0 + DB_FIN : nullptr);
    }
    float4 dsk0 = make_float4(0.f, 0.f, 0.f, 0.f), dsk1 = dsk0;   // gradients of the skip copies of layers 0 / 1

#pragma unroll 1
    for (int l = SEEME_DEN_NL - 1; l >= 0; --l) {
        const DenLayerOff& L = a.lay.L[l];
        const long long* __restrict__ mb = a.lb.m[l];
        const float* __restrict__ sv = sv0 + l * DT_LAYER;
        float* __restrict__ go = go0 + (size_t)l * DB_LAYER;
        float* const gw = w0 ? go : nullptr;                     // (writes by wave 0 only)
        const unsigned char* const dm = dm0 ? dm0 + l * DM_LAYER : nullptr;
        if (l == 1) dx = f4_add(dx, dsk1);
        if (l == 0) dx = f4_add(dx, dsk0);

        // ================= ffn: x4 = x3 + Wfo silu(u2) + b
        const float4 u2 = ld4(sv + DT_U2 + 4 * lane);
        if (gw) { st4(gw + DB_X_FO + 4 * lane, dm ? drop4(f4_silu(u2), dm + DM_O + 4 * lane) : f4_silu(u2)); st4(gw + DB_Y_FO + 4 * lane, dx); st4(XB + 4 * lane, dx); }
        __syncthreads();
        gemv_plain<GB_FO>(tid, wb, mb[0], XB, PART);
        float4 dy2;
        {
            float4 ds2 = part_sum<1, GS<WF32, GB_FO>::KS, 256>(PART, 0, 0, lane);
            if (dm) ds2 = drop4(ds2, dm + DM_O + 4 * lane);
            const float4 du2 = f4_mul(ds2, f4_dsilu(u2));
            const float4 xh = ld4(sv + DT_XHY2 + 4 * lane);
            const float4 lw = ld4(vp + L.fsnw + 4 * lane), lb = ld4(vp + L.fsnb + 4 * lane);
            const float4 ln = make_float4(xh.x * lw.x + lb.x, xh.y * lw.y + lb.y, xh.z * lw.z + lb.z, xh.w * lw.w + lb.w);
            const float4 sc = ld4(tt + 2560 + l * 1024 + 512 + 4 * lane);
            if (w0) { st4(dtt + 2560 + l * 1024 + 512 + 4 * lane, f4_mul(du2, ln)); st4(dtt + 2560 + l * 1024 + 768 + 4 * lane, du2); }
            const float4 dln = make_float4(du2.x * (1.f + sc.x), du2.y * (1.f + sc.y), du2.z * (1.f + sc.z), du2.w * (1.f + sc.w));
            dy2 = ln_bwd(dln, vp + L.fsnw, xh, sv[DT_RS + 4], lane, gw ? gw + DB_LN + 4 * 512 : nullptr);
        }
        float4 z1 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane < 32) z1 = ld4(sv + DT_Z1 + 4 * lane);
        if (gw) {
            st4(gw + DB_Y_F2 + 4 * lane, dy2); st4(XB + 4 * lane, dy2);
            if (lane < 32) {
                const float4 gz = make_float4(fast_gelu(z1.x), fast_gelu(z1.y), fast_gelu(z1.z), fast_gelu(z1.w));
                st4(gw + DB_X_F2 + 4 * lane, dm ? drop4(gz, dm + DM_F + 4 * lane) : gz);
            }
        }
        __syncthreads();
        gemv_plain<GB_F2>(tid, wb, mb[1], XB, PART);
        {
            float4 dz1 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (lane < 32) {
                float4 dg = part_sum<1, GS<WF32, GB_F2>::KS, FF_D>(PART, 0, 0, lane);
                if (dm) dg = drop4(dg, dm + DM_F + 4 * lane);
                dz1 = make_float4(dg.x * dgelu(z1.x), dg.y * dgelu(z1.y), dg.z * dgelu(z1.z), dg.w * dgelu(z1.w));
            }
            if (gw) {
                st4(gw + DB_X_F1 + 4 * lane, ld4(sv + DT_X3 + 4 * lane));
                if (lane < 32) { st4(gw + DB_Y_F1 + 4 * lane, dz1); st4(XB + 4 * lane, dz1); }
            }
        }
        __syncthreads();
        gemv_plain<GB_F1>(tid, wb, mb[2], XB, PART);
        const float4 dx3 = f4_add(dx, part_sum<1, GS<WF32, GB_F1>::KS, 256>(PART, 0, 0, lane));

        // ================= ca_block: x3 = x2 + Wpo silu(u) + b,  u = LN(y)(1+sc)+sh,  y = sum_j (qc . kc_j) vv_j
        const float4 u = ld4(sv + DT_U + 4 * lane);
        if (gw) { st4(gw + DB_X_CAO + 4 * lane, dm ? drop4(f4_silu(u), dm + DM_C + 4 * lane) : f4_silu(u)); st4(gw + DB_Y_CAO + 4 * lane, dx3); st4(XB + 4 * lane, dx3); }
        __syncthreads();
        gemv_plain<GB_CAO>(tid, wb, mb[3], XB, PART);
        float4 dqq;
        {
            float4 dsu = part_sum<1, GS<WF32, GB_CAO>::KS, 256>(PART, 0, 0, lane);
            if (dm) dsu = drop4(dsu, dm + DM_C + 4 * lane);
            const float4 du = f4_mul(dsu, f4_dsilu(u));
            const float4 xh = ld4(sv + DT_XHY + 4 * lane);
            const float4 lw = ld4(vp + L.csnw + 4 * lane), lb = ld4(vp + L.csnb + 4 * lane);
            const float4 ln = make_float4(xh.x * lw.x + lb.x, xh.y * lw.y + lb.y, xh.z * lw.z + lb.z, xh.w * lw.w + lb.w);
            const float4 sc = ld4(tt + 2560 + l * 1024 + 4 * lane);
            if (w0) { st4(dtt + 2560 + l * 1024 + 4 * lane, f4_mul(du, ln)); st4(dtt + 2560 + l * 1024 + 256 + 4 * lane, du); }
            const float4 dln = make_float4(du.x * (1.f + sc.x), du.y * (1.f + sc.y), du.z * (1.f + sc.z), du.w * (1.f + sc.w));
            const float4 dy = ln_bwd(dln, vp + L.csnw, xh, sv[DT_RS + 3], lane, gw ? gw + DB_LN + 3 * 512 : nullptr);
            // keys: softmax over the tokens per dim (recomputed), values from the table
            const float4 qc = ld4(sv + DT_QC + 4 * lane);
            float4 kc[DEN_MAXTOK - 2], dkc[DEN_MAXTOK - 2];
            float4 kmx = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY), ks = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < DEN_MAXTOK - 2; ++j)
                if (j < N) {
                    kc[j] = ld4(a.ctab + ((size_t)b * N + j) * SEEME_CROW + 2560 + l * 512 + 4 * lane);
                    kmx = make_float4(fmaxf(kmx.x, kc[j].x), fmaxf(kmx.y, kc[j].y), fmaxf(kmx.z, kc[j].z), fmaxf(kmx.w, kc[j].w));
                }
#pragma unroll
            for (int j = 0; j < DEN_MAXTOK - 2; ++j)
                if (j < N) {
                    kc[j] = make_float4(fast_exp(kc[j].x - kmx.x), fast_exp(kc[j].y - kmx.y), fast_exp(kc[j].z - kmx.z), fast_exp(kc[j].w - kmx.w));
                    ks = f4_add(ks, kc[j]);
                }
            const float4 rks = make_float4(fast_rcp(ks.x), fast_rcp(ks.y), fast_rcp(ks.z), fast_rcp(ks.w));
            float4 dqc = make_float4(0.f, 0.f, 0.f, 0.f), sdk = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < DEN_MAXTOK - 2; ++j)
                if (j < N) {
                    kc[j] = f4_mul(kc[j], rks);
                    const float4 vv = ld4(a.ctab + ((size_t)b * N + j) * SEEME_CROW + 2560 + l * 512 + 256 + 4 * lane);
                    const float ddot = wave_sum(f4_dot(dy, vv));
                    if (w0) st4(a.dctab + ((size_t)b * N + j) * SEEME_CROW + 2560 + l * 512 + 256 + 4 * lane, f4_scale(dy, sv[DT_RS + 8 + j]));
                    dqc = f4_fma(ddot, kc[j], dqc);
                    dkc[j] = f4_scale(qc, ddot);
                    sdk = f4_add(sdk, f4_mul(dkc[j], kc[j]));
                }
#pragma unroll
            for (int j = 0; j < DEN_MAXTOK - 2; ++j)
                if (j < N && w0) st4(a.dctab + ((size_t)b * N + j) * SEEME_CROW + 2560 + l * 512 + 4 * lane, f4_mul(kc[j], f4_sub(dkc[j], sdk)));
            const float s = wave_sum(f4_dot(dqc, qc));
            dqq = f4_mul(qc, make_float4(dqc.x - s, dqc.y - s, dqc.z - s, dqc.w - s));
        }
        const float4 xhc = ld4(sv + DT_XHC + 4 * lane);
        if (gw) {
            const float4 lw = ld4(vp + L.cnw + 4 * lane), lb = ld4(vp + L.cnb + 4 * lane);
            st4(gw + DB_X_CAQ + 4 * lane, make_float4(xhc.x * lw.x + lb.x, xhc.y * lw.y + lb.y, xhc.z * lw.z + lb.z, xhc.w * lw.w + lb.w));
            st4(gw + DB_Y_CAQ + 4 * lane, dqq); st4(XB + 4 * lane, dqq);
        }
        __syncthreads();
        gemv_plain<GB_CAQ>(tid, wb, mb[4], XB, PART);
        float4 dx2;
        {
            const float4 dqn = part_sum<1, GS<WF32, GB_CAQ>::KS, 256>(PART, 0, 0, lane);
            dx2 = f4_add(dx3, ln_bwd(dqn, vp + L.cnw, xhc, sv[DT_RS + 2], lane, gw ? gw + DB_LN + 2 * 512 : nullptr));
        }

        // ================= sa_block tail: x2 = LN2(x1 + W2 relu(W1 x1))
        const float4 dv2 = ln_bwd(dx2, vp + L.n2w, ld4(sv + DT_XH2 + 4 * lane), sv[DT_RS + 1], lane, gw ? gw + DB_LN + 1 * 512 : nullptr);
        const float4 dv2m = dm ? drop4(dv2, dm + DM_2 + 4 * lane) : dv2;      // through dropout2 into linear2's output
        if (gw) {
            st4(gw + DB_Y_L2 + 4 * lane, dv2m); st4(XB + 4 * lane, dv2m);
#pragma unroll
            for (int j = 0; j < 4; ++j) st4(gw + DB_X_L2 + 256 * j + 4 * lane, ld4(sv + DT_H + 256 * j + 4 * lane));
        }
        __syncthreads();
        gemv_plain<GB_L2>(tid, wb, mb[5], XB, PART);
        {
            float4 dz[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 h = ld4(sv + DT_H + 256 * j + 4 * lane);
                const float4 dh = part_sum<1, GS<WF32, GB_L2>::KS, FF_SA>(PART, 0, 256 * j, lane);
                // (the saved h is relu + dropout: h > 0 means kept and active; a kept element's gradient carries the 1 / (1 - p))
                const float kf = dm ? dsc : 1.f;
                dz[j] = make_float4(h.x > 0.f ? dh.x * kf : 0.f, h.y > 0.f ? dh.y * kf : 0.f, h.z > 0.f ? dh.z * kf : 0.f, h.w > 0.f ? dh.w * kf : 0.f);
            }
            __syncthreads();                                   // every wave has read PART before XB / PART are reused
            if (gw) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { st4(gw + DB_Y_L1 + 256 * j + 4 * lane, dz[j]); st4(XB + 256 * j + 4 * lane, dz[j]); }
            }
        }
        const float4 xh1 = ld4(sv + DT_XH1 + 4 * lane);
        if (gw) {
            const float4 lw = ld4(vp + L.n1w + 4 * lane), lb = ld4(vp + L.n1b + 4 * lane);
            st4(gw + DB_X_L1 + 4 * lane, make_float4(xh1.x * lw.x + lb.x, xh1.y * lw.y + lb.y, xh1.z * lw.z + lb.z, xh1.w * lw.w + lb.w));
        }
        __syncthreads();
        gemv_plain<GB_L1>(tid, wb, mb[6], XB, PART);
        const float4 dx1 = f4_add(dv2, part_sum<1, GS<WF32, GB_L1>::KS, 256>(PART, 0, 0, lane));
        const float4 dv1 = ln_bwd(dx1, vp + L.n1w, xh1, sv[DT_RS + 0], lane, gw ? gw + DB_LN : nullptr);

        // ================= attention over [x, xf.., emb] (token 0 only): o = Wo sum_j p_j v_j
        const float4 dv1m = dm ? drop4(dv1, dm + DM_1 + 4 * lane) : dv1;      // through dropout1 into out_proj's output
        if (gw) { st4(gw + DB_X_OUTP + 4 * lane, ld4(sv + DT_A + 4 * lane)); st4(gw + DB_Y_OUTP + 4 * lane, dv1m); st4(XB + 4 * lane, dv1m); }
        __syncthreads();
        gemv_plain<GB_OUTP>(tid, wb, mb[7], XB, PART);
        {
            const float4 da = part_sum<1, GS<WF32, GB_OUTP>::KS, 256>(PART, 0, 0, lane);
            const float4 q = ld4(sv + DT_QKV + 4 * lane), k0 = ld4(sv + DT_QKV + 256 + 4 * lane), v0 = ld4(sv + DT_QKV + 512 + 4 * lane);
            // d p_j = da . v_j ; softmax backward
            // a = sum_j p_j m_j v_j with the saved (un-dropped) p and the keep factors m_j in {0, 1 / (1 - p)}
            float dp[DEN_MAXTOK], p[DEN_MAXTOK];
            auto pk = [&](int j) { return dm ? (dm[DM_P + j] ? dsc : 0.f) : 1.f; };
            p[0] = sv[DT_P]; dp[0] = wave_sum(f4_dot(da, v0)) * pk(0);
            float dsum = p[0] * dp[0];
#pragma unroll
            for (int j = 0; j < DEN_MAXTOK - 2; ++j)
                if (j < N) {
                    p[1 + j] = sv[DT_P + 1 + j];
                    dp[1 + j] = wave_sum(f4_dot(da, ld4(a.ctab + ((size_t)b * N + j) * SEEME_CROW + l * 512 + 256 + 4 * lane))) * pk(1 + j);
                    dsum += p[1 + j] * dp[1 + j];
                }
            const float pt = sv[DT_P + 1 + N];
            const float dpt = wave_sum(f4_dot(da, ld4(tt + l * 512 + 256 + 4 * lane))) * pk(1 + N);
            dsum += pt * dpt;
            const float ds0 = p[0] * (dp[0] - dsum) * scale;
            float4 dq = f4_scale(k0, ds0);
            const float4 dk0 = f4_scale(q, ds0), dv0 = f4_scale(da, p[0] * pk(0));
#pragma unroll
            for (int j = 0; j < DEN_MAXTOK - 2; ++j)
                if (j < N) {
                    const float dsj = p[1 + j] * (dp[1 + j] - dsum) * scale;
                    dq = f4_fma(dsj, ld4(a.ctab + ((size_t)b * N + j) * SEEME_CROW + l * 512 + 4 * lane), dq);
                    if (w0) {
                        st4(a.dctab + ((size_t)b * N + j) * SEEME_CROW + l * 512 + 4 * lane, f4_scale(q, dsj));
                        st4(a.dctab + ((size_t)b * N + j) * SEEME_CROW + l * 512 + 256 + 4 * lane, f4_scale(da, p[1 + j] * pk(1 + j)));
                    }
                }
            const float dst = pt * (dpt - dsum) * scale;
            dq = f4_fma(dst, ld4(tt + l * 512 + 4 * lane), dq);
            if (w0) { st4(dtt + l * 512 + 4 * lane, f4_scale(q, dst)); st4(dtt + l * 512 + 256 + 4 * lane, f4_scale(da, pt * pk(1 + N))); }
            if (gw) {
                st4(gw + DB_X_INP + 4 * lane, ld4(sv + DT_X + 4 * lane));
                st4(gw + DB_Y_INP + 4 * lane, dq); st4(gw + DB_Y_INP + 256 + 4 * lane, dk0); st4(gw + DB_Y_INP + 512 + 4 * lane, dv0);
                st4(XB + 4 * lane, dq); st4(XB + 256 + 4 * lane, dk0); st4(XB + 512 + 4 * lane, dv0);
            }
        }
        __syncthreads();
        gemv_plain<GB_INP>(tid, wb, mb[8], XB, PART);
        float4 dxin = f4_add(dv1, part_sum<1, GS<WF32, GB_INP>::KS, 256>(PART, 0, 0, lane));

        // ================= skip connection: x = Ws [x_prev | skip] + b (layers 3, 4)
        if (l >= 3) {
            const float* xp = sv0 + (l - 1) * DT_LAYER + DT_X4;
            const float* xs = sv0 + (l == 3 ? 1 : 0) * DT_LAYER + DT_X4;
            if (gw) {
                st4(gw + DB_X_SKIP + 4 * lane, ld4(xp + 4 * lane)); st4(gw + DB_X_SKIP + 256 + 4 * lane, ld4(xs + 4 * lane));
                st4(gw + DB_Y_SKIP + 4 * lane, dxin); st4(XB + 4 * lane, dxin);
            }
            __syncthreads();
            gemv_plain<GB_SKIP>(tid, wb, mb[9], XB, PART);
            dx = part_sum<1, GS<WF32, GB_SKIP>::KS, 512>(PART, 0, 0, lane);
            const float4 ds = part_sum<1, GS<WF32, GB_SKIP>::KS, 512>(PART, 0, 256, lane);
            if (l == 3) dsk1 = ds; else dsk0 = ds;
        } else {
            dx = dxin;
        }
        __syncthreads();                                        // PART is rewritten by the next layer's first GEMV
    }
    if (w0) st4(go0 + DB_DX0 + 4 * lane, dx);
}

static_assert(DM_TOTAL == SEEME_DEN_DROP_BYTES, "dropout mask block: header constant out of date");
extern "C" int seeme_denoiser_backward(const SeemeDenoiserWeights* w, const void* img_bwd, int B, int N, const float* save,
                                       const float* ctab, const float* ttab, const int32_t* trow, const float* dout,
                                       float* gout, float* dctab, float* dttab, void* stream) {
    return seeme_denoiser_backward_drop(w, img_bwd, B, N, save, ctab, ttab, trow, dout, gout, dctab, dttab, nullptr, 1.f, 0, stream);
}

extern "C" int seeme_denoiser_backward_drop(const SeemeDenoiserWeights* w, const void* img_bwd, int B, int N, const float* save,
                                            const float* ctab, const float* ttab, const int32_t* trow, const float* dout,
                                            float* gout, float* dctab, float* dttab, const unsigned char* drop, float drop_scale,
                                            int xcds, void* stream) {
    if (B <= 0) return seeme_fail("denoiser_backward: B must be > 0");
    if (N < 1 || N > DEN_MAXTOK - 2) return seeme_fail("denoiser_backward: 1 <= N <= 4 condition tokens");
    if (w->nhead != 1 || w->sa_fold || w->wdtype != 0) return seeme_fail("denoiser_backward: one head, unfolded fp32 image");
    if (w->ff_sa != FF_SA || w->ff != FF_D) return seeme_fail("denoiser_backward: built for sa ff 1024 / ffn_dim 128");
    DenBwdArgs a;
    a.wb = img_bwd; a.lb = seeme_make_den_layout_bwd(); a.wb_bytes = (int)(a.lb.total * 4);
    a.vp = w->vp; a.lay = seeme_make_den_layout(FF_SA, FF_D);
    a.B = B; a.N = N; a.save = save; a.ctab = ctab; a.ttab = ttab; a.trow = trow; a.dout = dout;
    a.drop = drop; a.drop_scale = drop_scale;
    a.gout = gout; a.dctab = dctab; a.dttab = dttab;
    const int grid = den_xcd_grid(B, xcds, &a.xcds);
    hipLaunchKernelGGL(k_den_bwd, dim3(grid), dim3(DEN_THREADS), 0, (hipStream_t)stream, a);
    return seeme_check_launch("k_den_bwd");
}

// ---------------------------------------------------------------------------------------------
// Weight gradients of the chain: dW[n][k] = sum_b dy_b[n] x_b[k] for every linear of every layer, read straight from the
// per-sample backward buffer gout [B, ldg] that k_den_bwd wrote, in ONE launch.  (As one batched GEMM per matrix kind
// with K = B = 64 the ten hipBLASLt calls took 0.64 ms for 1 GFLOP; the work is a 32 MB write.)  A workgroup owns a
// 32 x 256 tile of one matrix: thread <-> column k, 32 accumulators, dy rows broadcast from LDS, b in order (exact fp32
// FMA chain: deterministic).
struct SeemeWgradTile { int x_col, y_col, ldo, nn, kk, pad; long long out_off; };
__global__ __launch_bounds__(256) void k_den_wgrad(const float* __restrict__ gout, int ldg, int B,
                                                   const SeemeWgradTile* __restrict__ tiles, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float sdy[64][32];
    const SeemeWgradTile t = tiles[blockIdx.x];
    const int tid = threadIdx.x;
    float acc[32];
#pragma unroll
    for (int n = 0; n < 32; ++n) acc[n] = 0.f;
    const float* xcol = gout + t.x_col + (tid < t.kk ? tid : 0);
    for (int b0 = 0; b0 < B; b0 += 64) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = tid + i * 256, bb = idx >> 5, n = idx & 31, b = b0 + bb;
            const float v = gout[(size_t)(b < B ? b : 0) * ldg + t.y_col + (n < t.nn ? n : 0)];
            sdy[bb][n] = (b < B && n < t.nn) ? v : 0.f;
        }
        __syncthreads();
        const int nb = min(64, B - b0);
        for (int bb0 = 0; bb0 < nb; bb0 += 8) {
            float xv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) xv[j] = xcol[(size_t)(b0 + (bb0 + j < nb ? bb0 + j : 0)) * ldg];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (bb0 + j < nb) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const float4 d = *reinterpret_cast<const float4*>(&sdy[bb0 + j][4 * q]);
                        acc[4 * q] += d.x * xv[j]; acc[4 * q + 1] += d.y * xv[j]; acc[4 * q + 2] += d.z * xv[j]; acc[4 * q + 3] += d.w * xv[j];
                    }
                }
            }
        }
    }
    if (tid < t.kk) {
        float* o = out + t.out_off + tid;
#pragma unroll
        for (int n = 0; n < 32; ++n)
            if (n < t.nn) o[(size_t)n * t.ldo] = acc[n];
    }
}

extern "C" int seeme_den_wgrad(const float* gout, int ldg, int B, const void* tiles, int n_tiles, float* out, void* stream) {
    if (B <= 0 || n_tiles <= 0) return seeme_fail("den_wgrad: empty");
    hipLaunchKernelGGL(k_den_wgrad, dim3((unsigned)n_tiles), dim3(256), 0, (hipStream_t)stream, gout, ldg, B,
                       (const SeemeWgradTile*)tiles, out);
    return seeme_check_launch("k_den_wgrad");
}


// Bias / LayerNorm gradients of the chain: out[q] = sum_b gout[b*ldg + idx[q]] (one gather-reduce instead of
// column-sum + index_select + copies), and the one row of query_pos.pe that is on the path.
__global__ __launch_bounds__(256) void k_den_vecgrad(const float* __restrict__ gout, int ldg, int B, const int64_t* __restrict__ idx,
                                                     int n, float* __restrict__ out, int dx0_col, float* __restrict__ dpe_row0) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= n + 256) return;
    const long col = q < n ? idx[q] : dx0_col + (q - n);
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += gout[(long)b * ldg + col];
    if (q < n) out[q] = s;
    else if (dpe_row0) dpe_row0[q - n] = s;
}

extern "C" int seeme_den_vecgrad(const float* gout, int ldg, int B, const int64_t* idx, int n, float* out, int dx0_col,
                                 float* dpe_row0, void* stream) {
    if (B <= 0 || n <= 0 || !gout || !idx || !out) return seeme_fail("den_vecgrad: bad arguments");
    hipLaunchKernelGGL(k_den_vecgrad, dim3((unsigned)((n + 256 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gout, ldg, B, idx, n,
                       out, dx0_col, dpe_row0);
    return seeme_check_launch("k_den_vecgrad");
}
