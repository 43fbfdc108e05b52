// misc_kernels.hip -- rotation-representation helpers (mld/utils/geometry2.py:33-117), renorm
// (mld/data/EgoBody.py:151-157) and the ResNet-PointNet scene encoder (EgoHMR/models/respointnet.py)
// sequenced over the fused fp32-MFMA linear kernel.
#include "common.hpp"
#include "api_util.hpp"

// ------------------------------------------------------------------ geometry (one thread per item)
__device__ __forceinline__ void quat_to_R(float w, float x, float y, float z, float* R) {
    // geometry2.py:74-95 (normalises first)
    const float n = sqrtf(w * w + x * x + y * y + z * z);
    w /= n; x /= n; y /= n; z /= n;
    const float w2 = w * w, x2 = x * x, y2 = y * y, z2 = z * z;
    const float wx = w * x, wy = w * y, wz = w * z, xy = x * y, xz = x * z, yz = y * z;
    R[0] = w2 + x2 - y2 - z2; R[1] = 2 * xy - 2 * wz;     R[2] = 2 * wy + 2 * xz;
    R[3] = 2 * wz + 2 * xy;   R[4] = w2 - x2 + y2 - z2;   R[5] = 2 * yz - 2 * wx;
    R[6] = 2 * xz - 2 * wy;   R[7] = 2 * wx + 2 * yz;     R[8] = w2 - x2 - y2 + z2;
}
__device__ __forceinline__ void aa_to_q(const float* t, float* q) {
    // geometry2.py:33-54: norm(theta + 1e-8), half angle
    const float ex = t[0] + 1e-8f, ey = t[1] + 1e-8f, ez = t[2] + 1e-8f;
    const float ang = sqrtf(ex * ex + ey * ey + ez * ez);
    const float h = ang * 0.5f, s = sinf(h);
    q[0] = cosf(h); q[1] = s * (t[0] / ang); q[2] = s * (t[1] / ang); q[3] = s * (t[2] / ang);
}

__global__ void k_geometry(int op, const float* __restrict__ in, float* __restrict__ out, int M) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    if (op == SEEME_GEO_AA_TO_QUAT) {
        float q[4];
        aa_to_q(in + (size_t)m * 3, q);
        for (int i = 0; i < 4; ++i) out[(size_t)m * 4 + i] = q[i];
    } else if (op == SEEME_GEO_AA_TO_ROTMAT) {
        float q[4], R[9];
        aa_to_q(in + (size_t)m * 3, q);
        quat_to_R(q[0], q[1], q[2], q[3], R);
        for (int i = 0; i < 9; ++i) out[(size_t)m * 9 + i] = R[i];
    } else if (op == SEEME_GEO_QUAT_TO_ROTMAT) {
        float R[9];
        const float* q = in + (size_t)m * 4;
        quat_to_R(q[0], q[1], q[2], q[3], R);
        for (int i = 0; i < 9; ++i) out[(size_t)m * 9 + i] = R[i];
    } else {  // rot6d -> rotmat (geometry2.py:98-117), Gram-Schmidt, F.normalize eps 1e-12
        const float* x = in + (size_t)m * 6;
        float a1[3], a2[3];
        if (op == SEEME_GEO_ROT6D_PROHMR) {   // reshape(-1,2,3).permute(0,2,1): a1 = x[0:3], a2 = x[3:6]
            a1[0] = x[0]; a1[1] = x[1]; a1[2] = x[2]; a2[0] = x[3]; a2[1] = x[4]; a2[2] = x[5];
        } else {                              // reshape(-1,3,2): a1 = x[0::2], a2 = x[1::2]
            a1[0] = x[0]; a1[1] = x[2]; a1[2] = x[4]; a2[0] = x[1]; a2[1] = x[3]; a2[2] = x[5];
        }
        const float n1 = fmaxf(sqrtf(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]), 1e-12f);
        const float b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
        const float d = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
        float u[3] = {a2[0] - d * b1[0], a2[1] - d * b1[1], a2[2] - d * b1[2]};
        const float n2 = fmaxf(sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]), 1e-12f);
        const float b2[3] = {u[0] / n2, u[1] / n2, u[2] / n2};
        const float b3[3] = {b1[1] * b2[2] - b1[2] * b2[1], b1[2] * b2[0] - b1[0] * b2[2], b1[0] * b2[1] - b1[1] * b2[0]};
        float* R = out + (size_t)m * 9;     // stack((b1,b2,b3), dim=-1): columns
        R[0] = b1[0]; R[1] = b2[0]; R[2] = b3[0];
        R[3] = b1[1]; R[4] = b2[1]; R[5] = b3[1];
        R[6] = b1[2]; R[7] = b2[2]; R[8] = b3[2];
    }
}

extern "C" int seeme_geometry(int op, const float* in, float* out, int M, void* stream) {
    if (M <= 0) return seeme_fail("geometry: M must be > 0");
    if (op < SEEME_GEO_AA_TO_QUAT || op > SEEME_GEO_ROT6D_DIFFUSION) return seeme_fail("geometry: unknown op");
    hipLaunchKernelGGL(k_geometry, dim3((M + 255) / 256), dim3(256), 0, (hipStream_t)stream, op, in, out, M);
    return seeme_check_launch("k_geometry");
}

// y[r][c] = x[r][c] * std[c] + mean[c]   (EgoBodyDataModule.renorm)
__global__ void k_renorm(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ stdv,
                         float* __restrict__ y, size_t n, int F) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % F);
    y[i] = x[i] * stdv[c] + mean[c];
}
extern "C" int seeme_renorm(const float* x, const float* mean, const float* stdv, float* y, long rows, int F, void* stream) {
    const size_t n = (size_t)rows * F;
    if (n == 0) return seeme_fail("renorm: empty");
    hipLaunchKernelGGL(k_renorm, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, mean, stdv, y, n, F);
    return seeme_check_launch("k_renorm");
}

// ------------------------------------------------------------------ PointNet
// per-scene max over points: x [B,P,C] -> y [B,C]   (ResnetPointnet.pool, respointnet.py:29-31)
__global__ __launch_bounds__(256) void k_colmax(const float* __restrict__ x, float* __restrict__ y, int P, int C) {
    __shared__ float red[4][64];
    const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), r = threadIdx.x >> 6;
    float m = -INFINITY;
    if (c < C)
        for (int p = r; p < P; p += 4) m = fmaxf(m, x[((size_t)b * P + p) * C + c]);
    red[r][threadIdx.x & 63] = m;
    __syncthreads();
    if (r == 0 && c < C)
        y[(size_t)b * C + c] = fmaxf(fmaxf(red[0][threadIdx.x], red[1][threadIdx.x]), fmaxf(red[2][threadIdx.x], red[3][threadIdx.x]));
}

static int lin(hipStream_t st, const float* A, int lda, const float* W, int ldw, const float* bias, const float* res,
               int res_mode, int seq, float* Y, int M, int N, int K, int pre_act) {
    LinearKArgs ka{};
    ka.a.A = A; ka.a.lda = lda; ka.a.K1 = K; ka.a.K = K; ka.a.W = W; ka.a.ldw = ldw; ka.a.bias = bias;
    ka.a.res = res; ka.a.ldr = N; ka.a.Y = Y; ka.a.ldy = N; ka.a.M = M; ka.a.N = N; ka.a.pre_act = pre_act; ka.a.eps = 1e-5f;
    if (res_mode == 2) { ka.seq_in = seq; ka.in_stride = seq; ka.out_stride = seq; ka.res_mode = 2; }
    return seeme_launch_linear(ka, st);
}

extern "C" size_t seeme_pointnet_workspace_bytes(int B, int P) {
    const size_t M = (size_t)B * P;
    return (M * (512 + 3 * 256) + (size_t)B * 256 * 4) * sizeof(float) + 256;
}

// ResnetPointnet.forward (respointnet.py:33-59).  The concat [net, pooled] never exists: relu distributes
// over the concat, so the pooled half of fc_0 / shortcut is a per-scene 256-vector added as a broadcast
// residual: 36.8 GF/scene executed instead of 52.5.
extern "C" int seeme_pointnet_encode(const SeemePointnetWeights* w, const float* points, int B, int P, float* out,
                                     void* workspace, size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (B <= 0 || P <= 0) return seeme_fail("pointnet: empty input");
    if (ws_bytes < seeme_pointnet_workspace_bytes(B, P)) return seeme_fail("pointnet: workspace too small");
    if (B > 65535) return seeme_fail("pointnet: B too large");
    const int M = B * P, H = 256;
    float* x512 = (float*)workspace;
    float* net = x512 + (size_t)M * 512;      // current features [M,256]
    float* hid = net + (size_t)M * H;         // fc_0 output
    float* xs = hid + (size_t)M * H;          // shortcut output
    float* pool = xs + (size_t)M * H;         // [B,256]
    float* v0 = pool + (size_t)B * H;         // pooled halves, [B,256] each
    float* vs = v0 + (size_t)B * H;
    float* pool2 = vs + (size_t)B * H;
    int rc;
    // fc_pos_0: 3 -> 512 (weights zero padded to K = 16)
    if ((rc = lin(st, points, 3, w->pos_w, 16, w->pos_b, nullptr, 0, 0, x512, M, 512, 3, SEEME_ACT_NONE))) return rc;
    // block_0 on the 512-wide features
    if ((rc = lin(st, x512, 512, w->fc0_w[0], 512, w->fc0_b[0], nullptr, 0, 0, hid, M, H, 512, SEEME_ACT_RELU))) return rc;
    if ((rc = lin(st, x512, 512, w->sc_w[0], 512, nullptr, nullptr, 0, 0, xs, M, H, 512, SEEME_ACT_NONE))) return rc;
    if ((rc = lin(st, hid, H, w->fc1_w[0], H, w->fc1_b[0], xs, 0, 0, net, M, H, H, SEEME_ACT_RELU))) return rc;
    for (int i = 1; i < 4; ++i) {
        hipLaunchKernelGGL(k_colmax, dim3((H + 63) / 64, B), dim3(256), 0, st, net, pool, P, H);
        if ((rc = seeme_check_launch("k_colmax"))) return rc;
        // pooled halves: v0 = W0[:,256:] relu(pool), vs = Ws[:,256:] pool
        if ((rc = lin(st, pool, H, w->fc0_w[i] + H, 512, nullptr, nullptr, 0, 0, v0, B, H, H, SEEME_ACT_RELU))) return rc;
        if ((rc = lin(st, pool, H, w->sc_w[i] + H, 512, nullptr, nullptr, 0, 0, vs, B, H, H, SEEME_ACT_NONE))) return rc;
        if ((rc = lin(st, net, H, w->fc0_w[i], 512, w->fc0_b[i], v0, 2, P, hid, M, H, H, SEEME_ACT_RELU))) return rc;
        if ((rc = lin(st, net, H, w->sc_w[i], 512, nullptr, vs, 2, P, xs, M, H, H, SEEME_ACT_NONE))) return rc;
        if ((rc = lin(st, hid, H, w->fc1_w[i], H, w->fc1_b[i], xs, 0, 0, net, M, H, H, SEEME_ACT_RELU))) return rc;
    }
    hipLaunchKernelGGL(k_colmax, dim3((H + 63) / 64, B), dim3(256), 0, st, net, pool2, P, H);
    if ((rc = seeme_check_launch("k_colmax"))) return rc;
    // fc_c(relu(pooled))  -> [B, out_dim]
    return lin(st, pool2, H, w->fcc_w, H, w->fcc_b, nullptr, 0, 0, out, B, w->out_dim, H, SEEME_ACT_RELU);
}

// ---------------------------------------------------------------------------------------------
// AdamW over a list of tensors in ONE launch (torch.optim.AdamW arithmetic, amsgrad off).  The reference trains with
// Lightning's default AdamW (mld/models/modeltype/base.py: configure_optimizers); PyTorch's multi-tensor path spends
// 27 launches and ~0.5 ms on the 204 trainable tensors (8.0 M elements) of stage 2, a pure HBM stream of 224 MB.
// chunks[c] = {tensor id, element offset, element count}; p / m / v / g = per-tensor base pointers.
struct AdamWChunk { int tensor, count; long long offset; };
// DEV: the step count and the learning rate come from device memory (dev = {step, lr}), so that the launch can sit in a
// captured hipGraph and still advance from replay to replay; the scalar factors are then formed here, in double.
template <bool DEV>
__global__ __launch_bounds__(256) void k_adamw(const AdamWChunk* __restrict__ chunks, float* const* __restrict__ ps,
                                               const float* const* __restrict__ gs, float* const* __restrict__ ms,
                                               float* const* __restrict__ vs, float decay, float one_minus_beta1, float beta2,
                                               float one_minus_beta2, float step_size, float sqrt_bias_c2, float eps,
                                               const float* __restrict__ dev, double beta1d, double beta2d, double wd) {
    if (DEV) {
        const double step = (double)dev[0], lr = (double)dev[1];
        decay = (float)(1.0 - lr * wd);
        step_size = (float)(lr / (1.0 - pow(beta1d, step)));
        sqrt_bias_c2 = (float)sqrt(1.0 - pow(beta2d, step));
    }
    const AdamWChunk ch = chunks[blockIdx.x];
    float* __restrict__ p = ps[ch.tensor] + ch.offset;
    const float* __restrict__ g = gs[ch.tensor] + ch.offset;
    float* __restrict__ m = ms[ch.tensor] + ch.offset;
    float* __restrict__ v = vs[ch.tensor] + ch.offset;
    // the scalar factors (1 - lr wd, 1 - beta, lr / bias_correction1) are formed in double on the host, as PyTorch does
    auto upd = [&](float& pw, float gw, float& mw, float& vw) {
        pw *= decay;                                               // p.mul_(1 - lr * wd)
        mw = mw + (gw - mw) * one_minus_beta1;                     // exp_avg.lerp_(grad, 1 - beta1)
        vw = vw * beta2 + gw * gw * one_minus_beta2;               // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
        pw -= step_size * (mw / (sqrtf(vw) / sqrt_bias_c2 + eps)); // p.addcdiv_(exp_avg, denom, -step_size)
    };
    const bool vec = ((reinterpret_cast<size_t>(p) | reinterpret_cast<size_t>(g) | reinterpret_cast<size_t>(m) | reinterpret_cast<size_t>(v)) & 15) == 0;
    if (vec) {
        const int n4 = ch.count >> 2;
        for (int i = threadIdx.x; i < n4; i += 256) {
            float4 pw = reinterpret_cast<float4*>(p)[i], mw = reinterpret_cast<float4*>(m)[i], vw = reinterpret_cast<float4*>(v)[i];
            const float4 gw = reinterpret_cast<const float4*>(g)[i];
            upd(pw.x, gw.x, mw.x, vw.x); upd(pw.y, gw.y, mw.y, vw.y); upd(pw.z, gw.z, mw.z, vw.z); upd(pw.w, gw.w, mw.w, vw.w);
            reinterpret_cast<float4*>(p)[i] = pw; reinterpret_cast<float4*>(m)[i] = mw; reinterpret_cast<float4*>(v)[i] = vw;
        }
        for (int i = (n4 << 2) + threadIdx.x; i < ch.count; i += 256) upd(p[i], g[i], m[i], v[i]);
    } else {
        for (int i = threadIdx.x; i < ch.count; i += 256) upd(p[i], g[i], m[i], v[i]);
    }
}

extern "C" int seeme_adamw_step(const void* chunks, int n_chunks, void* const* params, const void* const* grads, void* const* exp_avg,
                                void* const* exp_avg_sq, double lr, double beta1, double beta2, double eps, double weight_decay,
                                double step, void* stream) {
    if (n_chunks <= 0) return 0;
    if (!(step >= 1.0)) return seeme_fail("adamw: step must be >= 1");
    const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
    hipLaunchKernelGGL(k_adamw<false>, dim3((unsigned)n_chunks), dim3(256), 0, (hipStream_t)stream, (const AdamWChunk*)chunks,
                       (float* const*)params, (const float* const*)grads, (float* const*)exp_avg, (float* const*)exp_avg_sq,
                       (float)(1.0 - lr * weight_decay), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2),
                       (float)(lr / bc1), (float)sqrt(bc2), (float)eps, (const float*)nullptr, 0.0, 0.0, 0.0);
    return seeme_check_launch("k_adamw");
}

extern "C" int seeme_adamw_step_dev(const void* chunks, int n_chunks, void* const* params, const void* const* grads,
                                    void* const* exp_avg, void* const* exp_avg_sq, const float* step_lr, double beta1, double beta2,
                                    double eps, double weight_decay, void* stream) {
    if (n_chunks <= 0) return 0;
    if (!step_lr) return seeme_fail("adamw: step_lr is NULL");
    hipLaunchKernelGGL(k_adamw<true>, dim3((unsigned)n_chunks), dim3(256), 0, (hipStream_t)stream, (const AdamWChunk*)chunks,
                       (float* const*)params, (const float* const*)grads, (float* const*)exp_avg, (float* const*)exp_avg_sq,
                       0.f, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), 0.f, 0.f, (float)eps, step_lr, beta1, beta2,
                       weight_decay);
    return seeme_check_launch("k_adamw<dev>");
}
