// smpl_kernels.hip -- SMPL linear blend skinning (smplx.SMPL.forward / smplx.lbs.lbs restated,
// SURVEY.md App. C; call sites mld/models/modeltype/mld.py:764-770 etc.).
//
// k_smpl_joints : one wave per frame.  Rodrigues, joint regression in the collapsed form
//                 J = J_template + (J_regressor . shapedirs) beta (App. E7), kinematic chain, the 24
//                 posed joints and the 21 vertex-picked extra joints from a compact 21-vertex model.
//                 No mesh is formed: this is the whole job for training / metrics (joints[:24]).
// full mesh     : blend shapes as ONE fp32-MFMA GEMM  [M,224] x [V*3,224]^T (+ v_template as bias)
//                 through k_linear, then k_smpl_skin applies the per-vertex blended transforms.
#include "common.hpp"
#include "api_util.hpp"

#define SMPL_J 24
#define SMPL_EX 21
#define SMPL_FEAT 224   // 10 betas + 207 pose-blend features, zero padded to a multiple of 16

struct SmplKArgs {
    SeemeSmplModel m;
    const float* betas; const float* pose; const float* transl;
    int pose_is_rotmat, M;
    float* joints;      // [M,45,3]
    float* A;           // [M,24,12] skinning transforms (rows of [R|t]); may be NULL
    float* feat;        // [M,224] blend features; may be NULL
};

__device__ __forceinline__ void mat34_mul(const float* __restrict__ P, const float* __restrict__ Q, float* __restrict__ O) {
    // O = P * Q for affine 3x4 matrices (implicit last row 0 0 0 1)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v = P[r * 4 + 0] * Q[0 * 4 + c] + P[r * 4 + 1] * Q[1 * 4 + c] + P[r * 4 + 2] * Q[2 * 4 + c];
            if (c == 3) v += P[r * 4 + 3];
            O[r * 4 + c] = v;
        }
    }
}

__global__ __launch_bounds__(256) void k_smpl_joints(const SmplKArgs a) {
    __shared__ float sT[4][SMPL_J][12];     // local transforms [R_j | J_j - J_parent]
    __shared__ float sA[4][SMPL_J][12];     // skinning transforms
    __shared__ float sJ[4][SMPL_J][3];      // rest joints
    __shared__ float sPF[4][208];           // pose-blend features
    __shared__ float sV[4][64];             // posed extra vertices (63 coords)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int mraw = blockIdx.x * 4 + wave;
    const bool live = mraw < a.M;
    const int m = live ? mraw : a.M - 1;
    const float* beta = a.betas + (size_t)m * 10;

    if (lane < SMPL_J) {
        float R[9];
        if (a.pose_is_rotmat) {
#pragma unroll
            for (int i = 0; i < 9; ++i) R[i] = a.pose[((size_t)m * SMPL_J + lane) * 9 + i];
        } else {  // smplx.lbs.batch_rodrigues
            const float* r = a.pose + ((size_t)m * SMPL_J + lane) * 3;
            const float rx = r[0], ry = r[1], rz = r[2];
            const float ex = rx + 1e-8f, ey = ry + 1e-8f, ez = rz + 1e-8f;
            const float ang = sqrtf(ex * ex + ey * ey + ez * ez);
            const float dx = rx / ang, dy = ry / ang, dz = rz / ang;
            const float s = sinf(ang), c1 = 1.f - cosf(ang);
            // K = [[0,-dz,dy],[dz,0,-dx],[-dy,dx,0]],  R = I + s K + (1-c) K^2
            const float K[9] = {0.f, -dz, dy, dz, 0.f, -dx, -dy, dx, 0.f};
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const float kk = K[i * 3 + 0] * K[0 * 3 + j] + K[i * 3 + 1] * K[1 * 3 + j] + K[i * 3 + 2] * K[2 * 3 + j];
                    R[i * 3 + j] = (i == j ? 1.f : 0.f) + s * K[i * 3 + j] + c1 * kk;
                }
        }
        float J[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = a.m.J_template[lane * 3 + c];
#pragma unroll
            for (int l = 0; l < 10; ++l) v = fmaf(a.m.J_shapedirs[(lane * 3 + c) * 10 + l], beta[l], v);
            J[c] = v;
            sJ[wave][lane][c] = v;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) sT[wave][lane][i * 4 + j] = R[i * 3 + j];
        if (lane >= 1) {  // pose feature (R - I) of joints 1..23
#pragma unroll
            for (int i = 0; i < 9; ++i) sPF[wave][(lane - 1) * 9 + i] = R[i] - ((i == 0 || i == 4 || i == 8) ? 1.f : 0.f);
        }
        (void)J;
    }
    __syncthreads();
    if (lane < SMPL_J) {
        const int p = a.m.parents[lane];
#pragma unroll
        for (int c = 0; c < 3; ++c)
            sT[wave][lane][c * 4 + 3] = sJ[wave][lane][c] - (p >= 0 ? sJ[wave][p][c] : 0.f);
    }
    __syncthreads();
    const float tx = a.transl ? a.transl[(size_t)m * 3 + 0] : 0.f;
    const float ty = a.transl ? a.transl[(size_t)m * 3 + 1] : 0.f;
    const float tz = a.transl ? a.transl[(size_t)m * 3 + 2] : 0.f;
    if (lane < SMPL_J) {
        // G_j = T_root ... T_parent T_j, accumulated leaf-to-root (batch_rigid_transform)
        float G[12], N2[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) G[i] = sT[wave][lane][i];
        int p = a.m.parents[lane];
        for (int hop = 0; hop < SMPL_J && p >= 0; ++hop) {
            mat34_mul(&sT[wave][p][0], G, N2);
#pragma unroll
            for (int i = 0; i < 12; ++i) G[i] = N2[i];
            p = a.m.parents[p];
        }
        if (live) {
            float* jo = a.joints + ((size_t)m * 45 + lane) * 3;
            jo[0] = G[3] + tx; jo[1] = G[7] + ty; jo[2] = G[11] + tz;
        }
        // A_j = G_j with translation  t - R J_j   (rel_transforms)
        const float jx = sJ[wave][lane][0], jy = sJ[wave][lane][1], jz = sJ[wave][lane][2];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float tr = G[r * 4 + 3] - (G[r * 4 + 0] * jx + G[r * 4 + 1] * jy + G[r * 4 + 2] * jz);
            sA[wave][lane][r * 4 + 0] = G[r * 4 + 0];
            sA[wave][lane][r * 4 + 1] = G[r * 4 + 1];
            sA[wave][lane][r * 4 + 2] = G[r * 4 + 2];
            sA[wave][lane][r * 4 + 3] = tr;
        }
    }
    __syncthreads();
    if (live && a.A != nullptr)
        for (int i = lane; i < SMPL_J * 12; i += 64) a.A[(size_t)m * (SMPL_J * 12) + i] = (&sA[wave][0][0])[i];
    if (live && a.feat != nullptr)
        for (int i = lane; i < SMPL_FEAT; i += 64)
            a.feat[(size_t)m * SMPL_FEAT + i] = i < 10 ? beta[i] : (i < 217 ? sPF[wave][i - 10] : 0.f);
    // ---- 21 extra joints = posed vertices picked by id (compact model)
    if (lane < SMPL_EX * 3) {
        float v = a.m.ex_template[lane];
#pragma unroll
        for (int l = 0; l < 10; ++l) v = fmaf(a.m.ex_shapedirs[lane * 10 + l], beta[l], v);
        for (int k = 0; k < 207; ++k) v = fmaf(sPF[wave][k], a.m.ex_posedirs[k * 63 + lane], v);
        sV[wave][lane] = v;
    }
    __syncthreads();
    if (live && lane < SMPL_EX) {
        float T[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) T[i] = 0.f;
        for (int j = 0; j < SMPL_J; ++j) {
            const float w = a.m.ex_weights[lane * SMPL_J + j];
#pragma unroll
            for (int i = 0; i < 12; ++i) T[i] = fmaf(w, sA[wave][j][i], T[i]);
        }
        const float vx = sV[wave][lane * 3 + 0], vy = sV[wave][lane * 3 + 1], vz = sV[wave][lane * 3 + 2];
        float* jo = a.joints + ((size_t)m * 45 + SMPL_J + lane) * 3;
        jo[0] = T[0] * vx + T[1] * vy + T[2] * vz + T[3] + tx;
        jo[1] = T[4] * vx + T[5] * vy + T[6] * vz + T[7] + ty;
        jo[2] = T[8] * vx + T[9] * vy + T[10] * vz + T[11] + tz;
    }
}

// verts[m][v] (holding v_posed) <- (sum_j w[v][j] A[m][j]) * [v_posed;1] + transl
__global__ __launch_bounds__(256) void k_smpl_skin(const float* __restrict__ w, const float* __restrict__ A,
                                                   const float* __restrict__ transl, float* __restrict__ verts, int V,
                                                   int nvb) {
    __shared__ float sA[SMPL_J * 12];
    const int m = blockIdx.x / nvb, vb = blockIdx.x - m * nvb;
    for (int i = threadIdx.x; i < SMPL_J * 12; i += 256) sA[i] = A[(size_t)m * (SMPL_J * 12) + i];
    __syncthreads();
    const int v = vb * 256 + threadIdx.x;
    if (v >= V) return;
    float T[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = 0.f;
    const float* wr = w + (size_t)v * SMPL_J;
#pragma unroll 4
    for (int j = 0; j < SMPL_J; ++j) {
        const float wj = wr[j];
#pragma unroll
        for (int i = 0; i < 12; ++i) T[i] = fmaf(wj, sA[j * 12 + i], T[i]);
    }
    float* p = verts + ((size_t)m * V + v) * 3;
    const float x = p[0], y = p[1], z = p[2];
    const float tx = transl ? transl[(size_t)m * 3 + 0] : 0.f;
    const float ty = transl ? transl[(size_t)m * 3 + 1] : 0.f;
    const float tz = transl ? transl[(size_t)m * 3 + 2] : 0.f;
    p[0] = T[0] * x + T[1] * y + T[2] * z + T[3] + tx;
    p[1] = T[4] * x + T[5] * y + T[6] * z + T[7] + ty;
    p[2] = T[8] * x + T[9] * y + T[10] * z + T[11] + tz;
}

extern "C" size_t seeme_smpl_workspace_bytes(int M) {
    return (size_t)M * (SMPL_J * 12 + SMPL_FEAT) * sizeof(float) + 256;
}

extern "C" int seeme_smpl_lbs(const SeemeSmplModel* model, const float* betas, const float* pose, int pose_is_rotmat,
                              const float* transl, int M, float* joints, float* vertices,
                              void* workspace, size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (M <= 0) return seeme_fail("smpl_lbs: M must be > 0");
    SmplKArgs a{};
    a.m = *model; a.betas = betas; a.pose = pose; a.transl = transl; a.pose_is_rotmat = pose_is_rotmat; a.M = M;
    a.joints = joints;
    if (vertices != nullptr) {
        if (ws_bytes < seeme_smpl_workspace_bytes(M)) return seeme_fail("smpl_lbs: workspace too small");
        a.A = (float*)workspace;
        a.feat = a.A + (size_t)M * SMPL_J * 12;
    }
    hipLaunchKernelGGL(k_smpl_joints, dim3((M + 3) / 4), dim3(256), 0, st, a);
    int rc = seeme_check_launch("k_smpl_joints");
    if (rc || vertices == nullptr) return rc;
    const int N = model->V * 3;
    // v_posed = v_template + [beta | pose_feature] @ blend_w^T   (lbs steps 1 and 4 as one GEMM)
    rc = seeme_linear_simple(st, a.feat, SMPL_FEAT, model->blend_w, SMPL_FEAT, model->v_template, vertices, N, M, N,
                             SMPL_FEAT, SEEME_ACT_NONE, SEEME_ACT_NONE, nullptr, nullptr);
    if (rc) return rc;
    const int nvb = (model->V + 255) / 256;
    hipLaunchKernelGGL(k_smpl_skin, dim3((unsigned)nvb * (unsigned)M), dim3(256), 0, st, model->lbs_weights, a.A, transl,
                       vertices, model->V, nvb);
    return seeme_check_launch("k_smpl_skin");
}

// ---------------------------------------------------------------------------------------------
// Backward of the 24 posed joints w.r.t. the axis-angle pose and the translation (stage-1 training: the joints loss of
// train_vae_forward, mld.py:764-773,871-878, goes through smplx's lbs).  One wave per frame, lanes <-> joints:
//   forward again (Rodrigues, rest joints, world rotations W_j = W_parent R_j), then
//   p_j = p_parent + W_parent rel_j  gives, walking the tree from the leaves (one lane, 23 steps):
//     g_p[parent] += g_p[j];  g_W[parent] += g_p[j] rel_j^T + g_W[j] R_j^T;  g_R[j] = W_parent^T g_W[j]
//   and every lane turns its g_R into the gradient of its axis-angle vector through d(Rodrigues)/d(aa).
__global__ __launch_bounds__(256) void k_smpl_joints_bwd(const SeemeSmplModel m, const float* __restrict__ betas,
                                                         const float* __restrict__ pose, const float* __restrict__ djoints,
                                                         int dj_stride, float* __restrict__ dpose, float* __restrict__ dtransl, int M) {
    __shared__ float sR[4][SMPL_J][9], sW[4][SMPL_J][9], sJ[4][SMPL_J][3], sRel[4][SMPL_J][3];
    __shared__ float sGp[4][SMPL_J][3], sGW[4][SMPL_J][9], sGR[4][SMPL_J][9];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int mraw = blockIdx.x * 4 + wave;
    const bool live = mraw < M;
    const int f = live ? mraw : M - 1;
    const float* beta = betas + (size_t)f * 10;
    float aa[3] = {0.f, 0.f, 0.f};
    if (lane < SMPL_J) {
        const float* r = pose + ((size_t)f * SMPL_J + lane) * 3;
        aa[0] = r[0]; aa[1] = r[1]; aa[2] = r[2];
        const float ex = aa[0] + 1e-8f, ey = aa[1] + 1e-8f, ez = aa[2] + 1e-8f;
        const float ang = sqrtf(ex * ex + ey * ey + ez * ez);
        const float dx = aa[0] / ang, dy = aa[1] / ang, dz = aa[2] / ang;
        const float s = sinf(ang), c1 = 1.f - cosf(ang);
        const float K[9] = {0.f, -dz, dy, dz, 0.f, -dx, -dy, dx, 0.f};
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float kk = K[i * 3 + 0] * K[0 * 3 + j] + K[i * 3 + 1] * K[1 * 3 + j] + K[i * 3 + 2] * K[2 * 3 + j];
                sR[wave][lane][i * 3 + j] = (i == j ? 1.f : 0.f) + s * K[i * 3 + j] + c1 * kk;
            }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = m.J_template[lane * 3 + c];
#pragma unroll
            for (int l = 0; l < 10; ++l) v = fmaf(m.J_shapedirs[(lane * 3 + c) * 10 + l], beta[l], v);
            sJ[wave][lane][c] = v;
            sGp[wave][lane][c] = djoints[((size_t)f * dj_stride + lane) * 3 + c];
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) sGW[wave][lane][i] = 0.f;
    }
    __syncthreads();
    if (lane < SMPL_J) {
        const int p = m.parents[lane];
#pragma unroll
        for (int c = 0; c < 3; ++c) sRel[wave][lane][c] = sJ[wave][lane][c] - (p >= 0 ? sJ[wave][p][c] : 0.f);
        // world rotation: W_j = R_root ... R_parent R_j (leaf-to-root accumulation, as the forward kernel)
        float W[9], N2[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) W[i] = sR[wave][lane][i];
        int q = p;
        for (int hop = 0; hop < SMPL_J && q >= 0; ++hop) {
            const float* Rq = &sR[wave][q][0];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) N2[i * 3 + j] = Rq[i * 3 + 0] * W[0 * 3 + j] + Rq[i * 3 + 1] * W[1 * 3 + j] + Rq[i * 3 + 2] * W[2 * 3 + j];
#pragma unroll
            for (int i = 0; i < 9; ++i) W[i] = N2[i];
            q = m.parents[q];
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) sW[wave][lane][i] = W[i];
    }
    __syncthreads();
    if (lane == 0) {
        float tsum[3] = {0.f, 0.f, 0.f};
        for (int j = 0; j < SMPL_J; ++j)
            for (int c = 0; c < 3; ++c) tsum[c] += sGp[wave][j][c];
        if (live && dtransl != nullptr)
            for (int c = 0; c < 3; ++c) dtransl[(size_t)f * 3 + c] = tsum[c];
        for (int j = SMPL_J - 1; j >= 1; --j) {
            const int p = m.parents[j];
            const float* gp = &sGp[wave][j][0];
            const float* gW = &sGW[wave][j][0];
            const float* Rj = &sR[wave][j][0];
            const float* Wp = &sW[wave][p][0];
            for (int a = 0; a < 3; ++a) {
                sGp[wave][p][a] += gp[a];
                for (int b = 0; b < 3; ++b) {
                    // g_W[p] += g_p[j] rel_j^T + g_W[j] R_j^T ;  g_R[j] = W_p^T g_W[j]
                    sGW[wave][p][a * 3 + b] += gp[a] * sRel[wave][j][b] + gW[a * 3 + 0] * Rj[b * 3 + 0] + gW[a * 3 + 1] * Rj[b * 3 + 1] + gW[a * 3 + 2] * Rj[b * 3 + 2];
                    sGR[wave][j][a * 3 + b] = Wp[0 * 3 + a] * gW[0 * 3 + b] + Wp[1 * 3 + a] * gW[1 * 3 + b] + Wp[2 * 3 + a] * gW[2 * 3 + b];
                }
            }
        }
        for (int i = 0; i < 9; ++i) sGR[wave][0][i] = sGW[wave][0][i];
    }
    __syncthreads();
    if (live && lane < SMPL_J) {
        const float* g = &sGR[wave][lane][0];
        const float e[3] = {aa[0] + 1e-8f, aa[1] + 1e-8f, aa[2] + 1e-8f};
        const float th = sqrtf(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
        const float d[3] = {aa[0] / th, aa[1] / th, aa[2] / th};
        const float s = sinf(th), c = cosf(th), c1 = 1.f - c;
        const float dd = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        const float tr = g[0] + g[4] + g[8];
        const float w[3] = {g[7] - g[5], g[2] - g[6], g[3] - g[1]};                       // sum(gR o K(v)) = v . w
        float Gd[3], Gtd[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            Gd[a] = g[a * 3 + 0] * d[0] + g[a * 3 + 1] * d[1] + g[a * 3 + 2] * d[2];
            Gtd[a] = g[0 * 3 + a] * d[0] + g[1 * 3 + a] * d[1] + g[2 * 3 + a] * d[2];
        }
        const float dGd = d[0] * Gd[0] + d[1] * Gd[1] + d[2] * Gd[2];
        const float dw = d[0] * w[0] + d[1] * w[1] + d[2] * w[2];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float thk = e[k] / th;
            float pd[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) pd[j] = ((j == k ? 1.f : 0.f) - d[j] * thk) / th;
            const float pdw = pd[0] * w[0] + pd[1] * w[1] + pd[2] * w[2];
            const float pdGd = pd[0] * Gd[0] + pd[1] * Gd[1] + pd[2] * Gd[2];
            const float Gtdpd = Gtd[0] * pd[0] + Gtd[1] * pd[1] + Gtd[2] * pd[2];
            const float dpd = d[0] * pd[0] + d[1] * pd[1] + d[2] * pd[2];
            dpose[((size_t)f * SMPL_J + lane) * 3 + k] = c * thk * dw + s * pdw + s * thk * (dGd - dd * tr) + c1 * (pdGd + Gtdpd - 2.f * dpd * tr);
        }
    }
}

extern "C" int seeme_smpl_joints_backward(const SeemeSmplModel* model, const float* betas, const float* pose, const float* djoints,
                                          int dj_stride, float* dpose, float* dtransl, int M, void* stream) {
    if (!model || !betas || !pose || !djoints || !dpose || M <= 0 || dj_stride < SMPL_J) return seeme_fail("smpl_joints_backward: bad arguments");
    hipLaunchKernelGGL(k_smpl_joints_bwd, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, *model, betas, pose, djoints, dj_stride,
                       dpose, dtransl, M);
    return seeme_check_launch("k_smpl_joints_bwd");
}
