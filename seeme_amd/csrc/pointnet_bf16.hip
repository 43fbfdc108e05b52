// pointnet_bf16.hip -- ResNet-PointNet scene encoder (EgoHMR/models/respointnet.py:33-97) as fused
// bf16-MFMA kernels: the one part of the path with GEMMs big enough for the matrix cores
// (M = B x 20 000 points).
//
// One workgroup (4 waves) owns a tile of 128 points of one scene and runs a whole ResnetBlockFC on it:
//     hid = relu( W0[:, :256] relu(x) + W0[:, 256:] relu(pool) + b0 )          fc_0
//     out =       Ws[:, :256] x       + Ws[:, 256:] pool
//               + W1 hid + b1                                                   shortcut + fc_1, ONE accumulator
// The point features stay in LDS as bf16 (raw and relu'd copies), the hidden tile overwrites the relu'd copy,
// weights stream from L2 as bf16 rows (each wave reads its own 16-byte B fragments), accumulation is fp32
// (v_mfma_f32_16x16x32_bf16).  The per-scene max-pool of the block output is folded into the epilogue
// (tile max -> one atomic per column), so no pass over the [M,256] tensor is spent on pooling; activations
// travel between blocks as bf16 (half the HBM bytes).  The pooled halves are per-scene fp32 vectors
//  produced by the small fp32 linear kernel.
#include "common.hpp"
#include "api_util.hpp"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define PN_MT 128          // points per workgroup
#define PN_THREADS 512      // 8 waves
#define PN_H 256           // hidden width
#define PN_PADB 16         // bf16 elements of row padding (2 x 16-byte slots: conflict-free ds_read_b128)

__device__ __forceinline__ unsigned short f2bf(float x) { return __builtin_bit_cast(unsigned short, (__bf16)x); }
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// acc[mt][nt] += A[128 rows, K] (bf16 LDS, row stride lda elements) * W^T, W pre-packed in MFMA fragment order:
// Wp[(ntile * kstride + kb) * 64 + lane] = the 8 bf16 W[n = ntile*16 + (lane&15)][k = kb*32 + 8*(lane>>4) .. +7],
// so every wave-load is 1 KiB contiguous (fragment-shaped loads from a row-major matrix touch 16 lines of
// 64 B per instruction and are address-unit bound; cdna_hip_programming.md section 5, "glds vs register staging").
template <int MTL, int NTL>
__device__ __forceinline__ void tile_gemm_bf16(const unsigned short* __restrict__ As, int lda,
                                               const uint4* __restrict__ Wp, int kstride, int ntile0, int K32,
                                               f32x4 (&acc)[MTL][NTL]) {
    const int lane = threadIdx.x & 63, r = lane & 15, kq = lane >> 4;
    const uint4* wp[NTL];
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt) wp[nt] = Wp + (size_t)(ntile0 + nt) * kstride * 64 + lane;
    const unsigned short* ap = As + r * lda + 8 * kq;
    constexpr int PF = 4;
    uint4 br[PF][NTL];
#pragma unroll
    for (int u = 0; u < PF; ++u)
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) br[u][nt] = wp[nt][(u < K32 ? u : K32 - 1) * 64];
    // A fragments of the next k-block are read from LDS while the MFMAs of the current one issue
    uint4 an[MTL];
#pragma unroll
    for (int mt = 0; mt < MTL; ++mt) an[mt] = *reinterpret_cast<const uint4*>(ap + mt * 16 * lda);
    for (int kb0 = 0; kb0 < K32; kb0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int kb = kb0 + u;
            if (kb < K32) {
                bf16x8 b[NTL], a[MTL];
#pragma unroll
                for (int nt = 0; nt < NTL; ++nt) b[nt] = __builtin_bit_cast(bf16x8, br[u][nt]);
#pragma unroll
                for (int mt = 0; mt < MTL; ++mt) a[mt] = __builtin_bit_cast(bf16x8, an[mt]);
                const int kn = (kb + PF < K32) ? kb + PF : K32 - 1;
#pragma unroll
                for (int nt = 0; nt < NTL; ++nt) br[u][nt] = wp[nt][kn * 64];
                const int ka = (kb + 1 < K32) ? kb + 1 : kb;
#pragma unroll
                for (int mt = 0; mt < MTL; ++mt) an[mt] = *reinterpret_cast<const uint4*>(ap + mt * 16 * lda + ka * 32);
#pragma unroll
                for (int mt = 0; mt < MTL; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NTL; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
            }
        }
    }
}

// float atomic max through the ordered-integer trick (destination initialised to -inf)
__device__ __forceinline__ void atomic_max_f32(float* p, float v) {
    if (v >= 0.f) atomicMax(reinterpret_cast<int*>(p), __float_as_int(v));
    else atomicMin(reinterpret_cast<unsigned int*>(p), __float_as_uint(v));
}

struct PnBlockArgs {
    // input: FIRST block -> points [B,P,3] + fc_pos_0 (fp32 [512,3 (ld 16)], [512]); later blocks -> x bf16 [B*P,256]
    const float* points; const float* pos_w; const float* pos_b;
    const unsigned short* x;
    const uint4* w0; int ks0;               // fc_0, fragment-packed [16 n-tiles][ks0 k-blocks][64 lanes] (ks0 = 16: K = 512 packed; later blocks use k-blocks 0..7)
    const float* b0;
    const float* v0;                        // [B,256] pooled half of fc_0 (NULL in the first block)
    const uint4* w1;                        // fc_1, fragment-packed, 8 k-blocks
    const float* b1;
    const uint4* ws; int kss;               // shortcut, fragment-packed like fc_0
    const float* vs;                        // [B,256] pooled half of the shortcut (NULL in the first block)
    unsigned short* out;                    // [B*P,256] bf16 block output (may be NULL for the last block)
    float* pool;                            // [B,256] running max of the block output (pre-initialised to -inf)
    int P, first;
};

template <bool FIRST>
__global__ __launch_bounds__(PN_THREADS) void k_pn_block(const PnBlockArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // 8 waves = 2 row halves x 4 column quarters: two waves per SIMD, so one wave's MFMAs cover the other's LDS / L2 latency
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 15, kq = lane >> 4;
    const int mh = wave >> 2, nq = wave & 3, row0 = mh * (PN_MT / 2);
    const int scene = blockIdx.y, p0 = blockIdx.x * PN_MT;
    const int rows_valid = min(PN_MT, a.P - p0);
    constexpr int K = FIRST ? 512 : 256;
    constexpr int LDA = K + PN_PADB;                      // bf16 elements
    constexpr int LDH = PN_H + PN_PADB;
    // FIRST: one [128][528] tile regenerated from the points (relu'd, later raw); the hidden tile aliases it.
    // later: raw tile | relu tile; the hidden tile overwrites the relu tile.
    unsigned short* T0 = reinterpret_cast<unsigned short*>(smem_raw);
    unsigned short* T1 = FIRST ? T0 : T0 + PN_MT * LDA;
    float* Cs = reinterpret_cast<float*>(smem_raw);      // [128][264] fp32 epilogue tile (everything else dead by then)
    __shared__ float spts[PN_MT * 3];

    // x512 = fc_pos_0(p) (K = 3: plain FMAs) as bf16 into T0: thread <-> columns tid, tid+256 (weights in
    // registers), points broadcast from LDS, consecutive lanes write consecutive 2-byte elements
    float pw[3] = {0.f, 0.f, 0.f}, pb = 0.f;
    if (FIRST) {
        pw[0] = a.pos_w[tid * 16 + 0]; pw[1] = a.pos_w[tid * 16 + 1]; pw[2] = a.pos_w[tid * 16 + 2];
        pb = a.pos_b[tid];
    }
    auto gen_first = [&](bool do_relu) {
        for (int row = 0; row < PN_MT; ++row) {
            const float px = spts[row * 3 + 0], py = spts[row * 3 + 1], pz = spts[row * 3 + 2];
            float v = row < rows_valid ? pb + pw[0] * px + pw[1] * py + pw[2] * pz : 0.f;
            if (do_relu) v = fmaxf(v, 0.f);
            T0[row * LDA + tid] = f2bf(v);
        }
    };

    if (FIRST) {
        for (int i = tid; i < PN_MT * 3; i += PN_THREADS)
            spts[i] = (i / 3 < rows_valid) ? a.points[((size_t)scene * a.P + p0) * 3 + i] : 0.f;
        __syncthreads();
        gen_first(true);
    } else {
        // stage raw + relu copies of the bf16 input tile (8 bf16 = 16 B per thread step)
        const unsigned short* xin = a.x + ((size_t)scene * a.P + p0) * PN_H;
        // all 16 row-loads of a thread are in flight together (the tile is one HBM round trip, not sixteen)
        constexpr int NIT = PN_MT * (PN_H / 8) / PN_THREADS;
        uint4 v[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * PN_THREADS, row = idx >> 5, c8 = (idx & 31) * 8;
            v[it] = make_uint4(0u, 0u, 0u, 0u);
            if (row < rows_valid) v[it] = *reinterpret_cast<const uint4*>(xin + (size_t)row * PN_H + c8);
        }
        auto relu2 = [](unsigned u) { const unsigned m = (u >> 15) & 0x00010001u; return u & ~(m * 0xFFFFu); };   // packed bf16 relu
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * PN_THREADS, row = idx >> 5, c8 = (idx & 31) * 8;
            *reinterpret_cast<uint4*>(T0 + row * LDA + c8) = v[it];
            *reinterpret_cast<uint4*>(T1 + row * LDA + c8) = make_uint4(relu2(v[it].x), relu2(v[it].y), relu2(v[it].z), relu2(v[it].w));
        }
    }
    __syncthreads();

    // ---- fc_0 on relu(x): wave w owns hidden columns [64w, 64w+64)
    f32x4 acc[4][4];
    acc_zero(acc);
    tile_gemm_bf16<4, 4>((FIRST ? T0 : T1) + row0 * LDA, LDA, a.w0, a.ks0, nq * 4, K / 32, acc);
    __syncthreads();                                       // all waves done reading the relu tile
    {
        unsigned short* Hs = FIRST ? T0 : T1;              // hidden tile [128][LDH] overwrites it
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int c = nq * 64 + nt * 16 + r;
            const float bv = a.b0[c] + (FIRST ? 0.f : a.v0[(size_t)scene * PN_H + c]);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int i = 0; i < 4; ++i) Hs[(row0 + mt * 16 + 4 * kq + i) * LDH + c] = f2bf(fmaxf(acc[mt][nt][i] + bv, 0.f));
        }
    }
    __syncthreads();
    // ---- out = fc_1(hid) + shortcut(x), one accumulator
    acc_zero(acc);
    tile_gemm_bf16<4, 4>((FIRST ? T0 : T1) + row0 * LDH, LDH, a.w1, PN_H / 32, nq * 4, PN_H / 32, acc);
    if (FIRST) {
        __syncthreads();                                   // hidden tile consumed
        gen_first(false);                                  // raw x512 for the shortcut
        __syncthreads();
    }
    tile_gemm_bf16<4, 4>(T0 + row0 * LDA, LDA, a.ws, a.kss, nq * 4, K / 32, acc);
    __syncthreads();                                       // LDS tiles dead: reuse as the fp32 epilogue tile
    constexpr int LDC = PN_H + 8;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int c = nq * 64 + nt * 16 + r;
        const float bv = a.b1[c] + (FIRST ? 0.f : a.vs[(size_t)scene * PN_H + c]);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) Cs[(row0 + mt * 16 + 4 * kq + i) * LDC + c] = acc[mt][nt][i] + bv;
    }
    __syncthreads();
    // ---- epilogue: bf16 rows to HBM (512 B per row, coalesced) and the tile's column max -> per-scene pool
    if (a.out != nullptr) {
        unsigned short* op = a.out + ((size_t)scene * a.P + p0) * PN_H;
        for (int idx = tid; idx < rows_valid * (PN_H / 4); idx += PN_THREADS) {
            const int row = idx >> 6, c4 = (idx & 63) * 4;
            const float4 v = *reinterpret_cast<const float4*>(Cs + row * LDC + c4);
            const unsigned lo = (unsigned)f2bf(v.x) | ((unsigned)f2bf(v.y) << 16), hi = (unsigned)f2bf(v.z) | ((unsigned)f2bf(v.w) << 16);
            *reinterpret_cast<uint2*>(op + (size_t)row * PN_H + c4) = make_uint2(lo, hi);
        }
    }
    {
        float m = -INFINITY;                               // thread <-> (column, row parity): two atomics per column
        for (int row = tid >> 8; row < rows_valid; row += 2) m = fmaxf(m, Cs[row * LDC + (tid & 255)]);
        // the next block consumes the pooled vector through bf16 activations of equal rounding: pool the value that
        // is actually stored
        if (a.out != nullptr) m = bf2f(f2bf(m));
        atomic_max_f32(a.pool + (size_t)scene * PN_H + (tid & 255), m);
    }
}

__global__ void k_fill(float* p, float v, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

extern "C" size_t seeme_pointnet_bf16_workspace_bytes(int B, int P) {
    const size_t M = (size_t)B * P;
    return M * PN_H * 2 * sizeof(unsigned short) + (size_t)B * PN_H * 6 * sizeof(float) + 256;
}

static int small_lin(hipStream_t st, const float* A, const float* W, int ldw, float* Y, int M, int pre_act) {
    return seeme_linear_simple(st, A, PN_H, W, ldw, nullptr, Y, PN_H, M, PN_H, PN_H, SEEME_ACT_NONE, pre_act, nullptr, nullptr);
}

extern "C" int seeme_pointnet_encode_bf16(const SeemePointnetWeights* w, const SeemePointnetBf16* wb, const float* points,
                                          int B, int P, float* out, void* workspace, size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (B <= 0 || P <= 0 || B > 65535) return seeme_fail("pointnet_bf16: bad sizes");
    if (ws_bytes < seeme_pointnet_bf16_workspace_bytes(B, P)) return seeme_fail("pointnet_bf16: workspace too small");
    const size_t M = (size_t)B * P;
    unsigned short* xa = (unsigned short*)workspace;
    unsigned short* xb = xa + M * PN_H;
    float* pools = (float*)(xb + M * PN_H);        // 4 x [B,256]
    float* v0 = pools + (size_t)4 * B * PN_H;
    float* vs = v0 + (size_t)B * PN_H;
    const size_t npool = (size_t)4 * B * PN_H;
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((npool + 255) / 256)), dim3(256), 0, st, pools, -INFINITY, npool);
    int rc = seeme_check_launch("k_fill");
    if (rc) return rc;
    const dim3 grid((P + PN_MT - 1) / PN_MT, B);
    const size_t lds_first = (size_t)PN_MT * (512 + PN_PADB) * 2;                  // 135 168 B (>= the fp32 epilogue tile)
    const size_t lds_next = (size_t)2 * PN_MT * (PN_H + PN_PADB) * 2;              // 139 264 B
    SEEME_HIP(hipFuncSetAttribute((const void*)k_pn_block<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_first));
    SEEME_HIP(hipFuncSetAttribute((const void*)k_pn_block<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_next));
    unsigned short* cur = xa;
    unsigned short* nxt = xb;
    for (int i = 0; i < 4; ++i) {
        PnBlockArgs a{};
        a.P = P;
        a.w0 = (const uint4*)wb->fc0[i]; a.ks0 = 16; a.b0 = w->fc0_b[i];
        a.w1 = (const uint4*)wb->fc1[i]; a.b1 = w->fc1_b[i];
        a.ws = (const uint4*)wb->sc[i]; a.kss = 16;
        a.pool = pools + (size_t)i * B * PN_H;
        a.out = (i < 3) ? nxt : nullptr;            // the last block only feeds the final pool
        if (i == 0) {
            a.points = points; a.pos_w = w->pos_w; a.pos_b = w->pos_b;
            hipLaunchKernelGGL((k_pn_block<true>), grid, dim3(PN_THREADS), lds_first, st, a);
        } else {
            const float* pool_prev = pools + (size_t)(i - 1) * B * PN_H;
            // pooled halves in fp32: v0 = W0[:,256:] relu(pool), vs = Ws[:,256:] pool
            if ((rc = small_lin(st, pool_prev, w->fc0_w[i] + PN_H, 512, v0, B, SEEME_ACT_RELU))) return rc;
            if ((rc = small_lin(st, pool_prev, w->sc_w[i] + PN_H, 512, vs, B, SEEME_ACT_NONE))) return rc;
            a.x = cur; a.v0 = v0; a.vs = vs;
            hipLaunchKernelGGL((k_pn_block<false>), grid, dim3(PN_THREADS), lds_next, st, a);
        }
        if ((rc = seeme_check_launch("k_pn_block"))) return rc;
        if (i > 0) { unsigned short* t = cur; cur = nxt; nxt = t; } else { cur = nxt; nxt = xa; }
    }
    // fc_c(relu(pool of block_3))
    return seeme_linear_simple(st, pools + (size_t)3 * B * PN_H, PN_H, w->fcc_w, PN_H, w->fcc_b, out, w->out_dim, B, w->out_dim,
                               PN_H, SEEME_ACT_NONE, SEEME_ACT_RELU, nullptr, nullptr);
}
