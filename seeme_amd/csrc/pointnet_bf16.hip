// pointnet_bf16.hip -- ResNet-PointNet scene encoder (EgoHMR/models/respointnet.py:33-97) as fused
// bf16-MFMA kernels: the one part of the path with GEMMs big enough for the matrix cores
// (M = B x 20 000 points).
//
// A persistent 4-wave workgroup (two per CU) walks 64-point tiles of the scenes and runs a whole ResnetBlockFC on each:
//     hid = relu( W0[:, :256] relu(x) + W0[:, 256:] relu(pool) + b0 )          fc_0
//     out =       Ws[:, :256] x       + Ws[:, 256:] pool
//               + W1 hid + b1                                                   shortcut + fc_1, ONE accumulator
// The point features stay in LDS as bf16 (raw and relu'd copies), the hidden tile overwrites the relu'd copy,
// weights stream from L2 as fragment-packed bf16 through a register ring that runs across GEMMs and tiles,
// accumulation is fp32 (v_mfma_f32_16x16x32_bf16, weight fragment as the A operand so that a lane owns 16 consecutive
// features of one point).  Hidden tile, block output and the per-scene max-pool come straight from the accumulators;
// the pool reaches memory through one atomic per feature per scene change; activations travel between blocks as bf16
// (half the HBM bytes).  The pooled halves are per-scene fp32 vectors from k_pn_rows.  block_0
// generates its input fc_pos_0(points) on the matrix cores (split-bf16 operands) and evaluates its shortcut, folded
// through fc_pos_0 to a 3 -> 256 map, in the epilogue.  DESIGN.md section 5.2 has the measurements behind each choice.
#include "common.hpp"
#include "api_util.hpp"
#include "pointnet_v2.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#ifndef PN_MH_FIRST
#define PN_MH_FIRST 1      // first block: 64-point row groups per workgroup (see PN_MH_NEXT)
#endif
#ifndef PN_MH_NEXT
#define PN_MH_NEXT 1       // later blocks: 64-point row groups per workgroup (1: 4 waves, two workgroups per CU; 2: 8 waves, one)
#endif
#define PN_H 256           // hidden width
#define PN_PADB 16         // bf16 elements of row padding (2 x 16-byte slots: conflict-free ds_read_b128; measured: 8 is worse)

typedef unsigned int pn_u32x4 __attribute__((ext_vector_type(4)));
typedef short pn_s16x4 __attribute__((ext_vector_type(4)));
#ifndef PN_PF
#define PN_PF 4                                            // weight fragments in flight: PN_PF k-blocks x 4 n-tiles per wave
#endif

// One packed matrix as a buffer resource + this wave's n-tile offset (everything scalar; the lane part is one VGPR).
template <int KS>
struct PnMat {
    __amdgpu_buffer_rsrc_t rs; unsigned sbase;
    __device__ __forceinline__ PnMat(const uint4* Wp, int ntile0)
        : rs(__builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(Wp), 0, 16 * KS * 1024, 0x00020000)), sbase((unsigned)ntile0 * (KS * 1024u)) {}
    __device__ __forceinline__ pn_u32x4 frag(int nt, int kb) const {
        return __builtin_amdgcn_raw_buffer_load_b128(rs, (threadIdx.x & 63u) * 16u, sbase + (unsigned)(nt * KS + kb) * 1024u, 0);
    }
};

template <int NTL, int KS>
__device__ __forceinline__ void ring_prime(pn_u32x4 (&br)[PN_PF][NTL], const PnMat<KS>& m) {
#pragma unroll
    for (int u = 0; u < PN_PF; ++u)
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) br[u][nt] = m.frag(nt, u);
}

// The weight ring br is ONE stream over all GEMMs of all tiles: on entry it holds k-blocks 0..PF-1 of `cur`; the slot
// of k-block kb is re-filled with k-block kb+PF -- of `cur`, or past its end of `nxt` (the next GEMM, possibly of the
// next tile), so that the vector-memory path, which bounds these GEMMs (8 waves x 4 KiB per k-block at 64 B/clk),
// also works through the barriers, the hidden-tile write and the epilogue between GEMMs.
template <int MTL, int NTL, int KS, int KT, int KB0, int KN, int KSN>
__device__ __forceinline__ void tile_gemm_bf16(const unsigned short* __restrict__ As, int lda, const PnMat<KS>& cur,
                                               const PnMat<KSN>& nxt, bool has_next, pn_u32x4 (&br)[PN_PF][NTL], f32x4 (&acc)[MTL][NTL]) {
    // k-blocks [KB0, KB0 + KN) of a matrix with KT k-blocks; As holds the KN * 32 columns of this window
    const int lane = threadIdx.x & 63, r = lane & 15, kq = lane >> 4;
    const unsigned short* ap = As + r * lda + 8 * kq;
    static_assert(KT % PN_PF == 0 && KB0 % PN_PF == 0 && KN % PN_PF == 0 && PN_PF % 2 == 0, "ring slots must line up across GEMMs");
    // A fragments of the next k-block are read from LDS while the MFMAs of the current one issue
    uint4 ab[2][MTL];
#pragma unroll
    for (int mt = 0; mt < MTL; ++mt) ab[0][mt] = *reinterpret_cast<const uint4*>(ap + mt * 16 * lda);
#pragma unroll
    for (int kk = 0; kk < KN; ++kk) {
        const int u = kk % PN_PF, kb = KB0 + kk;
        if (kk + 1 < KN) {
#pragma unroll
            for (int mt = 0; mt < MTL; ++mt) ab[(kk + 1) & 1][mt] = *reinterpret_cast<const uint4*>(ap + mt * 16 * lda + (kk + 1) * 32);
        }
        // the scheduler is fenced per k-block: left alone it sinks the re-fills to just before their use
        // (vmcnt(0) inside every k-block) and the pipeline collapses
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < MTL; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, br[u][nt]), __builtin_bit_cast(bf16x8, ab[kk & 1][mt]), acc[mt][nt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (kb + PN_PF < KT) {
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt) br[u][nt] = cur.frag(nt, kb + PN_PF);
        } else if (has_next) {
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt) br[u][nt] = nxt.frag(nt, kb + PN_PF - KT);
        }
    }
}

// float atomic max through the ordered-integer trick (destination initialised to -inf)
__device__ __forceinline__ void atomic_max_f32(float* p, float v) {
    if (v >= 0.f) atomicMax(reinterpret_cast<int*>(p), __float_as_int(v));
    else atomicMin(reinterpret_cast<unsigned int*>(p), __float_as_uint(v));
}

#ifdef PN_DBG_TIMES
// debug build only: cycle stamps of one workgroup at the phase boundaries of its second tile (scripts/pn_times.py)
__device__ unsigned long long pn_dbg_times[32];
#ifndef PN_DBG_FIRST
#define PN_DBG_FIRST 0
#endif
#define PN_DBG(i) do { if (FIRST == (PN_DBG_FIRST != 0) && a.out != nullptr && blockIdx.x == 40 && t == t0 + 3 && threadIdx.x == 0) pn_dbg_times[i] = __builtin_readcyclecounter(); } while (0)
#else
#define PN_DBG(i) do {} while (0)
#endif

struct PnBlockArgs {
    // input: FIRST block -> points + fc_pos_0; later blocks -> x bf16 [B*P,256]
    const float* points; const uint2* posf;   // FIRST: points [B,P,3]; fc_pos_0 as split-bf16 MFMA fragments [32][64]
    const float* sc3;                       // FIRST: shortcut folded through fc_pos_0, [256][4] = (Ws Wp | Ws bp) fp32
    const unsigned short* x;
    const uint4* w0;                        // fc_0, fragment-packed [16 n-tiles][ks0 k-blocks][64 lanes] (ks0 = 16: K = 512 packed; later blocks use k-blocks 0..7)
    const float* b0;
    const float* v0;                        // [B,256] pooled half of fc_0 (NULL in the first block)
    const uint4* w1;                        // fc_1, fragment-packed, 8 k-blocks
    const float* b1;
    const uint4* ws;                        // shortcut, fragment-packed like fc_0 (later blocks)
    const float* vs;                        // [B,256] pooled half of the shortcut (NULL in the first block)
    unsigned short* out;                    // [B*P,256] bf16 block output (may be NULL for the last block)
    float* pool;                            // [B,256] running max of the block output (pre-initialised to -inf)
    int P, tiles_x, n_tiles;                // tiles per scene, tiles in all
};

// fmaxf() canonicalises both operands first (3 instructions); the values here are never signalling NaNs
// (as v_med3_f32 with +inf: one instruction the compiler schedules and hazard-checks itself)
__device__ __forceinline__ float pn_max(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, __builtin_inff()); }
typedef float pn_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 pn_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {   // one v_cvt_pk_bf16_f32
    return __builtin_bit_cast(unsigned, __builtin_convertvector(pn_f32x2{lo, hi}, pn_bf16x2));
}

// One PERSISTENT workgroup per CU walks the 128-point tiles (scene-major).  The matrix-core calls take the weight
// fragment as the A operand and the point fragment as B, so a lane ends up with 4 consecutive weight rows of ONE
// point; the host packs the weight rows so that those are 16 consecutive output features over the wave's four
// n-tiles (feature = 64 nq + 16 kq + 4 nt + i).  Hidden tile, block output and column max are therefore produced
// straight from the accumulators: 32-byte runs per lane, no fp32 staging tile, no pass over LDS for the pool.
// The next tile's rows (8 x 16 B per thread) are requested after the last GEMM of the current tile, so their HBM
// latency runs under the epilogue (vmcnt is in-order: requested earlier they would stall the first weight wait).
template <bool FIRST, int MH>
__global__ __launch_bounds__(256 * MH, 2) void k_pn_block(const PnBlockArgs a) {
    constexpr int MT = 64 * MH, NTHR = 256 * MH;          // points per tile, threads (4 MH waves)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // 8 waves = 2 row halves x 4 column quarters: two waves per SIMD, so one wave's MFMAs cover the other's LDS / L2 latency
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, r = lane & 15, kq = lane >> 4;
    const int mh = wave >> 2, nq = wave & 3, row0 = mh * 64;
    const int fbase = nq * 64 + kq * 16;                  // this lane's 16 consecutive output features
    constexpr int LDA = PN_H + PN_PADB;                   // bf16 elements
    constexpr int LDH = PN_H + PN_PADB;
    // FIRST: one [MT][272] tile holding one 256-column half of relu(x512) at a time (fc_0 runs as two K = 256
    // halves); the hidden tile aliases it.  later: raw tile | relu tile; the hidden tile overwrites the relu tile.
    unsigned short* T0 = reinterpret_cast<unsigned short*>(smem_raw);
    unsigned short* T1 = FIRST ? T0 : T0 + MT * LDA;
    __shared__ __attribute__((aligned(16))) float spts[MT * 3];
    __shared__ __attribute__((aligned(16))) float sb0[PN_H], sb1[PN_H];     // b0 (+ pooled half), b1 (+ pooled half / folded bias)
    __shared__ __attribute__((aligned(16))) float ssc[FIRST ? PN_H * 4 : 4];

    // FIRST: x512 = relu(fc_pos_0(p)) as bf16 into T0, on the matrix cores (as plain FMAs the 128 x 512 x K=3 map
    // cost more vector-ALU time than a K = 512 GEMM costs MFMA time).  fp32 accuracy from 16-bit operands by
    // splitting both sides, v = hi + lo: one v_mfma_f32_16x16x16_bf16 per 16 x 16 tile with the k slots
    //   k 0..2: p_hi w_hi   k 3..5: p_lo w_hi   k 6..8: p_hi w_lo   k 9: 1 b_hi   k 10: 1 b_lo
    // (p_lo w_lo ~ 2^-18 relative is dropped).  The weight side arrives as ready fragments (SeemePointnetBf16.posf).
    __shared__ __attribute__((aligned(16))) uint2 sposf[FIRST ? 32 * 64 : 1];
    if (FIRST) {
        for (int c = tid; c < 32 * 64; c += NTHR) sposf[c] = a.posf[c];
        for (int i = tid; i < PN_H * 4; i += NTHR) ssc[i] = a.sc3[i];
        if (tid < PN_H) { sb0[tid] = a.b0[tid]; sb1[tid] = a.b1[tid] + a.sc3[tid * 4 + 3]; }
    }

    // ---- prefetch registers of the next tile
    constexpr int NIT = FIRST ? 1 : MT * (PN_H / 8) / NTHR;   // 8 row segments of 16 B per thread
    pn_u32x4 pf[NIT];
    float pf_pt = 0.f;
    auto issue = [&](int tn) {
        const int sc = tn / a.tiles_x, q0 = (tn - sc * a.tiles_x) * MT, rv = min(MT, a.P - q0);
        if (FIRST) {
            // (buffer loads throughout: scalar bases, 32-bit lane offsets, rows past the end read as zeros -- per-thread
            // 64-bit addresses are loop invariants the compiler would keep, and spill, across the whole tile loop)
            const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(a.points + ((size_t)sc * a.P + q0) * 3), 0, rv * 12, 0x00020000);
            pf_pt = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rp, (unsigned)tid * 4u, 0, 0));
        } else {
            // the tile's rows are contiguous: one buffer over its valid bytes, rows past the end read as zeros;
            // all 8 loads of a thread are in flight together (the tile is one HBM round trip)
            const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<unsigned short*>(a.x + ((size_t)sc * a.P + q0) * PN_H), 0, rv * PN_H * 2, 0x00020000);
#pragma unroll
            for (int it = 0; it < NIT; ++it)
                pf[it] = __builtin_amdgcn_raw_buffer_load_b128(rt, (unsigned)tid * 16u, (unsigned)(it * NTHR * 16), 0);
        }
    };

    // FIRST: half h (columns [256 h, +256)) of relu(fc_pos_0(points)) into T0; wave (mh, nq) covers rows [64 mh, +64)
    // x columns [64 nq, +64) of the half.  One base address per operand, the (n4, mt) part is an instruction immediate.
    pn_s16x4 pfr[FIRST ? 4 : 1];                           // point fragments (B operand) of this lane's 4 row tiles
    auto gen_half = [&](int h) {
        const uint2* wfp = sposf + h * 1024 + nq * 256 + lane;
        unsigned short* xp = T0 + (row0 + r) * LDA + nq * 64 + 4 * kq;
#pragma unroll
        for (int n4 = 0; n4 < 4; ++n4) {
            const pn_s16x4 wf = __builtin_bit_cast(pn_s16x4, wfp[n4 * 64]);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const f32x4 c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wf, pfr[mt], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                // lane (r, kq) holds columns 64 nq + 16 n4 + 4 kq .. +3 (of the half) of point row0 + 16 mt + r
                *reinterpret_cast<uint2*>(xp + mt * 16 * LDA + n4 * 16) =
                    make_uint2(pack_bf16x2(pn_max(c[0], 0.f), pn_max(c[1], 0.f)), pack_bf16x2(pn_max(c[2], 0.f), pn_max(c[3], 0.f)));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // a workgroup owns a CONTIGUOUS range of tiles: mostly one scene, so the column max is kept in a register of
    // thread <-> feature across tiles and reaches the per-scene pool through one atomic per feature per scene change
    // (an atomic per tile and feature made every CU queue behind the same 256 addresses)
    __shared__ __attribute__((aligned(16))) float smax[MH][PN_H];
    const int t0 = (int)(((long long)blockIdx.x * a.n_tiles) / gridDim.x), t1 = (int)(((long long)(blockIdx.x + 1) * a.n_tiles) / gridDim.x);
    float run_max = -INFINITY;
    int prev_scene = -1, bias_scene = -1;
    // the weight stream: fc_0 | fc_1 | shortcut per tile, PN_PF k-blocks ahead across GEMM and tile boundaries
    const PnMat<16> m0(a.w0, nq * 4), ms(FIRST ? a.w0 : a.ws, nq * 4);
    const PnMat<PN_H / 32> m1(a.w1, nq * 4);
    pn_u32x4 br[PN_PF][4];
    if (t0 < t1) { ring_prime(br, m0); issue(t0); }
    for (int t = t0; t < t1; ++t) {
        const int scene = t / a.tiles_x, p0 = (t - scene * a.tiles_x) * MT;
        const int rows_valid = min(MT, a.P - p0);
        PN_DBG(0);
#ifdef PN_DBG_TIMES
        if (FIRST == (PN_DBG_FIRST != 0) && a.out != nullptr && blockIdx.x == 40 && threadIdx.x == 0 && (t - t0) % 10 == 3 && (t - t0) / 10 < 4)
            pn_dbg_times[16 + (t - t0) / 10] = __builtin_readcyclecounter();     // tiles 3, 13, 23, 33: cycles per tile in the steady state
#endif
        // ---- land the prefetched tile in LDS
        if (FIRST) {
            if (tid < MT * 3) spts[tid] = pf_pt;
            __syncthreads();
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int row = row0 + mt * 16 + r;
                const float px = spts[row * 3 + 0], py = spts[row * 3 + 1], pz = spts[row * 3 + 2];
                const unsigned hxy = pack_bf16x2(px, py), hz = pack_bf16x2(pz, 0.f) & 0xFFFFu;
                const unsigned lxy = pack_bf16x2(px - __uint_as_float(hxy << 16), py - __uint_as_float(hxy & 0xFFFF0000u));
                const unsigned lz = pack_bf16x2(pz - __uint_as_float(hz << 16), 0.f) & 0xFFFFu;
                const unsigned one = 0x3F80u;
                // kq 0: [phx phy | phz plx]   kq 1: [ply plz | phx phy]   kq 2: [phz 1 | 1 0]   kq 3: zeros
                const unsigned d0 = kq == 0 ? hxy : kq == 1 ? ((lxy >> 16) | (lz << 16)) : kq == 2 ? (hz | (one << 16)) : 0u;
                const unsigned d1 = kq == 0 ? (hz | (lxy << 16)) : kq == 1 ? hxy : kq == 2 ? one : 0u;
                pfr[mt] = __builtin_bit_cast(pn_s16x4, make_uint2(d0, d1));
            }
            gen_half(0);
        } else {
            auto relu2 = [](unsigned u) { const unsigned m = (u >> 15) & 0x00010001u; return u & ~(m * 0xFFFFu); };   // packed bf16 relu
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int idx = tid + it * NTHR, row = idx >> 5, c8 = (idx & 31) * 8;
                *reinterpret_cast<uint4*>(T0 + row * LDA + c8) = make_uint4(pf[it].x, pf[it].y, pf[it].z, pf[it].w);
                *reinterpret_cast<uint4*>(T1 + row * LDA + c8) = make_uint4(relu2(pf[it].x), relu2(pf[it].y), relu2(pf[it].z), relu2(pf[it].w));
            }
            // biases + pooled halves of this tile's scene: re-read only when the scene changes (tiles are scene-major;
            // as prefetch registers they were spilled right after their loads -- a vmcnt(0) under the tile prefetch)
            if (scene != bias_scene) {
                if (wave < 4) {
                    int ln;
                    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
                    const int c = wave * 64 + ln;
                    sb0[c] = a.b0[c] + a.v0[(size_t)scene * PN_H + c];
                    sb1[c] = a.b1[c] + a.vs[(size_t)scene * PN_H + c];
                }
                bias_scene = scene;
            }
        }
        __syncthreads();
        PN_DBG(1);

        // ---- fc_0 on relu(x): wave (mh, nq) owns rows [64 mh, +64) x hidden features [64 nq, +64)
        f32x4 acc[4][4];
        acc_zero(acc);
        if (FIRST) {
            tile_gemm_bf16<4, 4, 16, 16, 0, 8, PN_H / 32>(T0 + row0 * LDA, LDA, m0, m1, true, br, acc);
            __syncthreads();                                   // first half consumed
            gen_half(1);
            __syncthreads();
            tile_gemm_bf16<4, 4, 16, 16, 8, 8, PN_H / 32>(T0 + row0 * LDA, LDA, m0, m1, true, br, acc);
        } else {
            tile_gemm_bf16<4, 4, 16, 8, 0, 8, PN_H / 32>(T1 + row0 * LDA, LDA, m0, m1, true, br, acc);
        }
        PN_DBG(2);
        // fold the previous tile's column max into the running one (kept away from the loop top: the rare atomic
        // path would make the wait for the prefetched tile a full vmcnt(0) drain, stores included)
        if (wave < 4 && prev_scene >= 0) {                      // thread <-> feature 64 wave + lane
            // the lane id is formed HERE (volatile): kept across the tile loop it gets spilled, and its reload is a
            // vmcnt(0) drain of the weight ring in every tile
            int ln;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
            const int tt = wave * 64 + ln;
            run_max = pn_max(run_max, pn_max(smax[0][tt], smax[MH - 1][tt]));
            if (scene != prev_scene) {
                atomic_max_f32(a.pool + (size_t)prev_scene * PN_H + tt, run_max);
                run_max = -INFINITY;
            }
        }
        prev_scene = scene;
        __syncthreads();                                       // all waves done reading the relu tile
        PN_DBG(3);
        {
            unsigned short* Hs = FIRST ? T0 : T1;              // hidden tile [128][LDH] overwrites it
            float bv[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 b4 = *reinterpret_cast<const float4*>(sb0 + fbase + 4 * q);
                bv[4 * q] = b4.x; bv[4 * q + 1] = b4.y; bv[4 * q + 2] = b4.z; bv[4 * q + 3] = b4.w;
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                unsigned w[8];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    w[2 * nt] = pack_bf16x2(pn_max(acc[mt][nt][0] + bv[4 * nt], 0.f), pn_max(acc[mt][nt][1] + bv[4 * nt + 1], 0.f));
                    w[2 * nt + 1] = pack_bf16x2(pn_max(acc[mt][nt][2] + bv[4 * nt + 2], 0.f), pn_max(acc[mt][nt][3] + bv[4 * nt + 3], 0.f));
                }
                unsigned short* hp = Hs + (row0 + mt * 16 + r) * LDH + fbase;
                *reinterpret_cast<uint4*>(hp) = make_uint4(w[0], w[1], w[2], w[3]);
                *reinterpret_cast<uint4*>(hp + 8) = make_uint4(w[4], w[5], w[6], w[7]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        PN_DBG(4);
        // ---- out = fc_1(hid) + shortcut(x), one accumulator (FIRST: the shortcut is the folded K = 3 map of the epilogue)
        acc_zero(acc);
        if (FIRST) tile_gemm_bf16<4, 4, PN_H / 32, 8, 0, 8, 16>(T0 + row0 * LDH, LDH, m1, m0, t + 1 < t1, br, acc);
        else tile_gemm_bf16<4, 4, PN_H / 32, 8, 0, 8, 16>(T1 + row0 * LDH, LDH, m1, ms, true, br, acc);
        PN_DBG(5);
        if (!FIRST) tile_gemm_bf16<4, 4, 16, 8, 0, 8, 16>(T0 + row0 * LDA, LDA, ms, m0, t + 1 < t1, br, acc);
        PN_DBG(6);
        // ---- request the next tile, then finish this one from the accumulators
        if (t + 1 < t1) issue(t + 1);
        PN_DBG(7);
        {
            float bv[16], mx[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 b4 = *reinterpret_cast<const float4*>(sb1 + fbase + 4 * q);
                bv[4 * q] = b4.x; bv[4 * q + 1] = b4.y; bv[4 * q + 2] = b4.z; bv[4 * q + 3] = b4.w;
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) mx[j] = -INFINITY;
            float4 s4[FIRST ? 16 : 1];                        // FIRST: this lane's rows of the folded shortcut
            if (FIRST) {
#pragma unroll
                for (int j = 0; j < 16; ++j) s4[j] = *reinterpret_cast<const float4*>(ssc + (fbase + j) * 4);
            }
            const bool has_out = a.out != nullptr;
            const unsigned vo0 = (unsigned)((row0 + r) * PN_H + fbase) * 2u;
            const int row0r = row0 + r;
            // rows past the tile's valid bytes are dropped by the buffer bounds check
            const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
                has_out ? a.out + ((size_t)scene * a.P + p0) * PN_H : nullptr, 0, has_out ? rows_valid * PN_H * 2 : 0, 0x00020000);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int row = row0 + mt * 16 + r;
                int rvm = rows_valid - mt * 16;                // one lane value (row0 + r) for all row tiles (see vo0):
                asm volatile("" : "+s"(rvm));                  // the scalar side carries mt, kept from being re-associated
                const bool valid = row0r < rvm;
                float v[16];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[4 * nt + i] = acc[mt][nt][i] + bv[4 * nt + i];
                if (FIRST) {
                    const float px = spts[row * 3 + 0], py = spts[row * 3 + 1], pz = spts[row * 3 + 2];
#pragma unroll
                    for (int j = 0; j < 16; ++j) v[j] += s4[j].x * px + s4[j].y * py + s4[j].z * pz;
                }
                if (has_out) {
                    unsigned w[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) w[q] = pack_bf16x2(v[2 * q], v[2 * q + 1]);
                    // one lane offset for all row tiles; the row-tile part travels as the scalar offset (per-mt lane
                    // offsets are loop invariants the compiler keeps -- and spills -- across the tile loop)
                    __builtin_amdgcn_raw_buffer_store_b128(pn_u32x4{w[0], w[1], w[2], w[3]}, ro, vo0, (unsigned)(mt * 16 * PN_H * 2), 0);
                    __builtin_amdgcn_raw_buffer_store_b128(pn_u32x4{w[4], w[5], w[6], w[7]}, ro, vo0 + 16u, (unsigned)(mt * 16 * PN_H * 2), 0);
                }
                if (valid) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) mx[j] = pn_max(mx[j], v[j]);
                }
                __builtin_amdgcn_sched_barrier(0);             // one row tile at a time: the ring and the prefetched tile stay in registers
            }
            PN_DBG(10);
            // column max over the wave's 64 points: the 16 lanes of a DPP row hold 16 points of the same 16 features.
            // Reduce-scatter (15 exchanges instead of 16 x 4): each step a lane keeps half of its values and takes the
            // partner's copy of that half; lane r ends with feature r of the group.  Partners: r^8 (row_ror:8),
            // 7-r within the half row (row_half_mirror), r^2, r^1 (quad_perm).
            auto xch = [](float send, int ctrl_sel) {
                const int v = __float_as_int(send);
                return __int_as_float(ctrl_sel == 0 ? __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, true)
                                    : ctrl_sel == 1 ? __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true)
                                    : ctrl_sel == 2 ? __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true)
                                                    : __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true));
            };
            float a8[8], a4[4], a2[2], a1;
            const bool b3 = r & 8, b2 = r & 4, b1 = r & 2, b0 = r & 1;
#pragma unroll
            for (int j = 0; j < 8; ++j) a8[j] = pn_max(b3 ? mx[j + 8] : mx[j], xch(b3 ? mx[j] : mx[j + 8], 0));
#pragma unroll
            for (int j = 0; j < 4; ++j) a4[j] = pn_max(b2 ? a8[j + 4] : a8[j], xch(b2 ? a8[j] : a8[j + 4], 1));
#pragma unroll
            for (int j = 0; j < 2; ++j) a2[j] = pn_max(b1 ? a4[j + 2] : a4[j], xch(b1 ? a4[j] : a4[j + 2], 2));
            a1 = pn_max(b0 ? a2[1] : a2[0], xch(b0 ? a2[0] : a2[1], 3));
            // the next block consumes the pooled vector through bf16 activations of equal rounding: pool the value that
            // is actually stored (rounding is monotone, so the max is rounded once)
            if (has_out) a1 = __uint_as_float(pack_bf16x2(a1, 0.f) << 16);
            {
                int ln;                                        // formed here: see the running-max fold above
                asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
                smax[mh][nq * 64 + ln] = a1;                   // feature 64 nq + 16 kq + r
            }
        }
        PN_DBG(8);
        __syncthreads();                                       // tiles, biases and points of this tile are dead
        PN_DBG(9);
    }
    if (tid < PN_H && prev_scene >= 0) {
        run_max = pn_max(run_max, pn_max(smax[0][tid], smax[MH - 1][tid]));
        atomic_max_f32(a.pool + (size_t)prev_scene * PN_H + tid, run_max);
    }
}

#ifdef PN_DBG_TIMES
extern "C" int seeme_debug_pn_times(unsigned long long* host, int n) {
    SEEME_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(pn_dbg_times), sizeof(unsigned long long) * (size_t)(n < 32 ? n : 32)));
    return 0;
}
#endif


__global__ void k_fill(float* p, float v, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

extern "C" size_t seeme_pointnet_bf16_workspace_bytes(int B, int P) {
    const size_t M = (size_t)B * ((P + 15) / 16 * 16);        // rows per scene padded to the 16-point tiles of the fragment-order layout
    return M * PN_H * 2 * sizeof(unsigned short) + (size_t)B * PN_H * 6 * sizeof(float) + 256;
}

// The small fp32 maps of pooled vectors that sit on the dependency chain between block kernels, two 256-output maps
// per launch: the pooled halves of the next block (v0 = W0[:, 256:] relu(pool), vs = Ws[:, 256:] pool) and the final
// fc_c(relu(pool)) as its two row halves.  grid (scene, 4): a workgroup owns 128 of the 512 outputs, 4 lanes per
// output (a wave instruction reads 16 rows x 64 contiguous bytes), pooled vector from LDS.
struct PnRowsArgs {
    const float* pool;                      // [B,256]
    const float* w[2]; const float* bias[2];   // row-major, row stride ldw floats, columns [col0, col0 + 256); bias may be NULL
    float* y[2]; int ldy;                   // y[m][b * ldy + f]
    int ldw, col0, relu[2];
};
__global__ __launch_bounds__(512) void k_pn_rows(const PnRowsArgs a) {
    __shared__ __attribute__((aligned(16))) float sp[PN_H];
    const int tid = threadIdx.x, b = blockIdx.x, which = blockIdx.y >> 1, sub = tid & 3;
    const int f = (blockIdx.y & 1) * 128 + (tid >> 2);
    if (tid < PN_H) {
        const float p = a.pool[(size_t)b * PN_H + tid];
        sp[tid] = a.relu[which] ? fmaxf(p, 0.f) : p;
    }
    __syncthreads();
    const float4* row = reinterpret_cast<const float4*>(a.w[which] + (size_t)f * a.ldw + a.col0);
    const float4* p4 = reinterpret_cast<const float4*>(sp);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < PN_H / 16; ++it) {
        const float4 w = row[it * 4 + sub], p = p4[it * 4 + sub];
        acc[0] += w.x * p.x; acc[1] += w.y * p.y; acc[2] += w.z * p.z; acc[3] += w.w * p.w;
    }
    float v = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));   // lane ^ 1
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));   // lane ^ 2
    if (sub == 0) a.y[which][(size_t)b * a.ldy + f] = v + (a.bias[which] != nullptr ? a.bias[which][f] : 0.f);
}

#ifndef PN_CHUNK_MB_DEFAULT
#define PN_CHUNK_MB_DEFAULT 0
#endif
extern "C" int seeme_pointnet_encode_bf16(const SeemePointnetWeights* w, const SeemePointnetBf16* wb, const float* points,
                                          int B, int P, float* out, void* workspace, size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (B <= 0 || P <= 0 || B > 65535) return seeme_fail("pointnet_bf16: bad sizes");
    if (ws_bytes < seeme_pointnet_bf16_workspace_bytes(B, P)) return seeme_fail("pointnet_bf16: workspace too small");
    {   // Scene chunks: the activations between two blocks (2 x P x 512 B per scene) live in the SAME workspace region for every
        // chunk, sized to stay in the 256 MB Infinity Cache -- a whole batch of 64 x 20 000 points writes 655 MB per block and
        // reads it back from HBM.  SEEME_PN_CHUNK_MB: bytes of both activation buffers per chunk (0 = no chunking).
        static long chunk_mb = -1;
        if (chunk_mb < 0) { const char* e = getenv("SEEME_PN_CHUNK_MB"); chunk_mb = e ? atol(e) : PN_CHUNK_MB_DEFAULT; }
        const size_t per_scene = (size_t)((P + 15) / 16 * 16) * PN_H * 2 * 2;
        int chunk = chunk_mb > 0 ? (int)(((size_t)chunk_mb << 20) / per_scene) : B;
        if (chunk < 1) chunk = 1;
        if (chunk < B) {
            for (int c0 = 0; c0 < B; c0 += chunk) {
                const int nb = B - c0 < chunk ? B - c0 : chunk;
                const int rc = seeme_pointnet_encode_bf16(w, wb, points + (size_t)c0 * P * 3, nb, P, out + (size_t)c0 * w->out_dim, workspace,
                                                          seeme_pointnet_bf16_workspace_bytes(nb, P), stream);
                if (rc) return rc;
            }
            return 0;
        }
    }
    const size_t M = (size_t)B * ((P + 15) / 16 * 16);
    unsigned short* xa = (unsigned short*)workspace;
    unsigned short* xb = xa + M * PN_H;
    float* pools = (float*)(xb + M * PN_H);        // 4 x [B,256]
    float* v0 = pools + (size_t)4 * B * PN_H;
    float* vs = v0 + (size_t)B * PN_H;
    const size_t npool = (size_t)4 * B * PN_H;
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((npool + 255) / 256)), dim3(256), 0, st, pools, -INFINITY, npool);
    int rc = seeme_check_launch("k_fill");
    if (rc) return rc;
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        SEEME_HIP(hipGetDevice(&dev));
        SEEME_HIP(hipGetDeviceProperties(&prop, dev));
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    // persistent grids sized by LDS occupancy: two independent 4-wave workgroups (64-point tiles) per CU, whose
    // phases interleave (one 8-wave workgroup per CU ran its landing / hidden / epilogue phases with idle matrix cores)
    constexpr int MH_FIRST = PN_MH_FIRST, MH_NEXT = PN_MH_NEXT;
    if (((long long)(P + 63) / 64) * B > 0x7fffffffLL) return seeme_fail("pointnet_bf16: too many tiles");
    const size_t lds_first = (size_t)64 * MH_FIRST * (PN_H + PN_PADB) * 2;         // 34 816 B per 64-point tile (+ 26 KB static)
    const size_t lds_next = (size_t)2 * 64 * MH_NEXT * (PN_H + PN_PADB) * 2;       // 69 632 B per 64-point tile
    SEEME_HIP(hipFuncSetAttribute((const void*)k_pn_block<true, MH_FIRST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_first));
    SEEME_HIP(hipFuncSetAttribute((const void*)k_pn_block<false, MH_NEXT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_next));
    unsigned short* cur = xa;
    unsigned short* nxt = xb;
    // SEEME_PN_V2: unset / "1" = second-generation block kernels (pointnet_v2.hip), "0" = first generation.  The two keep
    // the activations between blocks in different layouts (fragment order / row-major), so it is all blocks or none.
    static int v2_mask = -1;
    if (v2_mask < 0) { const char* e = getenv("SEEME_PN_V2"); v2_mask = (e == nullptr || e[0] != '0') ? 0xF : 0; }
    const bool v2_ok = wb->stream[0] != nullptr && wb->sc3f != nullptr;
    for (int i = 0; i < 4; ++i) {
        const bool v2 = v2_ok && ((v2_mask >> i) & 1);
        if (i > 0) {
            const float* pool_prev = pools + (size_t)(i - 1) * B * PN_H;
            // pooled halves in fp32: v0 = W0[:,256:] relu(pool), vs = Ws[:,256:] pool
            PnRowsArgs ra{};
            ra.pool = pool_prev; ra.w[0] = w->fc0_w[i]; ra.w[1] = w->sc_w[i]; ra.y[0] = v0; ra.y[1] = vs; ra.ldy = PN_H;
            ra.ldw = 512; ra.col0 = PN_H; ra.relu[0] = 1; ra.relu[1] = 0;
            hipLaunchKernelGGL(k_pn_rows, dim3((unsigned)B, 4), dim3(512), 0, st, ra);
            if ((rc = seeme_check_launch("k_pn_rows"))) return rc;
        }
        if (v2) {
            PnBlock2Args a{};
            const int tp = seeme_pn_block2_tile_points();
            a.P = P; a.Ppad = (P + 15) / 16 * 16; a.tiles_x = (P + tp - 1) / tp; a.n_tiles = a.tiles_x * B;
            a.stream = (const uint4*)wb->stream[i]; a.b0 = w->fc0_b[i]; a.b1 = w->fc1_b[i];
            a.pool = pools + (size_t)i * B * PN_H;
            a.out = (i < 3) ? nxt : nullptr;
            if (i == 0) { a.points = points; a.posf = (const uint2*)wb->posf; a.sc3f = (const uint2*)wb->sc3f; }
            else { a.x = cur; a.v0 = v0; a.vs = vs; }
            if ((rc = seeme_pn_block2_launch(i == 0, a, n_cu, st))) return rc;
        } else {
            PnBlockArgs a{};
            const int mt = 64 * (i == 0 ? MH_FIRST : MH_NEXT), per_cu = 2 / (i == 0 ? MH_FIRST : MH_NEXT);
            a.P = P; a.tiles_x = (P + mt - 1) / mt; a.n_tiles = a.tiles_x * B;
            const dim3 grid((unsigned)(a.n_tiles < n_cu * per_cu ? a.n_tiles : n_cu * per_cu));
            a.w0 = (const uint4*)wb->fc0[i]; a.b0 = w->fc0_b[i];
            a.w1 = (const uint4*)wb->fc1[i]; a.b1 = w->fc1_b[i];
            a.ws = (const uint4*)wb->sc[i];
            a.pool = pools + (size_t)i * B * PN_H;
            a.out = (i < 3) ? nxt : nullptr;            // the last block only feeds the final pool
            if (i == 0) {
                a.points = points; a.posf = (const uint2*)wb->posf; a.sc3 = wb->sc3;
                hipLaunchKernelGGL((k_pn_block<true, MH_FIRST>), grid, dim3(256 * MH_FIRST), lds_first, st, a);
            } else {
                a.x = cur; a.v0 = v0; a.vs = vs;
                hipLaunchKernelGGL((k_pn_block<false, MH_NEXT>), grid, dim3(256 * MH_NEXT), lds_next, st, a);
            }
            if ((rc = seeme_check_launch("k_pn_block"))) return rc;
        }
        if (i > 0) { unsigned short* t = cur; cur = nxt; nxt = t; } else { cur = nxt; nxt = xa; }
    }
    // fc_c(relu(pool of block_3))
    if (w->out_dim == 512) {
        PnRowsArgs ra{};
        ra.pool = pools + (size_t)3 * B * PN_H; ra.w[0] = w->fcc_w; ra.w[1] = w->fcc_w + (size_t)PN_H * PN_H;
        ra.bias[0] = w->fcc_b; ra.bias[1] = w->fcc_b + PN_H; ra.y[0] = out; ra.y[1] = out + PN_H; ra.ldy = 512;
        ra.ldw = PN_H; ra.col0 = 0; ra.relu[0] = ra.relu[1] = 1;
        hipLaunchKernelGGL(k_pn_rows, dim3((unsigned)B, 4), dim3(512), 0, st, ra);
        return seeme_check_launch("k_pn_rows");
    }
    return seeme_linear_simple(st, pools + (size_t)3 * B * PN_H, PN_H, w->fcc_w, PN_H, w->fcc_b, out, w->out_dim, B, w->out_dim,
                               PN_H, SEEME_ACT_NONE, SEEME_ACT_RELU, nullptr, nullptr);
}
