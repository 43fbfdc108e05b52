// pointnet_v2.hip -- second generation of the fused ResnetBlockFC kernel (EgoHMR/models/respointnet.py:62-97) for the
// frozen scene encoder (respointnet.py:33-59; mld/models/modeltype/mld.py:911-922).
//
// What bounded the first generation (pointnet_bf16.hip): every wave streamed its own quarter of
// the block's three weight matrices from L2 for every 64-point tile -- 384 KiB per 64 points per workgroup, i.e. the
// vector-memory path of a CU (64 B/clk) had to run flat out for the matrix cores to run flat out, and each ran at half.
// Here the weights of a block are ONE packed stream of 24 slots x 16 KiB (host-packed in the order of use) that a
// 512-thread workgroup pulls ONCE per 256-point tile into a three-slot LDS ring (16 B/clk of the vector-memory path),
// and all eight waves read their weight fragments from that ring (ds_read_b128, lane-linear, conflict-free):
//
//   wave w owns points [32 w, 32 w + 32) of the tile and ALL 256 features of them, with the weight fragment as the
//   MFMA A operand and the points as B ("transposed" calls: D[feature][point]):
//     fc_0        acc0[16 feature tiles][2 point tiles]          K = 256 (block_0: 512), B = relu(x) from registers
//     hidden      relu(acc0 + b0 + pooled half) -> bf16 B fragments IN REGISTERS: the accumulator layout (lane = point,
//                 registers = 4 consecutive features) is the B layout of the next product up to a permutation of k, which
//                 the host applies to fc_1's columns -- the hidden tile never touches LDS, and no barrier separates the GEMMs
//     out (two halves of 128 features, 64 accumulator registers each):
//                 acc1 = Ws[half] x (B = raw x) + W1[half] hidden  -> + bias, max-pool, bf16 store
//   The tile's input rows arrive as B fragments straight from global memory, one k-block (2 x 16 B per lane) per slot, two
//   slots ahead, through a three-entry register ring: each k-block is fetched three times per tile (fc_0, and the shortcut
//   of either half; the repeats hit L2) -- keeping the whole tile in registers (64 VGPRs) spilled.
//   block_0 generates its 512 input features relu(fc_pos_0(p)) per k-block on the matrix cores (split-bf16 operands,
//   v_mfma_f32_16x16x16_bf16) straight into B fragments, and its shortcut -- folded through fc_pos_0 to a 3 -> 256 map --
//   is 16 more of those small MFMAs per half into the same accumulator.
//
// Activations between blocks travel in FRAGMENT ORDER, not row-major: [scene][16-point tile][k-block 8][lane 64][8] bf16,
// i.e. the 16 bytes a lane stores (its bf16-packed accumulator values of the feature tiles 2 kb, 2 kb + 1 for one point) are
// the 16 bytes the same lane of the next block loads as its B fragment of k-block kb.  Every store and every load of an
// activation is then ONE fully contiguous KiB per wave instruction (row-major, a lane's 32-byte runs made the store tail of
// a block cost 85 us of its 550), and the weights of the next block are packed in the matching ("permuted") k order.
//
// Per slot a wave issues 32 MFMAs (2 waves per SIMD: 1024 matrix-core cycles), 16 ds_read_b128, 2 global loads and 2
// ds_write_b128 of the ring, one barrier.  LDS read traffic 128 B/clk per CU (half of its peak), vector-memory traffic
// about 25 B/clk.  All staging is plain loads + ds_write, so every wait is the compiler's own counted wait.
#include "common.hpp"
#include "api_util.hpp"
#include "pointnet_v2.h"

typedef __bf16 p2_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int p2_u32x4 __attribute__((ext_vector_type(4)));
typedef short p2_s16x4 __attribute__((ext_vector_type(4)));
typedef short p2_s16x2 __attribute__((ext_vector_type(2)));
typedef float p2_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 p2_bf16x2 __attribute__((ext_vector_type(2)));

#define P2_H 256
#ifndef P2_NW
#define P2_NW 8                   // waves per workgroup: 8 = one workgroup per CU, 4 = two independent ones (their phases interleave)
#endif
#define P2_NT (64 * P2_NW)        // threads
#define P2_MT (32 * P2_NW)        // points per tile (32 per wave)
#define P2_SLOTS 24               // 16-KiB slots of the weight stream per tile
#define P2_SLOT_U4 1024           // uint4 per slot
#define P2_RING 3                 // ring positions: slot s is read during step s (its first fragments already before the barrier that
                                  // opens the step), slot s + 2 is written during step s, slot s + 3 is in flight from L2
#ifndef P2_SPB
#define P2_SPB 1                  // steps per barrier: the ring turns (barrier, ds_write of the staged weights, next requests) every
#endif                            // P2_SPB steps, on ring positions of P2_SPB x 16 KiB
#define P2_NQ (P2_SLOTS / P2_SPB) // ring turns per tile
#define P2_RING_BYTES (P2_RING * P2_SPB * 16384)
#ifndef P2_PRE
#define P2_PRE 4                  // weight fragments read ahead of the MFMAs that consume them
#endif
static_assert(16 % P2_PRE == 0, "the fragment ring carries over from slot to slot: 16 fragments per slot must be a multiple of its depth");
#ifndef P2_WLA
#define P2_WLA (P2_SPB == 1 ? 2 : 1)  // ring turns a weight slot spends in staging registers between its load and its ds_write (L2 latency)
#endif
#ifndef P2_XLA
#define P2_XLA 2                  // steps between the load of an input k-block and its use (first touch comes from HBM)
#endif
static_assert(P2_SLOTS % P2_SPB == 0 && P2_NQ % P2_WLA == 0 && P2_NQ % P2_RING == 0 && P2_SLOTS % (P2_XLA + 1) == 0,
              "ring positions are compile-time constants across tiles");
#ifndef P2_PRIO
#define P2_PRIO 1                 // s_setprio of waves 4..7
#endif

__device__ __forceinline__ float p2_max(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, __builtin_inff()); }
__device__ __forceinline__ unsigned p2_pack(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(p2_f32x2{lo, hi}, p2_bf16x2));
}
// relu on two packed bf16: as signed 16-bit integers negative floats are negative, so it is one v_pk_max_i16 with 0
__device__ __forceinline__ unsigned p2_relu2(unsigned u) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(p2_s16x2, u), p2_s16x2{0, 0}));
}
__device__ __forceinline__ p2_u32x4 p2_relu8(p2_u32x4 v) { return p2_u32x4{p2_relu2(v.x), p2_relu2(v.y), p2_relu2(v.z), p2_relu2(v.w)}; }
__device__ __forceinline__ void p2_atomic_max(float* p, float v) {
    if (v >= 0.f) atomicMax(reinterpret_cast<int*>(p), __float_as_int(v));
    else atomicMin(reinterpret_cast<unsigned int*>(p), __float_as_uint(v));
}

#ifndef P2_PIN_LOADS
#define P2_PIN_LOADS 1   // scheduling fence behind a step's requests (measured: 2.08 -> 1.99 ms, and no scratch)
#endif
#ifndef P2_SPREAD_ST
#define P2_SPREAD_ST 0  // 1: the 8 KiB a wave produces per output half leave through LDS, one KiB per step, instead of as one burst
                        // (measured: 2.05 ms against 1.99 ms for the burst -- a store in every step's queue costs more than the burst)
#endif
#ifndef P2_X_AUX
#define P2_X_AUX 0      // cache-policy bits of the activation loads (bit 1 = non-temporal)
#endif
#ifndef P2_ST_AUX
#define P2_ST_AUX 0     // ... and of the activation stores
#endif
#ifdef P2_DBG_TIMES
// debug build only (scripts/pn2_times.py): cycle stamps of every wave of one workgroup around the barrier and the MFMA
// section of every step of its fourth tile
__device__ unsigned long long p2_dbg[8][P2_SLOTS][6];
#ifndef P2_DBG_FIRST
#define P2_DBG_FIRST 0
#endif
#define P2_STAMP(i) do { if (FIRST == (P2_DBG_FIRST != 0) && a.out != nullptr && blockIdx.x == 40 && t == t0 + 3 && lane == 0) p2_dbg[wave][s][i] = __builtin_readcyclecounter(); } while (0)
extern "C" int seeme_debug_pn2_times(unsigned long long* host) {
    SEEME_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(p2_dbg), sizeof(unsigned long long) * 8 * P2_SLOTS * 6));
    return 0;
}
#else
#define P2_STAMP(i) do {} while (0)
#endif

template <bool FIRST>
__global__ __launch_bounds__(P2_NT, 2) void k_pn_block2(const PnBlock2Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // LDS carve (bytes): ring 3 x 16384 | sb0 1024 | sb1 1024 | smax P2_NW x 1024 | block_0: sposf 16384 | ssc3f 8192
    uint4* const ring = reinterpret_cast<uint4*>(smem);
    float* const sb0 = reinterpret_cast<float*>(smem + P2_RING_BYTES);
    float* const sb1 = sb0 + P2_H;
    float* const smax = sb1 + P2_H;                                   // [8][256]
    const uint2* const sposf = reinterpret_cast<const uint2*>(smem + P2_RING_BYTES + 2048 + P2_NW * 1024);
    const uint2* const ssc3f = sposf + 32 * 64;
#if P2_SPREAD_ST
    // staging of the block output: [wave][8 KiB], written at the end of an output half, drained one KiB per step.  Measured
    // with cycle stamps: issued as 8 stores at once, an epilogue step took 5.8-7.4 k cycles instead of ~1.6 k -- every CU
    // reaches its epilogue at about the same time, and 256 x 64 KiB leave for HBM in one burst, twice per tile.
    uint4* const sout = reinterpret_cast<uint4*>(smem + P2_RING_BYTES + 2048 + P2_NW * 1024 + (FIRST ? 16384 + 8192 : 0)) + (threadIdx.x >> 6) * 512 + (threadIdx.x & 63);
#endif

    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, r = lane & 15, kq = lane >> 4;
    const int row0 = wave * 32;

    if (FIRST) {
        uint2* wp = reinterpret_cast<uint2*>(smem + P2_RING_BYTES + 2048 + P2_NW * 1024);
        for (int c = tid; c < 32 * 64; c += P2_NT) wp[c] = a.posf[c];
        for (int c = tid; c < 16 * 64; c += P2_NT) wp[32 * 64 + c] = a.sc3f[c];
    }

    // ---- the weight stream: slot q of the tile program = stream[q % 24]; this wave moves 2 KiB of each slot
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(a.stream), 0, P2_SLOTS * 16384, 0x00020000);
    constexpr int WP = 16 * P2_SPB / P2_NW;                               // KiB pieces of a ring position per wave
    const unsigned w_lane = (unsigned)(wave * WP * 1024 + lane * 16);     // byte offset of this lane's 16 B inside a slot (first piece)
    p2_u32x4 stg[P2_WLA][WP];               // slot q waits in set q % P2_WLA: stored into the ring at the top of step q - 2, re-filled right after
    auto w_load = [&](int slot_in_tile, p2_u32x4 (&dst)[WP]) {
#pragma unroll
        for (int i = 0; i < WP; ++i) dst[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_lane + (unsigned)(i * 1024), (unsigned)(slot_in_tile * (P2_SPB * 16384)), 0);
    };
    auto w_store = [&](int ring_pos, const p2_u32x4 (&src)[WP]) {
        uint4* d = ring + ring_pos * (P2_SPB * P2_SLOT_U4) + wave * (WP * 64) + lane;
#pragma unroll
        for (int i = 0; i < WP; ++i) d[i * 64] = make_uint4(src[i].x, src[i].y, src[i].z, src[i].w);
    };
    const uint4* const rl = ring + lane;                                   // fragment f of ring position p: rl[p * 1024 + f * 64]

    // ---- tile range of this workgroup (contiguous: mostly one scene, so the running max stays in a register)
    const int t0 = (int)(((long long)blockIdx.x * a.n_tiles) / gridDim.x), t1 = (int)(((long long)(blockIdx.x + 1) * a.n_tiles) / gridDim.x);
    if (t0 >= t1) return;
    float run_max = -INFINITY;
    int prev_scene = -1, bias_scene = t0 / a.tiles_x;

    // ---- input of a tile: B fragments in registers.  later blocks: xf[kb][mt] = x[point row0 + 16 mt + r][32 kb + 8 kq .. +7];
    // block_0: the point as split-bf16 operand pfr[mt] of the small MFMAs
    p2_u32x4 xr[FIRST ? 1 : P2_XLA + 1][2]; // later blocks: x k-block ring, entry = step % (P2_XLA + 1)
    p2_s16x4 pfr[2];
    float pxyz[FIRST ? 2 : 1][3];
    auto tile_rsrc = [&](int tn) {           // buffer over the tile's 16-point groups that hold valid rows (the rest reads as zeros)
#ifdef P2_ABL_XSAME                          // timing-only ablation: every tile reads the activations of tile 0 (L2-resident; wrong results)
        tn = 0;
#endif
        const int sc = tn / a.tiles_x, q0 = (tn - sc * a.tiles_x) * P2_MT, rv = min(P2_MT, a.P - q0);
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.x + ((size_t)sc * a.Ppad + q0) * P2_H), 0,
                                                 ((rv + 15) / 16) * 16 * P2_H * 2, 0x00020000);
    };
    // fragment (point tile pt of the workgroup's tile, k-block kb) = 1 KiB at ((pt * 8 + kb) * 1024); this wave's point tiles: 2 wave + mt
    const unsigned f_lane = (unsigned)(wave * 2 * 8192 + lane * 16);
    auto x_load = [&](const __amdgpu_buffer_rsrc_t& rt, int kb, p2_u32x4 (&dst)[2]) {
        dst[0] = __builtin_amdgcn_raw_buffer_load_b128(rt, f_lane, (unsigned)(kb * 1024), P2_X_AUX);
        dst[1] = __builtin_amdgcn_raw_buffer_load_b128(rt, f_lane, (unsigned)(kb * 1024 + 8192), P2_X_AUX);
    };
    auto issue_pts = [&](int tn) {
        const int sc = tn / a.tiles_x, q0 = (tn - sc * a.tiles_x) * P2_MT, rv = min(P2_MT, a.P - q0);
        const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.points + ((size_t)sc * a.P + q0) * 3), 0, rv * 12, 0x00020000);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                pxyz[mt][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rp, (unsigned)((row0 + r) * 12), (unsigned)(mt * 192 + c * 4), 0));
    };
    auto make_pfr = [&]() {                   // block_0: (px, py, pz) -> hi/lo split B operand, k slots as SeemePointnetBf16.posf expects
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const float px = pxyz[mt][0], py = pxyz[mt][1], pz = pxyz[mt][2];
            const unsigned hxy = p2_pack(px, py), hz = p2_pack(pz, 0.f) & 0xFFFFu;
            const unsigned lxy = p2_pack(px - __uint_as_float(hxy << 16), py - __uint_as_float(hxy & 0xFFFF0000u));
            const unsigned lz = p2_pack(pz - __uint_as_float(hz << 16), 0.f) & 0xFFFFu;
            const unsigned one = 0x3F80u;
            const unsigned d0 = kq == 0 ? hxy : kq == 1 ? ((lxy >> 16) | (lz << 16)) : kq == 2 ? (hz | (one << 16)) : 0u;
            const unsigned d1 = kq == 0 ? (hz | (lxy << 16)) : kq == 1 ? hxy : kq == 2 ? one : 0u;
            pfr[mt] = __builtin_bit_cast(p2_s16x4, make_uint2(d0, d1));
        }
    };

#ifdef P2_STAGGER
    // all workgroups run the same 24-step program at the same pace: without this they stay in phase, and the memory system sees
    // every CU's first-touch loads and every CU's store tail at the same moments.  Spread the starting times over one tile period.
    for (int i = (int)(blockIdx.x % 16u) * P2_STAGGER; i > 0; --i) __builtin_amdgcn_s_sleep(32);
#endif
    // ---- prologue: slot 0 into the ring, slot 1 staged, the first tile's input requested
    {
        p2_u32x4 s0[WP], s1[WP];
        w_load(0, s0);
        w_load(1, s1);
        w_store(0, s0);
        w_store(1, s1);
    }
#pragma unroll
    for (int q = 2; q < 2 + P2_WLA; ++q) w_load(q, stg[q % P2_WLA]);
    __amdgpu_buffer_rsrc_t rx = FIRST ? rs_w : tile_rsrc(t0);
    if (FIRST) issue_pts(t0);
    else {
#pragma unroll
        for (int q = 0; q < P2_XLA; ++q) x_load(rx, q, xr[q]);
    }
    if (tid < P2_H) {                          // biases of the first tile's scene (later tiles: at step 1 / step 0 of the tile loop)
        const int sc0 = t0 / a.tiles_x;
        sb0[tid] = a.b0[tid] + (FIRST ? 0.f : a.v0[(size_t)sc0 * P2_H + tid]);
        sb1[tid] = a.b1[tid] + (FIRST ? 0.f : a.vs[(size_t)sc0 * P2_H + tid]);
    }
    // static priority for the second-dispatched half of the workgroup: with both waves of a SIMD at priority 0 the older
    // one wins every arbitration and finishes its step ~600 cycles before its partner (stamps), which it then spends at the barrier
    if (P2_NW == 8 && __builtin_amdgcn_readfirstlane(tid) >= 256) __builtin_amdgcn_s_setprio(P2_PRIO);
    __syncthreads();
    uint4 af[P2_PRE];                      // the first fragments of the slot about to be consumed
#pragma unroll
    for (int i = 0; i < P2_PRE; ++i) af[i] = rl[i * 64];

#if P2_SPREAD_ST
    __amdgpu_buffer_rsrc_t ro_prev = __builtin_amdgcn_make_buffer_rsrc((unsigned short*)nullptr, 0, 0, 0x00020000);
    bool pend_prev = false;
#endif
    for (int t = t0; t < t1; ++t) {
        const int scene = t / a.tiles_x, p0 = (t - scene * a.tiles_x) * P2_MT;
        const int rows_valid = min(P2_MT, a.P - p0);
        const bool more = t + 1 < t1;
        if (FIRST) make_pfr();
        const bool has_out = a.out != nullptr;
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
            has_out ? a.out + ((size_t)scene * a.Ppad + p0) * P2_H : nullptr, 0, has_out ? ((rows_valid + 15) / 16) * 16 * P2_H * 2 : 0, 0x00020000);

        f32x4 acc0[16][2];
        p2_u32x4 hf[8][2];
        f32x4 acc1[8][2];
#pragma unroll
        for (int s = 0; s < P2_SLOTS; ++s) {
            // (P2_SPREAD_ST drain: steps 16..23 carry half 0 of this tile, steps 0..7 half 1 of the previous tile; piece j = step % 8
            // = feature k-block 4 g + j / 2 of point tile j % 2 -- issued behind this step's barrier and requests, below)
            // ---- ring turn: slot s was written one step ago; after the barrier it is readable and the other position is free
            P2_STAMP(0);
            const int q = s / P2_SPB, h = s % P2_SPB;               // ring turn of the tile, step inside the turn
            if (h == 0) {
#ifndef P2_ABL_NOBAR
                __syncthreads();
#endif
#ifndef P2_ABL_NORING
                if (q + 2 < P2_NQ || more) w_store((q + 2) % P2_RING, stg[(q + 2) % P2_WLA]);
                if (q + 2 + P2_WLA < P2_NQ) w_load(q + 2 + P2_WLA, stg[(q + 2) % P2_WLA]);
                else if (more) w_load(q + 2 + P2_WLA - P2_NQ, stg[(q + 2) % P2_WLA]);
#endif
            }
            P2_STAMP(1);
            const uint4* const rp = rl + ((q % P2_RING) * P2_SPB + h) * P2_SLOT_U4;
            // the next step's fragments: inside this ring position, or the next position (complete since the barrier above)
            const uint4* const rn = rl + (h + 1 < P2_SPB ? (q % P2_RING) * P2_SPB + h + 1 : ((q + 1) % P2_RING) * P2_SPB) * P2_SLOT_U4;
            if (!FIRST) {                                      // x k-block of step s + 2 (k-block = step % 8), across the tile boundary
#ifndef P2_ABL_NOX
                if (s + P2_XLA == P2_SLOTS && more) rx = tile_rsrc(t + 1);
#endif
#ifndef P2_ABL_NOXLOAD
                if (s + P2_XLA < P2_SLOTS || more) x_load(rx, (s + P2_XLA) % 8, xr[(s + P2_XLA) % (P2_XLA + 1)]);
#endif
            }
#if P2_SPREAD_ST
            {
                // half 0 is complete at the end of step E0 (15; block_0: 19) and leaves during the steps up to 23; half 1 is
                // complete at the end of step 23 and leaves during steps 0..7 of the next tile
                constexpr int E0 = FIRST ? 19 : 15, PPS = 8 / (P2_SLOTS - 1 - E0);      // pieces per step of half 0
                if (s > E0) {
#pragma unroll
                    for (int u = 0; u < PPS; ++u) {
                        const int j = (s - E0 - 1) * PPS + u;
                        const uint4 v = sout[j * 64];
                        __builtin_amdgcn_raw_buffer_store_b128(p2_u32x4{v.x, v.y, v.z, v.w}, ro, f_lane,
                                                               (unsigned)((j >> 1) * 1024 + (j & 1) * 8192), P2_ST_AUX);
                    }
                } else if (s < 8 && pend_prev) {
                    const uint4 v = sout[s * 64];
                    __builtin_amdgcn_raw_buffer_store_b128(p2_u32x4{v.x, v.y, v.z, v.w}, ro_prev, f_lane,
                                                           (unsigned)((4 + (s >> 1)) * 1024 + (s & 1) * 8192), P2_ST_AUX);
                }
            }
#endif
#if P2_PIN_LOADS
            // keep the requests HERE: under register pressure the scheduler sinks them towards their use (ISA: input k-blocks
            // requested 0-40 MFMAs before use instead of 64), which turns the look-ahead into a stall on HBM latency
            __builtin_amdgcn_sched_barrier(0);
#endif

            if (s == 0) {
                // (the two rare, branchy pieces of a tile sit here, where only the rings are live: next to the accumulators
                // they made the register allocator spill accumulators around their branches)
                // fold the previous tile's column maxima (complete since the barrier above) into the running one
                if (tid < P2_H && prev_scene >= 0) {
                    float m = smax[tid];
#pragma unroll
                    for (int w = 1; w < P2_NW; ++w) m = p2_max(m, smax[w * P2_H + tid]);
                    run_max = p2_max(run_max, m);
                    if (scene != prev_scene) {
                        p2_atomic_max(a.pool + (size_t)prev_scene * P2_H + tid, run_max);
                        run_max = -INFINITY;
                    }
                }
                prev_scene = scene;
                // fc_1 / shortcut bias + pooled half of THIS tile's scene (read from step 8 on; the previous tile's reads ended
                // before the barrier above)
                if (!FIRST && scene != bias_scene) {
                    if (tid < P2_H) sb1[tid] = a.b1[tid] + a.vs[(size_t)scene * P2_H + tid];
                    bias_scene = scene;
                }
                __builtin_amdgcn_sched_barrier(0);
                // the accumulator starts at the bias (+ pooled half): feature tile nt holds features 16 nt + 4 kq + i
#pragma unroll
                for (int nt = 0; nt < 16; ++nt) {
                    const float4 b4 = *reinterpret_cast<const float4*>(sb0 + 16 * nt + 4 * kq);
                    acc0[nt][0] = f32x4{b4.x, b4.y, b4.z, b4.w};
                    acc0[nt][1] = acc0[nt][0];
                }
            }
            if (s == 1 && !FIRST && more) {
                // fc_0 bias + pooled half of the NEXT tile's scene (read at its step 0, behind this tile's remaining barriers)
                const int scn = (t + 1) / a.tiles_x;
                if (scn != scene && tid < P2_H) sb0[tid] = a.b0[tid] + a.v0[(size_t)scn * P2_H + tid];
            }

            constexpr int S0 = FIRST ? 16 : 8;                 // slots of fc_0
            if (s < S0) {
                // ---- fc_0, k-block s: acc0[nt][mt] += W0frag(nt) x relu(x)[mt]
                p2_u32x4 b[2];
                if (FIRST) {
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        const f32x4 c0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(p2_s16x4, sposf[(2 * s) * 64 + lane]), pfr[mt], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                        const f32x4 c1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(p2_s16x4, sposf[(2 * s + 1) * 64 + lane]), pfr[mt], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                        b[mt] = p2_relu8(p2_u32x4{p2_pack(c0[0], c0[1]), p2_pack(c0[2], c0[3]), p2_pack(c1[0], c1[1]), p2_pack(c1[2], c1[3])});
                    }
                } else {
                    b[0] = p2_relu8(xr[FIRST ? 0 : s % (P2_XLA + 1)][0]);
                    b[1] = p2_relu8(xr[FIRST ? 0 : s % (P2_XLA + 1)][1]);
                }
#pragma unroll
                for (int nt = 0; nt < 16; ++nt) {
                    const p2_bf16x8 wa = __builtin_bit_cast(p2_bf16x8, af[nt % P2_PRE]);
#ifndef P2_ABL_NOLDS
                    af[nt % P2_PRE] = nt + P2_PRE < 16 ? rp[(nt + P2_PRE) * 64] : rn[(nt + P2_PRE - 16) * 64];
#endif
#ifndef P2_ABL_NOMFMA
                    acc0[nt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, __builtin_bit_cast(p2_bf16x8, b[0]), acc0[nt][0], 0, 0, 0);
                    acc0[nt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, __builtin_bit_cast(p2_bf16x8, b[1]), acc0[nt][1], 0, 0, 0);
#else
                    acc0[nt][0][0] += __builtin_bit_cast(float, af[nt % P2_PRE].x ^ b[0].x); acc0[nt][1][0] += __builtin_bit_cast(float, af[nt % P2_PRE].y ^ b[1].x);
                    asm volatile("" :: "v"(wa));
#endif
                }
#ifndef P2_NO_SGB
#pragma unroll
                for (int nt = 0; nt < 16; ++nt) {          // pin the interleave: one fragment read, then the two MFMAs of an older one
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                }
#endif
                if (s == S0 - 1) {
                    // ---- hidden = relu(acc0 + bias) as the B fragments of fc_1 (k order: see the header; W1 is packed to match)
#pragma unroll
                    for (int kb = 0; kb < 8; ++kb) {
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) {
                            const f32x4 u = acc0[2 * kb][mt], v = acc0[2 * kb + 1][mt];
                            hf[kb][mt] = p2_relu8(p2_u32x4{p2_pack(u[0], u[1]), p2_pack(u[2], u[3]), p2_pack(v[0], v[1]), p2_pack(v[2], v[3])});
                        }
                    }
                }
            } else {
                // ---- output halves.  later blocks: slot S0 + 8 g + kb = Ws[half g] k-block kb (fragments 0..7, B = raw x) and
                // W1[half g] k-block kb (fragments 8..15, B = hidden); block_0: slot S0 + 4 g + p = W1[half g] k-blocks 2p, 2p + 1
                constexpr int PER_HALF = FIRST ? 4 : 8;
                const int g = (s - S0) / PER_HALF, q = (s - S0) % PER_HALF;
                if (q == 0) {
                    // the accumulator starts at the bias (+ pooled half): tile 8 g + n holds features 16 (8 g + n) + 4 kq + i
#pragma unroll
                    for (int n = 0; n < 8; ++n) {
                        const float4 b4 = *reinterpret_cast<const float4*>(sb1 + 16 * (8 * g + n) + 4 * kq);
                        acc1[n][0] = f32x4{b4.x, b4.y, b4.z, b4.w};
                        acc1[n][1] = acc1[n][0];
                    }
                }
#pragma unroll
                for (int part = 0; part < 2; ++part) {
                    const int kb = FIRST ? 2 * q + part : q;
                    p2_u32x4 b0v, b1v;
                    if (!FIRST && part == 0) { b0v = xr[FIRST ? 0 : s % (P2_XLA + 1)][0]; b1v = xr[FIRST ? 0 : s % (P2_XLA + 1)][1]; }
                    else { b0v = hf[kb][0]; b1v = hf[kb][1]; }
#pragma unroll
                    for (int n = 0; n < 8; ++n) {
                        const int f = part * 8 + n;
                        const p2_bf16x8 wa = __builtin_bit_cast(p2_bf16x8, af[f % P2_PRE]);
#ifndef P2_ABL_NOLDS
                        af[f % P2_PRE] = f + P2_PRE < 16 ? rp[(f + P2_PRE) * 64] : rn[(f + P2_PRE - 16) * 64];
#endif
#ifndef P2_ABL_NOMFMA
                        acc1[n][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, __builtin_bit_cast(p2_bf16x8, b0v), acc1[n][0], 0, 0, 0);
                        acc1[n][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, __builtin_bit_cast(p2_bf16x8, b1v), acc1[n][1], 0, 0, 0);
#else
                        acc1[n][0][0] += __builtin_bit_cast(float, af[f % P2_PRE].x ^ b0v.x); acc1[n][1][0] += __builtin_bit_cast(float, af[f % P2_PRE].y ^ b1v.x);
                        asm volatile("" :: "v"(wa));
#endif
                    }
                }
#ifndef P2_NO_SGB
#pragma unroll
                for (int f = 0; f < 16; ++f) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                }
#endif
#ifdef P2_ABL_NOEPI
                if (q == PER_HALF - 1 && rows_valid < 0) {
#else
                if (q == PER_HALF - 1) {
#endif
                    P2_STAMP(3);
                    if (FIRST) {
                        // folded shortcut (3 -> 256, bias included) on the matrix cores, into the same accumulator
#pragma unroll
                        for (int n = 0; n < 8; ++n)
#pragma unroll
                            for (int mt = 0; mt < 2; ++mt)
                                acc1[n][mt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(p2_s16x4, ssc3f[(8 * g + n) * 64 + lane]), pfr[mt], acc1[n][mt], 0, 0, 0);
                        if (g == 1 && more) issue_pts(t + 1);                 // (pfr holds this tile's points until make_pfr of the next)
                    }
                    // ---- epilogue of half g.  Store: the packed values of feature tiles (2 kl, 2 kl + 1) of one point ARE the next
                    // block's B fragment of k-block 4 g + kl: one contiguous KiB per wave instruction.
                    if (has_out) {
#pragma unroll
                        for (int kl = 0; kl < 4; ++kl)
#pragma unroll
                            for (int mt = 0; mt < 2; ++mt) {
                                const f32x4 u = acc1[2 * kl][mt], v = acc1[2 * kl + 1][mt];
#if P2_SPREAD_ST
                                sout[(2 * kl + mt) * 64] = make_uint4(p2_pack(u[0], u[1]), p2_pack(u[2], u[3]), p2_pack(v[0], v[1]), p2_pack(v[2], v[3]));
#elif !defined(P2_ABL_NOSTORE)
                                __builtin_amdgcn_raw_buffer_store_b128(p2_u32x4{p2_pack(u[0], u[1]), p2_pack(u[2], u[3]), p2_pack(v[0], v[1]), p2_pack(v[2], v[3])},
                                                                       ro, f_lane, (unsigned)((4 * g + kl) * 1024 + mt * 8192), P2_ST_AUX);
#endif
                            }
                    }
                    P2_STAMP(4);
                    // Column max over the wave's 32 points, four feature tiles (16 values per lane) at a time: reduce-scatter over
                    // the 16 lanes of a DPP row (15 exchanges); lane r ends with value r = 4 t + i of the group: feature
                    // 16 (8 g + 4 gl + t) + 4 kq + i
#pragma unroll
                    for (int gl = 0; gl < 2; ++gl) {
                        float mx[16];
                        const bool v0ok = row0 + r < rows_valid, v1ok = row0 + 16 + r < rows_valid;
#pragma unroll
                        for (int tq = 0; tq < 4; ++tq)
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                mx[4 * tq + i] = p2_max(v0ok ? acc1[4 * gl + tq][0][i] : -INFINITY, v1ok ? acc1[4 * gl + tq][1][i] : -INFINITY);
                        auto xch = [](float send, int sel) {
                            const int iv = __float_as_int(send);
                            return __int_as_float(sel == 0 ? __builtin_amdgcn_update_dpp(0, iv, 0x128, 0xF, 0xF, true)
                                                : sel == 1 ? __builtin_amdgcn_update_dpp(0, iv, 0x141, 0xF, 0xF, true)
                                                : sel == 2 ? __builtin_amdgcn_update_dpp(0, iv, 0x4E, 0xF, 0xF, true)
                                                           : __builtin_amdgcn_update_dpp(0, iv, 0xB1, 0xF, 0xF, true));
                        };
                        float a8[8], a4[4], a2[2], a1;
                        const bool b3 = r & 8, b2 = r & 4, b1 = r & 2, b0b = r & 1;
#pragma unroll
                        for (int j = 0; j < 8; ++j) a8[j] = p2_max(b3 ? mx[j + 8] : mx[j], xch(b3 ? mx[j] : mx[j + 8], 0));
#pragma unroll
                        for (int j = 0; j < 4; ++j) a4[j] = p2_max(b2 ? a8[j + 4] : a8[j], xch(b2 ? a8[j] : a8[j + 4], 1));
#pragma unroll
                        for (int j = 0; j < 2; ++j) a2[j] = p2_max(b1 ? a4[j + 2] : a4[j], xch(b1 ? a4[j] : a4[j + 2], 2));
                        a1 = p2_max(b0b ? a2[1] : a2[0], xch(b0b ? a2[0] : a2[1], 3));
                        if (has_out) a1 = __uint_as_float(p2_pack(a1, 0.f) << 16);   // pool the value the next block reads (rounding is monotone)
                        smax[wave * P2_H + 16 * (8 * g + 4 * gl + (r >> 2)) + 4 * kq + (r & 3)] = a1;
                        __builtin_amdgcn_sched_barrier(0);       // one group at a time: the epilogue's temporaries are not doubled
                    }
                }
            }
            P2_STAMP(2);
        }
#if P2_SPREAD_ST
        ro_prev = ro;
        pend_prev = has_out;
#endif
    }
#if P2_SPREAD_ST
    if (pend_prev) {                       // half 1 of the last tile
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint4 v = sout[j * 64];
            __builtin_amdgcn_raw_buffer_store_b128(p2_u32x4{v.x, v.y, v.z, v.w}, ro_prev, f_lane, (unsigned)((4 + (j >> 1)) * 1024 + (j & 1) * 8192), P2_ST_AUX);
        }
    }
#endif
    __syncthreads();
    if (tid < P2_H && prev_scene >= 0) {
        float m = smax[tid];
#pragma unroll
        for (int w = 1; w < P2_NW; ++w) m = p2_max(m, smax[w * P2_H + tid]);
        p2_atomic_max(a.pool + (size_t)prev_scene * P2_H + tid, p2_max(run_max, m));
    }
}

// launch helper used by seeme_pointnet_encode_bf16 (pointnet_bf16.hip)
int seeme_pn_block2_launch(bool first, const PnBlock2Args& a, int n_cu, hipStream_t st) {
    const size_t lds = P2_RING_BYTES + 2048 + P2_NW * 1024 + (first ? 16384 + 8192 : 0) + (P2_SPREAD_ST ? P2_NW * 8192 : 0);
    const int per_cu = 8 / P2_NW;
    const dim3 grid((unsigned)(a.n_tiles < n_cu * per_cu ? a.n_tiles : n_cu * per_cu));
    if (first) {
        SEEME_HIP(hipFuncSetAttribute((const void*)k_pn_block2<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_pn_block2<true>), grid, dim3(P2_NT), lds, st, a);
    } else {
        SEEME_HIP(hipFuncSetAttribute((const void*)k_pn_block2<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_pn_block2<false>), grid, dim3(P2_NT), lds, st, a);
    }
    return seeme_check_launch("k_pn_block2");
}
int seeme_pn_block2_tile_points() { return P2_MT; }
