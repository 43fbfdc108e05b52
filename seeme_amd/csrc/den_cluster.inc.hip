// den_cluster.inc.hip -- the sampling loop with ONE SAMPLE SPLIT OVER C WORKGROUPS (CUs)  (included by den_kernels.hip).
//
// Reference: the same chain as k_den_sample -- MldDenoiser.forward (mld_denoiser.py:151-244) with
// LinearTemporalDiffusionTransformerDecoderLayer (mdiff_transformer.py:286-304) inside SkipTransformerEncoder
// (cross_attention.py:67-83), driven by MLD._diffusion_reverse (mld.py:467-497).
//
// Why: at B = 32 (BASELINE configs[1]) k_den_sample keeps 32 of 256 CUs busy and each of them is bound by the 118 GB/s of
// one CU's vector-memory path (84 us per step for the 9 MB image).  Here a "cluster" of C = 2, 4 or 8 workgroups (one per CU)
// owns a sample: the three big matrices of a layer are split between them, the small ones are replicated, and a step needs
// TWO exchanges per layer:
//
//   A   in_proj' (q | k | W_o v [| y = skip linear, folded: layers 3, 4])  COLUMN-split by dims: CU c computes dims
//       c S .. c S + S - 1 (S = 256 / C) of q, k, v' (and y) from the full input vector (replicated)
//   X1  every CU publishes v'[S] (, y[S]) and its partial scores q.k over its dims (self, condition, time token); every CU
//       gathers them -> softmax over the 2 + N tokens, attention output, + residual, norm1: replicated, bit-identical
//   B   linear1 COLUMN-split (1024 / C hidden units per CU), ReLU                      -- no exchange
//   C   linear2 ROW-split over the CU's hidden units -> 256 partial sums per CU
//   X2  all-reduce of the 256 partial sums (fixed order of addition) -> + bias, residual, norm2, + ca_block term: replicated
//   D E F  ffn.linear1 / linear2 / proj_out (64 + 64 + 128 KB at 16 bit) REPLICATED on every CU -- no exchange
//
// The skip linears of layers 3 and 4 are folded into that layer's in_proj' on the host ([W_in' W_s ; W_s] acting on
// cat[x, skip], seeme_amd/mld_denoiser.py): no extra exchange.  One attention head, one condition token (the ca_block term
// comes from seeme_denoiser_ca_tables), no CFG: the configuration BASELINE's headline names; everything else stays on
// k_den_sample.
//
// Exchange = cdna_hip_programming.md Guideline 16, R2: every value travels as one aligned 8-byte {tag = epoch, value}
// granule written by ONE store; the epilogue wave of every CU re-reads its granules with sc1 loads (bypass L1) until every
// tag equals the epoch.  The epoch counts exchanges within the launch (never 0); the host zeroes the granule buffer before
// every launch.  X1 and X2 alternate, so a buffer is only rewritten after every reader has passed the exchange in between.
// Stores are write-through (sc1: visible to any CU of the chip) unless the cluster has established in its first exchange
// (which is always write-through) that all its workgroups report the same XCC id: then plain stores, which stay in that
// XCD's L2 where the peers' sc1 loads find them (probes/xchg_test.hip: 1.25 us instead of 2.0 us per exchange at C = 8).
// Correctness never depends on placement; speed does (blockIdx b and b + 8 share an XCD under round-robin dispatch).
// Every spin is bounded: a cluster that gives up sets a word of the exchange buffer's header and finishes without waiting.
//
// Weights: `units` of 8 waves x UL wave-loads x 1 KiB in order of use ([layer][CU][unit][wave][load][lane][16 B]); a wave-load
// is the MFMA B operand of one (16-output tile, k-block) pair.  A ring of RU units in registers runs RU units (half a layer)
// ahead of the consumer, across stage, layer and step boundaries.

#define DCL_MAGIC_EPOCH 0x7fffffffu
#define DCL_SPIN_LIMIT (1u << 22)

typedef __attribute__((address_space(1))) unsigned long long dcl_gu64;

template <typename WT> struct ClW {
    static constexpr int KL = WT::MFMA ? 32 : 16;      // k per wave-load (16-bit: v_mfma_f32_16x16x32; fp32: 4 x v_mfma_f32_16x16x4)
    static constexpr int UL = 256 / KL;                // wave-loads per unit
    static constexpr int RU = WT::MFMA ? 4 : 2;        // units in the register ring (32 x 16 B per lane either way)
    static constexpr unsigned UNIT_BYTES = 8u * UL * 1024u;
};
template <int C, bool Q = false> struct ClG {          // Q: several condition tokens -- the ca_block keeps its query and proj_out stages (G, H)
    static constexpr int S = 256 / C;                  // dims of q / k / v' / y per CU
    static constexpr int NB = 1024 / C;                // hidden units of the sa_block MLP per CU
    static constexpr int TA = S / 32;                  // tiles per wave in stage A (waves 0,1: q; 2,3: k; 4,5: v'; 6,7: y) = units per k-half
    static constexpr int TB = 8 / C;                   // tiles per wave in stage B = units of stage B = units of stage C
    static constexpr int U_A = 0, U_AS = TA, U_B = 2 * TA, U_C = U_B + TB;
    static constexpr int U_G = U_C + TB, U_H = U_G + 1;                       // (Q only) ca query: one unit, waves < S / 16; ca proj_out: two units
    static constexpr int U_D = Q ? U_H + 2 : U_C + TB, U_E = U_D + 1, U_F = U_E + 1, NU_REAL = U_F + 2;
    static constexpr int NU = (NU_REAL + 3) / 4 * 4;   // padded with phantom units: ring positions repeat per layer
    static constexpr int SC = Q ? 8 : 4;               // partial-score granules a CU publishes (2 + N values, padded)
    static constexpr int X1_G = 2 * S + SC;            // granules a CU publishes in X1: v'[S] | y[S] | partial scores [SC]
    // granules per sample: boot [16] | X1 [C][X1_G] | X2 [C][256] | X3 [C][8] (Q: partial softmax / key products of the ca query)
    static constexpr int G_BOOT = 0, G_X1 = 16, G_X2 = G_X1 + C * X1_G, G_X3 = G_X2 + C * 256, G_TOTAL = G_X3 + (Q ? C * 8 : 0);
    static constexpr int EPL = Q ? 3 : 2;              // exchanges (epochs) per layer
};
#define DCL_HDR_BYTES 256   // header of the exchange buffer: [0] give-up code, [1] clusters that used L2-local stores

template <typename WT> struct ClRing { u32x4 r[ClW<WT>::RU][ClW<WT>::UL]; };

struct ClArgs {
    const void* wgc; unsigned wgc_bytes; const float* vp;
    DenLayout lay;
    SeemeSampleArgs s;
    unsigned long long* xg;      // granules (behind the header)
    unsigned* hdr;
    int placement, flags, clusters;
    int spc;                     // k_den_cluster_ms: samples per cluster (<= 8)
};

// ---- one unit of the weight stream -> its ring slot
template <typename WT, int C, bool Q, int U>
__device__ __forceinline__ void cl_issue(ClRing<WT>& ring, int wave, unsigned voff, __amdgpu_buffer_rsrc_t rsrc,
                                         unsigned base_cur, unsigned base_next, bool skip_cur, bool skip_next) {
    typedef ClW<WT> W; typedef ClG<C, Q> G;
    constexpr int UU = U % G::NU;
    constexpr bool NXT = U >= G::NU;
    static_assert(U < 2 * G::NU && G::NU % W::RU == 0, "ring positions must repeat per layer");
    if constexpr (UU >= G::NU_REAL) return;            // phantom unit (padding of the layer program)
    if constexpr (Q && UU == G::U_G) { if (wave >= G::S / 16) return; }       // the ca query has S / 16 tiles: one per wave
    const bool sk = NXT ? skip_next : skip_cur;
    if constexpr (UU < G::U_B) {                       // stage A: the skip half only where the layer has a skip linear; the y waves likewise
        if constexpr (UU >= G::U_AS) { if (!sk) return; }
        if (wave >= 6 && !sk) return;
    }
    const unsigned soff = (NXT ? base_next : base_cur) + (unsigned)UU * W::UNIT_BYTES;
#pragma unroll
    for (int i = 0; i < W::UL; ++i) ring.r[UU % W::RU][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff + (unsigned)(i * 1024), 0);
}

// ---- GEMV input vectors in LDS.  16-bit weights: MFMA A-operand fragments [k-block][k-group 4][row 4][8 halves], row p =
// part p of the fp32 value (put_x); fp32 weights: the plain fp32 vector.
struct ClX { const char* base; int foff; int kbs; };     // kbs: bytes per k-block of the fragment buffer (rows x 4 k-groups x 16 B)
template <typename WT>
__device__ __forceinline__ ClX cl_xin(const float* buf, int lane) {
    ClX x; x.base = reinterpret_cast<const char*>(buf);
    x.foff = WT::MFMA ? ((lane >> 4) * 4 + ((lane & 15) & 3)) * 16 : (lane >> 4) * 16;
    x.kbs = WT::MFMA ? 256 : 64;
    return x;
}
template <typename WT>
__device__ __forceinline__ void cl_put1(float* buf, int k, float v) {     // one element (an MFMA output lane)
    if constexpr (!WT::MFMA) { buf[k] = v; }
    else {
        char* dst = reinterpret_cast<char*>(buf) + (((k >> 5) * 4 + ((k >> 3) & 3)) * 4) * 16 + (k & 7) * 2;
        if constexpr (WT::HALF) {
            const _Float16 hi = (_Float16)v;
            *reinterpret_cast<_Float16*>(dst) = hi;
            *reinterpret_cast<_Float16*>(dst + 16) = (_Float16)((v - (float)hi) * DEN_F16_LO_SCALE);
        } else {
            const __bf16 hi = (__bf16)v; const float r1 = v - (float)hi; const __bf16 mid = (__bf16)r1;
            *reinterpret_cast<__bf16*>(dst) = hi;
            *reinterpret_cast<__bf16*>(dst + 16) = mid;
            *reinterpret_cast<__bf16*>(dst + 32) = (__bf16)(r1 - (float)mid);
        }
    }
}
// value of the lane's output column from an accumulator (every 16-lane group holds the same 16 columns)
template <typename WT>
__device__ __forceinline__ float cl_out(const f32x4& a) {
    if constexpr (!WT::MFMA) return a.x;
    else if constexpr (WT::HALF) return fmaf(a.y, 1.f / DEN_F16_LO_SCALE, a.x);
    else return a.x + (a.y + a.z);
}

// ---- consume one unit: load i of the unit is load J0 + i of the stage; stage load j = (k-block j / TPW, tile j % TPW)
template <typename WT, int TPW, int J0, int KBOFF>
__device__ __forceinline__ void cl_consume(const u32x4 (&slot)[ClW<WT>::UL], const ClX& x, f32x4 (&acc)[TPW]) {
    typedef ClW<WT> W;
    uint4 a4 = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int i = 0; i < W::UL; ++i) {
        constexpr int dummy = 0; (void)dummy;
        const int j = J0 + i, kb = j / TPW - KBOFF, t = j % TPW;
        if constexpr (WT::MFMA) {
            if (i == 0 || t == 0) a4 = *reinterpret_cast<const uint4*>(x.base + x.foff + kb * x.kbs);
            if constexpr (WT::HALF) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, a4), __builtin_bit_cast(h16x8, slot[i]), acc[t], 0, 0, 0);
            else acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a4), __builtin_bit_cast(bf16x8, slot[i]), acc[t], 0, 0, 0);
        } else {
            if (i == 0 || t == 0) a4 = *reinterpret_cast<const uint4*>(x.base + x.foff + kb * x.kbs);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a4.x), __uint_as_float(slot[i].x), acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a4.y), __uint_as_float(slot[i].y), acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a4.z), __uint_as_float(slot[i].z), acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a4.w), __uint_as_float(slot[i].w), acc[t], 0, 0, 0);
        }
    }
}
template <int TPW>
__device__ __forceinline__ void cl_pin(f32x4 (&acc)[TPW]) {   // program-order pin (see Acc::pin)
#pragma unroll
    for (int t = 0; t < TPW; ++t) asm volatile("" : "+v"(acc[t]) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <int TPW>
__device__ __forceinline__ void cl_zero(f32x4 (&acc)[TPW]) {
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
}
// NUN consecutive units of one stage from unit U0 on: consume, then re-fill the freed slot with unit U + RU -- inline for the
// first NINL units of the call only.  Stages that end in an exchange (A, C) leave the re-fills of their LAST (up to RU) units
// to cl_refills, after the exchange: loads queued in this CU's vector-memory path right before the publish / poll are latency
// on the exchange (the hand-off's price sits in the consumer CU's own memory queue, MI355X_MICROARCH.md handoff-1to1).  Units
// further than RU from the end of such a stage must re-fill inline: their targets are consumed inside the same stage.
template <typename WT, int C, bool Q, int U0, int NUN, int TPW, int UI0, int KBOFF, int NINL, int... Is>
__device__ __forceinline__ void cl_units(ClRing<WT>& ring, const ClX& x, f32x4 (&acc)[TPW], bool active, int wave, unsigned voff,
                                         __amdgpu_buffer_rsrc_t rsrc, unsigned bc, unsigned bn, bool sc, bool sn, std::integer_sequence<int, Is...>) {
    typedef ClW<WT> W;
    (([&] {
        constexpr int U = U0 + Is;
        if (active) { cl_consume<WT, TPW, (UI0 + Is) * W::UL, KBOFF>(ring.r[U % W::RU], x, acc); }
        cl_pin<TPW>(acc);
        if constexpr (Is < NINL) {
            cl_issue<WT, C, Q, U + W::RU>(ring, wave, voff, rsrc, bc, bn, sc, sn);
            cl_pin<TPW>(acc);
        }
    }()), ...);
}
template <typename WT, int C, bool Q, int U0, int... Is>
__device__ __forceinline__ void cl_refills(ClRing<WT>& ring, int wave, unsigned voff, __amdgpu_buffer_rsrc_t rsrc, unsigned bc, unsigned bn,
                                           bool sc, bool sn, std::integer_sequence<int, Is...>) {
    (cl_issue<WT, C, Q, U0 + Is + ClW<WT>::RU>(ring, wave, voff, rsrc, bc, bn, sc, sn), ...);
    __builtin_amdgcn_sched_barrier(0);
}
constexpr int cl_clamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ---- the windowed load schedule (16-bit images, C = 8, one condition token).  Issuing a 1-KiB wave-load costs the CU's address path 16 cycles
// and the issuing wave waits its turn, so re-fills requested inline are exposed time: at a phase start every wave stands in the queue (phase B:
// 2.4 k of its 4 k cycles), before a publish the stores stand behind them.  But while the epilogue wave is in an exchange or in its serial
// vector algebra the other seven waves have nothing to do.  So they request ONE unit in each such window (56 KiB = 900 cycles of the path):
//   W?A  right after the epilogue wave has PUBLISHED (its stores go first), while it sleeps DCL_POLL_SLEEP x 64 cycles before its first poll:
//        the peers' granules are not due earlier than that, and the poll then finds the queue drained instead of standing behind the unit;
//   W?B  after its sweep, during the vector algebra;   W3  during the ffn epilogue
// -- released through words in LDS, accessed through the LDS address space (through a generic pointer the compiler emits flat loads waited for
// with vmcnt(0): seven waves spinning on those sit in the very queue the polls need).  The epilogue wave requests its eighth of a window's
// unit BEFORE it releases the others (behind them it would wait out their backlog).  The rest (next layer's A1, B, C) goes out in phase F.
//   units A0 A1 | B | C | D | E | F0 F1 = 0..7:  W1A D | W1B E | W2A F0 | W2B F1 | W3 next A0 | phase F: next A1 | inline in F: next B, C
#ifndef DCL_POLL_SLEEP
#define DCL_POLL_SLEEP 12
#endif
template <int... Us> using ClSeq = std::integer_sequence<int, Us...>;
template <int C, bool MF, bool Q> struct ClSched { static constexpr bool WIN = false; static constexpr int PRO = 0; typedef ClSeq<> W1A, W1B, W2A, W2B, WXA, WXB, W3, PF, AF; };
#ifndef DCL_SCHED
#define DCL_SCHED 9
#endif
#ifndef DCL_SCHED_C4
#define DCL_SCHED_C4 1     // ... and for clusters of four CUs
#endif
#ifndef DCL_SCHED_Q
#define DCL_SCHED_Q 1      // the windowed schedule also for two condition tokens (C = 8, 16-bit images)
#endif
#if DCL_SCHED == 8      // "early": every unit one window ahead of the inline scheme; the next layer's B and C still go out inline in phase F (PRO: units in flight at a layer's start)
template <> struct ClSched<8, true, false> { static constexpr bool WIN = true; static constexpr int PRO = 4; typedef ClSeq<> WXA, WXB; typedef ClSeq<4> W1A; typedef ClSeq<5> W1B; typedef ClSeq<6> W2A; typedef ClSeq<7> W2B;
                                             typedef ClSeq<8> W3; typedef ClSeq<9> PF; typedef ClSeq<10, 11> AF; };
#elif DCL_SCHED == 9    // "just in time": a unit goes out in the window before the phase that consumes it; only the next layer's A is requested inline (after F)
template <> struct ClSched<8, true, false> { static constexpr bool WIN = true; static constexpr int PRO = 2; typedef ClSeq<> WXA, WXB; typedef ClSeq<2, 3> W1A; typedef ClSeq<4> W1B; typedef ClSeq<5> W2A; typedef ClSeq<6> W2B;
                                             typedef ClSeq<7> W3; typedef ClSeq<> PF; typedef ClSeq<8, 9> AF; };
#elif DCL_SCHED == 10   // as 9, with the second exchange window taking two units as well, so that only the skip layers' A1 is left inline
template <> struct ClSched<8, true, false> { static constexpr bool WIN = true; static constexpr int PRO = 2; typedef ClSeq<> WXA, WXB; typedef ClSeq<2, 3> W1A; typedef ClSeq<4> W1B; typedef ClSeq<5, 6> W2A; typedef ClSeq<7> W2B;
                                             typedef ClSeq<8> W3; typedef ClSeq<> PF; typedef ClSeq<9> AF; };
#endif
// Two condition tokens (Q: the ca_block keeps its query G and proj_out H stages and a third exchange X3), just in time as schedule 9:
//   units A 0 | AS 1 | B 2 | C 3 | G 4 | H 5 6 | D 7 | E 8 | F 9 10 | (11 phantom);  ring slot = U % 4
//   W1A B C | W1B G | W2A H0 | W2B H1 | WXA (X3 published) D | WXB (X3 swept) E | W3 (ffn epilogue) F0 F1 | after F: next A, AS
template <> struct ClSched<8, true, true> { static constexpr bool WIN = DCL_SCHED_Q != 0; static constexpr int PRO = 2; typedef ClSeq<2, 3> W1A; typedef ClSeq<4> W1B; typedef ClSeq<5> W2A; typedef ClSeq<6> W2B;
                                            typedef ClSeq<7> WXA; typedef ClSeq<8> WXB; typedef ClSeq<9, 10> W3; typedef ClSeq<> PF; typedef ClSeq<12, 13> AF; };
// Four CUs per sample (16-bit images, 32 < B <= 64): units A 0 1 | AS 2 3 | B 4 5 | C 6 7 | D 8 | E 9 | F 10 11 (the table k_den_cluster_ms uses);
// with two condition tokens: ... | G 8 | H 9 10 | D 11 | E 12 | F 13 14 | (15 phantom)
#if DCL_SCHED_C4
template <> struct ClSched<4, true, false> { static constexpr bool WIN = true; static constexpr int PRO = 4; typedef ClSeq<4, 5> W1A; typedef ClSeq<6, 7> W1B; typedef ClSeq<8, 9> W2A; typedef ClSeq<10, 11> W2B;
                                             typedef ClSeq<> WXA, WXB; typedef ClSeq<12, 13> W3; typedef ClSeq<> PF; typedef ClSeq<14, 15> AF; };
template <> struct ClSched<4, true, true> { static constexpr bool WIN = true; static constexpr int PRO = 4; typedef ClSeq<4, 5> W1A; typedef ClSeq<6, 7> W1B; typedef ClSeq<8, 9> W2A; typedef ClSeq<10> W2B;
                                            typedef ClSeq<11> WXA; typedef ClSeq<12> WXB; typedef ClSeq<13, 14> W3; typedef ClSeq<> PF; typedef ClSeq<16, 17, 18, 19> AF; };
#endif
template <typename WT, int C, bool Q, int... Us>
__device__ __forceinline__ void cl_issue_seq(ClRing<WT>& ring, int wave, unsigned voff, __amdgpu_buffer_rsrc_t rsrc, unsigned bc, unsigned bn,
                                             bool sc, bool sn, ClSeq<Us...>) {
    (cl_issue<WT, C, Q, Us>(ring, wave, voff, rsrc, bc, bn, sc, sn), ...);
    __builtin_amdgcn_sched_barrier(0);
}
typedef __attribute__((address_space(3))) volatile int cl_lds_flag;
__device__ __forceinline__ void cl_flag_set(int* f, unsigned v) { *(cl_lds_flag*)f = (int)v; }
__device__ __forceinline__ void cl_flag_wait(const int* f, unsigned v) {      // (set by this workgroup's epilogue wave: always arrives)
    while (*(const cl_lds_flag*)f != (int)v) __builtin_amdgcn_s_sleep(1);
}

// ---- granules
__device__ __forceinline__ void cl_store_granule(unsigned long long* g, unsigned epoch, float v, bool local) {
    const unsigned long long x = ((unsigned long long)epoch << 32) | __float_as_uint(v);
    if (local) *(volatile dcl_gu64*)g = x;                                                       // stays in this XCD's L2
    else __hip_atomic_store((dcl_gu64*)g, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);         // sc1: write-through
}
__device__ __forceinline__ unsigned cl_tags_ok(const u32x4& a, unsigned epoch) { return (unsigned)(a.y == epoch) & (unsigned)(a.w == epoch); }   // (no short-circuit branches)

// The next layer's small operands -> the other half of the staging buffer, through registers: only what THIS CU reads (its slices of the
// in_proj / skip / linear1 biases, the LayerNorm and bias vectors of the replicated stages, the time token's K | V' and the ffn AdaLN rows, the
// condition token's K | V', the tabulated ca term): 22-23 pieces of at most 1 KiB = 17.9 KB at C = 8, three per wave: one requested at the top
// of each of the phases B, D and F and stored to LDS at its end, so that the load's latency passes behind the phase.  (As 36 KB of LDS-DMA -- stage_dma, what k_den_sample
// does -- the same operands cost this kernel 1.9 k cycles per layer: LDS-DMA moves ~40 GB/s per CU, and here nothing overlaps it.)
template <int C, bool Q>
struct ClStage {
    float4 r; int dst, n4;        // ONE piece per wave and phase (!Q: k = 0 in phase B, 1 in D, 2 in F; Q: B, C, H, D, F): 4 registers live across a phase
    static constexpr int NL1 = ClG<C, Q>::NB > 256 ? ClG<C, Q>::NB / 256 : 1;
    __device__ __forceinline__ void load(int k, int wave, int lane, const float* __restrict__ vpg, const DenLayerOff* __restrict__ L,
                                         const float* __restrict__ tt_row, int l, const SeemeSampleArgs& A, int b, int c, int ca_r, int ca_R) {
        typedef ClG<C, Q> G;
        const float* vb = vpg + L->skip_b;
        const int o_in = (int)(L->in_b - L->skip_b);
        const int i = wave + 8 * k;
        const float* src = vb; int d = 0, n = 0;
        if (i == 0) { d = c * G::S; src = vb + d; n = G::S / 4; }
        else if (i <= 3) { d = o_in + (i - 1) * 256 + c * G::S; src = vb + d; n = G::S / 4; }
        else if (i <= 13) {
            const int64_t f = i == 4 ? L->n1w : i == 5 ? L->n1b : i == 6 ? L->l2b : i == 7 ? L->n2w : i == 8 ? L->n2b : i == 9 ? L->f1b
                            : i == 10 ? L->f2b : i == 11 ? L->fsnw : i == 12 ? L->fsnb : L->fo_b;
            d = (int)(f - L->skip_b); src = vb + d; n = i == 9 ? FF_D / 4 : 64;
        }
        else if (i <= 15) { src = tt_row + l * 512 + (i - 14) * 256; d = VP_LAYER + (i - 14) * 256; n = 64; }                 // time token K | V'
        else if (i <= 17) { src = tt_row + 2560 + l * 1024 + 512 + (i - 16) * 256; d = VP_LAYER + 1024 + (i - 16) * 256; n = 64; }   // ffn AdaLN scale | shift
        else if (i < 18 + NL1) { d = (int)(L->l1b - L->skip_b) + c * G::NB + (i - 18) * 256; src = vb + d; n = G::NB >= 256 ? 64 : G::NB / 4; }
        else if constexpr (!Q) {
            const int j = i - 18 - NL1;
            if (j <= 1) { src = A.ctab + (size_t)b * SEEME_CROW + l * 512 + j * 256; d = VP_LAYER + STG_TT + j * 256; n = 64; }             // sa K | V' of the condition token
            else if (j == 2) { src = A.catab + (((size_t)b * ca_R + ca_r) * SEEME_DEN_NL + l) * 256; d = VP_LAYER + STG_TT + 1024; n = 64; }   // its tabulated ca term
        } else {
            const int j = i - 18 - NL1;
            if (j <= 1) { src = tt_row + 2560 + l * 1024 + j * 256; d = VP_LAYER + 512 + j * 256; n = 64; }                                 // ca AdaLN scale | shift
            else if (j <= 6) {
                const int64_t f = j == 2 ? L->cnw : j == 3 ? L->cnb : j == 4 ? L->csnw : j == 5 ? L->csnb : L->cao_b;
                d = (int)(f - L->skip_b); src = vb + d; n = 64;
            }
            else if (j == 7) { d = (int)(L->caq_b - L->skip_b) + c * G::S; src = vb + d; n = G::S / 4; }
            else if (j < 8 + 4 * A.N) {                // condition tokens: sa K | sa V' | ca key | ca value, 256 floats each
                const int t = (j - 8) >> 2, q = (j - 8) & 3;
                src = A.ctab + ((size_t)b * A.N + t) * SEEME_CROW + (q < 2 ? l * 512 + q * 256 : 2560 + l * 512 + (q - 2) * 256);
                d = VP_LAYER + STG_TT + t * 1024 + q * 256; n = 64;
            }
        }
        dst = d; n4 = n;
        r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane < n) r = *reinterpret_cast<const float4*>(src + 4 * lane);
    }
    __device__ __forceinline__ void store(int lane, float* __restrict__ stg) const {
        if (lane < n4) *reinterpret_cast<float4*>(stg + dst + 4 * lane) = r;
    }
};
// pieces: !Q 21 + NL1 <= 24 (three phases x 8 waves); Q 26 + NL1 + 4 N <= 40 (five phases) for N <= 2 (C = 2: N = 2 gives 36)
#define DCL_MAX_N 2

template <typename WT, int C, bool Q>
__global__ __launch_bounds__(DEN_THREADS) void k_den_cluster(const ClArgs ka) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef ClW<WT> W; typedef ClG<C, Q> G;
    // re-fills left until after the exchange: the last RU units of stage A (x half + skip half) and of stage C
    constexpr int A_DEF0 = cl_clamp(2 * G::TA - W::RU, 0, 2 * G::TA);          // first deferred unit of the 2 TA units of stage A
    constexpr int A_INL0 = cl_clamp(A_DEF0, 0, G::TA), A_INL1 = cl_clamp(A_DEF0 - G::TA, 0, G::TA), C_INL = cl_clamp(G::TB - W::RU, 0, G::TB);
    typedef ClSched<C, WT::MFMA, Q> SCH;
    constexpr bool WIN = SCH::WIN;       // (then stages A .. E request nothing inline: the windows do)
    const SeemeSampleArgs& A = ka.s;
    const DenLayout* __restrict__ lay = &ka.lay;
    const float* __restrict__ vp = ka.vp;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // which sample, which part of it: placement 0 = the cluster's workgroups have equal blockIdx % 8 (one XCD under
    // round-robin dispatch), placement 1 = consecutive blockIdx (C different XCDs).  Speed only.
    int b, c;
    if (ka.placement == 0) { const int x = blockIdx.x & 7, j = blockIdx.x >> 3; b = x * (ka.clusters / 8) + j / C; c = j % C; }
    else { b = blockIdx.x / C; c = blockIdx.x % C; }
    if (b >= A.B) return;                                      // (whole clusters: every member sees the same b)
    const bool epi = wave == 0;
    const int N = Q ? A.N : 1;
    const int stg_sz = VP_LAYER + STG_TT + N * 1024 + (Q ? 0 : 256);
    const int ca_R = A.trow_per_sample ? 1 : A.steps;

    float* CONSTV = smem;                        // [768]  query_pos.pe[0], encoder.norm.{weight,bias}
    float* KEEP = CONSTV + 768;                  // [256]  the latent (epilogue wave)
    float* STG = KEEP + 256;                     // [2][stg_sz] per-layer operands, double-buffered (stage_dma)
    float* XA = STG + 2 * stg_sz;                // [512]  GEMV input: layer input x / x2   (16-bit: fragments of 256 k = 2 KiB)
    float* XB = XA + 512;                        // [512]  GEMV input: x1 / u
    float* XH = XB + 512;                        // [2 NB] GEMV input: this CU's hidden units / ffn hidden (fragments of NB k)
    float* SKF = XH + 2 * (G::NB > 128 ? G::NB : 128);   // [2][512] outputs of layers 0, 1 (skip inputs of layers 4, 3)
    float* QS = SKF + 1024;                      // [S] q, [S] k of this CU's dims
    float* KS = QS + G::S;
    float* PART = KS + G::S;                     // [256]
    float* RES = PART + 256;                     // [256]  the fp32 residual stream
    int* FLG = reinterpret_cast<int*>(RES + 256);   // [4]
    constexpr int XZERO = 512 + 512 + 2 * (G::NB > 128 ? G::NB : 128) + 1024;

    const __amdgpu_buffer_rsrc_t wg = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ka.wgc), 0, (int)ka.wgc_bytes, 0x00020000);
    unsigned long long* const xg = ka.xg + (size_t)b * G::G_TOTAL;
    const __amdgpu_buffer_rsrc_t xr_ = __builtin_amdgcn_make_buffer_rsrc(xg, 0, G::G_TOTAL * 8, 0x00020000);
    const unsigned voff = (unsigned)wave * (unsigned)(W::UL * 1024) + (unsigned)lane * 16u;
    const int col = lane & 15;
    const ClX xa = cl_xin<WT>(XA, lane), xb = cl_xin<WT>(XB, lane), xh = cl_xin<WT>(XH, lane);
    const float sa_scale = 1.f / 16.f;           // one head of 256 dims
    if (epi) __builtin_amdgcn_s_setprio(3);

    typedef const __attribute__((address_space(4))) int32_t* CI32;
    typedef const __attribute__((address_space(4))) float* CF32;
    const CI32 trow_c = (CI32)(uintptr_t)A.trow;
    const CF32 coef_c = (CF32)(uintptr_t)A.coef;
    int row = A.trow_per_sample ? trow_c[b] : trow_c[0];

    // ---- prologue: constants, layer 0 operands, zeroed fragment buffers, first input, first RU units, the XCC census
    float4 xr = ld4(A.latents + (size_t)b * 256 + 4 * lane);
    for (int i = tid; i < 192; i += DEN_THREADS)
        st4(CONSTV + 4 * i, i < 64 ? ld4(vp + lay->pe0 + 4 * i) : (i < 128 ? ld4(vp + lay->fnw + 4 * (i - 64)) : ld4(vp + lay->fnb + 4 * (i - 128))));
    stage_dma<1, Q, 0, 0>(wave, lane, STG, vp, &lay->L[0], A.ttab + (size_t)row * SEEME_TROW, 0, A, b, N, 0, ca_R);
    for (int i = tid; i < XZERO / 4; i += DEN_THREADS) st4(XA + 4 * i, make_float4(0.f, 0.f, 0.f, 0.f));
    if (tid >= 1 && tid < 4) FLG[tid] = 0;                   // window flags (epochs are never 0)
    bool dead = false, local = false;
    if (epi) {
        // first exchange, always write-through: the XCC id of every workgroup of the cluster (HW_REG_XCC_ID, bits 3:0)
        const unsigned my_xcc = __builtin_amdgcn_s_getreg(0x1814) & 15u;
        if (lane == 0) cl_store_granule(xg + G::G_BOOT + c, DCL_MAGIC_EPOCH, __uint_as_float(my_xcc), false);
        unsigned spins = 0, idv = 0;
        for (;;) {
            const unsigned long long x = __hip_atomic_load((dcl_gu64*)(xg + G::G_BOOT + (lane % C)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            idv = (unsigned)x;
            if (__all((unsigned)(x >> 32) == DCL_MAGIC_EPOCH)) break;
            if (++spins > DCL_SPIN_LIMIT) { dead = true; if (lane == 0) atomicOr(ka.hdr, 1u); break; }
            __builtin_amdgcn_s_sleep(2);
        }
        local = !dead && __all(idv == my_xcc) && !(ka.flags & 1);
        if (lane == 0) { FLG[0] = local ? 1 : 0; if (local && c == 0) atomicAdd(ka.hdr + 1, 1u); }
    }
    wait_vmcnt0();
    __syncthreads();
    local = FLG[0] != 0;
    if (epi) {
        st4(KEEP + 4 * lane, xr);
        xr = f4_add(xr, ld4(CONSTV + 4 * lane));                // sample + query_pos (mld_denoiser.py:210)
        put_x<WT, 1>(XA, 0, 0, lane, xr);
        st4(RES + 4 * lane, xr);
    }
    ClRing<WT> ring;
    {
        const unsigned b0 = (unsigned)((0 * C + c) * G::NU) * W::UNIT_BYTES;
        cl_issue<WT, C, Q, 0>(ring, wave, voff, wg, b0, b0, false, false);
        cl_issue<WT, C, Q, 1>(ring, wave, voff, wg, b0, b0, false, false);
        if constexpr (W::RU == 4 && (!WIN || SCH::PRO == 4)) { cl_issue<WT, C, Q, 2>(ring, wave, voff, wg, b0, b0, false, false); cl_issue<WT, C, Q, 3>(ring, wave, voff, wg, b0, b0, false, false); }
    }
    __syncthreads();
    int cur = 0;

#pragma unroll 1
    for (int step = 0; step < A.steps; ++step) {
        if (step == 2) DEN_DBG(1);
        if (step == 3) DEN_DBG(2);
        const int step_next = step + 1 < A.steps ? step + 1 : step;
        const int row_next = A.trow_per_sample ? row : trow_c[step_next];
#pragma unroll 1
        for (int l = 0; l < SEEME_DEN_NL; ++l) {
            const DenLayerOff* __restrict__ L = &lay->L[l];
            const int ln = (l + 1 < SEEME_DEN_NL) ? l + 1 : 0;
            const DenLayerOff* __restrict__ Ln = &lay->L[ln];
            const bool skip = l >= 3, nskip = ln >= 3;
            const unsigned bc = (unsigned)((l * C + c) * G::NU) * W::UNIT_BYTES, bn = (unsigned)((ln * C + c) * G::NU) * W::UNIT_BYTES;
            const float* VP = STG + cur * stg_sz;
            const float* TTS = VP + VP_LAYER;
            const float* CT = TTS + STG_TT;
            const float* CA_ADD = CT + N * 1024;
            const float* v_skip_b = VP;
            const float* v_in_b = VP + (L->in_b - L->skip_b);
            const float* v_n1w = VP + (L->n1w - L->skip_b), *v_n1b = VP + (L->n1b - L->skip_b);
            const float* v_l1b = VP + (L->l1b - L->skip_b), *v_l2b = VP + (L->l2b - L->skip_b);
            const float* v_n2w = VP + (L->n2w - L->skip_b), *v_n2b = VP + (L->n2b - L->skip_b);
            const float* v_f1b = VP + (L->f1b - L->skip_b), *v_f2b = VP + (L->f2b - L->skip_b);
            const float* v_fsnw = VP + (L->fsnw - L->skip_b), *v_fsnb = VP + (L->fsnb - L->skip_b);
            const float* v_fo_b = VP + (L->fo_b - L->skip_b);
            const float* v_cnw = VP + (L->cnw - L->skip_b), *v_cnb = VP + (L->cnb - L->skip_b), *v_caq_b = VP + (L->caq_b - L->skip_b);
            const float* v_csnw = VP + (L->csnw - L->skip_b), *v_csnb = VP + (L->csnb - L->skip_b), *v_cao_b = VP + (L->cao_b - L->skip_b);
            (void)v_cnw; (void)v_cnb; (void)v_caq_b; (void)v_csnw; (void)v_csnb; (void)v_cao_b;
            const bool ywave = wave >= 6;
            const unsigned e1 = 1u + (unsigned)G::EPL * (unsigned)(step * SEEME_DEN_NL + l), e2 = e1 + 1u, e3 = e1 + 2u;   // epochs of this layer's exchanges
            (void)e3;

            // ================= A: in_proj' (+ folded skip linear), column-split by dims =================
            {
                f32x4 acc[G::TA];
                cl_zero<G::TA>(acc);
                const bool act = skip || !ywave;
                cl_units<WT, C, Q, G::U_A, G::TA, G::TA, 0, 0, A_INL0>(ring, xa, acc, act, wave, voff, wg, bc, bn, skip, nskip, std::make_integer_sequence<int, G::TA>{});
                {   // second k-half: the skip input (output of layer 1 for layer 3, of layer 0 for layer 4: xs.pop(), cross_attention.py:77-79)
                    const ClX xs = cl_xin<WT>(SKF + (l == 3 ? 512 : 0), lane);
                    cl_units<WT, C, Q, G::U_AS, G::TA, G::TA, G::TA, W::UL, A_INL1>(ring, xs, acc, skip, wave, voff, wg, bc, bn, skip, nskip, std::make_integer_sequence<int, G::TA>{});
                }
                if (act && lane < 16) {
#pragma unroll
                    for (int t = 0; t < G::TA; ++t) {
                        const int T = wave * G::TA + t, part = T / (G::S / 16), d = (T % (G::S / 16)) * 16 + col, D = c * G::S + d;
                        const float val = cl_out<WT>(acc[t]);
                        const unsigned epoch = e1;
                        if (part == 0) QS[d] = val + v_in_b[D];
                        else if (part == 1) KS[d] = val + v_in_b[256 + D];
                        else if (part == 2) cl_store_granule(xg + G::G_X1 + c * G::X1_G + d, epoch, val + v_in_b[512 + D], local);
                        else cl_store_granule(xg + G::G_X1 + c * G::X1_G + G::S + d, epoch, val + v_skip_b[D], local);
                    }
                }
            }
            __syncthreads(); DEN_DBG(0);
            if (epi) {
                // partial scores over this CU's dims: token 0 itself, the condition token(s), the time token (last; mdiff_transformer.py:295)
                const unsigned epoch = e1;
                constexpr int NT = Q ? DCL_MAX_N + 2 : 3;          // score slots: [0] self, [1 .. N] condition, [N + 1] time
                float ps[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) ps[j] = 0.f;
#pragma unroll
                for (int dd = 0; dd < G::S; dd += 64) {
                    const int d = dd + lane;
                    if (d < G::S) {
                        const float q = QS[d];
                        ps[0] = fmaf(q, KS[d], ps[0]);
#pragma unroll
                        for (int n = 0; n < NT - 2; ++n) if (n < N) ps[1 + n] = fmaf(q, CT[n * 1024 + c * G::S + d], ps[1 + n]);
                        ps[NT - 1] = fmaf(q, TTS[c * G::S + d], ps[NT - 1]);
                    }
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) ps[j] = wave_sum(ps[j]) * sa_scale;
                {
                    float pv = 0.f;
#pragma unroll
                    for (int j = 0; j < NT; ++j) pv = lane == j ? ps[j] : pv;
                    if (lane < G::SC) cl_store_granule(xg + G::G_X1 + c * G::X1_G + 2 * G::S + lane, epoch, pv, local);
                }
                if constexpr (WIN) {        // published: this wave's share of W1A, then the others'; the first poll waits until the unit has drained
                    __builtin_amdgcn_sched_barrier(0);
                    cl_issue_seq<WT, C, Q>(ring, wave, voff, wg, bc, bn, skip, nskip, typename SCH::W1A{});
                    if (lane == 0) cl_flag_set(FLG + 1, e1);
                    __builtin_amdgcn_s_sleep(DCL_POLL_SLEEP);
                }
                // ---- X1: gather v' (and y), all-reduce the scores
                const int pub = (4 * lane) / G::S, off = (4 * lane) % G::S;
                const unsigned o_v = (unsigned)((G::G_X1 + pub * G::X1_G + off) * 8);
                const unsigned o_s = (unsigned)((G::G_X1 + (lane % C) * G::X1_G + 2 * G::S) * 8);
                u32x4 gv0, gv1, gy0, gy1, gs[G::SC / 2];
                gy0 = gy1 = u32x4{0u, epoch, 0u, epoch};
                unsigned spins = 0;
                for (;;) {
                    gv0 = __builtin_amdgcn_raw_buffer_load_b128(xr_, o_v, 0, 16);
                    gv1 = __builtin_amdgcn_raw_buffer_load_b128(xr_, o_v + 16, 0, 16);
                    if (skip) {
                        gy0 = __builtin_amdgcn_raw_buffer_load_b128(xr_, o_v + G::S * 8, 0, 16);
                        gy1 = __builtin_amdgcn_raw_buffer_load_b128(xr_, o_v + G::S * 8 + 16, 0, 16);
                    }
#pragma unroll
                    for (int j = 0; j < G::SC / 2; ++j) gs[j] = __builtin_amdgcn_raw_buffer_load_b128(xr_, o_s + 16 * j, 0, 16);
                    unsigned ok = cl_tags_ok(gv0, epoch) & cl_tags_ok(gv1, epoch) & cl_tags_ok(gy0, epoch) & cl_tags_ok(gy1, epoch);
#pragma unroll
                    for (int j = 0; j < G::SC / 2; ++j) ok &= cl_tags_ok(gs[j], epoch);
                    if (__all(ok != 0u) || dead) break;
                    if (++spins > DCL_SPIN_LIMIT) { dead = true; if (lane == 0) atomicOr(ka.hdr, 2u); break; }
                }
                if constexpr (WIN) {
                    cl_issue_seq<WT, C, Q>(ring, wave, voff, wg, bc, bn, skip, nskip, typename SCH::W1B{});
                    if (lane == 0) cl_flag_set(FLG + 2, e1);
                }
                DEN_DBG(0);
                float sc[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    sc[j] = __uint_as_float((j & 1) ? gs[j >> 1].z : gs[j >> 1].x);
                    // sum over the C publishers (aligned groups of C lanes hold one each): butterfly, the same order on every CU
                    if (C >= 2) sc[j] += dpp_f(sc[j], 0);
                    if (C >= 4) sc[j] += dpp_f(sc[j], 1);
                    if (C >= 8) sc[j] += dpp_f(sc[j], 2);
                }
                float mx = fmaxf(sc[0], sc[NT - 1]);
#pragma unroll
                for (int n = 0; n < NT - 2; ++n) if (n < N) mx = fmaxf(mx, sc[1 + n]);
                float esum = 0.f;
#pragma unroll
                for (int j = 0; j < NT; ++j) { sc[j] = (j == 0 || j == NT - 1 || j - 1 < N) ? fast_exp(sc[j] - mx) : 0.f; esum += sc[j]; }
                const float inv = fast_rcp(esum);
                const float4 vv = make_float4(__uint_as_float(gv0.x), __uint_as_float(gv0.z), __uint_as_float(gv1.x), __uint_as_float(gv1.z));
                float4 att = f4_scale(vv, sc[0] * inv);
#pragma unroll
                for (int n = 0; n < NT - 2; ++n) if (n < N) att = f4_fma(sc[1 + n] * inv, ld4(CT + n * 1024 + 256 + 4 * lane), att);
                att = f4_fma(sc[NT - 1] * inv, ld4(TTS + 256 + 4 * lane), att);
                if (skip) xr = make_float4(__uint_as_float(gy0.x), __uint_as_float(gy0.z), __uint_as_float(gy1.x), __uint_as_float(gy1.z));
                xr = wave_ln(f4_add(xr, att), v_n1w, v_n1b, lane);        // the "values" carry out_proj: residual + norm1
                put_x<WT, 1>(XB, 0, 0, lane, xr);
            } else if constexpr (WIN) {
                cl_flag_wait(FLG + 1, e1);
                cl_issue_seq<WT, C, Q>(ring, wave, voff, wg, bc, bn, skip, nskip, typename SCH::W1A{});
                cl_flag_wait(FLG + 2, e1);
                cl_issue_seq<WT, C, Q>(ring, wave, voff, wg, bc, bn, skip, nskip, typename SCH::W1B{});
            }
            __syncthreads(); DEN_DBG(0);

            // ================= B: linear1 + ReLU, column-split =================
            // (a piece of the next layer's small operands is requested here and stored to the other half of the staging buffer at the end
            //  of the phase -- BEFORE the deferred re-fills: vmcnt is in order, and the store must not wait for them)
            ClStage<C, Q> nxt;
            const float* const tt_next = A.ttab + (size_t)(ln == 0 ? row_next : row) * SEEME_TROW;
            const int ca_next = A.trow_per_sample ? 0 : (ln == 0 ? step_next : step);
            nxt.load(0, wave, lane, vp, Ln, tt_next, ln, A, b, c, ca_next, ca_R);
            if constexpr (!WIN) cl_refills<WT, C, Q, G::U_A + A_DEF0>(ring, wave, voff, wg, bc, bn, skip, nskip, std::make_integer_sequence<int, 2 * G::TA - A_DEF0>{});   // slots of stage A
            {
                f32x4 acc[G::TB];
                cl_zero<G::TB>(acc);
                cl_units<WT, C, Q, G::U_B, G::TB, G::TB, 0, 0, (WIN ? 0 : G::TB)>(ring, xb, acc, true, wave, voff, wg, bc, bn, skip, nskip, std::make_integer_sequence<int, G::TB>{});
                if (lane < 16) {
#pragma unroll
                    for (int t = 0; t < G::TB; ++t) {
                        const int j = (wave * G::TB + t) * 16 + col;
                        cl_put1<WT>(XH, j, fmaxf(cl_out<WT>(acc[t]) + v_l1b[c * G::NB + j], 0.f));
                    }
                }
            }
            nxt.store(lane, STG + (cur ^ 1) * stg_sz);
            __syncthreads(); DEN_DBG(0);

            // ================= C: linear2, row-split -> X2 =================
            if constexpr (Q) nxt.load(1, wave, lane, vp, Ln, tt_next, ln, A, b, c, ca_next, ca_R);
            {
                f32x4 acc[2];
                cl_zero<2>(acc);
                cl_units<WT, C, Q, G::U_C, G::TB, 2, 0, 0, C_INL>(ring, xh, acc, true, wave, voff, wg, bc, bn, skip, nskip, std::make_integer_sequence<int, G::TB>{});
                if (lane < 32) {
                    const float val = (lane & 16) ? cl_out<WT>(acc[1]) : cl_out<WT>(acc[0]);
                    cl_store_granule(xg + G::G_X2 + c * 256 + (2 * wave + (lane >> 4)) * 16 + col, e2, val, local);
                }
            }
            if constexpr (Q) nxt.store(lane, STG + (cur ^ 1) * stg_sz);
            if (epi) {
                const unsigned epoch = e2;
                if constexpr (WIN) {
                    __builtin_amdgcn_sched_barrier(0);
                    cl_issue_seq<WT, C, Q>(ring, wave, voff, wg, bc, bn, skip, nskip, typename SCH::W2A{});
                    if (lane == 0) cl_flag_set(FLG + 1, e2);
                    __builtin_amdgcn_s_sleep(DCL_POLL_SLEEP);
                }
                float4 sum;
                unsigned spins = 0;
                for (;;) {
                    u32x4 g[C][2];
#pragma unroll
                    for (int p = 0; p < C; ++p) {
                        const unsigned o = (unsigned)((G::G_X2 + p * 256 + 4 * lane) * 8);
                        g[p][0] = __builtin_amdgcn_raw_buffer_load_b128(xr_, o, 0, 16);
                        g[p][1] = __builtin_amdgcn_raw_buffer_load_b128(xr_, o + 16, 0, 16);
                    }
                    unsigned ok = 1u;
                    sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int p = 0; p < C; ++p) {
                        ok &= cl_tags_ok(g[p][0], epoch) & cl_tags_ok(g[p][1], epoch);
                        sum.x += __uint_as_float(g[p][0].x); sum.y += __uint_as_float(g[p][0].z);
                        sum.z += __uint_as_float(g[p][1].x); sum.w += __uint_as_float(g[p][1].z);
                    }
                    if (__all(ok != 0u) || dead) break;
                    if (++spins > DCL_SPIN_LIMIT) { dead = true; if (lane == 0) atomicOr(ka.hdr, 4u); break; }
                }
                if constexpr (WIN) {
                    cl_issue_seq<WT, C, Q>(ring, wave, voff, wg, bc, bn, skip, nskip, typename SCH::W2B{});
                    if (lane == 0) cl_flag_set(FLG + 2, e2);
                }
                DEN_DBG(0);
                // + bias, residual, norm2
                xr = wave_ln(f4_add(xr, f4_add(sum, ld4(v_l2b + 4 * lane))), v_n2w, v_n2b, lane);
                if constexpr (!Q) {
                    // ONE condition token: x + Stylization(v) does not depend on x (seeme_denoiser_ca_tables)
                    xr = f4_add(xr, ld4(CA_ADD + 4 * lane));
                    put_x<WT, 1>(XA, 0, 0, lane, xr);
                } else {
                    put_x<WT, 1>(XB, 0, 0, lane, wave_ln(xr, v_cnw, v_cnb, lane));       // ca_block.norm -> input of the query (mdiff_transformer.py:229)
                }
                st4(RES + 4 * lane, xr);
            } else if constexpr (WIN) {
                cl_flag_wait(FLG + 1, e2);
                cl_issue_seq<WT, C, Q>(ring, wave, voff, wg, bc, bn, skip, nskip, typename SCH::W2A{});
                cl_flag_wait(FLG + 2, e2);
                cl_issue_seq<WT, C, Q>(ring, wave, voff, wg, bc, bn, skip, nskip, typename SCH::W2B{});
            }
            __syncthreads(); DEN_DBG(0);

            if constexpr (Q) {
                // ================= G: ca_block.query, column-split by dims (waves < S / 16 own one tile each) -> X3 =================
                if constexpr (!WIN) cl_refills<WT, C, Q, G::U_C + C_INL>(ring, wave, voff, wg, bc, bn, skip, nskip, std::make_integer_sequence<int, G::TB - C_INL>{});   // slots of stage C
                {
                    f32x4 acc[1];
                    cl_zero<1>(acc);
                    const bool act = wave < G::S / 16;
                    cl_units<WT, C, Q, G::U_G, 1, 1, 0, 0, 0>(ring, xb, acc, act, wave, voff, wg, bc, bn, skip, nskip, std::make_integer_sequence<int, 1>{});
                    if (act && lane < 16) QS[wave * 16 + col] = cl_out<WT>(acc[0]) + v_caq_b[c * G::S + wave * 16 + col];
                }
                __syncthreads(); DEN_DBG(0);
                if (epi) {
                    // The query softmax runs over all 256 dims (one head), the key softmax over the N tokens per dim (mdiff_transformer.py:231-232).
                    // This CU holds S dims: local maximum m, e = exp(q - m), l = sum e, and per token t_n = sum_d e_d kc_n[d]; the cluster
                    // combines them like a split softmax: M = max m_c, w_c = exp(m_c - M), dot_n = sum_c t_cn w_c / sum_c l_c w_c.
                    const unsigned epoch = e3;
                    float m = -INFINITY;
#pragma unroll
                    for (int dd = 0; dd < G::S; dd += 64) { const int d = dd + lane; if (d < G::S) m = fmaxf(m, QS[d]); }
                    m = wave_max(m);
                    float lsum = 0.f, tn[DCL_MAX_N];
#pragma unroll
                    for (int n = 0; n < DCL_MAX_N; ++n) tn[n] = 0.f;
#pragma unroll
                    for (int dd = 0; dd < G::S; dd += 64) {
                        const int d = dd + lane;
                        if (d < G::S) {
                            const float e = fast_exp(QS[d] - m);
                            lsum += e;
                            float kr[DCL_MAX_N], kmx = -INFINITY, ks = 0.f;
#pragma unroll
                            for (int n = 0; n < DCL_MAX_N; ++n) if (n < N) { kr[n] = CT[n * 1024 + 512 + c * G::S + d]; kmx = fmaxf(kmx, kr[n]); }
#pragma unroll
                            for (int n = 0; n < DCL_MAX_N; ++n) if (n < N) { kr[n] = fast_exp(kr[n] - kmx); ks += kr[n]; }
                            const float rks = fast_rcp(ks);
#pragma unroll
                            for (int n = 0; n < DCL_MAX_N; ++n) if (n < N) tn[n] = fmaf(e, kr[n] * rks, tn[n]);
                        }
                    }
                    lsum = wave_sum(lsum);
#pragma unroll
                    for (int n = 0; n < DCL_MAX_N; ++n) tn[n] = wave_sum(tn[n]);
                    {
                        float pv = lane == 0 ? m : (lane == 1 ? lsum : 0.f);
#pragma unroll
                        for (int n = 0; n < DCL_MAX_N; ++n) pv = lane == 2 + n ? tn[n] : pv;
                        if (lane < 8) cl_store_granule(xg + G::G_X3 + c * 8 + lane, epoch, pv, local);
                    }
                    if constexpr (WIN) {
                        __builtin_amdgcn_sched_barrier(0);
                        cl_issue_seq<WT, C, Q>(ring, wave, voff, wg, bc, bn, skip, nskip, typename SCH::WXA{});
                        if (lane == 0) cl_flag_set(FLG + 1, e3);
                        __builtin_amdgcn_s_sleep(DCL_POLL_SLEEP);
                    }
                    const unsigned o3 = (unsigned)((G::G_X3 + (lane % C) * 8) * 8);
                    u32x4 g3[4];
                    unsigned spins = 0;
                    for (;;) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) g3[j] = __builtin_amdgcn_raw_buffer_load_b128(xr_, o3 + 16 * j, 0, 16);
                        unsigned ok = 1u;
#pragma unroll
                        for (int j = 0; j < 4; ++j) ok &= cl_tags_ok(g3[j], epoch);
                        if (__all(ok != 0u) || dead) break;
                        if (++spins > DCL_SPIN_LIMIT) { dead = true; if (lane == 0) atomicOr(ka.hdr, 8u); break; }
                    }
                    if constexpr (WIN) {
                        cl_issue_seq<WT, C, Q>(ring, wave, voff, wg, bc, bn, skip, nskip, typename SCH::WXB{});
                        if (lane == 0) cl_flag_set(FLG + 2, e3);
                    }
                    DEN_DBG(0);
                    // lane group of C publishers: [m, l, t_0 .. ] of publisher lane % C
                    const float mc = __uint_as_float(g3[0].x), lc = __uint_as_float(g3[0].z);
                    float M = mc;
                    if (C >= 2) M = fmaxf(M, dpp_f(M, 0));
                    if (C >= 4) M = fmaxf(M, dpp_f(M, 1));
                    if (C >= 8) M = fmaxf(M, dpp_f(M, 2));
                    const float wc = fast_exp(mc - M);
                    float Lw = lc * wc;
                    if (C >= 2) Lw += dpp_f(Lw, 0);
                    if (C >= 4) Lw += dpp_f(Lw, 1);
                    if (C >= 8) Lw += dpp_f(Lw, 2);
                    const float rL = fast_rcp(Lw);
                    float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int n = 0; n < DCL_MAX_N; ++n) {
                        if (n < N) {
                            float t = __uint_as_float((n & 1) ? g3[1 + (n >> 1)].z : g3[1 + (n >> 1)].x) * wc;
                            if (C >= 2) t += dpp_f(t, 0);
                            if (C >= 4) t += dpp_f(t, 1);
                            if (C >= 8) t += dpp_f(t, 2);
                            y = f4_fma(t * rL, ld4(CT + n * 1024 + 768 + 4 * lane), y);            // (q k^T) v  (:236-237)
                        }
                    }
                    // StylizationBlock (mdiff_transformer.py:152-163): SiLU(LN(y) (1 + scale) + shift) -> proj_out.out_layers
                    const float4 hh = f4_adaln(wave_ln(y, v_csnw, v_csnb, lane), ld4(TTS + 512 + 4 * lane), ld4(TTS + 768 + 4 * lane));
                    put_x<WT, 1>(XB, 0, 0, lane, f4_silu(hh));
                } else if constexpr (WIN) {
                    cl_flag_wait(FLG + 1, e3);
                    cl_issue_seq<WT, C, Q>(ring, wave, voff, wg, bc, bn, skip, nskip, typename SCH::WXA{});
                    cl_flag_wait(FLG + 2, e3);
                    cl_issue_seq<WT, C, Q>(ring, wave, voff, wg, bc, bn, skip, nskip, typename SCH::WXB{});
                }
                __syncthreads(); DEN_DBG(0);
                // ================= H: ca_block.proj_out.out_layers + residual (replicated) =================
                if constexpr (!WIN) cl_refills<WT, C, Q, G::U_G>(ring, wave, voff, wg, bc, bn, skip, nskip, std::make_integer_sequence<int, 1>{});                    // slot of stage G
                nxt.load(2, wave, lane, vp, Ln, tt_next, ln, A, b, c, ca_next, ca_R);
                {
                    f32x4 acc[2];
                    cl_zero<2>(acc);
                    cl_units<WT, C, Q, G::U_H, 2, 2, 0, 0, (WIN ? 0 : 2)>(ring, xb, acc, true, wave, voff, wg, bc, bn, skip, nskip, std::make_integer_sequence<int, 2>{});
                    if (lane < 32) {
                        const int n = (2 * wave + (lane >> 4)) * 16 + col;
                        const float x3 = RES[n] + ((lane & 16) ? cl_out<WT>(acc[1]) : cl_out<WT>(acc[0])) + v_cao_b[n];
                        RES[n] = x3;
                        cl_put1<WT>(XA, n, x3);
                    }
                }
                nxt.store(lane, STG + (cur ^ 1) * stg_sz);
                __syncthreads(); DEN_DBG(0);
            }

            // ================= D: ffn.linear1 + GELU (replicated) =================
            nxt.load(Q ? 3 : 1, wave, lane, vp, Ln, tt_next, ln, A, b, c, ca_next, ca_R);
            if constexpr (!Q && !WIN) cl_refills<WT, C, Q, G::U_C + C_INL>(ring, wave, voff, wg, bc, bn, skip, nskip, std::make_integer_sequence<int, G::TB - C_INL>{});       // slots of stage C
            {
                f32x4 acc[1];
                cl_zero<1>(acc);
                cl_units<WT, C, Q, G::U_D, 1, 1, 0, 0, (WIN ? 0 : 1)>(ring, xa, acc, true, wave, voff, wg, bc, bn, skip, nskip, std::make_integer_sequence<int, 1>{});
                if (lane < 16) {
                    const int j = wave * 16 + col;
                    cl_put1<WT>(XH, j, fast_gelu(cl_out<WT>(acc[0]) + v_f1b[j]));
                }
            }
            nxt.store(lane, STG + (cur ^ 1) * stg_sz);
            __syncthreads(); DEN_DBG(0);
            // ================= E: ffn.linear2 -> LayerNorm, AdaLN, SiLU (replicated) =================
            {
                f32x4 acc[2];
                cl_zero<2>(acc);
                cl_units<WT, C, Q, G::U_E, 1, 2, 0, 0, (WIN ? 0 : 1)>(ring, xh, acc, true, wave, voff, wg, bc, bn, skip, nskip, std::make_integer_sequence<int, 1>{});
                if (lane < 32) PART[(2 * wave + (lane >> 4)) * 16 + col] = (lane & 16) ? cl_out<WT>(acc[1]) : cl_out<WT>(acc[0]);
            }
            __syncthreads(); DEN_DBG(0);
            if (epi) {
                if constexpr (WIN) {
                    cl_issue_seq<WT, C, Q>(ring, wave, voff, wg, bc, bn, skip, nskip, typename SCH::W3{});
                    if (lane == 0) cl_flag_set(FLG + 3, e2);
                }
                const float4 y2 = f4_add(ld4(PART + 4 * lane), ld4(v_f2b + 4 * lane));
                const float4 hh = f4_adaln(wave_ln(y2, v_fsnw, v_fsnb, lane), ld4(TTS + 1024 + 4 * lane), ld4(TTS + 1280 + 4 * lane));
                put_x<WT, 1>(XB, 0, 0, lane, f4_silu(hh));
            } else if constexpr (WIN) {
                cl_flag_wait(FLG + 3, e2);
                cl_issue_seq<WT, C, Q>(ring, wave, voff, wg, bc, bn, skip, nskip, typename SCH::W3{});
            }
            __syncthreads(); DEN_DBG(0);
            // ================= F: ffn.proj_out.out_layers + residual (replicated); writes the next layer's input =================
            nxt.load(Q ? 4 : 2, wave, lane, vp, Ln, tt_next, ln, A, b, c, ca_next, ca_R);
            if constexpr (WIN) cl_issue_seq<WT, C, Q>(ring, wave, voff, wg, bc, bn, skip, nskip, typename SCH::PF{});
            {
                f32x4 acc[2];
                cl_zero<2>(acc);
                cl_units<WT, C, Q, G::U_F, 2, 2, 0, 0, (WIN ? 0 : 2)>(ring, xb, acc, true, wave, voff, wg, bc, bn, skip, nskip, std::make_integer_sequence<int, 2>{});
                if constexpr (WIN) cl_issue_seq<WT, C, Q>(ring, wave, voff, wg, bc, bn, skip, nskip, typename SCH::AF{});
                if (lane < 32) {
                    const int n = (2 * wave + (lane >> 4)) * 16 + col;
                    const float xn = RES[n] + ((lane & 16) ? cl_out<WT>(acc[1]) : cl_out<WT>(acc[0])) + v_fo_b[n];
                    RES[n] = xn;
                    cl_put1<WT>(XA, n, xn);
                    if (l < 2) cl_put1<WT>(SKF + 512 * l, n, xn);                      // xs.append(x) (cross_attention.py:70-72)
                }
            }
            nxt.store(lane, STG + (cur ^ 1) * stg_sz);
            // (phantom units that pad the layer program are never consumed: the re-fills they would trigger go out here)
            if constexpr (!WIN) cl_refills<WT, C, Q, G::NU_REAL>(ring, wave, voff, wg, bc, bn, skip, nskip, std::make_integer_sequence<int, G::NU - G::NU_REAL>{});
            __syncthreads(); DEN_DBG(0);
            if (l + 1 < SEEME_DEN_NL) {
                if (epi && !nskip) xr = ld4(RES + 4 * lane);                          // residual of the next layer's attention
            } else {
                // ---- stack norm -> model output (cross_attention.py:82-83), scheduler.step (mld.py:495-497)
                if (epi) {
                    float4 e = wave_ln(ld4(RES + 4 * lane), CONSTV + 256, CONSTV + 512, lane);
                    if (A.sched == SEEME_SCHED_NONE) {
                        st4(KEEP + 4 * lane, e);
                    } else {
                        const CF32 cf = coef_c + (size_t)step * 8;
                        const float c0 = cf[0], c1 = cf[1], c2 = cf[2], c3 = cf[3], c4 = cf[4], c5 = cf[5], clip = cf[6], ptype = cf[7];
                        float4 nz = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (A.noise != nullptr) nz = ld4(A.noise + ((size_t)step * A.B + b) * 256 + 4 * lane);
                        const float4 lat = ld4(KEEP + 4 * lane);
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
                        st4(KEEP + 4 * lane, nl);
                        xr = f4_add(nl, ld4(CONSTV + 4 * lane));
                        put_x<WT, 1>(XA, 0, 0, lane, xr);
                        st4(RES + 4 * lane, xr);
                    }
                }
                __syncthreads(); DEN_DBG(0);
            }
            cur ^= 1;
        }
        row = row_next;
    }
    DEN_DBG(3);
    if (epi && c == 0) st4(A.out + (size_t)b * 256 + 4 * lane, ld4(KEEP + 4 * lane));
#pragma unroll
    for (int s = 0; s < W::RU; ++s) asm volatile("" ::"v"(ring.r[s][0].x));
}

template <int C>
static size_t cl_lds_bytes(int N, bool q) {
    typedef ClG<C> G;
    const int stg_sz = VP_LAYER + STG_TT + N * 1024 + (q ? 0 : 256);
    return (size_t)(768 + 256 + 2 * stg_sz + 512 + 512 + 2 * (G::NB > 128 ? G::NB : 128) + 1024 + 2 * G::S + 256 + 256 + 4) * sizeof(float);
}

extern "C" size_t seeme_den_cluster_xchg_bytes(int B, int C) {       // (the several-token layout: the larger of the two)
    const size_t per = C == 8 ? ClG<8, true>::G_TOTAL : (C == 4 ? ClG<4, true>::G_TOTAL : ClG<2, true>::G_TOTAL);
    const size_t clusters = (size_t)(B + 7) / 8 * 8;
    return DCL_HDR_BYTES + clusters * per * 8;
}
// query = 0: one condition token (tabulated ca term); 1: several (the ca_block's query and proj_out stages are units of the image).
// out[0] = units per (layer, CU) incl. phantom padding, [1] = bytes per unit, [2] = image bytes, [3] = wave-loads per unit, [4] = k per wave-load,
// [5..11] = first unit of A (x half), A (skip half), B, C, D, E, F, [12] = first unit of G (ca query; -1 without), [13] = of H (ca proj_out)
extern "C" int seeme_den_cluster_layout(int C, int wdtype, int query, int64_t* out, int cap) {
    if (cap < 14) return seeme_fail("den_cluster_layout: output too small");
    if (C != 2 && C != 4 && C != 8) return seeme_fail("den_cluster_layout: C must be 2, 4 or 8");
    const int TA = 256 / C / 32, TB = 8 / C;
    const int u_c = 2 * TA + TB, u_g = u_c + TB, u_d = query ? u_g + 3 : u_g;
    const int64_t NU = (u_d + 4 + 3) / 4 * 4, UL = wdtype == 0 ? 16 : 8, UB = 8 * UL * 1024;
    out[0] = NU; out[1] = UB; out[2] = (int64_t)SEEME_DEN_NL * C * NU * UB; out[3] = UL; out[4] = 256 / UL;
    out[5] = 0; out[6] = TA; out[7] = 2 * TA; out[8] = u_c; out[9] = u_d; out[10] = u_d + 1; out[11] = u_d + 2;
    out[12] = query ? u_g : -1; out[13] = query ? u_g + 1 : -1;
    return 0;
}

template <typename WT, int C, bool Q>
static int launch_den_cluster(const ClArgs& ka0, hipStream_t st) {
    ClArgs ka = ka0;
    ka.clusters = (ka.s.B + 7) / 8 * 8;
    const int grid = ka.clusters * C;
    // co-residency of every workgroup is what the in-launch exchanges rest on: one workgroup per CU of THIS device (a partitioned or
    // CU-masked device has fewer than 256)
    int dev = 0, cus = 0;
    SEEME_HIP(hipGetDevice(&dev));
    SEEME_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (grid > cus || grid > 256) return seeme_fail("denoiser_sample_cluster: B x C exceeds one workgroup per CU of this device");
    const size_t lds = cl_lds_bytes<C>(Q ? ka.s.N : 1, Q);
    if (lds > 160 * 1024 || lds <= 80 * 1024) return seeme_fail("denoiser_sample_cluster: LDS footprint must force one workgroup per CU");
    SEEME_HIP(hipFuncSetAttribute((const void*)k_den_cluster<WT, C, Q>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // granules and header are zeroed before EVERY launch (tags of an earlier launch must never match)
    SEEME_HIP(hipMemsetAsync(ka.hdr, 0, seeme_den_cluster_xchg_bytes(ka.s.B, C), st));
    hipLaunchKernelGGL((k_den_cluster<WT, C, Q>), dim3(grid), dim3(DEN_THREADS), lds, st, ka);
    return seeme_check_launch("k_den_cluster");
}
template <typename WT>
static int launch_den_cluster_c(const ClArgs& ka, int C, bool q, hipStream_t st) {
    if (C == 8) return q ? launch_den_cluster<WT, 8, true>(ka, st) : launch_den_cluster<WT, 8, false>(ka, st);
    if (C == 4) return q ? launch_den_cluster<WT, 4, true>(ka, st) : launch_den_cluster<WT, 4, false>(ka, st);
    if (C == 2) return q ? launch_den_cluster<WT, 2, true>(ka, st) : launch_den_cluster<WT, 2, false>(ka, st);
    return seeme_fail("denoiser_sample_cluster: C must be 2, 4 or 8");
}

static int den_cluster_ms_dispatch(const SeemeDenoiserWeights* w, const SeemeDenCluster* cl, const SeemeSampleArgs* a, ClArgs& ka, hipStream_t st);   // den_cluster_ms.inc.hip

extern "C" int seeme_denoiser_sample_cluster(const SeemeDenoiserWeights* w, const SeemeDenCluster* cl, const SeemeSampleArgs* a, void* stream) {
    if (a->B <= 0) return seeme_fail("denoiser_sample_cluster: B must be > 0");
    if (a->N < 1 || a->N > DCL_MAX_N) return seeme_fail("denoiser_sample_cluster: 1 or 2 condition tokens");
    const bool q = a->N > 1;
    if (!q && (a->catab == nullptr || a->force_query)) return seeme_fail("denoiser_sample_cluster: one condition token needs its ca table");
    if (q != (cl->query != 0)) return seeme_fail("denoiser_sample_cluster: the weight image was packed for the other ca_block variant (SeemeDenCluster.query)");
    if (w->nhead != 1 || !w->sa_fold) return seeme_fail("denoiser_sample_cluster: one attention head (folded out_proj)");
    if (a->cfg || a->save != nullptr) return seeme_fail("denoiser_sample_cluster: no CFG pair, no training forward");
    if (w->ff_sa != FF_SA || w->ff != FF_D) return seeme_fail("denoiser_sample_cluster: built for sa ff 1024 / ffn_dim 128");
    if (a->sched == SEEME_SCHED_NONE && a->steps != 1) return seeme_fail("denoiser_sample_cluster: SCHED_NONE needs steps == 1");
    if (a->steps < 1) return seeme_fail("denoiser_sample_cluster: steps must be >= 1");
    const bool ms = cl->samples > 1;
    if (cl->xchg == nullptr || (!ms && cl->xchg_bytes < seeme_den_cluster_xchg_bytes(a->B, cl->C))) return seeme_fail("denoiser_sample_cluster: exchange buffer too small");
    if (((uintptr_t)cl->xchg & 15) != 0) return seeme_fail("denoiser_sample_cluster: exchange buffer must be 16-byte aligned");
    int64_t lo[14];
    int rc = seeme_den_cluster_layout(cl->C, cl->wdtype, q ? 1 : 0, lo, 14);
    if (rc) return rc;
    ClArgs ka;
    ka.wgc = cl->wgc; ka.wgc_bytes = (unsigned)lo[2]; ka.vp = cl->vpc;
    ka.lay = seeme_make_den_layout(FF_SA, FF_D);
    ka.s = *a;
    ka.hdr = (unsigned*)cl->xchg;
    ka.xg = (unsigned long long*)((char*)cl->xchg + DCL_HDR_BYTES);
    ka.placement = cl->placement; ka.flags = cl->flags; ka.clusters = 0; ka.spc = 1;
    hipStream_t st = (hipStream_t)stream;
    if (ms) return den_cluster_ms_dispatch(w, cl, a, ka, st);
    if (cl->wdtype == 0) return launch_den_cluster_c<WF32>(ka, cl->C, q, st);
    if (cl->wdtype == 1) return launch_den_cluster_c<WBF16>(ka, cl->C, q, st);
    if (cl->wdtype == 2) return launch_den_cluster_c<WF16>(ka, cl->C, q, st);
    return seeme_fail("denoiser_sample_cluster: wdtype must be 0 (fp32), 1 (bf16) or 2 (fp16)");
}
