// den_cluster_ms.inc.hip -- k_den_cluster_ms<C>: the cluster kernel for LARGE batches (included by den_kernels.hip after den_cluster.inc.hip).
//
// Reference: the same chain as k_den_cluster / k_den_sample -- MldDenoiser.forward (mld_denoiser.py:151-244) driven by
// MLD._diffusion_reverse (mld.py:467-497); BASELINE configs[4] (1000-step DDPM, 512 sequences per GPU, fp16) is the case it is for.
//
// Why: above B = 32 the one-sample cluster needs more CUs than the chip has, and k_den_sample streams the whole 9 MB image per CU and
// step (two samples share it above B = 256): 97 us per step at B = 512, bound by the 118 GB/s of one CU's vector-memory path.  But the
// GEMVs run on the matrix cores with the input vector as the A operand, and only 2 of the 16 A rows (hi and lo half of the fp32 value)
// belong to a sample.  Here a cluster of C CUs owns up to EIGHT samples: row 2 s + p of every A operand is part p of sample s, so one
// pass over the cluster's weight slices serves eight chains (same wave-loads, same MFMAs as for one), wave s is the epilogue wave of
// sample s (softmax over the 2 + N tokens, the LayerNorms, AdaLN, scheduler step -- eight of them side by side instead of one with
// seven waves idle), and every sample has its own granule block and its own two exchanges per layer.
//   64 < B <= 512: C = 4, 64 clusters x ceil(B / 64) samples   (host policy, seeme_amd/mld_denoiser.py::_cluster_plan; C = 8 is built too and
//   measures slower: a CU's exchange volume grows with C x samples; up to B = 64 the windowed one-sample kernel on 4 CUs is faster)
// fp16 weight image (the bench default; two A rows per sample), one condition token (tabulated ca term) or two (template Q: query / proj_out
// stages and a third exchange per sample), one head, no CFG, one timestep row per step: everything else stays on k_den_sample / k_den_cluster.  The weight image, the unit program and the exchange
// protocol are k_den_cluster's (inline load schedule), results are bit-identical to it; LDS differs: the A-operand buffers hold 16 rows,
// and the per-layer vector operands are kept in a COMPACT block (only the slices this CU reads: 2 944 floats instead of 6 272 at C = 4)
// with the samples' K slices behind it; a sample's V' row and its tabulated ca term are read from global memory by its epilogue wave
// right before it waits for an exchange.  109 KB (C = 4) / 98 KB (C = 8) of LDS.
// Measured (50 DDIM steps, profiles/r03_g_cluster_ms.txt): B = 128 3.44 ms (k_den_cluster with 2 CUs per sample 4.30, k_den_sample 4.46),
// B = 256 3.66 (4.70), B = 512 4.37 (4.92); DESIGN.md section 5.1b has the three stages that led here and what bounds it now (the exchanges).

#ifndef DCLM_POLL_SLEEP
#define DCLM_POLL_SLEEP 8     // x 64 cycles between a wave's publish and its first poll
#endif
#ifndef DCLM_SENTINEL_FROM
#define DCLM_SENTINEL_FROM 5  // clusters with at least this many samples first poll ONE granule per (publisher, writing wave) -- a single 8-byte load per
#endif                        // lane and round -- and sweep once they match: one more round trip per exchange, but eight waves' full sweeps no longer
                              // circle through the CU's memory queue while they wait (B = 512: 4.94 -> 4.79 ms; B = 128, two samples: 3.60 -> 3.65,
                              // and k_den_cluster at B = 32: 2.44 -> 2.6 ms -- there the successful poll IS the sweep)

#ifndef DCLM_IDLE_FLAGS
#define DCLM_IDLE_FLAGS 1     // waves without a sample request their window share behind sample 0's publish / sweep (words in LDS), not at once
#endif
#ifndef DCLM_PUB_ALL
#define DCLM_PUB_ALL 1        // an epilogue wave's own window requests also wait until every sample of the CU has published
#endif
#ifndef DCLM_WIN
#define DCLM_WIN 1            // 1: every unit of the weight stream is requested inside an exchange / vector-algebra window (below); 0: inline, as consumed
#endif
// The windowed schedule.  Requested as consumed (k_den_cluster's inline schedule) the stream is exposed time: a 1-KiB wave-request costs the CU's
// address path 16+ cycles, the issuing wave waits its turn, and at 608-700 KB per layer and CU (C = 4) that is 26 k of a layer's 41 k cycles.  Here
// every wave is busy in every phase, so there are no idle waves to do the requesting -- but every wave has dead time of its own: between
// publishing and the first poll of an exchange, and behind its sweep.  Each wave requests ITS eighth of the coming units there:
//   W1A after the X1 publish (stage B's units), W1B behind the X1 sweep (stage C's), W2A after the X2 publish (D, E), W2B behind its sweep (F),
//   W3 in the ffn epilogue (next layer's A, x half), AF after stage F (next layer's A, skip half).  Ring slot of unit U = U % 4: a unit goes
//   out only after unit U - 4 was consumed (C = 4: A 0 1 | AS 2 3 | B 4 5 | C 6 7 | D 8 | E 9 | F 10 11;  C = 8: A 0 | AS 1 | B 2 | C 3 | D 4 | E 5 | F 6 7).
// Two condition tokens (Q): the unit tables and windows of ClSched<C, true, true> (den_cluster.inc.hip), WXA / WXB around the third exchange.
template <int C, bool Q> struct ClmSched;
template <> struct ClmSched<4, false> { static constexpr int PRO = 4; typedef ClSeq<4, 5> W1A; typedef ClSeq<6, 7> W1B; typedef ClSeq<8, 9> W2A; typedef ClSeq<10, 11> W2B; typedef ClSeq<> WXA, WXB; typedef ClSeq<12, 13> W3; typedef ClSeq<14, 15> AF; };
template <> struct ClmSched<8, false> { static constexpr int PRO = 2; typedef ClSeq<2, 3> W1A; typedef ClSeq<4> W1B; typedef ClSeq<5> W2A; typedef ClSeq<6> W2B; typedef ClSeq<> WXA, WXB; typedef ClSeq<7> W3; typedef ClSeq<8, 9> AF; };
template <> struct ClmSched<4, true> { static constexpr int PRO = 4; typedef ClSeq<4, 5> W1A; typedef ClSeq<6, 7> W1B; typedef ClSeq<8, 9> W2A; typedef ClSeq<10> W2B; typedef ClSeq<11> WXA; typedef ClSeq<12> WXB; typedef ClSeq<13, 14> W3; typedef ClSeq<16, 17, 18, 19> AF; };
template <> struct ClmSched<8, true> { static constexpr int PRO = 2; typedef ClSeq<2, 3> W1A; typedef ClSeq<4> W1B; typedef ClSeq<5> W2A; typedef ClSeq<6> W2B; typedef ClSeq<7> WXA; typedef ClSeq<8> WXB; typedef ClSeq<9, 10> W3; typedef ClSeq<12, 13> AF; };

// A rows per sample.  NARROW (fp16 only): two -- (hi, lo) -- so that 8 samples share the 16 rows and a lane group's accumulator rows 4 g .. 4 g + 3 hold
// two samples; otherwise four -- fp16 (hi, lo, -, -), bf16 (hi, mid, lo, -) -- 4 samples, one per lane group (half the epilogue work per lane: the
// form used while a cluster owns at most four samples).
template <typename WT, bool NARROW> struct ClmRows {
    static_assert(WT::HALF || !NARROW, "the bf16 split has three parts");
    static constexpr int MS = NARROW ? 8 : 4, SPL = NARROW ? 2 : 1, ROWB = NARROW ? 32 : 64;
};
template <typename WT, int C, bool Q = false, bool NARROW = true> struct ClM {
    typedef ClG<C, Q> G;
    static constexpr int MS = ClmRows<WT, NARROW>::MS;   // samples per cluster
    static constexpr int SPL = ClmRows<WT, NARROW>::SPL; // samples in a lane group's accumulator rows
    static constexpr int XKB = 1024;                     // bytes per k-block of an A-operand buffer: [k-group 4][row 16][8 halves]
    static constexpr int XBUF = 256 / 32 * XKB / 4;      // floats of a 256-k operand buffer (8 KiB)
    static constexpr int XHK = G::NB > 128 ? G::NB : 128;
    static constexpr int XHBUF = XHK / 32 * XKB / 4;
    // compact per-layer operand block (floats; the same positions for every layer)
    static constexpr int O_SKIPB = 0, O_INB = G::S /* q | k | v' slices */, O_N1W = 4 * G::S, O_N1B = O_N1W + 256, O_L2B = O_N1B + 256,
                         O_N2W = O_L2B + 256, O_N2B = O_N2W + 256, O_F1B = O_N2B + 256, O_F2B = O_F1B + FF_D, O_FSNW = O_F2B + 256,
                         O_FSNB = O_FSNW + 256, O_FOB = O_FSNB + 256, O_L1B = O_FOB + 256,
                         O_TK = O_L1B + G::NB /* time token K | V' | ffn AdaLN scale | shift */, O_TV = O_TK + 256, O_TSC = O_TV + 256, O_TSH = O_TSC + 256,
                         // (Q) ca AdaLN scale | shift, ca_block.norm weight | bias, proj_out.norm weight | bias, proj_out bias, this CU's slice of the query bias
                         O_CSC = O_TSH + 256, O_CSH = O_CSC + (Q ? 256 : 0), O_CNW = O_CSH + (Q ? 256 : 0), O_CNB = O_CNW + (Q ? 256 : 0),
                         O_CSNW = O_CNB + (Q ? 256 : 0), O_CSNB = O_CSNW + (Q ? 256 : 0), O_CAOB = O_CSNB + (Q ? 256 : 0), O_CAQB = O_CAOB + (Q ? 256 : 0),
                         O_SMP = O_CAQB + (Q ? G::S : 0);   // per sample [MS][SMPF]: this CU's dims of the condition token's K (Q: sa K of token 0 | token 1 |
                                                            // ca key of token 0 | token 1); V' / ca value rows and the tabulated ca term are read from global
                                                            // memory by the sample's epilogue wave, under its exchanges
    static constexpr int SMPF = (Q ? 4 : 1) * G::S;
    static constexpr int STG = O_SMP + MS * SMPF;
    static constexpr int NL1 = G::NB > 256 ? G::NB / 256 : 1;
    static constexpr int NSH = 14 + NL1 + 4 + (Q ? 8 : 0);   // pieces shared by the samples
    static constexpr int NKP = MS * SMPF / 256;          // pieces of the samples' K slices
    static constexpr int NPIECE = NSH + NKP;             // one piece per wave in each of the phases B, D, F (Q: B, C, H, D, F), as k_den_cluster
    static constexpr int NPH = Q ? 5 : 3;
    static_assert(NPIECE <= 8 * NPH && MS * SMPF % 256 == 0, "staging pieces");
    static constexpr int LDS_FLOATS = 768 + MS * 256 + 2 * STG + 2 * XBUF + XHBUF + 2 * XBUF + 2 * MS * G::S + 2 * MS * 256 + 4;
};

// A-operand buffers with 16 rows: row 2 s = hi half of sample s, row 2 s + 1 = lo half (scaled by DEN_F16_LO_SCALE)
__device__ __forceinline__ ClX clm_xin(const float* buf, int lane) {
    ClX x; x.base = reinterpret_cast<const char*>(buf);
    x.foff = ((lane >> 4) * 16 + (lane & 15)) * 16;
    x.kbs = 1024;
    return x;
}
template <typename WT, bool NARROW>
__device__ __forceinline__ void clm_put1(float* buf, int s, int k, float v) {
    char* dst = reinterpret_cast<char*>(buf) + (k >> 5) * 1024 + ((k >> 3) & 3) * 256 + s * ClmRows<WT, NARROW>::ROWB + (k & 7) * 2;
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
template <typename WT, bool NARROW>
__device__ __forceinline__ void clm_put4(float* buf, int s, int lane, float4 v) {      // values k = 4 lane .. 4 lane + 3 of sample s
    const int k = 4 * lane;
    char* dst = reinterpret_cast<char*>(buf) + (k >> 5) * 1024 + ((k >> 3) & 3) * 256 + s * ClmRows<WT, NARROW>::ROWB + (k & 7) * 2;
    if constexpr (WT::HALF) {
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        const h4 hi = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
        const h4 lo = {(_Float16)((v.x - (float)hi.x) * DEN_F16_LO_SCALE), (_Float16)((v.y - (float)hi.y) * DEN_F16_LO_SCALE),
                       (_Float16)((v.z - (float)hi.z) * DEN_F16_LO_SCALE), (_Float16)((v.w - (float)hi.w) * DEN_F16_LO_SCALE)};
        *reinterpret_cast<h4*>(dst) = hi;
        *reinterpret_cast<h4*>(dst + 16) = lo;
    } else {
        typedef __bf16 b4 __attribute__((ext_vector_type(4)));
        const b4 hi = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
        const float4 r1 = make_float4(v.x - (float)hi.x, v.y - (float)hi.y, v.z - (float)hi.z, v.w - (float)hi.w);
        const b4 mid = {(__bf16)r1.x, (__bf16)r1.y, (__bf16)r1.z, (__bf16)r1.w};
        const b4 lo = {(__bf16)(r1.x - (float)mid.x), (__bf16)(r1.y - (float)mid.y), (__bf16)(r1.z - (float)mid.z), (__bf16)(r1.w - (float)mid.w)};
        *reinterpret_cast<b4*>(dst) = hi;
        *reinterpret_cast<b4*>(dst + 16) = mid;
        *reinterpret_cast<b4*>(dst + 32) = lo;
    }
}
// the samples of a lane's accumulator: lane group g = lane >> 4 holds D rows 4 g .. 4 g + 3 = fp16: (hi, lo) of samples 2 g and 2 g + 1;
// bf16: (hi, mid, lo, -) of sample g
template <typename WT, bool NARROW>
__device__ __forceinline__ float clm_out(const f32x4& a, int j) {
    if constexpr (WT::HALF) return (NARROW && j) ? fmaf(a.w, 1.f / DEN_F16_LO_SCALE, a.z) : fmaf(a.y, 1.f / DEN_F16_LO_SCALE, a.x);
    else return a.x + (a.y + a.z);
}

// FLG[2] counts the samples whose epilogue wave has published its part of an exchange (ever: nact per exchange).  A window's requests go out
// only once ALL of this CU's samples have published: requests in the queue in front of another wave's granule stores delay the whole cluster.
__device__ __forceinline__ void clm_wait_published(const int* f, int target) {
    while (*(const cl_lds_flag*)f < target) __builtin_amdgcn_s_sleep(1);
}
__device__ __forceinline__ void clm_published(int* f, int lane, int target) {
#if DCLM_IDLE_FLAGS
    if (lane == 0) __hip_atomic_fetch_add((__attribute__((address_space(3))) int*)f, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#if DCLM_PUB_ALL
    clm_wait_published(f, target);
#endif
#endif
}

// one piece (<= 1 KiB) of the NEXT layer's operands per wave and phase k = 0 .. 5: requested at the top of the phase, stored to the other
// half of the staging buffer at its end (k_den_cluster's ClStage, with the compact destination layout and MS samples' rows)
template <typename WT, int C, bool Q, bool NARROW>
struct ClmStage {
    float4 r; int dst, n4;
    __device__ __forceinline__ void load(int k, int wave, int lane, const float* __restrict__ vpg, const DenLayerOff* __restrict__ L,
                                         const float* __restrict__ tt_row, int l, const SeemeSampleArgs& A, int b0, int nact, int c) {
        typedef ClM<WT, C, Q, NARROW> M; typedef ClG<C, Q> G;
        const float* vb = vpg + L->skip_b;
        const int i = wave + 8 * k;
        const float* src = vb; int d = 0, n = 0;
        if (i == 0) { src = vb + c * G::S; d = M::O_SKIPB; n = G::S / 4; }
        else if (i <= 3) { src = vb + (L->in_b - L->skip_b) + (i - 1) * 256 + c * G::S; d = M::O_INB + (i - 1) * G::S; n = G::S / 4; }
        else if (i <= 8) {
            const int64_t f = i == 4 ? L->n1w : i == 5 ? L->n1b : i == 6 ? L->l2b : i == 7 ? L->n2w : L->n2b;
            src = vpg + f; d = M::O_N1W + (i - 4) * 256; n = 64;
        }
        else if (i == 9) { src = vpg + L->f1b; d = M::O_F1B; n = FF_D / 4; }
        else if (i <= 13) {
            const int64_t f = i == 10 ? L->f2b : i == 11 ? L->fsnw : i == 12 ? L->fsnb : L->fo_b;
            src = vpg + f; d = M::O_F2B + (i - 10) * 256; n = 64;
        }
        else if (i < 14 + M::NL1) { src = vpg + L->l1b + c * G::NB + (i - 14) * 256; d = M::O_L1B + (i - 14) * 256; n = G::NB >= 256 ? 64 : G::NB / 4; }
        else if (i < 18 + M::NL1) {
            const int j = i - 14 - M::NL1;                 // time token K | V' | ffn AdaLN scale | shift
            src = j < 2 ? tt_row + l * 512 + j * 256 : tt_row + 2560 + l * 1024 + 512 + (j - 2) * 256;
            d = M::O_TK + j * 256; n = 64;
        }
        else if (Q && i < M::NSH) {
            const int j = i - 18 - M::NL1;                 // ca AdaLN scale | shift; ca_block.norm, proj_out.norm, proj_out bias; query bias slice
            if (j < 2) { src = tt_row + 2560 + l * 1024 + j * 256; d = M::O_CSC + j * 256; n = 64; }
            else if (j < 7) {
                const int64_t f = j == 2 ? L->cnw : j == 3 ? L->cnb : j == 4 ? L->csnw : j == 5 ? L->csnb : L->cao_b;
                src = vpg + f; d = M::O_CNW + (j - 2) * 256; n = 64;
            }
            else { src = vpg + L->caq_b + c * G::S; d = M::O_CAQB; n = G::S / 4; }
        }
        else if (i < M::NPIECE) {                          // the samples' K slices: 256 floats per piece, each lane from its own sample's row --
            // through a buffer over the condition tables with a 32-bit lane offset (a per-lane 64-bit address kept across the layer loop is
            // spilled, and its reload drains the vector-memory queue)
            int ll = lane;
            asm volatile("" : "+v"(ll));                 // (recomputed at every call: hoisted out of the layer loop the offset is spilled as well)
            const int e = (i - M::NSH) * 256 + 4 * ll, s = e / M::SMPF, r_ = e - s * M::SMPF, q = r_ / G::S, off = r_ - q * G::S;
            const int bs = b0 + (s < nact ? s : 0);        // (slots beyond the batch read sample b0: never used)
            const int N = Q ? 2 : 1, t = q & 1;            // q: sa K of token 0 [| token 1 | ca key of token 0 | token 1]
            const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A.ctab), 0, A.B * N * SEEME_CROW * 4, 0x00020000);
            const unsigned vo = (unsigned)(((bs * N + t) * SEEME_CROW + (q < 2 ? 0 : 2560) + c * G::S + off) * 4);
            dst = M::O_SMP + (i - M::NSH) * 256; n4 = 64;
            r = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rc, vo, (unsigned)(l * 512 * 4), 0));
            return;
        }
        dst = d; n4 = n;
        r = make_float4(0.f, 0.f, 0.f, 0.f);
#ifdef DCLM_ABL_NOSTAGE   // timing-only build: the layer operands are never re-staged (wrong results)
        if (k >= 0 && l >= 0 && lane < 0)
#else
        if (lane < n)
#endif
            r = *reinterpret_cast<const float4*>(src + 4 * lane);
    }
    __device__ __forceinline__ void store(int lane, float* __restrict__ stg) const {
        if (lane < n4) *reinterpret_cast<float4*>(stg + dst + 4 * lane) = r;
    }
};

template <typename WT, int C, bool Q, bool NARROW>
__global__ __launch_bounds__(DEN_THREADS) void k_den_cluster_ms(const ClArgs ka) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef ClW<WT> W; typedef ClG<C, Q> G; typedef ClM<WT, C, Q, NARROW> M;
    constexpr int SPL = M::SPL;
    constexpr int MS = M::MS;
    constexpr int N = Q ? 2 : 1;                 // condition tokens
    constexpr int NT = N + 2;                    // score slots: self, condition token(s), time
    constexpr int A_DEF0 = cl_clamp(2 * G::TA - W::RU, 0, 2 * G::TA);
    constexpr bool WIN = DCLM_WIN != 0;
    static_assert(WIN || !Q, "two condition tokens: windowed schedule only");
    constexpr int A_INL0 = WIN ? 0 : cl_clamp(A_DEF0, 0, G::TA), A_INL1 = WIN ? 0 : cl_clamp(A_DEF0 - G::TA, 0, G::TA), C_INL = WIN ? 0 : cl_clamp(G::TB - W::RU, 0, G::TB);
    typedef ClmSched<C, Q> SCH;
    const SeemeSampleArgs& A = ka.s;
    const DenLayout* __restrict__ lay = &ka.lay;
    const float* __restrict__ vp = ka.vp;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int kc, c;                                                   // cluster, member (placement: see k_den_cluster)
    if (ka.placement == 0) { const int x = blockIdx.x & 7, j = blockIdx.x >> 3; kc = x * (ka.clusters / 8) + j / C; c = j % C; }
    else { kc = blockIdx.x / C; c = blockIdx.x % C; }
    const int b0 = kc * ka.spc;
    const int nact = min(ka.spc, A.B - b0);                      // samples of this cluster (whole clusters: every member sees the same)
    if (nact <= 0) return;
    const bool epi = wave < nact;                                // wave s is the epilogue wave of sample s
    const int es = wave;
    const int bs = b0 + (epi ? es : 0);
    const int ca_R = A.steps;                                    // (one table row per step: trow_per_sample stays on the other kernels)
    const bool sentinel = ka.spc >= DCLM_SENTINEL_FROM;
    const int g2 = (lane >> 4) * SPL;                            // this lane's accumulators hold samples g2 .. g2 + SPL - 1

    float* CONSTV = smem;                        // [768]  query_pos.pe[0], encoder.norm.{weight,bias}
    float* KEEP = CONSTV + 768;                  // [MS][256] the latents
    float* STG = KEEP + MS * 256;                // [2][M::STG] per-layer operands, double-buffered
    float* XA = STG + 2 * M::STG;                // A-operand buffers (16 rows): layer input x / x2
    float* XB = XA + M::XBUF;                    //   x1 / u
    float* XH = XB + M::XBUF;                    //   this CU's hidden units / ffn hidden
    float* SKF = XH + M::XHBUF;                  // [2] outputs of layers 0, 1 (skip inputs of layers 4, 3)
    float* QS = SKF + 2 * M::XBUF;               // [MS][S] q, [MS][S] k of this CU's dims
    float* KS = QS + MS * G::S;
    float* PART = KS + MS * G::S;                // [MS][256]
    float* RES = PART + MS * 256;                // [MS][256] the fp32 residual streams
    int* FLG = reinterpret_cast<int*>(RES + MS * 256);
    constexpr int XZERO = 2 * M::XBUF + M::XHBUF + 2 * M::XBUF;

    const __amdgpu_buffer_rsrc_t wg = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ka.wgc), 0, (int)ka.wgc_bytes, 0x00020000);
    unsigned long long* const xg0 = ka.xg + (size_t)kc * MS * G::G_TOTAL;            // granule blocks of this cluster's samples
    unsigned long long* const xg = xg0 + (size_t)(epi ? es : 0) * G::G_TOTAL;        // ... of this wave's sample
    const __amdgpu_buffer_rsrc_t xr_ = __builtin_amdgcn_make_buffer_rsrc(xg, 0, G::G_TOTAL * 8, 0x00020000);
    // this wave's sample: its condition-table row and its block of the tabulated ca term, as buffers (scalar base + lane offset: a per-lane
    // 64-bit address kept across the layer loop is spilled, and its reload drains the vector-memory queue)
    const __amdgpu_buffer_rsrc_t rs_ct = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A.ctab + (size_t)bs * N * SEEME_CROW), 0, N * SEEME_CROW * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_ca = __builtin_amdgcn_make_buffer_rsrc(Q ? nullptr : const_cast<float*>(A.catab + (size_t)bs * ca_R * SEEME_DEN_NL * 256), 0,
                                                                           Q ? 0 : ca_R * SEEME_DEN_NL * 1024, 0x00020000);   // (Q: no tabulated ca term)
    const unsigned voff = (unsigned)wave * (unsigned)(W::UL * 1024) + (unsigned)lane * 16u;
    const int col = lane & 15;
    const ClX xa = clm_xin(XA, lane), xb = clm_xin(XB, lane), xh = clm_xin(XH, lane);
    const float sa_scale = 1.f / 16.f;

    typedef const __attribute__((address_space(4))) int32_t* CI32;
    typedef const __attribute__((address_space(4))) float* CF32;
    const CI32 trow_c = (CI32)(uintptr_t)A.trow;
    const CF32 coef_c = (CF32)(uintptr_t)A.coef;
    int row = trow_c[0];

    // ---- prologue
    float4 xr = make_float4(0.f, 0.f, 0.f, 0.f);
    if (epi) xr = ld4(A.latents + (size_t)bs * 256 + 4 * lane);
    for (int i = tid; i < 192; i += DEN_THREADS)
        st4(CONSTV + 4 * i, i < 64 ? ld4(vp + lay->pe0 + 4 * i) : (i < 128 ? ld4(vp + lay->fnw + 4 * (i - 64)) : ld4(vp + lay->fnb + 4 * (i - 128))));
    {
        ClmStage<WT, C, Q, NARROW> p;
#pragma unroll 1
        for (int k = 0; k < M::NPH; ++k) {
            p.load(k, wave, lane, vp, &lay->L[0], A.ttab + (size_t)row * SEEME_TROW, 0, A, b0, nact, c);
            p.store(lane, STG);
        }
    }
    for (int i = tid; i < XZERO / 4; i += DEN_THREADS) st4(XA + 4 * i, make_float4(0.f, 0.f, 0.f, 0.f));
    if (tid >= 2 && tid < 4) FLG[tid] = 0;                   // window flags (epochs are never 0)
    bool dead = false, local = false;
    if (wave == 0) {
        // first exchange, always write-through: the XCC id of every workgroup of the cluster (sample 0's boot granules)
        const unsigned my_xcc = __builtin_amdgcn_s_getreg(0x1814) & 15u;
        if (lane == 0) cl_store_granule(xg0 + G::G_BOOT + c, DCL_MAGIC_EPOCH, __uint_as_float(my_xcc), false);
        unsigned spins = 0, idv = 0;
        for (;;) {
            const unsigned long long x = __hip_atomic_load((dcl_gu64*)(xg0 + G::G_BOOT + (lane % C)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            idv = (unsigned)x;
            if (__all((unsigned)(x >> 32) == DCL_MAGIC_EPOCH)) break;
            if (++spins > DCL_SPIN_LIMIT) { dead = true; if (lane == 0) atomicOr(ka.hdr, 1u); break; }
            __builtin_amdgcn_s_sleep(2);
        }
        local = !dead && __all(idv == my_xcc) && !(ka.flags & 1);
        if (lane == 0) { FLG[0] = local ? 1 : 0; FLG[1] = dead ? 1 : 0; if (local && c == 0) atomicAdd(ka.hdr + 1, 1u); }
    }
    wait_vmcnt0();
    __syncthreads();
    local = FLG[0] != 0;
    dead = FLG[1] != 0;
    if (epi) {
        st4(KEEP + es * 256 + 4 * lane, xr);
        xr = f4_add(xr, ld4(CONSTV + 4 * lane));                // sample + query_pos (mld_denoiser.py:210)
        clm_put4<WT, NARROW>(XA, es, lane, xr);
        st4(RES + es * 256 + 4 * lane, xr);
    }
    ClRing<WT> ring;
    {
        const unsigned bb = (unsigned)((0 * C + c) * G::NU) * W::UNIT_BYTES;
        cl_issue<WT, C, Q, 0>(ring, wave, voff, wg, bb, bb, false, false);
        cl_issue<WT, C, Q, 1>(ring, wave, voff, wg, bb, bb, false, false);
        if constexpr (!WIN || SCH::PRO == 4) {      // (PRO = 2: units 2, 3 go out in the first X1 window)
            cl_issue<WT, C, Q, 2>(ring, wave, voff, wg, bb, bb, false, false);
            cl_issue<WT, C, Q, 3>(ring, wave, voff, wg, bb, bb, false, false);
        }
        static_assert(W::RU == 4, "prologue issues up to four units");
    }
    __syncthreads();
    int cur = 0;

#pragma unroll 1
    for (int step = 0; step < A.steps; ++step) {
        if (step == 2) DEN_DBG(1);
        if (step == 3) DEN_DBG(2);
        const int step_next = step + 1 < A.steps ? step + 1 : step;
        const int row_next = trow_c[step_next];
#pragma unroll 1
        for (int l = 0; l < SEEME_DEN_NL; ++l) {
            const int ln = (l + 1 < SEEME_DEN_NL) ? l + 1 : 0;
            const DenLayerOff* __restrict__ Ln = &lay->L[ln];
            const bool skip = l >= 3, nskip = ln >= 3;
            const unsigned bc = (unsigned)((l * C + c) * G::NU) * W::UNIT_BYTES, bn = (unsigned)((ln * C + c) * G::NU) * W::UNIT_BYTES;
            const float* VP = STG + cur * M::STG;
            float* const STGN = STG + (cur ^ 1) * M::STG;
            const float* SMP = VP + M::O_SMP + es * M::SMPF;     // this wave's sample: the condition tokens' K (and ca keys) over this CU's dims
            const bool ywave = wave >= 6;
            const unsigned e1 = 1u + (unsigned)G::EPL * (unsigned)(step * SEEME_DEN_NL + l), e2 = e1 + 1u, e3 = e1 + 2u;
            const int pub1 = nact * (int)e1, pub2 = nact * (int)e2, pub3 = nact * (int)e3;   // the LDS count of published samples after X1 / X2 / X3 of this layer (FLG[2])
            (void)e3; (void)pub3;
            ClmStage<WT, C, Q, NARROW> nxt;
            const float* const tt_next = A.ttab + (size_t)(ln == 0 ? row_next : row) * SEEME_TROW;

            // ================= A: in_proj' (+ folded skip linear), column-split by dims =================
            // (measured and not kept: in layers without a skip linear the ring slots of the skip half are idle through stage A, so stage C's units could
            //  go out at its top instead of in the X1 window -- but then they stand in front of stage A's v' granules: 3.50 -> 3.92 ms at B = 128)
            {
                f32x4 acc[G::TA];
                cl_zero<G::TA>(acc);
                const bool act = skip || !ywave;
                cl_units<WT, C, Q, G::U_A, G::TA, G::TA, 0, 0, A_INL0>(ring, xa, acc, act, wave, voff, wg, bc, bn, skip, nskip, std::make_integer_sequence<int, G::TA>{});
                {
                    const ClX xs = clm_xin(SKF + (l == 3 ? M::XBUF : 0), lane);
                    cl_units<WT, C, Q, G::U_AS, G::TA, G::TA, G::TA, W::UL, A_INL1>(ring, xs, acc, skip, wave, voff, wg, bc, bn, skip, nskip, std::make_integer_sequence<int, G::TA>{});
                }
                if (act) {
#pragma unroll
                    for (int t = 0; t < G::TA; ++t) {
                        const int T = wave * G::TA + t, part = T / (G::S / 16), d = (T % (G::S / 16)) * 16 + col;
#pragma unroll
                        for (int j = 0; j < SPL; ++j) {
                            const int sj = g2 + j;
                            if (sj < nact) {
                                const float val = clm_out<WT, NARROW>(acc[t], j);
                                unsigned long long* const xs_ = xg0 + (size_t)sj * G::G_TOTAL + G::G_X1 + c * G::X1_G;
                                if (part == 0) QS[sj * G::S + d] = val + VP[M::O_INB + d];
                                else if (part == 1) KS[sj * G::S + d] = val + VP[M::O_INB + G::S + d];
                                else if (part == 2) cl_store_granule(xs_ + d, e1, val + VP[M::O_INB + 2 * G::S + d], local);
                                else cl_store_granule(xs_ + G::S + d, e1, val + VP[M::O_SKIPB + d], local);
                            }
                        }
                    }
                }
            }
            __syncthreads(); DEN_DBG(0);
            if (epi) {
                const unsigned epoch = e1;
                float ps[NT];                                    // score slots: self, condition token(s), time (mdiff_transformer.py:295)
#pragma unroll
                for (int j = 0; j < NT; ++j) ps[j] = 0.f;
#pragma unroll
                for (int dd = 0; dd < G::S; dd += 64) {
                    const int d = dd + lane;
                    if (d < G::S) {
                        const float q = QS[es * G::S + d];
                        ps[0] = fmaf(q, KS[es * G::S + d], ps[0]);
#pragma unroll
                        for (int n = 0; n < N; ++n) ps[1 + n] = fmaf(q, SMP[n * G::S + d], ps[1 + n]);
                        ps[NT - 1] = fmaf(q, VP[M::O_TK + c * G::S + d], ps[NT - 1]);
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
                // the condition tokens' V' rows of this sample: requested here, they land while the exchange is waited for
                float4 cvp[N];
#pragma unroll
                for (int n = 0; n < N; ++n)
                    cvp[n] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_ct, (unsigned)lane * 16u, (unsigned)((n * SEEME_CROW + l * 512 + 256) * 4), 0));
                if constexpr (WIN) { __builtin_amdgcn_sched_barrier(0); clm_published(FLG + 2, lane, pub1); cl_issue_seq<WT, C, Q>(ring, wave, voff, wg, bc, bn, skip, nskip, typename SCH::W1A{}); }
                // ---- X1: gather v' (and y), all-reduce the scores
                const int pub = (4 * lane) / G::S, off = (4 * lane) % G::S;
                const unsigned o_v = (unsigned)((G::G_X1 + pub * G::X1_G + off) * 8);
                const unsigned o_s = (unsigned)((G::G_X1 + (lane % C) * G::X1_G + 2 * G::S) * 8);
                u32x4 gv0, gv1, gy0, gy1, gs[G::SC / 2];
                gy0 = gy1 = u32x4{0u, epoch, 0u, epoch};
                unsigned spins = 0;
                if (sentinel) {   // cheap wait first: one granule per (publisher, writing wave) -- the first of every 16-column tile of v' (and y) and the first
                    // score -- one 8-byte load per lane and round instead of the whole sweep (eight waves poll side by side on this CU)
                    constexpr int TPP = G::S / 16, NS1 = 2 * TPP + 1;               // sentinels per publisher
                    const int sp = lane / NS1, sq = lane - sp * NS1;
                    const bool has = lane < C * NS1 && (skip || sq < TPP || sq == 2 * TPP);
                    const unsigned so = (unsigned)((G::G_X1 + sp * G::X1_G + (sq < TPP ? sq * 16 : (sq < 2 * TPP ? G::S + (sq - TPP) * 16 : 2 * G::S))) * 8);
                    __builtin_amdgcn_s_sleep(DCLM_POLL_SLEEP);
                    while (!dead) {
                        const auto v2 = __builtin_amdgcn_raw_buffer_load_b64(xr_, has ? so : 0u, 0, 16);
                        if (__all(!has || (unsigned)v2[1] == epoch)) break;
                        if (++spins > DCL_SPIN_LIMIT) { dead = true; if (lane == 0) atomicOr(ka.hdr, 2u); break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                } else __builtin_amdgcn_s_sleep(DCLM_POLL_SLEEP);
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
                if constexpr (WIN) { cl_issue_seq<WT, C, Q>(ring, wave, voff, wg, bc, bn, skip, nskip, typename SCH::W1B{}); if (DCLM_IDLE_FLAGS && wave == 0 && lane == 0) cl_flag_set(FLG + 3, e1); }
                DEN_DBG(0);
                float sc[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    sc[j] = __uint_as_float((j & 1) ? gs[j >> 1].z : gs[j >> 1].x);
                    if (C >= 2) sc[j] += dpp_f(sc[j], 0);
                    if (C >= 4) sc[j] += dpp_f(sc[j], 1);
                    if (C >= 8) sc[j] += dpp_f(sc[j], 2);
                }
                float mx = sc[0];
#pragma unroll
                for (int j = 1; j < NT; ++j) mx = fmaxf(mx, sc[j]);
                float esum = 0.f;
#pragma unroll
                for (int j = 0; j < NT; ++j) { sc[j] = fast_exp(sc[j] - mx); esum += sc[j]; }
                const float inv = fast_rcp(esum);
                const float4 vv = make_float4(__uint_as_float(gv0.x), __uint_as_float(gv0.z), __uint_as_float(gv1.x), __uint_as_float(gv1.z));
                float4 att = f4_scale(vv, sc[0] * inv);
#pragma unroll
                for (int n = 0; n < N; ++n) att = f4_fma(sc[1 + n] * inv, cvp[n], att);
                att = f4_fma(sc[NT - 1] * inv, ld4(VP + M::O_TV + 4 * lane), att);
                if (skip) xr = make_float4(__uint_as_float(gy0.x), __uint_as_float(gy0.z), __uint_as_float(gy1.x), __uint_as_float(gy1.z));
                xr = wave_ln(f4_add(xr, att), VP + M::O_N1W, VP + M::O_N1B, lane);        // the "values" carry out_proj: residual + norm1
                clm_put4<WT, NARROW>(XB, es, lane, xr);
            } else if constexpr (WIN) {        // (waves without a sample: their eighth of both windows -- behind sample 0's publish and sweep, as in
                                               //  k_den_cluster: requested at once they would stand in front of the epilogue waves' granule stores)
                if (DCLM_IDLE_FLAGS) clm_wait_published(FLG + 2, pub1);
                cl_issue_seq<WT, C, Q>(ring, wave, voff, wg, bc, bn, skip, nskip, typename SCH::W1A{});
                if (DCLM_IDLE_FLAGS) cl_flag_wait(FLG + 3, e1);
                cl_issue_seq<WT, C, Q>(ring, wave, voff, wg, bc, bn, skip, nskip, typename SCH::W1B{});
            }
            __syncthreads(); DEN_DBG(0);

            // ================= B: linear1 + ReLU, column-split =================
            nxt.load(0, wave, lane, vp, Ln, tt_next, ln, A, b0, nact, c);
            if constexpr (!WIN) cl_refills<WT, C, Q, G::U_A + A_DEF0>(ring, wave, voff, wg, bc, bn, skip, nskip, std::make_integer_sequence<int, 2 * G::TA - A_DEF0>{});   // slots of stage A
            {
                f32x4 acc[G::TB];
                cl_zero<G::TB>(acc);
                cl_units<WT, C, Q, G::U_B, G::TB, G::TB, 0, 0, (WIN ? 0 : G::TB)>(ring, xb, acc, true, wave, voff, wg, bc, bn, skip, nskip, std::make_integer_sequence<int, G::TB>{});
#pragma unroll
                for (int t = 0; t < G::TB; ++t) {
                    const int jh = (wave * G::TB + t) * 16 + col;
                    const float bh = VP[M::O_L1B + jh];
#pragma unroll
                    for (int j = 0; j < SPL; ++j)
                        if (g2 + j < nact) clm_put1<WT, NARROW>(XH, g2 + j, jh, fmaxf(clm_out<WT, NARROW>(acc[t], j) + bh, 0.f));
                }
            }
            nxt.store(lane, STGN);
            __syncthreads(); DEN_DBG(0);

            // ================= C: linear2, row-split -> X2 =================
            if constexpr (Q) nxt.load(1, wave, lane, vp, Ln, tt_next, ln, A, b0, nact, c);
            {
                f32x4 acc[2];
                cl_zero<2>(acc);
                cl_units<WT, C, Q, G::U_C, G::TB, 2, 0, 0, C_INL>(ring, xh, acc, true, wave, voff, wg, bc, bn, skip, nskip, std::make_integer_sequence<int, G::TB>{});
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int n = (2 * wave + tt) * 16 + col;
#pragma unroll
                    for (int j = 0; j < SPL; ++j)
                        if (g2 + j < nact) cl_store_granule(xg0 + (size_t)(g2 + j) * G::G_TOTAL + G::G_X2 + c * 256 + n, e2, clm_out<WT, NARROW>(acc[tt], j), local);
                }
            }
            if constexpr (Q) nxt.store(lane, STGN);
            if (epi) {
                const unsigned epoch = e2;
                // the tabulated ca_block term of this (sample, step, layer): requested here, lands while the exchange is waited for
                float4 cadd = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (!Q) cadd = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_ca, (unsigned)lane * 16u, (unsigned)((step * SEEME_DEN_NL + l) * 1024), 0));
                if constexpr (WIN) { __builtin_amdgcn_sched_barrier(0); clm_published(FLG + 2, lane, pub2); cl_issue_seq<WT, C, Q>(ring, wave, voff, wg, bc, bn, skip, nskip, typename SCH::W2A{}); }
                float4 sum;
                unsigned spins = 0;
                if (sentinel) {   // cheap wait first: one granule per (publisher, writing wave): column 0 of the wave's first tile
                    const bool has = lane < 8 * C;
                    const unsigned so = (unsigned)((G::G_X2 + (lane >> 3) * 256 + (lane & 7) * 32) * 8);
                    __builtin_amdgcn_s_sleep(DCLM_POLL_SLEEP);
                    while (!dead) {
                        const auto v2 = __builtin_amdgcn_raw_buffer_load_b64(xr_, has ? so : 0u, 0, 16);
                        if (__all(!has || (unsigned)v2[1] == epoch)) break;
                        if (++spins > DCL_SPIN_LIMIT) { dead = true; if (lane == 0) atomicOr(ka.hdr, 4u); break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                } else __builtin_amdgcn_s_sleep(DCLM_POLL_SLEEP);
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
                if constexpr (WIN) { cl_issue_seq<WT, C, Q>(ring, wave, voff, wg, bc, bn, skip, nskip, typename SCH::W2B{}); if (DCLM_IDLE_FLAGS && wave == 0 && lane == 0) cl_flag_set(FLG + 3, e2); }
                DEN_DBG(0);
                // + bias, residual, norm2, + the tabulated ca_block term (one condition token: seeme_denoiser_ca_tables)
                xr = wave_ln(f4_add(xr, f4_add(sum, ld4(VP + M::O_L2B + 4 * lane))), VP + M::O_N2W, VP + M::O_N2B, lane);
                if constexpr (!Q) {
                    xr = f4_add(xr, cadd);
                    clm_put4<WT, NARROW>(XA, es, lane, xr);
                } else {
                    clm_put4<WT, NARROW>(XB, es, lane, wave_ln(xr, VP + M::O_CNW, VP + M::O_CNB, lane));   // ca_block.norm -> input of the query (mdiff_transformer.py:229)
                }
                st4(RES + es * 256 + 4 * lane, xr);
            } else if constexpr (WIN) {
                if (DCLM_IDLE_FLAGS) clm_wait_published(FLG + 2, pub2);
                cl_issue_seq<WT, C, Q>(ring, wave, voff, wg, bc, bn, skip, nskip, typename SCH::W2A{});
                if (DCLM_IDLE_FLAGS) cl_flag_wait(FLG + 3, e2);
                cl_issue_seq<WT, C, Q>(ring, wave, voff, wg, bc, bn, skip, nskip, typename SCH::W2B{});
            }
            __syncthreads(); DEN_DBG(0);

            if constexpr (Q) {
                // ================= G: ca_block.query, column-split by dims (waves < S / 16 own one tile each) -> X3 =================
                {
                    f32x4 acc[1];
                    cl_zero<1>(acc);
                    const bool act = wave < G::S / 16;
                    cl_units<WT, C, Q, G::U_G, 1, 1, 0, 0, 0>(ring, xb, acc, act, wave, voff, wg, bc, bn, skip, nskip, std::make_integer_sequence<int, 1>{});
                    if (act) {
                        const float bq = VP[M::O_CAQB + wave * 16 + col];
#pragma unroll
                        for (int j = 0; j < SPL; ++j)
                            if (g2 + j < nact) QS[(g2 + j) * G::S + wave * 16 + col] = clm_out<WT, NARROW>(acc[0], j) + bq;
                    }
                }
                __syncthreads(); DEN_DBG(0);
                if (epi) {
                    // split softmax of the query over the cluster (k_den_cluster, stage G / X3): this CU holds S dims of sample es
                    const unsigned epoch = e3;
                    float m = -INFINITY;
#pragma unroll
                    for (int dd = 0; dd < G::S; dd += 64) { const int d = dd + lane; if (d < G::S) m = fmaxf(m, QS[es * G::S + d]); }
                    m = wave_max(m);
                    float lsum = 0.f, tn[N];
#pragma unroll
                    for (int n = 0; n < N; ++n) tn[n] = 0.f;
#pragma unroll
                    for (int dd = 0; dd < G::S; dd += 64) {
                        const int d = dd + lane;
                        if (d < G::S) {
                            const float e = fast_exp(QS[es * G::S + d] - m);
                            lsum += e;
                            float kr[N], kmx = -INFINITY, ks = 0.f;
#pragma unroll
                            for (int n = 0; n < N; ++n) { kr[n] = SMP[(2 + n) * G::S + d]; kmx = fmaxf(kmx, kr[n]); }
#pragma unroll
                            for (int n = 0; n < N; ++n) { kr[n] = fast_exp(kr[n] - kmx); ks += kr[n]; }
                            const float rks = fast_rcp(ks);
#pragma unroll
                            for (int n = 0; n < N; ++n) tn[n] = fmaf(e, kr[n] * rks, tn[n]);
                        }
                    }
                    lsum = wave_sum(lsum);
#pragma unroll
                    for (int n = 0; n < N; ++n) tn[n] = wave_sum(tn[n]);
                    {
                        float pv = lane == 0 ? m : (lane == 1 ? lsum : 0.f);
#pragma unroll
                        for (int n = 0; n < N; ++n) pv = lane == 2 + n ? tn[n] : pv;
                        if (lane < 8) cl_store_granule(xg + G::G_X3 + c * 8 + lane, epoch, pv, local);
                    }
                    // the condition tokens' ca value rows of this sample: requested here, they land while the exchange is waited for
                    float4 cav[N];
#pragma unroll
                    for (int n = 0; n < N; ++n)
                        cav[n] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_ct, (unsigned)lane * 16u, (unsigned)((n * SEEME_CROW + 2560 + l * 512 + 256) * 4), 0));
                    __builtin_amdgcn_sched_barrier(0);
                    clm_published(FLG + 2, lane, pub3);
                    cl_issue_seq<WT, C, Q>(ring, wave, voff, wg, bc, bn, skip, nskip, typename SCH::WXA{});
                    __builtin_amdgcn_s_sleep(DCLM_POLL_SLEEP);
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
                    cl_issue_seq<WT, C, Q>(ring, wave, voff, wg, bc, bn, skip, nskip, typename SCH::WXB{});
                    if (DCLM_IDLE_FLAGS && wave == 0 && lane == 0) cl_flag_set(FLG + 3, e3);
                    DEN_DBG(0);
                    const float mc = __uint_as_float(g3[0].x), lc = __uint_as_float(g3[0].z);
                    float Mx = mc;
                    if (C >= 2) Mx = fmaxf(Mx, dpp_f(Mx, 0));
                    if (C >= 4) Mx = fmaxf(Mx, dpp_f(Mx, 1));
                    if (C >= 8) Mx = fmaxf(Mx, dpp_f(Mx, 2));
                    const float wc = fast_exp(mc - Mx);
                    float Lw = lc * wc;
                    if (C >= 2) Lw += dpp_f(Lw, 0);
                    if (C >= 4) Lw += dpp_f(Lw, 1);
                    if (C >= 8) Lw += dpp_f(Lw, 2);
                    const float rL = fast_rcp(Lw);
                    float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int n = 0; n < N; ++n) {
                        float t = __uint_as_float((n & 1) ? g3[1 + (n >> 1)].z : g3[1 + (n >> 1)].x) * wc;
                        if (C >= 2) t += dpp_f(t, 0);
                        if (C >= 4) t += dpp_f(t, 1);
                        if (C >= 8) t += dpp_f(t, 2);
                        y = f4_fma(t * rL, cav[n], y);                                              // (q k^T) v  (mdiff_transformer.py:236-237)
                    }
                    // StylizationBlock (mdiff_transformer.py:152-163): SiLU(LN(y) (1 + scale) + shift) -> proj_out.out_layers
                    const float4 hh = f4_adaln(wave_ln(y, VP + M::O_CSNW, VP + M::O_CSNB, lane), ld4(VP + M::O_CSC + 4 * lane), ld4(VP + M::O_CSH + 4 * lane));
                    clm_put4<WT, NARROW>(XB, es, lane, f4_silu(hh));
                } else {
                    if (DCLM_IDLE_FLAGS) clm_wait_published(FLG + 2, pub3);
                    cl_issue_seq<WT, C, Q>(ring, wave, voff, wg, bc, bn, skip, nskip, typename SCH::WXA{});
                    if (DCLM_IDLE_FLAGS) cl_flag_wait(FLG + 3, e3);
                    cl_issue_seq<WT, C, Q>(ring, wave, voff, wg, bc, bn, skip, nskip, typename SCH::WXB{});
                }
                __syncthreads(); DEN_DBG(0);
                // ================= H: ca_block.proj_out.out_layers + residual (replicated) =================
                nxt.load(2, wave, lane, vp, Ln, tt_next, ln, A, b0, nact, c);
                {
                    f32x4 acc[2];
                    cl_zero<2>(acc);
                    cl_units<WT, C, Q, G::U_H, 2, 2, 0, 0, 0>(ring, xb, acc, true, wave, voff, wg, bc, bn, skip, nskip, std::make_integer_sequence<int, 2>{});
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        const int n = (2 * wave + tt) * 16 + col;
                        const float bo = VP[M::O_CAOB + n];
#pragma unroll
                        for (int j = 0; j < SPL; ++j) {
                            const int sj = g2 + j;
                            if (sj < nact) {
                                const float x3 = RES[sj * 256 + n] + clm_out<WT, NARROW>(acc[tt], j) + bo;
                                RES[sj * 256 + n] = x3;
                                clm_put1<WT, NARROW>(XA, sj, n, x3);
                            }
                        }
                    }
                }
                nxt.store(lane, STGN);
                __syncthreads(); DEN_DBG(0);
            }

            // ================= D: ffn.linear1 + GELU (replicated) =================
            nxt.load(Q ? 3 : 1, wave, lane, vp, Ln, tt_next, ln, A, b0, nact, c);
            if constexpr (!WIN) cl_refills<WT, C, Q, G::U_C + C_INL>(ring, wave, voff, wg, bc, bn, skip, nskip, std::make_integer_sequence<int, G::TB - C_INL>{});       // slots of stage C
            {
                f32x4 acc[1];
                cl_zero<1>(acc);
                cl_units<WT, C, Q, G::U_D, 1, 1, 0, 0, (WIN ? 0 : 1)>(ring, xa, acc, true, wave, voff, wg, bc, bn, skip, nskip, std::make_integer_sequence<int, 1>{});
                const int jh = wave * 16 + col;
                const float bh = VP[M::O_F1B + jh];
#pragma unroll
                for (int j = 0; j < SPL; ++j)
                    if (g2 + j < nact) clm_put1<WT, NARROW>(XH, g2 + j, jh, fast_gelu(clm_out<WT, NARROW>(acc[0], j) + bh));
            }
            nxt.store(lane, STGN);
            __syncthreads(); DEN_DBG(0);
            // ================= E: ffn.linear2 -> LayerNorm, AdaLN, SiLU (replicated) =================
            {
                f32x4 acc[2];
                cl_zero<2>(acc);
                cl_units<WT, C, Q, G::U_E, 1, 2, 0, 0, (WIN ? 0 : 1)>(ring, xh, acc, true, wave, voff, wg, bc, bn, skip, nskip, std::make_integer_sequence<int, 1>{});
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int j = 0; j < SPL; ++j)
                        if (g2 + j < nact) PART[(g2 + j) * 256 + (2 * wave + tt) * 16 + col] = clm_out<WT, NARROW>(acc[tt], j);
            }
            __syncthreads(); DEN_DBG(0);
            if constexpr (WIN) cl_issue_seq<WT, C, Q>(ring, wave, voff, wg, bc, bn, skip, nskip, typename SCH::W3{});
            if (epi) {
                const float4 y2 = f4_add(ld4(PART + es * 256 + 4 * lane), ld4(VP + M::O_F2B + 4 * lane));
                const float4 hh = f4_adaln(wave_ln(y2, VP + M::O_FSNW, VP + M::O_FSNB, lane), ld4(VP + M::O_TSC + 4 * lane), ld4(VP + M::O_TSH + 4 * lane));
                clm_put4<WT, NARROW>(XB, es, lane, f4_silu(hh));
            }
            __syncthreads(); DEN_DBG(0);
            // ================= F: ffn.proj_out.out_layers + residual (replicated); writes the next layer's input =================
            nxt.load(Q ? 4 : 2, wave, lane, vp, Ln, tt_next, ln, A, b0, nact, c);
            {
                f32x4 acc[2];
                cl_zero<2>(acc);
                cl_units<WT, C, Q, G::U_F, 2, 2, 0, 0, (WIN ? 0 : 2)>(ring, xb, acc, true, wave, voff, wg, bc, bn, skip, nskip, std::make_integer_sequence<int, 2>{});
                if constexpr (WIN) cl_issue_seq<WT, C, Q>(ring, wave, voff, wg, bc, bn, skip, nskip, typename SCH::AF{});
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int n = (2 * wave + tt) * 16 + col;
                    const float bo = VP[M::O_FOB + n];
#pragma unroll
                    for (int j = 0; j < SPL; ++j) {
                        const int sj = g2 + j;
                        if (sj < nact) {
                            const float xn = RES[sj * 256 + n] + clm_out<WT, NARROW>(acc[tt], j) + bo;
                            RES[sj * 256 + n] = xn;
                            clm_put1<WT, NARROW>(XA, sj, n, xn);
                            if (l < 2) clm_put1<WT, NARROW>(SKF + M::XBUF * l, sj, n, xn);          // xs.append(x) (cross_attention.py:70-72)
                        }
                    }
                }
            }
            nxt.store(lane, STGN);
            if constexpr (!WIN) cl_refills<WT, C, Q, G::NU_REAL>(ring, wave, voff, wg, bc, bn, skip, nskip, std::make_integer_sequence<int, G::NU - G::NU_REAL>{});   // (phantom units)
            __syncthreads(); DEN_DBG(0);
            if (l + 1 < SEEME_DEN_NL) {
                if (epi && !nskip) xr = ld4(RES + es * 256 + 4 * lane);               // residual of the next layer's attention
            } else {
                // ---- stack norm -> model output (cross_attention.py:82-83), scheduler.step (mld.py:495-497)
                if (epi) {
                    float4 e = wave_ln(ld4(RES + es * 256 + 4 * lane), CONSTV + 256, CONSTV + 512, lane);
                    if (A.sched == SEEME_SCHED_NONE) {
                        st4(KEEP + es * 256 + 4 * lane, e);
                    } else {
                        const CF32 cf = coef_c + (size_t)step * 8;
                        const float c0 = cf[0], c1 = cf[1], c2 = cf[2], c3 = cf[3], c4 = cf[4], c5 = cf[5], clip = cf[6], ptype = cf[7];
                        float4 nz = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (A.noise != nullptr) nz = ld4(A.noise + ((size_t)step * A.B + bs) * 256 + 4 * lane);
                        const float4 lat = ld4(KEEP + es * 256 + 4 * lane);
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
                        st4(KEEP + es * 256 + 4 * lane, nl);
                        xr = f4_add(nl, ld4(CONSTV + 4 * lane));
                        clm_put4<WT, NARROW>(XA, es, lane, xr);
                        st4(RES + es * 256 + 4 * lane, xr);
                    }
                }
                __syncthreads(); DEN_DBG(0);
            }
            cur ^= 1;
        }
        row = row_next;
    }
    DEN_DBG(3);
    if (epi && c == 0) st4(A.out + (size_t)bs * 256 + 4 * lane, ld4(KEEP + es * 256 + 4 * lane));
#pragma unroll
    for (int s = 0; s < W::RU; ++s) asm volatile("" ::"v"(ring.r[s][0].x));
}

// clusters of a launch: ceil(B / spc) rounded up to a multiple of 8 (one per XCD group under either placement)
static int clm_clusters(int B, int spc) { return ((B + spc - 1) / spc + 7) / 8 * 8; }

extern "C" size_t seeme_den_cluster_ms_xchg_bytes(int B, int C, int spc) {      // (the two-token layout: the larger of the two)
    if (spc < 1 || spc > 8 || (C != 4 && C != 8)) return 0;
    const size_t per = C == 8 ? ClG<8, true>::G_TOTAL : ClG<4, true>::G_TOTAL;
    return DCL_HDR_BYTES + (size_t)clm_clusters(B, spc) * 8 * per * 8;
}

template <typename WT, int C, bool Q, bool NARROW>
static int launch_den_cluster_ms(const ClArgs& ka0, hipStream_t st) {
    ClArgs ka = ka0;
    ka.clusters = clm_clusters(ka.s.B, ka.spc);
    const int grid = ka.clusters * C;
    int dev = 0, cus = 0;
    SEEME_HIP(hipGetDevice(&dev));
    SEEME_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (grid > cus || grid > 256) return seeme_fail("denoiser_sample_cluster: clusters x C exceeds one workgroup per CU of this device");
    if (ka0.spc > ClM<WT, C, Q, NARROW>::MS) return seeme_fail("denoiser_sample_cluster: at most 8 (fp16) / 4 (bf16) samples per cluster");
    const size_t lds = (size_t)ClM<WT, C, Q, NARROW>::LDS_FLOATS * sizeof(float);
    if (lds > 160 * 1024 || lds <= 80 * 1024) return seeme_fail("denoiser_sample_cluster: LDS footprint must force one workgroup per CU");
    SEEME_HIP(hipFuncSetAttribute((const void*)k_den_cluster_ms<WT, C, Q, NARROW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    SEEME_HIP(hipMemsetAsync(ka.hdr, 0, seeme_den_cluster_ms_xchg_bytes(ka.s.B, C, ka.spc), st));
    hipLaunchKernelGGL((k_den_cluster_ms<WT, C, Q, NARROW>), dim3(grid), dim3(DEN_THREADS), lds, st, ka);
    return seeme_check_launch("k_den_cluster_ms");
}

// called by seeme_denoiser_sample_cluster when SeemeDenCluster.samples > 1
static int den_cluster_ms_dispatch(const SeemeDenoiserWeights* w, const SeemeDenCluster* cl, const SeemeSampleArgs* a, ClArgs& ka, hipStream_t st) {
    if (cl->wdtype != 2 && cl->wdtype != 1) return seeme_fail("denoiser_sample_cluster: several samples per cluster need a 16-bit weight image");
    if (cl->C != 4 && cl->C != 8) return seeme_fail("denoiser_sample_cluster: several samples per cluster need C = 4 or 8");
    if (cl->samples > 8) return seeme_fail("denoiser_sample_cluster: at most 8 samples per cluster");
    if (a->N != 1 && a->N != 2) return seeme_fail("denoiser_sample_cluster: several samples per cluster need one or two condition tokens");
    if (a->trow_per_sample) return seeme_fail("denoiser_sample_cluster: several samples per cluster share the step's table row");
    if (cl->xchg_bytes < seeme_den_cluster_ms_xchg_bytes(a->B, cl->C, cl->samples)) return seeme_fail("denoiser_sample_cluster: exchange buffer too small");
    ka.spc = cl->samples;
    if (cl->wdtype == 1) {      // bf16 image: four A rows per sample, up to 4 samples per cluster (4-CU clusters only)
        if (cl->C != 4) return seeme_fail("denoiser_sample_cluster: the bf16 image takes several samples per cluster at C = 4 only");
        return a->N == 2 ? launch_den_cluster_ms<WBF16, 4, true, false>(ka, st) : launch_den_cluster_ms<WBF16, 4, false, false>(ka, st);
    }
    if (cl->C == 4 && cl->samples <= 4)      // fp16, at most four samples per cluster: four A rows per sample, one sample per lane group
        return a->N == 2 ? launch_den_cluster_ms<WF16, 4, true, false>(ka, st) : launch_den_cluster_ms<WF16, 4, false, false>(ka, st);
    if (a->N == 2) return cl->C == 8 ? launch_den_cluster_ms<WF16, 8, true, true>(ka, st) : launch_den_cluster_ms<WF16, 4, true, true>(ka, st);
    return cl->C == 8 ? launch_den_cluster_ms<WF16, 8, false, true>(ka, st) : launch_den_cluster_ms<WF16, 4, false, true>(ka, st);
}
