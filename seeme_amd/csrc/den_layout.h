// den_layout.h -- layout of the packed denoiser weight image consumed by the sampling kernel.
//
// Single source of truth: seeme_den_layout() (exported from libseeme_hip.so) fills an int64 array
// with these offsets; seeme_amd/_pack.py asks the library for them and packs the state_dict
// accordingly.  Order = order of use inside one denoiser step, so the weight stream is sequential.
//
// wg offsets are in ELEMENTS of the weight dtype; a PyTorch [N,K] fp32 matrix is stored
// in "GEMV layout" [K/KV][N][KV], KV = 4; a bf16 / fp16 matrix as the per-wave matrix-core operand
// stream [wave 8][tile group][k-block K/32][tile in group][lane 64][8] (seeme_amd/mld_denoiser.py put_w).  vp offsets are in floats.
#pragma once
#include <stdint.h>

#define SEEME_DEN_NL 5

struct DenLayerOff {
    // ---- wg (matrices), in stream order
    int64_t skip;   // encoder.linear_blocks.{l-3}   [256,512]   (layers 3,4 only; -1 otherwise)
    int64_t inp;    // sa_block.self_attn.in_proj_weight [768,256]
    int64_t outp;   // sa_block.self_attn.out_proj.weight [256,256]
    int64_t l1;     // sa_block.linear1.weight [ff_sa,256]
    int64_t l2;     // sa_block.linear2.weight [256,ff_sa]
    int64_t caq;    // ca_block.query.weight [256,256]
    int64_t cao;    // ca_block.proj_out.out_layers.2.weight [256,256]
    int64_t f1;     // ffn.linear1.weight [ff,256]
    int64_t f2;     // ffn.linear2.weight [256,ff]
    int64_t fo;     // ffn.proj_out.out_layers.2.weight [256,256]
    // ---- vp (vectors)
    int64_t skip_b, in_b, out_b, n1w, n1b, l1b, l2b, n2w, n2b;
    int64_t cnw, cnb;       // ca_block.norm
    int64_t caq_b;
    int64_t csnw, csnb;     // ca_block.proj_out.norm
    int64_t cao_b;
    int64_t f1b, f2b;
    int64_t fsnw, fsnb;     // ffn.proj_out.norm
    int64_t fo_b;
};
#define SEEME_DEN_LAYER_FIELDS 30

struct DenLayout {
    DenLayerOff L[SEEME_DEN_NL];
    int64_t pe0, fnw, fnb;      // query_pos.pe[0], encoder.norm.{weight,bias}
    int64_t wg_total, vp_total;
};
#define SEEME_DEN_LAYOUT_FIELDS (SEEME_DEN_NL * SEEME_DEN_LAYER_FIELDS + 5)

static inline DenLayout seeme_make_den_layout(int ff_sa, int ff) {
    DenLayout lay;
    const int64_t D = 256;
    int64_t w = 0, v = 0;
    lay.pe0 = v; v += D;
    lay.fnw = v; v += D;
    lay.fnb = v; v += D;
    for (int l = 0; l < SEEME_DEN_NL; ++l) {
        DenLayerOff& o = lay.L[l];
        if (l >= 3) { o.skip = w; w += 2 * D * D; } else { o.skip = -1; }
        o.inp = w;  w += 3 * D * D;
        o.outp = w; w += D * D;
        o.l1 = w;   w += (int64_t)ff_sa * D;
        o.l2 = w;   w += (int64_t)ff_sa * D;
        o.caq = w;  w += D * D;
        o.cao = w;  w += D * D;
        o.f1 = w;   w += (int64_t)ff * D;
        o.f2 = w;   w += (int64_t)ff * D;
        o.fo = w;   w += D * D;
        o.skip_b = v; v += D;
        o.in_b = v;   v += 3 * D;
        o.out_b = v;  v += D;
        o.n1w = v; v += D;  o.n1b = v; v += D;
        o.l1b = v; v += ff_sa;
        o.l2b = v; v += D;
        o.n2w = v; v += D;  o.n2b = v; v += D;
        o.cnw = v; v += D;  o.cnb = v; v += D;
        o.caq_b = v; v += D;
        o.csnw = v; v += D; o.csnb = v; v += D;
        o.cao_b = v; v += D;
        o.f1b = v; v += ff;
        o.f2b = v; v += D;
        o.fsnw = v; v += D; o.fsnb = v; v += D;
        o.fo_b = v; v += D;
    }
    lay.wg_total = w;
    lay.vp_total = v;
    return lay;
}
