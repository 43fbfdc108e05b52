/* seeme_hip.h -- C-ABI of libseeme_hip.so: MI355X (gfx950) kernels for the SEE-ME
 * motion-latent-diffusion hot path.
 *
 * The reference (L-Scofano/SEEME) is pure Python/PyTorch and has no FFI boundary of its own
 * (SURVEY.md F1); its plug-in boundary is instantiate_from_config(cfg) in mld/config.py:25-32.
 * This header is therefore the *build-side* boundary proposed in SURVEY.md section 8(b): plain
 * pointers and sizes, no torch types.  Each entry point names the reference code it replaces.
 * The Python classes in seeme_amd/ (same constructor kwargs / methods / state_dict keys as the
 * reference classes) are the only callers; see INTEGRATION.md for the ctypes binding.
 *
 * Conventions
 *   - all tensor arguments are DEVICE pointers to contiguous fp32 (unless stated), row-major;
 *   - every function enqueues work on `stream` (a hipStream_t passed as void*) and returns
 *     without synchronising; no allocation, no host sync inside (hipGraph-capturable);
 *   - return value 0 = ok, non-zero = error; seeme_last_error() gives a thread-local message;
 *   - workspaces are caller-allocated device buffers; *_workspace_bytes() gives the size.
 */
#ifndef SEEME_HIP_H
#define SEEME_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SEEME_D 256        /* latent_dim[-1]; every config of the reference uses 256 */
#define SEEME_NLAYERS 5    /* VAE: hard-coded mld_vae.py:51; denoiser: configs/modules/denoiser.yaml:6 */

enum { SEEME_ACT_NONE = 0, SEEME_ACT_RELU = 1, SEEME_ACT_GELU = 2, SEEME_ACT_SILU = 3 };
enum { SEEME_SCHED_NONE = 0, SEEME_SCHED_DDIM = 1, SEEME_SCHED_DDPM = 2 };

int seeme_version(void);
const char* seeme_last_error(void);

/* ------------------------------------------------------------------ generic fused linear
 * Y[M,N] = LN?( act( pre(A)[M,K] @ W[N,K]^T + bias ) + res )
 * Replaces torch.nn.functional.linear (+ fused neighbours) wherever the path uses it, e.g.
 * skel_embedding mld_vae.py:147, final_layer :251, skip Linear(2D,D) cross_attention.py:58-60
 * (A | A2 concatenation without materialising it), ResnetBlockFC respointnet.py:88-97.
 * K may be any size: W rows must be readable for roundup16(K) floats (zero padded) -- ldw says so. */
typedef struct {
    const float* A;  int lda;         /* [M, K1] */
    const float* A2; int lda2;        /* optional [M, K-K1] (concat along K); NULL if unused */
    int K1;
    const float* W;  int ldw;         /* [N, >=roundup16(K)] */
    const float* bias;                /* [N] or NULL */
    const float* res; int ldr;        /* optional residual [M,N] added after act */
    const float* ln_w; const float* ln_b;          /* optional LayerNorm over N (N must be <= 256) */
    const float* pre_ln_w; const float* pre_ln_b;  /* optional LayerNorm over K applied to A rows first */
    float* Y; int ldy;
    int M, N, K;
    int pre_act;                      /* activation applied to A (after pre-LN), SEEME_ACT_* */
    int act;                          /* activation applied to the product */
    float eps;
} SeemeLinearArgs;
int seeme_linear(const SeemeLinearArgs* args, void* stream);

/* ------------------------------------------------------------------ Transformer VAE
 * Weights of mld.models.architectures.mld_vae.MldVae (state_dict names in comments,
 * SURVEY.md App. A).  All fp32 device pointers in PyTorch layout ([out,in]). */
typedef struct {
    const float *in_w, *in_b;       /* self_attn.in_proj_weight [768,256], in_proj_bias */
    const float *out_w, *out_b;     /* self_attn.out_proj.{weight,bias} */
    const float *l1_w, *l1_b;       /* linear1 [ff,256] */
    const float *l2_w, *l2_b;       /* linear2 [256,ff] */
    const float *n1_w, *n1_b, *n2_w, *n2_b;
    /* decoder layers only (TransformerDecoderLayer, cross_attention.py:319-367); NULL in encoder */
    const float *ca_in_w, *ca_in_b; /* multihead_attn.in_proj_* */
    const float *ca_out_w, *ca_out_b;
    const float *n3_w, *n3_b;
} SeemeXfLayer;

typedef struct {
    SeemeXfLayer layer[SEEME_NLAYERS];   /* input_blocks.0, .1, middle_block, output_blocks.0, .1 */
    const float *skip_w[2], *skip_b[2];  /* linear_blocks.{0,1} [256,512] */
    const float *norm_w, *norm_b;        /* stack-final LayerNorm */
} SeemeSkipStack;

/* fp16 copies of the VAE matrices packed in MFMA fragment order (see SeemePointnetBf16 for the layout; N padded to
 * 16, K padded to 32): the throughput mode of the VAE (fp16 MFMA operands, fp32 accumulation / residual stream). */
typedef struct { const uint16_t *in_w, *out_w, *l1_w, *l2_w; } SeemeXfLayerH;
typedef struct {
    SeemeXfLayerH enc[SEEME_NLAYERS], dec[SEEME_NLAYERS];
    const uint16_t *enc_skip[2], *dec_skip[2];
    const uint16_t *emb_w, *fin_w, *ca_fold_w;
} SeemeVaeWeightsH;

typedef struct {
    int nfeats;                 /* F */
    int ff;                     /* 128 (hard-coded mld_vae.py:53) */
    const float* token;         /* global_motion_token [2,256] */
    const float* pe_enc;        /* query_pos_encoder.pe [500,1,256] */
    const float* pe_dec;        /* query_pos_decoder.pe */
    const float* emb_w; int emb_ldw; /* skel_embedding.weight [256, roundup16(F)] zero-padded copy */
    const float* emb_b;
    const float* fin_w;         /* final_layer.weight [F,256] */
    const float* fin_b;
    SeemeSkipStack enc, dec;
    /* decoder cross-attention to the single latent token, folded on the host:
     * c_l = ca_fold_w[l*256:(l+1)*256] z + ca_fold_b[...],  ca_fold_w = out_proj_l . W_v_l,  ca_fold_b = out_proj_l b_v_l + b_o_l */
    const float* ca_fold_w;     /* [5*256, 256] */
    const float* ca_fold_b;     /* [5*256] */
    const SeemeVaeWeightsH* h16; /* NULL: fp32-exact MFMA path (parity); else fp16-operand MFMA path */
} SeemeVaeWeights;

size_t seeme_vae_workspace_bytes(int B, int T);

/* MldVae.encode (mld_vae.py:128-193) up to the posterior parameters:
 * features [B,T,F], lengths [B] (int32, device) -> mu [B,256], logvar [B,256]
 * (the caller forms std = exp(logvar)^0.5 and the Normal; rsample is RNG, host side). */
int seeme_vae_encode(const SeemeVaeWeights* w, const float* features, const int32_t* lengths,
                     int B, int T, float* mu, float* logvar, void* workspace, size_t ws_bytes, void* stream);

/* MldVae.decode, arch encoder_decoder (mld_vae.py:195-256): z [B,256], lengths -> feats [B,T,F]
 * (padded frames not zeroed, as in the reference :253). */
int seeme_vae_decode(const SeemeVaeWeights* w, const float* z, const int32_t* lengths,
                     int B, int T, float* feats, void* workspace, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ denoiser + sampling loop
 * mld.models.architectures.mld_denoiser.MldDenoiser (arch trans_enc, MD_TRANS layers
 * mdiff_transformer.py:257-304) and MLD._diffusion_reverse (mld/models/modeltype/mld.py:432-511).
 *
 * Weight image: the host packs the state_dict once into
 *   wg : the matrices in stream order, element type fp32, bf16 or fp16 (wdtype 0 / 1 / 2); a [N,K] PyTorch
 *        matrix is stored as [K/4][N][4] (fp32, vector-ALU path) or as the per-wave matrix-core operand stream
 *        [wave 8][K/32][N/128][lane 64][8] (16-bit), so that every wave-load is 1 KiB contiguous;
 *   vp : fp32 vectors (biases, LayerNorm params, pe row 0);
 * in the order documented in seeme_amd/csrc/den_layout.h; seeme_den_layout() exports the offsets.
 * sa_fold = 1 (one attention head): the image's in_proj V rows / bias and kv_cat_w / kv_cat_b V rows hold
 * W_o W_v and W_o b_v + b_o instead of W_v, b_v, and the out_proj slot is unused (seeme_amd/mld_denoiser.py).
 */
typedef struct {
    const void*  wg;   int wdtype;       /* 0 = fp32, 1 = bf16, 2 = fp16 */
    const float* vp;
    const int64_t* layout;               /* device copy of the seeme_den_layout() table (155 int64) */
    int nhead;                           /* 1, 2 or 4 */
    int ff_sa;                           /* 1024 (hard-coded mdiff_transformer.py:279) */
    int ff;                              /* 128 */
    /* PyTorch-layout fp32 tensors used by the per-call table builders */
    const float* kv_cat_w; const float* kv_cat_b;       /* [5*512,256]: sa in_proj K|V rows per layer */
    const float* style_cat_w; const float* style_cat_b; /* [5*1024,256]: ca emb_layers.1 | ffn emb_layers.1 */
    const float* time_w1; const float* time_b1; const float* time_w2; const float* time_b2;
    const float* ca_kv_w[SEEME_NLAYERS]; const float* ca_kv_b[SEEME_NLAYERS];   /* [512,256] key|value per layer */
    const float* ca_tn_w[SEEME_NLAYERS]; const float* ca_tn_b[SEEME_NLAYERS];   /* ca_block.text_norm */
    /* key|value of all layers with the text_norm affine folded in: [5*512,256] = W_l diag(tn_w_l),
     * bias W_l tn_b_l + b_l; applied to the affine-free LayerNorm of the condition (ln_ones / ln_zeros) */
    const float* ca_fold_w; const float* ca_fold_b;
    const float* ln_ones; const float* ln_zeros;                                /* [256] each */
    int sa_fold;                                                                /* 1: out_proj folded into V (nhead == 1) */
    /* ca_block.proj_out of each layer (StylizationBlock, mdiff_transformer.py:152-163), for seeme_denoiser_ca_tables */
    const float* ca_pn_w; const float* ca_pn_b;   /* proj_out.norm of the 5 layers, [5,256] each */
    const float* ca_po_w; const float* ca_po_b;   /* proj_out.out_layers.2 of the 5 layers, [5,256,256] and [5,256] */
} SeemeDenoiserWeights;

/* per-row time tables: floats per row = 5*512 (sa K|V of the time token) + 5*1024 (AdaLN scale|shift, ca|ffn) */
#define SEEME_TROW 7680
/* per-sample condition tables: floats per (sample, token) = 5 layers * (sa K|V 512 + ca key|value 512) */
#define SEEME_CROW 5120

size_t seeme_denoiser_workspace_bytes(int n_rows, int B, int N);

/* Build the time tables for `n_rows` timestep feature rows tfeat [n_rows,256] (sinusoidal features,
 * tools/embeddings.py:245-285, computed by the host) -> ttab [n_rows, SEEME_TROW].
 * Batch-invariant in sampling. */
int seeme_denoiser_time_tables(const SeemeDenoiserWeights* w, const float* tfeat, int n_rows,
                               float* ttab, void* workspace, size_t ws_bytes, void* stream);

/* Build the per-sample condition tables for cond [Bc,N,256] (batch-first) -> ctab [Bc,N,SEEME_CROW]
 * (step-invariant K/V of the condition tokens). */
int seeme_denoiser_cond_tables(const SeemeDenoiserWeights* w, const float* cond, int Bc, int N,
                               float* ctab, void* workspace, size_t ws_bytes, void* stream);

/* ONE condition token (N == 1): the ca_block's contribution does not depend on the latent (the key softmax over
 * a single token is 1, mdiff_transformer.py:231-237), so it is tabulated per (sample, table row, layer):
 *   catab[bc][r][l] = proj_out.out_layers( SiLU( LN(value_l(cond_bc)) * (1 + scale_{row,l}) + shift_{row,l} ) )
 * with row = trow[r] (trow_per_sample 0, R = n_rows rows per sample) or trow[bc % n_b] (trow_per_sample 1, R = 1,
 * n_b = number of trow entries).  ctab [Bc,1,SEEME_CROW]; catab [Bc,R,5,256].
 * workspace >= 5*Bc*R*256 floats. */
int seeme_denoiser_ca_tables(const SeemeDenoiserWeights* w, const float* ctab, const float* ttab, const int32_t* trow,
                             int trow_per_sample, int n_trow, int Bc, float* catab,
                             void* workspace, size_t ws_bytes, void* stream);

typedef struct {
    int B;                 /* samples (latents rows) */
    int N;                 /* condition tokens per sample */
    int steps;             /* loop iterations (1 for a plain forward) */
    int sched;             /* SEEME_SCHED_* ; NONE: out = eps of the single step */
    int cfg;               /* 1: ctab holds 2B samples (uncond first, mld.py:489) and guidance is applied */
    float guidance_scale;
    const float* latents;  /* [B,256] initial */
    const float* ctab;     /* [B or 2B, N, SEEME_CROW] */
    const float* ttab;     /* [rows, SEEME_TROW] */
    const int32_t* trow;   /* sampling: [steps] row of ttab per step; forward: [B] row per sample */
    int trow_per_sample;   /* 0: trow[step], 1: trow[b] */
    const float* coef;     /* [steps,8] scheduler scalars per step (see seeme_amd/schedulers.py) */
    const float* noise;    /* optional [steps,B,256] step noise (eta>0 / DDPM), NULL otherwise */
    float* out;            /* [B,256] */
    const float* catab;    /* N == 1: [B or 2B, R, 5, 256] from seeme_denoiser_ca_tables (R = steps, or 1 with trow_per_sample) */
    float* save;           /* training forward only (steps == 1, no CFG, unfolded fp32 image, query GEMV kept): [B, SEEME_DEN_SAVE_FLOATS]
                            * intermediates for seeme_denoiser_backward (seeme_amd/csrc/den_train.h); NULL otherwise */
    int force_query;       /* 1: keep the ca_block query / proj_out GEMVs even for one condition token (differentiable path) */
    const unsigned char* drop;   /* training forward only (with save): dropout keep-masks [B, SEEME_DEN_DROP_BYTES] of the MD layers'
                                  * nn.Dropout sites in training mode (layout csrc/den_train.h DM_*), or NULL (eval arithmetic) */
    float drop_scale;      /* 1 / (1 - p) */
    int xcds;              /* seeme_denoiser_sample only: 0 or 8 = workgroups dealt to all XCDs; k in 1..7 = the working workgroups sit on
                            * k XCDs (blockIdx % 8 < k; assumes the round-robin dispatch of SPX mode -- any other placement only changes which
                            * L2s are shared, never results).  Fewer L2s re-fetching the 9 MB image each step: B = 32 on 2 XCDs pulls 962 MB per
                            * 50-step launch from the Infinity Cache instead of 3.72 GB.  A choice of the CALLER (it knows whether the launch has
                            * the chip to itself); the library keeps no state about streams. */
} SeemeSampleArgs;
#define SEEME_DEN_DROP_BYTES 10960

int seeme_denoiser_sample(const SeemeDenoiserWeights* w, const SeemeSampleArgs* a, void* stream);

/* ---- the same loop with ONE SAMPLE SPLIT OVER C WORKGROUPS (csrc/den_cluster.inc.hip; MLD._diffusion_reverse, mld.py:467-497, at
 * batch sizes that leave most of the chip idle: B x C <= 256).  One attention head, one or two condition tokens, no CFG.
 *   wgc : the cluster weight image [layer 5][CU C][unit][wave 8][load][lane 64][16 B], packed by the host
 *         (seeme_amd/mld_denoiser.py, geometry from seeme_den_cluster_layout); the skip linears of layers 3, 4 are folded
 *         into that layer's in_proj;
 *   vpc : the vector array of seeme_den_layout() order with in_b of layers 3, 4 = W_in' b_skip + b_in';
 *   xchg: exchange workspace (>= seeme_den_cluster_xchg_bytes(B, C), 16-byte aligned), zeroed by the call on `stream`.  Word 0
 *         of it is non-zero after the launch if a cluster gave up waiting for a peer (results invalid), word 1 counts the
 *         clusters that ran with L2-local granule stores.
 *   placement 0: the workgroups of a cluster have equal blockIdx % 8 (one XCD under round-robin dispatch), 1: consecutive
 *   blockIdx.  flags bit 0: granule stores always write-through.  Placement and flags change speed, never results. */
typedef struct {
    const void*  wgc;  int wdtype;       /* 0 = fp32, 1 = bf16, 2 = fp16 */
    const float* vpc;
    int C;                               /* workgroups (CUs) per sample: 2, 4 or 8 */
    int placement;
    int flags;
    void* xchg; size_t xchg_bytes;
    int query;                           /* 0: image packed for ONE condition token (ca term from seeme_denoiser_ca_tables); 1: for two
                                          * (units of ca_block.query / proj_out included, one more exchange per layer) */
    int samples;                         /* 0 / 1: one sample per cluster (k_den_cluster).  2..8: the large-batch form (k_den_cluster_ms,
                                          * csrc/den_cluster_ms.inc.hip): a cluster owns up to `samples` chains that share its weight stream
                                          * (two or four MFMA A rows per sample, wave s = epilogue wave of sample s): fp16 image (C = 4 or 8) or bf16
                                          * image (C = 4, at most 4 samples), one or two condition tokens, one table row per step; ceil(B / samples) rounded up to 8 clusters x C <= CUs;
                                          * xchg >= seeme_den_cluster_ms_xchg_bytes(B, C, samples) */
} SeemeDenCluster;
size_t seeme_den_cluster_xchg_bytes(int B, int C);
size_t seeme_den_cluster_ms_xchg_bytes(int B, int C, int samples);
/* out[0] units per (layer, CU) incl. padding, [1] bytes per unit, [2] image bytes, [3] wave-loads per unit, [4] k per wave-load,
 * [5..11] first unit of stage A (x half), A (skip half), B, C, D, E, F, [12] of G (ca query), [13] of H (ca proj_out) (-1 when query = 0) */
int seeme_den_cluster_layout(int C, int wdtype, int query, int64_t* out, int cap);
int seeme_denoiser_sample_cluster(const SeemeDenoiserWeights* w, const SeemeDenCluster* cl, const SeemeSampleArgs* a, void* stream);

/* ---- stage-2 training (MLD._diffusion_process, mld.py:582-631, and the backward of MldDenoiser.forward) ----
 * Forward = seeme_denoiser_sample with steps 1, SCHED_NONE, per-sample rows, force_query 1 and `save` set, on the
 * unfolded fp32 image.  seeme_den_train_pack refreshes that image, its transposed twin and the vector array from the
 * parameter tensors (mats: [5][10] device pointers in den_layout.h order, NULL where a layer has no skip linear).
 * seeme_denoiser_backward walks the chain in reverse (one workgroup per sample) and writes, per sample, x and dy of
 * every linear + LayerNorm parameter terms + d(first layer input) into gout [B, DB_TOTAL] (seeme_amd/csrc/den_train.h),
 * and the gradients of the tables into dctab [B,N,SEEME_CROW] / dttab [B,SEEME_TROW].  The batch reductions
 * (dW = sum_b dy_b x_b^T) are left to the host.  seeme_den_train_layout: [5][10] offsets of the transposed image,
 * its size, floats of `save` per sample, floats of `gout` per sample, floats per layer block of `gout`. */
int seeme_den_train_layout(int64_t* out, int cap);
int seeme_den_train_pack(const float* const* mats, float* img_f, float* img_b, const float* const* vec_src,
                         const int* vec_n, const int64_t* vec_dst, int n_vec, float* vp, void* stream);
int seeme_denoiser_backward(const SeemeDenoiserWeights* w, const void* img_bwd, int B, int N, const float* save,
                            const float* ctab, const float* ttab, const int32_t* trow, const float* dout,
                            float* gout, float* dctab, float* dttab, void* stream);
/* The same with the dropout keep-masks the forward was given (SeemeSampleArgs.drop / drop_scale) and the caller's XCD packing
 * (SeemeSampleArgs.xcds; seeme_denoiser_backward runs unpacked). */
int seeme_denoiser_backward_drop(const SeemeDenoiserWeights* w, const void* img_bwd, int B, int N, const float* save,
                                 const float* ctab, const float* ttab, const int32_t* trow, const float* dout,
                                 float* gout, float* dctab, float* dttab, const unsigned char* drop, float drop_scale, int xcds, void* stream);

/* Offsets of the packed weight image (30 per layer x 5, then pe0, fnw, fnb, wg_total, vp_total). */
int seeme_den_layout(int ff_sa, int ff, int64_t* out, int cap);

/* ------------------------------------------------------------------ SMPL linear blend skinning
 * smplx.SMPL.forward (call sites mld/models/modeltype/mld.py:764-770 ...). */
typedef struct {
    int V;                        /* 6890 */
    const float* v_template;      /* [V,3] (flat [V*3] = bias of the blend GEMM) */
    const float* blend_w;         /* [V*3, 224]: cols 0..9 shapedirs, 10..216 posedirs^T, 217..223 zero */
    const float* lbs_weights;     /* [V,24] */
    const float* J_template;      /* [24,3]  = J_regressor @ v_template      (precomputed) */
    const float* J_shapedirs;     /* [24,3,10] = J_regressor @ shapedirs */
    const int32_t* parents;       /* [24] kinematic tree, parents[0] = -1 */
    /* compact copy of the model restricted to the 21 extra-joint vertices (smplx VertexJointSelector) */
    const float* ex_template;     /* [21,3] */
    const float* ex_shapedirs;    /* [21*3,10] */
    const float* ex_posedirs;     /* [207, 63] */
    const float* ex_weights;      /* [21,24] */
} SeemeSmplModel;

size_t seeme_smpl_workspace_bytes(int M);

/* pose: axis-angle [M,72] (pose2rot=True) or rotation matrices [M,24,9]; betas [M,10]; transl [M,3] or NULL.
 * joints [M,45,3] (24 posed joints + 21 vertex-picked joints); vertices [M,V,3] or NULL (joints-only
 * fast path: no mesh is formed). */
int seeme_smpl_lbs(const SeemeSmplModel* m, const float* betas, const float* pose, int pose_is_rotmat,
                   const float* transl, int M, float* joints, float* vertices,
                   void* workspace, size_t ws_bytes, void* stream);

/* Gradient of the 24 posed joints (the first 24 of seeme_smpl_lbs's joints, axis-angle pose) w.r.t. pose [M,72] and transl
 * [M,3] (dtransl may be NULL): djoints [M, dj_stride >= 24, 3].  Replaces autograd through smplx's lbs for the joints loss of
 * train_vae_forward (mld.py:764-773,871-878). */
int seeme_smpl_joints_backward(const SeemeSmplModel* m, const float* betas, const float* pose, const float* djoints, int dj_stride,
                               float* dpose, float* dtransl, int M, void* stream);

/* ------------------------------------------------------------------ rotation helpers / renorm
 * mld/utils/geometry2.py: aa_to_quat :33-54, aa_to_rotmat :56-72, quat_to_rotmat :74-95,
 * rot6d_to_rotmat :98-117 ('prohmr' / 'diffusion' column order).  in [M,3|4|6] -> out [M,4] or [M,3,3]. */
enum { SEEME_GEO_AA_TO_QUAT = 0, SEEME_GEO_AA_TO_ROTMAT = 1, SEEME_GEO_QUAT_TO_ROTMAT = 2,
       SEEME_GEO_ROT6D_PROHMR = 3, SEEME_GEO_ROT6D_DIFFUSION = 4 };
int seeme_geometry(int op, const float* in, float* out, int M, void* stream);
/* EgoBodyDataModule.renorm (mld/data/EgoBody.py:151-157): y = x * std[:F] + mean[:F], x [rows,F]. */
int seeme_renorm(const float* x, const float* mean, const float* stdv, float* y, long rows, int F, void* stream);

/* ------------------------------------------------------------------ PointNet scene encoder
 * EgoHMR.models.respointnet.ResnetPointnet(out_dim, hidden 256) (respointnet.py:6-59); PyTorch-layout
 * fp32 weights, fc_pos_0.weight zero padded to [512,16]. */
typedef struct {
    int out_dim;                          /* 512 */
    const float* pos_w; const float* pos_b;          /* fc_pos_0 [512,16 padded], [512] */
    const float* fc0_w[4]; const float* fc0_b[4];    /* block_i.fc_0 [256,512] */
    const float* fc1_w[4]; const float* fc1_b[4];    /* block_i.fc_1 [256,256] */
    const float* sc_w[4];                            /* block_i.shortcut [256,512], no bias */
    const float* fcc_w; const float* fcc_b;          /* fc_c [out_dim,256] */
} SeemePointnetWeights;
size_t seeme_pointnet_workspace_bytes(int B, int P);
/* points [B,P,3] -> out [B,out_dim] */
int seeme_pointnet_encode(const SeemePointnetWeights* w, const float* points, int B, int P, float* out,
                          void* workspace, size_t ws_bytes, void* stream);

/* bf16-MFMA variant (fp32 accumulation): the same weights as bf16 copies packed in MFMA fragment order,
 * Wp[((t * (K/32) + k/32) * 64 + lane][8] = W[row(t, lane&15)][32*(k/32) + 8*(lane>>4) .. +7] for n-tile t, with
 * the rows of an n-tile interleaved so that a lane of the kernel owns 16 consecutive output features:
 * row(t, q) = 64*(t/4) + 16*(q/4) + 4*(t%4) + q%4.
 * One fused persistent kernel per ResnetBlockFC on 128-point tiles, max-pool folded into the epilogue.  Results
 * differ from the fp32 path by bf16 rounding of weights and activations (tolerance stated in the tests). */
typedef struct {
    const uint16_t* fc0[4];     /* block_i.fc_0.weight   [256,512] packed */
    const uint16_t* fc1[4];     /* block_i.fc_1.weight   [256,256] packed */
    const uint16_t* sc[4];      /* block_i.shortcut.weight [256,512] packed (sc[0] unused: see sc3) */
    const uint16_t* posf;       /* fc_pos_0 (weight [512,3], bias) as split-bf16 operands of v_mfma_f32_16x16x16_bf16,
                                 * w = hi + lo: [32 n-tiles][64 lanes][4], lane = 16*kq + q for column 16*t + q:
                                 * kq 0: whx why whz whx | kq 1: why whz wlx wly | kq 2: wlz bh bl 0 | kq 3: 0 */
    const float* sc3;           /* block_0.shortcut folded through fc_pos_0 (both linear, no bias: respointnet.py:35,84,93):
                                 * [256][4] fp32 = ( Ws Wp | Ws bp ) */
    /* Second-generation block kernels (csrc/pointnet_v2.hip; used when stream[0] != NULL): per block ONE weight stream of
     * 24 slots x 16 fragments x 64 lanes x 8 bf16, in the order a 256-point tile consumes it.  Fragment (feature tile nt,
     * k-block kb): lane 16*kq + m, element j = W[16 nt + m][32 kb + 16 (j/4) + 4 kq + j%4] -- the k order in which an
     * accumulator tile pair is the next B operand; the activations between blocks are stored in that order too.
     *   block_0:   slots 0..15 fc_0 k-block = slot, fragment nt; slot 16 + 4 g + p: fc_1 tiles 8 g + n, k-blocks 2 p + kbi
     *              at fragment 8 kbi + n.
     *   block_1-3: slots 0..7 fc_0[:, :256] k-block = slot, fragment nt; slot 8 + 8 g + kb: fragments 0..7 shortcut[:, :256]
     *              tiles 8 g + n, fragments 8..15 fc_1 tiles 8 g + n, both k-block kb. */
    const uint16_t* stream[4];
    const uint16_t* sc3f;       /* sc3 as split-bf16 fragments like posf: [16 n-tiles][64 lanes][4] */
} SeemePointnetBf16;
size_t seeme_pointnet_bf16_workspace_bytes(int B, int P);
int seeme_pointnet_encode_bf16(const SeemePointnetWeights* w, const SeemePointnetBf16* wb, const float* points,
                               int B, int P, float* out, void* workspace, size_t ws_bytes, void* stream);

/* Weight gradients of the denoiser chain in one launch: for every tile {int x_col, y_col, ldo, nn, kk, pad; int64 out_off}
 * out[out_off + n*ldo + k] = sum_b gout[b*ldg + y_col + n] * gout[b*ldg + x_col + k], n < nn <= 32, k < kk <= 256, where
 * gout is the per-sample buffer of seeme_denoiser_backward (x and dy of every linear, csrc/den_train.h).  Replaces the
 * autograd weight-gradient accumulation of loss.backward() through MldDenoiser.forward (mld.py:582-631). */
int seeme_den_wgrad(const float* gout, int ldg, int B, const void* tiles, int n_tiles, float* out, void* stream);
/* The chain's vector gradients (biases, LayerNorm weights / biases) in one launch: out[q] = sum_b gout[b*ldg + idx[q]], q < n,
 * and dpe_row0[c] = sum_b gout[b*ldg + dx0_col + c], c < 256 (query_pos.pe row 0; NULL to skip).  Same reference as above. */
int seeme_den_vecgrad(const float* gout, int ldg, int B, const int64_t* idx, int n, float* out, int dx0_col, float* dpe_row0,
                      void* stream);

/* AdamW step of a list of fp32 tensors in one launch, torch.optim.AdamW arithmetic (amsgrad off): replaces the
 * optimiser step Lightning runs after training_step (reference: mld/models/modeltype/base.py configure_optimizers,
 * train.py:127-149).  chunks: device array of {int tensor, int count, int64 offset}; params / grads / exp_avg /
 * exp_avg_sq: device arrays of per-tensor base pointers; step = the 1-based step count (bias corrections and the
 * other scalar factors are formed in double, as PyTorch forms them). */
int seeme_adamw_step(const void* chunks, int n_chunks, void* const* params, const void* const* grads, void* const* exp_avg,
                     void* const* exp_avg_sq, double lr, double beta1, double beta2, double eps, double weight_decay,
                     double step, void* stream);

/* The same update with the step count and the learning rate read from device memory (step_lr = {step, lr}, the caller
 * increments step_lr[0] on the stream before the launch): nothing in the launch depends on host state, so a captured
 * hipGraph of the whole training step (MLD.capture_training_step) advances correctly from replay to replay.  Replaces the
 * same optimiser step as seeme_adamw_step. */
int seeme_adamw_step_dev(const void* chunks, int n_chunks, void* const* params, const void* const* grads, void* const* exp_avg,
                         void* const* exp_avg_sq, const float* step_lr, double beta1, double beta2, double eps,
                         double weight_decay, void* stream);

/* ------------------------------------------------------------------ stage-2 training step: the work around the chain
 * Replaces, for MLD.train_diffusion_forward / _diffusion_process (mld/models/modeltype/mld.py:582-631,887-1017), the
 * PyTorch ops (and their autograd backward) that build the chain's inputs: posterior rsample (mld_vae.py:186-193),
 * scheduler.add_noise (:604-606), the sinusoidal timestep features + TimestepEmbedding MLP (tools/embeddings.py:207-285),
 * output_scene (mld.py:257-261) and MldDenoiser.forward's per-layer condition / time tables (mld_denoiser.py:150-256:
 * self-attention K|V of the condition and time tokens, text_norm + linear-attention key|value, AdaLN scale|shift rows). */
typedef struct {
    int B, N;                 /* samples, condition tokens per sample */
    const float* dist;        /* [2, dist_rows, 256]: row 0 mu, row 1 logvar (seeme_vae_encode_dist); rows [0,B) = target,
                                 rows [B,2B) = condition motion when eps_c is set */
    int dist_rows;
    const float* eps_z;       /* [B,256] rsample noise of the target */
    const float* eps_c;       /* [B,256] rsample noise of the condition latent, or NULL (no 'interactee' condition) */
    int slot_c;               /* token slot of the condition latent in cond */
    float* cond;              /* [B,N,256] condition tokens (slot_c written here) */
    const float* noise;       /* [B,256] */
    const int64_t* timesteps; /* [B] */
    const float* acp;         /* alphas_cumprod [num_train_timesteps] */
    const float* freq;        /* [128] exp(-ln(max_period) * j / (128 - freq_shift)) */
    int flip_sin_to_cos;
    float* latents;           /* [B,256] z */
    float* noisy;             /* [B,256] x_t */
    float* tfeat;             /* [B,256] */
} SeemeGlueRows;
int seeme_glue_rows(const SeemeGlueRows* a, void* stream);
/* F.layer_norm(x, (256,)) without affine: xhat [M,256], rstd [M]. */
int seeme_glue_ln(const float* x, float* xhat, float* rstd, int M, void* stream);

/* One problem of seeme_grouped_gemm:  C[i,j] (+)= sum_s sum_{k < seg_len[s]} A_s[i,k] * B_s[k,j]  (+ bias[j]) (* epilogue)
 * with A_s[i,k] = pro_a(a[s][i*a_rs + k*a_ks[s]]) and B_s[k,j] = pro_b(b[s][k*b_ks[s] + j*b_cs]); prologue modes: 0 none, 1 SiLU,
 * 2 ReLU, 3 affine v*p0[idx] + p1[idx] (idx = k for A, j for B).  epi 1: multiply by SiLU'(e0[i*e_ld + j]); epi 2: by alpha.
 * colsum (needs
 * a_rs == 1): colsum[i] (+)= sum_s sum_k a[s][i,k] -- the bias gradient that comes free with a weight gradient.  tile0 /
 * tiles_n: position of the problem's 64x64 tiles in the launch. */
typedef struct {
    const float* a[10];
    const float* b[10];
    int seg_len[10];
    long a_ks[10];
    long b_ks[10];
    int nseg;
    long a_rs, b_cs;
    float* c;
    long ldc;
    int M, N;
    int a_pro;
    const float* a_p0;
    const float* a_p1;
    int b_pro;
    const float* b_p0;
    const float* b_p1;
    const float* bias;
    int epi;
    const float* e0;
    long e_ld;
    int accumulate;
    float* colsum;
    int tile0, tiles_n;
    int nbatch;               /* > 1: nbatch independent problems of this shape; a[s] / b[s] / c advance by the strides below */
    long a_bstride, b_bstride, c_bstride;
    float alpha;              /* epi 2: C = alpha * (A B + bias) */
    const float* addend;      /* optional [M,N] (row stride add_ld) added after the epilogue */
    long add_ld;
} SeemeGemmProblem;              /* accumulate: 0 store, 1 C += (one writer), 2 atomicAdd (nbatch members share C: split reduction) */
int seeme_grouped_gemm(const SeemeGemmProblem* probs_dev, int n_probs, int n_tiles, void* stream);
int seeme_gemm_problem_bytes(void);
/* Plain large fp32 GEMM on the matrix cores (the projections of the stage-1 training step): C[M,N] = A[M,K] B + bias[N] + addend,
 * B = W[N,K] (b_is_nt: y = x W^T, F.linear) or W[K,N] (data gradient dx = dy W).  M, N multiples of 128, K of 32, operands 16-byte
 * aligned with strides in multiples of 4 floats; anything else goes through seeme_grouped_gemm. */
/* Large weight gradient: G[Nout,Kin] += dY[M,Nout]^T X[M,Kin], gbias[Nout] += column sums of dY (NULL to skip); Nout, Kin
 * multiples of 128; split over row chunks with atomic accumulation (G must hold the value to add to). */
int seeme_wgrad128(const float* dY, long ldy, const float* X, long ldx, int M, int Nout, int Kin, float* G, long ldg, float* gbias,
                   void* stream);
int seeme_gemm128(const float* A, long lda, const float* B, long ldb, int b_is_nt, float* C, long ldc, int M, int N, int K,
                  const float* bias, const float* addend, long add_ld, void* stream);

/* Element-wise middle of the backward: d cond = sum_l dcs[l] + LayerNorm-backward(sum_l dxl[l] * tn_w[l]); text_norm
 * affine gradients g_tn_w[l] += sum_m dxl[l]*xhat, g_tn_b[l] += sum_m dxl[l]; d emb = sum_5 dea + SiLU'(emb) * sum_10 deb. */
typedef struct {
    int M, B;
    const float* dxl;         /* [5,M,256] */
    const float* dcs;         /* [5,M,256] */
    const float* xhat;        /* [M,256] */
    const float* rstd;        /* [M] */
    const float* tn_w[5];
    float* g_tn_w[5];
    float* g_tn_b[5];
    float* dcond;             /* [M,256] */
    const float* dea;         /* [5,B,256] */
    const float* deb;         /* [10,B,256] */
    const float* emb;         /* [B,256] */
    float* demb;              /* [B,256] */
} SeemeGlueMid;
int seeme_glue_mid(const SeemeGlueMid* a, void* stream);

/* ------------------------------------------------------------------ stage-1 (VAE) training: row kernels
 * The hand-written forward-with-saves / backward of MldVae.encode / decode for train_vae_forward (mld.py:633-885; layers
 * cross_attention.py:41-147,281-367).  GEMMs run on seeme_grouped_gemm; call order in seeme_amd/vae_train.py. */
typedef struct {
    const float* sub;         /* [M,256] sublayer output, or one row per sequence when sub_seq_rows > 0 */
    const float* res;         /* [M,256] residual or NULL */
    const float* gamma; const float* beta;
    float* y; float* xhat; float* rstd;
    long M; int sub_seq_rows; float eps;
} SeemeVtLn;
int seeme_vt_add_ln(const SeemeVtLn* a, void* stream);        /* y = LN(sub + res); keeps xhat, rstd */
typedef struct {
    const float* dy; const float* xhat; const float* rstd; const float* gamma;
    float* dpre;              /* [M,256] gradient w.r.t. (sub + res) */
    float* dgamma; float* dbeta;   /* accumulated (atomics) */
    long M; int accumulate;   /* 1: dpre += */
    const float* dy2;         /* optional second gradient of the same output (skip connection): dy + dy2 */
} SeemeVtLnBwd;
int seeme_vt_ln_bwd(const SeemeVtLnBwd* a, void* stream);
/* scores [B,S,S] = q k^T (unscaled) -> softmax(scale * s) over the keys [0, min(S, n_prefix + lengths[b])), zeros elsewhere. */
int seeme_vt_softmax_fwd(float* scores, const int32_t* lengths, int B, int S, int n_prefix, float scale, void* stream);
/* dp [rows,S] (gradient w.r.t. the probabilities) -> gradient w.r.t. the unscaled scores, in place. */
int seeme_vt_softmax_bwd(float* dp, const float* p, long rows, int S, float scale, void* stream);
/* exact GELU: dh == NULL: out = gelu(pre); else out = dh * gelu'(pre). */
int seeme_vt_gelu(const float* pre, const float* dh, float* out, long n, void* stream);
/* out[b,:] (+)= sum_s w[b,s] d[b,s,:]  (d [B,S,256]); w = wmask[b,s] ? scale : 0, or 1 when wmask is NULL. */
int seeme_vt_seq_sum(const float* d, float* out, int B, int S, int accumulate, const unsigned char* wmask, float scale, void* stream);
/* Inverted dropout with a given keep-mask (nn.Dropout / the attention-weight dropout of nn.MultiheadAttention in training,
 * cross_attention.py:264-273,324-337): out = x * mask * scale, scale = 1 / (1 - p); in place allowed; x / out 16-byte aligned, mask 4-byte aligned. */
int seeme_vt_dropout(const float* x, const unsigned char* mask, float scale, float* out, long n, void* stream);
/* Decoder cross-attention to the single latent token under dropout (cross_attention.py:357-362):
 * out[b,s,:] = ((wmask[b,s] ? scale : 0) * cvn[b,:] + bo) * (m2[b,s,:] ? scale : 0). */
int seeme_vt_cross_rows(const float* cvn, const float* bo, const unsigned char* wmask, const unsigned char* m2, float scale,
                        int B, int S, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SEEME_HIP_H */
