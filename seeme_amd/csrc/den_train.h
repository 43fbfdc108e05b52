// den_train.h -- buffers exchanged between the training forward (k_den_sample with SeemeSampleArgs.save), the
// backward kernel (k_den_bwd) and the host (seeme_amd/denoiser_train.py).  Offsets in floats, per sample.
#pragma once

// ---- saved by the forward, per layer
#define DT_X      0        // layer input x (after the skip linear)                        256
#define DT_QKV    256      // q | k0 | v0 of token 0                                       768
#define DT_P      1024     // attention probabilities: [0] self, [1..N] condition, [N+1] time      8
#define DT_A      1032     // attention output                                             256
#define DT_XH1    1288     // xhat of norm1                                                256
#define DT_H      1544     // relu(linear1)                                                1024
#define DT_XH2    2568     // xhat of norm2                                                256
#define DT_XHC    2824     // xhat of ca_block.norm                                        256
#define DT_QC     3080     // softmax(query)                                               256
#define DT_XHY    3336     // xhat of ca_block.proj_out.norm                               256
#define DT_U      3592     // AdaLN output (pre-SiLU) of the ca_block                      256
#define DT_X3     3848     // stream after the ca_block residual                           256
#define DT_Z1     4104     // ffn.linear1 pre-activation                                   128
#define DT_XHY2   4232     // xhat of ffn.proj_out.norm                                    256
#define DT_U2     4488     // AdaLN output (pre-SiLU) of the ffn                           256
#define DT_X4     4744     // layer output                                                 256
#define DT_RS     5000     // rstd of norm1, norm2, ca.norm, ca.proj_out.norm, ffn.proj_out.norm (5); q.k dots at +8 (4)   16
#define DT_LAYER  5016
#define DT_FIN    (5 * DT_LAYER)          // xhat of encoder.norm (256), its rstd at +256
#define DT_TOTAL  (5 * DT_LAYER + 264)

// ---- written by the backward, per layer: inputs X and output gradients dY of every linear (dW = dY^T X summed
// over the batch, db = sum dY), and the per-sample LayerNorm parameter gradients (dy*xhat | dy)
#define DB_X_INP   0       // x                256      dY: DB_Y_INP  768  (dq | dk0 | dv0)
#define DB_X_OUTP  256     // a                256          DB_Y_OUTP 256
#define DB_X_L1    512     // x1               256          DB_Y_L1   1024
#define DB_X_L2    768     // h                1024         DB_Y_L2   256
#define DB_X_CAQ   1792    // ca.norm(x2)      256          DB_Y_CAQ  256
#define DB_X_CAO   2048    // silu(u)          256          DB_Y_CAO  256
#define DB_X_F1    2304    // x3               256          DB_Y_F1   128
#define DB_X_F2    2560    // gelu(z1)         128          DB_Y_F2   256
#define DB_X_FO    2688    // silu(u2)         256          DB_Y_FO   256
#define DB_X_SKIP  2944    // [x_prev | skip]  512          DB_Y_SKIP 256      (layers 3, 4; zero otherwise)
#define DB_Y_INP   3456
#define DB_Y_OUTP  4224
#define DB_Y_L1    4480
#define DB_Y_L2    5504
#define DB_Y_CAQ   5760
#define DB_Y_CAO   6016
#define DB_Y_F1    6272
#define DB_Y_F2    6400
#define DB_Y_FO    6656
#define DB_Y_SKIP  6912
#define DB_LN      7168    // 5 LayerNorms x (dw 256 | db 256): norm1, norm2, ca.norm, ca.proj_out.norm, ffn.proj_out.norm
#define DB_LAYER   9728
#define DB_FIN     (5 * DB_LAYER)         // encoder.norm: dw 256 | db 256
#define DB_DX0     (5 * DB_LAYER + 512)   // gradient of the first layer's input (sample + query_pos)
#define DB_TOTAL   (5 * DB_LAYER + 768)

// ---- dropout keep-masks of the training forward / backward (bytes, per sample and layer; nn.Dropout sites of the reference's
// MD layer in training mode: mdiff_transformer.py:137-165 (StylizationBlock.out_layers), :241-254 (FFN), cross_attention.py:264-273
// (sa_block: attention weights of token 0's row, dropout1, the FFN's inner dropout, dropout2))
#define DM_P      0        // attention weights of token 0: [0] self, [1..N] condition, [N+1] time      8
#define DM_1      8        // sa_block.dropout1 (out_proj output)                          256
#define DM_H      264      // sa_block.dropout (relu(linear1))                             1024
#define DM_2      1288     // sa_block.dropout2 (linear2 output)                           256
#define DM_C      1544     // ca_block.proj_out.out_layers dropout (silu(u))               256
#define DM_F      1800     // ffn.dropout (gelu(linear1))                                  128
#define DM_O      1928     // ffn.proj_out.out_layers dropout (silu(u2))                   256
#define DM_LAYER  2192
#define DM_TOTAL  (5 * DM_LAYER)
