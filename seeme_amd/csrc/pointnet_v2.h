// Arguments of the second-generation PointNet block kernel (pointnet_v2.hip), launched from seeme_pointnet_encode_bf16.
#pragma once
#include <hip/hip_runtime.h>

struct PnBlock2Args {
    const float* points; const uint2* posf; const uint2* sc3f;   // block_0: points [B,P,3]; fc_pos_0 / folded shortcut as split-bf16 fragments
    const unsigned short* x;                // later blocks: activations in fragment order, [B][Ppad/16][8][64][8] bf16 (pointnet_v2.hip)
    const uint4* stream;                    // the block's weight stream, [24][16][64] x 16 B (SeemePointnetBf16.stream)
    const float* b0; const float* v0;       // fc_0 bias; pooled half [B,256] (NULL in block_0)
    const float* b1; const float* vs;       // fc_1 bias; pooled half of the shortcut [B,256] (NULL in block_0)
    unsigned short* out;                    // block output, same layout (NULL for the last block)
    float* pool;                            // [B,256] running max of the block output (pre-initialised to -inf)
    int P, Ppad, tiles_x, n_tiles;          // Ppad = P rounded up to 16
};
int seeme_pn_block2_launch(bool first, const PnBlock2Args& a, int n_cu, hipStream_t st);
int seeme_pn_block2_tile_points();   // points per tile of the build (32 per wave)
