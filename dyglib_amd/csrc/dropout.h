// Dropout of the training path (models/DyGFormer.py:429, :456-460) and the dense activation set the backward pass reads.
// Masks are never stored: the forward kernels (dygformer_train.hip unfused, dygformer_fused3.hip fused) and the backward pass draw them
// from the same counter-based hash of (seed, site, element index); site = 4 * layer + {0: attention probabilities, 1: attention output,
// 2: FFN activation, 3: FFN output}, element index = the element's offset in the dense row-major activation of that site.
#pragma once
#include "common.h"

namespace dygnn {
namespace train {

// 32-bit integer hash with full avalanche (two multiplies: integer multiplies run at quarter rate on CDNA, and the fused forward draws
// ~340 masks per lane and layer beside its MFMAs; a 64-bit splitmix costs six times as much)
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
struct Drop {
    uint32_t key0, key1, thresh; float scale;         // keep iff hash >= thresh; kept values are multiplied by 1/(1-p)
    __device__ __forceinline__ uint32_t site_key(uint32_t site) const { return mix32(key0 + 0x9E3779B9u * (site + 1u)) ^ key1; }     // wave-uniform
    __device__ __forceinline__ float mask32(uint32_t skey, uint32_t idx) const {      // branch-free: thresh = 0 (p = 0) keeps everything at scale 1
        return mix32(idx ^ skey) >= thresh ? scale : 0.0f;
    }
    // element `idx` of site `site`; indices beyond 2^32 fold their high word in (identity below 2^32, where mask32 may be called directly)
    __device__ __forceinline__ float mask(uint32_t site, uint64_t idx) const {
        return mask32(site_key(site), (uint32_t)idx + 0x27d4eb2fU * (uint32_t)(idx >> 32));
    }
};
inline Drop make_drop(float p, uint64_t seed) {
    Drop d;
    d.key0 = (uint32_t)seed;
    d.key1 = (uint32_t)(seed >> 32) * 0x85EBCA6Bu + 0x165667B1u;
    d.thresh = p <= 0.f ? 0u : (uint32_t)((double)p * 4294967296.0);
    d.scale = p <= 0.f ? 1.0f : (float)(1.0 / (1.0 - (double)p));
    return d;
}

// What the fused training forward (k_dygformer_fused3<.., true>) leaves in HBM for the backward pass: dense rows b * T + token (T = tokens
// per pair of THIS call), the layout of dygformer_train.hip's Plan.
struct TrainOut {
    float* X[DYGNN_MAX_LAYERS + 1];            // [M][D]   layer inputs; X[NL] = encoder output
    struct L {
        float *xn0, *m0, *r0;                  // LN0(x) [M][D], its mean / rstd [M]
        float* qkv;                            // [M][3D]  q | k | v with bias, q unscaled
        float *P, *Pd;                         // [B*H][T][T] softmax, softmax o dropout mask
        float* oa;                             // [M][D]   attention output, heads concatenated (before the out-projection)
        float *x1, *xn1, *m1, *r1;             // residual after attention, LN1 of it, its statistics
        float *hpre, *hact;                    // [M][4D]  FFN hidden before GELU; after GELU and dropout
    } layer[DYGNN_MAX_LAYERS];
    float* pooled;                             // [2][B][D] per-side token means
    Drop dr;
};

}  // namespace train
}  // namespace dygnn
