// Row-block chains of the TGAT / TGN layer (tgat_chain.hip): the dependent small products of a layer run inside ONE workgroup that owns
// 16 or 32 rows, activations in LDS, weights streamed from L2 straight into MFMA operands, so a layer is three launches
// (k_tgat_pre -> attention -> k_tgat_post) instead of nine and its intermediates (q_in, q, att, fc, merge_in, hid) never reach HBM.
#pragma once
#include "common.h"

namespace dygnn {
namespace chain {

// whether a 16-row block of these dims fits the LDS (a property of the model's dims, never of the batch)
bool fits(int Fn, int Ft, int Dkv, int H);

// rows i < n (or < *n_live): q_in = [h(self) | cos(b)] -> q = W_q q_in -> qk[i][h][:] = W_k,h^T q_ih          (models/modules.py:150-170)
struct PreArgs {
    const float* h_lower;          // layer >= 2: rows of the level below ([.][Fn]); NULL: layer 1 reads node_feat[lower_ids[i]]
    const float* node_feat;
    const int32_t* lower_ids;
    const int32_t* lower_map;      // layer 2 over a de-duplicated level 1: entry -> row of h_lower (or NULL)
    const int32_t* n_live;         // device-side row count (or NULL: n)
    const float *tw, *tb;          // time encoder
    const float *query_w, *key_w;  // [Dq][Dq], [Dq][Dkv]
    float* qk;                     // [n][H][Dkv]
    int64_t n;
    int Fn, Ft, Dkv, H;
};
int launch_pre(hipStream_t s, const PreArgs& a);

// rows i: att = W_v,h z_ih -> residual_fc + q_in -> LayerNorm -> [. | raw] -> relu(fc1) -> fc2 -> out[i][:]   (models/modules.py:186-199, :42-68)
struct PostArgs {
    const float* z;                // [n][H][Dkv]
    const float* h_lower;
    const float* node_feat;
    const int32_t* lower_ids;
    const int32_t* lower_map;
    const int32_t* n_live;
    const float *tw, *tb;
    const float *value_w, *res_w, *res_b, *ln_w, *ln_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
    float* out;                    // [n][Fn]
    int64_t n;
    int Fn, Ft, Dkv, H;
};
int launch_post(hipStream_t s, const PostArgs& a);

// TGN: nn.GRUCell over the listed nodes (row r = node list[r], r < *count <= max_rows): message and memory rows gathered, both gate
// products, gates, Mnew[node] and feat0[node] = Mnew[node] + raw[node] scattered -- one launch (MemoryModel.py:462-500)
struct GruArgs {
    const int32_t *list, *count;
    const float *msg, *M, *raw;
    const float *w_ih, *w_hh, *b_ih, *b_hh;
    float *Mnew, *feat0;
    int64_t max_rows;
    int Dm, Fn;
};
int launch_gru(hipStream_t s, const GruArgs& a);

}  // namespace chain
}  // namespace dygnn
