// Row-block chains of the TGAT / TGN layer (tgat_chain.hip): the dependent small products of a layer run inside ONE workgroup that owns
// 4 .. 32 rows, activations in LDS, weights streamed from the cache hierarchy straight into MFMA operands, so a layer is three launches
// (k_tgat_pre -> attention -> k_tgat_post) instead of nine and its intermediates (q_in, q, att, fc, merge_in, hid) never reach HBM.
#pragma once
#include "common.h"

namespace dygnn {
namespace chain {

// whether a 16-row block of these dims fits the LDS (a property of the model's dims, never of the batch)
bool fits(int Fn, int Ft, int Dkv, int H);

// The weights are re-laid per call into MFMA-operand fragments (1 KiB = the float4 of each of the 64 lanes, zero-padded tiles) in the
// order the waves consume them: a wave's weight load is then one contiguous KiB instead of 16 strided 64-byte pieces.  (The call
// takes raw state_dict pointers and knows nothing about their history, so the 2 MB per layer are packed every call: one launch.)
struct LayerPack { uint32_t q, k, v, r, f1, f2; };        // fragment offsets of the six products of a layer
struct PackPlan {
    LayerPack layer[DYGNN_MAX_LAYERS];
    uint32_t ih, hh;                                       // GRU (TGN only)
    uint32_t total;                                        // fragments
};
PackPlan plan_pack(int L, int Fn, int Ft, int Dkv, int H, int gru_Dm /* 0: no GRU */);
inline size_t pack_bytes(const PackPlan& p) { return (size_t)p.total * 1024; }
// TGN: the lists of a call's level-0 nodes, built in the same launch as the packing (independent work, one launch less).  Every level-0
// slot s = (entry q, position j: 0..k-1 a neighbour, k the entry itself) has written owner[id] = s (plain stores: one of a node's slots
// wins); the winning slot lists its node: list (pending message: GRU rows) or list2 (none: feat0 = memory + raw) and records which in
// pendf[id].  Slots count inside their workgroup, one global atomic per workgroup and list.
struct ListArgs {
    const int32_t* ids0;           // level-0 ids: [n entries | n*k neighbours]
    const int32_t* n_live;         // device-side entry count (or NULL: n)
    const int32_t *owner, *has_msg;
    int32_t *pendf, *count, *list, *count2, *list2;
    int64_t n, N;
    int k;
};
int pack(hipStream_t s, const PackPlan& p, int L, int Fn, int Ft, int Dkv, int H, const dygnn_tgat_weights* w, const dygnn_gru_weights* gru, int gru_Dm,
         float* dst, const ListArgs* lists = nullptr);

// rows i < n (or < *n_live): q_in = [h(self) | cos(b)] -> q = W_q q_in -> qk[i][h][:] = W_k,h^T q_ih          (models/modules.py:150-170)
struct PreArgs {
    const float* h_lower;          // layer >= 2: rows of the level below ([.][Fn]); NULL: layer 1 reads node_feat[lower_ids[i]]
    const float* node_feat;
    const int32_t* lower_ids;
    const int32_t* lower_map;      // layer 2 over a de-duplicated level 1: entry -> row of h_lower (or NULL)
    const int32_t* n_live;         // device-side row count (or NULL: n)
    const float *tw, *tb;          // time encoder
    const float* pk;               // packed weights
    uint32_t off_q, off_k;
    float* qk;                     // [n][H][Dkv]
    int64_t n;
    int Fn, Ft, Dkv, H;
};
int launch_pre(hipStream_t s, const PreArgs& a);

// rows i: att = W_v,h z_ih -> residual_fc + q_in -> LayerNorm -> [. | raw] -> relu(fc1) -> fc2 -> out[i][:]   (models/modules.py:186-199, :42-68)
// With `qk` given (and post_fuses_attention()) the kernel also runs the attention over its rows' k neighbours itself (tgat_attn.h) and z never
// leaves LDS; otherwise it reads z [n][H][Dkv] written by the attention kernel.
bool post_fuses_attention(int64_t n, int Fn, int Ft, int Dkv, int H, int k);
struct PostArgs {
    const float* z;                // [n][H][Dkv] (NULL when the attention is fused)
    const float* h_lower;
    const float* node_feat;
    const int32_t* lower_ids;
    const int32_t* lower_map;
    const int32_t* n_live;
    const float *tw, *tb;
    const float* pk;
    uint32_t off_v, off_r, off_f1, off_f2;
    const float *res_b, *ln_w, *ln_b, *fc1_b, *fc2_b;
    float* out;                    // [n][Fn]
    int64_t n;
    int Fn, Ft, Dkv, H;
    unsigned long long* stamps;    // diagnostic: s_memtime of wave 0 at the stage boundaries, 16 per workgroup (NULL: off)
    // fused attention (or qk = NULL): W_k^T q rows [n][H][Dkv], edge table, the level's neighbour edge ids / time differences [n][k]
    const float *qk, *edge_feat;
    const int32_t* nbr_eid;
    const float* nbr_dt;
    int k, Fe;
    float scale;
};
int launch_post(hipStream_t s, const PostArgs& a);

// TGN: nn.GRUCell over the listed nodes (row r = node list[r], r < *count <= max_rows): message and memory rows gathered, both gate
// products, gates, Mnew[node] and feat0[node] = Mnew[node] + raw[node] scattered (MemoryModel.py:462-500); and, in the same launch,
// feat0[node] = M[node] + raw[node] for the nodes of list2 (the call's nodes without a pending message)
struct GruArgs {
    const int32_t *list, *count, *list2, *count2;
    const float *msg, *M, *raw;
    const float* pk;
    uint32_t off_ih, off_hh;
    const float *b_ih, *b_hh;
    float *Mnew, *feat0;
    int64_t max_rows;
    int Dm, Fn;
};
int launch_gru(hipStream_t s, const GruArgs& a);

}  // namespace chain
}  // namespace dygnn
