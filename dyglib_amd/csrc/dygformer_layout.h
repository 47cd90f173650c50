// Packed-weight and workspace layouts shared by the generic and the fused DyGFormer paths.
#pragma once
#include "common.h"

namespace dygnn {

struct Dims {               // derived sizes (host side)
    int Fn, Fe, Ft, C, D, P, NL, H, hd, L;
    int Tside;              // max tokens per side = ceil(L / P)      (models/DyGFormer.py:223-226)
    int Tmax;               // token stride of the activation buffers = 2 * Tside
    int Smax;               // max padded positions per side = Tside * P
    int lut_rows;           // co-occurrence LUT rows: counts 0 .. 2*Smax
};

inline Dims make_dims(const dygnn_dygformer_config& c) {
    Dims d;
    d.Fn = c.node_feat_dim; d.Fe = c.edge_feat_dim; d.Ft = c.time_feat_dim; d.C = c.channel_embedding_dim;
    d.D = 4 * d.C; d.P = c.patch_size; d.NL = c.num_layers; d.H = c.num_heads; d.hd = d.H > 0 ? d.D / d.H : 0;
    d.L = c.max_input_sequence_length;
    d.Tside = d.P > 0 ? (d.L + d.P - 1) / d.P : 0;
    d.Tmax = 2 * d.Tside;
    d.Smax = d.Tside * d.P;
    d.lut_rows = d.Smax + 1;      // a count is at most the row length
    return d;
}

inline int check_config(const dygnn_dygformer_config* c) {
    DYGNN_REQUIRE(c != nullptr, "config is NULL");
    DYGNN_REQUIRE(c->node_feat_dim > 0 && c->edge_feat_dim > 0 && c->time_feat_dim > 0 && c->channel_embedding_dim > 0,
                  "config: feature dims must be positive");
    DYGNN_REQUIRE(c->patch_size > 0, "config: patch_size must be positive");
    DYGNN_REQUIRE(c->num_layers >= 1 && c->num_layers <= DYGNN_MAX_LAYERS, "config: num_layers must be in [1,%d]", DYGNN_MAX_LAYERS);
    DYGNN_REQUIRE(c->num_heads >= 1 && (4 * c->channel_embedding_dim) % c->num_heads == 0,
                  "config: embed_dim must be divisible by num_heads");                       // nn.MultiheadAttention assert
    // models/DyGFormer.py:209
    DYGNN_REQUIRE(c->max_input_sequence_length - 1 > 0, "Maximal number of neighbors for each node should be greater than 1!");
    return DYGNN_OK;
}

// ---- packed weights (float offsets) -----------------------------------------------------------
struct PackedLayout {
    size_t lut;                       // [lut_rows][C]      f(c) = W1 relu(W0 c + b0) + b1   (DyGFormer.py:332-335)
    size_t projT[4];                  // [K_ch][C]          transposed projection weights (node, edge, time, cooc)
    size_t inT[DYGNN_MAX_LAYERS];     // [D][3D]
    size_t outT[DYGNN_MAX_LAYERS];    // [D][D]
    size_t f0T[DYGNN_MAX_LAYERS];     // [D][4D]
    size_t f1T[DYGNN_MAX_LAYERS];     // [4D][D]
    size_t outputT;                   // [D][Fn]
    size_t fused3;                    // start of the token-owner fused kernel's section (one fragment stream)
    size_t total;                     // floats
};

size_t fused3_packed_floats(const Dims& d);   // defined in dygformer_fused3.hip

inline PackedLayout make_packed_layout(const Dims& d) {
    PackedLayout p;
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o += (n + 63) & ~size_t(63); return r; };   // 256-byte aligned sections
    p.lut = take((size_t)d.lut_rows * d.C);
    const int K[4] = {d.P * d.Fn, d.P * d.Fe, d.P * d.Ft, d.P * d.C};
    for (int c = 0; c < 4; ++c) p.projT[c] = take((size_t)K[c] * d.C);
    for (int l = 0; l < d.NL; ++l) {
        p.inT[l] = take((size_t)d.D * 3 * d.D);
        p.outT[l] = take((size_t)d.D * d.D);
        p.f0T[l] = take((size_t)d.D * 4 * d.D);
        p.f1T[l] = take((size_t)4 * d.D * d.D);
    }
    p.outputT = take((size_t)d.D * d.Fn);
    p.fused3 = o;
    o += (fused3_packed_floats(d) + 63) & ~size_t(63);
    p.total = o;
    return p;
}

// device-visible per-group sizes (workspace.dims)
struct CallDims { int32_t maxw_s, maxw_d, S_s, S_d, T_s, T_d, T, pad; };

// ---- workspace (byte offsets) -------------------------------------------------------------------
struct WorkspaceLayout {
    size_t dims;        // CallDims[groups]: max_window_src, max_window_dst, S_s, S_d, T_s, T_d, T, -
    size_t hist_len;    // int32[2B]
    size_t end_pos;     // int64[2B]
    size_t X, Xn, QKV, Hid;   // generic path activations, token stride Tmax
    size_t total;
};

inline WorkspaceLayout make_workspace_layout(const Dims& d, int64_t B) {
    WorkspaceLayout w;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 255) & ~size_t(255); return r; };
    w.dims = take((size_t)(B > 0 ? B : 1) * sizeof(CallDims));      // one CallDims per group, at most one group per pair
    w.hist_len = take((size_t)2 * B * sizeof(int32_t));
    w.end_pos = take((size_t)2 * B * sizeof(int64_t));
    const size_t rows = (size_t)B * d.Tmax;
    w.X = take(rows * d.D * sizeof(float));
    w.Xn = take(rows * d.D * sizeof(float));
    w.QKV = take(rows * 3 * d.D * sizeof(float));
    w.Hid = take(rows * 4 * d.D * sizeof(float));
    w.total = o;
    return w;
}

}  // namespace dygnn
