/*
 * dygnn.h — C ABI of libdygnn_hip.so: MI355X (gfx950) kernels for DyGLib's
 * temporal-neighbour-aggregation hot path.
 *
 * The reference (webster-781/DyGLib) is pure Python and has no FFI: its boundary for this
 * path is the duck-typed Python interface of utils/utils.py:71-302 (NeighborSampler) and
 * models/DyGFormer.py:11-317 (DyGFormer).  This header is what a binding for that path
 * binds; every entry point cites the reference code it replaces.  INTEGRATION.md shows the
 * ctypes stub and the two-line change to train_link_prediction.py.
 *
 * Conventions
 *   - plain C: pointers + sizes, no torch / HIP types in the signatures (dygnn_stream_t is a
 *     hipStream_t passed as void*; NULL = the null stream);
 *   - every pointer is a DEVICE pointer unless the parameter name ends in _host;
 *   - no allocation and no host synchronisation inside any *device* entry point: callers pass
 *     workspaces sized by the *_bytes() helpers, so calls can be captured in a hipGraph;
 *   - return value: 0 = ok, <0 = error (DYGNN_E_*); dygnn_last_error() returns a message for
 *     the calling thread.  Invalid arguments map to the reference's AssertionError sites,
 *     e.g. k <= 0 (utils/utils.py:157) or max_input_sequence_length <= 1 (DyGFormer.py:209).
 */
#ifndef DYGNN_H
#define DYGNN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DYGNN_OK            0
#define DYGNN_E_INVALID    -1   /* bad argument (reference: AssertionError / ValueError)   */
#define DYGNN_E_HIP        -2   /* HIP runtime error (launch failure, bad pointer ...)     */
#define DYGNN_E_UNSUPPORTED -3  /* shape outside what the kernels were built for           */
#define DYGNN_E_WORKSPACE  -4   /* caller workspace too small                              */

#define DYGNN_MAX_LAYERS 8

typedef void* dygnn_stream_t;   /* hipStream_t */

const char* dygnn_last_error(void);
/* ABI version of this header; bumped on any signature change. */
int dygnn_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * Temporal CSR: the time-sorted adjacency of NeighborSampler.__init__ (utils/utils.py:73-110)
 * built from an interaction list by get_neighbor_sampler (utils/utils.py:283-302).
 * Row r = node id r; row 0 = padding node, empty.  Inside a row entries ascend in time, ties
 * in edge-list order (stable sort, utils/utils.py:98-100).  Undirected: each interaction is
 * stored under both endpoints (utils/utils.py:298-300), src entry first.
 * ---------------------------------------------------------------------------------------- */
typedef struct dygnn_csr {
    int64_t        num_nodes;    /* rows = max node id + 1                                  */
    int64_t        num_entries;  /* 2 * number of interactions                              */
    const int64_t* indptr;       /* [num_nodes + 1]                                         */
    const int32_t* nbr;          /* [num_entries] neighbour node id                         */
    const int32_t* eid;          /* [num_entries] edge id                                   */
    const double*  ts;           /* [num_entries] interaction time (float64, as stored)     */
} dygnn_csr;

/* Host-side builder (C++, no GPU needed).  Inputs: the four parallel arrays of `Data`
 * (utils/DataLoader.py:46-64).  Outputs: caller-allocated host arrays of the sizes above with
 * num_nodes = max(src,dst)+1.  Replaces utils/utils.py:283-302 + :96-103. */
int dygnn_csr_build_host(int64_t num_edges, const int64_t* src_host, const int64_t* dst_host,
                         const int64_t* eid_host, const double* ts_host, int64_t num_nodes,
                         int64_t* indptr_host, int32_t* nbr_host, int32_t* eid_out_host, double* ts_out_host);

/* `uniform` neighbour sampling, host side (utils/utils.py:176-199): row r draws k positions in [0, hist_len[r]) exactly
 * as `RandomState.choice(a=hist_len[r], size=k)` of numpy's legacy MT19937 generator does (masked rejection on 32-bit outputs; rows with
 * hist_len <= 1 consume nothing and get zeros).  key[624] / *pos are the generator's state, taken from and written back to the sampler's
 * numpy RandomState by the caller, so the stream stays numpy's.  Host pointers; no GPU work. */
int dygnn_mt19937_choice_rows_host(uint32_t* key_host, int32_t* pos_host, const int32_t* hist_len_host, int64_t n, int32_t k,
                                   int32_t* sampled_host);

/* find_neighbors_before for n queries (utils/utils.py:130-147): hist_len[q] = i =
 * searchsorted(times[node], t, side='left') and end_pos[q] = indptr[node] + i (absolute CSR
 * index one past the last strictly-earlier interaction).  Either output may be NULL. */
int dygnn_find_neighbors_before(const dygnn_csr* csr_host, const int64_t* nodes, const double* times, int64_t n,
                                int32_t* hist_len, int64_t* end_pos, dygnn_stream_t stream);

/* get_historical_neighbors, strategy 'recent' (utils/utils.py:149-214, branch :200-209):
 * most recent k strictly-earlier interactions, RIGHT-aligned, zero-filled at the front.
 * Outputs [n,k]: int64 ids, int64 edge ids, float32 times (dtypes of utils/utils.py:161-167). */
int dygnn_sample_recent(const dygnn_csr* csr_host, const int64_t* nodes, const double* times, int64_t n, int32_t k,
                        int64_t* out_nbr, int64_t* out_eid, float* out_ts, dygnn_stream_t stream);

/* get_historical_neighbors, strategies 'uniform' / 'time_interval_aware' (utils/utils.py:176-199): the draws come
 * from the host (numpy RandomState replay, so they are bit-identical to the reference's), this entry point gathers
 * them: sel [n,k] int32 holds, per query, the positions INSIDE the node's time-sorted row (0 = oldest interaction) in
 * final output order, or -1 for "no neighbour" (rows of nodes without history stay zero, utils/utils.py:161-167).
 * Outputs [n,k]: int64 ids, int64 edge ids, float32 times. */
int dygnn_gather_selected(const dygnn_csr* csr_host, const int64_t* nodes, const int32_t* sel, int64_t n, int32_t k,
                          int64_t* out_nbr, int64_t* out_eid, float* out_ts, dygnn_stream_t stream);

/* DyGFormer window, phase 1 (get_all_first_hop_neighbors utils/utils.py:254-273 + the length
 * scan of pad_sequences models/DyGFormer.py:210-220): per query hist_len / end_pos as above and
 * *max_window = max_q min(hist_len[q], L-1) (device int32, overwritten).  The padded length is
 * S = roundup(*max_window + 1, patch_size) (models/DyGFormer.py:223-226). */
int dygnn_window_lengths(const dygnn_csr* csr_host, const int64_t* nodes, const double* times, int64_t n,
                         int32_t max_input_sequence_length, int32_t* hist_len, int64_t* end_pos,
                         int32_t* max_window, dygnn_stream_t stream);

/* DyGFormer window, phase 2 (pad_sequences models/DyGFormer.py:228-245): [n,S] LEFT-aligned
 * rows: col 0 = (node, edge 0, float32(t)); cols 1..m = the most recent m = min(hist_len, L-1)
 * interactions oldest->newest; zeros after.  int64 / int64 / float32 as :230-232. */
int dygnn_window_fill(const dygnn_csr* csr_host, const int64_t* nodes, const double* times, int64_t n,
                      int32_t max_input_sequence_length, int32_t S, const int32_t* hist_len, const int64_t* end_pos,
                      int64_t* out_ids, int64_t* out_eids, float* out_ts, dygnn_stream_t stream);

/* NeighborCooccurrenceEncoder.count_nodes_appearances (models/DyGFormer.py:337-393):
 * src_ids [n,S_s], dst_ids [n,S_d] int64 -> cnt_src [n,S_s,2], cnt_dst [n,S_d,2] float32 holding
 * [count in src row, count in dst row]; positions with id 0 give [0,0]. */
int dygnn_cooccurrence(const int64_t* src_ids, const int64_t* dst_ids, int64_t n, int32_t S_s, int32_t S_d,
                       float* cnt_src, float* cnt_dst, dygnn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * DyGFormer.compute_src_dst_node_temporal_embeddings (models/DyGFormer.py:68-194), eval mode.
 * ---------------------------------------------------------------------------------------- */
typedef struct dygnn_dygformer_config {
    int32_t node_feat_dim;              /* F_n (172)                         DyGFormer.py:36 */
    int32_t edge_feat_dim;              /* F_e (172)                         DyGFormer.py:37 */
    int32_t time_feat_dim;              /* F_t (100)                         DyGFormer.py:38 */
    int32_t channel_embedding_dim;      /* C (50); model dim D = 4C          DyGFormer.py:39 */
    int32_t patch_size;                 /* P                                 DyGFormer.py:40 */
    int32_t num_layers;                 /* <= DYGNN_MAX_LAYERS               DyGFormer.py:41 */
    int32_t num_heads;                  /* H, D % H == 0                     DyGFormer.py:42 */
    int32_t max_input_sequence_length;  /* L                                 DyGFormer.py:44 */
} dygnn_dygformer_config;

/* Raw parameter pointers in PyTorch layout (row-major [out,in]); names = state_dict keys
 * (SURVEY.md Appendix A). */
typedef struct dygnn_encoder_layer_weights {
    const float *in_proj_weight, *in_proj_bias;      /* [3D,D],[3D]  multi_head_attention     */
    const float *out_proj_weight, *out_proj_bias;    /* [D,D],[D]                             */
    const float *ffn0_weight, *ffn0_bias;            /* [4D,D],[4D]  linear_layers.0          */
    const float *ffn1_weight, *ffn1_bias;            /* [D,4D],[D]   linear_layers.1          */
    const float *norm0_weight, *norm0_bias;          /* [D]          norm_layers.0            */
    const float *norm1_weight, *norm1_bias;          /* [D]          norm_layers.1            */
} dygnn_encoder_layer_weights;

typedef struct dygnn_dygformer_weights {
    const float *time_w, *time_b;                    /* [F_t,1],[F_t] time_encoder.w          */
    const float *cooc_w0, *cooc_b0;                  /* [C,1],[C]    ..encode_layer.0         */
    const float *cooc_w1, *cooc_b1;                  /* [C,C],[C]    ..encode_layer.2         */
    const float *proj_node_w, *proj_node_b;          /* [C,P*F_n],[C]                         */
    const float *proj_edge_w, *proj_edge_b;          /* [C,P*F_e],[C]                         */
    const float *proj_time_w, *proj_time_b;          /* [C,P*F_t],[C]                         */
    const float *proj_cooc_w, *proj_cooc_b;          /* [C,P*C],[C]                           */
    dygnn_encoder_layer_weights layers[DYGNN_MAX_LAYERS];
    const float *output_w, *output_b;                /* [F_n,D],[F_n] output_layer            */
} dygnn_dygformer_weights;

/* Optional stage taps for parity tests (any member may be NULL).  Row-major, token stride
 * T_max = 2*ceil(L/P):  encoder_input / layer_out[l] are [B, T_max, D]. */
typedef struct dygnn_dygformer_taps {
    int32_t* seq_lens;                               /* [2]: S_src, S_dst of group 0          */
    float*   encoder_input;
    float*   layer_out[DYGNN_MAX_LAYERS];
    uint64_t* phase_cycles;                          /* diagnostic builds (-DDYGNN_STAMPS) only:
                                                        [4 workgroups][8 waves][32] s_memtime stamps */
    void*    ev_kernel_start;                        /* hipEvent_t (or NULL): recorded on `stream` immediately before / after the  */
    void*    ev_kernel_stop;                         /* launch of the dominant kernel of the call (the fused forward), so a caller  */
                                                     /* can time THAT kernel, not the call (bench.py's roofline line)               */
} dygnn_dygformer_taps;

/* Kernel-ready copy of the weights (transposed / MFMA-fragment order, co-occurrence LUT).
 * Re-pack after every optimizer step; packing is one cheap launch sequence. */
size_t dygnn_dygformer_packed_bytes(const dygnn_dygformer_config* cfg_host);
int dygnn_dygformer_pack(const dygnn_dygformer_config* cfg_host, const dygnn_dygformer_weights* w_host,
                         void* packed, size_t packed_bytes, dygnn_stream_t stream);
/* The weights changed IN PLACE (an optimizer step: same device addresses as at the last dygnn_dygformer_pack into `packed`):
 * refresh the copy with kernel launches only — no host work, no synchronisation (train_link_prediction.py:257 runs once per step).
 * fused_only != 0 refreshes just what the training entry points read (the fragment streams: two launches); the co-occurrence table and
 * the transposed copies of the inference paths are then stale until a call with fused_only = 0. */
int dygnn_dygformer_repack(const dygnn_dygformer_config* cfg_host, const dygnn_dygformer_weights* w_host,
                           void* packed, size_t packed_bytes, int32_t fused_only, dygnn_stream_t stream);

size_t dygnn_dygformer_workspace_bytes(const dygnn_dygformer_config* cfg_host, int64_t batch);
/* the same for one implementation choice (`impl` of dygnn_dygformer_forward): the fused kernels need only the per-query search
 * results, the generic path also its activation buffers; dygnn_dygformer_workspace_bytes is the size that serves every impl */
size_t dygnn_dygformer_workspace_bytes_for(const dygnn_dygformer_config* cfg_host, int64_t batch, int32_t impl);

/* One launch sequence for `batch` (src,dst,t) pairs.  group_size = G splits the pairs into consecutive
 * groups of G that are padded independently (each group has its own S_src/S_dst, models/DyGFormer.py:219-226):
 * the result of every group is bit-identical to a separate reference call on that group, so several
 * reference calls (e.g. the positive and the negative call of a step, evaluate_models_utils.py:126-136, or
 * several evaluation batches) run as ONE grid that keeps all 256 CUs busy.  G = 0 or G >= batch: one group.
 * pair_stride (caller-side fusion): 0, or batch / 2 when the batch is [positive calls ; negative calls] of the same
 * edges — pair i + pair_stride has the source and time of pair i (train_link_prediction.py:165-166, evaluate_models_utils.py:62-63).
 * The fused kernel then puts both pairs of an edge in one workgroup and gathers / projects the shared source side once; every row
 * is still bit-identical to the separate reference calls, and pairs whose (src, t) differ simply take the plain path.
 * impl: 0 = auto (the fused MFMA kernel when the shape is supported, else generic),
 *       1 = generic multi-kernel path (any shape),
 *       3 = fused kernel, token-owner layout (<= 128 tokens per pair; error if unsupported);
 *       2 (the first fused kernel, superseded) was removed in ABI 11 and is an error.
 * Node ids: the reference trusts them (an out-of-range id is an IndexError from numpy).  The host mirror raises
 * that IndexError for numpy inputs and validates the tables once per sampler (every CSR neighbour id < rows of node_feat,
 * every CSR edge id < rows of edge_feat); the kernels themselves never fault on a bad QUERY id: ids outside
 * [0, csr.num_nodes) are treated as the padding node 0 (empty history, zero feature row). */
int dygnn_dygformer_forward(const dygnn_dygformer_config* cfg_host, const dygnn_dygformer_weights* w_host,
                            const void* packed, const dygnn_csr* csr_host,
                            const float* node_feat, const float* edge_feat,
                            const int64_t* src, const int64_t* dst, const double* times, int64_t batch,
                            int64_t group_size, int64_t pair_stride, float* out_src, float* out_dst,
                            void* workspace, size_t workspace_bytes,
                            const dygnn_dygformer_taps* taps_host, int32_t impl, dygnn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * TGAT.compute_src_dst_node_temporal_embeddings (models/TGAT.py:48-136), eval mode, `recent`
 * sampling: L temporal-attention layers (MultiHeadAttention models/modules.py:99-206, mask =
 * neighbour id == 0 -> -1e10, post-LayerNorm) each followed by MergeLayer([out | raw node feat])
 * (models/TGAT.py:134).  Hop-(l+1) queries use the float32 neighbour times (models/TGAT.py:107-110).
 * ---------------------------------------------------------------------------------------- */
typedef struct dygnn_tgat_config {
    int32_t node_feat_dim, edge_feat_dim, time_feat_dim;   /* 172, 172, 100; each a multiple of 4   */
    int32_t num_layers;                                    /* 1..3                  models/TGAT.py:28 */
    int32_t num_heads;                                     /* (F_n+F_t) % H == 0    modules.py:120   */
    int32_t num_neighbors;                                 /* k <= 64               models/TGAT.py:49 */
} dygnn_tgat_config;

typedef struct dygnn_tgat_layer_weights {                  /* state_dict keys, PyTorch [out,in] layout */
    const float *query_w;                                  /* temporal_conv_layers.l.query_projection.weight [Dq,Dq]  */
    const float *key_w, *value_w;                          /* ...key_projection / value_projection.weight    [Dq,Dkv] */
    const float *ln_w, *ln_b;                              /* ...layer_norm.{weight,bias}                     [Dq]     */
    const float *res_w, *res_b;                            /* ...residual_fc.{weight,bias}              [Dq,Dq],[Dq]   */
    const float *fc1_w, *fc1_b;                            /* merge_layers.l.fc1                 [F_n,Dq+F_n],[F_n]    */
    const float *fc2_w, *fc2_b;                            /* merge_layers.l.fc2                 [F_n,F_n],[F_n]       */
} dygnn_tgat_layer_weights;

typedef struct dygnn_tgat_weights {
    const float *time_w, *time_b;                          /* time_encoder.w                                           */
    dygnn_tgat_layer_weights layers[DYGNN_MAX_LAYERS];
} dygnn_tgat_weights;

size_t dygnn_tgat_workspace_bytes(const dygnn_tgat_config* cfg_host, int64_t batch);
int dygnn_tgat_forward(const dygnn_tgat_config* cfg_host, const dygnn_tgat_weights* w_host, const dygnn_csr* csr_host,
                       const float* node_feat, const float* edge_feat,
                       const int64_t* src, const int64_t* dst, const double* times, int64_t batch,
                       float* out_src, float* out_dst, void* workspace, size_t workspace_bytes, dygnn_stream_t stream);

/* The same forward for a LIST of (node, time) roots: out [n_roots][node_feat_dim] = TGAT.compute_node_temporal_embeddings(ids, times,
 * num_layers) (models/TGAT.py:66-136), every root with its own time (`recent` sampling).  n_roots must be even; workspace as for
 * dygnn_tgat_forward with batch = n_roots / 2.  The evaluation step of evaluate_models_utils.py:126-136 needs the embeddings of
 * [sources ; destinations ; negative destinations] at the batch times: its negative call repeats the sources of its positive call
 * (:62-63), so one call on 3 B roots replaces two calls on 2 B roots each. */
int dygnn_tgat_forward_roots(const dygnn_tgat_config* cfg_host, const dygnn_tgat_weights* w_host, const dygnn_csr* csr_host,
                             const float* node_feat, const float* edge_feat, const int64_t* ids, const double* times, int64_t n_roots,
                             float* out, void* workspace, size_t workspace_bytes, dygnn_stream_t stream);

/* The same forward on PRE-SAMPLED neighbours, for the random strategies (uniform / time_interval_aware, e.g. the reference's
 * best TGAT configuration on Reddit, utils/load_configs.py:83-84): the draws must come from the sampler's numpy RandomState
 * in the reference's recursion order (models/TGAT.py:92-110), so the host builds the level sets and this entry point runs
 * everything else.  Level L = the 2*batch query nodes [src ; dst]; level l-1 = [level-l entries ; their k sampled neighbours
 * (row-major)].  For l = 1..L: nbr_eid[l] / nbr_dt[l] are [n_l, k] (edge id; float32(t_entry - float32(t_neighbour)),
 * models/TGAT.py:116-119); ids[l] for l = 0..L are the node ids of the level entries (int32).  All device pointers. */
typedef struct dygnn_tgat_levels {
    const int32_t* ids[DYGNN_MAX_LAYERS + 1];
    const int32_t* nbr_eid[DYGNN_MAX_LAYERS + 1];          /* [0] unused */
    const float* nbr_dt[DYGNN_MAX_LAYERS + 1];             /* [0] unused */
} dygnn_tgat_levels;
int dygnn_tgat_forward_levels(const dygnn_tgat_config* cfg_host, const dygnn_tgat_weights* w_host, const dygnn_tgat_levels* levels_host,
                              const float* node_feat, const float* edge_feat, int64_t batch,
                              float* out_src, float* out_dst, void* workspace, size_t workspace_bytes, dygnn_stream_t stream);

/* Diagnostic (synchronises `stream`): how many (node, time) entries the LAST dygnn_tgat_forward call on `workspace` (same cfg / batch)
 * had over its computed levels (*total = sum of the level sizes n_1 .. n_L, what the reference computes, models/TGAT.py:92-110) and how
 * many the library computed (*computed): with `recent` sampling a two-layer model computes every distinct entry of level 1 once. */
int dygnn_tgat_level_entries(const dygnn_tgat_config* cfg_host, int64_t batch, const void* workspace, int64_t* total_host, int64_t* computed_host,
                             dygnn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * TGN: MemoryModel.compute_src_dst_node_temporal_embeddings with model_name == 'TGN'
 * (models/MemoryModel.py:87-168): GRU memory update from the last pending raw message of every node
 * (MessageAggregator :267-300, GRUMemoryUpdater :490-501 = nn.GRUCell(2F_n+F_t+F_e, F_n)), temporal graph
 * attention over (memory + raw) node features (GraphAttentionEmbedding :548-664, same layer as TGAT), and —
 * for positive edges — memory persistence + new raw messages (:142-161, :212-251).
 * State lives in caller-owned device buffers and is MUTATED by positive calls; batches must be issued in
 * chronological order on one stream (the reference's strict batch-sequential semantics).
 * ---------------------------------------------------------------------------------------- */
typedef struct dygnn_gru_weights {          /* memory_updater.memory_updater.{weight_ih,weight_hh,bias_ih,bias_hh}           */
    const float *weight_ih, *weight_hh;     /* [3F_n, 2F_n+F_t+F_e], [3F_n, F_n]   (gate order r | z | n, nn.GRUCell)         */
    const float *bias_ih, *bias_hh;         /* [3F_n], [3F_n]                                                                  */
} dygnn_gru_weights;

typedef struct dygnn_tgn_state {
    int64_t  num_nodes;                     /* rows of node_raw_features (max node id + 1)                                     */
    float*   memory;                        /* [N, F_n]  memory_bank.node_memories                                             */
    float*   last_update;                   /* [N]       memory_bank.node_last_updated_times (float32)                         */
    float*   msg;                           /* [N, 2F_n+F_t+F_e]  last pending raw message per node                            */
    double*  msg_time;                      /* [N]       its interaction time                                                  */
    int32_t* has_msg;                       /* [N]       1 while a message is pending (list non-empty, MemoryModel.py:284)     */
} dygnn_tgn_state;

size_t dygnn_tgn_workspace_bytes(const dygnn_tgat_config* cfg_host, int64_t num_nodes, int64_t batch);
int dygnn_tgn_forward(const dygnn_tgat_config* cfg_host, const dygnn_tgat_weights* w_host, const dygnn_gru_weights* gru_host,
                      const dygnn_csr* csr_host, const float* node_feat, const float* edge_feat, const dygnn_tgn_state* state_host,
                      const int64_t* src, const int64_t* dst, const double* times, const int64_t* edge_ids /* NULL if !positive */,
                      int64_t batch, int32_t edges_are_positive, float* out_src, float* out_dst,
                      void* workspace, size_t workspace_bytes, dygnn_stream_t stream);

/* One evaluation / training step in ONE call (caller-side fusion): the reference issues the negative call and then the
 * positive call of a batch (evaluate_models_utils.py:85-107); both read the same state -- only the positive call writes it, at its end --
 * so the batch may hold both: the first n_positive pairs are the positive edges (they persist memories and leave new raw messages,
 * edge_ids [n_positive]), the remaining batch - n_positive pairs only read.  Rows and the state left behind are bit-identical to
 * dygnn_tgn_forward(negatives, edges_are_positive = 0) followed by dygnn_tgn_forward(positives, 1); the GRU update and every
 * kernel launch happen once instead of twice.  n_positive = batch / 0 reproduces the two modes of dygnn_tgn_forward. */
int dygnn_tgn_forward_step(const dygnn_tgat_config* cfg_host, const dygnn_tgat_weights* w_host, const dygnn_gru_weights* gru_host,
                           const dygnn_csr* csr_host, const float* node_feat, const float* edge_feat, const dygnn_tgn_state* state_host,
                           const int64_t* src, const int64_t* dst, const double* times, const int64_t* edge_ids /* [n_positive] */,
                           int64_t batch, int64_t n_positive, float* out_src, float* out_dst,
                           void* workspace, size_t workspace_bytes, dygnn_stream_t stream);

/* dygnn_tgn_forward_step on PRE-SAMPLED neighbour levels: the reference's TGN accepts any sample_neighbor_strategy
 * (models/MemoryModel.py:626-629 calls the sampler it was given; utils/utils.py:176-199 for 'uniform' / 'time_interval_aware', whose
 * draws consume the sampler's numpy RandomState in call order).  The host mirror replays those draws in the reference's order — ONE
 * get_historical_neighbors call on [src ; dst] per layer of the recursion, MemoryModel.py:104-131, :596-640 — and hands the levels over
 * in dygnn_tgat_levels' layout (batch = pairs; level L = [src ; dst]).  src / dst / times / edge_ids are read by the commit of a positive
 * call only (n_positive > 0).  Same workspace size as dygnn_tgn_forward. */
int dygnn_tgn_forward_levels(const dygnn_tgat_config* cfg_host, const dygnn_tgat_weights* w_host, const dygnn_gru_weights* gru_host,
                             const dygnn_tgat_levels* levels_host, const float* node_feat, const float* edge_feat,
                             const dygnn_tgn_state* state_host, const int64_t* src, const int64_t* dst, const double* times,
                             const int64_t* edge_ids /* [n_positive] */, int64_t batch, int64_t n_positive, float* out_src, float* out_dst,
                             void* workspace, size_t workspace_bytes, dygnn_stream_t stream);

/* ---- training: train_link_prediction.py:229-257 on the HIP path ------------------------------------
 * dygnn_dygformer_train_forward = models/DyGFormer.py:68-194 in TRAIN mode: dropout (probability dropout_p) on the attention
 * probabilities, the attention output and the FFN (models/DyGFormer.py:429-431, :456-460), masks drawn from a counter-based
 * generator keyed by `seed` (statistically, not bitwise, the reference's torch masks; dropout_p = 0 reproduces the eval
 * forward).  It keeps every activation the backward pass needs in `workspace` (size from
 * dygnn_dygformer_train_workspace_bytes; must stay untouched until dygnn_dygformer_backward of the same call returns) and
 * writes this call's padded lengths (S_src, S_dst) to seq_lens_host[2] (one host synchronisation of `stream`) — unless the caller
 * passes them in (both > 0, e.g. obtained with dygnn_window_lengths on a side stream), in which case the call stays asynchronous.
 * `packed` (may be NULL) = the kernel-ready copy of the CURRENT weights (dygnn_dygformer_pack / dygnn_dygformer_repack): with it, and a
 * shape the fused kernel takes, the forward is ONE kernel (the inference kernel plus dropout and the activation stores); without it the
 * product-by-product path runs.  Same results within fp32 rounding, same workspace contents for dygnn_dygformer_backward.
 * dygnn_dygformer_backward: `grads` has the layout of dygnn_dygformer_weights but its pointers are WRITABLE device buffers
 * of the parameter shapes that MUST BE ZERO on entry (the reductions accumulate into them); on return each holds
 * d(sum(out_src*grad_out_src) + sum(out_dst*grad_out_dst))/dparam.
 * The feature tables receive no gradient (constants in the reference, models/DyGFormer.py:28-29).
 * `packed` (may be NULL): the buffer the forward was given; with it the FFN blocks run backward as one fused kernel per layer. */
size_t dygnn_dygformer_train_workspace_bytes(const dygnn_dygformer_config* cfg_host, int64_t batch);
int dygnn_dygformer_train_forward(const dygnn_dygformer_config* cfg_host, const dygnn_dygformer_weights* w_host,
                                  const dygnn_csr* csr_host, const float* node_feat, const float* edge_feat,
                                  const int64_t* src, const int64_t* dst, const double* times, int64_t batch,
                                  float dropout_p, uint64_t seed, float* out_src, float* out_dst,
                                  void* workspace, size_t workspace_bytes, int32_t* seq_lens_host, const void* packed,
                                  dygnn_stream_t stream);
int dygnn_dygformer_backward(const dygnn_dygformer_config* cfg_host, const dygnn_dygformer_weights* w_host,
                             const dygnn_dygformer_weights* grads_host, const float* grad_out_src, const float* grad_out_dst,
                             int64_t batch, float dropout_p, uint64_t seed, const int32_t* seq_lens_host,
                             void* workspace, size_t workspace_bytes, const void* packed, dygnn_stream_t stream);

/* Caller-side link predictor, fused: sigmoid(MergeLayer(a,b)) with
 * MergeLayer = fc2(relu(fc1(cat(a,b)))) (models/modules.py:57-68; evaluate_models_utils.py:140-141).
 * a,b [n,dim]; fc1 [hidden, 2*dim]; fc2 [1,hidden]; out [n]. */
int dygnn_merge_layer_sigmoid(const float* a, const float* b, int64_t n, int32_t dim, int32_t hidden,
                              const float* fc1_w, const float* fc1_b, const float* fc2_w, const float* fc2_b,
                              float* out, dygnn_stream_t stream);
/* The same head for the TRAINING step (train_link_prediction.py:241-257): the logits z = MergeLayer(a, b) [n] (output_dim 1) and their
 * backward pass.  grad_logits = dL/dz [n]; grad_a, grad_b [n,dim] are written; the four parameter gradients are ACCUMULATED into buffers the
 * caller has zeroed; `workspace` = n * hidden floats, 16-byte aligned like a and b.  dim and hidden multiples of 4, hidden <= 192. */
int dygnn_merge_layer_logits(const float* a, const float* b, int64_t n, int32_t dim, int32_t hidden,
                             const float* fc1_w, const float* fc1_b, const float* fc2_w, const float* fc2_b,
                             float* out, dygnn_stream_t stream);
int dygnn_merge_layer_backward(const float* a, const float* b, int64_t n, int32_t dim, int32_t hidden,
                               const float* fc1_w, const float* fc1_b, const float* fc2_w, const float* grad_logits,
                               float* grad_a, float* grad_b, float* grad_fc1_w, float* grad_fc1_b, float* grad_fc2_w, float* grad_fc2_b,
                               float* workspace, dygnn_stream_t stream);


/* Evaluation metrics on the device, replacing the scikit-learn host round trip of
 * get_link_prediction_metrics / get_node_classification_metrics (utils/metrics.py:5-34; called per batch at
 * evaluate_models_utils.py:139-150 and per evaluation at :245-249).  predicts / labels: [n_groups, group_size] float32
 * (labels 0.0 / 1.0, as the reference builds them with ones_like / zeros_like); one group = one call of the reference
 * function.  Outputs (device, each nullable): average_precision [n_groups] (average_precision_score), roc_auc [n_groups]
 * (roc_auc_score; NaN where it would raise), bce_loss [n_groups] (torch.nn.BCELoss, mean; evaluate_models_utils.py:145),
 * status [n_groups] int32 (1 = only one class present: roc_auc_score raises ValueError there, and so does the wrapper).
 * Rank counts are exact; sums are float64 in a fixed order (reproducible); agreement with scikit-learn <= 1e-12. */
size_t dygnn_link_metrics_workspace_bytes(int64_t group_size, int64_t n_groups);
int dygnn_link_metrics(const float* predicts, const float* labels, int64_t group_size, int64_t n_groups,
                       double* average_precision, double* roc_auc, double* bce_loss, int32_t* status,
                       void* workspace, size_t workspace_bytes, dygnn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DYGNN_H */
