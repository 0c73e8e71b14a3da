/*
 * genie_hip.h — C ABI of libgenie_hip.so: MI355X (gfx950) kernels for GENIE's station <-> source-grid
 * message-passing hot path.
 *
 * The reference (imcbrearty/GENIE) is pure Python: it has NO plugin / FFI interface. The boundary this
 * library replaces is the body of three `torch.nn.Module.forward` methods and the third-party kernels they
 * call (PyG `MessagePassing.propagate`, `torch_scatter.scatter`):
 *
 *   DataAggregation.forward            /root/reference/Code/module.py:85-98
 *   BipartiteGraphOperator.forward     /root/reference/Code/module.py:224-229
 *   SpatialAggregation.forward/message /root/reference/Code/module.py:243-249
 *
 * as called from GCN_Detection_Network_extended.forward / forward_fixed / forward_fixed_source
 * (module.py:916-920, :973-977, :1010-1014). Each entry point below cites the lines it replaces.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to contiguous memory (e.g. torch `tensor.data_ptr()`), fp32 or int32;
 *  - `stream` is a `hipStream_t` passed as `void*` (NULL = default stream); calls enqueue work and return,
 *    they never synchronise and never allocate in the hot path;
 *  - the caller owns every input / output / workspace buffer; the library owns only what `genie_ctx_create`
 *    copies (base graphs, processing order) and its weight mirror; `genie_ctx_destroy` frees them;
 *  - return value 0 = ok, negative = error; `genie_last_error()` gives the message (thread-local);
 *  - product-graph node id p = g * n_sta + s (process_utils.py:720-722). Source nodes [0, n_grid) are OWNED by
 *    this context; source nodes [n_grid, n_grid_ext) are HALO rows (owned by another GPU when the grid is
 *    sharded over source nodes) that only appear as neighbours in `src_col`;
 *  - arithmetic: fp32 in, fp32 out, fp32 accumulation everywhere. On the reference's kNN graphs (8 station / 15 source
 *    neighbours) the P-sized stages multiply on the 16-bit matrix pipe with every fp32 operand as two fp16 pieces (x within
 *    one fp32 ulp, three partial products per product): fp32-class results, which need every hidden state of DataAggregation
 *    below 65504 in magnitude. The library checks that itself: at every weight commit it bounds those hidden states rigorously
 *    from the weights (inputs Slice, Mask in [-1, 1], as the reference produces them) and runs the fp32-MFMA kernels instead
 *    whenever the bound exceeds the fp16 range (genie_set_stage_precision / genie_stage_precision). edge_attr must stay
 *    below 65504 in magnitude (it is (grid - station) / scale_x_extend, process_continuous_days.py:630: O(1)).
 */
#ifndef GENIE_HIP_H
#define GENIE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct genie_ctx genie_ctx;

#define GENIE_OK 0
#define GENIE_ERR_ARG (-1)
#define GENIE_ERR_HIP (-2)
#define GENIE_ERR_STATE (-3)

/* Library / ABI version (major*100 + minor). */
int genie_version(void);
/* Message of the last error raised on this thread ("" if none). */
const char* genie_last_error(void);

/*
 * Create a context for one (stations x source-grid) product graph.
 * Replaces the graph objects cached by `set_adjacencies` (module.py:941-961); the inputs are the BASE kNN
 * graphs of process_utils.py:718-719 in CSR form (in-edges grouped by target), not the [2,E] product lists
 * of :720-721 — the Cartesian structure makes those implicit.
 *   sta_rowptr[n_sta+1], sta_col[sta_rowptr[n_sta]]   : station j -> station i edges, col = neighbour j
 *   src_rowptr[n_grid+1], src_col[src_rowptr[n_grid]] : source-node edges for OWNED nodes; col in [0,n_grid_ext)
 *   grid_order[n_grid]   : processing order of owned source nodes (a space-filling-curve order keeps the
 *                          neighbour rows of concurrently processed nodes in one XCD's L2); NULL = identity
 * All index arrays are int32 DEVICE pointers and are copied.
 */
int genie_ctx_create(genie_ctx** out, int n_sta, int n_grid, int n_grid_ext,
                     const int32_t* sta_rowptr, const int32_t* sta_col,
                     const int32_t* src_rowptr, const int32_t* src_col,
                     const int32_t* grid_order, float scale_rel);
/* Irregular product graph (`use_subgraph: True`, config.yaml:86; built by extract_inputs_adjacencies_subgraph,
 * process_utils.py:744-849): the product nodes are an explicit list of n_prod (station, source) pairs GROUPED BY SOURCE NODE
 * (process_utils.py:790-794), so `seg_rowptr[g] .. seg_rowptr[g+1]` is the row range of source node g (seg_rowptr[n_grid] =
 * n_prod), and both DataAggregation edge sets are CSR lists over product-node ids: p_sta_* = in-edges of A_in_sta (same
 * source node, neighbouring stations), p_src_* = in-edges of A_in_src, in stable edge order. src_rowptr / src_col = the base
 * source graph A_src (SpatialAggregation). Every [P, .] argument of the stage calls then has n_prod rows. The kernels that rely on
 * p = g * n_sta + s are not used (product-level CSR forms instead, inference and training); genie_nbr_mean is unavailable on such a
 * context, genie_embed_window* needs genie_set_subgraph_stations first; genie_set_edge_features / genie_set_absolute_pos take
 * positions per product node there ([n_prod, 3]). */
int genie_ctx_create_subgraph(genie_ctx** out, int n_sta, int n_grid, int64_t n_prod,
                              const int32_t* p_sta_rowptr, const int32_t* p_sta_col,
                              const int32_t* p_src_rowptr, const int32_t* p_src_col, const int32_t* seg_rowptr,
                              const int32_t* src_rowptr, const int32_t* src_col, const int32_t* grid_order, float scale_rel);
/* Station index of every product node of an irregular product graph (device pointer, n_prod int32 = row 0 of the reference's
 * A_src_in_sta, process_utils.py:790-794; copied): what the device embedding reads where a Cartesian graph has p % n_sta. `trv` of
 * genie_embed_window* is then [n_prod, 2], the travel times of the listed (station, source) pairs
 * (`trv_times[src, ind_use[sta], :]`, process_utils.py:605). */
int genie_set_subgraph_stations(genie_ctx* ctx, const int32_t* sta_of_prod, void* stream);
int genie_ctx_destroy(genie_ctx* ctx);
/* Optional station processing order: `order` (HOST pointer, n_sta int32, a permutation; typically the stations sorted along a
 * space-filling curve) = the caller's station id of the i-th station processed. A tile of the P-sized kernels is 16 consecutive
 * stations of one source node, and a station's neighbours are its nearest stations: with spatially sorted stations the rows a
 * tile gathers are shared between its lanes and adjacent in memory (config 2: stage 2 -6 %, config 4 with 2000 stations: whole
 * path -10 %). Purely internal: every input and output keeps the caller's station order (the split rows, c / wu / wv live in
 * processing order inside the workspace; genie_ws_export un-permutes). Honoured by the f16x2 stage 1 + pipelined stage 2
 * pair on Cartesian product graphs, ignored otherwise; NULL = off. All ranks of a sharded run must pass the same order. */
int genie_set_station_order(genie_ctx* ctx, const int32_t* order);
/* `use_phase_types: False` (config.yaml:91): genie_embed_window* then writes zeros into the phase-informed columns 2, 3 of Slice /
 * Mask (process_continuous_days.py:783-786); the caller passes every pick with phase 0 (:562-563) and zeroes `phase_label` for the
 * association heads (module.py:632-633, :706-707). Default: phase types in use. */
int genie_set_phase_types(genie_ctx* ctx, int use_phase_types);
/* `use_sign_input: True` (config.yaml:93; process_utils.py:610-614, the `use_sign_input` argument of extract_input_from_data): the
 * device embedding (genie_embed_window*) multiplies each of the four features of a product node by the sign of the NEGATIVE forward
 * difference of the series it is read from, at the index it is read at (+1 on the falling side of a pick's kernel, -1 on the rising
 * side, 0 on a flat stretch); Mask = |Slice| > 0.01 as before. Default off. */
int genie_set_sign_input(genie_ctx* ctx, int use_sign_input);
/* Work map of the row-layout stage 2 (k_stage2_ord: the association pass, the training forward of other graph shapes): 1 = blocks of 4
 * adjacent source nodes per workgroup, one node per wave (the default from 1024 stations up: the gathers leave L2 there and locality
 * pays), 0 = interleaved items, -1 = the default for the context's station count. Results do not depend on it (tests). The library reads
 * no environment variable; this is the one scheduling choice a caller can make. */
int genie_set_stage2_workmap(genie_ctx* ctx, int blocks_of_four);
/* Arithmetic of the G-sized tail of inference calls (Bipartite read-out, SpatialAggregation x3, SpatialDirect + TemporalAttention
 * on the grid): fp64_chains != 0 (default) = every Linear as a chain of fp64 MFMAs on the fp32 inputs and weights, fp64 PReLUs
 * and sums, one rounding to fp32 per kernel; 0 = fp32 MFMA chains (the arithmetic of the training forward, and the A/B form).
 * The fp32 chains are where almost all of the distance between (y, x) and the reference's fp64 run is made. */
int genie_set_tail_precision(genie_ctx* ctx, int fp64_chains);
/* Arithmetic of the P-sized stages on the reference's kNN graphs. mode 0 (default) = automatic: two-piece fp16 operands on the
 * 16-bit matrix pipe (k_stage1_h2 / k_stage2_h2) while the fp16 range guard of the committed weights holds, the fp32-MFMA kernels
 * otherwise; 1 = two-piece fp16 operands regardless of the guard (A/B runs; hidden states above 65504 become non-finite);
 * 2 = fp32 MFMA always. No environment variable takes part. */
int genie_set_stage_precision(genie_ctx* ctx, int mode);
/* What runs for the weights committed so far (commits pending ones): *mode as set, *f16x2_active 1 / 0, and the two numbers
 * the range guard compares with 60000: the largest rigorous bound of a hidden state that is split into fp16 pieces, and the
 * largest weight magnitude in the form that is rounded to fp16. Any out pointer may be NULL. Synchronises `stream` when weights
 * were pending (the guard's 16-byte read-back). */
int genie_stage_precision(genie_ctx* ctx, int* mode, int* f16x2_active, float* act_bound, float* weight_bound, void* stream);
/* Input range of the f16x2 kernels (round 5). The range guard above holds for inputs in [-1, 1] (process_utils.py:262-275, :610-629: the
 * only inputs the reference's pipeline produces) and, its bound being linear in the input magnitude, up to |input| <= *limit = 60000 /
 * act_bound. The split pass of every f16x2 stage-1 call checks the rows it reads and records the largest magnitude BEYOND that limit in a
 * word of host-mapped memory; this call returns it in *max_seen (0 = every input so far was inside the limit) WITHOUT synchronising: it
 * reflects the split passes that have completed. reset != 0 = ONE atomic fetch-and-clear: the value returned is exactly the value cleared
 * (kernels still in flight may be writing the word: a read followed by a separate store could drop what they wrote in between). Nobody
 * reads the word once the context is destroyed: check after the LAST call on a context too (the Python host does when it replaces a
 * context, at the read-backs it makes anyway, and warns from the destructor). A caller that sees a non-zero value must discard the
 * results of the calls issued since its last check and switch to genie_set_stage_precision(ctx, 2) (the Python host does both and
 * raises). Never set by the fp32 kernels, by the device embedding's split rows (values in [-1, 1] by construction) or in mode 2. */
int genie_input_range(genie_ctx* ctx, float* max_seen, float* limit, int reset);
/* Index errors found on the device since the last reset, read from host-mapped memory without synchronising (like genie_input_range):
 * bit 0 = a pick of a genie_lslc_fwd call indexed outside the time-pointer table (tpick outside dt_partition, or ipick outside the
 * stations of A_edges; the kernel clamps the index, the reference's indexing at Code/module.py:635-640 fails with a device-side
 * assertion reported at its next synchronisation). reset != 0 = atomic fetch-and-clear (see genie_input_range). The Python host raises
 * IndexError at its next call on the context, where it has just waited for the device anyway, and when it replaces the context. */
int genie_index_flags(genie_ctx* ctx, unsigned* flags, int reset);
/* Range check of a caller's index list on the device, no host round trip: sets `bit` (2, 4, ...; bit 0 belongs to genie_lslc_fwd) in the
 * word genie_index_flags returns when some idx[i] lies outside [lo, hi). The Python host uses bit 1 (value 2) for the station indices
 * `ipick` of the Arrivals head (Code/module.py:703-713 indexes `trv_out[:, ipick]` with them), which it clamps for its own use. */
int genie_index_check(genie_ctx* ctx, const int64_t* idx, int64_t n, int64_t lo, int64_t hi, unsigned bit, void* stream);
/* With a station processing order: registers the caller's STATIC edge_attr [P, 3] (A_src_in_edges.x, process_utils.py:722: a
 * function of the geometry only); the library keeps it in the form stage 2 consumes (two-piece fp16 operand fragments in
 * processing order, 32 B per product node) and uses that in every stage-2 call that is passed this same pointer; any other
 * edge_attr is converted per call. Call again if the contents change; NULL unregisters. No-op without a station order. */
int genie_set_static_edge_attr(genie_ctx* ctx, const float* edge_attr, void* stream);
/* Every buffer of the workspace that carries data from one call to the next (stage 1 -> stage 2: c, wu, wv; stage 2 ->
 * tail: Bipartite partials; SpatialAggregation / read-out scratch) exists several times; `slot` (0..15) selects the copy
 * used by the calls issued next (33 copies of the G-sized buffers: two batches of 16 windows in flight and one for single-stream calls; the P-sized rows have 4, indexed slot % 4). With several HIP streams a caller can run stage 1 of window i+1 (MFMA-bound), stage 2 of window
 * i (HBM-bound) and the G-sized tail of window i-1 (latency-bound) concurrently: all calls of one window use the same
 * slot, consecutive windows rotate through the slots, and the caller orders "stage 1 of window i+2 after stage 2 of window i" etc. with
 * events. Default slot 0. */
int genie_set_slot(genie_ctx* ctx, int slot);
/* The G-sized tail of `nwin` windows (whose stage-2 partials sit in slots slot0 .. slot0+nwin-1) in one set of launches:
 * Bipartite read-out, SpatialAggregation x3 -> x_spatial_out [nwin, G, 30], y_out [nwin, G, n_t] (genie_readout_grid) and,
 * if x_out != NULL, x_out [nwin, n_query, n_t] (genie_readout_query). Bit-identical to the per-window calls; replaces
 * nwin x (genie_bipartite_readout + genie_spatial_agg3_fwd + genie_readout_grid + genie_readout_query) in the apply
 * loop (process_continuous_days.py:761-810), whose windows are independent. */
int genie_tail_batched(genie_ctx* ctx, int slot0, int nwin, const float* pos, const float* x_query, const int32_t* knn,
                       int n_query, int k, const float* t_query, int n_t, float* x_spatial_out, float* y_out, float* x_out,
                       void* ws, void* stream);
/* Temporal scale of TemporalAttention: `scale_t = 3 * kernel_sig_t` (module.py:40); default 9.0. */
int genie_set_scale_t(genie_ctx* ctx, float scale_t);
/* `use_absolute_pos: True` (config.yaml:92; module.py:916, :971, :1007): every product node's input gets its station position and
 * its source position, each divided by 3 * scale_rel, appended (in_channels 4 -> 10). Positions are [n_sta,3] / [n_grid_ext,3]
 * fp32 device pointers; the 6 extra columns of init_trns.weight go to the registry entry "DataAggregation.init_trns.weight_abs"
 * ([30,6] = weight[:, 4:10]) and "DataAggregation.init_trns.weight" keeps the [30,8] layout (weight[:, [0:4, 10:14]]). The
 * neighbours' hidden states are recomputed with their own positions (one more K-step per neighbour in the f16x2 stage-1 kernel;
 * two k-steps in the generic fp32-MFMA kernel that serves ragged graphs). Null = off. */
int genie_set_absolute_pos(genie_ctx* ctx, const float* pos_sta, const float* pos_src, void* stream);
/* DataAggregationEdges variant (`use_updated_model_definition: True`, config.yaml:95; module.py:102-174): every message is
 * [x_j || phi(pos_j - pos_i) || phi(|pos_j - pos_i|)], phi(d) = sign(d) exp(-d^2 / (2 scale_rel^2)) (forward :1059-1072,
 * set_adjacencies :1102-1111). On the product graph the mean of those 4 edge features is a static vector per station
 * (station graph) / per source node (source graph); the library computes them from the positions ([n_sta,3] and
 * [n_grid_ext,3] device pointers, fp32) and applies the 4 edge-feature columns of l1_t?_2 / l2_t?_2 (registry entries
 * "DataAggregation.l?_t?_2.weight_pos", [out,4]) as per-node additive terms. The other columns keep the DataAggregation
 * layout: the caller passes `l1_t?_2.weight[:, [0:60, 64:68]]` and `l2_t?_2.weight[:, [0:90, 94:98]]` under the usual names.
 * Null positions switch back to plain DataAggregation. On an irregular product graph (genie_ctx_create_subgraph) the mean runs over
 * the PRESENT neighbours of a product node: both arguments are then [n_prod, 3] (the station's / the source node's position of every
 * product node) and the additive terms are per product node. */
int genie_set_edge_features(genie_ctx* ctx, const float* pos_sta, const float* pos_src, void* stream);

/*
 * Weight mirror. Parameter names are the reference's state_dict keys (e.g. "DataAggregation.l1_t1_2.weight",
 * layout [out, in] row-major exactly as nn.Linear stores it; PReLU slopes are 1-element tensors).
 * Enumerate with genie_weights_count/name/numel; copy with genie_weights_set (async D2D on `stream`).
 * The MFMA-fragment repack happens lazily on the next forward call (or explicitly with genie_weights_commit).
 */
int genie_weights_count(void);
const char* genie_weights_name(int i);
int64_t genie_weights_numel(int i);
int64_t genie_weights_offset(int i);      /* offset (floats) of parameter i inside the flat mirror */
int64_t genie_weights_blob_floats(void);  /* size (floats) of the flat mirror; every offset is 16-B aligned */
int genie_weights_set(genie_ctx* ctx, const char* name, const float* dev_ptr, int64_t numel, void* stream);
/* Upload the whole mirror at once: `blob` holds every parameter at genie_weights_offset(i). */
int genie_weights_set_blob(genie_ctx* ctx, const float* blob_dev, int64_t n_floats, void* stream);
int genie_weights_commit(genie_ctx* ctx, void* stream);

/* Bytes of caller-provided workspace the forward calls need (256-B aligned base required). */
size_t genie_workspace_bytes(const genie_ctx* ctx);

/*
 * DataAggregation, stage 1 = everything that does not need the SECOND pair of neighbour means (module.py:87-95).
 * For every OWNED product node, with h0 = PReLU(init_trns([Slice || Mask])) recomputed on the fly for the node and
 * for each of its neighbours from their raw 8 input floats (h0 is never stored):
 *   h1 = PReLU1([l1_t1_2[h0||mean_sta PReLU11(h0)||M] || l1_t2_2[h0||mean_src PReLU12(h0)||M]])
 *   u = PReLU21(l2_t1_1 h1), v = PReLU22(l2_t2_1 h1)
 *   wu = l2_t1_2.weight[:, 60:90] u,  wv = l2_t2_2.weight[:, 60:90] v      (mean_N(W x) = W mean_N(x): the 15-channel
 *                                                                         operands the second pair of means averages)
 *   c  = [l2_t1_2.weight[:, 0:60] h1 + l2_t1_2.weight[:, 90:94] M + bias || same for l2_t2_2]   (node-local part)
 *   slice, mask : [n_grid_ext*n_sta, 4] fp32 — halo rows (sharded case) are shipped raw, 8 floats per row.
 * c, wu, wv are kept in the workspace; h1, u, v never leave the registers.
 */
int genie_da_stage1(genie_ctx* ctx, const float* slice, const float* mask, void* ws, void* stream);
/* Sub-range form for the source-node-sharded case (the halo exchange of `wv` overlaps compute): only the owned
 * source nodes at positions [gi_begin, gi_end) of the processing order (`grid_order` of genie_ctx_create) are processed.
 * `first` != 0 on the first range call of a window: it runs the input split pass over ALL rows (owned + halo); later range
 * calls of the same window (same slice / mask / workspace / slot) pass 0. A caller that puts the nodes other ranks need first
 * in the processing order can start sending their `wv` rows while the rest of stage 1 runs. */
int genie_da_stage1_range(genie_ctx* ctx, const float* slice, const float* mask, int gi_begin, int gi_end, int first,
                          void* ws, void* stream);
/* Parity/debug variant: additionally writes h0 [n_grid*n_sta, 30] and h1 [n_grid*n_sta, 60]. */
int genie_da_stage1_debug(genie_ctx* ctx, const float* slice, const float* mask, float* h0_out, float* h1_out,
                          void* ws, void* stream);
/*
 * Halo access for the sharded case: device pointer / row pitch (floats) of the projected `wv` activations
 * inside the workspace, laid out [n_grid_ext*n_sta, pitch] (pitch = 16: 15 channels + 1 zero); rows
 * >= n_grid*n_sta must be filled by the caller (RCCL) between stage 1 and stage 2.
 */
float* genie_ws_v_ptr(const genie_ctx* ctx, void* ws);
int genie_ws_v_pitch(const genie_ctx* ctx);
/*
 * DataAggregation stage 2 + BipartiteGraphOperator (module.py:94-96 and :224-229), OWNED rows:
 *   x_latent = PReLU2(c + [mean_sta wu || mean_src wv])                                      -> optional output
 *   out_g    = PReLU_b2(fc2( sum_s max_c(M) * PReLU_b1(fc1[x_latent || edge_attr]) ))       -> bip_out[n_grid,15]
 *   edge_attr : [n_grid*n_sta, 3] (`A_src_in_edges.x`, process_continuous_days.py:630)
 *   x_latent_out : [n_grid*n_sta, 30] or NULL when the caller does not need it (forward_fixed_source)
 */
int genie_da_stage2_bipartite(genie_ctx* ctx, const float* mask, const float* edge_attr,
                              float* x_latent_out, float* bip_out, void* ws, void* stream);
/* The two halves of genie_da_stage2_bipartite as separate calls (so they can sit on different streams):
 * per-tile station-sum partials (the P-sized kernel), then r_g = sum of partials and out_g = PReLU_b2(fc2 r_g). */
int genie_da_stage2_partials(genie_ctx* ctx, const float* mask, const float* edge_attr, float* x_latent_out, void* ws,
                             void* stream);
/* ... and the sub-range form of the P-sized half: partials of the owned source nodes at positions [gi_begin, gi_end) of the
 * processing order only. Nodes without halo neighbours can run before the halo rows of `wv` have arrived. */
int genie_da_stage2_partials_range(genie_ctx* ctx, const float* mask, const float* edge_attr, float* x_latent_out,
                                   int gi_begin, int gi_end, void* ws, void* stream);
int genie_bipartite_readout(genie_ctx* ctx, float* bip_out, void* ws, void* stream);
/*
 * SpatialAggregation (module.py:243-249; instances :889-891) on the source graph of this context.
 * Requires an UNSHARDED source graph (n_grid_ext == n_grid): when the product graph is sharded over source
 * nodes, the [G,15] Bipartite output is all-gathered and the three G-sized layers run on a second context
 * that holds the whole source graph (n_sta = 1).
 *   layer : 1, 2 or 3 (c_in = 15 for layer 1 else 30)
 *   x_in  : [n_grid, c_in], pos : [n_grid, 3] metres (divided by scale_rel inside, module.py:245)
 *   out   : [n_grid, 30]
 * The edge-mean of PReLU3(fglobal(x_j)) over ALL edges (module.py:249) is an out-degree weighted node mean.
 */
int genie_spatial_agg_fwd(genie_ctx* ctx, int layer, const float* x_in, const float* pos, float* out,
                          void* ws, void* stream);
/* SpatialAggregation1 -> 2 -> 3 chained (module.py:1012-1014): every layer kernel also emits the per-node pre-pass
 * of the next one (x-part of the message Linear and the edge-mean partials). x_in15 [n_grid,15] -> out [n_grid,30]. */
int genie_spatial_agg3_fwd(genie_ctx* ctx, const float* x_in15, const float* pos, float* out, void* ws, void* stream);

/*
 * Fused single-GPU path = module.py:1010-1014: DataAggregation -> Bipartite_ReadIn -> SpatialAggregation1..3.
 *   x_spatial_out : [n_grid, 30]; x_latent_out optional ([P,30] or NULL); bip_out optional ([n_grid,15] or NULL)
 */
int genie_path_fwd(genie_ctx* ctx, const float* slice, const float* mask, const float* edge_attr,
                   const float* pos, float* x_spatial_out, float* x_latent_out, float* bip_out,
                   void* ws, void* stream);

/*
 * Read-out heads needed to return (y, x) from forward_fixed_source (module.py:1015-1018).
 *   genie_readout_grid : y[n_grid, n_t]  = TemporalAttention(SpatialDirect(x_spatial), t_query)     module.py:251-260, 299-331
 *   genie_readout_query: x[n_query, n_t] = TemporalAttention(SpatialAttention(x_spatial, x_query, x_grid), t_query)
 *                        module.py:262-297; `knn` [n_query, 10] int32 = indices of the 10 nearest grid nodes of every
 *                        query (the `knn(x_context/1000, x_query/1000, k=10)` of module.py:282, computed by the caller
 *                        once per query set); softmax is over those 10 edges per head, aggregation 'add', mean over heads.
 *   x_spatial [n_grid,30]; x_grid [n_grid,3], x_query [n_query,3] in metres; t_query [n_t] seconds, n_t <= 10.
 */
int genie_readout_grid(genie_ctx* ctx, const float* x_spatial, const float* t_query, int n_t, float* y_out, void* stream);
int genie_readout_query(genie_ctx* ctx, const float* x_spatial, const float* x_grid, const float* x_query,
                        const int32_t* knn, int n_query, int k, const float* t_query, int n_t, float* x_out,
                        void* ws, void* stream);
/* The same read-outs with the latent input of TemporalAttention exported next to them, as the 4-output forward needs it
 * (module.py:978-981): y_latent_out [n_grid, 30] = SpatialDirect(x_spatial) (:978), latent_out [n_query, 30] =
 * SpatialAttention(x_spatial, x_query, x_grid) (:981, `x_src` for the candidate sources). */
int genie_readout_grid_latent(genie_ctx* ctx, const float* x_spatial, const float* t_query, int n_t, float* y_out,
                              float* y_latent_out, void* stream);
int genie_readout_query_latent(genie_ctx* ctx, const float* x_spatial, const float* x_grid, const float* x_query,
                               const int32_t* knn, int n_query, int k, const float* t_query, int n_t, float* x_out,
                               float* latent_out, void* ws, void* stream);

/*
 * Pick -> Slice/Mask embedding on device = `extract_input_from_data` (process_utils.py:460-642, use_sign_input False),
 * the step that feeds the path once per window. Removes the [P,8] fp32 H2D copy per window.
 *   pick_t [n] float64 absolute pick times, pick_sta [n] int32 station index in the model's station order (-1 = not
 *   used), pick_phase [n] int32 (0 = P, 1 = S); the caller passes the picks inside (t0 - 2 sigma, t0 + max_t + 2 sigma)
 *   (process_utils.py:476). trv [n_grid_ext*n_sta, 2] fp32 theoretical P / S travel times per product node (static).
 *   emb_ws: scratch of 2 * n_sta * genie_embed_ntime(...) floats. Outputs slice_out / mask_out [n_grid_ext*n_sta, 4].
 */
int genie_embed_ntime(double t0, double max_t, double kernel_sig_t, double dt);
int genie_embed_window(genie_ctx* ctx, const double* pick_t, const int32_t* pick_sta, const int32_t* pick_phase, int n_picks,
                       double t0, double max_t, double kernel_sig_t, double dt, const float* trv, float* emb_ws,
                       float* slice_out, float* mask_out, void* stream);
/* Same, and additionally leaves the split input rows of the f16x2 stage-1 kernel in `workspace`: the NEXT genie_da_stage1 /
 * genie_path_fwd call on this context with exactly these slice_out / mask_out pointers and this workspace skips its
 * k_split_rows pass (one-shot; the caller must not modify Slice / Mask in between, and stage 1 must run on the same stream or
 * after it). A no-op extension on contexts that do not use the f16x2 kernel. */
int genie_embed_window_split(genie_ctx* ctx, const double* pick_t, const int32_t* pick_sta, const int32_t* pick_phase,
                             int n_picks, double t0, double max_t, double kernel_sig_t, double dt, const float* trv,
                             float* emb_ws, float* slice_out, float* mask_out, void* workspace, void* stream);

/* Neighbour means on the implicit product graph for [P, row_floats] fp32 rows (row_floats = 16 or 32, 16-byte aligned, or 30 = unpadded [P, 30] rows, 8-byte aligned):
 *   out_sta[(g,s)] = mean_k x_sta[(g, sta_nbr_k(s))]      (MessagePassing('mean') over A_in_sta)
 *   out_src[(g,s)] = mean_k x_src[(src_nbr_k(g), s)]      (... over A_in_src; x_src has n_grid_ext * n_sta rows)
 * Either pair may be null. Used by the association heads (DataAggregationAssociationPhase, module.py:395-400), whose
 * per-node Linears run on PyTorch-ROCm; an empty neighbourhood gives 0. */
int genie_nbr_mean(genie_ctx* ctx, const float* x_sta, const float* x_src, float* out_sta, float* out_src, int row_floats,
                   void* stream);
/* Backward of a single-slope PReLU over n contiguous fp32 values (training path): dx = dy * (x >= 0 ? 1 : slope),
 * dslope[0] = sum over x < 0 of dy * x, summed in a fixed order. `scratch` = 2048 floats; pointers 16-byte aligned. */
int genie_prelu_bwd(const float* x, const float* dy, const float* slope, int64_t n, float* dx, float* dslope, float* scratch,
                    void* stream);

/* Weight and bias gradients of a per-node Linear y = x W^T + b over N contiguous rows (training path):
 * dW[M, K] = dy^T x, db[M] = column sums of dy (db may be NULL), M <= 128, K <= 128, summed in a fixed order.
 * `scratch` holds genie_linear_bwd_scratch_floats(K) floats. */
int64_t genie_linear_bwd_scratch_floats(int K);
int genie_linear_bwd_wb(const float* x, const float* dy, int64_t N, int K, int M, float* dW, float* db, float* scratch, void* stream);

/* Adjoint of genie_nbr_mean (training): dx_sta[(g,j)] = sum_{i : j in N_sta(i)} g_sta[(g,i)] / deg(i), likewise for the source
 * graph. Deterministic (a gather over the reversed graphs, built once per context); unsharded Cartesian contexts only. */
int genie_nbr_mean_bwd(genie_ctx* ctx, const float* g_sta, const float* g_src, float* dx_sta, float* dx_src, int row_floats,
                       void* stream);

/* out_dev [n_blocks][2] = (HW_REG_XCC_ID, HW_REG_HW_ID) of the CU every workgroup of a probe launch ran on: workgroup b of a
 * launch lands on XCD b % 8, which the XCD-chunked sweeps of the stage kernels rely on (for speed only). */
int genie_where_am_i(int32_t* out_dev, int n_blocks, void* stream);
/* Grid caps (workgroups) of the read-out (k_readout, k_ro_pre) and SpatialAggregation kernels of the G-sized tail; 0 = default
 * (one read-out workgroup per CU, two SpatialAggregation workgroups per CU: lowest latency). In the window pipeline, where a
 * tail workgroup has a CU to itself while it lives, fewer workgroups cost the P-sized kernels less CU-time as long as the
 * tail chain still ends within two windows: 96 / 64 at config 2 on 256 CUs is 3.4 % faster end to end, 64 / 32 is 9 % slower. */
int genie_set_tail_grid(genie_ctx* ctx, int readout_workgroups, int sa_workgroups);

/* Training step of the path (train_GENIE_model.py:1786-1861): DataAggregation + the P-sized half of
 * Bipartite_ReadIn forward with the pre-activations kept, and their backward as three P-sized HIP passes.
 *   genie_da_train_fwd: the generic fp32 stage kernels (same arithmetic as genie_da_stage1 / genie_da_stage2_partials) with
 *     `save` (genie_train_save_floats(ctx) floats) filled; x_latent_out [P, 30] optional; r_out [n_grid, 30] = the per-source-
 *     node station sum of the gated Bipartite messages (module.py:229, before fc2: out_g = PReLU_b2(fc2 r_g) stays with the
 *     caller, G-sized).
 *   genie_da_train_bwd: d_r [n_grid, 32] (gradient of r_out, rows padded to 32 floats) -> grad_blob: gradients of every
 *     DataAggregation parameter and of Bipartite_ReadIn.fc1 / activate1, laid out like the weight mirror
 *     (genie_weights_offset(i), genie_weights_blob_floats() floats; entries of other parameters are zero). `scratch`:
 *     genie_train_scratch_floats(ctx) floats. Deterministic (fixed-order reduction of per-wave partials).
 * Unsharded product graphs, Cartesian or irregular (genie_ctx_create_subgraph: tiles of 16 consecutive product nodes, the transposed
 * means of the backward over the reversed PRODUCT-level graphs; round 4). With genie_set_edge_features / genie_set_absolute_pos in force (the two other model
 * definitions) the forward is those variants' inference kernels and the backward adds the gradients of their static-term weights
 * ("<layer>.weight_pos", "init_trns.weight_abs"): such a term is W_f f[n] with f a fixed table over the stations or the source
 * nodes, so dW_f = sum_n (sum of the kept gradient rows over the product nodes of n) (x) f[n] -- per-station / per-source-node
 * sums of the rows the three passes keep (k_gr_sum_sta / k_gr_sum_src), contracted with the table (k_static_dw), fixed order. */
size_t genie_train_save_floats(const genie_ctx* ctx);
size_t genie_train_scratch_floats(const genie_ctx* ctx);
int genie_da_train_fwd(genie_ctx* ctx, const float* slice, const float* mask, const float* edge_attr, float* save,
                       float* x_latent_out, float* r_out, void* ws, void* stream);
int genie_da_train_bwd(genie_ctx* ctx, const float* slice, const float* mask, const float* edge_attr, const float* save,
                       const float* d_r, float* scratch, float* grad_blob, void* stream);

/* Training step, the G- / Q-sized tail (round 3): Bipartite_ReadIn.fc2 (module.py:229), SpatialAggregation1..3 (:243-249, the
 * grid-wide edge-mean term included), SpatialDirect (:251-260), SpatialAttention (:262-297) and TemporalAttention (:299-331,
 * applied to both read-outs), i.e. everything of `forward_fixed_source` (:1011-1018) after the station sum, in both directions.
 *   genie_tail_train_fwd: call right after genie_da_train_fwd on the same workspace slot (it consumes the per-tile partials).
 *     It IS the inference tail (k_bip_out_m, k_sa_pre_m / k_sa_layer_m, k_ro_pre_m, k_readout_m); the layer inputs the backward
 *     recomputes from (r, bip, sa1, sa2, x_spatial) go to `tsave` (genie_tail_train_save_floats(ctx) floats; x_spatial [n_grid, 30]
 *     starts at float 112 * n_grid). y_out [n_grid, n_t], x_out [n_query, n_t]; y_latent_out [n_grid, 30] optional
 *     (SpatialDirect output, the input of the association heads).
 *   genie_tail_train_bwd: d_y [n_grid, n_t], d_x [n_query, n_t]; optional extra upstream gradients d_xs_extra [n_grid, 30]
 *     (other consumers of x_spatial), d_ylat_extra [n_grid, 30] (other consumers of y_latent) and d_qlat_extra [n_query, 30] (consumers
 *     of the SpatialAttention output of a query row, :981: the `x_src` rows the arrival head reads) -> d_r_out [n_grid, 32] (the
 *     input of genie_da_train_bwd) and grad_blob (genie_train_grad_floats() floats, zeroed by the call; weight-mirror layout):
 *     gradients of every tail parameter. rknn_rowptr [n_grid + 1] / rknn_edge [n_query * 10]: the query kNN table reversed =
 *     for every grid node the attention edges i * 10 + k that end in it, ascending. `scratch`:
 *     genie_tail_train_scratch_floats(ctx, n_query) floats. Deterministic: per-wave partial sums reduced in a fixed order,
 *     scatter-shaped gradients gathered over reversed graphs, no atomics.
 *   genie_train_bwd: genie_tail_train_bwd followed by genie_da_train_bwd (driven by the tail's d r) into ONE gradient blob:
 *     the whole backward of a `forward_fixed_source` training step (train_GENIE_model.py:1843-1846). */
size_t genie_tail_train_save_floats(const genie_ctx* ctx);
size_t genie_tail_train_scratch_floats(const genie_ctx* ctx, int n_query);
size_t genie_train_grad_floats(void);
int genie_tail_train_fwd(genie_ctx* ctx, const float* pos, const float* x_query, const int32_t* knn, int n_query, int k,
                         const float* t_query, int n_t, float* tsave, float* y_latent_out, float* y_out, float* x_out, void* ws,
                         void* stream);
int genie_tail_train_bwd(genie_ctx* ctx, const float* pos, const float* x_query, const int32_t* knn, const int32_t* rknn_rowptr,
                         const int32_t* rknn_edge, int n_query, int k, const float* t_query, int n_t, const float* tsave,
                         const float* d_y, const float* d_x, const float* d_xs_extra, const float* d_ylat_extra, const float* d_qlat_extra,
                         float* scratch, float* d_r_out, float* grad_blob, void* stream);
int genie_train_bwd(genie_ctx* ctx, const float* slice, const float* mask, const float* edge_attr, const float* save, const float* pos,
                    const float* x_query, const int32_t* knn, const int32_t* rknn_rowptr, const int32_t* rknn_edge, int n_query, int k,
                    const float* t_query, int n_t, const float* tsave, const float* d_y, const float* d_x, const float* d_xs_extra,
                    const float* d_ylat_extra, const float* d_qlat_extra, float* tail_scratch, float* front_scratch, float* d_r_scratch,
                    float* grad_blob, void* stream);

/* Association heads on the product graph, the P-sized part of `forward_fixed` after the source branch
 * (module.py:986-990): BipartiteGraphReadOutOperator (:333-352) followed by DataAggregationAssociationPhase (:356-403).
 *   y_latent [n_grid, 30] (SpatialDirect output), mask_src [n_grid] (`mask_out`, :985), x_latent [P, 30] (DataAggregation
 *   output, detached at :990), mask [P, 4], edge_attr [P, 3]  ->  out [P, 30]
 * assoc_ws: genie_assoc_workspace_bytes(ctx) bytes of scratch (16-byte aligned; tr / q1 / q2 rows). The call reuses the c / wu /
 * wv buffers of the current workspace slot: issue it after the DataAggregation stage 2 of the same window. Parameters under
 * their state_dict names ("BipartiteGraphReadOutOperator.*", "DataAggregationAssociationPhase.*"; the static-term columns of the two
 * other model definitions under "<name>_pos" / "<name>_abs", as for DataAggregation).
 * Unsharded product graphs (Cartesian, or irregular since round 4: product-level CSR forms of the kernels). */
size_t genie_assoc_workspace_bytes(const genie_ctx* ctx);
int genie_assoc_fwd(genie_ctx* ctx, const float* y_latent, const float* mask_src, const float* x_latent, const float* mask,
                    const float* edge_attr, float* out, void* assoc_ws, void* ws, void* stream);

/* Exact k nearest neighbours on the device, replacing the reference's `torch_cluster.knn(x_context / scale, x_query / scale, k)`
 * calls: out_idx [n_query, k] int32 = indices into x_context of the k nearest context points of every query, nearest first
 * (ties: smaller index first; -1 when fewer than k candidates exist). fp64 distances on the fp32 coordinates as given: the
 * reference's common factor 1 / 1000 (module.py:282, process_utils.py:718-719) does not change the order. exclude_self != 0: candidate i is skipped for query
 * i (x_query = x_context: the `knn(x, x, k + 1)` + `remove_self_loops` idiom of the base graphs). 1 <= k <= 16. */
int genie_knn(const float* x_context, int n_context, const float* x_query, int n_query, int k, int exclude_self,
              int32_t* out_idx, void* stream);

/* One-time graph setup, training call convention (round 5): `forward` receives the product edge lists of a NEW graph per sample
 * (train_GENIE_model.py:1722-1786; built at :1140-1149 as process_utils.py:720-721 builds them) and the library consumes only their base
 * graphs, so every sample's lists are verified to be the Cartesian product of their own first blocks:
 *   A_in_sta int64 [2][E_sta] (row 0 then row 1), E_sta = n_grid * e_sta, entry e = base_sta[e % e_sta] + n_sta * (e / e_sta);
 *   A_in_src int64 [2][E_src], E_src = n_sta * e_src, entry e = base_src[e % e_src] + (e / e_src), base_src multiples of n_sta.
 * `flags` (device int32, zeroed by the caller) gets bit 0 set when A_in_sta is not of that form, bit 1 when A_in_src is not. One pass,
 * no allocation, no synchronisation (replaces the `torch.equal` of two materialised copies in genie_amd/graph.py). */
int genie_product_check(const int64_t* A_in_sta, int64_t E_sta, const int64_t* A_in_src, int64_t E_src, int n_sta, int n_grid,
                        int32_t* flags, void* stream);

/* Training step of the P-sized association heads (round 3; module.py:986-990 inside train_GENIE_model.py:1786-1861):
 *   genie_assoc_train_fwd = genie_assoc_fwd in the caller's station order with the pre-activations of every layer kept in `asave`
 *     (genie_assoc_train_save_floats(ctx) floats: 20 blocks of 16 floats per product node);
 *   genie_assoc_train_bwd: d_s [P, 30] (gradient of the head's output) -> d_ylat_out [n_grid, 30] (gradient of y_latent; x_latent is
 *     detached at module.py:990) and grad_blob (genie_weights_blob_floats() floats, zeroed by the call, weight-mirror layout): gradients of
 *     every BipartiteGraphReadOutOperator / DataAggregationAssociationPhase parameter. Four P-sized passes (k_as_b3, k_train_b1<true>,
 *     k_as_b1, k_as_b0: the structure of DataAggregation's backward) + one G-sized (k_as_g); `scratch`:
 *     genie_assoc_train_scratch_floats(ctx) floats. Deterministic. Unsharded product graphs, Cartesian or irregular; under the two other model
 *     definitions the static-term weight gradients are added as in genie_da_train_bwd. */
size_t genie_assoc_train_save_floats(const genie_ctx* ctx);
size_t genie_assoc_train_scratch_floats(const genie_ctx* ctx);
int genie_assoc_train_fwd(genie_ctx* ctx, const float* y_latent, const float* mask_src, const float* x_latent, const float* mask,
                          const float* edge_attr, float* out, float* asave, void* assoc_ws, void* ws, void* stream);
int genie_assoc_train_bwd(genie_ctx* ctx, const float* y_latent, const float* mask_src, const float* x_latent, const float* mask,
                          const float* edge_attr, const float* asave, const float* d_s, float* scratch, float* d_ylat_out,
                          float* grad_blob, void* stream);

/* LocalSliceLgCollapse P (phase_head 0) / S (1), module.py:610-659, on the device: for every pick the 10 product nodes listed in the
 * time-pointer table a_edges [n_sta * l_dt * 10] (int32 product-node ids, `assemble_time_pointers_for_stations`, utils.py:602-622)
 * at (ipick, floor((tpick - t0) / dt)), those with |tpick - tlatent[e * tl_stride + tl_col]| < 2 eps kept, edge MLP on the rows of
 * s_rows [P, 30] (genie_assoc_fwd), mean, fc2: out [n_picks, 15]. t0 = dt_partition[0], dt = dt_partition[1] - dt_partition[0]: either
 * passed as numbers (dt_partition NULL) or read by the kernel from the caller's DEVICE array dt_partition [l_dt >= 2] (t0 / dt ignored:
 * a caller holding the partition on the device needs no read-back). tpick / phase_label fp32 [n_picks], ipick int32. An index outside
 * the table is clamped and reported through genie_index_flags. */
int genie_lslc_fwd(genie_ctx* ctx, int phase_head, const float* s_rows, const int32_t* a_edges, int64_t n_edges, int l_dt, float t0,
                   float dt, const float* dt_partition, float eps, const float* tlatent, int tl_stride, int tl_col, const float* tpick,
                   const int32_t* ipick, const float* phase_label, int n_picks, float* out, void* stream);

/* Backward of genie_lslc_fwd for training steps (round 3; k_lslc_bwd, forward recomputed per tile): d_out [n_picks, 15] -> the head's
 * fc1 / fc2 / PReLU-slope gradients ADDED into grad_blob (weight-mirror layout; zero it once before the two heads), the gradient of every
 * gathered s row in erow [n_picks * 10][32] and its product node in etgt [n_picks * 10] (-1 = dropped by the 2-eps filter, module.py:642-647).
 * genie_seg_rows adds those rows into d_s [P, 30] per product node, in the order of `order` (the edges sorted by etgt, stable): several
 * picks may gather the same node, and the sum must not depend on scheduling. part_scratch: genie_lslc_bwd_part_floats(n_picks) floats. */
size_t genie_lslc_bwd_part_floats(int n_picks);
int genie_lslc_bwd(genie_ctx* ctx, int phase_head, const float* s_rows, const int32_t* a_edges, int64_t n_edges, int l_dt, float t0,
                   float dt, const float* dt_partition, float eps, const float* tlatent, int tl_stride, int tl_col, const float* tpick,
                   const int32_t* ipick, const float* phase_label, int n_picks, const float* d_out, float* erow, int32_t* etgt,
                   float* part_scratch, float* grad_blob, void* stream);
int genie_seg_rows(const float* erow, const int32_t* etgt, const int32_t* order, int64_t n_edges, float* d_s, void* stream);

/* StationSourceAttentionMergedPhases (`Arrivals`, module.py:662-775; use_sparse = True, use_neighbor_assoc_edges = False) on
 * the device: out [n_src, n_arv, 2]. stime [n_src] (`tq_sample`), src_embed [n_src, 30] (`x_src`), trv_src [n_src, n_sta, 2]
 * (`trv_out_q`), arrival_p / arrival_s [n_arv, 15] (genie_lslc_fwd), tpick / phase_label fp32 [n_arv]. The picks grouped by station:
 * order [n_arv] = pick ids sorted by station (stable), and for each of the n_useg stations that have picks its id, the start of
 * its picks in `order` and their count. ctx_scratch: n_src * 192 floats; e0max_scratch: one int32 where the call leaves
 * `edge_index[0].max()` over the kept edges (:762-763), the pick index the reference uses for `self_link` / `null_link` (:764-765):
 * the null pick n_arv whenever some source has |stime| < 2 eps (the null pick of :715-718 then survives the time filter), otherwise
 * the largest pick index with a kept edge -- reproduced as the reference computes it. The segment softmax (:773) runs in streaming
 * form over chunks of 192 picks of a station, so a station may hold any number of picks. */
int genie_arrivals_fwd(genie_ctx* ctx, int n_src, const float* stime, const float* src_embed, const float* trv_src, int n_sta,
                       const float* arrival_p, const float* arrival_s, const float* tpick, const float* phase_label, int n_arv,
                       const int32_t* order, const int32_t* seg_sta, const int32_t* seg_start, const int32_t* seg_len, int n_useg,
                       float eps, float* ctx_scratch, int32_t* e0max_scratch, float* out, void* stream);

/* The same head inside a training step (train_GENIE_model.py:1786-1861 differentiates module.py:662-775). _train_fwd = the forward
 * above that also keeps, per (source, pick), the three normalised head aggregates and the softmax statistics in `save`
 * (genie_arrivals_train_save_floats(n_src, n_arv) floats); ctx_scratch / e0max_scratch must be kept for the backward as well.
 * genie_arrivals_bwd: d_out [n_src, n_arv, 2] -> d_src_embed [n_src, 30], d_arrival_p / d_arrival_s [n_arv, 15], and the gradients of
 * the head's 24 parameters ADDED into grad_blob (registry layout, genie_train_grad_floats() floats; the f_src_context_* entries are
 * written). The reference's per-call edge list is never built: queries / values are functions of (pick, source) and the two link
 * bits, so the backward is one pointwise pass over the (source, pick) targets (proj_1 / proj_2, d aggregate, the softmax's segment
 * term in closed form), one pass over the (source, station) pairs (forward of every entry recomputed, one sweep over the station's
 * targets, the edge MLPs backwards), and a per-source pass for the context MLP. No atomics, fixed summation order: bitwise
 * reproducible. scratch: genie_arrivals_bwd_scratch_floats(n_src, n_arv, n_useg) floats. */
int64_t genie_arrivals_train_save_floats(int n_src, int n_arv);
int genie_arrivals_train_fwd(genie_ctx* ctx, int n_src, const float* stime, const float* src_embed, const float* trv_src, int n_sta,
                             const float* arrival_p, const float* arrival_s, const float* tpick, const float* phase_label, int n_arv,
                             const int32_t* order, const int32_t* seg_sta, const int32_t* seg_start, const int32_t* seg_len, int n_useg,
                             float eps, float* ctx_scratch, int32_t* e0max_scratch, float* out, float* save, void* stream);
int64_t genie_arrivals_bwd_scratch_floats(int n_src, int n_arv, int n_useg);
int genie_arrivals_bwd(genie_ctx* ctx, int n_src, const float* stime, const float* src_embed, const float* trv_src, int n_sta,
                       const float* arrival_p, const float* arrival_s, const float* tpick, const float* phase_label, int n_arv,
                       const int32_t* order, const int32_t* seg_sta, const int32_t* seg_start, const int32_t* seg_len, int n_useg,
                       float eps, const float* ctx_scratch, const int32_t* e0max_scratch, const float* save, const float* d_out,
                       float* scratch, float* d_src_embed, float* d_arrival_p, float* d_arrival_s, float* grad_blob, void* stream);

/* Product-level CSRs of the irregular product graph of `use_subgraph: True` on the device (the two `subgraph(...)` loops of
 * extract_inputs_adjacencies_subgraph, process_utils.py:824-839). Product node n = the pair (pair_sta[n], pair_src[n]), pairs
 * sorted by (source, station) (:790-794); seg_rowptr[g] .. seg_rowptr[g+1] = the nodes of source node g; (sta_rowptr, sta_col) /
 * (src_rowptr, src_col) = in-edge CSRs of the base kNN graphs (:782-783). Two passes: _count fills the in-degrees
 * count_sta[n_prod] / count_src[n_prod]; the caller scans them into p_*_rowptr[n_prod + 1] (int32) and allocates the column
 * arrays; _fill writes the neighbours in base edge order. The results feed genie_ctx_create_subgraph directly. */
int genie_subgraph_csr_count(const int32_t* pair_sta, const int32_t* pair_src, int64_t n_prod, const int32_t* seg_rowptr,
                             const int32_t* sta_rowptr, const int32_t* sta_col, const int32_t* src_rowptr, const int32_t* src_col,
                             int32_t* count_sta, int32_t* count_src, void* stream);
int genie_subgraph_csr_fill(const int32_t* pair_sta, const int32_t* pair_src, int64_t n_prod, const int32_t* seg_rowptr,
                            const int32_t* sta_rowptr, const int32_t* sta_col, const int32_t* src_rowptr, const int32_t* src_col,
                            const int32_t* p_sta_rowptr, const int32_t* p_src_rowptr, int32_t* p_sta_col, int32_t* p_src_col,
                            void* stream);

/* Downstream reduction of the apply loop on the device (process_continuous_days.py:812-849): select entries of the stacked
 * output x [rows, cols] (fp32, row-major, e.g. Out_2 [n_query, len(tsteps_abs)]) without copying it to the host.
 *   mode 0: x > threshold                       = `np.where(Out_2 > 0.01)` (:812-813), row-major order
 *   mode 1: local maxima of every row with x >= threshold = scipy.signal.find_peaks(row, height = threshold) BEFORE its
 *           distance filter (:846): strict maxima and midpoints of flat tops, never the first / last sample
 * Two passes: genie_row_select_count fills counts[rows]; the caller turns them into exclusive offsets[rows] (int64) and
 * allocates the outputs; genie_row_select_fill writes (row, col, value) triplets in row-major order. */
int genie_row_select_count(const float* x, int rows, int64_t cols, float threshold, int mode, int32_t* counts, void* stream);
int genie_row_select_fill(const float* x, int rows, int64_t cols, float threshold, int mode, const int64_t* offsets,
                          int32_t* out_row, int32_t* out_col, float* out_val, void* stream);

/* Debug/parity access to intermediates kept in the workspace (which: 0 = c [P,30], 1 = wu [P,15], 2 = wv [P,15]);
 * copies de-padded rows into `out` (async). */
int genie_ws_export(genie_ctx* ctx, int which, void* ws, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GENIE_HIP_H */
