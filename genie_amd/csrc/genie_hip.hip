// genie_hip.hip — MI355X (gfx950 / CDNA4) kernels + C ABI for GENIE's station <-> source-grid
// message-passing hot path (see include/genie_hip.h for the boundary and the reference lines replaced).
//
// Design (DESIGN.md has the long form):
//  * product node p = g*S + s; a TILE is 16 consecutive stations of ONE source node g. Work items (g, tile) are swept by
//    persistent workgroups, XCD b%8 taking a contiguous chunk of a Morton order of the source grid.
//  * every per-node Linear is a matrix product D[ch, node] += W[ch, k] * X[k, node]: output channels are MFMA rows, nodes
//    are MFMA columns, so the accumulator of one layer IS the B operand of the next (weights are pre-permuted into that k
//    order, "A fragments", once per weight update and live in LDS; activations never leave registers between layers).
//  * stage 1 (k_stage1_h2, the default on the reference's 8 / 15-degree kNN graphs) runs on the 16-bit matrix pipe with every
//    fp32 operand split into two fp16 pieces (round to nearest: x within one fp32 ulp) and three partial products per product
//    (fp32 accumulation): fp32 MFMAs share the vector datapath on this hardware. Two tiles per wave, v_mfma_f32_32x32x16_f16.
//    The fp32 MFMA kernels (k_stage1, k_stage1_pcsr; v_mfma_f32_16x16x4_f32, lane (j = lane&15,
//    q = lane>>4) holds channels 16t+4q+{0..3} of node j) serve ragged / irregular graphs and use_absolute_pos.
//  * a neighbour's hidden state is RECOMPUTED from its raw input row instead of gathered (h0 is never stored), u / v are
//    projected through the neighbour-mean columns before they are averaged (64-B gather rows), and the node-local layer-2
//    terms are computed where h1 lives; stage 2 gathers, applies PReLU2 and the Bipartite message MLP, and reduces over the
//    stations of a tile; one partial row per tile, summed in fixed order (bitwise deterministic, no atomics).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <type_traits>
#include <string>
#include <vector>

#include "genie_hip.h"

// GENIE_TUNING=1 builds (tools/tune.py only) add run-time ablation switches (env GENIE_ABLATE) that SKIP parts of
// a kernel to attribute its time; they break the results and are compiled out of the product library.
#ifndef GENIE_TUNING
#define GENIE_TUNING 0
#endif
#if GENIE_TUNING
#define ABL(a, bit) (((a).abl >> (bit)) & 1)
#else
#define ABL(a, bit) 0
#endif

// read-once rows (c, Mask, edge_attr of stage 2) as non-temporal loads: measured SLOWER (0.354 vs 0.326 ms), off
#ifndef GENIE_S2_NT
#define GENIE_S2_NT 0
#endif
#if GENIE_S2_NT
#define GENIE_LD_STREAM(ptr) __builtin_nontemporal_load(ptr)
#else
#define GENIE_LD_STREAM(ptr) (*(ptr))
#endif

#ifndef GENIE_S2_WAVES
#define GENIE_S2_WAVES 2   // minimum waves per SIMD the register allocator of k_stage2_fast is held to (3 = 168 VGPRs with
                           // spills and more rows in flight than L2 keeps: 0.313 vs 0.302 ms, fabric reads +50 %)
#endif

#ifndef GENIE_H2_DEPTH
#define GENIE_H2_DEPTH 6   // k_stage1_h2: row loads in flight ahead of their use (3 .. 8 measured equal)
#endif

#ifndef GENIE_HOIST_WEIGHTS
#define GENIE_HOIST_WEIGHTS 0
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(GENIE_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));         \
    } while (0)

// ------------------------------------------------------------------------------------------------
// Weight registry: the path's parameters under the reference's state_dict names.
// ------------------------------------------------------------------------------------------------
struct Param {
    const char* name;
    int numel;
    int off;
};

enum {
    W_DA_INIT_W, W_DA_INIT_B, W_DA_L1T12_W, W_DA_L1T12_B, W_DA_L1T22_W, W_DA_L1T22_B,
    W_DA_L2T11_W, W_DA_L2T11_B, W_DA_L2T21_W, W_DA_L2T21_B, W_DA_L2T12_W, W_DA_L2T12_B,
    W_DA_L2T22_W, W_DA_L2T22_B, W_DA_ACT, W_DA_ACT11, W_DA_ACT12, W_DA_ACT1, W_DA_ACT21, W_DA_ACT22, W_DA_ACT2,
    W_BP_FC1_W, W_BP_FC1_B, W_BP_FC2_W, W_BP_FC2_B, W_BP_ACT1, W_BP_ACT2,
    W_SA1_FC1_W, W_SA1_FC1_B, W_SA1_FC2_W, W_SA1_FC2_B, W_SA1_FG_W, W_SA1_FG_B, W_SA1_ACT1, W_SA1_ACT2, W_SA1_ACT3,
    W_SA2_FC1_W, W_SA2_FC1_B, W_SA2_FC2_W, W_SA2_FC2_B, W_SA2_FG_W, W_SA2_FG_B, W_SA2_ACT1, W_SA2_ACT2, W_SA2_ACT3,
    W_SA3_FC1_W, W_SA3_FC1_B, W_SA3_FC2_W, W_SA3_FC2_B, W_SA3_FG_W, W_SA3_FG_B, W_SA3_ACT1, W_SA3_ACT2, W_SA3_ACT3,
    W_SD_W, W_SD_B, W_SD_ACT,
    W_TA_Q1_W, W_TA_Q1_B, W_TA_Q2_W, W_TA_Q2_B, W_TA_C1_W, W_TA_C1_B, W_TA_C2_W, W_TA_C2_B, W_TA_V1_W, W_TA_V1_B,
    W_TA_V2_W, W_TA_V2_B, W_TA_P1_W, W_TA_P1_B, W_TA_P2_W, W_TA_P2_B, W_TA_ACT1, W_TA_ACT2, W_TA_ACT3, W_TA_ACT4, W_TA_ACT5,
    W_SAT_Q_W, W_SAT_Q_B, W_SAT_C_W, W_SAT_C_B, W_SAT_V_W, W_SAT_V_B, W_SAT_P_W, W_SAT_P_B, W_SAT_ACT1, W_SAT_ACT2,
    // DataAggregationEdges (module.py:102-174): the 4 edge-feature columns of l1_t?_2 / l2_t?_2 (zero for DataAggregation)
    W_DA_L1T12_P, W_DA_L1T22_P, W_DA_L2T12_P, W_DA_L2T22_P,
    // use_absolute_pos (config.yaml:92): the 6 absolute-position columns of init_trns (zero otherwise)
    W_DA_INIT_ABS,
    // association heads (module.py:333-403): BipartiteGraphReadOutOperator + DataAggregationAssociationPhase
    W_RO_FC1_W, W_RO_FC1_B, W_RO_FC2_W, W_RO_FC2_B, W_RO_ACT1, W_RO_ACT2,
    W_AS_INIT_W, W_AS_INIT_B, W_AS_L1T11_W, W_AS_L1T11_B, W_AS_L1T21_W, W_AS_L1T21_B, W_AS_L1T12_W, W_AS_L1T12_B,
    W_AS_L1T22_W, W_AS_L1T22_B, W_AS_L2T11_W, W_AS_L2T11_B, W_AS_L2T21_W, W_AS_L2T21_B, W_AS_L2T12_W, W_AS_L2T12_B,
    W_AS_L2T22_W, W_AS_L2T22_B, W_AS_ACT, W_AS_ACT11, W_AS_ACT12, W_AS_ACT1, W_AS_ACT21, W_AS_ACT22, W_AS_ACT2,
    // pick-sized association heads (module.py:610-659): LocalSliceLgCollapse P / S
    W_LP_FC1_W, W_LP_FC1_B, W_LP_FC2_W, W_LP_FC2_B, W_LP_ACT1, W_LP_ACT2,
    W_LS_FC1_W, W_LS_FC1_B, W_LS_FC2_W, W_LS_FC2_B, W_LS_ACT1, W_LS_ACT2,
    // StationSourceAttentionMergedPhases (module.py:662-775, `Arrivals`)
    W_AR_Q1_W, W_AR_Q1_B, W_AR_Q2_W, W_AR_Q2_B, W_AR_C1_W, W_AR_C1_B, W_AR_C2_W, W_AR_C2_B, W_AR_V1_W, W_AR_V1_B, W_AR_V2_W, W_AR_V2_B,
    W_AR_P1_W, W_AR_P1_B, W_AR_P2_W, W_AR_P2_B, W_AR_ACT1, W_AR_ACT2, W_AR_ACT3, W_AR_ACT4,
    // DataAggregationAssociationPhaseEdges (module.py:407-480): the 4 edge-feature columns of l1_t?_2 / l2_t?_2, and the 6
    // absolute-position columns of its init_trns under use_absolute_pos (module.py:987-988); zero otherwise
    W_AS_L1T12_P, W_AS_L1T22_P, W_AS_L2T12_P, W_AS_L2T22_P, W_AS_INIT_ABS,
    W_COUNT
};

Param g_params[W_COUNT] = {
    {"DataAggregation.init_trns.weight", 30 * 8, 0}, {"DataAggregation.init_trns.bias", 30, 0},
    {"DataAggregation.l1_t1_2.weight", 30 * 64, 0}, {"DataAggregation.l1_t1_2.bias", 30, 0},
    {"DataAggregation.l1_t2_2.weight", 30 * 64, 0}, {"DataAggregation.l1_t2_2.bias", 30, 0},
    {"DataAggregation.l2_t1_1.weight", 30 * 60, 0}, {"DataAggregation.l2_t1_1.bias", 30, 0},
    {"DataAggregation.l2_t2_1.weight", 30 * 60, 0}, {"DataAggregation.l2_t2_1.bias", 30, 0},
    {"DataAggregation.l2_t1_2.weight", 15 * 94, 0}, {"DataAggregation.l2_t1_2.bias", 15, 0},
    {"DataAggregation.l2_t2_2.weight", 15 * 94, 0}, {"DataAggregation.l2_t2_2.bias", 15, 0},
    {"DataAggregation.activate.weight", 1, 0}, {"DataAggregation.activate11.weight", 1, 0},
    {"DataAggregation.activate12.weight", 1, 0}, {"DataAggregation.activate1.weight", 1, 0},
    {"DataAggregation.activate21.weight", 1, 0}, {"DataAggregation.activate22.weight", 1, 0},
    {"DataAggregation.activate2.weight", 1, 0},
    {"Bipartite_ReadIn.fc1.weight", 30 * 33, 0}, {"Bipartite_ReadIn.fc1.bias", 30, 0},
    {"Bipartite_ReadIn.fc2.weight", 15 * 30, 0}, {"Bipartite_ReadIn.fc2.bias", 15, 0},
    {"Bipartite_ReadIn.activate1.weight", 1, 0}, {"Bipartite_ReadIn.activate2.weight", 1, 0},
    {"SpatialAggregation1.fc1.weight", 30 * 23, 0}, {"SpatialAggregation1.fc1.bias", 30, 0},
    {"SpatialAggregation1.fc2.weight", 30 * 45, 0}, {"SpatialAggregation1.fc2.bias", 30, 0},
    {"SpatialAggregation1.fglobal.weight", 5 * 15, 0}, {"SpatialAggregation1.fglobal.bias", 5, 0},
    {"SpatialAggregation1.activate1.weight", 1, 0}, {"SpatialAggregation1.activate2.weight", 1, 0},
    {"SpatialAggregation1.activate3.weight", 1, 0},
    {"SpatialAggregation2.fc1.weight", 30 * 38, 0}, {"SpatialAggregation2.fc1.bias", 30, 0},
    {"SpatialAggregation2.fc2.weight", 30 * 60, 0}, {"SpatialAggregation2.fc2.bias", 30, 0},
    {"SpatialAggregation2.fglobal.weight", 5 * 30, 0}, {"SpatialAggregation2.fglobal.bias", 5, 0},
    {"SpatialAggregation2.activate1.weight", 1, 0}, {"SpatialAggregation2.activate2.weight", 1, 0},
    {"SpatialAggregation2.activate3.weight", 1, 0},
    {"SpatialAggregation3.fc1.weight", 30 * 38, 0}, {"SpatialAggregation3.fc1.bias", 30, 0},
    {"SpatialAggregation3.fc2.weight", 30 * 60, 0}, {"SpatialAggregation3.fc2.bias", 30, 0},
    {"SpatialAggregation3.fglobal.weight", 5 * 30, 0}, {"SpatialAggregation3.fglobal.bias", 5, 0},
    {"SpatialAggregation3.activate1.weight", 1, 0}, {"SpatialAggregation3.activate2.weight", 1, 0},
    {"SpatialAggregation3.activate3.weight", 1, 0},
    {"SpatialDirect.f_direct.weight", 30 * 30, 0}, {"SpatialDirect.f_direct.bias", 30, 0}, {"SpatialDirect.activate.weight", 1, 0},
    {"TemporalAttention.temporal_query_1.weight", 30, 0}, {"TemporalAttention.temporal_query_1.bias", 30, 0},
    {"TemporalAttention.temporal_query_2.weight", 75 * 30, 0}, {"TemporalAttention.temporal_query_2.bias", 75, 0},
    {"TemporalAttention.f_context_1.weight", 30 * 30, 0}, {"TemporalAttention.f_context_1.bias", 30, 0},
    {"TemporalAttention.f_context_2.weight", 75 * 30, 0}, {"TemporalAttention.f_context_2.bias", 75, 0},
    {"TemporalAttention.f_values_1.weight", 30 * 30, 0}, {"TemporalAttention.f_values_1.bias", 30, 0},
    {"TemporalAttention.f_values_2.weight", 75 * 30, 0}, {"TemporalAttention.f_values_2.bias", 75, 0},
    {"TemporalAttention.proj_1.weight", 30 * 15, 0}, {"TemporalAttention.proj_1.bias", 30, 0},
    {"TemporalAttention.proj_2.weight", 30, 0}, {"TemporalAttention.proj_2.bias", 1, 0},
    {"TemporalAttention.activate1.weight", 1, 0}, {"TemporalAttention.activate2.weight", 1, 0},
    {"TemporalAttention.activate3.weight", 1, 0}, {"TemporalAttention.activate4.weight", 1, 0},
    {"TemporalAttention.activate5.weight", 1, 0},
    {"SpatialAttention.f_queries.weight", 75 * 3, 0}, {"SpatialAttention.f_queries.bias", 75, 0},
    {"SpatialAttention.f_context.weight", 75 * 33, 0}, {"SpatialAttention.f_context.bias", 75, 0},
    {"SpatialAttention.f_values.weight", 75 * 33, 0}, {"SpatialAttention.f_values.bias", 75, 0},
    {"SpatialAttention.proj.weight", 30 * 15, 0}, {"SpatialAttention.proj.bias", 30, 0},
    {"SpatialAttention.activate1.weight", 1, 0}, {"SpatialAttention.activate2.weight", 1, 0},
    {"DataAggregation.l1_t1_2.weight_pos", 30 * 4, 0}, {"DataAggregation.l1_t2_2.weight_pos", 30 * 4, 0},
    {"DataAggregation.l2_t1_2.weight_pos", 15 * 4, 0}, {"DataAggregation.l2_t2_2.weight_pos", 15 * 4, 0},
    {"DataAggregation.init_trns.weight_abs", 30 * 6, 0},
    {"BipartiteGraphReadOutOperator.fc1.weight", 30 * 33, 0}, {"BipartiteGraphReadOutOperator.fc1.bias", 30, 0},
    {"BipartiteGraphReadOutOperator.fc2.weight", 15 * 30, 0}, {"BipartiteGraphReadOutOperator.fc2.bias", 15, 0},
    {"BipartiteGraphReadOutOperator.activate1.weight", 1, 0}, {"BipartiteGraphReadOutOperator.activate2.weight", 1, 0},
    {"DataAggregationAssociationPhase.init_trns.weight", 30 * 50, 0}, {"DataAggregationAssociationPhase.init_trns.bias", 30, 0},
    {"DataAggregationAssociationPhase.l1_t1_1.weight", 30 * 30, 0}, {"DataAggregationAssociationPhase.l1_t1_1.bias", 30, 0},
    {"DataAggregationAssociationPhase.l1_t2_1.weight", 30 * 30, 0}, {"DataAggregationAssociationPhase.l1_t2_1.bias", 30, 0},
    {"DataAggregationAssociationPhase.l1_t1_2.weight", 30 * 65, 0}, {"DataAggregationAssociationPhase.l1_t1_2.bias", 30, 0},
    {"DataAggregationAssociationPhase.l1_t2_2.weight", 30 * 65, 0}, {"DataAggregationAssociationPhase.l1_t2_2.bias", 30, 0},
    {"DataAggregationAssociationPhase.l2_t1_1.weight", 30 * 60, 0}, {"DataAggregationAssociationPhase.l2_t1_1.bias", 30, 0},
    {"DataAggregationAssociationPhase.l2_t2_1.weight", 30 * 60, 0}, {"DataAggregationAssociationPhase.l2_t2_1.bias", 30, 0},
    {"DataAggregationAssociationPhase.l2_t1_2.weight", 15 * 95, 0}, {"DataAggregationAssociationPhase.l2_t1_2.bias", 15, 0},
    {"DataAggregationAssociationPhase.l2_t2_2.weight", 15 * 95, 0}, {"DataAggregationAssociationPhase.l2_t2_2.bias", 15, 0},
    {"DataAggregationAssociationPhase.activate.weight", 1, 0}, {"DataAggregationAssociationPhase.activate11.weight", 1, 0},
    {"DataAggregationAssociationPhase.activate12.weight", 1, 0}, {"DataAggregationAssociationPhase.activate1.weight", 1, 0},
    {"DataAggregationAssociationPhase.activate21.weight", 1, 0}, {"DataAggregationAssociationPhase.activate22.weight", 1, 0},
    {"DataAggregationAssociationPhase.activate2.weight", 1, 0},
    {"LocalSliceLgCollapseP.fc1.weight", 30 * 32, 0}, {"LocalSliceLgCollapseP.fc1.bias", 30, 0},
    {"LocalSliceLgCollapseP.fc2.weight", 15 * 30, 0}, {"LocalSliceLgCollapseP.fc2.bias", 15, 0},
    {"LocalSliceLgCollapseP.activate1.weight", 1, 0}, {"LocalSliceLgCollapseP.activate2.weight", 1, 0},
    {"LocalSliceLgCollapseS.fc1.weight", 30 * 32, 0}, {"LocalSliceLgCollapseS.fc1.bias", 30, 0},
    {"LocalSliceLgCollapseS.fc2.weight", 15 * 30, 0}, {"LocalSliceLgCollapseS.fc2.bias", 15, 0},
    {"LocalSliceLgCollapseS.activate1.weight", 1, 0}, {"LocalSliceLgCollapseS.activate2.weight", 1, 0},
    {"Arrivals.f_arrival_query_1.weight", 30 * 36, 0}, {"Arrivals.f_arrival_query_1.bias", 30, 0},
    {"Arrivals.f_arrival_query_2.weight", 45 * 30, 0}, {"Arrivals.f_arrival_query_2.bias", 45, 0},
    {"Arrivals.f_src_context_1.weight", 30 * 33, 0}, {"Arrivals.f_src_context_1.bias", 30, 0},
    {"Arrivals.f_src_context_2.weight", 45 * 30, 0}, {"Arrivals.f_src_context_2.bias", 45, 0},
    {"Arrivals.f_values_1.weight", 30 * 38, 0}, {"Arrivals.f_values_1.bias", 30, 0},
    {"Arrivals.f_values_2.weight", 45 * 30, 0}, {"Arrivals.f_values_2.bias", 45, 0},
    {"Arrivals.proj_1.weight", 30 * 15, 0}, {"Arrivals.proj_1.bias", 30, 0},
    {"Arrivals.proj_2.weight", 2 * 30, 0}, {"Arrivals.proj_2.bias", 2, 0},
    {"Arrivals.activate1.weight", 1, 0}, {"Arrivals.activate2.weight", 1, 0},
    {"Arrivals.activate3.weight", 1, 0}, {"Arrivals.activate4.weight", 1, 0},
    {"DataAggregationAssociationPhase.l1_t1_2.weight_pos", 30 * 4, 0}, {"DataAggregationAssociationPhase.l1_t2_2.weight_pos", 30 * 4, 0},
    {"DataAggregationAssociationPhase.l2_t1_2.weight_pos", 15 * 4, 0}, {"DataAggregationAssociationPhase.l2_t2_2.weight_pos", 15 * 4, 0},
    {"DataAggregationAssociationPhase.init_trns.weight_abs", 30 * 6, 0},
};

int g_raw_total = 0;

void init_registry() {
    if (g_raw_total) return;
    int off = 0;
    for (int i = 0; i < W_COUNT; ++i) {
        g_params[i].off = off;
        off += (g_params[i].numel + 3) & ~3;  // keep every tensor 16-B aligned in the mirror
    }
    g_raw_total = off;
}

// ------------------------------------------------------------------------------------------------
// A-fragment packing. One MFMA "step" = one v_mfma_f32_16x16x4_f32: lane (i = lane&15, q = lane>>4)
// supplies A[i][q] = W[o0+i][col[q]]. Four steps (k-steps r = 0..3 of one 16-channel input block) form a
// GROUP stored as [lane][r] so a lane fetches its four A values with one ds_read_b128.
// ------------------------------------------------------------------------------------------------
struct StepDesc {
    int32_t mat_off;  // offset of W in the raw mirror, <0 = unused step (zeros)
    int32_t ld;       // input dimension of W
    int32_t o0;       // first output row of this 16-row tile
    int32_t rows;     // valid rows from o0 (<=16)
    int32_t col[4];   // input column supplied by lanes with q = 0..3, <0 = zero
    int32_t tr;       // 1 = the TRANSPOSE of W is applied (backward passes): A[i][q] = W[col[q]][o0 + i]
};
struct BiasDesc {
    int32_t off, o0, rows, pad;
};

struct StagePlan {
    std::vector<StepDesc> steps;  // 4 per group
    std::vector<BiasDesc> bias;   // one per 16-row output tile
    std::vector<int32_t> scal;    // raw offsets of PReLU slopes
    int n_groups() const { return (int)steps.size() / 4; }
    int packed_floats() const { return n_groups() * 256 + (int)bias.size() * 16 + 16; }
};

StepDesc unused_step() {
    StepDesc d;
    d.mat_off = -1; d.ld = 0; d.o0 = 0; d.rows = 0; d.tr = 0;
    d.col[0] = d.col[1] = d.col[2] = d.col[3] = -1;
    return d;
}

// group whose k-step r supplies input channel (c0 + 4q + r), valid while (4q+r) < nvalid
void add_block_group(StagePlan& p, int mat, int ld, int o0, int rows, int c0, int nvalid) {
    for (int r = 0; r < 4; ++r) {
        StepDesc d;
        d.mat_off = g_params[mat].off; d.ld = ld; d.o0 = o0; d.rows = rows; d.tr = 0;
        for (int q = 0; q < 4; ++q) d.col[q] = (4 * q + r < nvalid) ? (c0 + 4 * q + r) : -1;
        p.steps.push_back(d);
    }
}
// the same with W transposed (backward: dX = W^T dY): output element i = COLUMN (o0 + i) of W, i < rows; k-step r consumes
// ROW (c0 + 4q + r) of W, valid while (4q + r) < nvalid
void add_block_group_T(StagePlan& p, int mat, int ld, int o0, int rows, int c0, int nvalid) {
    for (int r = 0; r < 4; ++r) {
        StepDesc d;
        d.mat_off = g_params[mat].off; d.ld = ld; d.o0 = o0; d.rows = rows; d.tr = 1;
        for (int q = 0; q < 4; ++q) d.col[q] = (4 * q + r < nvalid) ? (c0 + 4 * q + r) : -1;
        p.steps.push_back(d);
    }
}
// group with explicit single steps: step r supplies column cols[r] + q for q < nq (else unused)
void add_scalar_group(StagePlan& p, int mat, int ld, int o0, int rows, const int* c0s, const int* nqs, int nsteps) {
    for (int r = 0; r < 4; ++r) {
        if (r >= nsteps) { p.steps.push_back(unused_step()); continue; }
        StepDesc d;
        d.mat_off = g_params[mat].off; d.ld = ld; d.o0 = o0; d.rows = rows; d.tr = 0;
        for (int q = 0; q < 4; ++q) d.col[q] = (q < nqs[r]) ? (c0s[r] + q) : -1;
        p.steps.push_back(d);
    }
}
void add_bias(StagePlan& p, int vec, int o0, int rows) {
    BiasDesc b; b.off = g_params[vec].off; b.o0 = o0; b.rows = rows; b.pad = 0;
    p.bias.push_back(b);
}

// Group index maps shared by host plan and device kernels --------------------------------------
// STAGE 1 (all dense work of DataAggregation up to the second pair of neighbour means)
//  init_trns (recomputed for the node itself AND for every gathered neighbour): out tile t; step 0 = Slice, 1 = Mask
#define G1_INIT(t) (t)
//  layer 1 (half h = l1_t1_2 / l1_t2_2, out tile t, input block b: 0,1 = h0; 2,3 = neighbour mean; 4 = Mask)
#define G1_L1(h, t, b) (2 + ((h) * 2 + (t)) * 5 + (b))
//  u / v (w = l2_t1_1 / l2_t2_1, out tile t, input block hb = h1 block 0..3)
#define G1_UV(w, t, hb) (22 + ((w) * 2 + (t)) * 4 + (hb))
//  projected gather operands wu = l2_t1_2[:, 60:90] u, wv = l2_t2_2[:, 60:90] v (w, input tile b of u / v)
#define G1_W(w, b) (38 + (w) * 2 + (b))
//  node-local part of layer 2: c_w = l2_t?_2[:, 0:60] h1 + l2_t?_2[:, 90:94] M + bias (b: 0..3 = h1 blocks, 4 = Mask)
#define G1_C(w, b) (42 + (w) * 5 + (b))
#define G1_GROUPS 52
//  bias tiles: 0,1 init_trns; 2..5 layer 1 (h,t); 6..9 u/v (w,t); 10,11 c (w)
#define G1_BIAS 12
// STAGE 2: bipartite fc1 (out tile t, b: 0 = o1 block, 1 = o2 block, 2 = edge_attr)
#define G2_BP(t, b) ((t) * 3 + (b))
#define G2_GROUPS 6
#define G2_BIAS 2

// ASSOCIATION stage A (k_assoc_a): BipartiteGraphReadOutOperator (module.py:343-352) + the per-node front of
// DataAggregationAssociationPhase (:389-396). fc1's edge_attr columns (out tile t); fc2 (input block b of the message);
// init_trns (out tile t; block 0 = s, 1,2 = x_latent, 3 = Mask; the mask1 column is a per-source-node term);
// l1_t1_1 / l1_t2_1 (w, out tile t, input block b of tr)
#define GA_FC1E(t) (t)
#define GA_FC2(b) (2 + (b))
#define GA_INIT(t, b) (4 + (t) * 4 + (b))
#define GA_Q(w, t, b) (12 + ((w) * 2 + (t)) * 2 + (b))
#define GA_GROUPS 20
//  bias tiles: 0 fc2; 1,2 init_trns; 3,4 l1_t1_1; 5,6 l1_t2_1
#define GA_BIAS 7
// ASSOCIATION stage B (k_assoc_b): layers 1-2 of DataAggregationAssociationPhase up to the second pair of neighbour means
// (:397-400), same structure as stage 1 of DataAggregation with 65- / 95-wide Linears (mask width 5; the mask1 column is a
// per-source-node term)
#define GB_L1(h, t, b) (((h) * 2 + (t)) * 5 + (b))
#define GB_UV(w, t, hb) (20 + ((w) * 2 + (t)) * 4 + (hb))
#define GB_W(w, b) (36 + (w) * 2 + (b))
#define GB_C(w, b) (40 + (w) * 5 + (b))
#define GB_GROUPS 50
//  bias tiles: 0..3 layer 1 (h,t); 4..7 u/v (w,t); 8,9 c (w)
#define GB_BIAS 10
// per-source-node terms of the association stages, AS_PG floats per source node: [0:30] fc1[:, 0:30] y_latent[g] + fc1 bias,
// [31] mask1[g]; mask1[g] x the mask1 column of init_trns [32:62], l1_t1_2 [64:94], l1_t2_2 [96:126], l2_t1_2 [128:143],
// l2_t2_2 [144:159]
constexpr int AS_PG = 160;

// STAGE 1 on the 16-bit matrix pipe (k_stage1_h2): 1-KB A fragments of v_mfma_f32_32x32x16_f16, lane (i = lane&31, h = lane>>5)
// holds 8 values = K slots (h, e = 0..7). Fragment ids: init_trns (2 fragments), then [block][K-step][piece], 2 pieces per K-step.
constexpr int H2_FA = 0;        // + m: [P|P], [Q|Q] of init_trns with P + Q = 16 W (K = two 8-wide input slices [x0 ; x1])
constexpr int H2_FL1 = 2;       // + ((t*4 + ks)*2 + piece): layer 1, half t, K-steps 0,1 = h0 block, 2,3 = mean block
constexpr int H2_FUVC = 18;     // + ((blk*4 + ks)*2 + piece): blk 0 = u, 1 = v, 2 = c; K-steps over h1 = [half 0 | half 1]
constexpr int H2_FW = 42;       // + (ks*2 + piece): [wu | wv] rows, K-steps 0,1 = u block, 2,3 = v block
constexpr int H2_FABS = 50;     // + m: [P|P], [Q|Q] of init_trns' absolute-position columns (use_absolute_pos; zero otherwise): K slots
                                // 0..2 = station position, 4..6 = source position of one piece (slots 3, 7 unused)
constexpr int H2_FRAGS = 52;
constexpr int H2_NBIAS = 6;     // init_trns, l1_t1_2, l1_t2_2, l2_t1_1, l2_t2_1, [l2_t1_2 | l2_t2_2]
constexpr int H2_IMG_FLOATS = H2_FRAGS * 256 + H2_NBIAS * 32 + 16;
constexpr int H2_TBL = H2_FRAGS * 512 + H2_NBIAS * 32 + 16;

// table entry: raw-mirror offset | piece << 28, or -1 for zero. Slot (h, e) of K-step kb (0/1) of a 32-channel block is
// channel 16 kb + 8 (e >> 2) + 4 h + (e & 3): registers 8kb..8kb+7 of the producing accumulator (see k_stage1_h2).
// Piece codes: 0 = W0 = rn16(W); 1 = rn16(16 (W - W0)) (the product it enters takes x0 / 16 as its other operand, which keeps
// the second piece out of fp16's subnormal range); 2 = rn16(16 W), 3 = rn16(16 W - piece 2): the input layer, computed 16 x too
// large as a whole.
void build_h2_table(std::vector<int32_t>& tbl) {
    constexpr int NP = 2;
    tbl.assign(H2_TBL, -1);
    auto put = [&](int f, int i, int h, int e, int piece, int off) {
        tbl[((size_t)f * 64 + (h * 32 + i)) * 8 + e] = off < 0 ? -1 : (off | (piece << 28));
    };
    for (int i = 0; i < 32; ++i)
        for (int h = 0; h < 2; ++h)
            for (int e = 0; e < 8; ++e) {
                const int off = i < 30 ? g_params[W_DA_INIT_W].off + i * 8 + e : -1;
                put(H2_FA + 0, i, h, e, 2, off);      // [P|P][x0;x1], [Q|Q][x0;x1] with P + Q = 16 W
                put(H2_FA + 1, i, h, e, 3, off);
                const int col = (e & 3) < 3 ? 3 * (e >> 2) + (e & 3) : -1;
                const int offa = (i < 30 && col >= 0) ? g_params[W_DA_INIT_ABS].off + i * 6 + col : -1;
                put(H2_FABS + 0, i, h, e, 2, offa);
                put(H2_FABS + 1, i, h, e, 3, offa);
            }
    // src(i, ch): raw offset of the weight multiplying channel ch (0..31) of the K-step's block into output row i
    auto dense = [&](int f0, int kb, auto src) {
        for (int i = 0; i < 32; ++i)
            for (int h = 0; h < 2; ++h)
                for (int e = 0; e < 8; ++e) {
                    const int ch = 16 * kb + 8 * (e >> 2) + 4 * h + (e & 3);
                    const int off = src(i, ch);
                    for (int piece = 0; piece < NP; ++piece) put(f0 + piece, i, h, e, piece, off);
                }
    };
    for (int t = 0; t < 2; ++t)
        for (int ks = 0; ks < 4; ++ks) {
            const int mat = g_params[t == 0 ? W_DA_L1T12_W : W_DA_L1T22_W].off, blk = ks >> 1;
            dense(H2_FL1 + (t * 4 + ks) * NP, ks & 1, [&](int i, int ch) {
                if (i >= 30) return -1;
                return mat + i * 64 + (ch < 30 ? 30 * blk + ch : 60 + 2 * blk + (ch - 30));   // pads: Mask columns
            });
        }
    for (int b = 0; b < 3; ++b)
        for (int ks = 0; ks < 4; ++ks) {
            const int half = ks >> 1;
            dense(H2_FUVC + (b * 4 + ks) * NP, ks & 1, [&](int i, int ch) {
                if (b < 2) {
                    if (i >= 30 || ch >= 30) return -1;
                    return g_params[b == 0 ? W_DA_L2T11_W : W_DA_L2T21_W].off + i * 60 + 30 * half + ch;
                }
                int mat, row;
                if (i < 15) { mat = g_params[W_DA_L2T12_W].off; row = i; }
                else if (i >= 16 && i < 31) { mat = g_params[W_DA_L2T22_W].off; row = i - 16; }
                else return -1;
                return mat + row * 94 + (ch < 30 ? 30 * half + ch : 90 + 2 * half + (ch - 30));
            });
        }
    for (int ks = 0; ks < 4; ++ks) {
        const int blk = ks >> 1;
        dense(H2_FW + ks * NP, ks & 1, [&](int i, int ch) {
            if (ch >= 30) return -1;
            if (blk == 0) return i < 15 ? g_params[W_DA_L2T12_W].off + i * 94 + 60 + ch : -1;
            return (i >= 16 && i < 31) ? g_params[W_DA_L2T22_W].off + (i - 16) * 94 + 60 + ch : -1;
        });
    }
    int32_t* bias = tbl.data() + (size_t)H2_FRAGS * 512;
    const int bvec[5] = {W_DA_INIT_B, W_DA_L1T12_B, W_DA_L1T22_B, W_DA_L2T11_B, W_DA_L2T21_B};
    for (int b = 0; b < 5; ++b)
        for (int i = 0; i < 30; ++i) bias[b * 32 + i] = g_params[bvec[b]].off + i;
    for (int i = 0; i < 15; ++i) {
        bias[5 * 32 + i] = g_params[W_DA_L2T12_B].off + i;
        bias[5 * 32 + 16 + i] = g_params[W_DA_L2T22_B].off + i;
    }
    int32_t* scal = bias + H2_NBIAS * 32;
    const int sv[6] = {W_DA_ACT, W_DA_ACT11, W_DA_ACT12, W_DA_ACT1, W_DA_ACT21, W_DA_ACT22};
    for (int k = 0; k < 6; ++k) scal[k] = g_params[sv[k]].off;
}

void build_plans(StagePlan& p1, StagePlan& p2) {
    // ---- stage 1
    for (int t = 0; t < 2; ++t) {
        const int c0s[2] = {0, 4}, nqs[2] = {4, 4};
        add_scalar_group(p1, W_DA_INIT_W, 8, 16 * t, std::min(16, 30 - 16 * t), c0s, nqs, 2);
        // steps 2, 3 of the group: station / source absolute-position columns (use_absolute_pos; zero weights otherwise)
        for (int r = 2; r < 4; ++r) {
            StepDesc& d = p1.steps[p1.steps.size() - 4 + r];
            d.mat_off = g_params[W_DA_INIT_ABS].off; d.ld = 6; d.o0 = 16 * t; d.rows = std::min(16, 30 - 16 * t); d.tr = 0;
            for (int q = 0; q < 4; ++q) d.col[q] = q < 3 ? 3 * (r - 2) + q : -1;
        }
    }
    for (int h = 0; h < 2; ++h)
        for (int t = 0; t < 2; ++t) {
            const int mat = h == 0 ? W_DA_L1T12_W : W_DA_L1T22_W;
            const int o0 = 16 * t, rows = std::min(16, 30 - 16 * t);
            add_block_group(p1, mat, 64, o0, rows, 0, 16);        // h0 ch 0..15
            add_block_group(p1, mat, 64, o0, rows, 16, 14);       // h0 ch 16..29
            add_block_group(p1, mat, 64, o0, rows, 30, 16);       // mean ch 0..15
            add_block_group(p1, mat, 64, o0, rows, 46, 14);       // mean ch 16..29
            const int c0s[1] = {60}, nqs[1] = {4};
            add_scalar_group(p1, mat, 64, o0, rows, c0s, nqs, 1);  // Mask
        }
    for (int w = 0; w < 2; ++w)
        for (int t = 0; t < 2; ++t) {
            const int mat = w == 0 ? W_DA_L2T11_W : W_DA_L2T21_W;
            const int o0 = 16 * t, rows = std::min(16, 30 - 16 * t);
            for (int hb = 0; hb < 4; ++hb)  // h1 block hb = (half, tile): channels half*30 + 16*tile + ...
                add_block_group(p1, mat, 60, o0, rows, (hb >> 1) * 30 + 16 * (hb & 1), (hb & 1) ? 14 : 16);
        }
    // mean_N(W x) = W mean_N(x): project u / v through the neighbour-mean columns of l2_t1_2 / l2_t2_2 (15 x 30) here,
    // so stage 2 gathers 16-float rows and adds their mean straight into its accumulator
    for (int w = 0; w < 2; ++w)
        for (int b = 0; b < 2; ++b)
            add_block_group(p1, w == 0 ? W_DA_L2T12_W : W_DA_L2T22_W, 94, 0, 15, 60 + 16 * b, b ? 14 : 16);
    // node-local part of layer 2 (h1 and Mask columns + bias), so h1 itself never leaves the registers
    for (int w = 0; w < 2; ++w) {
        const int mat = w == 0 ? W_DA_L2T12_W : W_DA_L2T22_W;
        for (int hb = 0; hb < 4; ++hb)
            add_block_group(p1, mat, 94, 0, 15, (hb >> 1) * 30 + 16 * (hb & 1), (hb & 1) ? 14 : 16);
        const int c0s[1] = {90}, nqs[1] = {4};
        add_scalar_group(p1, mat, 94, 0, 15, c0s, nqs, 1);
    }
    for (int t = 0; t < 2; ++t) add_bias(p1, W_DA_INIT_B, 16 * t, std::min(16, 30 - 16 * t));
    for (int h = 0; h < 2; ++h)
        for (int t = 0; t < 2; ++t) add_bias(p1, h == 0 ? W_DA_L1T12_B : W_DA_L1T22_B, 16 * t, std::min(16, 30 - 16 * t));
    for (int w = 0; w < 2; ++w)
        for (int t = 0; t < 2; ++t) add_bias(p1, w == 0 ? W_DA_L2T11_B : W_DA_L2T21_B, 16 * t, std::min(16, 30 - 16 * t));
    add_bias(p1, W_DA_L2T12_B, 0, 15);
    add_bias(p1, W_DA_L2T22_B, 0, 15);
    p1.scal.push_back(g_params[W_DA_ACT].off);
    p1.scal.push_back(g_params[W_DA_ACT11].off);
    p1.scal.push_back(g_params[W_DA_ACT12].off);
    p1.scal.push_back(g_params[W_DA_ACT1].off);
    p1.scal.push_back(g_params[W_DA_ACT21].off);
    p1.scal.push_back(g_params[W_DA_ACT22].off);
    // ---- stage 2
    for (int t = 0; t < 2; ++t) {
        const int o0 = 16 * t, rows = std::min(16, 30 - 16 * t);
        add_block_group(p2, W_BP_FC1_W, 33, o0, rows, 0, 15);    // o1 = x_latent[0:15]
        add_block_group(p2, W_BP_FC1_W, 33, o0, rows, 15, 15);   // o2 = x_latent[15:30]
        const int c0s[1] = {30}, nqs[1] = {3};
        add_scalar_group(p2, W_BP_FC1_W, 33, o0, rows, c0s, nqs, 1);  // edge_attr (3)
    }
    add_bias(p2, W_BP_FC1_B, 0, 16);
    add_bias(p2, W_BP_FC1_B, 16, 14);
    p2.scal.push_back(g_params[W_DA_ACT2].off);
    p2.scal.push_back(g_params[W_BP_ACT1].off);
}

void build_assoc_plans(StagePlan& pa, StagePlan& pb) {
    const int c1[1] = {30}, n3[1] = {3};
    for (int t = 0; t < 2; ++t) add_scalar_group(pa, W_RO_FC1_W, 33, 16 * t, std::min(16, 30 - 16 * t), c1, n3, 1);   // edge_attr columns
    add_block_group(pa, W_RO_FC2_W, 30, 0, 15, 0, 16);
    add_block_group(pa, W_RO_FC2_W, 30, 0, 15, 16, 14);
    for (int t = 0; t < 2; ++t) {
        const int o0 = 16 * t, rows = std::min(16, 30 - 16 * t);
        add_block_group(pa, W_AS_INIT_W, 50, o0, rows, 0, 15);      // s
        add_block_group(pa, W_AS_INIT_W, 50, o0, rows, 15, 16);     // x_latent 0..15
        add_block_group(pa, W_AS_INIT_W, 50, o0, rows, 31, 14);     // x_latent 16..29
        const int c0[1] = {46}, n4[1] = {4};
        add_scalar_group(pa, W_AS_INIT_W, 50, o0, rows, c0, n4, 1); // Mask (column 45 = mask1: per-source-node term)
    }
    for (int w = 0; w < 2; ++w)
        for (int t = 0; t < 2; ++t) {
            const int mat = w == 0 ? W_AS_L1T11_W : W_AS_L1T21_W;
            add_block_group(pa, mat, 30, 16 * t, std::min(16, 30 - 16 * t), 0, 16);
            add_block_group(pa, mat, 30, 16 * t, std::min(16, 30 - 16 * t), 16, 14);
        }
    add_bias(pa, W_RO_FC2_B, 0, 15);
    for (int t = 0; t < 2; ++t) add_bias(pa, W_AS_INIT_B, 16 * t, std::min(16, 30 - 16 * t));
    for (int t = 0; t < 2; ++t) add_bias(pa, W_AS_L1T11_B, 16 * t, std::min(16, 30 - 16 * t));
    for (int t = 0; t < 2; ++t) add_bias(pa, W_AS_L1T21_B, 16 * t, std::min(16, 30 - 16 * t));
    const int sa[5] = {W_RO_ACT1, W_RO_ACT2, W_AS_ACT, W_AS_ACT11, W_AS_ACT12};
    for (int k = 0; k < 5; ++k) pa.scal.push_back(g_params[sa[k]].off);
    // ---- stage B
    for (int h = 0; h < 2; ++h)
        for (int t = 0; t < 2; ++t) {
            const int mat = h == 0 ? W_AS_L1T12_W : W_AS_L1T22_W;
            const int o0 = 16 * t, rows = std::min(16, 30 - 16 * t);
            add_block_group(pb, mat, 65, o0, rows, 0, 16);
            add_block_group(pb, mat, 65, o0, rows, 16, 14);
            add_block_group(pb, mat, 65, o0, rows, 30, 16);
            add_block_group(pb, mat, 65, o0, rows, 46, 14);
            const int c0[1] = {61}, n4[1] = {4};
            add_scalar_group(pb, mat, 65, o0, rows, c0, n4, 1);
        }
    for (int w = 0; w < 2; ++w)
        for (int t = 0; t < 2; ++t) {
            const int mat = w == 0 ? W_AS_L2T11_W : W_AS_L2T21_W;
            for (int hb = 0; hb < 4; ++hb)
                add_block_group(pb, mat, 60, 16 * t, std::min(16, 30 - 16 * t), (hb >> 1) * 30 + 16 * (hb & 1), (hb & 1) ? 14 : 16);
        }
    for (int w = 0; w < 2; ++w)
        for (int b = 0; b < 2; ++b) add_block_group(pb, w == 0 ? W_AS_L2T12_W : W_AS_L2T22_W, 95, 0, 15, 60 + 16 * b, b ? 14 : 16);
    for (int w = 0; w < 2; ++w) {
        const int mat = w == 0 ? W_AS_L2T12_W : W_AS_L2T22_W;
        for (int hb = 0; hb < 4; ++hb) add_block_group(pb, mat, 95, 0, 15, (hb >> 1) * 30 + 16 * (hb & 1), (hb & 1) ? 14 : 16);
        const int c0[1] = {91}, n4[1] = {4};
        add_scalar_group(pb, mat, 95, 0, 15, c0, n4, 1);
    }
    for (int h = 0; h < 2; ++h)
        for (int t = 0; t < 2; ++t) add_bias(pb, h == 0 ? W_AS_L1T12_B : W_AS_L1T22_B, 16 * t, std::min(16, 30 - 16 * t));
    for (int w = 0; w < 2; ++w)
        for (int t = 0; t < 2; ++t) add_bias(pb, w == 0 ? W_AS_L2T11_B : W_AS_L2T21_B, 16 * t, std::min(16, 30 - 16 * t));
    add_bias(pb, W_AS_L2T12_B, 0, 15);
    add_bias(pb, W_AS_L2T22_B, 0, 15);
    pb.scal.push_back(g_params[W_AS_ACT1].off);
    pb.scal.push_back(g_params[W_AS_ACT21].off);
    pb.scal.push_back(g_params[W_AS_ACT22].off);
}

// Backward of DataAggregation + Bipartite_ReadIn (training) as three P-sized passes that mirror the forward
// stages in reverse (k_train_b2 / k_train_b1 / k_train_b0). Their dX chains are W^T products in the forward's MFMA layout
// (transposed A fragments); group index maps:
//  B2: x_latent gradient from the Bipartite message gradient: GT2(b, t) out block b of x_latent, in block t of dz
#define GT2(b, t) ((b) * 2 + (t))
#define GT2_GROUPS 4
//  B1: du / dv from the transposed-mean of do1 / do2; dh1 block hb from {du0, du1, dv0, dv1, do1, do2}; node-local part of
//      dh0 from dt = [dt1a dt1b dt2a dt2b]
#define GT_U(b) (b)
#define GT_V(b) (2 + (b))
#define GT_H(hb, src) (4 + (hb) * 6 + (src))
#define GT_D(b, k) (28 + (b) * 4 + (k))
#define GT1_GROUPS 36
//  B0: gradient of PReLU11(h0) / PReLU12(h0) from the transposed means of dt1 / dt2 (half h, out block b, in block k)
#define GT_Q(h, b, k) ((h) * 4 + (b) * 2 + (k))
#define GT0_GROUPS 8

void build_train_plans(StagePlan& p2, StagePlan& p1, StagePlan& p0) {
    for (int b = 0; b < 2; ++b)
        for (int t = 0; t < 2; ++t) add_block_group_T(p2, W_BP_FC1_W, 33, 15 * b, 15, 16 * t, t ? 14 : 16);
    p2.scal.push_back(g_params[W_DA_ACT2].off);
    p2.scal.push_back(g_params[W_BP_ACT1].off);
    for (int b = 0; b < 2; ++b) add_block_group_T(p1, W_DA_L2T12_W, 94, 60 + 16 * b, b ? 14 : 16, 0, 15);
    for (int b = 0; b < 2; ++b) add_block_group_T(p1, W_DA_L2T22_W, 94, 60 + 16 * b, b ? 14 : 16, 0, 15);
    for (int hb = 0; hb < 4; ++hb) {
        const int col0 = (hb >> 1) * 30 + 16 * (hb & 1), rows = (hb & 1) ? 14 : 16;
        for (int src = 0; src < 2; ++src) add_block_group_T(p1, W_DA_L2T11_W, 60, col0, rows, 16 * src, src ? 14 : 16);
        for (int src = 0; src < 2; ++src) add_block_group_T(p1, W_DA_L2T21_W, 60, col0, rows, 16 * src, src ? 14 : 16);
        add_block_group_T(p1, W_DA_L2T12_W, 94, col0, rows, 0, 15);
        add_block_group_T(p1, W_DA_L2T22_W, 94, col0, rows, 0, 15);
    }
    for (int b = 0; b < 2; ++b)
        for (int k = 0; k < 4; ++k)
            add_block_group_T(p1, k < 2 ? W_DA_L1T12_W : W_DA_L1T22_W, 64, 16 * b, b ? 14 : 16, 16 * (k & 1), (k & 1) ? 14 : 16);
    p1.scal.push_back(g_params[W_DA_ACT1].off);
    p1.scal.push_back(g_params[W_DA_ACT21].off);
    p1.scal.push_back(g_params[W_DA_ACT22].off);
    for (int h = 0; h < 2; ++h)
        for (int b = 0; b < 2; ++b)
            for (int k = 0; k < 2; ++k)
                add_block_group_T(p0, h == 0 ? W_DA_L1T12_W : W_DA_L1T22_W, 64, 30 + 16 * b, b ? 14 : 16, 16 * k, k ? 14 : 16);
    p0.scal.push_back(g_params[W_DA_ACT].off);
    p0.scal.push_back(g_params[W_DA_ACT11].off);
    p0.scal.push_back(g_params[W_DA_ACT12].off);
}

// G- / Q-sized tail on fp32 MFMA tiles (k_bip_out_m, k_sa_pre_m, k_sa_layer_m, k_ro_pre_m, k_readout_m): a wave owns 16 nodes,
// lane (j = lane&15, q = lane>>4) holds channels 16t + 4q + {0..3} of node j exactly as in the stage kernels, every per-node
// Linear is a chain of v_mfma_f32_16x16x4_f32 whose A fragments come from a k_pack_all image in LDS (one ds_read_b128 per 16 x 16
// weight block and wave instead of two LDS reads per scalar FMA) and whose result is the B operand of the next Linear.
// Plans (genie_ctx::plan[PL_*]) and their group index maps:
enum { PL_RO0 = 7, PL_RO1, PL_ROP, PL_SA1, PL_SA2, PL_SA3, PL_BIP, PL_LSP, PL_LSS, PL_ARR,
       // transposed weights of the tail's backward passes (train_tail_kernels.hpp)
       PL_TRO0, PL_TRO1, PL_TSN, PL_TSA1, PL_TSA2, PL_TSA3, PL_TBIP,
       // ... and of the association heads' backward passes (train_assoc_kernels.hpp)
       PL_TAB2, PL_TAB1, PL_TAB0, PL_TAG, PL_TLSP, PL_TLSS, PL_TARR, NPLAN };
// gradient maps (k_train_reduce): the three P-sized passes, then the tail's backward kernels
enum { TM_B2 = 0, TM_B1, TM_B0, TM_RO0, TM_RO1, TM_SN, TM_SAA1, TM_SAA2, TM_SAA3, TM_SAB1, TM_SAB2, TM_SAB3, TM_BIP,
       TM_AB3, TM_AB2, TM_AB1, TM_AB0, TM_AG, TM_LSP, TM_LSS, TM_ART, TM_ARE, NTM };
//  read-out heads (module.py:251-331), one image per MODE with the same map. FRONT: MODE 0 = SpatialDirect.f_direct (out tile t,
//  in block b), MODE 1 = SpatialAttention.proj (out tile t, b = 0; b = 1 unused). TemporalAttention: f_context_1 / f_values_1
//  (t, b), f_context_2 / f_values_2 with one out tile per HEAD h (rows 15h .. 15h+14, row 15 of the tile zero), proj_1 (t)
#define GR_FRONT(t, b) ((t) * 2 + (b))
#define GR_C1(t, b) (4 + (t) * 2 + (b))
#define GR_V1(t, b) (8 + (t) * 2 + (b))
#define GR_C2(h, b) (12 + (h) * 2 + (b))
#define GR_V2(h, b) (22 + (h) * 2 + (b))
#define GR_P1(t) (32 + (t))
#define GR_GROUPS 34
//  bias tiles: 0,1 front; 2,3 f_context_1; 4,5 f_values_1; 6..10 f_context_2 (head); 11..15 f_values_2 (head); 16,17 proj_1;
//  18,19 the proj_2 weight row. Scalars: 0 front PReLU, 1 SpatialAttention.activate1, 2..5 TemporalAttention.activate1/2/4/5,
//  6 proj_2.bias, 7 TemporalAttention.activate3
#define GR_BIAS 20
constexpr int GR_IMG_FLOATS = GR_GROUPS * 256 + GR_BIAS * 16 + 16;
//  k_ro_pre_m: per-grid-node parts of SpatialAttention's f_context (m = 0) / f_values (m = 1), head h, input block b; bias tiles m*5+h
#define GP(m, h, b) (((m) * 5 + (h)) * 2 + (b))
#define GP_GROUPS 20
#define GP_BIAS 10
constexpr int GP_IMG_FLOATS = GP_GROUPS * 256 + GP_BIAS * 16 + 16;
//  SpatialAggregation layer L (module.py:243-249): fc2 (out tile t; b = 0,1: x_i blocks, 2,3: edge-mean blocks), the NEXT layer's
//  fc1[:, 0:30] and fglobal (layers 1, 2), this layer's own fc1[:, 0:C] and fglobal (the pre-pass k_sa_pre_m)
#define GS_FC2(t, b) ((t) * 4 + (b))
#define GS_PJN(t, b) (8 + (t) * 2 + (b))
#define GS_FGN(b) (12 + (b))
#define GS_PJ(t, b) (14 + (t) * 2 + (b))
#define GS_FG(b) (18 + (b))
#define GS_GROUPS 20
//  bias tiles: 0,1 fc2; 2,3 fc1 (message bias); 4 next fglobal; 5 own fglobal. Scalars: 0 act1, 1 act2, 2 next act3, 3 own act3
#define GS_BIAS 6
constexpr int GS_IMG_FLOATS = GS_GROUPS * 256 + GS_BIAS * 16 + 16;
//  Bipartite read-out fc2 (15 x 30): input block b; bias tile 0; scalar 0 = activate2
#define GB_GROUPS2 2
#define GB_BIAS2 1
constexpr int GB2_IMG_FLOATS = GB_GROUPS2 * 256 + GB_BIAS2 * 16 + 16;

//  LocalSliceLgCollapse P / S (module.py:610-659): fc1 (out tile t; b = 0,1: the gathered s row, 2: [relative time, phase]),
//  fc2 (input block b); bias tiles 0,1 fc1, 2 fc2; scalars activate1, activate2
#define GL_FC1(t, b) ((t) * 3 + (b))
#define GL_FC2(b) (6 + (b))
#define GL_GROUPS 8
#define GL_BIAS 3
constexpr int GL_IMG_FLOATS = GL_GROUPS * 256 + GL_BIAS * 16 + 16;

//  StationSourceAttentionMergedPhases (module.py:662-775): f_arrival_query_1 / f_values_1 (out tile t; b = 0: arrival_p row, 1:
//  arrival_s row, 2: the six relative-time features, two k-steps; f_values_1 b = 3: [self_link, null_link]), f_arrival_query_2 /
//  f_values_2 (one out tile per head h, input block b), proj_1 (t)
#define GA_Q1(t, b) ((t) * 3 + (b))
#define GA_V1(t, b) (6 + (t) * 4 + (b))
#define GA_Q2(h, b) (14 + (h) * 2 + (b))
#define GA_V2(h, b) (20 + (h) * 2 + (b))
#define GA_P1(t) (26 + (t))
#define GA_GROUPS2 28
//  bias tiles: 0,1 query_1; 2,3 values_1; 4..6 query_2 (head); 7..9 values_2 (head); 10,11 proj_1; 12..15 proj_2 weight rows
//  (output m, tile t: 12 + 2m + t). Scalars: 0 activate2 (query), 1 activate3 (values), 2 activate4, 3,4 proj_2.bias
#define GA_BIAS2 16
constexpr int GA2_IMG_FLOATS = GA_GROUPS2 * 256 + GA_BIAS2 * 16 + 16;

void add_unused_group(StagePlan& p) {
    for (int r = 0; r < 4; ++r) p.steps.push_back(unused_step());
}

void build_tail_plans(StagePlan* plan) {
    auto rows2 = [](int t) { return t ? 14 : 16; };
    for (int mode = 0; mode < 2; ++mode) {
        StagePlan& p = plan[mode == 0 ? PL_RO0 : PL_RO1];
        for (int t = 0; t < 2; ++t) {
            if (mode == 0) {
                add_block_group(p, W_SD_W, 30, 16 * t, rows2(t), 0, 16);
                add_block_group(p, W_SD_W, 30, 16 * t, rows2(t), 16, 14);
            } else {
                add_block_group(p, W_SAT_P_W, 15, 16 * t, rows2(t), 0, 15);
                add_unused_group(p);
            }
        }
        for (int m = 0; m < 2; ++m)
            for (int t = 0; t < 2; ++t)
                for (int b = 0; b < 2; ++b) add_block_group(p, m == 0 ? W_TA_C1_W : W_TA_V1_W, 30, 16 * t, rows2(t), 16 * b, rows2(b));
        for (int m = 0; m < 2; ++m)
            for (int h = 0; h < 5; ++h)
                for (int b = 0; b < 2; ++b) add_block_group(p, m == 0 ? W_TA_C2_W : W_TA_V2_W, 30, 15 * h, 15, 16 * b, rows2(b));
        for (int t = 0; t < 2; ++t) add_block_group(p, W_TA_P1_W, 15, 16 * t, rows2(t), 0, 15);
        for (int t = 0; t < 2; ++t) add_bias(p, mode == 0 ? W_SD_B : W_SAT_P_B, 16 * t, rows2(t));
        for (int t = 0; t < 2; ++t) add_bias(p, W_TA_C1_B, 16 * t, rows2(t));
        for (int t = 0; t < 2; ++t) add_bias(p, W_TA_V1_B, 16 * t, rows2(t));
        for (int h = 0; h < 5; ++h) add_bias(p, W_TA_C2_B, 15 * h, 15);
        for (int h = 0; h < 5; ++h) add_bias(p, W_TA_V2_B, 15 * h, 15);
        for (int t = 0; t < 2; ++t) add_bias(p, W_TA_P1_B, 16 * t, rows2(t));
        for (int t = 0; t < 2; ++t) add_bias(p, W_TA_P2_W, 16 * t, rows2(t));
        const int sc[8] = {mode == 0 ? W_SD_ACT : W_SAT_ACT2, W_SAT_ACT1, W_TA_ACT1, W_TA_ACT2, W_TA_ACT4, W_TA_ACT5, W_TA_P2_B, W_TA_ACT3};
        for (int k = 0; k < 8; ++k) p.scal.push_back(g_params[sc[k]].off);
    }
    {
        StagePlan& p = plan[PL_ROP];
        for (int m = 0; m < 2; ++m)
            for (int h = 0; h < 5; ++h)
                for (int b = 0; b < 2; ++b) add_block_group(p, m == 0 ? W_SAT_C_W : W_SAT_V_W, 33, 15 * h, 15, 16 * b, rows2(b));
        for (int m = 0; m < 2; ++m)
            for (int h = 0; h < 5; ++h) add_bias(p, m == 0 ? W_SAT_C_B : W_SAT_V_B, 15 * h, 15);
        p.scal.push_back(g_params[W_SAT_ACT1].off);      // unused (a plan carries at least one scalar)
    }
    for (int layer = 1; layer <= 3; ++layer) {
        StagePlan& p = plan[PL_SA1 + layer - 1];
        const int base = layer == 1 ? W_SA1_FC1_W : (layer == 2 ? W_SA2_FC1_W : W_SA3_FC1_W);
        const int C = layer == 1 ? 15 : 30;
        for (int t = 0; t < 2; ++t) {
            if (C == 15) { add_block_group(p, base + 2, C + 30, 16 * t, rows2(t), 0, 15); add_unused_group(p); }
            else { add_block_group(p, base + 2, C + 30, 16 * t, rows2(t), 0, 16); add_block_group(p, base + 2, C + 30, 16 * t, rows2(t), 16, 14); }
            add_block_group(p, base + 2, C + 30, 16 * t, rows2(t), C, 16);
            add_block_group(p, base + 2, C + 30, 16 * t, rows2(t), C + 16, 14);
        }
        if (layer < 3) {
            const int nb = layer == 1 ? W_SA2_FC1_W : W_SA3_FC1_W;
            for (int t = 0; t < 2; ++t)
                for (int b = 0; b < 2; ++b) add_block_group(p, nb, 38, 16 * t, rows2(t), 16 * b, rows2(b));
            for (int b = 0; b < 2; ++b) add_block_group(p, nb + 4, 30, 0, 5, 16 * b, rows2(b));
        } else {
            for (int k = 0; k < 6; ++k) add_unused_group(p);
        }
        for (int t = 0; t < 2; ++t) {
            if (C == 15) { add_block_group(p, base, C + 8, 16 * t, rows2(t), 0, 15); add_unused_group(p); }
            else { add_block_group(p, base, C + 8, 16 * t, rows2(t), 0, 16); add_block_group(p, base, C + 8, 16 * t, rows2(t), 16, 14); }
        }
        if (C == 15) { add_block_group(p, base + 4, C, 0, 5, 0, 15); add_unused_group(p); }
        else { add_block_group(p, base + 4, C, 0, 5, 0, 16); add_block_group(p, base + 4, C, 0, 5, 16, 14); }
        for (int t = 0; t < 2; ++t) add_bias(p, base + 3, 16 * t, rows2(t));
        for (int t = 0; t < 2; ++t) add_bias(p, base + 1, 16 * t, rows2(t));
        if (layer < 3) add_bias(p, (layer == 1 ? W_SA2_FC1_W : W_SA3_FC1_W) + 5, 0, 5);
        else add_bias(p, base + 5, 0, 0);
        add_bias(p, base + 5, 0, 5);
        p.scal.push_back(g_params[base + 6].off);
        p.scal.push_back(g_params[base + 7].off);
        p.scal.push_back(g_params[(layer < 3 ? (layer == 1 ? W_SA2_FC1_W : W_SA3_FC1_W) : base) + 8].off);
        p.scal.push_back(g_params[base + 8].off);
    }
    for (int ph = 0; ph < 2; ++ph) {
        StagePlan& p = plan[ph == 0 ? PL_LSP : PL_LSS];
        const int base = ph == 0 ? W_LP_FC1_W : W_LS_FC1_W;
        for (int t = 0; t < 2; ++t) {
            add_block_group(p, base, 32, 16 * t, rows2(t), 0, 16);
            add_block_group(p, base, 32, 16 * t, rows2(t), 16, 14);
            const int c0[1] = {30}, n2[1] = {2};
            add_scalar_group(p, base, 32, 16 * t, rows2(t), c0, n2, 1);
        }
        for (int b = 0; b < 2; ++b) add_block_group(p, base + 2, 30, 0, 15, 16 * b, rows2(b));
        for (int t = 0; t < 2; ++t) add_bias(p, base + 1, 16 * t, rows2(t));
        add_bias(p, base + 3, 0, 15);
        p.scal.push_back(g_params[base + 4].off);
        p.scal.push_back(g_params[base + 5].off);
    }
    {
        StagePlan& p = plan[PL_ARR];
        for (int m = 0; m < 2; ++m) {            // f_arrival_query_1 (36 inputs), f_values_1 (38 inputs)
            const int mat = m == 0 ? W_AR_Q1_W : W_AR_V1_W, ld = m == 0 ? 36 : 38;
            for (int t = 0; t < 2; ++t) {
                add_block_group(p, mat, ld, 16 * t, rows2(t), 0, 15);
                add_block_group(p, mat, ld, 16 * t, rows2(t), 15, 15);
                const int c6[2] = {30, 34}, n6[2] = {4, 2};
                add_scalar_group(p, mat, ld, 16 * t, rows2(t), c6, n6, 2);
                if (m == 1) { const int c2[1] = {36}, n2[1] = {2}; add_scalar_group(p, mat, ld, 16 * t, rows2(t), c2, n2, 1); }
            }
        }
        for (int m = 0; m < 2; ++m)
            for (int h = 0; h < 3; ++h)
                for (int b = 0; b < 2; ++b) add_block_group(p, m == 0 ? W_AR_Q2_W : W_AR_V2_W, 30, 15 * h, 15, 16 * b, rows2(b));
        for (int t = 0; t < 2; ++t) add_block_group(p, W_AR_P1_W, 15, 16 * t, rows2(t), 0, 15);
        for (int t = 0; t < 2; ++t) add_bias(p, W_AR_Q1_B, 16 * t, rows2(t));
        for (int t = 0; t < 2; ++t) add_bias(p, W_AR_V1_B, 16 * t, rows2(t));
        for (int h = 0; h < 3; ++h) add_bias(p, W_AR_Q2_B, 15 * h, 15);
        for (int h = 0; h < 3; ++h) add_bias(p, W_AR_V2_B, 15 * h, 15);
        for (int t = 0; t < 2; ++t) add_bias(p, W_AR_P1_B, 16 * t, rows2(t));
        for (int m = 0; m < 2; ++m)
            for (int t = 0; t < 2; ++t) add_bias(p, W_AR_P2_W, 30 * m + 16 * t, rows2(t));
        p.scal.push_back(g_params[W_AR_ACT2].off);
        p.scal.push_back(g_params[W_AR_ACT3].off);
        p.scal.push_back(g_params[W_AR_ACT4].off);
        p.scal.push_back(g_params[W_AR_P2_B].off);
        p.scal.push_back(g_params[W_AR_P2_B].off + 1);
    }
    {
        StagePlan& p = plan[PL_BIP];
        for (int b = 0; b < 2; ++b) add_block_group(p, W_BP_FC2_W, 30, 0, 15, 16 * b, rows2(b));
        add_bias(p, W_BP_FC2_B, 0, 15);
        p.scal.push_back(g_params[W_BP_ACT2].off);
    }
}

// every plan of a context in ONE launch (a training step re-packs all ~30 images after each optimizer step): block -> plan by
// the plans' block offsets
struct PackPlan { const StepDesc* steps; const BiasDesc* bias; const int32_t* scal; float* out; int n_groups, n_bias, n_scal, block0; };
__device__ __forceinline__ void pack_one(const float* __restrict__ raw, const PackPlan& pl, int idx) {
    const int nw = pl.n_groups * 256;
    if (idx < nw) {
        const int grp = idx >> 8, lane = (idx & 255) >> 2, r = idx & 3;
        const StepDesc d = pl.steps[grp * 4 + r];
        const int i = lane & 15, q = lane >> 4;
        float v = 0.f;
        if (d.mat_off >= 0 && i < d.rows && d.col[q] >= 0)
            v = d.tr ? raw[d.mat_off + d.col[q] * d.ld + (d.o0 + i)] : raw[d.mat_off + (d.o0 + i) * d.ld + d.col[q]];
        pl.out[idx] = v;
    } else if (idx < nw + pl.n_bias * 16) {
        const int k = idx - nw, t = k >> 4, i = k & 15;
        const BiasDesc bd = pl.bias[t];
        pl.out[idx] = i < bd.rows ? raw[bd.off + bd.o0 + i] : 0.f;
    } else if (idx < nw + pl.n_bias * 16 + 16) {
        const int k = idx - nw - pl.n_bias * 16;
        pl.out[idx] = k < pl.n_scal ? raw[pl.scal[k]] : 0.f;
    }
}
__global__ void k_pack_all(const float* __restrict__ raw, const PackPlan* __restrict__ plans, int n_plans) {
    int s = 0;
    while (s + 1 < n_plans && (int)blockIdx.x >= plans[s + 1].block0) ++s;
    const PackPlan pl = plans[s];
    pack_one(raw, pl, ((int)blockIdx.x - pl.block0) * blockDim.x + threadIdx.x);
}

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
#if GENIE_TUNING
__device__ int g_abl_mfma;  // set from the host in tuning builds
#define MFMA16(a, b, c) (g_abl_mfma ? ((c) + (a) * (b)) : __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0))
#else
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#endif

// wave-level ordering point for data the lanes of one wave exchange through LDS (LDS operations of a wave complete in order)
#define GSYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); } while (0)
__device__ __forceinline__ float prelu1(float x, float a) { return x >= 0.f ? x : a * x; }
__device__ __forceinline__ f32x4 prelu4(f32x4 x, float a) {
    f32x4 y;
    y.x = prelu1(x.x, a); y.y = prelu1(x.y, a); y.z = prelu1(x.z, a); y.w = prelu1(x.w, a);
    return y;
}
// one 16-channel input block (k-steps r = 0..3) into one accumulator
__device__ __forceinline__ f32x4 mma_block(f32x4 acc, const f32x4 w, const f32x4 x) {
    acc = MFMA16(w.x, x.x, acc);
    acc = MFMA16(w.y, x.y, acc);
    acc = MFMA16(w.z, x.z, acc);
    acc = MFMA16(w.w, x.w, acc);
    return acc;
}

// Exact branch-free PReLU: max(x,0) + s*min(x,0) (one of the two terms is always zero)
__device__ __forceinline__ f32x4 prelu4u(f32x4 x, float s) {
    f32x4 y;
    y.x = fmaf(s, fminf(x.x, 0.f), fmaxf(x.x, 0.f));
    y.y = fmaf(s, fminf(x.y, 0.f), fmaxf(x.y, 0.f));
    y.z = fmaf(s, fminf(x.z, 0.f), fmaxf(x.z, 0.f));
    y.w = fmaf(s, fminf(x.w, 0.f), fmaxf(x.w, 0.f));
    return y;
}
// PReLU in two VALU ops when the side of the slope is known at compile time: max(x, s*x) for s <= 1, min(x, s*x)
// for s > 1 (exact: s*x is the exact PReLU value on the negative side and never wins on the positive side)
template <bool LE1>
__device__ __forceinline__ f32x4 prelu4s(f32x4 x, float s) {
    const f32x4 t = x * s;
    f32x4 y;
    if (LE1) { y.x = fmaxf(x.x, t.x); y.y = fmaxf(x.y, t.y); y.z = fmaxf(x.z, t.z); y.w = fmaxf(x.w, t.w); }
    else     { y.x = fminf(x.x, t.x); y.y = fminf(x.y, t.y); y.z = fminf(x.z, t.z); y.w = fminf(x.w, t.w); }
    return y;
}
// mean = sum * (1 / degree) added to the node-local term as ONE fused multiply-add per channel, spelled out so that every
// stage-2 kernel rounds the same way whatever the compiler would contract (the variants are bitwise equal by test)
__device__ __forceinline__ f32x4 fma4(f32x4 s, float inv, f32x4 c) {
    return f32x4{fmaf(s.x, inv, c.x), fmaf(s.y, inv, c.y), fmaf(s.z, inv, c.z), fmaf(s.w, inv, c.w)};
}
// PReLU_b(PReLU_a(z)) is again one PReLU: slope a*b on the negative side when a >= 0; when a < 0 the inner
// PReLU maps z < 0 to a*z > 0, which the outer one passes through: slope a.
__device__ __forceinline__ float compose_slopes(float a, float b) { return a >= 0.f ? a * b : a; }

// blocks of the training forward's saved pre-activations (16 channels each): h0, h1 = [t1 | t2], u, v, x_latent, Bipartite message
constexpr int SV_Z0 = 0, SV_T = 2, SV_UP = 6, SV_VP = 8, SV_O = 10, SV_ZB = 12, SV_BLOCKS = 14;
constexpr int ROWC = 32;   // row pitch (floats) of c = [c1 0..14,0 | c2 0..14,0]: the node-local layer-2 terms, 128 B
constexpr int ROWW = 16;   // row pitch of the projected gather operands wu / wv: 15 channels + 1 zero = 64 B
constexpr int WAVES = 4;   // waves per workgroup

struct DaArgs {
    int S, G, T;               // stations, source nodes this launch processes, tiles per source node = ceil(S/16)
    int gi0;                   // ... = positions [gi0, gi0 + G) of the processing order (sub-range launches of the sharded path; else 0)
    int seg;                   // source nodes per scheduling segment
    int abl;                   // GENIE_TUNING only: ablation bits
    int nxcd;                  // XCD-chunked sweep (8) or flat (1)
    const int32_t* sta_rowptr; const int32_t* sta_col;
    const int32_t* src_rowptr; const int32_t* src_col;
    const int32_t* order;
    const float* slice; const float* mask; const float* edge_attr;
    float* c; float* wu; float* wv;
    float* part;               // [G*T, 32] bipartite partial sums
    float* x_latent;           // optional [P,30]
    float* dbg_h0; float* dbg_h1;  // optional parity outputs [P,30] / [P,60]
    const int32_t* sta_user;       // station processing order (genie_set_station_order): internal station -> caller's station, or null
    const float* ea_int;           // with sta_user: edge_attr in processing order (genie_set_static_edge_attr), or null
    const float* mm_int;           // with sta_user: max_k Mask[p][k] in processing order, written by the split pass of this window
    const float* packed;       // packed A fragments for the stage
    const void* xs;            // k_stage1_h2: the two 16-B fp16 pieces of every [Slice || Mask] row, planar
    long long xs_plane;        // ... bytes per plane (= rows x 16)
    long long Pn;              // k_stage?_pcsr: number of product nodes (rowptr / col arrays are product-level there)
    const float* abs_sta;      // use_absolute_pos: [S][4] = {loc / (3 scale_rel), 0}, or null
    const float* abs_src;      // ... [G_ext][4] = {x_grid / (3 scale_rel), 0}
    const unsigned* abs_ts;    // k_stage1_h2<.., ABS>: [S][2] x 8 B = the fp16 pieces of a station's scaled position {x, y, z, 0} (processing order)
    const unsigned* abs_tg;    // ... [G_ext][2] x 8 B, source nodes
    const float* eb_sta;       // DataAggregationEdges: [S][48] per-station terms {layer 1 (30), 0, 0, layer 2 (15), 0}, or null
    const float* eb_src;       // ... [G][48] per-source-node terms
    const int32_t* src_tab;    // k_stage1_h2: [G][16] = {order[gi], its 15 source neighbours}, indexed by processing position gi
    float* save;               // training forward (generic kernels): pre-activations kept for the backward passes, 16-float blocks
                               // [SV_*][P][16] (genie_da_train_fwd), or null
    const float* slope2;       // stage 2: PReLU slope to use instead of the image's (association heads), or null
    int no_bip;                // stage 2: stop after x_latent (no Bipartite message / station sum): the association heads' last pass
    int rev;                   // k_stage2_fast: sweep every XCD's chunk backwards (the rows stage 1 wrote last are read first)
    int wgmap;                 // k_stage2_ord: blocks of 4 source nodes per workgroup, one node per wave (see the kernel)
};

// wave-uniform work item iterator. XCD x (blockIdx % 8, observed dispatch placement: used for speed only) sweeps
// its contiguous chunk of the processing order. Inside the chunk items are ordered in SEGMENTS of `seg` source
// nodes, station-tile major inside a segment: (tile 0 of seg nodes), (tile 1 of seg nodes), ...
struct ItemIter {
    int gbeg, gend, T, seg, xcd_, chunk_, lead_;    // chunk_: id of the (XCD, group) chunk; lead_: first workgroup of the chunk
    unsigned per_seg, m_per_seg, n_full, m_full, n_last, m_last, last_seg;   // divisors and their 2^32 reciprocals
    long long it, stride, nitems;
    // floor(x / d) for x < 2^31 with m = floor(2^32 / d): scalar multiply-high + at most two corrections (a hardware
    // integer division is ~30 dependent VALU ops + readfirstlanes per call, on every tile's critical path)
    static __device__ __forceinline__ unsigned fdiv(unsigned x, unsigned d, unsigned m, unsigned& r) {
        unsigned q = __umulhi(x, m);
        r = x - q * d;
        if (r >= d) { r -= d; ++q; }
        if (r >= d) { r -= d; ++q; }
        return q;
    }
    static __device__ __forceinline__ unsigned recip(unsigned d) { return d <= 1u ? 0xffffffffu : (unsigned)(0x100000000ull / d); }
    __device__ ItemIter(int G, int T_, int seg_, int nxcd, int wave, int gi0 = 0) {
        const int nx = (nxcd > 1 && gridDim.x >= nxcd && (gridDim.x % nxcd) == 0) ? nxcd : 1;
        const int xcd = blockIdx.x % nx;
        int lb = blockIdx.x / nx, nbx = gridDim.x / nx;
        xcd_ = xcd;
        gbeg = gi0 + (int)((long long)G * xcd / nx);
        gend = gi0 + (int)((long long)G * (xcd + 1) / nx);
        chunk_ = xcd; lead_ = lb == 0;
        T = T_;
        seg = seg_;
        nitems = (long long)(gend - gbeg) * T;
        const int wpb = blockDim.x >> 6;           // waves per workgroup
        it = (long long)lb * wpb + wave;
        stride = (long long)nbx * wpb;
        per_seg = (unsigned)seg * (unsigned)T;
        m_per_seg = recip(per_seg);
        n_full = (unsigned)seg;
        m_full = recip(n_full);
        last_seg = (unsigned)((gend - gbeg) / seg);            // index of the (possibly short or empty) last segment
        n_last = (unsigned)((gend - gbeg) - (int)last_seg * seg);
        if (n_last == 0) n_last = 1;
        m_last = recip(n_last);
    }
    // item -> (index into the processing order, station tile); 32-bit arithmetic (a chunk has < 2^31 items)
    __device__ void decode(long long item, int& gi, int& tb) const {
        unsigned rem, r2;
        const unsigned sidx = per_seg <= 1u ? (rem = 0u, (unsigned)item) : fdiv((unsigned)item, per_seg, m_per_seg, rem);
        const bool last = sidx >= last_seg;
        const unsigned n = last ? n_last : n_full;   // nodes in this (possibly last, short) segment
        const unsigned q = n <= 1u ? (r2 = 0u, rem) : fdiv(rem, n, last ? m_last : m_full, r2);
        tb = (int)q;
        gi = gbeg + (int)(sidx * (unsigned)seg) + (int)r2;
    }
};

// Neighbour sum of PReLU_s(init_trns [Slice || Mask]) with the 30-channel hidden state RECOMPUTED from the raw
// 8 input floats of every neighbour (2 k-steps x 2 out tiles = 4 MFMAs) instead of gathered from memory: a
// gathered row is 32 B instead of 128 B, so the whole neighbourhood working set stays in L2, and h0 is never
// stored. One call handles CH consecutive edges starting at e0 (2*CH dword loads per lane in flight); PRED adds
// the per-lane bound check used only for ragged (non-uniform degree) graphs. row(c) = row0 + c*stride.
// use_absolute_pos: the neighbour's hidden state also sees its station / source position. One of the two is the same as
// the centre node's (fixed value), the other is looked up by the neighbour id c in a [n][4] table.
struct AbsNbr {
    const float* tab;      // table indexed by the neighbour id, or null when absolute positions are off
    bool tab_is_station;   // the table is the station table (station-graph neighbours) / the source table
    float fixed;           // the centre node's value of the other coordinate (for this lane's q)
};

template <int CH, bool UNI, bool LE1, bool PRED>
__device__ __forceinline__ void recompute_chunk(const float* __restrict__ slice, const float* __restrict__ mask,
                                                long long row0, long long stride, int q,
                                                const int32_t* __restrict__ col, int e0, int ee,
                                                f32x4 w0, f32x4 w1, f32x4 b0, f32x4 b1, float slope,
                                                f32x4& s0, f32x4& s1, const AbsNbr ab = AbsNbr{nullptr, false, 0.f}) {
    long long off[CH];
    bool ok[CH];
    int cv = 0;
    if (UNI) cv = col[e0 + min((int)(__lane_id() & 7), CH - 1)];  // one coalesced load, broadcast below
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        ok[k] = PRED ? (e0 + k < ee) : true;
        const int c = UNI ? __builtin_amdgcn_readlane(cv, k) : col[ok[k] ? e0 + k : max(ee - 1, 0)];
        off[k] = (row0 + (long long)c * stride) * 4 + q;
    }
    float xs[CH], xm[CH], xv[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        xs[k] = slice[off[k]];
        xm[k] = mask[off[k]];
        if (ab.tab != nullptr) {
            const int c = UNI ? __builtin_amdgcn_readlane(cv, k) : col[ok[k] ? e0 + k : max(ee - 1, 0)];
            xv[k] = ab.tab[c * 4 + q];
        }
    }
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        f32x4 h_a = MFMA16(w0.x, xs[k], b0);
        f32x4 h_b = MFMA16(w1.x, xs[k], b1);
        h_a = MFMA16(w0.y, xm[k], h_a);
        h_b = MFMA16(w1.y, xm[k], h_b);
        if (ab.tab != nullptr) {
            const float lv = ab.tab_is_station ? xv[k] : ab.fixed, gv = ab.tab_is_station ? ab.fixed : xv[k];
            h_a = MFMA16(w0.z, lv, h_a);
            h_b = MFMA16(w1.z, lv, h_b);
            h_a = MFMA16(w0.w, gv, h_a);
            h_b = MFMA16(w1.w, gv, h_b);
        }
        h_a = prelu4s<LE1>(h_a, slope);
        h_b = prelu4s<LE1>(h_b, slope);
        if (PRED) {
            const float m = ok[k] ? 1.f : 0.f;
            h_a *= m; h_b *= m;
        }
        s0 += h_a;
        s1 += h_b;
    }
}

// Driver: uniform-degree neighbourhoods (every kNN graph) run as full chunks 8,4,2,1 with no predication and no
// divergent branch; ragged ones fall back to predicated chunks of 8. Edge order (= summation order) is kept.
template <bool UNI, bool LE1>
__device__ __forceinline__ void gather_recompute(const float* __restrict__ slice, const float* __restrict__ mask,
                                                 long long row0, long long stride, int q,
                                                 const int32_t* __restrict__ col, int eb, int ee,
                                                 f32x4 w0, f32x4 w1, f32x4 b0, f32x4 b1, float slope,
                                                 f32x4& s0, f32x4& s1, const AbsNbr ab = AbsNbr{nullptr, false, 0.f}) {
    const int n = ee - eb;
    const int nu = __builtin_amdgcn_readfirstlane(n);
    if (UNI || __all(n == nu)) {
        int e = eb, r = nu;
        for (; r >= 8; r -= 8, e += 8)
            recompute_chunk<8, UNI, LE1, false>(slice, mask, row0, stride, q, col, e, ee, w0, w1, b0, b1, slope, s0, s1, ab);
        if (r & 4) { recompute_chunk<4, UNI, LE1, false>(slice, mask, row0, stride, q, col, e, ee, w0, w1, b0, b1, slope, s0, s1, ab); e += 4; }
        if (r & 2) { recompute_chunk<2, UNI, LE1, false>(slice, mask, row0, stride, q, col, e, ee, w0, w1, b0, b1, slope, s0, s1, ab); e += 2; }
        if (r & 1) { recompute_chunk<1, UNI, LE1, false>(slice, mask, row0, stride, q, col, e, ee, w0, w1, b0, b1, slope, s0, s1, ab); }
    } else {
        for (int e = eb; __any(e < ee); e += 8)
            recompute_chunk<8, false, LE1, true>(slice, mask, row0, stride, q, col, e, ee, w0, w1, b0, b1, slope, s0, s1, ab);
    }
}

// Same structure for the 16-float projected operands wu / wv (one 16-B load per lane per neighbour).
template <int CH, bool UNI, bool PRED>
__device__ __forceinline__ void sum16_chunk(const float* __restrict__ base, long long stride,
                                            const int32_t* __restrict__ col, int e0, int ee, f32x4& s0) {
    const f32x4* r[CH];
    bool ok[CH];
    int cv = 0;
    if (UNI) cv = col[e0 + min((int)(__lane_id() & 7), CH - 1)];
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        ok[k] = PRED ? (e0 + k < ee) : true;
        const int c = UNI ? __builtin_amdgcn_readlane(cv, k) : col[ok[k] ? e0 + k : max(ee - 1, 0)];
        r[k] = (const f32x4*)(base + (long long)c * stride);
    }
    f32x4 y[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) y[k] = r[k][0];
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        if (PRED) y[k] *= ok[k] ? 1.f : 0.f;
        s0 += y[k];
    }
}
template <bool UNI>
__device__ __forceinline__ void gather_sum16(const float* __restrict__ base, long long stride,
                                             const int32_t* __restrict__ col, int eb, int ee, f32x4& s0) {
    const int n = ee - eb;
    const int nu = __builtin_amdgcn_readfirstlane(n);
    if (UNI || __all(n == nu)) {
        int e = eb, r = nu;
        for (; r >= 8; r -= 8, e += 8) sum16_chunk<8, UNI, false>(base, stride, col, e, ee, s0);
        if (r & 4) { sum16_chunk<4, UNI, false>(base, stride, col, e, ee, s0); e += 4; }
        if (r & 2) { sum16_chunk<2, UNI, false>(base, stride, col, e, ee, s0); e += 2; }
        if (r & 1) { sum16_chunk<1, UNI, false>(base, stride, col, e, ee, s0); }
    } else {
        for (int e = eb; __any(e < ee); e += 8) sum16_chunk<8, false, true>(base, stride, col, e, ee, s0);
    }
}

// ------------------------------------------------------------------------------------------------
// stage 1: everything of DataAggregation that does not need the SECOND pair of neighbour means
//   h0 = PReLU(init_trns [X || M])                                        module.py:87-88 (own node + every neighbour)
//   h1 = PReLU1([l1_t1_2 [h0 || mean_sta PReLU11(h0) || M] || l1_t2_2 [h0 || mean_src PReLU12(h0) || M]])   :90-92
//   u = PReLU21(l2_t1_1 h1), v = PReLU22(l2_t2_1 h1)                      :94-95
//   wu = l2_t1_2[:, 60:90] u,  wv = l2_t2_2[:, 60:90] v                  (operands of the second pair of means)
//   c  = [l2_t1_2[:, 0:60] h1 + l2_t1_2[:, 90:94] M + b || l2_t2_2[...]]  (node-local part of :94-95)
// Reads 32 B per product node (+ its neighbours' 32-B rows from L2), writes 256 B; h0 / h1 / u / v stay in VGPRs.
// ------------------------------------------------------------------------------------------------
// dense tail of stage 1 for one tile: layer 1 from (x0,x1 = own h0; n1*, n2* = neighbour means), then u / v, the
// projected operands wu / wv and the node-local layer-2 terms c; stores c, wu, wv (and h0 / h1 for parity runs)
// N independent accumulators x one 16-channel input block, k-step OUTER and accumulator INNER: consecutive MFMAs never
// target the same accumulator, so the 40-cycle dependent-issue latency of v_mfma_f32_16x16x4_f32 (32-cycle issue) is
// always covered (hipcc otherwise keeps the 4 dependent k-steps of one accumulator back to back).
template <int N>
__device__ __forceinline__ void mma_blocks(f32x4 (&acc)[N], const f32x4 (&w)[N], const f32x4 x) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int k = 0; k < N; ++k) acc[k] = MFMA16(w[k][r], x[r], acc[k]);
    }
}

// dense tail of stage 1 for one tile: layer 1 from (x0,x1 = own h0; n1*, n2* = neighbour means), then u / v, the
// projected operands wu / wv and the node-local layer-2 terms c; stores c, wu, wv (and h0 / h1 for parity runs)
__device__ __forceinline__ void stage1_dense(const DaArgs& a, const f32x4* lw, const float* lbias, int lane, int q,
                                             bool valid, long long p, int g, int sc, float mq, f32x4 x0, f32x4 x1, f32x4 n1a,
                                             f32x4 n1b, f32x4 n2a, f32x4 n2b, float a1, float a21, float a22) {
    if (a.dbg_h0 != nullptr && valid) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            a.dbg_h0[p * 30 + 4 * q + r] = x0[r];
            if (16 + 4 * q + r < 30) a.dbg_h0[p * 30 + 16 + 4 * q + r] = x1[r];
        }
    }
    // layer 1: tr1 = l1_t1_2 [h0 || n1 || M], tr2 = l1_t2_2 [h0 || n2 || M]; acc[k]: k = (half, tile)
    f32x4 acc[4], w4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = *(const f32x4*)(lbias + (2 + k) * 16 + 4 * q);
    if (a.eb_sta != nullptr) {   // DataAggregationEdges: the mean edge feature of a node is static, its Linear a per-node bias
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            acc[t] += *(const f32x4*)(a.eb_sta + (long long)sc * 48 + 16 * t + 4 * q);
            acc[2 + t] += *(const f32x4*)(a.eb_src + (long long)g * 48 + 16 * t + 4 * q);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) w4[k] = lw[G1_L1(k >> 1, k & 1, 0) * 64 + lane];
    mma_blocks<4>(acc, w4, x0);
#pragma unroll
    for (int k = 0; k < 4; ++k) w4[k] = lw[G1_L1(k >> 1, k & 1, 1) * 64 + lane];
    mma_blocks<4>(acc, w4, x1);
    {   // the neighbour-mean blocks feed only their own half: two accumulators per operand, interleave the two operands
        f32x4 wa[2], wb[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const f32x4 na = b == 0 ? n1a : n1b, nb = b == 0 ? n2a : n2b;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                wa[t] = lw[G1_L1(0, t, 2 + b) * 64 + lane];
                wb[t] = lw[G1_L1(1, t, 2 + b) * 64 + lane];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[0] = MFMA16(wa[0][r], na[r], acc[0]);
                acc[2] = MFMA16(wb[0][r], nb[r], acc[2]);
                acc[1] = MFMA16(wa[1][r], na[r], acc[1]);
                acc[3] = MFMA16(wb[1][r], nb[r], acc[3]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = MFMA16(lw[G1_L1(k >> 1, k & 1, 4) * 64 + lane].x, mq, acc[k]);
    if (a.save != nullptr && valid) {
#pragma unroll
        for (int k = 0; k < 4; ++k) *(f32x4*)(a.save + ((size_t)(SV_T + k) * a.Pn + p) * 16 + 4 * q) = acc[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = prelu4u(acc[k], a1);                      // h1 block k = (half, tile)
    if (a.dbg_h1 != nullptr && valid) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (16 * (k & 1) + 4 * q + r < 30) a.dbg_h1[p * 60 + 30 * (k >> 1) + 16 * (k & 1) + 4 * q + r] = acc[k][r];
    }
    // u = PReLU21(l2_t1_1 h1), v = PReLU22(l2_t2_1 h1); c_w = node-local layer-2 terms: 6 independent accumulators
    f32x4 o6[6], w6[6];
#pragma unroll
    for (int k = 0; k < 4; ++k) o6[k] = *(const f32x4*)(lbias + (6 + k) * 16 + 4 * q);
    o6[4] = *(const f32x4*)(lbias + 10 * 16 + 4 * q);
    o6[5] = *(const f32x4*)(lbias + 11 * 16 + 4 * q);
    if (a.eb_sta != nullptr) {
        o6[4] += *(const f32x4*)(a.eb_sta + (long long)sc * 48 + 32 + 4 * q);
        o6[5] += *(const f32x4*)(a.eb_src + (long long)g * 48 + 32 + 4 * q);
    }
#pragma unroll
    for (int hb = 0; hb < 4; ++hb) {
#pragma unroll
        for (int k = 0; k < 4; ++k) w6[k] = lw[G1_UV(k >> 1, k & 1, hb) * 64 + lane];
        w6[4] = lw[G1_C(0, hb) * 64 + lane];
        w6[5] = lw[G1_C(1, hb) * 64 + lane];
        mma_blocks<6>(o6, w6, acc[hb]);
    }
    o6[4] = MFMA16(lw[G1_C(0, 4) * 64 + lane].x, mq, o6[4]);
    o6[5] = MFMA16(lw[G1_C(1, 4) * 64 + lane].x, mq, o6[5]);
    if (a.save != nullptr && valid) {
#pragma unroll
        for (int k = 0; k < 4; ++k) *(f32x4*)(a.save + ((size_t)(SV_UP + k) * a.Pn + p) * 16 + 4 * q) = o6[k];
    }
    o6[0] = prelu4u(o6[0], a21); o6[1] = prelu4u(o6[1], a21);
    o6[2] = prelu4u(o6[2], a22); o6[3] = prelu4u(o6[3], a22);
    // wu = l2_t1_2[:, 60:90] u, wv = l2_t2_2[:, 60:90] v: two accumulators, interleaved
    f32x4 wuv[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, w2[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        w2[0] = lw[G1_W(0, b) * 64 + lane];
        w2[1] = lw[G1_W(1, b) * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            wuv[0] = MFMA16(w2[0][r], o6[b][r], wuv[0]);
            wuv[1] = MFMA16(w2[1][r], o6[2 + b][r], wuv[1]);
        }
    }
    if (valid && !ABL(a, 3)) {
        *(f32x4*)(a.c + p * ROWC + 4 * q) = o6[4];
        *(f32x4*)(a.c + p * ROWC + 16 + 4 * q) = o6[5];
        *(f32x4*)(a.wu + p * ROWW + 4 * q) = wuv[0];
        *(f32x4*)(a.wv + p * ROWW + 4 * q) = wuv[1];
    }
}

// generic stage 1: any CSR graphs (ragged degrees, empty neighbourhoods)
__global__ __launch_bounds__(256) void k_stage1(DaArgs a) {
    constexpr int NF4 = (G1_GROUPS * 256 + G1_BIAS * 16 + 16) / 4;
    __shared__ f32x4 lw[NF4];
    for (int i = threadIdx.x; i < NF4; i += 256) lw[i] = ((const f32x4*)a.packed)[i];
    __syncthreads();
    const float* lbias = (const float*)(lw + G1_GROUPS * 64);
    const float* lscal = lbias + G1_BIAS * 16;
    const float a0 = lscal[0], a1 = lscal[3], a21 = lscal[4], a22 = lscal[5];
    const float s11 = compose_slopes(a0, lscal[1]), s12 = compose_slopes(a0, lscal[2]);
    int lane = threadIdx.x & 63;
    const int j = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int S = a.S;
    ItemIter w(a.G, a.T, a.seg, a.nxcd, wave, a.gi0);
    for (; w.it < w.nitems; w.it += w.stride) {
        int gi, tb;
        w.decode(w.it, gi, tb);
        const int g = __builtin_amdgcn_readfirstlane(a.order[gi]);
#if !GENIE_HOIST_WEIGHTS
        asm volatile("" : "+v"(lane));  // A fragments are re-read from LDS per tile, not held in VGPRs across tiles
#endif
        const int s = tb * 16 + j;
        const bool valid = s < S;
        const int sc = valid ? s : S - 1;
        const long long p = (long long)g * S + sc;
        const float xs = a.slice[p * 4 + q];
        const float mq = a.mask[p * 4 + q];
        const f32x4 wi0 = lw[G1_INIT(0) * 64 + lane], wi1 = lw[G1_INIT(1) * 64 + lane];
        const f32x4 bi0 = *(const f32x4*)(lbias + 0 * 16 + 4 * q), bi1 = *(const f32x4*)(lbias + 1 * 16 + 4 * q);
        // own hidden state
        f32x4 x0 = MFMA16(wi0.x, xs, bi0), x1 = MFMA16(wi1.x, xs, bi1);
        x0 = MFMA16(wi0.y, mq, x0);
        x1 = MFMA16(wi1.y, mq, x1);
        float lq = 0.f, gq = 0.f;                 // use_absolute_pos: this node's station / source position channel q
        if (a.abs_sta != nullptr) {
            lq = a.abs_sta[sc * 4 + q];
            gq = a.abs_src[g * 4 + q];
            x0 = MFMA16(wi0.z, lq, x0); x1 = MFMA16(wi1.z, lq, x1);
            x0 = MFMA16(wi0.w, gq, x0); x1 = MFMA16(wi1.w, gq, x1);
        }
        if (a.save != nullptr && valid) {
            *(f32x4*)(a.save + ((size_t)(SV_Z0 + 0) * a.Pn + p) * 16 + 4 * q) = x0;
            *(f32x4*)(a.save + ((size_t)(SV_Z0 + 1) * a.Pn + p) * 16 + 4 * q) = x1;
        }
        x0 = prelu4u(x0, a0);
        x1 = prelu4u(x1, a0);
        // station-neighbour mean of PReLU11(h0): rows of the same source node
        f32x4 n1a = {0.f, 0.f, 0.f, 0.f}, n1b = {0.f, 0.f, 0.f, 0.f};
        {
            const int eb = a.sta_rowptr[sc], ee = a.sta_rowptr[sc + 1];
            if (!ABL(a, 0)) {
                if (s11 <= 1.f)
                    gather_recompute<false, true>(a.slice, a.mask, (long long)g * S, 1, q, a.sta_col, eb, ee, wi0, wi1, bi0,
                                                  bi1, s11, n1a, n1b, AbsNbr{a.abs_sta, true, gq});
                else
                    gather_recompute<false, false>(a.slice, a.mask, (long long)g * S, 1, q, a.sta_col, eb, ee, wi0, wi1, bi0,
                                                   bi1, s11, n1a, n1b, AbsNbr{a.abs_sta, true, gq});
            }
            const float inv = 1.f / (float)max(ee - eb, 1);
            n1a *= inv; n1b *= inv;
        }
        // source-neighbour mean of PReLU12(h0): same station, neighbouring source nodes (wave-uniform list)
        f32x4 n2a = {0.f, 0.f, 0.f, 0.f}, n2b = {0.f, 0.f, 0.f, 0.f};
        {
            const int eb = __builtin_amdgcn_readfirstlane(a.src_rowptr[g]);
            const int ee = __builtin_amdgcn_readfirstlane(a.src_rowptr[g + 1]);
            if (!ABL(a, 1)) {
                if (s12 <= 1.f)
                    gather_recompute<true, true>(a.slice, a.mask, (long long)sc, (long long)S, q, a.src_col, eb, ee, wi0, wi1,
                                                 bi0, bi1, s12, n2a, n2b, AbsNbr{a.abs_src, false, lq});
                else
                    gather_recompute<true, false>(a.slice, a.mask, (long long)sc, (long long)S, q, a.src_col, eb, ee, wi0, wi1,
                                                  bi0, bi1, s12, n2a, n2b, AbsNbr{a.abs_src, false, lq});
            }
            const float inv = 1.f / (float)max(ee - eb, 1);
            n2a *= inv; n2b *= inv;
        }
        stage1_dense(a, lw, lbias, lane, q, valid, p, g, sc, mq, x0, x1, n1a, n1b, n2a, n2b, a1, a21, a22);
    }
}

// Stage 1 on an IRREGULAR product graph (`use_subgraph: True`, process_utils.py:744-849): the product nodes are an arbitrary
// list of (station, source) pairs and both edge sets are CSR lists over PRODUCT-node ids (a.sta_rowptr/col, a.src_rowptr/col
// are indexed by product node here). A tile is 16 consecutive product nodes; same arithmetic as k_stage1.
__global__ __launch_bounds__(256) void k_stage1_pcsr(DaArgs a) {
    constexpr int NF4 = (G1_GROUPS * 256 + G1_BIAS * 16 + 16) / 4;
    __shared__ f32x4 lw[NF4];
    for (int i = threadIdx.x; i < NF4; i += 256) lw[i] = ((const f32x4*)a.packed)[i];
    __syncthreads();
    const float* lbias = (const float*)(lw + G1_GROUPS * 64);
    const float* lscal = lbias + G1_BIAS * 16;
    const float a0 = lscal[0], a1 = lscal[3], a21 = lscal[4], a22 = lscal[5];
    const float s11 = compose_slopes(a0, lscal[1]), s12 = compose_slopes(a0, lscal[2]);
    int lane = threadIdx.x & 63;
    const int j = lane & 15, q = lane >> 4;
    const long long ntiles = (a.Pn + 15) / 16;
    for (long long tile = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); tile < ntiles;
         tile += (long long)gridDim.x * (blockDim.x >> 6)) {
#if !GENIE_HOIST_WEIGHTS
        asm volatile("" : "+v"(lane));
#endif
        const long long pr = tile * 16 + j;
        const bool valid = pr < a.Pn;
        const long long p = valid ? pr : a.Pn - 1;
        const float xs = a.slice[p * 4 + q];
        const float mq = a.mask[p * 4 + q];
        const f32x4 wi0 = lw[G1_INIT(0) * 64 + lane], wi1 = lw[G1_INIT(1) * 64 + lane];
        const f32x4 bi0 = *(const f32x4*)(lbias + 0 * 16 + 4 * q), bi1 = *(const f32x4*)(lbias + 1 * 16 + 4 * q);
        f32x4 x0 = MFMA16(wi0.x, xs, bi0), x1 = MFMA16(wi1.x, xs, bi1);
        x0 = MFMA16(wi0.y, mq, x0);
        x1 = MFMA16(wi1.y, mq, x1);
        x0 = prelu4u(x0, a0);
        x1 = prelu4u(x1, a0);
        f32x4 n1a = {0.f, 0.f, 0.f, 0.f}, n1b = n1a, n2a = n1a, n2b = n1a;
        {
            const int eb = a.sta_rowptr[p], ee = a.sta_rowptr[p + 1];
            if (s11 <= 1.f) gather_recompute<false, true>(a.slice, a.mask, 0, 1, q, a.sta_col, eb, ee, wi0, wi1, bi0, bi1, s11, n1a, n1b);
            else gather_recompute<false, false>(a.slice, a.mask, 0, 1, q, a.sta_col, eb, ee, wi0, wi1, bi0, bi1, s11, n1a, n1b);
            const float inv = 1.f / (float)max(ee - eb, 1);
            n1a *= inv; n1b *= inv;
        }
        {
            const int eb = a.src_rowptr[p], ee = a.src_rowptr[p + 1];
            if (s12 <= 1.f) gather_recompute<false, true>(a.slice, a.mask, 0, 1, q, a.src_col, eb, ee, wi0, wi1, bi0, bi1, s12, n2a, n2b);
            else gather_recompute<false, false>(a.slice, a.mask, 0, 1, q, a.src_col, eb, ee, wi0, wi1, bi0, bi1, s12, n2a, n2b);
            const float inv = 1.f / (float)max(ee - eb, 1);
            n2a *= inv; n2b *= inv;
        }
        stage1_dense(a, lw, lbias, lane, q, valid, p, 0, 0, mq, x0, x1, n1a, n1b, n2a, n2b, a1, a21, a22);
    }
}

// the KS station-neighbour ids of one station as wide loads: a dword load whose lanes hit 16 different 32-B segments costs the
// texture path about as much as two and a half full 1-KB row loads (tools/: skeleton ablations of k_stage2_fast), and a tile
// issued KS of them
template <int KS>
__device__ __forceinline__ void load_sta_ids(const int32_t* __restrict__ sta_col, int sc, int (&sta)[KS]) {
    static_assert(KS % 4 == 0, "station-neighbour rows are read as 16-byte chunks");
    typedef int i32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int c = 0; c < KS / 4; ++c) {
        const i32x4 v = *(const i32x4*)(sta_col + sc * KS + 4 * c);
        sta[4 * c] = v.x; sta[4 * c + 1] = v.y; sta[4 * c + 2] = v.z; sta[4 * c + 3] = v.w;
    }
}

// ------------------------------------------------------------------------------------------------
// Stage 1 on the 16-bit matrix pipe with fp32-class operands ("f16x2").
//
// Measured on MI355X (tools/mfma_peak*.hip, tools/valu_rate.hip, tools/mfma_overlap.hip): v_mfma_f32_16x16x4_f32 runs at the
// fp32 VECTOR rate and does not overlap with VALU work, so the fp32-MFMA kernel above is bound by the sum of both; a 16-bit
// 32x32x16 MFMA does 16x the FLOPs in the same 32 cycles (and hides ~10 of them behind vector work).
//
//  * v_mfma_f32_32x32x16_f16: D[ch, node] for 32 channels x 32 nodes. A wave owns TWO 16-station tiles (lanes
//    0-15/32-47 and 16-31/48-63). Lane (j = lane&31, h = lane>>5) holds D channels 8*(r>>2) + 4h + (r&3), r = 0..15,
//    of node j; K-step ks of the next layer consumes registers 8ks..8ks+7 of both lanes of a node (16 channels), so an
//    accumulator block becomes B operands without any cross-lane movement. All our channel groups are 30 wide: one
//    32-block each; the two padding slots of a block (channels 30, 31: lane h = 1, registers 14, 15) carry the Mask
//    inputs of `cat(h, n, Mask)`.
//  * raw inputs arrive as 32-B rows [x0 | x1] of 8 fp16 each (x = Slice || Mask), written by k_split_rows, stored PLANAR
//    (piece q of row p at q * rows * 16 + p * 16: a half-wave reads one piece of 32 consecutive rows as 512 contiguous bytes).
//    A neighbour's hidden state is 2 MFMAs (K = 16 = the two 8-wide pieces).
//  * mean_k PReLU_s(z_k) = sum_k (al z_k + be |z_k|): two fused multiply-adds per neighbour value into one accumulator.
// ------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int XROW = 32;                 // bytes per split input row: two 16-B pieces
constexpr int XPC = 16;                  // bytes per piece
constexpr int H2_THREADS = 512;

// ---- f16x2: x ~ x0 + x1 with x0 = rn16(x), x1 = rn16(x - x0): 11 + 1 + 11 significant bits, i.e. within one fp32 ulp of x
// (exact when the residual needs <= 11 bits) while x1 stays a normal fp16 number (|x| >= 2^-2), within 2^-25 absolute below
// that (fp16 subnormals: the MFMA keeps them, tools/h2_probe.hip). Overflow needs |x| > 65504.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define MFMA32H(a, b, c) \
    __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, (a)), __builtin_bit_cast(f16x8, (b)), (c), 0, 0, 0)
__device__ __forceinline__ unsigned cvt_pk_f16(float lo, float hi) {            // round to nearest even, both halves
    unsigned r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float sub_f16_lo(float x, unsigned p) {              // x - float(p.lo), one exact fp32 operation
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p), "v"(x));
    return r;
}
__device__ __forceinline__ float sub_f16_hi(float x, unsigned p) {
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p), "v"(x));
    return r;
}
constexpr unsigned H2_SIXTEENTH = 0x2c002c00u;        // (1/16, 1/16) as an fp16 pair
__device__ __forceinline__ unsigned pk_mul_f16(unsigned p, unsigned c) {
    unsigned r;
    asm("v_pk_mul_f16 %0, %1, %2" : "=v"(r) : "v"(p), "v"(c));
    return r;
}
// one fp16 piece (low 16 bits) of v; codes as in build_h2_table
__device__ __forceinline__ unsigned f16_piece(float v, int piece) {
    if (piece >= 2) { v *= 16.f; piece -= 2; }
    const unsigned p0 = cvt_pk_f16(v, 0.f);
    if (piece == 0) return p0 & 0xffffu;
    const float r = sub_f16_lo(v, p0);
    return cvt_pk_f16(r, 0.f) & 0xffffu;
}
__device__ __forceinline__ unsigned f16_piece_w(float v, int piece) {            // weight pieces: code 1 = rn16(16 (W - W0))
    if (piece != 1) return f16_piece(v, piece);
    const unsigned p0 = cvt_pk_f16(v, 0.f);
    return cvt_pk_f16(16.f * sub_f16_lo(v, p0), 0.f) & 0xffffu;
}
// the split rows of one [Slice || Mask] row: two fp16 planes
__device__ __forceinline__ void store_split_row(unsigned* __restrict__ out, long long rows, long long p, const float (&v)[8]) {
    u32x4 o0, o1;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        o0[d] = cvt_pk_f16(v[2 * d], v[2 * d + 1]);
        o1[d] = cvt_pk_f16(sub_f16_lo(v[2 * d], o0[d]), sub_f16_hi(v[2 * d + 1], o0[d]));
    }
    *(u32x4*)(out + p * 4) = o0;
    *(u32x4*)(out + (rows + p) * 4) = o1;
}

__global__ void k_pack_h2(const float* __restrict__ raw, const int32_t* __restrict__ tbl, float* __restrict__ out, int nfrag,
                          int ntail) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < nfrag * 64) {
        u32x4 o;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            unsigned u[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int32_t ent = tbl[idx * 8 + 2 * d + k];
                u[k] = ent < 0 ? 0u : f16_piece_w(raw[ent & 0x0fffffff], (ent >> 28) & 3);
            }
            o[d] = u[0] | (u[1] << 16);
        }
        ((u32x4*)out)[idx] = o;
    } else if (idx < nfrag * 64 + ntail) {     // fp32 tail: bias blocks and PReLU slopes
        const int k = idx - nfrag * 64;
        const int32_t ent = tbl[nfrag * 512 + k];
        out[nfrag * 256 + k] = ent < 0 ? 0.f : raw[ent];
    }
}

// [Slice || Mask] rows (8 fp32) -> 32-B rows of two fp16x8 pieces
// sta_user (internal station -> caller's station, or null): the rows of a source node are written in the station processing order
__global__ void k_split_rows(const float* __restrict__ slice, const float* __restrict__ mask, long long rows,
                             unsigned* __restrict__ out, const int32_t* __restrict__ sta_user, int S, float* __restrict__ mm) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= rows) return;
    long long pu = p;
    if (sta_user != nullptr) {
        const long long g = p / S;
        pu = g * S + sta_user[(int)(p - g * S)];
    }
    const f32x4 s = *(const f32x4*)(slice + pu * 4), m = *(const f32x4*)(mask + pu * 4);
    if (sta_user != nullptr) mm[p] = fmaxf(fmaxf(m.x, m.y), fmaxf(m.z, m.w));      // the message mask of stage 2 (module.py:226)
    const float v[8] = {s.x, s.y, s.z, s.w, m.x, m.y, m.z, m.w};
    store_split_row(out, rows, p, v);
}

// Same with a station processing order, one workgroup per source node: the node's S rows are read in the caller's order
// (coalesced), staged in LDS, and written in processing order (coalesced); S <= SPLIT_G_MAXS rows fit the 64-KB staging buffer.
constexpr int SPLIT_G_MAXS = 2048;
__global__ __launch_bounds__(256) void k_split_rows_g(const float* __restrict__ slice, const float* __restrict__ mask, int S,
                                                      unsigned* __restrict__ out, const int32_t* __restrict__ sta_user,
                                                      float* __restrict__ mm, long long rows) {
    extern __shared__ __attribute__((aligned(16))) float stg[];        // [S][8]: Slice row | Mask row
    const long long base = (long long)blockIdx.x * S;
    for (int r = threadIdx.x; r < S; r += blockDim.x) {
        *(f32x4*)(stg + r * 8) = *(const f32x4*)(slice + (base + r) * 4);
        *(f32x4*)(stg + r * 8 + 4) = *(const f32x4*)(mask + (base + r) * 4);
    }
    __syncthreads();
    for (int r = threadIdx.x; r < S; r += blockDim.x) {
        const int u = sta_user[r];
        const f32x4 s = *(const f32x4*)(stg + u * 8), m = *(const f32x4*)(stg + u * 8 + 4);
        mm[base + r] = fmaxf(fmaxf(m.x, m.y), fmaxf(m.z, m.w));
        const float v[8] = {s.x, s.y, s.z, s.w, m.x, m.y, m.z, m.w};
        store_split_row(out, rows, base + r, v);
    }
}

// exact PReLU in two VALU ops for any slope: max(x, s*x) when s <= 1, min(x, s*x) otherwise, as med3(x, s*x, +-inf)
__device__ __forceinline__ f32x16 prelu16(f32x16 x, float s, float sel) {
    f32x16 y;
#pragma unroll
    for (int r = 0; r < 16; ++r) y[r] = __builtin_amdgcn_fmed3f(x[r], x[r] * s, sel);
    return y;
}
__device__ __forceinline__ f32x16 bias16(const float* lbias, int blk, int h) {
    f32x16 y;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const f32x4 t = *(const f32x4*)(lbias + blk * 32 + 8 * b + 4 * h);
        y[4 * b] = t.x; y[4 * b + 1] = t.y; y[4 * b + 2] = t.z; y[4 * b + 3] = t.w;
    }
    return y;
}
// training forward: a 32-channel accumulator (register r = channel 8 (r >> 2) + 4 h + (r & 3)) as two 16-float blocks of the
// block-planar save buffer [blk][P][16] the backward passes read (channels 30, 31 are padding there: zero)
__device__ __forceinline__ void h2_save32(float* __restrict__ save, long long Pn, int blk0, long long p, int h, const f32x16& v) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        f32x4 o = {v[4 * m], v[4 * m + 1], v[4 * m + 2], v[4 * m + 3]};
        if (m == 3 && h == 1) { o.z = 0.f; o.w = 0.f; }
        *(f32x4*)(save + ((size_t)(blk0 + (m >> 1)) * Pn + p) * 16 + 8 * (m & 1) + 4 * h) = o;
    }
}
// ------------------------------------------------------------------------------------------------
// STAGE 1, f16x2 form. An activation x is split into x0 = rn16(x), x1 = rn16(x - x0) (one v_cvt_pk_f16_f32 per pair and
// piece, one v_fma_mix_f32 per value) and a weight into W0 = rn16(W), W1' = rn16(16 (W - W0)); a K-step is THREE products,
// smallest first: W0 x1 + W1' (x0 / 16) + W0 x0 (x0 / 16: one v_pk_mul_f16 per pair). The scaled pair keeps the weight's second
// piece a normal fp16 number; without it that piece falls into fp16's subnormal range (absolute floor 2^-25) and the hidden
// states lose ~2x in accuracy (oracle-level emulation of the arithmetic on the o1_20x500 fixture: x_latent rms error vs fp64
// 1.30e-7 unscaled, 0.81e-7 scaled; three exact bf16 pieces with six products, the round-1..3 form: 0.68e-7; the reference's
// own fp32: 1.13e-7). Dropped: W1 x1 (2^-24 of a product) and the last-bit rounding of x1: the result is fp32-CLASS, not
// bit-for-bit fp32. The input layer (K = 8: [x0 ; x1] fill one K = 16 step) is computed 16 x too large as a whole,
// [P|P][x0;x1] + [Q|Q][x0;x1] with P + Q = 16 W and C = 16 b: the neighbour sums absorb the factor in their constants, the
// node's own h0 in its PReLU. Per wave-tile (32 nodes): 120 MFMAs (48 neighbour recompute, 24 layer 1, 36 u / v / c,
// 12 wu / wv; the bf16x3 form needed 216) and ~1500 vector instructions (2050).
// ------------------------------------------------------------------------------------------------
// lane k of every row of 16 lanes, broadcast to the row (DPP row_newbcast, gfx90a+)
template <int K_>
__device__ __forceinline__ int row_bcast(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x150 + K_, 0xf, 0xf, false); }
__device__ __forceinline__ int row_bcast_dyn(int v, int k) {     // k is a compile-time constant after unrolling
    switch (k) {
        case 0: return row_bcast<0>(v); case 1: return row_bcast<1>(v); case 2: return row_bcast<2>(v); case 3: return row_bcast<3>(v);
        case 4: return row_bcast<4>(v); case 5: return row_bcast<5>(v); case 6: return row_bcast<6>(v); case 7: return row_bcast<7>(v);
        case 8: return row_bcast<8>(v); case 9: return row_bcast<9>(v); case 10: return row_bcast<10>(v); case 11: return row_bcast<11>(v);
        case 12: return row_bcast<12>(v); case 13: return row_bcast<13>(v); case 14: return row_bcast<14>(v); default: return row_bcast<15>(v);
    }
}
template <int KS_>
__device__ __forceinline__ void split8h(const f32x16& v, u32x4 (&p)[3], unsigned sixteenth) {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const float a = v[8 * KS_ + 2 * d], b = v[8 * KS_ + 2 * d + 1];
        const unsigned p0 = cvt_pk_f16(a, b);
        p[0][d] = p0;
        p[1][d] = cvt_pk_f16(sub_f16_lo(a, p0), sub_f16_hi(b, p0));
        p[2][d] = pk_mul_f16(p0, sixteenth);
    }
}
// the three partial products of one K-step for N independent accumulators sharing the B pieces {x0, x1, x0 / 16}
template <int N>
__device__ __forceinline__ void mma3(f32x16 (&acc)[N], const f32x4* lw, const int (&f0)[N], int lane, const u32x4 (&b)[3]) {
    f32x4 w[N][2];
#pragma unroll
    for (int k = 0; k < N; ++k)
#pragma unroll
        for (int p = 0; p < 2; ++p) w[k][p] = lw[(f0[k] + p) * 64 + lane];
    constexpr int WP[3] = {0, 1, 0}, BP[3] = {1, 2, 0};
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int k = 0; k < N; ++k) acc[k] = MFMA32H(w[k][WP[t]], b[BP[t]], acc[k]);
}

template <int KS, int KP, bool EDGES, bool BIG, bool ABS = false, bool PCSR = false>
__global__ __launch_bounds__(H2_THREADS) void k_stage1_h2(DaArgs a) {
    // PCSR: irregular product graph (use_subgraph). A wave item is 32 consecutive product nodes; the neighbours of a node are
    // product-node ids from the product-level CSRs (at most KS / KP of them: a missing one is the node itself with weight 0,
    // the mean of an empty neighbourhood is 0); everything after the neighbour phase is the same code.
    static_assert(!(PCSR && (EDGES || ABS)), "irregular product graphs: default model definition only");
    typedef typename std::conditional<BIG, unsigned long long, unsigned>::type off_t_;
    constexpr int NF4 = H2_IMG_FLOATS / 4;
    __shared__ f32x4 lw[NF4];
    for (int i = threadIdx.x; i < NF4; i += H2_THREADS) lw[i] = ((const f32x4*)a.packed)[i];
    __syncthreads();
    const float* lbias = (const float*)(lw + H2_FRAGS * 64);
    const float* lscal = lbias + H2_NBIAS * 32;
    const float a0 = lscal[0], a1 = lscal[3], a21 = lscal[4], a22 = lscal[5];
    const float s11 = compose_slopes(a0, lscal[1]), s12 = compose_slopes(a0, lscal[2]);
    const float inf = __builtin_inff();
    const float sel0 = a0 <= 1.f ? inf : -inf, sel1 = a1 <= 1.f ? inf : -inf;
    const float sel21 = a21 <= 1.f ? inf : -inf, sel22 = a22 <= 1.f ? inf : -inf;
    // mean_k PReLU_s(z_k) = al * sum z_k + be * sum |z_k|; the z_k arrive 16 x too large
    const float al1 = (1.f + s11) / (32.f * KS), be1 = (1.f - s11) / (32.f * KS);
    const float al2 = (1.f + s12) / (32.f * KP), be2 = (1.f - s12) / (32.f * KP);
    const unsigned c16 = __builtin_amdgcn_readfirstlane(H2_SIXTEENTH);

    int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, half = (lane >> 4) & 1, jj = lane & 15;
    const bool hi = h != 0;
    const int S = a.S;
    ItemIter w(a.G, a.T, a.seg, a.nxcd, wave, a.gi0);
    const char* xs = (const char*)a.xs;
    const off_t_ la = hi ? (off_t_)a.xs_plane : (off_t_)0;          // lane h = 0 loads x0, lane h = 1 loads x1: [x0 ; x1] is one K = 16 step
    const off_t_ gstride = (off_t_)((unsigned)S * (unsigned)XPC);

    const f32x4 fa0 = lw[(H2_FA + 0) * 64 + lane], fa1 = lw[(H2_FA + 1) * 64 + lane];
    // use_absolute_pos: the six position columns of init_trns are one more K = 16 step per unit, B = {station piece, source piece}
    f32x4 fp0, fp1;
    if (ABS) { fp0 = lw[(H2_FABS + 0) * 64 + lane]; fp1 = lw[(H2_FABS + 1) * 64 + lane]; }
    const unsigned tp_h = (unsigned)h * 8u;           // piece tables: [node][piece] x 8 B
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    f32x16 biasA = bias16(lbias, 0, h);
#pragma unroll
    for (int r = 0; r < 16; ++r) biasA[r] *= 16.f;

    int jt = jj;
    auto fetch_ids = [&](long long pit_, int& idv_, int& sc_, bool& valid_, int (&sta_)[KS]) {
        int gi0, tb0, gi1, tb1;
        w.decode(2 * pit_, gi0, tb0);
        const bool second = 2 * pit_ + 1 < w.nitems;
        w.decode(second ? 2 * pit_ + 1 : 2 * pit_, gi1, tb1);
        idv_ = a.src_tab[(half ? gi1 : gi0) * 16 + jt];
        const int s = (half ? tb1 : tb0) * 16 + jt;
        valid_ = s < S && (second || !half);
        sc_ = s < S ? s : S - 1;
        load_sta_ids<KS>(a.sta_col, sc_, sta_);
    };
    constexpr int KPP = PCSR ? KP : 1;
    auto fetch_pcsr = [&](long long pit_, long long& p_, bool& valid_, int (&sta_)[KS], int (&src_)[KPP], int& ds_, int& dp_) {
        const long long pr = pit_ * 32 + (lane & 31);
        valid_ = pr < a.Pn;
        p_ = valid_ ? pr : a.Pn - 1;
        const int eb = a.sta_rowptr[p_], fb = a.src_rowptr[p_];
        ds_ = a.sta_rowptr[p_ + 1] - eb;
        dp_ = a.src_rowptr[p_ + 1] - fb;
#pragma unroll
        for (int q = 0; q < KS; ++q) {
            const int v = a.sta_col[max(eb + min(q, ds_ - 1), 0)];
            sta_[q] = q < ds_ ? v : (int)p_;
        }
#pragma unroll
        for (int q = 0; q < KPP; ++q) {
            const int v = a.src_col[max(fb + min(q, dp_ - 1), 0)];
            src_[q] = q < dp_ ? v : (int)p_;
        }
    };
    int idv = 0, sc = 0, sta_id[KS], src_id[KPP], dgs = 0, dgp = 0;
    long long pcur = 0;
    bool valid = false;
    // wave items: Cartesian = pairs of (source node, station tile) items of the XCD-aware sweep; PCSR = 32 consecutive product nodes
    const long long pit0 = PCSR ? (long long)blockIdx.x * (H2_THREADS / 64) + wave : w.it;
    const long long pstride = PCSR ? (long long)gridDim.x * (H2_THREADS / 64) : w.stride;
    const long long pend = PCSR ? (a.Pn + 31) / 32 : (w.nitems + 1) / 2;
    if (pit0 < pend) {
        if constexpr (PCSR) fetch_pcsr(pit0, pcur, valid, sta_id, src_id, dgs, dgp);
        else fetch_ids(pit0, idv, sc, valid, sta_id);
    }
    for (long long pit = pit0, pnext = 0; pit < pend; pit = pnext) {
        asm volatile("" : "+v"(lane));    // keeps the LDS fragment reads inside the loop (LICM would park them all in VGPRs)
        // idv: every row of 16 lanes holds {source node, its KP neighbours} of its own tile: one DPP row broadcast per id
        const int g = PCSR ? 0 : row_bcast<0>(idv);
        const long long p = PCSR ? pcur : (long long)g * S + sc;
        const off_t_ gbase0 = PCSR ? (off_t_)(unsigned long long)p * (off_t_)XPC : (off_t_)(unsigned)g * gstride;
        const unsigned sbase0 = PCSR ? 0u : (unsigned)sc * (unsigned)XPC;
        off_t_ gbase = gbase0 + la;                  // + this lane's plane: one multiply-add per neighbour row address
        off_t_ sbase = (off_t_)sbase0 + la;
        const int srcv = idv;
        // PCSR: per-lane weights of a present neighbour (16 x scaling and 1 / degree folded in)
        float alS = 0.f, beS = 0.f, alP = 0.f, beP = 0.f;
        if constexpr (PCSR) {
            const float is = 1.f / (float)max(dgs, 1), ip = 1.f / (float)max(dgp, 1);
            alS = (1.f + s11) * 0.03125f * is; beS = (1.f - s11) * 0.03125f * is;
            alP = (1.f + s12) * 0.03125f * ip; beP = (1.f - s12) * 0.03125f * ip;
        }

        // unit u: 0 = the node itself, 1..KS = station neighbours, KS+1..KS+KP = source neighbours
        constexpr int NU = 1 + KS + KP;
        static_assert(NU % 2 == 0, "units are processed in pairs");
        constexpr int DEPTH = ABS ? 4 : GENIE_H2_DEPTH;
        u32x4 buf[NU];
        u32x2 tp[NU], tso, tgo;          // ABS: the unit's own position piece; this tile's station / source piece
        if (ABS) {
            tso = *(const u32x2*)((const char*)a.abs_ts + (tp_h + (unsigned)sc * 16u));
            tgo = *(const u32x2*)((const char*)a.abs_tg + (tp_h + (unsigned)g * 16u));
        }
        auto issue = [&](int u) {
            off_t_ off;
            if (u == 0) off = gbase + sbase0;
            else if (PCSR) off = (off_t_)(unsigned)(u <= KS ? sta_id[u - 1] : src_id[(u - KS - 1) % KPP]) * (off_t_)XPC + la;
            else if (u <= KS) off = gbase + (unsigned)sta_id[u - 1] * (unsigned)XPC;
            else {
                const unsigned nb = (unsigned)row_bcast_dyn(srcv, u - KS);
                off = (BIG ? (off_t_)nb * gstride : (off_t_)__umul24(nb, (unsigned)gstride)) + sbase;
            }
            if (ABS && u > 0) {
                if (u <= KS) tp[u] = *(const u32x2*)((const char*)a.abs_ts + (tp_h + (unsigned)sta_id[u - 1] * 16u));
                else tp[u] = *(const u32x2*)((const char*)a.abs_tg + (tp_h + (unsigned)row_bcast_dyn(srcv, u - KS) * 16u));
            }
            if (ABL(a, 12) && u > 0) { buf[u] = buf[0]; return; }     // tuning: no neighbour-row loads
            buf[u] = *(const u32x4*)(xs + off);
        };
        const u32x4 own0 = *(const u32x4*)(xs + (gbase0 + sbase0));         // x0 of the own row (lanes h = 1: Mask pads)
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) issue(u);

        f32x16 sn, h0;          // sn: running mean_k PReLU_s(z_k) = sum_k (al z_k + be |z_k|), two fused multiply-adds per value
        u32x4 h0p[2][3], n1p[2][3], n2p[2][3];
        unsigned m01[3], m23[3];          // Mask pieces {x0, x1, x0 / 16} (lanes h = 1): fp16 pairs (M0,M1) and (M2,M3)
#pragma unroll
        for (int u = 0; u < NU; u += 2) {
#pragma unroll
            for (int d = 0; d < 2; ++d)
                if (u + DEPTH + d < NU) issue(u + DEPTH + d);
            asm volatile("" : "+v"(buf[u]), "+v"(buf[u + 1]));
            f32x16 z0, z1;
            if (ABS) {
                auto posb = [&](int uu) {
                    return uu == 0 ? u32x4{tso.x, tso.y, tgo.x, tgo.y}
                                   : uu <= KS ? u32x4{tp[uu].x, tp[uu].y, tgo.x, tgo.y} : u32x4{tso.x, tso.y, tp[uu].x, tp[uu].y};
                };
                const u32x4 p0 = posb(u), p1 = posb(u + 1);
                z0 = MFMA32H(fp1, p0, biasA); z1 = MFMA32H(fp1, p1, biasA);
                z0 = MFMA32H(fa1, buf[u], z0); z1 = MFMA32H(fa1, buf[u + 1], z1);
                z0 = MFMA32H(fp0, p0, z0); z1 = MFMA32H(fp0, p1, z1);
            } else {
                z0 = MFMA32H(fa1, buf[u], biasA); z1 = MFMA32H(fa1, buf[u + 1], biasA);
            }
            z0 = MFMA32H(fa0, buf[u], z0);
            z1 = MFMA32H(fa0, buf[u + 1], z1);
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                f32x16 z = d == 0 ? z0 : z1;
                const int uu = u + d;
                if (uu == 0) {
                    if (a.save != nullptr && valid) {
                        f32x16 zu;
#pragma unroll
                        for (int r = 0; r < 16; ++r) zu[r] = z[r] * 0.0625f;
                        h2_save32(a.save, a.Pn, SV_Z0, p, h, zu);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) h0[r] = __builtin_amdgcn_fmed3f(z[r] * 0.0625f, z[r] * (0.0625f * a0), sel0);
                    m01[0] = own0.z;   m23[0] = own0.w;
                    m01[1] = buf[0].z; m23[1] = buf[0].w;      // lane h = 1: buf = x1
                    m01[2] = pk_mul_f16(own0.z, c16); m23[2] = pk_mul_f16(own0.w, c16);
                } else {
                    float al = uu <= KS ? al1 : al2, be = uu <= KS ? be1 : be2;
                    if constexpr (PCSR) {
                        const bool present = uu <= KS ? uu - 1 < dgs : uu - KS - 1 < dgp;
                        al = present ? (uu <= KS ? alS : alP) : 0.f;
                        be = present ? (uu <= KS ? beS : beP) : 0.f;
                    }
                    if (uu == 1 || uu == KS + 1) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) sn[r] = fmaf(be, __builtin_fabsf(z[r]), al * z[r]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) sn[r] = fmaf(be, __builtin_fabsf(z[r]), fmaf(al, z[r], sn[r]));
                    }
                }
                if (uu == KS) { split8h<0>(sn, n1p[0], c16); split8h<1>(sn, n1p[1], c16); }
                if (uu == NU - 1) { split8h<0>(sn, n2p[0], c16); split8h<1>(sn, n2p[1], c16); }
            }
            asm volatile("" : "+v"(sn), "+v"(gbase), "+v"(sbase), "+v"(jt));
        }
        int idv_n = 0, sc_n = 0, sta_n[KS], src_n[KPP], dgs_n = 0, dgp_n = 0;
        long long p_n = 0;
        bool valid_n = false;
#pragma unroll
        for (int k = 0; k < KS; ++k) sta_n[k] = 0;
#pragma unroll
        for (int k = 0; k < KPP; ++k) src_n[k] = 0;
        pnext = pit + pstride;
        const bool has_next = pnext < pend;
        if (has_next) {
            if constexpr (PCSR) fetch_pcsr(pnext, p_n, valid_n, sta_n, src_n, dgs_n, dgp_n);
            else fetch_ids(pnext, idv_n, sc_n, valid_n, sta_n);
        }
        if (a.dbg_h0 != nullptr && valid) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = 8 * (r >> 2) + 4 * h + (r & 3);
                if (ch < 30) a.dbg_h0[p * 30 + ch] = h0[r];
            }
        }
        split8h<0>(h0, h0p[0], c16);
        split8h<1>(h0, h0p[1], c16);
#pragma unroll
        for (int q = 0; q < 3; ++q) {     // padding slots (channels 30, 31) carry the Mask: [h0 | M0 M1], [n | M2 M3]
            h0p[1][q].w = hi ? m01[q] : h0p[1][q].w;
            n1p[1][q].w = hi ? m23[q] : n1p[1][q].w;
            n2p[1][q].w = hi ? m23[q] : n2p[1][q].w;
        }
        // ---- layer 1: tr_t = l1_t{1,2}_2 [h0 || n_t || Mask], both halves at once
        f32x16 acc[2] = {bias16(lbias, 1, h), bias16(lbias, 2, h)};
        if (EDGES) {   // DataAggregationEdges: static per-station / per-source-node terms of layer 1
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const f32x4 es = *(const f32x4*)(a.eb_sta + (long long)sc * 48 + 8 * b + 4 * h);
                const f32x4 eg = *(const f32x4*)(a.eb_src + (long long)g * 48 + 8 * b + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) { acc[0][4 * b + e] += es[e]; acc[1][4 * b + e] += eg[e]; }
            }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int f0[2] = {H2_FL1 + (0 * 4 + ks) * 2, H2_FL1 + (1 * 4 + ks) * 2};
            mma3<2>(acc, lw, f0, lane, h0p[ks]);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {     // the neighbour-mean blocks differ per half: interleave by hand
            const int fa_ = H2_FL1 + (0 * 4 + 2 + ks) * 2, fb_ = H2_FL1 + (1 * 4 + 2 + ks) * 2;
            f32x4 wa[2], wb[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) { wa[q] = lw[(fa_ + q) * 64 + lane]; wb[q] = lw[(fb_ + q) * 64 + lane]; }
            constexpr int WP[3] = {0, 1, 0}, BP[3] = {1, 2, 0};
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                acc[0] = MFMA32H(wa[WP[t]], n1p[ks][BP[t]], acc[0]);
                acc[1] = MFMA32H(wb[WP[t]], n2p[ks][BP[t]], acc[1]);
            }
        }
        if (a.save != nullptr && valid) { h2_save32(a.save, a.Pn, SV_T, p, h, acc[0]); h2_save32(a.save, a.Pn, SV_T + 2, p, h, acc[1]); }
        acc[0] = prelu16(acc[0], a1, sel1);
        acc[1] = prelu16(acc[1], a1, sel1);
        asm volatile("" : "+v"(acc[0]), "+v"(acc[1]));
        if (a.dbg_h1 != nullptr && valid) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ch = 8 * (r >> 2) + 4 * h + (r & 3);
                    if (ch < 30) a.dbg_h1[p * 60 + 30 * t + ch] = acc[t][r];
                }
        }
        // ---- u, v and the node-local layer-2 terms c from h1 = [h1a (30) | M0 M1 | h1b (30) | M2 M3]
        f32x16 o3[3] = {bias16(lbias, 3, h), bias16(lbias, 4, h), bias16(lbias, 5, h)};
        if (EDGES) {   // ... and of the node-local layer-2 block c = [o1 (15), 0 | o2 (15), 0]
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const f32x4 es = *(const f32x4*)(a.eb_sta + (long long)sc * 48 + 32 + 8 * b + 4 * h);
                const f32x4 eg = *(const f32x4*)(a.eb_src + (long long)g * 48 + 32 + 8 * b + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) { o3[2][4 * b + e] += es[e]; o3[2][8 + 4 * b + e] += eg[e]; }
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            u32x4 hp[2][3];
            split8h<0>(acc[t], hp[0], c16);
            split8h<1>(acc[t], hp[1], c16);
#pragma unroll
            for (int q = 0; q < 3; ++q) hp[1][q].w = hi ? (t == 0 ? m01[q] : m23[q]) : hp[1][q].w;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const int ks = 2 * t + kb;
                const int f0[3] = {H2_FUVC + (0 * 4 + ks) * 2, H2_FUVC + (1 * 4 + ks) * 2, H2_FUVC + (2 * 4 + ks) * 2};
                mma3<3>(o3, lw, f0, lane, hp[kb]);
            }
        }
        if (valid) {
#pragma unroll
            for (int b = 0; b < 4; ++b)
                *(f32x4*)(a.c + p * ROWC + 8 * b + 4 * h) = f32x4{o3[2][4 * b], o3[2][4 * b + 1], o3[2][4 * b + 2], o3[2][4 * b + 3]};
        }
        if (a.save != nullptr && valid) { h2_save32(a.save, a.Pn, SV_UP, p, h, o3[0]); h2_save32(a.save, a.Pn, SV_VP, p, h, o3[1]); }
        o3[0] = prelu16(o3[0], a21, sel21);
        o3[1] = prelu16(o3[1], a22, sel22);
        asm volatile("" : "+v"(o3[0]), "+v"(o3[1]));
        // ---- projected gather operands [wu | wv] = [l2_t1_2[:, 60:90] u | l2_t2_2[:, 60:90] v]
        f32x16 ow[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { ow[0][r] = 0.f; ow[1][r] = 0.f; }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            u32x4 up[3], vp[3];
            if (kb == 0) { split8h<0>(o3[0], up, c16); split8h<0>(o3[1], vp, c16); }
            else { split8h<1>(o3[0], up, c16); split8h<1>(o3[1], vp, c16); }
            f32x4 wa[2], wb[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                wa[q] = lw[(H2_FW + (0 + kb) * 2 + q) * 64 + lane];
                wb[q] = lw[(H2_FW + (2 + kb) * 2 + q) * 64 + lane];
            }
            constexpr int WP[3] = {0, 1, 0}, BP[3] = {1, 2, 0};
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                ow[0] = MFMA32H(wa[WP[t]], up[BP[t]], ow[0]);
                ow[1] = MFMA32H(wb[WP[t]], vp[BP[t]], ow[1]);
            }
        }
        if (valid) {
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                *(f32x4*)(a.wu + p * ROWW + 8 * b + 4 * h) = f32x4{ow[0][4 * b], ow[0][4 * b + 1], ow[0][4 * b + 2], ow[0][4 * b + 3]};
                *(f32x4*)(a.wv + p * ROWW + 8 * b + 4 * h) =
                    f32x4{ow[1][8 + 4 * b], ow[1][8 + 4 * b + 1], ow[1][8 + 4 * b + 2], ow[1][8 + 4 * b + 3]};
            }
        }
        idv = idv_n; sc = sc_n; valid = valid_n; pcur = p_n; dgs = dgs_n; dgp = dgp_n;
#pragma unroll
        for (int k = 0; k < KS; ++k) sta_id[k] = sta_n[k];
#pragma unroll
        for (int k = 0; k < KPP; ++k) src_id[k] = src_n[k];
    }
}

// ------------------------------------------------------------------------------------------------
// stage 2: second pair of neighbour means (of the projected operands), PReLU2 -> x_latent; Bipartite fc1 + PReLU,
// mask gate, and the per-tile station sum.                      module.py:94-96, :229
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_stage2(DaArgs a) {
    constexpr int NF4 = (G2_GROUPS * 256 + G2_BIAS * 16 + 16) / 4;
    __shared__ f32x4 lw[NF4];
    for (int i = threadIdx.x; i < NF4; i += 256) lw[i] = ((const f32x4*)a.packed)[i];
    __syncthreads();
    const float* lbias = (const float*)(lw + G2_GROUPS * 64);
    const float* lscal = lbias + G2_BIAS * 16;
    const float a2 = a.slope2 != nullptr ? *a.slope2 : lscal[0], ab1 = lscal[1];
    int lane = threadIdx.x & 63;
    const int j = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int S = a.S;
    ItemIter w(a.G, a.T, a.seg, a.nxcd, wave, a.gi0);
    for (; w.it < w.nitems; w.it += w.stride) {
        int gi, tb;
        w.decode(w.it, gi, tb);
        const int g = __builtin_amdgcn_readfirstlane(a.order[gi]);
#if !GENIE_HOIST_WEIGHTS
        asm volatile("" : "+v"(lane));
#endif
        const int s = tb * 16 + j;
        const bool valid = s < S;
        const int sc = valid ? s : S - 1;
        const long long p = (long long)g * S + sc;
        f32x4 o[2];
        o[0] = *(const f32x4*)(a.c + p * ROWC + 4 * q);
        o[1] = *(const f32x4*)(a.c + p * ROWC + 16 + 4 * q);
        const float mq = a.mask[p * 4 + q];
        const float eq = q < 3 ? a.edge_attr[p * 3 + q] : 0.f;
        // neighbour means of the projected operands (16-float rows): they ARE the accumulator contributions
        f32x4 n1 = {0.f, 0.f, 0.f, 0.f}, n2 = {0.f, 0.f, 0.f, 0.f};
        {
            const int eb = a.sta_rowptr[sc], ee = a.sta_rowptr[sc + 1];
            const float* base = a.wu + (long long)g * S * ROWW + 4 * q;
            if (!ABL(a, 0)) gather_sum16<false>(base, ROWW, a.sta_col, eb, ee, n1);
            o[0] = fma4(n1, 1.f / (float)max(ee - eb, 1), o[0]);
        }
        {
            const int eb = __builtin_amdgcn_readfirstlane(a.src_rowptr[g]);
            const int ee = __builtin_amdgcn_readfirstlane(a.src_rowptr[g + 1]);
            const float* base = a.wv + (long long)sc * ROWW + 4 * q;
            if (!ABL(a, 1)) gather_sum16<true>(base, (long long)S * ROWW, a.src_col, eb, ee, n2);
            o[1] = fma4(n2, 1.f / (float)max(ee - eb, 1), o[1]);
        }
        if (a.save != nullptr && valid) {
            *(f32x4*)(a.save + ((size_t)(SV_O + 0) * a.Pn + p) * 16 + 4 * q) = o[0];
            *(f32x4*)(a.save + ((size_t)(SV_O + 1) * a.Pn + p) * 16 + 4 * q) = o[1];
        }
        o[0] = prelu4u(o[0], a2);   // x_latent[0:15]  (lane (j,q) holds channels 4q..4q+3, channel 15 is zero)
        o[1] = prelu4u(o[1], a2);   // x_latent[15:30]
        if (a.x_latent != nullptr && valid) {
            float* xl = a.x_latent + p * 30;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (4 * q + r < 15) {
                    xl[4 * q + r] = o[0][r];
                    xl[15 + 4 * q + r] = o[1][r];
                }
            }
        }
        if (a.no_bip) continue;
        // Bipartite message: m_p * PReLU_b1(fc1 [x_latent || edge_attr])
        f32x4 bp[2];
        bp[0] = *(const f32x4*)(lbias + 0 * 16 + 4 * q);
        bp[1] = *(const f32x4*)(lbias + 1 * 16 + 4 * q);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            bp[t] = mma_block(bp[t], lw[G2_BP(t, 0) * 64 + lane], o[0]);
            bp[t] = mma_block(bp[t], lw[G2_BP(t, 1) * 64 + lane], o[1]);
            bp[t] = MFMA16(lw[G2_BP(t, 2) * 64 + lane].x, eq, bp[t]);
            if (a.save != nullptr && valid) *(f32x4*)(a.save + ((size_t)(SV_ZB + t) * a.Pn + p) * 16 + 4 * q) = bp[t];
            bp[t] = prelu4u(bp[t], ab1);
        }
        float mm = fmaxf(mq, __shfl_xor(mq, 16));
        mm = fmaxf(mm, __shfl_xor(mm, 32));
        if (!valid) mm = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 v = bp[t] * mm;
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) {
                v.x += __shfl_xor(v.x, d);
                v.y += __shfl_xor(v.y, d);
                v.z += __shfl_xor(v.z, d);
                v.w += __shfl_xor(v.w, d);
            }
            if (j == 0) *(f32x4*)(a.part + ((long long)g * a.T + tb) * 32 + 16 * t + 4 * q) = v;
        }
    }
}

// Stage 2 on an irregular product graph (see k_stage1_pcsr). The Bipartite messages of a source node are not the rows of
// whole tiles here, so every node's gated message row is written in place of its c row and k_bip_out_seg sums the row range
// of each source node (product nodes are grouped by source node, process_utils.py:790-794) in row order.
__global__ __launch_bounds__(256) void k_stage2_pcsr(DaArgs a) {
    constexpr int NF4 = (G2_GROUPS * 256 + G2_BIAS * 16 + 16) / 4;
    __shared__ f32x4 lw[NF4];
    for (int i = threadIdx.x; i < NF4; i += 256) lw[i] = ((const f32x4*)a.packed)[i];
    __syncthreads();
    const float* lbias = (const float*)(lw + G2_GROUPS * 64);
    const float* lscal = lbias + G2_BIAS * 16;
    const float a2 = lscal[0], ab1 = lscal[1];
    int lane = threadIdx.x & 63;
    const int j = lane & 15, q = lane >> 4;
    const long long ntiles = (a.Pn + 15) / 16;
    for (long long tile = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); tile < ntiles;
         tile += (long long)gridDim.x * (blockDim.x >> 6)) {
#if !GENIE_HOIST_WEIGHTS
        asm volatile("" : "+v"(lane));
#endif
        const long long pr = tile * 16 + j;
        const bool valid = pr < a.Pn;
        const long long p = valid ? pr : a.Pn - 1;
        f32x4 o[2];
        o[0] = *(const f32x4*)(a.c + p * ROWC + 4 * q);
        o[1] = *(const f32x4*)(a.c + p * ROWC + 16 + 4 * q);
        const float mq = a.mask[p * 4 + q];
        const float eq = q < 3 ? a.edge_attr[p * 3 + q] : 0.f;
        f32x4 n1 = {0.f, 0.f, 0.f, 0.f}, n2 = {0.f, 0.f, 0.f, 0.f};
        {
            const int eb = a.sta_rowptr[p], ee = a.sta_rowptr[p + 1];
            gather_sum16<false>(a.wu + 4 * q, ROWW, a.sta_col, eb, ee, n1);
            o[0] = fma4(n1, 1.f / (float)max(ee - eb, 1), o[0]);
        }
        {
            const int eb = a.src_rowptr[p], ee = a.src_rowptr[p + 1];
            gather_sum16<false>(a.wv + 4 * q, ROWW, a.src_col, eb, ee, n2);
            o[1] = fma4(n2, 1.f / (float)max(ee - eb, 1), o[1]);
        }
        o[0] = prelu4u(o[0], a2);
        o[1] = prelu4u(o[1], a2);
        if (a.x_latent != nullptr && valid) {
            float* xl = a.x_latent + p * 30;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (4 * q + r < 15) {
                    xl[4 * q + r] = o[0][r];
                    xl[15 + 4 * q + r] = o[1][r];
                }
            }
        }
        f32x4 bp[2];
        bp[0] = *(const f32x4*)(lbias + 0 * 16 + 4 * q);
        bp[1] = *(const f32x4*)(lbias + 1 * 16 + 4 * q);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            bp[t] = mma_block(bp[t], lw[G2_BP(t, 0) * 64 + lane], o[0]);
            bp[t] = mma_block(bp[t], lw[G2_BP(t, 1) * 64 + lane], o[1]);
            bp[t] = MFMA16(lw[G2_BP(t, 2) * 64 + lane].x, eq, bp[t]);
            bp[t] = prelu4u(bp[t], ab1);
        }
        float mm = fmaxf(mq, __shfl_xor(mq, 16));
        mm = fmaxf(mm, __shfl_xor(mm, 32));
        if (valid) {      // the message row replaces the c row of the node (read above by these same lanes)
            *(f32x4*)(a.c + p * ROWC + 4 * q) = bp[0] * mm;
            *(f32x4*)(a.c + p * ROWC + 16 + 4 * q) = bp[1] * mm;
        }
    }
}

// k_stage2_fast for the production configuration (uniform-degree graphs, station processing order with a registered static
// edge_attr, static item stream, Bipartite half on), straight-line: the ISA of k_stage2_fast spends a fifth of its vector
// instructions on register copies at the joins of its option branches (the 15 source rows were copied out and back every tile),
// 34 ds_bpermute per tile on the station sum and a dozen uniform branches. Here
//  * the item after the last one is clamped to the last one, so every load of the software pipeline is unconditional and no
//    value has two definitions at a join;
//  * a tile's ids are its (wave-uniform) item number, one src_tab row and the 8 station-neighbour ids, which are loaded into the
//    registers the previous tile's ids have just left: nothing rotates but one register;
//  * the station sum over the 16 nodes of a tile is a DPP row reduction (row_shl:1, 2, 4, 8): lane 0 of every row adds the same
//    operands in the same tree as the xor butterfly of k_stage2 (bitwise identical), without the LDS round trips;
//  * the message mask is read by all four lanes of a node (one address) instead of max-reduced across them.
// Same arithmetic and summation order as k_stage2 / k_stage2_fast (bitwise identical results; tests).
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row_sum16_tree(float v) {      // lane 0 of every row of 16: the butterfly's sum tree
    v = dpp_add<0x101>(v);      // row_shl:1
    v = dpp_add<0x102>(v);
    v = dpp_add<0x104>(v);
    v = dpp_add<0x108>(v);
    return v;
}
// Row-layout loads: in the MFMA layout lane (j = lane & 15, q = lane >> 4) reads the 16-B chunk q of node j's row, so four
// CONSECUTIVE lanes touch four different rows and the texture path works on 16 useful bytes per 64-B request (measured: 24 B per
// clock and CU where contiguous row gathers reach 57). Every row is therefore loaded in the layout lane = 4 r + cq (node r = lane >> 2,
// chunk cq = lane & 3): four consecutive lanes read one 64-B row, sixteen consecutive source rows one contiguous KB. Everything up
// to x_latent is elementwise per (node, channel) and runs in that layout; x_latent, edge_attr and the gated mask then go through a
// 2.3-KB per-wave LDS scratch (rows of 36 floats) into the MFMA layout for fc1.
// XL: also store x_latent [P, 30] (caller's station order). NB: stop after x_latent (no Bipartite message / station sum): the
// last pass of the association heads (genie_assoc_fwd).
// Measured and dropped: other positions of the three load bursts (0.2627 / 0.2650 / 0.2649 ms), waves of a
// workgroup phased half an iteration apart by barriers (0.262 -> 0.290), streamed rows two tiles ahead (246 VGPRs, 0.226 -> 0.240),
// MFMA-layout loads (0.262 vs 0.226), station rows staged in LDS behind a barrier (0.282 -> 0.299 after a cold stage 1).
template <int KS, int KP, bool XL, bool NB = false>
__global__ __launch_bounds__(256, GENIE_S2_WAVES) void k_stage2_ord(DaArgs a) {
    constexpr int NF4 = (G2_GROUPS * 256 + G2_BIAS * 16 + 16) / 4;
    __shared__ f32x4 lw[NF4];
    __shared__ __attribute__((aligned(16))) float tsc[4 * 16 * 36];
    for (int i = threadIdx.x; i < NF4; i += blockDim.x) lw[i] = ((const f32x4*)a.packed)[i];
    __syncthreads();
    const float* lbias = (const float*)(lw + G2_GROUPS * 64);
    const float* lscal = lbias + G2_BIAS * 16;
    const float a2 = a.slope2 != nullptr ? *a.slope2 : lscal[0], ab1 = lscal[1];
    int lane = threadIdx.x & 63;
    const int j = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int jl = lane >> 2, ql = lane & 3;      // (node, chunk) this lane LOADS
    float* ts = tsc + wave * 16 * 36;
    const int S = a.S;
    ItemIter w(a.G, a.T, a.seg, a.nxcd, wave, a.gi0);
    // a.wgmap: a workgroup takes BLOCKS of 4 consecutive source nodes of its XCD's chunk, wave k sweeps the tiles of the k-th
    // node of the block: the four waves of a CU then read the source-neighbour rows of four adjacent source nodes (half of
    // them shared) for the same station tile at about the same time, and every station row wu[g] is gathered on one CU only
    if (a.wgmap) {
        const int nx = (a.nxcd > 1 && gridDim.x >= (unsigned)a.nxcd && (gridDim.x % a.nxcd) == 0) ? a.nxcd : 1;
        const int lb = blockIdx.x / nx, nbx = gridDim.x / nx, n = w.gend - w.gbeg, n_blk = (n + 3) / 4;
        int n_my = lb < n_blk ? (n_blk - lb + nbx - 1) / nbx : 0;
        if (n_my > 0 && 4 * (lb + (n_my - 1) * nbx) + wave >= n) --n_my;
        w.it = 0; w.stride = 1; w.nitems = (long long)n_my * a.T;
        w.lead_ = lb; w.chunk_ = nbx;                     // (reused as: first block, block stride)
    }
    if (w.it >= w.nitems) return;
    const char* wub = (const char*)a.wu;
    const char* wvb = (const char*)a.wv;
    const unsigned q16 = 16u * (unsigned)ql;
    const size_t gpitch = (size_t)S * 64u;                 // bytes of one source node's rows in wu / wv
    const unsigned m_T = ItemIter::recip((unsigned)a.T);

    // item -> wave-uniform (processing position gi, station tile tb); every XCD's chunk is swept BACKWARDS: the c / wu / wv rows
    // stage 1 wrote last (still in the Infinity Cache) are read first (0.278 -> 0.276 ms)
    auto item_of = [&](long long it, int& gi, int& tb) {
        const long long itr = w.nitems - 1 - it;
        if (a.wgmap) {
            unsigned rem;
            const unsigned kb = a.T <= 1 ? (rem = 0u, (unsigned)itr) : ItemIter::fdiv((unsigned)itr, (unsigned)a.T, m_T, rem);
            tb = (int)rem;
            gi = w.gbeg + 4 * (w.lead_ + (int)kb * w.chunk_) + wave;
        } else {
            w.decode(itr, gi, tb);
        }
        gi = __builtin_amdgcn_readfirstlane(gi);
        tb = __builtin_amdgcn_readfirstlane(tb);
    };
    struct Stream { f32x4 o[2]; float mq, eq; };                  // streamed rows of a tile: c, message mask, edge_attr
    struct Rows { f32x4 ru[KS], rv[KP]; } rows;                    // gathered rows
    Stream sA;
    int sta[KS];
    auto load_ids = [&](int gi, int tb, int& idv) {
        idv = a.src_tab[gi * 16 + j];
        const int s = tb * 16 + jl;
        load_sta_ids<KS>(a.sta_col, s < S ? s : S - 1, sta);
    };
    auto issue0 = [&](Stream& st, int idv, int tb) {
        const int g = __builtin_amdgcn_readlane(idv, 0);
        const int s = tb * 16 + jl, sc = s < S ? s : S - 1;
        long long p = (long long)g * S + sc;
        if (ABL(a, 9)) p &= 4095;          // tuning: streamed rows from a cache-resident region
        st.o[0] = *(const f32x4*)(a.c + p * ROWC + 4 * ql);
        st.o[1] = *(const f32x4*)(a.c + p * ROWC + 16 + 4 * ql);
        st.mq = NB ? 0.f : a.mm_int[p];
        st.eq = (!NB && ql < 3) ? a.ea_int[p * 3 + ql] : 0.f;
        const char* wug = wub + (ABL(a, 11) ? (size_t)0 : (size_t)g * gpitch);     // tuning bit 11: gathers hit one resident block
#pragma unroll
        for (int k = 0; k < KS; ++k) rows.ru[k] = ABL(a, 0) ? st.o[0] : *(const f32x4*)(wug + ((unsigned)sta[k] * 64u + q16));
    };
    auto issue_v = [&](int idv, int tb, int k0, int k1) {
        const int s = tb * 16 + jl, sc = s < S ? s : S - 1;
        const unsigned so = (unsigned)sc * 64u + q16;
#pragma unroll
        for (int k = 0; k < KP; ++k)
            if (k >= k0 && k < k1) {
                // the row base of a source neighbour is wave-uniform: kept opaque in an SGPR pair, so that the load is
                // `global_load v, voffset, s[base]` (left to itself hipcc hoists wvb + so into a VGPR pair and adds the
                // scalar part with a 64-bit vector multiply-add per neighbour: 3 vector instructions each)
                unsigned long long wvk = (unsigned long long)wvb + (ABL(a, 11) ? (size_t)k : (size_t)__builtin_amdgcn_readlane(idv, 1 + k)) * gpitch;
                asm volatile("" : "+s"(wvk));
                typedef const __attribute__((address_space(1))) char* gbytes;
                typedef const __attribute__((address_space(1))) f32x4* grow;
                rows.rv[k] = ABL(a, 1) ? rows.ru[0] : *(grow)((gbytes)wvk + so);
            }
    };
    constexpr int KH = (KP + 1) / 2;

    long long it = w.it;
    int gi_c, tb_c, gi_n, tb_n, idv_c, idv_n;
    item_of(it, gi_c, tb_c);
    load_ids(gi_c, tb_c, idv_c);
    issue0(sA, idv_c, tb_c);
    issue_v(idv_c, tb_c, 0, KP);
    {
        const long long itn = it + w.stride < w.nitems ? it + w.stride : it;
        item_of(itn, gi_n, tb_n);
        load_ids(gi_n, tb_n, idv_n);
    }
    for (;;) {
        asm volatile("" : "+v"(lane));
        const int g_c = __builtin_amdgcn_readlane(idv_c, 0);
        const bool has_next = it + w.stride < w.nitems;
        const long long it2 = it + 2 * w.stride < w.nitems ? it + 2 * w.stride : (has_next ? it + w.stride : it);
        int gi_2, tb_2, idv_2;
        item_of(it2, gi_2, tb_2);
        // (1) consume the rows of this tile: neighbour means of the projected operands in edge order, PReLU2 -> x_latent
        f32x4 n1 = {0.f, 0.f, 0.f, 0.f}, n2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < KS; ++k) n1 += rows.ru[k];
#pragma unroll
        for (int k = 0; k < KP; ++k) n2 += rows.rv[k];
        f32x4 o[2];
        o[0] = prelu4u(fma4(n1, 1.f / (float)KS, sA.o[0]), a2);
        o[1] = prelu4u(fma4(n2, 1.f / (float)KP, sA.o[1]), a2);
        float mq = sA.mq, eq = sA.eq;
        const int s_l = tb_c * 16 + jl;              // the node this lane loaded (not the node it holds in the MFMA layout)
        const bool valid_l = s_l < S;
        const f32x4 ol0 = o[0], ol1 = o[1];
        if (!NB) {      // row layout -> MFMA layout through the wave's LDS scratch: node r's row = [o1 (16) | o2 (16) | edge_attr (3) | gated mask]
            *(f32x4*)(ts + jl * 36 + 4 * ql) = o[0];
            *(f32x4*)(ts + jl * 36 + 16 + 4 * ql) = o[1];
            ts[jl * 36 + 32 + ql] = ql < 3 ? eq : (valid_l ? mq : 0.f);
            GSYNC();
            o[0] = *(const f32x4*)(ts + j * 36 + 4 * q);
            o[1] = *(const f32x4*)(ts + j * 36 + 16 + 4 * q);
            eq = q < 3 ? ts[j * 36 + 32 + q] : 0.f;
            mq = ts[j * 36 + 35];
        }
        asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(idv_n));
        // (2) first burst of the next tile's rows (the station-neighbour ids are dead after it)
        issue0(sA, idv_n, tb_n);
        if (NB) issue_v(idv_n, tb_n, 0, KP);
        if (XL && valid_l) {
            const int su = a.sta_user[s_l];
            float* xl = a.x_latent + ((long long)g_c * S + su) * 30;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * ql + r < 15) { xl[4 * ql + r] = ol0[r]; xl[15 + 4 * ql + r] = ol1[r]; }
        }
        f32x4 bp[2];
        bp[0] = *(const f32x4*)(lbias + 0 * 16 + 4 * q);
        bp[1] = *(const f32x4*)(lbias + 1 * 16 + 4 * q);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (NB) break;
            if (!ABL(a, 6)) {      // (tuning bit 6: no fc1 MFMAs)
                bp[t] = mma_block(bp[t], lw[G2_BP(t, 0) * 64 + lane], o[0]);
                bp[t] = mma_block(bp[t], lw[G2_BP(t, 1) * 64 + lane], o[1]);
                bp[t] = MFMA16(lw[G2_BP(t, 2) * 64 + lane].x, eq, bp[t]);
            } else bp[t] += o[0] + o[1] + eq;
            bp[t] = prelu4u(bp[t], ab1);
            // (3) second / third burst, behind the first / second output tile of fc1
            asm volatile("" : "+v"(bp[t]), "+v"(idv_n));
            if (t == 0) issue_v(idv_n, tb_n, 0, KH); else issue_v(idv_n, tb_n, KH, KP);
        }
        // (4) ids of the tile after next (the item after the last one repeats the last one: its loads are never consumed)
        load_ids(gi_2, tb_2, idv_2);
        // (5) mask gate and station sum of this tile
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (NB) break;
            f32x4 v = bp[t] * mq;
            v.x = row_sum16_tree(v.x); v.y = row_sum16_tree(v.y); v.z = row_sum16_tree(v.z); v.w = row_sum16_tree(v.w);
            if (j == 0) *(f32x4*)(a.part + ((long long)g_c * a.T + tb_c) * 32 + 16 * t + 4 * q) = v;
        }
        if (!has_next) break;
        it += w.stride;
        idv_c = idv_n; tb_c = tb_n;
        idv_n = idv_2; tb_n = tb_2;
    }
}

// ------------------------------------------------------------------------------------------------
// Association heads on the product graph: BipartiteGraphReadOutOperator (module.py:333-352) and
// DataAggregationAssociationPhase (:356-403) as three P-sized passes in the layout of the DataAggregation kernels
// (fp32 MFMA, tile = 16 stations of one source node, outputs of one Linear are the B operands of the next):
//   k_assoc_pre (G-sized): per-source-node terms pg[g] (the y_latent part of fc1, mask1 and the mask1 columns)
//   k_assoc_a: s = PReLU2(fc2(mask1 PReLU1(fc1[y_latent[g] || e_p])))                       :343-352
//              tr = PReLU(init_trns[s || x_latent || mask1 || Mask]), q1 = PReLU11(l1_t1_1 tr), q2 = PReLU12(l1_t2_1 tr)   :389-396
//   k_assoc_b: tr1 = PReLU1([l1_t1_2[tr || mean_sta q1 || mask] || l1_t2_2[tr || mean_src q2 || mask]])             :397-398
//              r1 = PReLU21(l2_t1_1 tr1), r2 = PReLU22(l2_t2_1 tr1); c / wu / wv exactly as stage 1 leaves them     :399-400
//   stage-2 kernel with `no_bip` (second pair of means + PReLU2 -> [P, 30])                                          :401
// Unlike DataAggregation the first-layer gather operand q is not a function of 8 raw floats, so q1 / q2 (32-float rows) are
// stored and gathered. tr / q1 / q2 and c / wu / wv live in (station) processing order like the stage-1 outputs.
// ------------------------------------------------------------------------------------------------
struct AsArgs {
    int S, G, T, seg, nxcd;
    const int32_t* order;
    const int32_t* sta_rowptr; const int32_t* sta_col; const int32_t* src_rowptr; const int32_t* src_col;
    const int32_t* sta_user;     // internal station -> caller's station (inputs are in the caller's order), or null
    const float* pg;             // [G][AS_PG]
    const float* ps;             // [S][AS_PS] static per-station terms of the two model variants (caller's station order), or null
    const float* x_latent; const float* mask; const float* edge_attr;   // caller's order: [P,30], [P,4], [P,3]
    float* tr; float* q1; float* q2;                                    // [P,32]
    float* c; float* wu; float* wv;
    const float* packed;
    float* save; long long Pn;   // training forward: pre-activations kept for the backward passes, [AV_*][Pn][16], or null
};
// blocks of the association phase's saved pre-activations: BipartiteGraphReadOutOperator fc1 (before PReLU and the mask gate) and
// fc2, init_trns, l1_t1_1 / l1_t2_1 (w, tile), [10, 11] = the output layer (written by the stage-2 kernel as its SV_O), layer 1
// (half, tile), l2_t1_1 / l2_t2_1 (w, tile)
constexpr int AV_Z1 = 0, AV_SV = 2, AV_TR = 3, AV_Q = 5, AV_O = 10, AV_T = 12, AV_UV = 16, AV_BLOCKS = 20;
static_assert(AV_O == SV_O, "the stage-2 kernel stores the output layer's pre-activations at SV_O");

struct AsPreOffs { int ro_fc1_w, ro_fc1_b, as_init_w, as_l1t12_w, as_l1t22_w, as_l2t12_w, as_l2t22_w;
                   int as_init_abs, as_l1t12_p, as_l1t22_p, as_l2t12_p, as_l2t22_p; };

// The two model variants in the association phase: under use_updated_model_definition the mean edge feature of a node's
// in-neighbourhood (static: mpos_sta [S][4] / mpos_src [G][4], genie_set_edge_features) enters l1_t?_2 / l2_t?_2 (module.py:462-467,
// :472-480); under use_absolute_pos the station / source positions / (3 scale_rel) (abs_sta [S][4] / abs_src [G][4]) are appended to
// the head's input (module.py:987-988, 6 more columns of init_trns). Both are per-station / per-source-node ADDITIVE terms of a
// pre-activation: the source-node ones are folded into pg, the station ones are ps [S][AS_PS]: [0:30] init_trns, [32:62]
// l1_t1_2, [64:79] l2_t1_2.
constexpr int AS_PS = 80;
__device__ __forceinline__ float dot4w(const float* __restrict__ w, const float* __restrict__ m, int n) {
    float v = 0.f;
    for (int c = 0; c < n; ++c) v = fmaf(w[c], m[c], v);
    return v;
}

__global__ __launch_bounds__(256) void k_assoc_pre(const float* __restrict__ raw, AsPreOffs o, const float* __restrict__ y_latent,
                                                   const float* __restrict__ mask_src, int G, const float* __restrict__ mpos_src,
                                                   const float* __restrict__ abs_src, float* __restrict__ pg) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= G * AS_PG) return;
    const int g = idx / AS_PG, k = idx - g * AS_PG;
    const float m = mask_src[g];
    float v = 0.f;
    if (k < 30) {
        v = raw[o.ro_fc1_b + k];
        const float* w = raw + o.ro_fc1_w + k * 33;
        for (int c = 0; c < 30; ++c) v = fmaf(w[c], y_latent[g * 30 + c], v);
    } else if (k == 31) v = m;
    else if (k >= 32 && k < 62) {
        v = m * raw[o.as_init_w + (k - 32) * 50 + 45];
        if (abs_src) v += dot4w(raw + o.as_init_abs + (k - 32) * 6 + 3, abs_src + g * 4, 3);
    } else if (k >= 64 && k < 94) v = m * raw[o.as_l1t12_w + (k - 64) * 65 + 60];
    else if (k >= 96 && k < 126) {
        v = m * raw[o.as_l1t22_w + (k - 96) * 65 + 60];
        if (mpos_src) v += dot4w(raw + o.as_l1t22_p + (k - 96) * 4, mpos_src + g * 4, 4);
    } else if (k >= 128 && k < 143) v = m * raw[o.as_l2t12_w + (k - 128) * 95 + 90];
    else if (k >= 144 && k < 159) {
        v = m * raw[o.as_l2t22_w + (k - 144) * 95 + 90];
        if (mpos_src) v += dot4w(raw + o.as_l2t22_p + (k - 144) * 4, mpos_src + g * 4, 4);
    }
    pg[idx] = v;
}

__global__ void k_assoc_ps(const float* __restrict__ raw, AsPreOffs o, int S, const float* __restrict__ mpos_sta,
                           const float* __restrict__ abs_sta, float* __restrict__ ps) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= S * AS_PS) return;
    const int s = idx / AS_PS, k = idx - s * AS_PS;
    float v = 0.f;
    if (k < 30) { if (abs_sta) v = dot4w(raw + o.as_init_abs + k * 6, abs_sta + s * 4, 3); }
    else if (k >= 32 && k < 62) { if (mpos_sta) v = dot4w(raw + o.as_l1t12_p + (k - 32) * 4, mpos_sta + s * 4, 4); }
    else if (k >= 64 && k < 79) { if (mpos_sta) v = dot4w(raw + o.as_l2t12_p + (k - 64) * 4, mpos_sta + s * 4, 4); }
    ps[idx] = v;
}

__device__ __forceinline__ f32x4 ld_row30(const float* row, int b, int q) {     // channels 16b + 4q .. +3 of a 30-float row (8-B aligned)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 lo = *(const f32x2*)(row + 16 * b + 4 * q);
    f32x2 hi = {0.f, 0.f};
    if (b == 0 || q < 3) hi = *(const f32x2*)(row + 16 * b + 4 * q + 2);
    return f32x4{lo.x, lo.y, hi.x, hi.y};
}

__global__ __launch_bounds__(256) void k_assoc_a(AsArgs a) {
    constexpr int NF4 = (GA_GROUPS * 256 + GA_BIAS * 16 + 16) / 4;
    __shared__ f32x4 lw[NF4];
    for (int i = threadIdx.x; i < NF4; i += 256) lw[i] = ((const f32x4*)a.packed)[i];
    __syncthreads();
    const float* lbias = (const float*)(lw + GA_GROUPS * 64);
    const float* lscal = lbias + GA_BIAS * 16;
    const float r1 = lscal[0], r2 = lscal[1], a0 = lscal[2], a11 = lscal[3], a12 = lscal[4];
    int lane = threadIdx.x & 63;
    const int j = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int S = a.S;
    ItemIter w(a.G, a.T, a.seg, a.nxcd, wave);
    for (; w.it < w.nitems; w.it += w.stride) {
        int gi, tb;
        w.decode(w.it, gi, tb);
        const int g = __builtin_amdgcn_readfirstlane(a.order[gi]);
        asm volatile("" : "+v"(lane));
        const int s = tb * 16 + j;
        const bool valid = s < S;
        const int sc = valid ? s : S - 1;
        const int su = a.sta_user != nullptr ? a.sta_user[sc] : sc;
        const long long pi = (long long)g * S + sc, pu = (long long)g * S + su;
        const float* pg = a.pg + (long long)g * AS_PG;
        const float eq = q < 3 ? a.edge_attr[pu * 3 + q] : 0.f;
        const float mq = a.mask[pu * 4 + q];
        const float m1 = pg[31];
        const f32x4 lat0 = ld_row30(a.x_latent + pu * 30, 0, q), lat1 = ld_row30(a.x_latent + pu * 30, 1, q);
        // BipartiteGraphReadOutOperator: one edge per product node, so aggr 'add' is the identity (module.py:343-352)
        f32x4 msg[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 z = *(const f32x4*)(pg + 16 * t + 4 * q);
            if (t == 1 && q == 3) z.w = 0.f;                                   // slot 31 carries mask1, not a channel
            z = MFMA16(lw[GA_FC1E(t) * 64 + lane].x, eq, z);
            if (a.save != nullptr && valid) *(f32x4*)(a.save + ((size_t)(AV_Z1 + t) * a.Pn + pi) * 16 + 4 * q) = z;
            msg[t] = prelu4u(z, r1) * m1;
        }
        f32x4 sv = *(const f32x4*)(lbias + 0 * 16 + 4 * q);
        sv = mma_block(sv, lw[GA_FC2(0) * 64 + lane], msg[0]);
        sv = mma_block(sv, lw[GA_FC2(1) * 64 + lane], msg[1]);
        if (a.save != nullptr && valid) *(f32x4*)(a.save + ((size_t)AV_SV * a.Pn + pi) * 16 + 4 * q) = sv;
        sv = prelu4u(sv, r2);
        // init_trns [s || x_latent || mask1 || Mask]
        f32x4 tr[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 acc = *(const f32x4*)(lbias + (1 + t) * 16 + 4 * q) + *(const f32x4*)(pg + 32 + 16 * t + 4 * q);
            if (a.ps != nullptr) acc += *(const f32x4*)(a.ps + (long long)su * AS_PS + 16 * t + 4 * q);
            acc = mma_block(acc, lw[GA_INIT(t, 0) * 64 + lane], sv);
            acc = mma_block(acc, lw[GA_INIT(t, 1) * 64 + lane], lat0);
            acc = mma_block(acc, lw[GA_INIT(t, 2) * 64 + lane], lat1);
            acc = MFMA16(lw[GA_INIT(t, 3) * 64 + lane].x, mq, acc);
            if (a.save != nullptr && valid) *(f32x4*)(a.save + ((size_t)(AV_TR + t) * a.Pn + pi) * 16 + 4 * q) = acc;
            tr[t] = prelu4u(acc, a0);
        }
        f32x4 qv[2][2];
#pragma unroll
        for (int wq = 0; wq < 2; ++wq)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 acc = *(const f32x4*)(lbias + (3 + 2 * wq + t) * 16 + 4 * q);
                acc = mma_block(acc, lw[GA_Q(wq, t, 0) * 64 + lane], tr[0]);
                acc = mma_block(acc, lw[GA_Q(wq, t, 1) * 64 + lane], tr[1]);
                if (a.save != nullptr && valid) *(f32x4*)(a.save + ((size_t)(AV_Q + 2 * wq + t) * a.Pn + pi) * 16 + 4 * q) = acc;
                qv[wq][t] = prelu4u(acc, wq == 0 ? a11 : a12);
            }
        if (valid) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                *(f32x4*)(a.tr + pi * 32 + 16 * t + 4 * q) = tr[t];
                *(f32x4*)(a.q1 + pi * 32 + 16 * t + 4 * q) = qv[0][t];
                *(f32x4*)(a.q2 + pi * 32 + 16 * t + 4 * q) = qv[1][t];
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_assoc_b(AsArgs a) {
    constexpr int NF4 = (GB_GROUPS * 256 + GB_BIAS * 16 + 16) / 4;
    __shared__ f32x4 lw[NF4];
    __shared__ __attribute__((aligned(16))) float tsc[4 * 16 * 68];
    for (int i = threadIdx.x; i < NF4; i += 256) lw[i] = ((const f32x4*)a.packed)[i];
    __syncthreads();
    const float* lbias = (const float*)(lw + GB_GROUPS * 64);
    const float* lscal = lbias + GB_BIAS * 16;
    const float a1 = lscal[0], a21 = lscal[1], a22 = lscal[2];
    int lane = threadIdx.x & 63;
    const int j = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // the 23 gathered 128-B rows of a node are read in the row layout lane = 4 r + cq (four consecutive lanes: one 64-B half of a
    // row); the four neighbour means then cross a per-wave LDS scratch into the MFMA layout (see k_stage2_ord, RL)
    const int jl = lane >> 2, ql = lane & 3;
    float* ts = tsc + wave * 16 * 68;
    const int S = a.S;
    ItemIter w(a.G, a.T, a.seg, a.nxcd, wave);
    for (; w.it < w.nitems; w.it += w.stride) {
        int gi, tb;
        w.decode(w.it, gi, tb);
        const int g = __builtin_amdgcn_readfirstlane(a.order[gi]);
        asm volatile("" : "+v"(lane));
        const int s = tb * 16 + j;
        const bool valid = s < S;
        const int sc = valid ? s : S - 1;
        const int su = a.sta_user != nullptr ? a.sta_user[sc] : sc;
        const long long pi = (long long)g * S + sc, pu = (long long)g * S + su;
        const float* pg = a.pg + (long long)g * AS_PG;
        const float mq = a.mask[pu * 4 + q];
        const f32x4 x0 = *(const f32x4*)(a.tr + pi * 32 + 4 * q), x1 = *(const f32x4*)(a.tr + pi * 32 + 16 + 4 * q);
        // neighbour means of q1 (stations of the same source node) and q2 (same station, neighbouring source nodes), edge order
        f32x4 n1a = {0.f, 0.f, 0.f, 0.f}, n1b = n1a, n2a = n1a, n2b = n1a;
        const int s_l = tb * 16 + jl, scl = s_l < S ? s_l : S - 1;        // the node whose rows this lane gathers
        {
            const int eb = a.sta_rowptr[scl], ee = a.sta_rowptr[scl + 1];
            const float* base = a.q1 + (long long)g * S * 32 + 4 * ql;
            for (int e = eb; __any(e < ee); e += 4) {
                f32x4 ra[4], rb[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool ok = e + k < ee;
                    const float* r = base + (long long)a.sta_col[ok ? e + k : max(ee - 1, 0)] * 32;
                    ra[k] = *(const f32x4*)r; rb[k] = *(const f32x4*)(r + 16);
                    if (!ok) { ra[k] = f32x4{0.f, 0.f, 0.f, 0.f}; rb[k] = ra[k]; }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) { n1a += ra[k]; n1b += rb[k]; }
            }
            const float inv = 1.f / (float)max(ee - eb, 1);
            n1a *= inv; n1b *= inv;
        }
        {
            const int eb = __builtin_amdgcn_readfirstlane(a.src_rowptr[g]);
            const int ee = __builtin_amdgcn_readfirstlane(a.src_rowptr[g + 1]);
            const float* base = a.q2 + (long long)scl * 32 + 4 * ql;
            int e = eb;
            for (; e + 4 <= ee; e += 4) {
                f32x4 ra[4], rb[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float* r = base + (long long)a.src_col[e + k] * S * 32;
                    ra[k] = *(const f32x4*)r; rb[k] = *(const f32x4*)(r + 16);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) { n2a += ra[k]; n2b += rb[k]; }
            }
            for (; e < ee; ++e) {
                const float* r = base + (long long)a.src_col[e] * S * 32;
                n2a += *(const f32x4*)r; n2b += *(const f32x4*)(r + 16);
            }
            const float inv = 1.f / (float)max(ee - eb, 1);
            n2a *= inv; n2b *= inv;
        }
        *(f32x4*)(ts + jl * 68 + 4 * ql) = n1a; *(f32x4*)(ts + jl * 68 + 16 + 4 * ql) = n1b;
        *(f32x4*)(ts + jl * 68 + 32 + 4 * ql) = n2a; *(f32x4*)(ts + jl * 68 + 48 + 4 * ql) = n2b;
        GSYNC();
        n1a = *(const f32x4*)(ts + j * 68 + 4 * q); n1b = *(const f32x4*)(ts + j * 68 + 16 + 4 * q);
        n2a = *(const f32x4*)(ts + j * 68 + 32 + 4 * q); n2b = *(const f32x4*)(ts + j * 68 + 48 + 4 * q);
        GSYNC();
        // layer 1
        f32x4 acc[4], w4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            acc[k] = *(const f32x4*)(lbias + k * 16 + 4 * q) + *(const f32x4*)(pg + 64 + 32 * (k >> 1) + 16 * (k & 1) + 4 * q);
        if (a.ps != nullptr) {
            acc[0] += *(const f32x4*)(a.ps + (long long)su * AS_PS + 32 + 4 * q);
            acc[1] += *(const f32x4*)(a.ps + (long long)su * AS_PS + 48 + 4 * q);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) w4[k] = lw[GB_L1(k >> 1, k & 1, 0) * 64 + lane];
        mma_blocks<4>(acc, w4, x0);
#pragma unroll
        for (int k = 0; k < 4; ++k) w4[k] = lw[GB_L1(k >> 1, k & 1, 1) * 64 + lane];
        mma_blocks<4>(acc, w4, x1);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const f32x4 na = b == 0 ? n1a : n1b, nb = b == 0 ? n2a : n2b;
            f32x4 wa[2], wb[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                wa[t] = lw[GB_L1(0, t, 2 + b) * 64 + lane];
                wb[t] = lw[GB_L1(1, t, 2 + b) * 64 + lane];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[0] = MFMA16(wa[0][r], na[r], acc[0]);
                acc[2] = MFMA16(wb[0][r], nb[r], acc[2]);
                acc[1] = MFMA16(wa[1][r], na[r], acc[1]);
                acc[3] = MFMA16(wb[1][r], nb[r], acc[3]);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = MFMA16(lw[GB_L1(k >> 1, k & 1, 4) * 64 + lane].x, mq, acc[k]);
        if (a.save != nullptr && valid) {
#pragma unroll
            for (int k = 0; k < 4; ++k) *(f32x4*)(a.save + ((size_t)(AV_T + k) * a.Pn + pi) * 16 + 4 * q) = acc[k];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = prelu4u(acc[k], a1);
        // r1 / r2 and the node-local layer-2 terms
        f32x4 o6[6], w6[6];
#pragma unroll
        for (int k = 0; k < 4; ++k) o6[k] = *(const f32x4*)(lbias + (4 + k) * 16 + 4 * q);
        o6[4] = *(const f32x4*)(lbias + 8 * 16 + 4 * q) + *(const f32x4*)(pg + 128 + 4 * q);
        o6[5] = *(const f32x4*)(lbias + 9 * 16 + 4 * q) + *(const f32x4*)(pg + 144 + 4 * q);
        if (a.ps != nullptr) o6[4] += *(const f32x4*)(a.ps + (long long)su * AS_PS + 64 + 4 * q);
#pragma unroll
        for (int hb = 0; hb < 4; ++hb) {
#pragma unroll
            for (int k = 0; k < 4; ++k) w6[k] = lw[GB_UV(k >> 1, k & 1, hb) * 64 + lane];
            w6[4] = lw[GB_C(0, hb) * 64 + lane];
            w6[5] = lw[GB_C(1, hb) * 64 + lane];
            mma_blocks<6>(o6, w6, acc[hb]);
        }
        o6[4] = MFMA16(lw[GB_C(0, 4) * 64 + lane].x, mq, o6[4]);
        o6[5] = MFMA16(lw[GB_C(1, 4) * 64 + lane].x, mq, o6[5]);
        if (a.save != nullptr && valid) {
#pragma unroll
            for (int k = 0; k < 4; ++k) *(f32x4*)(a.save + ((size_t)(AV_UV + k) * a.Pn + pi) * 16 + 4 * q) = o6[k];
        }
        o6[0] = prelu4u(o6[0], a21); o6[1] = prelu4u(o6[1], a21);
        o6[2] = prelu4u(o6[2], a22); o6[3] = prelu4u(o6[3], a22);
        f32x4 wuv[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, w2[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            w2[0] = lw[GB_W(0, b) * 64 + lane];
            w2[1] = lw[GB_W(1, b) * 64 + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                wuv[0] = MFMA16(w2[0][r], o6[b][r], wuv[0]);
                wuv[1] = MFMA16(w2[1][r], o6[2 + b][r], wuv[1]);
            }
        }
        if (valid) {
            *(f32x4*)(a.c + pi * ROWC + 4 * q) = o6[4];
            *(f32x4*)(a.c + pi * ROWC + 16 + 4 * q) = o6[5];
            *(f32x4*)(a.wu + pi * ROWW + 4 * q) = wuv[0];
            *(f32x4*)(a.wv + pi * ROWW + 4 * q) = wuv[1];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Backward of DataAggregation + Bipartite_ReadIn (training step, train_GENIE_model.py:1843-1861).
// The training forward is the generic stage kernels with `save` set (pre-activations of h0, h1, u, v, x_latent and the
// Bipartite message: 14 blocks of 16 floats per product node); the backward mirrors the stages in reverse order:
//   k_train_b2: d(station sum)[g] -> dz (message) -> dx_latent = fc1[:, 0:30]^T dz -> do = dx_latent PReLU2'(o)      store do
//   k_train_b1: transposed means of do1 / do2 (reversed base graphs) -> du, dv -> dh1 -> dt = dh1 PReLU1'(t)           store dt, dh0_local
//   k_train_b0: transposed means of dt1 / dt2 -> dh0 -> dz0
// Weight gradients are accumulated INSIDE the passes: dW[out, in] = sum over nodes dY[out] X[in] is an MFMA whose contraction
// runs over the 16 nodes of a tile (operands transposed through a per-wave LDS scratch), into register accumulators that live
// across all tiles of a wave; the terms that multiply a neighbour mean use the adjoint identity
//   sum_p dY[p] (x) mean_{k in N(p)} x[k] = sum_k (transposed mean of dY)[k] (x) x[k],
// so no mean is stored. Bias and PReLU-slope gradients are per-lane running sums. Every wave writes its partials once; a
// fixed-order reduction over the waves (k_train_reduce) makes the result run-to-run deterministic.
// ------------------------------------------------------------------------------------------------
constexpr int GR_DO = 0, GR_DT = 2, GR_DH0 = 6, GR_BLOCKS = 8;      // gradient rows kept between the passes: [GR_*][P][16]

struct AccDesc { int32_t mat_off, ld, row0, nrows, col0, ncols, n0, pad; };   // dW block: D[i][n] -> W[row0 + i][col0 + n], n0 <= n < ncols
struct VecDesc { int32_t off, row0, nrows, stride; };                // bias block: sum dY[i] -> b[(row0 + i) * stride]

struct TrArgs {
    int S, G, T, seg, nxcd;
    long long P;
    const int32_t* order;
    const int32_t* r_sta_rowptr; const int32_t* r_sta_col; const float* r_sta_w;    // reversed base graphs (out-edges, 1 / in-degree)
    const int32_t* r_src_rowptr; const int32_t* r_src_col; const float* r_src_w;
    const float* slice; const float* mask; const float* edge_attr;
    const float* save; float* gr;
    const float* dr;             // [G][32] gradient of the per-source-node station sum (Bipartite, before fc2)
    const float* packed;
    float* part;                 // per-wave partials: [wave][n_acc * 256 + n_vec * 16 + 16]
    int n_acc, n_vec;
    int sv_t, sv_up, sv_vp;      // k_train_b1: blocks of `save` holding the pre-activations of h1 / u / v (SV_T, SV_UP, SV_VP, or the
                                 // association phase's AV_T, AV_UV, AV_UV + 2)
    const float* pg;             // association phase (k_train_b1<true>, k_as_*): [G][AS_PG] per-source-node terms, pg[31] = mask1[g]
    const float* x_latent;       // association phase: [P, 30] DataAggregation output (an input of init_trns there)
    float* zsum;                 // k_as_b0: [G * T][32] per-tile station sums of d z1 (-> d y_latent, fc1's y_latent columns)
};

__device__ __forceinline__ f32x4 ldb(const float* buf, int blk, long long P, long long p, int q) {
    return *(const f32x4*)(buf + ((size_t)blk * P + p) * 16 + 4 * q);
}
__device__ __forceinline__ void stb(float* buf, int blk, long long P, long long p, int q, f32x4 v) {
    *(f32x4*)(buf + ((size_t)blk * P + p) * 16 + 4 * q) = v;
}
__device__ __forceinline__ f32x4 dprelu4(f32x4 x, float s) {     // PReLU'(x): 1 for x > 0, the slope otherwise
    return f32x4{x.x > 0.f ? 1.f : s, x.y > 0.f ? 1.f : s, x.z > 0.f ? 1.f : s, x.w > 0.f ? 1.f : s};
}
__device__ __forceinline__ float negsum4(f32x4 g, f32x4 x) {    // sum of g * min(x, 0): the slope gradient of PReLU
    return g.x * fminf(x.x, 0.f) + g.y * fminf(x.y, 0.f) + g.z * fminf(x.z, 0.f) + g.w * fminf(x.w, 0.f);
}
// V[ch 4q + r][node j] held by lane (j, q) -> vt[s] = V[ch j][node 4s + q]: the operand form of a node-contracting MFMA
__device__ __forceinline__ f32x4 tr16(f32x4 v, float* sc, int j, int q) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int r = 0; r < 4; ++r) sc[(4 * q + r) * 17 + j] = v[r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    f32x4 t;
#pragma unroll
    for (int s = 0; s < 4; ++s) t[s] = sc[j * 17 + 4 * s + q];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return t;
}
__device__ __forceinline__ f32x4 outer16(f32x4 acc, f32x4 at, f32x4 bt) {       // acc[out][in] += sum over the tile's nodes
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = MFMA16(at[s], bt[s], acc);
    return acc;
}
// transposed mean: sum over the out-edges e of `node` of w_e * rows[col_e], rows of 16 floats addressed by `rowof(block, col)`,
// for NB row blocks at once. The edges are taken EB at a time (indices, weights and the EB x NB rows of a batch are all in
// flight together; a lane past its last edge re-reads edge 0 with weight zero): with one index -> row load chain per edge the
// backward passes spent ~80 % of their time waiting on these gathers (one wave per SIMD, nothing to switch to). The sum keeps
// the edge order.
template <int NB, int EB = 4, typename F>
__device__ __forceinline__ void tmean_n(const int32_t* __restrict__ rp, const int32_t* __restrict__ col, const float* __restrict__ w,
                                        int node, bool uniform, F rowof, f32x4 (&out)[NB]) {
#pragma unroll
    for (int b = 0; b < NB; ++b) out[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    int eb = rp[node], ee = rp[node + 1];
    if (uniform) { eb = __builtin_amdgcn_readfirstlane(eb); ee = __builtin_amdgcn_readfirstlane(ee); }
    for (int e = eb; uniform ? (e < ee) : (bool)__any(e < ee); e += EB) {
        int c[EB];
        float ww[EB];
#pragma unroll
        for (int k = 0; k < EB; ++k) {
            const bool ok = e + k < ee;
            const int ei = ok ? e + k : 0;
            c[k] = col[ei];
            ww[k] = ok ? w[ei] : 0.f;
        }
        f32x4 r[NB][EB];
#pragma unroll
        for (int k = 0; k < EB; ++k)
#pragma unroll
            for (int b = 0; b < NB; ++b) r[b][k] = rowof(b, c[k]);
#pragma unroll
        for (int k = 0; k < EB; ++k)
#pragma unroll
            for (int b = 0; b < NB; ++b) out[b] += r[b][k] * ww[k];
    }
}
template <typename F>
__device__ __forceinline__ f32x4 tmean(const int32_t* rp, const int32_t* col, const float* w, int node, bool uniform, F rowof) {
    f32x4 o[1];
    tmean_n<1, 8>(rp, col, w, node, uniform, [&](int, int c) { return rowof(c); }, o);
    return o[0];
}
__device__ __forceinline__ void write_partials(const TrArgs& a, int wid, const f32x4* acc, int n_acc, const f32x4* vec, int n_vec,
                                               const float* scal, int n_scal, int lane, int j, int q) {
    float* out = a.part + (size_t)wid * ((size_t)a.n_acc * 256 + (size_t)a.n_vec * 16 + 16);
    for (int k = 0; k < n_acc; ++k) *(f32x4*)(out + (size_t)k * 256 + lane * 4) = acc[k];
    out += (size_t)a.n_acc * 256;
    for (int k = 0; k < n_vec; ++k) {
        f32x4 v = vec[k];
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            v.x += __shfl_xor(v.x, d); v.y += __shfl_xor(v.y, d); v.z += __shfl_xor(v.z, d); v.w += __shfl_xor(v.w, d);
        }
        if (j == 0) *(f32x4*)(out + k * 16 + 4 * q) = v;
    }
    out += (size_t)a.n_vec * 16;
    for (int k = 0; k < n_scal; ++k) {
        float v = scal[k];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d);
        if (lane == 0) out[k] = v;
    }
}

// ---- pass 2': Bipartite message + PReLU2.  accumulators: fc1 (t, {x_latent 0:15, x_latent 15:30, edge_attr}) = 6; vec: fc1 bias (2)
__global__ __launch_bounds__(256, 1) void k_train_b2(TrArgs a) {
    constexpr int NF4 = (GT2_GROUPS * 256 + 16) / 4;
    __shared__ f32x4 lw[NF4];
    __shared__ float tsc[4][16 * 17];
    for (int i = threadIdx.x; i < NF4; i += 256) lw[i] = ((const f32x4*)a.packed)[i];
    __syncthreads();
    const float* lscal = (const float*)(lw + GT2_GROUPS * 64);
    const float a2 = lscal[0], ab1 = lscal[1];
    int lane = threadIdx.x & 63;
    const int j = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* sc = tsc[wave];
    const int S = a.S;
    const long long P = a.P;
    f32x4 acc[6], vec[2];
    float scal[2] = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 6; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    vec[0] = vec[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    ItemIter w(a.G, a.T, a.seg, a.nxcd, wave);
    for (; w.it < w.nitems; w.it += w.stride) {
        int gi, tb;
        w.decode(w.it, gi, tb);
        const int g = __builtin_amdgcn_readfirstlane(a.order[gi]);
        asm volatile("" : "+v"(lane));
        const int s = tb * 16 + j;
        const bool valid = s < S;
        const int scn = valid ? s : S - 1;
        const long long p = (long long)g * S + scn;
        float mq = a.mask[p * 4 + q];
        float mm = fmaxf(mq, __shfl_xor(mq, 16));
        mm = fmaxf(mm, __shfl_xor(mm, 32));
        if (!valid) mm = 0.f;
        f32x4 eb = {0.f, 0.f, 0.f, 0.f};
        if (q == 0) { eb.x = a.edge_attr[p * 3]; eb.y = a.edge_attr[p * 3 + 1]; eb.z = a.edge_attr[p * 3 + 2]; }
        f32x4 dz[2], o[2], xl[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f32x4 zb = ldb(a.save, SV_ZB + t, P, p, q);
            const f32x4 d = *(const f32x4*)(a.dr + (long long)g * 32 + 16 * t + 4 * q) * mm;      // through the mask gate
            scal[1] += negsum4(d, zb);
            dz[t] = d * dprelu4(zb, ab1);
            vec[t] += dz[t];
            o[t] = ldb(a.save, SV_O + t, P, p, q);
            xl[t] = prelu4u(o[t], a2);
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            f32x4 dx = {0.f, 0.f, 0.f, 0.f};
            dx = mma_block(dx, lw[GT2(b, 0) * 64 + lane], dz[0]);
            dx = mma_block(dx, lw[GT2(b, 1) * 64 + lane], dz[1]);
            if (!valid) dx = f32x4{0.f, 0.f, 0.f, 0.f};
            scal[0] += negsum4(dx, o[b]);
            const f32x4 dob = dx * dprelu4(o[b], a2);
            if (valid) stb(a.gr, GR_DO + b, P, p, q, dob);
        }
        const f32x4 x0t = tr16(xl[0], sc, j, q), x1t = tr16(xl[1], sc, j, q), et = tr16(eb, sc, j, q);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f32x4 dt_ = tr16(dz[t], sc, j, q);
            acc[t * 3 + 0] = outer16(acc[t * 3 + 0], dt_, x0t);
            acc[t * 3 + 1] = outer16(acc[t * 3 + 1], dt_, x1t);
            acc[t * 3 + 2] = outer16(acc[t * 3 + 2], dt_, et);
        }
    }
    write_partials(a, blockIdx.x * 4 + wave, acc, 6, vec, 2, scal, 2, threadIdx.x & 63, j, q);
}

// ---- pass 1': layer 2 and the activation of layer 1.
// accumulators: l2_t1_2 {h1 x4, Mask, u x2 (adjoint)} = 7, l2_t2_2 = 7, l2_t1_1 (2 x h1 x4) = 8, l2_t2_1 = 8  -> 30
// vec: b(l2_t1_2), b(l2_t2_2), b(l2_t1_1) x2, b(l2_t2_1) x2 = 6; scal: a1, a21, a22
// AS: the same pass for DataAggregationAssociationPhase (module.py:397-401; 95-wide l2_t?_2 with mask width 5): the column of mask1
// (one value per source node) gets its gradient as two extra vectors.
template <bool AS>
__global__ __launch_bounds__(256, 1) void k_train_b1(TrArgs a) {
    constexpr int NF4 = (GT1_GROUPS * 256 + 16) / 4;
    __shared__ f32x4 lw[NF4];
    __shared__ float tsc[4][16 * 17];
    for (int i = threadIdx.x; i < NF4; i += 256) lw[i] = ((const f32x4*)a.packed)[i];
    __syncthreads();
    const float* lscal = (const float*)(lw + GT1_GROUPS * 64);
    const float a1 = lscal[0], a21 = lscal[1], a22 = lscal[2];
    int lane = threadIdx.x & 63;
    const int j = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* sc = tsc[wave];
    const int S = a.S;
    const long long P = a.P;
    constexpr int NV = AS ? 8 : 6;
    f32x4 acc[30], vec[NV];
    float scal[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 30; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < NV; ++k) vec[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int SVT = a.sv_t, SVU = a.sv_up, SVV = a.sv_vp;
    ItemIter w(a.G, a.T, a.seg, a.nxcd, wave);
    for (; w.it < w.nitems; w.it += w.stride) {
        int gi, tb;
        w.decode(w.it, gi, tb);
        const int g = __builtin_amdgcn_readfirstlane(a.order[gi]);
        asm volatile("" : "+v"(lane));
        const int s = tb * 16 + j;
        const bool valid = s < S;
        const int scn = valid ? s : S - 1;
        const long long p = (long long)g * S + scn;
        const float vm = valid ? 1.f : 0.f;
        f32x4 mb = {0.f, 0.f, 0.f, 0.f};
        if (q == 0) mb = *(const f32x4*)(a.mask + p * 4);
        const f32x4 do1 = ldb(a.gr, GR_DO + 0, P, p, q) * vm, do2 = ldb(a.gr, GR_DO + 1, P, p, q) * vm;
        if (AS) {
            const float m1 = a.pg[(long long)g * AS_PG + 31];
            vec[6] += do1 * m1; vec[7] += do2 * m1;
        }
        const float* gr = a.gr;
        const f32x4 tm1 = tmean(a.r_sta_rowptr, a.r_sta_col, a.r_sta_w, scn, false,
                                [&](int c) { return ldb(gr, GR_DO + 0, P, (long long)g * S + c, q); }) * vm;
        const f32x4 tm2 = tmean(a.r_src_rowptr, a.r_src_col, a.r_src_w, g, true,
                                [&](int c) { return ldb(gr, GR_DO + 1, P, (long long)c * S + scn, q); }) * vm;
        f32x4 t[4], h1[4], up[2], vp[2], u[2], v[2];
#pragma unroll
        for (int k = 0; k < 4; ++k) { t[k] = ldb(a.save, SVT + k, P, p, q); h1[k] = prelu4u(t[k], a1); }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            up[b] = ldb(a.save, SVU + b, P, p, q); u[b] = prelu4u(up[b], a21);
            vp[b] = ldb(a.save, SVV + b, P, p, q); v[b] = prelu4u(vp[b], a22);
        }
        // du = l2_t1_2[:, 60:90]^T tm1 through PReLU21', dv likewise
        f32x4 du[2], dv[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const f32x4 gu = mma_block(z, lw[GT_U(b) * 64 + lane], tm1), gv = mma_block(z, lw[GT_V(b) * 64 + lane], tm2);
            scal[1] += negsum4(gu, up[b]);
            scal[2] += negsum4(gv, vp[b]);
            du[b] = gu * dprelu4(up[b], a21);
            dv[b] = gv * dprelu4(vp[b], a22);
        }
        // dh1 and dt
        f32x4 dt[4];
#pragma unroll
        for (int hb = 0; hb < 4; ++hb) {
            f32x4 d = {0.f, 0.f, 0.f, 0.f};
            d = mma_block(d, lw[GT_H(hb, 0) * 64 + lane], du[0]);
            d = mma_block(d, lw[GT_H(hb, 1) * 64 + lane], du[1]);
            d = mma_block(d, lw[GT_H(hb, 2) * 64 + lane], dv[0]);
            d = mma_block(d, lw[GT_H(hb, 3) * 64 + lane], dv[1]);
            d = mma_block(d, lw[GT_H(hb, 4) * 64 + lane], do1);
            d = mma_block(d, lw[GT_H(hb, 5) * 64 + lane], do2);
            scal[0] += negsum4(d, t[hb]);
            dt[hb] = d * dprelu4(t[hb], a1);
            if (valid) stb(a.gr, GR_DT + hb, P, p, q, dt[hb]);
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) d = mma_block(d, lw[GT_D(b, k) * 64 + lane], dt[k]);
            if (valid) stb(a.gr, GR_DH0 + b, P, p, q, d);
        }
        vec[0] += do1; vec[1] += do2;
        vec[2] += du[0]; vec[3] += du[1]; vec[4] += dv[0]; vec[5] += dv[1];
        // weight gradients
        f32x4 h1t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) h1t[k] = tr16(h1[k], sc, j, q);
        const f32x4 mt = tr16(mb, sc, j, q);
        {
            const f32x4 d1t = tr16(do1, sc, j, q), d2t = tr16(do2, sc, j, q);
#pragma unroll
            for (int k = 0; k < 4; ++k) { acc[k] = outer16(acc[k], d1t, h1t[k]); acc[7 + k] = outer16(acc[7 + k], d2t, h1t[k]); }
            acc[4] = outer16(acc[4], d1t, mt);
            acc[11] = outer16(acc[11], d2t, mt);
            const f32x4 m1t = tr16(tm1, sc, j, q), m2t = tr16(tm2, sc, j, q);
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                acc[5 + b] = outer16(acc[5 + b], m1t, tr16(u[b], sc, j, q));
                acc[12 + b] = outer16(acc[12 + b], m2t, tr16(v[b], sc, j, q));
            }
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const f32x4 dut = tr16(du[b], sc, j, q), dvt = tr16(dv[b], sc, j, q);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc[14 + b * 4 + k] = outer16(acc[14 + b * 4 + k], dut, h1t[k]);
                acc[22 + b * 4 + k] = outer16(acc[22 + b * 4 + k], dvt, h1t[k]);
            }
        }
    }
    write_partials(a, blockIdx.x * 4 + wave, acc, 30, vec, NV, scal, 3, threadIdx.x & 63, j, q);
}

// ---- pass 0': layer 1 and init_trns.
// accumulators: init_trns (2 x [Slice || Mask]) = 2; l1_t1_2 {2 x (h0 x2, Mask), adjoint 2 x 2} = 10; l1_t2_2 = 10  -> 22
// vec: b(init_trns) x2, b(l1_t1_2) x2, b(l1_t2_2) x2 = 6; scal: a, a11, a12
__global__ __launch_bounds__(256, 1) void k_train_b0(TrArgs a) {
    constexpr int NF4 = (GT0_GROUPS * 256 + 16) / 4;
    __shared__ f32x4 lw[NF4];
    __shared__ float tsc[4][16 * 17];
    for (int i = threadIdx.x; i < NF4; i += 256) lw[i] = ((const f32x4*)a.packed)[i];
    __syncthreads();
    const float* lscal = (const float*)(lw + GT0_GROUPS * 64);
    const float a0 = lscal[0], a11 = lscal[1], a12 = lscal[2];
    int lane = threadIdx.x & 63;
    const int j = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* sc = tsc[wave];
    const int S = a.S;
    const long long P = a.P;
    f32x4 acc[22], vec[6];
    float scal[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 22; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 6; ++k) vec[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    ItemIter w(a.G, a.T, a.seg, a.nxcd, wave);
    for (; w.it < w.nitems; w.it += w.stride) {
        int gi, tb;
        w.decode(w.it, gi, tb);
        const int g = __builtin_amdgcn_readfirstlane(a.order[gi]);
        asm volatile("" : "+v"(lane));
        const int s = tb * 16 + j;
        const bool valid = s < S;
        const int scn = valid ? s : S - 1;
        const long long p = (long long)g * S + scn;
        const float vm = valid ? 1.f : 0.f;
        f32x4 xm = {0.f, 0.f, 0.f, 0.f}, mb = {0.f, 0.f, 0.f, 0.f};
        if (q == 0) { xm = *(const f32x4*)(a.slice + p * 4); mb = *(const f32x4*)(a.mask + p * 4); }
        if (q == 1) xm = *(const f32x4*)(a.mask + p * 4);
        const float* gr = a.gr;
        f32x4 z0[2], h0[2], dt[4], tmd1[2], tmd2[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            z0[b] = ldb(a.save, SV_Z0 + b, P, p, q);
            h0[b] = prelu4u(z0[b], a0);
        }
        tmean_n<2>(a.r_sta_rowptr, a.r_sta_col, a.r_sta_w, scn, false,
                   [&](int b, int c) { return ldb(gr, GR_DT + b, P, (long long)g * S + c, q); }, tmd1);
        tmean_n<2>(a.r_src_rowptr, a.r_src_col, a.r_src_w, g, true,
                   [&](int b, int c) { return ldb(gr, GR_DT + 2 + b, P, (long long)c * S + scn, q); }, tmd2);
#pragma unroll
        for (int b = 0; b < 2; ++b) { tmd1[b] *= vm; tmd2[b] *= vm; }
#pragma unroll
        for (int k = 0; k < 4; ++k) dt[k] = ldb(a.gr, GR_DT + k, P, p, q) * vm;
        f32x4 dz0[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            f32x4 dq1 = {0.f, 0.f, 0.f, 0.f}, dq2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                dq1 = mma_block(dq1, lw[GT_Q(0, b, k) * 64 + lane], tmd1[k]);
                dq2 = mma_block(dq2, lw[GT_Q(1, b, k) * 64 + lane], tmd2[k]);
            }
            scal[1] += negsum4(dq1, h0[b]);
            scal[2] += negsum4(dq2, h0[b]);
            const f32x4 dh0 = ldb(a.gr, GR_DH0 + b, P, p, q) * vm + dq1 * dprelu4(h0[b], a11) + dq2 * dprelu4(h0[b], a12);
            scal[0] += negsum4(dh0, z0[b]);
            dz0[b] = dh0 * dprelu4(z0[b], a0);
        }
        vec[0] += dz0[0]; vec[1] += dz0[1];
        vec[2] += dt[0]; vec[3] += dt[1]; vec[4] += dt[2]; vec[5] += dt[3];
        const f32x4 xmt = tr16(xm, sc, j, q), mt = tr16(mb, sc, j, q);
        const f32x4 h0t[2] = {tr16(h0[0], sc, j, q), tr16(h0[1], sc, j, q)};
        const f32x4 q1t[2] = {tr16(prelu4u(h0[0], a11), sc, j, q), tr16(prelu4u(h0[1], a11), sc, j, q)};
        const f32x4 q2t[2] = {tr16(prelu4u(h0[0], a12), sc, j, q), tr16(prelu4u(h0[1], a12), sc, j, q)};
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[b] = outer16(acc[b], tr16(dz0[b], sc, j, q), xmt);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int base = 2 + 10 * h;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const f32x4 dtt = tr16(dt[2 * h + b], sc, j, q);
                acc[base + b * 3 + 0] = outer16(acc[base + b * 3 + 0], dtt, h0t[0]);
                acc[base + b * 3 + 1] = outer16(acc[base + b * 3 + 1], dtt, h0t[1]);
                acc[base + b * 3 + 2] = outer16(acc[base + b * 3 + 2], dtt, mt);
                const f32x4 tmt = tr16(h == 0 ? tmd1[b] : tmd2[b], sc, j, q);
                acc[base + 6 + b * 2 + 0] = outer16(acc[base + 6 + b * 2 + 0], tmt, h == 0 ? q1t[0] : q2t[0]);
                acc[base + 6 + b * 2 + 1] = outer16(acc[base + 6 + b * 2 + 1], tmt, h == 0 ? q1t[1] : q2t[1]);
            }
        }
    }
    write_partials(a, blockIdx.x * 4 + wave, acc, 22, vec, 6, scal, 3, threadIdx.x & 63, j, q);
}

// fixed-order reduction of the per-wave partials into the gradient blob (registry layout of the weight mirror): a workgroup
// owns 32 entries; its 8 groups of 32 threads sum the waves w = group, group + 8, ... and the 8 sums are added in group order
__global__ __launch_bounds__(256) void k_train_reduce(const float* __restrict__ part, int n_waves, int n_acc, int n_vec, int n_scal,
                                                      const AccDesc* __restrict__ ad, const VecDesc* __restrict__ vd,
                                                      const int32_t* __restrict__ sd, float* __restrict__ blob, int accumulate) {
    __shared__ float ps[8][32];
    const int stride = n_acc * 256 + n_vec * 16 + 16;
    const int o = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int idx = blockIdx.x * 32 + o;
    float s = 0.f;
    if (idx < stride) {      // four independent partial sums (loads in flight), combined in a fixed order
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int wv = grp;
        for (; wv + 24 < n_waves; wv += 32) {
            s0 += part[(size_t)wv * stride + idx];
            s1 += part[(size_t)(wv + 8) * stride + idx];
            s2 += part[(size_t)(wv + 16) * stride + idx];
            s3 += part[(size_t)(wv + 24) * stride + idx];
        }
        for (; wv < n_waves; wv += 8) s0 += part[(size_t)wv * stride + idx];
        s = (s0 + s1) + (s2 + s3);
    }
    ps[grp][o] = s;
    __syncthreads();
    if (grp != 0 || idx >= stride) return;
    int dst = -1;
    if (idx < n_acc * 256) {
        const int k = idx >> 8, lane = (idx & 255) >> 2, r = idx & 3;
        const int i = 4 * (lane >> 4) + r, n = lane & 15;
        const AccDesc d = ad[k];
        if (i < d.nrows && n < d.ncols && n >= d.n0) dst = d.mat_off + (d.row0 + i) * d.ld + d.col0 + n;
    } else if (idx < n_acc * 256 + n_vec * 16) {
        const int k = (idx - n_acc * 256) >> 4, i = (idx - n_acc * 256) & 15;
        const VecDesc d = vd[k];
        if (i < d.nrows) dst = d.off + (d.row0 + i) * d.stride;
    } else {
        const int k = idx - n_acc * 256 - n_vec * 16;
        if (k < n_scal) dst = sd[k];
    }
    if (dst < 0) return;
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += ps[k][o];
    blob[dst] = accumulate ? blob[dst] + t : t;      // accumulate: parameters that several passes contribute to (each pass in stream order)
}

__global__ void k_part_sum(const float* __restrict__ part, int G, int T, float* __restrict__ r_out) {   // r[g] = sum over tiles
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= G * 30) return;
    const int g = idx / 30, c = idx - g * 30;
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += part[((size_t)g * T + t) * 32 + (c < 16 ? c : c)];
    r_out[idx] = s;
}

// sum over the 16 lanes of a DPP row (all lanes end with the total)
__device__ __forceinline__ float row_sum16(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));  // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
    return v;
}

// ------------------------------------------------------------------------------------------------
// Bipartite read-out of an irregular product graph (32 lanes per source node, weights transposed in LDS). The scalar form of
// the whole G-sized tail (32 lanes per node, both matvec operands from LDS: 3.5 LDS cycles per wave-FMA, tools/lds_matvec.hip)
// was replaced by the fp32-MFMA tile kernels below in round 2 (103.5 -> 45.4 us per window).
// ------------------------------------------------------------------------------------------------
constexpr int NPB = 8;  // nodes per 256-thread block

// Weight staging: global [rows][ld] row-major (nn.Linear layout) -> LDS [k][ldo] (k = input index, lane = output
// channel; conflict-free LDS writes and reads, strided but L1-resident global reads)
__device__ __forceinline__ void stage_transposed_ld(float* dst, const float* __restrict__ W, int rows, int ld, int ldo) {
    for (int i = threadIdx.x; i < ld * ldo; i += blockDim.x) {
        const int k = i / ldo, c = i - k * ldo;
        dst[i] = c < rows ? W[c * ld + k] : 0.f;
    }
}
__device__ __forceinline__ void stage_transposed(float* dst, const float* __restrict__ W, int rows, int ld) {
    stage_transposed_ld(dst, W, rows, ld, 32);
}

// r_g = sum of the message rows [seg[g], seg[g+1]) of a [P, 32] buffer (k_stage2_pcsr) in row order, out = PReLU_b2(fc2 r_g)  module.py:229
__global__ __launch_bounds__(256) void k_bip_out_seg(const float* __restrict__ rows, int G, const int32_t* __restrict__ seg,
                                                    const float* __restrict__ raw, int off_w, int off_b, int off_a,
                                                    float* __restrict__ out) {
    __shared__ float wt[30 * 32];
    __shared__ __attribute__((aligned(16))) float gx[NPB][32];
    stage_transposed(wt, raw + off_w, 15, 30);
    __syncthreads();
    const int c = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const float bias = c < 15 ? raw[off_b + c] : 0.f;
    const float act = raw[off_a];
    for (int g0 = blockIdx.x * NPB; g0 < G; g0 += gridDim.x * NPB) {
        const int g = g0 + grp;
        const bool ok = g < G;
        float r = 0.f;
        if (ok)
            for (long long pr = seg[g]; pr < seg[g + 1]; ++pr) r += rows[pr * 32 + c];
        gx[grp][c] = r;
        GSYNC();
        float o = bias;
#pragma unroll
        for (int k = 0; k < 30; ++k) o += wt[k * 32 + c] * gx[grp][k];
        GSYNC();
        if (ok && c < 15) out[(long long)g * 15 + c] = prelu1(o, act);
    }
}

// out-degree of every source node (number of edges whose message source is j)
__global__ void k_outdeg(const int32_t* __restrict__ col, long long E, int32_t* __restrict__ deg) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < E) atomicAdd(&deg[col[i]], 1);
}

struct SaArgs {
    int G, C;
    long long E;
    const float* x_in; const float* pos;
    const int32_t* rowptr; const int32_t* col; const int32_t* outdeg;
    const float* raw;
    int fc1_w, fc1_b, fc2_w, fc2_b, fg_w, fg_b, act1, act2, act3;     // this layer
    int nx_fc1_w, nx_fg_w, nx_fg_b, nx_act3;                           // next layer (k_sa_layer<.., NEXT = true>)
    float scale_rel;
    const float* pj_in;    // [G,32] x-part of this layer's messages: fc1.weight[:, 0:C] x_j
    const float* gpart_in; // [n_gpart_in][8] per-block partials of sum_j outdeg(j) PReLU3(fglobal x_j)
    int n_gpart_in;
    float* pj_out;         // [G,32] same for the next layer (NEXT) / for this layer (k_sa_pre)
    float* gpart_out;      // [gridDim][8]
    float* out;            // [G,30]
    const float* img;      // k_sa_pre_m / k_sa_layer_m: the layer's k_pack_all image (plan PL_SA1 + layer - 1)
    // batched tail: blockIdx.y = window; the window's copy of each buffer sits this many floats further on
    long long ws_x_in, ws_slot, ws_out;
};
__device__ __forceinline__ void sa_select_window(SaArgs& a) {
    const long long w = blockIdx.y;
    a.x_in += w * a.ws_x_in;
    if (a.pj_in) a.pj_in += w * a.ws_slot;
    if (a.gpart_in) a.gpart_in += w * a.ws_slot;
    if (a.pj_out) a.pj_out += w * a.ws_slot;
    if (a.gpart_out) a.gpart_out += w * a.ws_slot;
    if (a.out) a.out += w * a.ws_out;
}




// read-out heads (module.py:251-331): arguments shared by k_ro_pre_m / k_readout_m and their backward passes

struct RoArgs {
    int N, G, T;                 // nodes handled (G for MODE 0, Q for MODE 1), grid size, number of time queries (<= 16)
    int Nw;                      // batched tail: N = nwin * Nw node ids, window w = n / Nw reads x_spatial / cv of window w
    long long cv_ws;             // floats between the cv buffers of consecutive windows
    const float* x_spatial;      // [G,30]
    const float* x_grid;         // [G,3]   (MODE 1)
    const float* x_query;        // [Q,3]   (MODE 1)
    const int32_t* knn;          // [Q,10]  (MODE 1)
    const float* cv;             // [G,160] per-grid-node parts of f_context / f_values (MODE 1)
    const float* img;            // pre-transposed weight image of this MODE (k_pack_t)
    const float* t_query;        // [T]
    const float* raw;
    float scale_rel, scale_t;
    float* out;                  // [N,T]
    float* cv_out;               // MODE 0, optional: also write the per-grid-node table cv of MODE 1 (the work of k_ro_pre_m: one
    const float* pimg;           // pass over x_spatial and one launch less), with the PL_ROP image `pimg`; cv_ws as for `cv`
    float* lat_out;              // optional [N,30]: the head's latent input of TemporalAttention (MODE 0: SpatialDirect(x_spatial) = y_latent,
                                 // MODE 1: SpatialAttention(x_spatial, x_query)); k_readout_m only
    int o_sd_w, o_sd_b, o_sd_a;
    int o_q1w, o_q1b, o_q2w, o_q2b, o_c1w, o_c1b, o_c2w, o_c2b, o_v1w, o_v1b, o_v2w, o_v2b, o_p1w, o_p1b, o_p2w, o_p2b;
    int o_a1, o_a2, o_a3, o_a4, o_a5;
    int o_sq_w, o_sq_b, o_sc_w, o_sc_b, o_sv_w, o_sv_b, o_sp_w, o_sp_b, o_sa1, o_sa2;
};

// Per-grid-node part of SpatialAttention's edge Linears (module.py:290-291): f_context / f_values act on
// [x_j || edge_attr]; the x_j part  C_j = f_context.weight[:, 0:30] x_j,  V_j = f_values.weight[:, 0:30] x_j  is the same
// for every query that has j as a neighbour, so it is computed once per grid node: cv[j] = [C_j (75, pad 80) | V_j].
constexpr int CVP = 160;

constexpr int RO_K = 10;   // SpatialAttention neighbours (module.py:280 default k, asserted 10 elsewhere in the reference)
constexpr int RO_TMAX = 10;   // time queries per call (the reference uses 9, process_continuous_days.py:359)


// ------------------------------------------------------------------------------------------------
// The G- / Q-sized tail on fp32 MFMA tiles (plans PL_RO0 .. PL_BIP above): Bipartite read-out (module.py:229), SpatialAggregation
// (:243-249), SpatialDirect / SpatialAttention / TemporalAttention (:251-331) with the per-node Linears as MFMA chains over 16
// nodes per wave.
// ------------------------------------------------------------------------------------------------
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));

struct TlImg {                 // LDS copy of a k_pack_all image
    const f32x4* w; const float* bias; const float* scal;
};
__device__ __forceinline__ TlImg tl_stage_image(float* sm, const float* __restrict__ img, int n_groups, int n_bias) {
    const int n4 = (n_groups * 256 + n_bias * 16 + 16) / 4;
    for (int i = threadIdx.x; i < n4; i += blockDim.x) ((f32x4*)sm)[i] = ((const f32x4*)img)[i];
    TlImg im;
    im.w = (const f32x4*)sm; im.bias = sm + n_groups * 256; im.scal = im.bias + n_bias * 16;
    return im;
}
#define TLW(im, g) ((im).w[(g) * 64 + lane])
__device__ __forceinline__ f32x4 tl_bias(const TlImg& im, int tile, int q) { return *(const f32x4*)(im.bias + tile * 16 + 4 * q); }
// channel block t (channels 16t + 4q + {0..3}) of a 30-float row; the row is only 4-byte aligned and ends at channel 29
__device__ __forceinline__ f32x4 tl_load30(const float* __restrict__ row, int t, int q) {
    if (t == 1 && q == 3) { const f32x2u v = *(const f32x2u*)(row + 28); return f32x4{v.x, v.y, 0.f, 0.f}; }
    const f32x4u v = *(const f32x4u*)(row + 16 * t + 4 * q);
    return f32x4{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ void tl_store30(float* __restrict__ row, int t, int q, f32x4 v) {
    if (t == 1 && q == 3) { *(f32x2u*)(row + 28) = f32x2u{v.x, v.y}; return; }
    *(f32x4u*)(row + 16 * t + 4 * q) = f32x4u{v.x, v.y, v.z, v.w};
}
// the single channel block of a 15-float row
__device__ __forceinline__ f32x4 tl_load15(const float* __restrict__ row, int q) {
    if (q == 3) return f32x4{row[12], row[13], row[14], 0.f};
    const f32x4u v = *(const f32x4u*)(row + 4 * q);
    return f32x4{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ f32x4 tl_zero() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

// Bipartite read-out (module.py:229): r_g = sum over the tiles' partial rows in tile order, out = PReLU_b2(fc2 r_g).
__global__ __launch_bounds__(256) void k_bip_out_m(const float* __restrict__ part, int G, int T, const float* __restrict__ img,
                                                  float* __restrict__ out, long long part_ws, long long out_ws) {
    __shared__ __attribute__((aligned(16))) float sm[GB2_IMG_FLOATS + 4 * 16 * 36];
    const TlImg im = tl_stage_image(sm, img, GB_GROUPS2, GB_BIAS2);
    __syncthreads();
    part += blockIdx.y * part_ws;
    out += blockIdx.y * out_ws;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, q = lane >> 4;
    const int jl = lane >> 2, ql = lane & 3;          // row layout of the partial-row loads: four consecutive lanes read 64 contiguous bytes
    float* ts = sm + GB2_IMG_FLOATS + wave * 16 * 36;
    const float act = im.scal[0];
    const int ntiles = (G + 15) / 16;
    for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
        const int g = tile * 16 + j;
        const bool ok = g < G;
        const int gl = tile * 16 + jl;
        const float* pg = part + (long long)(gl < G ? gl : G - 1) * T * 32 + 4 * ql;
        f32x4 r0 = tl_zero(), r1 = tl_zero();
        int tb = 0;
        for (; tb + 4 <= T; tb += 4) {            // four rows in flight, added in tile order
            f32x4 v0[4], v1[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { v0[k] = *(const f32x4*)(pg + (tb + k) * 32); v1[k] = *(const f32x4*)(pg + (tb + k) * 32 + 16); }
#pragma unroll
            for (int k = 0; k < 4; ++k) { r0 += v0[k]; r1 += v1[k]; }
        }
        for (; tb < T; ++tb) { r0 += *(const f32x4*)(pg + tb * 32); r1 += *(const f32x4*)(pg + tb * 32 + 16); }
        *(f32x4*)(ts + jl * 36 + 4 * ql) = r0;    // row layout -> MFMA layout
        *(f32x4*)(ts + jl * 36 + 16 + 4 * ql) = r1;
        GSYNC();
        r0 = *(const f32x4*)(ts + j * 36 + 4 * q);
        r1 = *(const f32x4*)(ts + j * 36 + 16 + 4 * q);
        GSYNC();
        f32x4 o = tl_bias(im, 0, q);
        o = mma_block(o, TLW(im, 0), r0);
        o = mma_block(o, TLW(im, 1), r1);
        o = prelu4(o, act);
        if (ok) {
            float* og = out + (long long)g * 15 + 4 * q;
            og[0] = o.x; og[1] = o.y; og[2] = o.z;
            if (q < 3) og[3] = o.w;
        }
    }
}

// fixed-order reduction of the per-lane global-term partials (rows m = 4q + r < 5 of the fglobal tile) -> gpart[block][m]
__device__ __forceinline__ void tl_store_gpart(f32x4 acc, int lane, int wave, float* red, float* __restrict__ gpart_out) {
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
        acc.x += __shfl_xor(acc.x, d); acc.y += __shfl_xor(acc.y, d); acc.z += __shfl_xor(acc.z, d); acc.w += __shfl_xor(acc.w, d);
    }
    const int j = lane & 15, q = lane >> 4;
    if (j == 0 && q < 2) *(f32x4*)(red + wave * 8 + 4 * q) = acc;
    __syncthreads();
    if (threadIdx.x < 8) {
        float s = 0.f;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) s += red[k * 8 + threadIdx.x];
        gpart_out[blockIdx.x * 8 + threadIdx.x] = threadIdx.x < 5 ? s : 0.f;
    }
}

// Pre-pass of a SpatialAggregation layer (see k_sa_pre): pj[j] = fc1.weight[:, 0:C] x_j and the block partial of
// sum_j outdeg(j) PReLU3(fglobal x_j).
template <int C>
__global__ __launch_bounds__(256) void k_sa_pre_m(SaArgs a) {
    sa_select_window(a);
    __shared__ __attribute__((aligned(16))) float sm[GS_IMG_FLOATS + 32];
    const TlImg im = tl_stage_image(sm, a.img, GS_GROUPS, GS_BIAS);
    float* red = sm + GS_IMG_FLOATS;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, q = lane >> 4;
    const float act3 = im.scal[3];
    f32x4 acc = tl_zero();
    const int ntiles = (a.G + 15) / 16;
    for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
        const int g = tile * 16 + j;
        const bool ok = g < a.G;
        const float* row = a.x_in + (long long)(ok ? g : a.G - 1) * C;
        f32x4 xb[2];
        if (C == 15) { xb[0] = tl_load15(row, q); xb[1] = tl_zero(); }
        else { xb[0] = tl_load30(row, 0, q); xb[1] = tl_load30(row, 1, q); }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 pj = mma_block(tl_zero(), TLW(im, GS_PJ(t, 0)), xb[0]);
            if (C == 30) pj = mma_block(pj, TLW(im, GS_PJ(t, 1)), xb[1]);
            if (ok) *(f32x4*)(a.pj_out + (long long)g * 32 + 16 * t + 4 * q) = pj;
        }
        f32x4 gl = mma_block(tl_bias(im, 5, q), TLW(im, GS_FG(0)), xb[0]);
        if (C == 30) gl = mma_block(gl, TLW(im, GS_FG(1)), xb[1]);
        if (ok) acc += prelu4(gl, act3) * (float)a.outdeg[g];
    }
    tl_store_gpart(acc, lane, wave, red, a.gpart_out);
}

// k_bip_out_m + k_sa_pre_m<15> in one launch (the batched tail): the Bipartite output of a node is the input of
// SpatialAggregation1's pre-pass of the same node. Same MFMA chains as the two kernels (bitwise equal results).
__global__ __launch_bounds__(256) void k_bip_pre_m(const float* __restrict__ part, int T, const float* __restrict__ img_bip, long long part_ws,
                                                  SaArgs a) {
    sa_select_window(a);
    __shared__ __attribute__((aligned(16))) float sm[GB2_IMG_FLOATS + 4 * 16 * 36 + GS_IMG_FLOATS + 32];
    const TlImg im = tl_stage_image(sm, img_bip, GB_GROUPS2, GB_BIAS2);
    float* tsc = sm + GB2_IMG_FLOATS;
    const TlImg is = tl_stage_image(tsc + 4 * 16 * 36, a.img, GS_GROUPS, GS_BIAS);
    float* red = tsc + 4 * 16 * 36 + GS_IMG_FLOATS;
    __syncthreads();
    part += blockIdx.y * part_ws;
    const int G = a.G;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, q = lane >> 4;
    const int jl = lane >> 2, ql = lane & 3;
    float* ts = tsc + wave * 16 * 36;
    const float act = im.scal[0], act3 = is.scal[3];
    f32x4 acc = tl_zero();
    const int ntiles = (G + 15) / 16;
    for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
        const int g = tile * 16 + j;
        const bool ok = g < G;
        const int gl = tile * 16 + jl;
        const float* pg = part + (long long)(gl < G ? gl : G - 1) * T * 32 + 4 * ql;
        f32x4 r0 = tl_zero(), r1 = tl_zero();
        int tb = 0;
        for (; tb + 4 <= T; tb += 4) {            // four rows in flight, added in tile order
            f32x4 v0[4], v1[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { v0[k] = *(const f32x4*)(pg + (tb + k) * 32); v1[k] = *(const f32x4*)(pg + (tb + k) * 32 + 16); }
#pragma unroll
            for (int k = 0; k < 4; ++k) { r0 += v0[k]; r1 += v1[k]; }
        }
        for (; tb < T; ++tb) { r0 += *(const f32x4*)(pg + tb * 32); r1 += *(const f32x4*)(pg + tb * 32 + 16); }
        *(f32x4*)(ts + jl * 36 + 4 * ql) = r0;    // row layout -> MFMA layout
        *(f32x4*)(ts + jl * 36 + 16 + 4 * ql) = r1;
        GSYNC();
        r0 = *(const f32x4*)(ts + j * 36 + 4 * q);
        r1 = *(const f32x4*)(ts + j * 36 + 16 + 4 * q);
        GSYNC();
        f32x4 o = tl_bias(im, 0, q);
        o = mma_block(o, TLW(im, 0), r0);
        o = mma_block(o, TLW(im, 1), r1);
        o = prelu4(o, act);
        if (q == 3) o.w = 0.f;                    // channel 15 does not exist (tl_load15 of the stored row reads it as zero)
        if (ok) {
            float* og = a.out + (long long)g * 15 + 4 * q;
            og[0] = o.x; og[1] = o.y; og[2] = o.z;
            if (q < 3) og[3] = o.w;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f32x4 pj = mma_block(tl_zero(), TLW(is, GS_PJ(t, 0)), o);
            if (ok) *(f32x4*)(a.pj_out + (long long)g * 32 + 16 * t + 4 * q) = pj;
        }
        const f32x4 glb = mma_block(tl_bias(is, 5, q), TLW(is, GS_FG(0)), o);
        if (ok) acc += prelu4(glb, act3) * (float)a.outdeg[g];
    }
    tl_store_gpart(acc, lane, wave, red, a.gpart_out);
}

// One SpatialAggregation layer (see k_sa_layer): per-edge messages on the VALU (8 channels per lane), fc2 and the next layer's
// pre-pass as MFMA chains on the 16 nodes of the wave.
template <int C, bool NEXT>
__global__ __launch_bounds__(256) void k_sa_layer_m(SaArgs a) {
    sa_select_window(a);
    __shared__ __attribute__((aligned(16))) float sm[GS_IMG_FLOATS + 8 * 32 + 8 + 32 * 8 + 32 + 4 * 16 * 36];
    const TlImg im = tl_stage_image(sm, a.img, GS_GROUPS, GS_BIAS);
    float* w1p = sm + GS_IMG_FLOATS;         // fc1 columns C..C+7 (3 position + 5 global), [k][32]
    float* gsum = w1p + 8 * 32;
    float* gred = gsum + 8;                  // [32][8]
    float* red = gred + 32 * 8;
    float* tsc = red + 32;                   // per wave [16][36]: edge means, row layout -> MFMA layout
    for (int i = threadIdx.x; i < 8 * 32; i += blockDim.x) {
        const int k = i >> 5, cc = i & 31;
        w1p[i] = cc < 30 ? a.raw[a.fc1_w + cc * (C + 8) + C + k] : 0.f;
    }
    {   // global term: the producer's per-block partials in the same fixed two-level order as k_sa_layer
        const int m = threadIdx.x & 7, chunk = threadIdx.x >> 3;
        float sgl = 0.f;
        for (int b = chunk; b < a.n_gpart_in; b += 32) sgl += a.gpart_in[b * 8 + m];
        gred[chunk * 8 + m] = sgl;
        __syncthreads();
        if (threadIdx.x < 8) {
            float t = 0.f;
            for (int k = 0; k < 32; ++k) t += gred[k * 8 + threadIdx.x];
            gsum[threadIdx.x] = t;
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, q = lane >> 4;
    // The per-edge phase is elementwise per (node, channel) and runs in the ROW layout lane = 4 r + cq (node r, chunk cq): four
    // consecutive lanes read one 64-B half of a gathered pj row (in the MFMA layout they read 16-B chunks of four different
    // rows: a quarter of the texture path's rate); the edge means cross a per-wave LDS scratch into the MFMA layout for fc2.
    const int jl = lane >> 2, ql = lane & 3;
    float* ts = tsc + wave * 16 * 36;
    const float act1 = im.scal[0], act2 = im.scal[1], act3n = im.scal[2];
    f32x4 base[2], wp[3][2];
    {
        const float invE = 1.f / (float)(a.E > 0 ? a.E : 1);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            base[t] = tl_bias(im, 2 + t, ql);
#pragma unroll
            for (int m = 0; m < 5; ++m) base[t] += *(const f32x4*)(w1p + (3 + m) * 32 + 16 * t + 4 * ql) * (gsum[m] * invE);
#pragma unroll
            for (int d = 0; d < 3; ++d) wp[d][t] = *(const f32x4*)(w1p + d * 32 + 16 * t + 4 * ql);
        }
    }
    f32x4 acc = tl_zero();
    const int ntiles = (a.G + 15) / 16;
    for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
        const int i = tile * 16 + j;
        const bool ok = i < a.G;
        const int ic = ok ? i : a.G - 1;
        const float* row = a.x_in + (long long)ic * C;
        f32x4 xb[2];
        if (C == 15) { xb[0] = tl_load15(row, q); xb[1] = tl_zero(); }
        else { xb[0] = tl_load30(row, 0, q); xb[1] = tl_load30(row, 1, q); }
        // ---- edges of node il (row layout)
        const int il = tile * 16 + jl;
        const bool okl = il < a.G;
        const int icl = okl ? il : a.G - 1;
        const float pi0 = a.pos[icl * 3 + 0] / a.scale_rel, pi1 = a.pos[icl * 3 + 1] / a.scale_rel, pi2 = a.pos[icl * 3 + 2] / a.scale_rel;
        const int eb = a.rowptr[icl], ee = okl ? a.rowptr[icl + 1] : eb;
        f32x4 as[2] = {tl_zero(), tl_zero()};
        // edges in chunks of 4: ids, the gathered rows / positions in flight, then the arithmetic in edge order (no cross-lane
        // operation inside: the nodes of a wave may differ in trip count)
        for (int e0 = eb; e0 < ee; e0 += 4) {
            int jn[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) jn[k] = a.col[min(e0 + k, ee - 1)];
            f32x4 pjv[4][2];
            float q0[4], q1[4], q2[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                pjv[k][0] = *(const f32x4*)(a.pj_in + (long long)jn[k] * 32 + 4 * ql);
                pjv[k][1] = *(const f32x4*)(a.pj_in + (long long)jn[k] * 32 + 16 + 4 * ql);
                q0[k] = a.pos[jn[k] * 3 + 0]; q1[k] = a.pos[jn[k] * 3 + 1]; q2[k] = a.pos[jn[k] * 3 + 2];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float d0 = pi0 - q0[k] / a.scale_rel, d1 = pi1 - q1[k] / a.scale_rel, d2 = pi2 - q2[k] / a.scale_rel;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f32x4 m = pjv[k][t] + base[t];
                    m += wp[0][t] * d0;
                    m += wp[1][t] * d1;
                    m += wp[2][t] * d2;
                    if (e0 + k < ee) as[t] += prelu4(m, act1);
                }
            }
        }
        const float deg = (float)max(ee - eb, 1);
        *(f32x4*)(ts + jl * 36 + 4 * ql) = as[0] / deg;
        *(f32x4*)(ts + jl * 36 + 16 + 4 * ql) = as[1] / deg;
        GSYNC();
        const f32x4 av[2] = {*(const f32x4*)(ts + j * 36 + 4 * q), *(const f32x4*)(ts + j * 36 + 16 + 4 * q)};
        GSYNC();
        f32x4 o[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 v = mma_block(tl_bias(im, t, q), TLW(im, GS_FC2(t, 0)), xb[0]);
            if (C == 30) v = mma_block(v, TLW(im, GS_FC2(t, 1)), xb[1]);
            v = mma_block(v, TLW(im, GS_FC2(t, 2)), av[0]);
            v = mma_block(v, TLW(im, GS_FC2(t, 3)), av[1]);
            o[t] = prelu4(v, act2);
            if (ok) tl_store30(a.out + (long long)i * 30, t, q, o[t]);
        }
        if (NEXT) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 pj = mma_block(tl_zero(), TLW(im, GS_PJN(t, 0)), o[0]);
                pj = mma_block(pj, TLW(im, GS_PJN(t, 1)), o[1]);
                if (ok) *(f32x4*)(a.pj_out + (long long)i * 32 + 16 * t + 4 * q) = pj;
            }
            f32x4 gl = mma_block(tl_bias(im, 4, q), TLW(im, GS_FGN(0)), o[0]);
            gl = mma_block(gl, TLW(im, GS_FGN(1)), o[1]);
            if (ok) acc += prelu4(gl, act3n) * (float)a.outdeg[i];
        }
    }
    if (NEXT) tl_store_gpart(acc, lane, wave, red, a.gpart_out);
}

// Per-grid-node part of SpatialAttention's edge Linears (see k_ro_pre), biases included, in a head-padded layout:
// cv[j] = [f_context: head h at 16h + l (l < 15, slot 15 zero) | f_values: 80 + 16h + l], CVP floats per node.
__global__ __launch_bounds__(256) void k_ro_pre_m(const float* __restrict__ x_spatial, int G, const float* __restrict__ img,
                                                 float* __restrict__ cv, int Gw, long long cv_ws) {
    __shared__ __attribute__((aligned(16))) float sm[GP_IMG_FLOATS];
    const TlImg im = tl_stage_image(sm, img, GP_GROUPS, GP_BIAS);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, q = lane >> 4;
    const int ntiles = (G + 15) / 16;
    for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
        const int g = tile * 16 + j;
        const bool ok = g < G;
        const int gc = ok ? g : G - 1;
        const float* row = x_spatial + (long long)gc * 30;
        const f32x4 xb0 = tl_load30(row, 0, q), xb1 = tl_load30(row, 1, q);
        const int w = gc / Gw;
        float* o = cv + w * cv_ws + (long long)(gc - w * Gw) * CVP + 4 * q;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int h = 0; h < 5; ++h) {
                f32x4 v = mma_block(tl_bias(im, m * 5 + h, q), TLW(im, GP(m, h, 0)), xb0);
                v = mma_block(v, TLW(im, GP(m, h, 1)), xb1);
                if (ok) *(f32x4*)(o + m * 80 + h * 16) = v;
            }
    }
}

// Read-out heads (see k_readout). MODE 0: y = TemporalAttention(SpatialDirect(x_spatial)) per grid node; MODE 1:
// x = TemporalAttention(SpatialAttention(x_spatial, x_query, x_grid)) per query. SpatialAttention's per-edge arithmetic runs on
// the VALU head by head (a lane holds 4 of a head's 16 slots for its query; the head dot product is a 4-lane butterfly); the
// attention scores of TemporalAttention are an MFMA against the time-query fragments (score[t] = Q_h[t, :] . ctx_h), and the
// score x value products, per node, go through a per-wave LDS scratch (a lane needs all T x 5 scores of its node).
constexpr int RO_SCS = 68;      // floats per node in the score scratch: [5 heads][12 time slots] + pad
constexpr int ROM_LDS_FLOATS = GR_IMG_FLOATS + 5 * 256 + 10 * 80 + 4 * 16 * RO_SCS;
template <int MODE>
__global__ __launch_bounds__(256) void k_readout_m(RoArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const TlImg im = tl_stage_image(sm, a.img, GR_GROUPS, GR_BIAS);
    float* qf = sm + GR_IMG_FLOATS;          // [5][64][4]: A fragments of the temporal queries, head h
    float* et = qf + 5 * 256;                // [10][80] (MODE 1): f_queries columns 0..2, f_context / f_values edge columns, f_queries bias
    float* scr = et + 10 * 80;
    TlImg imp = im;                          // MODE 0 with cv_out: the PL_ROP image behind the score scratch
    if (MODE == 0 && a.cv_out != nullptr) imp = tl_stage_image(scr + 4 * 16 * RO_SCS, a.pimg, GP_GROUPS, GP_BIAS);
    {   // qf[h][lane][r] = query[t = lane & 15][head h][l = 4 (lane >> 4) + r], query = temporal_query_2(PReLU3(temporal_query_1(t / scale_t)))  :329
        const float act3 = a.raw[a.o_a3];
        for (int i = threadIdx.x; i < 5 * 256; i += blockDim.x) {
            const int h = i >> 8, ln = (i & 255) >> 2, r = i & 3, t = ln & 15, l = 4 * (ln >> 4) + r;
            float v = 0.f;
            if (t < a.T && l < 15) {
                const int ch = 15 * h + l;
                const float tq = a.t_query[t] / a.scale_t;
                v = a.raw[a.o_q2b + ch];
                for (int k = 0; k < 30; ++k) {
                    const float hq = prelu1(a.raw[a.o_q1w + k] * tq + a.raw[a.o_q1b + k], act3);
                    v += a.raw[a.o_q2w + ch * 30 + k] * hq;
                }
            }
            qf[i] = v;
        }
        if (MODE == 1) {
            for (int i = threadIdx.x; i < 10 * 80; i += blockDim.x) {
                const int m = i / 80, rem = i - m * 80, h = rem >> 4, l = rem & 15, ch = 15 * h + l;
                float v = 0.f;
                if (l < 15) {
                    if (m < 3) v = a.raw[a.o_sq_w + ch * 3 + m];
                    else if (m < 6) v = a.raw[a.o_sc_w + ch * 33 + 30 + (m - 3)];
                    else if (m < 9) v = a.raw[a.o_sv_w + ch * 33 + 30 + (m - 6)];
                    else v = a.raw[a.o_sq_b + ch];
                }
                et[i] = v;
            }
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, q = lane >> 4;
    const float fa = im.scal[0], sa1 = im.scal[1], act1 = im.scal[2], act2 = im.scal[3], act4 = im.scal[4], act5 = im.scal[5];
    const float b_p2 = im.scal[6];
    const float inv_sqrt_l = 1.f / sqrtf(15.f);
    float* ws = scr + (wave * 16 + j) * RO_SCS;
    const int ntiles = (a.N + 15) / 16;
    for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
        const int n = tile * 16 + j;
        const bool ok = n < a.N;
        const int nc = ok ? n : a.N - 1;
        f32x4 xin[2];
        if (MODE == 0) {
            const float* row = a.x_spatial + (long long)nc * 30;
            const f32x4 xb0 = tl_load30(row, 0, q), xb1 = tl_load30(row, 1, q);
            if (a.cv_out != nullptr) {                                                  // k_ro_pre_m's work for this node (same MFMA chains)
                const int w = nc / a.Nw;
                float* o = a.cv_out + w * a.cv_ws + (long long)(nc - w * a.Nw) * CVP + 4 * q;
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int h = 0; h < 5; ++h) {
                        f32x4 v = mma_block(tl_bias(imp, m * 5 + h, q), TLW(imp, GP(m, h, 0)), xb0);
                        v = mma_block(v, TLW(imp, GP(m, h, 1)), xb1);
                        if (ok) *(f32x4*)(o + m * 80 + h * 16) = v;
                    }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {                                               // SpatialDirect  :258-260
                f32x4 y = mma_block(tl_bias(im, t, q), TLW(im, GR_FRONT(t, 0)), xb0);
                y = mma_block(y, TLW(im, GR_FRONT(t, 1)), xb1);
                xin[t] = prelu4(y, fa);
            }
        } else {
            // SpatialAttention in the ROW layout lane = 4 r + cq (query r = lane >> 2, chunk cq = lane & 3): four consecutive lanes
            // read one 64-B head block of a gathered cv row (the MFMA layout reads 16-B chunks of four rows per quad: a quarter of
            // the texture path's rate on the 6.4 KB a query gathers), the head sums are butterflies inside a quad; the aggregated
            // vector crosses the wave's LDS scratch into the MFMA layout for proj.
            const int jl = lane >> 2, ql = lane & 3;
            const int n_l = tile * 16 + jl;
            const int ncl = n_l < a.N ? n_l : a.N - 1;
            const int wq = ncl / a.Nw, nl = ncl - wq * a.Nw;
            const float* cvw = a.cv + wq * a.cv_ws + 4 * ql;
            const float xq0 = a.x_query[nl * 3 + 0], xq1 = a.x_query[nl * 3 + 1], xq2 = a.x_query[nl * 3 + 2];
            int jn[RO_K];
            float e[RO_K][3];
#pragma unroll
            for (int k = 0; k < RO_K; ++k) jn[k] = a.knn[(long long)nl * RO_K + k];
#pragma unroll
            for (int k = 0; k < RO_K; ++k) {                                            // edge_attr  :283
                e[k][0] = (xq0 - a.x_grid[jn[k] * 3 + 0]) / a.scale_rel;
                e[k][1] = (xq1 - a.x_grid[jn[k] * 3 + 1]) / a.scale_rel;
                e[k][2] = (xq2 - a.x_grid[jn[k] * 3 + 2]) / a.scale_rel;
            }
            f32x4 xm = tl_zero();
#pragma unroll
            for (int h = 0; h < 5; ++h) {
                const float* eh = et + h * 16 + 4 * ql;
                const f32x4 bq = *(const f32x4*)(eh + 9 * 80);
                f32x4 wq_[3], wc_[3], wv_[3];
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    wq_[d] = *(const f32x4*)(eh + d * 80); wc_[d] = *(const f32x4*)(eh + (3 + d) * 80); wv_[d] = *(const f32x4*)(eh + (6 + d) * 80);
                }
                f32x4 cvk[RO_K];
#pragma unroll
                for (int k = 0; k < RO_K; ++k) cvk[k] = *(const f32x4*)(cvw + (long long)jn[k] * CVP + h * 16);
                float al[RO_K];
#pragma unroll
                for (int k = 0; k < RO_K; ++k) {                                        // alpha = PReLU1(sum_l q c / sqrt(L))  :293
                    f32x4 q4 = bq, c4 = cvk[k];
#pragma unroll
                    for (int d = 0; d < 3; ++d) { q4 += wq_[d] * e[k][d]; c4 += wc_[d] * e[k][d]; }
                    const f32x4 pr = q4 * c4;
                    al[k] = ((pr.x + pr.y) + pr.z) + pr.w;
                }
#pragma unroll
                for (int k = 0; k < RO_K; ++k) cvk[k] = *(const f32x4*)(cvw + (long long)jn[k] * CVP + 80 + h * 16);
#pragma unroll
                for (int k = 0; k < RO_K; ++k) {
                    al[k] += __shfl_xor(al[k], 1);
                    al[k] += __shfl_xor(al[k], 2);
                    al[k] = prelu1(al[k] * inv_sqrt_l, sa1);
                }
                float mx = al[0];                                                       // segment softmax over the K edges  :295
#pragma unroll
                for (int k = 1; k < RO_K; ++k) mx = fmaxf(mx, al[k]);
                float ssum = 0.f;
#pragma unroll
                for (int k = 0; k < RO_K; ++k) { al[k] = expf(al[k] - mx); ssum += al[k]; }
                const float den = ssum + 1e-16f;
                f32x4 gh = tl_zero();
#pragma unroll
                for (int k = 0; k < RO_K; ++k) {                                        // 'add' aggregation of alpha * v  :264,297
                    f32x4 v4 = cvk[k];
#pragma unroll
                    for (int d = 0; d < 3; ++d) v4 += wv_[d] * e[k][d];
                    gh += v4 * (al[k] / den);
                }
                xm += gh;
            }
            xm *= 0.2f;                                                                 // mean over heads  :285
            float* wsb = scr + wave * 16 * RO_SCS;
            *(f32x4*)(wsb + jl * RO_SCS + 4 * ql) = xm;                                 // row layout -> MFMA layout
            GSYNC();
            xm = *(const f32x4*)(wsb + j * RO_SCS + 4 * q);
            GSYNC();
#pragma unroll
            for (int t = 0; t < 2; ++t)                                                 // PReLU2(proj(.))  :285
                xin[t] = prelu4(mma_block(tl_bias(im, t, q), TLW(im, GR_FRONT(t, 0)), xm), fa);
        }
        if (a.lat_out != nullptr && ok) {
            tl_store30(a.lat_out + (long long)n * 30, 0, q, xin[0]);
            tl_store30(a.lat_out + (long long)n * 30, 1, q, xin[1]);
        }
        // ------------------------------------------------------------------ TemporalAttention on xin  :325-331
        f32x4 h1[2], h2[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 c1 = mma_block(tl_bias(im, 2 + t, q), TLW(im, GR_C1(t, 0)), xin[0]);
            c1 = mma_block(c1, TLW(im, GR_C1(t, 1)), xin[1]);
            h1[t] = prelu4(c1, act1);
            f32x4 v1 = mma_block(tl_bias(im, 4 + t, q), TLW(im, GR_V1(t, 0)), xin[0]);
            v1 = mma_block(v1, TLW(im, GR_V1(t, 1)), xin[1]);
            h2[t] = prelu4(v1, act2);
        }
        f32x4 val[5];
#pragma unroll
        for (int h = 0; h < 5; ++h) {
            f32x4 cx = mma_block(tl_bias(im, 6 + h, q), TLW(im, GR_C2(h, 0)), h1[0]);
            cx = mma_block(cx, TLW(im, GR_C2(h, 1)), h1[1]);
            // score[t, h] = ctx[h, :] . query[t, h, :] / sqrt(L): rows t = 4q + r of the result
            const f32x4 sc = mma_block(tl_zero(), ((const f32x4*)qf)[h * 64 + lane], cx) * inv_sqrt_l;
            if (q < 3) *(f32x4*)(ws + h * 12 + 4 * q) = sc;
            f32x4 vx = mma_block(tl_bias(im, 11 + h, q), TLW(im, GR_V2(h, 0)), h2[0]);
            val[h] = mma_block(vx, TLW(im, GR_V2(h, 1)), h2[1]);
        }
        GSYNC();
        const f32x4 w2a = tl_bias(im, 18, q), w2b = tl_bias(im, 19, q);
#pragma unroll 2
        for (int t = 0; t < a.T; ++t) {
            f32x4 z = tl_zero();                                                        // z[t, l] = mean_h score[t, h] val[h, l]
#pragma unroll
            for (int h = 0; h < 5; ++h) z += val[h] * ws[h * 12 + t];
            z = prelu4(z * 0.2f, act4);
            f32x4 pa = prelu4(mma_block(tl_bias(im, 16, q), TLW(im, GR_P1(0)), z), act5);        // proj_2(PReLU5(proj_1(.)))
            f32x4 pb = prelu4(mma_block(tl_bias(im, 17, q), TLW(im, GR_P1(1)), z), act5);
            float o = w2a.x * pa.x;
            o += w2a.y * pa.y; o += w2a.z * pa.z; o += w2a.w * pa.w;
            o += w2b.x * pb.x; o += w2b.y * pb.y; o += w2b.z * pb.z; o += w2b.w * pb.w;
            o += __shfl_xor(o, 16);
            o += __shfl_xor(o, 32);
            if (ok && q == 0) a.out[(long long)n * a.T + t] = o + b_p2;
        }
        GSYNC();
    }
}

// LocalSliceLgCollapse (module.py:610-659), pick-sized: per pick a the K = 10 product nodes of its station whose theoretical
// arrival is nearest the pick time (time-pointer table A_edges[(ipick * l_dt + t_index) * K + k], :635-640), those within
// 2 eps of the pick time kept (:642-647), message PReLU1(fc1[s[e] || (tpick - tlatent[e]) / eps || phase]) (:657-659), 'mean'
// over the kept edges (:612), PReLU2(fc2 .) (:651). A wave owns 16 picks (fp32-MFMA tile layout of the tail kernels).
constexpr int LS_K = 10;
struct LsArgs {
    int n_picks, l_dt;
    long long n_edges;            // entries of the time-pointer table (indices are clamped into it)
    float t0, dt, eps;
    const float* s;               // [P, 30] association embedding (genie_assoc_fwd)
    const int32_t* A_edges;       // [n_sta * l_dt * K] product-node ids
    const float* tlatent; int tl_stride, tl_col;     // theoretical arrival of product node e: tlatent[e * tl_stride + tl_col]
    const float* tpick; const int32_t* ipick; const float* phase;
    const float* img;
    float* out;                   // [n_picks, 15]
};
__global__ __launch_bounds__(256) void k_lslc(LsArgs a) {
    __shared__ __attribute__((aligned(16))) float sm[GL_IMG_FLOATS];
    const TlImg im = tl_stage_image(sm, a.img, GL_GROUPS, GL_BIAS);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, q = lane >> 4;
    const float act1 = im.scal[0], act2 = im.scal[1];
    const int ntiles = (a.n_picks + 15) / 16;
    for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
        const int p = tile * 16 + j;
        const bool ok = p < a.n_picks;
        const int pc = ok ? p : a.n_picks - 1;
        const float tp = a.tpick[pc], ph = a.phase[pc];
        const int ti = (int)floorf((tp - a.t0) / a.dt);                                   // :635
        long long base = ((long long)a.ipick[pc] * a.l_dt + ti) * LS_K;
        base = base < 0 ? 0 : (base > a.n_edges - LS_K ? a.n_edges - LS_K : base);
        f32x4 acc[2] = {tl_zero(), tl_zero()};
        float cnt = 0.f;
#pragma unroll 2
        for (int k = 0; k < LS_K; ++k) {
            const int e = a.A_edges[base + k];
            const float rt = tp - a.tlatent[(long long)e * a.tl_stride + a.tl_col];
            const bool keep = ok && fabsf(rt) < 2.0f * a.eps;                             // :642-645
            const float* row = a.s + (long long)e * 30;
            const f32x4 xb0 = tl_load30(row, 0, q), xb1 = tl_load30(row, 1, q);
            const float xs = q == 0 ? rt / a.eps : (q == 1 ? ph : 0.f);                   // columns 30, 31 of fc1
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 m = mma_block(tl_bias(im, t, q), TLW(im, GL_FC1(t, 0)), xb0);
                m = mma_block(m, TLW(im, GL_FC1(t, 1)), xb1);
                m = MFMA16(TLW(im, GL_FC1(t, 2)).x, xs, m);
                m = prelu4(m, act1);
                if (keep) acc[t] += m;
            }
            cnt += keep ? 1.f : 0.f;
        }
        const float den = fmaxf(cnt, 1.f);
        f32x4 o = mma_block(tl_bias(im, 2, q), TLW(im, GL_FC2(0)), acc[0] / den);
        o = mma_block(o, TLW(im, GL_FC2(1)), acc[1] / den);
        o = prelu4(o, act2);
        if (ok) {
            float* og = a.out + (long long)p * 15 + 4 * q;
            og[0] = o.x; og[1] = o.y; og[2] = o.z;
            if (q < 3) og[3] = o.w;
        }
    }
}

// StationSourceAttentionMergedPhases (module.py:662-775, use_sparse = True, use_neighbor_assoc_edges = False), pick-sized. For
// every source i and pick a the reference attends over the picks b of a's station plus a null pick (:703-718), keeps the edges
// whose observed - theoretical arrival time (P or S) is inside 2 eps (:722-729) -- a property of (b, i) alone --, and runs three
// edge MLPs; queries and the relative-time features depend on (b, i) only, the context on (i, self_link, null_link), the values
// on (b, i, self_link, null_link). One workgroup per (source i, station u with picks):
//   A1  keep flags of the station's picks + the null pick, compacted in order into an LDS list;
//   A2  per kept b (16 per wave, fp32-MFMA tiles): query -> the three head scores against the context of a plain edge and of a
//       self edge (the null pick: of a null edge), values of a plain edge and of a self edge (null pick: of a null edge) -> LDS;
//   B   per target a of the station (16 per wave): segment softmax over the kept list (the entry b == a takes its self variant),
//       'add' aggregation, mean over heads, proj_2(PReLU4(proj_1(.))).
// k_arr_ctx prepares the three context vectors of every source. Needs at least one source with |stime| < 2 eps (then the
// reference's `edge_index[0].max()` (:762-763) is the null pick, as assumed here); the host checks it and otherwise keeps the
// PyTorch restatement. The softmax runs in its streaming form over chunks of AR_CAP picks (running maximum, denominator and
// weighted value sum per target; the sum is divided by (denominator + 1e-16) at the end instead of every weight being divided
// first: rounding-order difference only), so a station may hold any number of picks.
constexpr int AR_CAP = 192;       // picks of a station (null included) per LDS chunk
constexpr int AR_ENT = 104;       // floats per kept entry: scores plain [3], self [3], pad 2, values plain [3][16], self [3][16]
constexpr int AT_STAT = 64;       // training forward, per (source, pick): normalised head aggregates [3][16], running max [3], denominator [3]
struct ArArgs {
    int n_src, n_sta, n_arv, n_useg;
    float eps;
    const float* stime;           // [n_src]
    const float* trv_src;         // [n_src, n_sta, 2]
    const float* ctx;             // [n_src][4][48] (k_arr_ctx): plain, self, null, self + null edge; head h at 16h
    const float* arv_p; const float* arv_s;          // [n_arv, 15]
    const float* tpick; const float* phase;          // [n_arv]
    const int32_t* order;         // picks sorted by station (stable)
    const int32_t* seg_sta; const int32_t* seg_start; const int32_t* seg_len;     // [n_useg] stations with picks
    const float* img;
    float* out;                   // [n_src, n_arv, 2]
    int* e0max;                   // [1] `edge_index[0].max()` over the kept edges (module.py:762-763), by k_arr_e0max: the pick the
                                  // reference treats as "the null pick"; = n_arv (the real null pick) whenever some source has
                                  // |stime| < 2 eps, i.e. always in practice
    float* save;                  // training forward: [n_src * n_arv][AT_STAT] (else null)
};

// e0max = max over the kept (pick b, source i) pairs of b, the null pick (index n_arv) included (module.py:740-763). A pick's
// edges towards source i survive the 2-eps filter or not as a whole (the test involves only b and i), and every pick has at
// least its self pair, so "b has a kept edge towards i" = "its test passes".
__global__ __launch_bounds__(256) void k_arr_e0max(ArArgs a) {
    const int i = blockIdx.x / a.n_useg, ug = blockIdx.x - i * a.n_useg;
    const int u = a.seg_sta[ug], r0 = a.seg_start[ug], L = a.seg_len[ug];
    const float eps = a.eps, st = a.stime[i];
    const float tp_src = a.trv_src[((long long)i * a.n_sta + u) * 2 + 0] + st, ts_src = a.trv_src[((long long)i * a.n_sta + u) * 2 + 1] + st;
    int best = -1;
    if (threadIdx.x == 0 && ug == 0 && fabsf(st) < 2.f * eps) best = a.n_arv;
    for (int r = threadIdx.x; r < L; r += blockDim.x) {
        const int b = a.order[r0 + r];
        const float tp = a.tpick[b];
        if (fabsf(tp - tp_src) < 2.f * eps || fabsf(tp - ts_src) < 2.f * eps) best = max(best, b);
    }
    if (best >= 0) atomicMax(a.e0max, best);
}

__global__ __launch_bounds__(128) void k_arr_ctx(const float* __restrict__ raw, int o_c1w, int o_c1b, int o_c2w, int o_c2b, int o_a1,
                                                const float* __restrict__ src_embed, const float* __restrict__ stime, int n_src,
                                                float* __restrict__ ctx) {
    __shared__ float hid[4][32];
    const int i = blockIdx.x;
    if (i >= n_src) return;
    const float act1 = raw[o_a1];
    for (int idx = threadIdx.x; idx < 4 * 30; idx += blockDim.x) {
        const int v = idx / 30, c = idx - v * 30;
        float t = raw[o_c1b + c];
        for (int k = 0; k < 30; ++k) t += raw[o_c1w + c * 33 + k] * src_embed[(long long)i * 30 + k];
        t += raw[o_c1w + c * 33 + 30] * stime[i];
        if (v & 1) t += raw[o_c1w + c * 33 + 31];      // self_link
        if (v & 2) t += raw[o_c1w + c * 33 + 32];      // null_link
        hid[v][c] = prelu1(t, act1);
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 4 * 48; idx += blockDim.x) {
        const int v = idx / 48, r = idx - v * 48, h = r >> 4, l = r & 15;
        float t = 0.f;
        if (l < 15) {
            const int ch = 15 * h + l;
            t = raw[o_c2b + ch];
            for (int k = 0; k < 30; ++k) t += raw[o_c2w + ch * 30 + k] * hid[v][k];
        }
        ctx[((long long)i * 4 + v) * 48 + r] = t;
    }
}

__global__ __launch_bounds__(256) void k_arrivals(ArArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const TlImg im = tl_stage_image(sm, a.img, GA_GROUPS2, GA_BIAS2);
    float* ent = sm + GA2_IMG_FLOATS;                 // [AR_CAP][AR_ENT]
    int* kept = (int*)(ent + AR_CAP * AR_ENT);        // [AR_CAP] position r in the station's pick list (L = the null pick)
    int* kbi = kept + AR_CAP;                         // [AR_CAP] its pick index (n_arv = the null pick)
    float* cx = (float*)(kbi + AR_CAP);               // [4][48] context vectors of this source
    int* wcnt = (int*)(cx + 192);                     // [4] per-wave counts of the compaction, [4] = total
    const int i = blockIdx.x / a.n_useg, ug = blockIdx.x - i * a.n_useg;
    const int u = a.seg_sta[ug], r0 = a.seg_start[ug], L = a.seg_len[ug];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, q = lane >> 4;
    const float eps = a.eps, st = a.stime[i];
    const float tp_src = a.trv_src[((long long)i * a.n_sta + u) * 2 + 0] + st, ts_src = a.trv_src[((long long)i * a.n_sta + u) * 2 + 1] + st;
    const float rel_null = -eps - (-eps + st);        // null pick: atime = -eps, theoretical time = -eps (:722-725)
    for (int k = threadIdx.x; k < 192; k += blockDim.x) cx[k] = a.ctx[(long long)i * 192 + k];
    // the pick index the reference takes for the null pick (:762-765): n_arv unless NO source keeps the real null pick
    const int E = *a.e0max;
    __syncthreads();      // the weight image and cx are complete before any wave reads them (the slopes and proj_2 rows below!)
    const float act2 = im.scal[0], act3 = im.scal[1], act4 = im.scal[2];
    const float e2 = eps * eps, sq = sqrtf(15.f);
    const f32x4 w2[2][2] = {{tl_bias(im, 12, q), tl_bias(im, 13, q)}, {tl_bias(im, 14, q), tl_bias(im, 15, q)}};
    // Targets in blocks of 256 (4 tiles of 16 per wave, their softmax state in registers); the station's picks + the null pick
    // (r = 0 .. L) in chunks of AR_CAP: A1 / A2 fill the LDS list with the kept picks of the chunk, B folds them into the running
    // (max, denominator, weighted value sum) of every target (one chunk = the plain two-pass segment softmax).
    for (int tb0 = 0; tb0 < L; tb0 += 256) {
        float mx[4][3], den[4][3];
        f32x4 agg[4][3];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int h = 0; h < 3; ++h) { mx[tt][h] = -INFINITY; den[tt][h] = 0.f; agg[tt][h] = tl_zero(); }
        for (int cb = 0; cb <= L; cb += AR_CAP) {
            __syncthreads();                          // the previous chunk's list is no longer read
            // ---- A1: ordered compaction of the kept picks of the chunk
            {
                const int r = cb + (int)threadIdx.x;
                bool keep = false;
                if ((int)threadIdx.x < AR_CAP) {
                    if (r < L) {
                        const float tp = a.tpick[a.order[r0 + r]];
                        keep = fabsf(tp - tp_src) < 2.f * eps || fabsf(tp - ts_src) < 2.f * eps;
                    } else if (r == L) keep = fabsf(rel_null) < 2.f * eps;
                }
                const unsigned long long bal = __ballot(keep);
                if (lane == 0) wcnt[wave] = __popcll(bal);
                __syncthreads();
                int off = 0;
                for (int k = 0; k < wave; ++k) off += wcnt[k];
                if (keep) {
                    const int pos = off + __popcll(bal & ((1ull << lane) - 1ull));
                    kept[pos] = r;
                    kbi[pos] = r == L ? a.n_arv : a.order[r0 + r];
                }
                if (threadIdx.x == 0) wcnt[4] = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
                __syncthreads();
            }
            const int K = wcnt[4];
            // ---- A2: queries / scores / values of the kept picks
            for (int tile = wave; tile * 16 < K; tile += 4) {
                const int kk = tile * 16 + j;
                const bool ok = kk < K;
                const int r = kept[ok ? kk : K - 1];
                const bool nul = r == L;
                const int b = nul ? 0 : a.order[r0 + r];
                const bool nl = (nul ? a.n_arv : b) == E;          // null_link of this pick's edges (:765)
                const float tp = nul ? 0.f : a.tpick[b];
                const float rp = nul ? rel_null : tp - tp_src, rs = nul ? rel_null : tp - ts_src;
                const float ph = nul ? -1.f : a.phase[b];
                const float f6[6] = {expf(-0.5f * (rp * rp) / e2), (rp > 0.f) - (rp < 0.f) + 0.f, ph,
                                     expf(-0.5f * (rs * rs) / e2), (rs > 0.f) - (rs < 0.f) + 0.f, ph};
                const float x0 = q == 0 ? f6[0] : (q == 1 ? f6[1] : (q == 2 ? f6[2] : f6[3]));     // columns 30 + q
                const float x1 = q == 0 ? f6[4] : (q == 1 ? f6[5] : 0.f);                          // columns 34 + q
                const f32x4 xp = nul ? tl_zero() : tl_load15(a.arv_p + (long long)b * 15, q);
                const f32x4 xs = nul ? tl_zero() : tl_load15(a.arv_s + (long long)b * 15, q);
                f32x4 hq[2], hv[2], hw[2];   // hidden layers: query, values of an edge without / with self_link (null_link = nl in both)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f32x4 z = mma_block(tl_bias(im, t, q), TLW(im, GA_Q1(t, 0)), xp);
                    z = mma_block(z, TLW(im, GA_Q1(t, 1)), xs);
                    z = MFMA16(TLW(im, GA_Q1(t, 2)).x, x0, z);
                    z = MFMA16(TLW(im, GA_Q1(t, 2)).y, x1, z);
                    hq[t] = prelu4(z, act2);
                    f32x4 v = mma_block(tl_bias(im, 2 + t, q), TLW(im, GA_V1(t, 0)), xp);
                    v = mma_block(v, TLW(im, GA_V1(t, 1)), xs);
                    v = MFMA16(TLW(im, GA_V1(t, 2)).x, x0, v);
                    v = MFMA16(TLW(im, GA_V1(t, 2)).y, x1, v);
                    const float lnk = (q == 1 && nl) ? 1.f : 0.f;          // k-step [self_link, null_link]: lane q = 0 / 1 supplies it
                    const f32x4 vb = MFMA16(TLW(im, GA_V1(t, 3)).x, lnk, v);
                    const f32x4 w = MFMA16(TLW(im, GA_V1(t, 3)).x, q == 0 ? 1.f : lnk, v);
                    hv[t] = prelu4(vb, act3);
                    hw[t] = prelu4(w, act3);
                }
                float* eo = ent + (long long)(ok ? kk : K - 1) * AR_ENT;
#pragma unroll
                for (int h = 0; h < 3; ++h) {
                    f32x4 qh = mma_block(tl_bias(im, 4 + h, q), TLW(im, GA_Q2(h, 0)), hq[0]);
                    qh = mma_block(qh, TLW(im, GA_Q2(h, 1)), hq[1]);
                    // context of an edge without / with self_link (variants: bit 0 self_link, bit 1 null_link)
                    const f32x4 c0 = *(const f32x4*)(cx + (nl ? 96 : 0) + h * 16 + 4 * q), c1 = *(const f32x4*)(cx + (nl ? 144 : 48) + h * 16 + 4 * q);
                    const f32x4 p0 = qh * c0, p1 = qh * c1;
                    float s0 = ((p0.x + p0.y) + p0.z) + p0.w, s1 = ((p1.x + p1.y) + p1.z) + p1.w;
                    s0 += __shfl_xor(s0, 16); s0 += __shfl_xor(s0, 32);
                    s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
                    f32x4 vh = mma_block(tl_bias(im, 7 + h, q), TLW(im, GA_V2(h, 0)), hv[0]);
                    vh = mma_block(vh, TLW(im, GA_V2(h, 1)), hv[1]);
                    f32x4 wh = mma_block(tl_bias(im, 7 + h, q), TLW(im, GA_V2(h, 0)), hw[0]);
                    wh = mma_block(wh, TLW(im, GA_V2(h, 1)), hw[1]);
                    if (ok) {
                        if (q == 0) { eo[h] = s0 / sq; eo[3 + h] = s1 / sq; }
                        *(f32x4*)(eo + 8 + h * 16 + 4 * q) = vh;
                        *(f32x4*)(eo + 56 + h * 16 + 4 * q) = wh;
                    }
                }
            }
            __syncthreads();
            // ---- B: fold the chunk into the targets' softmax state
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const int r = tb0 + (tt * 4 + wave) * 16 + j;
                if (tb0 + (tt * 4 + wave) * 16 >= L || K == 0) continue;          // (uniform per wave)
                // self_link = (e0 == e1 mod e0max), e1 = a + i n_arv (:764): the target pick itself when e0max = n_arv
                const int tsel = (r < L && E > 0) ? (int)(((long long)a.order[r0 + r] + (long long)i * a.n_arv) % E) : -1;
                float cm[3] = {mx[tt][0], mx[tt][1], mx[tt][2]};
                for (int k = 0; k < K; ++k) {
                    const float* e = ent + k * AR_ENT + (kbi[k] == tsel ? 3 : 0);
#pragma unroll
                    for (int h = 0; h < 3; ++h) cm[h] = fmaxf(cm[h], e[h]);
                }
#pragma unroll
                for (int h = 0; h < 3; ++h) {
                    const float sc = mx[tt][h] == -INFINITY ? 0.f : expf(mx[tt][h] - cm[h]);
                    den[tt][h] *= sc; agg[tt][h] *= sc; mx[tt][h] = cm[h];
                }
                for (int k = 0; k < K; ++k) {
                    const bool self = kbi[k] == tsel;
                    const float* e = ent + k * AR_ENT;
#pragma unroll
                    for (int h = 0; h < 3; ++h) {
                        const float ex = expf(e[(self ? 3 : 0) + h] - cm[h]);
                        den[tt][h] += ex;
                        agg[tt][h] += *(const f32x4*)(e + (self ? 56 : 8) + h * 16 + 4 * q) * ex;
                    }
                }
            }
        }
        // ---- every pick of the block: normalise, mean over heads (:760), proj_2(PReLU4(proj_1(.)))
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int rb = tb0 + (tt * 4 + wave) * 16;
            if (rb >= L) continue;
            const int r = rb + j;
            const bool ok = r < L;
            const f32x4 z = ((agg[tt][0] / (den[tt][0] + 1e-16f) + agg[tt][1] / (den[tt][1] + 1e-16f)) + agg[tt][2] / (den[tt][2] + 1e-16f)) / 3.f;
            if (a.save && ok) {
                float* sv = a.save + ((long long)i * a.n_arv + a.order[r0 + r]) * AT_STAT;
#pragma unroll
                for (int h = 0; h < 3; ++h) {
                    *(f32x4*)(sv + 16 * h + 4 * q) = agg[tt][h] / (den[tt][h] + 1e-16f);
                    if (q == 0) { sv[48 + h] = mx[tt][h]; sv[51 + h] = den[tt][h]; }
                }
            }
            f32x4 pa = prelu4(mma_block(tl_bias(im, 10, q), TLW(im, GA_P1(0)), z), act4);
            f32x4 pb = prelu4(mma_block(tl_bias(im, 11, q), TLW(im, GA_P1(1)), z), act4);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                float o = w2[m][0].x * pa.x;
                o += w2[m][0].y * pa.y; o += w2[m][0].z * pa.z; o += w2[m][0].w * pa.w;
                o += w2[m][1].x * pb.x; o += w2[m][1].y * pb.y; o += w2[m][1].z * pb.z; o += w2[m][1].w * pb.w;
                o += __shfl_xor(o, 16);
                o += __shfl_xor(o, 32);
                if (ok && q == 0) a.out[((long long)i * a.n_arv + a.order[r0 + r]) * 2 + m] = o + im.scal[3 + m];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Pick -> Slice/Mask embedding on device: `extract_input_from_data`,
// /root/reference/Code/process_utils.py:460-642 (use_sign_input = False). Step 1: per-station Gaussian-kernel time
// series of the P- and S-labelled picks by scatter-max (:499-569; max is order independent -> deterministic atomics).
// Step 2: every product node reads the series of its station at the theoretical P / S arrival index (:599-629).
// ------------------------------------------------------------------------------------------------
struct EmbArgs {
    const double* pick_t; const int32_t* pick_sta; const int32_t* pick_phase;
    int n_picks, n_time, n_extra, S;
    double t0, tref0, dt, sigma;
    float* emb;            // [2][S][n_time]: P-labelled series, then S-labelled
    const float* trv;      // [rows, 2] theoretical P / S travel time of every product node
    long long rows;
    float* slice; float* mask;
    unsigned* xs;          // optional: the split rows of k_stage1_h2, written together with Slice / Mask
    const int32_t* sta_inv; // station processing order of the split rows (caller's station -> internal), or null
    float* mm;              // with sta_inv: max of the Mask row, in processing order
};

__global__ void k_embed_scatter(EmbArgs a) {
    const int per = 2 * a.n_extra + 1;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)a.n_picks * per) return;
    const int pk = (int)(i / per), off = (int)(i - (long long)pk * per) - a.n_extra;
    const int sta = a.pick_sta[pk];
    const int ph = a.pick_phase[pk];
    if (sta < 0 || sta >= a.S || (ph != 0 && ph != 1)) return;
    const double t = a.pick_t[pk];
    const int idx = (int)((t - a.tref0) / a.dt) + off;                 // :514-515, :534
    if (idx < 0 || idx >= a.n_time) return;                            // :537
    const double tv = t - (a.tref0 + (double)idx * a.dt);              // abs_time_ref[idx] = arange(...)[idx]
    const float val = (float)exp(-0.5 * tv * tv / (a.sigma * a.sigma));   // :545, cast at torch.Tensor(vals) :563
    atomicMax((int*)(a.emb + ((long long)ph * a.S + sta) * a.n_time + idx), __float_as_int(val));   // val >= 0
}

__global__ void k_embed_edges(EmbArgs a) {   // overflow guard: first / last sample of every series is zero (:565-568)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * a.S) return;
    a.emb[(long long)i * a.n_time] = 0.f;
    a.emb[(long long)i * a.n_time + a.n_time - 1] = 0.f;
}

__global__ void k_embed_gather(EmbArgs a) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.rows) return;
    const int sta = (int)(p % a.S);
    const float2 tt = *(const float2*)(a.trv + p * 2);
    int ip = (int)((((double)tt.x + a.t0) - a.tref0) / a.dt);           // :605 (float64 arithmetic, truncation)
    int is = (int)((((double)tt.y + a.t0) - a.tref0) / a.dt);
    ip = min(max(ip, 0), a.n_time - 1);
    is = min(max(is, 0), a.n_time - 1);
    const float* ep = a.emb + (long long)sta * a.n_time;
    const float* es = a.emb + ((long long)a.S + sta) * a.n_time;
    f32x4 sl;
    sl.x = fmaxf(ep[ip], es[ip]);                                        // any-phase series = max(P, S)  :569, :612
    sl.y = fmaxf(ep[is], es[is]);                                        // :613
    sl.z = ep[ip];                                                       // :614
    sl.w = es[is];                                                       // :615
    f32x4 mk;
    mk.x = fabsf(sl.x) > 0.01f ? 1.f : 0.f; mk.y = fabsf(sl.y) > 0.01f ? 1.f : 0.f;      // :629
    mk.z = fabsf(sl.z) > 0.01f ? 1.f : 0.f; mk.w = fabsf(sl.w) > 0.01f ? 1.f : 0.f;
    *(f32x4*)(a.slice + p * 4) = sl;
    *(f32x4*)(a.mask + p * 4) = mk;
    if (a.xs != nullptr) {            // same rows as k_split_rows would produce from (sl, mk)
        const float v[8] = {sl.x, sl.y, sl.z, sl.w, mk.x, mk.y, mk.z, mk.w};
        const long long px = a.sta_inv != nullptr ? p - sta + a.sta_inv[sta] : p;
        if (a.sta_inv != nullptr) a.mm[px] = fmaxf(fmaxf(mk.x, mk.y), fmaxf(mk.z, mk.w));
        store_split_row(a.xs, a.rows, px, v);
    }
}

// de-pad rows of a workspace tensor for parity tests
// use_absolute_pos (config.yaml:92; module.py:1007): Slice gets locs[sta] / (3 scale_rel) and x_grid[src] / (3 scale_rel) appended
__global__ void k_abs_table(const float* __restrict__ pos, int n, float inv, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * 4) return;
    const int r = i >> 2, k = i & 3;
    out[i] = k < 3 ? pos[r * 3 + k] * inv : 0.f;
}

// the two fp16 pieces of every row of an [n][4] scaled-position table, [n][2] x 8 B; `perm` (or null): row i = table row perm[i]
__global__ void k_abs_pieces(const float* __restrict__ tab, const int32_t* __restrict__ perm, int n, unsigned* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const f32x4 v = *(const f32x4*)(tab + (size_t)(perm ? perm[i] : i) * 4);
    const unsigned a0 = cvt_pk_f16(v.x, v.y), b0 = cvt_pk_f16(v.z, 0.f);
    const unsigned a1 = cvt_pk_f16(sub_f16_lo(v.x, a0), sub_f16_hi(v.y, a0)), b1 = cvt_pk_f16(sub_f16_lo(v.z, b0), 0.f);
    *(u32x4*)(out + (size_t)i * 4) = u32x4{a0, b0, a1, b1};
}

// DataAggregationEdges (module.py:102-174, forward :1059-1072): every message carries phi(pos_j - pos_i) (3) and phi(|pos_j - pos_i|),
// phi(d) = sign(d) exp(-d^2 / (2 scale_rel^2)); after mean aggregation that is a STATIC 4-vector per node of a base graph.
__global__ void k_edge_feat(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, int n,
                            const float* __restrict__ pos, float scale_rel, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int eb = rowptr[i], ee = rowptr[i + 1];
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const float inv = 1.f / (scale_rel * scale_rel);
    for (int e = eb; e < ee; ++e) {
        const int j = col[e];
        float d[4];
        d[0] = pos[j * 3] - pos[i * 3]; d[1] = pos[j * 3 + 1] - pos[i * 3 + 1]; d[2] = pos[j * 3 + 2] - pos[i * 3 + 2];
        d[3] = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float sg = d[k] > 0.f ? 1.f : (d[k] < 0.f ? -1.f : 0.f);
            acc[k] += sg * expf(-0.5f * d[k] * d[k] * inv);
        }
    }
    const float w = ee > eb ? 1.f / (float)(ee - eb) : 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) out[i * 4 + k] = acc[k] * w;
}
// ... and its Linear a per-node additive term: row n = {W1pos (30x4) m_n, 0, 0, W2pos (15x4) m_n, 0}
__global__ void k_edge_bias(const float* __restrict__ raw, int off1, int off2, const float* __restrict__ mpos, int n,
                            float* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * 48) return;
    const int i = idx / 48, ch = idx - i * 48;
    const float* m = mpos + i * 4;
    float v = 0.f;
    if (ch < 30) { const float* wr = raw + off1 + ch * 4; v = wr[0] * m[0] + wr[1] * m[1] + wr[2] * m[2] + wr[3] * m[3]; }
    else if (ch >= 32 && ch < 47) { const float* wr = raw + off2 + (ch - 32) * 4; v = wr[0] * m[0] + wr[1] * m[1] + wr[2] * m[2] + wr[3] * m[3]; }
    out[idx] = v;
}

// Neighbour means on the implicit product graph for arbitrary row widths (association heads, module.py:389-403):
//   out_sta[(g,s)] = mean_k x_sta[(g, sta_nbr_k(s))],   out_src[(g,s)] = mean_k x_src[(src_nbr_k(g), s)]
// rows of CL * VW floats; CL lanes per node, every lane keeps up to 8 row chunks in flight; sums in edge order.
// With per-edge weights (w_sta / w_src non-null) the same kernel is the ADJOINT of the mean on the reversed graphs:
//   dx[j] = sum_{i : j in N(i)} g[i] / deg(i)   (genie_nbr_mean_bwd; edge lists = out-edges of j, weights 1 / in-degree of i)
template <int CL, int VW>       // CL lanes per row, VW floats per lane: rows of CL * VW floats (16 / 32 padded, or 30 unpadded)
__global__ __launch_bounds__(256) void k_nbr_mean(int S, int G, const int32_t* __restrict__ sta_rowptr, const int32_t* __restrict__ sta_col,
                                                  const int32_t* __restrict__ src_rowptr, const int32_t* __restrict__ src_col,
                                                  const float* __restrict__ x_sta, const float* __restrict__ x_src,
                                                  float* __restrict__ out_sta, float* __restrict__ out_src,
                                                  const float* __restrict__ w_sta = nullptr, const float* __restrict__ w_src = nullptr) {
    typedef float V __attribute__((ext_vector_type(VW)));
    constexpr int NPB_ = 256 / CL, RF = CL * VW;
    if ((int)threadIdx.x >= NPB_ * CL) return;
    const int cl = threadIdx.x % CL;
    const long long P = (long long)S * G;
    for (long long p = (long long)blockIdx.x * NPB_ + threadIdx.x / CL; p < P; p += (long long)gridDim.x * NPB_) {
        const int g = (int)(p / S), s = (int)(p - (long long)g * S);
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            const float* x = which == 0 ? x_sta : x_src;
            float* out = which == 0 ? out_sta : out_src;
            if (x == nullptr) continue;
            const int32_t* col = which == 0 ? sta_col : src_col;
            const float* ew = which == 0 ? w_sta : w_src;
            const int eb = which == 0 ? sta_rowptr[s] : src_rowptr[g], ee = which == 0 ? sta_rowptr[s + 1] : src_rowptr[g + 1];
            V acc = 0.f;
            for (int e0 = eb; e0 < ee; e0 += 8) {
                V v[8];
                float wk[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int e = min(e0 + k, ee - 1);
                    const int j = col[e];
                    const long long row = which == 0 ? (long long)g * S + j : (long long)j * S + s;
                    v[k] = *(const V*)(x + row * RF + VW * cl);
                    wk[k] = ew ? ew[e] : 1.f;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (e0 + k < ee) acc += v[k] * wk[k];
            }
            const float w = ew ? 1.f : (ee > eb ? 1.f / (float)(ee - eb) : 0.f);
            *(V*)(out + p * RF + VW * cl) = acc * w;
        }
    }
}

// PReLU backward for the training path (one slope per call): dx = dy * (x >= 0 ? 1 : a), da = sum_{x < 0} dy * x. PyTorch's
// own backward materialises a full-size slope gradient and reduces it in a second pass (0.9 ms per [2M, 30] tensor); this
// is one pass plus a fixed-order two-level sum (deterministic).
constexpr int PRELU_BLOCKS = 2048;
__global__ __launch_bounds__(256) void k_prelu_bwd(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ slope,
                                                   long long n, float* __restrict__ dx, float* __restrict__ partial) {
    const float a = slope[0];
    float acc = 0.f;
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const f32x4 xv = ((const f32x4*)x)[i], gv = ((const f32x4*)dy)[i];
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool neg = xv[k] < 0.f;
            o[k] = neg ? gv[k] * a : gv[k];
            acc += neg ? gv[k] * xv[k] : 0.f;
        }
        ((f32x4*)dx)[i] = o;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {      // tail elements
        const long long i = (n4 << 2) + threadIdx.x;
        const bool neg = x[i] < 0.f;
        dx[i] = neg ? dy[i] * a : dy[i];
        acc += neg ? dy[i] * x[i] : 0.f;
    }
    __shared__ float red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void k_prelu_bwd_sum(const float* __restrict__ partial, int nb, float* __restrict__ da) {
    __shared__ float red[256];
    float acc = 0.f;
    for (int i = threadIdx.x; i < nb; i += 256) acc += partial[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) da[0] = red[0];
}

// Weight and bias gradients of a per-node Linear over N rows (training path): dW[m][k] = sum_n dy[n][m] x[n][k],
// db[m] = sum_n dy[n][m], M <= 32, K <= 64 KC. The library GEMM for this shape (a [M, N] x [N, K] product with N = 2M rows) runs
// at 1-3 ms plus a separate 0.25 ms bias reduction; this reads x and dy once. Wave w owns outputs m in [8w, 8w+8), lane l the
// columns k = l + 64 c; dy rows are staged through LDS and read back as wave-uniform broadcasts. Partials per workgroup are
// summed by k_linear_bwd_sum in a fixed order.
constexpr int LBW_ROWS = 32, LBW_BLOCKS = 1024;
template <int KC>
__global__ __launch_bounds__(256) void k_linear_bwd_w(const float* __restrict__ x, const float* __restrict__ dy, long long N, int K, int M,
                                                      int ldy, float* __restrict__ partial) {
    __shared__ float sdy[LBW_ROWS][32];
    __shared__ float sb[8][32];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sm = threadIdx.x & 31, sr = threadIdx.x >> 5;          // staging role: column sm of rows sr + 8 q
    const int smc = min(sm, M - 1);
    int kc[KC];                                                       // lanes beyond K re-read column K-1; their sums are dropped
#pragma unroll
    for (int c = 0; c < KC; ++c) kc[c] = min(lane + 64 * c, K - 1);
    float acc[KC][8];
    float accb = 0.f;
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[c][j] = 0.f;
    const long long ntile = (N + LBW_ROWS - 1) / LBW_ROWS;
    for (long long t = blockIdx.x; t < ntile; t += gridDim.x) {
        const long long n0 = t * LBW_ROWS;
        const int nr = (int)min((long long)LBW_ROWS, N - n0);        // rows beyond N: dy staged as 0, x re-reads the last row
        const float* __restrict__ xt = x + n0 * K;
        const float* __restrict__ dt = dy + n0 * ldy;            // M <= 32 columns of rows that are ldy floats apart
        float st[LBW_ROWS / 8];
#pragma unroll
        for (int q = 0; q < LBW_ROWS / 8; ++q) {
            const int r = sr + 8 * q;
            const float v = dt[min(r, nr - 1) * ldy + smc];
            st[q] = (sm < M && r < nr) ? v : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < LBW_ROWS / 8; ++q) {
            sdy[sr + 8 * q][sm] = st[q];
            accb += st[q];
        }
        __syncthreads();
#pragma unroll
        for (int rb = 0; rb < LBW_ROWS; rb += 8) {
            float xv[8][KC];
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int c = 0; c < KC; ++c) xv[r][c] = xt[min(rb + r, nr - 1) * K + kc[c]];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const f32x4 d0 = *(const f32x4*)&sdy[rb + r][8 * w], d1 = *(const f32x4*)&sdy[rb + r][8 * w + 4];
#pragma unroll
                for (int c = 0; c < KC; ++c) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[c][j] += d0[j] * xv[r][c];
                        acc[c][4 + j] += d1[j] * xv[r][c];
                    }
                }
            }
        }
    }
    float* out = partial + (size_t)blockIdx.x * (32 * 64 * KC + 32);
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) out[(8 * w + j) * (64 * KC) + 64 * c + lane] = acc[c][j];
    sb[sr][sm] = accb;
    __syncthreads();
    if (threadIdx.x < 32) {
        float v = sb[0][sm];
#pragma unroll
        for (int q = 1; q < 8; ++q) v += sb[q][sm];
        out[32 * 64 * KC + sm] = v;
    }
}
__global__ __launch_bounds__(256) void k_linear_bwd_sum(const float* __restrict__ partial, int nb, int KC, int K, int M,
                                                        float* __restrict__ dW, float* __restrict__ db) {
    // 32 outputs x 8 slices of the workgroup partials per block; slices combined in a fixed order
    __shared__ float red[8][32];
    const int per = 32 * 64 * KC + 32;
    const int el = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + el;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (e < per) {
        const int per_sl = (nb + 7) / 8, b0 = sl * per_sl, b1 = min(nb, b0 + per_sl);
        int b = b0;
        for (; b + 4 <= b1; b += 4) {
            a0 += partial[(size_t)b * per + e];
            a1 += partial[(size_t)(b + 1) * per + e];
            a2 += partial[(size_t)(b + 2) * per + e];
            a3 += partial[(size_t)(b + 3) * per + e];
        }
        for (; b < b1; ++b) a0 += partial[(size_t)b * per + e];
    }
    red[sl][el] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (sl != 0 || e >= per) return;
    float v = red[0][el];
#pragma unroll
    for (int q = 1; q < 8; ++q) v += red[q][el];
    if (e < 32 * 64 * KC) {
        const int m = e / (64 * KC), k = e % (64 * KC);
        if (m < M && k < K) dW[m * K + k] = v;
    } else if (db && e - 32 * 64 * KC < M) db[e - 32 * 64 * KC] = v;
}

// XCC (XCD) id and CU id of the CU a workgroup runs on (genie_where_am_i): workgroup b of a launch lands on XCD b % 8
__global__ void k_where_am_i(int* __restrict__ out) {
    int xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(20, 0, 4)" : "=s"(xcc));       // HW_REG_XCC_ID
    asm volatile("s_getreg_b32 %0, hwreg(4, 0, 32)" : "=s"(hwid));      // HW_REG_HW_ID
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hwid; }
}

// ------------------------------------------------------------------------------------------------
// Exact k-nearest-neighbour search on the device: the `knn(x_context / 1000, x_query / 1000, k)` calls of the
// reference -- SpatialAttention's query edges (module.py:282; a NEW 112 000-point query set per candidate in the refine pass,
// process_continuous_days.py:926-980) and the base graphs of the product graph (`knn(x/1000, x/1000, k + 1)` +
// `remove_self_loops`, process_utils.py:718-719). Brute force in fp64 on the fp32 coordinates themselves (the
// common 1 / 1000 scale does not change the order; coordinate differences of fp32 values are exact in fp64) (3-D, n_context
// is 10^4..10^5: 10^9 pair distances = a millisecond): one wave per query, lane l scans candidates l, l + 64, ... keeping its K
// best in registers (sorted, ties by smaller index), then K rounds of a wave-wide lexicographic (distance, index) minimum pop
// the global K best in order. exclude_self: skip candidate == query id (query set = context set).
// ------------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(256) void k_knn(const float* __restrict__ xc, int nc, const float* __restrict__ xq, int nq, int k,
                                             int exclude_self, int32_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= nq) return;
    const double q0 = (double)xq[qi * 3 + 0], q1 = (double)xq[qi * 3 + 1], q2 = (double)xq[qi * 3 + 2];
    double bd[K];
    int bi[K];
#pragma unroll
    for (int t = 0; t < K; ++t) { bd[t] = __builtin_inf(); bi[t] = 0x7fffffff; }
    for (int c = lane; c < nc; c += 64) {
        if (exclude_self && c == qi) continue;
        const double d0 = q0 - (double)xc[c * 3 + 0], d1 = q1 - (double)xc[c * 3 + 1], d2 = q2 - (double)xc[c * 3 + 2];   // exact
        double d = __dadd_rn(__dadd_rn(__dmul_rn(d0, d0), __dmul_rn(d1, d1)), __dmul_rn(d2, d2));
        int id = c;
        if (d < bd[K - 1]) {          // candidates arrive in increasing index order: a tie never displaces an earlier entry
#pragma unroll
            for (int t = 0; t < K; ++t) {
                if (d < bd[t] || (d == bd[t] && id < bi[t])) {       // lexicographic (distance, index): a displaced entry that ties with
                                                                      // the next slot goes in front of it (it has the smaller index)
                    const double td = bd[t]; const int ti = bi[t];
                    bd[t] = d; bi[t] = id; d = td; id = ti;
                }
            }
        }
    }
    for (int r = 0; r < k; ++r) {
        double md = bd[0];
        int mi = bi[0];
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const double od = __shfl_xor(md, s);
            const int oi = __shfl_xor(mi, s);
            if (od < md || (od == md && oi < mi)) { md = od; mi = oi; }
        }
        if (bi[0] == mi && bd[0] == md) {        // the owner pops its head (indices are unique across lanes)
#pragma unroll
            for (int t = 0; t + 1 < K; ++t) { bd[t] = bd[t + 1]; bi[t] = bi[t + 1]; }
            bd[K - 1] = __builtin_inf(); bi[K - 1] = 0x7fffffff;
        }
        if (lane == 0) out[(long long)qi * k + r] = mi == 0x7fffffff ? -1 : mi;
    }
}

// ------------------------------------------------------------------------------------------------
// Downstream reduction of the apply loop on the device: the stacked query output Out_2 [rows = queries,
// cols = time steps] never leaves the GPU whole. MODE 0: entries above a threshold, `np.where(Out_2 > 0.01)`
// (process_continuous_days.py:812-813). MODE 1: the local maxima of every row that reach `height`, i.e. the first two steps
// of `scipy.signal.find_peaks(Out[i, :], height = thresh, ...)` (:846; scipy's `_local_maxima_1d`: a sample or the midpoint
// of a flat run that is strictly higher than both neighbours, never the first or last sample; then `x >= height`).
// One workgroup per row, chunks of 256 columns, selected entries written in column order at `offsets[row]` + rank
// (two passes: COUNT fills counts[row], the caller scans them; the second pass fills) -> row-major order like np.where.
// ------------------------------------------------------------------------------------------------
template <int MODE, bool COUNT>
__global__ __launch_bounds__(256) void k_row_select(const float* __restrict__ x, long long cols, float thr, int32_t* __restrict__ counts,
                                                    const long long* __restrict__ offsets, int32_t* __restrict__ out_row,
                                                    int32_t* __restrict__ out_col, float* __restrict__ out_val) {
    __shared__ int wsum[4];
    const long long row = blockIdx.x;
    const float* xr = x + row * cols;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long base = COUNT ? 0 : offsets[row];
    int total = 0;
    for (long long c0 = 0; c0 < cols; c0 += 256) {
        const long long i = c0 + threadIdx.x;
        bool sel = false;
        long long col = i;
        float v = 0.f;
        if (i < cols) {
            v = xr[i];
            if (MODE == 0) {
                sel = v > thr;
            } else if (i >= 1 && i + 1 < cols && v >= thr && xr[i - 1] < v) {      // rising edge into a candidate (flat) top
                long long e = i + 1;
                while (e < cols - 1 && xr[e] == v) ++e;
                if (xr[e] < v) { sel = true; col = (i + e - 1) / 2; }
            }
        }
        const unsigned long long b = __ballot(sel);
        const int rank = __popcll(b & ((1ull << lane) - 1ull)), wtot = __popcll(b);
        if (lane == 0) wsum[wave] = wtot;
        __syncthreads();
        int before = 0, all = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { if (k < wave) before += wsum[k]; all += wsum[k]; }
        if (!COUNT && sel) {
            const long long o = base + total + before + rank;
            out_row[o] = (int32_t)row; out_col[o] = (int32_t)col; out_val[o] = v;
        }
        total += all;
        __syncthreads();
    }
    if (COUNT && threadIdx.x == 0) counts[row] = total;
}

// Product-level CSRs of the IRREGULAR product graph of `use_subgraph` (process_utils.py:744-849) on the device. The product
// nodes are the (station, source) pairs sorted by (source, station): node n = (pair_sta[n], pair_src[n]), source node g owns
// nodes [seg[g], seg[g + 1]). In-edges of node n (the reference's `subgraph(...)` calls, :824-839):
//   station graph: m -> n for every base edge j -> pair_sta[n] whose pair (j, pair_src[n]) exists, in base edge order;
//   source graph:  m -> n for every base edge g' -> pair_src[n] whose pair (pair_sta[n], g') exists, in base edge order.
// One thread per product node; a pair is looked up by binary search in the station list of its source node. FILL = false
// counts the in-edges, FILL = true writes them behind the node's row pointer.
template <bool FILL>
__global__ __launch_bounds__(256) void k_subgraph_csr(const int32_t* __restrict__ pair_sta, const int32_t* __restrict__ pair_src,
                                                     long long N, const int32_t* __restrict__ seg,
                                                     const int32_t* __restrict__ sta_rowptr, const int32_t* __restrict__ sta_col,
                                                     const int32_t* __restrict__ src_rowptr, const int32_t* __restrict__ src_col,
                                                     int32_t* __restrict__ cnt_sta, int32_t* __restrict__ cnt_src,
                                                     const int32_t* __restrict__ p_sta_rowptr, const int32_t* __restrict__ p_src_rowptr,
                                                     int32_t* __restrict__ p_sta_col, int32_t* __restrict__ p_src_col) {
    const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int s = pair_sta[n], g = pair_src[n];
    auto find = [&](int sta, int src) -> int {          // product node of the pair (sta, src), or -1
        int lo = seg[src], hi = seg[src + 1];
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (pair_sta[mid] < sta) lo = mid + 1; else hi = mid;
        }
        return (lo < seg[src + 1] && pair_sta[lo] == sta) ? lo : -1;
    };
    int c1 = 0, c2 = 0;
    int32_t* o1 = FILL ? p_sta_col + p_sta_rowptr[n] : nullptr;
    int32_t* o2 = FILL ? p_src_col + p_src_rowptr[n] : nullptr;
    for (int e = sta_rowptr[s]; e < sta_rowptr[s + 1]; ++e) {
        const int m = find(sta_col[e], g);
        if (m >= 0) { if (FILL) o1[c1] = m; ++c1; }
    }
    for (int e = src_rowptr[g]; e < src_rowptr[g + 1]; ++e) {
        const int m = find(s, src_col[e]);
        if (m >= 0) { if (FILL) o2[c2] = m; ++c2; }
    }
    if (!FILL) { cnt_sta[n] = c1; cnt_src[n] = c2; }
}

__global__ void k_export(const float* __restrict__ src, long long rows, int pitch, int ncol, float* __restrict__ dst,
                         const int32_t* __restrict__ sta_user, int S) {
    // padded rows are [15 valid, 1 pad] blocks (c has two of them, wu / wv one); rows in station processing order -> caller's order
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * ncol) return;
    const long long r = idx / ncol;
    const int cc = (int)(idx % ncol);
    const int off = ncol == 15 ? cc : (cc / 15) * 16 + cc % 15;   // c = [c1 15,0 | c2 15,0]
    long long ru = r;
    if (sta_user != nullptr) {
        const long long g = r / S;
        ru = g * S + sta_user[(int)(r - g * S)];
    }
    dst[ru * ncol + cc] = src[r * pitch + off];
}
// rows [G][S][width] from station processing order to the caller's order (debug outputs) or, with `inv`, the other way (tables)
__global__ void k_permute_sta_rows(const float* __restrict__ src, long long rows, int width, const int32_t* __restrict__ map, int S,
                                   float* __restrict__ dst) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * width) return;
    const long long r = idx / width;
    const int cc = (int)(idx - r * width);
    const long long g = r / S;
    dst[(g * S + map[(int)(r - g * S)]) * width + cc] = src[idx];
}

#include "train_tail_kernels.hpp"
#include "train_assoc_kernels.hpp"
#include "train_arrival_kernels.hpp"

}  // namespace

// ================================================================================================
// context + C ABI
// ================================================================================================
struct genie_ctx {
    int S, G, G_ext, T;
    float scale_rel, scale_t;
    long long P, P_ext, E_src;
    int32_t *sta_rowptr, *sta_col, *src_rowptr, *src_col, *order, *outdeg;
    float* raw;
    bool dirty;
    StagePlan plan[NPLAN];     // 0, 1: DataAggregation stage 1 / 2; 2, 3: association stages A / B; 4, 5, 6: backward passes 2', 1', 0';
                               // PL_RO0 ...: the MFMA kernels of the G- / Q-sized tail
    void* d_packplans; int pack_blocks;     // k_pack_all's plan table (PackPlan[NPLAN]) and its grid
    StepDesc* d_steps[NPLAN];
    BiasDesc* d_bias[NPLAN];
    int32_t* d_scal[NPLAN];
    float* packed[NPLAN];
    AccDesc* d_acc[NTM]; VecDesc* d_vec[NTM]; int32_t* d_sc[NTM];   // gradient maps of the backward passes (k_train_reduce)
    int n_acc[NTM], n_vec[NTM], n_sc[NTM];
    float* train_save;         // ... and where those kernels keep the pre-activations (DaArgs.save)
    int force_generic;         // set for the duration of a training call: the generic fp32 stage kernels (caller's station order,
                               // pre-activations saved) run whatever the context would normally select
    float* as_pg;              // [G][AS_PG] per-source-node terms of the association stages (allocated on first use)
    float* as_ps;              // [S][AS_PS] per-station terms of the two model variants (allocated on first use)
    int32_t* d_h2tbl;          // k_pack_h2 source table
    // reversed base graphs (out-edges, weights 1 / in-degree of the target): built on the first genie_nbr_mean_bwd
    int32_t *r_sta_rowptr, *r_sta_col, *r_src_rowptr, *r_src_col;
    float *r_sta_w, *r_src_w;
    const float *xs_slice, *xs_mask;   // genie_embed_window_split: the (Slice, Mask) buffers whose split rows already sit in the workspace (one-shot)
    const void* xs_ws;
    int xs_mm_copy;            // ... and the copy (slot % GENIE_NBIG at embed time) its message-mask row `mm` was written to
    float *abs_sta, *abs_src;  // use_absolute_pos: [S][4], [G_ext][4] scaled positions; null = off
    unsigned *abs_ts, *abs_tg; // ... their fp16 pieces for k_stage1_h2 (stations in processing order), rebuilt when abs_dirty
    bool abs_dirty;
    // irregular product graph (`use_subgraph`): product-level CSRs, row range of every source node
    bool pcsr;
    bool pcsr_h2;              // ... with at most 8 / 15 neighbours per product node: k_stage1_h2<.., PCSR> applies
    int32_t *p_sta_rowptr, *p_sta_col, *p_src_rowptr, *p_src_col, *seg_rowptr;
    float *mpos_sta, *mpos_src, *ebias_sta, *ebias_src;   // DataAggregationEdges: mean edge features [n,4] and their Linear [n,48]
    bool has_edges;
    // station processing order (genie_set_station_order): internal -> caller's station, its inverse, the station graph in
    // internal labels, the per-station edge terms in internal order; null = the caller's order
    int32_t *sta_perm, *sta_inv, *sta_rowptr_p, *sta_col_p;
    float* ebias_sta_p;
    float* ea_int; const float* ea_user;   // genie_set_static_edge_attr: processing-order copy of the caller's static edge_attr
    float* ea_tmp;             // ... of an edge_attr that is not the registered one (permuted per call)
    int32_t* src_tab;          // [G][16] processing-order table of k_stage1_h2 (null unless kp_uni == 15)
    float* packed_h2;          // f16x2 weight image of k_stage1_h2
    int num_cu;
    int seg, bpc1, bpc2;       // sweep segments (env GENIE_SEG), workgroups per CU of the generic stage kernels
    int ks_uni, kp_uni;        // uniform in-degree of the station / source graph, -1 when ragged
    int use_fast;              // the reference's kNN graphs (ks_uni == 8 && kp_uni == 15): the pipelined kernels k_stage1_h2 / k_stage2_ord apply
    int bpc2o;                 // workgroups of k_stage2_ord per CU (its occupancy: three per CU)
    int s2_wgmap;              // k_stage2_ord: blocks of 4 source nodes per workgroup (large station counts)
    int bpc1b;                 // workgroups of k_stage1_h2 per CU in the grid (one is resident; more = dynamic balancing by the dispatcher)
    int use_h2;                // stage 1 on the 16-bit matrix pipe (k_stage1_h2); GENIE_S1=f32 selects the fp32-MFMA kernels
    // workspace offsets (floats)
    size_t o_xs, o_mm, o_c, o_wu, o_wv, o_part, o_sa0, o_sa1, o_bip, o_gpart, o_pj0, o_pj1, o_cv, ws_floats;
    size_t slot_stride;        // the G-sized buffers (o_part ... o_cv) exist GENIE_NSLOT times; `slot` selects the copy
    size_t big_stride;         // so do the P-sized stage-1 -> stage-2 buffers (c, wu, wv)
    int tail_cu_ro, tail_cu_sa; // grid caps (workgroups) of the read-out / SpatialAggregation kernels of the G-sized tail
    int slot;                  // lets window i+1's stage 1/2 overlap window i's G-sized kernels on another stream
};

namespace {

// The station processing order is honoured by k_split_rows / the embedding's split rows, k_stage1_h2 (through the relabelled
// station graph) and k_stage2_ord: active only while those are the kernels that run (use_absolute_pos together with the
// edge-feature variant takes the generic stage-1 kernel).
bool abs_generic(const genie_ctx* c) { return c->abs_sta != nullptr && (c->has_edges || !c->use_h2); }
bool sta_order_on(const genie_ctx* c) {
    return c->sta_perm != nullptr && !c->pcsr && c->use_h2 && !abs_generic(c) && !c->force_generic;
}

constexpr int GENIE_NSLOT = 33;  // copies of the G-sized per-window buffers (genie_set_slot): two batches of 16 windows in flight + one
                                 // for single-stream calls made while windows are pending
constexpr int GENIE_NBIG = 4;    // copies of the P-sized c / wu / wv rows: slot % GENIE_NBIG

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

void layout_ws(genie_ctx* c) {
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o = align_up(o + n, 64); return r; };
    c->o_xs = take((size_t)c->P_ext * (XROW / 4));   // single copy: written and read inside stage 1 only
    const size_t big0 = o;
    c->o_c = take((size_t)c->P * ROWC);
    c->o_wu = take((size_t)c->P * ROWW);
    c->o_wv = take((size_t)c->P_ext * ROWW);
    c->o_mm = take((size_t)c->P_ext);                // max_k Mask[p][k] in station processing order (split pass -> stage 2)
    c->big_stride = o - big0;
    o += (GENIE_NBIG - 1) * c->big_stride;    // further copies (slots 1..): stage 1 of window i+1 may run while stage 2 of window i reads
    const size_t small0 = o;
    c->o_part = take((size_t)c->G * c->T * 32);
    c->o_sa0 = take((size_t)c->G * 32);
    c->o_sa1 = take((size_t)c->G * 32);
    c->o_bip = take((size_t)c->G * 16);
    c->o_gpart = take(2 * 1024 * 8);
    c->o_pj0 = take((size_t)c->G * 32);
    c->o_pj1 = take((size_t)c->G * 32);
    c->o_cv = take((size_t)c->G * CVP);
    c->slot_stride = o - small0;
    o += (GENIE_NSLOT - 1) * c->slot_stride;  // further copies
    c->ws_floats = o;
}

template <typename T>
int dev_copy(T** dst, const T* src_dev, size_t n) {
    *dst = nullptr;
    if (n == 0) n = 1;
    HIP_TRY(hipMalloc((void**)dst, n * sizeof(T)));
    if (src_dev) HIP_TRY(hipMemcpy(*dst, src_dev, n * sizeof(T), hipMemcpyDeviceToDevice));
    return GENIE_OK;
}

int ensure_packed(genie_ctx* c, hipStream_t st) {
    if (!c->dirty) return GENIE_OK;
    if (!c->d_packplans) {
        std::vector<PackPlan> pl(NPLAN);
        int blocks = 0;
        for (int s = 0; s < NPLAN; ++s) {
            const StagePlan& p = c->plan[s];
            pl[s].steps = c->d_steps[s]; pl[s].bias = c->d_bias[s]; pl[s].scal = c->d_scal[s]; pl[s].out = c->packed[s];
            pl[s].n_groups = p.n_groups(); pl[s].n_bias = (int)p.bias.size(); pl[s].n_scal = (int)p.scal.size(); pl[s].block0 = blocks;
            blocks += (p.packed_floats() + 255) / 256;
        }
        HIP_TRY(hipMalloc(&c->d_packplans, sizeof(PackPlan) * NPLAN));
        HIP_TRY(hipMemcpy(c->d_packplans, pl.data(), sizeof(PackPlan) * NPLAN, hipMemcpyHostToDevice));
        c->pack_blocks = blocks;
    }
    k_pack_all<<<c->pack_blocks, 256, 0, st>>>(c->raw, (const PackPlan*)c->d_packplans, NPLAN);
    k_pack_h2<<<(H2_FRAGS * 64 + H2_NBIAS * 32 + 16 + 255) / 256, 256, 0, st>>>(c->raw, c->d_h2tbl, c->packed_h2, H2_FRAGS,
                                                                               H2_NBIAS * 32 + 16);
    if (c->has_edges) {
        k_edge_bias<<<(c->S * 48 + 255) / 256, 256, 0, st>>>(c->raw, g_params[W_DA_L1T12_P].off, g_params[W_DA_L2T12_P].off,
                                                            c->mpos_sta, c->S, c->ebias_sta);
        k_edge_bias<<<(c->G * 48 + 255) / 256, 256, 0, st>>>(c->raw, g_params[W_DA_L1T22_P].off, g_params[W_DA_L2T22_P].off,
                                                            c->mpos_src, c->G, c->ebias_src);
        if (c->sta_perm) {
            if (!c->ebias_sta_p) HIP_TRY(hipMalloc((void**)&c->ebias_sta_p, sizeof(float) * 48 * (size_t)c->S));
            k_permute_sta_rows<<<(c->S * 48 + 255) / 256, 256, 0, st>>>(c->ebias_sta, c->S, 48, c->sta_inv, c->S, c->ebias_sta_p);
        }
    }
    HIP_TRY(hipGetLastError());
    c->dirty = false;
    return GENIE_OK;
}

int da_grid_w(const genie_ctx* c, long long nitems_waves, int blocks_per_cu, int waves_per_block) {
    long long need = (nitems_waves + waves_per_block - 1) / waves_per_block;
    long long cap = (long long)c->num_cu * blocks_per_cu;
    long long g = std::min(need, cap);
    g = std::max<long long>(8, (g + 7) / 8 * 8);
    return (int)g;
}
int da_grid(const genie_ctx* c, long long nitems_waves, int blocks_per_cu) {
    long long need = (nitems_waves + WAVES - 1) / WAVES;
    long long cap = (long long)c->num_cu * blocks_per_cu;
    long long g = std::min(need, cap);
    g = std::max<long long>(8, (g + 7) / 8 * 8);
    return (int)g;
}

DaArgs make_da_args(const genie_ctx* c, float* ws) {
    DaArgs a;
    memset(&a, 0, sizeof(a));
    a.S = c->S; a.G = c->G; a.T = c->T;
    a.sta_rowptr = c->sta_rowptr; a.sta_col = c->sta_col; a.src_rowptr = c->src_rowptr; a.src_col = c->src_col;
    a.order = c->order;
    a.src_tab = c->src_tab;
    a.Pn = c->P;
    a.save = c->force_generic ? c->train_save : nullptr;
    if (c->pcsr) {
        a.sta_rowptr = c->p_sta_rowptr; a.sta_col = c->p_sta_col; a.src_rowptr = c->p_src_rowptr; a.src_col = c->p_src_col;
    }
    if (sta_order_on(c)) { a.sta_rowptr = c->sta_rowptr_p; a.sta_col = c->sta_col_p; a.sta_user = c->sta_perm; }
    a.abs_sta = c->abs_sta; a.abs_src = c->abs_src; a.abs_ts = c->abs_ts; a.abs_tg = c->abs_tg;
    a.eb_sta = c->has_edges ? (sta_order_on(c) ? c->ebias_sta_p : c->ebias_sta) : nullptr;
    a.eb_src = c->has_edges ? c->ebias_src : nullptr;
    a.seg = std::max(1, c->seg);
#if GENIE_TUNING
    { const char* e = getenv("GENIE_ABLATE"); a.abl = e ? atoi(e) : 0; }
#endif
    a.nxcd = 8;
    const size_t bo = (c->slot % GENIE_NBIG) * c->big_stride;
    a.c = ws + c->o_c + bo; a.wu = ws + c->o_wu + bo; a.wv = ws + c->o_wv + bo;
    a.part = ws + c->o_part + c->slot * c->slot_stride;
    return a;
}

// gradient maps of the three backward passes: accumulator / vector / scalar k of pass s -> entries of the gradient blob
int build_grad_maps(genie_ctx* c) {
    std::vector<AccDesc> acc[3];
    std::vector<VecDesc> vec[3];
    std::vector<int32_t> sc[3];
    auto A = [&](int s, int mat, int ld, int row0, int nrows, int col0, int ncols) {
        AccDesc d; d.mat_off = g_params[mat].off; d.ld = ld; d.row0 = row0; d.nrows = nrows; d.col0 = col0; d.ncols = ncols; d.n0 = 0; d.pad = 0;
        acc[s].push_back(d);
    };
    auto V = [&](int s, int vecid, int row0, int nrows) {
        VecDesc d; d.off = g_params[vecid].off; d.row0 = row0; d.nrows = nrows; d.stride = 1;
        vec[s].push_back(d);
    };
    auto rows2 = [](int b) { return b ? 14 : 16; };
    // pass 2': Bipartite fc1 (30 x 33)
    for (int t = 0; t < 2; ++t) {
        A(0, W_BP_FC1_W, 33, 16 * t, rows2(t), 0, 15);
        A(0, W_BP_FC1_W, 33, 16 * t, rows2(t), 15, 15);
        A(0, W_BP_FC1_W, 33, 16 * t, rows2(t), 30, 3);
    }
    for (int t = 0; t < 2; ++t) V(0, W_BP_FC1_B, 16 * t, rows2(t));
    sc[0] = {g_params[W_DA_ACT2].off, g_params[W_BP_ACT1].off};
    // pass 1': l2_t1_2 / l2_t2_2 (15 x 94), l2_t1_1 / l2_t2_1 (30 x 60)
    for (int w = 0; w < 2; ++w) {
        const int mat = w == 0 ? W_DA_L2T12_W : W_DA_L2T22_W;
        for (int k = 0; k < 4; ++k) A(1, mat, 94, 0, 15, (k >> 1) * 30 + 16 * (k & 1), rows2(k & 1));
        A(1, mat, 94, 0, 15, 90, 4);
        for (int b = 0; b < 2; ++b) A(1, mat, 94, 0, 15, 60 + 16 * b, rows2(b));
    }
    for (int w = 0; w < 2; ++w) {
        const int mat = w == 0 ? W_DA_L2T11_W : W_DA_L2T21_W;
        for (int b = 0; b < 2; ++b)
            for (int k = 0; k < 4; ++k) A(1, mat, 60, 16 * b, rows2(b), (k >> 1) * 30 + 16 * (k & 1), rows2(k & 1));
    }
    V(1, W_DA_L2T12_B, 0, 15); V(1, W_DA_L2T22_B, 0, 15);
    for (int b = 0; b < 2; ++b) V(1, W_DA_L2T11_B, 16 * b, rows2(b));
    for (int b = 0; b < 2; ++b) V(1, W_DA_L2T21_B, 16 * b, rows2(b));
    sc[1] = {g_params[W_DA_ACT1].off, g_params[W_DA_ACT21].off, g_params[W_DA_ACT22].off};
    // pass 0': init_trns (30 x 8), l1_t1_2 / l1_t2_2 (30 x 64)
    for (int b = 0; b < 2; ++b) A(2, W_DA_INIT_W, 8, 16 * b, rows2(b), 0, 8);
    for (int h = 0; h < 2; ++h) {
        const int mat = h == 0 ? W_DA_L1T12_W : W_DA_L1T22_W;
        for (int b = 0; b < 2; ++b) {
            A(2, mat, 64, 16 * b, rows2(b), 0, 16);
            A(2, mat, 64, 16 * b, rows2(b), 16, 14);
            A(2, mat, 64, 16 * b, rows2(b), 60, 4);
        }
        for (int b = 0; b < 2; ++b)
            for (int k = 0; k < 2; ++k) A(2, mat, 64, 16 * b, rows2(b), 30 + 16 * k, rows2(k));
    }
    for (int b = 0; b < 2; ++b) V(2, W_DA_INIT_B, 16 * b, rows2(b));
    for (int b = 0; b < 2; ++b) V(2, W_DA_L1T12_B, 16 * b, rows2(b));
    for (int b = 0; b < 2; ++b) V(2, W_DA_L1T22_B, 16 * b, rows2(b));
    sc[2] = {g_params[W_DA_ACT].off, g_params[W_DA_ACT11].off, g_params[W_DA_ACT12].off};
    const int want_acc[3] = {6, 30, 22}, want_vec[3] = {2, 6, 6};
    for (int s = 0; s < 3; ++s) {
        if ((int)acc[s].size() != want_acc[s] || (int)vec[s].size() != want_vec[s])
            return fail(GENIE_ERR_STATE, "internal: gradient maps do not match the backward kernels");
        c->n_acc[s] = (int)acc[s].size(); c->n_vec[s] = (int)vec[s].size(); c->n_sc[s] = (int)sc[s].size();
        HIP_TRY(hipMalloc((void**)&c->d_acc[s], sizeof(AccDesc) * acc[s].size()));
        HIP_TRY(hipMemcpy(c->d_acc[s], acc[s].data(), sizeof(AccDesc) * acc[s].size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc((void**)&c->d_vec[s], sizeof(VecDesc) * vec[s].size()));
        HIP_TRY(hipMemcpy(c->d_vec[s], vec[s].data(), sizeof(VecDesc) * vec[s].size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc((void**)&c->d_sc[s], sizeof(int32_t) * sc[s].size()));
        HIP_TRY(hipMemcpy(c->d_sc[s], sc[s].data(), sizeof(int32_t) * sc[s].size(), hipMemcpyHostToDevice));
    }
    return GENIE_OK;
}


// Transposed-weight plans of the tail's backward kernels (group maps GTR_* / GTN / GTS_* / GTB of train_tail_kernels.hpp)
void build_tail_train_plans(StagePlan* plan) {
    auto rows2 = [](int t) { return t ? 14 : 16; };
    for (int mode = 0; mode < 2; ++mode) {
        StagePlan& p = plan[mode == 0 ? PL_TRO0 : PL_TRO1];
        for (int k = 0; k < 2; ++k) add_block_group_T(p, W_TA_P1_W, 15, 0, 15, 16 * k, rows2(k));
        for (int m = 0; m < 2; ++m)
            for (int h = 0; h < 5; ++h)
                for (int b = 0; b < 2; ++b) add_block_group_T(p, m == 0 ? W_TA_V2_W : W_TA_C2_W, 30, 16 * b, rows2(b), 15 * h, 15);
        for (int m = 0; m < 2; ++m)
            for (int b = 0; b < 2; ++b)
                for (int k = 0; k < 2; ++k) add_block_group_T(p, m == 0 ? W_TA_V1_W : W_TA_C1_W, 30, 16 * b, rows2(b), 16 * k, rows2(k));
        for (int b = 0; b < 2; ++b)
            for (int k = 0; k < 2; ++k) {
                if (mode == 0) add_block_group_T(p, W_SD_W, 30, 16 * b, rows2(b), 16 * k, rows2(k));
                else if (b == 0) add_block_group_T(p, W_SAT_P_W, 15, 0, 15, 16 * k, rows2(k));
                else add_unused_group(p);
            }
        p.scal.push_back(g_params[W_TA_ACT1].off);          // unused (a plan carries at least one scalar)
    }
    {
        StagePlan& p = plan[PL_TSN];
        for (int m = 0; m < 2; ++m)
            for (int h = 0; h < 5; ++h)
                for (int b = 0; b < 2; ++b) add_block_group_T(p, m == 0 ? W_SAT_C_W : W_SAT_V_W, 33, 16 * b, rows2(b), 15 * h, 15);
        p.scal.push_back(g_params[W_SAT_ACT1].off);
    }
    for (int layer = 1; layer <= 3; ++layer) {
        StagePlan& p = plan[PL_TSA1 + layer - 1];
        const int base = layer == 1 ? W_SA1_FC1_W : (layer == 2 ? W_SA2_FC1_W : W_SA3_FC1_W);
        const int C = layer == 1 ? 15 : 30;
        auto xrows = [&](int b) { return C == 15 ? 15 : rows2(b); };
        for (int b = 0; b < 2; ++b)                       // fc2^T, x_i part
            for (int t = 0; t < 2; ++t) {
                if (C == 30 || b == 0) add_block_group_T(p, base + 2, C + 30, 16 * b, xrows(b), 16 * t, rows2(t));
                else add_unused_group(p);
            }
        for (int b = 0; b < 2; ++b)                       // fc2^T, edge-mean part
            for (int t = 0; t < 2; ++t) add_block_group_T(p, base + 2, C + 30, C + 16 * b, rows2(b), 16 * t, rows2(t));
        for (int b = 0; b < 2; ++b)                       // fc1[:, 0:C]^T
            for (int t = 0; t < 2; ++t) {
                if (C == 30 || b == 0) add_block_group_T(p, base, C + 8, 16 * b, xrows(b), 16 * t, rows2(t));
                else add_unused_group(p);
            }
        for (int b = 0; b < 2; ++b) {                     // fglobal^T
            if (C == 30 || b == 0) add_block_group_T(p, base + 4, C, 16 * b, xrows(b), 0, 5);
            else add_unused_group(p);
        }
        p.scal.push_back(g_params[base + 6].off);
    }
    {
        StagePlan& p = plan[PL_TBIP];
        for (int b = 0; b < 2; ++b) add_block_group_T(p, W_BP_FC2_W, 30, 16 * b, rows2(b), 0, 15);
        p.scal.push_back(g_params[W_BP_ACT2].off);
    }
    {   // association phase, pass 2' = k_train_b1<true>: the group map of build_train_plans' p1 with this head's weights
        StagePlan& p1 = plan[PL_TAB2];
        for (int b = 0; b < 2; ++b) add_block_group_T(p1, W_AS_L2T12_W, 95, 60 + 16 * b, rows2(b), 0, 15);
        for (int b = 0; b < 2; ++b) add_block_group_T(p1, W_AS_L2T22_W, 95, 60 + 16 * b, rows2(b), 0, 15);
        for (int hb = 0; hb < 4; ++hb) {
            const int col0 = (hb >> 1) * 30 + 16 * (hb & 1), rows = (hb & 1) ? 14 : 16;
            for (int src = 0; src < 2; ++src) add_block_group_T(p1, W_AS_L2T11_W, 60, col0, rows, 16 * src, src ? 14 : 16);
            for (int src = 0; src < 2; ++src) add_block_group_T(p1, W_AS_L2T21_W, 60, col0, rows, 16 * src, src ? 14 : 16);
            add_block_group_T(p1, W_AS_L2T12_W, 95, col0, rows, 0, 15);
            add_block_group_T(p1, W_AS_L2T22_W, 95, col0, rows, 0, 15);
        }
        for (int b = 0; b < 2; ++b)
            for (int k = 0; k < 4; ++k)
                add_block_group_T(p1, k < 2 ? W_AS_L1T12_W : W_AS_L1T22_W, 65, 16 * b, rows2(b), 16 * (k & 1), rows2(k & 1));
        p1.scal.push_back(g_params[W_AS_ACT1].off);
        p1.scal.push_back(g_params[W_AS_ACT21].off);
        p1.scal.push_back(g_params[W_AS_ACT22].off);
    }
    {
        StagePlan& p = plan[PL_TAB1];
        for (int h = 0; h < 2; ++h)
            for (int b = 0; b < 2; ++b)
                for (int k = 0; k < 2; ++k) add_block_group_T(p, h == 0 ? W_AS_L1T12_W : W_AS_L1T22_W, 65, 30 + 16 * b, rows2(b), 16 * k, rows2(k));
        for (int w = 0; w < 2; ++w)
            for (int b = 0; b < 2; ++b)
                for (int k = 0; k < 2; ++k) add_block_group_T(p, w == 0 ? W_AS_L1T11_W : W_AS_L1T21_W, 30, 16 * b, rows2(b), 16 * k, rows2(k));
        p.scal.push_back(g_params[W_AS_ACT].off);
        p.scal.push_back(g_params[W_AS_ACT11].off);
        p.scal.push_back(g_params[W_AS_ACT12].off);
    }
    {
        StagePlan& p = plan[PL_TAB0];
        for (int t = 0; t < 2; ++t) add_block_group_T(p, W_AS_INIT_W, 50, 0, 15, 16 * t, rows2(t));
        for (int b = 0; b < 2; ++b) add_block_group_T(p, W_RO_FC2_W, 30, 16 * b, rows2(b), 0, 15);
        p.scal.push_back(g_params[W_RO_ACT1].off);
        p.scal.push_back(g_params[W_RO_ACT2].off);
    }
    for (int ph = 0; ph < 2; ++ph) {
        StagePlan& p = plan[ph == 0 ? PL_TLSP : PL_TLSS];
        const int base = ph == 0 ? W_LP_FC1_W : W_LS_FC1_W;
        for (int b = 0; b < 2; ++b) add_block_group_T(p, base + 2, 30, 16 * b, rows2(b), 0, 15);
        for (int b = 0; b < 2; ++b)
            for (int t = 0; t < 2; ++t) add_block_group_T(p, base, 32, 16 * b, rows2(b), 16 * t, rows2(t));
        p.scal.push_back(g_params[base + 4].off);
    }
    {
        StagePlan& p = plan[PL_TAG];
        for (int b = 0; b < 2; ++b)
            for (int t = 0; t < 2; ++t) add_block_group_T(p, W_RO_FC1_W, 33, 16 * b, rows2(b), 16 * t, rows2(t));
        p.scal.push_back(g_params[W_RO_ACT1].off);
    }
    {
        StagePlan& p = plan[PL_TARR];             // the arrival head's transposed blocks (GTA_*)
        for (int t = 0; t < 2; ++t) add_block_group_T(p, W_AR_P1_W, 15, 0, 15, 16 * t, rows2(t));
        for (int m = 0; m < 2; ++m)
            for (int b = 0; b < 2; ++b)
                for (int h = 0; h < 3; ++h) add_block_group_T(p, m == 0 ? W_AR_V2_W : W_AR_Q2_W, 30, 16 * b, rows2(b), 15 * h, 15);
        for (int m = 0; m < 2; ++m)
            for (int s = 0; s < 2; ++s)
                for (int t = 0; t < 2; ++t) add_block_group_T(p, m == 0 ? W_AR_V1_W : W_AR_Q1_W, m == 0 ? 38 : 36, 15 * s, 15, 16 * t, rows2(t));
        p.scal.push_back(g_params[W_AR_ACT4].off);
    }
}

// gradient maps of the tail's backward kernels (accumulator / vector / scalar k of kernel TM_* -> entries of the gradient blob;
// the d(temporal query) blocks land behind the blob, at g_raw_total)
int build_tail_grad_maps(genie_ctx* c) {
    std::vector<AccDesc> acc[NTM];
    std::vector<VecDesc> vec[NTM];
    std::vector<int32_t> sc[NTM];
    auto A = [&](int s, int mat_off, int ld, int row0, int nrows, int col0, int ncols, int n0 = 0) {
        AccDesc d; d.mat_off = mat_off; d.ld = ld; d.row0 = row0; d.nrows = nrows; d.col0 = col0; d.ncols = ncols; d.n0 = n0; d.pad = 0;
        acc[s].push_back(d);
    };
    auto V = [&](int s, int off, int row0, int nrows, int stride = 1) {
        VecDesc d; d.off = off; d.row0 = row0; d.nrows = nrows; d.stride = stride;
        vec[s].push_back(d);
    };
    auto O = [](int w) { return g_params[w].off; };
    auto rows2 = [](int b) { return b ? 14 : 16; };
    for (int mode = 0; mode < 2; ++mode) {
        const int s = mode == 0 ? TM_RO0 : TM_RO1;
        for (int t = 0; t < 2; ++t) A(s, O(W_TA_P1_W), 15, 16 * t, rows2(t), 0, 15);
        for (int m = 0; m < 2; ++m)
            for (int t = 0; t < 2; ++t)
                for (int k = 0; k < 2; ++k) A(s, O(m == 0 ? W_TA_C1_W : W_TA_V1_W), 30, 16 * t, rows2(t), 16 * k, rows2(k));
        for (int m = 0; m < 2; ++m)
            for (int h = 0; h < 5; ++h)
                for (int k = 0; k < 2; ++k) A(s, O(m == 0 ? W_TA_C2_W : W_TA_V2_W), 30, 15 * h, 15, 16 * k, rows2(k));
        for (int h = 0; h < 5; ++h) A(s, g_raw_total, 75, 0, TQ_ROWS, 15 * h, 15);
        for (int t = 0; t < 2; ++t)
            for (int k = 0; k < 2; ++k) {
                if (mode == 0) A(s, O(W_SD_W), 30, 16 * t, rows2(t), 16 * k, rows2(k));
                else A(s, O(W_SAT_P_W), 15, 16 * t, k == 0 ? rows2(t) : 0, 0, 15);
            }
        if (mode == 1) {
            for (int m = 0; m < 3; ++m)
                for (int h = 0; h < 5; ++h) {
                    if (m == 0) A(s, O(W_SAT_Q_W), 3, 15 * h, 15, 0, 3);
                    else A(s, O(m == 1 ? W_SAT_C_W : W_SAT_V_W), 33, 15 * h, 15, 30, 3);
                }
            for (int h = 0; h < 5; ++h) A(s, O(W_SAT_Q_B) - 3, 1, 15 * h, 15, 0, 4, 3);      // column 3 of the f_queries blocks = bias gradient
        }
        for (int t = 0; t < 2; ++t) V(s, O(W_TA_P1_B), 16 * t, rows2(t));
        for (int t = 0; t < 2; ++t) V(s, O(W_TA_P2_W), 16 * t, rows2(t));
        for (int t = 0; t < 2; ++t) V(s, O(W_TA_C1_B), 16 * t, rows2(t));
        for (int t = 0; t < 2; ++t) V(s, O(W_TA_V1_B), 16 * t, rows2(t));
        for (int h = 0; h < 5; ++h) V(s, O(W_TA_C2_B), 15 * h, 15);
        for (int h = 0; h < 5; ++h) V(s, O(W_TA_V2_B), 15 * h, 15);
        for (int t = 0; t < 2; ++t) V(s, O(mode == 0 ? W_SD_B : W_SAT_P_B), 16 * t, rows2(t));
        sc[s] = {O(mode == 0 ? W_SD_ACT : W_SAT_ACT2), O(W_TA_ACT1), O(W_TA_ACT2), O(W_TA_ACT4), O(W_TA_ACT5), O(W_TA_P2_B)};
        if (mode == 1) sc[s].push_back(O(W_SAT_ACT1));
    }
    for (int m = 0; m < 2; ++m)
        for (int h = 0; h < 5; ++h)
            for (int k = 0; k < 2; ++k) A(TM_SN, O(m == 0 ? W_SAT_C_W : W_SAT_V_W), 33, 15 * h, 15, 16 * k, rows2(k));
    for (int m = 0; m < 2; ++m)
        for (int h = 0; h < 5; ++h) V(TM_SN, O(m == 0 ? W_SAT_C_B : W_SAT_V_B), 15 * h, 15);
    for (int layer = 1; layer <= 3; ++layer) {
        const int base = layer == 1 ? W_SA1_FC1_W : (layer == 2 ? W_SA2_FC1_W : W_SA3_FC1_W);
        const int C = layer == 1 ? 15 : 30;
        const int sa = TM_SAA1 + layer - 1, sb = TM_SAB1 + layer - 1;
        for (int t = 0; t < 2; ++t) {
            A(sa, O(base + 2), C + 30, 16 * t, rows2(t), 0, C == 15 ? 15 : 16);
            A(sa, O(base + 2), C + 30, 16 * t, C == 15 ? 0 : rows2(t), 16, 14);
            A(sa, O(base + 2), C + 30, 16 * t, rows2(t), C, 16);
            A(sa, O(base + 2), C + 30, 16 * t, rows2(t), C + 16, 14);
        }
        for (int t = 0; t < 2; ++t) V(sa, O(base + 3), 16 * t, rows2(t));
        for (int t = 0; t < 2; ++t) V(sa, O(base + 1), 16 * t, rows2(t));                     // d base = gradient of fc1's bias
        for (int d = 0; d < 3; ++d)
            for (int t = 0; t < 2; ++t) V(sa, O(base) + C + d, 16 * t, rows2(t), C + 8);      // fc1's position columns
        sc[sa] = {O(base + 7), O(base + 6)};
        for (int t = 0; t < 2; ++t) {
            A(sb, O(base), C + 8, 16 * t, rows2(t), 0, C == 15 ? 15 : 16);
            A(sb, O(base), C + 8, 16 * t, C == 15 ? 0 : rows2(t), 16, 14);
        }
        A(sb, O(base + 4), C, 0, 5, 0, C == 15 ? 15 : 16);
        A(sb, O(base + 4), C, 0, C == 15 ? 0 : 5, 16, 14);
        V(sb, O(base + 5), 0, 5);
        sc[sb] = {O(base + 8)};
    }
    for (int k = 0; k < 2; ++k) A(TM_BIP, O(W_BP_FC2_W), 30, 0, 15, 16 * k, rows2(k));
    V(TM_BIP, O(W_BP_FC2_B), 0, 15);
    sc[TM_BIP] = {O(W_BP_ACT2)};
    // ---- association phase
    sc[TM_AB3] = {O(W_AS_ACT2)};
    for (int w = 0; w < 2; ++w) {       // pass 2' (k_train_b1<true>): l2_t1_2 / l2_t2_2 (15 x 95), l2_t1_1 / l2_t2_1 (30 x 60)
        const int mat = O(w == 0 ? W_AS_L2T12_W : W_AS_L2T22_W);
        for (int k = 0; k < 4; ++k) A(TM_AB2, mat, 95, 0, 15, (k >> 1) * 30 + 16 * (k & 1), rows2(k & 1));
        A(TM_AB2, mat, 95, 0, 15, 91, 4);
        for (int b = 0; b < 2; ++b) A(TM_AB2, mat, 95, 0, 15, 60 + 16 * b, rows2(b));
    }
    for (int w = 0; w < 2; ++w) {
        const int mat = O(w == 0 ? W_AS_L2T11_W : W_AS_L2T21_W);
        for (int b = 0; b < 2; ++b)
            for (int k = 0; k < 4; ++k) A(TM_AB2, mat, 60, 16 * b, rows2(b), (k >> 1) * 30 + 16 * (k & 1), rows2(k & 1));
    }
    V(TM_AB2, O(W_AS_L2T12_B), 0, 15); V(TM_AB2, O(W_AS_L2T22_B), 0, 15);
    for (int b = 0; b < 2; ++b) V(TM_AB2, O(W_AS_L2T11_B), 16 * b, rows2(b));
    for (int b = 0; b < 2; ++b) V(TM_AB2, O(W_AS_L2T21_B), 16 * b, rows2(b));
    V(TM_AB2, O(W_AS_L2T12_W) + 90, 0, 15, 95); V(TM_AB2, O(W_AS_L2T22_W) + 90, 0, 15, 95);      // the mask1 columns
    sc[TM_AB2] = {O(W_AS_ACT1), O(W_AS_ACT21), O(W_AS_ACT22)};
    for (int h = 0; h < 2; ++h) {       // pass 1' (k_as_b1): l1_t1_2 / l1_t2_2 (30 x 65)
        const int mat = O(h == 0 ? W_AS_L1T12_W : W_AS_L1T22_W);
        for (int b = 0; b < 2; ++b) {
            A(TM_AB1, mat, 65, 16 * b, rows2(b), 0, 16);
            A(TM_AB1, mat, 65, 16 * b, rows2(b), 16, 14);
            A(TM_AB1, mat, 65, 16 * b, rows2(b), 61, 4);
        }
        for (int b = 0; b < 2; ++b)
            for (int k = 0; k < 2; ++k) A(TM_AB1, mat, 65, 16 * b, rows2(b), 30 + 16 * k, rows2(k));
    }
    for (int w = 0; w < 2; ++w)
        for (int b = 0; b < 2; ++b)
            for (int k = 0; k < 2; ++k) A(TM_AB1, O(w == 0 ? W_AS_L1T11_W : W_AS_L1T21_W), 30, 16 * b, rows2(b), 16 * k, rows2(k));
    for (int b = 0; b < 2; ++b) V(TM_AB1, O(W_AS_L1T12_B), 16 * b, rows2(b));
    for (int b = 0; b < 2; ++b) V(TM_AB1, O(W_AS_L1T22_B), 16 * b, rows2(b));
    for (int b = 0; b < 2; ++b) V(TM_AB1, O(W_AS_L1T12_W) + 60, 16 * b, rows2(b), 65);
    for (int b = 0; b < 2; ++b) V(TM_AB1, O(W_AS_L1T22_W) + 60, 16 * b, rows2(b), 65);
    for (int b = 0; b < 2; ++b) V(TM_AB1, O(W_AS_L1T11_B), 16 * b, rows2(b));
    for (int b = 0; b < 2; ++b) V(TM_AB1, O(W_AS_L1T21_B), 16 * b, rows2(b));
    sc[TM_AB1] = {O(W_AS_ACT), O(W_AS_ACT11), O(W_AS_ACT12)};
    for (int t = 0; t < 2; ++t) {       // pass 0' (k_as_b0): init_trns (30 x 50), the read-out operator's fc2 (15 x 30), fc1's edge columns
        A(TM_AB0, O(W_AS_INIT_W), 50, 16 * t, rows2(t), 0, 15);
        A(TM_AB0, O(W_AS_INIT_W), 50, 16 * t, rows2(t), 15, 16);
        A(TM_AB0, O(W_AS_INIT_W), 50, 16 * t, rows2(t), 31, 14);
        A(TM_AB0, O(W_AS_INIT_W), 50, 16 * t, rows2(t), 46, 4);
    }
    for (int k = 0; k < 2; ++k) A(TM_AB0, O(W_RO_FC2_W), 30, 0, 15, 16 * k, rows2(k));
    for (int t = 0; t < 2; ++t) A(TM_AB0, O(W_RO_FC1_W), 33, 16 * t, rows2(t), 30, 3);
    for (int t = 0; t < 2; ++t) V(TM_AB0, O(W_AS_INIT_B), 16 * t, rows2(t));
    for (int t = 0; t < 2; ++t) V(TM_AB0, O(W_AS_INIT_W) + 45, 16 * t, rows2(t), 50);
    V(TM_AB0, O(W_RO_FC2_B), 0, 15);
    for (int t = 0; t < 2; ++t) V(TM_AB0, O(W_RO_FC1_B), 16 * t, rows2(t));
    sc[TM_AB0] = {O(W_RO_ACT1), O(W_RO_ACT2)};
    for (int t = 0; t < 2; ++t)
        for (int k = 0; k < 2; ++k) A(TM_AG, O(W_RO_FC1_W), 33, 16 * t, rows2(t), 16 * k, rows2(k));
    for (int ph = 0; ph < 2; ++ph) {     // LocalSliceLgCollapse P / S (k_lslc_bwd): fc1 (30 x 32), fc2 (15 x 30)
        const int tm = ph == 0 ? TM_LSP : TM_LSS, base = ph == 0 ? W_LP_FC1_W : W_LS_FC1_W;
        for (int t = 0; t < 2; ++t) {
            A(tm, O(base), 32, 16 * t, rows2(t), 0, 16);
            A(tm, O(base), 32, 16 * t, rows2(t), 16, 14);
            A(tm, O(base), 32, 16 * t, rows2(t), 30, 2);
        }
        for (int k = 0; k < 2; ++k) A(tm, O(base + 2), 30, 0, 15, 16 * k, rows2(k));
        for (int t = 0; t < 2; ++t) V(tm, O(base + 1), 16 * t, rows2(t));
        V(tm, O(base + 3), 0, 15);
        sc[tm] = {O(base + 4), O(base + 5)};
    }
    {   // the arrival head: k_arrt_tgt_bwd (proj_1, proj_2), k_arrt_ent_bwd (the query / value edge MLPs; 3 context tiles it sums itself)
        for (int t = 0; t < 2; ++t) A(TM_ART, O(W_AR_P1_W), 15, 16 * t, rows2(t), 0, 15);
        for (int t = 0; t < 2; ++t) V(TM_ART, O(W_AR_P1_B), 16 * t, rows2(t));
        for (int m = 0; m < 2; ++m)
            for (int t = 0; t < 2; ++t) V(TM_ART, O(W_AR_P2_W), 30 * m + 16 * t, rows2(t));
        sc[TM_ART] = {O(W_AR_ACT4), O(W_AR_P2_B), O(W_AR_P2_B) + 1};
        for (int m = 0; m < 2; ++m) {
            const int mat = m == 0 ? W_AR_Q1_W : W_AR_V1_W, ld = m == 0 ? 36 : 38;
            for (int t = 0; t < 2; ++t) {
                A(TM_ARE, O(mat), ld, 16 * t, rows2(t), 0, 15);
                A(TM_ARE, O(mat), ld, 16 * t, rows2(t), 15, 15);
                A(TM_ARE, O(mat), ld, 16 * t, rows2(t), 30, m == 0 ? 6 : 8);
            }
        }
        for (int m = 0; m < 2; ++m)
            for (int h = 0; h < 3; ++h)
                for (int b = 0; b < 2; ++b) A(TM_ARE, O(m == 0 ? W_AR_Q2_W : W_AR_V2_W), 30, 15 * h, 15, 16 * b, rows2(b));
        for (int t = 0; t < 2; ++t) V(TM_ARE, O(W_AR_Q1_B), 16 * t, rows2(t));
        for (int t = 0; t < 2; ++t) V(TM_ARE, O(W_AR_V1_B), 16 * t, rows2(t));
        for (int h = 0; h < 3; ++h) V(TM_ARE, O(W_AR_Q2_B), 15 * h, 15);
        for (int h = 0; h < 3; ++h) V(TM_ARE, O(W_AR_V2_B), 15 * h, 15);
        sc[TM_ARE] = {O(W_AR_ACT2), O(W_AR_ACT3)};
    }
    const int want_acc[NTM] = {0, 0, 0, RB_NACC0, RB_NACC1, GTN_GROUPS, SBA_NACC, SBA_NACC, SBA_NACC, SBB_NACC, SBB_NACC, SBB_NACC, 2, 0, 30, 28, 12, 4, 8, 8,
                               AT_NACC, AE_NACC};
    const int want_vec[NTM] = {0, 0, 0, RB_NVEC, RB_NVEC, 10, SBA_NVEC, SBA_NVEC, SBA_NVEC, SBB_NVEC, SBB_NVEC, SBB_NVEC, 1, 0, 8, 12, 7, 0, 3, 3,
                               AT_NVEC, AE_NVEC};
    for (int s = TM_RO0; s < NTM; ++s) {
        if ((int)acc[s].size() != want_acc[s] || (int)vec[s].size() != want_vec[s] || sc[s].size() > 16)
            return fail(GENIE_ERR_STATE, "internal: tail gradient maps do not match the backward kernels");
        c->n_acc[s] = (int)acc[s].size(); c->n_vec[s] = (int)vec[s].size(); c->n_sc[s] = (int)sc[s].size();
        HIP_TRY(hipMalloc((void**)&c->d_acc[s], sizeof(AccDesc) * std::max<size_t>(1, acc[s].size())));
        if (!acc[s].empty()) HIP_TRY(hipMemcpy(c->d_acc[s], acc[s].data(), sizeof(AccDesc) * acc[s].size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc((void**)&c->d_vec[s], sizeof(VecDesc) * std::max<size_t>(1, vec[s].size())));
        if (!vec[s].empty()) HIP_TRY(hipMemcpy(c->d_vec[s], vec[s].data(), sizeof(VecDesc) * vec[s].size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc((void**)&c->d_sc[s], sizeof(int32_t) * std::max<size_t>(1, sc[s].size())));
        if (!sc[s].empty()) HIP_TRY(hipMemcpy(c->d_sc[s], sc[s].data(), sizeof(int32_t) * sc[s].size(), hipMemcpyHostToDevice));
    }
    return GENIE_OK;
}

struct CtxGuard {            // destroys a partially built context on every early return of the create calls
    genie_ctx* c;
    ~CtxGuard() { if (c) genie_ctx_destroy(c); }
};

// workgroups (4 waves x 16 nodes) of an MFMA tail kernel over n nodes, at most `cap`
int tl_blocks(long long n, int cap) { return (int)std::max<long long>(1, std::min<long long>((n + 63) / 64, cap)); }

int check_ws(const genie_ctx* c, const void* ws) {
    if (!c) return fail(GENIE_ERR_ARG, "null context");
    if (!ws) return fail(GENIE_ERR_ARG, "null workspace");
    if (((uintptr_t)ws & 255) != 0) return fail(GENIE_ERR_ARG, "workspace must be 256-byte aligned");
    return GENIE_OK;
}

}  // namespace

extern "C" {

int genie_version(void) { return 100; }
const char* genie_last_error(void) { return g_err.c_str(); }

int genie_weights_count(void) { init_registry(); return W_COUNT; }
const char* genie_weights_name(int i) { init_registry(); return (i >= 0 && i < W_COUNT) ? g_params[i].name : ""; }
int64_t genie_weights_numel(int i) { init_registry(); return (i >= 0 && i < W_COUNT) ? g_params[i].numel : -1; }

int genie_ctx_create(genie_ctx** out, int n_sta, int n_grid, int n_grid_ext, const int32_t* sta_rowptr,
                     const int32_t* sta_col, const int32_t* src_rowptr, const int32_t* src_col,
                     const int32_t* grid_order, float scale_rel) {
    init_registry();
    if (!out) return fail(GENIE_ERR_ARG, "out is null");
    *out = nullptr;
    if (n_sta < 1 || n_grid < 1 || n_grid_ext < n_grid) return fail(GENIE_ERR_ARG, "bad n_sta / n_grid / n_grid_ext");
    if (!sta_rowptr || !src_rowptr) return fail(GENIE_ERR_ARG, "null rowptr");
    genie_ctx* c = new genie_ctx();      // value-initialised: every pointer member starts null
    CtxGuard guard{c};                   // every early return below destroys the partially built context
    memset((void*)&c->S, 0, sizeof(int) * 4);
    c->S = n_sta; c->G = n_grid; c->G_ext = n_grid_ext; c->T = (n_sta + 15) / 16;
    c->scale_rel = scale_rel;
    c->scale_t = 9.0f;  // 3 * kernel_sig_t (module.py:40, train_config.yaml:17); override with genie_set_scale_t
    c->P = (long long)n_grid * n_sta; c->P_ext = (long long)n_grid_ext * n_sta;
    int32_t e_sta = 0, e_src = 0;
    HIP_TRY(hipMemcpy(&e_sta, sta_rowptr + n_sta, sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&e_src, src_rowptr + n_grid, sizeof(int32_t), hipMemcpyDeviceToHost));
    if (e_sta < 0 || e_src < 0) return fail(GENIE_ERR_ARG, "negative edge count in rowptr");
    if ((e_sta > 0 && !sta_col) || (e_src > 0 && !src_col)) return fail(GENIE_ERR_ARG, "null col array");
    c->E_src = e_src;
    {   // uniform-degree detection (kNN graphs) selects the pipelined stage-1 kernel
        std::vector<int32_t> rp((size_t)std::max(n_sta, n_grid) + 1);
        auto uniform = [&](const int32_t* dev, int n) -> int {
            if (hipMemcpy(rp.data(), dev, sizeof(int32_t) * ((size_t)n + 1), hipMemcpyDeviceToHost) != hipSuccess) return -1;
            const int k = rp[1] - rp[0];
            for (int i = 0; i < n; ++i)
                if (rp[i + 1] - rp[i] != k) return -1;
            return k;
        };
        c->ks_uni = uniform(sta_rowptr, n_sta);
        c->kp_uni = uniform(src_rowptr, n_grid);
    }
    int rc, rc_tr = 0;
    if ((rc = dev_copy(&c->sta_rowptr, sta_rowptr, (size_t)n_sta + 1))) return rc;
    if ((rc = dev_copy(&c->sta_col, sta_col, (size_t)e_sta))) return rc;
    if ((rc = dev_copy(&c->src_rowptr, src_rowptr, (size_t)n_grid + 1))) return rc;
    if ((rc = dev_copy(&c->src_col, src_col, (size_t)e_src))) return rc;
    if (grid_order) {
        if ((rc = dev_copy(&c->order, grid_order, (size_t)n_grid))) return rc;
    } else {
        std::vector<int32_t> id(n_grid);
        for (int i = 0; i < n_grid; ++i) id[i] = i;
        HIP_TRY(hipMalloc((void**)&c->order, sizeof(int32_t) * n_grid));
        HIP_TRY(hipMemcpy(c->order, id.data(), sizeof(int32_t) * n_grid, hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMalloc((void**)&c->outdeg, sizeof(int32_t) * (size_t)n_grid_ext));
    HIP_TRY(hipMemset(c->outdeg, 0, sizeof(int32_t) * (size_t)n_grid_ext));
    if (e_src > 0) k_outdeg<<<(e_src + 255) / 256, 256>>>(c->src_col, e_src, c->outdeg);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMalloc((void**)&c->raw, sizeof(float) * g_raw_total));
    HIP_TRY(hipMemset(c->raw, 0, sizeof(float) * g_raw_total));
    build_plans(c->plan[0], c->plan[1]);
    build_assoc_plans(c->plan[2], c->plan[3]);
    build_train_plans(c->plan[4], c->plan[5], c->plan[6]);
    build_tail_plans(c->plan);
    build_tail_train_plans(c->plan);
    if (c->plan[PL_TRO0].n_groups() != GTR_GROUPS || c->plan[PL_TRO1].n_groups() != GTR_GROUPS || c->plan[PL_TSN].n_groups() != GTN_GROUPS ||
        c->plan[PL_TSA1].n_groups() != GTS_GROUPS || c->plan[PL_TSA3].n_groups() != GTS_GROUPS || c->plan[PL_TBIP].n_groups() != GTB_GROUPS ||
        c->plan[PL_TAB2].n_groups() != GT1_GROUPS || c->plan[PL_TAB1].n_groups() != GA1_GROUPS || c->plan[PL_TAB0].n_groups() != GA0_GROUPS ||
        c->plan[PL_TAG].n_groups() != GAG_GROUPS || c->plan[PL_TLSP].n_groups() != GLT_GROUPS || c->plan[PL_TLSS].n_groups() != GLT_GROUPS ||
        c->plan[PL_TARR].n_groups() != GTA_GROUPS)
        return fail(GENIE_ERR_STATE, "internal: transposed tail plan does not match kernel group maps");
    if (c->plan[PL_RO0].n_groups() != GR_GROUPS || c->plan[PL_RO1].n_groups() != GR_GROUPS || (int)c->plan[PL_RO0].bias.size() != GR_BIAS ||
        (int)c->plan[PL_RO1].bias.size() != GR_BIAS || c->plan[PL_ROP].n_groups() != GP_GROUPS || (int)c->plan[PL_ROP].bias.size() != GP_BIAS ||
        c->plan[PL_SA1].n_groups() != GS_GROUPS || c->plan[PL_SA2].n_groups() != GS_GROUPS || c->plan[PL_SA3].n_groups() != GS_GROUPS ||
        (int)c->plan[PL_SA1].bias.size() != GS_BIAS || (int)c->plan[PL_SA3].bias.size() != GS_BIAS || c->plan[PL_BIP].n_groups() != GB_GROUPS2)
        return fail(GENIE_ERR_STATE, "internal: tail plan does not match kernel group maps");
    if (c->plan[4].n_groups() != GT2_GROUPS || c->plan[5].n_groups() != GT1_GROUPS || c->plan[6].n_groups() != GT0_GROUPS)
        return fail(GENIE_ERR_STATE, "internal: backward plan does not match kernel group maps");
    if ((rc_tr = build_grad_maps(c))) return rc_tr;
    if ((rc_tr = build_tail_grad_maps(c))) return rc_tr;
    if (c->plan[0].n_groups() != G1_GROUPS || c->plan[1].n_groups() != G2_GROUPS ||
        (int)c->plan[0].bias.size() != G1_BIAS || (int)c->plan[1].bias.size() != G2_BIAS ||
        c->plan[2].n_groups() != GA_GROUPS || c->plan[3].n_groups() != GB_GROUPS ||
        (int)c->plan[2].bias.size() != GA_BIAS || (int)c->plan[3].bias.size() != GB_BIAS)
        return fail(GENIE_ERR_STATE, "internal: stage plan does not match kernel group maps");
    for (int s = 0; s < NPLAN; ++s) {
        const StagePlan& p = c->plan[s];
        HIP_TRY(hipMalloc((void**)&c->d_steps[s], sizeof(StepDesc) * p.steps.size()));
        HIP_TRY(hipMemcpy(c->d_steps[s], p.steps.data(), sizeof(StepDesc) * p.steps.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc((void**)&c->d_bias[s], sizeof(BiasDesc) * std::max<size_t>(1, p.bias.size())));
        if (!p.bias.empty()) HIP_TRY(hipMemcpy(c->d_bias[s], p.bias.data(), sizeof(BiasDesc) * p.bias.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc((void**)&c->d_scal[s], sizeof(int32_t) * 16));
        HIP_TRY(hipMemcpy(c->d_scal[s], p.scal.data(), sizeof(int32_t) * p.scal.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc((void**)&c->packed[s], sizeof(float) * p.packed_floats()));
    }
    {
        std::vector<int32_t> tbl;
        build_h2_table(tbl);
        HIP_TRY(hipMalloc((void**)&c->d_h2tbl, sizeof(int32_t) * tbl.size()));
        HIP_TRY(hipMemcpy(c->d_h2tbl, tbl.data(), sizeof(int32_t) * tbl.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc((void**)&c->packed_h2, sizeof(float) * H2_IMG_FLOATS));
    }
    c->mpos_sta = c->mpos_src = c->ebias_sta = c->ebias_src = nullptr;
    c->has_edges = false;
    c->xs_slice = c->xs_mask = nullptr; c->xs_ws = nullptr; c->xs_mm_copy = 0;
    c->abs_sta = c->abs_src = nullptr; c->abs_ts = c->abs_tg = nullptr; c->abs_dirty = false;
    c->r_sta_rowptr = c->r_sta_col = c->r_src_rowptr = c->r_src_col = nullptr;
    c->sta_perm = c->sta_inv = c->sta_rowptr_p = c->sta_col_p = nullptr; c->ebias_sta_p = nullptr;
    c->ea_int = c->ea_tmp = nullptr; c->ea_user = nullptr;
    c->r_sta_w = c->r_src_w = nullptr;
    c->pcsr = false; c->pcsr_h2 = false;
    c->p_sta_rowptr = c->p_sta_col = c->p_src_rowptr = c->p_src_col = c->seg_rowptr = nullptr;
    c->src_tab = nullptr;
    if (c->kp_uni == 15) {
        std::vector<int32_t> ord(n_grid), col((size_t)e_src), tab((size_t)n_grid * 16);
        HIP_TRY(hipMemcpy(ord.data(), c->order, sizeof(int32_t) * n_grid, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(col.data(), c->src_col, sizeof(int32_t) * (size_t)e_src, hipMemcpyDeviceToHost));
        for (int gi = 0; gi < n_grid; ++gi) {
            const int gg = ord[gi];
            if (gg < 0 || gg >= n_grid) return fail(GENIE_ERR_ARG, "grid_order is not a permutation of the source nodes");
            tab[(size_t)gi * 16] = gg;
            for (int k = 0; k < 15; ++k) tab[(size_t)gi * 16 + 1 + k] = col[(size_t)gg * 15 + k];
        }
        HIP_TRY(hipMalloc((void**)&c->src_tab, sizeof(int32_t) * tab.size()));
        HIP_TRY(hipMemcpy(c->src_tab, tab.data(), sizeof(int32_t) * tab.size(), hipMemcpyHostToDevice));
    }
    c->dirty = true;
    int dev = 0;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    {
        const char* e;
        // scheduling segments: node-major sweeps (1) at config 2; at 2000 stations station-tile-major sweeps over segments of 16
        // source nodes keep the source-neighbour rows of stage 1 in L2 (config 4 on one GPU: stage 1 25.8 -> 24.8 ms)
        c->seg = (e = getenv("GENIE_SEG")) ? atoi(e) : (n_sta >= 1024 ? 16 : 1);
        // G-sized tail: few, long-lived workgroups. Next to the persistent P-sized kernels a tail workgroup only runs when one
        // of theirs retires and keeps that CU until it ends, so what the tail costs the main stream is its CU-time = workgroups x
        // duration, and most of a short tail workgroup is fixed cost (its LDS weight image). Two 62-KB read-out workgroups per CU
        // hide the gather latency of k_readout_m<1> (window 0.769 -> 0.765 ms).
        c->tail_cu_ro = c->num_cu * 2;          // genie_set_tail_grid
        c->tail_cu_sa = c->num_cu * 2;
        // persistent grids: exactly as many workgroups as are co-resident (a larger grid runs in two uneven rounds)
        int occ1 = 0, occ2 = 0, occo = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ1, k_stage1, 256, 0));
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ2, k_stage2, 256, 0));
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occo, k_stage2_ord<8, 15, false>, 256, 0));
        c->bpc1 = std::max(1, occ1);
        c->bpc2 = (e = getenv("GENIE_BPC2")) ? atoi(e) : std::max(1, occ2);
        // Large station counts (config 4: 2000 stations, 128 KB of wu / wv rows per source node): the gathers leave L2, and what
        // pays is locality, not concurrency: blocks of 4 adjacent source nodes per workgroup (one node per wave: the four waves
        // share half of their source rows, every wu block is gathered on one CU) and two workgroups per CU. Config 4 on one
        // GPU: stage 2 16.1 -> 11.8 ms (three workgroups, interleaved items: 16.1; two: 14.8; block map alone: 13.1). At 200
        // stations the same settings lose (0.266 -> 0.268 ms), hence by size.
        c->s2_wgmap = (e = getenv("GENIE_S2_WGMAP")) ? (atoi(e) != 0) : (n_sta >= 1024);
        c->bpc2o = (e = getenv("GENIE_BPC2")) ? atoi(e) : (c->s2_wgmap ? std::min(2, std::max(1, occo)) : std::max(1, occo));
        // the reference's kNN graphs (8 station / 15 source neighbours everywhere): pipelined kernels k_stage1_h2 / k_stage2_ord
        c->use_fast = c->ks_uni == 8 && c->kp_uni == 15;
        // f16x2 stage 1: 24-bit multiplicands (64-bit row offsets are a template variant); GENIE_S1=f32 = the generic fp32-MFMA
        // kernels, the A/B reference
        c->use_h2 = (c->use_fast && n_grid_ext < (1 << 24) && (long long)n_sta * XROW < (1 << 24) &&
                     !((e = getenv("GENIE_S1")) && strcmp(e, "f32") == 0));
        // 4 workgroups per CU in the grid, one resident: a workgroup held up by a tail kernel of the previous window then costs a
        // quarter of a share, not a whole one (pipelined window 0.877 -> 0.866 ms; no effect on the kernel alone)
        c->bpc1b = 4;
    }
#if GENIE_TUNING
    {
        const char* e = getenv("GENIE_ABLATE");
        int v = (e && (atoi(e) & 4)) ? 1 : 0;
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_abl_mfma), &v, sizeof(int)));
    }
#endif
    layout_ws(c);
    HIP_TRY(hipDeviceSynchronize());
    guard.c = nullptr;
    *out = c;
    return GENIE_OK;
}

int genie_ctx_create_subgraph(genie_ctx** out, int n_sta, int n_grid, int64_t n_prod, const int32_t* p_sta_rowptr,
                              const int32_t* p_sta_col, const int32_t* p_src_rowptr, const int32_t* p_src_col,
                              const int32_t* seg_rowptr, const int32_t* src_rowptr, const int32_t* src_col,
                              const int32_t* grid_order, float scale_rel) {
    if (!out) return fail(GENIE_ERR_ARG, "out is null");
    *out = nullptr;
    if (n_prod < 1 || n_prod >= (1ll << 31)) return fail(GENIE_ERR_ARG, "genie_ctx_create_subgraph: bad n_prod");
    if (!p_sta_rowptr || !p_src_rowptr || !seg_rowptr) return fail(GENIE_ERR_ARG, "genie_ctx_create_subgraph: null rowptr");
    int32_t* zeros = nullptr;       // empty base station graph: the station edges live in the product-level CSR
    HIP_TRY(hipMalloc((void**)&zeros, sizeof(int32_t) * ((size_t)n_sta + 1)));
    genie_ctx* c = nullptr;
    int rc = hipMemset(zeros, 0, sizeof(int32_t) * ((size_t)n_sta + 1)) == hipSuccess
                 ? genie_ctx_create(&c, n_sta, n_grid, n_grid, zeros, nullptr, src_rowptr, src_col, grid_order, scale_rel)
                 : fail(GENIE_ERR_HIP, "genie_ctx_create_subgraph: hipMemset failed");
    (void)hipFree(zeros);
    if (rc) return rc;
    CtxGuard guard{c};
    int32_t e1 = 0, e2 = 0, last = 0;
    HIP_TRY(hipMemcpy(&e1, p_sta_rowptr + n_prod, sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&e2, p_src_rowptr + n_prod, sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&last, seg_rowptr + n_grid, sizeof(int32_t), hipMemcpyDeviceToHost));
    if (e1 < 0 || e2 < 0 || last != (int32_t)n_prod || (e1 > 0 && !p_sta_col) || (e2 > 0 && !p_src_col)) {
        return fail(GENIE_ERR_ARG, "genie_ctx_create_subgraph: inconsistent CSR arrays (seg_rowptr[n_grid] must equal n_prod)");
    }
    if ((rc = dev_copy(&c->p_sta_rowptr, p_sta_rowptr, (size_t)n_prod + 1)) || (rc = dev_copy(&c->p_sta_col, p_sta_col, (size_t)e1)) ||
        (rc = dev_copy(&c->p_src_rowptr, p_src_rowptr, (size_t)n_prod + 1)) || (rc = dev_copy(&c->p_src_col, p_src_col, (size_t)e2)) ||
        (rc = dev_copy(&c->seg_rowptr, seg_rowptr, (size_t)n_grid + 1))) {
        return rc;
    }
    c->pcsr = true;
    c->P = c->P_ext = n_prod;
    {   // the f16x2 stage-1 kernel unrolls 8 station + 15 source neighbour slots per node: enough for every induced subgraph of
        // the reference's kNN product graphs (process_utils.py:824-839); other graphs keep the generic fp32-MFMA kernel
        std::vector<int32_t> r1((size_t)n_prod + 1), r2((size_t)n_prod + 1);
        HIP_TRY(hipMemcpy(r1.data(), p_sta_rowptr, sizeof(int32_t) * r1.size(), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(r2.data(), p_src_rowptr, sizeof(int32_t) * r2.size(), hipMemcpyDeviceToHost));
        int m1 = 0, m2 = 0;
        for (long long i = 0; i < n_prod; ++i) { m1 = std::max(m1, r1[i + 1] - r1[i]); m2 = std::max(m2, r2[i + 1] - r2[i]); }
        const char* e = getenv("GENIE_S1");
        c->pcsr_h2 = m1 <= 8 && m2 <= 15 && !(e && strcmp(e, "f32") == 0);
    }
    c->use_fast = c->use_h2 = 0;
    c->ks_uni = c->kp_uni = -1;
    layout_ws(c);
    guard.c = nullptr;
    *out = c;
    return GENIE_OK;
}

int genie_set_absolute_pos(genie_ctx* c, const float* pos_sta, const float* pos_src, void* stream) {
    if (!c) return fail(GENIE_ERR_ARG, "genie_set_absolute_pos: null context");
    if (!pos_sta || !pos_src) {
        (void)hipFree(c->abs_sta); (void)hipFree(c->abs_src); (void)hipFree(c->abs_ts); (void)hipFree(c->abs_tg);
        c->abs_sta = c->abs_src = nullptr; c->abs_ts = c->abs_tg = nullptr;
        return GENIE_OK;
    }
    if (c->pcsr) return fail(GENIE_ERR_STATE, "genie_set_absolute_pos: not available on an irregular product graph");
    if (!c->abs_sta) {
        HIP_TRY(hipMalloc((void**)&c->abs_sta, sizeof(float) * 4 * (size_t)c->S));
        HIP_TRY(hipMalloc((void**)&c->abs_src, sizeof(float) * 4 * (size_t)c->G_ext));
    }
    const float inv = 1.f / (3.f * c->scale_rel);
    hipStream_t st = (hipStream_t)stream;
    k_abs_table<<<(c->S * 4 + 255) / 256, 256, 0, st>>>(pos_sta, c->S, inv, c->abs_sta);
    k_abs_table<<<(c->G_ext * 4 + 255) / 256, 256, 0, st>>>(pos_src, c->G_ext, inv, c->abs_src);
    HIP_TRY(hipGetLastError());
    c->abs_dirty = true;
    return GENIE_OK;
}

int genie_set_edge_features(genie_ctx* c, const float* pos_sta, const float* pos_src, void* stream) {
    if (!c) return fail(GENIE_ERR_ARG, "genie_set_edge_features: null context");
    if (c->pcsr && pos_sta) return fail(GENIE_ERR_STATE, "genie_set_edge_features: not available on an irregular product graph");
    hipStream_t st = (hipStream_t)stream;
    if (!pos_sta || !pos_src) {      // back to plain DataAggregation
        c->has_edges = false;
        return GENIE_OK;
    }
    if (!c->mpos_sta) {
        HIP_TRY(hipMalloc((void**)&c->mpos_sta, sizeof(float) * 4 * (size_t)c->S));
        HIP_TRY(hipMalloc((void**)&c->mpos_src, sizeof(float) * 4 * (size_t)c->G));
        HIP_TRY(hipMalloc((void**)&c->ebias_sta, sizeof(float) * 48 * (size_t)c->S));
        HIP_TRY(hipMalloc((void**)&c->ebias_src, sizeof(float) * 48 * (size_t)c->G));
    }
    k_edge_feat<<<(c->S + 255) / 256, 256, 0, st>>>(c->sta_rowptr, c->sta_col, c->S, pos_sta, c->scale_rel, c->mpos_sta);
    k_edge_feat<<<(c->G + 255) / 256, 256, 0, st>>>(c->src_rowptr, c->src_col, c->G, pos_src, c->scale_rel, c->mpos_src);
    HIP_TRY(hipGetLastError());
    c->has_edges = true;
    c->dirty = true;
    return GENIE_OK;
}

int genie_set_scale_t(genie_ctx* c, float scale_t) {
    if (!c || !(scale_t > 0.f)) return fail(GENIE_ERR_ARG, "genie_set_scale_t: bad argument");
    c->scale_t = scale_t;
    return GENIE_OK;
}

int genie_set_station_order(genie_ctx* c, const int32_t* order_host) {
    if (!c) return fail(GENIE_ERR_ARG, "genie_set_station_order: null context");
    void* old[] = {c->sta_perm, c->sta_inv, c->sta_rowptr_p, c->sta_col_p, c->ea_int};
    for (void* q : old) (void)hipFree(q);
    c->sta_perm = c->sta_inv = c->sta_rowptr_p = c->sta_col_p = nullptr;
    c->ea_int = c->ea_tmp = nullptr; c->ea_user = nullptr;
    c->dirty = true; c->abs_dirty = true;
    if (!order_host || c->pcsr) return GENIE_OK;
    const int S = c->S;
    std::vector<int32_t> perm(order_host, order_host + S), inv((size_t)S, -1);
    for (int i = 0; i < S; ++i) {
        if (perm[i] < 0 || perm[i] >= S || inv[perm[i]] >= 0) return fail(GENIE_ERR_ARG, "genie_set_station_order: not a permutation of 0..n_sta-1");
        inv[perm[i]] = i;
    }
    std::vector<int32_t> rp((size_t)S + 1);
    HIP_TRY(hipMemcpy(rp.data(), c->sta_rowptr, sizeof(int32_t) * rp.size(), hipMemcpyDeviceToHost));
    const size_t E = (size_t)rp[S];
    std::vector<int32_t> col(E), rpp((size_t)S + 1, 0), colp(E);
    if (E) HIP_TRY(hipMemcpy(col.data(), c->sta_col, sizeof(int32_t) * E, hipMemcpyDeviceToHost));
    for (int i = 0; i < S; ++i) {          // internal station i = the caller's perm[i]: same neighbours, same edge order, new labels
        const int u = perm[i];
        rpp[(size_t)i + 1] = rpp[i] + (rp[u + 1] - rp[u]);
        for (int e = rp[u]; e < rp[u + 1]; ++e) colp[(size_t)rpp[i] + (e - rp[u])] = inv[col[e]];
    }
    HIP_TRY(hipMalloc((void**)&c->sta_perm, sizeof(int32_t) * S));
    HIP_TRY(hipMalloc((void**)&c->sta_inv, sizeof(int32_t) * S));
    HIP_TRY(hipMalloc((void**)&c->sta_rowptr_p, sizeof(int32_t) * ((size_t)S + 1)));
    HIP_TRY(hipMalloc((void**)&c->sta_col_p, sizeof(int32_t) * std::max<size_t>(E, 1)));
    HIP_TRY(hipMemcpy(c->sta_perm, perm.data(), sizeof(int32_t) * S, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->sta_inv, inv.data(), sizeof(int32_t) * S, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->sta_rowptr_p, rpp.data(), sizeof(int32_t) * rpp.size(), hipMemcpyHostToDevice));
    if (E) HIP_TRY(hipMemcpy(c->sta_col_p, colp.data(), sizeof(int32_t) * E, hipMemcpyHostToDevice));
    return GENIE_OK;
}

int genie_set_static_edge_attr(genie_ctx* c, const float* edge_attr, void* stream) {
    if (!c) return fail(GENIE_ERR_ARG, "genie_set_static_edge_attr: null context");
    c->ea_user = nullptr;
    if (!edge_attr || !c->sta_perm || c->pcsr) return GENIE_OK;       // nothing to prepare without a station processing order
    if (!c->ea_int) HIP_TRY(hipMalloc((void**)&c->ea_int, sizeof(float) * 3 * (size_t)c->P));
    k_permute_sta_rows<<<(unsigned)((c->P * 3 + 255) / 256), 256, 0, (hipStream_t)stream>>>(edge_attr, c->P, 3, c->sta_inv, c->S, c->ea_int);
    HIP_TRY(hipGetLastError());
    c->ea_user = edge_attr;
    return GENIE_OK;
}

int genie_set_slot(genie_ctx* c, int slot) {
    if (!c || slot < 0 || slot >= GENIE_NSLOT) return fail(GENIE_ERR_ARG, "genie_set_slot: slot must be in [0, GENIE_NSLOT)");
    c->slot = slot;
    return GENIE_OK;
}

int genie_ctx_destroy(genie_ctx* c) {
    if (!c) return GENIE_OK;
    (void)hipFree(c->d_packplans);
    for (int s = 7; s < NPLAN; ++s) { (void)hipFree(c->d_steps[s]); (void)hipFree(c->d_bias[s]); (void)hipFree(c->d_scal[s]); (void)hipFree(c->packed[s]); }
    for (int s = 3; s < NTM; ++s) { (void)hipFree(c->d_acc[s]); (void)hipFree(c->d_vec[s]); (void)hipFree(c->d_sc[s]); }
    void* ptrs[] = {c->sta_rowptr, c->sta_col, c->src_rowptr, c->src_col, c->order, c->outdeg, c->raw,
                    c->d_steps[0], c->d_steps[1], c->d_bias[0], c->d_bias[1],
                    c->d_scal[0], c->d_scal[1], c->packed[0], c->packed[1],
                    c->d_steps[2], c->d_steps[3], c->d_bias[2], c->d_bias[3], c->d_scal[2], c->d_scal[3], c->packed[2], c->packed[3],
                    c->d_steps[4], c->d_steps[5], c->d_steps[6], c->d_bias[4], c->d_bias[5], c->d_bias[6], c->d_scal[4], c->d_scal[5],
                    c->d_scal[6], c->packed[4], c->packed[5], c->packed[6], c->d_acc[0], c->d_acc[1], c->d_acc[2], c->d_vec[0],
                    c->d_vec[1], c->d_vec[2], c->d_sc[0], c->d_sc[1], c->d_sc[2],
                    c->as_pg, c->as_ps, c->d_h2tbl, c->packed_h2, c->src_tab,
                    c->mpos_sta, c->mpos_src, c->ebias_sta, c->ebias_src,
                    c->p_sta_rowptr, c->p_sta_col, c->p_src_rowptr, c->p_src_col, c->seg_rowptr, c->abs_sta, c->abs_src,
                    c->r_sta_rowptr, c->r_sta_col, c->r_src_rowptr, c->r_src_col, c->r_sta_w, c->r_src_w,
                    c->sta_perm, c->sta_inv, c->sta_rowptr_p, c->sta_col_p, c->ebias_sta_p, c->ea_int, c->ea_tmp};
    for (void* p : ptrs) (void)hipFree(p);
    delete c;
    return GENIE_OK;
}

int genie_weights_set(genie_ctx* c, const char* name, const float* dev_ptr, int64_t numel, void* stream) {
    if (!c || !name || !dev_ptr) return fail(GENIE_ERR_ARG, "genie_weights_set: null argument");
    for (int i = 0; i < W_COUNT; ++i) {
        if (strcmp(name, g_params[i].name) == 0) {
            if (numel != g_params[i].numel)
                return fail(GENIE_ERR_ARG, std::string("genie_weights_set: wrong numel for ") + name);
            HIP_TRY(hipMemcpyAsync(c->raw + g_params[i].off, dev_ptr, sizeof(float) * numel, hipMemcpyDeviceToDevice,
                                   (hipStream_t)stream));
            c->dirty = true;
            return GENIE_OK;
        }
    }
    return fail(GENIE_ERR_ARG, std::string("genie_weights_set: unknown parameter ") + name);
}

int64_t genie_weights_offset(int i) { init_registry(); return (i >= 0 && i < W_COUNT) ? g_params[i].off : -1; }
int64_t genie_weights_blob_floats(void) { init_registry(); return g_raw_total; }

int genie_weights_set_blob(genie_ctx* c, const float* blob_dev, int64_t n_floats, void* stream) {
    if (!c || !blob_dev) return fail(GENIE_ERR_ARG, "genie_weights_set_blob: null argument");
    if (n_floats != g_raw_total) return fail(GENIE_ERR_ARG, "genie_weights_set_blob: wrong blob size");
    HIP_TRY(hipMemcpyAsync(c->raw, blob_dev, sizeof(float) * n_floats, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    c->dirty = true;
    return GENIE_OK;
}

int genie_weights_commit(genie_ctx* c, void* stream) {
    if (!c) return fail(GENIE_ERR_ARG, "null context");
    return ensure_packed(c, (hipStream_t)stream);
}

size_t genie_workspace_bytes(const genie_ctx* c) { return c ? c->ws_floats * sizeof(float) : 0; }

namespace {
// gi_begin / gi_end: positions of the processing order this call covers (the whole grid: 0, G); do_split: run the input
// split pass over ALL rows (owned + halo) first -- the first range call of a window does, later ones reuse its rows
int run_stage1(genie_ctx* c, const float* slice, const float* mask, float* dbg_h0, float* dbg_h1, void* ws, void* stream,
               int gi_begin, int gi_end, bool do_split) {
    int rc = check_ws(c, ws);
    if (rc) return rc;
    if (!slice || !mask) return fail(GENIE_ERR_ARG, "genie_da_stage1: null input");
    if (gi_begin < 0 || gi_end > c->G || gi_begin > gi_end) return fail(GENIE_ERR_ARG, "genie_da_stage1: bad source-node range");
    const bool whole = gi_begin == 0 && gi_end == c->G;
    if (!whole && c->pcsr) return fail(GENIE_ERR_STATE, "genie_da_stage1_range: not available on an irregular product graph");
    hipStream_t st = (hipStream_t)stream;
    if ((rc = ensure_packed(c, st))) return rc;
    DaArgs a = make_da_args(c, (float*)ws);
    a.gi0 = gi_begin; a.G = gi_end - gi_begin;
    const long long n_tiles = (long long)a.G * c->T;
    a.slice = slice; a.mask = mask; a.packed = c->packed[0];
    a.dbg_h0 = dbg_h0; a.dbg_h1 = dbg_h1;
    float* dbg_tmp = nullptr;
    if ((dbg_h0 || dbg_h1) && sta_order_on(c)) {
        HIP_TRY(hipMalloc((void**)&dbg_tmp, sizeof(float) * 90 * (size_t)c->P));
        if (dbg_h0) a.dbg_h0 = dbg_tmp;
        if (dbg_h1) a.dbg_h1 = dbg_tmp + c->P * 30;
    }
    if (((c->force_generic && !c->use_h2) || abs_generic(c)) && !c->pcsr) {   // use_absolute_pos, training on other graph shapes: generic kernel (64-bit safe, any graph)
        if (n_tiles) k_stage1<<<da_grid(c, n_tiles, c->bpc1), 256, 0, st>>>(a);
    } else if (c->pcsr && c->pcsr_h2) {
        unsigned* xs = (unsigned*)((float*)ws + c->o_xs);
        k_split_rows<<<(unsigned)((c->P + 255) / 256), 256, 0, st>>>(slice, mask, c->P, xs, nullptr, c->S, nullptr);
        a.xs = xs; a.packed = c->packed_h2; a.xs_plane = c->P * (long long)XPC;
        const long long nitems = (c->P + 31) / 32;
        const int grid = (int)std::min<long long>((nitems + H2_THREADS / 64 - 1) / (H2_THREADS / 64), (long long)c->num_cu * c->bpc1b);
        if (c->P * XROW >= (1ll << 32)) k_stage1_h2<8, 15, false, true, false, true><<<grid, H2_THREADS, 0, st>>>(a);
        else k_stage1_h2<8, 15, false, false, false, true><<<grid, H2_THREADS, 0, st>>>(a);
    } else if (c->pcsr) {
        const long long ntiles = (c->P + 15) / 16;
        k_stage1_pcsr<<<(int)std::min<long long>((ntiles + 3) / 4, (long long)c->num_cu * c->bpc1), 256, 0, st>>>(a);
    } else if (c->use_h2) {
        unsigned* xs = (unsigned*)((float*)ws + c->o_xs);
        const bool presplit = (c->xs_slice == slice && c->xs_mask == mask && c->xs_ws == ws) || !do_split;   // genie_embed_window_split, one-shot
        if (do_split) { c->xs_slice = c->xs_mask = nullptr; c->xs_ws = nullptr; }
        float* mmw = (float*)ws + c->o_mm + (c->slot % GENIE_NBIG) * c->big_stride;
        if (presplit && do_split && sta_order_on(c) && c->xs_mm_copy != c->slot % GENIE_NBIG) {
            // the embedding ran under another slot: its message-mask row sits in a different copy than the one stage 2 of THIS
            // window reads (the split rows `xs` exist once). Bring it over (P_ext floats, same stream); callers avoid the copy by
            // selecting the window's slot before they embed (engine.embed_window does).
            HIP_TRY(hipMemcpyAsync(mmw, (const float*)ws + c->o_mm + c->xs_mm_copy * c->big_stride, sizeof(float) * (size_t)c->P_ext,
                                   hipMemcpyDeviceToDevice, st));
        }
        if (!presplit) {
            if (sta_order_on(c) && c->S <= SPLIT_G_MAXS) {
                HIP_TRY(hipFuncSetAttribute((const void*)k_split_rows_g, hipFuncAttributeMaxDynamicSharedMemorySize, SPLIT_G_MAXS * 32));
                k_split_rows_g<<<(unsigned)(c->P_ext / c->S), 256, (size_t)c->S * 32, st>>>(slice, mask, c->S, xs, c->sta_perm, mmw, c->P_ext);
            } else
                k_split_rows<<<(unsigned)((c->P_ext + 255) / 256), 256, 0, st>>>(slice, mask, c->P_ext, xs,
                                                                                sta_order_on(c) ? c->sta_perm : nullptr, c->S, mmw);
        }
        a.xs = xs; a.packed = c->packed_h2; a.xs_plane = c->P_ext * (long long)XPC;
        const int grid = da_grid_w(c, (n_tiles + 1) / 2, c->bpc1b, H2_THREADS / 64);
        const bool big = c->P_ext * XROW >= (1ll << 32);
        if (!n_tiles) {
        } else if (c->abs_sta) {
            if (c->abs_dirty || !c->abs_ts) {
                if (!c->abs_ts) {
                    HIP_TRY(hipMalloc((void**)&c->abs_ts, 16 * (size_t)c->S));
                    HIP_TRY(hipMalloc((void**)&c->abs_tg, 16 * (size_t)c->G_ext));
                }
                k_abs_pieces<<<(c->S + 255) / 256, 256, 0, st>>>(c->abs_sta, sta_order_on(c) ? c->sta_perm : nullptr, c->S, c->abs_ts);
                k_abs_pieces<<<(c->G_ext + 255) / 256, 256, 0, st>>>(c->abs_src, nullptr, c->G_ext, c->abs_tg);
                c->abs_dirty = false;
                a.abs_ts = c->abs_ts; a.abs_tg = c->abs_tg;
            }
            if (big) k_stage1_h2<8, 15, false, true, true><<<grid, H2_THREADS, 0, st>>>(a);
            else k_stage1_h2<8, 15, false, false, true><<<grid, H2_THREADS, 0, st>>>(a);
        } else if (c->has_edges) {
            if (big) k_stage1_h2<8, 15, true, true><<<grid, H2_THREADS, 0, st>>>(a);
            else k_stage1_h2<8, 15, true, false><<<grid, H2_THREADS, 0, st>>>(a);
        } else {
            if (big) k_stage1_h2<8, 15, false, true><<<grid, H2_THREADS, 0, st>>>(a);
            else k_stage1_h2<8, 15, false, false><<<grid, H2_THREADS, 0, st>>>(a);
        }
    } else if (n_tiles)
        k_stage1<<<da_grid(c, n_tiles, c->bpc1), 256, 0, st>>>(a);
    HIP_TRY(hipGetLastError());
    if (dbg_tmp) {     // parity outputs were written in station processing order: back to the caller's order
        if (dbg_h0) k_permute_sta_rows<<<(unsigned)((c->P * 30 + 255) / 256), 256, 0, st>>>(dbg_tmp, c->P, 30, c->sta_perm, c->S, dbg_h0);
        if (dbg_h1) k_permute_sta_rows<<<(unsigned)((c->P * 60 + 255) / 256), 256, 0, st>>>(dbg_tmp + c->P * 30, c->P, 60, c->sta_perm, c->S, dbg_h1);
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipFree(dbg_tmp));
    }
    return GENIE_OK;
}
}  // namespace

int genie_da_stage1(genie_ctx* c, const float* slice, const float* mask, void* ws, void* stream) {
    if (!c) return fail(GENIE_ERR_ARG, "null context");
    return run_stage1(c, slice, mask, nullptr, nullptr, ws, stream, 0, c->G, true);
}

int genie_da_stage1_range(genie_ctx* c, const float* slice, const float* mask, int gi_begin, int gi_end, int first, void* ws,
                          void* stream) {
    if (!c) return fail(GENIE_ERR_ARG, "null context");
    return run_stage1(c, slice, mask, nullptr, nullptr, ws, stream, gi_begin, gi_end, first != 0);
}

int genie_da_stage1_debug(genie_ctx* c, const float* slice, const float* mask, float* h0_out, float* h1_out, void* ws,
                          void* stream) {
    if (!c || !h0_out || !h1_out) return fail(GENIE_ERR_ARG, "genie_da_stage1_debug: null argument");
    return run_stage1(c, slice, mask, h0_out, h1_out, ws, stream, 0, c->G, true);
}

float* genie_ws_v_ptr(const genie_ctx* c, void* ws) {
    return (c && ws) ? (float*)ws + c->o_wv + (c->slot % GENIE_NBIG) * c->big_stride : nullptr;
}
int genie_ws_v_pitch(const genie_ctx* c) { (void)c; return ROWW; }

namespace {
int run_stage2(genie_ctx* c, const float* mask, const float* edge_attr, float* x_latent_out, void* ws, void* stream,
               int gi_begin, int gi_end, const float* slope2 = nullptr, int no_bip = 0);
}

int genie_da_stage2_partials(genie_ctx* c, const float* mask, const float* edge_attr, float* x_latent_out, void* ws,
                             void* stream) {
    if (!c) return fail(GENIE_ERR_ARG, "null context");
    return run_stage2(c, mask, edge_attr, x_latent_out, ws, stream, 0, c->G);
}

int genie_da_stage2_partials_range(genie_ctx* c, const float* mask, const float* edge_attr, float* x_latent_out, int gi_begin,
                                   int gi_end, void* ws, void* stream) {
    if (!c) return fail(GENIE_ERR_ARG, "null context");
    return run_stage2(c, mask, edge_attr, x_latent_out, ws, stream, gi_begin, gi_end);
}

namespace {
int run_stage2(genie_ctx* c, const float* mask, const float* edge_attr, float* x_latent_out, void* ws, void* stream,
               int gi_begin, int gi_end, const float* slope2, int no_bip) {
    int rc = check_ws(c, ws);
    if (rc) return rc;
    if (!mask || !edge_attr) return fail(GENIE_ERR_ARG, "genie_da_stage2_partials: null argument");
    if (gi_begin < 0 || gi_end > c->G || gi_begin > gi_end) return fail(GENIE_ERR_ARG, "genie_da_stage2_partials: bad source-node range");
    if (!(gi_begin == 0 && gi_end == c->G) && c->pcsr)
        return fail(GENIE_ERR_STATE, "genie_da_stage2_partials_range: not available on an irregular product graph");
    hipStream_t st = (hipStream_t)stream;
    if ((rc = ensure_packed(c, st))) return rc;
    DaArgs a = make_da_args(c, (float*)ws);
    a.gi0 = gi_begin; a.G = gi_end - gi_begin;
    a.slope2 = slope2; a.no_bip = no_bip;
    const long long n_tiles = (long long)a.G * c->T;
    if (n_tiles == 0) return GENIE_OK;
    a.mask = mask; a.edge_attr = edge_attr; a.x_latent = x_latent_out; a.packed = c->packed[1];
    a.ea_int = (sta_order_on(c) && c->ea_int && c->ea_user == edge_attr) ? c->ea_int : nullptr;
    a.mm_int = (const float*)ws + c->o_mm + (c->slot % GENIE_NBIG) * c->big_stride;
    if (c->force_generic && !c->pcsr) {
        k_stage2<<<da_grid(c, n_tiles, c->bpc2), 256, 0, st>>>(a);
    } else if (c->pcsr) {
        const long long ntiles = (c->P + 15) / 16;
        k_stage2_pcsr<<<(int)std::min<long long>((ntiles + 3) / 4, (long long)c->num_cu * c->bpc2), 256, 0, st>>>(a);
    } else if (c->use_fast && a.sta_user != nullptr && (!no_bip || x_latent_out != nullptr)) {
        // the production configuration: uniform 8 / 15-degree graphs, station processing order; the static edge_attr is registered
        // (genie_set_static_edge_attr), any other one is brought into processing order here (one extra pass over [P, 3])
        if (!no_bip && a.ea_int == nullptr) {
            if (!c->ea_tmp) HIP_TRY(hipMalloc((void**)&c->ea_tmp, sizeof(float) * 3 * (size_t)c->P));
            k_permute_sta_rows<<<(unsigned)((c->P * 3 + 255) / 256), 256, 0, st>>>(edge_attr, c->P, 3, c->sta_inv, c->S, c->ea_tmp);
            a.ea_int = c->ea_tmp;
        }
        const int grid = da_grid(c, n_tiles, c->bpc2o);
        a.wgmap = no_bip ? 0 : c->s2_wgmap;
        if (no_bip) k_stage2_ord<8, 15, true, true><<<grid, 256, 0, st>>>(a);
        else if (x_latent_out) k_stage2_ord<8, 15, true><<<grid, 256, 0, st>>>(a);
        else k_stage2_ord<8, 15, false><<<grid, 256, 0, st>>>(a);
    }
    else
        k_stage2<<<da_grid(c, n_tiles, c->bpc2), 256, 0, st>>>(a);
    HIP_TRY(hipGetLastError());
    return GENIE_OK;
}
}  // namespace

int genie_bipartite_readout(genie_ctx* c, float* bip_out, void* ws, void* stream) {
    int rc = check_ws(c, ws);
    if (rc) return rc;
    if (!bip_out) return fail(GENIE_ERR_ARG, "genie_bipartite_readout: null output");
    const float* part = (const float*)ws + c->o_part + c->slot * c->slot_stride;
    if (c->pcsr) {     // the messages sit in the c rows (k_stage2_pcsr)
        const int nb = std::min((c->G + NPB - 1) / NPB, c->num_cu * 8);
        k_bip_out_seg<<<nb, 256, 0, (hipStream_t)stream>>>((const float*)ws + c->o_c + (c->slot % GENIE_NBIG) * c->big_stride, c->G, c->seg_rowptr, c->raw,
                                                          g_params[W_BP_FC2_W].off, g_params[W_BP_FC2_B].off, g_params[W_BP_ACT2].off, bip_out);
    } else {
        { int rcp = ensure_packed(c, (hipStream_t)stream); if (rcp) return rcp; }
        k_bip_out_m<<<tl_blocks(c->G, c->num_cu * 2), 256, 0, (hipStream_t)stream>>>(part, c->G, c->T, c->packed[PL_BIP], bip_out, 0, 0);
    }
    HIP_TRY(hipGetLastError());
    return GENIE_OK;
}

int genie_da_stage2_bipartite(genie_ctx* c, const float* mask, const float* edge_attr, float* x_latent_out,
                              float* bip_out, void* ws, void* stream) {
    if (!bip_out) return fail(GENIE_ERR_ARG, "genie_da_stage2_bipartite: null argument");
    int rc = genie_da_stage2_partials(c, mask, edge_attr, x_latent_out, ws, stream);
    if (rc) return rc;
    return genie_bipartite_readout(c, bip_out, ws, stream);
}

namespace {
void sa_fill_layer(const genie_ctx* c, int layer, SaArgs& a) {
    const int base = layer == 1 ? W_SA1_FC1_W : (layer == 2 ? W_SA2_FC1_W : W_SA3_FC1_W);
    a.G = c->G; a.C = layer == 1 ? 15 : 30; a.E = c->E_src;
    a.rowptr = c->src_rowptr; a.col = c->src_col; a.outdeg = c->outdeg;
    a.raw = c->raw; a.scale_rel = c->scale_rel;
    a.fc1_w = g_params[base + 0].off; a.fc1_b = g_params[base + 1].off;
    a.fc2_w = g_params[base + 2].off; a.fc2_b = g_params[base + 3].off;
    a.fg_w = g_params[base + 4].off; a.fg_b = g_params[base + 5].off;
    a.act1 = g_params[base + 6].off; a.act2 = g_params[base + 7].off; a.act3 = g_params[base + 8].off;
    if (layer < 3) {
        const int nb = layer == 1 ? W_SA2_FC1_W : W_SA3_FC1_W;
        a.nx_fc1_w = g_params[nb + 0].off; a.nx_fg_w = g_params[nb + 4].off; a.nx_fg_b = g_params[nb + 5].off;
        a.nx_act3 = g_params[nb + 8].off;
    }
}
int sa_blocks(const genie_ctx* c) { return tl_blocks(c->G, std::min(1024, c->tail_cu_sa)); }

// The pre-pass of `layer` was already produced (by k_sa_pre_m or by the previous layer's NEXT tail) in pj / gpart buffer `cur`;
// with_next emits the next layer's pre-pass into the other buffer.
int sa_launch_layer(genie_ctx* c, int layer, const float* x_in, const float* pos, float* out, float* ws, int cur,
                    bool with_next, hipStream_t st) {
    SaArgs a;
    memset(&a, 0, sizeof(a));
    sa_fill_layer(c, layer, a);
    a.x_in = x_in; a.pos = pos; a.out = out;
    const size_t so = c->slot * c->slot_stride;
    float* pj[2] = {ws + c->o_pj0 + so, ws + c->o_pj1 + so};
    float* gp[2] = {ws + c->o_gpart + so, ws + c->o_gpart + so + 1024 * 8};
    a.pj_in = pj[cur]; a.gpart_in = gp[cur]; a.n_gpart_in = sa_blocks(c);
    a.pj_out = pj[cur ^ 1]; a.gpart_out = gp[cur ^ 1];
    const int nb = sa_blocks(c);
    { int rcp = ensure_packed(c, st); if (rcp) return rcp; }
    a.img = c->packed[PL_SA1 + layer - 1];
    if (layer == 1) {
        if (with_next) k_sa_layer_m<15, true><<<nb, 256, 0, st>>>(a); else k_sa_layer_m<15, false><<<nb, 256, 0, st>>>(a);
    } else {
        if (with_next) k_sa_layer_m<30, true><<<nb, 256, 0, st>>>(a); else k_sa_layer_m<30, false><<<nb, 256, 0, st>>>(a);
    }
    HIP_TRY(hipGetLastError());
    return GENIE_OK;
}
int sa_launch_pre(genie_ctx* c, int layer, const float* x_in, float* ws, int cur, hipStream_t st) {
    SaArgs a;
    memset(&a, 0, sizeof(a));
    sa_fill_layer(c, layer, a);
    a.x_in = x_in;
    a.pj_out = ws + (cur ? c->o_pj1 : c->o_pj0) + c->slot * c->slot_stride;
    a.gpart_out = ws + c->o_gpart + c->slot * c->slot_stride + (cur ? 1024 * 8 : 0);
    const int nb = sa_blocks(c);
    { int rcp = ensure_packed(c, st); if (rcp) return rcp; }
    a.img = c->packed[PL_SA1 + layer - 1];
    if (layer == 1) k_sa_pre_m<15><<<nb, 256, 0, st>>>(a); else k_sa_pre_m<30><<<nb, 256, 0, st>>>(a);
    HIP_TRY(hipGetLastError());
    return GENIE_OK;
}
}  // namespace

int genie_spatial_agg_fwd(genie_ctx* c, int layer, const float* x_in, const float* pos, float* out, void* ws,
                          void* stream) {
    int rc = check_ws(c, ws);
    if (rc) return rc;
    if (layer < 1 || layer > 3) return fail(GENIE_ERR_ARG, "genie_spatial_agg_fwd: layer must be 1, 2 or 3");
    if (!x_in || !pos || !out) return fail(GENIE_ERR_ARG, "genie_spatial_agg_fwd: null argument");
    if (c->G_ext != c->G) return fail(GENIE_ERR_STATE, "genie_spatial_agg_fwd needs an unsharded source graph");
    hipStream_t st = (hipStream_t)stream;
    if ((rc = sa_launch_pre(c, layer, x_in, (float*)ws, 0, st))) return rc;
    return sa_launch_layer(c, layer, x_in, pos, out, (float*)ws, 0, false, st);
}

int genie_spatial_agg3_fwd(genie_ctx* c, const float* x_in15, const float* pos, float* out, void* ws, void* stream) {
    int rc = check_ws(c, ws);
    if (rc) return rc;
    if (!x_in15 || !pos || !out) return fail(GENIE_ERR_ARG, "genie_spatial_agg3_fwd: null argument");
    if (c->G_ext != c->G) return fail(GENIE_ERR_STATE, "genie_spatial_agg3_fwd needs an unsharded source graph");
    hipStream_t st = (hipStream_t)stream;
    float* w = (float*)ws;
    if ((rc = sa_launch_pre(c, 1, x_in15, w, 0, st))) return rc;
    const size_t so = c->slot * c->slot_stride;
    if ((rc = sa_launch_layer(c, 1, x_in15, pos, w + c->o_sa0 + so, w, 0, true, st))) return rc;
    if ((rc = sa_launch_layer(c, 2, w + c->o_sa0 + so, pos, w + c->o_sa1 + so, w, 1, true, st))) return rc;
    return sa_launch_layer(c, 3, w + c->o_sa1 + so, pos, out, w, 0, false, st);
}

int genie_path_fwd(genie_ctx* c, const float* slice, const float* mask, const float* edge_attr, const float* pos,
                   float* x_spatial_out, float* x_latent_out, float* bip_out, void* ws, void* stream) {
    int rc = check_ws(c, ws);
    if (rc) return rc;
    if (!x_spatial_out || !pos) return fail(GENIE_ERR_ARG, "genie_path_fwd: null argument");
    float* w = (float*)ws;
    float* bip = bip_out ? bip_out : w + c->o_bip + c->slot * c->slot_stride;
    if ((rc = genie_da_stage1(c, slice, mask, ws, stream))) return rc;
    if ((rc = genie_da_stage2_bipartite(c, mask, edge_attr, x_latent_out, bip, ws, stream))) return rc;
    return genie_spatial_agg3_fwd(c, bip, pos, x_spatial_out, ws, stream);
}

namespace {
RoArgs make_ro_args(const genie_ctx* c) {
    RoArgs a;
    memset(&a, 0, sizeof(a));
    a.raw = c->raw; a.scale_rel = c->scale_rel; a.scale_t = c->scale_t; a.G = c->G;
    a.o_sd_w = g_params[W_SD_W].off; a.o_sd_b = g_params[W_SD_B].off; a.o_sd_a = g_params[W_SD_ACT].off;
    a.o_q1w = g_params[W_TA_Q1_W].off; a.o_q1b = g_params[W_TA_Q1_B].off; a.o_q2w = g_params[W_TA_Q2_W].off; a.o_q2b = g_params[W_TA_Q2_B].off;
    a.o_c1w = g_params[W_TA_C1_W].off; a.o_c1b = g_params[W_TA_C1_B].off; a.o_c2w = g_params[W_TA_C2_W].off; a.o_c2b = g_params[W_TA_C2_B].off;
    a.o_v1w = g_params[W_TA_V1_W].off; a.o_v1b = g_params[W_TA_V1_B].off; a.o_v2w = g_params[W_TA_V2_W].off; a.o_v2b = g_params[W_TA_V2_B].off;
    a.o_p1w = g_params[W_TA_P1_W].off; a.o_p1b = g_params[W_TA_P1_B].off; a.o_p2w = g_params[W_TA_P2_W].off; a.o_p2b = g_params[W_TA_P2_B].off;
    a.o_a1 = g_params[W_TA_ACT1].off; a.o_a2 = g_params[W_TA_ACT2].off; a.o_a3 = g_params[W_TA_ACT3].off;
    a.o_a4 = g_params[W_TA_ACT4].off; a.o_a5 = g_params[W_TA_ACT5].off;
    a.o_sq_w = g_params[W_SAT_Q_W].off; a.o_sq_b = g_params[W_SAT_Q_B].off; a.o_sc_w = g_params[W_SAT_C_W].off; a.o_sc_b = g_params[W_SAT_C_B].off;
    a.o_sv_w = g_params[W_SAT_V_W].off; a.o_sv_b = g_params[W_SAT_V_B].off; a.o_sp_w = g_params[W_SAT_P_W].off; a.o_sp_b = g_params[W_SAT_P_B].off;
    a.o_sa1 = g_params[W_SAT_ACT1].off; a.o_sa2 = g_params[W_SAT_ACT2].off;
    return a;
}
}  // namespace

int genie_readout_grid(genie_ctx* c, const float* x_spatial, const float* t_query, int n_t, float* y_out, void* stream) {
    if (!c || !x_spatial || !t_query || !y_out) return fail(GENIE_ERR_ARG, "genie_readout_grid: null argument");
    if (n_t < 1 || n_t > RO_TMAX) return fail(GENIE_ERR_ARG, "genie_readout_grid: 1 <= n_t <= 10 required");
    RoArgs a = make_ro_args(c);
    { int rcp = ensure_packed(c, (hipStream_t)stream); if (rcp) return rcp; }
    a.N = a.Nw = c->G; a.T = n_t; a.x_spatial = x_spatial; a.t_query = t_query; a.out = y_out;
    a.img = c->packed[PL_RO0];
    HIP_TRY(hipFuncSetAttribute((const void*)k_readout_m<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(float) * (ROM_LDS_FLOATS + GP_IMG_FLOATS))));
    k_readout_m<0><<<tl_blocks(a.N, c->tail_cu_ro), 256, sizeof(float) * ROM_LDS_FLOATS, (hipStream_t)stream>>>(a);
    HIP_TRY(hipGetLastError());
    return GENIE_OK;
}

namespace {
// x = TemporalAttention(SpatialAttention(x_spatial, x_query, x_grid)) per query, and / or the SpatialAttention output itself
int readout_query_impl(genie_ctx* c, const float* x_spatial, const float* x_grid, const float* x_query, const int32_t* knn,
                       int n_query, int k, const float* t_query, int n_t, float* x_out, float* lat_out, void* ws, void* stream) {
    { int rcw = check_ws(c, ws); if (rcw) return rcw; }
    if (!x_spatial || !x_grid || !x_query || !knn || !t_query || !x_out) return fail(GENIE_ERR_ARG, "genie_readout_query: null argument");
    if (k != RO_K) return fail(GENIE_ERR_ARG, "genie_readout_query: k must be 10 (module.py:280)");
    if (n_t < 1 || n_t > RO_TMAX) return fail(GENIE_ERR_ARG, "genie_readout_query: 1 <= n_t <= 10 required");
    if (n_query < 1) return fail(GENIE_ERR_ARG, "genie_readout_query: n_query < 1");
    RoArgs a = make_ro_args(c);
    a.N = a.Nw = n_query; a.T = n_t; a.x_spatial = x_spatial; a.x_grid = x_grid; a.x_query = x_query; a.knn = knn;
    a.t_query = t_query; a.out = x_out; a.lat_out = lat_out;
    { int rcp = ensure_packed(c, (hipStream_t)stream); if (rcp) return rcp; }
    float* cvbuf = (float*)ws + c->o_cv + c->slot * c->slot_stride;
    a.cv = cvbuf;
    k_ro_pre_m<<<tl_blocks(c->G, c->tail_cu_ro), 256, 0, (hipStream_t)stream>>>(x_spatial, c->G, c->packed[PL_ROP], cvbuf, c->G, 0);
    a.img = c->packed[PL_RO1];
    HIP_TRY(hipFuncSetAttribute((const void*)k_readout_m<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(float) * ROM_LDS_FLOATS)));
    k_readout_m<1><<<tl_blocks(a.N, c->tail_cu_ro), 256, sizeof(float) * ROM_LDS_FLOATS, (hipStream_t)stream>>>(a);
    HIP_TRY(hipGetLastError());
    return GENIE_OK;
}
}  // namespace

int genie_readout_query(genie_ctx* c, const float* x_spatial, const float* x_grid, const float* x_query, const int32_t* knn,
                        int n_query, int k, const float* t_query, int n_t, float* x_out, void* ws, void* stream) {
    return readout_query_impl(c, x_spatial, x_grid, x_query, knn, n_query, k, t_query, n_t, x_out, nullptr, ws, stream);
}

// The latent inputs of TemporalAttention next to the read-outs (the 4-output forward needs them, module.py:978-981):
// y_latent = SpatialDirect(x_spatial) [n_grid, 30] and / or SpatialAttention(x_spatial, x_query, x_grid) [n_query, 30].
int genie_readout_grid_latent(genie_ctx* c, const float* x_spatial, const float* t_query, int n_t, float* y_out, float* y_latent_out,
                              void* stream) {
    if (!c || !x_spatial || !t_query || !y_out || !y_latent_out) return fail(GENIE_ERR_ARG, "genie_readout_grid_latent: null argument");
    if (n_t < 1 || n_t > RO_TMAX) return fail(GENIE_ERR_ARG, "genie_readout_grid_latent: 1 <= n_t <= 10 required");
    RoArgs a = make_ro_args(c);
    { int rcp = ensure_packed(c, (hipStream_t)stream); if (rcp) return rcp; }
    a.N = a.Nw = c->G; a.T = n_t; a.x_spatial = x_spatial; a.t_query = t_query; a.out = y_out; a.lat_out = y_latent_out;
    a.img = c->packed[PL_RO0];
    HIP_TRY(hipFuncSetAttribute((const void*)k_readout_m<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(float) * (ROM_LDS_FLOATS + GP_IMG_FLOATS))));
    k_readout_m<0><<<tl_blocks(a.N, c->tail_cu_ro), 256, sizeof(float) * ROM_LDS_FLOATS, (hipStream_t)stream>>>(a);
    HIP_TRY(hipGetLastError());
    return GENIE_OK;
}
int genie_readout_query_latent(genie_ctx* c, const float* x_spatial, const float* x_grid, const float* x_query, const int32_t* knn,
                               int n_query, int k, const float* t_query, int n_t, float* x_out, float* latent_out, void* ws, void* stream) {
    if (!latent_out) return fail(GENIE_ERR_ARG, "genie_readout_query_latent: null argument");
    return readout_query_impl(c, x_spatial, x_grid, x_query, knn, n_query, k, t_query, n_t, x_out, latent_out, ws, stream);
}

// The G-sized tail of `nwin` windows in one set of launches (Bipartite read-out, SpatialAggregation x3, both read-out heads).
// The tail kernels are latency-bound at G = 10k nodes and, next to the persistent P-sized kernels of later windows, only
// advance when those retire workgroups -- every launch costs the P-sized kernels about one of its own durations. Batched, the
// fixed costs (weight images into LDS, launch, drain) are paid once per nwin windows. Window w uses workspace slot
// slot0 + w (where genie_da_stage2_partials left its partials); results are bit-identical to the per-window calls.
int genie_tail_batched(genie_ctx* c, int slot0, int nwin, const float* pos, const float* x_query, const int32_t* knn, int n_query,
                       int k, const float* t_query, int n_t, float* x_spatial_out, float* y_out, float* x_out, void* ws,
                       void* stream) {
    int rc = check_ws(c, ws);
    if (rc) return rc;
    if (!pos || !t_query || !x_spatial_out || !y_out) return fail(GENIE_ERR_ARG, "genie_tail_batched: null argument");
    if (nwin < 1 || slot0 < 0 || slot0 + nwin > GENIE_NSLOT) return fail(GENIE_ERR_ARG, "genie_tail_batched: windows must fit slots [0, 33)");
    if (n_t < 1 || n_t > RO_TMAX) return fail(GENIE_ERR_ARG, "genie_tail_batched: 1 <= n_t <= 10 required");
    if (x_out && (!x_query || !knn || n_query < 1 || k != RO_K)) return fail(GENIE_ERR_ARG, "genie_tail_batched: bad query arguments");
    if (c->pcsr) return fail(GENIE_ERR_STATE, "genie_tail_batched: not available with use_subgraph product graphs");
    if (c->G_ext != c->G) return fail(GENIE_ERR_STATE, "genie_tail_batched needs an unsharded source graph");
    if ((long long)nwin * std::max(c->G, n_query) > 0x7fffffffLL) return fail(GENIE_ERR_ARG, "genie_tail_batched: batch too large");
    { int rcp = ensure_packed(c, (hipStream_t)stream); if (rcp) return rcp; }
    hipStream_t st = (hipStream_t)stream;
    float* w = (float*)ws;
    const long long ss = (long long)c->slot_stride;
    const size_t so = (size_t)slot0 * c->slot_stride;
    // Bipartite read-out -> bip[slot] and the pre-pass of SpatialAggregation1, one launch
    const int nbx = tl_blocks(c->G, std::min(1024, std::max(32, c->num_cu * 2 / nwin)));
    float* pj[2] = {w + c->o_pj0 + so, w + c->o_pj1 + so};
    float* gp[2] = {w + c->o_gpart + so, w + c->o_gpart + so + 1024 * 8};
    const dim3 grid(nbx, nwin);
    {
        SaArgs a;
        memset(&a, 0, sizeof(a));
        sa_fill_layer(c, 1, a);
        a.out = w + c->o_bip + so; a.ws_out = ss; a.ws_slot = ss;
        a.pj_out = pj[0]; a.gpart_out = gp[0];
        a.img = c->packed[PL_SA1];
        k_bip_pre_m<<<grid, 256, 0, st>>>(w + c->o_part + so, c->T, c->packed[PL_BIP], ss, a);
    }
    // SpatialAggregation x3: bip -> sa0 -> sa1 -> x_spatial_out [nwin, G, 30]
    for (int layer = 1; layer <= 3; ++layer) {
        SaArgs a;
        memset(&a, 0, sizeof(a));
        sa_fill_layer(c, layer, a);
        const int cur = (layer - 1) & 1;
        a.pos = pos; a.ws_slot = ss;
        a.x_in = layer == 1 ? w + c->o_bip + so : (layer == 2 ? w + c->o_sa0 + so : w + c->o_sa1 + so);
        a.ws_x_in = ss;
        a.out = layer == 1 ? w + c->o_sa0 + so : (layer == 2 ? w + c->o_sa1 + so : x_spatial_out);
        a.ws_out = layer == 3 ? (long long)c->G * 30 : ss;
        a.pj_in = pj[cur]; a.gpart_in = gp[cur]; a.n_gpart_in = nbx;
        a.pj_out = pj[cur ^ 1]; a.gpart_out = gp[cur ^ 1];
        a.img = c->packed[PL_SA1 + layer - 1];
        if (layer == 1) k_sa_layer_m<15, true><<<grid, 256, 0, st>>>(a);
        else if (layer == 2) k_sa_layer_m<30, true><<<grid, 256, 0, st>>>(a);
        else k_sa_layer_m<30, false><<<grid, 256, 0, st>>>(a);
    }
    // read-out heads over the nwin * G grid nodes / nwin * Q queries
    RoArgs a = make_ro_args(c);
    a.T = n_t; a.x_spatial = x_spatial_out; a.t_query = t_query;
    {   // the grid read-out also leaves the per-grid-node table cv of the query read-out (k_ro_pre_m's work)
        RoArgs g = a;
        g.N = nwin * c->G; g.Nw = c->G; g.out = y_out; g.img = c->packed[PL_RO0];
        size_t lds = sizeof(float) * ROM_LDS_FLOATS;
        if (x_out) { g.cv_out = w + c->o_cv + so; g.cv_ws = ss; g.pimg = c->packed[PL_ROP]; lds += sizeof(float) * GP_IMG_FLOATS; }
        HIP_TRY(hipFuncSetAttribute((const void*)k_readout_m<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(float) * (ROM_LDS_FLOATS + GP_IMG_FLOATS))));
        k_readout_m<0><<<tl_blocks(g.N, c->tail_cu_ro), 256, lds, st>>>(g);
    }
    if (x_out) {
        RoArgs q = a;
        q.N = nwin * n_query; q.Nw = n_query; q.x_grid = pos; q.x_query = x_query; q.knn = knn; q.out = x_out;
        q.img = c->packed[PL_RO1]; q.cv = w + c->o_cv + so; q.cv_ws = ss;
        HIP_TRY(hipFuncSetAttribute((const void*)k_readout_m<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(float) * ROM_LDS_FLOATS)));
        k_readout_m<1><<<tl_blocks(q.N, c->tail_cu_ro), 256, sizeof(float) * ROM_LDS_FLOATS, st>>>(q);
    }
    HIP_TRY(hipGetLastError());
    return GENIE_OK;
}

int genie_embed_ntime(double t0, double max_t, double kernel_sig_t, double dt) {
    // len(np.arange(t0 - 3 sigma, t0 + max_t + 3 sigma + dt, dt))                        process_utils.py:499-504
    const double start = t0 - 3.0 * kernel_sig_t, stop = t0 + max_t + 3.0 * kernel_sig_t + dt;
    return (int)ceil((stop - start) / dt);
}

namespace {
int embed_window_impl(genie_ctx* c, const double* pick_t, const int32_t* pick_sta, const int32_t* pick_phase, int n_picks,
                      double t0, double max_t, double kernel_sig_t, double dt, const float* trv, float* emb_ws,
                      float* slice_out, float* mask_out, unsigned* xs, void* stream);
}

int genie_embed_window(genie_ctx* c, const double* pick_t, const int32_t* pick_sta, const int32_t* pick_phase, int n_picks,
                       double t0, double max_t, double kernel_sig_t, double dt, const float* trv, float* emb_ws,
                       float* slice_out, float* mask_out, void* stream) {
    return embed_window_impl(c, pick_t, pick_sta, pick_phase, n_picks, t0, max_t, kernel_sig_t, dt, trv, emb_ws, slice_out, mask_out,
                             nullptr, stream);
}

int genie_embed_window_split(genie_ctx* c, const double* pick_t, const int32_t* pick_sta, const int32_t* pick_phase, int n_picks,
                             double t0, double max_t, double kernel_sig_t, double dt, const float* trv, float* emb_ws,
                             float* slice_out, float* mask_out, void* ws, void* stream) {
    int rc = check_ws(c, ws);
    if (rc) return rc;
    unsigned* xs = c->use_h2 ? (unsigned*)((float*)ws + c->o_xs) : nullptr;
    rc = embed_window_impl(c, pick_t, pick_sta, pick_phase, n_picks, t0, max_t, kernel_sig_t, dt, trv, emb_ws, slice_out, mask_out, xs,
                           stream);
    if (rc == GENIE_OK && xs) { c->xs_slice = slice_out; c->xs_mask = mask_out; c->xs_ws = ws; c->xs_mm_copy = c->slot % GENIE_NBIG; }
    return rc;
}

namespace {
int embed_window_impl(genie_ctx* c, const double* pick_t, const int32_t* pick_sta, const int32_t* pick_phase, int n_picks,
                      double t0, double max_t, double kernel_sig_t, double dt, const float* trv, float* emb_ws,
                      float* slice_out, float* mask_out, unsigned* xs, void* stream) {
    if (!c || !trv || !emb_ws || !slice_out || !mask_out) return fail(GENIE_ERR_ARG, "genie_embed_window: null argument");
    if (n_picks > 0 && (!pick_t || !pick_sta || !pick_phase)) return fail(GENIE_ERR_ARG, "genie_embed_window: null pick array");
    if (!(dt > 0.0) || !(kernel_sig_t > 0.0) || !(max_t > 0.0)) return fail(GENIE_ERR_ARG, "genie_embed_window: bad dt / sigma / max_t");
    if (c->pcsr) return fail(GENIE_ERR_STATE, "genie_embed_window: not available on an irregular product graph");
    hipStream_t st = (hipStream_t)stream;
    EmbArgs a;
    memset(&a, 0, sizeof(a));
    a.pick_t = pick_t; a.pick_sta = pick_sta; a.pick_phase = pick_phase; a.n_picks = n_picks;
    a.S = c->S; a.t0 = t0; a.tref0 = t0 - 3.0 * kernel_sig_t; a.dt = dt; a.sigma = kernel_sig_t;
    a.n_time = genie_embed_ntime(t0, max_t, kernel_sig_t, dt);
    a.n_extra = (int)ceil(3.0 * kernel_sig_t / dt);                                       // process_utils.py:518
    a.emb = emb_ws; a.trv = trv; a.rows = c->P_ext; a.slice = slice_out; a.mask = mask_out; a.xs = xs;
    a.sta_inv = (xs && sta_order_on(c)) ? c->sta_inv : nullptr;
    a.mm = xs ? (float*)xs - c->o_xs + c->o_mm + (c->slot % GENIE_NBIG) * c->big_stride : nullptr;   // xs = workspace + o_xs
    HIP_TRY(hipMemsetAsync(emb_ws, 0, sizeof(float) * 2 * (size_t)a.S * a.n_time, st));
    if (n_picks > 0) {
        const long long n = (long long)n_picks * (2 * a.n_extra + 1);
        k_embed_scatter<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(a);
    }
    k_embed_edges<<<(2 * a.S + 255) / 256, 256, 0, st>>>(a);
    k_embed_gather<<<(unsigned)((a.rows + 255) / 256), 256, 0, st>>>(a);
    HIP_TRY(hipGetLastError());
    return GENIE_OK;
}
}  // namespace


int genie_nbr_mean(genie_ctx* c, const float* x_sta, const float* x_src, float* out_sta, float* out_src, int row_floats,
                   void* stream) {
    if (!c) return fail(GENIE_ERR_ARG, "genie_nbr_mean: null context");
    if ((x_sta && !out_sta) || (x_src && !out_src)) return fail(GENIE_ERR_ARG, "genie_nbr_mean: input without output");
    if (c->pcsr) return fail(GENIE_ERR_STATE, "genie_nbr_mean: not available on an irregular product graph");
    if (!x_sta && !x_src) return GENIE_OK;
    const int nb = std::min<long long>((c->P + 31) / 32, (long long)c->num_cu * 16);
    hipStream_t st = (hipStream_t)stream;
#define GENIE_NM(CL_, VW_) k_nbr_mean<CL_, VW_><<<nb, 256, 0, st>>>(c->S, c->G, c->sta_rowptr, c->sta_col, c->src_rowptr, c->src_col, x_sta, x_src, out_sta, out_src)
    switch (row_floats) {
        case 16: GENIE_NM(4, 4); break;
        case 32: GENIE_NM(8, 4); break;
        case 30: GENIE_NM(15, 2); break;
        default: return fail(GENIE_ERR_ARG, "genie_nbr_mean: row_floats must be 16, 30 or 32");
    }
#undef GENIE_NM
    HIP_TRY(hipGetLastError());
    return GENIE_OK;
}

namespace {
// out-edge CSR of a graph given as in-edge CSR (rowptr by target i, col = source j): for every j the targets i in increasing
// order, with weight 1 / in-degree(i)
int build_reversed(const int32_t* d_rowptr, const int32_t* d_col, int n_tgt, int n_src, int32_t** r_rowptr, int32_t** r_col, float** r_w) {
    std::vector<int32_t> rp((size_t)n_tgt + 1);
    HIP_TRY(hipMemcpy(rp.data(), d_rowptr, sizeof(int32_t) * rp.size(), hipMemcpyDeviceToHost));
    const size_t E = (size_t)rp[n_tgt];
    std::vector<int32_t> col(E);
    if (E) HIP_TRY(hipMemcpy(col.data(), d_col, sizeof(int32_t) * E, hipMemcpyDeviceToHost));
    std::vector<int32_t> rrp((size_t)n_src + 1, 0), rcol(E);
    std::vector<float> rw(E);
    for (size_t e = 0; e < E; ++e) {
        if (col[e] < 0 || col[e] >= n_src) return fail(GENIE_ERR_ARG, "neighbour id out of range");
        ++rrp[(size_t)col[e] + 1];
    }
    for (int j = 0; j < n_src; ++j) rrp[(size_t)j + 1] += rrp[j];
    std::vector<int32_t> fill(rrp.begin(), rrp.end() - 1);
    for (int i = 0; i < n_tgt; ++i) {
        const float w = 1.f / (float)std::max(1, rp[i + 1] - rp[i]);
        for (int e = rp[i]; e < rp[i + 1]; ++e) {
            const int32_t pos = fill[col[e]]++;
            rcol[pos] = i;
            rw[pos] = w;
        }
    }
    HIP_TRY(hipMalloc((void**)r_rowptr, sizeof(int32_t) * rrp.size()));
    HIP_TRY(hipMemcpy(*r_rowptr, rrp.data(), sizeof(int32_t) * rrp.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc((void**)r_col, sizeof(int32_t) * std::max<size_t>(E, 1)));
    HIP_TRY(hipMalloc((void**)r_w, sizeof(float) * std::max<size_t>(E, 1)));
    if (E) {
        HIP_TRY(hipMemcpy(*r_col, rcol.data(), sizeof(int32_t) * E, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(*r_w, rw.data(), sizeof(float) * E, hipMemcpyHostToDevice));
    }
    return GENIE_OK;
}
}  // namespace

int genie_nbr_mean_bwd(genie_ctx* c, const float* g_sta, const float* g_src, float* dx_sta, float* dx_src, int row_floats,
                       void* stream) {
    if (!c) return fail(GENIE_ERR_ARG, "genie_nbr_mean_bwd: null context");
    if ((g_sta && !dx_sta) || (g_src && !dx_src)) return fail(GENIE_ERR_ARG, "genie_nbr_mean_bwd: input without output");
    if (c->pcsr || c->G_ext != c->G) return fail(GENIE_ERR_STATE, "genie_nbr_mean_bwd: needs an unsharded Cartesian product graph");
    if (!g_sta && !g_src) return GENIE_OK;
    if (!c->r_sta_rowptr) {
        int rc;
        if ((rc = build_reversed(c->sta_rowptr, c->sta_col, c->S, c->S, &c->r_sta_rowptr, &c->r_sta_col, &c->r_sta_w))) return rc;
        if ((rc = build_reversed(c->src_rowptr, c->src_col, c->G, c->G, &c->r_src_rowptr, &c->r_src_col, &c->r_src_w))) return rc;
    }
    const int nb = std::min<long long>((c->P + 31) / 32, (long long)c->num_cu * 16);
    hipStream_t st = (hipStream_t)stream;
#define GENIE_NMB(CL_, VW_) k_nbr_mean<CL_, VW_><<<nb, 256, 0, st>>>(c->S, c->G, c->r_sta_rowptr, c->r_sta_col, c->r_src_rowptr, c->r_src_col, \
                                                            g_sta, g_src, dx_sta, dx_src, c->r_sta_w, c->r_src_w)
    switch (row_floats) {
        case 16: GENIE_NMB(4, 4); break;
        case 32: GENIE_NMB(8, 4); break;
        case 30: GENIE_NMB(15, 2); break;
        default: return fail(GENIE_ERR_ARG, "genie_nbr_mean_bwd: row_floats must be 16, 30 or 32");
    }
#undef GENIE_NMB
    HIP_TRY(hipGetLastError());
    return GENIE_OK;
}

int genie_prelu_bwd(const float* x, const float* dy, const float* slope, int64_t n, float* dx, float* dslope, float* scratch,
                    void* stream) {
    if (!slope || !dslope || !scratch || n < 0 || (n > 0 && (!x || !dy || !dx))) return fail(GENIE_ERR_ARG, "genie_prelu_bwd: bad argument");
    if ((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 15) != 0) return fail(GENIE_ERR_ARG, "genie_prelu_bwd: pointers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    k_prelu_bwd<<<PRELU_BLOCKS, 256, 0, st>>>(x, dy, slope, n, dx, scratch);
    k_prelu_bwd_sum<<<1, 256, 0, st>>>(scratch, PRELU_BLOCKS, dslope);
    HIP_TRY(hipGetLastError());
    return GENIE_OK;
}

int64_t genie_linear_bwd_scratch_floats(int K) { return (int64_t)LBW_BLOCKS * (32 * 64 * ((K + 63) / 64) + 32); }

int genie_linear_bwd_wb(const float* x, const float* dy, int64_t N, int K, int M, float* dW, float* db, float* scratch, void* stream) {
    if (!x || !dy || !dW || !scratch || N <= 0 || K <= 0 || M <= 0) return fail(GENIE_ERR_ARG, "genie_linear_bwd_wb: bad argument");
    if (M > 128 || K > 128) return fail(GENIE_ERR_ARG, "genie_linear_bwd_wb: supports M <= 128 outputs and K <= 128 inputs");
    hipStream_t st = (hipStream_t)stream;
    const int KC = (K + 63) / 64;
    const int nb = (int)std::min<int64_t>(LBW_BLOCKS, (N + LBW_ROWS - 1) / LBW_ROWS);
    const int per = 32 * 64 * KC + 32;
    for (int m0 = 0; m0 < M; m0 += 32) {       // 32 output columns per pass (x is re-read: the wide layers have few rows)
        const int mc = std::min(32, M - m0);
        if (KC == 1) k_linear_bwd_w<1><<<nb, 256, 0, st>>>(x, dy + m0, N, K, mc, M, scratch);
        else k_linear_bwd_w<2><<<nb, 256, 0, st>>>(x, dy + m0, N, K, mc, M, scratch);
        k_linear_bwd_sum<<<(per + 31) / 32, 256, 0, st>>>(scratch, nb, KC, K, mc, dW + (size_t)m0 * K, db ? db + m0 : nullptr);
    }
    HIP_TRY(hipGetLastError());
    return GENIE_OK;
}

namespace {
int train_grid(const genie_ctx* c) { return std::max(8, c->num_cu * 2 / 8 * 8); }      // 2 workgroups per CU (1: +10 %, 3: +6 %, 4: +1 % step time)
size_t train_part_floats(const genie_ctx* c) { return (size_t)train_grid(c) * 4 * (30 * 256 + 12 * 16 + 16); }
int train_check(const genie_ctx* c, const char* who) {
    if (c->pcsr || c->G_ext != c->G) return fail(GENIE_ERR_STATE, std::string(who) + ": needs an unsharded Cartesian product graph");
    if (c->has_edges || c->abs_sta) return fail(GENIE_ERR_STATE, std::string(who) + ": default model definition only");
    return GENIE_OK;
}
}  // namespace

size_t genie_train_save_floats(const genie_ctx* c) { return c ? (size_t)SV_BLOCKS * 16 * (size_t)c->P : 0; }
size_t genie_train_scratch_floats(const genie_ctx* c) { return c ? (size_t)GR_BLOCKS * 16 * (size_t)c->P + train_part_floats(c) : 0; }

int genie_da_train_fwd(genie_ctx* c, const float* slice, const float* mask, const float* edge_attr, float* save,
                       float* x_latent_out, float* r_out, void* ws, void* stream) {
    int rc = check_ws(c, ws);
    if (rc) return rc;
    if (!slice || !mask || !edge_attr || !save || !r_out) return fail(GENIE_ERR_ARG, "genie_da_train_fwd: null argument");
    if ((rc = train_check(c, "genie_da_train_fwd"))) return rc;
    c->force_generic = 1; c->train_save = save;
    rc = run_stage1(c, slice, mask, nullptr, nullptr, ws, stream, 0, c->G, true);
    if (!rc) rc = run_stage2(c, mask, edge_attr, x_latent_out, ws, stream, 0, c->G);
    c->force_generic = 0; c->train_save = nullptr;
    if (rc) return rc;
    const float* part = (const float*)ws + c->o_part + c->slot * c->slot_stride;
    k_part_sum<<<(c->G * 30 + 255) / 256, 256, 0, (hipStream_t)stream>>>(part, c->G, c->T, r_out);
    HIP_TRY(hipGetLastError());
    return GENIE_OK;
}

namespace {
int ensure_reversed(genie_ctx* c) {
    if (c->r_sta_rowptr) return GENIE_OK;
    int rc;
    if ((rc = build_reversed(c->sta_rowptr, c->sta_col, c->S, c->S, &c->r_sta_rowptr, &c->r_sta_col, &c->r_sta_w))) return rc;
    return build_reversed(c->src_rowptr, c->src_col, c->G, c->G, &c->r_src_rowptr, &c->r_src_col, &c->r_src_w);
}
int da_train_bwd_impl(genie_ctx* c, const float* slice, const float* mask, const float* edge_attr, const float* save,
                      const float* d_r, float* scratch, float* grad_blob, void* stream, bool zero_blob);
}  // namespace

int genie_da_train_bwd(genie_ctx* c, const float* slice, const float* mask, const float* edge_attr, const float* save,
                       const float* d_r, float* scratch, float* grad_blob, void* stream) {
    return da_train_bwd_impl(c, slice, mask, edge_attr, save, d_r, scratch, grad_blob, stream, true);
}

namespace {
int da_train_bwd_impl(genie_ctx* c, const float* slice, const float* mask, const float* edge_attr, const float* save,
                      const float* d_r, float* scratch, float* grad_blob, void* stream, bool zero_blob) {
    if (!c || !slice || !mask || !edge_attr || !save || !d_r || !scratch || !grad_blob)
        return fail(GENIE_ERR_ARG, "genie_da_train_bwd: null argument");
    int rc;
    if ((rc = train_check(c, "genie_da_train_bwd"))) return rc;
    hipStream_t st = (hipStream_t)stream;
    if ((rc = ensure_packed(c, st))) return rc;
    if ((rc = ensure_reversed(c))) return rc;
    if (zero_blob) HIP_TRY(hipMemsetAsync(grad_blob, 0, sizeof(float) * g_raw_total, st));
    TrArgs a;
    memset(&a, 0, sizeof(a));
    a.S = c->S; a.G = c->G; a.T = c->T; a.seg = std::max(1, c->seg);
    a.nxcd = 8;
    a.P = c->P; a.order = c->order;
    a.r_sta_rowptr = c->r_sta_rowptr; a.r_sta_col = c->r_sta_col; a.r_sta_w = c->r_sta_w;
    a.r_src_rowptr = c->r_src_rowptr; a.r_src_col = c->r_src_col; a.r_src_w = c->r_src_w;
    a.slice = slice; a.mask = mask; a.edge_attr = edge_attr; a.save = save; a.dr = d_r;
    a.gr = scratch; a.part = scratch + (size_t)GR_BLOCKS * 16 * (size_t)c->P;
    a.sv_t = SV_T; a.sv_up = SV_UP; a.sv_vp = SV_VP;
    const int grid = train_grid(c), n_waves = grid * 4;
    for (int s = 0; s < 3; ++s) {
        a.packed = c->packed[4 + s]; a.n_acc = c->n_acc[s]; a.n_vec = c->n_vec[s];
        if (s == 0) k_train_b2<<<grid, 256, 0, st>>>(a);
        else if (s == 1) k_train_b1<false><<<grid, 256, 0, st>>>(a);
        else k_train_b0<<<grid, 256, 0, st>>>(a);
        const int stride = a.n_acc * 256 + a.n_vec * 16 + 16;
        k_train_reduce<<<(stride + 31) / 32, 256, 0, st>>>(a.part, n_waves, a.n_acc, a.n_vec, c->n_sc[s], c->d_acc[s], c->d_vec[s],
                                                            c->d_sc[s], grad_blob, 0);
    }
    HIP_TRY(hipGetLastError());
    return GENIE_OK;
}
}  // namespace

// ---- training step: the G- / Q-sized tail (train_tail_kernels.hpp) -------------------------------------------------------
namespace {
constexpr int TT_R = 0, TT_BIP = 32, TT_SA1 = 48, TT_SA2 = 80, TT_XS = 112, TT_ROW = 144;     // floats per source node in `tsave`
int tt_grid(long long n) { return (int)std::max<long long>(1, std::min<long long>((n + 63) / 64, 160)); }
size_t tt_part_floats(int n_acc, int n_vec, int grid) { return (size_t)grid * 4 * ((size_t)n_acc * 256 + (size_t)n_vec * 16 + 16); }
struct TtScratch {           // offsets (floats) into the backward's scratch
    size_t dxs_a, dxs_b, eb, dxm, cv, pj, gpart, dxd, dan, dx0, dx1, part_ro, part_a, part_b, total;
};
TtScratch tt_layout(const genie_ctx* c, int n_query) {
    TtScratch t;
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o = align_up(o + n, 64); return r; };
    const size_t G = (size_t)c->G, Q = (size_t)std::max(1, n_query);
    t.dxs_a = take(G * 32); t.dxs_b = take(G * 32);
    t.eb = take(Q * RO_K * 12); t.dxm = take(Q * 16);
    t.cv = take(G * CVP);
    t.pj = take(G * 32); t.gpart = take(1024 * 8);
    t.dxd = take(G * 32); t.dan = take(G * 32); t.dx0 = take(G * 32); t.dx1 = take(G * 32);
    t.part_ro = take(tt_part_floats(RB_NACC1, RB_NVEC, tt_grid(std::max(G, Q))));
    t.part_a = take(tt_part_floats(GTN_GROUPS, 10, tt_grid(G)));      // also k_sat_node_bwd / k_bip_bwd (largest of the G-sized maps)
    t.part_b = take(tt_part_floats(SBA_NACC, SBA_NVEC, tt_grid(G)));
    t.total = o;
    return t;
}
int tt_reduce(genie_ctx* c, int tm, const float* part, int n_waves, float* blob, hipStream_t st) {
    const int stride = c->n_acc[tm] * 256 + c->n_vec[tm] * 16 + 16;
    k_train_reduce<<<(stride + 31) / 32, 256, 0, st>>>(part, n_waves, c->n_acc[tm], c->n_vec[tm], c->n_sc[tm], c->d_acc[tm], c->d_vec[tm],
                                                        c->d_sc[tm], blob, 1);
    return GENIE_OK;
}
}  // namespace

size_t genie_tail_train_save_floats(const genie_ctx* c) { return c ? (size_t)TT_ROW * (size_t)c->G : 0; }
size_t genie_tail_train_scratch_floats(const genie_ctx* c, int n_query) { return c ? tt_layout(c, n_query).total : 0; }
size_t genie_train_grad_floats(void) { init_registry(); return (size_t)g_raw_total + (size_t)TQ_ROWS * 75 + 16; }

int genie_tail_train_fwd(genie_ctx* c, const float* pos, const float* x_query, const int32_t* knn, int n_query, int k,
                         const float* t_query, int n_t, float* tsave, float* y_latent_out, float* y_out, float* x_out, void* ws,
                         void* stream) {
    int rc = check_ws(c, ws);
    if (rc) return rc;
    if (!pos || !x_query || !knn || !t_query || !tsave || !y_out || !x_out) return fail(GENIE_ERR_ARG, "genie_tail_train_fwd: null argument");
    if ((rc = train_check(c, "genie_tail_train_fwd"))) return rc;
    if (k != RO_K || n_query < 1 || n_t < 1 || n_t > RO_TMAX) return fail(GENIE_ERR_ARG, "genie_tail_train_fwd: k = 10, n_query >= 1, 1 <= n_t <= 10");
    hipStream_t st = (hipStream_t)stream;
    if ((rc = ensure_packed(c, st))) return rc;
    float* w = (float*)ws;
    const size_t so = c->slot * c->slot_stride;
    const size_t G = (size_t)c->G;
    float* r = tsave + TT_R * G; float* bip = tsave + TT_BIP * G; float* sa1 = tsave + TT_SA1 * G; float* sa2 = tsave + TT_SA2 * G;
    float* xs = tsave + TT_XS * G;
    // the same kernels, in the same order, as the inference tail (genie_bipartite_readout, genie_spatial_agg3_fwd, read-outs); the layer
    // inputs land in `tsave` instead of the workspace slot
    k_part_sum32<<<(c->G * 32 + 255) / 256, 256, 0, st>>>(w + c->o_part + so, c->G, c->T, r);
    k_bip_out_m<<<tl_blocks(c->G, c->tail_cu_sa), 256, 0, st>>>(w + c->o_part + so, c->G, c->T, c->packed[PL_BIP], bip, 0, 0);
    if ((rc = sa_launch_pre(c, 1, bip, w, 0, st))) return rc;
    if ((rc = sa_launch_layer(c, 1, bip, pos, sa1, w, 0, true, st))) return rc;
    if ((rc = sa_launch_layer(c, 2, sa1, pos, sa2, w, 1, true, st))) return rc;
    if ((rc = sa_launch_layer(c, 3, sa2, pos, xs, w, 0, false, st))) return rc;
    HIP_TRY(hipFuncSetAttribute((const void*)k_readout_m<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(float) * (ROM_LDS_FLOATS + GP_IMG_FLOATS))));
    HIP_TRY(hipFuncSetAttribute((const void*)k_readout_m<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(float) * ROM_LDS_FLOATS)));
    RoArgs a = make_ro_args(c);
    a.T = n_t; a.x_spatial = xs; a.t_query = t_query;
    {
        RoArgs g = a;
        g.N = g.Nw = c->G; g.out = y_out; g.lat_out = y_latent_out; g.img = c->packed[PL_RO0];
        k_readout_m<0><<<tl_blocks(g.N, c->tail_cu_ro), 256, sizeof(float) * ROM_LDS_FLOATS, st>>>(g);
    }
    float* cvbuf = w + c->o_cv + so;
    k_ro_pre_m<<<tl_blocks(c->G, c->tail_cu_ro), 256, 0, st>>>(xs, c->G, c->packed[PL_ROP], cvbuf, c->G, 0);
    {
        RoArgs q = a;
        q.N = q.Nw = n_query; q.x_grid = pos; q.x_query = x_query; q.knn = knn; q.out = x_out; q.img = c->packed[PL_RO1]; q.cv = cvbuf;
        k_readout_m<1><<<tl_blocks(q.N, c->tail_cu_ro), 256, sizeof(float) * ROM_LDS_FLOATS, st>>>(q);
    }
    HIP_TRY(hipGetLastError());
    return GENIE_OK;
}

int genie_tail_train_bwd(genie_ctx* c, const float* pos, const float* x_query, const int32_t* knn, const int32_t* rknn_rowptr,
                         const int32_t* rknn_edge, int n_query, int k, const float* t_query, int n_t, const float* tsave,
                         const float* d_y, const float* d_x, const float* d_xs_extra, const float* d_ylat_extra, const float* d_qlat_extra,
                         float* scratch, float* d_r_out, float* grad_blob, void* stream) {
    if (!c || !pos || !x_query || !knn || !rknn_rowptr || !rknn_edge || !t_query || !tsave || !d_y || !d_x || !scratch || !d_r_out || !grad_blob)
        return fail(GENIE_ERR_ARG, "genie_tail_train_bwd: null argument");
    int rc;
    if ((rc = train_check(c, "genie_tail_train_bwd"))) return rc;
    if (k != RO_K || n_query < 1 || n_t < 1 || n_t > RO_TMAX) return fail(GENIE_ERR_ARG, "genie_tail_train_bwd: k = 10, n_query >= 1, 1 <= n_t <= 10");
    hipStream_t st = (hipStream_t)stream;
    if ((rc = ensure_packed(c, st))) return rc;
    if ((rc = ensure_reversed(c))) return rc;
    const size_t G = (size_t)c->G;
    const float* r = tsave + TT_R * G; const float* bip = tsave + TT_BIP * G; const float* sa1 = tsave + TT_SA1 * G;
    const float* sa2 = tsave + TT_SA2 * G; const float* xs = tsave + TT_XS * G;
    const TtScratch L = tt_layout(c, n_query);
    float* S = scratch;
    HIP_TRY(hipMemsetAsync(grad_blob, 0, sizeof(float) * genie_train_grad_floats(), st));
    HIP_TRY(hipFuncSetAttribute((const void*)k_ro_bwd<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(float) * RB_LDS_FLOATS)));
    HIP_TRY(hipFuncSetAttribute((const void*)k_ro_bwd<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(float) * RB_LDS_FLOATS)));
    RoArgs ro = make_ro_args(c);
    ro.T = n_t; ro.x_spatial = xs; ro.t_query = t_query;
    {   // y branch: TemporalAttention + SpatialDirect
        RbArgs b;
        memset(&b, 0, sizeof(b));
        b.ro = ro; b.ro.N = b.ro.Nw = c->G; b.ro.img = c->packed[PL_RO0];
        b.timg = c->packed[PL_TRO0]; b.d_out = d_y; b.d_lat = d_ylat_extra; b.dxs = S + L.dxs_a;
        b.part = S + L.part_ro; b.n_acc = c->n_acc[TM_RO0]; b.n_vec = c->n_vec[TM_RO0];
        const int grid = tt_grid(c->G);
        k_ro_bwd<0><<<grid, 256, sizeof(float) * RB_LDS_FLOATS, st>>>(b);
        tt_reduce(c, TM_RO0, b.part, grid * 4, grad_blob, st);
    }
    {   // x branch: TemporalAttention + SpatialAttention (query side), then its grid-node side
        k_ro_pre_m<<<tl_blocks(c->G, c->tail_cu_ro), 256, 0, st>>>(xs, c->G, c->packed[PL_ROP], S + L.cv, c->G, 0);
        RbArgs b;
        memset(&b, 0, sizeof(b));
        b.ro = ro; b.ro.N = b.ro.Nw = n_query; b.ro.x_grid = pos; b.ro.x_query = x_query; b.ro.knn = knn; b.ro.cv = S + L.cv;
        b.ro.img = c->packed[PL_RO1];
        b.timg = c->packed[PL_TRO1]; b.d_out = d_x; b.d_lat = d_qlat_extra; b.eb = S + L.eb; b.dxm = S + L.dxm;
        b.part = S + L.part_ro; b.n_acc = c->n_acc[TM_RO1]; b.n_vec = c->n_vec[TM_RO1];
        const int grid = tt_grid(n_query);
        k_ro_bwd<1><<<grid, 256, sizeof(float) * RB_LDS_FLOATS, st>>>(b);
        tt_reduce(c, TM_RO1, b.part, grid * 4, grad_blob, st);
        k_tq_bwd<<<1, 256, 0, st>>>(c->raw, grad_blob + g_raw_total, t_query, n_t, c->scale_t, g_params[W_TA_Q1_W].off, g_params[W_TA_Q1_B].off,
                                    g_params[W_TA_Q2_W].off, g_params[W_TA_Q2_B].off, g_params[W_TA_ACT3].off, grad_blob);
        SnArgs n;
        memset(&n, 0, sizeof(n));
        n.G = c->G; n.nq = n_query; n.x_spatial = xs; n.x_grid = pos; n.x_query = x_query; n.r_rowptr = rknn_rowptr; n.r_edge = rknn_edge;
        n.eb = S + L.eb; n.dxm = S + L.dxm; n.raw = c->raw; n.o_sq_w = g_params[W_SAT_Q_W].off; n.o_sq_b = g_params[W_SAT_Q_B].off;
        n.scale_rel = c->scale_rel; n.timg = c->packed[PL_TSN]; n.dxs = S + L.dxs_b;
        n.part = S + L.part_a; n.n_acc = c->n_acc[TM_SN]; n.n_vec = c->n_vec[TM_SN];
        const int gn = tt_grid(c->G);
        k_sat_node_bwd<<<gn, 256, 0, st>>>(n);
        tt_reduce(c, TM_SN, n.part, gn * 4, grad_blob, st);
    }
    // SpatialAggregation 3, 2, 1
    const float* x_in[3] = {bip, sa1, sa2};
    float* dxl[2] = {S + L.dx0, S + L.dx1};
    const int gs = tt_grid(c->G);
    for (int layer = 3; layer >= 1; --layer) {
        SaArgs pre;
        memset(&pre, 0, sizeof(pre));
        sa_fill_layer(c, layer, pre);
        pre.x_in = x_in[layer - 1]; pre.pj_out = S + L.pj; pre.gpart_out = S + L.gpart; pre.img = c->packed[PL_SA1 + layer - 1];
        const int nbp = sa_blocks(c);
        if (layer == 1) k_sa_pre_m<15><<<nbp, 256, 0, st>>>(pre); else k_sa_pre_m<30><<<nbp, 256, 0, st>>>(pre);
        SbArgs a;
        memset(&a, 0, sizeof(a));
        a.G = c->G; a.C = layer == 1 ? 15 : 30; a.E = c->E_src; a.x_in = x_in[layer - 1]; a.pos = pos;
        a.rowptr = c->src_rowptr; a.col = c->src_col; a.outdeg = c->outdeg; a.r_rowptr = c->r_src_rowptr; a.r_col = c->r_src_col;
        a.raw = c->raw; a.fc1_w = g_params[(layer == 1 ? W_SA1_FC1_W : (layer == 2 ? W_SA2_FC1_W : W_SA3_FC1_W))].off;
        a.scale_rel = c->scale_rel; a.pj = S + L.pj; a.gpart = S + L.gpart; a.n_gpart = nbp;
        a.img = c->packed[PL_SA1 + layer - 1]; a.timg = c->packed[PL_TSA1 + layer - 1];
        if (layer == 3) { a.dout_a = S + L.dxs_a; a.dout_b = S + L.dxs_b; a.dout_x30 = d_xs_extra; }
        else a.dout_a = dxl[layer & 1];
        a.dxd = S + L.dxd; a.dan = S + L.dan; a.dx = dxl[(layer - 1) & 1];
        const int tma = TM_SAA1 + layer - 1, tmb = TM_SAB1 + layer - 1;
        a.part = S + L.part_b; a.n_acc = c->n_acc[tma]; a.n_vec = c->n_vec[tma];
        if (layer == 1) k_sa_bwd_a<15><<<gs, 256, 0, st>>>(a); else k_sa_bwd_a<30><<<gs, 256, 0, st>>>(a);
        tt_reduce(c, tma, a.part, gs * 4, grad_blob, st);
        a.part_a = S + L.part_b; a.n_waves_a = gs * 4; a.n_acc_a = c->n_acc[tma]; a.n_vec_a = c->n_vec[tma];
        a.part = S + L.part_a; a.n_acc = c->n_acc[tmb]; a.n_vec = c->n_vec[tmb]; a.blob = grad_blob;
        if (layer == 1) k_sa_bwd_b<15><<<gs, 256, 0, st>>>(a); else k_sa_bwd_b<30><<<gs, 256, 0, st>>>(a);
        tt_reduce(c, tmb, a.part, gs * 4, grad_blob, st);
    }
    {   // Bipartite_ReadIn.fc2: d bip (= the gradient of SpatialAggregation1's input) -> d r
        BbArgs b;
        memset(&b, 0, sizeof(b));
        b.G = c->G; b.r = r; b.dbip = dxl[0]; b.img = c->packed[PL_BIP]; b.timg = c->packed[PL_TBIP]; b.dr = d_r_out;
        b.part = S + L.part_a; b.n_acc = c->n_acc[TM_BIP]; b.n_vec = c->n_vec[TM_BIP];
        k_bip_bwd<<<gs, 256, 0, st>>>(b);
        tt_reduce(c, TM_BIP, b.part, gs * 4, grad_blob, st);
    }
    HIP_TRY(hipGetLastError());
    return GENIE_OK;
}

// Whole backward of a training step of `forward_fixed_source`: the tail (genie_tail_train_bwd) and then the P-sized front
// (genie_da_train_bwd) driven by the tail's d r, into one gradient blob.
int genie_train_bwd(genie_ctx* c, const float* slice, const float* mask, const float* edge_attr, const float* save, const float* pos,
                    const float* x_query, const int32_t* knn, const int32_t* rknn_rowptr, const int32_t* rknn_edge, int n_query, int k,
                    const float* t_query, int n_t, const float* tsave, const float* d_y, const float* d_x, const float* d_xs_extra,
                    const float* d_ylat_extra, const float* d_qlat_extra, float* tail_scratch, float* front_scratch, float* d_r_scratch,
                    float* grad_blob, void* stream) {
    int rc = genie_tail_train_bwd(c, pos, x_query, knn, rknn_rowptr, rknn_edge, n_query, k, t_query, n_t, tsave, d_y, d_x, d_xs_extra,
                                  d_ylat_extra, d_qlat_extra, tail_scratch, d_r_scratch, grad_blob, stream);
    if (rc) return rc;
    return da_train_bwd_impl(c, slice, mask, edge_attr, save, d_r_scratch, front_scratch, grad_blob, stream, false);
}




int genie_where_am_i(int32_t* out_dev, int n_blocks, void* stream) {
    if (!out_dev || n_blocks < 1) return fail(GENIE_ERR_ARG, "genie_where_am_i: bad argument");
    k_where_am_i<<<n_blocks, 64, 0, (hipStream_t)stream>>>(out_dev);
    HIP_TRY(hipGetLastError());
    return GENIE_OK;
}

int genie_set_tail_grid(genie_ctx* c, int readout_workgroups, int sa_workgroups) {
    if (!c || readout_workgroups < 0 || sa_workgroups < 0) return fail(GENIE_ERR_ARG, "genie_set_tail_grid: bad argument");
    c->tail_cu_ro = readout_workgroups > 0 ? readout_workgroups : c->num_cu * 2;
    c->tail_cu_sa = sa_workgroups > 0 ? sa_workgroups : c->num_cu * 2;
    return GENIE_OK;
}


size_t genie_assoc_workspace_bytes(const genie_ctx* c) { return c ? sizeof(float) * 3 * 32 * (size_t)c->P : 0; }

namespace {
int assoc_fwd_impl(genie_ctx* c, const float* y_latent, const float* mask_src, const float* x_latent, const float* mask,
                   const float* edge_attr, float* out, void* assoc_ws, void* ws, void* stream, float* save);
void assoc_pre_launch(genie_ctx* c, const float* y_latent, const float* mask_src, hipStream_t st) {
    AsPreOffs o;
    o.ro_fc1_w = g_params[W_RO_FC1_W].off; o.ro_fc1_b = g_params[W_RO_FC1_B].off; o.as_init_w = g_params[W_AS_INIT_W].off;
    o.as_l1t12_w = g_params[W_AS_L1T12_W].off; o.as_l1t22_w = g_params[W_AS_L1T22_W].off;
    o.as_l2t12_w = g_params[W_AS_L2T12_W].off; o.as_l2t22_w = g_params[W_AS_L2T22_W].off;
    o.as_init_abs = g_params[W_AS_INIT_ABS].off; o.as_l1t12_p = g_params[W_AS_L1T12_P].off; o.as_l1t22_p = g_params[W_AS_L1T22_P].off;
    o.as_l2t12_p = g_params[W_AS_L2T12_P].off; o.as_l2t22_p = g_params[W_AS_L2T22_P].off;
    const float* mpos_src = c->has_edges ? c->mpos_src : nullptr;
    const float* mpos_sta = c->has_edges ? c->mpos_sta : nullptr;
    k_assoc_pre<<<(c->G * AS_PG + 255) / 256, 256, 0, st>>>(c->raw, o, y_latent, mask_src, c->G, mpos_src, c->abs_src, c->as_pg);
    if (c->as_ps) k_assoc_ps<<<(c->S * AS_PS + 255) / 256, 256, 0, st>>>(c->raw, o, c->S, mpos_sta, c->abs_sta, c->as_ps);
}
}  // namespace

int genie_assoc_fwd(genie_ctx* c, const float* y_latent, const float* mask_src, const float* x_latent, const float* mask,
                    const float* edge_attr, float* out, void* assoc_ws, void* ws, void* stream) {
    return assoc_fwd_impl(c, y_latent, mask_src, x_latent, mask, edge_attr, out, assoc_ws, ws, stream, nullptr);
}

namespace {
int assoc_fwd_impl(genie_ctx* c, const float* y_latent, const float* mask_src, const float* x_latent, const float* mask,
                   const float* edge_attr, float* out, void* assoc_ws, void* ws, void* stream, float* save) {
    int rc = check_ws(c, ws);
    if (rc) return rc;
    if (!y_latent || !mask_src || !x_latent || !mask || !edge_attr || !out || !assoc_ws)
        return fail(GENIE_ERR_ARG, "genie_assoc_fwd: null argument");
    if (c->pcsr || c->G_ext != c->G) return fail(GENIE_ERR_STATE, "genie_assoc_fwd: needs an unsharded Cartesian product graph");
    if (((uintptr_t)assoc_ws & 15) != 0) return fail(GENIE_ERR_ARG, "genie_assoc_fwd: assoc_ws must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if ((rc = ensure_packed(c, st))) return rc;
    if (!c->as_pg) HIP_TRY(hipMalloc((void**)&c->as_pg, sizeof(float) * AS_PG * (size_t)c->G));
    const bool variant = c->has_edges || c->abs_sta != nullptr;
    if (variant && save) return fail(GENIE_ERR_STATE, "genie_assoc_train_fwd: default model definition only");
    if (variant && !c->as_ps) HIP_TRY(hipMalloc((void**)&c->as_ps, sizeof(float) * AS_PS * (size_t)c->S));
    assoc_pre_launch(c, y_latent, mask_src, st);
    if (save) { c->force_generic = 1; c->train_save = save; }      // training forward: caller's station order, pre-activations kept
    DaArgs d = make_da_args(c, (float*)ws);
    AsArgs a;
    memset(&a, 0, sizeof(a));
    a.S = c->S; a.G = c->G; a.T = c->T; a.seg = d.seg; a.nxcd = d.nxcd;
    a.order = c->order;
    a.sta_rowptr = d.sta_rowptr; a.sta_col = d.sta_col; a.src_rowptr = c->src_rowptr; a.src_col = c->src_col;
    a.sta_user = d.sta_user;
    a.pg = c->as_pg; a.ps = variant ? c->as_ps : nullptr; a.x_latent = x_latent; a.mask = mask; a.edge_attr = edge_attr;
    a.tr = (float*)assoc_ws; a.q1 = a.tr + 32 * (size_t)c->P; a.q2 = a.q1 + 32 * (size_t)c->P;
    a.c = d.c; a.wu = d.wu; a.wv = d.wv;
    a.save = save; a.Pn = c->P;
    const int grid = da_grid(c, (long long)c->G * c->T, std::max(1, c->bpc1));
    a.packed = c->packed[2];
    k_assoc_a<<<grid, 256, 0, st>>>(a);
    a.packed = c->packed[3];
    k_assoc_b<<<grid, 256, 0, st>>>(a);
    // second pair of neighbour means + PReLU2 = the stage-2 kernel of this context without its Bipartite half
    rc = run_stage2(c, mask, edge_attr, out, ws, stream, 0, c->G, c->raw + g_params[W_AS_ACT2].off, 1);
    if (save) { c->force_generic = 0; c->train_save = nullptr; }
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return GENIE_OK;
}
}  // namespace

// Training step of the P-sized association heads (train_assoc_kernels.hpp).
size_t genie_assoc_train_save_floats(const genie_ctx* c) { return c ? (size_t)AV_BLOCKS * 16 * (size_t)c->P : 0; }
size_t genie_assoc_train_scratch_floats(const genie_ctx* c) {
    return c ? (size_t)GR_BLOCKS * 16 * (size_t)c->P + train_part_floats(c) + (size_t)c->G * c->T * 32 + 64 : 0;
}

int genie_assoc_train_fwd(genie_ctx* c, const float* y_latent, const float* mask_src, const float* x_latent, const float* mask,
                          const float* edge_attr, float* out, float* asave, void* assoc_ws, void* ws, void* stream) {
    if (!c || !asave) return fail(GENIE_ERR_ARG, "genie_assoc_train_fwd: null argument");
    int rc = train_check(c, "genie_assoc_train_fwd");
    if (rc) return rc;
    return assoc_fwd_impl(c, y_latent, mask_src, x_latent, mask, edge_attr, out, assoc_ws, ws, stream, asave);
}

int genie_assoc_train_bwd(genie_ctx* c, const float* y_latent, const float* mask_src, const float* x_latent, const float* mask,
                          const float* edge_attr, const float* asave, const float* d_s, float* scratch, float* d_ylat_out,
                          float* grad_blob, void* stream) {
    if (!c || !y_latent || !mask_src || !x_latent || !mask || !edge_attr || !asave || !d_s || !scratch || !d_ylat_out || !grad_blob)
        return fail(GENIE_ERR_ARG, "genie_assoc_train_bwd: null argument");
    int rc;
    if ((rc = train_check(c, "genie_assoc_train_bwd"))) return rc;
    hipStream_t st = (hipStream_t)stream;
    if ((rc = ensure_packed(c, st))) return rc;
    if ((rc = ensure_reversed(c))) return rc;
    if (!c->as_pg) HIP_TRY(hipMalloc((void**)&c->as_pg, sizeof(float) * AS_PG * (size_t)c->G));
    assoc_pre_launch(c, y_latent, mask_src, st);          // pg[31] = mask1[g] (another forward may have overwritten the table)
    HIP_TRY(hipMemsetAsync(grad_blob, 0, sizeof(float) * g_raw_total, st));
    TrArgs a;
    memset(&a, 0, sizeof(a));
    a.S = c->S; a.G = c->G; a.T = c->T; a.seg = std::max(1, c->seg);
    a.nxcd = 8;
    a.P = c->P; a.order = c->order;
    a.r_sta_rowptr = c->r_sta_rowptr; a.r_sta_col = c->r_sta_col; a.r_sta_w = c->r_sta_w;
    a.r_src_rowptr = c->r_src_rowptr; a.r_src_col = c->r_src_col; a.r_src_w = c->r_src_w;
    a.mask = mask; a.edge_attr = edge_attr; a.save = asave; a.x_latent = x_latent; a.pg = c->as_pg;
    a.gr = scratch; a.part = scratch + (size_t)GR_BLOCKS * 16 * (size_t)c->P;
    a.zsum = a.part + train_part_floats(c);
    a.sv_t = AV_T; a.sv_up = AV_UV; a.sv_vp = AV_UV + 2;
    const int grid = train_grid(c), n_waves = grid * 4;
    const int tms[4] = {TM_AB3, TM_AB2, TM_AB1, TM_AB0};
    const int pls[4] = {-1, PL_TAB2, PL_TAB1, PL_TAB0};
    for (int s = 0; s < 4; ++s) {
        const int tm = tms[s];
        a.packed = pls[s] >= 0 ? c->packed[pls[s]] : nullptr; a.n_acc = c->n_acc[tm]; a.n_vec = c->n_vec[tm];
        if (s == 0) k_as_b3<<<grid, 256, 0, st>>>(a, d_s, c->raw + g_params[W_AS_ACT2].off);
        else if (s == 1) k_train_b1<true><<<grid, 256, 0, st>>>(a);
        else if (s == 2) k_as_b1<<<grid, 256, 0, st>>>(a);
        else k_as_b0<<<grid, 256, 0, st>>>(a);
        const int stride = a.n_acc * 256 + a.n_vec * 16 + 16;
        k_train_reduce<<<(stride + 31) / 32, 256, 0, st>>>(a.part, n_waves, a.n_acc, a.n_vec, c->n_sc[tm], c->d_acc[tm], c->d_vec[tm],
                                                            c->d_sc[tm], grad_blob, 0);
    }
    {
        AgArgs g;
        memset(&g, 0, sizeof(g));
        g.G = c->G; g.T = c->T; g.zsum = a.zsum; g.y_latent = y_latent; g.timg = c->packed[PL_TAG]; g.d_ylat = d_ylat_out;
        g.part = a.part; g.n_acc = c->n_acc[TM_AG]; g.n_vec = c->n_vec[TM_AG];
        const int gg = tt_grid(c->G);
        k_as_g<<<gg, 256, 0, st>>>(g);
        tt_reduce(c, TM_AG, g.part, gg * 4, grad_blob, st);
    }
    HIP_TRY(hipGetLastError());
    return GENIE_OK;
}

int genie_knn(const float* x_context, int n_context, const float* x_query, int n_query, int k, int exclude_self,
              int32_t* out_idx, void* stream) {
    if (!x_context || !x_query || !out_idx || n_context < 1 || n_query < 0 || k < 1 || k > 16)
        return fail(GENIE_ERR_ARG, "genie_knn: bad argument (1 <= k <= 16)");
    if (n_query == 0) return GENIE_OK;
    hipStream_t st = (hipStream_t)stream;
    const int nb = (n_query + 3) / 4;
    if (k <= 8) k_knn<8><<<nb, 256, 0, st>>>(x_context, n_context, x_query, n_query, k, exclude_self, out_idx);
    else if (k <= 10) k_knn<10><<<nb, 256, 0, st>>>(x_context, n_context, x_query, n_query, k, exclude_self, out_idx);
    else k_knn<16><<<nb, 256, 0, st>>>(x_context, n_context, x_query, n_query, k, exclude_self, out_idx);
    HIP_TRY(hipGetLastError());
    return GENIE_OK;
}

int genie_row_select_count(const float* x, int rows, int64_t cols, float threshold, int mode, int32_t* counts, void* stream) {
    if (!x || !counts || rows < 0 || cols < 0 || cols >= (1ll << 31) || (mode != 0 && mode != 1))
        return fail(GENIE_ERR_ARG, "genie_row_select_count: bad argument");
    if (rows == 0) return GENIE_OK;
    hipStream_t st = (hipStream_t)stream;
    if (mode == 0) k_row_select<0, true><<<rows, 256, 0, st>>>(x, cols, threshold, counts, nullptr, nullptr, nullptr, nullptr);
    else k_row_select<1, true><<<rows, 256, 0, st>>>(x, cols, threshold, counts, nullptr, nullptr, nullptr, nullptr);
    HIP_TRY(hipGetLastError());
    return GENIE_OK;
}

int genie_row_select_fill(const float* x, int rows, int64_t cols, float threshold, int mode, const int64_t* offsets,
                          int32_t* out_row, int32_t* out_col, float* out_val, void* stream) {
    if (!x || !offsets || !out_row || !out_col || !out_val || rows < 0 || cols < 0 || cols >= (1ll << 31) || (mode != 0 && mode != 1))
        return fail(GENIE_ERR_ARG, "genie_row_select_fill: bad argument");
    if (rows == 0) return GENIE_OK;
    hipStream_t st = (hipStream_t)stream;
    const long long* off = (const long long*)offsets;
    if (mode == 0) k_row_select<0, false><<<rows, 256, 0, st>>>(x, cols, threshold, nullptr, off, out_row, out_col, out_val);
    else k_row_select<1, false><<<rows, 256, 0, st>>>(x, cols, threshold, nullptr, off, out_row, out_col, out_val);
    HIP_TRY(hipGetLastError());
    return GENIE_OK;
}

int genie_lslc_fwd(genie_ctx* c, int phase_head, const float* s_rows, const int32_t* a_edges, int64_t n_edges, int l_dt, float t0,
                   float dt, float eps, const float* tlatent, int tl_stride, int tl_col, const float* tpick, const int32_t* ipick,
                   const float* phase_label, int n_picks, float* out, void* stream) {
    if (!c || !s_rows || !a_edges || !tlatent || !tpick || !ipick || !phase_label || !out) return fail(GENIE_ERR_ARG, "genie_lslc_fwd: null argument");
    if (phase_head < 0 || phase_head > 1 || l_dt < 1 || n_edges < LS_K || !(dt > 0.f) || !(eps > 0.f) || tl_stride < 1 || tl_col < 0 || tl_col >= tl_stride)
        return fail(GENIE_ERR_ARG, "genie_lslc_fwd: bad argument");
    if (n_picks < 1) return GENIE_OK;
    hipStream_t st = (hipStream_t)stream;
    { int rcp = ensure_packed(c, st); if (rcp) return rcp; }
    LsArgs a;
    memset(&a, 0, sizeof(a));
    a.n_picks = n_picks; a.l_dt = l_dt; a.n_edges = n_edges; a.t0 = t0; a.dt = dt; a.eps = eps;
    a.s = s_rows; a.A_edges = a_edges; a.tlatent = tlatent; a.tl_stride = tl_stride; a.tl_col = tl_col;
    a.tpick = tpick; a.ipick = ipick; a.phase = phase_label; a.img = c->packed[phase_head == 0 ? PL_LSP : PL_LSS]; a.out = out;
    k_lslc<<<tl_blocks(n_picks, c->num_cu * 4), 256, 0, st>>>(a);
    HIP_TRY(hipGetLastError());
    return GENIE_OK;
}

// Backward of genie_lslc_fwd (training; k_lslc_bwd): d_out [n_picks, 15] -> gradients of the head's fc1 / fc2 / PReLU slopes ADDED into
// grad_blob (weight-mirror layout; the caller zeroes it once for both heads), the gradient of every gathered s row in erow
// [n_picks * 10][32] and its product node in etgt [n_picks * 10] (-1: edge dropped by the 2-eps filter). genie_seg_rows then adds the
// rows to d s [P, 30] per product node in the order of `order` (edges sorted by etgt, stable): deterministic.
size_t genie_lslc_bwd_part_floats(int n_picks) { return tt_part_floats(8, 3, tt_grid(std::max(1, n_picks))); }

int genie_lslc_bwd(genie_ctx* c, int phase_head, const float* s_rows, const int32_t* a_edges, int64_t n_edges, int l_dt, float t0,
                   float dt, float eps, const float* tlatent, int tl_stride, int tl_col, const float* tpick, const int32_t* ipick,
                   const float* phase_label, int n_picks, const float* d_out, float* erow, int32_t* etgt, float* part_scratch,
                   float* grad_blob, void* stream) {
    if (!c || !s_rows || !a_edges || !tlatent || !tpick || !ipick || !phase_label || !d_out || !erow || !etgt || !part_scratch || !grad_blob)
        return fail(GENIE_ERR_ARG, "genie_lslc_bwd: null argument");
    if (phase_head < 0 || phase_head > 1 || l_dt < 1 || n_edges < LS_K || !(dt > 0.f) || !(eps > 0.f) || tl_stride < 1 || tl_col < 0 || tl_col >= tl_stride ||
        n_picks < 1)
        return fail(GENIE_ERR_ARG, "genie_lslc_bwd: bad argument");
    hipStream_t st = (hipStream_t)stream;
    { int rcp = ensure_packed(c, st); if (rcp) return rcp; }
    LbArgs b;
    memset(&b, 0, sizeof(b));
    b.f.n_picks = n_picks; b.f.l_dt = l_dt; b.f.n_edges = n_edges; b.f.t0 = t0; b.f.dt = dt; b.f.eps = eps;
    b.f.s = s_rows; b.f.A_edges = a_edges; b.f.tlatent = tlatent; b.f.tl_stride = tl_stride; b.f.tl_col = tl_col;
    b.f.tpick = tpick; b.f.ipick = ipick; b.f.phase = phase_label; b.f.img = c->packed[phase_head == 0 ? PL_LSP : PL_LSS];
    b.timg = c->packed[phase_head == 0 ? PL_TLSP : PL_TLSS];
    b.d_out = d_out; b.erow = erow; b.etgt = etgt;
    const int tm = phase_head == 0 ? TM_LSP : TM_LSS;
    b.part = part_scratch; b.n_acc = c->n_acc[tm]; b.n_vec = c->n_vec[tm];
    const int grid = tt_grid(n_picks);
    k_lslc_bwd<<<grid, 256, 0, st>>>(b);
    tt_reduce(c, tm, b.part, grid * 4, grad_blob, st);
    HIP_TRY(hipGetLastError());
    return GENIE_OK;
}

int genie_seg_rows(const float* erow, const int32_t* etgt, const int32_t* order, int64_t n_edges, float* ds, void* stream) {
    if (!erow || !etgt || !order || !ds || n_edges < 0) return fail(GENIE_ERR_ARG, "genie_seg_rows: bad argument");
    if (n_edges == 0) return GENIE_OK;
    k_seg_rows<<<(unsigned)((n_edges * 8 + 255) / 256), 256, 0, (hipStream_t)stream>>>(erow, etgt, order, n_edges, ds);
    HIP_TRY(hipGetLastError());
    return GENIE_OK;
}

namespace {
static int arrivals_fwd_impl(genie_ctx* c, int n_src, const float* stime, const float* src_embed, const float* trv_src, int n_sta,
                      const float* arrival_p, const float* arrival_s, const float* tpick, const float* phase_label, int n_arv,
                      const int32_t* order, const int32_t* seg_sta, const int32_t* seg_start, const int32_t* seg_len, int n_useg,
                      float eps, float* ctx_scratch, int32_t* e0max_scratch, float* out, float* save, hipStream_t st, ArArgs* a_out) {
    if (!c || !stime || !src_embed || !trv_src || !arrival_p || !arrival_s || !tpick || !phase_label || !order || !seg_sta || !seg_start ||
        !seg_len || !ctx_scratch || !e0max_scratch)
        return fail(GENIE_ERR_ARG, "genie_arrivals: null argument");
    if (n_src < 1 || n_sta < 1 || n_arv < 1 || n_useg < 1 || !(eps > 0.f)) return fail(GENIE_ERR_ARG, "genie_arrivals: bad argument");
    if ((long long)n_src * n_useg > 0x7fffffffLL || (long long)n_src * n_arv > 0x7fffffffLL)
        return fail(GENIE_ERR_ARG, "genie_arrivals: too many (source, station) or (source, pick) pairs");
    { int rcp = ensure_packed(c, st); if (rcp) return rcp; }
    ArArgs a;
    memset(&a, 0, sizeof(a));
    a.n_src = n_src; a.n_sta = n_sta; a.n_arv = n_arv; a.n_useg = n_useg; a.eps = eps;
    a.stime = stime; a.trv_src = trv_src; a.ctx = ctx_scratch; a.arv_p = arrival_p; a.arv_s = arrival_s; a.tpick = tpick; a.phase = phase_label;
    a.order = order; a.seg_sta = seg_sta; a.seg_start = seg_start; a.seg_len = seg_len; a.img = c->packed[PL_ARR]; a.out = out; a.e0max = e0max_scratch;
    a.save = save;
    if (a_out) { *a_out = a; return GENIE_OK; }        // the backward: ctx / e0max of the forward are still in the caller's scratch
    k_arr_ctx<<<n_src, 128, 0, st>>>(c->raw, g_params[W_AR_C1_W].off, g_params[W_AR_C1_B].off, g_params[W_AR_C2_W].off, g_params[W_AR_C2_B].off,
                                     g_params[W_AR_ACT1].off, src_embed, stime, n_src, ctx_scratch);
    HIP_TRY(hipMemsetAsync(e0max_scratch, 0xff, sizeof(int32_t), st));       // -1
    k_arr_e0max<<<n_src * n_useg, 256, 0, st>>>(a);
    const size_t lds = sizeof(float) * (GA2_IMG_FLOATS + AR_CAP * AR_ENT + 2 * AR_CAP + 192 + 8);
    HIP_TRY(hipFuncSetAttribute((const void*)k_arrivals, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    k_arrivals<<<n_src * n_useg, 256, lds, st>>>(a);
    HIP_TRY(hipGetLastError());
    return GENIE_OK;
}
constexpr int ARRT_GRID = 512;        // workgroups of the two backward passes (each wave owns one partial slot)
inline size_t arrt_stride(int n_acc, int n_vec) { return (size_t)n_acc * 256 + (size_t)n_vec * 16 + 16; }
}  // namespace

int genie_arrivals_fwd(genie_ctx* c, int n_src, const float* stime, const float* src_embed, const float* trv_src, int n_sta,
                       const float* arrival_p, const float* arrival_s, const float* tpick, const float* phase_label, int n_arv,
                       const int32_t* order, const int32_t* seg_sta, const int32_t* seg_start, const int32_t* seg_len, int n_useg,
                       float eps, float* ctx_scratch, int32_t* e0max_scratch, float* out, void* stream) {
    if (!out) return fail(GENIE_ERR_ARG, "genie_arrivals_fwd: null argument");
    return arrivals_fwd_impl(c, n_src, stime, src_embed, trv_src, n_sta, arrival_p, arrival_s, tpick, phase_label, n_arv, order, seg_sta,
                             seg_start, seg_len, n_useg, eps, ctx_scratch, e0max_scratch, out, nullptr, (hipStream_t)stream, nullptr);
}

int64_t genie_arrivals_train_save_floats(int n_src, int n_arv) { return (int64_t)n_src * n_arv * AT_STAT; }

int genie_arrivals_train_fwd(genie_ctx* c, int n_src, const float* stime, const float* src_embed, const float* trv_src, int n_sta,
                             const float* arrival_p, const float* arrival_s, const float* tpick, const float* phase_label, int n_arv,
                             const int32_t* order, const int32_t* seg_sta, const int32_t* seg_start, const int32_t* seg_len, int n_useg,
                             float eps, float* ctx_scratch, int32_t* e0max_scratch, float* out, float* save, void* stream) {
    if (!out || !save) return fail(GENIE_ERR_ARG, "genie_arrivals_train_fwd: null argument");
    return arrivals_fwd_impl(c, n_src, stime, src_embed, trv_src, n_sta, arrival_p, arrival_s, tpick, phase_label, n_arv, order, seg_sta,
                             seg_start, seg_len, n_useg, eps, ctx_scratch, e0max_scratch, out, save, (hipStream_t)stream, nullptr);
}

int64_t genie_arrivals_bwd_scratch_floats(int n_src, int n_arv, int n_useg) {
    if (n_src < 1 || n_arv < 1 || n_useg < 1) return 0;
    const int64_t tgt = (int64_t)n_src * n_arv;
    return tgt * AT_TG + tgt * 32 + (int64_t)n_src * n_useg * 192 + (int64_t)n_src * AC_STRIDE +
           (int64_t)ARRT_GRID * 4 * (int64_t)std::max(arrt_stride(AT_NACC, AT_NVEC), arrt_stride(AE_NACC, AE_NVEC));
}

int genie_arrivals_bwd(genie_ctx* c, int n_src, const float* stime, const float* src_embed, const float* trv_src, int n_sta,
                       const float* arrival_p, const float* arrival_s, const float* tpick, const float* phase_label, int n_arv,
                       const int32_t* order, const int32_t* seg_sta, const int32_t* seg_start, const int32_t* seg_len, int n_useg,
                       float eps, const float* ctx_scratch, const int32_t* e0max_scratch, const float* save, const float* d_out,
                       float* scratch, float* d_src_embed, float* d_arrival_p, float* d_arrival_s, float* grad_blob, void* stream) {
    if (!save || !d_out || !scratch || !d_src_embed || !d_arrival_p || !d_arrival_s || !grad_blob)
        return fail(GENIE_ERR_ARG, "genie_arrivals_bwd: null argument");
    hipStream_t st = (hipStream_t)stream;
    ArArgs a;
    { int rc = arrivals_fwd_impl(c, n_src, stime, src_embed, trv_src, n_sta, arrival_p, arrival_s, tpick, phase_label, n_arv, order, seg_sta,
                                 seg_start, seg_len, n_useg, eps, (float*)ctx_scratch, (int32_t*)e0max_scratch, nullptr, nullptr, st, &a);
      if (rc) return rc; }
    const long long n_tgt = (long long)n_src * n_arv;
    float* tg = scratch;
    float* darv = tg + n_tgt * AT_TG;
    float* cpair = darv + n_tgt * 32;
    float* cpart = cpair + (long long)n_src * n_useg * 192;
    float* part = cpart + (long long)n_src * AC_STRIDE;
    {
        AtArgs b;
        memset(&b, 0, sizeof(b));
        b.n_tgt = (int)n_tgt; b.img = c->packed[PL_ARR]; b.timg = c->packed[PL_TARR]; b.tstat = save; b.d_out = d_out; b.tg = tg;
        b.part = part; b.n_acc = c->n_acc[TM_ART]; b.n_vec = c->n_vec[TM_ART];
        const int grid = (int)std::min<long long>(ARRT_GRID, (n_tgt + 63) / 64);
        const size_t lds = sizeof(float) * (GA2_IMG_FLOATS + GTA_IMG_FLOATS + 4 * 16 * 17);
        HIP_TRY(hipFuncSetAttribute((const void*)k_arrt_tgt_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        k_arrt_tgt_bwd<<<grid, 256, lds, st>>>(b);
        int rc = tt_reduce(c, TM_ART, part, grid * 4, grad_blob, st);
        if (rc) return rc;
    }
    {
        AeArgs b;
        memset(&b, 0, sizeof(b));
        b.f = a; b.timg = c->packed[PL_TARR]; b.tg = tg; b.darv = darv; b.cpair = cpair;
        b.part = part; b.n_acc = c->n_acc[TM_ARE]; b.n_vec = c->n_vec[TM_ARE];
        const int grid = (int)std::min<long long>(ARRT_GRID, (long long)n_src * n_useg);
        const size_t lds = sizeof(float) * (GA2_IMG_FLOATS + GTA_IMG_FLOATS + 4 * 16 * 17 + 192 + AE_TCH * AT_TG + AE_TCH + 4 * 3 * 256);
        HIP_TRY(hipFuncSetAttribute((const void*)k_arrt_ent_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        k_arrt_ent_bwd<<<grid, 256, lds, st>>>(b);
        int rc = tt_reduce(c, TM_ARE, part, grid * 4, grad_blob, st);
        if (rc) return rc;
    }
    const int o1w = g_params[W_AR_C1_W].off, o1b = g_params[W_AR_C1_B].off, o2w = g_params[W_AR_C2_W].off, o2b = g_params[W_AR_C2_B].off,
              oa1 = g_params[W_AR_ACT1].off;
    k_arrt_ctx_bwd<<<n_src, 128, 0, st>>>(c->raw, o1w, o1b, o2w, o2b, oa1, src_embed, stime, n_src, n_useg, cpair, cpart, d_src_embed);
    k_arrt_ctx_red<<<(AC_PARAMS + 255) / 256, 256, 0, st>>>(cpart, n_src, o1w, o1b, o2w, o2b, oa1, grad_blob);
    k_arrt_pick_sum<<<(n_arv * 30 + 255) / 256, 256, 0, st>>>(darv, n_src, n_arv, d_arrival_p, d_arrival_s);
    HIP_TRY(hipGetLastError());
    return GENIE_OK;
}

int genie_subgraph_csr_count(const int32_t* pair_sta, const int32_t* pair_src, int64_t n_prod, const int32_t* seg_rowptr,
                             const int32_t* sta_rowptr, const int32_t* sta_col, const int32_t* src_rowptr, const int32_t* src_col,
                             int32_t* count_sta, int32_t* count_src, void* stream) {
    if (!pair_sta || !pair_src || !seg_rowptr || !sta_rowptr || !src_rowptr || !count_sta || !count_src || n_prod < 0 || n_prod >= (1ll << 31))
        return fail(GENIE_ERR_ARG, "genie_subgraph_csr_count: bad argument");
    if (n_prod == 0) return GENIE_OK;
    k_subgraph_csr<false><<<(unsigned)((n_prod + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        pair_sta, pair_src, n_prod, seg_rowptr, sta_rowptr, sta_col, src_rowptr, src_col, count_sta, count_src, nullptr, nullptr, nullptr, nullptr);
    HIP_TRY(hipGetLastError());
    return GENIE_OK;
}

int genie_subgraph_csr_fill(const int32_t* pair_sta, const int32_t* pair_src, int64_t n_prod, const int32_t* seg_rowptr,
                            const int32_t* sta_rowptr, const int32_t* sta_col, const int32_t* src_rowptr, const int32_t* src_col,
                            const int32_t* p_sta_rowptr, const int32_t* p_src_rowptr, int32_t* p_sta_col, int32_t* p_src_col,
                            void* stream) {
    if (!pair_sta || !pair_src || !seg_rowptr || !sta_rowptr || !src_rowptr || !p_sta_rowptr || !p_src_rowptr || !p_sta_col || !p_src_col ||
        n_prod < 0 || n_prod >= (1ll << 31))
        return fail(GENIE_ERR_ARG, "genie_subgraph_csr_fill: bad argument");
    if (n_prod == 0) return GENIE_OK;
    k_subgraph_csr<true><<<(unsigned)((n_prod + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        pair_sta, pair_src, n_prod, seg_rowptr, sta_rowptr, sta_col, src_rowptr, src_col, nullptr, nullptr, p_sta_rowptr, p_src_rowptr,
        p_sta_col, p_src_col);
    HIP_TRY(hipGetLastError());
    return GENIE_OK;
}

int genie_ws_export(genie_ctx* c, int which, void* ws, float* out, void* stream) {
    int rc = check_ws(c, ws);
    if (rc) return rc;
    if (!out) return fail(GENIE_ERR_ARG, "genie_ws_export: null out");
    float* w = (float*)ws;
    const float* src; long long rows; int pitch, ncol;
    switch (which) {
        case 0: src = w + c->o_c + (c->slot % GENIE_NBIG) * c->big_stride; rows = c->P; pitch = ROWC; ncol = 30; break;
        case 1: src = w + c->o_wu + (c->slot % GENIE_NBIG) * c->big_stride; rows = c->P; pitch = ROWW; ncol = 15; break;
        case 2: src = w + c->o_wv + (c->slot % GENIE_NBIG) * c->big_stride; rows = c->P; pitch = ROWW; ncol = 15; break;
        default: return fail(GENIE_ERR_ARG, "genie_ws_export: which must be 0..2");
    }
    const long long n = rows * ncol;
    k_export<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(src, rows, pitch, ncol, out, sta_order_on(c) ? c->sta_perm : nullptr, c->S);
    HIP_TRY(hipGetLastError());
    return GENIE_OK;
}

}  // extern "C"
