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
#include <array>
#include <map>
#include <mutex>
#include <utility>
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

// read-once rows (c, Mask, edge_attr of stage 2): plain loads (non-temporal ones measured slower, 0.354 vs 0.326 ms)
#define GENIE_LD_STREAM(ptr) (*(ptr))
// Register budgets and depths settled by measurement (profiles/EXPERIMENTS.md): k_stage2_ord / k_stage2_h2 are held to two waves per
// SIMD (three: spills, +50 % fabric reads, 0.313 vs 0.302 ms); k_stage1_h2 keeps 6 row loads in flight ahead of their use (3..8 equal)
#define GENIE_S2_WAVES 2
#define GENIE_S2H_WAVES 2
#define GENIE_H2_DEPTH 6

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
// Device-memory pool of the library's own allocations (graph tables, weight images, plan tables).
// The reference's training loop hands `forward` a new graph per sample (train_GENIE_model.py:1722-1786), so a context is destroyed
// and created per sample: ~100 hipMalloc / hipFree pairs through the driver (each hipFree also drains the device) were most of
// that cost. Freed blocks are kept per device and size class (sizes rounded up to 1/8-octave steps, so a context of 199 stations
// reuses the blocks of one of 200) and handed out again; above POOL_CAP cached bytes a freed block goes back to the driver.
// Blocks are reused without an implicit device synchronisation: gfree_sync() (one hipDeviceSynchronize, then gfree) where work
// that may still read the block can be in flight; genie_ctx_destroy synchronises once for all of its blocks.
// ------------------------------------------------------------------------------------------------
struct DevPool {
    std::multimap<size_t, void*> free_;
    size_t cached = 0;
};
std::map<int, DevPool> g_pools;
std::map<void*, std::pair<int, size_t>> g_pool_blocks;      // every block the pool handed out: (device, size class)
std::mutex g_pool_mutex;
constexpr size_t POOL_CAP = (size_t)3 << 30;

size_t pool_class(size_t n) {
    if (n < 256) return 256;
    int sh = 0;
    while ((n >> sh) >= 16) ++sh;                 // n = m * 2^sh with 8 <= m < 16
    const size_t step = (size_t)1 << sh;
    return (n + step - 1) / step * step;
}

hipError_t gmalloc(void** p, size_t bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const size_t cls = pool_class(bytes);
    {
        std::lock_guard<std::mutex> lk(g_pool_mutex);
        DevPool& P = g_pools[dev];
        auto it = P.free_.find(cls);
        if (it != P.free_.end()) {
            *p = it->second;
            P.cached -= cls;
            P.free_.erase(it);
            return hipSuccess;
        }
    }
    e = hipMalloc(p, cls);
    if (e != hipSuccess) {          // give the cache back to the driver and try once more
        std::lock_guard<std::mutex> lk(g_pool_mutex);
        DevPool& P = g_pools[dev];
        (void)hipDeviceSynchronize();
        for (auto& kv : P.free_) { g_pool_blocks.erase(kv.second); (void)hipFree(kv.second); }
        P.free_.clear(); P.cached = 0;
        e = hipMalloc(p, cls);
        if (e != hipSuccess) return e;
    }
    std::lock_guard<std::mutex> lk(g_pool_mutex);
    g_pool_blocks[*p] = std::make_pair(dev, cls);
    return hipSuccess;
}
template <typename T> hipError_t gmalloc(T** p, size_t bytes) { return gmalloc((void**)p, bytes); }

hipError_t gfree(void* p) {         // the caller guarantees that no launched work still uses the block
    if (!p) return hipSuccess;
    std::lock_guard<std::mutex> lk(g_pool_mutex);
    auto it = g_pool_blocks.find(p);
    if (it == g_pool_blocks.end()) return hipFree(p);
    DevPool& P = g_pools[it->second.first];          // the pool of the device the block lives on, whatever the current device is
    const size_t cls = it->second.second;
    if (P.cached + cls > POOL_CAP) { g_pool_blocks.erase(it); return hipFree(p); }
    P.free_.emplace(cls, p);
    P.cached += cls;
    return hipSuccess;
}

hipError_t gfree_sync(void* p) {
    if (!p) return hipSuccess;
    (void)hipDeviceSynchronize();
    return gfree(p);
}

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

// STAGE 2 on the 16-bit matrix pipe (k_stage2_h2): 1-KB A fragments of v_mfma_f32_16x16x32_f16 for Bipartite_ReadIn.fc1, lane
// (i = lane & 15, kg = lane >> 4) holds the 8 K slots (kg, e) of output row 16 t + i. K slots of the x_latent step: e < 4 = channel
// 4 kg + e of the first half (o1, 15 channels + 1 zero), e >= 4 = channel 4 kg + e - 4 of the second half; of the edge_attr step:
// group 0 = {e0, e1, e2, - | e0, e1, e2, -} (first | second pieces), group 1 = {e0, e1, e2 (/ 16), - | -}; see k_ea_frag.
constexpr int S2H_FW = 0;       // + 2 t + piece
constexpr int S2H_FE = 4;       // + t
constexpr int S2H_FRAGS = 6;
constexpr int S2H_IMG_FLOATS = S2H_FRAGS * 256 + 32 + 16;        // fragments, fc1 bias (32), slopes {DataAggregation.activate2, activate1}
constexpr int S2H_TBL = S2H_FRAGS * 512 + 32 + 16;

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

// k_stage2_h2's image: same entry format as build_h2_table (k_pack_h2 writes both)
void build_s2h_table(std::vector<int32_t>& tbl) {
    tbl.assign(S2H_TBL, -1);
    auto put = [&](int f, int i, int kg, int e, int piece, int off) {
        tbl[((size_t)f * 64 + (kg * 16 + i)) * 8 + e] = off < 0 ? -1 : (off | (piece << 28));
    };
    const int W = g_params[W_BP_FC1_W].off;
    for (int t = 0; t < 2; ++t)
        for (int i = 0; i < 16; ++i) {
            const int row = 16 * t + i;
            if (row >= 30) continue;
            for (int kg = 0; kg < 4; ++kg)
                for (int e = 0; e < 8; ++e) {
                    const int ch = 4 * kg + (e & 3);                  // channel inside its 15-wide half
                    if (ch < 15) {
                        const int off = W + row * 33 + (e < 4 ? ch : 15 + ch);
                        put(S2H_FW + 2 * t + 0, i, kg, e, 0, off);
                        put(S2H_FW + 2 * t + 1, i, kg, e, 1, off);
                    }
                    if ((e & 3) < 3) {                                // edge_attr step
                        const int off = W + row * 33 + 30 + (e & 3);
                        if (kg == 0) put(S2H_FE + t, i, kg, e, 0, off);                  // W0 x first / second pieces
                        if (kg == 1 && e < 4) put(S2H_FE + t, i, kg, e, 1, off);          // W1' x (first pieces / 16)
                    }
                }
        }
    int32_t* bias = tbl.data() + (size_t)S2H_FRAGS * 512;
    for (int i = 0; i < 30; ++i) bias[i] = g_params[W_BP_FC1_B].off + i;
    bias[32] = g_params[W_DA_ACT2].off;
    bias[33] = g_params[W_BP_ACT1].off;
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
#ifndef GENIE_ABL_MFMA
#define GENIE_ABL_MFMA GENIE_TUNING      // the run-time MFMA switch costs every fp32 MFMA a branch: -DGENIE_ABL_MFMA=0 for timing other hooks
#endif
#if GENIE_TUNING
__device__ int g_abl_mfma;  // set from the host in tuning builds
#endif
#if GENIE_ABL_MFMA
#define MFMA16(a, b, c) (g_abl_mfma ? ((c) + (a) * (b)) : __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0))
#else
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#endif

// wave-level ordering point for data the lanes of ONE wave exchange through LDS. The DS instructions of a wave execute in issue
// order, so a wavefront-scope fence (compiler ordering only, no s_waitcnt) is enough; the workgroup-scope fence used until round 4
// also drained every outstanding global load (vmcnt(0)) at each exchange.
#define GSYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
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
    const int32_t* ptile;      // irregular product graph: processing order of the kernel's tiles (16 or 32 consecutive product nodes), or null
    int np;                    // c / wv rows are NODE-PLANAR (k_stage1_h2 writes, k_stage2_h2 reads): inside the block of a source node,
                               // chunk q (16 B) of all S stations is contiguous: c [g][8][S] x 16 B, wv [g][4][S] x 16 B
    const unsigned* ea_frag;   // k_stage2_h2: edge_attr as B fragments (k_ea_frag), node-planar [g][2][S] x 16 B, processing order
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

// Irregular product graph: the n tiles of a kernel in processing order (DaArgs / TrArgs / AsArgs .ptile: by the space-filling-curve rank of their source
// node). XCD x (workgroup b runs on XCD b % 8) takes the contiguous chunk [n x / 8, n (x + 1) / 8) of that list and its workgroups
// stride over it, so the rows a tile gathers from neighbouring source nodes are in the L2 that is reading them anyway.
struct PtileIter {
    long long i, end, stride;
    __device__ PtileIter(long long n, int waves_per_wg, int wave) {
        const int nx = (gridDim.x >= 8 && (gridDim.x % 8) == 0) ? 8 : 1;
        const int xcd = blockIdx.x % nx, lb = blockIdx.x / nx, nbx = gridDim.x / nx;
        i = n * xcd / nx + (long long)lb * waves_per_wg + wave;
        end = n * (xcd + 1) / nx;
        stride = (long long)nbx * waves_per_wg;
    }
};
__device__ __forceinline__ long long ptile_at(const int32_t* ptile, long long i) { return ptile != nullptr ? (long long)ptile[i] : i; }

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

#include "stage_kernels.hpp"
#include "assoc_kernels.hpp"
#include "train_front_kernels.hpp"
#include "tail_kernels.hpp"
#include "aux_kernels.hpp"
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
    int2 *r_sta_cw, *r_src_cw;   // the same edges as (column, weight bits) pairs: one 8-byte load per edge (training passes)
    int32_t *rp_sta_rowptr, *rp_src_rowptr; int2 *rp_sta_cw, *rp_src_cw;   // irregular product graph: the reversed PRODUCT-level graphs
    int32_t *ptile16, *ptile32;  // irregular product graph: the tiles of 16 / 32 consecutive product nodes in the ORDER they are processed in
                                 // (by the space-filling-curve rank of their source node): neighbouring source nodes run together on one XCD
    const float *xs_slice, *xs_mask;   // genie_embed_window_split: the (Slice, Mask) buffers whose split rows already sit in the workspace (one-shot)
    const void* xs_ws;
    int xs_mm_copy;            // ... and the copy (slot % GENIE_NBIG at embed time) its message-mask row `mm` was written to
    float *abs_sta, *abs_src;  // use_absolute_pos: [S][4], [G_ext][4] scaled positions; null = off
    unsigned *abs_ts, *abs_tg; // ... their fp16 pieces for k_stage1_h2 (stations in processing order), rebuilt when abs_dirty
    bool abs_dirty;
    int abs_ts_order;          // station order the pieces were built in (1 = processing order)
    // irregular product graph (`use_subgraph`): product-level CSRs, row range of every source node
    bool pcsr;
    bool pcsr_h2;              // ... with at most 8 / 15 neighbours per product node: k_stage1_h2<.., PCSR> applies
    int32_t *p_sta_rowptr, *p_sta_col, *p_src_rowptr, *p_src_col, *seg_rowptr;
    int32_t* p_src_of;         // ... source node of every product node (built on the first genie_assoc_fwd)
    int32_t* p_sta_of = nullptr;   // ... station of every product node (genie_set_subgraph_stations: the device embedding needs it)
    float *mpos_sta, *mpos_src, *ebias_sta, *ebias_src;   // DataAggregationEdges: mean edge features [n,4] and their Linear [n,48]
    bool has_edges;
    // station processing order (genie_set_station_order): internal -> caller's station, its inverse, the station graph in
    // internal labels, the per-station edge terms in internal order; null = the caller's order
    int32_t *sta_perm, *sta_inv, *sta_rowptr_p, *sta_col_p;
    int32_t* sta_ident;        // 0 .. S-1: the station order of the training forward (pre-activations are stored in the caller's order)
    float* ebias_sta_p;
    float* ea_int; const float* ea_user;   // genie_set_static_edge_attr: processing-order copy of the caller's static edge_attr
    float* ea_tmp;             // ... of an edge_attr that is not the registered one (permuted per call)
    int32_t* src_tab;          // [G][16] processing-order table of k_stage1_h2 (null unless kp_uni == 15)
    float* packed_h2;          // f16x2 weight image of k_stage1_h2
    hipStream_t side_stream = nullptr;      // fork / join inside one call (genie_tail_train_bwd: the grid branch beside the query branch); the device's, not owned
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int device = 0;            // the HIP device the context was created on (genie_ctx_destroy drains THAT device)
    int num_cu;
    int seg, bpc1, bpc2;       // sweep segments (env GENIE_SEG), workgroups per CU of the generic stage kernels
    int ks_uni, kp_uni;        // uniform in-degree of the station / source graph, -1 when ragged
    int use_fast;              // the reference's kNN graphs (ks_uni == 8 && kp_uni == 15): the pipelined kernels k_stage1_h2 / k_stage2_ord apply
    int bpc2o;                 // workgroups of k_stage2_ord per CU (its occupancy: three per CU)
    int bpc2h;                 // workgroups of k_stage2_h2 per CU
    int s2_wgmap;              // k_stage2_ord: blocks of 4 source nodes per workgroup (large station counts)
    int bpc1b;                 // workgroups of k_stage1_h2 per CU in the grid (one is resident; more = dynamic balancing by the dispatcher)
    int use_h2;                // the SHAPE admits the f16x2 kernels (k_stage1_h2 / k_stage2_h2): uniform 8 / 15-degree graphs, 24-bit
                               // multiplicands; whether they run is h2_on(): precision mode + the fp16 range guard below
    int prec_mode;             // genie_set_stage_precision: 0 = auto (f16x2 while the range guard holds, else fp32 MFMA), 1 = f16x2, 2 = fp32
    bool range_ok;             // fp16 range guard (k_h2_range, evaluated at every weight commit): every hidden state the f16x2 kernels
                               // split into fp16 pieces is bounded below 60 000 for inputs in [-1, 1], and so is every weight
    float range_act, range_w;  // ... the two bounds it found (largest hidden-state bound, largest weight magnitude incl. the 16 x forms)
    float* d_range; float* h_range;     // device result / pinned host copy of k_h2_range
    unsigned* h_inflag = nullptr;       // host-mapped word: bits of the largest |input| a split pass saw BEYOND what the range guard verified
                                        // (flag_input_range), 0 = none; read by genie_input_range without synchronising
    // k_stage2_h2u: blocks of adjacent source nodes with the union of their neighbour rows, per launched range [gi_begin, gi_end) of the
    // processing order (the whole grid; the sharded path's four static sub-ranges): built on first use from the host copy of src_tab
    struct S2uTables { void* blocks; int32_t* xcd0; int nblk; };
    std::map<std::pair<int, int>, S2uTables> s2u;
    std::vector<int32_t> tab_host;
    int s2u_off;               // tuning: k_stage2_h2 (every source row through the texture path) where k_stage2_h2u applies
    int32_t* d_s2htbl;         // k_pack_h2 source table of k_stage2_h2's image
    float* packed_s2h;         // f16x2 weight image of k_stage2_h2 (Bipartite_ReadIn.fc1)
    unsigned *ea_frag, *ea_frag_tmp;    // edge_attr as B fragments of k_stage2_h2 (k_ea_frag): of the registered static edge_attr / of any other one
    bool tables_shared = false;   // the graph-independent tables below belong to the device's template context (model_tables): not freed here
    bool ws_np;                // layout of the c / wv rows the last stage 1 left in the workspace: node-planar (DaArgs.np) or rows
    std::vector<const void*> lds_attr_done;   // kernels whose MaxDynamicSharedMemorySize was raised for THIS context's device (the attribute
                               // is per device: a process-wide flag would skip the second GPU of a multi-GPU process)
    int xs_sta_order;          // genie_embed_window_split: the station-order state its split rows were written under
    int sign_input;            // genie_set_sign_input(1): the embedding tags every feature with the sign of the series' negative slope
    int no_phase;              // genie_set_phase_types(0): the embedding zeroes the phase-informed columns of Slice / Mask
    int tail_f32;              // genie_set_tail_precision(0): the G-sized tail on fp32 MFMA chains (default: fp64 chains, tail_kernels.hpp)
    int tail_train;            // set for the duration of a training forward: its tail keeps the fp32 chains (the backward recomputes with them)
    // workspace offsets (floats)
    size_t o_xs, o_mm, o_c, o_wu, o_wv, o_part, o_sa0, o_sa1, o_bip, o_gpart, o_pj0, o_pj1, o_cv, ws_floats;
    size_t slot_stride;        // the G-sized buffers (o_part ... o_cv) exist GENIE_NSLOT times; `slot` selects the copy
    size_t big_stride;         // so do the P-sized stage-1 -> stage-2 buffers (c, wu, wv)
    int tail_cu_ro, tail_cu_sa; // grid caps (workgroups) of the read-out / SpatialAggregation kernels of the G-sized tail
    int slot;                  // lets window i+1's stage 1/2 overlap window i's G-sized kernels on another stream
};

namespace {

// raise a kernel's dynamic-LDS limit once per context (= per device)
int raise_lds_limit(genie_ctx* c, const void* kern, int bytes) {
    for (const void* k : c->lds_attr_done) if (k == kern) return GENIE_OK;
    HIP_TRY(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    c->lds_attr_done.push_back(kern);
    return GENIE_OK;
}

// The station processing order is honoured by k_split_rows / the embedding's split rows, k_stage1_h2 (through the relabelled
// station graph) and k_stage2_ord: active only while those are the kernels that run (use_absolute_pos together with the
// edge-feature variant takes the generic stage-1 kernel).
// The f16x2 kernels run when the shape admits them AND the precision mode says so: auto = while the fp16 range guard holds for the
// committed weights (k_h2_range), else the fp32-MFMA kernels take over -- no environment variable, no non-finite output.
// rows of the static edge-feature tables (DataAggregationEdges): per station / per source node, per product node on an irregular graph
long long edge_rows_sta(const genie_ctx* c) { return c->pcsr ? c->P : c->S; }
long long edge_rows_src(const genie_ctx* c) { return c->pcsr ? c->P : c->G; }
// scheduling overrides of the tuning builds (-DGENIE_TUNING=1: tools/tune.py, tools/*_sweep.sh); the product library reads no
// environment variable
const char* tune_env(const char* name) {
#if GENIE_TUNING
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}
// stage 2 of an irregular product graph with the station sum folded in (k_stage2_pseg); tuning builds can switch back for A/B runs
bool pseg_on() {
    static const char* e = tune_env("GENIE_S2_PSEG");
    return !e || atoi(e) != 0;
}
bool h2_on(const genie_ctx* c) { return c->use_h2 && (c->prec_mode == 1 || (c->prec_mode == 0 && c->range_ok)); }
bool pcsr_h2_on(const genie_ctx* c) { return c->pcsr_h2 && (c->prec_mode == 1 || (c->prec_mode == 0 && c->range_ok)); }
int part_T(const genie_ctx* c) { return c->pcsr ? 1 : c->T; }      // rows of station-sum partials per source node in `part`
bool abs_generic(const genie_ctx* c) { return c->abs_sta != nullptr && (c->has_edges || !h2_on(c)); }
bool sta_order_on(const genie_ctx* c) {
    return c->sta_perm != nullptr && !c->pcsr && h2_on(c) && !abs_generic(c) && !c->force_generic;
}

// k_stage2_h2 is the stage 2 of the production configuration (the reference's kNN graphs in station processing order); the stage 1
// that feeds it writes c / wv node-planar (DaArgs.np). Both launch sites ask this.
// the G-sized tail of inference calls runs its Linears as fp64 MFMA chains (WIDE kernels) unless told otherwise
bool tail_wide(const genie_ctx* c) { return !c->tail_f32 && !c->tail_train; }
// training forward on the reference's kNN graphs with the f16x2 kernels in range: stage 2 is the production kernel (k_stage2_h2u,
// SAVE) in the CALLER's station order, so the stage 1 of a training call writes c / wv node-planar too
bool train_h2u_on(const genie_ctx* c) {
    return c->force_generic && !c->pcsr && c->train_save != nullptr && c->use_fast && h2_on(c) && c->src_tab != nullptr && !c->tab_host.empty() &&
           !c->s2u_off && !abs_generic(c);
}
bool s2h_on(const genie_ctx* c) {
#if GENIE_TUNING
    { static const bool old_pair = getenv("GENIE_S2_OLD") != nullptr; if (old_pair) return false; }   // A/B: row layout + k_stage2_ord
#endif
    return c->use_fast && sta_order_on(c);
}

// largest |Slice| / |Mask| entry for which the committed weights keep every hidden state of the f16x2 kernels below the fp16 range: the
// guard's bound holds for inputs in [-1, 1] and grows at most linearly with them (and the inputs themselves are split into fp16 pieces)
float input_limit(const genie_ctx* c) {
    const float lim = 60000.f / std::max(c->range_act, 1.f);
    return std::max(1.f, std::min(lim, 60000.f));
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
    HIP_TRY(gmalloc((void**)dst, n * sizeof(T)));
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
        HIP_TRY(gmalloc(&c->d_packplans, sizeof(PackPlan) * NPLAN));
        HIP_TRY(hipMemcpy(c->d_packplans, pl.data(), sizeof(PackPlan) * NPLAN, hipMemcpyHostToDevice));
        c->pack_blocks = blocks;
    }
    k_pack_all<<<c->pack_blocks, 256, 0, st>>>(c->raw, (const PackPlan*)c->d_packplans, NPLAN);
    k_pack_h2<<<(H2_FRAGS * 64 + H2_NBIAS * 32 + 16 + 255) / 256, 256, 0, st>>>(c->raw, c->d_h2tbl, c->packed_h2, H2_FRAGS,
                                                                               H2_NBIAS * 32 + 16);
    k_pack_h2<<<(S2H_FRAGS * 64 + 32 + 16 + 255) / 256, 256, 0, st>>>(c->raw, c->d_s2htbl, c->packed_s2h, S2H_FRAGS, 32 + 16);
    if (c->has_edges) {
        const long long ns = edge_rows_sta(c), ng = edge_rows_src(c);
        k_edge_bias<<<(unsigned)((ns * 48 + 255) / 256), 256, 0, st>>>(c->raw, g_params[W_DA_L1T12_P].off, g_params[W_DA_L2T12_P].off,
                                                                      c->mpos_sta, (int)ns, c->ebias_sta);
        k_edge_bias<<<(unsigned)((ng * 48 + 255) / 256), 256, 0, st>>>(c->raw, g_params[W_DA_L1T22_P].off, g_params[W_DA_L2T22_P].off,
                                                                      c->mpos_src, (int)ng, c->ebias_src);
        if (c->sta_perm) {
            if (!c->ebias_sta_p) HIP_TRY(gmalloc((void**)&c->ebias_sta_p, sizeof(float) * 48 * (size_t)c->S));
            k_permute_sta_rows<<<(c->S * 48 + 255) / 256, 256, 0, st>>>(c->ebias_sta, c->S, 48, c->sta_inv, c->S, c->ebias_sta_p);
        }
    }
    HIP_TRY(hipGetLastError());
    if (c->use_h2 || c->pcsr_h2) {
        // fp16 range guard of the f16x2 kernels: rigorous per-channel bounds of every hidden state they split into fp16 pieces, from
        // the weights just committed (inputs in [-1, 1]; the absolute-position / edge-term tables by their actual maxima). One tiny
        // launch and a 16-byte read-back per weight commit: the only host synchronisation of the library, and not in the window loop.
        RangeArgs ra;
        memset(&ra, 0, sizeof(ra));
        ra.raw = c->raw;
        const int ids[RG_N] = {W_DA_INIT_W, W_DA_INIT_B, W_DA_INIT_ABS, W_DA_L1T12_W, W_DA_L1T12_B, W_DA_L1T22_W, W_DA_L1T22_B, W_DA_L2T11_W,
                               W_DA_L2T11_B, W_DA_L2T21_W, W_DA_L2T21_B, W_DA_L2T12_W, W_DA_L2T12_B, W_DA_L2T22_W, W_DA_L2T22_B, W_DA_ACT,
                               W_DA_ACT11, W_DA_ACT12, W_DA_ACT1, W_DA_ACT21, W_DA_ACT22, W_DA_ACT2, W_BP_FC1_W};
        for (int k = 0; k < RG_N; ++k) ra.off[k] = g_params[ids[k]].off;
        ra.abs_sta = c->abs_sta; ra.n_abs_sta = c->abs_sta ? (c->pcsr ? c->P : (long long)c->S) * 4 : 0;
        ra.abs_src = c->abs_src; ra.n_abs_src = c->abs_src ? (c->pcsr ? c->P : (long long)c->G_ext) * 4 : 0;
        ra.eb_sta = c->has_edges ? c->ebias_sta : nullptr; ra.n_eb_sta = c->has_edges ? edge_rows_sta(c) * 48 : 0;
        ra.eb_src = c->has_edges ? c->ebias_src : nullptr; ra.n_eb_src = c->has_edges ? edge_rows_src(c) * 48 : 0;
        {   // long tables: partial maxima by many workgroups first (d_range[4 ..] holds 4 x RG_PART of them)
            const float** tp[4] = {&ra.abs_sta, &ra.abs_src, &ra.eb_sta, &ra.eb_src};
            long long* tn[4] = {&ra.n_abs_sta, &ra.n_abs_src, &ra.n_eb_sta, &ra.n_eb_src};
            for (int k = 0; k < 4; ++k)
                if (*tp[k] && *tn[k] > 4096) {
                    float* part = c->d_range + 4 + k * RG_PART;
                    k_tab_absmax<<<RG_PART, 256, 0, st>>>(*tp[k], *tn[k], part);
                    *tp[k] = part; *tn[k] = RG_PART;
                }
        }
        ra.out = c->d_range;
        k_h2_range<<<1, 256, 0, st>>>(ra);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(c->h_range, c->d_range, sizeof(float) * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        c->range_act = c->h_range[0]; c->range_w = c->h_range[1];
        c->range_ok = c->h_range[2] != 0.f;
    }
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
        HIP_TRY(gmalloc((void**)&c->d_acc[s], sizeof(AccDesc) * acc[s].size()));
        HIP_TRY(hipMemcpy(c->d_acc[s], acc[s].data(), sizeof(AccDesc) * acc[s].size(), hipMemcpyHostToDevice));
        HIP_TRY(gmalloc((void**)&c->d_vec[s], sizeof(VecDesc) * vec[s].size()));
        HIP_TRY(hipMemcpy(c->d_vec[s], vec[s].data(), sizeof(VecDesc) * vec[s].size(), hipMemcpyHostToDevice));
        HIP_TRY(gmalloc((void**)&c->d_sc[s], sizeof(int32_t) * sc[s].size()));
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
        HIP_TRY(gmalloc((void**)&c->d_acc[s], sizeof(AccDesc) * std::max<size_t>(1, acc[s].size())));
        if (!acc[s].empty()) HIP_TRY(hipMemcpy(c->d_acc[s], acc[s].data(), sizeof(AccDesc) * acc[s].size(), hipMemcpyHostToDevice));
        HIP_TRY(gmalloc((void**)&c->d_vec[s], sizeof(VecDesc) * std::max<size_t>(1, vec[s].size())));
        if (!vec[s].empty()) HIP_TRY(hipMemcpy(c->d_vec[s], vec[s].data(), sizeof(VecDesc) * vec[s].size(), hipMemcpyHostToDevice));
        HIP_TRY(gmalloc((void**)&c->d_sc[s], sizeof(int32_t) * std::max<size_t>(1, sc[s].size())));
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


namespace {
// The tables of a context that do not depend on its graph or its weights -- the stage plans (host), their step / bias / scalar descriptors,
// the gradient maps of the backward passes and the source tables of the f16x2 weight images (device) -- are built ONCE per device and
// process on a template context that is never destroyed; every context points at them (`tables_shared`). Building them per context was
// ~45 blocking host-to-device copies = most of the 2.4 ms genie_ctx_create cost per training sample (train_GENIE_model.py:1722-1786).
int init_model_tables(genie_ctx* c) {
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
    { int rc_tr; if ((rc_tr = build_grad_maps(c))) return rc_tr; if ((rc_tr = build_tail_grad_maps(c))) return rc_tr; }
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
    }
    {
        std::vector<int32_t> tbl;
        build_h2_table(tbl);
        HIP_TRY(hipMalloc((void**)&c->d_h2tbl, sizeof(int32_t) * tbl.size()));
        HIP_TRY(hipMemcpy(c->d_h2tbl, tbl.data(), sizeof(int32_t) * tbl.size(), hipMemcpyHostToDevice));
        build_s2h_table(tbl);
        HIP_TRY(hipMalloc((void**)&c->d_s2htbl, sizeof(int32_t) * tbl.size()));
        HIP_TRY(hipMemcpy(c->d_s2htbl, tbl.data(), sizeof(int32_t) * tbl.size(), hipMemcpyHostToDevice));
    }
    return GENIE_OK;
}

int model_tables(genie_ctx** out) {
    static std::map<int, genie_ctx*> tmpl;
    static std::mutex mu;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    auto it = tmpl.find(dev);
    if (it == tmpl.end()) {
        genie_ctx* t = new genie_ctx();
        int rc = init_model_tables(t);
        if (rc) return rc;              // (the half-built template leaks a few KB; the process cannot create contexts anyway)
        it = tmpl.emplace(dev, t).first;
    }
    *out = it->second;
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
    int rc;
    if ((rc = dev_copy(&c->sta_rowptr, sta_rowptr, (size_t)n_sta + 1))) return rc;
    if ((rc = dev_copy(&c->sta_col, sta_col, (size_t)e_sta))) return rc;
    if ((rc = dev_copy(&c->src_rowptr, src_rowptr, (size_t)n_grid + 1))) return rc;
    if ((rc = dev_copy(&c->src_col, src_col, (size_t)e_src))) return rc;
    if (grid_order) {
        if ((rc = dev_copy(&c->order, grid_order, (size_t)n_grid))) return rc;
    } else {
        std::vector<int32_t> id(n_grid);
        for (int i = 0; i < n_grid; ++i) id[i] = i;
        HIP_TRY(gmalloc((void**)&c->order, sizeof(int32_t) * n_grid));
        HIP_TRY(hipMemcpy(c->order, id.data(), sizeof(int32_t) * n_grid, hipMemcpyHostToDevice));
    }
    HIP_TRY(gmalloc((void**)&c->outdeg, sizeof(int32_t) * (size_t)n_grid_ext));
    HIP_TRY(hipMemset(c->outdeg, 0, sizeof(int32_t) * (size_t)n_grid_ext));
    if (e_src > 0) k_outdeg<<<(e_src + 255) / 256, 256>>>(c->src_col, e_src, c->outdeg);
    HIP_TRY(hipGetLastError());
    HIP_TRY(gmalloc((void**)&c->raw, sizeof(float) * g_raw_total));
    HIP_TRY(hipMemset(c->raw, 0, sizeof(float) * g_raw_total));
    {   // graph- and weight-independent tables: shared with the device's template context (model_tables)
        genie_ctx* tm = nullptr;
        if ((rc = model_tables(&tm))) return rc;
        for (int s = 0; s < NPLAN; ++s) {
            c->plan[s] = tm->plan[s];
            c->d_steps[s] = tm->d_steps[s]; c->d_bias[s] = tm->d_bias[s]; c->d_scal[s] = tm->d_scal[s];
            HIP_TRY(gmalloc((void**)&c->packed[s], sizeof(float) * c->plan[s].packed_floats()));
        }
        for (int s = 0; s < NTM; ++s) {
            c->d_acc[s] = tm->d_acc[s]; c->d_vec[s] = tm->d_vec[s]; c->d_sc[s] = tm->d_sc[s];
            c->n_acc[s] = tm->n_acc[s]; c->n_vec[s] = tm->n_vec[s]; c->n_sc[s] = tm->n_sc[s];
        }
        c->d_h2tbl = tm->d_h2tbl; c->d_s2htbl = tm->d_s2htbl;
        c->tables_shared = true;
        HIP_TRY(gmalloc((void**)&c->packed_h2, sizeof(float) * H2_IMG_FLOATS));
        HIP_TRY(gmalloc((void**)&c->packed_s2h, sizeof(float) * S2H_IMG_FLOATS));
        HIP_TRY(gmalloc((void**)&c->d_range, sizeof(float) * (4 + 4 * RG_PART)));
        HIP_TRY(hipHostMalloc((void**)&c->h_range, sizeof(float) * 4));
        HIP_TRY(hipHostMalloc((void**)&c->h_inflag, 64, hipHostMallocMapped));
        c->h_inflag[0] = 0u; c->h_inflag[1] = 0u;
        c->range_ok = true; c->prec_mode = 0;
    }
    c->mpos_sta = c->mpos_src = c->ebias_sta = c->ebias_src = nullptr;
    c->has_edges = false;
    c->xs_slice = c->xs_mask = nullptr; c->xs_ws = nullptr; c->xs_mm_copy = 0;
    c->abs_sta = c->abs_src = nullptr; c->abs_ts = c->abs_tg = nullptr; c->abs_dirty = false; c->abs_ts_order = 0;
    c->r_sta_rowptr = c->r_sta_col = c->r_src_rowptr = c->r_src_col = nullptr;
    c->sta_perm = c->sta_inv = c->sta_rowptr_p = c->sta_col_p = nullptr; c->ebias_sta_p = nullptr; c->sta_ident = nullptr;
    c->ea_int = c->ea_tmp = nullptr; c->ea_user = nullptr;
    c->r_sta_w = c->r_src_w = nullptr; c->r_sta_cw = c->r_src_cw = nullptr;
    c->rp_sta_rowptr = c->rp_src_rowptr = nullptr; c->rp_sta_cw = c->rp_src_cw = nullptr; c->ptile16 = c->ptile32 = nullptr;
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
        HIP_TRY(gmalloc((void**)&c->src_tab, sizeof(int32_t) * tab.size()));
        HIP_TRY(hipMemcpy(c->src_tab, tab.data(), sizeof(int32_t) * tab.size(), hipMemcpyHostToDevice));
        c->tab_host = tab;
    }
    c->dirty = true;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    c->device = dev;
    {   // (hipGetDeviceProperties fills a 1.5-KB struct through the driver: once per device and process)
        static std::map<int, int> cus;
        static std::mutex mu;
        std::lock_guard<std::mutex> lk(mu);
        auto it = cus.find(dev);
        if (it == cus.end()) {
            hipDeviceProp_t prop;
            HIP_TRY(hipGetDeviceProperties(&prop, dev));
            it = cus.emplace(dev, prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256).first;
        }
        c->num_cu = it->second;
    }
    {
        const char* e;
        // scheduling segments: node-major sweeps (1) at config 2; at 2000 stations station-tile-major sweeps over segments of 16
        // source nodes keep the source-neighbour rows of stage 1 in L2 (config 4 on one GPU: stage 1 25.8 -> 24.8 ms)
        c->seg = (e = tune_env("GENIE_SEG")) ? atoi(e) : (n_sta >= 1024 ? 16 : 1);
        // G-sized tail: few, long-lived workgroups. Next to the persistent P-sized kernels a tail workgroup only runs when one
        // of theirs retires and keeps that CU until it ends, so what the tail costs the main stream is its CU-time = workgroups x
        // duration, and most of a short tail workgroup is fixed cost (its LDS weight image). Two 62-KB read-out workgroups per CU
        // hide the gather latency of k_readout_m<1> (window 0.769 -> 0.765 ms).
        c->tail_cu_ro = c->num_cu * 2;          // genie_set_tail_grid
        c->tail_cu_sa = c->num_cu * 2;
        // persistent grids: exactly as many workgroups as are co-resident (a larger grid runs in two uneven rounds)
        int occ1 = 0, occ2 = 0, occo = 0, occh = 0;
        {   // (four driver queries: once per device and process)
            static std::map<int, std::array<int, 4>> occ;
            static std::mutex mu;
            std::lock_guard<std::mutex> lk(mu);
            auto it = occ.find(dev);
            if (it == occ.end()) {
                std::array<int, 4> o{};
                HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&o[0], k_stage1, 256, 0));
                HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&o[1], k_stage2, 256, 0));
                HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&o[2], k_stage2_ord<8, 15, false>, 256, 0));
                HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&o[3], k_stage2_h2<false, false>, 256, 0));
                it = occ.emplace(dev, o).first;
            }
            occ1 = it->second[0]; occ2 = it->second[1]; occo = it->second[2]; occh = it->second[3];
        }
        c->bpc1 = std::max(1, occ1);
        c->bpc2 = (e = tune_env("GENIE_BPC2")) ? atoi(e) : std::max(1, occ2);
        // Large station counts (config 4: 2000 stations, 128 KB of wu / wv rows per source node): the gathers leave L2, and what
        // pays is locality, not concurrency: blocks of 4 adjacent source nodes per workgroup (one node per wave: the four waves
        // share half of their source rows, every wu block is gathered on one CU) and two workgroups per CU. Config 4 on one
        // GPU: stage 2 16.1 -> 11.8 ms (three workgroups, interleaved items: 16.1; two: 14.8; block map alone: 13.1). At 200
        // stations the same settings lose (0.266 -> 0.268 ms), hence by size.
        c->s2_wgmap = (e = tune_env("GENIE_S2_WGMAP")) ? (atoi(e) != 0) : (n_sta >= 1024);
        {
            // two workgroups per CU: as fast as three (0.2367 / 0.2370 ms) with 8 % less fabric traffic (FETCH_SIZE 5.18e5 vs 5.62e5 KB)
            c->bpc2h = (e = tune_env("GENIE_BPC2")) ? atoi(e) : std::min(2, std::max(1, occh));
            c->s2u_off = tune_env("GENIE_S2_NOUNION") != nullptr;
        }
        c->bpc2o = (e = tune_env("GENIE_BPC2")) ? atoi(e) : std::min(2, std::max(1, occo));     // round 3, after the f16x2 stage 1: 2 beat 3 at 200 stations too (window 0.593 -> 0.588 ms)
        // the reference's kNN graphs (8 station / 15 source neighbours everywhere): pipelined kernels k_stage1_h2 / k_stage2_ord
        c->use_fast = c->ks_uni == 8 && c->kp_uni == 15;
        // f16x2 kernels: 24-bit multiplicands (64-bit row offsets are a template variant). Whether they run: h2_on()
        c->use_h2 = (c->use_fast && n_grid_ext < (1 << 24) && (long long)n_sta * XROW < (1 << 24));
        // one workgroup per CU = fully persistent: with the f16x2 stage 1 and tails batched 16 windows at a time this beats the 4
        // per CU of rounds 1-2 (window 0.5937 -> 0.5796 and 0.5663 -> 0.5532 ms on two boxes; 2: 0.5637, 3: 0.561, 6: 0.5726,
        // 8: 0.579; the weight image is staged once per CU instead of four times)
        c->bpc1b = 1;
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
    HIP_TRY(gmalloc((void**)&zeros, sizeof(int32_t) * ((size_t)n_sta + 1)));
    genie_ctx* c = nullptr;
    int rc = hipMemset(zeros, 0, sizeof(int32_t) * ((size_t)n_sta + 1)) == hipSuccess
                 ? genie_ctx_create(&c, n_sta, n_grid, n_grid, zeros, nullptr, src_rowptr, src_col, grid_order, scale_rel)
                 : fail(GENIE_ERR_HIP, "genie_ctx_create_subgraph: hipMemset failed");
    (void)gfree(zeros);
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
        c->pcsr_h2 = m1 <= 8 && m2 <= 15;
    }
    {   // processing order of the tiles: the product nodes stay where the caller put them (grouped by source node in HIS order), but
        // the tiles are taken in the space-filling-curve order of their source nodes, one contiguous chunk of that list per XCD, so
        // the rows a tile gathers (neighbouring source nodes) are being read by the same L2 at about the same time
        std::vector<int32_t> seg((size_t)n_grid + 1), ord((size_t)n_grid), rank((size_t)n_grid, 0);
        HIP_TRY(hipMemcpy(seg.data(), seg_rowptr, sizeof(int32_t) * seg.size(), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(ord.data(), c->order, sizeof(int32_t) * ord.size(), hipMemcpyDeviceToHost));
        for (int i = 0; i < n_grid; ++i) rank[ord[i]] = i;
        std::vector<int32_t> src_of((size_t)n_prod);
        for (int g = 0; g < n_grid; ++g) for (int32_t pr = seg[g]; pr < seg[g + 1]; ++pr) src_of[pr] = g;
        for (int w = 16; w <= 32; w += 16) {
            const int nt = (int)((n_prod + w - 1) / w);
            std::vector<int32_t> t((size_t)nt);
            for (int i = 0; i < nt; ++i) t[i] = i;
            std::stable_sort(t.begin(), t.end(), [&](int32_t x, int32_t y) { return rank[src_of[(size_t)x * w]] < rank[src_of[(size_t)y * w]]; });
            int32_t** dst = w == 16 ? &c->ptile16 : &c->ptile32;
            HIP_TRY(gmalloc((void**)dst, sizeof(int32_t) * (size_t)std::max(nt, 1)));
            HIP_TRY(hipMemcpy(*dst, t.data(), sizeof(int32_t) * (size_t)nt, hipMemcpyHostToDevice));
        }
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
        (void)hipDeviceSynchronize();        // pooled blocks are reused without the driver's implicit drain
        (void)gfree(c->abs_sta); (void)gfree(c->abs_src); (void)gfree(c->abs_ts); (void)gfree(c->abs_tg);
        c->abs_sta = c->abs_src = nullptr; c->abs_ts = c->abs_tg = nullptr;
        return GENIE_OK;
    }
    // irregular product graph: both arguments are [n_prod, 3] (the station's / the source node's position of every product node) and the
    // tables are per product node: a neighbour (a product-node id) is looked up like the node itself
    const long long ns = c->pcsr ? c->P : c->S, ng = c->pcsr ? c->P : c->G_ext;
    if (!c->abs_sta) {
        HIP_TRY(gmalloc((void**)&c->abs_sta, sizeof(float) * 4 * (size_t)ns));
        HIP_TRY(gmalloc((void**)&c->abs_src, sizeof(float) * 4 * (size_t)ng));
    }
    const float inv = 1.f / (3.f * c->scale_rel);
    hipStream_t st = (hipStream_t)stream;
    k_abs_table<<<(unsigned)((ns * 4 + 255) / 256), 256, 0, st>>>(pos_sta, (int)ns, inv, c->abs_sta);
    k_abs_table<<<(unsigned)((ng * 4 + 255) / 256), 256, 0, st>>>(pos_src, (int)ng, inv, c->abs_src);
    HIP_TRY(hipGetLastError());
    c->abs_dirty = true;
    c->dirty = true;             // the fp16 range guard bounds the hidden states with the maxima of these tables: re-evaluate it
    return GENIE_OK;
}

int genie_set_edge_features(genie_ctx* c, const float* pos_sta, const float* pos_src, void* stream) {
    if (!c) return fail(GENIE_ERR_ARG, "genie_set_edge_features: null context");
    hipStream_t st = (hipStream_t)stream;
    if (!pos_sta || !pos_src) {      // back to plain DataAggregation
        c->has_edges = false;
        return GENIE_OK;
    }
    // irregular product graph: the mean runs over the PRESENT neighbours of a product node, so both tables are per product node and the
    // positions arrive per product node too ([n_prod, 3]: the station's, the source node's)
    const long long ns = edge_rows_sta(c), ng = edge_rows_src(c);
    if (!c->mpos_sta) {
        HIP_TRY(gmalloc((void**)&c->mpos_sta, sizeof(float) * 4 * (size_t)ns));
        HIP_TRY(gmalloc((void**)&c->mpos_src, sizeof(float) * 4 * (size_t)ng));
        HIP_TRY(gmalloc((void**)&c->ebias_sta, sizeof(float) * 48 * (size_t)ns));
        HIP_TRY(gmalloc((void**)&c->ebias_src, sizeof(float) * 48 * (size_t)ng));
    }
    if (c->pcsr) {
        k_edge_feat<<<(unsigned)((ns + 255) / 256), 256, 0, st>>>(c->p_sta_rowptr, c->p_sta_col, (int)ns, pos_sta, c->scale_rel, c->mpos_sta);
        k_edge_feat<<<(unsigned)((ng + 255) / 256), 256, 0, st>>>(c->p_src_rowptr, c->p_src_col, (int)ng, pos_src, c->scale_rel, c->mpos_src);
    } else {
        k_edge_feat<<<(c->S + 255) / 256, 256, 0, st>>>(c->sta_rowptr, c->sta_col, c->S, pos_sta, c->scale_rel, c->mpos_sta);
        k_edge_feat<<<(c->G + 255) / 256, 256, 0, st>>>(c->src_rowptr, c->src_col, c->G, pos_src, c->scale_rel, c->mpos_src);
    }
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
    void* old[] = {c->sta_perm, c->sta_inv, c->sta_rowptr_p, c->sta_col_p, c->ea_int, c->ea_tmp, c->ea_frag, c->ea_frag_tmp};
    bool any_old = false;
    for (void* q : old) any_old = any_old || q != nullptr;
    if (any_old) (void)hipDeviceSynchronize();      // (a first call on a fresh context has nothing in flight and nothing to free)
    for (void* q : old) (void)gfree(q);
    c->sta_perm = c->sta_inv = c->sta_rowptr_p = c->sta_col_p = nullptr;
    c->ea_int = c->ea_tmp = nullptr; c->ea_user = nullptr; c->ea_frag = c->ea_frag_tmp = nullptr;
    c->xs_slice = c->xs_mask = nullptr; c->xs_ws = nullptr;      // split rows of an embedding made under the previous order are void
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
    HIP_TRY(gmalloc((void**)&c->sta_perm, sizeof(int32_t) * S));
    HIP_TRY(gmalloc((void**)&c->sta_inv, sizeof(int32_t) * S));
    HIP_TRY(gmalloc((void**)&c->sta_rowptr_p, sizeof(int32_t) * ((size_t)S + 1)));
    HIP_TRY(gmalloc((void**)&c->sta_col_p, sizeof(int32_t) * std::max<size_t>(E, 1)));
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
    if (!c->use_h2) return GENIE_OK;      // only k_stage2_h2 honours the station processing order in stage 2
    // k_stage2_h2 reads the static edge_attr as ready-made B fragments (32 B per product node, processing order)
    if (!c->ea_frag) HIP_TRY(gmalloc((void**)&c->ea_frag, 32 * (size_t)c->P));
    k_ea_frag<<<(unsigned)((c->P + 255) / 256), 256, 0, (hipStream_t)stream>>>(edge_attr, c->P, c->S, c->sta_perm, c->ea_frag);
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
    int cur = 0;
    (void)hipGetDevice(&cur);
    if (cur != c->device) (void)hipSetDevice(c->device);      // (a Python finaliser may run with another device current)
    (void)hipDeviceSynchronize();        // one drain for every block below: the pool hands them out again without one
    (void)gfree(c->d_packplans);
    if (c->tables_shared) {
        for (int s = 0; s < NPLAN; ++s) c->d_steps[s] = nullptr, c->d_bias[s] = nullptr, c->d_scal[s] = nullptr;
        for (int s = 0; s < NTM; ++s) c->d_acc[s] = nullptr, c->d_vec[s] = nullptr, c->d_sc[s] = nullptr;
        c->d_h2tbl = c->d_s2htbl = nullptr;
    }
    for (int s = 7; s < NPLAN; ++s) { (void)gfree(c->d_steps[s]); (void)gfree(c->d_bias[s]); (void)gfree(c->d_scal[s]); (void)gfree(c->packed[s]); }
    for (int s = 3; s < NTM; ++s) { (void)gfree(c->d_acc[s]); (void)gfree(c->d_vec[s]); (void)gfree(c->d_sc[s]); }
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
                    c->r_sta_rowptr, c->r_sta_col, c->r_src_rowptr, c->r_src_col, c->r_sta_w, c->r_src_w, c->r_sta_cw, c->r_src_cw, c->rp_sta_rowptr, c->rp_src_rowptr, c->rp_sta_cw, c->rp_src_cw, c->ptile16, c->ptile32,
                    c->sta_perm, c->sta_inv, c->sta_rowptr_p, c->sta_col_p, c->ebias_sta_p, c->ea_int, c->ea_tmp, c->sta_ident,
                    c->d_s2htbl, c->packed_s2h, c->ea_frag, c->ea_frag_tmp, c->d_range, c->abs_ts, c->abs_tg, c->p_src_of, c->p_sta_of};
    for (void* p : ptrs) (void)gfree(p);
    if (c->h_range) (void)hipHostFree(c->h_range);
    if (c->h_inflag) (void)hipHostFree(c->h_inflag);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    for (auto& kv : c->s2u) { (void)gfree(kv.second.blocks); (void)gfree(kv.second.xcd0); }
    if (cur != c->device) (void)hipSetDevice(cur);
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
    c->ws_np = false;
    float* dbg_tmp = nullptr;
    if ((dbg_h0 || dbg_h1) && sta_order_on(c)) {
        HIP_TRY(gmalloc((void**)&dbg_tmp, sizeof(float) * 90 * (size_t)c->P));
        if (dbg_h0) a.dbg_h0 = dbg_tmp;
        if (dbg_h1) a.dbg_h1 = dbg_tmp + c->P * 30;
    }
    if (((c->force_generic && !h2_on(c)) || abs_generic(c)) && !c->pcsr) {   // use_absolute_pos, training on other graph shapes: generic kernel (64-bit safe, any graph)
        if (n_tiles) k_stage1<<<da_grid(c, n_tiles, c->bpc1), 256, 0, st>>>(a);
    } else if (c->pcsr && pcsr_h2_on(c) && !(c->abs_sta && c->has_edges)) {      // (both options at once: the generic fp32-MFMA kernel below)
        unsigned* xs = (unsigned*)((float*)ws + c->o_xs);
        k_split_rows<<<(unsigned)((c->P + 255) / 256), 256, 0, st>>>(slice, mask, c->P, xs, nullptr, c->S, nullptr, input_limit(c), c->h_inflag);
        a.xs = xs; a.packed = c->packed_h2; a.xs_plane = c->P * (long long)XPC;
        const long long nitems = (c->P + 31) / 32;
        const int grid = (int)std::max<long long>(8, std::min<long long>((nitems + H2_THREADS / 64 - 1) / (H2_THREADS / 64), (long long)c->num_cu * c->bpc1b) / 8 * 8);
        const bool bigp = c->P * XROW >= (1ll << 32);
        a.ptile = c->ptile32;
        if (c->abs_sta) {      // position pieces per product node (genie_set_absolute_pos on a subgraph context)
            if (c->abs_dirty || !c->abs_ts) {
                if (!c->abs_ts) {
                    HIP_TRY(gmalloc((void**)&c->abs_ts, 16 * (size_t)c->P));
                    HIP_TRY(gmalloc((void**)&c->abs_tg, 16 * (size_t)c->P));
                }
                k_abs_pieces<<<(unsigned)((c->P + 255) / 256), 256, 0, st>>>(c->abs_sta, nullptr, (int)c->P, c->abs_ts);
                k_abs_pieces<<<(unsigned)((c->P + 255) / 256), 256, 0, st>>>(c->abs_src, nullptr, (int)c->P, c->abs_tg);
                c->abs_dirty = false; c->abs_ts_order = 0;
            }
            a.abs_ts = c->abs_ts; a.abs_tg = c->abs_tg;
            if (bigp) k_stage1_h2<8, 15, false, true, true, true><<<grid, H2_THREADS, 0, st>>>(a);
            else k_stage1_h2<8, 15, false, false, true, true><<<grid, H2_THREADS, 0, st>>>(a);
        } else if (c->has_edges) {
            if (bigp) k_stage1_h2<8, 15, true, true, false, true><<<grid, H2_THREADS, 0, st>>>(a);
            else k_stage1_h2<8, 15, true, false, false, true><<<grid, H2_THREADS, 0, st>>>(a);
        } else {
            if (bigp) k_stage1_h2<8, 15, false, true, false, true><<<grid, H2_THREADS, 0, st>>>(a);
            else k_stage1_h2<8, 15, false, false, false, true><<<grid, H2_THREADS, 0, st>>>(a);
        }
    } else if (c->pcsr) {
        const long long ntiles = (c->P + 15) / 16;
        a.ptile = c->ptile16;
        const long long gw = std::min<long long>((ntiles + 3) / 4, (long long)c->num_cu * c->bpc1);
        k_stage1_pcsr<<<(int)std::max<long long>(8, gw / 8 * 8), 256, 0, st>>>(a);
    } else if (h2_on(c)) {
        unsigned* xs = (unsigned*)((float*)ws + c->o_xs);
        // genie_embed_window_split, one-shot; its rows count only if they were written under the station-order state of THIS call
        // (a training forward, or other weights whose range guard flipped the kernels, re-split in their own order)
        const bool presplit = (c->xs_slice == slice && c->xs_mask == mask && c->xs_ws == ws && c->xs_sta_order == (sta_order_on(c) ? 1 : 0)) || !do_split;
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
                k_split_rows_g<<<(unsigned)(c->P_ext / c->S), 256, (size_t)c->S * 32, st>>>(slice, mask, c->S, xs, c->sta_perm, mmw, c->P_ext,
                                                                                            input_limit(c), c->h_inflag);
            } else
                k_split_rows<<<(unsigned)((c->P_ext + 255) / 256), 256, 0, st>>>(slice, mask, c->P_ext, xs,
                                                                                sta_order_on(c) ? c->sta_perm : nullptr, c->S, mmw,
                                                                                input_limit(c), c->h_inflag);
        }
        a.xs = xs; a.packed = c->packed_h2; a.xs_plane = c->P_ext * (long long)XPC;
        a.np = (s2h_on(c) || train_h2u_on(c)) ? 1 : 0;
        c->ws_np = a.np != 0;
        const int grid = da_grid_w(c, (n_tiles + 1) / 2, c->bpc1b, H2_THREADS / 64);
        const bool big = c->P_ext * XROW >= (1ll << 32);
        if (!n_tiles) {
        } else if (c->abs_sta) {
            const int so = sta_order_on(c) ? 1 : 0;      // a training forward runs in the caller's station order, inference in processing order
            if (c->abs_dirty || !c->abs_ts || c->abs_ts_order != so) {
                if (!c->abs_ts) {
                    HIP_TRY(gmalloc((void**)&c->abs_ts, 16 * (size_t)c->S));
                    HIP_TRY(gmalloc((void**)&c->abs_tg, 16 * (size_t)c->G_ext));
                }
                k_abs_pieces<<<(c->S + 255) / 256, 256, 0, st>>>(c->abs_sta, sta_order_on(c) ? c->sta_perm : nullptr, c->S, c->abs_ts);
                k_abs_pieces<<<(c->G_ext + 255) / 256, 256, 0, st>>>(c->abs_src, nullptr, c->G_ext, c->abs_tg);
                c->abs_dirty = false; c->abs_ts_order = so;
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
        HIP_TRY(gfree(dbg_tmp));
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

namespace {
// k_stage2_h2u's tables for the range [gb0, ge0) of the processing order: per XCD chunk (the chunks of ItemIter), blocks of up to
// S2U_NB consecutive source nodes whose neighbour rows (their union, in first-use order) fit S2U_UCAP rows; a block is cut short
// where the union would grow past that. Built once per range and kept with the context.
int get_s2u_tables(genie_ctx* c, int gb0, int ge0, const genie_ctx::S2uTables** out) {
    const auto key = std::make_pair(gb0, ge0);
    auto itf = c->s2u.find(key);
    if (itf != c->s2u.end()) { *out = &itf->second; return GENIE_OK; }
    const std::vector<int32_t>& tab = c->tab_host;
    std::vector<S2uBlock> blks;
    std::vector<int32_t> seen_blk((size_t)c->G_ext, -1), seen_at((size_t)c->G_ext, 0);
    int32_t serial = 0;
    int32_t x0[9];
    const int nxc = 8, n = ge0 - gb0;
    for (int x = 0; x < nxc; ++x) {
        x0[x] = (int32_t)blks.size();
        const int gb = gb0 + (int)((long long)n * x / nxc), ge = gb0 + (int)((long long)n * (x + 1) / nxc);
        int pos = gb;
        while (pos < ge) {
            S2uBlock b;
            memset(&b, 0, sizeof(b));
            b.gi0 = pos;
            // membership of a neighbour row in the block's union by a stamp per source node (`seen_blk[nb]` = serial of the block that
            // listed it, `seen_at[nb]` = its position there): the table of 10 000 source nodes builds in ~0.2 ms, where a linear search
            // of the union per neighbour took 2.5 ms of every context (re)build of a training sample
            ++serial;
            int32_t uni[S2U_UCAP + 15];
            int nuni = 0;
            while (pos < ge && b.n < S2U_NB) {
                int32_t where[15];
                const int before = nuni;
                for (int k = 0; k < 15; ++k) {
                    const int32_t nb = tab[(size_t)pos * 16 + 1 + k];
                    if (seen_blk[(size_t)nb] != serial) { seen_blk[(size_t)nb] = serial; seen_at[(size_t)nb] = nuni; uni[nuni++] = nb; }
                    where[k] = seen_at[(size_t)nb];
                }
                if (b.n > 0 && nuni > S2U_UCAP) {       // the node does not fit: take its additions back, it opens the next block
                    for (int u = before; u < nuni; ++u) seen_blk[(size_t)uni[u]] = -1;
                    nuni = before;
                    break;
                }
                b.idx[b.n][0] = tab[(size_t)pos * 16];
                for (int k = 0; k < 15; ++k) b.idx[b.n][1 + k] = where[k];
                ++b.n; ++pos;
            }
            for (int e = b.n; e < S2U_NB; ++e) b.idx[e][0] = -1;       // empty slots of a short block
            b.U = (int32_t)nuni;
            for (int u = 0; u < 64; ++u) b.ids[u] = uni[u < nuni ? u : 0];
            blks.push_back(b);
        }
    }
    x0[nxc] = (int32_t)blks.size();
    genie_ctx::S2uTables t;
    t.blocks = nullptr; t.xcd0 = nullptr; t.nblk = (int)blks.size();
    HIP_TRY(gmalloc(&t.blocks, sizeof(S2uBlock) * std::max<size_t>(1, blks.size())));
    HIP_TRY(hipMemcpy(t.blocks, blks.data(), sizeof(S2uBlock) * blks.size(), hipMemcpyHostToDevice));
    HIP_TRY(gmalloc((void**)&t.xcd0, sizeof(x0)));
    HIP_TRY(hipMemcpy(t.xcd0, x0, sizeof(x0), hipMemcpyHostToDevice));
    *out = &(c->s2u[key] = t);
    return GENIE_OK;
}
}  // namespace

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
    a.ea_int = nullptr;
    a.mm_int = (const float*)ws + c->o_mm + (c->slot % GENIE_NBIG) * c->big_stride;
    auto identity_order = [&]() -> int {
        if (!c->sta_ident) {
            std::vector<int32_t> id((size_t)c->S);
            for (int i = 0; i < c->S; ++i) id[i] = i;
            HIP_TRY(gmalloc((void**)&c->sta_ident, sizeof(int32_t) * id.size()));
            HIP_TRY(hipMemcpy(c->sta_ident, id.data(), sizeof(int32_t) * id.size(), hipMemcpyHostToDevice));
        }
        return GENIE_OK;
    };
    if (c->force_generic && !c->pcsr && a.save != nullptr && c->use_fast && !abs_generic(c) && no_bip && x_latent_out != nullptr &&
        gi_begin == 0 && gi_end == c->G) {
        // training forward of the association phase (its last pass): the pipelined fp32 stage 2 in the caller's station order with the
        // output layer's pre-activations kept, instead of the generic kernel (427 -> ~300 us at config 3)
        if ((rc = identity_order())) return rc;
        a.sta_user = c->sta_ident; a.wgmap = 0;
        k_stage2_ord<8, 15, true, true, true><<<da_grid(c, n_tiles, c->bpc2o), 256, 0, st>>>(a);
        HIP_TRY(hipGetLastError());
        return GENIE_OK;
    }
    if (c->force_generic && !c->pcsr && a.save != nullptr && c->use_fast && h2_on(c) && !abs_generic(c) && c->src_tab != nullptr && !no_bip) {
        // training forward on the reference's kNN graphs: the production stage 2 in the CALLER's station order (identity
        // processing order: the saved pre-activations of 1.8 GB stay contiguous stores), message mask from the split pass of
        // k_stage1_h2's launch (both model options at once run the generic stage 1, which has no split pass: generic stage 2 below)
        if ((rc = identity_order())) return rc;
        a.sta_user = c->sta_ident; a.ea_int = edge_attr; a.wgmap = 0;
        if (train_h2u_on(c) && c->ws_np && gi_begin == 0 && gi_end == c->G) {
            // k_stage2_h2u with the pre-activations kept (identity station order: edge_attr fragments built per call)
            a.np = 1; a.packed = c->packed_s2h;
            if (!c->ea_frag_tmp) HIP_TRY(gmalloc((void**)&c->ea_frag_tmp, 32 * (size_t)c->P));
            k_ea_frag<<<(unsigned)((c->P + 255) / 256), 256, 0, st>>>(edge_attr, c->P, c->S, nullptr, c->ea_frag_tmp);
            a.ea_frag = c->ea_frag_tmp;
            const genie_ctx::S2uTables* tb = nullptr;
            if ((rc = get_s2u_tables(c, gi_begin, gi_end, &tb))) return rc;
            const size_t lds = sizeof(float) * S2H_IMG_FLOATS + (size_t)S2U_UCAP * 1024;
            const long long items = (long long)tb->nblk * c->T;
            const int grid = (int)std::max<long long>(8, std::min<long long>((long long)c->num_cu * GENIE_S2U_BPC, (items + 7) / 8 * 8) / 8 * 8);
            const bool big = c->P_ext * 128 >= (1ll << 32);
            auto launch = [&](auto kern) -> int {
                if (int r = raise_lds_limit(c, (const void*)kern, 160 * 1024)) return r;
                kern<<<grid, S2U_WPB * 64, lds, st>>>(a, (const S2uBlock*)tb->blocks, tb->xcd0);
                return GENIE_OK;
            };
            if (x_latent_out) rc = big ? launch(k_stage2_h2u<true, true, true>) : launch(k_stage2_h2u<true, false, true>);
            else rc = big ? launch(k_stage2_h2u<false, true, true>) : launch(k_stage2_h2u<false, false, true>);
            if (rc) return rc;
        } else {
            const int grid = da_grid(c, n_tiles, c->bpc2o);
            if (x_latent_out) k_stage2_ord<8, 15, true, false, true><<<grid, 256, 0, st>>>(a);
            else k_stage2_ord<8, 15, false, false, true><<<grid, 256, 0, st>>>(a);
        }
    } else if (c->force_generic && !c->pcsr) {
        k_stage2<<<da_grid(c, n_tiles, c->bpc2), 256, 0, st>>>(a);
    } else if (c->pcsr) {
        const long long ntiles = (c->P + 15) / 16;
        a.ptile = c->ptile16;
        const long long gw = std::min<long long>((ntiles + 3) / 4, (long long)c->num_cu * c->bpc2);      // 2 .. 12 workgroups per CU: +-1 %
        if (!no_bip && !x_latent_out && !a.save && pseg_on()) {
            // inference: a wave per source node, the station sum folded into the pass (k_stage2_pseg), straight into the window's slot
            a.part = (float*)ws + c->o_part + c->slot * c->slot_stride;
            const long long gs = std::min<long long>((c->G + 3) / 4, (long long)c->num_cu * c->bpc2);
            k_stage2_pseg<<<(int)std::max<long long>(8, gs / 8 * 8), 256, 0, st>>>(a, c->seg_rowptr, c->G);
            HIP_TRY(hipGetLastError());
            return GENIE_OK;
        }
        k_stage2_pcsr<<<(int)std::max<long long>(8, gw / 8 * 8), 256, 0, st>>>(a);                       // (a multiple of the 8 XCDs)
        // the gated messages sit in the c rows, which exist GENIE_NBIG times only: their station sums (row order) go to the window's
        // own slot right away, one partial row per source node, so that every tail form (per window, side streams, batched) reads
        // `part` as it does on a Cartesian graph
        if (!no_bip)
            k_seg_sum32<<<(c->G * 32 + 255) / 256, 256, 0, st>>>((const float*)ws + c->o_c + (c->slot % GENIE_NBIG) * c->big_stride, c->seg_rowptr,
                                                                 c->G, (float*)ws + c->o_part + c->slot * c->slot_stride);
    } else if (c->use_fast && a.sta_user != nullptr && (!no_bip || x_latent_out != nullptr)) {
        // the production configuration: uniform 8 / 15-degree graphs, station processing order; the static edge_attr is registered
        // (genie_set_static_edge_attr), any other one is brought into processing order here (one extra pass over [P, 3])
        a.wgmap = no_bip ? 0 : c->s2_wgmap;
        if (no_bip) {      // last pass of the association heads: row-layout c / wu / wv written by k_assoc_b
            k_stage2_ord<8, 15, true, true><<<da_grid(c, n_tiles, c->bpc2o), 256, 0, st>>>(a);       // (3 / 4 / 6 workgroups per CU: 291-298 us against 286)
#if GENIE_TUNING
        } else if (!s2h_on(c)) {     // A/B reference (GENIE_S2_OLD): the round-3 stage 2 on row-layout c / wv
            if (!c->ea_tmp) HIP_TRY(gmalloc((void**)&c->ea_tmp, sizeof(float) * 3 * (size_t)c->P));
            k_permute_sta_rows<<<(unsigned)((c->P * 3 + 255) / 256), 256, 0, st>>>(edge_attr, c->P, 3, c->sta_inv, c->S, c->ea_tmp);
            a.ea_int = c->ea_tmp;
            k_stage2_ord<8, 15, false><<<da_grid(c, n_tiles, c->bpc2o), 256, 0, st>>>(a);
#endif
        } else {           // s2h_on(c): stage 1 of this window wrote c / wv node-planar
            if (!c->ws_np)     // (genie_set_stage_precision between a window's two stages would get here)
                return fail(GENIE_ERR_STATE, "stage 2: the last stage 1 left row-layout c / wv but the f16x2 stage 2 reads them node-planar "
                                             "(stage precision changed between the two stages of a window?)");
            a.np = 1; a.packed = c->packed_s2h;
            if (c->ea_frag && c->ea_user == edge_attr) a.ea_frag = c->ea_frag;
            else {
                if (!c->ea_frag_tmp) HIP_TRY(gmalloc((void**)&c->ea_frag_tmp, 32 * (size_t)c->P));
                k_ea_frag<<<(unsigned)((c->P + 255) / 256), 256, 0, st>>>(edge_attr, c->P, c->S, c->sta_perm, c->ea_frag_tmp);
                a.ea_frag = c->ea_frag_tmp;
            }
            const bool big = c->P_ext * 128 >= (1ll << 32);
            if (!c->tab_host.empty() && !c->s2u_off) {
                // source-neighbour rows of blocks of adjacent source nodes staged once in LDS (k_stage2_h2u), for this range of the order
                const genie_ctx::S2uTables* tb = nullptr;
                if ((rc = get_s2u_tables(c, gi_begin, gi_end, &tb))) return rc;
                const size_t lds = sizeof(float) * S2H_IMG_FLOATS + (size_t)S2U_UCAP * 1024;
                const long long items = (long long)tb->nblk * c->T;
                const long long gsz = std::min<long long>((long long)c->num_cu * GENIE_S2U_BPC, (items + 7) / 8 * 8);
                const int grid = (int)std::max<long long>(8, gsz / 8 * 8);
                auto launch = [&](auto kern) -> int {      // (the 70-KB dynamic LDS needs the attribute once per kernel and device)
                    if (int r = raise_lds_limit(c, (const void*)kern, 160 * 1024)) return r;
                    kern<<<grid, S2U_WPB * 64, lds, st>>>(a, (const S2uBlock*)tb->blocks, tb->xcd0);
                    return GENIE_OK;
                };
                if (x_latent_out) rc = big ? launch(k_stage2_h2u<true, true>) : launch(k_stage2_h2u<true, false>);
                else rc = big ? launch(k_stage2_h2u<false, true>) : launch(k_stage2_h2u<false, false>);
                if (rc) return rc;
            } else {
                const int grid = da_grid(c, n_tiles, c->bpc2h);
                if (x_latent_out) { if (big) k_stage2_h2<true, true><<<grid, 256, 0, st>>>(a); else k_stage2_h2<true, false><<<grid, 256, 0, st>>>(a); }
                else { if (big) k_stage2_h2<false, true><<<grid, 256, 0, st>>>(a); else k_stage2_h2<false, false><<<grid, 256, 0, st>>>(a); }
            }
        }
    }
    else {
        if (c->ws_np && !no_bip)       // (no_bip: the association heads' last pass reads the rows k_assoc_b wrote, whatever stage 1 left)
            return fail(GENIE_ERR_STATE, "stage 2: the last stage 1 left node-planar c / wv but the fp32 stage 2 reads rows (stage precision "
                                         "changed between the two stages of a window?)");
        k_stage2<<<da_grid(c, n_tiles, c->bpc2), 256, 0, st>>>(a);
    }
    HIP_TRY(hipGetLastError());
    return GENIE_OK;
}
}  // namespace

int genie_bipartite_readout(genie_ctx* c, float* bip_out, void* ws, void* stream) {
    int rc = check_ws(c, ws);
    if (rc) return rc;
    if (!bip_out) return fail(GENIE_ERR_ARG, "genie_bipartite_readout: null output");
    const float* part = (const float*)ws + c->o_part + c->slot * c->slot_stride;
    { int rcp = ensure_packed(c, (hipStream_t)stream); if (rcp) return rcp; }
    if (tail_wide(c)) k_bip_out_m<true><<<tl_blocks(c->G, c->num_cu * 2), 256, 0, (hipStream_t)stream>>>(part, c->G, part_T(c), c->packed[PL_BIP], bip_out, 0, 0);
    else k_bip_out_m<false><<<tl_blocks(c->G, c->num_cu * 2), 256, 0, (hipStream_t)stream>>>(part, c->G, part_T(c), c->packed[PL_BIP], bip_out, 0, 0);
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
// virtual blocks of SpatialAggregation's grid-wide mean (SaArgs.vg): a function of the source-node count only, so that the sum's
// partition -- hence every output bit -- is the same for any launch grid (per window, side streams, batches of 1..16 windows)
int sa_vg(const genie_ctx* c) { return tl_blocks(c->G, 512); }
void sa_fill_layer(const genie_ctx* c, int layer, SaArgs& a) {
    const int base = layer == 1 ? W_SA1_FC1_W : (layer == 2 ? W_SA2_FC1_W : W_SA3_FC1_W);
    a.G = c->G; a.C = layer == 1 ? 15 : 30; a.E = c->E_src; a.vg = sa_vg(c);
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
    a.pj_in = pj[cur]; a.gpart_in = gp[cur]; a.n_gpart_in = sa_vg(c);
    a.pj_out = pj[cur ^ 1]; a.gpart_out = gp[cur ^ 1];
    const int nb = sa_blocks(c);
    { int rcp = ensure_packed(c, st); if (rcp) return rcp; }
    a.img = c->packed[PL_SA1 + layer - 1];
    if (tail_wide(c)) {
        if (layer == 1) { if (with_next) k_sa_layer_m<15, true, true><<<nb, 256, 0, st>>>(a); else k_sa_layer_m<15, false, true><<<nb, 256, 0, st>>>(a); }
        else { if (with_next) k_sa_layer_m<30, true, true><<<nb, 256, 0, st>>>(a); else k_sa_layer_m<30, false, true><<<nb, 256, 0, st>>>(a); }
    } else if (layer == 1) {
        if (with_next) k_sa_layer_m<15, true, false><<<nb, 256, 0, st>>>(a); else k_sa_layer_m<15, false, false><<<nb, 256, 0, st>>>(a);
    } else {
        if (with_next) k_sa_layer_m<30, true, false><<<nb, 256, 0, st>>>(a); else k_sa_layer_m<30, false, false><<<nb, 256, 0, st>>>(a);
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
    if (tail_wide(c)) { if (layer == 1) k_sa_pre_m<15, true><<<nb, 256, 0, st>>>(a); else k_sa_pre_m<30, true><<<nb, 256, 0, st>>>(a); }
    else { if (layer == 1) k_sa_pre_m<15, false><<<nb, 256, 0, st>>>(a); else k_sa_pre_m<30, false><<<nb, 256, 0, st>>>(a); }
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
    if (c->G_ext != c->G) {
        if ((rc = genie_da_stage2_bipartite(c, mask, edge_attr, x_latent_out, bip, ws, stream))) return rc;
        return genie_spatial_agg3_fwd(c, bip, pos, x_spatial_out, ws, stream);
    }
    // the Bipartite read-out and the pre-pass of SpatialAggregation1 in ONE launch (k_bip_pre_m, the batched tail's first kernel:
    // the same MFMA chains, bitwise equal results), then the three layers: one launch less on the literal call's critical path
    if ((rc = genie_da_stage2_partials(c, mask, edge_attr, x_latent_out, ws, stream))) return rc;
    hipStream_t st = (hipStream_t)stream;
    if ((rc = ensure_packed(c, st))) return rc;
    const size_t so = c->slot * c->slot_stride;
    {
        SaArgs a;
        memset(&a, 0, sizeof(a));
        sa_fill_layer(c, 1, a);
        a.out = bip;
        a.pj_out = w + c->o_pj0 + so; a.gpart_out = w + c->o_gpart + so;
        a.img = c->packed[PL_SA1];
        const int nb = sa_blocks(c);
        if (tail_wide(c)) k_bip_pre_m<true><<<nb, 256, 0, st>>>(w + c->o_part + so, part_T(c), c->packed[PL_BIP], 0, a);
        else k_bip_pre_m<false><<<nb, 256, 0, st>>>(w + c->o_part + so, part_T(c), c->packed[PL_BIP], 0, a);
        HIP_TRY(hipGetLastError());
    }
    if ((rc = sa_launch_layer(c, 1, bip, pos, w + c->o_sa0 + so, w, 0, true, st))) return rc;
    if ((rc = sa_launch_layer(c, 2, w + c->o_sa0 + so, pos, w + c->o_sa1 + so, w, 1, true, st))) return rc;
    return sa_launch_layer(c, 3, w + c->o_sa1 + so, pos, x_spatial_out, w, 0, false, st);
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

namespace {
// grid read-out (k_readout_m<0>): fp64 chains for inference calls (their score scratch holds doubles), fp32 for the training forward
int launch_readout_grid(genie_ctx* c, const RoArgs& a, bool with_cv, hipStream_t st) {
    const bool wide = tail_wide(c);
    const size_t lds = sizeof(float) * (ROM_LDS_FLOATS + (wide ? 4 * 16 * RO_SCS : 0) + (with_cv ? GP_IMG_FLOATS : 0));
    const int lds_max = (int)(sizeof(float) * (ROM_LDS_FLOATS + 4 * 16 * RO_SCS + GP_IMG_FLOATS));
    if (wide) {
        HIP_TRY(hipFuncSetAttribute((const void*)k_readout_m<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
        k_readout_m<0, true><<<tl_blocks(a.N, c->tail_cu_ro), 256, lds, st>>>(a);
    } else {
        HIP_TRY(hipFuncSetAttribute((const void*)k_readout_m<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
        k_readout_m<0, false><<<tl_blocks(a.N, c->tail_cu_ro), 256, lds, st>>>(a);
    }
    return GENIE_OK;
}
}  // namespace

int genie_readout_grid(genie_ctx* c, const float* x_spatial, const float* t_query, int n_t, float* y_out, void* stream) {
    if (!c || !x_spatial || !t_query || !y_out) return fail(GENIE_ERR_ARG, "genie_readout_grid: null argument");
    if (n_t < 1 || n_t > RO_TMAX) return fail(GENIE_ERR_ARG, "genie_readout_grid: 1 <= n_t <= 10 required");
    RoArgs a = make_ro_args(c);
    { int rcp = ensure_packed(c, (hipStream_t)stream); if (rcp) return rcp; }
    a.N = a.Nw = c->G; a.T = n_t; a.x_spatial = x_spatial; a.t_query = t_query; a.out = y_out;
    a.img = c->packed[PL_RO0];
    { int rl = launch_readout_grid(c, a, false, (hipStream_t)stream); if (rl) return rl; }
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
    if (tail_wide(c)) k_ro_pre_m<true><<<tl_blocks(c->G, c->tail_cu_ro), 256, 0, (hipStream_t)stream>>>(x_spatial, c->G, c->packed[PL_ROP], cvbuf, c->G, 0);
    else k_ro_pre_m<false><<<tl_blocks(c->G, c->tail_cu_ro), 256, 0, (hipStream_t)stream>>>(x_spatial, c->G, c->packed[PL_ROP], cvbuf, c->G, 0);
    a.img = c->packed[PL_RO1];
    if (n_query <= 16384) {      // about one 16-query tile per wave of the launch: nothing else hides a wave's round trips (PF)
        HIP_TRY(hipFuncSetAttribute((const void*)k_readout_m<1, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(float) * ROM_LDS_FLOATS)));
        k_readout_m<1, false, true><<<tl_blocks(a.N, c->tail_cu_ro), 256, sizeof(float) * ROM_LDS_FLOATS, (hipStream_t)stream>>>(a);
    } else {
        HIP_TRY(hipFuncSetAttribute((const void*)k_readout_m<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(float) * ROM_LDS_FLOATS)));
        k_readout_m<1><<<tl_blocks(a.N, c->tail_cu_ro), 256, sizeof(float) * ROM_LDS_FLOATS, (hipStream_t)stream>>>(a);
    }
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
    { int rl = launch_readout_grid(c, a, false, (hipStream_t)stream); if (rl) return rl; }
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
        if (tail_wide(c)) k_bip_pre_m<true><<<grid, 256, 0, st>>>(w + c->o_part + so, part_T(c), c->packed[PL_BIP], ss, a);
        else k_bip_pre_m<false><<<grid, 256, 0, st>>>(w + c->o_part + so, part_T(c), c->packed[PL_BIP], ss, a);
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
        a.pj_in = pj[cur]; a.gpart_in = gp[cur]; a.n_gpart_in = sa_vg(c);
        a.pj_out = pj[cur ^ 1]; a.gpart_out = gp[cur ^ 1];
        a.img = c->packed[PL_SA1 + layer - 1];
        if (tail_wide(c)) {
            if (layer == 1) k_sa_layer_m<15, true, true><<<grid, 256, 0, st>>>(a);
            else if (layer == 2) k_sa_layer_m<30, true, true><<<grid, 256, 0, st>>>(a);
            else k_sa_layer_m<30, false, true><<<grid, 256, 0, st>>>(a);
        } else {
            if (layer == 1) k_sa_layer_m<15, true, false><<<grid, 256, 0, st>>>(a);
            else if (layer == 2) k_sa_layer_m<30, true, false><<<grid, 256, 0, st>>>(a);
            else k_sa_layer_m<30, false, false><<<grid, 256, 0, st>>>(a);
        }
    }
    // read-out heads over the nwin * G grid nodes / nwin * Q queries
    RoArgs a = make_ro_args(c);
    a.T = n_t; a.x_spatial = x_spatial_out; a.t_query = t_query;
    {   // the grid read-out also leaves the per-grid-node table cv of the query read-out (k_ro_pre_m's work)
        RoArgs g = a;
        g.N = nwin * c->G; g.Nw = c->G; g.out = y_out; g.img = c->packed[PL_RO0];
        if (x_out) { g.cv_out = w + c->o_cv + so; g.cv_ws = ss; g.pimg = c->packed[PL_ROP]; }
        { int rl = launch_readout_grid(c, g, x_out != nullptr, st); if (rl) return rl; }
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
    if ((rc = ensure_packed(c, (hipStream_t)stream))) return rc;       // (the range guard of the committed weights decides h2_on)
    unsigned* xs = (h2_on(c) && !c->pcsr) ? (unsigned*)((float*)ws + c->o_xs) : nullptr;     // (stage 1 of an irregular graph splits its rows itself)
    rc = embed_window_impl(c, pick_t, pick_sta, pick_phase, n_picks, t0, max_t, kernel_sig_t, dt, trv, emb_ws, slice_out, mask_out, xs,
                           stream);
    if (rc == GENIE_OK && xs) {
        c->xs_slice = slice_out; c->xs_mask = mask_out; c->xs_ws = ws; c->xs_mm_copy = c->slot % GENIE_NBIG;
        c->xs_sta_order = sta_order_on(c) ? 1 : 0;
    }
    return rc;
}

namespace {
int embed_window_impl(genie_ctx* c, const double* pick_t, const int32_t* pick_sta, const int32_t* pick_phase, int n_picks,
                      double t0, double max_t, double kernel_sig_t, double dt, const float* trv, float* emb_ws,
                      float* slice_out, float* mask_out, unsigned* xs, void* stream) {
    if (!c || !trv || !emb_ws || !slice_out || !mask_out) return fail(GENIE_ERR_ARG, "genie_embed_window: null argument");
    if (n_picks > 0 && (!pick_t || !pick_sta || !pick_phase)) return fail(GENIE_ERR_ARG, "genie_embed_window: null pick array");
    if (!(dt > 0.0) || !(kernel_sig_t > 0.0) || !(max_t > 0.0)) return fail(GENIE_ERR_ARG, "genie_embed_window: bad dt / sigma / max_t");
    if (c->pcsr && !c->p_sta_of)
        return fail(GENIE_ERR_STATE, "genie_embed_window: an irregular product graph needs the station of every product node (genie_set_subgraph_stations)");
    hipStream_t st = (hipStream_t)stream;
    EmbArgs a;
    memset(&a, 0, sizeof(a));
    a.sta_of = c->pcsr ? c->p_sta_of : nullptr;
    a.pick_t = pick_t; a.pick_sta = pick_sta; a.pick_phase = pick_phase; a.n_picks = n_picks;
    a.S = c->S; a.t0 = t0; a.tref0 = t0 - 3.0 * kernel_sig_t; a.dt = dt; a.sigma = kernel_sig_t;
    a.n_time = genie_embed_ntime(t0, max_t, kernel_sig_t, dt);
    a.n_extra = (int)ceil(3.0 * kernel_sig_t / dt);                                       // process_utils.py:518
    a.emb = emb_ws; a.trv = trv; a.rows = c->P_ext; a.slice = slice_out; a.mask = mask_out; a.xs = xs;
    a.no_phase = c->no_phase; a.sign_input = c->sign_input;
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
int build_reversed(const int32_t* d_rowptr, const int32_t* d_col, int n_tgt, int n_src, int32_t** r_rowptr, int32_t** r_col, float** r_w,
                   int2** r_cw = nullptr) {
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
    HIP_TRY(gmalloc((void**)r_rowptr, sizeof(int32_t) * rrp.size()));
    HIP_TRY(hipMemcpy(*r_rowptr, rrp.data(), sizeof(int32_t) * rrp.size(), hipMemcpyHostToDevice));
    HIP_TRY(gmalloc((void**)r_col, sizeof(int32_t) * std::max<size_t>(E, 1)));
    HIP_TRY(gmalloc((void**)r_w, sizeof(float) * std::max<size_t>(E, 1)));
    if (E) {
        HIP_TRY(hipMemcpy(*r_col, rcol.data(), sizeof(int32_t) * E, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(*r_w, rw.data(), sizeof(float) * E, hipMemcpyHostToDevice));
    }
    if (r_cw) {
        std::vector<int2> cw(std::max<size_t>(E, 1), int2{0, 0});
        for (size_t e = 0; e < E; ++e) { cw[e].x = rcol[e]; memcpy(&cw[e].y, &rw[e], 4); }
        HIP_TRY(gmalloc((void**)r_cw, sizeof(int2) * cw.size()));
        HIP_TRY(hipMemcpy(*r_cw, cw.data(), sizeof(int2) * cw.size(), hipMemcpyHostToDevice));
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
// Waves per workgroup of the lightest backward pass on a Cartesian graph (k_train_b2: 60 registers, 6 accumulator tiles). It streams
// its rows once and is bound by the latency of those loads; at the 4 waves of the heavy passes a CU held 8 such waves. 8 per
// workgroup: 289 -> 202 us (16: the same, and four times the partials for the reduction to read); tools/train_ab.sh. The same for the
// association phase's light passes did not pay: k_as_b3 174 -> 167 us, k_as_b0 (88 registers, 12 tiles) 301 -> 334 us: they keep 4.
// Every wave still writes its own partial slot; the scratch is sized by the heaviest pass (30 tiles x 4 waves).
constexpr int TR_WPB_LIGHT = 8;
// pass 1' as two waves per tile (k_train_b1s); the tuning builds can switch back to the one-wave kernel for A/B runs
#ifndef GENIE_B1_SPLIT_DEFAULT
#define GENIE_B1_SPLIT_DEFAULT 1          // (a build with 0 = the one-wave kernel, for same-box A/B runs of production code generation)
#endif
bool b1_split_on() {
    static const char* e = tune_env("GENIE_B1_SPLIT");
    return e ? atoi(e) != 0 : GENIE_B1_SPLIT_DEFAULT != 0;
}
int train_grid(const genie_ctx* c) {
#if GENIE_TUNING
    { static const char* e = getenv("GENIE_TRAIN_WG"); if (e) return std::max(8, c->num_cu * atoi(e) / 8 * 8); }
#endif
    return std::max(8, c->num_cu * 2 / 8 * 8);
}      // 2 workgroups per CU (1: +10 %, 3: +6 %, 4: +1 % step time)
size_t train_part_floats(const genie_ctx* c) { return (size_t)train_grid(c) * 4 * (30 * 256 + 12 * 16 + 16); }
// `variants`: the call also serves DataAggregationEdges / use_absolute_pos (the forward_fixed_source step does; the association heads'
// training step is the default model definition only)
int train_check(const genie_ctx* c, const char* who, bool variants = false, bool pcsr_ok = false) {
    if ((c->pcsr && !pcsr_ok) || c->G_ext != c->G) return fail(GENIE_ERR_STATE, std::string(who) + ": needs an unsharded Cartesian product graph");
    if (!variants && (c->has_edges || c->abs_sta)) return fail(GENIE_ERR_STATE, std::string(who) + ": default model definition only");
    return GENIE_OK;
}
// station sums live as [G][T][32] partial rows; on an irregular product graph the training calls keep ONE row per source node there
int ensure_src_of(genie_ctx* c, hipStream_t st) {
    if (c->p_src_of || !c->pcsr) return GENIE_OK;
    HIP_TRY(gmalloc((void**)&c->p_src_of, sizeof(int32_t) * (size_t)c->P));
    k_seg_owner<<<(c->G + 255) / 256, 256, 0, st>>>(c->seg_rowptr, c->G, c->p_src_of);
    HIP_TRY(hipGetLastError());
    return GENIE_OK;
}
// scratch of static_term_grads: per-source-node sums [G][16], per-station partial sums [SG_CHUNKS][S][16], slices of one dW block
size_t static_scratch_floats(const genie_ctx* c) { return (size_t)c->G * 16 + (size_t)(SG_CHUNKS + 1) * c->S * 16 + (size_t)SG_SLICES * 64; }

// Weight gradients of the static terms of DataAggregationEdges (l?_t?_2.weight_pos) and use_absolute_pos (init_trns.weight_abs) from
// the gradient rows the three passes left in `gr` (train_front_kernels.hpp, k_gr_sum_* / k_static_dw*).
// `which`: 1 = the layer-2 terms (from do), 2 = the layer-1 terms (dt), 4 = init_trns (dz0 / the association phase's dtrp); `assoc`:
// the association phase's parameters (its passes overwrite do with dtrp, so its do terms are taken between two passes).
int static_term_grads(genie_ctx* c, const float* gr, float* scr, float* grad_blob, hipStream_t st, int which = 7, bool assoc = false) {
    struct Term { int blk; bool sta; const float* f; int nf, w, ld, row0, rows, col0; };
    std::vector<Term> terms;
    const int w_l1[2] = {assoc ? W_AS_L1T12_P : W_DA_L1T12_P, assoc ? W_AS_L1T22_P : W_DA_L1T22_P};
    const int w_l2[2] = {assoc ? W_AS_L2T12_P : W_DA_L2T12_P, assoc ? W_AS_L2T22_P : W_DA_L2T22_P};
    const int w_abs = assoc ? W_AS_INIT_ABS : W_DA_INIT_ABS, blk_init = assoc ? GR_DTRP : GR_DH0;
    if (c->has_edges && (which & 2)) {
        for (int b = 0; b < 2; ++b) terms.push_back({GR_DT + b, true, c->mpos_sta, 4, w_l1[0], 4, 16 * b, std::min(16, 30 - 16 * b), 0});
        for (int b = 0; b < 2; ++b) terms.push_back({GR_DT + 2 + b, false, c->mpos_src, 4, w_l1[1], 4, 16 * b, std::min(16, 30 - 16 * b), 0});
    }
    if (c->has_edges && (which & 1)) {
        terms.push_back({GR_DO + 0, true, c->mpos_sta, 4, w_l2[0], 4, 0, 15, 0});
        terms.push_back({GR_DO + 1, false, c->mpos_src, 4, w_l2[1], 4, 0, 15, 0});
    }
    if (c->abs_sta && (which & 4)) {
        for (int b = 0; b < 2; ++b) terms.push_back({blk_init + b, true, c->abs_sta, 3, w_abs, 6, 16 * b, std::min(16, 30 - 16 * b), 0});
        for (int b = 0; b < 2; ++b) terms.push_back({blk_init + b, false, c->abs_src, 3, w_abs, 6, 16 * b, std::min(16, 30 - 16 * b), 3});
    }
    float* r_src = scr; float* p_sta = r_src + (size_t)c->G * 16; float* r_sta = p_sta + (size_t)SG_CHUNKS * c->S * 16;
    float* dpart = r_sta + (size_t)c->S * 16;
    for (const Term& t : terms) {
        const float* blk = gr + (size_t)t.blk * 16 * (size_t)c->P;
        if (c->pcsr) {      // irregular product graph: the tables are per product node, the gradient rows contract with them as they are
            k_static_dw<<<SG_SLICES, 256, 0, st>>>(blk, t.f, (int)c->P, dpart);
        } else if (t.sta) {
            k_gr_sum_sta<<<dim3((c->S * 4 + 255) / 256, SG_CHUNKS), 256, 0, st>>>(blk, c->S, c->G, p_sta);
            k_gr_sum_parts<<<(c->S * 16 + 63) / 64, 64, 0, st>>>(p_sta, SG_CHUNKS, c->S * 16, r_sta);
            k_static_dw<<<SG_SLICES, 256, 0, st>>>(r_sta, t.f, c->S, dpart);
        } else {
            k_gr_sum_src<<<(c->G + 3) / 4, 256, 0, st>>>(blk, c->S, c->G, r_src);
            k_static_dw<<<SG_SLICES, 256, 0, st>>>(r_src, t.f, c->G, dpart);
        }
        k_static_dw_sum<<<1, 64, 0, st>>>(dpart, t.rows, t.nf, grad_blob + g_params[t.w].off, t.ld, t.row0, t.col0);
    }
    HIP_TRY(hipGetLastError());
    return GENIE_OK;
}
}  // namespace

size_t genie_train_save_floats(const genie_ctx* c) { return c ? (size_t)SV_BLOCKS * 16 * (size_t)c->P : 0; }
size_t genie_train_scratch_floats(const genie_ctx* c) {
    return c ? (size_t)GR_BLOCKS * 16 * (size_t)c->P + train_part_floats(c) + static_scratch_floats(c) : 0;
}

int genie_da_train_fwd(genie_ctx* c, const float* slice, const float* mask, const float* edge_attr, float* save,
                       float* x_latent_out, float* r_out, void* ws, void* stream) {
    int rc = check_ws(c, ws);
    if (rc) return rc;
    if (!slice || !mask || !edge_attr || !save || !r_out) return fail(GENIE_ERR_ARG, "genie_da_train_fwd: null argument");
    if ((rc = train_check(c, "genie_da_train_fwd", true, true))) return rc;
    c->force_generic = 1; c->train_save = save;
    rc = run_stage1(c, slice, mask, nullptr, nullptr, ws, stream, 0, c->G, true);
    if (!rc) rc = run_stage2(c, mask, edge_attr, x_latent_out, ws, stream, 0, c->G);
    c->force_generic = 0; c->train_save = nullptr;
    if (rc) return rc;
    float* part = (float*)ws + c->o_part + c->slot * c->slot_stride;
    k_part_sum<<<(c->G * 30 + 255) / 256, 256, 0, (hipStream_t)stream>>>(part, c->G, part_T(c), r_out);
    HIP_TRY(hipGetLastError());
    return GENIE_OK;
}

namespace {
int ensure_reversed(genie_ctx* c) {
    if (c->r_sta_rowptr && c->r_sta_cw) return GENIE_OK;
    int rc;
    if (c->r_sta_rowptr) {      // built by genie_nbr_mean_bwd without the pair arrays: rebuild whole
        void* old[] = {c->r_sta_rowptr, c->r_sta_col, c->r_sta_w, c->r_src_rowptr, c->r_src_col, c->r_src_w};
        (void)hipDeviceSynchronize();
        for (void* q : old) (void)gfree(q);
        c->r_sta_rowptr = c->r_sta_col = c->r_src_rowptr = c->r_src_col = nullptr; c->r_sta_w = c->r_src_w = nullptr;
    }
    if (c->pcsr) {      // irregular product graph: the P-sized passes gather by product-node id over the reversed PRODUCT-level graphs
        const int np = (int)c->P;    // (the G-sized tail keeps the reversed base source graph below)
        int32_t* col_unused = nullptr;
        float* w_unused = nullptr;
        if ((rc = build_reversed(c->p_sta_rowptr, c->p_sta_col, np, np, &c->rp_sta_rowptr, &col_unused, &w_unused, &c->rp_sta_cw))) return rc;
        (void)gfree(col_unused); (void)gfree(w_unused);
        col_unused = nullptr; w_unused = nullptr;
        if ((rc = build_reversed(c->p_src_rowptr, c->p_src_col, np, np, &c->rp_src_rowptr, &col_unused, &w_unused, &c->rp_src_cw))) return rc;
        (void)gfree(col_unused); (void)gfree(w_unused);
        // base station graph: not part of a subgraph context; an empty reversed graph keeps the non-null contract of the callers
        std::vector<int32_t> zero((size_t)c->S + 1, 0);
        HIP_TRY(gmalloc((void**)&c->r_sta_rowptr, sizeof(int32_t) * zero.size()));
        HIP_TRY(hipMemcpy(c->r_sta_rowptr, zero.data(), sizeof(int32_t) * zero.size(), hipMemcpyHostToDevice));
        HIP_TRY(gmalloc((void**)&c->r_sta_col, sizeof(int32_t)));
        HIP_TRY(gmalloc((void**)&c->r_sta_w, sizeof(float)));
        HIP_TRY(gmalloc((void**)&c->r_sta_cw, sizeof(int2)));
        return build_reversed(c->src_rowptr, c->src_col, c->G, c->G, &c->r_src_rowptr, &c->r_src_col, &c->r_src_w, &c->r_src_cw);
    }
    if ((rc = build_reversed(c->sta_rowptr, c->sta_col, c->S, c->S, &c->r_sta_rowptr, &c->r_sta_col, &c->r_sta_w, &c->r_sta_cw))) return rc;
    return build_reversed(c->src_rowptr, c->src_col, c->G, c->G, &c->r_src_rowptr, &c->r_src_col, &c->r_src_w, &c->r_src_cw);
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
    if ((rc = train_check(c, "genie_da_train_bwd", true, true))) return rc;
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
    a.r_sta_cw = c->r_sta_cw; a.r_src_cw = c->r_src_cw;
    a.slice = slice; a.mask = mask; a.edge_attr = edge_attr; a.save = save; a.dr = d_r;
    a.gr = scratch; a.part = scratch + (size_t)GR_BLOCKS * 16 * (size_t)c->P;
    a.sv_t = SV_T; a.sv_up = SV_UP; a.sv_vp = SV_VP;
    a.store_dz0 = c->abs_sta != nullptr;
    if (c->pcsr) {
        if ((rc = ensure_src_of(c, st))) return rc;
        a.src_of = c->p_src_of; a.ptile = c->ptile16;
        a.r_sta_rowptr = c->rp_sta_rowptr; a.r_sta_cw = c->rp_sta_cw; a.r_src_rowptr = c->rp_src_rowptr; a.r_src_cw = c->rp_src_cw;
        a.r_sta_col = a.r_src_col = nullptr; a.r_sta_w = a.r_src_w = nullptr;       // (the PCSR passes read the pair arrays only)
    }
#if GENIE_TUNING
    { static const char* e = getenv("GENIE_TRABL"); a.abl = e ? atoi(e) : 0; }
#endif
    const int grid = train_grid(c);
    // 32-bit row offsets on scalar bases (ldo / sto) while every block of the kept rows lies below 4 GiB
    const bool o32 = !c->pcsr && (unsigned long long)SV_BLOCKS * (unsigned long long)c->P * 64ull < (1ull << 32);
    for (int s = 0; s < 3; ++s) {
        a.packed = c->packed[4 + s]; a.n_acc = c->n_acc[s]; a.n_vec = c->n_vec[s];
        // k_train_b1 holds one wave per SIMD (442 registers): one workgroup per CU is all that is ever resident. Its pair-split form
        // k_train_b1s (two waves per tile, <= 256 registers) runs two workgroups per CU and writes one partial per PAIR
        const bool split1 = s == 1 && o32 && b1_split_on();
        const int grid_s = (s == 1 && !split1) ? std::max(8, grid / 2 / 8 * 8) : grid;
        const int grid_w = (c->pcsr && s == 1) ? grid : grid_s;        // workgroups of this pass (= 4 waves of partials each)
        if (c->pcsr) {
            if (s == 0) k_train_b2<true><<<grid, 256, 0, st>>>(a);
            else if (s == 1) k_train_b1p<false><<<grid, 256, 0, st>>>(a);
            else k_train_b0<true><<<grid, 256, 0, st>>>(a);
        } else {
            if (s == 0) k_train_b2<false, TR_WPB_LIGHT><<<grid, TR_WPB_LIGHT * 64, 0, st>>>(a);
            else if (s == 1) {
                if (split1) k_train_b1s<false><<<grid_s, 256, 0, st>>>(a);
                else if (o32) k_train_b1<false, true><<<grid_s, 256, 0, st>>>(a); else k_train_b1<false><<<grid_s, 256, 0, st>>>(a);
            }
            else if (o32) k_train_b0<false, true><<<grid, 256, 0, st>>>(a);
            else k_train_b0<false><<<grid, 256, 0, st>>>(a);
        }
        const int stride = a.n_acc * 256 + a.n_vec * 16 + 16;
        const int wpb_s = (s == 0 && !c->pcsr) ? TR_WPB_LIGHT : (split1 ? 2 : 4);
        k_train_reduce<<<(stride + 31) / 32, 256, 0, st>>>(a.part, grid_w * wpb_s, a.n_acc, a.n_vec, c->n_sc[s], c->d_acc[s], c->d_vec[s],
                                                            c->d_sc[s], grad_blob, 0);
    }
    if (c->has_edges || c->abs_sta) {
        if ((rc = static_term_grads(c, a.gr, a.part + train_part_floats(c), grad_blob, st))) return rc;
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
    size_t dxs_a, dxs_b, eb, dxm, cv, pj, gpart, dxd, dan, dx0, dx1, part_ro, part_ro0, part_a, part_b, total;
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
    t.part_ro0 = take(tt_part_floats(RB_NACC1, RB_NVEC, tt_grid(G)));      // the grid branch's own partials: it runs beside the query branch
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
    if ((rc = train_check(c, "genie_tail_train_fwd", true, true))) return rc;
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
    k_part_sum32<<<(c->G * 32 + 255) / 256, 256, 0, st>>>(w + c->o_part + so, c->G, part_T(c), r);
    k_bip_out_m<false><<<tl_blocks(c->G, c->tail_cu_sa), 256, 0, st>>>(w + c->o_part + so, c->G, part_T(c), c->packed[PL_BIP], bip, 0, 0);
    if ((rc = raise_lds_limit(c, (const void*)k_readout_m<1>, (int)(sizeof(float) * ROM_LDS_FLOATS)))) return rc;   // (before tail_train is set: no
                                                                                                                     // early return may leave it on)
    c->tail_train = 1;           // fp32 chains: the backward recomputes every pre-activation of the tail with them
    rc = sa_launch_pre(c, 1, bip, w, 0, st);
    if (!rc) rc = sa_launch_layer(c, 1, bip, pos, sa1, w, 0, true, st);
    if (!rc) rc = sa_launch_layer(c, 2, sa1, pos, sa2, w, 1, true, st);
    if (!rc) rc = sa_launch_layer(c, 3, sa2, pos, xs, w, 0, false, st);
    RoArgs a = make_ro_args(c);
    a.T = n_t; a.x_spatial = xs; a.t_query = t_query;
    if (!rc) {
        RoArgs g = a;
        g.N = g.Nw = c->G; g.out = y_out; g.lat_out = y_latent_out; g.img = c->packed[PL_RO0];
        rc = launch_readout_grid(c, g, false, st);
    }
    c->tail_train = 0;
    if (rc) return rc;
    float* cvbuf = w + c->o_cv + so;
    k_ro_pre_m<false><<<tl_blocks(c->G, c->tail_cu_ro), 256, 0, st>>>(xs, c->G, c->packed[PL_ROP], cvbuf, c->G, 0);
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
    if ((rc = train_check(c, "genie_tail_train_bwd", true, true))) return rc;
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
    // The y branch (grid read-out: 61 us) and the x branch (query read-out + its grid-node side: 159 + 37 us) are independent until
    // SpatialAggregation3 and each is ONE wave tile per wave on a fraction of the CUs: the y branch runs on the context's side stream
    // beside the x branch and is joined (and reduced: both branches add into TemporalAttention's gradients) before the layers.
    if (!c->side_stream) {
        // ONE side stream per device for every context (a stream is a hardware queue: creating one per context cost the first
        // backward of every rebuilt context ~10 ms -- the reference's training loop builds a context per sample); events per context
        static std::map<int, hipStream_t> side;
        static std::mutex mu;
        std::lock_guard<std::mutex> lk(mu);
        hipStream_t& s = side[c->device];
        if (!s) HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
        c->side_stream = s;
    }
    const int grid_y = tt_grid(c->G);
    HIP_TRY(hipEventRecord(c->ev_fork, st));
    HIP_TRY(hipStreamWaitEvent(c->side_stream, c->ev_fork, 0));
    {   // y branch: TemporalAttention + SpatialDirect
        RbArgs b;
        memset(&b, 0, sizeof(b));
        b.ro = ro; b.ro.N = b.ro.Nw = c->G; b.ro.img = c->packed[PL_RO0];
        b.timg = c->packed[PL_TRO0]; b.d_out = d_y; b.d_lat = d_ylat_extra; b.dxs = S + L.dxs_a;
        b.part = S + L.part_ro0; b.n_acc = c->n_acc[TM_RO0]; b.n_vec = c->n_vec[TM_RO0];
        k_ro_bwd<0><<<grid_y, 256, sizeof(float) * RB_LDS_FLOATS, c->side_stream>>>(b);
        HIP_TRY(hipEventRecord(c->ev_join, c->side_stream));
    }
    {   // x branch: TemporalAttention + SpatialAttention (query side), then its grid-node side
        k_ro_pre_m<false><<<tl_blocks(c->G, c->tail_cu_ro), 256, 0, st>>>(xs, c->G, c->packed[PL_ROP], S + L.cv, c->G, 0);
        RbArgs b;
        memset(&b, 0, sizeof(b));
        b.ro = ro; b.ro.N = b.ro.Nw = n_query; b.ro.x_grid = pos; b.ro.x_query = x_query; b.ro.knn = knn; b.ro.cv = S + L.cv;
        b.ro.img = c->packed[PL_RO1];
        b.timg = c->packed[PL_TRO1]; b.d_out = d_x; b.d_lat = d_qlat_extra; b.eb = S + L.eb; b.dxm = S + L.dxm;
        b.part = S + L.part_ro; b.n_acc = c->n_acc[TM_RO1]; b.n_vec = c->n_vec[TM_RO1];
        const int grid = tt_grid(n_query);
        k_ro_bwd<1><<<grid, 256, sizeof(float) * RB_LDS_FLOATS, st>>>(b);
        tt_reduce(c, TM_RO1, b.part, grid * 4, grad_blob, st);
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
    HIP_TRY(hipStreamWaitEvent(st, c->ev_join, 0));          // join: the y branch's d x_spatial (dxs_a) and partials are complete
    tt_reduce(c, TM_RO0, S + L.part_ro0, grid_y * 4, grad_blob, st);
    k_tq_bwd<<<1, 256, 0, st>>>(c->raw, grad_blob + g_raw_total, t_query, n_t, c->scale_t, g_params[W_TA_Q1_W].off, g_params[W_TA_Q1_B].off,
                                g_params[W_TA_Q2_W].off, g_params[W_TA_Q2_B].off, g_params[W_TA_ACT3].off, grad_blob);
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
        if (layer == 1) k_sa_pre_m<15, false><<<nbp, 256, 0, st>>>(pre); else k_sa_pre_m<30, false><<<nbp, 256, 0, st>>>(pre);
        SbArgs a;
        memset(&a, 0, sizeof(a));
        a.G = c->G; a.C = layer == 1 ? 15 : 30; a.E = c->E_src; a.x_in = x_in[layer - 1]; a.pos = pos;
        a.rowptr = c->src_rowptr; a.col = c->src_col; a.outdeg = c->outdeg; a.r_rowptr = c->r_src_rowptr; a.r_col = c->r_src_col;
        a.raw = c->raw; a.fc1_w = g_params[(layer == 1 ? W_SA1_FC1_W : (layer == 2 ? W_SA2_FC1_W : W_SA3_FC1_W))].off;
        a.scale_rel = c->scale_rel; a.pj = S + L.pj; a.gpart = S + L.gpart; a.n_gpart = sa_vg(c);
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
    // irregular product graph: the edge-feature terms of BOTH sides are per product node (ps rows = product nodes), none in pg
    const float* mpos_src = (c->has_edges && !c->pcsr) ? c->mpos_src : nullptr;
    const float* mpos_sta = c->has_edges ? c->mpos_sta : nullptr;
    k_assoc_pre<<<(c->G * AS_PG + 255) / 256, 256, 0, st>>>(c->raw, o, y_latent, mask_src, c->G, mpos_src, c->pcsr ? nullptr : c->abs_src, c->as_pg);
    if (c->as_ps) {
        const long long rows = edge_rows_sta(c);
        k_assoc_ps<<<(unsigned)((rows * AS_PS + 255) / 256), 256, 0, st>>>(c->raw, o, rows, mpos_sta, c->abs_sta,
                                                                          (c->has_edges && c->pcsr) ? c->mpos_src : nullptr,
                                                                          c->pcsr ? c->abs_src : nullptr, c->as_ps);
    }
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
    if (c->G_ext != c->G) return fail(GENIE_ERR_STATE, "genie_assoc_fwd: needs an unsharded product graph");
    if (((uintptr_t)assoc_ws & 15) != 0) return fail(GENIE_ERR_ARG, "genie_assoc_fwd: assoc_ws must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if ((rc = ensure_packed(c, st))) return rc;
    if (!c->as_pg) HIP_TRY(gmalloc((void**)&c->as_pg, sizeof(float) * AS_PG * (size_t)c->G));
    const bool variant = c->has_edges || c->abs_sta != nullptr;
    if (variant && !c->as_ps) HIP_TRY(gmalloc((void**)&c->as_ps, sizeof(float) * AS_PS * (size_t)edge_rows_sta(c)));
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
    if (c->pcsr) {       // irregular product graph: product-level CSRs, the source node of every product node from the row ranges
        if ((rc = ensure_src_of(c, st))) return rc;
        a.sta_rowptr = c->p_sta_rowptr; a.sta_col = c->p_sta_col; a.src_rowptr = c->p_src_rowptr; a.src_col = c->p_src_col;
        a.sta_user = nullptr; a.src_of = c->p_src_of; a.ptile = c->ptile16;
        const long long ntiles = (c->P + 15) / 16;
        const int grid = (int)std::max<long long>(8, std::min<long long>((ntiles + 3) / 4, (long long)c->num_cu * std::max(1, c->bpc1)) / 8 * 8);
        a.packed = c->packed[2];
        k_assoc_a<true><<<grid, 256, 0, st>>>(a);
        a.packed = c->packed[3];
        k_assoc_b<true><<<grid, 256, 0, st>>>(a);
    } else {
        const int grid = da_grid(c, (long long)c->G * c->T, std::max(1, c->bpc1));
        a.packed = c->packed[2];
        // (40 registers, 21 KB of LDS: four workgroups per CU where stage 1's occupancy gave two -- 369 -> 322 us, six: 330; tools/assoc_ab.sh)
        k_assoc_a<false><<<da_grid(c, (long long)c->G * c->T, 4), 256, 0, st>>>(a);
        a.packed = c->packed[3];
        // k_assoc_b is bound by its 23 gathered 128-B rows per node and, at 4 waves per workgroup, by LDS to 8 waves per CU (every
        // workgroup holds its own 52-KB copy of the weight image): ONE workgroup of 16 waves per CU shares one copy -- 969 -> 824 us at
        // config 2 (12 waves: 882; 8 waves in one workgroup: 987; `tools/assoc_ab.sh`)
        k_assoc_b<false, 16><<<std::max(8, c->num_cu / 8 * 8), 1024, 0, st>>>(a);
    }
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
    return c ? (size_t)GR_BLOCKS * 16 * (size_t)c->P + train_part_floats(c) + (size_t)c->G * c->T * 32 + 64 + static_scratch_floats(c) : 0;
}

int genie_assoc_train_fwd(genie_ctx* c, const float* y_latent, const float* mask_src, const float* x_latent, const float* mask,
                          const float* edge_attr, float* out, float* asave, void* assoc_ws, void* ws, void* stream) {
    if (!c || !asave) return fail(GENIE_ERR_ARG, "genie_assoc_train_fwd: null argument");
    int rc = train_check(c, "genie_assoc_train_fwd", true, true);
    if (rc) return rc;
    return assoc_fwd_impl(c, y_latent, mask_src, x_latent, mask, edge_attr, out, assoc_ws, ws, stream, asave);
}

int genie_assoc_train_bwd(genie_ctx* c, const float* y_latent, const float* mask_src, const float* x_latent, const float* mask,
                          const float* edge_attr, const float* asave, const float* d_s, float* scratch, float* d_ylat_out,
                          float* grad_blob, void* stream) {
    if (!c || !y_latent || !mask_src || !x_latent || !mask || !edge_attr || !asave || !d_s || !scratch || !d_ylat_out || !grad_blob)
        return fail(GENIE_ERR_ARG, "genie_assoc_train_bwd: null argument");
    int rc;
    if ((rc = train_check(c, "genie_assoc_train_bwd", true, true))) return rc;
    hipStream_t st = (hipStream_t)stream;
    if ((rc = ensure_packed(c, st))) return rc;
    if ((rc = ensure_reversed(c))) return rc;
    if (!c->as_pg) HIP_TRY(gmalloc((void**)&c->as_pg, sizeof(float) * AS_PG * (size_t)c->G));
    assoc_pre_launch(c, y_latent, mask_src, st);          // pg[31] = mask1[g] (another forward may have overwritten the table)
    HIP_TRY(hipMemsetAsync(grad_blob, 0, sizeof(float) * g_raw_total, st));
    TrArgs a;
    memset(&a, 0, sizeof(a));
    a.S = c->S; a.G = c->G; a.T = c->T; a.seg = std::max(1, c->seg);
    a.nxcd = 8;
    a.P = c->P; a.order = c->order;
    a.r_sta_rowptr = c->r_sta_rowptr; a.r_sta_col = c->r_sta_col; a.r_sta_w = c->r_sta_w;
    a.r_src_rowptr = c->r_src_rowptr; a.r_src_col = c->r_src_col; a.r_src_w = c->r_src_w;
    a.r_sta_cw = c->r_sta_cw; a.r_src_cw = c->r_src_cw;
    a.mask = mask; a.edge_attr = edge_attr; a.save = asave; a.x_latent = x_latent; a.pg = c->as_pg;
    a.gr = scratch; a.part = scratch + (size_t)GR_BLOCKS * 16 * (size_t)c->P;
    a.zsum = a.part + train_part_floats(c);
    float* sscr = a.zsum + (size_t)c->G * c->T * 32 + 64;
    const bool variant = c->has_edges || c->abs_sta != nullptr;
    if (c->pcsr) {
        if ((rc = ensure_src_of(c, st))) return rc;
        a.src_of = c->p_src_of; a.ptile = c->ptile16;
        a.r_sta_rowptr = c->rp_sta_rowptr; a.r_sta_cw = c->rp_sta_cw; a.r_src_rowptr = c->rp_src_rowptr; a.r_src_cw = c->rp_src_cw;
        a.r_sta_col = a.r_src_col = nullptr; a.r_sta_w = a.r_src_w = nullptr;
    }
    a.sv_t = AV_T; a.sv_up = AV_UV; a.sv_vp = AV_UV + 2;
    const int grid = train_grid(c);
    const bool o32a = !c->pcsr && (unsigned long long)AV_BLOCKS * (unsigned long long)c->P * 64ull < (1ull << 32);      // (20 kept blocks: P < 3.35 M)
    const int tms[4] = {TM_AB3, TM_AB2, TM_AB1, TM_AB0};
    const int pls[4] = {-1, PL_TAB2, PL_TAB1, PL_TAB0};
    for (int s = 0; s < 4; ++s) {
        const int tm = tms[s];
        a.packed = pls[s] >= 0 ? c->packed[pls[s]] : nullptr; a.n_acc = c->n_acc[tm]; a.n_vec = c->n_vec[tm];
        const bool split1 = s == 1 && !c->pcsr && o32a && b1_split_on();      // k_train_b1s: two waves per tile, one partial per pair
        const int grid_s = (s == 1 && !c->pcsr && !split1) ? std::max(8, grid / 2 / 8 * 8) : grid;      // k_train_b1: one workgroup per CU (see da_train_bwd_impl)
        if (c->pcsr) {
            if (s == 0) k_as_b3<true><<<grid, 256, 0, st>>>(a, d_s, c->raw + g_params[W_AS_ACT2].off);
            else if (s == 1) k_train_b1p<true><<<grid, 256, 0, st>>>(a);
            else if (s == 2) k_as_b1<true><<<grid, 256, 0, st>>>(a);
            else k_as_b0<true><<<grid, 256, 0, st>>>(a);
        } else {
            if (s == 0) k_as_b3<false><<<grid, 256, 0, st>>>(a, d_s, c->raw + g_params[W_AS_ACT2].off);
            else if (s == 1) {
                if (split1) k_train_b1s<true><<<grid_s, 256, 0, st>>>(a);
                else if (o32a) k_train_b1<true, true><<<grid_s, 256, 0, st>>>(a); else k_train_b1<true><<<grid_s, 256, 0, st>>>(a);
            }
            else if (s == 2) { if (o32a) k_as_b1<false, true><<<grid, 256, 0, st>>>(a); else k_as_b1<false><<<grid, 256, 0, st>>>(a); }
            else k_as_b0<false><<<grid, 256, 0, st>>>(a);
        }
        const int stride = a.n_acc * 256 + a.n_vec * 16 + 16;
        k_train_reduce<<<(stride + 31) / 32, 256, 0, st>>>(a.part, grid_s * (split1 ? 2 : 4), a.n_acc, a.n_vec, c->n_sc[tm], c->d_acc[tm], c->d_vec[tm],
                                                            c->d_sc[tm], grad_blob, 0);
        // static terms of the two other model definitions: the layer-2 ones now (the next pass writes dtrp over do), the rest at the end;
        // on an irregular product graph the layer-1 ones after k_as_b1 already (k_as_b0<PCSR> leaves its d z1 rows in the dt blocks)
        if (variant && c->pcsr) {
            if ((s == 1 || s == 2) && (rc = static_term_grads(c, a.gr, sscr, grad_blob, st, s == 1 ? 1 : 6, true))) return rc;
        } else if (variant && (s == 1 || s == 3) && (rc = static_term_grads(c, a.gr, sscr, grad_blob, st, s == 1 ? 1 : 6, true))) return rc;
    }
    if (c->pcsr)      // k_as_b0<PCSR> left the d z1 rows in the dt blocks: one sum row per source node
        k_seg_sum_blocks<<<(c->G * 32 + 255) / 256, 256, 0, st>>>(a.gr + (size_t)GR_DT * 16 * (size_t)c->P, c->P, c->seg_rowptr, c->G, a.zsum);
    {
        AgArgs g;
        memset(&g, 0, sizeof(g));
        g.G = c->G; g.T = part_T(c); g.zsum = a.zsum; g.y_latent = y_latent; g.timg = c->packed[PL_TAG]; g.d_ylat = d_ylat_out;
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
    if (n_query >= 16384 && k <= 10) {        // a lane per query (k_knn_t): enough queries to fill the device that way
        const int nbt = (n_query + 255) / 256;
        if (k <= 8) k_knn_t<8><<<nbt, 256, 0, st>>>(x_context, n_context, x_query, n_query, k, exclude_self, out_idx);
        else k_knn_t<10><<<nbt, 256, 0, st>>>(x_context, n_context, x_query, n_query, k, exclude_self, out_idx);
        HIP_TRY(hipGetLastError());
        return GENIE_OK;
    }
    if (n_query <= 64 && n_context >= 4096 && k <= 10) {      // a workgroup of 16 waves per query (k_knn_b)
        if (k <= 8) k_knn_b<8><<<n_query, 1024, 0, st>>>(x_context, n_context, x_query, n_query, k, exclude_self, out_idx);
        else k_knn_b<10><<<n_query, 1024, 0, st>>>(x_context, n_context, x_query, n_query, k, exclude_self, out_idx);
        HIP_TRY(hipGetLastError());
        return GENIE_OK;
    }
    const int nb = (n_query + 3) / 4;
    if (k <= 8) k_knn<8><<<nb, 256, 0, st>>>(x_context, n_context, x_query, n_query, k, exclude_self, out_idx);
    else if (k <= 10) k_knn<10><<<nb, 256, 0, st>>>(x_context, n_context, x_query, n_query, k, exclude_self, out_idx);
    else k_knn<16><<<nb, 256, 0, st>>>(x_context, n_context, x_query, n_query, k, exclude_self, out_idx);
    HIP_TRY(hipGetLastError());
    return GENIE_OK;
}

int genie_product_check(const int64_t* A_in_sta, int64_t E_sta, const int64_t* A_in_src, int64_t E_src, int n_sta, int n_grid,
                        int32_t* flags, void* stream) {
    if (!A_in_sta || !A_in_src || !flags) return fail(GENIE_ERR_ARG, "genie_product_check: null argument");
    if (n_sta < 1 || n_grid < 1 || E_sta < 1 || E_src < 1 || E_sta % n_grid != 0 || E_src % n_sta != 0 ||
        E_sta / n_grid >= (1ll << 31) || E_src / n_sta >= (1ll << 31))
        return fail(GENIE_ERR_ARG, "genie_product_check: edge counts must be positive multiples of (n_grid, n_sta)");
    const long long n = (long long)E_sta + E_src;
    const int grid = (int)std::min<long long>((n + 255) / 256, 256 * 32);
    k_product_check<<<grid, 256, 0, (hipStream_t)stream>>>((const long long*)A_in_sta, (long long)E_sta, (int)(E_sta / n_grid),
                                                          (const long long*)A_in_src, (long long)E_src, (int)(E_src / n_sta), n_sta, n_grid, flags);
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
                   float dt, const float* dt_partition, float eps, const float* tlatent, int tl_stride, int tl_col, const float* tpick,
                   const int32_t* ipick, const float* phase_label, int n_picks, float* out, void* stream) {
    if (!c || !s_rows || !a_edges || !tlatent || !tpick || !ipick || !phase_label || !out) return fail(GENIE_ERR_ARG, "genie_lslc_fwd: null argument");
    if (phase_head < 0 || phase_head > 1 || l_dt < (dt_partition ? 2 : 1) || n_edges < LS_K || (!dt_partition && !(dt > 0.f)) || !(eps > 0.f) || tl_stride < 1 ||
        tl_col < 0 || tl_col >= tl_stride)
        return fail(GENIE_ERR_ARG, "genie_lslc_fwd: bad argument");
    if (n_picks < 1) return GENIE_OK;
    hipStream_t st = (hipStream_t)stream;
    { int rcp = ensure_packed(c, st); if (rcp) return rcp; }
    LsArgs a;
    memset(&a, 0, sizeof(a));
    a.n_picks = n_picks; a.l_dt = l_dt; a.n_edges = n_edges; a.t0 = t0; a.dt = dt; a.dtp = dt_partition; a.eps = eps;
    a.s = s_rows; a.A_edges = a_edges; a.tlatent = tlatent; a.tl_stride = tl_stride; a.tl_col = tl_col;
    a.tpick = tpick; a.ipick = ipick; a.phase = phase_label; a.img = c->packed[phase_head == 0 ? PL_LSP : PL_LSS]; a.out = out;
    a.flag = c->h_inflag ? c->h_inflag + 1 : nullptr;
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
                   float dt, const float* dt_partition, float eps, const float* tlatent, int tl_stride, int tl_col, const float* tpick,
                   const int32_t* ipick, const float* phase_label, int n_picks, const float* d_out, float* erow, int32_t* etgt,
                   float* part_scratch, float* grad_blob, void* stream) {
    if (!c || !s_rows || !a_edges || !tlatent || !tpick || !ipick || !phase_label || !d_out || !erow || !etgt || !part_scratch || !grad_blob)
        return fail(GENIE_ERR_ARG, "genie_lslc_bwd: null argument");
    if (phase_head < 0 || phase_head > 1 || l_dt < (dt_partition ? 2 : 1) || n_edges < LS_K || (!dt_partition && !(dt > 0.f)) || !(eps > 0.f) || tl_stride < 1 ||
        tl_col < 0 || tl_col >= tl_stride || n_picks < 1)
        return fail(GENIE_ERR_ARG, "genie_lslc_bwd: bad argument");
    hipStream_t st = (hipStream_t)stream;
    { int rcp = ensure_packed(c, st); if (rcp) return rcp; }
    LbArgs b;
    memset(&b, 0, sizeof(b));
    b.f.n_picks = n_picks; b.f.l_dt = l_dt; b.f.n_edges = n_edges; b.f.t0 = t0; b.f.dt = dt; b.f.dtp = dt_partition; b.f.eps = eps;
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

int genie_set_phase_types(genie_ctx* c, int use_phase_types) {
    if (!c) return fail(GENIE_ERR_ARG, "genie_set_phase_types: null context");
    c->no_phase = use_phase_types ? 0 : 1;
    return GENIE_OK;
}

int genie_set_subgraph_stations(genie_ctx* c, const int32_t* sta_of_prod, void* stream) {
    if (!c || !sta_of_prod) return fail(GENIE_ERR_ARG, "genie_set_subgraph_stations: null argument");
    if (!c->pcsr) return fail(GENIE_ERR_STATE, "genie_set_subgraph_stations: the context is a Cartesian product graph (station = p % n_sta)");
    if (!c->p_sta_of) HIP_TRY(gmalloc((void**)&c->p_sta_of, sizeof(int32_t) * (size_t)c->P));
    HIP_TRY(hipMemcpyAsync(c->p_sta_of, sta_of_prod, sizeof(int32_t) * (size_t)c->P, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return GENIE_OK;
}

int genie_set_stage2_workmap(genie_ctx* c, int blocks_of_four) {
    if (!c) return fail(GENIE_ERR_ARG, "genie_set_stage2_workmap: null context");
    c->s2_wgmap = blocks_of_four < 0 ? (c->S >= 1024) : (blocks_of_four != 0);
    return GENIE_OK;
}

int genie_set_sign_input(genie_ctx* c, int use_sign_input) {
    if (!c) return fail(GENIE_ERR_ARG, "genie_set_sign_input: null context");
    c->sign_input = use_sign_input ? 1 : 0;
    return GENIE_OK;
}

int genie_set_tail_precision(genie_ctx* c, int fp64_chains) {
    if (!c) return fail(GENIE_ERR_ARG, "genie_set_tail_precision: null context");
    c->tail_f32 = fp64_chains ? 0 : 1;
    return GENIE_OK;
}

int genie_set_stage_precision(genie_ctx* c, int mode) {
    if (!c || mode < 0 || mode > 2) return fail(GENIE_ERR_ARG, "genie_set_stage_precision: mode must be 0 (auto), 1 (f16x2) or 2 (fp32)");
    c->prec_mode = mode;
    c->xs_slice = c->xs_mask = nullptr; c->xs_ws = nullptr;
    return GENIE_OK;
}

int genie_stage_precision(genie_ctx* c, int* mode, int* f16x2_active, float* act_bound, float* weight_bound, void* stream) {
    if (!c) return fail(GENIE_ERR_ARG, "genie_stage_precision: null context");
    int rc = ensure_packed(c, (hipStream_t)stream);        // the range guard belongs to the committed weights
    if (rc) return rc;
    if (mode) *mode = c->prec_mode;
    if (f16x2_active) *f16x2_active = (c->pcsr ? pcsr_h2_on(c) : h2_on(c)) ? 1 : 0;
    if (act_bound) *act_bound = c->range_act;
    if (weight_bound) *weight_bound = c->range_w;
    return GENIE_OK;
}

int genie_input_range(genie_ctx* c, float* max_seen, float* limit, int reset) {
    if (!c) return fail(GENIE_ERR_ARG, "genie_input_range: null context");
    // reset = one atomic fetch-and-clear (kernels still in flight may be OR-ing / max-ing into the word: a plain store after a separate
    // read could drop what they wrote in between); the value reported is exactly the value cleared
    unsigned u = 0u;
    if (c->h_inflag) u = reset ? __atomic_exchange_n(c->h_inflag, 0u, __ATOMIC_SEQ_CST) : *(volatile unsigned*)c->h_inflag;
    if (max_seen) {
        float f;
        memcpy(&f, &u, 4);
        *max_seen = f;
    }
    if (limit) *limit = input_limit(c);
    return GENIE_OK;
}

int genie_index_flags(genie_ctx* c, unsigned* flags, int reset) {
    if (!c) return fail(GENIE_ERR_ARG, "genie_index_flags: null context");
    unsigned u = 0u;      // (reset: atomic fetch-and-clear, see genie_input_range)
    if (c->h_inflag) u = reset ? __atomic_exchange_n(c->h_inflag + 1, 0u, __ATOMIC_SEQ_CST) : *(volatile unsigned*)(c->h_inflag + 1);
    if (flags) *flags = u;
    return GENIE_OK;
}

int genie_index_check(genie_ctx* c, const int64_t* idx, int64_t n, int64_t lo, int64_t hi, unsigned bit, void* stream) {
    if (!c || (!idx && n > 0)) return fail(GENIE_ERR_ARG, "genie_index_check: null argument");
    if (bit == 0u || (bit & (bit - 1u)) != 0u || bit == 1u) return fail(GENIE_ERR_ARG, "genie_index_check: `bit` must be one bit above bit 0");
    if (n < 1 || !c->h_inflag) return GENIE_OK;
    const int grid = (int)std::min<int64_t>((n + 255) / 256, 1024);
    k_index_check<<<grid, 256, 0, (hipStream_t)stream>>>((const long long*)idx, n, lo, hi, bit, c->h_inflag + 1);
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
    k_export<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(src, rows, pitch, ncol, out, sta_order_on(c) ? c->sta_perm : nullptr, c->S,
                                                                         (c->ws_np && which != 1) ? 1 : 0);
    HIP_TRY(hipGetLastError());
    return GENIE_OK;
}

}  // extern "C"
