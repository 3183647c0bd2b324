// Host/device-shared descriptors of the stage executor (stage_exec.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dsat {

struct StageJob {            // 32 B, built by the host (dsat_api.hip: build_stage_plan)
    uint32_t w_off;          // float4 index of (first tile, k-block 0, lane 0) in the weight slab
    uint32_t tile_stride;    // float4 units between adjacent column tiles (= kblocks * 64)
    uint16_t kblocks;
    uint16_t ntiles;         // valid tiles in this job, 1..TPW
    uint16_t in_buf, out_buf;   // LDS buffer ids (StagePlan::buf_off / buf_ld)
    uint16_t out_kind;       // 0 = hidden layer: leaky_relu -> LDS out_buf; >0 = final-store kind
    uint16_t col0;           // first output column of the job within its layer
    uint32_t bias_off;       // float offset of bias[col0] inside the kernel's LDS bias region
    uint32_t pad[2];
};
static_assert(sizeof(StageJob) == 32, "StageJob is read as two 16-B scalar loads");

constexpr int STAGE_MAX = 4, STAGE_BUFS = 8;
constexpr int STAGE_TPW_MAX = 4;   // column tiles per job: 4 for 16-row tiles, 2 for 64-row tiles
constexpr int STAGE_PD = 4;    // k-blocks of weights in flight per wave
struct StagePlan {
    const StageJob* jobs;
    const float4* wslab;     // fragment-ordered weights of every layer (dsat_weights::slab)
    const float* bias_pack;  // biases of this kernel's layers, copied to LDS at off_bias
    int nstages;
    int first[STAGE_MAX + 1];
    int buf_off[STAGE_BUFS], buf_ld[STAGE_BUFS];
    int bias_floats, off_bias;
};


}  // namespace dsat
