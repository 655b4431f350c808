// stat_args.h -- argument block of the stat / jnn / prefix kernels.
#pragma once
#include "sgk_common.h"

namespace sgk {

enum { REG_WHOLE = 0, REG_ADAPT = 1, REG_POLYA = 2, REG_TAIL = 3 };

struct StatArgs {
    sgk_batch_t b;               // device pointers
    sgk_stat_rec_t *stat;        // stat output (or null)
    sgk_prefix_rec_t *prefix;    // prefix output / region source (or null)
    const uint64_t *seg_slots;   // jnn arena
    int32_t *seg_x, *seg_y;
    uint32_t *n_segs;
    uint32_t *err_count;         // workspace: number of reads whose segments overflowed their slots
    float *pa_out;               // stat+pa fused: pA of every sample, written by the median pass (or null)
};

int check_batch(const sgk_batch_t *b);
int launch_pa(const sgk_batch_t *b, float *out, hipStream_t st);
int launch_stat(const StatArgs &a, hipStream_t st);  // a.pa_out != null: fused stat + pa
int launch_jnn(const StatArgs &a, int rna, hipStream_t st);
int launch_prefix(const StatArgs &a, int rna, int pore, hipStream_t st);

}  // namespace sgk
