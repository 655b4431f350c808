// synth.h -- deterministic synthetic nanopore reads, shared by host and device.
//
// Counter-based and integer-only, so the host generator (sgk_synth_reads_host) and the
// device kernel produce bit-identical int16 samples: sample i of read r is a pure function
// of (seed, r, i).  Signal model: piecewise-constant level ~ N(90 pA, 12 pA) held for a
// geometric dwell (mean 9 samples "DNA", 36 samples "RNA"), plus ~N(0, 1.5 pA) noise, with
// digitisation 8192, range 1402.882324 and an integer offset in [0,20) per read.
// kind 1 ("RNA") reads of >= 20000 samples additionally carry a leader, a low-current
// adaptor stretch and a flat polyA stretch so that the prefix/jnn subtools find something.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SGK_HD __host__ __device__ inline
#else
#define SGK_HD static inline
#endif

#define SGK_SYNTH_DIGITISATION 8192.0
#define SGK_SYNTH_RANGE 1402.882324

SGK_HD uint64_t sgk_splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

typedef struct {
    uint64_t key;
    int64_t n;
    int32_t mu;      // body level mean in raw units
    int32_t offset;  // per-read integer offset (slow5 'offset' field)
    int32_t kind;
    int64_t s0, s1, s2;  // RNA structure: [0,s0) leader, [s0,s1) adaptor, [s1,s2) polyA, [s2,n) body
} sgk_synth_read_t;

SGK_HD sgk_synth_read_t sgk_synth_read_init(uint64_t seed, uint64_t read_index, int64_t n, int kind) {
    sgk_synth_read_t R;
    R.key = sgk_splitmix64(seed ^ (0xD1B54A32D192ED03ULL * (read_index + 1)));
    R.n = n;
    R.offset = (int32_t)(R.key % 20);
    R.mu = 525 - R.offset;
    R.kind = kind;
    R.s0 = R.s1 = R.s2 = 0;
    if (kind == 1 && n >= 20000) {
        const uint64_t k2 = sgk_splitmix64(R.key ^ 0xA5A5A5A5A5A5A5A5ULL);
        R.s0 = 500 + (int64_t)(k2 % 1000);
        R.s1 = R.s0 + 3000 + (int64_t)((k2 >> 16) % 3000);
        R.s2 = R.s1 + 1500 + (int64_t)((k2 >> 32) % 3000);
    }
    return R;
}

SGK_HD uint32_t sgk_sum16x4(uint64_t h) {
    return (uint32_t)(h & 0xFFFF) + (uint32_t)((h >> 16) & 0xFFFF) + (uint32_t)((h >> 32) & 0xFFFF) +
           (uint32_t)(h >> 48);
}

SGK_HD int16_t sgk_synth_sample(const sgk_synth_read_t &R, int64_t i) {
    // region lookup
    int region = 3;  // body
    int64_t rstart = 0;
    if (R.s2 > 0) {
        if (i < R.s0) { region = 0; rstart = 0; }
        else if (i < R.s1) { region = 1; rstart = R.s0; }
        else if (i < R.s2) { region = 2; rstart = R.s1; }
        else { region = 3; rstart = R.s2; }
    }
    const uint64_t h = sgk_splitmix64(R.key + 2ULL * (uint64_t)i);
    const uint32_t s3 = (uint32_t)((h >> 16) & 0xFFFF) + (uint32_t)((h >> 32) & 0xFFFF) + (uint32_t)(h >> 48);
    int32_t level, noise;
    if (region == 2) {
        level = R.mu - 120 + 175;               // flat polyA, ~30 pA above the adaptor mean
        noise = (int32_t)((s3 * 20u) >> 17) - 15;
    } else {
        const uint32_t sw = (R.kind == 1) ? 1820u : 7282u;  // 65536 / mean dwell
        int64_t j = i;
        while (j > rstart) {
            const uint64_t hj = (j == i) ? h : sgk_splitmix64(R.key + 2ULL * (uint64_t)j);
            if ((uint32_t)(hj & 0xFFFF) < sw) break;
            j--;
        }
        const uint32_t ih = sgk_sum16x4(sgk_splitmix64(R.key + 2ULL * (uint64_t)j + 1ULL));
        if (region == 1) level = R.mu - 120 - 121 + (int32_t)((ih * 35u) / 37837u);
        else level = R.mu - 242 + (int32_t)((ih * 70u) / 37837u);
        noise = (int32_t)((s3 * 35u) >> 17) - 26;
    }
    int32_t v = level + noise;
    v = v < 0 ? 0 : (v > 4000 ? 4000 : v);
    return (int16_t)v;
}
