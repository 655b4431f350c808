// host_util.h -- helpers of the host-pointer convenience layer (sgk_*_host and the shims).
#pragma once
#include <string.h>

#include <vector>

#include "sgk_common.h"

namespace sgk {

struct DevBuf {
    void *p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    int alloc(size_t bytes) {
        SGK_HIP_TRY(hipMalloc(&p, bytes ? bytes : 64));
        return SGK_OK;
    }
    template <typename T>
    T *as() {
        return static_cast<T *>(p);
    }
};

// A host batch (CSR offsets, reads back to back) repacked so that every read starts on a
// 64-sample (128-byte) boundary, uploaded to the device.
struct DeviceBatch {
    std::vector<uint64_t> offsets;  // n_reads, aligned starts
    std::vector<uint32_t> lengths;  // n_reads
    uint64_t n_samples = 0;         // allocation length (multiple of 64)
    uint32_t max_len = 0;
    DevBuf d_samples, d_offsets, d_lengths, d_dig, d_off, d_rng;
    sgk_batch_t view;

    int upload(const sgk_host_batch_t *hb) {
        if (!hb) return SGK_ERR_ARG;
        const size_t nr = hb->n_reads;
        if (nr && (!hb->samples || !hb->offsets || !hb->digitisation || !hb->offset || !hb->range))
            return SGK_ERR_ARG;
        if (sgk_device_count() <= 0) return SGK_ERR_NODEVICE;
        offsets.resize(nr);
        lengths.resize(nr);
        uint64_t o = 256;  // head room: the event fast path wants up to 256 readable samples before a read
        for (size_t r = 0; r < nr; ++r) {
            const uint64_t n = hb->offsets[r + 1] - hb->offsets[r];
            if (n > 0x7fffffffull) return SGK_ERR_ARG;  // nsample is int32 in the reference (misc.c:20)
            offsets[r] = o;
            lengths[r] = (uint32_t)n;
            if (n > max_len) max_len = (uint32_t)n;
            o += round_up(n, 64);
        }
        n_samples = o + 64;  // tail room (>= 16 samples after the last read)
        std::vector<int16_t> packed((size_t)n_samples, 0);
        for (size_t r = 0; r < nr; ++r)
            if (lengths[r])
                memcpy(&packed[offsets[r]], hb->samples + hb->offsets[r], (size_t)lengths[r] * sizeof(int16_t));
        int rc;
        if ((rc = d_samples.alloc(packed.size() * sizeof(int16_t))) != SGK_OK) return rc;
        if ((rc = d_offsets.alloc(nr * sizeof(uint64_t))) != SGK_OK) return rc;
        if ((rc = d_lengths.alloc(nr * sizeof(uint32_t))) != SGK_OK) return rc;
        if ((rc = d_dig.alloc(nr * sizeof(double))) != SGK_OK) return rc;
        if ((rc = d_off.alloc(nr * sizeof(double))) != SGK_OK) return rc;
        if ((rc = d_rng.alloc(nr * sizeof(double))) != SGK_OK) return rc;
        SGK_HIP_TRY(hipMemcpy(d_samples.p, packed.data(), packed.size() * sizeof(int16_t), hipMemcpyHostToDevice));
        if (nr) {
            SGK_HIP_TRY(hipMemcpy(d_offsets.p, offsets.data(), nr * sizeof(uint64_t), hipMemcpyHostToDevice));
            SGK_HIP_TRY(hipMemcpy(d_lengths.p, lengths.data(), nr * sizeof(uint32_t), hipMemcpyHostToDevice));
            SGK_HIP_TRY(hipMemcpy(d_dig.p, hb->digitisation, nr * sizeof(double), hipMemcpyHostToDevice));
            SGK_HIP_TRY(hipMemcpy(d_off.p, hb->offset, nr * sizeof(double), hipMemcpyHostToDevice));
            SGK_HIP_TRY(hipMemcpy(d_rng.p, hb->range, nr * sizeof(double), hipMemcpyHostToDevice));
        }
        view.samples = d_samples.as<int16_t>();
        view.offsets = d_offsets.as<uint64_t>();
        view.lengths = d_lengths.as<uint32_t>();
        view.digitisation = d_dig.as<double>();
        view.offset = d_off.as<double>();
        view.range = d_rng.as<double>();
        view.n_reads = hb->n_reads;
        view.max_read_len = max_len;
        view.n_samples = n_samples;
        return SGK_OK;
    }
};

// capacity-layout slots for a per-read variable-length output (events, segments)
template <typename F>
static inline std::vector<uint64_t> make_slots(const std::vector<uint32_t> &lengths, F slots_for) {
    std::vector<uint64_t> s(lengths.size() + 1);
    uint64_t o = 0;
    for (size_t r = 0; r < lengths.size(); ++r) {
        s[r] = o;
        o += slots_for(lengths[r]);
    }
    s[lengths.size()] = o;
    return s;
}

}  // namespace sgk
