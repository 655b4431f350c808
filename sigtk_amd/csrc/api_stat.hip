// TEMPORARY stubs (replaced by the real stat / jnn / prefix entry points)
#include "sgk_common.h"
extern "C" {
size_t sgk_stat_workspace_bytes(uint32_t, uint64_t, uint32_t) { return 64; }
int sgk_stat(const sgk_batch_t *, sgk_stat_rec_t *, void *, size_t, void *) { return SGK_ERR_ARG; }
int sgk_stat_pa(const sgk_batch_t *, sgk_stat_rec_t *, float *, void *, size_t, void *) { return SGK_ERR_ARG; }
size_t sgk_jnn_workspace_bytes(uint32_t, uint64_t, uint32_t) { return 64; }
int sgk_jnn(const sgk_batch_t *, int, const uint64_t *, int32_t *, int32_t *, uint32_t *, void *, size_t, void *) { return SGK_ERR_ARG; }
size_t sgk_prefix_workspace_bytes(uint32_t, uint64_t, uint32_t) { return 64; }
int sgk_prefix(const sgk_batch_t *, int, int, sgk_prefix_rec_t *, void *, size_t, void *) { return SGK_ERR_ARG; }
int sgk_stat_host(const sgk_host_batch_t *, sgk_stat_rec_t *) { return SGK_ERR_ARG; }
int sgk_jnn_host(const sgk_host_batch_t *, int, sgk_segs_host_t *) { return SGK_ERR_ARG; }
void sgk_segs_host_free(sgk_segs_host_t *) {}
int sgk_prefix_host(const sgk_host_batch_t *, int, int, sgk_prefix_rec_t *) { return SGK_ERR_ARG; }
}
