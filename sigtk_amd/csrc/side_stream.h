// side_stream.h -- a second stream for kernels that may run beside a launch's main kernel (the packed short reads of
// `event`, the long reads of `stat` / `jnn`): a small pool per device, one fork .. join per launch.
#pragma once
#include <mutex>

#include "sgk_common.h"

namespace sgk {

struct SideStream {
    std::mutex mu;
    hipStream_t s = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    bool tried = false;
};
// returns a locked side stream of the current device (unlock with x->mu.unlock()), or null (event_kernels.hip)
SideStream *side_acquire(int priority /* < 0 low, 0 the default, > 0 high */);

// One fork .. join on a side stream: joins on every exit path (an error return in between must not leave the caller's
// stream unordered behind work that still writes the workspace).
struct SideFork {
    SideStream *x = nullptr;
    hipStream_t main = nullptr;
    bool open(int priority, hipStream_t st) {
        // (a stream that is being captured into a graph keeps everything in itself: the library's events and streams are
        // not part of the caller's capture)
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return false;
        x = side_acquire(priority);
        if (!x) return false;
        main = st;
        if (hipEventRecord(x->fork, st) == hipSuccess && hipStreamWaitEvent(x->s, x->fork, 0) == hipSuccess) return true;
        x->mu.unlock();
        x = nullptr;
        return false;
    }
    hipStream_t stream() const { return x ? x->s : main; }
    void join() {
        if (!x) return;
        const bool ok = hipEventRecord(x->join, x->s) == hipSuccess && hipStreamWaitEvent(main, x->join, 0) == hipSuccess;
        if (!ok) (void)hipStreamSynchronize(x->s);
        x->mu.unlock();
        x = nullptr;
    }
    ~SideFork() { join(); }
};

}  // namespace sgk
