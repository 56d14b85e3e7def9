// Device scratch of the coordinate path (coords.hip, contours_device.hip): include after ap_common.h.
#pragma once
#include <vector>
#include "ap_common.h"

namespace ap {

// Scratch buffers of one call.  hipMalloc / hipFree per call cost more than the kernels here, and hipFree synchronises the whole
// DEVICE: with eight coordinate workers beside the segmenter's graph replays every slide stalled every other thread's queue.
// Each calling thread therefore keeps one grow-only arena per device (thread_local; released when the thread exits); a call
// carves its buffers from it in order and the arena is reset when the call returns (every call ends with a stream
// synchronisation, so nothing is in flight then).  Growth = one new, larger allocation; the old one is freed at the next reset.
struct Arena {
    char* base = nullptr; size_t cap = 0, used = 0; int device = -1;
    std::vector<void*> retired;        // outgrown blocks, still referenced by the running call
    ~Arena() { int d = 0; if (hipGetDevice(&d) == hipSuccess) release(); }      // (no runtime left at process exit: nothing to free)
    void release() {
        if (base) (void)hipFree(base);
        for (void* p : retired) (void)hipFree(p);
        base = nullptr; cap = used = 0; retired.clear();
    }
    void reset() {
        used = 0;
        for (void* p : retired) (void)hipFree(p);
        retired.clear();
    }
    int take(size_t bytes, void** out) {
        int dev = 0;
        AP_HIP_CHECK(hipGetDevice(&dev));
        if (dev != device) { release(); device = dev; }
        bytes = (bytes + 255) & ~(size_t)255;
        if (used + bytes > cap) {
            // buffers handed out earlier in this call stay valid in the retired block; the new block starts empty
            const size_t want = (cap * 2 > used + bytes ? cap * 2 : used + bytes) + (1u << 20);
            void* p = nullptr;
            AP_HIP_CHECK(hipMalloc(&p, want));
            if (base) retired.push_back(base);
            base = (char*)p; cap = want; used = 0;
        }
        *out = base + used;
        used += bytes;
        return AP_OK;
    }
};
inline Arena& arena() { thread_local Arena a; return a; }
struct ArenaScope { ~ArenaScope() { arena().reset(); } };

template <typename T> struct DevBuf {
    T* p = nullptr;
    int alloc(size_t n) { return arena().take((n ? n : 1) * sizeof(T), (void**)&p); }
};


}  // namespace ap
