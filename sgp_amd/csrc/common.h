// Shared host-side helpers for libsgp_amd.so (gfx950 only; no other target is supported).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/sgp_amd.h"

namespace sgp {

char* err_buf();                       // thread-local, 512 bytes
int fail(int code, const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, "%s: %s", what, hipGetErrorString(e));
    return 0;
}

// SGP_TUNE="key=value,key=value": the one debug / tuning hook (keys: sgp_amd/tune.py); default when absent
long tune(const char* key, long dflt);

// A zeroed 256-byte device word block for ONE launch's cross-workgroup counters (arrival counts, pacing):
// slots come from a per-device ring of 64, cleared on the launch stream, so launches in flight on different
// streams never share a counter.  nullptr when the allocation fails (callers then run unpaced).
unsigned* sync_slot(hipStream_t s);
// A second lane beside the caller's stream (per host thread and device, created on first use): work forked onto
// `stream` after fork(main) runs concurrently with what the caller enqueues on `main` next; join(main) makes `main`
// wait for it.  Both are event edges (legal under stream capture).  nullptr when the lane cannot be created.
struct SideLane {
    hipStream_t stream;
    hipEvent_t forked, done;
    bool fork(hipStream_t main) { return hipEventRecord(forked, main) == hipSuccess && hipStreamWaitEvent(stream, forked, 0) == hipSuccess; }
    bool join(hipStream_t main) { return hipEventRecord(done, stream) == hipSuccess && hipStreamWaitEvent(main, done, 0) == hipSuccess; }
};
SideLane* side_lane();


// Launch predicate (the trailing `pred, run_if` pair of the hop entries, include/sgp_amd.h): the launch runs only if
// *flag == want when the kernel starts (device-side choice between the split-fp16 hop and the exact kernels, no host
// round trip); flag == nullptr: unconditional.
struct Predicate { const int* flag; int want; };

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

typedef float f32x4 __attribute__((ext_vector_type(4)));

}  // namespace sgp

#define SGP_REQUIRE(cond, ...) \
    do { if (!(cond)) return sgp::fail(SGP_EINVAL, __VA_ARGS__); } while (0)
