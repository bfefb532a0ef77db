// ABI bookkeeping, error reporting and HIP-event timing helpers.
#include "common.h"
#include <string.h>
#include <stdlib.h>
#include <mutex>

namespace sgp {
static thread_local char g_err[512] = "";
char* err_buf() { return g_err; }
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
long tune(const char* key, long dflt) {
    const char* e = getenv("SGP_TUNE");
    if (!e) return dflt;
    const size_t kl = strlen(key);
    for (const char* p = e; *p;) {
        while (*p == ',' || *p == ' ') ++p;
        if (!strncmp(p, key, kl) && p[kl] == '=') return strtol(p + kl + 1, nullptr, 10);
        while (*p && *p != ',') ++p;
    }
    return dflt;
}
unsigned* sync_slot(hipStream_t s) {
    constexpr int kSlots = 64, kSlotBytes = 256, kDevs = 64;
    static std::mutex mu;
    static unsigned char* ring[kDevs] = {};
    static unsigned next[kDevs] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kDevs) return nullptr;
    unsigned char* base;
    unsigned slot;
    {
        std::lock_guard<std::mutex> lock(mu);
        if (!ring[dev]) {
            void* p = nullptr;
            if (hipMalloc(&p, kSlots * kSlotBytes) != hipSuccess) return nullptr;
            ring[dev] = static_cast<unsigned char*>(p);
        }
        base = ring[dev];
        slot = next[dev]++ % kSlots;
    }
    unsigned char* p = base + slot * kSlotBytes;
    if (hipMemsetAsync(p, 0, kSlotBytes, s) != hipSuccess) return nullptr;
    return reinterpret_cast<unsigned*>(p);
}
SideLane* side_lane() {
    constexpr int kDevs = 64;
    thread_local SideLane lanes[kDevs] = {};
    thread_local bool made[kDevs] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kDevs) return nullptr;
    if (!made[dev]) {
        SideLane l{};
        if (hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking) != hipSuccess) return nullptr;
        if (hipEventCreateWithFlags(&l.forked, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&l.done, hipEventDisableTiming) != hipSuccess) return nullptr;
        lanes[dev] = l;
        made[dev] = true;
    }
    return &lanes[dev];
}
}  // namespace sgp

extern "C" {

int sgp_abi_version(void) { return SGP_ABI_VERSION; }
const char* sgp_last_error(void) { return sgp::err_buf(); }
const char* sgp_build_arch(void) { return "gfx950"; }
int64_t sgp_tune_value(const char* key, int64_t dflt) { return key ? (int64_t)sgp::tune(key, (long)dflt) : dflt; }

int sgp_event_create(void** ev) {
    SGP_REQUIRE(ev != nullptr, "sgp_event_create: null out pointer");
    hipEvent_t e;
    hipError_t r = hipEventCreate(&e);
    if (r != hipSuccess) return sgp::fail((int)r, "hipEventCreate: %s", hipGetErrorString(r));
    *ev = (void*)e;
    return 0;
}
int sgp_event_destroy(void* ev) {
    hipError_t r = hipEventDestroy((hipEvent_t)ev);
    if (r != hipSuccess) return sgp::fail((int)r, "hipEventDestroy: %s", hipGetErrorString(r));
    return 0;
}
int sgp_event_record(void* ev, sgp_stream_t stream) {
    hipError_t r = hipEventRecord((hipEvent_t)ev, (hipStream_t)stream);
    if (r != hipSuccess) return sgp::fail((int)r, "hipEventRecord: %s", hipGetErrorString(r));
    return 0;
}
int sgp_event_elapsed_ms(void* start, void* stop, float* ms) {
    SGP_REQUIRE(ms != nullptr, "sgp_event_elapsed_ms: null out pointer");
    hipError_t r = hipEventSynchronize((hipEvent_t)stop);
    if (r == hipSuccess) r = hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop);
    if (r != hipSuccess) return sgp::fail((int)r, "hipEventElapsedTime: %s", hipGetErrorString(r));
    return 0;
}

}  // extern "C"
