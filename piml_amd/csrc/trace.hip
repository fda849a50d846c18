// Live per-stage timing of an EAGER step with HIP events on the launch stream (bench.py: roofline.kernels[].us).  A step
// replayed from a captured graph cannot carry events on this stack, so the bench queues a few replays, then runs the same
// step eagerly with a trace open: every stage of the library (network.hip, relfeat.hip) marks itself behind its launch,
// and the interval between two consecutive marks is the GPU time of the later stage (the queue in front keeps the GPU
// busy, so the intervals hold no host-side gaps).  Not a reference interface: measurement plumbing like piml_timer_*.
#include <cstring>
#include <mutex>

#include "common.hpp"
#include "trace.hpp"
#include "../../include/piml_hip.h"

namespace piml {
namespace {
constexpr int kMaxMarks = 64;
std::mutex g_mu;
bool g_open = false;
int g_count = 0, g_created = 0;
hipEvent_t g_ev[kMaxMarks];
const char* g_name[kMaxMarks];
}  // namespace

void trace_mark(const char* name, hipStream_t s) {
    if (!g_open) return;
    std::lock_guard<std::mutex> lock(g_mu);
    if (!g_open || g_count >= g_created) return;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return;
    if (hipEventRecord(g_ev[g_count], s) != hipSuccess) return;
    g_name[g_count++] = name;
}
}  // namespace piml

using namespace piml;

PIML_API int piml_trace_begin(void) {
    std::lock_guard<std::mutex> lock(g_mu);
    while (g_created < kMaxMarks) {
        if (hipError_t e = hipEventCreate(&g_ev[g_created])) return e;
        ++g_created;
    }
    g_count = 0;
    g_open = true;
    return hipSuccess;
}

PIML_API int piml_trace_mark(const char* name, void* stream) {
    // `name` must outlive the trace (string literals / interned strings of the caller)
    trace_mark(name, as_stream(stream));
    return hipSuccess;
}

// Closes the trace.  names: '\n'-separated stage names of marks 1 .. n-1 (mark 0 is the start), us[i] = time between
// mark i and mark i + 1.  Returns the number of intervals, or a negative hipError_t.
PIML_API int piml_trace_end(char* names, int names_cap, float* us, int us_cap) {
    std::lock_guard<std::mutex> lock(g_mu);
    g_open = false;
    if (g_count < 2) return 0;
    if (hipError_t e = hipEventSynchronize(g_ev[g_count - 1])) return -(int)e;
    int n = 0, pos = 0;
    if (names && names_cap > 0) names[0] = 0;
    for (int i = 1; i < g_count && n < us_cap; ++i, ++n) {
        float ms = 0.f;
        if (hipError_t e = hipEventElapsedTime(&ms, g_ev[i - 1], g_ev[i])) return -(int)e;
        us[n] = ms * 1e3f;
        const int len = (int)strlen(g_name[i]);
        if (names && pos + len + 2 <= names_cap) {
            memcpy(names + pos, g_name[i], len);
            pos += len;
            names[pos++] = '\n';
            names[pos] = 0;
        }
    }
    return n;
}
