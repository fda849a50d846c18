// Stage tracing (measurement plumbing, off by default): when a trace is open, every launch stage of the library records
// a HIP event behind its launch; piml_trace_end returns the time between consecutive events.  Private to libpiml_hip.so.
#pragma once
#include <hip/hip_runtime.h>

namespace piml {
// record "the stage `name` has been enqueued on s" (no-op unless a trace is open; never inside a stream capture)
void trace_mark(const char* name, hipStream_t s);
}  // namespace piml
