// Launch-side instrumentation shared by every kernel family of the library:
//
//  * NVT_PROF(name, algorithmic_bytes, stream) brackets the launches of one kernel family with
//    two HIP events ON THE STREAM THE KERNELS ARE LAUNCHED ON (nvt_prof_begin / nvt_prof_report,
//    include/nvt_hip.h).  Events come from a pool created once, so a profiled pass costs two
//    hipEventRecord per scope and nothing at all when profiling is off (one relaxed load).
//  * the same scope opens a roctx range named after the reference's @annotate of the step it
//    replaces (categorify.py:345,477,955,1054,1073,1149), so `rocprofv3 --marker-trace` output
//    can be attributed to operators / columns.  librocprofiler-sdk-roctx is dlopen'ed on first
//    use; absent library = no ranges, never an error.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nvt {

bool prof_enabled();
int prof_open(const char *name, uint64_t alg_bytes, hipStream_t stream);  // -> scope id or -1
void prof_close(int id, hipStream_t stream);
void roctx_push(const char *name);
void roctx_pop();

struct ProfScope {
  int id;
  hipStream_t s;
  ProfScope(const char *name, uint64_t bytes, hipStream_t stream) : id(-1), s(stream) {
    roctx_push(name);
    if (prof_enabled()) id = prof_open(name, bytes, stream);
  }
  ~ProfScope() {
    if (id >= 0) prof_close(id, s);
    roctx_pop();
  }
};

#define NVT_PROF_CAT2(a, b) a##b
#define NVT_PROF_CAT(a, b) NVT_PROF_CAT2(a, b)
#define NVT_PROF(name, bytes, stream) nvt::ProfScope NVT_PROF_CAT(_nvt_prof_, __LINE__)(name, bytes, stream)

}  // namespace nvt
