// Opt-in per-launch timing (hipEvents on the launch stream) used by bench.py for the roofline line.
// Off by default; when off a scope costs one branch.  Not usable during graph capture.
#pragma once
#include "pf_common.h"

namespace pf {

bool prof_enabled();
void prof_begin(hipStream_t s, const char *label, double flops, double bytes);
void prof_end(hipStream_t s);
// optional per-op tag (set by the executor when PF_PROFILE_OPS=1): labels become "<kernel> @<tag>"
void prof_set_tag(const char *tag);

struct ProfScope {
    hipStream_t s;
    bool on;
    ProfScope(hipStream_t s_, const char *label, double flops, double bytes) : s(s_), on(prof_enabled()) {
        if (on) prof_begin(s, label, flops, bytes);
    }
    ~ProfScope() {
        if (on) prof_end(s);
    }
};

}  // namespace pf
