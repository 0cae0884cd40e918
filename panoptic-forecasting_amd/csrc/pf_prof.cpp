#include "pf_prof.h"

#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace pf {
namespace {
struct Rec {
    hipEvent_t a, b;
    int label;
    double flops, bytes;
};
struct Agg {
    std::string label;
    int launches = 0;
    double ms = 0, flops = 0, bytes = 0;
};
std::mutex g_mu;
bool g_on = false;
std::vector<Rec> g_recs;
std::vector<std::string> g_labels;
std::vector<Agg> g_agg;
std::string g_tag;

int label_id(const char *l) {
    for (size_t i = 0; i < g_labels.size(); ++i)
        if (g_labels[i] == l) return (int)i;
    g_labels.emplace_back(l);
    return (int)g_labels.size() - 1;
}
void clear_locked() {
    for (Rec &r : g_recs) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    g_recs.clear();
    g_labels.clear();
    g_agg.clear();
}
}  // namespace

bool prof_enabled() { return g_on; }

void prof_set_tag(const char *tag) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_tag = tag ? tag : "";
}

void prof_begin(hipStream_t s, const char *label, double flops, double bytes) {
    std::lock_guard<std::mutex> lk(g_mu);
    Rec r;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    r.label = g_tag.empty() ? label_id(label) : label_id((std::string(label) + " @" + g_tag).c_str());
    r.flops = flops;
    r.bytes = bytes;
    (void)hipEventRecord(r.a, s);
    g_recs.push_back(r);
}

void prof_end(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_recs.empty()) (void)hipEventRecord(g_recs.back().b, s);
}

}  // namespace pf

using namespace pf;

extern "C" int pf_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (on) clear_locked();
    g_on = on != 0;
    return PF_OK;
}

extern "C" int pf_profile_collect(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_agg.assign(g_labels.size(), Agg());
    for (size_t i = 0; i < g_labels.size(); ++i) g_agg[i].label = g_labels[i];
    for (Rec &r : g_recs) {
        if (hipEventSynchronize(r.b) != hipSuccess) return fail(PF_EHIP, "profile: event sync failed");
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) return fail(PF_EHIP, "profile: elapsed failed");
        Agg &a = g_agg[r.label];
        a.launches++;
        a.ms += ms;
        a.flops += r.flops;
        a.bytes += r.bytes;
    }
    return (int)g_agg.size();
}

extern "C" int pf_profile_get(int i, char *label, size_t cap, int *launches, double *total_ms, double *flops,
                              double *bytes) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (i < 0 || i >= (int)g_agg.size() || !label || !launches || !total_ms || !flops || !bytes)
        return fail(PF_EINVAL, "pf_profile_get: bad index/argument");
    strncpy(label, g_agg[i].label.c_str(), cap);
    if (cap) label[cap - 1] = 0;
    *launches = g_agg[i].launches;
    *total_ms = g_agg[i].ms;
    *flops = g_agg[i].flops;
    *bytes = g_agg[i].bytes;
    return PF_OK;
}
