// placeholder until the HarDNet plan lands (next commit)
#include "pf_common.h"
extern "C" int pf_hardnet_plan_create(const void *, size_t, int, int, pf_plan **) { return pf::fail(PF_EUNSUPPORTED, "not built yet"); }
extern "C" void pf_hardnet_plan_destroy(pf_plan *) {}
extern "C" int pf_hardnet_workspace(const pf_plan *, int, int, int, size_t *) { return pf::fail(PF_EUNSUPPORTED, "not built yet"); }
extern "C" int pf_bg_forward(const pf_plan *, const void *, int, const float *, const uint8_t *, float, float, int, float, float, int, int, int, int, int, int, void *, int, float *, float *, void *, size_t, void *) { return pf::fail(PF_EUNSUPPORTED, "not built yet"); }
extern "C" int pf_hardnet_forward_dense(const pf_plan *, const float *, int, int, int, int, int, void *, int, float *, float *, void *, size_t, void *) { return pf::fail(PF_EUNSUPPORTED, "not built yet"); }
extern "C" int pf_hardnet_tensor_view(const pf_plan *, const char *, int, int, int, size_t *, int *, int *, int *) { return pf::fail(PF_EUNSUPPORTED, "not built yet"); }
extern "C" int pf_hardnet_flops(const pf_plan *, int, int, double *) { return pf::fail(PF_EUNSUPPORTED, "not built yet"); }
