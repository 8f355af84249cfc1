// temporary stubs (replaced by plan_controlnet.cpp / plan_adapter.cpp)
#include "ops.h"
extern "C" {
int ctrl_controlnet_param_count(const ctrl_controlnet_config*) { return 0; }
int ctrl_controlnet_param_spec(const ctrl_controlnet_config*, int, char*, int, int64_t*, int*) { CTRL_FAIL("not implemented"); }
int ctrl_controlnet_create(const ctrl_controlnet_config*, const ctrl_tensor_ref*, int, void*, ctrl_controlnet**) { CTRL_FAIL("not implemented"); }
void ctrl_controlnet_destroy(ctrl_controlnet*) {}
int ctrl_controlnet_forward(ctrl_controlnet*, const void*, int, int, int, int, const float*, int, const void*, int, int, const void*, int, float, int, void* const*, int, void*) { CTRL_FAIL("not implemented"); }
int ctrl_adapter_param_count(const ctrl_adapter_config*) { return 0; }
int ctrl_adapter_param_spec(const ctrl_adapter_config*, int, char*, int, int64_t*, int*) { CTRL_FAIL("not implemented"); }
int ctrl_adapter_create(const ctrl_adapter_config*, const ctrl_tensor_ref*, int, void*, ctrl_adapter**) { CTRL_FAIL("not implemented"); }
void ctrl_adapter_destroy(ctrl_adapter*) {}
int ctrl_adapter_forward(ctrl_adapter*, const void* const*, int, int, int, int, int, const float*, int, const void*, int, int, int, void* const*, int, void*) { CTRL_FAIL("not implemented"); }
}
