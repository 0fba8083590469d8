// Entry points declared in include/somi_hip.h whose kernels are not written yet: they refuse loudly.
#include "common.h"
using namespace somi;
extern "C" size_t somi_loss_workspace_bytes(const somi_loss_desc *) { return 0; }
extern "C" int somi_yolo_loss_f32(const somi_loss_desc *, float *, void *, size_t, somi_stream_t) {
    set_error("somi_yolo_loss_f32: not implemented yet");
    return SOMI_ENOTIMPL;
}
extern "C" size_t somi_wbf_workspace_bytes(int) { return 0; }
extern "C" int somi_wbf_f32(const float *, const float *, const int32_t *, const int32_t *, int, int, const float *, float, float,
                            float *, float *, int32_t *, int32_t *, void *, size_t, somi_stream_t) {
    set_error("somi_wbf_f32: not implemented yet");
    return SOMI_ENOTIMPL;
}
