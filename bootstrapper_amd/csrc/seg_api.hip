// placeholder until the segmentation kernels land (same round)
#include "common.h"
extern "C" {
int bsmi_seg_create(int, const int64_t*, bsmi_seg**) { BSMI_FAIL(BSMI_ERR_STATE, "segmentation kernels not built yet"); }
int bsmi_seg_destroy(bsmi_seg*) { return BSMI_OK; }
int bsmi_ws_fragments_u8(bsmi_seg*, const uint8_t*, const int64_t*, int, int, uint64_t*, uint64_t*, void*) { BSMI_FAIL(BSMI_ERR_STATE, "segmentation kernels not built yet"); }
int bsmi_agglomerate_mean_u8(bsmi_seg*, const uint8_t*, const uint64_t*, const int64_t*, const float*, int, uint64_t*, void*) { BSMI_FAIL(BSMI_ERR_STATE, "segmentation kernels not built yet"); }
int bsmi_seg_status(bsmi_seg*, void*) { BSMI_FAIL(BSMI_ERR_STATE, "segmentation kernels not built yet"); }
}
