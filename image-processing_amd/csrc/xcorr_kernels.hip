#include "mis_common.h"
void mis_xcorr_release(mis_ctx*) {}
extern "C" {
int mis_xcorr_tiles(mis_ctx* ctx, const mis_rect*, const mis_rect*, int, int, int, mis_offset*) { return mis_fail(ctx, MIS_ERR_UNSUPPORTED, "not built yet"); }
int mis_xcorr_tiles_device(mis_ctx* ctx, const mis_rect*, const mis_rect*, int, int, int, mis_offset*) { return mis_fail(ctx, MIS_ERR_UNSUPPORTED, "not built yet"); }
int mis_xcorr_tile_surfaces(mis_ctx* ctx, mis_rect, mis_rect, int, int, float*, float*) { return mis_fail(ctx, MIS_ERR_UNSUPPORTED, "not built yet"); }
int mis_fft2d_forward(mis_ctx* ctx, const void*, int, int, int, int, float*) { return mis_fail(ctx, MIS_ERR_UNSUPPORTED, "not built yet"); }
int mis_xcorr_surface_u8(mis_ctx* ctx, uint8_t*) { return mis_fail(ctx, MIS_ERR_UNSUPPORTED, "not built yet"); }
}
