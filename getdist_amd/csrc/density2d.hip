#include "ctx.hpp"
void gd_fft_cache_destroy(gd_ctx*) {}
extern "C" {
int gd_density2d(gd_ctx* ctx, int32_t, int32_t, const void*, const double*, const double*, const double*, const int32_t*, const int32_t*, int32_t, int32_t, void*, int32_t*) { return gd_fail(ctx, GD_ERR_BADARG, "nyi"); }
}
