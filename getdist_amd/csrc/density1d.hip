#include "ctx.hpp"
extern "C" {
int gd_dct1d(gd_ctx* ctx, int32_t, int32_t, const double*, double*) { return gd_fail(ctx, GD_ERR_BADARG, "nyi"); }
int gd_density1d(gd_ctx* ctx, int32_t, int32_t, const double*, const double*, const int32_t*, const int32_t*, int32_t, int32_t, double*, int32_t*) { return gd_fail(ctx, GD_ERR_BADARG, "nyi"); }
}
