#include "ctx.hpp"
extern "C" {
int gd_kopt2d(gd_ctx* ctx, int32_t, int32_t, const void*, const double*, const int32_t*, const double*, double*) { return gd_fail(ctx, GD_ERR_BADARG, "nyi"); }
}
