// marex_context.hip -- context, stream binding, per-kernel HIP-event timing (C ABI)
#include "marex_common.hip.h"

extern "C" int marex_abi_version(void) { return MAREX_ABI_VERSION; }

extern "C" int marex_create(int device, marex_ctx** out) {
    if (!out) return -1;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return -3;
    marex_ctx* c = new marex_ctx();
    c->device = device;
    *out = c;
    return 0;
}

extern "C" int marex_destroy(marex_ctx* ctx) {
    if (!ctx) return -1;
    drain_timers(ctx);
    if (ctx->shift_info) (void)hipFree(ctx->shift_info);
    if (ctx->thr_scratch) (void)hipFree(ctx->thr_scratch);
    if (ctx->detrend_scratch) (void)hipFree(ctx->detrend_scratch);
    if (ctx->morph_scratch) (void)hipFree(ctx->morph_scratch);
    delete ctx;
    return 0;
}

extern "C" const char* marex_last_error(marex_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

extern "C" int marex_set_stream(marex_ctx* ctx, void* s) {
    if (!ctx) return -1;
    ctx->stream = (hipStream_t)s;
    return 0;
}

extern "C" int marex_sync(marex_ctx* ctx) {
    if (!ctx) return -1;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    drain_timers(ctx);
    return 0;
}

extern "C" int marex_timing_enable(marex_ctx* ctx, int on) {
    if (!ctx) return -1;
    ctx->timing = on != 0;
    return 0;
}

extern "C" int marex_timing_reset(marex_ctx* ctx) {
    if (!ctx) return -1;
    drain_timers(ctx);
    memset(ctx->total_ms, 0, sizeof ctx->total_ms);
    memset(ctx->launches, 0, sizeof ctx->launches);
    return 0;
}

extern "C" int marex_timing_get(marex_ctx* ctx, int kid, double* total_ms, int64_t* launches) {
    if (!ctx || kid < 0 || kid >= MAREX_K_COUNT) return -1;
    drain_timers(ctx);
    if (total_ms) *total_ms = ctx->total_ms[kid];
    if (launches) *launches = ctx->launches[kid];
    return 0;
}
