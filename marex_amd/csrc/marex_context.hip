// marex_context.hip -- context, stream binding, per-kernel HIP-event timing (C ABI)
#include "marex_common.hip.h"

extern "C" int marex_abi_version(void) { return MAREX_ABI_VERSION; }

extern "C" int marex_workspace_bytes(const marex_workspace_cfg* cfg, size_t* bytes) {
    if (!cfg || !bytes || cfg->T <= 0 || cfg->T_out <= 0 || cfg->T_out > cfg->T || cfg->C <= 0) return -1;
    const size_t C = (size_t)cfg->C, To = (size_t)cfg->T_out;
    for (int i = 0; i < MAREX_WS_COUNT; ++i) bytes[i] = 0;
    bytes[MAREX_WS_ANOMALY] = To * C * sizeof(float);
    bytes[MAREX_WS_EXTREME] = To * C;
    bytes[MAREX_WS_THRESHOLDS] = (size_t)MAREX_NDOY * C * sizeof(float);
    if (cfg->list_rows > 0) {
        const int nper = marex_tail_lists(cfg->max_bucket, cfg->list_rows);
        if (nper < 1) return -1;
        const size_t nch = cfg->list_rows <= 16 ? 2 : 4;
        bytes[MAREX_WS_LISTS] = (size_t)MAREX_NDOY * (size_t)nper * nch * C * 16;
        bytes[MAREX_WS_AUX] = (size_t)MAREX_NDOY * C * sizeof(uint32_t);
    } else {
        bytes[MAREX_WS_BINS] = ((C + 15) / 16) * To * 16 * sizeof(uint16_t);
    }
    bytes[MAREX_WS_PER_CELL] = C * (sizeof(uint8_t) + sizeof(int32_t));
    bytes[MAREX_WS_TOTAL] = bytes[MAREX_WS_ANOMALY] + bytes[MAREX_WS_EXTREME] + 2 * bytes[MAREX_WS_THRESHOLDS] + bytes[MAREX_WS_LISTS] +
                            bytes[MAREX_WS_AUX] + bytes[MAREX_WS_BINS] + bytes[MAREX_WS_PER_CELL];
    return 0;
}

extern "C" int marex_create(int device, marex_ctx** out) {
    if (!out) return -1;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return -3;
    marex_ctx* c = new marex_ctx();
    c->device = device;
    // options: MAREX_<NAME>=<int> present in the environment NOW seed the table (experiments, A/B runs); later changes
    // go through marex_set_option
    extern char** environ;
    for (char** e = environ; e && *e; ++e) {
        if (strncmp(*e, "MAREX_", 6) != 0) continue;
        const char* eq = strchr(*e, '=');
        if (!eq || eq == *e + 6) continue;
        char* end = nullptr;
        const long v = strtol(eq + 1, &end, 10);
        if (end == eq + 1 || *end != '\0') continue;  // not an integer (paths, seed ranges ...)
        c->opts[std::string(*e + 6, (size_t)(eq - (*e + 6)))] = (int)v;
    }
    *out = c;
    return 0;
}

extern "C" int marex_set_option(marex_ctx* ctx, const char* name, int value) {
    if (!ctx || !name || !*name) return -1;
    ctx->opts[name] = value;
    return 0;
}

extern "C" int marex_clear_option(marex_ctx* ctx, const char* name) {
    if (!ctx) return -1;
    if (!name)
        ctx->opts.clear();
    else
        ctx->opts.erase(name);
    return 0;
}

extern "C" int marex_debug_counters(marex_ctx* ctx, uint64_t* out, int reset) {
    if (!ctx || !out) return -1;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    unsigned long long* d = ctx_debug_counters(ctx);
    if (!d) return fail(ctx, -2, "marex_debug_counters: allocation failed");
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(out, d, MAREX_DBG_COUNTERS * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (reset) HIP_TRY(ctx, hipMemsetAsync(d, 0, MAREX_DBG_COUNTERS * sizeof(unsigned long long), ctx->stream));
    return 0;
}

extern "C" int marex_destroy(marex_ctx* ctx) {
    if (!ctx) return -1;
    drain_timers(ctx);
    if (ctx->shift_info) (void)hipFree(ctx->shift_info);
    if (ctx->shift_plan) (void)hipFree(ctx->shift_plan);
    if (ctx->shift_lplan) (void)hipFree(ctx->shift_lplan);
    if (ctx->thr_scratch) (void)hipFree(ctx->thr_scratch);
    if (ctx->detrend_scratch) (void)hipFree(ctx->detrend_scratch);
    if (ctx->morph_scratch) (void)hipFree(ctx->morph_scratch);
    if (ctx->dbg_counters) (void)hipFree(ctx->dbg_counters);
    if (ctx->row_off) (void)hipFree(ctx->row_off);
    if (ctx->row_off_mask) (void)hipFree(ctx->row_off_mask);
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
