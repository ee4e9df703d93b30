/*
 * marex_hip.h -- C ABI of the MI355X (gfx950) hot path behind marEx-style preprocess_data().
 *
 * The reference (wienkers/marEx) is pure Python and has NO plugin / FFI interface: its boundary for
 * this path is the Python call marEx.preprocess_data() (marEx/detect.py:287-313).  This header is
 * therefore the build-defined drop-in boundary: one entry point per numerical stage
 * of that call, each citing the reference lines it replaces.  The Python host side
 * (marex_amd/detect.py) binds it with ctypes; INTEGRATION.md shows the stub a marEx maintainer
 * would add.
 *
 * Conventions
 *  - every function returns 0 on success, <0 on error; marex_last_error(ctx) gives the text.
 *  - no exceptions / longjmp cross the ABI; the library owns only the opaque context and the device scratch buffers
 *    inside it (freed by marex_destroy); every data buffer belongs to the caller.
 *  - ALL data pointers are DEVICE pointers (small tables included), except in the entry points suffixed "_h" (the chunk
 *    codec of the Zarr stores), which take HOST pointers.
 *  - work is enqueued on the context's HIP stream (marex_set_stream); marex_sync waits for it.
 *  - field layout is the reference's C-order (time, cells): element (t, c) at [t*C + c].
 *  - one context per (device, host thread).
 */
#ifndef MAREX_HIP_H
#define MAREX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MAREX_ABI_VERSION 2
#define MAREX_NDOY 366

typedef struct marex_ctx marex_ctx;

/* statistics of the threshold stage, device resident (detect.py:2707-2732 warnings) */
typedef struct marex_thr_stats {
    uint32_t min_key;   /* order-preserving key of the smallest un-clamped threshold (0xFFFFFFFF if none) */
    uint32_t max_key;   /* order-preserving key of the largest threshold (0 if none)                      */
    uint32_t n_too_low; /* thresholds < lower_bound (they are clamped to it)                              */
    uint32_t n_too_high;/* thresholds > upper_bound                                                       */
    uint32_t n_unresolved; /* outputs a threshold kernel gave up on (written as NaN): an internal error, the host
                              side raises ProcessingError when it is not 0 (never observed; marex_tails.hip, straggler passes) */
    uint32_t reserved[3];
} marex_thr_stats;  /* 32 bytes; callers zero it except min_key = 0xFFFFFFFF */

/* kernel ids for marex_timing_get */
enum {
    MAREX_K_SYNTH = 0,
    MAREX_K_SHIFTING = 1,     /* smoothing + rolling climatology + anomaly + binning + validation */
    MAREX_K_THRESHOLDS = 2,   /* pooled day-of-year histogram quantile                             */
    MAREX_K_MASK = 3,         /* anomaly >= threshold                                              */
    MAREX_K_TRANSPOSE = 4,
    MAREX_K_FIXED = 5,
    MAREX_K_DETREND = 6,
    MAREX_K_EXACT = 7,
    MAREX_K_GLOBAL = 8,
    MAREX_K_STDNORM = 9,      /* day-of-year std, 30-day wrapped rolling RMS, anomaly / STD                 */
    MAREX_K_MORPH = 10,       /* tracker pre-processing: spatial closing + opening, temporal closing       */
    MAREX_K_TAILS = 11,       /* tail extraction (upper end of every dayofyear histogram)                   */
    MAREX_K_COUNT = 12
};

int marex_abi_version(void);

/* Sizes of the caller-owned device buffers of one spatial block (the caller owns every data buffer, the library
 * only its handle and scratch).  bytes[MAREX_WS_COUNT] in the order of the enum; MAREX_WS_TOTAL = what one pass of
 * shifting_baseline / fixed_baseline + hobday_extreme needs beside the input (anomalies, mask bytes, thresholds in both layouts,
 * key lists + aux OR the bin matrix, per-cell mask and counts).  Returns -1 for a null pointer or an impossible shape. */
typedef struct marex_workspace_cfg {
    int64_t T;        /* input timesteps                                                              */
    int64_t T_out;    /* kept timesteps (T minus the first window_year_baseline years for shifting_baseline) */
    int64_t C;        /* cells of the block, overlap rows included                                      */
    int max_bucket;   /* rows of the largest dayofyear bucket                                           */
    int list_rows;    /* rows per key list: 15 (lists emitted by the anomaly kernel), 32 (extraction kernel), 0: bin-matrix path */
} marex_workspace_cfg;
enum {
    MAREX_WS_ANOMALY = 0,   /* float [T_out][C]                         */
    MAREX_WS_EXTREME = 1,   /* uint8 [T_out][C]                         */
    MAREX_WS_THRESHOLDS = 2,/* float [366][C], dayofyear-major AND cell-major: two buffers of this size */
    MAREX_WS_LISTS = 3,     /* key lists [366][NPER][nch][C] x 16 bytes (0 on the bin-matrix path)     */
    MAREX_WS_AUX = 4,       /* uint32 [366][C] (0 on the bin-matrix path)                              */
    MAREX_WS_BINS = 5,      /* uint16 [ceil(C/16)][T_out][16] (0 on the list path)                     */
    MAREX_WS_PER_CELL = 6,  /* uint8 mask [C] + int32 invalid_count [C]                                */
    MAREX_WS_TOTAL = 7,
    MAREX_WS_COUNT = 8
};
int marex_workspace_bytes(const marex_workspace_cfg* cfg, size_t* bytes);
/* A context owns small device scratch buffers that its launches share: use one context per (device, host thread,
 * stream); launches through one context are stream-ordered and must not overlap on different streams. */
int marex_create(int device, marex_ctx** out);
int marex_destroy(marex_ctx* ctx);
const char* marex_last_error(marex_ctx* ctx);
int marex_set_stream(marex_ctx* ctx, void* hip_stream);
int marex_sync(marex_ctx* ctx);

/* Tuning / diagnostic options (tile shapes, forced kernel variants ...; DESIGN.md lists them).  Options live in the context:
 * environment variables MAREX_<NAME>=<int> that exist when the context is created seed the table once, afterwards only
 * these calls change it -- nothing reads the environment at launch time.  name == NULL clears every option. */
int marex_set_option(marex_ctx* ctx, const char* name, int value);
int marex_clear_option(marex_ctx* ctx, const char* name);
/* Event counters of the tail kernels (host uint64[8], synchronises the stream): [0] column rebuilds, [1] unused, [2] extra passes for stragglers, [3] days walked (per tile and block),
 * [4] (4-cell, dayofyear) groups the mask kernel decided on the anomalies, [5..7] phase timers of -DMAREX_STAMPS builds.
 * reset != 0 zeroes them afterwards. */
int marex_debug_counters(marex_ctx* ctx, uint64_t* out8, int reset);

/* per-kernel device timing with HIP events on the context's stream (used by bench.py) */
int marex_timing_enable(marex_ctx* ctx, int on);
int marex_timing_reset(marex_ctx* ctx);
int marex_timing_get(marex_ctx* ctx, int kernel_id, double* total_ms, int64_t* launches);

/* Synthetic SST field of marex_amd/synth.py, bit-identical to the NumPy version (test/bench utility). */
int marex_synth_sst_f32(marex_ctx* ctx, const float* mean, const float* amp, const uint8_t* hemi,
                        const uint8_t* land, const float* seas /*[T,2]*/, const float* trend /*[T]*/,
                        uint64_t seed, int64_t cell_base, int64_t T, int64_t C, float* x /*[T,C]*/);

/*
 * Shifting-baseline anomaly, fused with validation counting and histogram binning.
 * Replaces: _validate_data_values (detect.py:205-279), smoothed_rolling_climatology (1691-1816),
 * rolling_climatology (1511-1688), _compute_anomaly_shifting_baseline (1819-1850), the year trim
 * (615-641) and the np.digitize step of _compute_histogram_quantile_2d (2622-2631).
 *
 *  year_plan[n_cal_years*366][4]  int32 {timestep, output row, bin-matrix row, 0} of (calendar-year index,
 *                           dayofyear-1); -1 where absent / not an output (rows of the first W years are
 *                           trimmed: detect.py:638-641).  16-byte aligned.  Built on the host from dt.year /
 *                           dt.dayofyear (marex_amd/calendar.py); bin-matrix rows are the kept timesteps sorted
 *                           by (dayofyear, time).
 *  write_clim               0: out = x - clim (anomaly); 1: out = clim (rolling_climatology API)
 *  edges[nb+1], bins        bin table and output bin matrix, uint16, T_out rows (both may be NULL: no binning).
 *                           BIN MATRIX LAYOUT (all entry points): blocks of 16 consecutive cells; element
 *                           (row r, cell c) at ((c >> 4) * T_out + r) * 16 + (c & 15); ceil(C/16)*T_out*16 elements;
 *                           rows are the kept timesteps sorted by (dayofyear, time) (rowb_index)
 *  mask[C]                  isfinite(x[0, c])                     (may be NULL)
 *  invalid_count[C]         number of non-finite x[t, c] over t; must be zeroed by the caller (may be NULL)
 */
int marex_shifting_baseline_f32(marex_ctx* ctx, const float* x, int64_t T, int64_t C,
                                const int32_t* year_plan, int n_cal_years, int W, int S,
                                int write_clim, const float* edges, int nb, int64_t T_out, float* out,
                                uint16_t* bins, uint8_t* mask, int32_t* invalid_count);

/*
 * Day-of-year thresholds from pooled histograms (approximate percentile method).
 * Replaces: the flox 2-D count + spatial pooling + per-cell _rolling_histogram_quantile +
 * NaN-masking + clamp of _compute_histogram_quantile_2d (detect.py:2638-2732, 2465-2559).
 *
 *  bins               uint16 bin matrix in the blocked layout described at marex_shifting_baseline_f32
 *  doy_start[367]     rows doy_start[d-1]..doy_start[d]-1 hold dayofyear d
 *  max_bucket         largest number of rows of one dayofyear (host knows doy_start); lets the kernel use
 *                     16-bit counters when max_bucket*wd*ws*ws <= 65535.  0 = unknown (32-bit counters)
 *  ny, nx             grid (cells = ny*nx, lon fastest); ny == 0: unstructured, no pooling (ws must be <= 1)
 *  first_anom[C]      first kept anomaly row; NaN there => threshold NaN (detect.py:2704)
 *  centres[nb]        float32 bin centres, centres[0] == 0
 *  q, wd, ws          quantile in (0,1], odd day window (3..365), odd spatial window (1 = none)
 *  lower/upper_bound  edges[3] / edges[-2]
 *  row0, row1         gridded only: thresholds are produced for grid rows row0 <= j < row1 (the rows a
 *                     latitude shard owns; its overlap rows only feed the pooling).  Other rows of
 *                     thr_doy_major are left untouched.  Pass 0, ny for everything.
 *  thr_doy_major      out [366, C]   (dayofyear-major working layout used by marex_mask_ge_doy_f32)
 *  stats              device struct, must be initialised {0xFFFFFFFF, 0, 0, 0} by the caller
 */
int marex_hobday_thresholds_f32(marex_ctx* ctx, const uint16_t* bins, int64_t T_out, int64_t C, int ny,
                                int nx, const int32_t* doy_start, int max_bucket, const float* first_anom,
                                const float* centres, int nb, double q, int wd, int ws,
                                float lower_bound, float upper_bound, int row0, int row1,
                                float* thr_doy_major, marex_thr_stats* stats);

/*
 * extreme[t, c] = anom[t, c] >= thr[doy(t), c]   (detect.py:2003-2004) and the count of True (833-835).
 *  doy_rows[T_out]   output rows sorted by (dayofyear, time); doy_start as above
 *  c0, c1            only cells c0 <= c < c1 are compared, written and counted (owned cells of a shard)
 *  n_true            device counter, must be zeroed by the caller (may be NULL)
 */
int marex_mask_ge_doy_f32(marex_ctx* ctx, const float* anom, const float* thr_doy_major,
                          const int32_t* doy_start, const int32_t* doy_rows, int64_t T_out, int64_t C,
                          int64_t c0, int64_t c1, uint8_t* extreme, unsigned long long* n_true);

/* The same mask, read from the bin matrix marex_shifting_baseline_f32 / marex_digitize_f32 produced for these anomalies
 * (`bins`, `edges[0..nb]` as passed there): a sample whose bin lies above / below the bin of its threshold is decided
 * without its value; only samples in the threshold's own bin or in the overflow bin nb are compared as numbers.  Same
 * result bit for bit, 2 instead of 4 bytes read per sample.  Falls back to marex_mask_ge_doy_f32 for shapes its 4-cell
 * kernel does not cover (C, c0, c1 not multiples of 4; unaligned pointers). */
int marex_mask_ge_doy_bins_f32(marex_ctx* ctx, const float* anom, const uint16_t* bins, const float* edges, int nb,
                               const float* thr_doy_major, const int32_t* doy_start, const int32_t* doy_rows, int64_t T_out,
                               int64_t C, int64_t c0, int64_t c1, uint8_t* extreme, unsigned long long* n_true);

/*
 * TAILS: the default representation of the day-of-year histograms of the approximate Hobday method for long series
 * (marEx/detect.py:2622-2648: np.digitize + the flox 2-D count by (dayofyear, bin)).  Only the upper end of a histogram
 * decides a high quantile, so instead of a dense (366 x nb) count per cell the device keeps the samples of every
 * (dayofyear, cell) bucket as 16-bit keys in short lists that are sorted descending; consumers read the top of each list
 * and stop at the first key they do not need:
 *   key = ((bin + 1) << 7) | pos    bin = np.digitize(anom, edges) - 1 (< nb <= 511), pos = index of the sample inside the
 *                                   bucket (output row doy_rows[doy_start[d] + pos], < 128); 0 = empty slot; samples the
 *                                   reference's histogram drops (NaN, >= edges[nb]) have no key
 *   lists  uint16, [366][NPER][NCH][C][8]  chunk j (0: the 8 largest ...) of list p of bucket (d, c).  A bucket's keys are
 *                                   partitioned over NPER = marex_tail_lists(max_bucket, list_rows) = ceil(max_bucket /
 *                                   list_rows) lists of <= list_rows keys, each sorted descending, in NCH = 2 (list_rows
 *                                   <= 16) or 4 (<= 32) chunks of 8 keys.  list_rows = 32: marex_tail_extract_f32 on any
 *                                   anomaly field; list_rows = 15: what marex_shifting_baseline_tails_f32 emits itself
 *   aux    uint32, [366][C]         bits 0..9: number of keys of the bucket (= the samples the reference counts);
 *                                   bit 15: the bucket holds a non-NaN value >= edges[nb] (beyond the table: no key, but an
 *                                   extreme of every finite threshold); bits 16..22 / 23..29: position of the first /
 *                                   second such sample, bit 30: the second exists, bit 31: more than two
 * Rows are grouped by dayofyear through doy_start / doy_rows; max_bucket = rows of the largest dayofyear (<= 128).
 */
int marex_tail_lists(int max_bucket, int list_rows);
int marex_tail_extract_f32(marex_ctx* ctx, const float* anom, int64_t T_out, int64_t C, const int32_t* doy_start,
                           const int32_t* doy_rows, int max_bucket, const float* edges, int nb, int list_rows, void* lists,
                           uint32_t* aux);

/* The shifting-baseline anomaly stage of marex_shifting_baseline_f32 emitting TAILS (list_rows = 15) instead of the bin
 * matrix: the kernel sorts the keys of 15 output years at a time and writes them as one list per dayofyear; dayofyears its
 * fast path does not take (irregular calendars, smoothing / baseline windows without a fast instance) get their lists from
 * the anomalies afterwards -- the lists are complete either way.  -4: nb > 511, max_bucket > 90 or C > 2^24. */
int marex_shifting_baseline_tails_f32(marex_ctx* ctx, const float* x, int64_t T, int64_t C, const int32_t* year_plan,
                                      int n_cal_years, int W, int S, const float* edges, int nb, int64_t T_out, float* out,
                                      uint8_t* mask, int32_t* invalid_count, const int32_t* doy_start, const int32_t* doy_rows,
                                      int max_bucket, void* lists, uint32_t* aux);

/* Day-of-year thresholds from tails: same result as marex_hobday_thresholds_f32 (detect.py:2638-2732, 2465-2559: pooled
 * counts, count-interpolated quantile, NaN where the first kept anomaly is NaN, clamp and warning statistics), arguments as
 * there plus the tails and the anomaly field they belong to (`anom`, [T_out, C]: its row 0 is first_anom).
 * Needs nb <= 511, max_bucket <= 128 in at most 6 lists, ws <= 7, max_bucket*wd*ws*ws <= 65535 and C <= 2^24 (else -4: use
 * the bin-matrix entry point). */
int marex_hobday_thresholds_tails_f32(marex_ctx* ctx, const void* lists, const uint32_t* aux, int list_rows, const float* anom,
                                      int64_t T_out, int64_t C, int ny, int nx, int max_bucket, const float* centres, int nb,
                                      double q, int wd, int ws, float lower_bound, float upper_bound, int row0, int row1,
                                      float* thr_doy_major, marex_thr_stats* stats);

/* The extreme mask from tails: same result as marex_mask_ge_doy_f32 (detect.py:2003-2004, 833-835) without reading the
 * anomalies, except for samples in the threshold's own bin and for buckets holding more than two values beyond the edge table
 * (up to two are placed from the positions in aux). */
int marex_mask_ge_doy_tails_f32(marex_ctx* ctx, const void* lists, const uint32_t* aux, int list_rows, int max_bucket,
                                const float* anom, const float* edges, int nb, const float* thr_doy_major,
                                const int32_t* doy_start, const int32_t* doy_rows, int64_t T_out, int64_t C, int64_t c0,
                                int64_t c1, uint8_t* extreme, unsigned long long* n_true);

/*
 * Fixed-baseline anomaly (detect.py:2299-2397): clim[d, c] = float32 nanmean of x over the timesteps with
 * dayofyear d (only those with use_row[t] != 0 when use_row is given: reference_period, 2334-2361),
 * out[t, c] = x[t, c] - clim[doy(t), c] for ALL timesteps.  doy_start / doy_rows describe all T rows
 * (calendar built without trim).  bins / mask / invalid_count as in marex_shifting_baseline_f32
 * (bins rows are the dayofyear-sorted rows; invalid_count must be zeroed by the caller).
 */
int marex_fixed_baseline_f32(marex_ctx* ctx, const float* x, int64_t T, int64_t C, const int32_t* doy_start,
                             const int32_t* doy_rows, const uint8_t* use_row, const float* edges, int nb,
                             float* out, uint16_t* bins, uint8_t* mask, int32_t* invalid_count);
/* The same stage with an optional per-cell value `sub[C]` taken off every sample on load (one float32 subtraction, as
 * detect.py:2222-2224 does for the whole field): folds the last pass of a force_zero_mean detrend into this one.
 * `max_bucket` = rows of the largest dayofyear bucket (0 = unknown): up to 128 rows without a bin matrix the field is read
 * once (buckets held in registers). */
int marex_fixed_baseline_sub_f32(marex_ctx* ctx, const float* x, const float* sub, int max_bucket, int64_t T, int64_t C,
                                 const int32_t* doy_start, const int32_t* doy_rows, const uint8_t* use_row,
                                 const float* edges, int nb, float* out, uint16_t* bins, uint8_t* mask,
                                 int32_t* invalid_count);

/* bins[rowb_index[t], c] = np.digitize(anom[t, c], edges) - 1 for rows with rowb_index[t] >= 0 (detect.py:2622-2631) */
int marex_digitize_f32(marex_ctx* ctx, const float* anom, int64_t T, int64_t C, const int32_t* rowb_index,
                       const float* edges, int nb, int64_t T_out, uint16_t* bins);

/*
 * Polynomial / harmonic detrend (detect.py:2143-2224).  pmodel[T, n_coef] = pinv(model) and
 * model_t[T, n_coef] = model^T, float64, computed on the host (marex_amd/calendar.py:detrend_model).
 * out = x - fl32(model^T (pmodel^T x)), minus its time mean when force_zero_mean.  mask / invalid_count
 * are written (not accumulated) when given.  The two reductions over time are float64 partial sums over blocks of
 * 1024 consecutive timesteps (ascending t) combined in ascending block order -- the order oracle.detrend_anomaly
 * fixes (the reference's BLAS order is unspecified).  Scratch lives in the context.
 */
int marex_detrend_f32(marex_ctx* ctx, const float* x, int64_t T, int64_t C, const double* pmodel,
                      const double* model_t, int n_coef, int force_zero_mean, float* out, uint8_t* mask,
                      int32_t* invalid_count);
/* detect.py:2143-2224 with force_zero_mean, the mean NOT yet subtracted: `out` = residuals, `mean[C]` = fl32(sum / T)
 * (the value marex_detrend_f32 would subtract); for callers that subtract while reading (marex_fixed_baseline_sub_f32). */
int marex_detrend_deferred_mean_f32(marex_ctx* ctx, const float* x, int64_t T, int64_t C, const double* pmodel,
                                    const double* model_t, int n_coef, float* out, float* mean, uint8_t* mask,
                                    int32_t* invalid_count);
/* detect.py:2400-2462 (detrend_fixed_baseline) as one chain: fit, residual mean, and the daily climatology stage recomputing
 * the residuals from x while it reads (3 reads and 1 write of the field; the residual field is never materialised).
 * n_coef <= 5 and max_bucket <= 128, else -4 (run marex_detrend_deferred_mean_f32 + marex_fixed_baseline_sub_f32 instead).
 * model_sorted[r][k] = model_t[doy_rows[r]][k]: the model rows in dayofyear-sorted row order (contiguous per bucket).
 * mask / invalid_count: of the RAW field, as marex_detrend_f32 reports them. */
int marex_detrend_fixed_baseline_f32(marex_ctx* ctx, const float* x, int64_t T, int64_t C, const double* pmodel,
                                     const double* model_t, const double* model_sorted, int n_coef, int force_zero_mean,
                                     const int32_t* doy_start,
                                     const int32_t* doy_rows, const uint8_t* use_row, int max_bucket, float* out,
                                     uint8_t* mask, int32_t* invalid_count);
/* The two fixed-baseline stages above (detect.py:2299-2397, 2400-2462) that also leave the sorted key lists of their own output
 * -- exactly what marex_tail_extract_f32(out, list_rows = 32) would write (TAILS, below): `lists` [366][NPER][4][C] x 16 bytes with
 * NPER = marex_tail_lists(max_bucket, 32), `aux` [366][C]; the bucket is in registers when its anomalies are formed, so the
 * threshold stage needs no extraction pass (a read of the whole anomaly field: 13 ms per 100-yr band, round 3).  `edges[0..nb]`
 * as for marex_tail_extract_f32; nb <= 511, max_bucket <= 128 (-4 otherwise: run the plain entry point + marex_tail_extract_f32). */
int marex_fixed_baseline_tails_f32(marex_ctx* ctx, const float* x, const float* sub, int max_bucket, int64_t T, int64_t C,
                                   const int32_t* doy_start, const int32_t* doy_rows, const uint8_t* use_row,
                                   const float* edges, int nb, float* out, uint8_t* mask, int32_t* invalid_count,
                                   void* lists, uint32_t* aux);
int marex_detrend_fixed_baseline_tails_f32(marex_ctx* ctx, const float* x, int64_t T, int64_t C, const double* pmodel,
                                           const double* model_t, const double* model_sorted, int n_coef,
                                           int force_zero_mean, const int32_t* doy_start, const int32_t* doy_rows,
                                           const uint8_t* use_row, int max_bucket, const float* edges, int nb, float* out,
                                           uint8_t* mask, int32_t* invalid_count, void* lists, uint32_t* aux);

/*
 * Exact Hobday percentile (detect.py:1921-1956): thr[d, c] = np.nanpercentile(anom[doy in window(d), c], p),
 * float32, NumPy 2.x "linear" rule, no spatial pooling; output layout [366, C] (= the reference's
 * (dayofyear, space) order for this method).  q32 = float32(p) / float32(100) computed by the caller with
 * NumPy; max_window_rows = largest number of rows in any wd-day window; *overflow (device int, zeroed by
 * the caller) counts cells whose selection buffer was too small (must stay 0).
 */
int marex_hobday_exact_f32(marex_ctx* ctx, const float* anom, int64_t T_out, int64_t C, const int32_t* doy_start,
                           const int32_t* doy_rows, int max_window_rows, float q32, double q, int wd,
                           float* thr_doy_major, int32_t* overflow);

/*
 * Global (constant-in-time) threshold per cell (detect.py:2873-2912).
 *  exact != 0 : float64 np.nanquantile(anom[:, c], q) ("linear"), edges/centres/stats/minmax unused.
 *  exact == 0 : _compute_histogram_quantile_1d (2737-2865) on float64 edges[nb+1] / centres[nb]; thresholds
 *               below lower_bound are clamped; stats counts the out-of-range thresholds and minmax[2]
 *               (device, initialised {+inf, -inf}) receives the un-clamped extremes for the warning text.
 */
int marex_global_threshold_f32(marex_ctx* ctx, const float* anom, int64_t T_out, int64_t C, double q, int exact,
                               const double* edges, const double* centres, int nb, double lower_bound,
                               double upper_bound, double* thr, marex_thr_stats* stats, double* minmax);

/* Validation verdict (marEx/detect.py:205-279, `_validate_data_values`) from what the anomaly entry points leave per cell
 * (`mask` = isfinite(x[0]), `invalid_count` = non-finite values over time), over cells c0 .. c1-1 (a shard's owned cells):
 * out4 (device int64[4]) = {ocean cells, invalid values in ocean cells, ocean cells affected, worst cell's count} --
 * the numbers of the reference's two error messages; shards combine them with one all-reduce (sum, sum, sum, max). */
int marex_validation_summary(marex_ctx* ctx, const uint8_t* mask, const int32_t* invalid_count, int64_t c0, int64_t c1,
                             int64_t* out4);

/* extreme[t, c] = anom[t, c] >= thr[c] (comparison in float64, detect.py:2915) and the count of True */
int marex_mask_ge_const_f32(marex_ctx* ctx, const float* anom, const double* thr, int64_t T_out, int64_t C,
                            uint8_t* extreme, unsigned long long* n_true);

/* std_normalise branch of detrend_harmonic, part 1 (marEx/detect.py:2257-2273):
 *   std_day[d, c]  = population standard deviation (ddof = 0) of anom[t, c] over the rows with dayofyear d + 1 --
 *                    NaN when a term is NaN or the dayofyear never occurs; float64 two-pass (mean, then squared
 *                    deviations, both summed in ascending time), one float64 sqrt, rounded to float32
 *   std_roll[d, c] = float32 sqrt of the centred `window`-day mean of float32(std_day^2) on the wrapped dayofyear
 *                    axis (offsets -window/2 .. window-1-window/2; float64 sum in ascending offset, / window,
 *                    rounded to float32; NaN when a term is NaN)  -- the reference's `STD` variable (window = 30).
 * doy_start / doy_rows: rows of `anom` grouped by dayofyear (int32 [367] / [T]).  std_day is caller workspace. */
int marex_std_rolling_doy_f32(marex_ctx* ctx, const float* anom, int64_t T, int64_t C, const int32_t* doy_start,
                              const int32_t* doy_rows, int window, float* std_day, float* std_roll);

/* part 2 (detect.py:2275-2278): out[t, c] = anom[t, c] / (std_roll[doy(t), c] > 1e-10 ? std_roll : NaN), float32 division */
int marex_div_doy_f32(marex_ctx* ctx, const float* anom, const float* std_roll, const int32_t* doy_start,
                      const int32_t* doy_rows, int64_t T, int64_t C, float* out);

/* Tracker pre-processing, gridded data (marEx/track.py:1520-1676): per timestep, pad the binary image by 2 R cells on
 * every side (regional_mode 0: np.pad "wrap" in both dimensions, 1: "edge"), binary closing then binary opening with the
 * disk x^2 + y^2 < R^2 + 1 (scipy.ndimage semantics, border_value 0), trim the padding, AND with the ocean mask.
 * data / out: uint8 [T, ny, nx] (0 / 1); mask: uint8 [ny, nx].  R = 0: only the mask is applied. */
int marex_fill_holes_u8(marex_ctx* ctx, const uint8_t* data, const uint8_t* mask, int64_t T, int ny, int nx, int R,
                        int regional_mode, uint8_t* out);

/* Temporal binary closing with T_fill + 1 consecutive steps (T_fill even), False outside the series
 * (track.py:1694-1721); the caller follows it with marex_fill_holes_u8(R / 2) as the reference does (1724). */
int marex_time_closing_u8(marex_ctx* ctx, const uint8_t* data, int64_t T, int64_t C, int T_fill, uint8_t* out);

/* Connected components of every timestep on its own (track.py:2013-2031 with time_connectivity = False): 8-connected in
 * (y, x), periodic in x when wrap_x.  labels[i] = 1 + smallest linear index (into the whole [T, ny, nx] array) of the
 * component of cell i, 0 = background -- unique across time like the reference's IDs, but numbered differently from
 * scipy's scan order; areas[r] = number of cells of the component whose smallest index is r, 0 elsewhere. */
int marex_label2d_i32(marex_ctx* ctx, const uint8_t* data, int64_t T, int ny, int nx, int wrap_x, int32_t* labels,
                      int32_t* areas);

/* out[i] = labels[i] > 0 && labels[i] != drop_label && areas[labels[i] - 1] >= area_threshold (track.py:1891-1903;
 * drop_label = the reference's `object_ids_keep[0] = -1`, which removes the first object of the list) */
int marex_filter_by_area_u8(marex_ctx* ctx, const int32_t* labels, const int32_t* areas, int64_t n,
                            double area_threshold, int drop_label, uint8_t* out);

/* The same two stages on an unstructured mesh (track.py:1543-1606 and 1932-2004): nbr int32 [3, C], 0-based, -1 = no
 * neighbour (the reference's `neighbours - 1`).  fill_holes: dilation by R = R sweeps of "OR over the cell and its listed
 * neighbours"; closing with the land set True before the erosion, then opening, NO final land mask (as in the reference).
 * label: components over the listed edges (undirected), land excluded; labels / areas as in marex_label2d_i32. */
int marex_fill_holes_mesh_u8(marex_ctx* ctx, const uint8_t* data, const uint8_t* mask, const int32_t* nbr, int64_t T,
                             int64_t C, int R, uint8_t* out);
int marex_label_mesh_i32(marex_ctx* ctx, const uint8_t* data, const uint8_t* mask, const int32_t* nbr, int64_t T, int64_t C,
                         int32_t* labels, int32_t* areas);

/* Host-side (the only entry point taking HOST pointers, hence "_h"): decode one Blosc-1 frame -- LZ4 codec or memcpy,
 * byte shuffle or none -- the chunk format of the reference's Zarr v2 stores ({"id": "blosc", "cname": "lz4",
 * "shuffle": 1}; examples/batch jobs/run_detect.py:55-83, the fixtures under tests/data).  0 = OK and *out_len = decoded bytes;
 * -5 malformed frame, -6 unsupported codec / filter. */
int marex_blosc_decompress_h(const void* src, int64_t srclen, void* dst, int64_t dstcap, int64_t* out_len);
/* Zstandard frames (RFC 8878, no dictionaries; the content checksum is skipped), host pointers: the inner codec of the Blosc
 * frames of the small coordinate arrays in the reference's stores ({"cname": "zstd"}; xr.open_zarr, run_detect.py:55).
 * Returns 0 and the decoded length, -5 for a malformed / unsupported stream, never reads or writes out of bounds. */
int marex_zstd_decompress_h(const uint8_t* src, int64_t srclen, uint8_t* dst, int64_t dstcap, int64_t* out_len);

/* Host-side inverse: compress nbytes bytes into one Blosc-1 frame (LZ4 codec; byte shuffle when shuffle != 0 and
 * typesize > 1; blocksize <= 0 = 256 KiB) -- the chunk format `extremes_ds.to_zarr(...)` produces through numcodecs'
 * default compressor (examples/batch jobs/run_detect.py:83).  Blocks are split into `typesize` streams exactly when every
 * c-blosc 1.x decoder expects it; incompressible streams / frames are stored, so dstcap >= nbytes + 16 always suffices.
 * 0 = OK and *out_len = frame bytes; -1 bad argument, -4 destination too small. */
int marex_blosc_compress_h(const void* src, int64_t nbytes, int typesize, int shuffle, int64_t blocksize, void* dst,
                           int64_t dstcap, int64_t* out_len);

/* Device-side chunk decoding (compressed bytes cross PCIe, the field is born in HBM): n_streams LZ4 block streams --
 * stream s = comp[src_off[s] .. +csize[s]) -> planes[dst_off[s] .. +rawsz[s]) (csize == rawsz: stored, copied) -- one
 * wave each; *status (device int, zeroed by the caller) counts malformed streams.  Then marex_unshuffle_place turns the
 * byte planes of every Blosc block (planes + blk_off[b], blk_ne[b] elements of `typesize` bytes, byte-shuffled or not)
 * into elements blk_elem0[b] .. +blk_valid[b] of the destination array `out`.  The host parses the frame headers
 * (marex_amd/zarr_io.py). */
int marex_lz4_decode_streams(marex_ctx* ctx, const uint8_t* comp, const int64_t* src_off, const int32_t* csize,
                             const int64_t* dst_off, const int32_t* rawsz, int n_streams, int max_raw, uint8_t* planes,
                             int32_t* status);
int marex_unshuffle_place(marex_ctx* ctx, const uint8_t* planes, const int64_t* blk_off, const int64_t* blk_elem0,
                          const int32_t* blk_ne, const int32_t* blk_valid, int n_blocks, int max_ne, int typesize, int shuffled,
                          uint8_t* out);

/* out[c, r] = in[r, c]  (thresholds [366, C] -> the reference's (cells, dayofyear) order) */
int marex_transpose_f32(marex_ctx* ctx, const float* in, int64_t rows, int64_t cols, float* out);

#ifdef __cplusplus
}
#endif
#endif /* MAREX_HIP_H */
