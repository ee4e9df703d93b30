// marex_morphology.hip -- tracker pre-processing stage (first half): binary closing + opening of the
// extreme mask with a disk (marEx/track.py:1520-1676, gridded branch) and the temporal closing (track.py:1678-1726).
//
// The image of one timestep is padded by 2R cells on every side (wrap = global grid, edge = regional mode), bit-packed
// 64 cells per word, and the four morphological passes (dilate, erode | erode, dilate) run on the packed rows: one
// thread per output word ORs, for every row offset dy of the disk, the row's word dilated horizontally by the disk's
// half-width at dy (shifts across the word boundaries through the two neighbouring words).  Erosion is the dilation of
// the complement with the outside set to 1 -- scipy's border_value = 0 for both operations, which the reference
// inherits through dask_image, so the contamination that creeps 4R cells in from the padded border is reproduced too.
#include "marex_common.hip.h"

typedef unsigned long long u64;

// padded packed layout: [T][Hp = ny + 4R][Wp = ceil((nx + 4R) / 64)] words; bit i of word w = cell x = 64 w + i of the
// padded row; bits past the row end are kept 0.
// one workgroup per padded row: the source row is resolved once (uniform), a wave packs 64 cells per ballot
__global__ void __launch_bounds__(256)
k_morph_pack(const unsigned char* __restrict__ data, long T, int ny, int nx, int R, int regional, int Hp, int Wp,
             u64* __restrict__ packed) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nxp = nx + 4 * R;
    const long rows = T * Hp;
    for (long r = blockIdx.x; r < rows; r += gridDim.x) {
        const int yy = (int)(r % Hp);
        const long t = r / Hp;
        int y = yy - 2 * R;
        if (regional) {  // np.pad mode="edge"
            y = y < 0 ? 0 : (y >= ny ? ny - 1 : y);
        } else {  // np.pad mode="wrap" (both dimensions, as the reference does)
            y %= ny;
            if (y < 0) y += ny;
        }
        const unsigned char* src = data + ((size_t)t * ny + y) * nx;
        for (int w0 = wave; w0 < Wp; w0 += 4 * 8) {  // 8 byte loads in flight per lane before the ballots
            unsigned char v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int xx = 64 * (w0 + 4 * u) + lane;
                v[u] = 0;
                if (w0 + 4 * u < Wp && xx < nxp) {
                    int x = xx - 2 * R;
                    if (regional) {
                        x = x < 0 ? 0 : (x >= nx ? nx - 1 : x);
                    } else {
                        x %= nx;  // np.pad "wrap" repeats as often as needed (grids narrower than 2 R)
                        if (x < 0) x += nx;
                    }
                    v[u] = src[x];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const u64 word = __ballot(v[u] != 0);
                if (lane == 0 && w0 + 4 * u < Wp) packed[(size_t)r * Wp + w0 + 4 * u] = word;
            }
        }
    }
}

// out = dilate(in) with the disk of radius R (half-width hw[|dy|] at row offset dy), the outside of the image
// counting as `outside` (0 or ~0); invert: operate on the complement and complement the result (= erosion).
__global__ void __launch_bounds__(256)
k_morph_pass(const u64* __restrict__ in, u64* __restrict__ out, long T, int Hp, int Wp, int nxp, int R,
             const int* __restrict__ hw, int invert) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long nwords = T * Hp * (long)Wp;
    if (idx >= nwords) return;
    const int w = (int)(idx % Wp);
    const long ty = idx / Wp;
    const int yy = (int)(ty % Hp);
    const long t = ty / Hp;
    const u64 flip = invert ? ~0ull : 0ull;      // complementing turns the stored 0 padding bits into the 1 outside
    const u64 outside = flip;
    const u64* img = in + (size_t)t * Hp * Wp;
    u64 acc = 0;
    for (int dy = -R; dy <= R; ++dy) {
        const int y = yy + dy;
        u64 L = outside, M = outside, Rw = outside;
        if (y >= 0 && y < Hp) {
            const u64* row = img + (size_t)y * Wp;
            M = row[w] ^ flip;
            if (w > 0) L = row[w - 1] ^ flip;
            if (w + 1 < Wp) Rw = row[w + 1] ^ flip;
        }
        const int wd = hw[dy < 0 ? -dy : dy];
        u64 d = M;
        for (int k = 1; k <= wd; ++k) d |= (M >> k) | (Rw << (64 - k)) | (M << k) | (L >> (64 - k));
        acc |= d;
    }
    acc ^= flip;
    // keep the bits past the row end at 0
    const int last_bits = nxp - 64 * (Wp - 1);
    if (w == Wp - 1 && last_bits < 64) acc &= (1ull << last_bits) - 1ull;
    out[idx] = acc;
}

// eight consecutive cells of a row per thread: 8 bits taken from one or two packed words, spread to bytes, ANDed with
// the ocean mask, one 8-byte store
__global__ void __launch_bounds__(256)
k_morph_unpack(const u64* __restrict__ packed, const unsigned char* __restrict__ mask, long T, int ny, int nx, int R,
               int Hp, int Wp, unsigned char* __restrict__ out) {
    const int ng = (nx + 7) >> 3;  // groups of 8 cells per row
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= T * ny * (long)ng) return;
    const int g = (int)(idx % ng);
    const long ty = idx / ng;
    const int y = (int)(ty % ny);
    const long t = ty / ny;
    const int x = g * 8, xx = x + 2 * R, yy = y + 2 * R;
    const u64* row = packed + ((size_t)t * Hp + yy) * Wp;
    const int w = xx >> 6, sh = xx & 63;
    u64 bits = row[w] >> sh;
    if (sh > 56 && w + 1 < Wp) bits |= row[w + 1] << (64 - sh);
    const unsigned char* mrow = mask + (size_t)y * nx + x;
    unsigned char* orow = out + ((size_t)t * ny + y) * nx + x;
    if (x + 8 <= nx && (nx & 7) == 0) {
        u64 spread = 0;  // byte i = bit i
#pragma unroll
        for (int i = 0; i < 8; ++i) spread |= ((bits >> i) & 1ull) << (8 * i);
        const u64 m = *reinterpret_cast<const u64*>(mrow);
        // mask bytes are 0 / 1
        *reinterpret_cast<u64*>(orow) = spread & m;
    } else {
        for (int i = 0; i < 8 && x + i < nx; ++i) orow[i] = (((bits >> i) & 1ull) && mrow[i]) ? 1 : 0;
    }
}

__global__ void __launch_bounds__(256)
k_mask_and(const unsigned char* __restrict__ data, const unsigned char* __restrict__ mask, long T, long C,
           unsigned char* __restrict__ out) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= T * C) return;
    out[idx] = (data[idx] && mask[idx % C]) ? 1 : 0;
}

// temporal closing with a centred window of 2c+1 steps, False outside the series (track.py:1694-1718: kernel of
// T_fill + 1 ones, T_fill even, padding of T_fill + 1 steps so that the border is never reached).
// One thread = 16 consecutive cells x a block of timesteps; bytes are 0 / 1, so OR / AND act on whole 16-byte vectors.
// The dilated rows of the block (+ c on either side) are kept in a small register ring.
#define TC_MAXC 8
#define TC_TBLOCK 32
__device__ __forceinline__ uint4 or4(uint4 a, uint4 b) { return make_uint4(a.x | b.x, a.y | b.y, a.z | b.z, a.w | b.w); }
__device__ __forceinline__ uint4 and4(uint4 a, uint4 b) { return make_uint4(a.x & b.x, a.y & b.y, a.z & b.z, a.w & b.w); }

__global__ void __launch_bounds__(256)
k_time_closing16(const unsigned char* __restrict__ data, long T, long C, int c, unsigned char* __restrict__ out) {
    const long c16 = ((long)blockIdx.x * 256 + threadIdx.x) * 16;
    if (c16 >= C) return;
    const long t0 = (long)blockIdx.y * TC_TBLOCK;
    const long t1 = t0 + TC_TBLOCK < T ? t0 + TC_TBLOCK : T;
    auto row = [&](long t) {
        return (t >= 0 && t < T) ? *reinterpret_cast<const uint4*>(data + (size_t)t * C + c16) : make_uint4(0, 0, 0, 0);
    };
    auto dil = [&](long t) {  // OR of rows t-c .. t+c; zero outside the series (the padded part of the reference's array)
        if (t < -(long)c - 1 || t > T + c) return make_uint4(0, 0, 0, 0);
        uint4 d = row(t - c);
        for (int i = -c + 1; i <= c; ++i) d = or4(d, row(t + i));
        return d;
    };
    for (long t = t0; t < t1; ++t) {
        uint4 e = dil(t - c);
        for (int j = -c + 1; j <= c; ++j) e = and4(e, dil(t + j));
        *reinterpret_cast<uint4*>(out + (size_t)t * C + c16) = e;
    }
}

__global__ void __launch_bounds__(256)
k_time_closing(const unsigned char* __restrict__ data, long T, long C, int c, unsigned char* __restrict__ out) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= T * C) return;
    const long cell = idx % C, t = idx / C;
    bool all = true;
    for (int j = -c; j <= c && all; ++j) {  // erosion of ...
        bool any = false;
        for (int i = -c; i <= c; ++i) {      // ... the dilation
            const long tt = t + j + i;
            if (tt >= 0 && tt < T && data[(size_t)tt * C + cell]) any = true;
        }
        all = any;
    }
    out[idx] = all ? 1 : 0;
}

static int ensure_scratch(marex_ctx* ctx, size_t need) {
    if (need > ctx->morph_scratch_bytes) {
        if (ctx->morph_scratch) HIP_TRY(ctx, hipFree(ctx->morph_scratch));
        ctx->morph_scratch = nullptr;
        ctx->morph_scratch_bytes = 0;
        HIP_TRY(ctx, hipMalloc((void**)&ctx->morph_scratch, need));
        ctx->morph_scratch_bytes = need;
    }
    return 0;
}

extern "C" int marex_fill_holes_u8(marex_ctx* ctx, const uint8_t* data, const uint8_t* mask, int64_t T, int ny, int nx,
                                   int R, int regional_mode, uint8_t* out) {
    if (!ctx) return -1;
    if (!data || !mask || !out || T <= 0 || ny <= 0 || nx <= 0) return fail(ctx, -1, "marex_fill_holes_u8: null pointer or empty shape");
    if (R < 0 || R > 63) return fail(ctx, -4, "marex_fill_holes_u8: R_fill must be in 0..63");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    LaunchTimer lt(ctx, MAREX_K_MORPH);
    const long C = (long)ny * nx;
    if (R == 0) {  // no morphology, only the land mask (track.py:1620-1621, 1674)
        hipLaunchKernelGGL(k_mask_and, dim3((unsigned)((T * C + 255) / 256)), dim3(256), 0, ctx->stream, data, mask, (long)T, C, out);
        HIP_TRY(ctx, hipGetLastError());
        return 0;
    }
    const int Hp = ny + 4 * R, nxp = nx + 4 * R, Wp = (nxp + 63) / 64;
    const size_t words = (size_t)T * Hp * Wp;
    const size_t need = 2 * words * sizeof(u64) + 64 * sizeof(int);
    if (int rc = ensure_scratch(ctx, need)) return rc;
    u64* a = reinterpret_cast<u64*>(ctx->morph_scratch);
    u64* b = a + words;
    int* hw_dev = reinterpret_cast<int*>(b + words);
    int hw[64];
    for (int dy = 0; dy <= R; ++dy) {  // se_kernel = x^2 + y^2 < R^2 + 1  (track.py:1613-1616)
        int w = 0;
        while ((w + 1) * (w + 1) + dy * dy < R * R + 1) ++w;
        hw[dy] = w;
    }
    HIP_TRY(ctx, hipMemcpyAsync(hw_dev, hw, (R + 1) * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // hw lives on this stack frame
    const unsigned gw = (unsigned)((words + 255) / 256);
    hipLaunchKernelGGL(k_morph_pack, dim3((unsigned)((long)T * Hp < 1048576 ? (long)T * Hp : 1048576)), dim3(256), 0, ctx->stream, data, (long)T, ny, nx, R, regional_mode, Hp, Wp, a);
    hipLaunchKernelGGL(k_morph_pass, dim3(gw), dim3(256), 0, ctx->stream, a, b, (long)T, Hp, Wp, nxp, R, hw_dev, 0);  // closing:
    hipLaunchKernelGGL(k_morph_pass, dim3(gw), dim3(256), 0, ctx->stream, b, a, (long)T, Hp, Wp, nxp, R, hw_dev, 1);  //   dilate, erode
    hipLaunchKernelGGL(k_morph_pass, dim3(gw), dim3(256), 0, ctx->stream, a, b, (long)T, Hp, Wp, nxp, R, hw_dev, 1);  // opening:
    hipLaunchKernelGGL(k_morph_pass, dim3(gw), dim3(256), 0, ctx->stream, b, a, (long)T, Hp, Wp, nxp, R, hw_dev, 0);  //   erode, dilate
    hipLaunchKernelGGL(k_morph_unpack, dim3((unsigned)((T * ny * (long)((nx + 7) / 8) + 255) / 256)), dim3(256), 0, ctx->stream, a,
                       mask, (long)T, ny, nx, R, Hp, Wp, out);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

extern "C" int marex_time_closing_u8(marex_ctx* ctx, const uint8_t* data, int64_t T, int64_t C, int T_fill, uint8_t* out) {
    if (!ctx) return -1;
    if (!data || !out || T <= 0 || C <= 0) return fail(ctx, -1, "marex_time_closing_u8: null pointer or empty shape");
    if (T_fill < 0 || (T_fill & 1)) return fail(ctx, -1, "marex_time_closing_u8: T_fill must be even and >= 0");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    LaunchTimer lt(ctx, MAREX_K_MORPH);
    if ((C & 15) == 0 && (((uintptr_t)data | (uintptr_t)out) & 15) == 0) {
        dim3 grid((unsigned)((C / 16 + 255) / 256), (unsigned)((T + TC_TBLOCK - 1) / TC_TBLOCK));
        hipLaunchKernelGGL(k_time_closing16, grid, dim3(256), 0, ctx->stream, data, (long)T, (long)C, T_fill / 2, out);
    } else {
        hipLaunchKernelGGL(k_time_closing, dim3((unsigned)((T * C + 255) / 256)), dim3(256), 0, ctx->stream, data, (long)T, (long)C,
                           T_fill / 2, out);
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Connected components per timestep (track.py:1912-2049 structured branch with time_connectivity = False: 8-connected
// in (y, x), periodic in x unless regional) and the area filter of filter_small_objects (track.py:1755-1911).
//
// Label of a cell = 1 + the smallest linear index (over the whole [T, ny, nx] array) of its component, 0 = background:
// concurrent union-find, one merge pass over the four "backward" neighbours (W, NW, N, NE), a compression pass, and a
// flattening pass that also accumulates the areas at the roots.  Label VALUES differ from scipy's scan-order numbering; memberships and areas do not.
// ------------------------------------------------------------------------------------------------
// Union-find in the style of ECL-CC: a root is hooked under a smaller root with ONE compare-and-swap that only succeeds
// while it is still a root; finds shorten the paths they walk with plain stores (path halving) -- safe because only
// non-roots are rewritten (a non-root never becomes a root again, so it can never be the target of a hook) and the new
// parent is always one of its ancestors.  Parents only decrease, so every loop terminates.
__device__ __forceinline__ int uf_find(int* __restrict__ parent, int i) {
    int p = parent[i];
    while (p != i) {
        const int g = parent[p];
        if (g != p) parent[i] = g;
        i = p;
        p = g;
    }
    return i;
}

__device__ __forceinline__ void uf_union(int* __restrict__ parent, int a, int b) {
    a = uf_find(parent, a);
    b = uf_find(parent, b);
    while (a != b) {
        if (a < b) {
            const int s = a;
            a = b;
            b = s;
        }
        const int old = atomicCAS(&parent[a], a, b);  // a > b: hook a under b if a is still a root
        if (old == a) return;
        a = uf_find(parent, old);  // a was hooked elsewhere meanwhile: continue from its new root
        b = uf_find(parent, b);
    }
}

// Initial parents: every True cell points at the first cell of its horizontal run inside its wave's 64 consecutive
// cells (ballot + count-leading-zeros, no memory traffic) -- most "west" unions never have to happen.
__global__ void __launch_bounds__(256)
k_ccl_init(const unsigned char* __restrict__ data, long n, int nx, int* __restrict__ parent) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool on = i < n && data[i];
    const bool row_start = i < n && (i % nx) == 0;
    const unsigned long long on_m = __ballot(on);
    // lane continues the run of lane-1: both set, same row
    const unsigned long long cont = on_m & (on_m << 1) & ~__ballot(row_start);
    if (i >= n) return;
    if (!on) {
        parent[i] = -1;
        return;
    }
    const unsigned long long upto = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
    const unsigned long long breaks = ~cont & upto;  // bit 0 is always a break (cont has bit 0 clear)
    const int start = 63 - __clzll((long long)breaks);
    parent[i] = (int)(i - lane + start);
}

__global__ void __launch_bounds__(256)
k_ccl_merge(const unsigned char* __restrict__ data, long T, int ny, int nx, int wrap_x, int* __restrict__ parent) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long n = T * ny * (long)nx;
    if (i >= n || !data[i]) return;
    const int x = (int)(i % nx);
    const int y = (int)((i / nx) % ny);
    const long base = i - x;  // start of this row
    // west: only where the in-wave run labelling could not see it (first lane of a wave) and across the periodic seam
    if (x > 0) {
        if ((threadIdx.x & 63) == 0 && data[i - 1]) uf_union(parent, (int)i, (int)(i - 1));
    } else if (wrap_x && nx > 1 && data[base + nx - 1]) {
        uf_union(parent, (int)i, (int)(base + nx - 1));
    }
    if (y > 0) {
        // One union per OVERLAP of a run with a run of the row above, not one per cell: a cell links to N only where
        // the overlap starts (its W neighbour is not part of the same overlap), to NW / NE only at run ends where the
        // diagonal is the sole contact.  In blobs that cover half the ocean this removes >90 % of the finds.
        const long up = base - nx;
        int xw = x - 1, xe = x + 1;
        bool okw = true, oke = true;
        if (xw < 0) {
            okw = wrap_x && nx > 1;
            xw += nx;
        }
        if (xe >= nx) {
            oke = wrap_x && nx > 1;
            xe -= nx;
        }
        const bool n_ = data[up + x] != 0;
        const bool nw = okw && data[up + xw] != 0, ne = oke && data[up + xe] != 0;
        const bool w_ = okw && data[base + xw] != 0, e_ = oke && data[base + xe] != 0;
        // x == 0 always links on its own account: on a periodic row an overlap may have no beginning otherwise
        const bool chain = x > 0;
        if (n_) {
            if (!(chain && w_ && nw)) uf_union(parent, (int)i, (int)(up + x));
        } else {
            if (nw && !(chain && w_)) uf_union(parent, (int)i, (int)(up + xw));
            if (ne && !e_) uf_union(parent, (int)i, (int)(up + xe));
        }
    }
}

// after all unions: point every cell straight at its root (plain stores are fine: nothing is merged any more)
__global__ void __launch_bounds__(256)
k_ccl_compress(long n, int* __restrict__ parent) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || parent[i] < 0) return;
    int r = (int)i, p = parent[i];
    while (p != r) {
        r = p;
        p = parent[r];
    }
    parent[i] = r;
}

// labels + areas; the lanes of a wave that share a root add their count with ONE atomic (a blob covering half the
// ocean would otherwise serialise hundreds of thousands of atomics on one address)
__global__ void __launch_bounds__(256)
k_ccl_flatten(long n, const int* __restrict__ parent, int* __restrict__ labels, int* __restrict__ areas) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    int r = -1;
    if (i < n) {
        r = parent[i];
        labels[i] = r < 0 ? 0 : r + 1;
    }
    unsigned long long todo = __ballot(r >= 0);
    while (todo) {
        const int lead = __ffsll((long long)todo) - 1;
        const int rl = __shfl(r, lead, 64);
        const unsigned long long same = __ballot(r == rl) & todo;
        if ((int)(threadIdx.x & 63) == lead) atomicAdd(&areas[rl], __popcll(same));
        todo &= ~same;
    }
}

__global__ void __launch_bounds__(256)
k_ccl_filter(const int* __restrict__ labels, const int* __restrict__ areas, long n, double area_threshold, int drop_label,
             unsigned char* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int l = labels[i];
    out[i] = (l > 0 && l != drop_label && (double)areas[l - 1] >= area_threshold) ? 1 : 0;
}

extern "C" int marex_label2d_i32(marex_ctx* ctx, const uint8_t* data, int64_t T, int ny, int nx, int wrap_x,
                                 int32_t* labels, int32_t* areas) {
    if (!ctx) return -1;
    if (!data || !labels || !areas || T <= 0 || ny <= 0 || nx <= 0) return fail(ctx, -1, "marex_label2d_i32: null pointer or empty shape");
    const long n = (long)T * ny * nx;
    if (n >= 2147483647L) return fail(ctx, -4, "marex_label2d_i32: more than 2^31 - 1 cells; label the series in time blocks");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    LaunchTimer lt(ctx, MAREX_K_MORPH);
    const unsigned g = (unsigned)((n + 255) / 256);
    // `labels` doubles as the parent array during the merge; `areas` must start at zero
    HIP_TRY(ctx, hipMemsetAsync(areas, 0, (size_t)n * sizeof(int), ctx->stream));
    hipLaunchKernelGGL(k_ccl_init, dim3(g), dim3(256), 0, ctx->stream, data, n, nx, labels);
    hipLaunchKernelGGL(k_ccl_merge, dim3(g), dim3(256), 0, ctx->stream, data, (long)T, ny, nx, wrap_x, labels);
    // flatten in place is a race (a cell's parent may be overwritten by its label while another walks through it):
    // write the labels to `areas`' sibling buffer instead -- here: a scratch copy of the parents
    if (int rc = ensure_scratch(ctx, (size_t)n * sizeof(int))) return rc;
    int* parent = reinterpret_cast<int*>(ctx->morph_scratch);
    hipLaunchKernelGGL(k_ccl_compress, dim3(g), dim3(256), 0, ctx->stream, n, labels);
    HIP_TRY(ctx, hipMemcpyAsync(parent, labels, (size_t)n * sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
    hipLaunchKernelGGL(k_ccl_flatten, dim3(g), dim3(256), 0, ctx->stream, n, parent, labels, areas);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

extern "C" int marex_filter_by_area_u8(marex_ctx* ctx, const int32_t* labels, const int32_t* areas, int64_t n,
                                       double area_threshold, int drop_label, uint8_t* out) {
    if (!ctx) return -1;
    if (!labels || !areas || !out || n <= 0) return fail(ctx, -1, "marex_filter_by_area_u8: null pointer or empty shape");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    LaunchTimer lt(ctx, MAREX_K_MORPH);
    hipLaunchKernelGGL(k_ccl_filter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, labels, areas, (long)n,
                       area_threshold, drop_label, out);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Unstructured meshes (track.py:1093-1117, 1543-1606, 1932-2004, 1776-1857): a cell's neighbourhood is itself plus up
// to three edge neighbours (nbr[3][C], 0-based, -1 = none); dilation by R = R sweeps of "OR over the neighbourhood".
// ------------------------------------------------------------------------------------------------
// pre-operation applied to every value read by the FIRST sweep of a group (fuses the reference's elementwise steps):
//   0: v                   1: !(v || land)      (b[:, ~mask] = True; ~b)
//   2: v && !land          (~(~v | land))       3: !v
__device__ __forceinline__ bool graph_pre(bool v, bool land, int pre) {
    return pre == 0 ? v : pre == 1 ? !(v || land) : pre == 2 ? (v && !land) : !v;
}

__global__ void __launch_bounds__(256)
k_graph_sweep(const unsigned char* __restrict__ in, const unsigned char* __restrict__ mask, const int* __restrict__ nbr,
              long T, long C, int pre, int dilate, unsigned char* __restrict__ out) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= T * C) return;
    const long c = idx % C;
    const unsigned char* row = in + (idx - c);
    bool v = graph_pre(row[c] != 0, mask[c] == 0, pre);
    if (dilate) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int n = nbr[(size_t)k * C + c];
            if (n >= 0) v = v || graph_pre(row[n] != 0, mask[n] == 0, pre);
        }
    }
    out[idx] = v ? 1 : 0;
}

extern "C" int marex_fill_holes_mesh_u8(marex_ctx* ctx, const uint8_t* data, const uint8_t* mask, const int32_t* nbr,
                                        int64_t T, int64_t C, int R, uint8_t* out) {
    if (!ctx) return -1;
    if (!data || !mask || !nbr || !out || T <= 0 || C <= 0) return fail(ctx, -1, "marex_fill_holes_mesh_u8: null pointer or empty shape");
    if (R < 0 || R > 1024) return fail(ctx, -4, "marex_fill_holes_mesh_u8: R_fill must be in 0..1024");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    LaunchTimer lt(ctx, MAREX_K_MORPH);
    const size_t n = (size_t)T * C;
    if (int rc = ensure_scratch(ctx, 2 * n)) return rc;
    unsigned char* bufs[2] = {ctx->morph_scratch, ctx->morph_scratch + n};
    const unsigned g = (unsigned)((n + 255) / 256);
    const unsigned char* cur = data;
    int flip = 0;
    // dilation | (land := True, complement) dilation | (complement, land := True, complement) dilation | complement, dilation
    const int pres[4] = {0, 1, 2, 3};
    for (int grp = 0; grp < 4; ++grp) {
        const int sweeps = R > 0 ? R : 1;  // R = 0: only the elementwise step of the group
        for (int s = 0; s < sweeps; ++s) {
            const bool last = grp == 3 && s == sweeps - 1;
            unsigned char* dst = last ? out : bufs[flip];
            hipLaunchKernelGGL(k_graph_sweep, dim3(g), dim3(256), 0, ctx->stream, cur, mask, nbr, (long)T, (long)C,
                               s == 0 ? pres[grp] : 0, R > 0 ? 1 : 0, dst);
            cur = dst;
            flip ^= 1;
        }
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// connected components per timestep over the mesh edges, land excluded (track.py:1985, 1947-1981)
__global__ void __launch_bounds__(256)
k_mesh_ccl_init(const unsigned char* __restrict__ data, const unsigned char* __restrict__ mask, long T, long C,
                int* __restrict__ parent) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < T * C) parent[i] = (data[i] && mask[i % C]) ? (int)i : -1;
}

__global__ void __launch_bounds__(256)
k_mesh_ccl_merge(const int* __restrict__ nbr, long T, long C, int* __restrict__ parent) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= T * C || parent[i] < 0) return;
    const long c = i % C, base = i - c;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int n = nbr[(size_t)k * C + c];
        // an edge is an undirected link whichever end lists it (connected_components(directed=False), track.py:1979)
        if (n >= 0 && n != c && parent[base + n] >= 0) uf_union(parent, (int)i, (int)(base + n));
    }
}

extern "C" int marex_label_mesh_i32(marex_ctx* ctx, const uint8_t* data, const uint8_t* mask, const int32_t* nbr, int64_t T,
                                    int64_t C, int32_t* labels, int32_t* areas) {
    if (!ctx) return -1;
    if (!data || !mask || !nbr || !labels || !areas || T <= 0 || C <= 0) return fail(ctx, -1, "marex_label_mesh_i32: null pointer or empty shape");
    const long n = (long)T * C;
    if (n >= 2147483647L) return fail(ctx, -4, "marex_label_mesh_i32: more than 2^31 - 1 cells; label the series in time blocks");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    LaunchTimer lt(ctx, MAREX_K_MORPH);
    const unsigned g = (unsigned)((n + 255) / 256);
    HIP_TRY(ctx, hipMemsetAsync(areas, 0, (size_t)n * sizeof(int), ctx->stream));
    hipLaunchKernelGGL(k_mesh_ccl_init, dim3(g), dim3(256), 0, ctx->stream, data, mask, (long)T, (long)C, labels);
    hipLaunchKernelGGL(k_mesh_ccl_merge, dim3(g), dim3(256), 0, ctx->stream, nbr, (long)T, (long)C, labels);
    hipLaunchKernelGGL(k_ccl_compress, dim3(g), dim3(256), 0, ctx->stream, n, labels);
    if (int rc = ensure_scratch(ctx, (size_t)n * sizeof(int))) return rc;
    int* parent = reinterpret_cast<int*>(ctx->morph_scratch);
    HIP_TRY(ctx, hipMemcpyAsync(parent, labels, (size_t)n * sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
    hipLaunchKernelGGL(k_ccl_flatten, dim3(g), dim3(256), 0, ctx->stream, n, parent, labels, areas);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}
