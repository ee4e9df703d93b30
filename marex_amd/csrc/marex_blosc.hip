// marex_blosc.hip -- host-side decoder for the chunk format of the reference's Zarr v2 stores (first
// step): Blosc-1 frames with the LZ4 codec and byte shuffle, which is what `run_detect.py` / the reference's own test
// fixtures are written with (.zarray: {"id": "blosc", "cname": "lz4", "shuffle": 1}).  Restated from the published
// formats (Blosc 1.x header / block layout, LZ4 block format); no third-party code.  Host pointers only ("_h").
#include "marex_common.hip.h"

namespace {

// LZ4 block format: sequences of [token][literal length+][literals][offset16][match length+]; the last sequence ends
// after its literals.  Returns the number of bytes written or -1 on malformed input / overflow.
long lz4_block_decode(const unsigned char* src, long srclen, unsigned char* dst, long dstcap) {
    const unsigned char* ip = src;
    const unsigned char* const iend = src + srclen;
    unsigned char* op = dst;
    unsigned char* const oend = dst + dstcap;
    while (ip < iend) {
        const unsigned token = *ip++;
        long lit = token >> 4;
        if (lit == 15) {
            unsigned b;
            do {
                if (ip >= iend) return -1;
                b = *ip++;
                lit += b;
            } while (b == 255);
        }
        if (lit > iend - ip || lit > oend - op) return -1;
        memcpy(op, ip, (size_t)lit);
        ip += lit;
        op += lit;
        if (ip >= iend) break;  // last sequence: literals only
        if (iend - ip < 2) return -1;
        const long offset = (long)ip[0] | ((long)ip[1] << 8);
        ip += 2;
        if (offset == 0 || offset > op - dst) return -1;
        long mlen = (long)(token & 15u) + 4;
        if ((token & 15u) == 15u) {
            unsigned b;
            do {
                if (ip >= iend) return -1;
                b = *ip++;
                mlen += b;
            } while (b == 255);
        }
        if (mlen > oend - op) return -1;
        const unsigned char* m = op - offset;
        for (long i = 0; i < mlen; ++i) op[i] = m[i];  // byte-wise: matches may overlap their own output
        op += mlen;
    }
    return (long)(op - dst);
}

inline unsigned rd32(const unsigned char* p) { return (unsigned)p[0] | ((unsigned)p[1] << 8) | ((unsigned)p[2] << 16) | ((unsigned)p[3] << 24); }

}  // namespace

// Decompress one Blosc-1 frame.  Returns 0 and the decoded size in *out_len, or a negative code:
//  -1 bad argument, -5 malformed frame, -6 unsupported codec / filter (anything but LZ4 or memcpy, byte shuffle or none)
extern "C" int marex_blosc_decompress_h(const void* src_v, int64_t srclen, void* dst_v, int64_t dstcap, int64_t* out_len) {
    if (!src_v || !dst_v || !out_len || srclen < 16) return -1;
    const unsigned char* src = static_cast<const unsigned char*>(src_v);
    unsigned char* dst = static_cast<unsigned char*>(dst_v);
    const unsigned flags = src[2];
    const long typesize = src[3];
    const long nbytes = rd32(src + 4), blocksize = rd32(src + 8), cbytes = rd32(src + 12);
    if (cbytes > srclen || nbytes > dstcap || typesize < 1) return -5;
    *out_len = nbytes;
    if (nbytes == 0) return 0;
    if (flags & 0x2) {  // memcpyed frame
        if (16 + nbytes > srclen) return -5;
        memcpy(dst, src + 16, (size_t)nbytes);
        return 0;
    }
    if (flags & 0x4) return -6;             // bit shuffle
    const unsigned codec = flags >> 5;      // 0 blosclz, 1 lz4 / lz4hc, 2 snappy, 3 zlib, 4 zstd
    if (codec != 1) return -6;
    if (blocksize <= 0) return -5;
    const bool shuffle = (flags & 0x1) && typesize > 1;
    const bool dont_split = (flags & 0x10) != 0;
    const long nblocks = (nbytes + blocksize - 1) / blocksize;
    if (16 + 4 * nblocks > srclen) return -5;
    std::vector<unsigned char> tmp((size_t)blocksize);
    for (long j = 0; j < nblocks; ++j) {
        const long bsize = (j == nblocks - 1 && nbytes % blocksize) ? nbytes % blocksize : blocksize;
        const bool leftover = bsize != blocksize;
        const long nsplits = (!dont_split && !leftover && typesize <= 16 && bsize % typesize == 0) ? typesize : 1;
        const long neblock = bsize / nsplits;
        long pos = rd32(src + 16 + 4 * j);
        unsigned char* out = shuffle ? tmp.data() : dst + j * blocksize;
        for (long s = 0; s < nsplits; ++s) {
            if (pos + 4 > srclen) return -5;
            const long cb = rd32(src + pos);
            pos += 4;
            if (cb < 0 || pos + cb > srclen) return -5;
            if (cb == neblock) {
                memcpy(out + s * neblock, src + pos, (size_t)neblock);
            } else if (lz4_block_decode(src + pos, cb, out + s * neblock, neblock) != neblock) {
                return -5;
            }
            pos += cb;
        }
        if (shuffle) {  // undo the byte shuffle: plane k of the block holds byte k of every element
            unsigned char* d = dst + j * blocksize;
            const long ne = bsize / typesize;
            for (long k = 0; k < typesize; ++k) {
                const unsigned char* plane = tmp.data() + k * ne;
                for (long i = 0; i < ne; ++i) d[i * typesize + k] = plane[i];
            }
            const long rest = bsize - ne * typesize;
            if (rest) memcpy(d + ne * typesize, tmp.data() + ne * typesize, (size_t)rest);
        }
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Device side: decode many LZ4 streams at once (one wave per stream) and undo the byte shuffle while placing the
// elements in the destination array -- compressed chunks go over PCIe, the float field is born in HBM.
//
// A wave walks its stream's sequences in lock step (token / lengths / offset are wave-uniform); literal and match
// copies are spread over the 64 lanes.  Matches are served from an LDS ring that mirrors the last RING bytes of the
// output (RING >= 64 KiB whenever the stream is longer than that: LZ4 offsets reach 65 535 bytes back), so nothing is
// ever re-read from global memory and the wave never waits for its own global stores; every step reads all its sources
// before it writes (the LDS queue of a wave is in order), and long overlapping matches
// (runs) are cut into pieces so that their source bytes are still in the ring.
// ------------------------------------------------------------------------------------------------
// stored (incompressible) splits: plain copies, spread over many threads
__global__ void __launch_bounds__(256)
k_stored_streams(const unsigned char* __restrict__ comp, const long* __restrict__ src_off, const int* __restrict__ csize,
                 const long* __restrict__ dst_off, const int* __restrict__ rawsz, unsigned char* __restrict__ out) {
    const int s = blockIdx.y;
    const int raw = rawsz[s];
    if (csize[s] != raw) return;
    const int i = (int)blockIdx.x * 256 + threadIdx.x;
    if (i < raw) out[dst_off[s] + i] = comp[src_off[s] + i];
}

__global__ void __launch_bounds__(64)
k_lz4_streams(const unsigned char* __restrict__ comp, const long* __restrict__ src_off, const int* __restrict__ csize,
              const long* __restrict__ dst_off, const int* __restrict__ rawsz, int ring_mask,
              unsigned char* __restrict__ out, int* __restrict__ status) {
    extern __shared__ unsigned char ring[];
    const int s = blockIdx.x, lane = threadIdx.x;
    const unsigned char* src = comp + src_off[s];
    unsigned char* dst = out + dst_off[s];
    const int cs = csize[s], raw = rawsz[s];
    if (cs == raw) return;  // stored split: k_stored_streams
    auto byte_at = [&](int p) -> unsigned { return src[p]; };  // wave-uniform address: one broadcast load (L1 resident)
    int ip = 0, op = 0;
    bool bad = false;
    while (ip < cs) {
        const unsigned token = byte_at(ip++);
        int lit = (int)(token >> 4);
        if (lit == 15) {
            unsigned b = 255;
            while (b == 255 && ip < cs) {
                b = byte_at(ip++);
                lit += (int)b;
            }
        }
        if (lit > cs - ip || lit > raw - op) {
            bad = true;
            break;
        }
        for (int i = lane; i < lit; i += 64) {
            const unsigned char c = (unsigned char)byte_at(ip + i);
            ring[(op + i) & ring_mask] = c;
            dst[op + i] = c;
        }
        ip += lit;
        op += lit;
        if (ip >= cs) break;  // last sequence: literals only
        if (cs - ip < 2) {
            bad = true;
            break;
        }
        const unsigned o_lo = byte_at(ip), o_hi = byte_at(ip + 1);
        const int offset = (int)(o_lo | (o_hi << 8));
        ip += 2;
        int mlen = (int)(token & 15u) + 4;
        if ((token & 15u) == 15u) {
            unsigned b = 255;
            while (b == 255 && ip < cs) {
                b = byte_at(ip++);
                mlen += (int)b;
            }
        }
        if (offset == 0 || offset > op || mlen > raw - op) {
            bad = true;
            break;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);  /* lgkmcnt(0): LDS only, never the global stores */  // the literals are in the ring before anybody reads them back
        if (offset >= 64) {
            for (int base = 0; base < mlen; base += 64) {
                const int i = base + lane;
                unsigned char v = 0;
                if (i < mlen) v = ring[(op - offset + i) & ring_mask];
                __builtin_amdgcn_s_waitcnt(0xC07F);  /* lgkmcnt(0): LDS only, never the global stores */
                if (i < mlen) {
                    ring[(op + i) & ring_mask] = v;
                    dst[op + i] = v;
                }
                __builtin_amdgcn_s_waitcnt(0xC07F);  /* lgkmcnt(0): LDS only, never the global stores */
            }
        } else {  // the last `offset` bytes repeat: pieces short enough that their source stays in the ring
            int done = 0;
            while (done < mlen) {
                const int piece = mlen - done < 8192 ? mlen - done : 8192;
                const int o = op + done;
                for (int i = lane; i < piece; i += 64) {
                    const unsigned char v = ring[(o - offset + (i % offset)) & ring_mask];
                    ring[(o + i) & ring_mask] = v;
                    dst[o + i] = v;
                }
                __builtin_amdgcn_s_waitcnt(0xC07F);  /* lgkmcnt(0): LDS only, never the global stores */
                done += piece;
            }
        }
        op += mlen;
    }
    if (lane == 0 && (bad || op != raw)) atomicAdd(status, 1);
}

// byte planes of a shuffled block -> elements at their place in the destination (typesize bytes each)
__global__ void __launch_bounds__(256)
k_unshuffle_place(const unsigned char* __restrict__ planes, const long* __restrict__ blk_off, const long* __restrict__ blk_elem0,
                  const int* __restrict__ blk_ne, const int* __restrict__ blk_valid, int typesize, int shuffled,
                  unsigned char* __restrict__ out) {
    const int b = blockIdx.y;
    const int e = (int)blockIdx.x * 256 + threadIdx.x;
    if (e >= blk_valid[b]) return;
    const int ne = blk_ne[b];
    const unsigned char* p = planes + blk_off[b];
    unsigned char* o = out + (size_t)(blk_elem0[b] + e) * typesize;
    if (shuffled) {
        for (int k = 0; k < typesize; ++k) o[k] = p[(size_t)k * ne + e];
    } else {
        for (int k = 0; k < typesize; ++k) o[k] = p[(size_t)e * typesize + k];
    }
}

extern "C" int marex_lz4_decode_streams(marex_ctx* ctx, const uint8_t* comp, const int64_t* src_off, const int32_t* csize,
                                        const int64_t* dst_off, const int32_t* rawsz, int n_streams, int max_raw,
                                        uint8_t* planes, int32_t* status) {
    if (!ctx) return -1;
    if (!comp || !src_off || !csize || !dst_off || !rawsz || !planes || !status || n_streams <= 0 || max_raw <= 0)
        return fail(ctx, -1, "marex_lz4_decode_streams: null pointer or empty table");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int ring = 1024;
    while (ring < max_raw && ring < 65536) ring <<= 1;  // power of two: ring positions are masked
    if (ring > 48 * 1024)
        HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_lz4_streams, hipFuncAttributeMaxDynamicSharedMemorySize, ring));
    hipLaunchKernelGGL(k_stored_streams, dim3((unsigned)((max_raw + 255) / 256), (unsigned)n_streams), dim3(256), 0, ctx->stream, comp,
                       reinterpret_cast<const long*>(src_off), csize, reinterpret_cast<const long*>(dst_off), rawsz, planes);
    hipLaunchKernelGGL(k_lz4_streams, dim3((unsigned)n_streams), dim3(64), (size_t)ring, ctx->stream, comp,
                       reinterpret_cast<const long*>(src_off), csize, reinterpret_cast<const long*>(dst_off), rawsz, ring - 1, planes,
                       status);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

extern "C" int marex_unshuffle_place(marex_ctx* ctx, const uint8_t* planes, const int64_t* blk_off, const int64_t* blk_elem0,
                                     const int32_t* blk_ne, const int32_t* blk_valid, int n_blocks, int max_ne, int typesize,
                                     int shuffled, uint8_t* out) {
    if (!ctx) return -1;
    if (!planes || !blk_off || !blk_elem0 || !blk_ne || !blk_valid || !out || n_blocks <= 0 || max_ne <= 0 || typesize < 1)
        return fail(ctx, -1, "marex_unshuffle_place: null pointer or empty table");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_unshuffle_place, dim3((unsigned)((max_ne + 255) / 256), (unsigned)n_blocks), dim3(256), 0, ctx->stream, planes,
                       reinterpret_cast<const long*>(blk_off), reinterpret_cast<const long*>(blk_elem0), blk_ne, blk_valid, typesize,
                       shuffled, out);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}
