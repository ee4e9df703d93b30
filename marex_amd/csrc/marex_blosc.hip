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
inline void wr32(unsigned char* p, unsigned v) {
    p[0] = (unsigned char)v;
    p[1] = (unsigned char)(v >> 8);
    p[2] = (unsigned char)(v >> 16);
    p[3] = (unsigned char)(v >> 24);
}

// Blosc splits a block into `typesize` streams when this holds (c-blosc 1.x blosc_d, also the rule of the releases
// that predate the "dont_split" flag -- frames written this way decode everywhere)
inline bool blosc_splits(unsigned flags, long typesize, long blocksize, bool leftover) {
    return !(flags & 0x10) && !leftover && typesize <= 16 && blocksize / typesize >= 128;
}

// LZ4 block compressor: greedy, one hash probe per position, matches extended eight bytes at a time.  Honours the
// format's end-of-block rules (the last match starts at least 12 bytes before the end, the last 5 bytes are
// literals).  Returns the compressed size, or -1 when the output would not fit in `cap` bytes.
long lz4_block_encode(const unsigned char* src, long n, unsigned char* dst, long cap) {
    constexpr int HLOG = 13;
    int table[1 << HLOG];
    for (int i = 0; i < (1 << HLOG); ++i) table[i] = -1;
    unsigned char* op = dst;
    unsigned char* const oend = dst + cap;
    long anchor = 0, ip = 0;
    auto emit = [&](long lit, long mlen, long offset) -> bool {  // mlen == 0: final literals
        const long need = 1 + lit / 255 + 1 + lit + (mlen ? 2 + mlen / 255 + 1 : 0);
        if (need > oend - op) return false;
        unsigned char* token = op++;
        *token = (unsigned char)((lit < 15 ? lit : 15) << 4);
        if (lit >= 15) {
            long r = lit - 15;
            for (; r >= 255; r -= 255) *op++ = 255;
            *op++ = (unsigned char)r;
        }
        memcpy(op, src + anchor, (size_t)lit);
        op += lit;
        if (mlen) {
            *op++ = (unsigned char)offset;
            *op++ = (unsigned char)(offset >> 8);
            const long m = mlen - 4;
            *token |= (unsigned char)(m < 15 ? m : 15);
            if (m >= 15) {
                long r = m - 15;
                for (; r >= 255; r -= 255) *op++ = 255;
                *op++ = (unsigned char)r;
            }
        }
        return true;
    };
    if (n >= 13) {
        const long mflimit = n - 12, matchlimit = n - 5;
        long misses = 0;
        while (ip <= mflimit) {
            const unsigned v = rd32(src + ip);
            const unsigned h = (v * 2654435761u) >> (32 - HLOG);
            const long cand = table[h];
            table[h] = (int)ip;
            if (cand < 0 || ip - cand > 65535 || rd32(src + cand) != v) {
                ip += 1 + (misses++ >> 6);  // incompressible stretches: widen the step
                continue;
            }
            misses = 0;
            long s = ip, c = cand;
            while (s > anchor && c > 0 && src[s - 1] == src[c - 1]) {  // extend backwards over pending literals
                --s;
                --c;
            }
            long e = ip + 4, ce = cand + 4;
            while (e + 8 <= matchlimit) {
                unsigned long long a, b;
                memcpy(&a, src + e, 8);
                memcpy(&b, src + ce, 8);
                if (a != b) {
                    const long same = __builtin_ctzll(a ^ b) >> 3;
                    e += same;
                    ce += same;
                    goto extended;
                }
                e += 8;
                ce += 8;
            }
            while (e < matchlimit && src[e] == src[ce]) {
                ++e;
                ++ce;
            }
        extended:
            if (!emit(s - anchor, e - s, s - c)) return -1;
            anchor = ip = e;
            if (ip - 2 > cand && ip - 2 <= mflimit) table[(rd32(src + ip - 2) * 2654435761u) >> (32 - HLOG)] = (int)(ip - 2);
        }
    }
    if (!emit(n - anchor, 0, 0)) return -1;
    return (long)(op - dst);
}

}  // namespace

// Compress `nbytes` bytes into one Blosc-1 frame (LZ4 codec, byte shuffle when shuffle != 0 and typesize > 1) -- what the
// reference's `Dataset.to_zarr` writes through numcodecs' default Blosc compressor.  blocksize <= 0 picks 256 KiB.  Data
// that LZ4 cannot shrink is stored (per stream, or the whole frame as a "memcpyed" frame), so dstcap >= nbytes + 16 always
// suffices.  0 = OK and *out_len = frame bytes; -1 bad argument, -4 destination too small.
extern "C" int marex_blosc_compress_h(const void* src_v, int64_t nbytes, int typesize, int shuffle, int64_t blocksize,
                                      void* dst_v, int64_t dstcap, int64_t* out_len) {
    if (!src_v || !dst_v || !out_len || nbytes < 0 || nbytes > 0x7fffffff - 16 || typesize < 1) return -1;
    if (dstcap < nbytes + 16) return -4;
    if (typesize > 255) typesize = 1;  // as Blosc: such items are treated as bytes
    const unsigned char* src = static_cast<const unsigned char*>(src_v);
    unsigned char* dst = static_cast<unsigned char*>(dst_v);
    if (blocksize <= 0) blocksize = 256 * 1024;
    if (blocksize > nbytes && nbytes > 0) blocksize = nbytes;
    if (blocksize > typesize) blocksize -= blocksize % typesize;  // whole elements per block; a ragged tail becomes the leftover block
    const bool do_shuffle = shuffle && typesize > 1;
    unsigned flags = (1u << 5) | (do_shuffle ? 0x1u : 0u);
    dst[0] = 2;  // Blosc format version
    dst[1] = 1;  // LZ4 format version
    dst[3] = (unsigned char)typesize;
    wr32(dst + 4, (unsigned)nbytes);
    wr32(dst + 8, (unsigned)blocksize);
    auto stored_frame = [&]() {
        dst[2] = (unsigned char)(flags | 0x2u);
        memcpy(dst + 16, src, (size_t)nbytes);
        wr32(dst + 12, (unsigned)(nbytes + 16));
        *out_len = nbytes + 16;
        return 0;
    };
    if (nbytes == 0) return stored_frame();
    const long nblocks = (long)((nbytes + blocksize - 1) / blocksize);
    long pos = 16 + 4 * nblocks;
    if (pos >= dstcap) return stored_frame();
    std::vector<unsigned char> tmp(do_shuffle ? (size_t)blocksize : 0);
    for (long j = 0; j < nblocks; ++j) {
        const long bsize = (j == nblocks - 1 && nbytes % blocksize) ? (long)(nbytes % blocksize) : (long)blocksize;
        const bool leftover = bsize != blocksize;
        const unsigned char* in = src + j * blocksize;
        if (do_shuffle) {  // plane k holds byte k of every element; a tail shorter than one element is copied
            const long ne = bsize / typesize;
            for (long k = 0; k < typesize; ++k) {
                unsigned char* plane = tmp.data() + k * ne;
                for (long i = 0; i < ne; ++i) plane[i] = in[i * typesize + k];
            }
            const long rest = bsize - ne * typesize;
            if (rest) memcpy(tmp.data() + ne * typesize, in + ne * typesize, (size_t)rest);
            in = tmp.data();
        }
        const long nsplits = blosc_splits(flags, typesize, (long)blocksize, leftover) && bsize % typesize == 0 ? typesize : 1;
        const long neblock = bsize / nsplits;
        wr32(dst + 16 + 4 * j, (unsigned)pos);
        for (long s = 0; s < nsplits; ++s) {
            // the frame may not grow past nbytes + 16: a stream that does not fit is the signal to store everything
            const long room = (long)(nbytes + 16) - pos - 4;
            long cb = room > 0 ? lz4_block_encode(in + s * neblock, neblock, dst + pos + 4, room < neblock - 1 ? room : neblock - 1) : -1;
            if (cb < 0) {  // not compressible: stored stream, marked by cbytes == raw size
                if (neblock > room) return stored_frame();
                memcpy(dst + pos + 4, in + s * neblock, (size_t)neblock);
                cb = neblock;
            }
            wr32(dst + pos, (unsigned)cb);
            pos += 4 + cb;
        }
    }
    dst[2] = (unsigned char)flags;
    wr32(dst + 12, (unsigned)pos);
    *out_len = pos;
    return 0;
}

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
    const long nblocks = (nbytes + blocksize - 1) / blocksize;
    if (16 + 4 * nblocks > srclen) return -5;
    std::vector<unsigned char> tmp((size_t)blocksize);
    for (long j = 0; j < nblocks; ++j) {
        const long bsize = (j == nblocks - 1 && nbytes % blocksize) ? nbytes % blocksize : blocksize;
        const bool leftover = bsize != blocksize;
        const long nsplits = (blosc_splits(flags, typesize, blocksize, leftover) && bsize % typesize == 0) ? typesize : 1;
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
