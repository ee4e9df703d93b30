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
