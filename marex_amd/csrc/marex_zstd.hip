// marex_zstd.hip -- Zstandard frame decoder, host side, written from the format description (RFC 8878).
//
// The reference's test stores compress their small coordinate arrays with Blosc's zstd codec (`{"cname": "zstd", "shuffle": 2}`,
// the lat / lon arrays of sst_gridded.zarr; xr.open_zarr in examples/batch jobs/run_detect.py:55 reads them through numcodecs).
// Data arrays are LZ4 (marex_blosc.hip).  This decoder covers what a zstd encoder can emit without a dictionary: raw / RLE /
// compressed blocks, raw / RLE / Huffman literals (1 or 4 streams, FSE-compressed or direct weights, tree reuse), sequences
// with predefined / RLE / FSE / repeated tables, repeat offsets, skippable frames, concatenated frames; the content checksum is
// skipped, dictionaries are refused.  Every read is bounds-checked: malformed input returns an error code, never a fault.
#include "marex_common.hip.h"

namespace {

struct ZErr {};  // thrown on malformed input (caught at the C boundary)
#define ZCHECK(c)            \
    do {                     \
        if (!(c)) throw ZErr(); \
    } while (0)

inline int highest_set_bit(uint64_t v) {
    ZCHECK(v != 0);
    return 63 - __builtin_clzll(v);
}

// ---- forward bit reader (LSB first): FSE table descriptions
struct FwdBits {
    const uint8_t* p;
    size_t n;
    size_t bit = 0;
    uint32_t read(int bits) {
        ZCHECK(bits <= 24 && bit + (size_t)bits <= n * 8);
        uint32_t v = 0;
        for (int i = 0; i < bits; ++i, ++bit) v |= (uint32_t)((p[bit >> 3] >> (bit & 7)) & 1) << i;
        return v;
    }
    void rewind(int bits) { bit -= (size_t)bits; }
    size_t bytes_consumed() const { return (bit + 7) >> 3; }
};

// ---- backward bit reader: entropy-coded streams are written forward and read from their last byte, whose highest set bit
// marks the end; reads past the beginning yield zeros and leave the offset negative (how the format signals exhaustion)
struct BackBits {
    const uint8_t* p;
    int64_t offset;  // bits not yet consumed
    BackBits(const uint8_t* src, size_t len) : p(src) {
        ZCHECK(len > 0 && src[len - 1] != 0);
        offset = (int64_t)len * 8 - (8 - highest_set_bit(src[len - 1]));
    }
    uint64_t read(int bits) {
        ZCHECK(bits >= 0 && bits <= 56);
        offset -= bits;
        int64_t off = offset;
        int nb = bits;
        if (off < 0) {
            nb += (int)(off < -64 ? -64 : off);
            off = 0;
        }
        uint64_t v = 0;
        if (nb > 0) {
            const size_t byte0 = (size_t)(off >> 3);
            const int sh = (int)(off & 7);
            // up to 64 bits from byte0: nb + sh <= 63
            for (int i = 0; i * 8 < nb + sh; ++i) v |= (uint64_t)p[byte0 + i] << (8 * i);
            v = (v >> sh) & ((nb >= 64) ? ~0ull : ((1ull << nb) - 1));
        }
        if (offset < 0) v = (-offset >= 64) ? 0 : (v << (-offset));
        return v;
    }
};

// ---- FSE
struct FseTable {
    int al = 0;  // accuracy log; size = 1 << al
    std::vector<uint8_t> sym, nbits;
    std::vector<uint16_t> base;
    bool valid = false;
};

void fse_build(FseTable& t, const int16_t* freq, int nsym, int al) {
    ZCHECK(al >= 0 && al <= 15 && nsym >= 1 && nsym <= 256);
    const int size = 1 << al;
    t.al = al;
    t.sym.assign(size, 0);
    t.nbits.assign(size, 0);
    t.base.assign(size, 0);
    std::vector<uint16_t> next(nsym, 0);
    int high = size;
    for (int s = 0; s < nsym; ++s) {
        if (freq[s] == -1) {
            ZCHECK(high > 0);
            t.sym[--high] = (uint8_t)s;
            next[s] = 1;
        } else {
            next[s] = (uint16_t)freq[s];
        }
    }
    const int step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
    int pos = 0;
    for (int s = 0; s < nsym; ++s) {
        for (int i = 0; i < freq[s]; ++i) {
            t.sym[pos] = (uint8_t)s;
            do {
                pos = (pos + step) & mask;
            } while (pos >= high);
        }
    }
    ZCHECK(pos == 0);
    for (int i = 0; i < size; ++i) {
        const int s = t.sym[i];
        const uint16_t nx = next[s]++;
        ZCHECK(nx != 0);
        const int nb = al - highest_set_bit(nx);
        t.nbits[i] = (uint8_t)nb;
        t.base[i] = (uint16_t)(((uint32_t)nx << nb) - size);
    }
    t.valid = true;
}

// table description at the front of `src`; returns the bytes it occupies
size_t fse_read_table(FseTable& t, const uint8_t* src, size_t n, int max_al, int max_sym) {
    FwdBits in{src, n};
    const int al = 5 + (int)in.read(4);
    ZCHECK(al <= max_al);
    int remaining = 1 << al;
    int16_t freq[256];
    int s = 0;
    while (remaining > 0 && s <= max_sym) {
        const int bits = highest_set_bit((uint64_t)remaining + 1) + 1;
        uint32_t val = in.read(bits);
        const uint32_t lower_mask = (1u << (bits - 1)) - 1;
        const uint32_t threshold = (1u << bits) - 1 - (uint32_t)(remaining + 1);
        if ((val & lower_mask) < threshold) {
            in.rewind(1);
            val &= lower_mask;
        } else if (val > lower_mask) {
            val -= threshold;
        }
        const int proba = (int)val - 1;
        remaining -= proba < 0 ? -proba : proba;
        freq[s++] = (int16_t)proba;
        if (proba == 0) {
            int rep = (int)in.read(2);
            for (;;) {
                for (int i = 0; i < rep && s <= max_sym; ++i) freq[s++] = 0;
                if (rep == 3)
                    rep = (int)in.read(2);
                else
                    break;
            }
        }
    }
    ZCHECK(remaining == 0 && s <= max_sym + 1);
    fse_build(t, freq, s, al);
    return in.bytes_consumed();
}

void fse_rle(FseTable& t, uint8_t symbol) {
    t.al = 0;
    t.sym.assign(1, symbol);
    t.nbits.assign(1, 0);
    t.base.assign(1, 0);
    t.valid = true;
}

// ---- Huffman (literals)
struct HufTable {
    int max_bits = 0;
    std::vector<uint8_t> sym, nbits;
    bool valid = false;
};

void huf_build(HufTable& h, const uint8_t* bits, int nsym) {
    int max_bits = 0;
    int rank_count[17] = {0};
    for (int i = 0; i < nsym; ++i) {
        ZCHECK(bits[i] <= 11);
        max_bits = bits[i] > max_bits ? bits[i] : max_bits;
        rank_count[bits[i]]++;
    }
    ZCHECK(max_bits >= 1);
    const size_t size = (size_t)1 << max_bits;
    h.max_bits = max_bits;
    h.sym.assign(size, 0);
    h.nbits.assign(size, 0);
    uint32_t rank_idx[18];
    rank_idx[max_bits] = 0;
    for (int i = max_bits; i >= 1; --i) {
        rank_idx[i - 1] = rank_idx[i] + (uint32_t)rank_count[i] * (1u << (max_bits - i));
        ZCHECK(rank_idx[i - 1] <= size);
        for (uint32_t k = rank_idx[i]; k < rank_idx[i - 1]; ++k) h.nbits[k] = (uint8_t)i;
    }
    ZCHECK(rank_idx[0] == size);
    for (int i = 0; i < nsym; ++i) {
        if (bits[i] == 0) continue;
        const uint32_t code = rank_idx[bits[i]], len = 1u << (max_bits - bits[i]);
        ZCHECK(code + len <= size);
        for (uint32_t k = 0; k < len; ++k) h.sym[code + k] = (uint8_t)i;
        rank_idx[bits[i]] += len;
    }
    h.valid = true;
}

void huf_from_weights(HufTable& h, uint8_t* weights, int n) {
    ZCHECK(n >= 1 && n < 256);
    uint64_t sum = 0;
    for (int i = 0; i < n; ++i) {
        ZCHECK(weights[i] <= 11);
        sum += weights[i] ? (1ull << (weights[i] - 1)) : 0;
    }
    ZCHECK(sum != 0);
    const int max_bits = highest_set_bit(sum) + 1;
    const uint64_t left = (1ull << max_bits) - sum;
    ZCHECK((left & (left - 1)) == 0);  // the implied last weight completes a power of two
    uint8_t bits[256];
    weights[n] = (uint8_t)(highest_set_bit(left) + 1);
    for (int i = 0; i <= n; ++i) bits[i] = weights[i] ? (uint8_t)(max_bits + 1 - weights[i]) : 0;
    huf_build(h, bits, n + 1);
}

// tree description at the front of src; returns its size
size_t huf_read_tree(HufTable& h, const uint8_t* src, size_t n) {
    ZCHECK(n >= 1);
    const int hb = src[0];
    uint8_t weights[257];
    int nw = 0;
    size_t used = 1;
    if (hb >= 128) {
        nw = hb - 127;
        const size_t nbytes = (size_t)(nw + 1) / 2;
        ZCHECK(1 + nbytes <= n);
        for (int i = 0; i < nw; ++i) weights[i] = (i & 1) ? (src[1 + i / 2] & 0xF) : (src[1 + i / 2] >> 4);
        used += nbytes;
    } else {
        ZCHECK(hb >= 1 && (size_t)1 + hb <= n);
        FseTable t;
        const size_t th = fse_read_table(t, src + 1, (size_t)hb, 6, 12);
        ZCHECK(th < (size_t)hb);
        BackBits in(src + 1 + th, (size_t)hb - th);
        uint32_t s1 = (uint32_t)in.read(t.al), s2 = (uint32_t)in.read(t.al);
        for (;;) {
            ZCHECK(nw < 254);
            weights[nw++] = t.sym[s1];
            s1 = t.base[s1] + (uint32_t)in.read(t.nbits[s1]);
            if (in.offset < 0) {
                weights[nw++] = t.sym[s2];
                break;
            }
            weights[nw++] = t.sym[s2];
            s2 = t.base[s2] + (uint32_t)in.read(t.nbits[s2]);
            if (in.offset < 0) {
                weights[nw++] = t.sym[s1];
                break;
            }
        }
        used += (size_t)hb;
    }
    huf_from_weights(h, weights, nw);
    return used;
}

void huf_decode_stream(const HufTable& h, const uint8_t* src, size_t n, uint8_t* out, size_t count) {
    BackBits in(src, n);
    const uint32_t mask = (1u << h.max_bits) - 1;
    uint32_t state = (uint32_t)in.read(h.max_bits);
    for (size_t i = 0; i < count; ++i) {
        ZCHECK(in.offset > -(int64_t)h.max_bits);
        out[i] = h.sym[state];
        const int nb = h.nbits[state];
        state = ((state << nb) + (uint32_t)in.read(nb)) & mask;
    }
    ZCHECK(in.offset == -(int64_t)h.max_bits);  // the stream ends exactly where the last symbol does
}

// ---- sequences
const uint32_t LL_BASE[36] = {0,  1,  2,  3,  4,  5,  6,  7,  8,  9,   10,  11,  12,  13,   14,   15,   16,   18,
                              20, 22, 24, 28, 32, 40, 48, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536};
const uint8_t LL_BITS[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
const uint32_t ML_BASE[53] = {3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16, 17, 18, 19,  20,  21,  22,   23,   24,   25,   26,   27,    28,   29,
                              30, 31, 32, 33, 34, 35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099, 8195, 16387, 32771, 65539};
const uint8_t ML_BITS[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                             0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
const int16_t LL_DEFAULT[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
const int16_t ML_DEFAULT[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
const int16_t OF_DEFAULT[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};

struct FrameState {
    HufTable huf;
    FseTable ll, of, ml;
    uint64_t rep[3] = {1, 4, 8};
};

// one table of the sequences section according to its mode; returns bytes consumed
size_t seq_table(FseTable& t, int mode, const uint8_t* src, size_t n, const int16_t* dflt, int ndflt, int dflt_al, int max_al, int max_sym) {
    switch (mode) {
        case 0:
            fse_build(t, dflt, ndflt, dflt_al);
            return 0;
        case 1:
            ZCHECK(n >= 1 && src[0] <= max_sym);
            fse_rle(t, src[0]);
            return 1;
        case 2:
            return fse_read_table(t, src, n, max_al, max_sym);
        default:
            ZCHECK(t.valid);  // repeat: the table of the previous block
            return 0;
    }
}

void decode_block(FrameState& fs, const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t& out_pos) {
    // ---- literals
    ZCHECK(n >= 1);
    const int ltype = src[0] & 3, sfmt = (src[0] >> 2) & 3;
    std::vector<uint8_t> lit;
    size_t pos = 0;
    if (ltype == 0 || ltype == 1) {
        size_t regen, hdr;
        if ((sfmt & 1) == 0) {
            regen = src[0] >> 3;
            hdr = 1;
        } else if (sfmt == 1) {
            ZCHECK(n >= 2);
            regen = (size_t)(src[0] | (src[1] << 8)) >> 4;
            hdr = 2;
        } else {
            ZCHECK(n >= 3);
            regen = (size_t)(src[0] | (src[1] << 8) | (src[2] << 16)) >> 4;
            hdr = 3;
        }
        pos = hdr;
        if (ltype == 0) {
            ZCHECK(pos + regen <= n);
            lit.assign(src + pos, src + pos + regen);
            pos += regen;
        } else {
            ZCHECK(pos + 1 <= n);
            lit.assign(regen, src[pos]);
            pos += 1;
        }
    } else {
        size_t regen, comp, hdr;
        int nstreams = 4;
        if (sfmt == 0 || sfmt == 1) {
            ZCHECK(n >= 3);
            const uint32_t v = src[0] | (src[1] << 8) | (src[2] << 16);
            regen = (v >> 4) & 0x3FF;
            comp = (v >> 14) & 0x3FF;
            hdr = 3;
            if (sfmt == 0) nstreams = 1;
        } else if (sfmt == 2) {
            ZCHECK(n >= 4);
            const uint32_t v = src[0] | (src[1] << 8) | (src[2] << 16) | ((uint32_t)src[3] << 24);
            regen = (v >> 4) & 0x3FFF;
            comp = (v >> 18) & 0x3FFF;
            hdr = 4;
        } else {
            ZCHECK(n >= 5);
            const uint64_t v = (uint64_t)src[0] | ((uint64_t)src[1] << 8) | ((uint64_t)src[2] << 16) | ((uint64_t)src[3] << 24) | ((uint64_t)src[4] << 32);
            regen = (size_t)((v >> 4) & 0x3FFFF);
            comp = (size_t)((v >> 22) & 0x3FFFF);
            hdr = 5;
        }
        pos = hdr;
        ZCHECK(pos + comp <= n);
        const uint8_t* p = src + pos;
        size_t left = comp;
        if (ltype == 2) {
            const size_t tsz = huf_read_tree(fs.huf, p, left);
            p += tsz;
            left -= tsz;
        } else {
            ZCHECK(fs.huf.valid);  // treeless: the tree of the previous compressed-literals block
        }
        lit.resize(regen);
        if (nstreams == 1) {
            huf_decode_stream(fs.huf, p, left, lit.data(), regen);
        } else {
            ZCHECK(left >= 6);
            const size_t s1 = p[0] | (p[1] << 8), s2 = p[2] | (p[3] << 8), s3 = p[4] | (p[5] << 8);
            ZCHECK(6 + s1 + s2 + s3 < left);
            const size_t s4 = left - 6 - s1 - s2 - s3;
            const size_t per = (regen + 3) / 4;
            ZCHECK(3 * per <= regen);
            const uint8_t* q = p + 6;
            huf_decode_stream(fs.huf, q, s1, lit.data(), per);
            huf_decode_stream(fs.huf, q + s1, s2, lit.data() + per, per);
            huf_decode_stream(fs.huf, q + s1 + s2, s3, lit.data() + 2 * per, per);
            huf_decode_stream(fs.huf, q + s1 + s2 + s3, s4, lit.data() + 3 * per, regen - 3 * per);
        }
        pos += comp;
    }
    // ---- sequences
    ZCHECK(pos < n);
    size_t nseq = src[pos++];
    if (nseq >= 128) {
        if (nseq == 255) {
            ZCHECK(pos + 2 <= n);
            nseq = (size_t)src[pos] + ((size_t)src[pos + 1] << 8) + 0x7F00;
            pos += 2;
        } else {
            ZCHECK(pos + 1 <= n);
            nseq = ((nseq - 128) << 8) + src[pos];
            pos += 1;
        }
    }
    size_t lit_pos = 0;
    auto put_literals = [&](size_t len) {
        ZCHECK(lit_pos + len <= lit.size() && out_pos + len <= cap);
        memcpy(dst + out_pos, lit.data() + lit_pos, len);
        lit_pos += len;
        out_pos += len;
    };
    if (nseq == 0) {
        ZCHECK(pos == n);
        put_literals(lit.size());
        return;
    }
    ZCHECK(pos < n);
    const int modes = src[pos++];
    ZCHECK((modes & 3) == 0);
    pos += seq_table(fs.ll, (modes >> 6) & 3, src + pos, n - pos, LL_DEFAULT, 36, 6, 9, 35);
    pos += seq_table(fs.of, (modes >> 4) & 3, src + pos, n - pos, OF_DEFAULT, 29, 5, 8, 31);
    pos += seq_table(fs.ml, (modes >> 2) & 3, src + pos, n - pos, ML_DEFAULT, 53, 6, 9, 52);
    ZCHECK(pos < n);
    BackBits in(src + pos, n - pos);
    uint32_t sl = (uint32_t)in.read(fs.ll.al), so = (uint32_t)in.read(fs.of.al), sm = (uint32_t)in.read(fs.ml.al);
    for (size_t i = 0; i < nseq; ++i) {
        const int oc = fs.of.sym[so], lc = fs.ll.sym[sl], mc = fs.ml.sym[sm];
        ZCHECK(oc <= 31 && lc <= 35 && mc <= 52);
        const uint64_t ov = (1ull << oc) + in.read(oc);
        const size_t mlen = ML_BASE[mc] + (size_t)in.read(ML_BITS[mc]);
        const size_t llen = LL_BASE[lc] + (size_t)in.read(LL_BITS[lc]);
        if (i + 1 < nseq) {
            sl = fs.ll.base[sl] + (uint32_t)in.read(fs.ll.nbits[sl]);
            sm = fs.ml.base[sm] + (uint32_t)in.read(fs.ml.nbits[sm]);
            so = fs.of.base[so] + (uint32_t)in.read(fs.of.nbits[so]);
        }
        ZCHECK(in.offset >= 0);
        uint64_t offset;
        if (ov > 3) {
            offset = ov - 3;
            fs.rep[2] = fs.rep[1];
            fs.rep[1] = fs.rep[0];
            fs.rep[0] = offset;
        } else {
            unsigned idx = (unsigned)ov - 1 + (llen == 0 ? 1 : 0);
            if (idx == 0) {
                offset = fs.rep[0];
            } else {
                offset = idx < 3 ? fs.rep[idx] : fs.rep[0] - 1;
                ZCHECK(offset != 0);
                if (idx > 1) fs.rep[2] = fs.rep[1];
                fs.rep[1] = fs.rep[0];
                fs.rep[0] = offset;
            }
        }
        put_literals(llen);
        ZCHECK(offset <= out_pos && out_pos + mlen <= cap);
        const uint8_t* from = dst + out_pos - offset;
        for (size_t k = 0; k < mlen; ++k) dst[out_pos + k] = from[k];  // may overlap: byte by byte
        out_pos += mlen;
    }
    ZCHECK(in.offset == 0);
    put_literals(lit.size() - lit_pos);
}

size_t decode_frames(const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
    size_t pos = 0, out_pos = 0;
    while (pos < n) {
        ZCHECK(pos + 4 <= n);
        const uint32_t magic = src[pos] | (src[pos + 1] << 8) | (src[pos + 2] << 16) | ((uint32_t)src[pos + 3] << 24);
        pos += 4;
        if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {  // skippable frame
            ZCHECK(pos + 4 <= n);
            const uint32_t sz = src[pos] | (src[pos + 1] << 8) | (src[pos + 2] << 16) | ((uint32_t)src[pos + 3] << 24);
            pos += 4;
            ZCHECK(pos + sz <= n);
            pos += sz;
            continue;
        }
        ZCHECK(magic == 0xFD2FB528u);
        ZCHECK(pos < n);
        const int fhd = src[pos++];
        const int fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, checksum = (fhd >> 2) & 1, did_flag = fhd & 3;
        ZCHECK((fhd & 0x08) == 0);
        if (!single) {
            ZCHECK(pos < n);
            pos += 1;  // window descriptor: the output buffer is the window
        }
        static const int did_bytes[4] = {0, 1, 2, 4};
        uint32_t did = 0;
        ZCHECK(pos + (size_t)did_bytes[did_flag] <= n);
        for (int i = 0; i < did_bytes[did_flag]; ++i) did |= (uint32_t)src[pos + i] << (8 * i);
        pos += (size_t)did_bytes[did_flag];
        ZCHECK(did == 0);  // dictionaries are not supported
        const int fcs_bytes = fcs_flag == 0 ? (single ? 1 : 0) : (fcs_flag == 1 ? 2 : (fcs_flag == 2 ? 4 : 8));
        ZCHECK(pos + (size_t)fcs_bytes <= n);
        uint64_t fcs = 0;
        for (int i = 0; i < fcs_bytes; ++i) fcs |= (uint64_t)src[pos + i] << (8 * i);
        if (fcs_bytes == 2) fcs += 256;
        pos += (size_t)fcs_bytes;
        const size_t frame_start = out_pos;
        FrameState fs;
        // matches may reach back to the beginning of THIS frame only: decode into a view that starts there
        for (;;) {
            ZCHECK(pos + 3 <= n);
            const uint32_t bh = src[pos] | (src[pos + 1] << 8) | (src[pos + 2] << 16);
            pos += 3;
            const int last = bh & 1, type = (bh >> 1) & 3;
            const size_t bsize = bh >> 3;
            size_t rel = out_pos - frame_start;
            if (type == 0) {
                ZCHECK(pos + bsize <= n && out_pos + bsize <= cap);
                memcpy(dst + out_pos, src + pos, bsize);
                out_pos += bsize;
                pos += bsize;
            } else if (type == 1) {
                ZCHECK(pos + 1 <= n && out_pos + bsize <= cap);
                memset(dst + out_pos, src[pos], bsize);
                out_pos += bsize;
                pos += 1;
            } else {
                ZCHECK(type == 2 && pos + bsize <= n);
                decode_block(fs, src + pos, bsize, dst + frame_start, cap - frame_start, rel);
                out_pos = frame_start + rel;
                pos += bsize;
            }
            if (last) break;
        }
        if (fcs_bytes) ZCHECK(out_pos - frame_start == fcs);
        if (checksum) {
            ZCHECK(pos + 4 <= n);
            pos += 4;  // XXH64 of the content, low 32 bits: not verified
        }
    }
    return out_pos;
}

}  // namespace

extern "C" int marex_zstd_decompress_h(const uint8_t* src, int64_t n, uint8_t* dst, int64_t cap, int64_t* out_n) {
    if (!src || !dst || !out_n || n < 0 || cap < 0) return -1;
    try {
        *out_n = (int64_t)decode_frames(src, (size_t)n, dst, (size_t)cap);
        return 0;
    } catch (const ZErr&) {
        return -5;  // malformed or unsupported stream
    } catch (const std::bad_alloc&) {
        return -2;
    }
}
