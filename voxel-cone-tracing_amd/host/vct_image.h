// vct_image.h -- decoders for the image containers an MTL's map_Kd / map_Ks / map_bump usually name.
//
// The reference decodes material maps with stb_image (R/Model.h:141-226 -> stbi_load: JPEG, PNG, TGA, BMP, ...); its
// sources are not used here.  Written from the published formats:
//   PNG   (ISO/IEC 15948): 8- and 16-bit grey / grey+alpha / RGB / RGBA / palette (+ tRNS), non-interlaced and Adam7;
//         zlib / DEFLATE (RFC 1950 / 1951) inflated by the small decoder below -- no libz, no libpng
//   JPEG  (ITU-T T.81) baseline sequential DCT, Huffman, 8-bit, 1 or 3 components, any sampling factors up to 2x2,
//         restart intervals; progressive and arithmetic-coded files are refused
//   BMP   uncompressed 24 / 32 bpp (BITMAPINFOHEADER and later), bottom-up or top-down
//   TGA   true-colour and grey, raw or run-length encoded (types 2, 3, 10, 11), 8 / 24 / 32 bpp
//   PPM   binary P6 / PGM P5, maxval 255
// Output: RGBA8, rows BOTTOM-UP (row 0 at v = 0) -- what the reference's aiProcess_FlipUVs + top-down stb rows amount to.
// Everything is bounds-checked: a truncated or corrupt file makes the decoder return false, never read outside the buffer.
#ifndef VCT_IMAGE_H_
#define VCT_IMAGE_H_

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

namespace vct_image {

struct Image {
    int w = 0, h = 0;
    std::vector<uint8_t> rgba;      // [h][w][4], row 0 = bottom row
};

// ---------------------------------------------------------------------------------------------- DEFLATE -------------
class Inflater {
public:
    Inflater(const uint8_t* p, size_t n) : p_(p), n_(n) {}
    bool run(std::vector<uint8_t>& out, size_t limit) {
        // zlib header (RFC 1950): CM = 8, no preset dictionary
        if (n_ < 2 || (p_[0] & 15) != 8 || ((p_[0] << 8 | p_[1]) % 31) != 0 || (p_[1] & 32)) return false;
        pos_ = 2;
        for (;;) {
            const int last = (int)bits(1), type = (int)bits(2);
            if (bad_) return false;
            if (type == 0) {
                bitbuf_ = 0; bitcnt_ = 0;                                  // to the byte boundary
                if (pos_ + 4 > n_) return false;
                const uint32_t len = p_[pos_] | p_[pos_ + 1] << 8, nlen = p_[pos_ + 2] | p_[pos_ + 3] << 8;
                pos_ += 4;
                if ((len ^ nlen) != 0xffffu || pos_ + len > n_ || out.size() + len > limit) return false;
                out.insert(out.end(), p_ + pos_, p_ + pos_ + len);
                pos_ += len;
            } else if (type == 1 || type == 2) {
                Huff lit, dist;
                if (type == 1) {
                    uint8_t l[288], d[30];
                    for (int i = 0; i < 288; ++i) l[i] = i < 144 ? 8 : (i < 256 ? 9 : (i < 280 ? 7 : 8));
                    for (int i = 0; i < 30; ++i) d[i] = 5;
                    if (!lit.build(l, 288) || !dist.build(d, 30)) return false;
                } else if (!dynamic(lit, dist)) return false;
                if (!block(lit, dist, out, limit)) return false;
            } else return false;
            if (last) return !bad_;
        }
    }

private:
    struct Huff {
        uint16_t count[16], symbol[288];
        bool build(const uint8_t* len, int n) {
            memset(count, 0, sizeof(count));
            for (int i = 0; i < n; ++i) ++count[len[i]];
            int left = 1;
            for (int l = 1; l < 16; ++l) { left = (left << 1) - count[l]; if (left < 0) return false; }   // over-subscribed
            uint16_t offs[16];
            offs[1] = 0;
            for (int l = 1; l < 15; ++l) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
            for (int i = 0; i < n; ++i) if (len[i]) symbol[offs[len[i]]++] = (uint16_t)i;
            return true;
        }
    };
    uint32_t bits(int need) {
        while (bitcnt_ < need) {
            if (pos_ >= n_) { bad_ = true; return 0; }
            bitbuf_ |= (uint32_t)p_[pos_++] << bitcnt_;
            bitcnt_ += 8;
        }
        const uint32_t v = bitbuf_ & ((1u << need) - 1u);
        bitbuf_ >>= need; bitcnt_ -= need;
        return v;
    }
    int decode(const Huff& h) {
        int code = 0, first = 0, index = 0;
        for (int l = 1; l < 16; ++l) {
            code |= (int)bits(1);
            if (bad_) return -1;
            const int c = h.count[l];
            if (code - c < first) return h.symbol[index + (code - first)];
            index += c; first += c; first <<= 1; code <<= 1;
        }
        return -1;
    }
    bool dynamic(Huff& lit, Huff& dist) {
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        const int nl = (int)bits(5) + 257, nd = (int)bits(5) + 1, nc = (int)bits(4) + 4;
        if (bad_ || nl > 286 || nd > 30) return false;
        uint8_t cl[19] = {0};
        for (int i = 0; i < nc; ++i) cl[order[i]] = (uint8_t)bits(3);
        Huff ch;
        if (bad_ || !ch.build(cl, 19)) return false;
        uint8_t len[286 + 30];
        int i = 0;
        while (i < nl + nd) {
            const int s = decode(ch);
            if (s < 0) return false;
            if (s < 16) len[i++] = (uint8_t)s;
            else {
                int rep, val = 0;
                if (s == 16) { if (!i) return false; val = len[i - 1]; rep = 3 + (int)bits(2); }
                else if (s == 17) rep = 3 + (int)bits(3);
                else rep = 11 + (int)bits(7);
                if (bad_ || i + rep > nl + nd) return false;
                while (rep--) len[i++] = (uint8_t)val;
            }
        }
        return len[256] != 0 && lit.build(len, nl) && dist.build(len + nl, nd);
    }
    bool block(const Huff& lit, const Huff& dist, std::vector<uint8_t>& out, size_t limit) {
        static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
        static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
        for (;;) {
            const int s = decode(lit);
            if (s < 0) return false;
            if (s < 256) { if (out.size() >= limit) return false; out.push_back((uint8_t)s); }
            else if (s == 256) return true;
            else {
                if (s > 285) return false;
                const int len = lbase[s - 257] + (int)bits(lext[s - 257]);
                const int ds = decode(dist);
                if (ds < 0 || ds > 29) return false;
                const size_t d = dbase[ds] + bits(dext[ds]);
                if (bad_ || d > out.size() || out.size() + (size_t)len > limit) return false;
                const size_t from = out.size() - d;
                for (int k = 0; k < len; ++k) out.push_back(out[from + (size_t)k]);
            }
        }
    }
    const uint8_t* p_;
    size_t n_, pos_ = 0;
    uint32_t bitbuf_ = 0;
    int bitcnt_ = 0;
    bool bad_ = false;
};

inline void put(Image& im, int x, int y_top_down, uint8_t r, uint8_t g, uint8_t b, uint8_t a) {
    uint8_t* d = &im.rgba[((size_t)(im.h - 1 - y_top_down) * im.w + x) * 4];
    d[0] = r; d[1] = g; d[2] = b; d[3] = a;
}
// the size limits every decoder applies BEFORE it sizes anything from header fields
inline bool dims_ok(int w, int h) { return w > 0 && h > 0 && w <= 32768 && h <= 32768 && (size_t)w * h <= ((size_t)1 << 28); }
inline bool alloc(Image& im, int w, int h) {
    if (!dims_ok(w, h)) return false;
    im.w = w; im.h = h;
    im.rgba.assign((size_t)w * h * 4, 0);
    return true;
}

// ------------------------------------------------------------------------------------------------- PNG --------------
inline uint32_t be32(const uint8_t* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

inline bool decode_png(const std::vector<uint8_t>& f, Image& im) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    if (f.size() < 8 + 25 || memcmp(f.data(), sig, 8) != 0) return false;
    size_t pos = 8;
    int w = 0, h = 0, depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    bool end = false;
    while (!end && pos + 12 <= f.size()) {
        const uint32_t len = be32(&f[pos]);
        if (pos + 12 + (size_t)len > f.size()) return false;
        const uint8_t* type = &f[pos + 4];
        const uint8_t* data = &f[pos + 8];
        if (!memcmp(type, "IHDR", 4)) {
            if (len != 13) return false;
            w = (int)be32(data); h = (int)be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12];
            if (data[10] != 0 || data[11] != 0 || interlace > 1) return false;
        } else if (!memcmp(type, "PLTE", 4)) plte.assign(data, data + len);
        else if (!memcmp(type, "tRNS", 4)) trns.assign(data, data + len);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!memcmp(type, "IEND", 4)) end = true;
        pos += 12 + (size_t)len;
    }
    int chans;
    switch (ctype) { case 0: chans = 1; break; case 2: chans = 3; break; case 3: chans = 1; break; case 4: chans = 2; break; case 6: chans = 4; break; default: return false; }
    const bool depth_ok = ctype == 3 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8)
                                     : (ctype == 0 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16) : (depth == 8 || depth == 16));
    if (!depth_ok || idat.empty() || (ctype == 3 && plte.size() < 3) || !dims_ok(w, h)) return false;
    const int bpp_bits = chans * depth, bpp = (bpp_bits + 7) / 8;
    // passes: one for a non-interlaced image, seven for Adam7 (x0, y0, dx, dy)
    static const int adam[7][4] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    size_t raw_size = 0;
    for (int p = 0; p < (interlace ? 7 : 1); ++p) {
        const int pw = interlace ? (w - adam[p][0] + adam[p][2] - 1) / adam[p][2] : w;
        const int ph = interlace ? (h - adam[p][1] + adam[p][3] - 1) / adam[p][3] : h;
        if (pw > 0 && ph > 0) raw_size += ((size_t)((size_t)pw * bpp_bits + 7) / 8 + 1) * ph;
    }
    // DEFLATE expands by at most 1032 : 1 (a 258-byte match per 2 bits): a header that promises more than the IDAT bytes
    // can hold is refused before anything is sized from it (a 100-byte file must not reserve gigabytes)
    if (raw_size > idat.size() * 1032 + 1024 || !alloc(im, w, h)) return false;
    std::vector<uint8_t> raw;
    raw.reserve(raw_size);
    if (!Inflater(idat.data(), idat.size()).run(raw, raw_size) || raw.size() != raw_size) return false;
    auto sample = [&](const uint8_t* row, int x, int c) -> int {       // sample c of pixel x, scaled to 8 bits (palette: the index)
        if (depth == 8) return row[(size_t)x * chans + c];
        if (depth == 16) return row[((size_t)x * chans + c) * 2];
        const int per = 8 / depth, v = (row[x / per] >> ((per - 1 - x % per) * depth)) & ((1 << depth) - 1);
        return ctype == 3 ? v : v * 255 / ((1 << depth) - 1);
    };
    auto sample16 = [&](const uint8_t* row, int x, int c) -> int { const uint8_t* q = row + ((size_t)x * chans + c) * 2; return q[0] << 8 | q[1]; };
    size_t off = 0;
    std::vector<uint8_t> prev, cur;
    for (int p = 0; p < (interlace ? 7 : 1); ++p) {
        const int x0 = interlace ? adam[p][0] : 0, y0 = interlace ? adam[p][1] : 0;
        const int dx = interlace ? adam[p][2] : 1, dy = interlace ? adam[p][3] : 1;
        const int pw = (w - x0 + dx - 1) / dx, ph = (h - y0 + dy - 1) / dy;
        if (pw <= 0 || ph <= 0) continue;
        const size_t stride = ((size_t)pw * bpp_bits + 7) / 8;
        prev.assign(stride, 0);
        cur.resize(stride);
        for (int y = 0; y < ph; ++y) {
            const int ft = raw[off++];
            const uint8_t* in = &raw[off];
            off += stride;
            for (size_t i = 0; i < stride; ++i) {
                const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= (size_t)bpp ? prev[i - bpp] : 0;
                int pr;
                switch (ft) {
                    case 0: pr = 0; break;
                    case 1: pr = a; break;
                    case 2: pr = b; break;
                    case 3: pr = (a + b) >> 1; break;
                    case 4: { const int q = a + b - c, pa = abs(q - a), pb = abs(q - b), pc = abs(q - c); pr = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
                    default: return false;
                }
                cur[i] = (uint8_t)(in[i] + pr);
            }
            for (int x = 0; x < pw; ++x) {
                uint8_t r, g, bl, al = 255;
                if (ctype == 3) {
                    const size_t idx = (size_t)sample(cur.data(), x, 0);
                    if (idx * 3 + 2 >= plte.size()) return false;
                    r = plte[idx * 3]; g = plte[idx * 3 + 1]; bl = plte[idx * 3 + 2];
                    if (idx < trns.size()) al = trns[idx];
                } else if (ctype == 0 || ctype == 4) {
                    r = g = bl = (uint8_t)sample(cur.data(), x, 0);
                    if (ctype == 4) al = (uint8_t)sample(cur.data(), x, 1);
                    else if (trns.size() >= 2) {        // the one transparent grey level, compared at full depth
                        const int key = trns[0] << 8 | trns[1];
                        const int v = depth == 16 ? sample16(cur.data(), x, 0)
                                                  : (depth == 8 ? cur[x] : ((cur[x / (8 / depth)] >> ((8 / depth - 1 - x % (8 / depth)) * depth)) & ((1 << depth) - 1)));
                        if (v == key) al = 0;
                    }
                } else {
                    r = (uint8_t)sample(cur.data(), x, 0); g = (uint8_t)sample(cur.data(), x, 1); bl = (uint8_t)sample(cur.data(), x, 2);
                    if (ctype == 6) al = (uint8_t)sample(cur.data(), x, 3);
                    else if (trns.size() >= 6) {
                        const int kr = trns[0] << 8 | trns[1], kg = trns[2] << 8 | trns[3], kb = trns[4] << 8 | trns[5];
                        const bool hit = depth == 16 ? (sample16(cur.data(), x, 0) == kr && sample16(cur.data(), x, 1) == kg && sample16(cur.data(), x, 2) == kb)
                                                     : (r == kr && g == kg && bl == kb);
                        if (hit) al = 0;
                    }
                }
                put(im, x0 + x * dx, y0 + y * dy, r, g, bl, al);
            }
            prev.swap(cur);
        }
    }
    return true;
}

// ------------------------------------------------------------------------------------------------ JPEG --------------
// Baseline sequential (SOF0; SOF1 with 8-bit samples decodes the same way).  IDCT in float (the separable form of
// T.81 A.3.3), YCbCr -> RGB per JFIF, chroma up-sampled by replication.  Decoders differ in the last bit of both --
// like the reference's stb_image does from libjpeg.
class Jpeg {
public:
    explicit Jpeg(const std::vector<uint8_t>& f) : f_(f) {}
    bool decode(Image& im) {
        if (f_.size() < 4 || f_[0] != 0xff || f_[1] != 0xd8) return false;
        size_t pos = 2;
        while (pos + 4 <= f_.size()) {
            if (f_[pos] != 0xff) return false;
            const int m = f_[pos + 1];
            if (m == 0xff) { ++pos; continue; }
            pos += 2;
            if (m == 0xd8 || (m >= 0xd0 && m <= 0xd7) || m == 0x01) continue;
            if (m == 0xd9) break;
            if (pos + 2 > f_.size()) return false;
            const size_t len = (size_t)f_[pos] << 8 | f_[pos + 1];
            if (len < 2 || pos + len > f_.size()) return false;
            const uint8_t* d = &f_[pos + 2];
            const size_t n = len - 2;
            if (m == 0xdb) { if (!dqt(d, n)) return false; }
            else if (m == 0xc4) { if (!dht(d, n)) return false; }
            else if (m == 0xc0 || m == 0xc1) { if (!sof(d, n)) return false; }
            else if (m == 0xc2 || (m >= 0xc3 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc)) return false;   // progressive / lossless / arithmetic
            else if (m == 0xdd) { if (n < 2) return false; restart_ = d[0] << 8 | d[1]; }
            else if (m == 0xda) {
                if (!sos(d, n)) return false;
                return scan(pos + len, im);
            }
            pos += len;
        }
        return false;
    }

private:
    struct Comp { int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, pred = 0; std::vector<uint8_t> plane; int pw = 0, ph = 0; };
    struct HT { uint8_t bits[17] = {0}; uint8_t vals[256] = {0}; int mincode[17], maxcode[18], valptr[17]; bool set = false; };
    bool dqt(const uint8_t* d, size_t n) {
        size_t p = 0;
        while (p < n) {
            const int pq = d[p] >> 4, tq = d[p] & 15;
            ++p;
            if (tq > 3 || p + (pq ? 128 : 64) > n) return false;
            for (int i = 0; i < 64; ++i) { q_[tq][i] = pq ? (d[p] << 8 | d[p + 1]) : d[p]; p += pq ? 2 : 1; }
        }
        return true;
    }
    bool dht(const uint8_t* d, size_t n) {
        size_t p = 0;
        while (p < n) {
            if (p + 17 > n) return false;
            const int tc = d[p] >> 4, th = d[p] & 15;
            if (tc > 1 || th > 3) return false;
            HT& t = ht_[tc][th];
            int total = 0;
            for (int i = 1; i <= 16; ++i) { t.bits[i] = d[p + i]; total += t.bits[i]; }
            p += 17;
            if (total > 256 || p + (size_t)total > n) return false;
            memcpy(t.vals, d + p, (size_t)total);
            p += (size_t)total;
            int code = 0, k = 0;
            for (int l = 1; l <= 16; ++l) {
                t.valptr[l] = k; t.mincode[l] = code;
                code += t.bits[l]; k += t.bits[l];
                t.maxcode[l] = t.bits[l] ? code - 1 : -1;
                code <<= 1;
            }
            t.maxcode[17] = 0x7fffffff;
            t.set = true;
        }
        return true;
    }
    bool sof(const uint8_t* d, size_t n) {
        if (n < 6 || d[0] != 8) return false;
        h_ = d[1] << 8 | d[2]; w_ = d[3] << 8 | d[4];
        const int nc = d[5];
        if ((nc != 1 && nc != 3) || n < 6 + (size_t)nc * 3 || !dims_ok(w_, h_)) return false;     // (the planes are sized from these)
        comp_.assign((size_t)nc, Comp());
        hmax_ = vmax_ = 1;
        for (int i = 0; i < nc; ++i) {
            Comp& c = comp_[(size_t)i];
            c.id = d[6 + i * 3]; c.h = d[7 + i * 3] >> 4; c.v = d[7 + i * 3] & 15; c.tq = d[8 + i * 3];
            if (c.h < 1 || c.h > 2 || c.v < 1 || c.v > 2 || c.tq > 3) return false;
            if (nc == 1) c.h = c.v = 1;      // T.81 A.2.2: a single-component scan is never interleaved -- one 8 x 8 block per MCU whatever the factors say
            if (c.h > hmax_) hmax_ = c.h;
            if (c.v > vmax_) vmax_ = c.v;
        }
        return true;
    }
    bool sos(const uint8_t* d, size_t n) {
        if (comp_.empty() || n < 1 || d[0] != comp_.size() || n < 1 + (size_t)d[0] * 2 + 3) return false;
        for (size_t i = 0; i < comp_.size(); ++i) {
            Comp* c = nullptr;
            for (auto& k : comp_) if (k.id == d[1 + i * 2]) c = &k;
            if (!c) return false;
            c->td = d[2 + i * 2] >> 4; c->ta = d[2 + i * 2] & 15;
            if (c->td > 3 || c->ta > 3 || !ht_[0][c->td].set || !ht_[1][c->ta].set) return false;
        }
        return true;
    }
    // entropy-coded segment reader: 0xff00 is a stuffed 0xff, any other marker ends the data (zeros are fed beyond it)
    int bit() {
        if (!cnt_) {
            int b = 0;
            if (pos_ < f_.size()) {
                b = f_[pos_];
                if (b == 0xff) {
                    const int nx = pos_ + 1 < f_.size() ? f_[pos_ + 1] : 0xd9;
                    if (nx == 0) pos_ += 2; else { b = 0; ++fed_; }          // at a marker: stay there
                } else ++pos_;
            } else ++fed_;
            buf_ = b; cnt_ = 8;
        }
        return (buf_ >> --cnt_) & 1;
    }
    int receive(int s) { int v = 0; while (s--) v = v << 1 | bit(); return v; }
    int huff(const HT& t) {
        int code = 0;
        for (int l = 1; l <= 16; ++l) {
            code = code << 1 | bit();
            if (t.maxcode[l] >= 0 && code <= t.maxcode[l] && code >= t.mincode[l]) return t.vals[t.valptr[l] + code - t.mincode[l]];
        }
        return -1;
    }
    static int extend(int v, int s) { return s && v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }
    bool block(Comp& c, float out[64]) {
        static const uint8_t zz[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                       35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
        float co[64] = {0};
        // a truncated file (or a tiny one with huge dimensions) must not decode millions of blocks out of fed zeros: a
        // few bytes of slack for the last block's tail, then the scan is corrupt
        if (fed_ > 16) return false;
        const int s = huff(ht_[0][c.td]);
        if (s < 0 || s > 11) return false;
        c.pred += extend(receive(s), s);
        co[0] = (float)(c.pred * q_[c.tq][0]);
        for (int k = 1; k < 64;) {
            const int rs = huff(ht_[1][c.ta]);
            if (rs < 0) return false;
            const int r = rs >> 4, sz = rs & 15;
            if (!sz) { if (r == 15) { k += 16; continue; } break; }
            k += r;
            if (k > 63) return false;
            co[zz[k]] = (float)(extend(receive(sz), sz) * q_[c.tq][k]);
            ++k;
        }
        // separable IDCT: f(x) = 1/2 sum_u C(u) F(u) cos((2x + 1) u pi / 16), rows then columns
        static float cs[8][8];
        static bool init = false;
        if (!init) {
            for (int x = 0; x < 8; ++x)
                for (int u = 0; u < 8; ++u) cs[x][u] = (u ? 1.0f : 0.70710678f) * 0.5f * cosf((float)((2 * x + 1) * u) * 0.19634954f);
            init = true;
        }
        float tmp[64];
        for (int y = 0; y < 8; ++y)
            for (int x = 0; x < 8; ++x) { float a = 0; for (int u = 0; u < 8; ++u) a += cs[x][u] * co[y * 8 + u]; tmp[y * 8 + x] = a; }
        for (int x = 0; x < 8; ++x)
            for (int y = 0; y < 8; ++y) { float a = 0; for (int v = 0; v < 8; ++v) a += cs[y][v] * tmp[v * 8 + x]; out[y * 8 + x] = a + 128.0f; }
        return true;
    }
    bool scan(size_t start, Image& im) {
        pos_ = start; cnt_ = 0; fed_ = 0;
        const int mcuw = 8 * hmax_, mcuh = 8 * vmax_, mx = (w_ + mcuw - 1) / mcuw, my = (h_ + mcuh - 1) / mcuh;
        for (auto& c : comp_) { c.pw = mx * c.h * 8; c.ph = my * c.v * 8; c.plane.assign((size_t)c.pw * c.ph, 0); c.pred = 0; }
        int until_restart = restart_;
        for (int j = 0; j < my; ++j)
            for (int i = 0; i < mx; ++i) {
                if (restart_ && until_restart == 0) {
                    cnt_ = 0;                                   // byte-align, then the RSTn marker
                    while (pos_ + 1 < f_.size() && !(f_[pos_] == 0xff && f_[pos_ + 1] >= 0xd0 && f_[pos_ + 1] <= 0xd7)) ++pos_;
                    if (pos_ + 1 >= f_.size()) return false;
                    pos_ += 2;
                    fed_ = 0;
                    for (auto& c : comp_) c.pred = 0;
                    until_restart = restart_;
                }
                for (auto& c : comp_)
                    for (int by = 0; by < c.v; ++by)
                        for (int bx = 0; bx < c.h; ++bx) {
                            float px[64];
                            if (!block(c, px)) return false;
                            const int ox = (i * c.h + bx) * 8, oy = (j * c.v + by) * 8;
                            for (int y = 0; y < 8; ++y)
                                for (int x = 0; x < 8; ++x) {
                                    const float v = px[y * 8 + x];
                                    c.plane[(size_t)(oy + y) * c.pw + ox + x] = (uint8_t)(v < 0.0f ? 0 : (v > 255.0f ? 255 : (int)(v + 0.5f)));
                                }
                        }
                --until_restart;
            }
        if (!alloc(im, w_, h_)) return false;
        for (int y = 0; y < h_; ++y)
            for (int x = 0; x < w_; ++x) {
                auto at = [&](const Comp& c) { return (float)c.plane[(size_t)(y * c.v / vmax_) * c.pw + (size_t)(x * c.h / hmax_)]; };
                if (comp_.size() == 1) { const uint8_t g = (uint8_t)at(comp_[0]); put(im, x, y, g, g, g, 255); continue; }
                const float Y = at(comp_[0]), cb = at(comp_[1]) - 128.0f, cr = at(comp_[2]) - 128.0f;
                auto cl = [](float v) { return (uint8_t)(v < 0.0f ? 0 : (v > 255.0f ? 255 : (int)(v + 0.5f))); };
                put(im, x, y, cl(Y + 1.402f * cr), cl(Y - 0.344136f * cb - 0.714136f * cr), cl(Y + 1.772f * cb), 255);
            }
        return true;
    }
    const std::vector<uint8_t>& f_;
    int q_[4][64] = {{0}};
    HT ht_[2][4];
    std::vector<Comp> comp_;
    int w_ = 0, h_ = 0, hmax_ = 1, vmax_ = 1, restart_ = 0;
    size_t pos_ = 0;
    int buf_ = 0, cnt_ = 0;
    int fed_ = 0;        // zero bytes fed since the last restart because the entropy-coded data had ended
};

// ------------------------------------------------------------------------------------------- BMP / TGA / PNM --------
inline bool decode_bmp(const std::vector<uint8_t>& f, Image& im) {
    if (f.size() < 54 || f[0] != 'B' || f[1] != 'M') return false;
    auto le32 = [&](size_t o) { return (uint32_t)f[o] | (uint32_t)f[o + 1] << 8 | (uint32_t)f[o + 2] << 16 | (uint32_t)f[o + 3] << 24; };
    const uint32_t off = le32(10), hdr = le32(14);
    if (hdr < 40) return false;
    const int w = (int)le32(18), hs = (int)le32(22), bpp = f[28] | f[29] << 8;
    const uint32_t comp = le32(30);
    if ((bpp != 24 && bpp != 32) || (comp != 0 && !(comp == 3 && bpp == 32))) return false;
    const int h = hs < 0 ? -hs : hs;
    const size_t stride = ((size_t)w * (bpp / 8) + 3) & ~(size_t)3;
    if (!dims_ok(w, h) || (size_t)off + stride * h > f.size() || !alloc(im, w, h)) return false;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const uint8_t* s = &f[off + stride * (size_t)y + (size_t)x * (bpp / 8)];
            const int top = hs < 0 ? y : h - 1 - y;                  // positive height: the file's row 0 is the bottom row
            put(im, x, top, s[2], s[1], s[0], bpp == 32 && comp == 3 ? s[3] : 255);
        }
    return true;
}

inline bool decode_tga(const std::vector<uint8_t>& f, Image& im) {
    if (f.size() < 18) return false;
    const int type = f[2], w = f[12] | f[13] << 8, h = f[14] | f[15] << 8, bits = f[16], bpp = bits / 8;
    const bool rle = type == 10 || type == 11, grey = type == 3 || type == 11;
    if (!(type == 2 || type == 3 || type == 10 || type == 11) || f[1] != 0) return false;
    if (grey ? bits != 8 : (bits != 24 && bits != 32)) return false;
    size_t pos = 18 + (size_t)f[0];
    const bool top_down = (f[17] & 0x20) != 0;
    const size_t npx = (size_t)w * h;
    // the pixels must be able to come out of the file (raw: all of them; RLE: at most 128 per packet of 1 + bpp bytes)
    if (!dims_ok(w, h) || pos > f.size() || (rle ? npx > (f.size() - pos) * 128 : pos + npx * (size_t)bpp > f.size())) return false;
    if (!alloc(im, w, h)) return false;
    std::vector<uint8_t> px(npx * (size_t)bpp);
    if (!rle) {
        if (pos + px.size() > f.size()) return false;
        memcpy(px.data(), &f[pos], px.size());
    } else {
        size_t o = 0;
        while (o < npx) {
            if (pos >= f.size()) return false;
            const int hd = f[pos++], cnt = (hd & 127) + 1;
            if (o + (size_t)cnt > npx) return false;
            if (hd & 128) {
                if (pos + (size_t)bpp > f.size()) return false;
                for (int k = 0; k < cnt; ++k) memcpy(&px[(o + (size_t)k) * bpp], &f[pos], (size_t)bpp);
                pos += (size_t)bpp;
            } else {
                if (pos + (size_t)cnt * bpp > f.size()) return false;
                memcpy(&px[o * bpp], &f[pos], (size_t)cnt * bpp);
                pos += (size_t)cnt * bpp;
            }
            o += (size_t)cnt;
        }
    }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const uint8_t* s = &px[((size_t)y * w + x) * bpp];
            const int top = top_down ? y : h - 1 - y;
            if (grey) put(im, x, top, s[0], s[0], s[0], 255);
            else put(im, x, top, s[2], s[1], s[0], bpp == 4 ? s[3] : 255);
        }
    return true;
}

inline bool decode_pnm(const std::vector<uint8_t>& f, Image& im) {
    if (f.size() < 3 || f[0] != 'P' || (f[1] != '6' && f[1] != '5')) return false;
    const int ch = f[1] == '6' ? 3 : 1;
    size_t pos = 2;
    int vals[3], got = 0;
    while (got < 3 && pos < f.size()) {
        while (pos < f.size() && (f[pos] == ' ' || f[pos] == '\n' || f[pos] == '\r' || f[pos] == '\t')) ++pos;
        if (pos < f.size() && f[pos] == '#') { while (pos < f.size() && f[pos] != '\n') ++pos; continue; }
        int v = 0, digits = 0;
        while (pos < f.size() && f[pos] >= '0' && f[pos] <= '9' && digits < 9) { v = v * 10 + (f[pos] - '0'); ++pos; ++digits; }
        if (!digits) return false;
        vals[got++] = v;
    }
    ++pos;      // the single whitespace after maxval
    if (got < 3 || vals[2] != 255 || !dims_ok(vals[0], vals[1]) || pos + (size_t)vals[0] * vals[1] * ch > f.size() || !alloc(im, vals[0], vals[1])) return false;
    for (int y = 0; y < im.h; ++y)
        for (int x = 0; x < im.w; ++x) {
            const uint8_t* s = &f[pos + ((size_t)y * im.w + x) * ch];
            put(im, x, y, s[0], s[ch == 3 ? 1 : 0], s[ch == 3 ? 2 : 0], 255);
        }
    return true;
}

// by content, not by file name (an .mtl often names a .tga that is a .png)
inline bool decode(const std::vector<uint8_t>& f, Image& im) {
    if (f.size() >= 8 && f[0] == 0x89 && f[1] == 'P') return decode_png(f, im);
    if (f.size() >= 3 && f[0] == 0xff && f[1] == 0xd8) return Jpeg(f).decode(im);
    if (f.size() >= 2 && f[0] == 'B' && f[1] == 'M') return decode_bmp(f, im);
    if (f.size() >= 2 && f[0] == 'P' && (f[1] == '6' || f[1] == '5')) return decode_pnm(f, im);
    return decode_tga(f, im);           // TGA has no signature: last
}

inline bool load(const std::string& path, Image& im) {
    FILE* fp = fopen(path.c_str(), "rb");
    if (!fp) return false;
    std::vector<uint8_t> buf;
    uint8_t tmp[65536];
    size_t n;
    while ((n = fread(tmp, 1, sizeof(tmp), fp)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    fclose(fp);
    return decode(buf, im);
}

}  // namespace vct_image
#endif
