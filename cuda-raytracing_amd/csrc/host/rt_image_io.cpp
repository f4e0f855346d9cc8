// rt_image_io.cpp -- PNG (plain and interlaced) / PPM / JPEG (sequential and progressive) readers, text overlay, display_image and the interaction handlers
// (ImageIO.hpp).  Host-only; nothing here is on the raycast path.
#include "ImageIO.hpp"

#include <cstdio>
#include <cstring>
#include <new>
#include <algorithm>

#include "../../../include/rt_hip.h"
#include "Camera.h"

namespace {

bool fail(std::string* error, const char* msg)
{
    if (error) *error = msg;
    return false;
}

bool read_file(const std::string& path, std::vector<uint8_t>& data)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    uint8_t buf[1 << 16];
    size_t n;
    data.clear();
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) data.insert(data.end(), buf, buf + n);
    fclose(f);
    return true;
}

// ------------------------------------------------------------------------------------------------ inflate (RFC 1951)

struct BitReader {
    const uint8_t* p;
    size_t n, pos = 0;
    uint32_t buf = 0;
    int cnt = 0;
    bool overrun = false;
    BitReader(const uint8_t* p_, size_t n_) : p(p_), n(n_) {}
    uint32_t bits(int need)                                     // LSB-first, need <= 16
    {
        while (cnt < need) {
            if (pos >= n) { overrun = true; return 0; }
            buf |= (uint32_t)p[pos++] << cnt;
            cnt += 8;
        }
        uint32_t v = buf & ((1u << need) - 1u);
        buf >>= need;
        cnt -= need;
        return v;
    }
    void align_byte() { buf = 0; cnt = 0; }
};

// canonical Huffman code: count[len] codes of each length, symbols ordered by (length, value)
struct Huffman {
    uint16_t count[16];
    uint16_t symbol[288];
};

// returns false for an over-subscribed set of lengths (an incomplete set is allowed, as in zlib, when it has one code)
bool build_huffman(Huffman& h, const uint8_t* lengths, int n)
{
    memset(h.count, 0, sizeof h.count);
    for (int i = 0; i < n; i++) h.count[lengths[i]]++;
    int left = 1;
    for (int len = 1; len < 16; len++) {
        left <<= 1;
        left -= h.count[len];
        if (left < 0) return false;
    }
    uint16_t offs[16];
    offs[1] = 0;
    for (int len = 1; len < 15; len++) offs[len + 1] = offs[len] + h.count[len];
    for (int i = 0; i < n; i++)
        if (lengths[i]) h.symbol[offs[lengths[i]]++] = (uint16_t)i;
    return true;
}

int decode_symbol(BitReader& br, const Huffman& h)
{
    int code = 0, first = 0, index = 0;
    for (int len = 1; len < 16; len++) {
        code |= (int)br.bits(1);
        if (br.overrun) return -1;
        const int count = h.count[len];
        if (code - count < first) return h.symbol[index + (code - first)];
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073,
                                4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

// `limit`: the most bytes the caller is prepared to receive (a deflate stream can expand about 1000-fold)
bool inflate_codes(BitReader& br, const Huffman& lit, const Huffman& dist, std::vector<uint8_t>& out, size_t limit)
{
    for (;;) {
        int sym = decode_symbol(br, lit);
        if (sym < 0) return false;
        if (sym < 256) { if (out.size() >= limit) return false; out.push_back((uint8_t)sym); continue; }
        if (sym == 256) return true;
        sym -= 257;
        if (sym >= 29) return false;
        const int len = kLenBase[sym] + (int)br.bits(kLenExtra[sym]);
        const int ds = decode_symbol(br, dist);
        if (ds < 0 || ds >= 30) return false;
        const size_t d = kDistBase[ds] + br.bits(kDistExtra[ds]);
        if (br.overrun || d > out.size() || out.size() + (size_t)len > limit) return false;
        const size_t from = out.size() - d;
        for (int k = 0; k < len; k++) out.push_back(out[from + k]);   // (may overlap: byte by byte)
    }
}

bool inflate_raw(BitReader& br, std::vector<uint8_t>& out, size_t limit)
{
    for (;;) {
        const int last = (int)br.bits(1), type = (int)br.bits(2);
        if (br.overrun) return false;
        if (type == 0) {                                        // stored
            br.align_byte();
            if (br.pos + 4 > br.n) return false;
            const unsigned len = br.p[br.pos] | (br.p[br.pos + 1] << 8), nlen = br.p[br.pos + 2] | (br.p[br.pos + 3] << 8);
            br.pos += 4;
            if ((len ^ 0xFFFFu) != nlen || br.pos + len > br.n || out.size() + len > limit) return false;
            out.insert(out.end(), br.p + br.pos, br.p + br.pos + len);
            br.pos += len;
        } else if (type == 1) {                                 // fixed codes
            uint8_t l[288];
            int i = 0;
            for (; i < 144; i++) l[i] = 8;
            for (; i < 256; i++) l[i] = 9;
            for (; i < 280; i++) l[i] = 7;
            for (; i < 288; i++) l[i] = 8;
            Huffman lit, dist;
            build_huffman(lit, l, 288);
            uint8_t dl[30];
            memset(dl, 5, sizeof dl);
            build_huffman(dist, dl, 30);
            if (!inflate_codes(br, lit, dist, out, limit)) return false;
        } else if (type == 2) {                                 // dynamic codes
            const int nlen = (int)br.bits(5) + 257, ndist = (int)br.bits(5) + 1, ncode = (int)br.bits(4) + 4;
            if (br.overrun || nlen > 286 || ndist > 30) return false;
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint8_t lengths[320];
            memset(lengths, 0, sizeof lengths);
            for (int i = 0; i < ncode; i++) lengths[order[i]] = (uint8_t)br.bits(3);
            Huffman lencode;
            if (!build_huffman(lencode, lengths, 19)) return false;
            uint8_t ll[320];
            int idx = 0;
            while (idx < nlen + ndist) {
                int sym = decode_symbol(br, lencode);
                if (sym < 0) return false;
                if (sym < 16) { ll[idx++] = (uint8_t)sym; continue; }
                int prev = 0, rep;
                if (sym == 16) {
                    if (idx == 0) return false;
                    prev = ll[idx - 1];
                    rep = 3 + (int)br.bits(2);
                } else if (sym == 17) rep = 3 + (int)br.bits(3);
                else rep = 11 + (int)br.bits(7);
                if (br.overrun || idx + rep > nlen + ndist) return false;
                while (rep--) ll[idx++] = (uint8_t)prev;
            }
            if (ll[256] == 0) return false;                     // no end-of-block code
            Huffman lit, dist;
            if (!build_huffman(lit, ll, nlen) || !build_huffman(dist, ll + nlen, ndist)) return false;
            if (!inflate_codes(br, lit, dist, out, limit)) return false;
        } else return false;
        if (last) return true;
    }
}

uint32_t adler32(const std::vector<uint8_t>& d)
{
    uint32_t a = 1, b = 0;
    for (size_t i = 0; i < d.size(); i++) { a = (a + d[i]) % 65521u; b = (b + a) % 65521u; }
    return (b << 16) | a;
}

uint32_t crc32_bytes(const uint8_t* p, size_t n, uint32_t crc = 0)
{
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        ready = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
    return ~crc;
}

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

}  // namespace

bool zlib_inflate(const uint8_t* src, size_t n, std::vector<uint8_t>& out, std::string* error, size_t max_out)
{
    if (n < 6) return fail(error, "zlib stream too short");
    if ((src[0] & 0x0F) != 8 || ((src[0] << 8) | src[1]) % 31 != 0 || (src[1] & 0x20)) return fail(error, "not a zlib deflate stream");
    BitReader br(src + 2, n - 2);
    out.clear();
    if (!inflate_raw(br, out, max_out)) return fail(error, "corrupt deflate data (or more output than expected)");
    br.align_byte();
    if (br.pos + 4 > br.n || be32(br.p + br.pos) != adler32(out)) return fail(error, "zlib checksum mismatch");
    return true;
}

// ------------------------------------------------------------------------------------------------------------- PNG

bool read_png_bgr(const std::string& path, std::vector<uint8_t>& bgr, int& width, int& height, std::string* error)
{
    std::vector<uint8_t> d;
    if (!read_file(path, d)) return fail(error, "cannot open file");
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (d.size() < 8 || memcmp(d.data(), sig, 8) != 0) return fail(error, "not a PNG file");
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, palette;
    bool seen_end = false;
    for (size_t pos = 8; pos + 12 <= d.size() && !seen_end;) {
        const uint32_t len = be32(&d[pos]);
        if (len > d.size() || pos + 12 + (size_t)len > d.size()) return fail(error, "truncated PNG chunk");
        const uint8_t* type = &d[pos + 4];
        const uint8_t* body = &d[pos + 8];
        if (crc32_bytes(type, 4 + (size_t)len) != be32(body + len)) return fail(error, "PNG chunk checksum mismatch");
        if (!memcmp(type, "IHDR", 4)) {
            if (len != 13) return fail(error, "bad IHDR");
            w = be32(body); h = be32(body + 4);
            depth = body[8]; ctype = body[9]; interlace = body[12];
            if (body[10] != 0 || body[11] != 0) return fail(error, "unknown PNG compression / filter method");
        } else if (!memcmp(type, "PLTE", 4)) palette.assign(body, body + len);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!memcmp(type, "IEND", 4)) seen_end = true;
        pos += 12 + (size_t)len;
    }
    if (ctype < 0 || !seen_end || w == 0 || h == 0 || w > 32768 || h > 32768) return fail(error, "incomplete PNG");
    if (interlace != 0 && interlace != 1) return fail(error, "unknown PNG interlace method");
    int channels;
    switch (ctype) {
        case 0: channels = 1; break;
        case 2: channels = 3; break;
        case 3: channels = 1; break;
        case 4: channels = 2; break;
        case 6: channels = 4; break;
        default: return fail(error, "unknown PNG colour type");
    }
    const bool depth_ok = (ctype == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) ||
                          (ctype == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
                          ((ctype == 2 || ctype == 4 || ctype == 6) && (depth == 8 || depth == 16));
    if (!depth_ok) return fail(error, "invalid PNG bit depth");
    if (ctype == 3 && (palette.empty() || palette.size() % 3)) return fail(error, "palette PNG without PLTE");
    // the image is one pass of every pixel, or the seven reduced images of Adam7 (PNG spec 8.2) one after another in the stream:
    // pass k holds the pixels (x0 + i * dx, y0 + j * dy), each pass filtered on its own
    struct Pass { uint32_t x0, y0, dx, dy; };
    static const Pass adam7[7] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    static const Pass whole = {0, 0, 1, 1};
    const Pass* passes = interlace ? adam7 : &whole;
    const int num_passes = interlace ? 7 : 1;
    auto pass_size = [&](const Pass& ps, uint32_t& pw, uint32_t& ph, size_t& row_bytes) {
        pw = w > ps.x0 ? (w - ps.x0 + ps.dx - 1) / ps.dx : 0;
        ph = h > ps.y0 ? (h - ps.y0 + ps.dy - 1) / ps.dy : 0;
        row_bytes = ((size_t)pw * channels * depth + 7) / 8;
    };
    size_t total = 0;
    for (int k = 0; k < num_passes; k++) {
        uint32_t pw, ph; size_t rb;
        pass_size(passes[k], pw, ph, rb);
        if (pw && ph) total += (size_t)ph * (rb + 1);
    }
    std::vector<uint8_t> raw;
    if (!zlib_inflate(idat.data(), idat.size(), raw, error, total)) return false;                         // no more than IHDR announces
    if (raw.size() < total) return fail(error, "PNG image data too short");
    const int bpp = std::max(1, channels * depth / 8);          // filter distance in bytes
    std::vector<uint8_t> out((size_t)w * h * 3);
    size_t base = 0;
    for (int k = 0; k < num_passes; k++) {
        const Pass& ps = passes[k];
        uint32_t pw, ph; size_t row_bytes;
        pass_size(ps, pw, ph, row_bytes);
        if (!pw || !ph) continue;
        // undo the scanline filters in place (PNG spec 9.2)
        for (uint32_t y = 0; y < ph; y++) {
            uint8_t* cur = &raw[base + (size_t)y * (row_bytes + 1)];
            const uint8_t* up = y ? cur - (row_bytes + 1) + 1 : nullptr;
            const int ft = *cur++;
            for (size_t i = 0; i < row_bytes; i++) {
                const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)bpp) ? up[i - bpp] : 0;
                int v = cur[i];
                switch (ft) {
                    case 0: break;
                    case 1: v += a; break;
                    case 2: v += b; break;
                    case 3: v += (a + b) >> 1; break;
                    case 4: v += paeth(a, b, c); break;
                    default: return fail(error, "unknown PNG filter type");
                }
                cur[i] = (uint8_t)v;
            }
        }
        // to B,G,R the way cv::imread(IMREAD_COLOR) does through libpng: 16-bit samples keep their high byte, low-depth
        // grey is scaled to 0..255, palette entries are looked up, alpha is dropped
        for (uint32_t y = 0; y < ph; y++) {
            const uint8_t* row = &raw[base + (size_t)y * (row_bytes + 1) + 1];
            uint8_t* orow = &out[(size_t)(ps.y0 + y * ps.dy) * w * 3];
            auto sample = [&](size_t index) -> int {            // index-th sample of the row, as stored
                if (depth == 8) return row[index];
                if (depth == 16) return row[2 * index];
                const int per = 8 / depth;
                return (row[index / per] >> ((per - 1 - (int)(index % per)) * depth)) & ((1 << depth) - 1);
            };
            for (uint32_t x = 0; x < pw; x++) {
                int r, g, b;
                if (ctype == 0 || ctype == 4) {
                    int v = sample((size_t)x * channels);
                    if (depth < 8) v = v * 255 / ((1 << depth) - 1);
                    r = g = b = v;
                } else if (ctype == 3) {
                    const size_t e = (size_t)sample(x) * 3;
                    if (e + 2 >= palette.size()) return fail(error, "palette index out of range");
                    r = palette[e]; g = palette[e + 1]; b = palette[e + 2];
                } else {
                    r = sample((size_t)x * channels); g = sample((size_t)x * channels + 1); b = sample((size_t)x * channels + 2);
                }
                uint8_t* o = orow + (size_t)(ps.x0 + x * ps.dx) * 3;
                o[0] = (uint8_t)b; o[1] = (uint8_t)g; o[2] = (uint8_t)r;
            }
        }
        base += (size_t)ph * (row_bytes + 1);
    }
    bgr.swap(out);
    width = (int)w; height = (int)h;
    return true;
}

// ------------------------------------------------------------------------------------------------------------- PPM

bool read_ppm_bgr(const std::string& path, std::vector<uint8_t>& bgr, int& width, int& height, std::string* error)
{
    std::vector<uint8_t> d;
    if (!read_file(path, d)) return fail(error, "cannot open file");
    size_t pos = 0;
    auto token = [&](std::string& out) {
        out.clear();
        while (pos < d.size()) {
            const int c = d[pos];
            if (c == '#') { while (pos < d.size() && d[pos] != '\n') pos++; continue; }
            if (!isspace(c)) break;
            pos++;
        }
        while (pos < d.size() && !isspace(d[pos])) out.push_back((char)d[pos++]);
        return !out.empty();
    };
    std::string magic, tw, th, tm;
    if (!token(magic) || magic != "P6" || !token(tw) || !token(th) || !token(tm)) return fail(error, "not a binary PPM (P6) file");
    const int w = atoi(tw.c_str()), h = atoi(th.c_str());
    if (w <= 0 || h <= 0 || atoi(tm.c_str()) != 255) return fail(error, "unsupported PPM header");
    pos++;                                                      // the single whitespace after maxval
    const size_t n = (size_t)w * h * 3;
    if (pos + n > d.size()) return fail(error, "PPM pixel data too short");
    std::vector<uint8_t> out(n);
    for (size_t i = 0; i < (size_t)w * h; i++) { out[3 * i] = d[pos + 3 * i + 2]; out[3 * i + 1] = d[pos + 3 * i + 1]; out[3 * i + 2] = d[pos + 3 * i]; }
    bgr.swap(out);
    width = w; height = h;
    return true;
}

bool read_image_bgr(const std::string& path, std::vector<uint8_t>& bgr, int& width, int& height, std::string* error)
try {
    uint8_t head[4] = {0, 0, 0, 0};
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return fail(error, "cannot open file");
    const size_t n = fread(head, 1, 4, f);
    fclose(f);
    if (n >= 4 && head[0] == 0x89 && head[1] == 'P' && head[2] == 'N' && head[3] == 'G') return read_png_bgr(path, bgr, width, height, error);
    if (n >= 2 && head[0] == 0xFF && head[1] == 0xD8) return read_jpeg_bgr(path, bgr, width, height, error);
    if (n >= 2 && head[0] == 'P' && head[1] == '6') return read_ppm_bgr(path, bgr, width, height, error);
    return fail(error, "unknown image format (PNG, baseline JPEG and binary PPM are supported)");
} catch (const std::bad_alloc&) {
    return fail(error, "out of memory while decoding the image");
}

// --------------------------------------------------------------------------------------------------------- overlay

namespace {
// 5x7 glyphs, one byte per row, bit 4 = leftmost column
struct Glyph { char ch; uint8_t rows[7]; };
const Glyph kFont[] = {
    {'0', {0x0E, 0x11, 0x13, 0x15, 0x19, 0x11, 0x0E}}, {'1', {0x04, 0x0C, 0x04, 0x04, 0x04, 0x04, 0x0E}},
    {'2', {0x0E, 0x11, 0x01, 0x02, 0x04, 0x08, 0x1F}}, {'3', {0x1E, 0x01, 0x01, 0x0E, 0x01, 0x01, 0x1E}},
    {'4', {0x02, 0x06, 0x0A, 0x12, 0x1F, 0x02, 0x02}}, {'5', {0x1F, 0x10, 0x1E, 0x01, 0x01, 0x11, 0x0E}},
    {'6', {0x06, 0x08, 0x10, 0x1E, 0x11, 0x11, 0x0E}}, {'7', {0x1F, 0x01, 0x02, 0x04, 0x08, 0x08, 0x08}},
    {'8', {0x0E, 0x11, 0x11, 0x0E, 0x11, 0x11, 0x0E}}, {'9', {0x0E, 0x11, 0x11, 0x0F, 0x01, 0x02, 0x0C}},
    {'A', {0x0E, 0x11, 0x11, 0x1F, 0x11, 0x11, 0x11}}, {'B', {0x1E, 0x11, 0x11, 0x1E, 0x11, 0x11, 0x1E}},
    {'C', {0x0E, 0x11, 0x10, 0x10, 0x10, 0x11, 0x0E}}, {'D', {0x1C, 0x12, 0x11, 0x11, 0x11, 0x12, 0x1C}},
    {'E', {0x1F, 0x10, 0x10, 0x1E, 0x10, 0x10, 0x1F}}, {'F', {0x1F, 0x10, 0x10, 0x1E, 0x10, 0x10, 0x10}},
    {'G', {0x0E, 0x11, 0x10, 0x17, 0x11, 0x11, 0x0F}}, {'H', {0x11, 0x11, 0x11, 0x1F, 0x11, 0x11, 0x11}},
    {'I', {0x0E, 0x04, 0x04, 0x04, 0x04, 0x04, 0x0E}}, {'J', {0x07, 0x02, 0x02, 0x02, 0x02, 0x12, 0x0C}},
    {'K', {0x11, 0x12, 0x14, 0x18, 0x14, 0x12, 0x11}}, {'L', {0x10, 0x10, 0x10, 0x10, 0x10, 0x10, 0x1F}},
    {'M', {0x11, 0x1B, 0x15, 0x15, 0x11, 0x11, 0x11}}, {'N', {0x11, 0x11, 0x19, 0x15, 0x13, 0x11, 0x11}},
    {'O', {0x0E, 0x11, 0x11, 0x11, 0x11, 0x11, 0x0E}}, {'P', {0x1E, 0x11, 0x11, 0x1E, 0x10, 0x10, 0x10}},
    {'Q', {0x0E, 0x11, 0x11, 0x11, 0x15, 0x12, 0x0D}}, {'R', {0x1E, 0x11, 0x11, 0x1E, 0x14, 0x12, 0x11}},
    {'S', {0x0F, 0x10, 0x10, 0x0E, 0x01, 0x01, 0x1E}}, {'T', {0x1F, 0x04, 0x04, 0x04, 0x04, 0x04, 0x04}},
    {'U', {0x11, 0x11, 0x11, 0x11, 0x11, 0x11, 0x0E}}, {'V', {0x11, 0x11, 0x11, 0x11, 0x11, 0x0A, 0x04}},
    {'W', {0x11, 0x11, 0x11, 0x15, 0x15, 0x15, 0x0A}}, {'X', {0x11, 0x11, 0x0A, 0x04, 0x0A, 0x11, 0x11}},
    {'Y', {0x11, 0x11, 0x11, 0x0A, 0x04, 0x04, 0x04}}, {'Z', {0x1F, 0x01, 0x02, 0x04, 0x08, 0x10, 0x1F}},
    {':', {0x00, 0x04, 0x04, 0x00, 0x04, 0x04, 0x00}}, {'.', {0x00, 0x00, 0x00, 0x00, 0x00, 0x0C, 0x0C}},
    {'-', {0x00, 0x00, 0x00, 0x1F, 0x00, 0x00, 0x00}}, {'+', {0x00, 0x04, 0x04, 0x1F, 0x04, 0x04, 0x00}},
    {'/', {0x01, 0x02, 0x02, 0x04, 0x08, 0x08, 0x10}}, {' ', {0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00}},
    {'?', {0x0E, 0x11, 0x01, 0x02, 0x04, 0x00, 0x04}},
};

const uint8_t* glyph_rows(char ch)
{
    if (ch >= 'a' && ch <= 'z') ch = (char)(ch - 'a' + 'A');
    const int n = (int)(sizeof kFont / sizeof kFont[0]);
    for (int i = 0; i < n; i++)
        if (kFont[i].ch == ch) return kFont[i].rows;
    return kFont[n - 1].rows;
}
}  // namespace

void overlay_text_bgr(uint8_t* bgr, int width, int height, size_t pitch, const std::string& text, int x, int y, int scale,
                      uint8_t b, uint8_t g, uint8_t r)
{
    if (!bgr || scale < 1) return;
    const int top = y - 7 * scale;                              // (x, y) is the bottom-left corner of the text
    for (size_t k = 0; k < text.size(); k++) {
        const uint8_t* rows = glyph_rows(text[k]);
        const int gx = x + (int)k * 6 * scale;                  // 5 columns + 1 of spacing
        for (int ry = 0; ry < 7 * scale; ry++)
            for (int rx = 0; rx < 5 * scale; rx++) {
                if (!((rows[ry / scale] >> (4 - rx / scale)) & 1)) continue;
                const int px = gx + rx, py = top + ry;
                if (px < 0 || py < 0 || px >= width || py >= height) continue;
                uint8_t* o = bgr + (size_t)py * pitch + 3 * (size_t)px;
                o[0] = b; o[1] = g; o[2] = r;
            }
    }
}

// ----------------------------------------------------------------------------------------------------- interaction

void on_mouse(int event, int x, int y, int, void* param)
{
    MouseParams* m = static_cast<MouseParams*>(param);          // kernel.cu:112-139
    if (event == RT_EVENT_LBUTTONDOWN) m->is_down = true;
    else if (event == RT_EVENT_LBUTTONUP) m->is_down = false;
    else if (event == RT_EVENT_MOUSEMOVE) {
        if (m->has_last && m->is_down) {
            const int dx = x - m->last_x, dy = y - m->last_y;
            m->pose->yaw = (float)((double)m->pose->yaw + dx * 0.001);        // float += int * double
            m->pose->pitch = (float)((double)m->pose->pitch + dy * -0.001);
        }
        m->last_x = x;
        m->last_y = y;
        m->has_last = true;
    }
}

bool on_key(int key, MouseParams& mouse_state)
{
    float3 step;                                                // kernel.cu:51-103
    switch (key) {
        case 'w': step = make_float3(0.0f, 0.1f, 0.0f); break;
        case 's': step = make_float3(0.0f, -0.1f, 0.0f); break;
        case 'a': step = make_float3(-0.1f, 0.0f, 0.0f); break;
        case 'd': step = make_float3(0.1f, 0.0f, 0.0f); break;
        case 'q': return false;
        default: return true;
    }
    const lre inv_camera_pose = invert_lre(*mouse_state.pose);
    const float3 new_pos = apply_lre(inv_camera_pose, step);
    mouse_state.pose->x = new_pos.x;
    mouse_state.pose->y = new_pos.y;
    mouse_state.pose->z = new_pos.z;
    return true;
}

int display_image(const uchar3* d_img, int width, int height, size_t pitch, double fps, MouseParams&, const char* path, void* stream)
{
    if (!d_img || width <= 0 || height <= 0) return RT_E_INVALID;
    std::vector<uint8_t> host((size_t)width * 3 * (size_t)height);
    // the copy is ordered on the stream the frame was rendered on (Camera::stream): a non-blocking stream is not ordered
    // against the null stream, and render_scene() is asynchronous by default (Camera.cu:38-39)
    int rc = rt_memcpy2d_d2h(host.data(), (size_t)width * 3, d_img, pitch, (size_t)width * 3, (size_t)height, stream);
    if (rc) return rc;
    // cv::putText(img, "FPS: " + std::to_string(fps), Point(10, 30), FONT_HERSHEY_SIMPLEX, 1.0, Scalar(0, 255, 0), 2):
    // Hershey simplex at scale 1 is about 22 pixels tall; the built-in 5x7 font at scale 3 is 21.
    overlay_text_bgr(host.data(), width, height, (size_t)width * 3, "FPS: " + std::to_string(fps), 10, 30, 3, 0, 255, 0);
    return write_png_bgr(path, host.data(), width, height, (size_t)width * 3);
}

// ------------------------------------------------------------------------------------------------------------ JPEG
//
// Sequential and progressive JPEG (SOF0 / SOF1 / SOF2, 8-bit, Huffman; grey or Y'CbCr with 4:4:4, 4:2:2 or 4:2:0 chroma;
// any number of scans, interleaved or not; restart intervals).  Every scan decodes into per-component coefficient arrays
// (progressive: spectral selection and successive approximation as in jdphuff.c, ITU T.81 annex G); after the last scan
// the blocks are dequantised and transformed.  libjpeg smooths blocks only while a progressive file is incomplete, and an
// incomplete file is refused here, so complete files need no smoothing to match it.
// The reference decodes with cv::imread, i.e. libjpeg with its default settings; a lossy format is only
// "the same texture" if the decoder reproduces that arithmetic, so every stage restates libjpeg's defaults
// (third-party, not in the reference tree; libjpeg 6b / libjpeg-turbo):
//   * inverse DCT: the "slow but accurate" integer transform (jidctint.c: 13-bit constants, 2 extra bits after pass 1);
//   * chroma upsampling: "fancy" triangle filters (jdsample.c: h2v1 3/4-1/4 horizontally, h2v2 9-3-3-1);
//   * colour: 16-bit fixed-point Y'CbCr -> RGB (jdcolor.c).
// tests/test_host_logic.py pins the output to Pillow's decoder (libjpeg-turbo, same defaults) bit for bit.
// Arithmetic-coded, lossless, 12-bit, CMYK and other sampling layouts are refused.

namespace {

const uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                             35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct JpegHuff {
    bool present = false;
    int mincode[17], maxcode[18], valptr[17];
    uint8_t vals[256];
};

struct JpegComponent {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int dc_pred = 0;
    int bw = 0, bh = 0;                 // blocks per row / column as stored (padded to whole MCUs)
    int dw = 0, dh = 0;                 // downsampled size in samples: ceil(image * h / hmax)
    bool q_latched = false;
    uint16_t q[64];                     // the quantisation table as it stood at the component's first scan (what libjpeg latches)
    std::vector<int16_t> coef;          // bw*bh blocks of 64 coefficients, natural order
    std::vector<uint8_t> plane;         // bw*8 x bh*8 samples
};

struct JpegBits {
    const uint8_t* p;
    size_t n, pos;
    uint32_t buf = 0;
    int cnt = 0;
    bool bad = false;
    int bit()
    {
        if (cnt == 0) {
            if (pos >= n) { bad = true; return 0; }
            uint8_t b = p[pos++];
            if (b == 0xFF) {
                if (pos < n && p[pos] == 0x00) pos++;           // stuffed zero
                else { bad = true; return 0; }                  // a marker inside entropy-coded data
            }
            buf = b;
            cnt = 8;
        }
        return (buf >> --cnt) & 1;
    }
    int receive(int s) { int v = 0; while (s--) v = (v << 1) | bit(); return v; }
    void reset() { cnt = 0; }
};

int jpeg_decode_huff(JpegBits& br, const JpegHuff& h)
{
    int code = 0;
    for (int len = 1; len <= 16; len++) {
        code = (code << 1) | br.bit();
        if (br.bad) return -1;
        if (h.maxcode[len] >= 0 && code <= h.maxcode[len] && code >= h.mincode[len]) return h.vals[h.valptr[len] + code - h.mincode[len]];
    }
    return -1;
}

inline int jpeg_extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

// jidctint.c (jpeg_idct_islow): dequantised coefficients in natural order -> 64 samples
void idct_islow(const int* in, uint8_t* out, int stride)
{
    const int CONST_BITS = 13, PASS1_BITS = 2;
    const long F_0_298 = 2446, F_0_390 = 3196, F_0_541 = 4433, F_0_765 = 6270, F_0_899 = 7373, F_1_175 = 9633, F_1_501 = 12299,
               F_1_847 = 15137, F_1_961 = 16069, F_2_053 = 16819, F_2_562 = 20995, F_3_072 = 25172;
    long ws[64];
    auto descale = [](long x, int n) { return (x + (1L << (n - 1))) >> n; };
    for (int pass = 0; pass < 2; pass++) {
        for (int k = 0; k < 8; k++) {
            long c[8];
            for (int i = 0; i < 8; i++) c[i] = pass == 0 ? in[8 * i + k] : ws[8 * k + i];   // pass 1: column k; pass 2: row k
            long z2 = c[2], z3 = c[6];
            long z1 = (z2 + z3) * F_0_541;
            long tmp2 = z1 + z3 * (-F_1_847), tmp3 = z1 + z2 * F_0_765;
            z2 = c[0]; z3 = c[4];
            long tmp0 = (z2 + z3) * (1L << CONST_BITS), tmp1 = (z2 - z3) * (1L << CONST_BITS);
            const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
            tmp0 = c[7]; tmp1 = c[5]; tmp2 = c[3]; tmp3 = c[1];
            z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
            long z4 = tmp1 + tmp3;
            const long z5 = (z3 + z4) * F_1_175;
            tmp0 *= F_0_298; tmp1 *= F_2_053; tmp2 *= F_3_072; tmp3 *= F_1_501;
            z1 *= -F_0_899; z2 *= -F_2_562; z3 *= -F_1_961; z4 *= -F_0_390;
            z3 += z5; z4 += z5;
            tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
            const long r[8] = {tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0, tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2, tmp10 - tmp3};
            if (pass == 0) {
                for (int i = 0; i < 8; i++) ws[8 * i + k] = descale(r[i], CONST_BITS - PASS1_BITS);
            } else {
                for (int i = 0; i < 8; i++) {
                    long v = descale(r[i], CONST_BITS + PASS1_BITS + 3) + 128;      // range_limit around CENTERJSAMPLE
                    out[k * stride + i] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
                }
            }
        }
    }
}

uint16_t be16(const uint8_t* p) { return (uint16_t)((p[0] << 8) | p[1]); }

}  // namespace

bool read_jpeg_bgr(const std::string& path, std::vector<uint8_t>& bgr, int& width, int& height, std::string* error)
{
    std::vector<uint8_t> d;
    if (!read_file(path, d)) return fail(error, "cannot open file");
    if (d.size() < 4 || d[0] != 0xFF || d[1] != 0xD8) return fail(error, "not a JPEG file");
    uint16_t qt[4][64];
    bool have_qt[4] = {false, false, false, false};
    JpegHuff hdc[4], hac[4];
    std::vector<JpegComponent> comps;
    int W = 0, H = 0, restart_interval = 0, hmax = 1, vmax = 1;
    int mcux = 0, mcuy = 0, scans = 0;
    bool have_frame = false, progressive = false, saw_eoi = false;
    size_t pos = 2;
    while (pos + 2 <= d.size()) {
        if (d[pos] != 0xFF) return fail(error, "JPEG marker expected");
        while (pos < d.size() && d[pos] == 0xFF) pos++;         // fill bytes
        if (pos >= d.size()) break;
        const int marker = d[pos++];
        if (marker == 0xD9) { saw_eoi = true; break; }          // EOI
        if (marker == 0x01 || (marker >= 0xD0 && marker <= 0xD7)) continue;
        if (pos + 2 > d.size()) return fail(error, "truncated JPEG");
        const size_t len = be16(&d[pos]);
        if (len < 2 || pos + len > d.size()) return fail(error, "truncated JPEG segment");
        const uint8_t* s = &d[pos + 2];
        const size_t n = len - 2;
        if (marker == 0xDB) {                                   // DQT
            for (size_t i = 0; i < n;) {
                const int pq = s[i] >> 4, tq = s[i] & 15;
                i++;
                if (tq > 3 || i + (pq ? 128 : 64) > n) return fail(error, "bad DQT");
                for (int k = 0; k < 64; k++) { qt[tq][kZigzag[k]] = pq ? be16(&s[i + 2 * k]) : s[i + k]; }
                i += pq ? 128 : 64;
                have_qt[tq] = true;
            }
        } else if (marker == 0xC4) {                            // DHT
            for (size_t i = 0; i < n;) {
                if (i + 17 > n) return fail(error, "bad DHT");
                const int tc = s[i] >> 4, th = s[i] & 15;
                if (tc > 1 || th > 3) return fail(error, "bad DHT");
                JpegHuff& h = tc ? hac[th] : hdc[th];
                int total = 0, code = 0;
                for (int l = 1; l <= 16; l++) {
                    const int c = s[i + l];
                    h.valptr[l] = total;
                    h.mincode[l] = code;
                    h.maxcode[l] = c ? code + c - 1 : -1;
                    code = (code + c) << 1;
                    total += c;
                }
                if (total > 256 || i + 17 + total > n) return fail(error, "bad DHT");
                memcpy(h.vals, &s[i + 17], total);
                h.present = true;
                i += 17 + total;
            }
        } else if (marker == 0xC0 || marker == 0xC1 || marker == 0xC2) {   // SOF0 / SOF1 / SOF2: baseline, extended sequential, progressive (Huffman)
            if (have_frame) return fail(error, "JPEG with more than one frame header");   // (a stale hmax / vmax would size the planes wrongly)
            if (n < 6 || s[0] != 8) return fail(error, "only 8-bit JPEG is supported");
            progressive = marker == 0xC2;
            hmax = vmax = 1;
            H = be16(&s[1]); W = be16(&s[3]);
            const int nc = s[5];
            if (W <= 0 || H <= 0 || (nc != 1 && nc != 3) || n < 6 + 3 * (size_t)nc) return fail(error, "unsupported JPEG frame (grey or 3 components)");
            if ((size_t)W * (size_t)H > ((size_t)1 << 28)) return fail(error, "JPEG frame too large");
            comps.assign(nc, JpegComponent());
            for (int c = 0; c < nc; c++) {
                comps[c].id = s[6 + 3 * c]; comps[c].h = s[7 + 3 * c] >> 4; comps[c].v = s[7 + 3 * c] & 15; comps[c].tq = s[8 + 3 * c];
                if (comps[c].h < 1 || comps[c].v < 1 || comps[c].tq > 3) return fail(error, "bad JPEG component");
                for (int k = 0; k < c; k++) if (comps[k].id == comps[c].id) return fail(error, "bad JPEG component");
                hmax = std::max(hmax, comps[c].h); vmax = std::max(vmax, comps[c].v);
            }
            if (nc == 1) { comps[0].h = comps[0].v = 1; hmax = vmax = 1; }       // a single component is never interleaved
            if (nc == 3) {
                const bool chroma_ok = comps[1].h == 1 && comps[1].v == 1 && comps[2].h == 1 && comps[2].v == 1;
                const bool luma_ok = (comps[0].h == 1 && comps[0].v == 1) || (comps[0].h == 2 && comps[0].v == 1) || (comps[0].h == 2 && comps[0].v == 2);
                if (!chroma_ok || !luma_ok) return fail(error, "unsupported JPEG chroma subsampling (4:4:4, 4:2:2, 4:2:0 only)");
            }
            mcux = (W + 8 * hmax - 1) / (8 * hmax); mcuy = (H + 8 * vmax - 1) / (8 * vmax);
            for (auto& c : comps) {
                c.bw = mcux * c.h; c.bh = mcuy * c.v;
                c.dw = (W * c.h + hmax - 1) / hmax; c.dh = (H * c.v + vmax - 1) / vmax;
                c.coef.assign((size_t)c.bw * c.bh * 64, 0);
            }
            have_frame = true;
        } else if (marker >= 0xC3 && marker <= 0xCF && marker != 0xC4 && marker != 0xC8 && marker != 0xCC) {
            return fail(error, "lossless / hierarchical / arithmetic JPEG is not supported");
        } else if (marker == 0xDD) {                            // DRI
            if (n < 2) return fail(error, "bad DRI");
            restart_interval = be16(s);
        } else if (marker == 0xDA) {                            // SOS: one scan (baseline files usually hold one, progressive files several)
            if (!have_frame) return fail(error, "JPEG scan before frame header");
            if (n < 1) return fail(error, "bad SOS");
            const int ns = s[0];
            if (ns < 1 || ns > (int)comps.size() || n < 1 + 2 * (size_t)ns + 3) return fail(error, "bad SOS");
            JpegComponent* sc[3];
            for (int k = 0; k < ns; k++) {
                JpegComponent* c = nullptr;
                for (auto& cc : comps) if (cc.id == s[1 + 2 * k]) c = &cc;
                if (!c || (k && c <= sc[k - 1])) return fail(error, "unexpected JPEG scan component order");
                c->td = s[2 + 2 * k] >> 4; c->ta = s[2 + 2 * k] & 15;
                if (c->td > 3 || c->ta > 3) return fail(error, "JPEG scan refers to a missing table");
                sc[k] = c;
            }
            const int Ss = s[1 + 2 * ns], Se = s[2 + 2 * ns], Ah = s[3 + 2 * ns] >> 4, Al = s[3 + 2 * ns] & 15;
            if (progressive) {
                // T.81 G.1.1.1: a DC scan (Ss = Se = 0) may interleave, an AC scan (1 <= Ss <= Se <= 63) holds one component;
                // a refinement scan improves the previous scan's precision by exactly one bit
                const bool ok = Ss <= Se && Se <= 63 && (Ss == 0 ? Se == 0 : ns == 1) && Al <= 13 && (Ah == 0 || Ah == Al + 1);
                if (!ok) return fail(error, "invalid progressive JPEG scan parameters");
            } else if (Ss != 0 || Se != 63 || Ah != 0 || Al != 0) return fail(error, "invalid sequential JPEG scan parameters");
            const bool need_dc = Ss == 0 && Ah == 0, need_ac = Se > 0;      // (a DC refinement scan reads raw bits only)
            for (int k = 0; k < ns; k++) {
                JpegComponent& c = *sc[k];
                if ((need_dc && !hdc[c.td].present) || (need_ac && !hac[c.ta].present) || !have_qt[c.tq]) return fail(error, "JPEG scan refers to a missing table");
                // a sequential frame codes every component exactly once: a second scan of one would be decoded on top of the
                // first one's coefficients (libjpeg refuses such files too)
                if (!progressive && c.q_latched) return fail(error, "JPEG component scanned twice in a sequential frame");
                if (!c.q_latched) { memcpy(c.q, qt[c.tq], sizeof c.q); c.q_latched = true; }
                c.dc_pred = 0;
            }
            // an interleaved scan walks MCUs (h x v blocks of every component); a one-component scan walks that component's
            // blocks in raster order over its real size, ceil(samples / 8) -- not the size padded to whole MCUs (T.81 A.2.2)
            const bool interleaved = ns > 1;
            const int ux = interleaved ? mcux : (sc[0]->dw + 7) / 8, uy = interleaved ? mcuy : (sc[0]->dh + 7) / 8;
            JpegBits br{d.data(), d.size(), pos + len};
            int until_restart = restart_interval, next_rst = 0, eobrun = 0;
            const int p1 = 1 << Al, m1 = -(1 << Al);
            auto refine = [&](int16_t& v) {                    // one correction bit for a coefficient that is already non-zero
                if (br.bit() && (v & p1) == 0) v = (int16_t)(v >= 0 ? v + p1 : v + m1);
            };
            for (int my = 0; my < uy; my++)
                for (int mx = 0; mx < ux; mx++) {
                    if (restart_interval && until_restart == 0) {
                        br.reset();
                        if (br.pos + 2 > br.n || br.p[br.pos] != 0xFF || br.p[br.pos + 1] != 0xD0 + next_rst) return fail(error, "JPEG restart marker missing");
                        br.pos += 2;
                        next_rst = (next_rst + 1) & 7;
                        until_restart = restart_interval;
                        eobrun = 0;
                        for (int k = 0; k < ns; k++) sc[k]->dc_pred = 0;
                    }
                    for (int ci = 0; ci < ns; ci++) {
                        JpegComponent& c = *sc[ci];
                        const int nbx = interleaved ? c.h : 1, nby = interleaved ? c.v : 1;
                        for (int by = 0; by < nby; by++)
                            for (int bx = 0; bx < nbx; bx++) {
                                const int gx = interleaved ? mx * c.h + bx : mx, gy = interleaved ? my * c.v + by : my;
                                int16_t* blk = &c.coef[((size_t)gy * c.bw + gx) * 64];
                                if (Ss == 0) {
                                    if (Ah == 0) {                          // DC, first (or only) pass
                                        const int t = jpeg_decode_huff(br, hdc[c.td]);
                                        if (t < 0 || t > 11) return fail(error, "corrupt JPEG data");
                                        if (t) c.dc_pred += jpeg_extend(br.receive(t), t);
                                        // a DC value is at most 11 bits + sign before the point transform (T.81 F.1.2.1): a running
                                        // sum outside that range is a corrupt stream, and unchecked it could overflow the int
                                        if (c.dc_pred < -32768 || c.dc_pred > 32767 || (c.dc_pred * p1) < -32768 || (c.dc_pred * p1) > 32767) return fail(error, "corrupt JPEG data");
                                        blk[0] = (int16_t)(c.dc_pred * p1);
                                    } else if (br.bit()) blk[0] = (int16_t)(blk[0] | p1);   // DC refinement: one more bit
                                }
                                if (!progressive) {                         // sequential: the 63 AC coefficients follow at once
                                    for (int k = 1; k < 64;) {
                                        const int rs = jpeg_decode_huff(br, hac[c.ta]);
                                        if (rs < 0) return fail(error, "corrupt JPEG data");
                                        const int r = rs >> 4, sz = rs & 15;
                                        if (sz == 0) {
                                            if (r == 15) { k += 16; continue; }
                                            break;                          // end of block
                                        }
                                        k += r;
                                        if (k > 63) return fail(error, "corrupt JPEG data");
                                        blk[kZigzag[k]] = (int16_t)jpeg_extend(br.receive(sz), sz);
                                        k++;
                                    }
                                } else if (Ss > 0 && Ah == 0) {             // AC band, first pass (jdphuff.c decode_mcu_AC_first)
                                    if (eobrun > 0) eobrun--;
                                    else
                                        for (int k = Ss; k <= Se; k++) {
                                            const int rs = jpeg_decode_huff(br, hac[c.ta]);
                                            if (rs < 0) return fail(error, "corrupt JPEG data");
                                            const int r = rs >> 4, sz = rs & 15;
                                            if (sz) {
                                                k += r;
                                                if (k > Se) return fail(error, "corrupt JPEG data");
                                                blk[kZigzag[k]] = (int16_t)(jpeg_extend(br.receive(sz), sz) * p1);
                                            } else if (r == 15) k += 15;
                                            else {                          // EOBr: this band ends here in this block and in the next eobrun blocks
                                                eobrun = (1 << r) - 1;
                                                if (r) eobrun += br.receive(r);
                                                break;
                                            }
                                        }
                                } else if (Ss > 0) {                        // AC band, refinement (decode_mcu_AC_refine)
                                    int k = Ss;
                                    if (eobrun == 0)
                                        for (; k <= Se; k++) {
                                            const int rs = jpeg_decode_huff(br, hac[c.ta]);
                                            if (rs < 0) return fail(error, "corrupt JPEG data");
                                            int r = rs >> 4, sz = rs & 15, value = 0;
                                            if (sz) {
                                                if (sz != 1) return fail(error, "corrupt JPEG data");
                                                value = br.bit() ? p1 : m1;                 // a coefficient that becomes non-zero in this pass
                                            } else if (r != 15) {
                                                eobrun = 1 << r;
                                                if (r) eobrun += br.receive(r);
                                                break;
                                            }
                                            // pass r coefficients that are still zero; every non-zero one on the way takes a correction bit
                                            for (; k <= Se; k++) {
                                                int16_t& v = blk[kZigzag[k]];
                                                if (v != 0) refine(v);
                                                else if (--r < 0) break;
                                            }
                                            if (value) {
                                                if (k > Se) return fail(error, "corrupt JPEG data");
                                                blk[kZigzag[k]] = (int16_t)value;
                                            }
                                        }
                                    if (eobrun > 0) {
                                        for (; k <= Se; k++) {
                                            int16_t& v = blk[kZigzag[k]];
                                            if (v != 0) refine(v);
                                        }
                                        eobrun--;
                                    }
                                }
                                if (br.bad) return fail(error, "truncated JPEG data");
                            }
                    }
                    if (restart_interval) until_restart--;
                }
            scans++;
            // the next marker follows the entropy-coded bytes (pad bits of the last byte are dropped)
            size_t q = br.pos;
            while (q + 1 < d.size() && !(d[q] == 0xFF && d[q + 1] != 0x00 && d[q + 1] != 0xFF && !(d[q + 1] >= 0xD0 && d[q + 1] <= 0xD7))) q++;
            pos = q;
            continue;
        }
        pos += len;
    }
    if (!scans) return fail(error, "JPEG without image data");
    // every component needs its coefficients: a sequential file one scan per component, a progressive file at least the DC scan
    // (libjpeg would show an incomplete progressive file smoothed; such a file is refused here rather than shown differently)
    for (auto& c : comps) if (!c.q_latched) return fail(error, "JPEG component without a scan");
    if (progressive && !saw_eoi) return fail(error, "incomplete progressive JPEG");
    // dequantise + inverse DCT (jddctmgr.c / jidctint.c), block by block
    for (auto& c : comps) {
        c.plane.assign((size_t)c.bw * 8 * c.bh * 8, 0);
        for (int gy = 0; gy < c.bh; gy++)
            for (int gx = 0; gx < c.bw; gx++) {
                const int16_t* blk = &c.coef[((size_t)gy * c.bw + gx) * 64];
                int deq[64];
                for (int k = 0; k < 64; k++) deq[k] = blk[k] * c.q[k];
                idct_islow(deq, &c.plane[(size_t)gy * 8 * c.bw * 8 + (size_t)gx * 8], c.bw * 8);
            }
        std::vector<int16_t>().swap(c.coef);
    }

    std::vector<uint8_t> out((size_t)W * H * 3);
    if (comps.size() == 1) {
        const JpegComponent& y = comps[0];
        for (int r = 0; r < H; r++)
            for (int x = 0; x < W; x++) {
                const uint8_t v = y.plane[(size_t)r * y.bw * 8 + x];
                uint8_t* o = &out[((size_t)r * W + x) * 3];
                o[0] = o[1] = o[2] = v;
            }
    } else {
        // chroma to full resolution (jdsample.c, do_fancy_upsampling): one output row at a time
        const JpegComponent& Y = comps[0];
        const bool h2 = Y.h == 2, v2 = Y.v == 2;
        std::vector<uint8_t> up[2] = {std::vector<uint8_t>((size_t)W + 2), std::vector<uint8_t>((size_t)W + 2)};
        std::vector<int> colsum;
        for (int r = 0; r < H; r++) {
            for (int ci = 0; ci < 2; ci++) {
                const JpegComponent& c = comps[1 + ci];
                const int stride = c.bw * 8, n = c.dw;          // n real input columns
                uint8_t* o = up[ci].data();
                auto row = [&](int rr) { return &c.plane[(size_t)std::min(std::max(rr, 0), c.dh - 1) * stride]; };
                if (!h2 && !v2) {
                    memcpy(o, row(r), (size_t)W);
                } else if (h2 && !v2) {                         // h2v1_fancy_upsample
                    const uint8_t* in = row(r);
                    if (n == 1) { o[0] = o[1] = in[0]; }
                    else {
                        o[0] = in[0];
                        o[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
                        for (int i = 1; i < n - 1; i++) {
                            const int v = in[i] * 3;
                            o[2 * i] = (uint8_t)((v + in[i - 1] + 1) >> 2);
                            o[2 * i + 1] = (uint8_t)((v + in[i + 1] + 2) >> 2);
                        }
                        o[2 * n - 2] = (uint8_t)((in[n - 1] * 3 + in[n - 2] + 1) >> 2);
                        if (2 * n - 1 < W + 2) o[2 * n - 1] = in[n - 1];
                    }
                } else {                                        // h2v2_fancy_upsample: 3/4 nearer row + 1/4 further row, then the same across
                    const int ir = r >> 1;
                    const uint8_t* near = row(ir);
                    const uint8_t* far = row((r & 1) ? ir + 1 : ir - 1);
                    colsum.resize((size_t)n);
                    for (int i = 0; i < n; i++) colsum[i] = near[i] * 3 + far[i];
                    if (n == 1) { o[0] = (uint8_t)((colsum[0] * 4 + 8) >> 4); o[1] = (uint8_t)((colsum[0] * 4 + 7) >> 4); }
                    else {
                        o[0] = (uint8_t)((colsum[0] * 4 + 8) >> 4);
                        o[1] = (uint8_t)((colsum[0] * 3 + colsum[1] + 7) >> 4);
                        for (int i = 1; i < n - 1; i++) {
                            o[2 * i] = (uint8_t)((colsum[i] * 3 + colsum[i - 1] + 8) >> 4);
                            o[2 * i + 1] = (uint8_t)((colsum[i] * 3 + colsum[i + 1] + 7) >> 4);
                        }
                        o[2 * n - 2] = (uint8_t)((colsum[n - 1] * 3 + colsum[n - 2] + 8) >> 4);
                        if (2 * n - 1 < W + 2) o[2 * n - 1] = (uint8_t)((colsum[n - 1] * 4 + 7) >> 4);
                    }
                }
            }
            // jdcolor.c ycc_rgb_convert
            const uint8_t* yr = &Y.plane[(size_t)r * Y.bw * 8];
            for (int x = 0; x < W; x++) {
                const int y = yr[x], cb = up[0][x] - 128, cr = up[1][x] - 128;
                int R = y + (int)((91881L * cr + 32768) >> 16);
                int B = y + (int)((116130L * cb + 32768) >> 16);
                int G = y + (int)((-22554L * cb + 32768 - 46802L * cr) >> 16);
                uint8_t* o = &out[((size_t)r * W + x) * 3];
                o[0] = (uint8_t)(B < 0 ? 0 : B > 255 ? 255 : B);
                o[1] = (uint8_t)(G < 0 ? 0 : G > 255 ? 255 : G);
                o[2] = (uint8_t)(R < 0 ? 0 : R > 255 ? 255 : R);
            }
        }
    }
    bgr.swap(out);
    width = W; height = H;
    return true;
}
