// exr_io.cpp — from-scratch OpenEXR reader/writer behind ir_exr_load / ir_exr_save.
//
// Replaces ExrImage::load_exr_image (reference core-engine/src/file_formats/exr.rs:26-49), which calls
// exr::prelude::read_first_rgba_layer_from_file from the third-party `exr` crate (1.73, not vendored): the
// first layer that has R, G and B channels is converted to f32, alpha is dropped and pixel (x, y) of the
// data window is stored at y*width + x.  Neither that crate nor OpenEXR exists in this environment, so the
// container format is implemented here from its public specification:
//   magic 20000630, version word (bit 9 tiled, bit 11 deep, bit 12 multipart), attribute list, offset table,
//   chunks = [part] y size data (scanline) or [part] tx ty lx ly size data (tile), channel-planar lines,
//   ZIP/ZIPS = zlib deflate over (byte de-interleave + delta predictor), RLE = the same filter + byte runs.
// Supported: single- and multi-part files, scanline and tiled (level 0) parts, HALF / FLOAT / UINT samples,
// compression NONE, RLE, ZIPS, ZIP, PIZ, PXR24 and the lossy B44 / B44A (the last four in exr_codecs.cpp).  DWAA/DWAB
// are reported as unsupported.
#include <zlib.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "../../include/insploray_hip.h"
#include "exr_codecs.h"

namespace {

thread_local std::string g_exr_error;

struct Reader {
    const uint8_t* p;
    size_t n, pos;
    bool ok = true;
    bool need(size_t k) {
        if (pos + k > n || pos + k < pos) ok = false;
        return ok;
    }
    uint8_t u8() { return need(1) ? p[pos++] : 0; }
    int32_t i32() {
        if (!need(4)) return 0;
        uint32_t v = (uint32_t)p[pos] | ((uint32_t)p[pos + 1] << 8) | ((uint32_t)p[pos + 2] << 16) | ((uint32_t)p[pos + 3] << 24);
        pos += 4;
        return (int32_t)v;
    }
    uint64_t u64() {
        if (!need(8)) return 0;
        uint64_t v = 0;
        for (int i = 7; i >= 0; --i) v = (v << 8) | p[pos + i];
        pos += 8;
        return v;
    }
    std::string cstr() {
        std::string s;
        while (need(1)) {
            char c = (char)p[pos++];
            if (!c) return s;
            s.push_back(c);
            if (s.size() > 255) {
                ok = false;
                break;
            }
        }
        return s;
    }
};

struct Channel {
    std::string name;
    int type;  // 0 UINT, 1 HALF, 2 FLOAT
    int plinear = 0;  // "perceptually linear" flag of the channel list: only B44(A) looks at it
    int xs, ys;
    size_t bytes() const { return type == 1 ? 2 : 4; }
};

struct Part {
    std::vector<Channel> channels;
    int compression = -1;
    int xmin = 0, ymin = 0, xmax = -1, ymax = -1;
    bool tiled = false;
    uint32_t tile_w = 0, tile_h = 0;
    int tile_mode = 0;
    int chunk_count = -1;
    std::string type;
    bool has_data_window = false;
};

float half_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1f;
    uint32_t man = h & 0x3ffu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else {
            int e = -1;
            do {
                man <<= 1;
                ++e;
            } while (!(man & 0x400u));
            man &= 0x3ffu;
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7f800000u | (man << 13);
    } else {
        bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    }
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}

uint16_t float_to_half(float f) {  // round to nearest even
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0));
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);  // overflows to inf after rounding
    if (x < 0x33000001u) return (uint16_t)sign;               // rounds to zero
    int exp = (int)(x >> 23) - 127 + 15;
    uint32_t man = x & 0x7fffffu;
    if (exp <= 0) {
        man |= 0x800000u;
        const int shift = 14 - exp;
        uint32_t h = man >> shift;
        const uint32_t rem = man & ((1u << shift) - 1), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (h & 1))) ++h;
        return (uint16_t)(sign | h);
    }
    uint32_t h = ((uint32_t)exp << 10) | (man >> 13);
    const uint32_t rem = man & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) ++h;
    return (uint16_t)(sign | h);
}

// inverse of the ZIP/RLE byte filter: delta predictor, then interleave the two halves
void unfilter(std::vector<uint8_t>& buf, std::vector<uint8_t>& scratch) {
    const size_t n = buf.size();
    if (n == 0) return;
    for (size_t i = 1; i < n; ++i) buf[i] = (uint8_t)(buf[i - 1] + buf[i] - 128);
    scratch.resize(n);
    const size_t half = (n + 1) / 2;
    size_t a = 0, b = half;
    for (size_t i = 0; i < n;) {
        scratch[i++] = buf[a++];
        if (i < n) scratch[i++] = buf[b++];
    }
    buf.swap(scratch);
}

void filter(const uint8_t* raw, size_t n, std::vector<uint8_t>& out) {
    out.resize(n);
    if (n == 0) return;
    const size_t half = (n + 1) / 2;
    size_t a = 0, b = half;
    for (size_t i = 0; i < n;) {
        out[a++] = raw[i++];
        if (i < n) out[b++] = raw[i++];
    }
    int p = out[0];
    for (size_t i = 1; i < n; ++i) {
        const int d = (int)out[i] - p + (128 + 256);
        p = out[i];
        out[i] = (uint8_t)d;
    }
}

bool rle_decode(const uint8_t* in, size_t in_n, std::vector<uint8_t>& out, size_t expect) {
    out.clear();
    out.reserve(expect);
    size_t i = 0;
    while (i < in_n) {
        const int8_t c = (int8_t)in[i++];
        if (c < 0) {
            const size_t k = (size_t)(-(int)c);
            if (i + k > in_n || out.size() + k > expect) return false;
            out.insert(out.end(), in + i, in + i + k);
            i += k;
        } else {
            const size_t k = (size_t)c + 1;
            if (i >= in_n || out.size() + k > expect) return false;
            out.insert(out.end(), k, in[i++]);
        }
    }
    return out.size() == expect;
}

void rle_encode(const std::vector<uint8_t>& in, std::vector<uint8_t>& out) {
    out.clear();
    const size_t n = in.size();
    size_t run_start = 0;
    while (run_start < n) {
        size_t run_end = run_start + 1;
        while (run_end < n && in[run_end] == in[run_start] && run_end - run_start < 128) ++run_end;
        if (run_end - run_start >= 3) {
            out.push_back((uint8_t)(run_end - run_start - 1));
            out.push_back(in[run_start]);
            run_start = run_end;
        } else {
            // literal block: until a run of >= 3 starts, at most 127 bytes
            size_t lit_end = run_start;
            while (lit_end < n && lit_end - run_start < 127) {
                if (lit_end + 2 < n && in[lit_end] == in[lit_end + 1] && in[lit_end] == in[lit_end + 2]) break;
                ++lit_end;
            }
            if (lit_end == run_start) lit_end = run_start + 1;
            out.push_back((uint8_t)(int8_t)(-(int)(lit_end - run_start)));
            out.insert(out.end(), in.begin() + run_start, in.begin() + lit_end);
            run_start = lit_end;
        }
    }
}

int lines_per_block(int compression) {
    switch (compression) {
        case 0: case 1: case 2: return 1;
        case 3: return 16;
        case 4: return 32;
        case 5: return 16;
        case 6: case 7: return 32;
        case 8: return 32;
        case 9: return 256;
        default: return -1;
    }
}

bool parse_header(Reader& r, Part& part) {
    bool any = false;
    for (;;) {
        std::string name = r.cstr();
        if (!r.ok) return false;
        if (name.empty()) break;
        any = true;
        std::string type = r.cstr();
        const int32_t size = r.i32();
        if (!r.ok || size < 0 || !r.need((size_t)size)) return false;
        Reader a{r.p + r.pos, (size_t)size, 0};
        r.pos += (size_t)size;
        if (name == "channels" && type == "chlist") {
            for (;;) {
                std::string cn = a.cstr();
                if (!a.ok) return false;
                if (cn.empty()) break;
                Channel c;
                c.name = cn;
                c.type = a.i32();
                c.plinear = a.u8();
                a.u8(); a.u8(); a.u8();
                c.xs = a.i32();
                c.ys = a.i32();
                if (!a.ok) return false;
                part.channels.push_back(c);
            }
        } else if (name == "compression" && type == "compression") {
            part.compression = a.u8();
        } else if (name == "dataWindow" && type == "box2i") {
            part.xmin = a.i32(); part.ymin = a.i32(); part.xmax = a.i32(); part.ymax = a.i32();
            part.has_data_window = a.ok;
        } else if (name == "tiles" && type == "tiledesc") {
            part.tile_w = (uint32_t)a.i32();
            part.tile_h = (uint32_t)a.i32();
            part.tile_mode = a.u8();
        } else if (name == "chunkCount" && type == "int") {
            part.chunk_count = a.i32();
        } else if (name == "type" && type == "string") {
            part.type.assign((const char*)a.p, a.n);
        }
        if (!a.ok) return false;
    }
    return any;
}

bool ieq(const std::string& a, const char* b) {
    size_t n = std::strlen(b);
    if (a.size() != n) return false;
    for (size_t i = 0; i < n; ++i) {
        char x = a[i], y = b[i];
        if (x >= 'a' && x <= 'z') x = (char)(x - 32);
        if (y >= 'a' && y <= 'z') y = (char)(y - 32);
        if (x != y) return false;
    }
    return true;
}

// indices of the R, G, B channels of the first layer that has all three ("R" or "<layer>.R")
bool find_rgb(const Part& p, int idx[3]) {
    // bare names first
    idx[0] = idx[1] = idx[2] = -1;
    for (size_t i = 0; i < p.channels.size(); ++i) {
        if (ieq(p.channels[i].name, "R")) idx[0] = (int)i;
        if (ieq(p.channels[i].name, "G")) idx[1] = (int)i;
        if (ieq(p.channels[i].name, "B")) idx[2] = (int)i;
    }
    if (idx[0] >= 0 && idx[1] >= 0 && idx[2] >= 0) return true;
    for (size_t i = 0; i < p.channels.size(); ++i) {
        const std::string& n = p.channels[i].name;
        const size_t dot = n.rfind('.');
        if (dot == std::string::npos) continue;
        if (!ieq(n.substr(dot + 1), "R")) continue;
        const std::string prefix = n.substr(0, dot + 1);
        int g = -1, b = -1;
        for (size_t j = 0; j < p.channels.size(); ++j) {
            const std::string& m = p.channels[j].name;
            if (m.size() == prefix.size() + 1 && m.compare(0, prefix.size(), prefix) == 0) {
                if (ieq(m.substr(prefix.size()), "G")) g = (int)j;
                if (ieq(m.substr(prefix.size()), "B")) b = (int)j;
            }
        }
        if (g >= 0 && b >= 0) {
            idx[0] = (int)i; idx[1] = g; idx[2] = b;
            return true;
        }
    }
    return false;
}

int set_err(int code, const std::string& msg) {
    g_exr_error = msg;
    return code;
}

// decode one chunk's pixel bytes into `raw` (uncompressed size `expect`)
// B44(A), channels flagged pLinear: half -> half(8 log x); non-finite and negative values -> 0.  (The published table is
// generated with the library's own half type; this one rounds 8 log(x), evaluated in double, to the nearest half.)
const uint16_t* b44_log_table() {
    static const std::vector<uint16_t> table = [] {
        std::vector<uint16_t> t(65536, 0);
        for (uint32_t i = 0; i < 65536; ++i) {
            if ((i & 0x7c00u) == 0x7c00u || (i & 0x8000u)) continue;
            const float x = half_to_float((uint16_t)i);
            t[i] = x > 0.0f ? float_to_half((float)(8.0 * std::log((double)x))) : float_to_half(-INFINITY);
        }
        return t;
    }();
    return table.data();
}

bool decode_block(int compression, const uint8_t* data, size_t size, size_t expect, std::vector<uint8_t>& raw,
                  std::vector<uint8_t>& scratch, const std::vector<int>& types, const std::vector<int>& plinear, int bw, int bh) {
    if (compression == 0 || size == expect) {  // NONE, or stored raw because compression did not help
        if (size != expect) return false;
        raw.assign(data, data + size);
        return true;
    }
    if (compression == 1) {
        if (!rle_decode(data, size, raw, expect)) return false;
        unfilter(raw, scratch);
        return true;
    }
    if (compression == 2 || compression == 3) {
        raw.resize(expect);
        uLongf out_n = (uLongf)expect;
        if (uncompress(raw.data(), &out_n, data, (uLong)size) != Z_OK || out_n != expect) return false;
        unfilter(raw, scratch);
        return true;
    }
    if (compression == 4) return exr::piz_decode(data, size, types, bw, bh, raw) && raw.size() == expect;
    if (compression == 5) return exr::pxr24_decode(data, size, types, bw, bh, raw) && raw.size() == expect;
    if (compression == 6 || compression == 7) {
        std::vector<const uint16_t*> to_linear(types.size(), nullptr);
        for (size_t c = 0; c < types.size(); ++c)
            if (types[c] == 1 && plinear[c]) to_linear[c] = b44_log_table();
        return exr::b44_decode(data, size, types, to_linear, bw, bh, raw) && raw.size() == expect;
    }
    return false;
}

float sample_value(const uint8_t* p, int type) {
    if (type == 1) return half_to_float((uint16_t)(p[0] | (p[1] << 8)));
    uint32_t v = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
    if (type == 0) return (float)v;
    float f;
    std::memcpy(&f, &v, 4);
    return f;
}

int load_from_memory(const uint8_t* bytes, size_t n, float** out_rgb, uint32_t* out_w, uint32_t* out_h) {
    Reader r{bytes, n, 0};
    if (r.i32() != 20000630) return set_err(IR_EIO, "not an OpenEXR file (bad magic)");
    const uint32_t version = (uint32_t)r.i32();
    if ((version & 0xff) != 2) return set_err(IR_EIO, "unsupported OpenEXR version");
    const bool single_tiled = version & 0x200u;
    const bool deep = version & 0x800u;
    const bool multipart = version & 0x1000u;
    if (deep) return set_err(IR_EIO, "deep EXR files are not supported");
    std::vector<Part> parts;
    for (;;) {
        Part p;
        if (!parse_header(r, p)) {
            if (!r.ok) return set_err(IR_EIO, "truncated or malformed EXR header");
            break;  // empty header terminates a multipart header list
        }
        if (!multipart) p.tiled = single_tiled;
        else p.tiled = (std::string(p.type.c_str()) == "tiledimage");
        parts.push_back(p);
        if (!multipart) break;
    }
    if (parts.empty()) return set_err(IR_EIO, "EXR file has no header");

    // chunk counts and offset tables, in part order
    std::vector<std::vector<uint64_t>> tables(parts.size());
    for (size_t pi = 0; pi < parts.size(); ++pi) {
        Part& p = parts[pi];
        if (!p.has_data_window || p.xmax < p.xmin || p.ymax < p.ymin) return set_err(IR_EIO, "EXR part without a valid dataWindow");
        const int64_t w = (int64_t)p.xmax - p.xmin + 1, h = (int64_t)p.ymax - p.ymin + 1;
        if (w > 65536 || h > 65536) return set_err(IR_ELIMIT, "EXR image larger than 65536 in one dimension");
        int64_t chunks;
        if (p.tiled) {
            if (p.tile_w == 0 || p.tile_h == 0) return set_err(IR_EIO, "tiled EXR part without a tile description");
            // a tile edge beyond the largest supported image edge (or with the sign bit set) cannot address the data window
            if (p.tile_w > 65536u || p.tile_h > 65536u) return set_err(IR_ELIMIT, "EXR tile size larger than 65536");
            if (multipart && p.chunk_count > 0) {
                chunks = p.chunk_count;
            } else {
                if ((p.tile_mode & 0xf) != 0) return set_err(IR_EIO, "mip/rip-mapped tiled EXR needs chunkCount (multipart) to be read");
                chunks = ((w + p.tile_w - 1) / p.tile_w) * ((h + p.tile_h - 1) / p.tile_h);
            }
        } else {
            const int lpb = lines_per_block(p.compression);
            if (lpb < 0) return set_err(IR_EIO, "unknown EXR compression");
            chunks = (h + lpb - 1) / lpb;
            if (multipart && p.chunk_count > 0) chunks = p.chunk_count;
        }
        // every chunk has an 8-byte table entry: a count the rest of the file cannot hold is a corrupt header
        if (chunks <= 0 || (uint64_t)chunks > (uint64_t)(n - r.pos) / 8) return set_err(IR_EIO, "truncated EXR offset table");
        tables[pi].resize((size_t)chunks);
        for (auto& o : tables[pi]) o = r.u64();
        if (!r.ok) return set_err(IR_EIO, "truncated EXR offset table");
    }

    // first part that has an RGB layer (exr.rs:27: read_first_rgba_layer_from_file)
    int rgb[3];
    size_t sel = parts.size();
    for (size_t pi = 0; pi < parts.size(); ++pi)
        if (find_rgb(parts[pi], rgb)) {
            sel = pi;
            break;
        }
    if (sel == parts.size()) return set_err(IR_EIO, "EXR file has no layer with R, G and B channels");
    const Part& p = parts[sel];
    if (p.compression > 7) {
        static const char* names[] = {"NONE", "RLE", "ZIPS", "ZIP", "PIZ", "PXR24", "B44", "B44A", "DWAA", "DWAB"};
        return set_err(IR_EIO, std::string("EXR compression ") + (p.compression < 10 ? names[p.compression] : "?") + " is not supported");
    }
    for (const Channel& c : p.channels)
        if (c.xs != 1 || c.ys != 1) return set_err(IR_EIO, "sub-sampled EXR channels are not supported");
    const uint32_t W = (uint32_t)(p.xmax - p.xmin + 1), H = (uint32_t)(p.ymax - p.ymin + 1);
    size_t bytes_per_pixel = 0;
    std::vector<size_t> chan_offset(p.channels.size());
    for (size_t c = 0; c < p.channels.size(); ++c) {
        chan_offset[c] = bytes_per_pixel;  // offset in units of "bytes per pixel column", scaled by width below
        bytes_per_pixel += p.channels[c].bytes();
    }
    float* img = (float*)std::malloc((size_t)W * H * 3 * sizeof(float));
    if (!img) return set_err(IR_ENOMEM, "out of memory for the EXR image");
    std::unique_ptr<float, void (*)(void*)> img_owner(img, std::free);  // released to the caller on success only
    std::memset(img, 0, (size_t)W * H * 3 * sizeof(float));
    std::vector<uint8_t> raw, scratch;
    std::vector<int> types, plinear;
    for (const Channel& c : p.channels) {
        types.push_back(c.type);
        plinear.push_back(c.plinear);
    }
    const int lpb = p.tiled ? 0 : lines_per_block(p.compression);

    auto bail = [&](const char* m) { return set_err(IR_EIO, m); };
    for (uint64_t off : tables[sel]) {
        Reader c{bytes, n, (size_t)off};
        if (off >= n) return bail("EXR chunk offset outside the file");
        if (multipart) {
            const int32_t part_no = c.i32();
            if ((size_t)part_no != sel) return bail("EXR chunk belongs to another part");
        }
        int x0, y0, bw, bh;
        if (p.tiled) {
            const int32_t tx = c.i32(), ty = c.i32(), lx = c.i32(), ly = c.i32();
            if (!c.ok) return bail("truncated EXR tile header");
            if (lx != 0 || ly != 0) continue;  // only the full-resolution level
            // origins in 64 bits: tx * tile_w overflows an int for crafted tile coordinates
            const int64_t x064 = (int64_t)tx * (int64_t)p.tile_w, y064 = (int64_t)ty * (int64_t)p.tile_h;
            if (tx < 0 || ty < 0 || x064 >= (int64_t)W || y064 >= (int64_t)H) return bail("EXR tile outside the data window");
            x0 = (int)x064;
            y0 = (int)y064;
            bw = (int)std::min<int64_t>((int64_t)p.tile_w, (int64_t)W - x064);
            bh = (int)std::min<int64_t>((int64_t)p.tile_h, (int64_t)H - y064);
        } else {
            const int32_t y = c.i32();
            if (!c.ok) return bail("truncated EXR chunk header");
            x0 = 0;
            y0 = y - p.ymin;
            if (y0 < 0 || y0 >= (int)H) return bail("EXR scanline outside the data window");
            bw = (int)W;
            bh = std::min<int>(lpb, (int)H - y0);
        }
        const int32_t size = c.i32();
        if (!c.ok || size < 0 || !c.need((size_t)size)) return bail("truncated EXR chunk");
        const size_t expect = (size_t)bw * bh * bytes_per_pixel;
        if (!decode_block(p.compression, bytes + c.pos, (size_t)size, expect, raw, scratch, types, plinear, bw, bh)) return bail("corrupt EXR chunk data");
        // lines are channel-planar: for each line, for each channel (file order), bw samples
        for (int ly = 0; ly < bh; ++ly) {
            const uint8_t* line = raw.data() + (size_t)ly * bw * bytes_per_pixel;
            for (int k = 0; k < 3; ++k) {
                const Channel& ch = p.channels[rgb[k]];
                const uint8_t* src = line + chan_offset[rgb[k]] * bw;
                float* dst = img + ((size_t)(y0 + ly) * W + x0) * 3 + k;
                const size_t bs = ch.bytes();
                for (int x = 0; x < bw; ++x) dst[3 * (size_t)x] = sample_value(src + bs * x, ch.type);
            }
        }
    }
    *out_rgb = img_owner.release();
    *out_w = W;
    *out_h = H;
    return IR_OK;
}

void put_i32(std::vector<uint8_t>& o, int32_t v) {
    for (int i = 0; i < 4; ++i) o.push_back((uint8_t)((uint32_t)v >> (8 * i)));
}
void put_str(std::vector<uint8_t>& o, const char* s) {
    while (*s) o.push_back((uint8_t)*s++);
    o.push_back(0);
}
void put_attr(std::vector<uint8_t>& o, const char* name, const char* type, const std::vector<uint8_t>& v) {
    put_str(o, name);
    put_str(o, type);
    put_i32(o, (int32_t)v.size());
    o.insert(o.end(), v.begin(), v.end());
}

// Nothing thrown by the containers below (bad_alloc / length_error on a hostile header) may cross the C ABI.
template <class F>
int guarded(const char* who, F&& body) {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        return set_err(IR_ENOMEM, std::string(who) + ": out of host memory");
    } catch (const std::exception& e) {
        return set_err(IR_EIO, std::string(who) + ": " + e.what());
    } catch (...) {
        return set_err(IR_EIO, std::string(who) + ": unexpected failure");
    }
}

}  // namespace

extern "C" {

const char* ir_exr_last_error(void) { return g_exr_error.c_str(); }

int ir_exr_load(const char* path, float** out_rgb, uint32_t* out_w, uint32_t* out_h) {
    if (!path || !out_rgb || !out_w || !out_h) return set_err(IR_EINVAL, "ir_exr_load: NULL argument");
    *out_rgb = nullptr;
    *out_w = *out_h = 0;
    return guarded("ir_exr_load", [&]() -> int {
        FILE* f = std::fopen(path, "rb");
        if (!f) return set_err(IR_EIO, std::string("cannot open ") + path);
        std::vector<uint8_t> bytes;
        try {
            uint8_t buf[1 << 16];
            size_t k;
            while ((k = std::fread(buf, 1, sizeof(buf), f)) > 0) bytes.insert(bytes.end(), buf, buf + k);
        } catch (...) {
            std::fclose(f);
            throw;
        }
        std::fclose(f);
        const int rc = load_from_memory(bytes.data(), bytes.size(), out_rgb, out_w, out_h);
        return rc;
    });
}

int ir_exr_save(const char* path, const float* rgb, uint32_t w, uint32_t h, int compression, int pixel_type) {
    if (!path || !rgb || w == 0 || h == 0) return set_err(IR_EINVAL, "ir_exr_save: NULL or empty image");
    if (compression < 0 || compression > 7)
        return set_err(IR_EINVAL, "ir_exr_save: compression must be 0 (NONE), 1 (RLE), 2 (ZIPS), 3 (ZIP), 4 (PIZ), 5 (PXR24), 6 (B44) or 7 (B44A)");
    if (pixel_type != 1 && pixel_type != 2) return set_err(IR_EINVAL, "ir_exr_save: pixel_type must be 1 (HALF) or 2 (FLOAT)");
    return guarded("ir_exr_save", [&]() -> int {
    std::vector<uint8_t> o;
    put_i32(o, 20000630);
    put_i32(o, 2);
    {
        std::vector<uint8_t> v;
        const char* names[3] = {"B", "G", "R"};
        for (int c = 0; c < 3; ++c) {
            put_str(v, names[c]);
            put_i32(v, pixel_type);
            v.push_back(0); v.push_back(0); v.push_back(0); v.push_back(0);
            put_i32(v, 1);
            put_i32(v, 1);
        }
        v.push_back(0);
        put_attr(o, "channels", "chlist", v);
    }
    put_attr(o, "compression", "compression", std::vector<uint8_t>{(uint8_t)compression});
    {
        std::vector<uint8_t> v;
        put_i32(v, 0); put_i32(v, 0); put_i32(v, (int32_t)w - 1); put_i32(v, (int32_t)h - 1);
        put_attr(o, "dataWindow", "box2i", v);
        put_attr(o, "displayWindow", "box2i", v);
    }
    put_attr(o, "lineOrder", "lineOrder", std::vector<uint8_t>{0});
    {
        std::vector<uint8_t> v(4);
        const float one = 1.0f;
        std::memcpy(v.data(), &one, 4);
        put_attr(o, "pixelAspectRatio", "float", v);
        put_attr(o, "screenWindowWidth", "float", v);
        std::vector<uint8_t> z(8, 0);
        put_attr(o, "screenWindowCenter", "v2f", z);
    }
    o.push_back(0);  // end of header

    const int lpb = lines_per_block(compression);
    const uint32_t chunks = (h + lpb - 1) / lpb;
    const size_t table_pos = o.size();
    o.resize(o.size() + (size_t)chunks * 8, 0);
    const size_t bs = pixel_type == 1 ? 2 : 4;
    std::vector<uint8_t> raw, filt, comp;
    for (uint32_t ci = 0; ci < chunks; ++ci) {
        const uint32_t y0 = ci * lpb, bh = std::min<uint32_t>(lpb, h - y0);
        raw.clear();
        for (uint32_t ly = 0; ly < bh; ++ly)
            for (int c = 0; c < 3; ++c) {  // file order B, G, R
                const int src = 2 - c;
                for (uint32_t x = 0; x < w; ++x) {
                    const float f = rgb[((size_t)(y0 + ly) * w + x) * 3 + src];
                    if (bs == 2) {
                        const uint16_t hv = float_to_half(f);
                        raw.push_back((uint8_t)(hv & 0xff));
                        raw.push_back((uint8_t)(hv >> 8));
                    } else {
                        uint32_t u;
                        std::memcpy(&u, &f, 4);
                        for (int i = 0; i < 4; ++i) raw.push_back((uint8_t)(u >> (8 * i)));
                    }
                }
            }
        const uint8_t* payload = raw.data();
        size_t payload_n = raw.size();
        if (compression == 1) {
            filter(raw.data(), raw.size(), filt);
            rle_encode(filt, comp);
            if (comp.size() < raw.size()) {
                payload = comp.data();
                payload_n = comp.size();
            }
        } else if (compression >= 4) {
            const std::vector<int> types(3, pixel_type);
            const bool ok = compression == 4   ? exr::piz_encode(raw.data(), types, (int)w, (int)bh, comp)
                            : compression == 5 ? exr::pxr24_encode(raw.data(), types, (int)w, (int)bh, comp)
                                               : exr::b44_encode(raw.data(), types, (int)w, (int)bh, compression == 7, comp);
            if (!ok) return set_err(IR_EIO, "EXR block encoder failed");
            if (comp.size() < raw.size()) {
                payload = comp.data();
                payload_n = comp.size();
            }
        } else if (compression >= 2) {
            filter(raw.data(), raw.size(), filt);
            uLongf cn = compressBound((uLong)filt.size());
            comp.resize(cn);
            if (compress2(comp.data(), &cn, filt.data(), (uLong)filt.size(), 6) != Z_OK) return set_err(IR_EIO, "zlib compress failed");
            if (cn < raw.size()) {
                payload = comp.data();
                payload_n = cn;
            }
        }
        const uint64_t off = o.size();
        for (int i = 0; i < 8; ++i) o[table_pos + (size_t)ci * 8 + i] = (uint8_t)(off >> (8 * i));
        put_i32(o, (int32_t)y0);
        put_i32(o, (int32_t)payload_n);
        o.insert(o.end(), payload, payload + payload_n);
    }
    FILE* f = std::fopen(path, "wb");
    if (!f) return set_err(IR_EIO, std::string("cannot create ") + path);
    const size_t wr = std::fwrite(o.data(), 1, o.size(), f);
    const int rc = std::fclose(f);
    if (wr != o.size() || rc != 0) return set_err(IR_EIO, std::string("short write to ") + path);
    return IR_OK;
    });
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// PNG writer for the resolved frame ("save image" is a reference TODO, README.md:104): 8-bit RGB, one zlib stream,
// filter 0 on every row.  argb: 0xAARRGGBB as ir_get_output_argb returns it, row 0 = BOTTOM row (renderer.rs:164-169;
// the frontend flips it when presenting), written top-down.
// ---------------------------------------------------------------------------------------------------------------
namespace {
void png_chunk(std::vector<uint8_t>& o, const char type[4], const uint8_t* data, size_t n) {
    const uint32_t len = (uint32_t)n;
    const uint8_t l[4] = {(uint8_t)(len >> 24), (uint8_t)(len >> 16), (uint8_t)(len >> 8), (uint8_t)len};
    o.insert(o.end(), l, l + 4);
    const size_t at = o.size();
    o.insert(o.end(), type, type + 4);
    if (n) o.insert(o.end(), data, data + n);
    const uint32_t c = (uint32_t)crc32(0L, o.data() + at, (uInt)(n + 4));
    const uint8_t cb[4] = {(uint8_t)(c >> 24), (uint8_t)(c >> 16), (uint8_t)(c >> 8), (uint8_t)c};
    o.insert(o.end(), cb, cb + 4);
}
}  // namespace

int ir_png_save(const char* path, const uint32_t* argb, uint32_t w, uint32_t h) {
    if (!path || !argb || w == 0 || h == 0) return set_err(IR_EINVAL, "ir_png_save: NULL or empty image");
    if ((uint64_t)w * h > (1ull << 28)) return set_err(IR_ELIMIT, "ir_png_save: image too large");
    return guarded("ir_png_save", [&]() -> int {
    std::vector<uint8_t> raw;
    raw.reserve((size_t)h * (1 + 3 * (size_t)w));
    for (uint32_t y = 0; y < h; ++y) {
        const uint32_t* row = argb + (size_t)(h - 1 - y) * w;
        raw.push_back(0);  // filter type None
        for (uint32_t x = 0; x < w; ++x) {
            raw.push_back((uint8_t)(row[x] >> 16));
            raw.push_back((uint8_t)(row[x] >> 8));
            raw.push_back((uint8_t)row[x]);
        }
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return set_err(IR_EIO, "ir_png_save: deflate failed");
    std::vector<uint8_t> o = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    const uint8_t ihdr[13] = {(uint8_t)(w >> 24), (uint8_t)(w >> 16), (uint8_t)(w >> 8), (uint8_t)w,
                              (uint8_t)(h >> 24), (uint8_t)(h >> 16), (uint8_t)(h >> 8), (uint8_t)h,
                              8, 2, 0, 0, 0};  // 8 bits, colour type 2 (RGB), deflate, adaptive filtering, no interlace
    png_chunk(o, "IHDR", ihdr, sizeof(ihdr));
    png_chunk(o, "IDAT", z.data(), zlen);
    png_chunk(o, "IEND", nullptr, 0);
    FILE* f = fopen(path, "wb");
    if (!f) return set_err(IR_EIO, std::string("ir_png_save: cannot open ") + path);
    const bool ok = fwrite(o.data(), 1, o.size(), f) == o.size();
    if (fclose(f) != 0 || !ok) return set_err(IR_EIO, std::string("ir_png_save: write failed: ") + path);
    return IR_OK;
    });
}
