// exr_codecs.cpp — the OpenEXR block codecs beyond zlib/RLE: PIZ (compression 4) and PXR24 (5), which lossless HDR
// environment maps are usually stored with, and the lossy fixed-rate B44 / B44A (6, 7).  Written from the published description of the
// formats (OpenEXR "Technical Introduction" + the layout of the reference implementation's chunks); no OpenEXR,
// tinyexr or `exr` crate source exists in this environment and no file written by another implementation is
// available here, so these codecs are pinned only by their own round trips and hand-made vectors
// (tests/test_exr.py): "parity unpinned" against third-party writers.
//
// A block arrives as `bh` lines of `bw` pixels; the decoded byte layout is the same as for every other codec:
// for each line, for each channel (file order), bw little-endian samples.
//
// PXR24: zlib over byte planes.  Per line and channel the samples are delta-coded (difference to the previous
//   sample of the line, starting from 0) and split into planes, most significant byte first: HALF = 2 planes of the
//   16-bit value, UINT = 4 planes, FLOAT = 3 planes of the value rounded to 24 bits (sign, 8 exponent, 15 mantissa).
// PIZ: u16 minNonZero, u16 maxNonZero, bitmap bytes [min..max] of the 16-bit values that occur (value 0 always
//   counts as present), i32 length, Huffman data.  The 16-bit values (HALF = 1, FLOAT/UINT = 2 per sample, low half
//   first) are laid out per channel as a bh x bw*size plane, mapped through the bitmap's rank table, transformed by a
//   2D Haar-like wavelet per channel and 16-bit component (14-bit arithmetic when the largest rank < 2^14, modulo
//   2^16 arithmetic otherwise), then Huffman-coded as one stream with a run-length escape symbol.
//   Huffman block: u32 im, iM, tableLength, nBits, reserved; code lengths for symbols im..iM packed 6 bits each
//   (59..62 = run of 2..5 zero lengths, 63 + 8 bits = run of 6..261); canonical codes are assigned from the longest
//   length (code 0) upwards; bits are written most significant first; symbol iM is the run-length escape: it is
//   followed by an 8-bit count of additional copies of the previous symbol.
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <queue>
#include <vector>

#include "exr_codecs.h"

namespace exr {
namespace {

constexpr int kHufEncSize = (1 << 16) + 1;  // 65536 values + the run-length symbol
constexpr int kShortZeroRun = 59, kLongZeroRun = 63, kShortestLongRun = 2 + kLongZeroRun - kShortZeroRun;
constexpr int kLongestLongRun = 255 + kShortestLongRun;

// ---------------------------------------------------------------------------------------------- bits
struct BitWriter {
    std::vector<uint8_t>& out;
    uint64_t c = 0;
    int lc = 0;
    uint64_t total = 0;
    explicit BitWriter(std::vector<uint8_t>& o) : out(o) {}
    void put(int n, uint64_t bits) {
        c = (c << n) | (bits & ((n >= 64) ? ~0ull : ((1ull << n) - 1)));
        lc += n;
        total += (uint64_t)n;
        while (lc >= 8) {
            lc -= 8;
            out.push_back((uint8_t)(c >> lc));
        }
    }
    void flush() {
        if (lc) out.push_back((uint8_t)((c << (8 - lc)) & 0xff));
        lc = 0;
    }
};
struct BitReader {
    const uint8_t* p;
    size_t n, pos = 0;
    uint64_t c = 0;
    int lc = 0;
    bool ok = true;
    uint32_t get(int k) {
        while (lc < k) {
            if (pos >= n) {
                ok = false;
                return 0;
            }
            c = (c << 8) | p[pos++];
            lc += 8;
        }
        lc -= k;
        return (uint32_t)((c >> lc) & ((1ull << k) - 1));
    }
};

// ---------------------------------------------------------------------------------------------- Huffman
// canonical codes from lengths: the longest codes get the numerically smallest values
void canonical_codes(const std::vector<uint8_t>& len, std::vector<uint64_t>& code) {
    uint64_t n[59] = {0};
    for (uint8_t l : len) n[l]++;
    uint64_t c = 0;
    for (int i = 58; i > 0; --i) {
        const uint64_t nc = (c + n[i]) >> 1;
        n[i] = c;
        c = nc;
    }
    code.assign(len.size(), 0);
    for (size_t i = 0; i < len.size(); ++i)
        if (len[i]) code[i] = n[len[i]]++;
}

// code lengths of an optimal prefix code for freq (symbols with freq 0 get length 0)
bool huffman_lengths(const std::vector<uint64_t>& freq, std::vector<uint8_t>& len) {
    struct Node {
        uint64_t f;
        int id;
    };
    auto cmp = [](const Node& a, const Node& b) { return a.f != b.f ? a.f > b.f : a.id > b.id; };
    std::priority_queue<Node, std::vector<Node>, decltype(cmp)> heap(cmp);
    const int n = (int)freq.size();
    std::vector<int> parent;
    parent.reserve(2 * (size_t)n);
    for (int i = 0; i < n; ++i) {
        parent.push_back(-1);
        if (freq[i]) heap.push(Node{freq[i], i});
    }
    len.assign((size_t)n, 0);
    if (heap.size() == 1) {
        len[(size_t)heap.top().id] = 1;
        return true;
    }
    while (heap.size() > 1) {
        const Node a = heap.top();
        heap.pop();
        const Node b = heap.top();
        heap.pop();
        const int id = (int)parent.size();
        parent.push_back(-1);
        parent[(size_t)a.id] = id;
        parent[(size_t)b.id] = id;
        heap.push(Node{a.f + b.f, id});
    }
    for (int i = 0; i < n; ++i) {
        if (!freq[i]) continue;
        int l = 0;
        for (int k = i; parent[(size_t)k] >= 0; k = parent[(size_t)k]) ++l;
        if (l > 58) return false;
        len[(size_t)i] = (uint8_t)l;
    }
    return true;
}

bool huf_compress(const std::vector<uint16_t>& in, std::vector<uint8_t>& out) {
    out.clear();
    if (in.empty()) return true;
    std::vector<uint64_t> freq((size_t)kHufEncSize, 0);
    for (uint16_t v : in) freq[v]++;
    int im = 0, iM = kHufEncSize - 2;
    while (!freq[(size_t)im]) ++im;
    while (!freq[(size_t)iM]) --iM;
    ++iM;  // the run-length symbol sits right after the largest value that occurs
    freq[(size_t)iM] = 1;
    std::vector<uint8_t> len;
    if (!huffman_lengths(freq, len)) return false;
    std::vector<uint64_t> code;
    canonical_codes(len, code);

    std::vector<uint8_t> table;
    {
        BitWriter w(table);
        for (int i = im; i <= iM; ++i) {
            const int l = len[(size_t)i];
            if (l == 0) {
                int zerun = 1;
                while (i < iM && zerun < kLongestLongRun) {
                    if (len[(size_t)i + 1] > 0) break;
                    ++i;
                    ++zerun;
                }
                if (zerun >= 2) {
                    if (zerun >= kShortestLongRun) {
                        w.put(6, kLongZeroRun);
                        w.put(8, (uint64_t)(zerun - kShortestLongRun));
                    } else {
                        w.put(6, (uint64_t)(kShortZeroRun + zerun - 2));
                    }
                    continue;
                }
            }
            w.put(6, (uint64_t)l);
        }
        w.flush();
    }
    std::vector<uint8_t> data;
    BitWriter w(data);
    const int rl = len[(size_t)iM];
    auto send = [&](uint16_t s, int run) {  // symbol s followed by `run` more copies
        const int l = len[s];
        if (l + rl + 8 < l * run) {
            w.put(l, code[s]);
            w.put(rl, code[(size_t)iM]);
            w.put(8, (uint64_t)run);
        } else {
            for (int k = 0; k <= run; ++k) w.put(l, code[s]);
        }
    };
    uint16_t s = in[0];
    int cs = 0;
    for (size_t i = 1; i < in.size(); ++i) {
        if (s == in[i] && cs < 255) {
            ++cs;
        } else {
            send(s, cs);
            cs = 0;
        }
        s = in[i];
    }
    send(s, cs);
    const uint64_t nbits = w.total;
    w.flush();
    auto put32 = [&](uint32_t v) {
        for (int i = 0; i < 4; ++i) out.push_back((uint8_t)(v >> (8 * i)));
    };
    put32((uint32_t)im);
    put32((uint32_t)iM);
    put32((uint32_t)table.size());
    put32((uint32_t)nbits);
    put32(0);
    out.insert(out.end(), table.begin(), table.end());
    out.insert(out.end(), data.begin(), data.end());
    return true;
}

bool huf_uncompress(const uint8_t* p, size_t n, std::vector<uint16_t>& out, size_t count) {
    out.clear();
    if (count == 0) return true;
    if (n < 20) return false;
    auto get32 = [&](size_t o) { return (uint32_t)p[o] | ((uint32_t)p[o + 1] << 8) | ((uint32_t)p[o + 2] << 16) | ((uint32_t)p[o + 3] << 24); };
    const uint32_t im = get32(0), iM = get32(4), table_len = get32(8), nbits = get32(12);
    if (im >= (uint32_t)kHufEncSize || iM >= (uint32_t)kHufEncSize || im > iM) return false;
    if ((size_t)table_len > n - 20) return false;
    std::vector<uint8_t> len((size_t)kHufEncSize, 0);
    {
        BitReader r{p + 20, table_len};
        for (uint32_t i = im; i <= iM; ++i) {
            const uint32_t l = r.get(6);
            if (!r.ok) return false;
            if (l == (uint32_t)kLongZeroRun) {
                uint32_t zerun = r.get(8) + kShortestLongRun;
                if (!r.ok || i + zerun > iM + 1) return false;
                i += zerun - 1;
            } else if (l >= (uint32_t)kShortZeroRun) {
                const uint32_t zerun = l - kShortZeroRun + 2;
                if (i + zerun > iM + 1) return false;
                i += zerun - 1;
            } else {
                len[i] = (uint8_t)l;
            }
        }
    }
    // canonical decoding tables: per length the first code and the symbols in increasing order
    uint64_t cnt[59] = {0}, first[59] = {0};
    for (uint32_t i = im; i <= iM; ++i) cnt[len[i]]++;
    {
        uint64_t c = 0;
        for (int i = 58; i > 0; --i) {
            const uint64_t nc = (c + cnt[i]) >> 1;
            first[i] = c;
            c = nc;
        }
    }
    size_t offset[60];
    offset[1] = 0;
    for (int l = 1; l < 59; ++l) offset[l + 1] = offset[l] + (size_t)cnt[l];
    std::vector<uint32_t> syms(offset[59]);
    {
        size_t fill[59];
        for (int l = 1; l < 59; ++l) fill[l] = offset[l];
        for (uint32_t i = im; i <= iM; ++i)
            if (len[i]) syms[fill[len[i]]++] = i;
    }
    const uint8_t* data = p + 20 + table_len;
    const size_t data_n = n - 20 - table_len;
    if ((uint64_t)nbits > 8ull * data_n) return false;
    out.reserve(count);
    size_t bit = 0;
    auto next_bit = [&]() -> uint32_t {
        const uint32_t b = (data[bit >> 3] >> (7 - (bit & 7))) & 1u;
        ++bit;
        return b;
    };
    while (bit < nbits) {
        uint64_t code = 0;
        int l = 0;
        uint32_t sym = 0;
        bool found = false;
        while (bit < nbits && l < 58) {
            code = (code << 1) | next_bit();
            ++l;
            if (cnt[l] && code >= first[l] && code - first[l] < cnt[l]) {
                sym = syms[offset[l] + (size_t)(code - first[l])];
                found = true;
                break;
            }
        }
        if (!found) return false;
        if (sym == iM) {  // run-length escape
            if (bit + 8 > nbits || out.empty()) return false;
            uint32_t run = 0;
            for (int k = 0; k < 8; ++k) run = (run << 1) | next_bit();
            if (out.size() + run > count) return false;
            out.insert(out.end(), run, out.back());
        } else {
            if (out.size() >= count) return false;
            out.push_back((uint16_t)sym);
        }
    }
    return out.size() == count;
}

// ---------------------------------------------------------------------------------------------- wavelet
inline void wenc14(uint16_t a, uint16_t b, uint16_t& l, uint16_t& h) {
    const int16_t as = (int16_t)a, bs = (int16_t)b;
    const int16_t ms = (int16_t)((as + bs) >> 1), ds = (int16_t)(as - bs);
    l = (uint16_t)ms;
    h = (uint16_t)ds;
}
inline void wdec14(uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) {
    const int16_t ls = (int16_t)l, hs = (int16_t)h;
    const int hi = hs, ai = ls + (hi & 1) + (hi >> 1);
    a = (uint16_t)(int16_t)ai;
    b = (uint16_t)(int16_t)(ai - hi);
}
constexpr int kAOffset = 1 << 15, kModMask = (1 << 16) - 1;
inline void wenc16(uint16_t a, uint16_t b, uint16_t& l, uint16_t& h) {
    const int ao = (a + kAOffset) & kModMask;
    int m = (ao + b) >> 1;
    int d = ao - b;
    if (d < 0) m = (m + kAOffset) & kModMask;
    d &= kModMask;
    l = (uint16_t)m;
    h = (uint16_t)d;
}
inline void wdec16(uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) {
    const int m = l, d = h;
    const int bb = (m - (d >> 1)) & kModMask;
    const int aa = (d + bb - kAOffset) & kModMask;
    b = (uint16_t)bb;
    a = (uint16_t)aa;
}

template <bool ENCODE>
void wav2(uint16_t* in, int nx, int ox, int ny, int oy, uint16_t mx) {
    const bool w14 = mx < (1 << 14);
    const int n = nx > ny ? ny : nx;
    auto enc = [&](uint16_t a, uint16_t b, uint16_t& l, uint16_t& h) { w14 ? wenc14(a, b, l, h) : wenc16(a, b, l, h); };
    auto dec = [&](uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) { w14 ? wdec14(l, h, a, b) : wdec16(l, h, a, b); };
    auto level = [&](int p, int p2) {
        uint16_t* py = in;
        uint16_t* const ey = in + (ptrdiff_t)oy * (ny - p2);
        const ptrdiff_t oy1 = (ptrdiff_t)oy * p, oy2 = (ptrdiff_t)oy * p2, ox1 = (ptrdiff_t)ox * p, ox2 = (ptrdiff_t)ox * p2;
        uint16_t i00, i01, i10, i11;
        for (; py <= ey; py += oy2) {
            uint16_t* px = py;
            uint16_t* const ex = py + (ptrdiff_t)ox * (nx - p2);
            for (; px <= ex; px += ox2) {
                uint16_t *p01 = px + ox1, *p10 = px + oy1, *p11 = p10 + ox1;
                if (ENCODE) {
                    enc(*px, *p01, i00, i01);
                    enc(*p10, *p11, i10, i11);
                    enc(i00, i10, *px, *p10);
                    enc(i01, i11, *p01, *p11);
                } else {
                    dec(*px, *p10, i00, i10);
                    dec(*p01, *p11, i01, i11);
                    dec(i00, i01, *px, *p01);
                    dec(i10, i11, *p10, *p11);
                }
            }
            if (nx & p) {  // odd column left over at this level
                uint16_t* p10 = px + oy1;
                if (ENCODE) enc(*px, *p10, i00, *p10); else dec(*px, *p10, i00, *p10);
                *px = i00;
            }
        }
        if (ny & p) {  // odd line left over at this level
            uint16_t* px = py;
            uint16_t* const ex = py + (ptrdiff_t)ox * (nx - p2);
            for (; px <= ex; px += ox2) {
                uint16_t* p01 = px + ox1;
                if (ENCODE) enc(*px, *p01, i00, *p01); else dec(*px, *p01, i00, *p01);
                *px = i00;
            }
        }
    };
    if (ENCODE) {
        int p = 1, p2 = 2;
        while (p2 <= n) {
            level(p, p2);
            p = p2;
            p2 <<= 1;
        }
    } else {
        int p = 1;
        while (p <= n) p <<= 1;
        p >>= 1;
        int p2 = p;
        p >>= 1;
        while (p >= 1) {
            level(p, p2);
            p2 = p;
            p >>= 1;
        }
    }
}

// plane layout of a PIZ block: channel c starts at start[c]; bh rows of bw * size16[c] values
size_t piz_planes(const std::vector<int>& types, int bw, int bh, std::vector<size_t>& start, std::vector<int>& size16) {
    size_t total = 0;
    start.clear();
    size16.clear();
    for (int t : types) {
        start.push_back(total);
        size16.push_back(t == 1 ? 1 : 2);
        total += (size_t)bw * bh * size16.back();
    }
    return total;
}

uint32_t float_to_f24(uint32_t i) {  // sign, 8 exponent, 15 mantissa bits; round to nearest, never into infinity
    const uint32_t s = i & 0x80000000u, e = i & 0x7f800000u, m = i & 0x007fffffu;
    uint32_t r;
    if (e == 0x7f800000u) {
        if (m) {
            uint32_t mm = m >> 8;
            r = (e >> 8) | mm | (mm == 0 ? 1u : 0u);  // keep a NaN a NaN
        } else {
            r = e >> 8;
        }
    } else {
        r = ((e | m) + (m & 0x00000080u)) >> 8;
        if (r >= 0x7f8000u) r = (e | m) >> 8;  // rounding would overflow into infinity: truncate
    }
    return (s >> 8) | r;
}

}  // namespace

// ---------------------------------------------------------------------------------------------- PIZ
bool piz_decode(const uint8_t* data, size_t size, const std::vector<int>& types, int bw, int bh, std::vector<uint8_t>& raw) {
    std::vector<size_t> start;
    std::vector<int> size16;
    const size_t total = piz_planes(types, bw, bh, start, size16);
    raw.assign(total * 2, 0);
    if (total == 0) return true;
    if (size < 8) return false;
    const uint32_t min_nz = data[0] | (data[1] << 8), max_nz = data[2] | (data[3] << 8);
    size_t pos = 4;
    std::vector<uint8_t> bitmap(8192, 0);
    if (min_nz <= max_nz) {
        if (max_nz >= 8192 || size < pos + (max_nz - min_nz + 1) + 4) return false;
        std::memcpy(&bitmap[min_nz], data + pos, max_nz - min_nz + 1);
        pos += max_nz - min_nz + 1;
    }
    std::vector<uint16_t> lut(65536, 0);
    uint32_t k = 0;
    for (uint32_t i = 0; i < 65536; ++i)
        if (i == 0 || (bitmap[i >> 3] & (1u << (i & 7)))) lut[k++] = (uint16_t)i;
    const uint16_t max_value = (uint16_t)(k - 1);
    const uint32_t length = (uint32_t)data[pos] | ((uint32_t)data[pos + 1] << 8) | ((uint32_t)data[pos + 2] << 16) | ((uint32_t)data[pos + 3] << 24);
    pos += 4;
    if ((size_t)length > size - pos) return false;
    std::vector<uint16_t> tmp;
    if (!huf_uncompress(data + pos, length, tmp, total)) return false;
    for (size_t c = 0; c < types.size(); ++c)
        for (int j = 0; j < size16[c]; ++j)
            wav2<false>(tmp.data() + start[c] + j, bw, size16[c], bh, bw * size16[c], max_value);
    for (uint16_t& v : tmp) v = lut[v];
    // planes -> lines: for each line, for each channel, bw samples (little endian)
    uint8_t* o = raw.data();
    for (int y = 0; y < bh; ++y)
        for (size_t c = 0; c < types.size(); ++c) {
            const uint16_t* row = tmp.data() + start[c] + (size_t)y * bw * size16[c];
            for (int x = 0; x < bw * size16[c]; ++x) {
                *o++ = (uint8_t)(row[x] & 0xff);
                *o++ = (uint8_t)(row[x] >> 8);
            }
        }
    return true;
}

bool piz_encode(const uint8_t* raw, const std::vector<int>& types, int bw, int bh, std::vector<uint8_t>& out) {
    std::vector<size_t> start;
    std::vector<int> size16;
    const size_t total = piz_planes(types, bw, bh, start, size16);
    std::vector<uint16_t> tmp(total);
    const uint8_t* in = raw;
    for (int y = 0; y < bh; ++y)
        for (size_t c = 0; c < types.size(); ++c) {
            uint16_t* row = tmp.data() + start[c] + (size_t)y * bw * size16[c];
            for (int x = 0; x < bw * size16[c]; ++x) {
                row[x] = (uint16_t)(in[0] | (in[1] << 8));
                in += 2;
            }
        }
    std::vector<uint8_t> bitmap(8192, 0);
    for (uint16_t v : tmp) bitmap[v >> 3] |= (uint8_t)(1u << (v & 7));
    bitmap[0] &= (uint8_t)~1u;  // zero is implied
    int min_nz = 8191, max_nz = 0;
    for (int i = 0; i < 8192; ++i)
        if (bitmap[(size_t)i]) {
            min_nz = std::min(min_nz, i);
            max_nz = std::max(max_nz, i);
        }
    std::vector<uint16_t> lut(65536, 0);
    uint32_t k = 0;
    for (uint32_t i = 0; i < 65536; ++i)
        if (i == 0 || (bitmap[i >> 3] & (1u << (i & 7)))) lut[i] = (uint16_t)k++;
    const uint16_t max_value = (uint16_t)(k - 1);
    for (uint16_t& v : tmp) v = lut[v];
    for (size_t c = 0; c < types.size(); ++c)
        for (int j = 0; j < size16[c]; ++j)
            wav2<true>(tmp.data() + start[c] + j, bw, size16[c], bh, bw * size16[c], max_value);
    std::vector<uint8_t> huf;
    if (!huf_compress(tmp, huf)) return false;
    out.clear();
    out.push_back((uint8_t)(min_nz & 0xff));
    out.push_back((uint8_t)(min_nz >> 8));
    out.push_back((uint8_t)(max_nz & 0xff));
    out.push_back((uint8_t)(max_nz >> 8));
    if (min_nz <= max_nz) out.insert(out.end(), bitmap.begin() + min_nz, bitmap.begin() + max_nz + 1);
    const uint32_t length = (uint32_t)huf.size();
    for (int i = 0; i < 4; ++i) out.push_back((uint8_t)(length >> (8 * i)));
    out.insert(out.end(), huf.begin(), huf.end());
    return true;
}

// ---------------------------------------------------------------------------------------------- PXR24
bool pxr24_decode(const uint8_t* data, size_t size, const std::vector<int>& types, int bw, int bh, std::vector<uint8_t>& raw) {
    size_t planes_bytes = 0, raw_bytes = 0;
    for (int t : types) {
        planes_bytes += (size_t)bw * (t == 1 ? 2 : (t == 2 ? 3 : 4));
        raw_bytes += (size_t)bw * (t == 1 ? 2 : 4);
    }
    planes_bytes *= (size_t)bh;
    raw_bytes *= (size_t)bh;
    std::vector<uint8_t> tmp(planes_bytes ? planes_bytes : 1);
    uLongf out_n = (uLongf)planes_bytes;
    if (uncompress(tmp.data(), &out_n, data, (uLong)size) != Z_OK || out_n != planes_bytes) return false;
    raw.assign(raw_bytes, 0);
    const uint8_t* p = tmp.data();
    uint8_t* o = raw.data();
    for (int y = 0; y < bh; ++y)
        for (int t : types) {
            uint32_t pixel = 0;
            if (t == 1) {
                const uint8_t *p0 = p, *p1 = p + bw;
                p += 2 * (size_t)bw;
                for (int x = 0; x < bw; ++x) {
                    pixel += ((uint32_t)p0[x] << 8) | p1[x];
                    *o++ = (uint8_t)(pixel & 0xff);
                    *o++ = (uint8_t)((pixel >> 8) & 0xff);
                }
            } else if (t == 2) {
                const uint8_t *p0 = p, *p1 = p + bw, *p2 = p + 2 * (size_t)bw;
                p += 3 * (size_t)bw;
                for (int x = 0; x < bw; ++x) {
                    pixel += ((uint32_t)p0[x] << 24) | ((uint32_t)p1[x] << 16) | ((uint32_t)p2[x] << 8);
                    for (int i = 0; i < 4; ++i) *o++ = (uint8_t)(pixel >> (8 * i));
                }
            } else {
                const uint8_t *p0 = p, *p1 = p + bw, *p2 = p + 2 * (size_t)bw, *p3 = p + 3 * (size_t)bw;
                p += 4 * (size_t)bw;
                for (int x = 0; x < bw; ++x) {
                    pixel += ((uint32_t)p0[x] << 24) | ((uint32_t)p1[x] << 16) | ((uint32_t)p2[x] << 8) | p3[x];
                    for (int i = 0; i < 4; ++i) *o++ = (uint8_t)(pixel >> (8 * i));
                }
            }
        }
    return true;
}

bool pxr24_encode(const uint8_t* raw, const std::vector<int>& types, int bw, int bh, std::vector<uint8_t>& out) {
    std::vector<uint8_t> tmp;
    const uint8_t* in = raw;
    for (int y = 0; y < bh; ++y)
        for (int t : types) {
            const size_t base = tmp.size();
            const int planes = t == 1 ? 2 : (t == 2 ? 3 : 4);
            tmp.resize(base + (size_t)planes * bw);
            uint32_t prev = 0;
            for (int x = 0; x < bw; ++x) {
                uint32_t v;
                if (t == 1) {
                    v = (uint32_t)in[0] | ((uint32_t)in[1] << 8);
                    in += 2;
                    const uint32_t d = (v - prev) & 0xffffu;
                    prev = v;
                    tmp[base + x] = (uint8_t)(d >> 8);
                    tmp[base + bw + x] = (uint8_t)d;
                } else {
                    v = (uint32_t)in[0] | ((uint32_t)in[1] << 8) | ((uint32_t)in[2] << 16) | ((uint32_t)in[3] << 24);
                    in += 4;
                    if (t == 2) {
                        v = float_to_f24(v);
                        const uint32_t d = (v - prev) & 0xffffffu;
                        prev = v;
                        tmp[base + x] = (uint8_t)(d >> 16);
                        tmp[base + bw + x] = (uint8_t)(d >> 8);
                        tmp[base + 2 * (size_t)bw + x] = (uint8_t)d;
                    } else {
                        const uint32_t d = v - prev;
                        prev = v;
                        tmp[base + x] = (uint8_t)(d >> 24);
                        tmp[base + bw + x] = (uint8_t)(d >> 16);
                        tmp[base + 2 * (size_t)bw + x] = (uint8_t)(d >> 8);
                        tmp[base + 3 * (size_t)bw + x] = (uint8_t)d;
                    }
                }
            }
        }
    uLongf cn = compressBound((uLong)tmp.size());
    out.resize(cn);
    if (compress2(out.data(), &cn, tmp.data(), (uLong)tmp.size(), 6) != Z_OK) return false;
    out.resize(cn);
    return true;
}

// ---------------------------------------------------------------------------------------------- B44 / B44A
// Lossy, fixed rate, HALF channels only (UINT / FLOAT channels travel uncompressed).  The block's data is channel after
// channel (not line by line); a HALF channel is cut into 4x4 pixel blocks, row-major, the right column / bottom row
// repeated where the block hangs over the edge.  The 16 half bit patterns s[i] are mapped to integers t[i] that order
// like the numbers they stand for (negative: ~s, else s | 0x8000; NaN and infinity become 0x8000 = zero), and a block is
// 14 bytes: t[0] (16 bits), a shift (6 bits) and 15 six-bit differences, biased by 32, along the chain
// 0 -> 4 -> 8 -> 12 (down the first column), then column 0 -> 1, 1 -> 2, 2 -> 3 in each of the four rows; the shift is the
// smallest for which every rounded difference fits.  B44A writes a block of 16 equal values as 3 bytes: t[0] and a
// third byte that no 14-byte block can have (0xfc: a shift of 63).  A channel flagged pLinear in the header was passed
// through exp(x / 8) before packing and needs 8 log(x) after unpacking (a 65 536-entry table the caller provides).
namespace {
inline int b44_shift_and_round(int x, int shift) {  // x / 2^shift, to nearest, ties to even
    x <<= 1;
    const int a = (1 << shift) - 1;
    shift += 1;
    const int b = (x >> shift) & 1;
    return (x + a + b) >> shift;
}

int b44_pack(const uint16_t s[16], uint8_t b[14], bool flat_fields, bool exact_max) {
    uint16_t t[16];
    for (int i = 0; i < 16; ++i) {
        if ((s[i] & 0x7c00) == 0x7c00) t[i] = 0x8000;
        else if (s[i] & 0x8000) t[i] = (uint16_t)~s[i];
        else t[i] = (uint16_t)(s[i] | 0x8000);
    }
    uint16_t t_max = 0;
    for (int i = 0; i < 16; ++i) t_max = std::max(t_max, t[i]);
    const int bias = 0x20;
    int shift = -1, d[16], r[15], r_min, r_max;
    do {
        shift += 1;
        for (int i = 0; i < 16; ++i) d[i] = b44_shift_and_round(t_max - t[i], shift);
        r[0] = d[0] - d[4] + bias;   r[1] = d[4] - d[8] + bias;    r[2] = d[8] - d[12] + bias;
        r[3] = d[0] - d[1] + bias;   r[4] = d[4] - d[5] + bias;    r[5] = d[8] - d[9] + bias;    r[6] = d[12] - d[13] + bias;
        r[7] = d[1] - d[2] + bias;   r[8] = d[5] - d[6] + bias;    r[9] = d[9] - d[10] + bias;   r[10] = d[13] - d[14] + bias;
        r[11] = d[2] - d[3] + bias;  r[12] = d[6] - d[7] + bias;   r[13] = d[10] - d[11] + bias; r[14] = d[14] - d[15] + bias;
        r_min = r_max = r[0];
        for (int i = 1; i < 15; ++i) {
            r_min = std::min(r_min, r[i]);
            r_max = std::max(r_max, r[i]);
        }
    } while (r_min < 0 || r_max > 0x3f);
    if (r_min == bias && r_max == bias && flat_fields) {
        b[0] = (uint8_t)(t[0] >> 8);
        b[1] = (uint8_t)t[0];
        b[2] = 0xfc;
        return 3;
    }
    if (exact_max) t[0] = (uint16_t)(t_max - (d[0] << shift));  // the block's largest value comes back as exactly as it can
    b[0] = (uint8_t)(t[0] >> 8);
    b[1] = (uint8_t)t[0];
    b[2] = (uint8_t)((shift << 2) | (r[0] >> 4));
    b[3] = (uint8_t)((r[0] << 4) | (r[1] >> 2));
    b[4] = (uint8_t)((r[1] << 6) | r[2]);
    b[5] = (uint8_t)((r[3] << 2) | (r[4] >> 4));
    b[6] = (uint8_t)((r[4] << 4) | (r[5] >> 2));
    b[7] = (uint8_t)((r[5] << 6) | r[6]);
    b[8] = (uint8_t)((r[7] << 2) | (r[8] >> 4));
    b[9] = (uint8_t)((r[8] << 4) | (r[9] >> 2));
    b[10] = (uint8_t)((r[9] << 6) | r[10]);
    b[11] = (uint8_t)((r[11] << 2) | (r[12] >> 4));
    b[12] = (uint8_t)((r[12] << 4) | (r[13] >> 2));
    b[13] = (uint8_t)((r[13] << 6) | r[14]);
    return 14;
}

inline void b44_from_ordered(uint16_t s[16]) {
    for (int i = 0; i < 16; ++i) s[i] = (s[i] & 0x8000) ? (uint16_t)(s[i] & 0x7fff) : (uint16_t)~s[i];
}

void b44_unpack14(const uint8_t b[14], uint16_t s[16]) {  // (16-bit wrap-around arithmetic, like the packer's)
    const unsigned shift = b[2] >> 2;  // < 13 (the caller took the 3-byte form otherwise)
    const unsigned bias = 0x20u << shift;
    auto step = [&](int from, unsigned six) { return (uint16_t)(s[from] + ((six & 0x3fu) << shift) - bias); };
    s[0] = (uint16_t)((b[0] << 8) | b[1]);
    s[4] = step(0, ((unsigned)b[2] << 4) | (b[3] >> 4));
    s[8] = step(4, ((unsigned)b[3] << 2) | (b[4] >> 6));
    s[12] = step(8, b[4]);
    s[1] = step(0, b[5] >> 2);
    s[5] = step(4, ((unsigned)b[5] << 4) | (b[6] >> 4));
    s[9] = step(8, ((unsigned)b[6] << 2) | (b[7] >> 6));
    s[13] = step(12, b[7]);
    s[2] = step(1, b[8] >> 2);
    s[6] = step(5, ((unsigned)b[8] << 4) | (b[9] >> 4));
    s[10] = step(9, ((unsigned)b[9] << 2) | (b[10] >> 6));
    s[14] = step(13, b[10]);
    s[3] = step(2, b[11] >> 2);
    s[7] = step(6, ((unsigned)b[11] << 4) | (b[12] >> 4));
    s[11] = step(10, ((unsigned)b[12] << 2) | (b[13] >> 6));
    s[15] = step(14, b[13]);
    b44_from_ordered(s);
}

void b44_unpack3(const uint8_t b[3], uint16_t s[16]) {
    s[0] = (uint16_t)((b[0] << 8) | b[1]);
    for (int i = 1; i < 16; ++i) s[i] = s[0];
    b44_from_ordered(s);
}
}  // namespace

bool b44_decode(const uint8_t* data, size_t size, const std::vector<int>& types, const std::vector<const uint16_t*>& to_linear, int bw,
                int bh, std::vector<uint8_t>& raw) {
    if (bw <= 0 || bh <= 0 || to_linear.size() != types.size()) return false;
    size_t line_bytes = 0;
    for (int t : types) line_bytes += (size_t)bw * (t == 1 ? 2 : 4);
    raw.assign(line_bytes * (size_t)bh, 0);
    const uint8_t *p = data, *const end = data + size;
    size_t chan_off = 0;
    for (size_t c = 0; c < types.size(); ++c) {
        if (types[c] != 1) {  // uncompressed: bh rows of bw four-byte samples
            const size_t row = (size_t)bw * 4;
            if ((size_t)(end - p) < row * (size_t)bh) return false;
            for (int y = 0; y < bh; ++y, p += row) std::memcpy(raw.data() + (size_t)y * line_bytes + chan_off, p, row);
            chan_off += row;
            continue;
        }
        for (int y = 0; y < bh; y += 4)
            for (int x = 0; x < bw; x += 4) {
                uint16_t s[16];
                if (end - p < 3) return false;
                if (p[2] >= (13 << 2)) {
                    b44_unpack3(p, s);
                    p += 3;
                } else {
                    if (end - p < 14) return false;
                    b44_unpack14(p, s);
                    p += 14;
                }
                if (to_linear[c])
                    for (int i = 0; i < 16; ++i) s[i] = to_linear[c][s[i]];
                for (int j = 0; j < 4 && y + j < bh; ++j) {
                    uint8_t* o = raw.data() + (size_t)(y + j) * line_bytes + chan_off + (size_t)x * 2;
                    for (int i = 0; i < 4 && x + i < bw; ++i) {
                        *o++ = (uint8_t)(s[4 * j + i] & 0xff);
                        *o++ = (uint8_t)(s[4 * j + i] >> 8);
                    }
                }
            }
        chan_off += (size_t)bw * 2;
    }
    return p == end;
}

bool b44_encode(const uint8_t* raw, const std::vector<int>& types, int bw, int bh, bool flat_fields, std::vector<uint8_t>& out) {
    if (bw <= 0 || bh <= 0) return false;
    size_t line_bytes = 0;
    for (int t : types) line_bytes += (size_t)bw * (t == 1 ? 2 : 4);
    out.clear();
    size_t chan_off = 0;
    for (int t : types) {
        if (t != 1) {
            const size_t row = (size_t)bw * 4;
            for (int y = 0; y < bh; ++y) out.insert(out.end(), raw + (size_t)y * line_bytes + chan_off, raw + (size_t)y * line_bytes + chan_off + row);
            chan_off += row;
            continue;
        }
        auto at = [&](int x, int y) {  // edge pixels repeat
            const uint8_t* q = raw + (size_t)std::min(y, bh - 1) * line_bytes + chan_off + (size_t)std::min(x, bw - 1) * 2;
            return (uint16_t)(q[0] | (q[1] << 8));
        };
        for (int y = 0; y < bh; y += 4)
            for (int x = 0; x < bw; x += 4) {
                uint16_t s[16];
                for (int j = 0; j < 4; ++j)
                    for (int i = 0; i < 4; ++i) s[4 * j + i] = at(x + i, y + j);
                uint8_t b[14];
                const int n = b44_pack(s, b, flat_fields, true);
                out.insert(out.end(), b, b + n);
            }
        chan_off += (size_t)bw * 2;
    }
    return true;
}

}  // namespace exr
