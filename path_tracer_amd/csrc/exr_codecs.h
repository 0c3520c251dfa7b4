// exr_codecs.h — PIZ, PXR24 and B44(A) block codecs of exr_io.cpp (see exr_codecs.cpp).
// `types` = sample type of each channel in file order (0 UINT, 1 HALF, 2 FLOAT); a block is bh lines of bw pixels;
// `raw` = for each line, for each channel, bw little-endian samples.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <vector>

namespace exr {
bool piz_decode(const uint8_t* data, size_t size, const std::vector<int>& types, int bw, int bh, std::vector<uint8_t>& raw);
bool piz_encode(const uint8_t* raw, const std::vector<int>& types, int bw, int bh, std::vector<uint8_t>& out);
bool pxr24_decode(const uint8_t* data, size_t size, const std::vector<int>& types, int bw, int bh, std::vector<uint8_t>& raw);
bool pxr24_encode(const uint8_t* raw, const std::vector<int>& types, int bw, int bh, std::vector<uint8_t>& out);
// B44 / B44A.  to_linear[c]: nullptr, or for a HALF channel flagged pLinear the 65 536-entry table half -> half(8 log x)
// applied after unpacking.  flat_fields: B44A's 3-byte blocks for 16 equal values.  (The encoder writes no pLinear channels.)
bool b44_decode(const uint8_t* data, size_t size, const std::vector<int>& types, const std::vector<const uint16_t*>& to_linear, int bw,
                int bh, std::vector<uint8_t>& raw);
bool b44_encode(const uint8_t* raw, const std::vector<int>& types, int bw, int bh, bool flat_fields, std::vector<uint8_t>& out);
}  // namespace exr
