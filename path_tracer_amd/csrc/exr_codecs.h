// exr_codecs.h — PIZ and PXR24 block codecs of exr_io.cpp (see exr_codecs.cpp).
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
}  // namespace exr
