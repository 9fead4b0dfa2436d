// Encoded page -> RGB8: replaces ImageHelper::new_from_raw_img_flow
// (/root/reference/retto-core/src/image_helper.rs:34-44 = image::load_from_memory(bytes)?.to_rgb8()).
// Host code, like the reference's: the decoders are lossless-format readers written here (PNG over zlib's
// inflate, binary / ASCII PNM, uncompressed BMP) plus a Huffman-DCT JPEG reader
// (sequential and progressive).  Conversion to RGB8 follows the `image`
// crate 0.25.6 rules the reference relies on: alpha is dropped (not blended), grey is replicated,
// 16-bit samples map to 8 bits as (v + 128) / 257, sub-byte grey is scaled to the full range,
// palettes are expanded, gamma / colour-profile chunks are ignored.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace rt {

// Throws RtError(RT_ERR_IMAGE) with the reason (unknown format, truncated stream, CRC mismatch, ...).
void decode_image(const uint8_t* data, size_t len, std::vector<uint8_t>* rgb, int* h, int* w);

}  // namespace rt
