// jpeg.hpp -- a JPEG decoder for the texture front-end (asset/texure/texture.go:25-150 hands every file to OpenImageIO, whose JPEG
// reader is libjpeg: 8-bit samples, 1 channel for greyscale files, 3 channels RGB otherwise).
//
// Baseline / extended sequential (SOF0, SOF1) and progressive (SOF2) Huffman JPEG, 8 bits per sample, 1 or 3 components, any sampling
// factors, restart intervals, interleaved and non-interleaved scans.  The arithmetic is libjpeg's DEFAULT pipeline restated, so the bytes
// equal what libjpeg / libjpeg-turbo (and therefore OpenImageIO, Pillow, ...) produce for the same file:
//   * the "islow" inverse DCT (Loeffler-Ligtenberg-Moschytz, 13-bit constants, two passes, the library's rounding and range limit),
//   * "fancy" chroma upsampling (the triangle filters of jdsample.c for 2h1v, 2h2v and -- as libjpeg-turbo -- 1h2v components; sample
//     replication for every other ratio and for components narrower than three samples),
//   * the 16-bit fixed-point YCbCr -> RGB tables of jdcolor.c.
// tests/test_scene_frontend.py compares it byte for byte with Pillow (libjpeg-turbo) over sizes, sampling modes, qualities, restart
// intervals and progressive files.  Not supported (reported as errors): arithmetic coding, lossless / hierarchical processes, 12-bit
// samples, CMYK / YCCK files.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "tracer.hpp"

namespace polaris {
namespace texture {

// pixels: rows top to bottom, `channels` (1 or 3) bytes per pixel
Error DecodeJPEG(const std::string &nameForErrors, const std::vector<uint8_t> &file, uint32_t *width, uint32_t *height, uint32_t *channels,
                 std::vector<uint8_t> *pixels);

} // namespace texture
} // namespace polaris
