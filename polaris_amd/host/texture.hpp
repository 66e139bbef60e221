// texture.hpp -- image files -> the tracer's texture blob (asset/texure/texture.go:25-150).
//
// The reference hands every file to OpenImageIO (un-vendored cgo dependency
// github.com/achilleasa/openimageigo, absent here) and keeps what comes back as
//   8-bit, 1 channel -> Luminance8      8-bit, 3|4 channels -> Rgba8 (alpha 255 added)
//   otherwise, 1 channel -> Luminance32F   otherwise, 3|4 channels -> Rgba32F (alpha 1.0 added)
// rows top to bottom, and rejects any other channel count.  This build decodes the formats it
// can without a library -- PNG (zlib), PNM (P2/P3/P5/P6), BMP, TGA, Radiance HDR -- into exactly
// that representation (wider-than-8-bit integer samples are normalised to [0,1] floats, which
// is OpenImageIO's integer->float conversion); JPEG (jpeg.hpp: libjpeg's default pipeline restated, 8-bit grey or RGB); gif/tiff/exr/webp are reported as unsupported.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "tracer.hpp"

namespace polaris {
namespace texture {

struct Texture { // texture.go:15-23
	uint32_t format = 0; // POLARIS_TEX_*
	uint32_t width = 0, height = 0;
	std::vector<uint8_t> data;
};

Error Load(const std::string &path, Texture *out);
Error Decode(const std::string &nameForErrors, const std::vector<uint8_t> &file, Texture *out);

} // namespace texture
} // namespace polaris
