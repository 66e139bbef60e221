// texture.cpp -- see texture.hpp.  Each decoder produces interleaved samples (8-bit or float),
// channel count 1/3/4, rows top to bottom; finish() applies the reference's layout rules
// (asset/texure/texture.go:60-147).
#include "texture.hpp"
#include "jpeg.hpp"

#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstring>

#include "polaris_hip.h"

namespace polaris {
namespace texture {
namespace {

struct Raw {
	uint32_t w = 0, h = 0, channels = 0;
	bool isFloat = false;
	std::vector<uint8_t> u8;
	std::vector<float> f32;
};

Error bad(const std::string &m) { return Error{POLARIS_E_BAD_SCENE, m}; }

Error finish(const std::string &name, const Raw &r, Texture *out) { // texture.go:57-147
	if (r.channels != 1 && r.channels != 3 && r.channels != 4)
		return bad("texture: unsupported channel count " + std::to_string(r.channels) + " while loading " + name);
	const size_t px = (size_t)r.w * r.h;
	out->width = r.w;
	out->height = r.h;
	if (!r.isFloat) {
		out->format = r.channels == 1 ? POLARIS_TEX_L8 : POLARIS_TEX_RGBA8;
		if (r.channels == 3) {
			out->data.resize(px * 4);
			for (size_t i = 0; i < px; i++) { memcpy(&out->data[i * 4], &r.u8[i * 3], 3); out->data[i * 4 + 3] = 255; }
		} else {
			out->data = r.u8;
		}
	} else {
		out->format = r.channels == 1 ? POLARIS_TEX_L32F : POLARIS_TEX_RGBA32F;
		std::vector<float> t;
		const std::vector<float> *src = &r.f32;
		if (r.channels == 3) {
			t.resize(px * 4);
			for (size_t i = 0; i < px; i++) { memcpy(&t[i * 4], &r.f32[i * 3], 12); t[i * 4 + 3] = 1.0f; }
			src = &t;
		}
		out->data.resize(src->size() * 4);
		memcpy(out->data.data(), src->data(), out->data.size());
	}
	return Error::Nil();
}

// ---- PNM (P2 P3 P5 P6) -----------------------------------------------------------------------
struct Cursor {
	const std::vector<uint8_t> &d;
	size_t p = 0;
	bool skipSpace() {
		for (;;) {
			while (p < d.size() && isspace(d[p])) p++;
			if (p < d.size() && d[p] == '#') { while (p < d.size() && d[p] != '\n') p++; continue; }
			return p < d.size();
		}
	}
	bool number(uint32_t *out) {
		if (!skipSpace() || !isdigit(d[p])) return false;
		uint64_t v = 0;
		while (p < d.size() && isdigit(d[p])) { v = v * 10 + (d[p++] - '0'); if (v > 0xFFFFFFFFull) return false; }
		*out = (uint32_t)v;
		return true;
	}
};

Error decodePNM(const std::string &name, const std::vector<uint8_t> &f, Raw *r) {
	const int kind = f[1] - '0';
	Cursor c{f, 2};
	uint32_t maxval = 0;
	if (!c.number(&r->w) || !c.number(&r->h) || !c.number(&maxval) || maxval == 0 || maxval > 65535 || r->w == 0 || r->h == 0 ||
	    (uint64_t)r->w * r->h > (1ull << 28))
		return bad("texture: malformed PNM header in " + name);
	r->channels = (kind == 2 || kind == 5) ? 1 : 3;
	const size_t n = (size_t)r->w * r->h * r->channels;
	std::vector<uint32_t> s(n);
	if (kind == 2 || kind == 3) {
		for (size_t i = 0; i < n; i++)
			if (!c.number(&s[i])) return bad("texture: truncated PNM data in " + name);
	} else {
		c.p++; // the single whitespace byte after maxval
		const size_t bps = maxval > 255 ? 2 : 1;
		if (c.p + n * bps > f.size()) return bad("texture: truncated PNM data in " + name);
		for (size_t i = 0; i < n; i++) s[i] = bps == 1 ? f[c.p + i] : (uint32_t)(f[c.p + 2 * i] << 8 | f[c.p + 2 * i + 1]);
	}
	if (maxval == 255) {
		r->u8.resize(n);
		for (size_t i = 0; i < n; i++) r->u8[i] = (uint8_t)(s[i] > 255 ? 255 : s[i]);
	} else if (maxval < 255) { // rescaled to the full 8-bit range
		r->u8.resize(n);
		for (size_t i = 0; i < n; i++) r->u8[i] = (uint8_t)((std::min(s[i], maxval) * 255u + maxval / 2) / maxval);
	} else {
		r->isFloat = true;
		r->f32.resize(n);
		for (size_t i = 0; i < n; i++) r->f32[i] = (float)std::min(s[i], maxval) / (float)maxval;
	}
	return Error::Nil();
}

// ---- PNG (non-interlaced) ----------------------------------------------------------------------
uint32_t be32(const uint8_t *p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

Error decodePNG(const std::string &name, const std::vector<uint8_t> &f, Raw *r) {
	size_t p = 8;
	uint32_t depth = 0, ctype = 0, interlace = 0;
	std::vector<uint8_t> idat, plte;
	bool haveHdr = false;
	while (p + 12 <= f.size()) {
		const uint32_t len = be32(&f[p]);
		const char *type = (const char *)&f[p + 4];
		if (p + 12 + (size_t)len > f.size()) return bad("texture: truncated PNG chunk in " + name);
		const uint8_t *body = &f[p + 8];
		if (!memcmp(type, "IHDR", 4) && len >= 13) {
			r->w = be32(body); r->h = be32(body + 4);
			depth = body[8]; ctype = body[9]; interlace = body[12];
			haveHdr = true;
		} else if (!memcmp(type, "PLTE", 4)) {
			plte.assign(body, body + len);
		} else if (!memcmp(type, "IDAT", 4)) {
			idat.insert(idat.end(), body, body + len);
		} else if (!memcmp(type, "IEND", 4)) {
			break;
		}
		p += 12 + (size_t)len;
	}
	if (!haveHdr || r->w == 0 || r->h == 0 || (uint64_t)r->w * r->h > (1ull << 28)) return bad("texture: malformed PNG header in " + name);
	if (interlace) return bad("texture: interlaced PNG is not supported (" + name + ")");
	uint32_t samples;
	switch (ctype) {
	case 0: samples = 1; break;
	case 2: samples = 3; break;
	case 3: samples = 1; break;
	case 4: samples = 2; break;
	case 6: samples = 4; break;
	default: return bad("texture: unknown PNG colour type in " + name);
	}
	if ((depth != 8 && depth != 16 && !(depth < 8 && (ctype == 0 || ctype == 3))) || (ctype == 3 && depth > 8))
		return bad("texture: unsupported PNG bit depth in " + name);
	const size_t bpp = std::max<size_t>(1, samples * depth / 8);         // filter distance in bytes
	const size_t stride = ((size_t)r->w * samples * depth + 7) / 8;
	std::vector<uint8_t> raw((stride + 1) * r->h);
	uLongf rawLen = (uLongf)raw.size();
	if (uncompress(raw.data(), &rawLen, idat.data(), (uLong)idat.size()) != Z_OK || rawLen != raw.size())
		return bad("texture: could not inflate PNG data in " + name);
	std::vector<uint8_t> img(stride * r->h);
	for (uint32_t y = 0; y < r->h; y++) { // un-filter
		const uint8_t ft = raw[(stride + 1) * y];
		const uint8_t *in = &raw[(stride + 1) * y + 1];
		uint8_t *cur = &img[stride * y];
		const uint8_t *up = y ? &img[stride * (y - 1)] : nullptr;
		for (size_t i = 0; i < stride; i++) {
			const int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
			int pred = 0;
			switch (ft) {
			case 0: pred = 0; break;
			case 1: pred = a; break;
			case 2: pred = b; break;
			case 3: pred = (a + b) / 2; break;
			case 4: {
				const int pa = std::abs(b - c), pb = std::abs(a - c), pc = std::abs(a + b - 2 * c);
				pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
				break;
			}
			default: return bad("texture: bad PNG filter in " + name);
			}
			cur[i] = (uint8_t)(in[i] + pred);
		}
	}
	const size_t px = (size_t)r->w * r->h;
	if (ctype == 3) { // palette -> RGB
		r->channels = 3;
		r->u8.resize(px * 3);
		for (uint32_t y = 0; y < r->h; y++)
			for (uint32_t x = 0; x < r->w; x++) {
				const size_t bit = (size_t)x * depth;
				const uint32_t idx = (img[stride * y + bit / 8] >> (8 - depth - bit % 8)) & ((1u << depth) - 1);
				for (int k = 0; k < 3; k++) r->u8[((size_t)y * r->w + x) * 3 + k] = idx * 3 + k < plte.size() ? plte[idx * 3 + k] : 0;
			}
		return Error::Nil();
	}
	r->channels = samples;
	if (depth == 16) {
		r->isFloat = true;
		r->f32.resize(px * samples);
		for (uint32_t y = 0; y < r->h; y++)
			for (size_t i = 0; i < (size_t)r->w * samples; i++)
				r->f32[(size_t)y * r->w * samples + i] = (float)(img[stride * y + 2 * i] << 8 | img[stride * y + 2 * i + 1]) / 65535.0f;
	} else if (depth == 8) {
		r->u8 = img;
	} else { // 1/2/4-bit grey, scaled to 8 bits
		r->u8.resize(px);
		const uint32_t maxv = (1u << depth) - 1;
		for (uint32_t y = 0; y < r->h; y++)
			for (uint32_t x = 0; x < r->w; x++) {
				const size_t bit = (size_t)x * depth;
				const uint32_t v = (img[stride * y + bit / 8] >> (8 - depth - bit % 8)) & maxv;
				r->u8[(size_t)y * r->w + x] = (uint8_t)(v * 255u / maxv);
			}
	}
	return Error::Nil();
}

// ---- BMP (uncompressed 8-bit palette / 24 / 32) ---------------------------------------------------
uint32_t le32(const uint8_t *p) { return p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
uint32_t le16(const uint8_t *p) { return p[0] | (uint32_t)p[1] << 8; }

Error decodeBMP(const std::string &name, const std::vector<uint8_t> &f, Raw *r) {
	if (f.size() < 54) return bad("texture: truncated BMP header in " + name);
	const uint32_t off = le32(&f[10]), hdr = le32(&f[14]);
	const int32_t w = (int32_t)le32(&f[18]), hs = (int32_t)le32(&f[22]);
	const uint32_t bpp = le16(&f[28]), comp = le32(&f[30]);
	if (hdr < 40 || w <= 0 || hs == 0 || (comp != 0 && comp != 3) || (bpp != 8 && bpp != 24 && bpp != 32))
		return bad("texture: unsupported BMP variant in " + name);
	const uint32_t h = (uint32_t)std::abs(hs);
	const size_t stride = (((size_t)w * bpp + 31) / 32) * 4;
	if ((uint64_t)w * h > (1ull << 28) || off + stride * h > f.size()) return bad("texture: truncated BMP data in " + name);
	r->w = (uint32_t)w; r->h = h;
	r->channels = bpp == 32 ? 4 : 3;
	r->u8.resize((size_t)w * h * r->channels);
	const uint8_t *pal = &f[14 + hdr];
	for (uint32_t y = 0; y < h; y++) {
		const uint8_t *row = &f[off + stride * (hs > 0 ? h - 1 - y : y)]; // positive height = bottom-up
		uint8_t *dst = &r->u8[(size_t)y * w * r->channels];
		for (int32_t x = 0; x < w; x++) {
			if (bpp == 8) {
				const uint8_t *e = pal + 4 * row[x];
				if (e + 3 > f.data() + f.size()) return bad("texture: BMP palette out of range in " + name);
				dst[3 * x] = e[2]; dst[3 * x + 1] = e[1]; dst[3 * x + 2] = e[0];
			} else {
				const uint8_t *s = row + (size_t)x * (bpp / 8);
				dst[r->channels * x] = s[2]; dst[r->channels * x + 1] = s[1]; dst[r->channels * x + 2] = s[0];
				if (bpp == 32) dst[4 * x + 3] = s[3];
			}
		}
	}
	return Error::Nil();
}

// ---- TGA (types 2, 3, 10, 11) ------------------------------------------------------------------
Error decodeTGA(const std::string &name, const std::vector<uint8_t> &f, Raw *r) {
	if (f.size() < 18) return bad("texture: truncated TGA header in " + name);
	const uint32_t idLen = f[0], cmap = f[1], type = f[2], w = le16(&f[12]), h = le16(&f[14]), bpp = f[16], desc = f[17];
	const bool rle = type == 10 || type == 11, grey = type == 3 || type == 11;
	if (cmap != 0 || !(type == 2 || type == 3 || rle) || w == 0 || h == 0 || (grey ? bpp != 8 : (bpp != 24 && bpp != 32)))
		return bad("texture: unsupported TGA variant in " + name);
	const size_t bytes = bpp / 8, px = (size_t)w * h;
	std::vector<uint8_t> pix(px * bytes);
	size_t p = 18 + idLen;
	if (!rle) {
		if (p + pix.size() > f.size()) return bad("texture: truncated TGA data in " + name);
		memcpy(pix.data(), &f[p], pix.size());
	} else {
		size_t o = 0;
		while (o < px) {
			if (p >= f.size()) return bad("texture: truncated TGA data in " + name);
			const uint32_t hd = f[p++], cnt = (hd & 127) + 1;
			if (o + cnt > px) return bad("texture: corrupt TGA run in " + name);
			if (hd & 128) {
				if (p + bytes > f.size()) return bad("texture: truncated TGA data in " + name);
				for (uint32_t i = 0; i < cnt; i++) memcpy(&pix[(o + i) * bytes], &f[p], bytes);
				p += bytes;
			} else {
				if (p + cnt * bytes > f.size()) return bad("texture: truncated TGA data in " + name);
				memcpy(&pix[o * bytes], &f[p], cnt * bytes);
				p += cnt * bytes;
			}
			o += cnt;
		}
	}
	r->w = w; r->h = h; r->channels = (uint32_t)(grey ? 1 : bytes);
	r->u8.resize(px * r->channels);
	const bool topDown = desc & 0x20, rightLeft = desc & 0x10;
	for (uint32_t y = 0; y < h; y++)
		for (uint32_t x = 0; x < w; x++) {
			const uint8_t *s = &pix[((size_t)(topDown ? y : h - 1 - y) * w + (rightLeft ? w - 1 - x : x)) * bytes];
			uint8_t *d = &r->u8[((size_t)y * w + x) * r->channels];
			if (grey) d[0] = s[0];
			else { d[0] = s[2]; d[1] = s[1]; d[2] = s[0]; if (bytes == 4) d[3] = s[3]; }
		}
	return Error::Nil();
}

// ---- Radiance HDR (RGBE) -------------------------------------------------------------------------
Error decodeHDR(const std::string &name, const std::vector<uint8_t> &f, Raw *r) {
	size_t p = 0;
	auto line = [&](std::string *out) {
		out->clear();
		while (p < f.size() && f[p] != '\n') out->push_back((char)f[p++]);
		if (p >= f.size()) return false;
		p++;
		return true;
	};
	std::string l;
	bool fmtOk = false;
	while (line(&l) && !l.empty())
		if (l.find("FORMAT=32-bit_rle_rgbe") != std::string::npos) fmtOk = true;
	int h = 0, w = 0;
	if (!fmtOk || !line(&l) || sscanf(l.c_str(), "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0 || (uint64_t)w * h > (1ull << 28))
		return bad("texture: unsupported Radiance HDR header in " + name);
	std::vector<uint8_t> rgbe((size_t)w * h * 4);
	for (int y = 0; y < h; y++) {
		uint8_t *row = &rgbe[(size_t)y * w * 4];
		if (w >= 8 && w < 32768 && p + 4 <= f.size() && f[p] == 2 && f[p + 1] == 2 && (f[p + 2] << 8 | f[p + 3]) == w) {
			p += 4;
			for (int ch = 0; ch < 4; ch++)
				for (int x = 0; x < w;) {
					if (p >= f.size()) return bad("texture: truncated HDR data in " + name);
					int cnt = f[p++];
					if (cnt > 128) {
						cnt -= 128;
						if (p >= f.size() || x + cnt > w) return bad("texture: corrupt HDR run in " + name);
						const uint8_t v = f[p++];
						while (cnt--) row[4 * x++ + ch] = v;
					} else {
						if (cnt == 0 || p + cnt > f.size() || x + cnt > w) return bad("texture: corrupt HDR run in " + name);
						while (cnt--) row[4 * x++ + ch] = f[p++];
					}
				}
		} else {
			if (p + (size_t)w * 4 > f.size()) return bad("texture: truncated HDR data in " + name);
			memcpy(row, &f[p], (size_t)w * 4);
			p += (size_t)w * 4;
		}
	}
	r->w = (uint32_t)w; r->h = (uint32_t)h; r->channels = 3; r->isFloat = true;
	r->f32.resize((size_t)w * h * 3);
	for (size_t i = 0; i < (size_t)w * h; i++) {
		const uint8_t *e = &rgbe[i * 4];
		const float s = e[3] ? std::ldexp(1.0f, (int)e[3] - (128 + 8)) : 0.0f;
		for (int k = 0; k < 3; k++) r->f32[i * 3 + k] = e[3] ? ((float)e[k] + 0.5f) * s : 0.0f;
	}
	return Error::Nil();
}

} // namespace

Error Decode(const std::string &name, const std::vector<uint8_t> &f, Texture *out) {
	if (!out) return Error{POLARIS_E_BAD_ARGUMENT, "texture output is null"};
	*out = Texture{};
	Raw r;
	Error e;
	static const uint8_t pngSig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
	if (f.size() >= 8 && !memcmp(f.data(), pngSig, 8)) e = decodePNG(name, f, &r);
	else if (f.size() >= 3 && f[0] == 'P' && (f[1] == '2' || f[1] == '3' || f[1] == '5' || f[1] == '6')) e = decodePNM(name, f, &r);
	else if (f.size() >= 2 && f[0] == 'B' && f[1] == 'M') e = decodeBMP(name, f, &r);
	else if (f.size() >= 10 && (!memcmp(f.data(), "#?RADIANCE", 10) || !memcmp(f.data(), "#?RGBE", 6))) e = decodeHDR(name, f, &r);
	else if (f.size() >= 3 && f[0] == 0xFF && f[1] == 0xD8 && f[2] == 0xFF) { // JPEG: 8-bit grey or RGB, as libjpeg (OpenImageIO's reader) delivers it
		r.isFloat = false;
		e = DecodeJPEG(name, f, &r.w, &r.h, &r.channels, &r.u8);
	} else if (name.size() >= 4 && !strcasecmp(name.c_str() + name.size() - 4, ".tga")) e = decodeTGA(name, f, &r);
	else return bad("texture: no decoder in this build for " + name + " (supported: png, jpeg, pnm, bmp, tga, hdr)");
	if (e) return e;
	return finish(name, r, out);
}

Error Load(const std::string &path, Texture *out) {
	FILE *fp = fopen(path.c_str(), "rb");
	if (!fp) return bad("texture: could not open " + path);
	std::vector<uint8_t> data;
	uint8_t buf[65536];
	size_t n;
	while ((n = fread(buf, 1, sizeof buf, fp)) > 0) data.insert(data.end(), buf, buf + n);
	fclose(fp);
	return Decode(path, data, out);
}

} // namespace texture
} // namespace polaris
