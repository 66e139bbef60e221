// jpeg.cpp -- see jpeg.hpp.  libjpeg's default decoding pipeline restated (ITU T.81 for the bit stream; the arithmetic of jidctint.c,
// jdsample.c and jdcolor.c for the samples), one pass over the file into coefficient arrays, then IDCT, upsampling and colour
// conversion over whole planes.
#include "jpeg.hpp"

#include <algorithm>
#include <cstring>
#include <new>

#include "polaris_hip.h"

namespace polaris {
namespace texture {
namespace {

struct Fail { std::string msg; };
[[noreturn]] void fail(const std::string &m) { throw Fail{m}; }

// zigzag position -> natural (row-major) position, T.81 figure A.6
const uint8_t kNatural[64 + 16] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
                                   63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63}; // (16 extra entries: a corrupt run cannot index past the block)

struct Huffman {
	bool present = false;
	uint8_t vals[256] = {};
	int32_t maxcode[18] = {}; // largest code of length l (-1: none)
	int32_t valoff[17] = {};  // vals index of the first code of length l, minus that code
	uint8_t lookBits[256] = {}, lookSym[256] = {}; // 8-bit lookahead: code length (0: longer than 8 bits) and symbol
	void build(const uint8_t counts[16], const uint8_t *symbols, int n) {
		present = true;
		memcpy(vals, symbols, (size_t)n);
		int32_t code = 0;
		int k = 0;
		memset(lookBits, 0, sizeof lookBits);
		for (int l = 1; l <= 16; l++) {
			valoff[l] = k - code;
			if (code + counts[l - 1] > (1 << l)) fail("bad Huffman table (more codes of one length than the length admits)");
			for (int i = 0; i < counts[l - 1]; i++, k++, code++) {
				if (l <= 8) {
					const int first = code << (8 - l), cnt = 1 << (8 - l);
					for (int j = 0; j < cnt; j++) { lookBits[first + j] = (uint8_t)l; lookSym[first + j] = symbols[k]; }
				}
			}
			maxcode[l] = counts[l - 1] ? code - 1 : -1;
			code <<= 1;
		}
		maxcode[17] = 0x7FFFFFFF;
	}
};

struct BitReader {
	const uint8_t *p = nullptr, *end = nullptr;
	uint32_t acc = 0;
	int n = 0;       // valid bits in acc (the low n)
	int marker = 0;  // a marker found inside the entropy-coded data (0: none yet); from then on the reader supplies zero bits
	int pad = 0;     // how many of the n buffered bits (the lowest) are such zero bits
	const uint8_t *markerAt = nullptr; // where that marker's 0xFF sits (null: the end of the file stood in for one)
	bool starved = false; // a decoder has consumed zero bits: the entropy-coded data ended before the scan did (libjpeg: insufficient_data)
	void fill() {
		while (n <= 24) {
			uint32_t b = 0;
			if (!marker && p < end) {
				b = *p++;
				if (b == 0xFF) {
					while (p < end && *p == 0xFF) p++; // fill bytes
					if (p >= end) { marker = 0xD9; b = 0; }          // the file ends inside a marker: as if EOI
					else if (*p != 0) { marker = *p; markerAt = p - 1; p++; b = 0; }
					else p++;                                       // a stuffed zero byte: the data byte is 0xFF
				}
			} else if (!marker) marker = 0xD9; // ran off the end of the file: as if EOI
			acc = (acc << 8) | b;
			n += 8;
			if (marker) pad += 8;
		}
	}
	void used() { if (n < pad) { pad = n; starved = true; } }
	uint32_t peek(int k) { if (n < k) fill(); return (acc >> (n - k)) & ((1u << k) - 1u); }
	uint32_t get(int k) { if (k == 0) return 0; const uint32_t v = peek(k); n -= k; used(); return v; }
	uint32_t bit() { return get(1); }
	int decode(const Huffman &h) {
		const uint32_t look = peek(8);
		if (h.lookBits[look]) { n -= h.lookBits[look]; used(); return h.lookSym[look]; }
		int32_t code = (int32_t)get(9);
		int l = 9;
		while (code > h.maxcode[l]) { code = (code << 1) | (int32_t)bit(); l++; }
		if (l > 16) fail("corrupt Huffman code");
		return h.vals[(code + h.valoff[l]) & 255];
	}
	// at a restart boundary: drop the rest of the current byte, take the RSTn marker
	void restart(int expect) {
		n = 0; acc = 0; pad = 0; starved = false;
		if (!marker) {
			while (p < end && *p != 0xFF) p++; // (a conforming stream has the marker right here)
			while (p < end && *p == 0xFF) p++;
			marker = p < end ? *p++ : 0xD9;
		}
		if (marker != 0xD0 + (expect & 7)) fail("missing restart marker");
		marker = 0;
		markerAt = nullptr;
	}
};

inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v + (int)(0xFFFFFFFFu << s) + 1 : v; } // HUFF_EXTEND

struct Component {
	int id = 0, h = 1, v = 1, tq = 0;
	int wb = 0, hb = 0; // blocks that carry image samples
	int pw = 0, ph = 0; // blocks allocated (whole MCUs of an interleaved scan)
	int dw = 0, dh = 0; // samples: the component's "downsampled" size
	int dcTable = 0, acTable = 0, pred = 0;
	std::vector<int16_t> coef;
	std::vector<uint8_t> plane; // wb * 8 samples per row, hb * 8 rows, after the IDCT
};

struct Decoder {
	const std::vector<uint8_t> &f;
	size_t pos = 0;
	uint32_t W = 0, H = 0;
	bool progressive = false, haveFrame = false;
	std::vector<Component> comps;
	int hmax = 1, vmax = 1;
	uint16_t quant[4][64] = {};
	bool haveQuant[4] = {};
	Huffman dc[4], ac[4];
	int restartInterval = 0;
	bool sawAdobe = false, sawJfif = false;
	int adobeTransform = 0;

	explicit Decoder(const std::vector<uint8_t> &file) : f(file) {}
	uint8_t u8() { if (pos >= f.size()) fail("unexpected end of file"); return f[pos++]; }
	uint32_t u16() { const uint32_t a = u8(); return a << 8 | u8(); }

	void parseDQT(size_t end) {
		while (pos < end) {
			const int b = u8(), pq = b >> 4, tq = b & 15;
			if (tq > 3 || pq > 1) fail("bad quantisation table");
			for (int i = 0; i < 64; i++) quant[tq][kNatural[i]] = (uint16_t)(pq ? u16() : u8());
			haveQuant[tq] = true;
		}
	}
	void parseDHT(size_t end) {
		while (pos < end) {
			const int b = u8(), tc = b >> 4, th = b & 15;
			if (tc > 1 || th > 3) fail("bad Huffman table id");
			uint8_t counts[16], symbols[256];
			int n = 0;
			for (int i = 0; i < 16; i++) { counts[i] = u8(); n += counts[i]; }
			if (n > 256) fail("bad Huffman table");
			for (int i = 0; i < n; i++) symbols[i] = u8();
			(tc ? ac[th] : dc[th]).build(counts, symbols, n);
		}
	}
	void parseSOF(int marker) {
		if (haveFrame) fail("more than one frame");
		progressive = marker == 0xC2;
		if (u8() != 8) fail("only 8-bit samples are supported");
		H = u16(); W = u16();
		const int nc = u8();
		if (W == 0 || H == 0) fail("empty image");
		if ((uint64_t)W * H > (1ull << 26)) fail("image too large (more than 2^26 pixels)");
		// every 8 x 8 block costs its scan at least four bits (a zero DC difference and an end-of-block): a file far shorter than that
		// cannot hold the frame it announces (a corrupt header must not allocate and transform gigabytes of zeros)
		if ((uint64_t)f.size() < (uint64_t)W * H / 512) fail("file too short for the image size it announces");
		if (nc == 4) fail("CMYK / YCCK files are not supported");
		if (nc != 1 && nc != 3) fail("unsupported number of components");
		comps.assign((size_t)nc, Component{});
		for (auto &c : comps) {
			c.id = u8();
			const int hv = u8();
			c.h = hv >> 4; c.v = hv & 15; c.tq = u8();
			if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) fail("bad component parameters");
			hmax = std::max(hmax, c.h); vmax = std::max(vmax, c.v);
		}
		if (nc == 1) { comps[0].h = comps[0].v = 1; hmax = vmax = 1; } // a single component is never subsampled (its MCU is one block)
		const int mcux = (int)((W + 8u * hmax - 1) / (8u * hmax)), mcuy = (int)((H + 8u * vmax - 1) / (8u * vmax));
		for (auto &c : comps) {
			c.dw = (int)(((uint64_t)W * c.h + hmax - 1) / hmax);
			c.dh = (int)(((uint64_t)H * c.v + vmax - 1) / vmax);
			c.wb = (c.dw + 7) / 8; c.hb = (c.dh + 7) / 8;
			c.pw = mcux * c.h; c.ph = mcuy * c.v;
			c.coef.assign((size_t)c.pw * c.ph * 64, 0);
		}
		haveFrame = true;
	}

	// ---- entropy-coded segments -------------------------------------------------------------------
	struct Scan { std::vector<int> ci; int ss = 0, se = 63, ah = 0, al = 0; };
	int eobrun = 0;

	void blockSequential(BitReader &br, Component &c, int16_t *b) { // T.81 F.2.2
		const Huffman &hd = dc[c.dcTable], &ha = ac[c.acTable];
		int s = br.decode(hd);
		if (s) { if (s > 15) fail("corrupt DC coefficient"); s = extend((int)br.get(s), s); }
		c.pred = (int)((uint32_t)c.pred + (uint32_t)s); // (wraps like the library's on a corrupt stream instead of overflowing)
		b[0] = (int16_t)c.pred;
		for (int k = 1; k < 64; k++) {
			const int rs = br.decode(ha), r = rs >> 4;
			s = rs & 15;
			if (s) {
				k += r;
				b[kNatural[k]] = (int16_t)extend((int)br.get(s), s);
			} else {
				if (r != 15) break;
				k += 15;
			}
		}
	}
	void blockDcFirst(BitReader &br, Component &c, int16_t *b, int al) { // T.81 G.1.2.1
		int s = br.decode(dc[c.dcTable]);
		if (s) { if (s > 15) fail("corrupt DC coefficient"); s = extend((int)br.get(s), s); }
		c.pred = (int)((uint32_t)c.pred + (uint32_t)s);
		b[0] = (int16_t)((uint32_t)c.pred << al);
	}
	void blockDcRefine(BitReader &br, int16_t *b, int al) { if (br.bit()) b[0] |= (int16_t)(1 << al); }
	void blockAcFirst(BitReader &br, Component &c, int16_t *b, const Scan &sc) { // T.81 G.1.2.2
		if (eobrun > 0) { eobrun--; return; }
		const Huffman &ha = ac[c.acTable];
		for (int k = sc.ss; k <= sc.se; k++) {
			const int rs = br.decode(ha), r = rs >> 4, s = rs & 15;
			if (s) {
				k += r;
				b[kNatural[k]] = (int16_t)(extend((int)br.get(s), s) * (1 << sc.al));
			} else {
				if (r == 15) { k += 15; continue; }
				eobrun = 1 << r;
				if (r) eobrun += (int)br.get(r);
				eobrun--;
				break;
			}
		}
	}
	void blockAcRefine(BitReader &br, Component &c, int16_t *b, const Scan &sc) { // T.81 G.1.2.3
		const Huffman &ha = ac[c.acTable];
		const int p1 = 1 << sc.al, m1 = -(1 << sc.al);
		int k = sc.ss;
		auto correct = [&](int16_t &t) { // a correction bit for a coefficient that is already non-zero
			if (br.bit() && (t & p1) == 0) t = (int16_t)(t + (t >= 0 ? p1 : m1));
		};
		if (eobrun == 0) {
			for (; k <= sc.se; k++) {
				const int rs = br.decode(ha);
				int r = rs >> 4, s = rs & 15;
				if (s) {
					s = br.bit() ? p1 : m1; // (the size of a newly non-zero coefficient is always 1)
				} else if (r != 15) {
					eobrun = 1 << r;
					if (r) eobrun += (int)br.get(r);
					break; // the rest of this block is handled as part of the end-of-band run
				}
				do { // skip the already non-zero coefficients (each takes a correction bit) and r zero ones
					int16_t &t = b[kNatural[k]];
					if (t != 0) correct(t);
					else if (--r < 0) break;
					k++;
				} while (k <= sc.se);
				if (s && k <= sc.se) b[kNatural[k]] = (int16_t)s;
			}
		}
		if (eobrun > 0) {
			for (; k <= sc.se; k++) {
				int16_t &t = b[kNatural[k]];
				if (t != 0) correct(t);
			}
			eobrun--;
		}
	}

	void decodeScan(const Scan &sc) {
		BitReader br;
		br.p = f.data() + pos; br.end = f.data() + f.size();
		for (int ci : sc.ci) comps[(size_t)ci].pred = 0;
		eobrun = 0;
		const bool interleaved = sc.ci.size() > 1;
		int mcux, mcuy;
		if (interleaved) { mcux = (int)((W + 8u * hmax - 1) / (8u * hmax)); mcuy = (int)((H + 8u * vmax - 1) / (8u * vmax)); }
		else { mcux = comps[(size_t)sc.ci[0]].wb; mcuy = comps[(size_t)sc.ci[0]].hb; }
		int untilRestart = restartInterval, nextRestart = 0;
		auto one = [&](Component &c, int by, int bx) {
			int16_t *b = &c.coef[((size_t)by * c.pw + bx) * 64];
			if (!progressive) blockSequential(br, c, b);
			else if (sc.ss == 0) { if (sc.ah == 0) blockDcFirst(br, c, b, sc.al); else blockDcRefine(br, b, sc.al); }
			else { if (sc.ah == 0) blockAcFirst(br, c, b, sc); else blockAcRefine(br, c, b, sc); }
		};
		for (int my = 0; my < mcuy; my++)
			for (int mx = 0; mx < mcux; mx++) {
				// the entropy-coded data ended early (a marker, or the end of the file, inside it): like libjpeg (jdhuff.c, insufficient_data) the MCU
				// in which that happened was completed with zero bits, the ones after it are left as they are -- until a restart marker resynchronises
				if (br.starved && !(restartInterval && untilRestart == 0)) { untilRestart -= restartInterval ? 1 : 0; continue; }
				if (restartInterval && untilRestart == 0) {
					br.restart(nextRestart);
					nextRestart = (nextRestart + 1) & 7;
					untilRestart = restartInterval;
					for (int ci : sc.ci) comps[(size_t)ci].pred = 0;
					eobrun = 0;
				}
				untilRestart--;
				if (interleaved) {
					for (int ci : sc.ci) {
						Component &c = comps[(size_t)ci];
						for (int y = 0; y < c.v; y++)
							for (int x = 0; x < c.h; x++) one(c, my * c.v + y, mx * c.h + x);
					}
				} else one(comps[(size_t)sc.ci[0]], my, mx);
			}
		// continue parsing behind the entropy-coded data: at the marker the reader ran into, or at the next one in the file
		if (br.marker) pos = br.markerAt ? (size_t)(br.markerAt - f.data()) : f.size(); // back onto its FF xx -- or the file is over
		else {
			pos = (size_t)(br.p - f.data());
			while (pos + 1 < f.size() && !(f[pos] == 0xFF && f[pos + 1] != 0x00 && f[pos + 1] != 0xFF && !(f[pos + 1] >= 0xD0 && f[pos + 1] <= 0xD7))) pos++;
		}
	}

	void parseSOS(size_t headerEnd) {
		if (!haveFrame) fail("scan before frame header");
		Scan sc;
		const int ns = u8();
		if (ns < 1 || ns > (int)comps.size()) fail("bad number of components in scan");
		for (int i = 0; i < ns; i++) {
			const int id = u8(), t = u8();
			int ci = -1;
			for (size_t k = 0; k < comps.size(); k++) if (comps[k].id == id) ci = (int)k;
			if (ci < 0) fail("scan names an unknown component");
			comps[(size_t)ci].dcTable = t >> 4; comps[(size_t)ci].acTable = t & 15;
			if ((t >> 4) > 3 || (t & 15) > 3) fail("bad table selector");
			sc.ci.push_back(ci);
		}
		sc.ss = u8(); sc.se = u8();
		const int a = u8();
		sc.ah = a >> 4; sc.al = a & 15;
		if (!progressive) { sc.ss = 0; sc.se = 63; sc.ah = sc.al = 0; }
		else {
			if (sc.ss > sc.se || sc.se > 63 || sc.al > 13 || (sc.ss == 0 && sc.se != 0) || (sc.ss != 0 && ns != 1)) fail("bad progressive scan parameters");
		}
		for (int ci : sc.ci) {
			const Component &c = comps[(size_t)ci];
			if ((!progressive || sc.ss == 0) && !(progressive && sc.ah) && !dc[c.dcTable].present) fail("missing DC Huffman table");
			if ((!progressive || sc.ss != 0) && !ac[c.acTable].present) fail("missing AC Huffman table");
		}
		pos = headerEnd; // the entropy-coded data starts behind the scan header
		decodeScan(sc);
	}

	void parse() {
		if (f.size() < 4 || f[0] != 0xFF || f[1] != 0xD8) fail("not a JPEG file");
		pos = 2;
		bool sawScan = false;
		for (;;) {
			if (pos >= f.size()) { if (sawScan) break; fail("no image data"); }
			if (f[pos] != 0xFF) { pos++; continue; } // (garbage between segments: skip, as libjpeg does with a warning)
			while (pos < f.size() && f[pos] == 0xFF) pos++;
			if (pos >= f.size()) break;
			const int m = f[pos++];
			if (m == 0xD9) break;                                  // EOI
			if (m == 0x00 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue; // stand-alone
			const size_t start = pos;
			const uint32_t len = u16();
			if (len < 2 || start + len > f.size()) fail("bad segment length");
			const size_t end = start + len;
			switch (m) {
			case 0xDB: parseDQT(end); break;
			case 0xC4: parseDHT(end); break;
			case 0xC0: case 0xC1: case 0xC2: parseSOF(m); break;
			case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xCB: case 0xCD: case 0xCE: case 0xCF: fail("lossless / hierarchical JPEG processes are not supported");
			case 0xC9: case 0xCA: case 0xCC: fail("arithmetic-coded JPEG is not supported");
			case 0xDD: restartInterval = (int)u16(); break;
			case 0xE0: if (len >= 7 && !memcmp(&f[pos], "JFIF", 5)) sawJfif = true; break;
			case 0xEE: if (len >= 14 && !memcmp(&f[pos], "Adobe", 5)) { sawAdobe = true; adobeTransform = f[pos + 11]; } break;
			case 0xDA: parseSOS(end); sawScan = true; continue; // (decodes the scan behind the header and leaves pos at the next marker)
			default: break;
			}
			pos = end;
		}
		if (!haveFrame || !sawScan) fail("no image data");
	}

	// ---- samples ---------------------------------------------------------------------------------
	static inline uint8_t rangeLimit(int32_t x) { // the post-IDCT table of jdmaster.c (prepare_range_limit_table), sample + 128
		const int32_t j = x & 1023;
		return (uint8_t)(j < 128 ? j + 128 : (j < 512 ? 255 : (j < 896 ? 0 : j - 896)));
	}
	static void idct(const int16_t *in, const uint16_t *q, uint8_t *out, int stride) { // jidctint.c, jpeg_idct_islow
		constexpr int32_t F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299, F1847 = 15137, F1961 = 16069,
		                  F2053 = 16819, F2562 = 20995, F3072 = 25172;
		constexpr int CONST_BITS = 13, PASS1_BITS = 2;
		auto descale = [](int64_t x, int n) -> int32_t { return (int32_t)((x + ((int64_t)1 << (n - 1))) >> n); };
		int32_t ws[64];
		for (int c = 0; c < 8; c++) {
			auto v = [&](int r) -> int64_t { return (int64_t)in[r * 8 + c] * q[r * 8 + c]; };
			int64_t z2 = v(2), z3 = v(6);
			int64_t z1 = (z2 + z3) * F0541;
			int64_t tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
			z2 = v(0); z3 = v(4);
			int64_t tmp0 = (z2 + z3) * (1 << CONST_BITS), tmp1 = (z2 - z3) * (1 << CONST_BITS);
			const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
			tmp0 = v(7); tmp1 = v(5); tmp2 = v(3); tmp3 = v(1);
			z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
			int64_t z4 = tmp1 + tmp3;
			const int64_t z5 = (z3 + z4) * F1175;
			tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
			z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
			z3 += z5; z4 += z5;
			tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
			ws[0 * 8 + c] = descale(tmp10 + tmp3, CONST_BITS - PASS1_BITS); ws[7 * 8 + c] = descale(tmp10 - tmp3, CONST_BITS - PASS1_BITS);
			ws[1 * 8 + c] = descale(tmp11 + tmp2, CONST_BITS - PASS1_BITS); ws[6 * 8 + c] = descale(tmp11 - tmp2, CONST_BITS - PASS1_BITS);
			ws[2 * 8 + c] = descale(tmp12 + tmp1, CONST_BITS - PASS1_BITS); ws[5 * 8 + c] = descale(tmp12 - tmp1, CONST_BITS - PASS1_BITS);
			ws[3 * 8 + c] = descale(tmp13 + tmp0, CONST_BITS - PASS1_BITS); ws[4 * 8 + c] = descale(tmp13 - tmp0, CONST_BITS - PASS1_BITS);
		}
		for (int r = 0; r < 8; r++) {
			const int32_t *w = ws + r * 8;
			int64_t z2 = w[2], z3 = w[6];
			int64_t z1 = (z2 + z3) * F0541;
			int64_t tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
			int64_t tmp0 = ((int64_t)w[0] + w[4]) * (1 << CONST_BITS), tmp1 = ((int64_t)w[0] - w[4]) * (1 << CONST_BITS);
			const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
			tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
			z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
			int64_t z4 = tmp1 + tmp3;
			const int64_t z5 = (z3 + z4) * F1175;
			tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
			z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
			z3 += z5; z4 += z5;
			tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
			constexpr int S = CONST_BITS + PASS1_BITS + 3;
			uint8_t *o = out + (size_t)r * stride;
			o[0] = rangeLimit(descale(tmp10 + tmp3, S)); o[7] = rangeLimit(descale(tmp10 - tmp3, S));
			o[1] = rangeLimit(descale(tmp11 + tmp2, S)); o[6] = rangeLimit(descale(tmp11 - tmp2, S));
			o[2] = rangeLimit(descale(tmp12 + tmp1, S)); o[5] = rangeLimit(descale(tmp12 - tmp1, S));
			o[3] = rangeLimit(descale(tmp13 + tmp0, S)); o[4] = rangeLimit(descale(tmp13 - tmp0, S));
		}
	}
	void inverseTransform() {
		for (auto &c : comps) {
			if (!haveQuant[c.tq]) fail("missing quantisation table");
			const int stride = c.wb * 8;
			c.plane.assign((size_t)stride * c.hb * 8, 0);
			for (int by = 0; by < c.hb; by++)
				for (int bx = 0; bx < c.wb; bx++) idct(&c.coef[((size_t)by * c.pw + bx) * 64], quant[c.tq], &c.plane[((size_t)by * 8) * stride + (size_t)bx * 8], stride);
			c.coef.clear();
			c.coef.shrink_to_fit();
		}
	}
	// One component at full resolution (jdsample.c): W x H samples, rows and columns beyond its own size replicated from the edge.
	std::vector<uint8_t> upsample(const Component &c) const {
		const int stride = c.wb * 8;
		auto row = [&](int y) { return &c.plane[(size_t)std::min(std::max(y, 0), c.dh - 1) * stride]; };
		const int hx = hmax / c.h, vx = vmax / c.v;
		const bool integral = hmax % c.h == 0 && vmax % c.v == 0;
		if (!integral) fail("fractional sampling ratios are not supported");
		const int ow = c.dw * hx, oh = c.dh * vx; // >= W, H
		std::vector<uint8_t> out((size_t)W * H);
		std::vector<uint8_t> line((size_t)ow + 2);
		const bool fancyH = hx == 2 && c.dw > 2;
		for (int oy = 0; oy < (int)H && oy < oh; oy++) {
			const int iy = oy / vx;
			// vertical part: the (virtual) input row this output row is expanded from, as 16-bit sums where the 2-tap vertical filter applies
			if (hx == 1 && vx == 1) { memcpy(&out[(size_t)oy * W], row(iy), W); continue; }
			if (hx == 2 && vx == 1) {
				const uint8_t *in = row(iy);
				if (fancyH) { // h2v1_fancy_upsample
					int i = 0;
					int v = in[0];
					line[0] = (uint8_t)v; line[1] = (uint8_t)((v * 3 + in[1] + 2) >> 2);
					for (i = 1; i < c.dw - 1; i++) { v = in[i] * 3; line[2 * i] = (uint8_t)((v + in[i - 1] + 1) >> 2); line[2 * i + 1] = (uint8_t)((v + in[i + 1] + 2) >> 2); }
					v = in[c.dw - 1];
					line[2 * i] = (uint8_t)((v * 3 + in[c.dw - 2] + 1) >> 2); line[2 * i + 1] = (uint8_t)v;
				} else for (int i = 0; i < c.dw; i++) line[2 * i] = line[2 * i + 1] = in[i];
				memcpy(&out[(size_t)oy * W], line.data(), W);
				continue;
			}
			if (hx == 2 && vx == 2 && fancyH) { // h2v2_fancy_upsample: 3/4 of the nearer input row + 1/4 of the further one, then the same across
				const uint8_t *in0 = row(iy), *in1 = row((oy & 1) ? iy + 1 : iy - 1);
				auto colsum = [&](int i) { return in0[i] * 3 + in1[i]; };
				int thiscol = colsum(0), nextcol = colsum(1), lastcol;
				line[0] = (uint8_t)((thiscol * 4 + 8) >> 4); line[1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
				lastcol = thiscol; thiscol = nextcol;
				int i;
				for (i = 1; i < c.dw - 1; i++) {
					nextcol = colsum(i + 1);
					line[2 * i] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4); line[2 * i + 1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
					lastcol = thiscol; thiscol = nextcol;
				}
				line[2 * i] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4); line[2 * i + 1] = (uint8_t)((thiscol * 4 + 7) >> 4);
				memcpy(&out[(size_t)oy * W], line.data(), W);
				continue;
			}
			if (hx == 1 && vx == 2) { // h1v2_fancy_upsample (libjpeg-turbo): 3/4 nearer row + 1/4 further row, bias 1 going up, 2 going down
				const uint8_t *in0 = row(iy), *in1 = row((oy & 1) ? iy + 1 : iy - 1);
				const int bias = (oy & 1) ? 2 : 1;
				uint8_t *o = &out[(size_t)oy * W];
				for (uint32_t i = 0; i < W; i++) o[i] = (uint8_t)((in0[i] * 3 + in1[i] + bias) >> 2);
				continue;
			}
			{ // int_upsample / h2v2_upsample: replication
				const uint8_t *in = row(iy);
				uint8_t *o = &out[(size_t)oy * W];
				for (uint32_t x = 0; x < W; x++) o[x] = in[std::min((int)x / hx, c.dw - 1)];
			}
		}
		return out;
	}

	void finish(uint32_t *w, uint32_t *h, uint32_t *channels, std::vector<uint8_t> *pixels) {
		inverseTransform();
		*w = W; *h = H;
		if (comps.size() == 1) {
			*channels = 1;
			const Component &c = comps[0];
			pixels->resize((size_t)W * H);
			for (uint32_t y = 0; y < H; y++) memcpy(&(*pixels)[(size_t)y * W], &c.plane[(size_t)y * c.wb * 8], W);
			return;
		}
		*channels = 3;
		const std::vector<uint8_t> p0 = upsample(comps[0]), p1 = upsample(comps[1]), p2 = upsample(comps[2]);
		pixels->resize((size_t)W * H * 3);
		// the colour space: JFIF means YCbCr; an Adobe marker says (transform 0 = RGB, 1 = YCbCr); neither: RGB iff the component ids spell it
		bool ycc = true;
		if (sawJfif) ycc = true;
		else if (sawAdobe) ycc = adobeTransform != 0;
		else if (comps[0].id == 'R' && comps[1].id == 'G' && comps[2].id == 'B') ycc = false;
		if (!ycc) {
			for (size_t i = 0; i < (size_t)W * H; i++) { (*pixels)[3 * i] = p0[i]; (*pixels)[3 * i + 1] = p1[i]; (*pixels)[3 * i + 2] = p2[i]; }
			return;
		}
		// jdcolor.c build_ycc_rgb_table / ycc_rgb_convert: 16-bit fixed point
		int32_t crR[256], cbB[256], crG[256], cbG[256];
		for (int i = 0; i < 256; i++) {
			const int32_t x = i - 128;
			crR[i] = (91881 * x + 32768) >> 16;
			cbB[i] = (116130 * x + 32768) >> 16;
			crG[i] = -46802 * x;
			cbG[i] = -22554 * x + 32768;
		}
		auto clamp = [](int32_t v) -> uint8_t { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); };
		for (size_t i = 0; i < (size_t)W * H; i++) {
			const int32_t y = p0[i], cb = p1[i], cr = p2[i];
			(*pixels)[3 * i] = clamp(y + crR[cr]);
			(*pixels)[3 * i + 1] = clamp(y + ((cbG[cb] + crG[cr]) >> 16));
			(*pixels)[3 * i + 2] = clamp(y + cbB[cb]);
		}
	}
};

} // namespace

Error DecodeJPEG(const std::string &name, const std::vector<uint8_t> &file, uint32_t *width, uint32_t *height, uint32_t *channels, std::vector<uint8_t> *pixels) {
	try {
		Decoder d(file);
		d.parse();
		d.finish(width, height, channels, pixels);
	} catch (const Fail &e) {
		return Error{POLARIS_E_BAD_SCENE, "texture: jpeg: " + e.msg + " while loading " + name};
	} catch (const std::bad_alloc &) {
		return Error{POLARIS_E_BAD_SCENE, "texture: jpeg: out of memory while loading " + name};
	}
	return Error::Nil();
}

} // namespace texture
} // namespace polaris
