// material_expr.cpp -- see material_expr.hpp.  A recursive-descent parser for the language the
// reference generates a yacc parser for (asset/material/material_expr.y); the token rules follow
// its hand-written lexer (:205-322) rule for rule, the semantic checks follow node.go:108-258.
#include "material_expr.hpp"

#include <algorithm>
#include <cctype>
#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "polaris_hip.h"

namespace polaris {
namespace material {

const char *const ParamReflectance = "reflectance", *const ParamSpecularity = "specularity",
                  *const ParamTransmittance = "transmittance", *const ParamRadiance = "radiance", *const ParamIntIOR = "intIOR",
                  *const ParamExtIOR = "extIOR", *const ParamScale = "scale", *const ParamRoughness = "roughness";

namespace {

std::string quoted(const std::string &s) { return "\"" + s + "\""; } // Go's %q for plain ASCII

std::string upper(std::string s) {
	for (char &c : s) c = (char)std::toupper((unsigned char)c);
	return s;
}

// Common entries of the reference's IOR table (asset/material/ior.go:12-257), upper-cased keys.
struct NamedIOR { const char *name; float ior; };
const NamedIOR kIORs[] = {
    {"AIR", 1.0002926f},      {"ALCOHOL", 1.329f},     {"ALUMINUM", 1.44f},   {"AMBER", 1.546f},     {"BRONZE", 1.18f},
    {"CHROMIUM", 2.97f},      {"COPPER", 1.10f},       {"CRYSTAL", 2.00f},    {"DIAMOND", 2.417f},   {"EMERALD", 1.576f},
    {"GLASS", 1.51714f},      {"GLASS, CROWN", 1.520f}, {"GLASS, FLINT, DENSE", 1.66f}, {"GLASS, FLINT, LIGHT", 1.58038f},
    {"GOLD", 0.47f},          {"ICE", 1.309f},         {"IRON", 1.51f},       {"JADEITE", 1.665f},   {"LEAD", 2.01f},
    {"MERCURY (LIQ)", 1.62f}, {"NYLON", 1.53f},        {"OBSIDIAN", 1.489f},  {"OPAL", 1.450f},      {"PEARL", 1.530f},
    {"PLASTIC", 1.460f},      {"PLEXIGLAS", 1.50f},    {"POLYSTYRENE", 1.55f}, {"QUARTZ", 1.544f},   {"RUBY", 1.760f},
    {"SAPPHIRE", 1.760f},     {"SILVER", 0.18f},       {"STEEL", 2.50f},      {"TOPAZ", 1.620f},     {"WATER", 1.33157f},
};

// ---- lexer, material_expr.y:196-343 ------------------------------------------------------------
enum Tok {
	T_EOF, T_LPAREN, T_RPAREN, T_LCURLY, T_RCURLY, T_COMMA, T_COLON, T_FLOAT, T_MATERIAL_NAME, T_TEXTURE,
	T_REFLECTANCE, T_SPECULARITY, T_TRANSMITTANCE, T_RADIANCE, T_INT_IOR, T_EXT_IOR, T_SCALE, T_ROUGHNESS,
	T_DIFFUSE, T_CONDUCTOR, T_ROUGH_CONDUCTOR, T_DIELECTRIC, T_ROUGH_DIELECTRIC, T_EMISSIVE,
	T_MIX, T_MIX_MAP, T_BUMP_MAP, T_NORMAL_MAP, T_DISPERSE
};

struct Lexer {
	const std::string &line;
	size_t pos = 0;
	std::string firstError; // "keep the first error we encountered", :346-351
	float fVal = 0;
	std::string sVal;

	explicit Lexer(const std::string &s) : line(s) {}
	void error(const std::string &e) { if (firstError.empty()) firstError = e; }
	int next() { return pos < line.size() ? (unsigned char)line[pos++] : -1; }

	Tok lex() {
		for (;;) {
			const int c = next();
			switch (c) {
			case -1: return T_EOF;
			case '(': return T_LPAREN;
			case ')': return T_RPAREN;
			case ',': return T_COMMA;
			case '{': return T_LCURLY;
			case '}': return T_RCURLY;
			case ':': return T_COLON;
			case ' ': case '\t': case '\n': case '\r': continue;
			case '"': return lexLiteral();
			default:
				if ((c >= '0' && c <= '9') || c == '.') return lexFloat(c);
				return lexIdentifier(c);
			}
		}
	}
	Tok lexFloat(int c) { // :236-261
		std::string buf(1, (char)c);
		for (;;) {
			c = next();
			if ((c >= '0' && c <= '9') || c == '.' || c == 'e' || c == 'E' || c == '+' || c == '-') { buf.push_back((char)c); continue; }
			break;
		}
		if (c != -1) pos--;
		errno = 0;
		char *end = nullptr;
		const float v = std::strtof(buf.c_str(), &end);
		if (end == buf.c_str() || *end != 0 || (errno == ERANGE && std::isinf(v))) {
			error("invalid float value " + quoted(buf));
			return T_EOF;
		}
		fVal = v;
		return T_FLOAT;
	}
	Tok lexLiteral() { // :264-284
		std::string buf;
		int c;
		for (;;) {
			c = next();
			if (c == -1 || c == '"') break;
			buf.push_back((char)c);
		}
		if (c == -1) { error("unterminated string litera"); return T_EOF; } // the reference's spelling
		sVal = buf;
		return IsTextureName(buf) ? T_TEXTURE : T_MATERIAL_NAME;
	}
	Tok lexIdentifier(int c) { // :287-330
		std::string buf(1, (char)c);
		for (;;) {
			c = next();
			if ((c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '_') { buf.push_back((char)c); continue; }
			break;
		}
		if (c != -1) pos--;
		sVal = buf;
		static const struct { const char *name; Tok tok; } words[] = {
		    {"diffuse", T_DIFFUSE}, {"conductor", T_CONDUCTOR}, {"roughConductor", T_ROUGH_CONDUCTOR}, {"dielectric", T_DIELECTRIC},
		    {"roughDielectric", T_ROUGH_DIELECTRIC}, {"emissive", T_EMISSIVE}, {"mix", T_MIX}, {"mixMap", T_MIX_MAP},
		    {"bumpMap", T_BUMP_MAP}, {"normalMap", T_NORMAL_MAP}, {"disperse", T_DISPERSE}, {"reflectance", T_REFLECTANCE},
		    {"specularity", T_SPECULARITY}, {"transmittance", T_TRANSMITTANCE}, {"radiance", T_RADIANCE}, {"intIOR", T_INT_IOR},
		    {"extIOR", T_EXT_IOR}, {"scale", T_SCALE}, {"roughness", T_ROUGHNESS}};
		for (const auto &w : words)
			if (buf == w.name) return w.tok;
		error("invalid expression " + quoted(buf));
		return T_EOF;
	}
};

// ---- parser, material_expr.y:76-176 ---------------------------------------------------------------
struct Parser {
	Lexer lx;
	Tok tok;
	bool failed = false;
	int depth = 0; // operator nesting (the reference's yacc stack gives out at 1000 states)

	explicit Parser(const std::string &s) : lx(s) { tok = lx.lex(); }
	void advance() { tok = lx.lex(); }
	bool fail() { failed = true; lx.error("syntax error"); return false; }
	bool expect(Tok t) {
		if (tok != t) return fail();
		advance();
		return true;
	}
	static bool isBxdf(Tok t) { return t >= T_DIFFUSE && t <= T_EMISSIVE; }
	static bool isOp(Tok t) { return t >= T_MIX && t <= T_DISPERSE; }

	bool float3(float out[3]) {
		if (!expect(T_LCURLY)) return false;
		for (int i = 0; i < 3; i++) {
			if (tok != T_FLOAT) return fail();
			out[i] = lx.fVal;
			advance();
			if (i < 2 && !expect(T_COMMA)) return false;
		}
		return expect(T_RCURLY);
	}
	bool texture(std::string *out) {
		if (tok != T_TEXTURE) return fail();
		*out = lx.sVal;
		advance();
		return true;
	}
	bool param(Param *p) {
		const Tok name = tok;
		if (name < T_REFLECTANCE || name > T_ROUGHNESS) return fail();
		p->name = lx.sVal;
		advance();
		if (!expect(T_COLON)) return false;
		switch (name) {
		case T_REFLECTANCE: case T_SPECULARITY: case T_TRANSMITTANCE: case T_RADIANCE: // float3_or_texture
			if (tok == T_TEXTURE) { p->kind = Param::Texture; return texture(&p->s); }
			p->kind = Param::Vec3;
			return float3(p->v);
		case T_INT_IOR: case T_EXT_IOR: // float_or_name
			if (tok == T_FLOAT) { p->kind = Param::Float; p->f = lx.fVal; advance(); return true; }
			if (tok == T_MATERIAL_NAME) { p->kind = Param::MaterialName; p->s = lx.sVal; advance(); return true; }
			return fail();
		case T_SCALE:
			if (tok != T_FLOAT) return fail();
			p->kind = Param::Float; p->f = lx.fVal; advance();
			return true;
		default: // roughness: float_or_texture
			if (tok == T_FLOAT) { p->kind = Param::Float; p->f = lx.fVal; advance(); return true; }
			if (tok == T_TEXTURE) { p->kind = Param::Texture; return texture(&p->s); }
			return fail();
		}
	}
	std::unique_ptr<Expr> bxdfSpec() {
		auto e = std::make_unique<Expr>();
		e->kind = Expr::Bxdf;
		e->bxdfType = BxdfTypeFromName(lx.sVal);
		advance();
		if (!expect(T_LPAREN)) return nullptr;
		if (tok != T_RPAREN) {
			for (;;) {
				Param p;
				if (!param(&p)) return nullptr;
				e->params.push_back(std::move(p));
				if (tok != T_COMMA) break;
				advance();
			}
		}
		if (!expect(T_RPAREN)) return nullptr;
		return e;
	}
	std::unique_ptr<Expr> arg() { // bxdf_or_op_spec
		if (isBxdf(tok)) return bxdfSpec();
		if (isOp(tok)) return opSpec();
		if (tok == T_MATERIAL_NAME) {
			auto e = std::make_unique<Expr>();
			e->kind = Expr::MaterialRef;
			e->ref = lx.sVal;
			advance();
			return e;
		}
		fail();
		return nullptr;
	}
	std::unique_ptr<Expr> opSpec() {
		struct Guard { int &d; Guard(int &x) : d(x) { d++; } ~Guard() { d--; } } guard(depth);
		if (depth > 200) { lx.error("expression nested too deeply"); failed = true; return nullptr; }
		const Tok op = tok;
		auto e = std::make_unique<Expr>();
		advance();
		if (!expect(T_LPAREN)) return nullptr;
		if (!(e->left = arg())) return nullptr;
		if (!expect(T_COMMA)) return nullptr;
		switch (op) {
		case T_MIX:
			e->kind = Expr::Mix;
			if (!(e->right = arg()) || !expect(T_COMMA)) return nullptr;
			if (tok != T_FLOAT) { fail(); return nullptr; }
			e->weight = lx.fVal;
			advance();
			break;
		case T_MIX_MAP:
			e->kind = Expr::MixMap;
			if (!(e->right = arg()) || !expect(T_COMMA) || !texture(&e->texture)) return nullptr;
			break;
		case T_BUMP_MAP:
			e->kind = Expr::BumpMap;
			if (!texture(&e->texture)) return nullptr;
			break;
		case T_NORMAL_MAP:
			e->kind = Expr::NormalMap;
			if (!texture(&e->texture)) return nullptr;
			break;
		default: // disperse
			e->kind = Expr::Disperse;
			if (!expect(T_INT_IOR) || !expect(T_COLON) || !float3(e->intIOR) || !expect(T_COMMA) || !expect(T_EXT_IOR) ||
			    !expect(T_COLON) || !float3(e->extIOR))
				return nullptr;
		}
		if (!expect(T_RPAREN)) return nullptr;
		return e;
	}
	std::unique_ptr<Expr> materialDef() {
		std::unique_ptr<Expr> e;
		if (isBxdf(tok)) e = bxdfSpec();
		else if (isOp(tok)) e = opSpec();
		else fail();
		if (e && tok != T_EOF) { fail(); e.reset(); }
		return e;
	}
};

bool allowed(uint32_t bxdf, const std::string &p) { // bxdfAllowedParameters, node.go:22-55
	const bool spec = p == ParamSpecularity, ii = p == ParamIntIOR, ei = p == ParamExtIOR, rough = p == ParamRoughness,
	           trans = p == ParamTransmittance;
	switch (bxdf) {
	case POLARIS_BXDF_EMISSIVE: return p == ParamRadiance || p == ParamScale;
	case POLARIS_BXDF_DIFFUSE: return p == ParamReflectance;
	case POLARIS_BXDF_CONDUCTOR: return spec || ii || ei;
	case POLARIS_BXDF_ROUGH_CONDUCTOR: return spec || ii || ei || rough;
	case POLARIS_BXDF_DIELECTRIC: return spec || trans || ii || ei;
	case POLARIS_BXDF_ROUGH_DIELECTRIC: return spec || trans || ii || ei || rough;
	}
	return false;
}

Error bad(const std::string &msg) { return Error{POLARIS_E_BAD_SCENE, msg}; }

Error validateParam(const Param &p) { // BxdfParamNode.Validate, node.go:136-163
	if (p.name == ParamReflectance) {
		if (p.kind == Param::Vec3 && (p.v[0] >= 1.0f || p.v[1] >= 1.0f || p.v[2] >= 1.0f))
			return bad("energy conservation violation for Parameter " + quoted(p.name) + "; ensure that all vector components are < 1.0");
	} else if (p.name == ParamSpecularity || p.name == ParamTransmittance) {
		if (p.kind == Param::Vec3 && (p.v[0] > 1.0f || p.v[1] > 1.0f || p.v[2] > 1.0f))
			return bad("energy conservation violation for Parameter " + quoted(p.name) + "; ensure that all vector components are <= 1.0");
	} else if (p.name == ParamRoughness) {
		if (p.kind == Param::Float && p.f > 1.0f) return bad("values for Parameter " + quoted(p.name) + " must be in the [0, 1] range");
	} else if (p.name == ParamIntIOR || p.name == ParamExtIOR) {
		if (p.kind == Param::MaterialName) {
			float ior;
			if (Error e = IOR(p.s, &ior)) return e;
		}
	}
	// Value.Validate(), node.go:108-134
	if (p.kind == Param::MaterialName && p.s.empty()) return bad("material name cannot be empty");
	if (p.kind == Param::Texture && p.s.empty()) return bad("no texture path specified");
	return Error::Nil();
}

} // namespace

bool IsTextureName(const std::string &s) { // supportedImageRegex, texture.go:5-8
	static const char *const exts[] = {"jpg", "jpeg", "gif", "png", "tga", "tiff", "bmp", "pnm", "hdr", "exr", "webp"};
	const size_t dot = s.rfind('.');
	if (dot == std::string::npos) return false;
	std::string ext = s.substr(dot + 1);
	for (char &c : ext) c = (char)std::tolower((unsigned char)c);
	for (const char *e : exts)
		if (ext == e) return true;
	return false;
}

Error IOR(const std::string &name, float *out) {
	const std::string key = upper(name);
	for (const NamedIOR &e : kIORs)
		if (key == e.name) { *out = e.ior; return Error::Nil(); }
	return bad("unknown material name " + quoted(name) + "; try specifying the IOR manually");
}

const char *BxdfName(uint32_t t) {
	switch (t) {
	case POLARIS_BXDF_EMISSIVE: return "emissive";
	case POLARIS_BXDF_DIFFUSE: return "diffuse";
	case POLARIS_BXDF_CONDUCTOR: return "conductor";
	case POLARIS_BXDF_ROUGH_CONDUCTOR: return "roughConductor";
	case POLARIS_BXDF_DIELECTRIC: return "dielectric";
	case POLARIS_BXDF_ROUGH_DIELECTRIC: return "roughDielectric";
	}
	return "invalid";
}

uint32_t BxdfTypeFromName(const std::string &n) {
	for (uint32_t t = POLARIS_BXDF_EMISSIVE; t <= POLARIS_BXDF_ROUGH_DIELECTRIC; t <<= 1)
		if (n == BxdfName(t)) return t;
	return 0;
}

Error Expr::Validate() const {
	switch (kind) {
	case MaterialRef:
		if (ref.empty()) return bad("material name cannot be empty");
		return Error::Nil();
	case Bxdf:
		if (bxdfType == 0) return bad("invalid BXDF type");
		for (const Param &p : params) {
			if (!allowed(bxdfType, p.name))
				return bad("bxdf type " + quoted(BxdfName(bxdfType)) + " does not support Parameter " + quoted(p.name));
			if (Error e = validateParam(p)) return e;
		}
		return Error::Nil();
	case Mix:
	case MixMap: { // node.go:205-240: the arguments are validated recursively
		const char *who = kind == Mix ? "mix" : "mixMap";
		const Expr *args[2] = {left.get(), right.get()};
		for (int i = 0; i < 2; i++) {
			if (!args[i]) return bad("missing expression argument " + std::to_string(i) + " for " + quoted(who));
			if (Error e = args[i]->Validate()) return bad(std::string(who) + " argument " + std::to_string(i) + ": " + e.msg);
		}
		if (kind == MixMap) {
			if (texture.empty()) return bad("MixMap: no texture path specified");
		} else if (weight < 0 || weight > 1.0f) {
			return bad("Mix: mix weight must be in the [0, 1] range");
		}
		return Error::Nil();
	}
	case BumpMap:
	case NormalMap: // node.go:169-189: only the operator's own fields (the reference does not descend)
		if (!left) return bad(std::string("missing expression argument for ") + quoted(kind == BumpMap ? "BumpMap" : "NormalMap"));
		if (texture.empty()) return bad(std::string(kind == BumpMap ? "BumpMap" : "NormalMap") + ": no texture path specified");
		return Error::Nil();
	case Disperse: // node.go:191-199
		if (!left) return bad("missing expression argument for \"Disperse\"");
		if (std::max(intIOR[0], std::max(intIOR[1], intIOR[2])) == 0.0f && std::max(extIOR[0], std::max(extIOR[1], extIOR[2])) == 0.0f)
			return bad("Disperse: at least one of the intIOR and extIOR parameters must contain a non-zero value");
		return Error::Nil();
	}
	return bad("unsupported node");
}

Error ParseExpression(const std::string &input, std::unique_ptr<Expr> *out) {
	Parser p(input);
	std::unique_ptr<Expr> e = p.materialDef();
	if (!p.lx.firstError.empty() || !e) return bad(p.lx.firstError.empty() ? "syntax error" : p.lx.firstError);
	if (out) *out = std::move(e);
	return Error::Nil();
}

} // namespace material
} // namespace polaris
