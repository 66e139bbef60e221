// wavefront_reader.cpp -- see wavefront_reader.hpp.  Statement handling, error texts and the
// order of side effects follow asset/scene/reader/wavefront.go.
#include "wavefront_reader.hpp"

#include <array>
#include <cerrno>
#include <cstdarg>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

#include "material_expr.hpp"

namespace polaris {
namespace reader {

using compiler::MeshInstance;
using compiler::Primitive;
using types::Mat4;
using types::Quat;
using types::Vec3;

namespace {

Error bad(const std::string &m) { return Error{POLARIS_E_BAD_SCENE, m}; }
constexpr size_t kMaxIncludeDepth = 32;

std::vector<std::string> fields(const std::string &line) { // strings.Fields
	std::vector<std::string> out;
	std::istringstream is(line);
	std::string t;
	while (is >> t) out.push_back(t);
	return out;
}

std::string fmt(const char *f, ...) __attribute__((format(printf, 1, 2)));
std::string fmt(const char *f, ...) {
	char buf[1024];
	va_list ap;
	va_start(ap, f);
	vsnprintf(buf, sizeof buf, f, ap);
	va_end(ap);
	return buf;
}

// strconv.ParseFloat(s, 32)
Error parseFloatToken(const std::string &s, double *out) {
	errno = 0;
	char *end = nullptr;
	const float v = std::strtof(s.c_str(), &end);
	if (s.empty() || end == s.c_str() || *end != 0) return bad("strconv.ParseFloat: parsing \"" + s + "\": invalid syntax");
	if (errno == ERANGE && std::isinf(v)) return bad("strconv.ParseFloat: parsing \"" + s + "\": value out of range");
	*out = (double)v;
	return Error::Nil();
}

Error parseFloat32(const std::vector<std::string> &tok, float *out) { // wavefront.go:786-797
	if (tok.size() < 2) return bad(fmt("unsupported syntax for \"%s\"; expected 1 argument; got %zu", tok[0].c_str(), tok.size() - 1));
	double v;
	if (Error e = parseFloatToken(tok[1], &v)) return e;
	*out = (float)v;
	return Error::Nil();
}

Error parseVec3(const std::vector<std::string> &tok, Vec3 *out) { // wavefront.go:800-814
	if (tok.size() < 4) return bad(fmt("unsupported syntax for \"%s\"; expected 3 arguments; got %zu", tok[0].c_str(), tok.size() - 1));
	float v[3];
	for (int i = 0; i < 3; i++) {
		double d;
		if (Error e = parseFloatToken(tok[1 + i], &d)) return e;
		v[i] = (float)d;
	}
	*out = {v[0], v[1], v[2]};
	return Error::Nil();
}

Error parseVec2(const std::vector<std::string> &tok, std::array<float, 2> *out) { // wavefront.go:817-830
	if (tok.size() < 3) return bad(fmt("unsupported syntax for \"%s\"; expected 2 arguments; got %zu", tok[0].c_str(), tok.size() - 1));
	for (int i = 0; i < 2; i++) {
		double d;
		if (Error e = parseFloatToken(tok[1 + i], &d)) return e;
		(*out)[i] = (float)d;
	}
	return Error::Nil();
}

std::string vec3Text(Vec3 v) { return fmt("{%f, %f, %f}", v.x, v.y, v.z); } // Vec3.String(), vector.go:59-61
std::string floatText(float v) { return fmt("%.9g", v); }                    // re-parses to the same float32, like Go's %v

std::string dirOf(const std::string &path) {
	const size_t slash = path.find_last_of('/');
	return slash == std::string::npos ? std::string(".") : (slash == 0 ? std::string("/") : path.substr(0, slash));
}

bool readAll(const std::string &path, std::string *out) {
	std::ifstream f(path, std::ios::binary);
	if (!f) return false;
	std::ostringstream ss;
	ss << f.rdbuf();
	*out = ss.str();
	return true;
}

} // namespace

std::string WavefrontMaterial::GetExpression() const {
	if (!MaterialExpression.empty()) return MaterialExpression;
	const bool isSpecular = Ks.MaxComponent() > 0.0f || !KsTex.empty();
	const bool isEmissive = Ke.MaxComponent() > 0.0f || !KeTex.empty();
	std::vector<std::string> args;
	auto texArg = [](const char *name, const std::string &tex) { return std::string(name) + ": \"" + tex + "\""; };
	auto vecArg = [](const char *name, Vec3 v) { return std::string(name) + ": " + vec3Text(v); };
	uint32_t bxdf;
	if (isSpecular) {
		bxdf = Ni == 0.0f ? POLARIS_BXDF_CONDUCTOR : POLARIS_BXDF_DIELECTRIC;
		if (!KsTex.empty()) args.push_back(texArg(material::ParamSpecularity, KsTex));
		else if (Ks.MaxComponent() > 0.0f) args.push_back(vecArg(material::ParamSpecularity, Ks));
		if (bxdf == POLARIS_BXDF_DIELECTRIC) {
			if (!TfTex.empty()) args.push_back(texArg(material::ParamTransmittance, TfTex));
			else if (Tf.MaxComponent() > 0.0f) args.push_back(vecArg(material::ParamTransmittance, Tf));
			args.push_back(std::string(material::ParamIntIOR) + ": " + floatText(Ni));
		}
	} else if (isEmissive) {
		bxdf = POLARIS_BXDF_EMISSIVE;
		if (!KeTex.empty()) args.push_back(texArg(material::ParamRadiance, KeTex));
		else if (Ke.MaxComponent() > 0.0f) args.push_back(vecArg(material::ParamRadiance, Ke));
		if (KeScaler != 0) args.push_back(std::string(material::ParamScale) + ": " + floatText(KeScaler));
	} else {
		bxdf = POLARIS_BXDF_DIFFUSE;
		if (!KdTex.empty()) args.push_back(texArg(material::ParamReflectance, KdTex));
		else if (Kd.MaxComponent() > 0.0f) args.push_back(vecArg(material::ParamReflectance, Kd));
	}
	std::string expr = std::string(material::BxdfName(bxdf)) + "(";
	for (size_t i = 0; i < args.size(); i++) expr += (i ? ", " : "") + args[i];
	expr += ")";
	// bump modifiers: a normal map wins over a bump map (wavefront.go:117-121)
	if (!NormalTex.empty()) expr = "normalMap(" + expr + ", \"" + NormalTex + "\")";
	else if (!BumpTex.empty()) expr = "bumpMap(" + expr + ", \"" + BumpTex + "\")";
	return expr;
}

Error WavefrontSceneReader::emitError(const std::string &file, int line, const std::string &msg) const { // :261-279
	std::string stack;
	for (size_t i = 0; i < errStack.size(); i++) stack += (i ? "\n" : "") + errStack[i];
	std::string text = file.empty() ? "error: " + msg + "\n" + stack : fmt("[%s: %d] error: ", file.c_str(), line) + msg + "\n" + stack;
	while (!text.empty() && text.back() == '\n') text.pop_back();
	size_t lead = 0;
	while (lead < text.size() && text[lead] == '\n') lead++;
	return bad(text.substr(lead));
}

int WavefrontSceneReader::defaultMaterial() { // :292-304
	auto it = matNameToIndex.find("");
	if (it == matNameToIndex.end()) {
		WavefrontMaterial m;
		m.Kd = {0.7f, 0.7f, 0.7f};
		materials.push_back(m);
		it = matNameToIndex.emplace("", int(materials.size()) - 1).first;
	}
	curMaterial = it->second;
	return curMaterial;
}

void WavefrontSceneReader::verifyLastParsedMesh() { // :443-449
	if (!rawScene.meshes.empty() && rawScene.meshes.back().primitives.empty()) rawScene.meshes.pop_back();
}

Error WavefrontSceneReader::parseFile(const std::string &path, bool asMaterials) {
	std::string content;
	if (!readAll(path, &content)) return bad("open " + path + ": no such file or directory");
	return asMaterials ? ParseMaterials(path, content) : Parse(path, content);
}

Error WavefrontSceneReader::Parse(const std::string &name, const std::string &content) {
	// offsets of this file's 1-based indices into the global lists (files may `call` others), :311-317
	const int relV = int(vertexList.size()), relUv = int(uvList.size()), relN = int(normalList.size());
	std::istringstream in(content);
	std::string line;
	int lineNum = 0;
	while (std::getline(in, line)) {
		lineNum++;
		const std::vector<std::string> tok = fields(line);
		if (tok.empty() || tok[0][0] == '#') continue;
		const std::string &cmd = tok[0];
		Error err;
		if (cmd == "call" || cmd == "mtllib") {
			if (tok.size() != 2) return emitError(name, lineNum, fmt("unsupported syntax for \"%s\"; expected 1 argument; got %zu", cmd.c_str(), tok.size() - 1));
			// (the reference recurses without a limit and dies on a file that calls itself)
			if (errStack.size() >= kMaxIncludeDepth) return emitError(name, lineNum, "files nested too deeply (a file that calls itself?)");
			errStack.insert(errStack.begin(), fmt("referenced from %s:%d [%s]", name.c_str(), lineNum, cmd.c_str()));
			std::string inc = tok[1];
			for (char &c : inc) if (c == '\\') c = '/';
			const std::string path = dirOf(name) + "/" + inc; // asset.NewResource, resource.go:46-64
			std::string body;
			if (!readAll(path, &body)) return emitError(name, lineNum, "open " + path + ": no such file or directory");
			if (Error e = cmd == "call" ? Parse(path, body) : ParseMaterials(path, body)) return e;
			errStack.erase(errStack.begin());
		} else if (cmd == "usemtl") {
			if (tok.size() != 2) return emitError(name, lineNum, fmt("unsupported syntax for 'usemtl'; expected 1 argument; got %zu", tok.size() - 1));
			const auto it = matNameToIndex.find(tok[1]);
			if (it == matNameToIndex.end()) return emitError(name, lineNum, "undefined material with name \"" + tok[1] + "\"");
			curMaterial = it->second;
		} else if (cmd == "v" || cmd == "vn") {
			Vec3 v;
			if ((err = parseVec3(tok, &v))) return emitError(name, lineNum, err.msg);
			(cmd == "v" ? vertexList : normalList).push_back(v);
		} else if (cmd == "vt") {
			std::array<float, 2> v{};
			if ((err = parseVec2(tok, &v))) return emitError(name, lineNum, err.msg);
			uvList.push_back(v);
		} else if (cmd == "g" || cmd == "o") {
			if (tok.size() < 2)
				return emitError(name, lineNum, fmt("unsupported syntax for \"%s\"; expected 1 argument for object name; got %zu", cmd.c_str(), tok.size() - 1));
			verifyLastParsedMesh();
			compiler::Mesh m;
			m.name = tok[1];
			rawScene.meshes.push_back(std::move(m));
		} else if (cmd == "f") {
			std::vector<Primitive> prims;
			if ((err = parseFace(tok, relV, relUv, relN, &prims))) return emitError(name, lineNum, err.msg);
			if (rawScene.meshes.empty()) { // no object defined yet: a default one
				compiler::Mesh m;
				m.name = "default";
				rawScene.meshes.push_back(std::move(m));
			}
			auto &dst = rawScene.meshes.back().primitives;
			dst.insert(dst.end(), prims.begin(), prims.end());
		} else if (cmd == "camera_fov") {
			if ((err = parseFloat32(tok, &rawScene.camera.fov))) return emitError(name, lineNum, err.msg);
		} else if (cmd == "camera_eye" || cmd == "camera_look" || cmd == "camera_up") {
			Vec3 *dst = cmd == "camera_eye" ? &rawScene.camera.eye : (cmd == "camera_look" ? &rawScene.camera.look : &rawScene.camera.up);
			Vec3 v; // the reference assigns the (zero) result even when parsing fails
			err = parseVec3(tok, &v);
			*dst = v;
			if (err) return emitError(name, lineNum, err.msg);
		} else if (cmd == "instance") {
			MeshInstance inst;
			if ((err = parseMeshInstance(tok, &inst))) return emitError(name, lineNum, err.msg);
			rawScene.instances.push_back(inst);
		}
	}
	verifyLastParsedMesh();
	return Error::Nil();
}

// instance mesh_name tX tY tZ yaw pitch roll sX sY sZ, wavefront.go:458-531
Error WavefrontSceneReader::parseMeshInstance(const std::vector<std::string> &tok, MeshInstance *out) {
	if (tok.size() != 11)
		return bad(fmt("unsupported syntax for \"instance\"; expected 10 arguments: mesh_name tX tY tZ yaw pitch roll sX sY sZ; got %zu", tok.size() - 1));
	int meshIndex = -1;
	for (size_t i = 0; i < rawScene.meshes.size(); i++)
		if (rawScene.meshes[i].name == tok[1]) { meshIndex = int(i); break; }
	if (meshIndex == -1) return bad("unknown mesh with name \"" + tok[1] + "\"");
	float t[3], r[3], s[3];
	for (int i = 0; i < 9; i++) {
		double v;
		if (Error e = parseFloatToken(tok[2 + i], &v)) return e;
		if (i < 3) t[i] = (float)v;
		else if (i < 6) r[i - 3] = (float)(v * (M_PI / 180.0)); // degrees -> radians in double, then float32
		else s[i - 6] = (float)v;
	}
	// M = S * R * T as the reference composes it (scaleMat.Mul4(rotMat.Mul4(transMat)))
	const Quat yaw = Quat::FromAxisAngle({1, 0, 0}, r[0]), pitch = Quat::FromAxisAngle({0, 1, 0}, r[1]), roll = Quat::FromAxisAngle({0, 0, 1}, r[2]);
	const Mat4 rotMat = roll.Mul(pitch.Mul(yaw)).Normalize().ToMat4();
	const Mat4 scaleMat = Mat4::Scale({s[0], s[1], s[2]}), transMat = Mat4::Translate({t[0], t[1], t[2]});
	const Mat4 m = scaleMat.Mul4(rotMat.Mul4(transMat));

	// the instance box is the mesh box moved by the TRANSLATION only (wavefront.go:514-519)
	Vec3 lo{3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f}, hi{-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
	for (const Primitive &p : rawScene.meshes[meshIndex].primitives) { lo = types::MinVec3(lo, p.bbox[0]); hi = types::MaxVec3(hi, p.bbox[1]); }
	const types::Vec4 a = transMat.Mul4x1({lo.x, lo.y, lo.z, 1}), b = transMat.Mul4x1({hi.x, hi.y, hi.z, 1});
	const Vec3 mn{a.x, a.y, a.z}, mx{b.x, b.y, b.z};
	out->meshIndex = uint32_t(meshIndex);
	memcpy(out->transform, m.m, sizeof m.m);
	out->hasBounds = true;
	out->bbox[0] = types::MinVec3(mn, mx);
	out->bbox[1] = types::MaxVec3(mn, mx);
	out->center = out->bbox[0].Add(out->bbox[1]).Mul(0.5f);
	return Error::Nil();
}

Error SelectFaceCoordIndex(const std::string &token, int coordListLen, int relOffset, int *out) { // :767-783
	errno = 0;
	char *end = nullptr;
	const long long index = std::strtoll(token.c_str(), &end, 10);
	if (token.empty() || end == token.c_str() || *end != 0) return bad("strconv.ParseInt: parsing \"" + token + "\": invalid syntax");
	if (errno == ERANGE || index > 2147483647LL || index < -2147483648LL) return bad("strconv.ParseInt: parsing \"" + token + "\": value out of range");
	const long long off = index < 0 ? (long long)coordListLen + index : (long long)relOffset + (index - 1);
	if (off < 0 || off >= coordListLen) return bad("index out of bounds");
	*out = int(off);
	return Error::Nil();
}

// wavefront.go:533-648
Error WavefrontSceneReader::parseFace(const std::vector<std::string> &tok, int relV, int relUv, int relN, std::vector<Primitive> *out) {
	if (tok.size() < 4 || tok.size() > 5)
		return bad(fmt("unsupported syntax for \"f\"; expected 3 arguments for triangular face or 4 arguments for a quad face; got %zu. "
		               "Select the triangulation option in your exporter", tok.size() - 1));
	Vec3 vertices[4], normals[4];
	float uv[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
	size_t expIndices = 0;
	bool hasNormals = false;
	for (size_t arg = 0; arg + 1 < tok.size(); arg++) {
		std::vector<std::string> vt; // strings.Split(token, "/")
		size_t start = 0;
		for (;;) {
			const size_t slash = tok[arg + 1].find('/', start);
			vt.push_back(tok[arg + 1].substr(start, slash == std::string::npos ? std::string::npos : slash - start));
			if (slash == std::string::npos) break;
			start = slash + 1;
		}
		if (arg == 0) expIndices = vt.size();
		else if (vt.size() != expIndices)
			return bad(fmt("expected each face argument to contain %zu indices; arg %zu contains %zu indices", expIndices, arg, vt.size()));
		if (vt[0].empty()) return bad(fmt("face argument %zu does not include a vertex index", arg));
		int off;
		if (Error e = SelectFaceCoordIndex(vt[0], int(vertexList.size()), relV, &off))
			return bad(fmt("could not parse vertex coord for face argument %zu: ", arg) + e.msg);
		vertices[arg] = vertexList[off];
		if (expIndices > 1 && !vt[1].empty()) {
			if (Error e = SelectFaceCoordIndex(vt[1], int(uvList.size()), relUv, &off))
				return bad(fmt("could not parse tex coord for face argument %zu: ", arg) + e.msg);
			uv[arg][0] = uvList[off][0];
			uv[arg][1] = uvList[off][1];
		}
		if (expIndices > 2 && !vt[2].empty()) {
			if (Error e = SelectFaceCoordIndex(vt[2], int(normalList.size()), relN, &off))
				return bad(fmt("could not parse normal coord for face argument %zu: ", arg) + e.msg);
			normals[arg] = normalList[off];
			hasNormals = true;
		}
	}
	if (curMaterial < 0) defaultMaterial();
	materials[curMaterial].Used = true;
	if (!hasNormals) { // face normal from the first three vertices, :595-604
		const Vec3 n = vertices[1].Sub(vertices[0]).Cross(vertices[2].Sub(vertices[0])).Normalize();
		for (Vec3 &d : normals) d = n;
	}
	const int tris[2][3] = {{0, 1, 2}, {0, 2, 3}};
	for (int t = 0; t < (tok.size() == 5 ? 2 : 1); t++) {
		Primitive p;
		for (int k = 0; k < 3; k++) {
			const int s = tris[t][k];
			p.vertices[k] = vertices[s];
			p.normals[k] = normals[s];
			p.uvs[k][0] = uv[s][0];
			p.uvs[k][1] = uv[s][1];
		}
		p.materialIndex = matNameToIndex[materials[curMaterial].Name];
		p.hasBounds = true;
		p.bbox[0] = types::MinVec3(p.vertices[0], types::MinVec3(p.vertices[1], p.vertices[2]));
		p.bbox[1] = types::MaxVec3(p.vertices[0], types::MaxVec3(p.vertices[1], p.vertices[2]));
		p.center = p.vertices[0].Add(p.vertices[1]).Add(p.vertices[2]).Mul((float)(1.0 / 3.0));
		out->push_back(p);
	}
	return Error::Nil();
}

Error WavefrontSceneReader::ParseMaterials(const std::string &name, const std::string &content) { // :651-761
	std::istringstream in(content);
	std::string line, matName;
	int lineNum = 0, cur = -1;
	while (std::getline(in, line)) {
		lineNum++;
		const std::vector<std::string> tok = fields(line);
		if (tok.empty() || tok[0][0] == '#') continue;
		const std::string &cmd = tok[0];
		if (cmd == "newmtl") {
			if (tok.size() != 2) return emitError(name, lineNum, fmt("unsupported syntax for \"newmtl\"; expected 1 argument; got %zu", tok.size() - 1));
			matName = tok[1];
			if (matNameToIndex.count(matName)) return emitError(name, lineNum, "material \"" + matName + "\" already defined");
			WavefrontMaterial m;
			m.Name = matName;
			m.AssetRelPath = name;
			materials.push_back(m);
			cur = int(materials.size()) - 1;
			matNameToIndex[matName] = cur;
			continue;
		}
		if (cur < 0) return emitError(name, lineNum, "got \"" + cmd + "\" without a \"newmtl\"");
		WavefrontMaterial &m = materials[cur];
		Error err;
		auto needArg = [&]() -> Error {
			if (tok.size() < 2) return emitError(name, lineNum, fmt("unsupported syntax for \"%s\"; expected 1 argument; got %zu", cmd.c_str(), tok.size() - 1));
			return Error::Nil();
		};
		if (cmd == "include") {
			if (Error e = needArg()) return e;
			const auto it = matNameToIndex.find(tok[1]);
			if (it == matNameToIndex.end()) return emitError(name, lineNum, "could not include unknown material \"" + tok[1] + "\"");
			m = WavefrontMaterial(materials[it->second]); // overwrite everything but the name
			m.Name = matName;
		} else if (cmd == "Kd" || cmd == "Ks" || cmd == "Ke" || cmd == "Tf") {
			Vec3 *dst = cmd == "Kd" ? &m.Kd : (cmd == "Ks" ? &m.Ks : (cmd == "Ke" ? &m.Ke : &m.Tf));
			Vec3 v;
			err = parseVec3(tok, &v);
			*dst = v;
		} else if (cmd == "Ni") {
			float v = 0;
			err = parseFloat32(tok, &v);
			m.Ni = v;
		} else if (cmd == "map_Kd" || cmd == "map_Ks" || cmd == "map_Ke" || cmd == "map_Tf" || cmd == "map_bump" || cmd == "map_normal") {
			if (Error e = needArg()) return e; // (the reference indexes tok[1] unchecked here)
			std::string *dst = cmd == "map_Kd" ? &m.KdTex : cmd == "map_Ks" ? &m.KsTex : cmd == "map_Ke" ? &m.KeTex : cmd == "map_Tf" ? &m.TfTex
			                   : cmd == "map_bump" ? &m.BumpTex : &m.NormalTex;
			*dst = tok[1];
		} else if (cmd == "mat_expr") {
			if (Error e = needArg()) return e;
			std::string joined;
			for (size_t i = 1; i < tok.size(); i++) joined += (i > 1 ? " " : "") + tok[i];
			m.MaterialExpression = joined;
		} else if (cmd == "KeScaler") {
			if (Error e = needArg()) return e;
			float v = 0;
			err = parseFloat32(tok, &v);
			m.KeScaler = v;
		}
		if (err) return emitError(name, lineNum, err.msg);
	}
	return Error::Nil();
}

void WavefrontSceneReader::CreateDefaultMeshInstances() { // :246-258
	for (size_t i = 0; i < rawScene.meshes.size(); i++) {
		Vec3 lo{3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f}, hi{-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
		for (const Primitive &p : rawScene.meshes[i].primitives) { lo = types::MinVec3(lo, p.bbox[0]); hi = types::MaxVec3(hi, p.bbox[1]); }
		MeshInstance inst;
		inst.meshIndex = uint32_t(i);
		inst.hasBounds = true;
		inst.bbox[0] = lo;
		inst.bbox[1] = hi;
		inst.center = lo.Add(hi).Mul(0.5f);
		rawScene.instances.push_back(inst);
	}
}

void WavefrontSceneReader::ProcessMaterials() { // :191-243
	std::map<int, int> wfToScene;
	std::vector<compiler::Material> pruned;
	for (size_t i = 0; i < materials.size(); i++) {
		WavefrontMaterial &wf = materials[i];
		if (wf.Name == compiler::SceneDiffuseMaterialName || wf.Name == compiler::SceneEmissiveMaterialName) wf.Used = true;
		compiler::Material m;
		m.name = wf.Name;
		m.expression = wf.GetExpression();
		m.assetRelPath = wf.AssetRelPath;
		m.used = wf.Used;
		if (!wf.Used) { pruned.push_back(m); continue; }
		rawScene.materials.push_back(m);
		wfToScene[int(i)] = int(rawScene.materials.size()) - 1;
	}
	for (compiler::Mesh &mesh : rawScene.meshes)
		for (Primitive &p : mesh.primitives) p.materialIndex = wfToScene[p.materialIndex];
	// unused materials go last: material expressions may still reference them by name
	rawScene.materials.insert(rawScene.materials.end(), pruned.begin(), pruned.end());
}

Error WavefrontSceneReader::finish(compiler::Output *out) {
	if (rawScene.instances.empty()) CreateDefaultMeshInstances();
	ProcessMaterials();
	return compiler::CompileScene(rawScene, out);
}

Error WavefrontSceneReader::Read(const std::string &path, compiler::Output *out) {
	if (Error e = parseFile(path, false)) return e;
	return finish(out);
}

Error WavefrontSceneReader::ReadString(const std::string &name, const std::string &content, compiler::Output *out) {
	if (Error e = Parse(name, content)) return e;
	return finish(out);
}

Error ReadScene(const std::string &filename, compiler::Output *out) { // reader.go:18-35
	auto ends = [&](const char *suffix) { const size_t n = strlen(suffix); return filename.size() >= n && !filename.compare(filename.size() - n, n, suffix); };
	if (ends(".obj")) {
		WavefrontSceneReader r;
		return r.Read(filename, out);
	}
	if (ends(".zip")) return Error{POLARIS_E_UNSUPPORTED, "readScene: compiled .zip scenes are Go gob streams and are not supported by this build"};
	return bad("readScene: unsupported file format");
}

} // namespace reader
} // namespace polaris
