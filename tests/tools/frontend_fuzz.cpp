// frontend_fuzz.cpp -- robustness driver for the scene front-end (test tool, not product).
// Built with -fsanitize=address,undefined by tests/test_frontend_robustness.py and run over
// valid, truncated and corrupted files: every input must end in "ok" or a clean error, never in
// a crash, an out-of-bounds access or a hang.
//
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined \
//       -Iinclude -Ipolaris_amd/host frontend_fuzz.cpp polaris_amd/host/{texture,material_expr,wavefront_reader,scene_compiler}.cpp -lz -fopenmp
#include <cstdio>
#include <cstring>
#include <string>

#include "material_expr.hpp"
#include "texture.hpp"
#include "wavefront_reader.hpp"

using namespace polaris;

static bool ends_with(const std::string &s, const char *suffix) {
	const size_t n = strlen(suffix);
	return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}

int main(int argc, char **argv) {
	int ok = 0, failed = 0;
	for (int i = 1; i < argc; i++) {
		const std::string path = argv[i];
		Error e;
		if (ends_with(path, ".obj")) {
			compiler::Output out;
			e = reader::ReadScene(path, &out);
			if (!e) { // the compiled arrays must be self-consistent enough for the tracer's own validation
				const PolarisSceneView v = out.View();
				if (v.num_triangles * 3 * 4 != out.vertices.size() || v.num_bvh_nodes == 0) { fprintf(stderr, "%s: inconsistent output\n", path.c_str()); return 2; }
			}
		} else if (ends_with(path, ".expr")) {
			FILE *f = fopen(path.c_str(), "rb");
			std::string text;
			if (f) { char buf[4096]; size_t n; while ((n = fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, n); fclose(f); }
			std::unique_ptr<material::Expr> expr;
			e = material::ParseExpression(text, &expr);
			if (!e) e = expr->Validate();
		} else {
			texture::Texture t;
			e = texture::Load(path, &t);
			if (!e && t.data.size() != (size_t)t.width * t.height * (t.format == POLARIS_TEX_L8 ? 1 : t.format == POLARIS_TEX_RGBA32F ? 16 : 4)) {
				fprintf(stderr, "%s: texture size mismatch\n", path.c_str());
				return 2;
			}
		}
		if (e) failed++; else ok++;
	}
	printf("ok=%d rejected=%d\n", ok, failed);
	return 0;
}
