// layout_corrupt.cpp -- feeds corrupted scenes to build_layout + the CPU traversal of layout_check.cpp
// under ASan/UBSan (tests/test_scene_layout.py).  usage: layout_check_asan scene.bin <trials> <seed>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "polaris_types.h"

extern "C" int layout_check_traverse(const PolarisSceneView *sc, int max_leaf_tris, const float *rays, uint32_t n, int any_hit, int32_t *hit,
                                     uint64_t *counters, char *err, size_t err_len);

int main(int argc, char **argv) {
	if (argc < 4) return 2;
	FILE *f = fopen(argv[1], "rb");
	if (!f) return 2;
	int64_t hdr[12];
	if (fread(hdr, sizeof hdr, 1, f) != 1) return 2;
	const size_t elem[10] = {sizeof(PolarisBvhNode), sizeof(PolarisMeshInstance), sizeof(PolarisMaterialNode), sizeof(PolarisEmissive),
	                         sizeof(PolarisTextureMetadata), 1, 16, 16, 8, 4};
	std::vector<std::vector<uint8_t>> clean(10);
	for (int i = 0; i < 10; i++) {
		clean[i].resize((size_t)hdr[i] * elem[i]);
		if (!clean[i].empty() && fread(clean[i].data(), clean[i].size(), 1, f) != 1) return 2;
	}
	fclose(f);
	const int trials = atoi(argv[2]);
	std::mt19937 rng((unsigned)atoi(argv[3]));
	std::vector<float> rays(8 * 2048);
	std::uniform_real_distribution<float> U(-3.0f, 3.0f);
	for (size_t i = 0; i < rays.size() / 8; i++) {
		float *r = &rays[8 * i];
		r[0] = U(rng); r[1] = U(rng) + 1; r[2] = U(rng); r[3] = 3.0e38f;
		r[4] = U(rng); r[5] = U(rng); r[6] = U(rng); r[7] = 0;
	}
	std::vector<int32_t> hit(6 * rays.size() / 8);
	static const uint32_t hostile[] = {0xFFFFFFFFu, 0x7FFFFFFFu, 0x80000000u, 0x80000001u, 1u, 0u, 0x00FFFFFFu, 0x7F800000u, 0x7FC00000u, 0xFF800000u, 12345678u};
	int accepted = 0, rejected = 0;
	for (int t = 0; t < trials; t++) {
		std::vector<std::vector<uint8_t>> a = clean;
		const int edits = 1 + (int)(rng() % 4);
		for (int e = 0; e < edits; e++) {
			// structural arrays (nodes, instances, materials, emissives, texture metadata, material index) get most of the attention
			static const int pick[] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 9, 9, 6};
			const int arr = pick[rng() % (sizeof pick / sizeof pick[0])];
			if (a[arr].size() < 4) continue;
			const size_t word = (rng() % (a[arr].size() / 4)) * 4;
			uint32_t v = hostile[rng() % (sizeof hostile / sizeof hostile[0])];
			if (rng() % 3 == 0) v = (uint32_t)rng();
			memcpy(&a[arr][word], &v, 4);
		}
		PolarisSceneView sc{};
		sc.bvh_nodes = (const PolarisBvhNode *)a[0].data(); sc.num_bvh_nodes = (uint32_t)hdr[0];
		sc.mesh_instances = (const PolarisMeshInstance *)a[1].data(); sc.num_mesh_instances = (uint32_t)hdr[1];
		sc.material_nodes = (const PolarisMaterialNode *)a[2].data(); sc.num_material_nodes = (uint32_t)hdr[2];
		sc.emissives = (const PolarisEmissive *)a[3].data(); sc.num_emissives = (uint32_t)hdr[3];
		sc.texture_meta = (const PolarisTextureMetadata *)a[4].data(); sc.num_textures = (uint32_t)hdr[4];
		sc.texture_data = a[5].data(); sc.texture_data_bytes = (uint32_t)hdr[5];
		sc.vertices = (const float *)a[6].data(); sc.normals = (const float *)a[7].data(); sc.uvs = (const float *)a[8].data();
		sc.material_index = (const uint32_t *)a[9].data(); sc.num_triangles = (uint32_t)hdr[9];
		sc.scene_diffuse_mat_index = (int32_t)hdr[10]; sc.scene_emissive_mat_index = (int32_t)hdr[11];
		char err[256] = {0};
		uint64_t counters[8] = {0};
		const int rc = layout_check_traverse(&sc, (int)(rng() % 3) * 2, rays.data(), (uint32_t)(rays.size() / 8), (int)(rng() % 2), hit.data(), counters, err, sizeof err);
		if (rc) rejected++; else accepted++;
	}
	printf("accepted=%d rejected=%d\n", accepted, rejected);
	return 0;
}
