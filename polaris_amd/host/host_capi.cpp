// host_capi.cpp -- a small C API over the C++ host layer so the pytest suite can drive it
// (ctypes).  Scheduler entry points use mock tracers (speed + last-frame stats), exactly like
// tracer/scheduler_test.go's mockTracer; renderer entry points need a GPU.
#include <cstring>
#include <deque>
#include <mutex>
#include <random>

#include "material_expr.hpp"
#include "renderer.hpp"
#include "scene_compiler.hpp"
#include "texture.hpp"
#include "wavefront_reader.hpp"

using namespace polaris;

namespace {
class MockTracer : public tracer::Tracer { // tracer/scheduler_test.go:82-123
public:
	uint32_t speed = 1;
	tracer::Stats stats;
	std::string Id() const override { return "mock"; }
	uint8_t Flags() const override { return tracer::Local; }
	uint32_t Speed() const override { return speed; }
	Error Init() override { return {}; }
	void Close() override {}
	tracer::Stats *GetStats() override { return &stats; }
	Error UpdateState(tracer::UpdateMode, tracer::ChangeType, const void *, tracer::Duration *) override { return {}; }
	Error Trace(tracer::BlockRequest *, tracer::Duration *) override { return {}; }
	Error MergeOutput(tracer::Tracer *, tracer::BlockRequest *, tracer::Duration *) override { return {}; }
	Error SyncFramebuffer(tracer::BlockRequest *, tracer::Duration *) override { return {}; }
};

struct SchedulerBox {
	std::unique_ptr<tracer::BlockScheduler> sch;
	std::vector<MockTracer> mocks;
};

struct RendererBox {
	std::unique_ptr<renderer::DefaultRenderer> r;
	std::string error;
	std::mt19937 rng;
	std::mutex rng_mu; // the workers draw concurrently, like the goroutines on Go's global math/rand (tracer.go:222)
	// injected seed lists (tests): tracer i draws from lists[i] while it has entries, then from the shared generator
	std::vector<std::deque<uint32_t>> lists;
	std::vector<tracer::hip::HipTracer *> hips; // owned by the renderer
};
} // namespace

extern "C" {

// kind: 0 naive, 1 perfect
void *polaris_host_scheduler_new(int kind, const uint32_t *speeds, uint32_t n) {
	auto *b = new SchedulerBox();
	b->sch = kind == 0 ? tracer::NaiveScheduler() : tracer::PerfectScheduler();
	b->mocks.resize(n);
	for (uint32_t i = 0; i < n; i++) b->mocks[i].speed = speeds[i];
	return b;
}
// block_h / render_ns: last-frame stats of every tracer (ignored by the naive scheduler)
void polaris_host_scheduler_schedule(void *h, const uint32_t *block_h, const int64_t *render_ns, uint32_t frame_h, uint32_t *out) {
	auto *b = static_cast<SchedulerBox *>(h);
	std::vector<tracer::Tracer *> raw;
	for (size_t i = 0; i < b->mocks.size(); i++) {
		if (block_h) b->mocks[i].stats.BlockH = block_h[i];
		if (render_ns) b->mocks[i].stats.RenderTime = tracer::Duration(render_ns[i]);
		raw.push_back(&b->mocks[i]);
	}
	auto rows = b->sch->Schedule(raw, frame_h);
	for (size_t i = 0; i < rows.size(); i++) out[i] = rows[i];
}
void polaris_host_scheduler_free(void *h) { delete static_cast<SchedulerBox *>(h); }

// Renderer over n_tracers HipTracers; device_indices may repeat a device (several tracers on one
// GPU) which is how the multi-tracer frame loop is exercised on a 1-GPU box.
void *polaris_host_renderer_new(const int *device_indices, uint32_t n_tracers, uint32_t primary, int scheduler_kind,
                                const PolarisSceneView *scene, const float eye[3], const float frustum[16], uint32_t w, uint32_t h,
                                uint32_t spp, uint32_t bounces, uint32_t min_rr, float exposure, uint32_t seed, char err[256]) {
	auto *box = new RendererBox();
	box->rng.seed(seed);
	box->lists.resize(n_tracers);
	auto src = [box]() {
		std::lock_guard<std::mutex> lk(box->rng_mu);
		return (uint32_t)box->rng();
	};
	auto devs = tracer::hip::Devices({});
	std::vector<std::unique_ptr<tracer::Tracer>> trs;
	for (uint32_t i = 0; i < n_tracers; i++) {
		int di = device_indices[i];
		if (di < 0 || (size_t)di >= devs.size()) { snprintf(err, 256, "device %d not available", di); delete box; return nullptr; }
		auto tracer_src = [box, i]() { // this tracer's injected list first (only its own worker thread pops it)
			std::lock_guard<std::mutex> lk(box->rng_mu);
			if (!box->lists[i].empty()) {
				const uint32_t v = box->lists[i].front();
				box->lists[i].pop_front();
				return v;
			}
			return (uint32_t)box->rng();
		};
		auto t = std::make_unique<tracer::hip::HipTracer>("hip-" + std::to_string(i), devs[di], tracer_src);
		if (Error e = t->Init()) { snprintf(err, 256, "%s", e.msg.c_str()); delete box; return nullptr; }
		box->hips.push_back(t.get());
		trs.push_back(std::move(t));
	}
	renderer::Options o;
	o.FrameW = w; o.FrameH = h; o.SamplesPerPixel = spp; o.NumBounces = bounces; o.MinBouncesForRR = min_rr; o.Exposure = exposure;
	box->r = std::make_unique<renderer::DefaultRenderer>(std::move(trs), primary, scheduler_kind == 0 ? tracer::NaiveScheduler() : tracer::PerfectScheduler(),
	                                                     o, src);
	tracer::FrameDims dims{w, h};
	tracer::CameraData cam;
	memcpy(cam.eye, eye, sizeof cam.eye);
	memcpy(cam.frustum, frustum, sizeof cam.frustum);
	Error e = box->r->UpdateAll(tracer::ChangeType::FrameDimensions, &dims);
	if (!e) e = box->r->UpdateAll(tracer::ChangeType::SceneData, scene);
	if (!e) e = box->r->UpdateAll(tracer::ChangeType::CameraData, &cam);
	if (e) { snprintf(err, 256, "%s", e.msg.c_str()); delete box; return nullptr; }
	return box;
}
// Test hooks: the host PRNG draws of tracer `tracer_index` (tracer.go:222, pipeline.go:146) come from this list until it is
// used up -- what lets a frame rendered by several tracers be compared with the oracle block by block; and a tracer option
// (polaris_hip_set_option) applied to every tracer of the renderer.
int polaris_host_renderer_push_seeds(void *h, uint32_t tracer_index, const uint32_t *seeds, size_t n) {
	auto *box = static_cast<RendererBox *>(h);
	if (tracer_index >= box->lists.size()) return POLARIS_E_BAD_ARGUMENT;
	std::lock_guard<std::mutex> lk(box->rng_mu);
	box->lists[tracer_index].insert(box->lists[tracer_index].end(), seeds, seeds + n);
	return 0;
}
int polaris_host_renderer_set_option(void *h, const char *key, int64_t value) {
	auto *box = static_cast<RendererBox *>(h);
	for (auto *t : box->hips)
		if (int rc = polaris_hip_set_option(t->Handle(), key, value)) { box->error = polaris_hip_last_error(t->Handle()); return rc; }
	return 0;
}
int polaris_host_renderer_render(void *h, uint32_t accumulated, uint32_t *rows_out, double *frame_ms) {
	auto *box = static_cast<RendererBox *>(h);
	Error e = box->r->renderFrame(accumulated);
	if (e) { box->error = e.msg; return e.code; }
	const auto &rows = box->r->BlockAssignments();
	if (rows_out) for (size_t i = 0; i < rows.size(); i++) rows_out[i] = rows[i];
	if (frame_ms) *frame_ms = std::chrono::duration<double, std::milli>(box->r->Stats().RenderTime).count();
	return 0;
}
// counters of tracer `tracer_index`'s last Trace (PolarisTraceStats) and the wall time of that Trace in milliseconds
int polaris_host_renderer_tracer_stats(void *h, uint32_t tracer_index, PolarisTraceStats *out, double *trace_ms) {
	auto *box = static_cast<RendererBox *>(h);
	if (tracer_index >= box->hips.size()) return POLARIS_E_BAD_ARGUMENT;
	if (out) *out = box->hips[tracer_index]->LastTraceStats();
	if (trace_ms) *trace_ms = std::chrono::duration<double, std::milli>(box->hips[tracer_index]->GetStats()->RenderTime).count();
	return 0;
}
// which branch the merges onto the PRIMARY took so far (polaris_hip_merge_counts: local / peer access / staged copy ...)
int polaris_host_renderer_merge_counts(void *h, uint64_t counts[POLARIS_MERGE_BRANCHES]) {
	auto *box = static_cast<RendererBox *>(h);
	auto *p = dynamic_cast<tracer::hip::HipTracer *>(box->r->Primary());
	if (!p) return POLARIS_E_UNSUPPORTED;
	return polaris_hip_merge_counts(p->Handle(), counts);
}
int polaris_host_renderer_read(void *h, uint8_t *rgba, size_t n_rgba, float *frame_acc, size_t n_floats) {
	auto *box = static_cast<RendererBox *>(h);
	auto *p = dynamic_cast<tracer::hip::HipTracer *>(box->r->Primary());
	if (!p) return POLARIS_E_UNSUPPORTED;
	if (rgba) if (Error e = p->ReadFrameBuffer(rgba, n_rgba)) { box->error = e.msg; return e.code; }
	if (frame_acc) if (Error e = p->ReadAccumulator(1, frame_acc, n_floats)) { box->error = e.msg; return e.code; }
	return 0;
}
int polaris_host_renderer_save(void *h, const char *path) {
	auto *box = static_cast<RendererBox *>(h);
	if (Error e = box->r->SaveFrameBuffer(path)) { box->error = e.msg; return e.code; }
	return 0;
}
int polaris_host_write_png(const char *path, const uint8_t *rgba, uint32_t w, uint32_t h) { return renderer::WritePNG(path, rgba, w, h).code; }
const char *polaris_host_renderer_error(void *h) { return static_cast<RendererBox *>(h)->error.c_str(); }
void polaris_host_renderer_free(void *h) {
	auto *box = static_cast<RendererBox *>(h);
	if (box) { box->r->Close(); delete box; }
}


// ---- scene compiler (polaris_amd/host/scene_compiler.cpp) ------------------------------------
// bvh.Build over boxes [n][6] = (min.xyz, max.xyz); centers = box centers.  Returns the node
// count; out_nodes (capacity cap) receives the nodes, leaf_sizes the item count of every leaf
// callback in call order.
uint32_t polaris_host_bvh_build(const float *boxes, uint32_t n, int min_leaf, PolarisBvhNode *out_nodes, uint32_t cap,
                                uint32_t *leaf_sizes, uint32_t *n_leaves) {
	std::vector<compiler::bvh::BoundedVolume> vols(n);
	for (uint32_t i = 0; i < n; i++) {
		const float *b = boxes + 6 * i;
		vols[i] = {{{b[0], b[1], b[2]}, {b[3], b[4], b[5]}}, {0.5f * (b[0] + b[3]), 0.5f * (b[1] + b[4]), 0.5f * (b[2] + b[5])}};
	}
	uint32_t leaves = 0;
	auto nodes = compiler::bvh::Build(vols, min_leaf, [&](PolarisBvhNode *leaf, const std::vector<uint32_t> &items) {
		leaf->ldata = -int32_t(items[0]);
		leaf->rdata = int32_t(items.size());
		if (leaf_sizes) leaf_sizes[leaves] = uint32_t(items.size());
		leaves++;
	});
	if (n_leaves) *n_leaves = leaves;
	for (uint32_t i = 0; i < nodes.size() && i < cap; i++) out_nodes[i] = nodes[i];
	return uint32_t(nodes.size());
}

struct CompiledBox { compiler::Output out; PolarisSceneView view; };

void *polaris_host_compile_scene(const float *prim_vertices, const float *prim_normals, const float *prim_uvs, const int32_t *prim_material,
                                 const uint32_t *mesh_prim_offsets, uint32_t n_meshes, const uint32_t *inst_mesh,
                                 const float *inst_transforms, uint32_t n_instances, const PolarisMaterialNode *nodes, uint32_t n_nodes,
                                 const int32_t *material_roots, uint32_t n_materials, const PolarisTextureMetadata *tex_meta, uint32_t n_tex,
                                 const uint8_t *tex_data, uint32_t n_tex_bytes, int32_t scene_diffuse, int32_t scene_emissive, int min_leaf,
                                 char err[256]) {
	compiler::Input in;
	in.meshes.resize(n_meshes);
	for (uint32_t m = 0; m < n_meshes; m++)
		for (uint32_t p = mesh_prim_offsets[m]; p < mesh_prim_offsets[m + 1]; p++) {
			compiler::Primitive pr{};
			for (int k = 0; k < 3; k++) {
				const float *v = prim_vertices + (size_t)p * 9 + 3 * k, *nn = prim_normals + (size_t)p * 9 + 3 * k;
				pr.vertices[k] = {v[0], v[1], v[2]};
				pr.normals[k] = {nn[0], nn[1], nn[2]};
				pr.uvs[k][0] = prim_uvs[(size_t)p * 6 + 2 * k];
				pr.uvs[k][1] = prim_uvs[(size_t)p * 6 + 2 * k + 1];
			}
			pr.materialIndex = prim_material[p];
			in.meshes[m].primitives.push_back(pr);
		}
	in.instances.resize(n_instances);
	for (uint32_t i = 0; i < n_instances; i++) {
		in.instances[i].meshIndex = inst_mesh[i];
		memcpy(in.instances[i].transform, inst_transforms + 16 * (size_t)i, 64);
	}
	in.materialNodes.assign(nodes, nodes + n_nodes);
	in.materialRoots.assign(material_roots, material_roots + n_materials);
	if (n_tex) in.textureMeta.assign(tex_meta, tex_meta + n_tex);
	if (n_tex_bytes) in.textureData.assign(tex_data, tex_data + n_tex_bytes);
	in.sceneDiffuseMatIndex = scene_diffuse;
	in.sceneEmissiveMatIndex = scene_emissive;
	in.minPrimitivesPerLeaf = min_leaf;
	auto *box = new CompiledBox();
	if (Error e = compiler::Compile(in, &box->out)) {
		if (err) snprintf(err, 256, "%s", e.msg.c_str());
		delete box;
		return nullptr;
	}
	box->view = box->out.View();
	return box;
}
const PolarisSceneView *polaris_host_compiled_view(void *h) { return &static_cast<CompiledBox *>(h)->view; }
void polaris_host_compiled_free(void *h) { delete static_cast<CompiledBox *>(h); }

// ---- scene front-end (wavefront_reader.cpp, material_expr.cpp, texture.cpp) ---------------------
// Returns 0 = parsed and valid, 1 = parse error, 2 = parsed but semantically invalid.
int polaris_host_material_check(const char *expr, char err[512]) {
	std::unique_ptr<material::Expr> e;
	if (Error pe = material::ParseExpression(expr, &e)) { if (err) snprintf(err, 512, "%s", pe.msg.c_str()); return 1; }
	if (Error ve = e->Validate()) { if (err) snprintf(err, 512, "%s", ve.msg.c_str()); return 2; }
	return 0;
}
int polaris_host_material_ior(const char *name, float *out) { return material::IOR(name, out) ? 1 : 0; }

// ReadScene (path != NULL) or the same from an in-memory .obj (content != NULL, name = its label).
// The handle is a CompiledBox: polaris_host_compiled_view/_free apply.
void *polaris_host_read_scene(const char *path, const char *name, const char *content, int min_leaf, char err[1024]) {
	auto *box = new CompiledBox();
	reader::WavefrontSceneReader r;
	if (min_leaf > 0) r.rawScene.minPrimitivesPerLeaf = min_leaf;
	Error e;
	if (path) {
		const std::string p(path);
		if (p.size() >= 4 && p.compare(p.size() - 4, 4, ".obj") == 0) e = r.Read(p, &box->out);
		else e = reader::ReadScene(p, &box->out);
	} else {
		e = r.ReadString(name ? name : "embedded", content ? content : "", &box->out);
	}
	if (e) {
		if (err) snprintf(err, 1024, "%s", e.msg.c_str());
		delete box;
		return nullptr;
	}
	box->view = box->out.View();
	return box;
}
// Camera of a compiled scene: fov/eye/look/up as parsed, and the CameraData (eye + frustum corner
// rays) for a frame of the given aspect (cmd/render.go:58 SetupProjection).
void polaris_host_compiled_camera(void *h, float aspect, int invert_y, float params[10], float eye[3], float frustum[16]) {
	scene::Camera cam = static_cast<CompiledBox *>(h)->out.camera;
	if (params) {
		params[0] = cam.FOV;
		params[1] = cam.Position.x; params[2] = cam.Position.y; params[3] = cam.Position.z;
		params[4] = cam.LookAt.x; params[5] = cam.LookAt.y; params[6] = cam.LookAt.z;
		params[7] = cam.Up.x; params[8] = cam.Up.y; params[9] = cam.Up.z;
	}
	cam.InvertY = invert_y != 0;
	cam.SetupProjection(aspect);
	const tracer::CameraData d = cam.Data();
	if (eye) memcpy(eye, d.eye, sizeof d.eye);
	if (frustum) memcpy(frustum, d.frustum, sizeof d.frustum);
}
// warnings joined by '\n' (missing textures, ...); returns the length needed
size_t polaris_host_compiled_warnings(void *h, char *buf, size_t cap) {
	std::string all;
	for (const std::string &w : static_cast<CompiledBox *>(h)->out.warnings) all += w + "\n";
	if (buf && cap) snprintf(buf, cap, "%s", all.c_str());
	return all.size() + 1;
}
// Parse-level probes for the reader tests: counts and the instance matrices of a parsed .obj.
// out_counts = {meshes, instances, materials(after pruning order), primitives of mesh 0}
int polaris_host_parse_obj(const char *content, uint32_t out_counts[4], float *inst_transforms, float *inst_boxes, uint32_t cap,
                           char *mat_exprs, size_t mat_cap, char err[1024]) {
	reader::WavefrontSceneReader r;
	if (Error e = r.Parse("embedded", content)) { if (err) snprintf(err, 1024, "%s", e.msg.c_str()); return 1; }
	if (r.rawScene.instances.empty()) r.CreateDefaultMeshInstances();
	r.ProcessMaterials();
	out_counts[0] = uint32_t(r.rawScene.meshes.size());
	out_counts[1] = uint32_t(r.rawScene.instances.size());
	out_counts[2] = uint32_t(r.rawScene.materials.size());
	out_counts[3] = r.rawScene.meshes.empty() ? 0 : uint32_t(r.rawScene.meshes[0].primitives.size());
	for (uint32_t i = 0; i < r.rawScene.instances.size() && i < cap; i++) {
		const compiler::MeshInstance &mi = r.rawScene.instances[i];
		if (inst_transforms) memcpy(inst_transforms + 16 * i, mi.transform, 64);
		if (inst_boxes) {
			float *b = inst_boxes + 9 * i;
			b[0] = mi.bbox[0].x; b[1] = mi.bbox[0].y; b[2] = mi.bbox[0].z; b[3] = mi.bbox[1].x; b[4] = mi.bbox[1].y; b[5] = mi.bbox[1].z;
			b[6] = mi.center.x; b[7] = mi.center.y; b[8] = mi.center.z;
		}
	}
	if (mat_exprs && mat_cap) {
		std::string all;
		for (const compiler::Material &m : r.rawScene.materials) all += m.name + "\t" + (m.used ? "1" : "0") + "\t" + m.expression + "\n";
		snprintf(mat_exprs, mat_cap, "%s", all.c_str());
	}
	return 0;
}
int polaris_host_parse_mtl(const char *content, char *mat_exprs, size_t mat_cap, char err[1024]) {
	reader::WavefrontSceneReader r;
	if (Error e = r.ParseMaterials("embedded", content)) { if (err) snprintf(err, 1024, "%s", e.msg.c_str()); return 1; }
	std::string all;
	for (const reader::WavefrontMaterial &m : r.materials) all += m.Name + "\t" + m.GetExpression() + "\n";
	if (mat_exprs && mat_cap) snprintf(mat_exprs, mat_cap, "%s", all.c_str());
	return 0;
}
int polaris_host_select_face_index(const char *token, int list_len, int rel_offset, int *out, char err[256]) {
	if (Error e = reader::SelectFaceCoordIndex(token, list_len, rel_offset, out)) { if (err) snprintf(err, 256, "%s", e.msg.c_str()); return 1; }
	return 0;
}
// Decode an image file into the tracer's texture representation.  Returns 0 and fills
// meta = {format, width, height, bytes}; data (capacity cap) receives the texels.
int polaris_host_texture_load(const char *path, uint32_t meta[4], uint8_t *data, size_t cap, char err[512]) {
	texture::Texture t;
	if (Error e = texture::Load(path, &t)) { if (err) snprintf(err, 512, "%s", e.msg.c_str()); return 1; }
	meta[0] = t.format; meta[1] = t.width; meta[2] = t.height; meta[3] = uint32_t(t.data.size());
	if (data && cap >= t.data.size()) memcpy(data, t.data.data(), t.data.size());
	return 0;
}

} // extern "C"
