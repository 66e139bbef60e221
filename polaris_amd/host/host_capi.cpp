// host_capi.cpp -- a small C API over the C++ host layer so the pytest suite can drive it
// (ctypes).  Scheduler entry points use mock tracers (speed + last-frame stats), exactly like
// tracer/scheduler_test.go's mockTracer; renderer entry points need a GPU.
#include <cstring>
#include <random>

#include "renderer.hpp"
#include "scene_compiler.hpp"

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
	auto src = [box]() { return (uint32_t)box->rng(); };
	auto devs = tracer::hip::Devices({});
	std::vector<std::unique_ptr<tracer::Tracer>> trs;
	for (uint32_t i = 0; i < n_tracers; i++) {
		int di = device_indices[i];
		if (di < 0 || (size_t)di >= devs.size()) { snprintf(err, 256, "device %d not available", di); delete box; return nullptr; }
		auto t = std::make_unique<tracer::hip::HipTracer>("hip-" + std::to_string(i), devs[di], src);
		if (Error e = t->Init()) { snprintf(err, 256, "%s", e.msg.c_str()); delete box; return nullptr; }
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
int polaris_host_renderer_render(void *h, uint32_t accumulated, uint32_t *rows_out, double *frame_ms) {
	auto *box = static_cast<RendererBox *>(h);
	Error e = box->r->renderFrame(accumulated);
	if (e) { box->error = e.msg; return e.code; }
	const auto &rows = box->r->BlockAssignments();
	if (rows_out) for (size_t i = 0; i < rows.size(); i++) rows_out[i] = rows[i];
	if (frame_ms) *frame_ms = std::chrono::duration<double, std::milli>(box->r->Stats().RenderTime).count();
	return 0;
}
int polaris_host_renderer_read(void *h, uint8_t *rgba, size_t n_rgba, float *frame_acc, size_t n_floats) {
	auto *box = static_cast<RendererBox *>(h);
	auto *p = dynamic_cast<tracer::hip::HipTracer *>(box->r->Primary());
	if (!p) return POLARIS_E_UNSUPPORTED;
	if (rgba) if (Error e = p->ReadFrameBuffer(rgba, n_rgba)) { box->error = e.msg; return e.code; }
	if (frame_acc) if (Error e = p->ReadAccumulator(1, frame_acc, n_floats)) { box->error = e.msg; return e.code; }
	return 0;
}
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

} // extern "C"
