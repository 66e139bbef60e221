// host_capi.cpp -- a small C API over the C++ host layer so the pytest suite can drive it
// (ctypes).  Scheduler entry points use mock tracers (speed + last-frame stats), exactly like
// tracer/scheduler_test.go's mockTracer; renderer entry points need a GPU.
#include <cstring>
#include <random>

#include "renderer.hpp"

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

} // extern "C"
