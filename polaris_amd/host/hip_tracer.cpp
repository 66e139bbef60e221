#include "hip_tracer.hpp"

#include <cstring>

namespace polaris {
namespace tracer {
namespace hip {

using clock_ = std::chrono::steady_clock;

std::vector<Device> Devices(const std::vector<std::string> &blacklist) { // renderer/default.go:204-224
	std::vector<Device> out;
	const int n = polaris_hip_device_count();
	for (int i = 0; i < n; i++) {
		char name[256] = {0};
		uint32_t cus = 0, mhz = 0;
		uint64_t mem = 0;
		if (polaris_hip_device_info(i, name, &cus, &mhz, &mem) != POLARIS_OK) continue;
		Device d{i, name, cus * mhz / 1000};
		bool skip = false;
		for (const auto &b : blacklist)
			if (!b.empty() && d.Name.find(b) != std::string::npos) skip = true;
		if (!skip) out.push_back(d);
	}
	return out;
}

HipTracer::HipTracer(std::string id, Device dev, SeedSource seeds) : id_(std::move(id)), dev_(std::move(dev)), seeds_(std::move(seeds)) {}
HipTracer::~HipTracer() { Close(); }

Error HipTracer::check(int rc) {
	if (rc == POLARIS_OK) return Error::Nil();
	const char *m = polaris_hip_last_error(h_);
	return Error{rc, std::string("hip tracer (") + dev_.Name + "): " + (m ? m : "")};
}

Error HipTracer::Init() {
	std::lock_guard<std::mutex> lk(mu_);
	return check(polaris_hip_create(dev_.Index, &h_));
}

void HipTracer::Close() {
	std::lock_guard<std::mutex> lk(mu_);
	if (h_) polaris_hip_destroy(h_);
	h_ = nullptr;
}

Error HipTracer::UpdateState(UpdateMode mode, ChangeType type, const void *data, Duration *took) {
	{
		std::lock_guard<std::mutex> lk(change_mu_);
		switch (type) {
		case ChangeType::FrameDimensions: dims_ = *static_cast<const FrameDims *>(data); has_dims_ = true; break;
		case ChangeType::SceneData: scene_ = static_cast<const PolarisSceneView *>(data); has_scene_ = true; break;
		case ChangeType::CameraData: cam_ = *static_cast<const CameraData *>(data); has_cam_ = true; break;
		default: return Error{POLARIS_E_BAD_ARGUMENT, "unsupported change type"};
		}
	}
	if (mode == UpdateMode::Synchronous) return commitChanges(took);
	if (took) *took = Duration{0};
	return Error::Nil();
}

Error HipTracer::commitChanges(Duration *took) { // tracer/opencl/tracer.go:161-192
	const auto start = clock_::now();
	std::lock_guard<std::mutex> lk(change_mu_);
	Error err;
	if (has_dims_ && !err) { err = check(polaris_hip_resize(h_, dims_.w, dims_.h)); has_dims_ = false; }
	if (has_scene_ && !err) { err = check(polaris_hip_upload_scene(h_, scene_)); has_scene_ = false; scene_ = nullptr; }
	if (has_cam_ && !err) { err = check(polaris_hip_set_camera(h_, cam_.eye, cam_.frustum)); has_cam_ = false; }
	stats_.UpdateTime = std::chrono::duration_cast<Duration>(clock_::now() - start);
	if (took) *took = stats_.UpdateTime;
	return err;
}

Error HipTracer::Trace(BlockRequest *req, Duration *took) { // tracer/opencl/tracer.go:194-247
	const auto start = clock_::now();
	if (Error e = commitChanges(nullptr)) return e;
	const size_t stride = 1 + req->num_bounces;
	std::vector<uint32_t> seeds((size_t)req->samples_per_pixel * stride);
	for (uint32_t s = 0; s < req->samples_per_pixel; s++) { // same draw order as the reference
		req->seed = seeds_();
		seeds[s * stride] = req->seed;
		for (uint32_t b = 0; b < req->num_bounces; b++) seeds[s * stride + 1 + b] = seeds_();
	}
	if (Error e = check(polaris_hip_trace(h_, req, seeds.data(), seeds.size(), &last_))) return e;
	req->accumulated_samples += req->samples_per_pixel; // tracer.go:240
	stats_.BlockW = req->block_w;
	stats_.BlockH = req->block_h;
	stats_.RenderTime = std::chrono::duration_cast<Duration>(clock_::now() - start);
	if (took) *took = stats_.RenderTime;
	return Error::Nil();
}

Error HipTracer::MergeOutput(Tracer *other, BlockRequest *req, Duration *took) { // tracer.go:279-286
	const auto start = clock_::now();
	auto *src = dynamic_cast<HipTracer *>(other);
	if (!src) return Error{POLARIS_E_UNSUPPORTED, "merge failed: unsupported tracer instance"};
	Error e = check(polaris_hip_merge(h_, src->h_, req));
	if (took) *took = std::chrono::duration_cast<Duration>(clock_::now() - start);
	return e;
}

Error HipTracer::SyncFramebuffer(BlockRequest *req, Duration *took) { // tracer.go:250-276
	const auto start = clock_::now();
	Error e = check(polaris_hip_sync_framebuffer(h_, req));
	if (took) *took = std::chrono::duration_cast<Duration>(clock_::now() - start);
	return e;
}

Error HipTracer::ReadFrameBuffer(uint8_t *rgba, size_t n) { return check(polaris_hip_read_framebuffer(h_, rgba, n)); }
Error HipTracer::ReadAccumulator(int which, float *out, size_t n) { return check(polaris_hip_read_accumulator(h_, which, out, n)); }

} // namespace hip
} // namespace tracer
} // namespace polaris
