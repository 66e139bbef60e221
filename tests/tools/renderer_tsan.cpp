// tests/tools/renderer_tsan.cpp -- the host frame loop (polaris_amd/host/renderer.cpp + hip_tracer.cpp + scheduler.cpp) under the
// thread sanitizer, over the mock C ABI of tests/tools/mock_polaris_hip.cpp (the reference's frame loop:
// renderer/default.go:106-196 -- a worker per tracer, primary.MergeOutput(tracer) from the workers, SyncFramebuffer on the main thread).
//
//     renderer_tsan <frames> <tracers> <rows>
//
// For both block schedulers: `frames` frames with accumulated_samples == 0 (every frame's primary Trace clears the frame accumulator
// while the other workers race to merge: the reset-epoch protocol), then a progressive run (accumulated > 0: no clear, the sums grow),
// asynchronous camera updates queued from ANOTHER thread while frames render (opengl.go:298-300 does that from the GLFW thread), and a
// renderer torn down right after its last frame.  After every frame every row of the primary's frame accumulator must hold exactly the
// number of frames accumulated.  Exit code 0 = all frames right; the sanitizer itself fails the process (TSAN_OPTIONS=exitcode=66) on
// a report.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>

#include "renderer.hpp"

using namespace polaris;

extern "C" void mock_polaris_set_max_sleep_us(int us);

// A tracer that is NOT a HipTracer (the reference's mockTracer, tracer/scheduler_test.go:82-123, with a frame to merge into): the
// renderer then cannot use reset epochs and orders the workers' merges behind the RETURN of the primary's Trace (renderer.cpp, the
// frameCv_ branch).  Same convention as the mock ABI: one float per row, a Trace writes 1.0 into its block's rows.
class PlainTracer : public tracer::Tracer {
public:
	PlainTracer(std::string id, uint32_t seed) : id_(std::move(id)), rng_(seed) {}
	std::string Id() const override { return id_; }
	uint8_t Flags() const override { return tracer::Local; }
	uint32_t Speed() const override { return 10; }
	Error Init() override { return Error::Nil(); }
	void Close() override {}
	tracer::Stats *GetStats() override { return &stats_; }
	Error UpdateState(tracer::UpdateMode, tracer::ChangeType type, const void *data, tracer::Duration *) override {
		std::lock_guard<std::mutex> lk(mu_);
		if (type == tracer::ChangeType::FrameDimensions) { rows_ = static_cast<const tracer::FrameDims *>(data)->h; trace_.assign(rows_, 0.0f); std::lock_guard<std::mutex> lkm(merge_mu_); frame_.assign(rows_, 0.0f); }
		return Error::Nil();
	}
	Error Trace(tracer::BlockRequest *r, tracer::Duration *) override {
		const auto start = std::chrono::steady_clock::now();
		std::lock_guard<std::mutex> lk(mu_);
		if (r->accumulated_samples == 0) { std::lock_guard<std::mutex> lkm(merge_mu_); std::fill(frame_.begin(), frame_.end(), 0.0f); }
		std::fill(trace_.begin(), trace_.end(), 0.0f);
		std::this_thread::sleep_for(std::chrono::microseconds(rng_() % 300u));
		for (uint32_t y = r->block_y; y < r->block_y + r->block_h; y++) trace_[y] = 1.0f;
		r->accumulated_samples += r->samples_per_pixel;
		stats_.BlockW = r->block_w; stats_.BlockH = r->block_h;
		stats_.RenderTime = std::chrono::duration_cast<tracer::Duration>(std::chrono::steady_clock::now() - start) + tracer::Duration(1);
		return Error::Nil();
	}
	Error MergeOutput(tracer::Tracer *other, tracer::BlockRequest *r, tracer::Duration *) override {
		auto *src = dynamic_cast<PlainTracer *>(other);
		if (!src) return Error{POLARIS_E_UNSUPPORTED, "merge failed: unsupported tracer instance"};
		std::vector<float> rows;
		{ std::lock_guard<std::mutex> lk(src->mu_); rows.assign(src->trace_.begin() + r->block_y, src->trace_.begin() + r->block_y + r->block_h); }
		std::lock_guard<std::mutex> lkm(merge_mu_);
		for (uint32_t i = 0; i < r->block_h; i++) frame_[r->block_y + i] += rows[i];
		return Error::Nil();
	}
	Error SyncFramebuffer(tracer::BlockRequest *, tracer::Duration *) override { std::lock_guard<std::mutex> lk(mu_); std::lock_guard<std::mutex> lkm(merge_mu_); return Error::Nil(); }
	std::vector<float> Frame() { std::lock_guard<std::mutex> lkm(merge_mu_); return frame_; }

private:
	std::string id_;
	std::mt19937 rng_;
	std::mutex mu_, merge_mu_;
	uint32_t rows_ = 0;
	std::vector<float> trace_, frame_;
	tracer::Stats stats_;
};

static int run_plain(int scheduler_kind, uint32_t frames, uint32_t n_tracers, uint32_t rows) {
	std::mt19937 rng(11u);
	std::mutex rng_mu;
	auto src = [&]() { std::lock_guard<std::mutex> lk(rng_mu); return (uint32_t)rng(); };
	std::vector<std::unique_ptr<tracer::Tracer>> trs;
	for (uint32_t i = 0; i < n_tracers; i++) trs.push_back(std::make_unique<PlainTracer>("plain-" + std::to_string(i), 500u + i));
	renderer::Options o;
	o.FrameW = 16; o.FrameH = rows; o.SamplesPerPixel = 2; o.NumBounces = 3; o.MinBouncesForRR = 2;
	renderer::DefaultRenderer r(std::move(trs), 0, scheduler_kind == 0 ? tracer::NaiveScheduler() : tracer::PerfectScheduler(), o, src);
	tracer::FrameDims dims{o.FrameW, o.FrameH};
	if (Error e = r.UpdateAll(tracer::ChangeType::FrameDimensions, &dims)) return 1;
	auto *prim = dynamic_cast<PlainTracer *>(r.Primary());
	for (uint32_t f = 0; f < frames; f++) {
		if (Error e = r.renderFrame(0)) { fprintf(stderr, "plain frame %u: %s\n", f, e.msg.c_str()); return 1; }
		const std::vector<float> fr = prim->Frame();
		for (uint32_t y = 0; y < rows; y++)
			if (fr[y] != 1.0f) { fprintf(stderr, "plain tracers, scheduler %d frame %u: row %u holds %g, expected 1\n", scheduler_kind, f, y, (double)fr[y]); return 1; }
	}
	return 0;
}

static int run(int scheduler_kind, uint32_t frames, uint32_t n_tracers, uint32_t rows) {
	std::mt19937 rng(7u + (unsigned)scheduler_kind);
	std::mutex rng_mu;
	auto src = [&]() { std::lock_guard<std::mutex> lk(rng_mu); return (uint32_t)rng(); }; // the workers draw concurrently, like Go's global math/rand
	auto devs = tracer::hip::Devices({});
	std::vector<std::unique_ptr<tracer::Tracer>> trs;
	for (uint32_t i = 0; i < n_tracers; i++) {
		auto t = std::make_unique<tracer::hip::HipTracer>("mock-" + std::to_string(i), devs[i % devs.size()], src);
		if (Error e = t->Init()) { fprintf(stderr, "init: %s\n", e.msg.c_str()); return 1; }
		trs.push_back(std::move(t));
	}
	renderer::Options o;
	o.FrameW = 16; o.FrameH = rows; o.SamplesPerPixel = 2; o.NumBounces = 3; o.MinBouncesForRR = 2;
	renderer::DefaultRenderer r(std::move(trs), 0, scheduler_kind == 0 ? tracer::NaiveScheduler() : tracer::PerfectScheduler(), o, src);
	tracer::FrameDims dims{o.FrameW, o.FrameH};
	PolarisSceneView scene{};
	tracer::CameraData cam{};
	if (Error e = r.UpdateAll(tracer::ChangeType::FrameDimensions, &dims)) { fprintf(stderr, "%s\n", e.msg.c_str()); return 1; }
	if (Error e = r.UpdateAll(tracer::ChangeType::SceneData, &scene)) { fprintf(stderr, "%s\n", e.msg.c_str()); return 1; }
	if (Error e = r.UpdateAll(tracer::ChangeType::CameraData, &cam)) { fprintf(stderr, "%s\n", e.msg.c_str()); return 1; }
	auto *prim = dynamic_cast<tracer::hip::HipTracer *>(r.Primary());
	std::vector<float> acc(rows);
	auto check = [&](float want, uint32_t f, const char *what) {
		if (Error e = prim->ReadAccumulator(1, acc.data(), acc.size())) { fprintf(stderr, "read: %s\n", e.msg.c_str()); return false; }
		for (uint32_t y = 0; y < rows; y++)
			if (acc[y] != want) {
				fprintf(stderr, "%s scheduler %d frame %u: row %u of the frame accumulator holds %g, expected %g (rows ", what, scheduler_kind, f, y, (double)acc[y], (double)want);
				for (uint32_t h : r.BlockAssignments()) fprintf(stderr, "%u ", h);
				fprintf(stderr, ")\n");
				return false;
			}
		uint64_t sum = 0;
		for (uint32_t h : r.BlockAssignments()) sum += h;
		if (sum != rows) { fprintf(stderr, "%s: block assignments add up to %llu, not %u\n", what, (unsigned long long)sum, rows); return false; }
		return true;
	};
	// the interactive renderer's other thread: camera updates queued ASYNCHRONOUSLY on the primary while frames render
	// (renderer/opengl.go:298-300 sends them from the GLFW thread; tracer.go:150-158,198 applies them at the next Trace)
	std::atomic<bool> stop{false};
	std::thread ui([&] {
		tracer::CameraData c{};
		for (uint32_t i = 0; !stop.load(); i++) {
			c.eye[0] = (float)i;
			(void)r.Primary()->UpdateState(tracer::UpdateMode::Asynchronous, tracer::ChangeType::CameraData, &c);
			std::this_thread::sleep_for(std::chrono::microseconds(40));
		}
	});
	struct Join { std::atomic<bool> &stop; std::thread &t; ~Join() { stop = true; t.join(); } } join{stop, ui};
	for (uint32_t f = 0; f < frames; f++) { // every frame resets: the merges of fast workers race the primary's Reset stage
		if (f % 50 == 17) mock_polaris_set_max_sleep_us(0);   // now and then no sleeps at all: the tightest interleaving
		if (f % 50 == 23) mock_polaris_set_max_sleep_us(300);
		if (Error e = r.renderFrame(0)) { fprintf(stderr, "frame %u: %s\n", f, e.msg.c_str()); return 1; }
		if (!check(1.0f, f, "reset")) return 1;
	}
	for (uint32_t f = 1; f <= 20; f++) { // progressive: nothing clears, every frame adds one (spp 2 per frame)
		if (Error e = r.renderFrame(2 * f)) { fprintf(stderr, "progressive frame %u: %s\n", f, e.msg.c_str()); return 1; }
		if (!check(1.0f + (float)f, f, "progressive")) return 1;
	}
	return 0; // the renderer goes out of scope right behind its last frame: Close() joins the workers, the workers close their tracers
}

int main(int argc, char **argv) {
	if (argc > 1 && std::string(argv[1]) == "--canary") { // is the sanitizer alive?  two threads, one unsynchronised counter: it must report (exit code 66)
		static int racy = 0;
		std::thread a([] { for (int i = 0; i < 100000; i++) racy++; }), b([] { for (int i = 0; i < 100000; i++) racy++; });
		a.join(); b.join();
		printf("canary: %d\n", racy);
		return 0;
	}
	const uint32_t frames = argc > 1 ? (uint32_t)atoi(argv[1]) : 500, n_tracers = argc > 2 ? (uint32_t)atoi(argv[2]) : 8, rows = argc > 3 ? (uint32_t)atoi(argv[3]) : 61;
	for (int kind = 0; kind < 2; kind++)
		if (int rc = run(kind, frames, n_tracers, rows)) return rc;
	for (int kind = 0; kind < 2; kind++)
		if (int rc = run_plain(kind, frames, n_tracers, rows)) return rc;
	printf("renderer_tsan: %u frames x %u tracers x 2 schedulers, HIP-ABI tracers (reset epochs; + 20 progressive frames each) and plain tracers: every frame right\n", frames, n_tracers);
	return 0;
}
