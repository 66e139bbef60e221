// tests/tools/mock_polaris_hip.cpp -- a CPU stand-in for the part of the C ABI (include/polaris_hip.h) that
// polaris_amd/host/hip_tracer.cpp binds, for the THREAD-SANITIZER run of the host layer (tests/test_host_tsan.py; VERDICT round 5,
// item 2).  Test infrastructure: it is linked into tests/_build/renderer_tsan only, never into the product.
//
// The reference's own pattern is a mock tracer behind the Tracer interface (tracer/scheduler_test.go:82-123).  Here the mock sits one
// level lower, behind the C ABI, so that the code under the sanitizer is the REAL host layer -- renderer.cpp's worker threads,
// hip_tracer.cpp's change buffer and seed draws, scheduler.cpp -- including the branch renderer.cpp takes for a HIP primary (reset
// epochs).  The mock keeps the library's locking protocol (polaris_hip.hip): `mu` is held by a Trace from start to end, the frame
// accumulator belongs to `merge_mu`, a Trace with accumulated_samples == 0 clears the frame accumulator and advances the reset epoch
// when it STARTS, merges from other threads land under merge_mu only, wait_reset blocks on the epoch.  A "Trace" sleeps a seeded
// random time and writes 1.0 into its block's rows of the trace accumulator: after a frame every row of the primary's frame
// accumulator must hold exactly (frames accumulated) -- a merge that landed before the Reset stage, twice, or not at all shows.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "polaris_hip.h"

struct polaris_hip_tracer {
	int device = 0;
	std::mutex mu, merge_mu;
	std::condition_variable reset_cv;
	uint64_t reset_epoch = 0;
	uint32_t W = 0, H = 0;
	std::vector<float> trace_acc, frame_acc; // one float per ROW (the mock's pixels are rows)
	bool have_scene = false, have_camera = false;
	std::string error;
	std::mt19937 rng;
	uint64_t traces = 0, merges = 0;
};

static std::atomic<int> g_max_sleep_us{300};
extern "C" void mock_polaris_set_max_sleep_us(int us) { g_max_sleep_us = us; }

extern "C" {

int polaris_hip_abi_version(void) { return POLARIS_HIP_ABI_VERSION; }
int polaris_hip_device_count(void) { return 8; }
int polaris_hip_device_info(int index, char name[256], uint32_t *cus, uint32_t *mhz, uint64_t *mem) {
	if (index < 0 || index >= 8) return POLARIS_E_NO_DEVICE;
	if (name) snprintf(name, 256, "mock-gpu-%d", index);
	if (cus) *cus = 256;
	if (mhz) *mhz = 2400;
	if (mem) *mem = 1ull << 30;
	return POLARIS_OK;
}
int polaris_hip_create(int device_index, polaris_hip_tracer **out) {
	if (!out) return POLARIS_E_BAD_ARGUMENT;
	auto *h = new polaris_hip_tracer();
	h->device = device_index;
	h->rng.seed(1000u + (unsigned)device_index);
	*out = h;
	return POLARIS_OK;
}
void polaris_hip_destroy(polaris_hip_tracer *h) { delete h; }
const char *polaris_hip_last_error(polaris_hip_tracer *h) { return h ? h->error.c_str() : ""; }

int polaris_hip_resize(polaris_hip_tracer *h, uint32_t w, uint32_t hh) {
	std::lock_guard<std::mutex> lk(h->mu);
	std::lock_guard<std::mutex> lkm(h->merge_mu);
	h->W = w; h->H = hh;
	h->trace_acc.assign(hh, 0.0f);
	h->frame_acc.assign(hh, 0.0f);
	return POLARIS_OK;
}
int polaris_hip_upload_scene(polaris_hip_tracer *h, const PolarisSceneView *) { std::lock_guard<std::mutex> lk(h->mu); h->have_scene = true; return POLARIS_OK; }
int polaris_hip_set_camera(polaris_hip_tracer *h, const float *, const float *) { std::lock_guard<std::mutex> lk(h->mu); h->have_camera = true; return POLARIS_OK; }

int polaris_hip_trace(polaris_hip_tracer *h, const PolarisBlockRequest *r, const uint32_t *seeds, size_t n_seeds, PolarisTraceStats *stats) {
	std::lock_guard<std::mutex> lk(h->mu); // a Trace holds the handle's lock from start to end
	if (!h->have_scene) { h->error = "no scene data uploaded"; return POLARIS_E_NO_SCENE_DATA; }
	if (!r || (uint64_t)r->block_y + r->block_h > h->H || n_seeds < (size_t)r->samples_per_pixel * (1 + r->num_bounces) || !seeds) { h->error = "bad request"; return POLARIS_E_BAD_ARGUMENT; }
	if (r->accumulated_samples == 0) { // the Reset stage: queued when the Trace STARTS, announced at once
		{
			std::lock_guard<std::mutex> lkm(h->merge_mu);
			std::fill(h->frame_acc.begin(), h->frame_acc.end(), 0.0f);
			h->reset_epoch++;
		}
		h->reset_cv.notify_all();
	}
	std::fill(h->trace_acc.begin(), h->trace_acc.end(), 0.0f); // ClearTraceAccumulator
	const int max_us = g_max_sleep_us.load();
	if (max_us > 0) std::this_thread::sleep_for(std::chrono::microseconds(h->rng() % (unsigned)max_us));
	for (uint32_t y = r->block_y; y < r->block_y + r->block_h; y++) h->trace_acc[y] = 1.0f;
	h->traces++;
	if (stats) { memset(stats, 0, sizeof *stats); stats->primary_rays = (uint64_t)h->W * r->block_h * r->samples_per_pixel; }
	return POLARIS_OK;
}

int polaris_hip_merge(polaris_hip_tracer *dst, polaris_hip_tracer *src, const PolarisBlockRequest *r) {
	if (!dst || !src || !r) return POLARIS_E_BAD_ARGUMENT;
	std::vector<float> rows; // the source's rows, snapshotted under ITS lock (its Trace has returned: the lock is free)
	{
		std::lock_guard<std::mutex> lk(src->mu);
		if ((uint64_t)r->block_y + r->block_h > src->H) return POLARIS_E_BAD_ARGUMENT;
		rows.assign(src->trace_acc.begin() + r->block_y, src->trace_acc.begin() + r->block_y + r->block_h);
	}
	std::lock_guard<std::mutex> lk(dst->merge_mu); // never dst->mu: the destination may be tracing
	if ((uint64_t)r->block_y + r->block_h > dst->H) return POLARIS_E_BAD_ARGUMENT;
	for (uint32_t i = 0; i < r->block_h; i++) dst->frame_acc[r->block_y + i] += rows[i];
	dst->merges++;
	return POLARIS_OK;
}

int polaris_hip_reset_frame(polaris_hip_tracer *h) {
	std::lock_guard<std::mutex> lk(h->mu);
	{
		std::lock_guard<std::mutex> lkm(h->merge_mu);
		std::fill(h->frame_acc.begin(), h->frame_acc.end(), 0.0f);
		h->reset_epoch++;
	}
	h->reset_cv.notify_all();
	return POLARIS_OK;
}
int polaris_hip_reset_epoch(polaris_hip_tracer *h, uint64_t *epoch) {
	std::lock_guard<std::mutex> lk(h->merge_mu);
	*epoch = h->reset_epoch;
	return POLARIS_OK;
}
int polaris_hip_wait_reset(polaris_hip_tracer *h, uint64_t epoch) {
	std::unique_lock<std::mutex> lk(h->merge_mu);
	// (wait_until on the SYSTEM clock: libstdc++ implements wait_for / steady-clock waits with pthread_cond_clockwait, which gcc 11's
	// thread sanitizer does not intercept -- it then believes the waiting thread still holds the mutex and reports phantom double locks
	// and races; pthread_cond_timedwait is intercepted)
	if (!h->reset_cv.wait_until(lk, std::chrono::system_clock::now() + std::chrono::seconds(60), [&] { return h->reset_epoch > epoch; })) return POLARIS_E_TIMEOUT;
	return POLARIS_OK;
}
int polaris_hip_sync_framebuffer(polaris_hip_tracer *h, const PolarisBlockRequest *) {
	std::lock_guard<std::mutex> lk(h->mu);
	std::lock_guard<std::mutex> lkm(h->merge_mu); // "wait for pending merges"
	return POLARIS_OK;
}
int polaris_hip_read_framebuffer(polaris_hip_tracer *, uint8_t *, size_t) { return POLARIS_E_UNSUPPORTED; }
// which = 1: the frame accumulator, ONE float per row (the mock's own convention; n_floats >= H)
int polaris_hip_read_accumulator(polaris_hip_tracer *h, int which, float *out, size_t n_floats) {
	std::lock_guard<std::mutex> lk(h->mu);
	std::lock_guard<std::mutex> lkm(h->merge_mu);
	if (n_floats < h->H) return POLARIS_E_BAD_ARGUMENT;
	const std::vector<float> &a = which ? h->frame_acc : h->trace_acc;
	memcpy(out, a.data(), h->H * sizeof(float));
	return POLARIS_OK;
}

} // extern "C"
