// hip_tracer.hpp -- tracer.Tracer on one MI355X through the C ABI (include/polaris_hip.h);
// the C++ twin of integration/go/tracer/hip/tracer.go.
#pragma once

#include <functional>
#include <map>
#include <mutex>

#include "polaris_hip.h"
#include "tracer.hpp"

namespace polaris {
namespace tracer {
namespace hip {

struct Device { // integration/go/tracer/hip/device.go
	int Index = 0;
	std::string Name;
	uint32_t Speed = 0; // compute units * MHz / 1000 (tracer/opencl/device/device.go:219)
};
std::vector<Device> Devices(const std::vector<std::string> &blacklist);

// Source of the host PRNG draws the reference takes from Go's global math/rand
// (tracer/opencl/tracer.go:222, pipeline.go:146).
using SeedSource = std::function<uint32_t()>;

class HipTracer : public Tracer {
public:
	HipTracer(std::string id, Device dev, SeedSource seeds);
	~HipTracer() override;
	std::string Id() const override { return id_; }
	uint8_t Flags() const override { return Local; }
	uint32_t Speed() const override { return dev_.Speed; }
	Error Init() override;
	void Close() override;
	Stats *GetStats() override { return &stats_; }
	Error UpdateState(UpdateMode, ChangeType, const void *data, Duration *took) override;
	Error Trace(BlockRequest *, Duration *took) override;
	Error MergeOutput(Tracer *other, BlockRequest *, Duration *took) override;
	Error SyncFramebuffer(BlockRequest *, Duration *took) override;

	Error ReadFrameBuffer(uint8_t *rgba, size_t n);
	Error ReadAccumulator(int which, float *out, size_t n);
	// ordering merges from other threads behind this tracer's Reset stage (include/polaris_hip.h, polaris_hip_reset_epoch)
	uint64_t ResetEpoch() const { uint64_t e = 0; if (h_) (void)polaris_hip_reset_epoch(h_, &e); return e; }
	// (non-nil after POLARIS_E_TIMEOUT: the awaited Reset never came -- the caller must not merge onto an uncleared accumulator)
	Error WaitReset(uint64_t epoch) const {
		if (!h_) return Error{POLARIS_E_BAD_ARGUMENT, "hip tracer: WaitReset on a closed tracer"};
		const int rc = polaris_hip_wait_reset(h_, epoch);
		if (rc == POLARIS_OK) return Error::Nil();
		const char *m = polaris_hip_last_error(h_);
		return Error{rc, std::string("hip tracer: ") + (m ? m : "wait_reset failed")};
	}
	void ResetFrame() { if (h_) (void)polaris_hip_reset_frame(h_); } // the Reset stage on its own (also advances the epoch)
	const PolarisTraceStats &LastTraceStats() const { return last_; }
	polaris_hip_tracer *Handle() const { return h_; }

private:
	Error commitChanges(Duration *took);
	Error check(int rc);
	std::string id_;
	Device dev_;
	SeedSource seeds_;
	polaris_hip_tracer *h_ = nullptr;
	Stats stats_;
	PolarisTraceStats last_{};
	std::mutex mu_;
	std::mutex change_mu_; // the change buffer: the interactive renderer queues camera updates from another thread (opengl.go:298-300)
	// change buffer: latest update of each type wins (tracer/opencl/tracer.go:150-158)
	bool has_dims_ = false, has_scene_ = false, has_cam_ = false;
	FrameDims dims_{};
	const PolarisSceneView *scene_ = nullptr;
	CameraData cam_{};
};

} // namespace hip
} // namespace tracer
} // namespace polaris
