// renderer.hpp -- the frame loop of renderer/default.go restated in C++ (no OpenGL, no CLI):
// one worker thread per tracer, Schedule -> Trace per block -> primary.MergeOutput -> primary
// SyncFramebuffer.  This is the caller side of the hot path ("next" row f-1 of SURVEY.md 8).
#pragma once

#include <condition_variable>
#include <deque>
#include <thread>

#include "hip_tracer.hpp"

namespace polaris {
namespace renderer {

struct Options { // renderer/options.go:3-23
	uint32_t FrameW = 1024, FrameH = 1024;
	uint32_t SamplesPerPixel = 16;
	float Exposure = 1.2f;
	uint32_t NumBounces = 5;
	uint32_t MinBouncesForRR = 3;
	std::vector<std::string> BlackListedDevices;
	std::string ForcePrimaryDevice;
};

struct TracerStat { // renderer/stats.go
	std::string Id;
	bool IsPrimary = false;
	uint32_t BlockH = 0;
	float FramePercent = 0.0f;
	tracer::Duration RenderTime{0};
};
struct FrameStats {
	std::vector<TracerStat> Tracers;
	tracer::Duration RenderTime{0};
};

// w*h RGBA8 pixels, rows top to bottom, as a PNG file
Error WritePNG(const std::string &path, const uint8_t *rgba, uint32_t w, uint32_t h);

class DefaultRenderer { // renderer/default.go:21-196
public:
	// tracers are created by the caller (the seam of INTEGRATION.md section 3) and owned here
	DefaultRenderer(std::vector<std::unique_ptr<tracer::Tracer>> tracers, size_t primary, std::unique_ptr<tracer::BlockScheduler> scheduler,
	                Options opts, tracer::hip::SeedSource seeds);
	~DefaultRenderer();
	Error UpdateAll(tracer::ChangeType type, const void *data); // the three synchronous UpdateState calls of NewDefault (:70-72)
	Error Render() { return renderFrame(0); }                  // default.go:101-103
	Error renderFrame(uint32_t accumulatedSamples);            // default.go:106-171
	void Close();
	const FrameStats &Stats() const { return stats_; }
	const std::vector<uint32_t> &BlockAssignments() const { return blockAssignments_; }
	tracer::Tracer *Primary() { return tracers_[primary_].get(); }
	// The SaveFrameBuffer post-process stage (tracer/opencl/pipeline.go:215-235): the primary's RGBA8
	// frame buffer as a PNG (8-bit RGBA, non-interlaced, like Go's image/png for an image.RGBA).
	Error SaveFrameBuffer(const std::string &imgFile);
	size_t NumTracers() const { return tracers_.size(); }

private:
	void jobWorker(size_t trIndex); // default.go:174-196
	struct Channel {
		std::mutex mu;
		std::condition_variable cv;
		std::deque<tracer::BlockRequest> q;
		bool closed = false;
	};
	std::vector<std::unique_ptr<tracer::Tracer>> tracers_;
	size_t primary_;
	std::unique_ptr<tracer::BlockScheduler> scheduler_;
	Options options_;
	tracer::hip::SeedSource seeds_;
	std::vector<std::unique_ptr<Channel>> jobChans_;
	std::vector<std::thread> workers_;
	std::mutex doneMu_;
	std::condition_variable doneCv_;
	std::deque<Error> done_;
	// The primary's Trace clears its frame accumulator when it STARTS (the pipeline's Reset stage, tracer.go:208-213), while the
	// other workers merge their blocks into that accumulator as they finish (default.go:188-191): a block that finished before the
	// primary's worker got going would be wiped.  The reference has this race; here a worker's merge waits until the primary's
	// Trace of the same frame has returned (the frame cannot complete earlier anyway).
	std::mutex frameMu_;
	std::condition_variable frameCv_;
	uint64_t frame_ = 0, primaryTraced_ = 0;
	uint64_t primaryEpoch_ = 0;  // the primary's reset epoch when the frame's blocks were handed out (HIP primaries)
	bool frameResets_ = false;   // this frame's Trace on the primary clears the frame accumulator (accumulated samples == 0)
	std::vector<uint32_t> blockAssignments_;
	FrameStats stats_;
	bool closed_ = false;
};

} // namespace renderer
} // namespace polaris
