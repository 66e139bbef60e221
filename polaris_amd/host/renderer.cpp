#include "renderer.hpp"

namespace polaris {
namespace renderer {

using clock_ = std::chrono::steady_clock;

DefaultRenderer::DefaultRenderer(std::vector<std::unique_ptr<tracer::Tracer>> tracers, size_t primary,
                                 std::unique_ptr<tracer::BlockScheduler> scheduler, Options opts, tracer::hip::SeedSource seeds)
    : tracers_(std::move(tracers)), primary_(primary), scheduler_(std::move(scheduler)), options_(std::move(opts)), seeds_(std::move(seeds)) {
	stats_.Tracers.resize(tracers_.size());
	for (size_t i = 0; i < tracers_.size(); i++) {
		stats_.Tracers[i].Id = tracers_[i]->Id();
		stats_.Tracers[i].IsPrimary = i == primary_;
		jobChans_.push_back(std::make_unique<Channel>());
	}
	for (size_t i = 0; i < tracers_.size(); i++) workers_.emplace_back([this, i] { jobWorker(i); });
}

DefaultRenderer::~DefaultRenderer() { Close(); }

Error DefaultRenderer::UpdateAll(tracer::ChangeType type, const void *data) {
	for (auto &tr : tracers_)
		if (Error e = tr->UpdateState(tracer::UpdateMode::Synchronous, type, data)) return e;
	return Error::Nil();
}

void DefaultRenderer::Close() { // default.go:91-98
	if (closed_) return;
	closed_ = true;
	for (auto &ch : jobChans_) {
		std::lock_guard<std::mutex> lk(ch->mu);
		ch->closed = true;
		ch->cv.notify_all();
	}
	for (auto &w : workers_) w.join();
}

void DefaultRenderer::jobWorker(size_t trIndex) {
	Channel &ch = *jobChans_[trIndex];
	for (;;) {
		tracer::BlockRequest blockReq;
		{
			std::unique_lock<std::mutex> lk(ch.mu);
			ch.cv.wait(lk, [&] { return ch.closed || !ch.q.empty(); });
			if (ch.q.empty()) break; // channel closed
			blockReq = ch.q.front();
			ch.q.pop_front();
		}
		Error err = tracers_[trIndex]->Trace(&blockReq);
		// merge this block into the primary's frame accumulator -- called from THIS worker onto the
		// primary, concurrently with the other workers (default.go:188-191)
		if (!err) err = tracers_[primary_]->MergeOutput(tracers_[trIndex].get(), &blockReq);
		{
			std::lock_guard<std::mutex> lk(doneMu_);
			done_.push_back(err);
		}
		doneCv_.notify_one();
	}
	tracers_[trIndex]->Close(); // default.go:176-178
}

Error DefaultRenderer::renderFrame(uint32_t accumulatedSamples) {
	tracer::BlockRequest blockReq{};
	blockReq.frame_w = options_.FrameW;
	blockReq.frame_h = options_.FrameH;
	blockReq.block_w = options_.FrameW;
	blockReq.samples_per_pixel = options_.SamplesPerPixel;
	blockReq.exposure = options_.Exposure;
	blockReq.num_bounces = options_.NumBounces;
	blockReq.min_bounces_for_rr = options_.MinBouncesForRR;
	blockReq.accumulated_samples = accumulatedSamples;
	blockReq.seed = seeds_();
	if (blockReq.samples_per_pixel == 0) blockReq.samples_per_pixel = 1; // progressive mode, default.go:120-122
	const auto start = clock_::now();

	std::vector<tracer::Tracer *> raw;
	for (auto &t : tracers_) raw.push_back(t.get());
	blockAssignments_ = scheduler_->Schedule(raw, blockReq.frame_h);
	for (size_t trIndex = 0; trIndex < blockAssignments_.size(); trIndex++) {
		const uint32_t blockH = blockAssignments_[trIndex];
		blockReq.block_h = blockH;
		{
			Channel &ch = *jobChans_[trIndex];
			std::lock_guard<std::mutex> lk(ch.mu);
			ch.q.push_back(blockReq); // a COPY per tracer (default.go:130)
			ch.cv.notify_one();
		}
		stats_.Tracers[trIndex].BlockH = blockH;
		stats_.Tracers[trIndex].FramePercent = 100.0f * float(blockH) / float(blockReq.frame_h);
		blockReq.block_y += blockH;
	}
	Error first;
	for (size_t pending = tracers_.size(); pending != 0; pending--) {
		std::unique_lock<std::mutex> lk(doneMu_);
		doneCv_.wait(lk, [&] { return !done_.empty(); });
		Error e = done_.front();
		done_.pop_front();
		if (e && !first) first = e;
	}
	if (first) return first;
	blockReq.block_y = 0; // post-process on the primary over the whole frame (default.go:158-161)
	blockReq.block_h = blockReq.frame_h;
	if (Error e = tracers_[primary_]->SyncFramebuffer(&blockReq)) return e;
	stats_.RenderTime = std::chrono::duration_cast<tracer::Duration>(clock_::now() - start);
	for (size_t i = 0; i < tracers_.size(); i++) stats_.Tracers[i].RenderTime = tracers_[i]->GetStats()->RenderTime;
	return Error::Nil();
}

} // namespace renderer
} // namespace polaris
