#include "renderer.hpp"

#include <zlib.h>

#include <cstdio>
#include <cstring>

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
		// A merge of this frame must land behind the Reset stage of the primary's Trace, which clears the frame accumulator when
		// it STARTS (tracer.go:208-213; the reference leaves the order to chance).  A HIP primary announces the queued clear
		// (reset epoch): a secondary that is done early merges while the primary still traces -- merges run on the primary's
		// merge stream, not under the lock its Trace holds.  Any other primary: wait until its Trace has returned.
		{
			std::unique_lock<std::mutex> lk(frameMu_);
			if (trIndex == primary_) {
				primaryTraced_ = frame_;
				frameCv_.notify_all();
				// a Trace that failed before it reached the device (a pending state change that could not be committed) has
				// announced nothing: release the workers that wait for this frame's reset
				auto *hp = dynamic_cast<tracer::hip::HipTracer *>(tracers_[primary_].get());
				if (err && frameResets_ && hp && hp->ResetEpoch() == primaryEpoch_) hp->ResetFrame();
			} else if (frameResets_) {
				auto *hp = dynamic_cast<tracer::hip::HipTracer *>(tracers_[primary_].get());
				if (hp) {
					const uint64_t epoch = primaryEpoch_;
					lk.unlock();
					const Error werr = hp->WaitReset(epoch); // a reset that never came: the block must not land on an uncleared frame
					if (!err && werr) err = werr;
				} else {
					frameCv_.wait(lk, [&] { return primaryTraced_ == frame_; });
				}
			}
		}
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
	{
		// scheduler.go:70-76 gives every tracer at least one row AFTER flooring the shares and only tops up when the blocks add up
		// to LESS than the frame: with very unequal speeds they add up to more ([7, 1, 1, 1] for 8 rows) and the reference's last
		// block runs off the frame.  The frame loop takes rows back from the tallest blocks (same rule as
		// polaris_amd/distributed.py::fit_rows); a no-op for every assignment that fits, so scheduler.cpp stays the reference's arithmetic.
		uint64_t sum = 0;
		for (uint32_t h : blockAssignments_) sum += h;
		while (sum > blockReq.frame_h) {
			size_t tallest = 0;
			for (size_t i = 1; i < blockAssignments_.size(); i++)
				if (blockAssignments_[i] > blockAssignments_[tallest]) tallest = i;
			if (blockAssignments_[tallest] <= 1) break;
			blockAssignments_[tallest]--;
			sum--;
		}
	}
	{
		std::lock_guard<std::mutex> lk(frameMu_);
		frame_++;
		frameResets_ = accumulatedSamples == 0;
		if (auto *hp = dynamic_cast<tracer::hip::HipTracer *>(tracers_[primary_].get())) primaryEpoch_ = hp->ResetEpoch();
	}
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

Error DefaultRenderer::SaveFrameBuffer(const std::string &imgFile) {
	auto *p = dynamic_cast<tracer::hip::HipTracer *>(Primary());
	if (!p) return Error{POLARIS_E_UNSUPPORTED, "SaveFrameBuffer: the primary tracer has no readable frame buffer"};
	const uint32_t W = options_.FrameW, H = options_.FrameH;
	std::vector<uint8_t> rgba((size_t)W * H * 4);
	if (Error e = p->ReadFrameBuffer(rgba.data(), rgba.size())) return e;
	return WritePNG(imgFile, rgba.data(), W, H);
}

Error WritePNG(const std::string &path, const uint8_t *rgba, uint32_t w, uint32_t h) {
	std::vector<uint8_t> raw((size_t)h * (1 + (size_t)w * 4));
	for (uint32_t y = 0; y < h; y++) { // filter type 0 on every row
		raw[(size_t)y * (1 + (size_t)w * 4)] = 0;
		memcpy(&raw[(size_t)y * (1 + (size_t)w * 4) + 1], rgba + (size_t)y * w * 4, (size_t)w * 4);
	}
	uLongf clen = compressBound((uLong)raw.size());
	std::vector<uint8_t> comp(clen);
	if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return Error{POLARIS_E_DEVICE, "png: deflate failed"};
	FILE *f = fopen(path.c_str(), "wb");
	if (!f) return Error{POLARIS_E_BAD_ARGUMENT, "open " + path + ": cannot create file"};
	auto be32 = [](uint8_t *p, uint32_t v) { p[0] = uint8_t(v >> 24); p[1] = uint8_t(v >> 16); p[2] = uint8_t(v >> 8); p[3] = uint8_t(v); };
	auto chunk = [&](const char type[4], const uint8_t *body, uint32_t len) {
		uint8_t hdr[8], crcb[4];
		be32(hdr, len);
		memcpy(hdr + 4, type, 4);
		uLong crc = crc32(0L, hdr + 4, 4);
		if (len) crc = crc32(crc, body, len);
		be32(crcb, (uint32_t)crc);
		fwrite(hdr, 1, 8, f);
		if (len) fwrite(body, 1, len, f);
		fwrite(crcb, 1, 4, f);
	};
	static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
	fwrite(sig, 1, 8, f);
	uint8_t ihdr[13];
	be32(ihdr, w); be32(ihdr + 4, h);
	ihdr[8] = 8; ihdr[9] = 6; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0; // 8-bit RGBA, no interlace
	chunk("IHDR", ihdr, 13);
	chunk("IDAT", comp.data(), (uint32_t)clen);
	chunk("IEND", nullptr, 0);
	const bool ok = !ferror(f);
	fclose(f);
	return ok ? Error::Nil() : Error{POLARIS_E_DEVICE, "png: write to " + path + " failed"};
}

} // namespace renderer
} // namespace polaris
