// scheduler.cpp -- tracer/scheduler.go restated.  The arithmetic (float64 scaling, truncation,
// "at least one row", remainder to tracer 0) is kept operation for operation; the known answers
// of tracer/scheduler_test.go:16-20,48-55 are pinned in tests/test_host_layer.py (test_scheduler_known_answers).
#include <cmath>

#include "tracer.hpp"

namespace polaris {
namespace tracer {

// assignBlocksBasedOnSpeed, scheduler.go:83-106
static std::vector<uint32_t> assignBlocksBasedOnSpeed(const std::vector<Tracer *> &tracers, uint32_t frameH) {
	std::vector<uint32_t> blockAssignment(tracers.size());
	uint32_t speedSum = 0;
	for (Tracer *tr : tracers) speedSum += tr->Speed();
	const double scaler = double(frameH) / double(speedSum);
	uint32_t assignedRows = 0;
	for (size_t idx = 0; idx < tracers.size(); idx++) {
		uint32_t blockH = uint32_t(std::fmax(1.0, double(tracers[idx]->Speed()) * scaler));
		assignedRows += blockH;
		blockAssignment[idx] = blockH;
	}
	if (assignedRows < frameH && !blockAssignment.empty()) blockAssignment[0] += frameH - assignedRows;
	return blockAssignment;
}

namespace {

class naiveScheduler : public BlockScheduler { // scheduler.go:12-30
	std::vector<uint32_t> blockAssignment;

public:
	std::vector<uint32_t> Schedule(const std::vector<Tracer *> &tracers, uint32_t frameH) override {
		if (blockAssignment.size() != tracers.size()) blockAssignment = assignBlocksBasedOnSpeed(tracers, frameH);
		return blockAssignment;
	}
};

class perfectScheduler : public BlockScheduler { // scheduler.go:32-80
	std::vector<uint32_t> blockAssignment;

public:
	std::vector<uint32_t> Schedule(const std::vector<Tracer *> &tracers, uint32_t frameH) override {
		if (blockAssignment.size() != tracers.size()) { // first call: naive distribution
			blockAssignment = assignBlocksBasedOnSpeed(tracers, frameH);
			return blockAssignment;
		}
		double total = 0.0;
		for (Tracer *tr : tracers) {
			Stats *st = tr->GetStats();
			total += double(st->BlockH) / double(st->RenderTime.count());
		}
		const double scaler = double(frameH) / total;
		uint32_t assignedRows = 0;
		for (size_t idx = 0; idx < tracers.size(); idx++) {
			Stats *st = tracers[idx]->GetStats();
			uint32_t blockH = uint32_t(std::fmax(1.0, std::floor(double(st->BlockH) / double(st->RenderTime.count()) * scaler)));
			assignedRows += blockH;
			blockAssignment[idx] = blockH;
		}
		if (assignedRows < frameH) blockAssignment[0] += frameH - assignedRows;
		return blockAssignment;
	}
};

} // namespace

std::unique_ptr<BlockScheduler> NaiveScheduler() { return std::make_unique<naiveScheduler>(); }
std::unique_ptr<BlockScheduler> PerfectScheduler() { return std::make_unique<perfectScheduler>(); }

} // namespace tracer
} // namespace polaris
