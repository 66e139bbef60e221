// tracer.hpp -- C++ restatement of the reference's tracer package (tracer/tracer.go,
// tracer/scheduler.go): the interface the backend plugs into and the two block schedulers.
// Go is absent from the build image, so the host layer above the C ABI is written in C++ with the
// reference's names, argument meaning and error behaviour (an empty Error is Go's nil).
#pragma once

#include <chrono>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "polaris_types.h"

namespace polaris {

struct Error {
	int code = 0;
	std::string msg;
	explicit operator bool() const { return code != 0; }
	static Error Nil() { return {}; }
};

namespace tracer {

using Duration = std::chrono::nanoseconds;

// tracer/tracer.go:6-34 -- PolarisBlockRequest has the same fields in the same order.
using BlockRequest = PolarisBlockRequest;

struct Stats { // tracer/tracer.go:37-47
	uint32_t BlockW = 0, BlockH = 0;
	Duration UpdateTime{0}, RenderTime{0};
};

enum Flag : uint8_t { Local = 1, Remote = 2, CpuDevice = 4 };             // tracer.go:49-61
enum class UpdateMode : uint8_t { Synchronous = 0, Asynchronous = 1 };    // tracer.go:63-69
enum class ChangeType : uint8_t { FrameDimensions = 0, SceneData = 1, CameraData = 2 }; // tracer.go:71-78

struct FrameDims { uint32_t w, h; };                 // payload of FrameDimensions ([2]uint32)
struct CameraData { float eye[3]; float frustum[16]; }; // what tracer.go:177-179 uses of scene.Camera

class Tracer { // tracer/tracer.go:80-111
public:
	virtual ~Tracer() = default;
	virtual std::string Id() const = 0;
	virtual uint8_t Flags() const = 0;
	virtual uint32_t Speed() const = 0;
	virtual Error Init() = 0;
	virtual void Close() = 0;
	virtual Stats *GetStats() = 0;
	// data: FrameDims* | const PolarisSceneView* | CameraData*, by ChangeType
	virtual Error UpdateState(UpdateMode, ChangeType, const void *data, Duration *took = nullptr) = 0;
	virtual Error Trace(BlockRequest *, Duration *took = nullptr) = 0;
	virtual Error MergeOutput(Tracer *other, BlockRequest *, Duration *took = nullptr) = 0;
	virtual Error SyncFramebuffer(BlockRequest *, Duration *took = nullptr) = 0;
};

class BlockScheduler { // tracer/scheduler.go:6-10
public:
	virtual ~BlockScheduler() = default;
	virtual std::vector<uint32_t> Schedule(const std::vector<Tracer *> &tracers, uint32_t frameH) = 0;
};

std::unique_ptr<BlockScheduler> NaiveScheduler();   // scheduler.go:19-30
std::unique_ptr<BlockScheduler> PerfectScheduler(); // scheduler.go:39-80

} // namespace tracer
} // namespace polaris
