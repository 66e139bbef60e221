// polaris_hip.hip -- host side of the C ABI declared in include/polaris_hip.h.
//
// Replaces the reference's Go host layer for the tracer path (tracer/opencl/tracer.go,
// pipeline.go, resources.go, buffers.go, device/*.go) with one HIP stream per tracer handle
// and NO host round trips inside a Trace: the reference ends every one of its ~21-26 launches
// per sample in clFinish and resets two ray counters from the host before every shadeHits
// (resources.go:230-238, device/kernel.go:124); here a whole batch of samples is enqueued
// back to back, live-ray counts stay on the device, and the only synchronisation is the
// one at the end of Trace (which the interface requires: Trace is synchronous).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include <unistd.h>

#include "kernels.h"
#include "polaris_hip.h"

using namespace pol;

#ifndef POLARIS_LDS_TOP_MAX_PAIRS
#define POLARIS_LDS_TOP_MAX_PAIRS (8u * kLdsTopNodes)
#endif

namespace {

thread_local std::string g_thread_error;
std::atomic<uint64_t> g_error_seq{0};

struct KernelTimer { double ms = 0.0; uint64_t launches = 0; };

struct DevBuf {
	void *p = nullptr;
	size_t bytes = 0;
};

} // namespace

struct polaris_hip_tracer {
	int device = 0;
	char pci_bus_id[32] = {}; // which GPU `device` is (device indices are per process); "" if the runtime does not say
	uint64_t merge_counts[POLARIS_MERGE_BRANCHES] = {}; // which branch every merge onto this tracer took (under merge_mu; polaris_hip_merge_counts)
	hipStream_t stream = nullptr;
	std::mutex mu;
	std::string error;
	// The frame accumulator belongs to the MERGE stream: the Reset stage, MergeOutput and merge_device run there, under
	// merge_mu only -- never under `mu`, which a Trace holds from start to end -- so a secondary's MergeOutput onto the
	// primary overlaps the primary's own Trace (Exec1DNoWait, tracer/opencl/resources.go:119; renderer/default.go:188-191
	// calls it from the secondaries' goroutines).  SyncFramebuffer makes the main stream wait for the merges queued so far.
	// Lock order: mu, then merge_mu.  W / H / frame_acc are written under both and may be read under either.
	hipStream_t merge_stream = nullptr;
	std::mutex merge_mu;
	hipEvent_t ev_merged = nullptr;
	// Reset epoch: how many times a Trace with accumulated_samples == 0 (or reset_frame) has got as far as queueing the clear
	// of the frame accumulator -- or has failed before it could.  A host that merges from other threads waits for the
	// primary's epoch to advance before it queues this frame's merges (polaris_hip_wait_reset), instead of for the
	// primary's whole Trace.
	uint64_t reset_epoch = 0;
	std::condition_variable reset_cv;
	std::string merge_error;              // last error of a merge (under merge_mu); `error` belongs to mu
	uint64_t error_seq = 0, merge_error_seq = 0; // which of the two is the newer one (g_error_seq)

	// frame-sized state (buffers.go:127-174)
	uint32_t W = 0, H = 0;
	float4 *trace_acc = nullptr, *frame_acc = nullptr; // trace_acc == ring[ring_pos]
	uchar4 *framebuffer = nullptr;
	// The trace accumulator as a ring (polaris_hip_ipc_export): with depth > 1 every Trace writes the next slot, so a peer
	// process may still read frame f's rows while frame f + 1 is traced.  Depth 1 = the reference's single buffer.
	float4 *ring[POLARIS_IPC_MAX_DEPTH] = {};
	uint32_t ring_depth = 1, ring_pos = 0;
	// One inter-process event PER RING SLOT, recorded at the end of the Trace that wrote the slot (once exported).  A slot's event
	// is re-recorded only when a Trace comes round to the slot again -- which the caller's protocol forbids while a peer may still
	// read it -- so a peer's wait on slot s's event names exactly the Trace whose rows it is about to read (round 4 had ONE event
	// re-recorded by every Trace: a primary merging frame f while the peer was already in Trace f + 1 waited for whichever it got).
	hipEvent_t ev_ipc_done[POLARIS_IPC_MAX_DEPTH] = {};
	// Events recorded by OTHER handles' merge streams behind their reads of this handle's trace accumulator
	// (polaris_hip_merge with dst != src): this handle's next Trace waits for them before it clears the rows.
	struct Reader { hipEvent_t ev; int device; };
	std::mutex readers_mu;
	std::vector<Reader> readers, reader_pool;

	// scene (buffers.go:180-201), re-laid out by scene_layout.h
	bool have_scene = false;
	std::vector<DevBuf> scene_bufs;
	BvhDev bvh{};
	SceneDev scene{};
	int max_stack = 0;
	int node_mode = kNodesGlobal; // where k_trace reads node records from (kernels.h NodeMode), resolved at upload
	int opt_node_mode = -1;       // -1 = by scene size
	uint32_t tex_bytes = 0; // size of the uploaded texture blob (without its padding)
	int trace_resident_per_cu = 6, occl_resident_per_cu = 6; // workgroups of the selected k_trace<closest | any hit> variant a CU holds at once (occupancy API, at upload)
	bool tiny_one = false;        // tiny-scene mode: the scene is one instance whose boxes all bound their subtrees (kernels.h k_trace, ONE); option tiny_one = 0 keeps the general variant
	int opt_tiny_one = 1;
	uint32_t tiny_lds_bytes = 0;  // tiny-scene mode: the dynamic LDS block of a k_trace workgroup (stack rows + tree + triangle records: plan_tiny_lds)
	int opt_lds_tris = -1;        // tiny-scene mode: triangle records kept in LDS; -1 = as many as fit, 0 = none (A/B aid)

	// camera (tracer.go:175-179)
	bool have_camera = false;
	CameraArgs cam{};

	// wavefront batch state: up to kMaxPipes pipelines (option "overlap", default 4; a Trace uses min(overlap, #batches) of
	// them) so that consecutive batches overlap: the sparse late-bounce launches of batch i run beside the dense early
	// bounces of batch i+1 on another stream
	struct Pipe {
		hipStream_t q = nullptr;
		size_t slots = 0; // capacity in slots
		Streams st{};
		// batched mode: one NEE record array (occ_e) and one shadow-ray count array (cnt_occ) PER BOUNCE, kept until k_fold_nee has
		// added the batch's unoccluded records to the per-path radiance (kernels.h, nee_unoccluded); [0] are st.occ_e / st.cnt_occ
		float4 *nee[POLARIS_MAX_BOUNCES] = {};
		uint8_t *vis[POLARIS_MAX_BOUNCES] = {};
		uint32_t *cnt_occ_b[POLARIS_MAX_BOUNCES] = {};
		uint32_t nee_bounces = 0; // how many of them are allocated
		std::vector<DevBuf> bufs;
		hipEvent_t done = nullptr; // recorded after the pipe's last resolve
	};
	static constexpr int kMaxPipes = 8;
	Pipe pipe[kMaxPipes];
	int opt_overlap = 4; // number of pipelines used (1 = no overlap)
	uint32_t *d_seeds = nullptr;
	size_t seeds_cap = 0;
	unsigned long long *d_stats = nullptr;
	float4 *d_cam_o = nullptr; // eye | FLT_MAX: the one origin record of every camera ray (launch_trace, camera)
	int num_cus = 256;
	void *staging = nullptr; // peer-merge staging strip
	size_t staging_bytes = 0;
	int opt_ipc_staged = 0;  // testing aid: peers opened from now on are merged through the staging strip (the path of a GPU without peer access)

	// options
	int64_t opt_samples_per_batch = 0; // 0 = auto
	int opt_exact = 0;
	int opt_packet_shadow = 0;  // shadow rays of the first N bounces go through the packet kernel too
	int opt_packet_primary = -1; // wave-packet traversal (k_trace_packet) for bounce 0: 1/0, -1 = by scene size (packet_primary below)
	bool packet_primary = true;  // resolved at upload
	int opt_time_kernels = 0;
	int opt_trace_wgs_per_cu = 0; // 0 = auto (what the LDS stack admits)
	int opt_trace_grid = 0;       // 0 = auto; > 0: the persistent traversal grid in workgroups (A/B aid)
	int opt_max_leaf_tris = -1;   // subdivide bigger triangle leaves at upload (0 = keep the caller's leaves, -1 = by scene size)
	int opt_stage_lds = 1;    // k_shade stages material nodes / lights / texture metadata in LDS when they fit
	int opt_shade_wave = 1;   // 1 = persistent wave-per-chunk shading (k_shade_wave), 0 = one workgroup per chunk (k_shade)
	int opt_shade_wgs_per_cu = 4;
	int opt_shade_wave_from = -1; // first bounce shaded by k_shade_wave; -1 = the bounce AFTER Russian roulette starts thinning the
	                              // chunks (min_bounces_for_rr + 1: the RR bounce itself still shades dense chunks); earlier bounces use k_shade
	int opt_shade_sort = -1; // first bounce whose rays k_shade groups by shading class; -1 = default (1), POLARIS_MAX_BOUNCES = never
	int opt_hit12 = 1;     // 12-byte hit records inside a Trace (A/B aid: 0 = 16)
	int opt_o12 = 1;       // 12-byte origins of the closest-hit rays inside a Trace (A/B aid: 0 = 16)
	int opt_traversal = 1; // 1 = persistent waves with lane refill (k_trace), 0 = one ray per lane (k_intersect/k_occlusion)

	// per-kernel timing (option time_kernels)
	struct Pending { const char *name; hipEvent_t a, b; };
	std::vector<Pending> pending;
	std::vector<Pending> merge_pending; // launches on the merge stream (under merge_mu)
	std::vector<hipEvent_t> event_pool;
	std::map<std::string, KernelTimer> timers;
	std::map<std::string, std::string> timer_symbol; // timer name -> the kernel symbol it last bracketed (polaris_hip_kernel_symbol)
	int last_shade_timer[POLARIS_MAX_BOUNCES] = {};            // per bounce of the last Trace: 0 shade_first, 1 shade_sort, 2 shade_plain, 3 shade_wave
	uint64_t last_shade_counts[3 * POLARIS_MAX_BOUNCES] = {}; // per bounce of the last Trace: shaded hits, shaded misses, emitter hits
	hipEvent_t ev_start = nullptr, ev_stop = nullptr, ev_fork = nullptr;
};

// Another process's trace accumulator ring, opened through HIP IPC on `owner`'s device.
struct polaris_hip_peer {
	polaris_hip_tracer *owner = nullptr;
	uint32_t depth = 0, W = 0, H = 0;
	void *mem[POLARIS_IPC_MAX_DEPTH] = {};
	hipEvent_t ev[POLARIS_IPC_MAX_DEPTH] = {}; // per slot: the peer's "the Trace that wrote this slot is done" event (null: the exporter had none)
	PolarisPeerInfo info{};                    // what the mapping is (polaris_hip_peer_info); info.same_device picks the merge branch counted
};


namespace {

int fail(polaris_hip_tracer *h, int code, const char *fmt, ...) {
	char buf[1024];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	if (h) { h->error = buf; h->error_seq = ++g_error_seq; }
	g_thread_error = buf;
	return code;
}

#define HIP_TRY(h, expr)                                                                              \
	do {                                                                                              \
		hipError_t e_ = (expr);                                                                       \
		if (e_ != hipSuccess) return fail(h, POLARIS_E_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); \
	} while (0)

template <typename T>
int dev_alloc(polaris_hip_tracer *h, std::vector<DevBuf> &pool, T **out, size_t count) {
	void *p = nullptr;
	size_t bytes = std::max<size_t>(count * sizeof(T), 16);
	HIP_TRY(h, hipMalloc(&p, bytes));
	pool.push_back({p, bytes});
	*out = (T *)p;
	return POLARIS_OK;
}

template <typename T>
int dev_upload(polaris_hip_tracer *h, std::vector<DevBuf> &pool, T **out, const void *src, size_t count) {
	int rc = dev_alloc(h, pool, out, count);
	if (rc) return rc;
	if (count) HIP_TRY(h, hipMemcpyAsync(*out, src, count * sizeof(T), hipMemcpyHostToDevice, h->stream));
	return POLARIS_OK;
}

void free_pool(std::vector<DevBuf> &pool) {
	for (auto &b : pool)
		if (b.p) (void)hipFree(b.p);
	pool.clear();
}

// Bracket a launch with events when time_kernels is on.
struct Timed {
	polaris_hip_tracer *h;
	const char *name;
	hipStream_t q;
	bool on_merge; // a launch on the merge stream: the caller holds merge_mu (not mu), so the event pool is not touched
	hipEvent_t a = nullptr, b = nullptr;
	Timed(polaris_hip_tracer *h_, const char *n, hipStream_t q_ = nullptr, bool merge = false) : h(h_), name(n), q(q_ ? q_ : h_->stream), on_merge(merge) {
		if (!h->opt_time_kernels) return;
		auto get = [&]() {
			hipEvent_t e;
			if (!on_merge && !h->event_pool.empty()) { e = h->event_pool.back(); h->event_pool.pop_back(); }
			else if (hipEventCreate(&e) != hipSuccess) e = nullptr;
			return e;
		};
		a = get(); b = get();
		if (a) (void)hipEventRecord(a, q);
	}
	~Timed() {
		if (!a || !b) return;
		(void)hipEventRecord(b, q);
		(on_merge ? h->merge_pending : h->pending).push_back({name, a, b});
	}
};

void collect_timers(polaris_hip_tracer *h) { // caller holds mu; the main stream must be idle
	auto take = [&](std::vector<polaris_hip_tracer::Pending> &list, bool may_be_running) {
		std::vector<polaris_hip_tracer::Pending> keep;
		for (auto &p : list) {
			float ms = 0.0f;
			const hipError_t e = hipEventElapsedTime(&ms, p.a, p.b);
			if (e == hipErrorNotReady && may_be_running) { (void)hipGetLastError(); keep.push_back(p); continue; }
			if (e == hipSuccess) {
				auto &t = h->timers[p.name];
				t.ms += ms;
				t.launches++;
			}
			h->event_pool.push_back(p.a);
			h->event_pool.push_back(p.b);
		}
		list.swap(keep);
	};
	take(h->pending, false);
	std::lock_guard<std::mutex> lk(h->merge_mu); // (another thread may be queueing a merge right now: its events stay pending)
	take(h->merge_pending, true);
}

// Everything queued on the merge stream so far happens before whatever is queued on `q` next (caller holds mu).
hipError_t join_merges(polaris_hip_tracer *h, hipStream_t q) {
	std::lock_guard<std::mutex> lk(h->merge_mu);
	hipError_t e = hipEventRecord(h->ev_merged, h->merge_stream);
	if (e == hipSuccess) e = hipStreamWaitEvent(q, h->ev_merged, 0);
	return e;
}

// Whatever other handles' merge streams still read of this handle's trace accumulator happens before what is queued on `q`
// next (caller holds mu).  The events go back to the pool: a later record simply re-arms them.
hipError_t wait_readers(polaris_hip_tracer *h, hipStream_t q) {
	std::lock_guard<std::mutex> lk(h->readers_mu);
	hipError_t first = hipSuccess;
	for (auto &r : h->readers) {
		const hipError_t e = hipStreamWaitEvent(q, r.ev, 0);
		if (first == hipSuccess) first = e;
		h->reader_pool.push_back(r);
	}
	h->readers.clear();
	return first;
}

// An event on `device` (the current device) for a read of src's trace accumulator; recorded by the caller, then handed to src.
hipEvent_t reader_event(polaris_hip_tracer *src, int device) {
	std::lock_guard<std::mutex> lk(src->readers_mu);
	for (size_t i = 0; i < src->reader_pool.size(); i++)
		if (src->reader_pool[i].device == device) {
			hipEvent_t e = src->reader_pool[i].ev;
			src->reader_pool.erase(src->reader_pool.begin() + (long)i);
			return e;
		}
	hipEvent_t e = nullptr;
	if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
	return e;
}

void free_ring(polaris_hip_tracer *h) { // caller holds mu + merge_mu, every stream idle
	for (uint32_t i = 0; i < POLARIS_IPC_MAX_DEPTH; i++) {
		if (h->ring[i]) (void)hipFree(h->ring[i]);
		h->ring[i] = nullptr;
	}
	h->trace_acc = nullptr;
	h->ring_depth = 1;
	h->ring_pos = 0;
}

hipError_t sync_all(polaris_hip_tracer *h) { // every pipeline of the handle idle (before anything the kernels use is freed)
	hipError_t first = h->merge_stream ? hipStreamSynchronize(h->merge_stream) : hipSuccess;
	for (int p = 0; p < polaris_hip_tracer::kMaxPipes; p++)
		if (h->pipe[p].q) {
			const hipError_t e = hipStreamSynchronize(h->pipe[p].q);
			if (first == hipSuccess) first = e;
		}
	return first;
}

int ensure_streams(polaris_hip_tracer *h, int p, size_t slots, bool want_inst, uint32_t nee_bounces = 1) {
	polaris_hip_tracer::Pipe &P = h->pipe[p];
	nee_bounces = std::max(1u, std::min<uint32_t>(nee_bounces, POLARIS_MAX_BOUNCES));
	if (slots <= P.slots && (!want_inst || P.st.hit_inst) && nee_bounces <= P.nee_bounces) return POLARIS_OK;
	slots = std::max(slots, P.slots);
	nee_bounces = std::max(nee_bounces, P.nee_bounces);
	HIP_TRY(h, hipStreamSynchronize(P.q));
	free_pool(P.bufs);
	P.st = Streams{};
	P.slots = 0;
	P.nee_bounces = 0;
	for (auto &q : P.nee) q = nullptr;
	for (auto &q : P.vis) q = nullptr;
	for (auto &q : P.cnt_occ_b) q = nullptr;
	const size_t wgs = slots / WG;
	int rc = 0;
	rc |= dev_alloc(h, P.bufs, &P.st.ray_o, slots);
	rc |= dev_alloc(h, P.bufs, &P.st.ray_d, slots);
	rc |= dev_alloc(h, P.bufs, &P.st.thr, slots);
	rc |= dev_alloc(h, P.bufs, &P.st.hit, slots);
	rc |= dev_alloc(h, P.bufs, &P.st.occ_o, slots);
	rc |= dev_alloc(h, P.bufs, &P.st.occ_d, slots);
	rc |= dev_alloc(h, P.bufs, &P.st.occ_e, slots);
	rc |= dev_alloc(h, P.bufs, &P.st.lsum, slots);
	rc |= dev_alloc(h, P.bufs, &P.st.cnt_ray, wgs);
	rc |= dev_alloc(h, P.bufs, &P.st.cnt_occ, wgs);
	rc |= dev_alloc(h, P.bufs, &P.st.pfx, wgs);
	rc |= dev_alloc(h, P.bufs, &P.st.wg_stat, wgs);
	rc |= dev_alloc(h, P.bufs, &P.st.emask[0], wgs * 8);
	rc |= dev_alloc(h, P.bufs, &P.st.emask[1], wgs * 8);
	if (want_inst) rc |= dev_alloc(h, P.bufs, &P.st.hit_inst, slots);
	P.nee[0] = P.st.occ_e;
	P.cnt_occ_b[0] = P.st.cnt_occ;
	rc |= dev_alloc(h, P.bufs, &P.vis[0], slots);
	P.st.vis = P.vis[0];
	for (uint32_t b = 1; b < nee_bounces; b++) {
		rc |= dev_alloc(h, P.bufs, &P.nee[b], slots);
		rc |= dev_alloc(h, P.bufs, &P.vis[b], slots);
		rc |= dev_alloc(h, P.bufs, &P.cnt_occ_b[b], wgs);
	}
	if (rc) { free_pool(P.bufs); P.st = Streams{}; for (auto &q : P.nee) q = nullptr; for (auto &q : P.vis) q = nullptr; for (auto &q : P.cnt_occ_b) q = nullptr; return rc; }
	P.slots = slots;
	P.nee_bounces = nee_bounces;
	return POLARIS_OK;
}

int check_request(polaris_hip_tracer *h, const PolarisBlockRequest *r) {
	if (!r) return fail(h, POLARIS_E_BAD_ARGUMENT, "block request is null");
	if (h->W == 0 || h->H == 0) return fail(h, POLARIS_E_BAD_ARGUMENT, "frame dimensions not set (UpdateState FrameDimensions)");
	if (r->frame_w != h->W || r->frame_h != h->H)
		return fail(h, POLARIS_E_BAD_ARGUMENT, "request frame %ux%u does not match the tracer's %ux%u", r->frame_w, r->frame_h, h->W, h->H);
	if (r->block_h == 0 || (uint64_t)r->block_y + r->block_h > h->H)
		return fail(h, POLARIS_E_BAD_ARGUMENT, "block rows [%u,%u) outside the frame", r->block_y, r->block_y + r->block_h);
	if (r->block_x != 0 || (r->block_w != 0 && r->block_w != h->W))
		return fail(h, POLARIS_E_BAD_ARGUMENT, "only full-width row blocks are supported (BlockW = FrameW, renderer/default.go:110)");
	return POLARIS_OK;
}

inline uint32_t grid_for(size_t n) { return (uint32_t)((n + WG - 1) / WG); }

// k_trace<ANY_HIT, STACK, NODES> of the uploaded scene: STACK from the exact depth the scene needs, NODES from its size
// (kernels.h, NodeMode).  fn = the kernel (for the occupancy query), block = its workgroup size.
template <bool ANY_HIT>
const void *trace_kernel(polaris_hip_tracer *h, int *block) {
	*block = h->node_mode == kNodesLdsAll ? kTinyBlock : WG;
	if (h->node_mode == kNodesLdsAll) return h->tiny_one ? (const void *)k_trace<ANY_HIT, 16, kNodesLdsAll, true> : (const void *)k_trace<ANY_HIT, 16, kNodesLdsAll, false>;
	const bool top = h->node_mode == kNodesLdsTop;
	if (h->max_stack <= 16) return top ? (const void *)k_trace<ANY_HIT, 16, kNodesLdsTop> : (const void *)k_trace<ANY_HIT, 16, kNodesGlobal>;
	if (h->max_stack <= 24) return top ? (const void *)k_trace<ANY_HIT, 24, kNodesLdsTop> : (const void *)k_trace<ANY_HIT, 24, kNodesGlobal>;
	return top ? (const void *)k_trace<ANY_HIT, 32, kNodesLdsTop> : (const void *)k_trace<ANY_HIT, 32, kNodesGlobal>;
}

// the symbol rocprofv3 prints for that kernel (bench.py names its roofline objects by it)
template <bool ANY_HIT>
std::string trace_symbol(polaris_hip_tracer *h) {
	char buf[96];
	const char *a = ANY_HIT ? "true" : "false";
	if (h->node_mode == kNodesLdsAll) snprintf(buf, sizeof buf, "pol::k_trace<%s, 16, %d, %s>", a, (int)kNodesLdsAll, h->tiny_one ? "true" : "false");
	else snprintf(buf, sizeof buf, "pol::k_trace<%s, %d, %d, false>", a, h->max_stack <= 16 ? 16 : (h->max_stack <= 24 ? 24 : 32), h->node_mode);
	return buf;
}

// camera: the closest-hit launch of a batch's camera rays -- their common origin comes from the handle's one-record buffer
// (kernels.h k_trace, o_mask), the origin stream is neither written nor read for them.
template <bool ANY_HIT>
hipError_t launch_trace(polaris_hip_tracer *h, polaris_hip_tracer::Pipe &P, const Streams &st_in, uint32_t grid, uint32_t chunks, float4 *acc, bool camera = false) {
	int block = WG;
	const void *fn = trace_kernel<ANY_HIT>(h, &block);
	Streams st = st_in;
	uint32_t o_mask = ~0u;
	if (camera && !ANY_HIT) { st.ray_o = h->d_cam_o; o_mask = 0u; }
	void *args[] = {(void *)&st, (void *)&h->bvh, (void *)&chunks, (void *)&acc, (void *)&h->d_stats, (void *)&o_mask};
	return hipLaunchKernel(fn, dim3(grid), dim3(block), args, h->node_mode == kNodesLdsAll ? h->tiny_lds_bytes : 0, P.q);
}

// Tiny-scene mode: lay out the dynamic LDS block of a k_trace workgroup for the uploaded scene (kernels.h, k_trace) -- the stack
// rows its tree needs, its pair records, and as many triangle records (slots in descending order of how often their leaf is
// reached, scene_layout.h) as still fit while TWO workgroups share a CU's LDS, which the occupancy query confirms.
int plan_tiny_lds(polaris_hip_tracer *h, size_t n_slots) {
	// (no dummy row: the bytes in front of the stack serve; a scene that IS one instance is entered at ray set-up without an exit
	// marker, which the layout's stack need counts)
	const uint32_t rows = (uint32_t)std::max(1, h->max_stack - (h->bvh.root_is_instance ? 1 : 0));
	constexpr uint32_t kRowBytes = kTinyBlock * (uint32_t)sizeof(int16_t), kTriBytes = 9 * sizeof(float) + sizeof(uint32_t);
	const uint32_t nodes = h->bvh.num_pairs * (uint32_t)sizeof(PairNode);
	hipDeviceProp_t prop;
	size_t lds_per_cu = 160 * 1024;
	if (hipGetDeviceProperties(&prop, h->device) == hipSuccess && prop.maxSharedMemoryPerMultiProcessor > 0) lds_per_cu = prop.maxSharedMemoryPerMultiProcessor;
	const size_t half = lds_per_cu / 2;
	const size_t slack = 256; // the kernel's static LDS (work cursor, dummy arrays)
	auto stack_off = [&](uint32_t tris) { return std::max<uint32_t>(kRowBytes, (nodes + tris * kTriBytes + 15u) & ~15u); };
	if (stack_off(0) + rows * kRowBytes + slack > half) return fail(h, POLARIS_E_DEVICE, "tiny-scene mode: stack and tree do not fit half a CU's LDS (%zu bytes)", half);
	uint32_t tris = (uint32_t)std::min<size_t>(n_slots, (half - slack - rows * kRowBytes - nodes - 16) / kTriBytes);
	// a wave with lanes on both fetch paths pays for both: below half of the slots the LDS copy costs more than it saves (measured:
	// 150 of the Cornell box's 808 slots +2 % frame time, 300 -1.5 %, 532 -3.5 %)
	if (tris < n_slots / 2) tris = 0;
	if (h->opt_lds_tris >= 0) tris = std::min<uint32_t>(tris, (uint32_t)h->opt_lds_tris);
	for (;;) {
		h->bvh.tiny_stack_off = stack_off(tris);
		h->bvh.lds_tris = tris;
		h->tiny_lds_bytes = h->bvh.tiny_stack_off + rows * kRowBytes;
		int worst = 8;
		for (int any = 0; any < 2; any++) {
			int block_unused = 0;
			const void *fn = any ? trace_kernel<true>(h, &block_unused) : trace_kernel<false>(h, &block_unused);
			// (the attribute belongs to the kernel, not to this handle: always the most any scene may ask for, so that handles
			// with different scenes in one process do not lower it under each other)
			HIP_TRY(h, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(half - slack)));
			int n = 0;
			if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, kTinyBlock, h->tiny_lds_bytes) != hipSuccess) { (void)hipGetLastError(); n = 2; } // (no answer: trust the arithmetic)
			worst = std::min(worst, n);
		}
		if (worst >= 2 || tris == 0) break;
		tris = tris > 32 ? tris - 32 : 0; // the allocation granularity was coarser than assumed: give some back
	}
	if (getenv("POLARIS_DEBUG")) fprintf(stderr, "[polaris] tiny mode: %u stack rows at %u, %u pair records, %u of %zu triangle records in LDS (%u bytes per workgroup)\n", rows, h->bvh.tiny_stack_off, h->bvh.num_pairs, h->bvh.lds_tris, n_slots, h->tiny_lds_bytes);
	return POLARIS_OK;
}

// Resident workgroups per CU of the k_trace variant launch_trace<ANY_HIT> picks.
template <bool ANY_HIT>
int trace_occupancy(polaris_hip_tracer *h) {
	int block = WG, n = 0;
	const void *fn = trace_kernel<ANY_HIT>(h, &block);
	if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, block, h->node_mode == kNodesLdsAll ? h->tiny_lds_bytes : 0) != hipSuccess || n < 1) n = h->node_mode == kNodesLdsAll ? 1 : (h->max_stack <= 24 ? 6 : 5);
	if (getenv("POLARIS_DEBUG")) fprintf(stderr, "[polaris] k_trace<%d> node mode %d: %d resident workgroups of %d threads per CU\n", (int)ANY_HIT, h->node_mode, n, block);
	return std::min(n, 8);
}

// One wavefront batch: K samples starting at sample s0, on pipeline p.
// Returns the first launch error of the batch (checked after every batch by the caller: a failed launch in batch 1 is reported
// before batch 2 is queued behind it).
hipError_t launch_batch(polaris_hip_tracer *h, int p, const PolarisBlockRequest *r, uint32_t s0, uint32_t K, uint32_t N, uint32_t Npad,
                        bool exact, hipEvent_t resolve_after) {
	hipError_t first_err = hipSuccess;
	auto note = [&](hipError_t e) { if (first_err == hipSuccess && e != hipSuccess) first_err = e; };
	polaris_hip_tracer::Pipe &P = h->pipe[p];
	const uint32_t B = r->num_bounces, stride = 1 + B;
	const uint32_t wgs_per_sample = Npad / WG, wgs = K * wgs_per_sample;
	hipStream_t q = P.q;
	P.st.hit12 = h->opt_hit12 ? 1u : 0u; // (a Trace never reads a hit's distance: kernels.h Streams::hit12)
	P.st.o12 = h->opt_o12 ? 1u : 0u;     // (... and its closest-hit rays all have the max distance FLT_MAX: Streams::o12)
	{
		Timed t(h, "generate", q);
		if (h->opt_time_kernels) h->timer_symbol["generate"] = "pol::k_generate";
		hipLaunchKernelGGL(k_generate, dim3(wgs), dim3(WG), 0, q, P.st, h->cam, h->d_seeds, stride, s0, N, Npad, h->W, r->block_y,
		                   exact ? 0 : 1, (B > 0 && (h->packet_primary || h->opt_traversal)) ? 0 : 1); // (neither the wave-packet kernel nor k_trace's camera launch reads the origin stream)
	}
	ShadeArgs A{};
	A.seeds = h->d_seeds; A.seed_stride = stride; A.first_sample = s0;
	A.N = N; A.Npad = Npad; A.W = h->W; A.blockY = r->block_y;
	A.min_rr = r->min_bounces_for_rr;
	A.exact = exact ? 1 : 0;
	// LDS-table variant of the shade kernels: only when all three tables fit (kernels.h, stage_scene)
	const bool staged = h->opt_stage_lds && h->scene.num_nodes <= kLdsMatNodes && h->scene.num_emissives <= kLdsLights &&
	                    h->scene.num_textures <= kLdsTextures;
	A.acc = exact ? h->trace_acc : P.st.lsum;
	// persistent grid: chunks are dealt to workgroups statically, so a workgroup that is not resident
	// from the start begins with its whole share still to do -- the grid must not exceed what the GPU
	// holds at once (measured: 8 workgroups per CU where 6 fit cost the closest-hit kernel 12 %).  One
	// batch at a time: exactly the resident capacity.  Several batches in flight: two thirds of it per
	// launch (the launches share the CUs; measured flat to +5 % across the bench scenes).
	// (LDS: 16-entry stack: 16 KB per
	// workgroup -> 8 by LDS, VGPRs allow 7-8 waves/SIMD; 32-entry: 5)
	auto grid_of = [&](int resident) {
		uint32_t per_cu = (uint32_t)std::max(1, resident);
		if (std::min(h->opt_overlap, (int)polaris_hip_tracer::kMaxPipes) > 1 && !exact && per_cu > 2) per_cu = std::max(2u, per_cu * 2u / 3u);
		// tiny-scene mode (two 1 024-thread workgroups fill a CU), small launches -- a row block of a multi-GPU frame: with two
		// batches in flight a launch gets half the GPU at best, and at fewer than 4 chunks per resident wave a full-size grid
		// leaves every wave a single chunk, i.e. nothing but its ramp-down.  One workgroup per CU then: a 64-row block of the
		// headline frame 2.10 -> 1.83 ms, a 128-row block 3.30 -> 3.13 ms (256 rows +-0, the full frame +1 %: not applied there)
		if (h->node_mode == kNodesLdsAll && std::min(h->opt_overlap, (int)polaris_hip_tracer::kMaxPipes) > 1 && !exact && per_cu == 2 &&
		    (uint64_t)wgs < 4ull * (uint64_t)h->num_cus * per_cu * (kTinyBlock / 64))
			per_cu = 1;
		if (h->opt_trace_wgs_per_cu > 0) per_cu = (uint32_t)h->opt_trace_wgs_per_cu;
		if (h->opt_trace_grid > 0) return std::min<uint32_t>(wgs, (uint32_t)h->opt_trace_grid); // (A/B aid: an absolute persistent grid)
		return std::min<uint32_t>(wgs, (uint32_t)h->num_cus * per_cu);
	};
	const uint32_t persistent = grid_of(h->trace_resident_per_cu), persistent_occl = grid_of(h->occl_resident_per_cu);
	for (uint32_t b = 0; b < B; b++) {
		Streams S = P.st; // (in place: k_shade / k_shade_wave read a chunk's rays before they write into it)
		if (!exact) { S.occ_e = P.nee[b]; S.vis = P.vis[b]; S.cnt_occ = P.cnt_occ_b[b]; } // this bounce's NEE records, visibility bytes and shadow-ray counts stay until k_fold_nee
		float4 *const occl_acc = exact ? A.acc : nullptr;                // batched: unoccluded rays only mark their record (deferred)
		{
			Timed t(h, (b == 0 && h->packet_primary) ? "intersect_packet" : "intersect", q);
			if (h->opt_time_kernels) {
				if (b == 0 && h->packet_primary) h->timer_symbol["intersect_packet"] = "pol::k_trace_packet<false, true>";
				else h->timer_symbol["intersect"] = h->opt_traversal ? trace_symbol<false>(h) : std::string("pol::k_intersect");
			}
			if (b == 0 && h->packet_primary)
				hipLaunchKernelGGL((k_trace_packet<false, true>), dim3(wgs), dim3(WG), 0, q, S, h->bvh, (float4 *)nullptr, h->d_stats, h->cam.eye);
			else if (h->opt_traversal)
				note(launch_trace<false>(h, P, S, persistent, wgs, nullptr, b == 0));
			else
				hipLaunchKernelGGL(k_intersect, dim3(wgs), dim3(WG), 0, q, S, h->bvh);
		}
		A.bounce = b;
		A.last_bounce = (b + 1 == B) ? 1 : 0;
		A.emask_in = b == 0 ? nullptr : P.st.emask[(b + 1) & 1]; // the masks the previous step wrote
		A.emask_out = P.st.emask[b & 1];
		{
			// one timer per kernel symbol: shade_first = k_shade<.., FIRST> (camera rays), shade_sort = k_shade<.., SORT, ..> (bounce
			// rays in class order), shade_plain = k_shade<.., false, false>, shade_wave = k_shade_wave
			const bool wave = h->opt_shade_wave && b > 0 && (int)b >= (h->opt_shade_wave_from >= 0 ? h->opt_shade_wave_from : (int)r->min_bounces_for_rr + 1); // (never the first bounce: k_shade_wave reads the previous step's emit masks)
			// bounce rays are shaded in the order of their material's shading class (kernels.h, k_shade SORT); camera rays are
			// coherent as they come (64 neighbouring pixels per wave)
			const bool sorted = b > 0 && h->scene.tri_bits < 31 && (int)b >= (h->opt_shade_sort >= 0 ? h->opt_shade_sort : 1);
			const int which = wave ? 3 : (b == 0 ? 0 : (sorted ? 1 : 2));
			static const char *const kShadeTimer[4] = {"shade_first", "shade_sort", "shade_plain", "shade_wave"};
			h->last_shade_timer[b] = which;
			Timed t(h, kShadeTimer[which], q);
			if (h->opt_time_kernels) {
				const char *l = staged ? "true" : "false";
				char buf[64];
				if (wave) snprintf(buf, sizeof buf, "pol::k_shade_wave<%s>", l);
				else snprintf(buf, sizeof buf, "pol::k_shade<%s, %s, %s>", l, which == 1 ? "true" : "false", which == 0 ? "true" : "false");
				h->timer_symbol[kShadeTimer[which]] = buf;
			}
			if (wave) {
				// persistent waves pull groups of kSparseGroup chunks: no more workgroups than the GPU holds at once (4 per CU at
				// the kernel's register count) nor than there are groups for their 4 waves
				const uint32_t groups = (wgs + kSparseGroup - 1) / kSparseGroup;
				const uint32_t grid = std::max(1u, std::min<uint32_t>((groups + 3) / 4, (uint32_t)h->num_cus * (uint32_t)std::max(1, h->opt_shade_wgs_per_cu)));
				if (staged) hipLaunchKernelGGL(k_shade_wave<true>, dim3(grid), dim3(WG), 0, q, S, h->scene, A, wgs);
				else hipLaunchKernelGGL(k_shade_wave<false>, dim3(grid), dim3(WG), 0, q, S, h->scene, A, wgs);
			} else {
				const void *fn;
				if (b == 0) fn = staged ? (const void *)k_shade<true, false, true> : (const void *)k_shade<false, false, true>;
				else if (sorted) fn = staged ? (const void *)k_shade<true, true, false> : (const void *)k_shade<false, true, false>;
				else fn = staged ? (const void *)k_shade<true, false, false> : (const void *)k_shade<false, false, false>;
				void *args[] = {(void *)&S, (void *)&h->scene, (void *)&A};
				note(hipLaunchKernel(fn, dim3(wgs), dim3(WG), args, 0, q));
			}
		}
		{
			Timed t(h, "scan", q);
			hipLaunchKernelGGL(k_scan, dim3(K), dim3(1024), 0, q, S, wgs_per_sample, b, A.last_bounce ? 0 : 1, h->d_stats);
		}
		{
			Timed t(h, "occlusion", q);
			if (h->opt_time_kernels) h->timer_symbol["occlusion"] = (int)b < h->opt_packet_shadow ? std::string("pol::k_trace_packet<true, false>") : (h->opt_traversal ? trace_symbol<true>(h) : std::string("pol::k_occlusion"));
			if ((int)b < h->opt_packet_shadow)
				hipLaunchKernelGGL(k_trace_packet<true>, dim3(wgs), dim3(WG), 0, q, S, h->bvh, occl_acc, h->d_stats, h->cam.eye);
			else if (h->opt_traversal)
				note(launch_trace<true>(h, P, S, persistent_occl, wgs, occl_acc));
			else
				hipLaunchKernelGGL(k_occlusion, dim3(wgs), dim3(WG), 0, q, S, h->bvh, occl_acc, h->d_stats);
		}
	}
	if (!exact && B > 0) { // accumulateEmissiveSamples of the whole batch: the marked NEE records of every bounce into the per-path radiance
		FoldArgs F{};
		for (uint32_t b = 0; b < B; b++) { F.nee[b] = P.nee[b]; F.vis[b] = P.vis[b]; F.cnt[b] = P.cnt_occ_b[b]; }
		F.bounces = B;
		Timed t(h, "fold", q);
		if (h->opt_time_kernels) h->timer_symbol["fold"] = "pol::k_fold_nee";
		hipLaunchKernelGGL(k_fold_nee, dim3(wgs), dim3(WG), 0, q, F, P.st.lsum);
	}
	if (!exact) {
		// batches resolve into the trace accumulator in sample order: wait for the previous batch's resolve
		if (resolve_after) note(hipStreamWaitEvent(q, resolve_after, 0));
		Timed t(h, "resolve", q);
		hipLaunchKernelGGL(k_resolve, dim3(grid_for(N)), dim3(WG), 0, q, P.st.lsum, h->trace_acc, K, N, Npad, r->block_y * h->W);
	}
	note(hipEventRecord(P.done, q));
	note(hipGetLastError()); // (launches through hipLaunchKernelGGL report here)
	return first_err;
}

} // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

int polaris_hip_abi_version(void) { return POLARIS_HIP_ABI_VERSION; }

int polaris_hip_device_count(void) {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

int polaris_hip_device_info(int index, char name[256], uint32_t *compute_units, uint32_t *clock_mhz, uint64_t *global_mem_bytes) {
	int n = polaris_hip_device_count();
	if (index < 0 || index >= n) return fail(nullptr, POLARIS_E_NO_DEVICE, "device %d out of range (%d devices)", index, n);
	hipDeviceProp_t p;
	HIP_TRY(nullptr, hipGetDeviceProperties(&p, index));
	if (name) { strncpy(name, p.name, 255); name[255] = 0; }
	if (compute_units) *compute_units = (uint32_t)p.multiProcessorCount;
	if (clock_mhz) *clock_mhz = (uint32_t)(p.clockRate / 1000);
	if (global_mem_bytes) *global_mem_bytes = (uint64_t)p.totalGlobalMem;
	return POLARIS_OK;
}

int polaris_hip_device_identity(int index, PolarisDeviceIdentity *out) {
	if (!out) return fail(nullptr, POLARIS_E_BAD_ARGUMENT, "device_identity: out is null");
	if (out->struct_size != sizeof(PolarisDeviceIdentity))
		return fail(nullptr, POLARIS_E_BAD_ARGUMENT, "device_identity: struct_size %u, this library's PolarisDeviceIdentity has %zu bytes", out->struct_size, sizeof(PolarisDeviceIdentity));
	const int n = polaris_hip_device_count();
	if (index < 0 || index >= n) return fail(nullptr, POLARIS_E_NO_DEVICE, "device %d out of range (%d devices)", index, n);
	hipDeviceProp_t p;
	HIP_TRY(nullptr, hipGetDeviceProperties(&p, index));
	memset(out, 0, sizeof *out);
	out->struct_size = sizeof *out;
	out->hip_index = index;
	if (hipDeviceGetPCIBusId(out->pci_bus_id, (int)sizeof out->pci_bus_id, index) != hipSuccess) { (void)hipGetLastError(); out->pci_bus_id[0] = 0; }
	hipUUID u;
	if (hipDeviceGetUuid(&u, index) == hipSuccess) memcpy(out->uuid, u.bytes, 16);
	else (void)hipGetLastError();
	out->compute_units = (uint32_t)p.multiProcessorCount;
	out->clock_mhz = (uint32_t)(p.clockRate / 1000);
	out->global_mem_bytes = (uint64_t)p.totalGlobalMem;
	snprintf(out->name, sizeof out->name, "%s", p.name);
	snprintf(out->gcn_arch, sizeof out->gcn_arch, "%s", p.gcnArchName);
	return POLARIS_OK;
}

int polaris_hip_can_access_peer(int device, int peer_device, int *can) {
	if (!can) return fail(nullptr, POLARIS_E_BAD_ARGUMENT, "can_access_peer: out is null");
	*can = 0;
	const int n = polaris_hip_device_count();
	if (device < 0 || device >= n || peer_device < 0 || peer_device >= n) return fail(nullptr, POLARIS_E_NO_DEVICE, "can_access_peer: device %d / %d out of range (%d devices)", device, peer_device, n);
	if (device == peer_device) return POLARIS_OK;
	HIP_TRY(nullptr, hipDeviceCanAccessPeer(can, device, peer_device));
	return POLARIS_OK;
}

int polaris_hip_create(int device_index, polaris_hip_tracer **out) {
	if (!out) return fail(nullptr, POLARIS_E_BAD_ARGUMENT, "out handle pointer is null");
	*out = nullptr;
	int n = polaris_hip_device_count();
	if (device_index < 0 || device_index >= n)
		return fail(nullptr, POLARIS_E_NO_DEVICE, "device %d out of range (%d HIP devices visible)", device_index, n);
	polaris_hip_tracer *h = new polaris_hip_tracer();
	h->device = device_index;
	if (hipDeviceGetPCIBusId(h->pci_bus_id, (int)sizeof h->pci_bus_id, device_index) != hipSuccess) { (void)hipGetLastError(); h->pci_bus_id[0] = 0; }
	hipError_t e = hipSetDevice(device_index);
	if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
	h->pipe[0].q = h->stream;
	for (int p = 1; p < polaris_hip_tracer::kMaxPipes && e == hipSuccess; p++) e = hipStreamCreateWithFlags(&h->pipe[p].q, hipStreamNonBlocking);
	for (int p = 0; p < polaris_hip_tracer::kMaxPipes && e == hipSuccess; p++) e = hipEventCreateWithFlags(&h->pipe[p].done, hipEventDisableTiming);
	if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming);
	if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->merge_stream, hipStreamNonBlocking);
	if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_merged, hipEventDisableTiming);
	if (e == hipSuccess) e = hipMalloc((void **)&h->d_stats, ST_COUNT * sizeof(unsigned long long));
	if (e == hipSuccess) e = hipMalloc((void **)&h->d_cam_o, sizeof(float4));
	if (e == hipSuccess) {
		hipDeviceProp_t p;
		if (hipGetDeviceProperties(&p, device_index) == hipSuccess && p.multiProcessorCount > 0) h->num_cus = p.multiProcessorCount;
	}
	if (e == hipSuccess) e = hipEventCreate(&h->ev_start);
	if (e == hipSuccess) e = hipEventCreate(&h->ev_stop);
	if (e != hipSuccess) {
		fail(nullptr, POLARIS_E_DEVICE, "creating tracer on device %d: %s", device_index, hipGetErrorString(e));
		delete h;
		return POLARIS_E_DEVICE;
	}
	*out = h;
	return POLARIS_OK;
}

void polaris_hip_destroy(polaris_hip_tracer *h) {
	if (!h) return;
	{
		std::lock_guard<std::mutex> lk(h->mu);
		(void)hipSetDevice(h->device);
		if (h->stream) (void)hipStreamSynchronize(h->stream);
		for (int p = 1; p < polaris_hip_tracer::kMaxPipes; p++)
			if (h->pipe[p].q) (void)hipStreamSynchronize(h->pipe[p].q);
		if (h->merge_stream) (void)hipStreamSynchronize(h->merge_stream);
		collect_timers(h);
		std::lock_guard<std::mutex> lk_merge(h->merge_mu);
		if (h->merge_stream) (void)hipStreamDestroy(h->merge_stream);
		if (h->ev_merged) (void)hipEventDestroy(h->ev_merged);
		for (auto e : h->event_pool) (void)hipEventDestroy(e);
		for (auto &P : h->pipe) {
			free_pool(P.bufs);
			if (P.done) (void)hipEventDestroy(P.done);
		}
		if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
		for (int p = 1; p < polaris_hip_tracer::kMaxPipes; p++)
			if (h->pipe[p].q) (void)hipStreamDestroy(h->pipe[p].q);
		free_pool(h->scene_bufs);
		free_ring(h);
		for (auto &e : h->ev_ipc_done)
			if (e) { (void)hipEventDestroy(e); e = nullptr; }
		{
			std::lock_guard<std::mutex> lk_r(h->readers_mu);
			for (auto &r : h->readers) (void)hipEventDestroy(r.ev);
			for (auto &r : h->reader_pool) (void)hipEventDestroy(r.ev);
			h->readers.clear();
			h->reader_pool.clear();
		}
		if (h->frame_acc) (void)hipFree(h->frame_acc);
		if (h->framebuffer) (void)hipFree(h->framebuffer);
		if (h->d_seeds) (void)hipFree(h->d_seeds);
		if (h->d_stats) (void)hipFree(h->d_stats);
		if (h->d_cam_o) (void)hipFree(h->d_cam_o);
		if (h->staging) (void)hipFree(h->staging);
		if (h->ev_start) (void)hipEventDestroy(h->ev_start);
		if (h->ev_stop) (void)hipEventDestroy(h->ev_stop);
		if (h->stream) (void)hipStreamDestroy(h->stream);
	}
	delete h;
}

const char *polaris_hip_last_error(polaris_hip_tracer *h) {
	if (!h) return g_thread_error.c_str();
	thread_local std::string copy; // a concurrent call may be failing on the same handle: read under the locks
	uint64_t seq;
	{
		std::lock_guard<std::mutex> lk(h->mu);
		copy = h->error;
		seq = h->error_seq;
	}
	{
		std::lock_guard<std::mutex> lk(h->merge_mu);
		if (h->merge_error_seq > seq) copy = h->merge_error; // the newer of the two
	}
	return copy.c_str();
}

int polaris_hip_resize(polaris_hip_tracer *h, uint32_t frame_w, uint32_t frame_h) {
	if (!h) return fail(nullptr, POLARIS_E_BAD_ARGUMENT, "handle is null");
	std::lock_guard<std::mutex> lk(h->mu);
	if (frame_w == 0 || frame_h == 0 || (uint64_t)frame_w * frame_h > (1ull << 28))
		return fail(h, POLARIS_E_BAD_ARGUMENT, "bad frame dimensions %ux%u", frame_w, frame_h);
	HIP_TRY(h, hipSetDevice(h->device));
	std::lock_guard<std::mutex> lk_merge(h->merge_mu); // (the frame accumulator is the merge stream's)
	HIP_TRY(h, sync_all(h));
	free_ring(h); // (an IPC export dies with the buffers: peers close, the tracer exports again)
	if (h->frame_acc) (void)hipFree(h->frame_acc);
	if (h->framebuffer) (void)hipFree(h->framebuffer);
	h->frame_acc = nullptr;
	h->framebuffer = nullptr;
	h->W = h->H = 0;
	const size_t F = (size_t)frame_w * frame_h;
	HIP_TRY(h, hipMalloc((void **)&h->ring[0], F * sizeof(float4)));
	h->trace_acc = h->ring[0];
	HIP_TRY(h, hipMalloc((void **)&h->frame_acc, F * sizeof(float4)));
	HIP_TRY(h, hipMalloc((void **)&h->framebuffer, F * sizeof(uchar4)));
	HIP_TRY(h, hipMemsetAsync(h->trace_acc, 0, F * sizeof(float4), h->stream));
	HIP_TRY(h, hipMemsetAsync(h->frame_acc, 0, F * sizeof(float4), h->stream));
	HIP_TRY(h, hipMemsetAsync(h->framebuffer, 0, F * sizeof(uchar4), h->stream));
	HIP_TRY(h, hipStreamSynchronize(h->stream));
	h->W = frame_w;
	h->H = frame_h;
	return POLARIS_OK;
}

int polaris_hip_upload_scene(polaris_hip_tracer *h, const PolarisSceneView *sc) {
	if (!h) return fail(nullptr, POLARIS_E_BAD_ARGUMENT, "handle is null");
	std::lock_guard<std::mutex> lk(h->mu);
	if (!sc) return fail(h, POLARIS_E_BAD_ARGUMENT, "scene view is null");
	SceneLayout L;
	// Defaults by scene size (measured): small scenes live in L2/LDS and are bound by
	// instruction issue -> small leaves and packet traversal of camera rays pay; scenes far larger than
	// the caches are bound by node fetches -> fewer, fuller leaves and one ray per lane.
	const int max_leaf = h->opt_max_leaf_tris >= 0 ? h->opt_max_leaf_tris : (sc->num_triangles <= 32768u ? 2 : 4);
	std::string err = build_layout(*sc, L, max_leaf);
	if (err == "@retry-without-subdivision") { // the deeper tree would not fit the traversal stack
		L = SceneLayout();
		err = build_layout(*sc, L, 0);
	}
	if (!err.empty()) return fail(h, POLARIS_E_BAD_SCENE, "%s", err.c_str());
	HIP_TRY(h, hipSetDevice(h->device));
	HIP_TRY(h, sync_all(h));
	free_pool(h->scene_bufs);
	h->have_scene = false;
	int rc = 0;
	PairNode *pairs; int2 *leaves; TriRec *tris; InstRec *insts;
	rc |= dev_upload(h, h->scene_bufs, &pairs, L.pairs.data(), L.pairs.size());
	rc |= dev_upload(h, h->scene_bufs, &leaves, L.leaves.data(), L.leaves.size());
	rc |= dev_upload(h, h->scene_bufs, &tris, L.tris.data(), L.tris.size());
	rc |= dev_upload(h, h->scene_bufs, &insts, L.insts.data(), L.insts.size());
	float4 *vertices, *normals; float2 *uvs; uint32_t *mat_index;
	PolarisMaterialNode *nodes; PolarisEmissive *emissives; PolarisTextureMetadata *tex_meta; uint8_t *tex_data; float *light_geo;
	// the triangle of every area light, packed (shading.h, SceneT::light_geo; indices were range-checked by build_layout)
	std::vector<float> geo((size_t)sc->num_emissives * kLightGeoFloats, 0.0f);
	for (uint32_t e = 0; e < sc->num_emissives; e++) {
		if (sc->emissives[e].type != POLARIS_EMISSIVE_AREA) continue;
		const size_t off = (size_t)sc->emissives[e].tri_index * 3;
		float *g = geo.data() + (size_t)e * kLightGeoFloats;
		for (int v = 0; v < 3; v++) {
			for (int k = 0; k < 3; k++) { g[4 * v + k] = sc->vertices[4 * (off + v) + k]; g[12 + 4 * v + k] = sc->normals[4 * (off + v) + k]; }
			g[24 + 2 * v] = sc->uvs[2 * (off + v)]; g[25 + 2 * v] = sc->uvs[2 * (off + v) + 1];
		}
		// what areaLightGetPdf (emissive_sampler.cl:117-173) and areaLightGetSample (:96) derive from the light alone, with the
		// kernels' own operations in the kernels' order (this file is compiled without FMA contraction; polaris_math.h is
		// bit-identical on the host): v0, v1 - v0, v2 - v0 through mul4x1 (util/transform.cl:9-16), normalize(cross(e1, e2)), 1 / area
		const float *m = sc->emissives[e].transform;
		auto xform = [&](const float p[3], float out[3]) {
			out[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
			out[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
			out[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
		};
		const float v0[3] = {g[0], g[1], g[2]};
		const float d1[3] = {g[4] - v0[0], g[5] - v0[1], g[6] - v0[2]}, d2[3] = {g[8] - v0[0], g[9] - v0[1], g[10] - v0[2]};
		float tv0[3], te1[3], te2[3];
		xform(v0, tv0); xform(d1, te1); xform(d2, te2);
		const float cx = te1[1] * te2[2] - te1[2] * te2[1], cy = te1[2] * te2[0] - te1[0] * te2[2], cz = te1[0] * te2[1] - te1[1] * te2[0];
		const float inv_len = 1.0f / pm_sqrt(cx * cx + cy * cy + cz * cz);
		for (int k = 0; k < 3; k++) { g[32 + k] = tv0[k]; g[36 + k] = te1[k]; g[40 + k] = te2[k]; }
		g[44] = cx * inv_len; g[45] = cy * inv_len; g[46] = cz * inv_len;
		g[30] = 1.0f / sc->emissives[e].area;
	}
	const size_t nv = (size_t)sc->num_triangles * 3;
	rc |= dev_upload(h, h->scene_bufs, &vertices, sc->vertices, nv);
	rc |= dev_upload(h, h->scene_bufs, &normals, sc->normals, nv);
	rc |= dev_upload(h, h->scene_bufs, &uvs, sc->uvs, nv);
	rc |= dev_upload(h, h->scene_bufs, &mat_index, sc->material_index, sc->num_triangles);
	rc |= dev_upload(h, h->scene_bufs, &nodes, sc->material_nodes, sc->num_material_nodes);
	rc |= dev_upload(h, h->scene_bufs, &emissives, sc->emissives, sc->num_emissives);
	rc |= dev_upload(h, h->scene_bufs, &tex_meta, sc->texture_meta, sc->num_textures);
	rc |= dev_upload(h, h->scene_bufs, &light_geo, geo.data(), geo.size());
	// the texture blob, padded: texels are fetched as the three dwords at their address whatever the format (shading.h, tex_fetch)
	rc |= dev_alloc(h, h->scene_bufs, &tex_data, (size_t)sc->texture_data_bytes + 16);
	if (!rc && sc->texture_data_bytes) HIP_TRY(h, hipMemcpyAsync(tex_data, sc->texture_data, sc->texture_data_bytes, hipMemcpyHostToDevice, h->stream));
	if (rc) { free_pool(h->scene_bufs); return rc; }
	HIP_TRY(h, hipStreamSynchronize(h->stream)); // host vectors in L die at return
	h->bvh = BvhDev{pairs, (uint32_t)L.pairs.size(), leaves, tris, insts, L.root_ref, 0, InstRec{}, 0, 0};
	if (L.root_ref < 0 && (((uint32_t)~L.root_ref) & 15u) == 0u && !(((uint32_t)~L.root_ref) & kBigLeafFlag)) { // the top-level tree is a single leaf ...
		const size_t root_inst = ((uint32_t)~L.root_ref) >> 4; // (... which, in the top-level tree, is an instance: its id is in the reference)
		if (root_inst < L.insts.size()) { // its record goes with the kernel arguments
			const InstH &I = L.insts[root_inst];
			h->bvh.root_is_instance = 1;
			h->bvh.root_inst.r0 = make_float4(I.r0[0], I.r0[1], I.r0[2], I.r0[3]);
			h->bvh.root_inst.r1 = make_float4(I.r1[0], I.r1[1], I.r1[2], I.r1[3]);
			h->bvh.root_inst.r2 = make_float4(I.r2[0], I.r2[1], I.r2[2], I.r2[3]);
			h->bvh.root_inst.meta = make_int4(I.root_ref, (int)I.rank, (int)I.pad[0], 0);
		}
	}
	h->scene = SceneDev{vertices, normals, uvs, mat_index, nodes, emissives, tex_meta, tex_data, sc->num_emissives,
	                    sc->scene_diffuse_mat_index, sc->num_material_nodes, sc->num_textures, light_geo, sc->num_emissives ? pm_rcp((float)(int)sc->num_emissives) : 0.0f, L.tri_bits};
	h->max_stack = L.max_stack;
	h->tex_bytes = sc->texture_data_bytes;
	// camera rays: the wave-packet kernel where a packet stays together AND the tree is small -- one instance, up to 32 K
	// triangles.  Since the per-ray kernel's instruction diet the two are level on the tiny scenes (headline 11.37 /
	// 11.31 / 11.39 ms with packets against 11.42 / 11.32 / 11.38 without, round 4: the packet kernel stays there, it keeps the
	// origin stream unwritten); on the 58 K-triangle ball the packet's dependent scalar node fetches cost more than the per-ray
	// kernel's gathers (camera rays 5.5 vs 4.1 ms per 32 spp, frame -1.8 %); in a scene of many instances the packet's lanes part
	// ways inside the instances (-5 % frame time per-ray on the 1 024-instance scene, -26 % on the 1 M-triangle terrain)
	// (round 4, later: with the triangle records in LDS the per-ray kernel of the tiny-scene mode is ahead -- camera rays 0.93 vs
	// 1.20 ms isolated, frame 10.96 vs 11.03 ms, three alternating runs: see below, after the node mode is known)
	if (h->opt_packet_primary < 0) h->packet_primary = sc->num_triangles <= 32768u && sc->num_mesh_instances == 1;
	// node records: whole tree in LDS for tiny scenes, its top for small ones, global memory otherwise (kernels.h NodeMode)
	// (16-bit stack entries: triangle slots and instance ids must fit 11 bits, and no leaf reference may carry kBigLeafFlag)
	// (... and the packed word of a slot holds the scene triangle in 11 bits too: scene_layout.h tiny_meta_word)
	const bool tiny_ok = L.pairs.size() <= (size_t)kTinyPairs && L.tris.size() <= kTinyMaxIndex && sc->num_triangles <= kTinyMaxIndex && L.insts.size() <= kTinyMaxIndex &&
	                     L.big_leaves == 0 && L.max_stack <= 16;
	h->node_mode = tiny_ok ? kNodesLdsAll : (L.pairs.size() <= (size_t)(POLARIS_LDS_TOP_MAX_PAIRS) ? kNodesLdsTop : kNodesGlobal);
	if (h->opt_node_mode >= 0 && (h->opt_node_mode != kNodesLdsAll || tiny_ok)) h->node_mode = h->opt_node_mode;
	h->tiny_lds_bytes = 0;
	h->tiny_one = h->node_mode == kNodesLdsAll && h->opt_tiny_one && h->bvh.root_is_instance && L.unbounded_boxes == 0;
	if (h->node_mode == kNodesLdsAll) {
		if (int prc = plan_tiny_lds(h, L.tris.size())) return prc;
		if (h->opt_packet_primary < 0 && h->bvh.lds_tris > 0) h->packet_primary = false;
	}
	h->trace_resident_per_cu = trace_occupancy<false>(h);
	h->occl_resident_per_cu = trace_occupancy<true>(h);
	h->have_scene = true;
	return POLARIS_OK;
}

int polaris_hip_set_camera(polaris_hip_tracer *h, const float eye[3], const float fr[16]) {
	if (!h) return fail(nullptr, POLARIS_E_BAD_ARGUMENT, "handle is null");
	std::lock_guard<std::mutex> lk(h->mu);
	if (!eye || !fr) return fail(h, POLARIS_E_BAD_ARGUMENT, "camera pointers are null");
	h->cam.tl = make_float4(fr[0], fr[1], fr[2], fr[3]);
	h->cam.tr = make_float4(fr[4], fr[5], fr[6], fr[7]);
	h->cam.bl = make_float4(fr[8], fr[9], fr[10], fr[11]);
	h->cam.br = make_float4(fr[12], fr[13], fr[14], fr[15]);
	h->cam.eye = make_float3(eye[0], eye[1], eye[2]);
	const float4 o4 = make_float4(eye[0], eye[1], eye[2], kFltMax); // what k_generate would write per camera ray (kernels.h)
	HIP_TRY(h, hipSetDevice(h->device));
	// (nothing of this handle reads the record now: the caller holds mu, and a Trace returns only when its kernels are done)
	HIP_TRY(h, hipMemcpy(h->d_cam_o, &o4, sizeof o4, hipMemcpyHostToDevice));
	h->have_camera = true;
	return POLARIS_OK;
}

int polaris_hip_set_option(polaris_hip_tracer *h, const char *key, int64_t value) {
	if (!h) return fail(nullptr, POLARIS_E_BAD_ARGUMENT, "handle is null");
	std::lock_guard<std::mutex> lk(h->mu);
	if (!key) return fail(h, POLARIS_E_BAD_ARGUMENT, "option key is null");
	const std::string k(key);
	if (k == "samples_per_batch") h->opt_samples_per_batch = value < 0 ? 0 : value;
	else if (k == "exact_accumulate") h->opt_exact = value != 0;
	else if (k == "packet_primary") { h->opt_packet_primary = value < 0 ? -1 : (value != 0); if (value >= 0) h->packet_primary = value != 0; }
	else if (k == "packet_shadow") h->opt_packet_shadow = (int)std::max<int64_t>(0, std::min<int64_t>(value, POLARIS_MAX_BOUNCES));
	else if (k == "time_kernels") h->opt_time_kernels = value != 0;
	else if (k == "node_mode") h->opt_node_mode = (int)std::max<int64_t>(-1, std::min<int64_t>(value, 2)); // next upload; 2 only where the scene is tiny enough
	else if (k == "traversal") h->opt_traversal = value != 0; // 0 = one ray per lane (k_intersect / k_occlusion), 1 = persistent waves with lane refill (k_trace)
	else if (k == "shade_wave") h->opt_shade_wave = value != 0;
	else if (k == "shade_wave_from") h->opt_shade_wave_from = (int)std::max<int64_t>(-1, std::min<int64_t>(value, POLARIS_MAX_BOUNCES));
	else if (k == "shade_wgs_per_cu") h->opt_shade_wgs_per_cu = (int)std::max<int64_t>(1, std::min<int64_t>(value, 64));
	else if (k == "stage_lds") h->opt_stage_lds = value != 0;
	else if (k == "shade_sort") h->opt_shade_sort = (int)std::max<int64_t>(-1, std::min<int64_t>(value, POLARIS_MAX_BOUNCES));
	else if (k == "ipc_staged") h->opt_ipc_staged = value != 0;
	else if (k == "overlap") h->opt_overlap = (int)std::max<int64_t>(1, std::min<int64_t>(value, polaris_hip_tracer::kMaxPipes));
	else if (k == "trace_grid") h->opt_trace_grid = (int)std::max<int64_t>(0, std::min<int64_t>(value, 1 << 20));
	else if (k == "trace_wgs_per_cu") h->opt_trace_wgs_per_cu = (int)std::max<int64_t>(0, std::min<int64_t>(value, 64));
	else if (k == "tiny_one") h->opt_tiny_one = value != 0; // next upload
	else if (k == "hit12") h->opt_hit12 = value != 0;
	else if (k == "o12") h->opt_o12 = value != 0;
	else if (k == "lds_tris") h->opt_lds_tris = (int)std::max<int64_t>(-1, std::min<int64_t>(value, 1 << 20)); // next upload
	else if (k == "max_leaf_tris") h->opt_max_leaf_tris = (int)std::max<int64_t>(-1, std::min<int64_t>(value, 1 << 20)); // next upload
	else return fail(h, POLARIS_E_BAD_ARGUMENT, "unknown option '%s'", key);
	return POLARIS_OK;
}

int polaris_hip_trace(polaris_hip_tracer *h, const PolarisBlockRequest *r, const uint32_t *seeds, size_t n_seeds,
                      PolarisTraceStats *stats) {
	if (!h) return fail(nullptr, POLARIS_E_BAD_ARGUMENT, "handle is null");
	std::lock_guard<std::mutex> lk(h->mu);
	// a Trace that resets the frame announces it (reset epoch) once the clear is queued -- or when it fails before that, so
	// that nobody waits for a reset that will not come
	struct ResetAnnounce {
		polaris_hip_tracer *h; bool due, done = false;
		void now() { if (due && !done) { done = true; { std::lock_guard<std::mutex> lk(h->merge_mu); h->reset_epoch++; } h->reset_cv.notify_all(); } }
		~ResetAnnounce() { now(); }
	} announce{h, r && r->accumulated_samples == 0};
	if (!h->have_scene) return fail(h, POLARIS_E_NO_SCENE_DATA, "no scene data uploaded"); // ErrNoSceneData, tracer.go:203-205
	if (int rc = check_request(h, r)) return rc;
	if (!h->have_camera) return fail(h, POLARIS_E_BAD_ARGUMENT, "camera not set (UpdateState CameraData)");
	const uint32_t B = r->num_bounces, spp = r->samples_per_pixel;
	if (B > POLARIS_MAX_BOUNCES) return fail(h, POLARIS_E_BAD_ARGUMENT, "num_bounces %u exceeds %d", B, POLARIS_MAX_BOUNCES);
	const size_t need_seeds = (size_t)spp * (1 + B);
	if (need_seeds && (!seeds || n_seeds < need_seeds))
		return fail(h, POLARIS_E_BAD_ARGUMENT, "seed list has %zu entries, need samples*(1+bounces) = %zu", n_seeds, need_seeds);
	const uint64_t N64 = (uint64_t)h->W * r->block_h;
	if (N64 > (1u << 24)) // the reference stores the path index as a float in ray.dir.w (util/ray.cl:9-12)
		return fail(h, POLARIS_E_BAD_ARGUMENT, "block has %llu pixels; the path index is exact only below 2^24", (unsigned long long)N64);
	const uint32_t N = (uint32_t)N64, Npad = (N + WG - 1) / WG * WG;

	HIP_TRY(h, hipSetDevice(h->device));
	const bool exact = h->opt_exact != 0;
	uint32_t K = 1;
	if (!exact) {
		// Up to ~32 M paths per batch, but at least two batches so that one's sparse late bounces run beside the other's
		// dense early ones.  Bigger batches amortise the tail of every persistent launch (a workgroup finishes the longest
		// of its last rays alone) over more chunks -- measured on the round-2 kernels: headline frame 2 x 16.8 M paths
		// 17.5 ms, 4 x 8.4 M 18.3 ms, 8 x 4.2 M 19.8 ms, 1 x 33.5 M 18.8 ms; 1024^2 x 256 spp: 33.5 M per batch 130.7 ms,
		// 16.8 M 134.9 ms, 8.4 M 142.4 ms.  (128 B of stream buffers per path: 4.3 GB per batch in flight, of 288 GB.)
		if (h->opt_samples_per_batch > 0) K = (uint32_t)std::min<int64_t>(h->opt_samples_per_batch, 4096);
		else K = std::min<uint32_t>(std::max<uint32_t>(1u, (uint32_t)((32u << 20) / Npad)), std::max(1u, (spp + 1) / 2));
		K = std::max<uint32_t>(1u, std::min(K, std::max(spp, 1u)));
	}
	// The batch buffers are 128 bytes per path slot + 16 per bounce (the NEE records are kept per bounce until the batch folds them) and up to `overlap`
	// batches are in flight: 4.3 GB per pipeline at 33.5 M slots, nothing on a 288 GB MI355X but not on a smaller or shared device.  K chosen automatically is first clamped by the
	// free device memory and, if an allocation still fails, halved and retried (a caller-chosen samples_per_batch is kept as it
	// is: its failure is reported).
	if (!exact && h->opt_samples_per_batch <= 0 && K > 1) {
		size_t free_b = 0, total_b = 0;
		if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
			size_t have = 0; // what the pipelines already hold counts as available
			const size_t slot_bytes = 144 + 17 * (size_t)std::max(1u, B); // stream buffers per path slot: 128 B + one NEE record array per bounce (+ the per-chunk words)
			for (auto &P : h->pipe) have += P.slots * (144 + 17 * (size_t)std::max(1u, P.nee_bounces));
			const size_t pipes = (size_t)std::max(1, std::min(h->opt_overlap, (int)polaris_hip_tracer::kMaxPipes));
			const size_t budget = (free_b + have) / 10 * 9;
			// (batches in flight = min(overlap, #batches): it grows towards `overlap` as K shrinks, so it is recomputed per step)
			auto need = [&](uint32_t k) { return (size_t)k * Npad * slot_bytes * std::min<size_t>(pipes, (spp + k - 1) / k); };
			while (K > 1 && need(K) > budget) K = (K + 1) / 2;
		}
	}
	uint32_t n_batches = 0;
	int n_pipes = 1;
	for (;;) {
		n_batches = spp ? (spp + K - 1) / K : 0;
		n_pipes = exact ? 1 : (int)std::max<uint32_t>(1u, std::min<uint32_t>((uint32_t)h->opt_overlap, n_batches));
		int rc = POLARIS_OK;
		for (int p = 0; p < n_pipes && rc == POLARIS_OK; p++) rc = ensure_streams(h, p, (size_t)K * Npad, false, exact ? 1u : B);
		if (rc == POLARIS_OK) break;
		if (exact || h->opt_samples_per_batch > 0 || K == 1) return rc;
		(void)hipGetLastError(); // out of memory: release every pipeline's buffers, halve the batch, try again
		for (auto &P : h->pipe)
			if (P.q && P.slots) { (void)hipStreamSynchronize(P.q); free_pool(P.bufs); P.st = Streams{}; P.slots = 0; P.nee_bounces = 0; for (auto &x : P.nee) x = nullptr; for (auto &x : P.vis) x = nullptr; for (auto &x : P.cnt_occ_b) x = nullptr; }
		K = (K + 1) / 2;
	}
	if (need_seeds > h->seeds_cap) {
		if (h->d_seeds) (void)hipFree(h->d_seeds);
		h->d_seeds = nullptr;
		h->seeds_cap = 0;
		HIP_TRY(h, hipMalloc((void **)&h->d_seeds, need_seeds * sizeof(uint32_t)));
		h->seeds_cap = need_seeds;
	}
	h->cam.texel = make_float2(1.0f / (float)h->W, 1.0f / (float)h->H); // resources.go:130-133
	const size_t F = (size_t)h->W * h->H;
	hipStream_t q = h->stream;
	// The next slot of the ring: a peer may still be reading the previous frame's rows (polaris_hip_ipc_export).  Advanced only
	// HERE, behind the last step that can fail for want of memory (the batch buffers, the seed list): after a Trace that failed
	// before it queued anything, trace_slot() / merge_slot / export_block still name the slot of the last frame that was traced.
	if (h->ring_depth > 1) {
		h->ring_pos = (h->ring_pos + 1) % h->ring_depth;
		h->trace_acc = h->ring[h->ring_pos];
	}
	HIP_TRY(h, hipEventRecord(h->ev_start, q));
	if (r->accumulated_samples == 0) { // pipeline Reset stage (tracer.go:208-213): the frame accumulator lives on the merge stream
		{
			std::lock_guard<std::mutex> lk_merge(h->merge_mu);
			HIP_TRY(h, hipMemsetAsync(h->frame_acc, 0, F * sizeof(float4), h->merge_stream));
		}
		announce.now(); // merges queued from here on land on the cleared accumulator
	}
	// A merge that still READS this tracer's trace accumulator (MergeOutput(self) runs on the merge stream; Trace -> MergeOutput(self)
	// -> Trace without a SyncFramebuffer in between is the progressive loop) must be done before the rows are cleared and
	// rewritten: a device-side wait, the host does not block.  (A merge queued by ANOTHER handle onto its own merge stream
	// -- dst != src -- is ordered by the peer-read fence in polaris_hip_merge: src's `readers` event, below.)
	HIP_TRY(h, join_merges(h, q));
	HIP_TRY(h, wait_readers(h, q));
	HIP_TRY(h, hipMemsetAsync(h->trace_acc, 0, F * sizeof(float4), q)); // ClearTraceAccumulator (tracer.go:215)
	HIP_TRY(h, hipMemsetAsync(h->d_stats, 0, ST_COUNT * sizeof(unsigned long long), q));
	if (need_seeds) HIP_TRY(h, hipMemcpyAsync(h->d_seeds, seeds, need_seeds * sizeof(uint32_t), hipMemcpyHostToDevice, q));
	if (n_pipes > 1) { // fork: the other pipelines start after the clears and the seed upload
		HIP_TRY(h, hipEventRecord(h->ev_fork, q));
		for (int p = 1; p < n_pipes; p++) HIP_TRY(h, hipStreamWaitEvent(h->pipe[p].q, h->ev_fork, 0));
	}
	// From here on batches are in flight on several streams: a failure must not return before every one
	// of them has drained (resize / upload_scene / destroy free what the kernels still write to).
	struct DrainOnError {
		polaris_hip_tracer *h; int n; bool armed = true;
		~DrainOnError() { if (armed) for (int p = 0; p < n; p++) (void)hipStreamSynchronize(h->pipe[p].q); }
	} drain{h, n_pipes};
	uint32_t bi = 0;
	for (uint32_t s0 = 0; s0 < spp; s0 += K, bi++) {
		const int p = (int)(bi % (uint32_t)n_pipes), prev = (int)((bi + (uint32_t)n_pipes - 1) % (uint32_t)n_pipes);
		HIP_TRY(h, launch_batch(h, p, r, s0, std::min(K, spp - s0), N, Npad, exact, (n_pipes > 1 && bi > 0) ? h->pipe[prev].done : nullptr));
	}
	for (int p = 1; p < n_pipes; p++) HIP_TRY(h, hipStreamWaitEvent(q, h->pipe[p].done, 0)); // join
	HIP_TRY(h, hipGetLastError());
	unsigned long long hs[ST_COUNT];
	HIP_TRY(h, hipMemcpyAsync(hs, h->d_stats, sizeof hs, hipMemcpyDeviceToHost, q));
	if (h->ev_ipc_done[h->ring_pos]) HIP_TRY(h, hipEventRecord(h->ev_ipc_done[h->ring_pos], q)); // what a peer process's merge of THIS slot waits for (polaris_hip_merge_ipc)
	HIP_TRY(h, hipEventRecord(h->ev_stop, q));
	HIP_TRY(h, hipStreamSynchronize(q));
	drain.armed = false; // the join above made q wait for every pipeline
	collect_timers(h);
	for (uint32_t b = 0; b < POLARIS_MAX_BOUNCES; b++) {
		h->last_shade_counts[3 * b] = b < B ? hs[ST_HITS_BOUNCE + b] : 0;
		h->last_shade_counts[3 * b + 1] = b < B ? hs[ST_MISSES_BOUNCE + b] : 0;
		h->last_shade_counts[3 * b + 2] = b < B ? hs[ST_EMITTERS_BOUNCE + b] : 0;
	}
	if (stats) {
		memset(stats, 0, sizeof *stats);
		stats->primary_rays = (uint64_t)N * spp;
		for (uint32_t b = 0; b < B; b++) {
			stats->shaded_hits += hs[ST_HITS_BOUNCE + b];
			stats->shaded_misses += hs[ST_MISSES_BOUNCE + b];
			stats->emitter_hits += hs[ST_EMITTERS_BOUNCE + b];
		}
		stats->unoccluded = hs[ST_UNOCCLUDED];
		if (B > 0) stats->rays_per_bounce[0] = (uint64_t)N * spp;
		for (uint32_t b = 1; b < B; b++) {
			stats->rays_per_bounce[b] = hs[ST_RAYS_BOUNCE + b];
			stats->indirect_rays += hs[ST_RAYS_BOUNCE + b];
		}
		for (uint32_t b = 0; b < B; b++) {
			stats->occl_per_bounce[b] = hs[ST_OCCL_BOUNCE + b];
			stats->occlusion_rays += hs[ST_OCCL_BOUNCE + b];
		}
		float ms = 0.0f;
		(void)hipEventElapsedTime(&ms, h->ev_start, h->ev_stop);
		stats->device_ms = ms;
	}
	return POLARIS_OK;
}

// MergeOutput, shared by polaris_hip_merge / _merge_slot (a tracer of this process) and polaris_hip_merge_ipc (another
// process's ring, mapped here).  Everything on dst's merge stream, under dst->merge_mu only.
static int merge_rows(polaris_hip_tracer *dst, polaris_hip_tracer *src, polaris_hip_peer *peer, int slot, const PolarisBlockRequest *r) {
	// The source's trace accumulator is complete (its Trace is synchronous and has returned): snapshot what is needed of the
	// source under ITS lock, briefly.  The destination is touched under merge_mu only -- never under dst->mu, which the
	// destination's own Trace holds from start to end: a secondary's MergeOutput overlaps the primary's Trace
	// (renderer/default.go:188-191 calls it from the secondaries' goroutines; Exec1DNoWait, resources.go:119).
	int src_device = dst->device;
	uint32_t src_w = 0, src_h = 0;
	const float4 *src_acc = nullptr;
	bool bad_slot = false;
	if (src) {
		std::lock_guard<std::mutex> lk_src(src->mu);
		src_device = src->device; src_w = src->W; src_h = src->H;
		if (slot < 0) src_acc = src->trace_acc;
		else if ((uint32_t)slot < src->ring_depth) src_acc = src->ring[slot];
		else bad_slot = true;
	}
	std::lock_guard<std::mutex> lk(dst->merge_mu);
	auto fail_merge = [&](int code, const char *msg) { // (dst->error belongs to dst->mu, which a running Trace holds)
		dst->merge_error = msg;
		dst->merge_error_seq = ++g_error_seq;
		g_thread_error = msg;
		return code;
	};
	if (peer) {
		if (peer->owner != dst) return fail_merge(POLARIS_E_BAD_ARGUMENT, "merge_ipc: the peer was opened by another tracer");
		if (slot < 0 || (uint32_t)slot >= peer->depth) bad_slot = true;
		else src_acc = (const float4 *)peer->mem[slot];
		src_w = peer->W; src_h = peer->H;
	}
	if (bad_slot) return fail_merge(POLARIS_E_BAD_ARGUMENT, "merge: ring slot out of range");
	if (!r) return fail_merge(POLARIS_E_BAD_ARGUMENT, "block request is null");
	if (dst->W == 0 || dst->H == 0) return fail_merge(POLARIS_E_BAD_ARGUMENT, "frame dimensions not set (UpdateState FrameDimensions)");
	if (r->frame_w != dst->W || r->frame_h != dst->H) return fail_merge(POLARIS_E_BAD_ARGUMENT, "merge: request frame does not match the tracer's");
	if (r->block_h == 0 || (uint64_t)r->block_y + r->block_h > dst->H) return fail_merge(POLARIS_E_BAD_ARGUMENT, "merge: block rows outside the frame");
	if (r->block_x != 0 || (r->block_w != 0 && r->block_w != dst->W)) return fail_merge(POLARIS_E_BAD_ARGUMENT, "only full-width row blocks are supported (BlockW = FrameW, renderer/default.go:110)");
	if (src_w != dst->W || src_h != dst->H || !src_acc) return fail_merge(POLARIS_E_BAD_ARGUMENT, "merge: source tracer has different frame dimensions");
	if (hipSetDevice(dst->device) != hipSuccess) return fail_merge(POLARIS_E_DEVICE, "merge: hipSetDevice failed");
	const size_t off = (size_t)r->block_y * dst->W, n = (size_t)r->block_h * dst->W;
	const float4 *rows = src_acc + off;
	hipStream_t q = dst->merge_stream;
	if (peer && peer->ev[slot] && hipStreamWaitEvent(q, peer->ev[slot], 0) != hipSuccess) {
		// The device-side wait is belt and braces: the slot was announced by a host message AFTER the peer's synchronous Trace returned, so
		// its rows are complete.  A runtime that refuses to wait for another process's / another device's event must not cost the frame:
		// drop the peer's events (the wait is not retried) and go on with the host-side ordering alone.
		(void)hipGetLastError();
		for (auto &e : peer->ev)
			if (e) { (void)hipEventDestroy(e); e = nullptr; }
		(void)hipGetLastError();
		if (getenv("POLARIS_DEBUG")) fprintf(stderr, "[polaris] merge_ipc: hipStreamWaitEvent on the peer's inter-process event failed; continuing with the host-side ordering\n");
	}
	int branch = peer ? (peer->info.same_device == 1 ? POLARIS_MERGE_IPC_LOCAL : (peer->info.same_device == 0 ? POLARIS_MERGE_IPC_PEER : POLARIS_MERGE_IPC_UNKNOWN)) : POLARIS_MERGE_LOCAL;
	auto need_staging = [&]() -> bool { // the merge stream's staging strip, big enough for this block (merges are serialised on the stream)
		if (dst->staging_bytes >= n * sizeof(float4)) return true;
		(void)hipStreamSynchronize(q);
		if (dst->staging) (void)hipFree(dst->staging);
		dst->staging = nullptr;
		dst->staging_bytes = 0;
		if (hipMalloc(&dst->staging, n * sizeof(float4)) != hipSuccess) return false;
		dst->staging_bytes = n * sizeof(float4);
		return true;
	};
	if (peer && peer->info.staged) { // no peer access to the ring's GPU (or forced): the RUNTIME moves the rows, the kernel adds a local strip
		if (!need_staging()) return fail_merge(POLARIS_E_DEVICE, "merge: out of device memory for the staging strip");
		if (hipMemcpyAsync(dst->staging, rows, n * sizeof(float4), hipMemcpyDefault, q) != hipSuccess) return fail_merge(POLARIS_E_DEVICE, "merge_ipc: hipMemcpyAsync from the peer's ring failed");
		rows = (const float4 *)dst->staging;
		branch = POLARIS_MERGE_IPC_STAGED;
	}
	if (!peer && src_device != dst->device) {
		branch = POLARIS_MERGE_PEER_ACCESS;
		int can = 0;
		if (hipDeviceCanAccessPeer(&can, dst->device, src_device) != hipSuccess) return fail_merge(POLARIS_E_DEVICE, "merge: hipDeviceCanAccessPeer failed");
		bool direct = false;
		if (can) {
			hipError_t e = hipDeviceEnablePeerAccess(src_device, 0);
			if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) direct = true;
			(void)hipGetLastError();
		}
		if (!direct) { // staged copy over xGMI / PCIe, then add (the staging strip is the merge stream's: merges are serialised on it)
			branch = POLARIS_MERGE_STAGED;
			if (!need_staging()) return fail_merge(POLARIS_E_DEVICE, "merge: out of device memory for the staging strip");
			if (hipMemcpyPeerAsync(dst->staging, dst->device, rows, src_device, n * sizeof(float4), q) != hipSuccess) return fail_merge(POLARIS_E_DEVICE, "merge: hipMemcpyPeerAsync failed");
			rows = (const float4 *)dst->staging;
		}
	}
	{
		Timed t(dst, "aggregate", q, true);
		hipLaunchKernelGGL(k_aggregate, dim3(grid_for(n)), dim3(WG), 0, q, rows, dst->frame_acc + off, (uint32_t)n);
	}
	if (hipGetLastError() != hipSuccess) return fail_merge(POLARIS_E_DEVICE, "merge: kernel launch failed");
	dst->merge_counts[branch]++;
	// src's next Trace clears and rewrites the rows just queued for reading: it waits (on the device) for this event.  With
	// src == dst the merge stream itself is joined by Trace (join_merges).  Recorded for a merge from a named ring slot too
	// (merge_slot): with a ring of depth 1 that slot IS what the next Trace clears, and with a deeper ring the wait is for a
	// kernel of microseconds.
	if (src && src != dst) {
		hipEvent_t e = reader_event(src, dst->device);
		if (!e || hipEventRecord(e, q) != hipSuccess) { // cannot fence on the device: finish the read now
			(void)hipGetLastError();
			if (e) (void)hipEventDestroy(e);
			if (hipStreamSynchronize(q) != hipSuccess) return fail_merge(POLARIS_E_DEVICE, "merge: hipStreamSynchronize failed");
		} else {
			std::lock_guard<std::mutex> lk_r(src->readers_mu);
			src->readers.push_back({e, dst->device});
		}
	}
	return POLARIS_OK; // asynchronous like Exec1DNoWait (resources.go:119); completed by sync_framebuffer
}

int polaris_hip_merge(polaris_hip_tracer *dst, polaris_hip_tracer *src, const PolarisBlockRequest *r) {
	if (!dst || !src) return fail(dst, POLARIS_E_BAD_ARGUMENT, "merge: null tracer handle");
	return merge_rows(dst, src, nullptr, -1, r);
}

int polaris_hip_merge_slot(polaris_hip_tracer *dst, polaris_hip_tracer *src, uint32_t slot, const PolarisBlockRequest *r) {
	if (!dst || !src) return fail(dst, POLARIS_E_BAD_ARGUMENT, "merge_slot: null tracer handle");
	if (slot >= POLARIS_IPC_MAX_DEPTH) return fail(dst, POLARIS_E_BAD_ARGUMENT, "merge_slot: ring slot out of range");
	return merge_rows(dst, src, nullptr, (int)slot, r);
}

int polaris_hip_merge_ipc(polaris_hip_tracer *dst, polaris_hip_peer *peer, uint32_t slot, const PolarisBlockRequest *r) {
	if (!dst || !peer) return fail(dst, POLARIS_E_BAD_ARGUMENT, "merge_ipc: null handle");
	if (slot >= POLARIS_IPC_MAX_DEPTH) return fail(dst, POLARIS_E_BAD_ARGUMENT, "merge_ipc: ring slot out of range");
	return merge_rows(dst, nullptr, peer, (int)slot, r);
}

int polaris_hip_trace_slot(polaris_hip_tracer *h, uint32_t *slot) {
	if (!h || !slot) return fail(h, POLARIS_E_BAD_ARGUMENT, "trace_slot: null argument");
	std::lock_guard<std::mutex> lk(h->mu);
	*slot = h->ring_pos;
	return POLARIS_OK;
}

// The trace accumulator becomes a ring of `depth` IPC-exportable buffers; `out` = what a peer process needs to map them.
int polaris_hip_ipc_export(polaris_hip_tracer *h, uint32_t depth, PolarisIpcExport *out) {
	if (!h) return fail(nullptr, POLARIS_E_BAD_ARGUMENT, "handle is null");
	std::lock_guard<std::mutex> lk(h->mu);
	if (!out) return fail(h, POLARIS_E_BAD_ARGUMENT, "ipc_export: out is null");
	if (depth < 1 || depth > POLARIS_IPC_MAX_DEPTH) return fail(h, POLARIS_E_BAD_ARGUMENT, "ipc_export: depth must be 1..%d", POLARIS_IPC_MAX_DEPTH);
	if (h->W == 0 || h->H == 0 || !h->ring[0]) return fail(h, POLARIS_E_BAD_ARGUMENT, "frame dimensions not set (UpdateState FrameDimensions)");
	static_assert(sizeof(hipIpcMemHandle_t) == 64 && sizeof(hipIpcEventHandle_t) == 64, "PolarisIpcExport holds 64-byte handles");
	HIP_TRY(h, hipSetDevice(h->device));
	std::lock_guard<std::mutex> lk_merge(h->merge_mu);
	HIP_TRY(h, sync_all(h));
	const size_t F = (size_t)h->W * h->H;
	for (uint32_t i = 0; i < POLARIS_IPC_MAX_DEPTH; i++) { // grow or shrink the ring to `depth` slots (slot 0 always exists)
		if (i < depth && !h->ring[i]) {
			HIP_TRY(h, hipMalloc((void **)&h->ring[i], F * sizeof(float4)));
			HIP_TRY(h, hipMemsetAsync(h->ring[i], 0, F * sizeof(float4), h->stream));
		} else if (i >= depth && h->ring[i]) {
			(void)hipFree(h->ring[i]);
			h->ring[i] = nullptr;
		}
	}
	HIP_TRY(h, hipStreamSynchronize(h->stream));
	h->ring_depth = depth;
	h->ring_pos = h->ring_pos % depth;
	h->trace_acc = h->ring[h->ring_pos];
	memset(out, 0, sizeof *out);
	out->abi_version = POLARIS_HIP_ABI_VERSION; out->depth = depth; out->frame_w = h->W; out->frame_h = h->H;
	out->device = h->device; out->pid = (uint32_t)getpid();
	static_assert(sizeof out->pci_bus_id == sizeof h->pci_bus_id, "the export carries the tracer's bus id as it is");
	memcpy(out->pci_bus_id, h->pci_bus_id, sizeof out->pci_bus_id);
	for (uint32_t i = 0; i < depth; i++) {
		hipIpcMemHandle_t mh;
		HIP_TRY(h, hipIpcGetMemHandle(&mh, h->ring[i]));
		memcpy(out->mem[i], &mh, 64);
	}
	// the per-slot "Trace done" events: optional (a runtime that cannot export events still has the host-side ordering: Trace is
	// synchronous, and a slot is only announced after the Trace that wrote it has returned).  All slots or none.
	bool events = true;
	for (uint32_t i = 0; i < POLARIS_IPC_MAX_DEPTH; i++) {
		if (i >= depth) {
			if (h->ev_ipc_done[i]) { (void)hipEventDestroy(h->ev_ipc_done[i]); h->ev_ipc_done[i] = nullptr; }
			continue;
		}
		if (!h->ev_ipc_done[i]) {
			hipEvent_t e = nullptr;
			if (hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventInterprocess) == hipSuccess) h->ev_ipc_done[i] = e;
			else { (void)hipGetLastError(); events = false; break; }
		}
		hipIpcEventHandle_t eh;
		if (hipEventRecord(h->ev_ipc_done[i], h->stream) == hipSuccess && hipStreamSynchronize(h->stream) == hipSuccess &&
		    hipIpcGetEventHandle(&eh, h->ev_ipc_done[i]) == hipSuccess) memcpy(out->event[i], &eh, 64);
		else { (void)hipGetLastError(); events = false; break; }
	}
	if (!events) {
		for (auto &e : h->ev_ipc_done)
			if (e) { (void)hipEventDestroy(e); e = nullptr; }
		memset(out->event, 0, sizeof out->event);
	}
	out->has_event = events ? 1 : 0;
	return POLARIS_OK;
}

int polaris_hip_ipc_open(polaris_hip_tracer *dst, const PolarisIpcExport *x, polaris_hip_peer **out) {
	if (!dst) return fail(nullptr, POLARIS_E_BAD_ARGUMENT, "handle is null");
	std::lock_guard<std::mutex> lk(dst->mu);
	if (!x || !out) return fail(dst, POLARIS_E_BAD_ARGUMENT, "ipc_open: null argument");
	*out = nullptr;
	if (x->abi_version != POLARIS_HIP_ABI_VERSION) return fail(dst, POLARIS_E_BAD_ARGUMENT, "ipc_open: the export was made by ABI version %u, this library is %d", x->abi_version, POLARIS_HIP_ABI_VERSION);
	if (x->depth < 1 || x->depth > POLARIS_IPC_MAX_DEPTH) return fail(dst, POLARIS_E_BAD_ARGUMENT, "ipc_open: bad ring depth %u", x->depth);
	if (x->frame_w != dst->W || x->frame_h != dst->H || dst->W == 0) return fail(dst, POLARIS_E_BAD_ARGUMENT, "ipc_open: the peer's frame %ux%u does not match the tracer's %ux%u", x->frame_w, x->frame_h, dst->W, dst->H);
	if (x->pid == (uint32_t)getpid()) return fail(dst, POLARIS_E_UNSUPPORTED, "ipc_open: the export comes from this process (HIP cannot open its own IPC handle): use polaris_hip_merge / polaris_hip_merge_slot");
	HIP_TRY(dst, hipSetDevice(dst->device));
	polaris_hip_peer *p = new polaris_hip_peer();
	p->owner = dst; p->depth = x->depth; p->W = x->frame_w; p->H = x->frame_h;
	{ // what the mapping is: the exporter's GPU by bus id, as this process sees it
		PolarisPeerInfo &I = p->info;
		I.struct_size = sizeof I; I.pid = x->pid; I.exporter_device = x->device; I.local_device = -1; I.same_device = -1; I.can_access_peer = -1;
		I.depth = x->depth;
		memcpy(I.pci_bus_id, x->pci_bus_id, sizeof I.pci_bus_id);
		I.pci_bus_id[sizeof I.pci_bus_id - 1] = 0;
		if (I.pci_bus_id[0] && dst->pci_bus_id[0]) {
			I.same_device = strcmp(I.pci_bus_id, dst->pci_bus_id) == 0 ? 1 : 0;
			int local = -1;
			if (hipDeviceGetByPCIBusId(&local, I.pci_bus_id) == hipSuccess) I.local_device = local;
			else (void)hipGetLastError();
			if (I.same_device == 1) I.local_device = dst->device;
			int can = 0;
			if (I.local_device >= 0 && I.local_device != dst->device) {
				if (hipDeviceCanAccessPeer(&can, dst->device, I.local_device) == hipSuccess) I.can_access_peer = can;
				else (void)hipGetLastError();
			}
		}
		// A kernel must never be pointed at memory its device cannot reach (a fault there can take the whole node down):
		//   the ring is on another GPU this process cannot even SEE (a per-rank visibility mask): refuse, the caller has its strip fallback;
		//   the ring is on a visible GPU without peer access: merges go through the staging strip (a runtime copy, then the add).
		if (I.same_device == 0 && I.local_device < 0) {
			delete p;
			return fail(dst, POLARIS_E_UNSUPPORTED, "ipc_open: the peer's ring lives on GPU %s, which is not visible to this process (device visibility mask?): "
			            "a peer mapping cannot be verified; use a transport that does not need one", I.pci_bus_id);
		}
		I.staged = (dst->opt_ipc_staged || (I.same_device == 0 && I.can_access_peer == 0)) ? 1 : 0;
	}
	auto undo = [&]() {
		for (uint32_t i = 0; i < POLARIS_IPC_MAX_DEPTH; i++)
			if (p->mem[i]) (void)hipIpcCloseMemHandle(p->mem[i]);
		for (auto &e : p->ev)
			if (e) (void)hipEventDestroy(e);
		delete p;
		(void)hipGetLastError();
	};
	for (uint32_t i = 0; i < x->depth; i++) {
		hipIpcMemHandle_t mh;
		memcpy(&mh, x->mem[i], 64);
		const hipError_t e = hipIpcOpenMemHandle(&p->mem[i], mh, hipIpcMemLazyEnablePeerAccess);
		if (e != hipSuccess) {
			p->mem[i] = nullptr;
			undo();
			return fail(dst, POLARIS_E_UNSUPPORTED, "ipc_open: hipIpcOpenMemHandle (slot %u, peer pid %u device %d): %s", i, x->pid, x->device, hipGetErrorString(e));
		}
	}
	if (x->has_event) { // optional: without them the host message that follows the peer's synchronous Trace is the only ordering
		for (uint32_t i = 0; i < x->depth; i++) {
			hipIpcEventHandle_t eh;
			memcpy(&eh, x->event[i], 64);
			if (hipIpcOpenEventHandle(&p->ev[i], eh) != hipSuccess) { // all or none
				(void)hipGetLastError();
				p->ev[i] = nullptr;
				for (uint32_t j = 0; j < i; j++) { (void)hipEventDestroy(p->ev[j]); p->ev[j] = nullptr; }
				break;
			}
		}
	}
	// Every slot's event was recorded (and waited for) once by the export, so each must read COMPLETE here.  One that does not -- its state
	// is not visible in this process (another device, another container) -- would make a merge's device-side wait hang instead of fail:
	// drop them all, the host message that follows the peer's synchronous Trace is ordering enough.
	if (p->ev[0]) {
		bool visible = true;
		for (uint32_t i = 0; i < x->depth && visible; i++) visible = hipEventQuery(p->ev[i]) == hipSuccess;
		if (!visible) {
			(void)hipGetLastError();
			for (auto &e : p->ev)
				if (e) { (void)hipEventDestroy(e); e = nullptr; }
			(void)hipGetLastError();
			if (getenv("POLARIS_DEBUG")) fprintf(stderr, "[polaris] ipc_open: the peer's inter-process events do not read complete here; merging on the host-side ordering alone\n");
		}
	}
	p->info.has_events = p->ev[0] ? 1u : 0u;
	*out = p;
	return POLARIS_OK;
}

int polaris_hip_peer_info(polaris_hip_peer *p, PolarisPeerInfo *out) {
	if (!p || !out) return fail(nullptr, POLARIS_E_BAD_ARGUMENT, "peer_info: null argument");
	if (out->struct_size != sizeof(PolarisPeerInfo)) return fail(nullptr, POLARIS_E_BAD_ARGUMENT, "peer_info: struct_size %u, this library's PolarisPeerInfo has %zu bytes", out->struct_size, sizeof(PolarisPeerInfo));
	std::lock_guard<std::mutex> lk(p->owner->merge_mu); // (a merge that finds the peer's events unusable drops them: has_events follows)
	*out = p->info;
	out->has_events = p->ev[0] ? 1u : 0u;
	return POLARIS_OK;
}

int polaris_hip_merge_counts(polaris_hip_tracer *dst, uint64_t counts[POLARIS_MERGE_BRANCHES]) {
	if (!dst || !counts) return fail(dst, POLARIS_E_BAD_ARGUMENT, "merge_counts: null argument");
	std::lock_guard<std::mutex> lk(dst->merge_mu);
	for (int i = 0; i < POLARIS_MERGE_BRANCHES; i++) counts[i] = dst->merge_counts[i];
	return POLARIS_OK;
}

int polaris_hip_ipc_close(polaris_hip_tracer *dst, polaris_hip_peer *p) {
	if (!dst || !p) return fail(dst, POLARIS_E_BAD_ARGUMENT, "ipc_close: null argument");
	if (p->owner != dst) return fail(dst, POLARIS_E_BAD_ARGUMENT, "ipc_close: the peer was opened by another tracer");
	std::lock_guard<std::mutex> lk(dst->merge_mu);
	(void)hipSetDevice(dst->device);
	(void)hipStreamSynchronize(dst->merge_stream); // a merge may still be reading the mapping
	for (uint32_t i = 0; i < POLARIS_IPC_MAX_DEPTH; i++)
		if (p->mem[i]) (void)hipIpcCloseMemHandle(p->mem[i]);
	for (auto &e : p->ev)
		if (e) (void)hipEventDestroy(e);
	(void)hipGetLastError();
	delete p;
	return POLARIS_OK;
}

int polaris_hip_export_block(polaris_hip_tracer *h, const PolarisBlockRequest *r, void *device_dst) {
	if (!h) return fail(nullptr, POLARIS_E_BAD_ARGUMENT, "handle is null");
	std::lock_guard<std::mutex> lk(h->mu);
	if (int rc = check_request(h, r)) return rc;
	if (!device_dst) return fail(h, POLARIS_E_BAD_ARGUMENT, "export: destination is null");
	HIP_TRY(h, hipSetDevice(h->device));
	const size_t off = (size_t)r->block_y * h->W, n = (size_t)r->block_h * h->W;
	HIP_TRY(h, hipMemcpyAsync(device_dst, h->trace_acc + off, n * sizeof(float4), hipMemcpyDeviceToDevice, h->stream));
	HIP_TRY(h, hipStreamSynchronize(h->stream));
	return POLARIS_OK;
}

int polaris_hip_merge_device(polaris_hip_tracer *dst, const void *device_rows, const PolarisBlockRequest *r) {
	if (!dst) return fail(nullptr, POLARIS_E_BAD_ARGUMENT, "handle is null");
	std::lock_guard<std::mutex> lk(dst->mu);
	if (int rc = check_request(dst, r)) return rc;
	if (!device_rows) return fail(dst, POLARIS_E_BAD_ARGUMENT, "merge_device: source is null");
	HIP_TRY(dst, hipSetDevice(dst->device));
	const size_t off = (size_t)r->block_y * dst->W, n = (size_t)r->block_h * dst->W;
	std::lock_guard<std::mutex> lk_merge(dst->merge_mu); // the frame accumulator is the merge stream's
	hipLaunchKernelGGL(k_aggregate, dim3(grid_for(n)), dim3(WG), 0, dst->merge_stream, (const float4 *)device_rows, dst->frame_acc + off,
	                   (uint32_t)n);
	HIP_TRY(dst, hipGetLastError());
	dst->merge_counts[POLARIS_MERGE_DEVICE_STRIP]++;
	HIP_TRY(dst, hipStreamSynchronize(dst->merge_stream)); // the caller owns device_rows: do not outlive it
	return POLARIS_OK;
}

int polaris_hip_reset_frame(polaris_hip_tracer *h) {
	if (!h) return fail(nullptr, POLARIS_E_BAD_ARGUMENT, "handle is null");
	std::lock_guard<std::mutex> lk(h->mu);
	if (h->W == 0 || h->H == 0) return fail(h, POLARIS_E_BAD_ARGUMENT, "frame dimensions not set (UpdateState FrameDimensions)");
	HIP_TRY(h, hipSetDevice(h->device));
	{
		std::lock_guard<std::mutex> lk_merge(h->merge_mu);
		HIP_TRY(h, hipMemsetAsync(h->frame_acc, 0, (size_t)h->W * h->H * sizeof(float4), h->merge_stream));
		h->reset_epoch++;
	}
	h->reset_cv.notify_all();
	return POLARIS_OK;
}

int polaris_hip_reset_epoch(polaris_hip_tracer *h, uint64_t *epoch) {
	if (!h || !epoch) return fail(nullptr, POLARIS_E_BAD_ARGUMENT, "reset_epoch: null argument");
	std::lock_guard<std::mutex> lk(h->merge_mu);
	*epoch = h->reset_epoch;
	return POLARIS_OK;
}

int polaris_hip_wait_reset(polaris_hip_tracer *h, uint64_t epoch) {
	if (!h) return fail(nullptr, POLARIS_E_BAD_ARGUMENT, "handle is null");
	std::unique_lock<std::mutex> lk(h->merge_mu);
	// (bounded: a caller that waits for a Trace nobody will issue gets an error instead of a hung thread)
	if (!h->reset_cv.wait_for(lk, std::chrono::seconds(120), [&] { return h->reset_epoch > epoch; })) {
		h->merge_error = "wait_reset: no Trace with accumulated_samples == 0 (or reset_frame) arrived within 120 s";
		h->merge_error_seq = ++g_error_seq;
		return POLARIS_E_TIMEOUT;
	}
	return POLARIS_OK;
}

int polaris_hip_sync_framebuffer(polaris_hip_tracer *h, const PolarisBlockRequest *r) {
	if (!h) return fail(nullptr, POLARIS_E_BAD_ARGUMENT, "handle is null");
	std::lock_guard<std::mutex> lk(h->mu);
	if (!h->have_scene) return fail(h, POLARIS_E_NO_SCENE_DATA, "no scene data uploaded"); // tracer.go:254-256
	if (int rc = check_request(h, r)) return rc;
	HIP_TRY(h, hipSetDevice(h->device));
	const size_t off = (size_t)r->block_y * h->W, n = (size_t)r->block_h * h->W;
	const float weight = (float)(1.0 / (float)(r->accumulated_samples + r->samples_per_pixel)); // resources.go:347
	HIP_TRY(h, join_merges(h, h->stream)); // "wait for pending merges" (tracer.go:258-262): everything queued on the merge stream so far
	{
		Timed t(h, "tonemap");
		hipLaunchKernelGGL(k_tonemap, dim3(grid_for(n)), dim3(WG), 0, h->stream, h->frame_acc + off, h->framebuffer + off, (uint32_t)n,
		                   weight, r->exposure);
	}
	HIP_TRY(h, hipGetLastError());
	HIP_TRY(h, hipStreamSynchronize(h->stream));
	collect_timers(h);
	return POLARIS_OK;
}

int polaris_hip_read_framebuffer(polaris_hip_tracer *h, uint8_t *rgba, size_t n_bytes) {
	if (!h) return fail(nullptr, POLARIS_E_BAD_ARGUMENT, "handle is null");
	std::lock_guard<std::mutex> lk(h->mu);
	const size_t need = (size_t)h->W * h->H * 4;
	if (!rgba || n_bytes < need || need == 0) return fail(h, POLARIS_E_BAD_ARGUMENT, "read_framebuffer: need %zu bytes", need);
	HIP_TRY(h, hipSetDevice(h->device));
	HIP_TRY(h, hipMemcpyAsync(rgba, h->framebuffer, need, hipMemcpyDeviceToHost, h->stream));
	HIP_TRY(h, hipStreamSynchronize(h->stream));
	return POLARIS_OK;
}

int polaris_hip_read_accumulator(polaris_hip_tracer *h, int which, float *out, size_t n_floats) {
	if (!h) return fail(nullptr, POLARIS_E_BAD_ARGUMENT, "handle is null");
	std::lock_guard<std::mutex> lk(h->mu);
	const size_t need = (size_t)h->W * h->H * 4;
	if (!out || n_floats < need || need == 0 || which < 0 || which > 1)
		return fail(h, POLARIS_E_BAD_ARGUMENT, "read_accumulator: need %zu floats, which in {0,1}", need);
	HIP_TRY(h, hipSetDevice(h->device));
	if (which == 1) HIP_TRY(h, join_merges(h, h->stream));
	HIP_TRY(h, hipMemcpyAsync(out, which == 0 ? h->trace_acc : h->frame_acc, need * sizeof(float), hipMemcpyDeviceToHost, h->stream));
	HIP_TRY(h, hipStreamSynchronize(h->stream));
	return POLARIS_OK;
}

int polaris_hip_tap_primary(polaris_hip_tracer *h, const PolarisBlockRequest *r, uint32_t seed, float *rays, int32_t *hit,
                            float *wuvt, int32_t *tri) {
	if (!h) return fail(nullptr, POLARIS_E_BAD_ARGUMENT, "handle is null");
	std::lock_guard<std::mutex> lk(h->mu);
	if (!h->have_scene) return fail(h, POLARIS_E_NO_SCENE_DATA, "no scene data uploaded");
	if (int rc = check_request(h, r)) return rc;
	if (!h->have_camera) return fail(h, POLARIS_E_BAD_ARGUMENT, "camera not set");
	const uint32_t N = h->W * r->block_h, Npad = (N + WG - 1) / WG * WG;
	HIP_TRY(h, hipSetDevice(h->device));
	if (int rc = ensure_streams(h, 0, std::max<size_t>(h->pipe[0].slots, Npad), true)) return rc;
	Streams &st0 = h->pipe[0].st;
	st0.hit12 = 0; // (the tap returns the hit distance: 16-byte records)
	st0.o12 = 0;
	if (h->seeds_cap < 1) {
		HIP_TRY(h, hipMalloc((void **)&h->d_seeds, 64 * sizeof(uint32_t)));
		h->seeds_cap = 64;
	}
	h->cam.texel = make_float2(1.0f / (float)h->W, 1.0f / (float)h->H);
	hipStream_t q = h->stream;
	HIP_TRY(h, hipMemcpyAsync(h->d_seeds, &seed, sizeof seed, hipMemcpyHostToDevice, q));
	hipLaunchKernelGGL(k_generate, dim3(Npad / WG), dim3(WG), 0, q, st0, h->cam, h->d_seeds, 1u, 0u, N, Npad, h->W, r->block_y, 1, 1);
	if (h->packet_primary) hipLaunchKernelGGL(k_trace_packet<false>, dim3(Npad / WG), dim3(WG), 0, q, st0, h->bvh, (float4 *)nullptr, h->d_stats, h->cam.eye);
	else hipLaunchKernelGGL(k_intersect, dim3(Npad / WG), dim3(WG), 0, q, st0, h->bvh);
	HIP_TRY(h, hipGetLastError());
	std::vector<float4> ro(N), rd(N), ht(N);
	std::vector<int> inst(N);
	HIP_TRY(h, hipMemcpyAsync(ro.data(), st0.ray_o, N * sizeof(float4), hipMemcpyDeviceToHost, q));
	HIP_TRY(h, hipMemcpyAsync(rd.data(), st0.ray_d, N * sizeof(float4), hipMemcpyDeviceToHost, q));
	HIP_TRY(h, hipMemcpyAsync(ht.data(), st0.hit, N * sizeof(float4), hipMemcpyDeviceToHost, q));
	HIP_TRY(h, hipMemcpyAsync(inst.data(), st0.hit_inst, N * sizeof(int), hipMemcpyDeviceToHost, q));
	HIP_TRY(h, hipStreamSynchronize(q));
	for (uint32_t i = 0; i < N; i++) {
		int t;
		memcpy(&t, &ht[i].w, 4);
		if (rays) {
			int pw;
			memcpy(&pw, &rd[i].w, 4);
			float *o = rays + 8 * (size_t)i;
			o[0] = ro[i].x; o[1] = ro[i].y; o[2] = ro[i].z; o[3] = ro[i].w;
			o[4] = rd[i].x; o[5] = rd[i].y; o[6] = rd[i].z; o[7] = (float)(pw & 0xFFFFFF); // util/ray.cl:11
		}
		if (hit) hit[i] = t >= 0 ? 1 : 0;
		if (t >= 0) t &= (int)((1u << h->scene.tri_bits) - 1u); // (the shading class rides above the triangle index)
		if (t >= 0) {
			if (wuvt) {
				float *o = wuvt + 4 * (size_t)i;
				o[0] = 1.0f - (ht[i].x + ht[i].y); o[1] = ht[i].x; o[2] = ht[i].y; o[3] = ht[i].z; // intersect.cl:283-288
			}
			if (tri) { tri[2 * (size_t)i] = inst[i]; tri[2 * (size_t)i + 1] = t; }
		}
	}
	return POLARIS_OK;
}

int polaris_hip_probe(polaris_hip_tracer *h, int kind, uint32_t index, uint32_t n, const float *in, float *out) {
	if (!h) return fail(nullptr, POLARIS_E_BAD_ARGUMENT, "handle is null");
	std::lock_guard<std::mutex> lk(h->mu);
	if (!h->have_scene) return fail(h, POLARIS_E_NO_SCENE_DATA, "no scene data uploaded");
	if (kind < 0 || kind > 3 || (n && (!in || !out))) return fail(h, POLARIS_E_BAD_ARGUMENT, "probe: bad kind or null buffers");
	const uint32_t limit = (kind == kProbeBxdf || kind == kProbeMaterial) ? h->scene.num_nodes : (kind == kProbeTexture ? h->scene.num_textures : h->scene.num_emissives);
	if (index >= limit) return fail(h, POLARIS_E_BAD_ARGUMENT, "probe: index %u out of range (%u)", index, limit);
	if (n == 0) return POLARIS_OK;
	HIP_TRY(h, hipSetDevice(h->device));
	const size_t nin = (size_t)n * kProbeIn[kind], nout = (size_t)n * kProbeOut[kind];
	float *d_in = nullptr, *d_out = nullptr;
	HIP_TRY(h, hipMalloc((void **)&d_in, nin * sizeof(float)));
	if (hipMalloc((void **)&d_out, nout * sizeof(float)) != hipSuccess) { (void)hipFree(d_in); return fail(h, POLARIS_E_DEVICE, "probe: out of device memory"); }
	hipStream_t q = h->stream;
	hipError_t e = hipMemcpyAsync(d_in, in, nin * sizeof(float), hipMemcpyHostToDevice, q);
	const bool staged = h->opt_stage_lds && h->scene.num_nodes <= kLdsMatNodes && h->scene.num_emissives <= kLdsLights &&
	                    h->scene.num_textures <= kLdsTextures;
	if (e == hipSuccess) {
		if (staged) hipLaunchKernelGGL(k_probe<true>, dim3(grid_for(n)), dim3(WG), 0, q, h->scene, kind, index, n, d_in, d_out);
		else hipLaunchKernelGGL(k_probe<false>, dim3(grid_for(n)), dim3(WG), 0, q, h->scene, kind, index, n, d_in, d_out);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, nout * sizeof(float), hipMemcpyDeviceToHost, q);
	if (e == hipSuccess) e = hipStreamSynchronize(q);
	else (void)hipStreamSynchronize(q);
	(void)hipFree(d_in);
	(void)hipFree(d_out);
	if (e != hipSuccess) return fail(h, POLARIS_E_DEVICE, "probe: %s", hipGetErrorString(e));
	return POLARIS_OK;
}

int polaris_hip_probe_intersect(polaris_hip_tracer *h, const float *rays, uint32_t n, int any_hit, int32_t *hit, float *wuvt, int32_t *tri) {
	if (!h) return fail(nullptr, POLARIS_E_BAD_ARGUMENT, "handle is null");
	std::lock_guard<std::mutex> lk(h->mu);
	if (!h->have_scene) return fail(h, POLARIS_E_NO_SCENE_DATA, "no scene data uploaded");
	if (n && (!rays || !hit)) return fail(h, POLARIS_E_BAD_ARGUMENT, "probe_intersect: null buffers");
	if (n > (1u << 24)) return fail(h, POLARIS_E_BAD_ARGUMENT, "probe_intersect: at most 2^24 rays per call");
	if (n == 0) return POLARIS_OK;
	for (uint32_t i = 0; i < n; i++) // (what the tracer's own rays satisfy by construction; the triangle tests rely on it: kernels.h, rcp_det)
		for (int k = 0; k < 3; k++)
			if (!(std::fabs(rays[8 * (size_t)i + k]) <= kMaxCoordinate) || !(std::fabs(rays[8 * (size_t)i + 4 + k]) <= 1024.0f))
				return fail(h, POLARIS_E_BAD_ARGUMENT, "probe_intersect: ray %u: origin beyond 2^40 or direction component beyond 2^10 (or not finite)", i);
	HIP_TRY(h, hipSetDevice(h->device));
	const uint32_t npad = (n + WG - 1) / WG * WG, wgs = npad / WG;
	if (int rc = ensure_streams(h, 0, std::max<size_t>(h->pipe[0].slots, npad), false)) return rc;
	polaris_hip_tracer::Pipe &P = h->pipe[0];
	P.st.hit12 = 0; // (the probe returns the hit distance: 16-byte records)
	P.st.o12 = 0;   // (... and takes arbitrary max distances)
	hipStream_t q = P.q;
	float *d_rays = nullptr;
	HIP_TRY(h, hipMalloc((void **)&d_rays, (size_t)n * 8 * sizeof(float)));
	hipError_t e = hipMemcpyAsync(d_rays, rays, (size_t)n * 8 * sizeof(float), hipMemcpyHostToDevice, q);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_probe_rays, dim3(wgs), dim3(WG), 0, q, P.st, d_rays, n, any_hit ? 1 : 0);
		// the traversal kernel the options select for bounce rays (any_hit: shadow rays); packet_primary=1 sends closest-hit
		// probes through the wave-packet kernel instead
		if (any_hit) {
			if (h->opt_packet_shadow > 0) hipLaunchKernelGGL(k_trace_packet<true>, dim3(wgs), dim3(WG), 0, q, P.st, h->bvh, P.st.lsum, h->d_stats, h->cam.eye);
			else if (h->opt_traversal) (void)launch_trace<true>(h, P, P.st, std::min<uint32_t>(wgs, (uint32_t)h->num_cus * (uint32_t)std::max(1, h->occl_resident_per_cu)), wgs, P.st.lsum);
			else hipLaunchKernelGGL(k_occlusion, dim3(wgs), dim3(WG), 0, q, P.st, h->bvh, P.st.lsum, h->d_stats);
		} else {
			if (h->opt_packet_primary == 1) hipLaunchKernelGGL(k_trace_packet<false>, dim3(wgs), dim3(WG), 0, q, P.st, h->bvh, (float4 *)nullptr, h->d_stats, h->cam.eye);
			else if (h->opt_traversal) (void)launch_trace<false>(h, P, P.st, std::min<uint32_t>(wgs, (uint32_t)h->num_cus * (uint32_t)std::max(1, h->trace_resident_per_cu)), wgs, nullptr);
			else hipLaunchKernelGGL(k_intersect, dim3(wgs), dim3(WG), 0, q, P.st, h->bvh);
		}
		e = hipGetLastError();
	}
	std::vector<float4> res(n);
	if (e == hipSuccess) e = hipMemcpyAsync(res.data(), any_hit ? P.st.lsum : P.st.hit, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost, q);
	const hipError_t e2 = hipStreamSynchronize(q);
	(void)hipFree(d_rays);
	if (e == hipSuccess) e = e2;
	if (e != hipSuccess) return fail(h, POLARIS_E_DEVICE, "probe_intersect: %s", hipGetErrorString(e));
	for (uint32_t i = 0; i < n; i++) {
		if (any_hit) { hit[i] = res[i].x == 0.0f ? 1 : 0; continue; } // occluded rays leave their cell untouched
		int t;
		memcpy(&t, &res[i].w, 4);
		hit[i] = t >= 0 ? 1 : 0;
		if (t >= 0) t &= (int)((1u << h->scene.tri_bits) - 1u);
		if (t >= 0 && wuvt) { float *o = wuvt + 4 * (size_t)i; o[0] = 1.0f - (res[i].x + res[i].y); o[1] = res[i].x; o[2] = res[i].y; o[3] = res[i].z; } // intersect.cl:283-288
		if (tri) tri[i] = t;
	}
	return POLARIS_OK;
}

int polaris_hip_selftest_rcp(polaris_hip_tracer *h, float lo, float hi, uint64_t *mismatches_inside, uint64_t *mismatches_outside, uint32_t *sample) {
	if (!h) return fail(nullptr, POLARIS_E_BAD_ARGUMENT, "handle is null");
	std::lock_guard<std::mutex> lk(h->mu);
	if (!mismatches_inside || !mismatches_outside) return fail(h, POLARIS_E_BAD_ARGUMENT, "selftest_rcp: null outputs");
	HIP_TRY(h, hipSetDevice(h->device));
	unsigned long long *d = nullptr, res[3] = {0, 0, 0};
	HIP_TRY(h, hipMalloc((void **)&d, sizeof res));
	hipError_t e = hipMemcpyAsync(d, res, sizeof res, hipMemcpyHostToDevice, h->stream);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_rcp_sweep, dim3((uint32_t)h->num_cus * 32u), dim3(WG), 0, h->stream, lo, hi, d);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipMemcpyAsync(res, d, sizeof res, hipMemcpyDeviceToHost, h->stream);
	const hipError_t e2 = hipStreamSynchronize(h->stream);
	(void)hipFree(d);
	if (e == hipSuccess) e = e2;
	if (e != hipSuccess) return fail(h, POLARIS_E_DEVICE, "selftest_rcp: %s", hipGetErrorString(e));
	*mismatches_inside = res[0];
	*mismatches_outside = res[1];
	if (sample) *sample = (uint32_t)res[2];
	return POLARIS_OK;
}

int polaris_hip_kernel_symbol(polaris_hip_tracer *h, const char *kernel, char symbol[128]) {
	if (!h) return fail(nullptr, POLARIS_E_BAD_ARGUMENT, "handle is null");
	std::lock_guard<std::mutex> lk(h->mu);
	if (!kernel || !symbol) return fail(h, POLARIS_E_BAD_ARGUMENT, "kernel_symbol: null argument");
	auto it = h->timer_symbol.find(kernel);
	snprintf(symbol, 128, "%s", it == h->timer_symbol.end() ? "" : it->second.c_str());
	return POLARIS_OK;
}

int polaris_hip_shade_counts(polaris_hip_tracer *h, uint64_t *counts, size_t n_counts) {
	if (!h) return fail(nullptr, POLARIS_E_BAD_ARGUMENT, "handle is null");
	std::lock_guard<std::mutex> lk(h->mu);
	if (!counts || n_counts < 4 * (size_t)POLARIS_MAX_BOUNCES) return fail(h, POLARIS_E_BAD_ARGUMENT, "shade_counts: need 4 * POLARIS_MAX_BOUNCES entries");
	for (int b = 0; b < POLARIS_MAX_BOUNCES; b++) {
		counts[4 * b] = h->last_shade_counts[3 * b];
		counts[4 * b + 1] = h->last_shade_counts[3 * b + 1];
		counts[4 * b + 2] = h->last_shade_counts[3 * b + 2];
		counts[4 * b + 3] = (uint64_t)h->last_shade_timer[b];
	}
	return POLARIS_OK;
}

int polaris_hip_kernel_ms(polaris_hip_tracer *h, const char *kernel, double *ms, uint64_t *launches) {
	if (!h) return fail(nullptr, POLARIS_E_BAD_ARGUMENT, "handle is null");
	std::lock_guard<std::mutex> lk(h->mu);
	if (!kernel) return fail(h, POLARIS_E_BAD_ARGUMENT, "kernel name is null");
	auto it = h->timers.find(kernel);
	KernelTimer t = it == h->timers.end() ? KernelTimer{} : it->second;
	if (it != h->timers.end()) h->timers.erase(it);
	if (ms) *ms = t.ms;
	if (launches) *launches = t.launches;
	return POLARIS_OK;
}

} // extern "C"
