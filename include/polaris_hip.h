/*
 * polaris_hip.h -- C ABI of the MI355X (gfx950) tracer backend for polaris.
 *
 * This is the drop-in boundary: exactly what a Go package `tracer/hip` implementing
 * tracer.Tracer (reference tracer/tracer.go:80-111) binds through cgo, replacing
 * tracer/opencl (+ tracer/opencl/device).  Plain pointers and sizes only; no C++/torch
 * types; every function returns 0 on success or a POLARIS_E_* code, never throws/aborts.
 * INTEGRATION.md shows the Go side.
 *
 * Method-by-method correspondence (reference file:line -> entry point):
 *   device.GetPlatformInfo / SelectDevices, Speed estimate
 *       tracer/opencl/device/platform.go, device.go:209-222  -> polaris_hip_device_count/_info
 *   opencl.NewTracer + Tracer.Init   tracer/opencl/tracer.go:58-117      -> polaris_hip_create
 *   Tracer.Close                     tracer/opencl/tracer.go:120-145     -> polaris_hip_destroy
 *   Tracer.UpdateState(FrameDimensions)  tracer.go:169-171 -> buffers.go:127-174 Resize
 *                                                                        -> polaris_hip_resize
 *   Tracer.UpdateState(SceneData)    tracer.go:172-174 -> buffers.go:180-201 UploadSceneData
 *                                                                        -> polaris_hip_upload_scene
 *   Tracer.UpdateState(CameraData)   tracer.go:175-179                   -> polaris_hip_set_camera
 *   Tracer.Trace                     tracer.go:194-247 + pipeline.go:94-213 -> polaris_hip_trace
 *   Tracer.MergeOutput               tracer.go:279-286, resources.go:108-124 -> polaris_hip_merge
 *   Tracer.SyncFramebuffer           tracer.go:250-276, resources.go:344-360 -> polaris_hip_sync_framebuffer
 *   SaveFrameBuffer's ReadData       pipeline.go:226-232                 -> polaris_hip_read_framebuffer
 *   errors                           tracer/opencl/errors.go:5-22        -> POLARIS_E_* + polaris_hip_last_error
 *
 * Threading: any entry point may be called from any OS thread (goroutines migrate); each
 * call binds the handle's device.  Calls on ONE handle are serialised by a per-handle mutex;
 * polaris_hip_merge(dst, src, ..) may be called concurrently from several threads onto one
 * dst (renderer/default.go:188-191 does exactly that) and with src == dst.
 * Ownership: the library copies everything it needs before returning; no caller pointer is
 * retained (cgo rule).
 */
#ifndef POLARIS_HIP_H
#define POLARIS_HIP_H

#include "polaris_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ABI history.  2: + reset_epoch / wait_reset, kernel_symbol, shade_counts.  3: + ipc_export / ipc_open / ipc_close / merge_ipc / merge_slot /
 * trace_slot (additions only: older callers keep working).  4: PolarisIpcExport carries one event handle PER RING SLOT (the blob grows from 352
 * to 544 bytes; ipc_open refuses a blob of another version); PolarisBvhBuildInput grew by `algorithm` (64 -> 72 bytes) and polaris_hip_build_bvh's
 * default for a zero-initialised caller changed from the linear BVH to the binned-SAH builder.  5: + device_identity / can_access_peer /
 * peer_info / merge_counts (which GPU a tracer really runs on, and which branch every merge took: what a multi-GPU bench line needs to prove
 * itself); PolarisIpcExport carries the exporting GPU's PCI bus id (544 -> 576 bytes); PolarisBvhBuildInput LEADS with `struct_size`
 * (72 -> 80 bytes): a caller built against another layout is refused instead of being read past its end. */
#define POLARIS_HIP_ABI_VERSION 5

/* status codes (0 = ok).  The first three mirror tracer/opencl/errors.go sentinels. */
#define POLARIS_OK                0
#define POLARIS_E_NO_SCENE_DATA   1  /* ErrNoSceneData: Trace/Sync before a scene upload (tracer.go:203-205,254-256) */
#define POLARIS_E_BAD_ARGUMENT    2  /* null pointer, block outside frame, too many bounces, short seed list */
#define POLARIS_E_NO_DEVICE       3  /* device index out of range / no HIP device */
#define POLARIS_E_DEVICE          4  /* a HIP runtime call failed; text in last_error */
#define POLARIS_E_BAD_SCENE       5  /* scene arrays inconsistent (index out of range, BVH deeper than the traversal stack) */
#define POLARIS_E_UNSUPPORTED     6  /* e.g. merge between handles that cannot reach each other */
#define POLARIS_E_TIMEOUT         7  /* polaris_hip_wait_reset: the awaited Reset stage did not arrive within 120 s */

typedef struct polaris_hip_tracer polaris_hip_tracer; /* opaque */

int polaris_hip_abi_version(void);

/* Device enumeration.  Speed estimate of the reference = compute_units * clock_mhz / 1000
 * (tracer/opencl/device/device.go:219); the Go side computes it from these two numbers. */
int polaris_hip_device_count(void);
int polaris_hip_device_info(int index, char name[256], uint32_t *compute_units, uint32_t *clock_mhz,
                            uint64_t *global_mem_bytes);

/* Which physical GPU a HIP device index of THIS process is.  Device indices are per process: a launcher that masks every rank to one
 * visible device (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES per rank) makes index 0 a different GPU in every rank, and a
 * misconfigured one makes it the SAME GPU in all of them -- the PCI bus id and the UUID tell which.  The reference names its devices by
 * the OpenCL device name and has ONE process see them all (tracer/opencl/device/platform.go, renderer/default.go:204-256: one tracer
 * per enumerated device); with one process per GPU the identity has to travel with the numbers (bench.py: config.devices,
 * config.distinct_gpus).  The caller sets struct_size = sizeof(PolarisDeviceIdentity). */
typedef struct PolarisDeviceIdentity {
	uint32_t struct_size;
	int32_t hip_index;
	char pci_bus_id[32];   /* hipDeviceGetPCIBusId, e.g. "0000:05:00.0" */
	uint8_t uuid[16];      /* hipDeviceGetUuid; all zero if the runtime reports none */
	uint32_t compute_units, clock_mhz;
	uint64_t global_mem_bytes;
	char name[64];         /* hipDeviceProp_t.name, truncated */
	char gcn_arch[32];     /* hipDeviceProp_t.gcnArchName, e.g. "gfx950:sramecc+:xnack-" */
} PolarisDeviceIdentity;
int polaris_hip_device_identity(int index, PolarisDeviceIdentity *out);
/* hipDeviceCanAccessPeer(device, peer_device) for two device indices of this process: *can = 1 / 0 (0 for device == peer_device, as HIP
 * answers).  What decides between the direct peer read and the staged copy of polaris_hip_merge (tracer/opencl/tracer.go:279-286: the
 * reference's devices share one context and the runtime migrates the buffer). */
int polaris_hip_can_access_peer(int device, int peer_device, int *can);

int polaris_hip_create(int device_index, polaris_hip_tracer **out);
void polaris_hip_destroy(polaris_hip_tracer *h);

/* Text of the last error on this handle (or, with h == NULL, of the calling thread's last
 * failed create/device call).  Valid until the next call on the same handle/thread. */
const char *polaris_hip_last_error(polaris_hip_tracer *h);

/* (Re)allocate every frame-sized buffer; clears the frame accumulator. */
int polaris_hip_resize(polaris_hip_tracer *h, uint32_t frame_w, uint32_t frame_h);

/* Validate, re-lay-out for traversal and upload the scene.  Everything is copied. */
int polaris_hip_upload_scene(polaris_hip_tracer *h, const PolarisSceneView *scene);

/* eye[3]; frustum[16] = corner rays TL, TR, BL, BR as float4 (scene.Camera.Frustrum). */
int polaris_hip_set_camera(polaris_hip_tracer *h, const float eye[3], const float frustum[16]);

/* Tunables (not part of the reference interface).  Keys:
 *   "samples_per_batch"  samples traced concurrently as one wavefront batch (default: auto)
 *   "exact_accumulate"   1 = add every contribution straight into the trace accumulator in
 *                        the reference's order (forces one sample per batch; bit-exact with
 *                        the CPU oracle), 0 = per-path radiance + ordered resolve (default)
 *   "packet_primary"     1 = wave-packet traversal for primary rays, 0 = per-ray, -1 = by scene
 *                        (default: packets for single-instance scenes of up to 32 K triangles,
 *                        except where the tiny-scene mode keeps the triangle records in LDS)
 *   "lds_tris"           NEXT upload, tiny-scene mode: triangle records kept in LDS beside the tree
 *                        (default -1: as many as fit half a CU's LDS, none if that is under half
 *                        of them; 0 = none)
 *   "tiny_one"           NEXT upload: 0 = the general tiny-scene kernel even where the variant for
 *                        single-instance scenes with bounding boxes applies (default 1)
 *   "time_kernels"       1 = bracket every kernel with HIP events (polaris_hip_kernel_ms)
 *   "overlap"            batches in flight on separate streams (1-8, default 4)
 *   "max_leaf_tris"      applies to the NEXT upload_scene: triangle leaves with more triangles
 *                        than this are subdivided where a surface-area split pays (default -1:
 *                        2 up to 32 K triangles, 4 above; 0 = keep the caller's leaves).  Never
 *                        changes a result: DESIGN.md 2 (HBM data layout)
 *   further A/B switches of the kernels ("traversal", "node_mode", "packet_shadow", "shade_wave",
 *   "shade_wave_from", "shade_sort", "shade_wgs_per_cu", "stage_lds", "trace_wgs_per_cu", "trace_grid", "hit12", "o12", "ipc_staged"): see
 *   DESIGN.md 3.  Apart from "exact_accumulate" (the order of the float sums) no option changes a
 *   result; an unknown key is POLARIS_E_BAD_ARGUMENT.  The batch size chosen automatically
 *   ("samples_per_batch" = 0) is clamped by the FREE device memory, and with it the order of the
 *   batched per-pixel float sums: non-exact output is bit-reproducible on one machine state, not
 *   across machines (exact_accumulate = 1 is, everywhere). */
int polaris_hip_set_option(polaris_hip_tracer *h, const char *key, int64_t value);

/*
 * Tracer.Trace: clears the frame accumulator if req->accumulated_samples == 0, clears the
 * trace accumulator, then traces req->samples_per_pixel samples over rows
 * [block_y, block_y+block_h).  The host PRNG draws of the reference (Go math/rand: one per
 * sample, tracer.go:222, plus one per bounce, pipeline.go:146) are passed in explicitly:
 *   seeds[s*(1+num_bounces)]       camera seed of sample s
 *   seeds[s*(1+num_bounces)+1+b]   shade seed of bounce b of sample s
 * n_seeds >= samples_per_pixel*(1+num_bounces).  Blocks until the device is done (the
 * reference's Trace is synchronous: every launch ends in clFinish, device/kernel.go:124).
 * stats may be NULL.
 */
int polaris_hip_trace(polaris_hip_tracer *h, const PolarisBlockRequest *req, const uint32_t *seeds,
                      size_t n_seeds, PolarisTraceStats *stats);

/* Tracer.MergeOutput: dst.frameAccumulator[rows of req] += src.traceAccumulator[rows of req].
 * Asynchronous like the reference (Exec1DNoWait, resources.go:119): queued on dst's MERGE stream -- the
 * stream that owns the frame accumulator (the Reset stage, merges) -- under a mutex of its own, so it does
 * not wait for a Trace running on dst (renderer/default.go:188-191 merges from the secondaries' goroutines
 * while the primary still traces); completed by polaris_hip_sync_framebuffer(dst).  The caller orders it
 * after src's Trace (synchronous) and after the START of dst's Trace of the same frame, which clears the
 * frame accumulator when accumulated_samples == 0 (the reference races there; polaris_amd/host/renderer.cpp
 * waits).  src may live on another GPU of the same process (peer access over xGMI, falling back to a
 * staged peer copy).  src must not be handed its next Trace's rows to overwrite before this merge has read them: the
 * library orders that on the device (src's next Trace waits for an event dst's merge stream records behind the read), so
 * the host need not sync in between. */
int polaris_hip_merge(polaris_hip_tracer *dst, polaris_hip_tracer *src, const PolarisBlockRequest *req);

/* One-process-per-GPU variant of the same exchange (bench.py under torch.distributed):
 * export copies the block's rows of the trace accumulator (block_h*frame_w float4) to a device
 * buffer the caller owns; merge_device adds such a strip, resident on dst's device, into
 * dst's frame accumulator. */
int polaris_hip_export_block(polaris_hip_tracer *h, const PolarisBlockRequest *req, void *device_dst);
int polaris_hip_merge_device(polaris_hip_tracer *dst, const void *device_rows, const PolarisBlockRequest *req);

/*
 * Cross-PROCESS merge: peer reads over HIP IPC (one process per GPU, the driver's launch contract).
 *
 * The reference gives all devices ONE shared OpenCL context, so the primary's aggregateAccumulator kernel reads a
 * secondary's traceAccumulator cl_mem directly (renderer/default.go:225-229, tracer/opencl/tracer.go:279-286,
 * resources.go:108-124).  Across processes the same read goes through an IPC mapping of the secondary's buffer:
 *
 *   secondary                                             primary
 *   polaris_hip_ipc_export(h, depth, &blob)   -- once, after resize; the blob is plain bytes for any channel -->
 *                                                         polaris_hip_ipc_open(dst, &blob, &peer)
 *   polaris_hip_trace(h, ..)      writes ring slot s = polaris_hip_trace_slot(h)
 *      -- "frame f is in slot s" (a host message AFTER Trace returned: Trace is synchronous) -->
 *                                                         polaris_hip_merge_ipc(dst, peer, s, req)   k_aggregate over the
 *                                                             peer-mapped rows, on dst's merge stream (xGMI peer read)
 *
 * ipc_export turns the tracer's trace accumulator into a RING of `depth` frame-sized buffers (1..POLARIS_IPC_MAX_DEPTH):
 * every later Trace writes the next slot, so the primary may still be reading frame f's rows while the secondary traces
 * frame f + 1 -- no copy, no staging strip.  The caller's protocol must guarantee that the primary has finished reading a
 * slot (its merge completed: sync_framebuffer) before the secondary's Trace comes round to it again; with the exchange one
 * frame behind the tracing that needs depth 3 (polaris_amd/distributed.py: PeerExchange states the argument).  The blob
 * also carries one inter-process event PER RING SLOT, recorded at the end of the Trace that wrote the slot; merge_ipc(slot)
 * makes the merge stream wait for that slot's event (device-side ordering on top of the host message; has_event = 0 if the
 * runtime could not export them).  Because a slot's event is only re-recorded when a Trace comes round to the slot again --
 * which the protocol above rules out while the slot may still be read -- the wait names exactly the Trace whose rows are
 * read, however far the secondary has run ahead (ABI 3 had one event for the whole ring, re-recorded by every Trace).  A resize
 * invalidates the export: peers close, the tracer exports again.  hipIpcOpenMemHandle cannot open a handle in the process
 * that created it: tracers of ONE process use polaris_hip_merge / polaris_hip_merge_slot.
 */
#define POLARIS_IPC_MAX_DEPTH 4
typedef struct PolarisIpcExport {
	uint32_t abi_version, depth, frame_w, frame_h;
	int32_t device;     /* HIP device index in the exporting process */
	uint32_t pid;       /* exporting process (diagnostics; opening in the same process is refused) */
	uint32_t has_event; /* 1: `event[i]` holds a hipIpcEventHandle_t for every slot i < depth */
	uint32_t reserved;
	uint8_t mem[POLARIS_IPC_MAX_DEPTH][64];   /* hipIpcMemHandle_t per ring slot */
	uint8_t event[POLARIS_IPC_MAX_DEPTH][64]; /* hipIpcEventHandle_t per ring slot: recorded at the end of the Trace that wrote the slot */
	char pci_bus_id[32];                      /* ABI 5: the exporting GPU (device indices mean nothing across processes; "" if unknown) */
} PolarisIpcExport;
typedef struct polaris_hip_peer polaris_hip_peer; /* opaque: another process's trace accumulator ring, mapped here */

int polaris_hip_ipc_export(polaris_hip_tracer *h, uint32_t depth, PolarisIpcExport *out);
int polaris_hip_ipc_open(polaris_hip_tracer *dst, const PolarisIpcExport *peer_export, polaris_hip_peer **out);
int polaris_hip_ipc_close(polaris_hip_tracer *dst, polaris_hip_peer *peer); /* waits for dst's merge stream first */
/* What a mapped ring really is (ABI 5): the exporter's GPU by PCI bus id, whether that is dst's OWN GPU (the mapping is then a second
 * mapping of local memory: ranks sharing a device, the one-GPU tests) or ANOTHER one (reads cross xGMI / PCIe), the exporter's GPU as a
 * device index of this process (-1: not visible here, e.g. under a per-rank visibility mask) and hipDeviceCanAccessPeer towards it
 * (-1: unknown).  ipc_open REFUSES (POLARIS_E_UNSUPPORTED) a ring on another GPU that this process cannot see at all -- there is no way to
 * tell whether the mapping would be reachable, and a kernel that reads an unreachable mapping faults the GPU; the caller falls back to
 * its strip transfers -- and marks a ring on a visible GPU without peer access for the staged path (POLARIS_MERGE_IPC_STAGED; option
 * "ipc_staged" = 1 forces that path for the peers opened afterwards: a testing aid).  The caller sets struct_size. */
typedef struct PolarisPeerInfo {
	uint32_t struct_size;
	uint32_t pid;             /* exporting process */
	int32_t exporter_device;  /* its HIP device index in THAT process */
	int32_t local_device;     /* the same GPU as a device index of this process; -1 = not visible here */
	int32_t same_device;      /* 1: the ring lives on dst's own GPU; 0: on another GPU; -1: unknown (no bus id on either side) */
	int32_t can_access_peer;  /* hipDeviceCanAccessPeer(dst's device, local_device); -1 = unknown */
	uint32_t depth, has_events;
	char pci_bus_id[32];
	int32_t staged;           /* 1: merges from this ring go through the staging strip (no peer access, or forced); 0: read where it lies */
	int32_t reserved;
} PolarisPeerInfo;
int polaris_hip_peer_info(polaris_hip_peer *peer, PolarisPeerInfo *out);
/* dst.frameAccumulator[rows of req] += peer.traceAccumulator ring[slot][rows of req]; asynchronous on dst's merge stream
 * like polaris_hip_merge, completed by polaris_hip_sync_framebuffer(dst). */
int polaris_hip_merge_ipc(polaris_hip_tracer *dst, polaris_hip_peer *peer, uint32_t slot, const PolarisBlockRequest *req);
/* The ring slot the last Trace wrote (0 without a ring; a Trace that fails before it has queued anything leaves it alone),
 * and polaris_hip_merge from a given slot of a tracer of THIS process (a primary that merges its own block one frame late
 * reads the slot of that frame, not the newest).  Like polaris_hip_merge, a merge_slot with src != dst is fenced on the
 * device against src's next Trace. */
int polaris_hip_trace_slot(polaris_hip_tracer *h, uint32_t *slot);
int polaris_hip_merge_slot(polaris_hip_tracer *dst, polaris_hip_tracer *src, uint32_t slot, const PolarisBlockRequest *req);

/* Which branch the merges onto `dst` took since it was created (ABI 5), counted per call under dst's merge lock:
 *   LOCAL         polaris_hip_merge / _merge_slot, source on dst's own device (src == dst included)
 *   PEER_ACCESS   ... source on another GPU of this process, read directly (hipDeviceEnablePeerAccess: xGMI peer read)
 *   STAGED        ... source on another GPU, no peer access: hipMemcpyPeerAsync into a staging strip, then added
 *   IPC_LOCAL     polaris_hip_merge_ipc, the peer's ring lives on dst's own GPU
 *   IPC_PEER      polaris_hip_merge_ipc, the peer's ring lives on another GPU and is read directly (IPC_UNKNOWN: the exporter gave no bus id)
 *   IPC_STAGED    polaris_hip_merge_ipc, the ring lives on a GPU this one has NO peer access to (hipDeviceCanAccessPeer = 0): the rows are
 *                 copied into a staging strip by the runtime (hipMemcpyAsync, which may go through the host) and added from there --
 *                 a kernel is never pointed at memory the device cannot reach
 *   DEVICE_STRIP  polaris_hip_merge_device (a strip some transport delivered: bench.py's fallback)
 * bench.py reports them in config.exchange_detail: a fallback that ran silently shows in the numbers' own line. */
#define POLARIS_MERGE_LOCAL 0
#define POLARIS_MERGE_PEER_ACCESS 1
#define POLARIS_MERGE_STAGED 2
#define POLARIS_MERGE_IPC_LOCAL 3
#define POLARIS_MERGE_IPC_PEER 4
#define POLARIS_MERGE_IPC_UNKNOWN 5
#define POLARIS_MERGE_DEVICE_STRIP 6
#define POLARIS_MERGE_IPC_STAGED 7
#define POLARIS_MERGE_BRANCHES 8
int polaris_hip_merge_counts(polaris_hip_tracer *dst, uint64_t counts[POLARIS_MERGE_BRANCHES]);

/* The pipeline's Reset stage on its own (tracer/opencl/tracer.go:208-213: clearAccumulator(frameAccumulator)).
 * Trace runs it whenever accumulated_samples == 0.  A host that merges a frame's blocks only after the NEXT
 * frame's Trace has started (bench.py keeps the strip exchange one frame behind the tracing) calls it before
 * merging, so that exactly one frame's blocks are ever summed.  Asynchronous on the handle's stream. */
int polaris_hip_reset_frame(polaris_hip_tracer *h);

/* Ordering merges from other threads against the Reset stage without waiting for a whole Trace.  The reset epoch of a
 * tracer counts the Traces with accumulated_samples == 0 (and reset_frame calls) that have queued the clear of the frame
 * accumulator -- or have failed before they could.  A frame loop reads the primary's epoch before it hands out the
 * frame's blocks; a worker that is about to merge into the primary first waits for the epoch to pass that value: its
 * merge then lands behind this frame's clear, while the primary is still tracing (the reference leaves this to chance,
 * renderer/default.go:188-191 against tracer.go:208-213; polaris_amd/host/renderer.cpp shows the use). */
int polaris_hip_reset_epoch(polaris_hip_tracer *h, uint64_t *epoch);
int polaris_hip_wait_reset(polaris_hip_tracer *h, uint64_t epoch); /* POLARIS_E_TIMEOUT after 120 s without that Reset */

/* Tracer.SyncFramebuffer: wait for pending merges, then tonemapSimpleReinhard over rows of
 * req with weight 1/(accumulated_samples+samples_per_pixel) into the RGBA8 frame buffer. */
int polaris_hip_sync_framebuffer(polaris_hip_tracer *h, const PolarisBlockRequest *req);

/* frame_w*frame_h*4 bytes RGBA8 (pipeline.go:226-232). */
int polaris_hip_read_framebuffer(polaris_hip_tracer *h, uint8_t *rgba, size_t n_bytes);

/* Radiance-level read-back for parity tests (no reference counterpart): which = 0 trace
 * accumulator, 1 frame accumulator; n_floats = frame_w*frame_h*4 (float3 at stride 4). */
int polaris_hip_read_accumulator(polaris_hip_tracer *h, int which, float *out, size_t n_floats);

/* Test tap: generate + intersect the primary rays of one sample with camera seed `seed` and
 * return rays [N][8], hit flags [N], (w,u,v,t) [N][4] and (instance, triangle) [N][2];
 * N = frame_w*block_h.  Any output pointer may be NULL. */
int polaris_hip_tap_primary(polaris_hip_tracer *h, const PolarisBlockRequest *req, uint32_t seed,
                            float *rays, int32_t *hit, float *wuvt, int32_t *tri);

/* Function-level test taps: the device-side samplers and BxDFs of the uploaded scene on caller-supplied
 * inputs, one probe per row of `in`, through the same table staging the shade kernels use.
 *   kind 0, BxDF of material LEAF node `index` (bxdf/bxdf.cl:31-105):
 *        in [n][13] = normal[3] uv[2] in_dir[3] sample[2] eval_dir[3]
 *        out[n][11] = bxdfGetSample value[3], sampled dir[3], pdf | bxdfGetPdf(eval_dir) | bxdfEval(eval_dir)[3]
 *   kind 1, texture `index` (samplers/texture_sampler.cl:14-252):
 *        in [n][2] = uv;  out[n][7] = texGetSample3f[3] | texGetSample1f | texGetBumpSample3f[3]
 *   kind 2, emissive `index` (samplers/emissive_sampler.cl:176-223):
 *        in [n][11] = point[3] normal[3] sample[2] pdf_dir[3]
 *        out[n][9]  = emissiveGetSample radiance[3] dir[3] pdf dist | emissiveGetPdf(pdf_dir)
 *   kind 3, material tree rooted at node `index` (matSelectNode, samplers/material_sampler.cl:21-95):
 *        in [n][8]  = normal[3] uv[2] | shading PRNG state[2] and path dispersion flags as bit patterns
 *        out[n][18] = selected leaf type (bits), int / ext IOR after the dispersion override | normal after bump / normal
 *                     maps[3] | tint[3] | path flags (bits) | PRNG state[2] (bits) | the leaf's k[3] and t[3] */
int polaris_hip_probe(polaris_hip_tracer *h, int kind, uint32_t index, uint32_t n, const float *in, float *out);

/* rayIntersectionQuery (any_hit = 0, kernels/intersect.cl:184-347) or rayIntersectionTest (any_hit != 0, :26-180) over n
 * arbitrary rays [n][8] = origin.xyz, maxDist, dir.xyz, unused, through the traversal kernel the options select
 * ("traversal", "node_mode", "packet_primary" = 1 / "packet_shadow" > 0 for the wave-packet kernel): hit[n]; for closest
 * hits also (w,u,v,t) [n][4] and the scene triangle index [n] (-1 = miss); either may be NULL. */
int polaris_hip_probe_intersect(polaris_hip_tracer *h, const float *rays, uint32_t n, int any_hit, int32_t *hit,
                                float *wuvt, int32_t *tri);

/* Self-test of the one place where the kernels do NOT use the compiler's correctly rounded division: 1 / det of the
 * triangle tests is v_rcp_f32 + one Newton step (3 instructions instead of 11).  Sweeps all 2^32 float bit patterns x on the
 * device and counts those whose result differs from 1.0f / x, inside and outside lo <= |x| <= hi (`sample` = one differing
 * pattern inside, may be NULL).  The claim the kernels rest on: none inside [2^-126, 2^126).  (The directions in
 * polaris_hip_probe_intersect must be finite with components <= 2^10, origins <= 2^40; scenes: DESIGN.md 2.) */
int polaris_hip_selftest_rcp(polaris_hip_tracer *h, float lo, float hi, uint64_t *mismatches_inside, uint64_t *mismatches_outside,
                             uint32_t *sample);

/*
 * BVH construction on the device -- an ALTERNATIVE producer of the scene's two-level BVH (the stretch; the
 * reference's own builder, asset/compiler/bvh/bvh_builder.go:100-308, scores ~1024 / (depth + 1) candidate planes per axis with
 * one goroutine each and is restated for the CPU in polaris_amd/host/scene_compiler.cpp).  Two algorithms (`algorithm`):
 *   POLARIS_BVH_SAH (0, the default)  binned surface-area heuristic, built level by level: 16 bins per axis, the cheapest of the
 *       3 x 15 planes per node (the criterion of bvh_builder.go:162-211, plane count aside), a node no plane separates is halved by
 *       position.  Frames trace within 1 % of the CPU-built tree; 1 M triangles build in ~7 ms.  Depth is bounded by
 *       the float range of the centroid extents plus log2(n) -- not by log2(n) alone: geometrically spaced outliers make a chain --
 *       and a tree deeper than the 32-entry traversal stack is refused by upload_scene, whoever built it.
 *   POLARIS_BVH_LBVH (1)  linear BVH: Morton order of the centroids (one radix sort), all inner nodes at once (Karras 2012), boxes
 *       fitted bottom-up, subtrees of up to max_leaf_tris triangles collapsed into the reference's kind of leaf.  2-6 x faster to
 *       build, deeper and looser: frames take 26-55 % longer on it, and a mesh that packs many triangles into one cell of the 30-bit
 *       grid can exceed the traversal stack (upload_scene then refuses the scene: use the SAH builder or a CPU producer).
 * One tree per mesh over its triangles, one over the instances' world boxes with one instance per leaf (compiler.go:88-103).  Output in
 * the reference's encoding (PolarisBvhNode; node 0 = the scene's root), ready for PolarisSceneView once the caller has put the
 * triangle arrays in the new order:
 *   tri_order[new position] = old triangle index (a permutation within every mesh's range; emissive tri_index values follow it);
 *   mesh_root[m] = the node a PolarisMeshInstance.bvh_root of mesh m must name.
 * nodes_capacity >= 2 * (num_instances + num_triangles).  The tree differs from the reference compiler's (another algorithm) --
 * parity is defined on the uploaded arrays, and any tree whose boxes bound their contents is traversed correctly.
 * device_ms (may be NULL) = the build on the device, vertices resident, without the read-back.  No tracer handle is involved.
 * in->struct_size must be sizeof(PolarisBvhBuildInput) (ABI 5): POLARIS_E_BAD_ARGUMENT otherwise.
 */
typedef struct PolarisBvhBuildInput {
	uint32_t struct_size;           /* = sizeof(PolarisBvhBuildInput): a caller built against another layout is refused (ABI 5) */
	const float *vertices;          /* float4 per vertex, 3 per triangle, as PolarisSceneView.vertices */
	uint32_t num_triangles;
	const uint32_t *mesh_first_tri; /* [num_meshes]: mesh m owns triangles [first, first + count); the ranges tile [0, num_triangles) in order */
	const uint32_t *mesh_num_tris;
	uint32_t num_meshes;
	const float *instance_boxes;    /* [num_instances][6]: world-space min.xyz, max.xyz of every mesh instance (the scene reader's boxes) */
	const uint32_t *instance_mesh;  /* [num_instances] */
	uint32_t num_instances;
	uint32_t max_leaf_tris;         /* 1..15 */
	uint32_t algorithm;             /* POLARIS_BVH_SAH (0, the default: binned surface-area heuristic, level by level) or POLARIS_BVH_LBVH (1: linear BVH, 2-6 x faster to build, 26-55 % slower to trace); ABI 4 */
} PolarisBvhBuildInput;
#define POLARIS_BVH_SAH 0u
#define POLARIS_BVH_LBVH 1u
int polaris_hip_build_bvh(int device, const PolarisBvhBuildInput *in, PolarisBvhNode *nodes, uint32_t nodes_capacity, uint32_t *num_nodes,
                          uint32_t *tri_order, uint32_t *mesh_root, double *device_ms);
const char *polaris_hip_build_bvh_error(void); /* text of the calling thread's last polaris_hip_build_bvh failure */

/* With option time_kernels=1: accumulated device milliseconds and launch count of the named
 * timer since the last call for that name.  Timers: "generate", "intersect_packet" (camera rays through
 * the wave-packet kernel), "intersect" (closest hit), "shade_first" / "shade_sort" /
 * "shade_plain" / "shade_wave" (one per shade kernel symbol), "scan", "occlusion", "fold" (the batch's NEE records into the
 * per-path radiance), "resolve", "aggregate", "tonemap". */
int polaris_hip_kernel_ms(polaris_hip_tracer *h, const char *kernel, double *ms, uint64_t *launches);

/* The kernel symbol (as rocprofv3 prints it, e.g. "pol::k_trace<false, 16, 2>") the named timer last
 * bracketed; "" if it has not run.  Measurement aid: bench.py names its roofline objects by it. */
int polaris_hip_kernel_symbol(polaris_hip_tracer *h, const char *kernel, char symbol[128]);

/* Shading events of the last Trace per bounce: counts[4 b + 0..2] = shaded hits, shaded misses, emitter
 * hits of the shade step of bounce b (their sums are PolarisTraceStats' totals), counts[4 b + 3] = which
 * shade timer that step ran under (0 shade_first, 1 shade_sort, 2 shade_plain, 3 shade_wave).
 * n_counts >= 4 * POLARIS_MAX_BOUNCES.  Measurement aid: algorithmic bytes per shade kernel symbol. */
int polaris_hip_shade_counts(polaris_hip_tracer *h, uint64_t *counts, size_t n_counts);

#ifdef __cplusplus
}
#endif
#endif /* POLARIS_HIP_H */
