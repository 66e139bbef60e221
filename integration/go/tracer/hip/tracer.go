// Package hip is the MI355X (gfx950) tracer backend for polaris.  It implements
// tracer.Tracer (tracer/tracer.go:80-111) by calling the C ABI of libpolaris_hip.so
// (include/polaris_hip.h) through cgo, and replaces tracer/opencl + tracer/opencl/device.
//
// NOTE: written without a Go toolchain (the build image has none); it has not been compiled.
// Every method is a thin call into the C ABI, which is what is built and tested.
package hip

/*
#cgo CFLAGS: -I${SRCDIR}/../../../../include
#cgo LDFLAGS: -L${SRCDIR}/../../../../polaris_amd/lib -lpolaris_hip -Wl,-rpath,${SRCDIR}/../../../../polaris_amd/lib
#include <stdlib.h>
#include "polaris_hip.h"
*/
import "C"

import (
	"errors"
	"fmt"
	"math/rand"
	"sync"
	"time"
	"unsafe"

	"github.com/achilleasa/polaris/asset/scene"
	"github.com/achilleasa/polaris/tracer"
)

var (
	// Same sentinel the OpenCL backend returns (tracer/opencl/errors.go).
	ErrNoSceneData = errors.New("hip tracer: no scene data uploaded")
)

// Tracer is one MI355X behind the tracer.Tracer interface.
type Tracer struct {
	sync.Mutex
	id     string
	dev    Device
	handle *C.polaris_hip_tracer
	stats  *tracer.Stats

	// Asynchronous updates are grouped by type; the latest wins (tracer/opencl/tracer.go:150-158).
	changeBuffer map[tracer.ChangeType]interface{}
}

// NewTracer mirrors opencl.NewTracer (tracer/opencl/tracer.go:58-72).
func NewTracer(id string, dev Device) (tracer.Tracer, error) {
	return &Tracer{id: id, dev: dev, stats: &tracer.Stats{}, changeBuffer: make(map[tracer.ChangeType]interface{})}, nil
}

func (tr *Tracer) Id() string          { return tr.id }
func (tr *Tracer) Flags() tracer.Flag  { return tracer.Local }
func (tr *Tracer) Speed() uint32       { return tr.dev.Speed }
func (tr *Tracer) Stats() *tracer.Stats { return tr.stats }

func (tr *Tracer) lastError() error {
	return fmt.Errorf("hip tracer (%s): %s", tr.dev.Name, C.GoString(C.polaris_hip_last_error(tr.handle)))
}

func (tr *Tracer) check(rc C.int) error {
	switch rc {
	case C.POLARIS_OK:
		return nil
	case C.POLARIS_E_NO_SCENE_DATA:
		return ErrNoSceneData
	default:
		return tr.lastError()
	}
}

// Init creates the device-side tracer (tracer/opencl/tracer.go:95-117).
func (tr *Tracer) Init() error {
	tr.Lock()
	defer tr.Unlock()
	if rc := C.polaris_hip_create(C.int(tr.dev.Index), &tr.handle); rc != C.POLARIS_OK {
		return fmt.Errorf("hip tracer (%s): %s", tr.dev.Name, C.GoString(C.polaris_hip_last_error(nil)))
	}
	return nil
}

// Close releases every device resource (tracer/opencl/tracer.go:120-145).
func (tr *Tracer) Close() {
	tr.Lock()
	defer tr.Unlock()
	if tr.handle != nil {
		C.polaris_hip_destroy(tr.handle)
		tr.handle = nil
	}
}

// UpdateState queues a change; synchronous updates are committed at once
// (tracer/opencl/tracer.go:150-158).
func (tr *Tracer) UpdateState(mode tracer.UpdateMode, changeType tracer.ChangeType, data interface{}) (time.Duration, error) {
	tr.changeBuffer[changeType] = data
	if mode == tracer.Synchronous {
		return tr.commitChanges()
	}
	return 0, nil
}

// commitChanges mirrors tracer/opencl/tracer.go:161-192.  All scene slices are COPIED by the
// library before the call returns: no Go pointer is retained (cgo pointer rule), unlike the
// reference's CL_MEM_USE_HOST_PTR aliasing (device/buffer.go:98-104).
func (tr *Tracer) commitChanges() (time.Duration, error) {
	if len(tr.changeBuffer) == 0 {
		return 0, nil
	}
	start := time.Now()
	for changeType, data := range tr.changeBuffer {
		var err error
		switch changeType {
		case tracer.FrameDimensions:
			dims := data.([2]uint32)
			err = tr.check(C.polaris_hip_resize(tr.handle, C.uint32_t(dims[0]), C.uint32_t(dims[1])))
		case tracer.SceneData:
			err = tr.uploadScene(data.(*scene.Scene))
		case tracer.CameraData:
			cam := data.(*scene.Camera)
			eye := [3]C.float{C.float(cam.Position[0]), C.float(cam.Position[1]), C.float(cam.Position[2])}
			var fr [16]C.float
			for c := 0; c < 4; c++ { // Frustrum = TL, TR, BL, BR (asset/scene/camera.go:125-141)
				for k := 0; k < 4; k++ {
					fr[4*c+k] = C.float(cam.Frustrum[c][k])
				}
			}
			err = tr.check(C.polaris_hip_set_camera(tr.handle, &eye[0], &fr[0]))
		default:
			err = fmt.Errorf("unsupported change type %d", changeType)
		}
		if err != nil {
			return time.Since(start), err
		}
	}
	tr.changeBuffer = make(map[tracer.ChangeType]interface{})
	tr.stats.UpdateTime = time.Since(start)
	return tr.stats.UpdateTime, nil
}

// uploadScene hands the ten flat slices of scene.Scene (asset/scene/optimized_scene.go:167-190)
// to the library; the Go structs are layout-identical to include/polaris_types.h.
func (tr *Tracer) uploadScene(sc *scene.Scene) error {
	var v C.PolarisSceneView
	if n := len(sc.BvhNodeList); n > 0 {
		v.bvh_nodes, v.num_bvh_nodes = (*C.PolarisBvhNode)(unsafe.Pointer(&sc.BvhNodeList[0])), C.uint32_t(n)
	}
	if n := len(sc.MeshInstanceList); n > 0 {
		v.mesh_instances, v.num_mesh_instances = (*C.PolarisMeshInstance)(unsafe.Pointer(&sc.MeshInstanceList[0])), C.uint32_t(n)
	}
	if n := len(sc.MaterialNodeList); n > 0 {
		v.material_nodes, v.num_material_nodes = (*C.PolarisMaterialNode)(unsafe.Pointer(&sc.MaterialNodeList[0])), C.uint32_t(n)
	}
	if n := len(sc.EmissivePrimitives); n > 0 {
		v.emissives, v.num_emissives = (*C.PolarisEmissive)(unsafe.Pointer(&sc.EmissivePrimitives[0])), C.uint32_t(n)
	}
	if n := len(sc.TextureData); n > 0 {
		v.texture_data, v.texture_data_bytes = (*C.uint8_t)(unsafe.Pointer(&sc.TextureData[0])), C.uint32_t(n)
	}
	if n := len(sc.TextureMetadata); n > 0 {
		v.texture_meta, v.num_textures = (*C.PolarisTextureMetadata)(unsafe.Pointer(&sc.TextureMetadata[0])), C.uint32_t(n)
	}
	if n := len(sc.MaterialIndex); n > 0 {
		v.vertices = (*C.float)(unsafe.Pointer(&sc.VertexList[0]))
		v.normals = (*C.float)(unsafe.Pointer(&sc.NormalList[0]))
		v.uvs = (*C.float)(unsafe.Pointer(&sc.UvList[0]))
		v.material_index = (*C.uint32_t)(unsafe.Pointer(&sc.MaterialIndex[0]))
		v.num_triangles = C.uint32_t(n)
	}
	v.scene_diffuse_mat_index = C.int32_t(sc.SceneDiffuseMatIndex)
	v.scene_emissive_mat_index = C.int32_t(sc.SceneEmissiveMatIndex)
	return tr.check(C.polaris_hip_upload_scene(tr.handle, &v))
}

func toC(req *tracer.BlockRequest) C.PolarisBlockRequest {
	// tracer.BlockRequest (tracer/tracer.go:6-34) and PolarisBlockRequest have the same 12
	// fields in the same order; copied field by field to stay independent of Go's padding rules.
	return C.PolarisBlockRequest{
		frame_w: C.uint32_t(req.FrameW), frame_h: C.uint32_t(req.FrameH),
		block_x: C.uint32_t(req.BlockX), block_y: C.uint32_t(req.BlockY),
		block_w: C.uint32_t(req.BlockW), block_h: C.uint32_t(req.BlockH),
		samples_per_pixel: C.uint32_t(req.SamplesPerPixel), num_bounces: C.uint32_t(req.NumBounces),
		min_bounces_for_rr: C.uint32_t(req.MinBouncesForRR), exposure: C.float(req.Exposure),
		seed: C.uint32_t(req.Seed), accumulated_samples: C.uint32_t(req.AccumulatedSamples),
	}
}

// Trace mirrors tracer/opencl/tracer.go:194-247.  The reference draws one math/rand value per
// sample (:222) and one per bounce (pipeline.go:146), interleaved; the same draws, in the same
// order, are made here and passed down as the seed list.
func (tr *Tracer) Trace(blockReq *tracer.BlockRequest) (time.Duration, error) {
	start := time.Now()
	if _, err := tr.commitChanges(); err != nil {
		return time.Since(start), err
	}
	stride := 1 + int(blockReq.NumBounces)
	seeds := make([]C.uint32_t, int(blockReq.SamplesPerPixel)*stride)
	for s := 0; s < int(blockReq.SamplesPerPixel); s++ {
		blockReq.Seed = rand.Uint32()
		seeds[s*stride] = C.uint32_t(blockReq.Seed)
		for b := 0; b < int(blockReq.NumBounces); b++ {
			seeds[s*stride+1+b] = C.uint32_t(rand.Uint32())
		}
	}
	creq := toC(blockReq)
	var seedPtr *C.uint32_t
	if len(seeds) > 0 {
		seedPtr = &seeds[0]
	}
	if err := tr.check(C.polaris_hip_trace(tr.handle, &creq, seedPtr, C.size_t(len(seeds)), nil)); err != nil {
		return time.Since(start), err
	}
	blockReq.AccumulatedSamples += blockReq.SamplesPerPixel // tracer.go:240
	tr.stats.BlockW = blockReq.BlockW
	tr.stats.BlockH = blockReq.BlockH
	tr.stats.RenderTime = time.Since(start)
	return tr.stats.RenderTime, nil
}

// MergeOutput mirrors tracer/opencl/tracer.go:279-286.
func (tr *Tracer) MergeOutput(other tracer.Tracer, blockReq *tracer.BlockRequest) (time.Duration, error) {
	start := time.Now()
	src, ok := other.(*Tracer)
	if !ok {
		return 0, fmt.Errorf("merge failed: unsupported tracer instance")
	}
	creq := toC(blockReq)
	return time.Since(start), tr.check(C.polaris_hip_merge(tr.handle, src.handle, &creq))
}

// ResetEpoch / WaitReset order a merge from another goroutine behind this tracer's Reset stage (Trace clears the frame
// accumulator when it starts, tracer/opencl/tracer.go:208-213) without waiting for its whole Trace: read the primary's
// epoch before the frame's blocks are handed out, and WaitReset(epoch) in a secondary's worker before it merges
// (INTEGRATION.md section 3).
func (tr *Tracer) ResetEpoch() uint64 {
	var e C.uint64_t
	C.polaris_hip_reset_epoch(tr.handle, &e)
	return uint64(e)
}

// A non-nil error (POLARIS_E_TIMEOUT: the awaited Reset never came) means the block must NOT be merged: it would land on
// an accumulator that was not cleared for this frame.
func (tr *Tracer) WaitReset(epoch uint64) error {
	return tr.check(C.polaris_hip_wait_reset(tr.handle, C.uint64_t(epoch)))
}

// SyncFramebuffer mirrors tracer/opencl/tracer.go:250-276 (wait, then tone-map).
func (tr *Tracer) SyncFramebuffer(blockReq *tracer.BlockRequest) (time.Duration, error) {
	start := time.Now()
	creq := toC(blockReq)
	return time.Since(start), tr.check(C.polaris_hip_sync_framebuffer(tr.handle, &creq))
}

// ReadFrameBuffer is what opencl.SaveFrameBuffer reads (tracer/opencl/pipeline.go:226-232).
func (tr *Tracer) ReadFrameBuffer(pix []uint8) error {
	if len(pix) == 0 {
		return nil
	}
	return tr.check(C.polaris_hip_read_framebuffer(tr.handle, (*C.uint8_t)(unsafe.Pointer(&pix[0])), C.size_t(len(pix))))
}

// ---- one tracer per PROCESS (INTEGRATION.md section 3b): the merge as a peer read through HIP IPC --------------------------

// IpcExport turns the trace accumulator into a ring of `depth` buffers (every Trace writes the next; TraceSlot says which) and
// returns the 576-byte blob (ABI 5: it carries the exporting GPU's PCI bus id) another process opens with IpcOpen.  Plain bytes: send them over any channel.
func (tr *Tracer) IpcExport(depth uint32) ([]byte, error) {
	var x C.PolarisIpcExport
	if err := tr.check(C.polaris_hip_ipc_export(tr.handle, C.uint32_t(depth), &x)); err != nil {
		return nil, err
	}
	return C.GoBytes(unsafe.Pointer(&x), C.int(unsafe.Sizeof(x))), nil
}

// Peer is another process's trace accumulator ring, mapped on this tracer's device.
type Peer struct{ p *C.polaris_hip_peer }

func (tr *Tracer) IpcOpen(blob []byte) (*Peer, error) {
	if len(blob) != int(unsafe.Sizeof(C.PolarisIpcExport{})) {
		return nil, fmt.Errorf("hip: an IPC export is %d bytes, got %d", unsafe.Sizeof(C.PolarisIpcExport{}), len(blob))
	}
	var p *C.polaris_hip_peer
	if err := tr.check(C.polaris_hip_ipc_open(tr.handle, (*C.PolarisIpcExport)(unsafe.Pointer(&blob[0])), &p)); err != nil {
		return nil, err
	}
	return &Peer{p}, nil
}

func (tr *Tracer) IpcClose(peer *Peer) error { return tr.check(C.polaris_hip_ipc_close(tr.handle, peer.p)) }

// MergeIpc is MergeOutput from a tracer of another process: the rows of blockReq of the peer's ring slot, read where they lie.
func (tr *Tracer) MergeIpc(peer *Peer, slot uint32, blockReq *tracer.BlockRequest) (time.Duration, error) {
	start := time.Now()
	creq := toC(blockReq)
	return time.Since(start), tr.check(C.polaris_hip_merge_ipc(tr.handle, peer.p, C.uint32_t(slot), &creq))
}

// PeerInfo says what a mapped ring really is (ABI 5): the exporter's GPU by PCI bus id, whether that is this tracer's own GPU (a
// second mapping of local memory) or another one (reads cross xGMI / PCIe), and hipDeviceCanAccessPeer towards it (-1 unknown).
type PeerInfo struct {
	Pid           uint32
	PCIBusID      string
	SameDevice    int
	LocalDevice   int
	CanAccessPeer int
	HasEvents     bool
	Staged        bool // merges from this ring go through a staging strip (no peer access to its GPU): a runtime copy, then the add
}

func (p *Peer) Info() (PeerInfo, error) {
	var i C.PolarisPeerInfo
	i.struct_size = C.uint32_t(unsafe.Sizeof(i))
	if rc := C.polaris_hip_peer_info(p.p, &i); rc != C.POLARIS_OK {
		return PeerInfo{}, fmt.Errorf("hip: peer_info failed (%d)", int(rc))
	}
	return PeerInfo{Pid: uint32(i.pid), PCIBusID: C.GoString(&i.pci_bus_id[0]), SameDevice: int(i.same_device), LocalDevice: int(i.local_device),
		CanAccessPeer: int(i.can_access_peer), HasEvents: i.has_events != 0, Staged: i.staged != 0}, nil
}

// MergeBranches names the entries of MergeCounts (POLARIS_MERGE_* in polaris_hip.h).
var MergeBranches = [...]string{"local", "peer-access", "staged", "ipc-local", "ipc-peer", "ipc-unknown", "device-strip", "ipc-staged"}

// MergeCounts reports which branch the merges onto this tracer took since it was created: a staged copy where a peer read was
// expected, or a mapping of local memory where another GPU was, shows here instead of in a timing nobody can explain.
func (tr *Tracer) MergeCounts() (map[string]uint64, error) {
	var c [C.POLARIS_MERGE_BRANCHES]C.uint64_t
	if err := tr.check(C.polaris_hip_merge_counts(tr.handle, &c[0])); err != nil {
		return nil, err
	}
	out := make(map[string]uint64, len(c))
	for i, name := range MergeBranches {
		out[name] = uint64(c[i])
	}
	return out, nil
}

// TraceSlot is the ring slot the last Trace wrote (0 without a ring).
func (tr *Tracer) TraceSlot() uint32 {
	var s C.uint32_t
	C.polaris_hip_trace_slot(tr.handle, &s)
	return uint32(s)
}
