package hip

/*
#include "polaris_hip.h"
*/
import "C"

import (
	"encoding/hex"
	"fmt"
	"strings"
	"unsafe"
)

// Device describes one HIP device; replaces tracer/opencl/device.Device for selection purposes.
type Device struct {
	Index int
	Name  string
	// Speed estimate exactly as the reference computes it: compute units * MHz / 1000
	// (tracer/opencl/device/device.go:219).
	Speed uint32
}

// Devices enumerates HIP devices, dropping names that contain a blacklisted substring
// (renderer/default.go:204-224).
func Devices(blacklist []string) []Device {
	var out []Device
	n := int(C.polaris_hip_device_count())
	for i := 0; i < n; i++ {
		var name [256]C.char
		var cus, mhz C.uint32_t
		var mem C.uint64_t
		if C.polaris_hip_device_info(C.int(i), &name[0], &cus, &mhz, &mem) != C.POLARIS_OK {
			continue
		}
		d := Device{Index: i, Name: C.GoString(&name[0]), Speed: uint32(cus) * uint32(mhz) / 1000}
		skip := false
		for _, b := range blacklist {
			if b != "" && strings.Contains(d.Name, b) {
				skip = true
			}
		}
		if !skip {
			out = append(out, d)
		}
	}
	return out
}

// Identity says which PHYSICAL GPU a device index of this process is (ABI 5).  Indices are per process: under a per-rank
// visibility mask (HIP_VISIBLE_DEVICES) index 0 is a different GPU in every process -- or, misconfigured, the same one in all of
// them; the PCI bus id and the UUID tell.  The reference identifies devices by their OpenCL name inside ONE process
// (renderer/default.go:204-256); a host with one process per GPU logs this next to its timings.
type Identity struct {
	Index    int
	PCIBusID string
	UUID     string // hex; all zeros if the runtime reports none
	Name     string
	GcnArch  string
	CUs      uint32
	ClockMHz uint32
	MemBytes uint64
}

func DeviceIdentity(index int) (Identity, error) {
	var d C.PolarisDeviceIdentity
	d.struct_size = C.uint32_t(unsafe.Sizeof(d))
	if rc := C.polaris_hip_device_identity(C.int(index), &d); rc != C.POLARIS_OK {
		return Identity{}, fmt.Errorf("hip: device %d: %s", index, C.GoString(C.polaris_hip_last_error(nil)))
	}
	return Identity{Index: index, PCIBusID: C.GoString(&d.pci_bus_id[0]), UUID: hex.EncodeToString(C.GoBytes(unsafe.Pointer(&d.uuid[0]), 16)),
		Name: C.GoString(&d.name[0]), GcnArch: C.GoString(&d.gcn_arch[0]), CUs: uint32(d.compute_units), ClockMHz: uint32(d.clock_mhz),
		MemBytes: uint64(d.global_mem_bytes)}, nil
}

// CanAccessPeer is hipDeviceCanAccessPeer for two device indices of this process: whether MergeOutput between tracers on them
// is a direct xGMI peer read or a staged copy (MergeCounts says which one really ran).
func CanAccessPeer(device, peer int) bool {
	var can C.int
	return C.polaris_hip_can_access_peer(C.int(device), C.int(peer), &can) == C.POLARIS_OK && can != 0
}
