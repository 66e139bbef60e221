package hip

/*
#include "polaris_hip.h"
*/
import "C"

import "strings"

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
