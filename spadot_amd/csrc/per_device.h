// A flag per HIP device of the process (index = hipGetDevice()): "has this been done on the device that is current now?"
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-device property of a kernel; one process normally drives one
// GPU, but every entry point takes its device from the caller's current device / stream, so a process-wide
// `static bool` would leave a second device without the attribute and its launch with more than 64 KiB of dynamic LDS
// would fail.  Reads as a bool:  static PerDeviceFlag done;  if (!done) { ...; done = true; }
#pragma once
#include <hip/hip_runtime.h>

struct PerDeviceFlag {
    unsigned char f[64] = {};
    unsigned char &cur() {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) d = 0;
        return f[d];
    }
    bool operator!() { return !cur(); }
    PerDeviceFlag &operator=(bool v) {
        cur() = v ? 1 : 0;
        return *this;
    }
};
