"""Calibrate rocprofv3's FETCH_SIZE on gfx950 for the EVM kernel's access pattern.

Run under `rocprofv3 --pmc FETCH_SIZE`: launches the library's calib_gather_kernel over buffers of
known size (each byte read exactly once; lane records of 416 B = one 13-cell step row, and 448 B =
one 14-cell rw row) and prints the byte counts, so counter/bytes gives the correction factor that
bench.py's roofline.traffic applies (profiles/r01_fetch_size_calibration.txt).
"""
import ctypes
import sys

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from zkevm_specs_amd import _lib

lib = _lib.load()
_lib.init(0)
ms = ctypes.c_float()
for nbytes, lane in ((1 << 30, 416), (1 << 30, 448), (1 << 30, 16)):
    rc = lib.zk_debug_calib_gather(ctypes.c_uint64(nbytes), ctypes.c_uint32(lane), ctypes.byref(ms))
    assert rc == 0
    n = nbytes // lane * lane
    print(f"calib lane_bytes={lane} bytes={n} ms={ms.value:.3f} GB/s={n / ms.value / 1e6:.1f}")
