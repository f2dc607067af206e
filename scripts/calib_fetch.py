"""Dev/measurement helper: run the known-byte-count stream copy (8 B per lane) a few times so that rocprofv3
--pmc FETCH_SIZE / WRITE_SIZE can be calibrated for the stage kernels' access shape."""
import ctypes as C, sys
sys.path.insert(0, ".")
import numpy as np
import mara3_amd
from mara3_amd.engine import DeviceArray
from mara3_amd import _lib as L
lib = mara3_amd.load_library()
n = 5 * 4096 * 4096          # one 671 MB field
a = DeviceArray(np.zeros(n))
b = DeviceArray.empty((n,))
for _ in range(5):
    L.check(lib.mh_calib_stream_copy(a.ptr, b.ptr, n, None))
L.check(lib.mh_device_synchronize())
print("copied", n * 8, "bytes x5")
