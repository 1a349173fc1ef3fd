"""dtype helpers of util/cython/tools.pyx:7-37."""
import numpy as np

float_type = np.float32          # sizeof(float) == 4 is asserted by the C ABI as well
int_type = np.int32


def is_ndarray(array, dtype):
    """True only for an owning ndarray of exactly `dtype` (tools.pyx:30-37)."""
    return isinstance(array, np.ndarray) and array.dtype == dtype and array.base is None
