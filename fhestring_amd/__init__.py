"""fhestring_amd: MI355X-native TFHE programmable-bootstrap backend under the
FheString server-key operations of MakisChristou/fhestring.

Python here is only a thin binding over the C ABI (include/fhestring_hip.h);
the product is the HIP library built from fhestring_amd/csrc.
"""
from ._lib import FhsError, lib, LIB_PATH  # noqa: F401
from .api import Context  # noqa: F401
