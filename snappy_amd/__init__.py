"""MI355X-native asset hashing / MCDF decode path of the Snappy plugin.

Thin Python access to libsnappyhash.so (see include/snappyhash.h).  All compute
runs in hand-written gfx950 HIP kernels behind the C ABI.
"""
from . import native  # noqa: F401
from .crypto import get_hash, sha1, sha1_batch, sha1_batch_device, sha1_files  # noqa: F401
from .mcdf import (  # noqa: F401
    lz4net_decode,
    lz4net_decoded_size,
    lz4net_encode,
    mcdf_decode_and_hash,
    mcdf_decode_and_hash_batch,
    mcdf_parse_header,
    mcdf_write_header,
    write_mcdf,
)
from .snapshot import import_mcdf, store_files, update_snapshot  # noqa: F401,E402
