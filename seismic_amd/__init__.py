"""seismic_amd — MI355X-native search hot path of Seismic behind the reference's Python API.

    from seismic_amd import SeismicIndex, SeismicIndexLV, SeismicIndexRaw, SeismicIndexRawLV
    from seismic_amd import SeismicDataset, SeismicDatasetLV, get_seismic_string

The compute path is libseismic_hip.so (hand-written HIP for gfx950 behind the C ABI in
include/seismic_hip.h). Importing this package does not load it; the first index operation does,
and fails loudly if the library or a HIP device is missing (no CPU fallback).
"""
from .index import (SeismicDataset, SeismicDatasetLV, SeismicIndex, SeismicIndexDotVByte, SeismicIndexLV, SeismicIndexRaw,
                    SeismicIndexRawLV, get_seismic_string, read_inner_format, write_inner_format)

__all__ = ["SeismicIndex", "SeismicIndexLV", "SeismicIndexDotVByte", "SeismicIndexRaw", "SeismicIndexRawLV", "SeismicDataset",
           "SeismicDatasetLV", "get_seismic_string", "read_inner_format", "write_inner_format"]
__version__ = "0.1.0"
