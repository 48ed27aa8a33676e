"""MI355X-native RAW image pipeline: drop-in for raw_image_pipeline::RawImagePipeline's per-frame
chain (debayer -> flip -> white balance -> colour calibration -> gamma -> vignetting -> colour
enhancer -> undistortion) as hand-written gfx950 HIP kernels behind the C-ABI of include/rip.h."""
from .pipeline import (IMAGE_COLOR, IMAGE_DEBAYERED, IMAGE_PROCESSED, IMAGE_RECT_MASK, LIB_PATH, TAP_COLOR,
                       TAP_DEBAYERED, TAP_PROCESSED, OutputPool, RawImagePipeline, RipAssertError, RipError, RipIOError,
                       load_library)

__all__ = ["RawImagePipeline", "OutputPool", "RipError", "RipAssertError", "RipIOError", "load_library", "LIB_PATH",
           "TAP_DEBAYERED", "TAP_COLOR", "TAP_PROCESSED", "IMAGE_DEBAYERED", "IMAGE_COLOR", "IMAGE_PROCESSED",
           "IMAGE_RECT_MASK"]
