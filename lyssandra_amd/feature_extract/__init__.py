from .preproc import preproc, l2_normalizer, preproc_device  # noqa: F401
from .pooling import spatial_pyramid_pool  # noqa: F401
