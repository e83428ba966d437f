"""cellvit_amd — MI355X-native CellViT inference hot path (encoder + decoder + HoVer post-proc)."""
from .spec import (ARCH_SAM, ARCH_VIT, CellViTConfig, cellvit256_config, cellvit_sam_config,  # noqa
                   param_specs)
