"""audiosourcesep_amd -- MI355X-native Glow forward / inverse / log-prob engine for mel-spectrogram tiles.

Drop-in for the ``flow_models.flow_builder.build_glow`` path of SamArgt/AudioSourceSep: Python host code
(same call surface) -> C ABI (``include/glowk.h``) -> hand-written gfx950 HIP kernels.
"""
from .config import GlowConfig, CONFIG_A, CONFIG_B, CONFIG_YAML  # noqa: F401

__all__ = ["GlowConfig", "CONFIG_A", "CONFIG_B", "CONFIG_YAML"]
