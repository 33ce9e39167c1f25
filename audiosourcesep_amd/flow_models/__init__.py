"""Mirror of the reference's ``flow_models`` package for the Glow path (build_glow and what it returns)."""
from .flow_builder import build_glow  # noqa: F401
from .flow_glow import GlowFlow  # noqa: F401
