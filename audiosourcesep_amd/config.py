"""Static description of one Glow flow (shapes only, no tensors).

Mirrors the arguments the reference passes to ``build_glow`` (flow_models/flow_builder.py:60-61)
plus the preprocessing kwargs consumed by ``SpecPreprocessing`` (flow_tfp_bijectors.py:365-370).
"""
from dataclasses import dataclass, asdict


@dataclass(frozen=True)
class GlowConfig:
    H: int = 64
    W: int = 64
    C: int = 1
    L: int = 3
    K: int = 32
    F: int = 512            # n_filters
    learntop: bool = True
    minval: float = -100.0  # melspec dB range, train_glow.py:275-277
    maxval: float = 20.0
    use_logit: bool = False
    alpha: float = 1e-10
    bn_eps: float = 1e-3    # Keras BatchNormalization default epsilon

    def __post_init__(self):
        if self.L not in (2, 3, 4):
            raise ValueError("L should be 2, 3 or 4")  # flow_builder.py:76-77
        s = 2 ** self.L
        if self.H % s or self.W % s:
            raise ValueError("H and W must be divisible by 2**L")

    def as_dict(self):
        return asdict(self)

    def level_shapes(self):
        """[(h, w, c)] seen by the steps of each block (flow_glow.py:63-77)."""
        out, h, w, c = [], self.H, self.W, self.C
        for _ in range(self.L):
            h, w, c = h // 2, w // 2, c * 4
            out.append((h, w, c))
            c //= 2
        return out

    def latent_shape(self):
        s = 2 ** self.L
        return (self.H // s, self.W // s, self.C * s * s)  # flow_builder.py:64-75

    def flop_per_tile(self):
        """Algorithmic FLOP of one forward+log-det pass of one tile (2 FLOP/MAC; conv1+conv2+conv3+1x1),
        the figure SURVEY section 8(d) quotes (25.72 G for 64x64 L3 K32 F512)."""
        tot = 0
        for (h, w, c) in self.level_shapes():
            ci = c // 2
            per_px = 2 * (9 * ci * self.F + self.F * self.F + 9 * self.F * c + c * c)
            tot += self.K * h * w * per_px
        return tot

    def act_bytes_per_tile(self):
        """Read+write of each step's [h,w,c] fp32 tensor (hiddens on chip), SURVEY section 8(d)."""
        return sum(self.K * h * w * c * 4 * 2 for (h, w, c) in self.level_shapes())


CONFIG_A = GlowConfig(H=32, W=32, C=1, L=2, K=16)     # BASELINE.json configs[1]
CONFIG_B = GlowConfig(H=64, W=64, C=1, L=3, K=32)     # BASELINE.json configs[2] (the metric's config)
CONFIG_YAML = GlowConfig(H=96, W=64, C=1, L=3, K=40)  # configs/melspec_glow.yml:4-15
