"""Scene container handed from the encoder to the decoder (reference: src/model/types.py:7-12)."""
from dataclasses import dataclass

from torch import Tensor


@dataclass
class Gaussians:
    means: Tensor        # [batch, gaussian, 3]
    covariances: Tensor  # [batch, gaussian, 3, 3]
    harmonics: Tensor    # [batch, gaussian, 3, d_sh]
    opacities: Tensor    # [batch, gaussian]
