"""Decoder interface (reference: src/model/decoder/decoder.py:17-45)."""
from __future__ import annotations

from abc import ABC, abstractmethod
from dataclasses import dataclass
from typing import Generic, Literal, Optional, TypeVar

from torch import Tensor, nn

from ..types import Gaussians

DepthRenderingMode = Literal["depth", "log", "disparity", "relative_disparity"]


@dataclass
class DecoderOutput:
    color: Optional[Tensor] = None  # [batch, view, 3, height, width]
    depth: Optional[Tensor] = None  # [batch, view, height, width]


T = TypeVar("T")


class Decoder(nn.Module, ABC, Generic[T]):
    cfg: T

    def __init__(self, cfg: T) -> None:
        super().__init__()
        self.cfg = cfg

    @abstractmethod
    def forward(self, gaussians: Gaussians, extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor,
                image_shape: tuple[int, int], depth_mode: DepthRenderingMode | None = None) -> DecoderOutput:
        ...
