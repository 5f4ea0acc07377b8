"""Encoder interface (reference: src/model/encoder/encoder.py:10-24)."""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Generic, TypeVar

from torch import nn

T = TypeVar("T")


class Encoder(nn.Module, ABC, Generic[T]):
    cfg: T

    def __init__(self, cfg: T) -> None:
        super().__init__()
        self.cfg = cfg

    @abstractmethod
    def forward(self, context: dict, **kwargs) -> dict:
        ...

    def get_data_shim(self):
        return lambda x: x
