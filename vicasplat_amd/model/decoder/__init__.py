"""get_decoder -- reference: src/model/decoder/__init__.py:11-12."""
from .decoder import Decoder, DecoderOutput
from .decoder_splatting_cuda import DecoderSplattingCUDA, DecoderSplattingCUDACfg

DECODERS = {"splatting_cuda": DecoderSplattingCUDA}
DecoderCfg = DecoderSplattingCUDACfg


def get_decoder(decoder_cfg: DecoderCfg) -> Decoder:
    return DECODERS[decoder_cfg.name](decoder_cfg)
