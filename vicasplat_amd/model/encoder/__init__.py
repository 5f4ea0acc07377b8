"""get_encoder -- reference: src/model/encoder/__init__.py:14-19."""
from .encoder import Encoder
from .vicasplat import OpacityMappingCfg, VicaSplat, VicaSplatCfg
from .common.gaussian_adapter import GaussianAdapterCfg

ENCODERS = {"vicasplat": (VicaSplat, None)}
EncoderCfg = VicaSplatCfg


def get_encoder(cfg: EncoderCfg):
    encoder_cls, _visualizer = ENCODERS[cfg.name]
    return encoder_cls(cfg), None


def default_cfg(**backbone_overrides) -> VicaSplatCfg:
    """config/model/encoder/vicasplat.yaml + backbone/vica.yaml + the experiment overrides (re10k_8view.yaml:31-33)."""
    bb = dict(img_size=256, patch_size=16, enc_embed_dim=1024, enc_depth=24, enc_num_heads=16, dec_embed_dim=768,
              dec_depth=12, dec_num_heads=12, mlp_ratio=4.0, temporal_rope_theta=30, rope_dim_list=[32, 32],
              use_blocked_causal_attention=True, use_framewise_modulation=True, use_cross_neighbor_attention=True,
              use_intrinsic_embedding=True)
    bb.update(backbone_overrides)
    return VicaSplatCfg(name="vicasplat", backbone=bb, visualizer=None,
                        gaussian_adapter=GaussianAdapterCfg(0.005, 0.04, 4, "softplus"), apply_bounds_shim=True,
                        opacity_mapping=OpacityMappingCfg(0.0, 0.0, 1), predict_opacity=False)
