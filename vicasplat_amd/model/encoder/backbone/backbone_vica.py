"""VicaNet backbone -- ViT-L/16 per-frame encoder + 12-block video/camera decoder -- MI355X-native.

Module tree and parameter names are the reference's (backbone_vica.py:340-448, croco/blocks.py:58-130,195-240;
state_dict keys: SURVEY.md Appendix C) so its checkpoints load with strict=True.  The nn.Modules only HOLD the
parameters: the forward pass drives the hand-written HIP operators (vicasplat_amd.ops) over flat token buffers:

  * residual streams live in f32 ([B*T, 257, C] image tokens, [B*T, C] camera tokens); every GEMM operand is
    f16 (default) or bf16 and accumulates in f32 -- the TF32-class precision the reference runs at
    (backbone_vica.py:9); LayerNorm / softmax statistics are f32;
  * q|k|v stay packed in the projection output, RoPE is applied in place, attention reads heads in place and
    writes token-major output: no reshape/transpose/cat/roll copies (backbone_vica.py:88-113,172-186);
  * the camera tokens ride in row 0 of each frame of an interleaved [B*T, 258, C] operand buffer so that ONE
    qkv GEMM / ONE attention launch serves image and camera queries (the blocked-causal mask of :585-593 is a
    per-query key-prefix length);
  * AdaLN scale/shift are fused into the LayerNorm kernel, gates and residual adds into the GEMM epilogue.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn

import os

from .... import ops

# A/B switch of the packed q | k | v route of the split class (VS_ATTN_PACKED=0: f32 q | k | v + attention_split_kernel, the round-3 path)
_ATTN_PACKED = os.environ.get("VS_ATTN_PACKED", "1") != "0"


class _Mlp(nn.Module):
    def __init__(self, dim: int, hidden: int):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class _Attention(nn.Module):
    def __init__(self, dim: int, num_heads: int):
        super().__init__()
        self.num_heads = num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)


class _CrossNeighborAttention(nn.Module):
    def __init__(self, dim: int, num_heads: int):
        super().__init__()
        self.num_heads = num_heads
        self.projq = nn.Linear(dim, dim, bias=True)
        self.projk = nn.Linear(dim, dim, bias=True)
        self.projv = nn.Linear(dim, dim, bias=True)
        self.proj = nn.Linear(dim, dim)


class _AdaLN(nn.Module):
    def __init__(self, dim: int, n_mods: int):
        super().__init__()
        self.n_mods = n_mods
        self.proj = nn.Linear(dim, n_mods * dim, bias=True)


class _EncBlock(nn.Module):
    def __init__(self, dim: int, heads: int, mlp_ratio: float):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Attention(dim, heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))


class _DecBlock(nn.Module):
    def __init__(self, dim: int, heads: int, mlp_ratio: float):
        super().__init__()
        self.cam_norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.modulation1 = _AdaLN(dim, 3)
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Attention(dim, heads)
        self.cam_norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.modulation2 = _AdaLN(dim, 6)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.cross_attn = _CrossNeighborAttention(dim, heads)
        self.norm3 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))
        self.mlp_cam = _Mlp(dim, int(dim * mlp_ratio))


class _PatchEmbed(nn.Module):
    def __init__(self, patch: int, dim: int):
        super().__init__()
        self.proj = nn.Conv2d(3, dim, kernel_size=patch, stride=patch)


class VicaNet(nn.Module):
    def __init__(self, img_size=224, patch_size=16, enc_embed_dim=1024, enc_depth=24, enc_num_heads=16, dec_embed_dim=768,
                 dec_depth=12, dec_num_heads=12, mlp_ratio=4.0, temporal_rope_theta=100, rope_dim_list=(16, 56, 56),
                 use_blocked_causal_attention=True, use_framewise_modulation=True, use_cross_neighbor_attention=True,
                 use_intrinsic_embedding=True, compute_dtype: torch.dtype = torch.float16):
        super().__init__()
        cfg = dict(img_size=img_size, patch_size=patch_size, enc_embed_dim=enc_embed_dim, enc_depth=enc_depth,
                   enc_num_heads=enc_num_heads, dec_embed_dim=dec_embed_dim, dec_depth=dec_depth, dec_num_heads=dec_num_heads,
                   mlp_ratio=mlp_ratio, temporal_rope_theta=temporal_rope_theta, rope_dim_list=list(rope_dim_list),
                   use_blocked_causal_attention=use_blocked_causal_attention, use_framewise_modulation=use_framewise_modulation,
                   use_cross_neighbor_attention=use_cross_neighbor_attention, use_intrinsic_embedding=use_intrinsic_embedding)
        self.config = SimpleNamespace(**cfg)
        if not (use_blocked_causal_attention and use_framewise_modulation and use_cross_neighbor_attention
                and len(cfg["rope_dim_list"]) == 2):
            raise NotImplementedError("only the configuration of the released experiments is implemented "
                                      "(config/model/encoder/backbone/vica.yaml, use_intrinsic_embedding true or false)")
        if enc_embed_dim % 64 or dec_embed_dim % 64 or enc_embed_dim // enc_num_heads != 64 or dec_embed_dim // dec_num_heads != 64:
            raise NotImplementedError("the HIP attention kernel is specialised for head_dim 64")
        # attribute fallbacks the heads rely on (dpt_head.py:105-110 reads net.dec_depth / enc_embed_dim / dec_embed_dim)
        self.enc_depth, self.dec_depth, self.enc_embed_dim, self.dec_embed_dim = enc_depth, dec_depth, enc_embed_dim, dec_embed_dim
        self.compute_dtype = compute_dtype
        self.split = False    # split operand class (ops.SplitWeight): f32 activations, weights packed as f16 (hi, lo) pairs -- set by VicaSplat.set_compute_dtype("split")
        self.patch_embed = _PatchEmbed(patch_size, enc_embed_dim)
        self.enc_blocks = nn.ModuleList([_EncBlock(enc_embed_dim, enc_num_heads, mlp_ratio) for _ in range(enc_depth)])
        self.enc_norm = nn.LayerNorm(enc_embed_dim, eps=1e-6)
        self.decoder_embed = nn.Linear(enc_embed_dim, dec_embed_dim, bias=True)
        self.dec_blocks = nn.ModuleList([_DecBlock(dec_embed_dim, dec_num_heads, mlp_ratio) for _ in range(dec_depth)])
        self.dec_norm = nn.LayerNorm(dec_embed_dim, eps=1e-6)
        self.camera_dec_norm = nn.LayerNorm(dec_embed_dim, eps=1e-6)
        self.intrinsic_encoder = nn.Linear(9, enc_embed_dim) if use_intrinsic_embedding else None   # backbone_vica.py:392-395
        self.camera_extrinsic_token = nn.Parameter(torch.empty(dec_embed_dim).normal_(std=0.02))
        self.camera_intrinsic_token = nn.Parameter(torch.empty(dec_embed_dim).normal_(std=0.02))
        self.gradient_checkpointing = False
        self.checkpoint_blocks = None      # (n_enc, n_dec) blocks that are checkpointed when the flag is on; None = all
        self._probe = None    # test hook: callable(name, tensor) on every block's output stream (enc%02d, dec%02d_img, dec%02d_cam)
        self._init_weights()
        self._w16: dict = {}
        self._w16_key = None
        self._tables: dict = {}

    def enable_gradient_checkpointing(self, blocks=None):
        """Per-block activation checkpointing of the TRAINING forward (train_forward.forward_train reads the flag); the
        inference forward below keeps nothing, so there is nothing to checkpoint (backbone_vica.py:464-474,504-516).
        blocks = (n_enc, n_dec): checkpoint only the FIRST n_enc encoder and n_dec decoder blocks (None = all, the reference's policy;
        "auto" = train_forward.auto_checkpoint_blocks: as few as the device memory allows for the batch at hand).  On a
        288 GB MI355X the configuration's 24 scenes per GPU need ~290 GB without and ~160 GB with full checkpointing: recomputing only as many
        blocks as the memory requires buys back most of the recomputation."""
        self.gradient_checkpointing = True
        self.checkpoint_blocks = None if blocks is None else ("auto" if blocks == "auto" else (int(blocks[0]), int(blocks[1])))

    def _init_weights(self):  # backbone_vica.py:427-448 (xavier on every Linear, incl. the AdaLN projections)
        w = self.patch_embed.proj.weight.data
        nn.init.xavier_uniform_(w.view(w.shape[0], -1))
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.LayerNorm):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    # ------------------------------------------------------------------------------------------------------
    # 16-bit operand copies of the GEMM weights (re-made when a parameter changes or moves)
    # ------------------------------------------------------------------------------------------------------
    def _weights16(self):
        key = (self.compute_dtype, self.split, self.patch_embed.proj.weight.device,
               sum(p._version for p in self.parameters()), id(self.patch_embed.proj.weight))
        if key == self._w16_key:
            return self._w16
        dt = self.compute_dtype
        c = (lambda t: ops.split_pack_weight(t)) if self.split else (lambda t: t.detach().to(dt).contiguous())
        W = {"patch": c(self.patch_embed.proj.weight.flatten(1)), "dec_embed": c(self.decoder_embed.weight)}
        for i, b in enumerate(self.enc_blocks):
            W[f"e{i}.qkv"], W[f"e{i}.proj"] = c(b.attn.qkv.weight), c(b.attn.proj.weight)
            W[f"e{i}.fc1"], W[f"e{i}.fc2"] = c(b.mlp.fc1.weight), c(b.mlp.fc2.weight)
        for i, b in enumerate(self.dec_blocks):
            W[f"d{i}.mod1"], W[f"d{i}.mod2"] = c(b.modulation1.proj.weight), c(b.modulation2.proj.weight)
            W[f"d{i}.qkv"], W[f"d{i}.proj"] = c(b.attn.qkv.weight), c(b.attn.proj.weight)
            ca = b.cross_attn
            W[f"d{i}.cqkv"] = c(torch.cat([ca.projq.weight, ca.projk.weight, ca.projv.weight], 0))
            W[f"d{i}.cqkv_b"] = torch.cat([ca.projq.bias, ca.projk.bias, ca.projv.bias], 0).detach().float().contiguous()
            W[f"d{i}.cproj"] = c(ca.proj.weight)
            W[f"d{i}.fc1"], W[f"d{i}.fc2"] = c(b.mlp.fc1.weight), c(b.mlp.fc2.weight)
            W[f"d{i}.cfc1"], W[f"d{i}.cfc2"] = c(b.mlp_cam.fc1.weight), c(b.mlp_cam.fc2.weight)
        self._w16, self._w16_key = W, key
        return W

    def _pos_tables(self, B: int, T: int, gh: int, gw: int, dev):
        key = (B, T, gh, gw, str(dev))
        if key in self._tables:
            return self._tables[key]
        use_intr = self.config.use_intrinsic_embedding
        ys, xs = torch.meshgrid(torch.arange(gh), torch.arange(gw), indexing="ij")
        frame = torch.stack([ys, xs], -1).reshape(gh * gw, 2)
        if use_intr:
            frame = torch.cat([frame, torch.tensor([[gh, 0]])], 0)                 # intrinsic token at (gh, 0) (backbone_vica.py:455-459)
        frame = frame.int()
        n1 = frame.shape[0]                                                        # tokens per frame: 257, or 256 without the intrinsic token
        pos_img = frame.repeat(B * T, 1).contiguous()                              # [B*T*n1, 2]
        # interleaved buffer: row 0 of each frame = camera token (temporal position t), rows 1.. = image tokens
        t_idx = torch.arange(T).repeat(B)
        cam = torch.stack([t_idx, torch.zeros_like(t_idx)], -1).int()[:, None]     # [B*T,1,2]
        pos_mix = torch.cat([cam, frame[None].expand(B * T, -1, -1)], 1).reshape(-1, 2).contiguous()
        kind = torch.zeros(B * T, n1 + 1, dtype=torch.uint8)
        kind[:, 0] = 1
        L = T * (n1 + 1)
        kvlen = torch.full((B, T, n1 + 1), L, dtype=torch.int32)
        kvlen[:, :, 0] = (torch.arange(T, dtype=torch.int32) + 1)[None] * (n1 + 1)  # camera query t sees frames <= t
        if not use_intr:
            kvlen[:, 0, 0] = L     # without the intrinsic embedding camera token 0 has global scope (backbone_vica.py:589-590)
        if T == 2:
            nb = [[1], [0]]
        else:
            nb = [[1, 1]] + [[t - 1, t + 1] for t in range(1, T - 1)] + [[T - 2, T - 2]]
        seg = []
        for b in range(B):
            for t in range(T):
                s = [((b * T + j) * n1, n1) for j in nb[t]]
                seg.append([s[0][0], s[0][1], s[1][0] if len(s) > 1 else 0, s[1][1] if len(s) > 1 else 0])
        tabs = dict(pos_img=pos_img.to(dev), pos_mix=pos_mix.to(dev), kind_mix=kind.reshape(-1).contiguous().to(dev),
                    kvlen=kvlen.reshape(-1).contiguous().to(dev), seg=torch.tensor(seg, dtype=torch.int32).to(dev))
        self._tables[key] = tabs
        return tabs

    # ------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, x: torch.Tensor, intrinsics: Optional[torch.Tensor] = None):
        """x [B,3,T,H,W] (normalised), intrinsics [B,T,3,3] -> (x_final, camera_extrinsic [B,T-1,C], None, intermediates)
        exactly as backbone_vica.py:526-582; the 13 intermediates are [B,T,N,C] with the intrinsic token dropped
        (only hooks 0, L/2, 3L/4, L -- the ones the DPT heads read -- are materialised; the others are None)."""
        if not x.is_cuda:
            raise RuntimeError("VicaNet.forward needs HIP device tensors: vicasplat_amd has no CPU fallback path")
        B, _, T, H, Wd = x.shape
        cfg, dt, dev = self.config, self.compute_dtype, x.device
        use_intr = cfg.use_intrinsic_embedding
        assert intrinsics is not None or not use_intr, "use_intrinsic_embedding=True needs intrinsics"
        p = cfg.patch_size
        gh, gw = H // p, Wd // p
        n = gh * gw
        N = n + (1 if use_intr else 0)
        BT = B * T
        Ce, Cd, He, Hd = cfg.enc_embed_dim, cfg.dec_embed_dim, cfg.enc_num_heads, cfg.dec_num_heads
        W = self._weights16()
        tabs = self._pos_tables(B, T, gh, gw, dev)
        f32 = dict(dtype=torch.float32, device=dev)
        f16 = dict(dtype=dt, device=dev)

        # ---- patch embedding as a GEMM over im2col rows (blocks.py:227-236) + intrinsic token (backbone_vica.py:535-536) ----
        frames = x.permute(0, 2, 1, 3, 4).reshape(BT, 3, gh, p, gw, p)
        cols = frames.permute(0, 2, 4, 1, 3, 5).reshape(BT * n, 3 * p * p).to(dt)
        xe = torch.empty(BT * N, Ce, **f32)
        ops.gemm(cols, W["patch"], self.patch_embed.proj.bias, xe, ops.EPI_STORE32, grp_in=n, grp_out=N, grp_off=0)
        if use_intr:
            xe.view(BT, N, Ce)[:, n] = ops.linear_f32(intrinsics.reshape(BT, 9), self.intrinsic_encoder.weight, self.intrinsic_encoder.bias)

        # ---- 24 encoder blocks (blocks.py:94-130) ----
        # split class: activations that only feed a GEMM (LayerNorm outputs, the MLP's hidden layer) are written by their producer in the
        # packed (hi, lo) form, so the consuming GEMM's main loop has no conversion to do (ops.split_act; +15-20 % on these GEMMs)
        act = (lambda r, c: ops.split_act(r, c, dev)) if self.split else (lambda r, c: torch.empty(r, c, **f16))
        # split class, round 4: q | k | v leave the projection's RoPE epilogue in the packed (hi, lo) form as well and the attention kernel
        # stages them by LDS-DMA without converting anything (ops.attention on int32 slices -> attention_sp_kernel)
        qkv_packed = self.split and _ATTN_PACKED
        qact = act if qkv_packed else (lambda r, c: torch.empty(r, c, **f16))
        cols3 = lambda t, C_: ((t.data if qkv_packed else t)[:, :C_], (t.data if qkv_packed else t)[:, C_:2 * C_], (t.data if qkv_packed else t)[:, 2 * C_:])
        h = act(BT * N, Ce)
        qkv = qact(BT * N, 3 * Ce)
        att = act(BT * N, Ce)
        hid = act(BT * N, int(Ce * cfg.mlp_ratio))
        for i, blk in enumerate(self.enc_blocks):
            ops.layernorm_mod(xe, blk.norm1.weight, blk.norm1.bias, h)
            ops.gemm_qkv_rope(h, W[f"e{i}.qkv"], blk.attn.qkv.bias, qkv, Ce, tabs["pos_img"], None, 100.0, 1.0)
            ops.attention(*cols3(qkv, Ce), att, nbatch=BT, H=He, Lq=N, Lk=N, q_batch_rows=N, k_batch_rows=N, split=self.split)
            ops.gemm(att, W[f"e{i}.proj"], blk.attn.proj.bias, xe, ops.EPI_RESID32)
            ops.layernorm_mod(xe, blk.norm2.weight, blk.norm2.bias, h)
            ops.gemm(h, W[f"e{i}.fc1"], blk.mlp.fc1.bias, hid, ops.EPI_GELU16)
            ops.gemm(hid, W[f"e{i}.fc2"], blk.mlp.fc2.bias, xe, ops.EPI_RESID32)
            if self._probe is not None:
                self._probe(f"enc{i:02d}", xe)
        enc16 = torch.empty(BT * N, Ce, **f16)
        ops.layernorm_mod(xe, self.enc_norm.weight, self.enc_norm.bias, enc16)
        del h, qkv, att, hid

        # ---- decoder (backbone_vica.py:482-524) ----
        inter: list = [None] * (cfg.dec_depth + 1)
        inter[0] = enc16.view(B, T, N, Ce)[:, :, :n]
        hooks = {cfg.dec_depth * 2 // 4, cfg.dec_depth * 3 // 4, cfg.dec_depth}
        xd = torch.empty(BT * N, Cd, **f32)
        ops.gemm(enc16, W["dec_embed"], self.decoder_embed.bias, xd, ops.EPI_STORE32)
        ti, te = self.camera_intrinsic_token.float(), self.camera_extrinsic_token.float()
        cam = torch.cat([ti.expand(B, 1, Cd), (ti + te).expand(B, T - 1, Cd)], 1).reshape(BT, Cd).contiguous()
        M2 = N + 1  # rows per frame in the interleaved buffer
        hmix = act(BT * M2, Cd)
        qkvm = qact(BT * M2, 3 * Cd)
        attm = act(BT * M2, Cd)
        h = act(BT * N, Cd)
        qkv = qact(BT * N, 3 * Cd)
        att = act(BT * N, Cd)
        hid = act(BT * N, int(Cd * cfg.mlp_ratio))
        cn = torch.empty(BT, Cd, **f32)
        cn16 = torch.empty(BT, Cd, **f16)
        chid = torch.empty(BT, int(Cd * cfg.mlp_ratio), **f16)
        mod1 = torch.empty(BT, 3 * Cd, **f32)
        mod2 = torch.empty(BT, 6 * Cd, **f32)
        theta = float(cfg.temporal_rope_theta)
        for i, blk in enumerate(self.dec_blocks):
            # -- AdaLN parameters from the frame's camera token (:289-293, :194-212)
            ops.layernorm_mod(cam, blk.cam_norm1.weight, blk.cam_norm1.bias, cn)
            ops.gemm(ops.silu_cast(cn, dt), W[f"d{i}.mod1"], blk.modulation1.proj.bias, mod1, ops.EPI_STORE32)
            # -- video/camera self-attention over the interleaved [cam_t, img_t] sequence (:76-126)
            ops.layernorm_mod(xd, blk.norm1.weight, blk.norm1.bias, hmix, scale=mod1[:, :Cd], shift=mod1[:, Cd:2 * Cd],
                              mod_rows=N, grp_in=N, grp_out=M2, grp_off=1)
            if self.split:
                hmix.data.view(BT, M2, Cd)[:, 0] = ops.split_pack_weight(cn, 0).data
            else:
                hmix.view(BT, M2, Cd)[:, 0] = cn.to(dt)
            ops.gemm_qkv_rope(hmix, W[f"d{i}.qkv"], blk.attn.qkv.bias, qkvm, Cd, tabs["pos_mix"], tabs["kind_mix"], 100.0, theta)
            ops.attention(*cols3(qkvm, Cd), attm, nbatch=B, H=Hd, Lq=T * M2, Lk=T * M2,
                          q_batch_rows=T * M2, k_batch_rows=T * M2, q_kvlen=tabs["kvlen"], split=self.split)
            ops.gemm(attm, W[f"d{i}.proj"], blk.attn.proj.bias, xd, ops.EPI_RESID32, gate=mod1[:, 2 * Cd:], gate_rows=N,
                     M=BT * N, a_grp_in=N, a_grp_out=M2, a_grp_off=1)
            ops.gemm(attm, W[f"d{i}.proj"], blk.attn.proj.bias, cam, ops.EPI_RESID32, M=BT, a_grp_in=1, a_grp_out=M2, a_grp_off=0)
            # -- second modulation set (:306-318)
            ops.layernorm_mod(cam, blk.cam_norm2.weight, blk.cam_norm2.bias, cn)
            cn16.copy_(cn)
            ops.gemm(ops.silu_cast(cn, dt), W[f"d{i}.mod2"], blk.modulation2.proj.bias, mod2, ops.EPI_STORE32)
            # -- cross-neighbour attention (:152-191): keys/values of frames t-1, t+1 gathered by row segments
            ops.layernorm_mod(xd, blk.norm2.weight, blk.norm2.bias, h, scale=mod2[:, :Cd], shift=mod2[:, Cd:2 * Cd], mod_rows=N)
            ops.gemm_qkv_rope(h, W[f"d{i}.cqkv"], W[f"d{i}.cqkv_b"], qkv, Cd, tabs["pos_img"], None, 100.0, 1.0)
            ops.attention(*cols3(qkv, Cd), att, nbatch=BT, H=Hd, Lq=N, q_batch_rows=N, kv_seg=tabs["seg"], split=self.split)
            ops.gemm(att, W[f"d{i}.cproj"], blk.cross_attn.proj.bias, xd, ops.EPI_RESID32, gate=mod2[:, 2 * Cd:3 * Cd], gate_rows=N)
            # -- MLPs (:323-333); the camera MLP reads cam_norm2(cam), not a fresh norm
            ops.layernorm_mod(xd, blk.norm3.weight, blk.norm3.bias, h, scale=mod2[:, 3 * Cd:4 * Cd], shift=mod2[:, 4 * Cd:5 * Cd], mod_rows=N)
            ops.gemm(h, W[f"d{i}.fc1"], blk.mlp.fc1.bias, hid, ops.EPI_GELU16)
            ops.gemm(hid, W[f"d{i}.fc2"], blk.mlp.fc2.bias, xd, ops.EPI_RESID32, gate=mod2[:, 5 * Cd:], gate_rows=N)
            ops.gemm(cn16, W[f"d{i}.cfc1"], blk.mlp_cam.fc1.bias, chid, ops.EPI_GELU16)
            ops.gemm(chid, W[f"d{i}.cfc2"], blk.mlp_cam.fc2.bias, cam, ops.EPI_RESID32)
            if self._probe is not None:
                self._probe(f"dec{i:02d}_img", xd)
                self._probe(f"dec{i:02d}_cam", cam)
            if (i + 1) in hooks and (i + 1) != cfg.dec_depth:
                inter[i + 1] = xd.view(B, T, N, Cd)[:, :, :n].to(dt, copy=True)   # (copy: xd keeps changing, and dt may be f32)
        last = torch.empty(BT * N, Cd, **f16)
        ops.layernorm_mod(xd, self.dec_norm.weight, self.dec_norm.bias, last)
        inter[cfg.dec_depth] = last.view(B, T, N, Cd)[:, :, :n]
        camn = torch.empty(BT, Cd, **f32)
        ops.layernorm_mod(cam, self.camera_dec_norm.weight, self.camera_dec_norm.bias, camn)
        camera = camn.view(B, T, Cd)
        return inter[-1], camera[:, 1:], (None if use_intr else camera[:, 0]), inter
