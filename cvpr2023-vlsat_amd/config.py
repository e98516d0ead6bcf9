"""Hyper-parameters and weight inventory of the VL-SAT eval forward path.

Key names follow the reference's ``config/mmgnet.json`` MODEL block (reference
``config/mmgnet.json:26-58``) and the ``state_dict`` key layout of ``Mmgnet``
(reference ``src/model/SGFN_MMG/model.py:20-159``); only the keys the eval
forward consumes are kept.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass, field


@dataclass
class VLSATConfig:
    # reference config/mmgnet.json:27,53-57 (shipped: N_LAYERS=2; BASELINE cfg2..5 use 3)
    N_LAYERS: int = 2
    NUM_HEADS: int = 8
    DIM_ATTEN: int = 256
    GCN_AGGR: str = "max"          # "max" | "add" | "mean"
    # config switches that change the path's ops (shipped values are the defaults)
    USE_GCN_EDGE: bool = True      # gate MLP on cat[q, k] (128->128->32) vs on q alone (64->128->32)  network_MMG.py:72-75,99-102
    WITH_BN: bool = False          # BatchNorm1d (eval affine) after fc1/fc2 of the relation heads        network_PointNet.py:320-337
    multi_rel_outputs: bool = True  # relation head ends in sigmoid (multi-label) vs log_softmax       SGFN_MMG/model.py:113-130
    USE_RGB: bool = False          # +3 point channels                                                   SGFN_MMG/model.py:31-35
    USE_NORMAL: bool = False       # +3 point channels
    feature_transform: bool = False  # STNkd 64x64 feature transform after conv1 of all three encoders  network_PointNet.py:52-86,146-150
    # forward(istrain=True) extras need triplet_projector_2d (SGFN_MMG/model.py:95-100,319-322); dead at eval, so its
    # weights are only part of the inventory when this is set
    train_outputs: bool = False
    clip_feat_dim: int = 512
    # fixed by Mmgnet.__init__ (reference SGFN_MMG/model.py:41-86)
    dim_node: int = 512
    dim_edge: int = 512
    dim_point: int = 3
    dim_descriptor: int = 11
    point_feature: int = 768
    num_obj_class: int = 160
    num_rel_class: int = 26
    # obj_logit_scale is never checkpointed: eval always sees log(1/0.07)
    obj_logit_scale: float = field(default_factory=lambda: math.log(1.0 / 0.07))

    def __post_init__(self):
        self.dim_point = 3 + (3 if self.USE_RGB else 0) + (3 if self.USE_NORMAL else 0)
        if self.GCN_AGGR not in ("max", "add", "mean"):
            raise ValueError(f"GCN_AGGR must be max/add/mean, got {self.GCN_AGGR}")
        if self.dim_node % self.NUM_HEADS or self.DIM_ATTEN % self.NUM_HEADS:
            raise ValueError("dim_node and DIM_ATTEN must be divisible by NUM_HEADS")


def param_shapes(cfg: VLSATConfig) -> "OrderedDict[str, tuple]":
    """Every tensor the eval forward reads, keyed exactly like the reference state_dict
    (prefix = top-level sub-module name, which is also the per-module checkpoint file
    name in reference ``model_base.py:65-71``)."""
    D, A, H = cfg.dim_node, cfg.DIM_ATTEN, cfg.NUM_HEADS
    dn, de, do = D // H, cfg.dim_edge // H, A // H
    s: "OrderedDict[str, tuple]" = OrderedDict()

    def lin(name, o, i):
        s[name + ".weight"] = (o, i)
        s[name + ".bias"] = (o,)

    def conv(name, o, i):
        s[name + ".weight"] = (o, i, 1)
        s[name + ".bias"] = (o,)

    def ln(name, d):
        s[name + ".weight"] = (d,)
        s[name + ".bias"] = (d,)

    def stn(prefix):        # STNkd(k=64): its BatchNorm layers ARE applied (eval affine), unlike PointNetfeat's own
        conv(prefix + ".conv1", 64, 64)
        conv(prefix + ".conv2", 128, 64)
        conv(prefix + ".conv3", 1024, 128)
        lin(prefix + ".fc1", 512, 1024)
        lin(prefix + ".fc2", 256, 512)
        lin(prefix + ".fc3", 64 * 64, 256)
        for bn, d in (("bn1", 64), ("bn2", 128), ("bn3", 1024), ("bn4", 512), ("bn5", 256)):
            for k in ("weight", "bias", "running_mean", "running_var"):
                s[f"{prefix}.{bn}.{k}"] = (d,)

    conv("obj_encoder.conv1", 64, cfg.dim_point)
    conv("obj_encoder.conv2", 128, 64)
    conv("obj_encoder.conv3", cfg.point_feature, 128)
    if cfg.feature_transform:
        stn("obj_encoder.fstn")
    for b in ("rel_encoder_2d", "rel_encoder_3d"):
        conv(b + ".conv1", 64, cfg.dim_descriptor)
        conv(b + ".conv2", 128, 64)
        conv(b + ".conv3", cfg.dim_edge, 128)
        if cfg.feature_transform:
            stn(b + ".fstn")
    lin("mlp_3d.0", D - 8, cfg.point_feature)
    for k in ("weight", "bias", "running_mean", "running_var"):
        s["mlp_3d.1." + k] = (D - 8,)
    lin("clip_adapter.fc1", 256, cfg.clip_feat_dim)
    lin("clip_adapter.fc2", cfg.clip_feat_dim, 256)
    for l in range(cfg.N_LAYERS):
        for a in ("self_attn", "cross_attn", "cross_attn_rel"):
            p = f"mmg.{a}.{l}"
            for f in ("fc_q", "fc_k", "fc_v", "fc_o"):
                lin(f"{p}.attention.{f}", D, D)
            ln(f"{p}.layer_norm", D)
        for g in ("gcn_2ds", "gcn_3ds"):
            p = f"mmg.{g}.{l}"
            lin(f"{p}.edgeatten.nn_edge.0", D + cfg.dim_edge, 2 * D + cfg.dim_edge)
            lin(f"{p}.edgeatten.nn_edge.2", cfg.dim_edge, D + cfg.dim_edge)
            if cfg.USE_GCN_EDGE:
                conv(f"{p}.edgeatten.nn.0", dn + de, dn + de)
                conv(f"{p}.edgeatten.nn.3", do, dn + de)
            else:
                conv(f"{p}.edgeatten.nn.0", 2 * dn, dn)
                conv(f"{p}.edgeatten.nn.3", do, 2 * dn)
            lin(f"{p}.edgeatten.proj_edge.0", cfg.dim_edge, cfg.dim_edge)
            lin(f"{p}.edgeatten.proj_query.0", D, D)
            lin(f"{p}.edgeatten.proj_value.0", A, D)
            lin(f"{p}.prop.0", D + A, D + A)
            lin(f"{p}.prop.2", D, D + A)
    lin("mmg.self_attn_fc.0", 32, 4)
    ln("mmg.self_attn_fc.2", 32)
    lin("mmg.self_attn_fc.3", 32, 32)
    ln("mmg.self_attn_fc.5", 32)
    lin("mmg.self_attn_fc.6", H, 32)
    for b in ("rel_predictor_3d", "rel_predictor_2d"):
        lin(b + ".fc1", 512, cfg.dim_edge)
        lin(b + ".fc2", 256, 512)
        lin(b + ".fc3", cfg.num_rel_class, 256)
        if cfg.WITH_BN:
            for bn, d in (("bn1", 512), ("bn2", 256)):
                for k in ("weight", "bias", "running_mean", "running_var"):
                    s[f"{b}.{bn}.{k}"] = (d,)
    lin("obj_predictor_3d", cfg.num_obj_class, cfg.clip_feat_dim)
    lin("obj_predictor_2d", cfg.num_obj_class, cfg.clip_feat_dim)
    if cfg.train_outputs:            # Sequential(Linear(1536,1024), Dropout, ReLU, Linear(1024,512))
        lin("triplet_projector_2d.0", 2 * D, 2 * D + cfg.dim_edge)
        lin("triplet_projector_2d.3", D, 2 * D)
    return s
