"""Model presets: the hyper-parameters of the reference configs on the hot path, as plain dicts.

Values (not code) taken from the reference config tree:
  configs/_base_/models/lora_dinov2_ms_masked.py:1-87   -> dinov2_ms_masked()
  configs/_base_/models/lora_dinov2_linear.py:1-53      -> dinov2_linear()
  configs/dg/gta2citys/dg_lora_dinov2_ms_masked.py:10-29 -> optim_cfg()
They use the reference's registry `type=` names and ctor kwargs so a user's own
configs/dg/*.py dicts are interchangeable with these.
"""
import copy

_PREPROC = dict(
    type="SegDataPreProcessor",
    mean=[123.675, 116.28, 103.53],
    std=[58.395, 57.12, 57.375],
    bgr_to_rgb=True,
    pad_val=0,
    seg_pad_val=255,
)

_CE = dict(type="CrossEntropyLoss", use_sigmoid=False, loss_weight=1.0)


def dinov2_backbone(depth=24, embed_dim=1024, num_heads=16, img_size=512):
    return dict(
        type="DinoVisionTransformer",
        patch_size=16,
        embed_dim=embed_dim,
        depth=depth,
        num_heads=num_heads,
        mlp_ratio=4,
        img_size=img_size,
        ffn_layer="mlp",
        init_values=1e-05,
        block_chunks=0,
        qkv_bias=True,
        proj_bias=True,
        ffn_bias=True,
    )


def lora_cfg(r=32, alpha=32, targets=("qkv",), dropout=0.1):
    return dict(r=r, lora_alpha=alpha, target_modules=list(targets), lora_dropout=dropout)


def linear_head(embed_dim=1024, channels=256, num_classes=19):
    return dict(
        type="LinearHead",
        in_channels=[embed_dim] * 4,
        in_index=[0, 1, 2, 3],
        channels=channels,
        dropout_ratio=0.1,
        num_classes=num_classes,
        norm_cfg=dict(type="GN", num_groups=32),
        align_corners=False,
        loss_decode=copy.deepcopy(_CE),
    )


def vfm_head(embed_dim=1024, channels=256, num_classes=19, depth=3, mask_ratio=0.2):
    return dict(
        type="VFMHead",
        transformer=dict(
            type="MaskTransformerDecoder",
            query_dim=channels,
            n_heads=8,
            d_head=64,
            depth=depth,
            dropout=0.1,
            mask_ratio=mask_ratio,
        ),
        in_channels=[embed_dim] * 4,
        in_index=[0, 1, 2, 3],
        channels=channels,
        dropout_ratio=0.1,
        num_classes=num_classes,
        norm_cfg=dict(type="GN", num_groups=32),
        align_corners=False,
        loss_decode=copy.deepcopy(_CE),
    )


def dinov2_ms_masked(depth=24, embed_dim=1024, num_heads=16, checkpoint=None, work_dir="work_dirs/tmp"):
    """The `MsVFMEncoderDecoder` model of BASELINE config 2/3."""
    return dict(
        type="MsVFMEncoderDecoder",
        data_preprocessor=dict(_PREPROC, size=(1024, 1024)),
        backbone=dict(
            type="LoRABackbone",
            backbone=dinov2_backbone(depth, embed_dim, num_heads),
            checkpoint=checkpoint,
            Lora_config=lora_cfg(),
        ),
        decode_head=linear_head(embed_dim),
        aux_head=vfm_head(embed_dim),
        detail_loss=1.0,
        scales=[1, 0.5],
        hr_crop_size=(512, 512),
        feature_scale=0.5,
        crop_coord_divisible=32,
        train_cfg=dict(work_dir=work_dir, log_config=dict(interval=50, img_interval=500)),
        test_cfg=dict(
            mode="ms_slide_inference",
            threadshod=0.968,
            conf=0.8,
            lr_img_size=(512, 1024),
            stride=[320, 320],
            crop_size=[512, 512],
        ),
    )


def dinov2_linear(depth=24, embed_dim=1024, num_heads=16, checkpoint=None):
    """Single-scale segmentor of BASELINE config 1 (DINOv2-L + LoRA + LinearHead, one 512^2 pass per sample): the
    `EncoderDecoder` form of configs/_base_/models/lora_*_linear.py with the DINOv2 backbone block of lora_dinov2_ms_masked.py."""
    return dict(
        type="EncoderDecoder",
        data_preprocessor=dict(_PREPROC, size=(512, 512)),
        backbone=dict(type="LoRABackbone", backbone=dinov2_backbone(depth, embed_dim, num_heads), checkpoint=checkpoint,
                      Lora_config=lora_cfg()),
        decode_head=linear_head(embed_dim),
        train_cfg=dict(),
        test_cfg=dict(mode="whole"),
    )


def optim_cfg():
    embed_multi = dict(lr_mult=1.0, decay_mult=0.0)
    return dict(
        optim_wrapper=dict(
            constructor="PEFTOptimWrapperConstructor",
            optimizer=dict(type="AdamW", lr=0.0001, weight_decay=0.05, eps=1e-8, betas=(0.9, 0.999)),
            paramwise_cfg=dict(
                custom_keys={
                    "norm": dict(decay_mult=0.0),
                    "query_embed": embed_multi,
                    "level_embed": embed_multi,
                    "learnable_tokens": embed_multi,
                    "reins.scale": embed_multi,
                },
                norm_decay_mult=0.0,
            ),
        ),
        param_scheduler=[dict(type="PolyLR", eta_min=0, power=0.9, begin=0, end=40000, by_epoch=False)],
    )


def eva02_backbone(depth=24, embed_dim=1024, num_heads=16, img_size=512):
    """configs/_base_/models/lora_eva02_ms_masked.py:25-49"""
    return dict(type="EVA2", depth=depth, drop_path_rate=0.1, embed_dim=embed_dim, img_size=img_size, in_chans=3,
                init_values=None, intp_freq=True, mlp_ratio=2.6666666666666665, naiveswiglu=True,
                norm_layer=dict(eps=1e-06, requires_grad=True, type="LN"), num_heads=num_heads,
                out_indices=[7, 11, 15, 23], patch_size=16, pt_hw_seq_len=16, qkv_bias=True, rope=True, subln=True,
                use_abs_pos_emb=True, use_checkpoint=False, use_rel_pos_bias=False, use_shared_rel_pos_bias=False, xattn=True)


def eva02_lora_cfg(dropout=0.1):
    """configs/_base_/models/lora_eva02_ms_masked.py:17-23"""
    return dict(r=32, lora_alpha=32, target_modules=["q_proj", "k_proj", "v_proj", "attn.proj"], lora_dropout=dropout)


def eva02_ms_masked(depth=24, checkpoint=None, work_dir="work_dirs/tmp"):
    """configs/_base_/models/lora_eva02_ms_masked.py (BASELINE config 4): EVA02-L + LoRA, LinearHead + VFMHead."""
    cfg = dinov2_ms_masked(work_dir=work_dir)
    cfg["backbone"] = dict(type="LoRABackbone", backbone=eva02_backbone(depth), checkpoint=checkpoint, Lora_config=eva02_lora_cfg())
    cfg["test_cfg"] = dict(mode="hr_slide_inference", stride=[320, 320], crop_size=[512, 512])  # fixed 512^2 grid
    return cfg


def sam_backbone(depth=32, embed_dim=1280, num_heads=16, img_size=512, global_idx=(7, 15, 23, 31), out_indices=(7, 15, 23, 31)):
    """configs/_base_/models/lora_sam_linear.py:16-27"""
    return dict(type="SAMViT", img_size=img_size, embed_dim=embed_dim, depth=depth, num_heads=num_heads,
                global_attn_indexes=list(global_idx), out_indices=list(out_indices), window_size=14, use_rel_pos=True)


def sam_linear(depth=32, checkpoint=None):
    """configs/_base_/models/lora_sam_linear.py (BASELINE config 5): EncoderDecoder, LoRA SAM-H, LinearHead, slide test."""
    return dict(
        type="EncoderDecoder",
        data_preprocessor=dict(_PREPROC, size=(512, 512)),
        backbone=dict(type="LoRABackbone", backbone=sam_backbone(depth), checkpoint=checkpoint, Lora_config=lora_cfg()),
        decode_head=dict(linear_head(1280, 320), in_channels=[1280] * 4),
        train_cfg=dict(),
        test_cfg=dict(mode="slide", stride=[320, 320], crop_size=[512, 512]),
    )


def clip_backbone(layers=24, width=1024, heads=16, input_resolution=512, out_indices=(7, 11, 15, 23)):
    """configs/_base_/models/lora_clip_ms_masked.py:16-29"""
    return dict(type="CLIPVisionTransformer", patch_size=16, width=width, output_dim=512, get_embeddings=False, drop_path_rate=0.1,
                layers=layers, input_resolution=input_resolution, style="pytorch", out_indices=list(out_indices), heads=heads)


def clip_lora_cfg(dropout=0.1):
    """configs/_base_/models/lora_clip_ms_masked.py:31-37"""
    return dict(r=32, lora_alpha=32, target_modules=["out_proj", "mlp.c_fc", "mlp.c_proj"], lora_dropout=dropout)


def clip_ms_masked(layers=24, checkpoint=None, work_dir="work_dirs/tmp"):
    """configs/_base_/models/lora_clip_ms_masked.py: CLIP ViT-L/16 + LoRA, LinearHead + VFMHead."""
    cfg = dinov2_ms_masked(work_dir=work_dir)
    cfg["backbone"] = dict(type="LoRABackbone", backbone=clip_backbone(layers), checkpoint=checkpoint, Lora_config=clip_lora_cfg())
    cfg["test_cfg"].pop("lr_img_size", None)   # lora_clip_ms_masked.py:78-84 has no lr_img_size (the key is read nowhere)
    return cfg


def sam_ms_masked(depth=32, checkpoint=None, work_dir="work_dirs/tmp", global_idx=(7, 15, 23, 31), out_indices=(7, 15, 23, 31)):
    """configs/_base_/models/lora_sam_ms_masked.py: SAM-ViT-H + LoRA(qkv), LinearHead(320) + VFMHead on 1280-wide taps."""
    cfg = dinov2_ms_masked(work_dir=work_dir)
    cfg["backbone"] = dict(type="LoRABackbone", backbone=sam_backbone(depth, global_idx=global_idx, out_indices=out_indices),
                           checkpoint=checkpoint, Lora_config=lora_cfg())
    cfg["decode_head"] = dict(linear_head(1280, 320), in_channels=[1280] * 4)
    cfg["aux_head"] = dict(vfm_head(1280), in_channels=[1280] * 4)
    cfg["test_cfg"] = dict(mode="hr_slide_inference", stride=[320, 320], crop_size=[512, 512])   # fixed 512^2 grid
    return cfg
