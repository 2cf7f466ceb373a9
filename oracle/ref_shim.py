"""TEST INFRASTRUCTURE ONLY - never imported by the product package.

Names-only shim that lets a handful of the reference's model files be imported *by
path* from /root/reference in THIS container (the reference never travels to the GPU
box).  It is used by oracle/gen_golden.py to produce tests/golden/*.npz, which pin
oracle/torch_ref.py (the CPU restatement).

What is real reference code when loaded through this shim (runs unmodified):
  rein/models/backbones/dino_v2.py + dino_layers/*, rein/models/heads/Transformer.py,
  the *bodies* of heads/linear_head.py, heads/VFMHead.py,
  segmentors/Ms_VFM_encoder_decoder.py, backbones/lora_backbone.py, backbones/utils.py
What is restated here because the third-party package is absent offline ("pinned by
restatement"): mmseg BaseDecodeHead / EncoderDecoder /
CrossEntropyLoss / accuracy / resize, mmcv ConvModule, mmengine BaseModule /
registry, peft LoRA Linear (peft 0.10.0 semantics).
"""
import importlib
import os
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

REF_ROOT = os.environ.get("VFMSEG_REFERENCE", "/root/reference")
PKG = "refrein"


# ----------------------------------------------------------------------------- registry
class _Registry:
    def __init__(self):
        self.table = {}

    def register_module(self, name=None, module=None, force=False):
        def deco(cls):
            self.table[name or cls.__name__] = cls
            return cls

        if module is not None:
            return deco(module)
        return deco

    def build(self, cfg, **default):
        cfg = dict(cfg)
        typ = cfg.pop("type")
        cls = self.table[typ] if isinstance(typ, str) else typ
        return cls(**cfg)


MODELS = _Registry()


class BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = init_cfg


class _Logger:
    @classmethod
    def get_current_instance(cls):
        return cls()

    def info(self, *a, **k):
        pass

    warning = info


# ----------------------------------------------------------------------------- mmseg bits
def resize(input, size=None, scale_factor=None, mode="nearest", align_corners=None, warning=True):
    return F.interpolate(input, size, scale_factor, mode, align_corners)


def add_prefix(inputs, prefix):
    return {f"{prefix}.{k}": v for k, v in inputs.items()}


def accuracy(pred, target, topk=1, thresh=None, ignore_index=None):
    # mmseg 1.2.2 losses/accuracy.py semantics for topk=1
    assert topk == 1
    pred_label = pred.argmax(dim=1)
    correct = pred_label.eq(target)
    if ignore_index is not None:
        valid = target != ignore_index
        correct = correct[valid]
        total = valid.sum()
    else:
        total = target.numel()
    eps = torch.finfo(torch.float32).eps
    return (correct.float().sum(0, keepdim=True) * (100.0 / (total + eps))).reshape(1)


@MODELS.register_module()
class CrossEntropyLoss(nn.Module):
    # mmseg 1.2.2: use_sigmoid=False, reduction='mean', avg_non_ignore=False
    def __init__(self, use_sigmoid=False, loss_weight=1.0, loss_name="loss_ce", **kw):
        super().__init__()
        assert not use_sigmoid
        self.loss_weight = loss_weight
        self._loss_name = loss_name

    @property
    def loss_name(self):
        return self._loss_name

    def forward(self, cls_score, label, weight=None, ignore_index=-100, **kw):
        loss = F.cross_entropy(cls_score, label, reduction="none", ignore_index=ignore_index)
        if weight is not None:
            loss = loss * weight.float()
        return self.loss_weight * loss.mean()


class ConvModule(nn.Module):
    # mmcv 2.1.0: conv(bias=False when norm) -> norm('gn') -> ReLU('activate')
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, norm_cfg=None,
                 act_cfg=dict(type="ReLU"), **kw):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, bias=norm_cfg is None)
        self.with_norm = norm_cfg is not None
        if self.with_norm:
            assert norm_cfg["type"] == "GN"
            self.gn = nn.GroupNorm(norm_cfg["num_groups"], out_channels)
        self.with_act = act_cfg is not None
        if self.with_act:
            self.activate = nn.ReLU(inplace=True)

    def forward(self, x):
        x = self.conv(x)
        if self.with_norm:
            x = self.gn(x)
        if self.with_act:
            x = self.activate(x)
        return x


def build_norm_layer(cfg, num_features, postfix=""):
    t = cfg["type"]
    if t == "LN":
        return "ln", nn.LayerNorm(num_features, eps=cfg.get("eps", 1e-5))
    raise NotImplementedError(t)


def _xformers_mea(q, k, v, attn_bias=None, p=0.0, scale=None):
    """xformers.ops.memory_efficient_attention on (B, N, H, D) tensors == exact softmax attention, scale D^-0.5."""
    assert attn_bias is None and p == 0.0
    o = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), scale=scale)
    return o.transpose(1, 2)


def _to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


def _drop_path(x, drop_prob=0.0, training=False):
    if drop_prob == 0.0 or not training:
        return x
    keep = 1 - drop_prob
    mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
    return x * mask / keep


class BaseDecodeHead(BaseModule):
    def __init__(self, in_channels, channels, *, num_classes=None, out_channels=None, threshold=None,
                 dropout_ratio=0.1, conv_cfg=None, norm_cfg=None, act_cfg=dict(type="ReLU"), in_index=-1,
                 input_transform=None, loss_decode=dict(type="CrossEntropyLoss", use_sigmoid=False, loss_weight=1.0),
                 ignore_index=255, sampler=None, align_corners=False, init_cfg=None):
        super().__init__(init_cfg)
        self.in_channels = in_channels
        self.in_index = in_index
        self.input_transform = input_transform
        self.channels = channels
        self.dropout_ratio = dropout_ratio
        self.norm_cfg = norm_cfg
        self.act_cfg = act_cfg
        self.ignore_index = ignore_index
        self.align_corners = align_corners
        self.num_classes = num_classes
        self.out_channels = out_channels or num_classes
        self.loss_decode = MODELS.build(loss_decode)
        self.sampler = None
        self.conv_seg = nn.Conv2d(channels, self.out_channels, kernel_size=1)
        self.dropout = nn.Dropout2d(dropout_ratio) if dropout_ratio > 0 else None

    def _transform_inputs(self, inputs):
        if self.input_transform == "multiple_select":
            return [inputs[i] for i in self.in_index]
        return inputs[self.in_index]

    def cls_seg(self, feat):
        if self.dropout is not None:
            feat = self.dropout(feat)
        return self.conv_seg(feat)

    def _stack_batch_gt(self, batch_data_samples):
        return torch.stack([d.gt_sem_seg.data for d in batch_data_samples], dim=0)

    def predict(self, inputs, batch_img_metas, test_cfg):
        seg_logits = self.forward(inputs)
        return self.predict_by_feat(seg_logits, batch_img_metas)

    def predict_by_feat(self, seg_logits, batch_img_metas):
        if isinstance(batch_img_metas[0]["img_shape"], torch.Size):
            size = batch_img_metas[0]["img_shape"]
        elif "pad_shape" in batch_img_metas[0]:
            size = batch_img_metas[0]["pad_shape"][:2]
        else:
            size = batch_img_metas[0]["img_shape"]
        return resize(seg_logits, size=size, mode="bilinear", align_corners=self.align_corners)


class _Cfg(dict):
    __getattr__ = dict.get

    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = _Cfg(v) if isinstance(v, dict) else v


class _PixelData:
    def __init__(self, data):
        self.data = data


class SegDataSample:
    def __init__(self, gt=None, metainfo=None):
        if gt is not None:
            self.gt_sem_seg = _PixelData(gt)
        self.metainfo = metainfo or {}


class _Preproc(nn.Module):
    def __init__(self, mean, std, **kw):
        super().__init__()
        self.register_buffer("mean", torch.tensor(mean).view(-1, 1, 1), False)
        self.register_buffer("std", torch.tensor(std).view(-1, 1, 1), False)


class EncoderDecoder(BaseModule):
    # mmseg 1.2.2 segmentors/encoder_decoder.py, restated
    def __init__(self, backbone, decode_head, neck=None, auxiliary_head=None, train_cfg=None, test_cfg=None,
                 data_preprocessor=None, pretrained=None, init_cfg=None):
        super().__init__(init_cfg)
        dp = dict(data_preprocessor or dict(mean=[0, 0, 0], std=[1, 1, 1]))
        dp.pop("type", None)
        self.data_preprocessor = _Preproc(**dp)
        self.backbone = MODELS.build(backbone)
        self.decode_head = MODELS.build(decode_head)
        self.align_corners = self.decode_head.align_corners
        self.num_classes = self.decode_head.num_classes
        self.out_channels = self.decode_head.out_channels
        self.train_cfg = _Cfg(train_cfg)
        self.test_cfg = _Cfg(test_cfg)
        self.with_neck = False

    def extract_feat(self, inputs):
        return self.backbone(inputs)

    def encode_decode(self, inputs, batch_img_metas):
        x = self.extract_feat(inputs)
        return self.decode_head.predict(x, batch_img_metas, self.test_cfg)

    def whole_inference(self, inputs, batch_img_metas):
        return self.encode_decode(inputs, batch_img_metas)

    def slide_inference(self, inputs, batch_img_metas):
        h_stride, w_stride = self.test_cfg.stride
        h_crop, w_crop = self.test_cfg.crop_size
        batch_size, _, h_img, w_img = inputs.size()
        h_grids = max(h_img - h_crop + h_stride - 1, 0) // h_stride + 1
        w_grids = max(w_img - w_crop + w_stride - 1, 0) // w_stride + 1
        preds = inputs.new_zeros((batch_size, self.out_channels, h_img, w_img))
        count_mat = inputs.new_zeros((batch_size, 1, h_img, w_img))
        for h_idx in range(h_grids):
            for w_idx in range(w_grids):
                y1 = h_idx * h_stride
                x1 = w_idx * w_stride
                y2 = min(y1 + h_crop, h_img)
                x2 = min(x1 + w_crop, w_img)
                y1 = max(y2 - h_crop, 0)
                x1 = max(x2 - w_crop, 0)
                crop_img = inputs[:, :, y1:y2, x1:x2]
                batch_img_metas[0]["img_shape"] = crop_img.shape[2:]
                crop_seg_logit = self.encode_decode(crop_img, batch_img_metas)
                preds += F.pad(crop_seg_logit, (int(x1), int(preds.shape[3] - x2), int(y1), int(preds.shape[2] - y2)))
                count_mat[:, :, y1:y2, x1:x2] += 1
        assert (count_mat == 0).sum() == 0
        return preds / count_mat


# ----------------------------------------------------------------------------- peft 0.10.0 LoRA, restated
class LoraConfig:
    def __init__(self, r, lora_alpha, target_modules, lora_dropout, bias="none"):
        self.r, self.lora_alpha, self.target_modules, self.lora_dropout = r, lora_alpha, list(target_modules), lora_dropout


class _LoraLinear(nn.Module):
    def __init__(self, base, r, alpha, p):
        super().__init__()
        self.base_layer = base
        self.lora_dropout = nn.ModuleDict({"default": nn.Dropout(p) if p > 0 else nn.Identity()})
        self.lora_A = nn.ModuleDict({"default": nn.Linear(base.in_features, r, bias=False)})
        self.lora_B = nn.ModuleDict({"default": nn.Linear(r, base.out_features, bias=False)})
        nn.init.kaiming_uniform_(self.lora_A["default"].weight, a=5 ** 0.5)
        nn.init.zeros_(self.lora_B["default"].weight)
        self.scaling = alpha / r

    @property
    def weight(self):
        return self.base_layer.weight

    @property
    def bias(self):
        return self.base_layer.bias

    def forward(self, x):
        y = self.base_layer(x)
        return y + self.lora_B["default"](self.lora_A["default"](self.lora_dropout["default"](x))) * self.scaling


class _Holder(nn.Module):
    def __init__(self, m):
        super().__init__()
        self.model = m

    def forward(self, *a, **k):
        return self.model(*a, **k)


class _PeftModel(nn.Module):
    def __init__(self, model):
        super().__init__()
        self.base_model = _Holder(model)  # keys: base_model.model.<...>

    def forward(self, *a, **k):
        return self.base_model(*a, **k)


def get_peft_model(model, cfg):
    targets = cfg.target_modules
    for name, mod in list(model.named_modules()):
        if isinstance(mod, nn.Linear) and any(name == t or name.endswith("." + t) for t in targets):
            parent = model
            parts = name.split(".")
            for p in parts[:-1]:
                parent = getattr(parent, p)
            setattr(parent, parts[-1], _LoraLinear(mod, cfg.r, cfg.lora_alpha, cfg.lora_dropout))
    for n, p in model.named_parameters():
        p.requires_grad = "lora_" in n
    return _PeftModel(model)


# ----------------------------------------------------------------------------- install
def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


_installed = False


def install():
    """Populate sys.modules with the stubs and synthetic parent packages."""
    global _installed
    if _installed:
        return
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError(f"reference tree not present at {REF_ROOT} (only exists in the build container)")
    os.environ.setdefault("XFORMERS_DISABLED", "1")
    _mod("mmseg"), _mod("mmseg.models"), _mod("mmengine"), _mod("mmcv")
    _mod("mmseg.models.builder", BACKBONES=MODELS, MODELS=MODELS)
    _mod("mmseg.registry", MODELS=MODELS)
    _mod("mmengine.model", BaseModule=BaseModule)
    _mod("mmengine.logging", MMLogger=_Logger)
    _mod("mmcv.cnn", ConvModule=ConvModule, build_norm_layer=build_norm_layer)
    _mod("mmseg.models.decode_heads")
    _mod("mmseg.models.decode_heads.decode_head", BaseDecodeHead=BaseDecodeHead)
    _mod("mmseg.models.utils", resize=resize)
    _mod("mmseg.models.losses", accuracy=accuracy)
    _mod("mmseg.models.segmentors", EncoderDecoder=EncoderDecoder)
    _mod("mmseg.utils", SampleList=list, add_prefix=add_prefix)
    _mod("mmseg.structures", SegDataSample=SegDataSample)
    _mod("peft", LoraConfig=LoraConfig, get_peft_model=get_peft_model)
    _mod("timm"), _mod("timm.models")
    _mod("timm.models.layers", drop_path=_drop_path, to_2tuple=_to_2tuple, trunc_normal_=nn.init.trunc_normal_)
    _mod("xformers")
    _mod("xformers.ops", memory_efficient_attention=_xformers_mea)
    sys.modules["xformers"].ops = sys.modules["xformers.ops"]
    _mod(PKG + ".models.backbones.beit", load_checkpoint=lambda *a, **k: None)
    # synthetic parent packages: real directories on __path__, their __init__.py NOT executed
    rein = os.path.join(REF_ROOT, "rein")
    for name, path in [
        (PKG, rein),
        (PKG + ".models", os.path.join(rein, "models")),
        (PKG + ".models.backbones", os.path.join(rein, "models", "backbones")),
        (PKG + ".models.heads", os.path.join(rein, "models", "heads")),
        (PKG + ".models.segmentors", os.path.join(rein, "models", "segmentors")),
    ]:
        m = _mod(name)
        m.__path__ = [path]
        m.__package__ = name
    _mod(PKG + ".utils", subplotimg=lambda *a, **k: None, add_prefix=add_prefix, resize=resize)
    _installed = True


def ref_import(dotted):
    """e.g. ref_import('models.backbones.dino_v2')"""
    install()
    return importlib.import_module(f"{PKG}.{dotted}")


def load_all():
    """Import the hot-path reference files; returns the registry."""
    ref_import("models.backbones.dino_v2")
    ref_import("models.backbones.lora_backbone")
    ref_import("models.heads.Transformer")
    ref_import("models.heads.linear_head")
    ref_import("models.heads.VFMHead")
    ref_import("models.segmentors.Ms_VFM_encoder_decoder")
    return MODELS


def load_sam():
    load_all()
    ref_import("models.backbones.sam_vit")
    return MODELS


def load_clip():
    load_all()
    ref_import("models.backbones.clip")
    return MODELS


def load_eva02():
    load_all()
    ref_import("models.backbones.eva_02")
    return MODELS
