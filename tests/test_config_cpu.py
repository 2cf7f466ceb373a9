"""Config loader / registry surface (CPU): _base_ inheritance, {{_base_.x}}, cfg-options, registry names."""
import os
import textwrap

import vfmseg_amd  # noqa: F401
from vfmseg_amd.config import Config, parse_cfg_options
from vfmseg_amd.registry import MODELS, OPTIM_WRAPPER_CONSTRUCTORS


def test_base_inheritance_and_substitution(tmp_path):
    (tmp_path / "base.py").write_text(textwrap.dedent("""
        crop_size = (512, 512)
        model = dict(type='EncoderDecoder', backbone=dict(type='X', depth=12), test_cfg=dict(mode='slide', stride=[320, 320]))
        pipeline = [dict(type='Resize', scale=(1024, 512)), dict(type='Pack')]
    """))
    (tmp_path / "child.py").write_text(textwrap.dedent("""
        from copy import deepcopy
        _base_ = ['./base.py']
        model = dict(backbone=dict(depth=24), test_cfg=dict(_delete_=True, mode='whole'))
        val = dict(pipeline={{_base_.pipeline}}, size={{_base_.crop_size}})
        lr = 1e-4 * 2
    """))
    cfg = Config.fromfile(str(tmp_path / "child.py"))
    assert cfg.model.type == "EncoderDecoder" and cfg.model.backbone.depth == 24 and cfg.model.backbone.type == "X"
    assert cfg.model.test_cfg == dict(mode="whole")
    assert cfg.val.pipeline[0]["type"] == "Resize" and tuple(cfg.val.size) == (512, 512)
    assert abs(cfg.lr - 2e-4) < 1e-12
    cfg.merge_from_dict(parse_cfg_options(["model.backbone.depth=2", "work_dir=foo"]))
    assert cfg.model.backbone.depth == 2 and cfg.work_dir == "foo"
    cfg.model.train_cfg = dict(work_dir="w")   # attribute mutation as tools/train.py:108-109 does
    assert cfg.model.train_cfg.work_dir == "w"


def test_registry_has_reference_names():
    for n in ["MsVFMEncoderDecoder", "LoraBackboneEncoderDecoder", "EncoderDecoder", "LoRABackbone", "DinoVisionTransformer",
              "LinearHead", "VFMHead", "MaskTransformerDecoder", "TransformerDecoder", "CrossEntropyLoss", "SegDataPreProcessor"]:
        assert n in MODELS, n
    assert "PEFTOptimWrapperConstructor" in OPTIM_WRAPPER_CONSTRUCTORS


def test_repo_config_builds_reference_key_scheme():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, "configs", "dg_lora_dinov2_ms_masked.py"))
    cfg.model.backbone.backbone.depth = 2
    m = MODELS.build(cfg.model)
    keys = set(m.state_dict())
    assert "backbone.model.base_model.model.blocks.1.attn.qkv.lora_A.default.weight" in keys
    assert "decode_head.output_upscaling.1.running_mean" in keys and "aux_decoder.transformer_decoder.mask_token" in keys
    from tests.helpers import model_shapes
    want = set(model_shapes(depth=2))
    assert keys == want, (sorted(keys - want)[:5], sorted(want - keys)[:5])


def test_bench_cpu_baseline_runs():
    """bench.py's cpu_baseline leg (the oracle timed on host cores) must keep working: it is only exercised at the end of a
    GPU bench run, so a stale name there goes unnoticed.  One oracle train step of the real configuration (~20 s on 8 cores)."""
    import importlib.util
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    sys.path.insert(0, root)
    spec.loader.exec_module(bench)
    out = bench.cpu_baseline()
    assert out["kind"] == "port" and out["unit"] == "images/s" and out["value"] > 0 and out["cores"] >= 1


def test_gate_that_cannot_fire_is_recognised():
    """segmentors.gate_cannot_fire decides whether a prediction's coarse pass may run beside its window pass: only when the confidence gate of
    Ms_VFM_encoder_decoder.py:430-461 (keep the coarse logits of a window whose fraction of pixels with max softmax > threadshod reaches conf)
    refines every window whatever the coarse logits say.  The reference's own setting (0.968 / 0.8) is a live gate."""
    from vfmseg_amd import presets
    from vfmseg_amd.segmentors import gate_cannot_fire
    assert gate_cannot_fire(0.968, 2.0) and gate_cannot_fire(1.0, 0.8) and gate_cannot_fire(1.5, 1e-6)
    assert not gate_cannot_fire(0.968, 0.8) and not gate_cannot_fire(0.999, 1.0) and not gate_cannot_fire(1.0, 0.0)
    tc = presets.dinov2_ms_masked()["test_cfg"]
    assert not gate_cannot_fire(tc["threadshod"], tc["conf"])
    # brute force against the gate's own arithmetic: fraction = (#pixels with p_max > thr) / #pixels, p_max in [0, 1]
    import itertools
    for thr, conf in itertools.product((0.0, 0.5, 0.968, 1.0, 1.2), (0.0, 0.3, 0.8, 1.0, 1.01)):
        always_refined = all((sum(p > thr for p in ps) / len(ps)) < conf for ps in ((0.0, 0.0), (1.0, 1.0), (0.2, 1.0), (0.97, 0.99)))
        if gate_cannot_fire(thr, conf):
            assert always_refined, (thr, conf)
