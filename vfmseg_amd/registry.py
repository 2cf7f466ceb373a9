"""Minimal registry with the mmengine surface the reference's configs rely on (`type="..."` dicts,
`@MODELS.register_module()`, `MODELS.build(cfg)`); rein/__init__.py:1-6 registers into mmseg's registries
by import side effect - here `import vfmseg_amd` does the same."""
import copy


class Registry:
    def __init__(self, name):
        self.name = name
        self._table = {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            key = name or cls.__name__
            if key in self._table and not force and self._table[key] is not cls:
                raise KeyError(f"{key} is already registered in {self.name}")
            self._table[key] = cls
            return cls

        if module is not None:
            return deco(module)
        return deco

    def get(self, key):
        return self._table.get(key)

    def __contains__(self, key):
        return key in self._table

    def build(self, cfg, **default_args):
        if cfg is None:
            return None
        if not isinstance(cfg, dict):
            return cfg  # already an object
        cfg = copy.deepcopy(dict(cfg))
        for k, v in default_args.items():
            cfg.setdefault(k, v)
        typ = cfg.pop("type")
        cls = self._table.get(typ) if isinstance(typ, str) else typ
        if cls is None:
            raise KeyError(f"{typ} is not registered in {self.name}; known: {sorted(self._table)}")
        return cls(**cfg)


MODELS = Registry("model")
BACKBONES = MODELS  # mmseg.models.builder.BACKBONES is the same registry object (dino_v2.py:17)
OPTIM_WRAPPER_CONSTRUCTORS = Registry("optim_wrapper_constructor")
METRICS = Registry("metric")
HOOKS = Registry("hook")
DATASETS = Registry("dataset")


def register_into_mmseg(force=True):
    """Optional: mirror every class into mmseg / mmengine's own registries, so that an UNMODIFIED mmengine Runner
    (tools/train.py of the reference, `Runner.from_cfg(cfg)`) resolves `type="MsVFMEncoderDecoder"` ... to the HIP-backed
    classes - what `import rein` does for the reference's classes (rein/__init__.py:1-6).  mmseg is not part of this image, so
    the call is a no-op returning False here; it is NOT exercised by the test-suite (no mmengine offline)."""
    try:
        from mmseg.registry import MODELS as MM_MODELS, METRICS as MM_METRICS  # type: ignore
        from mmengine.registry import OPTIM_WRAPPER_CONSTRUCTORS as MM_OWC  # type: ignore
    except Exception:
        return False
    for src, dst in ((MODELS, MM_MODELS), (METRICS, MM_METRICS), (OPTIM_WRAPPER_CONSTRUCTORS, MM_OWC)):
        for name, cls in src._table.items():
            dst.register_module(name=name, module=cls, force=force)
    return True
