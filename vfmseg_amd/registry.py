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
