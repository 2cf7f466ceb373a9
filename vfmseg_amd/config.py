"""A small loader for mmengine-style python configs - the features the reference's config tree uses
(SURVEY.md §5 "Config / flags"): `_base_` list inheritance with recursive dict merge (`_delete_=True` supported),
`{{_base_.name}}` substitution, arbitrary python in the file, attribute access + mutation, `--cfg-options k.a=v`.
mmengine itself is not available offline, and the configs are data a user brings from their reference checkout."""
import ast
import copy
import os
import re


class ConfigDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = _wrap(v)

    def __deepcopy__(self, memo):
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _wrap(v):
    if isinstance(v, dict) and not isinstance(v, ConfigDict):
        return ConfigDict({k: _wrap(x) for k, x in v.items()})
    if isinstance(v, list):
        return [_wrap(x) for x in v]
    if isinstance(v, tuple):
        return tuple(_wrap(x) for x in v)
    return v


def _merge(base, new):
    """mmengine Config._merge_a_into_b: dicts merge recursively unless the child says _delete_=True."""
    out = copy.deepcopy(base)
    for k, v in new.items():
        if isinstance(v, dict) and k in out and isinstance(out[k], dict) and not v.get("_delete_", False):
            out[k] = _merge(out[k], v)
        else:
            if isinstance(v, dict):
                v = {a: b for a, b in v.items() if a != "_delete_"}
            out[k] = copy.deepcopy(v)
    return out


_BASE_VAR = re.compile(r"\{\{\s*_base_\.([\w\.]+)\s*\}\}")


def _load_file(path):
    path = os.path.abspath(path)
    src = open(path).read()
    # pre-read `_base_` (a literal str or list of str at module level)
    base_files = []
    for node in ast.parse(src).body:
        if isinstance(node, ast.Assign) and any(getattr(t, "id", None) == "_base_" for t in node.targets):
            val = ast.literal_eval(node.value)
            base_files = [val] if isinstance(val, str) else list(val)
    base = {}
    for b in base_files:
        bcfg = _load_file(os.path.join(os.path.dirname(path), b))
        dup = set(base) & set(bcfg)
        if dup:
            raise KeyError(f"duplicate keys {sorted(dup)} in the bases of {path}")
        base.update(bcfg)

    # {{_base_.a.b}} -> python literal of the base value
    def sub(m):
        cur = base
        for part in m.group(1).split("."):
            cur = cur[part]
        return repr(cur)

    src = _BASE_VAR.sub(sub, src)
    src = re.sub(r"[\"']\{\{.*?\}\}[\"']", lambda m: m.group(0), src)
    scope = {"__file__": path}
    exec(compile(src, path, "exec"), scope)
    own = {k: v for k, v in scope.items() if not k.startswith("__") and k != "_base_" and not callable(v) and not isinstance(v, type(os))}
    return _merge(base, own)


class Config(ConfigDict):
    @staticmethod
    def fromfile(path):
        cfg = Config(_wrap(_load_file(path)))
        dict.__setattr__(cfg, "filename", os.path.abspath(path))
        return cfg

    def merge_from_dict(self, options):
        """--cfg-options: keys like 'model.backbone.depth' (values already python objects)."""
        for key, val in (options or {}).items():
            cur = self
            parts = key.split(".")
            for p in parts[:-1]:
                if p not in cur or not isinstance(cur[p], dict):
                    cur[p] = ConfigDict()
                cur = cur[p]
            cur[parts[-1]] = _wrap(val)


def parse_cfg_options(items):
    """['a.b=1', 'c=[1,2]', 'd=str'] -> dict (mmengine DictAction subset)."""
    out = {}
    for it in items or []:
        k, v = it.split("=", 1)
        try:
            out[k] = ast.literal_eval(v)
        except Exception:
            out[k] = v
    return out
