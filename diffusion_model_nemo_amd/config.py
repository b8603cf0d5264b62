"""Minimal Hydra/OmegaConf-compatible configuration loading for the reference's YAML files.

The reference instantiates its network, sampler and loss from `_target_` entries of a Hydra config
(diffusion_model_nemo/models/ddpm.py:27-29). This module loads those YAML files unchanged: `${a.b}`
interpolation, `???` mandatory values, dotted `a.b=c` overrides, attribute access, and `instantiate`, which maps
`diffusion_model_nemo.*` targets onto this package.
"""
import importlib
import re
from typing import Any, Iterable, Optional

import yaml

TARGET_ALIASES = (("diffusion_model_nemo.", "diffusion_model_nemo_amd."),)
_INTERP = re.compile(r"\$\{([^}]+)\}")


class MissingMandatoryValue(KeyError):
    pass


class Config(dict):
    """dict with attribute access; nested dicts are wrapped lazily."""

    def __getattr__(self, key):
        try:
            return self[key]
        except MissingMandatoryValue:
            raise
        except KeyError as e:
            raise AttributeError(key) from e

    def __setattr__(self, key, value):
        self[key] = value

    def __getitem__(self, key):
        v = dict.__getitem__(self, key)
        if isinstance(v, dict) and not isinstance(v, Config):
            v = Config(v)
            dict.__setitem__(self, key, v)
        if isinstance(v, str) and v == "???":
            raise MissingMandatoryValue(f"Missing mandatory value: {key}")
        return v

    def get(self, key, default=None):
        if key not in self:
            return default
        v = self[key]
        return default if v is None else v


def _lookup(root: dict, path: str):
    cur: Any = root
    for part in path.split("."):
        cur = dict.__getitem__(cur, part) if isinstance(cur, dict) else cur[int(part)]
    return cur


def _resolve(node, root, depth=0):
    if depth > 32:
        raise ValueError("interpolation cycle in config")
    if isinstance(node, dict):
        return Config({k: _resolve(v, root, depth) for k, v in node.items()})
    if isinstance(node, list):
        return [_resolve(v, root, depth) for v in node]
    if isinstance(node, str):
        m = _INTERP.fullmatch(node)
        if m:
            return _resolve(_lookup(root, m.group(1)), root, depth + 1)
        if _INTERP.search(node):
            return _INTERP.sub(lambda mm: str(_resolve(_lookup(root, mm.group(1)), root, depth + 1)), node)
    return node


def _parse_scalar(text: str):
    return yaml.safe_load(text)


def apply_overrides(cfg: dict, overrides: Iterable[str]):
    for ov in overrides:
        key, _, val = ov.partition("=")
        key = key.lstrip("+")
        cur = cfg
        parts = key.split(".")
        for p in parts[:-1]:
            if p not in cur or not isinstance(dict.__getitem__(cur, p), dict):
                dict.__setitem__(cur, p, {})
            cur = dict.__getitem__(cur, p)
        dict.__setitem__(cur, parts[-1], _parse_scalar(val))
    return cfg


def load_config(path: str, overrides: Optional[Iterable[str]] = None) -> Config:
    with open(path, "r") as f:
        raw = yaml.safe_load(f)
    if overrides:
        apply_overrides(raw, overrides)
    return _resolve(raw, raw)


def to_config(obj, overrides: Optional[Iterable[str]] = None) -> Config:
    """dict (possibly with interpolations) -> resolved Config."""
    import copy
    raw = copy.deepcopy(dict(obj))
    if overrides:
        apply_overrides(raw, overrides)
    return _resolve(raw, raw)


ALLOWED_TARGET_PREFIXES = ("diffusion_model_nemo_amd.",)


def resolve_target(target: str):
    """Maps a `_target_` string onto a class of THIS package. Only targets of the reference package
    (`diffusion_model_nemo.*`, aliased) or of this package are accepted: configs also arrive inside `.nemo` archives,
    and an arbitrary dotted path there would let a file name any importable callable to be run with its own arguments."""
    if not isinstance(target, str):
        raise ValueError(f"`_target_` must be a string, got {type(target).__name__}")
    for old, new in TARGET_ALIASES:
        if target.startswith(old):
            target = new + target[len(old):]
            break
    if not target.startswith(ALLOWED_TARGET_PREFIXES):
        raise ValueError(f"`_target_: {target}` is outside the diffusion_model_nemo / diffusion_model_nemo_amd packages; "
                         "refusing to import it")
    mod, _, name = target.rpartition(".")
    obj = getattr(importlib.import_module(mod), name)
    if not (isinstance(obj, type) and getattr(obj, "__module__", "").startswith(ALLOWED_TARGET_PREFIXES)):
        raise ValueError(f"`_target_: {target}` does not name a class defined in this package")
    return obj


def instantiate(cfg, **extra):
    """hydra.utils.instantiate for the subset the reference uses: a mapping with `_target_` and keyword arguments."""
    if cfg is None:
        return None
    if "_target_" not in cfg:
        raise ValueError("instantiate() needs a `_target_` entry")
    kwargs = {k: (dict(v) if isinstance(v, dict) else v) for k, v in dict.items(cfg) if k != "_target_"}
    kwargs.update(extra)
    return resolve_target(dict.__getitem__(cfg, "_target_"))(**kwargs)
