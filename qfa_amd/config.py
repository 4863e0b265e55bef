"""Configuration with the reference's keys, without yacs.

Same tree, defaults, YAML merge (``BASE`` chaining), ``--opts KEY VALUE`` pairs and command-line overrides as
reference QFA/config.py:15-152 / main.py:16-44; the container is a plain attribute dict (``cfg.DATA.BATCH_SIZE``).
"""
from __future__ import annotations

import copy
import os

import yaml


class Node(dict):
    """dict with attribute access; ``dump()`` writes YAML like yacs' CfgNode.dump()"""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def dump(self):
        def plain(n):
            return {k: plain(v) if isinstance(v, dict) else v for k, v in n.items()}
        return yaml.safe_dump(plain(self), default_flow_style=False)


def _node(d):
    return Node({k: _node(v) if isinstance(v, dict) else v for k, v in d.items()})


DEFAULTS = {                                        # reference QFA/config.py:15-63
    "BASE": [""], "TYPE": "train", "GPU": 0,
    "DATA": {"DATA_DIR": "", "VALIDATION_DIR": "", "OUTPUT_DIR": "", "CATALOG": "", "VALIDATION_CATALOG": "",
             "DATA_NUM": 10000, "VALIDATION_NUM": 1000, "BATCH_SIZE": 500, "SNR_MIN": 2, "SNR_MAX": 100, "Z_MIN": 2,
             "Z_MAX": 3.5, "NUM_MASK": 0, "LAMMIN": 1030.0, "LAMMAX": 1600.0, "LOGLAM_DELTA": 1e-4, "NPROCS": 24,
             "VALIDATION": False},
    "MODEL": {"NH": 8, "TAU": "becker", "RESUME": "",
              # not in the reference: its load_from_npz reads c0 from the file's 'beta' entry (QFA/model.py:295, quirk
              # Q1).  True keeps that (the reference's shipped models and stored answers need it); set it to False
              # for models trained and saved by this package, whose files hold the true c0.
              "REFERENCE_C0_QUIRK": True},
    "TRAIN": {"NEPOCHS": 500, "LEARNING_RATE": 1e-3, "WEIGHT_DECAY": 1e-1, "DECAY_ALPHA": 0.9, "DECAY_STEP": 10,
              "WINDOW_LENGTH_FOR_MU": 16},
}

# command-line flag -> config key (reference QFA/config.py:92-139)
ARG_KEYS = {
    "gpu": "GPU", "n_epochs": "TRAIN.NEPOCHS", "learning_rate": "TRAIN.LEARNING_RATE",
    "weight_decay": "TRAIN.WEIGHT_DECAY", "decay_alpha": "TRAIN.DECAY_ALPHA", "decay_step": "TRAIN.DECAY_STEP",
    "data_dir": "DATA.DATA_DIR", "validation_dir": "DATA.VALIDATION_DIR", "output_dir": "DATA.OUTPUT_DIR",
    "catalog": "DATA.CATALOG", "validation_catalog": "DATA.VALIDATION_CATALOG", "data_num": "DATA.DATA_NUM",
    "validation_num": "DATA.VALIDATION_NUM", "batch_size": "DATA.BATCH_SIZE", "snr_min": "DATA.SNR_MIN",
    "snr_max": "DATA.SNR_MAX", "z_min": "DATA.Z_MIN", "z_max": "DATA.Z_MAX", "num_mask": "DATA.NUM_MASK",
    "nprocs": "DATA.NPROCS", "validation": "DATA.VALIDATION", "tau": "MODEL.TAU", "type": "TYPE",
}
# keys of DEFAULTS the reference does not have (tests/test_cli_config.py pins everything else against
# tests/golden/g12_config.json, extracted from the reference's config.py / main.py)
EXTRA_KEYS = ("MODEL.REFERENCE_C0_QUIRK",)


def _set(cfg, dotted, value):
    node = cfg
    keys = dotted.split(".")
    for k in keys[:-1]:
        node = node[k]
    if keys[-1] not in node:
        raise KeyError(f"unknown config key {dotted}")
    old = node[keys[-1]]
    if isinstance(value, str) and not isinstance(old, str):          # KEY VALUE pairs arrive as text
        if isinstance(old, bool):
            value = value.strip().lower() in ("1", "true", "yes", "on")
        elif isinstance(old, int):
            value = int(value)
        elif isinstance(old, float):
            value = float(value)
        else:
            value = yaml.safe_load(value)
    if isinstance(old, float) and isinstance(value, int) and not isinstance(value, bool):
        value = float(value)
    node[keys[-1]] = value


def _merge(cfg, other, where):
    for k, v in other.items():
        if k not in cfg:
            raise KeyError(f"unknown config key {k} in {where}")
        if isinstance(v, dict):
            _merge(cfg[k], v, where)
        else:
            cfg[k] = v


def merge_from_file(cfg, path):
    """YAML file over the config, files named in its BASE list first (reference QFA/config.py:67-77)"""
    with open(path) as f:
        y = yaml.safe_load(f) or {}
    for base in y.get("BASE", [""]):
        if base:
            merge_from_file(cfg, os.path.join(os.path.dirname(path), base))
    _merge(cfg, y, path)


def get_config(args=None):
    """defaults <- --cfg file <- --opts pairs <- explicit flags (reference QFA/config.py:80-152)"""
    cfg = _node(copy.deepcopy(DEFAULTS))
    if args is None:
        return cfg
    if isinstance(getattr(args, "cfg", None), str):
        merge_from_file(cfg, args.cfg)
    opts = getattr(args, "opts", None)
    if opts:
        if len(opts) % 2:
            raise ValueError("--opts takes KEY VALUE pairs")
        for k, v in zip(opts[0::2], opts[1::2]):
            _set(cfg, k, v)
    for flag, key in ARG_KEYS.items():
        v = getattr(args, flag, None)
        if v:                                            # the reference ignores falsy values too
            _set(cfg, key, v)
    if getattr(args, "Nh", None):
        # the reference parses --Nh (main.py:25) and never applies it (QFA/config.py:92-139 has no such branch)
        import warnings
        warnings.warn("--Nh is accepted and ignored, as in the reference; use --opts MODEL.NH <n>", stacklevel=2)
    return cfg
