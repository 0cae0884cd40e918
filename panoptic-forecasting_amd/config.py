"""Command-line + YAML configuration with the reference's flag set and precedence (``utils/config.py:34-95``), so the
reference's shell scripts (``scripts/bg/run_export_bg_val.sh``, ``scripts/bg/run_bg_train.sh``) drive this package with
only the python path changed.

Precedence, lowest to highest: the ``config.yaml`` stored next to a checkpoint (``--load_model``) or in ``--working_dir``
(``--load_best_model`` / ``--continue_training``)  <  ``--config_file``  <  the command line itself (every parsed
flag becomes a top-level key, also when it is None — the reference merges ``vars(args)`` wholesale)  <
``--extra_args name.subname value`` pairs (dotted path, value parsed as list / bool / int / float / string).
"""
import argparse
import os

import yaml

BASE_FLAGS = (
    ('--working_dir', dict(required=True)),
    ('--config_file', {}),
    ('--no_gpu', dict(action='store_true')),
    ('--seed', dict(type=int, default=1)),
    ('--load_model', {}),
    ('--continue_training', dict(action='store_true')),
    ('--load_best_model', dict(action='store_true')),
    ('--extra_args', dict(nargs=2, action='append')),
    ('--dist_url', dict(default='env://', help='url used to set up distributed training')),
)


def parse_value(text):
    """'[a, b]' -> list (recursively), true/false -> bool, integers without a '.', floats, else the string itself."""
    if len(text) >= 2 and text[0] == '[' and text[-1] == ']':
        return [parse_value(part.strip()) for part in text[1:-1].split(',')]
    if text in ('True', 'true'):
        return True
    if text in ('False', 'false'):
        return False
    if '.' not in text:
        try:
            return int(text)
        except ValueError:
            pass
    try:
        return float(text)
    except ValueError:
        return text


def overlay(base, top):
    """Nested-dict merge: ``top`` wins; dicts present on both sides are merged key by key."""
    out = dict(base)
    for key, val in top.items():
        if key in base and isinstance(base[key], dict) and isinstance(val, dict):
            out[key] = overlay(base[key], val)
        else:
            out[key] = val
    return out


def set_dotted(params, dotted, value):
    node = params
    parts = dotted.split('.')
    for part in parts[:-1]:
        node = node.setdefault(part, {})
    node[parts[-1]] = value


def _read_yaml(path):
    with open(path) as f:
        return yaml.safe_load(f) or {}


def load_config(extra_flags=None, argv=None):
    """``extra_flags``: iterable of (flag, argparse kwargs) a driver adds to the base set; returns the params dict."""
    parser = argparse.ArgumentParser('')
    for flag, kw in tuple(BASE_FLAGS) + tuple(extra_flags or ()):
        parser.add_argument(flag, **kw)
    args = parser.parse_args(argv)
    stored = None
    if args.load_best_model or args.continue_training:
        stored = os.path.join(args.working_dir, 'config.yaml')
    elif args.load_model:
        stored = os.path.join(os.path.dirname(args.load_model), 'config.yaml')
    params = _read_yaml(stored) if stored else {}
    if args.config_file is not None:
        params = overlay(params, _read_yaml(args.config_file))
    params = overlay(params, vars(args))
    for name, text in (args.extra_args or ()):
        set_dotted(params, name, parse_value(text))
    return params
