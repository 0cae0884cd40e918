"""Task registry: ``build_model(params)`` for the tasks of the MI355X path.

Call contract of the reference's ``panoptic_forecasting.models.build_model`` (``models/__init__.py:16-41``), so its
experiment scripts run with one import changed: the task comes from ``params['task']``; the model is
moved to the GPU unless ``params['no_gpu']``; ``load_best_model`` restores ``<working_dir>/best_model``, otherwise
``load_model`` names a checkpoint.  Tasks outside the hot path (``fg``, ``odom``) are refused with a pointer to the
reference — this package has no fallback implementations.
"""
import importlib
import os

import torch

# task -> (module, class); imported lazily so that e.g. ``pc_transform`` users never touch the HarDNet packing code
_TASKS = {
    'bg': ('bg_model', 'BGModel'),
    'pc_transform': ('pc_transform_model', 'PCTransformModel'),
    'bg_forecast': ('bg_forecast_model', 'BGForecastModel'),       # the two stages fused on the device (new)
}
_OUT_OF_SCOPE = ('fg', 'odom')


def _checkpoint_path(params):
    if params.get('load_best_model'):
        return os.path.join(params['working_dir'], 'best_model')
    return params.get('load_model') or None


def build_model(params):
    task = params['task']
    print("Building model for task: ", task)          # the reference announces the task the same way
    if task in _OUT_OF_SCOPE:
        raise ValueError('task %r is outside the MI355X hot path (SURVEY.md §8): use the reference for it' % task)
    if task not in _TASKS:
        raise ValueError('task not recognized: ', task)
    module, cls = _TASKS[task]
    model = getattr(importlib.import_module('.' + module, __package__), cls)(params)
    if not params.get('no_gpu'):
        model.to(torch.device('cuda'))
    ckpt = _checkpoint_path(params)
    if ckpt:
        model.load(ckpt)
    return model
