"""Task registry — same dispatch as reference ``models/__init__.py:16-41``.

``build_model(params)`` keys on ``params['task']`` (``bg``, ``pc_transform``, plus the fused
``bg_forecast``), moves the model to the GPU unless ``params['no_gpu']`` and loads
``working_dir/best_model`` or ``load_model`` exactly like the reference.
"""
import os

import torch

from .bg_model import BGModel
from .pc_transform_model import PCTransformModel


def build_model(params):
    task = params['task']
    print("Building model for task: ", task)
    if task == 'bg':
        model = BGModel(params)
    elif task == 'pc_transform':
        model = PCTransformModel(params)
    elif task == 'bg_forecast':
        from .bg_forecast_model import BGForecastModel
        model = BGForecastModel(params)
    elif task in ('fg', 'odom'):
        raise ValueError('task %r is outside the MI355X hot path (SURVEY.md §8): use the reference for it' % task)
    else:
        raise ValueError('task not recognized: ', task)
    if not params.get('no_gpu'):
        model.to(torch.device('cuda'))
    if params.get('load_best_model'):
        model.load(os.path.join(params['working_dir'], 'best_model'))
    elif params.get('load_model'):
        print("LOADING MODEL FROM SPECIFIED PATH")
        model.load(params['load_model'])
    return model
