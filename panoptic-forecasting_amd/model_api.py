"""What the registry hands back: an ``nn.Module`` that only HOLDS parameters and forwards work to libpfhip.so.

The surface is the one the reference's callers rely on (``models/base_model.py:11-23``): ``predict(inputs, labels)``
and ``loss(inputs, labels)`` on dicts of tensors, ``save(path)`` / ``load(path)`` of a bare ``state_dict``.  Checkpoints
are interchangeable with the reference's in both directions (same keys, see ``bg_model.py``).
"""
import os
import tempfile
from typing import Dict, Optional

import torch
from torch import Tensor, nn

TensorDict = Dict[str, Tensor]


class BaseModel(nn.Module):
    """Task model = parameters + two entry points; subclasses implement the entry points they support."""

    def predict(self, inputs: TensorDict, labels: Optional[TensorDict] = None) -> TensorDict:
        raise NotImplementedError('%s has no predict()' % type(self).__name__)

    def loss(self, inputs: TensorDict, labels: TensorDict) -> TensorDict:
        raise NotImplementedError('%s has no loss()' % type(self).__name__)

    # -- checkpoints: a bare state_dict, written atomically (a killed export job never leaves half a file behind)
    def save(self, path: str) -> None:
        folder = os.path.dirname(os.path.abspath(path))
        os.makedirs(folder, exist_ok=True)
        fd, tmp = tempfile.mkstemp(dir=folder, suffix='.tmp')
        os.close(fd)
        try:
            torch.save(self.state_dict(), tmp)
            os.replace(tmp, path)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)

    def load(self, path: str) -> None:
        state = torch.load(path, map_location='cpu')
        if isinstance(state, dict) and 'model_state' in state and not any(torch.is_tensor(v) for v in state.values()):
            state = state['model_state']      # a wrapped checkpoint (hardnet.py:393-400 style)
        self.load_state_dict(state)
