"""Base class for the models -- mirror of reference models/base.py:1-31."""

import numpy as np
import torch.nn as nn


class BaseModel(nn.Module):
    """Base class for all models (reference: models/base.py:8-31)."""

    def forward(self, *inputs):
        raise NotImplementedError

    def __str__(self):
        params = sum(int(np.prod(p.size())) for p in self.parameters() if p.requires_grad)
        return super().__str__() + "\nTrainable parameters: {}".format(params)
