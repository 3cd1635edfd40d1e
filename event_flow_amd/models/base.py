"""Common parent of the model classes (the role of reference models/base.py): an `nn.Module` whose printout ends
with the number of trainable parameters, as the reference's does."""

import torch.nn as nn


class BaseModel(nn.Module):
    def forward(self, *inputs):
        raise NotImplementedError(f"{type(self).__name__} does not define forward()")

    def trainable_parameters(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)

    def __str__(self):
        return f"{super().__str__()}\nTrainable parameters: {self.trainable_parameters()}"
