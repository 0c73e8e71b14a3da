"""Test-only stand-in for torch_geometric.data.Data: an attribute bag with .to()."""
import torch


class Data(object):
    def __init__(self, x=None, edge_index=None, **kw):
        self.x = x
        self.edge_index = edge_index
        for k, v in kw.items():
            setattr(self, k, v)

    def to(self, device):
        for k, v in list(self.__dict__.items()):
            if torch.is_tensor(v):
                setattr(self, k, v.to(device))
        return self
