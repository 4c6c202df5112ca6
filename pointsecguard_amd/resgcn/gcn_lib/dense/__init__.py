"""`gcn_lib.dense`: the names the reference's architecture.py imports (ResGCN/gcn_lib/dense/__init__.py) --
layer containers, dilated kNN graph modules and the EdgeConv / dynamic-graph blocks -- re-exported explicitly."""
from . import torch_edge as _edge
from . import torch_nn as _nn
from . import torch_vertex as _vertex

for _module in (_nn, _edge, _vertex):
    for _name in getattr(_module, "__all__", [n for n in vars(_module) if not n.startswith("_")]):
        globals()[_name] = getattr(_module, _name)
del _module, _name
