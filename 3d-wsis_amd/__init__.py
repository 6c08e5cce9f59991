"""3d-wsis_amd: MI355X-native hot path of fpthink/3D-WSIS.

The directory name is not a Python identifier on purpose (it is the repo's package directory, see
DESIGN.md); it is used by putting it on ``sys.path`` so that the reference's own import names resolve
to this implementation:

    import spconv, pointgroup_ops                      # drop-in operator surface
    from torch_scatter import scatter
    import backbone_3D_WSIS, losses_3D_WSIS           # Network / MultiTaskLoss mirrors (model/)

``bootstrap()`` does that path set-up; ``importlib.import_module("3d-wsis_amd")`` runs it on import.
"""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))


def bootstrap():
    for p in (os.path.join(_HERE, "model"), _HERE):
        if p not in sys.path:
            sys.path.insert(0, p)


bootstrap()
