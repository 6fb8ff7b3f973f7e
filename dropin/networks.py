"""Model factory at the reference's import path (reference: networks.py:9-22).

Put this directory in front of the reference checkout on ``sys.path`` (or copy
it over the checkout): ``inference.py`` and ``app.py`` then build the HIP
separation network without any edit.
"""
from torch import nn

_SS_NETWORKS = ("MossFormer2_SS_16K", "MossFormer2_SS_8K", "MossFormer2_SS_8K_noise")


class network_wrapper(nn.Module):
    """``network_wrapper(args).ss_network`` is the separation model."""

    def __init__(self, args):
        super().__init__()
        self.args = args
        if args.network not in _SS_NETWORKS:
            print(f"in networks, {args.network} is not found!")
            return
        from models.mossformer2.mossformer2 import MossFormer2_SS
        self.ss_network = MossFormer2_SS(args).model

    def forward(self, mixture, visual=None):
        return self.ss_network(mixture)
