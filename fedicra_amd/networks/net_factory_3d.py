"""net_factory_3d (/root/reference/code/networks/net_factory_3d.py:7-20).  Only ``unet_3D`` is part of the hot-path
scope; the other 3D zoo entries of the reference (attention_unet, voxresnet, vnet) are not."""
from .unet_3D import unet_3D


def net_factory_3d(net_type="unet_3D", in_chns=1, class_num=2):
    if net_type == "unet_3D":
        return unet_3D(n_classes=class_num, in_channels=in_chns).cuda()
    if net_type in ("attention_unet", "voxresnet", "vnet"):
        raise NotImplementedError(f"net_factory_3d: '{net_type}' is outside the FedICRA hot-path scope (SURVEY.md 8)")
    return None
