"""net_factory_3d (/root/reference/code/networks/net_factory_3d.py:7-20): ``unet_3D`` and ``vnet`` (SURVEY.md section 8
a18); attention_unet and voxresnet of the reference's 3D zoo are not built."""
from .unet_3D import unet_3D, unet_3D_lc
from .vnet import VNet


def net_factory_3d(net_type="unet_3D", in_chns=1, class_num=2, args=None):
    if net_type == "unet_3D":
        return unet_3D(n_classes=class_num, in_channels=in_chns).cuda()
    if net_type == "unet_3D_lc":            # BASELINE configs[4]; args like net_factory's LC models (net_factory.py:24-26)
        return unet_3D_lc(n_classes=class_num, in_channels=in_chns, client_num=args.min_num_clients,
                          client_id=args.cid).cuda()
    if net_type == "vnet":
        return VNet(n_channels=in_chns, n_classes=class_num, normalization="batchnorm", has_dropout=True).cuda()
    if net_type in ("attention_unet", "voxresnet"):
        raise NotImplementedError(f"net_factory_3d: '{net_type}' is outside the FedICRA hot-path scope (SURVEY.md 8)")
    return None
