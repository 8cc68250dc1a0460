"""net_factory with the reference's signature (/root/reference/code/networks/net_factory.py:6-32).

Model strings on the hot path are served by the HIP-backed modules; the rest of the reference's
table (unet_cct, unet_cct_3h, unet_ds, efficient_unet, pnet) is out of scope
and returns None exactly like the reference does for an unknown string.
"""
from .unet import UNet, UNet_Head, UNet_LC, UNet_LC_MultiHead, UNet_MultiHead


def net_factory(args, net_type="unet", in_chns=1, class_num=3):
    if net_type == "unet":
        net = UNet(in_chns=in_chns, class_num=class_num)
    elif net_type == "unet_head":
        net = UNet_Head(in_chns=in_chns, class_num=class_num)
    elif net_type == "unet_multihead":
        net = UNet_MultiHead(in_chns=in_chns, class_num=class_num)
    elif net_type == "unet_lc":
        net = UNet_LC(in_chns=in_chns, class_num=class_num, pcs_num=1, emb_num=args.min_num_clients,
                      client_num=args.min_num_clients, client_id=args.cid)
    elif net_type == "unet_lc_multihead":
        net = UNet_LC_MultiHead(in_chns=in_chns, class_num=class_num, pcs_num=1, emb_num=args.min_num_clients,
                                client_num=args.min_num_clients, client_id=args.cid)
    else:
        return None
    return net.cuda()
