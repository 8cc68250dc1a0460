"""The paper's full procedure ("Ours": pCE + multi-scale tree-energy loss + 0.1 x gated CRF loss [+ FedICRA's LC loss]) --
mirror of /root/reference/code/flower_pCE_2D_GateCRFMsacleTreeEnergyLoss_Ours.py `MyClient._train` (:52-198): same batch
selection, freeze schedule, optimizer and learning-rate rule as the pCE client (flower_pCE_2D.py, inherited), with the
iteration's loss extended as at :131-163:

    loss = CE(ignore = num_classes) + MScaleRecurveTreeEnergyLoss(outputs, image x3, aux1, aux2, aux3, unlabeled, w_tree)
           + 0.1 * GatedCRF(softmax(outputs), [{weight 1, xy 6, rgb 0.1}], radius 5, image)       [+ alpha * loss_lc]

Only the two multi-head models define the three auxiliary maps the reference dereferences (`unet_multihead`,
`unet_lc_multihead`, :118-126); with any other model string the reference raises NameError at :142 -- here a ValueError
up front.  Everything device-side stays hipGraph-capturable (no host sync inside the iteration).
"""
from __future__ import annotations

import torch

from . import ops, streams
from .flower_pCE_2D import MyClient as _PCEClient
from .tree_energy import MScaleRecurveTreeEnergyLoss
from .utils.gate_crf_loss import ModelLossSemsegGatedCRF


class MyClient(_PCEClient):
    def __init__(self, args, model, trainloader, valloader, amp=False):
        super().__init__(args, model, trainloader, valloader, amp=amp)
        if args.model not in ("unet_multihead", "unet_lc_multihead"):
            raise ValueError("the tree-energy procedure needs a multi-head model (aux_output1..3): got %r" % (args.model,))
        self.tree_loss_multi = MScaleRecurveTreeEnergyLoss()
        self.gatecrf_loss = ModelLossSemsegGatedCRF()
        self.loss_gatedcrf_kernels_desc = [{"weight": 1, "xy": 6, "rgb": 0.1}]          # :68-69
        self.loss_gatedcrf_radius = 5
        self._lc_stream = None

    def _iteration(self, x, y, rec):
        args, opt = self.args, self.optimizer
        ops.begin_iteration(x.device)
        opt.zero_grad()
        out = self.model(x)
        if args.model == "unet_multihead":                                               # :121-123
            outputs, aux = out[0], out[6:9]
            heatmaps = None
        else:                                                                            # unet_lc_multihead, :127-128
            outputs, heatmaps, aux = out[0], out[6], out[7:10]
        loss_ce = ops.ce_loss(outputs.permute(0, 2, 3, 1), y, args.num_classes)          # :135
        unlabeled = (y == args.num_classes)                                              # :136
        three_channel = x.repeat(1, 3, 1, 1) if args.img_class == "faz" else x           # :138-141
        # The LC loss's no-grad forwards (:153-163) need nothing from the losses above them, and the tree-energy loss is
        # ~10 ms of dependent chains on a few dozen workgroups: the forwards run on a side stream beside it (no autograd
        # node is created there; fork / join on the current stream, so a captured iteration keeps a parallel branch).
        others = []
        if args.strategy in ["FedICRA"]:
            cur = torch.cuda.current_stream()
            self._lc_stream = streams.get("probe", x.device)   # (fedicra_amd/streams.py: never the stream this iteration runs / is captured on)
            assert self._lc_stream.cuda_stream != cur.cuda_stream
            self._lc_stream.wait_stream(cur)
            with torch.cuda.stream(self._lc_stream), torch.no_grad():
                ids = [c for c in range(args.min_num_clients) if c != args.cid]
                batched = self.model.model.probe_heatmaps(x, ids) if hasattr(self.model.model, "probe_heatmaps") else None
                if batched is not None:                      # all K-1 forwards as one batch of statistics groups
                    others = [h.detach() for h in batched]
                else:
                    for other_client in ids:
                        # logits and head outputs of these forwards are never read
                        others.append(self.model(x, other_client, heatmap_only=True)[-4][-1].detach())
        out_tree_loss = self.tree_loss_multi(outputs, three_channel, aux[0], aux[1], aux[2], unlabeled,
                                             args.tree_loss_weight)[0]                   # :142
        outputs_soft = torch.softmax(outputs.float(), dim=1)
        out_gatedcrf = self.gatecrf_loss(outputs_soft, self.loss_gatedcrf_kernels_desc, self.loss_gatedcrf_radius, x,
                                         x.shape[2], x.shape[3])["loss"]                 # :143-150
        loss = loss_ce + out_tree_loss + 0.1 * out_gatedcrf                              # :151
        loss_lc = None
        if args.strategy in ["FedICRA"]:                                                 # :153-163
            cur.wait_stream(self._lc_stream)
            acc = 0
            for hm in others:
                acc = acc + torch.nn.functional.mse_loss(heatmaps[-1], hm)
            loss_lc = -acc / (args.min_num_clients - 1)
            loss = torch.add(loss, loss_lc, alpha=args.alpha)
        if self.amp:
            self.scaler.scale(loss).backward()
            self.scaler.step(opt)
            self.scaler.update()
        else:
            loss.backward()
            opt.step()
        opt.advance_lr()
        rec.loss, rec.loss_ce, rec.loss_lc, rec.logits = loss.detach(), loss_ce.detach(), \
            (None if loss_lc is None else loss_lc.detach()), outputs.detach()
        rec.loss_tree, rec.loss_crf = out_tree_loss.detach(), out_gatedcrf.detach()
