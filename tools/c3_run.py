"""Functional / timing run of BASELINE config C3 on one GPU: FedICRA client (freeze schedule + LC loss with 7 no-grad
forwards), unet_lc, 12x3x512x512 bf16, hipGraph.  Not the headline benchmark (bench.py measures configs[1])."""
import sys, time, argparse, torch
sys.path.insert(0, '.')
from fedicra_amd import ops
from fedicra_amd.flower_common import MyModel
from fedicra_amd.flower_pCE_2D import MyClient
from fedicra_amd.networks import net_factory
from fedicra_amd.networks.unet import set_compute_dtype
from fedicra_amd.synth import phantom_batch
dev = torch.device('cuda', 0)
args = argparse.Namespace(strategy="FedICRA", amp=0, model="unet_lc", cid=2, min_num_clients=8, num_classes=3,
                          img_class="odoc", base_lr=0.01, max_iterations=30000, iters=6, rep_iters=3, alpha=1.0,
                          snapshot_path=None, use_graph=True)
torch.manual_seed(2022)
net = net_factory(args, net_type="unet_lc", in_chns=3, class_num=3)
set_compute_dtype(net, "bf16")
batches = []
for i in range(2):
    img, weak, _ = phantom_batch(12, 512, 3, 3, cid=2, index=i, labeled_frac=0.05)
    batches.append({"image": torch.from_numpy(img).to(dev), "label": torch.from_numpy(weak).to(dev)})
model = MyModel(args, net, batches, batches)
client = MyClient(args, model, batches, batches)
for r in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss, m = client._train({"iter_global": 60 + r, "iters": 6, "eval_iters": 12, "batch_size": 12, "stage": "fit"})
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"round {r}: loss {loss:.4f}  {dt*1e3/6:.2f} ms/iter  {12*6/dt:.1f} images/s  losses {client.last_losses}")
print("mem GB", torch.cuda.max_memory_allocated() / 1e9)
