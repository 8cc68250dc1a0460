"""Import shim for the reference (THIS CONTAINER ONLY; test tooling, never shipped as product).

Makes /root/reference/code importable on a CPU-only box: ``.cuda()`` becomes a
no-op and the third-party modules the image lacks (flwr, torchvision, medpy,
tree_filter_cuda, efficientnet_pytorch, tensorboardX, h5py, cv2) are registered
as empty stand-in *modules* so that ``import`` statements succeed -- none of
their functionality is used by the vectors gen_golden.py produces.  Used by
oracle/gen_golden.py to generate tests/golden/*.npz; nothing on the GPU box
imports this file (there is no /root/reference there).
"""
import collections
import sys
import types

import torch
import torch.nn as nn

REF_CODE = "/root/reference/code"


def install():
    if getattr(install, "_done", False):
        return
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        m.__path__ = []
        sys.modules[name] = m
        return m

    class Anything:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return None

        def __getattr__(self, n):
            return Anything()

    class StrategyBase:
        def __init__(self, **kw):
            self.__dict__.update(kw)

    fl = stub("flwr")
    fl.client = stub("flwr.client", Client=object)
    fl.common = stub("flwr.common", GetParametersRes=Anything, Status=Anything, FitRes=Anything,
                     EvaluateRes=Anything, GetPropertiesIns=Anything, GetPropertiesRes=Anything,
                     ndarrays_to_parameters=lambda x: x, parameters_to_ndarrays=lambda x: x)
    stub("flwr.common.logger", log=lambda *a, **k: None)
    fl.server = stub("flwr.server")
    stub("flwr.server.server", Server=object, fit_clients=None, evaluate_clients=None)
    stub("flwr.server.history", History=Anything)
    stub("flwr.server.strategy")
    stub("flwr.server.strategy.strategy", Strategy=object)
    for mod, cls in [("fedavg", "FedAvg"), ("fedadagrad", "FedAdagrad"), ("fedadam", "FedAdam"), ("fedyogi", "FedYogi")]:
        stub("flwr.server.strategy." + mod, **{cls: StrategyBase})
    stub("torchvision")
    stub("torchvision.utils", make_grid=None)
    stub("torchvision.models")
    stub("torchvision.models.densenet", DenseNet=object)
    stub("torchvision.models.resnet", BasicBlock=object, Bottleneck=object, ResNet=object)
    stub("medpy", metric=Anything())
    stub("tree_filter_cuda")
    stub("efficientnet_pytorch", EfficientNet=type("EfficientNet", (nn.Module,), {}))
    stub("efficientnet_pytorch.utils", url_map=collections.defaultdict(str),
         url_map_advprop=collections.defaultdict(str), get_model_params=None)
    # extra stand-ins needed to import flower_pCE_2D (MyClient._train) -- import-only
    stub("tensorboardX", SummaryWriter=Anything)
    sys.modules["torchvision"].transforms = stub("torchvision.transforms", Compose=Anything)
    sys.modules["flwr.server"].ServerConfig = Anything
    stub("flwr.server.client_manager", SimpleClientManager=Anything)
    sys.modules["flwr.common"].ndarray_to_bytes = lambda a: a
    stub("cv2")
    stub("h5py")
    sys.modules["medpy"].metric = stub("medpy.metric", binary=Anything())
    stub("skimage", measure=Anything())
    stub("skimage.measure")
    sys.path.insert(0, REF_CODE)
    install._done = True
