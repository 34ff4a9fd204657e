"""TEST INFRASTRUCTURE -- import harness for the upstream reference (read-only at
/root/reference, present ONLY in the build container, never on the GPU box).

It makes `models.modules.*` of the reference importable on a CPU-only machine by stubbing the
third-party packages the image lacks (natsort, cv2, torchvision, pytorch_lightning) and
neutralising the hard-coded `.to('cuda...')` calls
(VQLLFLOWDeformable_arch.py:244, deformableDecoder_arch.py:548).  Nothing of the reference is
copied: the modules are executed in place to (a) pin the oracle (oracle/torch_ref.py) and
(b) generate the golden vectors under tests/golden/ (tests/golden/make_golden.py).

Only tests/ and the golden generator may import this file.
"""
import contextlib
import os
import sys
import types

REF_ROOT = os.environ.get("GLARE_REFERENCE", "/root/reference")
REF_CODE = os.path.join(REF_ROOT, "code")


def available():
    return os.path.isdir(os.path.join(REF_CODE, "models", "modules"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_installed = False


def install():
    """Put the reference on sys.path with the stubs in place (idempotent)."""
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError("reference not present at %s" % REF_ROOT)
    import torch.nn as nn

    for name in ("natsort", "cv2"):
        if name not in sys.modules:
            _stub(name)
    if "torchvision" not in sys.modules:
        tv = _stub("torchvision")
        tv.utils = _stub("torchvision.utils", save_image=lambda *a, **k: None, make_grid=lambda *a, **k: None)
        tv.models = _stub("torchvision.models")
        tv.models.vgg = _stub("torchvision.models.vgg")
        tv.transforms = _stub("torchvision.transforms")
    if "pytorch_lightning" not in sys.modules:
        _stub("pytorch_lightning", LightningModule=nn.Module)
    if REF_CODE not in sys.path:
        sys.path.insert(0, REF_CODE)
    import models.modules.VQModel_arch as vqm  # noqa: E402

    vqm.VGGFeatureExtractor = lambda *a, **k: nn.Identity()
    _installed = True


@contextlib.contextmanager
def cpu_only():
    """Make Tensor.to('cuda...') a no-op while the reference's graph code runs on CPU."""
    import torch

    orig = torch.Tensor.to

    def to(self, *args, **kwargs):
        args = tuple(a for a in args if not (isinstance(a, str) and a.startswith("cuda")))
        if isinstance(kwargs.get("device"), str) and kwargs["device"].startswith("cuda"):
            kwargs.pop("device")
        if not args and not kwargs:
            return self
        return orig(self, *args, **kwargs)

    torch.Tensor.to = to
    try:
        yield
    finally:
        torch.Tensor.to = orig


def load_opt(conf="LOL.yml"):
    install()
    import options.options as option

    cwd = os.getcwd()
    os.chdir(REF_CODE)
    try:
        opt = option.parse(os.path.join(REF_CODE, "confs", conf), is_train=False)
    finally:
        os.chdir(cwd)
    opt["gpu_ids"] = None
    return option.dict_to_nonedict(opt)


def build_netG(opt=None):
    """VQLLFLOWDeformable with the LOL.yml architecture (confs/LOL.yml:69-104)."""
    install()
    import models.networks as networks

    opt = opt or load_opt()
    return networks.define_Flow(opt, step=0), opt


def build_vqgan(opt=None):
    install()
    import models.networks as networks

    opt = opt or load_opt()
    return networks.find_vqgan(opt), opt


def import_harness():
    """The inference script's helper functions (impad, t, PSNR; infer_dataset_lol.py:42,71-72)."""
    install()
    for name in ("pyiqa", "lpips"):
        if name not in sys.modules:
            _stub(name)
    if "skimage" not in sys.modules:
        sk = _stub("skimage", img_as_ubyte=lambda x: x)
        sk.metrics = _stub("skimage.metrics", structural_similarity=None, peak_signal_noise_ratio=None)
    sys.modules["natsort"].natsort = sys.modules["natsort"]
    cv2 = sys.modules["cv2"]
    if not hasattr(cv2, "COLORMAP_JET"):
        cv2.COLORMAP_JET = 2  # only used as a default argument (utils/utils2.py:110)
    import importlib

    return importlib.import_module("infer_dataset_lol")
