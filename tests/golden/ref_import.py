"""Import the PlaneDepth reference (read-only, /root/reference) in THIS container.

Test infrastructure only.  Used by ``make_golden.py`` to generate the golden
vectors committed next to it, and by ``tests/test_oracle_vs_reference.py``
(skipped automatically when /root/reference is absent, e.g. on the GPU box).
Nothing from the reference is copied: its modules are imported from where they
lie.  The reference needs packages this image lacks, so a few
empty stand-in modules are registered in ``sys.modules`` and ``.cuda()`` is
turned into the identity (there is no GPU here; the hot path then runs on CPU).
"""
import importlib
import os
import sys
import types

import torch
from torch import nn

REFERENCE_ROOT = os.environ.get("PLANEDEPTH_REFERENCE", "/root/reference")


def reference_available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "trainer.py"))


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _install_stand_ins():
    if "torchvision" not in sys.modules:
        class _ResNet(nn.Module):  # only ever subclassed, never run, on the hot path
            def __init__(self, *a, **k):
                super().__init__()

        resnet = _module("torchvision.models.resnet", ResNet=_ResNet,
                         BasicBlock=object, Bottleneck=object)
        models = _module("torchvision.models", resnet=resnet, ResNet=_ResNet)
        class _Normalize(nn.Module):  # PladeNet normalises its input image (plade_net.py:248); the backbone's output is
            def __init__(self, mean, std):  # replaced by hooks in the golden generator, so only the call has to exist
                super().__init__()
                self.mean, self.std = torch.tensor(mean)[None, :, None, None], torch.tensor(std)[None, :, None, None]

            def forward(self, x):
                return (x - self.mean) / self.std

        transforms = _module("torchvision.transforms", Normalize=_Normalize)
        _module("torchvision", models=models, transforms=transforms)
    if "torch._six" not in sys.modules:
        _module("torch._six", string_classes=(str, bytes))
    if "tensorboardX" not in sys.modules:
        _module("tensorboardX", SummaryWriter=object)
    if "IPython" not in sys.modules:
        _module("IPython", embed=lambda *a, **k: None)
    if "skimage" not in sys.modules:
        tr = _module("skimage.transform")
        _module("skimage", transform=tr)
    if "cv2" not in sys.modules:
        _module("cv2")


_loaded = {}


def load_reference():
    """Returns a namespace with the reference's ``layers``, ``trainer`` and ``networks`` modules."""
    if _loaded:
        return types.SimpleNamespace(**_loaded)
    if not reference_available():
        raise RuntimeError("reference tree not found at %s" % REFERENCE_ROOT)
    sys.dont_write_bytecode = True
    _install_stand_ins()
    # No GPU in the oracle container: make .cuda() a no-op so constants stay on CPU.
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    for name in ("layers", "utils", "networks", "trainer"):
        _loaded[name] = importlib.import_module(name)
    return types.SimpleNamespace(**_loaded)


def make_trainer_namespace(ref, H, W, **opt):
    """A stand-in for ``self`` good enough for the unbound hot-path methods of the reference Trainer."""
    defaults = dict(warp_type="disp_warp", match_aug=False, use_mixture_loss=True,
                    render_probability=False, automask=False, alpha_pc=0.0, alpha_self=0.0,
                    self_distillation=0.0, gamma_smooth=2.0, alpha_smooth=0.04, use_ssim=True)
    defaults.update(opt)
    target_sides = defaults.pop("target_sides", ["r"])
    ns = types.SimpleNamespace()
    ns.opt = types.SimpleNamespace(**defaults)
    ns.target_sides = target_sides
    ns.softmax = nn.Softmax(1)
    ns.ssim = ref.layers.SSIM()
    ns.backproject_depth = ref.layers.BackprojectDepth(H, W)
    ns.project_3d = ref.layers.Project3D(H, W)
    ns.homography_warp = ref.layers.HomographyWarp(H, W)
    ns.perceptual_loss = lambda *a, **k: torch.zeros(())  # VGG weights unavailable; out of scope
    ns.compute_reprojection_loss = lambda pred, target: ref.trainer.Trainer.compute_reprojection_loss(ns, pred, target)
    return ns
