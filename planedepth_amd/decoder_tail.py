"""Drop-in for the tail of the reference ``DepthDecoder.forward`` (networks/depth_decoder.py:256-291, softmax branch).

The reference computes, after its last convolutions::

    logits = self.convs["dispconv"](x) * padding_mask ; probability = softmax(logits)
    sigma = clamp(sigmoid(self.convs["sigmaconv"](x)), 0.01, 1) ; pi = probability
    probability = (pi / sigma * padding_mask) / sum ; disp = sum(probability * disp_layered) ; depth = 0.1*0.58*W/disp

as ~12 full-tensor passes.  ``fused_decoder_tail`` does it in one HIP kernel (and one for the backward) and writes the
same ``outputs`` entries.  It stays opt-in because ``networks/*`` are meant to drop in unchanged:
a maintainer replaces lines 256-291 of ``depth_decoder.py`` with::

    from planedepth_amd.decoder_tail import fused_decoder_tail
    fused_decoder_tail(self.outputs, self.convs["dispconv"](x),
                       self.convs["sigmaconv"](x) if self.use_mixture_loss else None,
                       use_mixture_loss=self.use_mixture_loss, all_ones_mask=(self.xz_levels + self.yz_levels == 0))

``DepthDecoder`` with ``--render_probability`` keeps the reference's own code (its compositing branch raises on its own: the
padding mask has N channels, ``dispconv`` N-1).  The live producer of ``outputs["dists"]`` is ``PladeNet``
(networks/plade_net.py:309-341), whose tail ``fused_plade_tail`` replaces the same way: lines 309-340 become::

    from planedepth_amd.decoder_tail import fused_plade_tail
    fused_plade_tail(self.outputs, self.conv0(dlog), self.conv_sigma(features) if self.use_mixture_loss else None,
                     use_mixture_loss=self.use_mixture_loss)
"""
import torch

from . import ops


class LazyLayers:
    """``outputs["probability"]`` / ``outputs["pi"]`` stand-in: has ``.shape`` (all the training loop reads,
    trainer.py:528, 610, 704) and materialises the tensor on first real use."""

    def __init__(self, shape, make, device=None, dtype=None):
        self.shape = tuple(shape)
        self.device = device
        self.dtype = dtype
        self._make = make
        self._value = None

    def dim(self):
        return len(self.shape)

    def size(self, d=None):
        return self.shape if d is None else self.shape[d]

    def tensor(self):
        """The materialised [B,N,H,W] tensor.  It carries NO gradient (nothing in the reference's losses back-propagates
        through ``probability`` / ``pi``: trainer.py reads them for their shape, the post-process under no_grad)."""
        if self._value is None:
            if torch.is_grad_enabled() and not getattr(LazyLayers, "_warned", False):
                import warnings
                LazyLayers._warned = True
                warnings.warn("planedepth_amd: outputs['probability'] / outputs['pi'] of the fused decoder tail are "
                              "materialised WITHOUT gradient (the reference's losses never back-propagate through them; "
                              "gradients flow through 'disp', 'logits' and 'sigma').  Use the unfused decoder if a custom "
                              "loss needs them differentiable.", stacklevel=3)
            self._value = self._make()
        return self._value

    def __getattr__(self, name):          # anything beyond shape / device / dtype: behave like the tensor
        # private and dunder names are never forwarded: copy / pickle look them up on objects created without __init__,
        # where forwarding would recurse through `_value` for ever
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.tensor(), name)

    def __getitem__(self, idx):
        return self.tensor()[idx]


def fused_decoder_tail(outputs, dispconv_out, sigmaconv_out=None, *, use_mixture_loss=True, all_ones_mask=False,
                       materialize_layers=False, fuse_sweep_backward=False):
    """Fills ``outputs`` with "logits", "sigma", "pi", "probability", "disp", "depth" as depth_decoder.py:258-291 does.
    Reads ``outputs["disp_layered"]`` and ``outputs["padding_mask"]`` (skipped when ``all_ones_mask`` says the decoder
    built it with ``torch.ones_like``, i.e. xy planes only).

    ``fuse_sweep_backward=True`` (xy planes, mixture loss, one target view): the promise that ``outputs["logits"]`` /
    ``["sigma"]`` are consumed — as far as gradients go — by the trainer's plane sweep alone.  The sweep's backward kernel
    then applies this tail's backward on the values it holds anyway and writes the conv outputs' gradients directly
    (``ops.TailLink``, ``pd_plane_sweep_bwd_tail``); the tail's own backward kernel, which re-reads the [B,N,H,W]-sized
    gradients the sweep has just written, no longer runs.  NOT detected: another differentiable consumer of ``outputs["sigma"]``
    (a regulariser) — its gradient would be added, in sigma space, to one already in conv-output space; keep the flag off then."""
    mask = None if all_ones_mask else outputs["padding_mask"]
    logits, sigma, disp, depth, layers = ops.decoder_tail(dispconv_out, sigmaconv_out, mask, outputs["disp_layered"],
                                                          use_mixture_loss=use_mixture_loss,
                                                          fuse_sweep_backward=fuse_sweep_backward)
    outputs["logits"] = logits
    if use_mixture_loss:
        outputs["sigma"] = sigma
    shape = dispconv_out.shape
    if materialize_layers:
        pi, prob = layers(want_pi=use_mixture_loss, want_probability=True)
        outputs["probability"] = prob
        if use_mixture_loss:
            outputs["pi"] = pi
    else:
        dev, dt = dispconv_out.device, dispconv_out.dtype
        outputs["probability"] = LazyLayers(shape, lambda: layers(False, True)[1], dev, dt)
        if use_mixture_loss:
            outputs["pi"] = LazyLayers(shape, lambda: layers(True, False)[0], dev, dt)
    outputs["disp"] = disp
    outputs["depth"] = depth
    return outputs


def fused_plade_tail(outputs, conv0_out, conv_sigma_out=None, *, use_mixture_loss=True, materialize_layers=False):
    """Fills ``outputs`` with "logits", "dists", "sigma", "pi", "probability", "disp", "depth" as plade_net.py:309-340 does
    with ``render_probability`` (alpha compositing of the N-1 logit channels of ``conv0`` against the distances between the
    depth layers).  Reads ``outputs["disp_layered"]`` (per-plane levels with the learnt residual, or the dense map with
    ground planes)."""
    B, Nm1, H, W = conv0_out.shape
    logits, dists, sigma, disp, depth, layers = ops.plade_tail(conv0_out, conv_sigma_out, outputs["disp_layered"],
                                                                use_mixture_loss=use_mixture_loss)
    outputs["logits"] = logits
    outputs["dists"] = dists
    if use_mixture_loss:
        outputs["sigma"] = sigma
    shape = (B, Nm1 + 1, H, W)
    if materialize_layers:
        pi, prob = layers(want_pi=use_mixture_loss, want_probability=True)
        outputs["probability"] = prob
        if use_mixture_loss:
            outputs["pi"] = pi
    else:
        dev, dt = conv0_out.device, conv0_out.dtype
        outputs["probability"] = LazyLayers(shape, lambda: layers(False, True)[1], dev, dt)
        if use_mixture_loss:
            outputs["pi"] = LazyLayers(shape, lambda: layers(True, False)[0], dev, dt)
    outputs["disp"] = disp
    outputs["depth"] = depth
    return outputs
