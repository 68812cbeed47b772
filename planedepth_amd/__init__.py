"""planedepth_amd — MI355X (gfx950) implementation of PlaneDepth's photometric-reconstruction hot path.

Importing the package does not need a GPU or the built library; calling any operator does (and raises otherwise).
"""
from . import decoder_tail, layers, ops, synthetic, trainer_path  # noqa: F401
from .decoder_tail import fused_decoder_tail  # noqa: F401
from .layers import (SSIM, BackprojectDepth, HomographyWarp, Project3D, disp_to_depth,  # noqa: F401
                     get_smooth_loss_disp, multimodal_loss)
from .trainer_path import (add_flip_right_inputs, compute_losses, compute_reprojection_loss, generate_post_process_disp,  # noqa: F401
                           patch_trainer, pred_novel_images, pred_self_images)

__version__ = "0.2.6"   # = pd_version() 260 of the library (tests/test_capi.py checks that they agree)
