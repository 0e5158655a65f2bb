"""scade_amd -- MI355X-native implementation of SCADE's per-ray rendering hot path.

Operator API mirrors the reference (mikacuy/scade):
  scade_amd.run_nerf_helpers  <->  model/run_nerf_helpers.py
  scade_amd.rendering          <->  the render operators of run_scade_scannet.py / run_scade_wild.py
All arithmetic is executed by libscade_hip.so (hand-written HIP for gfx950).
"""
from .run_nerf_helpers import (NeRF, DenseLayer, Embedder, get_embedder, get_rays, get_ray_dirs, get_ray_batch,
                               select_coordinates, sample_pdf, sample_pdf_joint,
                               sample_pdf_return_u, sample_pdf_joint_return_u, img2mse,
                               img2mse_masked, mse2psnr, to8b, to16b, compute_space_carving_loss)
from .rendering import (batchify, run_network, batchify_rays, render, render_hyp, compute_weights,
                     raw2depth, raw2outputs, perturb_z_vals, render_rays, make_network_query_fn)

__all__ = [n for n in dir() if not n.startswith("_")]
