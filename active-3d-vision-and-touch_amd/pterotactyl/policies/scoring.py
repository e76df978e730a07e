"""Batched scoring of candidate touches — the forward-only caller of the hot path.

The reference's greedy search evaluates candidates one at a time: ``ActiveTouch.best_step`` (``policies/environment.py:
167-214``) calls ``compute_obs(actions)`` (:221-249) once per candidate action — a full ``Deformation`` forward, three
Chamfer draws (``get_score`` :252-257) and device-to-host copies each, up to 50 times per environment step.  Meshes in a
batch are independent through the forward pass and the Chamfer loss, so the K candidates x E environment elements can
ride in ONE batch of K*E meshes: one stack call per refinement stage (M = K*E*N rows fills the chip; E*N alone is 2-7 k
rows, less than one round of the MFMA kernel), one nearest-neighbour launch per direction, one host copy.  Results are
identical to the sequential loop given the same surface samples (tests/test_gpu_trainer.py::test_batched_scoring).

Image features depend only on the environment element, so the image encoders run once on the E images and their maps are
shared by the K candidates (the reference recomputes them per candidate).
"""
import torch

from ..utility import utils


def stack_charts(charts_list):
    """K chart dicts (each batched over E environment elements, ``prepare_mesh`` / ``environment.py:355-364`` format)
    -> one dict batched over K*E, candidate-major (index = k*E + e)."""
    return {key: torch.cat([c[key] for c in charts_list], dim=0) for key in charts_list[0]}


def score_actions(deform, img, charts_list, gt_points, faces, number_points, loss_coeff, repeat=3, samples=None):
    """Scores of K candidate touch configurations for E environment elements.

    deform       : ``vision.model.Deformation`` (used under ``no_grad``: activations are not saved)
    img          : (E,3,256,256) images or the (E,1) dummy of image-free models
    charts_list  : K chart dicts, each what ``get_inputs(actions)`` (:260-365) builds for one candidate action
    gt_points    : (E,P,3) ground-truth clouds, shared by the candidates of an element
    samples      : optional injected (face_idx, u, v), each (repeat,E,number_points), reused for every candidate
    returns      : ``score`` (K,E) = loss_coeff * Chamfer (``get_score`` :252-257), ``verts`` (K,E,N,3), ``mask`` (K,E,N,1)
    """
    K = len(charts_list)
    charts = stack_charts(charts_list)
    dev = charts["vision_charts"].device
    E = gt_points.shape[0]
    with torch.no_grad():
        if getattr(deform.args, "use_img", False):
            img = img.to(dev)
            gmaps = [m.repeat(K, 1, 1, 1) for m in deform.img_encoder_global(img)]
            lmaps = [m.repeat(K, 1, 1, 1) for m in deform.img_encoder_local(img)]
        else:
            gmaps, lmaps = [], []
        verts, mask = deform.deform_with_maps(charts, gmaps, lmaps)
        # the E ground-truth clouds are shared by the K candidates of each element (candidate-major batch: mesh k*E + e is
        # compared with gt[e]): uploaded, sorted and boxed once, not K times (a3vt_chamfer_fwd_shared)
        gt = gt_points.to(dev).to(torch.float32)
        if samples is not None:
            samples = tuple(s.repeat(1, K, 1) for s in samples)
        cd = utils.chamfer_distance(verts, faces, gt, num=number_points, repeat=repeat, samples=samples)
        score = loss_coeff * cd
    n = verts.shape[1]
    return score.view(K, E), verts.view(K, E, n, 3), mask.view(K, E, n, 1)


def best_actions(score, taken_mask):
    """Greedy choice of :174-180: per element the lowest-scoring candidate not already taken.
    score (K,E); taken_mask (E,K) non-zero where the action was performed before.  Returns (E,) indices."""
    s = score.t().clone()
    s[taken_mask.to(s.device) != 0] = float("inf")
    return torch.argmin(s, dim=1)
