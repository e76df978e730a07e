"""Drop-in for ``pterotactyl/reconstruction/vision/train.py`` — the vision reconstruction trainer ``Engine``.

Same constructor argument (an argparse ``Namespace`` or any object with the same attributes, README.md:143-170),
same public methods and attributes (``__call__``, ``get_loaders``, ``train``, ``validate``, ``save``, ``load``,
``check_values``; ``encoder, optimizer, mesh_info, initial_mesh, n_vision_charts, epoch, best_loss,
current_loss, checkpoint_dir, results_dir``), same checkpoint files (``<ckpt>/model``, ``/optim``, ``/epoch.npy``,
``/config.json``; reference :211-267) and the same loss (``loss_coeff * chamfer.mean()`` for training :144,
``loss_coeff * chamfer.sum()`` accumulated over examples for validation :184-200).

Differences: the hot loop never blocks on the device except every ``log_interval`` iterations (the reference
calls ``loss.item()`` every step, :151-153); tensorboard / submitit are optional; early stop raises
``StopIteration`` instead of ``exit()`` (:284); with ``WORLD_SIZE`` > 1 each rank trains on its own shard of every
epoch (``distributed.rank_plan``: same permutation on all ranks, every ``world``-th item; python / numpy / torch re-seeded
with ``seed + rank`` after the weight broadcast so grasp choices and surface samples differ per rank), gradients are
averaged over RCCL from one flat bucket in two chunks (the one that is final after stage 2's backward starts early and
overlaps with the rest of the backward), BatchNorm running statistics follow rank 0 at validation / checkpoint time, and the
validation loss and example counts are summed over ranks whatever the loader (``a3vt_amd.distributed``).  Datasets: ``get_loaders`` reads the reference's on-disk
layout through ``utility/data_loaders.py`` (``args.data_root`` / ``PTEROTACTYL_DATA``; worker count ``args.num_workers``,
default 16 as the reference); or pass ``loaders=(train_loader, valid_loader)`` (e.g. ``a3vt_amd.synthetic.SyntheticLoader``).
Batches are uploaded one step ahead on a copy stream (``data_loaders.DevicePrefetcher``).
"""
import os

import numpy as np
import torch
import torch.optim as optim

from . import model
from ...utility import data_loaders, utils
from .... import distributed as adist
from .... import ops as _ops
from .... import optim as a3vt_optim

try:
    from torch.utils.tensorboard import SummaryWriter
except Exception:  # tensorboard is optional
    class SummaryWriter:
        def __init__(self, *a, **k):
            pass

        def add_scalars(self, *a, **k):
            pass


def pretrained_location(args):
    """Directory of the pretrained reconstruction model for ``args`` — the reference's choice (:218-241): one of four
    sub-directories of the ``pretrained`` package picked by ``(use_img, finger)``,
    ``<root>/reconstruction/vision/{v_t_p, v_t_g, t_p, t_g}/``.  ``<root>`` = ``args.pretrained_root``, else
    ``$PTEROTACTYL_PRETRAINED``, else an installed ``pterotactyl.pretrained`` package, else ``pretrained/`` next to this
    package.  ``args.pretrained_location`` (a directory holding config.json + model) overrides the whole rule."""
    explicit = getattr(args, "pretrained_location", None)
    if explicit:
        return explicit
    root = getattr(args, "pretrained_root", None) or os.environ.get("PTEROTACTYL_PRETRAINED")
    if not root:
        try:
            from pterotactyl import pretrained as _pre      # the reference's own package, when it is installed
            root = os.path.dirname(_pre.__file__)
        except Exception:
            root = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "pretrained")
    sub = ("v_t_p" if args.finger else "v_t_g") if args.use_img else ("t_p" if args.finger else "t_g")
    return os.path.join(root, "reconstruction", "vision", sub) + os.sep


class Engine:
    def __init__(self, args, loaders=None, template="vision_charts"):
        np.random.seed(args.seed)
        torch.manual_seed(args.seed)
        self.epoch = 0
        self.best_loss = 10000
        self.args = args
        self.last_improvement = 0
        self.vision_chart_location = template  # packaged atlas (reference: objects/vision_charts.obj) or an OBJ path
        self._loaders = loaders
        self.log_interval = getattr(args, "log_interval", 10)
        self.results_dir = os.path.join("results", args.exp_type, args.exp_id)
        os.makedirs(self.results_dir, exist_ok=True)
        self.checkpoint_dir = os.path.join("experiments/checkpoint/", args.exp_type, args.exp_id)
        os.makedirs(self.checkpoint_dir, exist_ok=True)
        if not self.args.eval:
            utils.save_config(self.checkpoint_dir, args)
        self.rank, self.world, self.local_rank = adist.init_from_env()
        self.bucket = None

    def setup(self):
        """Everything ``__call__`` does before touching data (reference :52-64)."""
        self.mesh_info, self.initial_mesh = utils.load_mesh_vision(self.args, self.vision_chart_location)
        self.n_vision_charts = self.initial_mesh.shape[0]
        self.encoder = model.Deformation(self.mesh_info, self.initial_mesh, self.args).to(self.initial_mesh.device)
        adist.broadcast_parameters(self.encoder)
        adist.broadcast_buffers(self.encoder)
        if self.world > 1:   # identical weights everywhere; from here on every rank draws its own random numbers
            adist.seed_rank(self.args.seed, self.rank)
        if not self.args.eval:
            params = list(self.encoder.parameters())
            # gradients that are final once the backward pass has left stage 2: their all-reduce starts as soon as the last of
            # them has arrived (two chunks, FlatGradBucket)
            early = [p for name in ("mesh_deform_2", "img_encoder_local") if hasattr(self.encoder, name)
                     for p in getattr(self.encoder, name).parameters()] if getattr(self.encoder, "num_stages", 3) > 1 else []
            if getattr(self, "bucket", None) is not None:
                self.bucket.close()
            # args.force_collectives (this package's knob, default off): issue the gradient all-reduces on a process group of
            # one rank as well — the RCCL path on a single GPU (tests)
            self.bucket = adist.FlatGradBucket(params, early=early, force_collectives=getattr(self.args, "force_collectives", False))
            # optim.Adam(params, lr, weight_decay=0) of the reference (:64) with its step as one launch of the library (a3vt_amd/optim.py;
            # args.library_adam = False — this package's knob — keeps torch's fused kernel: same state, same checkpoints)
            self.optimizer = a3vt_optim.make_adam(params, self.args.lr, library=getattr(self.args, "library_adam", True))

    def __call__(self):
        self.setup()
        writer = SummaryWriter(os.path.join("experiments/tensorboard/", self.args.exp_type))
        train_loader, valid_loaders = self.get_loaders()
        if self.args.eval:
            self.load()
            with torch.no_grad():
                self.validate(valid_loaders, writer)
            return self.current_loss
        self.load()
        for epoch in range(self.epoch, self.args.epochs):
            self.epoch = epoch
            if hasattr(getattr(train_loader, "sampler", None), "set_epoch"):
                train_loader.sampler.set_epoch(epoch)
            self.train(train_loader, writer)
            with torch.no_grad():
                self.validate(valid_loaders, writer)
            self.check_values()
        return self.best_loss

    def get_loaders(self):
        if self._loaders is not None:
            return self._loaders
        from torch.utils.data import DataLoader   # dataset classes: utility/data_loaders.py on args.data_root
        workers = getattr(self.args, "num_workers", 16)
        train_loader = ""
        if not self.args.eval:
            train_data = data_loaders.mesh_loader_vision(self.args, set_type="recon_train")
            # one shard per rank and epoch (world 1: the whole set, shuffled — the reference's shuffle=True, :96-103)
            sampler = adist.ShardSampler(len(train_data), self.rank, self.world, self.args.seed, shuffle=True, pad=True)
            train_loader = DataLoader(train_data, batch_size=self.args.batch_size, sampler=sampler, num_workers=workers,
                                      collate_fn=train_data.collate, pin_memory=True)
        valid_data = data_loaders.mesh_loader_vision(self.args, set_type="test" if self.args.eval else "valid")
        vsampler = adist.ShardSampler(len(valid_data), self.rank, self.world, self.args.seed, shuffle=False, pad=False)
        valid_loader = DataLoader(valid_data, batch_size=self.args.batch_size, sampler=vsampler, num_workers=workers,
                                  collate_fn=valid_data.collate, pin_memory=True)
        return train_loader, valid_loader

    def train_step(self, img, charts, gt_points, samples=None):
        """One optimisation step on device tensors; returns the (device) scalar loss.  No host sync.
        ``samples``: optional injected (face_idx, u, v) surface draws (parity tests); default = the Philox stream."""
        self.bucket.zero()
        verts = self.encoder(img, charts)[0]
        loss = utils.chamfer_distance(verts, self.mesh_info["faces_i32"], gt_points, num=self.args.number_points,
                                      samples=samples)
        loss = self.args.loss_coeff * loss.mean()
        loss.backward()
        self.bucket.all_reduce_mean()
        self.optimizer.step()
        return loss.detach()

    def train(self, data, writer):
        total_loss = torch.zeros((), device=self.initial_mesh.device)
        iterations = 0
        self.encoder.train()
        dev = self.initial_mesh.device
        for k, batch in enumerate(data_loaders.DevicePrefetcher(data, dev)):
            img = batch["img"].to(dev, non_blocking=True)
            gt_points = batch["gt_points"].to(dev, non_blocking=True)
            with torch.no_grad():
                charts = model.prepare_mesh(batch, self.initial_mesh, self.args)
            loss = self.train_step(img, charts, gt_points)
            total_loss += loss
            iterations += 1
            if self.log_interval and k % self.log_interval == 0 and self.rank == 0:
                print(f"Train || Epoch: {self.epoch}, loss: {loss.item():.2f}, b_ptp:  {self.best_loss:.2f}")
        if iterations:
            writer.add_scalars("train_loss", {self.args.exp_id: total_loss.item() / iterations}, self.epoch)

    def validate(self, valid_loader, writer):
        total_loss = torch.zeros((), device=self.initial_mesh.device)
        adist.broadcast_buffers(self.encoder)   # BatchNorm running statistics: every shard is scored with rank 0's (the saved ones)
        self.encoder.eval()
        num_examples = 0
        dev = self.initial_mesh.device
        for v, batch in enumerate(data_loaders.DevicePrefetcher(valid_loader, dev)):
            img = batch["img"].to(dev, non_blocking=True)
            gt_points = batch["gt_points"].to(dev, non_blocking=True)
            charts = model.prepare_mesh(batch, self.initial_mesh, self.args)
            verts = self.encoder(img, charts)[0]
            loss = utils.chamfer_distance(verts, self.mesh_info["faces_i32"], gt_points, num=self.args.number_points)
            total_loss += self.args.loss_coeff * loss.sum()
            num_examples += float(img.shape[0])
        if self.world > 1:
            # Always: with a sharded loader the ranks hold disjoint parts of the set; with an injected loader (no
            # ``sampler.plan``) they hold copies of it scored with different surface samples (per-rank seeds) — either way
            # the sums over ranks give every rank the SAME score, so check_values() takes the same decision everywhere
            # (a rank that stopped or saved alone would leave the others waiting in the gradient all-reduce).
            count = torch.tensor(float(num_examples), device=dev)
            adist.all_reduce_sum_(total_loss, count)
            num_examples = count.item()
        total = (total_loss / max(num_examples, 1.0)).item()
        if self.rank == 0:
            print("*******************************************************")
            print(f"Validation Accuracy: {total}")
            print("*******************************************************")
        if not self.args.eval:
            writer.add_scalars("valid_ptp", {self.args.exp_id: total}, self.epoch)
        self.current_loss = total

    def save(self):
        if self.rank != 0:
            return
        torch.save(self.encoder.state_dict(), self.checkpoint_dir + "/model")
        torch.save(self.optimizer.state_dict(), self.checkpoint_dir + "/optim")
        np.save(self.checkpoint_dir + "/epoch.npy", np.array([self.epoch + 1]))

    def load(self):
        if self.args.eval and getattr(self.args, "pretrained", False):
            location = pretrained_location(self.args)
            if not os.path.exists(os.path.join(location, "model")):
                raise FileNotFoundError(
                    f"a3vt: no pretrained model at {location} (the reference fetches its weights with download_models.sh into "
                    "pterotactyl/pretrained/; point args.pretrained_root / $PTEROTACTYL_PRETRAINED at that directory, or "
                    "args.pretrained_location at one holding config.json + model)")
            vision_args, _ = utils.load_model_config(location)
            self.mesh_info, self.initial_mesh = utils.load_mesh_vision(vision_args, self.vision_chart_location)
            self.n_vision_charts = self.initial_mesh.shape[0]
            self.encoder = model.Deformation(self.mesh_info, self.initial_mesh, vision_args).to(self.initial_mesh.device)
            self.encoder.load_state_dict(torch.load(os.path.join(location, "model"), map_location=self.initial_mesh.device))
            _ops.invalidate_bf16_copies()       # (load_state_dict's copy_ moves _version already; explicit for loaders that do not)
            return
        try:
            dev = self.initial_mesh.device
            self.encoder.load_state_dict(torch.load(self.checkpoint_dir + "/model", map_location=dev))
            _ops.invalidate_bf16_copies()
            self.optimizer.load_state_dict(torch.load(self.checkpoint_dir + "/optim", map_location=dev))
            self.epoch = int(np.load(self.checkpoint_dir + "/epoch.npy")[0])
        except (FileNotFoundError, AttributeError):
            return

    def check_values(self):
        if self.best_loss >= self.current_loss:
            improvement = self.best_loss - self.current_loss
            if self.rank == 0:
                print(f"Saving with {improvement:.3f} improvement in Chamfer Distance on Validation Set ")
            self.best_loss = self.current_loss
            self.last_improvement = 0
            self.save()
        else:
            self.last_improvement += 1
            if self.last_improvement >= self.args.patience:
                raise StopIteration(f"Over {self.args.patience} steps since last improvement")
