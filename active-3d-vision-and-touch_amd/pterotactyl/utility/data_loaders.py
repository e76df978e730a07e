"""``pterotactyl.utility.data_loaders`` — the vision-trainer dataset and its wire formats.

``mesh_loader_vision`` (reference ``utility/data_loaders.py:132-258``) reads the on-disk layout the reference's
``download_data.sh`` / ``data_making`` scripts produce and yields the batch dict the trainer and ``prepare_mesh`` consume:

    <root>/point_cloud_info/<id>.npy             (30000, 3) float   surface samples of the object
    <root>/images_colourful/<id>.npy             (256, 256, 3) uint8 rendered image
    <root>/touch_charts/<id>/touch_charts.npy    50 grasps x 4 fingers x 25 vertices x (x, y, z, mask)
    <split file>                                 dict: set name -> object ids  (reference ``objects/data_split.npy``)

    batch = {"names": [(path, grasps)], "gt_points": (B,P,3) f32, "img": (B,3,256,256) f32 in [0,1] | (B,1) dummy,
             "touch_charts": (B,G,4,25,4) | (B,G,25,4) with ``finger`` | (B,1) dummy}

The data root is a directory argument (``args.data_root``, or the ``PTEROTACTYL_DATA`` environment variable) instead of
the reference's package-relative constants (:18-29) — the dataset is a download, not part of either tree.
``DevicePrefetcher`` is the host->HBM leg: pinned staging buffers and a copy stream one batch ahead of the compute
stream, so a 65 ms GPU step is not stalled by the 7.7 MB (no image) / 58 MB (image) batch upload.
"""
import os
import random
from glob import glob

import numpy as np
import torch


def data_root(args=None):
    root = getattr(args, "data_root", None) or os.environ.get("PTEROTACTYL_DATA")
    if not root:
        raise RuntimeError("a3vt: set args.data_root or PTEROTACTYL_DATA to the directory holding point_cloud_info/, "
                           "images_colourful/, touch_charts/ (the reference's download_data.sh output)")
    return root


def load_split(args=None):
    """The reference's ``objects/data_split.npy`` (:26-28): pickled dict of object-id lists per set."""
    path = getattr(args, "data_split", None) or os.path.join(data_root(args), "data_split.npy")
    return np.load(path, allow_pickle=True).item()


class mesh_loader_vision(object):
    def __init__(self, args, set_type="train"):
        self.args = args
        self.set_type = set_type
        root = data_root(args)
        self.point_dir = os.path.join(root, "point_cloud_info")
        self.touch_dir = os.path.join(root, "touch_charts")
        self.image_dir = os.path.join(root, "images_colourful")
        self.obj_dir = os.path.join(root, "object_info")
        split = load_split(args)
        training = set_type in ("recon_train", "auto_train")
        self.get_instance = self.get_training_instance if training else self.get_validation_instance
        names = [os.path.splitext(os.path.basename(f))[0] for f in glob(os.path.join(self.image_dir, "*.npy"))]
        if getattr(args, "limit_data", False):
            random.Random(0).shuffle(names)
            names = names[:2000]
        wanted = set(split[set_type])
        self.object_names = []
        seed = 0
        for n in names:
            if n in wanted and os.path.exists(os.path.join(self.point_dir, n + ".npy")) and \
                    os.path.exists(os.path.join(self.touch_dir, n)):
                for _ in range(1 if training else 5):   # validation objects appear 5x with different grasp subsets (:160-170)
                    self.object_names.append([n, seed])
                    seed += 1
        print(f"The number of {set_type} set objects found : {len(self.object_names)}")

    def __len__(self):
        return len(self.object_names)

    def get_training_instance(self, index):
        obj, _ = random.choice(self.object_names)
        count = random.choice(range(0, self.args.num_grasps + 1))
        order = list(range(50))
        random.shuffle(order)
        return obj, order[:count]

    def get_validation_instance(self, index):
        obj, seed = self.object_names[index]
        order = list(range(50))
        if getattr(self.args, "val_grasps", -1) >= 0 and self.args.eval:
            count = self.args.val_grasps
        else:
            count = random.Random(seed).choice(range(0, self.args.num_grasps + 1))
        random.Random(seed).shuffle(order)
        return obj, order[:count]

    def get_points(self, obj):
        samples = np.load(os.path.join(self.point_dir, obj + ".npy"))
        np.random.shuffle(samples)
        return torch.FloatTensor(samples[: self.args.number_points])

    def get_image(self, obj):
        if not self.args.use_img:
            return torch.empty((1))
        img = np.load(os.path.join(self.image_dir, obj + ".npy"))
        return torch.FloatTensor(img).permute(2, 0, 1) / 255.0

    def get_touch_info(self, obj, grasps):
        if not self.args.use_touch:
            return torch.ones((1))
        remaining = self.args.num_grasps - len(grasps)
        charts = torch.FloatTensor(np.load(os.path.join(self.touch_dir, obj, "touch_charts.npy"))).view(50, 4, 25, 4)
        if self.args.finger:
            return torch.cat((charts[grasps][:, 1], torch.zeros(remaining, 25, 4)))
        return torch.cat((charts[grasps], torch.zeros(remaining, 4, 25, 4)))

    def __getitem__(self, index):
        obj, grasps = self.get_instance(index)
        return {"names": (os.path.join(self.obj_dir, obj), grasps), "gt_points": self.get_points(obj),
                "img": self.get_image(obj), "touch_charts": self.get_touch_info(obj, grasps)}

    def collate(self, batch):
        out = {"names": [item["names"] for item in batch]}
        for key in ("gt_points", "img", "touch_charts"):
            out[key] = torch.stack([item[key] for item in batch])
        return out


class DevicePrefetcher:
    """Iterates a loader of batch dicts, uploading tensors to ``device`` on a side stream one batch ahead.

    Each upload goes through a pinned staging buffer (re-used per key and shape) with ``non_blocking`` copies on a copy
    stream; the compute stream waits on the copy's event only when it first touches the batch.  Non-tensor entries
    (``names``) pass through."""

    def __init__(self, loader, device):
        self.loader, self.device = loader, torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)
        self._pinned = {}

    def _upload(self, batch):
        out, ready = {}, torch.cuda.Event()
        with torch.cuda.stream(self.stream):
            for k, v in batch.items():
                if not isinstance(v, torch.Tensor) or v.is_cuda:
                    out[k] = v
                    continue
                key = (k, tuple(v.shape), v.dtype)
                stage = self._pinned.get(key)
                if stage is None:   # two staging buffers per entry: batch n+1 is staged while batch n's copy may be in flight
                    stage = self._pinned[key] = [[torch.empty(v.shape, dtype=v.dtype, pin_memory=True), None] for _ in range(2)]
                slot = stage[0]
                stage.reverse()
                if slot[1] is not None:
                    slot[1].synchronize()   # the copy that last read this staging buffer has finished
                slot[0].copy_(v)
                out[k] = slot[0].to(self.device, non_blocking=True)
                slot[1] = ready
            ready.record(self.stream)
        return out, ready

    def __iter__(self):
        it = iter(self.loader)
        nxt = None
        try:
            nxt = self._upload(next(it))
        except StopIteration:
            return
        while nxt is not None:
            cur, ready = nxt
            try:
                nxt = self._upload(next(it))
            except StopIteration:
                nxt = None
            torch.cuda.current_stream(self.device).wait_event(ready)
            for v in cur.values():
                if isinstance(v, torch.Tensor):
                    v.record_stream(torch.cuda.current_stream(self.device))
            yield cur

    def __len__(self):
        return len(self.loader)
