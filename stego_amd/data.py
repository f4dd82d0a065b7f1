"""The cropped-dataset tree of the reference, reader and writer:

    {root}/cropped/{dataset}_{five|random}_crop_{ratio}/img/{split}/{i}.jpg
                                                       /label/{split}/{i}.png

written by `src/crop_datasets.py:76-123` (five crops per source image: image i * 5 + crop number; labels stored + 1 as uint8 PNG so
that "unlabelled" -1 becomes 0) and read by `CroppedDataset` `src/data.py:370-400` (target - 1, mask = target == -1).
torchvision is not part of this image: crops and tensor conversion are done with PIL / numpy / torch directly."""
import os
import random
from os.path import join

import numpy as np
import torch
from PIL import Image
from torch.utils.data import Dataset


def crop_dir(root, dataset_name, crop_type, crop_ratio):
    return join(root, "cropped", "{}_{}_crop_{}".format(dataset_name, crop_type, crop_ratio))


def five_crop_boxes(height, width, crop_h, crop_w):
    """(top, left) of the four corners and the centre crop, torchvision.transforms.functional.five_crop's order (tl, tr, bl, br,
    centre; centre = round((size - crop) / 2))."""
    if crop_w > width or crop_h > height:
        raise ValueError("Requested crop size {} is bigger than input size {}".format((crop_h, crop_w), (height, width)))
    ct, cl = int(round((height - crop_h) / 2.0)), int(round((width - crop_w) / 2.0))
    return [(0, 0), (0, width - crop_w), (height - crop_h, 0), (height - crop_h, width - crop_w), (ct, cl)]


def random_crop_boxes(height, width, crop_h, crop_w, seed, n=5):
    """crop_datasets.py:14-57: box i of image `seed` from hash((seed, i, 0)) / hash((seed, i, 1))."""
    if crop_w > width or crop_h > height:
        raise ValueError("Requested crop size {} is bigger than input size {}".format((crop_h, crop_w), (height, width)))
    return [(hash((seed, i, 0)) % (height - crop_h), hash((seed, i, 1)) % (width - crop_w)) for i in range(n)]


def write_cropped(root, dataset_name, crop_type, crop_ratio, split, items):
    """items: iterable of (img float [3,H,W] in [0,1], label int [H,W] with -1 = unlabelled).  Writes five crops per item exactly
    as RandomCropComputer.__getitem__ does (crop_datasets.py:112-123).  Returns the number of files per directory."""
    if crop_type not in ("five", "random"):
        raise ValueError('Unknown crop type {}'.format(crop_type))
    save = crop_dir(root, dataset_name, crop_type, crop_ratio)
    img_dir, label_dir = join(save, "img", split), join(save, "label", split)
    os.makedirs(img_dir, exist_ok=True)
    os.makedirs(label_dir, exist_ok=True)
    n = 0
    for item, (img, label) in enumerate(items):
        H, W = img.shape[1], img.shape[2]
        ch, cw = int(H * crop_ratio), int(W * crop_ratio)                        # _get_size, crop_datasets.py:62-68
        boxes = five_crop_boxes(H, W, ch, cw) if crop_type == "five" else random_crop_boxes(H, W, ch, cw, item)
        for crop_num, (t, l) in enumerate(boxes):
            img_num = item * 5 + crop_num
            im = img[:, t:t + ch, l:l + cw]
            lb = label[t:t + ch, l:l + cw]
            img_arr = im.mul(255).add(0.5).clamp(0, 255).permute(1, 2, 0).to("cpu", torch.uint8).numpy()
            label_arr = (lb + 1).to("cpu", torch.uint8).numpy()
            Image.fromarray(img_arr).save(join(img_dir, "{}.jpg".format(img_num)), "JPEG")
            Image.fromarray(label_arr).save(join(label_dir, "{}.png".format(img_num)), "PNG")
            n += 1
    return n


def to_tensor(pil_img):
    """PIL RGB -> float [3,H,W] in [0,1] (torchvision's ToTensor)."""
    return torch.from_numpy(np.asarray(pil_img, dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255)


def to_target_tensor(pil_label):
    """data.py ToTargetTensor: PIL label -> int64 [1,H,W]."""
    return torch.as_tensor(np.array(pil_label), dtype=torch.int64).unsqueeze(0)


def _seeded(fn, x, seed):
    """fn(x) with python's and torch's global generators reset to `seed` first: a random transform draws the same crop / flip for an image
    and for its label when both calls get the same seed (what reference src/data.py:388-394 does inline)."""
    random.seed(seed)
    torch.manual_seed(seed)
    return fn(x)


class CroppedDataset(Dataset):
    """Reader of the pre-cropped tree `cropped/{dataset}_{crop_type}_crop_{ratio}/{img,label}/{split}/{i}.{jpg,png}` that the reference's
    crop_datasets.py writes (format pinned by tests/golden/cropped_ref), with the reference dataset's constructor and item contract
    (src/data.py:370-400): item i -> (image, target [H, W] int64 with -1 = unlabelled, mask = target == -1); image and label go through
    their transforms under ONE random seed per item, drawn from numpy's global generator as the reference does (so that a run seeded
    like the reference sees the same augmentations)."""

    def __init__(self, root, dataset_name, crop_type, crop_ratio, image_set, transform=to_tensor, target_transform=to_target_tensor):
        super().__init__()
        self.dataset_name, self.split = dataset_name, image_set
        self.transform, self.target_transform = transform, target_transform
        self.root = crop_dir(root, dataset_name, crop_type, crop_ratio)
        self.img_dir, self.label_dir = (join(self.root, kind, image_set) for kind in ("img", "label"))
        self.num_images = len(os.listdir(self.img_dir))
        if self.num_images != len(os.listdir(self.label_dir)):
            raise AssertionError("%s: %d images but %d labels" % (self.root, self.num_images, len(os.listdir(self.label_dir))))

    def __len__(self):
        return self.num_images

    def __getitem__(self, index):
        seed = int(np.random.randint(2147483647))
        with Image.open(join(self.img_dir, "%d.jpg" % index)) as im:
            image = _seeded(self.transform, im.convert("RGB"), seed)
        with Image.open(join(self.label_dir, "%d.png" % index)) as lab:
            target = _seeded(self.target_transform, lab, seed) - 1      # the tree stores label + 1: 0 on disk = unlabelled
        return image, target.squeeze(0), target == -1
